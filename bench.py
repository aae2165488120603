#!/usr/bin/env python3
"""Headline benchmark: interpolated 1280x720 frames/sec (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one synthetic Adobe240-shaped 1280x720 frame pair (zero-padded to 736x1280 in
normalised space, inputs resident in HBM) -> 7 intermediate frames t = 1/8..7/8
(configs[1]: superslomo_original.ini inference).  Stage 1 runs once per pair, the 7 t values
are batched through stage 2.  N > 1: every rank processes its own pairs (weak scaling, no
data-path collective); value = all ranks' frames / max-over-ranks time.

Extra objects on the JSON line:
  roofline      dominant kernel family = the 48 convolution launches of a step (conv16_kernel + conv16_ups_kernel in the
                default precision mode f16f8: one fp16 MFMA + two block-scaled fp8 MFMAs per product).  achieved =
                algorithmic FLOP per step (5.855 TFLOP / pair at 736x1280, 7 t, stage 1 hoisted) / the summed
                duration of those launches, measured with HIP events on the launch stream in a single-stream region run right
                after the timed region; peak = the dense fp16 MFMA peak (2.5 PFLOP/s; 157.3 TFLOP/s in mode f32);
                mfma_issue_frac = issued fp16-MFMA units / peak; traffic = HBM bytes of those launches from rocprofv3 --pmc
                passes (profiles/*_pmc_traffic_summary.json).
  roofline_warp the HBM-bound gather kernels (compute_inputs + synthesis): algorithmic bytes
                (104 + 72 B/px per t) / their event-timed duration; peak 8 TB/s.
  cpu_baseline  the CPU oracle (torch CPU fp32 ops, pinned to the reference by golden fixtures),
                timed on this host on a bounded sample: 1 pair x 1 intermediate, reference-style
                loop (stage 1 recomputed per t).  A reported baseline, not the target.
  parity        max|HIP - oracle| over that same full-size frame (bar: 1e-3).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
for _p in (ROOT, PKG, os.path.join(PKG, "scripts")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

H_IN, W_IN, N_T = 720, 1280, 7
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_F16_MFMA_TFLOPS = 2500.0     # dense fp16/bf16 MFMA
PEAK_HBM_GBS = 8000.0
DTYPE_NOTE = {"f32": "f32", "f16": "f16 (f32 accumulate)",
              "f16x3": "f32 via 3x f16 MFMA on hi/lo-split operands (f32 accumulate)",
              "f16f8": "f32 via 1x f16 MFMA + 2x block-scaled fp8 MFMA (compensation products) on hi/lo-split operands (f32 accumulate)"}


def conv_flops_per_pair(h, w, n_t):
    from ssm_amd.weights import unet_layers
    scale = {"conv1": 1, "conv2": 2, "conv3": 4, "conv4": 8, "conv5": 16, "conv6": 32, "conv7": 16, "conv8": 8,
             "conv9": 4, "fuse_": 1, "final": 1}

    def stage(st):
        tot = 0.0
        for name, cin, cout, k in unet_layers(st, True):
            s = 2 if name.startswith("conv10") else 1 if name.startswith("conv11") else \
                [v for p, v in scale.items() if name.startswith(p)][0]
            tot += 2.0 * (h // s) * (w // s) * cin * cout * k * k
        return tot
    return stage(1) + n_t * stage(2)


TRAIN_DTYPE_NOTE = {
    "f16f8": "f32 parameters/activations/gradients; products via 1x f16 + 2x block-scaled fp8 MFMA (forward, data gradients) and 3x bf16 "
             "MFMA on hi/lo-split operands (weight gradients), f32 accumulate",
    "f32": "f32 (fp32 MFMA)"}


def train_bench(args):
    """BASELINE configs[2]: superslomo_original.ini training (FREEZE=FALSE), batch 16 = 2 samples per GPU x 8 of 352x352
    crops, t = i/8 per sample; forward + hand-written backward + gradient all-reduce (RCCL) + Adam.  A step = one
    batch of 2 samples per rank; value = samples/s over all ranks (weak scaling)."""
    from ssm_amd import dist as sdist
    from ssm_amd.config import load_config, synthetic_weight_overrides
    from ssm_amd.engine import KernelTimer, UNetPlan
    from ssm_amd.training import Trainer
    from ssm_amd.weights import synthetic_frames, synthetic_state_dict
    from models.superslomo_r import FullModel

    rank, local_rank, world = sdist.env_world()
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sdist.init("nccl")
    ov = synthetic_weight_overrides()
    ov[("STAGE1", "FREEZE")] = "FALSE"          # documented override: the shipped ini freezes both stages
    ov[("STAGE2", "FREEZE")] = "FALSE"
    cfg = load_config("superslomo_original.ini", ov)
    model = FullModel(cfg)
    model.stage1_model.load_state_dict(synthetic_state_dict(1))
    model.stage2_model.load_state_dict(synthetic_state_dict(2))
    if not args.no_perceptual:      # VGG16 conv4_3 feature loss (lambda_p = 20) with synthetic VGG weights: same FLOPs as the
        from ssm_amd.perceptual import VGGFeatures, synthetic_vgg_state_dict      # pretrained net the reference downloads
        model.loss.load_vgg16(synthetic_vgg_state_dict())
    model = model.to(dev).train()
    trainer = Trainer(model, cfg)
    B, S = 2, 352
    clips = torch.cat([synthetic_frames(3, S, S, seed=100 + 2 * rank + i) for i in range(B)], 0).to(dev)   # [B,3,3,S,S]
    xin, tgt = clips[:, [0, 2]].contiguous(), clips[:, 1:2].contiguous()
    t = torch.tensor([(3 + i + rank) % 7 + 1 for i in range(B)], dtype=torch.float32, device=dev).view(B, 1, 1, 1, 1) / 8.0
    ar = [0.0]

    def step():
        trainer.train_step(xin, tgt, t)
        ar[0] += trainer.last_allreduce_s

    def sync():
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    ar[0] = 0.0
    elapsed = sdist.timed_steps(step, args.steps, 0, sync)
    ar_ms = 1e3 * ar[0] / args.steps
    timer = KernelTimer()
    UNetPlan.timer = timer
    if not args.no_perceptual:
        VGGFeatures.timer = timer
    for _ in range(3):
        step()
    sync()
    UNetPlan.timer = None
    if not args.no_perceptual:
        VGGFeatures.timer = None
    out = {"metric": "training samples/sec (352x352 crops, forward+backward+Adam)", "value": round(B * world * args.steps / elapsed, 3),
           "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": TRAIN_DTYPE_NOTE.get(os.environ.get("SSM_TRAIN_PRECISION", "f16f8"), "f32"), "data": "synthetic",
           "config": {"workload": "superslomo_original.ini training, FREEZE=FALSE, %d samples/GPU of 352x352, %d GPU(s); "
                                  "losses: L1 reconstruction + 4 L1 warp terms + %s" % (B, world, "VGG16 conv4_3 perceptual term OFF" if args.no_perceptual else
                                                                                  "VGG16 conv4_3 perceptual term (synthetic VGG weights)"),
                      "global_batch": B * world},
           "allreduce": {"bytes": trainer.allreduce.bytes, "ms_per_step": round(ar_ms, 3)},
           "host_enqueue_ms_per_step": round(1e3 * sdist.timed_steps.last_enqueue_s / args.steps, 3)}
    if rank == 0:
        summ = timer.summary()
        out["time_split_ms_per_step"] = {fam: round(d["ms"] / 3, 3) for fam, d in summ.items()}
        out["tflops"] = {fam: round(d["flops"] / d["ms"] / 1e9, 1) for fam, d in summ.items() if d["flops"] > 0}
        if args.detail:
            det = {fam: {n: {"ms_per_step": v[0] / 3, "tflops": (v[1] / v[0] / 1e9 if v[0] > 0 else 0.0)}
                         for n, v in d["by_name"].items()} for fam, d in summ.items()}
            with open(args.detail, "w") as f:
                json.dump(det, f, indent=1)
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def recurrent_bench(args):
    """BASELINE configs[3]: superslomo_recurrent.ini (N_FRAMES=4, BOTTLENECK=CLSTM) at 720p.  A step = one clip of 4
    synthetic frames -> the 7 intermediates between its two middle frames; stage 1 (3 windows + its ConvBLSTM) runs
    once per clip, stage 2 encodes 3 windows x 7 t, runs its ConvBLSTM over them and decodes the middle window.
    Clips shard across ranks like frame pairs (no collective)."""
    from ssm_amd import dist as sdist
    from ssm_amd.config import load_config, synthetic_weight_overrides
    from ssm_amd.engine import KernelTimer, UNetPlan
    from ssm_amd.weights import synthetic_frames, synthetic_state_dict
    from models.superslomo_r import FullModel

    rank, local_rank, world = sdist.env_world()
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sdist.init("nccl")
    cfg = load_config("superslomo_recurrent.ini", synthetic_weight_overrides())
    kind = cfg.get("STAGE1", "BOTTLENECK")
    model = FullModel(cfg)
    p1, p2 = synthetic_state_dict(1, bottleneck=kind), synthetic_state_dict(2, bottleneck=kind)
    model.stage1_model.load_state_dict(p1)
    model.stage2_model.load_state_dict(p2)
    model.precision = args.precision or "f16f8"
    model = model.to(dev).eval()
    n_frames = cfg.getint("TRAIN", "N_FRAMES")
    clips = [synthetic_frames(n_frames, H_IN, W_IN, seed=42 + 2 * rank + i).to(dev) for i in range(2)]
    ts = [i / 8.0 for i in range(1, N_T + 1)]
    k = [0]

    def step():
        k[0] += 1
        return model.interpolate_windows(clips[k[0] % 2], ts)

    def sync():
        torch.cuda.synchronize(dev)

    elapsed = sdist.timed_steps(step, args.steps, args.warmup, sync)
    timer = KernelTimer()
    UNetPlan.timer = timer
    for _ in range(3):
        step()
    sync()
    UNetPlan.timer = None
    out = {"metric": "interpolated 1280x720 frames/sec (recurrent, N_FRAMES=%d)" % n_frames,
           "value": round(N_T * world * args.steps / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NOTE[model.precision], "data": "synthetic",
           "config": {"workload": "superslomo_recurrent.ini inference (BOTTLENECK=%s, parity unpinned: restated cells): synthetic "
                                  "%d-frame 1280x720 clip -> 7 intermediates of the middle window, stage 1 once per clip" % (kind, n_frames),
                      "precision": model.precision, "clips_per_step": 1, "frames_per_step": N_T}}
    if rank == 0:
        summ = timer.summary()
        conv = summ["conv"]
        out["time_split_ms_per_step"] = {"conv": round(conv["ms"] / 3, 3),
                                         "conv6_recurrent": round(sum(v[0] for n, v in conv["by_name"].items() if "conv6(" in n) / 3, 3)}
        out["tflops"] = round(conv["flops"] / conv["ms"] / 1e9, 1)
        if not args.no_cpu_baseline:
            from oracle import ssm_oracle as O
            xs = clips[k[0] % 2].cpu()
            t0 = time.time()
            want, _ = O.full_model_infer_windows(p1, p2, xs, torch.full((1, n_frames - 1, 1, 1, 1), 0.5), True, kind)
            cpu_s = time.time() - t0
            got = model.interpolate_windows(clips[k[0] % 2], [0.5])
            out["parity"] = {"max_abs_vs_oracle": float((got.cpu() - want).abs().max()), "tolerance": 1e-3, "frames": 1,
                             "size": "%dx%d" % tuple(xs.shape[-2:])}
            out["cpu_baseline"] = {"value": round(1.0 / cpu_s, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": "1 clip x 1 intermediate (t=0.5), torch CPU fp32 oracle; %.1f s" % cpu_s}
        if args.detail:
            with open(args.detail, "w") as f:
                json.dump({n: {"ms_per_step": v[0] / 3, "tflops": (v[1] / v[0] / 1e9 if v[0] > 0 else 0.0)}
                           for n, v in conv["by_name"].items()}, f, indent=1)
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true", help="skip the HIP-event brackets (roofline = null)")
    ap.add_argument("--precision", default=None, choices=["f32", "f16x3", "f16", "f16f8"],
                    help="conv arithmetic (default: models.superslomo_r.DEFAULT_PRECISION / $SSM_PRECISION)")
    ap.add_argument("--mode", default="infer", choices=["infer", "train", "recurrent"],
                    help="infer = the headline (BASELINE configs[1]); train = configs[2]: training step on 352x352 crops, "
                         "2 samples per GPU, gradient all-reduce over RCCL; recurrent = configs[3]: superslomo_recurrent.ini "
                         "(N_FRAMES=4, ConvBLSTM bottleneck) at 720p")
    ap.add_argument("--size", default="720p", choices=["720p", "4k"],
                    help="720p = BASELINE configs[1] (the headline); 4k = configs[4] shape (3840x2160, use --precision f16)")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams (= frame pairs in flight) per GPU")
    ap.add_argument("--no-perceptual", action="store_true", help="--mode train: leave the VGG16 perceptual loss term out")
    ap.add_argument("--graphs", type=int, default=0, help="1: replay each pair's launch sequence from a captured HIP graph")
    ap.add_argument("--detail", default=None, help="write the per-launch event-timer table (JSON) to this path")
    args = ap.parse_args()

    from ssm_amd import dist as sdist
    from ssm_amd.config import load_config, synthetic_weight_overrides
    from ssm_amd.engine import KernelTimer, UNetPlan
    from ssm_amd.weights import synthetic_frames, synthetic_state_dict
    from models.superslomo_r import FullModel

    if args.mode == "recurrent":
        return recurrent_bench(args)
    if args.mode == "train":
        return train_bench(args)
    rank, local_rank, world = sdist.env_world()
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py measures the HIP path; no GPU visible"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sdist.init("nccl")

    cfg = load_config("superslomo_original.ini", synthetic_weight_overrides())   # documented override: no weights ship
    model = FullModel(cfg)
    sd1, sd2 = synthetic_state_dict(1), synthetic_state_dict(2)
    model.stage1_model.load_state_dict(sd1)
    model.stage2_model.load_state_dict(sd2)
    model = model.to(dev).eval()
    import models.superslomo_r as ssm_r
    precision = args.precision or os.environ.get("SSM_PRECISION", ssm_r.DEFAULT_PRECISION)
    model.precision = precision

    h_in, w_in = (H_IN, W_IN) if args.size == "720p" else (2160, 3840)
    x = synthetic_frames(2, h_in, w_in, seed=42 + rank)          # [1,2,3,736,1280], normalised, zero-padded
    Hp, Wp = x.shape[-2:]
    img6 = x.reshape(1, 6, Hp, Wp).to(dev)
    t_dev = torch.tensor([i / 8.0 for i in range(1, N_T + 1)], dtype=torch.float32, device=dev)
    from ssm_amd.engine import PairPipeline
    sd1d = {k: v.detach() for k, v in model.stage1_model.state_dict().items()}
    sd2d = {k: v.detach() for k, v in model.stage2_model.state_dict().items()}
    pipe = PairPipeline(sd1d, sd2d, N_T, Hp, Wp, dev, True, precision, args.streams, graphs=bool(args.graphs))

    def step():                 # one pair -> 7 frames; consecutive steps alternate between the streams
        pipe.submit(img6, t_dev, want_aux=False)

    def sync():
        torch.cuda.synchronize(dev)

    # ---- timed region: `--streams` pairs in flight, no event brackets -------------------------------
    for _ in range(args.warmup):
        step()
    sync()
    elapsed = sdist.timed_steps(step, args.steps, 0, sync)

    # ---- roofline region: the same number of steps on ONE stream with HIP-event brackets around every
    # launch (with several pairs in flight the per-kernel spans overlap and cannot be attributed) --------
    timer = None
    if not args.no_kernel_timers and rank == 0:
        solo = pipe.engines[0]
        for _ in range(2):
            solo.run(img6, t_dev, want_aux=False)
        sync()
        timer = KernelTimer()
        UNetPlan.timer = timer
        t0 = time.perf_counter()
        for _ in range(args.steps):
            solo.run(img6, t_dev, want_aux=False)
        sync()
        solo_ms = 1e3 * (time.perf_counter() - t0) / args.steps
        UNetPlan.timer = None

    frames = N_T * args.steps * world
    value = frames / elapsed
    out = {
        "metric": "interpolated 1280x720 frames/sec" if args.size == "720p" else "interpolated 3840x2160 frames/sec", "value": round(value, 3), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE_NOTE[precision], "data": "synthetic",
        "config": {"precision": precision, "workload": "superslomo_original.ini inference: synthetic %dx%d pair (padded %dx%d) -> 7 "
                               "intermediates t=i/8, stage 1 once per pair, random-init (deterministic) weights"
                               % (w_in, h_in, Wp, Hp),
                   "pairs_per_step": 1, "frames_per_step": N_T, "streams_per_gpu": args.streams,
                   "parallelism": "pairs sharded, %d rank(s); %d pair(s) in flight per GPU on separate HIP streams"
                                  % (world, args.streams)},
    }

    if timer is not None and rank == 0:
        summ = timer.summary()
        flops_step = conv_flops_per_pair(Hp, Wp, N_T)
        conv = summ["conv"]
        conv_ms_step = conv["ms"] / args.steps
        ach = flops_step / (conv_ms_step * 1e-3) / 1e12
        traffic = None
        pmc_file, pmc_key = {"f32": ("r1b_pmc_traffic_summary.json", "conv_mfma_kernel"),
                             "f16x3": ("r1k_pmc_traffic_summary.json", "conv16_kernel"),
                             "f16f8": ("r1q_pmc_traffic_summary.json", "conv16_kernel")}.get(precision, (None, None))
        pmc = os.path.join(ROOT, "profiles", pmc_file) if pmc_file else ""
        if pmc and os.path.exists(pmc):   # HBM bytes per step from rocprofv3 --pmc passes (tools/pmc_traffic.sh), not live
            pj = json.load(open(pmc))
            traffic = sum(pj.get(k, {}).get("hbm_bytes_per_step", 0.0) for k in (pmc_key, "conv16_ups_kernel", "conv16_multi_kernel"))
        if precision == "f32":
            kname, peak, mfma_per_prod = "conv_mfma_kernel<*> (fp32 v_mfma_f32_32x32x2_f32)", PEAK_F32_MFMA_TFLOPS, 1
        else:
            kname, peak = "conv16_kernel<*> + conv16_multi_kernel<*> + conv16_ups_kernel<*> (v_mfma_f32_32x32x16_f16)", PEAK_F16_MFMA_TFLOPS
            # fp16-MFMA units per algorithmic product: f16x3 = 3; f16f8 = 1 fp16 + 2 block-scaled fp8 steps that cover 4x the K in the
            # cycles of one fp16 step (= 1/4 unit each, up to 4/3 padding on 3-tap rows)
            mfma_per_prod = {"f16x3": 3, "f16f8": 1.5}.get(precision, 1)
            if precision == "f16f8":
                kname = "conv16_kernel<*> + conv16_multi_kernel<*> + conv16_ups_kernel<*> (v_mfma_f32_32x32x16_f16 + v_mfma_scale_f32_32x32x64_f8f6f4)"
        out["roofline"] = {"bound": "mfma", "kernel": kname + ", all %d launches of a step" % (conv["launches"] // args.steps),
                           "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                           "mfma_issue_frac": round(mfma_per_prod * ach / peak, 4),
                           "region": "%d single-stream steps run right after the timed region (%.3f ms/step alone); the timed "
                                     "region keeps %d pair(s) in flight, where per-kernel spans overlap" % (args.steps, solo_ms, args.streams),
                           "note": "achieved = ALGORITHMIC conv FLOP / event-timed kernel time; mfma_issue_frac = issued "
                                   "MFMA FLOP / peak (%s fp16-MFMA units per algorithmic product)" % mfma_per_prod,
                           "traffic": traffic, "traffic_note": "HBM bytes per step of the conv launches, FETCH_SIZE x2 "
                           "(gfx950 correction) + WRITE_SIZE, separate rocprofv3 --pmc passes (tools/pmc_traffic.sh): profiles/%s" % pmc_file,
                           "flop_per_step": flops_step, "ms_per_step_in_kernel": round(conv_ms_step, 3)}
        wk = summ["warp"]
        wms = wk["ms"] / args.steps
        wach = wk["bytes"] / args.steps / (wms * 1e-3) / 1e9
        wtraffic = None
        if pmc and os.path.exists(pmc):   # same PMC passes as the conv figure: the two gather kernels' FETCH_SIZE x2 + WRITE_SIZE
            pj = json.load(open(pmc))
            wt = sum(v.get("hbm_bytes_per_step", 0.0) for k, v in pj.items() if k.startswith("flowinterp_inputs") or k.startswith("synthesize"))
            wtraffic = wt or None
        out["roofline_warp"] = {"bound": "hbm", "kernel": "flowinterp_inputs_kernel + synthesize_kernel",
                                "achieved": round(wach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": round(wach / PEAK_HBM_GBS, 4), "traffic": wtraffic,
                                "traffic_note": "HBM bytes per step of the two launches from the rocprofv3 --pmc passes (profiles/%s); above "
                                                "bytes_per_step (algorithmic) by the bilinear taps that miss L2 and the 16-byte HL8 records "
                                                "written for 10 of 16 channels" % (pmc_file or "-"),
                                "bytes_per_step": wk["bytes"] / args.steps, "ms_per_step_in_kernel": round(wms, 3)}
        if args.detail:
            det = {fam: {n: {"ms_per_step": v[0] / args.steps, "tflops": (v[1] / v[0] / 1e9 if v[0] > 0 else 0.0)}
                         for n, v in d["by_name"].items()} for fam, d in summ.items()}
            with open(args.detail, "w") as f:
                json.dump(det, f, indent=1)
        up = summ.get("upsample_cat", {"ms": 0.0})
        out["time_split_ms_per_step"] = {"conv": round(conv_ms_step, 3), "warp_blend": round(wms, 3),
                                         "upsample_cat": round(up["ms"] / args.steps, 3),
                                         "wall_single_stream": round(solo_ms, 3),
                                         "wall_timed_region": round(1e3 * elapsed / args.steps, 3)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ssm_oracle as O
        ts = [0.5]
        pair = torch.cat([x[:, 0], x[:, 1]], 1)
        cores = torch.get_num_threads()
        with torch.no_grad():
            t0 = time.perf_counter()
            want = O.interpolate_pair(sd1, sd2, pair, ts, hoist=False)     # reference-style loop
            cpu_s = time.perf_counter() - t0
            got = model.interpolate(x.to(dev), ts).cpu()
        err = max(float((got[i:i + 1] - want[i]).abs().max()) for i in range(len(ts)))
        out["cpu_baseline"] = {"value": round(len(ts) / cpu_s, 4), "unit": "frames/s", "cores": cores, "kind": "port",
                               "sample": "1 pair 736x1280 x %d intermediates (t=%s), torch CPU fp32 oracle, stage 1 "
                                         "recomputed per t like the reference loop; %.1f s" % (len(ts), ts, cpu_s)}
        out["parity"] = {"max_abs_vs_oracle": err, "tolerance": 1e-3, "frames": len(ts), "size": "736x1280"}

    if rank == 0:
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
