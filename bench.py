#!/usr/bin/env python3
"""Headline benchmark: interpolated 1280x720 frames/sec (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W          (N > 1: this process starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload = BASELINE configs[1] (superslomo_original.ini inference): synthetic Adobe240-shaped 1280x720 frame pairs
(zero-padded to 736x1280 in normalised space, resident in HBM) -> 7 intermediate frames t = 1/8..7/8 each; stage 1
runs once per pair, the 7 t values are batched through stage 2.  One "step" = `--pairs-per-step` (8) distinct pairs
= 56 frames.  N > 1: every rank processes its own pairs (weak scaling, no data-path collective); value = all ranks'
frames / max-over-ranks time.

The headline (`value`, `dtype`, `roofline`) is fp32 arithmetic throughout - every product, sum and transform in fp32 on
v_mfma_f32_32x32x2_f32 / fp32 VALU (north_star: "within 1e-3 ... in fp32"; SURVEY 8d roof = fp32 MFMA 157.3 TFLOP/s) - in mode
`f32w`: the 3x3 convolutions (71 % of the FLOPs) are evaluated as Winograd F(2x2,3x3), which needs 2.25x fewer multiplies than
the direct form for the same result (what cuDNN - `cudnn.benchmark = True`, scripts/main.py:296 of the reference - and MIOpen
choose for 3x3 fp32 layers too); the 7x7 / 5x5 / final convolutions run in the direct form.  In both fp32 modes the partial sums
of stage 2's conv1a / conv7a over their t-independent input channels (the frames; the stage-1 half of the cross-skip concat) are
computed once per pair instead of once per t - the second step of the stage-1 hoisting SURVEY Appendix B proposes ($SSM_HOIST=0
turns it off).  Mode `f32` (EVERY convolution in the direct form, an fmaf chain per output) is reported beside it under `modes`, and so are the split-fp16 modes (narrower than
fp32: options, not the configuration the metric is quoted on), each with its own frame rate, roofline and parity.

Objects on the JSON line:
  roofline      dominant kernel family = the 48 convolution launches of a pair.  achieved = algorithmic conv FLOP of
                the timed region (5.855 TFLOP per pair at 736x1280, 7 t, stage 1 hoisted) / its wall time
                (so everything that is not a convolution counts against it); detail.* = the same FLOP / the summed
                HIP-event durations of those launches in a single-stream region run right after the timed one (with
                several pairs in flight the per-kernel spans overlap), plus the per-kernel table with --detail.
                traffic = HBM bytes of those launches per pair from rocprofv3 --pmc passes (profiles/).
                `achieved` / `frac` count ALGORITHMIC (direct-form) FLOP as SURVEY 8d defines them, so in mode f32w they can
                exceed the matrix peak; `issued` = the multiply-adds the matrix cores actually execute (Winograd layers:
                direct FLOP / 2.25; hoisted channels once per pair) / the same time - the utilisation of the fp32 MFMA pipe, <= 1.
  roofline_warp the HBM-bound gather kernels (compute_inputs + synthesis): algorithmic bytes (104 + 72 B/px per t)
                / their event-timed duration; peak 8 TB/s.
  modes         f32 (direct form everywhere) / f16x3 / f16f8 (split fp16 (+fp8), narrower than fp32): frames/s, roofline, parity on the same pairs.
  cpu_baseline  the CPU oracle (torch CPU fp32 ops, pinned to the reference by golden fixtures) on this host: warm,
                best of a thread sweep; C1 (256x256, 1 t) and C2 (1 pair x 7 t) in the hoisted loop and in the
                reference-style loop that recomputes stage 1 per t.  A reported baseline, not the target.
  parity        max|HIP - oracle| over all 7 frames of that pair at 736x1280, per mode (bar: 1e-3).
  io            measured PCIe legs of the uint8 frame path (never part of `value`).
"""
import argparse
import json
import os
import socket
import subprocess
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
HEADLINE_PRECISION = "f32w"
DTYPE_NOTE = {"f32": "f32",
              "f32w": "f32",
              "f16": "f16 (f32 accumulate) - reduced precision",
              "f16x3": "split f16: 3x f16 MFMA on hi/lo-split f32 operands (~22-bit operands, f32 accumulate) - narrower than f32",
              "f16f8": "split f16+e4m3: 1x f16 MFMA + 2x block-scaled e4m3 MFMA for the compensation products (~15-bit products, "
                       "activations stored as f16 hi + e4m3 lo, f32 accumulate) - narrower than f32"}
# rocprofv3 --pmc traffic summaries (tools/pmc_traffic.sh) per mode: (file under profiles/, conv kernel family keys)
ALGO_NOTE = {"f32w": "fp32 throughout; 3x3 convolutions as Winograd F(4x4,3x3) (csrc/ssm_wino4.hip; conv11a in the sub-pixel form) or F(2x2,3x3) (csrc/ssm_wino.hip), 7x7 / 5x5 "
                     "convolutions as 2x2 blocks of F(4x4,4x4) (csrc/ssm_wino7.hip) / F(4x4,5x5) (csrc/ssm_wino5.hip), final convolutions in the direct form "
                     "(csrc/ssm_elem.hip); the t-independent input channels of stage 2's conv1a / conv7a convolved once per pair",
             "f32": "fp32 throughout; every convolution in the direct form (an fmaf chain per output; the t-independent input channels of "
                    "stage 2's conv1a / conv7a summed once per pair and added - SSM_HOIST=0 keeps one chain)"}
PMC_FILES = {"f32w": ("r20_pmc_traffic_f32w_summary.json", ("wino4_kernel", "wino7s_kernel", "wino7_kernel", "wino5s_kernel", "wino5_kernel", "wino1d_kernel", "wino2_kernel", "wino_kernel", "conv_mfma_kernel", "final_conv_valu_kernel", "final_conv_kernel"))}
# (the side modes report no `traffic`: their counter passes date from rounds 1-2 - profiles/r2_pmc_traffic_f32_summary.json, r1k / r1q - and the
# kernels have changed since; only the headline mode's summary is regenerated every round by tools/final_profiles.sh)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set as torch.distributed.run would) and relay rank 0's JSON line.  This parent never touches the GPU: it is
    called before any HIP / torch.cuda call, and it does not exec."""
    if under_profiler():
        sys.stderr.write("bench.py --gpus %d: a profiler is attached to this process; start the ranks with torch.distributed.run instead "
                         "(one profiler per rank), or profile `--gpus 1`\n" % n)
        sys.exit(2)
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies before the rendezvous would leave the others waiting for it until the store times out: poll, and end the launch
    # as soon as any rank has failed
    # rank 0's pipe is drained by a thread WHILE the ranks are polled: a line longer than the 64 KB pipe buffer (today's is ~20 KB) would
    # otherwise block rank 0 in write() forever while this loop waits for it to exit
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while True:
        rcs = [p.poll() for p in procs]
        if any(rc not in (None, 0) for rc in rcs) or all(rc == 0 for rc in rcs):
            break
        time.sleep(0.05)
    if any(rc not in (None, 0) for rc in rcs):
        for p in procs:
            if p.poll() is None:
                p.kill()
    rcs = [p.wait() for p in procs]
    reader.join(timeout=30)
    out = b"".join(chunks)
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: rank(s) failed: %s\n" % bad)
        sys.exit(1)
    sys.exit(0)


CONFIG_RUNS = (
    # key, BASELINE.json configs[] index, arguments of the child bench.py, needs a one-rank process group
    ("config3_train", 2, ["--mode", "train", "--precision", "f32w", "--steps", "20", "--warmup", "3", "--force-allreduce"], True),
    ("config4_recurrent", 3, ["--mode", "recurrent", "--precision", "f32w", "--steps", "10", "--warmup", "2"], False),
    ("config5_4k", 4, ["--size", "4k", "--precision", "f32w", "--streams", "1", "--pairs-per-batch", "1", "--pairs-per-step", "1", "--steps", "4",
                       "--warmup", "1", "--no-kernel-timers", "--no-clock-probes", "--parity-4k", "--modes-4k", "f16"], False),
)


def under_profiler():
    """rocprofv3 (and the older rocprof tools) preload a tool library that initialises HSA / HIP before Python starts: a process in that
    state must not start other programs (the GPU boxes refuse such an exec)."""
    env = os.environ
    return (any(k.startswith(("ROCPROF", "ROCPROFILER", "ROCTRACER")) for k in env) or bool(env.get("HSA_TOOLS_LIB"))
            or "rocprof" in env.get("LD_PRELOAD", "").lower())


def collect_configs(timeout_s=300):
    """BASELINE configs[2..4] beside the headline: short runs of `--mode train` (with the RCCL gradient buckets
    forced at world 1), `--mode recurrent` and `--size 4k`, each a fresh child process of THIS script, one after the other, started
    before this process has touched the GPU (no exec from a process that holds the device).  Returns {key: summary}."""
    out = {}
    if under_profiler():
        return {"skipped": "a profiler is attached to this process (its preloaded tool library initialises the GPU before main): no child processes are started"}
    for key, idx, argv, group in CONFIG_RUNS:
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if group:
            env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
        t0 = time.perf_counter()
        try:
            pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-configs"] + argv, env=env, stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE, timeout=timeout_s)
            line = [ln for ln in pr.stdout.decode().splitlines() if ln.startswith("{")]
            if pr.returncode != 0 or not line:
                out[key] = {"error": "rc %d: %s" % (pr.returncode, pr.stderr.decode()[-400:])}
                continue
            d = json.loads(line[-1])
        except subprocess.TimeoutExpired:
            out[key] = {"error": "timed out after %d s" % timeout_s}
            continue
        except OSError as e:          # the child could not be started (e.g. exec refused on this host)
            out[key] = {"error": "could not start the child: %s" % e}
            continue
        rf = d.get("roofline", {})
        rec = {"baseline_config": idx, "metric": d["metric"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "warmup": d["warmup"],
               "ms_per_step": d["ms_per_step"], "dtype": d["dtype"], "workload": d["config"].get("workload"),
               "roofline": {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac") if k in rf},
               "parity": d.get("parity"), "argv": " ".join(argv), "child_wall_s": round(time.perf_counter() - t0, 1)}
        for k in ("allreduce", "host_enqueue_ms_per_step", "host"):
            if k in d:
                rec[k] = d[k]
        if d.get("modes"):          # config 5: the fp16 MFMA path named by BASELINE.json beside the f32w answer
            rec["modes"] = {m: {"value": v.get("value"), "unit": v.get("unit"), "dtype": v.get("dtype"), "parity": v.get("parity"),
                                "roofline": {k: v.get("roofline", {}).get(k) for k in ("bound", "achieved", "peak", "unit", "frac")}}
                            for m, v in d["modes"].items()}
        out[key] = rec
    return out


def pin_rank_cpus(local_rank, local_world):
    """One process per GPU, N of them on one host: every rank keeps to its own contiguous 1/N of the CPUs this job may use, so that the
    launch threads of the ranks (each needs host_enqueue_ms_per_step of a core per step) and their HIP helper threads do not migrate over
    one another.  $SSM_RANK_AFFINITY=0 leaves the affinity alone.  Returns the CPU list, or None when nothing was changed."""
    if local_world <= 1 or os.environ.get("SSM_RANK_AFFINITY", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = sorted(os.sched_getaffinity(0))
    n = len(cpus) // local_world
    if n < 1:
        return None
    mine = cpus[local_rank * n:(local_rank + 1) * n]
    os.sched_setaffinity(0, mine)
    return mine


def setup_ranks(args):
    """(rank, local_rank, world, device).  --stub: CPU + gloo (launcher / timing-protocol test, no GPU work)."""
    from ssm_amd import dist as sdist
    rank, local_rank, world = sdist.env_world()
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    args.rank_cpus = pin_rank_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if args.stub:
        sdist.init("gloo")
        return rank, local_rank, world, torch.device("cpu")
    assert torch.cuda.is_available(), "bench.py measures the HIP path; no GPU visible"
    # one process per GPU: the device is the LOCAL rank (the driver starts N ranks on ONE node; a global rank would index past the node)
    ndev = torch.cuda.device_count()
    assert 0 <= local_rank < ndev, "LOCAL_RANK %d but %d visible GPU(s): one rank per GPU of the node" % (local_rank, ndev)
    torch.cuda.set_device(local_rank)
    assert torch.cuda.current_device() == local_rank
    free_b, total_b = torch.cuda.mem_get_info(local_rank)
    need = hbm_needed_bytes(args)
    assert free_b >= need, ("rank %d: %.0f GB of HBM free on cuda:%d, this configuration's plans need ~%.0f GB (activations are planned, "
                            "nothing aliased: DESIGN 2)" % (rank, free_b / 1e9, local_rank, need / 1e9))
    args.hbm = {"device_total_gb": round(total_b / 1e9, 1), "free_at_start_gb": round(free_b / 1e9, 1), "planned_need_gb": round(need / 1e9, 1)}
    sdist.init("nccl")
    return rank, local_rank, world, torch.device("cuda", local_rank)


def hbm_needed_bytes(args):
    """Conservative per-RANK HBM need of a configuration (one process per GPU, so per GPU): a UNetPlan keeps ~45 activations of a
    U-Net resident, 2.2 GB per batch entry at 736x1280, scaled by the pixels; a pass holds pairs_per_batch stage-1 entries +
    7 x that many stage-2 entries, `streams` passes in flight.  Training (2 x 352x352, forward activations + gradient planes +
    materialised upsample tensors + VGG16 + Adam state): 14 GB measured.  The parity / side-mode legs of a default line reuse the
    same memory after the timed plans are dropped."""
    if args.mode == "train":
        return int(20e9)
    px = (2176 * 3840) if args.size == "4k" else (736 * 1280)
    per_entry = 2.2e9 * px / (736 * 1280) / 2.0          # (2.2 GB covers a stage-1 + a stage-2 entry)
    if args.mode == "recurrent":
        return int(3 * 2 * per_entry * 2 + 8e9)
    entries = args.streams * args.pairs_per_batch * (1 + N_T)
    return int(entries * per_entry + 12e9)


def conv_flops_per_pair(h, w, n_t, issued_for=None, batch=2):
    """Algorithmic (direct-form) conv FLOP of a pair as SURVEY 8d counts them (stage 1 once, stage 2 per t).  issued_for = "f32" /
    "f32w": the multiply-adds the matrix cores actually execute in that mode: the t-independent input channels of stage 2's conv1a
    (6 of 16) and conv7a (512 of 1024) are convolved once per pair (ssm_amd.engine.UNetPlan.hoist); "f32w": 3x3 layers with >= 32
    output channels run as Winograd F(2x2,3x3), 16 instead of 36 multiply-adds per 2x2 outputs; 7x7 / 5x5 layers in the 1-D
    Winograd forms along x where the plan uses them (issued_factor)."""
    from ssm_amd.weights import unet_layers
    scale = {"conv1": 1, "conv2": 2, "conv3": 4, "conv4": 8, "conv5": 16, "conv6": 32, "conv7": 16, "conv8": 8,
             "conv9": 4, "fuse_": 1, "final": 1}

    def stage(st, reps):
        tot = 0.0
        for name, cin, cout, k in unet_layers(st, True):
            s = 2 if name.startswith("conv10") else 1 if name.startswith("conv11") else \
                [v for p, v in scale.items() if name.startswith(p)][0]
            fl = 2.0 * (h // s) * (w // s) * cin * cout * k * k * reps
            if issued_for in ("f32", "f32w") and st == 2 and reps > 1:
                if name == "conv1a":
                    fl *= (10.0 * reps + 6.0) / (16.0 * reps)
                elif name == "conv7a":
                    fl *= (reps + 1.0) / (2.0 * reps)
            if issued_for:
                fl *= issued_factor(name, issued_for, batch * reps, h, w)
            tot += fl
        return tot
    return stage(1, 1) + stage(2, n_t)


def layer_kernel_size(lname):
    """7 / 5 / 3 for a layer name of SURVEY Appendix A ("conv1a", "conv10b", "fuse_conv", "conv6(CLSTM)", "conv1a(pair)")."""
    if lname.startswith("conv1") and not lname.startswith(("conv10", "conv11")):
        return 7
    return 5 if lname.startswith("conv2") else 3


def issued_factor(lname, precision, B=14, H=736, W=1280):
    """Multiply-adds the matrix cores issue per direct-form multiply-add of this layer in this precision mode: the plan's algorithm
    for the layer (ssm_amd.engine.choose_algo: F(4x4,3x3) x 1/4, F(2x2,3x3) x 16/36, 7x7 as blocked F(4x4,4x4) x 1/4, 5x5 as F(4x4,5x5) x 64/400 (F(4,5) along x, the fallback: x 8/20), direct x 1).
    B, H, W: batch and full-resolution size of the plan the layer belongs to."""
    if precision != "f32w" or lname.startswith("final"):
        return 1.0
    from ssm_amd import engine
    from ssm_amd.weights import unet_layers
    base = lname.split("(")[0]
    if base == "conv6":            # recurrent bottleneck: gate convolutions in the F(2x2,3x3) form
        return 16.0 / 36.0
    ci, co, k = {n: (a, b, c) for n, a, b, c in unet_layers(2, True)}[base]
    s = engine.layer_scale(base)
    algo = engine.choose_algo(base, ci, co, k, B, H // s, W // s, base in engine.UNetPlan.UPS, True, True)
    return engine.ISSUED_FACTOR[algo](k)


def family_rooflines(by_name, n_pairs, precision, peak, batch=14, H=736, W=1280):
    """Per kernel family of the fp32 modes (3x3 / 7x7 / 5x5 / final): in-kernel time per pair, FLOP issued on the matrix cores,
    fraction of the fp32-MFMA peak - from the HIP-event brackets of the single-stream region."""
    fams = {}
    for n, (ms, fl, cnt) in by_name.items():
        lname = n.split(".", 1)[1]
        key = "final_conv (vector ALU)" if lname.startswith("final") else "%dx%d layers" % ((layer_kernel_size(lname),) * 2)
        d = fams.setdefault(key, {"ms": 0.0, "alg": 0.0, "iss": 0.0})
        d["ms"] += ms
        d["alg"] += fl
        d["iss"] += fl * issued_factor(lname, precision, batch, H, W)
    return {k: {"ms_per_pair_in_kernel": round(d["ms"] / n_pairs, 3), "flop_issued_per_pair": d["iss"] / n_pairs,
                "achieved_in_kernel": round(d["iss"] / d["ms"] / 1e9, 2), "frac_in_kernel": round(d["iss"] / d["ms"] / 1e9 / peak, 4),
                "algorithmic_achieved_in_kernel": round(d["alg"] / d["ms"] / 1e9, 2)} for k, d in fams.items()}


def warp_kernel_rate(dev, H, W, B=7, reps=20):
    """layers.warp as its own launch (ssm_warp_bilinear_fwd -> warp_kernel; the pipeline uses the fused gather kernels): SURVEY 8d's
    32 B/px (3 channels gathered + 2 flow channels read + 3 channels written, fp32) over its HIP-event time, smooth flows of a few pixels."""
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(5)
    img = torch.randn(B, 3, H, W, generator=g).to(dev)
    # a smooth flow field like the network's (a few pixels of translation + a slow spatial variation): neighbouring pixels sample
    # neighbouring taps.  (Independent random flows per pixel scatter every wave-level gather over ~10 rows: 1.9-2.1 TB/s.)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    flo = torch.stack([3.0 + 2.0 * torch.sin(xx / 97.0) * torch.cos(yy / 131.0), 2.0 + 1.5 * torch.cos(xx / 113.0 + yy / 89.0)], 0)
    flo = flo.unsqueeze(0).expand(B, -1, -1, -1).contiguous().to(dev)
    out = torch.empty_like(img)
    lib = hb.load()

    def call():
        hb.check(lib.ssm_warp_bilinear_fwd(hb.view_of(img), hb.view_of(flo), hb.view_of(out), B, 3, H, W, hb.stream_ptr()))

    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / reps
    gbs = 32.0 * B * H * W / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "warp_kernel", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(gbs / PEAK_HBM_GBS, 4), "ms_per_launch": round(ms, 4), "bytes_per_launch": 32.0 * B * H * W,
            "note": "batch %d of %dx%d, C-ABI call with a preallocated output, back-to-back launches" % (B, H, W)}


def family_clocks(eng, dev, peak, fams):
    """Shader clock under each conv family of the headline mode (one resident wave on a side stream comparing s_memtime with the
    100 MHz s_memrealtime, tools/clock_probe.hip, while stage 2's layers of that family run back to back on resident inputs):
    the power management holds different clocks under the Winograd and the direct kernels, so each family's roof at the clock it
    was delivered is peak x ghz / 2.4."""
    out = {}
    for k in (3, 7, 5):
        key = "%dx%d layers" % (k, k)
        if key not in fams:
            continue
        eng.s2.run_family(k)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        eng.s2.run_family(k)
        torch.cuda.synchronize(dev)
        reps = max(2, int(0.7 / max(time.perf_counter() - t0, 1e-4)))
        probe = start_clock_probe(dev, 0.4)
        if probe is None:
            return None
        for _ in range(reps):
            eng.s2.run_family(k)
        ghz = read_clock_probe(probe)
        torch.cuda.synchronize(dev)
        if ghz:
            out[key] = {"ghz": round(ghz, 3), "peak_at_clock": round(peak * ghz / 2.4, 1),
                        "frac_in_kernel_at_clock": round(fams[key]["achieved_in_kernel"] / (peak * ghz / 2.4), 4)}
    out["note"] = "stage-2 layers of one family back to back for 0.7 s, clock averaged over 0.4 s of it"
    return out


TRAIN_DTYPE_NOTE = {
    "f16f8": "f32 parameters/activations/gradients; products via 1x f16 + 2x block-scaled e4m3 MFMA (forward, data gradients) and 3x bf16 "
             "MFMA on hi/lo-split operands (weight gradients), f32 accumulate - narrower than f32",
    "f32": "f32 (fp32 MFMA)", "f32w": "f32 (fp32 MFMA; forward and data-gradient convolutions in the Winograd forms: 3x3 F(2x2,3x3), 7x7 / 5x5 blocked; weight gradients direct)"}


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

    rank, local_rank, world, dev = setup_ranks(args)
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
    train_mode = args.precision or os.environ.get("SSM_TRAIN_PRECISION", "f32")
    assert train_mode in ("f32", "f32w", "f16f8"), "--mode train: --precision f32 (default, direct form), f32w (3x3 layers as Winograd) or f16f8"
    model.train_precision = train_mode
    if args.force_allreduce:            # one-rank process group: still hand every gradient bucket to RCCL (ssm_amd.dist.GradientAllReduce.attach)
        os.environ["SSM_FORCE_ALLREDUCE"] = "1"
    trainer = Trainer(model, cfg)
    B, S = 2, 352
    clips = torch.cat([synthetic_frames(3, S, S, seed=100 + 2 * rank + i) for i in range(B)], 0).to(dev)   # [B,3,3,S,S]
    xin, tgt = clips[:, [0, 2]].contiguous(), clips[:, 1:2].contiguous()
    t = torch.tensor([(3 + i + rank) % 7 + 1 for i in range(B)], dtype=torch.float32, device=dev).view(B, 1, 1, 1, 1) / 8.0
    def step():
        trainer.train_step(xin, tgt, t)

    def sync():
        torch.cuda.synchronize(dev)

    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the step's forward (losses [B,4] and predicted frame) on the initial weights beside the CPU oracle's training_loss
        # (losses.py:196-249 restated); the gradients are held to the oracle's autograd and to the reference's own gradient fixture in
        # tests/test_hip_backward.py (minutes of CPU autograd: not repeated here)
        from oracle import ssm_oracle as O
        lr_, lp_, lw_ = model.loss.loss_weights
        vsd = None if args.no_perceptual else synthetic_vgg_state_dict()
        with torch.no_grad():
            pred_h, loss_h = model(xin, t, target_images=tgt, inference_mode=False)
            loss_o, pred_o = O.training_loss(synthetic_state_dict(1), synthetic_state_dict(2), torch.cat([xin[:, 0], xin[:, 1]], 1).cpu(),
                                             t.view(B, 1, 1, 1).cpu(), tgt[:, 0].cpu(), lr_, lw_, vsd, lp_ if vsd is not None else 0.0)
        parity = {"frame_max_abs_vs_oracle": float((pred_h.cpu() - pred_o).abs().max()), "tolerance": 1e-3,
                  "loss_rel_err_vs_oracle": float(((loss_h.cpu() - loss_o).abs() / loss_o.abs().clamp_min(1e-12))[:, 0].max()),
                  "what": "forward of one step (predicted frame, total loss per sample) on the initial weights vs the CPU oracle; gradients: "
                          "tests/test_hip_backward.py (oracle autograd + the reference's gradient fixture)"}
        model._drop_plans()
    for _ in range(args.warmup):
        step()
    sync()
    trainer.allreduce.exposed_seconds()          # (drop the warm-up's brackets)
    elapsed = sdist.timed_steps(step, args.steps, 0, sync)
    ar_brackets = len(trainer.allreduce._exposed_events)          # GPU + RCCL: one event bracket per timed step (a dead measurement shows as 0)
    ar_ms = 1e3 * trainer.allreduce.exposed_seconds() / args.steps
    in_region_host_s = sdist.timed_steps.last_enqueue_s
    # host time to ISSUE one step, measured like the inference line's: the queues are empty when the step starts, so nothing blocks the
    # host.  (Inside the timed region a host that is far ahead of the GPU fills the HIP queue and then blocks in the launch call: that
    # figure - kept below as in_timed_region - approaches the GPU's own step time however little the host needs.)
    enq = 0.0
    for _ in range(5):
        sync()
        t0 = time.perf_counter()
        step()
        enq += time.perf_counter() - t0
    sync()
    host_ms = 1e3 * enq / 5
    prog = getattr(trainer, "_prog", None)
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
           "dtype": TRAIN_DTYPE_NOTE[train_mode], "data": "synthetic",
           "config": {"workload": "superslomo_original.ini training, FREEZE=FALSE, %d samples/GPU of 352x352, %d GPU(s); "
                                  "losses: L1 reconstruction + 4 L1 warp terms + %s" % (B, world, "VGG16 conv4_3 perceptual term OFF" if args.no_perceptual else
                                                                                  "VGG16 conv4_3 perceptual term (synthetic VGG weights)"),
                      "global_batch": B * world},
           "allreduce": {"bytes": trainer.allreduce.bytes, "ms_per_step": round(ar_ms, 3), "buckets_per_step": trainer.allreduce.last_buckets,
                         "brackets": ar_brackets,
                         "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
                         "note": "ms_per_step = time the compute stream waited for the bucketed exchange (event brackets; the exchange overlaps the backward and never blocks the host)"},
           "host_enqueue_ms_per_step": round(host_ms, 3),
           "host": {"enqueue_ms_per_step": round(host_ms, 3), "enqueue_ms_per_step_in_timed_region": round(1e3 * in_region_host_s / args.steps, 3),
                    "launch_program": ({"nodes": prog["program"].n_nodes, "items": len(prog["program"].items),
                                        "host_items": sum(1 for it in prog["program"].items if it[0] == "py"), "streams": len(prog["program"].streams)}
                                       if prog is not None else None),
                    "note": "enqueue_ms_per_step: host time to issue one step with empty queues (5 steps, device synchronised before each); "
                            "in_timed_region: host time inside the timed loop / steps - a host far ahead of the GPU blocks in the launch call once "
                            "the HIP queue is full, so that figure tends to the GPU's step time"}}
    if parity is not None:
        out["parity"] = parity
    if rank == 0:
        summ = timer.summary()
        out["time_split_ms_per_step"] = {fam: round(d["ms"] / 3, 3) for fam, d in summ.items()}
        out["tflops"] = {fam: round(d["flops"] / d["ms"] / 1e9, 1) for fam, d in summ.items() if d["flops"] > 0}
        # roofline of the step (forward 93.9 GMAC per sample, backward = data + weight gradients of the same
        # convolutions, + the VGG16 conv4_3 passes): every MFMA-bound family against the MFMA peak.  `achieved` = the multiply-adds
        # the matrix cores ISSUE in one step (Winograd launches count 16/36 or 36/144 of their direct-form FLOP: engine.ISSUED_FACTOR,
        # recorded per launch by the brackets) / the step's wall time in the timed region - always <= peak; the direct-form figure the
        # step corresponds to is under `algorithmic`.  families.* = the same / the summed HIP-event durations of that family (in-kernel).
        peak = PEAK_F16_MFMA_TFLOPS if train_mode == "f16f8" else PEAK_F32_MFMA_TFLOPS
        fl_step = sum(d["flops"] for d in summ.values()) / 3.0
        is_step = sum(d["issued"] for d in summ.values()) / 3.0
        wall = 1e-3 * out["ms_per_step"]
        ach = is_step / wall / 1e12
        out["roofline"] = {
            "bound": "mfma",
            "kernel": ("conv16 / wgrad_bf16x3 kernels" if train_mode == "f16f8" else
                       "conv_mfma_kernel / wino2_kernel (forward, data gradients), wgrad kernels (weight gradients), VGG16 conv kernels: "
                       "all MFMA families of a step"),
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "flop_per_step": is_step,
            "algorithmic": {"achieved": round(fl_step / wall / 1e12, 2), "flop_per_step": fl_step,
                            "note": "direct-form FLOP of the same convolutions / the same wall time (can exceed the peak where Winograd forms run)"},
            "region": "the timed region: FLOP issued on the matrix cores in %d steps / its wall time (launch gaps, elementwise kernels, Adam "
                      "and the exchange count against it)" % args.steps,
            "families": {fam: {"flop_per_step": d["issued"] / 3.0, "direct_form_flop_per_step": d["flops"] / 3.0,
                               "ms_per_step_in_kernel": round(d["ms"] / 3, 3),
                               "achieved_in_kernel": round(d["issued"] / d["ms"] / 1e9, 2),
                               "frac_in_kernel": round(d["issued"] / d["ms"] / 1e9 / peak, 4),
                               "algorithmic_in_kernel": round(d["flops"] / d["ms"] / 1e9, 2)}
                         for fam, d in summ.items() if d["flops"] > 0},
            "traffic": None,
            "note": ("3x3 forward / data-gradient / VGG layers run as Winograd F(2x2,3x3) / F(4x4,3x3): 16/36 / 1/4 of their direct-form FLOP are issued; the "
                     "7x7 / 5x5 layers (forward and data gradients) in the blocked 7x7 / F(4x4,5x5) forms (1/4, 64/400); weight gradients and final "
                     "layers run in the direct form" if train_mode == "f32w" else
                     "every product issued in the direct form: FLOP counted = FLOP issued")}
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

    rank, local_rank, world, dev = setup_ranks(args)
    cfg = load_config("superslomo_recurrent.ini", synthetic_weight_overrides())
    kind = cfg.get("STAGE1", "BOTTLENECK")
    model = FullModel(cfg)
    p1, p2 = synthetic_state_dict(1, bottleneck=kind), synthetic_state_dict(2, bottleneck=kind)
    model.stage1_model.load_state_dict(p1)
    model.stage2_model.load_state_dict(p2)
    model.precision = args.precision or HEADLINE_PRECISION
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
        # roofline (SURVEY 8d, C4 = C2's layers at batch 3 windows (x 7 t in stage 2) + the ConvBLSTM gate convolutions): direct-form
        # conv FLOP of a clip from the per-launch brackets; issued = the multiply-saving layers' share scaled (mode f32w)
        peak = PEAK_F32_MFMA_TFLOPS if model.precision in ("f32", "f32w") else PEAK_F16_MFMA_TFLOPS
        fl_clip = conv["flops"] / 3.0
        issued = sum(v[1] / 3.0 * issued_factor(n.split(".", 1)[1], model.precision, N_T, 736, 1280) for n, v in conv["by_name"].items())
        ms_clip = out["ms_per_step"]
        out["roofline"] = {
            "bound": "mfma", "kernel": "wino2_kernel / wino_kernel (3x3 layers and gate convolutions), conv kernels of the 7x7 / 5x5 layers, final_conv_kernel",
            "achieved": round(issued / (1e-3 * ms_clip) / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(issued / (1e-3 * ms_clip) / 1e12 / peak, 4), "flop_per_clip": issued,
            "region": "the timed region: multiply-adds the matrix cores issue for %d clips / its wall time" % args.steps,
            "algorithmic": {"flop_per_clip": fl_clip, "achieved": round(fl_clip / (1e-3 * ms_clip) / 1e12, 2),
                            "x_peak": round(fl_clip / (1e-3 * ms_clip) / 1e12 / peak, 4),
                            "note": "direct-form FLOP (SURVEY 8d); exceeds the issued figure where layers run in a Winograd form"},
            "in_kernel": {"ms_per_clip": round(conv["ms"] / 3, 3), "issued_achieved": round(issued / (1e-3 * conv["ms"] / 3) / 1e12, 2),
                          "issued_frac": round(issued / (1e-3 * conv["ms"] / 3) / 1e12 / peak, 4)},
            "traffic": None}
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


def stub_bench(args):
    """--stub: the launcher, rendezvous, barrier-bracketed timing and max-over-ranks reduction of the real argument path
    on CPU (gloo) with a sleep as the per-pair work.  Used by tests/test_dist_gloo.py only; the line says so."""
    from ssm_amd import dist as sdist
    rank, local_rank, world, dev = setup_ranks(args)
    mine = sdist.assign_pairs(args.pairs_per_step * world, world, rank)

    if args.stub_fail_rank is not None and rank == args.stub_fail_rank:
        sys.exit(3)          # (launcher test: a rank that dies before the rendezvous completes must fail the whole launch)

    def step():
        for _ in mine:
            time.sleep(0.001 * (rank + 1))

    elapsed = sdist.timed_steps(step, args.steps, args.warmup, lambda: None)
    if rank == 0:
        print(json.dumps({"metric": "interpolated 1280x720 frames/sec", "value": round(N_T * len(mine) * world * args.steps / elapsed, 3),
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "stub (no GPU work: launcher / timing-protocol test only)",
                          "config": {"workload": "stub", "pairs_per_step": len(mine)},
                          "host": {"usable_cpus": usable_cpus(), "cpus_of_this_rank": args.rank_cpus}}))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def stub_train_bench(args):
    """--mode train --stub: the TRAINING launcher path at world N on CPU / gloo (the real one needs N GPUs: only RCCL at world 1 runs
    in `pytest -m gpu`).  Everything of the exchange is real - the two U-Nets' flat gradient buffers in their true layout
    (ssm_amd.backward.grad_layout: 38,848,553 floats = 155.4 MB, 4 buckets each, handed over tail first as the backward completes
    them), GradientAllReduce.reduce / __call__, the 1/world pre-scale, barrier-bracketed timing, max over ranks - only the gradients are
    synthetic: rank r contributes (r + 1) x a fixed pattern, so every rank can check the average it receives."""
    import torch.nn as nn
    from ssm_amd import dist as sdist
    from ssm_amd.backward import UNetGrad, grad_layout
    from ssm_amd.weights import unet_layers
    rank, local_rank, world, dev = setup_ranks(args)
    os.environ["SSM_FORCE_ALLREDUCE"] = "1"
    flats, buckets, params = [], [], []
    for stage in (1, 2):
        layers = {n: (ci, co, k) for n, ci, co, k in unet_layers(stage, True)}
        sizes, span, bk = grad_layout(layers, UNetGrad.N_BUCKETS)
        flat = torch.empty(sum(int(torch.Size(sh).numel()) for _, sh in sizes))
        off = 0
        for key, sh in sizes:
            n = int(torch.Size(sh).numel())
            prm = nn.Parameter(torch.empty(0))
            prm.data = torch.zeros(sh)
            prm.grad = flat[off:off + n].view(sh)
            params.append(prm)
            off += n
        flats.append(flat)
        buckets.append([(span[param_key_layer(b[0])][0], span[param_key_layer(b[-1])][1]) for b in bk])
    ar = sdist.GradientAllReduce(params)

    class _PG:          # what GradientAllReduce.attach configures on a PairGrad
        sync, sync_scale = None, 1.0
    pg = _PG()
    ar.attach(pg)
    assert pg.sync is ar and abs(pg.sync_scale - 1.0 / world) < 1e-12
    pattern = [torch.linspace(0.5, 1.5, f.numel()) for f in flats]
    checks = []

    def step():
        for u in (1, 0):                                   # the backward runs stage 2 first, then stage 1; buckets complete tail first
            flats[u].copy_(pattern[u] * float(rank + 1))
            for a, b in reversed(buckets[u]):
                view = flats[u][a:b]
                view.mul_(pg.sync_scale)
                pg.sync.reduce(view)
        ar()
        want = (world + 1) / 2.0                           # mean over ranks of (r + 1)
        checks.append(max(float((flats[u][::4097] - want * pattern[u][::4097]).abs().max()) for u in (0, 1)))

    elapsed = sdist.timed_steps(step, args.steps, args.warmup, lambda: None)
    assert max(checks) < 1e-5, "averaged gradients are off by %.3e" % max(checks)
    B = 2
    if rank == 0:
        print(json.dumps({"metric": "training samples/sec (352x352 crops, forward+backward+Adam)", "value": round(B * world * args.steps / elapsed, 3),
                          "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "stub (no GPU work: launcher / gradient-exchange test only)",
                          "config": {"workload": "stub of superslomo_original.ini training: the real gradient layout and exchange, synthetic gradients",
                                     "global_batch": B * world},
                          "allreduce": {"bytes": ar.bytes, "buckets_per_step": ar.last_buckets, "backend": torch.distributed.get_backend(),
                                        "ms_per_step": round(1e3 * ar.exposed_seconds() / max(args.steps + args.warmup, 1), 3),
                                        "max_abs_err_of_the_average": max(checks)},
                          "host": {"usable_cpus": usable_cpus(), "cpus_of_this_rank": args.rank_cpus}}))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def param_key_layer(name):
    from ssm_amd.weights import param_key
    return param_key(name, "weight")[:-len(".weight")]


def start_clock_probe(dev, seconds):
    """Diagnostic (optional: needs tools/libclockprobe.so, built by __graft_entry__.build()): one wave on a side stream spins
    for `seconds` and records shader-clock ticks against the 100 MHz reference."""
    import ctypes
    path = os.path.join(ROOT, "tools", "libclockprobe.so")
    if not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
        lib.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p]
        out = torch.zeros(2, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        lib.clock_probe_launch(out.data_ptr(), int(seconds * 1e8), ctypes.c_void_p(side.cuda_stream))
        return out, side
    except OSError:
        return None


def read_clock_probe(probe):
    out, side = probe
    side.synchronize()
    c, r = [int(v) for v in out.cpu()]
    return c / (r / 100e6) / 1e9 if r > 0 else None


def physical_cores():
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


def usable_cpus():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(sd1, sd2, pair):
    """The CPU oracle timed on this host (kind "port": torch CPU fp32 restatement pinned to the reference by fixtures).
    Thread count: a sweep over a C2-sized stage-1 pass (736x1280, ~2 s each; r4 swept on the 256x256 case, which stops scaling at 16
    threads and says nothing about 720p).  At the best count: the hoisted loop (stage 1 once, 7 t; its frames are the parity
    reference) and the reference-style loop (stage 1 recomputed per t, evaluate_interpolation_results.py:234-242) timed FOR REAL on
    a bounded sample of 3 of the pair's 7 t values.  C1 (256x256, one t) at the same count.  Returns (dict, frames of the 7 t, C1)."""
    from oracle import ssm_oracle as O
    from ssm_amd.weights import synthetic_frames
    logical, phys = usable_cpus(), physical_cores()
    cand = sorted({n for n in (8, 16, 32, 64, phys) if 1 <= n <= min(phys, logical)})     # SMT siblings only slow the oracle down
    x1 = synthetic_frames(2, 256, 256, seed=42)
    pair1 = torch.cat([x1[:, 0], x1[:, 1]], 1)
    sweep = {}
    ts = [i / 8.0 for i in range(1, N_T + 1)]
    with torch.no_grad():
        torch.set_num_threads(cand[0])
        t0 = time.perf_counter()
        O.stage1(sd1, pair)                                               # first call at these shapes (allocator, oneDNN primitives)
        s1_cold = time.perf_counter() - t0
        for n in cand:
            torch.set_num_threads(n)
            O.stage1(sd1, pair1)                                           # spin the thread pool up at this count
            t0 = time.perf_counter()
            O.stage1(sd1, pair)
            sweep[n] = time.perf_counter() - t0
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        t0 = time.perf_counter()
        want = O.interpolate_pair(sd1, sd2, pair, ts, hoist=True)
        hoisted_s = time.perf_counter() - t0
        sample_ts = [ts[0], ts[3], ts[6]]
        t0 = time.perf_counter()
        O.interpolate_pair(sd1, sd2, pair, sample_ts, hoist=False)
        recompute_s = time.perf_counter() - t0
        O.interpolate_pair(sd1, sd2, pair1, [0.5], hoist=False)
        t0 = time.perf_counter()
        want1 = O.interpolate_pair(sd1, sd2, pair1, [0.5], hoist=False)
        c1_s = time.perf_counter() - t0
    out = {"value": round(len(sample_ts) / recompute_s, 4), "unit": "frames/s", "cores": best, "kind": "port",
           "cores_note": "%d threads = the best of a sweep over a 736x1280 stage-1 pass; %d physical cores (%d logical CPUs usable) are present - "
                         "torch's batch-1 CPU convolutions do not scale beyond that" % (best, phys, logical),
           "sample": "reference-style loop (stage 1 + stage 2 per t) over 3 of the 7 t of one 736x1280 pair, torch CPU fp32 oracle, warm, "
                     "%d threads (best of a sweep over a 736x1280 stage-1 pass): %.1f s" % (best, recompute_s),
           "hoisted": {"value": round(N_T / hoisted_s, 4), "seconds": round(hoisted_s, 2), "sample": "1 pair x 7 t, stage 1 once"},
           "recompute": {"value": round(len(sample_ts) / recompute_s, 4), "seconds": round(recompute_s, 2), "frames": len(sample_ts),
                         "stage1_first_call_seconds": round(s1_cold, 2)},
           "c1_256x256": {"value": round(1.0 / c1_s, 3), "unit": "frames/s", "seconds": round(c1_s, 3)},
           "thread_sweep_c2_stage1_seconds": {str(k): round(v, 3) for k, v in sweep.items()},
           "host": {"logical_cpus_usable": logical, "physical_cores": phys}}
    return out, want, (x1, want1[0])


def family_parity(cfg, dev, weights, frames, modes, want32=None, x=None):
    """max|HIP - oracle| at 736x1280, t = 0.5, for one (weight family, frame family) in the given fp32 modes, beside the oracle's OWN
    rounding d = max|oracle fp32 - oracle float64| (tests/test_hip_model.py::test_720p_parity_families_both_fp32_forms
    asserts the same relations).  want32: the fp32 oracle frame at t = 0.5 when the caller already has it."""
    from models.superslomo_r import FullModel
    from oracle import ssm_oracle as O
    from ssm_amd.weights import synthetic_frames, synthetic_state_dict
    sdA, sdB = synthetic_state_dict(1, family=weights), synthetic_state_dict(2, family=weights)
    if x is None:
        x = synthetic_frames(2, H_IN, W_IN, seed=42 if frames == "texture" else 7, family=frames)
    pair = torch.cat([x[:, 0], x[:, 1]], 1)
    with torch.no_grad():
        if want32 is None:
            want32 = O.interpolate_pair(sdA, sdB, pair, [0.5])[0]
        w64 = O.interpolate_pair({k: v.double() for k, v in sdA.items()}, {k: v.double() for k, v in sdB.items()}, pair.double(), [0.5])[0]

    def st(e):
        e = e.abs().flatten()
        return {"max_abs": float(e.max()), "p9999": float(e.kthvalue(int(e.numel() * 0.9999)).values)}
    d = st(want32.double() - w64)
    rec = {"weights": weights, "frames": frames, "t": 0.5, "oracle_fp32_vs_f64": d, "modes": {}}
    m2 = FullModel(cfg)
    m2.stage1_model.load_state_dict(sdA)
    m2.stage2_model.load_state_dict(sdB)
    m2 = m2.to(dev).eval()
    for mode in modes:
        m2.precision = mode
        got = m2.interpolate(x.to(dev), [0.5]).cpu()
        a, b = st(got - want32), st(got.double() - w64)
        rec["modes"][mode] = {"max_abs_vs_oracle": a["max_abs"], "p9999_vs_oracle": a["p9999"], "max_abs_vs_f64": b["max_abs"],
                              "vs_oracle_in_units_of_d": round(a["max_abs"] / d["max_abs"], 3)}
    m2._drop_plans()
    del m2
    torch.cuda.empty_cache()
    return rec


def io_legs(dev, h, w, reps=10):
    """Measured PCIe legs of the uint8 frame path (ssm_amd.frames): 2 frames in, 7 frames out, pinned host memory."""
    from ssm_amd.frames import frames_from_u8, frames_to_u8
    from ssm_amd.weights import synthetic_frames_u8
    u8 = synthetic_frames_u8(2, h, w, seed=42).permute(0, 2, 3, 1).contiguous().pin_memory()      # [2,h,w,3]
    host_out = torch.empty(N_T, h, w, 3, dtype=torch.uint8).pin_memory()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    acc = [0.0] * 4
    for i in range(reps + 2):
        ev[0].record()
        d = u8.to(dev, non_blocking=True)
        ev[1].record()
        x = frames_from_u8(d)
        ev[2].record()
        y = frames_to_u8(x[:1].expand(N_T, -1, -1, -1).contiguous(), h, w)
        ev[3].record()
        host_out.copy_(y, non_blocking=True)
        ev[4].record()
        torch.cuda.synchronize(dev)
        if i >= 2:
            acc[0] += ev[0].elapsed_time(ev[1])
            acc[1] += ev[1].elapsed_time(ev[2])
            acc[3] += ev[3].elapsed_time(ev[4])
    return {"h2d_ms": round(acc[0] / reps, 4), "h2d_bytes": u8.numel(), "ingest_kernel_ms": round(acc[1] / reps, 4),
            "d2h_ms": round(acc[3] / reps, 4), "d2h_bytes": host_out.numel(),
            "note": "per pair: 2 uint8 frames in, 7 uint8 frames out, pinned memory; not part of `value` (inputs resident in HBM)"}


def streamed_with_io(pipe, dev, h, w, P, PB, t_dev, steps, sdist, sync, cfg=None):
    """The evaluator's loop END TO END (scripts/evaluate_interpolation_results.py:246-278: `.cuda().float()` in, `.cpu().numpy()` out per
    batch), streamed: uint8 pairs in PINNED host memory -> H2D -> ssm_frames_from_u8_fwd (normalise + pad) -> the timed pipeline ->
    ssm_frames_to_u8_fwd (crop + denormalise + uint8) -> D2H into pinned memory.  Every leg of a pass is queued on that pass's HIP stream:
    with `streams` passes in flight the copy engines move one pass's frames while the other passes' convolutions run, and nothing
    synchronises the host inside the loop.  Same pairs per step and passes in flight as the headline region; never part of `value`."""
    from ssm_amd.frames import frames_from_u8, frames_to_u8
    from ssm_amd.weights import synthetic_frames_u8
    nb = P // PB
    host_in = [torch.cat([synthetic_frames_u8(2, h, w, seed=42 + b * PB + j).permute(0, 2, 3, 1) for j in range(PB)], 0).contiguous().pin_memory()
               for b in range(nb)]                                                       # [2 PB, h, w, 3] uint8 per pass
    dev_in = [torch.empty(2 * PB, h, w, 3, dtype=torch.uint8, device=dev) for _ in range(pipe.n)]
    host_out = [torch.empty(PB * N_T, h, w, 3, dtype=torch.uint8).pin_memory() for _ in range(pipe.n)]
    state = {"i": 0}

    def step():
        for b in range(nb):
            k = state["i"] % pipe.n
            state["i"] += 1
            with torch.cuda.stream(pipe.streams[k]):
                dev_in[k].copy_(host_in[b], non_blocking=True)
                x = frames_from_u8(dev_in[k], cfg)                                      # [2 PB, 3, Hp, Wp]
                frames = pipe.engines[k].run(x.view(PB, 6, x.shape[2], x.shape[3]), t_dev, False)
                host_out[k].copy_(frames_to_u8(frames, h, w, cfg), non_blocking=True)

    for _ in range(2):
        step()
    sync()
    elapsed = sdist.timed_steps(step, steps, 0, sync)
    # the streamed frames are the frames of the resident-input path: the last pass of the last step against a direct evaluation
    k = (state["i"] - 1) % pipe.n
    x = frames_from_u8(host_in[nb - 1].to(dev), cfg)
    want = frames_to_u8(pipe.engines[k].run(x.view(PB, 6, x.shape[2], x.shape[3]), t_dev, False), h, w, cfg).cpu()
    sync()
    return {"value_with_io": round(N_T * P * steps / elapsed, 3), "unit": "frames/s", "ms_per_pair": round(1e3 * elapsed / steps / P, 3),
            "pairs": P * steps, "bitwise_equal_to_resident_path": bool(torch.equal(host_out[k], want)),
            "bytes_per_pair": {"h2d": 2 * h * w * 3, "d2h": N_T * h * w * 3},
            "note": "uint8 frames in pinned host memory -> H2D -> normalise + pad kernel -> the same pipeline (same passes in flight) -> crop + "
                    "denormalise + uint8 kernel -> D2H to pinned memory, all on the pass's stream, timed like `value` over the same number of "
                    "steps; `value` itself has its inputs resident in HBM (SURVEY 8d)"}


def infer_bench(args):
    from ssm_amd import dist as sdist
    from ssm_amd.config import load_config, synthetic_weight_overrides
    from ssm_amd.engine import KernelTimer, PairPipeline, UNetPlan
    from ssm_amd.weights import synthetic_frames, synthetic_state_dict
    from models.superslomo_r import FullModel

    rank, local_rank, world, dev = setup_ranks(args)
    cfg = load_config("superslomo_original.ini", synthetic_weight_overrides())   # documented override: no weights ship
    model = FullModel(cfg)
    sd1, sd2 = synthetic_state_dict(1), synthetic_state_dict(2)
    model.stage1_model.load_state_dict(sd1)
    model.stage2_model.load_state_dict(sd2)
    model = model.to(dev).eval()
    headline = args.precision or HEADLINE_PRECISION
    h_in, w_in = (H_IN, W_IN) if args.size == "720p" else (2160, 3840)
    P, PB = args.pairs_per_step, args.pairs_per_batch
    assert P % PB == 0, "--pairs-per-step must be a multiple of --pairs-per-batch"
    xs = [synthetic_frames(2, h_in, w_in, seed=42 + rank * P + i) for i in range(P)]     # [1,2,3,Hp,Wp] each
    Hp, Wp = xs[0].shape[-2:]
    pairs = [torch.cat([x.reshape(1, 6, Hp, Wp) for x in xs[i:i + PB]], 0).to(dev) for i in range(0, P, PB)]    # batches of PB pairs
    t_dev = torch.tensor([i / 8.0 for i in range(1, N_T + 1)], dtype=torch.float32, device=dev)
    sd1d = {k: v.detach() for k, v in model.stage1_model.state_dict().items()}
    sd2d = {k: v.detach() for k, v in model.stage2_model.state_dict().items()}
    flops_pair = conv_flops_per_pair(Hp, Wp, N_T)
    flops_issued = {m: conv_flops_per_pair(Hp, Wp, N_T, m, PB) for m in ("f32", "f32w")}

    def sync():
        torch.cuda.synchronize(dev)

    def run_mode(precision, steps, warmup, timers):
        pipe = PairPipeline(sd1d, sd2d, N_T, Hp, Wp, dev, True, precision, args.streams, graphs=bool(args.graphs), pairs_per_batch=PB)

        def step():                 # P pairs -> 7 frames each; consecutive batches of PB pairs alternate between the streams
            for img6 in pairs:
                pipe.submit(img6, t_dev, want_aux=False)

        for _ in range(warmup):
            step()
        sync()
        clk = start_clock_probe(dev, 2.0) if (rank == 0 and precision == headline and not args.no_clock_probes) else None
        elapsed = sdist.timed_steps(step, steps, 0, sync)
        ms_step = 1e3 * elapsed / steps
        res = {"value": N_T * P * world * steps / elapsed, "ms_per_step": ms_step, "ms_per_pair": ms_step / P, "elapsed_s": elapsed}
        # host time to ISSUE a step's launches (outside the timed region; the queues are empty when it starts, so nothing blocks the host)
        enq = 0.0
        for _ in range(3):
            sync()
            t0 = time.perf_counter()
            step()
            enq += time.perf_counter() - t0
        sync()
        res["host_enqueue_ms_per_step"] = 1e3 * enq / 3
        peak = PEAK_F32_MFMA_TFLOPS if precision in ("f32", "f32w") else PEAK_F16_MFMA_TFLOPS
        ach = flops_pair * P / (ms_step * 1e-3) / 1e12
        kname = {"f32w": "wino4_kernel<*, ups 0|1> (3x3 layers, F(4x4,3x3), v_mfma_f32_16x16x4_f32) + wino2_kernel<*> (3x3 layers on the 1/32 maps, "
                         "F(2x2,3x3)) + wino7s_kernel<*> (7x7 layers, 2x2 blocks of F(4x4,4x4), frequency-split, v_mfma_f32_16x16x4_f32) + wino5s_kernel<*> (5x5 layers, "
                         "F(4x4,5x5), frequency-split, v_mfma_f32_16x16x4_f32) + final_conv_valu_kernel<*> (v_fma_f32)",
                 "f32": "conv_mfma_kernel<*, ups 0|1> (v_mfma_f32_32x32x2_f32) + final_conv_valu_kernel<*> (v_fma_f32)",
                 "f16f8": "conv16_kernel<*> + conv16_multi_kernel<*> + conv16_ups_kernel<*> (v_mfma_f32_32x32x16_f16 + "
                          "v_mfma_scale_f32_32x32x64_f8f6f4)"}.get(precision, "conv16_kernel<*> + conv16_ups_kernel<*> (v_mfma_f32_32x32x16_f16)")
        pmc_file, pmc_keys = PMC_FILES.get(precision, (None, ()))
        pmc = os.path.join(ROOT, "profiles", pmc_file) if pmc_file else ""
        pj = json.load(open(pmc)) if pmc and os.path.exists(pmc) else None
        traffic = sum(pj.get(k, {}).get("hbm_bytes_per_step", 0.0) for k in pmc_keys) if pj else None
        fi = flops_issued.get(precision, flops_pair)
        iss = fi * P / (ms_step * 1e-3) / 1e12
        if precision in flops_issued:
            issued_note = ("the multiply-adds the matrix cores execute: the t-independent input channels of stage 2's conv1a (6 of 16) and "
                           "conv7a (512 of 1024) are convolved once per pair instead of once per t" +
                           ("; Winograd layers: direct-form FLOP x 1/4 (F(4x4,3x3)) or x 16/36 (F(2x2,3x3)), x 1/4 (7x7 as 2x2 blocks of F(4x4,4x4)), "
                            "x 64/400 (5x5 as F(4x4,5x5)), per layer as the plan picks the form" if precision == "f32w" else ""))
        else:
            issued_note = "direct-form FLOP (each product costs %s narrow MFMA operations in this mode)" % \
                          {"f16x3": "3 fp16", "f16f8": "1 fp16 + 2 fp8"}.get(precision, "1")
        res["roofline"] = {"bound": "mfma", "kernel": kname + ", the conv launches of a pair", "achieved": round(iss, 2), "peak": peak,
                           "unit": "TFLOP/s", "frac": round(iss / peak, 4),
                           "region": "the timed region: FLOP issued on the matrix cores for %d pairs / its wall time (%d batch(es) of %d "
                                     "pair(s) in flight; non-conv kernels and gaps count against it)" % (P * steps, args.streams, PB),
                           "flop_per_pair": fi, "flop_note": issued_note,
                           "algorithmic": {"flop_per_pair": flops_pair, "achieved": round(ach, 2), "x_peak": round(ach / peak, 4),
                                           "note": "direct-form FLOP as SURVEY 8d counts them (5.855 TFLOP per pair: stage 1 once, stage 2 "
                                                   "in full per t) / the same wall time: the rate a direct-form kernel would need for this "
                                                   "frame rate; above the matrix peak where the multiply-saving forms are used"},
                           "traffic": traffic or None,
                           "traffic_note": "HBM bytes per pair of the conv launches: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE from "
                                           "separate rocprofv3 --pmc passes (tools/pmc_traffic.sh): profiles/%s" % (pmc_file or "-")}
        if clk is not None:
            ghz = read_clock_probe(clk)
            if ghz:
                res["roofline"]["shader_clock"] = {
                    "ghz": round(ghz, 3), "peak_at_clock": round(peak * ghz / 2.4, 1), "frac_at_clock": round(iss / (peak * ghz / 2.4), 4),
                    "note": "average shader clock over the first 2 s of the timed region (one resident wave comparing s_memtime with the "
                            "100 MHz s_memrealtime, tools/clock_probe.hip); peak_at_clock = peak x ghz / 2.4 (single 7x7 / 5x5 layers run "
                            "back to back are power-managed down to 2.2 GHz, the mix of a whole pair holds the clock)"}
        if timers and rank == 0:
            # per-kernel brackets: ONE stream (with several pairs in flight the spans overlap and cannot be attributed)
            solo = pipe.engines[0]
            for img6 in pairs[:2]:
                solo.run(img6, t_dev, want_aux=False)
            sync()
            timer = KernelTimer()
            UNetPlan.timer = timer
            nb_solo = max(1, min(len(pairs), 8 // PB))
            n_solo = nb_solo * PB
            t0 = time.perf_counter()
            for img6 in pairs[:nb_solo]:
                solo.run(img6, t_dev, want_aux=False)
            sync()
            solo_ms = 1e3 * (time.perf_counter() - t0) / n_solo
            UNetPlan.timer = None
            summ = timer.summary()
            conv = summ["conv"]
            conv_ms = conv["ms"] / n_solo
            kach = flops_pair / (conv_ms * 1e-3) / 1e12
            mfma_per_prod = {"f16x3": 3, "f16f8": 1.5, "f32w": flops_issued["f32w"] / flops_pair, "f32": flops_issued["f32"] / flops_pair}.get(precision, 1)
            res["roofline"]["detail"] = {"region": "%d single-stream pairs run right after the timed region, HIP-event brackets around every "
                                                   "launch on the launch stream" % n_solo,
                                         "launches_per_batch": conv["launches"] // nb_solo, "pairs_per_batch": PB, "ms_per_pair_in_kernel": round(conv_ms, 3),
                                         "achieved_in_kernel": round(fi / (conv_ms * 1e-3) / 1e12, 2),
                                         "frac_in_kernel": round(fi / (conv_ms * 1e-3) / 1e12 / peak, 4),
                                         "algorithmic_achieved_in_kernel": round(kach, 2),
                                         "mfma_issue_frac_in_kernel": round(mfma_per_prod * kach / peak, 4),
                                         "wall_ms_per_pair_single_stream": round(solo_ms, 3)}
            if precision in flops_issued:
                res["roofline"]["families"] = family_rooflines(conv["by_name"], n_solo, precision, peak, PB * N_T, Hp, Wp)
                if precision == headline and not args.no_clock_probes:
                    res["roofline"]["family_clocks"] = family_clocks(solo, dev, peak, res["roofline"]["families"])
            wk = summ["warp"]
            wms = wk["ms"] / n_solo
            wach = wk["bytes"] / n_solo / (wms * 1e-3) / 1e9
            wt = None
            if pj:
                wt = sum(v.get("hbm_bytes_per_step", 0.0) for k, v in pj.items()
                         if k.startswith("flowinterp_inputs") or k.startswith("synthesize") or k.startswith("final_synth")) or None
            res["roofline_warp"] = {"bound": "hbm", "kernel": ", ".join(sorted(wk["by_name"])), "achieved": round(wach, 1),
                                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(wach / PEAK_HBM_GBS, 4), "traffic": wt,
                                    "bytes_per_pair": wk["bytes"] / n_solo, "ms_per_pair_in_kernel": round(wms, 3)}
            if precision == headline:
                res["roofline_warp"]["standalone_warp_kernel"] = warp_kernel_rate(dev, Hp, Wp)
            up = summ.get("upsample_cat", {"ms": 0.0})
            res["time_split_ms_per_pair"] = {"conv": round(conv_ms, 3), "warp_blend": round(wms, 3),
                                             "upsample_cat": round(up["ms"] / n_solo, 3), "wall_single_stream": round(solo_ms, 3),
                                             "wall_timed_region": round(ms_step / P, 3)}
            if args.detail:
                det = {fam: {n: {"ms_per_pair": v[0] / n_solo, "tflops": (v[1] / v[0] / 1e9 if v[0] > 0 else 0.0)}
                             for n, v in d["by_name"].items()} for fam, d in summ.items()}
                path = args.detail if precision == headline else args.detail.replace(".json", "") + "_" + precision + ".json"
                with open(path, "w") as f:
                    json.dump(det, f, indent=1)
        res["_pipe"] = pipe
        return res

    timers = not args.no_kernel_timers
    main_res = run_mode(headline, args.steps, args.warmup, timers)
    out = {
        "metric": "interpolated 1280x720 frames/sec" if args.size == "720p" else "interpolated 3840x2160 frames/sec",
        "value": round(main_res["value"], 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(main_res["ms_per_step"], 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE_NOTE[headline], "data": "synthetic",
        "config": {"precision": headline, "algorithm": ALGO_NOTE.get(headline, "see dtype"),
                   "workload": "superslomo_original.ini inference: synthetic %dx%d pairs (padded %dx%d) -> 7 intermediates t=i/8 each, "
                               "stage 1 once per pair, random-init (deterministic) weights" % (w_in, h_in, Wp, Hp),
                   "pairs_per_step": P, "pairs_per_batch": PB, "frames_per_step": N_T * P, "ms_per_pair": round(main_res["ms_per_pair"], 3),
                   "streams_per_gpu": args.streams,
                   "parallelism": "pairs sharded, %d rank(s); %d batch(es) of %d pair(s) in flight per GPU on separate HIP streams"
                                  % (world, args.streams, PB)},
    }
    for k in ("roofline", "roofline_warp", "time_split_ms_per_pair"):
        if k in main_res:
            out[k] = main_res[k]
    he = main_res["host_enqueue_ms_per_step"]
    out["host_enqueue_ms_per_step"] = round(he, 3)
    out["host"] = {"enqueue_ms_per_step": round(he, 3), "enqueue_share_of_step": round(he / main_res["ms_per_step"], 3),
                   "usable_cpus": usable_cpus(), "cpus_of_this_rank": args.rank_cpus,
                   "hbm": dict(getattr(args, "hbm", {}), peak_allocated_gb=round(torch.cuda.max_memory_allocated(dev) / 1e9, 1)),
                   "note": "one host thread per rank issues the launches; at N ranks per host the ranks need N x this share of a core each "
                           "(DESIGN 6): the step is GPU-bound as long as the share stays below 1"}
    if rank == 0 and world == 1 and not args.no_io and args.size == "720p":
        wio = streamed_with_io(main_res["_pipe"], dev, h_in, w_in, P, PB, t_dev, args.steps, sdist, sync, cfg)
        wio["vs_value"] = round(wio["value_with_io"] / out["value"], 4)
        out["with_io"] = wio
    if torch.distributed.is_initialized():          # rendezvous, timing barrier and max-over-ranks reduction ran on this backend
        out["dist"] = {"backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size()}

    results = {headline: main_res}
    # side modes: the 720p line reports the opt-in modes beside the headline; at 4K only what --modes-4k names (BASELINE config 5 as
    # worded - "fp16 MFMA conv path" - beside the f32w answer)
    side = ([m for m in args.modes.split(",") if m and m != headline] if args.size == "720p" else
            [m for m in args.modes_4k.split(",") if m and m != headline]) if world == 1 else []
    for m in side:
        main_res.pop("_pipe", None)     # free the previous mode's activations
        torch.cuda.empty_cache()
        r = results[m] = run_mode(m, max(2, args.steps // 2), args.warmup, timers)
        out.setdefault("modes", {})[m] = {"value": round(r["value"], 3), "unit": "frames/s", "ms_per_pair": round(r["ms_per_pair"], 3),
                                          "dtype": DTYPE_NOTE[m], "roofline": r["roofline"]}
        if m in ALGO_NOTE:
            out["modes"][m]["algorithm"] = ALGO_NOTE[m]

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.size == "720p":
        pair = torch.cat([xs[0][:, 0], xs[0][:, 1]], 1)
        base, want, (x1, want1) = cpu_baseline(sd1, sd2, pair)
        out["cpu_baseline"] = base
        # BASELINE configs[0] (plumbing): superslomo_eval.ini shape, one 256x256 pair, t = 0.5, HIP (headline mode) beside the CPU oracle
        model.precision = headline
        x1d = x1.to(dev)
        got1 = model.interpolate(x1d, [0.5])
        sync()
        t0 = time.perf_counter()
        for _ in range(20):
            model.interpolate(x1d, [0.5])
        sync()
        hip1 = 20.0 / (time.perf_counter() - t0)
        out["config1_256x256"] = {"hip_frames_per_s": round(hip1, 1), "cpu_frames_per_s": base["c1_256x256"]["value"],
                                  "max_abs_vs_oracle": float((got1.cpu() - want1).abs().max()), "mode": headline,
                                  "note": "one pair, one t, stage 1 not hoisted, one stream (latency, not throughput)"}
        ts = [i / 8.0 for i in range(1, N_T + 1)]
        par = {}
        for m in results:
            model.precision = m
            got = model.interpolate(xs[0].to(dev), ts).cpu()
            per_t = [float((got[i:i + 1] - want[i]).abs().max()) for i in range(N_T)]
            par[m] = {"max_abs_vs_oracle": max(per_t), "per_t": [round(e, 7) for e in per_t]}
            if m != headline:
                out["modes"][m]["parity"] = par[m]
        # Families: two weight families x two frame families, BOTH fp32 modes, at t = 0.5, each beside the
        # oracle's own fp32-vs-float64 distance d on that input.  `max_abs_vs_oracle` of the line = the worst LOW-GRADIENT family (the
        # imagery the 1e-3 contract is about); the hard-edge families are reported in units of d, which is itself > 1e-3 there.
        fams = []
        if not args.no_second_family:
            model._drop_plans()
            torch.cuda.empty_cache()
            for fw, ff in (("uniform", "texture"), ("smooth", "texture"), ("uniform", "edges"), ("smooth", "edges")):
                first = (fw, ff) == ("uniform", "texture")
                fams.append(family_parity(cfg, dev, fw, ff, ("f32w", "f32"), want32=want[3] if first else None, x=xs[0] if first else None))
        low = [f["modes"][headline]["max_abs_vs_oracle"] for f in fams if f["frames"] == "texture" and headline in f["modes"]]
        allf = [f["modes"][headline]["max_abs_vs_oracle"] for f in fams if headline in f["modes"]]
        in_d = [f["modes"][headline]["vs_oracle_in_units_of_d"] for f in fams if headline in f["modes"]]
        # One line, the whole truth: max_abs_vs_oracle = the TIMED pair over its 7 t; the low-gradient families
        # (the imagery the 1e-3 contract is about) and ALL families incl. the hard-edge ones (where the oracle's own fp32 rounding d is
        # itself > 1e-3) side by side, the worst family also in units of d.
        out["parity"] = {"max_abs_vs_oracle": par[headline]["max_abs_vs_oracle"], "per_t": par[headline]["per_t"], "tolerance": 1e-3,
                         "max_abs_low_gradient_families": max([par[headline]["max_abs_vs_oracle"]] + low),
                         "max_abs_all_families": max([par[headline]["max_abs_vs_oracle"]] + allf),
                         "worst_in_units_of_d": max(in_d) if in_d else None,
                         "tolerance_note": "1e-3 on low-gradient imagery; 2 d on hard edges, d = max|oracle fp32 - oracle fp64| on the same input "
                                           "(1.2-1.7e-3 there: the reference's own CPU fp32 path is that far from its float64 evaluation)",
                         "frames": N_T, "size": "%dx%d" % (Hp, Wp), "mode": headline, "families": fams,
                         "families_note": "weights: uniform = index-hash He-uniform (the fixtures' and the timed family), smooth = He-normal, "
                                          "Gaussian-windowed 7x7 / 5x5 filters, decoder gain 1.25; frames: texture = low-pass texture, 3-px motion, "
                                          "edges = full-contrast rectangles / bars / checkerboards, 28 x 20 px motion.  d = max|oracle fp32 - "
                                          "oracle float64|: what the reference's own fp32 arithmetic loses on that input.  On hard edges d is "
                                          "1.4-1.7e-3 (a step edge of height dI turns a flow rounding difference of e px into dI * e), so no fp32 "
                                          "evaluation - the reference's CPU path included - meets 1e-3 per pixel there; the direct form (f32) sits "
                                          "as far from the oracle as the multiply-saving forms (f32w) do, and stage 1 direct + stage 2 Winograd "
                                          "changes nothing (profiles/r11a_parity_families_720p.txt): f32w stays the default; the test asserts "
                                          "max|HIP - oracle| < 2 d and f32w < 1.25 x max(f32, d) for every family"}
    if rank == 0 and world == 1 and not args.no_io and args.size == "720p":
        out["io"] = io_legs(dev, h_in, w_in)

    if rank == 0 and getattr(args, "configs", None):
        out["configs"] = args.configs
    if rank == 0 and world == 1 and args.size == "4k" and args.parity_4k:
        from oracle import ssm_oracle as O
        torch.set_num_threads(min(usable_cpus(), physical_cores(), 32))
        pair = torch.cat([xs[0][:, 0], xs[0][:, 1]], 1)
        with torch.no_grad():
            want4 = O.interpolate_pair(sd1, sd2, pair, [0.5])[0]
        model.precision = headline
        got4 = model.interpolate(xs[0].to(dev), [0.5]).cpu()
        out["parity"] = {"max_abs_vs_oracle": float((got4 - want4).abs().max()), "tolerance": 1e-3, "frames": 1, "t": 0.5,
                         "size": "%dx%d" % (Hp, Wp), "mode": headline}

        def psnr(got):          # PSNR of the frame in [0, 1] pixel units (denormalised with the ImageNet std) against the fp32 CPU oracle
            from ssm_amd.weights import IMAGENET_STD
            e = (got - want4) * torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
            return float(10.0 * torch.log10(1.0 / (e.double() ** 2).mean()))
        out["parity"]["psnr_db_vs_oracle"] = round(psnr(got4), 2)
        for m in side:          # the narrower modes: PSNR is their parity statement
            model.precision = m
            gm = model.interpolate(xs[0].to(dev), [0.5]).cpu()
            out["modes"][m]["parity"] = {"psnr_db_vs_oracle": round(psnr(gm), 2), "max_abs_vs_oracle": float((gm - want4).abs().max()),
                                         "note": "vs the fp32 CPU oracle, one pair, t = 0.5; this mode is narrower than fp32: PSNR, not the 1e-3 bar"}
        model.precision = headline
    if rank == 0:
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--no-second-family", action="store_true", help="skip the parity leg on the second weight family (saves ~25 s of CPU oracle)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs-per-step", type=int, default=8, help="frame pairs per step (infer mode); 20 steps x 8 = 160 pairs")
    ap.add_argument("--pairs-per-batch", type=int, default=2,
                    help="pairs taken through one engine pass (stage 1 at this batch, stage 2 at 7x it): more workgroups per launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-io", action="store_true", help="skip the measured H2D / D2H legs")
    ap.add_argument("--no-kernel-timers", action="store_true", help="skip the HIP-event brackets (no roofline.detail)")
    ap.add_argument("--no-clock-probes", action="store_true",
                    help="skip the shader-clock probes (the per-family probe re-runs every conv family for 0.7 s: leave it out of "
                         "rocprofv3 kernel-stats runs, whose per-kernel sums are divided by the number of pairs)")
    ap.add_argument("--precision", default=None, choices=["f32", "f32w", "f16x3", "f16", "f16f8"],
                    help="headline conv arithmetic (default f32 = the reference's arithmetic)")
    ap.add_argument("--modes", default="f32,f16x3,f16f8", help="comma list of further modes reported under `modes` (N=1, 720p only); '' = none")
    ap.add_argument("--modes-4k", default="", help="--size 4k: comma list of further modes run after the headline one (e.g. f16: BASELINE config 5 as worded)")
    ap.add_argument("--mode", default="infer", choices=["infer", "train", "recurrent"],
                    help="infer = the headline (BASELINE configs[1]); train = configs[2]: training step on 352x352 crops, "
                         "2 samples per GPU, gradient all-reduce over RCCL; recurrent = configs[3]: superslomo_recurrent.ini "
                         "(N_FRAMES=4, ConvBLSTM bottleneck) at 720p")
    ap.add_argument("--size", default="720p", choices=["720p", "4k"],
                    help="720p = BASELINE configs[1] (the headline); 4k = configs[4] shape (3840x2160, use --precision f16)")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams (= engine passes in flight) per GPU")
    ap.add_argument("--no-perceptual", action="store_true", help="--mode train: leave the VGG16 perceptual loss term out")
    ap.add_argument("--force-allreduce", action="store_true",
                    help="--mode train in a process group of one rank: run the bucketed gradient all-reduce on RCCL anyway")
    ap.add_argument("--graphs", type=int, default=0, help="1: replay each pair's launch sequence from a captured HIP graph")
    ap.add_argument("--detail", default=None, help="write the per-launch event-timer table (JSON) to this path")
    ap.add_argument("--stub", action="store_true", help="CPU/gloo stand-in for the per-pair work (launcher test; no GPU)")
    ap.add_argument("--stub-fail-rank", type=int, default=None, help="--stub: this rank exits with an error before the rendezvous (launcher test)")
    ap.add_argument("--no-configs", action="store_true", help="skip the short runs of BASELINE configs 3, 4, 5 reported under `configs` (N = 1, 720p)")
    ap.add_argument("--parity-4k", action="store_true", help="--size 4k: one t of one pair against the CPU oracle (about a minute of host time)")
    args = ap.parse_args()
    # the CPU-oracle legs (parity, cpu_baseline; world 1 only) run on the host: at most the 16 CPUs a one-GPU job owns on a GPU box - torch's
    # default of one thread per visible core oversubscribes that share 8x (cpu_baseline() sweeps the count itself and reports what it used)
    torch.set_num_threads(max(1, min(16, usable_cpus())))

    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args.gpus)          # before anything touches the GPU; does not return
    args.configs = None
    if (args.gpus == 1 and "RANK" not in os.environ and args.mode == "infer" and args.size == "720p" and not args.stub and not args.no_configs):
        args.configs = collect_configs()          # children first: this process has not initialised the GPU yet
    if args.stub:
        return stub_train_bench(args) if args.mode == "train" else stub_bench(args)
    if args.mode == "recurrent":
        return recurrent_bench(args)
    if args.mode == "train":
        return train_bench(args)
    return infer_bench(args)


if __name__ == "__main__":
    main()
