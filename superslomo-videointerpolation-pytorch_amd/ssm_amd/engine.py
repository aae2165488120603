"""Launch plans for the hot path: frame pair -> intermediate frame(s).

A `UNetPlan` is one U-Net (stage 1 = flow computation, stage 2 = arbitrary-time
flow interpolation) at a fixed (batch, H, W) and precision mode: every activation
pre-allocated (fp32 padded planes in mode f32, HL8 hi/lo-fp16 planes in the split modes),
every filter repacked once, and a fixed sequence of C-ABI calls: 24 convolutions
with fused LeakyReLU / 2x2 mean / two-source concat, the five decoder "a"
convolutions with the concat + bilinear x2 upsample fused into their loader.  Topology restates scripts/models/flow_computation.py:155-289 and
scripts/models/flow_interpolation.py:159-281 of the reference.

`PairEngine` chains stage 1 -> compute_inputs -> stage 2 -> synthesis
(scripts/models/superslomo_r.py:250-293).  Stage 1 is t-independent, so for
several intermediates of one pair it runs ONCE and the t values are batched
through stage 2 (the reference's eval loop recomputes it per t,
scripts/evaluate_interpolation_results.py:234-242 - same numbers, fewer FLOPs).
"""
import os

import torch

from . import hipbind as hb
from .weights import RECURRENT_HIDDEN, RECURRENT_LAYERS, param_key, recurrent_convs, unet_layers


class KernelTimer:
    """HIP-event bracket around kernel launches on the stream they are issued on (torch's
    current stream, which is also what the C ABI receives).  Used by bench.py for the
    roofline figures; off by default."""

    def __init__(self):
        self.spans = []      # (family, name, flops, bytes, ev0, ev1, issued flops)

    def span(self, family, name, flops=0.0, nbytes=0.0, issued=None):
        """flops: direct-form FLOP of the launch; issued: the FLOP its matrix cores execute (Winograd forms issue fewer: ISSUED_FACTOR)."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.spans.append((family, name, flops, nbytes, e0, e1, flops if issued is None else issued))
        return e0, e1

    def summary(self):
        """family -> dict(ms, flops, bytes, launches); call after torch.cuda.synchronize()."""
        out = {}
        for fam, name, fl, nb, e0, e1, isd in self.spans:
            d = out.setdefault(fam, {"ms": 0.0, "flops": 0.0, "issued": 0.0, "bytes": 0.0, "launches": 0, "by_name": {}})
            ms = e0.elapsed_time(e1)
            d["ms"] += ms
            d["flops"] += fl
            d["issued"] += isd
            d["bytes"] += nb
            d["launches"] += 1
            n = d["by_name"].setdefault(name, [0.0, 0.0, 0])
            n[0] += ms
            n[1] += fl
            n[2] += 1
        return out


MODES = ("f32", "f32w", "f16x3", "f16", "f16f8")
# f32   : fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 padded planes, every convolution in the direct form (an fmaf chain per output)
# f32w  : the same fp32 planes and kernels, but the 3x3 convolutions run as Winograd F(2x2,3x3) on the fp32 MFMA (csrc/ssm_wino.hip):
#         all arithmetic fp32, 2.25x fewer matrix-core cycles for those layers; 7x7 / 5x5 / final convs stay direct
# f16x3 : fp16 MFMA on hi/lo-split operands, 3 MFMAs per product, fp32 accumulate (fp32-grade results)
# f16   : fp16 MFMA, hi*hi only (reduced precision; 4K / config 5)
# f16f8 : fp16 MFMA for hi*hi + two block-scaled fp8 MFMAs (K = 4 taps x 16 channels) for the compensation products, on the
#         Q8 form of the HL8 layout (fp32-grade results at half the matrix cost of f16x3)


def base_mode(mode):
    """Layout / kernel family of a precision mode: "f32w" shares everything with "f32" except the algorithm of the 3x3 layers."""
    return "f32" if mode == "f32w" else mode


# fp32 inference plans with several interpolation times per pair: convolve the t-independent input channels of stage 2's conv1a /
# conv7a once per pair (UNetPlan.hoist); $SSM_HOIST=0 evaluates them per t like the reference
HOIST_PAIR_PARTS = os.environ.get("SSM_HOIST", "1") != "0"

# layers of an f32w plan that stay on the direct kernel ($SSM_WINO_SKIP=conv11b,...; "all" = none in Winograd form)
WINO_SKIP = frozenset(n for n in os.environ.get("SSM_WINO_SKIP", "").split(",") if n)

# kernel sizes of an f32w plan that run as 1-D Winograd along x (F(2,7) / F(4,5), csrc/ssm_wino1d.hip); $SSM_WINO1D=7 / 5 / 0 narrows it
WINO1D = os.environ.get("SSM_WINO1D", "57")


# 7x7 layers of an f32w inference plan as 2x2 blocks of F(4x4,4x4) (csrc/ssm_wino7.hip: 12.25 multiplies per output instead of F(2,7)'s 28);
# $SSM_WINO7=0 keeps them on the 1-D form
WINO7 = os.environ.get("SSM_WINO7", "1")
# 5x5 layers of an f32w inference plan as two-dimensional F(4x4,5x5) (csrc/ssm_wino5.hip: 4 multiplies per output instead of F(4,5)'s 10);
# $SSM_WINO5=0 keeps them on the 1-D form
WINO5 = os.environ.get("SSM_WINO5", "1")


# 3x3 layers of an f32w plan as F(4x4,3x3) (csrc/ssm_wino4.hip) instead of F(2x2,3x3): $SSM_WINO4=0 keeps F(2x2); a comma list names layers
WINO4 = os.environ.get("SSM_WINO4", "1")


# ... and in the TRAINING plans (forward and data-gradient convolutions, where the cost model prefers it over F(2x2,3x3)): on since r5
# ($SSM_TRAIN_WINO4=0: off).  Gradient parity holds (tests/test_hip_backward.py green with it: 5.9e-5 at 64x64 like every other form
# that reorders a sum, 3.3e-4 at 2x352x352).  It did not pay while the step had other floors (r4: wall 20.8 -> 21.3 ms; r5 before the
# blocked forward, split-K and the flat element-wise kernels: + 0.3 %); with those gone: 15.46 -> 15.27 ms per step together with
# $SSM_TRAIN_DGRAD_WINO4 (profiles/r14i_more.txt).
TRAIN_WINO4 = os.environ.get("SSM_TRAIN_WINO4", "1")
# The 7x7 / 5x5 layers of a TRAINING plan's forward in the two-dimensional blocked forms of the inference plans (csrc/ssm_wino7.hip,
# ssm_wino5.hip).  r3 kept them direct after measuring the 1-D forms F(2,7) / F(4,5) in the forward: parameter gradients at 64x64 4e-4 from
# CPU autograd (bar 3e-4).  Measured in r5 with the 2-D forms (profiles/r14d_grad_matrix.txt): 5.9e-5 at 64x64 (direct forward: 2e-6;
# the data gradients' form makes no difference), 6.5e-4 at 2x352x352 like the direct forward - inside both bars, every test of
# tests/test_hip_backward.py green - and the step is 4 % faster (16.6 -> 15.95 ms).  $SSM_TRAIN_FWD_BLOCKED=0: direct forward
TRAIN_FWD_BLOCKED = os.environ.get("SSM_TRAIN_FWD_BLOCKED", "1")


# fused-upsample 3x3 layers of an f32w INFERENCE plan in the sub-pixel form (interior: a plain F(4x4,3x3) convolution of the low-res sources with 4 Cout
# effective channels in the 64-cout kernel form; border ring: the fused-upsample kernel; hb.PackedSubpixelWino4): conv11a, the one 32-cout
# fused-upsample layer (1.68 -> 1.36 ms at batch 7, profiles/r11l_subpixel_conv11a.txt); $SSM_WINO4_SUBPIXEL=0 keeps the fused-upsample kernel everywhere
WINO4_SUBPIXEL = tuple(n for n in os.environ.get("SSM_WINO4_SUBPIXEL", "conv11a").split(",") if n not in ("", "0"))


def wino4_enabled(name):
    return WINO4 not in ("0", "") and (WINO4 == "1" or name in WINO4.split(","))


def wino1d_enabled(k):
    """Does mode f32w evaluate the k x k layers in the 1-D Winograd form?  (bench.py: FLOP issued on the matrix cores)"""
    return k in (5, 7) and str(k) in WINO1D


def choose_algo(name, ci, co, k, nb, h, w, ups, wino, wino1d, wino4=None, blocked2d=None):
    """Algorithm of one convolution of an fp32 plan: "direct" (csrc/ssm_conv.hip), "wino" = F(2x2,3x3) (ssm_wino.hip), "wino4" =
    F(4x4,3x3) (ssm_wino4.hip), "wino1d" = F(2,7) / F(4,5) along x (ssm_wino1d.hip), "wino7" = the 7x7 layers as 2x2 blocks of F(4x4,4x4) (ssm_wino7.hip).  wino: the plan is mode f32w; wino1d: it is an
    inference plan (the 8-frequency forms stay out of the training plans).  Pure function of the problem: bench.py uses it to count the
    multiply-adds the matrix cores issue."""
    skip = name in WINO_SKIP or "all" in WINO_SKIP or name == "final_conv"
    wino4 = wino1d if wino4 is None else wino4
    # blocked2d: the plan may run the 7x7 / 5x5 layers in the two-dimensional blocked forms; wino1d: ... and, where those do not apply or
    # are switched off, in the 1-D forms (inference plans only: a training plan takes wino7 / wino5 or else the direct form - the 1-D
    # forms miss its gradient bar, TRAIN_FWD_BLOCKED above)
    blocked2d = wino1d if blocked2d is None else blocked2d
    if blocked2d and not skip and k == 7 and WINO7 not in ("0", "") and hb.wino7_supported(ci, co, h, w, k):
        return "wino7"
    if blocked2d and not skip and k == 5 and WINO5 not in ("0", "") and ci % 4 == 0 and hb.wino5_supported(ci, co, h, w, k):
        return "wino5"
    if wino1d and not skip and wino1d_enabled(k) and hb.wino1d_supported(ci, co, h, w, k):
        return "wino1d"
    if wino and not skip and hb.wino_supported(ci, co, h, w, k):
        if (wino4 and wino4_enabled(name) and hb.wino4_supported(ci, co, h, w, k)
                and (WINO4 != "1" or hb.wino4_preferred(ci, co, nb, h, w, ups))):
            return "wino4"
        return "wino"
    return "direct"


_ALGO_CLASS = {"direct": lambda: hb.PackedConv, "wino": lambda: hb.PackedWino, "wino4": lambda: hb.PackedWino4, "wino1d": lambda: hb.PackedWino1d,
               "wino7": lambda: hb.PackedWino7, "wino5": lambda: hb.PackedWino5}
# multiply-adds issued on the matrix cores per direct-form multiply-add, by algorithm and kernel size
ISSUED_FACTOR = {"direct": lambda k: 1.0, "wino": lambda k: 16.0 / 36.0, "wino4": lambda k: 36.0 / 144.0,
                 "wino1d": lambda k: 8.0 / 14.0 if k == 7 else 8.0 / 20.0, "wino7": lambda k: 196.0 / 784.0,
                 "wino5": lambda k: 64.0 / 400.0}


def issued_factor(pk):
    """Multiply-adds the matrix cores execute per direct-form multiply-add for a packed filter's algorithm (fp32 plans; 1 otherwise)."""
    return ISSUED_FACTOR.get(getattr(pk, "algo", "direct"), ISSUED_FACTOR["direct"])(pk.k)


def conv_fn(pk, ups=False):
    """The launcher that goes with a packed filter's algorithm."""
    if pk.algo == "wino7":
        return hb.conv2d_wino7
    if pk.algo == "wino5":
        return hb.conv2d_wino5
    if pk.algo == "wino1d":
        return hb.conv2d_wino1d
    if pk.algo == "wino4":
        return hb.conv2d_ups_wino4 if ups else hb.conv2d_wino4
    if pk.algo == "wino":
        return hb.conv2d_ups_wino if ups else hb.conv2d_wino
    return hb.conv2d_ups if ups else hb.conv2d


POOLED = ("conv1b", "conv2b", "conv3b", "conv4b", "conv5b")      # 2x2 mean fused into these convs
_SCALE = (("conv10", 2), ("conv11", 1), ("conv1", 1), ("conv2", 2), ("conv3", 4), ("conv4", 8), ("conv5", 16),
          ("conv6", 32), ("conv7", 16), ("conv8", 8), ("conv9", 4), ("fuse_conv", 1), ("final_conv", 1))


def layer_scale(name):
    """Down-sampling factor of the map a layer runs on."""
    for prefix, s in _SCALE:
        if name.startswith(prefix):
            return s
    raise KeyError(name)


class UNetPlan:
    timer = None     # a KernelTimer, or None
    RESCALE_EVERY = int(os.environ.get("SSM_RESCALE_EVERY", "256"))
    fast_layers = frozenset()    # diagnostics: "s<stage>.<layer>" names that run hi*hi only inside an f16x3 plan (tools/precision_scan.py)

    DECODER = ("conv7a", "conv7b", "conv8a", "conv8b", "conv9a", "conv9b", "conv10a", "conv10b", "conv11a", "conv11b",
               "fuse_conv", "final_conv")
    ENCODER_T = ("c1", "c2", "c3", "c4", "c5", "c6")      # encoder-batch tensors the decoder reads
    UPS = ("conv7a", "conv8a", "conv9a", "conv10a", "conv11a")      # convolutions with the concat+upsample fused in

    CONV_OUT = ("t1a", "c1", "t2a", "c2", "t3a", "c3", "t4a", "c4", "t5a", "c5", "t6a", "c6", "t7a", "c7", "t8a", "c8", "t9a", "c9",
                "t10a", "c10", "t11a", "c11", "tf")

    def __init__(self, stage, state_dict, B, H, W, device, cross_skip=True, mode="f32", fuse_upsample=True,
                 bottleneck="CONV", seq_len=1, dec=None, twins=False, final4=None, hoist=None):
        """B = encoder batch.  With a recurrent bottleneck the batch holds `seq_len` windows of B/seq_len
        sequences in time-major order (index = window * S + sequence).  dec = (b0, Bd): the decoder runs on
        encoder batch entries [b0, b0+Bd) only (inference returns the middle window); None = all."""
        assert mode in MODES, "precision mode must be one of %s" % (MODES,)
        self.wino = mode == "f32w"          # 3x3 layers in the Winograd form where the kernel supports the problem
        self.mode_name, mode = mode, base_mode(mode)
        assert bottleneck in ("CONV", "CLSTM", "CGRU"), "Unknown bottleneck type: %s" % bottleneck
        assert B % seq_len == 0
        self.bottleneck = bottleneck
        # twins: every convolution also writes its output as fp32 planes (and the input is kept in fp32) - what the
        # hand-written backward reads (LeakyReLU', weight-gradient operands) when the forward runs on the HL8 / Q8 kernels
        self.twins = bool(twins) and mode != "f32"
        self.scales = {}
        self.dec_b0, self.Bd = dec if dec is not None else (0, B)
        assert 0 <= self.dec_b0 and self.dec_b0 + self.Bd <= B
        self._b0 = 0            # batch offset applied to ENCODER_T views (set while the decoder runs)
        self._Bcur = B
        self.mode, self.hl8, self.q8 = mode, mode != "f32", mode == "f16f8"
        self.fuse_up = bool(fuse_upsample)      # concat+upsample fused into the consumer conv's loader (every mode)
        # 7x7 / 5x5 layers in a Winograd form (choose_algo: the blocked 2-D forms, else the 1-D ones): inference plans, and since r5 the
        # training plans too (TRAIN_FWD_BLOCKED above: the gradient bars hold with the 2-D forms)
        infer_plan = self.fuse_up and not twins
        self.wino1d = self.wino and infer_plan                                      # the 1-D forms: inference plans only
        self.blocked2d = self.wino and (infer_plan or TRAIN_FWD_BLOCKED == "1")     # wino7 / wino5
        self.wino4 = self.wino and (infer_plan or TRAIN_WINO4 == "1")          # F(4x4,3x3) for the 3x3 layers
        # hoist = (B1, G): stage-2 inference plan whose batch holds G interpolation times for each of B1 pairs (entry p*G + i).  The
        # parts of two convolutions' inputs that do not depend on t - the image channels 0:3 / 13:16 of conv1a's 16-channel input
        # (flow_interpolation.py:364-367) and the stage-1 half of the cross-skip concat in front of conv7a (:98-101,224-231) - are
        # convolved ONCE per pair (run_pair_parts) and enter the per-t launches as a pre-activation addend: the same sums in another
        # order, 6/16 of conv1a's and 1/2 of conv7a's multiply-adds done once instead of G times (like stage 1 itself)
        self.hoist = None
        self._pair_parts_ready = False     # hoisted plans: set by run_pair_parts(), asserted by the per-t conv1a / conv7a launches
        if hoist is not None and stage == 2 and mode == "f32" and self.fuse_up and not twins and hoist[1] > 1 and bottleneck == "CONV":
            assert B == hoist[0] * hoist[1] and dec is None
            self.hoist = (int(hoist[0]), int(hoist[1]))
        # final_conv (32 -> 4 / 5 channels) on its own kernel (ssm_final_conv_fwd) instead of a 32-cout tile (mode f32, inference plans);
        # stage 2 can then run the synthesis in its epilogue (run_decoder(synth=...))
        self.final4 = (mode == "f32" and self.fuse_up) if final4 is None else (bool(final4) and mode == "f32")
        if self.q8:
            assert self.fuse_up, "mode f16f8 covers the plan with the concat+upsample fused into the convolutions"
        if H % 32 or W % 32:
            raise AssertionError("H and W must be multiples of 32 (got %dx%d): the U-Net pools 5 times "
                                 "and concatenates skips (unchecked in the reference, fails in torch.cat)" % (H, W))
        self.stage, self.B, self.H, self.W, self.device = stage, B, H, W, device
        self.cross = bool(cross_skip) and stage == 2
        self.layers = {n: (ci, co, k) for n, ci, co, k in unet_layers(stage, cross_skip, bottleneck)}
        self.pk = {}
        self.refresh_weights(state_dict, check_shapes=True)
        self.rnn = None
        if bottleneck != "CONV":
            self.rnn = RecurrentBottleneck(bottleneck, state_dict, B // seq_len, seq_len, H // 32, W // 32, device, self.mode_name)
        Bd = self.Bd
        if self.hl8:
            P = lambda c, s: hb.HPlanes(B, c, H // s, W // s, device, q8=self.q8)  # noqa: E731
            D = lambda c, s: hb.HPlanes(Bd, c, H // s, W // s, device, q8=self.q8)  # noqa: E731
        else:
            P = lambda c, s: hb.Planes(B, c, H // s, W // s, device)  # noqa: E731
            D = lambda c, s: hb.Planes(Bd, c, H // s, W // s, device)  # noqa: E731
        cin0 = self.layers["conv1a"][0]
        cfin = self.layers["final_conv"][1]
        t = self.t = {}
        t["in"] = hb.HPlanes(B, cin0, H, W, device, groups=self.pk["conv1a"].cin_p // 8, q8=self.q8) if self.hl8 else P(cin0, 1)
        t["t1a"], t["c1"], t["p2"] = P(32, 1), P(32, 1), P(32, 2)
        t["t2a"], t["c2"], t["p3"] = P(64, 2), P(64, 2), P(64, 4)
        t["t3a"], t["c3"], t["p4"] = P(128, 4), P(128, 4), P(128, 8)
        t["t4a"], t["c4"], t["p5"] = P(256, 8), P(256, 8), P(256, 16)
        t["t5a"], t["c5"], t["p6"] = P(512, 16), P(512, 16), P(512, 32)
        t["c6"] = P(512, 32)
        if bottleneck == "CONV":
            t["t6a"] = P(512, 32)
        if not self.fuse_up:        # materialised concat+upsample tensors (the largest activations of the net)
            t["u7"] = D(1024 if self.cross else 512, 16)
            t["u8"], t["u9"], t["u10"], t["u11"] = D(1024, 8), D(512, 4), D(256, 2), D(128, 1)
        t["t7a"], t["c7"] = D(512, 16), D(512, 16)
        t["t8a"], t["c8"] = D(256, 8), D(256, 8)
        t["t9a"], t["c9"] = D(128, 4), D(128, 4)
        t["t10a"], t["c10"] = D(64, 2), D(64, 2)
        t["t11a"], t["c11"] = D(32, 1), D(32, 1)
        t["tf"] = D(32, 1)
        t["out"] = hb.Planes(Bd, cfin, H, W, device)     # final_conv always leaves fp32 planes (flows / logits)
        if self.hoist:
            t["pair1a"] = hb.Planes(self.hoist[0], 32, H, W, device)                  # conv1a's image-channel partial sums, per pair
            if self.cross:
                t["pair7a"] = hb.Planes(self.hoist[0], 512, H // 16, W // 16, device)  # conv7a's stage-1 half, per pair
        # (outer decoder levels run in the sub-pixel form on inference plans: _subpixel / ssm_amd.subpixel, profiles/DESIGN_history_r1-r3.md 3.1a)
        self.f32 = {}
        if self.twins:
            assert bottleneck == "CONV" and dec is None, "fp32 twins are for the training plan"
            for name in self.CONV_OUT + ("in",):
                ref = t[name]
                self.f32[name] = hb.Planes(B, ref.C if name != "in" else cin0, ref.H, ref.W, device)

    def refresh_weights(self, state_dict, check_shapes=False):
        """(Re)pack every filter from `state_dict` (training: the parameters change each optimizer step).  Mode f16f8 repacks all
        layers with ONE launch once the filters exist (hb.PackBatch over the parameters' storage); the first call, and any call
        with tensors at other addresses, goes layer by layer (it also chooses the power-of-two pre-scales on the host)."""
        self.sp, self._sd_for_sp = {}, state_dict        # sub-pixel decoder levels are rebuilt from the new weights on first use
        self.pk_pair = {}
        tensors = []
        for name in self.layers:
            tensors += [state_dict[param_key(name, "weight")], state_dict[param_key(name, "bias")]]
        batchable = self.q8 and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in tensors)
        # the one-launch repack reuses the power-of-two pre-scales chosen at the first pack; every RESCALE_EVERY-th refresh goes
        # the slow way and chooses them again from the current max|w| (one host sync), so filters that grew or shrank by a few
        # octaves during training get their fp16 / e4m3 range back
        self._refreshes = getattr(self, "_refreshes", 0) + 1
        if self._refreshes % self.RESCALE_EVERY == 0:
            self.scales, self._pack = {}, None
        if batchable and getattr(self, "_pack", None) is not None and self._pack[0] == hb.PackBatch.key(tensors):
            self._pack[1].run()
            if getattr(self, "rnn", None) is not None:
                self.rnn.refresh_weights(state_dict)
            return
        # fp32 plans (training repacks every step): one launch for all filters once they exist; hoisted inference plans pack slices
        batch32 = (not self.hl8 and not self.hoist and os.environ.get("SSM_PACK_BATCH", "1") != "0"
                   and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in tensors))
        if batch32 and getattr(self, "_pack32", None) is not None and self._pack32[0] == hb.PackBatch.key(tensors):
            self._pack32[1].run()
            if "final_conv" in self.layers:
                self.final_wb = (state_dict[param_key("final_conv", "weight")].detach(), state_dict[param_key("final_conv", "bias")].detach())
            if getattr(self, "rnn", None) is not None:
                self.rnn.refresh_weights(state_dict)
            return
        for name, (ci, co, k) in self.layers.items():
            w = state_dict[param_key(name, "weight")].to(device=self.device, dtype=torch.float32)
            b = state_dict[param_key(name, "bias")].to(device=self.device, dtype=torch.float32)
            if check_shapes:
                assert tuple(w.shape) == (co, ci, k, k), "%s: weight shape %s != %s" % (name, tuple(w.shape), (co, ci, k, k))
            s = layer_scale(name)
            if name == "final_conv":
                self.final_wb = (w.contiguous(), b.contiguous())
            if self.hl8:
                self.pk[name] = hb.PackedConv16(w, b, self.W // s, q8=self.q8, ups=self.fuse_up and name in self.UPS,
                                                scale=self.scales.get(name))
                self.scales.setdefault(name, self.pk[name].scale)      # later repacks (training) skip the host-side max|w|
            else:
                nb = self.Bd if name in self.DECODER else self.B
                ups = self.fuse_up and name in self.UPS
                cls = _ALGO_CLASS[choose_algo(name, ci, co, k, nb, self.H // s, self.W // s, ups, self.wino, self.wino1d, self.wino4, self.blocked2d)]()
                if self.hoist and name == "conv1a":
                    # per-t part: channels 3:13 (warped frames + estimated flows); per-pair part: the frames themselves in stage 1's
                    # input order (I0 = channels 13:16, I1 = channels 0:3), no bias, no activation
                    zb = torch.zeros_like(b)
                    self.pk[name] = cls(w[:, 3:13].contiguous(), b, nb, self.H // s, self.W // s)
                    self.pk_pair[name] = cls(torch.cat([w[:, 13:16], w[:, 0:3]], 1).contiguous(), zb, self.hoist[0], self.H // s, self.W // s)
                elif self.hoist and name == "conv7a" and self.cross:
                    zb = torch.zeros_like(b)
                    self.pk[name] = cls(w[:, :512].contiguous(), b, nb, self.H // s, self.W // s, ups=ups)
                    self.pk_pair[name] = cls(w[:, 512:].contiguous(), zb, self.hoist[0], self.H // s, self.W // s, ups=ups)
                elif (ups and cls is hb.PackedWino4 and name in WINO4_SUBPIXEL and not self.twins
                      and hb.subpixel_wino4_supported(ci, co, self.H // s, self.W // s)):
                    self.pk[name] = hb.PackedSubpixelWino4(w, b, nb, self.H // s, self.W // s)
                    batch32 = False          # (its effective filters are not part of the one-launch repack)
                else:
                    self.pk[name] = cls(w, b, nb, self.H // s, self.W // s, pool=name in POOLED, ups=ups)
        # mode f32w only: direct-form launches that leave most of the chip idle may run split over the input channels (hb.conv2d) - the
        # sums of this mode are reordered by its Winograd forms anyway; mode f32 keeps one fmaf chain per output, like the reference
        for pkk in list(self.pk.values()) + list(getattr(self, "pk_pair", {}).values()):
            if self.wino and isinstance(pkk, hb.PackedConv):
                pkk.split_ok = True
        self._pack32 = None
        if batch32:
            entries = [(self.pk[name], state_dict[param_key(name, "weight")], state_dict[param_key(name, "bias")], False)
                       for name in self.layers]
            self._pack32 = (hb.PackBatch.key(tensors), hb.PackBatch32(entries, self.device))
        self._pack = None
        if batchable:
            entries = [(self.pk[name], state_dict[param_key(name, "weight")], state_dict[param_key(name, "bias")], False)
                       for name in self.layers]
            self._pack = (hb.PackBatch.key(tensors), hb.PackBatch(entries, self.device))
        if getattr(self, "rnn", None) is not None:
            self.rnn.refresh_weights(state_dict)

    def _v(self, name, **kw):
        """View of a plan tensor; encoder-batch tensors are offset to the decoder's slice while it runs."""
        if self._b0 and name in self.ENCODER_T:
            kw["b0"] = self._b0
        return self.t[name].view(**kw)

    def _conv(self, name, src, dst, pool=None, src2=None, lrelu=True):
        pk = self.pk[name]
        s = self.t[src]
        d = self.t[dst]
        c2 = self.t[src2].C if src2 else 0
        tm = self.timer
        if tm is not None:
            fl = 2.0 * self._Bcur * s.H * s.W * pk.cout * pk.cin * pk.k * pk.k
            e0, e1 = tm.span("conv", "s%d.%s" % (self.stage, name), fl, issued=fl * issued_factor(pk))
            e0.record()
        v = self._v
        if self.hl8:
            final = name == "final_conv"
            y32 = v(dst) if final else (self.f32[dst].view() if self.twins else None)
            hb.conv2d_hl8(v(src), s.G * 8, v(src2) if src2 else None, c2, pk, None if final else v(dst),
                          y32, v(pool) if pool else None, self._Bcur, s.H, s.W,
                          lrelu=lrelu, fast=self.mode == "f16" or ("s%d.%s" % (self.stage, name)) in UNetPlan.fast_layers)
        elif self.hoist and name == "conv1a":
            assert self._pair_parts_ready, "hoisted stage-2 plan: run_pair_parts() must run before the per-t launches of a pass"
            conv_fn(pk)(v(src, c0=3), 10, None, 0, pk, v(dst), None, self._Bcur, s.H, s.W, lrelu=lrelu, add=self.t["pair1a"].view(), add_div=self.hoist[1])
        else:
            conv_fn(pk)(v(src), s.C, v(src2) if src2 else None, c2, pk, v(dst), v(pool) if pool else None, self._Bcur, s.H, s.W, lrelu=lrelu)
        if tm is not None:
            e1.record()

    def _bview(self, b, b_planes, b_broadcast, b_b0):
        """View of the second concat source: a plan tensor, or foreign planes (stage-1 encoding) at batch b_b0."""
        if b_planes is not None:
            return b_planes, b_planes.view(broadcast=b_broadcast, b0=b_b0)
        if b:
            return self.t[b], self._v(b)
        return None, None

    SUBPIXEL = tuple(n for n in os.environ.get("SSM_SUBPIXEL", "conv11a,conv10a").split(",") if n not in ("", "0"))

    def _subpixel(self, name, a, b):
        """The layer's SubpixelUpConv, built on first use (None: layer not selected / plan not eligible)."""
        if name not in self.SUBPIXEL or not self.q8 or self.twins or b is None:
            return None
        if name not in self.sp:
            from .subpixel import SubpixelUpConv
            A, Bp = self.t[a], self.t[b]
            wt = self._sd_for_sp[param_key(name, "weight")]
            assert wt.shape[1] == A.C + Bp.C == 8 * (A.G + Bp.G), "sub-pixel form expects whole channel groups"
            self.sp[name] = SubpixelUpConv(wt, self._sd_for_sp[param_key(name, "bias")], A.G, Bp.G, self.Bd, A.H, A.W, self.device)
        return self.sp[name]

    def _up_conv(self, name, a, b, u, dst, b_planes=None, b_broadcast=False, b_b0=0):
        """dst = conv(name)( upsample2x(cat[a, b]) ): fused in one kernel, or via the materialised tensor `u`."""
        if not self.fuse_up:
            self._up(a, b, u, b_planes=b_planes, b_broadcast=b_broadcast, b_b0=b_b0)
            return self._conv(name, u, dst)
        sp = self._subpixel(name, a, b) if b_planes is None else None
        if sp is not None:
            tm = self.timer
            if tm is not None:
                d, pk0 = self.t[dst], self.pk[name]
                e0, e1 = tm.span("conv", "s%d.%s" % (self.stage, name), 2.0 * self._Bcur * d.H * d.W * pk0.cout * pk0.cin * 9)
                e0.record()
            sp.run(lambda y0, x0: self._v(a, y0=y0, x0=x0), lambda y0, x0: self._v(b, y0=y0, x0=x0), self.t[dst])
            if tm is not None:
                e1.record()
            return
        pk = self.pk[name]
        A = self.t[a]
        Bp, bview = self._bview(b, b_planes, b_broadcast, b_b0)
        d = self.t[dst]
        tm = self.timer
        if tm is not None:
            fl = 2.0 * self._Bcur * d.H * d.W * pk.cout * pk.cin * 9
            e0, e1 = tm.span("conv", "s%d.%s" % (self.stage, name), fl, issued=fl * issued_factor(pk))
            e0.record()
        if self.hl8:
            hb.conv2d_ups_hl8(self._v(a), A.G * 8, bview, Bp.G * 8 if Bp else 0, pk,
                              d.view(), self.f32[dst].view() if self.twins else None, self._Bcur, d.H, d.W, lrelu=True,
                              fast=self.mode == "f16" or ("s%d.%s" % (self.stage, name)) in UNetPlan.fast_layers)
        elif isinstance(pk, hb.PackedSubpixelWino4) and b_planes is None:
            hb.conv2d_ups_subpixel_wino4(lambda y0, x0: self._v(a, y0=y0, x0=x0), A.C, (lambda y0, x0: self._v(b, y0=y0, x0=x0)) if b else None,
                                         Bp.C if Bp else 0, pk, lambda y0, x0: d.view(y0=y0, x0=x0), self._Bcur, d.H, d.W, lrelu=True)
        elif self.hoist and name == "conv7a" and self.cross:
            assert self._pair_parts_ready, "hoisted stage-2 plan: run_pair_parts() must run before the per-t launches of a pass"
            conv_fn(pk, True)(self._v(a), A.C, None, 0, pk, d.view(), self._Bcur, d.H, d.W, lrelu=True, add=self.t["pair7a"].view(), add_div=self.hoist[1])
        else:
            conv_fn(pk, True)(self._v(a), A.C, bview, Bp.C if Bp else 0, pk, d.view(), self._Bcur, d.H, d.W, lrelu=True)
        if tm is not None:
            e1.record()

    def _up(self, a, b, dst, b_planes=None, b_broadcast=False, b_b0=0):
        lib = hb.load()
        A = self.t[a]
        Bp, bview = self._bview(b, b_planes, b_broadcast, b_b0)
        tm = self.timer
        if tm is not None:   # algorithmic bytes: read the sources once, write 4x as many pixels
            e0, e1 = tm.span("upsample_cat", "s%d.%s" % (self.stage, dst),
                             nbytes=4.0 * self._Bcur * A.H * A.W * (A.C + (Bp.C if Bp else 0)) * 5)
            e0.record()
        if self.hl8:
            hb.check(lib.ssm_upsample2x_cat_hl8_fwd(self._v(a), A.G, bview if Bp else hb.NULL_HVIEW,
                                                    Bp.G if Bp else 0, self.t[dst].view(), self._Bcur, A.H, A.W,
                                                    hb.stream_ptr()))
        else:
            hb.check(lib.ssm_upsample2x_cat_fwd(self._v(a), A.C, bview if Bp else hb.NULL_VIEW,
                                                Bp.C if Bp else 0, self.t[dst].view(), self._Bcur, A.H, A.W,
                                                hb.stream_ptr()))
        if tm is not None:
            e1.record()

    def run_pair_parts(self, pair_planes, c6_planes):
        """Hoisted plan: the per-pair partial sums of conv1a (over the frame pair, stage 1's 6-channel input planes) and of conv7a
        (over the upsampled stage-1 bottleneck output), B1 entries each - launched once per pass, before the per-t stage-2 batch."""
        B1 = self.hoist[0]
        self._pair_parts_ready = True      # consumed by conv1a / conv7a of this pass, cleared when the decoder has run
        tm = self.timer
        pk = self.pk_pair["conv1a"]
        P = self.t["pair1a"]
        if tm is not None:
            fl = 2.0 * B1 * P.H * P.W * pk.cout * pk.cin * pk.k * pk.k
            e0, e1 = tm.span("conv", "s2.conv1a(pair)", fl, issued=fl * issued_factor(pk))
            e0.record()
        conv_fn(pk)(pair_planes.view(), 6, None, 0, pk, P.view(), None, B1, P.H, P.W, lrelu=False)
        if tm is not None:
            e1.record()
        if self.cross:
            pk = self.pk_pair["conv7a"]
            P = self.t["pair7a"]
            if tm is not None:
                fl = 2.0 * B1 * P.H * P.W * pk.cout * pk.cin * 9
                e0, e1 = tm.span("conv", "s2.conv7a(pair)", fl, issued=fl * issued_factor(pk))
                e0.record()
            conv_fn(pk, True)(c6_planes.view(), 512, None, 0, pk, P.view(), B1, P.H, P.W, lrelu=False)
            if tm is not None:
                e1.record()

    def run(self, cross_planes=None, cross_broadcast=False, cross_b0=0, synth=None):
        """Input must already be in self.t['in'].  Returns the Planes of final_conv's output (decoder batch).
        cross_planes: stage-1 conv6 output; its entries [cross_b0, cross_b0+Bd) pair with the decoder's batch
        (cross_broadcast: entry cross_b0 serves the whole batch).  synth: see run_decoder."""
        self.run_encoder()
        self.run_bottleneck()
        return self.run_decoder(cross_planes, cross_broadcast, cross_b0, synth)

    def run_encoder(self):
        c = self._conv
        self._b0, self._Bcur = 0, self.B
        c("conv1a", "in", "t1a")
        c("conv1b", "t1a", "c1", pool="p2")
        c("conv2a", "p2", "t2a")
        c("conv2b", "t2a", "c2", pool="p3")
        c("conv3a", "p3", "t3a")
        c("conv3b", "t3a", "c3", pool="p4")
        c("conv4a", "p4", "t4a")
        c("conv4b", "t4a", "c4", pool="p5")
        c("conv5a", "p5", "t5a")
        c("conv5b", "t5a", "c5", pool="p6")

    def run_family(self, k):
        """Diagnostics (bench.py: shader clock per kernel family): launch this plan's convolutions of kernel size k once more on
        the tensors a previous run() left resident - k = 7: conv1a, conv1b; 5: conv2a, conv2b; 3: every 3x3 layer but final_conv."""
        c = self._conv
        self._b0, self._Bcur = 0, self.B
        self._pair_parts_ready = self.hoist is not None      # the previous pass's partial sums are still resident
        if k == 7:
            c("conv1a", "in", "t1a")
            c("conv1b", "t1a", "c1", pool="p2")
        elif k == 5:
            c("conv2a", "p2", "t2a")
            c("conv2b", "t2a", "c2", pool="p3")
        else:
            assert self.rnn is None and (self.hoist or not self.cross) and self.dec_b0 == 0 and self.Bd == self.B
            for a, src, dst, pool in (("conv3a", "p3", "t3a", None), ("conv3b", "t3a", "c3", "p4"), ("conv4a", "p4", "t4a", None),
                                      ("conv4b", "t4a", "c4", "p5"), ("conv5a", "p5", "t5a", None), ("conv5b", "t5a", "c5", "p6")):
                c(a, src, dst, pool=pool)
            self.run_bottleneck()
            self._decode(c, self._up_conv, None, False, 0, final=False)

    def run_bottleneck(self):
        self._b0, self._Bcur = 0, self.B
        if self.rnn is not None:
            tm = self.timer
            if tm is not None:
                e0, e1 = tm.span("conv", "s%d.conv6(%s)" % (self.stage, self.bottleneck), self.rnn.flops())
                e0.record()
            self.rnn.run(self.t["p6"], self.t["c6"])
            if tm is not None:
                e1.record()
            return
        self._conv("conv6.0", "p6", "t6a")
        self._conv("conv6.1", "t6a", "c6")

    def run_decoder(self, cross_planes=None, cross_broadcast=False, cross_b0=0, synth=None):
        """synth (plans with final4 only): callable(tf_planes, w, b) that launches final_conv fused with the synthesis
        (ssm_final_conv_fwd with y3) - then the 5-channel map is not written and None is returned."""
        c, uc = self._conv, self._up_conv
        self._b0, self._Bcur = self.dec_b0, self.Bd
        try:
            return self._decode(c, uc, cross_planes, cross_broadcast, cross_b0, synth)
        finally:
            self._b0, self._Bcur = 0, self.B
            self._pair_parts_ready = False

    def _final(self, synth):
        """final_conv: flow_computation.py:145-153 / flow_interpolation.py:149-157."""
        if not self.final4:
            assert synth is None
            self._conv("final_conv", "tf", "out", lrelu=False)
            return self.t["out"]
        tf, out = self.t["tf"], self.t["out"]
        w, b = self.final_wb
        tm = self.timer
        if tm is not None:
            e0, e1 = tm.span("conv", "s%d.final_conv" % self.stage, 2.0 * self._Bcur * tf.H * tf.W * out.C * tf.C * 9)
            e0.record()
        if synth is not None:
            synth(tf, w, b)
        else:
            nv = hb.NULL_VIEW
            hb.check(hb.load().ssm_final_conv_fwd(tf.view(), w.data_ptr(), b.data_ptr(), out.C, out.view(), nv, nv, None, nv, nv,
                                                  self._Bcur, tf.H, tf.W, hb.stream_ptr()))
        if tm is not None:
            e1.record()
        return None if synth is not None else out

    def _decode(self, c, uc, cross_planes, cross_broadcast, cross_b0, synth=None, final=True):
        if self.cross:
            if cross_planes is None and not self.hoist:
                raise RuntimeError("stage 2 was built with CROSS_SKIP but no stage-1 encoding was given")
            uc("conv7a", "c6", None, "u7", "t7a", b_planes=cross_planes, b_broadcast=cross_broadcast, b_b0=cross_b0)
        else:
            uc("conv7a", "c6", None, "u7", "t7a")
        c("conv7b", "t7a", "c7")
        uc("conv8a", "c7", "c5", "u8", "t8a")
        c("conv8b", "t8a", "c8")
        uc("conv9a", "c8", "c4", "u9", "t9a")
        c("conv9b", "t9a", "c9")
        uc("conv10a", "c9", "c3", "u10", "t10a")
        c("conv10b", "t10a", "c10")
        uc("conv11a", "c10", "c2", "u11", "t11a")
        c("conv11b", "t11a", "c11")
        c("fuse_conv", "c11", "tf", src2="c1")
        return self._final(synth) if final else None


class RecurrentBottleneck:
    """conv6 = ConvBLSTM / ConvBGRU(in_channels=512, hidden_channels=512, kernel_size=(3,3), num_layers=2,
    batch_first=True) on the 1/32 maps of a window sequence (reference call sites
    scripts/models/flow_computation.py:73-88,208-211; definition restated from the un-vendored
    SreenivasVRao/ConvGRU-ConvLSTM-PyTorch package - parity unpinned, see oracle/ssm_oracle.py).

    Tensors hold T windows x S sequences in time-major batch order (slot k = batch entries [k*S, (k+1)*S)).
    Each gate convolution over cat[x_t, h_{t-1}] is split by input channel: the x part runs ONCE for the whole
    sequence (batch T*S), the h part once per step (batch S; skipped at the first step where h = 0); the pointwise
    kernel adds the two and writes the new state straight into the next convolution's input layout.  The reverse
    net is the same recurrence walked from the last slot to the first with its outputs stored per slot, which is
    the reference's "reverse the input, run, reverse the output" without moving data."""

    def __init__(self, kind, state_dict, S, T, h, w, device, mode="f32", prefix="conv6."):
        assert kind in ("CLSTM", "CGRU")
        self.kind, self.S, self.T, self.h, self.w, self.device = kind, S, T, h, w, device
        self.wino = mode == "f32w"          # gate convolutions (3x3, 512 / 256 -> 1024 / 512 channels) in the Winograd form
        mode = base_mode(mode)
        self.mode, self.hl8, self.prefix = mode, mode != "f32", prefix
        self.q8 = mode == "f16f8"
        self.flags = hb.SSM_FLAG_Q8 if self.q8 else 0
        self.hid = RECURRENT_HIDDEN
        self.ng = 4 * self.hid if kind == "CLSTM" else 2 * self.hid
        self.pk = {}
        self.refresh_weights(state_dict)
        P32 = lambda B, C: hb.Planes(B, C, h, w, device)  # noqa: E731
        X = (lambda B, C: hb.HPlanes(B, C, h, w, device, q8=self.q8)) if self.hl8 else P32  # noqa: E731
        self.gx, self.gh = P32(T * S, self.ng), P32(S, self.ng)
        self.seq0 = X(T * S, self.hid)                     # layer-0 output sequence (input of layer 1)
        if kind == "CLSTM":
            self.c = [P32(S, self.hid), P32(S, self.hid)]
        else:
            self.cx, self.ch, self.rh = P32(T * S, self.hid), P32(S, self.hid), X(S, self.hid)
            self.h32 = P32(T * S, self.hid) if self.hl8 else None     # fp32 copy of the state for the update rule

    def _pack(self, w, b, batch):
        if self.hl8:
            return hb.PackedConv16(w.contiguous(), b, self.w, q8=self.q8)
        cls = hb.PackedWino if (self.wino and hb.wino_supported(w.shape[1], w.shape[0], self.h, self.w, 3)) else hb.PackedConv
        return cls(w.contiguous(), b, batch, self.h, self.w)

    def refresh_weights(self, state_dict):
        dev = self.device
        for key, cin, hid, cout in recurrent_convs(self.kind):
            full = self.prefix + key[len("conv6."):]
            w = state_dict[full + ".weight"].to(device=dev, dtype=torch.float32)
            b = state_dict[full + ".bias"].to(device=dev, dtype=torch.float32)
            assert tuple(w.shape) == (cout, cin + hid, 3, 3), "%s: weight shape %s" % (full, tuple(w.shape))
            self.pk[key, "x"] = self._pack(w[:, :cin], b, self.T * self.S)
            self.pk[key, "h"] = self._pack(w[:, cin:], torch.zeros_like(b), self.S)

    def flops(self):
        per_px = 0.0
        for key, cin, hid, cout in recurrent_convs(self.kind):
            per_px += 2.0 * 9 * cout * (cin * self.T + hid * (self.T - 1))
        return per_px * self.S * self.h * self.w

    def _conv(self, pk, src_view, cin, dst_view, batch):
        if self.hl8:
            hb.conv2d_hl8(src_view, cin, None, 0, pk, None, dst_view, None, batch, self.h, self.w, lrelu=False,
                          fast=self.mode == "f16")
        else:
            fn = hb.conv2d_wino if pk.algo == "wino" else hb.conv2d
            fn(src_view, cin, None, 0, pk, dst_view, None, batch, self.h, self.w, lrelu=False)

    def _hview(self, planes, ch, slot):
        """View of `hid` channels starting at channel ch of slot `slot` in the layout the convolutions read."""
        return planes.view(ch // 8 if self.hl8 else ch, b0=slot * self.S)

    def _direction(self, net, x, out, ch_out, order):
        lib, st, S, hid = hb.load(), hb.stream_ptr(), self.S, self.hid
        nv, nh = hb.NULL_VIEW, hb.NULL_HVIEW
        for l in range(RECURRENT_LAYERS):
            cell = "conv6.%s.cell_list.%d." % (net, l)
            src, cin = (x, 512) if l == 0 else (self.seq0, hid)
            dst, ch = (self.seq0, 0) if l < RECURRENT_LAYERS - 1 else (out, ch_out)
            gkey = cell + ("conv" if self.kind == "CLSTM" else "conv_gates")
            self._conv(self.pk[gkey, "x"], src.view(), cin, self.gx.view(), self.T * S)
            if self.kind == "CGRU":
                self._conv(self.pk[cell + "conv_can", "x"], src.view(), cin, self.cx.view(), self.T * S)
            prev = None
            for i, k in enumerate(order):
                hv = self._hview(dst, ch, k)
                h16, h32 = (hv, nv) if self.hl8 else (nh, hv)
                if prev is not None:
                    self._conv(self.pk[gkey, "h"], self._hview(dst, ch, prev), hid, self.gh.view(), S)
                if self.kind == "CLSTM":
                    hb.check(lib.ssm_convlstm_cell_fwd(self.gx.view(b0=k * S), self.gh.view() if prev is not None else nv,
                                                       self.c[(i + 1) % 2].view() if prev is not None else nv,
                                                       self.c[i % 2].view(), h32, h16, S, hid, self.h, self.w, self.flags, st))
                else:
                    state = self.h32 if self.hl8 else dst        # fp32 state tensor and the channel it starts at
                    sch = 0 if self.hl8 else ch
                    if self.hl8:
                        h32 = state.view(0, b0=k * S)
                    if prev is not None:
                        hp = state.view(sch, b0=prev * S)
                        r16, r32 = (self.rh.view(), nv) if self.hl8 else (nh, self.rh.view())
                        hb.check(lib.ssm_convgru_reset_fwd(self.gx.view(b0=k * S), self.gh.view(), hp, r32, r16, S, hid,
                                                           self.h, self.w, self.flags, st))
                        self._conv(self.pk[cell + "conv_can", "h"], self.rh.view(), hid, self.ch.view(), S)
                        hb.check(lib.ssm_convgru_update_fwd(self.gx.view(b0=k * S), self.gh.view(), self.cx.view(b0=k * S),
                                                            self.ch.view(), hp, h32, h16, S, hid, self.h, self.w, self.flags, st))
                    else:
                        hb.check(lib.ssm_convgru_update_fwd(self.gx.view(b0=k * S), nv, self.cx.view(b0=k * S), nv, nv,
                                                            h32, h16, S, hid, self.h, self.w, self.flags, st))
                prev = k

    def run(self, x_fwd, out, x_rev=None):
        """x_fwd, out: [T*S, 512, h, w] plan tensors (HL8 in the fp16 modes, fp32 planes in mode f32).
        x_rev: the reverse net's input ALREADY in slot order (slot k = what it consumes for window k); default
        x_fwd, i.e. the reference's x_rev = time-reversed x_fwd."""
        self._direction("forward_net", x_fwd, out, 0, list(range(self.T)))
        self._direction("reverse_net", x_fwd if x_rev is None else x_rev, out, self.hid, list(range(self.T - 1, -1, -1)))
        return out


class PairEngine:
    """stage 1 (batch B1 = pairs) -> compute_inputs -> stage 2 (batch B2 = B1 * G) -> synthesis; stage-2 entry p*G + i is
    pair p at its i-th interpolation time.  G = 1: one t per sample (FullModel.forward).  B1 = 1: the t values of one pair
    (stage-1 tensors batch-broadcast).  B1 > 1 and G > 1: several pairs per pass - the convolutions then see 2-4x the
    workgroups per launch (less tail, fuller small maps) while the three kernels that mix stage-1 and stage-2 tensors run
    once per pair on batch-offset views."""

    def __init__(self, sd1, sd2, B1, B2, H, W, device, cross_skip=True, mode="f32", fuse_upsample=True, twins=False, fuse_final=None):
        assert B2 % B1 == 0, "stage-2 batch must be a multiple of the stage-1 batch (pairs x times)"
        full_mode, mode = mode, base_mode(mode)
        self.twins = bool(twins) and mode != "f32"
        self.B1, self.B2, self.G, self.H, self.W, self.device = B1, B2, B2 // B1, H, W, device
        self.cross = bool(cross_skip)
        self.mode, self.hl8, self.q8 = mode, mode != "f32", mode == "f16f8"
        self.bcast = (B1 == 1 and B2 > 1)
        self.grouped = B1 > 1 and self.G > 1
        self.s1 = UNetPlan(1, sd1, B1, H, W, device, cross_skip, full_mode, fuse_upsample, twins=twins, final4=fuse_final)
        self.s2 = UNetPlan(2, sd2, B2, H, W, device, cross_skip, full_mode, fuse_upsample, twins=twins, final4=fuse_final,
                           hoist=(B1, B2 // B1) if HOIST_PAIR_PARTS else None)
        self.fuse_final = self.s2.final4          # stage 2: final_conv + synthesis in one kernel (no 5-channel map)
        self.t_dev = torch.empty(B2, dtype=torch.float32, device=device)
        self.img = torch.empty(B2, 3, H, W, dtype=torch.float32, device=device)
        self.aux = torch.empty(B2, 5, H, W, dtype=torch.float32, device=device)
        self.img6 = None        # the caller's [B1,6,H,W] pairs, read in place by the gather kernels
        self.est = torch.empty(B2, 4, H, W, dtype=torch.float32, device=device) if self.hl8 else None   # Ft1^ | Ft0^
        self.c6x = None
        if self.grouped and self.cross and not self.s2.hoist:     # stage-1 bottleneck output repeated G times per pair: conv7a's second source
            c6 = self.s1.t["c6"]
            self.c6x = (hb.HPlanes(B2, c6.C, c6.H, c6.W, device, q8=self.q8) if self.hl8 else hb.Planes(B2, c6.C, c6.H, c6.W, device))

    def load_pair(self, img6):
        """img6: [B1,6,H,W] device tensor (I0 | I1 on the channel axis)."""
        assert tuple(img6.shape) == (self.B1, 6, self.H, self.W), "image pair tensor has shape %s" % (tuple(img6.shape),)
        self.img6 = img6.contiguous()
        self.s1.t["in"].load(self.img6)
        if self.twins:
            self.s1.f32["in"].load(self.img6)

    def run_stage1(self):
        return self.s1.run()

    def _groups(self):
        """(pair index or None, first stage-2 entry, entries, stage-1 tensors broadcast?) per launch of a mixing kernel."""
        if self.grouped:
            return [(p, p * self.G, self.G, True) for p in range(self.B1)]
        return [(None, 0, self.B2, self.bcast)]

    def _img6_view(self, p=None, bc=None):
        v = hb.view_of(self.img6 if p is None else self.img6[p:p + 1])
        if self.bcast if bc is None else bc:
            v.sb = 0
        return v

    def _cross_planes(self):
        """(planes, broadcast) of conv7a's cross-skip source for the stage-2 batch."""
        if not self.cross or self.s2.hoist:
            return None, False
        if not self.grouped:
            return self.s1.t["c6"], self.bcast
        c6, x, B1, G = self.s1.t["c6"], self.c6x, self.B1, self.G      # whole padded entries (frame included), one strided copy
        if self.hl8:
            E = c6.G * 2 * c6.Hp * c6.Wp * 8
            x.buf[:self.B2 * E].view(B1, G, E).copy_(c6.buf[:B1 * E].view(B1, 1, E).expand(-1, G, -1))
        else:
            x.full.view(B1, G, c6.C, c6.Hp, c6.Wp).copy_(c6.full.unsqueeze(1).expand(-1, G, -1, -1, -1))
        return self.c6x, False

    def run_stage2(self, t, want_aux=True, want_out5=False):
        """t: [B2] device tensor of interpolation times in (0,1) (or [G]: the same times for every pair).  want_out5: the
        caller reads stage 2's raw 5-channel map (s2.t["out"], the loss terms do) - the fused final_conv + synthesis kernel
        then writes it as well."""
        lib = hb.load()
        st = hb.stream_ptr()
        t = t.reshape(-1)
        if t.numel() == self.G and self.B1 > 1:
            hb.host_op(lambda: self.t_dev.view(self.B1, self.G).copy_(t.view(1, self.G).expand(self.B1, self.G), non_blocking=True))
        else:
            hb.host_op(lambda: self.t_dev.copy_(t, non_blocking=True))
        flow4 = self.s1.t["out"]
        in16 = self.s2.t["in"]
        tm = UNetPlan.timer
        px = float(self.B2 * self.H * self.W)
        tptr = self.t_dev.data_ptr()
        t_only = bool(self.s2.hoist) and not self.hl8      # hoisted plan: the frame channels of the 16-channel tensor are never read
        if tm is not None:   # 104 B/px (read 10 ch, write 16 ch); 80 B/px without the six pass-through frame channels
            e0, e1 = tm.span("warp", "flowinterp_inputs", nbytes=(80.0 if t_only else 104.0) * px)
            e0.record()
        for p, b0, n, bc in self._groups():
            i6 = self._img6_view(p, bc)
            f4 = flow4.view(broadcast=bc, b0=p or 0)
            if self.hl8:
                fn = lib.ssm_flowinterp_inputs_hq8_fwd if self.q8 else lib.ssm_flowinterp_inputs_hl8_fwd
                hb.check(fn(i6, f4, tptr + 4 * b0, in16.view(b0=b0), hb.view_of(self.est[b0:]), n, self.H, self.W, st))
                if self.twins:      # fp32 copy of the 16-channel stage-2 input for the weight gradient of conv1a
                    hb.check(lib.ssm_flowinterp_inputs_fwd(i6, f4, tptr + 4 * b0, self.s2.f32["in"].view(b0=b0), n, self.H, self.W, st))
            else:
                fn = lib.ssm_flowinterp_inputs_t_fwd if t_only else lib.ssm_flowinterp_inputs_fwd
                hb.check(fn(i6, f4, tptr + 4 * b0, in16.view(b0=b0), n, self.H, self.W, st))
        if tm is not None:
            e1.record()
        cross, cbc = self._cross_planes()
        if self.s2.hoist:
            self.s2.run_pair_parts(self.s1.t["in"], self.s1.t["c6"])

        def in16_view(b0):     # the synthesis reads channels 6..9 of its `in16` argument: in the split modes they alias the 4 est-flow planes
            if self.hl8:
                ev = hb.view_of(self.est[b0:])
                return hb.SsmView(ev.ptr - 4 * 6 * ev.sc, ev.sb, ev.sc, ev.sh)
            return in16.view(b0=b0)

        def aux_view(b0):
            return hb.view_of(self.aux[b0:]) if want_aux else hb.NULL_VIEW

        if self.fuse_final:
            def synth(tf, w, b):      # final_conv + synthesis in one kernel, per group of entries that share a pair
                for p, b0, n, bc in self._groups():
                    o5 = self.s2.t["out"].view(b0=b0) if want_out5 else hb.NULL_VIEW
                    hb.check(lib.ssm_final_conv_fwd(tf.view(b0=b0), w.data_ptr(), b.data_ptr(), 5, o5, self._img6_view(p, bc),
                                                    in16_view(b0), tptr + 4 * b0, hb.view_of(self.img[b0:]), aux_view(b0), n,
                                                    self.H, self.W, st))
            self.s2.run(cross_planes=cross, cross_broadcast=cbc, synth=synth)
            return self.img
        out5 = self.s2.run(cross_planes=cross, cross_broadcast=cbc)
        if tm is not None:   # 72 B/px (read 6+4+5 ch, write 3 ch)
            e0, e1 = tm.span("warp", "synthesize", nbytes=72.0 * px)
            e0.record()
        for p, b0, n, bc in self._groups():
            hb.check(lib.ssm_synthesize_fwd(self._img6_view(p, bc), in16_view(b0), out5.view(b0=b0), tptr + 4 * b0,
                                            hb.view_of(self.img[b0:]), aux_view(b0), n, self.H, self.W, st))
        if tm is not None:
            e1.record()
        return self.img

    def run(self, img6, t, want_aux=True, want_out5=False):
        self.load_pair(img6)
        self.run_stage1()
        return self.run_stage2(t, want_aux, want_out5)

    def intermediates(self):
        """(F01, F10, Ft1^, Ft0^, Ft1, Ft0, V0) as FullModel returns them
        (scripts/models/superslomo_r.py:108-150); torch views/copies on the device."""
        flow = self.s1.t["out"].interior
        if self.G > 1:
            flow = flow.repeat_interleave(self.G, 0) if self.B1 > 1 else flow.expand(self.B2, -1, -1, -1)
        if self.hl8:
            e1, e0 = self.est[:, 0:2], self.est[:, 2:4]
        else:
            in16 = self.s2.t["in"].interior
            e1, e0 = in16[:, 6:8], in16[:, 8:10]
        return (flow[:, 0:2], flow[:, 2:4], e1, e0, self.aux[:, 0:2], self.aux[:, 2:4], self.aux[:, 4:5])


class WindowEngine:
    """N_FRAMES-1 = T interpolation windows whose U-Nets are coupled by a recurrent bottleneck
    (BASELINE config 4; scripts/models/superslomo_r.py:152-293 with BOTTLENECK=CLSTM|CGRU).

    Stage 1 runs on all T windows of S1 clips (its flows feed every window's stage-2 input); stage 2 encodes all
    T windows of S2 sequences, runs its own recurrent bottleneck over them and decodes ONLY the middle window
    unless decode_all (the reference decodes all T and returns the middle one - same frame, 1/T of the decoder
    work).  Either S2 == S1 (FullModel.forward: one t per clip and window) or S1 == 1 and S2 = number of
    intermediate times of that clip (stage 1 hoisted out of the t loop; its tensors are batch-broadcast)."""

    def __init__(self, sd1, sd2, T, S1, S2, H, W, device, cross_skip=True, mode="f16f8", bottleneck="CLSTM",
                 decode_all=False):
        assert S2 == S1 or S1 == 1, "stage-2 sequences must equal stage-1 clips, or there must be one clip"
        self.T, self.S1, self.S2, self.H, self.W, self.device = T, S1, S2, H, W, device
        self.mid = T // 2
        full_mode, mode = mode, base_mode(mode)
        self.cross, self.mode, self.hl8, self.q8 = bool(cross_skip), mode, mode != "f32", mode == "f16f8"
        self.bcast = S1 == 1 and S2 > 1
        self.decode_all = decode_all
        b1, b2 = bottleneck if isinstance(bottleneck, (tuple, list)) else (bottleneck, bottleneck)
        self.s1 = UNetPlan(1, sd1, T * S1, H, W, device, cross_skip, full_mode, True, b1, seq_len=T)
        self.s2 = UNetPlan(2, sd2, T * S2, H, W, device, cross_skip, full_mode, True, b2, seq_len=T,
                           dec=None if decode_all else (self.mid * S2, S2))
        nd = T * S2 if decode_all else S2
        self.t_dev = torch.empty(T * S2, dtype=torch.float32, device=device)
        self.img = torch.empty(nd, 3, H, W, dtype=torch.float32, device=device)
        self.aux = torch.empty(nd, 5, H, W, dtype=torch.float32, device=device)
        self.est = torch.empty(T * S2, 4, H, W, dtype=torch.float32, device=device) if self.hl8 else None
        self.img6 = None

    def load_frames(self, frames):
        """frames [S1, T+1, 3, H, W] -> the T*S1 window pairs, time-major."""
        assert tuple(frames.shape) == (self.S1, self.T + 1, 3, self.H, self.W), "frame tensor has shape %s" % (tuple(frames.shape),)
        pairs = torch.cat([frames[:, :-1], frames[:, 1:]], dim=2)                 # [S1, T, 6, H, W]
        self.img6 = pairs.transpose(0, 1).reshape(self.T * self.S1, 6, self.H, self.W).contiguous()
        self.s1.t["in"].load(self.img6)

    def _img6_view(self, k):
        v = hb.view_of(self.img6[k * self.S1:(k + 1) * self.S1])
        if self.bcast:
            v.sb = 0
        return v

    def _est_view(self, b0):
        ev = hb.view_of(self.est[b0:])
        return hb.SsmView(ev.ptr - 4 * 6 * ev.sc, ev.sb, ev.sc, ev.sh)      # channels 6..9 alias the 4 est-flow planes

    def run(self, frames, t, want_aux=True):
        """t: [S2, T] interpolation times (per sequence and window).  Returns [S2,3,H,W] (middle window) or, with
        decode_all, [T*S2,3,H,W] in time-major order."""
        lib, st = hb.load(), hb.stream_ptr()
        T, S1, S2 = self.T, self.S1, self.S2
        self.load_frames(frames)
        self.s1.run()
        self.t_dev.copy_(t.to(torch.float32).reshape(S2, T).t().reshape(-1), non_blocking=True)
        flow4, in16 = self.s1.t["out"], self.s2.t["in"]
        tptr = self.t_dev.data_ptr()
        for k in range(T):
            fv = flow4.view(broadcast=self.bcast, b0=k * S1)
            if self.hl8:
                fn = lib.ssm_flowinterp_inputs_hq8_fwd if self.q8 else lib.ssm_flowinterp_inputs_hl8_fwd
                hb.check(fn(self._img6_view(k), fv, tptr + 4 * k * S2, in16.view(b0=k * S2),
                            hb.view_of(self.est[k * S2:]), S2, self.H, self.W, st))
            else:
                hb.check(lib.ssm_flowinterp_inputs_fwd(self._img6_view(k), fv, tptr + 4 * k * S2, in16.view(b0=k * S2),
                                                       S2, self.H, self.W, st))
        if self.decode_all:
            assert not self.bcast
            out5 = self.s2.run(cross_planes=self.s1.t["c6"] if self.cross else None)
            windows = range(T)
        else:
            out5 = self.s2.run(cross_planes=self.s1.t["c6"] if self.cross else None, cross_broadcast=self.bcast,
                               cross_b0=self.mid * S1)
            windows = [self.mid]
        for j, k in enumerate(windows):
            o0 = k * S2 if self.decode_all else 0
            in16_view = self._est_view(k * S2) if self.hl8 else in16.view(b0=k * S2)
            hb.check(lib.ssm_synthesize_fwd(self._img6_view(k), in16_view, out5.view(b0=o0), tptr + 4 * k * S2,
                                            hb.view_of(self.img[o0:]), hb.view_of(self.aux[o0:]) if want_aux else hb.NULL_VIEW,
                                            S2, self.H, self.W, st))
        return self.img

    def intermediates(self, k=None):
        """(F01, F10, Ft1^, Ft0^, Ft1, Ft0, V0) of window k (default: the middle one)."""
        k = self.mid if k is None else k
        S1, S2 = self.S1, self.S2
        flow = self.s1.t["out"].interior[k * S1:(k + 1) * S1]
        if self.bcast:
            flow = flow.expand(S2, -1, -1, -1)
        if self.hl8:
            e = self.est[k * S2:(k + 1) * S2]
            e1, e0 = e[:, 0:2], e[:, 2:4]
        else:
            in16 = self.s2.t["in"].interior[k * S2:(k + 1) * S2]
            e1, e0 = in16[:, 6:8], in16[:, 8:10]
        o0 = k * S2 if self.decode_all else 0
        aux = self.aux[o0:o0 + S2]
        return (flow[:, 0:2], flow[:, 2:4], e1, e0, aux[:, 0:2], aux[:, 2:4], aux[:, 4:5])


class PairPipeline:
    """Throughput mode: N PairEngines on N HIP streams, frame pairs dealt round-robin.

    Pairs are independent, so while one stream is in an HBM-bound kernel (concat+upsample, the gather
    kernels) or an under-filled one (stage 1 at batch 1 on the 1/16 and 1/32 maps) the other stream's
    MFMA-bound convolutions use the idle matrix cores / CUs.  Each engine owns its activations; the input
    pair is read in place.  Results of `submit` stay valid until that slot is reused (N pairs later)."""

    def __init__(self, sd1, sd2, n_t, H, W, device, cross_skip=True, mode="f32", n_streams=2, graphs=False, pairs_per_batch=1):
        """pairs_per_batch = P: every submit() takes P pairs [P,6,H,W] through one PairEngine pass (stage 1 at batch P,
        stage 2 at batch P * n_t) and returns [P * n_t,3,H,W], pair-major."""
        self.P = P = int(pairs_per_batch)
        self.engines = [PairEngine(sd1, sd2, P, P * n_t, H, W, device, cross_skip, mode) for _ in range(n_streams)]
        self.streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
        self.n = n_streams
        self._i = 0
        # graphs: each slot's ~60 launches of a pair are captured once into a HIP graph (hipGraph via torch.cuda.CUDAGraph,
        # the C-ABI launches land on the capturing stream) and replayed from static input buffers
        self.graphs = bool(graphs)
        self._g = [dict() for _ in range(n_streams)]
        if self.graphs:
            self._in = [torch.empty(P, 6, H, W, dtype=torch.float32, device=device) for _ in range(n_streams)]
            self._t = [torch.empty(n_t, dtype=torch.float32, device=device) for _ in range(n_streams)]

    def _graph(self, k, want_aux):
        g = self._g[k].get(want_aux)
        if g is None:
            eng = self.engines[k]
            eng.run(self._in[k], self._t[k], want_aux)          # warm-up on this stream: attribute calls, lazy state
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
                eng.run(self._in[k], self._t[k], want_aux)
            self._g[k][want_aux] = g
        return g

    def submit(self, img6, t, want_aux=False, clone=False):
        """Queue one batch of pairs [P,6,H,W] with the t vector [n_t] on the next stream.  Returns the [P*n_t,3,H,W] frames:
        the slot's own output buffer (valid until the slot is reused, N pairs later) or, with clone=True, a
        private copy made on the slot's stream.  Filled asynchronously: call sync() before reading."""
        k = self._i % self.n
        self._i += 1
        st = self.streams[k]
        caller = torch.cuda.current_stream()
        st.wait_stream(caller)        # inputs produced on the caller's stream
        img6.record_stream(st)        # ... and read on this one: keep the allocator from recycling them early
        with torch.cuda.stream(st):
            if self.graphs:
                self._in[k].copy_(img6.reshape(self._in[k].shape), non_blocking=True)
                self._t[k].copy_(t.reshape(-1), non_blocking=True)
                self._graph(k, bool(want_aux)).replay()
                out = self.engines[k].img
            else:
                out = self.engines[k].run(img6, t, want_aux)
            if clone:
                out = out.clone()
                out.record_stream(caller)
        return out

    def sync(self):
        for st in self.streams:
            torch.cuda.current_stream().wait_stream(st)
