"""Launch plans for the hot path: frame pair -> intermediate frame(s).

A `UNetPlan` is one U-Net (stage 1 = flow computation, stage 2 = arbitrary-time
flow interpolation) at a fixed (batch, H, W) and precision mode: every activation
pre-allocated (HL8 hi/lo-fp16 planes by default, fp32 padded planes in mode f32),
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
import torch

from . import hipbind as hb
from .weights import param_key, unet_layers


class KernelTimer:
    """HIP-event bracket around kernel launches on the stream they are issued on (torch's
    current stream, which is also what the C ABI receives).  Used by bench.py for the
    roofline figures; off by default."""

    def __init__(self):
        self.spans = []      # (family, name, flops, bytes, ev0, ev1)

    def span(self, family, name, flops=0.0, nbytes=0.0):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.spans.append((family, name, flops, nbytes, e0, e1))
        return e0, e1

    def summary(self):
        """family -> dict(ms, flops, bytes, launches); call after torch.cuda.synchronize()."""
        out = {}
        for fam, name, fl, nb, e0, e1 in self.spans:
            d = out.setdefault(fam, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "by_name": {}})
            ms = e0.elapsed_time(e1)
            d["ms"] += ms
            d["flops"] += fl
            d["bytes"] += nb
            d["launches"] += 1
            n = d["by_name"].setdefault(name, [0.0, 0.0, 0])
            n[0] += ms
            n[1] += fl
            n[2] += 1
        return out


MODES = ("f32", "f16x3", "f16")
# f32   : fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 padded planes
# f16x3 : fp16 MFMA on hi/lo-split operands, 3 MFMAs per product, fp32 accumulate (fp32-grade results)
# f16   : fp16 MFMA, hi*hi only (reduced precision; 4K / config 5)
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

    def __init__(self, stage, state_dict, B, H, W, device, cross_skip=True, mode="f32", fuse_upsample=True):
        assert mode in MODES, "precision mode must be one of %s" % (MODES,)
        self.mode, self.hl8 = mode, mode != "f32"
        self.fuse_up = bool(fuse_upsample) and self.hl8      # concat+upsample fused into the consumer conv's loader
        if H % 32 or W % 32:
            raise AssertionError("H and W must be multiples of 32 (got %dx%d): the U-Net pools 5 times "
                                 "and concatenates skips (unchecked in the reference, fails in torch.cat)" % (H, W))
        self.stage, self.B, self.H, self.W, self.device = stage, B, H, W, device
        self.cross = bool(cross_skip) and stage == 2
        self.layers = {n: (ci, co, k) for n, ci, co, k in unet_layers(stage, cross_skip)}
        self.pk = {}
        for name, (ci, co, k) in self.layers.items():
            w = state_dict[param_key(name, "weight")].to(device=device, dtype=torch.float32)
            b = state_dict[param_key(name, "bias")].to(device=device, dtype=torch.float32)
            assert tuple(w.shape) == (co, ci, k, k), "%s: weight shape %s != %s" % (name, tuple(w.shape), (co, ci, k, k))
            s = layer_scale(name)
            if self.hl8:
                self.pk[name] = hb.PackedConv16(w, b, W // s)
            else:
                self.pk[name] = hb.PackedConv(w, b, B, H // s, W // s, pool=name in POOLED)
        if self.hl8:
            P = lambda c, s: hb.HPlanes(B, c, H // s, W // s, device)  # noqa: E731
        else:
            P = lambda c, s: hb.Planes(B, c, H // s, W // s, device)  # noqa: E731
        cin0 = self.layers["conv1a"][0]
        cfin = self.layers["final_conv"][1]
        t = self.t = {}
        t["in"] = hb.HPlanes(B, cin0, H, W, device, groups=self.pk["conv1a"].cin_p // 8) if self.hl8 else P(cin0, 1)
        t["t1a"], t["c1"], t["p2"] = P(32, 1), P(32, 1), P(32, 2)
        t["t2a"], t["c2"], t["p3"] = P(64, 2), P(64, 2), P(64, 4)
        t["t3a"], t["c3"], t["p4"] = P(128, 4), P(128, 4), P(128, 8)
        t["t4a"], t["c4"], t["p5"] = P(256, 8), P(256, 8), P(256, 16)
        t["t5a"], t["c5"], t["p6"] = P(512, 16), P(512, 16), P(512, 32)
        t["t6a"], t["c6"] = P(512, 32), P(512, 32)
        if not self.fuse_up:        # materialised concat+upsample tensors (the largest activations of the net)
            t["u7"] = P(1024 if self.cross else 512, 16)
            t["u8"], t["u9"], t["u10"], t["u11"] = P(1024, 8), P(512, 4), P(256, 2), P(128, 1)
        t["t7a"], t["c7"] = P(512, 16), P(512, 16)
        t["t8a"], t["c8"] = P(256, 8), P(256, 8)
        t["t9a"], t["c9"] = P(128, 4), P(128, 4)
        t["t10a"], t["c10"] = P(64, 2), P(64, 2)
        t["t11a"], t["c11"] = P(32, 1), P(32, 1)
        t["tf"] = P(32, 1)
        t["out"] = hb.Planes(B, cfin, H, W, device)     # final_conv always leaves fp32 planes (flows / logits)

    def refresh_weights(self, state_dict):
        """Repack every filter from `state_dict` (training: the parameters change each optimizer step)."""
        for name, (ci, co, k) in self.layers.items():
            w = state_dict[param_key(name, "weight")].to(device=self.device, dtype=torch.float32)
            b = state_dict[param_key(name, "bias")].to(device=self.device, dtype=torch.float32)
            s = layer_scale(name)
            if self.hl8:
                self.pk[name] = hb.PackedConv16(w, b, self.W // s)
            else:
                self.pk[name] = hb.PackedConv(w, b, self.B, self.H // s, self.W // s, pool=name in POOLED)

    def _conv(self, name, src, dst, pool=None, src2=None, lrelu=True):
        pk = self.pk[name]
        s = self.t[src]
        d = self.t[dst]
        c2 = self.t[src2].C if src2 else 0
        tm = self.timer
        if tm is not None:
            e0, e1 = tm.span("conv", "s%d.%s" % (self.stage, name), 2.0 * self.B * s.H * s.W * pk.cout * pk.cin * pk.k * pk.k)
            e0.record()
        if self.hl8:
            final = name == "final_conv"
            hb.conv2d_hl8(s.view(), s.G * 8, self.t[src2].view() if src2 else None, c2, pk, None if final else d.view(),
                          d.view() if final else None, self.t[pool].view() if pool else None, self.B, s.H, s.W,
                          lrelu=lrelu, fast=self.mode == "f16")
        else:
            hb.conv2d(s.view(), s.C, self.t[src2].view() if src2 else None, c2, pk, d.view(),
                      self.t[pool].view() if pool else None, self.B, s.H, s.W, lrelu=lrelu)
        if tm is not None:
            e1.record()

    def _up_conv(self, name, a, b, u, dst, b_planes=None, b_broadcast=False):
        """dst = conv(name)( upsample2x(cat[a, b]) ): fused in one kernel, or via the materialised tensor `u`."""
        if not self.fuse_up:
            self._up(a, b, u, b_planes=b_planes, b_broadcast=b_broadcast)
            return self._conv(name, u, dst)
        pk = self.pk[name]
        A = self.t[a]
        Bp = b_planes if b_planes is not None else (self.t[b] if b else None)
        d = self.t[dst]
        tm = self.timer
        if tm is not None:
            e0, e1 = tm.span("conv", "s%d.%s" % (self.stage, name), 2.0 * self.B * d.H * d.W * pk.cout * pk.cin * 9)
            e0.record()
        hb.conv2d_ups_hl8(A.view(), A.G * 8, Bp.view(broadcast=b_broadcast) if Bp else None, Bp.G * 8 if Bp else 0, pk,
                          d.view(), None, self.B, d.H, d.W, lrelu=True, fast=self.mode == "f16")
        if tm is not None:
            e1.record()

    def _up(self, a, b, dst, b_planes=None, b_broadcast=False):
        lib = hb.load()
        A = self.t[a]
        Bp = b_planes if b_planes is not None else (self.t[b] if b else None)
        tm = self.timer
        if tm is not None:   # algorithmic bytes: read the sources once, write 4x as many pixels
            e0, e1 = tm.span("upsample_cat", "s%d.%s" % (self.stage, dst),
                             nbytes=4.0 * self.B * A.H * A.W * (A.C + (Bp.C if Bp else 0)) * 5)
            e0.record()
        if self.hl8:
            hb.check(lib.ssm_upsample2x_cat_hl8_fwd(A.view(), A.G, Bp.view(broadcast=b_broadcast) if Bp else hb.NULL_HVIEW,
                                                    Bp.G if Bp else 0, self.t[dst].view(), self.B, A.H, A.W,
                                                    hb.stream_ptr()))
        else:
            hb.check(lib.ssm_upsample2x_cat_fwd(A.view(), A.C, Bp.view(broadcast=b_broadcast) if Bp else hb.NULL_VIEW,
                                                Bp.C if Bp else 0, self.t[dst].view(), self.B, A.H, A.W,
                                                hb.stream_ptr()))
        if tm is not None:
            e1.record()

    def run(self, cross_planes=None, cross_broadcast=False):
        """Input must already be in self.t['in'].  Returns the Planes of final_conv's output."""
        c = self._conv
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
        c("conv6.0", "p6", "t6a")
        c("conv6.1", "t6a", "c6")
        uc = self._up_conv
        if self.cross:
            if cross_planes is None:
                raise RuntimeError("stage 2 was built with CROSS_SKIP but no stage-1 encoding was given")
            uc("conv7a", "c6", None, "u7", "t7a", b_planes=cross_planes, b_broadcast=cross_broadcast)
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
        c("final_conv", "tf", "out", lrelu=False)
        return self.t["out"]


class PairEngine:
    """stage 1 (batch B1) -> compute_inputs -> stage 2 (batch B2) -> synthesis.
    Either B2 == B1 (one t per sample: FullModel.forward) or B1 == 1 and B2 = number
    of intermediates of that pair (stage-1 tensors broadcast over the t batch)."""

    def __init__(self, sd1, sd2, B1, B2, H, W, device, cross_skip=True, mode="f32", fuse_upsample=True):
        assert B2 == B1 or B1 == 1, "stage-2 batch must equal stage-1 batch, or stage-1 batch must be 1"
        self.B1, self.B2, self.H, self.W, self.device = B1, B2, H, W, device
        self.cross = bool(cross_skip)
        self.mode, self.hl8 = mode, mode != "f32"
        self.bcast = (B1 == 1 and B2 > 1)
        self.s1 = UNetPlan(1, sd1, B1, H, W, device, cross_skip, mode, fuse_upsample)
        self.s2 = UNetPlan(2, sd2, B2, H, W, device, cross_skip, mode, fuse_upsample)
        self.t_dev = torch.empty(B2, dtype=torch.float32, device=device)
        self.img = torch.empty(B2, 3, H, W, dtype=torch.float32, device=device)
        self.aux = torch.empty(B2, 5, H, W, dtype=torch.float32, device=device)
        self.img6 = None        # the caller's [B1,6,H,W] pair, read in place by the two gather kernels
        self.est = torch.empty(B2, 4, H, W, dtype=torch.float32, device=device) if self.hl8 else None   # Ft1^ | Ft0^

    def load_pair(self, img6):
        """img6: [B1,6,H,W] device tensor (I0 | I1 on the channel axis)."""
        assert tuple(img6.shape) == (self.B1, 6, self.H, self.W), "image pair tensor has shape %s" % (tuple(img6.shape),)
        self.img6 = img6.contiguous()
        self.s1.t["in"].load(self.img6)

    def run_stage1(self):
        return self.s1.run()

    def _img6_view(self):
        v = hb.view_of(self.img6)
        if self.bcast:
            v.sb = 0
        return v

    def run_stage2(self, t, want_aux=True):
        """t: [B2] device tensor of interpolation times in (0,1)."""
        lib = hb.load()
        st = hb.stream_ptr()
        self.t_dev.copy_(t.reshape(-1), non_blocking=True)
        flow4 = self.s1.t["out"]
        in16 = self.s2.t["in"]
        bc = self.bcast
        tm = UNetPlan.timer
        px = float(self.B2 * self.H * self.W)
        if tm is not None:   # 104 B/px (read 10 ch, write 16 ch)
            e0, e1 = tm.span("warp", "flowinterp_inputs", nbytes=104.0 * px)
            e0.record()
        if self.hl8:
            hb.check(lib.ssm_flowinterp_inputs_hl8_fwd(self._img6_view(), flow4.view(broadcast=bc), self.t_dev.data_ptr(),
                                                       in16.view(), hb.view_of(self.est), self.B2, self.H, self.W, st))
        else:
            hb.check(lib.ssm_flowinterp_inputs_fwd(self._img6_view(), flow4.view(broadcast=bc), self.t_dev.data_ptr(),
                                                   in16.view(), self.B2, self.H, self.W, st))
        if tm is not None:
            e1.record()
        out5 = self.s2.run(cross_planes=self.s1.t["c6"] if self.cross else None, cross_broadcast=bc)
        if self.hl8:    # the kernel reads channels 6..9 of its `in16` argument: alias them onto the 4 est-flow planes
            ev = hb.view_of(self.est)
            in16_view = hb.SsmView(ev.ptr - 4 * 6 * ev.sc, ev.sb, ev.sc, ev.sh)
        else:
            in16_view = in16.view()
        if tm is not None:   # 72 B/px (read 6+4+5 ch, write 3 ch)
            e0, e1 = tm.span("warp", "synthesize", nbytes=72.0 * px)
            e0.record()
        hb.check(lib.ssm_synthesize_fwd(self._img6_view(), in16_view, out5.view(), self.t_dev.data_ptr(),
                                        hb.view_of(self.img), hb.view_of(self.aux) if want_aux else hb.NULL_VIEW,
                                        self.B2, self.H, self.W, st))
        if tm is not None:
            e1.record()
        return self.img

    def run(self, img6, t, want_aux=True):
        self.load_pair(img6)
        self.run_stage1()
        return self.run_stage2(t, want_aux)

    def intermediates(self):
        """(F01, F10, Ft1^, Ft0^, Ft1, Ft0, V0) as FullModel returns them
        (scripts/models/superslomo_r.py:108-150); torch views/copies on the device."""
        flow = self.s1.t["out"].interior
        if self.bcast:
            flow = flow.expand(self.B2, -1, -1, -1)
        if self.hl8:
            e1, e0 = self.est[:, 0:2], self.est[:, 2:4]
        else:
            in16 = self.s2.t["in"].interior
            e1, e0 = in16[:, 6:8], in16[:, 8:10]
        return (flow[:, 0:2], flow[:, 2:4], e1, e0, self.aux[:, 0:2], self.aux[:, 2:4], self.aux[:, 4:5])


class PairPipeline:
    """Throughput mode: N PairEngines on N HIP streams, frame pairs dealt round-robin.

    Pairs are independent, so while one stream is in an HBM-bound kernel (concat+upsample, the gather
    kernels) or an under-filled one (stage 1 at batch 1 on the 1/16 and 1/32 maps) the other stream's
    MFMA-bound convolutions use the idle matrix cores / CUs.  Each engine owns its activations; the input
    pair is read in place.  Results of `submit` stay valid until that slot is reused (N pairs later)."""

    def __init__(self, sd1, sd2, n_t, H, W, device, cross_skip=True, mode="f16x3", n_streams=2):
        self.engines = [PairEngine(sd1, sd2, 1, n_t, H, W, device, cross_skip, mode) for _ in range(n_streams)]
        self.streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
        self.n = n_streams
        self._i = 0

    def submit(self, img6, t, want_aux=False, clone=False):
        """Queue one pair [1,6,H,W] with its t vector on the next stream.  Returns the [n_t,3,H,W] frames:
        the slot's own output buffer (valid until the slot is reused, N pairs later) or, with clone=True, a
        private copy made on the slot's stream.  Filled asynchronously: call sync() before reading."""
        k = self._i % self.n
        self._i += 1
        st = self.streams[k]
        st.wait_stream(torch.cuda.current_stream())        # inputs produced on the caller's stream
        with torch.cuda.stream(st):
            out = self.engines[k].run(img6, t, want_aux)
            if clone:
                out = out.clone()
                out.record_stream(torch.cuda.current_stream())
        return out

    def sync(self):
        for st in self.streams:
            torch.cuda.current_stream().wait_stream(st)
