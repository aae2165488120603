"""Perceptual-loss feature network on the HIP kernels: torchvision's vgg16.features[:23] (conv1_1 ... relu4_3), the
phi of the reference's PerceptualLoss (scripts/models/losses.py:12-41).  Forward on fp32 padded planes (3x3 convs with
fused ReLU = the MFMA conv kernel with slope 0, MaxPool2d(2,2) = ssm_maxpool2_fwd) and the gradient wrt the input
image (the network is frozen, losses.py:30-33: no weight gradients) by the same kernels on transposed filters.

Weights: the reference downloads `vgg16(pretrained=True)`; there is no network here, so the extractor takes a
torchvision-format state dict (`features.<idx>.weight|bias`, e.g. a local vgg16-397923af.pth) or, for tests and the
benchmark, `synthetic_vgg_state_dict()`.  Architecture pinned, pretrained numbers unpinned.
"""
import numpy as np
import torch

from . import hipbind as hb
import os

from .engine import conv_fn
from .backward import transposed_filter
from .weights import _hash_uniform

# torchvision vgg16 "D" configuration up to relu4_3: (index in .features, cin, cout) and 'M' = MaxPool2d(2,2)
VGG16_CONV4_3 = [(0, 3, 64), (2, 64, 64), "M", (5, 64, 128), (7, 128, 128), "M", (10, 128, 256), (12, 256, 256), (14, 256, 256), "M",
                 (17, 256, 512), (19, 512, 512), (21, 512, 512)]


def synthetic_vgg_state_dict(gain=1.0):
    """Deterministic stand-in for the pretrained VGG16 (He-uniform for ReLU: activations stay O(1) through the 10 layers)."""
    sd = {}
    for item in VGG16_CONV4_3:
        if item == "M":
            continue
        idx, cin, cout = item
        bound = gain * float(np.sqrt(6.0 / (cin * 9)))
        w = (_hash_uniform("vgg16/features.%d/w" % idx, cout * cin * 9) * 2.0 - 1.0) * bound
        b = (_hash_uniform("vgg16/features.%d/b" % idx, cout) * 2.0 - 1.0) * 0.05
        sd["features.%d.weight" % idx] = torch.from_numpy(w.astype(np.float32).reshape(cout, cin, 3, 3).copy())
        sd["features.%d.bias" % idx] = torch.from_numpy(b.astype(np.float32).copy())
    return sd


VGG_WINO4 = os.environ.get("SSM_VGG_WINO4", "1") != "0"          # measured + 1.3 % of a training step, gradient bars unchanged (profiles/r12c_*)


class VGGFeatures:
    """phi(x) for a batch [B,3,H,W] (H, W multiples of 8) and d(phi)/dx applied to a feature gradient."""

    timer = None        # a ssm_amd.engine.KernelTimer, or None

    def __init__(self, state_dict, B, H, W, device, mode="f16f8"):
        """mode f16f8: the convolutions run on the fp16 + fp8 matrix path (Q8 operands; the fp32 planes the ReLU mask, the max
        pooling and the feature loss read are written by the same launches); mode f32: exact-fp32 MFMA kernels."""
        assert H % 8 == 0 and W % 8 == 0, "VGG16 conv4_3 pools three times: H, W must be multiples of 8"
        assert mode in ("f16f8", "f32", "f32w")
        self.B, self.H, self.W, self.device = B, H, W, device
        self.q8 = mode == "f16f8"
        self.wino = mode == "f32w"        # fp32 throughout, the 3x3 layers with 8+ input channels as Winograd F(2x2,3x3) (csrc/ssm_wino.hip)
        self.steps = []          # ("conv", idx, cin, cout, src, dst) | ("pool", src, dst)
        self.t, self.pk, self.w, self.q = {}, {}, {}, {}        # q: Q8 twins of the conv inputs (mode f16f8)
        self.b, self.pk_nb = {}, {}
        h, w, c, name = H, W, 3, "x"
        for item in VGG16_CONV4_3:
            if item == "M":
                dst = "p%d" % len(self.steps)
                self.t[dst] = hb.Planes(B, c, h // 2, w // 2, device)
                if self.q8:
                    self.q[dst] = hb.HPlanes(B, c, h // 2, w // 2, device, q8=True)
                self.steps.append(("pool", name, dst))
                h, w, name = h // 2, w // 2, dst
                continue
            idx, cin, cout = item
            wt = state_dict["features.%d.weight" % idx].to(device=device, dtype=torch.float32)
            bs = state_dict["features.%d.bias" % idx].to(device=device, dtype=torch.float32)
            assert tuple(wt.shape) == (cout, cin, 3, 3), "features.%d.weight has shape %s" % (idx, tuple(wt.shape))
            self.w[idx] = wt
            self.b[idx] = bs
            self.pk[idx] = hb.PackedConv16(wt, bs, w, q8=True) if self.q8 else self._cls(cin, cout, h, w)(wt, bs, B, h, w)
            if name == "x":
                self.t["x"] = hb.Planes(B, max(8, self.pk[idx].cin_p), h, w, device)     # 3 channels padded to the largest conv chunk
                if self.q8:
                    self.q["x"] = hb.HPlanes(B, 3, h, w, device, groups=self.pk[idx].cin_p // 8, q8=True)
            dst = "a%d" % idx
            self.t[dst] = hb.Planes(B, cout, h, w, device)
            if self.q8:
                self.q[dst] = hb.HPlanes(B, cout, h, w, device, q8=True)
            self.steps.append(("conv", idx, cin, cout, name, dst))
            c, name = cout, dst
        self.out = name
        self.g, self.pk_t = {}, {}

    def _cls(self, cin, cout, h, w):
        # mode f32w ($SSM_VGG_WINO4=0: off): F(4x4,3x3) where the library's cost model prefers it over F(2x2,3x3) at this batch
        if (self.wino and VGG_WINO4 and hb.wino4_supported(cin, cout, h, w) and hb.wino4_preferred(cin, cout, max(1, self.B // 2), h, w, False)):
            return hb.PackedWino4
        return hb.PackedWino if (self.wino and hb.wino_supported(cin, cout, h, w)) else hb.PackedConv

    def _pk(self, idx, nb, src):
        """Packed filter of conv `idx` for a launch over nb batch entries: the fp32 kernel's tile plan (and with it the packing)
        depends on the batch, and forward() runs on sub-batches (prediction and target separately)."""
        if self.q8:
            return self.pk[idx]
        pk = self.pk[idx]
        plan = (hb.wino_plan(pk.cin, pk.cout, nb, src.H, src.W) if pk.algo == "wino" else
                hb.wino4_plan(pk.cin, pk.cout, nb, src.H, src.W) if pk.algo == "wino4" else hb.conv_plan(3, pk.cin_p, pk.cout, nb, src.H, src.W))
        if plan[1:] == (pk.bn, pk.ck):
            return pk
        key = (idx, nb)
        if key not in self.pk_nb:
            self.pk_nb[key] = type(pk)(self.w[idx], self.b[idx], nb, src.H, src.W)
        return self.pk_nb[key]

    def _span(self, fam, name, flops, pk=None):
        tm = VGGFeatures.timer
        if tm is None:
            return None
        algo = getattr(pk, "algo", "") if pk is not None else ""
        issued = flops * (16.0 / 36.0 if algo == "wino" else 0.25 if algo == "wino4" else 1.0)      # F(2x2,3x3) / F(4x4,3x3) layers
        e0, e1 = tm.span(fam, name, flops, issued=issued)
        e0.record()
        return e1

    def forward(self, x, b0=0):
        """x: [nb,3,H,W] device tensor, written to batch entries [b0, b0+nb) of the plan (default: the whole batch)
        -> Planes [B,512,H/8,W/8] (valid until the next forward over those entries)."""
        hb.require_device(x, "perceptual-loss input")
        nb = x.shape[0]
        assert tuple(x.shape[1:]) == (3, self.H, self.W) and 0 <= b0 and b0 + nb <= self.B, "VGG input has shape %s" % (tuple(x.shape),)
        lib, st = hb.load(), hb.stream_ptr()
        xs = x if x.stride(3) == 1 else x.contiguous()
        hb.check(lib.ssm_copy_view(hb.view_of(xs), self.t["x"].view(b0=b0), nb, 3, self.H, self.W, st))
        if self.q8:
            q = self.q["x"]
            hb.check(lib.ssm_hq8_from_f32(hb.view_of(xs), q.view(b0=b0), nb, 3, q.G, self.H, self.W, st))
        for s in self.steps:
            if s[0] == "pool":
                src, dst = self.t[s[1]], self.t[s[2]]
                hb.check(lib.ssm_maxpool2_fwd(src.view(b0=b0), dst.view(b0=b0), nb, src.C, src.H, src.W, st))
                if self.q8:          # the next convolution reads the pooled tensor in the Q8 form
                    q = self.q[s[2]]
                    hb.check(lib.ssm_hq8_from_f32(dst.view(b0=b0), q.view(b0=b0), nb, dst.C, q.G, dst.H, dst.W, st))
                continue
            _, idx, cin, cout, sname, dname = s
            src, dst, pk = self.t[sname], self.t[dname], self._pk(idx, nb, self.t[sname])
            e1 = self._span("vgg_fwd", "features.%d" % idx, 2.0 * nb * src.H * src.W * cin * cout * 9, pk)
            if self.q8:
                hb.conv2d_hl8(self.q[sname].view(b0=b0), pk.cin_p, None, 0, pk, self.q[dname].view(b0=b0), dst.view(b0=b0), None, nb,
                              src.H, src.W, lrelu=True, slope=0.0)
            else:
                conv_fn(pk)(src.view(b0=b0), pk.cin_p, None, 0, pk, dst.view(b0=b0), None, nb, src.H, src.W, lrelu=True, slope=0.0)
            if e1 is not None:
                e1.record()
        return self.t[self.out]

    def _G(self, name, C=None):
        if name not in self.g:
            ref = self.t[name]
            self.g[name] = hb.Planes(self.B, C or ref.C, ref.H, ref.W, self.device)
        return self.g[name]

    def input_grad(self, dphi, nb=None):
        """dphi: Planes gradient wrt phi (first nb batch entries; default all) -> Planes [B,3,H,W] holding the gradient
        wrt the input image in its first nb entries."""
        lib, st = hb.load(), hb.stream_ptr()
        nb = self.B if nb is None else nb
        dy = dphi
        for s in reversed(self.steps):
            if s[0] == "pool":
                src = self.t[s[1]]
                dx = self._G(s[1])
                hb.check(lib.ssm_maxpool2_bwd(src.view(), dy.view(), dx.view(), nb, src.C, src.H, src.W, st))
                dy = dx
                continue
            _, idx, cin, cout, sname, dname = s
            Y = self.t[dname]
            if idx not in self.pk_t:
                wt_t, zb = transposed_filter(self.w[idx]), torch.zeros(cin, device=self.device)
                self.pk_t[idx] = (nb, hb.PackedConv16(wt_t, zb, Y.W, q8=True) if self.q8 else self._cls(cout, cin, Y.H, Y.W)(wt_t, zb, nb, Y.H, Y.W))
            assert self.pk_t[idx][0] == nb, "input_grad was planned for %d batch entries" % self.pk_t[idx][0]
            pk = self.pk_t[idx][1]
            key = "dz%d" % idx
            if key not in self.g:
                self.g[key] = hb.Planes(self.B, pk.cin_p, Y.H, Y.W, self.device)
                if self.q8:
                    self.g[key + "q"] = hb.HPlanes(self.B, pk.cin_p, Y.H, Y.W, self.device, q8=True)
            dzp = self.g[key]
            dx = self._G(sname, C=cin)
            e1 = self._span("vgg_bwd", "features.%d" % idx, 2.0 * nb * Y.H * Y.W * cin * cout * 9, pk)
            if self.q8:
                dzq = self.g[key + "q"]
                hb.check(lib.ssm_lrelu_bwd_q8(dy.view(), hb.NULL_VIEW, Y.view(), dzp.view(), dzq.view(), nb, cout, Y.H, Y.W, 0.0, 1, st))
                hb.conv2d_hl8(dzq.view(), pk.cin_p, None, 0, pk, None, dx.view(), None, nb, Y.H, Y.W, lrelu=False)
            else:
                hb.check(lib.ssm_lrelu_bwd(dy.view(), hb.NULL_VIEW, Y.view(), dzp.view(), nb, cout, Y.H, Y.W, 0.0, 1, st))
                conv_fn(pk)(dzp.view(), pk.cin_p, None, 0, pk, dx.view(), None, nb, Y.H, Y.W, lrelu=False)
            if e1 is not None:
                e1.record()
            dy = dx
        return dy


class PerceptualTerm:
    """mean_{C,h,w} (phi(pred) - phi(target))^2 per sample and its gradient wrt pred (losses.py:218,227:
    MSELoss(reduce=False), then the per-sample mean).  pred and target go through ONE VGG pass as a batch of 2B;
    the backward walks the first B entries only."""

    def __init__(self, vgg_state_dict, B, H, W, device, mode="f16f8"):
        self.B = B
        self.vgg = VGGFeatures(vgg_state_dict, 2 * B, H, W, device, mode)
        self.coef = torch.empty(B, dtype=torch.float32, device=device)
        self.both = torch.empty(2 * B, 3, H, W, dtype=torch.float32, device=device)
        self._side, self._pending = None, None
        self._sq_scratch = torch.empty(64 * B, dtype=torch.float32, device=device)          # slice sums of ssm_sqdiff_mean
        self._mse = torch.empty(B, dtype=torch.float32, device=device)

    def begin_target(self, target):
        """Start the target's half of the VGG pass on a second stream (the training step calls this before the U-Net forward: the
        target is known then, and the batch-2 U-Net leaves the GPU room).  forward(pred, target) with the SAME tensor then runs the
        predicted frames only and joins."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=target.device)
        hb.stream_wait(torch.cuda.current_stream(), self._side)      # the last step's readers of these buffers are queued before this
        with torch.cuda.stream(self._side):
            self.vgg.forward(target, b0=self.B)
        self._pending = target

    def forward(self, pred, target):
        """-> [B] unweighted per-sample feature MSE."""
        B = self.B
        if self._pending is not None and self._pending is target:
            self.vgg.forward(pred, b0=0)
            hb.stream_wait(self._side, torch.cuda.current_stream())
        else:
            hb.host_op(lambda: (self.both[:B].copy_(pred), self.both[B:].copy_(target)))
            self.vgg.forward(self.both)
        self._pending = None
        phi = self.vgg.t[self.vgg.out]
        out = self._mse          # (a buffer of the term: the launch below may be part of a recorded program)
        hb.check(hb.load().ssm_sqdiff_mean(phi.view(), phi.view(b0=B), self._sq_scratch.data_ptr(), out.data_ptr(), B, phi.C, phi.H, phi.W,
                                           hb.stream_ptr()))
        return out

    def grad_pred(self, weight):
        """weight [B] = (upstream gradient x lambda_p) per sample -> Planes [2B,3,H,W] whose first B entries hold
        d(loss)/d(pred)."""
        B, phi = self.B, self.vgg.t[self.vgg.out]
        n = float(phi.C * phi.H * phi.W)
        hb.host_op(lambda: self.coef.copy_(weight * (2.0 / n)))
        dphi = self.vgg._G(self.vgg.out)
        hb.check(hb.load().ssm_sqdiff_grad(phi.view(), phi.view(b0=B), self.coef.data_ptr(), dphi.view(), B, phi.C, phi.H, phi.W,
                                           hb.stream_ptr()))
        return self.vgg.input_grad(dphi, nb=B)
