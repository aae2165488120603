"""Operator primitives of the hot path, MI355X-native.

Same names, arguments and state-dict keys as the reference's
scripts/models/layers.py (conv :21-33, avg_pool :60-63, warp :73-120); the
arithmetic runs in hand-written HIP kernels behind the C ABI of
include/ssm_hip.h.  No CPU path: CPU tensors raise.
"""
import logging

import torch
import torch.nn as nn

from ssm_amd import hipbind as hb

log = logging.getLogger(__name__)


def _packed_for(conv_mod, B, H, W):
    """Repacked filter of an nn.Conv2d, cached on the module and refreshed when the
    parameters are replaced or written in place (load_state_dict, optimizer step) or the
    problem size selects another tile configuration."""
    w, b = conv_mod.weight, conv_mod.bias
    stamp = (w.data_ptr(), w._version, b.data_ptr(), b._version,
             hb.conv_plan(w.shape[2], w.shape[1], w.shape[0], B, H, W, False))
    cache = conv_mod.__dict__.get("_ssm_packed")
    if cache is None or cache[0] != stamp:
        cache = (stamp, hb.PackedConv(w, b, B, H, W, False))
        conv_mod.__dict__["_ssm_packed"] = cache
    return cache[1]


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class _ConvFn(torch.autograd.Function):
    """Op-level autograd of conv (+ LeakyReLU): data gradient = the forward kernel on the transposed, flipped filter,
    weight gradient = ssm_conv2d_wgrad, bias gradient = ssm_bias_grad (what torch's autograd does for the reference's
    nn.Conv2d / LeakyReLU pair).  The planned training step (ssm_amd.backward) is the fast path; this one makes every
    composition of the public operators differentiable."""

    @staticmethod
    def forward(ctx, x, weight, bias, conv_mod, lrelu, slope):
        y, src = _conv_run(conv_mod, x.detach(), lrelu, slope)
        ctx.src, ctx.lrelu, ctx.slope, ctx.mod = src, lrelu, slope, conv_mod
        ctx.save_for_backward(y, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        from ssm_amd.backward import transposed_filter
        y, weight = ctx.saved_tensors
        lib, st = hb.load(), hb.stream_ptr()
        B, co, H, W = y.shape
        ci, k = weight.shape[1], weight.shape[2]
        dy = dy.contiguous()
        pk_t = hb.PackedConv(transposed_filter(weight), torch.zeros(ci, device=y.device), B, H, W)
        dz = hb.Planes(B, pk_t.cin_p, H, W, y.device)
        hb.check(lib.ssm_lrelu_bwd(hb.view_of(dy), hb.NULL_VIEW, hb.view_of(y), dz.view(), B, co, H, W, ctx.slope,
                                   1 if ctx.lrelu else 0, st))
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(B, ci, H, W, dtype=torch.float32, device=y.device)
            hb.conv2d(dz.view(), pk_t.cin_p, None, 0, pk_t, hb.view_of(dx), None, B, H, W, lrelu=False)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            hb.check(lib.ssm_conv2d_wgrad(ctx.src.view(), dz.view(), dw.data_ptr(), B, ci, co, H, W, k, ci, 0, 1, st))
        if ctx.needs_input_grad[2]:
            db = torch.zeros(co, dtype=torch.float32, device=y.device)
            hb.check(lib.ssm_bias_grad(dz.view(), db.data_ptr(), B, co, H, W, st))
        return dx, dw, db, None, None, None


def conv_forward(conv_mod, x, lrelu, slope=0.1):
    """y = [LeakyReLU](conv2d(x)) for a stride-1 'same' nn.Conv2d, on the MFMA kernel."""
    hb.require_device(x, "conv input")
    if _needs_grad(x, conv_mod.weight, conv_mod.bias):
        return _ConvFn.apply(x, conv_mod.weight, conv_mod.bias, conv_mod, lrelu, slope)
    return _conv_run(conv_mod, x, lrelu, slope)[0]


def _conv_run(conv_mod, x, lrelu, slope):
    B, C, H, W = x.shape
    pk = _packed_for(conv_mod, B, H, W)
    assert C == pk.cin, "conv expects %d input channels, got %d" % (pk.cin, C)
    src = hb.Planes(B, pk.cin_p, H, W, x.device)       # zero frame + channels padded to the chunk size
    lib = hb.load()
    xs = x if x.stride(3) == 1 else x.contiguous()
    hb.check(lib.ssm_copy_view(hb.view_of(xs), src.view(), B, C, H, W, hb.stream_ptr()))
    y = torch.empty(B, pk.cout, H, W, dtype=torch.float32, device=x.device)
    hb.conv2d(src.view(), pk.cin_p, None, 0, pk, hb.view_of(y), None, B, H, W, lrelu=lrelu, slope=slope)
    return y, src


class HipConv2d(nn.Conv2d):
    """nn.Conv2d (parameters, state-dict keys, init unchanged) whose forward is the HIP kernel."""

    def _check(self):
        k = self.kernel_size[0]
        if not (self.kernel_size == (k, k) and k in (3, 5, 7) and self.stride == (1, 1) and self.dilation == (1, 1)
                and self.padding == ((k - 1) // 2, (k - 1) // 2) and self.groups == 1 and self.bias is not None):
            raise NotImplementedError("HIP conv covers what the reference instantiates: k in {3,5,7}, stride 1, "
                                      "'same' zero padding, dilation 1, bias; got %s" % (self,))

    def forward(self, x):
        self._check()
        return conv_forward(self, x, lrelu=False)


class ConvLReLU(nn.Sequential):
    """Sequential(Conv2d, LeakyReLU(0.1)) - keys `0.weight`, `0.bias` - executed as ONE fused kernel."""

    def forward(self, x):
        self[0]._check()
        return conv_forward(self[0], x, lrelu=True, slope=self[1].negative_slope)


def conv(in_planes, out_planes, kernel_size=3, stride=1, padding=1, dilation=1):
    return ConvLReLU(
        HipConv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation,
                  bias=True),
        nn.LeakyReLU(0.1, inplace=True),
    )


class AvgPool2Hip(nn.Module):
    def __init__(self, kernel_size=2, stride=None, padding=0):
        super().__init__()
        if kernel_size != 2 or stride not in (None, 2) or padding != 0:
            raise NotImplementedError("only the 2x2/stride-2 mean the reference uses is built")

    def forward(self, x):
        hb.require_device(x, "avg_pool input")
        return _PoolFn.apply(x) if _needs_grad(x) else _pool_run(x)


def _pool_run(x):
    x = x.contiguous()
    B, C, H, W = x.shape
    y = torch.empty(B, C, H // 2, W // 2, dtype=torch.float32, device=x.device)
    hb.check(hb.load().ssm_avgpool2_fwd(hb.view_of(x), hb.view_of(y), B, C, H, W, hb.stream_ptr()))
    return y


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _pool_run(x.detach())

    @staticmethod
    def backward(ctx, dy):          # every input pixel of a 2x2 window receives dy / 4
        dy = dy.contiguous()
        B, C, h, w = dy.shape
        dx = torch.empty(B, C, 2 * h, 2 * w, dtype=torch.float32, device=dy.device)
        hb.check(hb.load().ssm_lrelu_bwd(hb.NULL_VIEW, hb.view_of(dy), hb.NULL_VIEW, hb.view_of(dx), B, C, 2 * h, 2 * w, 1.0, 0,
                                         hb.stream_ptr()))
        return dx


def avg_pool(kernel_size=2, stride=None, padding=0):
    return AvgPool2Hip(kernel_size, stride, padding)


def upsample2x_cat(a, b=None):
    """F.upsample(torch.cat([a, b], 1), size=(2h, 2w), mode='bilinear') in one kernel."""
    hb.require_device(a, "upsample input")
    if _needs_grad(a, b):
        return _UpCatFn.apply(a, b)
    return _upcat_run(a, b)


class _UpCatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.shapes = (tuple(a.shape), None if b is None else tuple(b.shape))
        return _upcat_run(a.detach(), None if b is None else b.detach())

    @staticmethod
    def backward(ctx, du):
        sa, sb = ctx.shapes
        du = du.contiguous()
        B, Ca, h, w = sa
        da = torch.empty(sa, dtype=torch.float32, device=du.device)
        db = torch.empty(sb, dtype=torch.float32, device=du.device) if sb is not None else None
        hb.check(hb.load().ssm_upsample2x_cat_bwd(hb.view_of(du), hb.view_of(da), Ca, hb.view_of(db) if db is not None else hb.NULL_VIEW,
                                                  sb[1] if sb is not None else 0, B, h, w, 0, 0, hb.stream_ptr()))
        return da, db


def _upcat_run(a, b=None):
    a = a.contiguous()
    B, Ca, h, w = a.shape
    Cb = 0
    if b is not None:
        hb.require_device(b, "upsample input")
        b = b.contiguous()
        Cb = b.shape[1]
        assert b.shape[0] == B and tuple(b.shape[2:]) == (h, w), "cat operands differ in shape"
    y = torch.empty(B, Ca + Cb, 2 * h, 2 * w, dtype=torch.float32, device=a.device)
    hb.check(hb.load().ssm_upsample2x_cat_fwd(hb.view_of(a), Ca, hb.view_of(b) if b is not None else hb.NULL_VIEW, Cb,
                                              hb.view_of(y), B, h, w, hb.stream_ptr()))
    return y


def warp(x, flo):
    """Backward-warp `x` [B,C,H,W] by the flow `flo` [B,2,H,W] (u,v): bilinear, zeros outside."""
    hb.require_device(x, "warp image")
    hb.require_device(flo, "warp flow")
    if _needs_grad(x, flo):
        return _WarpFn.apply(x, flo)
    return _warp_run(x, flo)


class _WarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flo):
        x, flo = x.detach().contiguous(), flo.detach().contiguous()
        ctx.save_for_backward(x, flo)
        return _warp_run(x, flo)

    @staticmethod
    def backward(ctx, dy):
        x, flo = ctx.saved_tensors
        dy = dy.contiguous()
        B, C, H, W = x.shape
        dflow = torch.empty_like(flo) if ctx.needs_input_grad[1] else None
        dimg = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        hb.check(hb.load().ssm_warp_bilinear_bwd(hb.view_of(x), hb.view_of(flo), hb.view_of(dy),
                                                 hb.view_of(dflow) if dflow is not None else hb.NULL_VIEW,
                                                 hb.view_of(dimg) if dimg is not None else hb.NULL_VIEW, B, C, H, W, hb.stream_ptr()))
        return dimg, dflow


def _warp_run(x, flo):
    B, C, H, W = x.size()
    assert tuple(flo.shape) == (B, 2, H, W), "flow must be [B,2,H,W]"
    x, flo = x.contiguous(), flo.contiguous()
    out = torch.empty_like(x)
    hb.check(hb.load().ssm_warp_bilinear_fwd(hb.view_of(x), hb.view_of(flo), hb.view_of(out), B, C, H, W,
                                             hb.stream_ptr()))
    return out
