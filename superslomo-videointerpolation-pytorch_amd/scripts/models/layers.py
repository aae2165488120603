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
             hb.conv_config(w.shape[2], w.shape[0], B, H, W, False))
    cache = conv_mod.__dict__.get("_ssm_packed")
    if cache is None or cache[0] != stamp:
        cache = (stamp, hb.PackedConv(w, b, B, H, W, False))
        conv_mod.__dict__["_ssm_packed"] = cache
    return cache[1]


def conv_forward(conv_mod, x, lrelu, slope=0.1):
    """y = [LeakyReLU](conv2d(x)) for a stride-1 'same' nn.Conv2d, on the MFMA kernel."""
    hb.require_device(x, "conv input")
    if torch.is_grad_enabled() and (x.requires_grad or conv_mod.weight.requires_grad):
        raise NotImplementedError("the op-by-op HIP convolution carries no autograd graph: the training step runs as one "
                                  "planned forward/backward (FullModel.forward(inference_mode=False), ssm_amd.backward); "
                                  "call single operators under torch.no_grad()")
    B, C, H, W = x.shape
    pk = _packed_for(conv_mod, B, H, W)
    assert C == pk.cin, "conv expects %d input channels, got %d" % (pk.cin, C)
    src = hb.Planes(B, pk.cin_p, H, W, x.device)       # zero frame + channels padded to the chunk size
    lib = hb.load()
    xs = x if x.stride(3) == 1 else x.contiguous()
    hb.check(lib.ssm_copy_view(hb.view_of(xs), src.view(), B, C, H, W, hb.stream_ptr()))
    y = torch.empty(B, pk.cout, H, W, dtype=torch.float32, device=x.device)
    hb.conv2d(src.view(), pk.cin_p, None, 0, pk, hb.view_of(y), None, B, H, W, lrelu=lrelu, slope=slope)
    return y


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
        x = x.contiguous()
        B, C, H, W = x.shape
        y = torch.empty(B, C, H // 2, W // 2, dtype=torch.float32, device=x.device)
        hb.check(hb.load().ssm_avgpool2_fwd(hb.view_of(x), hb.view_of(y), B, C, H, W, hb.stream_ptr()))
        return y


def avg_pool(kernel_size=2, stride=None, padding=0):
    return AvgPool2Hip(kernel_size, stride, padding)


def upsample2x_cat(a, b=None):
    """F.upsample(torch.cat([a, b], 1), size=(2h, 2w), mode='bilinear') in one kernel."""
    hb.require_device(a, "upsample input")
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
    B, C, H, W = x.size()
    assert tuple(flo.shape) == (B, 2, H, W), "flow must be [B,2,H,W]"
    x, flo = x.contiguous(), flo.contiguous()
    out = torch.empty_like(x)
    hb.check(hb.load().ssm_warp_bilinear_fwd(hb.view_of(x), hb.view_of(flo), hb.view_of(out), B, C, H, W,
                                             hb.stream_ptr()))
    return out
