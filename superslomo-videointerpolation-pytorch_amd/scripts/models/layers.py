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
from ssm_amd import ops as _ops  # noqa: F401  (registers the torch.library operators torch.ops.ssm.*)

log = logging.getLogger(__name__)


def conv_forward(conv_mod, x, lrelu, slope=0.1):
    """y = [LeakyReLU](conv2d(x)) for a stride-1 'same' nn.Conv2d: the dispatcher op ssm::conv2d (ssm_amd.ops) - MFMA
    kernel forward, HIP data / weight / bias gradients behind its autograd registration."""
    hb.require_device(x, "conv input")
    return torch.ops.ssm.conv2d(x, conv_mod.weight, conv_mod.bias, bool(lrelu), float(slope))


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
        return torch.ops.ssm.avg_pool2(x)


def avg_pool(kernel_size=2, stride=None, padding=0):
    return AvgPool2Hip(kernel_size, stride, padding)


def upsample2x_cat(a, b=None):
    """F.upsample(torch.cat([a, b], 1), size=(2h, 2w), mode='bilinear') in one kernel (dispatcher op ssm::upsample2x_cat)."""
    hb.require_device(a, "upsample input")
    if b is not None:
        hb.require_device(b, "upsample input")
    return torch.ops.ssm.upsample2x_cat(a, b)


def warp(x, flo):
    """Backward-warp `x` [B,C,H,W] by the flow `flo` [B,2,H,W] (u,v): bilinear, zeros outside (dispatcher op ssm::warp)."""
    hb.require_device(x, "warp image")
    hb.require_device(flo, "warp flow")
    return torch.ops.ssm.warp(x, flo)
