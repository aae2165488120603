"""PyTorch-ROCm custom operators (torch.library, namespace `ssm`) over the C ABI of libssm_hip.so.

north_star: "host Python calling hand-written HIP through PyTorch-ROCm custom ops over a thin C-ABI".  Every operator of
the reference's scripts/models surface that is a tensor function is registered with the dispatcher - schema, a fake
(meta) implementation for shape inference, and an autograd formula whose backward is again HIP kernels - so
`torch.ops.ssm.*` is what `models.layers` / `models.flow_interpolation` call:

    ssm::conv2d            layers.conv = Conv2d (+ LeakyReLU)          scripts/models/layers.py:21-33
    ssm::avg_pool2         layers.avg_pool                              scripts/models/layers.py:60-63
    ssm::upsample2x_cat    F.upsample(torch.cat([a, b], 1), bilinear)   scripts/models/flow_computation.py:92-94,244-245
    ssm::warp              layers.warp                                  scripts/models/layers.py:73-120
    ssm::flowinterp_inputs FlowInterpolationModel.compute_inputs        scripts/models/flow_interpolation.py:338-372
    ssm::synthesize        extract_outputs + compute_output_image       scripts/models/flow_interpolation.py:374-429

CUDA (= HIP) only: there is no CPU kernel, a CPU tensor raises.  The planned whole-path engine (ssm_amd.engine) calls the
same C ABI directly on pre-allocated plans; these operators are the drop-in, composable form.
"""
import weakref
from collections import OrderedDict
from typing import Optional

import torch
from torch import Tensor
from torch.library import custom_op

from . import hipbind as hb

_PACKS = OrderedDict()      # (weight/bias identity + version, problem) -> (weakrefs of the source tensors, PackedConv); small LRU
_PACKS_MAX = 256


def _cached_pack(key, sources, build):
    """LRU lookup that ties an entry to the LIFETIME of the tensors it was packed from: an address + version key alone can be
    re-issued by the caching allocator to the parameters of a later model (same shapes, same construction sequence), and the entry
    holds only the packed copy.  A hit therefore needs every source tensor object of the entry to be the one passed in now."""
    ent = _PACKS.get(key)
    if ent is not None and all(r() is t for r, t in zip(ent[0], sources)):
        _PACKS.move_to_end(key)
        return ent[1]
    pk = build()
    _PACKS[key] = (tuple(weakref.ref(t) for t in sources), pk)
    _PACKS.move_to_end(key)
    if len(_PACKS) > _PACKS_MAX:
        _PACKS.popitem(last=False)
    return pk


def packed_filter(weight, bias, B, H, W):
    """Repacked filter for this problem, refreshed when the parameters are replaced or written in place."""
    key = ("fwd", weight.data_ptr(), weight._version, bias.data_ptr(), bias._version, tuple(weight.shape), B, H, W)
    return _cached_pack(key, (weight, bias), lambda: hb.PackedConv(weight.detach(), bias.detach(), B, H, W, False))


def packed_filter_transposed(weight, B, H, W):
    """The data-gradient filter W'[ci][co][ky][kx] = W[co][ci][k-1-ky][k-1-kx] (zero bias), cached like the forward one."""
    from .backward import transposed_filter
    key = ("bwd", weight.data_ptr(), weight._version, tuple(weight.shape), B, H, W)
    return _cached_pack(key, (weight,), lambda: hb.PackedConv(transposed_filter(weight.detach()),
                                                             torch.zeros(weight.shape[1], device=weight.device), B, H, W))


def _padded_input(x, cin_p):
    B, C, H, W = x.shape
    src = hb.Planes(B, cin_p, H, W, x.device)       # zero frame + channels padded to the chunk size
    xs = x if x.stride(3) == 1 else x.contiguous()
    hb.check(hb.load().ssm_copy_view(hb.view_of(xs), src.view(), B, C, H, W, hb.stream_ptr()))
    return src


# ---- conv2d ----------------------------------------------------------------------------------------------------------
@custom_op("ssm::conv2d", mutates_args=(), device_types="cuda")
def conv2d(x: Tensor, weight: Tensor, bias: Tensor, lrelu: bool, slope: float) -> Tensor:
    hb.require_device(x, "conv input")
    B, C, H, W = x.shape
    pk = packed_filter(weight, bias, B, H, W)
    assert C == pk.cin, "conv expects %d input channels, got %d" % (pk.cin, C)
    src = _padded_input(x, pk.cin_p)
    y = torch.empty(B, pk.cout, H, W, dtype=torch.float32, device=x.device)
    hb.conv2d(src.view(), pk.cin_p, None, 0, pk, hb.view_of(y), None, B, H, W, lrelu=lrelu, slope=slope)
    return y


@conv2d.register_fake
def _(x, weight, bias, lrelu, slope):
    return x.new_empty(x.shape[0], weight.shape[0], x.shape[2], x.shape[3])


def _conv_setup(ctx, inputs, output):
    x, weight, bias, lrelu, slope = inputs
    ctx.save_for_backward(x, weight, output)
    ctx.lrelu, ctx.slope = lrelu, slope
    ctx.cin_p = packed_filter(weight, bias, x.shape[0], x.shape[2], x.shape[3]).cin_p      # a cache hit: the forward just packed it


def _conv_backward(ctx, dy):
    """Data gradient = the forward kernel on the transposed, flipped filter, weight gradient = ssm_conv2d_wgrad, bias
    gradient = ssm_bias_grad (what torch's autograd does for the reference's nn.Conv2d / LeakyReLU pair)."""
    x, weight, y = ctx.saved_tensors
    lib, st = hb.load(), hb.stream_ptr()
    B, co, H, W = y.shape
    ci, k = weight.shape[1], weight.shape[2]
    dy = dy.contiguous()
    pk_t = packed_filter_transposed(weight, B, H, W)
    dz = hb.Planes(B, pk_t.cin_p, H, W, y.device)
    hb.check(lib.ssm_lrelu_bwd(hb.view_of(dy), hb.NULL_VIEW, hb.view_of(y), dz.view(), B, co, H, W, ctx.slope,
                               1 if ctx.lrelu else 0, st))
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        dx = torch.empty(B, ci, H, W, dtype=torch.float32, device=y.device)
        hb.conv2d(dz.view(), pk_t.cin_p, None, 0, pk_t, hb.view_of(dx), None, B, H, W, lrelu=False)
    if ctx.needs_input_grad[1]:
        src = _padded_input(x.detach(), ctx.cin_p)
        dw = torch.empty_like(weight)
        hb.check(lib.ssm_conv2d_wgrad(src.view(), dz.view(), dw.data_ptr(), B, ci, co, H, W, k, ci, 0, 1, st))
    if ctx.needs_input_grad[2]:
        db = torch.zeros(co, dtype=torch.float32, device=y.device)
        hb.check(lib.ssm_bias_grad(dz.view(), db.data_ptr(), B, co, H, W, st))
    return dx, dw, db, None, None


conv2d.register_autograd(_conv_backward, setup_context=_conv_setup)


# ---- avg_pool2 ---------------------------------------------------------------------------------------------------------
@custom_op("ssm::avg_pool2", mutates_args=(), device_types="cuda")
def avg_pool2(x: Tensor) -> Tensor:
    hb.require_device(x, "avg_pool input")
    x = x.contiguous()
    B, C, H, W = x.shape
    y = torch.empty(B, C, H // 2, W // 2, dtype=torch.float32, device=x.device)
    hb.check(hb.load().ssm_avgpool2_fwd(hb.view_of(x), hb.view_of(y), B, C, H, W, hb.stream_ptr()))
    return y


@avg_pool2.register_fake
def _(x):
    return x.new_empty(x.shape[0], x.shape[1], x.shape[2] // 2, x.shape[3] // 2)


def _pool_backward(ctx, dy):          # every input pixel of a 2x2 window receives dy / 4
    dy = dy.contiguous()
    B, C, h, w = dy.shape
    dx = torch.empty(B, C, 2 * h, 2 * w, dtype=torch.float32, device=dy.device)
    hb.check(hb.load().ssm_lrelu_bwd(hb.NULL_VIEW, hb.view_of(dy), hb.NULL_VIEW, hb.view_of(dx), B, C, 2 * h, 2 * w, 1.0, 0,
                                     hb.stream_ptr()))
    return dx


avg_pool2.register_autograd(_pool_backward)


# ---- upsample2x_cat ------------------------------------------------------------------------------------------------------
@custom_op("ssm::upsample2x_cat", mutates_args=(), device_types="cuda")
def upsample2x_cat(a: Tensor, b: Optional[Tensor]) -> Tensor:
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


@upsample2x_cat.register_fake
def _(a, b):
    return a.new_empty(a.shape[0], a.shape[1] + (b.shape[1] if b is not None else 0), 2 * a.shape[2], 2 * a.shape[3])


def _upcat_setup(ctx, inputs, output):
    a, b = inputs
    ctx.shapes = (tuple(a.shape), None if b is None else tuple(b.shape))


def _upcat_backward(ctx, du):
    sa, sb = ctx.shapes
    du = du.contiguous()
    B, Ca, h, w = sa
    da = torch.empty(sa, dtype=torch.float32, device=du.device)
    db = torch.empty(sb, dtype=torch.float32, device=du.device) if sb is not None else None
    hb.check(hb.load().ssm_upsample2x_cat_bwd(hb.view_of(du), hb.view_of(da), Ca, hb.view_of(db) if db is not None else hb.NULL_VIEW,
                                              sb[1] if sb is not None else 0, B, h, w, 0, 0, hb.stream_ptr()))
    return da, db


upsample2x_cat.register_autograd(_upcat_backward, setup_context=_upcat_setup)


# ---- warp ------------------------------------------------------------------------------------------------------------------
@custom_op("ssm::warp", mutates_args=(), device_types="cuda")
def warp(x: Tensor, flo: Tensor) -> Tensor:
    hb.require_device(x, "warp image")
    hb.require_device(flo, "warp flow")
    B, C, H, W = x.size()
    assert tuple(flo.shape) == (B, 2, H, W), "flow must be [B,2,H,W]"
    x, flo = x.contiguous(), flo.contiguous()
    out = torch.empty_like(x)
    hb.check(hb.load().ssm_warp_bilinear_fwd(hb.view_of(x), hb.view_of(flo), hb.view_of(out), B, C, H, W, hb.stream_ptr()))
    return out


@warp.register_fake
def _(x, flo):
    return torch.empty_like(x)


def _warp_setup(ctx, inputs, output):
    x, flo = inputs
    ctx.save_for_backward(x, flo)


def _warp_backward(ctx, dy):
    x, flo = ctx.saved_tensors
    x, flo, dy = x.contiguous(), flo.contiguous(), dy.contiguous()
    B, C, H, W = x.shape
    dflow = torch.empty_like(flo) if ctx.needs_input_grad[1] else None
    dimg = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
    hb.check(hb.load().ssm_warp_bilinear_bwd(hb.view_of(x), hb.view_of(flo), hb.view_of(dy),
                                             hb.view_of(dflow) if dflow is not None else hb.NULL_VIEW,
                                             hb.view_of(dimg) if dimg is not None else hb.NULL_VIEW, B, C, H, W, hb.stream_ptr()))
    return dimg, dflow


warp.register_autograd(_warp_backward, setup_context=_warp_setup)


# ---- flowinterp_inputs -----------------------------------------------------------------------------------------------------
@custom_op("ssm::flowinterp_inputs", mutates_args=(), device_types="cuda")
def flowinterp_inputs(img6: Tensor, flow4: Tensor, t: Tensor) -> Tensor:
    """[B,6,H,W] images, [B,4,H,W] stage-1 flows, t [B] in (0,1) -> [B,16,H,W] = cat[I1, g(I1,Ft1^), Ft1^, Ft0^, g(I0,Ft0^), I0]."""
    hb.require_device(img6, "image pair")
    hb.require_device(flow4, "flow tensor")
    img, flow, tv = img6.contiguous(), flow4.contiguous(), t.contiguous()
    B, _, H, W = img.shape
    out = torch.empty(B, 16, H, W, dtype=torch.float32, device=img.device)
    hb.check(hb.load().ssm_flowinterp_inputs_fwd(hb.view_of(img), hb.view_of(flow), tv.data_ptr(), hb.view_of(out), B, H, W,
                                                 hb.stream_ptr()))
    return out


@flowinterp_inputs.register_fake
def _(img6, flow4, t):
    return img6.new_empty(img6.shape[0], 16, img6.shape[2], img6.shape[3])


def _inputs_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _inputs_backward(ctx, d16):
    """Adjoint wrt the stage-1 flows (the images are data: no gradient)."""
    img, flow, tv = ctx.saved_tensors
    img, flow, d16 = img.contiguous(), flow.contiguous(), d16.contiguous()
    B, _, H, W = img.shape
    zero4 = torch.zeros(B, 4, H, W, dtype=torch.float32, device=img.device)
    zb = torch.zeros(B, dtype=torch.float32, device=img.device)
    dflow = torch.empty_like(flow)
    hb.check(hb.load().ssm_flowinterp_inputs_bwd(hb.view_of(img), hb.view_of(flow), hb.view_of(d16), hb.view_of(zero4),
                                                 tv.data_ptr(), zb.data_ptr(), hb.view_of(dflow), B, H, W, 0, hb.stream_ptr()))
    return None, dflow, None


flowinterp_inputs.register_autograd(_inputs_backward, setup_context=_inputs_setup)


# ---- synthesize --------------------------------------------------------------------------------------------------------------
@custom_op("ssm::synthesize", mutates_args=(), device_types="cuda")
def synthesize(img6: Tensor, in16: Tensor, out5: Tensor, t: Tensor) -> Tensor:
    """Visibility-weighted blend of the two frames warped by the refined flows -> [B,3,H,W]."""
    for x, n in ((img6, "image pair"), (in16, "stage-2 input"), (out5, "stage-2 output")):
        hb.require_device(x, n)
    img, xin, xout, tv = img6.contiguous(), in16.contiguous(), out5.contiguous(), t.contiguous()
    B, _, H, W = img.shape
    y = torch.empty(B, 3, H, W, dtype=torch.float32, device=img.device)
    hb.check(hb.load().ssm_synthesize_fwd(hb.view_of(img), hb.view_of(xin), hb.view_of(xout), tv.data_ptr(), hb.view_of(y),
                                          hb.NULL_VIEW, B, H, W, hb.stream_ptr()))
    return y


@synthesize.register_fake
def _(img6, in16, out5, t):
    return img6.new_empty(img6.shape[0], 3, img6.shape[2], img6.shape[3])


def _synth_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _synth_backward(ctx, dy):
    """Adjoint wrt the stage-2 output (5 ch) and the approximated flows (channels 6:10 of the 16-channel input)."""
    img, xin, xout, tv = ctx.saved_tensors
    img, xin, xout, dy = img.contiguous(), xin.contiguous(), xout.contiguous(), dy.contiguous()
    B, _, H, W = img.shape
    zb = torch.zeros(B, dtype=torch.float32, device=img.device)
    est = xin[:, 6:10].contiguous()
    dout5 = torch.empty_like(xout)
    dxin = torch.zeros_like(xin)
    dest = torch.empty(B, 4, H, W, dtype=torch.float32, device=img.device)
    hb.check(hb.load().ssm_synthesize_bwd(hb.view_of(img), hb.view_of(est), hb.view_of(xout), hb.view_of(dy), tv.data_ptr(),
                                          zb.data_ptr(), zb.data_ptr(), hb.view_of(dy), hb.view_of(dout5), hb.view_of(dest),
                                          B, H, W, 0, hb.stream_ptr()))
    dxin[:, 6:10] = dest
    return None, dxin, dout5, None


synthesize.register_autograd(_synth_backward, setup_context=_synth_setup)
