"""Backward pass of the hot path on the HIP kernels (training step, SURVEY 8f-1 / BASELINE config 3).

The reference trains through autograd over stock PyTorch ops; here every adjoint is a C-ABI call on fp32 padded
planes (the training plan runs in precision mode "f32"):
  conv data gradient   = the forward convolution kernel on the transposed, spatially flipped filter
  conv weight gradient = ssm_conv2d_wgrad, bias gradient = ssm_bias_grad
  LeakyReLU' + adjoint of the fused 2x2 mean = ssm_lrelu_bwd
  adjoint of concat + bilinear x2 = ssm_upsample2x_cat_bwd
  adjoints of compute_inputs / compute_output_image fused with the L1 loss gradients = ssm_flowinterp_inputs_bwd /
  ssm_synthesize_bwd
"""
import torch

from . import hipbind as hb


def transposed_filter(w):
    """OIHW filter of the data-gradient convolution: W'[ci][co][ky][kx] = W[co][ci][k-1-ky][k-1-kx]."""
    return w.detach().permute(1, 0, 2, 3).flip(2, 3).contiguous()


def lrelu_bwd(dy, dpool, y, dz, slope=0.1, has_act=True):
    """dz = (dy + 1/4 dpool^) * LeakyReLU'(y).  dy / dpool / y / dz: hb.Planes (dy or dpool may be None)."""
    hb.check(hb.load().ssm_lrelu_bwd(dy.view() if dy is not None else hb.NULL_VIEW,
                                     dpool.view() if dpool is not None else hb.NULL_VIEW,
                                     y.view() if y is not None else hb.NULL_VIEW, dz.view(), dz.B, dz.C, dz.H, dz.W, slope,
                                     1 if has_act else 0, hb.stream_ptr()))
    return dz


def bias_grad(dz, out):
    hb.check(hb.load().ssm_bias_grad(dz.view(), out.data_ptr(), dz.B, dz.C, dz.H, dz.W, hb.stream_ptr()))
    return out


def wgrad(x, dz, out, k):
    """out (OIHW fp32, contiguous) = sum_{b,y,x} dz * x(shifted).  x, dz: hb.Planes of the layer's input / dZ."""
    assert out.is_contiguous() and tuple(out.shape) == (dz.C, x.C, k, k)
    hb.check(hb.load().ssm_conv2d_wgrad(x.view(), dz.view(), out.data_ptr(), x.B, x.C, dz.C, x.H, x.W, k, hb.stream_ptr()))
    return out


def upsample_cat_bwd(du, da, db=None, acc_a=False, acc_b=False):
    hb.check(hb.load().ssm_upsample2x_cat_bwd(du.view(), da.view(), da.C, db.view() if db is not None else hb.NULL_VIEW,
                                              db.C if db is not None else 0, da.B, da.H, da.W, 1 if acc_a else 0,
                                              1 if acc_b else 0, hb.stream_ptr()))
