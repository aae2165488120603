"""uint8 frames <-> the path's normalised, zero-padded fp32 tensors, on the device.

Replaces the host-side preprocessing either side of the hot path:
  * ingest: `Interpolator.load_batch` + `normalize_tensor` (scripts/visualize_interpolation.py:61-88,257-262) and the
    evaluator's ToTensor + Normalize + EvalPad (scripts/utils/dataloaders/augmentations.py:141-200);
  * egress: `get_crop` + `denormalize` + `astype(uint8)` (scripts/evaluate_interpolation_results.py:143-163,192-202).
Frames cross PCIe as uint8 (4x fewer bytes than fp32) and every elementwise step is one HIP kernel each way.
"""
import ctypes
import math

import torch

from . import hipbind as hb
from .weights import IMAGENET_MEAN, IMAGENET_STD


def _f3(v):
    return (ctypes.c_float * 3)(*[float(x) for x in v])


def cfg_mean_std(cfg=None):
    """MODEL.PIXEL_MEAN / PIXEL_STD of the ini (configs/superslomo_original.ini:57-58)."""
    if cfg is None:
        return IMAGENET_MEAN, IMAGENET_STD
    return (tuple(float(p) for p in cfg.get("MODEL", "PIXEL_MEAN").split(",")),
            tuple(float(p) for p in cfg.get("MODEL", "PIXEL_STD").split(",")))


def padded_dims(h, w, multiple=32):
    """(Hp, Wp), (top, left): centred padding to the next multiple of 32
    (scripts/evaluate_interpolation_results.py:76-99 get_dims; visualize_interpolation.py:76-87)."""
    hp, wp = int(math.ceil(h / multiple) * multiple), int(math.ceil(w / multiple) * multiple)
    return (hp, wp), ((hp - h) // 2, (wp - w) // 2)


def frames_from_u8(frames_u8, cfg=None, pad_before_norm=False):
    """[N,H,W,3] uint8 RGB device tensor -> [N,3,Hp,Wp] normalised fp32, zero-padded to a multiple of 32.
    pad_before_norm=False: dataloader convention (pad value 0 in normalised space);
    True: visualiser convention (black pixels padded before normalisation)."""
    assert frames_u8.is_cuda and frames_u8.dtype == torch.uint8 and frames_u8.dim() == 4 and frames_u8.shape[3] == 3, \
        "frames must be a [N,H,W,3] uint8 tensor on the GPU"
    f = frames_u8.contiguous()
    n, h, w, _ = f.shape
    (hp, wp), (top, left) = padded_dims(h, w)
    mean, std = cfg_mean_std(cfg)
    out = torch.empty(n, 3, hp, wp, dtype=torch.float32, device=f.device)
    hb.check(hb.load().ssm_frames_from_u8_fwd(f.data_ptr(), hb.view_of(out), n, h, w, hp, wp, top, left, _f3(mean), _f3(std),
                                              1 if pad_before_norm else 0, hb.stream_ptr()))
    return out


def frames_to_u8(x, h, w, cfg=None, saturate=False):
    """[N,3,Hp,Wp] normalised fp32 -> [N,h,w,3] uint8 with the centred padding cropped away.
    saturate=False reproduces the reference's numpy cast (truncate, wrap); True rounds and clamps."""
    hb.require_device(x, "frame tensor")
    x = x.contiguous()
    n, _, hp, wp = x.shape
    top, left = (hp - h) // 2, (wp - w) // 2
    mean, std = cfg_mean_std(cfg)
    out = torch.empty(n, h, w, 3, dtype=torch.uint8, device=x.device)
    hb.check(hb.load().ssm_frames_to_u8_fwd(hb.view_of(x), out.data_ptr(), n, h, w, top, left, _f3(mean), _f3(std),
                                            1 if saturate else 0, hb.stream_ptr()))
    return out
