"""ctypes binding of libssm_hip.so (C ABI: include/ssm_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every compute
call goes through the C ABI with raw pointers.  There is NO fallback: if the
library is missing or a call fails this raises.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSM_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libssm_hip.so")

SSM_PADX = 4
SSM_PADY = 3
SSM_TAIL_SLACK_FLOATS = 1 << 16
SSM_FLAG_LRELU = 1
SSM_FLAG_MASK = 8      # the `add` view is a mask source: out = conv(x) * (add > 0 ? 1 : slope) (include/ssm_hip.h)


class SsmView(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("sb", ctypes.c_longlong), ("sc", ctypes.c_longlong),
                ("sh", ctypes.c_int)]


NULL_VIEW = SsmView(None, 0, 0, 0)


class SsmHView(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("sb", ctypes.c_longlong), ("sg", ctypes.c_longlong),
                ("sp", ctypes.c_longlong), ("sh", ctypes.c_int)]


NULL_HVIEW = SsmHView(None, 0, 0, 0, 0)
SSM_FLAG_FP16_FAST = 2
SSM_FLAG_Q8 = 4

_c_int, _c_float, _vp, _sz = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
_ip = ctypes.POINTER(ctypes.c_int)

# name -> (restype, argtypes); kept in step with include/ssm_hip.h (tests check every symbol)
SIGNATURES = {
    "ssm_abi_version": (_c_int, []),
    "ssm_last_error_string": (ctypes.c_char_p, []),
    "ssm_plane_dims": (None, [_c_int, _c_int, _ip, _ip]),
    "ssm_copy_view": (_c_int, [SsmView, SsmView, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_conv_config": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _ip, _ip]),
    "ssm_packed_weight_floats": (_sz, [_c_int, _c_int, _c_int, _c_int]),
    "ssm_packed_bias_floats": (_sz, [_c_int, _c_int]),
    "ssm_pack_weights": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_conv2d_fwd": (_c_int, [SsmView, _c_int, SsmView, _c_int, _vp, _vp, SsmView, SsmView, _c_int, _c_int,
                                _c_int, _c_int, _c_int, _c_float, _c_int, _vp]),
    "ssm_conv2d_add_fwd": (_c_int, [SsmView, _c_int, SsmView, _c_int, _vp, _vp, SsmView, SsmView, SsmView, _c_int, _c_int, _c_int,
                                    _c_int, _c_int, _c_int, _c_float, _c_int, _vp]),
    "ssm_conv2d_ups_add_fwd": (_c_int, [SsmView, _c_int, SsmView, _c_int, _vp, _vp, SsmView, SsmView, _c_int, _c_int, _c_int, _c_int,
                                        _c_int, _c_float, _c_int, _vp]),
    "ssm_wino_conv2d_add_fwd": (_c_int, [SsmView, _c_int, SsmView, _c_int, _vp, _vp, SsmView, SsmView, SsmView, _c_int, _c_int, _c_int,
                                         _c_int, _c_int, _c_float, _c_int, _vp]),
    "ssm_wino_conv2d_ups_add_fwd": (_c_int, [SsmView, _c_int, SsmView, _c_int, _vp, _vp, SsmView, SsmView, _c_int, _c_int, _c_int, _c_int,
                                             _c_int, _c_float, _c_int, _vp]),
    "ssm_wino_plan": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _ip, _ip, _ip]),
    "ssm_wino_splitk_plan": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _ip]),
    "ssm_conv_splitk_plan": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _ip]),
    "ssm_conv2d_splitk_fwd": (_c_int, [SsmView, _c_int, SsmView, _c_int, _vp, _vp, SsmView, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_wino_conv2d_splitk_fwd": (_c_int, [SsmView, _c_int, SsmView, _c_int, _vp, _vp, SsmView, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                            _c_int, _vp]),
    "ssm_splitk_finish_fwd": (_c_int, [SsmView, _c_int, SsmView, SsmView, SsmView, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _vp]),
    "ssm_wino_force_kind": (_c_int, [_c_int]),
    "ssm_wino_packed_weight_floats": (_sz, [_c_int, _c_int, _c_int]),
    "ssm_wino_pack_weights": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _vp]),
    "ssm_pack32_weights_batch": (_c_int, [_vp, _c_int, ctypes.c_longlong, _vp]),
    "ssm_pack32_wino_tiles_batch": (_c_int, [_vp, _c_int, ctypes.c_longlong, _c_int, _vp]),
    "ssm_wino4_plan": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _ip, _ip, _ip]),
    "ssm_wino4_force_kind": (_c_int, [_c_int]),
    "ssm_wino4_preferred": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int]),
    "ssm_wino_estimate": (ctypes.c_double, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int]),
    "ssm_wino4_packed_weight_floats": (_sz, [_c_int, _c_int]),
    "ssm_wino4_pack_weights": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _vp]),
    "ssm_wino4_conv2d_add_fwd": (_c_int, [SsmView, _c_int, SsmView, _c_int, _vp, _vp, SsmView, SsmView, SsmView, _c_int, _c_int, _c_int,
                                          _c_int, _c_int, _c_float, _c_int, _vp]),
    "ssm_wino4_conv2d_ups_add_fwd": (_c_int, [SsmView, _c_int, SsmView, _c_int, _vp, _vp, SsmView, SsmView, _c_int, _c_int, _c_int, _c_int,
                                              _c_int, _c_float, _c_int, _vp]),
    "ssm_wino4_conv2d_shuffle_fwd": (_c_int, [SsmView, _c_int, SsmView, _c_int, _vp, _vp, SsmView, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _vp]),
    "ssm_wino4_conv2d_ups_border_fwd": (_c_int, [SsmView, _c_int, SsmView, _c_int, _vp, _vp, SsmView, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int,
                                                 _vp]),
    "ssm_wino1d_plan": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _ip, _ip, _ip]),
    "ssm_wino1d_force_kind": (_c_int, [_c_int]),
    "ssm_wino1d_packed_weight_floats": (_sz, [_c_int, _c_int, _c_int, _c_int]),
    "ssm_wino1d_pack_weights": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_wino1d_conv2d_add_fwd": (_c_int, [SsmView, _c_int, _vp, _vp, SsmView, SsmView, SsmView, _c_int, _c_int, _c_int, _c_int, _c_int,
                                           _c_int, _c_float, _c_int, _vp]),
    "ssm_wino5_plan": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _ip]),
    "ssm_wino5_force_kind": (_c_int, [_c_int]),
    "ssm_wino5_packed_weight_floats": (_sz, [_c_int, _c_int]),
    "ssm_wino5_pack_weights": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _vp]),
    "ssm_wino5_conv2d_add_fwd": (_c_int, [SsmView, _c_int, _vp, _vp, SsmView, SsmView, SsmView, _c_int, _c_int, _c_int, _c_int, _c_int,
                                          _c_float, _c_int, _vp]),
    "ssm_wino7_plan": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _ip]),
    "ssm_wino7_force_kind": (_c_int, [_c_int]),
    "ssm_wino7_packed_weight_floats": (_sz, [_c_int, _c_int]),
    "ssm_wino7_pack_weights": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _vp]),
    "ssm_wino7_conv2d_add_fwd": (_c_int, [SsmView, _c_int, _vp, _vp, SsmView, SsmView, SsmView, _c_int, _c_int, _c_int, _c_int, _c_int,
                                          _c_float, _c_int, _vp]),
    "ssm_wino_conv2d_fwd": (_c_int, [SsmView, _c_int, SsmView, _c_int, _vp, _vp, SsmView, SsmView, _c_int, _c_int, _c_int, _c_int,
                                     _c_float, _c_int, _vp]),
    "ssm_wino_conv2d_ups_fwd": (_c_int, [SsmView, _c_int, SsmView, _c_int, _vp, _vp, SsmView, _c_int, _c_int, _c_int, _c_int,
                                         _c_float, _c_int, _vp]),
    "ssm_conv16_config": (_c_int, [_c_int, _c_int, _c_int, _ip, _ip]),
    "ssm_packed16_weight_halves": (_sz, [_c_int, _c_int, _c_int, _c_int]),
    "ssm_pack16_weights": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _vp]),
    "ssm_conv2d_hl8_fwd": (_c_int, [SsmHView, _c_int, SsmHView, _c_int, _vp, _vp, _c_float, SsmHView, SsmView, SsmHView,
                                    _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _vp]),
    "ssm_conv16q_config": (_c_int, [_c_int, _c_int, _c_int, _ip, _ip]),
    "ssm_conv16q_ups_config": (_c_int, [_c_int, _c_int, _ip, _ip]),
    "ssm_packed16q_weight_bytes": (_sz, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "ssm_pack16q_weights": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _vp]),
    "ssm_hq8_from_f32": (_c_int, [SsmView, SsmHView, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_hq8_to_f32": (_c_int, [SsmHView, SsmView, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_conv16_ups_config": (_c_int, [_c_int, _c_int, _ip, _ip]),
    "ssm_conv2d_ups_hl8_fwd": (_c_int, [SsmHView, _c_int, SsmHView, _c_int, _vp, _vp, _c_float, SsmHView, SsmView,
                                        _c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _vp]),
    "ssm_hl8_from_f32": (_c_int, [SsmView, SsmHView, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_hl8_to_f32": (_c_int, [SsmHView, SsmView, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_upsample2x_cat_hl8_fwd": (_c_int, [SsmHView, _c_int, SsmHView, _c_int, SsmHView, _c_int, _c_int, _c_int, _vp]),
    "ssm_flowinterp_inputs_hl8_fwd": (_c_int, [SsmView, SsmView, _vp, SsmHView, SsmView, _c_int, _c_int, _c_int, _vp]),
    "ssm_flowinterp_inputs_hq8_fwd": (_c_int, [SsmView, SsmView, _vp, SsmHView, SsmView, _c_int, _c_int, _c_int, _vp]),
    "ssm_frames_from_u8_fwd": (_c_int, [_vp, SsmView, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                        ctypes.POINTER(_c_float), ctypes.POINTER(_c_float), _c_int, _vp]),
    "ssm_frames_to_u8_fwd": (_c_int, [SsmView, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.POINTER(_c_float),
                                      ctypes.POINTER(_c_float), _c_int, _vp]),
    "ssm_warp_bilinear_bwd": (_c_int, [SsmView, SsmView, SsmView, SsmView, SsmView, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_lrelu_bwd": (_c_int, [SsmView, SsmView, SsmView, SsmView, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _vp]),
    "ssm_lrelu_bwd_q8": (_c_int, [SsmView, SsmView, SsmView, SsmView, SsmHView, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _vp]),
    "ssm_bias_grad": (_c_int, [SsmView, _vp, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_conv2d_wgrad_bias": (_c_int, [SsmView, SsmView, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_conv_plan": (_c_int, [_c_int] * 8 + [ctypes.POINTER(ctypes.c_int)] * 3),
    "ssm_conv_force_kind": (_c_int, [_c_int]),
    "ssm_conv2d_ups_fwd": (_c_int, [SsmView, _c_int, SsmView, _c_int, _vp, _vp, SsmView, _c_int, _c_int, _c_int, _c_int, ctypes.c_float,
                                    _c_int, _vp]),
    "ssm_conv2d_hl8_subpixel_fwd": (_c_int, [SsmHView, _c_int, SsmHView, _c_int, _vp, _vp, ctypes.c_float, SsmHView, _c_int, _c_int, _c_int,
                                              _c_int, _c_int, _c_float, _c_int, _vp]),
    "ssm_hl8_gather_cols": (_c_int, [SsmHView, _c_int, SsmHView, _c_int, SsmHView, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_conv16_subpixel_table_bytes": (ctypes.c_size_t, [_c_int]),
    "ssm_conv16_subpixel_plan": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, ctypes.c_float, _c_int, _vp, ctypes.c_size_t,
                                          ctypes.POINTER(ctypes.c_int)]),
    "ssm_conv16_subpixel_run": (_c_int, [_vp, _c_int, ctypes.POINTER(ctypes.c_int), _c_int, _c_int, _c_int, _vp]),
    "ssm_pack16q_job_blocks": (_c_int, [_c_int, _c_int, _c_int, _c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "ssm_pack16q_weights_batch": (_c_int, [_vp, _c_int, _c_int, _vp]),
    "ssm_bias_grad_acc": (_c_int, [SsmView, _vp, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_conv2d_wgrad": (_c_int, [SsmView, SsmView, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_conv2d_wgrad_bf16x3": (_c_int, [SsmView, SsmView, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_splitk_enable": (_c_int, [_c_int, _c_int]),
    "ssm_program_create": (_c_int, [ctypes.POINTER(_vp)]),
    "ssm_program_destroy": (_c_int, [_vp]),
    "ssm_program_begin": (_c_int, [_vp, ctypes.POINTER(_vp), _c_int]),
    "ssm_program_mark": (_c_int, [_vp, ctypes.POINTER(_c_int)]),
    "ssm_program_end": (_c_int, [_vp, ctypes.POINTER(_c_int)]),
    "ssm_program_run": (_c_int, [_vp, _c_int, _c_int, ctypes.POINTER(_vp), _c_int]),
    "ssm_stream_wait": (_c_int, [_vp, _vp]),
    "ssm_wgrad_wino_supported": (_c_int, [_c_int] * 5),
    "ssm_wgrad_wino_scratch_floats": (ctypes.c_longlong, [_c_int, _c_int]),
    "ssm_conv2d_wgrad_wino": (_c_int, [SsmView, SsmView, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_wgrad_wino_finish": (_c_int, [_vp, _c_int, _c_int, _c_float, _vp]),
    "ssm_upsample2x_cat_bwd": (_c_int, [SsmView, SsmView, _c_int, SsmView, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_upsample2x_cat_bwd_mask": (_c_int, [SsmView, SsmView, _c_int, SsmView, _c_int, SsmView, _c_float, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_synthesize_bwd": (_c_int, [SsmView, SsmView, SsmView, SsmView, _vp, _vp, _vp, SsmView, SsmView, SsmView, _c_int, _c_int,
                                    _c_int, _c_int, _vp]),
    "ssm_maxpool2_fwd": (_c_int, [SsmView, SsmView, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_maxpool2_bwd": (_c_int, [SsmView, SsmView, SsmView, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_sqdiff_grad": (_c_int, [SsmView, SsmView, _vp, SsmView, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_sqdiff_mean": (_c_int, [SsmView, SsmView, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_train_loss_sums": (_c_int, [SsmView, SsmView, SsmView, SsmView, SsmView, SsmView, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_flowinterp_inputs_bwd": (_c_int, [SsmView, SsmView, SsmView, SsmView, _vp, _vp, SsmView, _c_int, _c_int, _c_int, _c_int,
                                           _vp]),
    "ssm_convlstm_cell_fwd": (_c_int, [SsmView, SsmView, SsmView, SsmView, SsmView, SsmHView, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_convgru_reset_fwd": (_c_int, [SsmView, SsmView, SsmView, SsmView, SsmHView, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_convgru_update_fwd": (_c_int, [SsmView, SsmView, SsmView, SsmView, SsmView, SsmView, SsmHView, _c_int, _c_int, _c_int,
                                        _c_int, _c_int, _vp]),
    "ssm_convlstm_cell_bwd": (_c_int, [SsmView] * 7 + [_c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_convgru_reset_bwd": (_c_int, [SsmView] * 5 + [_c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_convgru_update_bwd": (_c_int, [SsmView] * 7 + [_c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_avgpool2_fwd": (_c_int, [SsmView, SsmView, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_upsample2x_cat_fwd": (_c_int, [SsmView, _c_int, SsmView, _c_int, SsmView, _c_int, _c_int, _c_int, _vp]),
    "ssm_warp_bilinear_fwd": (_c_int, [SsmView, SsmView, SsmView, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ssm_flowinterp_inputs_fwd": (_c_int, [SsmView, SsmView, _vp, SsmView, _c_int, _c_int, _c_int, _vp]),
    "ssm_final_conv_fwd": (_c_int, [SsmView, _vp, _vp, _c_int, SsmView, SsmView, SsmView, _vp, SsmView, SsmView, _c_int, _c_int, _c_int, _vp]),
    "ssm_flowinterp_inputs_t_fwd": (_c_int, [SsmView, SsmView, _vp, SsmView, _c_int, _c_int, _c_int, _vp]),
    "ssm_synthesize_fwd": (_c_int, [SsmView, SsmView, SsmView, _vp, SsmView, SsmView, _c_int, _c_int, _c_int, _vp]),
}

_lib = None


def load():
    """Load the library once.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libssm_hip.so not found at %s - build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C superslomo-videointerpolation-pytorch_amd/csrc`. There is no "
                "CPU/PyTorch fallback for the hot path." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        msg = load().ssm_last_error_string()
        raise RuntimeError("libssm_hip: %s (code %d)" % (msg.decode() if msg else "error", rc))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """torch's current HIP stream of the current device as the C ABI's `void *stream`.  The raw accessor skips the Stream object
    torch.cuda.current_stream() builds (9 us per call - with ~400 launches per training step that was 3.5 ms of host time per step)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- launch programs (csrc/ssm_program.cpp): record one pass through the C ABI, replay it without the interpreter ----------------------
_recorder = None          # the LaunchProgram that is recording (process-wide, like the C side), or None
_recorder_thread = None   # ... and the thread it records: other threads' host_op calls are plain calls (as their launches are on the C side)


class ProgramBusy(RuntimeError):
    """Another launch program is recording in this process (one at a time): record later."""


def host_op(fn):
    """Run `fn()` - framework-side GPU work inside a pass that may be recorded: a torch kernel on the plan's static tensors, an RCCL
    bucket.  Outside a recording it is just the call.  While a program records, the call also becomes an item of the program between two
    ranges of C-ABI nodes and is called again, on the stream that was current here, at every replay: `fn` must read and write STATIC
    tensors only (anything it allocates is gone by the next step) and return nothing."""
    if _recorder is None or threading.get_ident() != _recorder_thread:
        fn()
    else:
        _recorder._host_op(fn)


def stream_wait(src, dst):
    """`dst` (torch.cuda.Stream) waits for everything queued on `src` so far - torch's dst.wait_stream(src), through the C ABI so that
    the ordering is part of a recorded program (ssm_stream_wait)."""
    check(load().ssm_stream_wait(ctypes.c_void_p(src.cuda_stream), ctypes.c_void_p(dst.cuda_stream)))


class LaunchProgram:
    """One recorded pass: ranges of C-ABI nodes (replayed by ssm_program_run) interleaved with the framework-side calls of `host_op`.

        prog = LaunchProgram([main, side, ...])          # torch.cuda.Stream objects = the program's stream slots
        with prog.recording():
            step_body()                                   # launches run as usual and are recorded
        prog.replay()                                     # the same launches, ~2 us of host time each

    The body must be a pass whose pointers do not change from step to step (plans, static input buffers), warmed up before it is recorded
    (lazily built buffers, packed filters), and must route its torch-side GPU work through hb.host_op and its cross-stream ordering
    through hb.stream_wait."""

    def __init__(self, streams):
        assert 1 <= len(streams) <= 8
        self.streams = list(streams)
        self._raw = (ctypes.c_void_p * len(streams))(*[s.cuda_stream for s in streams])
        h = ctypes.c_void_p()
        check(load().ssm_program_create(ctypes.byref(h)))
        self._h = h
        self.items = []          # ("c", first, last) | ("py", fn, stream or None)
        self._cut = 0
        self.n_nodes = 0
        self.ready = False

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                load().ssm_program_destroy(self._h)
                self._h = None
        except Exception:          # noqa: BLE001 - interpreter shutdown
            pass

    def _mark(self):
        n = ctypes.c_int(0)
        check(load().ssm_program_mark(self._h, ctypes.byref(n)))
        if n.value > self._cut:
            self.items.append(("c", self._cut, n.value))
            self._cut = n.value

    def _host_op(self, fn):
        self._mark()
        fn()
        cur = torch.cuda.current_stream()
        self.items.append(("py", fn, None if cur.cuda_stream == self.streams[0].cuda_stream else cur))

    def recording(self):
        prog = self

        class _Ctx:
            def __enter__(self):
                global _recorder, _recorder_thread
                if _recorder is not None:
                    raise ProgramBusy("another launch program is recording in this process")
                rc = load().ssm_program_begin(prog._h, prog._raw, len(prog.streams))
                if rc and b"another program is recording" in load().ssm_last_error_string():
                    raise ProgramBusy("another launch program is recording in this process")
                check(rc)
                _recorder, _recorder_thread = prog, threading.get_ident()
                return prog

            def __exit__(self, et, ev, tb):
                global _recorder, _recorder_thread
                _recorder = _recorder_thread = None
                if et is None:
                    prog._mark()
                n = ctypes.c_int(0)
                rc = load().ssm_program_end(prog._h, ctypes.byref(n))
                if et is None:
                    check(rc)
                    prog.n_nodes = n.value
                    prog.ready = True
                return False
        return _Ctx()

    def replay(self):
        """Issue the recorded pass again.  The caller's current stream must be slot 0's."""
        assert self.ready, "nothing recorded"
        lib, h, raw, n = load(), self._h, self._raw, len(self.streams)
        run = lib.ssm_program_run
        for it in self.items:
            if it[0] == "c":
                rc = run(h, it[1], it[2], raw, n)
                if rc:
                    check(rc)
            elif it[2] is None:
                it[1]()
            else:
                with torch.cuda.stream(it[2]):
                    it[1]()


def require_device(t, what="tensor"):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError("%s must be a float32 tensor on the GPU (the HIP path has no CPU fallback); got %s on %s"
                           % (what, getattr(t, "dtype", type(t)), getattr(t, "device", "?")))


def view_of(t):
    """ssm_view of a 4-D torch tensor [B,C,H,W] whose last stride is 1 (any other strides)."""
    require_device(t)
    assert t.dim() == 4 and (t.shape[3] == 1 or t.stride(3) == 1), "need [B,C,H,W] with unit x-stride"
    return SsmView(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


def plane_dims(h, w):
    return h + 2 * SSM_PADY, (w + 2 * SSM_PADX + 3) // 4 * 4


class Planes:
    """A [B,C,H,W] activation in the padded-plane layout (include/ssm_hip.h): zero frame
    of SSM_PADY rows / SSM_PADX cols around every plane, rows 16-byte aligned, plus the
    readable tail slack the convolution's tile overshoot needs.  Zeroed once at
    allocation; kernels only write interiors."""

    def __init__(self, B, C, H, W, device):
        self.B, self.C, self.H, self.W = B, C, H, W
        self.Hp, self.Wp = plane_dims(H, W)
        n = B * C * self.Hp * self.Wp
        self.buf = torch.zeros(n + SSM_TAIL_SLACK_FLOATS, dtype=torch.float32, device=device)
        self.full = self.buf[:n].view(B, C, self.Hp, self.Wp)

    @property
    def interior(self):
        """torch view of the logical tensor (non-contiguous)."""
        return self.full[:, :, SSM_PADY:SSM_PADY + self.H, SSM_PADX:SSM_PADX + self.W]

    def view(self, c0=0, broadcast=False, b0=0, y0=0, x0=0):
        """ssm_view of channels c0.. of batch entries b0.. (broadcast: every batch index reads entry b0), its origin at pixel (y0, x0)."""
        sc = self.Hp * self.Wp
        base = self.buf.data_ptr() + 4 * (((b0 * self.C + c0) * self.Hp + SSM_PADY + y0) * self.Wp + SSM_PADX + x0)
        return SsmView(base, 0 if broadcast else self.C * sc, sc, self.Wp)

    def slice(self, c0, c):
        """Channels [c0, c0+c) as a Planes-like object over the same storage (same batch stride)."""
        parent = self

        class _Slice:
            B, C, H, W, Hp, Wp = parent.B, c, parent.H, parent.W, parent.Hp, parent.Wp

            @staticmethod
            def view(cc=0, broadcast=False):
                return parent.view(c0 + cc, broadcast)

            @property
            def interior(self):
                return parent.interior[:, c0:c0 + c]
        return _Slice()

    def load(self, x):
        """Copy a [B,C,H,W] device tensor into the interior (HIP strided copy)."""
        x = x if x.stride(3) == 1 else x.contiguous()
        lib = load()
        check(lib.ssm_copy_view(view_of(x), self.view(), self.B, self.C, self.H, self.W, stream_ptr()))
        return self

    def to_nchw(self):
        out = torch.empty(self.B, self.C, self.H, self.W, dtype=torch.float32, device=self.buf.device)
        lib = load()
        check(lib.ssm_copy_view(self.view(), view_of(out), self.B, self.C, self.H, self.W, stream_ptr()))
        return out


def conv_plan(k, cin, cout, B, H, W, pool=False, ups=False):
    """(kind, BN, CK) of the tile configuration ssm_conv2d_fwd / ssm_conv2d_ups_fwd will use for this problem."""
    lib = load()
    kind, bn, ck = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    check(lib.ssm_conv_plan(k, cin, cout, B, H, W, 1 if pool else 0, 1 if ups else 0, ctypes.byref(kind), ctypes.byref(bn),
                            ctypes.byref(ck)))
    return kind.value, bn.value, ck.value


def conv_config(k, cout, B, H, W, pool=False):
    """(BN, CK) of ssm_conv_config: the plan of a filter with as many input as output channels (historical form)."""
    lib = load()
    bn, ck = ctypes.c_int(0), ctypes.c_int(0)
    check(lib.ssm_conv_config(k, cout, B, H, W, 1 if pool else 0, ctypes.byref(bn), ctypes.byref(ck)))
    return bn.value, ck.value


class PackedConv:
    """Filter + bias of one convolution repacked for the tile configuration the
    library picks for the problem (k, Cout, batch, map size, fused pool).  An explicit
    handle owned by the Python side (packed-weight caches are
    created/destroyed by the caller)."""

    algo = "direct"

    def __init__(self, weight, bias, B, H, W, pool=False, ups=False):
        """B, H, W: batch and OUTPUT map of the launches this filter serves; ups: it feeds ssm_conv2d_ups_fwd."""
        require_device(weight, "conv weight")
        require_device(bias, "conv bias")
        self.cout, self.cin, self.k = weight.shape[0], weight.shape[1], weight.shape[2]
        self.ups = bool(ups)
        _, self.bn, self.ck = conv_plan(self.k, self.cin, self.cout, B, H, W, pool, ups)
        self.cin_p = (self.cin + self.ck - 1) // self.ck * self.ck
        lib = load()
        nw = lib.ssm_packed_weight_floats(self.cout, self.cin_p, self.k, self.bn)
        nb = lib.ssm_packed_bias_floats(self.cout, self.bn)
        self.w = torch.empty(nw, dtype=torch.float32, device=weight.device)
        self.b = torch.empty(nb, dtype=torch.float32, device=weight.device)
        wc, bc = weight.detach().contiguous(), bias.detach().contiguous()
        check(lib.ssm_pack_weights(wc.data_ptr(), bc.data_ptr(), self.w.data_ptr(), self.b.data_ptr(), self.cout,
                                   self.cin, self.cin_p, self.k, self.bn, stream_ptr()))


def conv2d(x1, c1, x2, c2, pk, y, pool, B, H, W, lrelu=True, slope=0.1, add=None, add_div=1):
    """add: optional pre-activation addend view [B / add_div, Cout, H, W] (batch entry b reads entry b // add_div)."""
    lib = load()
    assert pk.cin_p == c1 + c2, "packed filter expects %d input channels, got %d" % (pk.cin_p, c1 + c2)
    assert (pk.bn, pk.ck) == conv_plan(pk.k, c1 + c2, pk.cout, B, H, W, pool is not None)[1:], \
        "filter was packed for another tile configuration (batch/size/pool changed)"
    # launches that leave most of the chip idle run split over the input channels (see wino_splitk) - only where the plan allows it
    # (pk.split_ok: mode f32w; mode f32 and torch.ops.ssm.conv2d keep the reference's one fmaf chain per output)
    # (the split kernel offsets its FIRST source by the split's channel range: two-source launches fall through to the plain kernel)
    if pool is None and x2 is None and c2 == 0 and getattr(pk, "split_ok", False):
        key = (B, H, W)
        cache = pk.__dict__.setdefault("_splitk", {})
        if key not in cache:
            ks = ctypes.c_int(1)
            check(lib.ssm_conv_splitk_plan(pk.k, c1 + c2, pk.cout, B, H, W, ctypes.byref(ks)))
            cache[key] = ks.value
        ks = cache[key]
        if ks > 1:
            st = stream_ptr()
            scratch = pk.__dict__.setdefault("_splitk_part", {})
            skey = (getattr(st, "value", st), ks, B, H, W)
            if skey not in scratch:
                scratch[skey] = Planes(ks * B, pk.cout, H, W, pk.w.device)
            part = scratch[skey]
            check(lib.ssm_conv2d_splitk_fwd(x1, c1, x2 if x2 is not None else NULL_VIEW, c2, pk.w.data_ptr(), pk.b.data_ptr(), part.view(), ks,
                                            B, H, W, pk.cout, pk.k, st))
            check(lib.ssm_splitk_finish_fwd(part.view(), ks, y, NULL_VIEW, add if add is not None else NULL_VIEW, add_div, B, pk.cout, H, W,
                                            slope, SSM_FLAG_LRELU if lrelu else 0, st))
            return
    check(lib.ssm_conv2d_add_fwd(x1, c1, x2 if x2 is not None else NULL_VIEW, c2, pk.w.data_ptr(), pk.b.data_ptr(), y,
                                 pool if pool is not None else NULL_VIEW, add if add is not None else NULL_VIEW, add_div, B, H, W,
                                 pk.cout, pk.k, slope, SSM_FLAG_LRELU if lrelu else 0, stream_ptr()))


def conv2d_ups(a, c1, b, c2, pk, y, B, H, W, lrelu=True, slope=0.1, add=None, add_div=1):
    """conv3x3(upsample2x(cat[a, b])) in exact fp32: a, b LOW-res padded-plane views, H, W the OUTPUT size."""
    lib = load()
    assert pk.k == 3 and pk.cin_p == c1 + c2, "packed 3x3 filter expects %d input channels, got %d" % (pk.cin_p, c1 + c2)
    assert (pk.bn, pk.ck) == conv_plan(3, c1 + c2, pk.cout, B, H, W, False, True)[1:], \
        "filter was packed for another tile configuration (batch/size changed, or not packed with ups=True)"
    check(lib.ssm_conv2d_ups_add_fwd(a, c1, b if b is not None else NULL_VIEW, c2, pk.w.data_ptr(), pk.b.data_ptr(), y,
                                     add if add is not None else NULL_VIEW, add_div, B, H, W, pk.cout, slope,
                                     SSM_FLAG_LRELU if lrelu else 0, stream_ptr()))


# ---- 3x3 convolution as Winograd F(2x2,3x3) in fp32 (csrc/ssm_wino.hip) ------------------------------------
def wino_plan(cin, cout, B, H, W, ups=False):
    """(kind, BN, CK) of the Winograd tile configuration for the problem (ups: the fused-upsample entry point)."""
    lib = load()
    kind, bn, ck = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    check(lib.ssm_wino_plan(cin, cout, B, H, W, 1 if ups else 0, ctypes.byref(kind), ctypes.byref(bn), ctypes.byref(ck)))
    return kind.value, bn.value, ck.value


def wino_supported(cin, cout, H, W, k=3):
    """Can this layer run in the Winograd form?  (3x3, whole 8-channel chunks)"""
    return k == 3 and cin % 8 == 0          # (any map size: odd widths since r6 - the epilogue stores a tile's lone last column by itself)


class PackedWino:
    """3x3 filter pre-transformed (U = G g G^T) and packed for the Winograd kernel's tile configuration; an explicit handle owned
    by the Python side like PackedConv."""

    algo = "wino"

    def __init__(self, weight, bias, B, H, W, pool=False, ups=False):
        require_device(weight, "conv weight")
        require_device(bias, "conv bias")
        self.cout, self.cin, self.k = weight.shape[0], weight.shape[1], weight.shape[2]
        assert self.k == 3 and weight.shape[3] == 3, "the Winograd form is for 3x3 filters"
        self.ups = bool(ups)
        _, self.bn, self.ck = wino_plan(self.cin, self.cout, B, H, W, self.ups)
        assert self.cin % self.ck == 0, "Cin must be a multiple of %d" % self.ck
        self.cin_p = self.cin
        lib = load()
        nw = lib.ssm_wino_packed_weight_floats(self.cout, self.cin, self.bn)
        nb = lib.ssm_packed_bias_floats(self.cout, self.bn)
        self.w = torch.empty(nw, dtype=torch.float32, device=weight.device)
        self.b = torch.empty(nb, dtype=torch.float32, device=weight.device)
        wc, bc = weight.detach().contiguous(), bias.detach().contiguous()
        check(lib.ssm_wino_pack_weights(wc.data_ptr(), bc.data_ptr(), self.w.data_ptr(), self.b.data_ptr(), self.cout, self.cin,
                                        self.bn, stream_ptr()))


def wino_splitk(pk, B, H, W, ups=False):
    """Split factor the library proposes for this launch (1: none): launches that leave most of the chip idle - a 512 -> 512 layer on a
    22x22 map at batch 2 is 96-192 workgroups walking all 512 channels each - run KS workgroups per output tile over Cin / KS channels
    each and a second, element-wise launch adds the partial sums (csrc/ssm_wino.hip: ssm_wino_conv2d_splitk_fwd)."""
    key = (B, H, W, bool(ups))
    cache = pk.__dict__.setdefault("_splitk", {})
    if key not in cache:
        ks = ctypes.c_int(1)
        check(load().ssm_wino_splitk_plan(pk.cin, pk.cout, B, H, W, 1 if ups else 0, pk.bn, ctypes.byref(ks)))
        cache[key] = ks.value
    return cache[key]


def _flags(lrelu, mask, add):
    if mask:
        assert add is not None and not lrelu, "mask epilogue: a mask source and no activation"
        return SSM_FLAG_MASK
    return SSM_FLAG_LRELU if lrelu else 0


def _splitk_launch(x1, c1, x2, c2, pk, y, pool, add, add_div, ks, ups, B, H, W, lrelu, slope, mask=False):
    lib, st = load(), stream_ptr()
    scratch = pk.__dict__.setdefault("_splitk_part", {})          # the partial sums: one tensor per (stream, problem) - plans on other streams run beside this one
    key = (getattr(st, "value", st), ks, B, H, W)
    if key not in scratch:
        scratch[key] = Planes(ks * B, pk.cout, H, W, pk.w.device)
    part = scratch[key]
    check(lib.ssm_wino_conv2d_splitk_fwd(x1, c1, x2 if x2 is not None else NULL_VIEW, c2, pk.w.data_ptr(), pk.b.data_ptr(), part.view(), ks,
                                         1 if ups else 0, B, H, W, pk.cout, pk.bn, st))
    check(lib.ssm_splitk_finish_fwd(part.view(), ks, y, pool if pool is not None else NULL_VIEW, add if add is not None else NULL_VIEW, add_div,
                                    B, pk.cout, H, W, slope, _flags(lrelu, mask, add), st))


def conv2d_wino(x1, c1, x2, c2, pk, y, pool, B, H, W, lrelu=True, slope=0.1, add=None, add_div=1, mask=False):
    """mask: `add` is a mask source - the output is conv(x) * LeakyReLU'(add) (SSM_FLAG_MASK; the training step's data gradients)."""
    lib = load()
    assert pk.cin == c1 + c2, "packed filter expects %d input channels, got %d" % (pk.cin, c1 + c2)
    assert (pk.bn, pk.ck) == wino_plan(c1 + c2, pk.cout, B, H, W, False)[1:], "filter was packed for another tile configuration"
    ks = wino_splitk(pk, B, H, W, False)
    if ks > 1:
        return _splitk_launch(x1, c1, x2, c2, pk, y, pool, add, add_div, ks, False, B, H, W, lrelu, slope, mask)
    check(lib.ssm_wino_conv2d_add_fwd(x1, c1, x2 if x2 is not None else NULL_VIEW, c2, pk.w.data_ptr(), pk.b.data_ptr(), y,
                                      pool if pool is not None else NULL_VIEW, add if add is not None else NULL_VIEW, add_div, B, H, W,
                                      pk.cout, slope, _flags(lrelu, mask, add), stream_ptr()))


def conv2d_ups_wino(a, c1, b, c2, pk, y, B, H, W, lrelu=True, slope=0.1, add=None, add_div=1):
    """conv3x3(upsample2x(cat[a, b])) in the Winograd form: a, b LOW-res padded-plane views, H, W the OUTPUT size."""
    lib = load()
    assert pk.cin == c1 + c2, "packed filter expects %d input channels, got %d" % (pk.cin, c1 + c2)
    assert (pk.bn, pk.ck) == wino_plan(c1 + c2, pk.cout, B, H, W, True)[1:], "filter was packed for another tile configuration"
    ks = wino_splitk(pk, B, H, W, True)
    if ks > 1:
        return _splitk_launch(a, c1, b, c2, pk, y, None, add, add_div, ks, True, B, H, W, lrelu, slope)
    check(lib.ssm_wino_conv2d_ups_add_fwd(a, c1, b if b is not None else NULL_VIEW, c2, pk.w.data_ptr(), pk.b.data_ptr(), y,
                                          add if add is not None else NULL_VIEW, add_div, B, H, W, pk.cout, slope,
                                          SSM_FLAG_LRELU if lrelu else 0, stream_ptr()))


# ---- 3x3 convolution as Winograd F(4x4,3x3) in fp32 (csrc/ssm_wino4.hip) -------------------------------------------------
def wino4_plan(cin, cout, B, H, W, ups=False):
    lib = load()
    kind, bn, ck = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    check(lib.ssm_wino4_plan(cin, cout, B, H, W, 1 if ups else 0, ctypes.byref(kind), ctypes.byref(bn), ctypes.byref(ck)))
    return kind.value, bn.value, ck.value


def wino4_supported(cin, cout, H, W, k=3):
    """Can this layer run as F(4x4,3x3)?  (3x3, whole 4-channel chunks, 32-channel output blocks)"""
    return k == 3 and cin % 4 == 0 and cout % 32 == 0


def wino4_preferred(cin, cout, B, H, W, ups=False):
    """Does the library's cost model put F(4x4,3x3) ahead of F(2x2,3x3) for this problem?"""
    return bool(load().ssm_wino4_preferred(cin, cout, B, H, W, 1 if ups else 0))


class PackedWino4:
    """3x3 filter pre-transformed for F(4x4,3x3) (U = G g G^T, 36 frequencies) and packed [Cout/32][Cin][9][32][4]; an explicit
    handle owned by the Python side like PackedConv."""

    algo = "wino4"

    def __init__(self, weight, bias, B, H, W, pool=False, ups=False):
        require_device(weight, "conv weight")
        require_device(bias, "conv bias")
        self.cout, self.cin, self.k = weight.shape[0], weight.shape[1], weight.shape[2]
        assert self.k == 3 and weight.shape[3] == 3, "F(4x4,3x3) is for 3x3 filters"
        self.ups = bool(ups)
        _, self.bn, self.ck = wino4_plan(self.cin, self.cout, B, H, W, self.ups)
        self.cin_p = self.cin
        lib = load()
        self.w = torch.empty(lib.ssm_wino4_packed_weight_floats(self.cout, self.cin), dtype=torch.float32, device=weight.device)
        self.b = torch.empty(self.cout, dtype=torch.float32, device=weight.device)
        wc, bc = weight.detach().contiguous(), bias.detach().contiguous()
        check(lib.ssm_wino4_pack_weights(wc.data_ptr(), bc.data_ptr(), self.w.data_ptr(), self.b.data_ptr(), self.cout, self.cin, stream_ptr()))


def conv2d_wino4(x1, c1, x2, c2, pk, y, pool, B, H, W, lrelu=True, slope=0.1, add=None, add_div=1, mask=False):
    lib = load()
    assert pk.cin == c1 + c2, "packed filter expects %d input channels, got %d" % (pk.cin, c1 + c2)
    check(lib.ssm_wino4_conv2d_add_fwd(x1, c1, x2 if x2 is not None else NULL_VIEW, c2, pk.w.data_ptr(), pk.b.data_ptr(), y,
                                       pool if pool is not None else NULL_VIEW, add if add is not None else NULL_VIEW, add_div, B, H, W,
                                       pk.cout, slope, _flags(lrelu, mask, add), stream_ptr()))


def conv2d_ups_wino4(a, c1, b, c2, pk, y, B, H, W, lrelu=True, slope=0.1, add=None, add_div=1):
    """conv3x3(upsample2x(cat[a, b])) as F(4x4,3x3): a, b LOW-res padded-plane views, H, W the OUTPUT size."""
    lib = load()
    assert pk.cin == c1 + c2, "packed filter expects %d input channels, got %d" % (pk.cin, c1 + c2)
    check(lib.ssm_wino4_conv2d_ups_add_fwd(a, c1, b if b is not None else NULL_VIEW, c2, pk.w.data_ptr(), pk.b.data_ptr(), y,
                                           add if add is not None else NULL_VIEW, add_div, B, H, W, pk.cout, slope,
                                           SSM_FLAG_LRELU if lrelu else 0, stream_ptr()))


WINO4_BORDER_TH, WINO4_BORDER_TW = 16, 32          # SSM_WINO4_BORDER_TH / _TW of include/ssm_hip.h


def subpixel_wino4_supported(cin, cout, H, W):
    """Can conv3x3(upsample2x(.)) with a H x W OUTPUT run as sub-pixel interior + fused-upsample border ring?  (whole 16 x 32-pixel border
    tiles, an interior of whole 16 x 32 low-res tiles of the 64-cout form, cout a multiple of 32)"""
    ty, tx = H // WINO4_BORDER_TH, W // WINO4_BORDER_TW
    return (cin % 4 == 0 and cout % 32 == 0 and H % WINO4_BORDER_TH == 0 and W % WINO4_BORDER_TW == 0 and ty >= 4 and tx >= 4
            and ((ty - 2) * WINO4_BORDER_TH // 2) % 16 == 0 and ((tx - 2) * WINO4_BORDER_TW // 2) % 32 == 0)


class PackedSubpixelWino4:
    """conv3x3(upsample2x(cat[a, b])) with a H x W output as F(4x4,3x3) in the sub-pixel form (include/ssm_hip.h, ssm_wino4_conv2d_shuffle_fwd):
    the effective filters E[4 c + 2 pa + pb] = M_pa W_c M_pb^T of the interior (ssm_amd.subpixel.effective_filter, float64 then fp32) packed
    like any F(4x4,3x3) filter, beside the layer's ordinary fused-upsample filter for the border ring."""

    algo = "wino4"

    def __init__(self, weight, bias, B, H, W):
        from .subpixel import effective_filter
        self.cout, self.cin, self.k = weight.shape[0], weight.shape[1], 3
        assert subpixel_wino4_supported(self.cin, self.cout, H, W), "sub-pixel F(4x4,3x3): unsupported shape"
        self.border = PackedWino4(weight, bias, B, H, W, ups=True)
        e = effective_filter(weight.detach().to(torch.float32), "int", "int")                    # [(2 pa + pb) Co + c]
        e = e.view(4, self.cout, self.cin, 3, 3).permute(1, 0, 2, 3, 4).reshape(4 * self.cout, self.cin, 3, 3).contiguous()
        self.inner = PackedWino4(e, bias.detach().to(torch.float32).repeat_interleave(4).contiguous(), B, H // 2, W // 2)
        self.w, self.b = self.border.w, self.border.b          # (the handles the plan's bookkeeping looks at)
        self.ups, self.cin_p, self.bn, self.ck = True, self.cin, self.border.bn, self.border.ck


def conv2d_ups_subpixel_wino4(a_view, c1, b_view, c2, pk, y_view, B, H, W, lrelu=True, slope=0.1):
    """a_view / b_view / y_view: callables (y0, x0) -> ssm_view of the LOW-res sources / the output at that pixel (views at two origins are
    needed: the interior's and the map's); b_view None: one source."""
    lib = load()
    flags = SSM_FLAG_LRELU if lrelu else 0
    y0, x0 = WINO4_BORDER_TH // 2, WINO4_BORDER_TW // 2                       # low-res origin of the interior
    h, w = H // 2 - 2 * y0, W // 2 - 2 * x0
    check(lib.ssm_wino4_conv2d_shuffle_fwd(a_view(y0, x0), c1, b_view(y0, x0) if b_view is not None else NULL_VIEW, c2, pk.inner.w.data_ptr(),
                                           pk.inner.b.data_ptr(), y_view(2 * y0, 2 * x0), B, h, w, 4 * pk.cout, slope, flags, stream_ptr()))
    check(lib.ssm_wino4_conv2d_ups_border_fwd(a_view(0, 0), c1, b_view(0, 0) if b_view is not None else NULL_VIEW, c2, pk.border.w.data_ptr(),
                                              pk.border.b.data_ptr(), y_view(0, 0), B, H, W, pk.cout, slope, flags, stream_ptr()))


# ---- 7x7 / 5x5 convolutions as 1-D Winograd along x, F(2,7) / F(4,5), in fp32 (csrc/ssm_wino1d.hip) -----------------
def wino1d_plan(k, cin, cout, B, H, W):
    """(kind, BN, CK) of the 1-D Winograd tile configuration for the problem."""
    lib = load()
    kind, bn, ck = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    check(lib.ssm_wino1d_plan(k, cin, cout, B, H, W, ctypes.byref(kind), ctypes.byref(bn), ctypes.byref(ck)))
    return kind.value, bn.value, ck.value


def wino1d_supported(cin, cout, H, W, k):
    """Can this layer run in the 1-D Winograd form?  (7x7 with 32-channel output blocks, 5x5 with 32 / 64)"""
    return k in (5, 7) and cout % 32 == 0


class PackedWino1d:
    """7x7 / 5x5 filter with its rows pre-transformed (U[ky] = G g[ky]) and packed for the 1-D Winograd kernel's tile
    configuration; an explicit handle owned by the Python side like PackedConv."""

    algo = "wino1d"

    def __init__(self, weight, bias, B, H, W, pool=False, ups=False):
        require_device(weight, "conv weight")
        require_device(bias, "conv bias")
        assert not ups, "the 7x7 / 5x5 layers have no fused-upsample form"
        self.cout, self.cin, self.k = weight.shape[0], weight.shape[1], weight.shape[2]
        assert self.k in (5, 7) and weight.shape[3] == self.k, "the 1-D Winograd form is for 7x7 and 5x5 filters"
        self.ups = False
        _, self.bn, self.ck = wino1d_plan(self.k, self.cin, self.cout, B, H, W)
        self.cin_p = (self.cin + self.ck - 1) // self.ck * self.ck
        lib = load()
        nw = lib.ssm_wino1d_packed_weight_floats(self.cout, self.cin_p, self.k, self.bn)
        nb = lib.ssm_packed_bias_floats(self.cout, self.bn)
        self.w = torch.empty(nw, dtype=torch.float32, device=weight.device)
        self.b = torch.empty(nb, dtype=torch.float32, device=weight.device)
        wc, bc = weight.detach().contiguous(), bias.detach().contiguous()
        check(lib.ssm_wino1d_pack_weights(wc.data_ptr(), bc.data_ptr(), self.w.data_ptr(), self.b.data_ptr(), self.cout, self.cin,
                                          self.cin_p, self.k, self.bn, stream_ptr()))


def conv2d_wino1d(x1, c1, x2, c2, pk, y, pool, B, H, W, lrelu=True, slope=0.1, add=None, add_div=1):
    """Same call shape as conv2d / conv2d_wino; these layers take one source (x2 must be None)."""
    lib = load()
    assert x2 is None and c2 == 0, "the 1-D Winograd layers have no concatenated source"
    assert pk.cin_p == c1, "packed filter expects %d input channels, got %d" % (pk.cin_p, c1)
    assert (pk.bn, pk.ck) == wino1d_plan(pk.k, c1, pk.cout, B, H, W)[1:], "filter was packed for another tile configuration"
    check(lib.ssm_wino1d_conv2d_add_fwd(x1, c1, pk.w.data_ptr(), pk.b.data_ptr(), y, pool if pool is not None else NULL_VIEW,
                                        add if add is not None else NULL_VIEW, add_div, B, H, W, pk.cout, pk.k, slope,
                                        SSM_FLAG_LRELU if lrelu else 0, stream_ptr()))


# ---- 5x5 convolutions as two-dimensional Winograd F(4x4,5x5) in fp32 (csrc/ssm_wino5.hip) -----------------------------
def wino5_supported(cin, cout, H, W, k):
    """Can this layer run in the two-dimensional F(4x4,5x5) form?  (5x5, 32-channel output blocks; channels padded to 4)"""
    return k == 5 and cout % 32 == 0


class PackedWino5:
    """5x5 filter pre-transformed on both axes (U = G g G^T, 64 frequencies) and packed for csrc/ssm_wino5.hip; an explicit handle
    owned by the Python side like PackedConv."""

    algo = "wino5"

    def __init__(self, weight, bias, B, H, W, pool=False, ups=False):
        require_device(weight, "conv weight")
        require_device(bias, "conv bias")
        assert not ups, "the 5x5 layers have no fused-upsample form"
        self.cout, self.cin, self.k = weight.shape[0], weight.shape[1], weight.shape[2]
        assert self.k == 5 and weight.shape[3] == 5 and self.cout % 32 == 0, "the F(4x4,5x5) form is for 5x5 filters, Cout a multiple of 32"
        self.ups = False
        self.bn, self.ck = 32, 4
        self.cin_p = (self.cin + 3) // 4 * 4
        lib = load()
        self.w = torch.empty(lib.ssm_wino5_packed_weight_floats(self.cout, self.cin_p), dtype=torch.float32, device=weight.device)
        self.b = torch.empty(self.cout, dtype=torch.float32, device=weight.device)
        wc, bc = weight.detach().contiguous(), bias.detach().contiguous()
        check(lib.ssm_wino5_pack_weights(wc.data_ptr(), bc.data_ptr(), self.w.data_ptr(), self.b.data_ptr(), self.cout, self.cin, self.cin_p,
                                         stream_ptr()))


def conv2d_wino5(x1, c1, x2, c2, pk, y, pool, B, H, W, lrelu=True, slope=0.1, add=None, add_div=1, mask=False):
    """Same call shape as conv2d / conv2d_wino1d; these layers take one source (x2 must be None).  mask: as conv2d_wino (SSM_FLAG_MASK)."""
    lib = load()
    assert x2 is None and c2 == 0, "the 5x5 layers have no concatenated source"
    assert pk.cin_p == c1, "packed filter expects %d input channels, got %d" % (pk.cin_p, c1)
    check(lib.ssm_wino5_conv2d_add_fwd(x1, c1, pk.w.data_ptr(), pk.b.data_ptr(), y, pool if pool is not None else NULL_VIEW,
                                       add if add is not None else NULL_VIEW, add_div, B, H, W, pk.cout, slope,
                                       _flags(lrelu, mask, add), stream_ptr()))


# ---- 7x7 convolutions as 2x2 blocks of F(4x4,4x4) Winograd filters in fp32 (csrc/ssm_wino7.hip) ---------------------
def wino7_supported(cin, cout, H, W, k):
    """Can this layer run in the blocked two-dimensional Winograd form?  (7x7, 32-channel output blocks; any Cin)"""
    return k == 7 and cout % 32 == 0


class PackedWino7:
    """7x7 filter as four pre-transformed 4x4 blocks (U_b = G g_b G^T) packed for csrc/ssm_wino7.hip; an explicit handle owned by the
    Python side like PackedConv."""

    algo = "wino7"

    def __init__(self, weight, bias, B, H, W, pool=False, ups=False):
        require_device(weight, "conv weight")
        require_device(bias, "conv bias")
        assert not ups, "the 7x7 layers have no fused-upsample form"
        self.cout, self.cin, self.k = weight.shape[0], weight.shape[1], weight.shape[2]
        assert self.k == 7 and weight.shape[3] == 7, "the blocked Winograd form is for 7x7 filters"
        # cout_p: the channels the convolution writes (whole 32-channel blocks; zeros beyond cout - the output view must hold them)
        self.cout_p = (self.cout + 31) // 32 * 32
        self.ups = False
        self.bn, self.ck, self.cin_p = 32, 1, self.cin
        lib = load()
        self.w = torch.empty(lib.ssm_wino7_packed_weight_floats(self.cout, self.cin), dtype=torch.float32, device=weight.device)
        self.b = torch.empty(self.cout_p, dtype=torch.float32, device=weight.device)
        wc, bc = weight.detach().contiguous(), bias.detach().contiguous()
        check(lib.ssm_wino7_pack_weights(wc.data_ptr(), bc.data_ptr(), self.w.data_ptr(), self.b.data_ptr(), self.cout, self.cin, stream_ptr()))


def conv2d_wino7(x1, c1, x2, c2, pk, y, pool, B, H, W, lrelu=True, slope=0.1, add=None, add_div=1):
    """Same call shape as conv2d / conv2d_wino1d; these layers take one source (x2 must be None)."""
    lib = load()
    assert x2 is None and c2 == 0, "the 7x7 layers have no concatenated source"
    assert pk.cin == c1, "packed filter expects %d input channels, got %d" % (pk.cin, c1)
    check(lib.ssm_wino7_conv2d_add_fwd(x1, c1, pk.w.data_ptr(), pk.b.data_ptr(), y, pool if pool is not None else NULL_VIEW,
                                       add if add is not None else NULL_VIEW, add_div, B, H, W, pk.cout_p, slope,
                                       SSM_FLAG_LRELU if lrelu else 0, stream_ptr()))


# ---- HL8 (fp16 hi/lo) tensors and the fp16-MFMA convolution ------------------------------------
class HPlanes:
    """A [B,C,H,W] activation in the HL8 layout (include/ssm_hip.h): [B][G][hi|lo][Hp][Wp][8 x fp16],
    G = ceil(C/8) channel groups (optionally more: zero channels), zero frame, tail slack."""

    def __init__(self, B, C, H, W, device, groups=None, q8=False):
        self.B, self.C, self.H, self.W = B, C, H, W
        self.q8 = bool(q8)       # plane 1 = [fp8(x) | fp8(lo * 2^11)] (SSM_FLAG_Q8 operands) instead of fp16(lo)
        self.G = groups if groups is not None else (C + 7) // 8
        if q8 and self.G % 2:
            self.G += 1          # second planes are shared by pairs of groups
        self.Hp, self.Wp = plane_dims(H, W)
        n = B * self.G * 2 * self.Hp * self.Wp * 8
        self.buf = torch.zeros(n + 2 * SSM_TAIL_SLACK_FLOATS, dtype=torch.float16, device=device)

    def view(self, g0=0, broadcast=False, b0=0, y0=0, x0=0):
        """ssm_hview of channel groups g0.. of batch entries b0.. (broadcast: every batch index reads entry b0), starting at
        pixel (y0, x0)."""
        pix = self.Hp * self.Wp
        base = self.buf.data_ptr() + 16 * ((b0 * self.G + g0) * 2 * pix + (SSM_PADY + y0) * self.Wp + SSM_PADX + x0)
        return SsmHView(base, 0 if broadcast else self.G * 2 * pix, 2 * pix, pix, self.Wp)

    def load(self, x):
        x = x if x.stride(3) == 1 else x.contiguous()
        fn = load().ssm_hq8_from_f32 if self.q8 else load().ssm_hl8_from_f32
        check(fn(view_of(x), self.view(), self.B, self.C, self.G, self.H, self.W, stream_ptr()))
        return self

    def to_nchw(self):
        out = torch.empty(self.B, self.C, self.H, self.W, dtype=torch.float32, device=self.buf.device)
        fn = load().ssm_hq8_to_f32 if self.q8 else load().ssm_hl8_to_f32
        check(fn(self.view(), view_of(out), self.B, self.C, self.G, self.H, self.W, stream_ptr()))
        return out


class PackedConv16:
    """Filter split into fp16 hi/lo parts, scaled by a power of two so both parts sit in fp16's
    normal range, repacked for the fp16-MFMA kernel's tile configuration."""

    def __init__(self, weight, bias, W, q8=False, ups=False, scale=None, shape=None, device=None):
        """ups: the filter feeds ssm_conv2d_ups_hl8_fwd (W = its OUTPUT width); in Q8 form the packing follows that kernel's tile.
        scale: reuse a power-of-two pre-scale chosen earlier (training repacks every step; choosing it needs max|w| on the host).
        weight=None with shape=(cout, cin, k), device and scale: buffers only, filled later by a PackBatch."""
        if weight is None:
            assert q8 and scale is not None and shape is not None and device is not None
            self.cout, self.cin, self.k = shape
        else:
            require_device(weight, "conv weight")
            self.cout, self.cin, self.k = weight.shape[0], weight.shape[1], weight.shape[2]
            device = weight.device
        self.q8 = bool(q8)
        lib = load()
        bn, kys = ctypes.c_int(0), ctypes.c_int(0)
        if q8 and ups:
            check(lib.ssm_conv16q_ups_config(self.cout, W, ctypes.byref(bn), ctypes.byref(kys)))
        else:
            check((lib.ssm_conv16q_config if q8 else lib.ssm_conv16_config)(self.k, self.cout, W, ctypes.byref(bn), ctypes.byref(kys)))
        self.bn, self.kys = bn.value, kys.value
        self.cin_p = (self.cin + 15) // 16 * 16
        if scale is None:
            wmax = float(weight.detach().abs().max())
            import math
            scale = 2.0 ** (3 - math.ceil(math.log2(wmax))) if wmax > 0 else 1.0     # max|w|*scale in (4, 8]
        self.scale = float(scale)
        nb = lib.ssm_packed_bias_floats(self.cout, self.bn)
        self.b = torch.empty(nb, dtype=torch.float32, device=device)
        if weight is None:
            nbytes = lib.ssm_packed16q_weight_bytes(self.cout, self.cin_p, self.k, self.bn, self.kys)
            self.w = torch.empty(nbytes, dtype=torch.uint8, device=device)
            return
        wc, bc = weight.detach().contiguous(), bias.detach().contiguous()
        if q8:
            nbytes = lib.ssm_packed16q_weight_bytes(self.cout, self.cin_p, self.k, self.bn, self.kys)
            self.w = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
            check(lib.ssm_pack16q_weights(wc.data_ptr(), bc.data_ptr(), self.w.data_ptr(), self.b.data_ptr(), self.cout,
                                          self.cin, self.cin_p, self.k, self.bn, self.kys, self.scale, stream_ptr()))
            return
        nh = lib.ssm_packed16_weight_halves(self.cout, self.cin_p, self.k, self.bn)
        self.w = torch.empty(nh, dtype=torch.float16, device=weight.device)
        check(lib.ssm_pack16_weights(wc.data_ptr(), bc.data_ptr(), self.w.data_ptr(), self.b.data_ptr(), self.cout,
                                     self.cin, self.cin_p, self.k, self.bn, self.kys, self.scale, stream_ptr()))


class SsmSubpixelProblem(ctypes.Structure):
    """ssm_subpixel_problem (include/ssm_hip.h)."""
    _fields_ = [("x1", SsmHView), ("C1", ctypes.c_int), ("x2", SsmHView), ("C2", ctypes.c_int), ("w_packed", ctypes.c_void_p),
                ("bias_packed", ctypes.c_void_p), ("inv_wscale", ctypes.c_float), ("y_hl8", SsmHView), ("H", ctypes.c_int),
                ("W", ctypes.c_int), ("transposed", ctypes.c_int), ("skip_y", ctypes.c_int), ("skip_x", ctypes.c_int)]


class SsmPackJob(ctypes.Structure):
    """ssm_pack16q_job (include/ssm_hip.h)."""
    _fields_ = [("w", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("wp", ctypes.c_void_p), ("bp", ctypes.c_void_p),
                ("Cout", ctypes.c_int), ("Cin", ctypes.c_int), ("CinP", ctypes.c_int), ("k", ctypes.c_int), ("BN", ctypes.c_int),
                ("KYS", ctypes.c_int), ("scale", ctypes.c_float), ("transposed", ctypes.c_int), ("block_start", ctypes.c_int),
                ("row_blocks", ctypes.c_int)]


class PackBatch:
    """Every Q8 filter of a U-Net repacked by one launch (ssm_pack16q_weights_batch).  entries: (PackedConv16 with q8 buffers,
    fp32 OIHW weight tensor on the device, bias tensor or None, transposed) - transposed packs the data-gradient filter of the
    forward weight.  The job table holds raw pointers: rebuild it when `key()` of the parameters changes."""

    def __init__(self, entries, device):
        lib = load()
        jobs = (SsmPackJob * len(entries))()
        off = 0
        self.keep = []
        for j, (pk, w, b, transposed) in zip(jobs, entries):
            assert pk.q8 and w.is_contiguous() and w.dtype == torch.float32 and w.device == pk.w.device
            assert tuple(w.shape) == ((pk.cin, pk.cout, pk.k, pk.k) if transposed else (pk.cout, pk.cin, pk.k, pk.k))
            rb, bb = ctypes.c_int(0), ctypes.c_int(0)
            check(lib.ssm_pack16q_job_blocks(pk.cout, pk.cin_p, pk.k, pk.bn, ctypes.byref(rb), ctypes.byref(bb)))
            j.w, j.bias, j.wp, j.bp = w.data_ptr(), (b.data_ptr() if b is not None else None), pk.w.data_ptr(), pk.b.data_ptr()
            j.Cout, j.Cin, j.CinP, j.k, j.BN, j.KYS = pk.cout, pk.cin, pk.cin_p, pk.k, pk.bn, pk.kys
            j.scale, j.transposed, j.block_start, j.row_blocks = pk.scale, 1 if transposed else 0, off, rb.value
            off += rb.value + bb.value
            self.keep.append((pk, w, b))
        self.n, self.total = len(entries), off
        self.table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(device)

    @staticmethod
    def key(tensors):
        return tuple((t.data_ptr(), tuple(t.shape)) for t in tensors)

    def run(self):
        check(load().ssm_pack16q_weights_batch(self.table.data_ptr(), self.n, self.total, stream_ptr()))


class SsmPack32Job(ctypes.Structure):
    """ssm_pack32_job (include/ssm_hip.h)."""
    _fields_ = [("w", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("wp", ctypes.c_void_p), ("bp", ctypes.c_void_p),
                ("Cout", ctypes.c_int), ("Cin", ctypes.c_int), ("CinP", ctypes.c_int), ("k", ctypes.c_int), ("BN", ctypes.c_int),
                ("algo", ctypes.c_int), ("transposed", ctypes.c_int), ("nbias", ctypes.c_int), ("first", ctypes.c_longlong),
                ("total", ctypes.c_longlong)]


class PackBatch32:
    """Every fp32 filter of a U-Net repacked by one launch (ssm_pack32_weights_batch).  entries: (PackedConv / PackedWino /
    PackedWino1d / PackedWino4 whose buffers are refilled, fp32 OIHW parameter on the device, bias tensor or None, transposed) -
    transposed packs the data-gradient filter of the forward parameter (no torch flip / permute / copy).  The job table holds raw
    pointers: rebuild it when `PackBatch.key()` of the parameters changes."""

    ALGO = {"direct": 0, "wino": 1, "wino1d": 2, "wino4": 3, "wino7": 4, "wino5": 5}

    def __init__(self, entries, device):
        # F(2x2,3x3), direct 3x3 and (r6) F(4x4,3x3) jobs with whole tiles go to the tiled kernel (ssm_pack32_wino_tiles_batch: contiguous reads and writes), the rest to
        # the element-wise one; $SSM_PACK_TILES=0: everything element-wise
        tiled = [e for e in entries if self._tiled(e[0])] if os.environ.get("SSM_PACK_TILES", "1") != "0" else []
        entries = [e for e in entries if not any(e is t for t in tiled)]
        self.tiles = None
        if tiled:
            tj = (SsmPack32Job * len(tiled))()
            toff = 0
            for j, (pk, w, b, transposed) in zip(tj, tiled):
                assert w.is_contiguous() and w.dtype == torch.float32 and w.device == pk.w.device
                assert tuple(w.shape) == ((pk.cin, pk.cout, 3, 3) if transposed else (pk.cout, pk.cin, 3, 3))
                j.w, j.bias, j.wp, j.bp = w.data_ptr(), (b.data_ptr() if b is not None else None), pk.w.data_ptr(), pk.b.data_ptr()
                bn = self._tile_bn(pk)
                j.Cout, j.Cin, j.CinP, j.k, j.BN = pk.cout, pk.cin, pk.cin_p, 3, bn
                j.algo, j.transposed, j.nbias = self.ALGO[pk.algo], 1 if transposed else 0, pk.b.numel()
                j.first, j.total = toff, (pk.cout // bn) * (pk.cin // 16)
                toff += j.total
            self.tiles = (torch.frombuffer(bytearray(bytes(tj)), dtype=torch.uint8).to(device), len(tiled), toff, max(self._tile_bn(e[0]) for e in tiled))
            self.keep_tiled = list(tiled)
        jobs = (SsmPack32Job * max(len(entries), 1))()
        off = 0
        self.keep = []
        for j, (pk, w, b, transposed) in zip(jobs, entries):
            assert w.is_contiguous() and w.dtype == torch.float32 and w.device == pk.w.device
            assert tuple(w.shape) == ((pk.cin, pk.cout, pk.k, pk.k) if transposed else (pk.cout, pk.cin, pk.k, pk.k)), \
                "parameter %s does not match the packed filter (%d -> %d, k %d)" % (tuple(w.shape), pk.cin, pk.cout, pk.k)
            j.w, j.bias, j.wp, j.bp = w.data_ptr(), (b.data_ptr() if b is not None else None), pk.w.data_ptr(), pk.b.data_ptr()
            j.Cout, j.Cin, j.CinP, j.k, j.BN = pk.cout, pk.cin, pk.cin_p, pk.k, pk.bn
            j.algo, j.transposed, j.nbias = self.ALGO[pk.algo], 1 if transposed else 0, pk.b.numel()
            j.first, j.total = off, pk.w.numel()
            # the kernel stores 4 packed elements per thread: whole quads per job, 16-byte aligned buffers
            assert pk.w.numel() % 4 == 0 and pk.bn % 4 == 0 and pk.w.data_ptr() % 16 == 0, "packed filter is not made of whole quads"
            off += (max(pk.w.numel(), pk.b.numel()) + 3) // 4 * 4
            self.keep.append((pk, w, b))
        self.n, self.total = len(entries), off
        self.table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(device)

    @staticmethod
    def _tile_bn(pk):
        """Cout block of a job of the tiled kernel: F(4x4,3x3) filters are packed in 32-cout blocks whatever tile configuration launches them."""
        return 32 if getattr(pk, "algo", "direct") == "wino4" else pk.bn

    @classmethod
    def _tiled(cls, pk):
        per = {"wino": 16, "direct": 9, "wino4": 36}.get(getattr(pk, "algo", "direct"))          # packed floats per (cout, cin)
        if per is None or type(pk).__name__ == "PackedSubpixelWino4":
            return False
        bn = cls._tile_bn(pk)
        return (pk.k == 3 and bn in (32, 64) and pk.cout % bn == 0 and pk.cin % 16 == 0 and pk.cin_p == pk.cin
                and pk.w.numel() == (pk.cout // bn) * pk.cin * per * bn)

    def run(self):
        lib, st = load(), stream_ptr()
        if self.n:
            check(lib.ssm_pack32_weights_batch(self.table.data_ptr(), self.n, self.total, st))
        if self.tiles is not None:
            check(lib.ssm_pack32_wino_tiles_batch(self.tiles[0].data_ptr(), self.tiles[1], self.tiles[2], self.tiles[3], st))


class SsmWgradwFinishJob(ctypes.Structure):
    """ssm_wgradw_finish_job (include/ssm_hip.h)."""
    _fields_ = [("du", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("n", ctypes.c_int), ("pad_", ctypes.c_int)]


def wgrad_wino_supported(cin, cout, H, W, k):
    return bool(load().ssm_wgrad_wino_supported(cin, cout, H, W, k))


def wgrad_wino(x_view, dz_view, du, bias_acc, B, cin, cout, H, W, cin_total, ci_offset):
    """du [16, cout, cin_total] (fp32, zero before a step's first launch) += the Winograd-domain partial sums of one source."""
    assert du.is_contiguous() and du.dtype == torch.float32 and du.numel() == 16 * cout * cin_total
    check(load().ssm_conv2d_wgrad_wino(x_view, dz_view, du.data_ptr(), bias_acc.data_ptr() if bias_acc is not None else None, B, cin, cout,
                                       H, W, cin_total, ci_offset, stream_ptr()))


class WgradWinoFinish:
    """One finishing launch for several layers: dw += scale * G^T du G, du := 0 (ssm_wgrad_wino_finish).  entries: (du [16,Cout,Cin]
    scratch, dw [Cout,Cin,3,3] gradient slice) - raw pointers, static for the life of the training plan."""

    def __init__(self, entries, device):
        jobs = (SsmWgradwFinishJob * len(entries))()
        self.keep = list(entries)
        self.max_n = 0
        for j, (du, dw) in zip(jobs, entries):
            assert dw.is_contiguous() and du.is_contiguous() and dw.dim() == 4 and tuple(dw.shape[2:]) == (3, 3)
            n = dw.shape[0] * dw.shape[1]
            assert du.numel() == 16 * n
            j.du, j.dw, j.n = du.data_ptr(), dw.data_ptr(), n
            self.max_n = max(self.max_n, n)
        self.n = len(entries)
        self.table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(device)

    def run(self, scale=1.0):
        check(load().ssm_wgrad_wino_finish(self.table.data_ptr(), self.n, self.max_n, float(scale), stream_ptr()))


def conv2d_hl8(x1, c1, x2, c2, pk, y_hl8, y_f32, pool, B, H, W, lrelu=True, slope=0.1, fast=False):
    assert pk.cin_p == c1 + c2, "packed filter expects %d input channels, got %d" % (pk.cin_p, c1 + c2)
    flags = (SSM_FLAG_LRELU if lrelu else 0) | (SSM_FLAG_FP16_FAST if fast else 0) | (SSM_FLAG_Q8 if pk.q8 else 0)
    check(load().ssm_conv2d_hl8_fwd(x1, c1, x2 if x2 is not None else NULL_HVIEW, c2, pk.w.data_ptr(), pk.b.data_ptr(),
                                    1.0 / pk.scale, y_hl8 if y_hl8 is not None else NULL_HVIEW,
                                    y_f32 if y_f32 is not None else NULL_VIEW, pool if pool is not None else NULL_HVIEW,
                                    B, H, W, pk.cout, pk.k, slope, flags, stream_ptr()))


def conv2d_ups_hl8(a, c1, b, c2, pk, y_hl8, y_f32, B, H, W, lrelu=True, slope=0.1, fast=False):
    """conv3x3(upsample2x(cat[a, b])): a, b LOW-res HL8 views, H, W the OUTPUT size."""
    assert pk.k == 3 and pk.cin_p == c1 + c2, "packed 3x3 filter expects %d input channels, got %d" % (pk.cin_p, c1 + c2)
    flags = (SSM_FLAG_LRELU if lrelu else 0) | (SSM_FLAG_FP16_FAST if fast else 0) | (SSM_FLAG_Q8 if pk.q8 else 0)
    check(load().ssm_conv2d_ups_hl8_fwd(a, c1, b if b is not None else NULL_HVIEW, c2, pk.w.data_ptr(), pk.b.data_ptr(),
                                        1.0 / pk.scale, y_hl8 if y_hl8 is not None else NULL_HVIEW,
                                        y_f32 if y_f32 is not None else NULL_VIEW, B, H, W, pk.cout, slope, flags,
                                        stream_ptr()))
