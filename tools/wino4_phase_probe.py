#!/usr/bin/env python3
"""Where does a wino4 workgroup spend its cycles?  Diagnostics build only (make wabl; SSM_HIP_LIB=tools/wabl_libssm_hip.so,
SSM_WINO4_ABL=32): wave 0 of every workgroup sums s_memtime deltas per phase - wait + top barrier | expand + transform | mid barrier |
matrix loop | epilogue - and the kernel adds them to 7 device counters.  usage: python tools/wino4_phase_probe.py [B] [layer ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
os.environ.setdefault("SSM_HIP_LIB", os.path.join(ROOT, "tools", "wabl_libssm_hip.so"))
os.environ["SSM_WINO4_ABL"] = str(int(os.environ.get("SSM_WINO4_ABL", "0")) | 32)
import torch  # noqa: E402

from ssm_amd import hipbind as hb  # noqa: E402
from ssm_amd.engine import UNetPlan, layer_scale  # noqa: E402
from ssm_amd.weights import unet_layers  # noqa: E402

CAT = {"conv8a": 512, "conv9a": 256, "conv10a": 128, "conv11a": 64, "conv7a": 512, "fuse_conv": 32}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    names = sys.argv[2:] or ["conv11a", "conv11b", "fuse_conv", "conv9a", "conv5b"]
    dev = torch.device("cuda:0")
    lib = hb.load()
    lib.ssm_wino4_debug_buffer.argtypes = [ctypes.c_void_p]
    cnt = torch.zeros(16, dtype=torch.int64, device=dev)
    lib.ssm_wino4_debug_buffer(ctypes.c_void_p(cnt.data_ptr()))
    lib.ssm_wino4_force_kind(int(os.environ.get("W4KIND", "-1")))          # tile configuration (csrc/ssm_wino4.hip: SSM_W4_KINDS)
    H, W = 736, 1280
    print("%-10s %6s | per workgroup (wave 0), shader cycles: %9s %9s %9s %9s %9s | %9s | chunks" % (
        "layer", "ms", "wait+bar", "transform", "mid bar", "matrix", "epilogue", "total"))
    for name, cin, cout, k in unet_layers(2, True):
        if name not in names:
            continue
        s = layer_scale(name)
        h, w = H // s, W // s
        ups = name in UNetPlan.UPS
        c1 = CAT.get(name, cin)
        c2 = cin - c1
        hs, ws = (h // 2, w // 2) if ups else (h, w)
        xa = hb.Planes(B, c1, hs, ws, dev)
        xa.interior.normal_()
        xb = None
        if c2:
            xb = hb.Planes(B, c2, hs, ws, dev)
            xb.interior.normal_()
        y = hb.Planes(B, cout, h, w, dev)
        wt = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        bs = torch.randn(cout, device=dev) * 0.1
        pw = hb.PackedWino4(wt, bs, B, h, w, ups=ups)
        bv = xb.view() if xb is not None else None
        if ups:
            f = lambda: hb.conv2d_ups_wino4(xa.view(), c1, bv, c2, pw, y.view(), B, h, w)  # noqa: E731
        else:
            f = lambda: hb.conv2d_wino4(xa.view(), c1, bv, c2, pw, y.view(), None, B, h, w)  # noqa: E731
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        cnt.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        c = [int(v) for v in cnt.cpu()]
        n = max(c[6], 1)
        nc = cin // 4
        print("%-10s %6.3f | %44s %9d %9d %9d %9d %9d | %9d | %d   per chunk: %d %d %d %d = %d" % (
            name, e0.elapsed_time(e1), "", c[0] // n, c[1] // n, c[2] // n, c[3] // n, c[4] // n, c[5] // n, nc,
            c[0] // n // nc, c[1] // n // nc, c[2] // n // nc, c[3] // n // nc, (c[0] + c[1] + c[2] + c[3]) // n // nc) + (
            "   transform: reads %d arithmetic %d stores %d" % (c[8] // n // nc, c[9] // n // nc, c[10] // n // nc) if c[8] else ""), flush=True)


if __name__ == "__main__":
    main()
