#!/usr/bin/env python3
"""Diagnostic: shader clock while one conv layer runs back to back (is the matrix pipe clock- or issue-limited?).
usage: python tools/clock_under_load.py [--f32 | --wino] [--b N] [layer ...]   (default: a few stage-2 layers at B=7, 736x1280;
--f32: the fp32-MFMA kernel of ssm_conv.hip instead of the fp16 split kernels; --wino: the Winograd F(2x2,3x3) fp32 kernels of
ssm_wino.hip on the 3x3 layers)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from ssm_amd import hipbind as hb  # noqa: E402
from ssm_amd.engine import layer_scale  # noqa: E402
from ssm_amd.weights import unet_layers  # noqa: E402

probe = ctypes.CDLL(os.path.join(ROOT, "tools", "libclockprobe.so"))
probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p]


def measure(fn, ms=30.0):
    dev = torch.device("cuda:0")
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    probe.clock_probe_launch(out.data_ptr(), int(ms * 1e5), ctypes.c_void_p(side.cuda_stream))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 0
    while True:
        for _ in range(20):
            fn()
        n += 20
        e1.record()
        e1.synchronize()
        if e0.elapsed_time(e1) > ms * 1.3:
            break
    torch.cuda.synchronize()
    c, r = [int(v) for v in out.cpu()]
    return c / (r / 100e6) / 1e9, e0.elapsed_time(e1) / n


def main():
    argv = sys.argv[1:]
    f32 = "--f32" in argv
    wino = "--wino" in argv
    argv = [a for a in argv if a not in ("--f32", "--wino")]
    B = 7
    if "--b" in argv:
        i = argv.index("--b")
        B = int(argv[i + 1])
        del argv[i:i + 2]
    names = argv or ["conv1b", "conv2b", "conv4b", "conv8b", "conv9b", "conv10b", "conv11b"]
    H, W = 736, 1280
    dev = torch.device("cuda:0")
    ghz, _ = measure(lambda: None, 10.0)
    print("idle: %.3f GHz" % ghz)
    for name, cin, cout, k in unet_layers(2, True):
        if name not in names:
            continue
        s = layer_scale(name)
        h, w = H // s, W // s
        if wino:
            if k != 3 or cout < 32:
                continue
            pk = hb.PackedWino(torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5, torch.zeros(cout, device=dev), B, h, w)
            x = hb.Planes(B, cin, h, w, dev)
            x.interior.normal_()
            y = hb.Planes(B, cout, h, w, dev)
            fn = lambda: hb.conv2d_wino(x.view(), cin, None, 0, pk, y.view(), None, B, h, w)  # noqa: E731
            ghz, ms = measure(fn)
            gf = 2.0 * B * h * w * cout * cin * 9 / 1e9
            iss = gf / ms * 16.0 / 36.0
            print("%-8s wino   %.3f ms  %6.1f TF algorithmic  %6.1f TF issued  clock %.3f GHz  -> MFMA busy %.0f %% of the pipe at that clock"
                  % (name, ms, gf / ms, iss, ghz, 100 * iss / (157.3 * ghz / 2.4)))
            continue
        if f32:
            pk = hb.PackedConv(torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5, torch.zeros(cout, device=dev), B, h, w)
            x = hb.Planes(B, cin, h, w, dev)
            x.interior.normal_()
            y = hb.Planes(B, cout, h, w, dev)
            fn = lambda: hb.conv2d(x.view(), cin, None, 0, pk, y.view(), None, B, h, w)  # noqa: E731
            ghz, ms = measure(fn)
            gf = 2.0 * B * h * w * cout * cin * k * k / 1e9
            print("%-8s f32    %.3f ms  %6.1f TF  clock %.3f GHz  -> MFMA busy %.0f %% of the pipe at that clock"
                  % (name, ms, gf / ms, ghz, 100 * (gf / ms) / (157.3 * ghz / 2.4)))
            continue
        pk = hb.PackedConv16(torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5, torch.zeros(cout, device=dev), w)
        x = hb.HPlanes(B, cin, h, w, dev, groups=pk.cin_p // 8)
        x.buf.normal_()
        y = hb.HPlanes(B, cout, h, w, dev)
        for fast in (False, True):
            fn = lambda: hb.conv2d_hl8(x.view(), pk.cin_p, None, 0, pk, y.view(), None, None, B, h, w, fast=fast)  # noqa: E731
            ghz, ms = measure(fn)
            gf = 2.0 * B * h * w * cout * cin * k * k / 1e9
            issued = gf / ms * (1 if fast else 3)
            print("%-8s %-6s %.3f ms  %6.1f TF algorithmic  %7.1f TF issued  clock %.3f GHz  -> MFMA busy %.0f %% of the pipe at that clock"
                  % (name, "f16" if fast else "f16x3", ms, gf / ms, issued, ghz, 100 * issued / (2500.0 * ghz / 2.4)))


if __name__ == "__main__":
    main()
