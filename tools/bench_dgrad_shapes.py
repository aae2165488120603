#!/usr/bin/env python3
"""Diagnostic: the data-gradient convolutions of the fused-upsample layers at config 3's shapes (transposed: few input, many output channels at
the layer's OUTPUT resolution) in the direct, F(2x2,3x3) and F(4x4,3x3) kernels, batch 2, in isolation.
usage: python tools/bench_dgrad_shapes.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from ssm_amd import hipbind as hb  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device("cuda:0")
    B = 2
    print("%-22s %9s | %9s %9s %9s  (ms; F(4x4) preferred by the cost model?)" % ("layer (dX shape)", "GFLOP", "direct", "F(2x2)", "F(4x4)"))
    for name, cin, cout, hw in (("conv11a 32 -> 128", 32, 128, 352), ("conv10a 64 -> 256", 64, 256, 176), ("conv9a 128 -> 512", 128, 512, 88),
                                ("conv8a 256 -> 1024", 256, 1024, 44), ("conv7a 512 -> 1024", 512, 1024, 22), ("conv11b 32 -> 32", 32, 32, 352),
                                ("fuse_conv 32 -> 64", 32, 64, 352), ("conv10b 64 -> 64", 64, 64, 176), ("conv9b 128 -> 128", 128, 128, 88)):
        x = hb.Planes(B, cin, hw, hw, dev)
        x.interior.normal_()
        y = hb.Planes(B, cout, hw, hw, dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        b = torch.zeros(cout, device=dev)
        res = []
        for cls, fn in ((hb.PackedConv, hb.conv2d), (hb.PackedWino, hb.conv2d_wino), (hb.PackedWino4, hb.conv2d_wino4)):
            try:
                pk = cls(w, b, B, hw, hw)
                res.append(timed(lambda: fn(x.view(), pk.cin_p, None, 0, pk, y.view(), None, B, hw, hw, lrelu=False)))
            except (RuntimeError, AssertionError) as e:
                res.append(float("nan"))
        gf = 2.0 * B * hw * hw * cin * cout * 9 / 1e9
        print("%-22s %9.2f | %9.4f %9.4f %9.4f  %s" % (name, gf, res[0], res[1], res[2], hb.wino4_preferred(cin, cout, B, hw, hw, False)), flush=True)


if __name__ == "__main__":
    main()
