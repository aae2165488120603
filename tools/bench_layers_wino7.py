#!/usr/bin/env python3
"""Per-layer A/B of the 7x7 layers at the 736x1280 shapes of SURVEY Appendix A: the 1-D Winograd kernel (csrc/ssm_wino1d.hip, F(2,7) along
x) vs the blocked two-dimensional form (csrc/ssm_wino7.hip, 2x2 blocks of F(4x4,4x4)), with the fused 2x2 mean where the plan has it.
TFLOP/s are ALGORITHMIC (direct-form FLOPs); "issued" = the multiply-adds the matrix cores execute (x 8/14 | x 196/784).
usage: python tools/bench_layers_wino7.py [B] [H] [W] [kind|-1]      (kinds: 0 Z7A, 1 Z7B, 2 / 3 the same tiles in the frequency-split kernel of eight waves; -1: the plan's choice, $SSM_WINO7_SPLIT)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from ssm_amd import hipbind as hb  # noqa: E402

# (name, cin, cout, pooled): stage 1 / stage 2 (per-t part of the hoisted conv1a: 10 channels; its per-pair part: 6) shapes
LAYERS = [("s1.conv1a", 6, 32, False), ("s2.conv1a(t)", 10, 32, False), ("s2.conv1a(t)+add", 10, 32, False), ("conv1b", 32, 32, True)]


def timed(fn, n=5):
    for _ in range(2):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 736
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
    force = int(sys.argv[4]) if len(sys.argv) > 4 else -1
    dev = torch.device("cuda:0")
    hb.load().ssm_wino7_force_kind(force)
    tot = [0.0, 0.0, 0.0]
    print("%-13s %4s %4s %9s %9s | %8s %7s %7s | %8s %7s %7s | %6s %9s" % ("layer", "cin", "cout", "hxw", "GFLOP", "1-D ms", "TF/s", "issued",
                                                                          "2-D ms", "TF/s", "issued", "ratio", "max|diff|"))
    for name, cin, cout, pooled in LAYERS:
        wt = torch.randn(cout, cin, 7, 7, device=dev) / (cin * 49) ** 0.5
        bs = torch.randn(cout, device=dev) * 0.1
        p1 = hb.PackedWino1d(wt, bs, B, H, W, pool=pooled)
        p7 = hb.PackedWino7(wt, bs, B, H, W, pool=pooled)
        x = hb.Planes(B, p1.cin_p, H, W, dev)
        x.interior[:, :cin].normal_()
        y0, y1 = hb.Planes(B, cout, H, W, dev), hb.Planes(B, cout, H, W, dev)
        q0 = hb.Planes(B, cout, H // 2, W // 2, dev) if pooled else None
        q1 = hb.Planes(B, cout, H // 2, W // 2, dev) if pooled else None
        add = hb.Planes(max(B // 7, 1), cout, H, W, dev) if name.endswith("+add") and B % 7 == 0 else None      # the hoisted plan's addend: one per pair
        if add is not None:
            add.interior.normal_()
        akw = dict(add=add.view(), add_div=7) if add is not None else {}
        f0 = lambda: hb.conv2d_wino1d(x.view(), p1.cin_p, None, 0, p1, y0.view(), q0.view() if pooled else None, B, H, W, **akw)  # noqa: E731
        f1 = lambda: hb.conv2d_wino7(x.view(), cin, None, 0, p7, y1.view(), q1.view() if pooled else None, B, H, W, **akw)  # noqa: E731
        t0 = timed(f0)
        t1 = timed(f1)
        diff = float((y0.interior - y1.interior).abs().max())
        gf = 2.0 * B * H * W * cout * cin * 49 / 1e9
        tot[0] += gf
        tot[1] += t0
        tot[2] += t1
        print("%-13s %4d %4d %4dx%-4d %9.2f | %8.3f %7.1f %7.1f | %8.3f %7.1f %7.1f | %6.2f %9.2e" % (
            name, cin, cout, H, W, gf, t0, gf / t0, gf / t0 * 8 / 14, t1, gf / t1, gf / t1 / 4, t0 / t1, diff), flush=True)
        del x, y0, y1, q0, q1, p1, p7
    gf, t0, t1 = tot
    print("TOTAL 7x7 layers at batch %d: %.1f GFLOP; F(2,7) %.2f ms = %.1f TFLOP/s algorithmic, %.1f issued; blocked F(4x4,4x4) %.2f ms = %.1f "
          "TFLOP/s algorithmic, %.1f issued (fp32 MFMA peak 157.3)" % (B, gf, t0, gf / t0, gf / t0 * 8 / 14, t1, gf / t1, gf / t1 / 4))


if __name__ == "__main__":
    main()
