#!/usr/bin/env python3
"""Per-layer A/B of the 7x7 / 5x5 layers at the 736x1280 shapes of SURVEY Appendix A: direct fp32-MFMA kernel (csrc/ssm_conv.hip) vs
the 1-D Winograd kernel (csrc/ssm_wino1d.hip: F(2,7) / F(4,5) along x), with the fused 2x2 mean where the plan has it.  TFLOP/s are
ALGORITHMIC (direct-form FLOPs); "issued" = the multiply-adds the matrix cores execute in the Winograd form (x 8/14, x 8/20).
usage: python tools/bench_layers_wino1d.py [B] [H] [W] [kind|-1]      (kinds: 0 R7A, 1 R7B, 2 R5A, 3 R5B, 4 R5C)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from ssm_amd import hipbind as hb  # noqa: E402

# (name, cin, cout, k, scale, pooled): stage 1 / stage 2 (per-t part of the hoisted conv1a: 10 channels) shapes
LAYERS = [("s1.conv1a", 6, 32, 7, 1, False), ("s2.conv1a(t)", 10, 32, 7, 1, False), ("conv1b", 32, 32, 7, 1, True),
          ("conv2a", 32, 64, 5, 2, False), ("conv2b", 64, 64, 5, 2, True)]


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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 736
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
    force = int(sys.argv[4]) if len(sys.argv) > 4 else -1
    dev = torch.device("cuda:0")
    hb.load().ssm_wino1d_force_kind(force)
    tot = {7: [0.0, 0.0, 0.0], 5: [0.0, 0.0, 0.0]}
    print("%-13s %4s %4s %9s %9s | %9s %7s | %8s %7s %7s %4s | %6s %9s" % ("layer", "cin", "cout", "hxw", "GFLOP", "direct ms", "TF/s",
                                                                          "wino ms", "TF/s", "issued", "kind", "ratio", "max|diff|"))
    for name, cin, cout, k, s, pooled in LAYERS:
        h, w = H // s, W // s
        wt = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
        bs = torch.randn(cout, device=dev) * 0.1
        pk = hb.PackedConv(wt, bs, B, h, w, pool=pooled)
        try:
            pw = hb.PackedWino1d(wt, bs, B, h, w, pool=pooled)
        except (RuntimeError, AssertionError) as e:
            print("%-13s skipped: %s" % (name, str(e)[:80]), flush=True)
            continue
        cp = max(pk.cin_p, pw.cin_p)
        x = hb.Planes(B, cp, h, w, dev)
        x.interior[:, :cin].normal_()
        y0, y1 = hb.Planes(B, cout, h, w, dev), hb.Planes(B, cout, h, w, dev)
        p0 = hb.Planes(B, cout, h // 2, w // 2, dev) if pooled else None
        p1 = hb.Planes(B, cout, h // 2, w // 2, dev) if pooled else None
        f0 = lambda: hb.conv2d(x.view(), pk.cin_p, None, 0, pk, y0.view(), p0.view() if pooled else None, B, h, w)  # noqa: E731
        f1 = lambda: hb.conv2d_wino1d(x.view(), pw.cin_p, None, 0, pw, y1.view(), p1.view() if pooled else None, B, h, w)  # noqa: E731
        t0 = timed(f0) if not os.environ.get("NO_DIRECT") else float("nan")
        t1 = timed(f1)
        diff = float((y0.interior - y1.interior).abs().max()) if not os.environ.get("NO_DIRECT") else float("nan")
        gf = 2.0 * B * h * w * cout * cin * k * k / 1e9
        fac = 8.0 / 14.0 if k == 7 else 8.0 / 20.0
        tot[k][0] += gf
        tot[k][1] += t0
        tot[k][2] += t1
        print("%-13s %4d %4d %4dx%-4d %9.2f | %9.3f %7.1f | %8.3f %7.1f %7.1f %4d | %6.2f %9.2e" % (
            name, cin, cout, h, w, gf, t0, gf / t0, t1, gf / t1, fac * gf / t1, hb.wino1d_plan(k, cin, cout, B, h, w)[0], t0 / t1, diff), flush=True)
        del x, y0, y1, p0, p1, pk, pw
    for k, (gf, t0, t1) in tot.items():
        if t1 > 0:
            fac = 8.0 / 14.0 if k == 7 else 8.0 / 20.0
            print("TOTAL %dx%d layers: %.1f GFLOP; direct %.2f ms = %.1f TFLOP/s; 1-D winograd %.2f ms = %.1f TFLOP/s algorithmic, %.1f issued "
                  "(fp32 MFMA peak 157.3)" % (k, k, gf, t0, gf / t0, t1, gf / t1, fac * gf / t1))


if __name__ == "__main__":
    main()
