#!/usr/bin/env python3
"""Per-layer A/B of the 3x3 layers at the 736x1280 shapes of SURVEY Appendix A: direct fp32-MFMA kernel (csrc/ssm_conv.hip) vs the
Winograd F(2x2,3x3) fp32 kernel (csrc/ssm_wino.hip), plain and fused-upsample forms.  TFLOP/s are ALGORITHMIC (direct-form FLOPs).
usage: python tools/bench_layers_wino.py [B] [H] [W] [wino_kind|-1]      (W4=1: the Winograd column is F(4x4,3x3), csrc/ssm_wino4.hip;
W4KIND forces its tile configuration; ONLY=a,b: those layers only)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from ssm_amd import hipbind as hb  # noqa: E402
from ssm_amd.engine import UNetPlan, layer_scale  # noqa: E402
from ssm_amd.weights import unet_layers  # noqa: E402

CAT = {"conv8a": 512, "conv9a": 256, "conv10a": 128, "conv11a": 64, "conv7a": 512, "fuse_conv": 32}     # channels of the first source


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
    hb.load().ssm_wino_force_kind(force)
    w4 = os.environ.get("W4", "0") != "0"
    if w4:
        hb.load().ssm_wino4_force_kind(int(os.environ.get("W4KIND", "-1")))
    tot = [0.0, 0.0, 0.0]
    print("%-10s %5s %5s %9s %4s %9s | %8s %7s | %8s %7s %5s | %6s %9s" % ("layer", "cin", "cout", "hxw", "ups", "GFLOP", "direct ms", "TF/s",
                                                                       "wino ms", "TF/s", "kind", "ratio", "max|diff|"))
    for name, cin, cout, k in unet_layers(2, True):
        if k != 3 or cout < 32:
            continue
        if os.environ.get("ONLY") and name not in os.environ["ONLY"].split(","):
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
        y0, y1 = hb.Planes(B, cout, h, w, dev), hb.Planes(B, cout, h, w, dev)
        wt = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        bs = torch.randn(cout, device=dev) * 0.1
        pk = hb.PackedConv(wt, bs, B, h, w, ups=ups)
        if w % 2:
            continue
        try:
            pw = (hb.PackedWino4 if w4 else hb.PackedWino)(wt, bs, B, h, w, ups=ups)
        except (RuntimeError, AssertionError) as e:
            print("%-10s skipped: %s" % (name, str(e)[:80]), flush=True)
            continue
        bv = xb.view() if xb is not None else None
        from ssm_amd.engine import WINO4_SUBPIXEL
        subpix = w4 and ups and name in WINO4_SUBPIXEL and hb.subpixel_wino4_supported(cin, cout, h, w)          # the plan's form of this layer
        if subpix:
            psp = hb.PackedSubpixelWino4(wt, bs, B, h, w)
            f0 = lambda: hb.conv2d_ups(xa.view(), c1, bv, c2, pk, y0.view(), B, h, w)  # noqa: E731
            f1 = lambda: hb.conv2d_ups_subpixel_wino4(lambda yy, xx: xa.view(y0=yy, x0=xx), c1, (lambda yy, xx: xb.view(y0=yy, x0=xx)) if xb is not None else None,  # noqa: E731
                                                      c2, psp, lambda yy, xx: y1.view(y0=yy, x0=xx), B, h, w)
        elif ups:
            f0 = lambda: hb.conv2d_ups(xa.view(), c1, bv, c2, pk, y0.view(), B, h, w)  # noqa: E731
            f1 = lambda: (hb.conv2d_ups_wino4 if w4 else hb.conv2d_ups_wino)(xa.view(), c1, bv, c2, pw, y1.view(), B, h, w)  # noqa: E731
        else:
            f0 = lambda: hb.conv2d(xa.view(), c1, bv, c2, pk, y0.view(), None, B, h, w)  # noqa: E731
            f1 = lambda: (hb.conv2d_wino4 if w4 else hb.conv2d_wino)(xa.view(), c1, bv, c2, pw, y1.view(), None, B, h, w)  # noqa: E731
        t0 = timed(f0) if not os.environ.get("NO_DIRECT") else float("nan")
        try:
            t1 = timed(f1)
        except (RuntimeError, AssertionError) as e:        # forced configuration not applicable to this layer
            print("%-10s skipped: %s" % (name, str(e)[:80]), flush=True)
            continue
        diff = float((y0.interior - y1.interior).abs().max()) if not os.environ.get("NO_DIRECT") else float("nan")
        gf = 2.0 * B * h * w * cout * cin * 9 / 1e9
        tot[0] += gf
        tot[1] += t0
        tot[2] += t1
        print("%-10s %5d %5d %4dx%-4d %4d %9.2f | %8.3f %7.1f | %8.3f %7.1f %5d | %6.2f %9.2e" % (
            name, cin, cout, h, w, ups, gf, t0, gf / t0, t1, gf / t1, -4 if subpix else (hb.wino4_plan if w4 else hb.wino_plan)(cin, cout, B, h, w, ups)[0], t0 / t1, diff), flush=True)          # (kind -4: sub-pixel interior + border ring)
        del xa, xb, y0, y1, pk, pw
    print("TOTAL 3x3 layers: %.1f GFLOP; direct %.2f ms = %.1f TFLOP/s; winograd %.2f ms = %.1f TFLOP/s algorithmic (fp32 MFMA peak 157.3)" % (
        tot[0], tot[1], tot[0] / tot[1], tot[2], tot[0] / tot[2]))


if __name__ == "__main__":
    main()
