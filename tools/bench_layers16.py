#!/usr/bin/env python3
"""Per-layer timing of the fp16-MFMA convolution (HL8 layout) at the 736x1280 shapes.
usage: python tools/bench_layers16.py [B] [fast(0/1)]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from ssm_amd import hipbind as hb  # noqa: E402
from ssm_amd.engine import layer_scale  # noqa: E402
from ssm_amd.weights import unet_layers  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    fast = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
    q8 = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
    H, W = int(os.environ.get("SSM_BENCH_H", 736)), int(os.environ.get("SSM_BENCH_W", 1280))
    dev = torch.device("cuda:0")
    tot_t = tot_f = 0.0
    rows = []
    print("mode: %s   B=%d" % ("fp16 fast (1 MFMA)" if fast else ("fp16 + 2 x scaled fp8 (Q8)" if q8 else "fp16 split (3 MFMA, fp32-grade)"), B))
    # per-layer roofline: at fp16 the ridge is 2500 TFLOP/s / 8 TB/s = 312 FLOP/B, so a layer's roof is
    # min(MFMA peak / products per MAC, 8 TB/s x its arithmetic intensity).  Algorithmic bytes, one touch: the input's fp16 parts the mode
    # reads (hi only in the fast mode, hi + lo otherwise: 2 / 4 B per value), the HL8 output (hi + lo: 4 B per value; fp32 for final_conv),
    # the packed filter once.
    PEAK, HBM = 2500.0, 8.0e3          # TFLOP/s dense fp16 MFMA, GB/s
    mfma_per_mac = 1.0 if fast else (1.5 if q8 else 3.0)
    print("%-10s %5s %5s %2s %9s %9s %8s %8s | %8s %8s %9s %6s" % ("layer", "cin", "cout", "k", "hxw", "GFLOP", "ms", "TFLOP/s", "MB", "FLOP/B", "roof TF/s", "frac"))
    for name, cin, cout, k in unet_layers(2, True):
        s = layer_scale(name)
        h, w = H // s, W // s
        pk = hb.PackedConv16(torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5,
                             torch.zeros(cout, device=dev), w, q8=q8)
        x = hb.HPlanes(B, cin, h, w, dev, groups=pk.cin_p // 8, q8=q8)
        x.load(torch.randn(B, cin, h, w, device=dev))
        y = hb.HPlanes(B, cout, h, w, dev) if cout % 8 == 0 else None
        y32 = torch.empty(B, cout, h, w, device=dev) if y is None else None
        args = (x.view(), pk.cin_p, None, 0, pk, y.view() if y else None, hb.view_of(y32) if y32 is not None else None,
                None, B, h, w)
        for _ in range(2):
            hb.conv2d_hl8(*args, fast=fast)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            hb.conv2d_hl8(*args, fast=fast)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        gf = 2.0 * B * h * w * cout * cin * k * k / 1e9
        tot_t += ms
        tot_f += gf
        nbytes = B * h * w * (cin * (2.0 if fast else 4.0) + cout * 4.0) + cout * cin * k * k * (2.0 if fast else 4.0)
        ai = gf * 1e9 / nbytes
        roof = min(PEAK / mfma_per_mac, HBM * ai / 1e3)
        bound = "hbm" if HBM * ai / 1e3 < PEAK / mfma_per_mac else "mfma"
        rows.append((name, gf, ms, roof, bound))
        print("%-10s %5d %5d %2d %4dx%-4d %9.2f %8.3f %8.1f | %8.1f %8.0f %6.0f %-4s %6.3f" % (name, cin, cout, k, h, w, gf, ms, gf / ms, nbytes / 1e6, ai, roof, bound, gf / ms / roof))
        del x, y, pk
    print("TOTAL stage-2 convs: %.1f GFLOP in %.2f ms = %.1f TFLOP/s (algorithmic)" % (tot_f, tot_t, tot_f / tot_t))
    t_roof = sum(gf / roof for _, gf, _, roof, _ in rows)          # ms if every layer ran AT its own roof
    for bd in ("hbm", "mfma"):
        sel = [r for r in rows if r[4] == bd]
        if sel:
            print("  %s-bound layers (%d): %.2f ms measured, %.2f ms at their roofs -> %.3f of the per-layer min(MFMA, HBM) roofline"
                  % (bd, len(sel), sum(r[2] for r in sel), sum(r[1] / r[3] for r in sel), sum(r[1] / r[3] for r in sel) / sum(r[2] for r in sel)))
    print("  all layers: %.2f ms at the per-layer roofs vs %.2f ms measured = %.3f" % (t_roof, tot_t, t_roof / tot_t))


if __name__ == "__main__":
    main()
