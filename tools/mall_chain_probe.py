#!/usr/bin/env python3
"""Diagnostic: does a chain of full-resolution 32-channel layers run faster image by image (each layer's 120 MB output read back by the
next layer while it may still sit in the 256 MB Infinity Cache) than layer by layer over the whole batch (every tensor streams through
HBM)?  Three 32 -> 32 F(4x4,3x3) layers at 736x1280 - the shapes of conv11b / the tail of stage 2 - batch 7.
usage: python tools/mall_chain_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from ssm_amd import hipbind as hb  # noqa: E402


def timed(fn, n=10):
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
    B, C, H, W = 7, 32, 736, 1280
    t = [hb.Planes(B, C, H, W, dev) for _ in range(4)]
    t[0].interior.normal_()
    ws = [torch.randn(C, C, 3, 3, device=dev) / (C * 9) ** 0.5 for _ in range(3)]
    bs = [torch.zeros(C, device=dev) for _ in range(3)]
    for sub in (7, 1, 2):
        pks = [hb.PackedWino4(w, b, sub, H, W) for w, b in zip(ws, bs)]

        def layer_by_layer():
            for li in range(3):
                for b0 in range(0, B, sub):          # (sub = 7: one launch per layer)
                    nb = min(sub, B - b0)
                    if nb != sub:
                        continue
                    hb.conv2d_wino4(t[li].view(b0=b0), C, None, 0, pks[li], t[li + 1].view(b0=b0), None, nb, H, W)

        def image_by_image():
            for b0 in range(0, B - B % sub, sub):
                for li in range(3):
                    hb.conv2d_wino4(t[li].view(b0=b0), C, None, 0, pks[li], t[li + 1].view(b0=b0), None, sub, H, W)

        a, c = timed(layer_by_layer), timed(image_by_image)
        n = (B // sub) * sub
        print("sub-batch %d: layer by layer %.3f ms, chain by chain %.3f ms  (%d images x 3 layers; %.2f / %.2f TB/s of 240 MB per image and layer)"
              % (sub, a, c, n, 0.24 * n * 3 / a, 0.24 * n * 3 / c), flush=True)


if __name__ == "__main__":
    main()
