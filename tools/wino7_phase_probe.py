#!/usr/bin/env python3
"""Per-wave phase sums of the blocked 7x7 kernel (tuning build: make -C csrc w7alt W7TAG=tr W7FLAGS=-DW7_TRACE=1).
usage: SSM_HIP_LIB=$PWD/tools/w7tr_libssm_hip.so python tools/wino7_phase_probe.py [cin] [B] [add]      (add: with the per-pair addend, B % 7 == 0)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from ssm_amd import hipbind as hb  # noqa: E402


def main():
    cin = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    H, W, cout = 736, 1280, 32
    dev = torch.device("cuda:0")
    lib = hb.load()
    lib.ssm_wino7_debug_buffer.argtypes = [ctypes.c_void_p]
    cnt = torch.zeros(64, dtype=torch.int64, device=dev)          # 8 waves x 8 counters (the 4-wave kernel uses the first 4 rows)
    wt = torch.randn(cout, cin, 7, 7, device=dev) / (cin * 49) ** 0.5
    bs = torch.randn(cout, device=dev) * 0.1
    pk = hb.PackedWino7(wt, bs, B, H, W)
    x = hb.Planes(B, cin, H, W, dev)
    x.interior.normal_()
    y = hb.Planes(B, cout, H, W, dev)
    akw = {}
    if len(sys.argv) > 3 and sys.argv[3] == "add":
        add = hb.Planes(B // 7, cout, H, W, dev)
        add.interior.normal_()
        akw = dict(add=add.view(), add_div=7)
    for _ in range(3):
        hb.conv2d_wino7(x.view(), cin, None, 0, pk, y.view(), None, B, H, W, **akw)
    torch.cuda.synchronize()
    lib.ssm_wino7_debug_buffer(ctypes.c_void_p(cnt.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hb.conv2d_wino7(x.view(), cin, None, 0, pk, y.view(), None, B, H, W, **akw)
    e1.record()
    torch.cuda.synchronize()
    c = cnt.cpu().view(8, 8)
    print("cin %d batch %d: %.3f ms; iterations per workgroup %d (+2 without MFMAs)" % (cin, B, e0.elapsed_time(e1), cin))
    for w in range(8):
        if int(c[w, 5]) == 0:
            continue
        n = max(int(c[w, 5]), 1)
        work, dma, bar, epi, life = (float(c[w, i]) / n for i in range(5))
        print("wave %d (%d samples): lifetime %8.0f cycles = work %8.0f (%.0f per iteration) + DMA wait %6.0f (%.0f) + barrier wait %6.0f (%.0f) + "
              "epilogue %6.0f" % (w, n, life, work, work / (cin + 2), dma, dma / (cin + 2), bar, bar / (cin + 2), epi))


if __name__ == "__main__":
    main()
