#!/usr/bin/env python3
"""Timeline of ONE wino4 workgroup (tuning build `make wtrace` in csrc: -DW4_TRACE, no run-time switches in the loop): every wave of
workgroup 0 stamps s_memtime at its phase boundaries of the first 16 chunks - after the chunk's barrier | (unused) | after the matrix loop |
after the work behind it.   usage: W4KIND=6 python tools/wino4_timeline.py [layer]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
os.environ.setdefault("SSM_HIP_LIB", os.path.join(ROOT, "tools", "w4trace_libssm_hip.so"))          # csrc: make wtrace
import torch  # noqa: E402

from ssm_amd import hipbind as hb  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "conv5b"
    shapes = {"conv5b": (512, 512, 46, 80), "conv3b": (128, 128, 184, 320), "conv10b": (64, 64, 368, 640), "conv3a": (64, 128, 184, 320)}
    cin, cout, h, w = shapes[name]
    B = 7
    dev = torch.device("cuda:0")
    lib = hb.load()
    lib.ssm_wino4_debug_buffer.argtypes = [ctypes.c_void_p]
    cnt = torch.zeros(16 + 8 * 16 * 4, dtype=torch.int64, device=dev)
    lib.ssm_wino4_debug_buffer(ctypes.c_void_p(cnt.data_ptr()))
    lib.ssm_wino4_force_kind(int(os.environ.get("W4KIND", "-1")))
    x = hb.Planes(B, cin, h, w, dev)
    x.interior.normal_()
    y = hb.Planes(B, cout, h, w, dev)
    wt = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    bs = torch.randn(cout, device=dev) * 0.1
    pw = hb.PackedWino4(wt, bs, B, h, w)
    for _ in range(3):
        cnt.zero_()
        hb.conv2d_wino4(x.view(), cin, None, 0, pw, y.view(), None, B, h, w)
        torch.cuda.synchronize()
    t = cnt[16:].cpu().reshape(8, 16, 4)
    t0 = int(t[t > 0].min())
    print("%s kind %s: cycles since the first stamp; per wave and chunk: barrier passed | matrix start | matrix end | transform end" % (
        name, os.environ.get("W4KIND", "auto")))
    nch = cin // 4
    # kernel entry (slot 1 of chunk 0) -> first barrier passed | last stamped chunk's work done -> epilogue issued -> its stores complete
    for wv in range(8):
        ent, ep1, ep2 = int(t[wv, 0, 1]), int(t[wv, 1, 1]), int(t[wv, 2, 1])
        last = min(nch, 16) - 1
        print("wave %d: entry -> first barrier %6d | chunk %d done -> epilogue issued %6d -> stores complete %6d | entry -> end %7d (%d chunks%s)" % (
            wv, int(t[wv, 0, 0]) - ent, last, ep1 - int(t[wv, last, 3]), ep2 - ep1, ep2 - ent, nch, "" if nch <= 16 else ", only the first 16 stamped"))
    print()
    for ch in range(2, 8):
        for wv in range(8):
            r = [int(v) - t0 if int(v) else -1 for v in t[wv, ch]]
            print("chunk %2d wave %d: %7d %7d %7d %7d   (matrix %5d, after %5d)" % (ch, wv, r[0], r[1], r[2], r[3], r[2] - r[1], r[3] - r[2]))
        print()


if __name__ == "__main__":
    main()
