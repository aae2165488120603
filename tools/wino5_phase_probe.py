#!/usr/bin/env python3
"""Per-wave phase sums of the F(4x4,5x5) kernel (tuning build: hipcc ... -DW5_TRACE=1 -> tools/w5tr_libssm_hip.so).
usage: SSM_HIP_LIB=$PWD/tools/w5tr_libssm_hip.so python tools/wino5_phase_probe.py [cin] [cout] [B]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from ssm_amd import hipbind as hb  # noqa: E402


def main():
    cin = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    cout = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 14
    H, W = 368, 640
    dev = torch.device("cuda:0")
    lib = hb.load()
    lib.ssm_wino5_debug_buffer.argtypes = [ctypes.c_void_p]
    cnt = torch.zeros(64, dtype=torch.int64, device=dev)
    wt = torch.randn(cout, cin, 5, 5, device=dev) / (cin * 25) ** 0.5
    bs = torch.randn(cout, device=dev) * 0.1
    pk = hb.PackedWino5(wt, bs, B, H, W)
    x = hb.Planes(B, cin, H, W, dev)
    x.interior.normal_()
    y = hb.Planes(B, cout, H, W, dev)
    for _ in range(3):
        hb.conv2d_wino5(x.view(), cin, None, 0, pk, y.view(), None, B, H, W)
    torch.cuda.synchronize()
    lib.ssm_wino5_debug_buffer(ctypes.c_void_p(cnt.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hb.conv2d_wino5(x.view(), cin, None, 0, pk, y.view(), None, B, H, W)
    e1.record()
    torch.cuda.synchronize()
    ns = cin // 4
    if int(cnt[32:].abs().sum()) > 0:          # the frequency-split kernel (8 waves): matrix | transform | barrier behind each | epilogue | prologue
        c = cnt.cpu().view(8, 8)
        print("cin %d cout %d batch %d: %.3f ms; %d k-steps per workgroup (split kernel)" % (cin, cout, B, e0.elapsed_time(e1), ns))
        for w in range(8):
            n = max(int(c[w, 7]), 1)
            v = [float(c[w, i]) / n for i in range(6)]
            print("wave %d (half %d; %d samples), cycles per k-step: matrix %5.0f + wait/barrier %5.0f | transform %5.0f + wait/barrier %5.0f || prologue %6.0f, "
                  "epilogue %6.0f per workgroup" % (w, w >> 2, n, v[0] / ns, v[2] / ns, v[1] / ns, v[3] / ns, v[5], v[4]))
        return
    c = cnt[:32].cpu().view(4, 8)
    print("cin %d cout %d batch %d: %.3f ms; %d k-steps per workgroup" % (cin, cout, B, e0.elapsed_time(e1), ns))
    for w in range(4):
        n = max(int(c[w, 7]), 1)
        v = [float(c[w, i]) / n for i in range(7)]
        print("wave %d (%d samples), cycles per k-step: top wait %5.0f | row pass %5.0f | barrier %4.0f | column pass %5.0f | barrier %4.0f | matrix %5.0f || "
              "epilogue %6.0f per workgroup" % (w, n, v[0] / ns, v[1] / ns, v[4] / ns, v[2] / ns, v[5] / ns, v[3] / ns, v[6]))


if __name__ == "__main__":
    main()
