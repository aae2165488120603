#!/usr/bin/env python3
"""Diagnostic: do F(4x4,3x3) launches of DIFFERENT HIP streams share the chip?  One 512 -> 512 layer at 46x80 and batch B is
B * 9 * 8 workgroups of one CU each (64-cout form: 512 threads, ~140 KiB LDS): at B = 2, 144 workgroups on 256 CUs.  N launches
on one stream against the same N launches dealt over S streams (own input / output planes per stream): if the streams' kernels
run side by side, two streams at B = 2 take the time of one.
usage: python tools/stream_overlap_probe.py [N]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from ssm_amd import hipbind as hb  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    dev = torch.device("cuda:0")
    for cin, cout, h, w, B in ((512, 512, 46, 80, 2), (512, 512, 46, 80, 1), (256, 256, 92, 160, 2), (64, 64, 368, 640, 2)):
        wt = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        bs = torch.randn(cout, device=dev) * 0.1
        pw = hb.PackedWino4(wt, bs, B, h, w, ups=False)
        sets = []
        for _ in range(4):
            x = hb.Planes(B, cin, h, w, dev)
            x.interior.normal_()
            sets.append((x, hb.Planes(B, cout, h, w, dev), torch.cuda.Stream()))
        torch.cuda.synchronize()

        def run(S):
            for i in range(n):
                x, y, st = sets[i % S]
                with torch.cuda.stream(st):
                    hb.conv2d_wino4(x.view(), cin, None, 0, pw, y.view(), None, B, h, w)

        res = []
        for S in (1, 2, 3, 4):
            run(S)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(S)
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            res.append((S, (time.perf_counter() - t0) * 1e3 / n, t_host * 1e3 / n))
        kind = hb.wino4_plan(cin, cout, B, h, w, False)[0]
        print("%d -> %d at %dx%d, batch %d (kind %d): ms per launch (host enqueue)  " % (cin, cout, h, w, B, kind)
              + "  ".join("%d stream(s) %.4f (%.4f)" % r for r in res), flush=True)


if __name__ == "__main__":
    main()
