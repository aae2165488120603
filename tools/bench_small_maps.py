#!/usr/bin/env python3
"""Config 3's deep layers in isolation (2 x 352^2: 44-, 22- and 11-pixel maps, 256-1024 channels): F(2x2,3x3) + split-K with and without the
fused 2x2 mean, and the direct form, as the training plan launches them.  usage: python tools/bench_small_maps.py"""
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
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    B = 2
    if os.environ.get("FORCE_WKIND"):          # a tile configuration of csrc/ssm_wino.hip for every F(2x2,3x3) launch (7-11: the two-workgroup forms)
        hb.load().ssm_wino_force_kind(int(os.environ["FORCE_WKIND"]))
        print("forced F(2x2,3x3) configuration", os.environ["FORCE_WKIND"])
    print("%-8s %5s %5s %7s %5s | %9s %9s %9s | %s" % ("layer", "cin", "cout", "hxw", "pool", "wino us", "no-pool", "direct us", "split-K (wino)"))
    for name, ci, co, s, pool in (("conv4a", 128, 256, 44, False), ("conv4b", 256, 256, 44, True), ("conv5a", 256, 512, 22, False), ("conv5b", 512, 512, 22, True),
                                  ("conv6.0", 512, 512, 11, False), ("conv7b", 512, 512, 22, False), ("conv8b", 256, 256, 44, False), ("conv3b", 128, 128, 88, True)):
        x = hb.Planes(B, ci, s, s, dev)
        x.interior.normal_()
        y = hb.Planes(B, co, s, s, dev)
        yp = hb.Planes(B, co, s // 2, s // 2, dev) if pool else None
        wt = torch.randn(co, ci, 3, 3, device=dev) / (ci * 9) ** 0.5
        bs = torch.randn(co, device=dev) * 0.1
        t_w = t_wn = float("nan")
        ks = "-"
        if hb.wino_supported(ci, co, s, s, 3):
            pw = hb.PackedWino(wt, bs, B, s, s)
            ks = "%d (bn %d)" % (hb.wino_splitk(pw, B, s, s, False), pw.bn)
            t_w = timed(lambda: hb.conv2d_wino(x.view(), ci, None, 0, pw, y.view(), yp.view() if pool else None, B, s, s))
            t_wn = timed(lambda: hb.conv2d_wino(x.view(), ci, None, 0, pw, y.view(), None, B, s, s))
        pk = hb.PackedConv(wt, bs, B, s, s, pool=pool)
        pk.split_ok = True
        t_d = timed(lambda: hb.conv2d(x.view(), ci, None, 0, pk, y.view(), yp.view() if pool else None, B, s, s))
        print("%-8s %5d %5d %3dx%-3d %5s | %9.1f %9.1f %9.1f | %s" % (name, ci, co, s, s, pool, t_w, t_wn, t_d, ks), flush=True)


if __name__ == "__main__":
    main()
