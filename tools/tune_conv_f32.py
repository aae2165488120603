#!/usr/bin/env python3
"""Times every tile configuration of the fp32-MFMA convolution (csrc/ssm_conv.hip) on the layers of a U-Net pass and
marks the one the library's cost model picks.   usage: python tools/tune_conv_f32.py [B] [stage] [H] [W]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from ssm_amd import hipbind as hb  # noqa: E402
from ssm_amd.engine import POOLED, UNetPlan, layer_scale  # noqa: E402
from ssm_amd.weights import unet_layers  # noqa: E402

KINDS = ["K7", "K5", "K3N32", "K3N64", "K3N128", "K3N128S", "K3N64T", "K3N32T", "K3N128G", "K3N64G", "K3N64GS", "K3N32G",
         "K3N32GS", "K5G", "K7G"]
KS = {"K7": 7, "K7G": 7, "K5": 5, "K5G": 5}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    stage = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 736
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 1280
    only = os.environ.get("TUNE_ONLY", "").split(",") if os.environ.get("TUNE_ONLY") else None
    dev = torch.device("cuda:0")
    lib = hb.load()
    tot_auto = tot_best = tot_f = 0.0
    for name, cin, cout, k in unet_layers(stage, True):
        s = layer_scale(name)
        h, w = H // s, W // s
        ups, pool = name in UNetPlan.UPS, name in POOLED
        lib.ssm_conv_force_kind(-1)
        auto = hb.conv_plan(k, cin, cout, B, h, w, pool, ups)[0]
        sh, sw = (h // 2, w // 2) if ups else (h, w)
        wt = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
        bs = torch.zeros(cout, device=dev)
        y = hb.Planes(B, cout, h, w, dev)
        yp = hb.Planes(B, cout, h // 2, w // 2, dev) if pool else None
        gf = 2.0 * B * h * w * cout * cin * k * k / 1e9
        res = {}
        for ki, kn in enumerate(KINDS):
            if KS.get(kn, 3) != k or (kn == "K3N32T" and pool):
                continue
            if only and kn not in only:
                continue
            if kn in ("K3N64", "K3N64T", "K3N64G", "K3N64GS") and cout <= 32:
                continue
            if kn in ("K3N128", "K3N128S", "K3N128G") and cout <= 64:
                continue
            lib.ssm_conv_force_kind(ki)
            pk = hb.PackedConv(wt, bs, B, h, w, pool=pool, ups=ups)
            x = hb.Planes(B, pk.cin_p, sh, sw, dev)
            x.interior.normal_()

            def run():
                if ups:
                    hb.conv2d_ups(x.view(), pk.cin_p, None, 0, pk, y.view(), B, h, w)
                else:
                    hb.conv2d(x.view(), pk.cin_p, None, 0, pk, y.view(), yp.view() if pool else None, B, h, w)
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 5
            e0.record()
            for _ in range(n):
                run()
            e1.record()
            torch.cuda.synchronize()
            res[kn] = e0.elapsed_time(e1) / n
            del x, pk
        best = min(res, key=res.get)
        an = KINDS[auto] if KINDS[auto] in res else best
        tot_auto += res[an]
        tot_best += res[best]
        tot_f += gf
        print("%-10s %4d->%-4d k%d %4dx%-4d %s%s auto %-8s %.3f ms %6.1f TF | best %-8s %.3f ms %6.1f TF | %s"
              % (name, cin, cout, k, h, w, "U" if ups else " ", "P" if pool else " ", an, res[an], gf / res[an], best, res[best],
                 gf / res[best], "  ".join("%s %.3f" % (kn, v) for kn, v in sorted(res.items(), key=lambda kv: kv[1]))))
        sys.stdout.flush()
    print("TOTAL B=%d stage %d: auto %.2f ms (%.1f TF), best-per-layer %.2f ms (%.1f TF)"
          % (B, stage, tot_auto, tot_f / tot_auto, tot_best, tot_f / tot_best))


if __name__ == "__main__":
    main()
