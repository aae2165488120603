"""Winograd-domain weight gradient (csrc/ssm_wgradw.hip) beside the direct fp32-MFMA kernel on every 3x3 layer of the training step at
2 x 352 x 352 that the Winograd form serves, one stream, HIP-event timed; $SSM_WGRADW_TARGET sweeps the workgroups per launch.
Usage: python tools/bench_wgradw.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
for p in (ROOT, PKG):
    sys.path.insert(0, p)
import torch  # noqa: E402

from ssm_amd import backward as Bk  # noqa: E402
from ssm_amd import hipbind as hb  # noqa: E402

LAYERS = [  # name, cin, cout, scale, calls per step (both U-Nets)
    ("conv3a", 64, 128, 4, 2), ("conv3b", 128, 128, 4, 2), ("conv4a", 128, 256, 8, 2), ("conv4b", 256, 256, 8, 2),
    ("conv8a", 1024, 256, 8, 2), ("conv8b", 256, 256, 8, 2), ("conv9a", 512, 128, 4, 2), ("conv9b", 128, 128, 4, 2),
    ("conv10a", 256, 64, 2, 2), ("conv10b", 64, 64, 2, 2), ("conv11a", 128, 32, 1, 2), ("conv11b", 32, 32, 1, 2),
    ("fuse/src", 32, 32, 1, 4),
]


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda:0")
    B, H0, W0 = 2, 352, 352
    tot = [0.0, 0.0, 0.0]
    print("target %s" % os.environ.get("SSM_WGRADW_TARGET", "default"))
    print("%-9s %5s %4s %4s | %9s %6s | %9s %6s %9s | %8s" % ("layer", "cin", "cout", "HxW", "direct ms", "TF", "wino ms", "TF", "finish ms", "rel diff"))
    only = [n for n in os.environ.get("WW_LAYERS", "").split(",") if n]          # ablation builds: a few layers, no parity column
    for name, cin, cout, s, calls in LAYERS:
        if only and name not in only:
            continue
        H, W = H0 // s, W0 // s
        x = hb.Planes(B, cin, H, W, dev).load(torch.randn(B, cin, H, W, device=dev))
        dz = hb.Planes(B, cout, H, W, dev).load(torch.randn(B, cout, H, W, device=dev) * 1e-3)
        dw = torch.zeros(cout, cin, 3, 3, device=dev)
        db = torch.zeros(cout, device=dev)
        flops = 2.0 * B * H * W * cin * cout * 9
        t_dir = timed(lambda: Bk.wgrad(x, dz, dw, 3, zero_first=False, bias_acc=db), reps)
        du = torch.zeros(16, cout, cin, device=dev)
        dw2 = torch.zeros(cout, cin, 3, 3, device=dev)
        fin = hb.WgradWinoFinish([(du, dw2)], dev)
        t_w = timed(lambda: hb.wgrad_wino(x.view(), dz.view(), du, db, B, cin, cout, H, W, cin, 0), reps)
        t_f = timed(lambda: fin.run(), reps)
        # parity of one clean evaluation of each
        dw.zero_(), dw2.zero_(), du.zero_()
        Bk.wgrad(x, dz, dw, 3, zero_first=False)
        hb.wgrad_wino(x.view(), dz.view(), du, None, B, cin, cout, H, W, cin, 0)
        fin.run()
        rel = float((dw - dw2).abs().max() / dw.abs().max())
        tot[0] += t_dir * calls
        tot[1] += t_w * calls
        tot[2] += t_f * calls
        print("%-9s %5d %4d %4d | %9.4f %6.1f | %9.4f %6.1f %9.4f | %8.1e" % (name, cin, cout, H, t_dir, flops / t_dir * 1e-9, t_w,
                                                                         flops / t_w * 1e-9, t_f, rel))
    print("sum over a training step's calls: direct %.3f ms, Winograd domain %.3f ms + finishing (one launch per layer here) %.3f ms" % tuple(tot))


if __name__ == "__main__":
    main()
