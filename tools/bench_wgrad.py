"""Weight-gradient kernels in isolation: every (k, Cin per source, Cout, map size) of the training step at 2x352x352, the fp32-MFMA
kernel (ssm_conv2d_wgrad) beside the split-bf16 one (ssm_conv2d_wgrad_bf16x3), one stream, HIP-event timed.
Usage: python tools/bench_wgrad.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
for p in (ROOT, PKG):
    sys.path.insert(0, p)
import torch  # noqa: E402

from ssm_amd import backward as Bk  # noqa: E402
from ssm_amd import hipbind as hb  # noqa: E402

LAYERS = [  # name, k, cin (per source), cout, scale, calls per step (both U-Nets, both sources)
    ("conv1a(s2)", 7, 16, 32, 1, 1), ("conv1b", 7, 32, 32, 1, 2), ("conv2a", 5, 32, 64, 2, 2), ("conv2b", 5, 64, 64, 2, 2),
    ("conv3a", 3, 64, 128, 4, 2), ("conv3b", 3, 128, 128, 4, 2), ("conv4b", 3, 256, 256, 8, 2), ("conv5b", 3, 512, 512, 16, 2),
    ("conv6.x", 3, 512, 512, 32, 4), ("conv7b", 3, 512, 512, 16, 2), ("conv8a/src", 3, 512, 256, 8, 4), ("conv8b", 3, 256, 256, 8, 2),
    ("conv9a/src", 3, 256, 128, 4, 4), ("conv9b", 3, 128, 128, 4, 2), ("conv10a/src", 3, 128, 64, 2, 4), ("conv10b", 3, 64, 64, 2, 2),
    ("conv11a/src", 3, 64, 32, 1, 4), ("conv11b", 3, 32, 32, 1, 2), ("fuse/src", 3, 32, 32, 1, 4), ("final", 3, 32, 5, 1, 2),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda:0")
    B, H0, W0 = 2, 352, 352
    tot = {"f32": 0.0, "bf16x3": 0.0}
    print("%-12s %2s %4s %4s %4s | %9s %7s | %9s %7s" % ("layer", "k", "cin", "cout", "HxW", "f32 ms", "TF", "bf16x3 ms", "TF"))
    for name, k, cin, cout, s, calls in LAYERS:
        H, W = H0 // s, W0 // s
        x = hb.Planes(B, cin, H, W, dev).load(torch.randn(B, cin, H, W, device=dev))
        dz = hb.Planes(B, cout, H, W, dev).load(torch.randn(B, cout, H, W, device=dev) * 1e-3)
        dw = torch.empty(cout, cin, k, k, device=dev)
        flops = 2.0 * B * H * W * cin * cout * k * k
        row = []
        for split in (False, True):
            for _ in range(2):
                Bk.wgrad(x, dz, dw, k, split=split)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                Bk.wgrad(x, dz, dw, k, split=split)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            row += [ms, flops / ms * 1e-9]
            tot["bf16x3" if split else "f32"] += ms * calls
        print("%-12s %2d %4d %4d %4d | %9.4f %7.1f | %9.4f %7.1f" % (name, k, cin, cout, H, row[0], row[1], row[2], row[3]))
    print("sum over a training step's calls: f32 %.3f ms, bf16x3 %.3f ms" % (tot["f32"], tot["bf16x3"]))


if __name__ == "__main__":
    main()
