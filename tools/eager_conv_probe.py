#!/usr/bin/env python3
"""What the reference's own GPU path would get on this chip: PyTorch-ROCm eager `F.conv2d` (MIOpen, fp32, cudnn.benchmark=True like
scripts/main.py:296 of the reference) timed per layer at the 736x1280 shapes of SURVEY Appendix A.  Calibration only - nothing on
the product path calls MIOpen.
usage: python tools/eager_conv_probe.py [B] [H] [W]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from ssm_amd.engine import layer_scale  # noqa: E402
from ssm_amd.weights import unet_layers  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 736
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
    torch.backends.cudnn.benchmark = True
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    dev = torch.device("cuda:0")
    tot_t = tot_f = 0.0
    print("%-10s %5s %5s %2s %9s %9s %8s %8s" % ("layer", "cin", "cout", "k", "hxw", "GFLOP", "ms", "TFLOP/s"), flush=True)
    for name, cin, cout, k in unet_layers(2, True):
        s = layer_scale(name)
        h, w = H // s, W // s
        x = torch.randn(B, cin, h, w, device=dev)
        wt = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
        bs = torch.zeros(cout, device=dev)
        t0 = time.time()
        for _ in range(2):
            y = F.conv2d(x, wt, bs, padding=(k - 1) // 2)
        torch.cuda.synchronize()
        tfind = time.time() - t0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            y = F.conv2d(x, wt, bs, padding=(k - 1) // 2)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        gf = 2.0 * B * h * w * cout * cin * k * k / 1e9
        tot_t += ms
        tot_f += gf
        print("%-10s %5d %5d %2d %4dx%-4d %9.2f %8.3f %8.1f   (find %.1f s)" % (name, cin, cout, k, h, w, gf, ms, gf / ms, tfind), flush=True)
        del x, y, wt
    print("TOTAL stage-2 convs via MIOpen eager: %.1f GFLOP in %.2f ms = %.1f TFLOP/s algorithmic" % (tot_f, tot_t, tot_f / tot_t), flush=True)


if __name__ == "__main__":
    main()
