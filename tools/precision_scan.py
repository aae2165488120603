#!/usr/bin/env python3
"""Diagnostic: end-to-end sensitivity of the output frames to running ONE convolution with plain fp16 operands (hi*hi
only) inside the split-fp16 pipeline, at 736x1280 with 7 intermediates.  Prints max-abs difference to the all-split result
per layer, most sensitive first."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from ssm_amd.engine import PairEngine, UNetPlan  # noqa: E402
from ssm_amd.weights import synthetic_frames, synthetic_state_dict, unet_layers  # noqa: E402

dev = torch.device("cuda:0")
sd1 = {k: v.to(dev) for k, v in synthetic_state_dict(1).items()}
sd2 = {k: v.to(dev) for k, v in synthetic_state_dict(2).items()}
x = synthetic_frames(2, 720, 1280, seed=42)
H, W = x.shape[-2:]
img6 = x.reshape(1, 6, H, W).to(dev)
t = torch.tensor([i / 8.0 for i in range(1, 8)], device=dev)
eng = PairEngine(sd1, sd2, 1, 7, H, W, dev, True, "f16x3")
ref = eng.run(img6, t, want_aux=False).clone()
rows = []
for stage in (1, 2):
    for name, cin, cout, k in unet_layers(stage, True):
        UNetPlan.fast_layers = frozenset(["s%d.%s" % (stage, name)])
        out = eng.run(img6, t, want_aux=False)
        rows.append((float((out - ref).abs().max()), float((out - ref).pow(2).mean().sqrt()), "s%d.%s" % (stage, name)))
UNetPlan.fast_layers = frozenset()
rows.sort(reverse=True)
for mx, rms, n in rows:
    print("%-16s max-abs %.2e   rms %.2e" % (n, mx, rms))
