#!/usr/bin/env python3
"""Study (GPU, not a test; the oracle is only imported under tests/): parity of the HIP path against the CPU oracle at 736x1280 for other weight / frame families than the fixtures' (VERDICT r3 item 5):
weights "uniform" (index-hash He-uniform) | "smooth" (He-normal, windowed 7x7 / 5x5 filters, decoder gain 1.25); frames "texture"
(low-pass texture, 3-px motion) | "edges" (full-contrast rectangles, bars, checkerboards, 28 x 20 px motion).  Beside max|HIP - oracle| it
prints the oracle's OWN fp32 rounding (oracle fp32 vs oracle float64): what two correct fp32 evaluations can differ by on that input.
usage: python tests/study_parity_families.py [mode=f32w] [--f64]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from models.superslomo_r import FullModel  # noqa: E402
from oracle import ssm_oracle as O  # noqa: E402
from ssm_amd.config import load_config, synthetic_weight_overrides  # noqa: E402
from ssm_amd.weights import synthetic_frames, synthetic_state_dict  # noqa: E402


def stats(e):
    e = e.abs().flatten()
    return "max %.3e  p99.99 %.3e  rms %.3e" % (float(e.max()), float(e.kthvalue(int(e.numel() * 0.9999)).values), float(e.pow(2).mean().sqrt()))


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "f32w"
    f64 = "--f64" in sys.argv
    dev = torch.device("cuda:0")
    torch.set_num_threads(16)
    ts = [0.125, 0.5, 0.875]
    for fw, ff in (("uniform", "texture"), ("smooth", "texture"), ("uniform", "edges"), ("smooth", "edges")):
        sd1, sd2 = synthetic_state_dict(1, family=fw), synthetic_state_dict(2, family=fw)
        m = FullModel(load_config("superslomo_original.ini", synthetic_weight_overrides()))
        m.stage1_model.load_state_dict(sd1)
        m.stage2_model.load_state_dict(sd2)
        m = m.to(dev).eval()
        m.precision = mode
        x = synthetic_frames(2, 720, 1280, seed=42 if ff == "texture" else 7, family=ff)
        pair = torch.cat([x[:, 0], x[:, 1]], 1)
        with torch.no_grad():
            want = torch.cat(O.interpolate_pair(sd1, sd2, pair, ts), 0)
        got = m.interpolate(x.to(dev), ts).cpu()
        print("weights %-8s frames %-8s  %s vs oracle fp32: %s" % (fw, ff, mode, stats(got - want)), flush=True)
        if f64:
            with torch.no_grad():
                w64 = torch.cat(O.interpolate_pair({k: v.double() for k, v in sd1.items()}, {k: v.double() for k, v in sd2.items()}, pair.double(), ts), 0)
            print("%36s oracle fp32 vs float64: %s" % ("", stats(want.double() - w64)), flush=True)
            print("%36s %s vs float64:        %s" % ("", mode, stats(got.double() - w64)), flush=True)
        del m


if __name__ == "__main__":
    main()
