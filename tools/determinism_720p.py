"""Bitwise repeatability of the headline configuration: the same 720p pair N times through the 2-stream pipeline (both slots, the
convolutions of two pairs running beside each other) - every result must equal the one-stream result bit for bit.
Usage: python tools/determinism_720p.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
for p in (ROOT, PKG, os.path.join(PKG, "scripts")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from models.superslomo_r import FullModel  # noqa: E402
from ssm_amd.config import load_config, synthetic_weight_overrides  # noqa: E402
from ssm_amd.weights import synthetic_frames, synthetic_state_dict  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device("cuda:0")
    m = FullModel(load_config("superslomo_original.ini", synthetic_weight_overrides()))
    m.stage1_model.load_state_dict(synthetic_state_dict(1))
    m.stage2_model.load_state_dict(synthetic_state_dict(2))
    m = m.to(dev).eval()
    x = synthetic_frames(2, 720, 1280, seed=42).to(dev)
    ts = [i / 8.0 for i in range(1, 8)]
    want = m.interpolate(x, ts).clone()
    bad = 0
    for r in range(0, n, 20):
        many = m.interpolate_many([x] * min(20, n - r), ts, n_streams=2)
        bad += sum(0 if torch.equal(g, want) else 1 for g in many)
        del many
    print("720p, %d pairs through the 2-stream pipeline: %d differ from the one-stream result" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
