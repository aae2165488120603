#!/usr/bin/env python3
"""Study (GPU, not a test; the oracle is only imported under tests/): parity of the HIP path against the CPU oracle at 736x1280 for other
weight / frame families than the fixtures' (VERDICT r3 item 5, r4 item 1):
weights "uniform" (index-hash He-uniform) | "smooth" (He-normal, windowed 7x7 / 5x5 filters, decoder gain 1.25); frames "texture"
(low-pass texture, 3-px motion) | "edges" (full-contrast rectangles, bars, checkerboards, 28 x 20 px motion).  Beside max|HIP - oracle| it
prints the oracle's OWN fp32 rounding (oracle fp32 vs oracle float64): what two correct fp32 evaluations can differ by on that input.
Modes: any of ssm_amd.engine.MODES, plus the ablation "s1f32" = stage 1 in the direct form (mode f32) + stage 2 in the multiply-saving
forms (mode f32w): the flows, which a step edge amplifies, are then the direct form's.
usage: python tests/study_parity_families.py [modes=f32w,f32,s1f32] [--f64] [--edges-only] [--json out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from oracle import ssm_oracle as O  # noqa: E402
from ssm_amd.engine import PairEngine, UNetPlan  # noqa: E402
from ssm_amd.weights import synthetic_frames, synthetic_state_dict  # noqa: E402


def stats(e):
    e = e.abs().flatten()
    return {"max": float(e.max()), "p9999": float(e.kthvalue(int(e.numel() * 0.9999)).values), "rms": float(e.pow(2).mean().sqrt())}


def fmt(s):
    return "max %.3e  p99.99 %.3e  rms %.3e" % (s["max"], s["p9999"], s["rms"])


def run_mode(mode, sd1, sd2, pair_dev, ts):
    """One pair through a PairEngine in `mode` (or the stage-wise ablation)."""
    B2 = len(ts)
    H, W = pair_dev.shape[-2:]
    eng = PairEngine(sd1, sd2, 1, B2, H, W, pair_dev.device, True, "f32w" if mode == "s1f32" else mode)
    if mode == "s1f32":
        eng.s1 = UNetPlan(1, sd1, 1, H, W, pair_dev.device, True, "f32", True)
    out = eng.run(pair_dev, torch.tensor(ts, device=pair_dev.device), want_aux=False).clone().cpu()
    del eng
    torch.cuda.empty_cache()
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    modes = (args[0] if args else "f32w,f32,s1f32").split(",")
    f64 = "--f64" in sys.argv
    dev = torch.device("cuda:0")
    torch.set_num_threads(16)
    ts = [0.125, 0.5, 0.875]
    fams = (("uniform", "texture"), ("smooth", "texture"), ("uniform", "edges"), ("smooth", "edges"))
    if "--edges-only" in sys.argv:
        fams = fams[2:]
    rec = []
    for fw, ff in fams:
        sd1, sd2 = synthetic_state_dict(1, family=fw), synthetic_state_dict(2, family=fw)
        x = synthetic_frames(2, 720, 1280, seed=42 if ff == "texture" else 7, family=ff)
        pair = torch.cat([x[:, 0], x[:, 1]], 1)
        with torch.no_grad():
            want = torch.cat(O.interpolate_pair(sd1, sd2, pair, ts), 0)
        w64 = None
        if f64:
            with torch.no_grad():
                w64 = torch.cat(O.interpolate_pair({k: v.double() for k, v in sd1.items()}, {k: v.double() for k, v in sd2.items()}, pair.double(), ts), 0)
            s = stats(want.double() - w64)
            print("weights %-8s frames %-8s  oracle fp32 vs float64: %s" % (fw, ff, fmt(s)), flush=True)
            rec.append({"weights": fw, "frames": ff, "what": "oracle_fp32_vs_f64", **s})
        sd1d = {k: v.to(dev) for k, v in sd1.items()}
        sd2d = {k: v.to(dev) for k, v in sd2.items()}
        for mode in modes:
            got = run_mode(mode, sd1d, sd2d, pair.to(dev), ts)
            s = stats(got - want)
            print("weights %-8s frames %-8s  %-6s vs oracle fp32: %s" % (fw, ff, mode, fmt(s)), flush=True)
            rec.append({"weights": fw, "frames": ff, "what": mode + "_vs_oracle_fp32", **s})
            if w64 is not None:
                s = stats(got.double() - w64)
                print("%36s %-6s vs float64:     %s" % ("", mode, fmt(s)), flush=True)
                rec.append({"weights": fw, "frames": ff, "what": mode + "_vs_f64", **s})
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
