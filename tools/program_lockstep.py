"""Eager and program trainers in lockstep (one process, same init, same batches): per step the loss difference, the largest gradient
difference relative to the largest gradient, and the largest parameter difference.  python tools/program_lockstep.py [precision] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from models.superslomo_r import FullModel  # noqa: E402
from ssm_amd.config import load_config, synthetic_weight_overrides  # noqa: E402
from ssm_amd.perceptual import synthetic_vgg_state_dict  # noqa: E402
from ssm_amd.training import Trainer  # noqa: E402
from ssm_amd.weights import synthetic_frames, synthetic_state_dict  # noqa: E402


def make(dev, mode, programs, vgg):
    ov = synthetic_weight_overrides()
    ov[("STAGE1", "FREEZE")] = "TRUE" if os.environ.get("FREEZE1") else "FALSE"          # FREEZE1=1: only stage 2 trains
    ov[("STAGE2", "FREEZE")] = "FALSE"
    cfg = load_config("superslomo_original.ini", ov)
    m = FullModel(cfg)
    m.stage1_model.load_state_dict(synthetic_state_dict(1))
    m.stage2_model.load_state_dict(synthetic_state_dict(2))
    if vgg:
        m.loss.load_vgg16(synthetic_vgg_state_dict())
    m = m.to(dev).train()
    m.train_precision = mode
    tr = Trainer(m, cfg, programs=programs)
    if os.environ.get("LR0"):          # frozen parameters: every step's gradients are comparable to atomics noise
        for g in tr.optimizer.param_groups:
            g["lr"] = 0.0
    return m, tr


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    vgg = (sys.argv[3] if len(sys.argv) > 3 else "1") != "0"
    dev = torch.device("cuda:0")
    mE, tE = make(dev, mode, False, vgg)
    mP, tP = make(dev, mode, True, vgg)
    for i in range(steps):
        clips = torch.cat([synthetic_frames(3, 64, 64, seed=100 + 2 * i), synthetic_frames(3, 64, 64, seed=101 + 2 * i)], 0).to(dev)
        x, y, t = clips[:, [0, 2]].contiguous(), clips[:, 1:2].contiguous(), torch.tensor([0.5, 0.125 * (i % 7 + 1)], device=dev).view(2, 1, 1, 1, 1)
        lE, lP = tE.train_step(x, y, t), tP.train_step(x, y, t)
        torch.cuda.synchronize()
        gmax, worst, wname = 0.0, 0.0, ""
        pairs = [(n, a, b) for (n, a), (_, b) in zip(mE.named_parameters(), mP.named_parameters()) if a.grad is not None]
        for n, a, b in pairs:
            gmax = max(gmax, float(a.grad.abs().max()))
        for n, a, b in pairs:
            d = float((a.grad - b.grad).abs().max()) / gmax
            if d > worst:
                worst, wname = d, n
        pd = max(float((a - b).abs().max()) for a, b in zip(mE.parameters(), mP.parameters()))
        print("step %d [%s] loss %.6f  dloss rel %.2e | worst grad diff / max grad %.2e (%s) | max param diff %.2e"
              % (i, "replay" if tP._prog is not None and i > tP.PROGRAM_WARMUP else ("record" if tP._prog is not None else "eager"), float(lE[0]),
                 float((lE - lP).abs().max() / lE.abs().max()), worst, wname, pd))


if __name__ == "__main__":
    main()
