"""Is the prelude of a training step (range check of t, input copies: ~10 tiny torch launches before the program's first node) visible in the
step time?  GPU time per step of (a) Trainer.train_step, (b) the recorded program + optimizer.step() alone on the loaded inputs, 30 steps
each, no synchronisation inside the loops.  python tools/program_prelude_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from models.superslomo_r import FullModel  # noqa: E402
from ssm_amd.config import load_config, synthetic_weight_overrides  # noqa: E402
from ssm_amd.perceptual import synthetic_vgg_state_dict  # noqa: E402
from ssm_amd.training import Trainer  # noqa: E402
from ssm_amd.weights import synthetic_frames, synthetic_state_dict  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ov = synthetic_weight_overrides()
    ov[("STAGE1", "FREEZE")] = "FALSE"
    ov[("STAGE2", "FREEZE")] = "FALSE"
    cfg = load_config("superslomo_original.ini", ov)
    m = FullModel(cfg)
    m.stage1_model.load_state_dict(synthetic_state_dict(1))
    m.stage2_model.load_state_dict(synthetic_state_dict(2))
    m.loss.load_vgg16(synthetic_vgg_state_dict())
    m = m.to(dev).train()
    m.train_precision = "f32w"
    tr = Trainer(m, cfg, programs=True)
    clips = torch.cat([synthetic_frames(3, 352, 352, seed=100 + i) for i in range(2)], 0).to(dev)
    xin, tgt = clips[:, [0, 2]].contiguous(), clips[:, 1:2].contiguous()
    t = torch.tensor([0.5, 0.625], device=dev).view(2, 1, 1, 1, 1)
    for _ in range(6):
        tr.train_step(xin, tgt, t)
    pr = tr._prog
    prog = pr["program"]

    def full():
        tr.train_step(xin, tgt, t)

    def bare():
        for p, g in pr["grads"]:
            p.grad = g
        prog.replay()
        tr.optimizer.step()

    for name, fn in (("train_step", full), ("program + optimizer", bare), ("train_step", full), ("program + optimizer", bare)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            fn()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print("%-22s %.3f ms per step (host issue %.3f ms per step)" % (name, 1e3 * el / 30, 1e3 * host / 30), flush=True)


if __name__ == "__main__":
    main()
