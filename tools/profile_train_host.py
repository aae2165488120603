#!/usr/bin/env python3
"""Where does the HOST time of a training step go?  cProfile over N steps of bench.py's config-3 step (2 x 352x352, full loss),
plus the count of C-ABI launches per step.  usage: python tools/profile_train_host.py [precision] [steps]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from ssm_amd import hipbind as hb  # noqa: E402
from ssm_amd.config import load_config, synthetic_weight_overrides  # noqa: E402
from ssm_amd.perceptual import synthetic_vgg_state_dict  # noqa: E402
from ssm_amd.training import Trainer  # noqa: E402
from ssm_amd.weights import synthetic_frames, synthetic_state_dict  # noqa: E402
from models.superslomo_r import FullModel  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "f32w"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    ov = synthetic_weight_overrides()
    ov[("STAGE1", "FREEZE")] = "FALSE"
    ov[("STAGE2", "FREEZE")] = "FALSE"
    cfg = load_config("superslomo_original.ini", ov)
    model = FullModel(cfg)
    model.stage1_model.load_state_dict(synthetic_state_dict(1))
    model.stage2_model.load_state_dict(synthetic_state_dict(2))
    model.loss.load_vgg16(synthetic_vgg_state_dict())
    model = model.to(dev).train()
    model.train_precision = mode
    trainer = Trainer(model, cfg)
    clips = torch.cat([synthetic_frames(3, 352, 352, seed=100 + i) for i in range(2)], 0).to(dev)
    xin, tgt = clips[:, [0, 2]].contiguous(), clips[:, 1:2].contiguous()
    t = torch.tensor([0.5, 0.625], device=dev).view(2, 1, 1, 1, 1)
    for _ in range(3):
        trainer.train_step(xin, tgt, t)
    torch.cuda.synchronize()
    # count C-ABI calls per step
    lib = hb.load()
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.train_step(xin, tgt, t)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print("mode %s: host enqueue %.2f ms/step, wall %.2f ms/step" % (mode, 1e3 * host / steps, 1e3 * wall / steps))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        trainer.train_step(xin, tgt, t)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(35)
    st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
