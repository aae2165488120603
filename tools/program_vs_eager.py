"""Per-step losses of Trainer(programs=True) beside two eager trainers (the run-to-run noise of the atomics' summation order under Adam):
python tools/program_vs_eager.py [precision] [steps] [size]"""
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


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "f32w"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    dev = torch.device("cuda:0")
    batches = []
    for i in range(steps):
        clips = torch.cat([synthetic_frames(3, S, S, seed=100 + 2 * i), synthetic_frames(3, S, S, seed=101 + 2 * i)], 0).to(dev)
        batches.append((clips[:, [0, 2]].contiguous(), clips[:, 1:2].contiguous(), torch.tensor([0.5, 0.125 * (i % 7 + 1)], device=dev).view(2, 1, 1, 1, 1)))
    hist = {}
    for tag, programs in (("eager A", False), ("eager B", False), ("program", True)):
        ov = synthetic_weight_overrides()
        ov[("STAGE1", "FREEZE")] = "FALSE"
        ov[("STAGE2", "FREEZE")] = "FALSE"
        cfg = load_config("superslomo_original.ini", ov)
        m = FullModel(cfg)
        m.stage1_model.load_state_dict(synthetic_state_dict(1))
        m.stage2_model.load_state_dict(synthetic_state_dict(2))
        m.loss.load_vgg16(synthetic_vgg_state_dict())
        m = m.to(dev).train()
        m.train_precision = mode
        tr = Trainer(m, cfg, programs=programs)
        hist[tag] = [tr.train_step(x, y, t).cpu() for x, y, t in batches]
        if programs:
            pr = tr._prog["program"]
            print("program: %d nodes, %d items (%d host), %d streams" % (pr.n_nodes, len(pr.items), sum(1 for it in pr.items if it[0] == "py"), len(pr.streams)))
    for i in range(steps):
        a, b, c = hist["eager A"][i], hist["eager B"][i], hist["program"][i]
        print("step %d  total %.5f | eager A vs B rel %.2e | program vs eager A rel %.2e" % (i, float(a[0]), float((a - b).abs().max() / a.abs().max()),
                                                                                      float((a - c).abs().max() / a.abs().max())))


if __name__ == "__main__":
    main()
