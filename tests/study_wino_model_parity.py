"""Study (GPU box, not collected by pytest): whole-model parity of modes f32 / f32w against the CPU oracle at 64x64 and 736x1280."""
import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/superslomo-videointerpolation-pytorch_amd', '/root/repo/superslomo-videointerpolation-pytorch_amd/scripts']
from models.superslomo_r import FullModel
from oracle import ssm_oracle as O
from ssm_amd.config import load_config, synthetic_weight_overrides
from ssm_amd.weights import synthetic_frames, synthetic_state_dict
dev = torch.device("cuda:0")
cfg = load_config("superslomo_original.ini", synthetic_weight_overrides())
model = FullModel(cfg)
sd1, sd2 = synthetic_state_dict(1), synthetic_state_dict(2)
model.stage1_model.load_state_dict(sd1); model.stage2_model.load_state_dict(sd2)
model = model.to(dev).eval()
for (H, W, ts) in ((64, 64, [0.25, 0.5, 0.75]), (736, 1280, [0.125, 0.5])):
    x = synthetic_frames(2, H, W, seed=7)
    pair = torch.cat([x[:, 0], x[:, 1]], 1)
    want = torch.cat(O.interpolate_pair(sd1, sd2, pair, ts), 0)
    outs = {}
    for mode in ("f32", "f32w"):
        model.precision = mode
        got = model.interpolate(x.to(dev), ts).cpu()
        outs[mode] = got
        print("%dx%d %s: max|HIP - oracle| = %.3e" % (H, W, mode, (got - want).abs().max().item()), flush=True)
    print("   f32w vs f32: %.3e" % (outs["f32"] - outs["f32w"]).abs().max().item(), flush=True)
