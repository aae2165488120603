"""Host cost of replaying the training step's launch program: nodes, host items, microseconds per node; beside it the same count of
trivial launches (a 1-pixel ssm_copy_view) recorded and replayed - the floor hipLaunchKernel itself sets on this runtime.
python tools/program_host_cost.py [precision]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "scripts")]
import torch  # noqa: E402

from models.superslomo_r import FullModel  # noqa: E402
from ssm_amd import hipbind as hb  # noqa: E402
from ssm_amd.config import load_config, synthetic_weight_overrides  # noqa: E402
from ssm_amd.perceptual import synthetic_vgg_state_dict  # noqa: E402
from ssm_amd.training import Trainer  # noqa: E402
from ssm_amd.weights import synthetic_frames, synthetic_state_dict  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "f32w"
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
    m.train_precision = mode
    tr = Trainer(m, cfg, programs=True)
    clips = torch.cat([synthetic_frames(3, 352, 352, seed=100 + i) for i in range(2)], 0).to(dev)
    xin, tgt = clips[:, [0, 2]].contiguous(), clips[:, 1:2].contiguous()
    t = torch.tensor([0.5, 0.625], device=dev).view(2, 1, 1, 1, 1)
    for _ in range(5):
        tr.train_step(xin, tgt, t)
    torch.cuda.synchronize()
    prog = tr._prog["program"]
    n_py = sum(1 for it in prog.items if it[0] == "py")
    print("program: %d nodes, %d items (%d host-side), %d streams" % (prog.n_nodes, len(prog.items), n_py, len(prog.streams)))
    for name, fn in (("program.replay()", prog.replay), ("whole train_step()", lambda: tr.train_step(xin, tgt, t))):
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            torch.cuda.synchronize()          # empty queues: nothing blocks the host
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
        print("%-22s host %.3f ms (min %.3f) = %.2f us per node" % (name, 1e3 * sum(ts) / len(ts), 1e3 * min(ts), 1e6 * min(ts) / prog.n_nodes))
    # host items alone
    t0 = time.perf_counter()
    for it in prog.items:
        if it[0] == "py":
            if it[2] is None:
                it[1]()
            else:
                with torch.cuda.stream(it[2]):
                    it[1]()
    print("host items alone       host %.3f ms" % (1e3 * (time.perf_counter() - t0)))
    torch.cuda.synchronize()
    # the floor: the same number of trivial launches
    x, y = hb.Planes(1, 1, 2, 2, dev), hb.Planes(1, 1, 2, 2, dev)
    p2 = hb.LaunchProgram([torch.cuda.current_stream()])
    lib = hb.load()
    with p2.recording():
        for _ in range(prog.n_nodes):
            hb.check(lib.ssm_copy_view(x.view(), y.view(), 1, 1, 2, 2, hb.stream_ptr()))
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p2.replay()
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print("%d trivial launches replayed: host %.3f ms = %.2f us per launch" % (prog.n_nodes, 1e3 * min(ts), 1e6 * min(ts) / prog.n_nodes))
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(prog.n_nodes):
            lib.ssm_copy_view(x.view(), y.view(), 1, 1, 2, 2, hb.stream_ptr())
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print("%d trivial launches from Python: host %.3f ms = %.2f us per launch" % (prog.n_nodes, 1e3 * min(ts), 1e6 * min(ts) / prog.n_nodes))


if __name__ == "__main__":
    main()
