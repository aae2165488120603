"""The RCCL code path on the one GPU a test box has: `bench.py` started as a FRESH child process with the torchrun environment of a
one-rank job (RANK=0, WORLD_SIZE=1, MASTER_*), so dist.init("nccl"), the device barrier, the max-over-ranks reduction and - in
train mode - every bucket of the overlapped gradient all-reduce run on RCCL.  This is the multi-GPU evidence available without an
8-GPU node (the N > 1 plumbing itself is covered by the world-size-2 gloo tests in tests/test_dist_gloo.py).  Replaces the
reference's torch.nn.DataParallel (scripts/main.py:74-76)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _one_rank_env():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=_one_rank_env(), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_inference_bench_as_a_one_rank_rccl_job():
    out = _bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--modes", "", "--no-cpu-baseline", "--no-io", "--no-kernel-timers")
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["dist"] == {"backend": "nccl", "world_size": 1}, out.get("dist")


def test_training_bench_all_reduces_every_gradient_bucket_on_rccl():
    out = _bench("--gpus", "1", "--mode", "train", "--steps", "3", "--warmup", "2", "--no-perceptual", "--force-allreduce")
    ar = out["allreduce"]
    assert ar["backend"] == "nccl"
    assert ar["bytes"] == 4 * 38848553                      # 155.4 MB of fp32 gradients per step
    assert ar["buckets_per_step"] == 8                      # 4 buckets per U-Net, handed over as the backward completes them
    # event-bracketed wait of the compute stream behind the collectives: one bracket per timed step was read (a dead measurement
    # reports 0 brackets), and two events on one stream are never 0 apart; _finish_buckets asserts the buckets covered every byte
    assert ar["brackets"] == 3 and ar["ms_per_step"] > 0.0, ar
    assert out["value"] > 0
