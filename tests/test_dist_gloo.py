"""world_size-2 gloo test of the multi-GPU plumbing (ssm_amd.dist): pair sharding covers every pair
exactly once, the gathered N-rank result equals the 1-rank result, and the bench timing protocol
(barrier-bracketed, max over ranks) behaves.  CPU only - the per-pair work is a stand-in function;
the GPU path's per-pair determinism is covered by tests/test_hip_model.py::test_720p_properties."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd"))


def fake_interpolate(i):
    g = torch.Generator().manual_seed(100 + i)
    return torch.randn(7, 3, 4, 6, generator=g)


def worker(rank, world, port, n_pairs, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ssm_amd import dist as sd
    r, lr, w = sd.init("gloo")
    assert (r, w) == (rank, world)
    mine = sd.assign_pairs(n_pairs, world, rank)
    frames = {i: fake_interpolate(i) for i in mine}
    allf = sd.gather_frames(frames, n_pairs)
    ok = all(torch.equal(allf[i], fake_interpolate(i)) for i in range(n_pairs))
    # timing protocol: rank 1 is slower; every rank must report the max
    el = sd.timed_steps(lambda: time.sleep(0.01 * (rank + 1)), steps=5, warmup=1, sync_fn=lambda: None)
    ret[rank] = (mine, ok, el)
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharding_and_timing():
    world, n_pairs = 2, 7
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, free_port(), n_pairs, ret), nprocs=world, join=True)
    assert sorted(ret[0][0] + ret[1][0]) == list(range(n_pairs))
    assert set(ret[0][0]).isdisjoint(ret[1][0])
    assert ret[0][1] and ret[1][1]
    assert abs(ret[0][2] - ret[1][2]) < 1e-9 and ret[0][2] >= 0.1     # both see rank 1's 5 x 20 ms


def test_single_process_defaults():
    from ssm_amd import dist as sd
    assert sd.assign_pairs(5, 1, 0) == [0, 1, 2, 3, 4]
    assert sd.reduce_max(1.5) == 1.5
    out = sd.gather_frames({0: torch.zeros(1), 1: torch.ones(1)}, 2)
    assert len(out) == 2


def _ar_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ssm_amd import dist as sd
    sd.init("gloo")
    g = torch.Generator().manual_seed(5)
    params = [torch.nn.Parameter(torch.randn(7, 3, generator=g)), torch.nn.Parameter(torch.randn(11, generator=g)),
              torch.nn.Parameter(torch.randn(2, generator=g), requires_grad=False)]
    params[0].grad = torch.full((7, 3), float(rank + 1))
    params[1].grad = torch.arange(11, dtype=torch.float32) * (rank + 1)
    ar = sd.GradientAllReduce(params)
    ar()
    ret[rank] = (params[0].grad.clone(), params[1].grad.clone(), ar.bytes)
    dist.destroy_process_group()


def test_gradient_allreduce_averages_over_ranks():
    """All-reduced gradients == mean of the per-rank gradients (the build's multi-GPU check)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ar_worker, args=(2, free_port(), ret), nprocs=2, join=True)
    for r in (0, 1):
        assert torch.allclose(ret[r][0], torch.full((7, 3), 1.5))
        assert torch.allclose(ret[r][1], torch.arange(11, dtype=torch.float32) * 1.5)
        assert ret[r][2] == 4 * (21 + 11)


def _ar_flat_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ssm_amd import dist as sd
    sd.init("gloo")
    shapes = [(4, 3, 3, 3), (4,), (2, 4, 3, 3), (2,)]
    params = [torch.nn.Parameter(torch.zeros(sh)) for sh in shapes]
    # the layout of the planned training step: one flat buffer per U-Net, .grad = slices of it (two buffers = two runs)
    flats = [torch.arange(108 + 4, dtype=torch.float32) * (rank + 1), torch.arange(72 + 2, dtype=torch.float32) * (rank + 3)]
    params[0].grad, params[1].grad = flats[0][:108].view(shapes[0]), flats[0][108:].view(shapes[1])
    params[2].grad, params[3].grad = flats[1][:72].view(shapes[2]), flats[1][72:].view(shapes[3])
    ar = sd.GradientAllReduce(params)
    ar()
    ret[rank] = (ar.last_runs, [f.clone() for f in flats], params[2].grad.data_ptr() == flats[1].data_ptr())
    dist.destroy_process_group()


def test_gradient_allreduce_in_place_on_flat_buffers():
    """Gradients that are slices of flat buffers (ssm_amd.backward.UNetGrad.flat) are all-reduced in place, one collective per
    buffer, and stay slices of those buffers."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ar_flat_worker, args=(2, free_port(), ret), nprocs=2, join=True)
    for r in (0, 1):
        runs, flats, aliased = ret[r]
        assert runs == 2 and aliased
        assert torch.allclose(flats[0], torch.arange(112, dtype=torch.float32) * 1.5)
        assert torch.allclose(flats[1], torch.arange(74, dtype=torch.float32) * 3.5)


def _run_bench(*flags):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), env=env, capture_output=True, text=True,
                          timeout=300)


def test_bench_bare_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun): the parent starts 2 rank processes, relays exactly one JSON line with
    n_gpus = 2; the real argument path with the per-pair work stubbed (--stub, gloo) since this container has no GPU."""
    import json
    r = _run_bench("--gpus", "2", "--stub", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["value"] > 0
    # rank 1 sleeps 2 ms per pair, rank 0 1 ms: the line must carry the max over ranks (8 pairs x 2 ms)
    assert out["ms_per_step"] >= 16.0
    assert "stub" in out["data"]


def test_bench_eight_ranks_stub_launch():
    """The real `python bench.py --gpus 8` argument path at world 8 (--stub: gloo, a sleep as the per-pair work): one
    JSON line, n_gpus 8, the max over ranks (rank 7 sleeps 8 ms per pair), and every rank pinned to its own 1/8 of the usable CPUs."""
    import json
    r = _run_bench("--gpus", "8", "--stub", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["ms_per_step"] >= 8 * 8.0                     # 8 pairs x 8 ms on the slowest rank
    n = len(os.sched_getaffinity(0))
    if n >= 8:
        assert len(out["host"]["cpus_of_this_rank"]) == n // 8 and out["host"]["usable_cpus"] == n // 8
    # weak scaling: every rank ran 8 pairs -> 8 x 8 x 7 frames per step
    assert abs(out["value"] - 7 * 8 * 8 / (out["ms_per_step"] * 1e-3)) < 0.02 * out["value"]


def test_bench_eight_ranks_training_stub_exchanges_every_gradient_bucket():
    """`python bench.py --gpus 8 --mode train --stub`: the TRAINING launcher path at world 8 over gloo - the two
    U-Nets' flat gradient buffers in their true layout (155.4 MB), 4 buckets each handed to GradientAllReduce tail first, 1/world
    pre-scale, every rank checking the average it receives - with synthetic gradients (the real backward needs the GPUs)."""
    import json
    r = _run_bench("--gpus", "8", "--mode", "train", "--stub", "--steps", "1", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["unit"] == "samples/s" and out["config"]["global_batch"] == 16 and out["scaling"] == "weak"
    ar = out["allreduce"]
    assert ar["bytes"] == 4 * 38848553 and ar["buckets_per_step"] == 8 and ar["backend"] == "gloo"
    assert ar["max_abs_err_of_the_average"] < 1e-5 and ar["ms_per_step"] > 0.0          # (gloo blocks the host in wait(): exposed, and booked)


def test_gradient_layout_is_a_pure_function_of_the_layer_table():
    """ssm_amd.backward.grad_layout (shared by the GPU backward and the stub above): both U-Nets' keys in state-dict order, spans that
    tile the flat buffer, 4 buckets of consecutive layers that cover it."""
    from ssm_amd.backward import grad_layout
    from ssm_amd.weights import synthetic_state_dict, unet_layers
    total = 0
    for stage in (1, 2):
        layers = {n: (ci, co, k) for n, ci, co, k in unet_layers(stage, True)}
        sizes, span, buckets = grad_layout(layers, 4)
        sd = synthetic_state_dict(stage)
        assert [k for k, _ in sizes] == list(sd.keys()) and all(tuple(sd[k].shape) == tuple(sh) for k, sh in sizes)
        n = sum(sd[k].numel() for k in sd)
        ends = sorted(span.values())
        assert ends[0][0] == 0 and ends[-1][1] == n and all(a[1] == b[0] for a, b in zip(ends, ends[1:]))
        assert len(buckets) == 4 and [x for b in buckets for x in b] == list(layers)
        total += n
    assert total == 38848553


def test_bench_eight_ranks_one_failing_rank_fails_the_launch():
    """A rank that exits with an error (here before the rendezvous) ends the launch promptly: non-zero exit, no JSON line."""
    import time as _t
    t0 = _t.time()
    r = _run_bench("--gpus", "8", "--stub", "--stub-fail-rank", "5", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and "rank(s) failed" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert _t.time() - t0 < 120


def test_bench_launcher_fails_loudly_when_a_rank_fails():
    """Without a GPU the real ranks assert; the parent must exit non-zero and print no JSON line."""
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a GPU-less host")
    r = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def _ar_bucket_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ssm_amd import dist as sd
    sd.init("gloo")
    shapes = [(4, 3, 3, 3), (4,), (2, 4, 3, 3), (2,), (5, 2, 3, 3), (5,)]
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(sum(int(torch.Size(s).numel()) for s in shapes), generator=g)

    def make():
        params, flat, off = [torch.nn.Parameter(torch.zeros(s)) for s in shapes], local.clone(), 0
        for p in params:
            p.grad = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        return params, flat
    # flat form: one in-place all-reduce of the whole buffer after the backward
    params, flat = make()
    sd.GradientAllReduce(params)()
    # bucketed form: the backward hands over buckets tail-first (decoder before encoder), each pre-scaled by 1/world
    params2, flat2 = make()
    ar = sd.GradientAllReduce(params2)

    class FakePairGrad:
        sync, sync_scale = None, 1.0
    pg = FakePairGrad()
    ar.attach(pg)
    assert pg.sync is ar and pg.sync_scale == 1.0 / world
    bounds = [0, 112, 186, flat2.numel()]
    for a, b in reversed(list(zip(bounds[:-1], bounds[1:]))):
        v = flat2[a:b]
        v.mul_(pg.sync_scale)
        pg.sync.reduce(v)
    ar()
    # the exposed wait of the step is accumulated until it is read (CPU / gloo: the host seconds of Work.wait(); GPU: event brackets,
    # nothing blocks the host) and reset by the read
    waited = ar.exposed_seconds()
    assert waited >= 0.0 and ar.exposed_seconds() == 0.0
    ret[rank] = (flat.clone(), flat2.clone(), ar.last_buckets, params2[4].grad.data_ptr() == flat2[186:].data_ptr())
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_equals_flat():
    """the gradient exchange in buckets started during the backward (GradientAllReduce.attach / reduce) gives
    the same averaged gradients as one flat all-reduce after it, in place, and refuses to finish if a bucket is missing."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ar_bucket_worker, args=(2, free_port(), ret), nprocs=2, join=True)
    for r in (0, 1):
        flat, bucketed, n, aliased = ret[r]
        assert n == 3 and aliased
        assert torch.allclose(flat, bucketed, rtol=0, atol=1e-7)
    assert torch.equal(ret[0][1], ret[1][1])
