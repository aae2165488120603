"""Multi-GPU plumbing: one process per GPU, frame pairs sharded with NO data-path collective.

Inference shards naturally: every frame pair is independent, its t values stay on
the pair's GPU (they share stage 1).  The reference's torch.nn.DataParallel
(scripts/main.py:74-76, scripts/evaluate_interpolation_results.py:65-67: per-iteration parameter
broadcast + output gather in one process) is replaced by static round-robin assignment of pair
indices to ranks with replicated weights.  torch.distributed (backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests) is used only for the rendezvous, the timing barrier and the
max-over-ranks reduction of the elapsed time.
"""
import os
import time

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend):
    """Join the process group when launched by torch.distributed.run (RANK/WORLD_SIZE/MASTER_* in the environment),
    also at world size 1 so the same barrier / all-reduce path runs everywhere.  A plain `python bench.py` stays
    single-process with no rendezvous."""
    rank, local_rank, world = env_world()
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if launched and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def assign_pairs(n_pairs, world, rank):
    """Static round-robin: pair i -> rank i % world.  Returns this rank's pair indices."""
    return list(range(rank, n_pairs, world))


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def timed_steps(step_fn, steps, warmup, sync_fn):
    """The driver's timing contract: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by
    barrier + device sync on both sides; returns max-over-ranks elapsed seconds."""
    for _ in range(warmup):
        step_fn()
    sync_fn()
    barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    timed_steps.last_enqueue_s = time.perf_counter() - t0      # host time spent issuing the steps (launch-bound if ~= elapsed)
    sync_fn()
    barrier()
    el = time.perf_counter() - t0
    return reduce_max(el)


timed_steps.last_enqueue_s = 0.0


def reduce_max(value, device=None):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or _dist_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_frames(frames_by_index, n_pairs):
    """Collect {pair index: tensor} dicts of all ranks on every rank (host objects; results are
    written/gathered on the host).  Returns a list ordered by pair index."""
    if dist.is_initialized():
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, {k: v.cpu() for k, v in frames_by_index.items()})
    else:
        parts = [frames_by_index]
    merged = {}
    for p in parts:
        merged.update(p)
    assert sorted(merged) == list(range(n_pairs)), "some frame pairs were not processed"
    return [merged[i] for i in range(n_pairs)]


def _dist_device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


# ---- training: the one exchange step of the path -------------------------------------------------------
class GradientAllReduce:
    """Sum-then-average all-reduce of every trainable parameter's gradient in ONE flat fp32 buffer (38,848,553
    floats = 155.4 MB for the two U-Nets) over RCCL/xGMI.  Replaces DataParallel's per-iteration parameter
    broadcast + reduce_add + output gather (scripts/main.py:74-76).  One large collective per step suits xGMI's
    point-to-point links better than many small buckets.  This flat form serves gradients that arrive all at once
    (op-by-op autograd, foreign layouts); the planned training step uses the bucketed, overlapped form (attach)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.bytes = 4 * n
        self.last_runs = None
        self._works, self._bucket_bytes, self.last_buckets = [], 0, 0
        self._exposed_events, self._exposed_host_s = [], 0.0

    # ---- bucketed, overlapped form ----------------------------------------------------------------------
    def attach(self, pair_grad):
        """Hook into the planned backward (ssm_amd.backward.PairGrad): every bucket of a U-Net's flat gradient buffer is
        pre-scaled by 1/world and all-reduced (async, on the collective's own stream, ordered after the weight-gradient
        kernels that were queued before it) the moment its last layer's kernels are queued - the decoder's gradients
        travel over xGMI while the encoder's and the other U-Net's backward still computes.  Few, large buckets (4 per
        U-Net, ~20 MB): xGMI is point-to-point, ring steps are per-link bound, small messages waste it."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        # $SSM_FORCE_ALLREDUCE=1: run the bucketed exchange also in a process group of ONE rank (the collective is then an identity
        # on RCCL) - the only way to exercise the RCCL path of the training step on a one-GPU box
        forced = dist.is_initialized() and os.environ.get("SSM_FORCE_ALLREDUCE", "0") != "0"
        pair_grad.sync = self if (world > 1 or forced) else None
        pair_grad.sync_scale = 1.0 / world

    def reduce(self, view):
        """One completed bucket (already scaled by 1/world): start its sum all-reduce, in place."""
        self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))
        self._bucket_bytes += 4 * view.numel()

    def _finish_buckets(self):
        """Order the compute stream behind the bucket all-reduces of this step.  On the GPU nothing here blocks the HOST (r5: a stream
        synchronise at this point - there only to put a number on the wait - stopped the host's run-ahead once per step and made the
        exchange cost 2 ms of a 20 ms step even in a world of one): `Work.wait()` of the RCCL backend makes the current stream wait,
        and the exposed part - the time the compute stream sat behind the collectives - is bracketed by two events, read later through
        exposed_seconds().  Returns the seconds the caller was blocked (CPU / gloo: the whole wait; GPU: none)."""
        cuda = self.flat.is_cuda
        t0 = time.perf_counter()
        if cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self._works:
            w.wait()
        if cuda:
            e1.record()
            self._exposed_events.append((e0, e1))
            if len(self._exposed_events) > 4096:          # a long run that never reads the figure
                del self._exposed_events[:2048]
        # RCCL's Work.wait() only orders streams; every other backend (gloo, also on device tensors) blocks the caller inside wait()
        host_blocked = not (cuda and dist.get_backend() == "nccl")
        el = time.perf_counter() - t0 if host_blocked else 0.0
        self._exposed_host_s += el
        self.last_buckets, self._works = len(self._works), []
        covered, self._bucket_bytes = self._bucket_bytes, 0
        assert covered == self.bytes, "bucketed all-reduce covered %d of %d gradient bytes" % (covered, self.bytes)
        return el

    def exposed_seconds(self, reset=True):
        """Seconds the compute stream (GPU: event brackets, synchronised here) or the caller (CPU) waited for the bucketed exchange
        since the last call."""
        total = self._exposed_host_s
        for e0, e1 in self._exposed_events:
            e1.synchronize()
            total += 1e-3 * e0.elapsed_time(e1)
        if reset:
            self._exposed_events, self._exposed_host_s = [], 0.0
        return total

    MAX_RUNS = 8     # in-place path: the gradients form at most this many contiguous memory runs

    def runs(self):
        """Contiguous memory runs of the gradients in parameter order, as 1-D tensors aliasing them - or None if some gradient
        is missing / strided or there are more than MAX_RUNS runs.  The planned training step writes every gradient of a U-Net
        into one flat buffer (ssm_amd.backward.UNetGrad.flat), so the two U-Nets are two runs."""
        out, end = [], None
        for p in self.params:
            g = p.grad
            if g is None or not g.is_contiguous() or g.dtype != torch.float32:
                return None
            if out and g.untyped_storage().data_ptr() == out[-1][0].untyped_storage().data_ptr() and g.data_ptr() == end:
                out[-1][1] += g.numel()
            else:
                out.append([g, g.numel()])
                if len(out) > self.MAX_RUNS:
                    return None
            end = g.data_ptr() + 4 * g.numel()
        return [torch.as_strided(g, (n,), (1,)) for g, n in out]

    def __call__(self):
        """Average the gradients over the ranks (no-op when not distributed).  Returns seconds spent (host clock
        around an explicitly synchronised region when on a GPU).  Gradients that already sit in a few flat buffers are
        reduced in place (no gather / scatter copies: 2 x 96 launches per step); otherwise through one staging buffer."""
        if not dist.is_initialized() or (dist.get_world_size() == 1 and not self._works):
            return 0.0
        if self._works:          # the backward handed its buckets over as they completed (attach): only wait
            return self._finish_buckets()
        cuda = self.flat.is_cuda
        runs = self.runs()
        self.last_runs = None if runs is None else len(runs)
        if runs is not None:
            if cuda:
                torch.cuda.synchronize(self.flat.device)
            t0 = time.perf_counter()
            for r in runs:
                dist.all_reduce(r, op=dist.ReduceOp.SUM)
            if cuda:
                torch.cuda.synchronize(self.flat.device)
            el = time.perf_counter() - t0
            self._exposed_host_s += el          # (the non-bucketed paths block the host: their wait is all exposed)
            for r in runs:
                r.div_(dist.get_world_size())
            return el
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.grad.reshape(-1) if p.grad is not None else torch.zeros(n, device=self.flat.device))
            off += n
        if cuda:
            torch.cuda.synchronize(self.flat.device)
        t0 = time.perf_counter()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if cuda:
            torch.cuda.synchronize(self.flat.device)
        el = time.perf_counter() - t0
        self._exposed_host_s += el
        self.flat.div_(dist.get_world_size())
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n
        return el
