"""Trainer around the hot path: the optimisation recipe and checkpoint format of the reference's scripts/main.py
(Adam + StepLR :255-262, `losses.mean(0)[0]` as the scalar loss :138-144, checkpoint dict :218-245), one process per
GPU with an RCCL gradient all-reduce instead of torch.nn.DataParallel (:74-76).  Datasets are out of scope (SURVEY
section 2, #11): `train` consumes any iterable of (input [B,2,3,H,W], target [B,1,3,H,W], t_interp [B,1,1,1,1])
batches already on the device."""
import logging
import os

import torch

from .dist import GradientAllReduce

log = logging.getLogger(__name__)


class Trainer:
    """graphs (default: $SSM_TRAIN_GRAPH, off): forward + hand-written backward of a step are captured once into a HIP graph
    (torch.cuda.CUDAGraph; the C-ABI launches and the side-stream fork/join of the parameter gradients land on the capturing
    stream) and replayed from static input buffers; the gradient all-reduce and Adam stay outside it.  Off by default: on
    ROCm 7.2 the replay of the ~900-node graph costs the host as much as issuing the launches (17.8 ms vs 15.6 ms per step
    measured), so the step got slower (20.3 vs 18.0 ms); the eager step's launch count is what was cut instead."""

    def __init__(self, model, cfg, graphs=None, programs=None):
        self.model, self.cfg = model, cfg
        self.graphs = (os.environ.get("SSM_TRAIN_GRAPH", "0") != "0") if graphs is None else bool(graphs)
        self._graph = None
        # programs (default on; $SSM_TRAIN_PROGRAM=0: off): the planned step's ~900 C-ABI launches are recorded once into a launch program
        # (csrc/ssm_program.cpp, hipbind.LaunchProgram) and replayed by a handful of host calls - the step then costs the host ~2 us per
        # launch instead of the interpreter + ctypes price of each (11 of 15 ms).  Unlike the HIP graph above the gradient exchange stays
        # INSIDE the step (its buckets are host-side items of the program, overlapped with the backward as in the eager step).
        self.programs = (os.environ.get("SSM_TRAIN_PROGRAM", "1") != "0") if programs is None else bool(programs)
        self._prog, self._prog_seen = None, {}
        self.learning_rate = cfg.getfloat("TRAIN", "LEARNING_RATE")
        self.lr_period = cfg.getint("TRAIN", "LR_PERIOD")
        self.lr_decay = cfg.getfloat("TRAIN", "LR_DECAY")
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("nothing to train: both stages have FREEZE=TRUE (as the reference's ini files ship); set "
                             "STAGE1.FREEZE / STAGE2.FREEZE to FALSE")
        # same update rule as the reference's torch.optim.Adam (scripts/main.py:255-257); on the GPU the fused multi-tensor
        # implementation (2 launches instead of ~20 per step)
        fused = params[0].is_cuda and os.environ.get("SSM_FUSED_ADAM", "1") != "0"
        self.optimizer = torch.optim.Adam(params, lr=self.learning_rate, **({"fused": True} if fused else {}))
        self.lr_scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=self.lr_period, gamma=self.lr_decay)
        self.allreduce = GradientAllReduce(params)
        # the planned backward hands its gradient buckets over as they complete (not under graph capture: the collective stays outside)
        model.grad_sync = None if self.graphs else self.allreduce
        self.last_allreduce_s = 0.0
        self.start = 1
        self.configure_resume()

    def configure_resume(self):
        """Resume of scripts/main.py:263-284: when a stage is loaded from a checkpoint (LOADPREV) AND trains (not FREEZE), the
        optimizer state, the scheduler state and the epoch counter come from that checkpoint ("self.optimizer", "scheduler",
        "epoch" of the dict save_model writes); stage 1's checkpoint wins when both qualify.  The weights themselves were
        loaded by FullModel.load_weights (models.unetflow.get_model)."""
        cfg = self.cfg
        load_1, load_2 = cfg.getboolean("STAGE1", "LOADPREV"), cfg.getboolean("STAGE2", "LOADPREV")
        freeze_1, freeze_2 = cfg.getboolean("STAGE1", "FREEZE"), cfg.getboolean("STAGE2", "FREEZE")
        if load_1 and not freeze_1:
            ckpt_path = cfg.get("STAGE1", "WEIGHTS")
        elif load_2 and not freeze_2:
            ckpt_path = cfg.get("STAGE2", "WEIGHTS")
        else:
            return
        device = next(p for p in self.model.parameters()).device
        checkpoint = torch.load(ckpt_path, map_location=device)
        self.optimizer.load_state_dict(checkpoint["self.optimizer"])
        self.lr_scheduler.load_state_dict(checkpoint["scheduler"])
        self.start = max(checkpoint["epoch"], 1)
        log.info("Starting training from: %s", self.start)
        log.info("Scheduler: %s", self.lr_scheduler.get_last_lr())
        for param_group in self.optimizer.param_groups:
            log.info("Learning rate: %s", param_group["lr"])

    def train_step(self, input_images, target_images, t_interp, iteration=None):
        """forward_pass + backward + optimizer step (scripts/main.py:116-145,188-197).  Returns the [4] loss vector
        (total, reconstruction, warp, perceptual), batch-averaged."""
        n = self.cfg.getint("TRAIN", "N_FRAMES")
        assert input_images.shape[1] == n and target_images.shape[1] == t_interp.shape[1] == n - 1
        in_range = ((t_interp > 0) & (t_interp < 1)).all()
        if t_interp.is_cuda:
            torch._assert_async(in_range)          # device-side check: no host sync in the step loop
        else:
            assert bool(in_range), "Interpolation values out of bounds."
        if self.graphs and input_images.is_cuda and not self._timers_on():
            losses = self._replay(input_images, target_images, t_interp, iteration)
        elif self.programs and self._program_ok(input_images):
            losses = self._program_step(input_images, target_images, t_interp, iteration)
            self.last_allreduce_s = self.allreduce()
            self.optimizer.step()
            return losses.detach().clone()          # (the program writes its loss vector in place every step)
        else:
            self.optimizer.zero_grad()
            losses = self._forward_backward(input_images, target_images, t_interp, iteration)
        self.last_allreduce_s = self.allreduce()
        self.optimizer.step()
        return losses.detach()

    def _forward_backward(self, input_images, target_images, t_interp, iteration):
        _, losses = self.model(input_images, t_interp, target_images=target_images, iteration=iteration, inference_mode=False)
        losses = losses.mean(dim=0)
        losses[0].backward()
        return losses

    # ---- launch programs ---------------------------------------------------------------------------------------------------------
    PROGRAM_WARMUP = 2          # eager steps of a shape before its program is recorded (plans, packed filters, lazily built buffers)

    def _program_ok(self, input_images):
        """The planned one-window step (models.superslomo_r._TrainStep) on the GPU, fp32 plans, no event timers."""
        m = self.model
        if not (input_images.is_cuda and input_images.shape[1] == 2 and not m.recurrent and not self._timers_on()):
            return False
        mode = m.train_precision or os.environ.get("SSM_TRAIN_PRECISION", "f32")
        if mode not in ("f32", "f32w") or (m.loss.feature_extractor is not None) or os.environ.get("SSM_FUSED_LOSS", "1") == "0":
            return False
        return all(p.requires_grad for p in m.stage1_model.parameters()) or all(p.requires_grad for p in m.stage2_model.parameters())

    def _program_step(self, input_images, target_images, t_interp, iteration):
        # (the caller's current stream is slot 0 of the program: a step issued under another stream context records its own)
        from . import hipbind as hb
        key = (tuple(input_images.shape), str(input_images.device), torch.cuda.current_stream().cuda_stream)
        pr = self._prog
        if pr is not None and (pr["key"] != key or pr["train"] is not getattr(self.model, "_train", None)):
            pr = self._prog = None          # another shape, or the plans were dropped (load_state_dict): record again
        if pr is None:
            seen = self._prog_seen.get(key, 0)
            if seen < self.PROGRAM_WARMUP:
                self._prog_seen[key] = seen + 1
                self.optimizer.zero_grad()
                return self._forward_backward(input_images, target_images, t_interp, iteration)
            try:
                pr = self._record_program(key, input_images, target_images, t_interp)
            except hb.ProgramBusy:          # another thread of this process is recording (one at a time): this step eagerly, record later
                self.optimizer.zero_grad()
                return self._forward_backward(input_images, target_images, t_interp, iteration)
        else:
            self._load_inputs(pr, input_images, target_images, t_interp)
            for p, g in pr["grads"]:          # (an eager step in between drops the references: the program writes these tensors)
                p.grad = g
            pr["program"].replay()
        return pr["losses"]

    @staticmethod
    def _load_inputs(pr, input_images, target_images, t_interp):
        pr["img6"][:, 0:3].copy_(input_images[:, 0])
        pr["img6"][:, 3:6].copy_(input_images[:, 1])
        pr["target"].copy_(target_images[:, 0])
        pr["t"].copy_(t_interp.reshape(-1))

    def _record_program(self, key, input_images, target_images, t_interp):
        """Run one step through the planned forward and the hand-written backward DIRECTLY (models.superslomo_r._TrainStep's two halves,
        no autograd graph: the upstream gradient of `losses.mean(0)[0]` is the constant 1/B in column 0) while a LaunchProgram records,
        from static input buffers."""
        import types

        from models.superslomo_r import _TrainStep

        from . import hipbind as hb
        model = self.model
        B, _, _, H, W = input_images.shape
        dev = input_images.device
        pr = {"key": key, "img6": torch.empty(B, 6, H, W, dtype=torch.float32, device=dev),
              "target": torch.empty(B, 3, H, W, dtype=torch.float32, device=dev), "t": torch.empty(B, dtype=torch.float32, device=dev),
              "losses": torch.empty(4, dtype=torch.float32, device=dev)}
        self._load_inputs(pr, input_images, target_images, t_interp)
        eng, pg = model._train_engine(B, H, W, dev)
        pr["train"] = model._train
        streams = [torch.cuda.current_stream()]
        for st in [pg.u1.side, getattr(model.loss.perceptual_term(B, H, W, dev), "_side", None)] + list(pg.u1.more_sides):
            if st is not None and all(st.cuda_stream != x.cuda_stream for x in streams):
                streams.append(st)
        d_losses = torch.zeros(B, 4, dtype=torch.float32, device=dev)
        d_losses[:, 0] = 1.0 / B
        params = list(model.stage1_model.parameters()) + list(model.stage2_model.parameters())
        self.optimizer.zero_grad(set_to_none=True)          # the recorded backward then ASSIGNS .grad: slices of the plan's flat buffers
        ctx = types.SimpleNamespace(mark_non_differentiable=lambda *a: None)
        prog = hb.LaunchProgram(streams)
        with torch.no_grad(), prog.recording():
            _, losses = _TrainStep.forward(ctx, model, pr["img6"], pr["t"], pr["target"], *params)
            out = pr["losses"]
            hb.host_op(lambda: out.copy_(losses.mean(dim=0)))
            _TrainStep.backward(ctx, None, d_losses)
        pr["grads"] = [(p, p.grad) for p in params if p.requires_grad and p.grad is not None]
        pr["program"] = prog
        self._prog = pr
        log.info("training step recorded into a launch program: %d C-ABI nodes, %d items, %d stream(s), input %s", prog.n_nodes,
                 len(prog.items), len(streams), key[0])
        return pr

    @staticmethod
    def _timers_on():
        """Per-kernel event timers (bench.py's time split) record timing events: those steps run eagerly."""
        from .engine import UNetPlan
        from .perceptual import VGGFeatures
        return UNetPlan.timer is not None or VGGFeatures.timer is not None

    def _replay(self, input_images, target_images, t_interp, iteration):
        key = (tuple(input_images.shape), tuple(target_images.shape), tuple(t_interp.shape), str(input_images.device))
        if self._graph is None or self._graph[0] != key:
            self._capture(key, input_images, target_images, t_interp, iteration)
        _, g, (sx, st, stt), static_losses, static_grads = self._graph
        sx.copy_(input_images)
        st.copy_(target_images)
        stt.copy_(t_interp)
        for p, grad in static_grads:         # an eager step in between (zero_grad) drops the references: the graph writes these tensors
            p.grad = grad
        g.replay()
        return static_losses

    def _capture(self, key, input_images, target_images, t_interp, iteration):
        self._graph = None
        static = (input_images.clone(), target_images.clone(), t_interp.clone())
        side = torch.cuda.Stream(device=input_images.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):           # warm-up off the default stream: plans, packed filters, scales, lazily built buffers
            for _ in range(2):
                self.optimizer.zero_grad()
                self._forward_backward(*static, iteration)
        torch.cuda.current_stream().wait_stream(side)
        self.optimizer.zero_grad(set_to_none=True)      # the captured backward then ASSIGNS .grad: replays overwrite, never accumulate
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            static_losses = self._forward_backward(*static, iteration)
        static_grads = [(p, p.grad) for p in self.model.parameters() if p.requires_grad and p.grad is not None]
        self._graph = (key, g, static, static_losses, static_grads)
        log.info("training step captured into a HIP graph (%d parameter gradients, input %s)", len(static_grads), key[0])

    def train(self, batches, n_epochs=1, on_step=None):
        it = 0
        for epoch in range(n_epochs):
            for inp, tgt, t in batches:
                it += 1
                losses = self.train_step(inp, tgt, t, it)
                if on_step:
                    on_step(epoch, it, losses)
            self.lr_scheduler.step()
        return it

    def save_model(self, path, epoch):
        """Checkpoint in the reference's layout (scripts/main.py:218-245); loadable by models.unetflow.get_model."""
        os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
        torch.save({"epoch": epoch, "stage1_state_dict": self.model.stage1_model.state_dict(),
                    "stage2_state_dict": self.model.stage2_model.state_dict(), "self.optimizer": self.optimizer.state_dict(),
                    "scheduler": self.lr_scheduler.state_dict()}, path)
