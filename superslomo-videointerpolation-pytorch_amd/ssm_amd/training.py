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

    def __init__(self, model, cfg, graphs=None):
        self.model, self.cfg = model, cfg
        self.graphs = (os.environ.get("SSM_TRAIN_GRAPH", "0") != "0") if graphs is None else bool(graphs)
        self._graph = None
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
