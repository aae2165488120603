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
    def __init__(self, model, cfg):
        self.model, self.cfg = model, cfg
        self.learning_rate = cfg.getfloat("TRAIN", "LEARNING_RATE")
        self.lr_period = cfg.getint("TRAIN", "LR_PERIOD")
        self.lr_decay = cfg.getfloat("TRAIN", "LR_DECAY")
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("nothing to train: both stages have FREEZE=TRUE (as the reference's ini files ship); set "
                             "STAGE1.FREEZE / STAGE2.FREEZE to FALSE")
        self.optimizer = torch.optim.Adam(params, lr=self.learning_rate)
        self.lr_scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=self.lr_period, gamma=self.lr_decay)
        self.allreduce = GradientAllReduce(params)
        self.last_allreduce_s = 0.0

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
        _, losses = self.model(input_images, t_interp, target_images=target_images, iteration=iteration, inference_mode=False)
        losses = losses.mean(dim=0)
        self.optimizer.zero_grad()
        losses[0].backward()
        self.last_allreduce_s = self.allreduce()
        self.optimizer.step()
        return losses.detach()

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
