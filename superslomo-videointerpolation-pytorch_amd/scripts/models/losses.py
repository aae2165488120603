"""Training losses; interface of the reference's scripts/models/losses.py:44-249.

Scope note: the training step is the NEXT row of the scope table
(SURVEY 8f-1).  The forward values of the reconstruction and warp terms are
built here (warps on the HIP kernel); there is no backward yet, and the
perceptual term needs torchvision's pretrained VGG16 (losses.py:23), which is
not available offline - it is reported as 0 and flagged `perceptual_available`.
"""
import logging

import torch
import torch.nn as nn

from .layers import warp

log = logging.getLogger(__name__)


class SSMLosses(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.loss_weights = self.read_loss_weights(cfg)
        self.perceptual_available = False

    def read_loss_weights(self, cfg):
        lambda_r = cfg.getfloat("TRAIN", "LAMBDA_R")
        lambda_w = cfg.getfloat("TRAIN", "LAMBDA_W")
        lambda_p = cfg.getfloat("TRAIN", "LAMBDA_P")
        return lambda_r, lambda_p, lambda_w

    @staticmethod
    def _l1_mean(a, b):
        return (a - b).abs().reshape(a.shape[0], -1).mean(dim=1)

    def get_warp_loss(self, img_tensor, flowC_output, flowI_input, flowI_output, target_image):
        """Per-sample sums of the L1 warp terms (losses.py:113-170), gated by the FREEZE flags."""
        img_0, img_1 = img_tensor[:, 0:3], img_tensor[:, 3:6]
        s1 = s2 = torch.zeros(img_tensor.shape[0], device=img_tensor.device)
        if not self.cfg.getboolean("STAGE1", "FREEZE"):
            s1 = self._l1_mean(warp(img_1, flowC_output[:, 0:2]), img_0) + self._l1_mean(
                warp(img_0, flowC_output[:, 2:4]), img_1)
        if not self.cfg.getboolean("STAGE2", "FREEZE"):
            ft1 = flowI_input[:, 6:8] + flowI_output[:, 1:3]
            ft0 = flowI_input[:, 8:10] + flowI_output[:, 3:5]
            s2 = self._l1_mean(warp(img_0, ft0), target_image) + self._l1_mean(warp(img_1, ft1), target_image)
        return s1 + s2, s1, s2

    def forward(self, flowC_input, flowC_output, flowI_input, flowI_output, interpolated_image, target_image):
        """-> [B,4] = (total, lambda_r*L1 recon, lambda_w*warp, lambda_p*perceptual[=0 here])."""
        lambda_r, lambda_p, lambda_w = self.loss_weights
        with torch.no_grad():
            rec = lambda_r * self._l1_mean(interpolated_image, target_image)
            wrp = lambda_w * self.get_warp_loss(flowC_input, flowC_output, flowI_input, flowI_output, target_image)[0]
            per = torch.zeros_like(rec)
        return torch.stack([rec + wrp + per, rec, wrp, per], dim=1)
