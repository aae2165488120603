"""Training losses; interface of the reference's scripts/models/losses.py:44-249.

Forward values of all four entries of the reference's [B,4] loss tensor:
  [total, lambda_r * L1(I_t^, I_t), lambda_w * (sum of the L1 warp terms), lambda_p * MSE(phi(I_t^), phi(I_t))]
with the warps on the HIP sampler.  Scope: there is NO backward yet - the training step is the next
row - and the reference's phi = torchvision.models.vgg16(pretrained=True).features[:23] (losses.py:23,34)
needs weights from the network, so the perceptual term is computed only when a `feature_extractor` is supplied
(`perceptual_available`), otherwise reported as 0.
"""
import logging

import torch
import torch.nn as nn

from .layers import warp

log = logging.getLogger(__name__)


class SSMLosses(nn.Module):
    def __init__(self, cfg, feature_extractor=None):
        super().__init__()
        self.cfg = cfg
        self.loss_weights = self.read_loss_weights(cfg)
        self.feature_extractor = feature_extractor        # callable [B,3,H,W] -> features; None: term reported as 0

    @property
    def perceptual_available(self):
        return self.feature_extractor is not None

    def read_loss_weights(self, cfg):
        lambda_r = cfg.getfloat("TRAIN", "LAMBDA_R")
        lambda_w = cfg.getfloat("TRAIN", "LAMBDA_W")
        lambda_p = cfg.getfloat("TRAIN", "LAMBDA_P")
        return lambda_r, lambda_p, lambda_w

    @staticmethod
    def _mean(x):
        """Per-sample mean over everything but the batch axis (losses.py:225-233)."""
        return x.reshape(x.shape[0], -1).mean(dim=1)

    def warp_terms(self, img_tensor, flowC_output, est_flow_t1, est_flow_t0, flowI_output, target_image):
        """Per-sample warp loss (losses.py:113-170): |g(I1,F01) - I0| + |g(I0,F10) - I1| when stage 1 trains,
        |g(I0,Ft0) - I_t| + |g(I1,Ft1) - I_t| (refined flows) when stage 2 trains; the four L1 maps are summed
        before the mean."""
        img_0, img_1 = img_tensor[:, 0:3], img_tensor[:, 3:6]
        total = torch.zeros_like(target_image)
        if not self.cfg.getboolean("STAGE1", "FREEZE"):
            total = total + (warp(img_1, flowC_output[:, 0:2]) - img_0).abs() + (warp(img_0, flowC_output[:, 2:4]) - img_1).abs()
        if not self.cfg.getboolean("STAGE2", "FREEZE"):
            ft1 = est_flow_t1 + flowI_output[:, 1:3]
            ft0 = est_flow_t0 + flowI_output[:, 3:5]
            total = total + (warp(img_0, ft0) - target_image).abs() + (warp(img_1, ft1) - target_image).abs()
        return self._mean(total)

    def losses_from_parts(self, img_tensor, flowC_output, est_flow_t1, est_flow_t0, flowI_output, interpolated_image,
                          target_image):
        lambda_r, lambda_p, lambda_w = self.loss_weights
        with torch.no_grad():
            rec = lambda_r * self._mean((interpolated_image - target_image).abs())
            wrp = lambda_w * self.warp_terms(img_tensor, flowC_output, est_flow_t1, est_flow_t0, flowI_output, target_image)
            if self.feature_extractor is not None:
                per = lambda_p * self._mean((self.feature_extractor(interpolated_image) - self.feature_extractor(target_image)) ** 2)
            else:
                per = torch.zeros_like(rec)
        return torch.stack([rec + wrp + per, rec, wrp, per], dim=1)          # [B,4] (losses.py:236-249)

    def forward(self, flowC_input, flowC_output, flowI_input, flowI_output, interpolated_image, target_image):
        """Reference signature (losses.py:196-249): flowI_input is the 16-channel stage-2 input, of which channels
        6:8 / 8:10 (the approximated flows) are used."""
        return self.losses_from_parts(flowC_input, flowC_output, flowI_input[:, 6:8], flowI_input[:, 8:10], flowI_output,
                                      interpolated_image, target_image)
