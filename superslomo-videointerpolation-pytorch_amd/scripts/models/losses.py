"""Training losses; interface of the reference's scripts/models/losses.py:44-249.

Forward values of all four entries of the reference's [B,4] loss tensor:
  [total, lambda_r * L1(I_t^, I_t), lambda_w * (sum of the L1 warp terms), lambda_p * MSE(phi(I_t^), phi(I_t))]
with the warps on the HIP sampler; the gradients are ssm_amd.backward (fused into the synthesis / input adjoints).
phi = torchvision.models.vgg16(pretrained=True).features[:23] in the reference (losses.py:23,34); its pretrained
weights need the network, so here phi runs on the HIP kernels (ssm_amd.perceptual.VGGFeatures) from a
torchvision-format state dict given with `load_vgg16()` or the file named by $SSM_VGG16_WEIGHTS.  Without weights
(`perceptual_available` False) the term is reported as 0 and contributes no gradient.
"""
import os
import logging

import torch
import torch.nn as nn

from .layers import warp

log = logging.getLogger(__name__)


class _PerceptualFn(torch.autograd.Function):
    """Per-sample VGG16 conv4_3 feature MSE on the HIP extractor with its gradient wrt the predicted frame."""

    @staticmethod
    def forward(ctx, pred, target, term):
        ctx.term = term
        return term.forward(pred.detach(), target.detach())

    @staticmethod
    def backward(ctx, g):           # the VGG activations of the forward are still in the plan's buffers
        B = g.shape[0]
        return ctx.term.grad_pred(g.contiguous()).to_nchw()[:B, :3], None, None


class SSMLosses(nn.Module):
    def __init__(self, cfg, feature_extractor=None):
        super().__init__()
        self.cfg = cfg
        self.loss_weights = self.read_loss_weights(cfg)
        self.feature_extractor = feature_extractor        # optional callable [B,3,H,W] -> features (forward value only)
        self.__dict__["_vgg_sd"] = None
        self.__dict__["_pterm"] = None
        path = os.environ.get("SSM_VGG16_WEIGHTS")
        if path:
            self.load_vgg16(torch.load(path, map_location="cpu"))

    def load_vgg16(self, state_dict):
        """torchvision vgg16 state dict (keys `features.<idx>.weight|bias`; the classifier entries are ignored)."""
        self.__dict__["_vgg_sd"] = {k: v for k, v in state_dict.items() if k.startswith("features.")}
        self.__dict__["_pterm"] = None

    @property
    def perceptual_available(self):
        return self.feature_extractor is not None or self._vgg_sd is not None

    def perceptual_term(self, B, H, W, device):
        """The HIP VGG16 plan for this batch geometry (None without weights)."""
        if self._vgg_sd is None:
            return None
        # the model's training precision (FullModel._train_engine stores it here), else $SSM_TRAIN_PRECISION, else exact fp32
        mode = self.__dict__.get("train_precision") or os.environ.get("SSM_TRAIN_PRECISION", "f32")
        # (f32w: the extractor's 3x3 layers with 8+ input channels run as Winograd F(2x2,3x3) like the U-Nets' - fp32 throughout)
        key = (B, H, W, str(device), mode)
        if self._pterm is None or self._pterm[0] != key:
            from ssm_amd.perceptual import PerceptualTerm
            self.__dict__["_pterm"] = None
            self.__dict__["_pterm"] = (key, PerceptualTerm(self._vgg_sd, B, H, W, device, mode))
        return self._pterm[1]

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
        """Differentiable when its inputs carry an autograd graph (op-by-op path: every operator has its own backward);
        the planned training step calls it on detached tensors and supplies the gradients itself (ssm_amd.backward)."""
        lambda_r, lambda_p, lambda_w = self.loss_weights
        rec = lambda_r * self._mean((interpolated_image - target_image).abs())
        wrp = lambda_w * self.warp_terms(img_tensor, flowC_output, est_flow_t1, est_flow_t0, flowI_output, target_image)
        Bn, _, Hn, Wn = interpolated_image.shape
        pt = self.perceptual_term(Bn, Hn, Wn, interpolated_image.device)
        if pt is not None:
            if torch.is_grad_enabled() and interpolated_image.requires_grad:
                per = lambda_p * _PerceptualFn.apply(interpolated_image, target_image, pt)
            else:
                per = lambda_p * pt.forward(interpolated_image, target_image)
        elif self.feature_extractor is not None:
            per = lambda_p * self._mean((self.feature_extractor(interpolated_image) - self.feature_extractor(target_image)) ** 2)
        else:
            per = torch.zeros_like(rec)
        return torch.stack([rec + wrp + per, rec, wrp, per], dim=1)          # [B,4] (losses.py:236-249)

    def planned_losses(self, eng, interpolated_image, target_image):
        """The same [B,4] tensor for the planned training step (ssm_amd.engine.PairEngine `eng` holds the window's tensors): the L1 terms by
        ssm_train_loss_sums - two launches instead of ~25 small strided torch kernels between the forward and the backward - and the
        perceptual term by its own kernels.  Values only: the step supplies the gradients itself (ssm_amd.backward)."""
        from ssm_amd import hipbind as hb
        lambda_r, lambda_p, lambda_w = self.loss_weights
        Bn, _, Hn, Wn = interpolated_image.shape
        dev = interpolated_image.device
        key = (Bn, str(dev))
        if self.__dict__.get("_l1_buf", (None,))[0] != key:
            self.__dict__["_l1_buf"] = (key, torch.empty(128 * Bn, dtype=torch.float32, device=dev), torch.empty(Bn, 2, dtype=torch.float32, device=dev),
                                        torch.empty(Bn, 4, dtype=torch.float32, device=dev))
        sums = self._l1_buf[2]          # (buffers of the module: the launch below may be part of a recorded program, ssm_amd.hipbind.LaunchProgram)
        # the kernel indexes the stage-1 flows and the pair by the stage-2 batch entry: one interpolation time per sample (training plan)
        assert eng.G == 1 and eng.B1 == Bn, "planned_losses needs one interpolation time per pair (stage-2 batch %d, pairs %d)" % (eng.B2, eng.B1)
        in16, out5, flow4 = eng.s2.t["in"], eng.s2.t["out"], eng.s1.t["out"]
        est = hb.view_of(eng.est) if eng.hl8 else in16.view(6)
        pred, tgt = interpolated_image.contiguous(), target_image.contiguous()
        hb.check(hb.load().ssm_train_loss_sums(hb.view_of(eng.img6), flow4.view(), est, out5.view(), hb.view_of(pred), hb.view_of(tgt),
                                               self._l1_buf[1].data_ptr(), sums.data_ptr(), Bn, Hn, Wn,
                                               0 if self.cfg.getboolean("STAGE1", "FREEZE") else 1,
                                               0 if self.cfg.getboolean("STAGE2", "FREEZE") else 1, hb.stream_ptr()))
        n = 3.0 * Hn * Wn
        pt = self.perceptual_term(Bn, Hn, Wn, dev)
        per_raw = pt.forward(pred, tgt) if pt is not None else None
        out = self._l1_buf[3] if hb._recorder is not None else torch.empty(Bn, 4, dtype=torch.float32, device=dev)

        def assemble():
            rec, wrp = sums[:, 0] * (lambda_r / n), sums[:, 1] * (lambda_w / n)
            per = lambda_p * per_raw if per_raw is not None else torch.zeros_like(rec)
            out.copy_(torch.stack([rec + wrp + per, rec, wrp, per], dim=1))
        hb.host_op(assemble)
        return out

    def forward(self, flowC_input, flowC_output, flowI_input, flowI_output, interpolated_image, target_image):
        """Reference signature (losses.py:196-249): flowI_input is the 16-channel stage-2 input, of which channels
        6:8 / 8:10 (the approximated flows) are used."""
        return self.losses_from_parts(flowC_input, flowC_output, flowI_input[:, 6:8], flowI_input[:, 8:10], flowI_output,
                                      interpolated_image, target_image)
