"""FullModel: frame window -> intermediate frame, MI355X-native.

Interface of the reference's scripts/models/superslomo_r.py:33-293 (ctor from the
ini `cfg`, `forward(image_tensor, t_interp, target_images, iteration,
inference_mode)`, attributes `stage1_model`, `stage2_model`, `loss`), so the
reference's scripts (main.py:130-136, evaluate_interpolation_results.py:227,241,
visualize_interpolation.py:144) can call it unchanged.

Inference runs on ssm_amd.engine.PairEngine.  `interpolate()` is the hoisted form
of the evaluators' t-loop: stage 1 once per pair, all t values batched through
stage 2.
"""
import logging
import os

import torch
import torch.nn as nn

from ssm_amd import hipbind as hb
from ssm_amd.engine import PairEngine, WindowEngine

# Arithmetic of the convolutions on the planned path (ssm_amd.engine.MODES).  "f32w" (default) and "f32" are fp32 throughout
# (the reference's arithmetic: every product and sum in fp32 on v_mfma_f32_32x32x2_f32): "f32" evaluates every convolution in the
# direct form (an fmaf chain per output), "f32w" evaluates the 3x3 layers as Winograd F(2x2,3x3) - 2.25x fewer multiplies, the same
# result up to fp32 rounding (both sit 2e-4 from the CPU oracle at 736x1280; tests/emulate_winograd_precision.py).  Opt-in split
# modes, narrower than fp32: "f16x3" evaluates a product as three fp16 MFMAs on hi/lo-split operands (~22-bit operands), "f16f8"
# as one fp16 MFMA on the hi parts plus two block-scaled e4m3 MFMAs for the compensation terms (~15-bit products); both accumulate
# in fp32 and meet the 1e-3 frame bar in the tests at 2.5x / 2.9x the frame rate.  "f16" is plain fp16 inputs (config 5).
DEFAULT_PRECISION = "f32w"
DEFAULT_TRAIN_PRECISION = "f32"

from . import unetflow as unet
from .losses import SSMLosses

log = logging.getLogger(__name__)


def validate_target_tensor(model_forward_func):
    def func_wrapper(self, image_tensor, t_interp, target_images=None, iteration=None, inference_mode=True):
        if not inference_mode:
            assert target_images is not None, "No target found for loss."
            assert target_images.shape[1] == self.cfg.getint("TRAIN", "N_FRAMES") - 1, "Insufficient number of targets."
        return model_forward_func(self, image_tensor, t_interp, target_images, iteration, inference_mode)

    return func_wrapper


def _t_vector(t_values, device):
    """Interpolation times as a device vector; the (0,1) range check (validators.py:9-11) runs on the host copy when
    the values arrive as Python numbers / a CPU tensor, so the common case costs no device sync."""
    if isinstance(t_values, torch.Tensor) and t_values.is_cuda:
        t = t_values.to(torch.float32).reshape(-1)
        assert bool((t > 0).all() and (t < 1).all()), "Interpolation values out of bounds."
        return t
    host = torch.as_tensor(t_values, dtype=torch.float32).reshape(-1)
    assert bool((host > 0).all() and (host < 1).all()), "Interpolation values out of bounds."
    return host.to(device)


class _TrainStep(torch.autograd.Function):
    """forward = the HIP forward of one window + the [B,4] losses; backward = ssm_amd.backward.PairGrad.
    The parameters are inputs of the Function so autograd delivers their gradients to `.grad`."""

    @staticmethod
    def forward(ctx, model, img6, t, target, *params):
        B, _, H, W = img6.shape
        eng, pg = model._train_engine(B, H, W, img6.device)
        sd1 = {k: v.detach() for k, v in model.stage1_model.state_dict().items()}
        sd2 = {k: v.detach() for k, v in model.stage2_model.state_dict().items()}
        eng.s1.refresh_weights(sd1)
        eng.s2.refresh_weights(sd2)
        pt = model.loss.perceptual_term(B, H, W, img6.device)
        if pt is not None and os.environ.get("SSM_VGG_OVERLAP", "1") != "0":
            pt.begin_target(target)      # the target's VGG features do not depend on the forward: second stream, beside the U-Nets
        pred = eng.run(img6, t, want_aux=True, want_out5=True)
        if hb._recorder is not None:      # a recorded step (ssm_amd.training.Trainer programs): the frame lives in a buffer of the plan's life
            keep = model.__dict__.setdefault("_pred_static", {})
            stat = keep.get(tuple(pred.shape))
            if stat is None or stat.device != pred.device:
                stat = keep[tuple(pred.shape)] = torch.empty_like(pred)
            src = pred
            hb.host_op(lambda: stat.copy_(src))
            pred = stat
        else:
            pred = pred.clone()
        if model.loss.feature_extractor is None and os.environ.get("SSM_FUSED_LOSS", "1") != "0":
            losses = model.loss.planned_losses(eng, pred, target)          # the L1 terms in two launches (ssm_train_loss_sums)
        else:
            f01, f10, e1, e0, _, _, _ = eng.intermediates()
            losses = model.loss.losses_from_parts(img6, torch.cat([f01, f10], 1), e1, e0, eng.s2.t["out"].interior, pred, target)
        ctx.model, ctx.pg, ctx.sd, ctx.target = model, pg, (sd1, sd2), target
        ctx.mark_non_differentiable(pred)
        return pred, losses

    @staticmethod
    def backward(ctx, d_pred, d_losses):
        model = ctx.model
        lambda_r, lambda_p, lambda_w = model.loss.loss_weights
        train_s1 = not model.cfg.getboolean("STAGE1", "FREEZE")
        train_s2 = not model.cfg.getboolean("STAGE2", "FREEZE")
        d_losses = d_losses.contiguous()
        dy_extra = None
        B, _, H, W = ctx.target.shape
        pt = model.loss.perceptual_term(B, H, W, ctx.target.device)
        S = ctx.pg.loss_scale(max(lambda_r, lambda_w, lambda_p))      # power-of-two gradient scale of the f16f8 plan (1 for f32)
        if pt is not None:        # VGG activations of this step's forward are still in the plan's buffers
            dy_extra = pt.grad_pred((d_losses[:, 0] + d_losses[:, 3]) * (S * lambda_p)).view()
        named = [(stage + name, p) for stage, mod in (("stage1.", model.stage1_model), ("stage2.", model.stage2_model))
                 for name, p in mod.named_parameters()]
        for _, p in named:          # a gradient kept from an earlier backward lives in the buffers this backward rewrites: detach it
            if p.grad is not None and ctx.pg.owns(p.grad):
                p.grad = p.grad.clone()
        grads = ctx.pg.backward(ctx.sd[0], ctx.sd[1], ctx.target, d_losses, lambda_r, lambda_w, train_s1, train_s2,
                                dy_extra=dy_extra, lambda_p=lambda_p)
        # The gradients are handed to `.grad` directly (what AccumulateGrad would do, minus 96 clone launches per step): slices of the
        # per-U-Net flat buffers, valid until this model's next backward; an existing .grad is accumulated into, as autograd does.
        for key, p in named:
            g = grads.get(key)
            if g is None or not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
        return (None, None, None, None) + (None,) * len(named)


class FullModel(nn.Module):
    def __init__(self, cfg, writer=None):
        super().__init__()
        self.cfg = cfg
        self.writer = writer
        self.load_weights()
        self.freeze_weights()
        self.loss = SSMLosses(cfg)
        self._engine = None
        # a load_state_dict into the stage modules (checkpoint resume, main.py:263-284) writes the parameters IN PLACE: same
        # addresses, so the training plan would keep the power-of-two filter pre-scales it chose for the old values - drop the
        # plans, the next step rebuilds them from the loaded weights
        for m in (self.stage1_model, self.stage2_model):
            m.register_load_state_dict_post_hook(lambda module, incompatible_keys: self._drop_plans())

    def _drop_plans(self):
        self._engine = None
        self._pipe = None
        self._train = None

    def load_weights(self):
        # quirk kept for drop-in behaviour: BOTH paths are gated on STAGE1.LOADPREV (superslomo_r.py:46-52)
        load = self.cfg.getboolean("STAGE1", "LOADPREV")
        stage1_weights = self.cfg.get("STAGE1", "WEIGHTS") if load else None
        stage2_weights = self.cfg.get("STAGE2", "WEIGHTS") if load else None
        self.cross_skip = self.cfg.getboolean("STAGE2", "CROSS_SKIP")
        if self.cfg.get("STAGE1", "ENCODER") != "UNET":
            raise NotImplementedError
        log.info("STAGE 1 UNET")
        self.stage1_model = unet.get_model(stage1_weights, 6, 4, self.cross_skip, stage=1, cfg=self.cfg)
        log.info("STAGE 2 %s", self.cfg.get("STAGE2", "ENCODER"))
        self.stage2_model = unet.get_model(stage2_weights, 16, 5, self.cross_skip, stage=2, cfg=self.cfg)
        log.info("Cross stage Skip Connections Present? %s ", self.cross_skip)
        self.bottlenecks = (self.cfg.get("STAGE1", "BOTTLENECK"), self.cfg.get("STAGE2", "BOTTLENECK"))
        self.recurrent = any(b != "CONV" for b in self.bottlenecks)      # windows coupled through conv6 (config 4)

    def freeze_weights(self):
        for name, model in (("STAGE1", self.stage1_model), ("STAGE2", self.stage2_model)):
            if self.cfg.getboolean(name, "FREEZE"):
                log.info("Freezing %s model.", name.lower())
                model.eval()
                for param in model.parameters():
                    param.requires_grad = False
            else:
                log.info("Training %s model.", name.lower())

    def get_image_pairs(self, img_tensor):
        """[B,N,3,H,W] -> [B,N-1,6,H,W]: adjacent frames paired on the channel axis."""
        return torch.cat([img_tensor[:, :-1], img_tensor[:, 1:]], dim=2)

    # ---- engine plumbing ---------------------------------------------------------------
    def _stamp(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    precision = None    # "f32w" | "f32" | "f16x3" | "f16f8" | "f16"; None -> $SSM_PRECISION or the default above

    def engine_for(self, B1, B2, H, W, device):
        mode = self.precision or os.environ.get("SSM_PRECISION", DEFAULT_PRECISION)
        key = (B1, B2, H, W, str(device), mode, self._stamp())
        if self._engine is None or self._engine[0] != key:
            sd1 = {k: v.detach() for k, v in self.stage1_model.state_dict().items()}
            sd2 = {k: v.detach() for k, v in self.stage2_model.state_dict().items()}
            self._engine = None      # free the old plan's activations before allocating the new one
            self._engine = (key, PairEngine(sd1, sd2, B1, B2, H, W, device, self.cross_skip, mode))
        return self._engine[1]

    def window_engine_for(self, T, S1, S2, H, W, device, decode_all=False):
        mode = self.precision or os.environ.get("SSM_PRECISION", DEFAULT_PRECISION)
        key = ("win", T, S1, S2, H, W, str(device), mode, decode_all, self._stamp())
        if self._engine is None or self._engine[0] != key:
            sd1 = {k: v.detach() for k, v in self.stage1_model.state_dict().items()}
            sd2 = {k: v.detach() for k, v in self.stage2_model.state_dict().items()}
            self._engine = None
            self._engine = (key, WindowEngine(sd1, sd2, T, S1, S2, H, W, device, self.cross_skip, mode, self.bottlenecks,
                                              decode_all))
        return self._engine[1]

    @torch.no_grad()
    def interpolate_windows(self, frames, t_values):
        """One clip [1,N,3,H,W] of N = N_FRAMES frames -> [len(t_values),3,H,W]: the frames between the two middle
        inputs at every t (the per-t loop of evaluate_interpolation_results.py:234-242 for the recurrent
        configuration), stage 1 once per clip, the t values batched through stage 2."""
        hb.require_device(frames, "frame window")
        assert frames.dim() == 5 and frames.shape[0] == 1, "interpolate_windows() takes one clip [1,N,3,H,W]"
        T = frames.shape[1] - 1
        t = _t_vector(t_values, frames.device)
        eng = self.window_engine_for(T, 1, t.numel(), frames.shape[3], frames.shape[4], frames.device)
        return eng.run(frames.contiguous(), t[:, None].expand(-1, T), want_aux=False).clone()

    @torch.no_grad()
    def interpolate(self, image_pair, t_values):
        """One pair [1,2,3,H,W] (or [1,6,H,W]) -> [len(t_values),3,H,W]: stage 1 once, every t
        batched through stage 2 (the loop of evaluate_interpolation_results.py:213-244, hoisted)."""
        hb.require_device(image_pair, "image pair")
        img6 = image_pair.reshape(image_pair.shape[0], 6, *image_pair.shape[-2:])
        assert img6.shape[0] == 1, "interpolate() takes one frame pair"
        t = _t_vector(t_values, img6.device)
        eng = self.engine_for(1, t.numel(), img6.shape[2], img6.shape[3], img6.device)
        return eng.run(img6, t, want_aux=False).clone()

    @torch.no_grad()
    def interpolate_many(self, pairs, t_values, n_streams=2, pairs_per_batch=1):
        """Throughput form of interpolate(): a list of pairs ([1,2,3,H,W] each, same size) -> list of
        [len(t_values),3,H,W] tensors.  Passes of `pairs_per_batch` pairs are dealt round-robin to `n_streams` engines on
        separate HIP streams (ssm_amd.engine.PairPipeline) so under-filled and tail phases of one pass overlap the
        MFMA-bound convolutions of another; pairs_per_batch > 1 also gives every convolution launch that many times the
        workgroups (bench.py's configuration: 3 streams x 2 pairs).  A remainder that does not fill a pass goes through
        the one-pair engine."""
        from ssm_amd.engine import PairPipeline
        first = pairs[0]
        hb.require_device(first, "image pair")
        H, W = first.shape[-2:]
        t = _t_vector(t_values, first.device)
        P = max(1, int(pairs_per_batch))
        mode = self.precision or os.environ.get("SSM_PRECISION", DEFAULT_PRECISION)
        key = ("pipe", n_streams, P, t.numel(), H, W, str(first.device), mode, self._stamp())
        if getattr(self, "_pipe", None) is None or self._pipe[0] != key:
            sd1 = {k: v.detach() for k, v in self.stage1_model.state_dict().items()}
            sd2 = {k: v.detach() for k, v in self.stage2_model.state_dict().items()}
            self._pipe = None
            self._pipe = (key, PairPipeline(sd1, sd2, t.numel(), H, W, first.device, self.cross_skip, mode, n_streams,
                                            pairs_per_batch=P))
        pipe = self._pipe[1]
        full = len(pairs) // P * P
        outs = []
        for i in range(0, full, P):
            batch = pairs[i].reshape(1, 6, H, W) if P == 1 else torch.cat([pr.reshape(1, 6, H, W) for pr in pairs[i:i + P]], 0)
            outs.append(pipe.submit(batch, t, clone=True))
        pipe.sync()
        res = [o for out in outs for o in (out.split(t.numel()) if P > 1 else (out,))]
        for pr in pairs[full:]:
            res.append(self.interpolate(pr, t_values).clone())
        return res

    # ---- training step ------------------------------------------------------------------------------------------
    grad_sync = None            # a ssm_amd.dist.GradientAllReduce (set by the Trainer), or None
    train_precision = None      # "f32" (default; $SSM_TRAIN_PRECISION) | "f32w" (opt-in: fp32 throughout, forward + data-gradient 3x3 convolutions as Winograd F(2x2,3x3)) | "f16f8" (opt-in: split forward / data gradients, bf16x3 weight gradients)

    def _train_engine(self, B, H, W, device):
        """The training plan + its PairGrad.  f16f8: the inference plan (fused upsample, fp16 + fp8 matrix path) writing fp32
        twins of every conv output for the backward, data gradients on the same kernels; f32: exact-fp32 MFMA plan with
        materialised upsample tensors."""
        from ssm_amd.backward import PairGrad
        mode = self.train_precision or os.environ.get("SSM_TRAIN_PRECISION", DEFAULT_TRAIN_PRECISION)
        assert mode in ("f16f8", "f32", "f32w"), "training precision must be f32, f32w or f16f8"
        self.loss.__dict__["train_precision"] = mode        # the VGG16 term runs in the same arithmetic
        key = (B, H, W, str(device), mode)
        if getattr(self, "_train", None) is None or self._train[0] != key:
            sd1 = {k: v.detach() for k, v in self.stage1_model.state_dict().items()}
            sd2 = {k: v.detach() for k, v in self.stage2_model.state_dict().items()}
            self._train = None
            if mode in ("f32", "f32w"):        # f32w: forward and data-gradient 3x3 convolutions in the Winograd form
                eng = PairEngine(sd1, sd2, B, B, H, W, device, self.cross_skip, mode, fuse_upsample=False)
            else:
                eng = PairEngine(sd1, sd2, B, B, H, W, device, self.cross_skip, "f16f8", fuse_upsample=True, twins=True)
            self._train = (key, eng, PairGrad(eng))
        if self.grad_sync is not None:        # ssm_amd.training.Trainer: bucketed gradient all-reduce overlapped with the backward
            self.grad_sync.attach(self._train[2])
        return self._train[1], self._train[2]

    def _forward_with_losses(self, image_tensor, t_interp, target_images):
        """inference_mode=False (superslomo_r.py:204-243): every window is interpolated, the losses of all windows
        are averaged, the middle window's frame is returned.  With trainable parameters and autograd enabled the
        returned loss tensor carries the hand-written backward (ssm_amd.backward), so the reference Trainer's
        `losses.mean(0)[0].backward(); optimizer.step()` (scripts/main.py:138-197) works unchanged."""
        image_pairs = self.get_image_pairs(image_tensor)
        B, T = image_pairs.shape[:2]
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if T != 1 or self.recurrent:      # several windows and/or recurrent cells: op-by-op autograd
                return self._forward_op_by_op(image_pairs, t_interp, target_images)
            params = list(self.stage1_model.parameters()) + list(self.stage2_model.parameters())
            return _TrainStep.apply(self, image_pairs[:, 0].contiguous(), t_interp[:, 0].reshape(B).to(torch.float32),
                                    target_images[:, 0].contiguous(), *params)
        losses = torch.zeros(B, 4, device=image_tensor.device)
        est_img_t = None
        if self.recurrent:
            with torch.no_grad():
                H, W = image_tensor.shape[-2:]
                eng = self.window_engine_for(T, B, B, H, W, image_tensor.device, decode_all=True)
                preds = eng.run(image_tensor.contiguous(), t_interp.reshape(B, T), want_aux=True)
                for k in range(T):
                    f01, f10, e1, e0, _, _, _ = eng.intermediates(k)
                    sl = slice(k * B, (k + 1) * B)
                    losses = losses + self.loss.losses_from_parts(eng.img6[sl], torch.cat([f01, f10], 1), e1, e0,
                                                                  eng.s2.t["out"].interior[sl], preds[sl], target_images[:, k])
                est_img_t = preds[(T // 2) * B:(T // 2 + 1) * B].clone()
            return est_img_t, losses / T
        with torch.no_grad():
            for k in range(T):
                img6 = image_pairs[:, k].contiguous()
                eng = self.engine_for(B, B, img6.shape[2], img6.shape[3], img6.device)
                pred = eng.run(img6, t_interp[:, k].reshape(B).to(torch.float32), want_aux=True, want_out5=True)
                f01, f10, e1, e0, _, _, _ = eng.intermediates()
                out5 = eng.s2.t["out"].interior
                losses = losses + self.loss.losses_from_parts(img6, torch.cat([f01, f10], 1), e1, e0, out5, pred,
                                                              target_images[:, k])
                if k == T // 2:
                    est_img_t = pred.clone()
        return est_img_t, losses / T

    def _forward_op_by_op(self, image_pairs, t_interp, target_images):
        """The reference's own loop (superslomo_r.py:152-243; stage forward flow_computation.py:291-325) on the public
        operators, each carrying its autograd: encoders of all windows, the bottleneck over the window sequence (CONV
        per window, or the recurrent cells), decoders, losses averaged over the windows.  Slower than the planned step but
        valid for any number of windows and for the recurrent bottleneck."""
        B, T = image_pairs.shape[:2]
        s1, s2 = self.stage1_model, self.stage2_model
        pairs = [image_pairs[:, k].contiguous() for k in range(T)]
        ts = [t_interp[:, k].reshape(B, 1, 1, 1).to(torch.float32) for k in range(T)]
        e1 = [s1.encoder(p) for p in pairs]
        h1 = s1.bottleneck([e[-1] for e in e1])
        dec1 = [s1.decoder(h1[:, k], e1[k]) for k in range(T)]
        in16 = [s2.compute_inputs(pairs[k], dec1[k][1], ts[k]) for k in range(T)]
        e2 = [s2.encoder(x) for x in in16]
        h2 = s2.bottleneck([e[-1] for e in e2])
        losses, est_img_t = 0.0, None
        for k in range(T):
            out5 = s2.decoder(h2[:, k], e2[k], dec1[k][0])
            pred = s2.compute_output_image(pairs[k], in16[k], out5, ts[k])
            losses = losses + self.loss(pairs[k], dec1[k][1], in16[k], out5, pred, target_images[:, k].contiguous())
            if k == T // 2:
                est_img_t = pred.detach()
        return est_img_t, losses / T

    @validate_target_tensor
    def forward(self, image_tensor, t_interp, target_images=None, iteration=None, inference_mode=True):
        """image_tensor [B,N,3,H,W] normalised frames, t_interp [B,N-1,1,1,1] in (0,1).
        Inference: (I_t [B,3,H,W], (F01, F10, Ft1^, Ft0^, Ft1, Ft0, V0)) of the middle window."""
        hb.require_device(image_tensor, "image tensor")
        if not inference_mode:
            return self._forward_with_losses(image_tensor, t_interp, target_images)
        image_pairs = self.get_image_pairs(image_tensor)
        B, T = image_pairs.shape[:2]
        mid_idx = T // 2
        if iteration == 1:
            log.info("%s interpolation windows. Mid_idx: %s", T, mid_idx)
        if self.recurrent:      # windows coupled through conv6: all T encoded, the middle one decoded
            with torch.no_grad():
                H, W = image_tensor.shape[-2:]
                eng = self.window_engine_for(T, B, B, H, W, image_tensor.device)
                est_img_t = eng.run(image_tensor.contiguous(), t_interp.reshape(B, T), want_aux=True).clone()
                outputs = tuple(x.clone() for x in eng.intermediates())
            return est_img_t, outputs
        # CONV bottleneck: windows are independent and only the middle one is returned
        # (superslomo_r.py:237-238), so only that window is computed.
        with torch.no_grad():
            img6 = image_pairs[:, mid_idx]
            t = t_interp[:, mid_idx].reshape(B).to(torch.float32)
            eng = self.engine_for(B, B, img6.shape[2], img6.shape[3], img6.device)
            est_img_t = eng.run(img6, t, want_aux=True).clone()
            outputs = tuple(x.clone() for x in eng.intermediates())
        return est_img_t, outputs
