"""Shared body of the two stage U-Nets (the reference spells it out twice:
scripts/models/flow_computation.py:27-289 and flow_interpolation.py:27-281).

Module/parameter names are the weight ABI.  Two execution
paths exist on purpose and are tested against each other on the GPU:
  * `forward` runs a pre-planned launch sequence (ssm_amd.engine.UNetPlan): all
    activations stay in the padded-plane layout, pooling is fused into the
    producing convolution, concat+upsample is one kernel;
  * `encoder` / `bottleneck` / `decoder` (the reference's public methods) run
    op by op through models.layers on plain NCHW tensors.
"""
import logging

import torch
import torch.nn as nn

from ssm_amd import hipbind as hb
from ssm_amd.engine import UNetPlan

from .CLSTM.convgru import ConvBGRU
from .CLSTM.convlstm import ConvBLSTM
from .layers import HipConv2d, avg_pool, conv, upsample2x_cat

log = logging.getLogger(__name__)


class StageUNet(nn.Module):
    STAGE = 0
    precision = None        # plan arithmetic of a stage model called on its own: "f32" (default) | "f32w" | "f16x3" | "f16f8"

    def __init__(self, in_channels, out_channels, cross_skip, verbose=False, cfg=None):
        super().__init__()
        self.cross_skip_connect = cross_skip
        self.cfg = cfg
        self.verbose = verbose
        self.in_channels, self.out_channels = in_channels, out_channels
        self.bottleneck_type = self.cfg.get("STAGE%d" % self.STAGE, "BOTTLENECK")
        log.info("Stage %d model.", self.STAGE)
        log.info("Encoder: UNET.  Bottleneck: %s.", self.bottleneck_type)
        self.build_model(in_channels, out_channels)
        self._plans = {}

    # ---- construction (names = state-dict keys) -------------------------------------
    def conv7a_in_planes(self):
        return 512

    def build_model(self, in_channels, out_channels):
        self.conv1a = conv(in_channels, 32, kernel_size=7, padding=3)
        self.conv1b = conv(32, 32, kernel_size=7, padding=3)
        self.pool2 = avg_pool(2, None, 0)
        self.conv2a = conv(32, 64, kernel_size=5, padding=2)
        self.conv2b = conv(64, 64, kernel_size=5, padding=2)
        self.pool3 = avg_pool(2, None, 0)
        self.conv3a = conv(64, 128, kernel_size=3)
        self.conv3b = conv(128, 128, kernel_size=3)
        self.pool4 = avg_pool(2, None, 0)
        self.conv4a = conv(128, 256, kernel_size=3)
        self.conv4b = conv(256, 256, kernel_size=3)
        self.pool5 = avg_pool(2, None, 0)
        self.conv5a = conv(256, 512, kernel_size=3)
        self.conv5b = conv(512, 512, kernel_size=3)
        self.pool6 = avg_pool(2, None, 0)
        if self.bottleneck_type == "CONV":
            self.conv6 = nn.Sequential(conv(512, 512, kernel_size=3), conv(512, 512, kernel_size=3))
        elif self.bottleneck_type in ("CLSTM", "CGRU"):
            cls = ConvBLSTM if self.bottleneck_type == "CLSTM" else ConvBGRU
            self.conv6 = cls(in_channels=512, hidden_channels=512, kernel_size=(3, 3), num_layers=2, batch_first=True)
        else:
            raise Exception("Unknown bottleneck type: %s" % self.bottleneck_type)
        self.upsample7 = self.upsample8 = self.upsample9 = self.upsample10 = self.upsample11 = upsample2x_cat
        self.conv7a = conv(self.conv7a_in_planes(), 512, kernel_size=3)
        self.conv7b = conv(512, 512, kernel_size=3)
        self.conv8a = conv(1024, 256, kernel_size=3)
        self.conv8b = conv(256, 256, kernel_size=3)
        self.conv9a = conv(512, 128, kernel_size=3)
        self.conv9b = conv(128, 128, kernel_size=3)
        self.conv10a = conv(256, 64, kernel_size=3)
        self.conv10b = conv(64, 64, kernel_size=3)
        self.conv11a = conv(128, 32, kernel_size=3)
        self.conv11b = conv(32, 32, kernel_size=3)
        self.fuse_conv = conv(64, 32, kernel_size=3)
        self.final_conv = HipConv2d(32, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, bias=True)

    # ---- op-by-op path (reference's public methods) ----------------------------------
    def encoder(self, img_tensor):
        if self.verbose:
            log.info("Input: %s", str(img_tensor.shape))
        c1 = self.conv1b(self.conv1a(img_tensor))
        c2 = self.conv2b(self.conv2a(self.pool2(c1)))
        c3 = self.conv3b(self.conv3a(self.pool3(c2)))
        c4 = self.conv4b(self.conv4a(self.pool4(c3)))
        c5 = self.conv5b(self.conv5a(self.pool5(c4)))
        return c1, c2, c3, c4, c5, self.pool6(c5)

    def bottleneck(self, tensor_list):
        if self.bottleneck_type in ("CLSTM", "CGRU"):
            x_fwd = torch.stack(tensor_list, dim=1)
            x_rev = torch.stack(tensor_list[::-1], dim=1)
            return self.conv6(x_fwd, x_rev)
        return torch.stack([self.conv6(x) for x in tensor_list], dim=1)

    def _decode(self, conv6_out, encoder_outputs, cross=None):
        c1, c2, c3, c4, c5, _ = encoder_outputs
        x = self.conv7b(self.conv7a(self.upsample7(conv6_out, cross)))
        x = self.conv8b(self.conv8a(self.upsample8(x, c5)))
        x = self.conv9b(self.conv9a(self.upsample9(x, c4)))
        x = self.conv10b(self.conv10a(self.upsample10(x, c3)))
        x = self.conv11b(self.conv11a(self.upsample11(x, c2)))
        x = self.fuse_conv(torch.cat([x, c1], dim=1))
        return self.final_conv(x)

    # ---- planned path ------------------------------------------------------------------
    def _stamp(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def plan_for(self, B, H, W, device, seq_len=1):
        mode = self.precision or "f32"
        key = (B, H, W, str(device), seq_len, mode)
        stamp = self._stamp()
        hit = self._plans.get(key)
        if hit is None or hit[0] != stamp:
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            hit = (stamp, UNetPlan(self.STAGE, sd, B, H, W, device, self.cross_skip_connect,
                                   mode=mode, bottleneck=self.bottleneck_type, seq_len=seq_len))
            self._plans = {key: hit}          # one live plan per module: activations are large
        return hit[1]

    @staticmethod
    def _cross_planes(plan, B, H, W, device):
        """Buffer for the stage-1 bottleneck output in the layout the plan's conv7a reads (fp32 planes, or HL8 / Q8 in the split modes)."""
        if plan.hl8:
            return hb.HPlanes(B, 512, H // 32, W // 32, device, q8=plan.q8)
        return hb.Planes(B, 512, H // 32, W // 32, device)

    def _run_planned_windows(self, unet_in, cross_list=None):
        """unet_in [B,T,C,H,W] -> (list of T conv6 outputs, list of T final outputs), windows coupled by the
        recurrent bottleneck: all T windows go through one plan as a time-major batch."""
        hb.require_device(unet_in, "U-Net input")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("the recurrent bottleneck has no HIP backward; use torch.no_grad() or FREEZE=TRUE")
        B, T, C, H, W = unet_in.shape
        plan = self.plan_for(T * B, H, W, unet_in.device, seq_len=T)
        plan.t["in"].load(unet_in.transpose(0, 1).reshape(T * B, C, H, W))
        cross_planes = None
        if plan.cross:
            cross_planes = self._cross_planes(plan, T * B, H, W, unet_in.device).load(torch.cat(list(cross_list), dim=0))
        out = plan.run(cross_planes=cross_planes).to_nchw()
        c6 = plan.t["c6"].to_nchw()
        return list(c6.reshape(T, B, *c6.shape[1:]).unbind(0)), list(out.reshape(T, B, *out.shape[1:]).unbind(0))

    def _run_planned(self, x, cross=None):
        """x [B,C,H,W] -> (conv6_out [B,512,H/32,W/32], final [B,Cout,H,W]) as fresh NCHW tensors."""
        hb.require_device(x, "U-Net input")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("a stage U-Net called on its own carries no autograd graph: the hand-written backward runs "
                                      "over the whole window (FullModel.forward(inference_mode=False), ssm_amd.backward.PairGrad); "
                                      "use torch.no_grad() or FREEZE=TRUE here")
        B, C, H, W = x.shape
        plan = self.plan_for(B, H, W, x.device)
        plan.t["in"].load(x)
        cross_planes = None
        if plan.cross:
            cross_planes = self._cross_planes(plan, B, H, W, x.device).load(cross)
        out = plan.run(cross_planes=cross_planes)
        return plan.t["c6"].to_nchw(), out.to_nchw()
