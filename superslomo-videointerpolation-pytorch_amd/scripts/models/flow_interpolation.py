"""Stage 2: arbitrary-time flow-interpolation U-Net (16 -> 5 channels: visibility
logit | dFt->1 | dFt->0) plus the two fused kernels around it.
Interface of the reference's scripts/models/flow_interpolation.py:14-429."""
import logging

import torch

from ssm_amd import hipbind as hb
from ssm_amd import ops as _ops  # noqa: F401  (torch.ops.ssm.*)

from .unet_common import StageUNet

log = logging.getLogger(__name__)


def _t_vector(t, B, device):
    """[B,1,1,1] (or scalar) interpolation times -> contiguous [B] device vector."""
    t = torch.as_tensor(t, dtype=torch.float32, device=device).reshape(-1)
    if t.numel() == 1 and B > 1:
        t = t.expand(B)
    assert t.numel() == B, "one interpolation time per sample expected"
    return t.contiguous()


class FlowInterpolationModel(StageUNet):
    STAGE = 2

    def conv7a_in_planes(self):
        return 1024 if self.cross_skip_connect else 512

    def decoder(self, input_tensor, encoder_outputs, stage1_encoder_output=None):
        cross = stage1_encoder_output if self.cross_skip_connect else None
        return self._decode(input_tensor, encoder_outputs, cross)

    def forward(self, unet_in, stage1_encoder_output=None):
        """unet_in [B,T,16,H,W], stage1_encoder_output list of T [B,512,H/32,W/32] -> list of T [B,5,H,W]."""
        assert len(unet_in.shape) == 5, "Tensor not of shape: B T C H W"
        decodings = []
        if self.bottleneck_type != "CONV":
            return self._run_planned_windows(unet_in, stage1_encoder_output if self.cross_skip_connect else None)[1]
        for t in range(unet_in.shape[1]):
            cross = stage1_encoder_output[t] if self.cross_skip_connect else None
            decodings.append(self._run_planned(unet_in[:, t, ...], cross)[1])
        return decodings

    def compute_inputs(self, img_tensor, flow_pred_tensor, t):
        """[B,6,H,W] images, [B,4,H,W] stage-1 flows, t in (0,1) -> [B,16,H,W]
        = cat[I1, g(I1,Ft1^), Ft1^, Ft0^, g(I0,Ft0^), I0] in ONE kernel."""
        hb.require_device(img_tensor, "image pair")
        hb.require_device(flow_pred_tensor, "flow tensor")
        B, _, H, W = img_tensor.shape
        img, flow = img_tensor.contiguous(), flow_pred_tensor.contiguous()
        tv = _t_vector(t, B, img.device)
        out = torch.ops.ssm.flowinterp_inputs(img, flow, tv)      # dispatcher op; its autograd = the HIP adjoint wrt the flows
        if self.verbose:
            log.info("Generated Input tensor of shape: %s", str(out.shape))
        return out

    def extract_outputs(self, output_tensor):
        v_1t = torch.sigmoid(output_tensor[:, 0:1, ...])
        return v_1t, output_tensor[:, 1:3, ...], output_tensor[:, 3:5, ...], 1 - v_1t

    def compute_output_image(self, img_tensor, input_tensor, output_tensor, t, return_aux=False):
        """Visibility-weighted blend of the two frames warped by the refined flows, ONE kernel."""
        for x, n in ((img_tensor, "image pair"), (input_tensor, "stage-2 input"), (output_tensor, "stage-2 output")):
            hb.require_device(x, n)
        B, _, H, W = img_tensor.shape
        img, xin, xout = img_tensor.contiguous(), input_tensor.contiguous(), output_tensor.contiguous()
        tv = _t_vector(t, B, img.device)
        if not return_aux:
            return torch.ops.ssm.synthesize(img, xin, xout, tv)
        y = torch.empty(B, 3, H, W, dtype=torch.float32, device=img.device)
        aux = torch.empty(B, 5, H, W, dtype=torch.float32, device=img.device) if return_aux else None
        hb.check(hb.load().ssm_synthesize_fwd(hb.view_of(img), hb.view_of(xin), hb.view_of(xout), tv.data_ptr(),
                                              hb.view_of(y), hb.view_of(aux) if return_aux else hb.NULL_VIEW, B, H, W,
                                              hb.stream_ptr()))
        return (y, aux) if return_aux else y
