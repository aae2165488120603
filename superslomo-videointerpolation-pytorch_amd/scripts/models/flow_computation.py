"""Stage 1: flow-computation U-Net (6 -> 4 channels: F0->1 | F1->0).
Interface of the reference's scripts/models/flow_computation.py:14-325."""
import logging

from .unet_common import StageUNet

log = logging.getLogger(__name__)


class FlowComputationModel(StageUNet):
    STAGE = 1

    def decoder(self, input_tensor, encoder_outputs):
        final_out = self._decode(input_tensor, encoder_outputs)
        return (input_tensor if self.cross_skip_connect else None), final_out

    def forward(self, unet_in):
        """unet_in [B,T,6,H,W] -> list of T tuples (conv6_out or None, flow [B,4,H,W])."""
        assert len(unet_in.shape) == 5, "Tensor not of shape: B T C H W"
        decodings = []
        if self.bottleneck_type != "CONV":
            encs, flows = self._run_planned_windows(unet_in)
            return [(e if self.cross_skip_connect else None, f) for e, f in zip(encs, flows)]
        for t in range(unet_in.shape[1]):
            enc, flow = self._run_planned(unet_in[:, t, ...])
            decodings.append((enc if self.cross_skip_connect else None, flow))
        return decodings
