"""Factory + checkpoint loader; interface of the reference's scripts/models/unetflow.py:11-32."""
import logging

import torch

from .flow_computation import FlowComputationModel
from .flow_interpolation import FlowInterpolationModel

log = logging.getLogger(__name__)


def get_model(path, in_channels, out_channels, cross_skip, verbose=False, stage=1, cfg=None):
    assert stage in [1, 2], "Unsupported stage id."
    cls = FlowComputationModel if stage == 1 else FlowInterpolationModel
    model = cls(in_channels, out_channels, cross_skip, verbose=verbose, cfg=cfg)
    if path is None:
        log.info("Not loading weights for stage %s.", stage)
        return model
    data = torch.load(path, map_location="cpu")
    key = "stage%s_state_dict" % stage          # checkpoint layout written by scripts/main.py:218-245
    if key in data.keys():
        data = data[key]
        log.info("Loading weights for Stage %s UNet.", stage)
    model.load_state_dict(data)                 # strict, same keys/shapes as the reference
    return model
