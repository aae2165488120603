"""Shared host side of ConvBLSTM / ConvBGRU: parameters with the published key names, execution through
ssm_amd.engine.RecurrentBottleneck on fp32 padded planes (the op-by-op path; the planned path of
ssm_amd.engine.UNetPlan drives the same class in the plan's precision)."""
import torch
import torch.nn as nn

from ssm_amd import hipbind as hb
from ssm_amd.engine import RecurrentBottleneck
from ssm_amd.weights import RECURRENT_HIDDEN, RECURRENT_LAYERS


class BidirectionalBottleneck(nn.Module):
    KIND = None           # "CLSTM" | "CGRU"
    NET = None            # class of one direction's network

    def __init__(self, in_channels, hidden_channels, kernel_size, num_layers, bias=True, batch_first=False):
        super().__init__()
        if not (in_channels == 512 and hidden_channels == 2 * RECURRENT_HIDDEN and tuple(kernel_size) == (3, 3)
                and num_layers == RECURRENT_LAYERS and bias and batch_first):
            raise NotImplementedError("the HIP recurrent bottleneck covers what the reference instantiates: "
                                      "in_channels=512, hidden_channels=512, kernel_size=(3,3), num_layers=2, "
                                      "batch_first=True (flow_computation.py:73-88)")
        self.forward_net = self.NET(in_channels, hidden_channels // 2, kernel_size, num_layers, bias=bias, batch_first=batch_first)
        self.reverse_net = self.NET(in_channels, hidden_channels // 2, kernel_size, num_layers, bias=bias, batch_first=batch_first)
        self.__dict__["_ssm_plan"] = None

    def _plan(self, B, T, h, w, device):
        stamp = (B, T, h, w, str(device)) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        hit = self.__dict__.get("_ssm_plan")
        if hit is None or hit[0] != stamp:
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            rb = RecurrentBottleneck(self.KIND, sd, B, T, h, w, device, "f32", prefix="")
            hit = (stamp, rb, hb.Planes(T * B, 512, h, w, device), hb.Planes(T * B, 512, h, w, device),
                   hb.Planes(T * B, 512, h, w, device))
            self.__dict__["_ssm_plan"] = hit
        return hit[1:]

    def forward(self, xforward, xreverse):
        """xforward [B,T,512,h,w]; xreverse = the sequence the reverse net consumes (the reference passes the
        time-reversed stack).  Returns [B,T,512,h,w]: forward-net outputs | reverse-net outputs flipped back in time."""
        hb.require_device(xforward, "bottleneck input")
        hb.require_device(xreverse, "bottleneck input (reverse)")
        assert xforward.dim() == 5 and xforward.shape == xreverse.shape, "expected two [B,T,C,h,w] tensors"
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("the recurrent bottleneck has no HIP backward; run under torch.no_grad()")
        B, T, C, h, w = xforward.shape
        rb, pf, pr, out = self._plan(B, T, h, w, xforward.device)
        pf.load(xforward.transpose(0, 1).reshape(T * B, C, h, w))
        pr.load(xreverse.flip(1).transpose(0, 1).reshape(T * B, C, h, w))      # back to slot (window) order
        rb.run(pf, out, x_rev=pr)
        return out.to_nchw().reshape(T, B, 512, h, w).transpose(0, 1).contiguous()
