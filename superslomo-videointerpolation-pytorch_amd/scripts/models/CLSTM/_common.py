"""Shared host side of ConvBLSTM / ConvBGRU: parameters with the published key names, execution through
ssm_amd.engine.RecurrentBottleneck on fp32 padded planes (the op-by-op path; the planned path of
ssm_amd.engine.UNetPlan drives the same class in the plan's precision)."""
import torch
import torch.nn as nn

from ssm_amd import hipbind as hb
from ssm_amd.engine import RecurrentBottleneck
from ssm_amd.weights import RECURRENT_HIDDEN, RECURRENT_LAYERS


def _z(*shape, device):
    return torch.empty(*shape, dtype=torch.float32, device=device)


class _LSTMCellFn(torch.autograd.Function):
    """(h', c') = cell(gates [B,4Hc,h,w], c [B,Hc,h,w] or None) on ssm_convlstm_cell_fwd / _bwd."""

    @staticmethod
    def forward(ctx, gates, c_prev):
        gates = gates.detach().contiguous()
        cp = None if c_prev is None else c_prev.detach().contiguous()
        B, G4, h, w = gates.shape
        Hc = G4 // 4
        hn, cn = _z(B, Hc, h, w, device=gates.device), _z(B, Hc, h, w, device=gates.device)
        hb.check(hb.load().ssm_convlstm_cell_fwd(hb.view_of(gates), hb.NULL_VIEW, hb.view_of(cp) if cp is not None else hb.NULL_VIEW,
                                                 hb.view_of(cn), hb.view_of(hn), hb.NULL_HVIEW, B, Hc, h, w, 0, hb.stream_ptr()))
        ctx.save_for_backward(gates, cp if cp is not None else gates.new_zeros(0))
        ctx.has_c = cp is not None
        return hn, cn

    @staticmethod
    def backward(ctx, dh, dc):
        gates, cp = ctx.saved_tensors
        B, G4, h, w = gates.shape
        Hc = G4 // 4
        dh = dh.contiguous() if dh is not None else torch.zeros(B, Hc, h, w, device=gates.device)
        dg, dcp = _z(B, G4, h, w, device=gates.device), _z(B, Hc, h, w, device=gates.device)
        hb.check(hb.load().ssm_convlstm_cell_bwd(hb.view_of(gates), hb.NULL_VIEW, hb.view_of(cp) if ctx.has_c else hb.NULL_VIEW,
                                                 hb.view_of(dh), hb.view_of(dc.contiguous()) if dc is not None else hb.NULL_VIEW,
                                                 hb.view_of(dg), hb.view_of(dcp), B, Hc, h, w, hb.stream_ptr()))
        return dg, (dcp if ctx.has_c else None)


class _GRUResetFn(torch.autograd.Function):
    """rh = sigmoid(gates[:, :Hc]) * h."""

    @staticmethod
    def forward(ctx, gates, hp):
        gates, hp = gates.detach().contiguous(), hp.detach().contiguous()
        B, Hc, h, w = hp.shape
        rh = _z(B, Hc, h, w, device=hp.device)
        zero = torch.zeros_like(gates)
        hb.check(hb.load().ssm_convgru_reset_fwd(hb.view_of(gates), hb.view_of(zero), hb.view_of(hp), hb.view_of(rh), hb.NULL_HVIEW,
                                                 B, Hc, h, w, 0, hb.stream_ptr()))
        ctx.save_for_backward(gates, hp)
        return rh

    @staticmethod
    def backward(ctx, drh):
        gates, hp = ctx.saved_tensors
        B, Hc, h, w = hp.shape
        dg, dhp = torch.zeros_like(gates), _z(B, Hc, h, w, device=hp.device)
        hb.check(hb.load().ssm_convgru_reset_bwd(hb.view_of(gates), hb.view_of(hp), hb.view_of(drh.contiguous()), hb.view_of(dg),
                                                 hb.view_of(dhp), B, Hc, h, w, hb.stream_ptr()))
        return dg, dhp


class _GRUUpdateFn(torch.autograd.Function):
    """h' = (1-u) h + u tanh(cand), u = sigmoid(gates[:, Hc:]);  h may be None (first step)."""

    @staticmethod
    def forward(ctx, gates, cand, hp):
        gates, cand = gates.detach().contiguous(), cand.detach().contiguous()
        hp = None if hp is None else hp.detach().contiguous()
        B, Hc, h, w = cand.shape
        hn = _z(B, Hc, h, w, device=cand.device)
        nv = hb.NULL_VIEW
        zg, zc = (torch.zeros_like(gates), torch.zeros_like(cand)) if hp is not None else (None, None)
        hb.check(hb.load().ssm_convgru_update_fwd(hb.view_of(gates), hb.view_of(zg) if hp is not None else nv, hb.view_of(cand),
                                                  hb.view_of(zc) if hp is not None else nv, hb.view_of(hp) if hp is not None else nv,
                                                  hb.view_of(hn), hb.NULL_HVIEW, B, Hc, h, w, 0, hb.stream_ptr()))
        ctx.save_for_backward(gates, cand, hp if hp is not None else gates.new_zeros(0))
        ctx.has_h = hp is not None
        return hn

    @staticmethod
    def backward(ctx, dhn):
        gates, cand, hp = ctx.saved_tensors
        B, Hc, h, w = cand.shape
        dg, dq = torch.zeros_like(gates), _z(B, Hc, h, w, device=cand.device)
        dhp = _z(B, Hc, h, w, device=cand.device) if ctx.has_h else None
        nv = hb.NULL_VIEW
        hb.check(hb.load().ssm_convgru_update_bwd(hb.view_of(gates), hb.view_of(cand), hb.view_of(hp) if ctx.has_h else nv,
                                                  hb.view_of(dhn.contiguous()), hb.view_of(dg), hb.view_of(dq),
                                                  hb.view_of(dhp) if ctx.has_h else nv, B, Hc, h, w, hb.stream_ptr()))
        return dg, dq, dhp


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
        if torch.is_grad_enabled() and (xforward.requires_grad or any(p.requires_grad for p in self.parameters())):
            return self._forward_autograd(xforward, xreverse)
        B, T, C, h, w = xforward.shape
        rb, pf, pr, out = self._plan(B, T, h, w, xforward.device)
        pf.load(xforward.transpose(0, 1).reshape(T * B, C, h, w))
        pr.load(xreverse.flip(1).transpose(0, 1).reshape(T * B, C, h, w))      # back to slot (window) order
        rb.run(pf, out, x_rev=pr)
        return out.to_nchw().reshape(T, B, 512, h, w).transpose(0, 1).contiguous()

    # ---- training: the same recurrence step by step on operators that carry their own autograd -----------------------
    def _net_autograd(self, net, xs):
        from ..layers import conv_forward
        seq = xs
        for cell in net.cell_list:
            hid = cell.hidden_dim
            h = c = None
            outs = []
            for x in seq:
                B, _, hh, ww = x.shape
                hp = h if h is not None else torch.zeros(B, hid, hh, ww, dtype=torch.float32, device=x.device)
                if self.KIND == "CLSTM":
                    gates = conv_forward(cell.conv, torch.cat([x, hp], dim=1), lrelu=False)
                    h, c = _LSTMCellFn.apply(gates, c)
                else:
                    gates = conv_forward(cell.conv_gates, torch.cat([x, hp], dim=1), lrelu=False)
                    rh = _GRUResetFn.apply(gates, hp) if h is not None else hp
                    cand = conv_forward(cell.conv_can, torch.cat([x, rh], dim=1), lrelu=False)
                    h = _GRUUpdateFn.apply(gates, cand, h)
                outs.append(h)
            seq = outs
        return seq

    def _forward_autograd(self, xforward, xreverse):
        yf = self._net_autograd(self.forward_net, list(xforward.unbind(1)))
        yr = self._net_autograd(self.reverse_net, list(xreverse.unbind(1)))[::-1]
        return torch.stack([torch.cat([a, b], dim=1) for a, b in zip(yf, yr)], dim=1)
