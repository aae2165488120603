"""Numerical study (CPU, not a test): 3x3 layers as Winograd F(2x4, 3x3) - F(2,3) along y, F(4,3) along x (24 instead of 32
multiplies per 8 outputs) - in fp32 on the oracle's whole pair -> frame path, against float64 and against the direct form.

    python tests/emulate_winograd_f24_precision.py [H] [W] [t]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import ssm_oracle as O  # noqa: E402
from ssm_amd.weights import synthetic_frames, synthetic_state_dict  # noqa: E402

torch.set_num_threads(8)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
W = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ts = [float(sys.argv[3])] if len(sys.argv) > 3 else [0.5]

G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def wino24(x, w, b):
    dt = x.dtype
    g2, bt2, at2, g4, bt4, at4 = (m.to(dt) for m in (G2, BT2, AT2, G4, BT4, AT4))
    Bn, C, Hh, Ww = x.shape
    N = w.shape[0]
    U = torch.einsum("ik,nckl,jl->ijnc", g2, w, g4)                      # [4,6,N,C]
    xp = F.pad(x, (1, 1 + (-Ww) % 4, 1, 1))
    th, tw = Hh // 2, (Ww + 3) // 4
    d = torch.stack([torch.stack([xp[:, :, i:i + 2 * th:2, j:j + 4 * tw:4] for j in range(6)], 0) for i in range(4)], 0)   # [4,6,B,C,th,tw]
    V = torch.einsum("ik,klbcyx->ilbcyx", bt2, d)
    V = torch.einsum("jl,ilbcyx->ijbcyx", bt4, V)
    V = V.permute(0, 1, 3, 2, 4, 5).reshape(24, C, Bn * th * tw)
    M = torch.bmm(U.reshape(24, N, C), V).reshape(4, 6, N, Bn, th, tw)
    Y = torch.einsum("ai,ijnbyx->ajnbyx", at2, M)
    Y = torch.einsum("cj,ajnbyx->acnbyx", at4, Y)                         # [2,4,N,B,th,tw]
    y = torch.empty(N, Bn, Hh, 4 * tw, dtype=dt)
    for a in range(2):
        for c in range(4):
            y[:, :, a::2, c::4] = Y[a, c]
    return y[:, :, :, :Ww].permute(1, 0, 2, 3) + b.view(1, -1, 1, 1)


orig_conv = O.conv2d


def conv_w(x, w, b):
    if w.shape[-1] == 3 and w.shape[1] >= 32 and w.shape[0] >= 32 and x.shape[2] % 2 == 0 and x.shape[3] % 4 == 0:
        return wino24(x, w, b)
    return orig_conv(x, w, b)


def run(p1, p2, img6, dtype, wino):
    O.conv2d = conv_w if wino else orig_conv
    q1 = {k: v.to(dtype) for k, v in p1.items()}
    q2 = {k: v.to(dtype) for k, v in p2.items()}
    out = torch.cat(O.interpolate_pair(q1, q2, img6.to(dtype), ts), 0)
    O.conv2d = orig_conv
    return out


def main():
    xx = torch.randn(1, 8, 12, 16, dtype=torch.float64)
    ww = torch.randn(5, 8, 3, 3, dtype=torch.float64)
    bb = torch.randn(5, dtype=torch.float64)
    ref = F.conv2d(xx, ww, bb, padding=1)
    assert (wino24(xx, ww, bb) - ref).abs().max().item() < 1e-11
    print("single layer fp32: F(2x4) err %.2e, direct err %.2e" % ((wino24(xx.float(), ww.float(), bb.float()).double() - ref).abs().max().item(),
                                                                    (F.conv2d(xx.float(), ww.float(), bb.float(), padding=1).double() - ref).abs().max().item()))
    p1, p2 = synthetic_state_dict(1), synthetic_state_dict(2)
    x = synthetic_frames(2, H, W, seed=42)
    img6 = torch.cat([x[:, 0], x[:, 1]], 1)
    truth = run(p1, p2, img6, torch.float64, False)
    direct = run(p1, p2, img6, torch.float32, False)
    wino = run(p1, p2, img6, torch.float32, True)
    print("size %dx%d t=%s, 3x3 layers as F(2x4,3x3)" % (H, W, ts))
    print("  direct fp32 vs float64: %.3e" % (direct.double() - truth).abs().max().item())
    print("  F(2x4) fp32 vs float64: %.3e" % (wino.double() - truth).abs().max().item())
    print("  F(2x4) fp32 vs direct fp32: %.3e" % (wino - direct).abs().max().item())


if __name__ == "__main__":
    main()
