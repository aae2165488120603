"""Numerical study (CPU, not a test): 3x3 layers as Winograd F(my x mx, 3x3) with my, mx in {2, 4} - F(4x4,3x3): 36 instead of 64
multiplies per 16 outputs of F(2x2,3x3) - in fp32 on the oracle's whole pair -> frame path, against float64 and the direct form.
W1D=1 also evaluates the 7x7 / 5x5 layers in the 1-D forms of csrc/ssm_wino1d.hip (F(2,7), F(4,5) along x).

    python tests/emulate_winograd_f44_precision.py [H] [W] [my] [mx] [t]          (PTS=half: points 0, +-1, +-1/2 instead of 0, +-1, +-2; PTS=rec: 0, +-5/8, +-8/5 - the kernel's)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import ssm_oracle as O  # noqa: E402
from ssm_amd.weights import synthetic_frames, synthetic_state_dict  # noqa: E402

torch.set_num_threads(8)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
W = int(sys.argv[2]) if len(sys.argv) > 2 else 128
MY = int(sys.argv[3]) if len(sys.argv) > 3 else 4
MX = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ts = [float(sys.argv[5])] if len(sys.argv) > 5 else [0.5]


def cook_toom(m, r, pts):
    n = m + r - 1

    def V(k):
        M = np.zeros((n, k))
        for i, p in enumerate(pts):
            M[i] = [p ** j for j in range(k)]
        M[n - 1, k - 1] = 1.0
        return M
    Vn, Vr, Vm = V(n), V(r), V(m)
    AT, G, BT = Vm.T.copy(), Vr.copy(), np.linalg.inv(Vn).T.copy()
    for i in range(n):
        nz = np.abs(BT[i][np.abs(BT[i]) > 1e-12])
        f = nz.min()
        BT[i] /= f
        G[i] *= f
    return tuple(torch.tensor(M, dtype=torch.float64) for M in (AT, G, BT))


PTS4 = {"half": [0.0, 1.0, -1.0, 0.5, -0.5], "rec": [0.0, 0.625, -0.625, 1.6, -1.6]}.get(os.environ.get("PTS", ""), [0.0, 1.0, -1.0, 2.0, -2.0])
MATS = {2: cook_toom(2, 3, [0.0, 1.0, -1.0]), 4: cook_toom(4, 3, PTS4)}
MATS1D = {7: (2,) + cook_toom(2, 7, [0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5]), 5: (4,) + cook_toom(4, 5, [0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5])}


def wino2d(x, w, b):
    dt = x.dtype
    aty, gy, bty = (m.to(dt) for m in MATS[MY])
    atx, gx, btx = (m.to(dt) for m in MATS[MX])
    ny, nx = MY + 2, MX + 2
    Bn, C, Hh, Ww = x.shape
    N = w.shape[0]
    U = torch.einsum("ik,nckl,jl->ijnc", gy, w, gx)
    th, tw = (Hh + MY - 1) // MY, (Ww + MX - 1) // MX
    xp = F.pad(x, (1, 1 + tw * MX - Ww, 1, 1 + th * MY - Hh))
    d = torch.stack([torch.stack([xp[:, :, i:i + MY * th:MY, j:j + MX * tw:MX] for j in range(nx)], 0) for i in range(ny)], 0)
    V = torch.einsum("ik,klbcyx->ilbcyx", bty, d)
    V = torch.einsum("jl,ilbcyx->ijbcyx", btx, V)
    V = V.permute(0, 1, 3, 2, 4, 5).reshape(ny * nx, C, Bn * th * tw)
    M = torch.bmm(U.reshape(ny * nx, N, C), V).reshape(ny, nx, N, Bn, th, tw)
    Y = torch.einsum("ai,ijnbyx->ajnbyx", aty, M)
    Y = torch.einsum("cj,ajnbyx->acnbyx", atx, Y)
    y = torch.empty(N, Bn, MY * th, MX * tw, dtype=dt)
    for a in range(MY):
        for c in range(MX):
            y[:, :, a::MY, c::MX] = Y[a, c]
    return y[:, :, :Hh, :Ww].permute(1, 0, 2, 3) + b.view(1, -1, 1, 1)


def wino1d(x, w, b, r):
    m, AT, G, BT = MATS1D[r]
    AT, G, BT = (M.to(x.dtype) for M in (AT, G, BT))
    n, pad = m + r - 1, (r - 1) // 2
    Bn, C, Hh, Ww = x.shape
    N = w.shape[0]
    U = torch.einsum("fk,ncyk->fncy", G, w)
    tw = (Ww + m - 1) // m
    xp = F.pad(x, (pad, pad + tw * m - Ww, pad, pad))
    d = torch.stack([xp[:, :, :, j:j + m * tw:m] for j in range(n)], 0)
    V = torch.einsum("fj,jbchx->fbchx", BT, d)
    Ms = torch.stack([F.conv2d(V[f], U[f].unsqueeze(-1)) for f in range(n)], 0)
    Y = torch.einsum("af,fbnhx->abnhx", AT, Ms)
    y = torch.empty(Bn, N, Hh, m * tw, dtype=x.dtype)
    for a in range(m):
        y[:, :, :, a::m] = Y[a]
    return y[:, :, :, :Ww] + b.view(1, -1, 1, 1)


orig_conv = O.conv2d
W1D = os.environ.get("W1D", "0") != "0"


def conv_w(x, w, b):
    k = w.shape[-1]
    if k == 3 and w.shape[1] >= 32 and w.shape[0] >= 32:
        return wino2d(x, w, b)
    if W1D and k in (5, 7):
        return wino1d(x, w, b, k)
    return orig_conv(x, w, b)


def run(p1, p2, img6, dtype, wino):
    O.conv2d = conv_w if wino else orig_conv
    q1 = {k: v.to(dtype) for k, v in p1.items()}
    q2 = {k: v.to(dtype) for k, v in p2.items()}
    out = torch.cat(O.interpolate_pair(q1, q2, img6.to(dtype), ts), 0)
    O.conv2d = orig_conv
    return out


def main():
    xx = torch.randn(1, 32, 16, 16, dtype=torch.float64)
    ww = torch.randn(8, 32, 3, 3, dtype=torch.float64) / (32 * 9) ** 0.5
    bb = torch.randn(8, dtype=torch.float64)
    ref = F.conv2d(xx, ww, bb, padding=1)
    assert (wino2d(xx, ww, bb) - ref).abs().max().item() < 1e-10
    print("single 3x3 layer (unit-variance output) fp32: F(%dx%d) err %.2e, direct err %.2e" % (
        MY, MX, (wino2d(xx.float(), ww.float(), bb.float()).double() - ref).abs().max().item(),
        (F.conv2d(xx.float(), ww.float(), bb.float(), padding=1).double() - ref).abs().max().item()))
    p1, p2 = synthetic_state_dict(1), synthetic_state_dict(2)
    x = synthetic_frames(2, H, W, seed=42)
    img6 = torch.cat([x[:, 0], x[:, 1]], 1)
    truth = run(p1, p2, img6, torch.float64, False)
    direct = run(p1, p2, img6, torch.float32, False)
    wino = run(p1, p2, img6, torch.float32, True)
    print("size %dx%d t=%s, 3x3 layers as F(%dx%d,3x3)%s" % (H, W, ts, MY, MX, ", 7x7 / 5x5 as 1-D F(2,7) / F(4,5)" if W1D else ""))
    print("  direct fp32   vs float64: %.3e" % (direct.double() - truth).abs().max().item())
    print("  winograd fp32 vs float64: %.3e" % (wino.double() - truth).abs().max().item())
    print("  winograd fp32 vs direct fp32: %.3e" % (wino - direct).abs().max().item())


if __name__ == "__main__":
    main()
