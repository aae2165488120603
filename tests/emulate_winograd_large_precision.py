"""Numerical study (CPU, not a test): could the 5x5 and 7x7 layers run as Winograd F(2x2,5x5) / F(2x2,7x7) in fp32?
Cook-Toom transforms built in float64 for a chosen point set, applied in fp32 (nested 1-D), whole pair -> frame path on the oracle.

    python tests/emulate_winograd_large_precision.py [H] [W] [which: 5|7|57] [t]
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
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
WHICH = sys.argv[3] if len(sys.argv) > 3 else "57"
ts = [float(sys.argv[4])] if len(sys.argv) > 4 else [0.5]


def cook_toom(m, r, pts):
    """1-D F(m, r): y = AT [ (G g) .* (BT d) ], n = m + r - 1 points (the last one is infinity).  Returns AT [m,n], G [n,r], BT [n,n]."""
    n = m + r - 1
    assert len(pts) == n - 1
    # evaluation (Vandermonde incl. infinity) of degree-(k-1) polynomials at the points
    def V(k):
        M = np.zeros((n, k))
        for i, p in enumerate(pts):
            M[i] = [p ** j for j in range(k)]
        M[n - 1, k - 1] = 1.0
        return M
    # linear convolution s = g * d (sizes r and m) via evaluation/interpolation: s = Vn^-1 [ (Vr g) .* (Vm d) ]
    # correlation form F(m, r) is the transpose: y = Vm^T [ (Vr g) .* (Vn^-T d) ]
    Vn, Vr, Vm = V(n), V(r), V(m)
    AT = Vm.T
    G = Vr
    BT = np.linalg.inv(Vn).T
    # move the scaling out of BT into G (keeps BT entries small integers for the usual point sets)
    sc = np.abs(BT).max(axis=1)
    sc[sc == 0] = 1.0
    # scale rows of BT to have smallest-magnitude nonzero entry 1 (typical presentation); compensate in G
    for i in range(n):
        nz = np.abs(BT[i][np.abs(BT[i]) > 1e-12])
        f = nz.min()
        BT[i] /= f
        G[i] *= f
    return AT, G, BT


def check(m, r, pts):
    AT, G, BT = cook_toom(m, r, pts)
    rng = np.random.default_rng(0)
    g, d = rng.standard_normal(r), rng.standard_normal(m + r - 1)
    y = AT @ ((G @ g) * (BT @ d))
    want = np.array([sum(g[k] * d[i + k] for k in range(r)) for i in range(m)])
    assert np.abs(y - want).max() < 1e-9, (y, want)
    return AT, G, BT


PTS = {5: [0.0, 1.0, -1.0, 2.0, -2.0], 7: [0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5]}
MATS = {}
for r in (5, 7):
    AT, G, BT = check(2, r, PTS[r])
    MATS[r] = tuple(torch.tensor(M, dtype=torch.float64) for M in (AT, G, BT))
    print("F(2,%d): max|BT| %.3g  max|G| %.3g  max|AT| %.3g" % (r, np.abs(BT).max(), np.abs(G).max(), np.abs(AT).max()))


def wino_conv(x, w, b, r):
    """F(2x2, r x r) in x.dtype.  x [B,C,H,W] (even H, W), w [N,C,r,r]."""
    AT, G, BT = (M.to(x.dtype) for M in MATS[r])
    n = r + 1
    pad = (r - 1) // 2
    Bn, C, Hh, Ww = x.shape
    N = w.shape[0]
    U = torch.einsum("ik,nckl,jl->ijnc", G, w, G)                       # [n,n,N,C]
    xp = F.pad(x, (pad, pad, pad, pad))
    th, tw = Hh // 2, Ww // 2
    d = torch.stack([torch.stack([xp[:, :, i:i + 2 * th:2, j:j + 2 * tw:2] for j in range(n)], 0) for i in range(n)], 0)   # [n,n,B,C,th,tw]
    V = torch.einsum("ik,klbcyx->ilbcyx", BT, d)
    V = torch.einsum("jl,ilbcyx->ijbcyx", BT, V)                         # [n,n,B,C,th,tw]
    V = V.permute(0, 1, 3, 2, 4, 5).reshape(n * n, C, Bn * th * tw)
    M = torch.bmm(U.reshape(n * n, N, C), V).reshape(n, n, N, Bn, th, tw)
    Y = torch.einsum("ai,ijnbyx->ajnbyx", AT, M)
    Y = torch.einsum("cj,ajnbyx->acnbyx", AT, Y)                         # [2,2,N,B,th,tw]
    y = torch.empty(N, Bn, Hh, Ww, dtype=x.dtype)
    for a in range(2):
        for c in range(2):
            y[:, :, a::2, c::2] = Y[a, c]
    return y.permute(1, 0, 2, 3) + b.view(1, -1, 1, 1)


orig_conv = O.conv2d


def conv_w(x, w, b):
    k = w.shape[-1]
    if str(k) in WHICH and k in (5, 7):
        return wino_conv(x, w, b, k)
    return orig_conv(x, w, b)


def run(p1, p2, img6, dtype, wino):
    O.conv2d = conv_w if wino else orig_conv
    q1 = {k: v.to(dtype) for k, v in p1.items()}
    q2 = {k: v.to(dtype) for k, v in p2.items()}
    out = torch.cat(O.interpolate_pair(q1, q2, img6.to(dtype), ts), 0)
    O.conv2d = orig_conv
    return out


def main():
    for r in (5, 7):
        xx = torch.randn(1, 4, 12, 16, dtype=torch.float64)
        ww = torch.randn(3, 4, r, r, dtype=torch.float64)
        bb = torch.randn(3, dtype=torch.float64)
        e = (wino_conv(xx, ww, bb, r) - F.conv2d(xx, ww, bb, padding=(r - 1) // 2)).abs().max().item()
        assert e < 1e-9, e
        e32 = (wino_conv(xx.float(), ww.float(), bb.float(), r).double() - F.conv2d(xx, ww, bb, padding=(r - 1) // 2)).abs().max().item()
        d32 = (F.conv2d(xx.float(), ww.float(), bb.float(), padding=(r - 1) // 2).double() - F.conv2d(xx, ww, bb, padding=(r - 1) // 2)).abs().max().item()
        print("single %dx%d layer in fp32: winograd err %.2e, direct err %.2e" % (r, r, e32, d32))
    p1, p2 = synthetic_state_dict(1), synthetic_state_dict(2)
    x = synthetic_frames(2, H, W, seed=42)
    img6 = torch.cat([x[:, 0], x[:, 1]], 1)
    truth = run(p1, p2, img6, torch.float64, False)
    direct = run(p1, p2, img6, torch.float32, False)
    wino = run(p1, p2, img6, torch.float32, True)
    print("size %dx%d t=%s layers k in '%s' as F(2x2,kxk)" % (H, W, ts, WHICH))
    print("  direct fp32   vs float64: %.3e" % (direct.double() - truth).abs().max().item())
    print("  winograd fp32 vs float64: %.3e" % (wino.double() - truth).abs().max().item())
    print("  winograd fp32 vs direct fp32: %.3e" % (wino - direct).abs().max().item())


if __name__ == "__main__":
    main()
