"""Numerical study (CPU, not a test): the 7x7 / 5x5 layers as ONE-dimensional Winograd F(m, r) along x (direct form along y) in fp32.
Cook-Toom transforms built in float64 for a chosen point set, applied in fp32; whole pair -> frame path on the oracle.

    python tests/emulate_winograd_1d_precision.py [H] [W] [spec: e.g. 7:2,5:4 = F(2,7) and F(4,5)] [t]
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
SPEC = {int(a.split(":")[0]): int(a.split(":")[1]) for a in (sys.argv[3] if len(sys.argv) > 3 else "7:2,5:4").split(",")}
ts = [float(sys.argv[4])] if len(sys.argv) > 4 else [0.5]

POINTS = {6: [0.0, 1.0, -1.0, 2.0, -2.0],
          8: [0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5],
          10: [0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5, 4.0, -4.0] if os.environ.get("PTS10", "4") == "4" else
              [0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5, 0.25, -0.25]}


def cook_toom(m, r, pts):
    """1-D F(m, r): y = AT [ (G g) .* (BT d) ], n = m + r - 1 points (the last one is infinity)."""
    n = m + r - 1
    assert len(pts) == n - 1

    def V(k):
        M = np.zeros((n, k))
        for i, p in enumerate(pts):
            M[i] = [p ** j for j in range(k)]
        M[n - 1, k - 1] = 1.0
        return M
    Vn, Vr, Vm = V(n), V(r), V(m)
    AT, G, BT = Vm.T.copy(), Vr.copy(), np.linalg.inv(Vn).T.copy()
    for i in range(n):      # rows of BT scaled to smallest non-zero entry 1, compensated in G
        nz = np.abs(BT[i][np.abs(BT[i]) > 1e-12])
        f = nz.min()
        BT[i] /= f
        G[i] *= f
    rng = np.random.default_rng(0)
    g, d = rng.standard_normal(r), rng.standard_normal(n)
    y = AT @ ((G @ g) * (BT @ d))
    want = np.array([sum(g[k] * d[i + k] for k in range(r)) for i in range(m)])
    assert np.abs(y - want).max() < 1e-9
    return AT, G, BT


MATS = {}
for r, m in SPEC.items():
    AT, G, BT = cook_toom(m, r, POINTS[m + r - 1])
    MATS[r] = (m,) + tuple(torch.tensor(M, dtype=torch.float64) for M in (AT, G, BT))
    print("F(%d,%d): %d points, max|BT| %.3g  max|G| %.3g  max|AT| %.3g; multiplies per output %.2f (direct %d)"
          % (m, r, m + r - 1, np.abs(BT).max(), np.abs(G).max(), np.abs(AT).max(), (m + r - 1) / m, r))


def wino1d_conv(x, w, b, r):
    """F(m, r) along x, direct along y, in x.dtype.  x [B,C,H,W] (W % m == 0), w [N,C,r,r]."""
    m = MATS[r][0]
    AT, G, BT = (M.to(x.dtype) for M in MATS[r][1:])
    n = m + r - 1
    pad = (r - 1) // 2
    Bn, C, Hh, Ww = x.shape
    N = w.shape[0]
    U = torch.einsum("fk,ncyk->fncy", G, w)                             # [n,N,C,r(ky)]
    xp = F.pad(x, (pad, pad, pad, pad))
    tw = Ww // m
    d = torch.stack([xp[:, :, :, j:j + m * tw:m] for j in range(n)], 0)  # [n,B,C,Hp,tw]
    V = torch.einsum("fj,jbchx->fbchx", BT, d)                           # [n,B,C,Hp,tw]
    Ms = torch.stack([F.conv2d(V[f], U[f].unsqueeze(-1)) for f in range(n)], 0)     # [n,B,N,H,tw]
    Y = torch.einsum("af,fbnhx->abnhx", AT, Ms)                          # [m,B,N,H,tw]
    y = torch.empty(Bn, N, Hh, Ww, dtype=x.dtype)
    for a in range(m):
        y[:, :, :, a::m] = Y[a]
    return y + b.view(1, -1, 1, 1)


orig_conv = O.conv2d


def conv_w(x, w, b):
    k = w.shape[-1]
    if k in SPEC and x.shape[-1] % SPEC[k] == 0:
        return wino1d_conv(x, w, b, k)
    return orig_conv(x, w, b)


def run(p1, p2, img6, dtype, wino):
    O.conv2d = conv_w if wino else orig_conv
    q1 = {k: v.to(dtype) for k, v in p1.items()}
    q2 = {k: v.to(dtype) for k, v in p2.items()}
    out = torch.cat(O.interpolate_pair(q1, q2, img6.to(dtype), ts), 0)
    O.conv2d = orig_conv
    return out


def main():
    for r in SPEC:
        xx = torch.randn(1, 32, 12, 16, dtype=torch.float64)
        ww = torch.randn(8, 32, r, r, dtype=torch.float64) / (32 * r * r) ** 0.5
        bb = torch.randn(8, dtype=torch.float64)
        ref = F.conv2d(xx, ww, bb, padding=(r - 1) // 2)
        e = (wino1d_conv(xx, ww, bb, r) - ref).abs().max().item()
        assert e < 1e-9, e
        e32 = (wino1d_conv(xx.float(), ww.float(), bb.float(), r).double() - ref).abs().max().item()
        d32 = (F.conv2d(xx.float(), ww.float(), bb.float(), padding=(r - 1) // 2).double() - ref).abs().max().item()
        print("single %dx%d layer (32 channels, unit-variance output) in fp32: 1-D winograd err %.2e, direct err %.2e" % (r, r, e32, d32))
    p1, p2 = synthetic_state_dict(1), synthetic_state_dict(2)
    x = synthetic_frames(2, H, W, seed=42)
    img6 = torch.cat([x[:, 0], x[:, 1]], 1)
    truth = run(p1, p2, img6, torch.float64, False)
    direct = run(p1, p2, img6, torch.float32, False)
    wino = run(p1, p2, img6, torch.float32, True)
    print("size %dx%d t=%s layers %s as 1-D F(m,k) along x" % (H, W, ts, SPEC))
    print("  direct fp32   vs float64: %.3e" % (direct.double() - truth).abs().max().item())
    print("  winograd fp32 vs float64: %.3e" % (wino.double() - truth).abs().max().item())
    print("  winograd fp32 vs direct fp32: %.3e" % (wino - direct).abs().max().item())


if __name__ == "__main__":
    main()
