"""Numerical study (CPU, not a test): what does evaluating the 3x3 convolutions as Winograd F(2x2,3x3) in fp32 cost in parity?
The whole pair -> frame path is run on the oracle (a) as is (direct fp32 = the reference's arithmetic), (b) with every 3x3 convolution
whose Cin >= MIN_CIN replaced by an fp32 emulation of  Y = A^T [ (G g G^T) .* (B^T d B) ] A  (channel sum per frequency in fp32), and
(c) in float64 (the truth both are measured against).

    python tests/emulate_winograd_precision.py [H] [W] [min_cin] [t ...]      (default 256 256 32 0.5)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import ssm_oracle as O  # noqa: E402
from ssm_amd.weights import synthetic_frames, synthetic_state_dict  # noqa: E402

torch.set_num_threads(8)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
MIN_CIN = int(sys.argv[3]) if len(sys.argv) > 3 else 32
ts = [float(v) for v in sys.argv[4:]] or [0.5]


def winograd_conv3(x, w, b):
    """F(2x2,3x3), everything in x.dtype.  x [B,C,H,W] (H, W even), w [N,C,3,3]."""
    Bn, C, Hh, Ww = x.shape
    N = w.shape[0]
    # U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    g0, g1, g2 = w[:, :, 0], w[:, :, 1], w[:, :, 2]            # rows [N,C,3]
    r = [g0, 0.5 * (g0 + g1 + g2), 0.5 * (g0 - g1 + g2), g2]   # G g : 4 rows of [N,C,3]
    U = []
    for ri in r:
        c0, c1, c2 = ri[..., 0], ri[..., 1], ri[..., 2]
        U += [c0, 0.5 * (c0 + c1 + c2), 0.5 * (c0 - c1 + c2), c2]
    U = torch.stack(U, 0)                                       # [16,N,C]
    xp = F.pad(x, (1, 1, 1, 1))
    th, tw = Hh // 2, Ww // 2
    # d[i][j] = xp[..., 2ty+i, 2tx+j]
    d = [[xp[:, :, i:i + 2 * th:2, j:j + 2 * tw:2] for j in range(4)] for i in range(4)]
    # B^T d : rows  d0-d2, d1+d2, d2-d1, d1-d3
    bd = [[d[0][j] - d[2][j] for j in range(4)], [d[1][j] + d[2][j] for j in range(4)],
          [d[2][j] - d[1][j] for j in range(4)], [d[1][j] - d[3][j] for j in range(4)]]
    V = []
    for i in range(4):
        q = bd[i]
        V += [q[0] - q[2], q[1] + q[2], q[2] - q[1], q[1] - q[3]]
    V = torch.stack(V, 0).permute(0, 2, 1, 3, 4).reshape(16, C, Bn * th * tw)      # [16,C,P]
    M = torch.bmm(U, V).reshape(16, N, Bn, th, tw)                                   # [16,N,B,th,tw]
    m = [[M[4 * i + j] for j in range(4)] for i in range(4)]
    # A^T = [[1,1,1,0],[0,1,-1,-1]]
    trow = [[m[0][j] + m[1][j] + m[2][j] for j in range(4)], [m[1][j] - m[2][j] - m[3][j] for j in range(4)]]
    y = torch.empty(N, Bn, Hh, Ww, dtype=x.dtype)
    for a in range(2):
        q = trow[a]
        y[:, :, a::2, 0::2] = q[0] + q[1] + q[2]
        y[:, :, a::2, 1::2] = q[1] - q[2] - q[3]
    return y.permute(1, 0, 2, 3) + b.view(1, -1, 1, 1)


orig_conv = O.conv2d
COUNT = [0, 0]


def conv_w(x, w, b):
    if w.shape[-1] == 3 and w.shape[1] >= MIN_CIN and w.shape[0] >= 32 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
        COUNT[0] += 1
        return winograd_conv3(x, w, b)
    COUNT[1] += 1
    return orig_conv(x, w, b)


def run(p1, p2, img6, dtype, wino):
    O.conv2d = conv_w if wino else orig_conv
    q1 = {k: v.to(dtype) for k, v in p1.items()}
    q2 = {k: v.to(dtype) for k, v in p2.items()}
    t0 = time.time()
    out = torch.cat(O.interpolate_pair(q1, q2, img6.to(dtype), ts), 0)
    O.conv2d = orig_conv
    return out, time.time() - t0


def main():
    p1, p2 = synthetic_state_dict(1), synthetic_state_dict(2)
    x = synthetic_frames(2, H, W, seed=42)
    img6 = torch.cat([x[:, 0], x[:, 1]], 1)
    # self-check of the emulation in float64
    xx = torch.randn(1, 8, 12, 16, dtype=torch.float64)
    ww = torch.randn(5, 8, 3, 3, dtype=torch.float64)
    bb = torch.randn(5, dtype=torch.float64)
    e = (winograd_conv3(xx, ww, bb) - F.conv2d(xx, ww, bb, padding=1)).abs().max().item()
    assert e < 1e-12, e
    truth, tt = run(p1, p2, img6, torch.float64, False)
    direct, td = run(p1, p2, img6, torch.float32, False)
    wino, tw = run(p1, p2, img6, torch.float32, True)
    print("size %dx%d  t=%s  winograd for 3x3 layers with Cin >= %d (%d convs winograd, %d direct)" % (H, W, ts, MIN_CIN, COUNT[0], COUNT[1]))
    print("  direct fp32   vs float64: %.3e   (%.1f s)" % ((direct.double() - truth).abs().max().item(), td))
    print("  winograd fp32 vs float64: %.3e   (%.1f s)" % ((wino.double() - truth).abs().max().item(), tw))
    print("  winograd fp32 vs direct fp32 (what the parity tests see): %.3e" % (wino - direct).abs().max().item())


if __name__ == "__main__":
    main()
