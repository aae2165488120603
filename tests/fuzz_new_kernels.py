"""Study (GPU box, not collected by pytest): random shapes through the r1p/r1q kernels - ssm_conv2d_wgrad_bf16x3 against autograd of
the oracle convolution and the sub-pixel upsample+conv against upsample + conv of the oracle.  python tests/fuzz_new_kernels.py [n]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
for p in (ROOT, PKG):
    sys.path.insert(0, p)
import torch  # noqa: E402

from oracle import ssm_oracle as O  # noqa: E402
from ssm_amd import backward as Bk  # noqa: E402
from ssm_amd import hipbind as hb  # noqa: E402
from ssm_amd.subpixel import SubpixelUpConv  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda:0")
    rnd = random.Random(7)
    worst_w = worst_s = 0.0
    for it in range(n):
        k = rnd.choice([3, 3, 3, 5, 7])
        cin, cout = rnd.choice([3, 6, 16, 24, 32, 48, 64, 96, 128, 200]), rnd.choice([4, 5, 16, 32, 40, 64, 128, 136])
        B, H, W = rnd.choice([1, 2, 3]), rnd.randint(1, 40), rnd.randint(1, 150)
        g = torch.Generator().manual_seed(it)
        x = torch.randn(B, cin, H, W, generator=g)
        w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).requires_grad_()
        r = torch.randn(B, cout, H, W, generator=g) * 10.0 ** rnd.randint(-6, 1)
        (O.conv2d(x, w, torch.zeros(cout)) * r).sum().backward()
        xp = hb.Planes(B, cin, H, W, dev).load(x.to(dev))
        dzp = hb.Planes(B, cout, H, W, dev).load(r.to(dev))
        dw = Bk.wgrad(xp, dzp, torch.full((cout, cin, k, k), 3.0, device=dev), k, split=True)
        err = float((dw.cpu() - w.grad).abs().max() / (w.grad.abs().max() + 1e-30))
        worst_w = max(worst_w, err)
        assert err < 3e-4, ("wgrad", k, cin, cout, B, H, W, err)
    for it in range(n):
        ca, cb, co = rnd.choice([16, 32, 64, 128]), rnd.choice([0, 16, 32, 64]), rnd.choice([16, 32, 64, 96])
        B, h, w = rnd.choice([1, 2, 3]), rnd.randint(2, 24), rnd.randint(2, 90)
        g = torch.Generator().manual_seed(1000 + it)
        a = torch.randn(B, ca, h, w, generator=g)
        b = torch.randn(B, cb, h, w, generator=g) if cb else None
        wt = torch.randn(co, ca + cb, 3, 3, generator=g) / ((ca + cb) * 9) ** 0.5
        bias = torch.randn(co, generator=g) * 0.1
        A = hb.HPlanes(B, ca, h, w, dev, q8=True).load(a.to(dev))
        Bp = hb.HPlanes(B, cb, h, w, dev, q8=True).load(b.to(dev)) if cb else None
        aq, bq = A.to_nchw().cpu(), (Bp.to_nchw().cpu() if cb else None)
        want = O.conv2d_lrelu(O.upsample2x_bilinear(torch.cat([aq, bq], 1) if cb else aq), wt, bias)
        dst = hb.HPlanes(B, co, 2 * h, 2 * w, dev, q8=True)
        sp = SubpixelUpConv(wt, bias, A.G, Bp.G if cb else 0, B, h, w, dev)
        sp.run(lambda y0, x0: A.view(y0=y0, x0=x0), (lambda y0, x0: Bp.view(y0=y0, x0=x0)) if cb else None, dst)
        err = float((dst.to_nchw().cpu() - want).abs().max()) / max(float(want.abs().max()), 1.0)
        worst_s = max(worst_s, err)
        assert err < 2e-4, ("subpixel", B, ca, cb, co, h, w, err)
    print("fuzz: %d wgrad shapes (worst rel err %.2e), %d sub-pixel shapes (worst %.2e of scale): all within bars" % (n, worst_w, n, worst_s))


if __name__ == "__main__":
    main()
