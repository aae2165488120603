"""GPU parity of the fp16-MFMA convolution on HL8 (hi/lo fp16) activations against the CPU oracle.
Split mode (3 MFMAs per product) is held to fp32-grade tolerance; fast mode (1 MFMA, plain fp16 inputs)
to fp16-grade tolerance."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_hl8_roundtrip(dev):
    from ssm_amd import hipbind as hb
    x = torch.randn(2, 13, 9, 21) * 3
    x[0, 0, 0, 0] = 1e-6
    x[0, 1, 0, 0] = 1000.0
    p = hb.HPlanes(2, 13, 9, 21, dev).load(x.to(dev))
    y = p.to_nchw().cpu()
    assert ((y - x).abs() <= 3e-7 * x.abs() + 1e-7).all(), float((y - x).abs().max())   # 22+ bits
    assert float(p.buf.float().abs().sum()) > 0


CASES = [
    # k, cin, cout, B, H, W
    (7, 6, 32, 1, 16, 64), (7, 32, 32, 2, 21, 100),
    (5, 32, 64, 1, 16, 64), (5, 64, 64, 2, 19, 40),
    (3, 128, 32, 1, 16, 64), (3, 32, 32, 2, 9, 33), (3, 32, 5, 1, 16, 64),
    (3, 256, 64, 1, 16, 64), (3, 64, 64, 2, 11, 70),
    (3, 64, 128, 1, 8, 64), (3, 128, 256, 1, 12, 128), (3, 512, 512, 1, 23, 40), (3, 1024, 256, 1, 8, 96),
]


@pytest.mark.parametrize("k,cin,cout,B,H,W", CASES)
@pytest.mark.parametrize("fast", [False, True])
def test_conv16_vs_oracle(dev, k, cin, cout, B, H, W, fast):
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(k * 1000 + cin + cout + H + W)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    want = O.conv2d_lrelu(x, w, bias)
    pk = hb.PackedConv16(w.to(dev), bias.to(dev), W)
    src = hb.HPlanes(B, cin, H, W, dev, groups=pk.cin_p // 8).load(x.to(dev))
    y32 = torch.empty(B, cout, H, W, device=dev)
    dst = hb.HPlanes(B, cout, H, W, dev) if cout % 8 == 0 else None
    hb.conv2d_hl8(src.view(), pk.cin_p, None, 0, pk, dst.view() if dst else None, hb.view_of(y32), None, B, H, W,
                  lrelu=True, fast=fast)
    got = y32.cpu()
    tol = 2e-2 if fast else 5e-5
    err = float((got - want).abs().max())
    assert err < tol, "conv16 k%d %d->%d %dx%dx%d fast=%s: max err %.3e" % (k, cin, cout, B, H, W, fast, err)
    if dst is not None:
        got2 = dst.to_nchw().cpu()
        if fast:
            # mode FAST (r6) stores the hi plane only - no convolution of the mode reads a lo plane, and half the HBM bytes of the launch
            # go away: the HL8 output is the fp32 output rounded to fp16 (2^-11 relative), its lo plane keeps the zeros it was allocated with
            assert float((got2 - got).abs().max()) <= 2.0 ** -10 * float(got.abs().max()), "HL8 output is not fp16(fp32 output)"
            planes = dst.buf[:B * dst.G * 2 * dst.Hp * dst.Wp * 8].view(B, dst.G, 2, dst.Hp, dst.Wp, 8)
            assert float(planes[:, :, 1].abs().max()) == 0.0, "mode FAST wrote a lo plane"
        else:
            assert float((got2 - got).abs().max()) < 1e-5, "HL8 output differs from fp32 output"


def test_conv16_fused_pool_and_cat(dev):
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 16, 96
    a, b = torch.randn(B, 32, H, W, generator=g), torch.randn(B, 32, H, W, generator=g)
    w = torch.randn(32, 64, 3, 3, generator=g) / 24.0
    bias = torch.randn(32, generator=g) * 0.1
    want = O.conv2d_lrelu(torch.cat([a, b], 1), w, bias)
    pa, pb = hb.HPlanes(B, 32, H, W, dev).load(a.to(dev)), hb.HPlanes(B, 32, H, W, dev).load(b.to(dev))
    y, yp = hb.HPlanes(B, 32, H, W, dev), hb.HPlanes(B, 32, H // 2, W // 2, dev)
    pk = hb.PackedConv16(w.to(dev), bias.to(dev), W)
    hb.conv2d_hl8(pa.view(), 32, pb.view(), 32, pk, y.view(), None, yp.view(), B, H, W, lrelu=True)
    assert float((y.to_nchw().cpu() - want).abs().max()) < 5e-5
    assert float((yp.to_nchw().cpu() - O.avg_pool2(want)).abs().max()) < 5e-5


UPS_CASES = [
    # c1, c2, cout, B, h (low), w (low)      -> every fused tile configuration, image borders, ragged tiles, 1 source
    (32, 32, 32, 1, 8, 32), (64, 64, 32, 2, 12, 36), (128, 0, 32, 1, 4, 32),
    (128, 128, 64, 1, 8, 32), (64, 64, 64, 2, 5, 40),
    (256, 256, 128, 1, 6, 64), (64, 64, 128, 1, 10, 32),
    (512, 512, 256, 1, 6, 16), (128, 128, 512, 2, 3, 8), (16, 16, 128, 1, 2, 2),
]


@pytest.mark.parametrize("c1,c2,cout,B,h,w", UPS_CASES)
def test_conv16_fused_upsample_vs_oracle(dev, c1, c2, cout, B, h, w):
    """conv3x3(upsample2x(cat[a,b])) in ONE kernel (expander waves + matrix waves) against the oracle's
    upsample followed by conv; also with the second source batch-broadcast (the cross-skip case)."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(c1 + c2 + cout + h + w)
    a = torch.randn(B, c1, h, w, generator=g)
    b = torch.randn(1, c2, h, w, generator=g) if c2 else None          # broadcast over the batch
    cin = c1 + c2
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    cat = a if b is None else torch.cat([a, b.expand(B, -1, -1, -1)], 1)
    want = O.conv2d_lrelu(O.upsample2x_bilinear(cat), wt, bias)
    H, W = 2 * h, 2 * w
    pk = hb.PackedConv16(wt.to(dev), bias.to(dev), W)
    pa = hb.HPlanes(B, c1, h, w, dev).load(a.to(dev))
    pb = hb.HPlanes(1, c2, h, w, dev).load(b.to(dev)) if c2 else None
    y32 = torch.empty(B, cout, H, W, device=dev)
    dst = hb.HPlanes(B, cout, H, W, dev)
    hb.conv2d_ups_hl8(pa.view(), c1, pb.view(broadcast=True) if pb else None, c2, pk, dst.view(), hb.view_of(y32), B, H, W)
    got = y32.cpu()
    err = float((got - want).abs().max())
    assert err < 5e-5, "fused upsample conv %d+%d->%d %dx%dx%d: max err %.3e" % (c1, c2, cout, B, h, w, err)
    assert float((dst.to_nchw().cpu() - got).abs().max()) < 1e-5
    full = dst.buf[:B * dst.G * 2 * dst.Hp * dst.Wp * 8].view(B, dst.G, 2, dst.Hp, dst.Wp, 8).float().cpu().clone()
    full[:, :, :, hb.SSM_PADY:hb.SSM_PADY + H, hb.SSM_PADX:hb.SSM_PADX + W] = 0
    assert float(full.abs().max()) == 0.0, "zero frame of the output was written"
