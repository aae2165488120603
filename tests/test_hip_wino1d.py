"""GPU parity of the 1-D Winograd convolutions (csrc/ssm_wino1d.hip: 7x7 as F(2,7), 5x5 as F(4,5) along x, direct along y, fp32
throughout) for EVERY tile configuration - forced one by one through ssm_wino1d_force_kind - against the CPU oracle's direct
convolution (layers.conv of the reference, scripts/models/layers.py:21-33): plain, fused 2x2 mean, pre-activation addend, channel
counts off the chunk size, plain NCHW outputs (element-wise store path), ragged sizes (tiles overshoot the map on both axes, odd
sizes, maps smaller than one tile).  Bar 5e-5 like the direct kernel (outputs of magnitude ~1; the emulation of the form in
tests/emulate_winograd_1d_precision.py predicts 5-8e-6)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

KINDS = ["R7A", "R7B", "R5A", "R5B", "R5C"]
KSIZE = {"R7A": 7, "R7B": 7, "R5A": 5, "R5B": 5, "R5C": 5}
BN = {"R7A": 32, "R7B": 32, "R5A": 64, "R5B": 64, "R5C": 32}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _unforce():
    yield
    from ssm_amd import hipbind as hb
    hb.load().ssm_wino1d_force_kind(-1)


def _force(kind):
    from ssm_amd import hipbind as hb
    n = hb.load().ssm_wino1d_force_kind(KINDS.index(kind))
    assert n == len(KINDS), "tile-configuration list of the test is out of date (%d in the library)" % n


def _err(got, want):
    return float((got - want).abs().max())


@pytest.mark.parametrize("kind", KINDS)
def test_every_wino1d_configuration_plain_pool(dev, kind):
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(KINDS.index(kind))
    _force(kind)
    k, bn = KSIZE[kind], BN[kind]
    # (B, H, W, cin, cout): whole tiles; ragged both ways + odd channel count (padded to the chunk); smaller than a tile; odd sizes
    for B, H, W, cin, cout in ((2, 16, 64, 6, bn), (1, 23, 40, 7, 2 * bn), (3, 6, 2, 2, bn), (1, 9, 131, 16, bn), (2, 34, 96, 32, bn), (1, 3, 5, 4, bn)):
        x = torch.randn(B, cin, H, W, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        want = O.conv2d_lrelu(x, w, bias)
        assert hb.wino1d_plan(k, cin, cout, B, H, W)[0] == KINDS.index(kind)
        pool = H % 2 == 0 and W % 2 == 0
        pk = hb.PackedWino1d(w.to(dev), bias.to(dev), B, H, W, pool=pool)
        px = hb.Planes(B, pk.cin_p, H, W, dev)
        px.interior[:, :cin] = x.to(dev)
        y, yp = hb.Planes(B, cout, H, W, dev), hb.Planes(B, cout, max(H // 2, 1), max(W // 2, 1), dev)
        hb.conv2d_wino1d(px.view(), pk.cin_p, None, 0, pk, y.view(), yp.view() if pool else None, B, H, W, lrelu=True)
        got = y.to_nchw().cpu()
        assert _err(got, want) < 5e-5, "%s %dx%dx%d cin %d: conv %.3e" % (kind, B, H, W, cin, _err(got, want))
        if pool:
            gp = yp.to_nchw().cpu()
            assert _err(gp, O.avg_pool2(want)) < 5e-5, "%s: fused pool %.3e" % (kind, _err(gp, O.avg_pool2(want)))
            fullp = yp.full.cpu().clone()
            fullp[:, :, hb.SSM_PADY:hb.SSM_PADY + H // 2, hb.SSM_PADX:hb.SSM_PADX + W // 2] = 0
            assert float(fullp.abs().max()) == 0.0, "%s wrote outside the pooled interior" % kind
        full = y.full.cpu().clone()
        full[:, :, hb.SSM_PADY:hb.SSM_PADY + H, hb.SSM_PADX:hb.SSM_PADX + W] = 0
        assert float(full.abs().max()) == 0.0, "%s wrote outside the interior" % kind
        # no activation, plain NCHW output tensor (rows not 16-byte aligned for most W: the element-wise store path)
        yn = torch.full((B, cout, H, W), 7.0, device=dev)
        hb.conv2d_wino1d(px.view(), pk.cin_p, None, 0, pk, hb.view_of(yn), None, B, H, W, lrelu=False)
        assert _err(yn.cpu(), O.conv2d(x, w, bias)) < 5e-5, "%s %dx%dx%d: NCHW output" % (kind, B, H, W)


@pytest.mark.parametrize("kind", KINDS)
def test_wino1d_pre_activation_addend(dev, kind):
    """y = act(conv(x) + bias + add[b // add_div]) (ssm_wino1d_conv2d_add_fwd): the form stage 2's conv1a uses for the frame channels
    of its input, which are the same for the G interpolation times of a pair (flow_interpolation.py:364-367)."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(300 + KINDS.index(kind))
    _force(kind)
    k, bn = KSIZE[kind], BN[kind]
    B, div, cin, cout = 6, 3, 10, bn
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    for H, W in ((22, 44), (7, 37)):
        x = torch.randn(B, cin, H, W, generator=g)
        add = torch.randn(B // div, cout, H, W, generator=g)
        z = O.conv2d(x, w, bias) + add.repeat_interleave(div, 0)
        want = torch.where(z >= 0, z, z * 0.1)
        # the input as a channel window of a wider tensor (stage 2 reads channels 3:13 of its 16-channel input)
        p16 = hb.Planes(B, 16, H, W, dev)
        p16.interior[:, 3:3 + cin] = x.to(dev)
        pa = hb.Planes(B // div, cout, H, W, dev).load(add.to(dev))
        y = hb.Planes(B, cout, H, W, dev)
        pk = hb.PackedWino1d(w.to(dev), bias.to(dev), B, H, W)
        hb.conv2d_wino1d(p16.view(c0=3), cin, None, 0, pk, y.view(), None, B, H, W, add=pa.view(), add_div=div)
        assert _err(y.to_nchw().cpu(), want) < 5e-5, "%s %dx%d: %.3e" % (kind, H, W, _err(y.to_nchw().cpu(), want))


@pytest.mark.parametrize("k,cin,cout,H,W", [(7, 32, 32, 64, 160), (5, 64, 64, 48, 96)])
def test_wino1d_layer_shapes_and_scale_invariance(dev, k, cin, cout, H, W):
    """The layers the plan uses the form for (conv1b: 32 -> 32 7x7, conv2b: 64 -> 64 5x5), many chunks through the double buffer, and
    the same problem with activations x 2^12 and filters x 2^-9: the form is linear fp32 arithmetic, no operand range degrades it."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(5 + k)
    B = 2
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    for sx, sw in ((1.0, 1.0), (4096.0, 1.0 / 512)):
        want = O.conv2d_lrelu(x * sx, w * sw, bias * sx * sw)
        px = hb.Planes(B, cin, H, W, dev).load((x * sx).to(dev))
        y, yp = hb.Planes(B, cout, H, W, dev), hb.Planes(B, cout, H // 2, W // 2, dev)
        pk = hb.PackedWino1d((w * sw).to(dev), (bias * sx * sw).to(dev), B, H, W, pool=True)
        hb.conv2d_wino1d(px.view(), cin, None, 0, pk, y.view(), yp.view(), B, H, W)
        e = _err(y.to_nchw().cpu(), want)
        print("wino1d k=%d scale %g x %g: max err %.3e" % (k, sx, sw, e))
        assert e < 5e-5 * sx * sw, "scale %g x %g: %.3e" % (sx, sw, e)
        assert _err(yp.to_nchw().cpu(), O.avg_pool2(want)) < 5e-5 * sx * sw
