"""GPU parity of the blocked two-dimensional Winograd form of the 7x7 layers (csrc/ssm_wino7.hip: 2x2 blocks of F(4x4,4x4), fp32
throughout) for EVERY tile configuration - forced one by one through ssm_wino7_force_kind - against the CPU oracle's direct
convolution (layers.conv, scripts/models/layers.py:21-33): plain conv, fused 2x2 mean, pre-activation addend, plain NCHW outputs, every
channel count 1..9 (prologue / tail of the three-deep pipeline), ragged sizes (tiles overshoot the map on both axes, odd sizes, maps
smaller than one tile), the layer shapes of the plan and operand-scale invariance.
Bar 5e-5 like the other kernels at unit output scale (the emulation of the form, tests/emulate_winograd_7x7_blocked.py, predicts
2e-6 rms / 3e-5 max for 32 channels)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# 16x32-pixel workgroup tiles: tile groups of 8x2 tiles stacked / of 4x4 tiles side by side; ..S: the frequency-split kernel of eight waves (r5)
KINDS = ["Z7A", "Z7B", "Z7AS", "Z7BS"]
BAR = 5e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _unforce():
    yield
    from ssm_amd import hipbind as hb
    hb.load().ssm_wino7_force_kind(-1)


def _force(kind):
    from ssm_amd import hipbind as hb
    n = hb.load().ssm_wino7_force_kind(KINDS.index(kind))
    assert n == len(KINDS), "tile-configuration list of the test is out of date (%d in the library)" % n


def _err(got, want):
    return float((got - want).abs().max())


@pytest.mark.parametrize("kind", KINDS)
def test_every_wino7_configuration_plain_pool(dev, kind):
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(KINDS.index(kind))
    _force(kind)
    # (B, H, W, cin, cout): whole tiles; ragged both ways; smaller than a tile; odd sizes; two cout blocks; a wide ragged map
    for B, H, W, cin, cout in ((2, 16, 64, 6, 32), (1, 23, 40, 7, 64), (3, 6, 2, 2, 32), (1, 9, 131, 16, 32), (2, 34, 96, 32, 32), (1, 3, 5, 4, 32),
                               (1, 1, 1, 1, 32)):
        x = torch.randn(B, cin, H, W, generator=g)
        w = torch.randn(cout, cin, 7, 7, generator=g) / (cin * 49) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        want = O.conv2d_lrelu(x, w, bias)
        pool = H % 2 == 0 and W % 2 == 0
        pk = hb.PackedWino7(w.to(dev), bias.to(dev), B, H, W, pool=pool)
        px = hb.Planes(B, cin, H, W, dev).load(x.to(dev))
        y, yp = hb.Planes(B, cout, H, W, dev), hb.Planes(B, cout, max(H // 2, 1), max(W // 2, 1), dev)
        hb.conv2d_wino7(px.view(), cin, None, 0, pk, y.view(), yp.view() if pool else None, B, H, W, lrelu=True)
        got = y.to_nchw().cpu()
        assert _err(got, want) < BAR, "%s %dx%dx%d cin %d: conv %.3e" % (kind, B, H, W, cin, _err(got, want))
        if pool:
            gp = yp.to_nchw().cpu()
            assert _err(gp, O.avg_pool2(want)) < BAR, "%s: fused pool %.3e" % (kind, _err(gp, O.avg_pool2(want)))
            fullp = yp.full.cpu().clone()
            fullp[:, :, hb.SSM_PADY:hb.SSM_PADY + H // 2, hb.SSM_PADX:hb.SSM_PADX + W // 2] = 0
            assert float(fullp.abs().max()) == 0.0, "%s wrote outside the pooled interior" % kind
        full = y.full.cpu().clone()
        full[:, :, hb.SSM_PADY:hb.SSM_PADY + H, hb.SSM_PADX:hb.SSM_PADX + W] = 0
        assert float(full.abs().max()) == 0.0, "%s wrote outside the interior" % kind
        # no activation, plain NCHW output tensor (rows not 16-byte aligned for most W: the element-wise store path)
        yn = torch.full((B, cout, H, W), 7.0, device=dev)
        hb.conv2d_wino7(px.view(), cin, None, 0, pk, hb.view_of(yn), None, B, H, W, lrelu=False)
        assert _err(yn.cpu(), O.conv2d(x, w, bias)) < BAR, "%s %dx%dx%d: NCHW output" % (kind, B, H, W)


@pytest.mark.parametrize("kind", KINDS)
def test_wino7_every_short_channel_count(dev, kind):
    """1 .. 9 input channels: the pipeline runs the row pass two channels and the column pass one channel ahead of the MFMAs, its first
    two and last three iterations carry run-time conditions - every prologue / tail length, on a map of several workgroup tiles."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(50 + KINDS.index(kind))
    _force(kind)
    B, H, W, cout = 2, 40, 72, 32
    for cin in range(1, 10):
        x = torch.randn(B, cin, H, W, generator=g)
        w = torch.randn(cout, cin, 7, 7, generator=g) / (cin * 49) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        pk = hb.PackedWino7(w.to(dev), bias.to(dev), B, H, W)
        px = hb.Planes(B, cin, H, W, dev).load(x.to(dev))
        y = hb.Planes(B, cout, H, W, dev)
        hb.conv2d_wino7(px.view(), cin, None, 0, pk, y.view(), None, B, H, W)
        e = _err(y.to_nchw().cpu(), O.conv2d_lrelu(x, w, bias))
        assert e < BAR, "%s cin %d: %.3e" % (kind, cin, e)


@pytest.mark.parametrize("kind", KINDS)
def test_wino7_pre_activation_addend(dev, kind):
    """y = act(conv(x) + bias + add[b // add_div]) (ssm_wino7_conv2d_add_fwd): the form stage 2's conv1a uses for the frame channels
    of its input, which are the same for the G interpolation times of a pair (flow_interpolation.py:364-367)."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(300 + KINDS.index(kind))
    _force(kind)
    B, div, cin, cout = 6, 3, 10, 32
    w = torch.randn(cout, cin, 7, 7, generator=g) / (cin * 49) ** 0.5
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
        pk = hb.PackedWino7(w.to(dev), bias.to(dev), B, H, W)
        hb.conv2d_wino7(p16.view(c0=3), cin, None, 0, pk, y.view(), None, B, H, W, add=pa.view(), add_div=div)
        assert _err(y.to_nchw().cpu(), want) < BAR, "%s %dx%d: %.3e" % (kind, H, W, _err(y.to_nchw().cpu(), want))


def test_wino7_addend_with_fused_pool(dev):
    """Pre-activation addend AND fused 2x2 mean in one launch, on a map whose workgroups take the epilogue's three paths: whole tiles
    (outputs held in registers until the last addend row is used, then stored and pooled), a ragged right / bottom edge (element-wise
    path), and add_div = 1.  The plan never asks for the combination; the entry point allows it."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(77)
    for (B, div, H, W) in ((4, 2, 48, 96), (2, 1, 38, 76)):
        cin, cout = 5, 32
        w = torch.randn(cout, cin, 7, 7, generator=g) / (cin * 49) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        x = torch.randn(B, cin, H, W, generator=g)
        add = torch.randn(B // div, cout, H, W, generator=g)
        z = O.conv2d(x, w, bias) + add.repeat_interleave(div, 0)
        want = torch.where(z >= 0, z, z * 0.1)
        px = hb.Planes(B, cin, H, W, dev).load(x.to(dev))
        pa = hb.Planes(B // div, cout, H, W, dev).load(add.to(dev))
        y, yp = hb.Planes(B, cout, H, W, dev), hb.Planes(B, cout, H // 2, W // 2, dev)
        pk = hb.PackedWino7(w.to(dev), bias.to(dev), B, H, W, pool=True)
        hb.conv2d_wino7(px.view(), cin, None, 0, pk, y.view(), yp.view(), B, H, W, add=pa.view(), add_div=div)
        assert _err(y.to_nchw().cpu(), want) < BAR, "%dx%d: %.3e" % (H, W, _err(y.to_nchw().cpu(), want))
        assert _err(yp.to_nchw().cpu(), O.avg_pool2(want)) < BAR, "%dx%d: pooled" % (H, W)


def test_wino7_layer_shape_and_scale_invariance(dev):
    """conv1b as the plan runs it (32 -> 32, fused 2x2 mean) on a map of many workgroup tiles, and the same problem with activations
    x 2^12 and filters x 2^-9: the form is linear fp32 arithmetic with dyadic transform constants, no operand range degrades it."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(12)
    B, cin, cout, H, W = 2, 32, 32, 64, 160
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 7, 7, generator=g) / (cin * 49) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    for sx, sw in ((1.0, 1.0), (4096.0, 1.0 / 512)):
        want = O.conv2d_lrelu(x * sx, w * sw, bias * sx * sw)
        px = hb.Planes(B, cin, H, W, dev).load((x * sx).to(dev))
        y, yp = hb.Planes(B, cout, H, W, dev), hb.Planes(B, cout, H // 2, W // 2, dev)
        pk = hb.PackedWino7((w * sw).to(dev), (bias * sx * sw).to(dev), B, H, W, pool=True)
        hb.conv2d_wino7(px.view(), cin, None, 0, pk, y.view(), yp.view(), B, H, W)
        e = _err(y.to_nchw().cpu(), want)
        print("wino7 scale %g x %g: max err %.3e" % (sx, sw, e))
        assert e < BAR * sx * sw, "scale %g x %g: %.3e" % (sx, sw, e)
        assert _err(yp.to_nchw().cpu(), O.avg_pool2(want)) < BAR * sx * sw


def test_wino7_batch_pack_is_bit_identical(dev):
    """The one-launch repack (ssm_pack32_weights_batch, algo SSM_PACK_WINO7) fills the same bytes as ssm_wino7_pack_weights."""
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(3)
    w = torch.randn(32, 10, 7, 7, generator=g).to(dev)
    b = torch.randn(32, generator=g).to(dev)
    pk = hb.PackedWino7(w, b, 1, 32, 32)
    ref_w, ref_b = pk.w.clone(), pk.b.clone()
    pk.w.fill_(-1.0)
    pk.b.fill_(-1.0)
    hb.PackBatch32([(pk, w, b, False)], dev).run()
    torch.cuda.synchronize()
    assert torch.equal(pk.w, ref_w) and torch.equal(pk.b, ref_b)
