"""GPU parity of the Winograd F(4x4,3x3) fp32 convolution (csrc/ssm_wino4.hip) for EVERY tile configuration - forced one by one
through ssm_wino4_force_kind - against the CPU oracle's direct convolution: plain conv, two-source (torch.cat) input, fused 2x2 mean,
fused concat + bilinear x2 upsample + conv (scripts/models/flow_computation.py:244-247), pre-activation addend, plain NCHW outputs.
Ragged sizes: tiles overshoot the map on both axes, odd sizes, maps smaller than one tile.  Bar 5e-5 like the other kernels (outputs of
magnitude ~1; the emulation of the form predicts ~1e-5: tests/emulate_winograd_f44_precision.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

KINDS = ["X4A", "X4B", "X4C", "Y4A", "Y4B", "Y4C"]      # X4*: 32 couts per workgroup (256 threads); Y4*: 64 couts per workgroup (512 threads)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _unforce():
    yield
    from ssm_amd import hipbind as hb
    hb.load().ssm_wino4_force_kind(-1)


def _force(kind):
    from ssm_amd import hipbind as hb
    n = hb.load().ssm_wino4_force_kind(KINDS.index(kind))
    assert n == len(KINDS), "tile-configuration list of the test is out of date (%d in the library)" % n


def _err(got, want):
    return float((got - want).abs().max())


@pytest.mark.parametrize("kind", KINDS)
def test_every_wino4_configuration_plain_cat_pool(dev, kind):
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(KINDS.index(kind))
    _force(kind)
    for B, H, W, c1, c2, cout in ((2, 22, 44, 16, 8, 64), (1, 23, 40, 32, 0, 64), (3, 6, 2, 8, 8, 32), (1, 46, 80, 24, 8, 160), (1, 12, 20, 4, 0, 32),
                                  (1, 5, 7, 8, 0, 32), (2, 32, 64, 12, 4, 32)):
        if kind[0] == "Y":
            cout = -(-cout // 64) * 64
        a = torch.randn(B, c1, H, W, generator=g)
        b = torch.randn(B, max(c2, 1), H, W, generator=g)
        w = torch.randn(cout, c1 + c2, 3, 3, generator=g) / ((c1 + c2) * 9) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        x = torch.cat([a, b], 1) if c2 else a
        want = O.conv2d_lrelu(x, w, bias)
        assert hb.wino4_plan(c1 + c2, cout, B, H, W)[0] == KINDS.index(kind)
        pa = hb.Planes(B, c1, H, W, dev).load(a.to(dev))
        pb = hb.Planes(B, c2, H, W, dev).load(b.to(dev)) if c2 else None
        pool = H % 2 == 0 and W % 2 == 0
        y, yp = hb.Planes(B, cout, H, W, dev), hb.Planes(B, cout, max(H // 2, 1), max(W // 2, 1), dev)
        pk = hb.PackedWino4(w.to(dev), bias.to(dev), B, H, W, pool=pool)
        hb.conv2d_wino4(pa.view(), c1, pb.view() if c2 else None, c2, pk, y.view(), yp.view() if pool else None, B, H, W, lrelu=True)
        got = y.to_nchw().cpu()
        assert _err(got, want) < 5e-5, "%s %dx%dx%d: conv %.3e" % (kind, B, H, W, _err(got, want))
        if pool:
            gp = yp.to_nchw().cpu()
            assert _err(gp, O.avg_pool2(want)) < 5e-5, "%s: fused pool %.3e" % (kind, _err(gp, O.avg_pool2(want)))
            fullp = yp.full.cpu().clone()
            fullp[:, :, hb.SSM_PADY:hb.SSM_PADY + H // 2, hb.SSM_PADX:hb.SSM_PADX + W // 2] = 0
            assert float(fullp.abs().max()) == 0.0, "%s wrote outside the pooled interior" % kind
        full = y.full.cpu().clone()
        full[:, :, hb.SSM_PADY:hb.SSM_PADY + H, hb.SSM_PADX:hb.SSM_PADX + W] = 0
        assert float(full.abs().max()) == 0.0, "%s wrote outside the interior" % kind
        # no activation (final_conv-style call) into a plain NCHW tensor (element-wise store path for most W)
        yn = torch.full((B, cout, H, W), 7.0, device=dev)
        hb.conv2d_wino4(pa.view(), c1, pb.view() if c2 else None, c2, pk, hb.view_of(yn), None, B, H, W, lrelu=False)
        assert _err(yn.cpu(), O.conv2d(x, w, bias)) < 5e-5, "%s %dx%dx%d: NCHW output" % (kind, B, H, W)


UPS_SHAPES = [(1, 23, 40), (2, 5, 7), (1, 8, 48), (2, 11, 11), (1, 1, 1), (1, 3, 34)]     # LOW-res (B, h, w)


@pytest.mark.parametrize("kind", KINDS)
def test_every_wino4_configuration_fused_upsample(dev, kind):
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(100 + KINDS.index(kind))
    c1, c2, cout = 16, 8, 64
    w = torch.randn(cout, c1 + c2, 3, 3, generator=g) / ((c1 + c2) * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    _force(kind)
    for B, h, wd in UPS_SHAPES:
        H, W = 2 * h, 2 * wd
        a, b = torch.randn(B, c1, h, wd, generator=g), torch.randn(1, c2, h, wd, generator=g)     # b: batch-broadcast
        want = O.conv2d_lrelu(O.upsample2x_bilinear(torch.cat([a, b.expand(B, -1, -1, -1)], 1)), w, bias)
        pa, pb = hb.Planes(B, c1, h, wd, dev).load(a.to(dev)), hb.Planes(1, c2, h, wd, dev).load(b.to(dev))
        y = hb.Planes(B, cout, H, W, dev)
        pk = hb.PackedWino4(w.to(dev), bias.to(dev), B, H, W, ups=True)
        hb.conv2d_ups_wino4(pa.view(), c1, pb.view(broadcast=True), c2, pk, y.view(), B, H, W)
        got = y.to_nchw().cpu()
        assert _err(got, want) < 5e-5, "%s %dx%d: fused upsample conv %.3e" % (kind, h, wd, _err(got, want))
        pk1 = hb.PackedWino4(w[:, :c1].contiguous().to(dev), bias.to(dev), B, H, W, ups=True)
        hb.conv2d_ups_wino4(pa.view(), c1, None, 0, pk1, y.view(), B, H, W, lrelu=False)
        want1 = O.conv2d(O.upsample2x_bilinear(a), w[:, :c1].contiguous(), bias)
        assert _err(y.to_nchw().cpu(), want1) < 5e-5, "%s %dx%d: single-source" % (kind, h, wd)


@pytest.mark.parametrize("kind", KINDS)
def test_wino4_short_channel_counts(dev, kind):
    """1..5 chunks of 4 input channels: the double-buffered forms prefetch two (fused upsample: three) chunks ahead and run their last
    three chunks in a copy of the loop with run-time tail conditions - every tail length, plain and fused-upsample."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(400 + KINDS.index(kind))
    _force(kind)
    B, H, W, cout = 2, 20, 36, 64
    for cin in (4, 8, 12, 16, 20):
        w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        a = torch.randn(B, cin, H, W, generator=g)
        y = hb.Planes(B, cout, H, W, dev)
        pk = hb.PackedWino4(w.to(dev), bias.to(dev), B, H, W)
        hb.conv2d_wino4(hb.Planes(B, cin, H, W, dev).load(a.to(dev)).view(), cin, None, 0, pk, y.view(), None, B, H, W, lrelu=True)
        e = _err(y.to_nchw().cpu(), O.conv2d_lrelu(a, w, bias))
        assert e < 5e-5, "%s cin %d: %.3e" % (kind, cin, e)
        al = torch.randn(B, cin, H // 2, W // 2, generator=g)
        pku = hb.PackedWino4(w.to(dev), bias.to(dev), B, H, W, ups=True)
        hb.conv2d_ups_wino4(hb.Planes(B, cin, H // 2, W // 2, dev).load(al.to(dev)).view(), cin, None, 0, pku, y.view(), B, H, W)
        e = _err(y.to_nchw().cpu(), O.conv2d_lrelu(O.upsample2x_bilinear(al), w, bias))
        assert e < 5e-5, "%s cin %d fused upsample: %.3e" % (kind, cin, e)


def test_wino4_deep_channels_and_scale_invariance(dev):
    """512 input channels (128 chunks through the filter double buffer) at the 1/16 map, and the same problem with activations x 2^12
    and filters x 2^-9: the form is linear fp32 arithmetic - no operand range in which it degrades (unlike the split-fp16 modes)."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(5)
    B, H, W, cin, cout = 2, 46, 80, 512, 128
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    for sx, sw in ((1.0, 1.0), (4096.0, 1.0 / 512)):
        want = O.conv2d_lrelu(x * sx, w * sw, bias * sx * sw)
        px = hb.Planes(B, cin, H, W, dev).load((x * sx).to(dev))
        y = hb.Planes(B, cout, H, W, dev)
        pk = hb.PackedWino4((w * sw).to(dev), (bias * sx * sw).to(dev), B, H, W)
        hb.conv2d_wino4(px.view(), cin, None, 0, pk, y.view(), None, B, H, W)
        e = _err(y.to_nchw().cpu(), want)
        print("wino4 512 channels, scale %g x %g: max err %.3e" % (sx, sw, e))
        assert e < 5e-5 * sx * sw, "scale %g x %g: %.3e" % (sx, sw, e)


@pytest.mark.parametrize("kind", KINDS)
def test_wino4_pre_activation_addend(dev, kind):
    """y = act(conv(x) + bias + add[b // add_div]) (ssm_wino4_conv2d_add_fwd / ssm_wino4_conv2d_ups_add_fwd): the form stage 2 uses for
    the t-independent half of conv7a's input (one addend entry per pair serves the G interpolation times of that pair)."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(300 + KINDS.index(kind))
    _force(kind)
    B, div, cin, cout = 6, 3, 16, 64
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    for ups, (h, wd) in ((False, (22, 44)), (True, (11, 23)), (False, (7, 9))):
        H, W = (2 * h, 2 * wd) if ups else (h, wd)
        x = torch.randn(B, cin, h, wd, generator=g)
        add = torch.randn(B // div, cout, H, W, generator=g)
        xin = O.upsample2x_bilinear(x) if ups else x
        z = O.conv2d(xin, w, bias) + add.repeat_interleave(div, 0)
        want = torch.where(z >= 0, z, z * 0.1)
        px = hb.Planes(B, cin, h, wd, dev).load(x.to(dev))
        pa = hb.Planes(B // div, cout, H, W, dev).load(add.to(dev))
        y = hb.Planes(B, cout, H, W, dev)
        pk = hb.PackedWino4(w.to(dev), bias.to(dev), B, H, W, ups=ups)
        if ups:
            hb.conv2d_ups_wino4(px.view(), cin, None, 0, pk, y.view(), B, H, W, add=pa.view(), add_div=div)
        else:
            hb.conv2d_wino4(px.view(), cin, None, 0, pk, y.view(), None, B, H, W, add=pa.view(), add_div=div)
        assert _err(y.to_nchw().cpu(), want) < 5e-5, "%s ups=%d %dx%d: %.3e" % (kind, ups, H, W, _err(y.to_nchw().cpu(), want))


def test_wino4_subpixel_interior_plus_border_ring(dev):
    """conv3x3(upsample2x(cat[a, b])) (scripts/models/flow_computation.py:244-247) as the sub-pixel F(4x4,3x3) form on the map's interior
    (ssm_wino4_conv2d_shuffle_fwd: effective filters M_pa W M_pb^T, 4 Cout channels, pixel-shuffle store) + the fused-upsample kernel on the
    border ring of 16 x 32-pixel tiles (ssm_wino4_conv2d_ups_border_fwd): every output pixel written exactly once, 5e-5 from the oracle's
    upsample-then-convolve - two sources, one source, the smallest supported map, several batch entries, a 64-channel case."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(77)
    for B, h, w, c1, c2, cout in ((2, 32, 64, 8, 8, 32), (1, 48, 96, 16, 0, 32), (3, 32, 96, 4, 12, 64), (1, 64, 64, 32, 32, 32)):
        H, W = 2 * h, 2 * w
        assert hb.subpixel_wino4_supported(c1 + c2, cout, H, W)
        a = torch.randn(B, c1, h, w, generator=g)
        b = torch.randn(B, max(c2, 1), h, w, generator=g)
        x = torch.cat([a, b], 1) if c2 else a
        wt = torch.randn(cout, c1 + c2, 3, 3, generator=g) / ((c1 + c2) * 9) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        want = O.conv2d_lrelu(O.upsample2x_bilinear(x), wt, bias)
        pk = hb.PackedSubpixelWino4(wt.to(dev), bias.to(dev), B, H, W)
        pa = hb.Planes(B, c1, h, w, dev).load(a.to(dev))
        pb = hb.Planes(B, c2, h, w, dev).load(b.to(dev)) if c2 else None
        y = hb.Planes(B, cout, H, W, dev)
        y.interior.fill_(float("nan"))                       # every output pixel must be written by one of the two launches
        hb.conv2d_ups_subpixel_wino4(lambda y0, x0: pa.view(y0=y0, x0=x0), c1, (lambda y0, x0: pb.view(y0=y0, x0=x0)) if c2 else None, c2, pk,
                                     lambda y0, x0: y.view(y0=y0, x0=x0), B, H, W)
        got = y.to_nchw().cpu()
        assert bool(torch.isfinite(got).all()), "%dx%d: pixels left unwritten" % (H, W)
        e = _err(got, want)
        ring = got.clone()
        ring[:, :, 16:H - 16, 32:W - 32] = want[:, :, 16:H - 16, 32:W - 32]
        assert e < 5e-5, "%dx%dx%d c %d+%d -> %d: %.3e (border ring alone %.3e)" % (B, H, W, c1, c2, cout, e, _err(ring, want))
        full = y.full.cpu().clone()
        full[:, :, hb.SSM_PADY:hb.SSM_PADY + H, hb.SSM_PADX:hb.SSM_PADX + W] = 0
        assert float(full.abs().max()) == 0.0, "wrote outside the interior"
    assert not hb.subpixel_wino4_supported(256, 64, 368, 640) and hb.subpixel_wino4_supported(128, 32, 736, 1280)
