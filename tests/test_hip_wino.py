"""GPU parity of the Winograd F(2x2,3x3) fp32 convolution (csrc/ssm_wino.hip) for EVERY tile configuration - forced one by one
through ssm_wino_force_kind - against the CPU oracle's direct convolution: plain conv, two-source (torch.cat) input, fused 2x2 mean,
fused concat + bilinear x2 upsample + conv (scripts/models/flow_computation.py:244-247).  Ragged sizes: tiles overshoot the map on
both axes, odd heights, maps smaller than one tile.  Bar 5e-5 like the direct kernel (outputs of magnitude ~1; measured ~1e-6)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

KINDS = ["W64A", "W32A", "W128A", "W64G", "W128G", "W64H", "W128H", "V32A", "V32H", "V32G", "V64A", "V64G"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _unforce():
    yield
    from ssm_amd import hipbind as hb
    hb.load().ssm_wino_force_kind(-1)


def _force(kind):
    from ssm_amd import hipbind as hb
    n = hb.load().ssm_wino_force_kind(KINDS.index(kind))
    assert n == len(KINDS), "tile-configuration list of the test is out of date (%d in the library)" % n


def _err(got, want):
    return float((got - want).abs().max())


@pytest.mark.parametrize("kind", KINDS)
def test_every_wino_configuration_plain_cat_pool(dev, kind):
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(KINDS.index(kind))
    _force(kind)
    for B, H, W, c1, c2, cout in ((2, 22, 44, 16, 8, 40), (1, 23, 40, 32, 0, 64), (3, 6, 2, 8, 8, 32), (1, 46, 80, 24, 8, 136), (1, 12, 20, 8, 0, 32)):     # the last: ONE chunk
        a = torch.randn(B, c1, H, W, generator=g)
        b = torch.randn(B, max(c2, 1), H, W, generator=g)
        w = torch.randn(cout, c1 + c2, 3, 3, generator=g) / ((c1 + c2) * 9) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        x = torch.cat([a, b], 1) if c2 else a
        want = O.conv2d_lrelu(x, w, bias)
        assert hb.wino_plan(c1 + c2, cout, B, H, W)[0] == KINDS.index(kind)
        pa = hb.Planes(B, c1, H, W, dev).load(a.to(dev))
        pb = hb.Planes(B, c2, H, W, dev).load(b.to(dev)) if c2 else None
        pool = H % 2 == 0
        y, yp = hb.Planes(B, cout, H, W, dev), hb.Planes(B, cout, H // 2, W // 2, dev)
        pk = hb.PackedWino(w.to(dev), bias.to(dev), B, H, W, pool=pool)
        hb.conv2d_wino(pa.view(), c1, pb.view() if c2 else None, c2, pk, y.view(), yp.view() if pool else None, B, H, W, lrelu=True)
        got = y.to_nchw().cpu()
        assert _err(got, want) < 5e-5, "%s %dx%dx%d: conv %.3e" % (kind, B, H, W, _err(got, want))
        if pool:
            gp = yp.to_nchw().cpu()
            assert _err(gp, O.avg_pool2(want)) < 5e-5, "%s: fused pool %.3e" % (kind, _err(gp, O.avg_pool2(want)))
        full = y.full.cpu().clone()
        full[:, :, hb.SSM_PADY:hb.SSM_PADY + H, hb.SSM_PADX:hb.SSM_PADX + W] = 0
        assert float(full.abs().max()) == 0.0, "%s wrote outside the interior" % kind
        # no activation (final_conv-style call)
        hb.conv2d_wino(pa.view(), c1, pb.view() if c2 else None, c2, pk, y.view(), None, B, H, W, lrelu=False)
        assert _err(y.to_nchw().cpu(), O.conv2d(x, w, bias)) < 5e-5


@pytest.mark.parametrize("kind", KINDS)
def test_every_wino_configuration_odd_widths(dev, kind):
    """r6: maps of odd WIDTH (config 3's 11x11 bottleneck maps; any crop that is no multiple of 64) - the last 2x2 output tile of a row
    hangs one column over the map, and that column is the zero frame the next layer reads as padding: it must never be written.  Every
    tile configuration, plain / two sources / pre-activation addend / LeakyReLU' mask, against the oracle; the frame stays zero."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(100 + KINDS.index(kind))
    _force(kind)
    for B, H, W, c1, c2, cout in ((2, 11, 11, 16, 8, 40), (1, 22, 43, 32, 0, 64), (3, 5, 1, 8, 8, 32), (1, 9, 67, 24, 8, 72)):
        a = torch.randn(B, c1, H, W, generator=g)
        b = torch.randn(B, max(c2, 1), H, W, generator=g)
        w = torch.randn(cout, c1 + c2, 3, 3, generator=g) / ((c1 + c2) * 9) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        add = torch.randn(B, cout, H, W, generator=g)
        x = torch.cat([a, b], 1) if c2 else a
        z = O.conv2d(x, w, bias)
        pa = hb.Planes(B, c1, H, W, dev).load(a.to(dev))
        pb = hb.Planes(B, c2, H, W, dev).load(b.to(dev)) if c2 else None
        pz = hb.Planes(B, cout, H, W, dev).load(add.to(dev))
        pk = hb.PackedWino(w.to(dev), bias.to(dev), B, H, W)
        for what, kw, want in (("plain", {}, torch.where(z >= 0, z, z * 0.1)),
                               ("addend", {"add": pz.view()}, torch.where(z + add >= 0, z + add, (z + add) * 0.1)),
                               ("mask", {"add": pz.view(), "mask": True, "lrelu": False}, z * torch.where(add > 0, torch.ones_like(add), torch.full_like(add, 0.1)))):
            y = hb.Planes(B, cout, H, W, dev)
            hb.conv2d_wino(pa.view(), c1, pb.view() if c2 else None, c2, pk, y.view(), None, B, H, W, **kw)
            got = y.to_nchw().cpu()
            assert _err(got, want) < 5e-5, "%s %dx%dx%d %s: %.3e" % (kind, B, H, W, what, _err(got, want))
            full = y.full.cpu().clone()
            full[:, :, hb.SSM_PADY:hb.SSM_PADY + H, hb.SSM_PADX:hb.SSM_PADX + W] = 0
            assert float(full.abs().max()) == 0.0, "%s (%s) wrote outside the interior of a %dx%d map" % (kind, what, H, W)


UPS_SHAPES = [(1, 23, 40), (2, 5, 7), (1, 8, 48), (2, 11, 11), (1, 1, 1), (1, 3, 34)]     # LOW-res (B, h, w)


@pytest.mark.parametrize("kind", KINDS)
def test_every_wino_configuration_fused_upsample(dev, kind):
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(100 + KINDS.index(kind))
    c1, c2, cout = 16, 8, 40
    w = torch.randn(cout, c1 + c2, 3, 3, generator=g) / ((c1 + c2) * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    _force(kind)
    for B, h, wd in UPS_SHAPES:
        H, W = 2 * h, 2 * wd
        a, b = torch.randn(B, c1, h, wd, generator=g), torch.randn(1, c2, h, wd, generator=g)     # b: batch-broadcast
        want = O.conv2d_lrelu(O.upsample2x_bilinear(torch.cat([a, b.expand(B, -1, -1, -1)], 1)), w, bias)
        pa, pb = hb.Planes(B, c1, h, wd, dev).load(a.to(dev)), hb.Planes(1, c2, h, wd, dev).load(b.to(dev))
        y = hb.Planes(B, cout, H, W, dev)
        pk = hb.PackedWino(w.to(dev), bias.to(dev), B, H, W, ups=True)
        hb.conv2d_ups_wino(pa.view(), c1, pb.view(broadcast=True), c2, pk, y.view(), B, H, W)
        got = y.to_nchw().cpu()
        assert _err(got, want) < 5e-5, "%s %dx%d: fused upsample conv %.3e" % (kind, h, wd, _err(got, want))
        pk1 = hb.PackedWino(w[:, :c1].contiguous().to(dev), bias.to(dev), B, H, W, ups=True)
        hb.conv2d_ups_wino(pa.view(), c1, None, 0, pk1, y.view(), B, H, W, lrelu=False)
        want1 = O.conv2d(O.upsample2x_bilinear(a), w[:, :c1].contiguous(), bias)
        assert _err(y.to_nchw().cpu(), want1) < 5e-5, "%s %dx%d: single-source" % (kind, h, wd)


def test_wino_deep_channels_and_scale_invariance(dev):
    """512 input channels (64 chunks through the double buffer) at the 1/16 map, and the same problem with activations x 2^12 and
    filters x 2^-9: Winograd in fp32 is linear arithmetic - no operand range in which it degrades (unlike the split-fp16 modes)."""
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
        pk = hb.PackedWino((w * sw).to(dev), (bias * sx * sw).to(dev), B, H, W)
        hb.conv2d_wino(px.view(), cin, None, 0, pk, y.view(), None, B, H, W)
        assert _err(y.to_nchw().cpu(), want) < 5e-5 * sx * sw, "scale %g x %g: %.3e" % (sx, sw, _err(y.to_nchw().cpu(), want))


@pytest.mark.parametrize("kind", ["W64A", "V32H", "V64G"])
def test_wino_pre_activation_addend(dev, kind):
    """y = act(conv(x) + bias + add[b // add_div]) (ssm_wino_conv2d_add_fwd / ssm_wino_conv2d_ups_add_fwd): the form stage 2 uses for the
    t-independent half of conv7a's input (one addend entry per pair serves the G interpolation times of that pair)."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(300 + KINDS.index(kind))
    _force(kind)
    B, div, cin, cout = 6, 3, 16, 64
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    for ups, (h, wd) in ((False, (22, 44)), (True, (11, 23))):
        H, W = (2 * h, 2 * wd) if ups else (h, wd)
        x = torch.randn(B, cin, h, wd, generator=g)
        add = torch.randn(B // div, cout, H, W, generator=g)
        xin = O.upsample2x_bilinear(x) if ups else x
        z = O.conv2d(xin, w, bias) + add.repeat_interleave(div, 0)
        want = torch.where(z >= 0, z, z * 0.1)
        px = hb.Planes(B, cin, h, wd, dev).load(x.to(dev))
        pa = hb.Planes(B // div, cout, H, W, dev).load(add.to(dev))
        y = hb.Planes(B, cout, H, W, dev)
        pk = hb.PackedWino(w.to(dev), bias.to(dev), B, H, W, ups=ups)
        if ups:
            hb.conv2d_ups_wino(px.view(), cin, None, 0, pk, y.view(), B, H, W, add=pa.view(), add_div=div)
        else:
            hb.conv2d_wino(px.view(), cin, None, 0, pk, y.view(), None, B, H, W, add=pa.view(), add_div=div)
        assert _err(y.to_nchw().cpu(), want) < 5e-5, "%s ups=%d: %.3e" % (kind, ups, _err(y.to_nchw().cpu(), want))


def test_wino_split_k_bottleneck_layers(dev):
    """Launches that leave most of the chip idle run split over the input channels (ssm_wino_conv2d_splitk_fwd + ssm_splitk_finish_fwd,
    chosen by ssm_wino_splitk_plan inside hb.conv2d_wino / conv2d_ups_wino): config 3's bottleneck shapes - plain, two sources, fused
    2x2 mean, pre-activation addend, fused upsample, an odd height - against the oracle, and bit-identical from run to run (the partial
    sums are added in a fixed order)."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(77)
    seen = set()
    #            B  cin (c1)      cout  H   W   ups    pool   add
    for B, cin, c1, cout, H, W, ups, pool, addend in ((2, 512, 512, 512, 22, 22, False, True, False),
                                                     (2, 256, 128, 512, 22, 22, False, False, True),
                                                     (2, 1024, 512, 512, 22, 22, True, False, False),
                                                     (1, 512, 512, 64, 21, 26, False, False, False),
                                                     (2, 256, 256, 256, 44, 44, False, True, False),
                                                     (2, 512, 512, 512, 11, 11, False, False, False),          # r6: odd width (config 3's 11x11 maps)
                                                     (2, 512, 256, 512, 11, 11, False, False, True)):
        c2 = cin - c1
        h, wd = (H // 2, W // 2) if ups else (H, W)
        xa, xb = torch.randn(B, c1, h, wd, generator=g), (torch.randn(B, c2, h, wd, generator=g) if c2 else None)
        w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        add = torch.randn(B, cout, H, W, generator=g) if addend else None
        xin = torch.cat([xa, xb], 1) if c2 else xa
        z = O.conv2d(O.upsample2x_bilinear(xin) if ups else xin, w, bias)
        if add is not None:
            z = z + add
        want = torch.where(z >= 0, z, z * 0.1)
        pk = hb.PackedWino(w.to(dev), bias.to(dev), B, H, W, ups=ups)
        ks = hb.wino_splitk(pk, B, H, W, ups)
        seen.add(ks)
        assert ks > 1, "%d -> %d at %dx%d, batch %d: the plan does not split (KS = %d)" % (cin, cout, H, W, B, ks)
        pa, pb = hb.Planes(B, c1, h, wd, dev).load(xa.to(dev)), (hb.Planes(B, c2, h, wd, dev).load(xb.to(dev)) if c2 else None)
        pz = hb.Planes(B, cout, H, W, dev).load(add.to(dev)) if add is not None else None
        outs = []
        for _ in range(2):
            y, yp = hb.Planes(B, cout, H, W, dev), hb.Planes(B, cout, max(H // 2, 1), max(W // 2, 1), dev)
            if ups:
                hb.conv2d_ups_wino(pa.view(), c1, pb.view() if pb is not None else None, c2, pk, y.view(), B, H, W,
                                   add=pz.view() if pz is not None else None)
            else:
                hb.conv2d_wino(pa.view(), c1, pb.view() if pb is not None else None, c2, pk, y.view(), yp.view() if pool else None, B, H, W,
                               add=pz.view() if pz is not None else None)
            outs.append((y.to_nchw().cpu(), yp.to_nchw().cpu()))
        assert _err(outs[0][0], want) < 5e-5, "KS %d, %d -> %d at %dx%d ups=%d: %.3e" % (ks, cin, cout, H, W, ups, _err(outs[0][0], want))
        if pool:
            assert _err(outs[0][1], O.avg_pool2(want)) < 5e-5
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "split-K result differs between two runs"
        full = y.full.cpu().clone()
        full[:, :, hb.SSM_PADY:hb.SSM_PADY + H, hb.SSM_PADX:hb.SSM_PADX + W] = 0
        assert float(full.abs().max()) == 0.0, "split-K wrote outside the interior"
    assert len(seen) > 1, "the shapes of this test exercise one split factor only: %s" % seen
