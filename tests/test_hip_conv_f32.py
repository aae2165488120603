"""GPU parity of the fp32-MFMA convolution (csrc/ssm_conv.hip) for EVERY tile configuration - forced one by one through
ssm_conv_force_kind - against the CPU oracle: plain conv, two-source (torch.cat) input, fused 2x2 mean, and the fused
concat + bilinear x2 upsample + conv (scripts/models/flow_computation.py:244-247) against conv(upsample(cat)) of the
oracle.  Ragged sizes on purpose: tiles overshoot the map on both axes, 8x4 and 32x1 pixel groups."""
import pytest
import torch

pytestmark = pytest.mark.gpu

KINDS = ["K7", "K5", "K3N32", "K3N64", "K3N128", "K3N128S", "K3N64T", "K3N32T", "K3N128G", "K3N64G", "K3N64GS", "K3N32G",
         "K3N32GS", "K5G", "K7G"]
KS = {"K7": 7, "K7G": 7, "K5": 5, "K5G": 5}
NO_POOL = {"K3N32T"}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _unforce():
    yield
    from ssm_amd import hipbind as hb
    hb.load().ssm_conv_force_kind(-1)


def _force(kind):
    from ssm_amd import hipbind as hb
    n = hb.load().ssm_conv_force_kind(KINDS.index(kind))
    assert n == len(KINDS), "tile-configuration list of the test is out of date (%d in the library)" % n


def _err(got, want):
    return float((got - want).abs().max())


@pytest.mark.parametrize("kind", KINDS)
def test_every_tile_configuration_plain_cat_pool(dev, kind):
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    k = KS.get(kind, 3)
    g = torch.Generator().manual_seed(KINDS.index(kind))
    B, H, W = 2, 22, 44        # ragged for every tile (no TH / TW divides both) and even (pool)
    c1, c2, cout = 16, 8, 40   # cat of two sources; 40 couts: a partly filled cout block for BN = 32 / 64 / 128
    a, b = torch.randn(B, c1, H, W, generator=g), torch.randn(B, c2, H, W, generator=g)
    w = torch.randn(cout, c1 + c2, k, k, generator=g) / ((c1 + c2) * k * k) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    want = O.conv2d_lrelu(torch.cat([a, b], 1), w, bias)
    _force(kind)
    pool = kind not in NO_POOL
    kd, bn, ck = hb.conv_plan(k, c1 + c2, cout, B, H, W, pool)
    assert kd == KINDS.index(kind)
    pa, pb = hb.Planes(B, c1, H, W, dev).load(a.to(dev)), hb.Planes(B, c2, H, W, dev).load(b.to(dev))
    y, yp = hb.Planes(B, cout, H, W, dev), hb.Planes(B, cout, H // 2, W // 2, dev)
    pk = hb.PackedConv(w.to(dev), bias.to(dev), B, H, W, pool=pool)
    hb.conv2d(pa.view(), c1, pb.view(), c2, pk, y.view(), yp.view() if pool else None, B, H, W, lrelu=True)
    got = y.to_nchw().cpu()
    assert _err(got, want) < 5e-5, "%s: conv %.3e" % (kind, _err(got, want))
    if pool:
        gp = yp.to_nchw().cpu()
        assert _err(gp, O.avg_pool2(want)) < 5e-5, "%s: fused pool %.3e" % (kind, _err(gp, O.avg_pool2(want)))
    full = y.full.cpu().clone()
    full[:, :, hb.SSM_PADY:hb.SSM_PADY + H, hb.SSM_PADX:hb.SSM_PADX + W] = 0
    assert float(full.abs().max()) == 0.0, "%s wrote outside the interior" % kind


UPS_KINDS = [k for k in KINDS if k not in KS]
UPS_SHAPES = [(1, 23, 40), (2, 5, 7), (1, 8, 48), (2, 11, 11), (1, 1, 1), (1, 3, 34)]     # LOW-res (B, h, w)


@pytest.mark.parametrize("kind", UPS_KINDS)
def test_every_tile_configuration_fused_upsample(dev, kind):
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
        pk = hb.PackedConv(w.to(dev), bias.to(dev), B, H, W, ups=True)
        assert hb.conv_plan(3, c1 + c2, cout, B, H, W, False, True)[0] == KINDS.index(kind)
        hb.conv2d_ups(pa.view(), c1, pb.view(broadcast=True), c2, pk, y.view(), B, H, W)
        got = y.to_nchw().cpu()
        assert _err(got, want) < 5e-5, "%s %dx%d: fused upsample conv %.3e" % (kind, h, wd, _err(got, want))
        # single source
        pk1 = hb.PackedConv(w[:, :c1].contiguous().to(dev), bias.to(dev), B, H, W, ups=True)
        hb.conv2d_ups(pa.view(), c1, None, 0, pk1, y.view(), B, H, W, lrelu=False)
        want1 = O.conv2d(O.upsample2x_bilinear(a), w[:, :c1].contiguous(), bias)
        assert _err(y.to_nchw().cpu(), want1) < 5e-5, "%s %dx%d: single-source" % (kind, h, wd)


def test_automatic_plan_matches_oracle_on_the_unet_shapes(dev):
    """The configurations the cost model picks for the 1/16 and 1/32 maps of a 736x1280 frame (80x46, 40x23) and for
    the 352x352 crops (22x22, 11x11): conv + fused pool / fused upsample at a reduced channel count."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(3)
    for B, h, wd in ((1, 23, 40), (7, 23, 40), (2, 11, 11), (1, 46, 80)):
        cin, cout = 64, 96
        x = torch.randn(B, cin, h, wd, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        px = hb.Planes(B, cin, h, wd, dev).load(x.to(dev))
        y = hb.Planes(B, cout, h, wd, dev)
        pk = hb.PackedConv(w.to(dev), bias.to(dev), B, h, wd)
        hb.conv2d(px.view(), cin, None, 0, pk, y.view(), None, B, h, wd)
        want = O.conv2d_lrelu(x, w, bias)
        assert _err(y.to_nchw().cpu(), want) < 5e-5
        yu = hb.Planes(B, cout, 2 * h, 2 * wd, dev)
        pku = hb.PackedConv(w.to(dev), bias.to(dev), B, 2 * h, 2 * wd, ups=True)
        hb.conv2d_ups(px.view(), cin, None, 0, pku, yu.view(), B, 2 * h, 2 * wd)
        wantu = O.conv2d_lrelu(O.upsample2x_bilinear(x), w, bias)
        assert _err(yu.to_nchw().cpu(), wantu) < 5e-5


@pytest.mark.parametrize("form", ["1", "0"], ids=["valu", "mfma4x4"])
@pytest.mark.parametrize("nc", [4, 5, 8, 1])
def test_final_conv_both_forms_vs_oracle(dev, nc, form, monkeypatch):
    """final_conv (32 -> 4 / 5 channels, no activation) against the oracle's conv2d, ragged and tiny maps included (tile = 8 x 64):
    the vector-ALU form (the default for the model's 4 / 5 channels) and the v_mfma_f32_4x4x1_16B_f32 form ($SSM_FINAL_VALU=0 and
    every other channel count)."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    monkeypatch.setenv("SSM_FINAL_VALU", form)
    g = torch.Generator().manual_seed(40 + nc)
    w = torch.randn(nc, 32, 3, 3, generator=g) / (32 * 9) ** 0.5
    bias = torch.randn(nc, generator=g) * 0.1
    nv = hb.NULL_VIEW
    for B, H, W in ((1, 16, 64), (2, 13, 70), (1, 3, 5), (3, 32, 130)):
        x = torch.randn(B, 32, H, W, generator=g)
        px = hb.Planes(B, 32, H, W, dev).load(x.to(dev))
        y = hb.Planes(B, nc, H, W, dev)
        wd, bd = w.to(dev), bias.to(dev)
        hb.check(hb.load().ssm_final_conv_fwd(px.view(), wd.data_ptr(), bd.data_ptr(), nc, y.view(), nv, nv, None, nv, nv, B, H, W,
                                              hb.stream_ptr()))
        want = O.conv2d(x, w, bias)
        assert _err(y.to_nchw().cpu(), want) < 5e-5, "final_conv NC=%d %dx%dx%d: %.3e" % (nc, B, H, W, _err(y.to_nchw().cpu(), want))


@pytest.mark.parametrize("form", ["1", "0"], ids=["valu", "mfma4x4"])
def test_final_conv_fused_synthesis_vs_oracle(dev, form, monkeypatch):
    """final_conv + extract_outputs + compute_output_image (flow_interpolation.py:374-429) in one kernel == the oracle's
    synthesize(conv2d(.)) and == the two-kernel HIP path; the 5-channel map is optional.  Both forms of the convolution."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    monkeypatch.setenv("SSM_FINAL_VALU", form)
    g = torch.Generator().manual_seed(77)
    B, H, W = 3, 21, 75
    x = torch.randn(B, 32, H, W, generator=g)
    w = torch.randn(5, 32, 3, 3, generator=g) / (32 * 9) ** 0.5 * 3.0       # residual flows of a few px, logits of a few units
    bias = torch.randn(5, generator=g) * 0.1
    img6 = torch.randn(B, 6, H, W, generator=g)
    flow4 = torch.randn(B, 4, H, W, generator=g) * 2.0
    t = torch.tensor([0.125, 0.5, 0.875])
    in16 = O.flow_interp_inputs(img6, flow4, t.view(B, 1, 1, 1))
    out5 = O.conv2d(x, w, bias)
    want = O.synthesize(img6, in16, out5, t.view(B, 1, 1, 1))
    px = hb.Planes(B, 32, H, W, dev).load(x.to(dev))
    i6, i16, td = img6.to(dev), in16.to(dev), t.to(dev)
    y3 = torch.empty(B, 3, H, W, device=dev)
    aux = torch.empty(B, 5, H, W, device=dev)
    o5 = hb.Planes(B, 5, H, W, dev)
    wd, bd = w.to(dev), bias.to(dev)
    lib = hb.load()
    hb.check(lib.ssm_final_conv_fwd(px.view(), wd.data_ptr(), bd.data_ptr(), 5, o5.view(), hb.view_of(i6), hb.view_of(i16), td.data_ptr(),
                                    hb.view_of(y3), hb.view_of(aux), B, H, W, hb.stream_ptr()))
    assert _err(o5.to_nchw().cpu(), out5) < 5e-5
    assert _err(y3.cpu(), want) < 2e-4, "fused synthesis %.3e" % _err(y3.cpu(), want)      # 5e-5 of flow through |grad I| ~ 3
    # the two-kernel path on the SAME 5-channel map is bit-identical (one shared device function)
    y3b = torch.empty_like(y3)
    hb.check(lib.ssm_synthesize_fwd(hb.view_of(i6), hb.view_of(i16), o5.view(), td.data_ptr(), hb.view_of(y3b), hb.NULL_VIEW, B, H, W,
                                    hb.stream_ptr()))
    assert torch.equal(y3b, y3)
    v0 = 1 - torch.sigmoid(out5[:, 0:1])
    assert _err(aux[:, 4:5].cpu(), v0) < 1e-5 and _err(aux[:, 0:2].cpu(), in16[:, 6:8] + out5[:, 1:3]) < 5e-5


@pytest.mark.parametrize("kind,k", [("K7", 7), ("K7G", 7), ("K3N64", 3), ("K3N128G", 3), ("K5", 5)])
def test_direct_conv_pre_activation_addend(dev, kind, k):
    """y = act(conv(x) + bias + add[b // add_div]) (ssm_conv2d_add_fwd / ssm_conv2d_ups_add_fwd): the accumulators start from the addend.
    Stage 2 uses it for the image channels of conv1a (7x7, 10 of 16 input channels per t, the other 6 once per pair)."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(500 + KINDS.index(kind))
    _force(kind)
    B, div, cout = 4, 2, 128         # a whole number of cout blocks for every tile configuration (the addend is loaded unpredicated)
    cin = 12 if k == 3 else 10          # whole channel chunks of the tile configuration (3x3: 4, 5x5 / 7x7: 2)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    cases = [(False, (22, 44))] + ([(True, (11, 23))] if k == 3 else [])
    for ups, (h, wd) in cases:
        H, W = (2 * h, 2 * wd) if ups else (h, wd)
        x = torch.randn(B, 16, h, wd, generator=g)                    # the convolution reads channels 3:3+cin of a 16-channel tensor
        add = torch.randn(B // div, cout, H, W, generator=g)
        xs = x[:, 3:3 + cin]
        xin = O.upsample2x_bilinear(xs) if ups else xs
        z = O.conv2d(xin, w, bias) + add.repeat_interleave(div, 0)
        want = torch.where(z >= 0, z, z * 0.1)
        px = hb.Planes(B, 16, h, wd, dev).load(x.to(dev))
        pa = hb.Planes(B // div, cout, H, W, dev).load(add.to(dev))
        y = hb.Planes(B, cout, H, W, dev)
        pk = hb.PackedConv(w.to(dev), bias.to(dev), B, H, W, ups=ups)
        if ups:
            hb.conv2d_ups(px.view(c0=3), cin, None, 0, pk, y.view(), B, H, W, add=pa.view(), add_div=div)
        else:
            hb.conv2d(px.view(c0=3), cin, None, 0, pk, y.view(), None, B, H, W, add=pa.view(), add_div=div)
        assert _err(y.to_nchw().cpu(), want) < 5e-5, "%s ups=%d: %.3e" % (kind, ups, _err(y.to_nchw().cpu(), want))


def test_direct_conv_split_k_on_the_bottleneck_maps(dev):
    """The direct-form convolutions of config 3's bottleneck (512 -> 512 on 11x11 maps - odd width, no Winograd form - at batch 2) run split
    over the input channels inside hb.conv2d (ssm_conv2d_splitk_fwd + ssm_splitk_finish_fwd, KS from ssm_conv_splitk_plan): against the
    oracle, with and without a pre-activation addend, and bit-identical from run to run."""
    import ctypes
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(91)
    for B, cin, cout, H, W, addend in ((2, 512, 512, 11, 11, False), (2, 512, 512, 11, 11, True), (2, 256, 256, 11, 11, False), (1, 512, 64, 7, 9, False)):
        ks = ctypes.c_int(1)
        hb.check(hb.load().ssm_conv_splitk_plan(3, cin, cout, B, H, W, ctypes.byref(ks)))
        assert ks.value > 1, "%d -> %d at %dx%d, batch %d: the plan does not split" % (cin, cout, H, W, B)
        x = torch.randn(B, cin, H, W, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        add = torch.randn(B, cout, H, W, generator=g) if addend else None
        z = O.conv2d(x, w, bias) + (add if add is not None else 0.0)
        want = torch.where(z >= 0, z, z * 0.1)
        px = hb.Planes(B, cin, H, W, dev).load(x.to(dev))
        pz = hb.Planes(B, cout, H, W, dev).load(add.to(dev)) if add is not None else None
        pk = hb.PackedConv(w.to(dev), bias.to(dev), B, H, W)
        y1 = hb.Planes(B, cout, H, W, dev)
        hb.conv2d(px.view(), cin, None, 0, pk, y1.view(), None, B, H, W, add=pz.view() if pz is not None else None)      # not split: mode f32
        pk.split_ok = True          # what an f32w plan sets on its direct-form filters
        outs = []
        for _ in range(2):
            y = hb.Planes(B, cout, H, W, dev)
            hb.conv2d(px.view(), cin, None, 0, pk, y.view(), None, B, H, W, add=pz.view() if pz is not None else None)
            outs.append(y.to_nchw().cpu())
        assert "_splitk_part" in pk.__dict__, "the split path was not taken"
        assert _err(y1.to_nchw().cpu(), want) < 5e-5 and _err(outs[0], y1.to_nchw().cpu()) < 5e-5          # (4608-term fp32 sums in two orders)
        assert _err(outs[0], want) < 5e-5, "KS %d, %d -> %d at %dx%d: %.3e" % (ks.value, cin, cout, H, W, _err(outs[0], want))
        assert torch.equal(outs[0], outs[1]), "split-K result differs between two runs"
        full = y.full.cpu().clone()
        full[:, :, hb.SSM_PADY:hb.SSM_PADY + H, hb.SSM_PADX:hb.SSM_PADX + W] = 0
        assert float(full.abs().max()) == 0.0, "split-K wrote outside the interior"
