"""No Winograd-form kernel reads outside its input's padded plane: the memory behind the LAST plane of a conv input -
the tail slack include/ssm_hip.h lets a C-ABI caller leave unzeroed - is poisoned with NaN and a huge value, on map shapes whose
workgroup tiles overshoot on both axes and on exact multiples of the tile (the 7x7 form's bottom window row H + 3: the zero tap ky = 7,
which B^T d B still mixes into all 49 frequencies).  Every kernel must return finite outputs at the usual 5e-5 from the CPU oracle's
direct convolution (scripts/models/layers.py:21-33).  The per-lane LDS-DMA offsets clamp overshoot rows / 16-byte pieces to the zero
frame (csrc/ssm_wino4|5|7|wino|wino1d.hip), so the result does not depend on what follows the plane."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BAR = 5e-5


@pytest.fixture(autouse=True)
def _unforce():
    yield
    from ssm_amd import hipbind as hb
    hb.load().ssm_wino4_force_kind(-1)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def _poisoned_planes(hb, x, dev, poison):
    p = hb.Planes(x.shape[0], x.shape[1], x.shape[2], x.shape[3], dev).load(x.to(dev))
    p.buf[p.full.numel():] = poison          # everything behind the last plane
    return p


CASES = [  # (form, k, cin, cout, shapes)
    ("wino7", 7, 6, 32, ((1, 16, 32), (1, 32, 64), (1, 23, 40), (1, 9, 131))),
    ("wino5", 5, 8, 32, ((1, 16, 32), (1, 23, 40), (1, 9, 131))),
    ("wino1d", 7, 6, 32, ((1, 16, 32), (1, 23, 41))),
    ("wino1d", 5, 8, 32, ((1, 16, 32), (1, 23, 41))),
    ("wino4", 3, 8, 32, ((1, 16, 32), (1, 23, 40), (1, 9, 131), (1, 50, 66))),
    ("wino4:Y4A", 3, 64, 64, ((1, 16, 32), (1, 23, 40))),      # the 64-cout form, forced (kind 3 of csrc/ssm_wino4.hip SSM_W4_KINDS)
    ("wino", 3, 8, 32, ((1, 16, 32), (1, 23, 42), (1, 9, 130))),          # F(2x2,3x3): even W
]


@pytest.mark.parametrize("poison", [float("nan"), 3.0e30])
@pytest.mark.parametrize("form,k,cin,cout,shapes", CASES)
def test_outputs_do_not_depend_on_memory_behind_the_last_plane(dev, form, k, cin, cout, shapes, poison):
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    packers = {"wino7": (hb.PackedWino7, hb.conv2d_wino7), "wino5": (hb.PackedWino5, hb.conv2d_wino5), "wino1d": (hb.PackedWino1d, hb.conv2d_wino1d),
               "wino4": (hb.PackedWino4, hb.conv2d_wino4), "wino": (hb.PackedWino, hb.conv2d_wino)}
    Packed, conv = packers[form.split(":")[0]]
    if form.endswith(":Y4A"):
        assert hb.load().ssm_wino4_force_kind(3) == 6
    g = torch.Generator().manual_seed(k * 100 + cin)
    for B, H, W in shapes:
        x = torch.randn(B, cin, H, W, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        pk = Packed(w.to(dev), bias.to(dev), B, H, W)
        px = _poisoned_planes(hb, x, dev, poison)
        y = hb.Planes(B, cout, H, W, dev)
        conv(px.view(), cin, None, 0, pk, y.view(), None, B, H, W, lrelu=True)
        got = y.to_nchw().cpu()
        assert bool(torch.isfinite(got).all()), "%s k%d %dx%d: non-finite outputs (poison %g reached the tile)" % (form, k, H, W, poison)
        e = float((got - O.conv2d_lrelu(x, w, bias)).abs().max())
        assert e < BAR, "%s k%d %dx%dx%d: %.3e" % (form, k, B, H, W, e)


@pytest.mark.parametrize("poison", [float("nan"), 3.0e30])
@pytest.mark.parametrize("form", ["wino4", "wino"])
def test_fused_upsample_sources_poisoned_behind_the_last_plane(dev, form, poison):
    """conv3x3(upsample2x(cat[a, b])) (scripts/models/flow_computation.py:244-247): both low-res sources end in poisoned memory."""
    from oracle import ssm_oracle as O
    from ssm_amd import hipbind as hb
    Packed, conv = {"wino4": (hb.PackedWino4, hb.conv2d_ups_wino4), "wino": (hb.PackedWino, hb.conv2d_ups_wino)}[form]
    g = torch.Generator().manual_seed(77)
    for (h, w), (c1, c2, cout) in (((8, 16), (8, 8, 32)), ((11, 21), (8, 8, 32)), ((8, 16), (32, 32, 64))):
        if cout == 64 and form == "wino4":
            assert hb.load().ssm_wino4_force_kind(3) == 6          # the 64-cout form
        a, b = torch.randn(1, c1, h, w, generator=g), torch.randn(1, c2, h, w, generator=g)
        wt = torch.randn(cout, c1 + c2, 3, 3, generator=g) / ((c1 + c2) * 9) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        pk = Packed(wt.to(dev), bias.to(dev), 1, 2 * h, 2 * w, ups=True)
        pa, pb = _poisoned_planes(hb, a, dev, poison), _poisoned_planes(hb, b, dev, poison)
        y = hb.Planes(1, cout, 2 * h, 2 * w, dev)
        conv(pa.view(), c1, pb.view(), c2, pk, y.view(), 1, 2 * h, 2 * w)
        got = y.to_nchw().cpu()
        assert bool(torch.isfinite(got).all()), "%s ups %dx%d: non-finite outputs" % (form, h, w)
        want = O.conv2d_lrelu(O.upsample2x_bilinear(torch.cat([a, b], 1)), wt, bias)
        assert float((got - want).abs().max()) < BAR
