"""GPU parity of the Q8 operand form (precision mode "f16f8"): every product = one fp16 MFMA (a_hi*b_hi) + two block-scaled
fp8 MFMAs for the compensation terms (K = 4 taps x 16 channels, E8M0 scale 2^-11 on the lo operand), on the HL8 geometry with the
second planes holding [fp8(x) | fp8(lo*2^11)] per pair of channel groups.  Checked against the CPU oracle through the C ABI.
Tolerances: a Q8 tensor read back to fp32 carries |x| * 2^-16 of representation error (lo is kept to 4 bits), so op-level outputs
of magnitude <= 5 are held to 1.5e-4; end-to-end frames to the north-star 1e-3 (tests/test_hip_model.py runs this mode too)."""
import pytest
import torch

from oracle import ssm_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1.5e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def test_q8_layout_round_trip_and_guards(dev):
    from ssm_amd import hipbind as hb
    torch.manual_seed(0)
    x = torch.randn(2, 40, 9, 21) * 3
    x[0, 0, 0, 0], x[0, 1, 0, 0] = 600.0, -1e4          # beyond the fp8 range: the fp8 copies saturate, hi stays exact
    xp = hb.HPlanes(2, 40, 9, 21, dev, q8=True).load(x.to(dev))
    assert xp.G == 6                                     # 5 groups rounded up to a pair
    back = xp.to_nchw().cpu()
    assert bool(torch.isfinite(back).all())
    rel = ((back - x).abs() / x.abs().clamp_min(1.0)).max()
    assert float(rel) < 2.0 ** -11                      # never worse than plain fp16; typically 2^-16
    small = (x.abs() < 448)
    assert float(((back - x).abs() / x.abs().clamp_min(1.0))[small].max()) < 2.0 ** -14       # lo kept to ~4 bits below hi's 11
    lib = hb.load()
    odd = hb.HPlanes(1, 8, 4, 4, dev)                   # one group: not a valid Q8 tensor
    with pytest.raises(RuntimeError):
        hb.check(lib.ssm_hq8_from_f32(hb.view_of(torch.zeros(1, 8, 4, 4, device=dev)), odd.view(), 1, 8, 1, 4, 4, hb.stream_ptr()))


@pytest.mark.parametrize("k,cin,cout,B,H,W,pool", [
    (3, 128, 128, 2, 24, 70, False),      # N128 tile, ragged width
    (3, 64, 64, 1, 40, 64, True),         # N64 tile + fused 2x2 mean
    (3, 32, 32, 1, 32, 96, False),        # N32 tile (2 workgroups per CU)
    (3, 256, 256, 1, 12, 40, False),      # N128 narrow tile (single patch buffer)
    (3, 512, 256, 1, 8, 24, False),       # deep K, two cout blocks
    (3, 64, 32, 2, 16, 64, False),        # N32 deep-K tile
    (5, 64, 64, 2, 24, 64, True),         # k = 5: second tap group holds one real tap
    (7, 32, 32, 1, 24, 64, True),         # k = 7
    (7, 16, 32, 1, 16, 40, False),        # one 16-channel chunk
    (3, 6, 32, 1, 16, 32, False),         # channels padded to a chunk
    (3, 32, 5, 1, 16, 64, False),         # bare final_conv shape: fp32 output, no activation
])
def test_q8_conv_vs_oracle(dev, k, cin, cout, B, H, W, pool):
    from ssm_amd import hipbind as hb
    torch.manual_seed(k * 1000 + cin + cout)
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    b = torch.randn(cout) * 0.1
    x = torch.randn(B, cin, H, W)
    final = cout % 16 != 0
    want = O.conv2d(x, w, b) if final else O.conv2d_lrelu(x, w, b)
    pk = hb.PackedConv16(w.to(dev), b.to(dev), W, q8=True)
    xp = hb.HPlanes(B, cin, H, W, dev, groups=pk.cin_p // 8, q8=True).load(x.to(dev))
    yp = None if final else hb.HPlanes(B, cout, H, W, dev, q8=True)
    y32 = torch.empty(B, cout, H, W, device=dev) if final else None
    pp = hb.HPlanes(B, cout, H // 2, W // 2, dev, q8=True) if pool else None
    hb.conv2d_hl8(xp.view(), pk.cin_p, None, 0, pk, yp.view() if yp else None, hb.view_of(y32) if final else None,
                  pp.view() if pool else None, B, H, W, lrelu=not final)
    got = y32.cpu() if final else yp.to_nchw().cpu()
    assert float((got - want).abs().max()) < TOL
    if pool:
        assert float((pp.to_nchw().cpu() - O.avg_pool2(want)).abs().max()) < TOL


def test_q8_two_source_concat(dev):
    """fuse_conv shape: torch.cat of two 32-channel tensors read in place."""
    from ssm_amd import hipbind as hb
    torch.manual_seed(5)
    B, H, W = 2, 16, 64
    a, c = torch.randn(B, 32, H, W), torch.randn(B, 32, H, W)
    w, b = torch.randn(32, 64, 3, 3) / 24.0, torch.randn(32) * 0.1
    pk = hb.PackedConv16(w.to(dev), b.to(dev), W, q8=True)
    ap = hb.HPlanes(B, 32, H, W, dev, q8=True).load(a.to(dev))
    cp = hb.HPlanes(B, 32, H, W, dev, q8=True).load(c.to(dev))
    yp = hb.HPlanes(B, 32, H, W, dev, q8=True)
    hb.conv2d_hl8(ap.view(), 32, cp.view(), 32, pk, yp.view(), None, None, B, H, W, lrelu=True)
    assert float((yp.to_nchw().cpu() - O.conv2d_lrelu(torch.cat([a, c], 1), w, b)).abs().max()) < TOL


@pytest.mark.parametrize("ca,cb,cout,B,h,w", [(64, 64, 32, 1, 16, 40), (128, 128, 64, 2, 12, 32), (256, 256, 128, 1, 8, 24),
                                             (512, 512, 256, 1, 6, 10), (512, 0, 512, 1, 4, 6), (64, 64, 32, 1, 9, 17)])
def test_q8_fused_upsample_conv_vs_oracle(dev, ca, cb, cout, B, h, w):
    from ssm_amd import hipbind as hb
    torch.manual_seed(ca + cb + cout)
    wt = torch.randn(cout, ca + cb, 3, 3) / ((ca + cb) * 9) ** 0.5
    bs = torch.randn(cout) * 0.1
    a = torch.randn(B, ca, h, w)
    b = torch.randn(B, cb, h, w) if cb else None
    want = O.conv2d_lrelu(O.upsample2x_bilinear(torch.cat([a, b], 1) if cb else a), wt, bs)
    pk = hb.PackedConv16(wt.to(dev), bs.to(dev), 2 * w, q8=True, ups=True)
    ap = hb.HPlanes(B, ca, h, w, dev, q8=True).load(a.to(dev))
    bp = hb.HPlanes(B, cb, h, w, dev, q8=True).load(b.to(dev)) if cb else None
    yp = hb.HPlanes(B, cout, 2 * h, 2 * w, dev, q8=True)
    hb.conv2d_ups_hl8(ap.view(), ca, bp.view() if cb else None, cb, pk, yp.view(), None, B, 2 * h, 2 * w, lrelu=True)
    assert float((yp.to_nchw().cpu() - want).abs().max()) < TOL


def test_f16f8_pipeline_vs_f32_path_720p(dev):
    """The whole pair -> 7 frames path in mode f16f8 against the exact-fp32 MFMA mode at 736x1280 (both are held to 1e-3
    against the oracle; their mutual distance is the budget the fp8 compensation spends)."""
    from ssm_amd.engine import PairEngine
    from ssm_amd.weights import synthetic_frames, synthetic_state_dict
    sd1 = {k: v.to(dev) for k, v in synthetic_state_dict(1).items()}
    sd2 = {k: v.to(dev) for k, v in synthetic_state_dict(2).items()}
    x = synthetic_frames(2, 720, 1280, seed=3)
    H, W = x.shape[-2:]
    img6 = x.reshape(1, 6, H, W).to(dev)
    t = torch.tensor([i / 8.0 for i in range(1, 8)], device=dev)
    a = PairEngine(sd1, sd2, 1, 7, H, W, dev, True, "f16f8").run(img6, t, want_aux=False).clone()
    ref = PairEngine(sd1, sd2, 1, 7, H, W, dev, True, "f32").run(img6, t, want_aux=False)
    d = float((a - ref).abs().max())
    print("f16f8 vs f32 path at 736x1280: max-abs %.2e" % d)
    assert d < 7e-4


def test_batched_pack_matches_per_layer_pack(dev):
    """ssm_pack16q_weights_batch (one launch for many filters; transposed jobs read the forward OIHW tensor) writes the same bytes
    as ssm_pack16q_weights per layer on the forward filter / on the materialised transposed + flipped filter."""
    from ssm_amd import hipbind as hb
    from ssm_amd.backward import transposed_filter
    g = torch.Generator().manual_seed(77)
    shapes = [(32, 16, 7, 128), (64, 32, 5, 64), (128, 96, 3, 32), (5, 32, 3, 128), (256, 512, 3, 16), (32, 6, 7, 128)]
    entries_f, entries_t, want = [], [], []
    for co, ci, k, W in shapes:
        w = torch.randn(co, ci, k, k, generator=g).to(dev)
        b = torch.randn(co, generator=g).to(dev)
        ref_f = hb.PackedConv16(w, b, W, q8=True)
        ref_t = hb.PackedConv16(transposed_filter(w), torch.zeros(ci, device=dev), W, q8=True, scale=ref_f.scale)
        pf = hb.PackedConv16(None, None, W, q8=True, scale=ref_f.scale, shape=(co, ci, k), device=dev)
        pt = hb.PackedConv16(None, None, W, q8=True, scale=ref_f.scale, shape=(ci, co, k), device=dev)
        pf.w.fill_(0x5a), pt.w.fill_(0x5a), pf.b.fill_(3.0), pt.b.fill_(3.0)
        entries_f.append((pf, w, b, False))
        entries_t.append((pt, w, None, True))
        want.append((ref_f, ref_t))
    hb.PackBatch(entries_f + entries_t, dev).run()
    torch.cuda.synchronize()
    for (pf, _, _, _), (pt, _, _, _), (rf, rt), sh in zip(entries_f, entries_t, want, shapes):
        assert (pf.bn, pf.kys, pt.bn, pt.kys) == (rf.bn, rf.kys, rt.bn, rt.kys)
        assert torch.equal(pf.w, rf.w) and torch.equal(pf.b, rf.b), "forward filter %s" % (sh,)
        assert torch.equal(pt.w, rt.w) and torch.equal(pt.b, rt.b), "data-gradient filter %s" % (sh,)


@pytest.mark.parametrize("B,ca,cb,co,h,w", [(2, 64, 64, 32, 12, 40), (1, 32, 0, 64, 9, 70), (1, 128, 128, 128, 6, 33), (1, 16, 16, 16, 2, 2),
                                            (1, 256, 256, 256, 5, 7)])
def test_subpixel_upconv_matches_upsample_then_conv(dev, B, ca, cb, co, h, w):
    """conv3x3(upsample2x(cat[a, b])) in the sub-pixel form (9 launches of the plain Q8 kernel: main + 4 border strips + 4 corners)
    against the oracle's upsample + conv on the same Q8-representable inputs: interior, the two border rows / columns and the
    corners all within 1e-4 of the output's scale."""
    from ssm_amd import hipbind as hb
    from ssm_amd.subpixel import SubpixelUpConv
    g = torch.Generator().manual_seed(1000 * ca + co + h * w)
    a = torch.randn(B, ca, h, w, generator=g)
    b = torch.randn(B, cb, h, w, generator=g) if cb else None
    wt = torch.randn(co, ca + cb, 3, 3, generator=g) / ((ca + cb) * 9) ** 0.5
    bias = torch.randn(co, generator=g) * 0.1
    A = hb.HPlanes(B, ca, h, w, dev, q8=True).load(a.to(dev))
    Bp = hb.HPlanes(B, cb, h, w, dev, q8=True).load(b.to(dev)) if cb else None
    aq = A.to_nchw().cpu()                       # what the kernel actually reads (fp16 hi + fp8 lo)
    bq = Bp.to_nchw().cpu() if cb else None
    up = O.upsample2x_bilinear(torch.cat([aq, bq], 1) if cb else aq)
    want = O.conv2d_lrelu(up, wt, bias)
    dst = hb.HPlanes(B, co, 2 * h, 2 * w, dev, q8=True)
    sp = SubpixelUpConv(wt, bias, A.G, Bp.G if cb else 0, B, h, w, dev)
    sp.run(lambda y0, x0: A.view(y0=y0, x0=x0), (lambda y0, x0: Bp.view(y0=y0, x0=x0)) if cb else None, dst)
    got = dst.to_nchw().cpu()
    err = (got - want).abs()
    scale = float(want.abs().max())
    ring = torch.ones_like(err, dtype=torch.bool)
    ring[:, :, 2:-2, 2:-2] = False
    print("sub-pixel upconv %s: max err interior %.2e, border ring %.2e (scale %.2f)" % ((B, ca, cb, co, h, w), float(err[~ring].max()) if (~ring).any() else 0.0,
                                                                                         float(err[ring].max()), scale))
    assert float(err.max()) < 1e-4 * max(scale, 1.0) + 2e-4


def test_subpixel_levels_match_fused_upsample_plan(dev, monkeypatch):
    """Whole pair engines at 96x160, 3 intermediates: the plan with conv11a / conv10a in the sub-pixel form against the plan with
    the fused-upsample kernel everywhere - the two levels' outputs (both stages) and the frames agree to the fp8-compensation
    noise (2e-4 of the tensors' scale), borders included."""
    from ssm_amd.engine import PairEngine, UNetPlan
    from ssm_amd.weights import synthetic_frames, synthetic_state_dict
    sd1 = {k: v.to(dev) for k, v in synthetic_state_dict(1).items()}
    sd2 = {k: v.to(dev) for k, v in synthetic_state_dict(2).items()}
    x = synthetic_frames(2, 96, 160, seed=5).to(dev).reshape(1, 6, 96, 160)
    t = torch.tensor([0.25, 0.5, 0.875], device=dev)
    outs = {}
    for label, levels in (("fused", ()), ("subpixel", ("conv11a", "conv10a"))):
        monkeypatch.setattr(UNetPlan, "SUBPIXEL", levels)
        eng = PairEngine(sd1, sd2, 1, 3, 96, 160, dev, True, "f16f8")
        img = eng.run(x, t, False).clone()
        assert (len(eng.s1.sp), len(eng.s2.sp)) == ((2, 2) if levels else (0, 0))
        outs[label] = (img, {"%s.%s" % (st, n): pl.t[n].to_nchw() for st, pl in (("s1", eng.s1), ("s2", eng.s2)) for n in ("t11a", "t10a")})
    for n, a in outs["fused"][1].items():
        b = outs["subpixel"][1][n]
        err = (a - b).abs()
        assert float(err.max()) < 2e-4 * max(float(a.abs().max()), 1.0), n
        ring = float(torch.cat([err[:, :, :2].flatten(), err[:, :, -2:].flatten(), err[:, :, :, :2].flatten(), err[:, :, :, -2:].flatten()]).max())
        assert ring < 2e-4 * max(float(a.abs().max()), 1.0), n + " border"
    assert float((outs["fused"][0] - outs["subpixel"][0]).abs().max()) < 5e-4
