"""GPU parity of the recurrent bottleneck path (BOTTLENECK=CLSTM|CGRU, BASELINE config 4) against the CPU oracle's
restatement of the published ConvBLSTM / ConvBGRU cells.  PARITY UNPINNED with respect to the reference itself (its
submodule is empty, no golden exists); these tests pin the HIP path to the oracle, and tests/test_oracle_recurrent.py
pins the oracle to hand-computed known answers."""
import pytest
import torch

from oracle import ssm_oracle as O

pytestmark = pytest.mark.gpu

TOL_FRAME = 1e-3
TOL_STATE = 1e-4       # hidden states are in (-1, 1)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def _cfg(kind):
    from ssm_amd.config import load_config, synthetic_weight_overrides
    ov = synthetic_weight_overrides()
    ov[("STAGE1", "BOTTLENECK")] = ov[("STAGE2", "BOTTLENECK")] = kind
    return load_config("superslomo_recurrent.ini", ov)


def _model(kind, dev, precision):
    from models.superslomo_r import FullModel
    from ssm_amd.weights import synthetic_state_dict
    m = FullModel(_cfg(kind))
    m.stage1_model.load_state_dict(synthetic_state_dict(1, bottleneck=kind))
    m.stage2_model.load_state_dict(synthetic_state_dict(2, bottleneck=kind))
    m.precision = precision
    return m.to(dev).eval()


def test_cell_kernels_vs_formulas(dev):
    from ssm_amd import hipbind as hb
    lib = hb.load()
    torch.manual_seed(1)
    B, Hc, H, W = 3, 16, 5, 7
    gx, gh = torch.randn(B, 4 * Hc, H, W, device=dev) * 2, torch.randn(B, 4 * Hc, H, W, device=dev)
    c0 = torch.randn(B, Hc, H, W, device=dev)
    c1, h1 = torch.empty_like(c0), torch.empty_like(c0)
    h16 = hb.HPlanes(B, Hc, H, W, dev)
    hb.check(lib.ssm_convlstm_cell_fwd(hb.view_of(gx), hb.view_of(gh), hb.view_of(c0), hb.view_of(c1), hb.view_of(h1), h16.view(),
                                       B, Hc, H, W, 0, hb.stream_ptr()))
    i, f, o, g = torch.split((gx + gh).cpu(), Hc, dim=1)
    cw = torch.sigmoid(f) * c0.cpu() + torch.sigmoid(i) * torch.tanh(g)
    hw = torch.sigmoid(o) * torch.tanh(cw)
    assert float((c1.cpu() - cw).abs().max()) < 2e-6 and float((h1.cpu() - hw).abs().max()) < 2e-6
    assert float((h16.to_nchw().cpu() - hw).abs().max()) < 2e-6             # hi+lo fp16 carries ~22 bits
    # first step: no hidden-state terms
    hb.check(lib.ssm_convlstm_cell_fwd(hb.view_of(gx), hb.NULL_VIEW, hb.NULL_VIEW, hb.view_of(c1), hb.view_of(h1), hb.NULL_HVIEW,
                                       B, Hc, H, W, 0, hb.stream_ptr()))
    i, f, o, g = torch.split(gx.cpu(), Hc, dim=1)
    cw = torch.sigmoid(i) * torch.tanh(g)
    assert float((h1.cpu() - torch.sigmoid(o) * torch.tanh(cw)).abs().max()) < 2e-6
    # GRU halves
    g2x, g2h = gx[:, :2 * Hc].contiguous(), gh[:, :2 * Hc].contiguous()
    cx, ch = gx[:, 2 * Hc:3 * Hc].contiguous(), gh[:, 2 * Hc:3 * Hc].contiguous()
    rh, hn = torch.empty_like(c0), torch.empty_like(c0)
    hb.check(lib.ssm_convgru_reset_fwd(hb.view_of(g2x), hb.view_of(g2h), hb.view_of(c0), hb.view_of(rh), hb.NULL_HVIEW, B, Hc, H, W,
                                       0, hb.stream_ptr()))
    hb.check(lib.ssm_convgru_update_fwd(hb.view_of(g2x), hb.view_of(g2h), hb.view_of(cx), hb.view_of(ch), hb.view_of(c0),
                                        hb.view_of(hn), h16.view(), B, Hc, H, W, 0, hb.stream_ptr()))
    gam, bet = torch.split((g2x + g2h).cpu(), Hc, dim=1)
    assert float((rh.cpu() - torch.sigmoid(gam) * c0.cpu()).abs().max()) < 2e-6
    u = torch.sigmoid(bet)
    want = (1 - u) * c0.cpu() + u * torch.tanh((cx + ch).cpu())
    assert float((hn.cpu() - want).abs().max()) < 2e-6 and float((h16.to_nchw().cpu() - want).abs().max()) < 2e-6
    with pytest.raises(RuntimeError):          # hidden channels must be a multiple of 8
        hb.check(lib.ssm_convlstm_cell_fwd(hb.view_of(gx), hb.NULL_VIEW, hb.NULL_VIEW, hb.view_of(c1), hb.view_of(h1), hb.NULL_HVIEW,
                                           B, 12, H, W, 0, hb.stream_ptr()))


@pytest.mark.parametrize("kind", ["CLSTM", "CGRU"])
@pytest.mark.parametrize("mode", ["f32", "f32w", "f16x3", "f16f8"])
def test_bottleneck_engine_vs_oracle(dev, kind, mode):
    from ssm_amd import hipbind as hb
    from ssm_amd.engine import RecurrentBottleneck
    from ssm_amd.weights import synthetic_state_dict
    sd = {k: v for k, v in synthetic_state_dict(2, bottleneck=kind).items() if k.startswith("conv6.")}
    S, T, h, w = 2, 3, 6, 10
    torch.manual_seed(2)
    xs = [torch.randn(S, 512, h, w) * 0.3 for _ in range(T)]
    want = O.unet_bottleneck_recurrent(sd, kind, xs)
    rb = RecurrentBottleneck(kind, sd, S, T, h, w, dev, mode)
    P = (lambda *a: hb.HPlanes(*a, q8=mode == "f16f8")) if mode not in ("f32", "f32w") else hb.Planes
    x, out = P(T * S, 512, h, w, dev), P(T * S, 512, h, w, dev)
    x.load(torch.cat(xs, 0).to(dev))
    rb.run(x, out)
    got = out.to_nchw().cpu().reshape(T, S, 512, h, w)
    for k in range(T):
        assert float((got[k] - want[k]).abs().max()) < (3e-4 if mode == "f16f8" else TOL_STATE), (kind, mode, k)


@pytest.mark.parametrize("kind", ["CLSTM", "CGRU"])
def test_module_forward_with_explicit_reverse_input(dev, kind):
    """ConvBLSTM/ConvBGRU.forward(x_fwd, x_rev) honours an x_rev that is not the flip of x_fwd (the published API takes both)."""
    from models.CLSTM.convgru import ConvBGRU
    from models.CLSTM.convlstm import ConvBLSTM
    from ssm_amd.weights import synthetic_state_dict
    sd = {k[len("conv6."):]: v for k, v in synthetic_state_dict(1, bottleneck=kind).items() if k.startswith("conv6.")}
    mod = (ConvBLSTM if kind == "CLSTM" else ConvBGRU)(in_channels=512, hidden_channels=512, kernel_size=(3, 3), num_layers=2,
                                                       batch_first=True)
    mod.load_state_dict(sd)
    mod = mod.to(dev).eval()
    torch.manual_seed(3)
    xf, xr = torch.randn(2, 3, 512, 4, 4) * 0.3, torch.randn(2, 3, 512, 4, 4) * 0.3
    with torch.no_grad():
        got = mod(xf.to(dev), xr.to(dev)).cpu()
    p = {"conv6." + k: v for k, v in sd.items()}
    want = O.unet_bottleneck_recurrent(p, kind, list(xf.unbind(1)), list(xr.unbind(1)))
    assert got.shape == (2, 3, 512, 4, 4)
    assert float((got - torch.stack(want, 1)).abs().max()) < TOL_STATE
    with pytest.raises(NotImplementedError):
        ConvBLSTM(in_channels=64, hidden_channels=512, kernel_size=(3, 3), num_layers=2, batch_first=True)


@pytest.mark.parametrize("kind", ["CLSTM", "CGRU"])
def test_stage_models_windows(dev, kind):
    """Stage forward on [B,T,C,H,W] with coupled windows: planned path == oracle; op-by-op public methods == planned."""
    from models import unetflow
    from ssm_amd.weights import synthetic_frames, synthetic_state_dict
    cfg = _cfg(kind)
    p1 = synthetic_state_dict(1, bottleneck=kind)
    s1 = unetflow.get_model(None, 6, 4, True, stage=1, cfg=cfg)
    s1.load_state_dict(p1)
    s1 = s1.to(dev).eval()
    fr = torch.cat([synthetic_frames(4, 64, 64, seed=5), synthetic_frames(4, 64, 64, seed=6)], 0)      # [2,4,3,64,64]
    pairs = torch.cat([fr[:, :-1], fr[:, 1:]], dim=2)                                                    # [2,3,6,64,64]
    encs, flows = O.stage_forward(p1, list(pairs.unbind(1)), kind)
    with torch.no_grad():
        outs = s1(pairs.to(dev))
        assert len(outs) == 3
        for k, (enc, flow) in enumerate(outs):
            assert float((enc.cpu() - encs[k]).abs().max()) < TOL_STATE
            assert float((flow.cpu() - flows[k]).abs().max()) < 3e-4
        es = [s1.encoder(pairs[:, k].to(dev)) for k in range(3)]
        h = s1.bottleneck([e[-1] for e in es])
        assert tuple(h.shape) == (2, 3, 512, 2, 2)
        enc2, flow2 = s1.decoder(h[:, 1], es[1])
        assert float((flow2 - outs[1][1]).abs().max()) < 1e-4


@pytest.mark.parametrize("kind", ["CLSTM", "CGRU"])
@pytest.mark.parametrize("precision", ["f16f8", "f16x3", "f32", "f32w"])
def test_full_model_recurrent_vs_oracle(dev, kind, precision):
    from ssm_amd.weights import synthetic_frames, synthetic_state_dict
    m = _model(kind, dev, precision)
    p1, p2 = synthetic_state_dict(1, bottleneck=kind), synthetic_state_dict(2, bottleneck=kind)
    x = torch.cat([synthetic_frames(4, 64, 96, seed=7), synthetic_frames(4, 64, 96, seed=8)], 0)         # B=2, N=4
    t = torch.tensor([[0.25, 0.5, 0.75], [0.625, 0.125, 0.875]]).reshape(2, 3, 1, 1, 1)
    want_img, want_inter = O.full_model_infer_windows(p1, p2, x, t, True, kind)
    img, inter = m(x.to(dev), t.to(dev), inference_mode=True)
    assert float((img.cpu() - want_img).abs().max()) < TOL_FRAME
    for name, a, b in zip(("F01", "F10", "Ft1e", "Ft0e", "Ft1", "Ft0", "V0"), inter, want_inter):
        assert float((a.cpu() - b).abs().max()) < TOL_FRAME, name
    # the loss path (all windows decoded) returns the same middle-window frame
    tgt = torch.zeros(2, 3, 3, 64, 96, device=dev)
    img2, losses = m(x.to(dev), t.to(dev), target_images=tgt, inference_mode=False)
    # (two launch plans of the same arithmetic: the loss path decodes every window, so its batch - and with it tile configurations and
    # the split-K factors of the bottleneck layers, i.e. the order of the fp32 sums - differs from the inference plan's)
    assert float((img2 - img).abs().max()) < 5e-5 and tuple(losses.shape) == (2, 4) and bool(torch.isfinite(losses).all())
    if precision == "f32w":
        # A/B of the split itself: with split-K OFF in both plan functions the two plans meet the old 1e-5 bar, and the frame
        # moves by less than 3e-5 when the split is switched back on - the wider bar above is the split's reordering and nothing else
        from ssm_amd import hipbind as hb
        prev = hb.load().ssm_splitk_enable(0, 0)
        try:
            m0 = _model(kind, dev, precision)
            img_off, _ = m0(x.to(dev), t.to(dev), inference_mode=True)
            img2_off, _ = m0(x.to(dev), t.to(dev), target_images=tgt, inference_mode=False)
        finally:
            hb.load().ssm_splitk_enable(prev & 1, prev >> 1)
        assert float((img2_off - img_off).abs().max()) < 1e-5, "split-K off: the two plans differ by %.2e" % float((img2_off - img_off).abs().max())
        assert float((img_off - img).abs().max()) < 3e-5, "switching split-K on moves the frame by %.2e" % float((img_off - img).abs().max())


def test_interpolate_windows_hoisted(dev):
    """interpolate_windows (stage 1 once per clip, t batched) == one forward per t."""
    from ssm_amd.weights import synthetic_frames
    m = _model("CLSTM", dev, "f16f8")
    x = synthetic_frames(4, 64, 64, seed=9).to(dev)
    ts = [0.125, 0.5, 0.875]
    frames = m.interpolate_windows(x, ts)
    assert tuple(frames.shape) == (3, 3, 64, 64)
    for j, tv in enumerate(ts):
        img, _ = m(x, torch.full((1, 3, 1, 1, 1), tv, device=dev), inference_mode=True)
        assert float((frames[j] - img[0]).abs().max()) < 1e-4, tv


@pytest.mark.parametrize("kind", ["CLSTM", "CGRU"])
def test_bottleneck_module_gradients_vs_oracle_autograd(dev, kind):
    """Back-propagation through time over the HIP cell adjoints (ssm_convlstm_cell_bwd / ssm_convgru_*_bwd) and the conv
    backward kernels: gradients of a random functional of conv6(x_fwd, x_rev) wrt the inputs and all parameters."""
    from models.CLSTM.convgru import ConvBGRU
    from models.CLSTM.convlstm import ConvBLSTM
    from ssm_amd.weights import synthetic_state_dict
    sd = {k[len("conv6."):]: v for k, v in synthetic_state_dict(2, bottleneck=kind).items() if k.startswith("conv6.")}
    mod = (ConvBLSTM if kind == "CLSTM" else ConvBGRU)(in_channels=512, hidden_channels=512, kernel_size=(3, 3), num_layers=2,
                                                       batch_first=True)
    mod.load_state_dict(sd)
    mod = mod.to(dev).train()
    torch.manual_seed(4)
    B, T, h, w = 2, 3, 4, 6
    x = torch.randn(B, T, 512, h, w) * 0.3
    R = torch.randn(B, T, 512, h, w)
    xd = x.to(dev).requires_grad_()
    y = mod(xd, xd.flip(1))
    assert y.requires_grad and tuple(y.shape) == (B, T, 512, h, w)
    (y * R.to(dev)).sum().backward()
    p = {"conv6." + k: v.clone().requires_grad_() for k, v in sd.items()}
    xr = x.clone().requires_grad_()
    want = torch.stack(O.unet_bottleneck_recurrent(p, kind, list(xr.unbind(1))), 1)
    assert float((y.detach().cpu() - want.detach()).abs().max()) < TOL_STATE
    (want * R).sum().backward()
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))     # noqa: E731
    assert rel(xd.grad.cpu(), xr.grad) < 2e-3, "d input"
    for name, prm in mod.named_parameters():
        assert rel(prm.grad.cpu(), p["conv6." + name].grad) < 2e-3, name


def test_recurrent_full_model_trains(dev):
    """superslomo_recurrent.ini with FREEZE=FALSE: forward + losses + backward through both recurrent U-Nets; every
    parameter (including the ConvBLSTM cells of both stages) receives a finite, non-zero gradient and Adam lowers the loss."""
    from models.superslomo_r import FullModel
    from ssm_amd.config import load_config, synthetic_weight_overrides
    from ssm_amd.weights import synthetic_frames, synthetic_state_dict
    ov = synthetic_weight_overrides()
    ov[("STAGE1", "FREEZE")] = ov[("STAGE2", "FREEZE")] = "FALSE"
    m = FullModel(load_config("superslomo_recurrent.ini", ov))
    m.stage1_model.load_state_dict(synthetic_state_dict(1, bottleneck="CLSTM"))
    m.stage2_model.load_state_dict(synthetic_state_dict(2, bottleneck="CLSTM"))
    m = m.to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    x = synthetic_frames(4, 64, 64, seed=11).to(dev)
    tgt = synthetic_frames(3, 64, 64, seed=12).to(dev)
    t = torch.tensor([0.25, 0.5, 0.75], device=dev).view(1, 3, 1, 1, 1)
    hist = []
    for it in range(3):
        _, losses = m(x, t, tgt, None, False)
        loss = losses.mean(dim=0)[0]
        opt.zero_grad()
        loss.backward()
        if it == 0:
            for name, p in m.named_parameters():
                assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0, name
        opt.step()
        hist.append(float(loss.detach()))
    assert hist[-1] < hist[0], hist


@pytest.fixture(scope="module")
def oracle_config4_720p():
    """The CPU oracle's t = 0.5 frame of the config-4 clip, once for the three precision cases (it is ~40 s of host time)."""
    from ssm_amd.weights import synthetic_frames, synthetic_state_dict
    x = synthetic_frames(4, 720, 1280, seed=42)
    p1, p2 = synthetic_state_dict(1, bottleneck="CLSTM"), synthetic_state_dict(2, bottleneck="CLSTM")
    with torch.no_grad():
        want, _ = O.full_model_infer_windows(p1, p2, x, torch.full((1, 3, 1, 1, 1), 0.5), True, "CLSTM")
    return x, want


@pytest.mark.parametrize("precision", ["f32", "f32w", "f16f8"])
def test_recurrent_config4_at_720p(dev, precision, oracle_config4_720p):
    """BASELINE config 4 at its workload size: superslomo_recurrent.ini (N_FRAMES = 4, ConvBLSTM bottleneck) on a
    1280x720 clip (padded 736x1280): (a) deterministic, (b) the hoisted 7-t evaluation (stage 1 + its BLSTM once per clip,
    t values batched) equals one forward per t, (c) one t against the CPU oracle (restated cells: parity with the
    un-vendored upstream module is unpinned, oracle/ssm_oracle.py)."""
    m = _model("CLSTM", dev, precision)
    x, want = oracle_config4_720p
    assert tuple(x.shape) == (1, 4, 3, 736, 1280)
    xd = x.to(dev)
    ts = [i / 8.0 for i in range(1, 8)]
    a = m.interpolate_windows(xd, ts).clone()
    b = m.interpolate_windows(xd, ts)
    assert torch.isfinite(a).all() and torch.equal(a, b), "same clip must give bit-identical frames"
    # (b) hoisted vs per-t: two fp32 evaluations of the same function at different batch sizes (7 vs 1: other tile kinds, other
    # summation order; the network amplifies fp32 rounding ~1000x).  Neither is "the" answer; each is held to the CPU oracle below at
    # 5e-4 (the parity statement), and the spread between them to the same 5e-4 (measured 3.1e-4 with the F(4x4) forms, r4; 1e-3 -
    # the sum of the two oracle bars - would let a real regression of one evaluation through).
    per_t = {}
    for j in (0, 3, 6):
        one, _ = m(xd, torch.full((1, 3, 1, 1, 1), ts[j], device=dev), inference_mode=True)
        per_t[j] = one[0].clone()
        spread = float((one[0] - a[j]).abs().max())
        print("recurrent 720p [%s]: max|hoisted - per-t| at t=%.3f = %.3e" % (precision, ts[j], spread))
        assert spread < 5e-4, "hoisted != per-t at t=%.3f" % ts[j]
    bar = 6e-4 if precision == "f16f8" else 5e-4          # north-star tolerance 1e-3; measured 1.2e-4 in f32w (r3)
    err = float((a[3:4].cpu() - want).abs().max())
    err1 = float((per_t[3].cpu() - want[0]).abs().max())
    print("recurrent 720p [%s]: max|HIP - oracle| at t=0.5 = %.3e (hoisted), %.3e (per-t)" % (precision, err, err1))
    assert err < bar and err1 < bar, (err, err1)
