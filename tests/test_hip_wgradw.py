"""GPU parity of the Winograd-domain weight gradient of the 3x3 layers (csrc/ssm_wgradw.hip: F(2x2,3x3), dU = sum_tiles (A dZ A^T) (.)
(B^T x B), dW = G^T dU G) against CPU autograd of the oracle's convolution (layers.conv, scripts/models/layers.py:21-33) - the quantity
ssm_conv2d_wgrad computes in the direct form.  Every tile configuration (64 x 64, 32 x 64, 32 x 32 couts x cins per workgroup), ragged maps
(odd sizes, a lone last tile row, partial groups of 8 tiles), channel counts that do not fill a block, two-source layers, the fused bias
gradient, the scratch left zeroed and the accumulate-into-dW contract.  Bar 2e-4 of the largest entry like the direct kernel
(tests/test_hip_backward.py); measured ~2e-6."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BAR = 2e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def rel_err(got, want):
    return float((got - want).abs().max() / (want.abs().max() + 1e-12))


def _reference(x, dz):
    """dW, db of y = conv3x3(x, w) + b under the upstream gradient dz (fp64 autograd on the CPU)."""
    from oracle import ssm_oracle as O
    cout, cin = dz.shape[1], x.shape[1]
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    y = O.conv2d(x.double(), w, b)
    (y * dz.double()).sum().backward()
    return w.grad.float(), b.grad.float()


@pytest.mark.parametrize("B,H,W,cin,cout", [
    (2, 44, 44, 64, 64),        # 64 x 64 blocks; 22 tiles per row = 2.75 groups; 11 pairs of tile rows
    (1, 46, 88, 128, 64),       # 23 tile rows: a lone last one
    (2, 41, 43, 64, 128),       # odd sizes: the last tile row / column are half outside the image
    (1, 40, 176, 96, 32),       # 32-cout form, cins not a multiple of 64
    (2, 88, 40, 32, 32),        # 32 x 32 form
    (1, 44, 52, 40, 72),        # neither channel count fills its block
    (3, 48, 48, 256, 64),       # several cin blocks, three batch entries
])
def test_wgrad_wino_vs_autograd(dev, B, H, W, cin, cout):
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(B * 1000 + H + W + cin + cout)
    x = torch.randn(B, cin, H, W, generator=g)
    dz = torch.randn(B, cout, H, W, generator=g) * 0.1
    want_w, want_b = _reference(x, dz)
    xp = hb.Planes(B, cin, H, W, dev).load(x.to(dev))
    dzp = hb.Planes(B, cout, H, W, dev).load(dz.to(dev))
    du = torch.zeros(16, cout, cin, device=dev)
    db = torch.full((cout,), 0.5, device=dev)
    dw = torch.full((cout, cin, 3, 3), 0.25, device=dev)
    hb.wgrad_wino(xp.view(), dzp.view(), du, db, B, cin, cout, H, W, cin, 0)
    fin = hb.WgradWinoFinish([(du, dw)], dev)
    fin.run()
    assert rel_err(dw.cpu() - 0.25, want_w) < BAR, "dW %.3e" % rel_err(dw.cpu() - 0.25, want_w)
    assert rel_err(db.cpu() - 0.5, want_b) < BAR, "db %.3e" % rel_err(db.cpu() - 0.5, want_b)
    assert float(du.abs().max()) == 0.0, "the finishing launch leaves the scratch zeroed"
    # a second step accumulates into dW like the direct kernel with zero_first = 0, scaled
    hb.wgrad_wino(xp.view(), dzp.view(), du, None, B, cin, cout, H, W, cin, 0)
    fin.run(scale=0.5)
    assert rel_err(dw.cpu() - 0.25, 1.5 * want_w) < BAR


def test_wgrad_wino_two_sources_and_several_layers_per_finish(dev):
    """A two-source layer (fuse_conv: cat[c11, c1], scripts/models/flow_computation.py:276-281) fills its filter's input range by two
    launches; one finishing launch serves two layers of different sizes."""
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 48, 64
    xa, xb = torch.randn(B, 32, H, W, generator=g), torch.randn(B, 32, H, W, generator=g)
    dz = torch.randn(B, 32, H, W, generator=g) * 0.1
    want_w, _ = _reference(torch.cat([xa, xb], 1), dz)
    pa, pb = hb.Planes(B, 32, H, W, dev).load(xa.to(dev)), hb.Planes(B, 32, H, W, dev).load(xb.to(dev))
    dzp = hb.Planes(B, 32, H, W, dev).load(dz.to(dev))
    du = torch.zeros(16, 32, 64, device=dev)
    hb.wgrad_wino(pa.view(), dzp.view(), du, None, B, 32, 32, H, W, 64, 0)
    hb.wgrad_wino(pb.view(), dzp.view(), du, None, B, 32, 32, H, W, 64, 32)
    # second layer: 64 -> 64 on its own map
    x2, dz2 = torch.randn(1, 64, 44, 44, generator=g), torch.randn(1, 64, 44, 44, generator=g)
    want2, _ = _reference(x2, dz2)
    p2, dzp2 = hb.Planes(1, 64, 44, 44, dev).load(x2.to(dev)), hb.Planes(1, 64, 44, 44, dev).load(dz2.to(dev))
    du2 = torch.zeros(16, 64, 64, device=dev)
    hb.wgrad_wino(p2.view(), dzp2.view(), du2, None, 1, 64, 64, 44, 44, 64, 0)
    dw, dw2 = torch.zeros(32, 64, 3, 3, device=dev), torch.zeros(64, 64, 3, 3, device=dev)
    hb.WgradWinoFinish([(du, dw), (du2, dw2)], dev).run()
    assert rel_err(dw.cpu(), want_w) < BAR
    assert rel_err(dw2.cpu(), want2) < BAR
    assert float(du.abs().max()) == 0.0 and float(du2.abs().max()) == 0.0


def test_wgrad_wino_matches_the_direct_kernel_at_a_training_shape(dev):
    """conv10a of BASELINE config 3 (256 -> 64 on 176 x 176, batch 2): the two forms of the same gradient side by side."""
    from ssm_amd import backward as Bk
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(9)
    B, H, W, cin, cout = 2, 176, 176, 256, 64
    x = torch.randn(B, cin, H, W, generator=g)
    dz = torch.randn(B, cout, H, W, generator=g) * 1e-3
    xp, dzp = hb.Planes(B, cin, H, W, dev).load(x.to(dev)), hb.Planes(B, cout, H, W, dev).load(dz.to(dev))
    direct = Bk.wgrad(xp, dzp, torch.empty(cout, cin, 3, 3, device=dev), 3)
    du, dw = torch.zeros(16, cout, cin, device=dev), torch.zeros(cout, cin, 3, 3, device=dev)
    hb.wgrad_wino(xp.view(), dzp.view(), du, None, B, cin, cout, H, W, cin, 0)
    hb.WgradWinoFinish([(du, dw)], dev).run()
    assert rel_err(dw, direct) < 5e-5, "%.3e" % rel_err(dw, direct)


def test_wgrad_wino_refuses_what_it_cannot_run(dev):
    from ssm_amd import hipbind as hb
    assert hb.wgrad_wino_supported(64, 64, 44, 44, 3) and not hb.wgrad_wino_supported(64, 64, 22, 22, 3)
    assert not hb.wgrad_wino_supported(64, 64, 44, 44, 5) and not hb.wgrad_wino_supported(16, 32, 352, 352, 3)
    x = torch.zeros(1, 32, 44, 44, device=dev)           # plain NCHW: no zero frame
    du = torch.zeros(16, 32, 32, device=dev)
    with pytest.raises(RuntimeError, match="padded-plane"):
        hb.wgrad_wino(hb.view_of(x), hb.view_of(x), du, None, 1, 32, 32, 44, 44, 32, 0)
