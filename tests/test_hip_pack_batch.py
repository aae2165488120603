"""The one-launch repack of a U-Net's fp32 filters (ssm_pack32_weights_batch, csrc/ssm_pack.hip) against the per-layer pack entry
points it replaces in the training step - every form (direct, F(2x2,3x3), F(2,7) / F(4,5) along x, F(4x4,3x3)), forward and
data-gradient (transposed + flipped, read straight from the forward OIHW parameter; the adjoint of layers.conv,
scripts/models/layers.py:21-33).  Bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def test_batched_fp32_pack_matches_per_layer_packs(dev):
    from ssm_amd import hipbind as hb
    from ssm_amd.backward import transposed_filter
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 64, 96
    cases = [(hb.PackedConv, 7, 6, 32), (hb.PackedConv, 5, 32, 64), (hb.PackedConv, 3, 64, 40), (hb.PackedConv, 3, 32, 5),
             (hb.PackedWino, 3, 16, 64), (hb.PackedWino, 3, 64, 32), (hb.PackedWino1d, 7, 6, 32), (hb.PackedWino1d, 7, 32, 32),
             (hb.PackedWino1d, 5, 32, 64), (hb.PackedWino4, 3, 8, 32), (hb.PackedWino4, 3, 64, 96),
             # whole tiles of BN couts x 16 input channels: the tiled kernel (ssm_pack32_wino_tiles_batch), forward and transposed
             (hb.PackedConv, 3, 64, 64), (hb.PackedConv, 3, 128, 32), (hb.PackedWino, 3, 128, 128), (hb.PackedWino, 3, 48, 96),
             (hb.PackedWino4, 3, 128, 64), (hb.PackedWino4, 3, 48, 160)]          # r6: F(4x4,3x3) jobs of whole tiles too
    entries, want = [], []
    for cls, k, cin, cout in cases:
        w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
        b = (torch.randn(cout, generator=g) * 0.1).to(dev)
        ref = cls(w, b, B, H, W)                                # per-layer pack: the reference
        tgt = cls(torch.zeros_like(w), torch.zeros_like(b), B, H, W)
        tgt.w.fill_(7.0)
        tgt.b.fill_(7.0)
        entries.append((tgt, w, b, False))
        want.append((ref, tgt, "%s k%d %d->%d" % (cls.__name__, k, cin, cout)))
        # data-gradient filter of the same parameter, where that form supports the transposed shape (Cout' = cin)
        ok = {hb.PackedConv: True, hb.PackedWino: cout % 8 == 0, hb.PackedWino1d: cin % 32 == 0 and (k == 7 or cin % 64 == 0 or cin == 32),
              hb.PackedWino4: cin % 32 == 0 and cout % 4 == 0}[cls]
        if ok:
            wt = transposed_filter(w)
            reft = cls(wt, torch.zeros(cin, device=dev), B, H, W)
            tgtt = cls(torch.zeros_like(wt), torch.zeros(cin, device=dev), B, H, W)
            tgtt.w.fill_(7.0)
            tgtt.b.fill_(7.0)
            entries.append((tgtt, w, None, True))
            want.append((reft, tgtt, "%s k%d %d->%d transposed" % (cls.__name__, k, cin, cout)))
    batch = hb.PackBatch32(entries, dev)
    assert batch.tiles is not None and batch.tiles[1] >= 10 and batch.n >= 8, "the cases of this test no longer cover both pack kernels: %d tiled, %d element-wise" % (batch.tiles[1], batch.n)
    assert sum(1 for e in batch.keep_tiled if e[0].algo == "wino4") >= 4 and any(e[0].algo == "wino4" for e in [(k[0],) for k in batch.keep])
    batch.run()
    torch.cuda.synchronize()
    for ref, tgt, tag in want:
        assert torch.equal(ref.w, tgt.w), "%s: packed filter differs (max %.3e)" % (tag, float((ref.w - tgt.w).abs().max()))
        assert torch.equal(ref.b, tgt.b), "%s: packed bias differs" % tag
