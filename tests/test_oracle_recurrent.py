"""CPU checks of the oracle's recurrent bottleneck (ConvBLSTM / ConvBGRU restatement, PARITY UNPINNED: the
reference's submodule is empty, so there is no golden vector - see oracle/ssm_oracle.py).  What can be pinned
without the source: hand-computed known answers of the published cell equations, the bidirectional time
alignment implied by the reference's call `conv6(x_fwd, x_rev=reversed)` (flow_computation.py:208-211), the output
shape the reference asserts (:313-314), and the weight ABI of the mirror modules."""
import math

import pytest
import torch

from oracle import ssm_oracle as O


def _lstm_params(bias_vec, hid=2, cin=3):
    p = {}
    for net in ("forward_net", "reverse_net"):
        for l in range(2):
            c = cin if l == 0 else hid
            p["conv6.%s.cell_list.%d.conv.weight" % (net, l)] = torch.zeros(4 * hid, c + hid, 3, 3)
            p["conv6.%s.cell_list.%d.conv.bias" % (net, l)] = bias_vec.clone()
    return p


def test_convlstm_known_answer():
    """Zero filters, biases i=f=o=0, g=+inf-ish: c_t = 0.5 c_{t-1} + 0.5, h_t = 0.5 tanh(c_t) in both layers."""
    hid = 2
    bias = torch.cat([torch.zeros(3 * hid), torch.full((hid,), 20.0)])
    p = _lstm_params(bias, hid)
    xs = [torch.randn(1, 3, 4, 5) for _ in range(3)]
    out = O.unet_bottleneck_recurrent(p, "CLSTM", xs)
    c, want = 0.0, []
    for _ in range(3):
        c = 0.5 * c + 0.5
        want.append(0.5 * math.tanh(c))
    for k in range(3):
        assert out[k].shape == (1, 2 * hid, 4, 5)
        assert torch.allclose(out[k][:, :hid], torch.full((1, hid, 4, 5), want[k]), atol=1e-6)          # forward net: step k
        assert torch.allclose(out[k][:, hid:], torch.full((1, hid, 4, 5), want[2 - k]), atol=1e-6)      # reverse net: step T-1-k


def test_convgru_known_answer():
    """Zero filters: update u = s(b_u), candidate n = tanh(b_c): h_t = (1-u) h_{t-1} + u n."""
    hid, cin = 2, 3
    p = {}
    for net in ("forward_net", "reverse_net"):
        for l in range(2):
            c = cin if l == 0 else hid
            pre = "conv6.%s.cell_list.%d." % (net, l)
            p[pre + "conv_gates.weight"] = torch.zeros(2 * hid, c + hid, 3, 3)
            p[pre + "conv_gates.bias"] = torch.cat([torch.full((hid,), -1.0), torch.full((hid,), 0.5)])
            p[pre + "conv_can.weight"] = torch.zeros(hid, c + hid, 3, 3)
            p[pre + "conv_can.bias"] = torch.full((hid,), 0.3)
    out = O.unet_bottleneck_recurrent(p, "CGRU", [torch.randn(2, cin, 3, 3) for _ in range(4)])
    u, n, h = 1 / (1 + math.exp(-0.5)), math.tanh(0.3), 0.0
    for k in range(4):
        h = (1 - u) * h + u * n
        assert torch.allclose(out[k][:, :hid], torch.full((2, hid, 3, 3), h), atol=1e-6)
        assert torch.allclose(out[3 - k][:, hid:], torch.full((2, hid, 3, 3), h), atol=1e-6)


@pytest.mark.parametrize("kind", ["CLSTM", "CGRU"])
def test_bidirectional_causality(kind):
    """Forward half of slot k depends on windows <= k only, reverse half on windows >= k only."""
    from ssm_amd.weights import synthetic_state_dict
    p = {k: v for k, v in synthetic_state_dict(1, bottleneck=kind).items() if k.startswith("conv6.")}
    torch.manual_seed(0)
    xs = [torch.randn(1, 512, 2, 3) * 0.3 for _ in range(3)]
    base = O.unet_bottleneck_recurrent(p, kind, xs)
    assert all(o.shape == (1, 512, 2, 3) for o in base)                 # what flow_computation.py:313-314 asserts
    for j in range(3):
        ys = [x.clone() for x in xs]
        ys[j] += 0.5
        pert = O.unet_bottleneck_recurrent(p, kind, ys)
        for k in range(3):
            dfwd = float((pert[k][:, :256] - base[k][:, :256]).abs().max())
            drev = float((pert[k][:, 256:] - base[k][:, 256:]).abs().max())
            assert (dfwd > 1e-4) == (j <= k), (j, k, dfwd)
            assert (drev > 1e-4) == (j >= k), (j, k, drev)


def test_windows_model_equals_pairwise_model_for_conv_bottleneck():
    from ssm_amd.weights import synthetic_frames, synthetic_state_dict
    p1, p2 = synthetic_state_dict(1), synthetic_state_dict(2)
    x = synthetic_frames(4, 64, 64)
    t = torch.tensor([0.25, 0.5, 0.75]).reshape(1, 3, 1, 1, 1)
    a = O.full_model_infer(p1, p2, x, t)
    b = O.full_model_infer_windows(p1, p2, x, t)
    assert float((a[0] - b[0]).abs().max()) == 0.0
    assert all(float((u - v).abs().max()) == 0.0 for u, v in zip(a[1], b[1]))


@pytest.mark.parametrize("kind,n1,n2", [("CLSTM", 37114084, 39489349), ("CGRU", 31214820, 33590085)])
def test_recurrent_mirror_weight_abi(kind, n1, n2):
    """Mirror modules carry the published key names; strict load of the synthetic dicts (no GPU needed)."""
    from models.superslomo_r import FullModel
    from ssm_amd.config import load_config, synthetic_weight_overrides
    from ssm_amd.weights import synthetic_state_dict
    ov = synthetic_weight_overrides()
    ov[("STAGE1", "BOTTLENECK")] = ov[("STAGE2", "BOTTLENECK")] = kind
    m = FullModel(load_config("superslomo_recurrent.ini", ov))
    assert m.recurrent and m.cfg.getint("TRAIN", "N_FRAMES") == 4
    m.stage1_model.load_state_dict(synthetic_state_dict(1, bottleneck=kind))       # strict
    m.stage2_model.load_state_dict(synthetic_state_dict(2, bottleneck=kind))
    assert sum(p.numel() for p in m.stage1_model.parameters()) == n1
    assert sum(p.numel() for p in m.stage2_model.parameters()) == n2
    keys = [k for k in m.stage1_model.state_dict() if k.startswith("conv6.")]
    leaf = "conv.weight" if kind == "CLSTM" else "conv_gates.weight"
    assert "conv6.forward_net.cell_list.0." + leaf in keys and "conv6.reverse_net.cell_list.1." + leaf in keys
    with pytest.raises(RuntimeError):                 # no CPU fallback for the recurrent path either
        m(torch.zeros(1, 4, 3, 32, 32), torch.full((1, 3, 1, 1, 1), 0.5))


def test_oracle_vgg16_matches_torchvision_layout():
    """The oracle's phi against an nn.Sequential built the way torchvision builds vgg16.features (cfg 'D':
    64,64,M,128,128,M,256,256,256,M,512,512,512,M,...; conv3x3 pad 1 + ReLU(inplace)) and cut at [:23] like
    losses.py:34 - pins the layer indices / key names of the state dict the HIP extractor consumes."""
    import torch.nn as nn
    from ssm_amd.perceptual import VGG16_CONV4_3, synthetic_vgg_state_dict
    layers, cin = [], 3
    for v in [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    feats = nn.Sequential(*layers)[:23]
    assert isinstance(feats[22], nn.ReLU) and isinstance(feats[21], nn.Conv2d)
    sd = synthetic_vgg_state_dict()
    feats.load_state_dict({k[len("features."):]: v for k, v in sd.items()})          # strict: same indices and shapes
    assert [i[0] for i in VGG16_CONV4_3 if i != "M"] == [i for i, m in enumerate(feats) if isinstance(m, nn.Conv2d)]
    x = torch.randn(2, 3, 32, 40)
    with torch.no_grad():
        want = feats(x)
    got = O.vgg16_conv4_3(sd, x)
    assert got.shape == (2, 512, 4, 5) and float((got - want).abs().max()) < 1e-5
    assert float(got.abs().mean()) > 1e-2               # the synthetic weights keep the features alive
    val = O.perceptual_loss(sd, x, x + 0.1)
    assert val.shape == (2,) and bool((val > 0).all())
