"""Pin the CPU oracle (oracle/ssm_oracle.py) to fixtures produced by the
reference itself (tests/golden/make_golden.py).  No GPU."""
import numpy as np
import torch

from oracle import ssm_oracle as O
from ssm_amd.weights import normalize_and_pad, synthetic_state_dict

TOL = 2e-5   # oracle vs reference, both fp32 CPU; differences are reassociation only


def T(a):
    return torch.from_numpy(np.asarray(a))


def maxdiff(a, b):
    return float((a - T(b)).abs().max())


def test_conv_ops(golden):
    g = golden("ops")
    for tag in ("conv_k7_c6_n32", "conv_k5_c32_n64", "conv_k3_c64_n32", "conv_k3_c32_n5"):
        x, w, b = T(g[tag + "_x"]), T(g[tag + "_w"]), T(g[tag + "_b"])
        assert maxdiff(O.conv2d_lrelu(x, w, b), g[tag + "_y"]) < TOL
        assert maxdiff(O.conv2d(x, w, b), g[tag + "_ylin"]) < TOL


def test_pool_upsample(golden):
    g = golden("ops")
    assert maxdiff(O.avg_pool2(T(g["pool_x"])), g["pool_y"]) < 1e-6
    cat = torch.cat([T(g["up_a"]), T(g["up_b"])], 1)
    assert maxdiff(O.upsample2x_bilinear(cat), g["up_y"]) < 1e-6


def test_warp(golden):
    g = golden("ops")
    assert maxdiff(O.warp(T(g["warp_img"]), T(g["warp_flo"])), g["warp_y"]) < 1e-5


def test_flow_interp_inputs_and_synthesis(golden):
    g = golden("ops")
    img6, flow4, out5 = T(g["fi_img6"]), T(g["fi_flow4"]), T(g["fi_out5"])
    for i, tv in enumerate((0.125, 0.5, 0.875)):
        t = torch.full((2, 1, 1, 1), tv)
        in16 = O.flow_interp_inputs(img6, flow4, t)
        assert maxdiff(in16, g["fi_in16_%d" % i]) < 1e-5
        img = O.synthesize(img6, T(g["fi_in16_%d" % i]), out5, t)
        assert maxdiff(img, g["fi_img_%d" % i]) < 5e-5


def test_stages(golden):
    g = golden("stages_64")
    pair = T(g["pair"])
    p1 = synthetic_state_dict(1, True)
    c6, flow = O.stage1(p1, pair)
    assert maxdiff(c6, g["s1_conv6"]) < TOL
    assert maxdiff(flow, g["s1_flow"]) < TOL
    for cross in (True, False):
        p2 = synthetic_state_dict(2, cross)
        o5 = O.stage2(p2, T(g["s2_in16"]), T(g["s1_conv6"]) if cross else None)
        assert maxdiff(o5, g["s2_out5_cross%d" % int(cross)]) < TOL


def test_full_model(golden):
    g = golden("fullmodel_small")
    p1, p2 = synthetic_state_dict(1, True), synthetic_state_dict(2, True)
    x = normalize_and_pad(T(g["a_u8"]))
    pair = torch.cat([x[:, 0], x[:, 1]], 1)
    outs = O.interpolate_pair(p1, p2, pair, [i / 8.0 for i in range(1, 8)])
    for i, o in enumerate(outs, 1):
        assert maxdiff(o, g["a_img_t%d" % i]) < 1e-4
    img, inter = O.full_model_infer(p1, p2, x, torch.full((1, 1, 1, 1, 1), 3 / 8.0))
    assert maxdiff(img, g["a_img_t3"]) < 1e-4
    for n, v in zip(("F01", "F10", "Ft1e", "Ft0e", "Ft1", "Ft0", "V0"), inter):
        assert maxdiff(v, g["a_%s_t3" % n]) < 1e-4, n
    # B=2, per-sample t, non-square
    u8b = T(g["b_u8"])
    xb = torch.cat([normalize_and_pad(u8b[0]), normalize_and_pad(u8b[1])], 0)
    img, _ = O.full_model_infer(p1, p2, xb, T(g["b_t"]))
    assert maxdiff(img, g["b_img"]) < 1e-4
    # unpadded 90x120 -> 96x128
    xc = normalize_and_pad(T(g["c_u8"]))
    pc = torch.cat([xc[:, 0], xc[:, 1]], 1)
    for i, o in zip((1, 4, 7), O.interpolate_pair(p1, p2, pc, [1 / 8.0, 4 / 8.0, 7 / 8.0], hoist=False)):
        assert maxdiff(o, g["c_img_t%d" % i]) < 1e-4


def test_config1_256(golden):
    g = golden("config1_256")
    p1, p2 = synthetic_state_dict(1, True), synthetic_state_dict(2, True)
    x = normalize_and_pad(T(g["u8"]))
    img, inter = O.full_model_infer(p1, p2, x, torch.full((1, 1, 1, 1, 1), 0.5))
    assert maxdiff(img[:, :, ::4, ::4], g["img_sub4"]) < 2e-4
    assert maxdiff(img[:, :, 96:160, 96:160], g["img_center64"]) < 2e-4
    assert abs(img.double().abs().sum().item() - float(g["img_abssum"])) < 1e-5 * float(g["img_abssum"])
    assert maxdiff(inter[0][:, :, ::4, ::4], g["F01_sub4"]) < 2e-4


def test_training_backward_gradients(golden):
    """The oracle's autograd (oracle.training_loss, `losses.mean(0)[0].backward()`) against gradients produced by the imported
    reference's own FullModel(..., inference_mode=False) + backward at 64x64 (tests/golden/make_golden.py::make_grads; reference
    scripts/models/losses.py:196-249, superslomo_r.py:240-243): the [B,4] losses, five parameter gradients of each stage in full
    (conv6.1.0.weight every 8th channel pair) and the sum / abs-sum of every one of the 96 gradients."""
    g = golden("train_grads_64")
    u8 = T(g["u8"])
    clip = torch.cat([normalize_and_pad(u8[0]), normalize_and_pad(u8[1])], 0)
    xin, tgt = clip[:, [0, 2]], clip[:, 1]
    img6 = torch.cat([xin[:, 0], xin[:, 1]], 1)
    p1 = {k: v.clone().requires_grad_() for k, v in synthetic_state_dict(1, True).items()}
    p2 = {k: v.clone().requires_grad_() for k, v in synthetic_state_dict(2, True).items()}
    losses, pred = O.training_loss(p1, p2, img6, T(g["t"]).view(2, 1, 1, 1), tgt, 60.0, 10.0)
    losses.mean(0)[0].backward()
    assert maxdiff(pred.detach(), g["img"]) < 1e-4
    want = T(g["losses"])
    assert float(((losses.detach() - want).abs() / want.abs().clamp_min(1.0)).max()) < 1e-5
    for st, p in ((1, p1), (2, p2)):
        for k in ("conv1a.0.weight", "conv6.1.0.weight", "conv11b.0.bias", "final_conv.weight", "final_conv.bias"):
            got = p[k].grad[::8, ::8] if k == "conv6.1.0.weight" else p[k].grad
            w = T(g["s%d.%s" % (st, k)])
            assert float((got - w).abs().max() / w.abs().max()) < 1e-5, (st, k)      # measured: <= 6e-7
        names = [str(n) for n in g["s%d.names" % st]]
        assert names == sorted(p)
        for n, s_want, a_want in zip(names, g["s%d.sum" % st], g["s%d.abssum" % st]):
            gr = p[n].grad.double()
            assert abs(gr.abs().sum().item() - a_want) < 1e-4 * a_want, (st, n)
            assert abs(gr.sum().item() - s_want) < 1e-4 * a_want, (st, n)
