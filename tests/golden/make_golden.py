#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE itself (build container only).

    python tests/golden/make_golden.py            # needs /root/reference

The reference's Python is imported from /root/reference/scripts with the three
stubs SURVEY.md 8c names (empty `models.CLSTM.*`, a stub `torchvision`, a no-op
`Tensor.cuda`).  Only DATA is written here (inputs + the reference's outputs,
.npz); no reference source travels.  Weights come from
`ssm_amd.weights.synthetic_state_dict` (pure function of name/index) and are
loaded through the reference's own `load_state_dict`, so the fixtures do not
store them.
"""
import configparser
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd"))
from ssm_amd.weights import normalize_and_pad, synthetic_frames_u8, synthetic_state_dict  # noqa: E402


def import_reference():
    for name, attrs in (("models.CLSTM", {}), ("models.CLSTM.convgru", {"ConvBGRU": None}),
                        ("models.CLSTM.convlstm", {"ConvBLSTM": None})):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        if name == "models.CLSTM":
            m.__path__ = []
        sys.modules[name] = m
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")

    class _VGG(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.features = torch.nn.Sequential(*[torch.nn.Identity() for _ in range(31)])

    tvm.vgg16 = lambda pretrained=False: _VGG()
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, os.path.join(REF, "scripts"))
    from models import flow_interpolation, layers, superslomo_r, unetflow  # noqa
    return layers, unetflow, superslomo_r


def make_cfg(cross_skip=True):
    cfg = configparser.RawConfigParser()
    cfg.read(os.path.join(REF, "configs", "superslomo_original.ini"))
    cfg.set("STAGE1", "LOADPREV", "FALSE")
    cfg.set("STAGE2", "LOADPREV", "FALSE")
    cfg.set("STAGE2", "CROSS_SKIP", "TRUE" if cross_skip else "FALSE")
    return cfg


def npy(t):
    return t.detach().cpu().numpy().astype(np.float32)


GRAD_KEYS = ("conv1a.0.weight", "conv6.1.0.weight", "conv11b.0.bias", "final_conv.weight", "final_conv.bias")


def make_grads(ssm):
    """Training-mode forward + BACKWARD of the imported reference at 64x64: `losses.mean(0)[0].backward()` as
    Trainer does (main.py:138-141 of the reference), gradients of a few parameters of both stages -> train_grads_64.npz.  Same
    inputs as train_forward_64.npz; per-sample t differ so the t-dependent terms of the adjoint are exercised.  LAMBDA_P = 0:
    the perceptual term needs the pretrained VGG16 the environment cannot fetch, so the fixture pins the
    reconstruction + four warp terms (lambda_r = 60, lambda_w = 10 of the shipped ini)."""
    cfgT = make_cfg(True)
    cfgT.set("TRAIN", "LAMBDA_P", "0")
    cfgT.set("STAGE1", "FREEZE", "FALSE")
    cfgT.set("STAGE2", "FREEZE", "FALSE")
    with torch.enable_grad():
        fmT = ssm.FullModel(cfgT)
        fmT.stage1_model.load_state_dict(synthetic_state_dict(1, True))
        fmT.stage2_model.load_state_dict(synthetic_state_dict(2, True))
        fmT.train()
        u8t = torch.stack([synthetic_frames_u8(3, 64, 64, seed=60), synthetic_frames_u8(3, 64, 64, seed=61)])
        clip = torch.cat([normalize_and_pad(u8t[0]), normalize_and_pad(u8t[1])], 0)
        xin, tgt = clip[:, [0, 2]], clip[:, 1:2]
        tt = torch.tensor([0.375, 0.625]).view(2, 1, 1, 1, 1)
        img, losses = fmT(xin, tt, tgt, None, False)
        losses.mean(0)[0].backward()
    out = {"u8": u8t.numpy(), "t": npy(tt), "img": npy(img), "losses": npy(losses)}
    for st, mod in ((1, fmT.stage1_model), (2, fmT.stage2_model)):
        params = dict(mod.named_parameters())
        for k in GRAD_KEYS:
            g = params[k].grad
            out["s%d.%s" % (st, k)] = npy(g[::8, ::8] if k == "conv6.1.0.weight" else g)      # 512x512x3x3: every 8th (cout, cin)
        # every parameter's gradient, condensed: sum and abs-sum (96 tensors x 2 numbers)
        out["s%d.names" % st] = np.array(sorted(params))
        out["s%d.sum" % st] = np.array([params[k].grad.double().sum().item() for k in sorted(params)])
        out["s%d.abssum" % st] = np.array([params[k].grad.double().abs().sum().item() for k in sorted(params)])
    np.savez_compressed(os.path.join(HERE, "train_grads_64.npz"), **out)
    print("training-mode losses [B,4]:", losses.detach())
    print("train_grads_64.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "train_grads_64.npz")) / 1024))


def main():
    torch.set_grad_enabled(False)
    torch.manual_seed(1234)
    layers, unetflow, ssm = import_reference()
    if "--grads-only" in sys.argv:
        return make_grads(ssm)
    rng = np.random.RandomState(7)

    # ---------------- per-op fixtures -------------------------------------
    ops = {}
    for k, cin, cout in ((7, 6, 32), (5, 32, 64), (3, 64, 32), (3, 32, 5)):
        x = torch.from_numpy(rng.randn(1, cin, 16, 24).astype(np.float32))
        m = layers.conv(cin, cout, kernel_size=k, padding=(k - 1) // 2)
        w = torch.from_numpy((rng.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32))
        b = torch.from_numpy((rng.randn(cout) * 0.1).astype(np.float32))
        m[0].weight.copy_(w)
        m[0].bias.copy_(b)
        tag = "conv_k%d_c%d_n%d" % (k, cin, cout)
        ops[tag + "_x"], ops[tag + "_w"], ops[tag + "_b"] = npy(x), npy(w), npy(b)
        ops[tag + "_y"] = npy(m(x.clone()))
        ops[tag + "_ylin"] = npy(m[0](x.clone()))          # no activation (final_conv form)
    x = torch.from_numpy(rng.randn(2, 5, 12, 20).astype(np.float32))
    ops["pool_x"], ops["pool_y"] = npy(x), npy(layers.avg_pool(2, None, 0)(x))
    a = torch.from_numpy(rng.randn(2, 3, 6, 10).astype(np.float32))
    b2 = torch.from_numpy(rng.randn(2, 2, 6, 10).astype(np.float32))
    cat = torch.cat([a, b2], 1)
    ops["up_a"], ops["up_b"] = npy(a), npy(b2)
    ops["up_y"] = npy(torch.nn.functional.upsample(cat, size=(12, 20), mode="bilinear"))
    # warp: random flows (some far out of bounds), integer and half-pixel shifts
    img = torch.from_numpy(rng.randn(2, 3, 17, 23).astype(np.float32))
    flo = torch.from_numpy((rng.randn(2, 2, 17, 23) * 4).astype(np.float32))
    flo[0, :, 0:3, :] = 2.0
    flo[0, :, 3:6, :] = -1.5
    flo[1, 0, :, 0:4] = 40.0
    flo[1, 1, :, 4:8] = -30.0
    flo[1, :, 8:10, :] = 0.0
    ops["warp_img"], ops["warp_flo"], ops["warp_y"] = npy(img), npy(flo), npy(layers.warp(img, flo))

    cfg = make_cfg(True)
    s2 = unetflow.get_model(None, 16, 5, True, stage=2, cfg=cfg)
    img6 = torch.from_numpy(rng.randn(2, 6, 20, 28).astype(np.float32))
    flow4 = torch.from_numpy((rng.randn(2, 4, 20, 28) * 3).astype(np.float32))
    out5 = torch.from_numpy((rng.randn(2, 5, 20, 28) * 1.5).astype(np.float32))
    ops["fi_img6"], ops["fi_flow4"], ops["fi_out5"] = npy(img6), npy(flow4), npy(out5)
    for i, tv in enumerate((0.125, 0.5, 0.875)):
        t = torch.full((2, 1, 1, 1), tv)
        in16 = s2.compute_inputs(img6, flow4, t)
        ops["fi_in16_%d" % i] = npy(in16)
        ops["fi_img_%d" % i] = npy(s2.compute_output_image(img6, in16, out5, t))
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **ops)

    # ---------------- stage fixtures (64x64) -------------------------------
    st = {}
    x6 = normalize_and_pad(synthetic_frames_u8(2, 64, 64, seed=42))  # [1,2,3,64,64]
    pair = torch.cat([x6[:, 0], x6[:, 1]], 1)
    st["pair"] = npy(pair)
    s1 = unetflow.get_model(None, 6, 4, True, stage=1, cfg=cfg)
    s1.load_state_dict(synthetic_state_dict(1, True))
    s1.eval()
    (enc, flow), = s1(pair[:, None])
    st["s1_conv6"], st["s1_flow"] = npy(enc), npy(flow)
    for cross in (True, False):
        c = make_cfg(cross)
        m2 = unetflow.get_model(None, 16, 5, cross, stage=2, cfg=c)
        m2.load_state_dict(synthetic_state_dict(2, cross))
        m2.eval()
        t = torch.full((1, 1, 1, 1), 0.375)
        in16 = m2.compute_inputs(pair, flow, t)
        (o5,) = m2(in16[:, None], [enc] if cross else None)
        st["s2_in16"] = npy(in16)
        st["s2_out5_cross%d" % int(cross)] = npy(o5)
    np.savez_compressed(os.path.join(HERE, "stages_64.npz"), **st)

    # ---------------- FullModel fixtures -----------------------------------
    def build_full(cross=True):
        fm = ssm.FullModel(make_cfg(cross))
        fm.stage1_model.load_state_dict(synthetic_state_dict(1, cross))
        fm.stage2_model.load_state_dict(synthetic_state_dict(2, cross))
        return fm.eval()

    fm = build_full(True)
    names = ("F01", "F10", "Ft1e", "Ft0e", "Ft1", "Ft0", "V0")
    full = {}
    # 64x64, B=1, all 7 t
    u8 = synthetic_frames_u8(2, 64, 64, seed=43)
    x = normalize_and_pad(u8)
    full["a_u8"] = u8.numpy()
    for i in range(1, 8):
        t = torch.full((1, 1, 1, 1, 1), i / 8.0)
        img, inter = fm(x, t, inference_mode=True)
        full["a_img_t%d" % i] = npy(img)
        if i == 3:
            for n, v in zip(names, inter):
                full["a_%s_t3" % n] = npy(v)
    # 64x96 (non-square), B=2, per-sample t
    u8b = torch.stack([synthetic_frames_u8(2, 64, 96, seed=44), synthetic_frames_u8(2, 64, 96, seed=45)])
    xb = torch.cat([normalize_and_pad(u8b[0]), normalize_and_pad(u8b[1])], 0)
    tb = torch.tensor([0.25, 0.75]).view(2, 1, 1, 1, 1)
    img, _ = fm(xb, tb, inference_mode=True)
    full["b_u8"], full["b_t"], full["b_img"] = u8b.numpy(), npy(tb), npy(img)
    # 90x120 unpadded -> padded 96x128, B=1, t in {1/8, 4/8, 7/8}
    u8c = synthetic_frames_u8(2, 90, 120, seed=46)
    xc = normalize_and_pad(u8c)
    full["c_u8"] = u8c.numpy()
    for i in (1, 4, 7):
        img, _ = fm(xc, torch.full((1, 1, 1, 1, 1), i / 8.0), inference_mode=True)
        full["c_img_t%d" % i] = npy(img)
    np.savez_compressed(os.path.join(HERE, "fullmodel_small.npz"), **full)

    # ---------------- config 1: 256x256, t=0.5 ------------------------------
    c1 = {}
    u8 = synthetic_frames_u8(2, 256, 256, seed=42)
    x = normalize_and_pad(u8)
    img, inter = fm(x, torch.full((1, 1, 1, 1, 1), 0.5), inference_mode=True)
    c1["u8"] = u8.numpy()
    c1["img_sub4"] = npy(img[:, :, ::4, ::4])          # every 4th pixel
    c1["img_center64"] = npy(img[:, :, 96:160, 96:160])
    c1["img_sum"] = np.float64(img.double().sum().item())
    c1["img_abssum"] = np.float64(img.double().abs().sum().item())
    c1["F01_sub4"] = npy(inter[0][:, :, ::4, ::4])
    c1["flow_absmax"] = np.float32(max(inter[0].abs().max().item(), inter[1].abs().max().item()))
    np.savez_compressed(os.path.join(HERE, "config1_256.npz"), **c1)

    # ---------------- training-mode forward: [B,4] losses (a-13), FREEZE=FALSE so every warp term is on ------
    # The perceptual term runs on the torchvision STUB (identity features), i.e. lambda_p * MSE(pred, target);
    # the HIP-side test reproduces it with an identity feature extractor.  Forward values only (no backward).
    cfgT = make_cfg(True)
    cfgT.set("STAGE1", "FREEZE", "FALSE")
    cfgT.set("STAGE2", "FREEZE", "FALSE")
    fmT = ssm.FullModel(cfgT)
    fmT.stage1_model.load_state_dict(synthetic_state_dict(1, True))
    fmT.stage2_model.load_state_dict(synthetic_state_dict(2, True))
    u8t = torch.stack([synthetic_frames_u8(3, 64, 64, seed=60), synthetic_frames_u8(3, 64, 64, seed=61)])   # [2,3,3,64,64]
    clip = torch.cat([normalize_and_pad(u8t[0]), normalize_and_pad(u8t[1])], 0)                           # [2,3,3,64,64]
    xin = clip[:, [0, 2]]                         # frames 0 and 2 are the inputs, frame 1 is the target
    tgt = clip[:, 1:2]
    tt = torch.tensor([0.5, 0.5]).view(2, 1, 1, 1, 1)
    img, losses = fmT(xin, tt, tgt, None, False)
    np.savez_compressed(os.path.join(HERE, "train_forward_64.npz"), u8=u8t.numpy(), t=npy(tt), img=npy(img),
                        losses=npy(losses))
    print("training-mode losses [B,4]:", losses)
    make_grads(ssm)

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-24s %8.1f KB" % (f, os.path.getsize(os.path.join(HERE, f)) / 1024))
    print("flow |max| @256: %.3f px ; V0 range: %.3f..%.3f" % (
        float(c1["flow_absmax"]), inter[6].min().item(), inter[6].max().item()))


if __name__ == "__main__":
    main()
