"""CPU oracle for the Super SloMo frame-pair -> intermediate-frame path.

TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product: only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it.  The product path (`superslomo-videointerpolation-pytorch_amd/`)
runs hand-written HIP kernels and fails loudly when they are missing.

It is a restatement, in plain PyTorch CPU fp32 ops and explicit index
arithmetic, of what the reference computes on the hot path.  Each function
cites the reference file:line it follows (paths relative to the reference
repository root).  Parity is PINNED: `tests/golden/make_golden.py` imports the
reference itself (in the build container) and commits its outputs as fixtures;
`tests/test_oracle_golden.py` holds this file to those fixtures.

PARITY UNPINNED for one part (stated in DESIGN.md): the ConvBLSTM/ConvBGRU
bottleneck (section "recurrent bottleneck" below).  Its source is an un-vendored
submodule of the reference (.gitmodules:1-3 -> SreenivasVRao/ConvGRU-ConvLSTM-PyTorch,
pinned commit not recorded, directory empty), so no golden vector of the
reference can be produced for it; the section restates that package's published
cell equations and is anchored on the reference's call sites only.  The
pretrained-VGG perceptual loss is unpinned too (weights need the network).
"""

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # scripts/models/layers.py:32


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def conv2d(x, w, b):
    """Plain stride-1 'same' cross-correlation with zero padding and bias.
    scripts/models/layers.py:22-31 (nn.Conv2d, padding=(k-1)/2, dilation 1);
    final_conv: scripts/models/flow_computation.py:145-153."""
    k = w.shape[-1]
    return F.conv2d(x, w, b, stride=1, padding=(k - 1) // 2)


def conv2d_lrelu(x, w, b):
    """conv + LeakyReLU(0.1).  scripts/models/layers.py:21-33."""
    y = conv2d(x, w, b)
    return torch.where(y >= 0, y, y * LRELU_SLOPE)


def avg_pool2(x):
    """2x2 mean, stride 2, no padding.  scripts/models/layers.py:60-63."""
    B, C, H, W = x.shape
    v = x.reshape(B, C, H // 2, 2, W // 2, 2)
    return (v[:, :, :, 0, :, 0] + v[:, :, :, 0, :, 1] + v[:, :, :, 1, :, 0] + v[:, :, :, 1, :, 1]) * 0.25


def _up2_axis(x, dim):
    """Half-pixel (align_corners=False) x2 linear resize along `dim`:
    out[2i] = .25 x[i-1] + .75 x[i], out[2i+1] = .75 x[i] + .25 x[i+1], edge-clamped."""
    n = x.shape[dim]
    idx = torch.arange(n)
    lo = x.index_select(dim, (idx - 1).clamp(min=0))
    hi = x.index_select(dim, (idx + 1).clamp(max=n - 1))
    even = 0.25 * lo + 0.75 * x
    odd = 0.75 * x + 0.25 * hi
    out = torch.stack([even, odd], dim=dim + 1)
    shape = list(x.shape)
    shape[dim] = 2 * n
    return out.reshape(shape)


def upsample2x_bilinear(x):
    """F.upsample(x, size=(2h,2w), mode='bilinear') == align_corners=False.
    scripts/models/flow_computation.py:92-94 (and :103,:113,:124,:135)."""
    return _up2_axis(_up2_axis(x, 2), 3)


def warp(img, flo):
    """Backward warp: sample img at (x+u, y+v), bilinear, zeros outside.
    scripts/models/layers.py:73-120.  The reference normalises the sampling
    grid to [-1,1] (:112-113) and grid_sample(align_corners=True) maps it back;
    both steps are replayed in fp32 so coordinates round identically."""
    B, C, H, W = img.shape
    xx = torch.arange(W, dtype=torch.float32).view(1, 1, W).expand(B, H, W)
    yy = torch.arange(H, dtype=torch.float32).view(1, H, 1).expand(B, H, W)
    gx = 2.0 * (xx + flo[:, 0]) / max(W - 1, 1) - 1.0
    gy = 2.0 * (yy + flo[:, 1]) / max(H - 1, 1) - 1.0
    ix = ((gx + 1.0) / 2.0) * (W - 1)
    iy = ((gy + 1.0) / 2.0) * (H - 1)
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    x1 = x0 + 1
    y1 = y0 + 1
    w_nw = (x1 - ix) * (y1 - iy)
    w_ne = (ix - x0) * (y1 - iy)
    w_sw = (x1 - ix) * (iy - y0)
    w_se = (ix - x0) * (iy - y0)
    flat = img.reshape(B, C, H * W)
    out = torch.zeros_like(img)
    for xs, ys, wt in ((x0, y0, w_nw), (x1, y0, w_ne), (x0, y1, w_sw), (x1, y1, w_se)):
        inb = (xs >= 0) & (xs <= W - 1) & (ys >= 0) & (ys <= H - 1)
        lin = (ys.clamp(0, H - 1) * W + xs.clamp(0, W - 1)).long().view(B, 1, H * W).expand(B, C, H * W)
        val = flat.gather(2, lin).view(B, C, H, W)
        out = out + val * (wt * inb.to(img.dtype)).unsqueeze(1)
    return out


def flow_interp_inputs(img6, flow4, t):
    """Stage-2 input tensor.  scripts/models/flow_interpolation.py:338-372.
    t: [B,1,1,1].  Channel order (ABI): I1, g(I1), Ft1^, Ft0^, g(I0), I0."""
    f01 = flow4[:, 0:2]
    f10 = flow4[:, 2:4]
    ft0 = -(1 - t) * t * f01 + (t ** 2) * f10          # :353
    ft1 = ((1 - t) ** 2) * f01 - t * (1 - t) * f10     # :356
    i0 = img6[:, 0:3]
    i1 = img6[:, 3:6]
    g1 = warp(i1, ft1)
    g0 = warp(i0, ft0)
    return torch.cat([i1, g1, ft1, ft0, g0, i0], dim=1)


def synthesize(img6, in16, out5, t):
    """Visibility-weighted blend of the two warped frames.
    scripts/models/flow_interpolation.py:374-429."""
    v1 = torch.sigmoid(out5[:, 0:1])
    v0 = 1 - v1
    ft1 = in16[:, 6:8] + out5[:, 1:3]
    ft0 = in16[:, 8:10] + out5[:, 3:5]
    i0 = img6[:, 0:3]
    i1 = img6[:, 3:6]
    p0 = v0 * warp(i0, ft0)
    p1 = v1 * warp(i1, ft1)
    num = (1 - t) * p0 + t * p1
    den = (1 - t) * v0 + t * v1
    return num / den


# --------------------------------------------------------------------------
# the two U-Nets (CONV bottleneck)
# --------------------------------------------------------------------------
def _cl(p, name, x):
    return conv2d_lrelu(x, p[name + ".0.weight"], p[name + ".0.bias"])


def unet_encoder(p, x):
    """flow_computation.py:155-204 / flow_interpolation.py:159-208."""
    c1 = _cl(p, "conv1b", _cl(p, "conv1a", x))
    c2 = _cl(p, "conv2b", _cl(p, "conv2a", avg_pool2(c1)))
    c3 = _cl(p, "conv3b", _cl(p, "conv3a", avg_pool2(c2)))
    c4 = _cl(p, "conv4b", _cl(p, "conv4a", avg_pool2(c3)))
    c5 = _cl(p, "conv5b", _cl(p, "conv5a", avg_pool2(c4)))
    return c1, c2, c3, c4, c5, avg_pool2(c5)


def unet_bottleneck_conv(p, x):
    """CONV bottleneck: Sequential(conv, conv).  flow_computation.py:68-71,:213-218."""
    y = conv2d_lrelu(x, p["conv6.0.0.weight"], p["conv6.0.0.bias"])
    return conv2d_lrelu(y, p["conv6.1.0.weight"], p["conv6.1.0.bias"])


def unet_decoder(p, c6, enc, cross=None):
    """flow_computation.py:222-289 / flow_interpolation.py:210-281.
    Skips are concatenated at the LOWER resolution, then the whole concat is
    upsampled (:244-245)."""
    c1, c2, c3, c4, c5, _ = enc
    x = c6 if cross is None else torch.cat([c6, cross], dim=1)
    x = _cl(p, "conv7b", _cl(p, "conv7a", upsample2x_bilinear(x)))
    x = _cl(p, "conv8b", _cl(p, "conv8a", upsample2x_bilinear(torch.cat([x, c5], 1))))
    x = _cl(p, "conv9b", _cl(p, "conv9a", upsample2x_bilinear(torch.cat([x, c4], 1))))
    x = _cl(p, "conv10b", _cl(p, "conv10a", upsample2x_bilinear(torch.cat([x, c3], 1))))
    x = _cl(p, "conv11b", _cl(p, "conv11a", upsample2x_bilinear(torch.cat([x, c2], 1))))
    x = _cl(p, "fuse_conv", torch.cat([x, c1], 1))
    return conv2d(x, p["final_conv.weight"], p["final_conv.bias"])


def stage1(p, img6):
    """FlowComputationModel on one window: returns (conv6_out, flow4).
    flow_computation.py:291-325."""
    enc = unet_encoder(p, img6)
    c6 = unet_bottleneck_conv(p, enc[-1])
    return c6, unet_decoder(p, c6, enc)


def stage2(p, in16, enc1=None):
    """FlowInterpolationModel on one window: returns out5.
    flow_interpolation.py:296-336 (cross-skip cat at :224-231)."""
    enc = unet_encoder(p, in16)
    c6 = unet_bottleneck_conv(p, enc[-1])
    return unet_decoder(p, c6, enc, cross=enc1)


# --------------------------------------------------------------------------
# orchestration
# --------------------------------------------------------------------------
def full_model_infer(p1, p2, image_tensor, t_interp, cross_skip=True):
    """FullModel.forward(inference_mode=True), CONV bottleneck.
    scripts/models/superslomo_r.py:250-293 (+ :90-106, :152-248).
    image_tensor [B,N,3,H,W], t_interp [B,N-1,1,1,1].
    Returns (img_t, (F01, F10, Ft1^, Ft0^, Ft1, Ft0, V0)) of the middle window."""
    Bn, N = image_tensor.shape[:2]
    T = N - 1
    mid = T // 2
    res = None
    for k in range(T):
        img6 = torch.cat([image_tensor[:, k], image_tensor[:, k + 1]], dim=1)
        c6, flow4 = stage1(p1, img6)
        t = t_interp[:, k]
        in16 = flow_interp_inputs(img6, flow4, t)
        out5 = stage2(p2, in16, c6 if cross_skip else None)
        img_t = synthesize(img6, in16, out5, t)
        if k == mid:
            v0 = 1 - torch.sigmoid(out5[:, 0:1])
            res = (img_t, (flow4[:, 0:2], flow4[:, 2:4], in16[:, 6:8], in16[:, 8:10],
                           in16[:, 6:8] + out5[:, 1:3], in16[:, 8:10] + out5[:, 3:5], v0))
    return res


def interpolate_pair(p1, p2, img6, ts, cross_skip=True, hoist=True):
    """One frame pair -> len(ts) intermediates (the eval loop of
    scripts/evaluate_interpolation_results.py:213-244).  hoist=False recomputes
    stage 1 for every t exactly like the reference loop; hoist=True computes it
    once (same numbers, fewer FLOPs)."""
    outs = []
    s1 = stage1(p1, img6) if hoist else None
    for tv in ts:
        c6, flow4 = s1 if hoist else stage1(p1, img6)
        t = torch.full((img6.shape[0], 1, 1, 1), float(tv), dtype=torch.float32)
        in16 = flow_interp_inputs(img6, flow4, t)
        out5 = stage2(p2, in16, c6 if cross_skip else None)
        outs.append(synthesize(img6, in16, out5, t))
    return outs


# --------------------------------------------------------------------------
# recurrent bottleneck (BOTTLENECK = CLSTM | CGRU) - PARITY UNPINNED
# --------------------------------------------------------------------------
# Third-party arithmetic: `ConvBLSTM` / `ConvBGRU` of SreenivasVRao/ConvGRU-ConvLSTM-PyTorch, imported by the
# reference as models.CLSTM.{convlstm,convgru} (flow_computation.py:7-8) from an empty submodule directory.
# Anchors in the reference: constructor arguments (in_channels=512, hidden_channels=512, kernel_size=(3,3),
# num_layers=2, batch_first=True; flow_computation.py:73-88), the call `conv6(x_fwd, x_rev)` with
# x_rev = the time-reversed stack (:208-211) and the asserted result shape [B,T,512,h,w] (:313-314).
# Published algorithm restated here:
#   * ConvLSTM cell: one conv over cat[x, h] -> 4*hidden channels split in the order i, f, o, g;
#     c' = sigmoid(f)*c + sigmoid(i)*tanh(g); h' = sigmoid(o)*tanh(c'); zero initial state.
#   * ConvGRU cell: conv_gates over cat[x, h] -> 2*hidden split (gamma, beta): reset = sigmoid(gamma),
#     update = sigmoid(beta); conv_can over cat[x, reset*h] -> tanh = candidate;
#     h' = (1-update)*h + update*candidate.
#   * Multi-layer: layer l consumes the full output sequence of layer l-1.
#   * Bidirectional: two independent nets of hidden_channels//2 each; the reverse net runs on x_rev, its
#     output sequence is flipped back in time and concatenated after the forward net's on the channel axis.
#   * State-dict keys: conv6.{forward_net,reverse_net}.cell_list.<l>.conv.{weight,bias} (LSTM),
#     ...cell_list.<l>.{conv_gates,conv_can}.{weight,bias} (GRU).
def convlstm_cell(x, h, c, w, b):
    hc = w.shape[0] // 4
    z = conv2d(torch.cat([x, h], dim=1), w, b)
    i, f, o, g = torch.split(z, hc, dim=1)
    c_next = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c_next), c_next


def convgru_cell(x, h, wg, bg, wc, bc):
    hc = wc.shape[0]
    gamma, beta = torch.split(conv2d(torch.cat([x, h], dim=1), wg, bg), hc, dim=1)
    reset, update = torch.sigmoid(gamma), torch.sigmoid(beta)
    cand = torch.tanh(conv2d(torch.cat([x, reset * h], dim=1), wc, bc))
    return (1 - update) * h + update * cand


def _recurrent_net(p, prefix, kind, xs, num_layers=2):
    """One direction: list of T [B,C,h,w] -> list of T outputs of the last layer."""
    seq = xs
    for l in range(num_layers):
        cell = "%scell_list.%d." % (prefix, l)
        if kind == "CLSTM":
            w, b = p[cell + "conv.weight"], p[cell + "conv.bias"]
            hc = w.shape[0] // 4
        else:
            wg, bg = p[cell + "conv_gates.weight"], p[cell + "conv_gates.bias"]
            wc, bc = p[cell + "conv_can.weight"], p[cell + "conv_can.bias"]
            hc = wc.shape[0]
        B, _, hh, ww = seq[0].shape
        h = torch.zeros(B, hc, hh, ww, dtype=seq[0].dtype)
        c = torch.zeros_like(h)
        outs = []
        for x in seq:
            if kind == "CLSTM":
                h, c = convlstm_cell(x, h, c, w, b)
            else:
                h = convgru_cell(x, h, wg, bg, wc, bc)
            outs.append(h)
        seq = outs
    return seq


def unet_bottleneck_recurrent(p, kind, xs_fwd, xs_rev=None, prefix="conv6."):
    """conv6(x_fwd, x_rev) of flow_computation.py:208-211: list of T pool6 tensors -> list of T [B,512,h,w]."""
    assert kind in ("CLSTM", "CGRU")
    xs_rev = list(xs_fwd[::-1]) if xs_rev is None else xs_rev
    yf = _recurrent_net(p, prefix + "forward_net.", kind, list(xs_fwd))
    yr = _recurrent_net(p, prefix + "reverse_net.", kind, list(xs_rev))[::-1]
    return [torch.cat([a, b], dim=1) for a, b in zip(yf, yr)]


def stage_forward(p, xs, bottleneck="CONV", cross=None):
    """A stage model's forward on T windows (flow_computation.py:291-325 / flow_interpolation.py:296-336):
    encoders, the bottleneck over the window sequence, decoders.  xs: list of T inputs; cross: list of T
    stage-1 encodings or None.  Returns (list of conv6 outputs, list of final outputs)."""
    encs = [unet_encoder(p, x) for x in xs]
    if bottleneck == "CONV":
        hs = [unet_bottleneck_conv(p, e[-1]) for e in encs]
    else:
        hs = unet_bottleneck_recurrent(p, bottleneck, [e[-1] for e in encs])
    outs = [unet_decoder(p, h, e, cross=None if cross is None else cross[k]) for k, (h, e) in enumerate(zip(hs, encs))]
    return hs, outs


def full_model_infer_windows(p1, p2, image_tensor, t_interp, cross_skip=True, bottleneck="CONV"):
    """FullModel.forward(inference_mode=True) for any N_FRAMES and bottleneck (superslomo_r.py:152-293):
    stage 1 on all T windows, compute_inputs per window with t_interp[:, k], stage 2 on all windows, the middle
    window's frame and intermediates returned.  Equals full_model_infer for BOTTLENECK=CONV."""
    N = image_tensor.shape[1]
    T = N - 1
    mid = T // 2
    pairs = [torch.cat([image_tensor[:, k], image_tensor[:, k + 1]], dim=1) for k in range(T)]
    enc1, flows = stage_forward(p1, pairs, bottleneck)
    in16 = [flow_interp_inputs(pairs[k], flows[k], t_interp[:, k]) for k in range(T)]
    _, out5 = stage_forward(p2, in16, bottleneck, cross=enc1 if cross_skip else None)
    k = mid
    img_t = synthesize(pairs[k], in16[k], out5[k], t_interp[:, k])
    v0 = 1 - torch.sigmoid(out5[k][:, 0:1])
    return img_t, (flows[k][:, 0:2], flows[k][:, 2:4], in16[k][:, 6:8], in16[k][:, 8:10],
                   in16[k][:, 6:8] + out5[k][:, 1:3], in16[k][:, 8:10] + out5[k][:, 3:5], v0)


# --------------------------------------------------------------------------
# perceptual loss (training only) - architecture pinned, pretrained weights UNPINNED
# --------------------------------------------------------------------------
# torchvision.models.vgg16().features[:23] = [conv3x3+ReLU]x2, MaxPool, [conv+ReLU]x2, MaxPool, [conv+ReLU]x3,
# MaxPool, [conv+ReLU]x3 (conv indices 0,2,5,7,10,12,14,17,19,21; channels 64,64,128,128,256,256,256,512,512,512).
# torchvision is not installed here and the pretrained numbers need the network, so `p` is any state dict with
# torchvision's keys (`features.<idx>.weight|bias`).
_VGG16_CONV4_3 = (0, 2, "M", 5, 7, "M", 10, 12, 14, "M", 17, 19, 21)


def vgg16_conv4_3(p, x):
    """PerceptualLoss.vgg_conv4_3 (scripts/models/losses.py:23-39): features[:23] ends with the ReLU after conv4_3."""
    for item in _VGG16_CONV4_3:
        if item == "M":
            x = F.max_pool2d(x, kernel_size=2, stride=2)
        else:
            x = F.relu(F.conv2d(x, p["features.%d.weight" % item], p["features.%d.bias" % item], padding=1))
    return x


def perceptual_loss(p, pred, target):
    """Per-sample mean of MSELoss(reduce=False)(phi(pred), phi(target)) (losses.py:38-41,218,227)."""
    d = vgg16_conv4_3(p, pred) - vgg16_conv4_3(p, target)
    return (d * d).flatten(1).mean(dim=1)


# --------------------------------------------------------------------------
# frame formats either side of the path (uint8 <-> normalised padded tensors)
# --------------------------------------------------------------------------
def training_loss(p1, p2, img6, t, target, lambda_r, lambda_w, vgg=None, lambda_p=0.0):
    """FullModel.forward(inference_mode=False) for one window with FREEZE=FALSE on both stages, written in differentiable torch
    ops: the reference's [B,4] loss tensor `[total, lambda_r*L1(I_t^, I_t), lambda_w*sum of the four L1 warp terms, lambda_p*
    perceptual]` (scripts/models/losses.py:103-170,196-249; superslomo_r.py:204-243) and the predicted frame.  img6 [B,6,H,W],
    t [B,1,1,1], target [B,3,H,W].  `losses.mean(0)[0].backward()` is what the reference Trainer does (scripts/main.py:138-141)."""
    c6, flow4 = stage1(p1, img6)
    in16 = flow_interp_inputs(img6, flow4, t)
    out5 = stage2(p2, in16, c6)
    pred = synthesize(img6, in16, out5, t)
    i0, i1 = img6[:, 0:3], img6[:, 3:6]
    m = lambda z: z.flatten(1).mean(1)        # noqa: E731
    ft1, ft0 = in16[:, 6:8] + out5[:, 1:3], in16[:, 8:10] + out5[:, 3:5]
    wrp = ((warp(i1, flow4[:, 0:2]) - i0).abs() + (warp(i0, flow4[:, 2:4]) - i1).abs()
           + (warp(i0, ft0) - target).abs() + (warp(i1, ft1) - target).abs())
    rec, wrp = lambda_r * m((pred - target).abs()), lambda_w * m(wrp)
    per = lambda_p * perceptual_loss(vgg, pred, target) if vgg is not None else torch.zeros_like(rec)
    return torch.stack([rec + wrp + per, rec, wrp, per], 1), pred


def frames_from_u8(frames_u8, mean, std, pad_before_norm=False, multiple=32):
    """[N,H,W,3] uint8 -> [N,3,Hp,Wp] fp32.  pad_before_norm=False: ToTensor + Normalize + EvalPad
    (scripts/utils/dataloaders/augmentations.py:141-200, zero pad in normalised space);
    True: load_batch pads the 0-255 frames with 0, then normalize_tensor
    (scripts/visualize_interpolation.py:61-88,257-262)."""
    x = frames_u8.permute(0, 3, 1, 2).float()
    n, _, h, w = x.shape
    hp, wp = -(-h // multiple) * multiple, -(-w // multiple) * multiple
    top, left = (hp - h) // 2, (wp - w) // 2
    pad = [left, wp - w - left, top, hp - h - top]
    m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    if pad_before_norm:
        return (F.pad(x, pad, mode="constant", value=0) / 255.0 - m) / s
    return F.pad((x / 255.0 - m) / s, pad, mode="constant", value=0)


def frames_to_u8(x, h, w, mean, std):
    """get_crop + denormalize + *255 + numpy astype(uint8)
    (scripts/evaluate_interpolation_results.py:143-163,192-202).  The numpy cast truncates toward zero and
    wraps modulo 256; it is reproduced through an explicit int64 step so the result does not depend on the
    platform's out-of-range float->uint8 behaviour."""
    n, _, hp, wp = x.shape
    top, left = (hp - h) // 2, (wp - w) // 2
    b = x.permute(0, 2, 3, 1)[:, top:top + h, left:left + w, :]
    m = torch.tensor(mean, dtype=torch.float32).view(1, 1, 1, 3)
    s = torch.tensor(std, dtype=torch.float32).view(1, 1, 1, 3)
    b = (b * s + m) * 255.0
    return (torch.trunc(b).to(torch.int64) & 255).to(torch.uint8)
