"""Layer table of the two U-Nets and a deterministic, RNG-independent weight fill.

The layer table restates the shapes built by the reference constructors
(scripts/models/flow_computation.py:27-153, flow_interpolation.py:27-157);
state-dict keys are the weight ABI (`<name>.0.weight|bias`, `conv6.{0,1}.0.*`,
`final_conv.*`), tensors are OIHW fp32.

Trained weights are not distributed with the reference (weights/README.org:3),
so parity fixtures and the benchmark use `synthetic_state_dict`: every value is
a pure function of (tensor name, flat index), which makes the fill identical in
the build container (where the reference is imported to produce goldens), on
the GPU box and across torch versions.  The scale is He-uniform for
LeakyReLU(0.1) so that activations stay O(1) through all 24 layers and the
predicted flows are a few pixels - a vanishing network would make end-to-end
parity vacuous.
"""

import zlib

import numpy as np
import torch

# (name, cin, cout, k); stage-2 differences handled in unet_layers()
_ENC = [
    ("conv1a", None, 32, 7), ("conv1b", 32, 32, 7),
    ("conv2a", 32, 64, 5), ("conv2b", 64, 64, 5),
    ("conv3a", 64, 128, 3), ("conv3b", 128, 128, 3),
    ("conv4a", 128, 256, 3), ("conv4b", 256, 256, 3),
    ("conv5a", 256, 512, 3), ("conv5b", 512, 512, 3),
    ("conv6.0", 512, 512, 3), ("conv6.1", 512, 512, 3),
]
_DEC = [
    ("conv7a", None, 512, 3), ("conv7b", 512, 512, 3),
    ("conv8a", 1024, 256, 3), ("conv8b", 256, 256, 3),
    ("conv9a", 512, 128, 3), ("conv9b", 128, 128, 3),
    ("conv10a", 256, 64, 3), ("conv10b", 64, 64, 3),
    ("conv11a", 128, 32, 3), ("conv11b", 32, 32, 3),
    ("fuse_conv", 64, 32, 3),
    ("final_conv", 32, None, 3),
]


RECURRENT_HIDDEN = 256      # per direction: ConvBLSTM/ConvBGRU(hidden_channels=512) splits it over the two nets
RECURRENT_LAYERS = 2


def recurrent_convs(kind):
    """Convolutions of the recurrent bottleneck (BOTTLENECK=CLSTM|CGRU): list of (state-dict prefix, cin_x,
    hidden, cout).  Filters are [cout, cin_x + hidden, 3, 3] over cat[x, h] (ConvLSTM: cout = 4*hidden, gate order
    i,f,o,g; ConvGRU: conv_gates 2*hidden (reset, update) and conv_can hidden).  Key names follow the published
    SreenivasVRao/ConvGRU-ConvLSTM-PyTorch modules the reference constructs at flow_computation.py:73-88."""
    assert kind in ("CLSTM", "CGRU"), kind
    out = []
    for net in ("forward_net", "reverse_net"):
        for l in range(RECURRENT_LAYERS):
            cin = 512 if l == 0 else RECURRENT_HIDDEN
            cell = "conv6.%s.cell_list.%d." % (net, l)
            if kind == "CLSTM":
                out.append((cell + "conv", cin, RECURRENT_HIDDEN, 4 * RECURRENT_HIDDEN))
            else:
                out.append((cell + "conv_gates", cin, RECURRENT_HIDDEN, 2 * RECURRENT_HIDDEN))
                out.append((cell + "conv_can", cin, RECURRENT_HIDDEN, RECURRENT_HIDDEN))
    return out


def unet_layers(stage, cross_skip=True, bottleneck="CONV"):
    """List of (name, cin, cout, k) for stage 1 (6->4) or stage 2 (16->5).  With a recurrent bottleneck the two
    conv6 layers are absent (see recurrent_convs)."""
    assert stage in (1, 2), "Unsupported stage id."
    cin0, cout_final = (6, 4) if stage == 1 else (16, 5)
    c7 = 1024 if (stage == 2 and cross_skip) else 512
    out = []
    for name, cin, cout, k in _ENC + _DEC:
        if name.startswith("conv6.") and bottleneck != "CONV":
            continue
        if name == "conv1a":
            cin = cin0
        if name == "conv7a":
            cin = c7
        if name == "final_conv":
            cout = cout_final
        out.append((name, cin, cout, k))
    return out


def param_key(name, what):
    """State-dict key of a layer's weight/bias (final_conv is a bare Conv2d)."""
    return "%s.%s" % (name, what) if name == "final_conv" else "%s.0.%s" % (name, what)


def _hash_uniform(tag, n):
    """n floats in [0,1), a pure function of (tag, index): splitmix64 finaliser."""
    seed = np.uint64(zlib.crc32(tag.encode()) * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + seed
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def _hash_normal(tag, n):
    """n standard-normal floats, a pure function of (tag, index): Box-Muller on two hash-uniform streams (float64 inside)."""
    u1 = _hash_uniform(tag + "/u1", n).astype(np.float64)
    u2 = _hash_uniform(tag + "/u2", n).astype(np.float64)
    return (np.sqrt(-2.0 * np.log(u1 + 2.0 ** -25)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def synthetic_state_dict(stage, cross_skip=True, gain=1.0, bottleneck="CONV", family="uniform"):
    """Deterministic OIHW fp32 state dict with the reference's keys.

    family "uniform" (the fixtures' and the benchmark's): index-hash He-uniform values.  family "smooth" (second family of the parity
    tests, r4): He-NORMAL values (heavier tails, same variance), the 7x7 / 5x5 filters multiplied by a Gaussian window (low-pass,
    spatially structured filters like trained first layers) and renormalised to the same variance, decoder layers at gain 1.25 -
    another point of the parity margin than one weight distribution gives."""
    assert family in ("uniform", "smooth"), family
    if family == "smooth":
        return _smooth_state_dict(stage, cross_skip, gain, bottleneck)
    sd = {}
    if bottleneck != "CONV":
        for prefix, cin, hid, cout in recurrent_convs(bottleneck):
            fan_in = (cin + hid) * 9
            bound = 4.0 * gain * float(np.sqrt(6.0 / fan_in))   # x4: gate pre-activations O(1), |h| ~ 0.2
            tag = "stage%d/%s" % (stage, prefix)
            w = (_hash_uniform(tag + "/w", cout * (cin + hid) * 9) * 2.0 - 1.0) * bound
            b = (_hash_uniform(tag + "/b", cout) * 2.0 - 1.0) * 0.05
            sd[prefix + ".weight"] = torch.from_numpy(w.astype(np.float32).reshape(cout, cin + hid, 3, 3).copy())
            sd[prefix + ".bias"] = torch.from_numpy(b.astype(np.float32).copy())
    for name, cin, cout, k in unet_layers(stage, cross_skip, bottleneck):
        fan_in = cin * k * k
        bound = gain * float(np.sqrt(6.0 / (1.01 * fan_in)))
        if name == "final_conv":
            bound *= 2.0  # flows of a few px, visibility logits away from 0
        wkey, bkey = param_key(name, "weight"), param_key(name, "bias")
        tag = "stage%d/%s" % (stage, name)
        w = (_hash_uniform(tag + "/w", cout * cin * k * k) * 2.0 - 1.0) * bound
        b = (_hash_uniform(tag + "/b", cout) * 2.0 - 1.0) * 0.05
        sd[wkey] = torch.from_numpy(w.astype(np.float32).reshape(cout, cin, k, k).copy())
        sd[bkey] = torch.from_numpy(b.astype(np.float32).copy())
    return sd


def _smooth_state_dict(stage, cross_skip, gain, bottleneck):
    assert bottleneck == "CONV", "the second weight family covers the convolutional U-Nets"
    sd = {}
    for name, cin, cout, k in unet_layers(stage, cross_skip, bottleneck):
        fan_in = cin * k * k
        std = gain * float(np.sqrt(2.0 / (1.01 * fan_in)))
        if name == "final_conv":
            std *= 2.0
        elif name[:5] in ("conv7", "conv8", "conv9") or name[:6] in ("conv10", "conv11") or name == "fuse_conv":
            std *= 1.25
        tag = "smooth/stage%d/%s" % (stage, name)
        w = _hash_normal(tag + "/w", cout * cin * k * k).reshape(cout, cin, k, k).astype(np.float64)
        if k >= 5:
            ax = np.arange(k, dtype=np.float64) - (k - 1) / 2.0
            win = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / (2.0 * (k / 4.0) ** 2))
            w = w * (win / np.sqrt((win ** 2).mean()))
        b = (_hash_uniform(tag + "/b", cout) * 2.0 - 1.0) * 0.05
        sd[param_key(name, "weight")] = torch.from_numpy((w * std).astype(np.float32).copy())
        sd[param_key(name, "bias")] = torch.from_numpy(b.astype(np.float32).copy())
    return sd


IMAGENET_MEAN = (0.485, 0.456, 0.406)   # configs/superslomo_original.ini:57
IMAGENET_STD = (0.229, 0.224, 0.225)    # configs/superslomo_original.ini:58


def synthetic_frames_u8(n_frames, h, w, seed=42):
    """Synthetic Adobe240-shaped clip: smooth low-pass texture plus fine grain,
    translated (3,2) px per frame so the scene has real motion.  uint8 RGB
    [n_frames, 3, h, w]."""
    rng = np.random.RandomState(seed)
    m = 16
    big = rng.rand(3, h // 8 + 2 * m, w // 8 + 2 * m).astype(np.float32)
    t = torch.from_numpy(big)[None]
    t = torch.nn.functional.interpolate(t, scale_factor=8, mode="bicubic", align_corners=False)[0]
    t = (t - t.min()) / (t.max() - t.min())
    fine = torch.from_numpy(rng.rand(3, t.shape[1], t.shape[2]).astype(np.float32)) * 0.08
    t = (t * 0.92 + fine).clamp(0, 1)
    frames = []
    for i in range(n_frames):
        dx, dy = 3 * i, 2 * i
        crop = t[:, 8 * m - dy: 8 * m - dy + h, 8 * m - dx: 8 * m - dx + w]
        frames.append(torch.round(crop * 255.0).clamp(0, 255).to(torch.uint8))
    return torch.stack(frames)


def synthetic_frames_edges_u8(n_frames, h, w, seed=7, motion=(28, 20)):
    """Second frame family of the parity tests (r4): HARD edges - flat rectangles, bars and a checkerboard at full contrast over a
    coarse gradient - translated `motion` (default 28, 20) px per frame: large motion and step edges instead of the low-pass texture
    with 3-px motion of synthetic_frames_u8.  uint8 RGB [n_frames, 3, h, w]."""
    rng = np.random.RandomState(seed)
    mx, my = abs(motion[0]) * n_frames + 8, abs(motion[1]) * n_frames + 8
    H, W = h + 2 * my, w + 2 * mx
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.stack([0.25 + 0.5 * xx / W, 0.25 + 0.5 * yy / H, 0.5 + 0.0 * xx]).astype(np.float32)
    for _ in range(60):          # flat rectangles
        y0, x0 = rng.randint(0, H - 8), rng.randint(0, W - 8)
        hh, ww = rng.randint(8, max(9, H // 4)), rng.randint(8, max(9, W // 4))
        img[:, y0:y0 + hh, x0:x0 + ww] = rng.rand(3, 1, 1).astype(np.float32)
    for i in range(12):          # thin bars, both directions
        p = rng.randint(0, W - 4)
        img[:, :, p:p + 1 + i % 3] = float(i % 2)
        p = rng.randint(0, H - 4)
        img[:, p:p + 1 + i % 3, :] = float((i + 1) % 2)
    cy, cx = H // 3, W // 3          # a one-pixel and an eight-pixel checkerboard
    img[:, cy:cy + 96, cx:cx + 96] = (((yy[cy:cy + 96, cx:cx + 96] + xx[cy:cy + 96, cx:cx + 96]) % 2) > 0.5).astype(np.float32)
    img[:, cy + 128:cy + 256, cx:cx + 128] = ((((yy[cy + 128:cy + 256, cx:cx + 128] // 8) + (xx[cy + 128:cy + 256, cx:cx + 128] // 8)) % 2) > 0.5)
    t = torch.from_numpy(img)
    frames = []
    for i in range(n_frames):
        dx, dy = motion[0] * i, motion[1] * i
        crop = t[:, my + dy: my + dy + h, mx + dx: mx + dx + w]
        frames.append(torch.round(crop * 255.0).clamp(0, 255).to(torch.uint8))
    return torch.stack(frames)


def normalize_and_pad(u8, pad_to=32):
    """Input contract of the path (SURVEY a-0): x/255, ImageNet-normalise
    (scripts/utils/dataloaders/augmentations.py:141-200), then zero-pad IN
    NORMALISED SPACE to a multiple of 32, centred
    (scripts/utils/dataloaders/default_reader.py:266-271).
    uint8 [N,3,h,w] -> float32 [1,N,3,Hp,Wp]."""
    n, _, h, w = u8.shape
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32).view(1, 3, 1, 1)
    x = (u8.to(torch.float32) / 255.0 - mean) / std
    hp = (h + pad_to - 1) // pad_to * pad_to
    wp = (w + pad_to - 1) // pad_to * pad_to
    top, left = (hp - h) // 2, (wp - w) // 2
    out = torch.zeros(1, n, 3, hp, wp, dtype=torch.float32)
    out[0, :, :, top: top + h, left: left + w] = x
    return out


def synthetic_frames(n_frames, h, w, seed=42, pad_to=32, family="texture"):
    """normalize_and_pad(synthetic_frames_u8(...)): float32 [1,n_frames,3,Hp,Wp].  family "edges": synthetic_frames_edges_u8."""
    assert family in ("texture", "edges"), family
    u8 = synthetic_frames_u8(n_frames, h, w, seed) if family == "texture" else synthetic_frames_edges_u8(n_frames, h, w, seed)
    return normalize_and_pad(u8, pad_to)
