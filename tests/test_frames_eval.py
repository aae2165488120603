"""Frame-format kernels (bit-exact vs the oracle) and the evaluator/visualiser host logic."""
import numpy as np
import pytest
import torch

from oracle import ssm_oracle as O
from ssm_amd.weights import IMAGENET_MEAN, IMAGENET_STD


def test_sliding_window_matches_reference_semantics():
    from ssm_amd.evaluation import sliding_window, t_values
    assert list(sliding_window(4, 2)) == [[0, 1], [1, 2], [2, 3]]
    # N_FRAMES=4: window centred on the pair, clamped at both ends (visualize_interpolation.py:277-286)
    assert list(sliding_window(4, 4)) == [[0, 0, 1, 2], [0, 1, 2, 3], [1, 2, 3, 3]]
    assert list(sliding_window(17, 2, is_fps_240=True)) == [[0, 8], [8, 16]]
    assert t_values(8) == [i / 8.0 for i in range(1, 8)]


def test_evaluation_windows_known_answers():
    """pad_clip_edges / generate_sliding_windows (default_reader.py:209-248), answers worked out by hand from that code."""
    from ssm_amd.evaluation import generate_sliding_windows as gsw, inference_item_indexes, pad_clip_edges
    # N_FRAMES = 2: no edge padding; a clip that ends on an input has a full last window
    assert pad_clip_edges(9, 2) == (list(range(9)), 7)
    assert list(gsw(9, 2)) == [(list(range(9)), 7)]
    assert pad_clip_edges(17, 2) == (list(range(17)), 7)
    assert list(gsw(17, 2)) == [(list(range(0, 9)), 7), (list(range(8, 17)), 7)]
    # 20 images: last_idx 19 = 2 * 8 + 3 -> three real targets in the last window, five copies of the last input (16) complete it
    idx, n_last = pad_clip_edges(20, 2)
    assert n_last == 3 and idx == list(range(20)) + [16] * 5
    assert list(gsw(20, 2)) == [(list(range(0, 9)), 7), (list(range(8, 17)), 7), ([16, 17, 18, 19, 16, 16, 16, 16, 16], 3)]
    # N_FRAMES = 4: 8 copies of image 0 in front; behind 8 (+ 5) copies of indexes[last_input] of the FRONT-PADDED list = image 8
    idx, n_last = pad_clip_edges(20, 4)
    assert n_last == 3 and idx == [0] * 8 + list(range(20)) + [8] * 13 and len(idx) == 41
    w = list(gsw(20, 4))
    assert [n for _, n in w] == [7, 7, 3] and all(len(win) == 25 for win, _ in w)
    assert w[0][0] == [0] * 8 + list(range(17)) and w[1][0] == list(range(20)) + [8] * 5 and w[2][0] == list(range(8, 20)) + [8] * 13
    idx, n_last = pad_clip_edges(17, 4)           # ends on an input: full last window, 8 copies of indexes[16] = image 8
    assert n_last == 7 and idx == [0] * 8 + list(range(17)) + [8] * 8
    assert [win for win, _ in gsw(17, 4)] == [[0] * 8 + list(range(17)), list(range(17)) + [8] * 8]
    idx, n_last = pad_clip_edges(9, 4)            # last_input 8 -> indexes[8] of the padded list = image 0
    assert idx == [0] * 8 + list(range(9)) + [0] * 8 and [n for _, n in gsw(9, 4)] == [7]
    # positions inside a window: inputs every 8 images, targets strictly between the two middle inputs
    assert inference_item_indexes(2) == ([0, 8], list(range(1, 8)))
    assert inference_item_indexes(4) == ([0, 8, 16, 24], list(range(9, 16)))
    win = w[2][0]                                 # the last window of the 20-image clip, N_FRAMES = 4
    assert [win[i] for i in inference_item_indexes(4)[0]] == [8, 16, 8, 8]
    assert [win[i] for i in inference_item_indexes(4)[1]][:3] == [17, 18, 19]
    with pytest.raises(AssertionError):
        list(gsw(5, 4, interp_factor=8, reqd_images=60))      # a window longer than the padded clip is filled with None upstream


def test_metrics_known_answers():
    from ssm_amd.evaluation import eval_single_image, interpolation_error, psnr, ssim
    rng = np.random.RandomState(0)
    a = rng.randint(0, 256, (48, 64, 3)).astype(np.uint8)
    assert psnr(a, a) == float("inf") and interpolation_error(a, a) == 0.0 and abs(ssim(a, a) - 1.0) < 1e-12
    b = a.copy()
    b[..., 0] = np.clip(a[..., 0].astype(int) + 4, 0, 255)     # +4 on red where not saturated
    d = a.astype(float) - b.astype(float)
    assert abs(psnr(a, b) - 10 * np.log10(255 ** 2 / np.mean(d ** 2))) < 1e-9
    assert abs(interpolation_error(a, b) - np.mean(np.abs(d[..., 0]))) < 1e-9
    p, s, ie = eval_single_image(a, b)
    assert 30 < p < 50 and 0.9 < s < 1.0 and 0 < ie <= 4
    # SSIM decreases with distortion and is symmetric
    c = np.clip(a.astype(int) + rng.randint(-40, 41, a.shape), 0, 255).astype(np.uint8)
    assert ssim(a, c) < ssim(a, b) and abs(ssim(a, c) - ssim(c, a)) < 1e-12


def test_oracle_frame_formats_roundtrip():
    rng = np.random.RandomState(1)
    u8 = torch.from_numpy(rng.randint(0, 256, (2, 45, 70, 3)).astype(np.uint8))
    for pbn in (False, True):
        x = O.frames_from_u8(u8, IMAGENET_MEAN, IMAGENET_STD, pad_before_norm=pbn)
        assert tuple(x.shape) == (2, 3, 64, 96)
        back = O.frames_to_u8(x, 45, 70, IMAGENET_MEAN, IMAGENET_STD)
        assert (back.int() - u8.int()).abs().max() <= 1          # truncation may lose one grey level
    assert float(O.frames_from_u8(u8, IMAGENET_MEAN, IMAGENET_STD)[0, :, 0, 0].abs().max()) == 0.0
    want = (0.0 - IMAGENET_MEAN[0]) / IMAGENET_STD[0]
    assert abs(float(O.frames_from_u8(u8, IMAGENET_MEAN, IMAGENET_STD, True)[0, 0, 0, 0]) - want) < 1e-6


@pytest.mark.gpu
def test_frame_kernels_bit_exact_vs_oracle():
    from ssm_amd import frames as F
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(2)
    for (n, h, w) in ((2, 45, 70), (1, 720, 1280), (3, 64, 64)):
        u8 = torch.from_numpy(rng.randint(0, 256, (n, h, w, 3)).astype(np.uint8))
        for pbn in (False, True):
            got = F.frames_from_u8(u8.to(dev), None, pad_before_norm=pbn).cpu()
            want = O.frames_from_u8(u8, IMAGENET_MEAN, IMAGENET_STD, pad_before_norm=pbn)
            assert torch.equal(got, want), "ingest %s pad_before_norm=%s" % ((n, h, w), pbn)
        # egress: values inside and far outside [0,255] (the reference's cast wraps), NaN-free
        x = torch.randn(n, 3, -(-h // 32) * 32, -(-w // 32) * 32) * 1.5
        got = F.frames_to_u8(x.to(dev), h, w).cpu()
        want = O.frames_to_u8(x, h, w, IMAGENET_MEAN, IMAGENET_STD)
        assert torch.equal(got, want), "egress %s" % ((n, h, w),)
        sat = F.frames_to_u8(x.to(dev), h, w, saturate=True).cpu()
        assert int(sat.max()) <= 255 and (sat.int() - want.int()).abs().float().mean() < 64


@pytest.mark.gpu
def test_interpolate_clip_end_to_end():
    """uint8 clip in -> uint8 intermediates out (ingest kernel, 2-stream pipeline, egress kernel); checked
    against the CPU oracle pipeline on the same clip; PSNR between consecutive outputs is finite."""
    from models.superslomo_r import FullModel
    from ssm_amd.config import load_config, synthetic_weight_overrides
    from ssm_amd.evaluation import interpolate_clip, psnr
    from ssm_amd.weights import synthetic_frames_u8, synthetic_state_dict
    dev = torch.device("cuda:0")
    m = FullModel(load_config("superslomo_original.ini", synthetic_weight_overrides()))
    sd1, sd2 = synthetic_state_dict(1), synthetic_state_dict(2)
    m.stage1_model.load_state_dict(sd1)
    m.stage2_model.load_state_dict(sd2)
    m = m.to(dev).eval()
    clip = synthetic_frames_u8(3, 60, 90, seed=5).permute(0, 2, 3, 1).contiguous()      # [3,60,90,3]
    out = interpolate_clip(m, clip.to(dev), upsample_rate=4, saturate=True).cpu()
    assert tuple(out.shape) == (2, 3, 60, 90, 3) and out.dtype == torch.uint8
    x = O.frames_from_u8(clip, IMAGENET_MEAN, IMAGENET_STD, pad_before_norm=True)
    for p in range(2):
        pair = torch.cat([x[p], x[p + 1]])[None]
        want = O.interpolate_pair(sd1, sd2, pair, [0.25, 0.5, 0.75])
        for k in range(3):
            ref = torch.round((want[k].permute(0, 2, 3, 1)[0, 2:62, 3:93] * torch.tensor(IMAGENET_STD)
                               + torch.tensor(IMAGENET_MEAN)) * 255.0).clamp(0, 255)
            assert (out[p, k].float() - ref).abs().max() <= 1.0        # <= one grey level
    assert np.isfinite(psnr(out[0, 0].numpy(), out[0, 1].numpy()))


@pytest.mark.gpu
def test_visualize_cli_writes_interleaved_pngs(tmp_path):
    """The visualiser entry point (reference flags) on 3 PNG frames, rate 4: 3 originals + 2x3 intermediates, in order."""
    from PIL import Image
    import visualize_interpolation as V
    from models.superslomo_r import FullModel
    from ssm_amd.config import CONFIG_DIR, load_config, synthetic_weight_overrides
    from ssm_amd.weights import synthetic_frames_u8, synthetic_state_dict
    import os
    clip = synthetic_frames_u8(3, 60, 90, seed=5).permute(0, 2, 3, 1).numpy()
    src = tmp_path / "in"
    src.mkdir()
    for i in range(3):
        Image.fromarray(clip[i]).save(src / ("f_%03d.png" % i))
    m = FullModel(load_config("superslomo_original.ini", synthetic_weight_overrides()))
    m.stage1_model.load_state_dict(synthetic_state_dict(1))
    m.stage2_model.load_state_dict(synthetic_state_dict(2))
    # the shipped ini has LOADPREV=TRUE (no weights are distributed) -> hand the CLI a ready model
    n = V.main(["-c", os.path.join(CONFIG_DIR, "superslomo_original.ini"), "--expt", "t", "--log", str(tmp_path / "l.log"),
                "--input_dir", str(src), "--img_type", "png", "--upsample_rate", "4", "--output_dir", str(tmp_path / "out")],
               model=m)
    files = sorted(os.listdir(tmp_path / "out" / "t" / "images"))
    assert n == 9 and files == ["img_%05d.png" % i for i in range(9)]
    assert np.array_equal(np.asarray(Image.open(tmp_path / "out" / "t" / "images" / "img_00004.png")), clip[1])
    mid = np.asarray(Image.open(tmp_path / "out" / "t" / "images" / "img_00002.png"))
    assert mid.shape == (60, 90, 3) and mid.std() > 1.0


@pytest.mark.gpu
def test_evaluator_loop_on_a_synthetic_clip():
    """Evaluator (evaluate_interpolation_results.py:35-278) over a 12-image clip, N_FRAMES = 2: two windows, 7 + 3 scored frames (the
    last window keeps its real targets only).  The loop's bookkeeping - windows, the t loop, trimming, crop / denormalise / uint8, the
    running lists and their means - against a direct recomputation from the same device pipeline (its numerics against the CPU oracle:
    test_interpolate_clip_end_to_end above and tests/test_hip_model.py), and one window against the CPU oracle."""
    from models.superslomo_r import FullModel
    from ssm_amd import frames as F
    from ssm_amd.config import load_config, synthetic_weight_overrides
    from ssm_amd.evaluation import Evaluator, clip_samples, eval_single_image, generate_sliding_windows, inference_item_indexes, t_values
    from ssm_amd.weights import synthetic_frames_u8, synthetic_state_dict
    dev = torch.device("cuda:0")
    cfg = load_config("superslomo_original.ini", synthetic_weight_overrides())
    m = FullModel(cfg)
    sd1, sd2 = synthetic_state_dict(1), synthetic_state_dict(2)
    m.stage1_model.load_state_dict(sd1)
    m.stage2_model.load_state_dict(sd2)
    m = m.to(dev).eval()
    h, w = 60, 90
    clip = synthetic_frames_u8(12, h, w, seed=11).permute(0, 2, 3, 1).contiguous()
    ev = Evaluator(cfg, m, h, w, dataset="ADOBE")
    assert (ev.H_REF, ev.W_REF, ev.H_START, ev.W_START) == (64, 96, 2, 3)
    got = ev.run_evaluation(clip_samples(clip.to(dev), cfg, n_frames=2))
    assert got["frames"] == 10 and len(ev.video_PSNR) == len(ev.video_IE) == len(ev.video_SSIM) == 10
    x = F.frames_from_u8(clip.to(dev), cfg, pad_before_norm=False)
    ins, tg = inference_item_indexes(2)
    P, S, E = [], [], []
    for win, n in generate_sliding_windows(12, 2):
        outs = m.interpolate(torch.stack([x[win[ins[0]]], x[win[ins[1]]]])[None], t_values(8))
        o8 = F.frames_to_u8(outs[:n], h, w, cfg).cpu().numpy()
        # the targets take the reference's round trip too (normalise -> denormalise -> truncating cast: evaluate_interpolation_results.py:
        # 143-163 applies convert_tensor_to_numpy_image to the target batch as well; it can cost a grey level)
        t8 = F.frames_to_u8(x[[win[tg[k]] for k in range(n)]], h, w, cfg).cpu().numpy()
        for k in range(n):
            assert win[tg[k]] < 12                                   # a scored target is a real image of the clip, never a pad copy
            assert int(np.abs(t8[k].astype(int) - clip[win[tg[k]]].numpy().astype(int)).max()) <= 1
            p, s, e = eval_single_image(t8[k], o8[k])
            P.append(p), S.append(s), E.append(e)
    assert abs(got["PSNR"] - np.mean(P)) < 1e-9 and abs(got["IE"] - np.mean(E)) < 1e-9 and abs(got["SSIM"] - np.mean(S)) < 1e-9
    # one window against the CPU oracle: uint8 frames may differ by one grey level
    xo = O.frames_from_u8(clip, IMAGENET_MEAN, IMAGENET_STD, pad_before_norm=False)
    want = O.interpolate_pair(sd1, sd2, torch.cat([xo[0], xo[8]])[None], [0.5])[0]
    w8 = O.frames_to_u8(want, h, w, IMAGENET_MEAN, IMAGENET_STD)[0]
    g8 = F.frames_to_u8(m.interpolate(torch.stack([x[0], x[8]])[None], [0.5]), h, w, cfg)[0].cpu()
    assert float(((g8.int() - w8.int()).abs() > 1).float().mean()) < 1e-3          # (the reference's cast wraps at 0 / 256: a pixel there may flip)
