"""Evaluator / visualiser logic around the hot path, host side.

Mirrors what `Evaluator` (scripts/evaluate_interpolation_results.py) and `Interpolator`
(scripts/visualize_interpolation.py) do around `FullModel`: sliding windows over a frame list, the t loop (hoisted:
stage 1 once per pair), crop/denormalise/quantise (HIP kernels in ssm_amd.frames) and the three quality metrics.
PSNR and IE follow the reference's call sites exactly.  SSIM restates skimage's `structural_similarity(...,
multichannel=True, gaussian_weights=True)` (Wang et al. 2004 with skimage's conventions: sigma 1.5, 11x11 support,
sample covariance, border crop); skimage is not installed in this image, so SSIM is NOT pinned against it.
"""
import numpy as np
import torch

from . import frames as F


def sliding_window(n_images, n_frames=2, is_fps_240=False):
    """Index windows of `Interpolator.sliding_window` (scripts/visualize_interpolation.py:270-288): for every adjacent
    pair (i, i+1) the N_FRAMES-long window centred on it, indices clamped at the clip's ends.  With is_fps_240 the
    clip is first subsampled by 8.  Yields lists of indices into the ORIGINAL image list."""
    idx = list(range(n_images))[::8] if is_fps_240 else list(range(n_images))
    half = (n_frames - 1) // 2
    for a in range(len(idx) - 1):
        locs = [min(max(i, 0), len(idx) - 1) for i in range(a - half, a + 1 + half + 1)]
        yield [idx[i] for i in locs]


def t_values(interp_factor):
    """t = idx / interp_factor, idx = 1..interp_factor-1 (scripts/evaluate_interpolation_results.py:204-211,234-239)."""
    return [i / float(interp_factor) for i in range(1, interp_factor)]


def psnr(target_u8, output_u8):
    """skimage.metrics.peak_signal_noise_ratio on uint8 images (data_range 255):
    scripts/evaluate_interpolation_results.py:102."""
    err = np.mean((target_u8.astype(np.float64) - output_u8.astype(np.float64)) ** 2)
    return float("inf") if err == 0 else float(10.0 * np.log10(255.0 ** 2 / err))


def interpolation_error(target_u8, output_u8):
    """IE = mean over pixels of the RGB error norm: scripts/evaluate_interpolation_results.py:106-107."""
    d = target_u8.astype(float) - output_u8.astype(float)
    return float(np.mean(np.sqrt(np.sum(d * d, axis=2))))


def ssim(target_u8, output_u8):
    """structural_similarity(target, output, multichannel=True, gaussian_weights=True) restated (unpinned, see
    module docstring): per channel, Gaussian-weighted local moments (sigma 1.5, truncate 3.5 -> 11 taps, reflect
    boundary), sample covariance (N/(N-1), N = 11^2), K1=0.01, K2=0.03, data_range 255, mean over the image
    without a 5-pixel border, then mean over channels."""
    from scipy.ndimage import gaussian_filter
    sigma, trunc = 1.5, 3.5
    win = 2 * int(trunc * sigma + 0.5) + 1
    cov_norm = win * win / (win * win - 1.0)
    c1, c2 = (0.01 * 255.0) ** 2, (0.03 * 255.0) ** 2
    pad = (win - 1) // 2
    vals = []
    for c in range(target_u8.shape[2]):
        x, y = target_u8[..., c].astype(np.float64), output_u8[..., c].astype(np.float64)
        g = lambda a: gaussian_filter(a, sigma=sigma, truncate=trunc, mode="reflect")  # noqa: E731
        ux, uy = g(x), g(y)
        vx, vy, vxy = cov_norm * (g(x * x) - ux * ux), cov_norm * (g(y * y) - uy * uy), cov_norm * (g(x * y) - ux * uy)
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
        vals.append(s[pad:-pad, pad:-pad].mean())
    return float(np.mean(vals))


def eval_single_image(target_u8, output_u8):
    """(PSNR, SSIM, IE) like Evaluator.eval_single_image (scripts/evaluate_interpolation_results.py:101-108)."""
    return psnr(target_u8, output_u8), ssim(target_u8, output_u8), interpolation_error(target_u8, output_u8)


@torch.no_grad()
def interpolate_clip(model, frames_u8, upsample_rate=8, cfg=None, pad_before_norm=True, saturate=False, n_streams=2):
    """The visualiser's loop (scripts/visualize_interpolation.py:105-221) for N_FRAMES=2 on device-resident uint8
    frames [N,H,W,3]: every adjacent pair -> upsample_rate-1 intermediates.  Returns uint8 [N-1, rate-1, H, W, 3].
    Ingest/egress are HIP kernels; pairs are dealt to `n_streams` HIP streams."""
    n, h, w, _ = frames_u8.shape
    x = F.frames_from_u8(frames_u8, cfg, pad_before_norm=pad_before_norm)           # [N,3,Hp,Wp]
    pairs = [torch.stack([x[i], x[i + 1]])[None] for a in sliding_window(n, 2) for i in [a[0]]]
    outs = model.interpolate_many(pairs, t_values(upsample_rate), n_streams=n_streams)
    return torch.stack([F.frames_to_u8(o, h, w, cfg, saturate=saturate) for o in outs])
