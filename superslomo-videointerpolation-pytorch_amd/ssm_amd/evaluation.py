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


# ---- the evaluation reader's windows (scripts/utils/dataloaders/default_reader.py) -------------------------------------------------
# images a window spans for N_FRAMES inputs (default_reader.py:36): (N_FRAMES - 1) * 8 + 1
REQD_IMAGES = {2: 9, 4: 25, 6: 41, 8: 57}


def inference_item_indexes(n_frames, interp_factor=8):
    """(input positions, ground-truth positions) INSIDE a window (default_reader.py:131-151): the inputs every
    interp_factor images, the targets = everything strictly between the two middle inputs."""
    input_idx = [i * interp_factor for i in range(n_frames)]
    mid = len(input_idx) // 2 - 1
    return input_idx, list(range(input_idx[mid] + 1, input_idx[mid + 1]))


def pad_clip_edges(n_images, n_frames=2, interp_factor=8):
    """default_reader.py:209-232 on the index list 0..n_images-1 -> (padded index list, n_last_window).

    The clip is padded so that the first window's middle pair starts at image 0 and the last window's middle pair ends at or
    after the last image: interp_factor * (N_FRAMES/2 - 1) copies of image 0 in front; behind, the same number of copies of the
    last INPUT image, plus - when the clip does not end on an input - enough copies to complete the last window, whose
    n_last_window = last_idx % interp_factor real targets are the only ones scored.
    Kept quirk: the pad value is looked up as `indexes[last_input]` AFTER the front padding was prepended, so for N_FRAMES >= 4
    it is image last_input - left_padding (clamped at 0 by the front copies), not image last_input."""
    assert n_images >= 1
    left = right = interp_factor * (n_frames // 2 - 1)
    last_idx = n_images - 1
    if last_idx % interp_factor == 0:
        n_last_window = interp_factor - 1          # the last window is full
    else:
        n_last_window = last_idx % interp_factor
        right += interp_factor - n_last_window
    indexes = [0] * left + list(range(n_images))
    last_input = (last_idx // interp_factor) * interp_factor
    return indexes + [indexes[last_input]] * right, n_last_window


def _windowed(seq, n, step):
    """more_itertools.windowed(seq, n, step=step) (fill value None), as default_reader.py:238 uses it: a window after the first
    n items and after every `step` more; a shorter-than-n sequence or a leftover of fewer than min(step, n) items is filled."""
    from collections import deque
    window, i = deque(maxlen=n), n
    for item in seq:
        window.append(item)
        i -= 1
        if not i:
            i = step
            yield tuple(window)
    size = len(window)
    if size == 0:
        return
    if size < n:
        yield tuple(window) + (None,) * (n - size)
    elif 0 < i < min(step, n):
        window += (None,) * i
        yield tuple(window)


def generate_sliding_windows(n_images, n_frames=2, interp_factor=8, reqd_images=None):
    """default_reader.py:234-248: yields (image indices of one window, number of valid targets): windows of
    REQD_IMAGES[N_FRAMES] images every interp_factor images over the padded clip; every window scores interp_factor - 1
    targets except the last, which scores n_last_window.  (The reference yields paths; indices here.)"""
    reqd = REQD_IMAGES[n_frames] if reqd_images is None else reqd_images
    indexes, n_last = pad_clip_edges(n_images, n_frames, interp_factor)
    windows = list(_windowed(indexes, reqd, interp_factor))
    for k, win in enumerate(windows):
        assert None not in win, "clip of %d images is too short for windows of %d" % (n_images, reqd)
        yield list(win), (n_last if k == len(windows) - 1 else interp_factor - 1)


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


# ---- Evaluator (scripts/evaluate_interpolation_results.py:35-278) on the device pipeline ------------------------------------------

def clip_samples(frames_u8, cfg=None, n_frames=2, interp_factor=8):
    """What the evaluation dataset yields for one clip (default_reader.py:86-107 get_inference_item over
    generate_sliding_windows, batch 1): (input [1,N,3,Hp,Wp], target [1,interp_factor-1,3,Hp,Wp], [n_targets]) per window, built
    on the device from a uint8 clip [L,H,W,3]: Normalize -> ToTensor -> EvalPad (zero rows in normalised space) is one HIP kernel."""
    x = F.frames_from_u8(frames_u8, cfg, pad_before_norm=False)
    input_idx, target_idx = inference_item_indexes(n_frames, interp_factor)
    for win, n in generate_sliding_windows(frames_u8.shape[0], n_frames, interp_factor):
        yield (x[[win[i] for i in input_idx]][None], x[[win[i] for i in target_idx]][None], [n])


class Evaluator:
    """`Evaluator` of scripts/evaluate_interpolation_results.py with the model on the HIP path: per window the t loop
    (:213-244, hoisted - stage 1 once per window), trimming of the last window to its valid targets (:110-141), crop ->
    denormalise -> uint8 (:143-163,192-202; one HIP kernel), PSNR / IE / SSIM per frame (:101-108,165-190), running lists and
    their means (:246-278).  `samples`: an iterable of (input [B,N,3,Hp,Wp], target [B,T,3,Hp,Wp], n_avail[B]) - the reference
    builds it from its dataset classes (file readers: out of scope); `clip_samples` builds it from a device-resident clip."""

    def __init__(self, cfg, model, h_in, w_in, dataset="ADOBE"):
        assert dataset in ["SINTEL_HFR", "ADOBE", "SLOWFLOW", "VIMEO"], "Invalid dataset."
        self.cfg, self.model, self.dataset = cfg, model, dataset
        self.video_PSNR, self.video_IE, self.video_SSIM = [], [], []
        (self.H_REF, self.W_REF), (self.H_START, self.W_START) = F.padded_dims(h_in, w_in)
        self.H_IN, self.W_IN = h_in, w_in
        self.n_frames = cfg.getint("TRAIN", "N_FRAMES")
        self.interp_factor = 32 if dataset == "SINTEL_HFR" else 8

    def get_t_values(self):
        """VIMEO: the middle frame only (:220-221); otherwise every position 1 .. interp_factor-1 (:234-235)."""
        if self.dataset == "VIMEO":
            return [4.0 / self.interp_factor]
        return t_values(self.interp_factor)

    @torch.no_grad()
    def interpolate_frames(self, current_batch):
        """[B,N,3,Hp,Wp] -> list over t of [B,3,Hp,Wp] (:213-244).  One clip at a time through the hoisted entry points."""
        ts = self.get_t_values()
        per_clip = []
        for b in range(current_batch.shape[0]):
            clip = current_batch[b:b + 1].contiguous()
            if self.n_frames == 2:
                per_clip.append(self.model.interpolate(clip, ts))                # [T,3,Hp,Wp]
            else:
                per_clip.append(self.model.interpolate_windows(clip, ts))
        out = torch.stack(per_clip, 0)                                            # [B,T,3,Hp,Wp]
        res = [out[:, k] for k in range(len(ts))]
        assert len(res) == len(ts) and tuple(res[0].shape[1:]) == (3, self.H_REF, self.W_REF)      # validators.py:54-66
        return res

    def convert_tensor_to_numpy_image(self, batch):
        """get_crop + denormalize + astype(uint8) (:143-163,192-202): [B,3,Hp,Wp] -> uint8 [B,H_IN,W_IN,3] (the reference's
        truncating, wrapping cast)."""
        return F.frames_to_u8(batch, self.H_IN, self.W_IN, self.cfg, saturate=False).cpu().numpy()

    def get_scores(self, output_batch, target_batch):
        out, tgt = self.convert_tensor_to_numpy_image(output_batch), self.convert_tensor_to_numpy_image(target_batch)
        ps, ies, ss = [], [], []
        for k in range(out.shape[0]):
            p, s, ie = eval_single_image(tgt[k], out[k])
            ps.append(p)
            ies.append(ie)
            ss.append(s)
        return ps, ies, ss

    def eval_batch(self, input_batch, target_batch, n_avail):
        outputs = torch.stack(self.interpolate_frames(input_batch), dim=1)        # B T C H W
        assert outputs.shape[0] == len(n_avail)
        outs, tgts = [], []
        for b, n in enumerate(n_avail):
            n = int(n)
            keep = n if n < self.interp_factor - 1 else outputs.shape[1]          # the clip's last window: its real targets only
            outs.append(outputs[b, :keep])
            tgts.append(target_batch[b, :keep])
        p, ie, s = self.get_scores(torch.cat(outs, 0), torch.cat(tgts, 0))
        self.video_PSNR.extend(p)
        self.video_IE.extend(ie)
        self.video_SSIM.extend(s)

    def run_evaluation(self, samples):
        for input_batch, target_batch, n_avail in samples:
            self.eval_batch(input_batch.float(), target_batch.float(), n_avail)
        return {"PSNR": float(np.mean(self.video_PSNR)), "IE": float(np.mean(self.video_IE)),
                "SSIM": float(np.mean(self.video_SSIM)), "frames": len(self.video_PSNR)}
