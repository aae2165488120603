#!/usr/bin/env python3
"""Interpolate a folder of frames; CLI of the reference's scripts/visualize_interpolation.py (same flags, same output
naming: <output_dir>/<expt>/images/img_00000.png, originals and intermediates interleaved), MI355X-native inside:

  PNG -> uint8 on the GPU -> ingest kernel (RGB, /255, normalise, pad to x32)        [ssm_amd.frames]
      -> FullModel.interpolate_many (stage 1 once per pair, all t batched, pairs on 2 HIP streams)
      -> egress kernel (crop, denormalise, uint8) -> PNG

Differences from the reference, on purpose: frames are cropped back to their original size before they are written
(the reference writes the zero-padded canvas), PIL replaces cv2 (not installed here), flow maps are saved as raw .npy
instead of colour-wheel PNGs (flo_utils is out of scope), and N_FRAMES must be 2 (CONV bottleneck).
"""
import argparse
import configparser
import glob
import logging
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.dirname(HERE), HERE):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from models import superslomo_r as ssm  # noqa: E402
from ssm_amd import evaluation as E  # noqa: E402
from ssm_amd import frames as F  # noqa: E402

log = logging.getLogger(__name__)


def getargs(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-c", "--config", required=True, default="config.ini", help="Path to config.ini file.")
    parser.add_argument("--expt", required=True, help="Experiment Name.")
    parser.add_argument("--log", required=True, help="Path to logfile.")
    parser.add_argument("--input_dir", required=True, help="Directory with input images.")
    parser.add_argument("--img_type", required=True, help="Image type")
    parser.add_argument("--is_fps_240", action="store_true", help="Is input footage 240 fps?")
    parser.add_argument("--upsample_rate", type=int, default=8,
                        help="Integer upsampling rate. For 30FPS -> 240FP, use 8. For 1080FPS, use 36.")
    parser.add_argument("--show_intermediate_outputs", action="store_true",
                        help="Save occlusion maps, optical flow maps etc.?")
    parser.add_argument("--output_dir", required=True, help="Directory to output.")
    return parser.parse_args(argv)


class Interpolator:
    def __init__(self, cfg, args, model=None):
        self.cfg, self.args = cfg, args
        self.n_frames = cfg.getint("TRAIN", "N_FRAMES")
        if self.n_frames != 2:
            raise NotImplementedError("N_FRAMES=%d needs the recurrent bottleneck (unpinned upstream); use N_FRAMES=2"
                                      % self.n_frames)
        self.model = (model if model is not None else ssm.FullModel(cfg)).cuda().eval()
        base = os.path.join(args.output_dir, args.expt)
        self.img_dir = os.path.join(base, "images")
        self.visibility_dir = os.path.join(base, "visibility_map")
        self.flow_dir = os.path.join(base, "refined_flow")
        os.makedirs(self.img_dir, exist_ok=True)
        if args.show_intermediate_outputs:
            os.makedirs(self.visibility_dir, exist_ok=True)
            os.makedirs(self.flow_dir, exist_ok=True)

    @staticmethod
    def load_frames(paths):
        from PIL import Image
        return torch.from_numpy(np.stack([np.asarray(Image.open(p).convert("RGB")) for p in paths]))   # [N,H,W,3] uint8

    def save(self, img_u8, count, out_dir, prefix="img"):
        from PIL import Image
        Image.fromarray(img_u8).save(os.path.join(out_dir, prefix + "_" + str(count).zfill(5) + ".png"))

    @torch.no_grad()
    def interpolate_frames(self):
        a = self.args
        paths = sorted(glob.glob(os.path.join(a.input_dir, "*." + a.img_type.lower())))
        log.info("Looking for %s images in %s. 240 FPS: %s. Found %d.", a.img_type, a.input_dir, a.is_fps_240, len(paths))
        if len(paths) < 2:
            raise FileNotFoundError("need at least two *.%s frames in %s" % (a.img_type.lower(), a.input_dir))
        windows = list(E.sliding_window(len(paths), self.n_frames, a.is_fps_240))
        used = sorted({i for w in windows for i in w})
        frames = self.load_frames([paths[i] for i in used]).cuda()
        pos = {i: k for k, i in enumerate(used)}
        h, w = frames.shape[1:3]
        x = F.frames_from_u8(frames, self.cfg, pad_before_norm=True)       # visualiser convention (:76-87,:137)
        ts = E.t_values(a.upsample_rate)
        count = 0
        pairs = [torch.stack([x[pos[w0]], x[pos[w1]]])[None] for w0, w1 in windows]
        if a.show_intermediate_outputs:          # per-t forward keeps the reference's tuple of intermediates
            outs = []
            for pr in pairs:
                per_t = []
                for t in ts:
                    img, inter = self.model(pr, torch.full((1, 1, 1, 1, 1), t, device=pr.device), inference_mode=True)
                    per_t.append((img, inter))
                outs.append(per_t)
        else:
            outs = self.model.interpolate_many(pairs, ts)
        frames_cpu = frames.cpu().numpy()
        for k, (w0, w1) in enumerate(windows):
            self.save(frames_cpu[pos[w0]], count, self.img_dir)
            count += 1
            if a.show_intermediate_outputs:
                for img, inter in outs[k]:
                    top, left = (img.shape[2] - h) // 2, (img.shape[3] - w) // 2
                    v0 = inter[6][0, 0, top:top + h, left:left + w]
                    self.save((v0 * 255.0).clamp(0, 255).byte().cpu().numpy(), count, self.visibility_dir, "visibility")
                    np.save(os.path.join(self.flow_dir, "flow_t1_%05d.npy" % count),
                            inter[4][0, :, top:top + h, left:left + w].cpu().numpy())
                    np.save(os.path.join(self.flow_dir, "flow_t0_%05d.npy" % count),
                            inter[5][0, :, top:top + h, left:left + w].cpu().numpy())
                    self.save(F.frames_to_u8(img, h, w, self.cfg)[0].cpu().numpy(), count, self.img_dir)
                    log.info("Interpolated frame: %s", count)
                    count += 1
            else:
                u8 = F.frames_to_u8(outs[k], h, w, self.cfg).cpu().numpy()
                for j in range(u8.shape[0]):
                    self.save(u8[j], count, self.img_dir)
                    log.info("Interpolated frame: %s", count)
                    count += 1
        self.save(frames_cpu[pos[windows[-1][1]]], count, self.img_dir)
        return count + 1


def main(argv=None, model=None):
    args = getargs(argv)
    config = configparser.RawConfigParser()
    logging.basicConfig(filename=args.log, level=logging.INFO)
    if not config.read(args.config):
        raise FileNotFoundError(args.config)
    n = Interpolator(config, args, model=model).interpolate_frames()
    log.info("Interpolation complete.")
    return n


if __name__ == "__main__":
    main()
