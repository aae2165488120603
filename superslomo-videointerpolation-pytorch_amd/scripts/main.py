#!/usr/bin/env python3
"""Training entry point; CLI of the reference's scripts/main.py (-c/--config --expt --log [--msg]), one process per GPU
(launch with `python -m torch.distributed.run --nproc-per-node N scripts/main.py ...` for data parallelism: the gradient
all-reduce over RCCL replaces torch.nn.DataParallel, scripts/main.py:74-76 of the reference).

The dataset layer of the reference (scripts/utils/dataset.py, dataloaders/*) is out of scope, so
batches come from `--synthetic_batches N` (deterministic synthetic clips with the dataloader's tensor contract); plug a
real loader into ssm_amd.training.Trainer.train for actual data.  Checkpoints use the reference's layout
(<CKPT_DIR>/<expt>/<expt>_EPOCH_0001.pt, scripts/main.py:218-245) and load back through models.unetflow.get_model.
"""
import argparse
import configparser
import logging
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.dirname(HERE), HERE):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from models import superslomo_r as ssm  # noqa: E402
from ssm_amd import dist as sdist  # noqa: E402
from ssm_amd.training import Trainer  # noqa: E402
from ssm_amd.weights import synthetic_frames  # noqa: E402

log = logging.getLogger(__name__)


def getargs(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-c", "--config", required=True, default="config.ini", help="Path to config.ini file.")
    parser.add_argument("--expt", required=True, help="Experiment Name.")
    parser.add_argument("--log", required=True, help="Path to logfile.")
    parser.add_argument("--msg", default=None, help="(Optional) Details of experiment stored with logfile.")
    parser.add_argument("--synthetic_batches", type=int, default=0, help="train on this many synthetic batches per epoch")
    return parser.parse_args(argv)


def synthetic_batches(cfg, n, device, rank=0):
    b = cfg.getint("TRAIN", "BATCH_SIZE")
    h, w = cfg.getint("TRAIN", "CROP_IMH"), cfg.getint("TRAIN", "CROP_IMW")
    for i in range(n):
        clips = torch.cat([synthetic_frames(3, h, w, seed=1000 * rank + 10 * i + k) for k in range(b)], 0).to(device)
        t = torch.tensor([(i + k) % 7 + 1 for k in range(b)], dtype=torch.float32, device=device).view(b, 1, 1, 1, 1) / 8.0
        yield clips[:, [0, 2]].contiguous(), clips[:, 1:2].contiguous(), t        # DATALOADER.T_SAMPLE=RANDOM convention


def main(argv=None, model=None):
    args = getargs(argv)
    cfg = configparser.RawConfigParser()
    logging.basicConfig(filename=args.log, level=logging.INFO)
    if not cfg.read(args.config):
        raise FileNotFoundError(args.config)
    if args.msg:
        log.info(args.msg)
    torch.manual_seed(cfg.getint("SEED", "VALUE"))
    rank, local_rank, world = sdist.env_world()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sdist.init("nccl")
    model = (model if model is not None else ssm.FullModel(cfg)).to(dev).train()
    trainer = Trainer(model, cfg)
    if args.synthetic_batches <= 0:
        raise NotImplementedError("the reference's dataset readers are out of scope here; pass --synthetic_batches N, or "
                                  "feed ssm_amd.training.Trainer.train your own (input, target, t_interp) batches")
    n_epochs, save_every = cfg.getint("TRAIN", "N_EPOCHS"), cfg.getint("TRAIN", "SAVE_EVERY")
    ckpt_dir = os.path.join(cfg.get("TRAIN", "CKPT_DIR"), args.expt)
    last = None
    for epoch in range(trainer.start, n_epochs + 1):          # trainer.start: 1, or the checkpoint's epoch when resuming (main.py:263-284)
        trainer.train(synthetic_batches(cfg, args.synthetic_batches, dev, rank), n_epochs=1,
                      on_step=lambda e, it, losses: log.info("epoch %d it %d losses %s", epoch, it, losses.tolist()))
        if rank == 0 and epoch % save_every == 0:
            last = os.path.join(ckpt_dir, args.expt + "_EPOCH_" + str(epoch).zfill(4) + ".pt")
            trainer.save_model(last, epoch)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return last


if __name__ == "__main__":
    main()
