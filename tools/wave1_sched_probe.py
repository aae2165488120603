#!/usr/bin/env python3
"""Runs tools/wave1_sched_probe.hip: placement of 4 ds_read_b128 + 8 ds_read2_b32 + 32 VALU in the 16 MFMA gaps of a k-step."""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libwave1sched.so"))
dev = torch.device("cuda:0")
blocks, iters = 256, 2000
out = torch.zeros(blocks * 256, device=dev)
cyc = torch.zeros(blocks, dtype=torch.int64, device=dev)
NAMES = {0: "MFMA only", 1: "12 LDS reads, one per gap", 2: "LDS one per gap + 32 VALU in two gaps (16+16)", 3: "LDS one per gap + 32 VALU spread 2 per gap",
         4: "32 VALU in two gaps, no LDS", 5: "12 LDS + 8 VALU in gap 0, 16 VALU in gaps 9 and 11 (first kernel)", 6: "LDS one per gap + 32 VALU in four gaps (8 each)",
         7: "LDS one per gap + 32 VALU in ONE gap", 8: "b128 one per gap, read2 two per gap, VALU 16+16", 9: "LDS in four gaps (2,2,4,4), VALU 16+16"}
for mode in range(10):
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        rc = lib.wave1_sched_launch(mode, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(cyc.data_ptr()), blocks, iters, st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.wave1_sched_launch(mode, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(cyc.data_ptr()), blocks, iters, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    c = cyc.float().mean().item() / iters
    tf = blocks * 4 * iters * 16 * 4096.0 / ms / 1e9
    print("mode %d  %-70s: %7.1f ticks per k-step, %.3f ms, %.1f TFLOP/s issued" % (mode, NAMES[mode], c, ms, tf), flush=True)
