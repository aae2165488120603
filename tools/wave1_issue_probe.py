#!/usr/bin/env python3
"""Runs tools/wave1_issue_probe.hip: shader cycles per 16-MFMA k-step of a wave that owns its SIMD, against the number of VALU / LDS
instructions placed in the MFMA gaps (1024 = the matrix pipe never waits)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libwave1probe.so"))
dev = torch.device("cuda:0")
blocks, iters = 256, 2000
out = torch.zeros(blocks * 256, device=dev)
cyc = torch.zeros(blocks, dtype=torch.int64, device=dev)
for nv, nl, nl2 in ((0, 0, 0), (2, 0, 0), (4, 0, 0), (8, 0, 0), (12, 0, 0), (16, 0, 0), (0, 4, 0), (0, 0, 8), (0, 4, 8), (2, 4, 8), (4, 4, 8)):
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        rc = lib.wave1_probe_launch(nv, nl, nl2, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(cyc.data_ptr()), blocks, iters, st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.wave1_probe_launch(nv, nl, nl2, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(cyc.data_ptr()), blocks, iters, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    c = cyc.float().mean().item() / iters
    tf = blocks * 4 * iters * 16 * 4096.0 / ms / 1e9
    print("VALU/gap %2d  ds_read_b128/k-step %d  ds_read2_b32/k-step %d : %7.1f memtime ticks per k-step, %.3f ms, %.1f TFLOP/s issued" % (nv, nl, nl2, c, ms, tf), flush=True)
