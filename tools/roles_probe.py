#!/usr/bin/env python3
"""Diagnostic (tools/roles_probe.hip): shader cycles per chunk of a four-waves-per-SIMD roles form - two matrix waves (18 fp32 MFMAs + 10 LDS operand
reads each) and two helper waves (transform mix + LDS-DMA) per SIMD, one barrier per chunk.  36 MFMAs per SIMD and chunk = 1152 cycles.
build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/roles_probe.hip -o tools/librolesprobe.so"""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "librolesprobe.so"))
lib.roles_launch.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
blocks, iters = 256, 400
out = torch.zeros(blocks * 16, dtype=torch.int64, device=dev)
sink = torch.zeros(4096 * 10, device=dev)
for mode, nv, nd, what in ((0, 0, 0, "matrix waves only (2 x 18 MFMA + 2 x 10 ds_read_b128 per SIMD)"), (2, 42, 0, "helper waves only, 42 fma + 17 LDS each"),
                           (1, 0, 0, "both roles, helpers: 17 LDS instructions only"), (1, 21, 0, "both roles, helpers 21 fma"), (1, 42, 0, "both roles, helpers 42 fma"),
                           (1, 42, 4, "both roles, helpers 42 fma + 4 LDS-DMA"), (1, 84, 4, "both roles, helpers 84 fma + 4 LDS-DMA"), (1, 84, 8, "both roles, helpers 84 fma + 8 LDS-DMA"), (1, 60, 6, "both roles, helpers 60 fma + 6 LDS-DMA"), (2, 84, 8, "helper waves only, 84 fma + 8 LDS-DMA"), (3, 0, 0, "16 mixed waves: 18 MFMA + reads, no transform work"),
                           (3, 12, 1, "16 mixed waves: 18 MFMA + 12 fma + 5 LDS + 1 DMA each"), (3, 24, 1, "16 mixed waves: 18 MFMA + 24 fma + 5 LDS + 1 DMA each")):
    for _ in range(2):
        out.zero_()
        assert lib.roles_launch(mode, iters, nv, nd, blocks, out.data_ptr(), sink.data_ptr(), None) == 0
        torch.cuda.synchronize()
    t = out.cpu().reshape(blocks, 16).double().mean(0) / iters
    print("mode %d %-62s cycles / chunk: matrix waves %6.0f  helper waves %6.0f" % (mode, what, t[:8].mean(), t[8:].mean()))
print("(36 MFMAs per SIMD and chunk = 1152 cycles; the shipped 64-cout form: 3340 per 72)")
