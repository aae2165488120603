#!/usr/bin/env python3
"""Diagnostic (tools/mfma_valu_probe.hip): shader cycles per iteration of fp32-MFMA and vector streams that share a SIMD.
build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/mfma_valu_probe.hip -o tools/libmfmavalu.so"""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libmfmavalu.so"))
lib.mfma_valu_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
blocks, iters = 256, 400
out = torch.zeros(blocks * 8, dtype=torch.int64, device=dev)
sink = torch.zeros(4096 * 10, device=dev)
names = {0: "8 waves MFMA only (36 / iteration)", 1: "waves 0-3 MFMA, waves 4-7 vector (144 fma / iteration)", 2: "8 waves: 36 MFMA + 72 fma interleaved",
         3: "waves 4-7 vector only", 4: "waves 0-3 MFMA only", 5: "8 waves: 36 MFMA + 144 fma interleaved",
         6: "waves 0-3 MFMA, waves 4-7 LDS-DMA (12 x 1 KiB / iteration)", 7: "waves 4-7 LDS-DMA only", 8: "8 waves: 36 MFMA + 6 LDS-DMA interleaved",
         9: "waves 0-3 MFMA, waves 4-7 36 ds_read_b128 / iteration", 10: "waves 0-3: 36 MFMA + 144 fma, waves 4-7: 36 MFMA",
         11: "waves 0-3: 36 MFMA, waves 4-7: 36 MFMA + 144 fma", 12: "8 waves: 36 MFMA + 36 pk_fma interleaved",
         13: "8 waves: 36 MFMA + 72 pk_fma interleaved", 14: "8 waves: 36 MFMA + 144 pk_fma interleaved",
         15: "waves 0-3 MFMA, waves 4-7 vector (144 pk_fma / iteration)", 16: "waves 4-7 pk_fma only (144 / iteration)",
         17: "8 waves: 36 MFMA + 36 fma interleaved"}
for mode in (4, 3, 16, 0, 1, 15, 17, 2, 5, 12, 13, 14, 7, 6, 8, 9, 10, 11):
    for _ in range(2):
        out.zero_()
        lib.mfma_valu_launch(mode, iters, blocks, out.data_ptr(), sink.data_ptr(), None)
        torch.cuda.synchronize()
    t = out.cpu().reshape(blocks, 8).double().mean(0) / iters
    print("mode %d %-58s cycles / iteration per wave: %s" % (mode, names[mode], " ".join("%6.0f" % v for v in t)))
print("(36 MFMAs of 32 cycles = 1152; 144 fma of 4 cycles = 576)")
