#!/usr/bin/env python3
"""Diagnostic: bare-loop rate of the product pipeline per K=64 for the shipped f16x3 scheme vs fp16 main + block-scaled
fp6 / fp8 correction products (tools/mfma_mix_probe.hip).  Prints algorithmic TFLOP/s (2*M*N*K per product) and the shader clock."""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libmfmamix.so"))
lib.mfma_mix_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
probe = ctypes.CDLL(os.path.join(ROOT, "tools", "libclockprobe.so"))
probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p]

dev = torch.device("cuda:0")
blocks, iters = 256 * 8, 1500
out = torch.empty(blocks * 256, device=dev)
clk = torch.zeros(2, dtype=torch.int64, device=dev)
side = torch.cuda.Stream()
names = {0: "12 x f16 (f16x3)", 1: "4 x f16 + 2 x scaled fp6", 2: "4 x f16 + 2 x scaled fp8"}
for rep in range(2):
    for variant in (0, 1, 2):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            lib.mfma_mix_launch(variant, out.data_ptr(), blocks, iters, st)
        torch.cuda.synchronize()
        probe.clock_probe_launch(clk.data_ptr(), int(40 * 1e5), ctypes.c_void_p(side.cuda_stream))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 12
        for _ in range(n):
            lib.mfma_mix_launch(variant, out.data_ptr(), blocks, iters, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        flops = n * blocks * 4.0 * iters * 4 * (2.0 * 32 * 32 * 64)
        c, r = [int(v) for v in clk.cpu()]
        print("%-28s %7.1f ms  %6.0f TFLOP/s algorithmic  clock %.3f GHz  finite=%s"
              % (names[variant], ms, flops / ms / 1e9, c / (r / 100e6) / 1e9, bool(torch.isfinite(out).all())))
