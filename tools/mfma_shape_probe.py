#!/usr/bin/env python3
"""Diagnostic: bare fp16 MFMA loops, 32x32x16 vs 16x16x32, sustained TFLOP/s and shader clock (power-limited regime)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libmfmaprobe.so"))
lib.mfma_probe_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
probe = ctypes.CDLL(os.path.join(ROOT, "tools", "libclockprobe.so"))
probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p]

dev = torch.device("cuda:0")
blocks, iters = 256 * 8, 4000
out = torch.empty(blocks * 256, device=dev)
clk = torch.zeros(2, dtype=torch.int64, device=dev)
side = torch.cuda.Stream()
for rep in range(2):
    for shape in (32, 16):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            lib.mfma_probe_launch(shape, out.data_ptr(), blocks, iters, st)
        torch.cuda.synchronize()
        probe.clock_probe_launch(clk.data_ptr(), int(40 * 1e5), ctypes.c_void_p(side.cuda_stream))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 12
        for _ in range(n):
            lib.mfma_probe_launch(shape, out.data_ptr(), blocks, iters, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        flops = n * blocks * 4 * iters * 16 * 32768.0
        c, r = [int(v) for v in clk.cpu()]
        ghz = c / (r / 100e6) / 1e9
        tf = flops / ms / 1e9
        print("shape %2d: %.1f ms  %.0f TFLOP/s  clock %.3f GHz  -> %.0f %% of the pipe at that clock" % (shape, ms, tf, ghz, 100 * tf / (2500 * ghz / 2.4)))
