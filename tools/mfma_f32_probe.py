#!/usr/bin/env python3
"""Diagnostic: fp32-MFMA loops (32x32x2 vs 16x16x4; operands in registers vs re-read from LDS like the convolution):
sustained TFLOP/s and shader clock.  Answers: what is the practical fp32-MFMA roof of this chip under a realistic
operand stream, and does the 16x16x4 form (4x less accumulator traffic per FLOP) buy clock?"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libmfmaf32probe.so"))
lib.mfma_f32_probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
probe = ctypes.CDLL(os.path.join(ROOT, "tools", "libclockprobe.so"))
probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p]

dev = torch.device("cuda:0")
blocks, iters = 256 * 8, 600
out = torch.empty(blocks * 256, device=dev)
clk = torch.zeros(2, dtype=torch.int64, device=dev)
side = torch.cuda.Stream()
for rep in range(2):
    for shape in (32, 16):
        for lds_ops in (0, 1):
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(3):
                lib.mfma_f32_probe_launch(shape, lds_ops, out.data_ptr(), blocks, iters, st)
            torch.cuda.synchronize()
            probe.clock_probe_launch(clk.data_ptr(), int(40 * 1e5), ctypes.c_void_p(side.cuda_stream))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 12
            for _ in range(n):
                lib.mfma_f32_probe_launch(shape, lds_ops, out.data_ptr(), blocks, iters, st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            flops = n * blocks * 4 * iters * 262144.0
            c, r = [int(v) for v in clk.cpu()]
            ghz = c / (r / 100e6) / 1e9
            tf = flops / ms / 1e9
            print("%s operands %s: %.1f ms  %.1f TFLOP/s  clock %.3f GHz  -> %.0f %% of the pipe at that clock"
                  % ("32x32x2" if shape == 32 else "16x16x4", "from LDS " if lds_ops else "in registers", ms, tf, ghz,
                     100 * tf / (157.3 * ghz / 2.4)))
