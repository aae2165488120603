#!/usr/bin/env python3
"""Runs tools/valu_rate_probe.hip (hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/valu_rate_probe.hip -o tools/libvalurate.so):
nanoseconds and s_memtime ticks per fp32 vector instruction of a wave, with one and two waves per SIMD and nothing else on the CU."""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libvalurate.so"))
dev = torch.device("cuda:0")
blocks, iters = 256, 4000
names = {0: "8 independent chains, inline constant", 1: "one dependent chain", 2: "8 independent chains, 32-bit literal", 3: "two chains",
         4: "add / sub / fmac mix with literals (transform-like)", 5: "8 independent chains + s_nop 0 after each"}
out = torch.zeros(blocks * 512, device=dev)
cyc = torch.zeros(blocks, dtype=torch.int64, device=dev)
for threads in (256, 512, 1024):
    for mode in range(6):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        args = (mode, threads, blocks, iters, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(cyc.data_ptr()), st)
        for _ in range(2):
            assert lib.valu_rate_launch(*args) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.valu_rate_launch(*args)
        e1.record()
        torch.cuda.synchronize()
        n = iters * 64
        print("%d waves/SIMD  mode %d %-52s %.2f ns per instruction per wave, %.2f ticks" % (threads // 256, mode, names[mode], e0.elapsed_time(e1) * 1e6 / n,
                                                                                           cyc.float().mean().item() / n), flush=True)
