#!/usr/bin/env python3
"""Bring-up check of the scaled fp8 MFMA operand convention and the fp8 conversion (tools/fp8_mfma_bringup.hip)."""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libfp8bringup.so"))
vp = ctypes.c_void_p
lib.fp8_mfma_launch.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, vp]
lib.cvt_fp8_launch.argtypes = [vp, vp, ctypes.c_int, vp]
dev = torch.device("cuda:0")
st = vp(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
# 1. conversion vs torch's OCP e4m3fn
x = torch.cat([torch.randn(4096) * 3, torch.tensor([0.0, 1e-4, 0.001, 0.002, 0.0039, 447.0, 448.0, 500.0, 1e4, -1e4, 0.0175, -0.3])]).to(dev)
x = x[: x.numel() // 2 * 2].contiguous()
y = torch.empty(x.numel(), dtype=torch.uint8, device=dev)
lib.cvt_fp8_launch(x.data_ptr(), y.data_ptr(), x.numel(), st)
torch.cuda.synchronize()
got = y.view(torch.float8_e4m3fn).to(torch.float32).cpu()
want = x.cpu().to(torch.float8_e4m3fn).to(torch.float32)
bad = (got != want) & ~(got.isnan() & want.isnan())
print("cvt_pk_fp8_f32 vs torch e4m3fn: mismatches %d of %d" % (int(bad.sum()), x.numel()))
for i in torch.nonzero(bad).flatten()[:8]:
    print("   x=%g  hw=%g  torch=%g" % (float(x[i]), float(got[i]), float(want[i])))
print("   large values:", [(float(a), float(b)) for a, b in zip(x[-8:-2].cpu(), got[-8:-2])])
# 2. MFMA with my operand convention
A = (torch.randint(-8, 9, (32, 64)).float() * 0.25)
B = (torch.randint(-8, 9, (32, 64)).float() * 0.5)          # B[col][k]
a8 = A.to(torch.float8_e4m3fn).view(torch.uint8).to(dev).contiguous()
b8 = B.to(torch.float8_e4m3fn).view(torch.uint8).to(dev).contiguous()
d = torch.zeros(32, 32, device=dev)
for sa, sb in ((127, 127), (127, 116), (116, 127), (120, 130)):
    s4 = lambda s: s * 0x01010101          # noqa: E731  the same E8M0 byte in all four byte lanes
    lib.fp8_mfma_launch(a8.data_ptr(), b8.data_ptr(), d.data_ptr(), s4(sa), s4(sb), st)
    torch.cuda.synchronize()
    want = (A @ B.t()) * 2.0 ** (sa - 127) * 2.0 ** (sb - 127)
    print("scaled fp8 MFMA scale_a=%d scale_b=%d: max|hw - ref| = %.3e (ref max %.1f)" % (sa, sb, float((d.cpu() - want).abs().max()), float(want.abs().max())))
