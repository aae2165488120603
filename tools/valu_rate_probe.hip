// How fast does ONE wave per SIMD issue plain fp32 vector instructions on gfx950?  (tools/valu_rate_probe.py)
// The transform phases of csrc/ssm_wino5.hip / ssm_wino7.hip run with one wave per SIMD and no MFMA beside them; their measured cost is
// ~10 cycles per vector instruction.  Modes: 0 independent (8 chains, inline constants), 1 one dependent chain, 2 independent with 32-bit
// literal constants (8-byte encodings), 3 two chains, 4 independent v_add/v_sub mix with literals as the transforms issue them,
// 5 = mode 0 with an s_nop 0 after every instruction.  threads = 256 (one wave per SIMD) or 512 (two).
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int MODE>
__global__ void valu_rate_kernel(float *out, long long *cyc, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float x = out[threadIdx.x & 63];
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0)
                asm volatile("v_fmac_f32 %0, 0.5, %8\n v_fmac_f32 %1, 0.5, %8\n v_fmac_f32 %2, 0.5, %8\n v_fmac_f32 %3, 0.5, %8\n"
                             "v_fmac_f32 %4, 0.5, %8\n v_fmac_f32 %5, 0.5, %8\n v_fmac_f32 %6, 0.5, %8\n v_fmac_f32 %7, 0.5, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
            if (MODE == 1)
                asm volatile("v_fmac_f32 %0, 0.5, %0\n v_fmac_f32 %0, 0.5, %0\n v_fmac_f32 %0, 0.5, %0\n v_fmac_f32 %0, 0.5, %0\n"
                             "v_fmac_f32 %0, 0.5, %0\n v_fmac_f32 %0, 0.5, %0\n v_fmac_f32 %0, 0.5, %0\n v_fmac_f32 %0, 0.5, %0\n"
                             : "+v"(a0) : "v"(x));
            if (MODE == 2)
                asm volatile("v_fmac_f32 %0, 0x40a80000, %8\n v_fmac_f32 %1, 0x40a80000, %8\n v_fmac_f32 %2, 0x40a80000, %8\n v_fmac_f32 %3, 0x40a80000, %8\n"
                             "v_fmac_f32 %4, 0x40a80000, %8\n v_fmac_f32 %5, 0x40a80000, %8\n v_fmac_f32 %6, 0x40a80000, %8\n v_fmac_f32 %7, 0x40a80000, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
            if (MODE == 3)
                asm volatile("v_fmac_f32 %0, 0.5, %0\n v_fmac_f32 %1, 0.5, %1\n v_fmac_f32 %0, 0.5, %0\n v_fmac_f32 %1, 0.5, %1\n"
                             "v_fmac_f32 %0, 0.5, %0\n v_fmac_f32 %1, 0.5, %1\n v_fmac_f32 %0, 0.5, %0\n v_fmac_f32 %1, 0.5, %1\n"
                             : "+v"(a0), "+v"(a1) : "v"(x));
            if (MODE == 4)
                asm volatile("v_sub_f32 %0, %0, %1\n v_sub_f32 %2, %3, %4\n v_fmac_f32 %0, 0x40a80000, %2\n v_add_f32 %5, %6, %7\n"
                             "v_fmac_f32 %5, 0xc0880000, %3\n v_mul_f32 %6, 0x40200000, %6\n v_fma_f32 %7, %1, 0.5, -%6\n v_add_f32 %4, %5, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
            if (MODE == 5)
                asm volatile("v_fmac_f32 %0, 0.5, %8\n s_nop 0\n v_fmac_f32 %1, 0.5, %8\n s_nop 0\n v_fmac_f32 %2, 0.5, %8\n s_nop 0\n v_fmac_f32 %3, 0.5, %8\n s_nop 0\n"
                             "v_fmac_f32 %4, 0.5, %8\n s_nop 0\n v_fmac_f32 %5, 0.5, %8\n s_nop 0\n v_fmac_f32 %6, 0.5, %8\n s_nop 0\n v_fmac_f32 %7, 0.5, %8\n s_nop 0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

extern "C" int valu_rate_launch(int mode, int threads, int blocks, int iters, float *out, long long *cyc, hipStream_t st) {
    switch (mode) {
    case 0: valu_rate_kernel<0><<<blocks, threads, 0, st>>>(out, cyc, iters); break;
    case 1: valu_rate_kernel<1><<<blocks, threads, 0, st>>>(out, cyc, iters); break;
    case 2: valu_rate_kernel<2><<<blocks, threads, 0, st>>>(out, cyc, iters); break;
    case 3: valu_rate_kernel<3><<<blocks, threads, 0, st>>>(out, cyc, iters); break;
    case 4: valu_rate_kernel<4><<<blocks, threads, 0, st>>>(out, cyc, iters); break;
    case 5: valu_rate_kernel<5><<<blocks, threads, 0, st>>>(out, cyc, iters); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}
