// Diagnostic: bare-loop rate of the split-fp16 product pipeline per K = 64 channels and accumulator tile:
//   variant 0: 12 x v_mfma_f32_32x32x16_f16                      (a_hi*b_hi, a_hi*b_lo, a_lo*b_hi: the shipped f16x3 scheme)
//   variant 1:  4 x v_mfma_f32_32x32x16_f16 + 2 x v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 (e2m3) operands
//   variant 2:  4 x f16 + 2 x scaled fp8 (e4m3) operands
// Operands live in registers (pseudo-random bit patterns), 4 independent accumulators per wave, all CUs busy.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/mfma_mix_probe.hip -o tools/libmfmamix.so
#include <hip/hip_runtime.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int VARIANT>
__global__ __launch_bounds__(256) void mix_loop(float *out, int iters) {
    h8 ah[4], bh[4], al[4], bl[4];
    i32x8 aq[2], bq[2];
    unsigned s = threadIdx.x * 2654435761u + 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            ah[i][e] = (_Float16)((float)((int)(rnd() >> 20) - 2048) * (1.0f / 2048.0f));
            bh[i][e] = (_Float16)((float)((int)(rnd() >> 20) - 2048) * (1.0f / 2048.0f));
            al[i][e] = (_Float16)((float)((int)(rnd() >> 20) - 2048) * (1.0f / 4194304.0f));
            bl[i][e] = (_Float16)((float)((int)(rnd() >> 20) - 2048) * (1.0f / 4194304.0f));
        }
    for (int i = 0; i < 2; ++i)
        for (int e = 0; e < 8; ++e) {
            aq[i][e] = (int)(rnd() & (VARIANT == 2 ? 0x3f3f3f3fu : 0xffffffffu));   // fp8: keep exponents small (no NaN/inf patterns)
            bq[i][e] = (int)(rnd() & (VARIANT == 2 ? 0x3f3f3f3fu : 0xffffffffu));
        }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int scale = 0x7f7f7f7f;          // E8M0 = 127 -> 2^0 in every byte
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[(i + k) & 3], bh[k], acc[i], 0, 0, 0);
                if (VARIANT == 0) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[(i + k) & 3], bl[k], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[(i + k) & 3], bh[k], acc[i], 0, 0, 0);
                }
            }
        if (VARIANT != 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                constexpr int F = VARIANT == 1 ? 2 : 0;         // 2 = fp6 e2m3, 0 = fp8 e4m3
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aq[i & 1], bq[0], acc[i], F, F, 0, scale, 0, scale);
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aq[(i + 1) & 1], bq[1], acc[i], F, F, 0, scale, 0, scale);
            }
        }
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) t += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

// algorithmic flops per launch = blocks * 4 waves * iters * 4 accumulators * (2 * 32 * 32 * 64)
extern "C" int mfma_mix_launch(int variant, float *out, int blocks, int iters, void *stream) {
    if (variant == 0) hipLaunchKernelGGL(mix_loop<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
    else if (variant == 1) hipLaunchKernelGGL(mix_loop<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
    else hipLaunchKernelGGL(mix_loop<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
    return (int)hipGetLastError();
}
