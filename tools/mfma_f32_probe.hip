// Diagnostic: sustained rate + shader clock of fp32-MFMA loops whose operands are re-read from LDS every step, as the
// convolution does: v_mfma_f32_32x32x2_f32 (wave tile 32 couts x 128 px: 1 A + 4 B ds_read_b32 per 4 MFMAs) against
// v_mfma_f32_16x16x4_f32 (same wave tile and LDS bytes: 2 A + 8 B reads per 16 MFMAs = 2 k-steps of the other form).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/mfma_f32_probe.hip -o tools/libmfmaf32probe.so
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool LDS_OPS>
__global__ __launch_bounds__(256, 2) void mfma_f32_loop(float *out, int iters, unsigned seed) {
    __shared__ float lds[8192];
    unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = threadIdx.x; i < 8192; i += 256) {
        s = s * 1664525u + 1013904223u;
        lds[i] = (float)((int)(s >> 8) % 2001 - 1000) * 1e-3f;     // "random" mantissas: realistic toggling
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float acc_sum = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float a = lds[lane], b[4] = {lds[64 + lane], lds[128 + lane], lds[192 + lane], lds[256 + lane]};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                if (LDS_OPS) {
                    const int o = ((it * 16 + st) * 72) & 4095;
                    a = lds[o + lane];
#pragma unroll
                    for (int m = 0; m < 4; ++m) b[m] = lds[o + 1024 + m * 32 + lane];
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[m], acc[m], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc_sum += acc[i][r];
    } else {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i)
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        float a[2] = {lds[lane], lds[64 + lane]}, b[8];
        for (int m = 0; m < 8; ++m) b[m] = lds[128 + m * 64 + lane];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                if (LDS_OPS) {
                    const int o = ((it * 8 + st) * 144) & 4095;
                    a[0] = lds[o + lane];
                    a[1] = lds[o + 64 + lane];
#pragma unroll
                    for (int m = 0; m < 8; ++m) b[m] = lds[o + 1024 + m * 16 + lane];
                }
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int m = 0; m < 8; ++m) acc[n * 8 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[n], b[m], acc[n * 8 + m], 0, 0, 0);
            }
        }
        for (int i = 0; i < 16; ++i)
            for (int r = 0; r < 4; ++r) acc_sum += acc[i][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc_sum;
}

// flops per launch = blocks * 4 waves * iters * 262144  (shape 32: 16 steps x 4 MFMAs x 4096; shape 16: 8 x 16 x 2048)
extern "C" int mfma_f32_probe_launch(int shape, int lds_ops, float *out, int blocks, int iters, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (shape == 32 && lds_ops) hipLaunchKernelGGL((mfma_f32_loop<32, true>), dim3(blocks), dim3(256), 0, st, out, iters, 12345u);
    else if (shape == 32) hipLaunchKernelGGL((mfma_f32_loop<32, false>), dim3(blocks), dim3(256), 0, st, out, iters, 12345u);
    else if (lds_ops) hipLaunchKernelGGL((mfma_f32_loop<16, true>), dim3(blocks), dim3(256), 0, st, out, iters, 12345u);
    else hipLaunchKernelGGL((mfma_f32_loop<16, false>), dim3(blocks), dim3(256), 0, st, out, iters, 12345u);
    return (int)hipGetLastError();
}
