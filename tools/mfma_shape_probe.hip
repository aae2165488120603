// Diagnostic: sustained rate of bare fp16 MFMA loops (operands in registers) for the two tile shapes, all CUs busy.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/mfma_shape_probe.hip -o tools/libmfmaprobe.so
#include <hip/hip_runtime.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float seed) {
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (_Float16)(seed * (float)((threadIdx.x * 7 + i * 13 + e * 3) % 17 - 8) * 0.01f);
            b[i][e] = (_Float16)(seed * (float)((threadIdx.x * 5 + i * 11 + e * 7) % 19 - 9) * 0.01f);
        }
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + k) & 3], b[k], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + k) & 3], b[k], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// flops per launch = blocks * 4 waves * iters * (shape 32: 16 MFMAs * 32768 | shape 16: 32 MFMAs * 16384)
extern "C" int mfma_probe_launch(int shape, float *out, int blocks, int iters, void *stream) {
    if (shape == 32)
        hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 1.0f);
    else
        hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 1.0f);
    return (int)hipGetLastError();
}
