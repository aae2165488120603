// Diagnostic: how much other work fits between the MFMAs of ONE wave that has its SIMD to itself (the Winograd kernel's situation:
// 16 accumulators x 16 registers per wave, one 4-wave workgroup per CU)?  A k-step = 16 independent v_mfma_f32_32x32x2_f32; after each
// MFMA the wave issues NV independent VALU adds and, after the first one, NL ds_read_b128 + NL2 ds_read2_b32.  Reports shader cycles
// per k-step (s_memtime), 1024 = the matrix pipe never waits.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/wave1_issue_probe.hip -o tools/libwave1probe.so
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NL, int NL2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(float *out, long long *cyc, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)((i * 2654435761u >> 20) & 1023) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[16];
    for (int f = 0; f < 16; ++f)
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    float a[16], b[16], x[16];
    for (int i = 0; i < 16; ++i) {
        a[i] = lds[i * 64 + lane];
        b[i] = lds[1024 + i * 64 + lane];
        x[i] = lds[2048 + i * 64 + lane];
    }
    f32x4 la[NL > 0 ? NL : 1];
    float lb[NL2 > 0 ? 2 * NL2 : 1];
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[f], b[f], acc[f], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (f == 0) {
#pragma unroll
                for (int l = 0; l < NL; ++l) la[l] = *(const f32x4 *)(lds + ((it * 7 + l) & 15) * 256 + lane * 4);
#pragma unroll
                for (int l = 0; l < NL2; ++l) {
                    lb[2 * l] = lds[4096 + ((it + l) & 7) * 160 + lane * 2 + 3];
                    lb[2 * l + 1] = lds[4096 + ((it + l) & 7) * 160 + lane * 2 + 4];
                }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) x[(f + v) & 15] = x[(f + v) & 15] + b[(f + v + 3) & 15];
            if (f == 15) {
#pragma unroll
                for (int l = 0; l < NL; ++l) x[l & 15] += la[l][0] + la[l][3];
#pragma unroll
                for (int l = 0; l < 2 * NL2; ++l) x[l & 15] += lb[l];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int f = 0; f < 16; ++f)
        for (int r = 0; r < 16; ++r) s += acc[f][r];
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CASE(nv, nl, nl2) \
    if (NV == nv && NL == nl && NL2 == nl2) { hipLaunchKernelGGL((probe<nv, nl, nl2>), dim3(blocks), dim3(256), 100 * 1024, st, out, cyc, iters); return (int)hipGetLastError(); }

extern "C" int wave1_probe_launch(int NV, int NL, int NL2, float *out, long long *cyc, int blocks, int iters, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    static bool once = false;
    if (!once) {
        once = true;
#define ATTR(nv, nl, nl2) hipFuncSetAttribute((const void *)probe<nv, nl, nl2>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        ATTR(0, 0, 0) ATTR(2, 0, 0) ATTR(4, 0, 0) ATTR(8, 0, 0) ATTR(12, 0, 0) ATTR(16, 0, 0) ATTR(0, 4, 0) ATTR(0, 0, 8) ATTR(0, 4, 8) ATTR(2, 4, 8) ATTR(4, 4, 8)
    }
    CASE(0, 0, 0) CASE(2, 0, 0) CASE(4, 0, 0) CASE(8, 0, 0) CASE(12, 0, 0) CASE(16, 0, 0) CASE(0, 4, 0) CASE(0, 0, 8) CASE(0, 4, 8) CASE(2, 4, 8) CASE(4, 4, 8)
    return -1;
}
