// Diagnostic (second form of wave1_issue_probe): WHERE in the 16 MFMA gaps of a k-step should a single-wave-per-SIMD kernel put its
// LDS reads and VALU work?  MODE selects a placement; reports shader ticks per k-step (1024 = the matrix pipe never waits).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/wave1_sched_probe.hip -o tools/libwave1sched.so
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// per gap f: number of ds_read_b128, ds_read2_b32, VALU adds
struct Plan { int b128[16], r2[16], nv[16]; };
constexpr Plan plan(int mode) {
    Plan p{};
    for (int f = 0; f < 16; ++f) p.b128[f] = p.r2[f] = p.nv[f] = 0;
    if (mode == 1 || mode == 2 || mode == 3 || mode == 6 || mode == 7) {      // LDS one per gap (gaps 0..11)
        for (int f = 0; f < 4; ++f) p.b128[f] = 1;
        for (int f = 4; f < 12; ++f) p.r2[f] = 1;
    }
    if (mode == 2 || mode == 4) { p.nv[12] = 16; p.nv[13] = 16; }
    if (mode == 3) for (int f = 0; f < 16; ++f) p.nv[f] = 2;
    if (mode == 5) { p.b128[0] = 4; p.r2[0] = 8; p.nv[0] = 8; p.nv[9] = 16; p.nv[11] = 16; }     // the first Winograd kernel's layout
    if (mode == 6) { p.nv[12] = 8; p.nv[13] = 8; p.nv[14] = 8; p.nv[15] = 8; }
    if (mode == 7) { p.nv[14] = 32; }
    if (mode == 8) { for (int f = 0; f < 6; ++f) p.b128[f] = (f < 4), p.r2[f] = 0; for (int f = 0; f < 4; ++f) p.r2[4 + f] = 2; p.nv[12] = 16; p.nv[13] = 16; }
    if (mode == 9) { p.b128[0] = 2; p.b128[1] = 2; p.r2[2] = 4; p.r2[3] = 4; p.nv[12] = 16; p.nv[13] = 16; }
    return p;
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(float *out, long long *cyc, int iters) {
    extern __shared__ float lds[];
    constexpr Plan P = plan(MODE);
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)((i * 2654435761u >> 20) & 1023) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[16];
    for (int f = 0; f < 16; ++f)
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    float a[16], b[16], x[32];
    for (int i = 0; i < 16; ++i) {
        a[i] = lds[i * 64 + lane];
        b[i] = lds[1024 + i * 64 + lane];
        x[i] = lds[2048 + i * 64 + lane];
        x[16 + i] = lds[3072 + i * 64 + lane];
    }
    f32x4 la[4];
    float lb[16];
    for (int i = 0; i < 4; ++i) la[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 16; ++i) lb[i] = 0.f;
    const float *pa = lds + lane * 4, *pb = lds + 4096 + lane * 2 + 3;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        int ia = 0, ib = 0, iv = 0;
        const int o = (it & 7) * 1024;
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[f], b[f], acc[f], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int l = 0; l < P.b128[f]; ++l, ++ia) la[ia & 3] = *(const f32x4 *)(pa + o + (ia & 3) * 256);
#pragma unroll
            for (int l = 0; l < P.r2[f]; ++l, ++ib) {
                lb[2 * (ib & 7)] = pb[o + (ib & 7) * 36];
                lb[2 * (ib & 7) + 1] = pb[o + (ib & 7) * 36 + 1];
            }
#pragma unroll
            for (int v = 0; v < P.nv[f]; ++v, ++iv) x[iv & 31] = x[iv & 31] + x[(iv + 16) & 31];
            __builtin_amdgcn_sched_barrier(0);
        }
        // consume the loaded values once per k-step (keeps the loads alive; 1 VALU each is part of every mode's fixed cost)
        if (ia) x[0] += la[0][0] + la[1][1] + la[2][2] + la[3][3];
        if (ib) x[1] += lb[0] + lb[3] + lb[5] + lb[15];
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int f = 0; f < 16; ++f)
        for (int r = 0; r < 16; ++r) s += acc[f][r];
    for (int i = 0; i < 32; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
int launch(float *out, long long *cyc, int blocks, int iters, hipStream_t st) {
    (void)hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(256), 100 * 1024, st, out, cyc, iters);
    return (int)hipGetLastError();
}

extern "C" int wave1_sched_launch(int mode, float *out, long long *cyc, int blocks, int iters, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
        case 0: return launch<0>(out, cyc, blocks, iters, st);
        case 1: return launch<1>(out, cyc, blocks, iters, st);
        case 2: return launch<2>(out, cyc, blocks, iters, st);
        case 3: return launch<3>(out, cyc, blocks, iters, st);
        case 4: return launch<4>(out, cyc, blocks, iters, st);
        case 5: return launch<5>(out, cyc, blocks, iters, st);
        case 6: return launch<6>(out, cyc, blocks, iters, st);
        case 7: return launch<7>(out, cyc, blocks, iters, st);
        case 8: return launch<8>(out, cyc, blocks, iters, st);
        case 9: return launch<9>(out, cyc, blocks, iters, st);
    }
    return -1;
}
