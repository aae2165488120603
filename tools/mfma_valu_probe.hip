// Diagnostic: do vector instructions of one wave issue beside the fp32 MFMAs of the other wave of the same SIMD?  One 512-thread
// workgroup per CU (two waves per SIMD: w and w + 4).  Modes:
//   0  all 8 waves: 36 independent v_mfma_f32_16x16x4_f32 per iteration
//   1  waves 0..3 MFMA only, waves 4..7 vector only (144 v_fma_f32 per iteration, 8 independent chains)
//   2  all 8 waves: 36 MFMAs with 2 v_fma_f32 behind each (72 per iteration)
//   3  waves 4..7 vector only, waves 0..3 idle
//   4  waves 0..3 MFMA only, waves 4..7 idle
//   5  all 8 waves: 36 MFMAs with 4 v_fma_f32 behind each (144 per iteration)
//   6  waves 0..3 MFMA only, waves 4..7 LDS-DMA only (12 global_load_lds_dwordx4 of 1 KiB per iteration from an L2-resident buffer)
//   7  waves 4..7 LDS-DMA only, waves 0..3 idle
//   8  all 8 waves: 36 MFMAs with one LDS-DMA behind every sixth (6 per iteration)
//   9  waves 0..3 MFMA only, waves 4..7 LDS reads only (36 ds_read_b128 per iteration)
//  10  waves 0..3: 36 MFMAs with 4 v_fma_f32 behind each, waves 4..7: 36 MFMAs only   (all the vector work on the favoured wave)
//  11  waves 0..3: 36 MFMAs only, waves 4..7: 36 MFMAs with 4 v_fma_f32 behind each
//  12  all 8 waves: 36 MFMAs with 1 v_pk_fma_f32 behind each (36 per iteration = the arithmetic of mode 2)
//  13  all 8 waves: 36 MFMAs with 2 v_pk_fma_f32 behind each (72 per iteration = the arithmetic of mode 5)
//  14  all 8 waves: 36 MFMAs with 4 v_pk_fma_f32 behind each (144 per iteration)
//  15  waves 0..3 MFMA only, waves 4..7 vector only (144 v_pk_fma_f32 per iteration)
//  16  waves 4..7 v_pk_fma_f32 only, waves 0..3 idle
//  17  all 8 waves: 36 MFMAs with 1 v_fma_f32 behind each (36 per iteration)
// Each wave reports its shader cycles per iteration (s_memtime) into out[block][wave].
#include <hip/hip_runtime.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 2) void mfma_valu_kernel(int mode, int iters, unsigned long long *out, float *sink) {
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (mode == 10) mode = wid < 4 ? 5 : 0;
    if (mode == 11) mode = wid < 4 ? 0 : 5;
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int npk = mode == 12 ? 1 : mode == 13 ? 2 : mode == 14 ? 4 : 0;
    const bool pkonly = (mode == 15 || mode == 16) && wid >= 4;
    const bool mfpk = npk > 0;
    const bool mf1 = mode == 17;
    if (mode == 15 && wid < 4) mode = 4;
    const bool mf = mode == 0 || mode == 2 || mode == 5 || mode == 8 || ((mode == 1 || mode == 4 || mode == 6 || mode == 9) && wid < 4);
    const bool dm = (mode == 6 || mode == 7) && wid >= 4;
    const bool lr = mode == 9 && wid >= 4;
    const float *gsrc = sink + 4096 + wid * 4096;          // 16 KiB per wave, L2-resident after the first pass
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds + wid * 16384;
    const int voff = (threadIdx.x & 63) * 16;
    const bool va = mode == 2 || mode == 5 || ((mode == 1 || mode == 3) && wid >= 4);
    const int nv = mode == 2 ? 2 : mode == 5 ? 4 : 0;
    f32x4 acc[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    float c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = (float)i;
    asm volatile("" : "+v"(a), "+v"(b));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 8) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                if (i % 6 == 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    const unsigned m0v = lds0 + (i / 6) * 1024;
                    const float *base = gsrc + (i / 6) * 256;
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(m0v) : "memory", "m0");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else if (dm) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const unsigned m0v = lds0 + i * 1024;
                const float *base = gsrc + i * 256;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(m0v) : "memory", "m0");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else if (lr) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 r = {0.f, 0.f, 0.f, 0.f};
        const f4 *l4 = (const f4 *)lds + wid * 1024 + (threadIdx.x & 63);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                f4 t = l4[(i & 15) * 64];
                asm volatile("" : "+v"(t));
                r += t;
            }
        }
        c[0] += r[0] + r[1] + r[2] + r[3];
    } else if (mf && nv == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 36; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
    } else if (mf && nv == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                c[(2 * i) & 7] = __builtin_fmaf(c[(2 * i) & 7], 1.0001f, 0.5f);
                c[(2 * i + 1) & 7] = __builtin_fmaf(c[(2 * i + 1) & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (mf && nv == 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 4; ++k) c[(4 * i + k) & 7] = __builtin_fmaf(c[(4 * i + k) & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (mfpk) {
        f2 cp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) cp[i] = f2{(float)i, (float)i + 0.5f};
        f2 k1 = {1.0001f, 1.0002f}, k2 = {0.5f, 0.25f};
        asm volatile("" : "+v"(k1), "+v"(k2));
        auto body = [&](auto NPK) __attribute__((always_inline)) {
            constexpr int n = decltype(NPK)::value;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 36; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < n; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(cp[(n * i + k) & 7]) : "v"(k1), "v"(k2));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (npk == 1) body(std::integral_constant<int, 1>{});
        else if (npk == 2) body(std::integral_constant<int, 2>{});
        else body(std::integral_constant<int, 4>{});
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] += cp[i][0] + cp[i][1];
    } else if (mf1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                c[i & 7] = __builtin_fmaf(c[i & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (pkonly) {
        f2 cp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) cp[i] = f2{(float)i, (float)i + 0.5f};
        f2 k1 = {1.0001f, 1.0002f}, k2 = {0.5f, 0.25f};
        asm volatile("" : "+v"(k1), "+v"(k2));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 144; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(cp[i & 7]) : "v"(k1), "v"(k2));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] += cp[i][0] + cp[i][1];
    } else if (va) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 144; ++i) {
                c[i & 7] = __builtin_fmaf(c[i & 7], 1.0001f, 0.5f);
                if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 36; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i];
    if (s == 123.456f) sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wid] = (t1 - t0);
}

extern "C" int mfma_valu_launch(int mode, int iters, int blocks, unsigned long long *out, float *sink, void *stream) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)mfma_valu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
        attr = true;
    }
    hipLaunchKernelGGL(mfma_valu_kernel, dim3(blocks), dim3(512), 8 * 16384, (hipStream_t)stream, mode, iters, out, sink);
    return (int)hipGetLastError();
}
