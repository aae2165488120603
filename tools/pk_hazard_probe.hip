// Probe for a gfx950 forwarding hazard seen in ssm_elem.hip's flow kernel (profiles/README.md, "packed-fp32 hazard"):
//
//     v_pk_mul_f32 v[6:7], v[6:7], v[28:29]
//     s_nop 0
//     v_pk_add_f32 v[6:7], v[14:15], v[6:7] op_sel:[0,1] op_sel_hi:[1,0]     ; lo = v14 + v7, hi = v15 + v6
//
// Under another queue's block-scaled MFMA kernel the low result came out as v14 + 0 in lanes 48..63 of rare waves.
// The victim kernel runs that sequence (NOPS wait states between the two packed ops, or the unswizzled form) against
// scalar v_mul/v_add on the same values and counts mismatching lanes; the co-runner kernels keep the matrix pipes busy
// from a second stream.  Build: hipcc --offload-arch=gfx950 -O3 tools/pk_hazard_probe.hip -o tools/pk_hazard_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef int i8v __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

// VARIANT 0: swizzled consumer, s_nop 0 (what the compiler emitted); 1: s_nop 1; 2: s_nop 3; 3: no swizzle, s_nop 0;
// 4: swizzled consumer, no s_nop at all; 5: swizzled, producer is two scalar v_mul_f32 instead of v_pk_mul_f32
template <int VARIANT>
__global__ __launch_bounds__(256) void victim(const float *__restrict__ in, unsigned *__restrict__ bad, unsigned long long *__restrict__ lanes,
                                              int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float a = in[(tid * 4 + 0) & 0xffff], b = in[(tid * 4 + 1) & 0xffff], c = in[(tid * 4 + 2) & 0xffff], d = in[(tid * 4 + 3) & 0xffff];
    unsigned nbad = 0;
    for (int i = 0; i < iters; ++i) {
        f2 x = {a, b}, y = {c, d}, k = {0.1875f + (float)(i & 7), -0.0625f};
        float p_lo, p_hi, want_lo, want_hi;      // scalar reference, kept out of the vectoriser's hands
        asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %5" : "=&v"(p_lo), "=&v"(p_hi) : "v"(x.x), "v"(x.y), "v"(k.x), "v"(k.y));
        if (VARIANT == 3)
            asm volatile("v_add_f32 %0, %2, %4\n\tv_add_f32 %1, %3, %5" : "=&v"(want_lo), "=&v"(want_hi) : "v"(y.x), "v"(y.y), "v"(p_lo), "v"(p_hi));
        else
            asm volatile("v_add_f32 %0, %2, %4\n\tv_add_f32 %1, %3, %5" : "=&v"(want_lo), "=&v"(want_hi) : "v"(y.x), "v"(y.y), "v"(p_hi), "v"(p_lo));
        if (VARIANT == 0)
            asm volatile("v_pk_mul_f32 %0, %0, %2\n\ts_nop 0\n\tv_pk_add_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(x) : "v"(y), "v"(k));
        else if (VARIANT == 1)
            asm volatile("v_pk_mul_f32 %0, %0, %2\n\ts_nop 1\n\tv_pk_add_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(x) : "v"(y), "v"(k));
        else if (VARIANT == 2)
            asm volatile("v_pk_mul_f32 %0, %0, %2\n\ts_nop 3\n\tv_pk_add_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(x) : "v"(y), "v"(k));
        else if (VARIANT == 3)
            asm volatile("v_pk_mul_f32 %0, %0, %2\n\ts_nop 0\n\tv_pk_add_f32 %0, %1, %0" : "+v"(x) : "v"(y), "v"(k));
        else if (VARIANT == 4)
            asm volatile("v_pk_mul_f32 %0, %0, %2\n\tv_pk_add_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(x) : "v"(y), "v"(k));
        else {
            float xl = x.x, xh = x.y;
            asm volatile("v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3" : "+v"(xl), "+v"(xh) : "v"(k.x), "v"(k.y));
            x = f2{xl, xh};
            asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(x) : "v"(y));
        }
        if (__float_as_uint(x.x) != __float_as_uint(want_lo) || __float_as_uint(x.y) != __float_as_uint(want_hi)) {
            ++nbad;
            atomicOr(lanes, 1ull << (threadIdx.x & 63));
        }
        a = a * 1.0001f + 0.001f;
        b = b * 0.9999f - 0.002f;
        c += 0.003f;
        d -= 0.001f;
    }
    if (nbad) atomicAdd(bad, nbad);
}

// co-runners: 0 = block-scaled fp8 MFMA (the f16f8 convolutions' instruction), 1 = fp16 MFMA, 2 = plain VALU fma,
// 3 = fp16 MFMA fed by ds_read_b128 (the convolutions' inner loop shape), 4 = ds_read_b128 only
template <int KIND>
__global__ __launch_bounds__(256) void corunner(float *__restrict__ out, int iters) {
    f16v acc0 = {0}, acc1 = {0};
    float s = threadIdx.x * 1e-3f;
    __shared__ __attribute__((aligned(16))) char lds[32768];
    if (KIND >= 3) {
        for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<int *>(lds)[i] = 0x3c003c00 + i;
        __syncthreads();
        h8 a = {0}, b = {0};
        for (int i = 0; i < iters; ++i) {
            const int off = ((i * 4096 + threadIdx.x * 16) & 32767) & ~15;
            const h8 x0 = *reinterpret_cast<const h8 *>(lds + off);
            const h8 x1 = *reinterpret_cast<const h8 *>(lds + ((off + 8192) & 32767));
            if (KIND == 3) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0, x1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1, x0, acc1, 0, 0, 0);
            } else {
                a += x0;
                b += x1;
            }
        }
        s += (float)a[0] + (float)b[1];
    } else if (KIND == 0) {
        i8v a, b;
        for (int e = 0; e < 8; ++e) { a[e] = 0x38383838 + threadIdx.x; b[e] = 0x30303030 + e; }
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc0, 0, 0, 0, 127, 0, 116);
            acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, acc1, 0, 0, 0, 127, 0, 116);
        }
    } else if (KIND == 1) {
        h8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * threadIdx.x); b[e] = (_Float16)(0.5f + e); }
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
        }
    } else {
        for (int i = 0; i < iters * 16; ++i) s = __builtin_fmaf(s, 1.0001f, 0.5f);
    }
    float r = s;
    for (int e = 0; e < 16; ++e) r += acc0[e] + acc1[e];
    if (r == 12345.678f) out[0] = r;     // keep the loop alive
}

template <int V>
static void run_variant(const char *label, const float *din, unsigned *dbad, unsigned long long *dlanes, float *dout, hipStream_t sv, hipStream_t sc,
                        int co_kind) {
    CHECK(hipMemsetAsync(dbad, 0, 4, sv));
    CHECK(hipMemsetAsync(dlanes, 0, 8, sv));
    CHECK(hipDeviceSynchronize());
    const int rounds = 30;
    for (int r = 0; r < rounds; ++r) {
        if (co_kind == 0) hipLaunchKernelGGL(corunner<0>, dim3(512), dim3(256), 0, sc, dout, 20000);
        if (co_kind == 1) hipLaunchKernelGGL(corunner<1>, dim3(512), dim3(256), 0, sc, dout, 40000);
        if (co_kind == 2) hipLaunchKernelGGL(corunner<2>, dim3(512), dim3(256), 0, sc, dout, 40000);
        if (co_kind == 3) hipLaunchKernelGGL(corunner<3>, dim3(512), dim3(256), 0, sc, dout, 40000);
        if (co_kind == 4) hipLaunchKernelGGL(corunner<4>, dim3(512), dim3(256), 0, sc, dout, 40000);
        for (int j = 0; j < 8; ++j) hipLaunchKernelGGL(victim<V>, dim3(1024), dim3(256), 0, sv, din, dbad, dlanes, 400);
        CHECK(hipDeviceSynchronize());
    }
    unsigned bad = 0;
    unsigned long long lanes = 0;
    CHECK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&lanes, dlanes, 8, hipMemcpyDeviceToHost));
    const double total = (double)rounds * 8 * 1024 * 256 * 400;
    printf("  %-44s mismatching lane-iterations %10u of %.3g   lane mask %016llx\n", label, bad, total, lanes);
}

int main() {
    std::vector<float> h(1 << 16);
    srand(5);
    for (auto &v : h) v = (float)(rand() & 0xffff) / 65536.0f * 4.0f - 2.0f;
    float *din, *dout;
    unsigned *dbad;
    unsigned long long *dlanes;
    CHECK(hipMalloc(&din, h.size() * 4));
    CHECK(hipMalloc(&dout, 64));
    CHECK(hipMalloc(&dbad, 4));
    CHECK(hipMalloc(&dlanes, 8));
    CHECK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipStream_t sv, sc;
    CHECK(hipStreamCreate(&sv));
    CHECK(hipStreamCreate(&sc));
    const char *co_names[6] = {"block-scaled fp8 MFMA", "fp16 MFMA", "VALU fma", "fp16 MFMA fed by ds_read_b128", "ds_read_b128 only", "nothing"};
    for (int co = 0; co < 6; ++co) {
        printf("co-runner on the second stream: %s\n", co_names[co]);
        run_variant<0>("pk_mul; s_nop 0; pk_add op_sel swap", din, dbad, dlanes, dout, sv, sc, co);
        run_variant<4>("pk_mul; pk_add op_sel swap (no nop)", din, dbad, dlanes, dout, sv, sc, co);
        run_variant<1>("pk_mul; s_nop 1; pk_add op_sel swap", din, dbad, dlanes, dout, sv, sc, co);
        run_variant<2>("pk_mul; s_nop 3; pk_add op_sel swap", din, dbad, dlanes, dout, sv, sc, co);
        run_variant<3>("pk_mul; s_nop 0; pk_add (no swap)", din, dbad, dlanes, dout, sv, sc, co);
        run_variant<5>("v_mul x2; pk_add op_sel swap", din, dbad, dlanes, dout, sv, sc, co);
    }
    return 0;
}
