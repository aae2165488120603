// Bring-up check for the block-scaled fp8 matrix instruction used by the fp16 + 2 x fp8 product scheme:
//   D[32x32] = (sum_k A[row][k] * B[k][col]) * 2^(scale_a-127) * 2^(scale_b-127),   K = 64
// with MY operand convention: lane l (r = l & 31, h = l >> 5) passes 32 fp8 bytes = k-block h of row r (A) / column r (B),
// byte i of the lane vector = k index 32*h + i.  Also exposes the f32 -> fp8 (e4m3) conversion the epilogues will use.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/fp8_mfma_bringup.hip -o tools/libfp8bringup.so
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// a8, b8: [32][64] bytes (row-major: row/col r, k); d: [32][32] floats
__global__ void fp8_mfma_kernel(const unsigned char *a8, const unsigned char *b8, float *d, int scale_a, int scale_b) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    i32x8 a, b;
    const int *pa = reinterpret_cast<const int *>(a8 + r * 64 + h * 32);
    const int *pb = reinterpret_cast<const int *>(b8 + r * 64 + h * 32);
    for (int i = 0; i < 8; ++i) {
        a[i] = pa[i];
        b[i] = pb[i];
    }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, scale_a, 0, scale_b);
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;      // standard 32x32 C/D map
        d[row * 32 + r] = acc[i];
    }
}

__global__ void cvt_fp8_kernel(const float *x, unsigned char *y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n) {
        const int w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
        y[2 * i] = (unsigned char)(w & 0xff);
        y[2 * i + 1] = (unsigned char)((w >> 8) & 0xff);
    }
}

extern "C" int fp8_mfma_launch(const void *a8, const void *b8, float *d, int scale_a, int scale_b, void *stream) {
    hipLaunchKernelGGL(fp8_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned char *)a8, (const unsigned char *)b8, d,
                       scale_a, scale_b);
    return (int)hipGetLastError();
}
extern "C" int cvt_fp8_launch(const float *x, void *y, int n, void *stream) {
    hipLaunchKernelGGL(cvt_fp8_kernel, dim3((n / 2 + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, (unsigned char *)y, n);
    return (int)hipGetLastError();
}
