// HBM-bound kernels of the path: backward-warp bilinear sampler, the fused stage-2
// input builder, the fused visibility blend, 2x2 mean, concat+bilinear x2, strided copy.
// All take ssm_view tensors (include/ssm_hip.h).  Lane = pixel along x, so every
// plane access of a wave is one contiguous row segment; the gathers of the warp hit
// L2/MALL (displacements are a few pixels).  Compiled with -ffp-contract=off so the
// coordinate arithmetic rounds exactly like the reference's unfused fp32 CPU ops.
#include "ssm_common.h"

namespace {

__device__ __forceinline__ float *vp(const ssm_view &v, int b, int c, int y) {
    return v.ptr + (long long)b * v.sb + (long long)c * v.sc + (long long)y * v.sh;
}

// ---- bilinear sampler -----------------------------------------------------------------
// Sampling position of output pixel (x,y) displaced by (u,v), computed the way
// layers.warp does (scripts/models/layers.py:100-119): normalise to [-1,1] with
// max(size-1,1), then grid_sample(align_corners=True) maps back ((g+1)/2*(size-1)).
struct Taps {
    int o00, o01, o10, o11;      // offsets inside a plane (row*sh + col); -1 = outside -> contributes 0
    float w00, w01, w10, w11;    // nw, ne, sw, se
};

__device__ __forceinline__ Taps make_taps(int x, int y, float u, float v, int H, int W, int sh) {
    const float wd = (float)(W - 1 > 1 ? W - 1 : 1), hd = (float)(H - 1 > 1 ? H - 1 : 1);
    const float gx = 2.0f * ((float)x + u) / wd - 1.0f;
    const float gy = 2.0f * ((float)y + v) / hd - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    const float x0 = floorf(ix), y0 = floorf(iy);
    const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    Taps t;
    t.w00 = (x1 - ix) * (y1 - iy);
    t.w01 = (ix - x0) * (y1 - iy);
    t.w10 = (x1 - ix) * (iy - y0);
    t.w11 = (ix - x0) * (iy - y0);
    const float wm = (float)(W - 1), hm = (float)(H - 1);
    const bool bx0 = x0 >= 0.f && x0 <= wm, bx1 = x1 >= 0.f && x1 <= wm;
    const bool by0 = y0 >= 0.f && y0 <= hm, by1 = y1 >= 0.f && y1 <= hm;
    const int xi0 = bx0 ? (int)x0 : 0, xi1 = bx1 ? (int)x1 : 0;
    const int yi0 = by0 ? (int)y0 : 0, yi1 = by1 ? (int)y1 : 0;
    t.o00 = (bx0 && by0) ? yi0 * sh + xi0 : -1;
    t.o01 = (bx1 && by0) ? yi0 * sh + xi1 : -1;
    t.o10 = (bx0 && by1) ? yi1 * sh + xi0 : -1;
    t.o11 = (bx1 && by1) ? yi1 * sh + xi1 : -1;
    return t;
}

__device__ __forceinline__ float sample(const float *__restrict__ plane, const Taps &t) {
    const float a = t.o00 >= 0 ? plane[t.o00] : 0.f;
    const float b = t.o01 >= 0 ? plane[t.o01] : 0.f;
    const float c = t.o10 >= 0 ? plane[t.o10] : 0.f;
    const float d = t.o11 >= 0 ? plane[t.o11] : 0.f;
    float r = a * t.w00;
    r = r + b * t.w01;
    r = r + c * t.w10;
    r = r + d * t.w11;
    return r;
}

// blocks are (64 x-lanes, 4 rows); grid (ceil(W/64), ceil(H/4), B)
#define SSM_PIXEL_INDEX()                                     \
    const int x = blockIdx.x * 64 + threadIdx.x;              \
    const int y = blockIdx.y * 4 + threadIdx.y;               \
    const int b = blockIdx.z;                                 \
    if (x >= W || y >= H) return;

__global__ __launch_bounds__(256) void warp_kernel(ssm_view img, ssm_view flow, ssm_view out, int C, int H, int W) {
    SSM_PIXEL_INDEX();
    const float u = vp(flow, b, 0, y)[x], v = vp(flow, b, 1, y)[x];
    const Taps t = make_taps(x, y, u, v, H, W, img.sh);
    for (int c = 0; c < C; ++c) vp(out, b, c, y)[x] = sample(vp(img, b, c, 0), t);
}

// FlowInterpolationModel.compute_inputs, scripts/models/flow_interpolation.py:338-372
__global__ __launch_bounds__(256) void flowinterp_inputs_kernel(ssm_view img6, ssm_view flow4, const float *__restrict__ tarr,
                                                                ssm_view out16, int H, int W) {
    SSM_PIXEL_INDEX();
    const float t = tarr[b];
    const float omt = 1.0f - t;
    const float f01u = vp(flow4, b, 0, y)[x], f01v = vp(flow4, b, 1, y)[x];
    const float f10u = vp(flow4, b, 2, y)[x], f10v = vp(flow4, b, 3, y)[x];
    const float c00 = (-omt) * t, c01 = t * t;       // :353  -(1-t)*t*F01 + t^2*F10
    const float c10 = omt * omt, c11 = t * omt;      // :356  (1-t)^2*F01 - t(1-t)*F10
    const float ft0u = c00 * f01u + c01 * f10u, ft0v = c00 * f01v + c01 * f10v;
    const float ft1u = c10 * f01u - c11 * f10u, ft1v = c10 * f01v - c11 * f10v;
    const Taps t1 = make_taps(x, y, ft1u, ft1v, H, W, img6.sh);
    const Taps t0 = make_taps(x, y, ft0u, ft0v, H, W, img6.sh);
    // channel order is ABI (:364-367): I1, g(I1,Ft1), Ft1, Ft0, g(I0,Ft0), I0
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        vp(out16, b, c, y)[x] = vp(img6, b, 3 + c, y)[x];
        vp(out16, b, 3 + c, y)[x] = sample(vp(img6, b, 3 + c, 0), t1);
        vp(out16, b, 10 + c, y)[x] = sample(vp(img6, b, c, 0), t0);
        vp(out16, b, 13 + c, y)[x] = vp(img6, b, c, y)[x];
    }
    vp(out16, b, 6, y)[x] = ft1u;
    vp(out16, b, 7, y)[x] = ft1v;
    vp(out16, b, 8, y)[x] = ft0u;
    vp(out16, b, 9, y)[x] = ft0v;
}

// extract_outputs + compute_output_image, scripts/models/flow_interpolation.py:374-429
__global__ __launch_bounds__(256) void synthesize_kernel(ssm_view img6, ssm_view in16, ssm_view out5, const float *__restrict__ tarr,
                                                         ssm_view y3, ssm_view aux, int H, int W) {
    SSM_PIXEL_INDEX();
    const float t = tarr[b];
    const float omt = 1.0f - t;
    const float v1 = 1.0f / (1.0f + expf(-vp(out5, b, 0, y)[x]));
    const float v0 = 1.0f - v1;
    const float ft1u = vp(in16, b, 6, y)[x] + vp(out5, b, 1, y)[x];
    const float ft1v = vp(in16, b, 7, y)[x] + vp(out5, b, 2, y)[x];
    const float ft0u = vp(in16, b, 8, y)[x] + vp(out5, b, 3, y)[x];
    const float ft0v = vp(in16, b, 9, y)[x] + vp(out5, b, 4, y)[x];
    const Taps t0 = make_taps(x, y, ft0u, ft0v, H, W, img6.sh);
    const Taps t1 = make_taps(x, y, ft1u, ft1v, H, W, img6.sh);
    const float den = omt * v0 + t * v1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float p0 = v0 * sample(vp(img6, b, c, 0), t0);
        const float p1 = v1 * sample(vp(img6, b, 3 + c, 0), t1);
        vp(y3, b, c, y)[x] = (omt * p0 + t * p1) / den;
    }
    if (aux.ptr) {
        vp(aux, b, 0, y)[x] = ft1u;
        vp(aux, b, 1, y)[x] = ft1v;
        vp(aux, b, 2, y)[x] = ft0u;
        vp(aux, b, 3, y)[x] = ft0v;
        vp(aux, b, 4, y)[x] = v0;
    }
}

__global__ __launch_bounds__(256) void copy_view_kernel(ssm_view src, ssm_view dst, int C, int H, int W) {
    SSM_PIXEL_INDEX();
    for (int c = 0; c < C; ++c) vp(dst, b, c, y)[x] = vp(src, b, c, y)[x];
}

// layers.avg_pool(2): scripts/models/layers.py:60-63.  H, W here are the OUTPUT dims.
__global__ __launch_bounds__(256) void avgpool2_kernel(ssm_view xin, ssm_view yout, int C, int H, int W) {
    SSM_PIXEL_INDEX();
    for (int c = 0; c < C; ++c) {
        const float *r0 = vp(xin, b, c, 2 * y), *r1 = vp(xin, b, c, 2 * y + 1);
        const float2 a = *reinterpret_cast<const float2 *>(r0 + 2 * x);
        const float2 d = *reinterpret_cast<const float2 *>(r1 + 2 * x);
        vp(yout, b, c, y)[x] = (((a.x + a.y) + d.x) + d.y) * 0.25f;
    }
}

// F.upsample(cat[a,b], size=(2h,2w), mode="bilinear"), align_corners=False:
// scripts/models/flow_computation.py:92-94,:244-245.  One thread per SOURCE pixel
// writes its 2x2 block.  Index/lambda pairs follow ATen's half-pixel rule:
// Y=2i -> rows (i-1,i) with (.25,.75) [row 0: (0,0),(1,0)]; Y=2i+1 -> rows (i,i+1) with
// (.75,.25), the upper row clamped at h-1.   H, W here are the SOURCE dims.
__global__ __launch_bounds__(256) void upsample2x_cat_kernel(ssm_view a, int Ca, ssm_view bsrc, int Cb, ssm_view yout, int H, int W) {
    SSM_PIXEL_INDEX();
    const int ym = y > 0 ? y - 1 : 0, yp = y < H - 1 ? y + 1 : y;
    const int xm = x > 0 ? x - 1 : 0, xp = x < W - 1 ? x + 1 : x;
    const float ly0a = y > 0 ? 0.25f : 1.0f, ly0b = y > 0 ? 0.75f : 0.0f;   // output row 2y   : rows (ym, y)
    const float lx0a = x > 0 ? 0.25f : 1.0f, lx0b = x > 0 ? 0.75f : 0.0f;   // output col 2x   : cols (xm, x)
    const int ry0 = y > 0 ? ym : 0, ry0b = y > 0 ? y : 0;
    const int cx0 = x > 0 ? xm : 0, cx0b = x > 0 ? x : 0;
    const int C = Ca + Cb;
    for (int c = 0; c < C; ++c) {
        const ssm_view &s = c < Ca ? a : bsrc;
        const int cc = c < Ca ? c : c - Ca;
        const float *pT = vp(s, b, cc, ry0), *pTb = vp(s, b, cc, ry0b);   // rows for Y = 2y
        const float *pM = vp(s, b, cc, y), *pB = vp(s, b, cc, yp);        // rows for Y = 2y+1
        // Y = 2y
        const float e00 = ly0a * (lx0a * pT[cx0] + lx0b * pT[cx0b]) + ly0b * (lx0a * pTb[cx0] + lx0b * pTb[cx0b]);
        const float e01 = ly0a * (0.75f * pT[x] + 0.25f * pT[xp]) + ly0b * (0.75f * pTb[x] + 0.25f * pTb[xp]);
        // Y = 2y+1
        const float e10 = 0.75f * (lx0a * pM[cx0] + lx0b * pM[cx0b]) + 0.25f * (lx0a * pB[cx0] + lx0b * pB[cx0b]);
        const float e11 = 0.75f * (0.75f * pM[x] + 0.25f * pM[xp]) + 0.25f * (0.75f * pB[x] + 0.25f * pB[xp]);
        float *o0 = vp(yout, b, c, 2 * y) + 2 * x, *o1 = vp(yout, b, c, 2 * y + 1) + 2 * x;
        *reinterpret_cast<float2 *>(o0) = make_float2(e00, e01);
        *reinterpret_cast<float2 *>(o1) = make_float2(e10, e11);
    }
}

inline dim3 pix_grid(int B, int H, int W) { return dim3((W + 63) / 64, (H + 3) / 4, B); }
inline bool even_view(const ssm_view &v) { return ((reinterpret_cast<size_t>(v.ptr) & 7) == 0) && v.sh % 2 == 0 && v.sc % 2 == 0 && v.sb % 2 == 0; }

}  // namespace

#define SSM_CHECK_DIMS(name)                                                                         \
    SSM_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535 && (H + 3) / 4 <= 65535, name ": bad sizes B=%d H=%d W=%d", B, H, W)

extern "C" int ssm_copy_view(ssm_view src, ssm_view dst, int B, int C, int H, int W, void *stream) {
    SSM_CHECK_DIMS("copy_view");
    SSM_REQUIRE(src.ptr && dst.ptr && C > 0, "copy_view: null pointer / C");
    hipLaunchKernelGGL(copy_view_kernel, pix_grid(B, H, W), dim3(64, 4), 0, (hipStream_t)stream, src, dst, C, H, W);
    return ssm::check_launch("ssm_copy_view");
}

extern "C" int ssm_avgpool2_fwd(ssm_view x, ssm_view y, int B, int C, int H, int W, void *stream) {
    SSM_CHECK_DIMS("avgpool2");
    SSM_REQUIRE(x.ptr && y.ptr && C > 0, "avgpool2: null pointer / C");
    SSM_REQUIRE(H % 2 == 0 && W % 2 == 0, "avgpool2: H and W must be even (got %dx%d)", H, W);
    SSM_REQUIRE(even_view(x), "avgpool2: input view must be 8-byte aligned with even strides");
    hipLaunchKernelGGL(avgpool2_kernel, pix_grid(B, H / 2, W / 2), dim3(64, 4), 0, (hipStream_t)stream, x, y, C, H / 2, W / 2);
    return ssm::check_launch("ssm_avgpool2_fwd");
}

extern "C" int ssm_upsample2x_cat_fwd(ssm_view a, int Ca, ssm_view b, int Cb, ssm_view y, int B, int H, int W, void *stream) {
    SSM_CHECK_DIMS("upsample2x_cat");
    SSM_REQUIRE(a.ptr && y.ptr && Ca > 0 && Cb >= 0 && (Cb == 0 || b.ptr), "upsample2x_cat: null pointer / channels");
    SSM_REQUIRE(even_view(y), "upsample2x_cat: output view must be 8-byte aligned with even strides");
    hipLaunchKernelGGL(upsample2x_cat_kernel, pix_grid(B, H, W), dim3(64, 4), 0, (hipStream_t)stream, a, Ca, Cb ? b : a, Cb, y, H, W);
    return ssm::check_launch("ssm_upsample2x_cat_fwd");
}

extern "C" int ssm_warp_bilinear_fwd(ssm_view img, ssm_view flow, ssm_view out, int B, int C, int H, int W, void *stream) {
    SSM_CHECK_DIMS("warp");
    SSM_REQUIRE(img.ptr && flow.ptr && out.ptr && C > 0, "warp: null pointer / C");
    SSM_REQUIRE((long long)H * img.sh < 0x7fffffffLL, "warp: plane too large");
    hipLaunchKernelGGL(warp_kernel, pix_grid(B, H, W), dim3(64, 4), 0, (hipStream_t)stream, img, flow, out, C, H, W);
    return ssm::check_launch("ssm_warp_bilinear_fwd");
}

extern "C" int ssm_flowinterp_inputs_fwd(ssm_view img6, ssm_view flow4, const float *t, ssm_view out16, int B, int H, int W, void *stream) {
    SSM_CHECK_DIMS("flowinterp_inputs");
    SSM_REQUIRE(img6.ptr && flow4.ptr && out16.ptr && t, "flowinterp_inputs: null pointer");
    SSM_REQUIRE((long long)H * img6.sh < 0x7fffffffLL, "flowinterp_inputs: plane too large");
    hipLaunchKernelGGL(flowinterp_inputs_kernel, pix_grid(B, H, W), dim3(64, 4), 0, (hipStream_t)stream, img6, flow4, t, out16, H, W);
    return ssm::check_launch("ssm_flowinterp_inputs_fwd");
}

extern "C" int ssm_synthesize_fwd(ssm_view img6, ssm_view in16, ssm_view out5, const float *t, ssm_view y3, ssm_view aux, int B, int H, int W, void *stream) {
    SSM_CHECK_DIMS("synthesize");
    SSM_REQUIRE(img6.ptr && in16.ptr && out5.ptr && y3.ptr && t, "synthesize: null pointer");
    SSM_REQUIRE((long long)H * img6.sh < 0x7fffffffLL, "synthesize: plane too large");
    hipLaunchKernelGGL(synthesize_kernel, pix_grid(B, H, W), dim3(64, 4), 0, (hipStream_t)stream, img6, in16, out5, t, y3, aux, H, W);
    return ssm::check_launch("ssm_synthesize_fwd");
}
