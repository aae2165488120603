// HBM-bound kernels of the path: backward-warp bilinear sampler, the fused stage-2
// input builder, the fused visibility blend, 2x2 mean, concat+bilinear x2, strided copy.
// All take ssm_view tensors (include/ssm_hip.h).  Lane = pixel along x, so every
// plane access of a wave is one contiguous row segment; the gathers of the warp hit
// L2/MALL (displacements are a few pixels).  Compiled with -ffp-contract=off so the
// coordinate arithmetic rounds exactly like the reference's unfused fp32 CPU ops.
#include "ssm_common.h"

#include <cstdlib>

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
// IMG = false: only the ten t-dependent channels 3:13 are written (hoisted stage-2 plans convolve the frame channels 0:3 / 13:16
// once per pair from the pair itself and never read them here)
template <bool IMG>
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
        if constexpr (IMG) vp(out16, b, c, y)[x] = vp(img6, b, 3 + c, y)[x];
        vp(out16, b, 3 + c, y)[x] = sample(vp(img6, b, 3 + c, 0), t1);
        vp(out16, b, 10 + c, y)[x] = sample(vp(img6, b, c, 0), t0);
        if constexpr (IMG) vp(out16, b, 13 + c, y)[x] = vp(img6, b, c, y)[x];
    }
    vp(out16, b, 6, y)[x] = ft1u;
    vp(out16, b, 7, y)[x] = ft1v;
    vp(out16, b, 8, y)[x] = ft0u;
    vp(out16, b, 9, y)[x] = ft0v;
}

// extract_outputs + compute_output_image, scripts/models/flow_interpolation.py:374-429, for one pixel: o5 = the five
// channels of stage 2's final_conv at (b, y, x).  Shared by synthesize_kernel and final_conv_kernel<.., SYNTH> so the
// two paths round identically.
__device__ __forceinline__ void synth_pixel(const ssm_view &img6, const ssm_view &in16, const float (&o5)[5], float t, const ssm_view &y3,
                                            const ssm_view &aux, int b, int y, int x, int H, int W) {
    const float omt = 1.0f - t;
    const float v1 = 1.0f / (1.0f + expf(-o5[0]));
    const float v0 = 1.0f - v1;
    const float ft1u = vp(in16, b, 6, y)[x] + o5[1];
    const float ft1v = vp(in16, b, 7, y)[x] + o5[2];
    const float ft0u = vp(in16, b, 8, y)[x] + o5[3];
    const float ft0v = vp(in16, b, 9, y)[x] + o5[4];
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

__global__ __launch_bounds__(256) void synthesize_kernel(ssm_view img6, ssm_view in16, ssm_view out5, const float *__restrict__ tarr,
                                                         ssm_view y3, ssm_view aux, int H, int W) {
    SSM_PIXEL_INDEX();
    float o5[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) o5[c] = vp(out5, b, c, y)[x];
    synth_pixel(img6, in16, o5, tarr[b], y3, aux, b, y, x, H, W);
}

// ---- final_conv (32 -> 4 or 5 channels, no activation) [+ synthesis] ------------------------------------------------
// flow_computation.py:145-153 / flow_interpolation.py:149-157 (+ :374-429).  A 32-cout MFMA tile wastes 84 % of the
// matrix work on these layers (20 TFLOP/s, 1.1 ms of a 46 ms pair); v_mfma_f32_4x4x1_16B_f32 fits them: 16 blocks of
// (4 couts x 1 k) x (1 k x 4 pixels) - with the filter replicated over the blocks one instruction is 4 couts x 64
// pixels x 1 k, the pixel on the lane, full fp32-MFMA rate, nothing padded for 4 couts (8 for the 5 of stage 2).
// Exact fp32 (an fmaf chain per output like the 32x32x2 form).  Stage 2 never writes its 5-channel map: the lane that
// holds a pixel's five sums runs the synthesis arithmetic on them directly.
// Tile 8 rows x 64 columns, 4 waves x 2 rows; per chunk of 4 input channels the [4][10][72] patch arrives by LDS-DMA
// (double-buffered), each wave keeps the chunk's 36 x NG4 filter values in registers for both of its rows.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SSM_GLDS16(gp, lp)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp),      \
                                     (__attribute__((address_space(3))) void *)(lp), 16, 0, 0)

struct FinalParams {
    ssm_view x;               // [B,32,H,W] padded planes
    const float *w, *bias;    // OIHW [NC][32][3][3], [NC]
    int NC;
    ssm_view out;             // [B,NC,H,W] (ptr NULL with SYNTH)
    ssm_view img6, in16, y3, aux;
    const float *t;
    int H, W, tilesX, tilesY;
};

template <int NG4, bool SYNTH>
__global__ __launch_bounds__(256, 2) void final_conv_kernel(const FinalParams p) {
    constexpr int CIN = 32, CK = 4, TH = 8, TW = 64, PH = TH + 2, PW = TW + 8, PSZ = CK * PH * PW;
    constexpr int NPQ = PSZ / 4, NG = (NPQ + 63) / 64, STAGE = NG * 256, NI = (NG + 3) / 4;
    constexpr int WFL = CIN * 9 * NG4 * 4;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE + WFL];
    float *wl = lds + 2 * STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int id = ssm_xcd_tile(blockIdx.x, gridDim.x);
    const int tx = id % p.tilesX;
    id /= p.tilesX;
    const int ty = id % p.tilesY;
    const int b = id / p.tilesY;
    const int x0 = tx * TW, y0 = ty * TH;

    // filter -> LDS as [cin][tap][group][4 couts], zero beyond NC
    for (int i = tid; i < WFL; i += 256) {
        const int co = i % (NG4 * 4), ct = i / (NG4 * 4);       // ct = cin*9 + tap
        wl[i] = co < p.NC ? p.w[(long long)co * (CIN * 9) + ct] : 0.f;
    }
    const float *pbase = p.x.ptr + (long long)b * p.x.sb + (long long)(y0 - 1) * p.x.sh + (x0 - 4);
    int off[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = (i * 4 + wid) * 64 + lane;
        if (q < NPQ) {
            const int c = q / (PH * (PW / 4)), rem = q - c * (PH * (PW / 4));
            const int r = rem / (PW / 4), j = rem - r * (PW / 4);
            off[i] = (int)(c * p.x.sc) + r * p.x.sh + 4 * j;
        } else {
            off[i] = 0;       // tail of the last 1-KiB piece: lands in the stage's padding
        }
    }
    auto issue = [&](int ch, int stage) {
        const float *pb = pbase + (long long)(ch * CK) * p.x.sc;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int g = i * 4 + wid;
            if (g < NG) SSM_GLDS16(pb + off[i], lds + stage * STAGE + g * 256);
        }
    };
    f32x4 acc[2][NG4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int g = 0; g < NG4; ++g) acc[r][g] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue(0, 0);
    for (int ch = 0; ch < CIN / CK; ++ch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                      // chunk ch (and, first time, the filter) is in LDS; chunk ch-1 is consumed
        if (ch + 1 < CIN / CK) issue(ch + 1, (ch + 1) & 1);
        // macro-step = one filter row of one input channel (3 taps x 2 rows x NG4 MFMAs); its operands are fetched into the
        // other register set right after the first MFMA of the step before (the compiler, left alone, re-reads the filter
        // value from LDS immediately in front of every group of MFMAs and waits for it)
        const float *sw = wl + (ch * CK * 9 * NG4) * 4 + (lane & 3);
        const float *sb = lds + (ch & 1) * STAGE + (2 * wid) * PW + lane + 3;
        float wv[2][3][NG4], xv[2][2][3];
        auto fetch = [&](int m, int buf) {
            const int c = m / 3, ky = m % 3;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int g = 0; g < NG4; ++g) wv[buf][kx][g] = sw[((c * 9 + ky * 3 + kx) * NG4 + g) * 4];
#pragma unroll
                for (int r = 0; r < 2; ++r) xv[buf][r][kx] = sb[(c * PH + r + ky) * PW + kx];
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int m = 0; m < CK * 3; ++m) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int g = 0; g < NG4; ++g) {
                        acc[r][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[m & 1][kx][g], xv[m & 1][r][kx], acc[r][g], 0, 0, 0);
                        if (kx == 0 && r == 0 && g == 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (m + 1 < CK * 3) fetch(m + 1, (m + 1) & 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const int x = x0 + lane;
    if (x >= p.W) return;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int y = y0 + 2 * wid + r;
        if (y >= p.H) continue;
        float o[NG4 * 4];
#pragma unroll
        for (int g = 0; g < NG4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) o[g * 4 + i] = acc[r][g][i] + (g * 4 + i < p.NC ? p.bias[g * 4 + i] : 0.f);
        if constexpr (SYNTH) {
            const float o5[5] = {o[0], o[1], o[2], o[3], o[4]};
            synth_pixel(p.img6, p.in16, o5, p.t[b], p.y3, p.aux, b, y, x, p.H, p.W);
        }
        if (p.out.ptr) {
#pragma unroll
            for (int c = 0; c < NG4 * 4; ++c)
                if (c < p.NC) vp(p.out, b, c, y)[x] = o[c];
        }
    }
}

// ---- final_conv on the vector ALUs (r5) ----------------------------------------------------------------------------------------------
// 32 -> NC <= 5 channels is too narrow for any MFMA shape to pay: the 4x4x1 form above issues one LDS read per MFMA (12 reads per 12
// MFMAs of 8 cycles) and runs at 0.26 of the pipe, a third of it on padding couts.  Here a lane owns two vertically adjacent pixels and
// every product is one v_fma_f32 with the filter value broadcast from LDS: per input channel 12 + 18 LDS reads feed 18 NC FMAs, the
// kernel is bound by vector issue (4 cycles per instruction, 6 waves per SIMD).  Same tile (8 x 64 pixels, 4 waves x 2 rows), the same
// LDS-DMA patch stages and the same summation order (cin, ky, kx: an fmaf chain per output) as final_conv_kernel: bit-identical sums.
template <int NC, bool SYNTH, int RPT>
__global__ __launch_bounds__(256, 4) void final_conv_valu_kernel(const FinalParams p) {
    // RPT rows per lane (4 waves x RPT rows = the tile's height): the filter values read for a tap serve RPT pixels
    constexpr int CIN = 32, CK = 4, TH = 4 * RPT, TW = 64, PH = TH + 2, PW = TW + 8, PSZ = CK * PH * PW;
    constexpr int NPQ = PSZ / 4, NG = (NPQ + 63) / 64, STAGE = NG * 256, NI = (NG + 3) / 4;
    constexpr int WFL = CIN * 9 * 8;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE + WFL];
    float *wl = lds + 2 * STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int id = ssm_xcd_tile(blockIdx.x, gridDim.x);
    const int tx = id % p.tilesX;
    id /= p.tilesX;
    const int ty = id % p.tilesY;
    const int b = id / p.tilesY;
    const int x0 = tx * TW, y0 = ty * TH;

    // filter -> LDS as [cin][tap][8 couts], zero beyond NC
    for (int i = tid; i < WFL; i += 256) {
        const int co = i & 7, ct = i >> 3;       // ct = cin*9 + tap
        wl[i] = co < NC ? p.w[(long long)co * (CIN * 9) + ct] : 0.f;
    }
    const float *pbase = p.x.ptr + (long long)b * p.x.sb + (long long)(y0 - 1) * p.x.sh + (x0 - 4);
    int off[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = (i * 4 + wid) * 64 + lane;
        if (q < NPQ) {
            const int c = q / (PH * (PW / 4)), rem = q - c * (PH * (PW / 4));
            const int r = rem / (PW / 4), j = rem - r * (PW / 4);
            off[i] = (int)(c * p.x.sc) + r * p.x.sh + 4 * j;
        } else {
            off[i] = 0;       // tail of the last 1-KiB piece: lands in the stage's padding
        }
    }
    auto issue = [&](int ch, int stage) {
        const float *pb = pbase + (long long)(ch * CK) * p.x.sc;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int g = i * 4 + wid;
            if (g < NG) SSM_GLDS16(pb + off[i], lds + stage * STAGE + g * 256);
        }
    };
    float acc[RPT][NC];
#pragma unroll
    for (int r = 0; r < RPT; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[r][c] = 0.f;

    issue(0, 0);
    for (int ch = 0; ch < CIN / CK; ++ch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                      // chunk ch (and, first time, the filter) is in LDS; chunk ch-1 is consumed
        if (ch + 1 < CIN / CK) issue(ch + 1, (ch + 1) & 1);
        const float *sb = lds + (ch & 1) * STAGE + (RPT * wid) * PW + lane + 3;
#pragma unroll 1
        for (int c = 0; c < CK; ++c) {          // (one channel at a time: unrolled, the compiler hoists all 120 LDS reads of a chunk and spills)
            float xr[RPT + 2][3];          // rows RPT wid - 1 .. RPT wid + RPT of this channel at columns x - 1 .. x + 1
#pragma unroll
            for (int r = 0; r < RPT + 2; ++r)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) xr[r][kx] = sb[(c * PH + r) * PW + kx];
            const float *wc = wl + ((ch * CK + c) * 9) * 8;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x4 w4 = *(const f32x4 *)(wc + (ky * 3 + kx) * 8);          // (one address for the whole wave: a broadcast read)
                    const float w5 = NC > 4 ? wc[(ky * 3 + kx) * 8 + 4] : 0.f;
#pragma unroll
                    for (int r = 0; r < RPT; ++r) {
                        const float xv = xr[r + ky][kx];
#pragma unroll
                        for (int co = 0; co < NC; ++co) acc[r][co] = __builtin_fmaf(co < 4 ? w4[co] : w5, xv, acc[r][co]);
                    }
                }
        }
    }
    const int x = x0 + lane;
    if (x >= p.W) return;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int y = y0 + RPT * wid + r;
        if (y >= p.H) continue;
        float o[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) o[c] = acc[r][c] + p.bias[c];
        if constexpr (SYNTH) {
            const float o5[5] = {o[0], o[1], o[2], o[3], o[NC > 4 ? 4 : 0]};
            synth_pixel(p.img6, p.in16, o5, p.t[b], p.y3, p.aux, b, y, x, p.H, p.W);
        }
        if (p.out.ptr) {
#pragma unroll
            for (int c = 0; c < NC; ++c) vp(p.out, b, c, y)[x] = o[c];
        }
    }
}

__global__ __launch_bounds__(256) void copy_view_kernel(ssm_view src, ssm_view dst, int C, int H, int W) {
    SSM_PIXEL_INDEX();
    for (int c = 0; c < C; ++c) vp(dst, b, c, y)[x] = vp(src, b, c, y)[x];
}

// layers.avg_pool(2): scripts/models/layers.py:60-63.  H, W here are the OUTPUT dims.
// channels are spread over blockIdx.z in groups of 4 so small maps with many channels still fill the chip
__global__ __launch_bounds__(256) void avgpool2_kernel(ssm_view xin, ssm_view yout, int C, int H, int W, int cgroups) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z / cgroups, c0 = (blockIdx.z - b * cgroups) * 4;
    if (x >= W || y >= H) return;
    for (int c = c0; c < c0 + 4 && c < C; ++c) {
        const float *r0 = vp(xin, b, c, 2 * y), *r1 = vp(xin, b, c, 2 * y + 1);
        const float2 a = *reinterpret_cast<const float2 *>(r0 + 2 * x);
        const float2 d = *reinterpret_cast<const float2 *>(r1 + 2 * x);
        vp(yout, b, c, y)[x] = (((a.x + a.y) + d.x) + d.y) * 0.25f;
    }
}

// Second half of a split-K convolution (csrc/ssm_wino.hip, ssm_wino_conv2d_splitk_fwd): y = act(sum_ks part[ks * B + b] + addend), the
// partial sums added in the order ks = 0, 1, ... (deterministic), + the fused 2x2 mean (vertical pairs first, like the convolution
// kernels' epilogues).  One thread = a 2x2 block of one channel (W even, H any), the blocks of the whole batch in one flat index: the
// maps this runs on are 22-46 pixels wide - a grid shaped after the map would leave most lanes of a wave without a block.
__global__ __launch_bounds__(256) void splitk_finish_kernel(ssm_view part, int KS, ssm_view yout, ssm_view pool, ssm_view add, int adiv, int B,
                                                            int C, int H, int W, float sl, float msl, long long total) {          // msl > 0: the addend is a mask source (SSM_FLAG_MASK), slope msl
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int bw = W / 2, bh = (H + 1) / 2;
    const int x = (int)(idx % bw);
    long long r = idx / bw;
    const int y = (int)(r % bh);
    r /= bh;
    const int c = (int)(r % C), b = (int)(r / C);
    const bool two = 2 * y + 1 < H;
    float2 r0 = make_float2(0.f, 0.f), r1 = make_float2(0.f, 0.f);
    for (int k = 0; k < KS; ++k) {
        const float2 a0 = *reinterpret_cast<const float2 *>(vp(part, k * B + b, c, 2 * y) + 2 * x);
        r0.x += a0.x;
        r0.y += a0.y;
        if (two) {
            const float2 a1 = *reinterpret_cast<const float2 *>(vp(part, k * B + b, c, 2 * y + 1) + 2 * x);
            r1.x += a1.x;
            r1.y += a1.y;
        }
    }
    if (add.ptr) {
        const float2 z0 = *reinterpret_cast<const float2 *>(vp(add, b / adiv, c, 2 * y) + 2 * x);
        if (msl > 0.f) {
            r0.x *= z0.x > 0.f ? 1.f : msl;
            r0.y *= z0.y > 0.f ? 1.f : msl;
        } else {
            r0.x += z0.x;
            r0.y += z0.y;
        }
        if (two) {
            const float2 z1 = *reinterpret_cast<const float2 *>(vp(add, b / adiv, c, 2 * y + 1) + 2 * x);
            if (msl > 0.f) {
                r1.x *= z1.x > 0.f ? 1.f : msl;
                r1.y *= z1.y > 0.f ? 1.f : msl;
            } else {
                r1.x += z1.x;
                r1.y += z1.y;
            }
        }
    }
    r0.x = fmaxf(r0.x, r0.x * sl);
    r0.y = fmaxf(r0.y, r0.y * sl);
    r1.x = fmaxf(r1.x, r1.x * sl);
    r1.y = fmaxf(r1.y, r1.y * sl);
    *reinterpret_cast<float2 *>(vp(yout, b, c, 2 * y) + 2 * x) = r0;
    if (two) *reinterpret_cast<float2 *>(vp(yout, b, c, 2 * y + 1) + 2 * x) = r1;
    if (pool.ptr && two) vp(pool, b, c, y)[x] = ((r0.x + r1.x) + (r0.y + r1.y)) * 0.25f;
}

// the same for odd map widths (the 11x11 bottleneck maps of config 3: direct-form split-K, no fused mean): one thread = one pixel
__global__ __launch_bounds__(256) void splitk_finish1_kernel(ssm_view part, int KS, ssm_view yout, ssm_view add, int adiv, int B, int C, int H,
                                                             int W, float sl, float msl, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % W);
    long long r = idx / W;
    const int y = (int)(r % H);
    r /= H;
    const int c = (int)(r % C), b = (int)(r / C);
    float v = 0.f;
    for (int k = 0; k < KS; ++k) v += vp(part, k * B + b, c, y)[x];
    if (add.ptr) {
        const float z = vp(add, b / adiv, c, y)[x];
        if (msl > 0.f) v *= z > 0.f ? 1.f : msl;
        else v += z;
    }
    vp(yout, b, c, y)[x] = fmaxf(v, v * sl);
}

// F.upsample(cat[a,b], size=(2h,2w), mode="bilinear"), align_corners=False:
// scripts/models/flow_computation.py:92-94,:244-245.  One thread owns TWO adjacent source
// pixels of one row and writes their 2x4 output block as two 16-byte stores per channel
// (lanes along x: 1 KiB contiguous per wave-instruction); channels are spread over
// blockIdx.z so small maps with many channels still fill the chip.  Index/lambda pairs
// follow ATen's half-pixel rule: Y=2i -> rows (i-1,i) with (.25,.75) [Y=0: row 0 alone];
// Y=2i+1 -> rows (i,i+1) with (.75,.25), upper row clamped at h-1; same along x.
// H, W = SOURCE dims.  UP_CPT channels per thread.
#define UP_CPT 4
template <bool VEC4>
__global__ __launch_bounds__(256) void upsample2x_cat_kernel(ssm_view a, int Ca, ssm_view bsrc, int Cb, ssm_view yout, int H, int W, int cgroups) {
    const int xp = blockIdx.x * 32 + threadIdx.x;      // source pixel pair index
    const int y = blockIdx.y * 8 + threadIdx.y;
    const int b = blockIdx.z / cgroups;
    const int cg = blockIdx.z - b * cgroups;
    const int x = 2 * xp;
    if (x >= W || y >= H) return;
    const int ym = y > 0 ? y - 1 : 0, yp = y < H - 1 ? y + 1 : y;
    const bool pair = x + 1 < W;   // odd W: the last thread of a row owns a single source pixel
    const int xm = x > 0 ? x - 1 : 0, x1 = pair ? x + 1 : x, x2 = x + 2 < W ? x + 2 : W - 1;
    const float wyT0 = y > 0 ? 0.25f : 1.0f, wyT1 = y > 0 ? 0.75f : 0.0f;   // Y = 2y   : rows (ym, y)
    const float wx00 = x > 0 ? 0.25f : 1.0f, wx01 = x > 0 ? 0.75f : 0.0f;   // X = 2x   : cols (xm, x)
    const int C = Ca + Cb;
    const int c0 = cg * UP_CPT;
#pragma unroll
    for (int i = 0; i < UP_CPT; ++i) {
        const int c = c0 + i;
        if (c >= C) break;
        const ssm_view &s = c < Ca ? a : bsrc;
        const int cc = c < Ca ? c : c - Ca;
        const float *r0 = vp(s, b, cc, ym), *r1 = vp(s, b, cc, y), *r2 = vp(s, b, cc, yp);
        float v[3][4];
        v[0][0] = r0[xm]; v[0][1] = r0[x]; v[0][2] = r0[x1]; v[0][3] = r0[x2];
        v[1][0] = r1[xm]; v[1][1] = r1[x]; v[1][2] = r1[x1]; v[1][3] = r1[x2];
        v[2][0] = r2[xm]; v[2][1] = r2[x]; v[2][2] = r2[x1]; v[2][3] = r2[x2];
        // horizontal pass for the three source rows: X = 2x, 2x+1, 2x+2, 2x+3
        float h[3][4];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            h[r][0] = wx00 * v[r][0] + wx01 * v[r][1];
            h[r][1] = 0.75f * v[r][1] + 0.25f * v[r][2];
            h[r][2] = 0.25f * v[r][1] + 0.75f * v[r][2];
            h[r][3] = 0.75f * v[r][2] + 0.25f * v[r][3];
        }
        float4 o0, o1;
        o0.x = wyT0 * h[0][0] + wyT1 * h[1][0]; o0.y = wyT0 * h[0][1] + wyT1 * h[1][1];
        o0.z = wyT0 * h[0][2] + wyT1 * h[1][2]; o0.w = wyT0 * h[0][3] + wyT1 * h[1][3];
        o1.x = 0.75f * h[1][0] + 0.25f * h[2][0]; o1.y = 0.75f * h[1][1] + 0.25f * h[2][1];
        o1.z = 0.75f * h[1][2] + 0.25f * h[2][2]; o1.w = 0.75f * h[1][3] + 0.25f * h[2][3];
        float *d0 = vp(yout, b, c, 2 * y) + 2 * x, *d1 = vp(yout, b, c, 2 * y + 1) + 2 * x;
        if (VEC4 && pair) {
            *reinterpret_cast<float4 *>(d0) = o0;
            *reinterpret_cast<float4 *>(d1) = o1;
        } else if (VEC4) {
            *reinterpret_cast<float2 *>(d0) = make_float2(o0.x, o0.y);
            *reinterpret_cast<float2 *>(d1) = make_float2(o1.x, o1.y);
        } else {   // unaligned destination (plain NCHW with a width that is not a multiple of 4)
            d0[0] = o0.x; d0[1] = o0.y; d1[0] = o1.x; d1[1] = o1.y;
            if (pair) { d0[2] = o0.z; d0[3] = o0.w; d1[2] = o1.z; d1[3] = o1.w; }
        }
    }
}

// ---- HL8 (fp16 hi/lo, 8-channel groups) variants ------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ const char *hp(const ssm_hview &v, int b, int g, int y, int x) {
    return (const char *)v.ptr + ((long long)b * v.sb + (long long)g * v.sg + (long long)y * v.sh + x) * 16;
}

__device__ __forceinline__ void hl8_load(const ssm_hview &v, int b, int g, int y, int x, float (&o)[8]) {
    const char *s = hp(v, b, g, y, x);
    const h8 hi = *reinterpret_cast<const h8 *>(s), lo = *reinterpret_cast<const h8 *>(s + v.sp * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)hi[e] + (float)lo[e];
}

__device__ __forceinline__ void hl8_store(const ssm_hview &v, int b, int g, int y, int x, const float (&o)[8]) {
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        hi[e] = (_Float16)o[e];
        lo[e] = (_Float16)(o[e] - (float)hi[e]);
    }
    char *d = const_cast<char *>(hp(v, b, g, y, x));
    *reinterpret_cast<h8 *>(d) = hi;
    *reinterpret_cast<h8 *>(d + v.sp * 16) = lo;
}

// concat + bilinear x2 on HL8 tensors: one thread = one SOURCE pixel of one 8-channel group -> its 2x2
// output block (each output pixel is one 16-byte hi store + one 16-byte lo store).  Same half-pixel
// index/lambda rule as upsample2x_cat_kernel above.  H, W = source dims; Ga/Gb channel groups.
__global__ __launch_bounds__(256) void upsample2x_cat_hl8_kernel(ssm_hview a, int Ga, ssm_hview bsrc, int Gb, ssm_hview yout, int H, int W) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int G = Ga + Gb;
    const int b = blockIdx.z / G, g = blockIdx.z - b * G;
    if (x >= W || y >= H) return;
    const ssm_hview &s = g < Ga ? a : bsrc;
    const int gg = g < Ga ? g : g - Ga;
    const int ym = y > 0 ? y - 1 : 0, yp = y < H - 1 ? y + 1 : y;
    const int xm = x > 0 ? x - 1 : 0, xp = x < W - 1 ? x + 1 : x;
    const float wy0 = y > 0 ? 0.25f : 1.0f, wy1 = y > 0 ? 0.75f : 0.0f;     // Y = 2y   : rows (ym, y)
    const float wx0 = x > 0 ? 0.25f : 1.0f, wx1 = x > 0 ? 0.75f : 0.0f;     // X = 2x   : cols (xm, x)
    float v[3][3][8];
    const int ys[3] = {ym, y, yp}, xs[3] = {xm, x, xp};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) hl8_load(s, b, gg, ys[r], xs[c], v[r][c]);
    float o00[8], o01[8], o10[8], o11[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float h0[3], h1[3];      // horizontal pass: X = 2x and X = 2x+1, for the three rows
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            h0[r] = wx0 * v[r][0][e] + wx1 * v[r][1][e];
            h1[r] = 0.75f * v[r][1][e] + 0.25f * v[r][2][e];
        }
        o00[e] = wy0 * h0[0] + wy1 * h0[1];
        o01[e] = wy0 * h1[0] + wy1 * h1[1];
        o10[e] = 0.75f * h0[1] + 0.25f * h0[2];
        o11[e] = 0.75f * h1[1] + 0.25f * h1[2];
    }
    hl8_store(yout, b, g, 2 * y, 2 * x, o00);
    hl8_store(yout, b, g, 2 * y, 2 * x + 1, o01);
    hl8_store(yout, b, g, 2 * y + 1, 2 * x, o10);
    hl8_store(yout, b, g, 2 * y + 1, 2 * x + 1, o11);
}

// Q8 form of a PAIR of channel groups (include/ssm_hip.h): hi planes as fp16; the even group's second plane holds the fp8 values of
// both groups, the odd group's the fp8 (lo * 2^11) of both.
__device__ __forceinline__ int pack4_fp8(float a, float b, float c, float d) {
    const float lim = 448.0f;                     // e4m3fn: beyond 448 -> NaN
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a, -lim, lim), __builtin_amdgcn_fmed3f(b, -lim, lim), 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(c, -lim, lim), __builtin_amdgcn_fmed3f(d, -lim, lim), w, true);
}

__device__ __forceinline__ void hq8_store_pair(const ssm_hview &v, int b, int g_even, int y, int x, const float (&a)[8], const float (&c)[8]) {
    h8 ha, hc;
    float la[8], lc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ha[e] = (_Float16)a[e];
        hc[e] = (_Float16)c[e];
        la[e] = (a[e] - (float)ha[e]) * 2048.0f;
        lc[e] = (c[e] - (float)hc[e]) * 2048.0f;
    }
    char *d0 = const_cast<char *>(hp(v, b, g_even, y, x));
    char *d1 = d0 + v.sg * 16;
    *reinterpret_cast<h8 *>(d0) = ha;
    *reinterpret_cast<h8 *>(d1) = hc;
    typedef int i4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<i4 *>(d0 + v.sp * 16) = i4{pack4_fp8(a[0], a[1], a[2], a[3]), pack4_fp8(a[4], a[5], a[6], a[7]),
                                                  pack4_fp8(c[0], c[1], c[2], c[3]), pack4_fp8(c[4], c[5], c[6], c[7])};
    *reinterpret_cast<i4 *>(d1 + v.sp * 16) = i4{pack4_fp8(la[0], la[1], la[2], la[3]), pack4_fp8(la[4], la[5], la[6], la[7]),
                                                  pack4_fp8(lc[0], lc[1], lc[2], lc[3]), pack4_fp8(lc[4], lc[5], lc[6], lc[7])};
}

// compute_inputs writing the 16-channel tensor straight into HL8 (two groups; Q8: the fp8 second planes) plus the four
// approximated flow channels (Ft1^ u,v | Ft0^ u,v) as fp32 planes for the synthesis kernel.
template <bool Q8>
__global__ __launch_bounds__(256) void flowinterp_inputs_hl8_kernel(ssm_view img6, ssm_view flow4, const float *__restrict__ tarr,
                                                                    ssm_hview out16, ssm_view flows, int H, int W) {
    SSM_PIXEL_INDEX();
    const float t = tarr[b];
    const float omt = 1.0f - t;
    const float f01u = vp(flow4, b, 0, y)[x], f01v = vp(flow4, b, 1, y)[x];
    const float f10u = vp(flow4, b, 2, y)[x], f10v = vp(flow4, b, 3, y)[x];
    const float c00 = (-omt) * t, c01 = t * t;
    const float c10 = omt * omt, c11 = t * omt;
    const float ft0u = c00 * f01u + c01 * f10u, ft0v = c00 * f01v + c01 * f10v;
    const float ft1u = c10 * f01u - c11 * f10u, ft1v = c10 * f01v - c11 * f10v;
    const Taps t1 = make_taps(x, y, ft1u, ft1v, H, W, img6.sh);
    const Taps t0 = make_taps(x, y, ft0u, ft0v, H, W, img6.sh);
    float ga[8], gb[8];     // channels 0..7 and 8..15 (order: I1, g(I1), Ft1, Ft0, g(I0), I0)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ga[c] = vp(img6, b, 3 + c, y)[x];
        ga[3 + c] = sample(vp(img6, b, 3 + c, 0), t1);
        gb[2 + c] = sample(vp(img6, b, c, 0), t0);
        gb[5 + c] = vp(img6, b, c, y)[x];
    }
    ga[6] = ft1u;
    ga[7] = ft1v;
    gb[0] = ft0u;
    gb[1] = ft0v;
    if (Q8) {
        hq8_store_pair(out16, b, 0, y, x, ga, gb);
    } else {
        hl8_store(out16, b, 0, y, x, ga);
        hl8_store(out16, b, 1, y, x, gb);
    }
    vp(flows, b, 0, y)[x] = ft1u;
    vp(flows, b, 1, y)[x] = ft1v;
    vp(flows, b, 2, y)[x] = ft0u;
    vp(flows, b, 3, y)[x] = ft0v;
}

// ---- frame formats either side of the path ---------------------------------------------------------
// uint8 HWC RGB frames -> normalised, zero-padded fp32 NCHW: fuses the evaluator's ToTensor + Normalize +
// EvalPad (scripts/utils/dataloaders/augmentations.py:141-200: pad value 0 AFTER normalisation) or the
// visualiser's load_batch + normalize_tensor (scripts/visualize_interpolation.py:61-88,257-262: zero pixels
// padded BEFORE normalisation -> pad value (0/255-mean)/std).  out [N,3,Hp,Wp]; (top,left) = pad offsets.
__global__ __launch_bounds__(256) void frames_from_u8_kernel(const unsigned char *__restrict__ in, ssm_view out, int H, int W,
                                                             int Hp, int Wp, int top, int left, float m0, float m1, float m2,
                                                             float s0, float s1, float s2, int pad_before_norm) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (x >= Wp || y >= Hp) return;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    const int sy = y - top, sx = x - left;
    const bool inside = sy >= 0 && sy < H && sx >= 0 && sx < W;
    const unsigned char *px = in + (((long long)b * H + (inside ? sy : 0)) * W + (inside ? sx : 0)) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v;
        if (inside) v = ((float)px[c] / 255.0f - mean[c]) / sd[c];
        else v = pad_before_norm ? (0.0f / 255.0f - mean[c]) / sd[c] : 0.0f;
        vp(out, b, c, y)[x] = v;
    }
}

// normalised fp32 NCHW -> cropped uint8 HWC: get_crop + denormalize + *255 + astype(uint8)
// (scripts/evaluate_interpolation_results.py:143-163,192-202; scripts/visualize_interpolation.py:223-237,264-268).
// mode 0 = numpy's float->uint8 cast as the reference performs it (truncate toward zero, wrap modulo 256);
// mode 1 = round to nearest and saturate (what a video writer wants).
__global__ __launch_bounds__(256) void frames_to_u8_kernel(ssm_view in, unsigned char *__restrict__ out, int H, int W, int top,
                                                           int left, float m0, float m1, float m2, float s0, float s1, float s2,
                                                           int mode) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (x >= W || y >= H) return;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    unsigned char *px = out + (((long long)b * H + y) * W + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = vp(in, b, c, y + top)[x + left] * sd[c] + mean[c];
        v = v * 255.0f;
        int q;
        if (mode == 0) {
            const float t = truncf(v);
            q = (t >= -2147483648.0f && t < 2147483648.0f) ? ((int)t & 255) : 0;
        } else {
            const float r = rintf(v);
            q = r < 0.f ? 0 : (r > 255.f ? 255 : (int)r);
        }
        px[c] = (unsigned char)q;
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
    SSM_LAUNCH(copy_view_kernel, pix_grid(B, H, W), dim3(64, 4), 0, (hipStream_t)stream, src, dst, C, H, W);
    return ssm::check_launch("ssm_copy_view");
}

extern "C" int ssm_avgpool2_fwd(ssm_view x, ssm_view y, int B, int C, int H, int W, void *stream) {
    SSM_CHECK_DIMS("avgpool2");
    SSM_REQUIRE(x.ptr && y.ptr && C > 0, "avgpool2: null pointer / C");
    SSM_REQUIRE(H % 2 == 0 && W % 2 == 0, "avgpool2: H and W must be even (got %dx%d)", H, W);
    SSM_REQUIRE(even_view(x), "avgpool2: input view must be 8-byte aligned with even strides");
    const int cgroups = (C + 3) / 4;
    SSM_REQUIRE((long long)B * cgroups <= 65535, "avgpool2: B*C too large for one launch");
    SSM_LAUNCH(avgpool2_kernel, pix_grid(B * cgroups, H / 2, W / 2), dim3(64, 4), 0, (hipStream_t)stream, x, y, C, H / 2, W / 2, cgroups);
    return ssm::check_launch("ssm_avgpool2_fwd");
}

extern "C" int ssm_splitk_finish_fwd(ssm_view part, int KS, ssm_view y, ssm_view pool, ssm_view add, int add_div, int B, int C, int H, int W,
                                     float slope, int flags, void *stream) {
    SSM_CHECK_DIMS("splitk_finish");
    SSM_REQUIRE(part.ptr && y.ptr && C > 0 && KS >= 1 && KS <= 8, "splitk_finish: null pointer / C / KS");
    if (W % 2) {          // odd width: one pixel per thread, no fused mean
        SSM_REQUIRE(!pool.ptr, "splitk_finish: the fused 2x2 mean needs even H, W");
        SSM_REQUIRE(!add.ptr || (add_div >= 1 && B % add_div == 0), "splitk_finish: addend divisor");
        const long long tot1 = (long long)B * C * H * W;
        SSM_REQUIRE(tot1 <= 0x7fffffffLL * 256LL, "splitk_finish: problem too large for one launch");
        SSM_LAUNCH(splitk_finish1_kernel, dim3((unsigned)((tot1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part, KS, y, add,
                           add_div < 1 ? 1 : add_div, B, C, H, W, (flags & SSM_FLAG_LRELU) ? slope : 1.f, (flags & SSM_FLAG_MASK) ? slope : 0.f, tot1);
        return ssm::check_launch("ssm_splitk_finish_fwd");
    }
    SSM_REQUIRE(even_view(part) && even_view(y), "splitk_finish: partial sums and output must be 8-byte aligned views with even strides");
    SSM_REQUIRE(!add.ptr || (even_view(add) && add_div >= 1 && B % add_div == 0), "splitk_finish: addend view / divisor");
    SSM_REQUIRE(!pool.ptr || (H % 2 == 0), "splitk_finish: the fused 2x2 mean needs even H, W");
    const long long total = (long long)B * C * ((H + 1) / 2) * (W / 2);
    SSM_REQUIRE(total <= 0x7fffffffLL * 256LL, "splitk_finish: problem too large for one launch");
    const float sl = (flags & SSM_FLAG_LRELU) ? slope : 1.f;
    SSM_LAUNCH(splitk_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part, KS, y, pool, add,
                       add_div < 1 ? 1 : add_div, B, C, H, W, sl, (flags & SSM_FLAG_MASK) ? slope : 0.f, total);
    return ssm::check_launch("ssm_splitk_finish_fwd");
}

extern "C" int ssm_upsample2x_cat_fwd(ssm_view a, int Ca, ssm_view b, int Cb, ssm_view y, int B, int H, int W, void *stream) {
    SSM_CHECK_DIMS("upsample2x_cat");
    SSM_REQUIRE(a.ptr && y.ptr && Ca > 0 && Cb >= 0 && (Cb == 0 || b.ptr), "upsample2x_cat: null pointer / channels");
    const bool vec4 = ssm::aligned16(y.ptr) && y.sh % 4 == 0 && y.sc % 4 == 0 && y.sb % 4 == 0;
    const int cgroups = (Ca + Cb + UP_CPT - 1) / UP_CPT;
    SSM_REQUIRE((long long)B * cgroups <= 65535, "upsample2x_cat: B*C too large for one launch");
    const dim3 grid(((W + 1) / 2 + 31) / 32, (H + 7) / 8, B * cgroups);
    if (vec4)
        SSM_LAUNCH(upsample2x_cat_kernel<true>, grid, dim3(32, 8), 0, (hipStream_t)stream, a, Ca, Cb ? b : a, Cb, y, H, W, cgroups);
    else
        SSM_LAUNCH(upsample2x_cat_kernel<false>, grid, dim3(32, 8), 0, (hipStream_t)stream, a, Ca, Cb ? b : a, Cb, y, H, W, cgroups);
    return ssm::check_launch("ssm_upsample2x_cat_fwd");
}

extern "C" int ssm_warp_bilinear_fwd(ssm_view img, ssm_view flow, ssm_view out, int B, int C, int H, int W, void *stream) {
    SSM_CHECK_DIMS("warp");
    SSM_REQUIRE(img.ptr && flow.ptr && out.ptr && C > 0, "warp: null pointer / C");
    SSM_REQUIRE((long long)H * img.sh < 0x7fffffffLL, "warp: plane too large");
    SSM_LAUNCH(warp_kernel, pix_grid(B, H, W), dim3(64, 4), 0, (hipStream_t)stream, img, flow, out, C, H, W);
    return ssm::check_launch("ssm_warp_bilinear_fwd");
}

extern "C" int ssm_flowinterp_inputs_fwd(ssm_view img6, ssm_view flow4, const float *t, ssm_view out16, int B, int H, int W, void *stream) {
    SSM_CHECK_DIMS("flowinterp_inputs");
    SSM_REQUIRE(img6.ptr && flow4.ptr && out16.ptr && t, "flowinterp_inputs: null pointer");
    SSM_REQUIRE((long long)H * img6.sh < 0x7fffffffLL, "flowinterp_inputs: plane too large");
    SSM_LAUNCH(flowinterp_inputs_kernel<true>, pix_grid(B, H, W), dim3(64, 4), 0, (hipStream_t)stream, img6, flow4, t, out16, H, W);
    return ssm::check_launch("ssm_flowinterp_inputs_fwd");
}

extern "C" int ssm_flowinterp_inputs_t_fwd(ssm_view img6, ssm_view flow4, const float *t, ssm_view out16, int B, int H, int W, void *stream) {
    SSM_CHECK_DIMS("flowinterp_inputs_t");
    SSM_REQUIRE(img6.ptr && flow4.ptr && out16.ptr && t, "flowinterp_inputs_t: null pointer");
    SSM_REQUIRE((long long)H * img6.sh < 0x7fffffffLL, "flowinterp_inputs_t: plane too large");
    SSM_LAUNCH(flowinterp_inputs_kernel<false>, pix_grid(B, H, W), dim3(64, 4), 0, (hipStream_t)stream, img6, flow4, t, out16, H, W);
    return ssm::check_launch("ssm_flowinterp_inputs_t_fwd");
}

extern "C" int ssm_synthesize_fwd(ssm_view img6, ssm_view in16, ssm_view out5, const float *t, ssm_view y3, ssm_view aux, int B, int H, int W, void *stream) {
    SSM_CHECK_DIMS("synthesize");
    SSM_REQUIRE(img6.ptr && in16.ptr && out5.ptr && y3.ptr && t, "synthesize: null pointer");
    SSM_REQUIRE((long long)H * img6.sh < 0x7fffffffLL, "synthesize: plane too large");
    SSM_LAUNCH(synthesize_kernel, pix_grid(B, H, W), dim3(64, 4), 0, (hipStream_t)stream, img6, in16, out5, t, y3, aux, H, W);
    return ssm::check_launch("ssm_synthesize_fwd");
}

extern "C" int ssm_final_conv_fwd(ssm_view x, const float *w_oihw, const float *bias, int NC, ssm_view out, ssm_view img6, ssm_view in16,
                                  const float *t, ssm_view y3, ssm_view aux, int B, int H, int W, void *stream) {
    SSM_CHECK_DIMS("final_conv");
    SSM_REQUIRE(x.ptr && w_oihw && bias && NC >= 1 && NC <= 8, "final_conv: null pointer / 1..8 output channels");
    SSM_REQUIRE(ssm::aligned16(x.ptr) && x.sh % 4 == 0 && x.sc % 4 == 0 && x.sb % 4 == 0 && x.sh >= W + 2 * SSM_PADX,
                "final_conv: input is not a padded-plane view");
    SSM_REQUIRE(4LL * x.sc < 0x7fffffffLL, "final_conv: channel stride too large");
    const bool synth = y3.ptr != nullptr;
    if (synth) {
        SSM_REQUIRE(NC == 5 && img6.ptr && in16.ptr && t, "final_conv: the synthesis epilogue needs the 5-channel stage-2 filter, img6, in16, t");
        SSM_REQUIRE((long long)H * img6.sh < 0x7fffffffLL, "final_conv: plane too large");
    } else {
        SSM_REQUIRE(out.ptr, "final_conv: no output");
    }
    FinalParams p;
    p.x = x; p.w = w_oihw; p.bias = bias; p.NC = NC; p.out = out; p.img6 = img6; p.in16 = in16; p.y3 = y3; p.aux = aux; p.t = t;
    // the vector-ALU form for the two filters of the model (4 and 5 channels; $SSM_FINAL_VALU=0: the 4x4x1-MFMA form, which also serves other NC).
    // Both forms sit on the read of the 32 input planes (profiles/r11s_final_conv_valu_ab.txt: four rows per lane instead of two changes nothing)
    const char *env = getenv("SSM_FINAL_VALU");          // read per call: the parity tests run both forms in one process
    const bool valu = !(env && atoi(env) == 0);
    const bool use_valu = valu && (NC == 4 || NC == 5);
    p.H = H; p.W = W; p.tilesX = (W + 63) / 64; p.tilesY = (H + 7) / 8;
    const long long blocks = (long long)p.tilesX * p.tilesY * B;
    SSM_REQUIRE(blocks > 0 && blocks <= 0x7fffffffLL, "final_conv: grid out of range");
    const dim3 grid((unsigned)blocks), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (use_valu) {
        if (synth) SSM_LAUNCH((final_conv_valu_kernel<5, true, 2>), grid, blk, 0, st, p);
        else if (NC == 4) SSM_LAUNCH((final_conv_valu_kernel<4, false, 2>), grid, blk, 0, st, p);
        else SSM_LAUNCH((final_conv_valu_kernel<5, false, 2>), grid, blk, 0, st, p);
    } else if (synth) SSM_LAUNCH((final_conv_kernel<2, true>), grid, blk, 0, st, p);
    else if (NC <= 4) SSM_LAUNCH((final_conv_kernel<1, false>), grid, blk, 0, st, p);
    else SSM_LAUNCH((final_conv_kernel<2, false>), grid, blk, 0, st, p);
    return ssm::check_launch("ssm_final_conv_fwd");
}

extern "C" int ssm_upsample2x_cat_hl8_fwd(ssm_hview a, int Ga, ssm_hview b, int Gb, ssm_hview y, int B, int H, int W, void *stream) {
    SSM_CHECK_DIMS("upsample2x_cat_hl8");
    SSM_REQUIRE(a.ptr && y.ptr && Ga > 0 && Gb >= 0 && (Gb == 0 || b.ptr), "upsample2x_cat_hl8: null pointer / groups");
    SSM_REQUIRE((long long)B * (Ga + Gb) <= 65535, "upsample2x_cat_hl8: B*G too large for one launch");
    SSM_LAUNCH(upsample2x_cat_hl8_kernel, dim3((W + 63) / 64, (H + 3) / 4, B * (Ga + Gb)), dim3(64, 4), 0,
                       (hipStream_t)stream, a, Ga, Gb ? b : a, Gb, y, H, W);
    return ssm::check_launch("ssm_upsample2x_cat_hl8_fwd");
}

extern "C" int ssm_flowinterp_inputs_hl8_fwd(ssm_view img6, ssm_view flow4, const float *t, ssm_hview out16, ssm_view flows, int B, int H, int W, void *stream) {
    SSM_CHECK_DIMS("flowinterp_inputs_hl8");
    SSM_REQUIRE(img6.ptr && flow4.ptr && out16.ptr && flows.ptr && t, "flowinterp_inputs_hl8: null pointer");
    SSM_REQUIRE((long long)H * img6.sh < 0x7fffffffLL, "flowinterp_inputs_hl8: plane too large");
    SSM_LAUNCH(flowinterp_inputs_hl8_kernel<false>, pix_grid(B, H, W), dim3(64, 4), 0, (hipStream_t)stream, img6, flow4, t, out16, flows, H, W);
    return ssm::check_launch("ssm_flowinterp_inputs_hl8_fwd");
}

extern "C" int ssm_flowinterp_inputs_hq8_fwd(ssm_view img6, ssm_view flow4, const float *t, ssm_hview out16, ssm_view flows, int B, int H,
                                             int W, void *stream) {
    SSM_CHECK_DIMS("flowinterp_inputs_hq8");
    SSM_REQUIRE(img6.ptr && flow4.ptr && out16.ptr && flows.ptr && t, "flowinterp_inputs_hq8: null pointer");
    SSM_REQUIRE((long long)H * img6.sh < 0x7fffffffLL, "flowinterp_inputs_hq8: plane too large");
    SSM_LAUNCH(flowinterp_inputs_hl8_kernel<true>, pix_grid(B, H, W), dim3(64, 4), 0, (hipStream_t)stream, img6, flow4, t, out16, flows, H, W);
    return ssm::check_launch("ssm_flowinterp_inputs_hq8_fwd");
}

extern "C" int ssm_frames_from_u8_fwd(const unsigned char *frames_hwc, ssm_view out, int N, int H, int W, int Hp, int Wp, int top,
                                      int left, const float *mean3, const float *std3, int pad_before_norm, void *stream) {
    SSM_REQUIRE(frames_hwc && out.ptr && mean3 && std3, "frames_from_u8: null pointer");
    SSM_REQUIRE(N > 0 && N <= 65535 && H > 0 && W > 0 && Hp >= H + top && Wp >= W + left && top >= 0 && left >= 0,
                "frames_from_u8: bad geometry %dx%d -> %dx%d at (%d,%d)", H, W, Hp, Wp, top, left);
    SSM_LAUNCH(frames_from_u8_kernel, pix_grid(N, Hp, Wp), dim3(64, 4), 0, (hipStream_t)stream, frames_hwc, out, H, W, Hp,
                       Wp, top, left, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], pad_before_norm);
    return ssm::check_launch("ssm_frames_from_u8_fwd");
}

extern "C" int ssm_frames_to_u8_fwd(ssm_view in, unsigned char *frames_hwc, int N, int H, int W, int top, int left,
                                    const float *mean3, const float *std3, int mode, void *stream) {
    SSM_REQUIRE(frames_hwc && in.ptr && mean3 && std3, "frames_to_u8: null pointer");
    SSM_REQUIRE(N > 0 && N <= 65535 && H > 0 && W > 0 && top >= 0 && left >= 0 && (mode == 0 || mode == 1), "frames_to_u8: bad arguments");
    SSM_LAUNCH(frames_to_u8_kernel, pix_grid(N, H, W), dim3(64, 4), 0, (hipStream_t)stream, in, frames_hwc, H, W, top, left,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], mode);
    return ssm::check_launch("ssm_frames_to_u8_fwd");
}
