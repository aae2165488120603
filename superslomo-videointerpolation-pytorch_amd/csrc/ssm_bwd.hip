// Backward kernels of the training step (BASELINE config 3), fp32 planes.
//
// The reference trains through autograd over the stock ops of scripts/models/*; these are the hand-written
// adjoints of the forward kernels of this library.  What is NOT here is the data-gradient of the convolution:
// dX = conv(dZ, W^T flipped) is the forward kernel (ssm_conv2d_fwd) on a repacked filter.
//
//   lrelu_bwd       dZ = (dY + 1/4 dP[y/2][x/2]) * (Y > 0 ? 1 : slope)        LeakyReLU' and the fused 2x2-mean adjoint
//   bias_grad       db[c] = sum_{b,y,x} dZ
//   wgrad           dW[co][ci][ky][kx] = sum_{b,y,x} dZ[b,co,y,x] * X[b,ci,y+ky-p,x+kx-p]      (fp32 MFMA GEMM over pixels)
//   upsample_cat_bwd  adjoint of cat + bilinear x2 (F.upsample align_corners=False, edge clamp folded in)
//   synth_bwd       adjoint of extract_outputs + compute_output_image, fused with the gradients of the L1
//                   reconstruction loss and of the two stage-2 warp-loss terms (losses.py:113-170,217)
//   inputs_bwd      adjoint of compute_inputs (flow approximation + the two warps wrt their flows), fused with the
//                   two stage-1 warp-loss terms
#include "ssm_common.h"

#include <cstdlib>

namespace {

__device__ __forceinline__ float *vp(const ssm_view &v, int b, int c, int y) {
    return v.ptr + (long long)b * v.sb + (long long)c * v.sc + (long long)y * v.sh;
}

#define SSM_PIXEL_INDEX()                                     \
    const int x = blockIdx.x * 64 + threadIdx.x;              \
    const int y = blockIdx.y * 4 + threadIdx.y;               \
    const int b = blockIdx.z;                                 \
    if (x >= W || y >= H) return;

inline dim3 pix_grid(int B, int H, int W) { return dim3((W + 63) / 64, (H + 3) / 4, B); }

// channels are spread over blockIdx.z in groups of BWD_CPT so small maps with many channels still fill the chip
#define BWD_CPT 4
// dZ = (dY + 0.25 * dPool[y / 2][x / 2]) * LeakyReLU'(Y): the adjoint of layers.conv's activation and of the fused 2x2 mean, over ONE flat
// index (r5): a thread owns VW consecutive pixels of one row of one channel.  A grid shaped after the map
// leaves most lanes of a wave idle on the 11-46 pixel wide maps of a training step (58 launches per step), and on the wide maps four
// pixels per thread move as 16-byte pieces (VW = 4: W a multiple of 4 and 16-byte aligned views; VW = 1 otherwise).
template <int VW>
__global__ __launch_bounds__(256) void lrelu_bwd_flat_kernel(ssm_view dy, ssm_view dpool, ssm_view yv, ssm_view dz, int C, int H, int W,
                                                             float slope, int has_act, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int wq = W / VW;
    const int x = (int)(idx % wq) * VW;
    long long r = idx / wq;
    const int y = (int)(r % H);
    r /= H;
    const int c = (int)(r % C), b = (int)(r / C);
    float g[VW];
    if constexpr (VW == 4) {
        float4 v = dy.ptr ? *reinterpret_cast<const float4 *>(vp(dy, b, c, y) + x) : make_float4(0.f, 0.f, 0.f, 0.f);
        g[0] = v.x, g[1] = v.y, g[2] = v.z, g[3] = v.w;
        if (dpool.ptr) {
            const float2 q = *reinterpret_cast<const float2 *>(vp(dpool, b, c, y >> 1) + (x >> 1));
            g[0] += 0.25f * q.x;
            g[1] += 0.25f * q.x;
            g[2] += 0.25f * q.y;
            g[3] += 0.25f * q.y;
        }
        if (has_act) {
            const float4 a = *reinterpret_cast<const float4 *>(vp(yv, b, c, y) + x);
            g[0] *= a.x > 0.f ? 1.0f : slope;
            g[1] *= a.y > 0.f ? 1.0f : slope;
            g[2] *= a.z > 0.f ? 1.0f : slope;
            g[3] *= a.w > 0.f ? 1.0f : slope;
        }
        *reinterpret_cast<float4 *>(vp(dz, b, c, y) + x) = make_float4(g[0], g[1], g[2], g[3]);
    } else {
        g[0] = dy.ptr ? vp(dy, b, c, y)[x] : 0.f;
        if (dpool.ptr) g[0] += 0.25f * vp(dpool, b, c, y >> 1)[x >> 1];
        if (has_act) g[0] *= (vp(yv, b, c, y)[x] > 0.f) ? 1.0f : slope;
        vp(dz, b, c, y)[x] = g[0];
    }
}

// lrelu_bwd that ALSO writes dZ in the Q8 operand form (include/ssm_hip.h) for the data-gradient convolution on the fp16 + fp8
// matrix path; a thread's 4 channels are half of an 8-channel group.
typedef _Float16 bh4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int bwd_pack4_fp8(float a, float b, float c, float d) {
    const float lim = 448.0f;
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a, -lim, lim), __builtin_amdgcn_fmed3f(b, -lim, lim), 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(c, -lim, lim), __builtin_amdgcn_fmed3f(d, -lim, lim), w, true);
}

__global__ __launch_bounds__(256) void lrelu_bwd_q8_kernel(ssm_view dy, ssm_view dpool, ssm_view yv, ssm_view dz, ssm_hview dq, int C,
                                                           int H, int W, float slope, int has_act, int cgroups) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z / cgroups, cg = blockIdx.z - b * cgroups;
    if (x >= W || y >= H) return;
    float v[BWD_CPT];
#pragma unroll
    for (int i = 0; i < BWD_CPT; ++i) {
        const int c = cg * BWD_CPT + i;
        float g = 0.f;
        if (c < C) {
            g = dy.ptr ? vp(dy, b, c, y)[x] : 0.f;
            if (dpool.ptr) g += 0.25f * vp(dpool, b, c, y >> 1)[x >> 1];
            if (has_act) g *= (vp(yv, b, c, y)[x] > 0.f) ? 1.0f : slope;
            vp(dz, b, c, y)[x] = g;
        }
        v[i] = g;
    }
    const int grp = cg >> 1, half = cg & 1, odd = grp & 1;
    char *rec = (char *)dq.ptr + ((long long)b * dq.sb + (long long)grp * dq.sg + (long long)y * dq.sh + x) * 16;
    bh4 hi;
#pragma unroll
    for (int i = 0; i < 4; ++i) hi[i] = (_Float16)v[i];
    *reinterpret_cast<bh4 *>(rec + half * 8) = hi;
    char *even_rec = rec - odd * dq.sg * 16 + dq.sp * 16;
    *reinterpret_cast<int *>(even_rec + odd * 8 + half * 4) = bwd_pack4_fp8(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<int *>(even_rec + dq.sg * 16 + odd * 8 + half * 4) =
        bwd_pack4_fp8((v[0] - (float)hi[0]) * 2048.f, (v[1] - (float)hi[1]) * 2048.f, (v[2] - (float)hi[2]) * 2048.f, (v[3] - (float)hi[3]) * 2048.f);
}

__global__ __launch_bounds__(256) void bias_grad_kernel(ssm_view dz, float *__restrict__ db, int B, int H, int W) {
    const int c = blockIdx.x;
    float s = 0.f;
    const int rows = B * H;
    for (int r = blockIdx.y * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.y * 4) {       // one wave per image row
        const int bb = r / H, yy = r - bb * H;
        const float *row = vp(dz, bb, c, yy);
        for (int xx = threadIdx.x & 63; xx < W; xx += 64) s += row[xx];
    }
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(db + c, red[0]);
}

// Weight gradient on the fp32 matrix cores (v_mfma_f32_32x32x2_f32):
//     dW[co][col] = sum_pixels dZ[co][p] * X[ci(col)][p + tap(col)],      col = ci*k*k + tap  (OIHW order)
// GEMM with M = couts, N = (ci, tap) columns, K = pixels.  A workgroup (4 waves) owns NTC*32 couts x 256 columns (a wave: CT = 2
// column tiles of 32), walks the image in steps of RR rows x SEG pixels (strided by gridDim.z over the batch's steps) and adds its
// partial sums with fp32 atomics.  Per step the dZ tile [co][RR*SEG] and the XR = KS+RR-1 activation rows of the XC input channels the
// 256 columns touch are staged in LDS (register-prefetched: the loads of step s+1 are issued before the MFMAs of step s).
//   A operand (dZ): rows of 132 floats (33 quads: odd, so the 16 lanes of a ds_read_b128 group hit 16 different quads) - a lane reads
//     FOUR k-steps at once: within a group of 8 pixels MFMA m (0..3) takes pixel m as k = 0 (lanes 0..31) and pixel 4 + m as k = 1
//     (lanes 32..63), so each half-wave reads 4 consecutive pixels (any pairing of pixels with k is a valid order of the sum).
//   B operand (X): every lane owns one column = (channel, filter row, filter column) and reads the staged rows at its constant offset
//     + the pixel: four ds_read_b32 with immediate offsets per group (the filter column shifts the alignment, so no wide read);
//     columns past Cin*k*k read a zeroed row instead of being masked.
// Everything that depends only on the thread (which float4 of a tile it stages, where it lands in LDS, its validity) is computed ONCE:
// the first version recomputed that index arithmetic every step - 5-11 vector instructions per MFMA on the training step's layers
// (tools/pmc_wgrad.sh), and a vector instruction beside the fp32 MFMA costs ~3 matrix cycles (profiles/DESIGN_history_r1-r3.md 3.2g).  SEG x RR = 64x2,
// 32x4 or 16x8 by the map's width: the deep layers' 22- and 11-pixel rows fill a step with more rows instead of padding.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float wg_f4 __attribute__((ext_vector_type(4)));

template <int KS, int XC, int NTC, int SEG, int RR>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(ssm_view x, ssm_view dz, float *__restrict__ dw, float *__restrict__ db, int B,
                                                            int Cin, int Cout, int H, int W, int cin_total, int ci_offset) {
    constexpr int KS2 = KS * KS, PAD = (KS - 1) / 2, CT = 2;
    static_assert(RR * SEG == 128 && SEG % 8 == 0, "a step stages 128 pixels per cout");
    constexpr int DS = RR * SEG + 4;                 // dZ tile row stride: 33 quads
    constexpr int XR = KS + RR - 1;                  // staged activation rows
    constexpr int XV = SEG / 4 + 2;                  // float4 per staged row: columns [xs-4, xs+SEG+4)
    constexpr int RS = 4 * XV + 1;                   // odd row stride (the lanes of a B read differ in row and filter column)
    constexpr int NDZ = NTC * 32 * RR * (SEG / 4);   // float4 in a dZ tile
    constexpr int NX = XC * XR * XV;                 // float4 in an activation tile
    constexpr int LDZ = (NDZ + 255) / 256, LX = (NX + 255) / 256;
    constexpr int ZROW = XC * XR * RS;               // a zeroed row behind the activation tile (columns that do not exist)
    constexpr int OROW = ZROW + SEG + 16;            // ... and a row of ones: the column Cin*k*k of the GEMM is the bias gradient (db != null)
    __shared__ __attribute__((aligned(16))) float sdz[NTC * 32 * DS];
    __shared__ float sx[OROW + SEG + 16];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = tid >> 6;
    const int co0 = blockIdx.y * (NTC * 32);
    const int colbase = blockIdx.x * (128 * CT);
    const int ncols = Cin * KS2;
    const int c_lo = colbase / KS2;                  // first input channel this workgroup touches
    int colOff[CT], cci[CT], tap[CT], rstride[CT];
    bool cvalid[CT], cbias[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int col = colbase + (wid * CT + t) * 32 + l31;      // this lane's (ci, tap) columns
        cvalid[t] = col < ncols;
        cbias[t] = db != nullptr && col == ncols;                 // sum_p dZ[co][p] * 1: no extra MFMA, the lane would idle otherwise
        cci[t] = cvalid[t] ? col / KS2 : c_lo;
        tap[t] = cvalid[t] ? col - cci[t] * KS2 : 0;
        const int ky = tap[t] / KS, kx = tap[t] - ky * KS;
        // + row*RS + pixel = LDS index of this lane's B value of MFMA 0 of a pixel group (k = half: pixels +0 / +4)
        colOff[t] = cvalid[t] ? ((cci[t] - c_lo) * XR + ky) * RS + kx + (4 - PAD) + 4 * half : (cbias[t] ? OROW : ZROW);
        rstride[t] = cvalid[t] ? RS : 0;
    }
    for (int i = tid; i < SEG + 16; i += 256) {
        sx[ZROW + i] = 0.f;
        sx[OROW + i] = 1.f;
    }
    f32x16 acc[NTC][CT];
#pragma unroll
    for (int n = 0; n < NTC; ++n)
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][t][r] = 0.f;

    // ---- what this thread stages, fixed for the whole launch ------------------------------------------------------------------
    int dzo[LDZ], dzl[LDZ], dzq[LDZ];        // global element offset from the step's (b, y, xs) origin | LDS float index | rr + (4 x4 << 8), -1: nothing
#pragma unroll
    for (int i = 0; i < LDZ; ++i) {
        const int f = tid + 256 * i;
        const int c = f / (RR * (SEG / 4)), rem = f - c * (RR * (SEG / 4));
        const int rr = rem / (SEG / 4), x4 = rem - rr * (SEG / 4);
        const bool ok = f < NDZ && co0 + c < Cout;
        dzo[i] = (int)((long long)c * dz.sc) + rr * dz.sh + 4 * x4;
        dzl[i] = c * DS + rr * SEG + 4 * x4;
        dzq[i] = ok ? (rr | ((4 * x4) << 8)) : -1;
        if (!(f < NDZ)) dzl[i] = -1;         // (a cout past Cout still gets its zeros written: the tile row exists)
    }
    int xo[LX], xl[LX], xq[LX];              // the same for the activation rows; xq = ry, -1: nothing to load (zeros are written)
#pragma unroll
    for (int i = 0; i < LX; ++i) {
        const int f = tid + 256 * i;
        const int c = f / (XR * XV), rem = f - c * (XR * XV);
        const int ry = rem / XV, x4 = rem - ry * XV;
        const bool ok = f < NX && c_lo + c < Cin;
        xo[i] = (int)((long long)c * x.sc) + (ry - PAD) * x.sh + 4 * x4 - 4;
        xl[i] = f < NX ? (c * XR + ry) * RS + 4 * x4 : -1;
        xq[i] = ok ? ry : -1;
    }
    const float *dzb0 = dz.ptr + (long long)co0 * dz.sc;
    const float *xb0 = x.ptr + (long long)c_lo * x.sc;

    const int rgroups = (H + RR - 1) / RR, nseg = (W + SEG - 1) / SEG;
    const int nsteps_total = B * rgroups * nseg;     // staging steps of the whole image batch; this workgroup takes every gridDim.z-th
    wg_f4 pdz[LDZ], px[LX];
    auto decode = [&](int s, int &b, int &y, int &xs) {
        const int rp = s / nseg;
        xs = (s - rp * nseg) * SEG;
        b = rp / rgroups;
        y = (rp - b * rgroups) * RR;
    };
    auto prefetch = [&](int s) {
        int b, y, xs;
        decode(s, b, y, xs);
        const float *dzb = dzb0 + (long long)b * dz.sb + (long long)y * dz.sh + xs;
        const float *xb = xb0 + (long long)b * x.sb + (long long)y * x.sh + xs;
#pragma unroll
        for (int i = 0; i < LDZ; ++i) {
            wg_f4 v = {0.f, 0.f, 0.f, 0.f};
            const int rr = dzq[i] & 255, xq4 = dzq[i] >> 8;
            const int left = W - xs - xq4;            // pixels of this float4 inside the row
            if (dzq[i] >= 0 && y + rr < H && left > 0) {
                v = *reinterpret_cast<const wg_f4 *>(dzb + dzo[i]);
                if (left < 4) {                       // zero tail: those products vanish
                    if (left < 2) v[1] = 0.f;
                    if (left < 3) v[2] = 0.f;
                    v[3] = 0.f;
                }
            }
            pdz[i] = v;
        }
#pragma unroll
        for (int i = 0; i < LX; ++i) {
            wg_f4 v = {0.f, 0.f, 0.f, 0.f};
            // rows / columns outside the image come from the padded-plane zero frame; columns past the frame only meet zeroed dZ
            if (xq[i] >= 0 && y + xq[i] - PAD < H + PAD) v = *reinterpret_cast<const wg_f4 *>(xb + xo[i]);
            px[i] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < LDZ; ++i)
            if (dzl[i] >= 0) *reinterpret_cast<wg_f4 *>(sdz + dzl[i]) = pdz[i];
#pragma unroll
        for (int i = 0; i < LX; ++i)
            if (xl[i] >= 0) {
                float *d = sx + xl[i];
                d[0] = px[i][0];
                d[1] = px[i][1];
                d[2] = px[i][2];
                d[3] = px[i][3];
            }
    };

    const wg_f4 *sdz4 = reinterpret_cast<const wg_f4 *>(sdz);
    const int aBase = (l31 * DS) / 4 + half;          // quad index of this lane's A values of pixel group 0 (row n*32 + l31)
    int s = blockIdx.z;
    if (s < nsteps_total) prefetch(s);
    for (; s < nsteps_total; s += gridDim.z) {
        __syncthreads();                             // everyone is done reading the previous tile
        commit();
        __syncthreads();
        if (s + (int)gridDim.z < nsteps_total) prefetch(s + gridDim.z);
        int b, y, xs;
        decode(s, b, y, xs);
        const int ngrp = (min(SEG, W - xs) + 7) >> 3;          // groups of 8 pixels = 4 MFMA k-steps
#pragma unroll
        for (int rr = 0; rr < RR; ++rr) {
            const float *bp0 = sx + colOff[0] + rr * rstride[0], *bp1 = sx + colOff[1] + rr * rstride[1];
            const wg_f4 *ap = sdz4 + aBase + (rr * SEG) / 4;
            for (int g = 0; g < ngrp; ++g) {
                wg_f4 av[NTC];
#pragma unroll
                for (int n = 0; n < NTC; ++n) av[n] = ap[n * (32 * DS / 4) + 2 * g];
                float bv[CT][4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    bv[0][m] = bp0[8 * g + m];
                    bv[1][m] = bp1[8 * g + m];
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < NTC; ++n)
#pragma unroll
                        for (int t = 0; t < CT; ++t) acc[n][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[n][m], bv[t][m], acc[n][t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < CT; ++t)
        if (cvalid[t]) {
#pragma unroll
            for (int n = 0; n < NTC; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + n * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co < Cout) atomicAdd(dw + ((long long)co * cin_total + ci_offset + cci[t]) * KS2 + tap[t], acc[n][t][r]);
                }
        } else if (cbias[t]) {
#pragma unroll
            for (int n = 0; n < NTC; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + n * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co < Cout) atomicAdd(db + co, acc[n][t][r]);
                }
        }
}

// Weight gradient on the bf16 matrix cores with split operands (v_mfma_f32_32x32x16_bf16, the training plan of mode f16f8):
//     v = hi + lo + r,  hi = bf16(v),  lo = bf16(v - hi),  |r| <= 2^-17 |v|;     dz*x ~= hi*hi + hi*lo + lo*hi   (fp32 accumulate)
// bf16 keeps fp32's exponent, so gradients of any magnitude need no scaling; the dropped terms are ~2^-17 relative per product.
// GEMM per filter tap: D[co][ci] = sum_p dZ[co][p] * X[ci][p + tap],  M = 32 couts (A operand, rows of dZ, 8 consecutive pixels per
// lane = one aligned 16-byte LDS read of the bf16 row), N = 32 input channels (B operand), K = 16 pixels of one image row.  The
// horizontal tap offset kx is applied in REGISTERS: a lane reads the 16-pixel window [q-4, q+12) of its channel's row once (8+16+8
// bytes, aligned) and every kx takes its 8 pixels from it - whole dwords for even offsets, v_alignbit_b32 for odd ones - so one
// window feeds KS taps x 3 products.  The vertical offset selects the staged row: a workgroup owns TY filter rows (all 3 of a 3x3
// filter, with the 3 activation rows kept in an LDS ring while it walks down the image; one row of a 5x5/7x7 filter) of a
// (32*WAN couts) x (32*WBN input channels) tile; its 4 waves split the tile and, when the tile is smaller than four 32x32 blocks,
// the 16-pixel steps of a segment (WKN).  fp32 planes are converted to hi/lo bf16 while staging (register-prefetched: the loads of
// step s+1 are issued before the MFMAs of step s).  Image rows are split over gridDim.z; partial sums are added with fp32 atomics.
typedef __bf16 wg_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf2 __attribute__((ext_vector_type(2)));
typedef float wg_f2 __attribute__((ext_vector_type(2)));
typedef int wg_i4 __attribute__((ext_vector_type(4)));
typedef int wg_i2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned wg_pk_bf16(float a, float b) {        // low half = bf16(a), high half = bf16(b), round to nearest even
    const wg_f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wg_bf2));
}

// 4 fp32 -> hi (2 dwords) and lo (2 dwords)
__device__ __forceinline__ void wg_split4(const float4 &v, wg_i2 &hi, wg_i2 &lo) {
    const unsigned h0 = wg_pk_bf16(v.x, v.y), h1 = wg_pk_bf16(v.z, v.w);
    hi = wg_i2{(int)h0, (int)h1};
    lo = wg_i2{(int)wg_pk_bf16(v.x - __uint_as_float(h0 << 16), v.y - __uint_as_float(h0 & 0xffff0000u)),
               (int)wg_pk_bf16(v.z - __uint_as_float(h1 << 16), v.w - __uint_as_float(h1 & 0xffff0000u))};
}

constexpr int wg_odd16(int bytes) { return (((bytes + 15) / 16) | 1) * 16; }      // LDS row pitch: an odd number of 16-byte units

// (launch bounds: two workgroups per CU except the 7-tap / 128-pixel instantiation, whose 7 windows x 3 products per lane need more than
// the 128 registers two co-resident workgroups leave a wave - at (256, 2) it spilled 4 registers to scratch; check_isa.sh fences that)
template <int KS, int TY, int SEG, int WAN, int WBN>
__global__ __launch_bounds__(256, (KS == 7 && SEG == 128) ? 1 : 2) void wgrad_bf16x3_kernel(ssm_view x, ssm_view dz, float *__restrict__ dw, int B, int Cin, int Cout,
                                                              int H, int W, int cin_total, int ci_offset, int steps_per_slice) {
    constexpr int P = (KS - 1) / 2, WKN = 4 / (WAN * WBN), NK = SEG / 16, NKY = KS / TY;
    static_assert(WAN * WBN * WKN == 4 && NK % WKN == 0 && NKY * TY == KS && P <= 3, "tile configuration");
    constexpr int COT = 32 * WAN, CIT = 32 * WBN;
    constexpr int DZB = wg_odd16(SEG * 2);           // bytes per staged dZ row (bf16)
    constexpr int XE = SEG + 16;                     // staged activation columns [xs-8, xs+SEG+8)
    constexpr int XB = wg_odd16(XE * 2);
    constexpr int NDZ4 = COT * (SEG / 4), NX4 = CIT * (XE / 4);
    constexpr int LDZ = (NDZ4 + 255) / 256, LX = (NX4 + 255) / 256;
    constexpr int TAPS = TY * KS;
    constexpr int EPI_FLOATS = WAN * 8 * CIT * TAPS;  // epilogue: one 8-row band of every cout tile, in dW order [co][ci][ky][kx]
    constexpr int STAGE_BYTES = 2 * COT * DZB + 2 * TY * CIT * XB;
    constexpr int SMEM_BYTES = STAGE_BYTES > 4 * WKN * EPI_FLOATS ? STAGE_BYTES : 4 * WKN * EPI_FLOATS;
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    char *const s_dzh = smem, *const s_dzl = smem + COT * DZB, *const s_xh = smem + 2 * COT * DZB,
                *const s_xl = smem + 2 * COT * DZB + TY * CIT * XB;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5, wid = tid >> 6;
    const int wa = wid / (WBN * WKN), wb = (wid / WKN) % WBN, wk = wid % WKN;
    const int ci0 = blockIdx.x * CIT;
    const int cog = blockIdx.y / NKY, ky0 = (blockIdx.y - cog * NKY) * TY;
    const int co0 = cog * COT;
    const int nseg = (W + SEG - 1) / SEG;
    const int total = B * nseg * H;
    const int s0 = blockIdx.z * steps_per_slice;
    const int s1 = min(total, s0 + steps_per_slice);
    if (s0 >= s1) return;

    f32x16 acc[TY][KS];
#pragma unroll
    for (int a = 0; a < TY; ++a)
#pragma unroll
        for (int t = 0; t < KS; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;

    auto decode = [&](int s, int &b, int &y, int &xs) {      // y fastest: a slice walks down the image, the ring stays warm
        const int t = s / H;
        y = s - t * H;
        b = t / nseg;
        xs = (t - b * nseg) * SEG;
    };
    // one float4 of an activation row: columns col..col+3 of row yy (may lie in the zero frame), zero outside [-4, W+4)
    auto load_x4 = [&](int b, int c, int yy, int col) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ci0 + c < Cin && col >= -4 && col < W + 4) {
            v = *reinterpret_cast<const float4 *>(vp(x, b, ci0 + c, yy) + col);
            const int left = W + 4 - col;
            if (left < 4) {
                if (left < 2) v.y = 0.f;
                if (left < 3) v.z = 0.f;
                v.w = 0.f;
            }
        }
        return v;
    };
    auto store_x4 = [&](int slot, int f, const float4 &v) {
        const int c = f / (XE / 4), x4 = f - c * (XE / 4);
        wg_i2 hi, lo;
        wg_split4(v, hi, lo);
        const int off = (slot * CIT + c) * XB + 8 * x4;
        *reinterpret_cast<wg_i2 *>(s_xh + off) = hi;
        *reinterpret_cast<wg_i2 *>(s_xl + off) = lo;
    };
    float4 pdz[LDZ], px[LX];
    auto prefetch = [&](int s) {
        int b, y, xs;
        decode(s, b, y, xs);
#pragma unroll
        for (int i = 0; i < LDZ; ++i) {
            const int f = tid + 256 * i;
            const int c = f / (SEG / 4), x4 = f - c * (SEG / 4);
            const int col = xs + 4 * x4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < NDZ4 && co0 + c < Cout && col < W) {
                v = *reinterpret_cast<const float4 *>(vp(dz, b, co0 + c, y) + col);
                const int left = W - col;                  // zero tail: those products vanish
                if (left < 4) {
                    if (left < 2) v.y = 0.f;
                    if (left < 3) v.z = 0.f;
                    v.w = 0.f;
                }
            }
            pdz[i] = v;
        }
        const int yy = y + ky0 + TY - 1 - P;               // the newest activation row of this step
#pragma unroll
        for (int i = 0; i < LX; ++i) {
            const int f = tid + 256 * i;
            const int c = f / (XE / 4), x4 = f - c * (XE / 4);
            px[i] = f < NX4 ? load_x4(b, c, yy, xs - 8 + 4 * x4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](int s) {
        int b, y, xs;
        decode(s, b, y, xs);
#pragma unroll
        for (int i = 0; i < LDZ; ++i) {
            const int f = tid + 256 * i;
            if (f < NDZ4) {
                const int c = f / (SEG / 4), x4 = f - c * (SEG / 4);
                wg_i2 hi, lo;
                wg_split4(pdz[i], hi, lo);
                *reinterpret_cast<wg_i2 *>(s_dzh + c * DZB + 8 * x4) = hi;
                *reinterpret_cast<wg_i2 *>(s_dzl + c * DZB + 8 * x4) = lo;
            }
        }
        const int slot = (y + ky0 + TY - 1 - P + 3) % TY;
#pragma unroll
        for (int i = 0; i < LX; ++i) {
            const int f = tid + 256 * i;
            if (f < NX4) store_x4(slot, f, px[i]);
        }
        if (TY > 1 && (s == s0 || y == 0)) {              // ring is cold: stage the older rows of this step directly
            for (int r = 0; r < TY - 1; ++r) {
                const int yy = y + ky0 + r - P;
                const int sl = (yy + 3) % TY;
                for (int f = tid; f < NX4; f += 256) {
                    const int c = f / (XE / 4), x4 = f - c * (XE / 4);
                    store_x4(sl, f, load_x4(b, c, yy, xs - 8 + 4 * x4));
                }
            }
        }
    };

    prefetch(s0);
    for (int s = s0; s < s1; ++s) {
        __syncthreads();                                   // everyone is done reading the previous step's tiles
        commit(s);
        __syncthreads();
        if (s + 1 < s1) prefetch(s + 1);
        int b, y, xs;
        decode(s, b, y, xs);
        const int nk = (min(SEG, W - xs) + 15) >> 4;       // 16-pixel steps that hold image pixels
        for (int j = wk; j < nk; j += WKN) {
            const int q = 16 * j + 8 * half;               // this lane's 8 pixels: columns xs+q .. xs+q+7
            const int arow = (wa * 32 + l31) * DZB + 2 * q;
            const wg_i4 ah = *reinterpret_cast<const wg_i4 *>(s_dzh + arow);
            const wg_i4 al = *reinterpret_cast<const wg_i4 *>(s_dzl + arow);
            const wg_bf8 Ah = __builtin_bit_cast(wg_bf8, ah), Al = __builtin_bit_cast(wg_bf8, al);
#pragma unroll
            for (int a = 0; a < TY; ++a) {
                const int slot = (y + ky0 + a - P + 3) % TY;
                const int xrow = (slot * CIT + wb * 32 + l31) * XB + 2 * (q + 8);      // staged column 0 = image column xs-8
                int wh[8], wl[8];                          // window: pixels q-4 .. q+11 of this lane's channel, 2 per dword
                {
                    const wg_i2 h0 = *reinterpret_cast<const wg_i2 *>(s_xh + xrow - 8), l0 = *reinterpret_cast<const wg_i2 *>(s_xl + xrow - 8);
                    const wg_i4 h1 = *reinterpret_cast<const wg_i4 *>(s_xh + xrow), l1 = *reinterpret_cast<const wg_i4 *>(s_xl + xrow);
                    const wg_i2 h2 = *reinterpret_cast<const wg_i2 *>(s_xh + xrow + 16), l2 = *reinterpret_cast<const wg_i2 *>(s_xl + xrow + 16);
                    wh[0] = h0[0]; wh[1] = h0[1]; wh[2] = h1[0]; wh[3] = h1[1]; wh[4] = h1[2]; wh[5] = h1[3]; wh[6] = h2[0]; wh[7] = h2[1];
                    wl[0] = l0[0]; wl[1] = l0[1]; wl[2] = l1[0]; wl[3] = l1[1]; wl[4] = l1[2]; wl[5] = l1[3]; wl[6] = l2[0]; wl[7] = l2[1];
                }
#pragma unroll
                for (int t = 0; t < KS; ++t) {
                    const int idx = t - P + 4;             // window element of the first pixel this tap needs (compile-time after unrolling)
                    const int d = idx >> 1;
                    wg_i4 bh, bl;
                    if ((idx & 1) == 0) {
                        bh = wg_i4{wh[d], wh[d + 1], wh[d + 2], wh[d + 3]};
                        bl = wg_i4{wl[d], wl[d + 1], wl[d + 2], wl[d + 3]};
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            bh[i] = (int)__builtin_amdgcn_alignbit((unsigned)wh[d + i + 1], (unsigned)wh[d + i], 16);
                            bl[i] = (int)__builtin_amdgcn_alignbit((unsigned)wl[d + i + 1], (unsigned)wl[d + i], 16);
                        }
                    }
                    const wg_bf8 Bh = __builtin_bit_cast(wg_bf8, bh), Bl = __builtin_bit_cast(wg_bf8, bl);
                    acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[a][t], 0, 0, 0);
                    acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc[a][t], 0, 0, 0);
                    acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc[a][t], 0, 0, 0);
                }
            }
        }
    }
    // Epilogue.  The accumulators have the input channel on the lane, i.e. a stride of KS*KS floats in dW: written straight out, every
    // lane's atomic would be its own L2 operation on a line that hundreds of other workgroups also hit (measured: 10x the MFMA time).
    // So the tile goes through LDS in dW order, 8 cout rows at a time: each of the WKN waves that share a tile stores its partial in
    // its own copy, the copies are summed on the way out, and the global atomics run along contiguous (ci, ky, kx) runs - 16 lanes per
    // cache line.  The barriers wait for LDS only (not for the atomics in flight), so a band's atomics overlap the next band's stores.
    float *const tile = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // staging buffers (g = 0) / the previous band are no longer read
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = 4 * half + rr;                 // cout row within the band: co = co0 + wa*32 + 8*g + row
            float *d = tile + wk * EPI_FLOATS + ((wa * 8 + row) * CIT + wb * 32 + l31) * TAPS;
#pragma unroll
            for (int a = 0; a < TY; ++a)
#pragma unroll
                for (int t = 0; t < KS; ++t) d[a * KS + t] = acc[a][t][4 * g + rr];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int e = tid; e < EPI_FLOATS; e += 256) {
            const int wr = e / (CIT * TAPS), rem = e - wr * (CIT * TAPS);      // wr = wa*8 + row
            const int cil = rem / TAPS, tp = rem - cil * TAPS;
            const int co = co0 + (wr >> 3) * 32 + 8 * g + (wr & 7);
            const int ci = ci0 + cil;
            float v = tile[e];
#pragma unroll
            for (int w = 1; w < WKN; ++w) v += tile[w * EPI_FLOATS + e];
            if (co < Cout && ci < Cin)
                atomicAdd(dw + ((long long)co * cin_total + ci_offset + ci) * (KS * KS) + ky0 * KS + tp, v);
        }
    }
}

// adjoint of  U = upsample2x(cat[a, b])  (see upsample2x_cat_kernel): one thread per LOW-res pixel and channel
// gathers its 4x4 hi-res neighbourhood.  1-D weights of x[i] in U(Y): Y=2i: .75 (1 at i=0); Y=2i+1: .75 (1 at
// i=h-1); Y=2i+2: .25 if i+1<h; Y=2i-1: .25 if i>0.
__global__ __launch_bounds__(256) void upsample_cat_bwd_kernel(ssm_view du, ssm_view da, int Ca, ssm_view dbv, int Cb, int H, int W,
                                                               int acc_a, int acc_b, int cgroups, ssm_view ya, float msl) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;      // H, W = LOW-res dims
    const int b = blockIdx.z / cgroups, cg = blockIdx.z - b * cgroups;
    if (x >= W || y >= H) return;
    float wy[4], wx[4];
    wy[0] = y > 0 ? 0.25f : 0.f;
    wy[1] = y > 0 ? 0.75f : 1.0f;
    wy[2] = y < H - 1 ? 0.75f : 1.0f;
    wy[3] = y < H - 1 ? 0.25f : 0.f;
    wx[0] = x > 0 ? 0.25f : 0.f;
    wx[1] = x > 0 ? 0.75f : 1.0f;
    wx[2] = x < W - 1 ? 0.75f : 1.0f;
    wx[3] = x < W - 1 ? 0.25f : 0.f;
    const int C = Ca + Cb;
#pragma unroll
    for (int ci = 0; ci < BWD_CPT; ++ci) {
        const int c = cg * BWD_CPT + ci;
        if (c >= C) break;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int Y = 2 * y - 1 + j;
            if (wy[j] == 0.f) continue;
            const float *row = vp(du, b, c, Y);
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (wx[i] != 0.f) t += wx[i] * row[2 * x - 1 + i];
            s += wy[j] * t;
        }
        if (c < Ca) {
            float *d = vp(da, b, c, y) + x;
            float o = acc_a ? *d + s : s;
            if (ya.ptr) o *= vp(ya, b, c, y)[x] > 0.f ? 1.f : msl;          // dZ of the layer that produced `a`: x LeakyReLU'(its output)
            *d = o;
        } else {
            float *d = vp(dbv, b, c - Ca, y) + x;
            *d = acc_b ? *d + s : s;
        }
    }
}

// The same adjoint for even W and 16-byte aligned hi-res rows (r5): a thread owns TWO neighbouring low-res pixels of one channel - the six
// hi-res columns 2x-1 .. 2x+4 of a row arrive as one 16-byte piece + two scalars (12 loads for two outputs instead of 32) - over one flat
// index (the maps are 22-176 pixels wide: a grid shaped after the map leaves lanes idle).
__global__ __launch_bounds__(256) void upsample_cat_bwd2_kernel(ssm_view du, ssm_view da, int Ca, ssm_view dbv, int Cb, int H, int W, int acc_a,
                                                                int acc_b, long long total, ssm_view ya, float msl) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int C = Ca + Cb, wp = W / 2;
    const int x = 2 * (int)(idx % wp);          // H, W = LOW-res dims
    long long r = idx / wp;
    const int y = (int)(r % H);
    r /= H;
    const int c = (int)(r % C), b = (int)(r / C);
    const float wy[4] = {y > 0 ? 0.25f : 0.f, y > 0 ? 0.75f : 1.0f, y < H - 1 ? 0.75f : 1.0f, y < H - 1 ? 0.25f : 0.f};
    // 1-D weights of the six columns for the two outputs x (columns 0..3) and x + 1 (columns 2..5)
    const float a0 = x > 0 ? 0.25f : 0.f, a1 = x > 0 ? 0.75f : 1.0f, a2 = 0.75f, a3 = 0.25f;                       // x < W - 1 always (W even)
    const float b2 = 0.25f, b3 = 0.75f, b4 = x + 1 < W - 1 ? 0.75f : 1.0f, b5 = x + 1 < W - 1 ? 0.25f : 0.f;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (wy[j] == 0.f) continue;
        const float *row = vp(du, b, c, 2 * y - 1 + j) + 2 * x;
        const float4 m = *reinterpret_cast<const float4 *>(row);          // columns 2x .. 2x+3
        const float l = a0 != 0.f ? row[-1] : 0.f, rr = b5 != 0.f ? row[4] : 0.f;
        // (same association as the one-pixel kernel: left to right within a row, rows top to bottom)
        float t0 = 0.f, t1 = 0.f;
        if (a0 != 0.f) t0 += a0 * l;
        t0 += a1 * m.x;
        t0 += a2 * m.y;
        t0 += a3 * m.z;
        t1 += b2 * m.y;
        t1 += b3 * m.z;
        t1 += b4 * m.w;
        if (b5 != 0.f) t1 += b5 * rr;
        s0 += wy[j] * t0;
        s1 += wy[j] * t1;
    }
    float2 *d = reinterpret_cast<float2 *>((c < Ca ? vp(da, b, c, y) : vp(dbv, b, c - Ca, y)) + x);
    const bool acc = c < Ca ? acc_a : acc_b;
    float2 o = make_float2(s0, s1);
    if (acc) {
        const float2 p = *d;
        o.x += p.x;
        o.y += p.y;
    }
    if (ya.ptr && c < Ca) {
        const float2 m = *reinterpret_cast<const float2 *>(vp(ya, b, c, y) + x);
        o.x *= m.x > 0.f ? 1.f : msl;
        o.y *= m.y > 0.f ? 1.f : msl;
    }
    *d = o;
}

// ---- bilinear sampler with derivatives --------------------------------------------------------------------
struct TapsD {
    int o00, o01, o10, o11;
    float fx, fy;                // ix - x0, iy - y0
};

__device__ __forceinline__ TapsD make_taps_d(int x, int y, float u, float v, int H, int W, int sh) {
    const float wd = (float)(W - 1 > 1 ? W - 1 : 1), hd = (float)(H - 1 > 1 ? H - 1 : 1);
    const float gx = 2.0f * ((float)x + u) / wd - 1.0f, gy = 2.0f * ((float)y + v) / hd - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    const float x0 = floorf(ix), y0 = floorf(iy), x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    const float wm = (float)(W - 1), hm = (float)(H - 1);
    const bool bx0 = x0 >= 0.f && x0 <= wm, bx1 = x1 >= 0.f && x1 <= wm;
    const bool by0 = y0 >= 0.f && y0 <= hm, by1 = y1 >= 0.f && y1 <= hm;
    const int xi0 = bx0 ? (int)x0 : 0, xi1 = bx1 ? (int)x1 : 0, yi0 = by0 ? (int)y0 : 0, yi1 = by1 ? (int)y1 : 0;
    TapsD t;
    t.o00 = (bx0 && by0) ? yi0 * sh + xi0 : -1;
    t.o01 = (bx1 && by0) ? yi0 * sh + xi1 : -1;
    t.o10 = (bx0 && by1) ? yi1 * sh + xi0 : -1;
    t.o11 = (bx1 && by1) ? yi1 * sh + xi1 : -1;
    t.fx = ix - x0;
    t.fy = iy - y0;
    return t;
}

// value and d/d(ix), d/d(iy) of the zero-padded bilinear sample (d ix / d u = 1: the reference's normalise /
// un-normalise pair is the identity map)
__device__ __forceinline__ void sample_d(const float *__restrict__ plane, const TapsD &t, float &val, float &dvx, float &dvy) {
    const float a = t.o00 >= 0 ? plane[t.o00] : 0.f, b = t.o01 >= 0 ? plane[t.o01] : 0.f;
    const float c = t.o10 >= 0 ? plane[t.o10] : 0.f, d = t.o11 >= 0 ? plane[t.o11] : 0.f;
    const float gx = 1.0f - t.fx, gy = 1.0f - t.fy;
    val = a * gx * gy + b * t.fx * gy + c * gx * t.fy + d * t.fx * t.fy;
    dvx = (b - a) * gy + (d - c) * t.fy;
    dvy = (c - a) * gx + (d - b) * t.fx;
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// Adjoint of the stand-alone backward warp (layers.warp): dflow = sum_c dy[c] * d sample / d(u,v); dimg (ptr NULL = not
// wanted) += dy[c] scattered with the four bilinear weights (atomics: neighbouring outputs share taps).
__global__ __launch_bounds__(256) void warp_bwd_kernel(ssm_view img, ssm_view flow, ssm_view dy, ssm_view dflow, ssm_view dimg, int C,
                                                       int H, int W) {
    SSM_PIXEL_INDEX();
    const TapsD t = make_taps_d(x, y, vp(flow, b, 0, y)[x], vp(flow, b, 1, y)[x], H, W, img.sh);
    const float gx = 1.0f - t.fx, gy = 1.0f - t.fy;
    float du = 0.f, dv = 0.f;
    for (int c = 0; c < C; ++c) {
        float val, sx, sy;
        sample_d(vp(img, b, c, 0), t, val, sx, sy);
        const float g = vp(dy, b, c, y)[x];
        du += g * sx;
        dv += g * sy;
        if (dimg.ptr) {        // same tap offsets: dimg must share img's row stride
            float *pl = vp(dimg, b, c, 0);
            if (t.o00 >= 0) atomicAdd(pl + t.o00, g * gx * gy);
            if (t.o01 >= 0) atomicAdd(pl + t.o01, g * t.fx * gy);
            if (t.o10 >= 0) atomicAdd(pl + t.o10, g * gx * t.fy);
            if (t.o11 >= 0) atomicAdd(pl + t.o11, g * t.fx * t.fy);
        }
    }
    if (dflow.ptr) {
        vp(dflow, b, 0, y)[x] = du;
        vp(dflow, b, 1, y)[x] = dv;
    }
}

// Loss + synthesis adjoint.  cr[b], cw[b] = per-sample coefficients of d(L1 recon mean) and d(L1 warp mean)
// (lambda * upstream / (3*H*W)); stage2_terms = 0 when STAGE2.FREEZE drops the two refined-flow warp terms.
// Outputs: dout5 [B,5,H,W] and dest [B,4,H,W] (gradient wrt the approximated flows Ft1^(u,v) | Ft0^(u,v)).
__global__ __launch_bounds__(256) void synth_bwd_kernel(ssm_view img6, ssm_view est, ssm_view out5, ssm_view target,
                                                        const float *__restrict__ tarr, const float *__restrict__ cr,
                                                        const float *__restrict__ cw, ssm_view dyx, ssm_view dout5, ssm_view dest,
                                                        int H, int W, int stage2_terms) {
    SSM_PIXEL_INDEX();
    const float t = tarr[b], omt = 1.0f - t;
    const float v1 = 1.0f / (1.0f + expf(-vp(out5, b, 0, y)[x])), v0 = 1.0f - v1;
    const float ft1u = vp(est, b, 0, y)[x] + vp(out5, b, 1, y)[x], ft1v = vp(est, b, 1, y)[x] + vp(out5, b, 2, y)[x];
    const float ft0u = vp(est, b, 2, y)[x] + vp(out5, b, 3, y)[x], ft0v = vp(est, b, 3, y)[x] + vp(out5, b, 4, y)[x];
    const TapsD t0 = make_taps_d(x, y, ft0u, ft0v, H, W, img6.sh), t1 = make_taps_d(x, y, ft1u, ft1v, H, W, img6.sh);
    const float den = omt * v0 + t * v1;
    float dv0 = 0.f, dv1 = 0.f, dden = 0.f, d0x = 0.f, d0y = 0.f, d1x = 0.f, d1y = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float w0, w0x, w0y, w1, w1x, w1y;
        sample_d(vp(img6, b, c, 0), t0, w0, w0x, w0y);
        sample_d(vp(img6, b, 3 + c, 0), t1, w1, w1x, w1y);
        const float tg = vp(target, b, c, y)[x];
        const float num = omt * v0 * w0 + t * v1 * w1;
        const float pred = num / den;
        float dpred = cr[b] * sgn(pred - tg);
        if (dyx.ptr) dpred += vp(dyx, b, c, y)[x];     // gradient of the other loss terms (perceptual) wrt the frame
        const float dnum = dpred / den;
        dden -= dpred * pred / den;
        float dw0 = dnum * omt * v0, dw1 = dnum * t * v1;
        dv0 += dnum * omt * w0;
        dv1 += dnum * t * w1;
        if (stage2_terms) {        // |g(I0,Ft0) - I_t| + |g(I1,Ft1) - I_t|
            dw0 += cw[b] * sgn(w0 - tg);
            dw1 += cw[b] * sgn(w1 - tg);
        }
        d0x += dw0 * w0x;
        d0y += dw0 * w0y;
        d1x += dw1 * w1x;
        d1y += dw1 * w1y;
    }
    dv0 += dden * omt;
    dv1 += dden * t;
    vp(dout5, b, 0, y)[x] = (dv1 - dv0) * v1 * v0;
    vp(dout5, b, 1, y)[x] = d1x;
    vp(dout5, b, 2, y)[x] = d1y;
    vp(dout5, b, 3, y)[x] = d0x;
    vp(dout5, b, 4, y)[x] = d0y;
    vp(dest, b, 0, y)[x] = d1x;
    vp(dest, b, 1, y)[x] = d1y;
    vp(dest, b, 2, y)[x] = d0x;
    vp(dest, b, 3, y)[x] = d0y;
}

// compute_inputs adjoint: din16 [B,16,H,W] (gradient of the stage-2 input) and dest (from synth_bwd) -> dflow4
// (gradient wrt stage 1's F01 | F10), plus the two stage-1 warp-loss terms |g(I1,F01)-I0| + |g(I0,F10)-I1|.
__global__ __launch_bounds__(256) void inputs_bwd_kernel(ssm_view img6, ssm_view flow4, ssm_view din16, ssm_view dest,
                                                         const float *__restrict__ tarr, const float *__restrict__ cw,
                                                         ssm_view dflow4, int H, int W, int stage1_terms) {
    SSM_PIXEL_INDEX();
    const float t = tarr[b], omt = 1.0f - t;
    const float f01u = vp(flow4, b, 0, y)[x], f01v = vp(flow4, b, 1, y)[x];
    const float f10u = vp(flow4, b, 2, y)[x], f10v = vp(flow4, b, 3, y)[x];
    const float c00 = (-omt) * t, c01 = t * t, c10 = omt * omt, c11 = t * omt;
    const float ft0u = c00 * f01u + c01 * f10u, ft0v = c00 * f01v + c01 * f10v;
    const float ft1u = c10 * f01u - c11 * f10u, ft1v = c10 * f01v - c11 * f10v;
    const TapsD t1 = make_taps_d(x, y, ft1u, ft1v, H, W, img6.sh), t0 = make_taps_d(x, y, ft0u, ft0v, H, W, img6.sh);
    // in16 = [I1(0:3), g(I1,Ft1)(3:6), Ft1(6:8), Ft0(8:10), g(I0,Ft0)(10:13), I0(13:16)]
    float d1u = vp(din16, b, 6, y)[x] + vp(dest, b, 0, y)[x], d1v = vp(din16, b, 7, y)[x] + vp(dest, b, 1, y)[x];
    float d0u = vp(din16, b, 8, y)[x] + vp(dest, b, 2, y)[x], d0v = vp(din16, b, 9, y)[x] + vp(dest, b, 3, y)[x];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float val, gx, gy;
        sample_d(vp(img6, b, 3 + c, 0), t1, val, gx, gy);
        const float g1 = vp(din16, b, 3 + c, y)[x];
        d1u += g1 * gx;
        d1v += g1 * gy;
        sample_d(vp(img6, b, c, 0), t0, val, gx, gy);
        const float g0 = vp(din16, b, 10 + c, y)[x];
        d0u += g0 * gx;
        d0v += g0 * gy;
    }
    float dF01u = c10 * d1u + c00 * d0u, dF01v = c10 * d1v + c00 * d0v;
    float dF10u = -c11 * d1u + c01 * d0u, dF10v = -c11 * d1v + c01 * d0v;
    if (stage1_terms) {
        const TapsD a = make_taps_d(x, y, f01u, f01v, H, W, img6.sh), bq = make_taps_d(x, y, f10u, f10v, H, W, img6.sh);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float val, gx, gy;
            sample_d(vp(img6, b, 3 + c, 0), a, val, gx, gy);           // g(I1, F01) vs I0
            float s = cw[b] * sgn(val - vp(img6, b, c, y)[x]);
            dF01u += s * gx;
            dF01v += s * gy;
            sample_d(vp(img6, b, c, 0), bq, val, gx, gy);              // g(I0, F10) vs I1
            s = cw[b] * sgn(val - vp(img6, b, 3 + c, y)[x]);
            dF10u += s * gx;
            dF10v += s * gy;
        }
    }
    vp(dflow4, b, 0, y)[x] = dF01u;
    vp(dflow4, b, 1, y)[x] = dF01v;
    vp(dflow4, b, 2, y)[x] = dF10u;
    vp(dflow4, b, 3, y)[x] = dF10v;
}


// ---- perceptual-loss pieces (VGG16 features[:23], scripts/models/losses.py:12-41) ---------------------------------
// MaxPool2d(2,2): y = max of the 2x2 window.  Backward routes the gradient to the FIRST maximum of the window in
// row-major scan order (what torch's max_pool2d backward does with its saved argmax).
#define SSM_PIXEL_CGROUP()                                            \
    const int x = blockIdx.x * 64 + threadIdx.x;                      \
    const int y = blockIdx.y * 4 + threadIdx.y;                       \
    const int b = blockIdx.z / cgroups;                               \
    const int c0 = (blockIdx.z - b * cgroups) * BWD_CPT;              \
    const int c1 = c0 + BWD_CPT < C ? c0 + BWD_CPT : C;               \
    if (x >= W || y >= H) return;

__global__ __launch_bounds__(256) void maxpool2_kernel(ssm_view xin, ssm_view yout, int C, int H, int W, int cgroups) {   // H, W = OUTPUT dims
    SSM_PIXEL_CGROUP();
    for (int c = c0; c < c1; ++c) {
        const float *r0 = vp(xin, b, c, 2 * y) + 2 * x, *r1 = vp(xin, b, c, 2 * y + 1) + 2 * x;
        vp(yout, b, c, y)[x] = fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r1[0], r1[1]));
    }
}

__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(ssm_view xin, ssm_view dy, ssm_view dx, int C, int H, int W, int cgroups) {   // pooled dims
    SSM_PIXEL_CGROUP();
    for (int c = c0; c < c1; ++c) {
        const float *r0 = vp(xin, b, c, 2 * y) + 2 * x, *r1 = vp(xin, b, c, 2 * y + 1) + 2 * x;
        const float v[4] = {r0[0], r0[1], r1[0], r1[1]};
        int am = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (v[i] > v[am]) am = i;
        const float g = vp(dy, b, c, y)[x];
        float *d0 = vp(dx, b, c, 2 * y) + 2 * x, *d1 = vp(dx, b, c, 2 * y + 1) + 2 * x;
        d0[0] = am == 0 ? g : 0.f;
        d0[1] = am == 1 ? g : 0.f;
        d1[0] = am == 2 ? g : 0.f;
        d1[1] = am == 3 ? g : 0.f;
    }
}

// out = coef[b] * (a - b): gradient of coef/2 * sum (a-b)^2 wrt a (the MSE feature loss, losses.py:40,218-233)
__global__ __launch_bounds__(256) void sqdiff_grad_kernel(ssm_view a, ssm_view bb, const float *__restrict__ coef, ssm_view out, int C,
                                                          int H, int W, int cgroups) {
    SSM_PIXEL_CGROUP();
    const float k = coef[b];
    for (int c = c0; c < c1; ++c) vp(out, b, c, y)[x] = k * (vp(a, b, c, y)[x] - vp(bb, b, c, y)[x]);
}

// per-sample mean of (a - b)^2 over C x H x W in two deterministic launches: SQD_CHUNKS blocks per sample each reduce a contiguous slice
// (in-thread sums, then a fixed tree through LDS), then one thread per sample adds the slices in order (r5: replaces four strided torch
// kernels - subtract, square, copy to contiguous, reduce - of the perceptual term's forward)
#define SQD_CHUNKS 64
__global__ __launch_bounds__(256) void sqdiff_partial_kernel(ssm_view a, ssm_view bb, float *__restrict__ partial, int C, int H, int W) {
    __shared__ float red[256];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const long long n = (long long)C * H * W, per = (n + SQD_CHUNKS - 1) / SQD_CHUNKS;
    const long long e0 = chunk * per, e1 = e0 + per < n ? e0 + per : n;
    float s = 0.f;
    for (long long e = e0 + threadIdx.x; e < e1; e += 256) {
        const int x = (int)(e % W);
        const long long r = e / W;
        const int y = (int)(r % H), c = (int)(r / H);
        const float d = vp(a, b, c, y)[x] - vp(bb, b, c, y)[x];
        s += d * d;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[b * SQD_CHUNKS + chunk] = red[0];
}

__global__ void sqdiff_finish_kernel(const float *__restrict__ partial, float *__restrict__ out, int B, float inv_n) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int k = 0; k < SQD_CHUNKS; ++k) s += partial[b * SQD_CHUNKS + k];
    out[b] = s * inv_n;
}

// The L1 terms of the training loss (losses.py:113-170,196-233) of the planned step as per-sample SUMS in two deterministic launches
// (r5: the step's forward spent ~25 small strided torch kernels on them, back to back on an otherwise idle GPU):
//   out[b][0] = sum_{c,y,x} |pred - target|
//   out[b][1] = sum_{c,y,x} ( [stage 1 trains] (|g(I1,F01) - I0| + |g(I0,F10) - I1|) + [stage 2 trains] (|g(I0,Ft0) - I_t| + |g(I1,Ft1) - I_t|) )
// with the refined flows Ft1 = est[0:2] + out5[1:3], Ft0 = est[2:4] + out5[3:5] and the bilinear sampler of the adjoint kernels above.
__global__ __launch_bounds__(256) void loss_terms_partial_kernel(ssm_view img6, ssm_view flow4, ssm_view est, ssm_view out5, ssm_view pred,
                                                                 ssm_view target, float *__restrict__ partial, int H, int W, int s1terms,
                                                                 int s2terms) {
    __shared__ float red[2][256];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int n = H * W, per = (n + SQD_CHUNKS - 1) / SQD_CHUNKS;
    const int e0 = chunk * per, e1 = e0 + per < n ? e0 + per : n;
    float srec = 0.f, swrp = 0.f;
    for (int e = e0 + (int)threadIdx.x; e < e1; e += 256) {
        const int y = e / W, x = e - y * W;
        float tg[3], i0[3], i1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            tg[c] = vp(target, b, c, y)[x];
            srec += fabsf(vp(pred, b, c, y)[x] - tg[c]);
            i0[c] = vp(img6, b, c, y)[x];
            i1[c] = vp(img6, b, 3 + c, y)[x];
        }
        if (s2terms) {
            const float ft1u = vp(est, b, 0, y)[x] + vp(out5, b, 1, y)[x], ft1v = vp(est, b, 1, y)[x] + vp(out5, b, 2, y)[x];
            const float ft0u = vp(est, b, 2, y)[x] + vp(out5, b, 3, y)[x], ft0v = vp(est, b, 3, y)[x] + vp(out5, b, 4, y)[x];
            const TapsD t0 = make_taps_d(x, y, ft0u, ft0v, H, W, img6.sh), t1 = make_taps_d(x, y, ft1u, ft1v, H, W, img6.sh);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float w0, w1, gx, gy;
                sample_d(vp(img6, b, c, 0), t0, w0, gx, gy);
                sample_d(vp(img6, b, 3 + c, 0), t1, w1, gx, gy);
                swrp += fabsf(w0 - tg[c]) + fabsf(w1 - tg[c]);
            }
        }
        if (s1terms) {
            const TapsD a = make_taps_d(x, y, vp(flow4, b, 0, y)[x], vp(flow4, b, 1, y)[x], H, W, img6.sh);
            const TapsD bq = make_taps_d(x, y, vp(flow4, b, 2, y)[x], vp(flow4, b, 3, y)[x], H, W, img6.sh);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float va, vb, gx, gy;
                sample_d(vp(img6, b, 3 + c, 0), a, va, gx, gy);           // g(I1, F01) vs I0
                sample_d(vp(img6, b, c, 0), bq, vb, gx, gy);              // g(I0, F10) vs I1
                swrp += fabsf(va - i0[c]) + fabsf(vb - i1[c]);
            }
        }
    }
    red[0][threadIdx.x] = srec;
    red[1][threadIdx.x] = swrp;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[(b * SQD_CHUNKS + chunk) * 2] = red[0][0];
        partial[(b * SQD_CHUNKS + chunk) * 2 + 1] = red[1][0];
    }
}

__global__ void loss_terms_finish_kernel(const float *__restrict__ partial, float *__restrict__ out, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (sample, term)
    if (i >= 2 * B) return;
    const int b = i >> 1, term = i & 1;
    float s = 0.f;
    for (int k = 0; k < SQD_CHUNKS; ++k) s += partial[(b * SQD_CHUNKS + k) * 2 + term];
    out[b * 2 + term] = s;
}

}  // namespace

#define SSM_CHECK_DIMS(name)                                                                         \
    SSM_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535 && (H + 3) / 4 <= 65535, name ": bad sizes B=%d H=%d W=%d", B, H, W)

extern "C" int ssm_lrelu_bwd(ssm_view dy, ssm_view dpool, ssm_view y, ssm_view dz, int B, int C, int H, int W, float slope,
                             int has_act, void *stream) {
    SSM_CHECK_DIMS("lrelu_bwd");
    SSM_REQUIRE(dz.ptr && (dy.ptr || dpool.ptr) && (!has_act || y.ptr) && C > 0, "lrelu_bwd: null pointer / C");
    auto al16 = [](const ssm_view &v) { return !v.ptr || (ssm::aligned16(v.ptr) && v.sh % 4 == 0 && v.sc % 4 == 0 && v.sb % 4 == 0); };
    auto al8 = [](const ssm_view &v) { return !v.ptr || ((reinterpret_cast<size_t>(v.ptr) & 7) == 0 && v.sh % 2 == 0 && v.sc % 2 == 0 && v.sb % 2 == 0); };
    const bool vec = W % 4 == 0 && al16(dy) && al16(y) && al16(dz) && al8(dpool);
    const long long total = (long long)B * C * H * (vec ? W / 4 : W);
    SSM_REQUIRE(total <= 0x7fffffffLL * 256LL, "lrelu_bwd: problem too large for one launch");
    const dim3 grid((unsigned)((total + 255) / 256));
    if (vec) SSM_LAUNCH(lrelu_bwd_flat_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, dy, dpool, y, dz, C, H, W, slope, has_act, total);
    else SSM_LAUNCH(lrelu_bwd_flat_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, dy, dpool, y, dz, C, H, W, slope, has_act, total);
    return ssm::check_launch("ssm_lrelu_bwd");
}

extern "C" int ssm_lrelu_bwd_q8(ssm_view dy, ssm_view dpool, ssm_view y, ssm_view dz, ssm_hview dz_q8, int B, int C, int H, int W, float slope,
                                int has_act, void *stream) {
    SSM_CHECK_DIMS("lrelu_bwd_q8");
    SSM_REQUIRE(dz.ptr && dz_q8.ptr && (dy.ptr || dpool.ptr) && (!has_act || y.ptr) && C > 0, "lrelu_bwd_q8: null pointer / C");
    SSM_REQUIRE(ssm::aligned16(dz_q8.ptr), "lrelu_bwd_q8: the Q8 view must be 16-byte aligned");
    const int cgroups = (C + BWD_CPT - 1) / BWD_CPT;
    SSM_REQUIRE((long long)B * cgroups <= 65535, "lrelu_bwd_q8: B*C too large for one launch");
    SSM_LAUNCH(lrelu_bwd_q8_kernel, dim3((W + 63) / 64, (H + 3) / 4, B * cgroups), dim3(64, 4), 0, (hipStream_t)stream, dy, dpool, y,
                       dz, dz_q8, C, H, W, slope, has_act, cgroups);
    return ssm::check_launch("ssm_lrelu_bwd_q8");
}

static int bias_grad_launch(ssm_view dz, float *db, int B, int C, int H, int W, int zero_first, void *stream) {
    SSM_REQUIRE(dz.ptr && db && B > 0 && C > 0 && H > 0 && W > 0, "bias_grad: bad arguments");
    if (zero_first) {
        hipError_t e = ssm::memset_async(db, 0, sizeof(float) * (size_t)C, (hipStream_t)stream);
        if (e != hipSuccess) {
            ssm::set_error("bias_grad: memset failed: %s", hipGetErrorString(e));
            return SSM_E_LAUNCH;
        }
    }
    int chunks = (B * H + 63) / 64;          // ~16 rows per wave
    if (chunks < 1) chunks = 1;
    if (chunks > 1024) chunks = 1024;
    SSM_LAUNCH(bias_grad_kernel, dim3(C, chunks), dim3(256), 0, (hipStream_t)stream, dz, db, B, H, W);
    return ssm::check_launch("ssm_bias_grad");
}

extern "C" int ssm_bias_grad(ssm_view dz, float *db, int B, int C, int H, int W, void *stream) {
    return bias_grad_launch(dz, db, B, C, H, W, 1, stream);
}

extern "C" int ssm_bias_grad_acc(ssm_view dz, float *db, int B, int C, int H, int W, void *stream) {
    return bias_grad_launch(dz, db, B, C, H, W, 0, stream);
}

static int wgrad_launch(ssm_view x, ssm_view dz, float *dw_oihw, float *db, int B, int Cin, int Cout, int H, int W, int k, int cin_total,
                        int ci_offset, int zero_first, void *stream) {
    SSM_REQUIRE(x.ptr && dz.ptr && dw_oihw && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "wgrad: bad arguments");
    SSM_REQUIRE(ci_offset >= 0 && ci_offset + Cin <= cin_total, "wgrad: channel range [%d,%d) outside the filter's %d inputs", ci_offset,
                ci_offset + Cin, cin_total);
    SSM_REQUIRE(x.sh >= W + 2 * SSM_PADX && dz.sh >= W + 3, "wgrad: x and dz must be padded-plane views (zero frame)");
    SSM_REQUIRE(ssm::aligned16(x.ptr) && ssm::aligned16(dz.ptr) && x.sh % 4 == 0 && x.sc % 4 == 0 && x.sb % 4 == 0 && dz.sh % 4 == 0 &&
                    dz.sc % 4 == 0 && dz.sb % 4 == 0,
                "wgrad: views must be 16-byte aligned with strides that are multiples of 4 floats");
    hipStream_t st = (hipStream_t)stream;
    if (zero_first) {
        hipError_t e = ssm::memset_async(dw_oihw, 0, sizeof(float) * (size_t)Cout * cin_total * k * k, st);
        if (e != hipSuccess) {
            ssm::set_error("wgrad: memset failed: %s", hipGetErrorString(e));
            return SSM_E_LAUNCH;
        }
    }
    const int ntc = Cout > 32 ? 2 : 1;
    const int gx = (Cin * k * k + (db ? 1 : 0) + 255) / 256, gy = (Cout + ntc * 32 - 1) / (ntc * 32);
    const int tiles = gx * gy;
    const int seg = W > 32 ? 64 : (W > 16 ? 32 : 16), rr = 128 / seg;       // a step = rr image rows x seg pixels
    // Workgroups of one launch: at most `target` = one round of two co-resident workgroups per CU (measured per layer of the training
    // step, tools/bench_wgrad.py with $SSM_WGRAD_TARGET: 512 -> 4.8 ms per step, 768 -> 6.0, 1024 -> 5.5, 1536 -> 6.0; rounding the
    // split UP, as the first version did, put 1029 workgroups on 1024 slots and a third, almost empty round behind them - and every
    // extra slice multiplies the atomic adds)
    static const int target_env = [] {
        const char *e = getenv("SSM_WGRAD_TARGET");          // tuning knob
        return e ? atoi(e) : 0;
    }();
    const int target = target_env > 0 ? target_env : 512;
    int split = target / tiles;
    const int rows = B * ((H + rr - 1) / rr) * ((W + seg - 1) / seg);       // staging steps
    if (split > rows) split = rows;
    if (split < 1) split = 1;
    if (split > 65535) split = 65535;
    const dim3 grid(gx, gy, split);
#define SSM_WGRAD_L(KS_, XC_, NTC_, SEG_, RR_)                                                                                       \
    SSM_LAUNCH((wgrad_mfma_kernel<KS_, XC_, NTC_, SEG_, RR_>), grid, dim3(256), 0, st, x, dz, dw_oihw, db, B, Cin, Cout, H, W, \
                       cin_total, ci_offset)
#define SSM_WGRAD_S(KS_, XC_, NTC_)                              \
    if (seg == 64) SSM_WGRAD_L(KS_, XC_, NTC_, 64, 2);           \
    else if (seg == 32) SSM_WGRAD_L(KS_, XC_, NTC_, 32, 4);      \
    else SSM_WGRAD_L(KS_, XC_, NTC_, 16, 8)
#define SSM_WGRAD(KS_, XC_)                 \
    if (ntc == 2) { SSM_WGRAD_S(KS_, XC_, 2); } \
    else { SSM_WGRAD_S(KS_, XC_, 1); }
    switch (k) {          // XC = input channels a 256-column workgroup can touch: ceil((256 + k*k - 1) / (k*k))
        case 3: SSM_WGRAD(3, 30) break;
        case 5: SSM_WGRAD(5, 12) break;
        case 7: SSM_WGRAD(7, 7) break;
        default: ssm::set_error("wgrad: kernel size %d unsupported", k); return SSM_E_UNSUPPORTED;
    }
#undef SSM_WGRAD
#undef SSM_WGRAD_S
#undef SSM_WGRAD_L
    return ssm::check_launch("ssm_conv2d_wgrad");
}

extern "C" int ssm_conv2d_wgrad(ssm_view x, ssm_view dz, float *dw_oihw, int B, int Cin, int Cout, int H, int W, int k, int cin_total,
                                int ci_offset, int zero_first, void *stream) {
    return wgrad_launch(x, dz, dw_oihw, nullptr, B, Cin, Cout, H, W, k, cin_total, ci_offset, zero_first, stream);
}

extern "C" int ssm_conv2d_wgrad_bias(ssm_view x, ssm_view dz, float *dw_oihw, float *db_acc, int B, int Cin, int Cout, int H, int W, int k,
                                     int cin_total, int ci_offset, int zero_first, void *stream) {
    SSM_REQUIRE(db_acc, "wgrad_bias: null bias-gradient pointer");
    return wgrad_launch(x, dz, dw_oihw, db_acc, B, Cin, Cout, H, W, k, cin_total, ci_offset, zero_first, stream);
}

extern "C" int ssm_conv2d_wgrad_bf16x3(ssm_view x, ssm_view dz, float *dw_oihw, int B, int Cin, int Cout, int H, int W, int k, int cin_total,
                                       int ci_offset, int zero_first, void *stream) {
    SSM_REQUIRE(x.ptr && dz.ptr && dw_oihw && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "wgrad_bf16x3: bad arguments");
    SSM_REQUIRE(ci_offset >= 0 && ci_offset + Cin <= cin_total, "wgrad_bf16x3: channel range [%d,%d) outside the filter's %d inputs",
                ci_offset, ci_offset + Cin, cin_total);
    SSM_REQUIRE(x.sh >= W + 2 * SSM_PADX && dz.sh >= W + 3, "wgrad_bf16x3: x and dz must be padded-plane views (zero frame)");
    SSM_REQUIRE(ssm::aligned16(x.ptr) && ssm::aligned16(dz.ptr) && x.sh % 4 == 0 && x.sc % 4 == 0 && x.sb % 4 == 0 && dz.sh % 4 == 0 &&
                    dz.sc % 4 == 0 && dz.sb % 4 == 0,
                "wgrad_bf16x3: views must be 16-byte aligned with strides that are multiples of 4 floats");
    hipStream_t st = (hipStream_t)stream;
    if (zero_first) {
        hipError_t e = ssm::memset_async(dw_oihw, 0, sizeof(float) * (size_t)Cout * cin_total * k * k, st);
        if (e != hipSuccess) {
            ssm::set_error("wgrad_bf16x3: memset failed: %s", hipGetErrorString(e));
            return SSM_E_LAUNCH;
        }
    }
    const int ca = (Cout + 31) / 32, cb = (Cin + 31) / 32;
    // split of the image rows over gridDim.z: about two workgroups per CU, each walking >= 16 rows (shorter slices pay the ring
    // warm-up and the atomic epilogue more often: 2048 workgroups x 8 rows measured 40 % slower over the step's layers)
    const int tgt_wgs = 512, min_steps = 16;
    // tile configuration: (filter size, filter rows per workgroup, segment, cout tiles, cin tiles of the 4 waves)
#define SSM_WGRAD16(KS_, TY_, SEG_, WAN_, WBN_)                                                                                   \
    {                                                                                                                             \
        const int gx = (cb + WBN_ - 1) / WBN_, gy = ((ca + WAN_ - 1) / WAN_) * (KS_ / TY_);                                       \
        const int total = B * ((W + SEG_ - 1) / SEG_) * H;                                                                        \
        int nsl = (tgt_wgs + gx * gy - 1) / (gx * gy);                                                                            \
        if (nsl > (total + min_steps - 1) / min_steps) nsl = (total + min_steps - 1) / min_steps;                                 \
        if (nsl < 1) nsl = 1;                                                                                                     \
        if (nsl > 65535) nsl = 65535;                                                                                             \
        const int sps = (total + nsl - 1) / nsl;                                                                                  \
        nsl = (total + sps - 1) / sps;                                                                                            \
        SSM_LAUNCH((wgrad_bf16x3_kernel<KS_, TY_, SEG_, WAN_, WBN_>), dim3(gx, gy, nsl), dim3(256), 0, st, x, dz, dw_oihw, B, \
                           Cin, Cout, H, W, cin_total, ci_offset, sps);                                                           \
    }
    switch (k) {
        case 3:
            if (ca >= 2 && cb >= 2) SSM_WGRAD16(3, 3, 32, 2, 2)
            else if (cb >= 2) SSM_WGRAD16(3, 3, 64, 1, 2)
            else if (ca >= 2) SSM_WGRAD16(3, 3, 64, 2, 1)
            else SSM_WGRAD16(3, 3, 64, 1, 1)
            break;
        case 5:
            if (ca >= 2 && cb >= 2) SSM_WGRAD16(5, 1, 64, 2, 2)
            else if (ca >= 2) SSM_WGRAD16(5, 1, 128, 2, 1)
            else SSM_WGRAD16(5, 1, 128, 1, 1)
            break;
        case 7: SSM_WGRAD16(7, 1, 128, 1, 1) break;      // the reference's 7x7 layers have 32 outputs and <= 32 inputs (one tile per filter row)
        default: ssm::set_error("wgrad_bf16x3: kernel size %d unsupported", k); return SSM_E_UNSUPPORTED;
    }
#undef SSM_WGRAD16
    return ssm::check_launch("ssm_conv2d_wgrad_bf16x3");
}

static int upsample_cat_bwd_launch(ssm_view du, ssm_view da, int Ca, ssm_view db, int Cb, ssm_view ya, float msl, int B, int H, int W, int acc_a,
                                   int acc_b, void *stream) {
    SSM_CHECK_DIMS("upsample2x_cat_bwd");
    SSM_REQUIRE(du.ptr && da.ptr && Ca > 0 && Cb >= 0 && (Cb == 0 || db.ptr), "upsample2x_cat_bwd: null pointer / channels");
    auto al16 = [](const ssm_view &v) { return ssm::aligned16(v.ptr) && v.sh % 4 == 0 && v.sc % 4 == 0 && v.sb % 4 == 0; };
    auto al8 = [](const ssm_view &v) { return (reinterpret_cast<size_t>(v.ptr) & 7) == 0 && v.sh % 2 == 0 && v.sc % 2 == 0 && v.sb % 2 == 0; };
    if (W % 2 == 0 && al16(du) && al8(da) && (Cb == 0 || al8(db)) && (!ya.ptr || al8(ya))) {
        const long long total = (long long)B * (Ca + Cb) * H * (W / 2);
        SSM_REQUIRE(total <= 0x7fffffffLL * 256LL, "upsample2x_cat_bwd: problem too large for one launch");
        SSM_LAUNCH(upsample_cat_bwd2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, du, da, Ca,
                           Cb ? db : da, Cb, H, W, acc_a, acc_b, total, ya, msl);
        return ssm::check_launch("ssm_upsample2x_cat_bwd");
    }
    const int cgroups = (Ca + Cb + BWD_CPT - 1) / BWD_CPT;
    SSM_REQUIRE((long long)B * cgroups <= 65535, "upsample2x_cat_bwd: B*C too large for one launch");
    SSM_LAUNCH(upsample_cat_bwd_kernel, dim3((W + 63) / 64, (H + 3) / 4, B * cgroups), dim3(64, 4), 0, (hipStream_t)stream, du, da,
                       Ca, Cb ? db : da, Cb, H, W, acc_a, acc_b, cgroups, ya, msl);
    return ssm::check_launch("ssm_upsample2x_cat_bwd");
}

extern "C" int ssm_upsample2x_cat_bwd(ssm_view du, ssm_view da, int Ca, ssm_view db, int Cb, int B, int H, int W, int acc_a, int acc_b,
                                      void *stream) {
    return upsample_cat_bwd_launch(du, da, Ca, db, Cb, ssm_view{nullptr, 0, 0, 0}, 1.f, B, H, W, acc_a, acc_b, stream);
}

extern "C" int ssm_upsample2x_cat_bwd_mask(ssm_view du, ssm_view da, int Ca, ssm_view db, int Cb, ssm_view ya, float slope, int B, int H, int W,
                                           int acc_a, int acc_b, void *stream) {
    SSM_REQUIRE(ya.ptr, "upsample2x_cat_bwd_mask: null mask source");
    return upsample_cat_bwd_launch(du, da, Ca, db, Cb, ya, slope, B, H, W, acc_a, acc_b, stream);
}

extern "C" int ssm_synthesize_bwd(ssm_view img6, ssm_view est4, ssm_view out5, ssm_view target, const float *t, const float *c_rec,
                                  const float *c_warp, ssm_view dy_extra, ssm_view dout5, ssm_view dest4, int B, int H, int W,
                                  int stage2_terms, void *stream) {
    SSM_CHECK_DIMS("synthesize_bwd");
    SSM_REQUIRE(img6.ptr && est4.ptr && out5.ptr && target.ptr && t && c_rec && c_warp && dout5.ptr && dest4.ptr, "synthesize_bwd: null pointer");
    SSM_LAUNCH(synth_bwd_kernel, pix_grid(B, H, W), dim3(64, 4), 0, (hipStream_t)stream, img6, est4, out5, target, t, c_rec,
                       c_warp, dy_extra, dout5, dest4, H, W, stage2_terms);
    return ssm::check_launch("ssm_synthesize_bwd");
}

extern "C" int ssm_flowinterp_inputs_bwd(ssm_view img6, ssm_view flow4, ssm_view din16, ssm_view dest4, const float *t, const float *c_warp,
                                         ssm_view dflow4, int B, int H, int W, int stage1_terms, void *stream) {
    SSM_CHECK_DIMS("flowinterp_inputs_bwd");
    SSM_REQUIRE(img6.ptr && flow4.ptr && din16.ptr && dest4.ptr && t && c_warp && dflow4.ptr, "flowinterp_inputs_bwd: null pointer");
    SSM_LAUNCH(inputs_bwd_kernel, pix_grid(B, H, W), dim3(64, 4), 0, (hipStream_t)stream, img6, flow4, din16, dest4, t, c_warp,
                       dflow4, H, W, stage1_terms);
    return ssm::check_launch("ssm_flowinterp_inputs_bwd");
}

extern "C" int ssm_maxpool2_fwd(ssm_view x, ssm_view y, int B, int C, int H, int W, void *stream) {
    SSM_CHECK_DIMS("maxpool2");
    SSM_REQUIRE(x.ptr && y.ptr && C > 0 && H % 2 == 0 && W % 2 == 0, "maxpool2: null pointer / odd size");
    const int cgroups = (C + BWD_CPT - 1) / BWD_CPT;
    SSM_REQUIRE((long long)B * cgroups <= 65535, "maxpool2: B*C too large for one launch");
    SSM_LAUNCH(maxpool2_kernel, pix_grid(B * cgroups, H / 2, W / 2), dim3(64, 4), 0, (hipStream_t)stream, x, y, C, H / 2, W / 2, cgroups);
    return ssm::check_launch("ssm_maxpool2_fwd");
}

extern "C" int ssm_maxpool2_bwd(ssm_view x, ssm_view dy, ssm_view dx, int B, int C, int H, int W, void *stream) {
    SSM_CHECK_DIMS("maxpool2_bwd");
    SSM_REQUIRE(x.ptr && dy.ptr && dx.ptr && C > 0 && H % 2 == 0 && W % 2 == 0, "maxpool2_bwd: null pointer / odd size");
    const int cgroups = (C + BWD_CPT - 1) / BWD_CPT;
    SSM_REQUIRE((long long)B * cgroups <= 65535, "maxpool2_bwd: B*C too large for one launch");
    SSM_LAUNCH(maxpool2_bwd_kernel, pix_grid(B * cgroups, H / 2, W / 2), dim3(64, 4), 0, (hipStream_t)stream, x, dy, dx, C, H / 2, W / 2,
                       cgroups);
    return ssm::check_launch("ssm_maxpool2_bwd");
}

extern "C" int ssm_sqdiff_grad(ssm_view a, ssm_view b, const float *coef, ssm_view out, int B, int C, int H, int W, void *stream) {
    SSM_CHECK_DIMS("sqdiff_grad");
    SSM_REQUIRE(a.ptr && b.ptr && coef && out.ptr && C > 0, "sqdiff_grad: null pointer");
    const int cgroups = (C + BWD_CPT - 1) / BWD_CPT;
    SSM_REQUIRE((long long)B * cgroups <= 65535, "sqdiff_grad: B*C too large for one launch");
    SSM_LAUNCH(sqdiff_grad_kernel, pix_grid(B * cgroups, H, W), dim3(64, 4), 0, (hipStream_t)stream, a, b, coef, out, C, H, W, cgroups);
    return ssm::check_launch("ssm_sqdiff_grad");
}

extern "C" int ssm_train_loss_sums(ssm_view img6, ssm_view flow4, ssm_view est4, ssm_view out5, ssm_view pred, ssm_view target, float *scratch,
                                   float *out, int B, int H, int W, int stage1_terms, int stage2_terms, void *stream) {
    SSM_CHECK_DIMS("train_loss_sums");
    SSM_REQUIRE(img6.ptr && flow4.ptr && est4.ptr && out5.ptr && pred.ptr && target.ptr && scratch && out, "train_loss_sums: null pointer");
    SSM_REQUIRE((long long)H * img6.sh < 0x7fffffffLL, "train_loss_sums: plane too large");
    SSM_LAUNCH(loss_terms_partial_kernel, dim3(SQD_CHUNKS, B), dim3(256), 0, (hipStream_t)stream, img6, flow4, est4, out5, pred, target,
                       scratch, H, W, stage1_terms, stage2_terms);
    SSM_LAUNCH(loss_terms_finish_kernel, dim3((2 * B + 63) / 64), dim3(64), 0, (hipStream_t)stream, scratch, out, B);
    return ssm::check_launch("ssm_train_loss_sums");
}

extern "C" int ssm_sqdiff_mean(ssm_view a, ssm_view b, float *scratch, float *out, int B, int C, int H, int W, void *stream) {
    SSM_CHECK_DIMS("sqdiff_mean");
    SSM_REQUIRE(a.ptr && b.ptr && scratch && out && C > 0, "sqdiff_mean: null pointer");
    SSM_LAUNCH(sqdiff_partial_kernel, dim3(SQD_CHUNKS, B), dim3(256), 0, (hipStream_t)stream, a, b, scratch, C, H, W);
    SSM_LAUNCH(sqdiff_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, scratch, out, B,
                       1.0f / ((float)C * (float)H * (float)W));
    return ssm::check_launch("ssm_sqdiff_mean");
}

extern "C" int ssm_warp_bilinear_bwd(ssm_view img, ssm_view flow, ssm_view dy, ssm_view dflow, ssm_view dimg, int B, int C, int H, int W,
                                     void *stream) {
    SSM_CHECK_DIMS("warp_bwd");
    SSM_REQUIRE(img.ptr && flow.ptr && dy.ptr && C > 0 && (dflow.ptr || dimg.ptr), "warp_bwd: null pointer / C");
    SSM_REQUIRE(!dimg.ptr || dimg.sh == img.sh, "warp_bwd: dimg must have the image's row stride");
    SSM_REQUIRE((long long)H * img.sh < 0x7fffffffLL, "warp_bwd: plane too large");
    SSM_LAUNCH(warp_bwd_kernel, pix_grid(B, H, W), dim3(64, 4), 0, (hipStream_t)stream, img, flow, dy, dflow, dimg, C, H, W);
    return ssm::check_launch("ssm_warp_bilinear_bwd");
}
