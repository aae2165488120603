// fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces layers.conv / final_conv of the reference (scripts/models/layers.py:21-33,
// scripts/models/flow_computation.py:145-153): stride-1 'same' cross-correlation,
// zero padding, bias, optional LeakyReLU, optional fused 2x2 average pool
// (scripts/models/layers.py:60-63), optional two-source input (torch.cat on C) and - the
// decoder step of scripts/models/flow_computation.py:244-247 - optional fused
// F.upsample(torch.cat([a, b], 1), bilinear x2) in front of the convolution.
//
// GEMM view, per batch element:  D[cout][pixel] = sum_k W[cout][k] * X[k][pixel],
// k = (cin, ky, kx).  The MFMA A operand is the filter (32 couts x 2 k), the B
// operand is the activation (2 k x 32 pixels), so the accumulator has the pixel
// on the lane.  A 32-pixel group is 32x1 (consecutive x of one row: every store
// instruction writes whole 128-byte row segments of NCHW planes) or, for maps
// whose width is not near a multiple of 32 (80, 40, 22, 11 ...), 8x4.
//
// Data movement: the input lives in the padded-plane layout (include/ssm_hip.h),
// so a tile's halo is just a bigger rectangle - no bounds tests.  Per chunk of CK
// input channels a workgroup stages  [CK][k*k][BN] filter taps  and a
// [CK][TH+k-1][TW+8] activation patch into LDS with LDS-DMA (global_load_lds
// dwordx4: no VGPR round trip), double-buffered: chunk c+1 is in flight while
// chunk c feeds the matrix cores; one barrier per chunk.  One MFMA k-step takes
// its two k from two consecutive input channels at the same tap (lanes 0-31 /
// 32-63), so each operand fetch is one conflict-free ds_read_b32 at a
// compile-time offset from a per-lane base.
//
// Fused upsample (UPS): the DMA brings the LOW-res [CK][TH/2+2][TW/2+8] patch of
// the chunk instead; the workgroup expands it in LDS to the hi-res patch the MFMA
// loop reads (ATen's half-pixel rule with edge-clamped source indices, exact zeros
// outside the image = the convolution's zero padding).  fp32 MFMA is 16x slower
// than the fp16 matrix path, so the ~10 VALU operations per patch element vanish
// beside the chunk's MFMAs (the other workgroup of the CU keeps the pipe busy
// meanwhile), and the concatenated, upsampled tensor - the largest activation of
// every decoder level - never touches HBM.
#include "ssm_common.h"

#include <atomic>
#include <mutex>
#include <type_traits>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct ConvParams {
    const float *src1;
    const float *src2;
    long long sb1, sb2;  // batch strides
    long long sc;        // channel stride (both sources)
    int sh;              // row stride (both sources)
    int C1, Cin;         // channels of source 1, total
    const float *wpk;
    const float *bias;
    float *dst;
    long long dsb, dsc;
    int dsh;
    float *pool;
    long long psb, psc;
    int psh;
    int H, W, Cout;      // OUTPUT map
    int hs, ws;          // UPS: source map (H/2, W/2)
    int tilesX, tilesY, NB;
    float slope;
    int lrelu;
    int abl;             // diagnostics build only (make ablate): 1 = no LDS-DMA after the second chunk, 2 = no stores
    // optional pre-activation addend [B / adiv][Cout][H][W] (batch entry b reads entry b / adiv): the part of the sum that does
    // not depend on the batch index - e.g. the image channels of stage 2's first convolution, equal for the 7 t of a pair
    const float *add;
    long long asb, asc;
    int ash, adiv;
    // split-K launches (ssm_conv2d_splitk_fwd): NSPLIT workgroups share an output tile, workgroup ks sums input channels
    // [ks, ks + 1) * Cin / NSPLIT and stores its raw sums (the bias with ks = 0; no activation) as batch entry ks * ksB + b of dst
    int NSPLIT, ksB;
};

template <int KS_, int NT_, int WN_, int MTY_, int MTX_, int WY_, int WX_, int CK_, int GW_ = 32>
struct Cfg {
    static constexpr int KS = KS_, NT = NT_, WN = WN_, MTY = MTY_, MTX = MTX_, WY = WY_, WX = WX_, CK = CK_;
    static constexpr int GW = GW_, GH = 32 / GW_;   // a 32-pixel group = GH rows x GW columns
    static constexpr int KS2 = KS * KS, PAD = (KS - 1) / 2;
    static constexpr int BN = 32 * NT * WN;       // output channels per workgroup
    static constexpr int TH = MTY * GH * WY;      // output rows per workgroup
    static constexpr int TW = MTX * GW * WX;      // output columns per workgroup
    static constexpr int MT = MTY * MTX;          // 32-pixel groups per wave
    static constexpr int PH = TH + KS - 1;        // patch rows
    static constexpr int PW = TW + 8;             // patch columns (16-byte aligned both ends)
    static constexpr int PW4 = PW / 4;
    static constexpr int WSZ = CK * KS2 * BN;     // filter floats per chunk
    static constexpr int PSZ = CK * PH * PW;      // patch floats per chunk
    // fused upsample: low-res raw patch rows y0/2-1 .. (y0+TH)/2, columns x0/2-4 .. x0/2+TW/2+3
    static constexpr int LH = TH / 2 + 2, LW = TW / 2 + 8, LW4 = LW / 4;
    static constexpr int RSZ = CK * LH * LW;
    // fused-upsample expander: positions of 2x2 hi-res blocks per tile, channel groups over the 256 threads, channels per thread
    static constexpr int NPOS = (TH / 2 + 1) * (TW / 2 + 1);
    static constexpr int CG = (2 * NPOS <= 256 && CK >= 2) ? ((4 * NPOS <= 256 && CK >= 4) ? ((8 * NPOS <= 256 && CK >= 8) ? 8 : 4) : 2) : 1;
    static constexpr int CPT = CK / CG;
    static constexpr bool POOL_OK = (GW == 32) ? (MTY % 2 == 0) : true;
    static constexpr bool UPS_OK = (KS == 3) && (TH % 2 == 0);
    static_assert(GW == 32 || GW == 8, "pixel group is 32x1 or 8x4");
    static_assert(WN * WY * WX == 4, "4 waves per workgroup");
    static_assert(CK % 2 == 0, "one MFMA k-step = two input channels");
    static_assert(TW % 16 == 0, "patch rows must keep the 8x4 groups on distinct LDS banks");
};

// LDS plan of one kernel variant: two DMA stages [filter | patch or raw patch] (+ the expanded patch)
template <class C, bool UPS>
struct Lds {
    static constexpr int DSZ = UPS ? C::RSZ : C::PSZ;            // DMA'd activation floats per chunk
    static constexpr int DH = UPS ? C::LH : C::PH, DW4 = UPS ? C::LW4 : C::PW4;
    static constexpr int NWQ = C::WSZ / 4, NDQ = DSZ / 4, NQ = NWQ + NDQ;  // 16-byte pieces
    static constexpr int NG = (NQ + 63) / 64;     // 1-KiB wave-instructions per chunk
    static constexpr int STAGE = NG * 256;        // floats per LDS stage
    static constexpr int NI = (NG + 3) / 4;       // LDS-DMA instructions per wave per chunk
    static constexpr int HIP = 2 * STAGE;         // offset of the expanded patch (UPS)
    static constexpr int BYTES = (2 * STAGE + (UPS ? C::PSZ : 0)) * 4;
    static_assert(BYTES <= 65536, "LDS budget (two workgroups per CU)");
};

#define SSM_GLDS16(gp, lp)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp),      \
                                     (__attribute__((address_space(3))) void *)(lp), 16, 0, 0)

// SPLIT: the split-K instantiation (ssm_conv2d_splitk_fwd; compiled for the tile configurations of CONV_SPLIT_OK only - the large-tile
// configurations sit at the register cap, and the split's three extra scalars tip them into scratch)
template <class C, bool UPS, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using L = Lds<C, UPS>;
    constexpr int KS = C::KS, KS2 = C::KS2, BN = C::BN, PH = C::PH, PW = C::PW, NT = C::NT, MT = C::MT;
    constexpr int GW = C::GW, GH = C::GH;

    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int gy = l31 / GW, gx = l31 % GW;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid % C::WN, wy = (wid / C::WN) % C::WY, wx = wid / (C::WN * C::WY);

    int id = ssm_xcd_tile(blockIdx.x, gridDim.x);
    const int ks = SPLIT ? id % p.NSPLIT : 0;          // (split-K partners are neighbours in the grid: they read the same patches)
    if (SPLIT) id /= p.NSPLIT;
    const int nb = id % p.NB;
    id /= p.NB;
    const int tx = id % p.tilesX;
    id /= p.tilesX;
    const int ty = id % p.tilesY;
    const int b = id / p.tilesY;
    const int x0 = tx * C::TW, y0 = ty * C::TH;
    const int nchunks = SPLIT ? p.Cin / C::CK / p.NSPLIT : p.Cin / C::CK;          // (split launches are single-source: the offsets below cover them)

    // DMA origin: element (c, y0-PAD, x0-4) of the padded planes, or (c, y0/2-1, x0/2-4) of the low-res source
    const long long porg = UPS ? (long long)(y0 / 2 - 1) * p.sh + (x0 / 2 - 4) : (long long)(y0 - C::PAD) * p.sh + (x0 - 4);
    const long long cbeg = SPLIT ? (long long)ks * nchunks * C::CK : 0;          // first input channel of this workgroup
    const float *pbase1 = p.src1 + (long long)b * p.sb1 + porg + cbeg * p.sc;
    const float *pbase2 = p.src2 + (long long)b * p.sb2 + porg;
    const float *wbase = p.wpk + ((long long)nb * p.Cin + cbeg) * (KS2 * BN);

    // per-lane source offset of each LDS-DMA piece this wave issues (same for every chunk)
    int off[L::NI];
    bool isw[L::NI];
#pragma unroll
    for (int i = 0; i < L::NI; ++i) {
        const int q = (i * 4 + wid) * 64 + lane;
        if (q < L::NWQ) {
            isw[i] = true;
            off[i] = q * 4;
        } else if (q < L::NQ) {
            const int qq = q - L::NWQ;
            const int c = qq / (L::DH * L::DW4);
            const int rem = qq - c * (L::DH * L::DW4);
            const int r = rem / L::DW4;
            const int j = rem - r * L::DW4;
            isw[i] = false;
            off[i] = (int)(c * p.sc) + r * p.sh + 4 * j;
        } else {  // tail of the last 1-KiB piece: lands in the stage's padding
            isw[i] = true;
            off[i] = 0;
        }
    }

    // LDS-DMA of one chunk.  A 1-KiB wave-instruction (group g) whose 64 pieces all come from the filter, or all from the
    // activation patch, uses the saddr form - wave-uniform 64-bit base in SGPRs + the per-lane 32-bit byte offset computed once -
    // and costs no vector instruction; only the one group that straddles the filter | patch boundary selects its base per lane.
    // (The builtin's 64-bit per-lane address cost 7 VALU instructions per DMA, every chunk.)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
    // issue_k(ch, stage, k): this wave's k-th DMA instruction of chunk `ch` (group g = 4k + wave), k < NI.
    auto issue_k = [&](int ch, int stage, int k) {
        const int c0 = ch * C::CK;
        const float *pb = (c0 < p.C1) ? pbase1 + (long long)c0 * p.sc : pbase2 + (long long)(c0 - p.C1) * p.sc;
        const float *wb = wbase + (long long)c0 * (KS2 * BN);
        float *ls = lds + stage * L::STAGE;
        const unsigned lsb = lds0 + (unsigned)(stage * L::STAGE) * 4u;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int g = 4 * k + w;
            if (w == wid && g < L::NG) {          // wave-uniform
                const bool pure_w = g * 64 + 63 < L::NWQ, pure_p = g * 64 >= L::NWQ;
#ifndef SSM_DMA_SADDR
#define SSM_DMA_SADDR 1
#endif
                if (SSM_DMA_SADDR && (pure_w || pure_p)) {
                    const float *base = pure_w ? wb : pb;
                    const unsigned m0v = lsb + (unsigned)g * 1024u;
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                                 :: "v"(off[k] * 4), "s"(base), "s"(m0v) : "memory", "m0");
                } else {
                    const float *gp = (isw[k] ? wb : pb) + off[k];
                    SSM_GLDS16(gp, ls + g * 256);
                }
            }
        }
    };
    auto issue = [&](int ch, int stage) {
#pragma unroll
        for (int k = 0; k < L::NI; ++k) issue_k(ch, stage, k);
    };

    // (Bias: added AFTER the k-loop, see the epilogue.)
    f32x16 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;
    if (p.add) {
        // The accumulators start from the addend (register r of lane (l31, half) = cout (r&3) + 8*(r>>2) + 4*half, pixel l31 of the
        // group): the loads fly while the first chunk's DMA is in flight.  Addressing as in the store loop - a wave-uniform base
        // per cout (SGPRs) + ONE 32-bit per-lane byte offset per pixel group - so the 16 x NT x MT loads need no registers beyond
        // the accumulators (per-load 64-bit addresses cost 30-40 VGPRs = one resident workgroup per CU).  No predicates: Cout is a
        // multiple of 32 (checked on the host), pixels beyond the map read the addend's frame / slack and are never stored.
        const float *ab = p.add + (long long)(b / p.adiv) * p.asb;
        const int ax = x0 + wx * (C::MTX * GW) + gx, ay = y0 + wy * (C::MTY * GH) + gy;
        unsigned aoff[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
            aoff[m] = 4u * ((unsigned)(4 * half) * (unsigned)p.asc + (unsigned)(ay + (m / C::MTX) * GH) * (unsigned)p.ash + (unsigned)(ax + (m % C::MTX) * GW));
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cu = nb * BN + (wn * NT + n) * 32 + (r & 3) + 8 * (r >> 2);          // uniform
                const char *bp = (const char *)(ab + (long long)cu * p.asc);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[n][m][r] = *(const float *)(bp + aoff[m]);
            }
    }
    // per-lane operand bases (floats): filter inside a stage, activation inside the patch
    const int aBase = half * (KS2 * BN) + wn * (NT * 32) + l31;
    const int bBase = half * (PH * PW) + (wy * C::MTY * GH + gy) * PW + wx * (C::MTX * GW) + gx + (4 - C::PAD);

    issue(0, 0);
    for (int ch = 0; ch < nchunks; ++ch) {
        // chunk ch has landed for every wave; every wave is done reading chunk ch-1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // the DMA of chunk ch+1 is issued from inside the MFMA loop below, one instruction per macro-step (SSM_DMA_SPREAD): the
        // requests do not hit the memory system as one burst behind the barrier, and the first MFMA does not wait for their issue
#ifndef SSM_DMA_SPREAD
#define SSM_DMA_SPREAD 1
#endif
#ifdef SSM_CONV_ABLATE
        const bool dma_next = ch + 1 < nchunks && !((p.abl & 1) && ch >= 1);
#else
        const bool dma_next = ch + 1 < nchunks;
#endif
        if (!SSM_DMA_SPREAD && dma_next) issue(ch + 1, (ch + 1) & 1);

        const float *stg = lds + (ch & 1) * L::STAGE;
        if constexpr (UPS) {
            // expand the low-res chunk: one work unit = the 2x2 hi-res block between low-res pixels (i,j)..(i+1,j+1).  A thread
            // owns ONE position (pi, pj) of the tile - its clamps, weights, in-image flags and LDS offsets are computed once per
            // chunk - and walks CPT of the chunk's channels with constant strides (the first version did the index arithmetic,
            // two divisions included, per unit: 3x the vector instructions on the 8x64 tiles).
            constexpr int PRW = C::TW / 2 + 1, NPOS = C::NPOS, CG = C::CG, CPT = C::CPT;
            constexpr int LH = C::LH, LW = C::LW;
            static_assert(NPOS <= 256 && C::CK % CG == 0, "expander mapping");
            const float *raw = stg + C::WSZ;
            float *hip = lds + L::HIP;
            int tl = tid;
            // big tiles (several channels per thread) recompute the geometry per chunk: hoisted out of the k-loop it would hold ~12
            // VGPRs across the MFMAs = one resident workgroup less per CU; the small tiles have the registers and let it hoist
            if constexpr (CPT > 1) asm volatile("" : "+v"(tl));
            const int cg = tl / NPOS, pos = tl - cg * NPOS;
            if (cg < CG) {
                const int ly0 = y0 / 2 - 1, lx0 = x0 / 2 - 1;
                const int pi = pos / PRW, pj = pos - pi * PRW;
                const int i = ly0 + pi, j = lx0 + pj;
                const int i0 = min(max(i, 0), p.hs - 1), i1 = min(max(i + 1, 0), p.hs - 1);
                const int j0 = min(max(j, 0), p.ws - 1), j1 = min(max(j + 1, 0), p.ws - 1);
                // clamped pairs (image border) take the single source value exactly, like ATen's lambda = 0
                const float xa = j0 == j1 ? 1.f : 0.75f, xb = j0 == j1 ? 0.f : 0.25f;
                const float ya = i0 == i1 ? 1.f : 0.75f, yb = i0 == i1 ? 0.f : 0.25f;
                const int Y = 2 * i + 1, X = 2 * j + 1;
                const bool yt = Y >= 0 && Y < p.H, yb2 = Y + 1 < p.H, xl = X >= 0 && X < p.W, xr = X + 1 < p.W;
                const float m00 = (yt && xl) ? 1.f : 0.f, m01 = (yt && xr) ? 1.f : 0.f, m10 = (yb2 && xl) ? 1.f : 0.f, m11 = (yb2 && xr) ? 1.f : 0.f;
                const float *r0 = raw + (cg * CPT * LH + (i0 - ly0)) * LW + 3 - lx0;
                const float *r1 = raw + (cg * CPT * LH + (i1 - ly0)) * LW + 3 - lx0;
                float *d = hip + (cg * CPT * PH + 2 * pi) * PW + 2 * pj + 3;
#pragma unroll 1
                for (int cc = 0; cc < CPT; ++cc) {
                    const float v00 = r0[cc * LH * LW + j0], v01 = r0[cc * LH * LW + j1];
                    const float v10 = r1[cc * LH * LW + j0], v11 = r1[cc * LH * LW + j1];
                    const float h00 = xa * v00 + xb * v01, h01 = xb * v00 + xa * v01;   // columns 2j+1, 2j+2 of low row i
                    const float h10 = xa * v10 + xb * v11, h11 = xb * v10 + xa * v11;   // ... of low row i+1
                    d[cc * PH * PW] = m00 * (ya * h00 + yb * h10);
                    d[cc * PH * PW + 1] = m01 * (ya * h01 + yb * h11);
                    d[cc * PH * PW + PW] = m10 * (yb * h00 + ya * h10);
                    d[cc * PH * PW + PW + 1] = m11 * (yb * h01 + ya * h11);
                }
            }
            __syncthreads();
        }
        const float *sa = stg + aBase;
        const float *sb = (UPS ? lds + L::HIP : stg + C::WSZ) + bBase;
        // Operand fetches are software-pipelined in a second register set: a macro-step = G k-steps (>= 4 MFMAs); the
        // fetches of macro-step i+1 are issued right after the FIRST MFMA of macro-step i, so when the wave waits for
        // them (the compiler's s_waitcnt lgkmcnt(0) in front of macro-step i+1) they have had >= 3 MFMA times to land and
        // nothing younger is outstanding.  (Left to itself the compiler sinks every ds_read next to its use and exposes
        // the LDS latency two or three times per k-step.)  sched_barrier pins source order = issue order.
        constexpr int S = (C::CK / 2) * KS2;          // k-steps per chunk (2 input channels x 1 tap each)
        constexpr int G = NT * MT >= 4 ? 1 : (NT * MT == 2 ? 2 : 4);
        constexpr int NM = (S + G - 1) / G;
        float a[2][G][NT], bv[2][G][MT];
        auto fetch = [&](int ms, int buf) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int s = ms * G + g;
                if (s < S) {
                    const int cp = s / KS2, t = s % KS2, ky = t / KS, kx = t % KS;
#pragma unroll
                    for (int n = 0; n < NT; ++n) a[buf][g][n] = sa[(2 * cp * KS2 + t) * BN + n * 32];
#pragma unroll
                    for (int my = 0; my < C::MTY; ++my)
#pragma unroll
                        for (int mx = 0; mx < C::MTX; ++mx)
                            bv[buf][g][my * C::MTX + mx] = sb[(2 * cp * PH + my * GH + ky) * PW + mx * GW + kx];
                }
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int ms = 0; ms < NM; ++ms) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (ms * G + g < S) {
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ms & 1][g][n], bv[ms & 1][g][m], acc[n][m], 0, 0, 0);
                            if (g == 0 && n == 0 && m == 0) {
                                __builtin_amdgcn_sched_barrier(0);
                                if (ms + 1 < NM) fetch(ms + 1, (ms + 1) & 1);
                                if (SSM_DMA_SPREAD && ms < L::NI && dma_next) issue_k(ch + 1, (ch + 1) & 1, ms);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (SSM_DMA_SPREAD && L::NI > NM) {
            if (dma_next) {
#pragma unroll
                for (int k = NM; k < L::NI; ++k) issue_k(ch + 1, (ch + 1) & 1, k);
            }
        }
    }

    // ---- epilogue: bias, LeakyReLU, store (and fused 2x2 mean) -------------------------
    // Bias: added after the k-loop like the reference's Conv2d (sum of products, then + bias: the same single rounding, so
    // LeakyReLU branches agree with the reference also for outputs that cancel to ~0) - as one extra MFMA k-step per
    // accumulator with A = the bias column (k = 0) and B = a row of ones: NT vector loads and ONE wait per tile.  (A vector
    // load of the bias inside the store loop put an s_waitcnt vmcnt(0) - which also waits for the previous STORES - in front
    // of each of the 16 register rounds; starting the accumulators from the bias changes the rounding order.)
    {
        float abias[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float bv0 = p.bias[nb * BN + (wn * NT + n) * 32 + l31];       // every lane loads (no divergent branch), half 1 drops it
            abias[n] = (half || (SPLIT && ks != 0)) ? 0.f : bv0;
        }
        const float ones = half ? 0.f : 1.f;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(abias[n], ones, acc[n][m], 0, 0, 0);
    }
    // accumulator register r of lane (l31, half) = cout (r&3) + 8*(r>>2) + 4*half, pixel l31 of the group
    const int xbase = x0 + wx * (C::MTX * GW) + gx;
    const int ybase = y0 + wy * (C::MTY * GH) + gy;
#ifdef SSM_CONV_ABLATE
    if ((p.abl & 2) && acc[0][0][0] != 12345.678f) return;
#endif
    // Lean store loop: one wave-uniform 64-bit base per (cout, batch) + one 32-bit per-lane offset per pixel group (saddr form
    // of global_store), the in-image predicate evaluated once per pixel group, LeakyReLU as max(t, t*slope) (slope 1 = off).
    // The first version recomputed a 64-bit address and three predicates per store: ~16 vector instructions per store, 1600 per
    // wave and tile - a fifth of the MFMA time of a 32->32 3x3 tile.
    const float sl = p.lrelu ? p.slope : 1.f;
    float *dstb = p.dst + (long long)(SPLIT ? b + ks * p.ksB : b) * p.dsb;
    float *poolb = p.pool ? p.pool + (long long)b * p.psb : nullptr;
    const int cu0 = nb * BN + wn * (NT * 32);                    // first cout of this wave's block (uniform)
    const bool full = cu0 + NT * 32 <= p.Cout;                   // uniform: no cout padding in this wave's block
    // per lane: ONE offset into a cout plane (and one into a pooled plane); the pixel groups of the wave differ by uniform amounts
    // (BYTE offsets, < 2^31: one batch entry of a 32-cout block; the stores use the saddr form: 64-bit SGPR base + 32-bit VGPR offset)
    const unsigned pbase = 4u * ((unsigned)(4 * half) * (unsigned)p.dsc + (unsigned)ybase * (unsigned)p.dsh + (unsigned)xbase);
    const unsigned qbase = 4u * ((unsigned)(4 * half) * (unsigned)p.psc + (unsigned)(ybase >> 1) * (unsigned)p.psh + (unsigned)(xbase >> 1));
    auto st = [](const float *base, unsigned off_bytes, float val) {
        asm volatile("global_store_dword %0, %1, %2" ::"v"(off_bytes), "v"(val), "s"(base) : "memory");
    };
    const bool even = (GW == 32) ? !(l31 & 1) : (!(gx & 1) && !(gy & 1));      // the lane that writes a 2x2 mean
    bool pok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) pok[m] = ybase + (m / C::MTX) * GH < p.H && xbase + (m % C::MTX) * GW < p.W;
    auto store_all = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cu = cu0 + n * 32 + (r & 3) + 8 * (r >> 2);       // uniform; this lane's cout = cu + 4*half
                const bool cok = FULL || cu + 4 * half < p.Cout;
                float *bp = dstb + (long long)cu * p.dsc;
                float v[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float t = acc[n][m][r];
                    v[m] = fmaxf(t, t * sl);
                    float *bpm = bp + ((m / C::MTX) * GH) * p.dsh + (m % C::MTX) * GW;        // uniform
                    if (pok[m] && cok) st(bpm, pbase, v[m]);
                }
                if (poolb) {
                    float *qp = poolb + (long long)cu * p.psc;
                    if constexpr (GW == 32) {
                        if constexpr (C::MTY % 2 == 0) {
#pragma unroll
                            for (int my = 0; my < C::MTY; my += 2)
#pragma unroll
                                for (int mx = 0; mx < C::MTX; ++mx) {
                                    float sm = v[my * C::MTX + mx] + v[(my + 1) * C::MTX + mx];
                                    sm += __shfl_xor(sm, 1);
                                    float *qpm = qp + (my / 2) * p.psh + mx * (GW / 2);
                                    if (pok[my * C::MTX + mx] && even && cok) st(qpm, qbase, sm * 0.25f);
                                }
                        }
                    } else {   // 8x4 group: the 2x2 neighbours are lanes ^GW (y) and ^1 (x); same association as the 32x1 form
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            float sm = v[m] + __shfl_xor(v[m], GW);
                            sm += __shfl_xor(sm, 1);
                            float *qpm = qp + ((m / C::MTX) * GH / 2) * p.psh + (m % C::MTX) * (GW / 2);
                            if (pok[m] && even && cok) st(qpm, qbase, sm * 0.25f);
                        }
                    }
                }
            }
        }
    };
    if (full) store_all(std::true_type{});
    else store_all(std::false_type{});
}

// ---- tile configurations ------------------------------------------------------------
//                  KS NT WN MTY MTX WY WX CK GW      BN   TH  TW
using CfgK7 = Cfg<7, 1, 1, 2, 2, 4, 1, 2>;        //  32    8  64   conv1a / conv1b
using CfgK5 = Cfg<5, 2, 1, 2, 2, 4, 1, 2>;        //  64    8  64   conv2a / conv2b
using CfgK3N32 = Cfg<3, 1, 1, 2, 2, 4, 1, 4>;     //  32    8  64   conv11a/b, fuse, final
using CfgK3N64 = Cfg<3, 2, 1, 2, 2, 4, 1, 4>;     //  64    8  64   conv10a/b
using CfgK3N128 = Cfg<3, 2, 2, 2, 2, 2, 1, 4>;    // 128    4  64   conv3..conv9 (wide maps)
using CfgK3N128S = Cfg<3, 2, 2, 2, 1, 2, 1, 4>;   // 128    4  32   same, maps where 64-wide tiles waste columns
// small maps (1/16, 1/32 resolution): more, smaller workgroups to cover 256 CUs
using CfgK3N64T = Cfg<3, 1, 2, 2, 1, 2, 1, 4>;    //  64    4  32
using CfgK3N32T = Cfg<3, 1, 1, 1, 1, 4, 1, 8>;    //  32    4  32   (odd row tile: no fused pool / upsample)
// 8x4 pixel groups: maps 80 / 40 / 22 / 11 ... wide, where 32-wide row segments waste a third of the MFMA columns
using CfgK3N128G = Cfg<3, 2, 2, 2, 2, 2, 1, 4, 8>;   // 128   16  16
using CfgK3N64G = Cfg<3, 2, 1, 1, 2, 2, 2, 4, 8>;    //  64    8  32
using CfgK3N64GS = Cfg<3, 2, 1, 1, 2, 4, 1, 4, 8>;   //  64   16  16
using CfgK3N32G = Cfg<3, 1, 1, 1, 2, 2, 2, 8, 8>;    //  32    8  32
using CfgK3N32GS = Cfg<3, 1, 1, 1, 1, 2, 2, 8, 8>;   //  32    8  16
using CfgK5G = Cfg<5, 2, 1, 1, 2, 2, 2, 2, 8>;       //  64    8  32
using CfgK7G = Cfg<7, 1, 1, 1, 2, 2, 2, 2, 8>;       //  32    8  32

#define SSM_CONV_KINDS(X)                                                                                      \
    X(K7, CfgK7) X(K5, CfgK5) X(K3N32, CfgK3N32) X(K3N64, CfgK3N64) X(K3N128, CfgK3N128) X(K3N128S, CfgK3N128S) \
    X(K3N64T, CfgK3N64T) X(K3N32T, CfgK3N32T) X(K3N128G, CfgK3N128G) X(K3N64G, CfgK3N64G) X(K3N64GS, CfgK3N64GS) \
    X(K3N32G, CfgK3N32G) X(K3N32GS, CfgK3N32GS) X(K5G, CfgK5G) X(K7G, CfgK7G)

enum ConvKind {
#define X(name, cfg) name,
    SSM_CONV_KINDS(X)
#undef X
        NKIND
};

struct KindInfo {
    int ks, bn, th, tw, ck, nt, mt, cpt;
    bool pool_ok, ups_ok;
};

template <class C>
constexpr KindInfo info_of() {
    return KindInfo{C::KS, C::BN, C::TH, C::TW, C::CK, C::NT, C::MT, C::CPT, C::POOL_OK, C::UPS_OK};
}

constexpr KindInfo kInfo[NKIND] = {
#define X(name, cfg) info_of<cfg>(),
    SSM_CONV_KINDS(X)
#undef X
};

std::atomic<int> g_force_kind{-1};    // tests / tuning only (ssm_conv_force_kind)

// Resident workgroups per CU of every kernel instance (registers / LDS), asked from the runtime once; 2 where there is no device
// (the plan then is still a pure function of the problem - nothing is launched there).
template <class C, bool UPS>
int query_occupancy() {
    if constexpr (UPS && !C::UPS_OK) {
        return 1;
    } else {
        int n = 0;
        auto kern = conv_mfma_kernel<C, UPS>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 256, Lds<C, UPS>::BYTES) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            n = 2;
        }
        return n > 8 ? 8 : n;
    }
}

const int (*occupancy_table())[2] {
    static int occ[NKIND][2];
    static std::once_flag once;
    std::call_once(once, [] {
        int i = 0;
#define X(name, cfg)                        \
    occ[i][0] = query_occupancy<cfg, false>(); \
    occ[i][1] = query_occupancy<cfg, true>();  \
    ++i;
        SSM_CONV_KINDS(X)
#undef X
    });
    return occ;
}

// Estimated duration (cycles) of a launch with tile configuration `ki`.  A workgroup needs `mf` cycles of matrix pipe (NT*MT
// MFMAs of 64 cycles per k-step) plus `oth` cycles that are not MFMA (per chunk: barrier + DMA wait, the upsample expansion - more
// per chunk where a thread walks several channels -; per workgroup: prologue + epilogue).  k co-resident workgroups share the pipe
// of their CU: a round of them takes max(k*mf + oth, mf + oth); the launch is its full rounds (k = occupancy) plus one round of
// the remainder spread over the 256 CUs.  Constants fitted to per-layer sweeps of every configuration at batch 1 ... 28
// (tools/tune_conv_f32.py; the picks are within 0.4 % of the per-layer best on average, 1.1 % at worst).
double estimate_cycles(const KindInfo &ki, int occ, int cin8, int Cout, int B, int H, int W, int ups) {
    const long long tiles = (long long)B * ((W + ki.tw - 1) / ki.tw) * ((H + ki.th - 1) / ki.th);
    const long long nwg = tiles * ((Cout + ki.bn - 1) / ki.bn);
    const double mf = (double)ki.nt * ki.mt * (cin8 * ki.ks * ki.ks / 2) * 64.0;
    const double chunks = (double)cin8 / ki.ck;
    const double oth = chunks * (800.0 + (ups ? 400.0 + 400.0 * ki.cpt : 0.0)) + 20000.0;
    const long long slots = 256LL * occ;
    const long long full = nwg / slots, rem = nwg % slots;
    auto round_time = [&](double k) { return (k * mf + oth > mf + oth) ? k * mf + oth : mf + oth; };
    double t = (double)full * round_time((double)occ);
    if (rem) t += round_time((double)((rem + 255) / 256));
    return t;
}

int pick_kind(int k, int Cin, int Cout, int B, int H, int W, int pool, int ups) {
    if (k != 3 && k != 5 && k != 7) return -1;
    const int forced = g_force_kind.load();
    if (forced >= 0 && forced < NKIND && kInfo[forced].ks == k) return forced;
    const int cin8 = (Cin + 7) / 8 * 8;
    int best = -1;
    double bt = 0.0;
    for (int pass = 0; pass < 2 && best < 0; ++pass) {      // second pass: kernel sizes without a narrow cout block (k = 5, Cout <= 32)
        for (int i = 0; i < NKIND; ++i) {
            const KindInfo &ki = kInfo[i];
            if (ki.ks != k) continue;
            if (pool && !ki.pool_ok) continue;
            if (ups && !ki.ups_ok) continue;
            if (pass == 0 && ki.bn > 32 && ki.bn / 2 >= ((Cout + 31) / 32) * 32) continue;   // over half of the cout block = padding
            const double t = estimate_cycles(ki, occupancy_table()[i][ups ? 1 : 0], cin8, Cout, B, H, W, ups);
            if (best < 0 || t < bt * 0.999) {
                best = i;
                bt = t;
            }
        }
    }
    return best;
}

// tile configurations that have a split-K instantiation: the 32-cout 3x3 one the small maps run on (ssm_conv_plan's pick for 11x11 .. 22x22)
template <class C>
constexpr bool conv_split_ok() { return std::is_same<C, CfgK3N32T>::value; }

template <class C, bool UPS>
int launch(ConvParams &p, int B, hipStream_t st) {
    p.tilesX = (p.W + C::TW - 1) / C::TW;
    p.tilesY = (p.H + C::TH - 1) / C::TH;
    p.NB = (p.Cout + C::BN - 1) / C::BN;
    if (p.pool && !C::POOL_OK) {
        ssm::set_error("conv: fused pool needs an even row tile");
        return SSM_E_UNSUPPORTED;
    }
    const long long blocks = (long long)p.tilesX * p.tilesY * p.NB * B * p.NSPLIT;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        ssm::set_error("conv: grid of %lld workgroups out of range", blocks);
        return SSM_E_ARG;
    }
    if constexpr (UPS && !C::UPS_OK) {
        ssm::set_error("conv: this tile configuration has no fused-upsample form");
        return SSM_E_UNSUPPORTED;
    } else {
        constexpr int lds_bytes = Lds<C, UPS>::BYTES;
        if (p.NSPLIT > 1) {
            if constexpr (!UPS && conv_split_ok<C>()) {
                SSM_LAUNCH((conv_mfma_kernel<C, false, true>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, p);
                return ssm::check_launch("ssm_conv2d_splitk_fwd");
            } else {
                ssm::set_error("conv_splitk: this tile configuration has no split-K form");
                return SSM_E_UNSUPPORTED;
            }
        }
        auto kern = conv_mfma_kernel<C, UPS>;
        SSM_LAUNCH(kern, dim3((unsigned)blocks), dim3(256), lds_bytes, st, p);
        return ssm::check_launch(UPS ? "ssm_conv2d_ups_fwd" : "ssm_conv2d_fwd");
    }
}

template <bool UPS>
int dispatch(int kind, ConvParams &p, int B, hipStream_t st) {
    switch (kind) {
#define X(name, cfg) \
    case name: return launch<cfg, UPS>(p, B, st);
        SSM_CONV_KINDS(X)
#undef X
    }
    return SSM_E_UNSUPPORTED;
}

__global__ void pack_weights_kernel(const float *__restrict__ w, const float *__restrict__ bias,
                                    float *__restrict__ wp, float *__restrict__ bp, int Cout, int Cin,
                                    int CinP, int KS2, int BN, long long total, int nbias) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        // packed index -> (nb, cin, tap, n)
        long long r = i;
        const int n = (int)(r % BN);
        r /= BN;
        const int tap = (int)(r % KS2);
        r /= KS2;
        const int cin = (int)(r % CinP);
        const int nb = (int)(r / CinP);
        const int co = nb * BN + n;
        float v = 0.f;
        if (co < Cout && cin < Cin) v = w[((long long)co * Cin + cin) * KS2 + tap];
        wp[i] = v;
    }
    if (i < nbias) bp[i] = (i < Cout) ? bias[i] : 0.f;
}

int fill_common(ConvParams &p, ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed,
                ssm_view y, int H, int W, int Cout, float slope, int flags, int CK, int srcW) {
    SSM_REQUIRE(H > 0 && W > 0 && Cout > 0 && C1 > 0 && C2 >= 0, "conv: bad sizes");
    SSM_REQUIRE(x1.ptr && y.ptr && w_packed && bias_packed, "conv: null pointer");
    SSM_REQUIRE(C1 % CK == 0 && C2 % CK == 0, "conv: channel counts (%d,%d) must be multiples of %d", C1, C2, CK);
    SSM_REQUIRE(ssm::aligned16(x1.ptr) && x1.sh % 4 == 0 && x1.sc % 4 == 0 && x1.sb % 4 == 0,
                "conv: input 1 is not a padded-plane view (16-byte alignment)");
    SSM_REQUIRE(x1.sh >= srcW + 2 * SSM_PADX, "conv: input 1 row stride %d leaves no zero frame for W=%d", x1.sh, srcW);
    SSM_REQUIRE(ssm::aligned16(w_packed), "conv: packed filter must be 16-byte aligned");
    if (C2 > 0) {
        SSM_REQUIRE(x2.ptr && ssm::aligned16(x2.ptr) && x2.sb % 4 == 0, "conv: input 2 is not a padded-plane view");
        SSM_REQUIRE(x2.sh == x1.sh && x2.sc == x1.sc, "conv: cat sources must share row/channel strides");
    }
    SSM_REQUIRE((long long)CK * x1.sc < 0x7fffffffLL, "conv: channel stride too large");
    p.src1 = x1.ptr;
    p.src2 = C2 > 0 ? x2.ptr : x1.ptr;
    p.sb1 = x1.sb;
    p.sb2 = C2 > 0 ? x2.sb : 0;
    p.sc = x1.sc;
    p.sh = x1.sh;
    p.C1 = C1;
    p.Cin = C1 + C2;
    p.wpk = w_packed;
    p.bias = bias_packed;
    p.dst = y.ptr;
    p.dsb = y.sb;
    p.dsc = y.sc;
    p.dsh = y.sh;
    p.pool = nullptr;
    p.psb = p.psc = 0;
    p.psh = 0;
    p.H = H;
    p.W = W;
    p.hs = H / 2;
    p.ws = W / 2;
    p.Cout = Cout;
    p.slope = slope;
    p.lrelu = (flags & SSM_FLAG_LRELU) ? 1 : 0;
    p.abl = 0;
    p.add = nullptr;
    p.asb = p.asc = 0;
    p.ash = 0;
    p.adiv = 1;
    p.NSPLIT = 1;
    p.ksB = 0;
#ifdef SSM_CONV_ABLATE
    if (const char *e = getenv("SSM_CONV_ABL")) p.abl = atoi(e);
#endif
    return SSM_OK;
}

}  // namespace

extern "C" int ssm_conv_plan(int k, int Cin, int Cout, int B, int H, int W, int pool, int ups, int *kind, int *BN, int *CK) {
    const int kd = pick_kind(k, Cin, Cout, B, H, W, pool, ups);
    if (kd < 0) {
        ssm::set_error("conv: no tile configuration for kernel size %d (3, 5, 7 are supported%s)", k, ups ? "; fused upsample: 3" : "");
        return SSM_E_UNSUPPORTED;
    }
    if (kind) *kind = kd;
    if (BN) *BN = kInfo[kd].bn;
    if (CK) *CK = kInfo[kd].ck;
    return SSM_OK;
}

extern "C" int ssm_conv_force_kind(int kind) {
    g_force_kind.store(kind >= 0 && kind < NKIND ? kind : -1);
    return NKIND;
}

extern "C" int ssm_conv_config(int k, int Cout, int B, int H, int W, int pool, int *BN, int *CK) {
    // historical form (no Cin): the plan of a filter with as many input as output channels
    return ssm_conv_plan(k, Cout, Cout, B, H, W, pool, 0, nullptr, BN, CK);
}

extern "C" size_t ssm_packed_weight_floats(int Cout, int CinP, int k, int BN) {
    const size_t nb = (size_t)(Cout + BN - 1) / BN;
    return nb * (size_t)CinP * k * k * BN;
}

extern "C" size_t ssm_packed_bias_floats(int Cout, int BN) { return (size_t)(Cout + BN - 1) / BN * BN; }

extern "C" int ssm_pack_weights(const float *w, const float *bias, float *wp, float *bp, int Cout, int Cin,
                                int CinP, int k, int BN, void *stream) {
    SSM_REQUIRE(w && bias && wp && bp, "pack_weights: null pointer");
    SSM_REQUIRE(Cout > 0 && Cin > 0 && CinP >= Cin && BN > 0 && BN % 32 == 0, "pack_weights: bad sizes");
    const long long total = (long long)ssm_packed_weight_floats(Cout, CinP, k, BN);
    const int nbias = (int)ssm_packed_bias_floats(Cout, BN);
    const long long n = total > nbias ? total : nbias;
    SSM_LAUNCH(pack_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       bias, wp, bp, Cout, Cin, CinP, k * k, BN, total, nbias);
    return ssm::check_launch("ssm_pack_weights");
}

namespace {
int set_add(ConvParams &p, ssm_view add, int add_div, int B, int BN) {
    if (!add.ptr) return SSM_OK;
    SSM_REQUIRE(add_div >= 1 && B % add_div == 0, "conv: the addend serves %d batch entries each, batch %d is no multiple", add_div, B);
    // the kernel loads the addend for every cout of its block without predicates: no padded couts allowed
    SSM_REQUIRE(p.Cout % BN == 0, "conv: the addend form needs Cout (%d) to be a multiple of the cout block (%d) of the tile configuration", p.Cout, BN);
    SSM_REQUIRE(4LL * (4 * add.sc + (long long)(p.H + 64) * add.sh) < 0x7fffffffLL, "conv: addend plane too large for 32-bit offsets");
    p.add = add.ptr;
    p.asb = add.sb;
    p.asc = add.sc;
    p.ash = add.sh;
    p.adiv = add_div;
    return SSM_OK;
}
}  // namespace

extern "C" int ssm_conv2d_add_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                                  ssm_view pool, ssm_view add, int add_div, int B, int H, int W, int Cout, int k, float slope, int flags,
                                  void *stream) {
    int kind = 0, BN = 0, CK = 0;
    SSM_REQUIRE(B > 0, "conv: bad batch");
    const int rc = ssm_conv_plan(k, C1 + C2, Cout, B, H, W, pool.ptr ? 1 : 0, 0, &kind, &BN, &CK);
    if (rc != SSM_OK) return rc;
    ConvParams p;
    const int rf = fill_common(p, x1, C1, x2, C2, w_packed, bias_packed, y, H, W, Cout, slope, flags, CK, W);
    if (rf != SSM_OK) return rf;
    const int ra = set_add(p, add, add_div, B, BN);
    if (ra != SSM_OK) return ra;
    if (pool.ptr) {
        SSM_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv: fused pool needs even H, W");
        p.pool = pool.ptr;
        p.psb = pool.sb;
        p.psc = pool.sc;
        p.psh = pool.sh;
    }
    return dispatch<false>(kind, p, B, (hipStream_t)stream);
}

// ---- split-K for launches that leave most of the chip idle (r5; the direct-form twin of ssm_wino_conv2d_splitk_fwd, csrc/ssm_wino.hip) ----
// The bottleneck convolutions of config 3 on the 11x11 maps (odd width: no Winograd form) are a few dozen workgroups that each walk all
// 512 input channels.  ssm_conv_splitk_plan proposes KS for the tile configuration the filter was packed for; the partial maps are
// finished by ssm_splitk_finish_fwd (csrc/ssm_elem.hip) like the Winograd form's.
extern "C" int ssm_conv_splitk_plan(int k, int Cin, int Cout, int B, int H, int W, int *KS) {
    SSM_REQUIRE(KS, "conv splitk_plan: null pointer");
    *KS = 1;
    const int enabled = ssm::splitk_switch(1).load(std::memory_order_relaxed);          // ($SSM_CONV_SPLITK, ssm_splitk_enable)
    if (!enabled || Cin < 128) return SSM_OK;
    const int kd = pick_kind(k, Cin, Cout, B, H, W, 0, 0);
    if (kd != K3N32T) return SSM_OK;          // (the one configuration with a split-K instantiation, see conv_split_ok)
    const long long nwg = (long long)B * ((W + kInfo[kd].tw - 1) / kInfo[kd].tw) * ((H + kInfo[kd].th - 1) / kInfo[kd].th) *
                          ((Cout + kInfo[kd].bn - 1) / kInfo[kd].bn);
    int ks = 1;
    while (ks < 8 && nwg * ks * 2 <= 512 && Cin % (ks * 2 * kInfo[kd].ck) == 0 && Cin / (ks * 2) >= 64) ks *= 2;
    *KS = ks;
    return SSM_OK;
}

extern "C" int ssm_conv2d_splitk_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed, ssm_view part,
                                     int KS, int B, int H, int W, int Cout, int k, void *stream) {
    int kind = 0, BN = 0, CK = 0;
    SSM_REQUIRE(B > 0 && KS >= 1 && KS <= 8, "conv_splitk: bad batch / split (KS = %d)", KS);
    const int rc = ssm_conv_plan(k, C1 + C2, Cout, B, H, W, 0, 0, &kind, &BN, &CK);          // the configuration the filter was packed for
    if (rc != SSM_OK) return rc;
    SSM_REQUIRE((C1 + C2) % (KS * CK) == 0, "conv_splitk: Cin = %d is no multiple of KS x chunk = %d x %d", C1 + C2, KS, CK);
    SSM_REQUIRE(C2 == 0 || KS == 1, "conv_splitk: one source only (the kernel offsets its first source by the split's channel range)");
    ConvParams p;
    const int rf = fill_common(p, x1, C1, x2, C2, w_packed, bias_packed, part, H, W, Cout, 0.f, 0, CK, W);
    if (rf != SSM_OK) return rf;
    p.NSPLIT = KS;
    p.ksB = B;
    return dispatch<false>(kind, p, B, (hipStream_t)stream);
}

extern "C" int ssm_conv2d_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed,
                              const float *bias_packed, ssm_view y, ssm_view pool, int B, int H, int W, int Cout,
                              int k, float slope, int flags, void *stream) {
    const ssm_view none = {nullptr, 0, 0, 0};
    return ssm_conv2d_add_fwd(x1, C1, x2, C2, w_packed, bias_packed, y, pool, none, 1, B, H, W, Cout, k, slope, flags, stream);
}

extern "C" int ssm_conv2d_ups_add_fwd(ssm_view a, int C1, ssm_view b, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                                      ssm_view add, int add_div, int B, int H, int W, int Cout, float slope, int flags, void *stream) {
    int kind = 0, BN = 0, CK = 0;
    SSM_REQUIRE(B > 0, "conv_ups: bad batch");
    SSM_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv_ups: the output of a x2 upsample has even H, W (got %dx%d)", H, W);
    const int rc = ssm_conv_plan(3, C1 + C2, Cout, B, H, W, 0, 1, &kind, &BN, &CK);
    if (rc != SSM_OK) return rc;
    ConvParams p;
    const int rf = fill_common(p, a, C1, b, C2, w_packed, bias_packed, y, H, W, Cout, slope, flags, CK, W / 2);
    if (rf != SSM_OK) return rf;
    const int ra = set_add(p, add, add_div, B, BN);
    if (ra != SSM_OK) return ra;
    return dispatch<true>(kind, p, B, (hipStream_t)stream);
}

extern "C" int ssm_conv2d_ups_fwd(ssm_view a, int C1, ssm_view b, int C2, const float *w_packed, const float *bias_packed,
                                  ssm_view y, int B, int H, int W, int Cout, float slope, int flags, void *stream) {
    const ssm_view none = {nullptr, 0, 0, 0};
    return ssm_conv2d_ups_add_fwd(a, C1, b, C2, w_packed, bias_packed, y, none, 1, B, H, W, Cout, slope, flags, stream);
}
