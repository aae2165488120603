// 7x7 and 5x5 convolutions as ONE-dimensional Winograd along x - F(2,7) and F(4,5), eight frequencies each - with the direct form
// along y, on the CDNA4 fp32 matrix cores (v_mfma_f32_32x32x2_f32); all arithmetic fp32.
//
// Same operator as ssm_conv.hip for k = 7 / 5 (layers.conv of the reference, scripts/models/layers.py:21-33: stride-1 'same'
// cross-correlation, zero padding, bias, LeakyReLU; the encoder's first two levels, scripts/models/flow_computation.py:36-45;
// fused 2x2 mean, scripts/models/layers.py:60-63), evaluated per row y and tile of M consecutive outputs x = M tx .. M tx + M - 1 as
//
//      Y[cout][y][M tx + a] = sum_f AT[a][f] * M_f[cout][y][tx],      M_f = sum_cin sum_ky U_f[cout][cin][ky] * V_f[cin][y + ky][tx]
//      U_f = sum_kx G[f][kx] w[cout][cin][ky][kx]   (filter rows transformed once per plan, ssm_wino1d_pack_weights)
//      V_f = sum_j  BT[f][j] d[cin][row][M tx - pad + j],  j = 0..7   (8-point input transform of a row, once per chunk in LDS)
//
// over the interpolation points {0, 1, -1, 2, -2, 1/2, -1/2, inf}: 8 multiplies per 2 outputs instead of 14 (k = 7), per 4 outputs
// instead of 20 (k = 5), i.e. 1.75x / 2.5x fewer matrix-core cycles than the direct form.  The two-dimensional forms F(2x2,7x7) /
// F(2x2,5x5) cost O(n^3) vector work per tile and channel - more than the matrix work saved at 32 output channels; in the 1-D form a
// transformed input row is shared by the KS filter rows and all output channels: 26 vector operations per (cin, row, tile) feed
// 8 x KS x Cout multiply-adds.  Numerics (tests/emulate_winograd_1d_precision.py): a single layer 5-8e-6 from a float64 evaluation
// (direct form 2-3e-6) at unit output scale; the whole pair -> frame path unchanged within its fp32 noise.
//
// GEMM view: for each frequency f and filter row ky one fp32 MFMA step  M_f[cout][tile] += U_f[cout][cin pair] V_f[cin pair][tile]:
// A = 32 couts x 2 input channels, B = 2 input channels x 32 tiles, the tile on the lane.  A wave owns a 32-cout x 32-tile block for all
// 8 frequencies (8 accumulators of 16 registers = 128 of its 256 registers: two workgroups per CU, one hides the other's barriers /
// transform / epilogue).  A tile group is 16 tiles x 2 rows, so the 2x2 mean pairs lanes l and l ^ 16.  The output transform is
// lane-local, and a lane stores its M outputs as one 8- / 16-byte piece (a wave writes 128- / 256-byte row segments).  The bias rides
// on the accumulator of the point p = 1: AT[a][1] = 1 for every output a.
//
// Data movement as in ssm_conv.hip: padded planes (the halo is a bigger rectangle), per chunk of CK input channels the [CK][KS][2][BN][4]
// filter values and the raw [CK][TH+KS-1][TW+8] patch arrive by LDS-DMA (global_load_lds_dwordx4, saddr form), double-buffered; the
// workgroup transforms the raw rows into V [CK][rows][2][tiles][4] (both operands of a macro-step are two conflict-free ds_read_b128
// for 8 MFMAs).  The raw patch lands SHIFT floats into its LDS region so that every tile's 8-float window is 8- / 16-byte aligned.
#include "ssm_common.h"

#include <atomic>
#include <mutex>
#include <type_traits>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

struct W1Params {
    const float *src;
    long long sb, sc;    // batch / channel strides
    int sh;              // row stride
    int Cin;
    const float *wpk;    // U, [Cout/BN][Cin][KS][2][BN][4]
    const float *bias;
    float *dst;
    long long dsb, dsc;
    int dsh;
    float *pool;
    long long psb, psc;
    int psh;
    int H, W, Cout;
    int tilesX, tilesY, NB;
    float slope;
    int lrelu;
    int vec;             // 1: outputs / addend / pooled outputs may be moved as aligned M-float pieces (alignment + W % M == 0 checked on the host)
    int abl;             // diagnostics build only: 1 no LDS-DMA in the loop, 2 no stores, 4 no transform
    const float *add;    // optional pre-activation addend [B / adiv][Cout][H][W] (ssm_conv2d_add_fwd)
    long long asb, asc;
    int ash, adiv;
};

// WN cout blocks x (WTY x WTX) tile groups = 4 waves; a tile group = 16 tiles x 2 rows; a tile = M consecutive outputs of one row
template <int KS_, int M_, int WN_, int WTY_, int WTX_, int CK_>
struct W1Cfg {
    static constexpr int KS = KS_, M = M_, WN = WN_, WTY = WTY_, WTX = WTX_, CK = CK_;
    static constexpr int PAD = (KS - 1) / 2, NF = 8;
    static constexpr int GTX = 16, GTY = 2;
    static constexpr int BN = 32 * WN;
    static constexpr int TH = GTY * WTY, NTX = GTX * WTX, TW = M * NTX;      // output rows / tiles per row / output columns per workgroup
    static constexpr int PH = TH + KS - 1, PW = TW + 8, PW4 = PW / 4;          // raw patch rows y0-PAD .., columns x0-4 .. x0+TW+3
    static constexpr int SHIFT = M == 2 ? 1 : 2;                               // floats: window start M tx + (4 - PAD) + SHIFT is a multiple of M
    static constexpr int USZ = CK * KS * 2 * BN * 4;                           // filter floats per chunk
    static constexpr int RSZ = CK * PH * PW;                                   // raw patch floats per chunk
    static constexpr int VSZ = CK * PH * 2 * NTX * 4;                          // transformed rows
    static constexpr int NU = CK * PH * NTX;                                   // transform units (cin, row, tile) per chunk
    static_assert(M + KS - 1 == NF, "F(2,7) or F(4,5): eight points");
    static_assert(WN * WTY * WTX == 4, "4 waves per workgroup");
    static_assert(CK % 2 == 0 && USZ % 256 == 0, "one MFMA k-step = two input channels; filter stage = whole 1-KiB DMA groups");
    static_assert((M * 0 + (4 - PAD) + SHIFT) % M == 0, "aligned windows");
};

template <class C>
struct W1Lds {
    static constexpr int NGU = C::USZ / 256;                    // 1-KiB groups of filter per chunk
    static constexpr int NDQ = C::RSZ / 4;                      // 16-byte pieces of raw patch per chunk
    static constexpr int NGP = (NDQ + 63) / 64;
    static constexpr int NG = NGU + NGP;
    static constexpr int STAGE = NG * 256 + 256;                // floats per stage (+ 1 KiB: the patch lands SHIFT floats in)
    static constexpr int NIU = (NGU + 3) / 4, NIP = (NGP + 3) / 4, NI = NIU + NIP;   // DMA instructions per wave per chunk
    static constexpr int VOFF = 2 * STAGE;                      // transformed rows behind the two stages
    static constexpr int BYTES = (2 * STAGE + C::VSZ) * 4;
    static_assert(BYTES <= 80 * 1024, "LDS budget (two workgroups per CU)");
};

#ifdef SSM_WINO_ABLATE
#define W1ABL(bit) (p.abl & (bit))
#else
#define W1ABL(bit) 0
#endif

template <class C>
__global__ __launch_bounds__(256, 2) void wino1d_kernel(const W1Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using L = W1Lds<C>;
    constexpr int KS = C::KS, M = C::M, BN = C::BN, PH = C::PH, PW = C::PW, CK = C::CK, NTX = C::NTX;

    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid % C::WN, wty = (wid / C::WN) % C::WTY, wtx = wid / (C::WN * C::WTY);
    const int tyl = l31 >> 4, txl = l31 & 15;

    int id = ssm_xcd_tile(blockIdx.x, gridDim.x);
    const int nb = id % p.NB;
    id /= p.NB;
    const int tx = id % p.tilesX;
    id /= p.tilesX;
    const int ty = id % p.tilesY;
    const int b = id / p.tilesY;
    const int x0 = tx * C::TW, y0 = ty * C::TH;

    const float *pbase = p.src + (long long)b * p.sb + (long long)(y0 - C::PAD) * p.sh + (x0 - 4);
    const float *wbase = p.wpk + (long long)nb * p.Cin * (KS * 2 * BN * 4);

    // per-lane source offsets (bytes) of the raw-patch pieces this wave brings per chunk; the filter pieces are linear
    int poff[L::NIP];
#pragma unroll
    for (int i = 0; i < L::NIP; ++i) {
        const int qq = (i * 4 + wid) * 64 + lane;
        if (qq < L::NDQ) {
            const int c = qq / (PH * C::PW4);
            const int rem = qq - c * (PH * C::PW4);
            const int r = rem / C::PW4;
            const int j = rem - r * C::PW4;
            // overshoot rows / pieces are read from the zero frame, never from behind the padded plane (see ssm_wino4.hip)
            const int re = min(r, p.H + (SSM_PADY - 1) - (y0 - C::PAD)), fe = min(4 * j, ((p.W + 2 * SSM_PADX + 3) & ~3) - 4 - x0);
            poff[i] = ((int)(c * p.sc) + re * p.sh + fe) * 4;
        } else {
            poff[i] = 0;          // tail of the last 1-KiB piece: lands in the stage's padding
        }
    }
    const int uoff = lane * 16;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
    // k-th DMA instruction of this wave for chunk `ch` into stage `stage`: k < NIU filter group 4k + wave, else raw patch
    auto issue_k = [&](int ch, int stage, int k) {
        const int c0 = ch * CK;
        const unsigned lsb = lds0 + (unsigned)(stage * L::STAGE) * 4u;
        if (k < L::NIU) {
            const int g = 4 * k + wid;
            if (g < L::NGU) {
                const float *base = wbase + (long long)c0 * (KS * 2 * BN * 4) + g * 256;
                const unsigned m0v = lsb + (unsigned)g * 1024u;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(uoff), "s"(base), "s"(m0v) : "memory", "m0");
            }
        } else {
            const int kk = k - L::NIU;
            const int g = 4 * kk + wid;
            if (g < L::NGP) {
                const float *base = pbase + (long long)c0 * p.sc;
                const unsigned m0v = lsb + (unsigned)(L::NGU + g) * 1024u + 4u * C::SHIFT;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(poff[kk]), "s"(base), "s"(m0v) : "memory", "m0");
            }
        }
    };
    auto issue = [&](int ch, int stage) {
#pragma unroll
        for (int k = 0; k < L::NI; ++k) issue_k(ch, stage, k);
    };

    f32x16 acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    const int nchunks = p.Cin / CK;
    issue(0, 0);
    // Bias: AT[a][1] = 1 for every output a of a tile (the point p = 1), so the accumulator of frequency 1 starts from the bias - one
    // MFMA k-step with A = the bias column and B = a row of ones, issued while the first chunk is in flight.
    {
        const float bv0 = p.bias[nb * BN + wn * 32 + l31];
        const float ab = half ? 0.f : bv0, ones = half ? 0.f : 1.f;
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab, ones, acc[1], 0, 0, 0);
    }

    // per-lane operand bases in f32x4 units: U of (cin = 2cp + half, ky, fq, cout = wn*32 + l31) inside a stage; V of (cin, row, fq, tile)
    const f32x4 *lds4 = (const f32x4 *)lds;
    const int aBase = half * (KS * 2 * BN) + wn * 32 + l31;
    const int bBase = L::VOFF / 4 + (half * PH + wty * C::GTY + tyl) * (2 * NTX) + wtx * C::GTX + txl;

    constexpr int S = (CK / 2) * KS;           // macro-steps per chunk: (cin pair, filter row) = 8 MFMAs
    f32x4 a[2][2], bq[2][2];
    auto fetch = [&](int stage, int s, int buf) {
        const int cp = s / KS, ky = s % KS;
        const int ai = stage * (L::STAGE / 4) + aBase + (2 * cp * KS + ky) * (2 * BN);
        const int bi = bBase + (2 * cp * PH + ky) * (2 * NTX);
        a[buf][0] = lds4[ai];
        a[buf][1] = lds4[ai + BN];
        bq[buf][0] = lds4[bi];
        bq[buf][1] = lds4[bi + NTX];
    };

    for (int ch = 0; ch < nchunks; ++ch) {
        const int stage = ch & 1;
        // chunk ch has landed for every wave; every wave is done with the MFMAs of chunk ch-1 (V and the other stage are free)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool dma_next = ch + 1 < nchunks && !(W1ABL(1) && ch >= 1);
#ifndef W1_DMA_EARLY
#define W1_DMA_EARLY 0      // 1: the whole DMA of chunk ch+1 as a burst right behind the barrier (its stage was released by chunk ch-1)
#endif                      // - measured 2-3 % slower than one instruction per macro-step (conv1b 6.22 vs 6.01 ms at batch 14)
        if (W1_DMA_EARLY && dma_next) issue(ch + 1, stage ^ 1);

        // ---- input transform: raw rows -> V.  One unit = (cin, row, tile): 8 floats in, 8 frequencies out -------------------------
        if (!W1ABL(4) || ch == 0) {
            const float *raw = lds + stage * L::STAGE + L::NGU * 256 + C::SHIFT;
            f32x4 *vout = (f32x4 *)(lds + L::VOFF);
            // (all window reads of a thread's units first, then the arithmetic: one LDS latency per chunk, not one per unit)
            constexpr int NUI = (C::NU + 255) / 256;
            float d[NUI][8];
#pragma unroll
            for (int ui = 0; ui < NUI; ++ui) {
                const int u = ui * 256 + tid;
                if (ui * 256 + 256 <= C::NU || u < C::NU) {
                    const int t = u % NTX, cr = u / NTX;          // cr = cin * PH + row
                    const float *rp = raw + cr * PW + M * t + (4 - C::PAD);
                    if constexpr (M == 2) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x2 q = *(const f32x2 *)(rp + 2 * j);
                            d[ui][2 * j] = q[0];
                            d[ui][2 * j + 1] = q[1];
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const f32x4 q = *(const f32x4 *)(rp + 4 * j);
                            d[ui][4 * j] = q[0];
                            d[ui][4 * j + 1] = q[1];
                            d[ui][4 * j + 2] = q[2];
                            d[ui][4 * j + 3] = q[3];
                        }
                    }
                }
            }
#pragma unroll
            for (int ui = 0; ui < NUI; ++ui) {
                const int u = ui * 256 + tid;
                if (ui * 256 + 256 <= C::NU || u < C::NU) {
                    const int t = u % NTX, cr = u / NTX;
                    const float *e = d[ui];
                    // BT of the points {0, 1, -1, 2, -2, 1/2, -1/2, inf} (the F(6,3) matrix of Lavin & Gray; it depends on the
                    // points only, not on the split of the 8 points into outputs and taps); scalar arithmetic (check_isa.sh fences v_pk_*)
                    const float v0 = (e[0] - e[6]) + 5.25f * (e[4] - e[2]);
                    const float v7 = (e[7] - e[1]) + 5.25f * (e[3] - e[5]);
                    const float t1 = (e[2] + e[6]) - 4.25f * e[4], t2 = (e[1] + e[5]) - 4.25f * e[3];
                    const float t3 = (e[6] + 0.25f * e[2]) - 1.25f * e[4], t4 = (0.5f * e[1] - 2.5f * e[3]) + 2.f * e[5];
                    const float t5 = (e[6] + 4.f * e[2]) - 5.f * e[4], t6 = (2.f * e[1] - 2.5f * e[3]) + 0.5f * e[5];
                    const f32x4 o0 = {v0, t1 + t2, t1 - t2, t3 + t4};
                    const f32x4 o1 = {t3 - t4, t5 + t6, t5 - t6, v7};
                    vout[cr * (2 * NTX) + t] = o0;
                    vout[cr * (2 * NTX) + NTX + t] = o1;
                }
            }
        }
        __syncthreads();

        // ---- matrix phase: S macro-steps of 8 MFMAs; operands of step s+1 fetched behind the first MFMA of step s; the DMA of
        // chunk ch+1 issued one instruction per macro-step --------------------------------------------------------------------------
        fetch(stage, 0, 0);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int buf = s & 1;
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][f >> 2][f & 3], bq[buf][f >> 2][f & 3], acc[f], 0, 0, 0);
                if (f == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 1 < S) fetch(stage, s + 1, buf ^ 1);
                    if (!W1_DMA_EARLY && s < L::NI && dma_next) issue_k(ch + 1, stage ^ 1, s);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (L::NI > S && !W1_DMA_EARLY) {
            if (dma_next) {
#pragma unroll
                for (int k = S; k < L::NI; ++k) issue_k(ch + 1, stage ^ 1, k);
            }
        }
    }

    // ---- epilogue: output transform, addend, LeakyReLU, stores (and the fused 2x2 mean) -------------------------------------------
#ifdef SSM_WINO_ABLATE
    if ((p.abl & 2) && acc[0][0] != 12345.678f) return;
#endif
    {
        const int px = x0 + (wtx * C::GTX + txl) * M, py = y0 + wty * C::GTY + tyl;
        const float sl = p.lrelu ? p.slope : 1.f;
        float *dstb = p.dst + (long long)b * p.dsb;
        float *poolb = p.pool ? p.pool + (long long)b * p.psb : nullptr;
        const int cu0 = nb * BN + wn * 32;          // Cout is a multiple of BN (checked on the host): no cout padding
        const unsigned pb = 4u * ((unsigned)(4 * half) * (unsigned)p.dsc + (unsigned)py * (unsigned)p.dsh + (unsigned)px);
        const unsigned qb = 4u * ((unsigned)(4 * half) * (unsigned)p.psc + (unsigned)(py >> 1) * (unsigned)p.psh + (unsigned)(px >> 1));
        const bool rok = py < p.H;
        const bool vok = rok && px + M <= p.W && p.vec;          // whole tile inside the map, as one aligned piece
        auto st4 = [](const float *base, unsigned off_bytes, f32x4 val) {
            asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(off_bytes), "v"(val), "s"(base) : "memory");
        };
        auto st2 = [](const float *base, unsigned off_bytes, f32x2 val) {
            asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off_bytes), "v"(val), "s"(base) : "memory");
        };
        auto st1 = [](const float *base, unsigned off_bytes, float val) {
            asm volatile("global_store_dword %0, %1, %2" ::"v"(off_bytes), "v"(val), "s"(base) : "memory");
        };
        const float *addb = p.add ? p.add + (long long)(b / p.adiv) * p.asb + (long long)(4 * half) * p.asc + (long long)py * p.ash + px : nullptr;
        // M = 2: the addend pair of output register r + 4 is fetched at the top of iteration r (the stores below are ordered asm
        // statements: a load issued behind one cannot move above it, and fetched in its own iteration every addend would wait out its
        // whole latency - 16 exposed latencies per workgroup, 0.2 ms per pair on stage 2's conv1a)
        constexpr int AQD = 4;
        f32x2 addq[M == 2 ? 16 : 1];
        auto addr_of = [&](int r) { return addb + (long long)(cu0 + (r & 3) + 8 * (r >> 2)) * p.asc; };
        if constexpr (M == 2) {
            if (addb && vok) {
#pragma unroll
                for (int r = 0; r < AQD; ++r) addq[r] = *(const f32x2 *)addr_of(r);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cu = cu0 + (r & 3) + 8 * (r >> 2);          // uniform; this lane's cout = cu + 4 * half
            if constexpr (M == 2) {
                if (addb && vok && r + AQD < 16) addq[r + AQD] = *(const f32x2 *)addr_of(r + AQD);
            }
            float y[M];
            {
                const float s12p = acc[1][r] + acc[2][r], s12m = acc[1][r] - acc[2][r];
                const float s34p = acc[3][r] + acc[4][r], s34m = acc[3][r] - acc[4][r];
                const float s56p = acc[5][r] + acc[6][r], s56m = acc[5][r] - acc[6][r];
                y[0] = (acc[0][r] + s12p) + (s34p + s56p);
                if constexpr (M == 2) {
                    y[1] = (s12m + acc[7][r]) + (2.f * s34m + 0.5f * s56m);
                } else {
                    y[1] = s12m + (2.f * s34m + 0.5f * s56m);
                    y[2] = s12p + (4.f * s34p + 0.25f * s56p);
                    y[3] = (s12m + acc[7][r]) + (8.f * s34m + 0.125f * s56m);
                }
            }
            if (addb) {
                const float *ap = addb + (long long)cu * p.asc;
                if (vok) {
                    if constexpr (M == 2) {
                        y[0] += addq[r][0];
                        y[1] += addq[r][1];
                    } else {
                        const f32x4 q = *(const f32x4 *)ap;
                        y[0] += q[0];
                        y[1] += q[1];
                        y[2] += q[2];
                        y[3] += q[3];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < M; ++e)
                        if (rok && px + e < p.W) y[e] += ap[e];
                }
            }
#pragma unroll
            for (int e = 0; e < M; ++e) y[e] = fmaxf(y[e], y[e] * sl);
            float *bp = dstb + (long long)cu * p.dsc;
            if (vok) {
                if constexpr (M == 2) {
                    const f32x2 o = {y[0], y[1]};
                    st2(bp, pb, o);
                } else {
                    const f32x4 o = {y[0], y[1], y[2], y[3]};
                    st4(bp, pb, o);
                }
            } else {
#pragma unroll
                for (int e = 0; e < M; ++e)
                    if (rok && px + e < p.W) st1(bp + e, pb, y[e]);
            }
            if (poolb) {
                // 2x2 mean: vertical pairs first (lane l and l ^ 16 hold rows y, y + 1 of the same tile), then the horizontal pair -
                // the association of the direct kernel
                float *qp = poolb + (long long)cu * p.psc;
                float sv[M];
#pragma unroll
                for (int e = 0; e < M; ++e) sv[e] = y[e] + __shfl_xor(y[e], 16);
                const bool pok = tyl == 0 && rok;          // H, W even (checked on the host)
                if constexpr (M == 2) {
                    if (pok && px + 2 <= p.W) st1(qp, qb, (sv[0] + sv[1]) * 0.25f);
                } else {
                    const f32x2 o = {(sv[0] + sv[1]) * 0.25f, (sv[2] + sv[3]) * 0.25f};
                    if (pok && px + 4 <= p.W && p.vec) st2(qp, qb, o);
                    else if (pok) {
                        if (px + 2 <= p.W) st1(qp, qb, o[0]);
                        if (px + 4 <= p.W) st1(qp + 1, qb, o[1]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- tile configurations -----------------------------------------------------------------------------------------------
//                    KS M WN WTY WTX CK        BN  TH   TW
using R7A = W1Cfg<7, 2, 1, 4, 1, 2>;      //    32   8   32     conv1a / conv1b
using R7B = W1Cfg<7, 2, 1, 2, 2, 2>;      //    32   4   64
using R5A = W1Cfg<5, 4, 2, 2, 1, 2>;      //    64   4   64     conv2a / conv2b
using R5B = W1Cfg<5, 4, 2, 1, 2, 2>;      //    64   2  128
using R5C = W1Cfg<5, 4, 1, 4, 1, 2>;      //    32   8   64     (32-cout blocks: Cout not a multiple of 64)

#define SSM_W1_KINDS(X) X(R7A_, R7A) X(R7B_, R7B) X(R5A_, R5A) X(R5B_, R5B) X(R5C_, R5C)

enum W1Kind {
#define X(name, cfg) name,
    SSM_W1_KINDS(X)
#undef X
        NW1KIND
};

struct W1KindInfo {
    int ks, m, bn, th, tw, ck, ph;
};

template <class C>
constexpr W1KindInfo w1info_of() {
    return W1KindInfo{C::KS, C::M, C::BN, C::TH, C::TW, C::CK, C::PH};
}

constexpr W1KindInfo kW1Info[NW1KIND] = {
#define X(name, cfg) w1info_of<cfg>(),
    SSM_W1_KINDS(X)
#undef X
};

std::atomic<int> g_force_w1kind{-1};

// Estimated duration (cycles) of a launch: two co-resident workgroups per CU share the matrix pipe; a CU-round of two workgroups
// costs their matrix work (Cin/2 x KS x 8 MFMAs of 64 cycles each, per wave) plus what the neighbour cannot hide (per chunk: two
// barriers + the transform, by its (cin, row, tile) units; per workgroup: prologue + epilogue); whole rounds only.  Measured at batch 14
// (tools/bench_layers_wino1d.py): conv1b 6.05 ms with the 8x32 tile vs 6.32 with 4x64, conv2b 2.28 (4x64) vs 2.42 (2x128).
double estimate_w1(const W1KindInfo &ki, int Cin, int Cout, int B, int H, int W) {
    const long long tiles = (long long)B * ((W + ki.tw - 1) / ki.tw) * ((H + ki.th - 1) / ki.th);
    const long long nwg = tiles * (Cout / ki.bn);
    const double mf = (double)(Cin / 2) * ki.ks * 8.0 * 64.0;
    const double chunks = (double)Cin / ki.ck;
    const double nu = (double)ki.ck * ki.ph * (ki.tw / ki.m);          // transform units per chunk: the taller tiles share more of the halo rows
    const double per = 2.0 * mf * 1.1 + chunks * (300.0 + 3.0 * nu) + 6000.0;
    const long long full = nwg / 512, rem = nwg % 512;
    double t = (double)full * per;
    if (rem) t += rem > 256 ? per : mf * 1.2 + chunks * 800.0 + 12000.0;
    return t;
}

int pick_w1kind(int k, int Cin, int Cout, int B, int H, int W) {
    const int forced = g_force_w1kind.load();
    int best = -1;
    double bt = 0.0;
    for (int i = 0; i < NW1KIND; ++i) {
        const W1KindInfo &ki = kW1Info[i];
        if (ki.ks != k || Cout % ki.bn || Cin % ki.ck) continue;
        if (i == forced) return i;
        const double t = estimate_w1(ki, Cin, Cout, B, H, W);
        if (best < 0 || t < bt * 0.999) {
            best = i;
            bt = t;
        }
    }
    return best;
}

template <class C>
int w1launch(W1Params &p, int B, hipStream_t st) {
    p.tilesX = (p.W + C::TW - 1) / C::TW;
    p.tilesY = (p.H + C::TH - 1) / C::TH;
    p.NB = p.Cout / C::BN;
    const long long blocks = (long long)p.tilesX * p.tilesY * p.NB * B;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        ssm::set_error("wino1d conv: grid of %lld workgroups out of range", blocks);
        return SSM_E_ARG;
    }
    constexpr int lds_bytes = W1Lds<C>::BYTES;
    auto kern = wino1d_kernel<C>;
    static std::atomic<uint64_t> lds_reserved{0};          // one bit per device: the attribute is per (kernel, device)
    const hipError_t attr_rc = ssm::reserve_lds(lds_reserved, (const void *)kern, lds_bytes);
    if (attr_rc != hipSuccess) {
        ssm::set_error("wino1d conv: cannot reserve %d bytes of LDS: %s", lds_bytes, hipGetErrorString(attr_rc));
        return SSM_E_LAUNCH;
    }
    SSM_LAUNCH(kern, dim3((unsigned)blocks), dim3(256), lds_bytes, st, p);
    return ssm::check_launch("ssm_wino1d_conv2d_fwd");
}

int w1dispatch(int kind, W1Params &p, int B, hipStream_t st) {
    switch (kind) {
#define X(name, cfg) \
    case name: return w1launch<cfg>(p, B, st);
        SSM_W1_KINDS(X)
#undef X
    }
    return SSM_E_UNSUPPORTED;
}

// U_f[ky] = sum_kx G[f][kx] w[ky][kx];  G[f][k] = c_f p_f^k over the points p = {0, 1, -1, 2, -2, 1/2, -1/2}, c = {1, -2/9, -2/9, 1/90,
// 1/90, 32/45, 32/45} (the scaling that goes with BT above), G[7][k] = [k == KS-1] (the point at infinity).  Evaluated in float64,
// rounded once.  Packed index -> (nb, cin, ky, fq, n, e), frequency f = 4 fq + e.
__global__ void wino1d_pack_kernel(const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ wp,
                                   float *__restrict__ bp, int Cout, int Cin, int CinP, int KS, int BN, long long total, int nbias) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        long long r = i;
        const int e = (int)(r % 4);
        r /= 4;
        const int n = (int)(r % BN);
        r /= BN;
        const int fq = (int)(r % 2);
        r /= 2;
        const int ky = (int)(r % KS);
        r /= KS;
        const int cin = (int)(r % CinP);
        const int nb = (int)(r / CinP);
        const int co = nb * BN + n, f = 4 * fq + e;
        double val = 0.0;
        if (co < Cout && cin < Cin) {
            const float *g = w + (((long long)co * Cin + cin) * KS + ky) * KS;
            if (f == 7) {
                val = (double)g[KS - 1];
            } else {
                const double pt[7] = {0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5};
                const double cf[7] = {1.0, -2.0 / 9.0, -2.0 / 9.0, 1.0 / 90.0, 1.0 / 90.0, 32.0 / 45.0, 32.0 / 45.0};
                double pw = 1.0;
                for (int k = 0; k < KS; ++k) {
                    val += pw * (double)g[k];
                    pw *= pt[f];
                }
                val *= cf[f];
            }
        }
        wp[i] = (float)val;
    }
    if (i < nbias) bp[i] = (i < Cout) ? bias[i] : 0.f;
}

}  // namespace

extern "C" int ssm_wino1d_plan(int k, int Cin, int Cout, int B, int H, int W, int *kind, int *BN, int *CK) {
    const int kd = (k == 7 || k == 5) ? pick_w1kind(k, (Cin + 1) / 2 * 2, Cout, B, H, W) : -1;
    if (kd < 0) {
        ssm::set_error("wino1d conv: no tile configuration for k=%d Cin=%d Cout=%d (k = 7 / 5, Cout a multiple of 32)", k, Cin, Cout);
        return SSM_E_UNSUPPORTED;
    }
    if (kind) *kind = kd;
    if (BN) *BN = kW1Info[kd].bn;
    if (CK) *CK = kW1Info[kd].ck;
    return SSM_OK;
}

extern "C" int ssm_wino1d_force_kind(int kind) {
    g_force_w1kind.store(kind >= 0 && kind < NW1KIND ? kind : -1);
    return NW1KIND;
}

extern "C" size_t ssm_wino1d_packed_weight_floats(int Cout, int CinP, int k, int BN) {
    return (size_t)(Cout / BN) * (size_t)CinP * k * 2 * BN * 4;
}

extern "C" int ssm_wino1d_pack_weights(const float *w, const float *bias, float *wp, float *bp, int Cout, int Cin, int CinP, int k, int BN,
                                       void *stream) {
    SSM_REQUIRE(w && bias && wp && bp, "wino1d pack_weights: null pointer");
    SSM_REQUIRE((k == 7 || k == 5) && Cout > 0 && Cin > 0 && CinP >= Cin && BN > 0 && BN % 32 == 0 && Cout % BN == 0,
                "wino1d pack_weights: bad sizes (k = 7 / 5, Cout a multiple of the cout block)");
    const long long total = (long long)ssm_wino1d_packed_weight_floats(Cout, CinP, k, BN);
    const int nbias = (int)ssm_packed_bias_floats(Cout, BN);
    const long long n = total > nbias ? total : nbias;
    SSM_LAUNCH(wino1d_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, bias, wp, bp, Cout, Cin,
                       CinP, k, BN, total, nbias);
    return ssm::check_launch("ssm_wino1d_pack_weights");
}

extern "C" int ssm_wino1d_conv2d_add_fwd(ssm_view x, int Cin, const float *w_packed, const float *bias_packed, ssm_view y, ssm_view pool,
                                         ssm_view add, int add_div, int B, int H, int W, int Cout, int k, float slope, int flags, void *stream) {
    int kind = 0, BN = 0, CK = 0;
    SSM_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && Cin > 0, "wino1d conv: bad sizes");
    const int rc = ssm_wino1d_plan(k, Cin, Cout, B, H, W, &kind, &BN, &CK);
    if (rc != SSM_OK) return rc;
    const int M = kW1Info[kind].m;
    SSM_REQUIRE(x.ptr && y.ptr && w_packed && bias_packed, "wino1d conv: null pointer");
    SSM_REQUIRE(Cin % CK == 0, "wino1d conv: the channel count (%d) must be a multiple of %d (pad the view)", Cin, CK);
    SSM_REQUIRE(ssm::aligned16(x.ptr) && x.sh % 4 == 0 && x.sc % 4 == 0 && x.sb % 4 == 0,
                "wino1d conv: the input is not a padded-plane view (16-byte alignment)");
    SSM_REQUIRE(x.sh >= W + 2 * SSM_PADX, "wino1d conv: input row stride %d leaves no zero frame for W=%d", x.sh, W);
    SSM_REQUIRE(ssm::aligned16(w_packed), "wino1d conv: packed filter must be 16-byte aligned");
    SSM_REQUIRE((long long)CK * x.sc * 4 < 0x7fffffffLL, "wino1d conv: channel stride too large");
    W1Params p;
    p.src = x.ptr;
    p.sb = x.sb;
    p.sc = x.sc;
    p.sh = x.sh;
    p.Cin = Cin;
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
    p.Cout = Cout;
    p.slope = slope;
    p.lrelu = (flags & SSM_FLAG_LRELU) ? 1 : 0;
    p.abl = 0;
    p.add = nullptr;
    p.asb = p.asc = 0;
    p.ash = 0;
    p.adiv = 1;
#ifdef SSM_WINO_ABLATE
    if (const char *e = getenv("SSM_WINO1D_ABL")) p.abl = atoi(e);
#endif
    const size_t am = (size_t)M * 4 - 1;          // M-float pieces: 8- / 16-byte alignment of every view the kernel moves them through
    bool vec = W % M == 0 && (reinterpret_cast<size_t>(y.ptr) & am) == 0 && y.sh % M == 0 && y.sc % M == 0 && y.sb % M == 0;
    if (add.ptr) {
        SSM_REQUIRE(add_div >= 1 && B % add_div == 0, "wino1d conv: the addend serves %d batch entries each, batch %d is no multiple", add_div, B);
        p.add = add.ptr;
        p.asb = add.sb;
        p.asc = add.sc;
        p.ash = add.sh;
        p.adiv = add_div;
        vec = vec && (reinterpret_cast<size_t>(add.ptr) & am) == 0 && add.sh % M == 0 && add.sc % M == 0 && add.sb % M == 0;
    }
    if (pool.ptr) {
        SSM_REQUIRE(H % 2 == 0 && W % 2 == 0, "wino1d conv: fused pool needs even H, W");
        p.pool = pool.ptr;
        p.psb = pool.sb;
        p.psc = pool.sc;
        p.psh = pool.sh;
        const size_t pm = (size_t)(M / 2) * 4 - 1;
        vec = vec && (reinterpret_cast<size_t>(pool.ptr) & pm) == 0 && pool.sh % (M / 2) == 0 && pool.sc % (M / 2) == 0 && pool.sb % (M / 2) == 0;
    }
    p.vec = vec ? 1 : 0;
    return w1dispatch(kind, p, B, (hipStream_t)stream);
}
