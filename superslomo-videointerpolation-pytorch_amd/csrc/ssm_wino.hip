// 3x3 convolution as Winograd F(2x2,3x3) on the CDNA4 fp32 matrix cores (v_mfma_f32_32x32x2_f32), all arithmetic fp32.
//
// Same operator as ssm_conv.hip for k = 3 (layers.conv of the reference, scripts/models/layers.py:21-33: stride-1 'same'
// cross-correlation, zero padding, bias, LeakyReLU; fused 2x2 mean, scripts/models/layers.py:60-63; two-source input =
// torch.cat on C; fused F.upsample(torch.cat([a, b], 1), bilinear x2), scripts/models/flow_computation.py:244-247), evaluated as
//
//      Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A          per 2x2 output tile (d = its 4x4 input patch, g = the 3x3 filter)
//
// i.e. 16 multiplies per (cin, cout, 4 outputs) instead of 36: 2.25x fewer matrix-core cycles than the direct form for the same
// result in exact arithmetic.  In fp32 the rounding differs from the direct fmaf chain by about as much as a different summation
// order does (transform entries are 0, +-1, +-1/2 only): tests/emulate_winograd_precision.py measures both forms 2.3e-4 from a
// float64 evaluation of the whole pair -> frame path at 736x1280.
//
// GEMM view: for each of the 16 "frequencies" f,  M_f[cout][tile] = sum_cin U_f[cout][cin] * V_f[cin][tile].  A wave owns a
// 32-cout x 32-tile block for ALL 16 frequencies: 16 accumulators of 16 registers (the 256 accumulation VGPRs of a wave that has
// the SIMD to itself - one 4-wave workgroup per CU).  With every frequency of a tile on the same lane, both transforms are
// lane-local: the input transform B^T d B (32 add/sub per k-step, from 16 LDS values of the activation patch) feeds the B operands,
// the output transform A^T M A (24 add/sub per cout) runs on the accumulators in the epilogue, and each lane stores 2x2 pixel blocks
// (8-byte stores, a wave writes 256-byte row segments).  The filter is pre-transformed once per plan (U = G g G^T, 16 floats per
// (cout, cin), ssm_wino_pack_weights) in the order the A operands are read: [cin][f/4][cout][f%4] -> one ds_read_b128 per 4 frequencies.
//
// Data movement: as in ssm_conv.hip the input is the padded-plane layout, a tile's halo is a bigger rectangle; per chunk of CK input
// channels the workgroup stages [CK][16][BN] filter values and the [CK][TH+2][TW+8] patch with LDS-DMA, double-buffered.  The one
// barrier per chunk sits inside the LAST k-step of the chunk: by then all LDS reads of the chunk are complete (the last k-step's
// operands are in registers), so its stage is handed to the DMA of chunk c+2 at once, chunk c+1 (issued a chunk ago) becomes visible,
// and the operands of its first k-step are fetched and transformed behind the remaining MFMAs - the matrix pipe never waits for LDS.
//
// Two kernel forms share this file: wino_kernel (this description: 16 frequencies per wave, one workgroup per CU) and wino2_kernel
// (further down: 8 frequencies per wave, two workgroups per CU - what the plan picks for most layers; profiles/DESIGN_history_r1-r3.md 3.2d says why).
#include "ssm_common.h"

#include <cstring>

#include <atomic>
#include <mutex>
#include <type_traits>
#include <cstdlib>

#ifndef WS_B64
#define WS_B64 1        // B-operand rows as two aligned ds_read_b64 (patch stored one float in); 0: ds_read2_b32 on the aligned patch
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

struct WinoParams {
    const float *src1;
    const float *src2;
    long long sb1, sb2;  // batch strides
    long long sc;        // channel stride (both sources)
    int sh;              // row stride (both sources)
    int C1, Cin;         // channels of source 1, total
    const float *wpk;    // U, [Cout/BN][Cin][4][BN][4]
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
    int abl;             // diagnostics build only (make wabl): 1 no LDS-DMA in the loop, 2 no stores, 4 no B fetch/transform, 8 no A fetch, 16 no barrier
    // optional pre-activation addend [B / adiv][Cout][H][W] (batch entry b reads entry b / adiv): the batch-independent part of the
    // sum, e.g. the stage-1 half of stage 2's conv7a (cross skip), equal for the 7 t of a pair; added after the output transform
    const float *add;
    long long asb, asc;
    int ash, adiv;
    // split-K launches (wino2_kernel only; ssm_wino_conv2d_splitk_fwd): KS workgroups share an output tile, workgroup ks sums input
    // channels [ks, ks + 1) * Cin / KS and stores its raw sums (the bias rides with ks = 0; no activation) as batch entry ks * ksB + b of dst
    int KS, ksB;
};

// WN cout blocks x (WTY x WTX) tile groups = 4 waves; a tile group = GTW x (32/GTW) Winograd tiles of 2x2 pixels
// NBLK = 4: wino_kernel (a wave = one block, all 16 frequencies); NBLK = 2: wino2_kernel (a wave = one block, 8 frequencies)
template <int WN_, int WTY_, int WTX_, int GTW_, int CK_, int NBLK_ = 4, int NST_ = 2>
struct WCfg {
    static constexpr int WN = WN_, WTY = WTY_, WTX = WTX_, GTW = GTW_, GTH = 32 / GTW_, CK = CK_, NBLK = NBLK_;
    static constexpr int NST = NST_;           // LDS stages of the DMA ring (form 2; form 1 is double-buffered)
    static constexpr int BN = 32 * WN;
    static constexpr int TH = 2 * GTH * WTY, TW = 2 * GTW * WTX;      // output pixels per workgroup
    static constexpr int PH = TH + 2, PW = TW + 8, PW4 = PW / 4;       // patch rows y0-1 .. y0+TH, columns x0-4 .. x0+TW+3 (stored from patch column 1)
    static constexpr int USZ = CK * 16 * BN, PSZ = CK * PH * PW;
    static constexpr int LH = TH / 2 + 2, LW = TW / 2 + 8, LW4 = LW / 4;   // fused upsample: low-res raw patch
    static constexpr int RSZ = CK * LH * LW;
    static constexpr int NPOS = (TH / 2 + 1) * (TW / 2 + 1);
    static constexpr int CG = (2 * NPOS <= 256 && CK >= 2) ? ((4 * NPOS <= 256 && CK >= 4) ? ((8 * NPOS <= 256 && CK >= 8) ? 8 : 4) : 2) : 1;
    static constexpr int CPT = CK / CG;
    static_assert(WN * WTY * WTX == NBLK && (NBLK == 4 || NBLK == 2), "4 waves per workgroup: 4 blocks, or 2 blocks x 2 frequency halves");
    static_assert(GTW == 8 || GTW == 16 || GTW == 32, "tile group is 32x1, 16x2 or 8x4 tiles");
    static_assert(CK % 4 == 0 && USZ % 256 == 0, "even number of k-steps per chunk; filter stage = whole 1-KiB DMA groups");
    static_assert(NPOS <= 256, "fused-upsample expander: one position per thread");
};

template <class C, bool UPS>
struct WLds {
    static constexpr int DSZ = UPS ? C::RSZ : C::PSZ;
    static constexpr int DH = UPS ? C::LH : C::PH, DW4 = UPS ? C::LW4 : C::PW4;
    static constexpr int NGU = C::USZ / 256;                    // 1-KiB groups of filter per chunk
    static constexpr int NDQ = DSZ / 4;                         // 16-byte pieces of activation per chunk
    static constexpr int NGP = (NDQ + 63) / 64;
    static constexpr int NG = NGU + NGP;
    static constexpr int STAGE = NG * 256 + 256;                // floats per stage (+ 1 KiB: the patch lands one float in)
    static constexpr int NIU = (NGU + 3) / 4, NIP = (NGP + 3) / 4, NI = NIU + NIP;   // DMA instructions per wave per chunk
    static constexpr int NMIN = NGU / 4 + NGP / 4;              // ... of the wave that issues fewest (counted vmcnt waits)
    static constexpr int NST = C::NST;                          // stages of the DMA ring
    static constexpr int HIP = NST * STAGE;                     // expanded patch (UPS)
    static constexpr int BYTES = (NST * STAGE + (UPS ? C::PSZ : 0)) * 4;
    static_assert(BYTES <= (C::NBLK == 4 ? 160 : 80) * 1024, "LDS budget (one or two workgroups per CU)");
};

#ifdef SSM_WINO_ABLATE
#define WABL(bit) (p.abl & (bit))
#else
#define WABL(bit) 0
#endif

template <class C, bool UPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_kernel(const WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using L = WLds<C, UPS>;
    constexpr int BN = C::BN, PH = C::PH, PW = C::PW, CK = C::CK;

    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid % C::WN, wty = (wid / C::WN) % C::WTY, wtx = wid / (C::WN * C::WTY);
    const int tyl = l31 / C::GTW, txl = l31 % C::GTW;

    int id = ssm_xcd_tile(blockIdx.x, gridDim.x);
    const int nb = id % p.NB;
    id /= p.NB;
    const int tx = id % p.tilesX;
    id /= p.tilesX;
    const int ty = id % p.tilesY;
    const int b = id / p.tilesY;
    const int x0 = tx * C::TW, y0 = ty * C::TH;

    // The DMA lands the patch ONE FLOAT into its LDS region (16-byte aligned global pieces, 4-byte aligned LDS destination): the 4x4
    // patch of tile tx then starts at the EVEN patch column 2tx+4, so each of its rows is two aligned ds_read_b64 - conflict-free at
    // 256 B/clk, where ds_read2_b32 at a lane stride of two dwords is a 2-way bank conflict on both dwords (8 LDS cycles, not 2).
    const long long porg = UPS ? (long long)(y0 / 2 - 1) * p.sh + (x0 / 2 - 4) : (long long)(y0 - 1) * p.sh + (x0 - 4);
    const float *pbase1 = p.src1 + (long long)b * p.sb1 + porg;
    const float *pbase2 = p.src2 + (long long)b * p.sb2 + porg;
    const float *wbase = p.wpk + (long long)nb * p.Cin * (16 * BN);

    // per-lane source offsets (bytes) of the activation pieces this wave brings per chunk; the filter pieces are linear
    int poff[L::NIP];
#pragma unroll
    for (int i = 0; i < L::NIP; ++i) {
        const int qq = (i * 4 + wid) * 64 + lane;
        if (qq < L::NDQ) {
            const int c = qq / (L::DH * L::DW4);
            const int rem = qq - c * (L::DH * L::DW4);
            const int r = rem / L::DW4;
            const int j = rem - r * L::DW4;
            // overshoot rows / pieces are read from the source's zero frame, never from behind the padded plane (see ssm_wino4.hip)
            const int sH = UPS ? p.hs : p.H, sW = UPS ? p.ws : p.W, sy = UPS ? y0 / 2 - 1 : y0 - 1, sx = UPS ? x0 / 2 : x0;
            const int re = min(r, sH + (SSM_PADY - 1) - sy), fe = min(4 * j, ((sW + 2 * SSM_PADX + 3) & ~3) - 4 - sx);
            poff[i] = ((int)(c * p.sc) + re * p.sh + fe) * 4;
        } else {
            poff[i] = 0;          // tail of the last 1-KiB piece: lands in the stage's padding
        }
    }
    const int uoff = lane * 16;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;

    // k-th DMA instruction of this wave for chunk `ch` into ring stage `stage`: k < NIU filter group 4k + wave, else activation
    auto issue_k = [&](int ch, int stage, int k) {
        const int c0 = ch * CK;
        const unsigned lsb = lds0 + (unsigned)(stage * L::STAGE) * 4u;
        if (k < L::NIU) {
            const int g = 4 * k + wid;
            if (g < L::NGU) {
                const float *base = wbase + (long long)c0 * (16 * BN) + g * 256;
                const unsigned m0v = lsb + (unsigned)g * 1024u;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(uoff), "s"(base), "s"(m0v) : "memory", "m0");
            }
        } else {
            const int kk = k - L::NIU;
            const int g = 4 * kk + wid;
            if (g < L::NGP) {
                const float *base = (c0 < p.C1) ? pbase1 + (long long)c0 * p.sc : pbase2 + (long long)(c0 - p.C1) * p.sc;
                const unsigned m0v = lsb + (unsigned)(L::NGU + g) * 1024u + ((UPS || !WS_B64) ? 0u : 4u);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(poff[kk]), "s"(base), "s"(m0v) : "memory", "m0");
            }
        }
    };
    auto issue = [&](int ch, int stage) {
#pragma unroll
        for (int k = 0; k < L::NI; ++k) issue_k(ch, stage, k);
    };

    f32x16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    // per-lane operand bases (floats) inside a stage: U of (cin = 2cp + half, cout = wn*32 + l31), patch of the lane's tile
    const int aBase = half * (4 * BN) + (wn * 32 + l31);          // in f32x4 units
    const int bBase = (half * (PH * PW) + ((wty * C::GTH + tyl) * 2) * PW + (wtx * C::GTW + txl) * 2 + 3 + WS_B64) / 2;      // f32x2 units

    constexpr int S = CK / 2;                  // k-steps per chunk
    const int nchunks = p.Cin / CK;

    f32x4 a[2][4];
    float d[16], v[2][16];
    // Operand fetches, ONE LDS instruction per call so that the k-loop can place them one per MFMA gap: with a single wave per SIMD
    // nothing else fills the issue slots, and tools/wave1_sched_probe.py shows an LDS instruction costs the in-order stream ~40
    // cycles - one fits under a 64-cycle MFMA, a burst of 12 behind one MFMA stalls the matrix pipe for ~450 cycles per k-step.
    // ai / bi: float index of the k-step's operands inside lds[] (one VGPR each, opaque to the optimizer so that the 4 + 8 reads of a
    // k-step use immediate offsets instead of one address computation per read).
    const f32x4 *lds4 = (const f32x4 *)lds;       // A operands: 16-byte units (ai counts f32x4)
    auto ldA = [&](int ai, int q, int buf) { a[buf][q] = lds4[ai + q * BN]; };
    const f32x2 *lds2 = (const f32x2 *)lds;       // B operands: 8-byte units (bi counts f32x2)
    auto ldB = [&](int bi, int h) {
        const int i = h >> 1, j = (h & 1) * 2;
#if WS_B64
        const f32x2 t2 = lds2[bi + (i * PW + j) / 2];
        float e0 = t2[0], e1 = t2[1];
#else
        float e0 = lds[2 * bi + 1 + i * PW + j], e1 = lds[2 * bi + 1 + i * PW + j + 1];
#endif
        d[4 * i + j] = e0;            // (pinned as scalars where they are consumed: transform_rows)
        d[4 * i + j + 1] = e1;
    };
    auto fetchA = [&](int ai, int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ldA(ai, q, buf);
    };
    auto fetchD = [&](int bi) {
#pragma unroll
        for (int h = 0; h < 8; ++h) ldB(bi, h);
    };
    // V = B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], in two halves of 16 add/sub that go into separate MFMA gaps
    float t[16];
    auto transform_rows = [&]() {
        // the patch values arrive as 8-byte pairs: pin each as a scalar HERE (where the wait for the loads belongs anyway), so that no
        // packed-fp32 arithmetic is formed on the pairs (profiles/DESIGN_history_r1-r3.md 3.3 fence).  Pinning at the load would put an s_waitcnt lgkmcnt(0)
        // behind every ds_read_b64 - the full LDS latency, eight times per k-step.
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(d[i]));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[j] = d[j] - d[8 + j];
            t[4 + j] = d[4 + j] + d[8 + j];
            t[8 + j] = d[8 + j] - d[4 + j];
            t[12 + j] = d[4 + j] - d[12 + j];
        }
    };
    auto transform_cols = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[buf][4 * i] = t[4 * i] - t[4 * i + 2];
            v[buf][4 * i + 1] = t[4 * i + 1] + t[4 * i + 2];
            v[buf][4 * i + 2] = t[4 * i + 2] - t[4 * i + 1];
            v[buf][4 * i + 3] = t[4 * i + 1] - t[4 * i + 3];
        }
        // pin the results here: without this the optimizer sinks each subtraction next to the MFMA that consumes it (a dependent
        // VALU -> MFMA pair with wait states in front of every matrix instruction of the next k-step)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(v[buf][i]));
    };

    // fused upsample: low-res raw chunk -> hi-res patch; one thread = one 2x2 hi-res block position, walking CPT channels (ssm_conv.hip)
    auto expand = [&](const float *stg) {
        if constexpr (UPS) {
            constexpr int PRW = C::TW / 2 + 1, NPOS = C::NPOS, CG = C::CG, CPT = C::CPT;
            constexpr int LH = C::LH, LW = C::LW;
            const float *raw = stg + C::USZ;
            float *hip = lds + L::HIP;
            const int cg = tid / NPOS, pos = tid - cg * NPOS;
            if (cg < CG) {
                const int ly0 = y0 / 2 - 1, lx0 = x0 / 2 - 1;
                const int pi = pos / PRW, pj = pos - pi * PRW;
                const int i = ly0 + pi, j = lx0 + pj;
                const int i0 = min(max(i, 0), p.hs - 1), i1 = min(max(i + 1, 0), p.hs - 1);
                const int j0 = min(max(j, 0), p.ws - 1), j1 = min(max(j + 1, 0), p.ws - 1);
                const float xa = j0 == j1 ? 1.f : 0.75f, xb = j0 == j1 ? 0.f : 0.25f;      // column 2j+1 = xa x[j0] + xb x[j0+1]
                const float ca = j0 == j1 ? 1.f : 0.25f, cb = j0 == j1 ? 0.f : 0.75f;      // column 2j+2 = ca x[j0] + cb x[j0+1]
                const float ya = i0 == i1 ? 1.f : 0.75f, yb = i0 == i1 ? 0.f : 0.25f;
                const int Y = 2 * i + 1, X = 2 * j + 1;
                const bool yt = Y >= 0 && Y < p.H, yb2 = Y + 1 < p.H, xl = X >= 0 && X < p.W, xr = X + 1 < p.W;
                const float m00 = (yt && xl) ? 1.f : 0.f, m01 = (yt && xr) ? 1.f : 0.f, m10 = (yb2 && xl) ? 1.f : 0.f, m11 = (yb2 && xr) ? 1.f : 0.f;
                // all 2 x CPT x 2 low-res values first (immediate offsets off four per-thread bases), then the arithmetic, then the
                // stores: the phase is issue-bound instead of CPT read -> compute -> write latency chains.  The right neighbour is read
                // at j0 + 1 even where the source index is clamped (j1 == j0): its weight xb is then exactly 0 and the value - the
                // zero frame or a neighbouring pixel - is finite.
                const float *r0 = raw + (cg * CPT * LH + (i0 - ly0)) * LW + 3 - lx0 + j0;
                const float *r1 = raw + (cg * CPT * LH + (i1 - ly0)) * LW + 3 - lx0 + j0;
                float *dd = hip + (cg * CPT * PH + 2 * pi) * PW + 2 * pj + 3 + WS_B64;       // hi-res pixel x0 + 2pj - 1 -> patch column 2pj + 4 (3 without the one-float shift)
                float v00[CPT], v01[CPT], v10[CPT], v11[CPT];
#pragma unroll
                for (int cc = 0; cc < CPT; ++cc) {
                    v00[cc] = r0[cc * LH * LW];
                    v01[cc] = r0[cc * LH * LW + 1];
                    v10[cc] = r1[cc * LH * LW];
                    v11[cc] = r1[cc * LH * LW + 1];
                }
#pragma unroll
                for (int cc = 0; cc < CPT; ++cc) {
                    const float h00 = xa * v00[cc] + xb * v01[cc], h01 = ca * v00[cc] + cb * v01[cc];
                    const float h10 = xa * v10[cc] + xb * v11[cc], h11 = ca * v10[cc] + cb * v11[cc];
                    dd[cc * PH * PW] = m00 * (ya * h00 + yb * h10);
                    dd[cc * PH * PW + 1] = m01 * (ya * h01 + yb * h11);
                    dd[cc * PH * PW + PW] = m10 * (yb * h00 + ya * h10);
                    dd[cc * PH * PW + PW + 1] = m11 * (yb * h01 + ya * h11);
                }
            }
            __syncthreads();
        }
    };

    // ---- prologue: chunks 0 and 1 in flight, operands of the first k-step ------------------------------------------
    issue(0, 0);
    if (nchunks > 1) issue(1, 1);
    // Bias: A^T M A adds M[1][1] to all four outputs of the tile, so accumulator 5 starts from the bias - one MFMA k-step with A =
    // the bias column and B = a row of ones (one vector load per lane), issued while the first chunks are still in flight.
    {
        const float bv0 = p.bias[nb * BN + wn * 32 + l31];
        const float ab = half ? 0.f : bv0, ones = half ? 0.f : 1.f;
        acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab, ones, acc[5], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    expand(lds);
    constexpr int PO = UPS ? L::HIP : C::USZ;          // patch offset: inside the stage, or the expanded patch
    fetchA(aBase, 0);
    fetchD(PO / 2 + bBase);
    transform_rows();
    transform_cols(0);

    // Chunk loop.  The barrier sits INSIDE the last k-step of a chunk (after its first MFMA is queued): at that point every LDS
    // read of chunk ch has been issued and has completed (the last k-step's operands are in registers), so the stage of chunk ch is
    // free for the DMA of chunk ch+2, and chunk ch+1 - issued one chunk ago - is waited for and becomes visible to all waves; the
    // operands of its first k-step are then fetched and transformed behind the remaining MFMAs of chunk ch.
#ifndef WS_TR0
#define WS_TR0 13      // MFMA gap that takes the row half of the input transform (16 VALU)
#endif
#ifndef WS_TR1
#define WS_TR1 13      // ... the column half (same gap: the probe prefers the VALU work in ONE gap)
#endif
    for (int ch = 0; ch < nchunks; ++ch) {
        const int so = (ch & 1) * L::STAGE, so_n = ((ch + 1) & 1) * L::STAGE;
        const bool dma = ch + 2 < nchunks && !WABL(1);
        const bool more = ch + 1 < nchunks;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            const bool last = s == S - 1;
            int ai = (last ? so_n : so + (s + 1) * (32 * BN)) / 4 + aBase;
            int bi = ((UPS ? L::HIP : (last ? so_n : so) + C::USZ) + (last ? 0 : (s + 1) * (2 * PH * PW))) / 2 + bBase;
            asm volatile("" : "+v"(ai), "+v"(bi));
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][f >> 2][f & 3], v[cur][f], acc[f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (!last) {
                    if (f < 4) {
                        if (!WABL(8)) ldA(ai, f, nxt);
                    } else if (f < 12) {
                        if (!WABL(4)) ldB(bi, f - 4);
                    }
                    if (!WABL(4)) {
                        if (f == WS_TR0) transform_rows();
                        if (f == WS_TR1) transform_cols(nxt);
                    }
                } else if (more) {
                    if (f == 0 && !WABL(16)) {
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        __syncthreads();
                    }
                    if (f >= 1 && f < 5) {
                        if (!WABL(8)) ldA(ai, f - 1, nxt);
                    } else if (f >= 5 && f < 13) {
                        if constexpr (!UPS) {
                            if (!WABL(4)) ldB(bi, f - 5);
                        }
                    }
                    if (f >= 1) {                                           // DMA of chunk ch+2 into the stage just freed, one per gap
                        constexpr int PER = (L::NI + 14) / 15;
#pragma unroll
                        for (int j = 0; j < PER; ++j) {
                            const int k = (f - 1) * PER + j;
                            if (k < L::NI && dma) issue_k(ch + 2, ch & 1, k);
                        }
                    }
                    if constexpr (!UPS) {
                        if (f == 14 && !WABL(4)) {
                            transform_rows();
                            transform_cols(nxt);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (UPS) {
            if (more) {
                __syncthreads();           // every wave is done with the expanded patch of chunk ch
                expand(lds + so_n);
                fetchD(L::HIP / 2 + bBase);
                transform_rows();
                transform_cols(0);
            }
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------
    const int px = x0 + (wtx * C::GTW + txl) * 2, py = y0 + (wty * C::GTH + tyl) * 2;
#ifdef SSM_WINO_ABLATE
    if ((p.abl & 2) && acc[0][0] != 12345.678f) return;
#endif
    const float sl = (p.lrelu & 1) ? p.slope : 1.f;
    const bool amask = (p.lrelu & 2) != 0;          // SSM_FLAG_MASK: the addend view is a mask source (see ssm_hip.h)
    float *dstb = p.dst + (long long)b * p.dsb;
    float *poolb = p.pool ? p.pool + (long long)b * p.psb : nullptr;
    const int cu0 = nb * BN + wn * 32;
    const bool full = cu0 + 32 <= p.Cout;
    const unsigned pb0 = 4u * ((unsigned)(4 * half) * (unsigned)p.dsc + (unsigned)py * (unsigned)p.dsh + (unsigned)px);
    const unsigned pb1 = pb0 + 4u * (unsigned)p.dsh;
    const unsigned qb = 4u * ((unsigned)(4 * half) * (unsigned)p.psc + (unsigned)(py >> 1) * (unsigned)p.psh + (unsigned)(px >> 1));
    const bool ok0 = py < p.H && px < p.W, ok1 = py + 1 < p.H && px < p.W, x1ok = px + 1 < p.W;
    auto st2 = [](const float *base, unsigned off_bytes, f32x2 val) {
        asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off_bytes), "v"(val), "s"(base) : "memory");
    };
    auto st1 = [](const float *base, unsigned off_bytes, float val) {
        asm volatile("global_store_dword %0, %1, %2" ::"v"(off_bytes), "v"(val), "s"(base) : "memory");
    };
    const float *addb = p.add ? p.add + (long long)(b / p.adiv) * p.asb + (long long)(4 * half) * p.asc + (long long)py * p.ash + px : nullptr;
    auto store_all = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cu = cu0 + (r & 3) + 8 * (r >> 2);      // uniform; this lane's cout = cu + 4*half
            const bool cok = FULL || cu + 4 * half < p.Cout;
            f32x2 ad0 = {0.f, 0.f}, ad1 = {0.f, 0.f};
            if (addb) {
                if (ok0 && cok) ad0 = *(const f32x2 *)(addb + (long long)cu * p.asc);
                if (ok1 && cok) ad1 = *(const f32x2 *)(addb + (long long)cu * p.asc + p.ash);
            }
            // Y = A^T M A,  A^T = [1 1 1 0; 0 1 -1 -1]
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] = acc[j][r] + acc[4 + j][r] + acc[8 + j][r];
                s1[j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
            }
            float y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
            float y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
            if (amask) {          // y * LeakyReLU'(m): the data gradient of a layer leaves as dZ of the layer below (training step)
                y00 *= ad0[0] > 0.f ? 1.f : p.slope;
                y01 *= ad0[1] > 0.f ? 1.f : p.slope;
                y10 *= ad1[0] > 0.f ? 1.f : p.slope;
                y11 *= ad1[1] > 0.f ? 1.f : p.slope;
            } else {
                y00 += ad0[0];
                y01 += ad0[1];
                y10 += ad1[0];
                y11 += ad1[1];
            }
            y00 = fmaxf(y00, y00 * sl);
            y01 = fmaxf(y01, y01 * sl);
            y10 = fmaxf(y10, y10 * sl);
            y11 = fmaxf(y11, y11 * sl);
            float *bp = dstb + (long long)cu * p.dsc;
            f32x2 r0 = {y00, y01}, r1 = {y10, y11};
            if (ok0 && cok) {
                if (x1ok) st2(bp, pb0, r0);
                else st1(bp, pb0, y00);          // odd map width: the tile's second column is the zero frame - never written
            }
            if (ok1 && cok) {
                if (x1ok) st2(bp, pb1, r1);
                else st1(bp, pb1, y10);
            }
            if (poolb) {
                float *qp = poolb + (long long)cu * p.psc;
                const float sm = ((y00 + y10) + (y01 + y11)) * 0.25f;
                if (ok1 && cok) st1(qp, qb, sm);
            }
            __builtin_amdgcn_sched_barrier(0);      // one cout at a time: 16 accumulator reads live, not 256
        }
    };
    if (full) store_all(std::true_type{});
    else store_all(std::false_type{});
}


// =====================================================================================================================
// Second form: TWO workgroups per CU.  A wave owns 8 of the 16 frequencies (rows 2h, 2h+1 of the 4x4 frequency grid, h = wave & 1)
// of one 32-cout x 32-tile block: 8 accumulators = 128 registers, so two waves share a SIMD and the prologue (first DMA), the
// barriers, the fused-upsample expansion and the epilogue of one workgroup hide behind the MFMAs of the other - what the first form,
// alone on its CU, exposes (tools/r4_abl.sh: 9 us of overhead per workgroup on a 32-channel layer against 7 us of MFMA).  Per k-step a
// wave fetches 2 (not 4) filter quads and 3 (not 4) patch rows and does half of the input transform; the two halves of the output
// transform meet through LDS once per tile: Y rows = (M0 + M1 | M1) + (M2 | -M2 - M3).
template <class C, bool UPS, int FH>
__device__ __forceinline__ void wino2_body(const WinoParams &p, float *lds) {
    using L = WLds<C, UPS>;
    constexpr int BN = C::BN, PH = C::PH, PW = C::PW, CK = C::CK;
    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wb = wid >> 1;                                          // block of this wave pair
    const int wn = wb % C::WN, wty = (wb / C::WN) % C::WTY, wtx = wb / (C::WN * C::WTY);
    const int tyl = l31 / C::GTW, txl = l31 % C::GTW;

    int id = ssm_xcd_tile(blockIdx.x, gridDim.x);
    const int ks = id % p.KS;          // (split-K partners are neighbours in the grid: they read the same patches)
    id /= p.KS;
    const int nb = id % p.NB;
    id /= p.NB;
    const int tx = id % p.tilesX;
    id /= p.tilesX;
    const int ty = id % p.tilesY;
    const int b = id / p.tilesY;
    const int x0 = tx * C::TW, y0 = ty * C::TH;

    // The DMA lands the patch ONE FLOAT into its LDS region (16-byte aligned global pieces, 4-byte aligned LDS destination): the 4x4
    // patch of tile tx then starts at the EVEN patch column 2tx+4, so each of its rows is two aligned ds_read_b64 - conflict-free at
    // 256 B/clk, where ds_read2_b32 at a lane stride of two dwords is a 2-way bank conflict on both dwords (8 LDS cycles, not 2).
    const long long porg = UPS ? (long long)(y0 / 2 - 1) * p.sh + (x0 / 2 - 4) : (long long)(y0 - 1) * p.sh + (x0 - 4);
    const float *pbase1 = p.src1 + (long long)b * p.sb1 + porg;
    const float *pbase2 = p.src2 + (long long)b * p.sb2 + porg;
    const float *wbase = p.wpk + (long long)nb * p.Cin * (16 * BN);

    int poff[L::NIP];
#pragma unroll
    for (int i = 0; i < L::NIP; ++i) {
        const int qq = (i * 4 + wid) * 64 + lane;
        if (qq < L::NDQ) {
            const int c = qq / (L::DH * L::DW4);
            const int rem = qq - c * (L::DH * L::DW4);
            const int r = rem / L::DW4;
            const int j = rem - r * L::DW4;
            // overshoot rows / pieces are read from the source's zero frame, never from behind the padded plane (see ssm_wino4.hip)
            const int sH = UPS ? p.hs : p.H, sW = UPS ? p.ws : p.W, sy = UPS ? y0 / 2 - 1 : y0 - 1, sx = UPS ? x0 / 2 : x0;
            const int re = min(r, sH + (SSM_PADY - 1) - sy), fe = min(4 * j, ((sW + 2 * SSM_PADX + 3) & ~3) - 4 - sx);
            poff[i] = ((int)(c * p.sc) + re * p.sh + fe) * 4;
        } else {
            poff[i] = 0;
        }
    }
    const int uoff = lane * 16;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
    const int nchunks = p.Cin / CK / p.KS;
    const int cbeg = ks * nchunks * CK;          // first input channel of this workgroup
    auto issue_k = [&](int ch, int stage, int k) {
        const int c0 = cbeg + ch * CK;
        const unsigned lsb = lds0 + (unsigned)(stage * L::STAGE) * 4u;
        if (k < L::NIU) {
            const int g = 4 * k + wid;
            if (g < L::NGU) {
                const float *base = wbase + (long long)c0 * (16 * BN) + g * 256;
                const unsigned m0v = lsb + (unsigned)g * 1024u;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(uoff), "s"(base), "s"(m0v) : "memory", "m0");
            }
        } else {
            const int kk = k - L::NIU;
            const int g = 4 * kk + wid;
            if (g < L::NGP) {
                const float *base = (c0 < p.C1) ? pbase1 + (long long)c0 * p.sc : pbase2 + (long long)(c0 - p.C1) * p.sc;
                const unsigned m0v = lsb + (unsigned)(L::NGU + g) * 1024u + ((UPS || !WS_B64) ? 0u : 4u);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(poff[kk]), "s"(base), "s"(m0v) : "memory", "m0");
            }
        }
    };
    auto issue = [&](int ch, int stage) {
#pragma unroll
        for (int k = 0; k < L::NI; ++k) issue_k(ch, stage, k);
    };

    f32x16 acc[8];          // local frequency fl <-> grid position (row 2*FH + fl/4, column fl%4)
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    const int aBase = half * (4 * BN) + (wn * 32 + l31) + FH * 2 * BN;           // f32x4 units; filter quads 2FH, 2FH+1
    const int bBase = (half * (PH * PW) + ((wty * C::GTH + tyl) * 2 + FH) * PW + (wtx * C::GTW + txl) * 2 + 3 + WS_B64) / 2;   // f32x2 units; patch rows FH .. FH+2

    constexpr int S = CK / 2;
    const f32x4 *lds4 = (const f32x4 *)lds;
    f32x4 a[2][2];
    float d[12], t[8], v[2][8];
    auto ldA = [&](int ai, int q, int buf) { a[buf][q] = lds4[ai + q * BN]; };
    const f32x2 *lds2 = (const f32x2 *)lds;       // B operands: 8-byte units (bi counts f32x2)
    auto ldB = [&](int bi, int h) {
        const int i = h >> 1, j = (h & 1) * 2;
#if WS_B64
        const f32x2 t2 = lds2[bi + (i * PW + j) / 2];
        float e0 = t2[0], e1 = t2[1];
#else
        float e0 = lds[2 * bi + 1 + i * PW + j], e1 = lds[2 * bi + 1 + i * PW + j + 1];
#endif
        d[4 * i + j] = e0;            // (pinned as scalars where they are consumed: transform_rows)
        d[4 * i + j + 1] = e1;
    };
    // rows 2FH, 2FH+1 of B^T d from patch rows e0, e1, e2 = d rows FH, FH+1, FH+2
    auto transform_rows = [&]() {
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("" : "+v"(d[i]));      // scalars from here on (see wino_kernel)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (FH == 0) {
                t[j] = d[j] - d[8 + j];          // d0 - d2
                t[4 + j] = d[4 + j] + d[8 + j];  // d1 + d2
            } else {
                t[j] = d[4 + j] - d[j];          // d2 - d1
                t[4 + j] = d[j] - d[8 + j];      // d1 - d3
            }
        }
    };
    auto transform_cols = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v[buf][4 * i] = t[4 * i] - t[4 * i + 2];
            v[buf][4 * i + 1] = t[4 * i + 1] + t[4 * i + 2];
            v[buf][4 * i + 2] = t[4 * i + 2] - t[4 * i + 1];
            v[buf][4 * i + 3] = t[4 * i + 1] - t[4 * i + 3];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(v[buf][i]));
    };

    auto expand = [&](const float *stg) {
        if constexpr (UPS) {
            constexpr int PRW = C::TW / 2 + 1, NPOS = C::NPOS, CG = C::CG, CPT = C::CPT;
            constexpr int LH = C::LH, LW = C::LW;
            const float *raw = stg + C::USZ;
            float *hip = lds + L::HIP;
            const int cg = tid / NPOS, pos = tid - cg * NPOS;
            if (cg < CG) {
                const int ly0 = y0 / 2 - 1, lx0 = x0 / 2 - 1;
                const int pi = pos / PRW, pj = pos - pi * PRW;
                const int i = ly0 + pi, j = lx0 + pj;
                const int i0 = min(max(i, 0), p.hs - 1), i1 = min(max(i + 1, 0), p.hs - 1);
                const int j0 = min(max(j, 0), p.ws - 1), j1 = min(max(j + 1, 0), p.ws - 1);
                const float xa = j0 == j1 ? 1.f : 0.75f, xb = j0 == j1 ? 0.f : 0.25f;      // column 2j+1 = xa x[j0] + xb x[j0+1]
                const float ca = j0 == j1 ? 1.f : 0.25f, cb = j0 == j1 ? 0.f : 0.75f;      // column 2j+2 = ca x[j0] + cb x[j0+1]
                const float ya = i0 == i1 ? 1.f : 0.75f, yb = i0 == i1 ? 0.f : 0.25f;
                const int Y = 2 * i + 1, X = 2 * j + 1;
                const bool yt = Y >= 0 && Y < p.H, yb2 = Y + 1 < p.H, xl = X >= 0 && X < p.W, xr = X + 1 < p.W;
                const float m00 = (yt && xl) ? 1.f : 0.f, m01 = (yt && xr) ? 1.f : 0.f, m10 = (yb2 && xl) ? 1.f : 0.f, m11 = (yb2 && xr) ? 1.f : 0.f;
                const float *r0 = raw + (cg * CPT * LH + (i0 - ly0)) * LW + 3 - lx0 + j0;
                const float *r1 = raw + (cg * CPT * LH + (i1 - ly0)) * LW + 3 - lx0 + j0;
                float *dd = hip + (cg * CPT * PH + 2 * pi) * PW + 2 * pj + 3 + WS_B64;       // hi-res pixel x0 + 2pj - 1 -> patch column 2pj + 4 (3 without the one-float shift)
                float v00[CPT], v01[CPT], v10[CPT], v11[CPT];
#pragma unroll
                for (int cc = 0; cc < CPT; ++cc) {
                    v00[cc] = r0[cc * LH * LW];
                    v01[cc] = r0[cc * LH * LW + 1];
                    v10[cc] = r1[cc * LH * LW];
                    v11[cc] = r1[cc * LH * LW + 1];
                }
#pragma unroll
                for (int cc = 0; cc < CPT; ++cc) {
                    const float h00 = xa * v00[cc] + xb * v01[cc], h01 = ca * v00[cc] + cb * v01[cc];
                    const float h10 = xa * v10[cc] + xb * v11[cc], h11 = ca * v10[cc] + cb * v11[cc];
                    dd[cc * PH * PW] = m00 * (ya * h00 + yb * h10);
                    dd[cc * PH * PW + 1] = m01 * (ya * h01 + yb * h11);
                    dd[cc * PH * PW + PW] = m10 * (yb * h00 + ya * h10);
                    dd[cc * PH * PW + PW + 1] = m11 * (yb * h01 + ya * h11);
                }
            }
            __syncthreads();
        }
    };

    // DMA ring of NST stages: chunk c lives in stage c % NST; the first NST chunks are issued here, chunk c + NST at the barrier inside
    // chunk c.  Waits are COUNTED: `s_waitcnt vmcnt(k * NMIN)` leaves the k newest chunks' DMAs in flight (every wave issues at least
    // NMIN instructions per chunk, vmcnt retires in order), so with NST > 2 a chunk has NST - 1 chunk times to land instead of one -
    // on boxes / layers where the L2 / MALL answers slowly the double buffer loses 10 %.
    auto wait_newer = [&](int k) {        // wait until at most the k newest chunks of this wave's DMAs are outstanding (k uniform)
        if (k <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::NMIN) : "memory");
        else if (k == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L::NMIN) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * L::NMIN) : "memory");
    };
    // The counted waits rely on the chunk loop issuing NO other VMEM instruction (global load / store) than these DMAs: any such
    // instruction would shift the vmcnt positions.  Stores happen only in the epilogue, after the last wait.
    static_assert(L::NST >= 2 && L::NST <= 5 && 3 * L::NMIN <= 63, "ring depth / vmcnt range");
    // the epilogue reuses the DMA stages (from lds[0]) as the exchange buffer of the wave pairs: [2 blocks][16 registers][64 lanes]
    // quads = 32 KiB; with UPS the expanded patch starts right behind the stages (L::HIP) and must not be overrun
    static_assert(C::NBLK == 4 || L::NST * L::STAGE * 4 >= 2 * 16 * 64 * 16, "exchange buffer does not fit in the DMA stages");
#pragma unroll
    for (int c = 0; c < L::NST; ++c)
        if (c < nchunks) issue(c, c);
    if (FH == 0 && ks == 0) {        // bias: accumulator of frequency (1,1) = local 5 of the first half
        const float bv0 = p.bias[nb * BN + wn * 32 + l31];
        const float ab = half ? 0.f : bv0, ones = half ? 0.f : 1.f;
        acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab, ones, acc[5], 0, 0, 0);
    }
    wait_newer(min(nchunks, (int)L::NST) - 1 > 3 ? 3 : min(nchunks, (int)L::NST) - 1);      // chunk 0 has landed
    __syncthreads();
    expand(lds);
    constexpr int PO = UPS ? L::HIP : C::USZ;
#pragma unroll
    for (int q = 0; q < 2; ++q) ldA(aBase, q, 0);
#pragma unroll
    for (int h = 0; h < 6; ++h) ldB(PO / 2 + bBase, h);
    transform_rows();
    transform_cols(0);

    for (int ch = 0; ch < nchunks; ++ch) {
        const int st = ch % L::NST;
        const int so = st * L::STAGE, so_n = (st + 1 == L::NST ? 0 : st + 1) * L::STAGE;
        const bool dma = ch + L::NST < nchunks && !WABL(1);
        const bool more = ch + 1 < nchunks;
        // chunks newer than ch+1 whose DMAs may stay in flight across this chunk's barrier
        const int newer = min((int)L::NST - 2, nchunks - 2 - ch);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            const bool last = s == S - 1;
            int ai = (last ? so_n : so + (s + 1) * (32 * BN)) / 4 + aBase;
            int bi = ((UPS ? L::HIP : (last ? so_n : so) + C::USZ) + (last ? 0 : (s + 1) * (2 * PH * PW))) / 2 + bBase;
            asm volatile("" : "+v"(ai), "+v"(bi));
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][f >> 2][f & 3], v[cur][f], acc[f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#ifndef WS_BFIRST
#define WS_BFIRST 1        // patch rows first (gaps 0-2), filter quads after them: the transform in gaps 6-7 finds its operands landed
#endif
                if (!last) {
                    if (WS_BFIRST) {
                        if (f < 3) {
                            if (!WABL(4)) {
                                ldB(bi, 2 * f);
                                ldB(bi, 2 * f + 1);
                            }
                        } else if (f < 5) {
                            if (!WABL(8)) ldA(ai, f - 3, nxt);
                        }
                    } else {
                        if (f < 2) ldA(ai, f, nxt);
                        else if (f < 5) {
                            ldB(bi, 2 * (f - 2));
                            ldB(bi, 2 * (f - 2) + 1);
                        }
                    }
                    if (!WABL(4)) {
                        if (f == 6) transform_rows();
                        if (f == 7) transform_cols(nxt);
                    }
                } else if (more) {
                    if (f == 0 && !WABL(16)) {
                        wait_newer(newer > 3 ? 3 : newer);                  // chunk ch+1 has landed (this wave's share of it)
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __syncthreads();
                    }
                    if (WS_BFIRST) {
                        if (f >= 1 && f < 4) {
                            if constexpr (!UPS) {
                                if (!WABL(4)) {
                                    ldB(bi, 2 * (f - 1));
                                    ldB(bi, 2 * (f - 1) + 1);
                                }
                            }
                        } else if (f >= 4 && f < 6) {
                            if (!WABL(8)) ldA(ai, f - 4, nxt);
                        }
                    } else {
                        if (f >= 1 && f < 3) ldA(ai, f - 1, nxt);
                        else if (f >= 3 && f < 6) {
                            if constexpr (!UPS) {
                                ldB(bi, 2 * (f - 3));
                                ldB(bi, 2 * (f - 3) + 1);
                            }
                        }
                    }
                    if (f >= 1) {
                        constexpr int PER = (L::NI + 6) / 7;
#pragma unroll
                        for (int j = 0; j < PER; ++j) {
                            const int k = (f - 1) * PER + j;
                            if (k < L::NI && dma) issue_k(ch + L::NST, st, k);
                        }
                    }
                    if constexpr (!UPS) {
                        if (f == 7 && !WABL(4)) {
                            transform_rows();
                            transform_cols(nxt);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (UPS) {
            if (more) {
                __syncthreads();
                expand(lds + so_n);
#pragma unroll
                for (int h = 0; h < 6; ++h) ldB(L::HIP / 2 + bBase, h);
                transform_rows();
                transform_cols(0);
            }
        }
    }

    // ---- epilogue: column half of A^T M A per wave, the two row halves meet through LDS ------------------------------------------
#ifdef SSM_WINO_ABLATE
    if ((p.abl & 2) && acc[0][0] != 12345.678f) return;
#endif
    __syncthreads();                               // every LDS read of the last chunk is complete: the stages become the exchange buffer
    // Each wave turns its two frequency rows into its share of the two output rows of every cout: 4 values per accumulator register r,
    //   half 0 (rows 0, 1 of M): (c0 + c1 | c1),   half 1 (rows 2, 3): (c2 | -c2 - c3),   c_i = (m_i0 + m_i1 + m_i2 | m_i1 - m_i2 - m_i3),
    // and the two shares add up to Y.  The work after the exchange (sum, addend, activation, stores) is split between the two waves of a
    // pair: half 0 finishes registers 0..7, half 1 registers 8..15 - each wave sends the shares of the OTHER wave's registers through
    // LDS.  (One wave finishing all 16 while its partner idles made the epilogue twice as long; on the 32-channel full-resolution
    // layers the epilogue is a quarter of the workgroup's time.)
    auto share = [&](int r) -> f32x4 {
        const float ca0 = acc[0][r] + acc[1][r] + acc[2][r], ca1 = acc[1][r] - acc[2][r] - acc[3][r];
        const float cb0 = acc[4][r] + acc[5][r] + acc[6][r], cb1 = acc[5][r] - acc[6][r] - acc[7][r];
        if constexpr (FH == 0) return f32x4{ca0 + cb0, ca1 + cb1, cb0, cb1};
        else return f32x4{ca0, ca1, -ca0 - cb0, -ca1 - cb1};
    };
    constexpr int RO = FH == 0 ? 0 : 8;             // registers this wave finishes; it sends the other eight
    f32x4 *xb4 = (f32x4 *)lds + (wb * 16) * 64 + lane;       // [block][r][lane] quads
#pragma unroll
    for (int r = 8 - RO; r < 16 - RO; ++r) {
        xb4[r * 64] = share(r);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    {
        const int px = x0 + (wtx * C::GTW + txl) * 2, py = y0 + (wty * C::GTH + tyl) * 2;
        const float sl = (p.lrelu & 1) ? p.slope : 1.f;
        const bool amask = (p.lrelu & 2) != 0;          // SSM_FLAG_MASK: the addend view is a mask source (see ssm_hip.h)
        float *dstb = p.dst + (long long)(b + ks * p.ksB) * p.dsb;
        float *poolb = p.pool ? p.pool + (long long)b * p.psb : nullptr;
        const int cu0 = nb * BN + wn * 32;
        const bool full = cu0 + 32 <= p.Cout;
        const unsigned pb0 = 4u * ((unsigned)(4 * half) * (unsigned)p.dsc + (unsigned)py * (unsigned)p.dsh + (unsigned)px);
        const unsigned pb1 = pb0 + 4u * (unsigned)p.dsh;
        const unsigned qb = 4u * ((unsigned)(4 * half) * (unsigned)p.psc + (unsigned)(py >> 1) * (unsigned)p.psh + (unsigned)(px >> 1));
        const bool ok0 = py < p.H && px < p.W, ok1 = py + 1 < p.H && px < p.W, x1ok = px + 1 < p.W;
        auto st2 = [](const float *base, unsigned off_bytes, f32x2 val) {
            asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off_bytes), "v"(val), "s"(base) : "memory");
        };
        auto st1 = [](const float *base, unsigned off_bytes, float val) {
            asm volatile("global_store_dword %0, %1, %2" ::"v"(off_bytes), "v"(val), "s"(base) : "memory");
        };
        const float *addb = p.add ? p.add + (long long)(b / p.adiv) * p.asb + (long long)(4 * half) * p.asc + (long long)py * p.ash + px : nullptr;
        auto store_all = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
            for (int r = RO; r < RO + 8; ++r) {
                const int cu = cu0 + (r & 3) + 8 * (r >> 2);
                const bool cok = FULL || cu + 4 * half < p.Cout;
                f32x2 ad0 = {0.f, 0.f}, ad1 = {0.f, 0.f};
                if (addb) {
                    if (ok0 && cok) ad0 = *(const f32x2 *)(addb + (long long)cu * p.asc);
                    if (ok1 && cok) ad1 = *(const f32x2 *)(addb + (long long)cu * p.asc + p.ash);
                }
                const f32x4 q = xb4[r * 64];
                const f32x4 m = share(r);
                float y00 = m[0] + q[0], y01 = m[1] + q[1];
                float y10 = m[2] + q[2], y11 = m[3] + q[3];
                if (amask) {
                    y00 *= ad0[0] > 0.f ? 1.f : p.slope;
                    y01 *= ad0[1] > 0.f ? 1.f : p.slope;
                    y10 *= ad1[0] > 0.f ? 1.f : p.slope;
                    y11 *= ad1[1] > 0.f ? 1.f : p.slope;
                } else {
                    y00 += ad0[0];
                    y01 += ad0[1];
                    y10 += ad1[0];
                    y11 += ad1[1];
                }
                y00 = fmaxf(y00, y00 * sl);
                y01 = fmaxf(y01, y01 * sl);
                y10 = fmaxf(y10, y10 * sl);
                y11 = fmaxf(y11, y11 * sl);
                float *bp = dstb + (long long)cu * p.dsc;
                f32x2 r0 = {y00, y01}, r1 = {y10, y11};
                if (ok0 && cok) {
                    if (x1ok) st2(bp, pb0, r0);
                    else st1(bp, pb0, y00);          // odd map width: the tile's second column is the zero frame - never written
                }
                if (ok1 && cok) {
                    if (x1ok) st2(bp, pb1, r1);
                    else st1(bp, pb1, y10);
                }
                if (poolb) {
                    float *qp = poolb + (long long)cu * p.psc;
                    const float sm = ((y00 + y10) + (y01 + y11)) * 0.25f;
                    if (ok1 && cok) st1(qp, qb, sm);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (full) store_all(std::true_type{});
        else store_all(std::false_type{});
    }
}

template <class C, bool UPS>
__global__ __launch_bounds__(256, 2) void wino2_kernel(const WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (A first-round stagger of the two workgroups that share a CU - half a workgroup's matrix time of s_sleep for workgroups 256..511 -
    // was measured on the same box and changed nothing: 195.2 vs 196.4 TFLOP/s over the 3x3 layers.)
    // the frequency half is wave-uniform: both bodies contain the same sequence of barriers
    if ((__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) == 0) wino2_body<C, UPS, 0>(p, lds);
    else wino2_body<C, UPS, 1>(p, lds);
}

// ---- tile configurations ------------------------------------------------------------------------------------------
//                     WN WTY WTX GTW CK        BN   TH  TW (pixels)
using W64A = WCfg<2, 2, 1, 32, 8>;        //    64    4  64
using W32A = WCfg<1, 4, 1, 32, 8>;        //    32    8  64
using W128A = WCfg<4, 1, 1, 32, 4>;       //   128    2  64
using W64G = WCfg<2, 2, 1, 8, 8>;         //    64   16  16
using W128G = WCfg<4, 1, 1, 8, 4>;        //   128    8  16
using W64H = WCfg<2, 2, 1, 16, 8>;        //    64    8  32
using W128H = WCfg<4, 1, 1, 16, 4>;       //   128    4  32
// two workgroups per CU (wino2_kernel): 2 blocks x 2 frequency halves
using V32A = WCfg<1, 2, 1, 32, 8, 2>;     //    32    4  64
using V32H = WCfg<1, 2, 1, 16, 8, 2>;     //    32    8  32
using V32G = WCfg<1, 2, 1, 8, 8, 2>;      //    32   16  16
using V64A = WCfg<2, 1, 1, 32, 4, 2>;     //    64    2  64
using V64G = WCfg<2, 1, 1, 8, 4, 2>;      //    64    8  16

#define SSM_WINO_KINDS(X)                                                                                          \
    X(W64A_, W64A) X(W32A_, W32A) X(W128A_, W128A) X(W64G_, W64G) X(W128G_, W128G) X(W64H_, W64H) X(W128H_, W128H) \
    X(V32A_, V32A) X(V32H_, V32H) X(V32G_, V32G) X(V64A_, V64A) X(V64G_, V64G)

enum WinoKind {
#define X(name, cfg) name,
    SSM_WINO_KINDS(X)
#undef X
        NWKIND
};

struct WKindInfo {
    int bn, th, tw, ck, nblk, nst;
};

template <class C>
constexpr WKindInfo winfo_of() {
    return WKindInfo{C::BN, C::TH, C::TW, C::CK, C::NBLK, C::NST};
}

constexpr WKindInfo kWInfo[NWKIND] = {
#define X(name, cfg) winfo_of<cfg>(),
    SSM_WINO_KINDS(X)
#undef X
};

std::atomic<int> g_force_wkind{-1};

// Estimated duration (cycles) of a launch.  The matrix work of one CU-round is mf = Cin/2 k-steps x 16 MFMAs x 64 cycles: one
// workgroup of the first form (all 16 frequencies per wave, 256 resident workgroups), or two co-resident workgroups of the second
// form (8 frequencies per wave, 512 resident).  Around it: an efficiency factor (operand fetches and transforms share the issue
// stream with the MFMAs), a fixed cost per round (prologue + epilogue: exposed in the first form, mostly hidden behind the
// neighbour in the second), the fused-upsample expansion per chunk, and whole rounds only (all workgroups of a launch are equal).
// Constants fitted to sweeps of every configuration over the layer shapes at batch 2 and 7 (tools/r4_sweep.sh; the picks are
// within 0.4 % of the per-layer best in sum, 3 % at worst).
double estimate_wino(const WKindInfo &ki, int Cin, int Cout, int B, int H, int W, int ups) {
    const long long tiles = (long long)B * ((W + ki.tw - 1) / ki.tw) * ((H + ki.th - 1) / ki.th);
    const long long nwg = tiles * ((Cout + ki.bn - 1) / ki.bn);
    const double mf = (double)(Cin / 2) * 16.0 * 64.0;
    const double chunks = (double)Cin / ki.ck;
    if (ki.nblk == 4) return (double)((nwg + 255) / 256) * (mf * 1.25 + 22000.0 + (ups ? 900.0 * chunks : 0.0));
    // (the 8x32-pixel tile measures ~1 % ahead of the 4x64 one where both divide the map: shorter patch rows per DMA piece)
    const double per = (mf * 1.3 + 6000.0 + (ups ? 500.0 * chunks : 0.0)) * ((ki.th == 8 && ki.tw == 32) ? 0.995 : 1.0);
    const long long full = nwg / 512, rem = nwg % 512;
    double t = (double)full * per;
    if (rem) t += rem > 256 ? per : mf * 0.5 * 1.25 + 16000.0 + (ups ? 500.0 * chunks : 0.0);     // a last round of lone workgroups
    return t;
}

int pick_wkind(int Cin, int Cout, int B, int H, int W, int ups) {
    const int forced = g_force_wkind.load();
    if (forced >= 0 && forced < NWKIND) return forced;
    int best = -1;
    double bt = 0.0;
    for (int i = 0; i < NWKIND; ++i) {
        const WKindInfo &ki = kWInfo[i];
        if (ki.bn > 32 && ki.bn / 2 >= ((Cout + 31) / 32) * 32) continue;      // over half of the cout block would be padding
        if (Cin % ki.ck) continue;
        const double t = estimate_wino(ki, Cin, Cout, B, H, W, ups);
        if (best < 0 || t < bt * 0.999) {
            best = i;
            bt = t;
        }
    }
    return best;
}

template <class C, bool UPS>
int wlaunch(WinoParams &p, int B, hipStream_t st) {
    p.tilesX = (p.W + C::TW - 1) / C::TW;
    p.tilesY = (p.H + C::TH - 1) / C::TH;
    p.NB = (p.Cout + C::BN - 1) / C::BN;
    const long long blocks = (long long)p.tilesX * p.tilesY * p.NB * B * p.KS;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        ssm::set_error("wino conv: grid of %lld workgroups out of range", blocks);
        return SSM_E_ARG;
    }
    constexpr int lds_bytes = WLds<C, UPS>::BYTES;
    void (*kern)(const WinoParams);
    if constexpr (C::NBLK == 4) kern = wino_kernel<C, UPS>;
    else kern = wino2_kernel<C, UPS>;
    static std::atomic<uint64_t> lds_reserved{0};          // one bit per device: the attribute is per (kernel, device)
    const hipError_t attr_rc = ssm::reserve_lds(lds_reserved, (const void *)kern, lds_bytes);
    if (attr_rc != hipSuccess) {
        ssm::set_error("wino conv: cannot reserve %d bytes of LDS: %s", lds_bytes, hipGetErrorString(attr_rc));
        return SSM_E_LAUNCH;
    }
    SSM_LAUNCH(kern, dim3((unsigned)blocks), dim3(256), lds_bytes, st, p);
    return ssm::check_launch(UPS ? "ssm_wino_conv2d_ups_fwd" : "ssm_wino_conv2d_fwd");
}

template <bool UPS>
int wdispatch(int kind, WinoParams &p, int B, hipStream_t st) {
    switch (kind) {
#define X(name, cfg) \
    case name: return wlaunch<cfg, UPS>(p, B, st);
        SSM_WINO_KINDS(X)
#undef X
    }
    return SSM_E_UNSUPPORTED;
}

// U = G g G^T,  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]; packed index -> (nb, cin, q, n, e), frequency f = 4q + e = 4i + j
__global__ void wino_pack_kernel(const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ wp,
                                 float *__restrict__ bp, int Cout, int Cin, int BN, long long total, int nbias) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        long long r = i;
        const int e = (int)(r % 4);
        r /= 4;
        const int n = (int)(r % BN);
        r /= BN;
        const int q = (int)(r % 4);
        r /= 4;
        const int cin = (int)(r % Cin);
        const int nb = (int)(r / Cin);
        const int co = nb * BN + n;
        float val = 0.f;
        if (co < Cout) {
            const float *g = w + ((long long)co * Cin + cin) * 9;
            float row[3];      // row q of G g
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float g0 = g[c], g1 = g[3 + c], g2 = g[6 + c];
                row[c] = q == 0 ? g0 : (q == 1 ? 0.5f * (g0 + g1 + g2) : (q == 2 ? 0.5f * (g0 - g1 + g2) : g2));
            }
            val = e == 0 ? row[0] : (e == 1 ? 0.5f * (row[0] + row[1] + row[2]) : (e == 2 ? 0.5f * (row[0] - row[1] + row[2]) : row[2]));
        }
        wp[i] = val;
    }
    if (i < nbias) bp[i] = (i < Cout) ? bias[i] : 0.f;
}

int wfill(WinoParams &p, ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed, ssm_view y, int H,
          int W, int Cout, float slope, int flags, int CK, int srcW) {
    SSM_REQUIRE(H > 0 && W > 0 && Cout > 0 && C1 > 0 && C2 >= 0, "wino conv: bad sizes");
    SSM_REQUIRE(x1.ptr && y.ptr && w_packed && bias_packed, "wino conv: null pointer");
    SSM_REQUIRE(C1 % CK == 0 && C2 % CK == 0, "wino conv: channel counts (%d,%d) must be multiples of %d", C1, C2, CK);
    SSM_REQUIRE(ssm::aligned16(x1.ptr) && x1.sh % 4 == 0 && x1.sc % 4 == 0 && x1.sb % 4 == 0,
                "wino conv: input 1 is not a padded-plane view (16-byte alignment)");
    SSM_REQUIRE(x1.sh >= srcW + 2 * SSM_PADX, "wino conv: input 1 row stride %d leaves no zero frame for W=%d", x1.sh, srcW);
    SSM_REQUIRE(ssm::aligned16(w_packed), "wino conv: packed filter must be 16-byte aligned");
    SSM_REQUIRE((reinterpret_cast<size_t>(y.ptr) & 7) == 0 && y.sh % 2 == 0 && y.sc % 2 == 0 && y.sb % 2 == 0,
                "wino conv: output view must be 8-byte aligned (2x2 pixel blocks are stored as row pairs)");
    if (C2 > 0) {
        SSM_REQUIRE(x2.ptr && ssm::aligned16(x2.ptr) && x2.sb % 4 == 0, "wino conv: input 2 is not a padded-plane view");
        SSM_REQUIRE(x2.sh == x1.sh && x2.sc == x1.sc, "wino conv: cat sources must share row/channel strides");
    }
    SSM_REQUIRE((long long)CK * x1.sc * 4 < 0x7fffffffLL, "wino conv: channel stride too large");
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
    p.lrelu = ((flags & SSM_FLAG_LRELU) ? 1 : 0) | ((flags & SSM_FLAG_MASK) ? 2 : 0);
    p.abl = 0;
    p.add = nullptr;
    p.asb = p.asc = 0;
    p.ash = 0;
    p.adiv = 1;
    p.KS = 1;
    p.ksB = 0;
#ifdef SSM_WINO_ABLATE
    if (const char *e = getenv("SSM_WINO_ABL")) p.abl = atoi(e);
#endif
    return SSM_OK;
}

int wset_add(WinoParams &p, ssm_view add, int add_div, int B) {
    if (!add.ptr) return SSM_OK;
    SSM_REQUIRE(add_div >= 1 && B % add_div == 0, "wino conv: the addend serves %d batch entries each, batch %d is no multiple", add_div, B);
    SSM_REQUIRE((reinterpret_cast<size_t>(add.ptr) & 7) == 0 && add.sh % 2 == 0 && add.sc % 2 == 0 && add.sb % 2 == 0,
                "wino conv: the addend view must be 8-byte aligned (read as row pairs)");
    p.add = add.ptr;
    p.asb = add.sb;
    p.asc = add.sc;
    p.ash = add.sh;
    p.adiv = add_div;
    return SSM_OK;
}

}  // namespace

extern "C" int ssm_wino_plan(int Cin, int Cout, int B, int H, int W, int ups, int *kind, int *BN, int *CK) {
    const int kd = pick_wkind(Cin, Cout, B, H, W, ups);          // (odd widths: the epilogue stores the last column alone, r6)
    if (kd < 0) {
        ssm::set_error("wino conv: no tile configuration for Cin=%d Cout=%d on a %dx%d map (needs Cin a multiple of 8)", Cin, Cout, H, W);
        return SSM_E_UNSUPPORTED;
    }
    if (kind) *kind = kd;
    if (BN) *BN = kWInfo[kd].bn;
    if (CK) *CK = kWInfo[kd].ck;
    return SSM_OK;
}

extern "C" double ssm_wino_estimate(int Cin, int Cout, int B, int H, int W, int ups) {
    const int kd = (Cin % 8 == 0) ? pick_wkind(Cin, Cout, B, H, W, ups) : -1;
    return kd < 0 ? -1.0 : estimate_wino(kWInfo[kd], Cin, Cout, B, H, W, ups);
}

extern "C" int ssm_wino_force_kind(int kind) {
    g_force_wkind.store(kind >= 0 && kind < NWKIND ? kind : -1);
    return NWKIND;
}

extern "C" size_t ssm_wino_packed_weight_floats(int Cout, int Cin, int BN) {
    const size_t nb = (size_t)(Cout + BN - 1) / BN;
    return nb * (size_t)Cin * 16 * BN;
}

extern "C" int ssm_wino_pack_weights(const float *w, const float *bias, float *wp, float *bp, int Cout, int Cin, int BN, void *stream) {
    SSM_REQUIRE(w && bias && wp && bp, "wino pack_weights: null pointer");
    SSM_REQUIRE(Cout > 0 && Cin > 0 && BN > 0 && BN % 32 == 0, "wino pack_weights: bad sizes");
    const long long total = (long long)ssm_wino_packed_weight_floats(Cout, Cin, BN);
    const int nbias = (int)ssm_packed_bias_floats(Cout, BN);
    const long long n = total > nbias ? total : nbias;
    SSM_LAUNCH(wino_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, bias, wp, bp, Cout, Cin,
                       BN, total, nbias);
    return ssm::check_launch("ssm_wino_pack_weights");
}

extern "C" int ssm_wino_conv2d_add_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                                       ssm_view pool, ssm_view add, int add_div, int B, int H, int W, int Cout, float slope, int flags,
                                       void *stream) {
    int kind = 0, BN = 0, CK = 0;
    SSM_REQUIRE(B > 0, "wino conv: bad batch");
    const int rc = ssm_wino_plan(C1 + C2, Cout, B, H, W, 0, &kind, &BN, &CK);
    if (rc != SSM_OK) return rc;
    WinoParams p;
    const int rf = wfill(p, x1, C1, x2, C2, w_packed, bias_packed, y, H, W, Cout, slope, flags, CK, W);
    if (rf != SSM_OK) return rf;
    const int ra = wset_add(p, add, add_div, B);
    if (ra != SSM_OK) return ra;
    if (pool.ptr) {
        SSM_REQUIRE(H % 2 == 0 && W % 2 == 0, "wino conv: fused pool needs even H, W");
        p.pool = pool.ptr;
        p.psb = pool.sb;
        p.psc = pool.sc;
        p.psh = pool.sh;
    }
    return wdispatch<false>(kind, p, B, (hipStream_t)stream);
}

// ---- split-K for launches that leave most of the chip idle (r5) -------------------------------------------------------------------
// A 512 -> 512 layer on a 22x22 map at batch 2 (config 3's bottleneck layers) is 96-192 workgroups that each walk all 512 input channels:
// one partial round, as long as ONE workgroup's channel loop (0.10 ms for 4.6 GFLOP).  KS workgroups per output tile, each over Cin / KS
// channels, fill the idle CUs; their raw sums land as KS x B batch entries of a scratch tensor and ssm_splitk_finish_fwd (csrc/ssm_elem.hip)
// adds them in a fixed order (deterministic), then applies the addend, the activation and the fused 2x2 mean.
namespace {
// The two-workgroups-per-CU configuration (wino2_kernel) with BN couts per block for a split launch: the one whose tiles cover the map with
// the FEWEST workgroups (ties: the cost model).  A split launch is one partial round - as long as one workgroup's channel loop, whatever the
// tile shape (r6: conv5b of config 3, 512 -> 512 on 22x22 at batch 2, takes 80 us as 384 workgroups of 4x64-pixel tiles AND as 352 of 2x64,
// profiles/r34_small_maps_forced.txt) - so what shortens it is a deeper split, and the split is capped by co-residency: 8x32-pixel tiles
// cover a 22x22 map with 3 tiles instead of 6 (4x64: two thirds of every tile row is overshoot), half the workgroups, twice the split.
int pick_wkind_split(int CinPart, int Cout, int B, int H, int W, int ups, int BN) {
    static const bool by_cost_only = [] {          // $SSM_WINO_SPLIT_TILES=cost: the r5 rule (cost model alone), for A/B runs
        const char *e = getenv("SSM_WINO_SPLIT_TILES");
        return e && !strcmp(e, "cost");
    }();
    int best = -1;
    double bt = 0.0;
    long long bw = 0;
    for (int i = 0; i < NWKIND; ++i) {
        const WKindInfo &ki = kWInfo[i];
        if (ki.nblk != 2 || ki.bn != BN || CinPart % ki.ck) continue;
        const long long nwg = by_cost_only ? 0 : (long long)B * ((W + ki.tw - 1) / ki.tw) * ((H + ki.th - 1) / ki.th) * ((Cout + ki.bn - 1) / ki.bn);
        const double t = estimate_wino(ki, CinPart, Cout, B, H, W, ups);
        if (best < 0 || nwg < bw || (nwg == bw && t < bt * 0.999)) {
            best = i;
            bt = t;
            bw = nwg;
        }
    }
    return best;
}
}  // namespace

extern "C" int ssm_wino_splitk_plan(int Cin, int Cout, int B, int H, int W, int ups, int BN, int *KS) {
    SSM_REQUIRE(KS, "wino splitk_plan: null pointer");
    *KS = 1;
    const int enabled = ssm::splitk_switch(0).load(std::memory_order_relaxed);          // ($SSM_WINO_SPLITK, ssm_splitk_enable)
    if (!enabled || BN % 32 || Cin < 128) return SSM_OK;
    const int kd = pick_wkind_split(Cin, Cout, B, H, W, ups, BN);
    if (kd < 0) return SSM_OK;
    const WKindInfo &ki = kWInfo[kd];
    const long long nwg = (long long)B * ((W + ki.tw - 1) / ki.tw) * ((H + ki.th - 1) / ki.th) * ((Cout + ki.bn - 1) / ki.bn);
    // double the split while all workgroups stay co-resident (2 per CU) and a workgroup keeps at least 64 channels
    int ks = 1;
    while (ks < 8 && nwg * ks * 2 <= 512 && Cin % (ks * 2 * 8) == 0 && Cin / (ks * 2) >= 64) ks *= 2;
    *KS = ks;
    return SSM_OK;
}

extern "C" int ssm_wino_conv2d_splitk_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed, ssm_view part,
                                          int KS, int ups, int B, int H, int W, int Cout, int BN, void *stream) {
    SSM_REQUIRE(B > 0 && KS >= 1 && KS <= 8 && (C1 + C2) % KS == 0, "wino conv_splitk: bad batch / split (KS = %d, Cin = %d)", KS, C1 + C2);
    SSM_REQUIRE(!ups || (H % 2 == 0 && W % 2 == 0), "wino conv_splitk: the output of a x2 upsample has even H, W (got %dx%d)", H, W);
    const int kind = pick_wkind_split((C1 + C2) / KS, Cout, B, H, W, ups, BN);
    if (kind < 0) {
        ssm::set_error("wino conv_splitk: no two-workgroup configuration of %d couts for Cin/KS = %d on a %dx%d map", BN, (C1 + C2) / KS, H, W);
        return SSM_E_UNSUPPORTED;
    }
    const int CK = kWInfo[kind].ck;
    SSM_REQUIRE(((C1 + C2) / KS) % CK == 0, "wino conv_splitk: Cin / KS = %d is no multiple of the chunk (%d)", (C1 + C2) / KS, CK);
    WinoParams p;
    const int rf = wfill(p, x1, C1, x2, C2, w_packed, bias_packed, part, H, W, Cout, 0.f, 0, CK, ups ? W / 2 : W);
    if (rf != SSM_OK) return rf;
    p.KS = KS;
    p.ksB = B;
    return ups ? wdispatch<true>(kind, p, B, (hipStream_t)stream) : wdispatch<false>(kind, p, B, (hipStream_t)stream);
}

extern "C" int ssm_wino_conv2d_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                                   ssm_view pool, int B, int H, int W, int Cout, float slope, int flags, void *stream) {
    const ssm_view none = {nullptr, 0, 0, 0};
    return ssm_wino_conv2d_add_fwd(x1, C1, x2, C2, w_packed, bias_packed, y, pool, none, 1, B, H, W, Cout, slope, flags, stream);
}

extern "C" int ssm_wino_conv2d_ups_add_fwd(ssm_view a, int C1, ssm_view b, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                                           ssm_view add, int add_div, int B, int H, int W, int Cout, float slope, int flags, void *stream) {
    int kind = 0, BN = 0, CK = 0;
    SSM_REQUIRE(B > 0, "wino conv_ups: bad batch");
    SSM_REQUIRE(H % 2 == 0 && W % 2 == 0, "wino conv_ups: the output of a x2 upsample has even H, W (got %dx%d)", H, W);
    const int rc = ssm_wino_plan(C1 + C2, Cout, B, H, W, 1, &kind, &BN, &CK);
    if (rc != SSM_OK) return rc;
    WinoParams p;
    const int rf = wfill(p, a, C1, b, C2, w_packed, bias_packed, y, H, W, Cout, slope, flags, CK, W / 2);
    if (rf != SSM_OK) return rf;
    const int ra = wset_add(p, add, add_div, B);
    if (ra != SSM_OK) return ra;
    return wdispatch<true>(kind, p, B, (hipStream_t)stream);
}

extern "C" int ssm_wino_conv2d_ups_fwd(ssm_view a, int C1, ssm_view b, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                                       int B, int H, int W, int Cout, float slope, int flags, void *stream) {
    const ssm_view none = {nullptr, 0, 0, 0};
    return ssm_wino_conv2d_ups_add_fwd(a, C1, b, C2, w_packed, bias_packed, y, none, 1, B, H, W, Cout, slope, flags, stream);
}
