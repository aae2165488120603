// 3x3 convolution as Winograd F(4x4,3x3) on the CDNA4 fp32 matrix cores (v_mfma_f32_16x16x4_f32), all arithmetic fp32.
//
// Same operator as ssm_wino.hip / ssm_conv.hip for k = 3 (layers.conv of the reference, scripts/models/layers.py:21-33: stride-1
// 'same' cross-correlation, zero padding, bias, LeakyReLU; fused 2x2 mean, scripts/models/layers.py:60-63; two-source input =
// torch.cat on C; fused F.upsample(torch.cat([a, b], 1), bilinear x2), scripts/models/flow_computation.py:244-247), evaluated as
//
//      Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A          per 4x4 output tile (d = its 6x6 input patch, g = the 3x3 filter)
//
// over the points {0, +-a, +-b, inf}, a = 5/8, b = 8/5: 36 multiplies per (cin, cout, 16 outputs) instead of 144 - 4x fewer matrix-core
// cycles than the direct form, 1.78x fewer than F(2x2,3x3).  The usual points {0, +-1, +-2} make all transform constants dyadic, but
// in fp32 they cost accuracy: with 512 input channels a layer sits 3.2e-6 rms / 4.6e-5 max from a float64 evaluation at unit output
// scale; the reciprocal pair (5/8, 8/5) - same operation count, the constants become fused multiply-add operands - brings that to
// 1.4e-6 rms / 1.1e-5 max (direct form: 4e-7 rms; per-layer bar 5e-5).  The whole pair -> frame path at 736x1280 is unchanged within
// its fp32 noise (tests/emulate_winograd_f44_precision.py).
//
// Structure (as csrc/ssm_wino1d.hip): the 36 frequencies of a tile cost 144 vector operations per (cin, tile) - too many to sit in
// the MFMA loop of every wave as in ssm_wino.hip - so the workgroup transforms the chunk's patch ONCE into V [cin][f/4][tile][4] in
// LDS (two threads per (cin, tile): three frequency rows each), and the matrix loop only reads operands: per 4 frequencies one
// ds_read_b128 of U and one of V feed 4 MFMAs.  GEMM view per frequency: M_f[cout][tile] = sum_cin U_f[cout][cin] V_f[cin][tile] with
// v_mfma_f32_16x16x4_f32 (A = 16 couts x 4 cin, B = 4 cin x 16 tiles): a wave owns 16 couts x 16 tiles for ALL 36 frequencies = 36
// accumulators of 4 registers, so the output transform A^T M A is lane-local, a lane finishes whole 4x4 pixel tiles (16-byte stores,
// the 2x2 mean is lane-local too), and two workgroups share a CU (<= 256 registers per wave).  The bias is added after the output
// transform (sum of products, then + bias: the reference's order).
//
// Data movement as in ssm_conv.hip / ssm_wino.hip: padded planes, per chunk of 4 input channels the [4][9][32][4] filter values per
// 32-cout block (double-buffered) and the [4][TH+2][TW+8] patch (or, fused upsample, the low-res [4][TH/2+2][TW/2+8] patch, expanded in
// LDS) arrive by LDS-DMA, one instruction per frequency group INSIDE the matrix loop of the chunk before (a burst at a phase boundary
// costs ~200 issue cycles a piece, among MFMAs a few tens).
//
// Two kernel forms (profiles/DESIGN_history_r1-r3.md 3.2f-g has the measurements behind them):
//   wino4_kernel<W4Cfg<.., NCB = 1>>  256 threads, 32 couts x 32 tiles, two workgroups per CU, two barriers per chunk
//                                     [patch landed] expand, transform [filter landed] matrix loop; single-buffered patch and V.
//   wino4_kernel<W4Cfg<.., NCB = 2>>  512 threads, 64 couts x 32 tiles, one workgroup per CU: the transform serves twice the MFMAs
//                                     (a vector instruction beside the fp32 MFMA costs ~3 matrix cycles - tools/mfma_valu_probe.py -
//                                     so fewer of them per MFMA is the only thing that hides a transform); patch, hi-res patch and V
//                                     double-buffered, ONE barrier per chunk, the next chunk's transform in the slots of the matrix loop.
// Whatever is wave-uniform at run time (which part of a (cin, tile) a thread transforms, whether later chunks exist) selects a
// straight-line INSTANCE of the loop instead of being tested inside it: a scalar branch in front of an LDS read exposes its latency.
#include "ssm_common.h"

#include <atomic>
#include <mutex>
#include <type_traits>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

struct W4Params {
    const float *src1;
    const float *src2;
    long long sb1, sb2;  // batch strides
    long long sc;        // channel stride (both sources)
    int sh;              // row stride (both sources)
    int C1, Cin;         // channels of source 1, total
    const float *wpk;    // U, [Cout/32][Cin][9][32][4]
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
    int vec;             // 1: outputs / addend / pooled outputs may be moved as aligned 16- / 8-byte pieces (checked on the host)
    int abl;             // diagnostics build only: 1 no LDS-DMA in the loop, 2 no stores, 4 no transform
    unsigned long long *dbg;   // diagnostics build only ($SSM_WINO4_ABL & 32): per-phase shader-cycle sums of wave 0 of every workgroup
    int trace_block;     // tuning build (-DW4_TRACE): the workgroup whose waves stamp their timeline ($SSM_W4_TRACE_BLOCK)
    int stagger;         // s_sleep units (64 cycles) by which the second workgroup of every CU starts late (first round; $SSM_WINO4_STAGGER)
    int border;          // fused-upsample launches: 1 = only the workgroup tiles on the map's border ring (ssm_wino4_conv2d_ups_border_fwd)
    const float *add;    // optional pre-activation addend [B / adiv][Cout][H][W] (ssm_conv2d_add_fwd)
    long long asb, asc;
    int ash, adiv;
};

// NCB cout blocks of 32 x 2 cout halves x 2 tile groups = 4 NCB waves; a tile group = GTX x GTY tiles of 4x4 pixels (16 tiles); the
// groups sit WTY x WTX.  NCB = 1: 256 threads, two workgroups per CU.  NCB = 2 (64 couts): 512 threads, one workgroup per CU - the
// transform of a (cin, tile) serves twice the MFMAs and is split over four threads instead of two (the vector pipe, not the matrix
// pipe, is what a 32-cout workgroup saturates first: ~1000 cycles of vector issue per wave and chunk beside 1152 of MFMA).
template <int GTX_, int WTY_, int WTX_, int NCB_ = 1>
struct W4Cfg {
    static constexpr int GTX = GTX_, GTY = 16 / GTX_, WTY = WTY_, WTX = WTX_, NCB = NCB_;
    static constexpr int NW = 4 * NCB, THREADS = 64 * NW;
    static constexpr int CK = 4, BN = 32 * NCB, NT = 32;                       // chunk = one MFMA k-step of 4 input channels
    static constexpr int NTX = GTX * WTX, NTY = GTY * WTY;                     // tiles per workgroup, by axis
    static constexpr int TH = 4 * NTY, TW = 4 * NTX;                           // output pixels per workgroup
    static constexpr int PH = TH + 2, PW = TW + 8, PW4 = PW / 4;               // patch rows y0-1 .., columns x0-4 .. x0+TW+3
    static constexpr int SHIFT = 1;                                            // floats: a tile's window starts at patch column 4 Tx + 3 + SHIFT
    static constexpr int USZ1 = CK * 9 * 32 * 4, USZ = NCB * USZ1;             // filter floats per chunk: per 32-cout block, per workgroup
    static constexpr int PSZ = CK * PH * PW;                                   // patch floats per chunk
    static constexpr int VSZ = CK * 9 * NT * 4;                                // transformed patch
    static constexpr int LH = TH / 2 + 2, LW = TW / 2 + 8, LW4 = LW / 4;       // fused upsample: low-res raw patch
    static constexpr int RSZ = CK * LH * LW;
    static constexpr int NPOS = (TH / 2 + 1) * (TW / 2 + 1);                   // 2x2 hi-res block positions of the expander
    static_assert(WTY * WTX == 2 && (GTX == 4 || GTX == 8 || GTX == 16) && (NCB == 1 || NCB == 2), "two tile groups of 16 tiles");
    static_assert(USZ1 % 256 == 0 && NPOS <= 256, "filter stage = whole 1-KiB DMA groups; expander: one position per thread");
};

template <class C, bool UPS>
struct W4Lds {
    static constexpr int DSZ = UPS ? C::RSZ : C::PSZ;
    static constexpr int DH = UPS ? C::LH : C::PH, DW4 = UPS ? C::LW4 : C::PW4;
    static constexpr int NGU = C::USZ / 256;                    // 1-KiB groups of filter per chunk
    static constexpr int NDQ = DSZ / 4;                         // 16-byte pieces of (raw) patch per chunk
    static constexpr int NGP = (NDQ + 63) / 64;
    // Waves that issue the chunk's LDS-DMA.  512-thread form: only waves 0..3 - the SIMD's issue arbitration favours the older wave of
    // a pair (w, w + 4), waves 0..3 reach the chunk's barrier ~600 cycles before waves 4..7 (tools/wino4_timeline.py) and would idle
    // there; with the whole DMA issue on them the pair finishes together.
    static constexpr int NW = C::NCB == 2 ? 4 : C::NW;
    static constexpr int NIU = (NGU + NW - 1) / NW, NIP = (NGP + NW - 1) / NW, NI = NIU + NIP;   // DMA instructions per issuing wave per chunk
    static constexpr int NBUF = C::NCB == 2 ? 2 : 1;            // 512-thread form: patch, hi-res patch and V double-buffered (one barrier per chunk)
    static constexpr int UOFF = 0;                              // two filter stages
    static constexpr int DOFF = 2 * C::USZ;                     // the DMA'd patch (plain: lands SHIFT floats in; UPS: the low-res raw patch)
    static constexpr int DCAP = NGP * 256 + 256;
    static constexpr int NBD = 2;                               // the DMA'd (raw) patch is double-buffered in both forms: it runs two chunks ahead
    static constexpr int HOFF = DOFF + NBD * DCAP;              // UPS: the expanded hi-res patch
    static constexpr int HCAP = C::PSZ + 4;
    static constexpr int POFF = UPS ? HOFF : DOFF;              // the patch the transform reads (its floats start at + SHIFT in the plain form)
    static constexpr int PCAP = UPS ? HCAP : DCAP;              // ... and the distance between its two buffers
    static constexpr int VOFF = HOFF + (UPS ? NBUF * HCAP : 0); // transformed patch
    static constexpr int BYTES = (VOFF + NBUF * C::VSZ) * 4;
    static_assert(VOFF % 4 == 0 && DOFF % 4 == 0 && HOFF % 4 == 0 && HCAP % 4 == 0 && DCAP % 4 == 0, "16-byte aligned regions");
    static_assert(BYTES + (C::NCB == 2 ? 256 : 0) <= (C::NCB == 1 ? 80 : 160) * 1024, "LDS budget (two workgroups of 256 or one of 512 per CU; 64-cout form: + its biases)");
};

// interpolation points 0, +-PA, +-PB, inf (PA * PB = 1); the transform matrices in the monic form:
//   B^T rows: [a2b2 0 -(a2+b2) 0 1 0], [0 -+a b2  -b2  +-a 1 0], [0 -+b a2  -a2  +-b 1 0], [0 a2b2 0 -(a2+b2) 0 1]
//   A^T[k][f] = p_f^k (k = 0..3; the point at infinity contributes to k = 3 only);  G[f] = [1 p p^2] / prod_{q != p}(p - q), G[inf] = [0 0 1]
#ifndef W4_INTERLEAVE
#define W4_INTERLEAVE 1      // 512-thread form: the next chunk's transform in the slots of the matrix loop (0: as one block behind the loop)
#endif
#define W4_PA 0.625
#define W4_PB 1.6
constexpr float kA = (float)W4_PA, kB = (float)W4_PB, kA2 = (float)(W4_PA * W4_PA), kB2 = (float)(W4_PB * W4_PB);
constexpr float kA3 = (float)(W4_PA * W4_PA * W4_PA), kB3 = (float)(W4_PB * W4_PB * W4_PB);
constexpr float kP0 = (float)(W4_PA * W4_PA * W4_PB * W4_PB), kS2 = (float)(W4_PA * W4_PA + W4_PB * W4_PB);

// Frequency index of (row-frequency i, column-frequency j), i, j = 0..5 over the points 0, +a, -a, +b, -b, inf: the column halves
// j in {0,1,2} / {3,4,5} own 18 consecutive indices each - the two threads that share a (cin, tile) transform write whole quads.
__host__ __device__ constexpr int w4_freq(int i, int j) { return 18 * (j / 3) + 3 * i + (j % 3); }

// row pass: one patch row d[0..5] -> the three column-frequencies of half hh (hh = 0: points 0, +a, -a; hh = 1: +b, -b, inf)
__device__ __forceinline__ void w4_row_pass(int hh, const float *d, float *x) {
    if (hh == 0) {
        const float te = d[4] - kB2 * d[2], to = d[3] - kB2 * d[1];
        x[0] = (kP0 * d[0] - kS2 * d[2]) + d[4];
        x[1] = te + kA * to;
        x[2] = te - kA * to;
    } else {
        const float ue = d[4] - kA2 * d[2], uo = d[3] - kA2 * d[1];
        x[0] = ue + kB * uo;
        x[1] = ue - kB * uo;
        x[2] = (kP0 * d[1] - kS2 * d[3]) + d[5];
    }
}

// column pass: six values along y -> the six row-frequencies
__device__ __forceinline__ void w4_col_pass(float x0, float x1, float x2, float x3, float x4, float x5, float *v, int stride) {
    const float te = x4 - kB2 * x2, to = x3 - kB2 * x1;
    const float ue = x4 - kA2 * x2, uo = x3 - kA2 * x1;
    v[0] = (kP0 * x0 - kS2 * x2) + x4;
    v[stride] = te + kA * to;
    v[2 * stride] = te - kA * to;
    v[3 * stride] = ue + kB * uo;
    v[4 * stride] = ue - kB * uo;
    v[5 * stride] = (kP0 * x1 - kS2 * x3) + x5;
}

// half of the column pass: the row-frequencies 3 hq .. 3 hq + 2 (same expressions as w4_col_pass)
__device__ __forceinline__ void w4_col_pass_half(int hq, float x0, float x1, float x2, float x3, float x4, float x5, float *v, int stride) {
    if (hq == 0) {
        const float te = x4 - kB2 * x2, to = x3 - kB2 * x1;
        v[0] = (kP0 * x0 - kS2 * x2) + x4;
        v[stride] = te + kA * to;
        v[2 * stride] = te - kA * to;
    } else {
        const float ue = x4 - kA2 * x2, uo = x3 - kA2 * x1;
        v[0] = ue + kB * uo;
        v[stride] = ue - kB * uo;
        v[2 * stride] = (kP0 * x1 - kS2 * x3) + x5;
    }
}

// four threads per (cin, tile): thread part = 2 hh + hq holds the 9 consecutive frequencies 9 part .. 9 part + 8 (v[3 ii + jj], row-
// frequency 3 hq + ii, column-frequency 3 hh + jj) -> V [fq][NT tiles][4] at vo (f32x4 units, this tile's column)
template <int NT>
__device__ __forceinline__ void w4_store_v9(int part, const float *v, f32x4 *vo) {
    float *vf = (float *)vo;
    if (part == 0) {                 // f 0..8: quads 0, 1, element 0 of quad 2
        vo[0] = f32x4{v[0], v[1], v[2], v[3]};
        vo[NT] = f32x4{v[4], v[5], v[6], v[7]};
        vf[2 * NT * 4] = v[8];
    } else if (part == 1) {          // f 9..17: elements 1..3 of quad 2, quad 3, elements 0, 1 of quad 4
        vf[2 * NT * 4 + 1] = v[0];
        *(f32x2 *)(vf + 2 * NT * 4 + 2) = f32x2{v[1], v[2]};
        vo[3 * NT] = f32x4{v[3], v[4], v[5], v[6]};
        *(f32x2 *)(vf + 4 * NT * 4) = f32x2{v[7], v[8]};
    } else if (part == 2) {          // f 18..26: elements 2, 3 of quad 4, quad 5, elements 0..2 of quad 6
        *(f32x2 *)(vf + 4 * NT * 4 + 2) = f32x2{v[0], v[1]};
        vo[5 * NT] = f32x4{v[2], v[3], v[4], v[5]};
        *(f32x2 *)(vf + 6 * NT * 4) = f32x2{v[6], v[7]};
        vf[6 * NT * 4 + 2] = v[8];
    } else {                         // f 27..35: element 3 of quad 6, quads 7, 8
        vf[6 * NT * 4 + 3] = v[0];
        vo[7 * NT] = f32x4{v[1], v[2], v[3], v[4]};
        vo[8 * NT] = f32x4{v[5], v[6], v[7], v[8]};
    }
}

// a thread's 18 frequencies v[3 i + jj] of half hh -> V [fq][NT tiles][4] at vo (f32x4 units, this tile's column): quads 0..3 whole
// and the low half of quad 4 (hh = 0), or the high half of quad 4 and quads 5..8 (hh = 1)
template <int NT>
__device__ __forceinline__ void w4_store_v(int hh, const float *v, f32x4 *vo) {
    if (hh == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) vo[g * NT] = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
        *(f32x2 *)(vo + 4 * NT) = f32x2{v[16], v[17]};
    } else {
        *((f32x2 *)(vo + 4 * NT) + 1) = f32x2{v[0], v[1]};
#pragma unroll
        for (int g = 0; g < 4; ++g) vo[(5 + g) * NT] = f32x4{v[2 + 4 * g], v[3 + 4 * g], v[4 + 4 * g], v[5 + 4 * g]};
    }
}

#ifdef SSM_WINO_ABLATE
#define W4ABL(bit) (p.abl & (bit))
#elif defined(W4_CT_ABL)          // compile-time ablation (tuning builds: no per-slot branches, unlike the run-time switch)
#define W4ABL(bit) (W4_CT_ABL & (bit))
#else
#define W4ABL(bit) 0
#endif

// ---- epilogue of both kernel forms: Y = A^T M A per accumulator register (4 couts per lane), + bias, addend, LeakyReLU, stores, fused
// 2x2 mean.  A^T = [1 1 1 1 1 0; 0 a -a b -b 0; 0 a^2 a^2 b^2 b^2 0; 0 a^3 -a^3 b^3 -b^3 1].  cu0: first cout of the wave's 16-cout
// block (this lane holds couts cu0 + 4 q + r), (px, py): the lane's 4x4 output tile.
// SHUF (the sub-pixel form of conv3x3(upsample2x(x)), ssm_wino4_conv2d_shuffle_fwd): the launch is a plain convolution of the LOW-res
// map with 4 Cout effective output channels, channel 4 c + 2 a + b = the filter of output parity (a, b) of real channel c - so the four
// accumulator elements r of a lane are the four parities of ONE real channel, and its 4x4 low-res tile becomes an 8x8 block of the 2H x 2W
// output: pixel (2 (py + i) + a, 2 (px + e) + b).  dst / dsc / dsh describe that output (real channels).
template <bool SHUF>
__device__ __forceinline__ void w4_epilogue_shuffle(const W4Params &p, const f32x4 (&acc)[36], const float (&bv)[4], int b, int cu0, int q, int px, int py);

__device__ __forceinline__ void w4_epilogue(const W4Params &p, const f32x4 (&acc)[36], const float (&bv)[4], int b, int cu0, int q, int px, int py) {
        const float sl = (p.lrelu & 1) ? p.slope : 1.f;
        const bool amask = (p.lrelu & 2) != 0;          // SSM_FLAG_MASK: the addend view is a mask source (see ssm_hip.h)
        float *dstb = p.dst + (long long)b * p.dsb;
        float *poolb = p.pool ? p.pool + (long long)b * p.psb : nullptr;
                const unsigned pb = 4u * ((unsigned)(4 * q) * (unsigned)p.dsc + (unsigned)py * (unsigned)p.dsh + (unsigned)px);
        const unsigned qb = 4u * ((unsigned)(4 * q) * (unsigned)p.psc + (unsigned)(py >> 1) * (unsigned)p.psh + (unsigned)(px >> 1));
        const bool vok = py + 4 <= p.H && px + 4 <= p.W && p.vec;          // whole tile inside the map, rows as aligned 16-byte pieces
        auto st4 = [](const float *base, unsigned off_bytes, f32x4 val) {
            asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(off_bytes), "v"(val), "s"(base) : "memory");
        };
        auto st2 = [](const float *base, unsigned off_bytes, f32x2 val) {
            asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off_bytes), "v"(val), "s"(base) : "memory");
        };
        auto st1 = [](const float *base, unsigned off_bytes, float val) {
            asm volatile("global_store_dword %0, %1, %2" ::"v"(off_bytes), "v"(val), "s"(base) : "memory");
        };
        const float *addb = p.add ? p.add + (long long)(b / p.adiv) * p.asb + (long long)(4 * q) * p.asc + (long long)py * p.ash + px : nullptr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cu = cu0 + r;          // uniform; this lane's cout = cu + 4 * q
            float t[4][6];                   // A^T M: over the frequency rows i, for every frequency column j
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float m0 = acc[w4_freq(0, j)][r], m1 = acc[w4_freq(1, j)][r], m2 = acc[w4_freq(2, j)][r], m3 = acc[w4_freq(3, j)][r],
                            m4 = acc[w4_freq(4, j)][r], m5 = acc[w4_freq(5, j)][r];
                const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                t[0][j] = (m0 + s1) + s2;
                t[1][j] = kA * d1 + kB * d2;
                t[2][j] = kA2 * s1 + kB2 * s2;
                t[3][j] = (kA3 * d1 + m5) + kB3 * d2;
            }
            float y[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float s1 = t[i][1] + t[i][2], d1 = t[i][1] - t[i][2], s2 = t[i][3] + t[i][4], d2 = t[i][3] - t[i][4];
                y[i][0] = ((t[i][0] + s1) + s2) + bv[r];
                y[i][1] = (kA * d1 + kB * d2) + bv[r];
                y[i][2] = (kA2 * s1 + kB2 * s2) + bv[r];
                y[i][3] = ((kA3 * d1 + t[i][5]) + kB3 * d2) + bv[r];
            }
            if (addb) {
                const float *ap = addb + (long long)cu * p.asc;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (vok) {
                        const f32x4 z = *(const f32x4 *)(ap + (long long)i * p.ash);
                        if (amask) {          // y * LeakyReLU'(m): the data gradient of a layer leaves as dZ of the layer below (training step)
#pragma unroll
                            for (int e = 0; e < 4; ++e) y[i][e] *= z[e] > 0.f ? 1.f : p.slope;
                        } else {
                            y[i][0] += z[0];
                            y[i][1] += z[1];
                            y[i][2] += z[2];
                            y[i][3] += z[3];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (py + i < p.H && px + e < p.W) {
                                const float z = ap[(long long)i * p.ash + e];
                                if (amask) y[i][e] *= z > 0.f ? 1.f : p.slope;
                                else y[i][e] += z;
                            }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) y[i][e] = fmaxf(y[i][e], y[i][e] * sl);
            float *bp = dstb + (long long)cu * p.dsc;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (vok) {
                    st4(bp + (long long)i * p.dsh, pb, f32x4{y[i][0], y[i][1], y[i][2], y[i][3]});
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (py + i < p.H && px + e < p.W) st1(bp + (long long)i * p.dsh + e, pb, y[i][e]);
                }
            }
            if (poolb) {
                // 2x2 mean, vertical pairs first then the horizontal pair (the association of the direct kernel); H, W even (host check)
                float *qp = poolb + (long long)cu * p.psc;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float o0 = ((y[2 * i][0] + y[2 * i + 1][0]) + (y[2 * i][1] + y[2 * i + 1][1])) * 0.25f;
                    const float o1 = ((y[2 * i][2] + y[2 * i + 1][2]) + (y[2 * i][3] + y[2 * i + 1][3])) * 0.25f;
                    const bool rok = py + 2 * i < p.H;
                    if (rok && px + 4 <= p.W && p.vec) st2(qp + (long long)i * p.psh, qb, f32x2{o0, o1});
                    else if (rok) {
                        if (px + 2 <= p.W) st1(qp + (long long)i * p.psh, qb, o0);
                        if (px + 4 <= p.W) st1(qp + (long long)i * p.psh + 1, qb, o1);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}

// one cout register r of a lane: y = A^T M A + bias, LeakyReLU (the arithmetic of w4_epilogue)
__device__ __forceinline__ void w4_output_tile(const f32x4 (&acc)[36], int r, float bias, float sl, float (&y)[4][4]) {
    float t[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float m0 = acc[w4_freq(0, j)][r], m1 = acc[w4_freq(1, j)][r], m2 = acc[w4_freq(2, j)][r], m3 = acc[w4_freq(3, j)][r],
                    m4 = acc[w4_freq(4, j)][r], m5 = acc[w4_freq(5, j)][r];
        const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
        t[0][j] = (m0 + s1) + s2;
        t[1][j] = kA * d1 + kB * d2;
        t[2][j] = kA2 * s1 + kB2 * s2;
        t[3][j] = (kA3 * d1 + m5) + kB3 * d2;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float s1 = t[i][1] + t[i][2], d1 = t[i][1] - t[i][2], s2 = t[i][3] + t[i][4], d2 = t[i][3] - t[i][4];
        y[i][0] = ((t[i][0] + s1) + s2) + bias;
        y[i][1] = (kA * d1 + kB * d2) + bias;
        y[i][2] = (kA2 * s1 + kB2 * s2) + bias;
        y[i][3] = ((kA3 * d1 + t[i][5]) + kB3 * d2) + bias;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) y[i][e] = fmaxf(y[i][e], y[i][e] * sl);
}

template <>
__device__ __forceinline__ void w4_epilogue_shuffle<true>(const W4Params &p, const f32x4 (&acc)[36], const float (&bv)[4], int b, int cu0, int q, int px, int py) {
    const float sl = (p.lrelu & 1) ? p.slope : 1.f;
    float *dstb = p.dst + (long long)b * p.dsb + (long long)(cu0 >> 2) * p.dsc;          // real channel cu0 / 4 (+ q per lane)
    const unsigned pb = 4u * ((unsigned)q * (unsigned)p.dsc + (unsigned)(2 * py) * (unsigned)p.dsh + (unsigned)(2 * px));
    const bool vok = py + 4 <= p.H && px + 4 <= p.W && p.vec;          // whole low-res tile inside the region, output rows as aligned 16-byte pieces
    auto st4 = [](const float *base, unsigned off_bytes, f32x4 val) {
        // (s_nop 1: a store of more than 64 bits followed by a vector write of its data registers needs 2 wait states on gfx940+, and the
        // compiler does not see inside the asm - csrc/check_hazard.py caught exactly that here)
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(off_bytes), "v"(val), "s"(base) : "memory");
    };
    auto st1 = [](const float *base, unsigned off_bytes, float val) {
        asm volatile("global_store_dword %0, %1, %2" ::"v"(off_bytes), "v"(val), "s"(base) : "memory");
    };
#pragma unroll
    for (int a = 0; a < 2; ++a) {          // output rows 2 (py + i) + a: the parities (a, 0) and (a, 1) interleave along x
        float y0[4][4], y1[4][4];
        w4_output_tile(acc, 2 * a, bv[2 * a], sl, y0);
        w4_output_tile(acc, 2 * a + 1, bv[2 * a + 1], sl, y1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *rowp = dstb + (long long)(2 * i + a) * p.dsh;
            if (vok) {
                st4(rowp, pb, f32x4{y0[i][0], y1[i][0], y0[i][1], y1[i][1]});
                st4(rowp + 4, pb, f32x4{y0[i][2], y1[i][2], y0[i][3], y1[i][3]});
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (py + i < p.H && px + e < p.W) {
                        st1(rowp + 2 * e, pb, y0[i][e]);
                        st1(rowp + 2 * e + 1, pb, y1[i][e]);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <class C, bool UPS, bool SHUF = false>
__global__ __launch_bounds__(C::THREADS, 2) void wino4_kernel(const W4Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using L = W4Lds<C, UPS>;
    constexpr int BN = C::BN, PH = C::PH, PW = C::PW, CK = C::CK, NT = C::NT, NW = C::NW, NCB = C::NCB;

    const int tid = threadIdx.x;
    const int lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wid & 1, tg = (wid >> 1) & 1, blk = wid >> 2;          // cout half, tile group, 32-cout block of this wave

    int id = ssm_xcd_tile(blockIdx.x, gridDim.x);
    const int nb = id % p.NB;
    id /= p.NB;
    int tx, ty, b;
    if (UPS && C::NCB == 1 && p.border) {          // (256-thread fused-upsample form) only the tiles of the border ring: top row, bottom row, then (left, right) per row between (host: >= 2 x 2 tiles)
        const int nbt = 2 * p.tilesX + 2 * (p.tilesY - 2);
        const int k = id % nbt;
        b = id / nbt;
        if (k < 2 * p.tilesX) {
            ty = k < p.tilesX ? 0 : p.tilesY - 1;
            tx = k < p.tilesX ? k : k - p.tilesX;
        } else {
            const int kk = k - 2 * p.tilesX;
            ty = 1 + (kk >> 1);
            tx = (kk & 1) ? p.tilesX - 1 : 0;
        }
    } else {
        tx = id % p.tilesX;
        id /= p.tilesX;
        ty = id % p.tilesY;
        b = id / p.tilesY;
    }
    const int x0 = tx * C::TW, y0 = ty * C::TH;

    const long long porg = UPS ? (long long)(y0 / 2 - 1) * p.sh + (x0 / 2 - 4) : (long long)(y0 - 1) * p.sh + (x0 - 4);
    const float *pbase1 = p.src1 + (long long)b * p.sb1 + porg;
    const float *pbase2 = p.src2 + (long long)b * p.sb2 + porg;
    const float *wbase = p.wpk + (long long)nb * NCB * p.Cin * (9 * 32 * 4);          // the packed filter is in 32-cout blocks

    // per-lane source offsets (bytes) of the patch pieces this wave brings per chunk; the filter pieces are linear
    int poff[L::NIP];
#pragma unroll
    for (int i = 0; i < L::NIP; ++i) {
        const int qq = (i * L::NW + (wid & (L::NW - 1))) * 64 + lane;
        if (qq < L::NDQ) {
            const int c = qq / (L::DH * L::DW4);
            const int rem = qq - c * (L::DH * L::DW4);
            const int r = rem / L::DW4;
            const int j = rem - r * L::DW4;
            // rows below the source's bottom zero frame / 16-byte pieces right of its padded row: read from the frame's last row / piece
            // (zeros) - a tile that overshoots the map never brings in another plane's pixels or the memory behind the last plane, whose
            // values B^T d B would mix into the tile's valid outputs (times a zero, up to rounding - or NaN).  Per-lane constants: free.
            const int sH = UPS ? p.hs : p.H, sW = UPS ? p.ws : p.W, sy = UPS ? y0 / 2 - 1 : y0 - 1, sx = UPS ? x0 / 2 : x0;
            const int re = min(r, sH + (SSM_PADY - 1) - sy), fe = min(4 * j, ((sW + 2 * SSM_PADX + 3) & ~3) - 4 - sx);
            poff[i] = ((int)(c * p.sc) + re * p.sh + fe) * 4;
        } else {
            poff[i] = 0;          // tail of the last 1-KiB piece: lands in the region's padding
        }
    }
    const int uoff = lane * 16;
    int uoffk[L::NIU];          // 256-thread form: filter piece k of this wave = one scalar base per chunk + this lane offset
#pragma unroll
    for (int k = 0; k < L::NIU; ++k) uoffk[k] = lane * 16 + k * (L::NW * 1024);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;

    // k-th DMA instruction of this wave for chunk `ch`: k < NIU filter group 4k + wave into filter stage `stage`, else the patch
    auto issue_k = [&](int ch, int stage, int k, int pbuf = 0) {
        constexpr int NW = L::NW;          // (the issuing waves)
        if (NCB == 2 && wid >= NW) return;
        const int c0 = ch * CK;
        if (k < L::NIU) {
            const int g = NW * k + wid;
            if (NW * k + NW - 1 < L::NGU || g < L::NGU) {          // (the first part is a compile-time fact: no branch)
                constexpr int G1 = C::USZ1 / 256;                  // 1-KiB pieces per 32-cout block
                const int gb = NCB == 1 ? 0 : g / G1;
                const float *base = wbase + ((long long)gb * p.Cin + c0) * (9 * 32 * 4) + (g - gb * G1) * 256;
                const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(L::UOFF + stage * C::USZ) * 4u + (unsigned)g * 1024u);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(uoff), "s"(base), "s"(m0v) : "memory", "m0");
            }
        } else {
            const int kk = k - L::NIU;
            const int g = NW * kk + wid;
            if (NW * kk + NW - 1 < L::NGP || g < L::NGP) {
                const float *base = (c0 < p.C1) ? pbase1 + (long long)c0 * p.sc : pbase2 + (long long)(c0 - p.C1) * p.sc;
                const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(L::DOFF + pbuf * L::DCAP) * 4u + (unsigned)g * 1024u + (UPS ? 0u : 4u * C::SHIFT));
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(poff[kk]), "s"(base), "s"(m0v) : "memory", "m0");
            }
        }
    };

    f32x4 acc[36];
#pragma unroll
    for (int f = 0; f < 36; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = p.Cin / CK;
    // Tuning knob ($SSM_WINO4_STAGGER, 256-thread form): the second workgroup of every CU (dispatch order: workgroup i + 256) starts
    // late, so that the two co-resident workgroups are out of phase.  Measured: no effect (profiles/r6g_w4_stagger_b7.txt) - the two
    // waves of a SIMD serialise on the issue port whatever their phases are (profiles/DESIGN_history_r1-r3.md 3.2g).
    if (p.stagger > 0 && ((blockIdx.x >> 8) & 1)) {
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }
    // DMA order of a chunk: the patch pieces first (the transform needs them at the top of the chunk), then the filter pieces (needed
    // only by the matrix phase): the waits are counted - vmcnt(filter pieces of this wave) at the top, vmcnt(0) before the mid barrier
    static_assert(L::NI <= (NCB == 2 ? 18 : 9), "DMA issue slots of the matrix loop: one per frequency group (two in the 512-thread form)");
    auto issue_n = [&](int ch, int stage, int n) { issue_k(ch, stage, n < L::NIP ? L::NIU + n : n - L::NIP); };
    const bool u_full = wid < L::NGU - NW * (L::NIU - 1);          // this wave brings NIU filter pieces per chunk (else NIU - 1)
    if constexpr (NCB == 1) {
#pragma unroll
        for (int n = 0; n < L::NI; ++n) issue_n(0, 0, n);          // patch (buffer 0) and filter (stage 0) of chunk 0
        if (p.Cin / CK > 1) {
#pragma unroll
            for (int k = L::NIU; k < L::NI; ++k) issue_k(1, 0, k, 1);  // patch of chunk 1 (buffer 1)
        }
    }
    // bias of this lane's four couts (cb*16 + 4q + r): added after the output transform.  64-cout form: parked in LDS across the chunk
    // loop (the loop runs at the 256-register cap; four registers held for the epilogue were spilled to scratch memory)
    float bv[4];
    if constexpr (NCB == 2) {
        if (tid < BN) lds[L::BYTES / 4 + tid] = p.bias[nb * BN + tid];          // (visible after the first barrier below)
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = p.bias[nb * BN + blk * 32 + cb * 16 + 4 * q + r];
    }

    // ---- per-thread constants of the transform phase: unit = (cin, tile), two threads per unit (frequency rows 0..2 | 3..5) -------
    const int thh = (wid >> 1) & 1;                  // wave-uniform: which three column-frequencies this thread computes
    const int thq = wid >> 2;                        // NCB = 2: ... and which three row-frequencies of them
    const int tu = tid & 127;                        // unit: cin = tu / 32, tile = tu % 32
    const int tcin = tu >> 5, ttile = tu & 31;
    // tile index -> position inside the workgroup's tile: group g2 = tile / 16, (gy, gx) inside the group
    const int tgx = (ttile & 15) % C::GTX, tgy = (ttile & 15) / C::GTX, tg2 = ttile >> 4;
    const int tTx = (tg2 % C::WTX) * C::GTX + tgx, tTy = (tg2 / C::WTX) * C::GTY + tgy;
    const int t_src = L::POFF + (UPS ? 0 : C::SHIFT) + (tcin * PH + 4 * tTy) * PW + 4 * tTx + 3 + (UPS ? C::SHIFT : 0);   // floats; 16-byte aligned
    const int t_dst = L::VOFF / 4 + (tcin * 9) * NT + ttile;                                                              // f32x4 units

    // ---- the matrix loop's operand bases (f32x4 units): U of (cin = 4cp + q, fq, cout = cb*16 + l15), V of (cin, fq, tile) ---------
    const f32x4 *lds4 = (const f32x4 *)lds;
    const int aBase = L::UOFF / 4 + blk * (C::USZ1 / 4) + q * (9 * 32) + cb * 16 + l15;
    const int bBase = L::VOFF / 4 + q * (9 * NT) + tg * 16 + l15;

    // fused upsample: low-res raw chunk -> hi-res patch; one thread = one 2x2 hi-res block position, walking the chunk's channels
    // (CB, CN: the channels of the chunk this call expands - the 512-thread form splits them between its two halves of 256 threads).
    // What depends on the position only - source offsets, the edge cases of ATen's half-pixel rule, the in-image masks - is computed once per
    // thread; a call rebuilds the twelve weights from one word of flags (the masks folded into the row weights: a mask is 0 or 1, same
    // values) and is then 2 reads, 16 multiply-adds and 4 writes per channel (r3: 218 vector instructions per call of 4 channels).
    // (kept across the chunk loop: three offsets and one word of flags - holding the twelve weights themselves spilled the 64-cout kernels)
    int ex_r0 = 0, ex_r1 = 0, ex_dd = 0, ex_fl = 0;          // flags: 1 j0 == j1, 2 i0 == i1, 4 / 8 / 16 / 32 the 2x2 block's pixels inside the map, 64 thread has a position
    if constexpr (UPS) {
        constexpr int PRW = C::TW / 2 + 1, NPOS = C::NPOS, LW = C::LW;
        const int etid = threadIdx.x & 255;
        const int ly0 = y0 / 2 - 1, lx0 = x0 / 2 - 1;
        const int pi = etid / PRW, pj = etid - pi * PRW;
        const int i = ly0 + pi, j = lx0 + pj;
        const int i0 = min(max(i, 0), p.hs - 1), i1 = min(max(i + 1, 0), p.hs - 1);
        const int j0 = min(max(j, 0), p.ws - 1), j1 = min(max(j + 1, 0), p.ws - 1);
        const int Y = 2 * i + 1, X = 2 * j + 1;
        const bool yt = Y >= 0 && Y < p.H, yb2 = Y + 1 < p.H, xl = X >= 0 && X < p.W, xr = X + 1 < p.W;
        ex_fl = (j0 == j1 ? 1 : 0) | (i0 == i1 ? 2 : 0) | ((yt && xl) ? 4 : 0) | ((yt && xr) ? 8 : 0) | ((yb2 && xl) ? 16 : 0) | ((yb2 && xr) ? 32 : 0) |
                (etid < NPOS ? 64 : 0);
        // (the right neighbour is read at j0 + 1 even where the source index is clamped: its weight is then exactly 0 and the value -
        // the zero frame or a neighbouring pixel - is finite)
        ex_r0 = (i0 - ly0) * LW + 3 - lx0 + j0;
        ex_r1 = (i1 - ly0) * LW + 3 - lx0 + j0;
        ex_dd = (2 * pi) * PW + 2 * pj + 3 + C::SHIFT;       // hi-res pixel x0 + 2pj - 1 -> patch column 2pj + 3 (+ SHIFT)
    }
    auto expand = [&](int rbuf, int hbuf, auto CB, auto CN) {
        if constexpr (UPS) {
            constexpr int LH = C::LH, LW = C::LW;
            constexpr int cbeg = decltype(CB)::value, cnum = decltype(CN)::value;
            const float *raw = lds + L::DOFF + rbuf * L::DCAP + cbeg * LH * LW;
            float *hip = lds + L::HOFF + hbuf * L::HCAP + cbeg * PH * PW;
            if (ex_fl & 64) {
                const bool jeq = ex_fl & 1, ieq = ex_fl & 2;
                const float ex_xa = jeq ? 1.f : 0.75f, ex_xb = jeq ? 0.f : 0.25f;      // column 2j+1 = xa x[j0] + xb x[j0+1]
                const float ex_ca = jeq ? 1.f : 0.25f, ex_cw = jeq ? 0.f : 0.75f;      // column 2j+2 = ca x[j0] + cw x[j0+1]
                const float ya = ieq ? 1.f : 0.75f, yb = ieq ? 0.f : 0.25f;
                const float ex_a00 = (ex_fl & 4) ? ya : 0.f, ex_b00 = (ex_fl & 4) ? yb : 0.f;        // row 2i+1: ya (row i0) + yb (row i1)
                const float ex_a01 = (ex_fl & 8) ? ya : 0.f, ex_b01 = (ex_fl & 8) ? yb : 0.f;
                const float ex_a10 = (ex_fl & 16) ? yb : 0.f, ex_b10 = (ex_fl & 16) ? ya : 0.f;      // row 2i+2: yb (row i0) + ya (row i1)
                const float ex_a11 = (ex_fl & 32) ? yb : 0.f, ex_b11 = (ex_fl & 32) ? ya : 0.f;
                const float *r0 = raw + ex_r0, *r1 = raw + ex_r1;
                float *dd = hip + ex_dd;
                float v00[cnum], v01[cnum], v10[cnum], v11[cnum];
#pragma unroll
                for (int cc = 0; cc < cnum; ++cc) {
                    v00[cc] = r0[cc * LH * LW];
                    v01[cc] = r0[cc * LH * LW + 1];
                    v10[cc] = r1[cc * LH * LW];
                    v11[cc] = r1[cc * LH * LW + 1];
                }
#pragma unroll
                for (int cc = 0; cc < cnum; ++cc) {
                    const float h00 = ex_xa * v00[cc] + ex_xb * v01[cc], h01 = ex_ca * v00[cc] + ex_cw * v01[cc];
                    const float h10 = ex_xa * v10[cc] + ex_xb * v11[cc], h11 = ex_ca * v10[cc] + ex_cw * v11[cc];
                    dd[cc * PH * PW] = ex_a00 * h00 + ex_b00 * h10;
                    dd[cc * PH * PW + 1] = ex_a01 * h01 + ex_b01 * h11;
                    dd[cc * PH * PW + PW] = ex_a10 * h00 + ex_b10 * h10;
                    dd[cc * PH * PW + PW + 1] = ex_a11 * h01 + ex_b11 * h11;
                }
            }
        }
    };

    // V = B^T d B of one (cin, tile): this thread's three column-frequencies (row pass first, then the row-frequencies of each).  The
    // halves (HH: column-frequencies 0..2 | 3..5; NCB = 2 also HQ: row-frequencies 0..2 | 3..5) are wave-uniform, and each combination
    // is its own straight-line instance: with the selection inside, every row's window read sat behind a scalar branch and waited
    // out its LDS latency alone - six exposed latencies per chunk, ~1450 cycles for 80 vector instructions.
    auto transform_as = [&](auto HH, auto HQ, int sbuf, int vbuf) {
        constexpr int hh = decltype(HH)::value, hq = decltype(HQ)::value;
        const float *rp = lds + t_src + sbuf * L::PCAP;
        f32x4 *vo = (f32x4 *)lds + t_dst + vbuf * (C::VSZ / 4);
        f32x4 a4[6];
        f32x2 a2[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            a4[i] = *(const f32x4 *)(rp + i * PW);
            a2[i] = *(const f32x2 *)(rp + i * PW + 4);
        }
        float X[6][3];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float d[6] = {a4[i][0], a4[i][1], a4[i][2], a4[i][3], a2[i][0], a2[i][1]};       // (scalar arithmetic only: check_isa.sh fences v_pk_*)
            w4_row_pass(hh, d, X[i]);
        }
        if constexpr (NCB == 1) {
            float v[18];            // frequency w4_freq(i, 3 hh + jj) - 18 hh = 3 i + jj
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) w4_col_pass(X[0][jj], X[1][jj], X[2][jj], X[3][jj], X[4][jj], X[5][jj], v + jj, 3);
            w4_store_v<NT>(hh, v, vo);
        } else {
            float v[9];             // frequency w4_freq(3 hq + ii, 3 hh + jj) - 18 hh - 9 hq = 3 ii + jj
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) w4_col_pass_half(hq, X[0][jj], X[1][jj], X[2][jj], X[3][jj], X[4][jj], X[5][jj], v + jj, 3);
            w4_store_v9<NT>(2 * hh + hq, v, vo);
        }
    };
    auto transform = [&](int sbuf = 0, int vbuf = 0) {
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        if constexpr (NCB == 1) {
            if (thh == 0) transform_as(I0{}, I0{}, sbuf, vbuf);
            else transform_as(I1{}, I0{}, sbuf, vbuf);
        } else {
            if (thh == 0) {
                if (thq == 0) transform_as(I0{}, I0{}, sbuf, vbuf);
                else transform_as(I0{}, I1{}, sbuf, vbuf);
            } else {
                if (thq == 0) transform_as(I1{}, I0{}, sbuf, vbuf);
                else transform_as(I1{}, I1{}, sbuf, vbuf);
            }
        }
    };

#ifdef SSM_WINO_ABLATE
    unsigned long long tph[5] = {0, 0, 0, 0, 0};       // wait + top barrier | expand + transform | mid barrier | matrix loop | epilogue
    const bool stamp = (p.abl & 32) && p.dbg;
    unsigned long long tk = stamp ? __builtin_amdgcn_s_memtime() : 0;
    const unsigned long long tstart = tk;
#define W4STAMP(i)                                              \
    if (stamp) {                                                \
        const unsigned long long tn = __builtin_amdgcn_s_memtime(); \
        tph[i] += tn - tk;                                      \
        tk = tn;                                                \
    }
    // $SSM_WINO4_ABL & 128: timeline of workgroup 0 - every wave writes s_memtime at its phase boundaries of the first 16 chunks to
    // dbg[16 + (wave * 16 + chunk) * 4 + point]
    const bool trace = (p.abl & 128) && p.dbg && blockIdx.x == 0;
#define W4TRACE(chv, pt)                                                                                             \
    if (trace && (chv) < 16 && lane == 0) p.dbg[16 + (wid * 16 + (chv)) * 4 + (pt)] = __builtin_amdgcn_s_memtime();
#elif defined(W4_TRACE)
    // tuning build (-DW4_TRACE): the same timeline without run-time switches in the loop - the stamps of workgroup 0 go to 4 KiB of
    // LDS behind the kernel's regions (no vector-memory traffic that the loop's vmcnt waits would see) and to dbg[16 ..] at the end
#define W4STAMP(i)
    unsigned long long *ltrace = (unsigned long long *)(lds + L::BYTES / 4 + 64);
    const bool trace = p.dbg && blockIdx.x == (unsigned)p.trace_block;
#define W4TRACE(chv, pt) \
    if (trace && (chv) < 16 && lane == 0) ltrace[(wid * 16 + (chv)) * 4 + (pt)] = __builtin_amdgcn_s_memtime();
    if (trace && lane == 0) ltrace[(wid * 16 + 0) * 4 + 1] = __builtin_amdgcn_s_memtime();          // kernel entry (slot 1 of chunk 0)
#else
#define W4STAMP(i)
#define W4TRACE(chv, pt)
#endif
    f32x4 a[3], bq[3];
    // ---- matrix phase of one chunk: 9 groups of 4 frequencies = 36 MFMAs of 32 cycles; the operands of group g+2 are fetched behind the
    // first MFMA of group g (a ring of three register sets: a group of four MFMAs alone is shorter than the LDS latency); dma(g): the
    // g-th LDS-DMA instruction of a later chunk, one per group behind its second MFMA (an LDS-DMA issued in a burst at a phase boundary
    // costs ~200 cycles a piece, among MFMAs a few tens)
    auto matrix = [&](int stage, int vbuf, auto dma, auto slot) __attribute__((always_inline)) {
        const int ai = aBase + stage * (C::USZ / 4), bi = bBase + vbuf * (C::VSZ / 4);
        a[0] = lds4[ai];
        bq[0] = lds4[bi];
        a[1] = lds4[ai + 32];
        bq[1] = lds4[bi + NT];
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            const int cur = g % 3, nxt = (g + 2) % 3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[4 * g + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][e], bq[cur][e], acc[4 * g + e], 0, 0, 0);
                if (e == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + 2 < 9 && !W4ABL(8)) {          // (tuning builds, bit 8: the matrix loop re-uses the operands it holds)
                        a[nxt] = lds4[ai + (g + 2) * 32];
                        bq[nxt] = lds4[bi + (g + 2) * NT];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (NCB == 2) {          // slot(m): the m-th piece of the next chunk's transform, behind MFMA m
                    slot(4 * g + e);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (NCB == 2) {          // two DMA slots per group (behind its second and fourth MFMA)
                    if ((e == 1 || e == 3) && 2 * g + (e == 3) < L::NI) {
                        dma(2 * g + (e == 3));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if (e == 1 && g < L::NI) {
                    dma(g);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if constexpr (NCB == 1) {
        // Two barriers per chunk: [patch of chunk ch landed | V free] expand, transform [filter of chunk ch landed | V complete] matrix
        // loop, which carries the DMA of chunk ch+1: the patch into the buffer the transform has finished with, the filter into the stage
        // the matrix loop of chunk ch-1 has released.
        // NEXT (compile-time): a chunk follows - its DMA rides in this chunk's matrix loop.  The scalar operands of those DMA instructions
        // are formed HERE, once per chunk: with the test "is there a next chunk" in every slot, each piece sat behind its own scalar
        // branch with its own copy of the address arithmetic (7 scalar instructions per filter piece, 18 per patch piece) - and a
        // scalar instruction costs a wave's issue slot like any other (tuning build, bit 32: every piece from one address: conv11b
        // 0.558 -> 0.467 ms, fuse_conv 0.935 -> 0.779 at batch 7, as much as having no DMA at all).
        // N1 / N2 (compile-time): chunks ch+1 / ch+2 exist.  Order of a wave's DMA inside matrix(ch): the filter pieces of chunk ch+1, then
        // the patch pieces of chunk ch+2 (into the patch buffer chunk ch has just been transformed from).  The patch - every
        // workgroup's own rows, an HBM-latency read - so has two chunks to arrive, the filter (L2-resident, shared by all
        // workgroups) one; the counted waits: at the top the patch of chunk ch (younger: filter(ch), patch(ch+1)), before the
        // matrix loop the filter of chunk ch (younger: patch(ch+1)).
        static_assert(L::NGP % L::NW == 0, "every wave brings NIP patch pieces per chunk");
        auto chunk1 = [&](int ch, auto N1, auto N2) __attribute__((always_inline)) {
            constexpr bool n1 = decltype(N1)::value, n2 = decltype(N2)::value;
            const int stage = ch & 1;
            if (u_full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::NIU + (n1 ? L::NIP : 0)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::NIU - 1 + (n1 ? L::NIP : 0)) : "memory");
            __syncthreads();
            W4STAMP(0)
            if constexpr (UPS) {
                expand(stage, 0, std::integral_constant<int, 0>{}, std::integral_constant<int, CK>{});
                __syncthreads();
            }
            if (!W4ABL(4) || ch == 0) transform(UPS ? 0 : stage, 0);
            W4STAMP(1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n1 ? L::NIP : 0) : "memory");          // the filter of chunk ch
            __syncthreads();
            W4STAMP(2)
            const int (&uk)[L::NIU] = uoffk;          // (named here: the nested lambda of a generic lambda does not capture it implicitly)
            const int (&pk)[L::NIP] = poff;
            const int c1 = (ch + 1) * CK, c2 = (ch + 2) * CK;
            const float *fb = wbase + wid * 256 + (long long)c1 * (9 * 32 * 4);                           // filter piece k of this wave: + k * 4 KiB
            const float *pb = (c2 < p.C1) ? pbase1 + (long long)c2 * p.sc : pbase2 + (long long)(c2 - p.C1) * p.sc;
            const unsigned mu = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(L::UOFF + (stage ^ 1) * C::USZ) * 4u + (unsigned)wid * 1024u);
            const unsigned mp = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(L::DOFF + stage * L::DCAP) * 4u + (unsigned)wid * 1024u + (UPS ? 0u : 4u * C::SHIFT));
            matrix(stage, 0, [&](int n) {
                if (W4ABL(1) && ch >= 1) return;
                constexpr int NW = L::NW;
                if (n < L::NIU) {
                    if constexpr (n1) {
                        if (NW * n + NW - 1 < L::NGU || NW * n + wid < L::NGU)
                            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(uk[n]), "s"(fb), "s"(mu + (unsigned)(n * NW * 1024)) : "memory", "m0");
                    }
                } else if constexpr (n2) {
                    const int k = n - L::NIU;
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(pk[k]), "s"(pb), "s"(mp + (unsigned)(k * NW * 1024)) : "memory", "m0");
                }
            }, [](int) {});
            W4STAMP(3)
        };
        using T = std::true_type;
        using F = std::false_type;
        int ch = 0;
        for (; ch + 2 < nchunks; ++ch) chunk1(ch, T{}, T{});
        if (ch + 1 < nchunks) chunk1(ch++, T{}, F{});
        chunk1(ch, F{}, F{});

    } else {
        // One barrier per chunk (patch, hi-res patch and V double-buffered): behind its MFMAs of chunk ch a wave transforms chunk ch+1
        // (and, fused upsample, expands chunk ch+2) while the other wave of its SIMD is still feeding the matrix pipe - the two waves
        // of a SIMD belong to the same workgroup here, and with two barriers per chunk they would sit in the same phase all the time.
        //   plain:  the DMA inside matrix(ch) brings the filter of chunk ch+1 and the patch of chunk ch+2
        //   UPS:    ... the filter of chunk ch+1 and the raw patch of chunk ch+3 (raw ch+2 is expanded behind transform(ch+1))
        using T = std::true_type;
        using F = std::false_type;
#pragma unroll
        for (int k = 0; k < L::NI; ++k) issue_k(0, 0, k, 0);
        if (nchunks > 1) {
#pragma unroll
            for (int k = L::NIU; k < L::NI; ++k) issue_k(1, 0, k, 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        using C0 = std::integral_constant<int, 0>;
        using C2 = std::integral_constant<int, 2>;
        auto expand_half = [&](int rbuf, int hbuf) {          // waves 0..3: channels 0, 1 of the chunk; waves 4..7: channels 2, 3
            if (blk == 0) expand(rbuf, hbuf, C0{}, C2{});
            else expand(rbuf, hbuf, C2{}, C2{});
        };
        if constexpr (UPS) {
            expand_half(0, 0);
            if (nchunks > 1) expand_half(1, 1);
            __syncthreads();
            if (nchunks > 2) {
#pragma unroll
                for (int k = L::NIU; k < L::NI; ++k) issue_k(2, 0, k, 0);
            }
        }
        transform(0, 0);
        // The transform of chunk ch+1 rides in the slots of the matrix loop of chunk ch, one or two LDS / up to seven vector instructions
        // behind an MFMA:
        //   slots 0..5    window row i (one ds_read_b128 + one ds_read_b64)
        //   slots 3..8    row pass of row i - 3 (its read is three MFMAs old)
        //   slots 9..17   the three column passes, three slots each
        //   slots 18..22  the V stores
        // so that a wave never leaves the matrix pipe without queued work for a whole vector phase.  The role (HH, HQ) is wave-uniform:
        // the loop is instantiated per role and carries no branch.
        f32x4 r4[6];
        f32x2 r2[6];
        float tX[6][3];
        // In the loop only waves 4..7 transform - TWO threads per (cin, tile), each the row pass of its three column-frequencies (42
        // operations) and their whole column pass (42) = 84 per thread, 18 frequencies stored as 4 x 16 + 8 bytes - while their SIMD
        // partners, waves 0..3, issue the chunk's DMA and (fused upsample) expand all four channels.  (r3 gave every wave a quarter unit:
        // each row pass was computed twice, 126 vector instructions per SIMD and chunk instead of 84; same-box A/B at batch 14: 3 % over
        // the 3x3 layers, profiles/r8k_wino4_half_units_ab.txt.  The block-form transform of the prologue keeps the quarter units.)
        float tv18[18];
        auto tstep2 = [&](auto HH, int m, const float *src, f32x4 *vo) __attribute__((always_inline)) {
            constexpr int hh = decltype(HH)::value;
            if (m < 6) {
                r4[m] = *(const f32x4 *)(src + m * PW);
                r2[m] = *(const f32x2 *)(src + m * PW + 4);
            }
            if (m >= 3 && m < 9) {
                const int i = m - 3;
                const float d[6] = {r4[i][0], r4[i][1], r4[i][2], r4[i][3], r2[i][0], r2[i][1]};
                w4_row_pass(hh, d, tX[i]);
            }
            if (m >= 9 && m < 18) {
                const int jj = (m - 9) / 3, st3 = (m - 9) % 3;
                const float x0v = tX[0][jj], x1v = tX[1][jj], x2v = tX[2][jj], x3v = tX[3][jj], x4v = tX[4][jj], x5v = tX[5][jj];
                if (st3 == 0) {          // (the expressions of w4_col_pass)
                    tv18[jj] = (kP0 * x0v - kS2 * x2v) + x4v;
                    tv18[15 + jj] = (kP0 * x1v - kS2 * x3v) + x5v;
                }
                if (st3 == 1) {
                    const float te = x4v - kB2 * x2v, to = x3v - kB2 * x1v;
                    tv18[3 + jj] = te + kA * to;
                    tv18[6 + jj] = te - kA * to;
                }
                if (st3 == 2) {
                    const float ue = x4v - kA2 * x2v, uo = x3v - kA2 * x1v;
                    tv18[9 + jj] = ue + kB * uo;
                    tv18[12 + jj] = ue - kB * uo;
                }
            }
            if (m >= 18 && m < 23) {          // the five stores of w4_store_v, one per slot
                const int k = m - 18;
                if (hh == 0) {
                    if (k < 4) vo[k * NT] = f32x4{tv18[4 * k], tv18[4 * k + 1], tv18[4 * k + 2], tv18[4 * k + 3]};
                    else *(f32x2 *)(vo + 4 * NT) = f32x2{tv18[16], tv18[17]};
                } else {
                    if (k == 0) *((f32x2 *)(vo + 4 * NT) + 1) = f32x2{tv18[0], tv18[1]};
                    else vo[(4 + k) * NT] = f32x4{tv18[4 * k - 2], tv18[4 * k - 1], tv18[4 * k], tv18[4 * k + 1]};
                }
            }
        };
        // one chunk; STEADY: chunks ch+1 .. ch+3 exist (compile-time: the steady state carries no per-slot branch on the tail conditions)
        auto chunk = [&](int ch, auto STEADY, auto HH, auto HQ) __attribute__((always_inline)) {
            constexpr bool steady = decltype(STEADY)::value;
            const bool m1 = steady || ch + 1 < nchunks, m2 = steady || ch + 2 < nchunks, m3 = steady || ch + 3 < nchunks;
            const int st = ch & 1;
            // V(ch) complete, filter of chunk ch and (raw) patch of the next chunk(s) landed, every wave done with chunk ch-1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!W4ABL(16) || ch == 0) __syncthreads();          // (tuning builds, bit 16: no chunk barrier - wrong results, timing only)
            W4STAMP(0)
            W4TRACE(ch, 0)
            const float *tsrc = lds + t_src + (st ^ 1) * L::PCAP;
            f32x4 *tdst = (f32x4 *)lds + t_dst + (st ^ 1) * (C::VSZ / 4);
            matrix(st, st, [&](int g) {
                if (W4ABL(1) || decltype(HQ)::value != 0) return;          // (HQ = wave >> 2: waves 4..7 issue no DMA)
                if (g < L::NIU) {
                    if (m1) issue_k(ch + 1, st ^ 1, g, 0);
                } else if constexpr (UPS) {
                    if (m3) issue_k(ch + 3, 0, g, st ^ 1);
                } else {
                    if (m2) issue_k(ch + 2, 0, g, st);
                }
            }, [&](int m) {
                if constexpr (decltype(HQ)::value == 1) {
                    if (W4_INTERLEAVE && m1 && !W4ABL(4)) tstep2(HH, m, tsrc, tdst);
                }
            });
            W4STAMP(3)
            W4TRACE(ch, 2)
            if (!W4_INTERLEAVE && m1 && !W4ABL(4)) transform_as(HH, HQ, st ^ 1, st ^ 1);          // (block form: always the quarter arrangement)
            if constexpr (UPS) {
                // waves 4..7 carried the whole transform and are the longer role of a SIMD pair: waves 0..3 expand all four channels (a
                // 3 : 1 or 2 : 2 split of the expansion between the halves measured 17-23 % SLOWER, profiles/r8z_wino4_expand_split.txt)
                if (m2 && decltype(HQ)::value == 0) expand(st, st, C0{}, std::integral_constant<int, CK>{});
            }
            W4STAMP(1)
            W4TRACE(ch, 3)
        };
        auto chunks = [&](auto HH, auto HQ) __attribute__((always_inline)) {
            // Static priority for the transforming waves: they are the longer role of every SIMD pair (36 MFMAs + 76 interleaved vector
            // instructions against 36 MFMAs + DMA), and at equal priority the arbiter serves the older wave 0..3 first.  -3.3 % on the
            // 64-cout layers, same box, both run orders (profiles/r9b_wino4_static_priority.txt); priority 3 measures the same as 1.
            if constexpr (decltype(HQ)::value == 1) __builtin_amdgcn_s_setprio(1);
            int ch = 0;
            for (; ch + 3 < nchunks; ++ch) chunk(ch, T{}, HH, HQ);
            for (; ch < nchunks; ++ch) chunk(ch, F{}, HH, HQ);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        if (thh == 0) {
            if (thq == 0) chunks(I0{}, I0{});
            else chunks(I0{}, I1{});
        } else {
            if (thq == 0) chunks(I1{}, I0{});
            else chunks(I1{}, I1{});
        }
    }

    // ---- epilogue: Y = A^T M A per accumulator register (4 couts per lane), addend, LeakyReLU, stores, fused 2x2 mean -------------
    // A^T = [1 1 1 1 1 0; 0 a -a b -b 0; 0 a^2 a^2 b^2 b^2 0; 0 a^3 -a^3 b^3 -b^3 1]
#if defined(SSM_WINO_ABLATE) || defined(W4_CT_ABL)
    if (W4ABL(2) && acc[0][0] != 12345.678f) return;
#endif
    {
        if constexpr (NCB == 2) {
            const f32x4 b4 = *(const f32x4 *)(lds + L::BYTES / 4 + blk * 32 + cb * 16 + 4 * q);
            bv[0] = b4[0], bv[1] = b4[1], bv[2] = b4[2], bv[3] = b4[3];
        }
        const int gx = l15 % C::GTX, gy = l15 / C::GTX;
        const int Tx = (tg % C::WTX) * C::GTX + gx, Ty = (tg / C::WTX) * C::GTY + gy;
        if constexpr (SHUF) w4_epilogue_shuffle<true>(p, acc, bv, b, nb * BN + blk * 32 + cb * 16, q, x0 + 4 * Tx, y0 + 4 * Ty);
        else w4_epilogue(p, acc, bv, b, nb * BN + blk * 32 + cb * 16, q, x0 + 4 * Tx, y0 + 4 * Ty);
    }
#if defined(W4_TRACE) && !defined(SSM_WINO_ABLATE)
    if (trace) {
        if (lane == 0) ltrace[(wid * 16 + 1) * 4 + 1] = __builtin_amdgcn_s_memtime();          // epilogue issued (slot 1 of chunk 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) ltrace[(wid * 16 + 2) * 4 + 1] = __builtin_amdgcn_s_memtime();          // ... and its stores complete (slot 1 of chunk 2)
        __syncthreads();
        for (int i = tid; i < NW * 64; i += C::THREADS) p.dbg[16 + i] = ltrace[i];
    }
#endif
#ifdef SSM_WINO_ABLATE
    if (stamp) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W4STAMP(4)
        if (tid == 0) {
            for (int i = 0; i < 5; ++i) atomicAdd(p.dbg + i, tph[i]);
            atomicAdd(p.dbg + 5, tk - tstart);
            atomicAdd(p.dbg + 6, 1ULL);
        }
    }
#endif
}

// ---- tile configurations ---------------------------------------------------------------------------------------------------------
//                     GTX WTY WTX          tiles of 4x4 px     TH   TW
using X4A = W4Cfg<8, 2, 1>;      //          8 x 4                16   32
using X4B = W4Cfg<16, 2, 1>;     //         16 x 2                 8   64
using X4C = W4Cfg<4, 1, 2>;      //          8 x 4 (4x4 groups)   16   32

// 64 couts x 32 tiles per workgroup of 8 waves (same tile shapes as X4*)
using Y4A = W4Cfg<8, 2, 1, 2>;
using Y4B = W4Cfg<16, 2, 1, 2>;
using Y4C = W4Cfg<4, 1, 2, 2>;

// (r5: a third form, 32 couts x 64 tiles per workgroup of 8 waves - one filter slab and one barrier per chunk for twice the tiles, every
// wave carrying a half-unit transform behind its MFMAs - was built, passed the suite and measured 10 % SLOWER than two co-resident
// 256-thread workgroups on conv11b / fuse_conv (profiles/r11j_wino4t_layers.txt): removed)

#define SSM_W4_KINDS(X) X(X4A_, X4A) X(X4B_, X4B) X(X4C_, X4C) X(Y4A_, Y4A) X(Y4B_, Y4B) X(Y4C_, Y4C)

enum W4Kind {
#define X(name, cfg) name,
    SSM_W4_KINDS(X)
#undef X
        NW4KIND
};

struct W4KindInfo {
    int th, tw, bn;
};

constexpr W4KindInfo kW4Info[NW4KIND] = {
#define X(name, cfg) W4KindInfo{cfg::TH, cfg::TW, cfg::BN},
    SSM_W4_KINDS(X)
#undef X
};

std::atomic<int> g_force_w4kind{-1};
#if defined(SSM_WINO_ABLATE) || defined(W4_TRACE)
std::atomic<unsigned long long *> g_w4dbg{nullptr};      // diagnostics builds only (ssm_wino4_debug_buffer; `make wabl` / `make wtrace`)
#endif

// Estimated duration (cycles) of a launch, fitted to tools/bench_layers_wino.py at batch 7 (profiles/r7b_wino4_layers_b7.txt: time x clock
// / rounds = chunks x c + e per workgroup).  256-thread form: two co-resident workgroups per CU, c = 4500 cycles per chunk of 4 input
// channels (2 x 36 MFMAs of 32 cycles are 2304 of them), e = 11 k of prologue + epilogue (the other workgroup of the CU covers most of
// it); whole rounds of 512 workgroups.  64-cout form: one workgroup per CU, c = 3600 for twice the couts, e = 26 k (nothing covers its
// first loads and its stores), rounds of 256 - the better form from ~16 chunks on.
double estimate_w4(const W4KindInfo &ki, int Cin, int Cout, int B, int H, int W, int ups) {
    const long long tiles = (long long)B * ((W + ki.tw - 1) / ki.tw) * ((H + ki.th - 1) / ki.th);
    const long long nwg = tiles * (Cout / ki.bn);
    const double chunks = (double)Cin / 4.0;
    if (ki.bn == 64) {          // (with the DMA issue on waves 0..3: c = 3500, e = 24 k; ahead of the 256-thread form from 16 chunks on)
        const double per = chunks * 3500.0 + 24000.0;
        return (double)((nwg + 255) / 256) * per;
    }
    const double per = chunks * (4500.0 + (ups ? 250.0 : 0.0)) + 11000.0;
    const long long full = nwg / 512, rem = nwg % 512;
    double t = (double)full * per;
    if (rem) t += rem > 256 ? per : chunks * (3300.0 + (ups ? 250.0 : 0.0)) + 11000.0;     // a last round of lone workgroups
    return t;
}

int pick_w4kind(int Cin, int Cout, int B, int H, int W, int ups) {
    const int forced = g_force_w4kind.load();
    if (forced >= 0 && forced < NW4KIND) return forced;
    int best = -1;
    double bt = 0.0;
    static const int allow_wide = [] {
        const char *e = getenv("SSM_WINO4_WIDE");
        return e ? atoi(e) : 1;
    }();
    for (int i = 0; i < NW4KIND; ++i) {
        if (kW4Info[i].bn == 64 && (!allow_wide || Cout % 64)) continue;
        const double t = estimate_w4(kW4Info[i], Cin, Cout, B, H, W, ups);
        if (best < 0 || t < bt * 0.999) {
            best = i;
            bt = t;
        }
    }
    return best;
}

template <class C, bool UPS, bool SHUF = false>
int w4launch(W4Params &p, int B, hipStream_t st) {
    p.tilesX = (p.W + C::TW - 1) / C::TW;
    p.tilesY = (p.H + C::TH - 1) / C::TH;
    if (p.Cout % C::BN) {
        ssm::set_error("wino4 conv: tile configuration of %d couts per workgroup, Cout = %d", C::BN, p.Cout);
        return SSM_E_UNSUPPORTED;
    }
    p.NB = p.Cout / C::BN;
    // (no read outside the padded plane: the per-lane DMA offsets clamp overshoot rows / pieces to the zero frame, see wino4_kernel)
    if (p.border && (!UPS || p.tilesX < 2 || p.tilesY < 2)) {
        ssm::set_error("wino4 conv: a border-ring launch needs the fused-upsample form and at least 2 x 2 workgroup tiles (%d x %d)", p.tilesX, p.tilesY);
        return SSM_E_UNSUPPORTED;
    }
    const long long ntiles = p.border ? 2LL * p.tilesX + 2LL * (p.tilesY - 2) : (long long)p.tilesX * p.tilesY;
    const long long blocks = ntiles * p.NB * B;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        ssm::set_error("wino4 conv: grid of %lld workgroups out of range", blocks);
        return SSM_E_ARG;
    }
    void (*kern)(const W4Params) = wino4_kernel<C, UPS, SHUF>;
    int lds_bytes = W4Lds<C, UPS>::BYTES + (C::NCB == 2 ? C::BN * 4 : 0);          // (64-cout form: + the workgroup's biases)
    const int threads = C::THREADS;
#ifdef W4_TRACE
    lds_bytes += 4096;
#endif
#ifdef SSM_WINO_ABLATE
    if (const char *e = getenv("SSM_WINO4_SOLO"))          // diagnostics: one workgroup per CU (the LDS request leaves no room for a second)
        if (atoi(e) && lds_bytes < 100 * 1024) lds_bytes = 100 * 1024;
#endif
    static std::atomic<uint64_t> lds_reserved{0};          // one bit per device: the attribute is per (kernel, device)
    const hipError_t attr_rc = ssm::reserve_lds(lds_reserved, (const void *)kern, lds_bytes);
    if (attr_rc != hipSuccess) {
        ssm::set_error("wino4 conv: cannot reserve %d bytes of LDS: %s", lds_bytes, hipGetErrorString(attr_rc));
        return SSM_E_LAUNCH;
    }
    SSM_LAUNCH(kern, dim3((unsigned)blocks), dim3(threads), lds_bytes, st, p);
    return ssm::check_launch(UPS ? "ssm_wino4_conv2d_ups_fwd" : "ssm_wino4_conv2d_fwd");
}

template <bool UPS>
int w4dispatch(int kind, W4Params &p, int B, hipStream_t st) {
    switch (kind) {
#define X(name, cfg) \
    case name: return w4launch<cfg, UPS>(p, B, st);
        SSM_W4_KINDS(X)
#undef X
    }
    ssm::set_error("wino4 conv: tile configuration %d is not available for this problem", kind);
    return SSM_E_UNSUPPORTED;
}

// U = G g G^T with G[f] = [1 p p^2] / prod_{q != p} (p - q) over the finite points p = 0, +a, -a, +b, -b and G[inf] = [0 0 1], evaluated in
// float64 and rounded once; packed index -> (nb, cin, fq, n, e), frequency f = 4 fq + e = w4_freq(i, j)
__global__ void wino4_pack_kernel(const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ wp,
                                  float *__restrict__ bp, int Cout, int Cin, long long total, int nbias) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) {
        long long r = idx;
        const int e = (int)(r % 4);
        r /= 4;
        const int n = (int)(r % 32);
        r /= 32;
        const int fq = (int)(r % 9);
        r /= 9;
        const int cin = (int)(r % Cin);
        const int nb = (int)(r / Cin);
        const int co = nb * 32 + n, f = 4 * fq + e, i = (f % 18) / 3, j = 3 * (f / 18) + f % 3;          // f = w4_freq(i, j)
        double val = 0.0;
        if (co < Cout) {
            // row f of G (fixed-index locals only: a [6][3] table indexed at run time lived in scratch memory)
            auto grow = [](int f, double (&g)[3]) {
                const double p = f == 0 ? 0.0 : f == 1 ? W4_PA : f == 2 ? -W4_PA : f == 3 ? W4_PB : -W4_PB;
                if (f == 5) {
                    g[0] = g[1] = 0.0;
                    g[2] = 1.0;
                    return;
                }
                double nrm = 1.0;
#pragma unroll
                for (int o = 0; o < 5; ++o) {
                    const double po = o == 0 ? 0.0 : o == 1 ? W4_PA : o == 2 ? -W4_PA : o == 3 ? W4_PB : -W4_PB;
                    if (o != f) nrm *= p - po;
                }
                g[0] = 1.0 / nrm;
                g[1] = p / nrm;
                g[2] = p * p / nrm;
            };
            double gi[3], gj[3];
            grow(i, gi);
            grow(j, gj);
            const float *g = w + ((long long)co * Cin + cin) * 9;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) val += gi[a] * (double)g[3 * a + c] * gj[c];
        }
        wp[idx] = (float)val;
    }
    if (idx < nbias) bp[idx] = (idx < Cout) ? bias[idx] : 0.f;
}

int w4fill(W4Params &p, ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed, ssm_view y, ssm_view pool,
           ssm_view add, int add_div, int B, int H, int W, int Cout, float slope, int flags, int srcW) {
    SSM_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && C1 > 0 && C2 >= 0, "wino4 conv: bad sizes");
    SSM_REQUIRE(Cout % 32 == 0, "wino4 conv: Cout (%d) must be a multiple of 32", Cout);
    SSM_REQUIRE(x1.ptr && y.ptr && w_packed && bias_packed, "wino4 conv: null pointer");
    SSM_REQUIRE(C1 % 4 == 0 && C2 % 4 == 0, "wino4 conv: channel counts (%d,%d) must be multiples of 4", C1, C2);
    SSM_REQUIRE(ssm::aligned16(x1.ptr) && x1.sh % 4 == 0 && x1.sc % 4 == 0 && x1.sb % 4 == 0,
                "wino4 conv: input 1 is not a padded-plane view (16-byte alignment)");
    SSM_REQUIRE(x1.sh >= srcW + 2 * SSM_PADX, "wino4 conv: input 1 row stride %d leaves no zero frame for W=%d", x1.sh, srcW);
    SSM_REQUIRE(ssm::aligned16(w_packed), "wino4 conv: packed filter must be 16-byte aligned");
    if (C2 > 0) {
        SSM_REQUIRE(x2.ptr && ssm::aligned16(x2.ptr) && x2.sb % 4 == 0, "wino4 conv: input 2 is not a padded-plane view");
        SSM_REQUIRE(x2.sh == x1.sh && x2.sc == x1.sc, "wino4 conv: cat sources must share row/channel strides");
    }
    SSM_REQUIRE(4LL * x1.sc * 4 < 0x7fffffffLL, "wino4 conv: channel stride too large");
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
    p.border = 0;
    p.add = nullptr;
    p.asb = p.asc = 0;
    p.ash = 0;
    p.adiv = 1;
#ifdef SSM_WINO_ABLATE
    if (const char *e = getenv("SSM_WINO4_ABL")) p.abl = atoi(e);
#endif
#if defined(SSM_WINO_ABLATE) || defined(W4_TRACE)
    p.dbg = g_w4dbg.load();
#else
    p.dbg = nullptr;
#endif
    static const int stagger = [] {
        const char *e = getenv("SSM_WINO4_STAGGER");
        return e ? atoi(e) : 0;
    }();
    p.stagger = stagger;
    p.trace_block = 0;
#ifdef W4_TRACE
    if (const char *e = getenv("SSM_W4_TRACE_BLOCK")) p.trace_block = atoi(e);
#endif
    bool vec = W % 4 == 0 && ssm::aligned16(y.ptr) && y.sh % 4 == 0 && y.sc % 4 == 0 && y.sb % 4 == 0;
    if (add.ptr) {
        SSM_REQUIRE(add_div >= 1 && B % add_div == 0, "wino4 conv: the addend serves %d batch entries each, batch %d is no multiple", add_div, B);
        p.add = add.ptr;
        p.asb = add.sb;
        p.asc = add.sc;
        p.ash = add.sh;
        p.adiv = add_div;
        vec = vec && ssm::aligned16(add.ptr) && add.sh % 4 == 0 && add.sc % 4 == 0 && add.sb % 4 == 0;
    }
    if (pool.ptr) {
        SSM_REQUIRE(H % 2 == 0 && W % 2 == 0, "wino4 conv: fused pool needs even H, W");
        p.pool = pool.ptr;
        p.psb = pool.sb;
        p.psc = pool.sc;
        p.psh = pool.sh;
        vec = vec && (reinterpret_cast<size_t>(pool.ptr) & 7) == 0 && pool.sh % 2 == 0 && pool.sc % 2 == 0 && pool.sb % 2 == 0;
    }
    p.vec = vec ? 1 : 0;
    return SSM_OK;
}

}  // namespace

extern "C" int ssm_wino4_plan(int Cin, int Cout, int B, int H, int W, int ups, int *kind, int *BN, int *CK) {
    if (Cin % 4 || Cout % 32 || Cin <= 0 || Cout <= 0) {
        ssm::set_error("wino4 conv: no tile configuration for Cin=%d Cout=%d (Cin a multiple of 4, Cout a multiple of 32)", Cin, Cout);
        return SSM_E_UNSUPPORTED;
    }
    if (kind) *kind = pick_w4kind(Cin, Cout, B, H, W, ups);
    if (BN) *BN = 32;
    if (CK) *CK = 4;
    return SSM_OK;
}

#if defined(SSM_WINO_ABLATE) || defined(W4_TRACE)
// diagnostics builds only (tools/wabl_libssm_hip.so, tools/w4trace_libssm_hip.so - never lib/libssm_hip.so, whose exports are exactly
// include/ssm_hip.h): 7 device counters that the ablation build fills when $SSM_WINO4_ABL has bit 32 set / the wave timeline
extern "C" int ssm_wino4_debug_buffer(unsigned long long *dev_counters) {
    g_w4dbg.store(dev_counters);
    return SSM_OK;
}
#endif

// 1 when the plan should run this 3x3 layer as F(4x4,3x3) rather than F(2x2,3x3).  Measured per layer at 736x1280, batch 7
// (tools/bench_layers_wino.py, W4=1 against the default): F(4x4) is 5-35 % faster everywhere except on the 23x40 maps (32 tiles of 16
// pixels per workgroup: half of every tile row is overshoot; 0.21 vs 0.15 ms).  The two cost models are not calibrated against each
// other, so the rule is stated directly.
extern "C" int ssm_wino4_preferred(int Cin, int Cout, int B, int H, int W, int ups) {
    if (Cin % 4 || Cout % 32 || Cin <= 0 || Cout <= 0) return 0;
    if ((long long)H * W < 2048) return 0;
    return 1;
}

extern "C" int ssm_wino4_force_kind(int kind) {
    g_force_w4kind.store(kind >= 0 && kind < NW4KIND ? kind : -1);
    return NW4KIND;
}

extern "C" size_t ssm_wino4_packed_weight_floats(int Cout, int Cin) { return (size_t)(Cout / 32) * (size_t)Cin * 9 * 32 * 4; }

extern "C" int ssm_wino4_pack_weights(const float *w, const float *bias, float *wp, float *bp, int Cout, int Cin, void *stream) {
    SSM_REQUIRE(w && bias && wp && bp, "wino4 pack_weights: null pointer");
    SSM_REQUIRE(Cout > 0 && Cin > 0 && Cout % 32 == 0, "wino4 pack_weights: bad sizes (Cout a multiple of 32)");
    const long long total = (long long)ssm_wino4_packed_weight_floats(Cout, Cin);
    const int nbias = Cout;
    const long long n = total > nbias ? total : nbias;
    SSM_LAUNCH(wino4_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, bias, wp, bp, Cout, Cin,
                       total, nbias);
    return ssm::check_launch("ssm_wino4_pack_weights");
}

extern "C" int ssm_wino4_conv2d_add_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                                        ssm_view pool, ssm_view add, int add_div, int B, int H, int W, int Cout, float slope, int flags,
                                        void *stream) {
    int kind = 0;
    const int rc = ssm_wino4_plan(C1 + C2, Cout, B, H, W, 0, &kind, nullptr, nullptr);
    if (rc != SSM_OK) return rc;
    W4Params p;
    const int rf = w4fill(p, x1, C1, x2, C2, w_packed, bias_packed, y, pool, add, add_div, B, H, W, Cout, slope, flags, W);
    if (rf != SSM_OK) return rf;
    return w4dispatch<false>(kind, p, B, (hipStream_t)stream);
}

// ---- the sub-pixel form of conv3x3(upsample2x(cat[a, b])) for the INTERIOR of the map + the ordinary fused-upsample kernel for its border ring
// (scripts/models/flow_computation.py:244-247; r5).  conv3x3(upsample2x(x)) is linear in x: output parity (a, b) is a 3x3 convolution of the
// LOW-res map with the effective filter M_a W M_b^T (ssm_amd/subpixel.py), so away from the border the layer is a plain 3x3 convolution with
// 4 Cout outputs and a pixel-shuffle store - no upsampled patch to expand in LDS, and a 32-cout full-resolution layer (conv11a) becomes a 128-cout
// half-resolution one that runs in the 64-cout form.  At the border the bilinear rule clamps while the convolution zero-pads: those workgroup
// tiles (16 x 32 output pixels) stay with the fused-upsample kernel.
extern "C" int ssm_wino4_conv2d_shuffle_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                                            int B, int H, int W, int Cout4, float slope, int flags, void *stream) {
    SSM_REQUIRE(Cout4 > 0 && Cout4 % 64 == 0, "wino4 conv_shuffle: 4 Cout (%d) must be a multiple of 64", Cout4);
    int kind = 0;
    const int rc = ssm_wino4_plan(C1 + C2, Cout4, B, H, W, 0, &kind, nullptr, nullptr);
    if (rc != SSM_OK) return rc;
    if (kW4Info[kind].bn != 64) kind = Y4A_;
    W4Params p;
    const ssm_view none = {nullptr, 0, 0, 0};
    const int rf = w4fill(p, x1, C1, x2, C2, w_packed, bias_packed, y, none, none, 1, B, H, W, Cout4, slope, flags, W);
    if (rf != SSM_OK) return rf;
    switch (kind) {
        case Y4A_: return w4launch<Y4A, false, true>(p, B, (hipStream_t)stream);
        case Y4B_: return w4launch<Y4B, false, true>(p, B, (hipStream_t)stream);
        default: return w4launch<Y4C, false, true>(p, B, (hipStream_t)stream);
    }
}

extern "C" int ssm_wino4_conv2d_ups_border_fwd(ssm_view a, int C1, ssm_view b, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                                               int B, int H, int W, int Cout, float slope, int flags, void *stream) {
    SSM_REQUIRE(H % 2 == 0 && W % 2 == 0, "wino4 conv_ups_border: the output of a x2 upsample has even H, W (got %dx%d)", H, W);
    SSM_REQUIRE(C1 + C2 > 0 && (C1 + C2) % 4 == 0 && Cout % 32 == 0, "wino4 conv_ups_border: Cin a multiple of 4, Cout of 32");
    W4Params p;
    const ssm_view none = {nullptr, 0, 0, 0};
    const int rf = w4fill(p, a, C1, b, C2, w_packed, bias_packed, y, none, none, 1, B, H, W, Cout, slope, flags, W / 2);
    if (rf != SSM_OK) return rf;
    p.border = 1;
    return w4launch<X4A, true>(p, B, (hipStream_t)stream);          // (the 16 x 32-pixel tile: SSM_WINO4_BORDER_TH / _TW of include/ssm_hip.h)
}

extern "C" int ssm_wino4_conv2d_ups_add_fwd(ssm_view a, int C1, ssm_view b, int C2, const float *w_packed, const float *bias_packed, ssm_view y,
                                            ssm_view add, int add_div, int B, int H, int W, int Cout, float slope, int flags, void *stream) {
    int kind = 0;
    SSM_REQUIRE(H % 2 == 0 && W % 2 == 0, "wino4 conv_ups: the output of a x2 upsample has even H, W (got %dx%d)", H, W);
    const int rc = ssm_wino4_plan(C1 + C2, Cout, B, H, W, 1, &kind, nullptr, nullptr);
    if (rc != SSM_OK) return rc;
    W4Params p;
    const ssm_view none = {nullptr, 0, 0, 0};
    const int rf = w4fill(p, a, C1, b, C2, w_packed, bias_packed, y, none, add, add_div, B, H, W, Cout, slope, flags, W / 2);
    if (rf != SSM_OK) return rf;
    return w4dispatch<true>(kind, p, B, (hipStream_t)stream);
}
