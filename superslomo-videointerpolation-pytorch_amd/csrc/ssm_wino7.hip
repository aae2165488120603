// 7x7 convolution as a BLOCKED two-dimensional Winograd form on the CDNA4 fp32 matrix cores (v_mfma_f32_16x16x4_f32), all arithmetic fp32.
//
// Same operator as the k = 7 case of ssm_conv.hip / ssm_wino1d.hip (layers.conv of the reference, scripts/models/layers.py:21-33: stride-1
// 'same' cross-correlation, zero padding, bias, LeakyReLU; fused 2x2 mean, scripts/models/layers.py:60-63; pre-activation addend for the
// hoisted part of stage 2's conv1a) - the first two layers of both U-Nets (scripts/models/flow_computation.py:36-45), 32 output channels
// at full resolution, 24 % of the direct-form multiply-adds of a pair.
//
// The one-dimensional form F(2,7) of ssm_wino1d.hip needs 8 x 7 = 56 multiplies per two outputs (28 per output; the direct form 49).  A
// two-dimensional F(m x m, 7x7) is out of reach in fp32 (m + 6 points per axis) - but a 7x7 filter is 2x2 BLOCKS of 4x4 taps (padded with
// one zero row / column), and for blocks of r = 4 taps on tiles of m = 4 outputs the four blocks of a tile read input windows that are
// whole tiles apart:
//
//      y(T) = sum_{b = (by,bx)} g_b * x[window at 4 (T + b) - 3]   =   A^T [ sum_cin sum_b U_b .* V(T + b) ] A
//      U_b = G g_b G^T (7x7 frequencies),   V(P) = B^T d(P) B,   d(P) = the 7x7 input window at rows / columns 4 P - 3 .. 4 P + 3
//
// i.e. ONE input transform per tile position P serves the four (block, neighbouring tile) pairs that read it, the four blocks are four
// more terms of the channel sum (GEMM depth K = 4 Cin), and a 4x4 output tile costs 4 x 49 multiplies per (cin, cout): 12.25 per output
// - 2.3 x fewer matrix-core cycles than F(2,7), 4 x fewer than the direct form.  Seven points {0, +-1, +-2, 1/2, inf} per axis: every
// constant of B^T and A^T is dyadic (exact in fp32); a 32-channel layer sits 2.6e-6 rms / 2.9e-5 max from float64 at unit output scale
// (F(2,7): 1.0e-6 / 6.7e-6; F(4x4,3x3) of ssm_wino4.hip with 512 channels: 1.4e-6 / 1.1e-5) and the pair -> frame path at 736x1280 is
// unchanged within its fp32 noise (2.32e-4 from float64 in the direct, the 1-D and this form: profiles/README.md r8).
//
// GEMM view per frequency f: M_f[cout][tile] = sum_{cin, b} U_f[cout][cin, b] V_f[cin][tile + b] on v_mfma_f32_16x16x4_f32 with the four
// blocks of ONE input channel as the k-step (A = 16 couts x 4 blocks, B = 4 blocks x 16 tiles: lane group q reads the transformed window
// of its tile shifted by block q - a per-lane constant offset into V).  A wave owns 16 couts x 16 tiles for all 49 frequencies (49
// accumulators of 4 registers), so the output transform A^T M A is lane-local and a lane finishes whole 4x4 pixel tiles.  One workgroup
// of four waves (32 couts x 32 tiles of 4x4 pixels) per CU, one wave per SIMD: the fp32 MFMA and vector instructions share the issue
// port (profiles/DESIGN_history_r1-r3.md 3.2g), two waves on a SIMD only serialise.
//
// Input transform with the overlap of neighbouring windows used (7 rows at stride 4): a ROW pass per (patch row, position column) - each
// patch row is transformed once although two vertically neighbouring windows read it - leaves X [h][row][px][4] in LDS, a COLUMN pass per
// (position, pair of column-frequencies) reads seven rows of X and writes V [14 quads][positions][4].  20 vector operations per 7-point
// pass (monic rows of B^T, factored through first differences; even / odd parts shared by a +-p pair).
//
// Pipeline, one barrier per input channel:  [barrier]  DMA U(c+1), patch(c+3) | row pass (c+2) | column pass (c+1) | 49 MFMAs of channel c,
// the transform pieces and the LDS-DMA instructions placed in the slots behind the MFMAs (everything double-buffered; 97 KiB of LDS).
#include "ssm_common.h"
#include "ssm_wino7_pack.h"

#include <atomic>
#include <mutex>
#include <type_traits>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef W7_VPAD
#define W7_VPAD 0            // 1: position rows of V padded against the operand reads' bank conflict - measured 1 % SLOWER (profiles/r11d_wino7_vpad_ab.txt): off
#endif

namespace {

struct W7Params {
    const float *src;
    long long sb, sc;    // batch / channel stride
    int sh;              // row stride
    int Cin;
    const float *wpk;    // U, [Cout/32][Cin][14][4][32][4]
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
    int vec;             // 1: outputs / addend / pooled outputs may be moved as aligned 16- / 8-byte pieces (checked on the host)
    const float *add;    // optional pre-activation addend [B / adiv][Cout][H][W]
    long long asb, asc;
    int ash, adiv;
    unsigned long long *dbg;   // tuning build (-DW7_TRACE) only: per-wave phase sums of workgroup 0
};

// 2 cout halves x 2 tile groups = 4 waves; a tile group = GTX x GTY tiles of 4x4 pixels (16 tiles), the groups sit WTY x WTX.
template <int GTX_, int WTY_, int WTX_>
struct W7Cfg {
    static constexpr int GTX = GTX_, GTY = 16 / GTX_, WTY = WTY_, WTX = WTX_;
    static constexpr int NTX = GTX * WTX, NTY = GTY * WTY;                     // tiles per workgroup, by axis
    static constexpr int TH = 4 * NTY, TW = 4 * NTX;                           // output pixels per workgroup
    static constexpr int NPX = NTX + 1, NPY = NTY + 1, NP = NPX * NPY;         // window positions (one more than tiles per axis)
    static constexpr int PH = TH + 7, PW = TW + 8;                             // patch rows y0-3 .. y0+TH+3, columns x0-4 .. x0+TW+3
    static constexpr int SHIFT = 3;                                            // floats: window P starts at patch column 4 P + 1 + SHIFT (16-byte aligned)
    static constexpr int NFQ = SSM_W7_NFQ;                                     // quads of frequencies (7 row-frequencies x 2)
    static constexpr int USZ = NFQ * 4 * 32 * 4;                               // filter floats per input channel
    // V [quad][position][4]: the 16 tiles of a lane group read GTY rows of GTX consecutive positions (+ the block's shift); with the rows
    // NPX = 9 units apart the 16th lane's 16 bytes fall on the first lane's banks (a 2-way conflict on every operand read of the matrix
    // loop: LDS bank-conflict share 0.39, profiles/r10e_wino_sq_counters.txt) - position rows are stored NPV units apart, NPV = GTX mod 16
    static constexpr int NPV = !W7_VPAD ? NPX : GTX == 8 ? 24 : GTX == 4 ? 20 : NPX;
    static constexpr int NPT = (NPY - 1) * NPV + NPX;                          // units per quad
    static constexpr int VSZ = NFQ * NPT * 4;                                  // transformed windows
    static constexpr int XSZ = 2 * PH * NPX * 4;                               // row-pass results, [h][row][px][4]
    static constexpr int NDQ = PH * PW / 4, NGP = (NDQ + 63) / 64;             // 16-byte pieces / 1-KiB DMA groups of the patch
    static constexpr int PCAP = NGP * 256 + 256;                               // floats per patch buffer (tail of the last group + shift)
    static constexpr int NGU = USZ / 256;                                      // 1-KiB DMA groups of the filter
    static constexpr int NIU = (NGU + 3) / 4, NIP = (NGP + 3) / 4;             // DMA instructions per wave and channel
    static constexpr int NRU = PH * NPX;                                       // row-pass units
    static constexpr int UOFF = 0, VOFF = 2 * USZ, XOFF = VOFF + 2 * VSZ, POFF = XOFF + 2 * XSZ;
    static constexpr int BYTES = (POFF + 2 * PCAP) * 4;
    static_assert(WTY * WTX == 2 && (GTX == 4 || GTX == 8 || GTX == 16), "two tile groups of 16 tiles");
    static_assert(NRU <= 256 && NP <= 64, "one row-pass unit per thread, one position per lane in the column pass");
    static_assert(PW % 4 == 0 && USZ % 256 == 0 && NGU % 4 == 0 && NIP == 1, "whole DMA groups; one patch piece per wave");
    static_assert(BYTES <= 160 * 1024, "LDS budget");
};

// Seven points 0, +1, -1, +2, -2, 1/2, inf.  B^T in the monic form: row p = the coefficients of M(x) / (x - p), M(x) = N(x) (x - 1/2),
// N(x) = x (x^2 - 1) (x^2 - 4) - so every row but p = 1/2 factors through the differences e_i = d_{i+1} - d_i / 2, on which the rows of
// the symmetric points share their even / odd parts (20 vector operations per 7-point pass):
//   e_i = d_{i+1} - d_i / 2 (i = 0..5)
//   p = 0:   e4 - 5 e2 + 4 e0          p = +-1: (e4 - 4 e2) +- (e3 - 4 e1)          p = +-2: (e4 - e2) +- 2 (e3 - e1)
//   p = 1/2: d5 - 5 d3 + 4 d1          inf:     e5 - 5 e3 + 4 e1
// The scan over point sets (tests/emulate_winograd_7x7_blocked.py, fp32 operation by operation): this set 2.6e-6 rms / 2.9e-5 max at
// 32 channels and unit output scale; the symmetric set {0, +-1, +-1/2, +-2} 3.3e-6 / 4.7e-5 (its whole error sits on the first output of
// a tile, the plain sum of all 49 frequencies); sets with a point beyond 2 or two points inside 1/2 are 3-100 x worse.
__device__ __forceinline__ void w7_bt_e(const float (&d)[7], float (&e)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) e[i] = d[i + 1] - 0.5f * d[i];
}
__device__ __forceinline__ void w7_bt_a(const float (&e)[6], float (&c)[7]) {          // frequencies 0, 1, 2
    c[0] = (e[4] - 5.f * e[2]) + 4.f * e[0];
    const float e1 = e[4] - 4.f * e[2], o1 = e[3] - 4.f * e[1];
    c[1] = e1 + o1;
    c[2] = e1 - o1;
}
__device__ __forceinline__ void w7_bt_b(const float (&d)[7], const float (&e)[6], float (&c)[7]) {          // frequencies 3 .. 6
    const float e2 = e[4] - e[2], o2 = e[3] - e[1];
    c[3] = e2 + 2.f * o2;
    c[4] = e2 - 2.f * o2;
    c[5] = (d[5] - 5.f * d[3]) + 4.f * d[1];
    c[6] = (e[5] - 5.f * e[3]) + 4.f * e[1];
}

// A^T = [1 1 1 1 1 1 0; 0 1 -1 2 -2 1/2 0; 0 1 1 4 4 1/4 0; 0 1 -1 8 -8 1/8 1]
__device__ __forceinline__ void w7_at(float m0, float m1, float m2, float m3, float m4, float m5, float m6, float (&y)[4]) {
    const float s1 = m1 + m2, t1 = m1 - m2, s2 = m3 + m4, t2 = m3 - m4;
    y[0] = ((m0 + s1) + s2) + m5;
    y[1] = (t1 + 2.f * t2) + 0.5f * m5;
    y[2] = (s1 + 4.f * s2) + 0.25f * m5;
    y[3] = ((t1 + 8.f * t2) + 0.125f * m5) + m6;
}

// ---- epilogue: Y = A^T M A per accumulator register (4 couts per lane), + bias, addend, LeakyReLU, stores, fused 2x2 mean.
// cu0: first cout of the wave's 16-cout block (this lane holds couts cu0 + 4 q + r), (px, py): the lane's 4x4 output tile.
// FAST: every tile of the wave lies inside the map and rows move as aligned 16-byte pieces - the element-wise edge path (a branch per
// element even when no lane takes it) is not compiled in.  HOLD (FAST with an addend): the 16 output rows of the lane's four couts stay in
// registers and are stored after the last addend row has been used.  The stores are inline assembly (scalar base + 32-bit lane offset),
// invisible to the compiler's wait-count pass: a compiler-placed wait for an addend load also waits for every store issued before it, so
// stores between the loads cost store round trips (tools/bench_layers_wino7.py, conv1a with the addend: 2.11 -> 2.01 ms at batch 14).
// NR couts of the lane starting at accumulator element r0 (the 4-wave kernel: all four; a wave of the frequency-split kernel finishes two);
// tile(r, y): the 4x4 tile of element r before the bias.
// zin (HOLD only): the NR x 4 addend rows already requested by the caller (the split kernel asks for them behind the barrier of its last
// channel: their latency runs under that channel's MFMAs), or nullptr.
template <bool FAST, bool HOLD, int NR, class Tile>
__device__ __forceinline__ void w7_epilogue(const W7Params &p, Tile tile, const float (&bv)[4], int b, int cu0, int r0, int q, int px, int py,
                                            const f32x4 *zin = nullptr) {
    const float sl = p.lrelu ? p.slope : 1.f;
    float *dstb = p.dst + (long long)b * p.dsb;
    float *poolb = p.pool ? p.pool + (long long)b * p.psb : nullptr;
    const unsigned pb = 4u * ((unsigned)(4 * q) * (unsigned)p.dsc + (unsigned)py * (unsigned)p.dsh + (unsigned)px);
    const unsigned qb = 4u * ((unsigned)(4 * q) * (unsigned)p.psc + (unsigned)(py >> 1) * (unsigned)p.psh + (unsigned)(px >> 1));
    const bool vok = FAST || (py + 4 <= p.H && px + 4 <= p.W && p.vec);          // whole tile inside the map, rows as aligned 16-byte pieces
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
    // the addend rows of cout r + 1 are requested while cout r is transformed (four independent 16-byte loads, one cout ahead): their
    // latency runs beside the output transform instead of in front of each cout's stores
    constexpr int ZN = HOLD ? NR : 2;          // HOLD: all addend rows are requested up front (no store stands between them and their use)
    f32x4 zadd[ZN][4];
    const bool hasadd = p.add != nullptr;          // uniform (addb is a per-lane pointer: a test of it compiles to a divergent branch)
    const bool zvec = hasadd && vok;
    auto zload = [&](int r) {
        if (zvec) {
#pragma unroll
            for (int i = 0; i < 4; ++i) zadd[r % ZN][i] = *(const f32x4 *)(addb + (long long)(cu0 + r) * p.asc + (long long)i * p.ash);
        }
    };
    if (HOLD && zin) {
#pragma unroll
        for (int rr = 0; rr < NR; ++rr)
#pragma unroll
            for (int i = 0; i < 4; ++i) zadd[(r0 + rr) % ZN][i] = zin[rr * 4 + i];
    } else {
        zload(r0);
        if (HOLD) {
#pragma unroll
            for (int rr = 1; rr < NR; ++rr) zload(r0 + rr);
        }
    }
    f32x4 yk[HOLD ? NR : 1][4];
    auto emit = [&](int r, const float (&y)[4][4]) {
        const int cu = cu0 + r;
        float *bp = dstb + (long long)cu * p.dsc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (vok) {
                st4(bp + (long long)i * p.dsh, pb, f32x4{y[i][0], y[i][1], y[i][2], y[i][3]});
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)      // edge path: one scalar base per cout (16 (row, element) bases at once overflow the scalar registers)
                    if (py + i < p.H && px + e < p.W) st1(bp, pb + 4u * ((unsigned)i * (unsigned)p.dsh + (unsigned)e), y[i][e]);
            }
        }
        if (poolb) {
            // 2x2 mean, vertical pairs first then the horizontal pair (the association of the direct kernel); H, W even (host check)
            float *qp = poolb + (long long)cu * p.psc;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float o0 = ((y[2 * i][0] + y[2 * i + 1][0]) + (y[2 * i][1] + y[2 * i + 1][1])) * 0.25f;
                const float o1 = ((y[2 * i][2] + y[2 * i + 1][2]) + (y[2 * i][3] + y[2 * i + 1][3])) * 0.25f;
                const bool rok = FAST || py + 2 * i < p.H;
                if (FAST || (rok && px + 4 <= p.W && p.vec)) st2(qp + (long long)i * p.psh, qb, f32x2{o0, o1});
                else if (rok) {
                    if (px + 2 <= p.W) st1(qp, qb + 4u * (unsigned)i * (unsigned)p.psh, o0);
                    if (px + 4 <= p.W) st1(qp, qb + 4u * ((unsigned)i * (unsigned)p.psh + 1u), o1);
                }
            }
        }
    };
#pragma unroll
    for (int r = r0; r < r0 + NR; ++r) {
        const int cu = cu0 + r;          // uniform; this lane's cout = cu + 4 * q
        if (!HOLD && r + 1 < r0 + NR) zload(r + 1);
        float y[4][4];
        tile(r, y);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) y[i][e] += bv[r];
        if (hasadd) {
            const float *ap = addb + (long long)cu * p.asc;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (vok) {
                    const f32x4 z = zadd[r % ZN][i];
                    y[i][0] += z[0];
                    y[i][1] += z[1];
                    y[i][2] += z[2];
                    y[i][3] += z[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (py + i < p.H && px + e < p.W) y[i][e] += ap[(long long)i * p.ash + e];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) y[i][e] = fmaxf(y[i][e], y[i][e] * sl);
        if (HOLD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) yk[r - r0][i] = f32x4{y[i][0], y[i][1], y[i][2], y[i][3]};
        } else {
            emit(r, y);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (HOLD) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const float y[4][4] = {{yk[k][0][0], yk[k][0][1], yk[k][0][2], yk[k][0][3]}, {yk[k][1][0], yk[k][1][1], yk[k][1][2], yk[k][1][3]},
                                   {yk[k][2][0], yk[k][2][1], yk[k][2][2], yk[k][2][3]}, {yk[k][3][0], yk[k][3][1], yk[k][3][2], yk[k][3][3]}};
            emit(r0 + k, y);
        }
    }
}

// Y = A^T M A of accumulator element r, all 49 frequencies in one wave (the 4-wave kernel)
__device__ __forceinline__ void w7_tile_full(const f32x4 (&acc)[49], int r, float (&y)[4][4]) {
    float t[4][7];                   // A^T M: over the row-frequencies, for every column-frequency
#pragma unroll
    for (int cf = 0; cf < 7; ++cf) {
        float y4[4];
        w7_at(acc[cf][r], acc[7 + cf][r], acc[14 + cf][r], acc[21 + cf][r], acc[28 + cf][r], acc[35 + cf][r], acc[42 + cf][r], y4);
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i][cf] = y4[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) w7_at(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], t[i][6], y[i]);
}

#ifndef W7_ABL
#define W7_ABL 0             // tuning builds (make w7alt W7FLAGS=-DW7_ABL=n): 1 no LDS-DMA, 2 no transform in the steady-state loop; 4 no epilogue
#endif
#ifndef W7_UNROLL2
#define W7_UNROLL2 0
#endif
#ifndef W7_INTERLEAVE
#define W7_INTERLEAVE 1      // the transform pieces and the DMA issue ride in the slots of the matrix loop (0: as blocks in front of it)
#endif

template <class C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino7_kernel(const W7Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PH = C::PH, PW = C::PW, NPX = C::NPX, NP = C::NP, NFQ = C::NFQ;

    const int tid = threadIdx.x;
    const int lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wid & 1, tg = wid >> 1;          // cout half, tile group of this wave

    int id = ssm_xcd_tile(blockIdx.x, gridDim.x);
    const int nb = id % p.NB;
    id /= p.NB;
    const int tx = id % p.tilesX;
    id /= p.tilesX;
    const int ty = id % p.tilesY;
    const int b = id / p.tilesY;
    const int x0 = tx * C::TW, y0 = ty * C::TH;

    const float *pbase = p.src + (long long)b * p.sb + (long long)(y0 - 3) * p.sh + (x0 - 4);
    const float *wbase = p.wpk + (long long)nb * p.Cin * C::USZ;

    // per-lane source offset (bytes) of the patch piece this wave brings per channel (piece = 16 bytes of a patch row)
    int poff;
    {
        const int qq = wid * 64 + lane;
        if (qq < C::NDQ) {
            // Rows below the bottom zero frame and 16-byte pieces right of the padded row are read from the frame's last row / last piece
            // instead (zeros): a tile that overshoots the map, and the bottom window's row H + 3 (the zero tap ky = 7, which B^T d B still
            // mixes into every frequency), never bring in what lies behind the plane - another plane's pixels, or the caller's unzeroed
            // memory behind the last one (NaN there would reach valid outputs).  Costs nothing in the loop: the offsets are per-lane constants.
            const int r = qq / (PW / 4), j = qq - r * (PW / 4);
            const int re = min(r, p.H + (SSM_PADY - 1) - (y0 - 3)), fe = min(4 * j, ((p.W + 2 * SSM_PADX + 3) & ~3) - 4 - x0);
            poff = (re * p.sh + fe) * 4;
        } else {
            poff = 0;          // tail of the last 1-KiB piece: lands in the buffer's padding
        }
    }
    const int uoff = lane * 16;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;

    auto dma = [](const float *base, int voff_bytes, unsigned m0v) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(base), "s"(m0v) : "memory", "m0");
    };
    // k-th filter piece of this wave for channel c into stage buf (= c & 1; k = 0 .. NIU-1); the patch piece of channel c into buffer buf
    auto dma_u = [&](int c, int k, int buf) {
        const int g = 4 * k + wid;
        const float *base = wbase + (long long)c * C::USZ + g * 256;
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(C::UOFF + buf * C::USZ) * 4u + (unsigned)g * 1024u);
        dma(base, uoff, m0v);
    };
    auto dma_p = [&](int c, int buf) {
        if (wid < C::NGP) {
            const float *base = pbase + (long long)c * p.sc;
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(C::POFF + buf * C::PCAP + C::SHIFT) * 4u + (unsigned)wid * 1024u);
            dma(base, poff, m0v);
        }
    };

    f32x4 acc[49];
#pragma unroll
    for (int f = 0; f < 49; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};

    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = p.bias[nb * 32 + cb * 16 + 4 * q + r];

    // ---- row pass: unit = (patch row, position column); 2 x 16-byte reads, 21 vector operations, 2 x 16-byte writes -------------------
    // (no predicates in the loop: the threads / lanes beyond the last unit repeat it - same reads, same values written to the same place.
    // Exec-masked pieces between the MFMAs split the loop into basic blocks, and the register allocator then copied all 196 accumulator
    // registers once per channel: 195 v_accvgpr_mov per 49 MFMAs.)
    const int r_unit = min(tid, C::NRU - 1);
    const int r_row = r_unit / NPX, r_px = r_unit - r_row * NPX;
    const int r_src = C::POFF + C::SHIFT + r_row * PW + 4 * r_px + 1;           // floats; 16-byte aligned
    const int r_dst = C::XOFF + (r_row * NPX + r_px) * 4;                        // plane h = 0; plane 1 at + PH * NPX * 4
    // ---- column pass: unit = (position, pair j of column-frequencies), j = wave; 7 x 8-byte reads, 2 x 21 operations, 7 x 8-byte writes
    const int c_pos = min(lane, NP - 1);
    const int c_py = c_pos / NPX, c_px = c_pos - c_py * NPX;
    const int c_src = C::XOFF + ((wid >> 1) * PH * NPX + (4 * c_py) * NPX + c_px) * 4 + (wid & 1) * 2;
    const int c_dst = C::VOFF + ((wid >> 1) * C::NPT + c_py * C::NPV + c_px) * 4 + (wid & 1) * 2;    // + rf * 2 * NPT * 4 per row-frequency

    // ---- operand bases of the matrix loop (f32x4 units): U of (quad, block q, cout cb*16 + l15), V of (quad, position of the tile + block q)
    const f32x4 *lds4 = (const f32x4 *)lds;
    const int gx = l15 % C::GTX, gy = l15 / C::GTX;
    const int Tx = (tg % C::WTX) * C::GTX + gx, Ty = (tg / C::WTX) * C::GTY + gy;
    const int aBase = C::UOFF / 4 + q * 32 + cb * 16 + l15;
    const int bBase = C::VOFF / 4 + (Ty + (q >> 1)) * C::NPV + Tx + (q & 1);

    const int n = p.Cin;
    f32x4 ra[2];                       // row pass: the window row
    float rc[7], re[6];
    f32x2 cx[7];                       // column pass: seven rows of X (two column-frequencies)
    float cv0[7], cv1[7], ce0[6], ce1[6];
    f32x4 a[3], bq[3];

    // The transform of a chunk is cut into pieces that ride behind the MFMAs (a piece = a few LDS instructions or ~10 vector operations):
    //   piece 0        row pass: the two reads            piece 1 .. 3   row pass: arithmetic (differences | f 0..2 | f 3..6), piece 4: writes
    //   piece 5, 6     column pass: the seven reads       piece 7 .. 12  column pass: arithmetic, two frequencies x 3; piece 13, 14: writes
    auto tpiece = [&](auto J, int k, bool doR, bool doC, int rbuf, int xrbuf, int xcbuf, int vbuf) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;          // this wave's pair of column-frequencies (j = 3: the single frequency 6)
        if (k == 0 && doR) {
            const float *src = lds + r_src + rbuf * C::PCAP;
            ra[0] = *(const f32x4 *)src;
            ra[1] = *(const f32x4 *)(src + 4);
        }
        if (k >= 1 && k <= 3 && doR) {
            const float d[7] = {ra[0][0], ra[0][1], ra[0][2], ra[0][3], ra[1][0], ra[1][1], ra[1][2]};
            if (k == 1) w7_bt_e(d, re);
            if (k == 2) w7_bt_a(re, rc);
            if (k == 3) w7_bt_b(d, re, rc);
        }
        if (k == 4 && doR) {
            float *dst = lds + r_dst + xrbuf * C::XSZ;
            *(f32x4 *)dst = f32x4{rc[0], rc[1], rc[2], rc[3]};
            *(f32x4 *)(dst + PH * NPX * 4) = f32x4{rc[4], rc[5], rc[6], 0.f};
        }
        if ((k == 5 || k == 6) && doC) {
            const float *src = lds + c_src + xcbuf * C::XSZ;
#pragma unroll
            for (int i = (k == 5 ? 0 : 4); i < (k == 5 ? 4 : 7); ++i) cx[i] = *(const f32x2 *)(src + i * NPX * 4);
        }
        if (k >= 7 && k <= 12 && doC) {
            const int which = (k - 7) / 3, part = (k - 7) % 3;
            if (which == 0 || j < 3) {
                const float d[7] = {cx[0][which], cx[1][which], cx[2][which], cx[3][which], cx[4][which], cx[5][which], cx[6][which]};
                if (which == 0) {
                    if (part == 0) w7_bt_e(d, ce0);
                    if (part == 1) w7_bt_a(ce0, cv0);
                    if (part == 2) w7_bt_b(d, ce0, cv0);
                } else {
                    if (part == 0) w7_bt_e(d, ce1);
                    if (part == 1) w7_bt_a(ce1, cv1);
                    if (part == 2) w7_bt_b(d, ce1, cv1);
                }
            }
        }
        if ((k == 13 || k == 14) && doC) {
            float *dst = lds + c_dst + vbuf * C::VSZ;
#pragma unroll
            for (int rf = (k == 13 ? 0 : 4); rf < (k == 13 ? 4 : 7); ++rf) {
                if (j < 3) *(f32x2 *)(dst + rf * 2 * C::NPT * 4) = f32x2{cv0[rf], cv1[rf]};
                else dst[rf * 2 * C::NPT * 4] = cv0[rf];
            }
        }
    };
    // slot of the matrix loop (0 .. 48, behind MFMA m) -> piece: the row pass in slots 0 .. 10, the column pass in slots 12 .. 40
    auto piece_of_slot = [](int m) constexpr -> int {
        switch (m) {
            case 0: return 0;
            case 4: return 1;
            case 6: return 2;
            case 8: return 3;
            case 10: return 4;
            case 12: return 5;
            case 14: return 6;
            case 19: return 7;
            case 21: return 8;
            case 23: return 9;
            case 25: return 10;
            case 27: return 11;
            case 29: return 12;
            case 32: return 13;
            case 34: return 14;
            default: return -1;
        }
    };
    // slot -> DMA instruction of this wave (NIU filter pieces, then the patch piece): slots 1, 3, 5, ...
    auto dma_of_slot = [](int m) constexpr -> int { return (m % 2 == 1 && m / 2 < C::NIU + 1) ? m / 2 : -1; };

    // ---- 49 MFMAs per input channel: 14 quads of frequencies (the odd quads hold three), operands of quad g+2 fetched behind the first MFMA
    // of quad g (a ring of three register sets); slot(m) behind the m-th MFMA of the iteration.  The loop is ROTATED by two quads: a wave
    // alone on its SIMD would wait out a full LDS latency behind every barrier (its first operands can only be fetched once the barrier has
    // released V), so the last two quads of channel c (7 MFMAs, operands fetched into ha / hb before the barrier) are issued behind the
    // barrier of channel c+1, in front of its own quads, while its first operand fetches are in flight.
    //   held: the previous channel left quads 12, 13 to do;  hold: leave this channel's quads 12, 13 to the next iteration
    f32x4 ha[2], hb[2];
    auto matrix = [&](int st, bool held, bool hold, auto slot) __attribute__((always_inline)) {
        const int ai = aBase + st * (C::USZ / 4), bi = bBase + st * (C::VSZ / 4);
        a[0] = lds4[ai];
        bq[0] = lds4[bi];
        a[1] = lds4[ai + 128];
        bq[1] = lds4[bi + C::NPT];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            if (held) acc[42 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[m >> 2][m & 3], hb[m >> 2][m & 3], acc[42 + m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            slot(m);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int g = 0; g < NFQ - 2; ++g) {
            const int cur = g % 3, nxt = (g + 2) % 3;
            const int rf = g >> 1, h = g & 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (h && e == 3) continue;          // (the odd quads hold three frequencies)
                const int f = rf * 7 + 4 * h + e;
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][e], bq[cur][e], acc[f], 0, 0, 0);
                if (e == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + 2 < NFQ - 2) {
                        a[nxt] = lds4[ai + (g + 2) * 128];
                        bq[nxt] = lds4[bi + (g + 2) * C::NPT];
                    } else {          // g = 10, 11: the operands of quads 12, 13
                        ha[g - (NFQ - 4)] = lds4[ai + (g + 2) * 128];
                        hb[g - (NFQ - 4)] = lds4[bi + (g + 2) * C::NPT];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                slot(7 + f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!hold) {
#pragma unroll
            for (int m = 0; m < 7; ++m) acc[42 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[m >> 2][m & 3], hb[m >> 2][m & 3], acc[42 + m], 0, 0, 0);
        }
    };

#ifdef W7_TRACE
    unsigned long long tph[4] = {0, 0, 0, 0};
    unsigned long long tk = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tk;
#define W7STAMP(i)                                                  \
    {                                                               \
        const unsigned long long tn = __builtin_amdgcn_s_memtime(); \
        tph[i] += tn - tk;                                          \
        tk = tn;                                                    \
    }
#else
#define W7STAMP(i)
#endif
    // ---- one iteration of the pipeline (c from -2): DMA U(c+1), patch(c+3) | row pass (c+2) | column pass (c+1) | matrix (c) ----------
    // PAR: c & 1 as a compile-time fact (the steady-state loop runs two iterations per trip: every buffer then sits at an immediate offset
    // from per-lane base addresses that never change - with a run-time parity each LDS access of the transform cost a vector add), or -1
    auto iter = [&](auto J, int c, auto STEADY, auto PAR) __attribute__((always_inline)) {
        constexpr bool steady = decltype(STEADY)::value;
        constexpr int par_ct = decltype(PAR)::value;
        const int par = par_ct >= 0 ? par_ct : (c & 1);
        const bool doM = steady || c >= 0;
        const bool doC = steady || (c + 1 >= 0 && c + 1 < n);
        const bool doR = steady || c + 2 < n;
        const bool doP = steady || c + 3 < n;
        W7STAMP(0)          // -> [0]: the iteration's work (matrix loop + transform pieces)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W7STAMP(1)          // -> [1]: waiting for this wave's LDS-DMA
        __syncthreads();
        W7STAMP(2)          // -> [2]: waiting at the barrier
        const int rbuf = par, xrbuf = par, xcbuf = par ^ 1, vbuf = par ^ 1;          // (c + 2) & 1 == c & 1
        auto slot = [&](int m) __attribute__((always_inline)) {
            const int d = dma_of_slot(m);
            if (d >= 0 && !(steady && (W7_ABL & 1))) {
                if (d < C::NIU) {
                    if (doC) dma_u(c + 1, d, par ^ 1);
                } else if (doP) {
                    dma_p(c + 3, par ^ 1);
                }
            }
            const int k = piece_of_slot(m);
            if (k >= 0 && !(steady && (W7_ABL & 2))) tpiece(J, k, doR, doC, rbuf, xrbuf, xcbuf, vbuf);
        };
        const bool held = steady || c >= 1, hold = steady || c + 1 < n;
        if (doM && W7_INTERLEAVE) {
            matrix(par, held, hold, slot);
        } else {
#pragma unroll
            for (int m = 0; m < 49; ++m) {
                slot(m);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (doM) matrix(par, held, hold, [](int) {});
        }
    };

    dma_p(0, 0);
    auto run = [&](auto J) __attribute__((always_inline)) {
        using T = std::true_type;
        using F = std::false_type;
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        using PR = std::integral_constant<int, -1>;
        iter(J, -2, F{}, P0{});
        iter(J, -1, F{}, P1{});
        int c = 0;
        if (n > 0) iter(J, c++, F{}, PR{});          // (the steady-state copy finishes the previous channel's last quads: from channel 1 on)
#if W7_UNROLL2          // two iterations per trip, buffer parity at compile time: saves ~8 vector adds per channel, but the register allocator then
                        // routes 5 of the 49 accumulators through VGPRs every trip (40 v_accvgpr_read / _write per channel) - off
        for (; c + 4 < n; c += 2) {
            iter(J, c, T{}, P0{});
            iter(J, c + 1, T{}, P1{});
        }
#else
        for (; c + 3 < n; ++c) iter(J, c, T{}, PR{});
#endif
        for (; c < n; ++c) iter(J, c, F{}, PR{});
    };
    if (wid == 0) run(std::integral_constant<int, 0>{});
    else if (wid == 1) run(std::integral_constant<int, 1>{});
    else if (wid == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 3>{});

    if ((W7_ABL & 4) && acc[0][0] != 12345.678f) return;
    W7STAMP(0)
    {
        const int px = x0 + 4 * Tx, py = y0 + 4 * Ty;
        const bool edge = !(py + 4 <= p.H && px + 4 <= p.W && p.vec);
        auto tile = [&](int r, float (&y)[4][4]) __attribute__((always_inline)) { w7_tile_full(acc, r, y); };
        if (__builtin_amdgcn_ballot_w64(edge) != 0) w7_epilogue<false, false, 4>(p, tile, bv, b, nb * 32 + cb * 16, 0, q, px, py);
        else if (p.add) w7_epilogue<true, true, 4>(p, tile, bv, b, nb * 32 + cb * 16, 0, q, px, py);
        else w7_epilogue<true, false, 4>(p, tile, bv, b, nb * 32 + cb * 16, 0, q, px, py);
    }
#ifdef W7_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W7STAMP(3)          // -> [3]: epilogue
    if (p.dbg && lane == 0 && (blockIdx.x % 64) == 0) {          // a sample of the workgroups
        for (int i = 0; i < 4; ++i) atomicAdd(p.dbg + wid * 8 + i, tph[i]);
        atomicAdd(p.dbg + wid * 8 + 4, tk - tstart);
        atomicAdd(p.dbg + wid * 8 + 5, 1ULL);
    }
#endif
}

// ---- the frequency-split form (r5): EIGHT waves, two per SIMD -----------------------------------------------------------------------
// The pair of waves on a SIMD (w4 and w4 + 4) shares the 16 couts x 16 tiles of one wave of wino7_kernel and splits its 49 frequencies by
// column-frequency: half 0 owns cf 0..3 (the even quads of U / V: 28 accumulators), half 1 cf 4..6 (the odd quads: 21).  Same workgroup
// tile, same LDS layout and the same pipeline as above; what changes is who issues what:
//      half 0:  28 MFMAs + the chunk's LDS-DMA (filter + patch) + the ROW pass (one unit per thread, 256 threads)
//      half 1:  21 MFMAs + the COLUMN pass (wave w4 = pair w4 of column-frequencies, one position per lane)
// (half 1 is the longer role, 2200 against 1575 cycles of work per channel - but dealing the column pass over seven waves, one
// column-frequency each, raised the instruction count and measured 2 % slower: profiles/r11v_wino7_split.txt)
// so each wave carries about half of the vector / LDS instructions the lone wave carried, and the two instruction streams of a SIMD
// interleave: one wave's vector work issues while the other's MFMA occupies the matrix pipe (tools/mfma_valu_probe.py: two mixed waves
// reach 0.85 of the pipe where a lone mixed wave reaches 0.65).  112 / 84 accumulator registers: two waves fit a SIMD's 512.
// Output transform: t = A^T M over the row-frequencies is local to the half that owns the column-frequency; each half then forms its
// PARTIAL 4x4 tile over its own column-frequencies, the pair exchanges partial tiles through LDS and each wave finishes two of the lane's
// four couts:  y = (partial_0 + partial_1) + bias.
template <int FH>
__device__ __forceinline__ void w7s_partial(const f32x4 (&acc)[28], int r, float (&y)[4][4]) {
    constexpr int NE = FH ? 3 : 4;
    float t[4][NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        float y4[4];
        w7_at(acc[e][r], acc[NE + e][r], acc[2 * NE + e][r], acc[3 * NE + e][r], acc[4 * NE + e][r], acc[5 * NE + e][r], acc[6 * NE + e][r], y4);
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i][e] = y4[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (FH == 0) {          // column-frequencies 0, +1, -1, +2 of A^T
            const float m0 = t[i][0], s1 = t[i][1] + t[i][2], t1 = t[i][1] - t[i][2], m3 = t[i][3];
            y[i][0] = (m0 + s1) + m3;
            y[i][1] = t1 + 2.f * m3;
            y[i][2] = s1 + 4.f * m3;
            y[i][3] = t1 + 8.f * m3;
        } else {                // -2, 1/2, inf
            const float m4 = t[i][0], m5 = t[i][1], m6 = t[i][NE - 1];
            y[i][0] = m4 + m5;
            y[i][1] = 0.5f * m5 - 2.f * m4;
            y[i][2] = 4.f * m4 + 0.25f * m5;
            y[i][3] = (0.125f * m5 - 8.f * m4) + m6;
        }
    }
}

#ifndef W7S_ABL
#define W7S_ABL 0            // tuning builds (make w7alt W7FLAGS=-DW7S_ABL=n), steady-state loop only: 1 no LDS-DMA, 2 no transform pieces, 8 no operand re-reads
#endif
template <class C>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino7s_kernel(const W7Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PH = C::PH, PW = C::PW, NPX = C::NPX, NP = C::NP;
    static_assert(C::BYTES >= 8 * 2 * 4 * 64 * 16, "the partial-tile exchange of the epilogue fits the loop's LDS");

    const int tid = threadIdx.x;
    const int lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fh = wid >> 2, w4 = wid & 3;          // frequency half; waves w4 and w4 + 4 sit on SIMD w4
    const int cb = w4 & 1, tg = w4 >> 1;            // cout half, tile group of the pair

    int id = ssm_xcd_tile(blockIdx.x, gridDim.x);
    const int nb = id % p.NB;
    id /= p.NB;
    const int tx = id % p.tilesX;
    id /= p.tilesX;
    const int ty = id % p.tilesY;
    const int b = id / p.tilesY;
    const int x0 = tx * C::TW, y0 = ty * C::TH;

    const float *pbase = p.src + (long long)b * p.sb + (long long)(y0 - 3) * p.sh + (x0 - 4);
    const float *wbase = p.wpk + (long long)nb * p.Cin * C::USZ;

    int poff;          // (half 0) per-lane source offset of the wave's patch piece, overshoot clamped to the zero frame as in wino7_kernel
    {
        const int qq = w4 * 64 + lane;
        if (qq < C::NDQ) {
            const int r = qq / (PW / 4), j = qq - r * (PW / 4);
            const int re = min(r, p.H + (SSM_PADY - 1) - (y0 - 3)), fe = min(4 * j, ((p.W + 2 * SSM_PADX + 3) & ~3) - 4 - x0);
            poff = (re * p.sh + fe) * 4;
        } else {
            poff = 0;
        }
    }
    const int uoff = lane * 16;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;

    auto dma = [](const float *base, int voff_bytes, unsigned m0v) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(base), "s"(m0v) : "memory", "m0");
    };
    auto dma_u = [&](int c, int k, int buf) {
        const int g = 4 * k + w4;
        const float *base = wbase + (long long)c * C::USZ + g * 256;
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(C::UOFF + buf * C::USZ) * 4u + (unsigned)g * 1024u);
        dma(base, uoff, m0v);
    };
    auto dma_p = [&](int c, int buf) {
        if (w4 < C::NGP) {
            const float *base = pbase + (long long)c * p.sc;
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(C::POFF + buf * C::PCAP + C::SHIFT) * 4u + (unsigned)w4 * 1024u);
            dma(base, poff, m0v);
        }
    };

    f32x4 acc[28];          // [row-frequency][this half's column-frequency]: 7 x 4 (half 0) / 7 x 3 (half 1)
#pragma unroll
    for (int f = 0; f < 28; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};

    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = p.bias[nb * 32 + cb * 16 + 4 * q + r];

    // row pass (half 0): unit = (patch row, position column), one per thread of the half; column pass (half 1): unit = (position, pair w4)
    const int r_unit = min(w4 * 64 + lane, C::NRU - 1);
    const int r_row = r_unit / NPX, r_px = r_unit - r_row * NPX;
    const int r_src = C::POFF + C::SHIFT + r_row * PW + 4 * r_px + 1;
    const int r_dst = C::XOFF + (r_row * NPX + r_px) * 4;
    const int c_pos = min(lane, NP - 1);
    const int c_py = c_pos / NPX, c_px = c_pos - c_py * NPX;
    const int c_src = C::XOFF + ((w4 >> 1) * PH * NPX + (4 * c_py) * NPX + c_px) * 4 + (w4 & 1) * 2;
    const int c_dst = C::VOFF + ((w4 >> 1) * C::NPT + c_py * C::NPV + c_px) * 4 + (w4 & 1) * 2;

    const f32x4 *lds4 = (const f32x4 *)lds;
    const int gx = l15 % C::GTX, gy = l15 / C::GTX;
    const int Tx = (tg % C::WTX) * C::GTX + gx, Ty = (tg / C::WTX) * C::GTY + gy;
    const int aBase = C::UOFF / 4 + q * 32 + cb * 16 + l15 + fh * 128;                            // this half's quads: 2 rf + fh
    const int bBase = C::VOFF / 4 + (Ty + (q >> 1)) * C::NPV + Tx + (q & 1) + fh * C::NPT;

    const int n = p.Cin;
    f32x4 ra[2];
    float rc[7], re[6];
    f32x2 cx[7];
    float cv0[7], cv1[7], ce0[6], ce1[6];
    f32x4 a[3], bq[3];

    // the wave's epilogue variant is known up front; with an addend and no edge tile in the wave, the 2 x 4 addend rows of the wave's couts
    // are requested behind the barrier of the LAST channel (no LDS-DMA is issued after it: nothing else waits on the vector-memory counter)
    const int opx = x0 + 4 * Tx, opy = y0 + 4 * Ty;
    const bool edge_any = __builtin_amdgcn_ballot_w64(!(opy + 4 <= p.H && opx + 4 <= p.W && p.vec)) != 0;
    const bool zpre_on = p.add != nullptr && !edge_any;
    f32x4 zpre[8];
    auto addend_prefetch = [&]() __attribute__((always_inline)) {
        if (zpre_on) {
            const float *addb = p.add + (long long)(b / p.adiv) * p.asb + (long long)(4 * q + nb * 32 + cb * 16 + 2 * fh) * p.asc + (long long)opy * p.ash + opx;
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) zpre[k * 4 + i] = *(const f32x4 *)(addb + (long long)k * p.asc + (long long)i * p.ash);
        }
    };

    // transform pieces (as in wino7_kernel): 0 .. 4 the row pass, 5 .. 14 the column pass of pair J
    auto tpiece = [&](auto J, int k, bool doR, bool doC, int rbuf, int xrbuf, int xcbuf, int vbuf) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        if (k == 0 && doR) {
            const float *src = lds + r_src + rbuf * C::PCAP;
            ra[0] = *(const f32x4 *)src;
            ra[1] = *(const f32x4 *)(src + 4);
        }
        if (k >= 1 && k <= 3 && doR) {
            const float d[7] = {ra[0][0], ra[0][1], ra[0][2], ra[0][3], ra[1][0], ra[1][1], ra[1][2]};
            if (k == 1) w7_bt_e(d, re);
            if (k == 2) w7_bt_a(re, rc);
            if (k == 3) w7_bt_b(d, re, rc);
        }
        if (k == 4 && doR) {
            float *dst = lds + r_dst + xrbuf * C::XSZ;
            *(f32x4 *)dst = f32x4{rc[0], rc[1], rc[2], rc[3]};
            *(f32x4 *)(dst + PH * NPX * 4) = f32x4{rc[4], rc[5], rc[6], 0.f};
        }
        if ((k == 5 || k == 6) && doC) {
            const float *src = lds + c_src + xcbuf * C::XSZ;
#pragma unroll
            for (int i = (k == 5 ? 0 : 4); i < (k == 5 ? 4 : 7); ++i) cx[i] = *(const f32x2 *)(src + i * NPX * 4);
        }
        if (k >= 7 && k <= 12 && doC) {
            const int which = (k - 7) / 3, part = (k - 7) % 3;
            if (which == 0 || j < 3) {
                const float d[7] = {cx[0][which], cx[1][which], cx[2][which], cx[3][which], cx[4][which], cx[5][which], cx[6][which]};
                if (which == 0) {
                    if (part == 0) w7_bt_e(d, ce0);
                    if (part == 1) w7_bt_a(ce0, cv0);
                    if (part == 2) w7_bt_b(d, ce0, cv0);
                } else {
                    if (part == 0) w7_bt_e(d, ce1);
                    if (part == 1) w7_bt_a(ce1, cv1);
                    if (part == 2) w7_bt_b(d, ce1, cv1);
                }
            }
        }
        if ((k == 13 || k == 14) && doC) {
            float *dst = lds + c_dst + vbuf * C::VSZ;
#pragma unroll
            for (int rf = (k == 13 ? 0 : 4); rf < (k == 13 ? 4 : 7); ++rf) {
                if (j < 3) *(f32x2 *)(dst + rf * 2 * C::NPT * 4) = f32x2{cv0[rf], cv1[rf]};
                else dst[rf * 2 * C::NPT * 4] = cv0[rf];
            }
        }
    };
    // half 0, slot m of 28: DMA instructions (NIU filter pieces, then the patch piece) in the odd slots, the row pass in the even ones;
    // half 1, slot m of 21: the column pass
    auto dma_of_slot = [](int m) constexpr -> int { return (m % 2 == 1 && m / 2 < C::NIU + 1) ? m / 2 : -1; };
    auto rpiece_of_slot = [](int m) constexpr -> int { return m == 0 ? 0 : m == 4 ? 1 : m == 6 ? 2 : m == 8 ? 3 : m == 10 ? 4 : -1; };
    auto cpiece_of_slot = [](int m) constexpr -> int {
        switch (m) {
            case 0: return 5;
            case 1: return 6;
            case 4: return 7;
            case 5: return 8;
            case 7: return 9;
            case 8: return 10;
            case 10: return 11;
            case 11: return 12;
            case 14: return 13;
            case 16: return 14;
            default: return -1;
        }
    };

    // 7 quads of this half per input channel (4 / 3 MFMAs each), operands of quad rf + 2 fetched behind the first MFMA of quad rf
    auto matrix = [&](auto FHc, int st, bool steady, auto slot) __attribute__((always_inline)) {
        constexpr int FH = decltype(FHc)::value, NE = FH ? 3 : 4;
        const int ai = aBase + st * (C::USZ / 4), bi = bBase + st * (C::VSZ / 4);
        a[0] = lds4[ai];
        bq[0] = lds4[bi];
        a[1] = lds4[ai + 256];
        bq[1] = lds4[bi + 2 * C::NPT];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rf = 0; rf < 7; ++rf) {
            const int cur = rf % 3, nxt = (rf + 2) % 3;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                acc[rf * NE + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][e], bq[cur][e], acc[rf * NE + e], 0, 0, 0);
                if (e == 0 && rf + 2 < 7 && !(steady && (W7S_ABL & 8))) {
                    __builtin_amdgcn_sched_barrier(0);
                    a[nxt] = lds4[ai + (rf + 2) * 256];
                    bq[nxt] = lds4[bi + (rf + 2) * 2 * C::NPT];
                }
                __builtin_amdgcn_sched_barrier(0);
                slot(rf * NE + e);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

#ifdef W7_TRACE
    unsigned long long tph[4] = {0, 0, 0, 0};
    unsigned long long tk = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tk;
#endif
    // one iteration of the pipeline (c from -2): DMA U(c+1), patch(c+3) | row pass (c+2) | column pass (c+1) | matrix (c)
    auto iter = [&](auto FHc, auto J, int c, auto STEADY) __attribute__((always_inline)) {
        constexpr int FH = decltype(FHc)::value, NS = FH ? 21 : 28;
        constexpr bool steady = decltype(STEADY)::value;
        const int par = c & 1;
        const bool doM = steady || c >= 0;
        const bool doC = steady || (c + 1 >= 0 && c + 1 < n);
        const bool doR = steady || c + 2 < n;
        const bool doP = steady || (c + 3 < n && c > -2);          // (the patch of channel 1 and the filter of channel 0 are requested up
        const bool doU = steady || (c + 1 < n && c >= 0);          //  front, with channel 0's patch)
        W7STAMP(0)
        if (FH == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W7STAMP(1)
        __syncthreads();
        W7STAMP(2)
        if (!steady && c == n - 1) addend_prefetch();
        const int rbuf = par, xrbuf = par, xcbuf = par ^ 1, vbuf = par ^ 1;
        auto slot = [&](int m) __attribute__((always_inline)) {
            if (steady && (W7S_ABL & 2) && !(FH == 0 && dma_of_slot(m) >= 0)) return;
            if (FH == 0) {
                const int d = dma_of_slot(m);
                if (d >= 0 && !(steady && (W7S_ABL & 1))) {
                    if (d < C::NIU) {
                        if (doU) dma_u(c + 1, d, par ^ 1);
                    } else if (doP) {
                        dma_p(c + 3, par ^ 1);
                    }
                }
                const int k = rpiece_of_slot(m);
                if (k >= 0) tpiece(J, k, doR, false, rbuf, xrbuf, xcbuf, vbuf);
            } else {
                const int k = cpiece_of_slot(m);
                if (k >= 0) tpiece(J, k, false, doC, rbuf, xrbuf, xcbuf, vbuf);
            }
        };
        if (doM) {
            matrix(FHc, par, steady, slot);
        } else {
#pragma unroll
            for (int m = 0; m < NS; ++m) {
                slot(m);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    auto run = [&](auto FHc, auto J) __attribute__((always_inline)) {
        using T = std::true_type;
        using F = std::false_type;
        iter(FHc, J, -2, F{});
        iter(FHc, J, -1, F{});
        int c = 0;
        for (; c + 3 < n; ++c) iter(FHc, J, c, T{});
        for (; c < n; ++c) iter(FHc, J, c, F{});
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    if (fh == 0) {
        dma_p(0, 0);
        if (n > 1) dma_p(1, 1);          // both patch buffers and the first filter at once: one memory latency in front of the first row
#pragma unroll
        for (int k = 0; k < C::NIU; ++k) dma_u(0, k, 0);          // pass instead of one per prologue iteration
        run(H0{}, std::integral_constant<int, 0>{});
    } else if (w4 == 0) run(H1{}, std::integral_constant<int, 0>{});
    else if (w4 == 1) run(H1{}, std::integral_constant<int, 1>{});
    else if (w4 == 2) run(H1{}, std::integral_constant<int, 2>{});
    else run(H1{}, std::integral_constant<int, 3>{});

    W7STAMP(0)
    // ---- epilogue: partial tiles of the four couts, exchange, this wave's two couts (elements 2 fh, 2 fh + 1) finished
    __syncthreads();          // every wave is out of the matrix loop: LDS is free
    f32x4 *ex = (f32x4 *)lds;          // [wave][k][row][lane]
    float yo[2][4][4];
    auto halves = [&](auto FHc) __attribute__((always_inline)) {
        constexpr int FH = decltype(FHc)::value;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float y[4][4];
            w7s_partial<FH>(acc, 2 * (1 - FH) + k, y);          // the partner's couts
#pragma unroll
            for (int i = 0; i < 4; ++i) ex[((wid * 2 + k) * 4 + i) * 64 + lane] = f32x4{y[i][0], y[i][1], y[i][2], y[i][3]};
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) w7s_partial<FH>(acc, 2 * FH + k, yo[k]);
    };
    if (fh == 0) halves(H0{});
    else halves(H1{});
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 z = ex[(((wid ^ 4) * 2 + k) * 4 + i) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) yo[k][i][e] += z[e];
        }
    {
        const int px = opx, py = opy;
        const int r0 = 2 * fh;
        auto tile = [&](int r, float (&y)[4][4]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) y[i][e] = yo[r & 1][i][e];
        };
        auto fin = [&](auto R0) __attribute__((always_inline)) {
            constexpr int r0c = decltype(R0)::value;
            if (edge_any) w7_epilogue<false, false, 2>(p, tile, bv, b, nb * 32 + cb * 16, r0c, q, px, py);
            else if (p.add) w7_epilogue<true, true, 2>(p, tile, bv, b, nb * 32 + cb * 16, r0c, q, px, py, zpre);
            else w7_epilogue<true, false, 2>(p, tile, bv, b, nb * 32 + cb * 16, r0c, q, px, py);
        };
        if (r0 == 0) fin(std::integral_constant<int, 0>{});
        else fin(std::integral_constant<int, 2>{});
    }
#ifdef W7_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W7STAMP(3)
    if (p.dbg && lane == 0 && (blockIdx.x % 64) == 0) {
        for (int i = 0; i < 4; ++i) atomicAdd(p.dbg + wid * 8 + i, tph[i]);
        atomicAdd(p.dbg + wid * 8 + 4, tk - tstart);
        atomicAdd(p.dbg + wid * 8 + 5, 1ULL);
    }
#endif
}

// ---- tile configurations ---------------------------------------------------------------------------------------------------------
//                     GTX WTY WTX          tiles of 4x4 px     TH   TW
using Z7A = W7Cfg<8, 2, 1>;      //          8 x 4                16   32
using Z7B = W7Cfg<4, 1, 2>;      //          8 x 4 (4x4 groups)   16   32

// (name, configuration, 1 = the frequency-split kernel of eight waves)
#define SSM_W7_KINDS(X) X(Z7A_, Z7A, 0) X(Z7B_, Z7B, 0) X(Z7AS_, Z7A, 1) X(Z7BS_, Z7B, 1)

enum W7Kind {
#define X(name, cfg, split) name,
    SSM_W7_KINDS(X)
#undef X
        NW7KIND
};

struct W7KindInfo {
    int th, tw, split;
};

constexpr W7KindInfo kW7Info[NW7KIND] = {
#define X(name, cfg, split) W7KindInfo{cfg::TH, cfg::TW, split},
    SSM_W7_KINDS(X)
#undef X
};

std::atomic<int> g_force_w7kind{-1};
#ifdef W7_TRACE
std::atomic<unsigned long long *> g_w7dbg{nullptr};
#endif

// the configuration with the fewest workgroup-rounds (tile overshoot included); ties go to the 16x32-pixel tile
int pick_w7kind(int Cout, int B, int H, int W) {
    const int forced = g_force_w7kind.load();
    if (forced >= 0 && forced < NW7KIND) return forced;
    // $SSM_WINO7_SPLIT=0: the 4-wave kernel (read per call: A/B runs and the parity tests use both in one process)
    const char *env = getenv("SSM_WINO7_SPLIT");
    const int split = !(env && atoi(env) == 0);
    int best = -1;
    long long bt = -1;
    for (int i = 0; i < NW7KIND; ++i) {
        if (kW7Info[i].split != split) continue;
        const long long nwg = (long long)B * ((W + kW7Info[i].tw - 1) / kW7Info[i].tw) * ((H + kW7Info[i].th - 1) / kW7Info[i].th) * (Cout / 32);
        const long long rounds = (nwg + 255) / 256;
        if (bt < 0 || rounds < bt) {
            best = i;
            bt = rounds;
        }
    }
    return best;
}

template <class C, int SPLIT>
int w7launch(W7Params &p, int B, hipStream_t st) {
    p.tilesX = (p.W + C::TW - 1) / C::TW;
    p.tilesY = (p.H + C::TH - 1) / C::TH;
    p.NB = p.Cout / 32;
    // (no read outside the padded plane: the per-lane DMA offsets clamp overshoot rows / pieces to the zero frame, see wino7_kernel)
    const long long blocks = (long long)p.tilesX * p.tilesY * p.NB * B;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        ssm::set_error("wino7 conv: grid of %lld workgroups out of range", blocks);
        return SSM_E_ARG;
    }
    void (*kern)(const W7Params) = SPLIT ? wino7s_kernel<C> : wino7_kernel<C>;
    constexpr int lds_bytes = C::BYTES;
    static std::atomic<uint64_t> lds_reserved{0};          // one bit per device: the attribute is per (kernel, device)
    const hipError_t attr_rc = ssm::reserve_lds(lds_reserved, (const void *)kern, lds_bytes);
    if (attr_rc != hipSuccess) {
        ssm::set_error("wino7 conv: cannot reserve %d bytes of LDS: %s", lds_bytes, hipGetErrorString(attr_rc));
        return SSM_E_LAUNCH;
    }
    SSM_LAUNCH(kern, dim3((unsigned)blocks), dim3(SPLIT ? 512 : 256), lds_bytes, st, p);
    return ssm::check_launch("ssm_wino7_conv2d_add_fwd");
}

int w7dispatch(int kind, W7Params &p, int B, hipStream_t st) {
    switch (kind) {
#define X(name, cfg, split) \
    case name: return w7launch<cfg, split>(p, B, st);
        SSM_W7_KINDS(X)
#undef X
    }
    return SSM_E_UNSUPPORTED;
}

__global__ void wino7_pack_kernel(const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ wp, float *__restrict__ bp,
                                  int Cout, int Cin, long long total, int nbias) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i < total) {
        float out[4];
        auto at = [&](int co, int ci, int ky, int kx) { return w[(((long long)co * Cin + ci) * 7 + ky) * 7 + kx]; };
        ssm_w7_pack_quad(at, Cout, Cin, i, out);
        *reinterpret_cast<f32x4 *>(wp + i) = f32x4{out[0], out[1], out[2], out[3]};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (i + e < nbias) bp[i + e] = (i + e < Cout) ? bias[i + e] : 0.f;
}

}  // namespace

extern "C" int ssm_wino7_plan(int Cin, int Cout, int B, int H, int W, int *kind) {
    if (Cin <= 0 || Cout <= 0 || Cout % 32) {
        ssm::set_error("wino7 conv: no tile configuration for Cin=%d Cout=%d (Cout a multiple of 32)", Cin, Cout);
        return SSM_E_UNSUPPORTED;
    }
    if (kind) *kind = pick_w7kind(Cout, B, H, W);
    return SSM_OK;
}

#ifdef W7_TRACE
// tuning build only (make w7alt W7FLAGS=-DW7_TRACE=1; never lib/libssm_hip.so): 4 x 8 device counters, per wave of the sampled workgroups
// the shader cycles spent in [work, DMA wait, barrier wait, epilogue], the lifetime and the number of samples
extern "C" int ssm_wino7_debug_buffer(unsigned long long *dev_counters) {
    g_w7dbg.store(dev_counters);
    return SSM_OK;
}
#endif

extern "C" int ssm_wino7_force_kind(int kind) {
    g_force_w7kind.store(kind >= 0 && kind < NW7KIND ? kind : -1);
    return NW7KIND;
}

// (Cout rounded up to whole 32-channel blocks: the pack fills the channels beyond Cout with zeros - a caller that convolves with a
// narrower filter, e.g. the data gradient of a layer with fewer than 32 inputs, launches the convolution with the padded count)
extern "C" size_t ssm_wino7_packed_weight_floats(int Cout, int Cin) { return (size_t)((Cout + 31) / 32) * (size_t)Cin * SSM_W7_NFQ * 4 * 32 * 4; }

extern "C" int ssm_wino7_pack_weights(const float *w, const float *bias, float *wp, float *bp, int Cout, int Cin, void *stream) {
    SSM_REQUIRE(w && bias && wp && bp, "wino7 pack_weights: null pointer");
    SSM_REQUIRE(Cout > 0 && Cin > 0, "wino7 pack_weights: bad sizes");
    SSM_REQUIRE(ssm::aligned16(wp), "wino7 pack_weights: the packed filter must be 16-byte aligned");
    const long long total = (long long)ssm_wino7_packed_weight_floats(Cout, Cin);
    const int nbias = (Cout + 31) / 32 * 32;          // (bias_packed holds whole blocks too)
    const long long n = (total > nbias ? total : nbias) / 4 + 1;
    SSM_LAUNCH(wino7_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, bias, wp, bp, Cout, Cin, total,
                       nbias);
    return ssm::check_launch("ssm_wino7_pack_weights");
}

extern "C" int ssm_wino7_conv2d_add_fwd(ssm_view x, int Cin, const float *w_packed, const float *bias_packed, ssm_view y, ssm_view pool,
                                        ssm_view add, int add_div, int B, int H, int W, int Cout, float slope, int flags, void *stream) {
    int kind = 0;
    SSM_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && Cin > 0, "wino7 conv: bad sizes");
    const int rc = ssm_wino7_plan(Cin, Cout, B, H, W, &kind);
    if (rc != SSM_OK) return rc;
    SSM_REQUIRE(x.ptr && y.ptr && w_packed && bias_packed, "wino7 conv: null pointer");
    SSM_REQUIRE(ssm::aligned16(x.ptr) && x.sh % 4 == 0 && x.sc % 4 == 0 && x.sb % 4 == 0,
                "wino7 conv: the input is not a padded-plane view (16-byte alignment)");
    SSM_REQUIRE(x.sh >= W + 2 * SSM_PADX, "wino7 conv: input row stride %d leaves no zero frame for W=%d", x.sh, W);
    SSM_REQUIRE(ssm::aligned16(w_packed), "wino7 conv: packed filter must be 16-byte aligned");
    SSM_REQUIRE(64LL * x.sh * 4 < 0x7fffffffLL, "wino7 conv: row stride too large");
    W7Params p;
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
    p.add = nullptr;
    p.asb = p.asc = 0;
    p.ash = 0;
    p.adiv = 1;
    bool vec = W % 4 == 0 && ssm::aligned16(y.ptr) && y.sh % 4 == 0 && y.sc % 4 == 0 && y.sb % 4 == 0;
    if (add.ptr) {
        SSM_REQUIRE(add_div >= 1 && B % add_div == 0, "wino7 conv: the addend serves %d batch entries each, batch %d is no multiple", add_div, B);
        p.add = add.ptr;
        p.asb = add.sb;
        p.asc = add.sc;
        p.ash = add.sh;
        p.adiv = add_div;
        vec = vec && ssm::aligned16(add.ptr) && add.sh % 4 == 0 && add.sc % 4 == 0 && add.sb % 4 == 0;
    }
    if (pool.ptr) {
        SSM_REQUIRE(H % 2 == 0 && W % 2 == 0, "wino7 conv: fused pool needs even H, W");
        p.pool = pool.ptr;
        p.psb = pool.sb;
        p.psc = pool.sc;
        p.psh = pool.sh;
        vec = vec && (reinterpret_cast<size_t>(pool.ptr) & 7) == 0 && pool.sh % 2 == 0 && pool.sc % 2 == 0 && pool.sb % 2 == 0;
    }
    p.vec = vec ? 1 : 0;
#ifdef W7_TRACE
    p.dbg = g_w7dbg.load();
#else
    p.dbg = nullptr;
#endif
    return w7dispatch(kind, p, B, (hipStream_t)stream);
}
