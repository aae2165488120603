// 5x5 convolution as two-dimensional Winograd F(4x4,5x5) on the CDNA4 fp32 matrix cores (v_mfma_f32_16x16x4_f32), all arithmetic fp32.
//
// Same operator as the k = 5 case of ssm_conv.hip / ssm_wino1d.hip (layers.conv of the reference, scripts/models/layers.py:21-33: stride-1
// 'same' cross-correlation, zero padding, bias, LeakyReLU; fused 2x2 mean, scripts/models/layers.py:60-63) - conv2a / conv2b of both U-Nets
// (scripts/models/flow_computation.py:43-45), 32 -> 64 and 64 -> 64 channels at half resolution.
//
//      Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A      per 4x4 output tile (d = its 8x8 input window, g = the 5x5 filter)
//
// over the eight points {0, +-1, +-2, +-1/2, inf} of the one-dimensional F(4,5) form of ssm_wino1d.hip, now on both axes: 64 multiplies
// per 16 outputs and (cin, cout) = 4 per output instead of 10 (F(4,5) along x) or 25 (direct).  In fp32 a 64-channel layer sits 3.2e-6
// rms / 2.8e-5 max from float64 at unit output scale (tests/emulate_winograd_5x5_2d.py; the 1-D form: 1.4e-6 / 1.1e-5) - the same
// level as the blocked 7x7 form of ssm_wino7.hip, whose structure this kernel shares:
//
// GEMM per frequency f: M_f[cout][tile] = sum_cin U_f[cout][cin] V_f[cin][tile], k-step = 4 input channels; a wave owns 16 couts x 16
// tiles for all 64 frequencies (64 accumulators of 4 registers: one wave per SIMD), the output transform is lane-local.  One workgroup
// of four waves = 32 couts x 32 tiles of 4x4 pixels.  Input transform with the overlap of neighbouring windows used (8 rows at stride
// 4): a ROW pass per (channel, patch row, tile column) leaves X in LDS, a COLUMN pass per (channel, half of the column-frequencies,
// tile) writes V [16 quads][4 channels][32 tiles][4]; wave w does both passes of channel w, every LDS access of the passes is 16 bytes
// wide.  26 vector operations per 8-point pass.
//
// Per k-step three phases, two barriers (the 160 KiB of LDS hold two filter stages of 32 KiB, two patches, ONE X and ONE V):
//   [barrier] row pass, column pass [barrier] 64 MFMAs, the DMA of the next k-step's filter and patch one instruction per quad
// The phases are serial on purpose: 256 accumulator registers leave no second wave per SIMD and no second V / X in LDS to run them beside
// the MFMAs.  A lone wave issues one vector instruction per ~5 cycles (tools/valu_rate_probe.py), so the two passes (182 vector + 44 LDS
// instructions) cost ~2000 cycles per k-step beside 2048 of MFMA: the form runs at ~0.33 of the matrix pipe and still beats F(4,5)
// along x (0.71 of the pipe at 2.5 x the multiplies) by 1.27 x.
#include "ssm_common.h"
#include "ssm_wino5_pack.h"

#include <atomic>
#include <mutex>
#include <type_traits>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

struct W5Params {
    const float *src;
    long long sb, sc;    // batch / channel stride
    int sh;              // row stride
    int Cin;
    const float *wpk;    // U, [Cout/32][Cin/4][16 quads][4 cin][32][4]
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
    unsigned long long *dbg;   // tuning build (-DW5_TRACE) only: per-wave phase sums of sampled workgroups
};

// 2 cout halves x 2 tile groups = 4 waves; a tile group = GTX x GTY tiles of 4x4 pixels (16 tiles), the groups sit WTY x WTX.
template <int GTX_, int WTY_, int WTX_>
struct W5Cfg {
    static constexpr int GTX = GTX_, GTY = 16 / GTX_, WTY = WTY_, WTX = WTX_;
    static constexpr int NTX = GTX * WTX, NTY = GTY * WTY, NT = 32;            // tiles per workgroup
    static constexpr int TH = 4 * NTY, TW = 4 * NTX;                           // output pixels per workgroup
    static constexpr int CK = 4;                                               // input channels per k-step
    static constexpr int PH = TH + 4, PW = TW + 8;                             // patch rows y0-2 .. y0+TH+1, columns x0-4 .. x0+TW+3
    static constexpr int SHIFT = 2;                                            // floats: the window of tile column Tx starts at patch column 4 Tx + 2 + SHIFT (16-byte aligned)
    static constexpr int NFQ = 16;                                             // quads of frequencies (8 row-frequencies x 2)
    static constexpr int USZ = NFQ * CK * 32 * 4;                              // filter floats per k-step
    // LDS strides padded against bank conflicts: the column pass reads rows 4 Ty + i of X for all tiles at once (unpadded: 4 rows = 512
    // bytes apart = the same banks, 8-way conflicts) and writes V for two channels at once (unpadded: 512 bytes apart); measured
    // 1500 -> cycles per k-step in the column pass (tools/wino5_phase_probe.py)
    static constexpr int NTP = NT + 2;                                         // V: f32x4 units between the channels of a quad
    static constexpr int XRW = NTX * 4 + 4;                                    // X: floats per row
    static constexpr int VSZ = NFQ * CK * NTP * 4;                             // transformed windows of a k-step
    static constexpr int XPL = PH * XRW;                                       // one plane (channel, half) of row-pass results
    static constexpr int XSZ = CK * 2 * XPL;
    static constexpr int NDQ = CK * PH * PW / 4, NGP = (NDQ + 63) / 64;        // 16-byte pieces / 1-KiB DMA groups of the patch
    static constexpr int PCAP = NGP * 256 + 256;
    static constexpr int NGU = USZ / 256, NIU = NGU / 4, NIP = (NGP + 3) / 4;  // DMA instructions per wave and k-step
    static constexpr int NRU = PH * NTX, NRR = (NRU + 63) / 64;                // row-pass units of a channel, rounds of one wave
    static constexpr int UOFF = 0, VOFF = 2 * USZ, XOFF = VOFF + VSZ, POFF = XOFF + XSZ;
    static constexpr int BYTES = (POFF + 2 * PCAP) * 4;
    static_assert(WTY * WTX == 2 && (GTX == 4 || GTX == 8 || GTX == 16), "two tile groups of 16 tiles");
    static_assert(PW % 4 == 0 && NGU % 4 == 0, "whole DMA groups");
    static_assert(BYTES <= 160 * 1024, "LDS budget");
};

// 8-point transform B^T over the points 0, +1, -1, +2, -2, +1/2, -1/2, inf (the matrix of ssm_wino1d.hip: it depends on the points only)
__device__ __forceinline__ void w5_bt(const float (&e)[8], float (&f)[8]) {
    f[0] = (e[0] - e[6]) + 5.25f * (e[4] - e[2]);
    f[7] = (e[7] - e[1]) + 5.25f * (e[3] - e[5]);
    const float t1 = (e[2] + e[6]) - 4.25f * e[4], t2 = (e[1] + e[5]) - 4.25f * e[3];
    const float t3 = (e[6] + 0.25f * e[2]) - 1.25f * e[4], t4 = (0.5f * e[1] - 2.5f * e[3]) + 2.f * e[5];
    const float t5 = (e[6] + 4.f * e[2]) - 5.f * e[4], t6 = (2.f * e[1] - 2.5f * e[3]) + 0.5f * e[5];
    f[1] = t1 + t2;
    f[2] = t1 - t2;
    f[3] = t3 + t4;
    f[4] = t3 - t4;
    f[5] = t5 + t6;
    f[6] = t5 - t6;
}

// A^T = [1 1 1 1 1 1 1 0; 0 1 -1 2 -2 1/2 -1/2 0; 0 1 1 4 4 1/4 1/4 0; 0 1 -1 8 -8 1/8 -1/8 1]
__device__ __forceinline__ void w5_at(float m0, float m1, float m2, float m3, float m4, float m5, float m6, float m7, float (&y)[4]) {
    const float s1 = m1 + m2, t1 = m1 - m2, s2 = m3 + m4, t2 = m3 - m4, s3 = m5 + m6, t3 = m5 - m6;
    y[0] = (m0 + s1) + (s2 + s3);
    y[1] = (t1 + 2.f * t2) + 0.5f * t3;
    y[2] = (s1 + 4.f * s2) + 0.25f * s3;
    y[3] = ((t1 + 8.f * t2) + 0.125f * t3) + m7;
}

// ---- epilogue: Y = A^T M A per accumulator register (4 couts per lane), + bias, addend, LeakyReLU, stores, fused 2x2 mean.
// cu0: first cout of the wave's 16-cout block (this lane holds couts cu0 + 4 q + r), (px, py): the lane's 4x4 output tile.
// FAST (chosen per wave): every tile of the wave lies inside the map and rows move as aligned 16-byte pieces - the element-wise edge
// path, a branch per element even when no lane takes it, is not compiled in (ssm_wino7.hip: - 2.7 % on a full-resolution layer)
template <bool FAST>
__device__ __forceinline__ void w5_epilogue(const W5Params &p, const f32x4 (&acc)[64], const float (&bv)[4], int b, int cu0, int q, int px, int py) {
    const float sl = (p.lrelu & 1) ? p.slope : 1.f;
    const bool amask = (p.lrelu & 2) != 0;          // SSM_FLAG_MASK: the addend view is a mask source (see ssm_hip.h; r6: the 5x5 data gradients too)
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
    f32x4 zadd[2][4];
    const bool zvec = p.add && vok;
    auto zload = [&](int r) {
        if (zvec) {
#pragma unroll
            for (int i = 0; i < 4; ++i) zadd[r & 1][i] = *(const f32x4 *)(addb + (long long)(cu0 + r) * p.asc + (long long)i * p.ash);
        }
    };
    zload(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cu = cu0 + r;          // uniform; this lane's cout = cu + 4 * q
        if (r + 1 < 4) zload(r + 1);
        float t[4][8];                   // A^T M: over the row-frequencies, for every column-frequency
#pragma unroll
        for (int cf = 0; cf < 8; ++cf) {
            float y4[4];
            w5_at(acc[cf][r], acc[8 + cf][r], acc[16 + cf][r], acc[24 + cf][r], acc[32 + cf][r], acc[40 + cf][r], acc[48 + cf][r], acc[56 + cf][r], y4);
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i][cf] = y4[i];
        }
        float y[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float y4[4];
            w5_at(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], t[i][6], t[i][7], y4);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[i][e] = y4[e] + bv[r];
        }
        if (p.add) {          // (uniform; addb is a per-lane pointer)
            const float *ap = addb + (long long)cu * p.asc;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (vok) {
                    const f32x4 z = zadd[r & 1][i];
                    if (amask) {
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
                                y[i][e] = amask ? y[i][e] * (z > 0.f ? 1.f : p.slope) : y[i][e] + z;
                            }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) y[i][e] = fmaxf(y[i][e], y[i][e] * sl);
        // the wait for the prefetched addend rows of cout r + 1 goes in front of the inline-assembly stores, which the compiler's wait-count
        // pass does not see (ssm_wino7.hip, w7_epilogue)
        if (zvec && r + 1 < 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(zadd[(r + 1) & 1][i]));
        }
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
                const bool rok = FAST || py + 2 * i < p.H;
                if (FAST || (rok && px + 4 <= p.W && p.vec)) st2(qp + (long long)i * p.psh, qb, f32x2{o0, o1});
                else if (rok) {
                    if (px + 2 <= p.W) st1(qp + (long long)i * p.psh, qb, o0);
                    if (px + 4 <= p.W) st1(qp + (long long)i * p.psh + 1, qb, o1);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

#if !W5_SPLIT
template <class C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino5_kernel(const W5Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PH = C::PH, PW = C::PW, NTX = C::NTX, CK = C::CK, NFQ = C::NFQ;

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

    const float *pbase = p.src + (long long)b * p.sb + (long long)(y0 - 2) * p.sh + (x0 - 4);
    const float *wbase = p.wpk + (long long)nb * (p.Cin / CK) * C::USZ;

    // per-lane source offsets (bytes) of the patch pieces this wave brings per k-step (piece = 16 bytes of a patch row of one channel)
    int poff[C::NIP];
#pragma unroll
    for (int i = 0; i < C::NIP; ++i) {
        const int qq = (i * 4 + wid) * 64 + lane;
        if (qq < C::NDQ) {
            const int c = qq / (PH * (PW / 4)), rem = qq - c * (PH * (PW / 4));
            // rows below the bottom zero frame / pieces right of the padded row: read from the frame's last row / piece (zeros) - a tile
            // overshoot never brings in another plane's pixels or the memory behind the last plane (see ssm_wino7.hip)
            const int r = rem / (PW / 4), j = rem - r * (PW / 4);
            const int re = min(r, p.H + (SSM_PADY - 1) - (y0 - 2)), fe = min(4 * j, ((p.W + 2 * SSM_PADX + 3) & ~3) - 4 - x0);
            poff[i] = ((int)(c * p.sc) + re * p.sh + fe) * 4;
        } else {
            poff[i] = 0;          // tail of the last 1-KiB piece: lands in the buffer's padding
        }
    }
    const int uoff = lane * 16;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
    auto dma = [](const float *base, int voff_bytes, unsigned m0v) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(base), "s"(m0v) : "memory", "m0");
    };
    // the n-th DMA instruction of this wave for k-step s (into filter stage / patch buffer s & 1): n < NIP a patch piece, else a filter piece
    auto issue_n = [&](int s, int n) {
        const int buf = s & 1;
        if (n < C::NIP) {
            const int g = 4 * n + wid;
            if (4 * n + 3 < C::NGP || g < C::NGP) {
                const float *base = pbase + (long long)(s * CK) * p.sc;
                const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(C::POFF + buf * C::PCAP + C::SHIFT) * 4u + (unsigned)g * 1024u);
                dma(base, poff[n], m0v);
            }
        } else {
            const int g = 4 * (n - C::NIP) + wid;
            const float *base = wbase + (long long)s * C::USZ + g * 256;
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(C::UOFF + buf * C::USZ) * 4u + (unsigned)g * 1024u);
            dma(base, uoff, m0v);
        }
    };
    static_assert(C::NIP + C::NIU <= 16, "one DMA slot per quad of the matrix phase");

    f32x4 acc[64];
#pragma unroll
    for (int f = 0; f < 64; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = p.bias[nb * 32 + cb * 16 + 4 * q + r];

    // ---- row pass: wave w transforms the rows of channel w of the k-step - the rows its own column pass reads, so no barrier stands
    // between the two passes; unit = (patch row, tile column), NRR rounds of 64 lanes (lanes beyond the last unit repeat it) -----------
    int r_src[C::NRR], r_dst[C::NRR];
#pragma unroll
    for (int k = 0; k < C::NRR; ++k) {
        const int u = min(lane + 64 * k, C::NRU - 1);
        const int c = wid, row = u / NTX, t = u - row * NTX;
        r_src[k] = C::POFF + C::SHIFT + (c * PH + row) * PW + 4 * t + 2;          // floats; 16-byte aligned
        r_dst[k] = C::XOFF + c * 2 * C::XPL + row * C::XRW + t * 4;               // half 0; half 1 at + XPL
    }
    // ---- column pass: unit = (channel = wave, half h = lane >> 5 of the column-frequencies, tile): one unit per thread.  16-byte reads
    // and writes: the 8-byte form (a pair of column-frequencies per thread) used half of the banks per instruction and cost ~500 of the
    // pass's 1580 cycles (profiles/r9i_wino5_first_phases_and_layers.txt) ------------------------------------------------------------
    const int c_tile = lane & 31;
    const int c_gx = (c_tile & 15) % C::GTX, c_gy = (c_tile & 15) / C::GTX, c_g2 = c_tile >> 4;
    const int c_Tx = (c_g2 % C::WTX) * C::GTX + c_gx, c_Ty = (c_g2 / C::WTX) * C::GTY + c_gy;
    const int c_src = C::XOFF + (wid * 2 + (lane >> 5)) * C::XPL + (4 * c_Ty) * C::XRW + c_Tx * 4;
    const int c_dst = C::VOFF + (((lane >> 5) * CK + wid) * C::NTP + c_tile) * 4;                      // + rf * 2 * CK * NTP * 4

    // ---- operand bases of the matrix phase (f32x4 units): U of (quad, channel q, cout cb*16 + l15), V of (quad, channel q, tile) ------
    const f32x4 *lds4 = (const f32x4 *)lds;
    const int aBase = C::UOFF / 4 + q * 32 + cb * 16 + l15;
    const int bBase = C::VOFF / 4 + q * C::NTP + tg * 16 + l15;
    const int gx = l15 % C::GTX, gy = l15 / C::GTX;
    const int Tx = (tg % C::WTX) * C::GTX + gx, Ty = (tg / C::WTX) * C::GTY + gy;

    const int nsteps = p.Cin / CK;
#pragma unroll
    for (int n = 0; n < C::NIP + C::NIU; ++n) issue_n(0, n);
#ifdef W5_TRACE
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tk = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tk;
#define W5STAMP(i)                                                  \
    {                                                               \
        const unsigned long long tn = __builtin_amdgcn_s_memtime(); \
        tph[i] += tn - tk;                                          \
        tk = tn;                                                    \
    }
#else
#define W5STAMP(i)
#endif
    for (int s = 0; s < nsteps; ++s) {
        const bool more = s + 1 < nsteps;
        W5STAMP(3)          // [3] matrix phase (incl. the DMA issue)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();          // filter and patch of k-step s landed; the MFMAs of k-step s - 1 are done with V and the other filter stage
        W5STAMP(0)          // [0] DMA wait + top barrier
        // ---- row pass ---------------------------------------------------------------------------------------------------------------
        {
            f32x4 ra[C::NRR][2];
#pragma unroll
            for (int k = 0; k < C::NRR; ++k) {
                const float *src = lds + r_src[k] + (s & 1) * C::PCAP;
                ra[k][0] = *(const f32x4 *)src;
                ra[k][1] = *(const f32x4 *)(src + 4);
            }
#pragma unroll
            for (int k = 0; k < C::NRR; ++k) {
                const float e[8] = {ra[k][0][0], ra[k][0][1], ra[k][0][2], ra[k][0][3], ra[k][1][0], ra[k][1][1], ra[k][1][2], ra[k][1][3]};
                float f[8];
                w5_bt(e, f);
                float *dst = lds + r_dst[k];
                *(f32x4 *)dst = f32x4{f[0], f[1], f[2], f[3]};
                *(f32x4 *)(dst + C::XPL) = f32x4{f[4], f[5], f[6], f[7]};
            }
        }
        W5STAMP(1)          // [1] row pass
        __builtin_amdgcn_wave_barrier();          // X of channel w is written and read by wave w only: the LDS serves a wave's accesses in order
        W5STAMP(4)          // [4] (no barrier behind the row pass)
        // ---- column pass ------------------------------------------------------------------------------------------------------------
        {
            f32x4 cx[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) cx[i] = *(const f32x4 *)(lds + c_src + i * C::XRW);
            float f[4][8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d[8] = {cx[0][e], cx[1][e], cx[2][e], cx[3][e], cx[4][e], cx[5][e], cx[6][e], cx[7][e]};
                // the values arrive as 16-byte quads: pin each as a scalar so that no packed-fp32 arithmetic is formed on neighbours
                // (check_isa.sh fences v_pk_*_f32; the four passes are the same arithmetic on the four elements of every quad)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(d[i]));
                w5_bt(d, f[e]);
            }
#pragma unroll
            for (int rf = 0; rf < 8; ++rf) *(f32x4 *)(lds + c_dst + rf * 2 * CK * C::NTP * 4) = f32x4{f[0][rf], f[1][rf], f[2][rf], f[3][rf]};
        }
        W5STAMP(2)          // [2] column pass
        __syncthreads();
        W5STAMP(5)          // [5] barrier behind the column pass
        // ---- 64 MFMAs: 16 quads of frequencies, operands of quad g + 2 fetched behind the first MFMA of quad g ------------------------
        {
            const int ai = aBase + (s & 1) * (C::USZ / 4), bi = bBase;
            f32x4 a[3], bq[3];
            a[0] = lds4[ai];
            bq[0] = lds4[bi];
            a[1] = lds4[ai + CK * 32];
            bq[1] = lds4[bi + CK * C::NTP];
#pragma unroll
            for (int g = 0; g < NFQ; ++g) {
                const int cur = g % 3, nxt = (g + 2) % 3;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[4 * g + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][e], bq[cur][e], acc[4 * g + e], 0, 0, 0);
                    if (e == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (g + 2 < NFQ) {
                            a[nxt] = lds4[ai + (g + 2) * CK * 32];
                            bq[nxt] = lds4[bi + (g + 2) * CK * C::NTP];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // the DMA of the next k-step, one instruction per quad behind its second MFMA (in a burst in front of the row pass
                    // the 11 instructions cost ~200 cycles each: 5600 cycles per k-step instead of 4300)
                    if (e == 1 && g < C::NIP + C::NIU && more) {
                        issue_n(s + 1, g);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    W5STAMP(3)
    {
        const int px = x0 + 4 * Tx, py = y0 + 4 * Ty;
        // one copy only: with 256 accumulator registers a second inlined copy of the epilogue (the FAST split of ssm_wino7.hip) spills 8
        // vector registers, and the epilogue is 9 % of a 64-channel workgroup's life
        w5_epilogue<false>(p, acc, bv, b, nb * 32 + cb * 16, q, px, py);
    }
#ifdef W5_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.dbg && lane == 0 && (blockIdx.x % 64) == 0) {          // a sample of the workgroups
        const unsigned long long tn = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 6; ++i) atomicAdd(p.dbg + wid * 8 + i, tph[i]);
        atomicAdd(p.dbg + wid * 8 + 6, tn - tk);          // epilogue
        atomicAdd(p.dbg + wid * 8 + 7, 1ULL);
        (void)tstart;
    }
#endif
}

#endif          // !W5_SPLIT (the r4 kernel, tuning builds)

#if W5_SPLIT
// =====================================================================================================================================
// The frequency-split form (r5, W5_SPLIT = 1, default): EIGHT waves per workgroup, two per SIMD.  The wave pair of a SIMD shares the 16
// couts x 16 tiles of one r4 wave and splits its 64 frequencies by column-frequency: half 0 = {0, 1, 2, 7}, half 1 = {3, 4, 5, 6} - the
// two halves of the 8-point transform B^T that share no sub-expression - 32 accumulators (128 registers) per wave.  With one wave per
// SIMD (r4) the vector instructions of the two passes could only run in front of the MFMAs (a lone wave cannot issue beside its own
// MFMA: 1880 + 2780 cycles per k-step for 2048 cycles of MFMA); a second wave on the SIMD issues them BESIDE the first wave's MFMAs
// (tools/mfma_valu_probe.py mode 1: an MFMA-only wave keeps its 32 cycles per instruction next to a vector-only wave).  Per k-step s
// two phases, two barriers:
//      phase A(s):  waves of half 0: 32 MFMAs on V_0(s) + the DMA of U(s+1)      | waves of half 1: row + column pass -> V_1(s)
//      phase B(s):  waves of half 1: 32 MFMAs on V_1(s) + the DMA of patch(s+3)  | waves of half 0: row + column pass -> V_0(s+1)
// so V needs no second buffer (each half is written in the phase in which the other half is multiplied), X is one plane set shared
// in time by the halves, the patch has three buffers (patch(k) is read in B(k-1) and A(k); requested two k-steps ahead), U two stages.
// Transform work of a wave = channel (wave & 3) of the k-step: row pass = 3 rounds of (2 x 16-byte read, 14 operations, 16-byte
// write), column pass = lane (tile, pair of the half's column-frequencies): 8 x 8-byte reads, 2 x 26 operations, 8 x 8-byte writes.
// Epilogue: the output transform is linear in the frequencies - each wave transforms its half (all 8 row-frequencies, 4 column-
// frequencies) into a partial 4x4 tile per cout, the pair exchanges partial tiles through LDS (each wave finishes two of the four
// couts of a lane): y = lo + hi + bias, addend, LeakyReLU, stores, fused 2x2 mean as in the r4 epilogue.
__device__ __forceinline__ void w5_bt_lo(const float (&e)[8], float (&f)[4]) {          // frequencies 0, 1, 2, 7 (the expressions of w5_bt)
    f[0] = (e[0] - e[6]) + 5.25f * (e[4] - e[2]);
    f[3] = (e[7] - e[1]) + 5.25f * (e[3] - e[5]);
    const float t1 = (e[2] + e[6]) - 4.25f * e[4], t2 = (e[1] + e[5]) - 4.25f * e[3];
    f[1] = t1 + t2;
    f[2] = t1 - t2;
}
__device__ __forceinline__ void w5_bt_hi(const float (&e)[8], float (&f)[4]) {          // frequencies 3, 4, 5, 6
    const float t3 = (e[6] + 0.25f * e[2]) - 1.25f * e[4], t4 = (0.5f * e[1] - 2.5f * e[3]) + 2.f * e[5];
    const float t5 = (e[6] + 4.f * e[2]) - 5.f * e[4], t6 = (2.f * e[1] - 2.5f * e[3]) + 0.5f * e[5];
    f[0] = t3 + t4;
    f[1] = t3 - t4;
    f[2] = t5 + t6;
    f[3] = t5 - t6;
}

template <class C>
struct W5SLds {          // LDS map of the split form (floats)
    static constexpr int NPB = 3;                                              // patch buffers
    // X row stride: the column pass reads rows 4 Ty + i as 8-byte pairs, 32 lanes = two tile rows per LDS pass: 4 rows apart must be half
    // the banks apart (4 XRW = 32 mod 64)
    static constexpr int XRW = C::NTX * 4 + 8, XPL = C::PH * XRW;
    static constexpr int XSZ1 = C::CK * XPL;                                   // ONE plane per channel (the halves use X in different phases)
    static constexpr int UOFF = 0, VOFF = 2 * C::USZ, XOFF = VOFF + C::VSZ, POFF = XOFF + XSZ1;
    static constexpr int BYTES = (POFF + NPB * C::PCAP) * 4;
    static constexpr int EXSZ = 8 * 2 * 4 * 64 * 4;                            // epilogue exchange: 8 waves x 2 couts x 4 rows x 64 lanes x 4
    static_assert(BYTES <= 160 * 1024 && EXSZ <= 2 * C::USZ, "LDS budget; the exchange area reuses the filter stages");
};

template <class C>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino5s_kernel(const W5Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using L = W5SLds<C>;
    constexpr int PH = C::PH, PW = C::PW, NTX = C::NTX, CK = C::CK, NTP = C::NTP;

    const int tid = threadIdx.x;
    const int lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wid & 1, tg = (wid >> 1) & 1, fh = wid >> 2;          // cout half, tile group, frequency half of this wave
    const int w4 = wid & 3;                                              // its channel of the k-step in the passes; its share of the DMA

    int id = ssm_xcd_tile(blockIdx.x, gridDim.x);
    const int nb = id % p.NB;
    id /= p.NB;
    const int tx = id % p.tilesX;
    id /= p.tilesX;
    const int ty = id % p.tilesY;
    const int b = id / p.tilesY;
    const int x0 = tx * C::TW, y0 = ty * C::TH;

    const float *pbase = p.src + (long long)b * p.sb + (long long)(y0 - 2) * p.sh + (x0 - 4);
    const float *wbase = p.wpk + (long long)nb * (p.Cin / CK) * C::USZ;

    // per-lane source offsets (bytes) of the patch pieces this wave brings per k-step (piece = 16 bytes of a patch row of one channel);
    // rows below the bottom zero frame / pieces right of the padded row are read from the frame's last row / piece (see ssm_wino7.hip)
    int poff[C::NIP];
#pragma unroll
    for (int i = 0; i < C::NIP; ++i) {
        const int qq = (i * 4 + w4) * 64 + lane;
        if (qq < C::NDQ) {
            const int c = qq / (PH * (PW / 4)), rem = qq - c * (PH * (PW / 4));
            const int r = rem / (PW / 4), j = rem - r * (PW / 4);
            const int re = min(r, p.H + (SSM_PADY - 1) - (y0 - 2)), fe = min(4 * j, ((p.W + 2 * SSM_PADX + 3) & ~3) - 4 - x0);
            poff[i] = ((int)(c * p.sc) + re * p.sh + fe) * 4;
        } else {
            poff[i] = 0;          // tail of the last 1-KiB piece: lands in the buffer's padding
        }
    }
    const int uoff = lane * 16;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
    auto dma = [](const float *base, int voff_bytes, unsigned m0v) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(base), "s"(m0v) : "memory", "m0");
    };
    // The LDS-DMA of a k-step rides in the matrix phases, one instruction per quad (a burst costs ~200 cycles a piece); its bases are formed
    // once per phase - a wave cannot issue beside its own MFMAs, so every scalar instruction in a matrix phase lengthens it by its issue
    // time (the first split kernel recomputed 64-bit bases per piece: 12 scalar instructions each).  Issued by the transforming waves
    // instead it measured 5 % slower (profiles/r11f_wino5_split_ab.txt): their phase, throttled to one vector instruction per ~12 cycles
    // beside the other wave's MFMAs, is the longer one.
    //   filter of k-step s: piece 4 k + w4 (k = 0 .. 7) of the 32, into stage s & 1;  patch of k-step s: piece 4 k + w4 (k < NIP) into buffer pb
    struct DmaBase {
        const float *src;
        unsigned m0;
    };
    auto u_base = [&](int s) {
        return DmaBase{wbase + (long long)s * C::USZ + w4 * 256,
                       (unsigned)__builtin_amdgcn_readfirstlane(lds0 + (unsigned)(L::UOFF + (s & 1) * C::USZ) * 4u + (unsigned)w4 * 1024u)};
    };
    auto p_base = [&](int s, int pb) {
        return DmaBase{pbase + (long long)(s * CK) * p.sc,
                       (unsigned)__builtin_amdgcn_readfirstlane(lds0 + (unsigned)(L::POFF + pb * C::PCAP + C::SHIFT) * 4u + (unsigned)w4 * 1024u)};
    };
    auto dma_u = [&](const DmaBase &bs, int k) { dma(bs.src + k * 1024, uoff, bs.m0 + (unsigned)k * 4096u); };
    auto dma_p = [&](const DmaBase &bs, int k) {
        if (4 * k + 3 < C::NGP || 4 * k + w4 < C::NGP) dma(bs.src, poff[k], bs.m0 + (unsigned)k * 4096u);
    };
    static_assert(C::NGU == 32, "8 filter pieces per wave of half 1: the transform's 8 DMA slots");

    f32x4 acc[32];
#pragma unroll
    for (int f = 0; f < 32; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- row pass: unit = (patch row, tile column) of channel w4, NRR rounds of 64 lanes (lanes beyond the last unit repeat it) --------
    int r_src[C::NRR], r_dst[C::NRR];
#pragma unroll
    for (int k = 0; k < C::NRR; ++k) {
        const int u = min(lane + 64 * k, C::NRU - 1);
        const int row = u / NTX, t = u - row * NTX;
        r_src[k] = L::POFF + C::SHIFT + (w4 * PH + row) * PW + 4 * t + 2;          // floats; 16-byte aligned
        r_dst[k] = L::XOFF + w4 * L::XPL + row * L::XRW + t * 4;
    }
    // ---- column pass: lane = (tile, pair cp of the half's four column-frequencies) of channel w4: the whole 8-point transform of two
    // columns (a split by row-frequency halves would diverge inside the wave: both halves' code for every lane) ---------------------------
    const int c_tile = lane >> 1, cp = lane & 1;
    const int c_gx = (c_tile & 15) % C::GTX, c_gy = (c_tile & 15) / C::GTX, c_g2 = c_tile >> 4;
    const int c_Tx = (c_g2 % C::WTX) * C::GTX + c_gx, c_Ty = (c_g2 / C::WTX) * C::GTY + c_gy;
    const int c_src = L::XOFF + w4 * L::XPL + (4 * c_Ty) * L::XRW + c_Tx * 4 + 2 * cp;
    const int c_dst = L::VOFF + ((fh * CK + w4) * NTP + c_tile) * 4 + 2 * cp;            // quad 2 rf + fh: + rf * 2 * CK * NTP * 4

    // ---- operand bases of the matrix phase (f32x4 units): U of (quad, channel q, cout cb*16 + l15), V of (quad, channel q, tile) ------
    const f32x4 *lds4 = (const f32x4 *)lds;
    const int aBase = L::UOFF / 4 + fh * (CK * 32) + q * 32 + cb * 16 + l15;
    const int bBase = L::VOFF / 4 + fh * (CK * NTP) + q * NTP + tg * 16 + l15;
    const int gx = l15 % C::GTX, gy = l15 / C::GTX;
    const int Tx = (tg % C::WTX) * C::GTX + gx, Ty = (tg / C::WTX) * C::GTY + gy;

    // both passes of this wave's channel for its frequency half: patch buffer pb -> V_fh; slot(0 .. 7): an LDS-DMA instruction each
    auto transform = [&](auto FH, int pb, auto slot) __attribute__((always_inline)) {
        constexpr int h = decltype(FH)::value;
        {
            f32x4 ra[C::NRR][2];
#pragma unroll
            for (int k = 0; k < C::NRR; ++k) {
                const float *src = lds + r_src[k] + pb * C::PCAP;
                ra[k][0] = *(const f32x4 *)src;
                ra[k][1] = *(const f32x4 *)(src + 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            slot(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < C::NRR; ++k) {
                const float e[8] = {ra[k][0][0], ra[k][0][1], ra[k][0][2], ra[k][0][3], ra[k][1][0], ra[k][1][1], ra[k][1][2], ra[k][1][3]};
                float f[4];
                if (h == 0) w5_bt_lo(e, f);
                else w5_bt_hi(e, f);
                *(f32x4 *)(lds + r_dst[k]) = f32x4{f[0], f[1], f[2], f[3]};
                __builtin_amdgcn_sched_barrier(0);
                slot(1 + k);          // (NRR = 3: slots 1 .. 3)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_wave_barrier();          // X of channel w4 is written and read by this wave only: the LDS serves a wave's accesses in order
        {
            f32x2 cx[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) cx[i] = *(const f32x2 *)(lds + c_src + i * L::XRW);
            __builtin_amdgcn_sched_barrier(0);
            slot(4);
            __builtin_amdgcn_sched_barrier(0);
            float f[2][8];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float d[8] = {cx[0][e], cx[1][e], cx[2][e], cx[3][e], cx[4][e], cx[5][e], cx[6][e], cx[7][e]};
                // (pinned as scalars: no packed-fp32 arithmetic on the two elements of a pair - check_isa.sh fences v_pk_*_f32)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(d[i]));
                w5_bt(d, f[e]);
                __builtin_amdgcn_sched_barrier(0);
                slot(5 + e);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int rf = 0; rf < 8; ++rf) *(f32x2 *)(lds + c_dst + rf * 2 * CK * NTP * 4) = f32x2{f[0][rf], f[1][rf]};
            __builtin_amdgcn_sched_barrier(0);
            slot(7);
        }
    };
    static_assert(C::NRR == 3, "DMA slots 1 .. 3 of the transform = the row pass's rounds");

    // 32 MFMAs: the 8 quads 2 g + fh of filter stage `st`; operands of quad g + 2 fetched behind the first MFMA of quad g; dmaf(g): one
    // LDS-DMA instruction behind the second MFMA of quad g (bases formed before the phase: 2 scalar adds + m0 per piece)
    auto matrix = [&](int st, auto dmaf) __attribute__((always_inline)) {
        const int ai = aBase + st * (C::USZ / 4), bi = bBase;
        f32x4 a[3], bq[3];
        a[0] = lds4[ai];
        bq[0] = lds4[bi];
        a[1] = lds4[ai + 2 * CK * 32];
        bq[1] = lds4[bi + 2 * CK * NTP];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int cur = g % 3, nxt = (g + 2) % 3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[4 * g + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][e], bq[cur][e], acc[4 * g + e], 0, 0, 0);
                if (e == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + 2 < 8) {
                        a[nxt] = lds4[ai + (g + 2) * 2 * CK * 32];
                        bq[nxt] = lds4[bi + (g + 2) * 2 * CK * NTP];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (e == 1) {
                    dmaf(g);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

#ifdef W5_TRACE
    // tuning build: per-wave cycle sums [0] matrix phases, [1] transform phases, [2] wait + barrier behind a matrix phase, [3] ... behind a
    // transform phase, [4] epilogue, [5] prologue
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tk = __builtin_amdgcn_s_memtime();
#define W5S(i)                                                      \
    {                                                               \
        const unsigned long long tn = __builtin_amdgcn_s_memtime(); \
        tph[i] += tn - tk;                                          \
        tk = tn;                                                    \
    }
#else
#define W5S(i)
#endif
    const int nsteps = p.Cin / CK;
    // ---- prologue: U(0) by the waves of half 0, patches 0 .. 2 by the waves of half 1 (the roles of the loop); then V_0(0) -------------
    if (fh == 0) {
        const DmaBase ub = u_base(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) dma_u(ub, k);
    } else {
        for (int s = 0; s < 3 && s < nsteps; ++s) {
            const DmaBase pbs = p_base(s, s);
#pragma unroll
            for (int k = 0; k < C::NIP; ++k) dma_p(pbs, k);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto noslot = [](int) {};
    if (fh == 0) transform(I0{}, 0, noslot);
    __syncthreads();
    W5S(5)

    const bool first = w4 < C::NGP - 4 * (C::NIP - 1);          // this wave brings NIP patch pieces per k-step (else NIP - 1)
    if (fh == 0) {
        int pb1 = 1 % L::NPB;          // patch buffer of k-step s + 1
        for (int s = 0; s < nsteps; ++s) {
            const bool more = s + 1 < nsteps;
            // phase A(s): 32 MFMAs on V_0(s), the filter of k-step s + 1 requested behind them (into the stage k-step s - 1 released)
            const DmaBase ub = u_base(more ? s + 1 : s);
            matrix(s & 1, [&](int g) {
                if (more) dma_u(ub, g);
            });
            W5S(0)
            __syncthreads();
            W5S(2)
            // phase B(s): V_0(s + 1)
            if (more) transform(I0{}, pb1, noslot);
            pb1 = pb1 + 1 == L::NPB ? 0 : pb1 + 1;
            W5S(1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the filter of k-step s + 1 (a whole phase old)
            __syncthreads();
            W5S(3)
        }
    } else {
        int pb = 0;                     // patch buffer of k-step s (= the buffer patch(s + 3) is requested into behind it)
        for (int s = 0; s < nsteps; ++s) {
            // phase A(s): V_1(s)
            transform(I1{}, pb, noslot);
            W5S(1)
            // patch(s + 1) is read from phase B(s) on: it must have landed; patch(s + 2), requested in B(s - 1), may still be in flight
            if (s >= 1 && s + 2 < nsteps) {
                if (first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NIP) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NIP - 1) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            W5S(3)
            // phase B(s): 32 MFMAs on V_1(s), patch(s + 3) requested behind them (into the buffer of patch(s): its last reader was A(s))
            const bool more3 = s + 3 < nsteps;
            const DmaBase pbs = p_base(more3 ? s + 3 : s, pb);
            matrix(s & 1, [&](int g) {
                if (g < C::NIP && more3) dma_p(pbs, g);
            });
            W5S(0)
            pb = pb + 1 == L::NPB ? 0 : pb + 1;
            __syncthreads();
            W5S(2)
        }
    }

    // ---- epilogue -------------------------------------------------------------------------------------------------------------------------
    // partial output transform of this wave's half: t[i][e] = sum_rf A^T[i][rf] M[rf][e] (all row-frequencies), then over its four column-
    // frequencies: half 0 (cf 0, 1, 2, 7): y0 = (m0 + s1), y1 = t1, y2 = s1, y3 = t1 + m7;  half 1 (cf 3 .. 6): y0 = s2 + s3,
    // y1 = 2 t2 + t3 / 2, y2 = 4 s2 + s3 / 4, y3 = 8 t2 + t3 / 8  (A^T of w5_at, split by column)
    auto epilogue = [&](auto FHC) __attribute__((always_inline)) {
    constexpr int fh = decltype(FHC)::value;          // (compile-time: the cout index r selects accumulator elements)
    auto partial = [&](int r, float (&y)[4][4]) __attribute__((always_inline)) {
        float t[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float y4[4];
            w5_at(acc[e][r], acc[4 + e][r], acc[8 + e][r], acc[12 + e][r], acc[16 + e][r], acc[20 + e][r], acc[24 + e][r], acc[28 + e][r], y4);
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i][e] = y4[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (fh == 0) {
                const float s1 = t[i][1] + t[i][2], d1 = t[i][1] - t[i][2];
                y[i][0] = t[i][0] + s1;
                y[i][1] = d1;
                y[i][2] = s1;
                y[i][3] = d1 + t[i][3];
            } else {
                const float s2 = t[i][0] + t[i][1], d2 = t[i][0] - t[i][1], s3 = t[i][2] + t[i][3], d3 = t[i][2] - t[i][3];
                y[i][0] = s2 + s3;
                y[i][1] = 2.f * d2 + 0.5f * d3;
                y[i][2] = 4.f * s2 + 0.25f * s3;
                y[i][3] = 8.f * d2 + 0.125f * d3;
            }
        }
    };
    // (every MFMA phase is behind the loop's last barrier: the filter stages are free - they hold the exchange area now)
    f32x4 *ex = (f32x4 *)lds;
    {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {          // the two couts the OTHER wave of the pair finishes
            float y[4][4];
            partial(2 * (1 - fh) + rr, y);
#pragma unroll
            for (int i = 0; i < 4; ++i) ex[((wid * 2 + rr) * 4 + i) * 64 + lane] = f32x4{y[i][0], y[i][1], y[i][2], y[i][3]};
        }
    }
    __syncthreads();
    {
        const int px = x0 + 4 * Tx, py = y0 + 4 * Ty;
        const int cu0 = nb * 32 + cb * 16;
        const float sl = (p.lrelu & 1) ? p.slope : 1.f;
    const bool amask = (p.lrelu & 2) != 0;          // SSM_FLAG_MASK: the addend view is a mask source (see ssm_hip.h; r6: the 5x5 data gradients too)
        float *dstb = p.dst + (long long)b * p.dsb;
        float *poolb = p.pool ? p.pool + (long long)b * p.psb : nullptr;
        const unsigned pbo = 4u * ((unsigned)(4 * q) * (unsigned)p.dsc + (unsigned)py * (unsigned)p.dsh + (unsigned)px);
        const unsigned qbo = 4u * ((unsigned)(4 * q) * (unsigned)p.psc + (unsigned)(py >> 1) * (unsigned)p.psh + (unsigned)(px >> 1));
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
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * fh + rr;
            const int cu = cu0 + r;          // uniform; this lane's cout = cu + 4 * q
            const float bvr = p.bias[cu + 4 * q];
            float y[4][4];
            partial(r, y);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 o = ex[(((wid ^ 4) * 2 + rr) * 4 + i) * 64 + lane];
                // lo + hi in the same order whichever wave finishes the cout
#pragma unroll
                for (int e = 0; e < 4; ++e) y[i][e] = ((fh == 0 ? y[i][e] + o[e] : o[e] + y[i][e])) + bvr;
            }
            if (p.add) {          // (uniform; addb is a per-lane pointer)
                const float *ap = addb + (long long)cu * p.asc;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (vok) {
                        const f32x4 z = *(const f32x4 *)(ap + (long long)i * p.ash);
                        if (amask) {
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
                                y[i][e] = amask ? y[i][e] * (z > 0.f ? 1.f : p.slope) : y[i][e] + z;
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
                    st4(bp + (long long)i * p.dsh, pbo, f32x4{y[i][0], y[i][1], y[i][2], y[i][3]});
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (py + i < p.H && px + e < p.W) st1(bp + (long long)i * p.dsh + e, pbo, y[i][e]);
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
                    if (rok && px + 4 <= p.W && p.vec) st2(qp + (long long)i * p.psh, qbo, f32x2{o0, o1});
                    else if (rok) {
                        if (px + 2 <= p.W) st1(qp + (long long)i * p.psh, qbo, o0);
                        if (px + 4 <= p.W) st1(qp + (long long)i * p.psh + 1, qbo, o1);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    };
    if (fh == 0) epilogue(I0{});
    else epilogue(I1{});
#ifdef W5_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W5S(4)
    if (p.dbg && lane == 0 && (blockIdx.x % 64) == 0) {          // a sample of the workgroups
        for (int i = 0; i < 6; ++i) atomicAdd(p.dbg + wid * 8 + i, tph[i]);
        atomicAdd(p.dbg + wid * 8 + 7, 1ULL);
    }
#endif
}

#endif          // W5_SPLIT

// ---- tile configurations ---------------------------------------------------------------------------------------------------------
//                     GTX WTY WTX          tiles of 4x4 px     TH   TW
using F5A = W5Cfg<8, 2, 1>;      //          8 x 4                16   32
using F5B = W5Cfg<4, 1, 2>;      //          8 x 4 (4x4 groups)   16   32

#define SSM_W5_KINDS(X) X(F5A_, F5A) X(F5B_, F5B)

enum W5Kind {
#define X(name, cfg) name,
    SSM_W5_KINDS(X)
#undef X
        NW5KIND
};

std::atomic<int> g_force_w5kind{-1};
#ifdef W5_TRACE
std::atomic<unsigned long long *> g_w5dbg{nullptr};
#endif

template <class C>
int w5launch(W5Params &p, int B, hipStream_t st) {
    p.tilesX = (p.W + C::TW - 1) / C::TW;
    p.tilesY = (p.H + C::TH - 1) / C::TH;
    p.NB = p.Cout / 32;
    // (no read outside the padded plane: the per-lane DMA offsets clamp overshoot rows / pieces to the zero frame, see wino5_kernel)
    const long long blocks = (long long)p.tilesX * p.tilesY * p.NB * B;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        ssm::set_error("wino5 conv: grid of %lld workgroups out of range", blocks);
        return SSM_E_ARG;
    }
#if W5_SPLIT
    void (*kern)(const W5Params) = wino5s_kernel<C>;
    constexpr int lds_bytes = W5SLds<C>::BYTES, threads = 512;
#else
    void (*kern)(const W5Params) = wino5_kernel<C>;
    constexpr int lds_bytes = C::BYTES, threads = 256;
#endif
    static std::atomic<uint64_t> lds_reserved{0};          // one bit per device: the attribute is per (kernel, device)
    const hipError_t attr_rc = ssm::reserve_lds(lds_reserved, (const void *)kern, lds_bytes);
    if (attr_rc != hipSuccess) {
        ssm::set_error("wino5 conv: cannot reserve %d bytes of LDS: %s", lds_bytes, hipGetErrorString(attr_rc));
        return SSM_E_LAUNCH;
    }
    SSM_LAUNCH(kern, dim3((unsigned)blocks), dim3(threads), lds_bytes, st, p);
    return ssm::check_launch("ssm_wino5_conv2d_add_fwd");
}

int w5dispatch(int kind, W5Params &p, int B, hipStream_t st) {
    switch (kind) {
#define X(name, cfg) \
    case name: return w5launch<cfg>(p, B, st);
        SSM_W5_KINDS(X)
#undef X
    }
    return SSM_E_UNSUPPORTED;
}

__global__ void wino5_pack_kernel(const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ wp, float *__restrict__ bp,
                                  int Cout, int Cin, int CinP, long long total, int nbias) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i < total) {
        float out[4];
        auto at = [&](int co, int ci, int ky, int kx) { return w[(((long long)co * Cin + ci) * 5 + ky) * 5 + kx]; };
        ssm_w5_pack_quad(at, Cout, Cin, CinP, i, out);
        *reinterpret_cast<f32x4 *>(wp + i) = f32x4{out[0], out[1], out[2], out[3]};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (i + e < nbias) bp[i + e] = (i + e < Cout) ? bias[i + e] : 0.f;
}

}  // namespace

extern "C" int ssm_wino5_plan(int Cin, int Cout, int B, int H, int W, int *kind) {
    if (Cin <= 0 || Cout <= 0 || Cout % 32) {
        ssm::set_error("wino5 conv: no tile configuration for Cin=%d Cout=%d (Cout a multiple of 32)", Cin, Cout);
        return SSM_E_UNSUPPORTED;
    }
    const int forced = g_force_w5kind.load();
    if (kind) *kind = (forced >= 0 && forced < NW5KIND) ? forced : 0;
    return SSM_OK;
}

#ifdef W5_TRACE
// tuning build only (-DW5_TRACE=1; never lib/libssm_hip.so): 4 x 8 device counters, per wave of the sampled workgroups the shader cycles in
// [top wait, row pass, column pass, matrix phase, barrier 2, barrier 3, epilogue] and the number of samples
extern "C" int ssm_wino5_debug_buffer(unsigned long long *dev_counters) {
    g_w5dbg.store(dev_counters);
    return SSM_OK;
}
#endif

extern "C" int ssm_wino5_force_kind(int kind) {
    g_force_w5kind.store(kind >= 0 && kind < NW5KIND ? kind : -1);
    return NW5KIND;
}

extern "C" size_t ssm_wino5_packed_weight_floats(int Cout, int CinP) { return (size_t)(Cout / 32) * (size_t)(CinP / 4) * 16 * 4 * 32 * 4; }

extern "C" int ssm_wino5_pack_weights(const float *w, const float *bias, float *wp, float *bp, int Cout, int Cin, int CinP, void *stream) {
    SSM_REQUIRE(w && bias && wp && bp, "wino5 pack_weights: null pointer");
    SSM_REQUIRE(Cout > 0 && Cin > 0 && Cout % 32 == 0 && CinP >= Cin && CinP % 4 == 0, "wino5 pack_weights: bad sizes (Cout a multiple of 32, CinP of 4)");
    SSM_REQUIRE(ssm::aligned16(wp), "wino5 pack_weights: the packed filter must be 16-byte aligned");
    const long long total = (long long)ssm_wino5_packed_weight_floats(Cout, CinP);
    const int nbias = Cout;
    const long long n = (total > nbias ? total : nbias) / 4 + 1;
    SSM_LAUNCH(wino5_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, bias, wp, bp, Cout, Cin, CinP,
                       total, nbias);
    return ssm::check_launch("ssm_wino5_pack_weights");
}

extern "C" int ssm_wino5_conv2d_add_fwd(ssm_view x, int Cin, const float *w_packed, const float *bias_packed, ssm_view y, ssm_view pool,
                                        ssm_view add, int add_div, int B, int H, int W, int Cout, float slope, int flags, void *stream) {
    int kind = 0;
    SSM_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && Cin > 0, "wino5 conv: bad sizes");
    const int rc = ssm_wino5_plan(Cin, Cout, B, H, W, &kind);
    if (rc != SSM_OK) return rc;
    SSM_REQUIRE(x.ptr && y.ptr && w_packed && bias_packed, "wino5 conv: null pointer");
    SSM_REQUIRE(Cin % 4 == 0, "wino5 conv: the channel count (%d) must be a multiple of 4 (pad the view)", Cin);
    SSM_REQUIRE(ssm::aligned16(x.ptr) && x.sh % 4 == 0 && x.sc % 4 == 0 && x.sb % 4 == 0,
                "wino5 conv: the input is not a padded-plane view (16-byte alignment)");
    SSM_REQUIRE(x.sh >= W + 2 * SSM_PADX, "wino5 conv: input row stride %d leaves no zero frame for W=%d", x.sh, W);
    SSM_REQUIRE(ssm::aligned16(w_packed), "wino5 conv: packed filter must be 16-byte aligned");
    SSM_REQUIRE(4LL * x.sc * 4 < 0x7fffffffLL, "wino5 conv: channel stride too large");
    W5Params p;
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
    p.lrelu = ((flags & SSM_FLAG_LRELU) ? 1 : 0) | ((flags & SSM_FLAG_MASK) ? 2 : 0);
    p.add = nullptr;
    p.asb = p.asc = 0;
    p.ash = 0;
    p.adiv = 1;
    bool vec = W % 4 == 0 && ssm::aligned16(y.ptr) && y.sh % 4 == 0 && y.sc % 4 == 0 && y.sb % 4 == 0;
    if (add.ptr) {
        SSM_REQUIRE(add_div >= 1 && B % add_div == 0, "wino5 conv: the addend serves %d batch entries each, batch %d is no multiple", add_div, B);
        p.add = add.ptr;
        p.asb = add.sb;
        p.asc = add.sc;
        p.ash = add.sh;
        p.adiv = add_div;
        vec = vec && ssm::aligned16(add.ptr) && add.sh % 4 == 0 && add.sc % 4 == 0 && add.sb % 4 == 0;
    }
    if (pool.ptr) {
        SSM_REQUIRE(H % 2 == 0 && W % 2 == 0, "wino5 conv: fused pool needs even H, W");
        p.pool = pool.ptr;
        p.psb = pool.sb;
        p.psc = pool.sc;
        p.psh = pool.sh;
        vec = vec && (reinterpret_cast<size_t>(pool.ptr) & 7) == 0 && pool.sh % 2 == 0 && pool.sc % 2 == 0 && pool.sb % 2 == 0;
    }
    p.vec = vec ? 1 : 0;
#ifdef W5_TRACE
    p.dbg = g_w5dbg.load();
#else
    p.dbg = nullptr;
#endif
    return w5dispatch(kind, p, B, (hipStream_t)stream);
}
