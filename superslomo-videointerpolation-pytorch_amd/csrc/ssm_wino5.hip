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
    void (*kern)(const W5Params) = wino5_kernel<C>;
    constexpr int lds_bytes = C::BYTES;
    static std::once_flag once;
    static hipError_t attr_rc = hipSuccess;
    std::call_once(once, [&] { attr_rc = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); });
    if (attr_rc != hipSuccess) {
        ssm::set_error("wino5 conv: cannot reserve %d bytes of LDS: %s", lds_bytes, hipGetErrorString(attr_rc));
        return SSM_E_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds_bytes, st, p);
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
    hipLaunchKernelGGL(wino5_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, bias, wp, bp, Cout, Cin, CinP,
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
    p.lrelu = (flags & SSM_FLAG_LRELU) ? 1 : 0;
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
