// fp16-MFMA implicit-GEMM convolution (v_mfma_f32_32x32x16_f16) on "HL8" activations.
//
// Same operator as ssm_conv.hip (layers.conv / final_conv of the reference,
// scripts/models/layers.py:21-33, flow_computation.py:145-153) at 16x the matrix-core rate.
// Every fp32 value x is carried as TWO fp16 numbers  hi = fp16(x), lo = fp16(x - hi)
// (22-23 significant bits together) and a product is evaluated as
//        a*b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (fp32 accumulate)
// i.e. three MFMAs per 32x32x16 block: 16/3 = 5.3x the fp32-MFMA rate at fp32-grade accuracy
// (mode SPLIT3).  Mode FAST issues only a_hi*b_hi (plain fp16 inputs, fp32 accumulate): the
// reduced-precision path for 4K (BASELINE config 5), judged by PSNR instead of 1e-3.
//
// HL8 layout (include/ssm_hip.h): [B][C/8][2 = hi|lo][Hp][Wp][8 x fp16], zero frame like the
// fp32 padded planes.  A pixel's 8 channels are one 16-byte unit = one MFMA operand fragment
// (lane = pixel, 8 consecutive k = 8 channels), fetched from LDS with ONE conflict-free
// ds_read_b128; a k-step of 16 = channel groups (2q, 2q+1) on lane halves at one filter tap.
//
// Pipeline per workgroup: K loop over 16-channel chunks x filter-row stages.  The activation
// patch [4 planes][TH+k-1][TW+k-1] x 16 B is double-buffered per chunk, the filter stage
// [KYS*k taps][2][2][BN] x 16 B per iteration, both by LDS-DMA; the next chunk's patch is
// fetched in slices spread over the current chunk's iterations.  One barrier per iteration.
#include "ssm_common.h"

#include <mutex>
#include <set>
#include <utility>

#ifndef SSM_C16_SCHED
#define SSM_C16_SCHED 2
#endif
#ifndef SSM_C16_PRIO
#define SSM_C16_PRIO 0
#endif
#ifndef SSM_C16_PRODUCERS   // 1: DMA-only producer waves stage the operands (0: every wave issues its share of the DMA)
#define SSM_C16_PRODUCERS 1
#endif
#ifndef SSM_C16_ABL      // diagnostics (wrong results): 1 = no DMA in the loop, 2 = also no barriers, 3 = also no LDS reads
#define SSM_C16_ABL 0
#endif

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct Conv16Params {
    const char *src1, *src2;   // first interior pixel of group 0, hi plane
    long long sb1, sb2;        // batch strides (16-byte pixels)
    long long sg, sp;          // group stride, hi->lo stride (pixels), both sources
    int sh;                    // row stride (pixels)
    int C1, Cin;               // channels of source 1 / total (multiples of 16)
    const char *wpk;
    const float *bias;
    float wscale;              // 2^-s: filters are stored multiplied by 2^s
    char *dh;                  // HL8 destination (or null)
    long long dhsb, dhsg, dhsp;
    int dhsh;
    float *df;                 // fp32 destination view (or null)
    long long dfsb, dfsc;
    int dfsh;
    char *ph;                  // HL8 pooled destination (or null)
    long long phsb, phsg, phsp;
    int phsh;
    int H, W, Cout;
    int tilesX, tilesY, NB;
    float slope;
    int lrelu;
    int shuf;                  // 0, or Cr: the Cout = 4*Cr outputs are the four sub-pixel parities (pa, pb) of Cr channels; output
                               // (y, x) of channel (2*pa + pb)*Cr + c is stored at pixel (2y + pa, 2x + pb) of channel c
    int shuf_t;                // shuf with the kernel's axes swapped relative to the image: pixel (2x + pb, 2y + pa)
    int skipy, skipx;          // shuf: do not store the first / last row (column) of this launch's extent (another problem owns them)
};

// several problems of one tile configuration in one launch (sub-pixel decoder level: main + border strips + corners, disjoint outputs)
#define SSM_MULTI_MAX 12
struct Conv16Multi {
    const Conv16Params *table;     // device memory
    int n;
    int start[SSM_MULTI_MAX + 1];  // first workgroup of each problem; start[n] = grid size
};

// MODE_: 0 = plain fp16 (hi*hi), 1 = split (3 fp16 MFMAs per product), 2 = Q8 (1 fp16 MFMA + 2 block-scaled fp8 MFMAs per product)
template <int KS_, int KYS_, int NT_, int WN_, int MTY_, int MTX_, int WY_, int WX_, int PBUFS_, int MODE_>
struct Cfg16 {
    static constexpr int KS = KS_, KYS = KYS_, NT = NT_, WN = WN_, MTY = MTY_, MTX = MTX_, WY = WY_, WX = WX_;
    static constexpr int PBUFS = PBUFS_;      // 2: patch double-buffered in the workgroup (1 workgroup / CU);
                                              // 1: single patch buffer, latency hidden by a 2nd workgroup on the CU
    static constexpr bool SPLIT3 = MODE_ == 1, Q8 = MODE_ == 2;
    // FAST (plain fp16, BASELINE config 5): only the hi planes exist for this mode - the patch holds the two channel groups' hi planes,
    // the epilogue stores hi only (the lo planes of a plan's tensors stay at their initial zeros: no convolution of the mode reads them,
    // and hi + 0 is what the other readers - the fused-upsample expander, to_nchw - then see).  Half the HBM and LDS-DMA bytes of a
    // launch: the 32-channel full-resolution 3x3 layers are HBM-bound at fp16 (ridge 312 FLOP/B; profiles/r17l_layers16_4k_fast.txt).
    static constexpr bool FAST = MODE_ == 0;
    static constexpr int NPL = FAST ? 2 : 4;     // planes of a 16-channel chunk in the patch: group x (hi | second plane)
    static constexpr int NQ = (KS + 3) / 4;      // K=64 correction steps per filter row (4 taps x 16 channels each)
    static constexpr int PAD = (KS - 1) / 2;
    static constexpr int NW = WN * WY * WX, NTHREADS = 64 * NW;
    static constexpr int BN = 32 * NT * WN, TH = MTY * WY, TW = 32 * MTX * WX, MT = MTY * MTX;
    static constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
    static constexpr int NIT = KS / KYS;                          // iterations (filter-row stages) per chunk
    static constexpr int PATCH_PIECES = NPL * PH * PW;            // 16-byte pieces
    static constexpr int PNI = (PATCH_PIECES + 63) / 64;          // 1-KiB DMA instructions per patch
    static constexpr int PATCH_BYTES = PNI * 1024;
    // filter stage: [tap][h][part][BN] x 16 B;  Q8: [tap][h][BN] (fp16 hi) then [row][quad][operand][piece][half][BN] (fp8)
    static constexpr int WST_PIECES = Q8 ? KYS * KS * 2 * BN + KYS * NQ * 8 * BN : KYS * KS * 4 * BN;
    static constexpr int WNI = WST_PIECES / 64;
    static constexpr int WST_BYTES = WST_PIECES * 16;
    static constexpr int LDS_BYTES = PBUFS * PATCH_BYTES + 2 * WST_BYTES;
    static constexpr int BLOCKS_PER_CU = (2 * LDS_BYTES <= 160 * 1024 && NW <= 4) ? 2 : 1;
    // Producer waves: NP extra waves that only issue the LDS-DMA of the next filter stage / patch slice, so the matrix
    // waves' in-order instruction streams hold nothing but ds_read_b128 and MFMA (each global_load_lds costs its
    // issuing wave 60-180 cycles; ablation in profiles/README.md r1h).
    static constexpr int NP = (SSM_C16_PRODUCERS == 2 || (SSM_C16_PRODUCERS == 1 && NW >= 8)) ? 4 : 0;   // 4-wave tiles: iterations too short, measured slower
    static constexpr int NLOAD = NP ? NP : NW;                    // waves that share the DMA instructions
    static constexpr int NTHREADS_ALL = 64 * (NW + NP);
    static constexpr int PM = (PNI + NLOAD - 1) / NLOAD;          // patch DMA instructions per loading wave
    static constexpr int WM = (WNI + NLOAD - 1) / NLOAD;          // filter DMA instructions per loading wave per stage
    static_assert(KS % KYS == 0, "filter rows per stage must divide k");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(WST_PIECES % 64 == 0, "filter stage is whole DMA instructions");
};

#define SSM_GLDS16B(gp, lp)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp),       \
                                     (__attribute__((address_space(3))) void *)(lp), 16, 0, 0)

__device__ __forceinline__ void hi_store(char *plane_hi, float v0, float v1, float v2, float v3) {          // mode FAST: the hi plane only
    h4 hi;
    hi[0] = (_Float16)v0; hi[1] = (_Float16)v1; hi[2] = (_Float16)v2; hi[3] = (_Float16)v3;
    *reinterpret_cast<h4 *>(plane_hi) = hi;
}

__device__ __forceinline__ void split_store(char *plane_hi, long long sp_bytes, float v0, float v1, float v2, float v3) {
    h4 hi, lo;
    hi[0] = (_Float16)v0; hi[1] = (_Float16)v1; hi[2] = (_Float16)v2; hi[3] = (_Float16)v3;
    lo[0] = (_Float16)(v0 - (float)hi[0]); lo[1] = (_Float16)(v1 - (float)hi[1]);
    lo[2] = (_Float16)(v2 - (float)hi[2]); lo[3] = (_Float16)(v3 - (float)hi[3]);
    *reinterpret_cast<h4 *>(plane_hi) = hi;
    *reinterpret_cast<h4 *>(plane_hi + sp_bytes) = lo;
}

// ---- Q8 layout: plane 1 of a pixel group holds [8 x fp8(x) | 8 x fp8((x - fp16(x)) * 2^11)] instead of 8 x fp16(lo) ----
__device__ __forceinline__ float clamp448(float v) { return __builtin_amdgcn_fmed3f(v, -448.0f, 448.0f); }      // e4m3fn: beyond 448 -> NaN

__device__ __forceinline__ int pack4_fp8(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(a), clamp448(b), 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(c), clamp448(d), w, true);
}

// 4 consecutive channels (offset half*4 inside 8-channel group g) of one pixel.  `pix_hi` = the pixel's 16-byte record in the
// hi plane of group g.  Second planes are shared by a PAIR of groups (16 channels = one K chunk): the even group's holds
// [fp8(x) of the even group | fp8(x) of the odd group], the odd group's [fp8(lo*2^11) even | fp8(lo*2^11) odd], so that one
// 16-byte read per tap is half of a 32-byte fp8 MFMA operand with no register shuffling.
__device__ __forceinline__ void split_store_q8(char *pix_hi, long long sp_bytes, long long sg_bytes, int g, int half, float v0, float v1,
                                               float v2, float v3) {
    h4 hi;
    hi[0] = (_Float16)v0; hi[1] = (_Float16)v1; hi[2] = (_Float16)v2; hi[3] = (_Float16)v3;
    *reinterpret_cast<h4 *>(pix_hi + half * 8) = hi;
    const int odd = g & 1;
    char *even_rec = pix_hi - odd * sg_bytes + sp_bytes;          // second plane of the even group of the pair
    *reinterpret_cast<int *>(even_rec + odd * 8 + half * 4) = pack4_fp8(v0, v1, v2, v3);
    *reinterpret_cast<int *>(even_rec + sg_bytes + odd * 8 + half * 4) =
        pack4_fp8((v0 - (float)hi[0]) * 2048.0f, (v1 - (float)hi[1]) * 2048.0f, (v2 - (float)hi[2]) * 2048.0f, (v3 - (float)hi[3]) * 2048.0f);
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float fp8_byte(int word, int i) {          // byte i of a word of four e4m3 values -> fp32
    switch (i) {
        case 0: return __builtin_amdgcn_cvt_f32_fp8(word, 0);
        case 1: return __builtin_amdgcn_cvt_f32_fp8(word, 1);
        case 2: return __builtin_amdgcn_cvt_f32_fp8(word, 2);
        default: return __builtin_amdgcn_cvt_f32_fp8(word, 3);
    }
}

// One pipeline iteration in Q8 mode.  Per filter tap ONE fp16 MFMA (a_hi*b_hi); per filter row and group of 4 taps two
// block-scaled fp8 MFMAs with K = 64 = 4 taps x 16 channels: fp8(a)*fp8(b_lo*2^11) and fp8(a_lo*2^11)*fp8(b), each with the
// E8M0 scale 2^-11 on the "lo" operand.  Lane half h covers taps (4j+2h, 4j+2h+1) of quad j; taps beyond the filter row have
// zero filter bytes and re-read the row's last tap on the activation side.
//   sb / sa   : lane bases of the fp16 operands (group = lane half)        sbq / saq : lane bases of the q operands
template <class C>
__device__ __forceinline__ void conv16_compute_q8(f32x16 (&acc)[C::NT][C::MT], const char *sb, const char *sa, const char *sbq,
                                                  const char *saq, int hoff16) {
    constexpr int KS = C::KS, KYS = C::KYS, BN = C::BN, PH = C::PH, PW = C::PW, NT = C::NT, MT = C::MT, NQ = C::NQ;
    constexpr int ATAP = 2 * BN * 16;                 // fp16 filter bytes per tap
    constexpr int GQ = 2 * PH * PW * 16;              // fp8(x) plane of the chunk -> fp8(lo) plane
    constexpr int SC_HI = 0x7f7f7f7f, SC_LO = 0x74747474;      // E8M0 127 = 2^0, 116 = 2^-11
    constexpr int NQR = 4 * (NT + MT);                // fp8 operand reads (16 bytes each) per quad
    h8 ah[2][NT], bh[2][MT];
    auto fetch_one = [&](int tl, int r, h8 (&fa)[NT], h8 (&fb)[MT]) {
        const int kyy = tl / KS, kx = tl - kyy * KS;
        if (r < NT) {
            fa[r] = *reinterpret_cast<const h8 *>(sa + tl * ATAP + r * 32 * 16);
        } else {
            const int m = r - NT, my = m / C::MTX, mx = m - my * C::MTX;
            fb[m] = *reinterpret_cast<const h8 *>(sb + ((my + kyy) * PW + mx * 32 + kx) * 16);
        }
    };
    i32x8 a8[NT][2], b8[MT][2];                       // [tile][0 = fp8(x), 1 = fp8(lo * 2^11)]: two 16-byte reads each
    // fp8 operand read r of quad (ry, j): r < 4*NT: filter (tile, operand, piece); else activation (tile, plane, piece)
    auto q_read = [&](int ry, int j, int r) {
        if (r < 4 * NT) {
            const int n = r >> 2, op = (r >> 1) & 1, pc = r & 1;
            const i32x4 v = *reinterpret_cast<const i32x4 *>(saq + ((((ry * NQ + j) * 2 + op) * 2 + pc) * 2) * BN * 16 + n * 32 * 16);
            a8[n][op][4 * pc + 0] = v[0]; a8[n][op][4 * pc + 1] = v[1]; a8[n][op][4 * pc + 2] = v[2]; a8[n][op][4 * pc + 3] = v[3];
        } else {
            const int q = r - 4 * NT, m = q >> 2, pl = (q >> 1) & 1, pc = q & 1;
            const int k0 = 4 * j + pc < KS - 1 ? 4 * j + pc : KS - 1;              // tap of lane half 0
            const int k1 = 4 * j + 2 + pc < KS - 1 ? 4 * j + 2 + pc : KS - 1;      // tap of lane half 1
            const int my = m / C::MTX, mx = m - my * C::MTX;
            const int o = ((my + ry) * PW + mx * 32 + k0) * 16 + pl * GQ;
            const int hd = (k1 - k0) == 0 ? 0 : ((k1 - k0) == 1 ? hoff16 : 2 * hoff16);
            const i32x4 v = *reinterpret_cast<const i32x4 *>(sbq + o + hd);
            b8[m][pl][4 * pc + 0] = v[0]; b8[m][pl][4 * pc + 1] = v[1]; b8[m][pl][4 * pc + 2] = v[2]; b8[m][pl][4 * pc + 3] = v[3];
        }
    };
#pragma unroll
    for (int r = 0; r < NT + MT; ++r) fetch_one(0, r, ah[0], bh[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tl = 0; tl < KYS * KS; ++tl) {
        const int cur = tl & 1;
        const int ry = tl / KS, kx = tl - ry * KS, j = kx / 4;
        const bool more = tl + 1 < KYS * KS;
        const int qtaps = KS - 4 * j < 4 ? KS - 4 * j : 4;          // real taps in this quad
        const int gaps = qtaps * NT * MT;                             // MFMA gaps the quad's fp8 reads are spread over
        const int per = (NQR + gaps - 1) / gaps;                      // fp8 reads per gap
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int idx = n * MT + m;
                acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][n], bh[cur][m], acc[n][m], 0, 0, 0);
                if (more && idx < NT + MT) fetch_one(tl + 1, idx, ah[cur ^ 1], bh[cur ^ 1]);
                const int gq = (kx & 3) * NT * MT + idx;              // gap index inside the quad
#pragma unroll
                for (int u = 0; u < per; ++u)
                    if (gq * per + u < NQR) q_read(ry, j, gq * per + u);
                __builtin_amdgcn_sched_barrier(0);
            }
        if (more) {
#pragma unroll
            for (int r = NT * MT; r < NT + MT; ++r) fetch_one(tl + 1, r, ah[cur ^ 1], bh[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((kx & 3) == 3 || kx == KS - 1) {          // last tap of the quad: the two correction products
#pragma unroll
            for (int op = 0; op < 2; ++op)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[n][m] = op == 0 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[n][0], b8[m][1], acc[n][m], 0, 0, 0, SC_HI, 0, SC_LO)
                                            : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[n][1], b8[m][0], acc[n][m], 0, 0, 0, SC_LO, 0, SC_HI);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// One pipeline iteration of a wave: KYS*KS filter taps x (NT x MT) 32x32 tiles.  sb / sa = this lane's base
// addresses in the activation patch / filter stage (LDS).  Operand fragments are fetched one tap ahead of the
// MFMAs that consume them.
template <class C>
__device__ __forceinline__ void conv16_compute(f32x16 (&acc)[C::NT][C::MT], const char *sb, const char *sa) {
    constexpr int KS = C::KS, KYS = C::KYS, BN = C::BN, PH = C::PH, PW = C::PW, NT = C::NT, MT = C::MT;
    constexpr int B_LO = PH * PW * 16;     // hi -> lo plane inside the patch
    constexpr int A_LO = BN * 16;          // hi -> lo inside a filter tap
    // operand fragments of tap tl (software-pipelined one tap ahead of the MFMAs that consume them)
    auto fetch = [&](int tl, h8 (&ah)[NT], h8 (&al)[NT], h8 (&bh)[MT], h8 (&bl)[MT]) {
        const int kyy = tl / KS, kx = tl - kyy * KS;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            ah[n] = *reinterpret_cast<const h8 *>(sa + (tl * 4 * BN + n * 32) * 16);
            if (C::SPLIT3) al[n] = *reinterpret_cast<const h8 *>(sa + (tl * 4 * BN + n * 32) * 16 + A_LO);
        }
#pragma unroll
        for (int my = 0; my < C::MTY; ++my)
#pragma unroll
            for (int mx = 0; mx < C::MTX; ++mx) {
                const int o = ((my + kyy) * PW + mx * 32 + kx) * 16;
                bh[my * C::MTX + mx] = *reinterpret_cast<const h8 *>(sb + o);
                if (C::SPLIT3) bl[my * C::MTX + mx] = *reinterpret_cast<const h8 *>(sb + o + B_LO);
            }
    };
    h8 ah[2][NT], al[2][NT], bh[2][MT], bl[2][MT];
    fetch(0, ah[0], al[0], bh[0], bl[0]);
#if SSM_C16_SCHED == 2
    // Source order = issue order (a sched_barrier after every step): MFMAs round-robin over the accumulators (no
    // back-to-back dependence), and the next tap's ds_read_b128 go ONE per MFMA gap.  A burst of reads ahead of the
    // MFMAs blocks the wave's in-order issue behind the LDS queue while the matrix pipe idles (ablation: r1h).
    constexpr int NA = (C::SPLIT3 ? 2 : 1) * NT;                 // A-fragment reads per tap
    constexpr int NRD = NA + (C::SPLIT3 ? 2 : 1) * MT;           // ds_read_b128 per tap
    constexpr int NPR = C::SPLIT3 ? 3 : 1;                       // products per (n, m)
    constexpr int NMF = NPR * NT * MT;
    auto fetch_one = [&](int tl, int r, h8 (&fah)[NT], h8 (&fal)[NT], h8 (&fbh)[MT], h8 (&fbl)[MT]) {
        const int kyy = tl / KS, kx = tl - kyy * KS;
        if (r < NA) {
            const int n = r % NT, lo = r / NT;
            const h8 v = *reinterpret_cast<const h8 *>(sa + (tl * 4 * BN + n * 32) * 16 + lo * A_LO);
            if (lo) fal[n] = v; else fah[n] = v;
        } else {
            const int q = r - NA, m = q % MT, lo = q / MT;
            const int my = m / C::MTX, mx = m - my * C::MTX;
            const h8 v = *reinterpret_cast<const h8 *>(sb + ((my + kyy) * PW + mx * 32 + kx) * 16 + lo * B_LO);
            if (lo) fbl[m] = v; else fbh[m] = v;
        }
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tl = 0; tl < KYS * KS; ++tl) {
        const int cur = tl & 1;
        const bool more = tl + 1 < KYS * KS;
#pragma unroll
        for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int idx = (pr * NT + n) * MT + m;
                    acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 2 ? al[cur][n] : ah[cur][n], pr == 1 ? bl[cur][m] : bh[cur][m],
                                                                       acc[n][m], 0, 0, 0);
                    if (more && idx < NRD) fetch_one(tl + 1, idx, ah[cur ^ 1], al[cur ^ 1], bh[cur ^ 1], bl[cur ^ 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
        if (more) {
#pragma unroll
            for (int r = NMF; r < NRD; ++r) fetch_one(tl + 1, r, ah[cur ^ 1], al[cur ^ 1], bh[cur ^ 1], bl[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#else
#pragma unroll
    for (int tl = 0; tl < KYS * KS; ++tl) {
        const int cur = tl & 1;
#if SSM_C16_ABL < 3
        if (tl + 1 < KYS * KS) fetch(tl + 1, ah[cur ^ 1], al[cur ^ 1], bh[cur ^ 1], bl[cur ^ 1]);
#endif
#if SSM_C16_SCHED
        __builtin_amdgcn_sched_barrier(0);      // keep the next tap's LDS reads ahead of this tap's MFMAs
#endif
#if SSM_C16_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][n], bh[cur][m], acc[n][m], 0, 0, 0);
                if (C::SPLIT3) {
                    acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][n], bl[cur][m], acc[n][m], 0, 0, 0);
                    acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][n], bh[cur][m], acc[n][m], 0, 0, 0);
                }
            }
#if SSM_C16_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#if SSM_C16_SCHED
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
#endif
}

// bias, LeakyReLU, hi/lo split, stores (HL8 and/or fp32 planes), fused 2x2 mean.
template <class C>
__device__ __forceinline__ void conv16_epilogue(const Conv16Params &p, f32x16 (&acc)[C::NT][C::MT], int nb, int b, int x0, int y0,
                                                int wn, int wy, int wx, int l31, int half) {
    constexpr int BN = C::BN, NT = C::NT, MT = C::MT;
    // ---- epilogue ---------------------------------------------------------------------------
    // register r of lane (l31, half): cout (r&3) + 8*(r>>2) + 4*half of the 32-cout tile, pixel l31
    const int xbase = x0 + wx * (C::MTX * 32) + l31;
    const int ybase = y0 + wy * C::MTY;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int ct = nb * BN + (wn * NT + n) * 32;      // first cout of this 32-cout tile
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int co0 = ct + rq * 8 + half * 4;       // this lane's 4 consecutive couts
            float bias[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[e] = p.bias[co0 + e];
            float v[MT][4];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[n][m][rq * 4 + e] * p.wscale + bias[e];
                    if (p.lrelu) t = t > 0.f ? t : t * p.slope;
                    v[m][e] = t;
                }
#pragma unroll
            for (int my = 0; my < C::MTY; ++my)
#pragma unroll
                for (int mx = 0; mx < C::MTX; ++mx) {
                    const int m = my * C::MTX + mx;
                    const int y = ybase + my, x = xbase + mx * 32;
                    if (y < p.H && x < p.W) {
                        if (p.shuf) {          // sub-pixel form of conv(upsample2x(.)): pixel-shuffle store (Q8 / HL8 destination only)
                            const bool owned = !((p.skipy && (y == 0 || y == p.H - 1)) || (p.skipx && (x == 0 || x == p.W - 1)));
                            if (co0 < p.Cout && owned) {
                                const int par = co0 / p.shuf, cr = co0 - par * p.shuf;
                                const int Y = p.shuf_t ? 2 * x + (par & 1) : 2 * y + (par >> 1);
                                const int X = p.shuf_t ? 2 * y + (par >> 1) : 2 * x + (par & 1);
                                char *d = p.dh + (((long long)b * p.dhsb + (long long)(cr >> 3) * p.dhsg + (long long)Y * p.dhsh + X) * 16);
                                if constexpr (C::Q8)
                                    split_store_q8(d, p.dhsp * 16, p.dhsg * 16, cr >> 3, half, v[m][0], v[m][1], v[m][2], v[m][3]);
                                else if constexpr (C::FAST)
                                    hi_store(d + half * 8, v[m][0], v[m][1], v[m][2], v[m][3]);
                                else
                                    split_store(d + half * 8, p.dhsp * 16, v[m][0], v[m][1], v[m][2], v[m][3]);
                            }
                            continue;
                        }
                        if (p.dh && co0 < p.Cout) {
                            char *d = p.dh + (((long long)b * p.dhsb + (long long)(co0 >> 3) * p.dhsg + (long long)y * p.dhsh + x) * 16);
                            if constexpr (C::Q8)
                                split_store_q8(d, p.dhsp * 16, p.dhsg * 16, co0 >> 3, half, v[m][0], v[m][1], v[m][2], v[m][3]);
                            else if constexpr (C::FAST)
                                hi_store(d + half * 8, v[m][0], v[m][1], v[m][2], v[m][3]);
                            else
                                split_store(d + half * 8, p.dhsp * 16, v[m][0], v[m][1], v[m][2], v[m][3]);
                        }
                        if (p.df) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (co0 + e < p.Cout) p.df[(long long)b * p.dfsb + (long long)(co0 + e) * p.dfsc + (long long)y * p.dfsh + x] = v[m][e];
                        }
                    }
                }
            if (p.ph) {
                if constexpr (C::MTY % 2 == 0) {
#pragma unroll
                    for (int my = 0; my < C::MTY; my += 2)
#pragma unroll
                        for (int mx = 0; mx < C::MTX; ++mx) {
                            float s[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float t = v[my * C::MTX + mx][e] + v[(my + 1) * C::MTX + mx][e];
                                t += __shfl_xor(t, 1);
                                s[e] = t * 0.25f;
                            }
                            const int y = ybase + my, x = xbase + mx * 32;
                            if (!(l31 & 1) && y < p.H && x < p.W && co0 < p.Cout) {
                                char *d = p.ph + (((long long)b * p.phsb + (long long)(co0 >> 3) * p.phsg + (long long)(y >> 1) * p.phsh + (x >> 1)) * 16);
                                if constexpr (C::Q8)
                                    split_store_q8(d, p.phsp * 16, p.phsg * 16, co0 >> 3, half, s[0], s[1], s[2], s[3]);
                                else if constexpr (C::FAST)
                                    hi_store(d + half * 8, s[0], s[1], s[2], s[3]);
                                else
                                    split_store(d + half * 8, p.phsp * 16, s[0], s[1], s[2], s[3]);
                            }
                        }
                }
            }
        }
    }
}

template <class C>
__device__ __forceinline__ void conv16_body(const Conv16Params &p, const int blk, const int nblk) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int KYS = C::KYS, BN = C::BN, PH = C::PH, PW = C::PW, NT = C::NT, MT = C::MT, NW = C::NW, NL = C::NLOAD;
    char *const pbuf0 = lds;
    char *const wbuf0 = lds + C::PBUFS * C::PATCH_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = C::NP ? wid >= NW : true;          // issues DMA
    const bool matrix = wid < NW;                          // runs MFMAs
    const int lw = C::NP ? wid - NW : wid;                 // index among the loading waves
    const int wn = wid % C::WN, wy = (wid / C::WN) % C::WY, wx = wid / (C::WN * C::WY);

    int id = ssm_xcd_tile(blk, nblk);
    const int nb = id % p.NB;
    id /= p.NB;
    const int tx = id % p.tilesX;
    id /= p.tilesX;
    const int ty = id % p.tilesY;
    const int b = id / p.tilesY;
    const int x0 = tx * C::TW, y0 = ty * C::TH;

    const int nchunks = p.Cin / 16;
    const int total_it = nchunks * C::NIT;
    const long long porg = (long long)(y0 - C::PAD) * p.sh + (x0 - C::PAD);     // pixels
    const char *pbase1 = p.src1 + ((long long)b * p.sb1 + porg) * 16;
    const char *pbase2 = p.src2 + ((long long)b * p.sb2 + porg) * 16;
    const char *wbase = p.wpk + (long long)nb * nchunks * C::NIT * C::WST_BYTES;

    // ---- loading side: per-lane source byte offsets of this wave's patch DMA instructions (same for every chunk)
    int poff[C::PM];
    if (loader) {
#pragma unroll
        for (int m = 0; m < C::PM; ++m) {
            int q = (lw + NL * m) * 64 + lane;
            if (q >= C::PATCH_PIECES) q = C::PATCH_PIECES - 1;   // tail lanes re-read the last piece into the padding
            const int pl = q / (PH * PW);
            const int rem = q - pl * (PH * PW);
            const int r = rem / PW;
            const int c = rem - r * PW;
            const int grp = C::FAST ? pl : pl >> 1, part = C::FAST ? 0 : pl & 1;
            poff[m] = (int)(((long long)grp * p.sg + (long long)part * p.sp + (long long)r * p.sh + c) * 16);
        }
    }
    const int woff = (lw * 64 + lane) * 16;

    auto patch_src = [&](int ch) -> const char * {
        const int c0 = ch * 16;
        return (c0 < p.C1) ? pbase1 + (long long)(c0 >> 3) * p.sg * 16 : pbase2 + (long long)((c0 - p.C1) >> 3) * p.sg * 16;
    };
    // slices [jlo, jhi) (of NIT) of the patch of chunk ch -> pbuf[ch&1]
    auto issue_patch = [&](int ch, int jlo, int jhi) {
        const char *ps = patch_src(ch);
        char *pb = pbuf0 + (C::PBUFS == 2 ? (ch & 1) : 0) * C::PATCH_BYTES;
#pragma unroll
        for (int m = 0; m < C::PM; ++m) {
            const int ii = lw + NL * m;
            const int j = m % C::NIT;
            if (ii < C::PNI && j >= jlo && j < jhi) SSM_GLDS16B(ps + poff[m], pb + ii * 1024);
        }
    };
    auto issue_w = [&](int it) {
        const char *ws = wbase + (long long)it * C::WST_BYTES + woff;
        char *wb = wbuf0 + (it & 1) * C::WST_BYTES;
#pragma unroll
        for (int m = 0; m < C::WM; ++m) {
            const int ii = lw + NL * m;
            if (ii < C::WNI) SSM_GLDS16B(ws + m * (NL * 1024), wb + ii * 1024);
        }
    };
    // Per iteration `it` (chunk ch, filter-row stage j), after the barrier that makes stage `it` visible: the loaders
    // start stage it+1 (its buffer's readers were iteration it-1) and slice j of the next chunk's patch (PBUFS == 2: the
    // other buffer, whose readers were chunk ch-1).  PBUFS == 1: the single patch buffer is refilled at the chunk
    // boundary behind an extra barrier; the CU's second workgroup covers that latency.
    auto load_step = [&](int it, int ch, int j) {
#if SSM_C16_ABL < 1
        if (it + 1 < total_it) issue_w(it + 1);
        if (C::PBUFS == 2 && ch + 1 < nchunks) issue_patch(ch + 1, j, j + 1);
#endif
    };

    if constexpr (C::NP > 0) {
        if (!matrix) {                                 // ---- producer waves: DMA only ----
            issue_patch(0, 0, C::NIT);
            issue_w(0);
            for (int it = 0, ch = 0, j = 0; it < total_it; ++it) {
                if (C::PBUFS == 1 && j == 0 && ch > 0) {
                    __syncthreads();                   // every matrix wave is done reading the previous chunk's patch
                    issue_patch(ch, 0, C::NIT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                load_step(it, ch, j);
                if (++j == C::NIT) j = 0, ++ch;
            }
            return;
        }
    }

    // ---- matrix waves ----
    f32x16 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;

    // per-lane operand byte offsets inside a patch buffer / filter stage
    const int bOff = ((half * (C::NPL / 2) * PH + wy * C::MTY) * PW + wx * (C::MTX * 32) + l31) * 16;
    const int aOff = C::Q8 ? (half * BN + wn * (NT * 32) + l31) * 16 : (half * 2 * BN + wn * (NT * 32) + l31) * 16;
    const int bqOff = ((PH + wy * C::MTY) * PW + wx * (C::MTX * 32) + l31) * 16;      // q plane of group 0 (Q8 mode)

    if (C::NP == 0) {
        issue_patch(0, 0, C::NIT);
        issue_w(0);
    }
    for (int it = 0, ch = 0, j = 0; it < total_it; ++it) {
        if (C::PBUFS == 1 && j == 0 && ch > 0) {
            __syncthreads();
            if (C::NP == 0) issue_patch(ch, 0, C::NIT);
        }
        if (C::NP == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (C::NP == 0) load_step(it, ch, j);
        const char *pb = pbuf0 + (C::PBUFS == 2 ? (ch & 1) : 0) * C::PATCH_BYTES + j * (KYS * PW * 16);
        const char *sb = pb + bOff;
        const char *sa = wbuf0 + (it & 1) * C::WST_BYTES + aOff;
        if constexpr (C::Q8)
            conv16_compute_q8<C>(acc, sb, sa, pb + bqOff, sa + KYS * C::KS * 2 * BN * 16, half * 16);
        else
            conv16_compute<C>(acc, sb, sa);
        if (++j == C::NIT) j = 0, ++ch;
    }

    conv16_epilogue<C>(p, acc, nb, b, x0, y0, wn, wy, wx, l31, half);
}

// =====================================================================================================
// Fused  conv3x3( F.upsample(torch.cat([a, b], 1), x2, bilinear) )   (scripts/models/flow_computation.py:
// 244-247 and the four sites like it): the concatenated, upsampled tensor is never written to HBM.
//
// Wave-specialised workgroup: NWM "matrix" waves run exactly the pipeline iteration of conv16_kernel on a
// hi-res activation patch in LDS; NWE "expander" waves (a) issue ALL LDS-DMA of the workgroup - the LOW-res raw
// patch [4 planes][TH/2+2][TW/2+2] of the next-but-one chunk and the filter stages - and (b) bilinearly expand
// the next chunk's raw patch into the other hi-res patch buffer (hi+lo -> fp32, ATen's half-pixel rule with
// edge clamping, zero outside the image = the convolution's zero padding, re-split to hi/lo).  VALU expansion
// and MFMA run on different pipes of the same CU, so the expansion hides under the matrix work; the conv
// reads 4x fewer activation bytes than from a materialised upsampled tensor.  One barrier per iteration.
template <int KYS_, int NT_, int WN_, int MTY_, int MTX_, int WY_, int WX_, int NWE_, int MODE_>
struct CfgUps {
    static constexpr int KS = 3, KYS = KYS_, NT = NT_, WN = WN_, MTY = MTY_, MTX = MTX_, WY = WY_, WX = WX_;
    static constexpr bool SPLIT3 = MODE_ == 1, Q8 = MODE_ == 2, FAST = MODE_ == 0;
    static constexpr int NPL = FAST ? 2 : 4;          // planes of a 16-channel chunk (mode FAST: the two groups' hi planes only, see Cfg16)
    static constexpr int NQ = 1;
    static constexpr int PAD = 1;
    static constexpr int NWM = WN * WY * WX, NWE = NWE_, NTHREADS = 64 * (NWM + NWE);
    static constexpr int BN = 32 * NT * WN, TH = MTY * WY, TW = 32 * MTX * WX, MT = MTY * MTX;
    static constexpr int PH = TH + 2, PW = TW + 2;                 // hi-res patch
    static constexpr int RH = TH / 2 + 2, RW = TW / 2 + 2;         // low-res raw patch
    static constexpr int NIT = KS / KYS;
    static constexpr int PATCH_BYTES = (NPL * PH * PW * 16 + 1023) / 1024 * 1024;
    static constexpr int RAW_PIECES = NPL * RH * RW;
    static constexpr int RNI = (RAW_PIECES + 63) / 64;
    static constexpr int RAW_BYTES = RNI * 1024;
    static constexpr int WST_PIECES = Q8 ? KYS * KS * 2 * BN + KYS * NQ * 8 * BN : KYS * KS * 4 * BN;
    static constexpr int WNI = WST_PIECES / 64;
    static constexpr int WST_BYTES = WST_PIECES * 16;
    static constexpr int LDS_BYTES = 2 * PATCH_BYTES + 2 * RAW_BYTES + 2 * WST_BYTES;
    static constexpr int RM = (RNI + NWE - 1) / NWE, WM = (WNI + NWE - 1) / NWE;
    static constexpr int PH2 = PH / 2, PW2 = PW / 2, NUNITS = 2 * PH2 * PW2;     // 2x2 output blocks x 2 channel groups
    static_assert(TH % 2 == 0 && TW % 2 == 0, "even tile");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(WST_PIECES % 64 == 0, "filter stage is whole DMA instructions");
};

__device__ __forceinline__ void h8_to_f32(const char *hi_ptr, int lo_off, float (&o)[8]) {
    const h8 hi = *reinterpret_cast<const h8 *>(hi_ptr), lo = *reinterpret_cast<const h8 *>(hi_ptr + lo_off);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)hi[e] + (float)lo[e];
}

__device__ __forceinline__ void f32_to_h8(char *hi_ptr, int lo_off, const float (&v)[8], bool zero) {
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = zero ? 0.f : v[e];
        hi[e] = (_Float16)x;
        lo[e] = (_Float16)(x - (float)hi[e]);
    }
    *reinterpret_cast<h8 *>(hi_ptr) = hi;
    *reinterpret_cast<h8 *>(hi_ptr + lo_off) = lo;
}

template <class C>
__global__ __launch_bounds__(C::NTHREADS, 1) void conv16_ups_kernel(const Conv16Params p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int KYS = C::KYS, BN = C::BN, PH = C::PH, PW = C::PW, NT = C::NT, MT = C::MT, RH = C::RH, RW = C::RW;
    char *const pbuf0 = lds;
    char *const rbuf0 = lds + 2 * C::PATCH_BYTES;
    char *const wbuf0 = rbuf0 + 2 * C::RAW_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_mfma = wid < C::NWM;
    const int wn = wid % C::WN, wy = (wid / C::WN) % C::WY, wx = (wid / (C::WN * C::WY)) % C::WX;   // matrix waves
    const int ew = wid - C::NWM, etid = tid - C::NWM * 64;                                            // expander waves

    int id = ssm_xcd_tile(blockIdx.x, gridDim.x);
    const int nb = id % p.NB;
    id /= p.NB;
    const int tx = id % p.tilesX;
    id /= p.tilesX;
    const int ty = id % p.tilesY;
    const int b = id / p.tilesY;
    const int x0 = tx * C::TW, y0 = ty * C::TH;          // hi-res tile origin (even)
    const int ly0 = y0 / 2 - 1, lx0 = x0 / 2 - 1;        // low-res origin of the raw patch (may be -1: zero frame)
    const int lh = p.H / 2, lw = p.W / 2;                // low-res image size

    const int nchunks = p.Cin / 16;
    const int total_it = nchunks * C::NIT;

    // ---- expander-side state ---------------------------------------------------------------------------
    const long long rorg = (long long)ly0 * p.sh + lx0;                       // pixels, relative to the interior origin
    const char *rbase1 = p.src1 + ((long long)b * p.sb1 + rorg) * 16;
    const char *rbase2 = p.src2 + ((long long)b * p.sb2 + rorg) * 16;
    const char *wbase = p.wpk + (long long)nb * nchunks * C::NIT * C::WST_BYTES;
    int roff[C::RM];
#pragma unroll
    for (int m = 0; m < C::RM; ++m) {
        int q = ((ew < 0 ? 0 : ew) + C::NWE * m) * 64 + lane;
        if (q >= C::RAW_PIECES) q = C::RAW_PIECES - 1;
        const int pl = q / (RH * RW);
        const int rem = q - pl * (RH * RW);
        const int r = rem / RW;
        const int c = rem - r * RW;
        const int grp = C::FAST ? pl : pl >> 1, part = C::FAST ? 0 : pl & 1;
        roff[m] = (int)(((long long)grp * p.sg + (long long)part * p.sp + (long long)r * p.sh + c) * 16);
    }
    auto issue_raw = [&](int ch) {
        const int c0 = ch * 16;
        const char *rs = (c0 < p.C1) ? rbase1 + (long long)(c0 >> 3) * p.sg * 16 : rbase2 + (long long)((c0 - p.C1) >> 3) * p.sg * 16;
        char *rb = rbuf0 + (ch & 1) * C::RAW_BYTES;
#pragma unroll
        for (int m = 0; m < C::RM; ++m) {
            const int ii = ew + C::NWE * m;
            if (ii < C::RNI) SSM_GLDS16B(rs + roff[m], rb + ii * 1024);
        }
    };
    auto issue_w = [&](int it) {
        const char *ws = wbase + (long long)it * C::WST_BYTES + (ew * 64 + lane) * 16;
        char *wb = wbuf0 + (it & 1) * C::WST_BYTES;
#pragma unroll
        for (int m = 0; m < C::WM; ++m) {
            const int ii = ew + C::NWE * m;
            if (ii < C::WNI) SSM_GLDS16B(ws + m * (C::NWE * 1024), wb + ii * 1024);
        }
    };
    // Bilinear x2 of chunk ch: raw[ch&1] -> patch[ch&1].  Work unit = one 2x2 output block of 4 channels (half a
    // group), so every expander lane is busy even on the small tiles; the units are processed in rounds of
    // NWE*64, round r in pipeline iteration r % nsl (all rounds when nsl == 1).
    auto expand = [&](int ch, int j, int nsl) {
        const char *rb = rbuf0 + (ch & 1) * C::RAW_BYTES;
        char *pb = pbuf0 + (ch & 1) * C::PATCH_BYTES;
        constexpr int R_LO = RH * RW * 16, P_LO = PH * PW * 16;
        constexpr int NHU = 2 * C::NUNITS;                          // half-units
        constexpr int ROUNDS = (NHU + C::NWE * 64 - 1) / (C::NWE * 64);
#pragma unroll 1
        for (int r = 0; r < ROUNDS; ++r) {
            if (r % nsl != j) continue;
            const int u = r * (C::NWE * 64) + etid;
            if (u >= NHU) continue;
            const int hs = u & 1;                                   // which 4 of the group's 8 channels
            const int uu = u >> 1;
            const int g = uu / (C::PH2 * C::PW2);
            const int rem = uu - g * (C::PH2 * C::PW2);
            const int by = rem / C::PW2, bx = rem - by * C::PW2;
            // low-res rows i, i+1 and columns jx, jx+1 (image coordinates), clamped like ATen clamps its source index
            const int i = ly0 + by, jx = lx0 + bx;
            const int r0 = min(max(i, 0), lh - 1) - ly0, r1 = min(max(i + 1, 0), lh - 1) - ly0;
            const int c0 = min(max(jx, 0), lw - 1) - lx0, c1 = min(max(jx + 1, 0), lw - 1) - lx0;
            const char *rg = rb + (g * (C::NPL / 2) * RH * RW) * 16 + hs * 8;
            auto ld = [&](int rr, int cc, float (&o)[4]) {
                const h4 hi = *reinterpret_cast<const h4 *>(rg + (rr * RW + cc) * 16);
                if constexpr (C::Q8) {      // lo bytes of both groups sit in the odd group's second plane (raw plane 3)
                    const int w = *reinterpret_cast<const int *>(rb + (3 * RH * RW + rr * RW + cc) * 16 + g * 8 + hs * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (float)hi[e] + fp8_byte(w, e) * (1.0f / 2048.0f);
                } else if constexpr (C::FAST) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (float)hi[e];
                } else {
                    const h4 lo = *reinterpret_cast<const h4 *>(rg + (rr * RW + cc) * 16 + R_LO);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (float)hi[e] + (float)lo[e];
                }
            };
            // horizontal pass first (rows i and i+1); X = 2j+1: cols (j,j+1) x (.75,.25), X = 2j+2: (.25,.75)
            float t0[4], t1[4], u0[4], u1[4];
            {
                float a[4], bb[4];
                ld(r0, c0, a);
                ld(r0, c1, bb);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    t0[e] = 0.75f * a[e] + 0.25f * bb[e];
                    t1[e] = 0.25f * a[e] + 0.75f * bb[e];
                }
                ld(r1, c0, a);
                ld(r1, c1, bb);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u0[e] = 0.75f * a[e] + 0.25f * bb[e];
                    u1[e] = 0.25f * a[e] + 0.75f * bb[e];
                }
            }
            // hi-res coordinates of the block; positions outside the image are the convolution's zero padding
            const int Y = 2 * i + 1, X = 2 * jx + 1;
            const bool zy0 = Y < 0 || Y >= p.H, zy1 = Y + 1 < 0 || Y + 1 >= p.H;
            const bool zx0 = X < 0 || X >= p.W, zx1 = X + 1 < 0 || X + 1 >= p.W;
            char *pg = pb + (g * (C::NPL / 2) * PH * PW) * 16 + ((2 * by) * PW + 2 * bx) * 16 + hs * 8;
            auto st = [&](char *d, float wa, float wb, const float (&ta)[4], const float (&ua)[4], bool zero) {
                h4 hi, lo;
                float x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[e] = zero ? 0.f : wa * ta[e] + wb * ua[e];
                    hi[e] = (_Float16)x[e];
                    lo[e] = (_Float16)(x[e] - (float)hi[e]);
                }
                *reinterpret_cast<h4 *>(d) = hi;
                if constexpr (C::Q8) {      // d = group g's hi plane + pixel + hs*8: the fp8 planes are patch planes 1 and 3
                    char *q = d - (g * 2 * PH * PW) * 16 - hs * 8 + P_LO + g * 8 + hs * 4;
                    *reinterpret_cast<int *>(q) = pack4_fp8(x[0], x[1], x[2], x[3]);
                    *reinterpret_cast<int *>(q + 2 * P_LO) = pack4_fp8((x[0] - (float)hi[0]) * 2048.0f, (x[1] - (float)hi[1]) * 2048.0f,
                                                                      (x[2] - (float)hi[2]) * 2048.0f, (x[3] - (float)hi[3]) * 2048.0f);
                } else if constexpr (!C::FAST) {
                    *reinterpret_cast<h4 *>(d + P_LO) = lo;
                }
            };
            // vertical pass: Y = 2i+1: rows (i, i+1) x (.75, .25); Y = 2i+2: (.25, .75)
            st(pg, 0.75f, 0.25f, t0, u0, zy0 || zx0);
            st(pg + 16, 0.75f, 0.25f, t1, u1, zy0 || zx1);
            st(pg + PW * 16, 0.25f, 0.75f, t0, u0, zy1 || zx0);
            st(pg + PW * 16 + 16, 0.25f, 0.75f, t1, u1, zy1 || zx1);
        }
    };

    // ---- matrix-side operand bases ---------------------------------------------------------------------
    const int bOff = ((half * (C::NPL / 2) * PH + wy * C::MTY) * PW + wx * (C::MTX * 32) + l31) * 16;
    const int aOff = C::Q8 ? (half * BN + wn * (NT * 32) + l31) * 16 : (half * 2 * BN + wn * (NT * 32) + l31) * 16;
    const int bqOff = ((PH + wy * C::MTY) * PW + wx * (C::MTX * 32) + l31) * 16;

    // ---- prologue: raw(0), raw(1), filter stage 0; expand chunk 0 -----------------------------------------
    if (!is_mfma) {
        issue_raw(0);
        if (nchunks > 1) issue_raw(1);
        issue_w(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (!is_mfma) expand(0, 0, 1);
    __syncthreads();

    // Two role-specific loops with the same barrier count (registers: max of the roles, not their sum).
    if (is_mfma) {
        f32x16 acc[NT][MT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;
        for (int it = 0; it < total_it; ++it) {
            const int ch = it / C::NIT, j = it - ch * C::NIT;
            const char *pb = pbuf0 + (ch & 1) * C::PATCH_BYTES + j * (KYS * PW * 16);
            const char *sa = wbuf0 + (it & 1) * C::WST_BYTES + aOff;
            if constexpr (C::Q8)
                conv16_compute_q8<C>(acc, pb + bOff, sa, pb + bqOff, sa + KYS * C::KS * 2 * BN * 16, half * 16);
            else
                conv16_compute<C>(acc, pb + bOff, sa);
            __syncthreads();
        }
        conv16_epilogue<C>(p, acc, nb, b, x0, y0, wn, wy, wx, l31, half);
    } else {
        for (int it = 0; it < total_it; ++it) {
            const int ch = it / C::NIT, j = it - ch * C::NIT;
            // Buffers: raw(c) -> raw[c&1], expanded into patch[c&1], consumed by the matrix waves during chunk c.
            // Every wave has passed the barrier that ended iteration it-1, so
            //  - filter stage (it+1)&1 is free (its readers were iteration it-1),
            //  - at j == 0: raw[ch&1] is free (chunk ch was expanded while chunk ch-1 ran, or in the prologue) and
            //    patch[(ch+1)&1] is free (its readers were chunk ch-1),
            //  - raw(ch+1) has landed (issued one chunk ago, waited for below).
            if (it + 1 < total_it) issue_w(it + 1);
            if (j == 0 && ch + 2 < nchunks) issue_raw(ch + 2);
            if (ch + 1 < nchunks) expand(ch + 1, j, C::NIT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
}

template <class C>
__global__ __launch_bounds__(C::NTHREADS_ALL, (C::BLOCKS_PER_CU * (C::NW + C::NP) + 3) / 4) void conv16_kernel(const Conv16Params p) {
    conv16_body<C>(p, blockIdx.x, gridDim.x);
}

template <class C>
__global__ __launch_bounds__(C::NTHREADS_ALL, (C::BLOCKS_PER_CU * (C::NW + C::NP) + 3) / 4) void conv16_multi_kernel(const Conv16Multi m) {
    int i = 0;
    for (int j = 1; j < m.n; ++j)
        if ((int)blockIdx.x >= m.start[j]) i = j;
    const Conv16Params p = m.table[i];
    conv16_body<C>(p, (int)blockIdx.x - m.start[i], m.start[i + 1] - m.start[i]);
}

// ---- tile configurations --------------------------------------------------------------------------
//                     KS KYS NT WN MTY MTX WY WX PBUFS        waves  BN   TH  TW    LDS     workgroups/CU
template <int S> using C16K7 = Cfg16<7, 1, 1, 1, 2, 1, 4, 1, 1, S>;       //  4   32    8  32    63 KB   2
template <int S> using C16K5 = Cfg16<5, 1, 2, 1, 2, 1, 4, 2, 2, S>;       //  8   64    8  64   145 KB   1
template <int S> using C16K3N32 = Cfg16<3, (S == 2 ? 1 : 3), 1, 1, 2, 2, 4, 1, 1, S>;    //  4   32    8  64    80 KB   2   (short K)
template <int S> using C16K3N32D = Cfg16<3, (S == 2 ? 1 : 3), 1, 1, 2, 1, 4, 2, 2, S>;   //  8   32    8  64   121 KB   1   (Cin >= 64)
// narrow maps (W <= 48): all three filter rows per stage also in the Q8 form - a third of the barrier-separated iterations per
// workgroup, which is what a layer with fewer workgroups than CUs pays for (0.07 ms per 512-channel layer at 22x22 regardless of FLOPs)
template <int S> using C16K3N32V = Cfg16<3, 3, 1, 1, 2, 2, 4, 1, 1, S>;      //  4   32    8  64   (Cin < 128)
template <int S> using C16K3N32W = Cfg16<3, 3, 1, 1, 2, 1, 4, 2, 2, S>;      //  8   32    8  64   (Cin >= 128)
template <int S> using C16K3N64 = Cfg16<3, (S == 2 ? 1 : 3), 2, 1, 2, 1, 4, 2, 2, S>;    //  8   64    8  64   158 KB   1
template <int S> using C16K3N128 = Cfg16<3, 1, 2, 2, 2, 1, 2, 2, 2, S>;   //  8  128    4  64    99 KB   1   //  8  128    4  64    99 KB   1
template <int S> using C16K3N128S = Cfg16<3, 1, 2, 2, 2, 1, 2, 1, (S == 2 ? 1 : 2), S>;   // Q8: single patch buffer, 2 workgroups per CU  //  4  128    4  32    83 KB   1

//                        KYS NT WN MTY MTX WY WX NWE           matrix+expander waves  BN  TH  TW   LDS
template <int S> using U3N32 = CfgUps<3, 1, 1, 2, 1, 4, 2, 8, S>;    //  8 + 8   32   8  64  146 KB   conv11a (expansion-heaviest: 8 expander waves, measured 1.90 -> 1.72 ms)
template <int S> using U3N64 = CfgUps<1, 2, 1, 2, 1, 4, 2, 4, S>;    //  8 + 4   64   8  64  134 KB   conv10a
template <int S> using U3N128 = CfgUps<1, 2, 2, 2, 1, 2, 2, 4, S>;   //  8 + 4  128   4  64  116 KB   conv9a
template <int S> using U3N128S = CfgUps<1, 2, 2, 2, 1, 2, 1, 4, S>;  //  4 + 4  128   4  32   84 KB   conv7a, conv8a

enum Kind16 { H7 = 0, H5, H3N32, H3N32D, H3N64, H3N128, H3N128S, H3N32V, H3N32W };

int pick16(int k, int Cout, int W, int Cin = 0) {
    if (k == 7) return H7;
    if (k == 5) return H5;
    if (k != 3) return -1;
    // Narrow maps (the 1/16 and 1/32 levels of 352-pixel crops, the 1/32 level of 720p): a 128-cout tile leaves a few dozen
    // workgroups, each streaming a quarter of the layer's filter through one CU (0.10 ms per layer regardless of its FLOPs);
    // 32-cout tiles give 4x the workgroups and a quarter of the filter bytes each.
    // (measured: training step 140 -> 147 samples/s, 720p inference unchanged; a threshold of 96 costs the 720p 1/16 level 4 %)
    if (W <= 48) return Cin >= 128 ? H3N32W : H3N32V;                   // same BN / KYS: the packed filter does not depend on Cin
    if (Cout <= 32) return Cin >= 128 ? H3N32D : H3N32;
    if (Cout <= 64) return H3N64;
    const int w64 = (W + 63) / 64 * 64, w32 = (W + 31) / 32 * 32;
    return w32 < w64 ? H3N128S : H3N128;
}

// Kernels here use up to 158 KB of dynamic LDS: opt in once per (kernel, device); re-entrant.
bool reserve_lds(const void *kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, dev})) return true;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        ssm::set_error("conv16: cannot reserve %d bytes of LDS: %s", bytes, hipGetErrorString(e));
        return false;
    }
    done.insert({kernel, dev});
    return true;
}

template <class C>
void dims16(int *BN, int *KYS) {
    *BN = C::BN;
    *KYS = C::KYS;
}

template <class C>
int launch16(Conv16Params &p, int B, hipStream_t st) {
    p.tilesX = (p.W + C::TW - 1) / C::TW;
    p.tilesY = (p.H + C::TH - 1) / C::TH;
    p.NB = (p.Cout + C::BN - 1) / C::BN;
    const long long blocks = (long long)p.tilesX * p.tilesY * p.NB * B;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        ssm::set_error("conv16: grid of %lld workgroups out of range", blocks);
        return SSM_E_ARG;
    }
    if (!reserve_lds((const void *)conv16_kernel<C>, C::LDS_BYTES)) return SSM_E_LAUNCH;
    SSM_LAUNCH(conv16_kernel<C>, dim3((unsigned)blocks), dim3(C::NTHREADS_ALL), C::LDS_BYTES, st, p);
    return ssm::check_launch("ssm_conv2d_hl8_fwd");
}

template <class C>
int launch16_ups(Conv16Params &p, int B, hipStream_t st) {
    p.tilesX = (p.W + C::TW - 1) / C::TW;
    p.tilesY = (p.H + C::TH - 1) / C::TH;
    p.NB = (p.Cout + C::BN - 1) / C::BN;
    const long long blocks = (long long)p.tilesX * p.tilesY * p.NB * B;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        ssm::set_error("conv16_ups: grid of %lld workgroups out of range", blocks);
        return SSM_E_ARG;
    }
    if (!reserve_lds((const void *)conv16_ups_kernel<C>, C::LDS_BYTES)) return SSM_E_LAUNCH;
    SSM_LAUNCH(conv16_ups_kernel<C>, dim3((unsigned)blocks), dim3(C::NTHREADS), C::LDS_BYTES, st, p);
    return ssm::check_launch("ssm_conv2d_ups_hl8_fwd");
}

template <int S>
int dispatch16_ups(Conv16Params &p, int B, hipStream_t st) {
    switch (pick16(3, p.Cout, p.W, p.Cin)) {
        case H3N32:
        case H3N32D:
        case H3N32V:
        case H3N32W: return launch16_ups<U3N32<S>>(p, B, st);
        case H3N64: return launch16_ups<U3N64<S>>(p, B, st);
        case H3N128: return launch16_ups<U3N128<S>>(p, B, st);
        case H3N128S: return launch16_ups<U3N128S<S>>(p, B, st);
    }
    return SSM_E_UNSUPPORTED;
}

template <int S>
int dispatch16(Conv16Params &p, int B, int k, hipStream_t st) {
    switch (pick16(k, p.Cout, p.W, p.Cin)) {
        case H7: return launch16<C16K7<S>>(p, B, st);
        case H5: return launch16<C16K5<S>>(p, B, st);
        case H3N32: return launch16<C16K3N32<S>>(p, B, st);
        case H3N32D: return launch16<C16K3N32D<S>>(p, B, st);
        case H3N32V: return launch16<C16K3N32V<S>>(p, B, st);
        case H3N32W: return launch16<C16K3N32W<S>>(p, B, st);
        case H3N64: return launch16<C16K3N64<S>>(p, B, st);
        case H3N128: return launch16<C16K3N128<S>>(p, B, st);
        case H3N128S: return launch16<C16K3N128S<S>>(p, B, st);
    }
    return SSM_E_UNSUPPORTED;
}

// OIHW fp32 -> [nb][chunk16][stage j][tap in stage][h][part][BN][8] fp16, values multiplied by `scale` (2^s)
__global__ void pack16_kernel(const float *__restrict__ w, const float *__restrict__ bias, _Float16 *__restrict__ wp,
                              float *__restrict__ bp, int Cout, int Cin, int CinP, int KS, int KYS, int BN, float scale,
                              long long total, int nbias) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        long long r = i;
        const int e = (int)(r % 8); r /= 8;
        const int n = (int)(r % BN); r /= BN;
        const int part = (int)(r % 2); r /= 2;
        const int h = (int)(r % 2); r /= 2;
        const int tl = (int)(r % (KYS * KS)); r /= (KYS * KS);
        const int j = (int)(r % (KS / KYS)); r /= (KS / KYS);
        const int ch = (int)(r % (CinP / 16));
        const int nb = (int)(r / (CinP / 16));
        const int co = nb * BN + n, ci = ch * 16 + h * 8 + e;
        const int ky = j * KYS + tl / KS, kx = tl % KS;
        float v = 0.f;
        if (co < Cout && ci < Cin) v = w[(((long long)co * Cin + ci) * KS + ky) * KS + kx] * scale;
        const _Float16 hi = (_Float16)v;
        wp[i] = part ? (_Float16)(v - (float)hi) : hi;
    }
    if (i < nbias) bp[i] = (i < Cout) ? bias[i] : 0.f;
}

// Q8 filter packing: per (cout block, 16-channel chunk, stage of KYS filter rows) one LDS image of the stage:
//   [tap][h][BN][8] fp16(w*scale)   then   [row][quad][operand][piece][half][BN][16] fp8, where lane half `half` of quad j covers
//   taps kx = 4j + 2*half + piece (zero bytes beyond the row), byte = group*8 + e, operand 0 = fp8(w*scale), 1 = fp8(lo*2^11).
// One thread = one filter row of one (cout, 16-channel chunk): it reads the chunk's 16 x KS weights of the row (the 16 x KS^2 block of
// a cout is contiguous in OIHW, so the KS threads of a block of rows use every byte of the cache lines they touch) and emits the
// row's fp16 and fp8 pieces.
// transposed: `w` is the OIHW filter of the FORWARD convolution ([Cin][Cout][KS][KS] seen from here) and the packed filter is its
// data-gradient counterpart W'[co][ci][ky][kx] = w[ci][co][KS-1-ky][KS-1-kx] (no materialised permute/flip).
template <int KS>
__device__ __forceinline__ void pack16q_row(const float *__restrict__ w, char *__restrict__ wp, int Cout, int Cin, int CinP, int KYS, int BN,
                                            float scale, long long stage_bytes, long long qbase, long long i, int transposed) {
    constexpr int NQ = (KS + 3) / 4;
    const int ky = (int)(i % KS);
    const int n = (int)((i / KS) % BN);
    const int ch = (int)((i / KS / BN) % (CinP / 16));
    const int nb = (int)(i / KS / BN / (CinP / 16));
    const int co = nb * BN + n;
    {
        const int j = ky / KYS, ry = ky - j * KYS;
        char *st = wp + (((long long)nb * (CinP / 16) + ch) * (KS / KYS) + j) * stage_bytes;
        float v[16][KS];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ci = ch * 16 + e;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx)
                v[e][kx] = (co < Cout && ci < Cin)
                               ? (transposed ? w[(((long long)ci * Cout + co) * KS + (KS - 1 - ky)) * KS + (KS - 1 - kx)]
                                             : w[(((long long)co * Cin + ci) * KS + ky) * KS + kx]) * scale
                               : 0.f;
        }
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                h8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (_Float16)v[h * 8 + e][kx];
                *reinterpret_cast<h8 *>(st + ((long long)((ry * KS + kx) * 2 + h) * BN + n) * 16) = hv;
            }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int op = 0; op < 2; ++op)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int kx = 4 * q + 2 * half + pc;
                        float x[16];
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float t = kx < KS ? v[e][kx < KS ? kx : 0] : 0.f;
                            x[e] = op ? (t - (float)(_Float16)t) * 2048.0f : t;
                        }
                        const i32x4 word = {pack4_fp8(x[0], x[1], x[2], x[3]), pack4_fp8(x[4], x[5], x[6], x[7]),
                                            pack4_fp8(x[8], x[9], x[10], x[11]), pack4_fp8(x[12], x[13], x[14], x[15])};
                        *reinterpret_cast<i32x4 *>(st + qbase + ((((((long long)(ry * NQ + q) * 2 + op) * 2 + pc) * 2 + half) * BN + n) * 16)) = word;
                    }
    }
}

template <int KS>
__global__ void pack16q_kernel(const float *__restrict__ w, char *__restrict__ wp, int Cout, int Cin, int CinP, int KYS, int BN, float scale,
                               long long stage_bytes, long long qbase, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    pack16q_row<KS>(w, wp, Cout, Cin, CinP, KYS, BN, scale, stage_bytes, qbase, i, 0);
}

// Every filter of a U-Net in ONE launch (training repacks all of them after each optimizer step: 2 launches per layer and a
// torch permute/flip/copy per data-gradient filter made the step launch-bound).  jobs: device array; block b belongs to the job with
// block_start <= b < block_start + row_blocks + bias_blocks.
__global__ __launch_bounds__(64) void pack16q_batch_kernel(const ssm_pack16q_job *__restrict__ jobs, int n_jobs) {
    int lo = 0, hi = n_jobs - 1;
    const int blk = blockIdx.x;
    while (lo < hi) {                                   // last job whose block_start <= blk
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block_start <= blk) lo = mid;
        else hi = mid - 1;
    }
    const ssm_pack16q_job jb = jobs[lo];
    const int rel = blk - jb.block_start;
    if (rel >= jb.row_blocks) {                         // packed bias (zeros for a data-gradient filter)
        const int i = (rel - jb.row_blocks) * 64 + threadIdx.x;
        const int nbias = (jb.Cout + jb.BN - 1) / jb.BN * jb.BN;
        if (i < nbias) jb.bp[i] = (i < jb.Cout && jb.bias) ? jb.bias[i] : 0.f;
        return;
    }
    const long long i = (long long)rel * 64 + threadIdx.x;
    const long long total = (long long)((jb.Cout + jb.BN - 1) / jb.BN) * (jb.CinP / 16) * jb.BN * jb.k;
    if (i >= total) return;
    const long long sb = (long long)(jb.KYS * jb.k * 2 + jb.KYS * ((jb.k + 3) / 4) * 8) * jb.BN * 16;
    const long long qbase = (long long)jb.KYS * jb.k * 2 * jb.BN * 16;
    switch (jb.k) {
        case 3: pack16q_row<3>(jb.w, (char *)jb.wp, jb.Cout, jb.Cin, jb.CinP, jb.KYS, jb.BN, jb.scale, sb, qbase, i, jb.transposed); break;
        case 5: pack16q_row<5>(jb.w, (char *)jb.wp, jb.Cout, jb.Cin, jb.CinP, jb.KYS, jb.BN, jb.scale, sb, qbase, i, jb.transposed); break;
        case 7: pack16q_row<7>(jb.w, (char *)jb.wp, jb.Cout, jb.Cin, jb.CinP, jb.KYS, jb.BN, jb.scale, sb, qbase, i, jb.transposed); break;
        default: break;
    }
}

// fp32 view [B,C,H,W] -> Q8 form of the HL8 geometry, and back (x ~= hi + fp8_lo * 2^-11)
__global__ __launch_bounds__(256) void to_hq8_kernel(ssm_view src, ssm_hview dst, int C, int G, int H, int W) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z / G, g = blockIdx.z - b * G;
    if (x >= W || y >= H) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = g * 8 + e;
        v[e] = c < C ? src.ptr[(long long)b * src.sb + (long long)c * src.sc + (long long)y * src.sh + x] : 0.f;
    }
    char *d = (char *)dst.ptr + ((long long)b * dst.sb + (long long)g * dst.sg + (long long)y * dst.sh + x) * 16;
    split_store_q8(d, dst.sp * 16, dst.sg * 16, g, 0, v[0], v[1], v[2], v[3]);
    split_store_q8(d, dst.sp * 16, dst.sg * 16, g, 1, v[4], v[5], v[6], v[7]);
}

__global__ __launch_bounds__(256) void from_hq8_kernel(ssm_hview src, ssm_view dst, int C, int G, int H, int W) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z / G, g = blockIdx.z - b * G;
    if (x >= W || y >= H) return;
    const char *s = (const char *)src.ptr + ((long long)b * src.sb + (long long)g * src.sg + (long long)y * src.sh + x) * 16;
    const h8 hi = *reinterpret_cast<const h8 *>(s);
    const int odd = g & 1;         // the lo bytes of both groups of a pair live in the ODD group's second plane
    const int *ql = reinterpret_cast<const int *>(s + (1 - odd) * src.sg * 16 + src.sp * 16 + odd * 8);
    const int q[2] = {ql[0], ql[1]};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = g * 8 + e;
        const float lo = fp8_byte(q[e >> 2], e & 3);
        if (c < C) dst.ptr[(long long)b * dst.sb + (long long)c * dst.sc + (long long)y * dst.sh + x] = (float)hi[e] + lo * (1.0f / 2048.0f);
    }
}

// fp32 view [B,C,H,W] -> HL8 (C padded with zero channels up to 8*G), and back
__global__ __launch_bounds__(256) void to_hl8_kernel(ssm_view src, ssm_hview dst, int C, int G, int H, int W) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z / G, g = blockIdx.z - b * G;
    if (x >= W || y >= H) return;
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = g * 8 + e;
        const float v = c < C ? src.ptr[(long long)b * src.sb + (long long)c * src.sc + (long long)y * src.sh + x] : 0.f;
        hi[e] = (_Float16)v;
        lo[e] = (_Float16)(v - (float)hi[e]);
    }
    char *d = (char *)dst.ptr + ((long long)b * dst.sb + (long long)g * dst.sg + (long long)y * dst.sh + x) * 16;
    *reinterpret_cast<h8 *>(d) = hi;
    *reinterpret_cast<h8 *>(d + dst.sp * 16) = lo;
}

__global__ __launch_bounds__(256) void from_hl8_kernel(ssm_hview src, ssm_view dst, int C, int G, int H, int W) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z / G, g = blockIdx.z - b * G;
    if (x >= W || y >= H) return;
    const char *s = (const char *)src.ptr + ((long long)b * src.sb + (long long)g * src.sg + (long long)y * src.sh + x) * 16;
    const h8 hi = *reinterpret_cast<const h8 *>(s), lo = *reinterpret_cast<const h8 *>(s + src.sp * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = g * 8 + e;
        if (c < C) dst.ptr[(long long)b * dst.sb + (long long)c * dst.sc + (long long)y * dst.sh + x] = (float)hi[e] + (float)lo[e];
    }
}

}  // namespace

extern "C" int ssm_conv16_config(int k, int Cout, int W, int *BN, int *KYS) {
    switch (pick16(k, Cout, W)) {
        case H7: dims16<C16K7<1>>(BN, KYS); break;
        case H5: dims16<C16K5<1>>(BN, KYS); break;
        case H3N32:
        case H3N32D: dims16<C16K3N32<1>>(BN, KYS); break;
        case H3N32V:
        case H3N32W: dims16<C16K3N32W<1>>(BN, KYS); break;
        case H3N64: dims16<C16K3N64<1>>(BN, KYS); break;
        case H3N128: dims16<C16K3N128<1>>(BN, KYS); break;
        case H3N128S: dims16<C16K3N128S<1>>(BN, KYS); break;
        default:
            ssm::set_error("conv16: kernel size %d unsupported (3, 5, 7 are)", k);
            return SSM_E_UNSUPPORTED;
    }
    return SSM_OK;
}

extern "C" size_t ssm_packed16_weight_halves(int Cout, int CinP, int k, int BN) {
    return (size_t)((Cout + BN - 1) / BN) * (size_t)(CinP / 16) * k * k * 4 * BN * 8;
}

extern "C" int ssm_pack16_weights(const float *w, const float *bias, void *wp, float *bp, int Cout, int Cin, int CinP,
                                  int k, int BN, int KYS, float scale, void *stream) {
    SSM_REQUIRE(w && bias && wp && bp, "pack16: null pointer");
    SSM_REQUIRE(Cout > 0 && Cin > 0 && CinP >= Cin && CinP % 16 == 0 && BN % 32 == 0 && k % KYS == 0, "pack16: bad sizes");
    const long long total = (long long)ssm_packed16_weight_halves(Cout, CinP, k, BN);
    const int nbias = (int)ssm_packed_bias_floats(Cout, BN);
    const long long n = total > nbias ? total : nbias;
    SSM_LAUNCH(pack16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, bias,
                       (_Float16 *)wp, bp, Cout, Cin, CinP, k, KYS, BN, scale, total, nbias);
    return ssm::check_launch("ssm_pack16_weights");
}

extern "C" int ssm_hl8_from_f32(ssm_view src, ssm_hview dst, int B, int C, int G, int H, int W, void *stream) {
    SSM_REQUIRE(src.ptr && dst.ptr && B > 0 && C > 0 && G * 8 >= C && H > 0 && W > 0, "hl8_from_f32: bad arguments");
    SSM_REQUIRE((long long)B * G <= 65535, "hl8_from_f32: B*G too large");
    SSM_LAUNCH(to_hl8_kernel, dim3((W + 63) / 64, (H + 3) / 4, B * G), dim3(64, 4), 0, (hipStream_t)stream, src, dst, C, G, H, W);
    return ssm::check_launch("ssm_hl8_from_f32");
}

extern "C" int ssm_hl8_to_f32(ssm_hview src, ssm_view dst, int B, int C, int G, int H, int W, void *stream) {
    SSM_REQUIRE(src.ptr && dst.ptr && B > 0 && C > 0 && G * 8 >= C && H > 0 && W > 0, "hl8_to_f32: bad arguments");
    SSM_REQUIRE((long long)B * G <= 65535, "hl8_to_f32: B*G too large");
    SSM_LAUNCH(from_hl8_kernel, dim3((W + 63) / 64, (H + 3) / 4, B * G), dim3(64, 4), 0, (hipStream_t)stream, src, dst, C, G, H, W);
    return ssm::check_launch("ssm_hl8_to_f32");
}

// ---- Q8 operand form (flag SSM_FLAG_Q8): filter packing and layout conversion ----
extern "C" int ssm_conv16q_config(int k, int Cout, int W, int *BN, int *KYS) {
    switch (pick16(k, Cout, W)) {
        case H7: dims16<C16K7<2>>(BN, KYS); break;
        case H5: dims16<C16K5<2>>(BN, KYS); break;
        case H3N32:
        case H3N32D: dims16<C16K3N32<2>>(BN, KYS); break;
        case H3N32V:
        case H3N32W: dims16<C16K3N32W<2>>(BN, KYS); break;
        case H3N64: dims16<C16K3N64<2>>(BN, KYS); break;
        case H3N128: dims16<C16K3N128<2>>(BN, KYS); break;
        case H3N128S: dims16<C16K3N128S<2>>(BN, KYS); break;
        default:
            ssm::set_error("conv16q: kernel size %d unsupported (3, 5, 7 are)", k);
            return SSM_E_UNSUPPORTED;
    }
    return SSM_OK;
}

static inline long long q8_stage_bytes(int k, int KYS, int BN) { return (long long)(KYS * k * 2 + KYS * ((k + 3) / 4) * 8) * BN * 16; }

extern "C" size_t ssm_packed16q_weight_bytes(int Cout, int CinP, int k, int BN, int KYS) {
    return (size_t)((Cout + BN - 1) / BN) * (size_t)(CinP / 16) * (size_t)(k / KYS) * (size_t)q8_stage_bytes(k, KYS, BN);
}

extern "C" int ssm_pack16q_weights(const float *w, const float *bias, void *wp, float *bp, int Cout, int Cin, int CinP, int k, int BN,
                                   int KYS, float scale, void *stream) {
    SSM_REQUIRE(w && bias && wp && bp, "pack16q: null pointer");
    SSM_REQUIRE(Cout > 0 && Cin > 0 && CinP >= Cin && CinP % 16 == 0 && BN % 32 == 0 && k % KYS == 0, "pack16q: bad sizes");
    const long long stages = (long long)((Cout + BN - 1) / BN) * (CinP / 16) * (k / KYS);
    const long long sb = q8_stage_bytes(k, KYS, BN);
    const long long qbase = (long long)KYS * k * 2 * BN * 16;
    const long long total = (long long)((Cout + BN - 1) / BN) * (CinP / 16) * BN * k;
    const dim3 grid((unsigned)((total + 63) / 64));
    (void)stages;
    switch (k) {
        case 3: SSM_LAUNCH(pack16q_kernel<3>, grid, dim3(64), 0, (hipStream_t)stream, w, (char *)wp, Cout, Cin, CinP, KYS, BN, scale, sb, qbase, total); break;
        case 5: SSM_LAUNCH(pack16q_kernel<5>, grid, dim3(64), 0, (hipStream_t)stream, w, (char *)wp, Cout, Cin, CinP, KYS, BN, scale, sb, qbase, total); break;
        case 7: SSM_LAUNCH(pack16q_kernel<7>, grid, dim3(64), 0, (hipStream_t)stream, w, (char *)wp, Cout, Cin, CinP, KYS, BN, scale, sb, qbase, total); break;
        default: ssm::set_error("pack16q: kernel size %d unsupported", k); return SSM_E_UNSUPPORTED;
    }
    const int nbias = (int)ssm_packed_bias_floats(Cout, BN);
    SSM_LAUNCH(pack16_kernel, dim3((unsigned)((nbias + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, bias, (_Float16 *)nullptr, bp,
                       Cout, Cin, CinP, k, KYS, BN, scale, 0LL, nbias);
    return ssm::check_launch("ssm_pack16q_weights");
}

extern "C" int ssm_pack16q_job_blocks(int Cout, int CinP, int k, int BN, int *row_blocks, int *bias_blocks) {
    SSM_REQUIRE(row_blocks && bias_blocks && Cout > 0 && CinP % 16 == 0 && BN % 32 == 0, "pack16q_job_blocks: bad arguments");
    const long long total = (long long)((Cout + BN - 1) / BN) * (CinP / 16) * BN * k;
    *row_blocks = (int)((total + 63) / 64);
    *bias_blocks = (int)((ssm_packed_bias_floats(Cout, BN) + 63) / 64);
    return SSM_OK;
}

extern "C" int ssm_pack16q_weights_batch(const ssm_pack16q_job *jobs_device, int n_jobs, int total_blocks, void *stream) {
    SSM_REQUIRE(jobs_device && n_jobs > 0 && total_blocks > 0, "pack16q_batch: bad arguments");
    SSM_LAUNCH(pack16q_batch_kernel, dim3((unsigned)total_blocks), dim3(64), 0, (hipStream_t)stream, jobs_device, n_jobs);
    return ssm::check_launch("ssm_pack16q_weights_batch");
}

extern "C" int ssm_hq8_from_f32(ssm_view src, ssm_hview dst, int B, int C, int G, int H, int W, void *stream) {
    SSM_REQUIRE(src.ptr && dst.ptr && B > 0 && C > 0 && G * 8 >= C && G % 2 == 0 && H > 0 && W > 0, "hq8_from_f32: bad arguments (Q8 tensors hold an even number of channel groups)");
    SSM_REQUIRE((long long)B * G <= 65535, "hq8_from_f32: B*G too large");
    SSM_LAUNCH(to_hq8_kernel, dim3((W + 63) / 64, (H + 3) / 4, B * G), dim3(64, 4), 0, (hipStream_t)stream, src, dst, C, G, H, W);
    return ssm::check_launch("ssm_hq8_from_f32");
}

extern "C" int ssm_hq8_to_f32(ssm_hview src, ssm_view dst, int B, int C, int G, int H, int W, void *stream) {
    SSM_REQUIRE(src.ptr && dst.ptr && B > 0 && C > 0 && G * 8 >= C && G % 2 == 0 && H > 0 && W > 0, "hq8_to_f32: bad arguments (Q8 tensors hold an even number of channel groups)");
    SSM_REQUIRE((long long)B * G <= 65535, "hq8_to_f32: B*G too large");
    SSM_LAUNCH(from_hq8_kernel, dim3((W + 63) / 64, (H + 3) / 4, B * G), dim3(64, 4), 0, (hipStream_t)stream, src, dst, C, G, H, W);
    return ssm::check_launch("ssm_hq8_to_f32");
}

extern "C" int ssm_conv2d_hl8_fwd(ssm_hview x1, int C1, ssm_hview x2, int C2, const void *w_packed, const float *bias_packed,
                                  float wscale, ssm_hview y_hl8, ssm_view y_f32, ssm_hview pool_hl8, int B, int H, int W,
                                  int Cout, int k, float slope, int flags, void *stream) {
    SSM_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && C1 > 0 && C2 >= 0, "conv16: bad sizes");
    SSM_REQUIRE(x1.ptr && w_packed && bias_packed && (y_hl8.ptr || y_f32.ptr), "conv16: null pointer");
    SSM_REQUIRE(C1 % 16 == 0 && C2 % 16 == 0, "conv16: channel counts (%d,%d) must be multiples of 16", C1, C2);
    SSM_REQUIRE(ssm::aligned16(x1.ptr) && ssm::aligned16(w_packed), "conv16: 16-byte alignment");
    SSM_REQUIRE(x1.sh >= W + 2 * SSM_PADX, "conv16: input row stride %d leaves no zero frame for W=%d", x1.sh, W);
    if (C2 > 0) SSM_REQUIRE(x2.ptr && x2.sh == x1.sh && x2.sg == x1.sg && x2.sp == x1.sp, "conv16: cat sources must share strides");
    if (y_hl8.ptr) SSM_REQUIRE(Cout % 8 == 0, "conv16: HL8 output needs Cout %% 8 == 0 (got %d)", Cout);
    if (pool_hl8.ptr) SSM_REQUIRE(H % 2 == 0 && W % 2 == 0 && Cout % 8 == 0, "conv16: fused pool needs even H, W and Cout %% 8 == 0");
    SSM_REQUIRE(x1.sg * 32 < 0x7fffffffLL, "conv16: plane too large for 32-bit piece offsets");
    if (pick16(k, Cout, W) < 0) {
        ssm::set_error("conv16: kernel size %d unsupported", k);
        return SSM_E_UNSUPPORTED;
    }
    Conv16Params p;
    p.src1 = (const char *)x1.ptr;
    p.src2 = C2 > 0 ? (const char *)x2.ptr : (const char *)x1.ptr;
    p.sb1 = x1.sb;
    p.sb2 = C2 > 0 ? x2.sb : 0;
    p.sg = x1.sg;
    p.sp = x1.sp;
    p.sh = x1.sh;
    p.C1 = C1;
    p.Cin = C1 + C2;
    p.wpk = (const char *)w_packed;
    p.bias = bias_packed;
    p.wscale = wscale;
    p.dh = (char *)y_hl8.ptr;
    p.dhsb = y_hl8.sb; p.dhsg = y_hl8.sg; p.dhsp = y_hl8.sp; p.dhsh = y_hl8.sh;
    p.df = y_f32.ptr;
    p.dfsb = y_f32.sb; p.dfsc = y_f32.sc; p.dfsh = y_f32.sh;
    p.ph = (char *)pool_hl8.ptr;
    p.phsb = pool_hl8.sb; p.phsg = pool_hl8.sg; p.phsp = pool_hl8.sp; p.phsh = pool_hl8.sh;
    p.H = H; p.W = W; p.Cout = Cout;
    p.slope = slope;
    p.lrelu = (flags & SSM_FLAG_LRELU) ? 1 : 0;
    p.shuf = 0;
    p.shuf_t = 0;
    p.skipy = p.skipx = 0;
    hipStream_t st = (hipStream_t)stream;
    if (flags & SSM_FLAG_Q8) {
        SSM_REQUIRE((!y_hl8.ptr && !pool_hl8.ptr) || Cout % 16 == 0, "conv16: Q8 output needs Cout %% 16 == 0 (got %d)", Cout);
        return dispatch16<2>(p, B, k, st);
    }
    if (flags & SSM_FLAG_FP16_FAST) return dispatch16<0>(p, B, k, st);
    return dispatch16<1>(p, B, k, st);
}

// Sub-pixel form of conv3x3(upsample2x(cat[x1, x2])): a plain 3x3 convolution of the LOW-res tensors with 4*Cr outputs (effective
// filters per output parity, built by the caller) whose epilogue scatters channel (2*pa + pb)*Cr + c of low-res pixel (y, x) to
// pixel (2y + pa, 2x + pb) of channel c of y_hl8 (transposed: (2x + pb, 2y + pa), for column strips run on transposed copies).
// y_hl8 points at the destination pixel of low-res pixel (0, 0) of this launch.  Q8 form only.
extern "C" int ssm_conv2d_hl8_subpixel_fwd(ssm_hview x1, int C1, ssm_hview x2, int C2, const void *w_packed, const float *bias_packed,
                                           float wscale, ssm_hview y_hl8, int B, int H, int W, int Cr, int transposed, float slope,
                                           int flags, void *stream) {
    SSM_REQUIRE(B > 0 && H > 0 && W > 0 && Cr > 0 && C1 > 0 && C2 >= 0, "conv16 subpixel: bad sizes");
    SSM_REQUIRE(x1.ptr && w_packed && bias_packed && y_hl8.ptr, "conv16 subpixel: null pointer");
    SSM_REQUIRE((flags & SSM_FLAG_Q8) && Cr % 16 == 0, "conv16 subpixel: Q8 operands and Cr %% 16 == 0 (got %d)", Cr);
    SSM_REQUIRE(C1 % 16 == 0 && C2 % 16 == 0, "conv16 subpixel: channel counts (%d,%d) must be multiples of 16", C1, C2);
    SSM_REQUIRE(ssm::aligned16(x1.ptr) && ssm::aligned16(w_packed) && ssm::aligned16(y_hl8.ptr), "conv16 subpixel: 16-byte alignment");
    if (C2 > 0) SSM_REQUIRE(x2.ptr && x2.sh == x1.sh && x2.sg == x1.sg && x2.sp == x1.sp, "conv16 subpixel: cat sources must share strides");
    SSM_REQUIRE(x1.sg * 32 < 0x7fffffffLL, "conv16 subpixel: plane too large for 32-bit piece offsets");
    Conv16Params p;
    p.src1 = (const char *)x1.ptr;
    p.src2 = C2 > 0 ? (const char *)x2.ptr : (const char *)x1.ptr;
    p.sb1 = x1.sb;
    p.sb2 = C2 > 0 ? x2.sb : 0;
    p.sg = x1.sg;
    p.sp = x1.sp;
    p.sh = x1.sh;
    p.C1 = C1;
    p.Cin = C1 + C2;
    p.wpk = (const char *)w_packed;
    p.bias = bias_packed;
    p.wscale = wscale;
    p.dh = (char *)y_hl8.ptr;
    p.dhsb = y_hl8.sb; p.dhsg = y_hl8.sg; p.dhsp = y_hl8.sp; p.dhsh = y_hl8.sh;
    p.df = nullptr;
    p.dfsb = p.dfsc = 0; p.dfsh = 0;
    p.ph = nullptr;
    p.phsb = p.phsg = p.phsp = 0; p.phsh = 0;
    p.H = H; p.W = W; p.Cout = 4 * Cr;
    p.slope = slope;
    p.lrelu = (flags & SSM_FLAG_LRELU) ? 1 : 0;
    p.shuf = Cr;
    p.shuf_t = transposed ? 1 : 0;
    p.skipy = p.skipx = 0;
    return dispatch16<2>(p, B, 3, (hipStream_t)stream);
}

// ---- the nine problems of a sub-pixel decoder level in ONE launch ------------------------------------------------------------------
// All problems use the tile configuration of pick16(3, 4*Cr, wcfg) (their filters are packed for it) and own disjoint output pixels
// (skip_y / skip_x), so they need no ordering: the strips' and corners' few workgroups run beside the main problem's instead of as
// eight latency-bound launches behind it.
template <class C>
static int fill_problem(Conv16Params &p, int B, long long &blocks) {
    p.tilesX = (p.W + C::TW - 1) / C::TW;
    p.tilesY = (p.H + C::TH - 1) / C::TH;
    p.NB = (p.Cout + C::BN - 1) / C::BN;
    blocks = (long long)p.tilesX * p.tilesY * p.NB * B;
    return SSM_OK;
}

template <class C>
static int launch16_multi(const Conv16Multi &m, hipStream_t st) {
    if (!reserve_lds((const void *)conv16_multi_kernel<C>, C::LDS_BYTES)) return SSM_E_LAUNCH;
    SSM_LAUNCH(conv16_multi_kernel<C>, dim3((unsigned)m.start[m.n]), dim3(C::NTHREADS_ALL), C::LDS_BYTES, st, m);
    return ssm::check_launch("ssm_conv16_subpixel_run");
}

extern "C" size_t ssm_conv16_subpixel_table_bytes(int n) { return sizeof(Conv16Params) * (size_t)(n > 0 ? n : 0); }

extern "C" int ssm_conv16_subpixel_plan(const ssm_subpixel_problem *pr, int n, int B, int Cr, int wcfg, float slope, int flags,
                                        void *table_host, size_t table_bytes, int *block_start) {
    SSM_REQUIRE(pr && table_host && block_start && n > 0 && n <= SSM_MULTI_MAX && B > 0 && Cr > 0 && Cr % 16 == 0, "subpixel_plan: bad arguments");
    SSM_REQUIRE(table_bytes >= sizeof(Conv16Params) * (size_t)n, "subpixel_plan: table buffer too small");
    SSM_REQUIRE(flags & SSM_FLAG_Q8, "subpixel_plan: Q8 operands only");
    const int kind = pick16(3, 4 * Cr, wcfg);
    Conv16Params *tab = (Conv16Params *)table_host;
    long long off = 0;
    for (int i = 0; i < n; ++i) {
        const ssm_subpixel_problem &q = pr[i];
        SSM_REQUIRE(q.x1.ptr && q.w_packed && q.bias_packed && q.y_hl8.ptr && q.H > 0 && q.W > 0 && q.C1 > 0 && q.C1 % 16 == 0 && q.C2 % 16 == 0,
                    "subpixel_plan: problem %d malformed", i);
        if (q.C2 > 0) SSM_REQUIRE(q.x2.ptr && q.x2.sh == q.x1.sh && q.x2.sg == q.x1.sg && q.x2.sp == q.x1.sp, "subpixel_plan: cat sources must share strides");
        Conv16Params &p = tab[i];
        p.src1 = (const char *)q.x1.ptr;
        p.src2 = q.C2 > 0 ? (const char *)q.x2.ptr : (const char *)q.x1.ptr;
        p.sb1 = q.x1.sb;
        p.sb2 = q.C2 > 0 ? q.x2.sb : 0;
        p.sg = q.x1.sg; p.sp = q.x1.sp; p.sh = q.x1.sh;
        p.C1 = q.C1; p.Cin = q.C1 + q.C2;
        p.wpk = (const char *)q.w_packed;
        p.bias = q.bias_packed;
        p.wscale = q.inv_wscale;
        p.dh = (char *)q.y_hl8.ptr;
        p.dhsb = q.y_hl8.sb; p.dhsg = q.y_hl8.sg; p.dhsp = q.y_hl8.sp; p.dhsh = q.y_hl8.sh;
        p.df = nullptr; p.dfsb = p.dfsc = 0; p.dfsh = 0;
        p.ph = nullptr; p.phsb = p.phsg = p.phsp = 0; p.phsh = 0;
        p.H = q.H; p.W = q.W; p.Cout = 4 * Cr;
        p.slope = slope;
        p.lrelu = (flags & SSM_FLAG_LRELU) ? 1 : 0;
        p.shuf = Cr; p.shuf_t = q.transposed ? 1 : 0;
        p.skipy = q.skip_y ? 1 : 0; p.skipx = q.skip_x ? 1 : 0;
        long long blocks = 0;
        switch (kind) {
            case H3N32: fill_problem<C16K3N32<2>>(p, B, blocks); break;
            case H3N32D: fill_problem<C16K3N32D<2>>(p, B, blocks); break;
            case H3N32V: fill_problem<C16K3N32V<2>>(p, B, blocks); break;
            case H3N32W: fill_problem<C16K3N32W<2>>(p, B, blocks); break;
            case H3N64: fill_problem<C16K3N64<2>>(p, B, blocks); break;
            case H3N128: fill_problem<C16K3N128<2>>(p, B, blocks); break;
            case H3N128S: fill_problem<C16K3N128S<2>>(p, B, blocks); break;
            default: ssm::set_error("subpixel_plan: no tile configuration"); return SSM_E_UNSUPPORTED;
        }
        block_start[i] = (int)off;
        off += blocks;
        SSM_REQUIRE(off < 0x7fffffffLL, "subpixel_plan: grid too large");
    }
    block_start[n] = (int)off;
    return SSM_OK;
}

extern "C" int ssm_conv16_subpixel_run(const void *table_device, int n, const int *block_start, int Cr, int wcfg, int Cin, void *stream) {
    SSM_REQUIRE(table_device && block_start && n > 0 && n <= SSM_MULTI_MAX, "subpixel_run: bad arguments");
    Conv16Multi m;
    m.table = (const Conv16Params *)table_device;
    m.n = n;
    for (int i = 0; i <= n; ++i) m.start[i] = block_start[i];
    for (int i = n + 1; i <= SSM_MULTI_MAX; ++i) m.start[i] = block_start[n];
    hipStream_t st = (hipStream_t)stream;
    switch (pick16(3, 4 * Cr, wcfg, Cin)) {
        case H3N32: return launch16_multi<C16K3N32<2>>(m, st);
        case H3N32D: return launch16_multi<C16K3N32D<2>>(m, st);
        case H3N32V: return launch16_multi<C16K3N32V<2>>(m, st);
        case H3N32W: return launch16_multi<C16K3N32W<2>>(m, st);
        case H3N64: return launch16_multi<C16K3N64<2>>(m, st);
        case H3N128: return launch16_multi<C16K3N128<2>>(m, st);
        case H3N128S: return launch16_multi<C16K3N128S<2>>(m, st);
    }
    return SSM_E_UNSUPPORTED;
}

// Image columns of a two-source HL8 / Q8 tensor as the rows of a (zero-framed) tensor with Ga + Gb groups - the input of the
// sub-pixel convolution's column strips: dst rows 0..ncols-1 = columns col0.., and (col1 >= 0) dst rows ncols+1..2*ncols = columns
// col1.. (row ncols stays untouched: the zero gap between the two strips' neighbourhoods).
__global__ __launch_bounds__(256) void gather_cols_kernel(ssm_hview a, int Ga, ssm_hview bsrc, int Gb, ssm_hview dst, int H, int col0, int ncols,
                                                          int col1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int G = Ga + Gb;
    const int b = blockIdx.z / G, g = blockIdx.z - b * G;
    const int r = blockIdx.y;
    if (i >= H || r == ncols) return;
    const int col = r < ncols ? col0 + r : col1 + (r - ncols - 1);
    const ssm_hview &s = g < Ga ? a : bsrc;
    const int gg = g < Ga ? g : g - Ga;
    typedef int gi4 __attribute__((ext_vector_type(4)));
    const char *sp = (const char *)s.ptr + ((long long)b * s.sb + (long long)gg * s.sg + (long long)i * s.sh + col) * 16;
    char *dp = (char *)dst.ptr + ((long long)b * dst.sb + (long long)g * dst.sg + (long long)r * dst.sh + i) * 16;
    *reinterpret_cast<gi4 *>(dp) = *reinterpret_cast<const gi4 *>(sp);
    *reinterpret_cast<gi4 *>(dp + dst.sp * 16) = *reinterpret_cast<const gi4 *>(sp + s.sp * 16);
}

extern "C" int ssm_hl8_gather_cols(ssm_hview a, int Ga, ssm_hview b, int Gb, ssm_hview dst, int B, int H, int col0, int ncols, int col1,
                                   void *stream) {
    SSM_REQUIRE(a.ptr && dst.ptr && Ga > 0 && Gb >= 0 && (Gb == 0 || b.ptr) && B > 0 && H > 0 && ncols > 0 && col0 >= 0, "gather_cols: bad arguments");
    SSM_REQUIRE((long long)B * (Ga + Gb) <= 65535 && ncols <= 1024, "gather_cols: grid too large");
    const int rows = col1 >= 0 ? 2 * ncols + 1 : ncols;
    SSM_LAUNCH(gather_cols_kernel, dim3((H + 255) / 256, rows, B * (Ga + Gb)), dim3(256), 0, (hipStream_t)stream, a, Ga, b, Gb, dst, H,
                       col0, ncols, col1);
    return ssm::check_launch("ssm_hl8_gather_cols");
}

// conv3x3(upsample2x(cat[a, b])) with a, b LOW-res HL8 tensors [B,C1|C2,H/2,W/2]; H, W = output (hi-res) size.
// Uses the filter packing of ssm_conv16_config(3, Cout, W) with KYS taken from ssm_conv16_ups_config.
extern "C" int ssm_conv16_ups_config(int Cout, int W, int *BN, int *KYS) {
    switch (pick16(3, Cout, W, 0)) {
        case H3N32:
        case H3N32D:
        case H3N32V:
        case H3N32W: *BN = U3N32<1>::BN; *KYS = U3N32<1>::KYS; break;
        case H3N64: *BN = U3N64<1>::BN; *KYS = U3N64<1>::KYS; break;
        case H3N128: *BN = U3N128<1>::BN; *KYS = U3N128<1>::KYS; break;
        case H3N128S: *BN = U3N128S<1>::BN; *KYS = U3N128S<1>::KYS; break;
        default: return SSM_E_UNSUPPORTED;
    }
    return SSM_OK;
}

extern "C" int ssm_conv16q_ups_config(int Cout, int W, int *BN, int *KYS) {
    switch (pick16(3, Cout, W, 0)) {
        case H3N32:
        case H3N32D:
        case H3N32V:
        case H3N32W: *BN = U3N32<2>::BN; *KYS = U3N32<2>::KYS; break;
        case H3N64: *BN = U3N64<2>::BN; *KYS = U3N64<2>::KYS; break;
        case H3N128: *BN = U3N128<2>::BN; *KYS = U3N128<2>::KYS; break;
        case H3N128S: *BN = U3N128S<2>::BN; *KYS = U3N128S<2>::KYS; break;
        default: return SSM_E_UNSUPPORTED;
    }
    return SSM_OK;
}

extern "C" int ssm_conv2d_ups_hl8_fwd(ssm_hview a, int C1, ssm_hview b, int C2, const void *w_packed, const float *bias_packed,
                                      float wscale, ssm_hview y_hl8, ssm_view y_f32, int B, int H, int W, int Cout, float slope,
                                      int flags, void *stream) {
    SSM_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && Cout > 0 && C1 > 0 && C2 >= 0, "conv16_ups: bad sizes");
    SSM_REQUIRE(a.ptr && w_packed && bias_packed && (y_hl8.ptr || y_f32.ptr), "conv16_ups: null pointer");
    SSM_REQUIRE(C1 % 16 == 0 && C2 % 16 == 0, "conv16_ups: channel counts (%d,%d) must be multiples of 16", C1, C2);
    SSM_REQUIRE(ssm::aligned16(a.ptr) && ssm::aligned16(w_packed), "conv16_ups: 16-byte alignment");
    SSM_REQUIRE(a.sh >= W / 2 + 2 * SSM_PADX, "conv16_ups: source row stride %d leaves no frame for w=%d", a.sh, W / 2);
    if (C2 > 0) SSM_REQUIRE(b.ptr && b.sh == a.sh && b.sg == a.sg && b.sp == a.sp, "conv16_ups: cat sources must share strides");
    if (y_hl8.ptr) SSM_REQUIRE(Cout % 8 == 0, "conv16_ups: HL8 output needs Cout %% 8 == 0 (got %d)", Cout);
    SSM_REQUIRE(a.sg * 32 < 0x7fffffffLL, "conv16_ups: plane too large for 32-bit piece offsets");
    Conv16Params p;
    p.src1 = (const char *)a.ptr;
    p.src2 = C2 > 0 ? (const char *)b.ptr : (const char *)a.ptr;
    p.sb1 = a.sb;
    p.sb2 = C2 > 0 ? b.sb : 0;
    p.sg = a.sg; p.sp = a.sp; p.sh = a.sh;
    p.C1 = C1;
    p.Cin = C1 + C2;
    p.wpk = (const char *)w_packed;
    p.bias = bias_packed;
    p.wscale = wscale;
    p.dh = (char *)y_hl8.ptr;
    p.dhsb = y_hl8.sb; p.dhsg = y_hl8.sg; p.dhsp = y_hl8.sp; p.dhsh = y_hl8.sh;
    p.df = y_f32.ptr;
    p.dfsb = y_f32.sb; p.dfsc = y_f32.sc; p.dfsh = y_f32.sh;
    p.ph = nullptr;
    p.phsb = p.phsg = p.phsp = 0; p.phsh = 0;
    p.H = H; p.W = W; p.Cout = Cout;
    p.slope = slope;
    p.lrelu = (flags & SSM_FLAG_LRELU) ? 1 : 0;
    p.shuf = 0;
    p.shuf_t = 0;
    p.skipy = p.skipx = 0;
    hipStream_t st = (hipStream_t)stream;
    if (flags & SSM_FLAG_Q8) {
        SSM_REQUIRE(!y_hl8.ptr || Cout % 16 == 0, "conv16_ups: Q8 output needs Cout %% 16 == 0 (got %d)", Cout);
        return dispatch16_ups<2>(p, B, st);
    }
    if (flags & SSM_FLAG_FP16_FAST) return dispatch16_ups<0>(p, B, st);
    return dispatch16_ups<1>(p, B, st);
}
