// fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces layers.conv / final_conv of the reference (scripts/models/layers.py:21-33,
// scripts/models/flow_computation.py:145-153): stride-1 'same' cross-correlation,
// zero padding, bias, optional LeakyReLU, optional fused 2x2 average pool
// (scripts/models/layers.py:60-63) and optional two-source input (torch.cat on C).
//
// GEMM view, per batch element:  D[cout][pixel] = sum_k W[cout][k] * X[k][pixel],
// k = (cin, ky, kx).  The MFMA A operand is the filter (32 couts x 2 k), the B
// operand is the activation (2 k x 32 pixels), so the accumulator has the pixel
// on the lane: consecutive lanes own consecutive x of one output row and every
// store instruction writes whole 128-byte row segments of NCHW planes.
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
#include "ssm_common.h"

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
    int H, W, Cout;
    int tilesX, tilesY, NB;
    float slope;
    int lrelu;
};

template <int KS_, int NT_, int WN_, int MTY_, int MTX_, int WY_, int WX_, int CK_>
struct Cfg {
    static constexpr int KS = KS_, NT = NT_, WN = WN_, MTY = MTY_, MTX = MTX_, WY = WY_, WX = WX_, CK = CK_;
    static constexpr int KS2 = KS * KS, PAD = (KS - 1) / 2;
    static constexpr int BN = 32 * NT * WN;       // output channels per workgroup
    static constexpr int TH = MTY * WY;           // output rows per workgroup
    static constexpr int TW = 32 * MTX * WX;      // output columns per workgroup
    static constexpr int MT = MTY * MTX;          // 32-pixel tiles per wave
    static constexpr int PH = TH + KS - 1;        // patch rows
    static constexpr int PW = TW + 8;             // patch columns (16-byte aligned both ends)
    static constexpr int PW4 = PW / 4;
    static constexpr int WSZ = CK * KS2 * BN;     // filter floats per chunk
    static constexpr int PSZ = CK * PH * PW;      // patch floats per chunk
    static constexpr int NWQ = WSZ / 4, NPQ = PSZ / 4, NQ = NWQ + NPQ;  // 16-byte pieces
    static constexpr int NG = (NQ + 63) / 64;     // 1-KiB wave-instructions per chunk
    static constexpr int STAGE = NG * 256;        // floats per LDS stage
    static constexpr int NI = (NG + 3) / 4;       // LDS-DMA instructions per wave per chunk
    static constexpr int LDS_BYTES = 2 * STAGE * 4;
    static_assert(WN * WY * WX == 4, "4 waves per workgroup");
    static_assert(CK % 2 == 0, "one MFMA k-step = two input channels");
    static_assert(LDS_BYTES <= 65536, "LDS budget");
};

#define SSM_GLDS16(gp, lp)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp),      \
                                     (__attribute__((address_space(3))) void *)(lp), 16, 0, 0)

template <class C>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int KS = C::KS, KS2 = C::KS2, BN = C::BN, PH = C::PH, PW = C::PW, NT = C::NT, MT = C::MT;

    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid % C::WN, wy = (wid / C::WN) % C::WY, wx = wid / (C::WN * C::WY);

    int id = ssm_xcd_tile(blockIdx.x, gridDim.x);
    const int nb = id % p.NB;
    id /= p.NB;
    const int tx = id % p.tilesX;
    id /= p.tilesX;
    const int ty = id % p.tilesY;
    const int b = id / p.tilesY;
    const int x0 = tx * C::TW, y0 = ty * C::TH;

    // patch origin = element (c, y0-PAD, x0-4) of the padded planes
    const long long porg = (long long)(y0 - C::PAD) * p.sh + (x0 - 4);
    const float *pbase1 = p.src1 + (long long)b * p.sb1 + porg;
    const float *pbase2 = p.src2 + (long long)b * p.sb2 + porg;
    const float *wbase = p.wpk + (long long)nb * p.Cin * (KS2 * BN);

    // per-lane source offset of each LDS-DMA piece this wave issues (same for every chunk)
    int off[C::NI];
    bool isw[C::NI];
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
        const int q = (i * 4 + wid) * 64 + lane;
        if (q < C::NWQ) {
            isw[i] = true;
            off[i] = q * 4;
        } else if (q < C::NQ) {
            const int qq = q - C::NWQ;
            const int c = qq / (PH * C::PW4);
            const int rem = qq - c * (PH * C::PW4);
            const int r = rem / C::PW4;
            const int j = rem - r * C::PW4;
            isw[i] = false;
            off[i] = (int)(c * p.sc) + r * p.sh + 4 * j;
        } else {  // tail of the last 1-KiB piece: lands in the stage's padding
            isw[i] = true;
            off[i] = 0;
        }
    }

    auto issue = [&](int ch, int stage) {
        const int c0 = ch * C::CK;
        const float *pb = (c0 < p.C1) ? pbase1 + (long long)c0 * p.sc : pbase2 + (long long)(c0 - p.C1) * p.sc;
        const float *wb = wbase + (long long)c0 * (KS2 * BN);
        float *ls = lds + stage * C::STAGE;
#pragma unroll
        for (int i = 0; i < C::NI; ++i) {
            const int g = i * 4 + wid;
            if (g < C::NG) {
                const float *gp = (isw[i] ? wb : pb) + off[i];
                SSM_GLDS16(gp, ls + g * 256);
            }
        }
    };

    f32x16 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;

    // per-lane operand bases inside a stage (floats)
    const int aBase = half * (KS2 * BN) + wn * (NT * 32) + l31;
    const int bBase = C::WSZ + half * (PH * PW) + (wy * C::MTY) * PW + wx * (C::MTX * 32) + l31 + (4 - C::PAD);

    const int nchunks = p.Cin / C::CK;
    issue(0, 0);
    for (int ch = 0; ch < nchunks; ++ch) {
        // chunk ch has landed for every wave; every wave is done reading chunk ch-1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ch + 1 < nchunks) issue(ch + 1, (ch + 1) & 1);

        const float *sa = lds + (ch & 1) * C::STAGE + aBase;
        const float *sb = lds + (ch & 1) * C::STAGE + bBase;
#pragma unroll
        for (int cp = 0; cp < C::CK / 2; ++cp) {
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    float a[NT], bv[MT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) a[n] = sa[(2 * cp * KS2 + ky * KS + kx) * BN + n * 32];
#pragma unroll
                    for (int my = 0; my < C::MTY; ++my)
#pragma unroll
                        for (int mx = 0; mx < C::MTX; ++mx)
                            bv[my * C::MTX + mx] = sb[(2 * cp * PH + my + ky) * PW + mx * 32 + kx];
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[n], bv[m], acc[n][m], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: bias, LeakyReLU, store (and fused 2x2 mean) -------------------------
    // accumulator register r of lane (l31, half) = cout (r&3) + 8*(r>>2) + 4*half, pixel l31
    const int xbase = x0 + wx * (C::MTX * 32) + l31;
    const int ybase = y0 + wy * C::MTY;
    float *dstb = p.dst + (long long)b * p.dsb;
    float *poolb = p.pool ? p.pool + (long long)b * p.psb : nullptr;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = (wn * NT + n) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;  // within the cout block
            const int co = nb * BN + cl;
            const float bias = p.bias[co];
            const bool cok = co < p.Cout;
            float v[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float t = acc[n][m][r] + bias;
                if (p.lrelu) t = t > 0.f ? t : t * p.slope;
                v[m] = t;
            }
#pragma unroll
            for (int my = 0; my < C::MTY; ++my)
#pragma unroll
                for (int mx = 0; mx < C::MTX; ++mx) {
                    const int y = ybase + my, x = xbase + mx * 32;
                    if (cok && y < p.H && x < p.W) dstb[(long long)co * p.dsc + (long long)y * p.dsh + x] = v[my * C::MTX + mx];
                }
            if (poolb) {
                if constexpr (C::MTY % 2 == 0) {
#pragma unroll
                    for (int my = 0; my < C::MTY; my += 2)
#pragma unroll
                        for (int mx = 0; mx < C::MTX; ++mx) {
                            float s = v[my * C::MTX + mx] + v[(my + 1) * C::MTX + mx];
                            s += __shfl_xor(s, 1);
                            const int y = ybase + my, x = xbase + mx * 32;
                            if (cok && !(l31 & 1) && y < p.H && x < p.W)
                                poolb[(long long)co * p.psc + (long long)(y >> 1) * p.psh + (x >> 1)] = s * 0.25f;
                        }
                }
            }
        }
    }
}

// ---- tile configurations ------------------------------------------------------------
//            KS NT WN MTY MTX WY WX CK        BN   TH  TW
using CfgK7 = Cfg<7, 1, 1, 2, 2, 4, 1, 2>;  //  32    8  64   conv1a / conv1b
using CfgK5 = Cfg<5, 2, 1, 2, 2, 4, 1, 2>;  //  64    8  64   conv2a / conv2b
using CfgK3N32 = Cfg<3, 1, 1, 2, 2, 4, 1, 4>;   //  32    8  64   conv11a/b, fuse, final
using CfgK3N64 = Cfg<3, 2, 1, 2, 2, 4, 1, 4>;   //  64    8  64   conv10a/b
using CfgK3N128 = Cfg<3, 2, 2, 2, 2, 2, 1, 4>;  // 128    4  64   conv3..conv9 (wide maps)
using CfgK3N128S = Cfg<3, 2, 2, 2, 1, 2, 1, 4>;  // 128   4  32   same, maps where 64-wide tiles waste columns
// small maps (1/16, 1/32 resolution at batch 1): more, smaller workgroups to cover 256 CUs
using CfgK3N64T = Cfg<3, 1, 2, 2, 1, 2, 1, 4>;   //  64    4  32
using CfgK3N32T = Cfg<3, 1, 1, 1, 1, 4, 1, 8>;   //  32    4  32   (odd row tile: no fused pool)

enum ConvKind { K7 = 0, K5, K3N32, K3N64, K3N128, K3N128S, K3N64T, K3N32T, NKIND };

constexpr int kFillBlocks = 256;   // one workgroup per CU

template <class C>
long long grid_blocks(int B, int H, int W, int Cout) {
    return (long long)B * ((W + C::TW - 1) / C::TW) * ((H + C::TH - 1) / C::TH) * ((Cout + C::BN - 1) / C::BN);
}

int pick_kind(int k, int Cout, int B, int H, int W, int pool) {
    if (k == 7) return K7;
    if (k == 5) return K5;
    if (k != 3) return -1;
    if (Cout <= 32) return K3N32;
    if (Cout <= 64) return K3N64;
    const int w64 = (W + 63) / 64 * 64, w32 = (W + 31) / 32 * 32;
    const bool narrow = w32 < w64;
    const long long nb = narrow ? grid_blocks<CfgK3N128S>(B, H, W, Cout) : grid_blocks<CfgK3N128>(B, H, W, Cout);
    if (nb >= kFillBlocks) return narrow ? K3N128S : K3N128;
    if (pool || grid_blocks<CfgK3N64T>(B, H, W, Cout) >= kFillBlocks) return K3N64T;
    return K3N32T;
}

template <class C>
int launch(ConvParams &p, int B, hipStream_t st) {
    p.tilesX = (p.W + C::TW - 1) / C::TW;
    p.tilesY = (p.H + C::TH - 1) / C::TH;
    p.NB = (p.Cout + C::BN - 1) / C::BN;
    if (p.pool && (C::MTY % 2 != 0)) {
        ssm::set_error("conv: fused pool needs an even row tile");
        return SSM_E_UNSUPPORTED;
    }
    const long long blocks = (long long)p.tilesX * p.tilesY * p.NB * B;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        ssm::set_error("conv: grid of %lld workgroups out of range", blocks);
        return SSM_E_ARG;
    }
    hipLaunchKernelGGL(conv_mfma_kernel<C>, dim3((unsigned)blocks), dim3(256), C::LDS_BYTES, st, p);
    return ssm::check_launch("ssm_conv2d_fwd");
}

template <class C>
void cfg_dims(int *BN, int *CK) {
    *BN = C::BN;
    *CK = C::CK;
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

}  // namespace

extern "C" int ssm_conv_config(int k, int Cout, int B, int H, int W, int pool, int *BN, int *CK) {
    const int kind = pick_kind(k, Cout, B, H, W, pool);
    switch (kind) {
        case K7: cfg_dims<CfgK7>(BN, CK); break;
        case K5: cfg_dims<CfgK5>(BN, CK); break;
        case K3N32: cfg_dims<CfgK3N32>(BN, CK); break;
        case K3N64: cfg_dims<CfgK3N64>(BN, CK); break;
        case K3N128: cfg_dims<CfgK3N128>(BN, CK); break;
        case K3N128S: cfg_dims<CfgK3N128S>(BN, CK); break;
        case K3N64T: cfg_dims<CfgK3N64T>(BN, CK); break;
        case K3N32T: cfg_dims<CfgK3N32T>(BN, CK); break;
        default:
            ssm::set_error("conv: kernel size %d unsupported (3, 5, 7 are)", k);
            return SSM_E_UNSUPPORTED;
    }
    return SSM_OK;
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
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       bias, wp, bp, Cout, Cin, CinP, k * k, BN, total, nbias);
    return ssm::check_launch("ssm_pack_weights");
}

extern "C" int ssm_conv2d_fwd(ssm_view x1, int C1, ssm_view x2, int C2, const float *w_packed,
                              const float *bias_packed, ssm_view y, ssm_view pool, int B, int H, int W, int Cout,
                              int k, float slope, int flags, void *stream) {
    int BN = 0, CK = 0;
    const int rc = ssm_conv_config(k, Cout, B, H, W, pool.ptr ? 1 : 0, &BN, &CK);
    if (rc != SSM_OK) return rc;
    SSM_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && C1 > 0 && C2 >= 0, "conv: bad sizes");
    SSM_REQUIRE(x1.ptr && y.ptr && w_packed && bias_packed, "conv: null pointer");
    SSM_REQUIRE(C1 % CK == 0 && C2 % CK == 0, "conv: channel counts (%d,%d) must be multiples of %d", C1, C2, CK);
    SSM_REQUIRE(ssm::aligned16(x1.ptr) && x1.sh % 4 == 0 && x1.sc % 4 == 0 && x1.sb % 4 == 0,
                "conv: input 1 is not a padded-plane view (16-byte alignment)");
    SSM_REQUIRE(x1.sh >= W + 2 * SSM_PADX, "conv: input 1 row stride %d leaves no zero frame for W=%d", x1.sh, W);
    SSM_REQUIRE(ssm::aligned16(w_packed), "conv: packed filter must be 16-byte aligned");
    if (C2 > 0) {
        SSM_REQUIRE(x2.ptr && ssm::aligned16(x2.ptr) && x2.sb % 4 == 0, "conv: input 2 is not a padded-plane view");
        SSM_REQUIRE(x2.sh == x1.sh && x2.sc == x1.sc, "conv: cat sources must share row/channel strides");
    }
    SSM_REQUIRE((long long)CK * x1.sc < 0x7fffffffLL, "conv: channel stride too large");
    if (pool.ptr) SSM_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv: fused pool needs even H, W");

    ConvParams p;
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
    p.pool = pool.ptr;
    p.psb = pool.sb;
    p.psc = pool.sc;
    p.psh = pool.sh;
    p.H = H;
    p.W = W;
    p.Cout = Cout;
    p.slope = slope;
    p.lrelu = (flags & SSM_FLAG_LRELU) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    switch (pick_kind(k, Cout, B, H, W, pool.ptr ? 1 : 0)) {
        case K7: return launch<CfgK7>(p, B, st);
        case K5: return launch<CfgK5>(p, B, st);
        case K3N32: return launch<CfgK3N32>(p, B, st);
        case K3N64: return launch<CfgK3N64>(p, B, st);
        case K3N128: return launch<CfgK3N128>(p, B, st);
        case K3N128S: return launch<CfgK3N128S>(p, B, st);
        case K3N64T: return launch<CfgK3N64T>(p, B, st);
        case K3N32T: return launch<CfgK3N32T>(p, B, st);
    }
    return SSM_E_UNSUPPORTED;
}
