// Weight gradient of the 3x3 layers in the WINOGRAD DOMAIN, F(2x2,3x3), fp32 throughout (training step, SURVEY 8f-1 / BASELINE
// config 3: what autograd computes for nn.Conv2d.weight of scripts/models/layers.py:21-33 under scripts/main.py:138-197).
//
// The forward of such a layer is  Y_t = A^T [ sum_ci U (.) V_t ] A  per 2x2 output tile t, with U = G g G^T (4x4 per (co, ci)) and
// V_t = B^T d_t B (the 4x4 input patch of the tile).  Everything is linear in g, so with dM_t = A dY_t A^T (a 2x2 block of dZ spread
// over the 16 frequencies)
//       dU_f[co][ci] = sum_tiles dM_f[co][t] * V_f[ci][t]          16 GEMMs, M = couts, N = cins, K = tiles      (this kernel)
//       dg[co][ci]   = G^T dU[co][ci] G                             4x4 -> 3x3, once per (co, ci)                 (finishing launch)
// 16 multiplies per 2x2 tile instead of the direct form's 36 (ssm_bwd.hip wgrad_mfma_kernel, which stays for 7x7 / 5x5, for the
// small maps and as mode f32) - the same 16/36 the forward and the data gradients of the f32w training plan already run at.
//
// GEMM on v_mfma_f32_32x32x2_f32 (k = 2 tiles per instruction).  A workgroup of 4 waves owns 32*MB couts x 32*NB cins; WAVE w OWNS
// ROW-FREQUENCY w (the four frequencies f = 4 w + j) of that block: 4 x MB x NB accumulators of 16 registers (MB = NB = 2: 256).
// A wave transforms ONLY its own row-frequency - the row pass of B^T / A picks one combination of two raw rows - straight from the raw
// rows in LDS into the MFMA operand registers: lane (c = lane & 31, h = lane >> 5) holds channel c's values for the four tiles
// 4 h .. 4 h + 3 of a group of 8 tiles, which IS the operand layout of four k-steps (MFMA m takes tile m from the lower half-wave and
// tile 4 + m from the upper one).  No transformed tensor is ever written: per group of 8 tiles a wave issues 64 MFMAs (MB = NB = 2)
// for ~90 vector and 24 LDS instructions.
//   step  = (batch entry, pair of tile rows, group of 8 tiles): 4 output rows x 16 pixels; staged raw: 6 input rows x 24 columns of
//           every cin of the block (columns 16 g - 4 .. 16 g + 19: the halo on 16-byte pieces) and 4 rows x 16 pixels of dZ.
//           Channel strides are an odd number of 16-byte units (37 / 17): the 32 channels of a half-wave hit different bank groups.
//   steps are dealt round-robin over gridDim.z (neighbouring workgroups walk neighbouring groups: shared halos meet in L2); the next
//           step's global loads are issued before the MFMAs of the current one (register prefetch, as in wgrad_mfma_kernel).
//   partial sums are added to dU [16][Cout][cin_total] with fp32 atomics; the finishing launch turns dU into dW, adds it to the OIHW
//           gradient and leaves dU zeroed for the next step.  The bias gradient is dM_(1,1) summed over tiles: wave 1 of the workgroups
//           with blockIdx.x == 0 carries it.
// Signs: A's last row is (0, -1); the kernel computes that row / column with +1 and the finishing launch flips the sign of the
// frequencies with exactly one index equal to 3.
#include "ssm_common.h"

#include <cstdlib>

// Tuning builds (`make wwalt WWTAG=a1 WWFLAGS=-DWW_ABL=1`): compile-time ablations of the step loop - 1: no global fetch / LDS commit after
// the first step, 2: no MFMAs (operands kept alive), 4: no transforms (raw values as operands), 8: no barrier.  Results are wrong.
#ifndef WW_ABL
#define WW_ABL 0
#endif

namespace {

typedef float ww_f16 __attribute__((ext_vector_type(16)));
typedef float ww_f4 __attribute__((ext_vector_type(4)));

template <int MB_, int NB_, int OCC_>
struct WwCfg {
    static constexpr int MB = MB_, NB = NB_, OCC = OCC_;
    static constexpr int CO = 32 * MB, CI = 32 * NB;
    static constexpr int XCHQ = 37, DZCHQ = 17;              // 16-byte units per staged channel: 6 rows x 6 (+1), 4 rows x 4 (+1)
    static constexpr int X_FLOATS = 4 * CI * XCHQ, DZ_FLOATS = 4 * CO * DZCHQ;
};

struct WwParams {
    ssm_view x, dz;
    float *du, *db;
    int B, Cin, Cout, H, W, cin_total, ci_offset;
    int PR, GPR, TR;          // pairs of tile rows, groups of 8 tiles per tile row, tile rows
    int nsteps;
    float inv_gpr, inv_pr;
};

template <class C>
__global__ __launch_bounds__(256, C::OCC) void wgradw_kernel(const WwParams p) {
    constexpr int MB = C::MB, NB = C::NB;
    constexpr int BUF = C::X_FLOATS + C::DZ_FLOATS;          // one staging buffer: x tile, dZ tile
    extern __shared__ __attribute__((aligned(16))) float ww_lds[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int fi = tid >> 6;                                  // this wave's row-frequency
    const int ci0 = blockIdx.x * C::CI, co0 = blockIdx.y * C::CO;
    const int H = p.H, W = p.W;

    // row pass of this wave's row-frequency: V row = x[ra] + sgn * x[rb]   (dM: below; signs: see the header)
    const int ra = fi == 0 ? 0 : (fi == 2 ? 2 : 1);
    const int rb = fi == 0 ? 2 : (fi == 1 ? 2 : (fi == 2 ? 1 : 3));
    const float sgn = fi == 1 ? 1.f : -1.f;

    ww_f16 acc[MB][NB][4];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][j][r] = 0.f;
    float bsum[MB];
#pragma unroll
    for (int a = 0; a < MB; ++a) bsum[a] = 0.f;
    const bool do_bias = p.db != nullptr && blockIdx.x == 0 && fi == 1;

    // ---- what this thread stages: a FIXED 16-byte position of the staged tile for every XSTEP-th / 16th channel - offsets affine in the
    // loop counter (nothing per item is kept in registers: the accumulators own the register file).  Pieces outside the image - the
    // convolution's zero padding, channels past Cin / Cout - are fetched from a piece of the planes' zero frame instead: no masks.
    // Addresses are a uniform base + a 32-bit byte offset per lane (the launcher checks the views span < 2 GiB).
    constexpr int XSTEP = 7, NXI = (C::CI + XSTEP - 1) / XSTEP;          // 36 positions x 7 channel lanes = 252 threads
    const int xpos = tid % 36, xcl = tid / 36;                            // (threads 252..255: xcl = 7, no x items)
    const int xr = xpos / 6, xqd = xpos - 6 * xr;
    const int zpos = tid & 15, zcl = tid >> 4;                            // 16 positions x 16 channel lanes
    const int zr = zpos >> 2, zqd = zpos & 3;
    constexpr int NZI = C::CO / 16;
    const char *xbase = reinterpret_cast<const char *>(p.x.ptr - 2 * p.x.sh);           // frame row -2 of the first plane: 16 bytes of zeros
    const char *zbase = reinterpret_cast<const char *>(p.dz.ptr - 2 * p.dz.sh);
    const unsigned xoff0 = 4u * (unsigned)((ci0 + xcl) * (int)p.x.sc + (xr + 1) * p.x.sh + 4 * xqd - 4);
    const unsigned zoff0 = 4u * (unsigned)((co0 + zcl) * (int)p.dz.sc + (zr + 2) * p.dz.sh + 4 * zqd);
    const int xl0 = 4 * (xcl * C::XCHQ + xpos), zl0 = C::X_FLOATS + 4 * (zcl * C::DZCHQ + zpos);
    const unsigned xcs = 4u * XSTEP * (unsigned)p.x.sc, zcs = 4u * 16u * (unsigned)p.dz.sc;

    ww_f4 px[NXI], pz[NZI];
    auto prefetch = [&](int s) {
        // s -> (batch entry, pair of tile rows, group) without integer division
        int t = (int)(((float)s + 0.5f) * p.inv_gpr), g = s - t * p.GPR;
        if (g < 0) --t, g += p.GPR;                    // (the reciprocal is off by at most one: fix up)
        if (g >= p.GPR) ++t, g -= p.GPR;
        int b = (int)(((float)t + 0.5f) * p.inv_pr), pr = t - b * p.PR;
        if (pr < 0) --b, pr += p.PR;
        if (pr >= p.PR) ++b, pr -= p.PR;
        const int y0 = 4 * pr, x0 = 16 * g;
        const int ry = y0 - 1 + xr, xc0 = x0 - 4 + 4 * xqd;
        const bool xin = xcl < XSTEP && ry >= 0 && ry < H && xc0 >= 0 && xc0 < W;
        const unsigned xo = xoff0 + 4u * (unsigned)(b * (int)p.x.sb + y0 * p.x.sh + x0);
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int c = xcl + XSTEP * i;
            const bool ok = xin && c < C::CI && ci0 + c < p.Cin;
            px[i] = *reinterpret_cast<const ww_f4 *>(xbase + (ok ? xo + i * xcs : 0u));
        }
        const bool zin = y0 + zr < H && x0 + 4 * zqd < W;
        const unsigned zo = zoff0 + 4u * (unsigned)(b * (int)p.dz.sb + y0 * p.dz.sh + x0);
#pragma unroll
        for (int i = 0; i < NZI; ++i) {
            const bool ok = zin && co0 + zcl + 16 * i < p.Cout;
            pz[i] = *reinterpret_cast<const ww_f4 *>(zbase + (ok ? zo + i * zcs : 0u));
        }
    };
    auto commit = [&](float *buf) {
#pragma unroll
        for (int i = 0; i < NXI; ++i)
            if (xcl < XSTEP && xcl + XSTEP * i < C::CI) *reinterpret_cast<ww_f4 *>(buf + xl0 + i * (4 * XSTEP * C::XCHQ)) = px[i];
#pragma unroll
        for (int i = 0; i < NZI; ++i) *reinterpret_cast<ww_f4 *>(buf + zl0 + i * (4 * 16 * C::DZCHQ)) = pz[i];
    };

    // ---- this lane's operand sources inside a staged buffer (floats), raw rows -> MFMA operands.
    // dM row of row-frequency fi = dz[row za] + qz * dz[row zb]: (r0, +0), (r0, +r1), (r0, -r1), (r1, +0) - the "+0" rows are read from 8
    // zeroed floats behind the two buffers (every lane the same address: a broadcast), so the row pass is one FMA per value for every wave.
    constexpr int ZERO_AT = 2 * BUF;
    if (tid < 8) ww_lds[ZERO_AT + tid] = 0.f;
    const int xlane = 4 * (l31 * C::XCHQ) + 8 * half + 3;                       // + (2 u + row) * 24 + 128 * XCHQ * nb
    const int zlane = C::X_FLOATS + 4 * (l31 * C::DZCHQ) + 8 * half;            // + (2 u + row) * 16 + 128 * DZCHQ * mb
    const int za_row = fi == 3 ? 16 : 0;                                        // first dz row of the combination (floats)
    const bool zb_zero = fi == 0 || fi == 3;
    const float qz = fi == 2 ? -1.f : 1.f;
    struct RawV {
        float a0, b0, a9, b9;
        ww_f4 a1, a5, b1, b5;
    };
    struct RawA {
        ww_f4 a0, a1, b0, b1;
    };
    auto load_v = [&](const float *buf, int u, int b, RawV &r) {
        const float *xa = buf + xlane + b * (128 * C::XCHQ) + (2 * u + ra) * 24;
        const float *xb = buf + xlane + b * (128 * C::XCHQ) + (2 * u + rb) * 24;
        r.a0 = xa[0], r.b0 = xb[0], r.a9 = xa[9], r.b9 = xb[9];
        r.a1 = *reinterpret_cast<const ww_f4 *>(xa + 1), r.a5 = *reinterpret_cast<const ww_f4 *>(xa + 5);
        r.b1 = *reinterpret_cast<const ww_f4 *>(xb + 1), r.b5 = *reinterpret_cast<const ww_f4 *>(xb + 5);
    };
    auto load_a = [&](const float *buf, int cur_off, int u, int a, RawA &r) {
        const float *z0 = buf + zlane + a * (128 * C::DZCHQ) + (2 * u) * 16 + za_row;
        const float *z1 = zb_zero ? ww_lds + ZERO_AT : buf + zlane + a * (128 * C::DZCHQ) + (2 * u) * 16 + 16;
        r.a0 = *reinterpret_cast<const ww_f4 *>(z0), r.a1 = *reinterpret_cast<const ww_f4 *>(z0 + 4);
        r.b0 = *reinterpret_cast<const ww_f4 *>(z1), r.b1 = *reinterpret_cast<const ww_f4 *>(z1 + 4);
    };
    // the transforms in PIECES of 4-5 vector instructions, dealt between the MFMAs of the previous phase (pieces in order)
    auto xform_v_piece = [&](int k, const RawV &r, float (&d)[10], float (&v)[4][4]) {          // 6 pieces; v[tile m][column-frequency j]
#if WW_ABL & 4
        if (k == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m) v[m][0] = r.a1[m], v[m][1] = r.a5[m], v[m][2] = r.b1[m], v[m][3] = r.b5[m];
        }
        return;
#endif
        if (k == 0) {
            d[0] = r.a0 + sgn * r.b0;
#pragma unroll
            for (int q = 0; q < 4; ++q) d[1 + q] = r.a1[q] + sgn * r.b1[q];
        } else if (k == 1) {
            d[9] = r.a9 + sgn * r.b9;
#pragma unroll
            for (int q = 0; q < 4; ++q) d[5 + q] = r.a5[q] + sgn * r.b5[q];
        } else if (k < 6) {
            const int m = k - 2;
            v[m][0] = d[2 * m] - d[2 * m + 2];
            v[m][1] = d[2 * m + 1] + d[2 * m + 2];
            v[m][2] = d[2 * m + 2] - d[2 * m + 1];
            v[m][3] = d[2 * m + 1] - d[2 * m + 3];
        }
    };
    auto xform_a_piece = [&](int k, const RawA &r, float (&e)[8], float (&am)[4][4], float &bs) {          // 4 pieces
#if WW_ABL & 4
        if (k == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m) am[m][0] = r.a0[m], am[m][1] = r.a1[m], am[m][2] = r.b0[m], am[m][3] = r.b1[m];
        }
        return;
#endif
        if (k == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) e[q] = r.a0[q] + qz * r.b0[q];
        } else if (k == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) e[4 + q] = r.a1[q] + qz * r.b1[q];
        } else if (k < 4) {
#pragma unroll
            for (int m = 2 * (k - 2); m < 2 * (k - 2) + 2; ++m) {
                am[m][0] = e[2 * m];
                am[m][1] = e[2 * m] + e[2 * m + 1];
                am[m][2] = e[2 * m] - e[2 * m + 1];
                am[m][3] = e[2 * m + 1];
            }
            if (do_bias) bs += (am[2 * (k - 2)][1] + am[2 * (k - 2) + 1][1]);
        }
    };

    // ---- steps.  Two staging buffers: while a step's MFMAs read one, the next step's tiles (fetched into registers at the top of the
    // step) are written to the other behind them - ONE barrier per step.  Inside a step the work is 2 NB phases (tile row u of the pair,
    // cin block b) of 16 MB MFMAs in 8 chunks; the raw rows of phase p + 2 are requested at the top of phase p and those of phase p + 1
    // transformed piece by piece between the chunks (sched_barrier fences keep the pieces where they are put): a wave - alone on its
    // SIMD - waits for LDS once per step, and its vector instructions issue in the shadow of the matrix pipe.
    constexpr int P = 2 * NB;
    int s = blockIdx.z;
    if (s < p.nsteps) {
        prefetch(s);
        commit(ww_lds);
    }
    __syncthreads();
    for (int it = 0; s < p.nsteps; s += gridDim.z, ++it) {
        const float *cur = ww_lds + (it & 1) * BUF;
        const bool more = !(WW_ABL & 1) && s + (int)gridDim.z < p.nsteps;
        if (more) prefetch(s + gridDim.z);
        RawV rv[2];
        RawA rA[2][MB];
        float am[2][MB][4][4], vop[2][4][4], dd[10], ee[MB][8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int a = 0; a < MB; ++a) load_a(cur, 0, u, a, rA[u][a]);
        load_v(cur, 0, 0, rv[0]);
        load_v(cur, P > 2 ? 0 : 1, P > 2 ? 1 : 0, rv[1]);          // phase 1
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) xform_a_piece(k, rA[0][a], ee[a], am[0][a], bsum[a]);
#pragma unroll
        for (int k = 0; k < 6; ++k) xform_v_piece(k, rv[0], dd, vop[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ph = 0; ph < P; ++ph) {
            const int u = ph / NB, b = ph % NB;
            const bool next_v = ph + 1 < P, next_a = next_v && (ph + 1) % NB == 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int m = c >> 1, j0 = 2 * (c & 1);
#pragma unroll
                for (int j = j0; j < j0 + 2; ++j)
#pragma unroll
                    for (int a = 0; a < MB; ++a)
#if WW_ABL & 2
                        asm volatile("" ::"v"(am[u][a][m][j]), "v"(vop[ph & 1][m][j]));
#else
                        acc[a][b][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(am[u][a][m][j], vop[ph & 1][m][j], acc[a][b][j], 0, 0, 0);
#endif
                // behind the chunk: a piece of the next phase's transforms
                if (next_v && c < 6) xform_v_piece(c, rv[(ph + 1) & 1], dd, vop[(ph + 1) & 1]);
                if (next_a) {
                    if (MB == 2) xform_a_piece(c & 3, rA[1][c >> 2], ee[c >> 2], am[1][c >> 2], bsum[c >> 2]);
                    else if (c < 4) xform_a_piece(c, rA[1][0], ee[0], am[1][0], bsum[0]);
                }
                // the raw rows of phase ph + 2 (into the registers phase ph's transform emptied a phase ago)
                if (c == 0 && ph + 2 < P) load_v(cur, (ph + 2) / NB, (ph + 2) % NB, rv[ph & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) commit(ww_lds + ((it + 1) & 1) * BUF);
#if !(WW_ABL & 8)
        __syncthreads();
#endif
    }

    // ---- partial sums -> dU[f][co][ci_offset + ci]
    const bool whole = co0 + C::CO <= p.Cout;
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int ci = ci0 + 32 * b + l31;
            if (ci >= p.Cin) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float *base = p.du + ((long long)(4 * fi + j) * p.Cout + co0 + 32 * a + 4 * half) * p.cin_total + p.ci_offset + ci;
                if (whole) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) atomicAdd(base + (long long)((r & 3) + 8 * (r >> 2)) * p.cin_total, acc[a][b][j][r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = co0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (co < p.Cout) atomicAdd(base + (long long)((r & 3) + 8 * (r >> 2)) * p.cin_total, acc[a][b][j][r]);
                    }
                }
            }
        }
    if (do_bias) {
#pragma unroll
        for (int a = 0; a < MB; ++a) {
            const int co = co0 + 32 * a + l31;
            if (co < p.Cout) atomicAdd(p.db + co, bsum[a]);
        }
    }
}

// dW[co][ci][3][3] += G^T dU[.][co][ci] G with G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]] (the sign convention of wgradw_kernel
// folded in), times `scale`; dU is left ZEROED for the next step's atomics.  One thread per (co, ci) of one job; blockIdx.y = job.
struct WwFinishJob {
    float *du, *dw;
    int n;              // Cout * cin_total
    int pad_;
};

__global__ __launch_bounds__(256) void wgradw_finish_kernel(const WwFinishJob *__restrict__ jobs, float scale) {
    const WwFinishJob jb = jobs[blockIdx.y];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= jb.n) return;
    float u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float *q = jb.du + (long long)(4 * i + j) * jb.n + e;
            const float v = *q;
            *q = 0.f;
            u[i][j] = ((i == 3) != (j == 3)) ? -v : v;
        }
    float t[3][4];                                   // G^T u: rows a = 0..2
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float hs = 0.5f * (u[1][j] + u[2][j]), hd = 0.5f * (u[1][j] - u[2][j]);
        t[0][j] = u[0][j] + hs;
        t[1][j] = hd;
        t[2][j] = hs + u[3][j];
    }
    float *w = jb.dw + (long long)e * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float hs = 0.5f * (t[a][1] + t[a][2]), hd = 0.5f * (t[a][1] - t[a][2]);
        w[3 * a + 0] += scale * (t[a][0] + hs);
        w[3 * a + 1] += scale * hd;
        w[3 * a + 2] += scale * (hs + t[a][3]);
    }
}

template <class C>
int ww_launch(const WwParams &p, int target, hipStream_t st) {
    const int gx = (p.Cin + C::CI - 1) / C::CI, gy = (p.Cout + C::CO - 1) / C::CO;
    int split = target / (gx * gy);
    if (split > p.nsteps) split = p.nsteps;
    if (split < 1) split = 1;
    if (split > 65535) split = 65535;
    constexpr int lds_bytes = 2 * 4 * (C::X_FLOATS + C::DZ_FLOATS) + 32;          // two staging buffers + 8 zeroed floats
    static std::atomic<uint64_t> lds_reserved{0};          // one bit per device: the attribute is per (kernel, device)
    const hipError_t attr_rc = ssm::reserve_lds(lds_reserved, (const void *)wgradw_kernel<C>, lds_bytes);
    if (attr_rc != hipSuccess) {
        ssm::set_error("wgrad_wino: cannot reserve %d bytes of LDS: %s", lds_bytes, hipGetErrorString(attr_rc));
        return SSM_E_LAUNCH;
    }
    SSM_LAUNCH(wgradw_kernel<C>, dim3(gx, gy, split), dim3(256), lds_bytes, st, p);
    return ssm::check_launch("ssm_conv2d_wgrad_wino");
}

}  // namespace

extern "C" int ssm_wgrad_wino_supported(int Cin, int Cout, int H, int W, int k) {
    // the Winograd domain pays where K = tiles is long against the 16 x Cout x Cin partial sums every workgroup adds: maps of 40+ pixels
    return k == 3 && Cin >= 32 && Cout >= 32 && H >= 40 && W >= 40;
}

extern "C" long long ssm_wgrad_wino_scratch_floats(int Cout, int cin_total) { return 16LL * Cout * cin_total; }

extern "C" int ssm_conv2d_wgrad_wino(ssm_view x, ssm_view dz, float *du, float *db_acc, int B, int Cin, int Cout, int H, int W, int cin_total,
                                     int ci_offset, void *stream) {
    SSM_REQUIRE(x.ptr && dz.ptr && du && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "wgrad_wino: bad arguments");
    SSM_REQUIRE(ci_offset >= 0 && ci_offset + Cin <= cin_total, "wgrad_wino: channel range [%d,%d) outside the filter's %d inputs", ci_offset,
                ci_offset + Cin, cin_total);
    SSM_REQUIRE(x.sh >= W + 2 * SSM_PADX && dz.sh >= W + 3, "wgrad_wino: x and dz must be padded-plane views (zero frame)");
    SSM_REQUIRE(ssm::aligned16(x.ptr) && ssm::aligned16(dz.ptr) && x.sh % 4 == 0 && x.sc % 4 == 0 && x.sb % 4 == 0 && dz.sh % 4 == 0 &&
                    dz.sc % 4 == 0 && dz.sb % 4 == 0,
                "wgrad_wino: views must be 16-byte aligned with strides that are multiples of 4 floats");
    // (the kernel addresses both tensors with 32-bit byte offsets from their first plane)
    SSM_REQUIRE(((long long)B * x.sb + (long long)(Cin + 64) * x.sc) * 4 < 0x7fffffffLL && ((long long)B * dz.sb + (long long)(Cout + 64) * dz.sc) * 4 < 0x7fffffffLL,
                "wgrad_wino: tensors must span less than 2 GiB");
    SSM_REQUIRE(x.sb >= 0 && dz.sb >= 0 && x.sc > 0 && dz.sc > 0, "wgrad_wino: negative strides");
    WwParams p;
    p.x = x;
    p.dz = dz;
    p.du = du;
    p.db = db_acc;
    p.B = B, p.Cin = Cin, p.Cout = Cout, p.H = H, p.W = W, p.cin_total = cin_total, p.ci_offset = ci_offset;
    p.TR = (H + 1) / 2;
    p.PR = (p.TR + 1) / 2;
    p.GPR = ((W + 1) / 2 + 7) / 8;
    const long long ns = (long long)B * p.PR * p.GPR;
    SSM_REQUIRE(ns < (1LL << 20), "wgrad_wino: problem too large for one launch (%lld steps)", ns);
    p.nsteps = (int)ns;
    p.inv_gpr = 1.f / (float)p.GPR;
    p.inv_pr = 1.f / (float)p.PR;
    // workgroups of one launch: every workgroup adds 16 x 32 MB x 32 NB partial sums with atomics whatever its share of K, and the
    // launch shares the chip with the data-gradient stream: about half a round of the 256 CUs ($SSM_WGRADW_TARGET: tuning knob)
    static const int target_env = [] {
        const char *e = getenv("SSM_WGRADW_TARGET");
        return e ? atoi(e) : 0;
    }();
    const int target = target_env > 0 ? target_env : 128;
    hipStream_t st = (hipStream_t)stream;
    // (r6, measured and not kept: the 32-cout x 64-cin form at two workgroups per CU for every layer - twice the workgroups, each transforming
    // V for half the couts: 0.202 against 0.197 ms at 128, 0.233 against 0.142 at 256 on conv8a, profiles/r21c_ww_cfg12.txt)
    if (Cout > 32) return ww_launch<WwCfg<2, 2, 1>>(p, target, st);
    if (Cin > 32) return ww_launch<WwCfg<1, 2, 1>>(p, target, st);
    return ww_launch<WwCfg<1, 1, 2>>(p, 2 * target, st);
}

// jobs: DEVICE array of n_jobs {du, dw, n, pad} records (struct layout of ssm_wgradw_finish_job in ssm_hip.h); max_n = the largest n.
extern "C" int ssm_wgrad_wino_finish(const void *jobs_dev, int n_jobs, int max_n, float scale, void *stream) {
    SSM_REQUIRE(jobs_dev && n_jobs > 0 && n_jobs <= 65535 && max_n > 0, "wgrad_wino_finish: bad arguments");
    SSM_LAUNCH(wgradw_finish_kernel, dim3((max_n + 255) / 256, n_jobs), dim3(256), 0, (hipStream_t)stream,
                       (const WwFinishJob *)jobs_dev, scale);
    return ssm::check_launch("ssm_wgrad_wino_finish");
}
