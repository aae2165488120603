// Every fp32 filter of a U-Net repacked by ONE launch from a device job table (training: the parameters change each optimizer
// step; the per-layer pack entry points cost ~100 launches per U-Net and step, and the data-gradient filters another ~190 torch
// launches for flip / permute / contiguous).  A job names the OIHW parameter, the packed destination and the form:
//   direct  (ssm_pack_weights,        csrc/ssm_conv.hip)    [Cout/BN][CinP][k*k][BN]
//   wino    (ssm_wino_pack_weights,   csrc/ssm_wino.hip)    [Cout/BN][Cin][4][BN][4]        U = G g G^T, F(2x2,3x3)
//   wino1d  (ssm_wino1d_pack_weights, csrc/ssm_wino1d.hip)  [Cout/BN][CinP][k][2][BN][4]    U[ky] = G g[ky], F(2,7) / F(4,5)
//   wino4   (ssm_wino4_pack_weights,  csrc/ssm_wino4.hip)   [Cout/32][Cin][9][32][4]        U = G g G^T, F(4x4,3x3)
//   wino5   (ssm_wino5_pack_weights,  csrc/ssm_wino5.hip)   [Cout/32][CinP/4][16][4][32][4]  U = G g G^T, F(4x4,5x5)
//   wino7   (ssm_wino7_pack_weights,  csrc/ssm_wino7.hip)   [Cout/32][Cin][14][4][32][4]    U_b = G g_b G^T, 2x2 blocks of F(4x4,4x4)
// `transposed` packs the DATA-GRADIENT filter of the forward parameter, W'[ci][co][ky][kx] = W[co][ci][k-1-ky][k-1-kx] (what
// ssm_amd.backward.transposed_filter materialises), read straight from the OIHW tensor; such jobs have no bias (zeros).  Every
// element is computed by the same arithmetic as the per-layer kernels (tests/test_hip_pack_batch.py holds them bit-identical).
#include "ssm_common.h"

#include <type_traits>
#include "ssm_wino5_pack.h"
#include "ssm_wino7_pack.h"

namespace {

struct Src {          // logical filter W'(co, ci, ky, kx) of a job
    const float *w;
    int O, I, k, transposed;       // the parameter is [O][I][k][k]
    __device__ float at(int co, int ci, int ky, int kx) const {
        if (!transposed) return w[(((long long)co * I + ci) * k + ky) * k + kx];
        return w[(((long long)ci * I + co) * k + (k - 1 - ky)) * k + (k - 1 - kx)];
    }
};

typedef float pk_f4 __attribute__((ext_vector_type(4)));

// A thread produces FOUR consecutive packed elements (every form keeps 4 values of one (cout block, cin, ...) adjacent: 4 couts of the
// direct form, the 4 frequencies of a quad in the Winograd forms) and stores them as one 16-byte piece: the filter values they share are
// read once, and the job lookup is paid per quad.  The first version (one element per thread) spent 1.6 ms of a 20 ms training step in
// this kernel (profiles/r7g: 4 launches of 0.41 ms).  The arithmetic per element is unchanged (bit-identical to the per-layer kernels).
__device__ pk_f4 pack_direct4(const Src &s, const ssm_pack32_job &j, long long i) {
    long long r = i;
    const int n = (int)(r % j.BN);          // multiple of 4 (BN is)
    r /= j.BN;
    const int KS2 = j.k * j.k;
    const int tap = (int)(r % KS2);
    r /= KS2;
    const int cin = (int)(r % j.CinP);
    const int nb = (int)(r / j.CinP);
    pk_f4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int co = nb * j.BN + n + q;
        v[q] = (co < j.Cout && cin < j.Cin) ? s.at(co, cin, tap / j.k, tap % j.k) : 0.f;
    }
    return v;
}

__device__ pk_f4 pack_wino_4(const Src &s, const ssm_pack32_job &j, long long i) {      // as wino_pack_kernel; i: e = 0
    long long r = i / 4;
    const int n = (int)(r % j.BN);
    r /= j.BN;
    const int q = (int)(r % 4);
    r /= 4;
    const int cin = (int)(r % j.Cin);
    const int nb = (int)(r / j.Cin);
    const int co = nb * j.BN + n;
    if (co >= j.Cout) return pk_f4{0.f, 0.f, 0.f, 0.f};
    float row[3];      // row q of G g
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g0 = s.at(co, cin, 0, c), g1 = s.at(co, cin, 1, c), g2 = s.at(co, cin, 2, c);
        row[c] = q == 0 ? g0 : (q == 1 ? 0.5f * (g0 + g1 + g2) : (q == 2 ? 0.5f * (g0 - g1 + g2) : g2));
    }
    return pk_f4{row[0], 0.5f * (row[0] + row[1] + row[2]), 0.5f * (row[0] - row[1] + row[2]), row[2]};
}

__device__ pk_f4 pack_wino1d_4(const Src &s, const ssm_pack32_job &j, long long i) {    // as wino1d_pack_kernel; i: e = 0
    long long r = i / 4;
    const int n = (int)(r % j.BN);
    r /= j.BN;
    const int fq = (int)(r % 2);
    r /= 2;
    const int ky = (int)(r % j.k);
    r /= j.k;
    const int cin = (int)(r % j.CinP);
    const int nb = (int)(r / j.CinP);
    const int co = nb * j.BN + n;
    pk_f4 out = {0.f, 0.f, 0.f, 0.f};
    if (co < j.Cout && cin < j.Cin) {
        double g[7];
        for (int kx = 0; kx < j.k; ++kx) g[kx] = (double)s.at(co, cin, ky, kx);
        const double pt[7] = {0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5};
        const double cf[7] = {1.0, -2.0 / 9.0, -2.0 / 9.0, 1.0 / 90.0, 1.0 / 90.0, 32.0 / 45.0, 32.0 / 45.0};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = 4 * fq + e;
            double val = 0.0;
            if (f == 7) {
                val = g[j.k - 1];
            } else {
                double pw = 1.0;
                for (int kx = 0; kx < j.k; ++kx) {
                    val += pw * g[kx];
                    pw *= pt[f];
                }
                val *= cf[f];
            }
            out[e] = (float)val;
        }
    }
    return out;
}

struct W4G {
    double g[6][3];
};
constexpr W4G w4g_table() {
    W4G t{};
    const double pt[5] = {0.0, 0.625, -0.625, 1.6, -1.6};
    for (int f5 = 0; f5 < 5; ++f5) {
        double nrm = 1.0;
        for (int o = 0; o < 5; ++o)
            if (o != f5) nrm *= pt[f5] - pt[o];
        t.g[f5][0] = 1.0 / nrm;
        t.g[f5][1] = pt[f5] / nrm;
        t.g[f5][2] = pt[f5] * pt[f5] / nrm;
    }
    t.g[5][0] = t.g[5][1] = 0.0;
    t.g[5][2] = 1.0;
    return t;
}

__device__ pk_f4 pack_wino4_4(const Src &s, const ssm_pack32_job &j, long long idx) {   // as wino4_pack_kernel (points 0, +-5/8, +-8/5, inf); idx: e = 0
    long long r = idx / 4;
    const int n = (int)(r % 32);
    r /= 32;
    const int fq = (int)(r % 9);
    r /= 9;
    const int cin = (int)(r % j.Cin);
    const int nb = (int)(r / j.Cin);
    const int co = nb * 32 + n;
    pk_f4 out = {0.f, 0.f, 0.f, 0.f};
    if (co < j.Cout) {
        // G = [1 p p^2] / prod_{q != p}(p - q) per point, [0 0 1] for the point at infinity: a compile-time table (r6: every thread used to
        // evaluate it - five fp64 divisions and three more per row - for its four values; the step's repack of the F(4x4) filters took 96 us)
        constexpr W4G T = w4g_table();
        const double (&G)[6][3] = T.g;
        double g[3][3];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) g[a][c] = (double)s.at(co, cin, a, c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = 4 * fq + e, fi = (f % 18) / 3, fj = 3 * (f / 18) + f % 3;          // f = w4_freq(fi, fj) of csrc/ssm_wino4.hip
            double val = 0.0;
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < 3; ++c) val += G[fi][a] * g[a][c] * G[fj][c];
            out[e] = (float)val;
        }
    }
    return out;
}

__global__ void pack32_batch_kernel(const ssm_pack32_job *__restrict__ jobs, int njobs, long long total) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;          // the first of this thread's four elements
    if (i >= total) return;
    int lo = 0, hi = njobs - 1;          // the job whose [first, first + count) holds i (every job starts at a multiple of 4)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first <= i) lo = mid;
        else hi = mid - 1;
    }
    const ssm_pack32_job j = jobs[lo];
    const long long e = i - j.first;
    const Src s{j.w, j.transposed ? j.Cin : j.Cout, j.transposed ? j.Cout : j.Cin, j.k, j.transposed};
    if (e < j.total) {
        pk_f4 v;
        switch (j.algo) {
            case SSM_PACK_WINO: v = pack_wino_4(s, j, e); break;
            case SSM_PACK_WINO1D: v = pack_wino1d_4(s, j, e); break;
            case SSM_PACK_WINO4: v = pack_wino4_4(s, j, e); break;
            case SSM_PACK_WINO5: {
                float o4[4];
                ssm_w5_pack_quad([&](int co, int ci, int ky, int kx) { return s.at(co, ci, ky, kx); }, j.Cout, j.Cin, j.CinP, e, o4);
                v = pk_f4{o4[0], o4[1], o4[2], o4[3]};
                break;
            }
            case SSM_PACK_WINO7: {
                float o4[4];
                ssm_w7_pack_quad([&](int co, int ci, int ky, int kx) { return s.at(co, ci, ky, kx); }, j.Cout, j.Cin, e, o4);
                v = pk_f4{o4[0], o4[1], o4[2], o4[3]};
                break;
            }
            default: v = pack_direct4(s, j, e); break;
        }
        *reinterpret_cast<pk_f4 *>(j.wp + e) = v;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (e + q < j.nbias) j.bp[e + q] = (j.bias && e + q < j.Cout) ? j.bias[e + q] : 0.f;
}

// ---- F(2x2,3x3) and direct-form 3x3 filters by TILES (r5): the jobs that carry most of a U-Net's bytes -------------------------------------------------
// pack32_batch_kernel gives neighbouring threads neighbouring couts: their nine weights sit Cin x 36 bytes apart, every thread fetches its
// own 36 bytes (0.9 TB/s; 1.15 ms of a 15 ms training step for the four repacks).  Here a workgroup owns a tile of BN couts x 16 input
// channels: it reads the tile as rows of 144 (transposed: BN x 9) CONTIGUOUS floats into LDS and writes the 16 x 4 x BN quads of the tile,
// which are one contiguous 1024 x BN byte run of the packed filter.  Same expressions as pack_wino_4: bit-identical.
// jobs: `first` = first tile of the job, `total` = its tiles = (Cout / BN) x (Cin / 16); Cout % BN == 0, Cin % 16 == 0, k == 3, BN <= 64.
// r6: F(4x4,3x3) jobs too (BN = 32: that form is packed in 32-cout blocks whatever the launch's tile configuration).
__global__ __launch_bounds__(256) void pack32_wino_tiles_kernel(const ssm_pack32_job *__restrict__ jobs, int njobs, long long total) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const long long t = blockIdx.x;
    if (t >= total) return;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first <= t) lo = mid;
        else hi = mid - 1;
    }
    const ssm_pack32_job j = jobs[lo];
    const int tl = (int)(t - j.first), nct = j.Cin / 16, nb = tl / nct, cin0 = (tl - nb * nct) * 16, BN = j.BN;
    const int tid = threadIdx.x;
    // the parameter is [Cout][Cin][3][3], or for a transposed job [Cin][Cout][3][3] (Cout, Cin: the PACKED filter's)
    // rows of 144 (transposed: BN x 9) contiguous floats, 16-byte aligned (Cin, Cout multiples of 16): moved as 16-byte pieces, the row
    // length a compile-time constant per case (r6: one float per thread and a run-time division per element made this launch
    // instruction-bound at 1.9 TB/s)
    auto stage = [&](auto rowlen4_tag) {
        constexpr int RL4 = decltype(rowlen4_tag)::value;          // 16-byte pieces per row
        const int nrows = j.transposed ? 16 : BN;
        for (int idx = tid; idx < nrows * RL4; idx += 256) {
            const int r = idx / RL4, c4 = idx - r * RL4;
            const float *src = j.transposed ? j.w + ((long long)(cin0 + r) * j.Cout + (long long)nb * BN) * 9
                                            : j.w + ((long long)(nb * BN + r) * j.Cin + cin0) * 9;
            reinterpret_cast<pk_f4 *>(wl)[idx] = reinterpret_cast<const pk_f4 *>(src)[c4];
        }
    };
    if (!j.transposed) stage(std::integral_constant<int, 36>{});
    else if (BN == 32) stage(std::integral_constant<int, 72>{});
    else stage(std::integral_constant<int, 144>{});
    __syncthreads();
    if (j.algo == SSM_PACK_DIRECT) {          // direct form [nb][cin][tap][n]: the tile is the run of 16 x 9 x BN floats of (nb, cin0 .. cin0 + 15)
        float *outd = j.wp + ((long long)nb * j.Cin + cin0) * 9 * BN;
        const int nq = BN / 4, shq = BN == 32 ? 3 : 4;          // (BN is 32 or 64: shifts instead of run-time divisions)
        for (int qd = tid; qd < 16 * 9 * nq; qd += 256) {
            const int n4 = (qd & (nq - 1)) * 4, tap = (qd >> shq) % 9, cl = (qd >> shq) / 9;
            const int tp = j.transposed ? 8 - tap : tap;
            pk_f4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (j.transposed ? wl + (cl * BN + n4 + q) * 9 : wl + ((n4 + q) * 16 + cl) * 9)[tp];
            *reinterpret_cast<pk_f4 *>(outd + ((long long)(cl * 9 + tap) * BN + n4)) = v;
        }
        if (tl == 0)
            for (int i = tid; i < j.nbias; i += 256) j.bp[i] = (j.bias && i < j.Cout) ? j.bias[i] : 0.f;
        return;
    }
    if (j.algo == SSM_PACK_WINO4) {          // F(4x4,3x3) [nb][cin][fq][32][4] (BN = 32): the tile is the run of 16 x 9 x 32 quads of (nb, cin0 .. cin0 + 15)
        constexpr W4G T = w4g_table();          // (r6: these jobs were element-wise - every thread fetched its own 36 bytes, 96 us per U-Net and repack)
        float *out4 = j.wp + ((long long)nb * j.Cin + cin0) * (9 * 32 * 4);
        for (int qd = tid; qd < 16 * 9 * 32; qd += 256) {
            const int n = qd & 31, fq = (qd >> 5) % 9, cl = qd / (9 * 32);
            const float *gl = j.transposed ? wl + (cl * BN + n) * 9 : wl + (n * 16 + cl) * 9;
            double g[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) g[a][c] = (double)(j.transposed ? gl[8 - (3 * a + c)] : gl[3 * a + c]);
            pk_f4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {          // the expressions of pack_wino4_4 (bit-identical)
                const int f = 4 * fq + e, fi = (f % 18) / 3, fj = 3 * (f / 18) + f % 3;
                double val = 0.0;
                for (int a = 0; a < 3; ++a)
                    for (int c = 0; c < 3; ++c) val += T.g[fi][a] * g[a][c] * T.g[fj][c];
                o[e] = (float)val;
            }
            *reinterpret_cast<pk_f4 *>(out4 + (long long)qd * 4) = o;
        }
        if (tl == 0)
            for (int i = tid; i < j.nbias; i += 256) j.bp[i] = (j.bias && i < j.Cout) ? j.bias[i] : 0.f;
        return;
    }
    float *out = j.wp + ((long long)nb * j.Cin + cin0) * 4 * BN * 4;
    const int shn = BN == 32 ? 5 : 6;
    for (int qd = tid; qd < 64 * BN; qd += 256) {
        const int n = qd & (BN - 1), q = (qd >> shn) & 3, cl = qd >> (shn + 2);
        const float *g = j.transposed ? wl + (cl * BN + n) * 9 : wl + (n * 16 + cl) * 9;
        float row[3];      // row q of G g
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g0 = j.transposed ? g[8 - c] : g[c], g1 = j.transposed ? g[5 - c] : g[3 + c], g2 = j.transposed ? g[2 - c] : g[6 + c];
            row[c] = q == 0 ? g0 : (q == 1 ? 0.5f * (g0 + g1 + g2) : (q == 2 ? 0.5f * (g0 - g1 + g2) : g2));
        }
        *reinterpret_cast<pk_f4 *>(out + (long long)qd * 4) = pk_f4{row[0], 0.5f * (row[0] + row[1] + row[2]), 0.5f * (row[0] - row[1] + row[2]), row[2]};
    }
    if (tl == 0)
        for (int i = tid; i < j.nbias; i += 256) j.bp[i] = (j.bias && i < j.Cout) ? j.bias[i] : 0.f;
}

}  // namespace

extern "C" int ssm_pack32_wino_tiles_batch(const ssm_pack32_job *jobs_device, int n_jobs, long long total_tiles, int max_bn, void *stream) {
    SSM_REQUIRE(jobs_device && n_jobs > 0 && total_tiles > 0 && total_tiles <= 0x7fffffffLL, "pack32 wino tiles: empty / oversized job table");
    SSM_REQUIRE(max_bn == 32 || max_bn == 64, "pack32 wino tiles: cout blocks of 32 or 64 (got %d)", max_bn);
    SSM_LAUNCH(pack32_wino_tiles_kernel, dim3((unsigned)total_tiles), dim3(256), (size_t)max_bn * 144 * sizeof(float), (hipStream_t)stream,
                       jobs_device, n_jobs, total_tiles);
    return ssm::check_launch("ssm_pack32_wino_tiles_batch");
}

extern "C" int ssm_pack32_weights_batch(const ssm_pack32_job *jobs_device, int n_jobs, long long total_elements, void *stream) {
    SSM_REQUIRE(jobs_device && n_jobs > 0 && total_elements > 0, "pack32 batch: empty job table");
    // (every job's `first`, `total` and BN are multiples of 4 and its packed buffer is 16-byte aligned: a thread stores 4 elements)
    const long long blocks = ((total_elements + 3) / 4 + 255) / 256;
    SSM_REQUIRE(blocks <= 0x7fffffffLL, "pack32 batch: %lld elements out of range", total_elements);
    SSM_LAUNCH(pack32_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, jobs_device, n_jobs, total_elements);
    return ssm::check_launch("ssm_pack32_weights_batch");
}
