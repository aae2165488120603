// Pointwise halves of the recurrent bottleneck (BOTTLENECK=CLSTM|CGRU, reference call sites
// scripts/models/flow_computation.py:73-88,208-211 / flow_interpolation.py:73-88,284-287).
// The gate convolutions run on the MFMA conv kernels, split into an input part (batched over the whole
// sequence) and a hidden-state part (one launch per step); these kernels add the two pre-activations,
// apply the gates and write the new state - as fp32 planes and/or straight into the HL8 tensor the next
// convolution reads.  One thread = one pixel x one 8-channel group.  The cell equations restate the
// published ConvLSTM / ConvGRU cells of SreenivasVRao/ConvGRU-ConvLSTM-PyTorch (an un-vendored submodule of
// the reference: parity UNPINNED).
#include "ssm_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float *vp(const ssm_view &v, int b, int c, int y) {
    return v.ptr + (long long)b * v.sb + (long long)c * v.sc + (long long)y * v.sh;
}

__device__ __forceinline__ int pack4_fp8(float a, float b, float c, float d) {
    const float lim = 448.0f;                     // e4m3fn: beyond 448 -> NaN
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a, -lim, lim), __builtin_amdgcn_fmed3f(b, -lim, lim), 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(c, -lim, lim), __builtin_amdgcn_fmed3f(d, -lim, lim), w, true);
}

// q8 = 0: HL8 (second plane = fp16 lo).  q8 = 1: Q8 form (include/ssm_hip.h): second planes shared by the pair of groups
// (g & ~1, g | 1): even group's = fp8(x) of both, odd group's = fp8(lo * 2^11) of both; the view must start at an even group.
__device__ __forceinline__ void hl8_store(const ssm_hview &v, int b, int g, int y, int x, const float (&o)[8], int q8) {
    h8 hi;
    float lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        hi[e] = (_Float16)o[e];
        lo[e] = o[e] - (float)hi[e];
    }
    char *d = (char *)v.ptr + ((long long)b * v.sb + (long long)g * v.sg + (long long)y * v.sh + x) * 16;
    *reinterpret_cast<h8 *>(d) = hi;
    if (q8) {
        const int odd = g & 1;
        typedef int i2 __attribute__((ext_vector_type(2)));
        char *even_rec = d - odd * v.sg * 16 + v.sp * 16;
        *reinterpret_cast<i2 *>(even_rec + odd * 8) = i2{pack4_fp8(o[0], o[1], o[2], o[3]), pack4_fp8(o[4], o[5], o[6], o[7])};
        *reinterpret_cast<i2 *>(even_rec + v.sg * 16 + odd * 8) =
            i2{pack4_fp8(lo[0] * 2048.f, lo[1] * 2048.f, lo[2] * 2048.f, lo[3] * 2048.f),
               pack4_fp8(lo[4] * 2048.f, lo[5] * 2048.f, lo[6] * 2048.f, lo[7] * 2048.f)};
    } else {
        h8 l16;
#pragma unroll
        for (int e = 0; e < 8; ++e) l16[e] = (_Float16)lo[e];
        *reinterpret_cast<h8 *>(d + v.sp * 16) = l16;
    }
}

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

#define SSM_CELL_INDEX()                                      \
    const int x = blockIdx.x * 64 + threadIdx.x;              \
    const int y = blockIdx.y * 4 + threadIdx.y;               \
    const int G = Hc >> 3;                                    \
    const int b = blockIdx.z / G, g = blockIdx.z - b * G;     \
    if (x >= W || y >= H) return;

// ConvLSTM cell: pre-activations [i | f | o | g] (Hc channels each);
//   c' = sigmoid(f) * c + sigmoid(i) * tanh(g);   h' = sigmoid(o) * tanh(c')
__global__ __launch_bounds__(256) void convlstm_cell_kernel(ssm_view gx, ssm_view gh, ssm_view cprev, ssm_view cnext, ssm_view h32,
                                                            ssm_hview h16, int Hc, int H, int W, int q8) {
    SSM_CELL_INDEX();
    float hv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = g * 8 + e;
        float pi = vp(gx, b, c, y)[x], pf = vp(gx, b, Hc + c, y)[x], po = vp(gx, b, 2 * Hc + c, y)[x], pg = vp(gx, b, 3 * Hc + c, y)[x];
        if (gh.ptr) {
            pi = pi + vp(gh, b, c, y)[x];
            pf = pf + vp(gh, b, Hc + c, y)[x];
            po = po + vp(gh, b, 2 * Hc + c, y)[x];
            pg = pg + vp(gh, b, 3 * Hc + c, y)[x];
        }
        const float cp = cprev.ptr ? vp(cprev, b, c, y)[x] : 0.0f;
        const float cn = sigm(pf) * cp + sigm(pi) * tanhf(pg);
        vp(cnext, b, c, y)[x] = cn;
        hv[e] = sigm(po) * tanhf(cn);
        if (h32.ptr) vp(h32, b, c, y)[x] = hv[e];
    }
    if (h16.ptr) hl8_store(h16, b, g, y, x, hv, q8);
}

// ConvGRU, first half: gates [gamma | beta]; reset = sigmoid(gamma); writes reset * h (input of the candidate conv).
__global__ __launch_bounds__(256) void convgru_reset_kernel(ssm_view gx, ssm_view gh, ssm_view hprev, ssm_view rh32, ssm_hview rh16,
                                                            int Hc, int H, int W, int q8) {
    SSM_CELL_INDEX();
    float rv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = g * 8 + e;
        const float pr = vp(gx, b, c, y)[x] + vp(gh, b, c, y)[x];
        rv[e] = sigm(pr) * vp(hprev, b, c, y)[x];
        if (rh32.ptr) vp(rh32, b, c, y)[x] = rv[e];
    }
    if (rh16.ptr) hl8_store(rh16, b, g, y, x, rv, q8);
}

// ConvGRU, second half: update = sigmoid(beta); h' = (1 - update) * h + update * tanh(candidate pre-activation).
__global__ __launch_bounds__(256) void convgru_update_kernel(ssm_view gx, ssm_view gh, ssm_view cx, ssm_view ch, ssm_view hprev,
                                                             ssm_view h32, ssm_hview h16, int Hc, int H, int W, int q8) {
    SSM_CELL_INDEX();
    float hv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = g * 8 + e;
        float pu = vp(gx, b, Hc + c, y)[x], pc = vp(cx, b, c, y)[x];
        if (gh.ptr) pu = pu + vp(gh, b, Hc + c, y)[x];
        if (ch.ptr) pc = pc + vp(ch, b, c, y)[x];
        const float u = sigm(pu);
        const float hp = hprev.ptr ? vp(hprev, b, c, y)[x] : 0.0f;
        hv[e] = (1.0f - u) * hp + u * tanhf(pc);
        if (h32.ptr) vp(h32, b, c, y)[x] = hv[e];
    }
    if (h16.ptr) hl8_store(h16, b, g, y, x, hv, q8);
}

// ---- adjoints of the cells (training through the recurrent bottleneck; fp32 views, one thread = pixel x 8 channels) ----
// ConvLSTM: given d h' and d c' (NULL = 0) -> d gates [i|f|o|g] and d c.
__global__ __launch_bounds__(256) void convlstm_cell_bwd_kernel(ssm_view gx, ssm_view gh, ssm_view cprev, ssm_view dh, ssm_view dcn,
                                                                ssm_view dg, ssm_view dcp, int Hc, int H, int W) {
    SSM_CELL_INDEX();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = g * 8 + e;
        float pi = vp(gx, b, c, y)[x], pf = vp(gx, b, Hc + c, y)[x], po = vp(gx, b, 2 * Hc + c, y)[x], pg = vp(gx, b, 3 * Hc + c, y)[x];
        if (gh.ptr) {
            pi = pi + vp(gh, b, c, y)[x];
            pf = pf + vp(gh, b, Hc + c, y)[x];
            po = po + vp(gh, b, 2 * Hc + c, y)[x];
            pg = pg + vp(gh, b, 3 * Hc + c, y)[x];
        }
        const float cp = cprev.ptr ? vp(cprev, b, c, y)[x] : 0.0f;
        const float si = sigm(pi), sf = sigm(pf), so = sigm(po), tg = tanhf(pg);
        const float cn = sf * cp + si * tg, tc = tanhf(cn);
        const float dhv = vp(dh, b, c, y)[x];
        const float dct = (dcn.ptr ? vp(dcn, b, c, y)[x] : 0.0f) + dhv * so * (1.0f - tc * tc);
        vp(dg, b, c, y)[x] = dct * tg * si * (1.0f - si);
        vp(dg, b, Hc + c, y)[x] = dct * cp * sf * (1.0f - sf);
        vp(dg, b, 2 * Hc + c, y)[x] = dhv * tc * so * (1.0f - so);
        vp(dg, b, 3 * Hc + c, y)[x] = dct * si * (1.0f - tg * tg);
        vp(dcp, b, c, y)[x] = dct * sf;
    }
}

// ConvGRU reset half: rh = s(gamma) * h.  Given d rh -> d gamma (written to dgates[0:Hc]) and d h.
__global__ __launch_bounds__(256) void convgru_reset_bwd_kernel(ssm_view gates, ssm_view hprev, ssm_view drh, ssm_view dgates, ssm_view dhp,
                                                                int Hc, int H, int W) {
    SSM_CELL_INDEX();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = g * 8 + e;
        const float r = sigm(vp(gates, b, c, y)[x]), h = vp(hprev, b, c, y)[x], d = vp(drh, b, c, y)[x];
        vp(dgates, b, c, y)[x] = d * h * r * (1.0f - r);
        vp(dhp, b, c, y)[x] = d * r;
    }
}

// ConvGRU update half: h' = (1-u) h + u tanh(q), u = s(beta).  Given d h' -> d beta (dgates[Hc:2Hc]), d q and d h.
__global__ __launch_bounds__(256) void convgru_update_bwd_kernel(ssm_view gates, ssm_view cand, ssm_view hprev, ssm_view dhn, ssm_view dgates,
                                                                 ssm_view dcand, ssm_view dhp, int Hc, int H, int W) {
    SSM_CELL_INDEX();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = g * 8 + e;
        const float u = sigm(vp(gates, b, Hc + c, y)[x]), n = tanhf(vp(cand, b, c, y)[x]);
        const float h = hprev.ptr ? vp(hprev, b, c, y)[x] : 0.0f, d = vp(dhn, b, c, y)[x];
        vp(dgates, b, Hc + c, y)[x] = d * (n - h) * u * (1.0f - u);
        vp(dcand, b, c, y)[x] = d * u * (1.0f - n * n);
        if (dhp.ptr) vp(dhp, b, c, y)[x] = d * (1.0f - u);
    }
}

inline dim3 cell_grid(int B, int Hc, int H, int W) { return dim3((W + 63) / 64, (H + 3) / 4, B * (Hc / 8)); }

}  // namespace

#define SSM_CELL_DIMS(what)                                                                                          \
    SSM_REQUIRE(B > 0 && Hc > 0 && Hc % 8 == 0 && H > 0 && W > 0, what ": bad sizes (hidden channels must be a multiple of 8)"); \
    SSM_REQUIRE((long long)B * (Hc / 8) <= 65535, what ": B*Hc too large for one launch")

extern "C" int ssm_convlstm_cell_fwd(ssm_view gates_x, ssm_view gates_h, ssm_view c_prev, ssm_view c_next, ssm_view h_f32,
                                     ssm_hview h_hl8, int B, int Hc, int H, int W, int flags, void *stream) {
    const int q8 = (flags & SSM_FLAG_Q8) ? 1 : 0;
    SSM_REQUIRE(!q8 || Hc % 16 == 0, "convlstm_cell: a Q8 output needs hidden channels in multiples of 16");
    SSM_CELL_DIMS("convlstm_cell");
    SSM_REQUIRE(gates_x.ptr && c_next.ptr && (h_f32.ptr || h_hl8.ptr), "convlstm_cell: null pointer");
    SSM_REQUIRE(!h_hl8.ptr || ssm::aligned16(h_hl8.ptr), "convlstm_cell: HL8 output must be 16-byte aligned");
    SSM_LAUNCH(convlstm_cell_kernel, cell_grid(B, Hc, H, W), dim3(64, 4), 0, (hipStream_t)stream, gates_x, gates_h, c_prev,
                       c_next, h_f32, h_hl8, Hc, H, W, q8);
    return ssm::check_launch("ssm_convlstm_cell_fwd");
}

extern "C" int ssm_convgru_reset_fwd(ssm_view gates_x, ssm_view gates_h, ssm_view h_prev, ssm_view rh_f32, ssm_hview rh_hl8, int B,
                                     int Hc, int H, int W, int flags, void *stream) {
    const int q8 = (flags & SSM_FLAG_Q8) ? 1 : 0;
    SSM_REQUIRE(!q8 || Hc % 16 == 0, "convgru_reset: a Q8 output needs hidden channels in multiples of 16");
    SSM_CELL_DIMS("convgru_reset");
    SSM_REQUIRE(gates_x.ptr && gates_h.ptr && h_prev.ptr && (rh_f32.ptr || rh_hl8.ptr), "convgru_reset: null pointer");
    SSM_REQUIRE(!rh_hl8.ptr || ssm::aligned16(rh_hl8.ptr), "convgru_reset: HL8 output must be 16-byte aligned");
    SSM_LAUNCH(convgru_reset_kernel, cell_grid(B, Hc, H, W), dim3(64, 4), 0, (hipStream_t)stream, gates_x, gates_h, h_prev,
                       rh_f32, rh_hl8, Hc, H, W, q8);
    return ssm::check_launch("ssm_convgru_reset_fwd");
}

extern "C" int ssm_convgru_update_fwd(ssm_view gates_x, ssm_view gates_h, ssm_view cand_x, ssm_view cand_h, ssm_view h_prev,
                                      ssm_view h_f32, ssm_hview h_hl8, int B, int Hc, int H, int W, int flags, void *stream) {
    const int q8 = (flags & SSM_FLAG_Q8) ? 1 : 0;
    SSM_REQUIRE(!q8 || Hc % 16 == 0, "convgru_update: a Q8 output needs hidden channels in multiples of 16");
    SSM_CELL_DIMS("convgru_update");
    SSM_REQUIRE(gates_x.ptr && cand_x.ptr && (h_f32.ptr || h_hl8.ptr), "convgru_update: null pointer");
    SSM_REQUIRE((gates_h.ptr != nullptr) == (h_prev.ptr != nullptr) && (cand_h.ptr != nullptr) == (h_prev.ptr != nullptr),
                "convgru_update: hidden-state inputs must be all present or all absent (first step)");
    SSM_REQUIRE(!h_hl8.ptr || ssm::aligned16(h_hl8.ptr), "convgru_update: HL8 output must be 16-byte aligned");
    SSM_LAUNCH(convgru_update_kernel, cell_grid(B, Hc, H, W), dim3(64, 4), 0, (hipStream_t)stream, gates_x, gates_h, cand_x,
                       cand_h, h_prev, h_f32, h_hl8, Hc, H, W, q8);
    return ssm::check_launch("ssm_convgru_update_fwd");
}

extern "C" int ssm_convlstm_cell_bwd(ssm_view gates_x, ssm_view gates_h, ssm_view c_prev, ssm_view dh, ssm_view dc_next, ssm_view dgates,
                                     ssm_view dc_prev, int B, int Hc, int H, int W, void *stream) {
    SSM_CELL_DIMS("convlstm_cell_bwd");
    SSM_REQUIRE(gates_x.ptr && dh.ptr && dgates.ptr && dc_prev.ptr, "convlstm_cell_bwd: null pointer");
    SSM_LAUNCH(convlstm_cell_bwd_kernel, cell_grid(B, Hc, H, W), dim3(64, 4), 0, (hipStream_t)stream, gates_x, gates_h, c_prev, dh,
                       dc_next, dgates, dc_prev, Hc, H, W);
    return ssm::check_launch("ssm_convlstm_cell_bwd");
}

extern "C" int ssm_convgru_reset_bwd(ssm_view gates, ssm_view h_prev, ssm_view drh, ssm_view dgates, ssm_view dh_prev, int B, int Hc, int H,
                                     int W, void *stream) {
    SSM_CELL_DIMS("convgru_reset_bwd");
    SSM_REQUIRE(gates.ptr && h_prev.ptr && drh.ptr && dgates.ptr && dh_prev.ptr, "convgru_reset_bwd: null pointer");
    SSM_LAUNCH(convgru_reset_bwd_kernel, cell_grid(B, Hc, H, W), dim3(64, 4), 0, (hipStream_t)stream, gates, h_prev, drh, dgates,
                       dh_prev, Hc, H, W);
    return ssm::check_launch("ssm_convgru_reset_bwd");
}

extern "C" int ssm_convgru_update_bwd(ssm_view gates, ssm_view cand, ssm_view h_prev, ssm_view dh_next, ssm_view dgates, ssm_view dcand,
                                      ssm_view dh_prev, int B, int Hc, int H, int W, void *stream) {
    SSM_CELL_DIMS("convgru_update_bwd");
    SSM_REQUIRE(gates.ptr && cand.ptr && dh_next.ptr && dgates.ptr && dcand.ptr, "convgru_update_bwd: null pointer");
    SSM_REQUIRE((h_prev.ptr != nullptr) == (dh_prev.ptr != nullptr), "convgru_update_bwd: h_prev and dh_prev go together");
    SSM_LAUNCH(convgru_update_bwd_kernel, cell_grid(B, Hc, H, W), dim3(64, 4), 0, (hipStream_t)stream, gates, cand, h_prev, dh_next,
                       dgates, dcand, dh_prev, Hc, H, W);
    return ssm::check_launch("ssm_convgru_update_bwd");
}
