/* ssm_oracle.c - plain-C restatement of the primitive operators of the Super SloMo hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/ssm_oracle.py header): a torch-free second checker
 * for small cases.  Contiguous NCHW fp32 everywhere.  Each function cites the reference
 * file:line it follows (paths relative to the reference repository root).  Pinned by
 * tests/test_oracle_golden.py against fixtures generated from the reference itself.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no fused multiply-add, so the
 * coordinate arithmetic rounds like the reference's separate fp32 CPU ops).
 */
#include <math.h>
#include <stddef.h>

#define IDX4(c, y, x, C, H, W) (((size_t)(c) * (H) + (y)) * (W) + (x))

/* scripts/models/layers.py:21-33 (Conv2d stride 1, pad (k-1)/2, bias [+ LeakyReLU slope]);
 * lrelu=0 gives the bare final_conv (scripts/models/flow_computation.py:145-153). */
void oracle_conv2d(const float *x, const float *w, const float *bias, float *y, int B, int Cin, int H, int W,
                   int Cout, int k, int lrelu, float slope) {
    const int pad = (k - 1) / 2;
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < H; ++oy)
                for (int ox = 0; ox < W; ++ox) {
                    double acc = 0.0; /* wide accumulator: an order-independent reference value */
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int ky = 0; ky < k; ++ky) {
                            const int iy = oy + ky - pad;
                            if (iy < 0 || iy >= H) continue;
                            for (int kx = 0; kx < k; ++kx) {
                                const int ix = ox + kx - pad;
                                if (ix < 0 || ix >= W) continue;
                                acc += (double)x[(size_t)b * Cin * H * W + IDX4(ci, iy, ix, Cin, H, W)] *
                                       (double)w[(((size_t)co * Cin + ci) * k + ky) * k + kx];
                            }
                        }
                    float v = (float)(acc + (double)bias[co]);
                    if (lrelu && !(v > 0.f)) v = v * slope;
                    y[(size_t)b * Cout * H * W + IDX4(co, oy, ox, Cout, H, W)] = v;
                }
}

/* scripts/models/layers.py:60-63: AvgPool2d(2, stride 2). */
void oracle_avgpool2(const float *x, float *y, int BC, int H, int W) {
    const int h = H / 2, w = W / 2;
    for (int n = 0; n < BC; ++n)
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < w; ++j) {
                const float *p = x + ((size_t)n * H + 2 * i) * W + 2 * j;
                y[((size_t)n * h + i) * w + j] = (((p[0] + p[1]) + p[W]) + p[W + 1]) * 0.25f;
            }
}

/* F.upsample(x, size=(2h,2w), mode="bilinear") = align_corners False, half-pixel centres,
 * scripts/models/flow_computation.py:92-94. */
void oracle_upsample2x(const float *x, float *y, int BC, int h, int w) {
    const int H = 2 * h, W = 2 * w;
    for (int n = 0; n < BC; ++n)
        for (int Y = 0; Y < H; ++Y) {
            float sy = ((float)Y + 0.5f) * 0.5f - 0.5f;
            if (sy < 0.f) sy = 0.f;
            const int y0 = (int)sy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
            const float ly = sy - (float)y0;
            for (int X = 0; X < W; ++X) {
                float sx = ((float)X + 0.5f) * 0.5f - 0.5f;
                if (sx < 0.f) sx = 0.f;
                const int x0 = (int)sx, x1 = x0 + (x0 < w - 1 ? 1 : 0);
                const float lx = sx - (float)x0;
                const float *p = x + (size_t)n * h * w;
                y[((size_t)n * H + Y) * W + X] =
                    (1.f - ly) * ((1.f - lx) * p[y0 * w + x0] + lx * p[y0 * w + x1]) +
                    ly * ((1.f - lx) * p[y1 * w + x0] + lx * p[y1 * w + x1]);
            }
        }
}

static float sample_bilinear_zeros(const float *plane, int H, int W, float ix, float iy) {
    const float x0 = floorf(ix), y0 = floorf(iy), x1 = x0 + 1.f, y1 = y0 + 1.f;
    const float w00 = (x1 - ix) * (y1 - iy), w01 = (ix - x0) * (y1 - iy);
    const float w10 = (x1 - ix) * (iy - y0), w11 = (ix - x0) * (iy - y0);
    float r = 0.f;
#define TAP(xs, ys, wt) \
    if ((xs) >= 0.f && (xs) <= (float)(W - 1) && (ys) >= 0.f && (ys) <= (float)(H - 1)) r += plane[(int)(ys) * W + (int)(xs)] * (wt)
    TAP(x0, y0, w00);
    TAP(x1, y0, w01);
    TAP(x0, y1, w10);
    TAP(x1, y1, w11);
#undef TAP
    return r;
}

/* scripts/models/layers.py:73-120: normalise (x+u, y+v) to [-1,1] (:112-113), then
 * grid_sample(bilinear, zeros, align_corners=True) maps back and samples (:119). */
static void warp_coords(int x, int y, float u, float v, int H, int W, float *ix, float *iy) {
    const float wd = (float)(W - 1 > 1 ? W - 1 : 1), hd = (float)(H - 1 > 1 ? H - 1 : 1);
    const float gx = 2.0f * ((float)x + u) / wd - 1.0f, gy = 2.0f * ((float)y + v) / hd - 1.0f;
    *ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    *iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
}

void oracle_warp(const float *img, const float *flo, float *out, int B, int C, int H, int W) {
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float ix, iy;
                warp_coords(x, y, flo[((size_t)b * 2 * H + y) * W + x], flo[(((size_t)b * 2 + 1) * H + y) * W + x], H, W, &ix, &iy);
                for (int c = 0; c < C; ++c)
                    out[(((size_t)b * C + c) * H + y) * W + x] = sample_bilinear_zeros(img + ((size_t)b * C + c) * H * W, H, W, ix, iy);
            }
}

/* scripts/models/flow_interpolation.py:338-372 (channel order :364-367). */
void oracle_flowinterp_inputs(const float *img6, const float *flow4, const float *t, float *out16, int B, int H, int W) {
    const size_t P = (size_t)H * W;
    for (int b = 0; b < B; ++b) {
        const float tt = t[b], omt = 1.0f - tt;
        const float c00 = (-omt) * tt, c01 = tt * tt, c10 = omt * omt, c11 = tt * omt;
        const float *I = img6 + (size_t)b * 6 * P, *F = flow4 + (size_t)b * 4 * P;
        float *O = out16 + (size_t)b * 16 * P;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t p = (size_t)y * W + x;
                const float ft0u = c00 * F[p] + c01 * F[2 * P + p], ft0v = c00 * F[P + p] + c01 * F[3 * P + p];
                const float ft1u = c10 * F[p] - c11 * F[2 * P + p], ft1v = c10 * F[P + p] - c11 * F[3 * P + p];
                float x1, y1, x0, y0;
                warp_coords(x, y, ft1u, ft1v, H, W, &x1, &y1);
                warp_coords(x, y, ft0u, ft0v, H, W, &x0, &y0);
                for (int c = 0; c < 3; ++c) {
                    O[(size_t)c * P + p] = I[(size_t)(3 + c) * P + p];
                    O[(size_t)(3 + c) * P + p] = sample_bilinear_zeros(I + (size_t)(3 + c) * P, H, W, x1, y1);
                    O[(size_t)(10 + c) * P + p] = sample_bilinear_zeros(I + (size_t)c * P, H, W, x0, y0);
                    O[(size_t)(13 + c) * P + p] = I[(size_t)c * P + p];
                }
                O[6 * P + p] = ft1u;
                O[7 * P + p] = ft1v;
                O[8 * P + p] = ft0u;
                O[9 * P + p] = ft0v;
            }
    }
}

/* scripts/models/flow_interpolation.py:374-429. */
void oracle_synthesize(const float *img6, const float *in16, const float *out5, const float *t, float *y3, int B, int H, int W) {
    const size_t P = (size_t)H * W;
    for (int b = 0; b < B; ++b) {
        const float tt = t[b], omt = 1.0f - tt;
        const float *I = img6 + (size_t)b * 6 * P, *X = in16 + (size_t)b * 16 * P, *O = out5 + (size_t)b * 5 * P;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t p = (size_t)y * W + x;
                const float v1 = 1.0f / (1.0f + expf(-O[p])), v0 = 1.0f - v1;
                float x1, y1, x0, y0;
                warp_coords(x, y, X[6 * P + p] + O[P + p], X[7 * P + p] + O[2 * P + p], H, W, &x1, &y1);
                warp_coords(x, y, X[8 * P + p] + O[3 * P + p], X[9 * P + p] + O[4 * P + p], H, W, &x0, &y0);
                const float den = omt * v0 + tt * v1;
                for (int c = 0; c < 3; ++c) {
                    const float p0 = v0 * sample_bilinear_zeros(I + (size_t)c * P, H, W, x0, y0);
                    const float p1 = v1 * sample_bilinear_zeros(I + (size_t)(3 + c) * P, H, W, x1, y1);
                    y3[((size_t)b * 3 + c) * P + p] = (omt * p0 + tt * p1) / den;
                }
            }
    }
}
