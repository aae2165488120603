// Filter transform of the 5x5 layers' two-dimensional Winograd form (csrc/ssm_wino5.hip), shared by the per-layer pack kernel and the
// one-launch batch repack (csrc/ssm_pack.hip): one thread = one quad of four frequencies.
//      U = G g G^T,   G[f][k] = c_f p_f^k over the points p = {0, 1, -1, 2, -2, 1/2, -1/2}, c = {1, -2/9, -2/9, 1/90, 1/90, 32/45, 32/45}
// (the scaling that goes with B^T in the kernel), G[7][k] = [k == 4] (the point at infinity).  Evaluated in float64, rounded once.
// Packed layout [Cout/32][CinP/4][16 quads][4 channels][32 couts][4]: quad fq = 2 rf + h holds four column-frequencies of row-frequency
// rf; channels beyond Cin are zero.  W5_SPLIT (the frequency-split kernel, default): half h = 0 holds the column-frequencies {0, 1, 2, 7},
// h = 1 {3, 4, 5, 6} - the two halves of the 8-point transform that share no sub-expression (ssm_wino5.hip: w5_bt_lo / w5_bt_hi), so the
// two waves of a SIMD each transform and multiply their own half.  W5_SPLIT = 0 (tuning builds: the r4 kernel): cf = 4 h + e.
#pragma once
#ifndef W5_SPLIT
#define W5_SPLIT 1
#endif

template <class At>
__device__ inline void ssm_w5_pack_quad(const At &at, int Cout, int Cin, int CinP, long long idx, float out[4]) {
    long long r = idx / 4;
    const int n = (int)(r % 32);
    r /= 32;
    const int cq = (int)(r % 4);
    r /= 4;
    const int fq = (int)(r % 16);
    r /= 16;
    const int ks = (int)(r % (CinP / 4));
    const int nb = (int)(r / (CinP / 4));
    const int co = nb * 32 + n, cin = 4 * ks + cq, rf = fq >> 1, h = fq & 1;
    out[0] = out[1] = out[2] = out[3] = 0.f;
    if (co >= Cout || cin >= Cin) return;
    const double pt[7] = {0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5};
    const double cs[7] = {1.0, -2.0 / 9.0, -2.0 / 9.0, 1.0 / 90.0, 1.0 / 90.0, 32.0 / 45.0, 32.0 / 45.0};
    double row[5];         // (G g)[rf][c]
    for (int c = 0; c < 5; ++c) {
        double v = 0.0, pw = 1.0;
        for (int k = 0; k < 5; ++k) {
            v += (rf == 7 ? (k == 4 ? 1.0 : 0.0) : cs[rf] * pw) * (double)at(co, cin, k, c);
            pw *= rf == 7 ? 1.0 : pt[rf];
        }
        row[c] = v;
    }
    for (int e = 0; e < 4; ++e) {
#if W5_SPLIT
        const int cf = h == 0 ? (e == 3 ? 7 : e) : 3 + e;
#else
        const int cf = 4 * h + e;
#endif
        double v = 0.0, pw = 1.0;
        for (int k = 0; k < 5; ++k) {
            v += row[k] * (cf == 7 ? (k == 4 ? 1.0 : 0.0) : cs[cf] * pw);
            pw *= cf == 7 ? 1.0 : pt[cf];
        }
        out[e] = (float)v;
    }
}
