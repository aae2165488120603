// Filter transform of the 7x7 layers' two-dimensional Winograd form (csrc/ssm_wino7.hip), shared by the per-layer pack kernel and the
// one-launch batch repack (csrc/ssm_pack.hip): one thread = one quad of four frequencies.
//
// The 7x7 filter, padded with a zero row and column to 8x8, is cut into 2x2 BLOCKS of 4x4 taps, g_b[a][c] = w[4 by + a][4 bx + c]
// (b = 2 by + bx), and each block is a F(4x4,4x4) Winograd filter on the seven points {0, +1, -1, +2, -2, 1/2, inf}:
//      U_b = G g_b G^T,    G[i][k] = c_i p_i^k,   c = {-1/2, -1/3, 1/9, 1/36, -1/60, 32/45},   G[inf] = [0 0 0 1]
// (the scaling that goes with the monic rows of B^T in the kernel).  Evaluated in float64, rounded once.
// Packed layout [Cout/32][Cin][14 quads][4 blocks][32 couts][4]: quad fq = 2 rf + h holds the column-frequencies cf = 4 h .. 4 h + 3 of
// row-frequency rf (the fourth element of the odd quads is padding: zero).
#pragma once

#define SSM_W7_NFQ 14

template <class At>
__device__ inline void ssm_w7_pack_quad(const At &at, int Cout, int Cin, long long idx, float out[4]) {
    long long r = idx / 4;
    const int n = (int)(r % 32);
    r /= 32;
    const int b = (int)(r % 4);
    r /= 4;
    const int fq = (int)(r % SSM_W7_NFQ);
    r /= SSM_W7_NFQ;
    const int cin = (int)(r % Cin);
    const int nb = (int)(r / Cin);
    const int co = nb * 32 + n;
    out[0] = out[1] = out[2] = out[3] = 0.f;
    if (co >= Cout) return;
    const double pt[7] = {0.0, 1.0, -1.0, 2.0, -2.0, 0.5, 0.0};
    const double cs[7] = {-0.5, -1.0 / 3.0, 1.0 / 9.0, 1.0 / 36.0, -1.0 / 60.0, 32.0 / 45.0, 0.0};
    const int by = b >> 1, bx = b & 1, rf = fq >> 1, h = fq & 1;
    double g[4][4];
    for (int a = 0; a < 4; ++a)
        for (int c = 0; c < 4; ++c) {
            const int ky = 4 * by + a, kx = 4 * bx + c;
            g[a][c] = (ky < 7 && kx < 7) ? (double)at(co, cin, ky, kx) : 0.0;
        }
    double Gr[4];          // row rf of G
    {
        double pw = 1.0;
        for (int k = 0; k < 4; ++k) {
            Gr[k] = rf == 6 ? (k == 3 ? 1.0 : 0.0) : cs[rf] * pw;
            pw *= pt[rf];
        }
    }
    double row[4];         // (G g)[rf][c]
    for (int c = 0; c < 4; ++c) row[c] = Gr[0] * g[0][c] + Gr[1] * g[1][c] + Gr[2] * g[2][c] + Gr[3] * g[3][c];
    for (int e = 0; e < 4; ++e) {
        const int cf = 4 * h + e;
        if (cf >= 7) continue;
        double pw = 1.0, val = 0.0;
        for (int k = 0; k < 4; ++k) {
            val += row[k] * (cf == 6 ? (k == 3 ? 1.0 : 0.0) : cs[cf] * pw);
            pw *= pt[cf];
        }
        out[e] = (float)val;
    }
}
