// conv_shared.h — declarations shared by the direct (conv.hip) and the Winograd (wino.hip) 3x3 kernels.
#pragma once
#include "common.h"

// The BatchNorm-backward epilogue of a data-gradient launch (see conv3x3_mfma_fwd2_k in conv.hip for the derivation).
struct ConvBnRed {
    const float* pooled;     // [B][T][F][Cout] forward output of the block whose BatchNorm is being differentiated
    const float* gamma;      // [Cout]
    const float* beta;       // [Cout]
    const float* ybelow;     // != NULL: the block below stores its conv output — channels whose xhat cannot be recovered from the pooled
    const float* mean;       // output (|gamma| < |beta| / 64) then contribute 0 here and are recomputed by sed_bn_bwd_finalize_small_gamma
    const float* rstd;
    float keep, inv_keep;    // 1 - p, 1 / (1 - p)
    int pf, pt, Fy, Ty;      // pool and the extents of ybelow
    // RG (the block below is the recomputed 1-channel first block, pool (1,2)): also its weight-gradient sums, see the kernel
    const float* x1;         // the network input [B][RGC][Fy][Ty]
    const unsigned char* bits;   // arg-max bits of the block below: [B][T][F][Cout/4] bytes, bit e = channel 4q+e took the second time row
    float* rgp;              // out: [rows][Cout][1 + 9 RGC] = (sum g, R_k) per workgroup
    float invXT;             // 1 / (2 TT + 2)
};

#ifdef __HIPCC__
// fragment order [co/64][component][ci/32][(ci%32)/8][(co%64)/32][lane = co%32 + 32 ((ci%8)/4)][ci%4]: everything a workgroup (one
// channel half) streams is one contiguous 16 K 64 floats, and inside it every (component, step, channel tile) offset is a constant
__host__ __device__ inline size_t wino_frag_index(int comp, int k, int n, int K, int N) {
    const int cc = k >> 5, g = (k & 31) >> 3, h = (k & 7) >> 2, j = k & 3, coh = n >> 6, nt = (n >> 5) & 1, r = n & 31;
    (void)N;
    return (((((((size_t)coh * 16 + comp) * (K >> 5) + cc) * 4 + g) * 2 + nt) * 64) + r + 32 * h) * 4 + j;
}
#define WN_ZTAIL 256        // zero floats behind the packed weights: the source of the patch's zero padding (LDS-DMA cannot write a constant)
// one thread per (co, ci): U = G g G^T for the forward (g[a][c] = w[co][ci][kh = c][kw = a]: a runs along time, c along mel) and for
// the data gradient (contraction over co, g'[a][c] = w[co][ci][2 - c][2 - a])
// gamma / rv (may be NULL): the weights of output channel co are scaled by gamma[co] / sqrt(rv[co] + eps) first (inference: BatchNorm folded)
__device__ __forceinline__ void wino_pack_one(const float* __restrict__ w, float* __restrict__ uf, float* __restrict__ ud, int Cout, int Cin, int i,
                                              const float* __restrict__ gamma = nullptr, const float* __restrict__ rv = nullptr, float eps = 0.f) {
    if (i < WN_ZTAIL) {
        if (uf) uf[(size_t)16 * Cout * Cin + i] = 0.f;
        if (ud) ud[(size_t)16 * Cout * Cin + i] = 0.f;
    }
    if (i >= Cout * Cin) return;
    const int ci = i % Cin, co = i / Cin;
    float g[3][3];                                   // [a = kw][c = kh]
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) g[kw][kh] = w[(size_t)i * 9 + kh * 3 + kw];
    if (gamma) {
        const float sc = gamma[co] / sqrtf(rv[co] + eps);
#pragma unroll
        for (int k = 0; k < 9; ++k) g[k / 3][k % 3] *= sc;
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float* dst = pass ? ud : uf;
        if (!dst) continue;
        float t[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g0 = pass ? g[2][2 - c] : g[0][c], g1 = pass ? g[1][2 - c] : g[1][c], g2 = pass ? g[0][2 - c] : g[2][c];
            t[0][c] = g0;
            t[1][c] = 0.5f * (g0 + g1 + g2);
            t[2][c] = 0.5f * (g0 - g1 + g2);
            t[3][c] = g2;
        }
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            const float u0 = t[xi][0], u1 = 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]), u2 = 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]), u3 = t[xi][2];
            const float uu[4] = {u0, u1, u2, u3};
#pragma unroll
            for (int nu = 0; nu < 4; ++nu)
                dst[pass ? wino_frag_index(xi * 4 + nu, co, ci, Cout, Cin) : wino_frag_index(xi * 4 + nu, ci, co, Cin, Cout)] = uu[nu];
        }
    }
}
#endif

// wino.hip: F(2x2, 3x3) forward / data gradient of the 128-channel blocks (see the file header)
int sed_internal_wino_rows(int B, int Cin, int F, int T, int Cout);
int sed_internal_wino_launch(const float* x, const float* uq, const float* bias, float* y, float* stat, const ConvBnRed* br,
                             int rgc, int B, int Cin, int F, int T, int Cout, hipStream_t s);
