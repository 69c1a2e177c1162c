// conv1.hip — first conv block with the convolution RECOMPUTED instead of stored.
//
// Block 1 of the reference stack (sed.py:88-92,106-107 with ch = 1 at sed.py:86) turns a 5 MB input into a
// 671 MB conv output (config 2) whose only readers are BatchNorm's statistics, the normalise/ReLU/pool pass and
// their backward.  With Cin <= 2 one output costs 9..18 FMAs, far less than moving it through HBM, so none of the
// four passes materialises it: each recomputes conv(x) from the L2-resident input tile in LDS.
//   A  stats      : sum / sum^2 partials                         (reads x)
//   B  fwd        : BN scale/shift + ReLU + max-pool + dropout    (reads x, writes pooled)
//   C  bwd reduce : sum g, sum g*xhat                             (reads x, dpooled)
//   D  bwd apply  : dy on the fly -> dW (9*Cin taps), dbias       (reads x, dpooled; dy is never written;
//                                                                   the network input needs no data gradient)
// HBM traffic of block 1 drops from ~5.0 GB to ~1.0 GB per step at config 2.  Same arithmetic order in all four
// passes, so the recomputed values are bit-identical between passes; fixed-order two-stage reductions.
#include "common.h"

#define C1_TT 4
#define C1_MAXBLOCKS 2048

template <int CIN>
struct C1W { f32x4 w[9 * CIN]; f32x4 b; };

template <int CIN>
__device__ __forceinline__ void c1_load_w(C1W<CIN>& W, const float* __restrict__ wp, const float* __restrict__ bias,
                                          int Cout, int cg) {
    // wp [9][Cout][Cin]
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int k = 0; k < 4; ++k) W.w[tap * CIN + ci][k] = wp[((size_t)tap * Cout + cg * 4 + k) * CIN + ci];
    W.b = bias ? *(const f32x4*)(bias + cg * 4) : (f32x4){0, 0, 0, 0};
}

template <int CIN>
__device__ __forceinline__ f32x4 c1_conv(const C1W<CIN>& W, const float* __restrict__ halo, int tl, int f, int F2) {
    f32x4 acc = W.b;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const float* hp = halo + ((tl + kw) * F2 + f + kh) * CIN;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) acc += hp[ci] * W.w[(kh * 3 + kw) * CIN + ci];
        }
    return acc;
}

// (1,2) pool: the conv outputs at time rows tl and tl+1 of one mel column from ONE 4x3xCIN halo patch (12 LDS reads
// instead of 18); same FMA order as c1_conv, so the values are bit-identical to the other passes.
template <int CIN>
__device__ __forceinline__ void c1_conv2(const C1W<CIN>& W, const float* __restrict__ halo, int tl, int f, int F2,
                                         f32x4& v0, f32x4& v1) {
    float hv[4][3][CIN];
#pragma unroll
    for (int kw = 0; kw < 4; ++kw)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const float* hp = halo + ((tl + kw) * F2 + f + kh) * CIN;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) hv[kw][kh][ci] = hp[ci];
        }
    v0 = W.b;
    v1 = W.b;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                v0 += hv[kw][kh][ci] * W.w[(kh * 3 + kw) * CIN + ci];
                v1 += hv[kw + 1][kh][ci] * W.w[(kh * 3 + kw) * CIN + ci];
            }
}

template <int CIN>
__device__ __forceinline__ void c1_stage(float* halo, const float* __restrict__ x, int b, int t0, int F, int T) {
    const int F2 = F + 2;
    const int n = (C1_TT + 2) * F2 * CIN;
    for (int i = threadIdx.x; i < n; i += 256) {          // time fastest: contiguous in the NCHW input
        int tt = i % (C1_TT + 2), ff = (i / (C1_TT + 2)) % F2, ci = i / ((C1_TT + 2) * F2);
        int t = t0 + tt - 1, f = ff - 1;
        float v = 0.f;
        if (t >= 0 && t < T && f >= 0 && f < F) v = x[(((size_t)b * CIN + ci) * F + f) * T + t];
        halo[(tt * F2 + ff) * CIN + ci] = v;
    }
}

// MODE 0: stats  1: forward  2: backward reduce  3: backward apply + weight gradient
//      4: routing codes (inspection / parity tests): the decisions of passes 2 and 3 — same search, same expressions — as
//         one byte per pooled element in `out` ([B][Tp][Fp][C]: 0 = ReLU gate closed, 1 + w = window element w = df*pt + dt)
// P12: the pool is (1,2): unrolled window, both conv outputs of a window from one halo patch, kept for the apply loop
template <int CIN, int MODE, bool P12>
__global__ __launch_bounds__(256) void conv1_fused_k(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ sum_g, const float* __restrict__ sum_gx,
    const float* __restrict__ dout, float* __restrict__ out, float* __restrict__ partials,
    int B, int F, int T, int C, int pf, int pt, float drop_p, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
    if (seed_dev) seed += seed_dev[0] * 0x9E3779B97F4A7C15ull;   // per-step salt kept on the device (graph replay)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (P12) { pf = 1; pt = 2; }
    const int F2 = F + 2;
    float* halo = smem;                                       // [(TT+2)][F2][CIN]
    const int tid = threadIdx.x;
    const int C4 = C >> 2, nslots = 256 / C4;
    const int cg = tid % C4, slot = tid / C4;
    const bool active = slot < nslots;
    const int tblocks = (T + C1_TT - 1) / C1_TT, ntiles = B * tblocks;
    const int Tp = T / pt, Fp = F / pf;
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const float invN = 1.f / ((float)B * (float)T * (float)F);

    C1W<CIN> W;
    if (active) c1_load_w<CIN>(W, wp, bias, C, cg);
    f32x4 sc = {0, 0, 0, 0}, sh = sc, mu = sc, rs = sc, sg = sc, sgx = sc;
    if (active && MODE >= 1) {
        sc = *(const f32x4*)(scale + cg * 4);
        sh = *(const f32x4*)(shift + cg * 4);
        if (MODE >= 2 && MODE != 4) { mu = *(const f32x4*)(mean + cg * 4); rs = *(const f32x4*)(rstd + cg * 4); }
        if (MODE == 3) { sg = *(const f32x4*)(sum_g + cg * 4) * invN; sgx = *(const f32x4*)(sum_gx + cg * 4) * invN; }
    }
    f32x4 a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};              // stats / (sum g, sum g xhat) / (dbias, sum g xhat again)
    f32x4 dw[MODE == 3 ? 9 * CIN : 1];
    if (MODE == 3) {
#pragma unroll
        for (int k = 0; k < 9 * CIN; ++k) dw[k] = (f32x4){0, 0, 0, 0};
    }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tblocks, t0 = (tile - b * tblocks) * C1_TT;
        __syncthreads();
        c1_stage<CIN>(halo, x, b, t0, F, T);
        __syncthreads();
        if (!active) continue;
        if (MODE == 0) {
            for (int p = slot; p < C1_TT * F; p += nslots) {
                int tl = p / F, f = p - tl * F;
                if (t0 + tl >= T) break;
                f32x4 v = c1_conv<CIN>(W, halo, tl, f, F2);
                a1 += v;
                a2 += v * v;
            }
        } else {
            const int tpn = C1_TT / pt;                          // pooled rows in this tile (pt divides TT)
            for (int op = slot; op < tpn * Fp; op += nslots) {
                int tpl = op / Fp, fp = op - tpl * Fp;
                int tp = t0 / pt + tpl;
                if (tp >= Tp) break;
                size_t oi = ((((size_t)b * Tp + tp) * Fp + fp) * C4 + cg) * 4;   // channels-last index of the pooled element
                f32x4 vw[2];
                if (P12 && MODE >= 1) c1_conv2<CIN>(W, halo, tpl * 2, fp, F2, vw[0], vw[1]);
                if (MODE == 1) {
                    f32x4 m = {0, 0, 0, 0};
#pragma unroll
                    for (int df = 0; df < (P12 ? 1 : pf); ++df)
#pragma unroll
                        for (int dt = 0; dt < (P12 ? 2 : pt); ++dt) {
                            f32x4 z = (P12 ? vw[dt] : c1_conv<CIN>(W, halo, tpl * pt + dt, fp * pf + df, F2)) * sc + sh;
#pragma unroll
                            for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], z[k]);
                        }
                    if (P12 && partials) {           // arg-max of the (1,2) window, one bit per channel (first maximum wins a tie)
                        const f32x4 z0 = vw[0] * sc + sh, z1 = vw[1] * sc + sh;
                        unsigned bt = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) bt |= (z1[k] > z0[k] ? 1u : 0u) << k;
                        reinterpret_cast<unsigned char*>(partials)[oi >> 2] = (unsigned char)bt;
                    }
                    if (drop_p > 0.f) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) m[k] *= sed_drop_mult(seed, oi + k, drop_p, inv_keep);
                    }
                    *(f32x4*)(out + oi) = m;
                } else {
                    f32x4 g = {0, 0, 0, 0};
                    if (MODE != 4) {
                        g = *(const f32x4*)(dout + oi);
                        if (drop_p > 0.f) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) g[k] *= sed_drop_mult(seed, oi + k, drop_p, inv_keep);
                        }
                    }
                    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, bx = {0, 0, 0, 0};
                    int bidx[4] = {0, 0, 0, 0};
                    int widx = 0;
#pragma unroll
                    for (int df = 0; df < (P12 ? 1 : pf); ++df)
#pragma unroll
                        for (int dt = 0; dt < (P12 ? 2 : pt); ++dt, ++widx) {
                            f32x4 v = P12 ? vw[dt] : c1_conv<CIN>(W, halo, tpl * pt + dt, fp * pf + df, F2);
                            f32x4 z = v * sc + sh;
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (z[k] > best[k]) { best[k] = z[k]; bx[k] = (v[k] - mu[k]) * rs[k]; bidx[k] = widx; }
                        }
                    if (MODE == 4) {
                        uchar4 code;
                        code.x = best[0] > 0.f ? (unsigned char)(1 + bidx[0]) : 0;
                        code.y = best[1] > 0.f ? (unsigned char)(1 + bidx[1]) : 0;
                        code.z = best[2] > 0.f ? (unsigned char)(1 + bidx[2]) : 0;
                        code.w = best[3] > 0.f ? (unsigned char)(1 + bidx[3]) : 0;
                        *(uchar4*)(reinterpret_cast<unsigned char*>(out) + oi) = code;
                        continue;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (!(best[k] > 0.f)) g[k] = 0.f;
                    if (MODE == 2) {
                        a1 += g;
                        a2 += g * bx;
                    } else {
                        widx = 0;
#pragma unroll
                        for (int df = 0; df < (P12 ? 1 : pf); ++df)
#pragma unroll
                            for (int dt = 0; dt < (P12 ? 2 : pt); ++dt, ++widx) {
                                const int tl = tpl * pt + dt, f = fp * pf + df;
                                f32x4 v = P12 ? vw[dt] : c1_conv<CIN>(W, halo, tl, f, F2);
                                f32x4 o;
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    float xh = (v[k] - mu[k]) * rs[k];
                                    float gk = (bidx[k] == widx) ? g[k] : 0.f;
                                    o[k] = sc[k] * (gk - sg[k] - xh * sgx[k]);
                                    a2[k] += gk * xh;          // dgamma once more (used for channels with gamma == 0, see the reduce kernel)
                                }
                                a1 += o;
#pragma unroll
                                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                                    for (int kw = 0; kw < 3; ++kw) {
                                        const float* hp = halo + ((tl + kw) * F2 + f + kh) * CIN;
#pragma unroll
                                        for (int ci = 0; ci < CIN; ++ci) dw[(kh * 3 + kw) * CIN + ci] += hp[ci] * o;
                                    }
                            }
                    }
                }
            }
        }
    }
    if (MODE == 1 || MODE == 4) return;
    // block reduction in fixed slot order, NVC values per round: the apply pass runs beside the conv weight-gradient kernel,
    // which leaves ~38 KB of LDS per CU, so its 1 + 9*CIN values go through a 20 KB buffer in rounds of 5
    constexpr int NV = (MODE == 3) ? 2 + 9 * CIN : 2;       // MODE 3: dbias, 9*CIN weight-gradient taps, sum g*xhat
    constexpr int NVC = (MODE == 3) ? 5 : 2;
    float* red = smem;                                        // [nslots][NVC][C]
#pragma unroll
    for (int v0 = 0; v0 < NV; v0 += NVC) {
        __syncthreads();
        if (active) {
#pragma unroll
            for (int j = 0; j < NVC; ++j) {
                const int v = v0 + j;
                if (v < NV) {
                    f32x4 val = (v == 0) ? a1 : ((MODE == 3 && v <= 9 * CIN) ? dw[(v - 1) < 9 * CIN ? (v - 1) : 0] : a2);
                    *(f32x4*)(red + (slot * NVC + j) * C + cg * 4) = val;
                }
            }
        }
        __syncthreads();
        const int nv = (NV - v0 < NVC) ? NV - v0 : NVC;
        for (int i = tid; i < nv * C; i += 256) {
            int j = i / C, c = i - j * C;
            float a = 0.f;
            for (int s = 0; s < nslots; ++s) a += red[(s * NVC + j) * C + c];
            partials[(size_t)blockIdx.x * NV * C + (size_t)(v0 + j) * C + c] = a;
        }
    }
}

// dW[co][ci][tap] = sum_r part[r][1 + tap*Cin + ci][co]   (row 0 of each block is the bias gradient, the last row sum g*xhat)
// (256-thread workgroups: it runs on the auxiliary stream beside the conv weight gradients, see reduce_rows_k)
// The last row repeats dgamma with the decisions of the apply pass.  It is only USED for a channel whose gamma is exactly 0
// with beta > 0: there the fused reduction in the data gradient above (conv.hip, ConvBnRed) cannot recover xhat from the
// block's output, and this block keeps no conv output to read it from.
__global__ __launch_bounds__(256) void conv1_wgrad_reduce_k(const float* __restrict__ part, int rows, int Cin, int C,
                                                            float* __restrict__ dw, float* __restrict__ db,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ dgamma) {
    __shared__ double s1[32][9];
    const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int NV = 2 + 9 * Cin, n = NV * C;
    const int i = blockIdx.x * 8 + cl;
    double a = 0.0;
    if (i < n)
#pragma unroll 8
        for (int r = sl; r < rows; r += 32) a += (double)part[(size_t)r * n + i];
    s1[sl][cl] = a;
    __syncthreads();
    if (sl == 0 && i < n) {
        double A = 0.0;
        for (int s = 0; s < 32; ++s) A += s1[s][cl];
        int which = i / C, co = i - which * C;
        if (which == 0) db[co] = (float)A;
        else if (which == NV - 1) {
            if (dgamma && gamma && beta && gamma[co] == 0.f && beta[co] > 0.f) dgamma[co] = (float)A;
        } else {
            int tc = which - 1, tap = tc / Cin, ci = tc - tap * Cin;
            dw[((size_t)co * Cin + ci) * 9 + tap] = (float)A;
        }
    }
}

// ── pass A without recomputing the convolution: the statistics of a linear map come from the moments of its input ──
// y[c] = b[c] + sum_k w[c][k] v[k] with v = the 9*CIN shifted inputs of a position (zero outside the image), so
//   sum y   = N b + w . S1            S1[k]    = sum_pos v[k]
//   sum y^2 = w' G w + 2 b w . S1 + N b^2      G[k][k'] = sum_pos v[k] v[k']   (upper triangle kept)
// One pass over the 5 MB input instead of a second full recompute of the 671 MB conv output (config 2: 70 -> ~10 us).
// Partial sums per thread (a handful of positions) and per block in fp32, across blocks and in the quadratic form in fp64.
#define C1_GT 16      // time rows per tile of the moment pass
#define C1_GR 4       // tiles staged per round
template <int CIN>
__global__ __launch_bounds__(256) void conv1_gram_k(const float* __restrict__ x, float* __restrict__ partials, int B, int F, int T) {
    constexpr int NK = 9 * CIN, NG = NK + NK * (NK + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int F2 = F + 2;
    float* halo = smem;                                       // [(GT+2)][F2][CIN]
    const int tid = threadIdx.x;
    const int tblocks = (T + C1_GT - 1) / C1_GT, ntiles = B * tblocks;
    float acc[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) acc[i] = 0.f;
    // C1_GR tiles per round: all their halo loads are in flight together (one global latency per round, not per tile)
    const int hn = (C1_GT + 2) * F2 * CIN;
    for (int tile0 = blockIdx.x * C1_GR; tile0 < ntiles; tile0 += gridDim.x * C1_GR) {
        __syncthreads();
        for (int i = tid; i < C1_GR * hn; i += 256) {         // time fastest: contiguous in the NCHW input
            const int u = i / hn, j = i - u * hn;
            const int tile = tile0 + u;
            int tt = j % (C1_GT + 2), ff = (j / (C1_GT + 2)) % F2, ci = j / ((C1_GT + 2) * F2);
            float v = 0.f;
            if (tile < ntiles) {
                const int b = tile / tblocks, t = (tile - b * tblocks) * C1_GT + tt - 1, f = ff - 1;
                if (t >= 0 && t < T && f >= 0 && f < F) v = x[(((size_t)b * CIN + ci) * F + f) * T + t];
            }
            halo[u * hn + (tt * F2 + ff) * CIN + ci] = v;
        }
        __syncthreads();
        for (int p = tid; p < C1_GR * C1_GT * F; p += 256) {
            const int u = p / (C1_GT * F), q = p - u * (C1_GT * F);
            const int tile = tile0 + u;
            if (tile >= ntiles) break;
            const int tl = q / F, f = q - tl * F;
            const int b = tile / tblocks, t0 = (tile - b * tblocks) * C1_GT;
            if (t0 + tl >= T) continue;
            const float* hb = halo + u * hn;
            float v[NK];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) v[(kh * 3 + kw) * CIN + ci] = hb[((tl + kw) * F2 + f + kh) * CIN + ci];
            int g = NK;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                acc[k] += v[k];
#pragma unroll
                for (int k2 = k; k2 < NK; ++k2) acc[g++] += v[k] * v[k2];
            }
        }
    }
    // block sums in wave order (fixed), one value at a time through the wave reduction + a 4-float exchange
    __syncthreads();
    float* red = smem;                                         // [NG][4]
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const float sum = wave_sum(acc[i]);
        if (lane == 0) red[i * 4 + wave] = sum;
    }
    __syncthreads();
    for (int i = tid; i < NG; i += 256)
        partials[(size_t)blockIdx.x * NG + i] = (red[i * 4] + red[i * 4 + 1]) + (red[i * 4 + 2] + red[i * 4 + 3]);
}

// stat[0][c] = sum y, stat[1][c] = sum y^2 of channel c (one partial row in the format of the stored path's conv epilogue)
template <int CIN>
__global__ __launch_bounds__(256) void conv1_gram_finalize_k(const float* __restrict__ partials, int nblk, const float* __restrict__ wp,
                                                             const float* __restrict__ bias, double count, int C, float* __restrict__ stat,
                                                             double* __restrict__ gram_out) {
    constexpr int NK = 9 * CIN, NG = NK + NK * (NK + 1) / 2;
    __shared__ double G[NG];
    __shared__ double Gq[4][64];
    {   // G[i] = sum over the blocks' partial rows in a fixed order: thread (i % 64, q) sums rows q, q+4, ... (8 loads in flight),
        // then the four row classes are added in order
        const int il = threadIdx.x & 63, q = threadIdx.x >> 6;
        for (int i0 = 0; i0 < NG; i0 += 64) {
            const int i = i0 + il;
            double a = 0.0;
            if (i < NG) {
#pragma unroll 8
                for (int r = q; r < nblk; r += 4) a += (double)partials[(size_t)r * NG + i];
            }
            Gq[q][il] = a;
            __syncthreads();
            if (q == 0 && i < NG) G[i] = (Gq[0][il] + Gq[1][il]) + (Gq[2][il] + Gq[3][il]);
            __syncthreads();
        }
    }
    __syncthreads();
    if (gram_out)
        for (int i = threadIdx.x; i < NG; i += 256) gram_out[i] = G[i];
    for (int c = threadIdx.x; c < C; c += 256) {
        double w[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) w[k] = (double)wp[((size_t)(k / CIN) * C + c) * CIN + (k % CIN)];      // wp [9][C][CIN]
        const double b = bias ? (double)bias[c] : 0.0;
        double ws1 = 0.0, q = 0.0;
        int g = NK;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            ws1 += w[k] * G[k];
#pragma unroll
            for (int k2 = k; k2 < NK; ++k2) { q += (k2 == k ? 1.0 : 2.0) * w[k] * w[k2] * G[g]; ++g; }
        }
        stat[c] = (float)(count * b + ws1);
        stat[C + c] = (float)(q + 2.0 * b * ws1 + count * b * b);
    }
}

// ── input moments for 3 or 4 input channels (config 5: 4-channel input) ──
// conv1_gram_k keeps all NK + NK(NK+1)/2 accumulators of a thread in registers: 54 at one channel, 189 at two, 702 at four.
// (Round 3 cut the matrix into row blocks of 4 taps, 148 accumulators per thread: LDS-bound, 1.39 ms at config 5; superseded.)
#define C1_MM_BLOCKS 1024     // workgroups of conv1_moments_mfma_k (4 per CU: 21 KB of LDS each at 128 mel bins x 4 channels)
// ── the moments of 3 and 4 input channels as a Gram matrix on the matrix cores (round 4) ──
// S1 and G are V^T V for V = [positions x (9 CIN shifted inputs, 1)]: a rank-4-update per instruction with
// v_mfma_f32_16x16x4_f32.  The NK + 1 features (the constant 1 gives S1 as the last column) are padded to NT = 2 or 3 tiles of 16;
// lane l of a wave holds feature 16 t + (l & 15) of position (l >> 4) of the current group of four consecutive mel positions —
// one ds_read_b32 per feature tile from the halo tile (per-lane offsets, a constant-1 and a zero word for the padding) — and
// the NT (NT + 1) / 2 products of the upper triangle take these registers as A and B operands alike: 3 LDS reads per 6 MFMAs
// where round 3's vector kernel read ~24 values per position and was LDS-bound (1.39 ms at config 5; this kernel: 0.33 ms).
// Partials [workgroup][tile pair][256] fp32 (a wave sums 2048 positions at config 5), summed in fp64 in a fixed order.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int CIN>
__global__ __launch_bounds__(256) void conv1_moments_mfma_k(const float* __restrict__ x, float* __restrict__ partials, int B, int F, int T) {
    constexpr int NK = 9 * CIN, NT = (NK + 1 + 15) / 16, NP = NT * (NT + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int F2 = F + 2;
    const int hn = (C1_GT + 2) * F2 * CIN;
    float* halo = smem;                                       // [(GT+2)][F2][CIN], then {1.0f, 0.0f}
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, kq = lane >> 4;
    const int tblocks = (T + C1_GT - 1) / C1_GT, ntiles = B * tblocks;
    int off[NT];
    bool real[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int fi = 16 * t + m;
        real[t] = fi < NK;
        const int k = real[t] ? fi : 0, tap = k / CIN, ci = k - tap * CIN, kh = tap / 3, kw = tap - kh * 3;
        off[t] = real[t] ? (kw * F2 + kh) * CIN + ci + kq * CIN : hn + (fi == NK ? 0 : 1);
    }
    f32x4_t acc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) acc[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (tid == 0) { halo[hn] = 1.f; halo[hn + 1] = 0.f; }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tblocks, t0 = (tile - b * tblocks) * C1_GT;
        __syncthreads();
        for (int j = tid; j < hn; j += 256) {                 // time fastest: contiguous in the NCHW input
            int tt = j % (C1_GT + 2), ff = (j / (C1_GT + 2)) % F2, ci = j / ((C1_GT + 2) * F2);
            const int t = t0 + tt - 1, f = ff - 1;
            float v = 0.f;
            if (t >= 0 && t < T && f >= 0 && f < F) v = x[(((size_t)b * CIN + ci) * F + f) * T + t];
            halo[(tt * F2 + ff) * CIN + ci] = v;
        }
        __syncthreads();
        for (int row = wave; row < C1_GT && t0 + row < T; row += 4) {
            for (int f0 = 0; f0 < F; f0 += 4) {
                const int base = (row * F2 + f0) * CIN;
                const bool live = f0 + kq < F;                 // a ragged last group: positions beyond the mel axis contribute nothing
                float a[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float v = halo[real[t] ? base + off[t] : off[t]];
                    a[t] = live ? v : 0.f;
                }
                int q = 0;
#pragma unroll
                for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                    for (int tb = ta; tb < NT; ++tb, ++q)
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], a[tb], acc[q], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    float* red = smem;                                        // [4 waves][NP][256]
#pragma unroll
    for (int q = 0; q < NP; ++q) *(f32x4_t*)(red + ((wave * NP + q) * 64 + lane) * 4) = acc[q];
    __syncthreads();
    float* dst = partials + (size_t)blockIdx.x * NP * 256;
    for (int i = tid; i < NP * 256; i += 256)
        dst[i] = (red[i] + red[NP * 256 + i]) + (red[2 * NP * 256 + i] + red[3 * NP * 256 + i]);
}

// fp64 sums of the partials in workgroup order -> the packed moment vector (S1, then the upper triangle row by row).  Element
// (row r, column c) of a 16 x 16 accumulator tile lies in lane 16 (r / 4) + c, register r % 4.
template <int CIN>
__global__ __launch_bounds__(64) void conv1_moments_mfma_sum_k(const float* __restrict__ partials, int nblk, double* __restrict__ gram_out) {
    constexpr int NK = 9 * CIN, NT = (NK + 1 + 15) / 16, NP = NT * (NT + 1) / 2;
    const int k = blockIdx.x, k2 = blockIdx.y, lane = threadIdx.x;     // one wave per entry (k, k2) of the upper triangle;
    if (k2 < k || (k2 == k && k == NK)) return;                          // column NK = the constant-1 feature = S1
    const int ta = k >> 4, tb = k2 >> 4, r = k & 15, c = k2 & 15;
    const int q = ta * NT - (ta * (ta - 1)) / 2 + (tb - ta);
    const int e = (q * 64 + 16 * (r >> 2) + c) * 4 + (r & 3);
    double sum = 0.0;
    for (int i = lane; i < nblk; i += 64) sum += (double)partials[(size_t)i * NP * 256 + e];
    sum = wave_sum_d(sum);
    if (lane == 0) gram_out[k2 == NK ? k : NK + k * NK - (k * (k - 1)) / 2 + (k2 - k)] = sum;
}

// the statistics row (sum y, sum y^2 per channel) from the moments: fp64 quadratic form
template <int CIN>
__global__ __launch_bounds__(256) void conv1_moments_stat_k(const double* __restrict__ G, const float* __restrict__ wp,
                                                            const float* __restrict__ bias, double count, int C, float* __restrict__ stat) {
    constexpr int NK = 9 * CIN;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // one wave per channel, lanes over the rows k
    if (c >= C) return;
    const double b = bias ? (double)bias[c] : 0.0;
    double ws1 = 0.0, q = 0.0;
    if (lane < NK) {
        const int k = lane;
        const double wk = (double)wp[((size_t)(k / CIN) * C + c) * CIN + (k % CIN)];
        ws1 = wk * G[k];
        for (int k2 = k; k2 < NK; ++k2) {
            const double wk2 = (double)wp[((size_t)(k2 / CIN) * C + c) * CIN + (k2 % CIN)];
            q += (k2 == k ? 1.0 : 2.0) * wk * wk2 * G[NK + k * NK - (k * (k - 1)) / 2 + (k2 - k)];
        }
    }
    ws1 = wave_sum_d(ws1);
    q = wave_sum_d(q);
    if (lane == 0) {
        stat[c] = (float)(count * b + ws1);
        stat[C + c] = (float)(q + 2.0 * b * ws1 + count * b * b);
    }
}

// ── backward of the recomputed block WITHOUT recomputing it (round 3) ──
// The old apply pass (MODE 3) recomputes both conv outputs of every window, the dropout hash and BatchNorm, forms dy for both
// rows and accumulates 2 x 9*CIN taps: ~240 vector instructions per output quad, and beside the MFMA weight gradients (where it
// has to run) vector work advances at 1/8 of its rate.  Everything except ONE sum per tap follows from quantities already at
// hand.  With v_k(pos) the 9*CIN shifted inputs, dy = sc (g_r - sg - xhat sgx), xhat = (y - mu) rs and y = b + sum_k' w_k' v_k':
//   dW_k = sum_pos v_k dy = sc [ R_k - sg S1_k - sgx rs ( b S1_k + sum_k' w_k' G_kk' - mu S1_k ) ]
// S1 and G are the input moments of the forward statistics pass (conv1_gram_k, kept in fp64), sg / sgx come from the fused
// reduction in the data gradient above, and R_k = sum over POOLED elements of g * v_k(arg-max position) is the only new sum:
// g = dout / (1-p) where the block's pooled output is > 0 (kept and gate open), the arg-max row is one bit per element written
// by the forward pass.  No conv, no hash, no BatchNorm arithmetic: 2 x 9*CIN FMAs per quad and two streamed tensors.
// The conv-bias gradient sum_pos dy = sc [ sum g - N sg - sgx rs (sum y - N mu) ] and, for a channel with gamma == 0, dgamma
// = rs ( b sum g + sum_k w_k R_k - mu sum g ) come out of the same sums in the assembling kernel.
// (launch bounds: <= 85 VGPRs, so that TWO workgroups fit into the 184 registers per lane the weight gradient leaves a SIMD)
#define RG_U 4   // positions per thread and round of conv1_rgrad_k (loads in flight)
template <int CIN>
__global__ __launch_bounds__(256, CIN == 1 ? 4 : 2) void conv1_rgrad_k(
    const float* __restrict__ x, const float* __restrict__ dout, const float* __restrict__ pooled,
    const unsigned char* __restrict__ bits, float* __restrict__ partials, int B, int F, int T, int C, float inv_keep) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int F2 = F + 2;
    float* halo = smem;                                       // [(TT+2)][F2][CIN]
    const int tid = threadIdx.x;
    const int C4 = C >> 2, nslots = 256 / C4;
    const int cg = tid % C4, slot = tid / C4;
    const bool active = slot < nslots;
    const int tblocks = (T + C1_TT - 1) / C1_TT, ntiles = B * tblocks;
    const int Tp = T >> 1;
    const float invF = 1.0f / (float)F;
    f32x4 a1 = {0, 0, 0, 0};
    f32x4 rk[9 * CIN];
#pragma unroll
    for (int k = 0; k < 9 * CIN; ++k) rk[k] = (f32x4){0, 0, 0, 0};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tblocks, t0 = (tile - b * tblocks) * C1_TT;
        __syncthreads();
        c1_stage<CIN>(halo, x, b, t0, F, T);
        __syncthreads();
        if (!active) continue;
        // This pass runs beside the MFMA weight gradient of the block above, whose fp32 MFMAs leave the VALU about one issue
        // slot per 64 cycles: its duration there is its vector-instruction count, so the loop is written for few instructions —
        // every global address is (tile base, wave-uniform) + op * C + 4 cg, the halo row pointer needs the one division
        // (sed_fdiv), and each tap is two fused multiply-adds per channel pair.
        const int tp0 = t0 >> 1;
        const int npos = ((Tp - tp0 < C1_TT / 2) ? Tp - tp0 : C1_TT / 2) * F;
        const size_t tbase = (((size_t)b * Tp + tp0) * F) * C;
        const char* const gb = (const char*)(dout + tbase);
        const char* const qb = (const char*)(pooled + tbase);
        const unsigned char* const bb = bits + (tbase >> 2);
        // RG_U positions per round with all their loads issued first: beside the weight gradient one workgroup (4 waves) fits a
        // CU, so the ~2-4 us of a loaded HBM round trip is hidden by loads in flight, not by other waves
        for (int op0 = slot; op0 < npos; op0 += RG_U * nslots) {
            f32x4 g[RG_U], pv[RG_U];
            unsigned bt[RG_U];
#pragma unroll
            for (int u = 0; u < RG_U; ++u) {
                const int op = op0 + u * nslots;
                const unsigned o4 = (unsigned)((op < npos ? op : npos - 1) * C4 + cg);     // clamped: a valid address either way
                g[u] = *(const f32x4*)(gb + o4 * 16u);
                pv[u] = *(const f32x4*)(qb + o4 * 16u);
                bt[u] = bb[o4];
            }
#pragma unroll
            for (int u = 0; u < RG_U; ++u) {
                const int op = op0 + u * nslots;
                const bool live = op < npos;
                const int opc = live ? op : npos - 1;
                const int tpl = sed_fdiv(opc, invF), f = opc - tpl * F;
                f32x4 g0, g1;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float gv = (live && pv[u][k] > 0.f) ? g[u][k] : 0.f;     // the 1 / (1-p) is applied to the partial sums below
                    const bool second = (bt[u] >> k) & 1u;
                    g0[k] = second ? 0.f : gv;
                    g1[k] = second ? gv : 0.f;
                }
                a1 += g0 + g1;
                const float* hp = halo + ((tpl * 2) * F2 + f) * CIN;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    float hv[4][CIN];                              // the four halo rows under the window's two conv rows
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int ci = 0; ci < CIN; ++ci) hv[rr][ci] = hp[(rr * F2 + kh) * CIN + ci];
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                        for (int ci = 0; ci < CIN; ++ci) {
                            f32x4& acc = rk[(kh * 3 + kw) * CIN + ci];
#pragma unroll
                            for (int k = 0; k < 4; ++k) acc[k] = __builtin_fmaf(hv[kw + 1][ci], g1[k], __builtin_fmaf(hv[kw][ci], g0[k], acc[k]));
                        }
                }
            }
        }
    }
    a1 *= inv_keep;
#pragma unroll
    for (int k = 0; k < 9 * CIN; ++k) rk[k] *= inv_keep;
    // block reduction in fixed slot order, ONE value per round through a 4 KB buffer: this pass runs beside the MFMA weight
    // gradient, which leaves a CU 28 KB of LDS — with the 20 KB buffer of conv1_fused_k only one workgroup fits there (an eighth
    // of the stand-alone occupancy: that, not issue contention, is why passes crawl beside that kernel)
    constexpr int NV = 1 + 9 * CIN, NVC = 1;
    float* red = smem;                                        // [nslots][NVC][C]
#pragma unroll
    for (int v0 = 0; v0 < NV; v0 += NVC) {
        __syncthreads();
        if (active) {
#pragma unroll
            for (int j = 0; j < NVC; ++j) {
                const int v = v0 + j;
                if (v < NV) *(f32x4*)(red + (slot * NVC + j) * C + cg * 4) = (v == 0) ? a1 : rk[(v - 1) < 9 * CIN ? (v - 1) : 0];
            }
        }
        __syncthreads();
        const int nv = (NV - v0 < NVC) ? NV - v0 : NVC;
        for (int i = tid; i < nv * C; i += 256) {
            int j = i / C, c = i - j * C;
            float a = 0.f;
            for (int s2 = 0; s2 < nslots; ++s2) a += red[(s2 * NVC + j) * C + c];
            partials[((size_t)blockIdx.x * C + c) * NV + (v0 + j)] = a;      // [workgroup][channel][value]: the assembling kernel reads a channel's NV values as one run
        }
    }
}

// ── the same sums on the matrix cores (round 4; 3 and 4 input channels, C a multiple of 64) ──
// R[k][c] = sum over pooled positions of g~[pos][c] * v_k(arg-max row of (pos, c)) is a product V^T G~ once the routed gradient
// is split by the arg-max bit: one v_mfma_f32_32x32x2_f32 takes the two un-pooled time rows of a pooled position as its two k
// indices — A[feature][row parity] = the shifted input at that row (one ds_read_b32 per lane from the halo tile; the constant-1
// feature behind the taps gives sum g~ = partial value 0), B[row parity][channel] = g~ where the bit names that row, else 0.
// Per pooled position: 2 feature tiles x 4 channel tiles = 8 MFMAs against ~80 vector instructions per channel quad in
// conv1_rgrad_k<4>, whose 2.1 ms of vector work trailed the last weight gradient at config 5 (the fp32 MFMA shares the vector
// ALU: a vector kernel beside it costs its own stand-alone time).  A wave owns one pooled row of the tile and all 128 channels of
// a 64-channel group (blockIdx.y): 4 accumulator tiles, so that three waves per SIMD fit; the four waves of a workgroup are
// summed through LDS, one tile per round.
#define C1_RM_GT 8            // un-pooled time rows per tile: four pooled rows, one per wave
#define C1_RM_CT 2            // 32-channel tiles per wave (blockIdx.y = a group of 64 channels): 4 accumulator tiles at 4 input channels
#define C1_RM_D 8             // mel positions of gradient / pooled / bit loads in flight per lane
template <int CIN>
__global__ __launch_bounds__(256) void conv1_rgrad_mfma_k(
    const float* __restrict__ x, const float* __restrict__ dout, const float* __restrict__ pooled,
    const unsigned char* __restrict__ bits, float* __restrict__ partials, int B, int F, int T, int C, float inv_keep) {
    constexpr int NK = 9 * CIN, NV = 1 + NK, NFT = (NV + 31) / 32, CT = C1_RM_CT, D = C1_RM_D;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int F2 = F + 2, Tp = T >> 1;
    const int hn = (C1_RM_GT + 2) * F2 * CIN;
    float* halo = smem;                                       // [(GT+2)][F2][CIN], then {1.0f, 0.0f}
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, kk = lane >> 5;
    const int tblocks = (T + C1_RM_GT - 1) / C1_RM_GT, ntiles = B * tblocks;
    const int cg = blockIdx.y * (32 * CT);
    int aoff[NFT];
    bool areal[NFT];
#pragma unroll
    for (int ta = 0; ta < NFT; ++ta) {
        const int fi = 32 * ta + i;
        areal[ta] = fi < NK;
        const int k = areal[ta] ? fi : 0, tap = k / CIN, ci = k - tap * CIN, kh = tap / 3, kw = tap - kh * 3;
        aoff[ta] = areal[ta] ? ((kw + kk) * F2 + kh) * CIN + ci : hn + (fi == NK ? 0 : 1);
    }
    f32x16 acc[NFT][CT];
#pragma unroll
    for (int ta = 0; ta < NFT; ++ta)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[ta][ct][j] = 0.f;
    if (tid == 0) { halo[hn] = 1.f; halo[hn + 1] = 0.f; }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tblocks, t0 = (tile - b * tblocks) * C1_RM_GT;
        __syncthreads();
        for (int j = tid; j < hn; j += 256) {                 // time fastest: contiguous in the NCHW input
            int tt = j % (C1_RM_GT + 2), ff = (j / (C1_RM_GT + 2)) % F2, ci = j / ((C1_RM_GT + 2) * F2);
            const int t = t0 + tt - 1, f = ff - 1;
            float v = 0.f;
            if (t >= 0 && t < T && f >= 0 && f < F) v = x[(((size_t)b * CIN + ci) * F + f) * T + t];
            halo[(tt * F2 + ff) * CIN + ci] = v;
        }
        __syncthreads();
        const int tp = (t0 >> 1) + wave;                       // this wave's pooled row
        if (tp < Tp) {
            const size_t prow = ((size_t)b * Tp + tp) * F;
            const float* dp = dout + prow * C + cg + i;
            const float* pp = pooled + prow * C + cg + i;
            const unsigned char* bp = bits + prow * (C >> 2) + ((cg + i) >> 2);
            const int rowbase = (2 * wave) * F2 * CIN;         // halo origin of un-pooled row 2 (tp - t0/2), mel 0 (tap (0,0) = one row / column up)
            // the gradient, the pooled value and the bit byte of D mel positions are in flight per lane: an un-pipelined load per
            // position made this loop a chain of HBM round trips (first version: 13 ms instead of 1)
            float dv[D][CT], qv[D][CT];
            unsigned bv8[D][CT];
            auto load = [&](int slot, int f) {
                const int fc = f < F ? f : F - 1;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    dv[slot][ct] = dp[(size_t)fc * C + ct * 32];
                    qv[slot][ct] = pp[(size_t)fc * C + ct * 32];
                    bv8[slot][ct] = bp[(size_t)fc * (C >> 2) + ct * 8];
                }
            };
#pragma unroll
            for (int sl = 0; sl < D; ++sl) load(sl, sl);
            for (int f0 = 0; f0 < F; f0 += D) {
#pragma unroll
                for (int sl = 0; sl < D; ++sl) {
                    const int f = f0 + sl;
                    const int base = rowbase + (f < F ? f : F - 1) * CIN;
                    float a[NFT];
#pragma unroll
                    for (int ta = 0; ta < NFT; ++ta) a[ta] = halo[areal[ta] ? base + aoff[ta] : aoff[ta]];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const float bv = (f < F && qv[sl][ct] > 0.f && (int)((bv8[sl][ct] >> (i & 3)) & 1u) == kk) ? dv[sl][ct] * inv_keep : 0.f;
#pragma unroll
                        for (int ta = 0; ta < NFT; ++ta)
                            acc[ta][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], bv, acc[ta][ct], 0, 0, 0);
                    }
                    load(sl, f + D);
                }
            }
        }
    }
    // four waves -> one partial row: a 32 x 32 tile per round through LDS (16 KB), fixed order
    float* red = smem;                                        // [4 waves][32 rows][32 cols]
#pragma unroll
    for (int ta = 0; ta < NFT; ++ta)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; ++j) red[(wave * 32 + (j & 3) + 8 * (j >> 2) + 4 * kk) * 32 + i] = acc[ta][ct][j];
            __syncthreads();
            for (int e = tid; e < 1024; e += 256) {
                const int row = e >> 5, col = e & 31, fi = 32 * ta + row;
                if (fi <= NK) {
                    const float sum = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
                    partials[((size_t)blockIdx.x * C + cg + ct * 32 + col) * NV + (fi == NK ? 0 : 1 + fi)] = sum;
                }
            }
        }
}

// index of G(k, k2), k <= k2, in the moment vector [S1 (NK) | upper triangle row by row] of conv1_gram_k
__device__ __forceinline__ int c1_gidx(int k, int k2, int NK) { return NK + k * NK - (k * (k - 1)) / 2 + (k2 - k); }

// one workgroup per output channel: R and sum g summed over the partial rows in fp64 (fixed order), then the closed form
__global__ __launch_bounds__(256) void conv1_wgrad_assemble_k(
    const float* __restrict__ part, int rows, int Cin, int C, const double* __restrict__ gram, const float* __restrict__ wp,
    const float* __restrict__ bias, const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ scale,
    const float* __restrict__ sum_g, const float* __restrict__ sum_gx, double count, float* __restrict__ dw, float* __restrict__ db,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ dgamma) {
    const int co = blockIdx.x, NK = 9 * Cin, NV = 1 + NK;
    __shared__ double sv[37];                                 // [0] sum g, [1..NK] R_k
    __shared__ double sw[4][37];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    {   // a thread sums whole rows: the NV values of (workgroup r, channel co) are one 40..148-byte run
        double a[37];
#pragma unroll
        for (int v = 0; v < 37; ++v) a[v] = 0.0;
        for (int r = threadIdx.x; r < rows; r += 256) {
            const float* pr = part + ((size_t)r * C + co) * NV;
#pragma unroll
            for (int v = 0; v < 37; ++v)
                if (v < NV) a[v] += (double)pr[v];
        }
#pragma unroll
        for (int v = 0; v < 37; ++v) {
            if (v < NV) {
                const double t = wave_sum_d(a[v]);
                if (lane == 0) sw[wv][v] = t;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < NV) sv[threadIdx.x] = (sw[0][threadIdx.x] + sw[1][threadIdx.x]) + (sw[2][threadIdx.x] + sw[3][threadIdx.x]);
    __syncthreads();
    if (wv != 0) return;
    const double b = bias ? (double)bias[co] : 0.0, mu = mean[co], rs = rstd[co], sc = scale[co];
    // sum g * xhat of this channel EXACTLY, from the same R_k the weight gradient is made of:
    //   sum g xhat = rstd (sum g y - mean sum g),   sum g y = b sum g + sum_k w_k R_k   (y = b + sum_k w_k v_k at the arg-max).
    // The value the data gradient's epilogue forms from the pooled output — xhat = (z - beta)/gamma — loses eps |beta / gamma| for a
    // small |gamma| and does not exist for gamma == 0 (round-3 advisor); this block keeps no conv output to fall back on.  With
    // sum_gx == NULL the block's own sums are used for EVERY channel, in the weight / bias gradient below and as dgamma (the
    // plan's single-device path).  With sum_gx given (sums of a recomputing reduce pass, or all-reduced over ranks for
    // synchronised BatchNorm, where the local R_k are not the global ones) it is used, and the own value only replaces dgamma of
    // gamma == 0 channels, as before.
    double wr = 0.0, ws1 = 0.0;
    for (int k2 = 0; k2 < NK; ++k2) {
        const double w2 = (double)wp[((size_t)(k2 / Cin) * C + co) * Cin + (k2 % Cin)];
        ws1 += w2 * gram[k2];
        wr += w2 * sv[1 + k2];
    }
    const double sgx_sum = rs * (b * sv[0] + wr - mu * sv[0]);
    const bool own = sum_gx == nullptr;
    const double sg = (double)sum_g[co] / count, sgx = (own ? sgx_sum : (double)sum_gx[co]) / count;
    if (lane < NK) {
        const int k = lane;
        double wg = 0.0;                                       // sum_k' w_k' G(k, k')
        for (int k2 = 0; k2 < NK; ++k2) {
            const double w2 = (double)wp[((size_t)(k2 / Cin) * C + co) * Cin + (k2 % Cin)];
            wg += w2 * gram[k <= k2 ? c1_gidx(k, k2, NK) : c1_gidx(k2, k, NK)];
        }
        const double s1 = gram[k];
        const double val = sc * (sv[1 + k] - sg * s1 - sgx * rs * (b * s1 + wg - mu * s1));
        const int tap = k / Cin, ci = k - tap * Cin;
        dw[((size_t)co * Cin + ci) * 9 + tap] = (float)val;
    }
    if (lane == 63) {
        const double sum_y = count * b + ws1;
        db[co] = (float)(sc * (sv[0] - count * sg - sgx * rs * (sum_y - count * mu)));
        if (dgamma && (own || (gamma && beta && gamma[co] == 0.f && beta[co] > 0.f))) dgamma[co] = (float)sgx_sum;
    }
}

static size_t c1_lds(int Cin, int F, int C, int mode) {
    size_t halo = (size_t)(C1_TT + 2) * (F + 2) * Cin * sizeof(float);
    size_t nv = mode == 3 ? 5 : 2;                            // values per reduction round (NVC in the kernel)
    size_t red = (size_t)256 * 4 * nv * sizeof(float);        // nslots*NVC*C = 256*4*NVC
    return halo > red ? halo : red;
}

static int c1_shape_ok(int Cin, int F, int T, int C, int pool_f, int pool_t, int max_cin);
extern "C" int sed_conv1_fused_supported(int Cin, int F, int T, int C, int pool_f, int pool_t) {
    return c1_shape_ok(Cin, F, T, C, pool_f, pool_t, 2);
}
static int c1_shape_ok(int Cin, int F, int T, int C, int pool_f, int pool_t, int max_cin) {
    if (Cin < 1 || Cin > max_cin || C < 4 || F < 1 || T < 1 || C % 4 != 0 || C / 4 > 256 || (256 % (C / 4)) != 0) return 0;
    if (pool_t < 1 || pool_f < 1 || C1_TT % pool_t != 0 || T % C1_TT != 0 || F % pool_f != 0 || T % pool_t != 0) return 0;
    if (c1_lds(Cin, F, C, 3) > 150 * 1024) return 0;
    return 1;
}

extern "C" int sed_conv1_fused_rows(int B, int T) {
    long n = (long)B * ((T + C1_TT - 1) / C1_TT);
    return (int)(n < C1_MAXBLOCKS ? n : C1_MAXBLOCKS);
}

template <int MODE>
static int c1_launch(const float* x, const float* wp, const float* bias, const float* scale, const float* shift,
                     const float* mean, const float* rstd, const float* sum_g, const float* sum_gx, const float* dout,
                     float* out, float* partials, int B, int Cin, int F, int T, int C, int pf, int pt, float drop_p,
                     uint64_t seed, const uint64_t* seed_dev, hipStream_t s) {
    size_t lds = c1_lds(Cin, F, C, MODE);
    int grid = sed_conv1_fused_rows(B, T);
    const bool p12 = (pf == 1 && pt == 2);
#define C1_LAUNCH(CIN_, P12_)                                                                                          \
    do {                                                                                                               \
        if (lds > 48 * 1024)                                                                                           \
            (void)hipFuncSetAttribute((const void*)conv1_fused_k<CIN_, MODE, P12_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        conv1_fused_k<CIN_, MODE, P12_><<<grid, 256, lds, s>>>(x, wp, bias, scale, shift, mean, rstd, sum_g, sum_gx, dout, out, \
                                                              partials, B, F, T, C, pf, pt, drop_p, seed, seed_dev);  \
    } while (0)
    if (Cin == 1) { if (p12) C1_LAUNCH(1, true); else C1_LAUNCH(1, false); }
    else if (Cin == 2) { if (p12) C1_LAUNCH(2, true); else C1_LAUNCH(2, false); }
    else if ((MODE == 1 || MODE == 4) && p12) {           // 3 / 4 input channels: the forward pass only (statistics from the blocked moment
        if (Cin == 3) C1_LAUNCH(3, true);  // kernel, backward through sed_conv3x3_dgrad_bnred + sed_conv1_bwd_wgrad)
        else C1_LAUNCH(4, true);
    } else {
        sed_set_error("conv1: %d input channels are supported by the forward pass with the (1,2) pool only", Cin);
        return SED_EUNSUPPORTED;
    }
#undef C1_LAUNCH
    return 0;
}

#define C1_CHECK(who)                                                                                             \
    SED_REQUIRE(sed_conv1_fused_supported(Cin, F, T, C, pf, pt), who ": shape Cin=%d F=%d T=%d C=%d pool=(%d,%d) is not " \
                "supported by the fused first block", Cin, F, T, C, pf, pt)

extern "C" size_t sed_conv1_stats_workspace_bytes(int B, int Cin, int T) {
    const int nk = 9 * Cin;
    if (Cin > 2) {      // the MFMA Gram kernel: up to 1024 workgroups x (tile pairs) x 256 floats, + the fp64 moment vector
        const int nt = (nk + 1 + 15) / 16, np = nt * (nt + 1) / 2;
        return (size_t)C1_MM_BLOCKS * np * 256 * sizeof(float) + (size_t)(nk + nk * (nk + 1) / 2) * sizeof(double);
    }
    return (size_t)256 * (nk + nk * (nk + 1) / 2) * sizeof(float);      // at most 256 partial rows (one per workgroup)
}

extern "C" int sed_conv1_stats(const float* x, const float* wp, const float* bias, float* stat_partials, void* workspace,
                               int B, int Cin, int F, int T, int C, double* gram_out, void* stream) {
    SED_REQUIRE(x && wp && stat_partials && workspace, "conv1_stats: null pointer");
    SED_REQUIRE(c1_shape_ok(Cin, F, T, C, 1, 1, 4), "conv1_stats: shape Cin=%d F=%d T=%d C=%d is not supported", Cin, F, T, C);
    hipStream_t s = as_stream(stream);
    SedProfScope prof(SED_K_CONV_SMALL_FWD, s, 4.0 * B * Cin * (double)F * T);
    if (Cin > 2) {
        const int nk = 9 * Cin, nt = (nk + 1 + 15) / 16, np = nt * (nt + 1) / 2;
        int nb = B * ((T + C1_GT - 1) / C1_GT);
        if (nb > C1_MM_BLOCKS) nb = C1_MM_BLOCKS;
        size_t lds = ((size_t)(C1_GT + 2) * (F + 2) * Cin + 4) * sizeof(float), red = (size_t)4 * np * 256 * sizeof(float);
        if (red > lds) lds = red;
        SED_REQUIRE(lds <= 150 * 1024, "conv1_stats: F=%d Cin=%d needs %zu B of LDS", F, Cin, lds);
        const double cnt = (double)B * T * F;
        // the fp64 moments go to the caller's array, or behind the partial rows in the workspace when the caller does not want them
        double* G = gram_out ? gram_out : (double*)((float*)workspace + (size_t)C1_MM_BLOCKS * np * 256);
#define C1_MOM(CIN_)                                                                                                              \
    do {                                                                                                                          \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)conv1_moments_mfma_k<CIN_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        conv1_moments_mfma_k<CIN_><<<nb, 256, lds, s>>>(x, (float*)workspace, B, F, T);                                             \
        conv1_moments_mfma_sum_k<CIN_><<<dim3(nk + 1, nk + 1), 64, 0, s>>>((const float*)workspace, nb, G);                       \
        conv1_moments_stat_k<CIN_><<<cdiv(C, 4), 256, 0, s>>>(G, wp, bias, cnt, C, stat_partials);                                 \
    } while (0)
        if (Cin == 3) C1_MOM(3); else C1_MOM(4);
#undef C1_MOM
        SED_LAUNCH_CHECK("conv1_moments");
        return 0;
    }
    const int pf = 1, pt = 1;
    C1_CHECK("conv1_stats");
    int grid = (B * ((T + C1_GT - 1) / C1_GT) + C1_GR - 1) / C1_GR;
    if (grid > 256) grid = 256;                              // one workgroup per CU; at most 256 partial rows for the finalisation
    const int nk = 9 * Cin, ng = nk + nk * (nk + 1) / 2;
    size_t lds = (size_t)C1_GR * (C1_GT + 2) * (F + 2) * Cin * sizeof(float), red = (size_t)ng * 4 * sizeof(float);
    SED_REQUIRE(lds <= 150 * 1024, "conv1_stats: F=%d Cin=%d needs %zu B of LDS", F, Cin, lds);
    if (red > lds) lds = red;
    const double count = (double)B * T * F;
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void*)conv1_gram_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)conv1_gram_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (Cin == 1) {
        conv1_gram_k<1><<<grid, 256, lds, s>>>(x, (float*)workspace, B, F, T);
        SED_LAUNCH_CHECK("conv1_gram");
        conv1_gram_finalize_k<1><<<1, 256, 0, s>>>((const float*)workspace, grid, wp, bias, count, C, stat_partials, gram_out);
    } else {
        conv1_gram_k<2><<<grid, 256, lds, s>>>(x, (float*)workspace, B, F, T);
        SED_LAUNCH_CHECK("conv1_gram");
        conv1_gram_finalize_k<2><<<1, 256, 0, s>>>((const float*)workspace, grid, wp, bias, count, C, stat_partials, gram_out);
    }
    SED_LAUNCH_CHECK("conv1_gram_finalize");
    return 0;
}

extern "C" int sed_conv1_bn_relu_pool_drop_fwd(const float* x, const float* wp, const float* bias, const float* scale,
                                               const float* shift, float* out, int B, int Cin, int F, int T, int C,
                                               int pf, int pt, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                                               unsigned char* argmax_bits, void* stream) {
    SED_REQUIRE(x && wp && scale && shift && out, "conv1_fwd: null pointer");
    SED_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "conv1_fwd: drop_p=%f out of [0,1)", drop_p);
    SED_REQUIRE(c1_shape_ok(Cin, F, T, C, pf, pt, (pf == 1 && pt == 2) ? 4 : 2), "conv1_fwd: shape Cin=%d F=%d T=%d C=%d pool=(%d,%d) is not "
                "supported by the fused first block", Cin, F, T, C, pf, pt);
    hipStream_t s = as_stream(stream);
    SedProfScope prof(SED_K_BN_FWD, s, 4.0 * B * C * (double)(T / pt) * (F / pf));
    SED_REQUIRE(!argmax_bits || (pf == 1 && pt == 2), "conv1_fwd: arg-max bits exist for the (1,2) pool only");
    SED_TRY(c1_launch<1>(x, wp, bias, scale, shift, nullptr, nullptr, nullptr, nullptr, nullptr, out, reinterpret_cast<float*>(argmax_bits), B, Cin, F, T, C, pf, pt, drop_p, seed, seed_dev, s));
    SED_LAUNCH_CHECK("conv1_fwd");
    return 0;
}

extern "C" int sed_conv1_route(const float* x, const float* wp, const float* bias, const float* scale, const float* shift,
                               unsigned char* route, int B, int Cin, int F, int T, int C, int pf, int pt, void* stream) {
    SED_REQUIRE(x && wp && scale && shift && route, "conv1_route: null pointer");
    SED_REQUIRE(c1_shape_ok(Cin, F, T, C, pf, pt, (pf == 1 && pt == 2) ? 4 : 2), "conv1_route: shape Cin=%d F=%d T=%d C=%d pool=(%d,%d) is not "
                "supported by the fused first block", Cin, F, T, C, pf, pt);
    SED_REQUIRE(pf * pt <= 254, "conv1_route: a %dx%d window does not fit the one-byte code", pf, pt);
    SED_TRY(c1_launch<4>(x, wp, bias, scale, shift, nullptr, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<float*>(route), nullptr,
                         B, Cin, F, T, C, pf, pt, 0.f, 0, nullptr, as_stream(stream)));
    SED_LAUNCH_CHECK("conv1_route");
    return 0;
}

extern "C" int sed_conv1_bwd_reduce(const float* x, const float* wp, const float* bias, const float* dout,
                                    const float* scale, const float* shift, const float* mean, const float* rstd,
                                    float* partials, int B, int Cin, int F, int T, int C, int pf, int pt, float drop_p,
                                    uint64_t seed, const uint64_t* seed_dev, void* stream) {
    SED_REQUIRE(x && wp && dout && scale && shift && mean && rstd && partials, "conv1_bwd_reduce: null pointer");
    C1_CHECK("conv1_bwd_reduce");
    hipStream_t s = as_stream(stream);
    SedProfScope prof(SED_K_BN_BWD_REDUCE, s, 4.0 * B * C * (double)(T / pt) * (F / pf));
    c1_launch<2>(x, wp, bias, scale, shift, mean, rstd, nullptr, nullptr, dout, nullptr, partials, B, Cin, F, T, C, pf, pt, drop_p, seed, seed_dev, s);
    SED_LAUNCH_CHECK("conv1_bwd_reduce");
    return 0;
}

extern "C" size_t sed_conv1_bwd_apply_workspace_bytes(int B, int Cin, int T, int C) {
    return (size_t)sed_conv1_fused_rows(B, T) * (2 + 9 * Cin) * C * sizeof(float);
}

extern "C" int sed_conv1_bwd_apply_wgrad(const float* x, const float* wp, const float* bias, const float* dout,
                                         const float* scale, const float* shift, const float* mean, const float* rstd,
                                         const float* sum_g, const float* sum_gx, float* dw_oihw, float* dbias,
                                         void* workspace, int B, int Cin, int F, int T, int C, int pf, int pt,
                                         float drop_p, uint64_t seed, const uint64_t* seed_dev,
                                         const float* gamma, const float* beta, float* dgamma, void* stream) {
    SED_REQUIRE(x && wp && dout && scale && shift && mean && rstd && sum_g && sum_gx && dw_oihw && dbias && workspace,
                "conv1_bwd_apply_wgrad: null pointer");
    C1_CHECK("conv1_bwd_apply_wgrad");
    hipStream_t s = as_stream(stream);
    SedProfScope prof(SED_K_BN_BWD_APPLY, s, 4.0 * B * C * (double)(T / pt) * (F / pf));
    c1_launch<3>(x, wp, bias, scale, shift, mean, rstd, sum_g, sum_gx, dout, nullptr, (float*)workspace, B, Cin, F, T, C, pf, pt, drop_p, seed, seed_dev, s);
    SED_LAUNCH_CHECK("conv1_bwd_apply_wgrad");
    int rows = sed_conv1_fused_rows(B, T), n = (2 + 9 * Cin) * C;
    conv1_wgrad_reduce_k<<<cdiv(n, 8), 256, 0, s>>>((const float*)workspace, rows, Cin, C, dw_oihw, dbias, gamma, beta, dgamma);
    SED_LAUNCH_CHECK("conv1_wgrad_reduce");
    return 0;
}

// ── the backward of the recomputed block from its pooled output, arg-max bits and input moments (see conv1_rgrad_k) ──
extern "C" int sed_conv1_rgrad_supported(int Cin, int F, int T, int C, int pf, int pt) {
    return pf == 1 && pt == 2 && c1_shape_ok(Cin, F, T, C, pf, pt, 4);
}
extern "C" size_t sed_conv1_moments_doubles(int Cin) { const int nk = 9 * Cin; return (size_t)nk + (size_t)nk * (nk + 1) / 2; }
// partial rows of conv1_rgrad_k: two persistent workgroups per CU are plenty for a streaming pass, and the assembling kernel
// that sums them runs beside an MFMA kernel, where every row costs
static int c1_rgrad_rows(int B, int T) {          // (512 rows instead of 2048: the pass beside the weight gradient 1.14 -> 1.58 ms)
    return sed_conv1_fused_rows(B, T);
}
// The assembling half of sed_conv1_bwd_wgrad alone, from partial sums another kernel formed: partials [rows][C][1 + 9 Cin]
// = (sum g, R_k) — sed_conv3x3_dgrad_bnred_rg writes them from the epilogue of the data gradient of the block above.
extern "C" int sed_conv1_bwd_wgrad_assemble(const float* partials, int rows, const double* moments, const float* wp, const float* bias,
                                            const float* mean, const float* rstd, const float* scale, const float* sum_g,
                                            const float* sum_gx, float* dw_oihw, float* dbias, int B, int Cin, int F, int T, int C,
                                            const float* gamma, const float* beta, float* dgamma, void* stream) {
    SED_REQUIRE(partials && moments && wp && mean && rstd && scale && sum_g && dw_oihw && dbias, "conv1_bwd_wgrad_assemble: null pointer");
    SED_REQUIRE(rows > 0 && B > 0 && Cin >= 1 && Cin <= 4 && F > 0 && T > 0 && C > 0, "conv1_bwd_wgrad_assemble: bad shape");
    conv1_wgrad_assemble_k<<<C, 256, 0, as_stream(stream)>>>(partials, rows, Cin, C, moments, wp, bias, mean, rstd, scale, sum_g, sum_gx,
                                                             (double)B * T * F, dw_oihw, dbias, gamma, beta, dgamma);
    SED_LAUNCH_CHECK("conv1_wgrad_assemble");
    return 0;
}

extern "C" size_t sed_conv1_bwd_wgrad_workspace_bytes(int B, int Cin, int T, int C) {
    return (size_t)c1_rgrad_rows(B, T) * (1 + 9 * Cin) * C * sizeof(float);
}
extern "C" int sed_conv1_bwd_wgrad(const float* x, const float* dout, const float* pooled, const unsigned char* argmax_bits,
                                   const double* moments, const float* wp, const float* bias, const float* mean, const float* rstd,
                                   const float* scale, const float* sum_g, const float* sum_gx, float* dw_oihw, float* dbias,
                                   void* workspace, int B, int Cin, int F, int T, int C, float drop_p,
                                   const float* gamma, const float* beta, float* dgamma, void* stream) {
    SED_REQUIRE(x && dout && pooled && argmax_bits && moments && wp && mean && rstd && scale && sum_g && dw_oihw && dbias && workspace,
                "conv1_bwd_wgrad: null pointer");
    SED_REQUIRE(sed_conv1_rgrad_supported(Cin, F, T, C, 1, 2), "conv1_bwd_wgrad: shape Cin=%d F=%d T=%d C=%d is not supported", Cin, F, T, C);
    SED_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "conv1_bwd_wgrad: drop_p=%f out of [0,1)", drop_p);
    hipStream_t s = as_stream(stream);
    SedProfScope prof(SED_K_BN_BWD_APPLY, s, 2.0 * 4.0 * B * C * (double)(T / 2) * F);
    size_t lds = (size_t)(C1_TT + 2) * (F + 2) * Cin * sizeof(float);      // halo tile, or the one-value reduction buffer
    if (lds < (size_t)256 * 4 * sizeof(float)) lds = (size_t)256 * 4 * sizeof(float);
    int grid = c1_rgrad_rows(B, T);
    const float inv_keep = 1.f / (1.f - drop_p);
    const size_t lds_m = ((size_t)(C1_RM_GT + 2) * (F + 2) * Cin + 4) * sizeof(float) > (size_t)4096 * sizeof(float)
                             ? ((size_t)(C1_RM_GT + 2) * (F + 2) * Cin + 4) * sizeof(float) : (size_t)4096 * sizeof(float);
    if (Cin >= 3 && C % (32 * C1_RM_CT) == 0 && lds_m <= 64 * 1024) {
        // 3 / 4 input channels: the sums on the matrix cores (conv1_rgrad_mfma_k); one partial row per workgroup as before
        int g2 = B * ((T + C1_RM_GT - 1) / C1_RM_GT);
        if (g2 < grid) grid = g2;
        if (Cin == 3) {
            if (lds_m > 48 * 1024) (void)hipFuncSetAttribute((const void*)conv1_rgrad_mfma_k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
            conv1_rgrad_mfma_k<3><<<dim3(grid, C / (32 * C1_RM_CT)), 256, lds_m, s>>>(x, dout, pooled, argmax_bits, (float*)workspace, B, F, T, C, inv_keep);
        } else {
            if (lds_m > 48 * 1024) (void)hipFuncSetAttribute((const void*)conv1_rgrad_mfma_k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
            conv1_rgrad_mfma_k<4><<<dim3(grid, C / (32 * C1_RM_CT)), 256, lds_m, s>>>(x, dout, pooled, argmax_bits, (float*)workspace, B, F, T, C, inv_keep);
        }
    } else {
#define C1_RGRAD(CIN_)                                                                                                        \
    do {                                                                                                                      \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)conv1_rgrad_k<CIN_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        conv1_rgrad_k<CIN_><<<grid, 256, lds, s>>>(x, dout, pooled, argmax_bits, (float*)workspace, B, F, T, C, inv_keep);    \
    } while (0)
        switch (Cin) {
            case 1: C1_RGRAD(1); break;
            case 2: C1_RGRAD(2); break;
            case 3: C1_RGRAD(3); break;
            default: C1_RGRAD(4); break;
        }
#undef C1_RGRAD
    }
    SED_LAUNCH_CHECK("conv1_rgrad");
    conv1_wgrad_assemble_k<<<C, 256, 0, s>>>((const float*)workspace, grid, Cin, C, moments, wp, bias, mean, rstd, scale, sum_g, sum_gx,
                                           (double)B * T * F, dw_oihw, dbias, gamma, beta, dgamma);
    SED_LAUNCH_CHECK("conv1_wgrad_assemble");
    return 0;
}
