// bnpool.hip — BatchNorm2d statistics + fused normalise/ReLU/max-pool/dropout, forward and backward.
//
// Replaces aten::native_batch_norm(+_backward), clamp_min/threshold_backward,
// max_pool2d_with_indices(+_backward) and bernoulli_/mul reached from reference
// sed.py:89-92,107 and crnn_lightning.py:48-52.  All kernels are pure HBM streaming passes over the
// channels-last conv output: one wave covers 64 lanes x 16 B = two full 128-channel rows.
// Reductions are two-stage, fixed order (deterministic=True, train_lightning.py:47), no float atomics.
#include "common.h"

#define BN_MAX_BLOCKS 1024

// ───────────────────────── statistics ─────────────────────────
// partials [rows][2][C] -> per-channel sums.  The pass is pure latency (a few MB scattered over other XCDs' L2), so it is laid
// out for loads in flight, not for coalescing: a workgroup owns TWO channels, its 1024 threads are 512 row slices, and a thread's
// 2 x rows/512 loads (16 at the 4096 rows of config 2) are all issued before the first is used — one or two round trips
// instead of eight (round 3: 21 -> 9 us on the forward chain).  Fixed summation order: slice, then lane tree, then wave order.
__device__ __forceinline__ void bn_two_channel_sums(const float* __restrict__ part, int rows, int C, int c, bool live,
                                                    double (*sred)[2][2], double* A_out, double* Q_out) {
    const int sl = threadIdx.x >> 1, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double a = 0.0, q = 0.0;
    if (live) {
#pragma unroll 8
        for (int r = sl; r < rows; r += 512) {
            const float* p0 = part + (size_t)r * 2 * C + c;
            a += (double)p0[0];
            q += (double)p0[C];
        }
    }
#pragma unroll
    for (int o = 2; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); q += __shfl_xor(q, o, 64); }       // lanes of equal parity = one channel
    if (lane < 2) { sred[wv][lane][0] = a; sred[wv][lane][1] = q; }
    __syncthreads();
    double A = 0.0, Q = 0.0;
    if (threadIdx.x < 2) {
        for (int w = 0; w < 16; ++w) { A += sred[w][threadIdx.x][0]; Q += sred[w][threadIdx.x][1]; }
    }
    *A_out = A; *Q_out = Q;
}

__global__ __launch_bounds__(1024) void bn_finalize_train_k(
    const float* __restrict__ part, int rows, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
    float eps, float* __restrict__ mean_o, float* __restrict__ rstd_o, float* __restrict__ scale_o,
    float* __restrict__ shift_o) {
    __shared__ double sred[16][2][2];
    const int c = blockIdx.x * 2 + (threadIdx.x & 1);
    double A, Q;
    bn_two_channel_sums(part, rows, C, c, c < C, sred, &A, &Q);
    if (threadIdx.x < 2 && c < C) {
        double m = A / count;
        double var = Q / count - m * m;
        if (var < 0.0) var = 0.0;
        float rstd = (float)(1.0 / sqrt(var + (double)eps));
        float g = gamma[c], bt = beta[c];
        mean_o[c] = (float)m;
        rstd_o[c] = rstd;
        float sc = g * rstd;
        scale_o[c] = sc;
        shift_o[c] = bt - (float)m * sc;
        if (rmean) {
            double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
        }
    }
}

extern "C" int sed_bn_finalize_train(const float* part, int rows, int C, double count, const float* gamma,
                                     const float* beta, float* rmean, float* rvar, float momentum, float eps,
                                     float* mean, float* rstd, float* scale, float* shift, void* stream) {
    SED_REQUIRE(part && gamma && beta && mean && rstd && scale && shift, "bn_finalize_train: null pointer");
    SED_REQUIRE(rows > 0 && C > 0 && count > 0, "bn_finalize_train: bad sizes rows=%d C=%d", rows, C);
    bn_finalize_train_k<<<cdiv(C, 2), 1024, 0, as_stream(stream)>>>(part, rows, C, count, gamma, beta, rmean, rvar,
                                                                      momentum, eps, mean, rstd, scale, shift);
    SED_LAUNCH_CHECK("bn_finalize_train");
    return 0;
}

// sums[0..C) = sum x, sums[C..2C) = sum x^2 over all partial rows (fixed order, double accumulation)
__global__ __launch_bounds__(1024) void bn_stat_sums_k(const float* __restrict__ part, int rows, int C, float* __restrict__ sums) {
    __shared__ double s1[32][33], s2[32][33];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double a = 0.0, q = 0.0;
    if (c < C)
#pragma unroll 8                                  // independent loads in flight: the partials sit in another XCD's L2
        for (int r = sl; r < rows; r += 32) {
            a += (double)part[(size_t)r * 2 * C + c];
            q += (double)part[(size_t)r * 2 * C + C + c];
        }
    s1[sl][cl] = a;
    s2[sl][cl] = q;
    __syncthreads();
    if (sl == 0 && c < C) {
        double A = 0.0, Q = 0.0;
        for (int s = 0; s < 32; ++s) { A += s1[s][cl]; Q += s2[s][cl]; }
        sums[c] = (float)A;
        sums[C + c] = (float)Q;
    }
}

__global__ void bn_finalize_from_sums_k(const float* __restrict__ sums, int C, double count, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                        float momentum, float eps, float* mean_o, float* rstd_o, float* scale_o, float* shift_o) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double m = (double)sums[c] / count;
    double var = (double)sums[C + c] / count - m * m;
    if (var < 0.0) var = 0.0;
    float rstd = (float)(1.0 / sqrt(var + (double)eps));
    mean_o[c] = (float)m;
    rstd_o[c] = rstd;
    float sc = gamma[c] * rstd;
    scale_o[c] = sc;
    shift_o[c] = beta[c] - (float)m * sc;
    if (rmean) {
        double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

extern "C" int sed_bn_stat_sums(const float* part, int rows, int C, float* sums, void* stream) {
    SED_REQUIRE(part && sums && rows > 0 && C > 0, "bn_stat_sums: bad arguments");
    bn_stat_sums_k<<<cdiv(C, 32), 1024, 0, as_stream(stream)>>>(part, rows, C, sums);
    SED_LAUNCH_CHECK("bn_stat_sums");
    return 0;
}

extern "C" int sed_bn_finalize_from_sums(const float* sums, int C, double count, const float* gamma, const float* beta,
                                         float* rmean, float* rvar, float momentum, float eps, float* mean, float* rstd,
                                         float* scale, float* shift, void* stream) {
    SED_REQUIRE(sums && gamma && beta && mean && rstd && scale && shift && C > 0 && count > 0, "bn_finalize_from_sums: bad arguments");
    bn_finalize_from_sums_k<<<cdiv(C, 128), 128, 0, as_stream(stream)>>>(sums, C, count, gamma, beta, rmean, rvar, momentum, eps,
                                                                        mean, rstd, scale, shift);
    SED_LAUNCH_CHECK("bn_finalize_from_sums");
    return 0;
}

__global__ void bn_finalize_eval_k(const float* g, const float* b, const float* rm, const float* rv, float eps,
                                   int C, float* scale, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sc = g[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = b[c] - rm[c] * sc;
}

extern "C" int sed_bn_finalize_eval(const float* gamma, const float* beta, const float* rm, const float* rv,
                                    float eps, int C, float* scale, float* shift, void* stream) {
    SED_REQUIRE(gamma && beta && rm && rv && scale && shift && C > 0, "bn_finalize_eval: bad arguments");
    bn_finalize_eval_k<<<cdiv(C, 256), 256, 0, as_stream(stream)>>>(gamma, beta, rm, rv, eps, C, scale, shift);
    SED_LAUNCH_CHECK("bn_finalize_eval");
    return 0;
}

// out[c] = sum_r part[r*row_stride + c]
// 256-thread workgroups (8 channels x 32 row slices) here and in the other finalisation kernels below: they run on the
// auxiliary stream beside a weight-gradient kernel that leaves every CU only ~80 VGPRs per SIMD lane and ~45 KB of LDS,
// where a 1024-thread workgroup cannot be placed at all and would wait for the whole MFMA kernel to drain.
__global__ __launch_bounds__(256) void reduce_rows_k(const float* __restrict__ part, int rows, int C,
                                                     int row_stride, float* __restrict__ out) {
    __shared__ double s1[32][9];
    const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    double a = 0.0;
    if (c < C) {
        double a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int r = sl;
        for (; r + 96 < rows; r += 128) {
            const float* p0 = part + (size_t)r * row_stride + c;
            a += (double)p0[0];
            a1 += (double)p0[(size_t)32 * row_stride];
            a2 += (double)p0[(size_t)64 * row_stride];
            a3 += (double)p0[(size_t)96 * row_stride];
        }
        for (; r < rows; r += 32) a += (double)part[(size_t)r * row_stride + c];
        a += a1 + a2 + a3;
    }
    s1[sl][cl] = a;
    __syncthreads();
    if (sl == 0 && c < C) {
        double A = 0.0;
        for (int s = 0; s < 32; ++s) A += s1[s][cl];
        out[c] = (float)A;
    }
}

extern "C" int sed_reduce_rows(const float* part, int rows, int C, int row_stride, float* out, void* stream) {
    SED_REQUIRE(part && out && rows > 0 && C > 0 && row_stride >= C, "reduce_rows: bad arguments");
    reduce_rows_k<<<cdiv(C, 8), 256, 0, as_stream(stream)>>>(part, rows, C, row_stride, out);
    SED_LAUNCH_CHECK("reduce_rows");
    return 0;
}

// ───────────────────────── forward: normalise + ReLU + pool + dropout ─────────────────────────
// channels-last output: one thread per output float4.
__global__ __launch_bounds__(256) void bn_relu_pool_drop_fwd_k(
    const float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ out, int B, int T, int F, int C, int pf, int pt, float drop_p, uint64_t seed,
    const uint64_t* __restrict__ seed_dev) {
    if (seed_dev) seed += seed_dev[0] * 0x9E3779B97F4A7C15ull;   // per-step salt kept on the device (graph replay)
    const int Tp = T / pt, Fp = F / pf, C4 = C >> 2;
    const size_t n = (size_t)B * Tp * Fp * C4;
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int c4 = (int)(i % C4);
        size_t pos = i / C4;
        int fp = (int)(pos % Fp);
        size_t bt = pos / Fp;                       // b*Tp + tp
        int tp = (int)(bt % Tp);
        size_t b = bt / Tp;
        f32x4 sc = *(const f32x4*)(scale + c4 * 4), sh = *(const f32x4*)(shift + c4 * 4);
        f32x4 m = {0, 0, 0, 0};                     // relu(max) == max(0, ...)
        for (int dt = 0; dt < pt; ++dt)
            for (int df = 0; df < pf; ++df) {
                f32x4 v = *(const f32x4*)(y + (((b * T + tp * pt + dt) * F + fp * pf + df) * (size_t)C) + c4 * 4);
                v = v * sc + sh;
#pragma unroll
                for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
            }
        if (drop_p > 0.f) {
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] *= sed_drop_mult(seed, i * 4 + k, drop_p, inv_keep);
        }
        *(f32x4*)(out + i * 4) = m;
    }
}

// GRU-order output [B][Tp][C][Fp]: one block per (b,tp) row group, LDS transpose.
__global__ __launch_bounds__(256) void bn_relu_pool_drop_fwd_tcf_k(
    const float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ out, int B, int T, int F, int C, int pf, int pt, float drop_p, uint64_t seed,
    const uint64_t* __restrict__ seed_dev) {
    if (seed_dev) seed += seed_dev[0] * 0x9E3779B97F4A7C15ull;   // per-step salt kept on the device (graph replay)
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [Fp][C+1]
    const int Tp = T / pt, Fp = F / pf, C4 = C >> 2, LD = C + 1;
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (int bt = blockIdx.x; bt < B * Tp; bt += gridDim.x) {
        int tp = bt % Tp;
        size_t b = bt / Tp;
        __syncthreads();
        for (int i = threadIdx.x; i < Fp * C4; i += 256) {
            int c4 = i % C4, fp = i / C4;
            f32x4 sc = *(const f32x4*)(scale + c4 * 4), sh = *(const f32x4*)(shift + c4 * 4);
            f32x4 m = {0, 0, 0, 0};
            for (int dt = 0; dt < pt; ++dt)
                for (int df = 0; df < pf; ++df) {
                    f32x4 v = *(const f32x4*)(y + (((b * T + tp * pt + dt) * F + fp * pf + df) * (size_t)C) + c4 * 4);
                    v = v * sc + sh;
#pragma unroll
                    for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
                }
            if (drop_p > 0.f) {
                size_t oi = (((size_t)bt * Fp + fp) * C4 + c4) * 4;      // logical channels-last index
#pragma unroll
                for (int k = 0; k < 4; ++k) m[k] *= sed_drop_mult(seed, oi + k, drop_p, inv_keep);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) tile[fp * LD + c4 * 4 + k] = m[k];
        }
        __syncthreads();
        float* o = out + (size_t)bt * C * Fp;
        for (int i = threadIdx.x; i < C * Fp; i += 256) {
            int c = i / Fp, fp = i - c * Fp;
            o[i] = tile[fp * LD + c];
        }
    }
}

static int check_pool(const char* who, int B, int T, int F, int C, int pf, int pt) {
    SED_REQUIRE(B > 0 && T > 0 && F > 0 && C > 0 && pf > 0 && pt > 0, "%s: bad shape", who);
    SED_REQUIRE(C % 4 == 0, "%s: C must be a multiple of 4 (got %d)", who, C);
    // like nn.MaxPool2d (ceil_mode=False; sed.py:90) a ragged tail is dropped: T' = floor(T/pt), F' = floor(F/pf); the tail
    // still takes part in the batch statistics and, in the backward, receives their gradient terms
    SED_REQUIRE(T / pt >= 1 && F / pf >= 1, "%s: T=%d / F=%d smaller than the pool (%d,%d)", who, T, F, pt, pf);
    return 0;
}

extern "C" int sed_bn_relu_pool_drop_fwd(const float* y, const float* scale, const float* shift, float* out, int B,
                                         int T, int F, int C, int pf, int pt, int out_tcf, float drop_p,
                                         uint64_t seed, const uint64_t* seed_dev, void* stream) {
    SED_REQUIRE(y && scale && shift && out, "bn_relu_pool_drop_fwd: null pointer");
    SED_TRY(check_pool("bn_relu_pool_drop_fwd", B, T, F, C, pf, pt));
    SED_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "bn_relu_pool_drop_fwd: drop_p=%f out of [0,1)", drop_p);
    hipStream_t s = as_stream(stream);
    const int Tp = T / pt, Fp = F / pf;
    SedProfScope prof(SED_K_BN_FWD, s, 4.0 * B * C * ((double)T * F + (double)Tp * Fp));
    if (!out_tcf) {
        size_t n = (size_t)B * Tp * Fp * (C / 4);
        int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
        bn_relu_pool_drop_fwd_k<<<grid, 256, 0, s>>>(y, scale, shift, out, B, T, F, C, pf, pt, drop_p, seed, seed_dev);
    } else {
        size_t lds = (size_t)Fp * (C + 1) * sizeof(float);
        SED_REQUIRE(lds <= 150 * 1024, "bn_relu_pool_drop_fwd: F'*C tile too large for LDS");
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)bn_relu_pool_drop_fwd_tcf_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int grid = B * Tp < 4096 ? B * Tp : 4096;
        bn_relu_pool_drop_fwd_tcf_k<<<grid, 256, lds, s>>>(y, scale, shift, out, B, T, F, C, pf, pt, drop_p, seed, seed_dev);
    }
    SED_LAUNCH_CHECK("bn_relu_pool_drop_fwd");
    return 0;
}

// ── routing decisions of the block (inspection / parity tests): which element of every pooling window the gradient takes ──
// The same expressions as the forward and backward kernels (z = y*scale + shift, first maximum wins, gate = max > 0), so the
// codes are exactly the decisions those kernels make.  route [B][Tp][Fp][C]: 0 = ReLU gate closed, 1 + w = window element
// w = df*pt + dt.  Independent of dropout (the keep-mask multiplies afterwards).
__global__ __launch_bounds__(256) void bn_relu_pool_route_k(
    const float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
    unsigned char* __restrict__ route, int B, int T, int F, int C, int pf, int pt) {
    const int Tp = T / pt, Fp = F / pf, C4 = C >> 2;
    const size_t n = (size_t)B * Tp * Fp * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int c4 = (int)(i % C4);
        size_t pos = i / C4;
        int fp = (int)(pos % Fp);
        size_t bt = pos / Fp;
        int tp = (int)(bt % Tp);
        size_t b = bt / Tp;
        f32x4 sc = *(const f32x4*)(scale + c4 * 4), sh = *(const f32x4*)(shift + c4 * 4);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bidx[4] = {0, 0, 0, 0};
        int widx = 0;
        for (int df = 0; df < pf; ++df)
            for (int dt = 0; dt < pt; ++dt, ++widx) {
                f32x4 v = *(const f32x4*)(y + (((b * T + tp * pt + dt) * F + fp * pf + df) * (size_t)C) + c4 * 4);
                f32x4 z = v * sc + sh;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (z[k] > best[k]) { best[k] = z[k]; bidx[k] = widx; }
            }
        uchar4 code;
        code.x = best[0] > 0.f ? (unsigned char)(1 + bidx[0]) : 0;
        code.y = best[1] > 0.f ? (unsigned char)(1 + bidx[1]) : 0;
        code.z = best[2] > 0.f ? (unsigned char)(1 + bidx[2]) : 0;
        code.w = best[3] > 0.f ? (unsigned char)(1 + bidx[3]) : 0;
        *(uchar4*)(route + i * 4) = code;
    }
}

extern "C" int sed_bn_relu_pool_route(const float* y, const float* scale, const float* shift, unsigned char* route,
                                      int B, int T, int F, int C, int pf, int pt, void* stream) {
    SED_REQUIRE(y && scale && shift && route, "bn_relu_pool_route: null pointer");
    SED_TRY(check_pool("bn_relu_pool_route", B, T, F, C, pf, pt));
    SED_REQUIRE(pf * pt <= 254, "bn_relu_pool_route: a %dx%d window does not fit the one-byte code", pf, pt);
    size_t n = (size_t)B * (T / pt) * (F / pf) * (C / 4);
    int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    bn_relu_pool_route_k<<<grid, 256, 0, as_stream(stream)>>>(y, scale, shift, route, B, T, F, C, pf, pt);
    SED_LAUNCH_CHECK("bn_relu_pool_route");
    return 0;
}

// ───────────────────────── backward ─────────────────────────
extern "C" int sed_bn_bwd_rows(int B, int T, int pool_t) {
    long r = (long)B * (T / (pool_t > 0 ? pool_t : 1));
    return (int)(r < BN_MAX_BLOCKS ? r : BN_MAX_BLOCKS);
}

// MODE 0: reduce (partials [grid][2][C] of sum g, sum g*xhat)
// MODE 1: apply  (dy, and dbias partials [grid][C])
// P12: the pool is (1,2) (the time-pooled topology of the reference): the window loops unroll, both conv outputs of a
// window are loaded together and kept for the apply loop instead of being read twice.
template <int MODE, bool P12>
__global__ __launch_bounds__(256) void bn_relu_pool_drop_bwd_k(
    const float* __restrict__ y, const float* __restrict__ dout, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ sum_g, const float* __restrict__ sum_gx, float* __restrict__ dy,
    float* __restrict__ partials, int B, int T, int F, int C, int pf, int pt, int out_tcf, float drop_p,
    uint64_t seed, const uint64_t* __restrict__ seed_dev) {
    if (seed_dev) seed += seed_dev[0] * 0x9E3779B97F4A7C15ull;   // per-step salt kept on the device (graph replay)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Tp = T / pt, Fp = F / pf, C4 = C >> 2, LD = C + 1;
    float* tile = smem;                                   // [Fp][C+1] (tcf only)
    const int tid = threadIdx.x;
    const int nslots = 256 / C4 > 0 ? 256 / C4 : 1;       // C4 <= 256 checked by the host
    const int c4 = tid % C4, slot = tid / C4;
    const bool active = slot < nslots;
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const float invN = 1.f / ((float)B * (float)T * (float)F);

    f32x4 sc = {0, 0, 0, 0}, sh = sc, mu = sc, rs = sc, sg = sc, sgx = sc;
    if (active) {
        sc = *(const f32x4*)(scale + c4 * 4);
        sh = *(const f32x4*)(shift + c4 * 4);
        mu = *(const f32x4*)(mean + c4 * 4);
        rs = *(const f32x4*)(rstd + c4 * 4);
        if (MODE == 1) {
            sg = *(const f32x4*)(sum_g + c4 * 4) * invN;
            sgx = *(const f32x4*)(sum_gx + c4 * 4) * invN;
        }
    }
    f32x4 a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};

    for (int bt = blockIdx.x; bt < B * Tp; bt += gridDim.x) {
        int tp = bt % Tp;
        size_t b = bt / Tp;
        if (out_tcf) {
            __syncthreads();
            const float* d = dout + (size_t)bt * C * Fp;
            for (int i = tid; i < C * Fp; i += 256) {
                int c = i / Fp, fp = i - c * Fp;
                tile[fp * LD + c] = d[i];
            }
            __syncthreads();
        }
        if (!active) continue;
        for (int fp = slot; fp < Fp; fp += nslots) {
            f32x4 g;
            if (out_tcf) {
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = tile[fp * LD + c4 * 4 + k];
            } else {
                g = *(const f32x4*)(dout + (((size_t)bt * Fp + fp) * C4 + c4) * 4);
            }
            if (drop_p > 0.f) {
                size_t oi = (((size_t)bt * Fp + fp) * C4 + c4) * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] *= sed_drop_mult(seed, oi + k, drop_p, inv_keep);
            }
            // locate the first maximum of the window (max_pool2d keeps the first index on ties)
            f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            f32x4 bx = {0, 0, 0, 0};
            int bidx[4] = {0, 0, 0, 0};
            int widx = 0;
            f32x4 vw[2];
            if (P12) {
                const float* yp = y + (((b * T + tp * 2) * F + fp) * (size_t)C) + c4 * 4;
                vw[0] = *(const f32x4*)yp;
                vw[1] = *(const f32x4*)(yp + (size_t)F * C);
            }
#pragma unroll
            for (int df = 0; df < (P12 ? 1 : pf); ++df)   // window order = (f, t): row-major over (H=F, W=T)
#pragma unroll
                for (int dt = 0; dt < (P12 ? 2 : pt); ++dt, ++widx) {
                    f32x4 v = P12 ? vw[dt] : *(const f32x4*)(y + (((b * T + tp * pt + dt) * F + fp * pf + df) * (size_t)C) + c4 * 4);
                    f32x4 z = v * sc + sh;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (z[k] > best[k]) { best[k] = z[k]; bx[k] = (v[k] - mu[k]) * rs[k]; bidx[k] = widx; }
                }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!(best[k] > 0.f)) g[k] = 0.f;          // ReLU gate
            if (MODE == 0) {
                a1 += g;
                a2 += g * bx;
            } else {
                widx = 0;
#pragma unroll
                for (int df = 0; df < (P12 ? 1 : pf); ++df)
#pragma unroll
                    for (int dt = 0; dt < (P12 ? 2 : pt); ++dt, ++widx) {
                        size_t off = (((b * T + tp * pt + dt) * F + fp * pf + df) * (size_t)C) + c4 * 4;
                        f32x4 v = P12 ? vw[dt] : *(const f32x4*)(y + off);
                        f32x4 o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float xh = (v[k] - mu[k]) * rs[k];
                            float gk = (bidx[k] == widx) ? g[k] : 0.f;
                            o[k] = sc[k] * (gk - sg[k] - xh * sgx[k]);
                        }
                        *(f32x4*)(dy + off) = o;
                        a1 += o;
                    }
            }
        }
        if (MODE == 1) {
            // positions no pooling window covers (floor pooling): no gradient arrives from above, the statistics terms remain
            auto tail = [&](int t, int f) {
                const size_t off = (((b * T + t) * F + f) * (size_t)C) + c4 * 4;
                const f32x4 v = *(const f32x4*)(y + off);
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = sc[k] * (-sg[k] - (v[k] - mu[k]) * rs[k] * sgx[k]);
                *(f32x4*)(dy + off) = o;
                a1 += o;
            };
            const int ftail = F - Fp * pf, ttail = T - Tp * pt;
            for (int i = slot; i < pt * ftail; i += nslots) tail(tp * pt + i / ftail, Fp * pf + i % ftail);
            if (tp == Tp - 1)
                for (int i = slot; i < ttail * F; i += nslots) tail(Tp * pt + i / F, i % F);
        }
    }
    // block reduction of the per-thread partial sums, fixed slot order
    __syncthreads();
    float* red = smem;                                    // [nslots][2][C]
    if (active) {
        *(f32x4*)(red + (slot * 2 + 0) * C + c4 * 4) = a1;
        if (MODE == 0) *(f32x4*)(red + (slot * 2 + 1) * C + c4 * 4) = a2;
    }
    __syncthreads();
    const int nout = (MODE == 0 ? 2 : 1) * C;
    for (int i = tid; i < nout; i += 256) {
        int which = i / C, c = i - which * C;
        float a = 0.f;
        for (int s = 0; s < nslots; ++s) a += red[(s * 2 + which) * C + c];
        partials[(size_t)blockIdx.x * nout + i] = a;
    }
}

static size_t bwd_lds(int F, int C, int pf, int out_tcf) {
    size_t red = (size_t)2 * 256 * 4 * sizeof(float) + 2 * C * sizeof(float);
    size_t t = out_tcf ? (size_t)(F / pf) * (C + 1) * sizeof(float) : 0;
    return red > t ? red : t;
}

extern "C" int sed_bn_relu_pool_drop_bwd_reduce(const float* y, const float* dout, const float* scale,
                                                const float* shift, const float* mean, const float* rstd,
                                                float* partials, int B, int T, int F, int C, int pf, int pt,
                                                int out_tcf, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                                                void* stream) {
    SED_REQUIRE(y && dout && scale && shift && mean && rstd && partials, "bn_bwd_reduce: null pointer");
    SED_TRY(check_pool("bn_bwd_reduce", B, T, F, C, pf, pt));
    SED_REQUIRE(C / 4 <= 256, "bn_bwd_reduce: C=%d too large (max 1024)", C);
    size_t lds = bwd_lds(F, C, pf, out_tcf);
    SED_REQUIRE(lds <= 150 * 1024, "bn_bwd_reduce: tile too large for LDS");
    const bool p12 = (pf == 1 && pt == 2);
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void*)bn_relu_pool_drop_bwd_k<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)bn_relu_pool_drop_bwd_k<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    int grid = sed_bn_bwd_rows(B, T, pt);
    SedProfScope prof(SED_K_BN_BWD_REDUCE, as_stream(stream), 4.0 * B * C * ((double)T * F + (double)(T / pt) * (F / pf)));
    if (p12)
        bn_relu_pool_drop_bwd_k<0, true><<<grid, 256, lds, as_stream(stream)>>>(y, dout, scale, shift, mean, rstd, nullptr, nullptr,
                                                                                 nullptr, partials, B, T, F, C, pf, pt, out_tcf, drop_p, seed, seed_dev);
    else
        bn_relu_pool_drop_bwd_k<0, false><<<grid, 256, lds, as_stream(stream)>>>(y, dout, scale, shift, mean, rstd, nullptr, nullptr,
                                                                                  nullptr, partials, B, T, F, C, pf, pt, out_tcf, drop_p, seed, seed_dev);
    SED_LAUNCH_CHECK("bn_bwd_reduce");
    return 0;
}

// ── reduction pass of the block that feeds the GRU, from its own pooled output ──
// The same identities as the BNR epilogue of the conv data gradient (conv.hip): the pooled output q = relu(max z) mask/(1-p)
// is > 0 exactly where the gradient passes, g = dout/(1-p) there, and the BatchNorm output at the arg-max is z = q (1-p), so
// xhat = (z - beta)/gamma.  The pass reads the pooled tensor and its gradient ([B][Tp][C][Fp], the GRU layout) instead of
// the conv output (pt*pf times larger) and needs neither the arg-max search nor the dropout hash.  gamma == 0 (z constant):
// beta <= 0 -> nothing passes; beta > 0 -> every window is a tie whose arg-max is its first element, xhat from the conv output.
// Thread t owns the float4s i = t + 256 k of a row, which lie in channel i / (Fp/4) for every row: BP_K accumulator pairs per
// thread, summed per channel in fixed order at the end (deterministic).
#define BP_K 8           // max float4 per thread and row; the kernel is instantiated for the exact count (registers: it runs beside a GEMM)
template <int BPK, int NT>
__global__ __launch_bounds__(NT) void bn_bwd_reduce_pooled_tcf_k(
    const float* __restrict__ pooled, const float* __restrict__ dout, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ y, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ partials, int B, int T, int F, int C, int pf, int pt, float drop_p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [2][C * Fp / 4]
    const int Tp = T / pt, Fp = F / pf, F4 = Fp >> 2, n4 = C * F4, tid = threadIdx.x;
    const float keep = 1.f - drop_p, inv_keep = 1.f / keep;
    float a1[BPK], a2[BPK], kr[BPK], nb[BPK];
    unsigned slow = 0;                                    // bit k: |gamma| < |beta| / 64 (incl. gamma == 0 with beta > 0): xhat from the conv output
#pragma unroll
    for (int k = 0; k < BPK; ++k) {
        a1[k] = a2[k] = kr[k] = nb[k] = 0.f;
        const int i = tid + NT * k;
        if (i < n4) {
            const int c = i / F4;
            const float gm = gamma[c], bt = beta[c];
            const bool sl = gm == 0.f ? bt > 0.f : fabsf(gm) * 64.f < fabsf(bt);       // see the BNR epilogue in conv.hip
            const float rg = (gm != 0.f && !sl) ? 1.0f / gm : 0.f;
            kr[k] = keep * rg;
            nb[k] = -bt * rg;
            if (sl) slow |= 1u << k;
        }
    }
    const long rows = (long)B * Tp;
    constexpr int CH = BPK < 8 ? BPK : 8;                 // float4 pairs in flight per thread (registers: this pass runs beside a GEMM)
#pragma unroll 1
    for (long r = blockIdx.x; r < rows; r += gridDim.x) {
        const f32x4* q4 = (const f32x4*)(pooled + (size_t)r * C * Fp);
        const f32x4* d4 = (const f32x4*)(dout + (size_t)r * C * Fp);
#pragma unroll
        for (int k0 = 0; k0 < BPK; k0 += CH) {
            f32x4 qv[CH], dv[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int i = tid + NT * (k0 + k);
                if (k0 + k < BPK && i < n4) { qv[k] = q4[i]; dv[k] = d4[i]; }
            }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int i = tid + NT * (k0 + k);
                if (k0 + k < BPK && i < n4) {
                    f32x4 g0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) g0[e] = qv[k][e] > 0.f ? dv[k][e] : 0.f;
                    const f32x4 gx = g0 * (qv[k] * kr[k0 + k] + nb[k0 + k]);          // gamma == 0 channels: kr = nb = 0, added below
                    a1[k0 + k] += (g0[0] + g0[1]) + (g0[2] + g0[3]);
                    a2[k0 + k] += (gx[0] + gx[1]) + (gx[2] + gx[3]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (slow) {                                           // rare: xhat of the window's arg-max (searched like the forward does) from the conv output
#pragma unroll 1
        for (int k = 0; k < BPK; ++k) {
            if (!((slow >> k) & 1u)) continue;
            const int i = tid + NT * k, c = i / F4, fp0 = (i - c * F4) * 4;
            const float mu = mean[c], rs = rstd[c], sc = scale[c], sh = shift[c];
            float s2 = 0.f;
            for (long r = blockIdx.x; r < rows; r += gridDim.x) {
                const f32x4 qv = ((const f32x4*)(pooled + (size_t)r * C * Fp))[i], dv = ((const f32x4*)(dout + (size_t)r * C * Fp))[i];
                const long b = r / Tp, tp = r - b * Tp;
                for (int e = 0; e < 4; ++e) {
                    if (!(qv[e] > 0.f)) continue;
                    float best = -INFINITY, ybest = 0.f;
                    for (int df = 0; df < pf; ++df)
                        for (int dt = 0; dt < pt; ++dt) {
                            const float yv = y[(((size_t)b * T + (size_t)tp * pt + dt) * F + (size_t)(fp0 + e) * pf + df) * C + c];
                            const float zz = __builtin_fmaf(yv, sc, sh);
                            if (zz > best) { best = zz; ybest = yv; }
                        }
                    s2 += dv[e] * ((ybest - mu) * rs);
                }
            }
#pragma unroll
            for (int kk = 0; kk < BPK; ++kk) if (kk == k) a2[kk] += s2;
        }
    }
#pragma unroll
    for (int k = 0; k < BPK; ++k) {
        const int i = tid + NT * k;
        if (i < n4) { smem[i] = a1[k] * inv_keep; smem[n4 + i] = a2[k] * inv_keep; }
    }
    __syncthreads();
    for (int j = tid; j < 2 * C; j += NT) {
        const int which = j / C, c = j - which * C;
        float a = 0.f;
        for (int f = 0; f < F4; ++f) a += smem[which * n4 + c * F4 + f];
        partials[(size_t)blockIdx.x * 2 * C + j] = a;
    }
}

extern "C" int sed_bn_bwd_reduce_pooled_supported(int F, int C, int pool_f, int pool_t, int out_tcf) {
    if (!out_tcf || F <= 0 || C <= 0 || pool_f <= 0 || pool_t <= 0 || F % pool_f != 0) return 0;
    const int Fp = F / pool_f;
    return (Fp % 4 == 0 && (long)C * (Fp / 4) <= 1024L * 4) ? 1 : 0;       // 256 threads x <= 8 float4, or 1024 x <= 4
}

extern "C" int sed_bn_bwd_reduce_pooled(const float* pooled, const float* dout, const float* gamma, const float* beta,
                                        const float* y, const float* mean, const float* rstd, const float* scale, const float* shift,
                                        float* partials, int B, int T, int F, int C, int pf, int pt, int out_tcf, float drop_p, void* stream) {
    SED_REQUIRE(pooled && dout && gamma && beta && y && mean && rstd && scale && shift && partials, "bn_bwd_reduce_pooled: null pointer");
    SED_TRY(check_pool("bn_bwd_reduce_pooled", B, T, F, C, pf, pt));
    SED_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "bn_bwd_reduce_pooled: bad drop_p");
    SED_REQUIRE(sed_bn_bwd_reduce_pooled_supported(F, C, pf, pt, out_tcf),
                "bn_bwd_reduce_pooled: needs the GRU layout, F/pool_f a multiple of 4 and C*F/pool_f <= 16384 (use sed_bn_relu_pool_drop_bwd_reduce)");
    const int grid = sed_bn_bwd_rows(B, T, pt);
    const size_t lds = (size_t)2 * C * (F / pf / 4) * sizeof(float);
    SedProfScope prof(SED_K_BN_BWD_REDUCE, as_stream(stream), 8.0 * B * C * (double)(T / pt) * (F / pf));
    const long n4 = (long)C * (F / pf / 4);
    const int nt = n4 <= 256L * BP_K ? 256 : 1024, need = cdiv(n4, nt);
#define BP_LAUNCH(K_, NT_) bn_bwd_reduce_pooled_tcf_k<K_, NT_><<<grid, NT_, lds, as_stream(stream)>>>(pooled, dout, gamma, beta, y, mean, rstd, scale, shift, partials, B, T, F, C, pf, pt, drop_p)
    if (nt == 256) {
        switch (need) {
            case 1: BP_LAUNCH(1, 256); break;   case 2: BP_LAUNCH(2, 256); break;   case 3: BP_LAUNCH(3, 256); break;
            case 4: BP_LAUNCH(4, 256); break;   case 5: BP_LAUNCH(5, 256); break;   case 6: BP_LAUNCH(6, 256); break;
            case 7: BP_LAUNCH(7, 256); break;   default: BP_LAUNCH(8, 256); break;
        }
    } else {
        switch (need) {                     // 9..16 float4 per thread at 256 threads: 3..4 at 1024
            case 3: BP_LAUNCH(3, 1024); break;   default: BP_LAUNCH(4, 1024); break;
        }
    }
#undef BP_LAUNCH
    SED_LAUNCH_CHECK("bn_bwd_reduce_pooled");
    return 0;
}

// The channels the fused reductions cannot serve (see ConvBnRed in conv.hip): |gamma| < |beta| / 64, incl. gamma == 0 with
// beta > 0.  Their sum g*xhat is recomputed HERE, by the two-channel workgroup that finalises them, from the tensors the exact
// xhat lives in: g = dpooled / (1-p) where the block's pooled output is > 0 (kept AND gate open), the window of the conv output
// searched exactly as the forward and the apply pass search it (z = fma(y, scale, shift), first maximum), xhat = (y_max - mean)
// rstd.  A cold path (such channels are rare) inside a launch that exists anyway; fixed order (thread slices, lane tree, wave
// order): deterministic.
struct BnSmallGamma {
    const float* dpooled;    // [B][Tp][Fp][C] gradient of the block's pooled output (channels-last)
    const float* pooled;     // [B][Tp][Fp][C] the pooled output itself
    const float* y;          // [B][T][F][C] conv output
    const float* gamma; const float* beta; const float* mean; const float* rstd; const float* scale; const float* shift;
    int B, T, F, pf, pt; float inv_keep;
};
__global__ __launch_bounds__(1024) void bn_bwd_finalize_k(const float* __restrict__ part, int rows, int C, float* sum_g,
                                                          float* sum_gx, float* dgamma, float* dbeta, BnSmallGamma sg) {
    __shared__ double sred[16][2][2];
    const int c = blockIdx.x * 2 + (threadIdx.x & 1);
    double A, Q;
    bn_two_channel_sums(part, rows, C, c, c < C, sred, &A, &Q);      // same layout [rows][2][C] as the forward statistics
    if (sg.y) {
        bool small = false;
        if (c < C) {
            const float gm = sg.gamma[c], bt = sg.beta[c];
            small = gm == 0.f ? bt > 0.f : fabsf(gm) * 64.f < fabsf(bt);
        }
        if (__syncthreads_or(small)) {                     // workgroup-uniform: almost never taken
            __shared__ double sq[16][2];
            const int Tp = sg.T / sg.pt, Fp = sg.F / sg.pf;
            const long npos = (long)sg.B * Tp * Fp;
            double q = 0.0;
            if (small) {
                const float mu = sg.mean[c], rs = sg.rstd[c], sc = sg.scale[c], sh = sg.shift[c];
                for (long pos = threadIdx.x >> 1; pos < npos; pos += 512) {
                    if (!(sg.pooled[pos * C + c] > 0.f)) continue;
                    const long bt_ = pos / Fp, fp = pos - bt_ * Fp, b = bt_ / Tp, tp = bt_ - b * Tp;
                    float best = -INFINITY, ybest = 0.f;
                    for (int df = 0; df < sg.pf; ++df)
                        for (int dt = 0; dt < sg.pt; ++dt) {
                            const float yv = sg.y[((b * sg.T + tp * sg.pt + dt) * sg.F + fp * sg.pf + df) * C + c];
                            const float zz = __builtin_fmaf(yv, sc, sh);
                            if (zz > best) { best = zz; ybest = yv; }
                        }
                    q += (double)(sg.dpooled[pos * C + c] * sg.inv_keep) * (double)((ybest - mu) * rs);
                }
            }
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
            for (int o = 2; o < 64; o <<= 1) q += __shfl_xor(q, o, 64);
            if (lane < 2) sq[wv][lane] = q;
            __syncthreads();
            if (threadIdx.x < 2 && small) {
                double t = 0.0;
                for (int w = 0; w < 16; ++w) t += sq[w][threadIdx.x];
                Q = t;
            }
        }
    }
    if (threadIdx.x < 2 && c < C) {
        sum_g[c] = (float)A;
        sum_gx[c] = (float)Q;
        if (dbeta) dbeta[c] = (float)A;
        if (dgamma) dgamma[c] = (float)Q;
    }
}

extern "C" int sed_bn_bwd_finalize(const float* partials, int rows, int C, float* sum_g, float* sum_gx,
                                   float* dgamma, float* dbeta, void* stream) {
    SED_REQUIRE(partials && sum_g && sum_gx && rows > 0 && C > 0, "bn_bwd_finalize: bad arguments");
    bn_bwd_finalize_k<<<cdiv(C, 2), 1024, 0, as_stream(stream)>>>(partials, rows, C, sum_g, sum_gx, dgamma, dbeta, BnSmallGamma{});
    SED_LAUNCH_CHECK("bn_bwd_finalize");
    return 0;
}

extern "C" int sed_bn_bwd_finalize_small_gamma(const float* partials, int rows, int C, float* sum_g, float* sum_gx,
                                               float* dgamma, float* dbeta, const float* dpooled, const float* pooled,
                                               const float* y, const float* gamma, const float* beta, const float* mean,
                                               const float* rstd, const float* scale, const float* shift,
                                               int B, int T, int F, int pf, int pt, float drop_p, void* stream) {
    SED_REQUIRE(partials && sum_g && sum_gx && rows > 0 && C > 0, "bn_bwd_finalize_small_gamma: bad arguments");
    SED_REQUIRE(dpooled && pooled && y && gamma && beta && mean && rstd && scale && shift, "bn_bwd_finalize_small_gamma: null pointer");
    SED_TRY(check_pool("bn_bwd_finalize_small_gamma", B, T, F, C, pf, pt));
    SED_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "bn_bwd_finalize_small_gamma: drop_p=%f out of [0,1)", drop_p);
    BnSmallGamma sg{dpooled, pooled, y, gamma, beta, mean, rstd, scale, shift, B, T, F, pf, pt, 1.f / (1.f - drop_p)};
    bn_bwd_finalize_k<<<cdiv(C, 2), 1024, 0, as_stream(stream)>>>(partials, rows, C, sum_g, sum_gx, dgamma, dbeta, sg);
    SED_LAUNCH_CHECK("bn_bwd_finalize_small_gamma");
    return 0;
}

extern "C" int sed_bn_relu_pool_drop_bwd_apply(const float* y, const float* dout, const float* scale,
                                               const float* shift, const float* mean, const float* rstd,
                                               const float* sum_g, const float* sum_gx, float* dy,
                                               float* dbias_partials, int B, int T, int F, int C, int pf, int pt,
                                               int out_tcf, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                                               void* stream) {
    SED_REQUIRE(y && dout && scale && shift && mean && rstd && sum_g && sum_gx && dy && dbias_partials,
                "bn_bwd_apply: null pointer");
    SED_TRY(check_pool("bn_bwd_apply", B, T, F, C, pf, pt));
    SED_REQUIRE(C / 4 <= 256, "bn_bwd_apply: C=%d too large (max 1024)", C);
    size_t lds = bwd_lds(F, C, pf, out_tcf);
    SED_REQUIRE(lds <= 150 * 1024, "bn_bwd_apply: tile too large for LDS");
    const bool p12 = (pf == 1 && pt == 2);
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void*)bn_relu_pool_drop_bwd_k<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)bn_relu_pool_drop_bwd_k<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    int grid = sed_bn_bwd_rows(B, T, pt);
    SedProfScope prof(SED_K_BN_BWD_APPLY, as_stream(stream), 4.0 * B * C * (2.0 * T * F + (double)(T / pt) * (F / pf)));
    if (p12)
        bn_relu_pool_drop_bwd_k<1, true><<<grid, 256, lds, as_stream(stream)>>>(y, dout, scale, shift, mean, rstd, sum_g, sum_gx, dy,
                                                                                 dbias_partials, B, T, F, C, pf, pt, out_tcf, drop_p, seed, seed_dev);
    else
        bn_relu_pool_drop_bwd_k<1, false><<<grid, 256, lds, as_stream(stream)>>>(y, dout, scale, shift, mean, rstd, sum_g, sum_gx, dy,
                                                                                  dbias_partials, B, T, F, C, pf, pt, out_tcf, drop_p, seed, seed_dev);
    SED_LAUNCH_CHECK("bn_bwd_apply");
    return 0;
}
