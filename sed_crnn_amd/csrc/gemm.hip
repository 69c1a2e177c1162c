// gemm.hip — dense fp32 GEMM on v_mfma_f32_32x32x2_f32 + the small time-distributed dense head.
//
// Replaces aten::mm / addmm under nn.GRU's input projections and nn.Linear (reference
// sed.py:101-103,111-112; crnn_lightning.py:61-64,71-73) and their autograd transposes.
// The GRU input projection (M=B*T', K=C*F, N=3H) is the one genuine dense GEMM of the path, hence MFMA.
// fp32 in / fp32 accumulate: bit-for-bit a k-ordered fmaf chain, no reduced precision.
#include "common.h"

#define GM_BK 16
#define GM_LDK 20   // k-contiguous LDS row: 16 + 4 pad floats -> conflict-free ds_read_b128

// A_KC / B_KC: operand is contiguous along k (true) or along m/n (false).
template <int WM, int WN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_k(
    const float* __restrict__ A, long a_si, long a_sk, const float* __restrict__ Bm, long b_sk, long b_sj,
    float* __restrict__ C, long ldc, const float* __restrict__ bias, float beta, int M, int N, int K,
    int a_vec, int b_vec) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int A_FLOATS = A_KC ? BM * GM_LDK : GM_BK * BM;
    constexpr int B_FLOATS = B_KC ? BN * GM_LDK : GM_BK * BN;
    __shared__ __attribute__((aligned(16))) float As[A_FLOATS];
    __shared__ __attribute__((aligned(16))) float Bs[B_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int wm0 = (wave >> 1) * 32 * WM, wn0 = (wave & 1) * 32 * WN;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int t = 0; t < WM; ++t)
#pragma unroll
        for (int u = 0; u < WN; ++u)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][u][j] = 0.f;

    f32x4 ra[WM], rb[WN];
    auto load_a = [&](int k0) {
#pragma unroll
        for (int u = 0; u < WM; ++u) {
            int i = tid + u * 256;
            f32x4 v = {0, 0, 0, 0};
            if (A_KC) {
                int row = i >> 2, kq = i & 3;
                int m = m0 + row, k = k0 + kq * 4;
                if (m < M) {
                    const float* p = A + (long)m * a_si + k;
                    if (a_vec && k + 3 < K) v = *(const f32x4*)p;
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (k + e < K) v[e] = p[e];
                }
            } else {
                int kk = i / (BM / 4), mq = i - kk * (BM / 4);
                int m = m0 + mq * 4, k = k0 + kk;
                if (k < K) {
                    const float* p = A + (long)k * a_sk + m;
                    if (a_vec && m + 3 < M) v = *(const f32x4*)p;
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (m + e < M) v[e] = p[e];
                }
            }
            ra[u] = v;
        }
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int u = 0; u < WN; ++u) {
            int i = tid + u * 256;
            f32x4 v = {0, 0, 0, 0};
            if (B_KC) {
                int row = i >> 2, kq = i & 3;
                int n = n0 + row, k = k0 + kq * 4;
                if (n < N) {
                    const float* p = Bm + (long)n * b_sj + k;
                    if (b_vec && k + 3 < K) v = *(const f32x4*)p;
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (k + e < K) v[e] = p[e];
                }
            } else {
                int kk = i / (BN / 4), nq = i - kk * (BN / 4);
                int n = n0 + nq * 4, k = k0 + kk;
                if (k < K) {
                    const float* p = Bm + (long)k * b_sk + n;
                    if (b_vec && n + 3 < N) v = *(const f32x4*)p;
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < N) v[e] = p[e];
                }
            }
            rb[u] = v;
        }
    };
    auto store_ab = [&]() {
#pragma unroll
        for (int u = 0; u < WM; ++u) {
            int i = tid + u * 256;
            if (A_KC) { int row = i >> 2, kq = i & 3; *(f32x4*)(As + row * GM_LDK + kq * 4) = ra[u]; }
            else { int kk = i / (BM / 4), mq = i - kk * (BM / 4); *(f32x4*)(As + kk * BM + mq * 4) = ra[u]; }
        }
#pragma unroll
        for (int u = 0; u < WN; ++u) {
            int i = tid + u * 256;
            if (B_KC) { int row = i >> 2, kq = i & 3; *(f32x4*)(Bs + row * GM_LDK + kq * 4) = rb[u]; }
            else { int kk = i / (BN / 4), nq = i - kk * (BN / 4); *(f32x4*)(Bs + kk * BN + nq * 4) = rb[u]; }
        }
    };

    load_a(0);
    load_b(0);
    for (int k0 = 0; k0 < K; k0 += GM_BK) {
        __syncthreads();
        store_ab();
        __syncthreads();
        if (k0 + GM_BK < K) { load_a(k0 + GM_BK); load_b(k0 + GM_BK); }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4 af[WM], bf[WN];
#pragma unroll
            for (int t = 0; t < WM; ++t) {
                if (A_KC) af[t] = *(const f32x4*)(As + (wm0 + t * 32 + r) * GM_LDK + g * 8 + 4 * h);
                else
#pragma unroll
                    for (int j = 0; j < 4; ++j) af[t][j] = As[(g * 8 + 4 * h + j) * BM + wm0 + t * 32 + r];
            }
#pragma unroll
            for (int u = 0; u < WN; ++u) {
                if (B_KC) bf[u] = *(const f32x4*)(Bs + (wn0 + u * 32 + r) * GM_LDK + g * 8 + 4 * h);
                else
#pragma unroll
                    for (int j = 0; j < 4; ++j) bf[u][j] = Bs[(g * 8 + 4 * h + j) * BN + wn0 + u * 32 + r];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < WM; ++t)
#pragma unroll
                    for (int u = 0; u < WN; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t][j], bf[u][j], acc[t][u], 0, 0, 0);
        }
    }

#pragma unroll
    for (int t = 0; t < WM; ++t)
#pragma unroll
        for (int u = 0; u < WN; ++u) {
            int col = n0 + wn0 + u * 32 + r;
            if (col >= N) continue;
            float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                int row = m0 + wm0 + t * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
                if (row < M) {
                    float* cp = C + (long)row * ldc + col;
                    float v = acc[t][u][j] + bv;
                    if (beta != 0.f) v += beta * (*cp);
                    *cp = v;
                }
            }
        }
}

template <int WM, int WN>
static void launch_gemm(bool akc, bool bkc, dim3 grid, hipStream_t s, const float* A, long a_si, long a_sk,
                        const float* B, long b_sk, long b_sj, float* C, long ldc, const float* bias, float beta,
                        int M, int N, int K, int av, int bv) {
    if (akc && bkc) gemm_f32_k<WM, WN, true, true><<<grid, 256, 0, s>>>(A, a_si, a_sk, B, b_sk, b_sj, C, ldc, bias, beta, M, N, K, av, bv);
    else if (akc && !bkc) gemm_f32_k<WM, WN, true, false><<<grid, 256, 0, s>>>(A, a_si, a_sk, B, b_sk, b_sj, C, ldc, bias, beta, M, N, K, av, bv);
    else if (!akc && bkc) gemm_f32_k<WM, WN, false, true><<<grid, 256, 0, s>>>(A, a_si, a_sk, B, b_sk, b_sj, C, ldc, bias, beta, M, N, K, av, bv);
    else gemm_f32_k<WM, WN, false, false><<<grid, 256, 0, s>>>(A, a_si, a_sk, B, b_sk, b_sj, C, ldc, bias, beta, M, N, K, av, bv);
}

extern "C" int sed_gemm_f32(const float* A, long a_si, long a_sk, const float* B, long b_sk, long b_sj, float* C,
                            long ldc, const float* bias, float beta, int M, int N, int K, void* stream) {
    SED_REQUIRE(A && B && C, "gemm_f32: null pointer");
    SED_REQUIRE(M > 0 && N > 0 && K > 0 && ldc >= N, "gemm_f32: bad sizes M=%d N=%d K=%d ldc=%ld", M, N, K, ldc);
    SED_REQUIRE(a_si == 1 || a_sk == 1, "gemm_f32: A must be contiguous along i or k (strides %ld,%ld)", a_si, a_sk);
    SED_REQUIRE(b_sk == 1 || b_sj == 1, "gemm_f32: B must be contiguous along k or j (strides %ld,%ld)", b_sk, b_sj);
    bool akc = (a_sk == 1), bkc = (b_sk == 1);
    // when both strides are 1 (a vector) either reading is valid; prefer k-contiguous
    int av = (((uintptr_t)A & 15) == 0) && ((akc ? a_si : a_sk) % 4 == 0);
    int bv = (((uintptr_t)B & 15) == 0) && ((bkc ? b_sj : b_sk) % 4 == 0);
    hipStream_t s = as_stream(stream);
    SedProfScope prof(SED_K_GEMM, s, 2.0 * M * (double)N * K);
    long blocks22 = (long)cdiv(M, 128) * cdiv(N, 128);
    if (M > 64 && N > 64 && blocks22 >= 192) {
        launch_gemm<2, 2>(akc, bkc, dim3(cdiv(N, 128), cdiv(M, 128)), s, A, a_si, a_sk, B, b_sk, b_sj, C, ldc, bias, beta, M, N, K, av, bv);
    } else {
        launch_gemm<1, 1>(akc, bkc, dim3(cdiv(N, 64), cdiv(M, 64)), s, A, a_si, a_sk, B, b_sk, b_sj, C, ldc, bias, beta, M, N, K, av, bv);
    }
    SED_LAUNCH_CHECK("gemm_f32");
    return 0;
}

// ───────────────────────── small dense head ─────────────────────────
// y[m][n] = act(b[n] + sum_k x[m][k] W[n][k]); one wave per row m, lanes split k.
__global__ __launch_bounds__(256) void linear_fwd_k(const float* __restrict__ x, const float* __restrict__ W,
                                                    const float* __restrict__ b, float* __restrict__ y, int M,
                                                    int K, int N, int relu) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    for (int n = 0; n < N; ++n) {
        float a = 0.f;
        for (int k = lane; k < K; k += 64) a += x[(size_t)m * K + k] * W[(size_t)n * K + k];
        a = wave_sum(a);
        if (lane == 0) {
            a += b ? b[n] : 0.f;
            if (relu) a = fmaxf(a, 0.f);
            y[(size_t)m * N + n] = a;
        }
    }
}

extern "C" int sed_linear_fwd(const float* x, const float* W, const float* b, float* y, int M, int K, int N,
                              int relu, void* stream) {
    SED_REQUIRE(x && W && y && M > 0 && K > 0 && N > 0, "linear_fwd: bad arguments");
    linear_fwd_k<<<cdiv(M, 4), 256, 0, as_stream(stream)>>>(x, W, b, y, M, K, N, relu);
    SED_LAUNCH_CHECK("linear_fwd");
    return 0;
}

#define LIN_CHUNK 64   // rows per partial block in the weight gradient

__global__ void linear_relu_mask_k(const float* __restrict__ y, float* __restrict__ dy, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !(y[i] > 0.f)) dy[i] = 0.f;
}
// dx[m][k] = sum_n dy[m][n] W[n][k]
__global__ void linear_dx_k(const float* __restrict__ dy, const float* __restrict__ W, float* __restrict__ dx,
                            int M, int K, int N) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * K) return;
    int k = (int)(i % K);
    long m = i / K;
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += dy[m * N + n] * W[(size_t)n * K + k];
    dx[i] = a;
}
// partial[chunk][n*K + k] = sum_{m in chunk} dy[m][n] x[m][k];  partial_b[chunk][n] = sum dy[m][n]
__global__ void linear_dw_partial_k(const float* __restrict__ dy, const float* __restrict__ x,
                                    float* __restrict__ part, int M, int K, int N) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int tot = N * K + N;
    if (i >= tot) return;
    int m_lo = blockIdx.y * LIN_CHUNK, m_hi = m_lo + LIN_CHUNK < M ? m_lo + LIN_CHUNK : M;
    float a = 0.f;
    if (i < N * K) {
        int n = i / K, k = i - n * K;
        for (int m = m_lo; m < m_hi; ++m) a += dy[(size_t)m * N + n] * x[(size_t)m * K + k];
    } else {
        int n = i - N * K;
        for (int m = m_lo; m < m_hi; ++m) a += dy[(size_t)m * N + n];
    }
    part[(size_t)blockIdx.y * tot + i] = a;
}
__global__ void linear_dw_reduce_k(const float* __restrict__ part, int chunks, int tot, int NK,
                                   float* __restrict__ dW, float* __restrict__ db) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tot) return;
    double a = 0.0;
    for (int c = 0; c < chunks; ++c) a += (double)part[(size_t)c * tot + i];
    if (i < NK) dW[i] = (float)a;
    else if (db) db[i - NK] = (float)a;
}

extern "C" size_t sed_linear_bwd_workspace_bytes(int M, int K, int N) {
    return (size_t)cdiv(M, LIN_CHUNK) * ((size_t)N * K + N) * sizeof(float);
}

extern "C" int sed_linear_bwd(const float* x, const float* W, const float* y, float* dy, float* dx, float* dW,
                              float* db, void* workspace, int M, int K, int N, int relu, void* stream) {
    SED_REQUIRE(x && W && dy && dW && workspace && M > 0 && K > 0 && N > 0, "linear_bwd: bad arguments");
    SED_REQUIRE(!relu || y, "linear_bwd: relu=1 needs the forward output y");
    hipStream_t s = as_stream(stream);
    if (relu) {
        long n = (long)M * N;
        linear_relu_mask_k<<<cdiv(n, 256), 256, 0, s>>>(y, dy, n);
        SED_LAUNCH_CHECK("linear_relu_mask");
    }
    if (dx) {
        long n = (long)M * K;
        linear_dx_k<<<cdiv(n, 256), 256, 0, s>>>(dy, W, dx, M, K, N);
        SED_LAUNCH_CHECK("linear_dx");
    }
    int tot = N * K + N, chunks = cdiv(M, LIN_CHUNK);
    linear_dw_partial_k<<<dim3(cdiv(tot, 256), chunks), 256, 0, s>>>(dy, x, (float*)workspace, M, K, N);
    SED_LAUNCH_CHECK("linear_dw_partial");
    linear_dw_reduce_k<<<cdiv(tot, 256), 256, 0, s>>>((const float*)workspace, chunks, tot, N * K, dW, db);
    SED_LAUNCH_CHECK("linear_dw_reduce");
    return 0;
}
