// data.hip — GPU-resident minibatch assembly: the work HitWindowDataset.__getitem__ does per
// sample in Python workers (reference sed.py:64-79; decorte_datamodule.py:39-49,77-111), utils.split_in_seqs /
// split_multi_channels (utils.py:15-41) and StandardScaler.fit (feature.py:127-128), as HBM-bound kernels over a
// fold that stays resident in device memory.
#include "common.h"

// One workgroup per (sample, chunk of JC feature columns): gather L frames x JC of the C*F mel bins, transpose through
// LDS to the network input layout x[b][c][f][t] (time contiguous), zero the SpecAugment masks; the first chunk also
// max-pools the labels.  Any window size (config 5: 512 frames x 4 x 128 bins = 1 MB per sample) goes through 64 KB tiles.
__global__ __launch_bounds__(256) void window_batch_k(
    const float* __restrict__ mel, const float* __restrict__ lab, long N, int C, int F, int K,
    const int* __restrict__ starts, const int* __restrict__ tmask, const int* __restrict__ fmask, int M, int tw, int fw,
    float* __restrict__ x, float* __restrict__ y, int L, int pool, int JC) {
    extern __shared__ __attribute__((aligned(16))) float tile[];     // [L][JC + 1]
    const int b = blockIdx.x, j0 = blockIdx.y * JC, tid = threadIdx.x;
    const int CF = C * F, LD = JC + 1;
    const int jc = (CF - j0 < JC) ? CF - j0 : JC;
    long s0 = starts[b];
    if (s0 < 0) s0 = 0;
    if (s0 + L > N) s0 = N - L;                                        // the reference's end-of-fold fallback
    for (int i = tid; i < L * jc; i += 256) {
        int t = i / jc, j = i - t * jc;
        tile[t * LD + j] = mel[(s0 + t) * CF + j0 + j];
    }
    __syncthreads();
    float* xo = x + ((size_t)b * CF + j0) * L;
    for (int i = tid; i < jc * L; i += 256) {
        int j = i / L, t = i - j * L;                                  // j0 + j = c*F + f
        int f = (j0 + j) % F;
        float v = tile[t * LD + j];
        for (int m = 0; m < M; ++m) {
            int t0 = tmask ? tmask[b * M + m] : -1, f0 = fmask ? fmask[b * M + m] : -1;
            if ((t0 >= 0 && t >= t0 && t < t0 + tw) || (f0 >= 0 && f >= f0 && f < f0 + fw)) v = 0.f;
        }
        xo[i] = v;
    }
    if (blockIdx.y != 0) return;
    const int Lo = L / pool;
    for (int i = tid; i < Lo * K; i += 256) {
        int to = i / K, k = i - to * K;
        float m = -INFINITY;
        for (int j = 0; j < pool; ++j) m = fmaxf(m, lab[(s0 + (long)to * pool + j) * K + k]);
        y[(size_t)b * Lo * K + i] = m;
    }
}

extern "C" int sed_window_batch(const float* mel, const float* lab, long N, int C, int F, int K, const int* starts,
                                const int* tmask, const int* fmask, int n_masks, int time_w, int freq_w, float* x,
                                float* y, int B, int L, int pool, void* stream) {
    SED_REQUIRE(mel && lab && starts && x && y, "window_batch: null pointer");
    SED_REQUIRE(N >= L && C > 0 && F > 0 && K > 0 && B > 0 && L > 0 && pool > 0 && L % pool == 0, "window_batch: bad sizes");
    SED_REQUIRE(n_masks >= 0 && (n_masks == 0 || (tmask && fmask)), "window_batch: masks requested but not given");
    const int CF = C * F;
    int JC = (int)((size_t)(64 * 1024) / ((size_t)L * sizeof(float))) - 1;       // columns per tile: L*(JC+1) floats <= 64 KB
    if (JC > CF) JC = CF;
    if (JC < 1) JC = 1;
    const size_t lds = (size_t)L * (JC + 1) * sizeof(float);
    SED_REQUIRE(lds <= 150 * 1024, "window_batch: a window of %d frames is too long for one LDS tile", L);
    const int nch = (CF + JC - 1) / JC;
    SED_REQUIRE(nch <= 65535, "window_batch: too many column chunks");
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)window_batch_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    window_batch_k<<<dim3((unsigned)B, (unsigned)nch), 256, lds, as_stream(stream)>>>(
        mel, lab, N, C, F, K, starts, n_masks ? tmask : nullptr, n_masks ? fmask : nullptr, n_masks, time_w, freq_w, x, y, L, pool, JC);
    SED_LAUNCH_CHECK("window_batch");
    return 0;
}

// feat [N][C*F] -> out [N/S][C][F][S] (time_last=1: the network input layout) or [N/S][C][S][F] (utils.py layout).
// One workgroup per (sequence, channel, chunk of FC mel bins): an [S][FC] tile goes through LDS so that both the read
// (along the mel axis) and the write (along the time axis for time_last) are contiguous; any C, F and S up to 19 000 frames.
__global__ __launch_bounds__(256) void pack_sequences_k(const float* __restrict__ feat, int C, int F, int S, int FC,
                                                        int time_last, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float tile[];     // [S][FC + 1]
    const int n = blockIdx.x, c = blockIdx.y, f0 = blockIdx.z * FC, tid = threadIdx.x, CF = C * F, LD = FC + 1;
    const int fc = (F - f0 < FC) ? F - f0 : FC;
    for (int i = tid; i < S * fc; i += 256) {
        int t = i / fc, j = i - t * fc;
        tile[t * LD + j] = feat[((size_t)n * S + t) * CF + c * F + f0 + j];
    }
    __syncthreads();
    float* o = out + ((size_t)n * C + c) * F * S;                     // [F][S] or [S][F] of this (sequence, channel)
    for (int i = tid; i < S * fc; i += 256) {
        if (time_last) { int j = i / S, t = i - j * S; o[(size_t)(f0 + j) * S + t] = tile[t * LD + j]; }
        else { int t = i / fc, j = i - t * fc; o[(size_t)t * F + f0 + j] = tile[t * LD + j]; }
    }
}

extern "C" int sed_pack_sequences(const float* feat, long N, int C, int F, int S, int time_last, float* out, void* stream) {
    SED_REQUIRE(feat && out && N >= S && C > 0 && F > 0 && S > 0, "pack_sequences: bad arguments");
    SED_REQUIRE(C <= 65535 && N / S <= 0x7fffffffL, "pack_sequences: C=%d / %ld sequences exceed the grid", C, N / S);
    int FC = (int)((size_t)(64 * 1024) / ((size_t)S * sizeof(float))) - 1;      // mel bins per tile: S*(FC+1) floats <= 64 KB
    if (FC > F) FC = F;
    if (FC < 1) FC = 1;
    const size_t lds = (size_t)S * (FC + 1) * sizeof(float);
    SED_REQUIRE(lds <= 150 * 1024, "pack_sequences: seq_len=%d too long for one LDS tile", S);
    const int nfc = (F + FC - 1) / FC;
    SED_REQUIRE(nfc <= 65535, "pack_sequences: too many mel chunks");
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)pack_sequences_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    pack_sequences_k<<<dim3((unsigned)(N / S), (unsigned)C, (unsigned)nfc), 256, lds, as_stream(stream)>>>(feat, C, F, S, FC, time_last, out);
    SED_LAUNCH_CHECK("pack_sequences");
    return 0;
}

// per-column mean and population standard deviation with sklearn StandardScaler's rules (feature.py:127-128;
// sklearn/preprocessing/_data.py, utils/extmath.py:_incremental_mean_and_var): float64 accumulation, variance from the
// CENTRED second pass  var = (sum d^2 - (sum d)^2 / N) / N  with d = x - mean, and a column whose variance is within the
// rounding bound of that algorithm,  var <= N*eps*var + (N*mean*eps)^2  (_is_constant_feature), gets scale 1.
#define CS_BLOCKS 256
template <bool CENTRED>
__global__ __launch_bounds__(256) void colstats_partial_k(const float* __restrict__ x, long N, int F, int ld,
                                                          const double* __restrict__ mean, double* __restrict__ part) {
    // F <= 256 columns of a matrix with row stride ld; block handles rows [r0, r1); thread (col, slice)
    const int nsl = 256 / F > 0 ? 256 / F : 1;
    const int col = threadIdx.x % F, sl = threadIdx.x / F;
    long per = (N + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = r0 + per < N ? r0 + per : N;
    double a = 0.0, q = 0.0;
    const double mu = CENTRED ? mean[col] : 0.0;
    if (sl < nsl)
        for (long r = r0 + sl; r < r1; r += nsl) { double v = (double)x[r * ld + col] - mu; a += v; q += v * v; }
    __shared__ double s1[256], s2[256];
    s1[threadIdx.x] = a; s2[threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < F) {
        double A = 0.0, Q = 0.0;
        for (int s = 0; s < nsl; ++s) { A += s1[s * F + threadIdx.x]; Q += s2[s * F + threadIdx.x]; }
        part[((size_t)blockIdx.x * 2) * F + threadIdx.x] = A;
        part[((size_t)blockIdx.x * 2 + 1) * F + threadIdx.x] = Q;
    }
}
__global__ void colstats_mean_k(const double* __restrict__ part, int nb, long N, int F, double* __restrict__ mean) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= F) return;
    double A = 0.0;
    for (int b = 0; b < nb; ++b) A += part[((size_t)b * 2) * F + c];
    mean[c] = A / (double)N;
}
__global__ void colstats_final_k(const double* __restrict__ part, int nb, long N, int F, const double* __restrict__ mean_d,
                                 double* mean, double* stdv) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= F) return;
    double D = 0.0, Q = 0.0;
    for (int b = 0; b < nb; ++b) { D += part[((size_t)b * 2) * F + c]; Q += part[((size_t)b * 2 + 1) * F + c]; }
    const double n = (double)N, m = mean_d[c];
    double var = (Q - D * D / n) / n;
    if (var < 0.0) var = 0.0;
    const double eps = 2.220446049250313e-16, nme = n * m * eps;
    const bool constant = var <= n * eps * var + nme * nme;
    mean[c] = m;
    stdv[c] = constant ? 1.0 : sqrt(var);
}

extern "C" size_t sed_col_mean_std_workspace_bytes(int F) { return ((size_t)CS_BLOCKS * 2 + 1) * F * sizeof(double); }

extern "C" int sed_col_mean_std(const float* x, long N, int F, double* mean, double* stdv, void* workspace, void* stream) {
    SED_REQUIRE(x && mean && stdv && workspace && N > 0 && F > 0, "col_mean_std: bad arguments");
    hipStream_t s = as_stream(stream);
    int nb = N < CS_BLOCKS ? (int)N : CS_BLOCKS;
    // wide matrices (multichannel folds: 4 x 128 = 512 columns) go through in chunks of 256 columns, each with its own
    // slice of the workspace
    for (int c0 = 0; c0 < F; c0 += 256) {
        const int Fc = F - c0 < 256 ? F - c0 : 256;
        double* part = (double*)workspace + (size_t)c0 * (CS_BLOCKS * 2 + 1);
        double* mean_d = part + (size_t)CS_BLOCKS * 2 * Fc;
        colstats_partial_k<false><<<nb, 256, 0, s>>>(x + c0, N, Fc, F, nullptr, part);
        SED_LAUNCH_CHECK("colstats_partial");
        colstats_mean_k<<<cdiv(Fc, 64), 64, 0, s>>>(part, nb, N, Fc, mean_d);
        SED_LAUNCH_CHECK("colstats_mean");
        colstats_partial_k<true><<<nb, 256, 0, s>>>(x + c0, N, Fc, F, mean_d, part);
        SED_LAUNCH_CHECK("colstats_partial_centred");
        colstats_final_k<<<cdiv(Fc, 64), 64, 0, s>>>(part, nb, N, Fc, mean_d, mean + c0, stdv + c0);
        SED_LAUNCH_CHECK("colstats_final");
    }
    return 0;
}

// StandardScaler.transform (feature.py:128-129) on a float32 matrix: sklearn's in-place `X -= mean_; X /= scale_` with
// float64 statistics rounds to float32 after EACH of the two steps; so does this.  In place allowed.
__global__ __launch_bounds__(256) void col_standardize_k(const float* __restrict__ x, long n, int F, const double* __restrict__ mean,
                                                         const double* __restrict__ stdv, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int c = (int)(i % F);
        float d = (float)((double)x[i] - mean[c]);
        out[i] = (float)((double)d / stdv[c]);
    }
}
extern "C" int sed_col_standardize(const float* x, long N, int F, const double* mean, const double* stdv, float* out, void* stream) {
    SED_REQUIRE(x && mean && stdv && out && N > 0 && F > 0, "col_standardize: bad arguments");
    long n = N * F;
    long nb = (n + 255) / 256;
    col_standardize_k<<<(unsigned)(nb < 4096 ? nb : 4096), 256, 0, as_stream(stream)>>>(x, n, F, mean, stdv, out);
    SED_LAUNCH_CHECK("col_standardize");
    return 0;
}

// ───────────────────────── segment-based metric counts on the device (metrics.py:20-68) ─────────────────────────
// One thread per 1-second block of `block` consecutive rows of the concatenated [rows][K] prediction / label
// matrices (blocks straddle window boundaries exactly like the reference).  Integer counts, so the result is exact and
// order-independent:  out[0..5]  frame-wise  TP, Nref, Nsys, S, D, I
//                     out[6..8]  blocks incl. the partial last one (F1: ceil)   TP, Nref, Nsys
//                     out[9..12] full blocks only (ER: floor)                    S, D, I, Nref
__global__ void segment_counts_k(const float* __restrict__ pred, const float* __restrict__ lab, long rows, int K, int block,
                                 float thr, unsigned long long* __restrict__ out) {
    long bi = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long nceil = (rows + block - 1) / block, nfloor = rows / block;
    unsigned long long c[SED_SEGMENT_COUNTS] = {};
    if (bi < nceil) {
        long r0 = bi * block, r1 = r0 + block < rows ? r0 + block : rows;
        unsigned ob = 0, tb = 0;                                   // per-class block maxima as bit masks (K <= 32)
        for (long r = r0; r < r1; ++r) {
            unsigned fp = 0, fn = 0;
            for (int k = 0; k < K; ++k) {
                const float lv = lab[r * K + k];
                bool o = pred[r * K + k] > thr, t = lv == 1.f;
                ob |= (unsigned)o << k; tb |= (unsigned)t << k;
                c[0] += (o && t); c[1] += t; c[2] += o;
                fp += (o && !t); fn += (t && !o);
                const bool t0 = lv >= 0.f && lv < 1.f, t1 = lv >= 1.f && lv < 2.f;    // uint8 truncation of the label
                c[13] += (!o && t0); c[14] += (o && t0); c[15] += (!o && t1); c[16] += (o && t1);
            }
            c[3] += fp < fn ? fp : fn; c[4] += fn > fp ? fn - fp : 0; c[5] += fp > fn ? fp - fn : 0;
        }
        unsigned tp = __popc(ob & tb), nref = __popc(tb), nsys = __popc(ob), fp = __popc(ob & ~tb), fn = __popc(tb & ~ob);
        c[6] += tp; c[7] += nref; c[8] += nsys;
        if (bi < nfloor) { c[9] += fp < fn ? fp : fn; c[10] += fn > fp ? fn - fp : 0; c[11] += fp > fn ? fp - fn : 0; c[12] += nref; }
    }
#pragma unroll
    for (int i = 0; i < SED_SEGMENT_COUNTS; ++i) {
        unsigned long long v = c[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(out + i, v);
    }
}

extern "C" int sed_segment_counts(const float* pred, const float* lab, long rows, int K, int block, float threshold,
                                  unsigned long long* counts17, void* stream) {
    SED_REQUIRE(pred && lab && counts17 && rows > 0 && K > 0 && K <= 32 && block > 0, "segment_counts: bad arguments (K <= 32)");
    hipStream_t s = as_stream(stream);
    hipError_t e = hipMemsetAsync(counts17, 0, SED_SEGMENT_COUNTS * sizeof(unsigned long long), s);
    if (e != hipSuccess) { sed_set_error("segment_counts: memset: %s", hipGetErrorString(e)); return (int)e; }
    long nceil = (rows + block - 1) / block;
    segment_counts_k<<<cdiv(nceil, 256), 256, 0, s>>>(pred, lab, rows, K, block, threshold, counts17);
    SED_LAUNCH_CHECK("segment_counts");
    return 0;
}
