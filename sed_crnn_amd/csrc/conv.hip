// conv.hip — 3x3 / stride 1 / zero-pad 1 convolution over (mel, time), channels-last.
//
// Replaces aten::mkldnn_convolution / convolution_backward reached from nn.Conv2d at
// reference sed.py:88,107 and crnn_lightning.py:47.
//
// Two code paths (gfx950 only):
//   * small  : Cin*Cout small (first layer, Cin in {1,2,4}; the 16-channel Lightning net).
//              Direct VALU convolution, HBM-bound: coalesced 16 B/lane channel-last stores,
//              mel-band halo tile in LDS; weights in registers (Cin <= 4) or LDS.
//   * mfma   : Cin%32==0, Cout%32==0 (the 128-channel layers, K = 9*Cin = 1152): implicit GEMM
//              on v_mfma_f32_32x32x2_f32 (exact fp32), (TT+2)x(FT+2) halo tile of 32 input
//              channels in LDS read with conflict-free ds_read_b128 (TT time rows x FT mel columns per
//              block, any mel width), weights streamed L2 -> registers in MFMA fragment order.
// Both write per-block BatchNorm partial sums (sum, sum of squares) from the epilogue so the
// conv output is not re-read for the statistics.
#include "common.h"
#include "conv_shared.h"
#include <type_traits>

#define CV_CIC 32   // input channels per LDS chunk
#define CV_LD 36    // padded LDS row (floats): 16-B slots 9*row -> conflict-free ds_read_b128
#define CV_MTW 5    // max 32-row tiles per wave
#define WG_FT 40     // wgrad: max mel columns per tile (2 time rows at that width; TT*FT <= 2*WG_FT)
#define WG_NX 6      // wgrad: DMA items (float4) per thread of the halo tile   (4*(WG_FT+2)*8 <= 256*WG_NX)
#define WG_ND 10     // wgrad: DMA items (float4) per thread of the dY tile     (2*WG_FT*32    <= 256*WG_ND)

#define CV_TPAD 56   // fwd: extra floats per halo time-row (bank-conflict-free mel wrap-around, see kernel)
#define CV_NH 8      // fwd (the fp32 kernel's vmcnt(12) = 4 + CV_NH is written out in its asm): max float4 per thread of the halo tile ((TT+2)*(FT+2)*8 <= 256*CV_NH)

struct ConvPlan {
    int kind;       // 0 small, 1 mfma, -1 unsupported
    int TT;         // time rows per block tile
    int FT, nft;    // mfma: mel columns per block tile, mel tiles (FT == F, nft == 1 on the small path)
    int nct;        // mfma: 32-wide co tiles per block
    int tblocks;    // ceil(T/TT)
    int rows;       // stat partial rows = B * tblocks * nft
    int nco, nci;   // small path: output channels per workgroup, input channels per launch
    size_t lds;
    double score;   // mfma: the tile score of conv_tile (compares the even-time-row tile of the pooling epilogue with the free choice)
};

// MFMA block tile: TT time rows x FT mel columns (+1 halo each side) with TT*FT <= limit output rows (every wave always
// runs its CV_MTW 32-row tiles, so rows below the limit are idle MFMA cycles) and at most 256*CV_NH/8 halo rows.
// Score = useful share of the MFMA rows (tile fill x mel coverage x time coverage), discounted by the halo share that is
// staged per tile; measured on MI355X at F=128: 32x5 125 TFLOP/s, 26x6 119, 32x4 102, 64x1 54.
// pooled: the inference kernel whose epilogue pools time pairs — an even number of time rows, at most 32 mel columns.
static bool conv_tile(int F, int T, int limit, int* TT_out, int* FT_out, int* nft_out, double* score_out = nullptr, bool pooled = false) {
    double best = -1.0;
    int last_ft = -1;
    for (int nft = 1; nft <= F; ++nft) {
        int FT = cdiv(F, nft);
        FT += FT & 1;                                    // even (the weight-gradient kernel walks mel pairs)
        if (FT == last_ft) continue;
        last_ft = FT;
        const int nf = cdiv(F, FT);
        // tall tiles (TT > 8) only for a full-width narrow mel axis (the mel-pooled topologies: F = 8, 4 after pooling),
        // and only while two blocks still fit the LDS of a CU
        const int tt_max = (nft == 1) ? 64 : 8;
        for (int TT = 1; TT <= tt_max && TT <= T; ++TT) {
            if (TT * FT > limit || (TT + 2) * (FT + 2) * 8 > 256 * CV_NH) continue;
            if (pooled && ((TT & 1) || FT > 32)) continue;
            if (TT > 8 && (size_t)2 * (TT + 2) * ((FT + 2) * CV_LD + CV_TPAD) * sizeof(float) > 80 * 1024) continue;
            double util = ((double)(TT * FT) / limit) * ((double)F / ((double)nf * FT)) * ((double)T / ((double)cdiv(T, TT) * TT));
            double score = util / (1.0 + 0.25 * ((double)(TT + 2) * (FT + 2) / (TT * FT) - 1.0));
            if (score > best) { best = score; *TT_out = TT; *FT_out = FT; *nft_out = nf; }
        }
        if (FT <= 2) break;
    }
    if (score_out) *score_out = best;
    return best > 0.0;
}

static ConvPlan conv_plan(int B, int Cin, int F, int T, int Cout, int x_is_nchw, bool pooled = false) {
    ConvPlan p{};
    p.kind = -1;
    p.FT = F; p.nft = 1;
    if (B <= 0 || Cin <= 0 || F <= 0 || T <= 0 || Cout <= 0) return p;
    if (!x_is_nchw && Cin % CV_CIC == 0 && Cout % 32 == 0) {
        int nct = (Cout % 128 == 0) ? 4 : (Cout % 64 == 0 ? 2 : 1);
        int mparts = 4 / nct;
        int limit = 32 * CV_MTW * mparts;
        if (!conv_tile(F, T, limit, &p.TT, &p.FT, &p.nft, &p.score, pooled)) return p;
        p.kind = 1; p.nct = nct;
        p.lds = (size_t)2 * (p.TT + 2) * ((p.FT + 2) * CV_LD + CV_TPAD) * sizeof(float);
        const size_t epi = (size_t)(4 * 1024 + 256) * sizeof(float);      // epilogue: four 32x32 transpose scratches + the stat exchange
        if (p.lds < epi) p.lds = epi;
        p.lds += (size_t)limit * sizeof(unsigned);                        // the epilogue's row-offset table (one entry per MFMA row)
    }
    if (p.kind < 0) {
        if (Cout % 4 != 0) return p;
        int TT = 4;
        if (TT > T) TT = T;
        // output channels per workgroup: all of them up to 64; input channels per launch: as many as fit 144 KB of LDS
        // beside the NCO weights per input channel
        p.nco = Cout < 64 ? Cout : 64;
        const size_t per_ci = ((size_t)9 * p.nco + (size_t)(TT + 2) * (F + 2)) * sizeof(float);
        size_t nci = (size_t)(144 * 1024) / per_ci;
        if (nci < 1) return p;                          // a mel axis of > 6 000 bins
        if (nci > (size_t)Cin) nci = Cin;
        p.nci = (int)nci;
        size_t lds = per_ci * nci;
        size_t red = (size_t)2 * 256 * 4 * sizeof(float);
        if (lds < red) lds = red;
        p.kind = 0; p.TT = TT; p.nct = 0; p.lds = lds;
    }
    p.tblocks = cdiv(T, p.TT);
    p.rows = B * p.tblocks * p.nft;
    return p;
}

// ───────────────────────── weight packing ─────────────────────────
// MFMA fragment order of a (Cin -> Cout) tap matrix: [tap][ci/32][ (ci%32)/8 ][co/32][lane = co%32 + 32*((ci%8)/4)][ci%4]
// i.e. exactly the B operand of v_mfma_f32_32x32x2_f32 for 4 consecutive k-steps, 1 KiB per wave-load.
__host__ __device__ inline size_t conv_frag_index(int tap, int co, int ci, int Cout, int Cin) {
    int cc = ci >> 5, g = (ci & 31) >> 3, h = (ci & 7) >> 2, j = ci & 3;
    int cot = co >> 5, r = co & 31;
    return ((((((size_t)tap * (Cin >> 5) + cc) * 4 + g) * (Cout >> 5) + cot) * 64) + r + 32 * h) * 4 + j;
}

__global__ void conv_pack_w_k(const float* __restrict__ w, float* __restrict__ wf,
                              float* __restrict__ wd, int Cout, int Cin) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int n = Cout * Cin * 9;
    if (i >= n) return;
    int tap = i % 9, ci = (i / 9) % Cin, co = i / (9 * Cin);
    float v = w[i];
    const bool frag = (Cin % 32 == 0) && (Cout % 32 == 0);
    if (wf) wf[frag ? conv_frag_index(tap, co, ci, Cout, Cin) : ((size_t)tap * Cout + co) * Cin + ci] = v;
    // dgrad: dx[pos][ci] = sum_tap' sum_co dy[pos + tap' - 1][co] * w[co][ci][flip(tap')]  (a Cout -> Cin conv)
    if (wd) wd[frag ? conv_frag_index(8 - tap, ci, co, Cin, Cout) : ((size_t)(8 - tap) * Cin + ci) * Cout + co] = v;
}

extern "C" int sed_conv3x3_pack_weights(const float* w, float* wf, float* wd, int Cout, int Cin, void* stream) {
    SED_REQUIRE(w && Cout > 0 && Cin > 0, "conv3x3_pack_weights: bad arguments");
    int n = Cout * Cin * 9;
    conv_pack_w_k<<<cdiv(n, 256), 256, 0, as_stream(stream)>>>(w, wf, wd, Cout, Cin);
    SED_LAUNCH_CHECK("conv_pack_w");
    return 0;
}

struct ConvPackMulti {
    const float* w[SED_MAX_CONV]; float* wf[SED_MAX_CONV]; float* wd[SED_MAX_CONV]; int Cout[SED_MAX_CONV], Cin[SED_MAX_CONV];
    // inference (optional, per layer): BatchNorm on running statistics as scale = gamma / sqrt(var + eps), shift = beta - mean scale,
    // written to scale_out / shift_out; fold != 0: the packed weights are w * scale[co] and bias_out = bias * scale + shift
    const float* gamma[SED_MAX_CONV]; const float* beta[SED_MAX_CONV]; const float* rm[SED_MAX_CONV]; const float* rv[SED_MAX_CONV];
    const float* bias[SED_MAX_CONV]; float* scale_out[SED_MAX_CONV]; float* shift_out[SED_MAX_CONV]; float* bias_out[SED_MAX_CONV];
    int fold[SED_MAX_CONV]; float eps;
    int wino_f[SED_MAX_CONV], wino_d[SED_MAX_CONV];      // wf / wd of this layer is the Winograd-transformed packing (wino.hip) instead
    // one more slice of the grid (blockIdx.y == n): a row-major [rows][C*Fp] matrix whose columns are re-ordered from the
    // reference's GRU feature order c*Fp + f (sed.py:108-110) to the channels-last order f*C + c of a pooled conv output
    const float* perm_src[2]; float* perm_dst; int perm_rows, perm_C, perm_Fp;      // two sources of perm_rows rows each -> [2 perm_rows][K]
};
__global__ void conv_pack_w_multi_k(ConvPackMulti a, int n) {
    const int l = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (l == n) {
        // one row per workgroup and round: read it as it lies (coalesced), write it re-ordered (coalesced) through LDS
        extern __shared__ __attribute__((aligned(16))) float prow[];
        const int K = a.perm_C * a.perm_Fp;
        for (int row = blockIdx.x; row < 2 * a.perm_rows; row += gridDim.x) {
            const float* src = a.perm_src[row >= a.perm_rows] + (size_t)(row >= a.perm_rows ? row - a.perm_rows : row) * K;
            __syncthreads();
            for (int k = threadIdx.x; k < K; k += blockDim.x) prow[k] = src[k];
            __syncthreads();
            float* dst = a.perm_dst + (size_t)row * K;
            for (int k = threadIdx.x; k < K; k += blockDim.x) {
                const int f = k / a.perm_C, c = k - f * a.perm_C;
                dst[k] = prow[c * a.perm_Fp + f];
            }
        }
        return;
    }
    const int Cout = a.Cout[l], Cin = a.Cin[l];
    if (a.wino_f[l] | a.wino_d[l])
        wino_pack_one(a.w[l], a.wino_f[l] ? a.wf[l] : nullptr, a.wino_d[l] ? a.wd[l] : nullptr, Cout, Cin, i,
                      a.fold[l] ? a.gamma[l] : nullptr, a.rv[l], a.eps);
    if (a.gamma[l] && i < Cout) {
        const float sc = a.gamma[l][i] / sqrtf(a.rv[l][i] + a.eps), sh = a.beta[l][i] - a.rm[l][i] * sc;      // = bn_finalize_eval_k
        if (a.scale_out[l]) { a.scale_out[l][i] = sc; a.shift_out[l][i] = sh; }
        if (a.fold[l]) a.bias_out[l][i] = (a.bias[l] ? a.bias[l][i] : 0.f) * sc + sh;
    }
    if (i >= Cout * Cin * 9) return;
    const int tap = i % 9, ci = (i / 9) % Cin, co = i / (9 * Cin);
    float v = a.w[l][i];
    if (a.fold[l]) v *= a.gamma[l][co] / sqrtf(a.rv[l][co] + a.eps);
    const bool frag = (Cin % 32 == 0) && (Cout % 32 == 0);
    if (a.wf[l] && !a.wino_f[l]) a.wf[l][frag ? conv_frag_index(tap, co, ci, Cout, Cin) : ((size_t)tap * Cout + co) * Cin + ci] = v;
    if (a.wd[l] && !a.wino_d[l]) a.wd[l][frag ? conv_frag_index(8 - tap, ci, co, Cin, Cout) : ((size_t)(8 - tap) * Cin + ci) * Cout + co] = v;
}
static int set_lds_early(size_t bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_pack_w_multi_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { sed_set_error("hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e)); return (int)e; }
    return 0;
}
static int pack_multi_launch(const ConvPackMulti& a, int n, void* stream) {
    int nmax = 0;
    for (int l = 0; l < n; ++l) {
        const int nl = a.Cout[l] * a.Cin[l] * 9;
        if (nl > nmax) nmax = nl;
    }
    const int gx = cdiv(nmax > 0 ? nmax : 1, 256);
    const size_t lds = a.perm_src[0] ? (size_t)a.perm_C * a.perm_Fp * sizeof(float) : 0;      // one row of the re-ordered matrix
    SED_REQUIRE(lds <= 64 * 1024, "conv_pack: a GRU input row of %zu bytes does not fit the re-ordering buffer", lds);
    if (lds > 48 * 1024) SED_TRY(set_lds_early(lds));
    conv_pack_w_multi_k<<<dim3(gx, n + (a.perm_src[0] ? 1 : 0)), 256, lds, as_stream(stream)>>>(a, n);
    SED_LAUNCH_CHECK("conv_pack_w_multi");
    return 0;
}
int sed_internal_conv_pack_multi(int n, const float* const* w, float* const* wf, float* const* wd, const int* Cout, const int* Cin,
                                 const int* wino_f, const int* wino_d, void* stream) {
    SED_REQUIRE(n > 0 && n <= SED_MAX_CONV && w && wf && wd && Cout && Cin, "conv_pack_multi: bad arguments");
    ConvPackMulti a{};
    for (int l = 0; l < n; ++l) {
        SED_REQUIRE(w[l] && Cout[l] > 0 && Cin[l] > 0, "conv_pack_multi: bad layer %d", l);
        a.w[l] = w[l]; a.wf[l] = wf[l]; a.wd[l] = wd[l]; a.Cout[l] = Cout[l]; a.Cin[l] = Cin[l];
        a.wino_f[l] = (wino_f && wino_f[l] && wf[l]) ? 1 : 0;
        a.wino_d[l] = (wino_d && wino_d[l] && wd[l]) ? 1 : 0;
        SED_REQUIRE(!(a.wino_f[l] | a.wino_d[l]) || (Cout[l] % 64 == 0 && Cin[l] % 64 == 0), "conv_pack_multi: layer %d cannot take the Winograd packing", l);
    }
    return pack_multi_launch(a, n, stream);
}
// The inference form: every layer's packing, the BatchNorm coefficients on running statistics (scale / shift, for the layers
// that still apply them in a pass of their own), the folded weights + bias of the layers whose conv epilogue pools (fold[l]),
// and the column re-ordering of the first GRU layer's input weights when the last block's output stays channels-last — ONE launch.
int sed_internal_conv_pack_eval(int n, const float* const* w, const float* const* bias, const float* const* gamma,
                                const float* const* beta, const float* const* rm, const float* const* rv, float eps,
                                float* const* wf, float* const* scale, float* const* shift, float* const* bias_folded, const int* fold,
                                const int* wino, const int* Cout, const int* Cin, const float* perm_src0, const float* perm_src1, float* perm_dst,
                                int perm_rows, int perm_C, int perm_Fp, void* stream) {
    SED_REQUIRE(n > 0 && n <= SED_MAX_CONV && w && gamma && beta && rm && rv && wf && scale && shift && fold && Cout && Cin, "conv_pack_eval: bad arguments");
    ConvPackMulti a{};
    for (int l = 0; l < n; ++l) {
        SED_REQUIRE(w[l] && gamma[l] && beta[l] && rm[l] && rv[l] && wf[l] && Cout[l] > 0 && Cin[l] > 0, "conv_pack_eval: bad layer %d", l);
        SED_REQUIRE(fold[l] ? (bias_folded && bias_folded[l]) : (scale[l] && shift[l]), "conv_pack_eval: layer %d lacks its outputs", l);
        a.w[l] = w[l]; a.wf[l] = wf[l]; a.Cout[l] = Cout[l]; a.Cin[l] = Cin[l];
        a.gamma[l] = gamma[l]; a.beta[l] = beta[l]; a.rm[l] = rm[l]; a.rv[l] = rv[l]; a.bias[l] = bias ? bias[l] : nullptr;
        a.scale_out[l] = scale[l]; a.shift_out[l] = shift[l]; a.bias_out[l] = bias_folded ? bias_folded[l] : nullptr; a.fold[l] = fold[l];
        a.wino_f[l] = (wino && wino[l]) ? 1 : 0;
        SED_REQUIRE(!a.wino_f[l] || (Cout[l] % 64 == 0 && Cin[l] % 64 == 0), "conv_pack_eval: layer %d cannot take the Winograd packing", l);
    }
    a.eps = eps;
    if (perm_src0) {
        SED_REQUIRE(perm_src1 && perm_dst && perm_rows > 0 && perm_C > 0 && perm_Fp > 0, "conv_pack_eval: bad permutation");
        a.perm_src[0] = perm_src0; a.perm_src[1] = perm_src1; a.perm_dst = perm_dst; a.perm_rows = perm_rows; a.perm_C = perm_C; a.perm_Fp = perm_Fp;
    }
    return pack_multi_launch(a, n, stream);
}

// ── 3-term bf16-split path (EXPERIMENT, explicit opt-in: mode 1 of the *_ex entries; never the default) ──
// w = hi + lo with hi = bf16(w), lo = bf16(w - hi); a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16
// (fp32 accumulate): three bf16 MFMAs of 32 cycles per 16 k against eight fp32 MFMAs of 64 cycles, i.e. 5.3x the matrix
// rate; the dropped lo*lo term and the 16-bit split leave a relative error of ~4e-6 on a K = 1152 sum (fp32: 3e-7).
// Fragment order: [tap][ci/32][(ci%32)/16][co/32][hi|lo][lane = co%32 + 32*((ci%16)/8)][ci%8] bf16 — the B operand of one
// 32x32x16 MFMA per 1 KiB wave-load; same byte count as the fp32 packing, so it lives in the same workspace region.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline size_t conv_frag_index_b(int tap, int co, int ci, int Cout, int Cin, int part) {
    int cc = ci >> 5, g = (ci & 31) >> 4, h = (ci & 15) >> 3, j = ci & 7;
    int cot = co >> 5, r = co & 31;
    return (((((((size_t)tap * (Cin >> 5) + cc) * 2 + g) * (Cout >> 5) + cot) * 2 + part) * 64) + r + 32 * h) * 8 + j;
}

__global__ void conv_pack_w_bf16x3_k(const float* __restrict__ w, __bf16* __restrict__ wf, __bf16* __restrict__ wd, int Cout, int Cin) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int n = Cout * Cin * 9;
    if (i >= n) return;
    int tap = i % 9, ci = (i / 9) % Cin, co = i / (9 * Cin);
    const float v = w[i];
    const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
    if (wf) { wf[conv_frag_index_b(tap, co, ci, Cout, Cin, 0)] = hi; wf[conv_frag_index_b(tap, co, ci, Cout, Cin, 1)] = lo; }
    if (wd) { wd[conv_frag_index_b(8 - tap, ci, co, Cin, Cout, 0)] = hi; wd[conv_frag_index_b(8 - tap, ci, co, Cin, Cout, 1)] = lo; }
}

extern "C" int sed_conv3x3_pack_weights_ex(const float* w, float* wf, float* wd, int Cout, int Cin, int mode, void* stream) {
    if (mode == 0 || Cin % 32 != 0 || Cout % 32 != 0) return sed_conv3x3_pack_weights(w, wf, wd, Cout, Cin, stream);
    SED_REQUIRE(mode == 1, "conv3x3_pack_weights_ex: unknown mode %d", mode);
    SED_REQUIRE(w && Cout > 0 && Cin > 0, "conv3x3_pack_weights_ex: bad arguments");
    int n = Cout * Cin * 9;
    conv_pack_w_bf16x3_k<<<cdiv(n, 256), 256, 0, as_stream(stream)>>>(w, (__bf16*)wf, (__bf16*)wd, Cout, Cin);
    SED_LAUNCH_CHECK("conv_pack_w_bf16x3");
    return 0;
}

// ───────────────────────── small direct forward ─────────────────────────
// Channel counts the MFMA path does not take (Cin or Cout not a multiple of 32).  A workgroup computes NCO output channels
// (blockIdx.z) of one time block from NCI input channels [ci0, ci0 + NCI) held in LDS with their weights; wide layers run as
// several launches over the input-channel chunks, each adding to y (first: bias, last: the statistics partials).  A fallback:
// correct for any channel counts that are multiples of 4, not tuned.
__global__ __launch_bounds__(256) void conv3x3_small_fwd_k(
    const float* __restrict__ x, int x_nchw, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ stat, int B, int Cin, int F, int T, int Cout, int TT,
    int ci0, int NCI, int NCO, int first, int last) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int co0 = blockIdx.z * NCO;
    const int nco = (Cout - co0 < NCO) ? Cout - co0 : NCO;            // multiples of 4
    float* w_s = smem;                        // [9][NCI][nco]
    float* halo = smem + 9 * NCI * NCO;       // [TT+2][F+2][NCI]
    const int tid = threadIdx.x;
    const int b = blockIdx.y, t0 = blockIdx.x * TT;
    const int F2 = F + 2;

    for (int i = tid; i < 9 * NCI * nco; i += 256) {
        int ci = i % NCI, co = (i / NCI) % nco, tap = i / (NCI * nco);
        w_s[(tap * NCI + ci) * nco + co] = wp[((size_t)tap * Cout + co0 + co) * Cin + ci0 + ci];
    }
    const int hn = (TT + 2) * F2 * NCI;
    if (x_nchw) {
        for (int i = tid; i < hn; i += 256) {         // tt fastest: time is contiguous in NCHW
            int tt = i % (TT + 2), ff = (i / (TT + 2)) % F2, ci = i / ((TT + 2) * F2);
            int t = t0 + tt - 1, f = ff - 1;
            float v = 0.f;
            if (t >= 0 && t < T && f >= 0 && f < F) v = x[(((size_t)b * Cin + ci0 + ci) * F + f) * T + t];
            halo[(tt * F2 + ff) * NCI + ci] = v;
        }
    } else {
        for (int i = tid; i < hn; i += 256) {
            int ci = i % NCI, ff = (i / NCI) % F2, tt = i / (NCI * F2);
            int t = t0 + tt - 1, f = ff - 1;
            float v = 0.f;
            if (t >= 0 && t < T && f >= 0 && f < F) v = x[(((size_t)b * T + t) * F + f) * Cin + ci0 + ci];
            halo[i] = v;
        }
    }
    __syncthreads();

    const int ncg = nco >> 2;
    const int nslots = 256 / ncg;
    const int cg = tid % ncg, slot = tid / ncg;
    const bool active = slot < nslots;
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    if (active) {
        f32x4 bv = {0, 0, 0, 0};
        if (bias && first) bv = *(const f32x4*)(bias + co0 + cg * 4);
        for (int p = slot; p < TT * F; p += nslots) {
            int tl = p / F, f = p - tl * F;
            if (t0 + tl >= T) break;
            float* yp = y + (((size_t)b * T + t0 + tl) * F + f) * Cout + co0 + cg * 4;
            f32x4 acc = first ? bv : *(const f32x4*)yp;
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw) {
                    const float* hp = halo + ((tl + kw) * F2 + f + kh) * NCI;
                    const float* wq = w_s + ((kh * 3 + kw) * NCI) * nco + cg * 4;
                    for (int ci = 0; ci < NCI; ++ci) {
                        float xv = hp[ci];
                        f32x4 w4 = *(const f32x4*)(wq + ci * nco);
                        acc += xv * w4;
                    }
                }
            *(f32x4*)yp = acc;
            s1 += acc;
            s2 += acc * acc;
        }
    }
    if (stat && last) {
        __syncthreads();
        float* red = smem;                     // [2][nslots][nco]
        if (active) {
            *(f32x4*)(red + (slot)*nco + cg * 4) = s1;
            *(f32x4*)(red + (nslots + slot) * nco + cg * 4) = s2;
        }
        __syncthreads();
        const size_t row = (size_t)b * gridDim.x + blockIdx.x;
        for (int i = tid; i < 2 * nco; i += 256) {
            int which = i / nco, co = i - which * nco;
            float a = 0.f;
            for (int s = 0; s < nslots; ++s) a += red[(which * nslots + s) * nco + co];
            stat[row * 2 * Cout + (size_t)which * Cout + co0 + co] = a;
        }
    }
}

// small, Cin <= 4: the 9*CIN weight float4 of a thread's four output channels live in registers (the generic kernel
// reads them from LDS for every FMA group, which bounds it at ~1.7 TB/s of output at Cin = 4); LDS holds the halo only.
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_small_fwd_c_k(
    const float* __restrict__ x, int x_nchw, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ stat, int B, int F, int T, int Cout, int TT) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* halo = smem;                       // [TT+2][F+2][CIN]; reused as [2][nslots][Cout] for the statistics
    const int tid = threadIdx.x;
    const int b = blockIdx.y, t0 = blockIdx.x * TT;
    const int F2 = F + 2;
    const int ncg = Cout >> 2;
    const int nslots = 256 / ncg;
    const int cg = tid % ncg, slot = tid / ncg;
    const bool active = slot < nslots;

    f32x4 w[9 * CIN];
    f32x4 bv = {0, 0, 0, 0};
    if (active) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int k = 0; k < 4; ++k) w[tap * CIN + ci][k] = wp[((size_t)tap * Cout + cg * 4 + k) * CIN + ci];
        if (bias) bv = *(const f32x4*)(bias + cg * 4);
    }
    const int hn = (TT + 2) * F2 * CIN;
    for (int i = tid; i < hn; i += 256) {
        int tt, ff, ci;
        if (x_nchw) { tt = i % (TT + 2); ff = (i / (TT + 2)) % F2; ci = i / ((TT + 2) * F2); }   // time contiguous in NCHW
        else { ci = i % CIN; ff = (i / CIN) % F2; tt = i / (CIN * F2); }
        int t = t0 + tt - 1, f = ff - 1;
        float v = 0.f;
        if (t >= 0 && t < T && f >= 0 && f < F)
            v = x_nchw ? x[(((size_t)b * CIN + ci) * F + f) * T + t] : x[(((size_t)b * T + t) * F + f) * CIN + ci];
        halo[(tt * F2 + ff) * CIN + ci] = v;
    }
    __syncthreads();

    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    if (active) {
        for (int p = slot; p < TT * F; p += nslots) {
            int tl = p / F, f = p - tl * F;
            if (t0 + tl >= T) break;
            f32x4 acc = bv;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float* hp = halo + ((tl + kw) * F2 + f + kh) * CIN;
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) acc += hp[ci] * w[(kh * 3 + kw) * CIN + ci];
                }
            *(f32x4*)(y + (((size_t)b * T + t0 + tl) * F + f) * Cout + cg * 4) = acc;
            s1 += acc;
            s2 += acc * acc;
        }
    }
    if (stat) {
        __syncthreads();
        float* red = smem;                     // [2][nslots][Cout]
        if (active) {
            *(f32x4*)(red + (slot)*Cout + cg * 4) = s1;
            *(f32x4*)(red + (nslots + slot) * Cout + cg * 4) = s2;
        }
        __syncthreads();
        const size_t row = (size_t)b * gridDim.x + blockIdx.x;
        for (int i = tid; i < 2 * Cout; i += 256) {
            int which = i / Cout, co = i - which * Cout;
            float a = 0.f;
            for (int s = 0; s < nslots; ++s) a += red[(which * nslots + s) * Cout + co];
            stat[row * 2 * Cout + i] = a;
        }
    }
}

// ───────────────────────── MFMA implicit-GEMM forward ─────────────────────────
// grid (ceil(T/TT) * nft, B, Cout/(32*NCT)); 256 threads = 4 waves; block tile = TT time rows x FT mel columns.
// wave w: co tile ct = w % NCT, row part mp = w / NCT; row tiles mt = mp + i*(4/NCT).
// Weights are streamed L2 -> registers in fragment order (one coalesced 1 KiB load per wave and k-group, no LDS, no
// per-tap barrier); the halo tile is double-buffered in LDS with the next 32-channel chunk's global loads issued before
// the MFMA loop of the current one: one barrier per chunk.
// BNR (the kernel run as a DATA GRADIENT, y = the gradient of the pooled output of the block below): the epilogue also
// forms that block's BatchNorm-backward sums, so its separate reduction pass (a re-read of its conv output) disappears.
// They follow from the block's own pooled OUTPUT q (forward, same shape as y) without the conv output, the arg-max or
// the dropout hash:   q = relu(max z) * keep_mask / (1-p)  is > 0 exactly where the gradient passes (kept AND gate open),
//   g = y / (1-p) there (0 elsewhere),   and the BatchNorm output at the arg-max is z = q (1-p), so xhat = (z - beta) / gamma.
// `stat` then receives (sum g, sum g*xhat) per block in the layout of the forward statistics (= sed_bn_bwd_finalize's
// input).  Recovering xhat from z divides a rounding error of eps |z| by gamma: for |gamma| << |beta| (z = gamma xhat + beta is
// nearly constant) it is lost — eps |beta / gamma| — and for gamma == 0 it does not exist (round-3 advisor).  Such channels
// contribute 0 to sum g*xhat here (sum g is exact for every channel) and get that sum from where the exact xhat is:
//   * a block below that stores its conv output: |gamma| < |beta| / 64 (error bound 4e-6 on the fast path) — the finalising
//     kernel recomputes those channels from the conv output, searching each window as the forward does
//     (sed_bn_bwd_finalize_small_gamma; a cold path inside a launch that exists anyway);
//   * a recomputed first block: gamma == 0 only here; its own passes form sum g*xhat for every channel (conv1_wgrad_assemble_k
//     exactly, from the R_k; the recomputing apply pass for gamma == 0).

// EV (inference, sed.py:128-141 with optim=None): BatchNorm is folded into the packed weights and the bias on the host side of
// the launch (running statistics are constants), and the epilogue applies ReLU + the (1,2) time pool before anything is
// written: `y` is the POOLED output [B][T/2][F][Cout]; the un-pooled tensor never exists (in + 0.5 out bytes per block).
template <int NCT, int MINW, bool BNR = false, int RGC = 0, bool EV = false>
__global__ __launch_bounds__(256, MINW) void conv3x3_mfma_fwd2_k(
    const float* __restrict__ x, const float* __restrict__ wq, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ stat, int B, int Cin, int F, int T, int Cout, int TT, int FT, int nft,
    float invF, float invF2, ConvBnRed br = ConvBnRed{}) {
    constexpr int MPARTS = 4 / NCT;
    constexpr int WROWS = 32 * NCT;
    constexpr bool RG = RGC > 0;                       // the block below is the recomputed first block with RGC input channels
    static_assert(!EV || (NCT == 4 && !BNR && RGC == 0), "the pooled inference epilogue: one wave owns all rows of its 32 channels");
    constexpr int EPI = EV ? 8 * 1024 + 256 : 4 * 1024 + 256;    // floats of transpose scratch (+ stat exchange) the epilogue needs
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int F2 = FT + 2;
    const int HR = (TT + 2) * F2;
    // LDS pitch of one halo time-row: F2 rows of CV_LD floats + CV_TPAD.  The pad makes the 16-B slot of flattened
    // position p equal (9*p + const) mod 16 across the mel wrap-around inside a 32-row MFMA tile (F2*9 + 14 = FT*9 mod 16),
    // so every ds_read_b128 lane group stays conflict-free (PMC: 35 % bank-conflict cycles without it).
    const int TP = F2 * CV_LD + CV_TPAD;
    const int HB = (TT + 2) * TP;
    // invF = 1 / FT, invF2 = 1 / (FT + 2) come from the host (sed_fdiv's reciprocals: two float divisions fewer per lane)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in launch order, so with the
    // plain (x = tile, y = sequence) order the neighbours of a tile — which re-read its halo rows and columns — sit on other
    // XCDs and every halo element comes from HBM again (1.375x the input at 8 x 20 tiles).  Launch index i = xcd + 8 j is
    // mapped to sequence xcd + 8 (j / tiles), tile j % tiles when the batch is a multiple of 8: one sequence's tiles share an L2.
    int bx = blockIdx.x, by = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int id = by * (int)gridDim.x + bx, xcd = id & 7, j = id >> 3;
        by = xcd + 8 * (j / (int)gridDim.x);
        bx = j % (int)gridDim.x;
    }
    const int tb = bx / nft, f0 = (bx - tb * nft) * FT;
    const int b = by, t0 = tb * TT, co0 = blockIdx.z * WROWS;
    const int ct = wave % NCT, mp = wave / NCT;
    const int MROWS = TT * FT;
    const int nMT = (MROWS + 31) >> 5;
    const int nchunks = Cin / CV_CIC, ncot = Cout >> 5, cot = blockIdx.z * NCT + ct;

    int abase[CV_MTW];
#pragma unroll
    for (int i = 0; i < CV_MTW; ++i) {
        int p = (mp + i * MPARTS) * 32 + r;
        if (p >= MROWS) p = MROWS - 1;
        int tl = sed_fdiv(p, invF), f = p - tl * FT;
        abase[i] = tl * TP + f * CV_LD + 4 * h;
    }
    // Accumulators start at the bias: a register holds one output channel (lane r) of 16 rows, so the epilogue has no add left.
    const float bv = bias ? bias[co0 + ct * 32 + r] : 0.f;
    f32x16 acc[CV_MTW];
#pragma unroll
    for (int i = 0; i < CV_MTW; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = bv;

    // Row-offset table of the epilogue, behind the halo buffers and the transpose scratch: entry [mt][rq][k] = byte offset of
    // output row p = 32 mt + rq + 8 k inside sequence b (all Cout channels of one position), ~0 for a row outside the output.
    // Computed once per workgroup here instead of once per lane and store there: while the co-resident workgroup is in its
    // MFMA loop (fp32 MFMAs occupy the VALU) this workgroup's vector instructions issue at about one per MFMA, so the
    // epilogue's duration IS its VALU instruction count x 64 cycles (measured with s_memrealtime stamps: 850 VALU instructions,
    // 22 us per workgroup, against 3.4 us with the CU to itself) — address arithmetic was half of them.
    unsigned* rowtab = (unsigned*)(smem + (2 * HB > EPI ? 2 * HB : EPI));
    // RG: a second table (the position of a row's 3 x 4 input window in xs) and xs itself, the (FT + 2) x (2 TT + 2) patch of
    // the 1-channel network input under this tile's pooled rows (time fastest, zero outside the input)
    unsigned* xofftab = rowtab + 32 * CV_MTW * MPARTS;
    float* xs = (float*)(xofftab + 32 * CV_MTW * MPARTS);
    const int XT = 2 * TT + 2;
    for (int sidx = tid; sidx < nMT * 32; sidx += 256) {
        const int p = (sidx & ~31) + ((sidx >> 2) & 7) + 8 * (sidx & 3);
        const int tl = sed_fdiv(p, invF), f = f0 + p - tl * FT;
        const bool ok = p < MROWS && t0 + tl < T && f < F;
        if (EV)     // a row of the table = the SECOND row of a time pair that floor pooling keeps -> its pooled output row (t0 is even)
            rowtab[sidx] = (ok && (tl & 1) && t0 + tl < (T & ~1)) ? (unsigned)((((t0 + tl) >> 1) * F + f) * Cout) * 4u : 0xFFFFFFFFu;
        else
            rowtab[sidx] = ok ? (unsigned)(((t0 + tl) * F + f) * Cout) * 4u : 0xFFFFFFFFu;
        if (RG) xofftab[sidx] = (unsigned)((p - tl * FT) * XT + 2 * tl);
    }
    if (RG) {
        for (int i = tid; i < F2 * XT; i += 256) {
            const int ff = sed_fdiv(i, br.invXT), tt = i - ff * XT;
            const int f = f0 + ff - 1, t = 2 * t0 + tt - 1;
            const bool in = f >= 0 && f < F && t >= 0 && t < br.Ty;
#pragma unroll
            for (int ci = 0; ci < RGC; ++ci)
                xs[ci * F2 * XT + i] = in ? br.x1[(((size_t)b * RGC + ci) * F + f) * br.Ty + t] : 0.f;
        }
    }

    // Halo staging.  Every lane issues all CV_NH loads of a chunk from a clamped (always valid) address and zeroes the
    // out-of-range ones when it commits them: a fixed number of load instructions per chunk is what lets the hand-counted
    // vmcnt waits of the weight fragments below step over a halo prefetch in flight.
    f32x4 ph[CV_NH];
    int pdst[CV_NH];                      // LDS float offset of each staged float4 (-1: none)
    unsigned hoff[CV_NH];                 // element offset of each staged float4 in x, chunk 0
    unsigned hmask = 0;                   // bit u: the float4 is inside the input (else zero padding)
#pragma unroll
    for (int u = 0; u < CV_NH; ++u) {
        int i = tid + u * 256;
        int row = i >> 3, tt = sed_fdiv(row, invF2), ff = row - tt * F2;
        pdst[u] = (i < HR * 8) ? tt * TP + ff * CV_LD + (i & 7) * 4 : -1;
        int t = t0 + tt - 1, f = f0 + ff - 1;
        if (i < HR * 8 && t >= 0 && t < T && f >= 0 && f < F) hmask |= 1u << u;
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        f = f < 0 ? 0 : (f >= F ? F - 1 : f);
        hoff[u] = (((unsigned)b * T + t) * F + f) * Cin + (i & 7) * 4;       // < 2^32 elements: checked by the host
    }
    // Both kinds of global load in the loop are issued from inline asm and waited for with hand-counted vmcnt (see the
    // weight fragments below): left to hipcc, the conditional prefetch made it wait for everything outstanding before each
    // re-issue into the staging registers.
    auto fetch = [&](int cc) {
#pragma unroll
        for (int u = 0; u < CV_NH; ++u) {
            const float* pu = x + (size_t)hoff[u] + cc * CV_CIC;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ph[u]) : "v"(pu));
        }
    };
    auto commit = [&](float* buf) {       // caller: the loads have retired (a vmcnt wait that covers them has been executed)
#pragma unroll
        for (int u = 0; u < CV_NH; ++u) asm volatile("" : "+v"(ph[u]));
#pragma unroll
        for (int u = 0; u < CV_NH; ++u) {
            if (pdst[u] >= 0) *(f32x4*)(buf + pdst[u]) = ((hmask >> u) & 1) ? ph[u] : (f32x4){0, 0, 0, 0};
        }
    };
    // Weight fragments: four 1 KiB wave loads per (chunk, tap) step, straight from L2 in MFMA B-fragment order, one step
    // ahead.  They are issued from inline asm and waited for with hand-counted vmcnt: hipcc's own counting gives up at the
    // loop back-edge and waited for the loads it had just issued (vmcnt(1) at the first MFMA of every tap: the L2 round trip
    // exposed nine times per chunk, and with it the halo prefetch, which retires in order in front of them).
    const f32x4* wl = (const f32x4*)wq + (size_t)cot * 64 + lane;
    auto load_b = [&](f32x4* bq, int cc, int tap) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4* pg = wl + (((size_t)tap * nchunks + cc) * 4 + g) * ncot * 64;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq[g]) : "v"(pg));
        }
    };
    auto bind = [&](f32x4* bq) {          // the fragments are valid from here on (orders their uses after the wait)
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(bq[g]));
    };

    fetch(0);
    asm volatile("s_waitcnt vmcnt(0)");
    commit(smem);
    f32x4 bfs[2][4];
    load_b(bfs[0], 0, 0);
    __syncthreads();
    const int nsteps = 9 * nchunks;                       // (chunk, tap) steps, two per loop iteration: the two register sets
    auto step = [&](int st, int cc, int tap, f32x4* cur, f32x4* nxt) {      // swap roles without a copy
        const bool more = cc + 1 < nchunks;
        const bool last = st + 1 >= nsteps;
        if (!last) load_b(nxt, tap < 8 ? cc : cc + 1, tap < 8 ? tap + 1 : 0);
        if (tap == 0 && more) fetch(cc + 1);              // AFTER the tap-1 fragments: vmcnt retires in order
        __builtin_amdgcn_sched_barrier(0);
        // `cur` was issued one step ago.  Younger than it: this step's 4 fragment loads and, in taps 0 and 1 of a chunk that
        // prefetches, the CV_NH halo loads issued in tap 0.  (A smaller count than necessary only waits longer.)
        // Measured alternatives: one load per 8-k group spread over the tap (-4 %), nine taps fully unrolled (-2 %).
        if (last) asm volatile("s_waitcnt vmcnt(0)");
        else if (more && tap < 2) { static_assert(CV_NH == 8, "the vmcnt immediate below is 4 + CV_NH"); asm volatile("s_waitcnt vmcnt(12)"); }
        else asm volatile("s_waitcnt vmcnt(4)");
        bind(cur);
        const float* halo = smem + (cc & 1) * HB;
        const int kh = tap / 3, kw = tap - kh * 3;
        const int toff = kw * TP + kh * CV_LD;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 af[CV_MTW];
#pragma unroll
            for (int i = 0; i < CV_MTW; ++i)
                af[i] = *(const f32x4*)(halo + abase[i] + toff + g * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < CV_MTW; ++i)          // tiles past nMT read clamped rows and are never stored
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][j], cur[g][j], acc[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tap == 8) {
            if (more) commit(smem + ((cc + 1) & 1) * HB);     // prefetched in tap 0: retired by the vmcnt(4) of taps 2..8
            __syncthreads();
        }
    };
    {
        int cc = 0, tap = 0;
#pragma unroll 1
        for (int st = 0; st < nsteps; st += 2) {
            step(st, cc, tap, bfs[0], bfs[1]);
            if (++tap == 9) { tap = 0; ++cc; }
            if (st + 1 < nsteps) {
                step(st + 1, cc, tap, bfs[1], bfs[0]);
                if (++tap == 9) { tap = 0; ++cc; }
            }
        }
    }

    // Epilogue.  An accumulator register holds ONE output channel per lane (32 channels x 2 rows per register), so storing it
    // directly is 80 global_store_dword per lane and tile set, 128-byte pieces: store-issue bound, and with both co-resident
    // workgroups of a CU in lock step nothing hides it (~19 % of the kernel).  Each 32x32 tile is instead transposed through
    // 4 KB of the (now free) halo buffer: 16 ds_write_b32, then 4 ds_read_b128 give every lane 4 consecutive channels of a
    // row, and the tile leaves in 4 global_store_dwordx4 per lane (8 full 128-byte rows per instruction).  A 32-float row
    // stride is conflict-free for both the b32 writes and the b128 lane groups.
    f32x2 s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
    float s1 = 0.f, s2 = 0.f;
    float* tsc = smem + wave * 1024;                 // the last loop barrier already passed: the halo buffers are free
    const int rq = lane >> 3, c4 = (lane & 7) * 4;
    // a tile that lies wholly inside the output (the common case) skips the per-element range checks: block-uniform branch
    const bool interior = (MROWS == nMT * 32) && (t0 + TT <= T) && (f0 + FT <= F);
    // wave-uniform base of this wave's 32 channels in sequence b; a lane adds its row offset (table) and 4 c4 bytes
    char* const yb = (char*)(y + (size_t)b * (EV ? T >> 1 : T) * F * Cout + co0 + ct * 32);
    const char* const qb = BNR ? (const char*)(br.pooled + (size_t)b * T * F * Cout + co0 + ct * 32) : nullptr;
    // BNR: per-lane constants of this lane's four channels (transposed phase: lane = row group rq, channels c4..c4+3):
    // xhat = q (1-p)/gamma - beta/gamma = q * q_kr + q_nb
    f32x4 q_kr = {0, 0, 0, 0}, q_nb = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
    if (BNR) {
        const int cb = co0 + ct * 32 + c4;
        const f32x4 q_beta = *(const f32x4*)(br.beta + cb);
        const f32x4 gm = *(const f32x4*)(br.gamma + cb);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool zero = gm[e] == 0.f || (br.ybelow != nullptr && fabsf(gm[e]) * 64.f < fabsf(q_beta[e]));
            const float rg = zero ? 0.f : 1.0f / gm[e];
            q_kr[e] = br.keep * rg;
            q_nb[e] = -q_beta[e] * rg;
        }
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // RG: the first block's weight-gradient sums R[kh*3+kw] = sum g~ x[f+kh-1][2t'+sel+kw-1] (sel = the arg-max bit), which
    // conv1_rgrad_k otherwise forms in a pass of its own over dx (this kernel's output), the pooled tensor and the bits: 750 MB
    // that trailed the last weight gradient by 0.13 ms.  Here dx and the pooled values are in registers already.
    f32x4 R[RG ? 9 * RGC : 1];
#pragma unroll
    for (int k = 0; k < (RG ? 9 * RGC : 1); ++k) R[k] = (f32x4){0, 0, 0, 0};
    const unsigned char* const bq = RG ? br.bits + (size_t)b * T * F * (Cout >> 2) + ((co0 + ct * 32 + c4) >> 2) : nullptr;
    auto store_tiles = [&](auto checked) {
        constexpr bool CHK = decltype(checked)::value;
#pragma unroll
        for (int i = 0; i < CV_MTW; ++i) {
            int mt = mp + i * MPARTS;
            if (EV) {
                if (mt < nMT) {
                    // Pooled inference epilogue.  The wave owns every row of its 32 channels (mt = i), so the time pair
                    // (tl, tl + 1) of a mel column — rows p and p + FT, FT <= 32 — lies in this tile or the previous one:
                    // a two-tile ring of transposed tiles (8 KB per wave in the free halo buffers) holds both when the second
                    // arrives.  Lane (rq, c4) then takes the rows rq + 8 k of this tile that close a pair: max(a, b, 0) of four
                    // channels, one dwordx4 store to the pooled row the table names.
                    float* ring = smem + wave * 2048;
                    const int slot = (mt & 1) * 1024;
#pragma unroll
                    for (int j = 0; j < 16; ++j) ring[slot + ((j & 3) + 8 * (j >> 2) + 4 * h) * 32 + r] = acc[i][j];
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const u32x4 rr = *(const u32x4*)(rowtab + mt * 32 + rq * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (rr[k] != 0xFFFFFFFFu) {
                            const int pb = mt * 32 + rq + 8 * k;
                            const f32x4 vb = *(const f32x4*)(ring + (pb & 63) * 32 + c4);
                            const f32x4 va = *(const f32x4*)(ring + ((pb - FT) & 63) * 32 + c4);
                            f32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaxf(va[e], vb[e]), 0.f);
                            *(f32x4*)(yb + rr[k] + (unsigned)(c4 * 4)) = o;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            } else if (mt < nMT) {
                const u32x4 rr = *(const u32x4*)(rowtab + mt * 32 + rq * 4);              // rows rq + 8 k of this tile
                const u32x4 ro = rr + (unsigned)(c4 * 4);
                u32x4 xo = {0, 0, 0, 0};
                unsigned bt[4] = {0, 0, 0, 0};
                if (RG) {
                    xo = *(const u32x4*)(xofftab + mt * 32 + rq * 4);
                    if (RGC < 2) {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (!CHK || rr[k] != 0xFFFFFFFFu) bt[k] = bq[rr[k] >> 4];
                    }
                }
                // BNR: this tile's pooled values, requested before the transpose so that they arrive under it
                f32x4 pq[4];
                constexpr bool PQ_EARLY = RGC < 2;      // two input channels: 18 tap accumulators leave no room for four prefetched rows
                if (BNR && PQ_EARLY) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        pq[k] = (f32x4){0, 0, 0, 0};
                        if (!CHK || rr[k] != 0xFFFFFFFFu) pq[k] = *(const f32x4*)(qb + ro[k]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) tsc[((j & 3) + 8 * (j >> 2) + 4 * h) * 32 + r] = acc[i][j];
                if (!BNR) {
                    if (!CHK) {
#pragma unroll
                        for (int j = 0; j < 16; j += 2) {
                            const f32x2 v = {acc[i][j], acc[i][j + 1]};
                            s1v += v;
                            s2v += v * v;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const int p = mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
                            const int tl = sed_fdiv(p, invF), f = f0 + p - tl * FT;
                            if (p < MROWS && t0 + tl < T && f < F) {
                                s1 += acc[i][j];
                                s2 += acc[i][j] * acc[i][j];
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // one wave: LDS ops complete in order; compiler order only
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 v = *(const f32x4*)(tsc + (rq + 8 * k) * 32 + c4);
                    const bool ok = !CHK || rr[k] != 0xFFFFFFFFu;
                    if (BNR && !PQ_EARLY) {
                        pq[k] = (f32x4){0, 0, 0, 0};
                        if (ok) { pq[k] = *(const f32x4*)(qb + ro[k]); bt[k] = bq[rr[k] >> 4]; }
                    }
                    if (ok) *(f32x4*)(yb + ro[k]) = v;
                    if (BNR && ok) {
                        // g = v / (1-p) where the pooled value is > 0; the 1 / (1-p) is applied to the sums at the end
                        f32x4 g0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) g0[e] = pq[k][e] > 0.f ? v[e] : 0.f;
                        a1 += g0;
                        a2 += g0 * (pq[k] * q_kr + q_nb);
                        if (RG) {
                            f32x4 g1;                         // the share of the window's second time row
#pragma unroll
                            for (int e = 0; e < 4; ++e) g1[e] = ((bt[k] >> e) & 1u) ? g0[e] : 0.f;
                            const f32x4 gA = g0 - g1;
#pragma unroll
                            for (int ci = 0; ci < RGC; ++ci) {
                                const float* xb = xs + ci * F2 * XT + xo[k];
#pragma unroll
                                for (int kh = 0; kh < 3; ++kh) {
                                    const f32x2 x01 = *(const f32x2*)(xb + kh * XT), x23 = *(const f32x2*)(xb + kh * XT + 2);
                                    R[(kh * 3 + 0) * RGC + ci] += gA * x01[0] + g1 * x01[1];
                                    R[(kh * 3 + 1) * RGC + ci] += gA * x01[1] + g1 * x23[0];
                                    R[(kh * 3 + 2) * RGC + ci] += gA * x23[0] + g1 * x23[1];
                                }
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    };
    if (interior && !EV) store_tiles(std::false_type{});
    else store_tiles(std::true_type{});
    if (stat && !EV) {
        float* red = smem + 4 * 1024;               // [4 waves][2][32], behind the four transpose scratches
        if (BNR) {
            a1 *= br.inv_keep;
            a2 *= br.inv_keep;
            // lanes with equal c4 (8 row groups) hold partial sums of the same four channels
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) { a1[e] += __shfl_xor(a1[e], o, 64); a2[e] += __shfl_xor(a2[e], o, 64); }
            }
            if (lane < 8) {
                *(f32x4*)(red + (wave * 2 + 0) * 32 + c4) = a1;
                *(f32x4*)(red + (wave * 2 + 1) * 32 + c4) = a2;
            }
            if (RG) {                                 // [4 waves][9 RGC][32] behind `red`: this wave's 32 channels of the R_k
                float* red2 = smem + 4 * 1024 + 256;
#pragma unroll
                for (int k = 0; k < 9 * RGC; ++k) {
                    R[k] *= br.inv_keep;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                        for (int o = 8; o < 64; o <<= 1) R[k][e] += __shfl_xor(R[k][e], o, 64);
                    }
                    if (lane < 8) *(f32x4*)(red2 + (wave * 9 * RGC + k) * 32 + c4) = R[k];
                }
            }
        } else {
            s1 += s1v[0] + s1v[1];
            s2 += s2v[0] + s2v[1];
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (h == 0) { red[(wave * 2 + 0) * 32 + r] = s1; red[(wave * 2 + 1) * 32 + r] = s2; }
        }
        __syncthreads();
        const size_t row = (size_t)b * gridDim.x + bx;
        if (tid < 2 * WROWS) {
            int which = tid / WROWS, c = tid - which * WROWS;
            int cti = c >> 5, cr = c & 31;
            float a = 0.f;
#pragma unroll
            for (int m = 0; m < MPARTS; ++m) a += red[((m * NCT + cti) * 2 + which) * 32 + cr];
            stat[row * 2 * Cout + which * Cout + co0 + c] = a;
        }
        if (RG && tid < WROWS) {                      // MPARTS == 1: wave ct owns channels 32 ct .. 32 ct + 31
            const int cti = tid >> 5, cr = tid & 31;
            float* o = br.rgp + (row * Cout + co0 + tid) * (1 + 9 * RGC);
            o[0] = red[(cti * 2 + 0) * 32 + cr];      // sum g (= the BatchNorm sum above)
            const float* red2 = smem + 4 * 1024 + 256;
#pragma unroll
            for (int k = 0; k < 9 * RGC; ++k) o[1 + k] = red2[(cti * 9 * RGC + k) * 32 + cr];
        }
    }
}

template <int NCT, int MINW>
__global__ __launch_bounds__(256, MINW) void conv3x3_mfma_fwd_bf16x3_k(
    const float* __restrict__ x, const float* __restrict__ wq, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ stat, int B, int Cin, int F, int T, int Cout, int TT, int FT, int nft) {
    constexpr int MPARTS = 4 / NCT;
    constexpr int WROWS = 32 * NCT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int F2 = FT + 2;
    const int HR = (TT + 2) * F2;
    // LDS pitch of one halo time-row: F2 rows of CV_LD floats + CV_TPAD.  The pad makes the 16-B slot of flattened
    // position p equal (9*p + const) mod 16 across the mel wrap-around inside a 32-row MFMA tile (F2*9 + 14 = FT*9 mod 16),
    // so every ds_read_b128 lane group stays conflict-free (PMC: 35 % bank-conflict cycles without it).
    const int TP = F2 * CV_LD + CV_TPAD;
    const int HB = (TT + 2) * TP;
    const float invF = 1.0f / (float)FT, invF2 = 1.0f / (float)F2;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int tb = blockIdx.x / nft, f0 = (blockIdx.x - tb * nft) * FT;
    const int b = blockIdx.y, t0 = tb * TT, co0 = blockIdx.z * WROWS;
    const int ct = wave % NCT, mp = wave / NCT;
    const int MROWS = TT * FT;
    const int nMT = (MROWS + 31) >> 5;
    const int nchunks = Cin / CV_CIC, ncot = Cout >> 5, cot = blockIdx.z * NCT + ct;

    int abase[CV_MTW];
#pragma unroll
    for (int i = 0; i < CV_MTW; ++i) {
        int p = (mp + i * MPARTS) * 32 + r;
        if (p >= MROWS) p = MROWS - 1;
        int tl = sed_fdiv(p, invF), f = p - tl * FT;
        abase[i] = tl * TP + f * CV_LD + 4 * h;
    }
    f32x16 acc[CV_MTW];
#pragma unroll
    for (int i = 0; i < CV_MTW; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

    // staging and weight fragments as in the fp32 kernel (asm loads, hand-counted vmcnt, clamped unconditional halo loads), with
    // the fragments TWO steps ahead in three register sets: a step is only 30 MFMAs of 32 cycles here, shorter than an L2 trip
    f32x4 ph[CV_NH];
    int pdst[CV_NH];                      // LDS float offset of each staged float4 (-1: none)
    unsigned hoff[CV_NH];                 // element offset of each staged float4 in x, chunk 0
    unsigned hmask = 0;                   // bit u: the float4 is inside the input (else zero padding)
#pragma unroll
    for (int u = 0; u < CV_NH; ++u) {
        int i = tid + u * 256;
        int row = i >> 3, tt = sed_fdiv(row, invF2), ff = row - tt * F2;
        pdst[u] = (i < HR * 8) ? tt * TP + ff * CV_LD + (i & 7) * 4 : -1;
        int t = t0 + tt - 1, f = f0 + ff - 1;
        if (i < HR * 8 && t >= 0 && t < T && f >= 0 && f < F) hmask |= 1u << u;
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        f = f < 0 ? 0 : (f >= F ? F - 1 : f);
        hoff[u] = (((unsigned)b * T + t) * F + f) * Cin + (i & 7) * 4;       // < 2^32 elements: checked by the host
    }
    auto fetch = [&](int cc) {
#pragma unroll
        for (int u = 0; u < CV_NH; ++u) {
            const float* pu = x + (size_t)hoff[u] + cc * CV_CIC;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ph[u]) : "v"(pu));
        }
    };
    // a position's 32 channels are stored as 32 hi bf16 (64 B) followed by 32 lo bf16 (64 B): the same 128-byte row and
    // 16-byte slot structure as the fp32 kernel (hi k-group g <-> fp32 group g, lo k-group g <-> fp32 group 2+g), so the
    // conflict-free pitch analysis carries over
    auto commit = [&](float* buf) {       // caller: the loads have retired (a vmcnt wait that covers them has been executed)
#pragma unroll
        for (int u = 0; u < CV_NH; ++u) asm volatile("" : "+v"(ph[u]));
#pragma unroll
        for (int u = 0; u < CV_NH; ++u) {
            if (pdst[u] >= 0) {
                const f32x4 v = ((hmask >> u) & 1) ? ph[u] : (f32x4){0, 0, 0, 0};
                bf16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) { hi[e] = (__bf16)v[e]; lo[e] = (__bf16)(v[e] - (float)hi[e]); }
                const int q = (tid + u * 256) & 7;
                float* rowp = buf + pdst[u] - 4 * q;
                *(bf16x4*)(rowp + 2 * q) = hi;
                *(bf16x4*)(rowp + 16 + 2 * q) = lo;
            }
        }
    };
    // bq[2g + part]: k-group g (16 channels), part 0 = hi, 1 = lo; one 1 KiB wave-load each
    const bf16x8* wl = (const bf16x8*)wq + (size_t)cot * 128 + lane;
    auto load_b = [&](bf16x8* bq, int cc, int tap) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const bf16x8* pg = wl + (((size_t)tap * nchunks + cc) * 2 + g) * ncot * 128 + part * 64;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq[2 * g + part]) : "v"(pg));
            }
    };

    fetch(0);
    asm volatile("s_waitcnt vmcnt(0)");
    commit(smem);
    bf16x8 bfs[3][4];
    const int nsteps = 9 * nchunks;                       // (chunk, tap) steps, three per loop iteration (nsteps is a multiple of 9)
    load_b(bfs[0], 0, 0);
    load_b(bfs[1], 0, 1);                                 // nsteps >= 9
    __syncthreads();
    // Step st uses the set issued two steps ago and issues the set of step st + 2.  Younger than `cur` when it is waited for: the
    // sets of steps st + 1 and st + 2 as far as they exist (4 loads each) and, in taps 0..2 of a chunk that prefetches, the
    // CV_NH halo loads issued in tap 0 right behind that step's set.  (A smaller count than necessary only waits longer.)
    auto step = [&](int st, int cc, int tap, bf16x8* cur, bf16x8* nn) {
        const bool more = cc + 1 < nchunks;
        const int ahead = nsteps - 1 - st;                // steps after this one
        if (ahead >= 2) {
            int c2 = cc, t2 = tap + 2;
            if (t2 >= 9) { t2 -= 9; ++c2; }
            load_b(nn, c2, t2);
        }
        if (tap == 0 && more) fetch(cc + 1);
        __builtin_amdgcn_sched_barrier(0);
        static_assert(CV_NH == 8, "the vmcnt immediates below are 4*k + CV_NH");
        if (ahead >= 2) {
            if (more && tap < 3) asm volatile("s_waitcnt vmcnt(16)");
            else asm volatile("s_waitcnt vmcnt(8)");
        } else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)");
        else asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(cur[g]));
        const float* halo = smem + (cc & 1) * HB;
        const int kh = tap / 3, kw = tap - kh * 3;
        const int toff = kw * TP + kh * CV_LD;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            bf16x8 ah[CV_MTW], al[CV_MTW];
#pragma unroll
            for (int i = 0; i < CV_MTW; ++i) {
                ah[i] = *(const bf16x8*)(halo + abase[i] + toff + g * 8);
                al[i] = *(const bf16x8*)(halo + abase[i] + toff + 16 + g * 8);
            }
#pragma unroll
            for (int i = 0; i < CV_MTW; ++i) {        // tiles past nMT read clamped rows and are never stored
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], cur[2 * g], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], cur[2 * g + 1], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], cur[2 * g], acc[i], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tap == 8) {
            if (more) commit(smem + ((cc + 1) & 1) * HB);     // prefetched in tap 0: retired by the vmcnt(8) of taps 3..8
            __syncthreads();
        }
    };
    {
        int cc = 0, tap = 0;
#pragma unroll 1
        for (int st = 0; st < nsteps; st += 3) {          // taps (0,1,2), (3,4,5), (6,7,8) of one chunk: the sets rotate in place
            step(st, cc, tap, bfs[0], bfs[2]);
            step(st + 1, cc, tap + 1, bfs[1], bfs[0]);
            step(st + 2, cc, tap + 2, bfs[2], bfs[1]);
            tap += 3;
            if (tap == 9) { tap = 0; ++cc; }
        }
    }

    // Epilogue.  An accumulator register holds ONE output channel per lane (32 channels x 2 rows per register), so storing it
    // directly is 80 global_store_dword per lane and tile set, 128-byte pieces: store-issue bound, and with both co-resident
    // workgroups of a CU in lock step nothing hides it (~19 % of the kernel).  Each 32x32 tile is instead transposed through
    // 4 KB of the (now free) halo buffer: 16 ds_write_b32, then 4 ds_read_b128 give every lane 4 consecutive channels of a
    // row, and the tile leaves in 4 global_store_dwordx4 per lane (8 full 128-byte rows per instruction).  A 32-float row
    // stride is conflict-free for both the b32 writes and the b128 lane groups.
    const int co = co0 + ct * 32 + r;
    const float bv = bias ? bias[co] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    float* tsc = smem + wave * 1024;                 // the last loop barrier already passed: the halo buffers are free
    const int rq = lane >> 3, c4 = (lane & 7) * 4;
    // a tile that lies wholly inside the output (the common case) skips the per-element range checks: block-uniform branch
    const bool interior = (MROWS == nMT * 32) && (t0 + TT <= T) && (f0 + FT <= F);
    auto store_tiles = [&](auto checked) {
        constexpr bool CHK = decltype(checked)::value;
#pragma unroll
        for (int i = 0; i < CV_MTW; ++i) {
            int mt = mp + i * MPARTS;
            if (mt < nMT) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    int row = (j & 3) + 8 * (j >> 2) + 4 * h;
                    float v = acc[i][j] + bv;
                    tsc[row * 32 + r] = v;
                    bool ok = true;
                    if (CHK) {
                        int p = mt * 32 + row;
                        int tl = sed_fdiv(p, invF), f = f0 + p - tl * FT;
                        ok = p < MROWS && t0 + tl < T && f < F;
                    }
                    if (ok) {
                        s1 += v;
                        s2 += v * v;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // one wave: LDS ops complete in order; compiler order only
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int row = rq + 8 * k;
                    int p = mt * 32 + row;
                    int tl = sed_fdiv(p, invF), f = f0 + p - tl * FT;
                    f32x4 v = *(const f32x4*)(tsc + row * 32 + c4);
                    if (!CHK || (p < MROWS && t0 + tl < T && f < F))
                        *(f32x4*)(y + (((size_t)b * T + t0 + tl) * F + f) * Cout + co0 + ct * 32 + c4) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    };
    if (interior) store_tiles(std::false_type{});
    else store_tiles(std::true_type{});
    if (stat) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        float* red = smem + 4 * 1024;               // [4 waves][2][32], behind the four transpose scratches
        if (h == 0) { red[(wave * 2 + 0) * 32 + r] = s1; red[(wave * 2 + 1) * 32 + r] = s2; }
        __syncthreads();
        const size_t row = (size_t)b * gridDim.x + blockIdx.x;
        if (tid < 2 * WROWS) {
            int which = tid / WROWS, c = tid - which * WROWS;
            int cti = c >> 5, cr = c & 31;
            float a = 0.f;
#pragma unroll
            for (int m = 0; m < MPARTS; ++m) a += red[((m * NCT + cti) * 2 + which) * 32 + cr];
            stat[row * 2 * Cout + which * Cout + co0 + c] = a;
        }
    }
}

extern "C" int sed_conv3x3_stat_rows(int B, int Cin, int F, int T, int Cout, int x_is_nchw) {
    ConvPlan p = conv_plan(B, Cin, F, T, Cout, x_is_nchw);
    return p.kind >= 0 ? p.rows : 0;
}

template <typename K>
static int set_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) { sed_set_error("hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e)); return (int)e; }
    }
    return 0;
}

// LDS of the data gradient with the first block's weight-gradient sums: + the window-offset table + the input patch
static size_t dgrad_rg_lds(const ConvPlan& p, int cin1) {
    return p.lds + (size_t)32 * CV_MTW * sizeof(unsigned) + ((size_t)cin1 * (2 * p.TT + 2) * (p.FT + 2) + 4) * sizeof(float);
}
// The RG epilogue parks its per-wave tap sums in `red2` = floats [4352, 4352 + 4*9*cin1*32) of the LDS, without a workgroup
// barrier; the row tables and the input patch, which slower waves may still be reading, start at float max(2 HB, 4352).  The
// two must not meet (round-3 advisor: for small tiles, e.g. F = 8 / T = 4, they did): such shapes keep conv1_rgrad_k.
static bool dgrad_rg_tile_ok(const ConvPlan& p, int cin1) {
    const size_t hb2 = (size_t)2 * (p.TT + 2) * ((p.FT + 2) * CV_LD + CV_TPAD);
    return hb2 >= (size_t)4 * 1024 + 256 + (size_t)4 * 9 * cin1 * 32;
}

extern "C" int sed_conv3x3_fwd(const float* x, int x_is_nchw, const float* wp, const float* bias, float* y,
                               float* stat, int B, int Cin, int F, int T, int Cout, void* stream) {
    return sed_conv3x3_fwd_ex(x, x_is_nchw, wp, bias, y, stat, B, Cin, F, T, Cout, 0, stream);
}

extern "C" int sed_conv3x3_fwd_ex(const float* x, int x_is_nchw, const float* wp, const float* bias, float* y,
                                  float* stat, int B, int Cin, int F, int T, int Cout, int mode, void* stream) {
    SED_REQUIRE(x && wp && y, "conv3x3_fwd: null pointer");
    SED_REQUIRE(mode == 0 || mode == 1, "conv3x3_fwd: unknown mode %d", mode);
    if (Cin % 32 != 0 || Cout % 32 != 0 || x_is_nchw) mode = 0;      // pack_weights_ex made the same decision
    SED_REQUIRE(B > 0 && Cin > 0 && F > 0 && T > 0 && Cout > 0, "conv3x3_fwd: bad shape B=%d Cin=%d F=%d T=%d Cout=%d", B, Cin, F, T, Cout);
    ConvPlan p = conv_plan(B, Cin, F, T, Cout, x_is_nchw);
    SED_REQUIRE(p.kind >= 0, "conv3x3_fwd: unsupported shape Cin=%d Cout=%d F=%d (need Cout%%4==0 and a tile that fits LDS)", Cin, Cout, F);
    hipStream_t s = as_stream(stream);
    const double npos = (double)B * T * F;
    SedProfScope prof(p.kind == 0 ? SED_K_CONV_SMALL_FWD : SED_K_CONV_MFMA_FWD, s,
                      p.kind == 0 ? 4.0 * npos * (Cin + Cout) : 2.0 * 9.0 * Cin * Cout * npos);
    if (p.kind == 0 && Cin <= 4) {
        size_t lds = (size_t)(p.TT + 2) * (F + 2) * Cin * sizeof(float), red = (size_t)2 * 256 * 4 * sizeof(float);
        if (lds < red) lds = red;
        dim3 grid(p.tblocks, B);
        switch (Cin) {
            case 1: conv3x3_small_fwd_c_k<1><<<grid, 256, lds, s>>>(x, x_is_nchw, wp, bias, y, stat, B, F, T, Cout, p.TT); break;
            case 2: conv3x3_small_fwd_c_k<2><<<grid, 256, lds, s>>>(x, x_is_nchw, wp, bias, y, stat, B, F, T, Cout, p.TT); break;
            case 3: conv3x3_small_fwd_c_k<3><<<grid, 256, lds, s>>>(x, x_is_nchw, wp, bias, y, stat, B, F, T, Cout, p.TT); break;
            default: conv3x3_small_fwd_c_k<4><<<grid, 256, lds, s>>>(x, x_is_nchw, wp, bias, y, stat, B, F, T, Cout, p.TT); break;
        }
    } else if (p.kind == 0) {
        SED_TRY(set_lds(conv3x3_small_fwd_k, p.lds));
        const dim3 grid(p.tblocks, B, cdiv(Cout, p.nco));
        for (int ci0 = 0; ci0 < Cin; ci0 += p.nci) {
            const int nci = Cin - ci0 < p.nci ? Cin - ci0 : p.nci;
            conv3x3_small_fwd_k<<<grid, 256, p.lds, s>>>(x, x_is_nchw, wp, bias, y, stat, B, Cin, F, T, Cout, p.TT, ci0, nci, p.nco,
                                                         ci0 == 0, ci0 + nci >= Cin);
        }
    } else {
        dim3 grid(p.tblocks * p.nft, B, Cout / (32 * p.nct));
        if ((size_t)B * T * F * Cin >= ((size_t)1 << 32)) {
            sed_set_error("conv3x3_fwd: input of %zu elements exceeds the 32-bit staging offsets of the MFMA kernels", (size_t)B * T * F * Cin);
            return -1;
        }
        if ((size_t)T * F * Cout >= ((size_t)1 << 30)) {
            sed_set_error("conv3x3_fwd: one sequence's output of %zu elements exceeds the 32-bit row offsets of the MFMA kernels", (size_t)T * F * Cout);
            return -1;
        }
        if (mode == 1) {                     // explicit opt-in: 3-term bf16-split MFMA (wp must come from pack_weights_ex(mode 1))
            if (p.nct == 4) {
                SED_TRY(set_lds((conv3x3_mfma_fwd_bf16x3_k<4, 2>), p.lds));
                conv3x3_mfma_fwd_bf16x3_k<4, 2><<<grid, 256, p.lds, s>>>(x, wp, bias, y, stat, B, Cin, F, T, Cout, p.TT, p.FT, p.nft);
            } else if (p.nct == 2) {
                SED_TRY(set_lds((conv3x3_mfma_fwd_bf16x3_k<2, 2>), p.lds));
                conv3x3_mfma_fwd_bf16x3_k<2, 2><<<grid, 256, p.lds, s>>>(x, wp, bias, y, stat, B, Cin, F, T, Cout, p.TT, p.FT, p.nft);
            } else {
                SED_TRY(set_lds((conv3x3_mfma_fwd_bf16x3_k<1, 2>), p.lds));
                conv3x3_mfma_fwd_bf16x3_k<1, 2><<<grid, 256, p.lds, s>>>(x, wp, bias, y, stat, B, Cin, F, T, Cout, p.TT, p.FT, p.nft);
            }
        } else if (p.nct == 4) {
            SED_TRY(set_lds((conv3x3_mfma_fwd2_k<4, 2>), p.lds));
            conv3x3_mfma_fwd2_k<4, 2><<<grid, 256, p.lds, s>>>(x, wp, bias, y, stat, B, Cin, F, T, Cout, p.TT, p.FT, p.nft, 1.0f / (float)p.FT, 1.0f / (float)(p.FT + 2));
        } else if (p.nct == 2) {
            SED_TRY(set_lds((conv3x3_mfma_fwd2_k<2, 2>), p.lds));
            conv3x3_mfma_fwd2_k<2, 2><<<grid, 256, p.lds, s>>>(x, wp, bias, y, stat, B, Cin, F, T, Cout, p.TT, p.FT, p.nft, 1.0f / (float)p.FT, 1.0f / (float)(p.FT + 2));
        } else {
            SED_TRY(set_lds((conv3x3_mfma_fwd2_k<1, 2>), p.lds));
            conv3x3_mfma_fwd2_k<1, 2><<<grid, 256, p.lds, s>>>(x, wp, bias, y, stat, B, Cin, F, T, Cout, p.TT, p.FT, p.nft, 1.0f / (float)p.FT, 1.0f / (float)(p.FT + 2));
        }
    }
    SED_LAUNCH_CHECK("conv3x3_fwd");
    return 0;
}

// ── inference: conv + BatchNorm (running statistics, folded) + ReLU + (1,2) time pool in one launch ──
static size_t conv_eval_lds(const ConvPlan& p) {
    size_t lds = (size_t)2 * (p.TT + 2) * ((p.FT + 2) * CV_LD + CV_TPAD) * sizeof(float);
    const size_t epi = (size_t)(8 * 1024 + 256) * sizeof(float);         // the two-tile transpose ring of four waves
    if (lds < epi) lds = epi;
    return lds + (size_t)32 * CV_MTW * sizeof(unsigned);
}
// The tile of the pooling epilogue: the 128-wide exact-fp32 MFMA tile with an EVEN number of time rows (a time pair never
// straddles two workgroups) of at most 32 mel columns (the pair lies within two consecutive 32-row tiles), two workgroups per
// CU.  Where the free choice of conv_tile is not such a tile, the best such tile is taken if it scores within 2 % of it —
// measured at 128 mel bins: 5 x 32 (free) 125 TFLOP/s, 6 x 26 119, 4 x 32 102: there the un-fused pair of launches is no slower,
// so those shapes keep it (kind = -1 here).
static ConvPlan conv_eval_plan(int B, int Cin, int F, int T, int Cout) {
    ConvPlan none{};
    none.kind = -1;
    if (B <= 0 || Cin <= 0 || F <= 0 || T < 2 || Cout <= 0) return none;
    const ConvPlan nat = conv_plan(B, Cin, F, T, Cout, 0);
    if (nat.kind != 1 || nat.nct != 4) return none;
    ConvPlan p = nat;
    if ((p.TT & 1) || p.FT > 32) {
        p = conv_plan(B, Cin, F, T, Cout, 0, true);
        if (p.kind != 1 || p.score < 0.98 * nat.score) return none;
    }
    return conv_eval_lds(p) <= 80 * 1024 ? p : none;
}
extern "C" int sed_conv3x3_bn_relu_pool_eval_supported(int B, int Cin, int F, int T, int Cout) {
    if (B <= 0 || Cin <= 0 || F <= 0 || T < 2 || Cout <= 0) return 0;
    if ((size_t)B * T * F * Cin >= ((size_t)1 << 32) || (size_t)T * F * Cout >= ((size_t)1 << 30)) return 0;
    return conv_eval_plan(B, Cin, F, T, Cout).kind == 1 ? 1 : 0;
}
extern "C" int sed_conv3x3_pack_weights_bn_folded(const float* w, const float* bias, const float* gamma, const float* beta,
                                                  const float* running_mean, const float* running_var, float eps,
                                                  float* wf, float* bias_folded, int Cout, int Cin, void* stream) {
    SED_REQUIRE(w && gamma && beta && running_mean && running_var && wf && bias_folded && Cout > 0 && Cin > 0, "conv3x3_pack_weights_bn_folded: bad arguments");
    ConvPackMulti a{};
    a.w[0] = w; a.wf[0] = wf; a.Cout[0] = Cout; a.Cin[0] = Cin; a.gamma[0] = gamma; a.beta[0] = beta; a.rm[0] = running_mean; a.rv[0] = running_var;
    a.bias[0] = bias; a.bias_out[0] = bias_folded; a.fold[0] = 1; a.eps = eps;
    return pack_multi_launch(a, 1, stream);
}
extern "C" int sed_conv3x3_bn_relu_pool_eval(const float* x, const float* wp_folded, const float* bias_folded, float* pooled,
                                             int B, int Cin, int F, int T, int Cout, void* stream) {
    SED_REQUIRE(x && wp_folded && bias_folded && pooled, "conv3x3_bn_relu_pool_eval: null pointer");
    SED_REQUIRE(sed_conv3x3_bn_relu_pool_eval_supported(B, Cin, F, T, Cout), "conv3x3_bn_relu_pool_eval: shape B=%d Cin=%d F=%d T=%d Cout=%d "
                "is not supported (sed_conv3x3_bn_relu_pool_eval_supported; use sed_conv3x3_fwd_ex + sed_bn_relu_pool_drop_fwd)", B, Cin, F, T, Cout);
    const ConvPlan p = conv_eval_plan(B, Cin, F, T, Cout);
    const size_t lds = conv_eval_lds(p);
    hipStream_t s = as_stream(stream);
    SedProfScope prof(SED_K_CONV_MFMA_FWD, s, 2.0 * 9.0 * Cin * Cout * (double)B * T * F);
    dim3 grid(p.tblocks * p.nft, B, Cout / 128);
    SED_TRY(set_lds((conv3x3_mfma_fwd2_k<4, 2, false, 0, true>), lds));
    conv3x3_mfma_fwd2_k<4, 2, false, 0, true><<<grid, 256, lds, s>>>(x, wp_folded, bias_folded, pooled, nullptr, B, Cin, F, T, Cout, p.TT, p.FT, p.nft,
                                                                   1.0f / (float)p.FT, 1.0f / (float)(p.FT + 2));
    SED_LAUNCH_CHECK("conv3x3_bn_relu_pool_eval");
    return 0;
}

// Data gradient of conv block l fused with the BatchNorm-backward reduction of block l-1 (see ConvBnRed above): the exact-fp32
// MFMA kernel with the BNR epilogue.  dy [B][T][F][C] -> dx [B][T][F][Cin] (= the gradient of block l-1's pooled output) and
// partials [rows][2][Cin] = (sum g, sum g*xhat) of block l-1, rows = sed_conv3x3_dgrad_bnred_rows (0: this shape does not take
// the MFMA path, use sed_conv3x3_fwd_ex + sed_bn_relu_pool_drop_bwd_reduce).
extern "C" int sed_conv3x3_dgrad_bnred_rows(int B, int C, int F, int T, int Cin) {
    ConvPlan p = conv_plan(B, C, F, T, Cin, 0);
    return (p.kind == 1 && p.nct == 4) ? p.rows : 0;
}

static int dgrad_bnred_impl(const float* dy, const float* wp_dgrad, float* dx, float* partials, const float* pooled,
                            const float* gamma, const float* beta, const float* conv_out_below, const float* mean,
                            const float* rstd, float drop_p, int pool_f, int pool_t, int Fy, int Ty,
                            const float* x1, int cin1, const unsigned char* bits, float* rg_partials,
                            int B, int C, int F, int T, int Cin, void* stream) {
    SED_REQUIRE(dy && wp_dgrad && dx && partials && pooled && gamma && beta && mean && rstd, "conv3x3_dgrad_bnred: null pointer");
    SED_REQUIRE(B > 0 && C > 0 && F > 0 && T > 0 && Cin > 0, "conv3x3_dgrad_bnred: bad shape B=%d C=%d F=%d T=%d Cin=%d", B, C, F, T, Cin);
    SED_REQUIRE(drop_p >= 0.f && drop_p < 1.f && pool_f >= 1 && pool_t >= 1, "conv3x3_dgrad_bnred: bad drop_p / pool");
    SED_REQUIRE(Ty / pool_t == T && Fy / pool_f == F, "conv3x3_dgrad_bnred: conv output %dx%d does not pool (%d,%d) to %dx%d", Ty, Fy, pool_f, pool_t, T, F);
    ConvPlan p = conv_plan(B, C, F, T, Cin, 0);
    SED_REQUIRE(p.kind == 1 && p.nct == 4, "conv3x3_dgrad_bnred: C=%d -> Cin=%d does not take the 128-wide MFMA path", C, Cin);
    if ((size_t)B * T * F * C >= ((size_t)1 << 32)) {
        sed_set_error("conv3x3_dgrad_bnred: input of %zu elements exceeds the 32-bit staging offsets of the MFMA kernels", (size_t)B * T * F * C);
        return -1;
    }
    if ((size_t)T * F * Cin >= ((size_t)1 << 30)) {
        sed_set_error("conv3x3_dgrad_bnred: one sequence's output of %zu elements exceeds the 32-bit row offsets of the MFMA kernels", (size_t)T * F * Cin);
        return -1;
    }
    hipStream_t s = as_stream(stream);
    SedProfScope prof(SED_K_CONV_MFMA_DGRAD, s, 2.0 * 9.0 * C * Cin * (double)B * T * F);
    ConvBnRed br{pooled, gamma, beta, conv_out_below, mean, rstd, 1.f - drop_p, 1.f / (1.f - drop_p), pool_f, pool_t, Fy, Ty,
                 x1, bits, rg_partials, 1.0f / (float)(2 * p.TT + 2)};
    dim3 grid(p.tblocks * p.nft, B, Cin / (32 * p.nct));
    if (x1) {
        const size_t lds = dgrad_rg_lds(p, cin1);
        SED_REQUIRE(bits && rg_partials && (cin1 == 1 || cin1 == 2) && pool_f == 1 && pool_t == 2 && Ty == 2 * T && lds <= 80 * 1024 &&
                    dgrad_rg_tile_ok(p, cin1),
                    "conv3x3_dgrad_bnred_rg: needs 1 or 2 input channels, pool (1,2), an even conv time extent and a tile that leaves two workgroups per CU "
                    "and room for the tap-sum exchange (sed_conv3x3_dgrad_bnred_rg_rows)");
        if (cin1 == 1) {
            SED_TRY(set_lds((conv3x3_mfma_fwd2_k<4, 2, true, 1>), lds));
            conv3x3_mfma_fwd2_k<4, 2, true, 1><<<grid, 256, lds, s>>>(dy, wp_dgrad, nullptr, dx, partials, B, C, F, T, Cin, p.TT, p.FT, p.nft,
                                                                     1.0f / (float)p.FT, 1.0f / (float)(p.FT + 2), br);
        } else {
            SED_TRY(set_lds((conv3x3_mfma_fwd2_k<4, 2, true, 2>), lds));
            conv3x3_mfma_fwd2_k<4, 2, true, 2><<<grid, 256, lds, s>>>(dy, wp_dgrad, nullptr, dx, partials, B, C, F, T, Cin, p.TT, p.FT, p.nft,
                                                                     1.0f / (float)p.FT, 1.0f / (float)(p.FT + 2), br);
        }
    } else {
        SED_TRY(set_lds((conv3x3_mfma_fwd2_k<4, 2, true>), p.lds));
        conv3x3_mfma_fwd2_k<4, 2, true><<<grid, 256, p.lds, s>>>(dy, wp_dgrad, nullptr, dx, partials, B, C, F, T, Cin, p.TT, p.FT, p.nft,
                                                              1.0f / (float)p.FT, 1.0f / (float)(p.FT + 2), br);
    }
    SED_LAUNCH_CHECK("conv3x3_dgrad_bnred");
    return 0;
}

extern "C" int sed_conv3x3_dgrad_bnred(const float* dy, const float* wp_dgrad, float* dx, float* partials, const float* pooled,
                                       const float* gamma, const float* beta, const float* conv_out_below, const float* mean,
                                       const float* rstd, float drop_p, int pool_f, int pool_t, int Fy, int Ty,
                                       int B, int C, int F, int T, int Cin, void* stream) {
    return dgrad_bnred_impl(dy, wp_dgrad, dx, partials, pooled, gamma, beta, conv_out_below, mean, rstd, drop_p, pool_f, pool_t, Fy, Ty,
                            nullptr, 0, nullptr, nullptr, B, C, F, T, Cin, stream);
}

// The same launch when the block below is the recomputed 1-channel first block with pool (1,2): the epilogue also forms that
// block's weight-gradient sums (sum g, R_k per channel and workgroup -> rg_partials [rows][Cin][1 + 9 Cin1], rows =
// sed_conv3x3_dgrad_bnred_rg_rows) from the network input x1 [B][Cin1][F][2T], Cin1 = 1 or 2, and the arg-max bits of its forward, for
// sed_conv1_bwd_wgrad_assemble.  conv_out_below does not exist for a recomputed block: gamma == 0 channels as in the plain entry.
extern "C" int sed_conv3x3_dgrad_bnred_rg_rows(int B, int C, int F, int T, int Cin, int Cin1) {
    if (Cin1 < 1 || Cin1 > 2) return 0;
    ConvPlan p = conv_plan(B, C, F, T, Cin, 0);
    return (p.kind == 1 && p.nct == 4 && dgrad_rg_lds(p, Cin1) <= 80 * 1024 && dgrad_rg_tile_ok(p, Cin1)) ? p.rows : 0;
}
extern "C" int sed_conv3x3_dgrad_bnred_rg(const float* dy, const float* wp_dgrad, float* dx, float* partials, const float* pooled,
                                          const float* gamma, const float* beta, const float* mean, const float* rstd, float drop_p,
                                          const float* x1, int Cin1, const unsigned char* argmax_bits, float* rg_partials,
                                          int B, int C, int F, int T, int Cin, void* stream) {
    SED_REQUIRE(x1 && argmax_bits && rg_partials, "conv3x3_dgrad_bnred_rg: null pointer");
    return dgrad_bnred_impl(dy, wp_dgrad, dx, partials, pooled, gamma, beta, nullptr, mean, rstd, drop_p, 1, 2, F, 2 * T,
                            x1, Cin1, argmax_bits, rg_partials, B, C, F, T, Cin, stream);
}

// ───────────────────────── weight gradient ─────────────────────────
struct WgradPlan {
    int kind;        // 0 small, 1 mfma
    int TT, FT, nft; // block tile: TT time rows x FT mel columns, nft mel tiles (mfma); FT == F on the small path
    int ntiles, ngroups, tblocks;
    int v2;          // mfma, exact fp32: the position-contiguous kernel (2 time rows x 40 or 32 mel columns)
    size_t lds, slab_floats, zrow_floats;
};

// bytes of the two-buffer bf16 hi/lo LDS image of conv3x3_wgrad_bf16x3_k (x pitch 64 B, dy pitch 320 B, 16-position k-steps)
static size_t wgrad_bf16x3_lds(int TT, int FT) {
    const size_t HR = (size_t)(TT + 2) * (FT + 2), MPAD = (size_t)((TT * FT + 15) / 16) * 16;
    if (HR * 8 > 256 * 8 || MPAD * 32 > 256 * 14) return (size_t)1 << 30;      // the loader's item budget (WB_NX / WB_ND)
    return 2 * (HR * 128 + MPAD * 640);
}

static WgradPlan wgrad_plan(int B, int Cin, int F, int T, int Cout, int x_is_nchw, int mode = 0) {
    WgradPlan p{};
    if (B <= 0 || Cin <= 0 || F <= 0 || T <= 0 || Cout <= 0) { p.kind = 0; p.TT = 1; p.FT = 1; p.nft = 1; p.tblocks = 1; return p; }
    p.kind = (!x_is_nchw && Cin % 32 == 0 && Cout % 128 == 0) ? 1 : 0;
    p.FT = F; p.nft = 1;
    if (p.kind == 1) {
        // TT time rows x (even) FT <= WG_FT mel columns with TT*FT <= 2*WG_FT positions, two LDS buffers (the DMA item
        // budget of the kernel: WG_NX / WG_ND); more rows for a narrow mel axis (the mel-pooled topologies)
        p.nft = cdiv(F, WG_FT);
        p.FT = cdiv(F, p.nft);
        p.FT += p.FT & 1;
        p.TT = 2;
        while (p.TT + 2 <= 62 && p.TT + 2 <= T + (T & 1) && (p.TT + 2) * p.FT <= 2 * WG_FT &&
               (p.TT + 4) * (p.FT + 2) * 8 <= 256 * WG_NX)
            p.TT += 2;
        p.lds = (size_t)2 * ((size_t)(p.TT + 2) * (p.FT + 2) * 32 + (size_t)p.TT * p.FT * 128) * sizeof(float);
        // position-contiguous kernel: mel extents that cut into 40- or 32-column tiles (the BASELINE shapes: F = 40, F = 128)
        if (mode == 0 && T >= 2) {
            const int n40 = cdiv(F, 40), n32 = cdiv(F, 32);
            const int ft = (F % 8 == 0 && n40 * 40 == F) ? 40 : ((F % 8 == 0 && n32 * 32 == F) ? 32 : 0);
            if (ft) {
                p.v2 = 1; p.FT = ft; p.nft = F / ft; p.TT = 2;
                const int px = ft + 4;
                p.lds = (size_t)2 * ((size_t)32 * (4 * px + 4) + (size_t)128 * (2 * ft + 4)) * sizeof(float);
            }
        }
        if (mode == 1) {           // the padded dy pitch makes the image larger than the fp32 one: fewer time rows per tile
            p.TT = 1;
            while (p.TT + 1 <= 62 && p.TT + 1 <= T && wgrad_bf16x3_lds(p.TT + 1, p.FT) <= 160 * 1024) p.TT += 1;
            p.lds = wgrad_bf16x3_lds(p.TT, p.FT);
        }
    } else {
        p.TT = 4;
        if (p.TT > T) p.TT = T;
        size_t a = (((size_t)(p.TT + 2) * (F + 2) + 3) & ~(size_t)3) * sizeof(float);
        size_t red = (size_t)256 * 36 * sizeof(float);
        p.lds = a + red;
    }
    p.tblocks = cdiv(T, p.TT);
    p.ntiles = B * p.tblocks * p.nft;
    // mfma: one resident block per CU at Cin=128 (grid = ngroups x Cin/32); small (HBM-bound): enough blocks to fill every CU 4x
    const int maxg = (p.kind == 1) ? 64 : 1024;
    p.ngroups = p.ntiles < maxg ? p.ntiles : maxg;
    p.slab_floats = (size_t)p.ngroups * (p.v2 ? 16 : 9) * Cin * Cout;      // (the Winograd form of the position-contiguous kernel: 16 components)
    // the position-contiguous kernel reads rows outside the image from a zero-filled row behind the slabs
    p.zrow_floats = p.v2 ? (((size_t)(p.FT + 4) * (Cin > Cout ? Cin : Cout) + 64 + 63) / 64) * 64 : 0;      // at the START of the workspace
    return p;
}

extern "C" size_t sed_conv3x3_wgrad_workspace_bytes(int B, int Cin, int F, int T, int Cout) {
    if (B <= 0 || Cin <= 0 || F <= 0 || T <= 0 || Cout <= 0) return 0;
    WgradPlan a = wgrad_plan(B, Cin, F, T, Cout, 0), b = wgrad_plan(B, Cin, F, T, Cout, 1);
    size_t m = a.slab_floats > b.slab_floats ? a.slab_floats : b.slab_floats;
    return (m + a.zrow_floats) * sizeof(float);
}

// small: slabs [group][ci][9][Cout]
__global__ __launch_bounds__(256) void conv3x3_small_wgrad_k(
    const float* __restrict__ x, int x_nchw, const float* __restrict__ dy, float* __restrict__ slabs,
    int B, int Cin, int F, int T, int Cout, int TT, int tblocks, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int F2 = F + 2;
    float* hp = smem;                               // [(TT+2)][F2] one input channel
    float* red = smem + (((TT + 2) * F2 + 3) & ~3); // [nslots][9][Cout]
    const int tid = threadIdx.x;
    const int ncg = Cout >> 2, nslots = 256 / ncg;
    const int cg = tid % ncg, slot = tid / ncg;
    const bool active = slot < nslots;
    for (int ci = 0; ci < Cin; ++ci) {
        f32x4 acc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = (f32x4){0, 0, 0, 0};
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            int b = tile / tblocks, t0 = (tile - b * tblocks) * TT;
            __syncthreads();
            for (int i = tid; i < (TT + 2) * F2; i += 256) {
                int tt, ff;
                if (x_nchw) { tt = i % (TT + 2); ff = i / (TT + 2); } else { ff = i % F2; tt = i / F2; }
                int t = t0 + tt - 1, f = ff - 1;
                float v = 0.f;
                if (t >= 0 && t < T && f >= 0 && f < F)
                    v = x_nchw ? x[(((size_t)b * Cin + ci) * F + f) * T + t] : x[(((size_t)b * T + t) * F + f) * Cin + ci];
                hp[tt * F2 + ff] = v;
            }
            __syncthreads();
            if (active) {
                for (int p = slot; p < TT * F; p += nslots) {
                    int tl = p / F, f = p - tl * F;
                    if (t0 + tl >= T) break;
                    f32x4 d4 = *(const f32x4*)(dy + (((size_t)b * T + t0 + tl) * F + f) * Cout + cg * 4);
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
                            acc[kh * 3 + kw] += hp[(tl + kw) * F2 + f + kh] * d4;
                }
            }
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int k = 0; k < 9; ++k) *(f32x4*)(red + (slot * 9 + k) * Cout + cg * 4) = acc[k];
        }
        __syncthreads();
        for (int i = tid; i < 9 * Cout; i += 256) {
            float a = 0.f;
            for (int s = 0; s < nslots; ++s) a += red[s * 9 * Cout + i];
            slabs[(((size_t)blockIdx.x * Cin + ci) * 9) * Cout + i] = a;
        }
    }
}

// small, Cin <= 4 (the first layer of the multichannel configurations): every input channel in ONE pass over dY
// (the generic kernel above re-reads dY once per input channel), 9*CIN float4 accumulators per thread.
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_small_wgrad_c_k(
    const float* __restrict__ x, int x_nchw, const float* __restrict__ dy, float* __restrict__ slabs,
    int B, int F, int T, int Cout, int TT, int tblocks, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int F2 = F + 2;
    const int hn = (TT + 2) * F2 * CIN;
    float* hp = smem;                               // [(TT+2)][F2][CIN]
    float* red = smem + ((hn + 3) & ~3);            // [nslots][9][Cout]
    const int tid = threadIdx.x;
    const int ncg = Cout >> 2, nslots = 256 / ncg;
    const int cg = tid % ncg, slot = tid / ncg;
    const bool active = slot < nslots;
    f32x4 acc[9 * CIN];
#pragma unroll
    for (int k = 0; k < 9 * CIN; ++k) acc[k] = (f32x4){0, 0, 0, 0};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int b = tile / tblocks, t0 = (tile - b * tblocks) * TT;
        __syncthreads();
        for (int i = tid; i < hn; i += 256) {
            int tt, ff, ci;
            if (x_nchw) { tt = i % (TT + 2); ff = (i / (TT + 2)) % F2; ci = i / ((TT + 2) * F2); }
            else { ci = i % CIN; ff = (i / CIN) % F2; tt = i / (CIN * F2); }
            int t = t0 + tt - 1, f = ff - 1;
            float v = 0.f;
            if (t >= 0 && t < T && f >= 0 && f < F)
                v = x_nchw ? x[(((size_t)b * CIN + ci) * F + f) * T + t] : x[(((size_t)b * T + t) * F + f) * CIN + ci];
            hp[(tt * F2 + ff) * CIN + ci] = v;
        }
        __syncthreads();
        if (active) {
            for (int p = slot; p < TT * F; p += nslots) {
                int tl = p / F, f = p - tl * F;
                if (t0 + tl >= T) break;
                f32x4 d4 = *(const f32x4*)(dy + (((size_t)b * T + t0 + tl) * F + f) * Cout + cg * 4);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const float* q = hp + ((tl + kw) * F2 + f + kh) * CIN;
#pragma unroll
                        for (int ci = 0; ci < CIN; ++ci) acc[(kh * 3 + kw) * CIN + ci] += q[ci] * d4;
                    }
            }
        }
    }
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
        __syncthreads();
        if (active) {
#pragma unroll
            for (int k = 0; k < 9; ++k) *(f32x4*)(red + (slot * 9 + k) * Cout + cg * 4) = acc[k * CIN + ci];
        }
        __syncthreads();
        for (int i = tid; i < 9 * Cout; i += 256) {
            float a = 0.f;
            for (int s = 0; s < nslots; ++s) a += red[s * 9 * Cout + i];
            slabs[(((size_t)blockIdx.x * CIN + ci) * 9) * Cout + i] = a;
        }
    }
}

// small reduce: dw[co][ci][tap] = sum_g slabs[g][ci][tap][co]; block = 32 outputs x 32 group slices
__global__ __launch_bounds__(1024) void conv_wgrad_reduce_small_k(const float* __restrict__ slabs, float* __restrict__ dw,
                                                                   int ngroups, int Cin, int Cout) {
    __shared__ float s1[32][33];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int n = Cin * 9 * Cout;
    const int i = blockIdx.x * 32 + cl;
    float a = 0.f;
    if (i < n)
        for (int g = sl; g < ngroups; g += 32) a += slabs[(size_t)g * n + i];
    s1[sl][cl] = a;
    __syncthreads();
    if (sl == 0 && i < n) {
        float A = 0.f;
        for (int s = 0; s < 32; ++s) A += s1[s][cl];
        int co = i % Cout, tap = (i / Cout) % 9, ci = i / (9 * Cout);
        dw[((size_t)co * Cin + ci) * 9 + tap] = A;
    }
}

// mfma: grid (ngroups, Cin/32, Cout/128); slabs [group][9][Cin][Cout]
// D[ci][co] += X[pos+tap][ci] * dY[pos][co]: M = 32 input channels, N = 4 waves x 32 out channels, K = positions; all nine
// taps share one dY read (9 accumulator tiles per wave); operands of k-step s+1 are read from LDS while the 9 MFMAs of
// step s issue (explicit two-stage register pipeline, one wave per SIMD).
// A block walks tiles of TT time rows x FT mel columns (TT*FT <= 80 positions: 2 x 40 at F = 40) through two LDS buffers; the next tile is brought in by
// global_load_lds_dwordx4 (LDS-DMA: no VGPR staging, no commit pass, one barrier per tile) while the MFMA loop runs on
// the current one.  The register-staged predecessor (4-row tiles, whole next tile prefetched into 112 VGPRs) ran at the
// same 117 TFLOP/s but held 428 VGPRs and 114 KB of LDS per CU, which kept every other kernel off the CU; this one
// holds 251 VGPRs, so the HBM-bound BatchNorm / first-block backward passes on the auxiliary stream really run beside it.
// MT = false: one mel tile (FT >= F); MT = true: mel-tiled (any F).
// The LDS image is lane-linear ([position][32 ci] / [position][128 co]): float4 number i of a tile goes to byte 16*i,
// so wave w's u-th load instruction covers items u*256 + 64*w .. +63 with a wave-uniform LDS base.
typedef __attribute__((address_space(1))) const void* sed_gptr_t;
typedef __attribute__((address_space(3))) void* sed_lptr_t;

template <bool MT>
__global__ __launch_bounds__(256, MT ? 1 : 2) void conv3x3_mfma_wgrad_k(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slabs,
    int B, int Cin, int F, int T, int Cout, int TT, int FT, int nft, int tblocks, int ntiles, unsigned* __restrict__ arrive) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (arrive && threadIdx.x == 0) (void)__hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // "this workgroup is resident" (sed_internal_stream_gate)
    constexpr int NX = WG_NX, ND = WG_ND;
    const int F2 = FT + 2;
    const int HR = (TT + 2) * F2;
    const int MROWS = TT * FT;
    const int XH = HR * 32, BUF = XH + MROWS * 128;
    const float invF = 1.0f / (float)FT, invF2 = 1.0f / (float)F2;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 128;

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;

    // per-thread, tile-invariant: element offset from the tile origin (b, t0, f0) and the halo / tile coordinates
    //   xt = tt | ff << 6 (halo coordinates; ff huge: item past the tile), dt = tl | fl << 6 (fl huge: none)
    int xo[NX], xt[NX], dofs[ND], dt[ND];
#pragma unroll
    for (int u = 0; u < NX; ++u) {
        int i = tid + u * 256, row = i >> 3, q = i & 7;
        int tt = sed_fdiv(row, invF2), ff = row - tt * F2;
        xt[u] = (i < HR * 8) ? (tt | (ff << 6)) : (1 << 24);
        xo[u] = ((tt - 1) * F + (ff - 1)) * Cin + ci0 + q * 4;
    }
#pragma unroll
    for (int u = 0; u < ND; ++u) {
        int i = tid + u * 256, row = i >> 5, q = i & 31;
        int tl = sed_fdiv(row, invF), fl = row - tl * FT;
        dt[u] = (i < MROWS * 32) ? (tl | (fl << 6)) : (1 << 24);
        dofs[u] = (tl * F + fl) * Cout + co0 + q * 4;
    }
    auto issue = [&](int tile, float* buf) {          // global -> LDS (DMA); padding / out-of-range items are zeroed
        int b = tile / (tblocks * nft), rem = tile - b * (tblocks * nft);
        int tb = rem / nft, f0 = MT ? (rem - tb * nft) * FT : 0, t0 = tb * TT;
        const float* xb = x + (((size_t)b * T + t0) * F + f0) * Cin;
        const float* db = dy + (((size_t)b * T + t0) * F + f0) * Cout;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int i = tid + u * 256;
            if (i < HR * 8) {
                int t = t0 + (xt[u] & 63) - 1, f = f0 + (xt[u] >> 6) - 1;
                if ((unsigned)t < (unsigned)T && (unsigned)f < (unsigned)F)
                    __builtin_amdgcn_global_load_lds((sed_gptr_t)(xb + xo[u]), (sed_lptr_t)(buf + (u * 256 + wave * 64) * 4), 16, 0, 0);
                else
                    *(f32x4*)(buf + i * 4) = (f32x4){0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int i = tid + u * 256;
            if (i < MROWS * 32) {
                if (t0 + (dt[u] & 63) < T && f0 + (dt[u] >> 6) < F)
                    __builtin_amdgcn_global_load_lds((sed_gptr_t)(db + dofs[u]), (sed_lptr_t)(buf + XH + (u * 256 + wave * 64) * 4), 16, 0, 0);
                else
                    *(f32x4*)(buf + XH + i * 4) = (f32x4){0, 0, 0, 0};
            }
        }
    };
    auto compute = [&](const float* buf) {
        const float* xh = buf;
        const float* dys = buf + XH;
        const int nfs = FT >> 1;
        for (int tl = 0; tl < TT; ++tl) {
            const float* xrow = xh + (tl * F2 + h) * 32 + r;
            const float* drow = dys + (tl * FT + h) * 128 + wave * 32 + r;
            float a0[9], a1[9], b0, b1;
            auto ld = [&](int fs, float* a, float& bq) {
                const float* xp = xrow + fs * 64;
                bq = drow[fs * 256];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) a[kh * 3 + kw] = xp[(kw * F2 + kh) * 32];
            };
            ld(0, a0, b0);
            int fs = 0;
            for (; fs + 1 < nfs; fs += 2) {
                ld(fs + 1, a1, b1);
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[k], b0, acc[k], 0, 0, 0);
                if (fs + 2 < nfs) ld(fs + 2, a0, b0);
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[k], b1, acc[k], 0, 0, 0);
            }
            if (fs < nfs) {
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[k], b0, acc[k], 0, 0, 0);
            }
        }
    };

    int tile = blockIdx.x, cur = 0;
    if (tile < ntiles) issue(tile, smem);
    __syncthreads();                                 // drains the DMA (vmcnt) and the zero writes
    for (; tile < ntiles; tile += gridDim.x) {
        const int nxt = tile + gridDim.x;
        if (nxt < ntiles) issue(nxt, smem + (cur ^ 1) * BUF);   // that buffer was last read before the previous barrier
        compute(smem + cur * BUF);
        __syncthreads();
        cur ^= 1;
    }
    float* sl = slabs + (size_t)blockIdx.x * 9 * Cin * Cout;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int row = (j & 3) + 8 * (j >> 2) + 4 * h;
            sl[((size_t)k * Cin + ci0 + row) * Cout + co0 + wave * 32 + r] = acc[k][j];
        }
}

// ── weight gradient, position-contiguous LDS image (round 3) ──
// Same product, same grid / slabs / fixed-order reduction as conv3x3_mfma_wgrad_k, different operand path.  With ONE wave per
// SIMD every instruction issued between two MFMAs costs MFMA issue time (round-2 ablation: ten ds_read_b32 per nine MFMAs
// cost 12 %, the per-tile LDS-DMA 8.6 %), so the lever is instructions per MFMA.  K runs over positions, and the k index of an
// MFMA can be assigned to positions freely as long as A and B agree: here a k-chunk is 8 consecutive mel positions of one
// time row, MFMA j (0..3) takes position f0 + j in its k = 0 half (lanes 0-31) and f0 + 4 + j in its k = 1 half.  With the tile
// held TRANSPOSED in LDS (x^T [32 ci][4 halo rows][PX], dy^T [128 co][2 rows][FT], positions contiguous) a lane fetches
//   * its dy operands of all four MFMAs of a chunk with ONE ds_read_b128,
//   * its x operands with TWO ds_read_b128 per time row of the halo: the three mel taps kh = 0..2 of MFMA j are elements
//     j + kh of that 8-float window (register operands, no shuffles, no further reads),
// i.e. 7 LDS reads per 36 MFMAs (9 taps x 4 k-steps) instead of 40, all offsets immediates of the fully unrolled tile body
// (no address arithmetic in the loop).  The transposition happens while staging: global float4 (4 channels of one position) ->
// registers -> four ds_write_b32; the next tile's 13-16 float4 per thread are loaded in batches two k-chunks ahead of their
// commit, so at most two batches (32-40 VGPRs) are in flight and no load is waited for.  Row strides are an odd number of
// 16-byte granules (conflict-free b128 reads across the 32 channel lanes).
template <int FT, bool WINO = false>
__global__ __launch_bounds__(256, 1) void conv3x3_mfma_wgrad2_k(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slabs, const float* __restrict__ zrow,
    int B, int Cin, int F, int T, int Cout, int nft, int tblocks, int ntiles, unsigned* __restrict__ arrive) {
    if (arrive && threadIdx.x == 0) (void)__hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // "this workgroup is resident" (sed_internal_stream_gate)
    constexpr int TT = 2, F2 = FT + 2;
    constexpr int PX = FT + 4;                      // halo row pitch: the second b128 of the k = 1 half reaches column FT + 3
    constexpr int XCI = (TT + 2) * PX + 4;          // floats per input channel
    constexpr int DCO = TT * FT + 4;                // floats per output channel
    static_assert(FT % 8 == 0 && PX % 4 == 0 && ((XCI / 4) & 1) == 1 && ((DCO / 4) & 1) == 1, "pitches: 16-B aligned, odd granule strides");
    constexpr int XBUF = 32 * XCI, BUF = XBUF + 128 * DCO;
    constexpr int HR = (TT + 2) * F2, MROWS = TT * FT;
    constexpr int NCH = WINO ? FT / 8 : TT * (FT / 8);      // k-chunks per tile: 8 positions of a time row; WINO: 4 Winograd tiles (2 x 8 positions)
    // staging batches of the NEXT tile: batch b is loaded before the MFMAs of chunk b and committed before those of chunk
    // NCH - NB + b, i.e. NCH - NB chunks (>= 5 us of MFMA work) later: a commit that has to wait for its load stalls the MFMA
    // pipe (the wave issues in order), so the distance must cover the HBM latency under load with a wide margin
    constexpr int NB = NCH >= 10 ? 4 : (NCH >= 6 ? 3 : 2);
    static_assert(NB >= 1 && 2 * NB <= NCH, "staging schedule");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 128;
    (void)HR; (void)MROWS;

    constexpr int NACC = WINO ? 16 : 9;            // WINO: [nu][channel tile] of this wave's row combination xi = wave
    f32x16 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;

    // Staging of the NEXT tile, organised so that the per-tile work is scalar: wave w stages halo row tt = w of x (GX slots of
    // 8 positions x 8 channel quads) and half of time row tl = w >> 1 of dy (ND slots of 8 positions x 8 channel quads).  A
    // slot's global address is a wave-uniform row pointer (clamped in scalar code) + a per-lane offset computed once; its
    // LDS destination is a per-lane constant; whether the row lies inside the image is wave-uniform (zeros are written
    // instead); only the two slots that hold the mel padding columns need a per-lane select, and only at a mel edge.
    // 8 positions x 8 quads per wave store: bank = 16 (q mod 4) + position -> two-way conflicts at worst.
    constexpr int GX = (F2 + 7) / 8;                // x slots per wave (one halo row)
    constexpr int ND = (FT / 8) * 4 / 2;            // dy slots per wave (half a row: FT/8 position groups x 4 quad groups / 2 waves)
    constexpr int NS = GX + ND;
    const int lp = lane >> 3, lq = lane & 7;
    int xso[GX], xld[GX], dso[ND], dld[ND];
#pragma unroll
    for (int k = 0; k < GX; ++k) {
        int pc = k * 8 + lp;
        pc = pc < F2 ? pc : F2 - 1;                 // slots past the row repeat its last position (same value written twice)
        xso[k] = (pc - 1) * Cin + ci0 + 4 * lq;
        xld[k] = (4 * lq) * XCI + wave * PX + pc;
    }
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        const int j = (wave & 1) * ND + k, pg = j >> 2, qg = j & 3;
        const int pp = pg * 8 + lp, q = qg * 8 + lq;
        dso[k] = pp * Cout + co0 + 4 * q;
        dld[k] = XBUF + (4 * q) * DCO + (wave >> 1) * FT + pp;
    }
    f32x4 st[NS];
    // wave-uniform row pointers of the tile being staged.  A row outside the image (the halo row above the first / below the
    // last time row of a sequence, the second row of a ragged last tile) is read from `zrow`, a zero-filled row in the
    // caller's workspace: no per-element select, no second code path.
    const float* xrow = x;
    const float* drow = dy;
    bool pad_l = false, pad_r = false;              // wave-uniform: the tile touches the left / right end of the mel axis
    auto origin = [&](int tile) {
        const int b = tile / (tblocks * nft), rem = tile - b * (tblocks * nft);
        const int tb = rem / nft;
        const int f0 = (rem - tb * nft) * FT, t0 = tb * TT;
        const int tx = t0 + wave - 1, td = t0 + (wave >> 1);
        pad_l = f0 == 0;
        pad_r = f0 + FT >= F;
        xrow = (unsigned)tx < (unsigned)T ? x + (((size_t)b * T + tx) * F + f0) * Cin : zrow + Cin - ci0;
        drow = td < T ? dy + (((size_t)b * T + td) * F + f0) * Cout : zrow - co0;
    };
    auto load_item = [&](int k) {
        if (k < GX) {
            int off = xso[k];
            // the padding columns (f = -1 at the left mel edge, f = F at the right one) are loaded from their valid neighbour
            if (k == 0) off = (pad_l && lp == 0) ? off + Cin : off;
            if (k * 8 + 7 >= F2 - 1) off = (pad_r && k * 8 + lp >= F2 - 1) ? (F2 - 3) * Cin + ci0 + 4 * lq : off;
            st[k] = *(const f32x4*)(xrow + off);
        } else {
            st[k] = *(const f32x4*)(drow + dso[k - GX]);
        }
    };
    auto commit_item = [&](int k, float* buf) {
        float* d = buf + (k < GX ? xld[k] : dld[k - GX]);
        constexpr f32x4 Z = {0, 0, 0, 0};
        f32x4 v = st[k];
        if (k < GX) {
            if (k == 0 && pad_l && lp == 0) v = Z;                                       // the two padding columns of the mel axis
            if (k * 8 + 7 >= F2 - 1 && pad_r && k * 8 + lp >= F2 - 1) v = Z;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e * (k < GX ? XCI : DCO)] = v[e];
    };

    // operand fetch of one k-chunk (time row tl, mel columns fc*8 .. fc*8+7)
    // (the window's last two floats are never used: a b64 for the second half keeps hipcc from overlapping the dead registers of
    // one b128 with the destination of the next, which made it serialise the reads with lgkmcnt waits)
    struct Ops { f32x4 xa[3]; f32x2 xb[3]; f32x4 b; };
    auto fetch_ops = [&](const float* buf, int tl, int fc, Ops& o) {
        const float* xp = buf + r * XCI + tl * PX + fc * 8 + 4 * h;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            o.xa[kw] = *(const f32x4*)(xp + kw * PX);
            o.xb[kw] = *(const f32x2*)(xp + kw * PX + 4);
        }
        o.b = *(const f32x4*)(buf + XBUF + (wave * 32 + r) * DCO + tl * FT + fc * 8 + 4 * h);
    };
    auto mfma_chunk = [&](const Ops& o) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int e = j + kh;
                    const float a = e < 4 ? o.xa[kw][e] : o.xb[kw][e - 4];
                    acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, o.b[j], acc[kh * 3 + kw], 0, 0, 0);
                }
    };

    // ── WINO: the weight gradient in the Winograd domain (F(2x2,3x3), see wino.hip) ──
    // dU[xi,nu][ci][co] = sum over 2x2 output tiles of (B^T d B)[xi,nu][ci] (A dY A^T)[xi,nu][co], 16 products per tile and channel pair
    // instead of 36; dW = G^T dU G is applied by the slab reduction.  A tile row = this kernel's tile (2 time rows, 4 halo rows); a
    // k-chunk = 4 Winograd tiles = 8 mel columns, MFMA j takes tile 2h + j of the chunk.  Wave xi forms its own row combination of the
    // patch (d[ra] + sg d[rb]) and of the gradient rows (y0, y0 + y1, y0 - y1, y1) and holds the four column combinations nu for all
    // four 32-channel tiles of the workgroup's 128 output channels: 16 accumulators.  The signs of A's last row / column (-y1) are
    // left to the reduction: the products are linear in them.  Which rows a wave combines is DATA (row offsets and a coefficient in
    // scalar registers), not control flow: every wave runs the same straight-line chunk (y[ya] + ys y[yb], ys = 0 for the plain rows).
    const int w_ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1), w_rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float w_sg = wave == 1 ? 1.f : -1.f;
    const int w_ya = wave == 3 ? FT : 0, w_yb = wave == 0 ? 0 : FT;
    const float w_ys = wave == 1 ? 1.f : (wave == 2 ? -1.f : 0.f);
    struct WOps { f32x4 a[2]; f32x2 a2[2]; f32x4 b[4][2]; };
    auto wfetch = [&](const float* buf, int kc, WOps& o) {
        const float* xp = buf + r * XCI + kc * 8 + 4 * h;
        o.a[0] = *(const f32x4*)(xp + w_ra * PX);
        o.a2[0] = *(const f32x2*)(xp + w_ra * PX + 4);
        o.a[1] = *(const f32x4*)(xp + w_rb * PX);
        o.a2[1] = *(const f32x2*)(xp + w_rb * PX + 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float* dp = buf + XBUF + (nt * 32 + r) * DCO + kc * 8 + 4 * h;
            o.b[nt][0] = *(const f32x4*)(dp + w_ya);
            o.b[nt][1] = *(const f32x4*)(dp + w_yb);
        }
    };
    // the transformed operands of one chunk: v[j][nu] (A, this wave's xi) and z[nt][j][nu] (B)
    struct WT { float v[2][4]; float z[4][2][4]; };
    auto wtransform = [&](const WOps& o, WT& t) {
        float u[6];
#pragma unroll
        for (int c = 0; c < 4; ++c) u[c] = o.a[0][c] + w_sg * o.a[1][c];
        u[4] = o.a2[0][0] + w_sg * o.a2[1][0];
        u[5] = o.a2[0][1] + w_sg * o.a2[1][1];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            t.v[j][0] = u[2 * j] - u[2 * j + 2];
            t.v[j][1] = u[2 * j + 1] + u[2 * j + 2];
            t.v[j][2] = u[2 * j + 2] - u[2 * j + 1];
            t.v[j][3] = u[2 * j + 1] - u[2 * j + 3];
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            // this wave's combination of the two gradient rows: columns (c0, c1) of tile 2h, (c0, c1) of tile 2h + 1
            const f32x4 yr = o.b[nt][0] + w_ys * o.b[nt][1];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float c0 = yr[2 * j], c1 = yr[2 * j + 1];
                t.z[nt][j][0] = c0; t.z[nt][j][1] = c0 + c1; t.z[nt][j][2] = c0 - c1; t.z[nt][j][3] = c1;
            }
        }
    };
    auto wmfma = [&](const WT& t) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nu = 0; nu < 4; ++nu)
                    acc[nu * 4 + nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(t.v[j][nu], t.z[nt][j][nu], acc[nu * 4 + nt], 0, 0, 0);
    };

    constexpr int BS = (NS + NB - 1) / NB;
    // a group walks a CONTIGUOUS run of tiles (consecutive time rows of one sequence): the two halo rows a tile shares with its
    // predecessor were read by this very CU one tile earlier and come from L2 instead of HBM (a strided walk handed neighbouring
    // tiles to workgroups on different XCDs, i.e. different L2s: PMC 1.5x the algorithmic traffic)
    const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    int tile = blockIdx.x * per, cur = 0;
    const int tend = tile + per < ntiles ? tile + per : ntiles;
    if (tile < tend) {                              // first tile: staged in one go
        origin(tile);
#pragma unroll
        for (int k = 0; k < NS; ++k) load_item(k);
#pragma unroll
        for (int k = 0; k < NS; ++k) commit_item(k, smem);
    }
    __syncthreads();
    for (; tile < tend; ++tile) {
        const int nxt = tile + 1;
        origin(nxt < tend ? nxt : tile);            // the last tile re-stages itself (nobody reads that image): no branch
        const float* buf = smem + cur * BUF;
        float* nbuf = smem + (cur ^ 1) * BUF;       // last read before the previous barrier
        if (WINO) {
            // operands are read one chunk ahead of their transform and transformed one chunk ahead of their MFMAs: a chunk's 32 MFMAs
            // start with everything in registers (transforming at the head of the chunk left the MFMA pipe idle for ~40 vector
            // instructions, five times per tile)
            WOps w;
            WT ta, tb;
            wfetch(buf, 0, w);
            wtransform(w, ta);
            if (NCH > 1) wfetch(buf, 1, w);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                WT& tc = (c & 1) ? tb : ta;
                WT& tn = (c & 1) ? ta : tb;
                __builtin_amdgcn_sched_barrier(0);
                if (c + 1 < NCH) wtransform(w, tn);
                if (c + 2 < NCH) wfetch(buf, c + 2, w);
                if (c < NB) {
#pragma unroll
                    for (int k = c * BS; k < (c + 1) * BS && k < NS; ++k) load_item(k);
                }
                if (c >= NCH - NB) {
#pragma unroll
                    for (int k = (c - (NCH - NB)) * BS; k < (c - (NCH - NB) + 1) * BS && k < NS; ++k) commit_item(k, nbuf);
                }
                wmfma(tc);
#pragma unroll
                for (int g = 0; g < 32; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x322, 3, 0);      // up to three of: VALU, VMEM read, DS read, DS write
                }
            }
        } else {
        // one tile: NCH chunks of 36 MFMAs; the staging of the next tile rides along (loads before chunks 0..NB-1, commits
        // before chunks NCH-NB..NCH-1)
        Ops o0, o1;
        fetch_ops(buf, 0, 0, o0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            Ops& oc = (c & 1) ? o1 : o0;
            Ops& on = (c & 1) ? o0 : o1;
            // everything between the two scheduling barriers is issued BEFORE this chunk's 36 MFMAs (which depend on none of
            // it): the next chunk's operands are a whole chunk ahead of their use
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < NCH) fetch_ops(buf, (c + 1) / (FT / 8), (c + 1) % (FT / 8), on);
            if (c < NB) {
#pragma unroll
                for (int k = c * BS; k < (c + 1) * BS && k < NS; ++k) load_item(k);
            }
            if (c >= NCH - NB) {
#pragma unroll
                for (int k = (c - (NCH - NB)) * BS; k < (c - (NCH - NB) + 1) * BS && k < NS; ++k) commit_item(k, nbuf);
            }
            mfma_chunk(oc);
            // Issue order inside the chunk: ONE non-MFMA instruction per MFMA gap.  A global_load_dwordx4 holds the wave's
            // issue for ~60 cycles and an LDS instruction for 8-16 (one wave per SIMD issues in order), each fits in the 64-cycle
            // shadow of the MFMA in front of it, but a burst of them in front of the chunk does not (measured: the four loads
            // of a batch issued back to back cost 4 % of the kernel, the commits 5 %).
#pragma unroll
            for (int g = 0; g < 34; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x322, 1, 0);      // one of: VALU, VMEM read, DS read, DS write
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        cur ^= 1;
    }
    if (WINO) {                                       // slabs [group][16 components][Cin][Cout]
        float* sl = slabs + (size_t)blockIdx.x * 16 * Cin * Cout;
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
                    sl[((size_t)(wave * 4 + nu) * Cin + ci0 + row) * Cout + co0 + nt * 32 + r] = acc[nu * 4 + nt][j];
                }
        return;
    }
    float* sl = slabs + (size_t)blockIdx.x * 9 * Cin * Cout;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int row = (j & 3) + 8 * (j >> 2) + 4 * h;
            sl[((size_t)k * Cin + ci0 + row) * Cout + co0 + wave * 32 + r] = acc[k][j];
        }
}

// ── weight gradient on the 3-term bf16 split (EXPERIMENT, mode 1 of sed_conv3x3_wgrad_ex) ──
// dW[tap][ci][co] = sum_pos x[pos+tap][ci] dy[pos][co]: the contraction index is the POSITION, while both operands are
// stored position-major (channels contiguous), so the k-contiguous fragments of v_mfma_f32_32x32x16_bf16 are gathered with
// ds_read_b64_tr_b16: per 16-lane group a 4-position x 16-channel block comes back transposed (lane = channel, element =
// position); two reads give a lane its 8 k values.  LDS image per tile: x halo [position][32 ci] bf16, hi and lo arrays with
// a 64-byte pitch, dy [position][128 co] bf16, hi and lo arrays with a 320-byte pitch (both conflict-free for the
// transposed reads: the 8 row/group chunks of a 32-lane half land on the 8 distinct 32-byte slots of the bank period).
// Eight waves: 0-3 compute (wave w: co 32w..32w+31, nine tap accumulators), 4-7 load the next tile (global fp32 ->
// hi/lo split -> LDS) while the others run the MFMA loop.  Same tiles, slabs and fixed-order slab reduction as the fp32 kernel.
typedef __attribute__((address_space(3))) bf16x4* sed_lds_bf16x4_t;
__device__ __forceinline__ bf16x8 lds_tr8(const __bf16* p0, const __bf16* p1) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((sed_lds_bf16x4_t)(p0));
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((sed_lds_bf16x4_t)(p1));
    return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

#define WB_DYP 160      // dy row pitch in bf16 (320 B)
#define WB_NX 8         // loader items (float4) per lane of the halo tile:  HR*8    <= 256*WB_NX
#define WB_ND 14        // loader items (float4) per lane of the dY tile:    MPAD*32 <= 256*WB_ND
template <bool MT>
__global__ __launch_bounds__(512, 1) void conv3x3_wgrad_bf16x3_k(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slabs,
    int B, int Cin, int F, int T, int Cout, int TT, int FT, int nft, int tblocks, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int F2 = FT + 2;
    const int HR = (TT + 2) * F2;
    const int MROWS = TT * FT;
    const int KS = (MROWS + 15) >> 4, MPAD = KS * 16;
    // per buffer (bf16 units): x hi [HR][32], x lo [HR][32], dy hi [MPAD][160], dy lo [MPAD][160]
    const int XP = HR * 32, DP = MPAD * WB_DYP, BUFH = 2 * XP + 2 * DP;
    const float invF = 1.0f / (float)FT, invF2 = 1.0f / (float)F2;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 128;
    __bf16* lds = reinterpret_cast<__bf16*>(smem);

    if (wave >= 4) {
        // ── loaders: global fp32 -> registers (one tile ahead of the LDS image, two ahead of the MFMA loop) -> (hi, lo)
        // bf16 -> LDS.  All loads of a tile are issued back to back and only waited for one tile later. ──
        constexpr int NX = WB_NX, ND = WB_ND;
        const int lt = tid - 256;
        int xo[NX], xt[NX], dofs[ND], dt[ND];            // tile-invariant item coordinates, as in the fp32 kernel
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            int i = lt + u * 256, row = i >> 3, qq = i & 7;
            int tt = sed_fdiv(row, invF2), ff = row - tt * F2;
            xt[u] = (i < HR * 8) ? (tt | (ff << 8)) : (1 << 24);
            xo[u] = ((tt - 1) * F + (ff - 1)) * Cin + ci0 + qq * 4;
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            int i = lt + u * 256, row = i >> 5, qq = i & 31;
            int tl = sed_fdiv(row, invF), fl = row - tl * FT;
            dt[u] = (row < MROWS) ? (tl | (fl << 8)) : (1 << 24);
            dofs[u] = (tl * F + fl) * Cout + co0 + qq * 4;
        }
        f32x4 rx[NX], rd[ND];
        auto fetch = [&](int tile) {
            int b = tile / (tblocks * nft), rem = tile - b * (tblocks * nft);
            int tb = rem / nft, f0 = MT ? (rem - tb * nft) * FT : 0, t0 = tb * TT;
            const float* xb = x + (((size_t)b * T + t0) * F + f0) * Cin;
            const float* db = dy + (((size_t)b * T + t0) * F + f0) * Cout;
#pragma unroll
            for (int u = 0; u < NX; ++u) {
                const int t = t0 + (xt[u] & 255) - 1, f = f0 + (xt[u] >> 8) - 1;
                rx[u] = (f32x4){0, 0, 0, 0};
                if ((unsigned)t < (unsigned)T && (unsigned)f < (unsigned)F) rx[u] = *(const f32x4*)(xb + xo[u]);
            }
#pragma unroll
            for (int u = 0; u < ND; ++u) {
                rd[u] = (f32x4){0, 0, 0, 0};
                if (t0 + (dt[u] & 255) < T && f0 + (dt[u] >> 8) < F) rd[u] = *(const f32x4*)(db + dofs[u]);
            }
        };
        auto split_store = [&](const f32x4 v, __bf16* hi_at, __bf16* lo_at) {
            bf16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) { hi[e] = (__bf16)v[e]; lo[e] = (__bf16)(v[e] - (float)hi[e]); }
            *(bf16x4*)hi_at = hi;
            *(bf16x4*)lo_at = lo;
        };
        auto commit = [&](__bf16* buf) {
#pragma unroll
            for (int u = 0; u < NX; ++u) {
                const int i = lt + u * 256;
                if (i < HR * 8) split_store(rx[u], buf + i * 4, buf + XP + i * 4);
            }
#pragma unroll
            for (int u = 0; u < ND; ++u) {
                const int i = lt + u * 256, at = (i >> 5) * WB_DYP + (i & 31) * 4;
                if (i < MPAD * 32) split_store(rd[u], buf + 2 * XP + at, buf + 2 * XP + DP + at);
            }
        };
        int tile = blockIdx.x, cur = 0;
        if (tile < ntiles) { fetch(tile); commit(lds); }
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
        __syncthreads();
        for (; tile < ntiles; tile += gridDim.x) {
            const int nxt = tile + gridDim.x, nn = nxt + gridDim.x;
            if (nxt < ntiles) commit(lds + (cur ^ 1) * BUFH);       // that buffer was last read before the previous barrier
            if (nn < ntiles) fetch(nn);                             // in flight across the barrier (plain loads: no vmcnt drain)
            __syncthreads();
            cur ^= 1;
        }
        return;
    }

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;

    // transposed-read lane roles: 16-lane group G = lane >> 4 covers channels 16*(G&1)..+15 and k half h = G >> 1;
    // lane 4q+p of a group addresses block row q (a position), channels 4p..4p+3
    const int G = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int h = G >> 1, cb = 16 * (G & 1) + 4 * pp;
    auto compute = [&](const __bf16* buf) {
        const __bf16* xh = buf;
        const __bf16* xl = buf + XP;
        const __bf16* dh = buf + 2 * XP + wave * 32 + cb;
        const __bf16* dl = dh + DP;
        for (int ks = 0; ks < KS; ++ks) {
            // the two positions this lane addresses in this k-step (t = 0, 1): m = ks*16 + 8h + 4t + q
            int m0 = ks * 16 + 8 * h + q, m1 = m0 + 4;
            const bf16x8 bhi = lds_tr8(dh + m0 * WB_DYP, dh + m1 * WB_DYP);
            const bf16x8 blo = lds_tr8(dl + m0 * WB_DYP, dl + m1 * WB_DYP);
            if (m0 >= MROWS) m0 = MROWS - 1;                  // padded positions carry dy = 0; keep the x address in range
            if (m1 >= MROWS) m1 = MROWS - 1;
            const int tl0 = sed_fdiv(m0, invF), tl1 = sed_fdiv(m1, invF);
            const int hp0 = (tl0 * F2 + (m0 - tl0 * FT)) * 32 + cb, hp1 = (tl1 * F2 + (m1 - tl1 * FT)) * 32 + cb;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int sh = (kw * F2 + kh) * 32;
                    const bf16x8 ahi = lds_tr8(xh + hp0 + sh, xh + hp1 + sh);
                    const bf16x8 alo = lds_tr8(xl + hp0 + sh, xl + hp1 + sh);
                    const int k = kh * 3 + kw;
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, acc[k], 0, 0, 0);
                }
        }
    };

    int tile = blockIdx.x, cur = 0;
    __syncthreads();                                     // the first tile has landed
    for (; tile < ntiles; tile += gridDim.x) {
        compute(lds + cur * BUFH);
        __syncthreads();
        cur ^= 1;
    }
    const int r = lane & 31, hh = lane >> 5;
    float* sl = slabs + (size_t)blockIdx.x * 9 * Cin * Cout;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int row = (j & 3) + 8 * (j >> 2) + 4 * hh;
            sl[((size_t)k * Cin + ci0 + row) * Cout + co0 + wave * 32 + r] = acc[k][j];
        }
}

// mfma reduce: dw[co][ci][tap] = sum_g slabs[g][tap][ci][co]
__global__ void conv_wgrad_reduce_mfma_k(const float* __restrict__ slabs, float* __restrict__ dw,
                                         int ngroups, int Cin, int Cout) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int n = Cin * 9 * Cout;
    if (i >= n) return;
    int co = i % Cout, ci = (i / Cout) % Cin, tap = i / (Cin * Cout);
    float a = 0.f;
#pragma unroll 8
    for (int g = 0; g < ngroups; ++g) a += slabs[(size_t)g * n + i];
    dw[((size_t)co * Cin + ci) * 9 + tap] = a;
}

// Winograd slabs: dU[xi][nu] = s_xi s_nu sum_g slabs[g][xi*4+nu][ci][co] (s_3 = -1: the sign of A's last row, see the kernel), then
// dW = G^T dU G with G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]]; the first index of dU runs along time (kw), the second along mel (kh)
__global__ __launch_bounds__(1024) void conv_wgrad_reduce_wino_k(const float* __restrict__ slabs, float* __restrict__ dw, int ngroups, int Cin, int Cout) {
    // a workgroup: 64 consecutive (ci, co) pairs x 16 components, wave k = component k: every load instruction is one coalesced
    // 256-byte piece, and a thread keeps 8 of them in flight (a one-load-at-a-time chain per thread ran this pass at 1 TB/s);
    // the components meet in LDS and the first wave applies the two 4 -> 3 transforms
    __shared__ float U[16][64];
    const int n = Cin * Cout;
    const int il = threadIdx.x & 63, k = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    if (i < n) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* sp = slabs + (size_t)k * n + i;
        int g = 0;
        for (; g + 8 <= ngroups; g += 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += sp[(size_t)(g + e) * 16 * n];
        }
        for (; g < ngroups; ++g) a[0] += sp[(size_t)g * 16 * n];
        const float t = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        U[k][il] = (((k >> 2) == 3) != ((k & 3) == 3)) ? -t : t;
    }
    __syncthreads();
    if (threadIdx.x >= 64 || i >= n) return;
    const int co = i % Cout, ci = i / Cout;
    float t[3][4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
        const float u0 = U[nu][il], u1 = U[4 + nu][il], u2 = U[8 + nu][il], u3 = U[12 + nu][il];
        t[0][nu] = u0 + 0.5f * (u1 + u2);
        t[1][nu] = 0.5f * (u1 - u2);
        t[2][nu] = u3 + 0.5f * (u1 + u2);
    }
    float* o = dw + ((size_t)co * Cin + ci) * 9;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        o[0 * 3 + kw] = t[kw][0] + 0.5f * (t[kw][1] + t[kw][2]);
        o[1 * 3 + kw] = 0.5f * (t[kw][1] - t[kw][2]);
        o[2 * 3 + kw] = t[kw][3] + 0.5f * (t[kw][1] + t[kw][2]);
    }
}

extern "C" size_t sed_conv3x3_wgrad_zero_row_bytes(int B, int Cin, int F, int T, int Cout) {
    if (B <= 0 || Cin <= 0 || F <= 0 || T <= 0 || Cout <= 0 || Cout % 4 != 0) return 0;
    return wgrad_plan(B, Cin, F, T, Cout, 0, 0).zrow_floats * sizeof(float);
}

extern "C" int sed_conv3x3_wgrad(const float* x, int x_is_nchw, const float* dy, float* dw, void* workspace,
                                 int B, int Cin, int F, int T, int Cout, void* stream) {
    return sed_conv3x3_wgrad_ex(x, x_is_nchw, dy, dw, workspace, B, Cin, F, T, Cout, 0, stream);
}

extern "C" int sed_conv3x3_wgrad_ex(const float* x, int x_is_nchw, const float* dy, float* dw, void* workspace,
                                    int B, int Cin, int F, int T, int Cout, int mode, void* stream) {
    return sed_internal_conv3x3_wgrad(x, x_is_nchw, dy, dw, workspace, B, Cin, F, T, Cout, mode, nullptr, stream);
}

// workgroups of the exact-fp32 MFMA weight gradient of this shape (they are persistent: one per CU for the kernel's lifetime,
// and each announces itself on `arrive`); 0 for the shapes / modes whose kernel does not announce
int sed_internal_conv3x3_wgrad_workgroups(int B, int Cin, int F, int T, int Cout, int x_is_nchw, int mode) {
    if (B <= 0 || Cin <= 0 || F <= 0 || T <= 0 || Cout <= 0 || Cout % 4 != 0) return 0;
    mode &= ~(SED_WGRAD_ZERO_ROW_CLEAN | SED_WGRAD_DIRECT);
    if (mode != 0) return 0;
    const WgradPlan p = wgrad_plan(B, Cin, F, T, Cout, x_is_nchw, mode);
    return p.kind == 1 ? p.ngroups * (Cin / 32) * (Cout / 128) : 0;
}

int sed_internal_conv3x3_wgrad(const float* x, int x_is_nchw, const float* dy, float* dw, void* workspace,
                               int B, int Cin, int F, int T, int Cout, int mode, unsigned* arrive, void* stream) {
    SED_REQUIRE(x && dy && dw && workspace, "conv3x3_wgrad: null pointer");
    const bool zero_row_clean = (mode & SED_WGRAD_ZERO_ROW_CLEAN) != 0;      // the caller cleared sed_conv3x3_wgrad_zero_row_bytes()
    const bool direct = (mode & SED_WGRAD_DIRECT) != 0;                      // the 36-product kernel instead of the Winograd form
    mode &= ~(SED_WGRAD_ZERO_ROW_CLEAN | SED_WGRAD_DIRECT);
    SED_REQUIRE(mode == 0 || mode == 1, "conv3x3_wgrad: unknown mode %d", mode);
    SED_REQUIRE(Cout % 4 == 0, "conv3x3_wgrad: Cout must be a multiple of 4 (got %d)", Cout);
    WgradPlan p = wgrad_plan(B, Cin, F, T, Cout, x_is_nchw, mode);
    hipStream_t s = as_stream(stream);
    float* slabs = (float*)workspace + p.zrow_floats;      // [zero row (exact-fp32 position-contiguous kernel only)][slabs]
    int n = Cin * 9 * Cout;
    const double npos = (double)B * T * F;
    SedProfScope prof(p.kind == 1 ? SED_K_CONV_MFMA_WGRAD : SED_K_CONV_SMALL_WGRAD, s,
                      p.kind == 1 ? 2.0 * 9.0 * Cin * Cout * npos : 4.0 * npos * (Cin + Cout));
    if (p.kind == 1 && mode == 1) {
        dim3 grid(p.ngroups, Cin / 32, Cout / 128);
        const size_t lds = p.lds;
        SED_REQUIRE(lds <= 160 * 1024, "conv3x3_wgrad (bf16x3): tile %dx%d needs %zu B of LDS", p.TT, p.FT, lds);
        if (p.nft == 1) {
            SED_TRY(set_lds(conv3x3_wgrad_bf16x3_k<false>, lds));
            conv3x3_wgrad_bf16x3_k<false><<<grid, 512, lds, s>>>(x, dy, slabs, B, Cin, F, T, Cout, p.TT, p.FT, p.nft, p.tblocks, p.ntiles);
        } else {
            SED_TRY(set_lds(conv3x3_wgrad_bf16x3_k<true>, lds));
            conv3x3_wgrad_bf16x3_k<true><<<grid, 512, lds, s>>>(x, dy, slabs, B, Cin, F, T, Cout, p.TT, p.FT, p.nft, p.tblocks, p.ntiles);
        }
        SED_LAUNCH_CHECK("conv3x3_wgrad_bf16x3");
        conv_wgrad_reduce_mfma_k<<<cdiv(n, 256), 256, 0, s>>>(slabs, dw, p.ngroups, Cin, Cout);
    } else if (p.kind == 1 && p.v2) {
        dim3 grid(p.ngroups, Cin / 32, Cout / 128);
        float* zrow = (float*)workspace;
        if (!zero_row_clean) {
            hipError_t e = hipMemsetAsync(zrow, 0, p.zrow_floats * sizeof(float), s);
            if (e != hipSuccess) { sed_set_error("conv3x3_wgrad: hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
        }
        if (!direct) {
            if (p.FT == 40) {
                SED_TRY(set_lds((conv3x3_mfma_wgrad2_k<40, true>), p.lds));
                conv3x3_mfma_wgrad2_k<40, true><<<grid, 256, p.lds, s>>>(x, dy, slabs, zrow, B, Cin, F, T, Cout, p.nft, p.tblocks, p.ntiles, arrive);
            } else {
                SED_TRY(set_lds((conv3x3_mfma_wgrad2_k<32, true>), p.lds));
                conv3x3_mfma_wgrad2_k<32, true><<<grid, 256, p.lds, s>>>(x, dy, slabs, zrow, B, Cin, F, T, Cout, p.nft, p.tblocks, p.ntiles, arrive);
            }
            SED_LAUNCH_CHECK("conv3x3_mfma_wgrad2 (Winograd)");
            conv_wgrad_reduce_wino_k<<<cdiv(Cin * Cout, 64), 1024, 0, s>>>(slabs, dw, p.ngroups, Cin, Cout);
        } else {
            if (p.FT == 40) {
                SED_TRY(set_lds((conv3x3_mfma_wgrad2_k<40>), p.lds));
                conv3x3_mfma_wgrad2_k<40><<<grid, 256, p.lds, s>>>(x, dy, slabs, zrow, B, Cin, F, T, Cout, p.nft, p.tblocks, p.ntiles, arrive);
            } else {
                SED_TRY(set_lds((conv3x3_mfma_wgrad2_k<32>), p.lds));
                conv3x3_mfma_wgrad2_k<32><<<grid, 256, p.lds, s>>>(x, dy, slabs, zrow, B, Cin, F, T, Cout, p.nft, p.tblocks, p.ntiles, arrive);
            }
            SED_LAUNCH_CHECK("conv3x3_mfma_wgrad2");
            conv_wgrad_reduce_mfma_k<<<cdiv(n, 256), 256, 0, s>>>(slabs, dw, p.ngroups, Cin, Cout);
        }
    } else if (p.kind == 1) {
        dim3 grid(p.ngroups, Cin / 32, Cout / 128);
        if (p.nft == 1) {
            SED_TRY(set_lds(conv3x3_mfma_wgrad_k<false>, p.lds));
            conv3x3_mfma_wgrad_k<false><<<grid, 256, p.lds, s>>>(x, dy, slabs, B, Cin, F, T, Cout, p.TT, p.FT, p.nft, p.tblocks, p.ntiles, arrive);
        } else {
            SED_TRY(set_lds(conv3x3_mfma_wgrad_k<true>, p.lds));
            conv3x3_mfma_wgrad_k<true><<<grid, 256, p.lds, s>>>(x, dy, slabs, B, Cin, F, T, Cout, p.TT, p.FT, p.nft, p.tblocks, p.ntiles, arrive);
        }
        SED_LAUNCH_CHECK("conv3x3_mfma_wgrad");
        conv_wgrad_reduce_mfma_k<<<cdiv(n, 256), 256, 0, s>>>(slabs, dw, p.ngroups, Cin, Cout);
    } else {
        SED_REQUIRE(256 % (Cout / 4) == 0 || Cout / 4 <= 256, "conv3x3_wgrad: unsupported Cout=%d", Cout);
        const size_t lds_c = ((((size_t)(p.TT + 2) * (F + 2) * Cin + 3) & ~(size_t)3) + (size_t)256 * 36) * sizeof(float);
        if (Cin == 2 && lds_c <= 150 * 1024) {
            SED_TRY(set_lds(conv3x3_small_wgrad_c_k<2>, lds_c));
            conv3x3_small_wgrad_c_k<2><<<p.ngroups, 256, lds_c, s>>>(x, x_is_nchw, dy, slabs, B, F, T, Cout, p.TT, p.tblocks, p.ntiles);
        } else if (Cin == 3 && lds_c <= 150 * 1024) {
            SED_TRY(set_lds(conv3x3_small_wgrad_c_k<3>, lds_c));
            conv3x3_small_wgrad_c_k<3><<<p.ngroups, 256, lds_c, s>>>(x, x_is_nchw, dy, slabs, B, F, T, Cout, p.TT, p.tblocks, p.ntiles);
        } else if (Cin == 4 && lds_c <= 150 * 1024) {
            SED_TRY(set_lds(conv3x3_small_wgrad_c_k<4>, lds_c));
            conv3x3_small_wgrad_c_k<4><<<p.ngroups, 256, lds_c, s>>>(x, x_is_nchw, dy, slabs, B, F, T, Cout, p.TT, p.tblocks, p.ntiles);
        } else {
            SED_TRY(set_lds(conv3x3_small_wgrad_k, p.lds));
            conv3x3_small_wgrad_k<<<p.ngroups, 256, p.lds, s>>>(x, x_is_nchw, dy, slabs, B, Cin, F, T, Cout, p.TT, p.tblocks, p.ntiles);
        }
        SED_LAUNCH_CHECK("conv3x3_small_wgrad");
        conv_wgrad_reduce_small_k<<<cdiv(n, 32), 1024, 0, s>>>(slabs, dw, p.ngroups, Cin, Cout);
    }
    SED_LAUNCH_CHECK("conv3x3_wgrad_reduce");
    return 0;
}
