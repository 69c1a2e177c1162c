// gru.hip — the serial part of nn.GRU (reference sed.py:101-102,111; crnn_lightning.py:61-62,71).
//
// The input projections x W_ih^T + b_ih of all timesteps are one dense GEMM (gemm.hip).  What is left is
// the recurrence: T' strictly serial steps of gh = h W_hh^T + b_hh followed by the gate math.  Per-step
// kernel launches would cost more than the arithmetic, so each workgroup runs the whole time loop of
// one (batch tile, direction) in-kernel: W_hh stays resident in registers (one gate row per thread,
// H <= 128) or is streamed from L2 (larger H), h lives in LDS.  fp32 MFMA runs at the fp32 VALU rate on
// gfx950, so the per-step matvec is plain v_fma with a small batch tile (more workgroups in flight)
// rather than a 32-wide MFMA tile.
//
// PyTorch gate convention (r,z,n):  r = s(gi_r+gh_r)  z = s(gi_z+gh_z)  n = tanh(gi_n + r*gh_n)
//                                   h' = (1-z)*n + z*h,  h0 = 0.
#include "common.h"

// batch rows per workgroup (1, 2 or 4): the smallest tile that still gives every CU a workgroup (2*ceil(B/bt) >= 256),
// since a smaller tile shortens the serial step (measured at H=128, B=128: tile 2 -> 1 = 2.45 -> 1.6 us per step);
// at most 2 for H >= 128 (register budget of the backward kernel)
static inline int gru_bt(int H, int B) {
    int bt = B >= 512 ? 4 : (B >= 256 ? 2 : 1);
    if (H >= 128 && bt > 2) bt = 2;
    return bt;
}

__global__ void gru_pack_whh_t_k(const float* __restrict__ w0, const float* __restrict__ w1,
                                 float* __restrict__ wt, int H) {
    // wt[dir][k][g] = whh[dir][g][k]
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int n = 3 * H * H;
    if (i >= 2 * n) return;
    int dir = i / n, e = i - dir * n;
    int g = e / H, k = e - g * H;
    const float* w = dir ? w1 : w0;
    wt[(size_t)dir * n + (size_t)k * 3 * H + g] = w[e];
}

constexpr int gru_nt(int hreg) { return hreg > 0 ? ((3 * hreg + 63) / 64) * 64 : 1024; }

// Where a thread's gate row of W_hh lives.  HTOT == HREG: all of it in registers (H <= 128).  HREG == 0: all of it streamed
// from L2 every step (any H).  Otherwise (round 4, H = 256): the first HREG weights in registers, the next HLDS in LDS (loaded
// once; [j][3H], a thread reads its own column: conflict-free), the remaining HTOT - HREG - HLDS streamed.  At H = 256 the
// matrix is 786 KB per direction — five times the LDS of a CU — and a step that streams all of it is bound by the CU's L2 port
// (64 B/clk: 5.1 us); with 96 + 48 of 256 weights on chip 44 % is streamed.
template <int HREG, int GRU_BT, int HTOT = HREG, int HLDS = 0>
__global__ __launch_bounds__(gru_nt(HTOT)) void gru_seq_fwd_k(
    const float* __restrict__ gi, const float* __restrict__ wt, const float* __restrict__ bhh0,
    const float* __restrict__ bhh1, float* __restrict__ out, float* __restrict__ saved, int B, int T, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* h_s = smem;                    // [BT][H]
    float* gh_s = smem + GRU_BT * H;      // [BT][3H]
    float* w_l = smem + GRU_BT * 4 * H;   // [HLDS][3H] (hybrid only)
    const int g = threadIdx.x;
    const int dir = blockIdx.y, b0 = blockIdx.x * GRU_BT;
    const int H3 = 3 * H;
    const float* wtd = wt + (size_t)dir * H3 * H;
    const float* bhh = dir ? bhh1 : bhh0;
    const bool row = g < H3;
    const bool gate = g < H;
    constexpr bool HYB = HTOT != HREG && HREG > 0;

    float wreg[HREG > 0 ? HREG : 1];
    if (HREG > 0 && row) {
#pragma unroll
        for (int k = 0; k < HREG; ++k) wreg[k] = wtd[(size_t)k * H3 + g];
    }
    if (HYB && row) {
        for (int j = 0; j < HLDS; ++j) w_l[j * H3 + g] = wtd[(size_t)(HREG + j) * H3 + g];
    }
    const float bias = row ? bhh[g] : 0.f;
    for (int i = g; i < GRU_BT * H; i += blockDim.x) h_s[i] = 0.f;
    __syncthreads();

    // input projections of a step are fetched TWO steps ahead (a miss to HBM / another XCD's L2 is ~1-2 us, about a whole
    // step): the copy at the end of step s then waits for loads that were issued during step s-1
    float gr[GRU_BT], gz[GRU_BT], gn[GRU_BT], nr[GRU_BT], nz[GRU_BT], nn[GRU_BT], mr[GRU_BT], mz[GRU_BT], mn[GRU_BT];
    auto fetch_gi = [&](int s, float* a, float* bq, float* c) {
        const int tt = dir ? (T - 1 - s) : s;
#pragma unroll
        for (int b = 0; b < GRU_BT; ++b) {
            int bg = b0 + b;
            if (gate && bg < B && s < T) {
                const float* p = gi + (((size_t)bg * T + tt) * 2 + dir) * H3;
                a[b] = p[g]; bq[b] = p[H + g]; c[b] = p[2 * H + g];
            } else { a[b] = bq[b] = c[b] = 0.f; }
        }
    };
    fetch_gi(0, gr, gz, gn);
    fetch_gi(1, nr, nz, nn);
    for (int s = 0; s < T; ++s) {
        const int tt = dir ? (T - 1 - s) : s;
        fetch_gi(s + 2, mr, mz, mn);
        if (row) {
            float acc[GRU_BT];
#pragma unroll
            for (int b = 0; b < GRU_BT; ++b) acc[b] = bias;
            if (HREG > 0) {
#pragma unroll
                for (int k = 0; k < HREG; k += 4) {
#pragma unroll
                    for (int b = 0; b < GRU_BT; ++b) {
                        f32x4 hv = *(const f32x4*)(h_s + b * H + k);
                        acc[b] += wreg[k] * hv[0] + wreg[k + 1] * hv[1] + wreg[k + 2] * hv[2] + wreg[k + 3] * hv[3];
                    }
                }
            }
            if (HYB) {
#pragma unroll 4
                for (int j = 0; j < HLDS; j += 4) {
                    const float w0 = w_l[j * H3 + g], w1 = w_l[(j + 1) * H3 + g], w2 = w_l[(j + 2) * H3 + g], w3 = w_l[(j + 3) * H3 + g];
#pragma unroll
                    for (int b = 0; b < GRU_BT; ++b) {
                        f32x4 hv = *(const f32x4*)(h_s + b * H + HREG + j);
                        acc[b] += w0 * hv[0] + w1 * hv[1] + w2 * hv[2] + w3 * hv[3];
                    }
                }
            }
            if (HREG == 0 || HYB) {
                for (int k = HYB ? HREG + HLDS : 0; k < H; k += 4) {
                    float w0 = wtd[(size_t)k * H3 + g], w1 = wtd[(size_t)(k + 1) * H3 + g];
                    float w2 = wtd[(size_t)(k + 2) * H3 + g], w3 = wtd[(size_t)(k + 3) * H3 + g];
#pragma unroll
                    for (int b = 0; b < GRU_BT; ++b) {
                        f32x4 hv = *(const f32x4*)(h_s + b * H + k);
                        acc[b] += w0 * hv[0] + w1 * hv[1] + w2 * hv[2] + w3 * hv[3];
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < GRU_BT; ++b) gh_s[b * H3 + g] = acc[b];
        }
        __syncthreads();
        if (gate) {
#pragma unroll
            for (int b = 0; b < GRU_BT; ++b) {
                int bg = b0 + b;
                float ghr = gh_s[b * H3 + g], ghz = gh_s[b * H3 + H + g], ghn = gh_s[b * H3 + 2 * H + g];
                float hp = h_s[b * H + g];
                // v_exp_f32 / v_rcp_f32 forms (abs. error ~1e-7, the serial step is latency-bound): sigmoid, and
                // tanh(a) = sign(a) (1 - e)/(1 + e) with e = exp(-2|a|), which never overflows
                float r = __fdividef(1.f, 1.f + __expf(-(gr[b] + ghr)));
                float z = __fdividef(1.f, 1.f + __expf(-(gz[b] + ghz)));
                float a = gn[b] + r * ghn;
                float e2 = __expf(-2.f * fabsf(a));
                float n = copysignf(__fdividef(1.f - e2, 1.f + e2), a);
                float hn = (1.f - z) * n + z * hp;
                h_s[b * H + g] = hn;
                if (bg < B) {
                    out[((size_t)bg * T + tt) * 2 * H + dir * H + g] = hn;
                    if (saved) {
                        float* sp = saved + ((((size_t)bg * T + tt) * 2 + dir) * 5) * H + g;
                        sp[0] = r; sp[H] = z; sp[2 * H] = n; sp[3 * H] = ghn; sp[4 * H] = hp;
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < GRU_BT; ++b) {
            gr[b] = nr[b]; gz[b] = nz[b]; gn[b] = nn[b];
            nr[b] = mr[b]; nz[b] = mz[b]; nn[b] = mn[b];
        }
        __syncthreads();
    }
}

template <int HREG, int GRU_BT, int HTOT = HREG, int HLDS = 0>
__global__ __launch_bounds__(gru_nt(HTOT)) void gru_seq_bwd_k(
    const float* __restrict__ dout, const float* __restrict__ saved, const float* __restrict__ whh0,
    const float* __restrict__ whh1, float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ bpart,
    int B, int T, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H3 = 3 * H;
    float* dgh_s = smem;                         // [BT][3H]
    float* part_s = smem + GRU_BT * H3;          // [3][BT][H]
    float* w_l = smem + GRU_BT * 2 * H3;         // [HLDS][3H] (hybrid only)
    const int g = threadIdx.x;
    const int dir = blockIdx.y, b0 = blockIdx.x * GRU_BT;
    const float* whh = dir ? whh1 : whh0;
    const bool row = g < H3;
    const bool gate = g < H;
    const int p = g / H, k = g - p * H;          // thread (p,k): column k of gate block p
    constexpr bool HYB = HTOT != HREG && HREG > 0;

    float wreg[HREG > 0 ? HREG : 1];
    if (HREG > 0 && row) {
#pragma unroll
        for (int j = 0; j < HREG; ++j) wreg[j] = whh[((size_t)p * H + j) * H + k];
    }
    if (HYB && row) {
        for (int j = 0; j < HLDS; ++j) w_l[j * H3 + g] = whh[((size_t)p * H + HREG + j) * H + k];
    }
    for (int i = g; i < 3 * GRU_BT * H; i += blockDim.x) part_s[i] = 0.f;
    float dhc[GRU_BT];                           // direct part z*dh carried by the gate thread
#pragma unroll
    for (int b = 0; b < GRU_BT; ++b) dhc[b] = 0.f;
    float sb0 = 0.f, sb1 = 0.f, sb2 = 0.f, sb3 = 0.f;   // bias-gradient partial sums over this block's rows and steps
    __syncthreads();

    float cs[GRU_BT][6], ns[GRU_BT][6], ms[GRU_BT][6];   // r, z, n, gh_n, h_prev, dout of the current / next / next-but-one step
    auto fetch_sv = [&](int s, float (*q)[6]) {
        const int tt = dir ? s : (T - 1 - s);
#pragma unroll
        for (int b = 0; b < GRU_BT; ++b) {
            int bg = b0 + b;
            if (gate && bg < B && s < T) {
                const float* sp = saved + ((((size_t)bg * T + tt) * 2 + dir) * 5) * H + g;
                q[b][0] = sp[0]; q[b][1] = sp[H]; q[b][2] = sp[2 * H]; q[b][3] = sp[3 * H]; q[b][4] = sp[4 * H];
                q[b][5] = dout[((size_t)bg * T + tt) * 2 * H + dir * H + g];
            }
        }
    };
    // two steps ahead, as in the forward kernel (one step ahead where the 1024-thread variant has no registers to spare)
    constexpr bool AHEAD2 = !(HREG == 0 && GRU_BT == 4) && !HYB;      // (hybrid: the register budget goes to the weights)
    fetch_sv(0, cs);
    if (AHEAD2) fetch_sv(1, ns);
    for (int s = 0; s < T; ++s) {
        const int tt = dir ? s : (T - 1 - s);    // reverse of the forward processing order
        if (AHEAD2) fetch_sv(s + 2, ms);
        else fetch_sv(s + 1, ns);
        if (gate) {
#pragma unroll
            for (int b = 0; b < GRU_BT; ++b) {
                int bg = b0 + b;
                float dr_pre = 0.f, dz_pre = 0.f, dn_pre = 0.f, dghn = 0.f, carry = 0.f;
                if (bg < B) {
                    float r = cs[b][0], z = cs[b][1], n = cs[b][2], ghn = cs[b][3], hp = cs[b][4];
                    float dh = cs[b][5] + dhc[b] +
                               part_s[(0 * GRU_BT + b) * H + g] + part_s[(1 * GRU_BT + b) * H + g] +
                               part_s[(2 * GRU_BT + b) * H + g];
                    float dn = dh * (1.f - z);
                    float dz = dh * (hp - n);
                    carry = dh * z;
                    dn_pre = dn * (1.f - n * n);
                    dz_pre = dz * z * (1.f - z);
                    float dr = dn_pre * ghn;
                    dghn = dn_pre * r;
                    dr_pre = dr * r * (1.f - r);
                    size_t o = (((size_t)bg * T + tt) * 2 + dir) * H3 + g;
                    dgi[o] = dr_pre; dgi[o + H] = dz_pre; dgi[o + 2 * H] = dn_pre;
                    dgh[o] = dr_pre; dgh[o + H] = dz_pre; dgh[o + 2 * H] = dghn;
                }
                dhc[b] = carry;
                sb0 += dr_pre; sb1 += dz_pre; sb2 += dn_pre; sb3 += dghn;
                dgh_s[b * H3 + g] = dr_pre; dgh_s[b * H3 + H + g] = dz_pre; dgh_s[b * H3 + 2 * H + g] = dghn;
            }
        }
        __syncthreads();
        if (row) {
            float acc[GRU_BT];
#pragma unroll
            for (int b = 0; b < GRU_BT; ++b) acc[b] = 0.f;
            if (HREG > 0) {
#pragma unroll
                for (int j = 0; j < HREG; j += 4) {
#pragma unroll
                    for (int b = 0; b < GRU_BT; ++b) {
                        f32x4 dv = *(const f32x4*)(dgh_s + b * H3 + p * H + j);
                        acc[b] += wreg[j] * dv[0] + wreg[j + 1] * dv[1] + wreg[j + 2] * dv[2] + wreg[j + 3] * dv[3];
                    }
                }
            }
            if (HYB) {
#pragma unroll 4
                for (int j = 0; j < HLDS; j += 4) {
                    const float w0 = w_l[j * H3 + g], w1 = w_l[(j + 1) * H3 + g], w2 = w_l[(j + 2) * H3 + g], w3 = w_l[(j + 3) * H3 + g];
#pragma unroll
                    for (int b = 0; b < GRU_BT; ++b) {
                        f32x4 dv = *(const f32x4*)(dgh_s + b * H3 + p * H + HREG + j);
                        acc[b] += w0 * dv[0] + w1 * dv[1] + w2 * dv[2] + w3 * dv[3];
                    }
                }
            }
            if (HREG == 0 || HYB) {
                for (int j = HYB ? HREG + HLDS : 0; j < H; j += 4) {
                    const float* wp = whh + ((size_t)p * H + j) * H + k;
                    float w0 = wp[0], w1 = wp[H], w2 = wp[2 * H], w3 = wp[3 * H];
#pragma unroll
                    for (int b = 0; b < GRU_BT; ++b) {
                        f32x4 dv = *(const f32x4*)(dgh_s + b * H3 + p * H + j);
                        acc[b] += w0 * dv[0] + w1 * dv[1] + w2 * dv[2] + w3 * dv[3];
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < GRU_BT; ++b) part_s[(p * GRU_BT + b) * H + k] = acc[b];
        }
#pragma unroll
        for (int b = 0; b < GRU_BT; ++b)
#pragma unroll
            for (int e = 0; e < 6; ++e) { cs[b][e] = ns[b][e]; if (AHEAD2) ns[b][e] = ms[b][e]; }
        __syncthreads();
    }
    if (bpart && gate) {
        float* bp = bpart + ((size_t)(blockIdx.x * 2 + dir) * 4) * H + g;
        bp[0] = sb0; bp[H] = sb1; bp[2 * H] = sb2; bp[3 * H] = sb3;
    }
}

// db_ih = (sum dr, sum dz, sum dn_pre), db_hh = (sum dr, sum dz, sum d(gh_n)); partials summed in block order
__global__ void gru_bias_grad_k(const float* __restrict__ bpart, int nblk, int H, float* __restrict__ dbih0,
                                float* __restrict__ dbih1, float* __restrict__ dbhh0, float* __restrict__ dbhh1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * 4 * H) return;
    int j = i % H, c = (i / H) % 4, dir = i / (4 * H);
    float a = 0.f;
#pragma unroll 8
    for (int b = 0; b < nblk; ++b) a += bpart[((size_t)(b * 2 + dir) * 4 + c) * H + j];
    float* dbih = dir ? dbih1 : dbih0;
    float* dbhh = dir ? dbhh1 : dbhh0;
    if (c < 3) dbih[c * H + j] = a;
    if (c < 2) dbhh[c * H + j] = a;
    if (c == 3) dbhh[2 * H + j] = a;
}

static int gru_threads(int H) { return ((3 * H + 63) / 64) * 64; }
// H = 256, one sample per workgroup: weights 0..GRU_HR-1 of a gate row in registers, the next GRU_HL in LDS, the rest streamed
#define GRU_HR 96
#define GRU_HL 48
#define GRU_HR_FWD 116     // the forward kernel has 20 more registers to spare (148 of the 168 a 768-thread workgroup may use at HR = 96)

extern "C" size_t sed_gru_seq_workspace_bytes(int H) { return (size_t)2 * 3 * H * H * sizeof(float); }

#define GRU_BT3(HR, KERNEL, ...)                                                              \
    if (bt == 1) KERNEL<HR, 1><<<grid, nt, lds, s>>>(__VA_ARGS__);                            \
    else if (bt == 2) KERNEL<HR, 2><<<grid, nt, lds, s>>>(__VA_ARGS__);                       \
    else KERNEL<HR, 4><<<grid, nt, lds, s>>>(__VA_ARGS__)
#define GRU_DISPATCH(KERNEL, ...)                                                             \
    switch (H) {                                                                              \
        case 8: GRU_BT3(8, KERNEL, __VA_ARGS__); break;                                       \
        case 16: GRU_BT3(16, KERNEL, __VA_ARGS__); break;                                     \
        case 32: GRU_BT3(32, KERNEL, __VA_ARGS__); break;                                     \
        case 64: GRU_BT3(64, KERNEL, __VA_ARGS__); break;                                     \
        case 128:                                                                             \
            if (bt == 1) KERNEL<128, 1><<<grid, nt, lds, s>>>(__VA_ARGS__);                   \
            else KERNEL<128, 2><<<grid, nt, lds, s>>>(__VA_ARGS__);                           \
            break;                                                                            \
        case 256:                                                                             \
            if (bt == 1) { hyb = 1; KERNEL<GRU_HRX, 1, 256, GRU_HL><<<grid, nt, lds + (size_t)GRU_HL * 768 * sizeof(float), s>>>(__VA_ARGS__); } \
            else { GRU_BT3(0, KERNEL, __VA_ARGS__); }                                         \
            break;                                                                            \
        default: GRU_BT3(0, KERNEL, __VA_ARGS__); break;                                      \
    }

extern "C" int sed_gru_seq_fwd(const float* gi, const float* const* whh, const float* const* bhh, float* out,
                               float* saved, void* workspace, int B, int T, int H, void* stream) {
    SED_REQUIRE(gi && whh && bhh && out && workspace && whh[0] && whh[1] && bhh[0] && bhh[1], "gru_seq_fwd: null pointer");
    SED_REQUIRE(B > 0 && T > 0 && H > 0 && H % 4 == 0 && 3 * H <= 1024, "gru_seq_fwd: H=%d must be a multiple of 4 and <= 341", H);
    hipStream_t s = as_stream(stream);
    float* wt = (float*)workspace;
    int n = 2 * 3 * H * H;
    gru_pack_whh_t_k<<<cdiv(n, 256), 256, 0, s>>>(whh[0], whh[1], wt, H);
    SED_LAUNCH_CHECK("gru_pack_whh_t");
    const int bt = gru_bt(H, B);
    dim3 grid(cdiv(B, bt), 2);
    int nt = gru_threads(H);
    size_t lds = (size_t)bt * 4 * H * sizeof(float);
    SedProfScope prof(SED_K_GRU_FWD, s, 2.0 * 2 * B * (double)T * 3 * H * H);
    int hyb = 0;
    if (H == 256 && bt == 1)
        (void)hipFuncSetAttribute((const void*)gru_seq_fwd_k<GRU_HR_FWD, 1, 256, GRU_HL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define GRU_HRX GRU_HR_FWD
    GRU_DISPATCH(gru_seq_fwd_k, gi, wt, bhh[0], bhh[1], out, saved, B, T, H);
#undef GRU_HRX
    (void)hyb;
    SED_LAUNCH_CHECK("gru_seq_fwd");
    return 0;
}

extern "C" size_t sed_gru_seq_bwd_workspace_bytes(int B, int H) { return (size_t)cdiv(B, gru_bt(H, B)) * 2 * 4 * H * sizeof(float); }

extern "C" int sed_gru_seq_bwd(const float* dout, const float* saved, const float* const* whh, float* dgi,
                               float* dgh, float* const* dbih, float* const* dbhh, void* workspace, int B, int T,
                               int H, void* stream) {
    SED_REQUIRE(dout && saved && whh && whh[0] && whh[1] && dgi && dgh, "gru_seq_bwd: null pointer");
    const bool want_bias = dbih && dbhh;
    SED_REQUIRE(!want_bias || (workspace && dbih[0] && dbih[1] && dbhh[0] && dbhh[1]), "gru_seq_bwd: bias gradients need all four outputs and a workspace");
    float* bpart = want_bias ? (float*)workspace : nullptr;
    SED_REQUIRE(B > 0 && T > 0 && H > 0 && H % 4 == 0 && 3 * H <= 1024, "gru_seq_bwd: H=%d must be a multiple of 4 and <= 341", H);
    hipStream_t s = as_stream(stream);
    const int bt = gru_bt(H, B);
    dim3 grid(cdiv(B, bt), 2);
    int nt = gru_threads(H);
    size_t lds = (size_t)bt * 6 * H * sizeof(float);
    SedProfScope prof(SED_K_GRU_BWD, s, 2.0 * 2 * B * (double)T * 3 * H * H);
    int hyb = 0;
    if (H == 256 && bt == 1)
        (void)hipFuncSetAttribute((const void*)gru_seq_bwd_k<GRU_HR, 1, 256, GRU_HL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define GRU_HRX GRU_HR
    GRU_DISPATCH(gru_seq_bwd_k, dout, saved, whh[0], whh[1], dgi, dgh, bpart, B, T, H);
#undef GRU_HRX
    (void)hyb;
    SED_LAUNCH_CHECK("gru_seq_bwd");
    if (want_bias) {
        gru_bias_grad_k<<<cdiv(8 * H, 256), 256, 0, s>>>(bpart, cdiv(B, bt), H, dbih[0], dbih[1], dbhh[0], dbhh[1]);
        SED_LAUNCH_CHECK("gru_bias_grad");
    }
    return 0;
}
