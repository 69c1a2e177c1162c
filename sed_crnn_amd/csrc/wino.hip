// wino.hip — the 3x3 convolution of the 128-channel blocks as Winograd F(2x2, 3x3), exact-fp32 MFMA.
//
// Same operator as conv3x3_mfma_fwd2_k (conv.hip; reference sed.py:88,107 nn.Conv2d forward and, with flipped / transposed
// weights, its data gradient), 2.25x fewer multiplications: a 2x2 output tile is computed from its 4x4 input patch d as
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A,
// 16 products per input / output channel pair instead of 36.  On gfx950 the fp32 MFMA (v_mfma_f32_32x32x2_f32: 256 flop per
// cycle and CU) is the ceiling of the direct kernel (0.87 of it reached), so the only way below it is fewer MFMAs.  The
// transforms use the coefficients 0, +-1 (input, output) and 1, 1/2 (weights): in fp32 the result differs from the direct sum by
// a few ulp of the accumulated magnitude (measured in tests/test_gpu_kernels.py next to the direct kernel's own error), far
// inside the 1e-3 the north star asks of the outputs — and CLOSER to float64 than the direct kernel's (each component sums K = 128
// terms instead of K = 1152: max error 2.4e-6 against 1.1e-5 of the output scale).  Wide mel axes (128 bins) are cut into column
// groups so that a block's patch fits the LDS.
//
// GEMM view: 16 independent products M_k[tile][co] = sum_ci V_k[tile][ci] U_k[ci][co], k = (xi, nu) the component,
// tile = a 2x2 output tile.  A workgroup owns 64 consecutive tiles of one sequence (tiles linearised over (time pair, mel pair):
// any mel width fills the MFMA rows) x 64 output channels; wave xi holds the four components (xi, 0..3) of 2 x 2 MFMA tiles
// (32 tiles x 32 channels each): 16 accumulators = 256 registers, ONE wave per SIMD.
//   * input: the patch of the block's tile rows (2 TR + 2 time rows x F + 2 mel columns x 32 channels per slice) comes in by
//     LDS-DMA (global_load_lds_dwordx4, no staging registers), double buffered, one barrier per 32-channel slice.  LDS image:
//     8 positions per KiB, quad-major inside ([quad][position]): a wave's DMA covers 8 whole positions (coalesced), a lane's
//     ds_read_b128 of (position, quad) is conflict-free over consecutive positions, and the k-group g of a step is an
//     immediate offset.  Mel columns are stored even ones first, so that the tiles of a row are consecutive positions.
//   * the input transform happens in registers on the way to the A operand: wave xi reads the two time rows of its row
//     combination (B^T d: d0 - d2, d1 + d2, d2 - d1, d1 - d3), 4 columns each, and forms the four column combinations on
//     register PAIRS (v_pk_fma_f32 / v_pk_add_f32): 16 ds_read_b128 + 32 vector instructions per 64 MFMAs, shared by both
//     channel tiles, one step ahead of their use.
//   * weights: U = G g G^T is formed by the packing launch in MFMA B-fragment order, [co/64][component][ci/32][(ci%32)/8][(co%64)/32]
//     [lane][4]; a wave streams its 4 components x 2 channel tiles from L2, the fragments of a component for the next
//     step requested as soon as this step's MFMAs of that component have issued (rolling, 48 MFMAs ahead).
//   * the instruction order of a step is written out, slot by slot behind each MFMA (see the kernel): one wave per SIMD issues in
//     order, and what an instruction costs beside the MFMAs is its issue slot.
//   * output transform: in registers along nu (4 -> 2), across the four waves (xi) through LDS, then the same row-wise
//     epilogue as the direct kernel: 4 channels x 8 rows per store instruction, BatchNorm statistics (forward) or the
//     BatchNorm-backward sums of the block below (data gradient, ConvBnRed; RGC: + the first block's tap sums) from the values
//     in registers, or ReLU + the (1,2) time pool (EV, inference: the two time rows of a tile are the pooling pair).
// Fixed summation order: run-to-run identical results.
#include "common.h"
#include "conv_shared.h"
#include <type_traits>

#define WN_CIN 128          // input channels (the contraction): 4 slices of 32, 16 steps of 8 — fully unrolled
#define WN_NHMAX 14         // KiB blocks of the patch per wave (56 KiB per buffer)

typedef __attribute__((address_space(1))) const void* sed_gptr_t;
typedef __attribute__((address_space(3))) void* sed_lptr_t;

struct WinoGeo {
    int Fw;        // tiles per row of a COLUMN GROUP (the mel axis is cut into ncg groups of Fw tiles when a whole row's patch is too large)
    int ncg, Tw, ntile, nblk, TR, F2, H2, HR, nb8, ncoh, NH;      // ntile / nblk: per sequence and column group
    float invFw, invF2;
    size_t lds;
    unsigned long long* dbg;      // measurement only (sed_conv3x3_wino_phase_ticks): per-phase s_memrealtime sums, NULL in production
};

// ───────────────────────── weight transform + packing ─────────────────────────
__global__ void conv_pack_wino_k(const float* __restrict__ w, float* __restrict__ uf, float* __restrict__ ud, int Cout, int Cin,
                                 const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ rm, const float* __restrict__ rv, float eps, float* __restrict__ bias_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (gamma && i < Cout) {                         // inference: BatchNorm on running statistics folded into weights and bias
        const float sc = gamma[i] / sqrtf(rv[i] + eps);
        bias_out[i] = (bias ? bias[i] : 0.f) * sc + (beta[i] - rm[i] * sc);
    }
    wino_pack_one(w, uf, ud, Cout, Cin, i, gamma, rv, eps);
}

extern "C" size_t sed_conv3x3_wino_packed_floats(int Cout, int Cin) { return (size_t)16 * Cout * Cin + WN_ZTAIL; }
extern "C" int sed_conv3x3_wino_pack_weights(const float* w, float* uf, float* ud, int Cout, int Cin, void* stream) {
    SED_REQUIRE(w && (uf || ud) && Cout > 0 && Cin > 0 && Cout % 64 == 0 && Cin % 64 == 0, "conv3x3_wino_pack_weights: bad arguments (channel counts: multiples of 64)");
    conv_pack_wino_k<<<cdiv(Cout * Cin, 256), 256, 0, as_stream(stream)>>>(w, uf, ud, Cout, Cin, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr);
    SED_LAUNCH_CHECK("conv_pack_wino");
    return 0;
}
extern "C" int sed_conv3x3_wino_pack_weights_bn_folded(const float* w, const float* bias, const float* gamma, const float* beta,
                                                       const float* running_mean, const float* running_var, float eps,
                                                       float* uf, float* bias_folded, int Cout, int Cin, void* stream) {
    SED_REQUIRE(w && gamma && beta && running_mean && running_var && uf && bias_folded && Cout > 0 && Cin > 0 && Cout % 64 == 0 && Cin % 64 == 0,
                "conv3x3_wino_pack_weights_bn_folded: bad arguments (channel counts: multiples of 64)");
    conv_pack_wino_k<<<cdiv(Cout * Cin, 256), 256, 0, as_stream(stream)>>>(w, uf, nullptr, Cout, Cin, bias, gamma, beta, running_mean, running_var, eps, bias_folded);
    SED_LAUNCH_CHECK("conv_pack_wino (folded)");
    return 0;
}

// ───────────────────────── geometry ─────────────────────────
static bool wino_geo(int B, int Cin, int F, int T, int Cout, WinoGeo* g) {
    if (B <= 0 || Cin != WN_CIN || Cout <= 0 || Cout % 64 != 0 || F < 2 || T < 2 || (F & 1) || (T & 1)) return false;
    if ((size_t)B * T * F * Cin >= ((size_t)1 << 32) || (size_t)T * F * Cout >= ((size_t)1 << 30)) return false;
    g->Tw = T / 2;
    // 64 consecutive tiles of a column group: its tile rows' patch (2 TR + 2 time rows x 2 Fw + 2 mel columns) has to fit WN_NHMAX KiB
    // per wave and 32-channel slice.  Whole rows first (no column halo between workgroups); 128 mel bins: two groups of 32 tiles
    // (two tile rows per block: 6 x 66 positions instead of 4 x 130)
    bool ok = false;
    for (int ncg = 1; ncg <= 8 && !ok; ncg *= 2) {
        if ((F / 2) % ncg) break;
        g->ncg = ncg; g->Fw = F / 2 / ncg; g->ntile = g->Fw * g->Tw; g->nblk = cdiv(g->ntile, 64);
        int TR = 0;
        for (int blk = 0; blk < g->nblk; ++blk) {
            const int q1 = blk * 64 + 63 < g->ntile ? blk * 64 + 63 : g->ntile - 1;
            const int n = q1 / g->Fw - (blk * 64) / g->Fw + 1;
            if (n > TR) TR = n;
        }
        g->TR = TR; g->F2 = 2 * g->Fw + 2; g->H2 = g->F2 / 2; g->HR = (2 * TR + 2) * g->F2; g->nb8 = cdiv(g->HR, 8);
        g->NH = cdiv(g->nb8, 4);
        ok = g->NH <= WN_NHMAX;
    }
    if (!ok) return false;
    g->NH = WN_NHMAX;
    g->ncoh = Cout / 64;
    g->invFw = 1.0f / (float)g->Fw; g->invF2 = 1.0f / (float)g->F2;
    // two patch buffers; the epilogue's exchange of the four waves' partial output transforms (4 x 2 x 2 x 2 tiles of 4 KB) + the
    // statistics exchange reuse them; the row table behind
    size_t fl = (size_t)2 * g->NH * 1024;
    if (fl < 32768 + 512) fl = 32768 + 512;
    g->lds = (fl + 64) * sizeof(float);              // + the row table
    g->dbg = nullptr;
    return g->lds <= 160 * 1024;
}

int sed_internal_wino_rows(int B, int Cin, int F, int T, int Cout) {
    WinoGeo g;
    return wino_geo(B, Cin, F, T, Cout, &g) ? B * g.nblk * g.ncg : 0;
}

// packed fp32 pairs (see the kernel)
__device__ __forceinline__ f32x2 wn_pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 wn_pk_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 wn_pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// ───────────────────────── the kernel ─────────────────────────
// RGC > 0 (data gradient of the block above the recomputed first block with RGC input channels, pool (1,2)): the epilogue also forms
// that block's weight-gradient sums R_k from the gradient values in registers, exactly as the direct kernel's RG epilogue does
// (conv.hip): the network input under the block's tile rows is held in LDS, the arg-max bits pick the time row.
// EV (inference, sed.py:128-141 with optim=None): BatchNorm is folded into the transformed weights and the bias by the packing
// launch, and the epilogue applies ReLU + the (1,2) time pool before anything is written — the two time rows of a 2x2 tile ARE a
// pooling pair: `y` is the pooled output [B][T/2][F][Cout], the un-pooled tensor never exists.
// ABL (measurement builds only, -DWN_ABLATION: wrong results on purpose): 1 = no transform arithmetic, 2 = no weight-fragment loads
// in the loop, 4 = no operand reads from LDS in the loop, 8 = no patch DMA in the loop
template <int NH, bool BNR, int RGC = 0, bool EV = false, int ABL = 0>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_k(
    const float* __restrict__ x, const float* __restrict__ uq, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ stat, int B, int F, int T, int Cout, WinoGeo geo, ConvBnRed br) {
    constexpr int CIN = WN_CIN, NCHUNK = CIN / 32, NSTEP = 4 * NCHUNK;
    constexpr int HBUF = NH * 1024;                       // floats per patch buffer (NH KiB per wave)
    constexpr bool RG = RGC > 0;
    static_assert(!RG || BNR, "the tap sums belong to the BatchNorm-backward epilogue");
    static_assert(!EV || !BNR, "the pooling epilogue belongs to the inference forward");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    unsigned long long tk0 = 0, tk1 = 0, tk2 = 0;
    if (geo.dbg) tk0 = __builtin_amdgcn_s_memrealtime();

    // workgroup -> (sequence, tile block, channel half).  Workgroups are dealt round-robin to the 8 XCDs in launch order: the
    // channel halves of a block and the blocks of a sequence are given to the same XCD (they share the input patch through its L2)
    int blka, coh, b;                                  // blka = block x column group of the sequence
    {
        const int bx = blockIdx.x, by = blockIdx.y, nba = geo.nblk * geo.ncg;
        if ((gridDim.y & 7) == 0) {
            const int id = by * (int)gridDim.x + bx, xcd = id & 7, j = id >> 3;
            coh = j % geo.ncoh;
            const int jj = j / geo.ncoh;
            b = xcd + 8 * (jj / nba);
            blka = jj % nba;
        } else {
            coh = bx % geo.ncoh; blka = bx / geo.ncoh; b = by;
        }
    }
    const int blk = blka / geo.ncg, cgp = blka - blk * geo.ncg;      // neighbouring column groups are neighbours in launch order
    const int fg0 = 2 * cgp * geo.Fw;                                   // first mel column of the group
    const int Fw = geo.Fw, F2 = geo.F2, H2 = geo.H2, ntile = geo.ntile;
    const int q0 = blk * 64, ty0 = q0 / Fw;
    const int co0 = coh * 64;

    // this wave's row combination of B^T d: rows (ra, rb), d[ra] + sg d[rb]
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sg = wave == 1 ? 1.f : -1.f;

    // patch staging (LDS-DMA): item = KiB block k = 4 u + wave, lane = 8 quad + position-in-block.  Every lane of every block issues
    // (no branch: the loads interleave with the MFMAs); a lane whose position is zero padding, or lies past the patch, reads the
    // zero tail behind the packed weights instead.
    const float* hp[NH];
    {
        const int Q = lane >> 3, pp = lane & 7;
        const float* zrow = uq + (size_t)16 * CIN * Cout;
#pragma unroll
        for (int u = 0; u < NH; ++u) {
            const int P = (u * 4 + wave) * 8 + pp;
            const int tt = sed_fdiv(P, geo.invF2), cx = P - tt * F2;
            const int ff = cx < H2 ? 2 * cx : 2 * (cx - H2) + 1;
            const int t = 2 * ty0 - 1 + tt, f = fg0 + ff - 1;
            const bool in = P < geo.HR && t >= 0 && t < T && f >= 0 && f < F;
            hp[u] = in ? x + ((((size_t)b * T + t) * F + f) * CIN + Q * 4) : zrow + Q * 4;
        }
    }
    auto issue = [&](int cc, float* buf) {
#pragma unroll
        for (int u = 0; u < NH; ++u)
            __builtin_amdgcn_global_load_lds((sed_gptr_t)(hp[u] + cc * 32), (sed_lptr_t)(buf + (u * 4 + wave) * 256), 16, 0, 0);
    };
    // the first slice is requested before anything else is set up (its latency is the workgroup's prologue: 4.4 us)
    issue(0, smem);
    __builtin_amdgcn_sched_barrier(0);

    // LDS float offset of (patch position P, quad h) for the 2 m-tiles x 2 rows x 4 columns this lane reads every step
    int aoff[2][2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        int q = q0 + mt * 32 + r;
        q = q < ntile ? q : ntile - 1;
        const int ty = sed_fdiv(q, geo.invFw), tf = q - ty * Fw;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int P = (2 * (ty - ty0) + (s ? rb : ra)) * F2 + tf + (c & 1) * H2 + (c >> 1);
                aoff[mt][s][c] = (P >> 3) * 256 + h * 32 + (P & 7) * 4;
            }
    }


    // the epilogue's table: byte offset of output position (2 ty, 2 tf), channel 0, inside sequence b; ~0: no such tile
    unsigned* rowtab = (unsigned*)(smem + (2 * HBUF > 32768 + 512 ? 2 * HBUF : 32768 + 512));
    // RG: a second table (where the tile's 3 x 4 input windows start in xs) and xs itself, the (F + 2) x (4 TR + 2) patch of the
    // network input under this block's tile rows (time fastest, zero outside the input), one plane per input channel
    unsigned* xofftab = rowtab + 64;
    float* xs = (float*)(xofftab + 64);
    const int XT = 4 * geo.TR + 2;
    if (tid < 64) {
        const int q = q0 + tid;
        const int ty = sed_fdiv(q < ntile ? q : 0, geo.invFw), tf = q - ty * Fw;
        rowtab[tid] = q < ntile ? (unsigned)(((EV ? ty : 2 * ty) * F + fg0 + 2 * tf) * Cout) * 4u : 0xFFFFFFFFu;
        if (RG) xofftab[tid] = (unsigned)((fg0 + 2 * tf) * XT + 4 * (ty - ty0));
    }
    if (RG) {
        for (int i = tid; i < (F + 2) * XT; i += 256) {
            const int ff = i / XT, tt = i - ff * XT;
            const int f = ff - 1, t = 4 * ty0 - 1 + tt;          // (all F + 2 mel columns: indexed by the absolute column)
            const bool in = f >= 0 && f < F && t >= 0 && t < br.Ty;
#pragma unroll
            for (int ci = 0; ci < RGC; ++ci)
                xs[ci * (F + 2) * XT + i] = in ? br.x1[(((size_t)b * RGC + ci) * F + f) * br.Ty + t] : 0.f;
        }
    }

    f32x16 acc[4][2][2];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[nu][mt][nt][j] = 0.f;

    const f32x4* wl = (const f32x4*)uq + (size_t)(coh * 16 + wave * 4) * (NCHUNK * 4 * 2 * 64);      // wave-uniform
    f32x4 bq[4][2];
    const unsigned lane16 = (unsigned)lane * 16u;
    auto load_b = [&](int nu, int st) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)          // (scalar base + 32-bit lane offset: the address costs scalar adds, no vector instruction)
            bq[nu][nt] = *(const f32x4*)((const char*)(wl + ((nu * NSTEP + st) * 2 + nt) * 64) + lane16);
    };
    // Operands of a step: vp[parity][mt][nu][half] — the four k values (j) of an MFMA group as two register PAIRS.  The input transform
    // runs on pairs (v_pk_fma_f32 / v_pk_add_f32, written out: hipcc splits <2 x float> arithmetic into scalar instructions): beside the
    // MFMAs an instruction costs its issue slot, not the vector ALU's time, so 32 instead of 64 per step.  Inline assembly is invisible to
    // sched_group_barrier, so a step's instruction order is written out instead: after MFMA i of the 64 comes "slot i" —
    //   slots 0-15   the 16 ds_read_b128 of the NEXT step's patch values (m-tile, row of the wave's row combination, column),
    //   slots 16-47  its 32 transform instructions: per m-tile 8 x  u = d[ra] + sg d[rb]  then 8 x column combinations,
    //   slot 16 nu + 15  the next step's two weight fragments of component nu (its MFMAs of this step have all been issued),
    //   slots 1, 5, 9, ...  one patch DMA each in the first step of a slice
    //   (measured alternatives, no better: the fragments one per slot at their last use, the DMA in the empty slots 49-62) —
    // and a scheduling barrier after every slot keeps it that way.
    static_assert(NH <= 16, "the patch DMA of a slice rides in every fourth slot of its first step");
    f32x2 vp[2][2][4][2];
    f32x4 raw[2][2][4];
    f32x2 up[2][4][2];
    const f32x2 sg2 = {sg, sg};
    auto slot_read = [&](const float* buf, int g, int i) {                 // i = 0..15
        const int mt = i >> 3, s2 = (i >> 2) & 1, c = i & 3;
        raw[mt][s2][c] = *(const f32x4*)(buf + aoff[mt][s2][c] + g * 64);
    };
    auto slot_xform = [&](int k, f32x2 (&v)[2][4][2]) {                    // k = 0..31
        const int mt = k >> 4, q = k & 15, hf = q & 1;
        if (q < 8) {
            const int c = q >> 1;
            const f32x2 da = hf ? __builtin_shufflevector(raw[mt][0][c], raw[mt][0][c], 2, 3) : __builtin_shufflevector(raw[mt][0][c], raw[mt][0][c], 0, 1);
            const f32x2 db = hf ? __builtin_shufflevector(raw[mt][1][c], raw[mt][1][c], 2, 3) : __builtin_shufflevector(raw[mt][1][c], raw[mt][1][c], 0, 1);
            up[mt][c][hf] = wn_pk_fma(sg2, db, da);
        } else {
            const int nu = (q - 8) >> 1;
            if (nu == 0) v[mt][0][hf] = wn_pk_sub(up[mt][0][hf], up[mt][2][hf]);
            else if (nu == 1) v[mt][1][hf] = wn_pk_add(up[mt][1][hf], up[mt][2][hf]);
            else if (nu == 2) v[mt][2][hf] = wn_pk_sub(up[mt][2][hf], up[mt][1][hf]);
            else v[mt][3][hf] = wn_pk_sub(up[mt][1][hf], up[mt][3][hf]);
        }
    };
    auto issue_one = [&](int cc, float* buf, int u) {
        __builtin_amdgcn_global_load_lds((sed_gptr_t)(hp[u] + cc * 32), (sed_lptr_t)(buf + (u * 4 + wave) * 256), 16, 0, 0);
    };

#pragma unroll
    for (int nu = 0; nu < 4; ++nu) load_b(nu, 0);
    __syncthreads();                                   // drains the DMA (vmcnt)
#pragma unroll
    for (int i = 0; i < 16; ++i) slot_read(smem, 0, i);
#pragma unroll
    for (int k = 0; k < 32; ++k) slot_xform(k, vp[0]);
    if (geo.dbg) tk1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
        const int cc = st >> 2, g = st & 3, par = st & 1;
        const bool more = cc + 1 < NCHUNK, nxt = st + 1 < NSTEP;
        const float* nbuf = smem + (((st + 1) >> 2) & 1) * HBUF;     // the next step's slice (the other buffer after step (cc, 3): complete since the barrier of (cc, 2))
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const int nu = i >> 4, j = (i >> 2) & 3, mt = (i >> 1) & 1, nt = i & 1;
            acc[nu][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[par][mt][nu][j >> 1][j & 1], bq[nu][nt][j], acc[nu][mt][nt], 0, 0, 0);
            if (nxt && !(ABL & 4)) {
                if (i < 16) slot_read(nbuf, (st + 1) & 3, i);
                else if (i < 48 && !(ABL & 1)) slot_xform(i - 16, vp[par ^ 1]);
            }
            if (nxt && (i & 15) == 15 && !(ABL & 2)) load_b(nu, st + 1);
            if (g == 0 && more && !(ABL & 8) && (i & 3) == 1 && (i >> 2) < NH) issue_one(cc + 1, smem + ((cc + 1) & 1) * HBUF, i >> 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (nxt && (ABL & 5)) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) { vp[par ^ 1][a][c][0] = vp[par][a][c][0]; vp[par ^ 1][a][c][1] = vp[par][a][c][1]; }
        }
        if (g == 2 && more) __syncthreads();           // the next slice's patch is complete (DMA drained) and visible
    }
    __syncthreads();                                   // every wave has read its last operands: the patch buffers are free
    if (geo.dbg) tk2 = __builtin_amdgcn_s_memrealtime();

    // ── output transform ──
    // along nu in registers: Z_j = (M0 + M1 + M2, M1 - M2 - M3); each wave parks its Z[xi][j][mt][nt] as a transposed 32 x 32 tile
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f32x16 z0 = acc[0][mt][nt] + acc[1][mt][nt] + acc[2][mt][nt];
            const f32x16 z1 = acc[1][mt][nt] - acc[2][mt][nt] - acc[3][mt][nt];
            float* zb0 = smem + ((((wave * 2 + 0) * 2 + mt) * 2 + nt) << 10);
            float* zb1 = smem + ((((wave * 2 + 1) * 2 + mt) * 2 + nt) << 10);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
                zb0[row * 32 + r] = z0[j];
                zb1[row * 32 + r] = z1[j];
            }
        }
    __builtin_amdgcn_sched_barrier(0);                 // (the exchange tiles are written: their registers are free for the pooled values)
    // across xi: wave w finishes m-tile w >> 1, channel tile w & 1; lane = (row group rq, channels c4 .. c4+3)
    const int mt = wave >> 1, nt = wave & 1;
    const int rq = lane >> 3, c4 = (lane & 7) * 4;
    const int cb = co0 + nt * 32 + c4;
    f32x4 bv = {0, 0, 0, 0};
    if (bias) bv = *(const f32x4*)(bias + cb);
    char* const yb = (char*)(y + (size_t)b * (EV ? T >> 1 : T) * F * Cout + cb);
    const char* const qb = BNR ? (const char*)(br.pooled + (size_t)b * T * F * Cout + cb) : nullptr;
    f32x4 q_kr = {0, 0, 0, 0}, q_nb = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
    if (BNR) {
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
    const unsigned rstride = (unsigned)(F * Cout) * 4u, cstride = (unsigned)Cout * 4u;
    f32x4 R[RG ? 9 * RGC : 1];
#pragma unroll
    for (int k = 0; k < (RG ? 9 * RGC : 1); ++k) R[k] = (f32x4){0, 0, 0, 0};
    const unsigned char* const bitq = RG ? br.bits + (size_t)b * T * F * (Cout >> 2) + (cb >> 2) : nullptr;
    // BNR: the pooled values (and RG: the arg-max bytes) of this lane's output positions are requested ahead of their use — one
    // dependent global load per position inside the loop below cost 8 us per workgroup, one row ahead still 4
    unsigned rov[4], xov[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        rov[k] = rowtab[mt * 32 + rq + 8 * k];
        xov[k] = RG ? xofftab[mt * 32 + rq + 8 * k] : 0u;
    }
    // (all four rows at once, in front of the exchange barrier: 16 loads in flight together instead of four dependent rounds;
    // two input channels: 72 tap accumulators leave room for two rows only)
    constexpr int PF = RGC == 2 ? 2 : 4;
    f32x4 pqv[PF][BNR ? 4 : 1];
    unsigned btv[PF][RG ? 4 : 1];
    auto prefetch = [&](int k, f32x4* pqd, unsigned* btd) {
        if (!BNR) return;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned off = rov[k] + i * rstride + jj * cstride;
                pqd[jj * 2 + i] = (f32x4){0, 0, 0, 0};
                if (RG) btd[jj * 2 + i] = 0u;
                if (rov[k] != 0xFFFFFFFFu) {
                    pqd[jj * 2 + i] = *(const f32x4*)(qb + off);
                    if (RG) btd[jj * 2 + i] = bitq[off >> 4];
                }
            }
    };
    if (PF == 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) prefetch(k, pqv[k], btv[k]);
    } else prefetch(0, pqv[0], btv[0]);

    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int m = rq + 8 * k;
        const unsigned ro = rov[k];
        if (PF == 2 && k + 1 < 4) prefetch(k + 1, pqv[(k + 1) & 1], btv[(k + 1) & 1]);
        if (ro == 0xFFFFFFFFu) continue;
        const unsigned xo = xov[k];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            f32x4 z[4];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) z[xi] = *(const f32x4*)(smem + ((((xi * 2 + jj) * 2 + mt) * 2 + nt) << 10) + m * 32 + c4);
            const f32x4 o[2] = {z[0] + z[1] + z[2] + bv, z[1] - z[2] - z[3] + bv};
            if (EV) {
                f32x4 pl;
#pragma unroll
                for (int e = 0; e < 4; ++e) pl[e] = fmaxf(fmaxf(o[0][e], o[1][e]), 0.f);
                *(f32x4*)(yb + ro + jj * cstride) = pl;
                continue;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned off = ro + i * rstride + jj * cstride;
                const f32x4 v = o[i];
                *(f32x4*)(yb + off) = v;
                if (BNR) {
                    const f32x4 pq = pqv[k & (PF - 1)][jj * 2 + i];
                    f32x4 g0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) g0[e] = pq[e] > 0.f ? v[e] : 0.f;
                    a1 += g0;
                    a2 += g0 * (pq * q_kr + q_nb);
                    if (RG) {
                        const unsigned bt = btv[k & (PF - 1)][jj * 2 + i];
                        f32x4 g1;                         // the share of the window's second time row
#pragma unroll
                        for (int e = 0; e < 4; ++e) g1[e] = ((bt >> e) & 1u) ? g0[e] : 0.f;
                        const f32x4 gA = g0 - g1;
#pragma unroll
                        for (int ci = 0; ci < RGC; ++ci) {
                            const float* xb = xs + ci * (F + 2) * XT + xo + jj * XT + 2 * i;
#pragma unroll
                            for (int kh = 0; kh < 3; ++kh) {
                                const f32x2 x01 = *(const f32x2*)(xb + kh * XT), x23 = *(const f32x2*)(xb + kh * XT + 2);
                                R[(kh * 3 + 0) * RGC + ci] += gA * x01[0] + g1 * x01[1];
                                R[(kh * 3 + 1) * RGC + ci] += gA * x01[1] + g1 * x23[0];
                                R[(kh * 3 + 2) * RGC + ci] += gA * x23[0] + g1 * x23[1];
                            }
                        }
                    }
                } else {
                    a1 += v;
                    a2 += v * v;
                }
            }
        }
    }
    if (stat && !EV) {
        // a1 / a2: forward (sum y, sum y^2), data gradient (sum g, sum g*xhat) of this lane's four channels over its rows
        float* red = smem + 32768;                    // [4 waves][2][32] behind the exchange tiles
        if (BNR) { a1 *= br.inv_keep; a2 *= br.inv_keep; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) { a1[e] += __shfl_xor(a1[e], o, 64); a2[e] += __shfl_xor(a2[e], o, 64); }
        }
        if (lane < 8) {
            *(f32x4*)(red + (wave * 2 + 0) * 32 + c4) = a1;
            *(f32x4*)(red + (wave * 2 + 1) * 32 + c4) = a2;
        }
        float* red2 = xs + RGC * (F + 2) * XT;              // [4 waves][9 RGC][32]: this wave's 32 channels of the R_k (behind xs, which others may still read)
        if (RG) {
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
        __syncthreads();
        const size_t row = (size_t)b * geo.nblk * geo.ncg + blka;
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63, cn = c >> 5, cr = c & 31;
            const float a = red[((0 * 2 + cn) * 2 + which) * 32 + cr] + red[((1 * 2 + cn) * 2 + which) * 32 + cr];
            stat[row * 2 * Cout + which * Cout + co0 + c] = a;
        }
        if (RG && tid < 64) {                           // channels co0 + tid: waves cn (m-tile 0) and 2 + cn (m-tile 1)
            const int cn = tid >> 5, cr = tid & 31;
            float* o = br.rgp + (row * Cout + co0 + tid) * (1 + 9 * RGC);
            o[0] = red[((0 * 2 + cn) * 2 + 0) * 32 + cr] + red[((1 * 2 + cn) * 2 + 0) * 32 + cr];      // sum g (= the BatchNorm sum above)
#pragma unroll
            for (int k = 0; k < 9 * RGC; ++k)
                o[1 + k] = red2[((0 * 2 + cn) * 9 * RGC + k) * 32 + cr] + red2[((1 * 2 + cn) * 9 * RGC + k) * 32 + cr];
        }
    }
    if (geo.dbg && tid == 0) {
        const unsigned long long tk3 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(geo.dbg + 0, tk1 - tk0);
        atomicAdd(geo.dbg + 1, tk2 - tk1);
        atomicAdd(geo.dbg + 2, tk3 - tk2);
        atomicAdd(geo.dbg + 3, 1ull);
    }
}

template <typename K>
static int wino_set_lds(K kernel, size_t bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { sed_set_error("hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e)); return (int)e; }
    return 0;
}

static unsigned long long* g_wino_dbg = nullptr;
#ifdef WN_ABLATION
static int g_wino_abl = 0;
extern "C" int sed_conv3x3_wino_ablate(int mask) { g_wino_abl = mask; return 0; }
#endif
// measurement only: buf = 4 device uint64 (prologue, main loop, epilogue ticks of the 100 MHz clock summed over workgroups, workgroup
// count), accumulated by every Winograd forward / data-gradient launch until reset with NULL
extern "C" int sed_conv3x3_wino_phase_ticks(unsigned long long* buf) { g_wino_dbg = buf; return 0; }

int sed_internal_wino_launch(const float* x, const float* uq, const float* bias, float* y, float* stat, const ConvBnRed* br,
                             int rgc, int B, int Cin, int F, int T, int Cout, hipStream_t s) {      // rgc == -1: the pooling inference epilogue
    WinoGeo g;
    SED_REQUIRE(wino_geo(B, Cin, F, T, Cout, &g), "conv3x3_wino: shape B=%d Cin=%d F=%d T=%d Cout=%d is not supported (sed_conv3x3_wino_rows)", B, Cin, F, T, Cout);
    SED_REQUIRE(rgc >= -1 && rgc <= 2 && (rgc <= 0 || (br && br->x1 && br->bits && br->rgp)) && (rgc >= 0 || (!br && !stat)), "conv3x3_wino: bad epilogue arguments");
    g.dbg = g_wino_dbg;
    if (rgc > 0) {       // + the window-offset table, the input patch and the tap-sum exchange of the four waves
        g.lds += ((size_t)64 + (size_t)rgc * (F + 2) * (4 * g.TR + 2) + (size_t)4 * 9 * rgc * 32) * sizeof(float);
        SED_REQUIRE(g.lds <= 160 * 1024, "conv3x3_wino: the first block's input patch does not fit the LDS (F=%d)", F);
    }
    const dim3 grid(g.nblk * g.ncg * g.ncoh, B);
    const ConvBnRed none{};
#define WN_LAUNCH(NHv, BNRv, RGv, EVv)                                                                                          \
    do {                                                                                                                           \
        SED_TRY(wino_set_lds((conv3x3_wino_k<NHv, BNRv, RGv, EVv>), g.lds));                                                       \
        conv3x3_wino_k<NHv, BNRv, RGv, EVv><<<grid, 256, g.lds, s>>>(x, uq, bias, y, stat, B, F, T, Cout, g, br ? *br : none);     \
    } while (0)
#ifdef WN_ABLATION
    if (g_wino_abl && rgc == 0 && !br) {
        switch (g_wino_abl) {
            case 1: { SED_TRY(wino_set_lds((conv3x3_wino_k<WN_NHMAX, false, 0, false, 1>), g.lds)); conv3x3_wino_k<WN_NHMAX, false, 0, false, 1><<<grid, 256, g.lds, s>>>(x, uq, bias, y, stat, B, F, T, Cout, g, none); break; }
            case 2: { SED_TRY(wino_set_lds((conv3x3_wino_k<WN_NHMAX, false, 0, false, 2>), g.lds)); conv3x3_wino_k<WN_NHMAX, false, 0, false, 2><<<grid, 256, g.lds, s>>>(x, uq, bias, y, stat, B, F, T, Cout, g, none); break; }
            case 4: { SED_TRY(wino_set_lds((conv3x3_wino_k<WN_NHMAX, false, 0, false, 5>), g.lds)); conv3x3_wino_k<WN_NHMAX, false, 0, false, 5><<<grid, 256, g.lds, s>>>(x, uq, bias, y, stat, B, F, T, Cout, g, none); break; }
            case 8: { SED_TRY(wino_set_lds((conv3x3_wino_k<WN_NHMAX, false, 0, false, 8>), g.lds)); conv3x3_wino_k<WN_NHMAX, false, 0, false, 8><<<grid, 256, g.lds, s>>>(x, uq, bias, y, stat, B, F, T, Cout, g, none); break; }
            default: { SED_TRY(wino_set_lds((conv3x3_wino_k<WN_NHMAX, false, 0, false, 15>), g.lds)); conv3x3_wino_k<WN_NHMAX, false, 0, false, 15><<<grid, 256, g.lds, s>>>(x, uq, bias, y, stat, B, F, T, Cout, g, none); break; }
        }
        SED_LAUNCH_CHECK("conv3x3_wino (ablation)");
        return 0;
    }
#endif
    if (rgc == 1) WN_LAUNCH(WN_NHMAX, true, 1, false);
    else if (rgc == 2) WN_LAUNCH(WN_NHMAX, true, 2, false);
    else if (rgc == -1) WN_LAUNCH(WN_NHMAX, false, 0, true);
    else if (br) WN_LAUNCH(WN_NHMAX, true, 0, false);
    else WN_LAUNCH(WN_NHMAX, false, 0, false);
#undef WN_LAUNCH
    SED_LAUNCH_CHECK("conv3x3_wino");
    return 0;
}

extern "C" int sed_conv3x3_wino_rows(int B, int Cin, int F, int T, int Cout) { return sed_internal_wino_rows(B, Cin, F, T, Cout); }

extern "C" int sed_conv3x3_wino_fwd(const float* x, const float* uq, const float* bias, float* y, float* stat,
                                    int B, int Cin, int F, int T, int Cout, void* stream) {
    SED_REQUIRE(x && uq && y, "conv3x3_wino_fwd: null pointer");
    hipStream_t s = as_stream(stream);
    SedProfScope prof(SED_K_CONV_MFMA_FWD, s, 2.0 * 9.0 * Cin * Cout * (double)B * T * F);
    return sed_internal_wino_launch(x, uq, bias, y, stat, nullptr, 0, B, Cin, F, T, Cout, s);
}

extern "C" int sed_conv3x3_wino_dgrad_bnred(const float* dy, const float* ud, float* dx, float* partials, const float* pooled,
                                            const float* gamma, const float* beta, const float* conv_out_below, const float* mean,
                                            const float* rstd, float drop_p, int pool_f, int pool_t, int Fy, int Ty,
                                            int B, int C, int F, int T, int Cin, void* stream) {
    SED_REQUIRE(dy && ud && dx && partials && pooled && gamma && beta && mean && rstd, "conv3x3_wino_dgrad_bnred: null pointer");
    SED_REQUIRE(drop_p >= 0.f && drop_p < 1.f && pool_f >= 1 && pool_t >= 1, "conv3x3_wino_dgrad_bnred: bad drop_p / pool");
    SED_REQUIRE(Ty / pool_t == T && Fy / pool_f == F, "conv3x3_wino_dgrad_bnred: conv output %dx%d does not pool (%d,%d) to %dx%d", Ty, Fy, pool_f, pool_t, T, F);
    hipStream_t s = as_stream(stream);
    SedProfScope prof(SED_K_CONV_MFMA_DGRAD, s, 2.0 * 9.0 * C * Cin * (double)B * T * F);
    const ConvBnRed br{pooled, gamma, beta, conv_out_below, mean, rstd, 1.f - drop_p, 1.f / (1.f - drop_p), pool_f, pool_t, Fy, Ty,
                       nullptr, nullptr, nullptr, 0.f};
    return sed_internal_wino_launch(dy, ud, nullptr, dx, partials, &br, 0, B, C, F, T, Cin, s);
}

// sed_conv3x3_dgrad_bnred_rg in the Winograd form; rows of rg_partials = sed_conv3x3_wino_rg_rows
extern "C" int sed_conv3x3_wino_rg_rows(int B, int C, int F, int T, int Cin, int Cin1) {
    WinoGeo g;
    if (Cin1 < 1 || Cin1 > 2 || !wino_geo(B, C, F, T, Cin, &g)) return 0;
    const size_t lds = g.lds + ((size_t)64 + (size_t)Cin1 * (F + 2) * (4 * g.TR + 2) + (size_t)4 * 9 * Cin1 * 32) * sizeof(float);
    return lds <= 160 * 1024 ? B * g.nblk * g.ncg : 0;
}
extern "C" int sed_conv3x3_wino_dgrad_bnred_rg(const float* dy, const float* ud, float* dx, float* partials, const float* pooled,
                                               const float* gamma, const float* beta, const float* mean, const float* rstd, float drop_p,
                                               const float* x1, int Cin1, const unsigned char* argmax_bits, float* rg_partials,
                                               int B, int C, int F, int T, int Cin, void* stream) {
    SED_REQUIRE(dy && ud && dx && partials && pooled && gamma && beta && mean && rstd && x1 && argmax_bits && rg_partials, "conv3x3_wino_dgrad_bnred_rg: null pointer");
    SED_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (Cin1 == 1 || Cin1 == 2), "conv3x3_wino_dgrad_bnred_rg: bad drop_p / input channels");
    hipStream_t s = as_stream(stream);
    SedProfScope prof(SED_K_CONV_MFMA_DGRAD, s, 2.0 * 9.0 * C * Cin * (double)B * T * F);
    const ConvBnRed br{pooled, gamma, beta, nullptr, mean, rstd, 1.f - drop_p, 1.f / (1.f - drop_p), 1, 2, F, 2 * T,
                       x1, argmax_bits, rg_partials, 0.f};
    return sed_internal_wino_launch(dy, ud, nullptr, dx, partials, &br, Cin1, B, C, F, T, Cin, s);
}

// inference: conv + BatchNorm (running statistics, folded by sed_conv3x3_wino_pack_weights_bn_folded) + ReLU + (1,2) time pool
extern "C" int sed_conv3x3_wino_bn_relu_pool_eval(const float* x, const float* uf_folded, const float* bias_folded, float* pooled,
                                                  int B, int Cin, int F, int T, int Cout, void* stream) {
    SED_REQUIRE(x && uf_folded && bias_folded && pooled, "conv3x3_wino_bn_relu_pool_eval: null pointer");
    hipStream_t s = as_stream(stream);
    SedProfScope prof(SED_K_CONV_MFMA_FWD, s, 2.0 * 9.0 * Cin * Cout * (double)B * T * F);
    return sed_internal_wino_launch(x, uf_folded, bias_folded, pooled, nullptr, nullptr, -1, B, Cin, F, T, Cout, s);
}
