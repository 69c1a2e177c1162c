// net.hip — the whole-network plan: TimePooledCRNN.forward (reference sed.py:105-112,
// crnn_lightning.py:66-73) and its backward as a fixed sequence of launches on one stream.
//
// No host synchronisation, no allocation: every intermediate lives in the caller's workspace at
// offsets that are a pure function of the config, so a step is capturable in a hipGraph and the
// backward can be issued in stages (head+GRU, conv3, conv2, conv1) to overlap the RCCL all-reduce.
#include <string.h>
#include "common.h"

namespace {

struct ConvL { int Cin, C, T, F, Tp, Fp, pf, pt, rows, bn_rows, nchw, fused; float drop;
               int red_rows;      // > 0: this block's BatchNorm-backward sums come out of the data gradient of the block above (that many partial rows)
               int rg_rows;       // > 0 (block 0 only): so do its weight-gradient sums (sed_conv3x3_dgrad_bnred_rg), into c1_ws
               int rgrad;         // recomputed first block: its weight gradient comes from the pooled output, arg-max bits and input moments
               int wino, wino_d;  // forward / data gradient of this block run as Winograd F(2x2,3x3) (wino.hip)
               int ev; };         // inference plan only: BatchNorm folded into the packed weights, ReLU + (1,2) pool in the conv epilogue
                                  // (sed_conv3x3_bn_relu_pool_eval): the block's un-pooled output is never written
struct GruL { int in, H; };

struct Layout {
    int n_conv, n_gru, n_dense, M, Tp, Fp, feat;
    ConvL cv[SED_MAX_CONV];
    GruL gr[SED_MAX_GRU];
    int dK[SED_MAX_DENSE], dN[SED_MAX_DENSE];
    // float offsets into the workspace
    size_t wp_f[SED_MAX_CONV], wp_d[SED_MAX_CONV], conv_out[SED_MAX_CONV], stat[SED_MAX_CONV];
    size_t mean[SED_MAX_CONV], rstd[SED_MAX_CONV], scale[SED_MAX_CONV], shift[SED_MAX_CONV];
    size_t pooled[SED_MAX_CONV], bn_sums[SED_MAX_CONV], c1_stat_ws, c1_bits, c1_mom;
    size_t bias_f[SED_MAX_CONV];   // inference: the BatchNorm-folded conv bias of a block whose epilogue pools
    size_t wih_perm;               // inference, last block pooled in its conv epilogue (output stays channels-last): weight_ih_l0 of both
                                   // directions [6H][F'*C] with the columns re-ordered from c*F'+f (sed.py:108-110) to f*C+c; 0 = not used
    size_t gi[SED_MAX_GRU], gout[SED_MAX_GRU], saved[SED_MAX_GRU], gru_ws;
    size_t act[SED_MAX_DENSE];
    // backward only
    size_t wgrad_zrow;       // floats at the start of wgrad_ws / wgrad_ws_aux that the backward keeps zero (sed_conv3x3_wgrad_zero_row_bytes)
    size_t wg_arrive;        // one granule right in front of wgrad_ws (cleared by the same memset as its zero row): word 0 = arrival
                             // counter of the first deferred weight gradient's workgroups (sed_internal_stream_gate)
    size_t bn_part, sum_g, sum_gx, dbias_part, dconv[SED_MAX_CONV], gradA, wgrad_ws, wgrad_ws_aux, c1_ws, dgi[SED_MAX_GRU], dgh[SED_MAX_GRU], gru_bws, dgout[SED_MAX_GRU];
    size_t dact[SED_MAX_DENSE], lin_ws, gemm_ws, gemm_ws_aux;
    size_t total;     // floats
};

struct Carver {
    size_t off = 0;
    size_t take(size_t n) { size_t o = off; off += (n + 63) & ~(size_t)63; return o; }   // 256-byte granules
};

int build_layout(const sed_net_cfg* c, int training, Layout* L) {
    SED_REQUIRE(c, "net: null config");
    SED_REQUIRE(c->B > 0 && c->Cin > 0 && c->F > 0 && c->T > 0, "net: bad input shape B=%d Cin=%d F=%d T=%d", c->B, c->Cin, c->F, c->T);
    SED_REQUIRE(c->n_conv >= 1 && c->n_conv <= SED_MAX_CONV, "net: n_conv=%d out of range", c->n_conv);
    SED_REQUIRE(c->n_gru >= 1 && c->n_gru <= SED_MAX_GRU, "net: n_gru=%d out of range", c->n_gru);
    SED_REQUIRE(c->n_dense >= 1 && c->n_dense <= SED_MAX_DENSE, "net: n_dense=%d out of range", c->n_dense);
    SED_REQUIRE(c->conv_mode == 0 || c->conv_mode == 1, "net: conv_mode=%d (0 = exact fp32, 1 = bf16x3 experiment)", c->conv_mode);
    SED_REQUIRE((c->flags & ~(SED_NET_AUX_FIRST | SED_NET_NO_GATE | SED_NET_DIRECT_CONV)) == 0, "net: unknown flags 0x%x", c->flags);
    memset(L, 0, sizeof(*L));
    L->n_conv = c->n_conv; L->n_gru = c->n_gru; L->n_dense = c->n_dense;
    Carver cv;
    int Cin = c->Cin, T = c->T, F = c->F;
    size_t max_pool = 0, max_wgrad = 0, c1_ws = 64, c1_stat_ws = 64;
    int max_bn_rows = 0, maxC = 0;
    for (int l = 0; l < c->n_conv; ++l) {
        ConvL& q = L->cv[l];
        q.Cin = Cin; q.C = c->C[l]; q.T = T; q.F = F; q.pf = c->pool_f[l]; q.pt = c->pool_t[l];
        q.drop = c->drop_p[l]; q.nchw = (l == 0);
        SED_REQUIRE(q.C > 0 && q.C % 4 == 0, "net: conv channels C[%d]=%d must be a positive multiple of 4", l, q.C);
        SED_REQUIRE(q.pf >= 1 && q.pt >= 1 && T / q.pt >= 1 && F / q.pf >= 1,       // floor pooling, like nn.MaxPool2d
                    "net: block %d: T=%d/F=%d smaller than the pool (%d,%d)", l, T, F, q.pt, q.pf);
        SED_REQUIRE(q.drop >= 0.f && q.drop < 1.f, "net: drop_p[%d]=%f out of [0,1)", l, q.drop);
        q.Tp = T / q.pt; q.Fp = F / q.pf;
        // block 1 with <= 2 input channels: the conv output is recomputed in every pass and never stored (conv1.hip)
        q.fused = (l == 0 && c->n_conv > 1 && sed_conv1_fused_supported(q.Cin, q.F, q.T, q.C, q.pf, q.pt)) ? 1 : 0;
        // 3 or 4 input channels (config 5): recomputed as well when the whole moment-based chain is available — statistics from
        // the blocked moment kernel, (1,2) pool, and the data gradient of block 1 on the MFMA path (it forms this block's
        // BatchNorm-backward sums; the recomputing reduce / apply passes do not exist beyond 2 channels)
        if (l == 0 && !q.fused && c->n_conv > 1 && q.Cin <= 4 && c->conv_mode == 0 &&
            sed_conv1_rgrad_supported(q.Cin, q.F, q.T, q.C, q.pf, q.pt) &&
            sed_conv3x3_dgrad_bnred_rows(c->B, c->C[1], q.F / q.pf, q.T / q.pt, q.C) > 0)
            q.fused = 1;
        // 128 input channels, exact fp32: the Winograd form (2.25x fewer MFMAs, same epilogues) unless the caller asks for the direct kernels
        const bool wino_ok = l > 0 && c->conv_mode == 0 && !(c->flags & SED_NET_DIRECT_CONV);
        q.wino = (wino_ok && sed_conv3x3_wino_rows(c->B, q.Cin, q.F, q.T, q.C) > 0) ? 1 : 0;
        q.wino_d = (wino_ok && training && sed_conv3x3_wino_rows(c->B, q.C, q.F, q.T, q.Cin) > 0) ? 1 : 0;
        if (q.fused) {
            q.rows = 1;                                        // statistics from the input moments: one partial row
            q.bn_rows = sed_conv1_fused_rows(c->B, q.T);
        } else {
            q.rows = q.wino ? sed_conv3x3_wino_rows(c->B, q.Cin, q.F, q.T, q.C) : sed_conv3x3_stat_rows(c->B, q.Cin, q.F, q.T, q.C, q.nchw);
            SED_REQUIRE(q.rows > 0, "net: conv block %d (Cin=%d C=%d F=%d) is not supported by any conv kernel", l, q.Cin, q.C, q.F);
            q.bn_rows = sed_bn_bwd_rows(c->B, q.T, q.pt);
        }
        // inference: blocks >= 1 on the 128-wide exact-fp32 MFMA tile with the reference's (1,2) pool (sed.py:90) end in the pooling
        // epilogue; everything else (and every training plan) keeps the conv output + the BatchNorm/ReLU/pool pass
        q.ev = (!training && l > 0 && c->conv_mode == 0 && q.pf == 1 && q.pt == 2 &&
                (q.wino || sed_conv3x3_bn_relu_pool_eval_supported(c->B, q.Cin, q.F, q.T, q.C))) ? 1 : 0;
        // (.wino of a pooling-epilogue block: the Winograd kernel with that epilogue; .wino_d is a training-only notion)
        size_t nout = (q.fused || q.ev) ? 64 : (size_t)c->B * q.T * q.F * q.C, npool = (size_t)c->B * q.Tp * q.Fp * q.C;
        L->wp_f[l] = cv.take(q.wino ? sed_conv3x3_wino_packed_floats(q.C, q.Cin) : (size_t)9 * q.C * q.Cin);
        L->wp_d[l] = cv.take(q.wino_d ? sed_conv3x3_wino_packed_floats(q.C, q.Cin) : (size_t)9 * q.C * q.Cin);
        L->conv_out[l] = cv.take(nout);
        L->stat[l] = cv.take((size_t)q.rows * 2 * q.C);
        L->mean[l] = cv.take(q.C); L->rstd[l] = cv.take(q.C); L->scale[l] = cv.take(q.C); L->shift[l] = cv.take(q.C);
        L->pooled[l] = cv.take(npool);
        L->bn_sums[l] = cv.take((size_t)2 * q.C);
        L->bias_f[l] = cv.take(q.C);
        if (npool > max_pool) max_pool = npool;
        if (q.fused) {
            // the (1,2)-pooled block takes the moment-based backward (sed_conv1_bwd_wgrad: nothing is recomputed)
            q.rgrad = training && sed_conv1_rgrad_supported(q.Cin, q.F, q.T, q.C, q.pf, q.pt) ? 1 : 0;
            c1_ws = sed_conv1_bwd_apply_workspace_bytes(c->B, q.Cin, q.T, q.C) / sizeof(float);
            const size_t w2 = sed_conv1_bwd_wgrad_workspace_bytes(c->B, q.Cin, q.T, q.C) / sizeof(float);
            if (w2 > c1_ws) c1_ws = w2;
            const size_t sws = sed_conv1_stats_workspace_bytes(c->B, q.Cin, q.T) / sizeof(float);
            if (sws > c1_stat_ws) c1_stat_ws = sws;
        } else {
            size_t wg = sed_conv3x3_wgrad_workspace_bytes(c->B, q.Cin, q.F, q.T, q.C) / sizeof(float);
            // the shared zero row: kept clean by the backward only when every block that shares the workspace has the same
            // one (a block without it, or with a shorter one, would put its slabs on top of the others' zero row)
            if (l > 0) {
                const size_t zr = c->conv_mode == 0 ? sed_conv3x3_wgrad_zero_row_bytes(c->B, q.Cin, q.F, q.T, q.C) / sizeof(float) : 0;
                if (l == 1) L->wgrad_zrow = zr;
                else if (zr != L->wgrad_zrow) L->wgrad_zrow = 0;
            }
            // block 0's weight gradient runs on the auxiliary stream beside the MFMA weight gradients: its own scratch
            if (l == 0 && c->n_conv > 1) c1_ws = wg + 64;
            else if (wg > max_wgrad) max_wgrad = wg;
        }
        if (q.bn_rows > max_bn_rows) max_bn_rows = q.bn_rows;
        if (q.C > maxC) maxC = q.C;
        Cin = q.C; T = q.Tp; F = q.Fp;
    }
    // exact-fp32 MFMA blocks: the data gradient of block l forms the BatchNorm-backward sums of block l-1 in its epilogue
    for (int l = 1; l < c->n_conv; ++l) {
        const ConvL& q = L->cv[l];
        const int rr = (c->conv_mode != 0) ? 0 : q.wino_d ? sed_conv3x3_wino_rows(c->B, q.C, q.F, q.T, q.Cin)
                                                           : sed_conv3x3_dgrad_bnred_rows(c->B, q.C, q.F, q.T, q.Cin);
        L->cv[l - 1].red_rows = rr;
        if (rr > max_bn_rows) max_bn_rows = rr;
        // ... and, for the recomputed first block with 1 or 2 input channels, its weight-gradient sums too (1 + 9 Cin floats per
        // channel and workgroup)
        if (l == 1 && rr > 0 && L->cv[0].fused && L->cv[0].rgrad && L->cv[0].Cin <= 2) {
            const int gr = q.wino_d ? sed_conv3x3_wino_rg_rows(c->B, q.C, q.F, q.T, q.Cin, L->cv[0].Cin)
                                    : sed_conv3x3_dgrad_bnred_rg_rows(c->B, q.C, q.F, q.T, q.Cin, L->cv[0].Cin);
            L->cv[0].rg_rows = gr;
            const size_t need = (size_t)gr * q.Cin * (1 + 9 * L->cv[0].Cin);
            if (need > c1_ws) c1_ws = need;
        }
    }
    L->c1_stat_ws = cv.take(c1_stat_ws);
    L->c1_bits = L->cv[0].rgrad ? cv.take(((size_t)c->B * L->cv[0].Tp * L->cv[0].Fp * (L->cv[0].C / 4) + 3) / 4) : 0;      // one byte per channel quad
    L->c1_mom = L->cv[0].rgrad ? cv.take(2 * sed_conv1_moments_doubles(L->cv[0].Cin)) : 0;                               // doubles (the carver's 256-byte granules keep them aligned)
    L->Tp = T; L->Fp = F; L->feat = Cin * F; L->M = c->B * T;
    const size_t M = (size_t)L->M;
    // (a GRU input row of up to 64 KB: the re-ordering goes through LDS one row at a time; wider rows keep the un-fused last block)
    if (L->cv[c->n_conv - 1].ev && (size_t)L->feat * sizeof(float) > 64 * 1024) {
        ConvL& q = L->cv[c->n_conv - 1];
        q.ev = 0;
        L->conv_out[c->n_conv - 1] = cv.take((size_t)c->B * q.T * q.F * q.C);
    }
    L->wih_perm = (L->cv[c->n_conv - 1].ev && c->H[0] > 0) ? cv.take((size_t)6 * c->H[0] * L->feat) : 0;
    int in = L->feat, maxH = 0, max2H = 0;
    for (int i = 0; i < c->n_gru; ++i) {
        int H = c->H[i];
        SED_REQUIRE(H > 0 && H % 4 == 0 && 3 * H <= 1024, "net: GRU hidden H[%d]=%d must be a multiple of 4 and <= 341", i, H);
        L->gr[i].in = in; L->gr[i].H = H;
        L->gi[i] = cv.take(M * 6 * H);
        L->gout[i] = cv.take(M * 2 * H);
        L->saved[i] = training ? cv.take(M * 10 * H) : 0;
        if (H > maxH) maxH = H;
        in = 2 * H;
        if (in > max2H) max2H = in;
    }
    L->gru_ws = cv.take((size_t)6 * maxH * maxH);
    size_t max_lin_ws = 0;
    for (int j = 0; j < c->n_dense; ++j) {
        SED_REQUIRE(c->D[j] > 0, "net: dense size D[%d]=%d", j, c->D[j]);
        L->dK[j] = in; L->dN[j] = c->D[j];
        L->act[j] = (j < c->n_dense - 1) ? cv.take(M * c->D[j]) : 0;
        size_t w = sed_linear_bwd_workspace_bytes(L->M, in, c->D[j]) / sizeof(float);
        if (w > max_lin_ws) max_lin_ws = w;
        in = c->D[j];
    }
    if (training) {
        L->bn_part = cv.take((size_t)max_bn_rows * 2 * maxC);
        L->sum_g = cv.take((size_t)2 * maxC); L->sum_gx = 0;     // sum_gx = sum_g + C of the block (one all-reduce region)
        L->dbias_part = cv.take((size_t)max_bn_rows * maxC);
        // one gradient buffer per conv block: the weight gradient of block l (auxiliary stream) may still read dconv[l]
        // while the main stream already writes dconv[l-1]
        for (int l = 0; l < c->n_conv; ++l)
            L->dconv[l] = L->cv[l].fused ? 0 : cv.take((size_t)c->B * L->cv[l].T * L->cv[l].F * L->cv[l].C);
        size_t ga = max_pool > M * (size_t)L->feat ? max_pool : M * (size_t)L->feat;
        L->gradA = cv.take(ga);
        L->wg_arrive = cv.take(64);
        L->wgrad_ws = cv.take(max_wgrad + 64);               // (contiguous with wg_arrive: the carver hands out whole granules)
        L->wgrad_ws_aux = cv.take(max_wgrad + 64);
        L->c1_ws = cv.take(c1_ws);
        // gate gradients per GRU layer: the weight-gradient GEMMs of layer i (auxiliary stream) still read them while the
        // recurrence of layer i-1 writes its own on the main stream
        for (int i = 0; i < c->n_gru; ++i) {
            L->dgi[i] = cv.take(M * 6 * c->H[i]);
            L->dgh[i] = cv.take(M * 6 * c->H[i]);
        }
        L->gru_bws = cv.take(sed_gru_seq_bwd_workspace_bytes(c->B, maxH) / sizeof(float));
        for (int i = 0; i < c->n_gru; ++i) L->dgout[i] = cv.take(M * 2 * c->H[i]);
        for (int j = 0; j < c->n_dense - 1; ++j) L->dact[j] = cv.take(M * c->D[j]);
        L->lin_ws = cv.take(max_lin_ws);
    }
    {   // split-K scratch of the GEMMs: forward input projections (small batches), and in training the weight gradients
        size_t gw = 0;
        for (int i = 0; i < c->n_gru; ++i) {
            size_t f = sed_gemm_f32_workspace_bytes(L->M, 6 * c->H[i], L->gr[i].in) / sizeof(float);
            if (f > gw) gw = f;
            if (training) {
                size_t a = sed_gemm_f32_workspace_bytes(3 * c->H[i], c->H[i], L->M) / sizeof(float);
                size_t b = sed_gemm_f32_workspace_bytes(6 * c->H[i], L->gr[i].in, L->M) / sizeof(float);
                if (a > gw) gw = a;
                if (b > gw) gw = b;
            }
        }
        L->gemm_ws = cv.take(gw + 64);
        L->gemm_ws_aux = training ? cv.take(gw + 64) : 0;          // split-K scratch of the GEMMs issued on the auxiliary stream
    }
    L->total = cv.off;
    return 0;
}

inline uint64_t layer_seed(uint64_t seed, int l) { return seed + 0x9E3779B97F4A7C15ull * (uint64_t)(l + 1); }

}  // namespace

extern "C" int sed_net_out_shape(const sed_net_cfg* cfg, int* Tp, int* Fp) {
    Layout L;
    SED_TRY(build_layout(cfg, 0, &L));
    if (Tp) *Tp = L.Tp;
    if (Fp) *Fp = L.Fp;
    return 0;
}

extern "C" size_t sed_net_workspace_bytes(const sed_net_cfg* cfg, int training) {
    Layout L;
    if (build_layout(cfg, training, &L) != 0) return 0;
    return L.total * sizeof(float);
}

// phases [pb, pe): 2l = conv block l (+ statistic sums), 2l+1 = finalise + normalise/pool, 2*n_conv = GRU + head
static int forward_impl(const sed_net_cfg* c, const sed_net_params* p, const float* x, float* logits, void* workspace,
                        int training, uint64_t seed, const uint64_t* seed_dev, int pb, int pe, float count_scale, void* stream) {
    SED_REQUIRE(p && x && logits && workspace, "net_forward: null pointer");
    Layout L;
    SED_TRY(build_layout(c, training, &L));
    SED_REQUIRE(pb >= 0 && pe <= 2 * L.n_conv + 1 && pb < pe, "net_forward: bad phase range [%d,%d)", pb, pe);
    float* ws = (float*)workspace;
    const int B = c->B;
    // a call that starts at phase 0 packs the weights of EVERY conv layer in one launch (exact-fp32 layouts), instead of one
    // 5 us launch per layer standing between the layers of the critical chain
    // inference (no phases, exact fp32): ONE packing launch (fragments, BatchNorm coefficients on running statistics, folded
    // weights + bias of the pooling-epilogue blocks, re-ordered GRU input weights), then one launch per conv block
    const bool eval_plan = !training && c->conv_mode == 0 && pb == 0 && pe == 2 * L.n_conv + 1;
    const ConvL& qtop = L.cv[L.n_conv - 1];
    const bool wih_permuted = eval_plan && qtop.ev && L.wih_perm != 0;
    if (eval_plan) {
        const float* w[SED_MAX_CONV]; const float* bs[SED_MAX_CONV]; const float* gm[SED_MAX_CONV]; const float* bt[SED_MAX_CONV];
        const float* rm[SED_MAX_CONV]; const float* rv[SED_MAX_CONV];
        float* wf[SED_MAX_CONV]; float* sc[SED_MAX_CONV]; float* sh[SED_MAX_CONV]; float* bf[SED_MAX_CONV];
        int fold[SED_MAX_CONV], co[SED_MAX_CONV], ci[SED_MAX_CONV], wn[SED_MAX_CONV];
        for (int l = 0; l < L.n_conv; ++l) {
            SED_REQUIRE(p->conv_w[l] && p->conv_b[l] && p->bn_g[l] && p->bn_b[l] && p->bn_rm[l] && p->bn_rv[l],
                        "net_forward: missing parameters of conv block %d", l);
            w[l] = p->conv_w[l]; bs[l] = p->conv_b[l]; gm[l] = p->bn_g[l]; bt[l] = p->bn_b[l]; rm[l] = p->bn_rm[l]; rv[l] = p->bn_rv[l];
            wf[l] = ws + L.wp_f[l]; sc[l] = ws + L.scale[l]; sh[l] = ws + L.shift[l]; bf[l] = ws + L.bias_f[l];
            fold[l] = L.cv[l].ev; co[l] = L.cv[l].C; ci[l] = L.cv[l].Cin; wn[l] = L.cv[l].ev && L.cv[l].wino;
        }
        if (wih_permuted) SED_REQUIRE(p->gru_wih[0][0] && p->gru_wih[0][1], "net_forward: missing parameters of GRU layer 0");
        SED_TRY(sed_internal_conv_pack_eval(L.n_conv, w, bs, gm, bt, rm, rv, c->bn_eps, wf, sc, sh, bf, fold, wn, co, ci,
                                            wih_permuted ? p->gru_wih[0][0] : nullptr, wih_permuted ? p->gru_wih[0][1] : nullptr,
                                            ws + L.wih_perm, 3 * L.gr[0].H, qtop.C, qtop.Fp, stream));
        for (int l = 0; l < L.n_conv; ++l) {
            const ConvL& q = L.cv[l];
            const float* in = (l == 0) ? x : ws + L.pooled[l - 1];
            const int last = (l == L.n_conv - 1);
            if (q.fused) {
                SED_TRY(sed_conv1_bn_relu_pool_drop_fwd(in, ws + L.wp_f[l], p->conv_b[l], ws + L.scale[l], ws + L.shift[l], ws + L.pooled[l],
                                                        B, q.Cin, q.F, q.T, q.C, q.pf, q.pt, 0.f, 0, nullptr, nullptr, stream));
            } else if (q.ev && q.wino) {
                SED_TRY(sed_conv3x3_wino_bn_relu_pool_eval(in, ws + L.wp_f[l], ws + L.bias_f[l], ws + L.pooled[l], B, q.Cin, q.F, q.T, q.C, stream));
            } else if (q.ev) {          // (the last block's output stays channels-last: the GRU projection reads re-ordered weights)
                SED_TRY(sed_conv3x3_bn_relu_pool_eval(in, ws + L.wp_f[l], ws + L.bias_f[l], ws + L.pooled[l], B, q.Cin, q.F, q.T, q.C, stream));
            } else {
                SED_TRY(sed_conv3x3_fwd_ex(in, q.nchw, ws + L.wp_f[l], p->conv_b[l], ws + L.conv_out[l], nullptr, B, q.Cin, q.F, q.T, q.C, 0, stream));
                SED_TRY(sed_bn_relu_pool_drop_fwd(ws + L.conv_out[l], ws + L.scale[l], ws + L.shift[l], ws + L.pooled[l], B,
                                                  q.T, q.F, q.C, q.pf, q.pt, last, 0.f, 0, nullptr, stream));
            }
        }
    }
    const bool packed_up_front = !eval_plan && pb == 0 && c->conv_mode == 0 && L.n_conv > 1;
    if (packed_up_front) {
        const float* w[SED_MAX_CONV]; float* wf[SED_MAX_CONV]; float* wd[SED_MAX_CONV]; int co[SED_MAX_CONV], ci[SED_MAX_CONV];
        int wnf[SED_MAX_CONV], wnd[SED_MAX_CONV];
        for (int l = 0; l < L.n_conv; ++l) {
            SED_REQUIRE(p->conv_w[l], "net_forward: missing parameters of conv block %d", l);
            w[l] = p->conv_w[l]; wf[l] = ws + L.wp_f[l]; wd[l] = (training && l > 0) ? ws + L.wp_d[l] : nullptr;
            co[l] = L.cv[l].C; ci[l] = L.cv[l].Cin; wnf[l] = L.cv[l].wino; wnd[l] = L.cv[l].wino_d;
        }
        SED_TRY(sed_internal_conv_pack_multi(L.n_conv, w, wf, wd, co, ci, wnf, wnd, stream));
    }
    for (int l = 0; l < L.n_conv && !eval_plan; ++l) {
        const ConvL& q = L.cv[l];
        const float* in = (l == 0) ? x : ws + L.pooled[l - 1];
        const bool do_a = pb <= 2 * l && 2 * l < pe, do_b = pb <= 2 * l + 1 && 2 * l + 1 < pe;
        if (!do_a && !do_b) continue;
        const bool one_shot = do_a && do_b && count_scale == 1.f;      // unsynchronised: the fused finalise kernel
        SED_REQUIRE(p->conv_w[l] && p->conv_b[l] && p->bn_g[l] && p->bn_b[l] && p->bn_rm[l] && p->bn_rv[l],
                    "net_forward: missing parameters of conv block %d", l);
        if (do_a) {
            if (!packed_up_front) {
                float* const wdp = (training && l > 0) ? ws + L.wp_d[l] : nullptr;
                if (q.wino || (q.wino_d && wdp))
                    SED_TRY(sed_conv3x3_wino_pack_weights(p->conv_w[l], q.wino ? ws + L.wp_f[l] : nullptr, q.wino_d ? wdp : nullptr, q.C, q.Cin, stream));
                if (!q.wino || (wdp && !q.wino_d))
                    SED_TRY(sed_conv3x3_pack_weights_ex(p->conv_w[l], q.wino ? nullptr : ws + L.wp_f[l], q.wino_d ? nullptr : wdp,
                                                        q.C, q.Cin, (l > 0) ? c->conv_mode : 0, stream));
            }
            if (q.fused) {
                if (training) SED_TRY(sed_conv1_stats(in, ws + L.wp_f[l], p->conv_b[l], ws + L.stat[l], ws + L.c1_stat_ws, B, q.Cin, q.F, q.T, q.C,
                                                      q.rgrad ? (double*)(ws + L.c1_mom) : nullptr, stream));
            } else if (q.wino) {
                SED_TRY(sed_conv3x3_wino_fwd(in, ws + L.wp_f[l], p->conv_b[l], ws + L.conv_out[l], training ? ws + L.stat[l] : nullptr,
                                             B, q.Cin, q.F, q.T, q.C, stream));
            } else {
                SED_TRY(sed_conv3x3_fwd_ex(in, q.nchw, ws + L.wp_f[l], p->conv_b[l], ws + L.conv_out[l],
                                           training ? ws + L.stat[l] : nullptr, B, q.Cin, q.F, q.T, q.C, (l > 0) ? c->conv_mode : 0, stream));
            }
            if (training && !one_shot) SED_TRY(sed_bn_stat_sums(ws + L.stat[l], q.rows, q.C, ws + L.bn_sums[l], stream));
        }
        if (!do_b) continue;
        const double count = (double)B * q.T * q.F;
        if (training && one_shot)
            SED_TRY(sed_bn_finalize_train(ws + L.stat[l], q.rows, q.C, count, p->bn_g[l], p->bn_b[l],
                                          p->bn_rm[l], p->bn_rv[l], c->bn_momentum, c->bn_eps, ws + L.mean[l],
                                          ws + L.rstd[l], ws + L.scale[l], ws + L.shift[l], stream));
        else if (training)
            SED_TRY(sed_bn_finalize_from_sums(ws + L.bn_sums[l], q.C, count * (double)count_scale, p->bn_g[l], p->bn_b[l],
                                              p->bn_rm[l], p->bn_rv[l], c->bn_momentum, c->bn_eps, ws + L.mean[l],
                                              ws + L.rstd[l], ws + L.scale[l], ws + L.shift[l], stream));
        else
            SED_TRY(sed_bn_finalize_eval(p->bn_g[l], p->bn_b[l], p->bn_rm[l], p->bn_rv[l], c->bn_eps, q.C,
                                         ws + L.scale[l], ws + L.shift[l], stream));
        const int last = (l == L.n_conv - 1);
        if (q.fused)
            SED_TRY(sed_conv1_bn_relu_pool_drop_fwd(in, ws + L.wp_f[l], p->conv_b[l], ws + L.scale[l], ws + L.shift[l],
                                                    ws + L.pooled[l], B, q.Cin, q.F, q.T, q.C, q.pf, q.pt,
                                                    training ? q.drop : 0.f, layer_seed(seed, l), seed_dev,
                                                    (training && q.rgrad) ? (unsigned char*)(ws + L.c1_bits) : nullptr, stream));
        else
            SED_TRY(sed_bn_relu_pool_drop_fwd(ws + L.conv_out[l], ws + L.scale[l], ws + L.shift[l], ws + L.pooled[l], B,
                                              q.T, q.F, q.C, q.pf, q.pt, last, training ? q.drop : 0.f,
                                              layer_seed(seed, l), seed_dev, stream));
    }
    if (pe <= 2 * L.n_conv) return 0;
    const int M = L.M;
    const float* gin = ws + L.pooled[L.n_conv - 1];          // [M][C*F'] in the reference feature order
    for (int i = 0; i < L.n_gru; ++i) {
        const int H = L.gr[i].H, K = L.gr[i].in;
        for (int d = 0; d < 2; ++d)
            SED_REQUIRE(p->gru_wih[i][d] && p->gru_whh[i][d] && p->gru_bih[i][d] && p->gru_bhh[i][d],
                        "net_forward: missing parameters of GRU layer %d dir %d", i, d);
        // both directions in ONE GEMM (N = 6H) when their weights/biases are adjacent (the flat arena lays them so)
        const float* wih[2] = {p->gru_wih[i][0], p->gru_wih[i][1]};
        if (i == 0 && wih_permuted) { wih[0] = ws + L.wih_perm; wih[1] = wih[0] + (size_t)3 * H * K; }      // channels-last feature columns
        const bool fused = wih[1] == wih[0] + (size_t)3 * H * K && p->gru_bih[i][1] == p->gru_bih[i][0] + 3 * H;
        if (fused) {
            SED_TRY(sed_gemm_f32_ws(gin, K, 1, wih[0], 1, K, ws + L.gi[i], 6 * H, p->gru_bih[i][0], M, 6 * H, K, ws + L.gemm_ws, stream));
        } else {
            for (int d = 0; d < 2; ++d)
                SED_TRY(sed_gemm_f32(gin, K, 1, wih[d], 1, K, ws + L.gi[i] + d * 3 * H, 6 * H,
                                     p->gru_bih[i][d], 0.f, M, 3 * H, K, stream));
        }
        const float* whh[2] = {p->gru_whh[i][0], p->gru_whh[i][1]};
        const float* bhh[2] = {p->gru_bhh[i][0], p->gru_bhh[i][1]};
        SED_TRY(sed_gru_seq_fwd(ws + L.gi[i], whh, bhh, ws + L.gout[i], training ? ws + L.saved[i] : nullptr,
                                ws + L.gru_ws, B, L.Tp, H, stream));
        gin = ws + L.gout[i];
    }
    const float* din = gin;
    for (int j = 0; j < L.n_dense; ++j) {
        SED_REQUIRE(p->dense_w[j] && p->dense_b[j], "net_forward: missing parameters of dense layer %d", j);
        const int last = (j == L.n_dense - 1);
        float* y = last ? logits : ws + L.act[j];
        SED_TRY(sed_linear_fwd(din, p->dense_w[j], p->dense_b[j], y, M, L.dK[j], L.dN[j], !last, stream));
        din = y;
    }
    return 0;
}

extern "C" int sed_net_forward(const sed_net_cfg* c, const sed_net_params* p, const float* x, float* logits,
                               void* workspace, int training, uint64_t seed, const uint64_t* seed_dev, void* stream) {
    return forward_impl(c, p, x, logits, workspace, training, seed, seed_dev, 0, 2 * (c ? c->n_conv : 0) + 1, 1.f, stream);
}

extern "C" int sed_net_forward_phases(const sed_net_cfg* c, const sed_net_params* p, const float* x, float* logits,
                                      void* workspace, int training, uint64_t seed, int phase_begin, int phase_end,
                                      float count_scale, void* stream) {
    SED_REQUIRE(count_scale >= 1.f, "net_forward_phases: count_scale must be >= 1");
    return forward_impl(c, p, x, logits, workspace, training, seed, nullptr, phase_begin, phase_end, count_scale, stream);
}

extern "C" int sed_net_sync_region(const sed_net_cfg* c, int backward, int block, size_t* offset_bytes, size_t* n_floats) {
    Layout L;
    SED_TRY(build_layout(c, 1, &L));
    SED_REQUIRE(block >= 0 && block < L.n_conv && offset_bytes && n_floats, "net_sync_region: bad arguments");
    *offset_bytes = (backward ? L.sum_g : L.bn_sums[block]) * sizeof(float);
    *n_floats = (size_t)2 * L.cv[block].C;
    return 0;
}

// Where an intermediate of the plan lives in the caller's workspace (a pure function of the config): lets a host read
// activations (feature maps for inspection) and lets the parity tests compare intermediates, not only end results.
extern "C" int sed_net_workspace_region(const sed_net_cfg* c, int training, const char* name, int index,
                                        size_t* offset_bytes, size_t* n_floats) {
    Layout L;
    SED_TRY(build_layout(c, training, &L));
    SED_REQUIRE(name && offset_bytes && n_floats, "net_workspace_region: null pointer");
    const bool conv_idx = index >= 0 && index < L.n_conv, gru_idx = index >= 0 && index < L.n_gru;
    size_t off = 0, n = 0;
    bool ok = false;
    auto is = [&](const char* q) { return strcmp(name, q) == 0; };
    if (conv_idx) {
        const ConvL& q = L.cv[index];
        const size_t nout = (size_t)c->B * q.T * q.F * q.C, npool = (size_t)c->B * q.Tp * q.Fp * q.C;
        if (is("conv_out") && !q.fused && !q.ev) { off = L.conv_out[index]; n = nout; ok = true; }
        else if (is("pooled")) { off = L.pooled[index]; n = npool; ok = true; }
        else if (is("mean")) { off = L.mean[index]; n = q.C; ok = true; }
        else if (is("rstd")) { off = L.rstd[index]; n = q.C; ok = true; }
        else if (is("scale")) { off = L.scale[index]; n = q.C; ok = true; }
        else if (is("shift")) { off = L.shift[index]; n = q.C; ok = true; }
        else if (is("dconv") && training && !q.fused) { off = L.dconv[index]; n = nout; ok = true; }
    }
    if (!ok && gru_idx) {
        const size_t M = (size_t)L.M, H = L.gr[index].H;
        if (is("gi")) { off = L.gi[index]; n = M * 6 * H; ok = true; }
        else if (is("gru_out")) { off = L.gout[index]; n = M * 2 * H; ok = true; }
        else if (is("dgru_out") && training) { off = L.dgout[index]; n = M * 2 * H; ok = true; }
    }
    if (!ok && training && index == 0) {
        if (is("grad_act")) {           // the gradient w.r.t. the pooled output currently being back-propagated (reused per block)
            off = L.gradA;
            size_t mp = 0;
            for (int l = 0; l < L.n_conv; ++l) {
                const size_t np = (size_t)c->B * L.cv[l].Tp * L.cv[l].Fp * L.cv[l].C;
                if (np > mp) mp = np;
            }
            n = mp > (size_t)L.M * L.feat ? mp : (size_t)L.M * L.feat;
            ok = true;
        } else if (is("bn_sums_bwd")) { off = L.sum_g; n = 2 * (size_t)L.cv[0].C; ok = true; }
    }
    if (!ok && training && (index == 0 || index == 1) && is("wgrad_zero_row") && L.wgrad_zrow > 0) {
        off = index ? L.wgrad_ws_aux : L.wgrad_ws; n = L.wgrad_zrow; ok = true;      // the rows the backward keeps zero (main / auxiliary scratch)
    }
    SED_REQUIRE(ok, "net_workspace_region: no region '%s'[%d] in this plan (training=%d)", name, index, training);
    *offset_bytes = off * sizeof(float);
    *n_floats = n;
    return 0;
}

// The ReLU-gate / pooling arg-max decisions of conv block `block` in the LAST training forward that ran on this workspace, as
// the backward kernels make them (sed_bn_relu_pool_route on the stored conv output, sed_conv1_route for a recomputed first
// block).  route: caller-owned [B][T_l/pt][F_l/pf][C] bytes.  For parity tests: an oracle that routes its gradients with
// THESE decisions differs from the plan by rounding only, however many near-ties the batch holds.
extern "C" int sed_net_routing(const sed_net_cfg* c, const sed_net_params* p, const float* x, const void* workspace, int block,
                               unsigned char* route, void* stream) {
    SED_REQUIRE(p && x && workspace && route, "net_routing: null pointer");
    Layout L;
    SED_TRY(build_layout(c, 1, &L));
    SED_REQUIRE(block >= 0 && block < L.n_conv, "net_routing: block %d out of range", block);
    const float* ws = (const float*)workspace;
    const ConvL& q = L.cv[block];
    if (q.fused) {
        SED_REQUIRE(p->conv_b[block], "net_routing: missing parameters of conv block %d", block);
        return sed_conv1_route(x, ws + L.wp_f[block], p->conv_b[block], ws + L.scale[block], ws + L.shift[block], route,
                               c->B, q.Cin, q.F, q.T, q.C, q.pf, q.pt, stream);
    }
    return sed_bn_relu_pool_route(ws + L.conv_out[block], ws + L.scale[block], ws + L.shift[block], route,
                                  c->B, q.T, q.F, q.C, q.pf, q.pt, stream);
}

// BatchNorm/ReLU/pool/dropout backward of block l on stream `st` (for the fused first block: everything of block 0).
// part 1 = reduction pass (sum g, sum g*xhat -> ws.sum_g[0..2C), dgamma, dbeta), part 2 = apply pass (dconv[l] + conv-bias
// gradient; fused block: + weight gradient), 3 = both.  count_scale > 1: the sums were all-reduced over that many ranks.
static int bn_backward(const Layout& L, const sed_net_cfg* c, const sed_net_params* p, const sed_net_params* g,
                       const float* x, float* ws, uint64_t seed, const uint64_t* seed_dev, int l, int part, float count_scale, void* st) {
    const ConvL& q = L.cv[l];
    const int B = c->B, last = (l == L.n_conv - 1);
    const uint64_t sd = layer_seed(seed, l);
    SED_REQUIRE(g->conv_w[l] && g->conv_b[l] && g->bn_g[l] && g->bn_b[l], "net_backward: missing gradient buffers of conv block %d", l);
    float* sum_g = ws + L.sum_g;
    float* sum_gx = sum_g + q.C;
    if ((part & 1) && q.red_rows > 0 && !q.fused) {   // the partial sums already lie in bn_part (epilogue of the data gradient above);
        // channels whose xhat the pooled output cannot give (|gamma| < |beta| / 64) are recomputed from the stored conv output
        SED_TRY(sed_bn_bwd_finalize_small_gamma(ws + L.bn_part, q.red_rows, q.C, sum_g, sum_gx, g->bn_g[l], g->bn_b[l], ws + L.gradA,
                                                ws + L.pooled[l], ws + L.conv_out[l], p->bn_g[l], p->bn_b[l], ws + L.mean[l], ws + L.rstd[l],
                                                ws + L.scale[l], ws + L.shift[l], B, q.T, q.F, q.pf, q.pt, q.drop, st));
    } else if ((part & 1) && q.red_rows > 0) {
        SED_TRY(sed_bn_bwd_finalize(ws + L.bn_part, q.red_rows, q.C, sum_g, sum_gx, g->bn_g[l], g->bn_b[l], st));
    } else if (part & 1) {
        if (q.fused)
            SED_TRY(sed_conv1_bwd_reduce(x, ws + L.wp_f[l], p->conv_b[l], ws + L.gradA, ws + L.scale[l], ws + L.shift[l],
                                         ws + L.mean[l], ws + L.rstd[l], ws + L.bn_part, B, q.Cin, q.F, q.T, q.C, q.pf, q.pt,
                                         q.drop, sd, seed_dev, st));
        else if (sed_bn_bwd_reduce_pooled_supported(q.F, q.C, q.pf, q.pt, last))
            // the block that feeds the GRU: the sums from its pooled output and that tensor's gradient (1/pt of the conv output)
            SED_TRY(sed_bn_bwd_reduce_pooled(ws + L.pooled[l], ws + L.gradA, p->bn_g[l], p->bn_b[l], ws + L.conv_out[l],
                                             ws + L.mean[l], ws + L.rstd[l], ws + L.scale[l], ws + L.shift[l], ws + L.bn_part, B, q.T, q.F, q.C, q.pf, q.pt, last, q.drop, st));
        else
            SED_TRY(sed_bn_relu_pool_drop_bwd_reduce(ws + L.conv_out[l], ws + L.gradA, ws + L.scale[l], ws + L.shift[l],
                                                     ws + L.mean[l], ws + L.rstd[l], ws + L.bn_part, B, q.T, q.F, q.C, q.pf,
                                                     q.pt, last, q.drop, sd, seed_dev, st));
        SED_TRY(sed_bn_bwd_finalize(ws + L.bn_part, q.bn_rows, q.C, sum_g, sum_gx, g->bn_g[l], g->bn_b[l], st));
    }
    if (part & 2) {
        if (count_scale != 1.f) SED_TRY(sed_scale(sum_g, 2 * q.C, 1.f / count_scale, st));   // global sums / global count
        if (q.fused && q.rgrad && q.rg_rows > 0) {         // the sums came out of the data gradient above: assemble only
            // (single device: sum g*xhat from the block's own R_k, exact for every gamma; synchronised BatchNorm: the all-reduced sums)
            SED_TRY(sed_conv1_bwd_wgrad_assemble(ws + L.c1_ws, q.rg_rows, (const double*)(ws + L.c1_mom), ws + L.wp_f[l], p->conv_b[l],
                                                 ws + L.mean[l], ws + L.rstd[l], ws + L.scale[l], sum_g, count_scale == 1.f ? nullptr : sum_gx, g->conv_w[l], g->conv_b[l],
                                                 B, q.Cin, q.F, q.T, q.C, p->bn_g[l], p->bn_b[l], g->bn_g[l], st));
        } else if (q.fused && q.rgrad) {
            SED_TRY(sed_conv1_bwd_wgrad(x, ws + L.gradA, ws + L.pooled[l], (const unsigned char*)(ws + L.c1_bits), (const double*)(ws + L.c1_mom),
                                        ws + L.wp_f[l], p->conv_b[l], ws + L.mean[l], ws + L.rstd[l], ws + L.scale[l], sum_g,
                                        (count_scale == 1.f && q.red_rows > 0) ? nullptr : sum_gx,
                                        g->conv_w[l], g->conv_b[l], ws + L.c1_ws, B, q.Cin, q.F, q.T, q.C, q.drop,
                                        q.red_rows > 0 ? p->bn_g[l] : nullptr, q.red_rows > 0 ? p->bn_b[l] : nullptr,
                                        q.red_rows > 0 ? g->bn_g[l] : nullptr, st));
        } else if (q.fused) {
            SED_TRY(sed_conv1_bwd_apply_wgrad(x, ws + L.wp_f[l], p->conv_b[l], ws + L.gradA, ws + L.scale[l], ws + L.shift[l],
                                              ws + L.mean[l], ws + L.rstd[l], sum_g, sum_gx, g->conv_w[l], g->conv_b[l],
                                              ws + L.c1_ws, B, q.Cin, q.F, q.T, q.C, q.pf, q.pt, q.drop, sd, seed_dev,
                                              q.red_rows > 0 ? p->bn_g[l] : nullptr, q.red_rows > 0 ? p->bn_b[l] : nullptr,
                                              q.red_rows > 0 ? g->bn_g[l] : nullptr, st));
        } else {
            SED_TRY(sed_bn_relu_pool_drop_bwd_apply(ws + L.conv_out[l], ws + L.gradA, ws + L.scale[l], ws + L.shift[l],
                                                    ws + L.mean[l], ws + L.rstd[l], sum_g, sum_gx, ws + L.dconv[l],
                                                    ws + L.dbias_part, B, q.T, q.F, q.C, q.pf, q.pt, last, q.drop, sd, seed_dev, st));
            SED_TRY(sed_reduce_rows(ws + L.dbias_part, q.bn_rows, q.C, q.C, g->conv_b[l], st));
        }
    }
    return 0;
}

// data gradient of block l >= 1 into gradA; where the shapes allow it the epilogue also writes the BatchNorm-backward partial
// sums of block l-1 into bn_part (cv[l-1].red_rows > 0), which bn_backward then only finalises
static int dgrad(const Layout& L, const sed_net_cfg* c, const sed_net_params* p, const float* x, float* ws, int l, void* st) {
    const ConvL& q = L.cv[l];
    const ConvL& u = L.cv[l - 1];
    if (u.rg_rows > 0 && q.wino_d)
        return sed_conv3x3_wino_dgrad_bnred_rg(ws + L.dconv[l], ws + L.wp_d[l], ws + L.gradA, ws + L.bn_part, ws + L.pooled[l - 1],
                                               p->bn_g[l - 1], p->bn_b[l - 1], ws + L.mean[l - 1], ws + L.rstd[l - 1], u.drop,
                                               x, u.Cin, (const unsigned char*)(ws + L.c1_bits), ws + L.c1_ws, c->B, q.C, q.F, q.T, q.Cin, st);
    if (u.rg_rows > 0)
        return sed_conv3x3_dgrad_bnred_rg(ws + L.dconv[l], ws + L.wp_d[l], ws + L.gradA, ws + L.bn_part, ws + L.pooled[l - 1],
                                          p->bn_g[l - 1], p->bn_b[l - 1], ws + L.mean[l - 1], ws + L.rstd[l - 1], u.drop,
                                          x, u.Cin, (const unsigned char*)(ws + L.c1_bits), ws + L.c1_ws, c->B, q.C, q.F, q.T, q.Cin, st);
    if (q.wino_d)
        return sed_conv3x3_wino_dgrad_bnred(ws + L.dconv[l], ws + L.wp_d[l], ws + L.gradA, ws + L.bn_part, ws + L.pooled[l - 1],
                                            p->bn_g[l - 1], p->bn_b[l - 1], u.fused ? nullptr : ws + L.conv_out[l - 1],
                                            ws + L.mean[l - 1], ws + L.rstd[l - 1], u.drop, u.pf, u.pt, u.F, u.T, c->B, q.C, q.F, q.T, q.Cin, st);
    if (u.red_rows > 0)
        return sed_conv3x3_dgrad_bnred(ws + L.dconv[l], ws + L.wp_d[l], ws + L.gradA, ws + L.bn_part, ws + L.pooled[l - 1],
                                       p->bn_g[l - 1], p->bn_b[l - 1],
                                       u.fused ? nullptr : ws + L.conv_out[l - 1],      // a recomputed first block keeps no conv output: its
                                       ws + L.mean[l - 1], ws + L.rstd[l - 1],          // small-gamma channels are finished by its own passes
                                       u.drop, u.pf, u.pt, u.F, u.T, c->B, q.C, q.F, q.T, q.Cin, st);
    return sed_conv3x3_fwd_ex(ws + L.dconv[l], 0, ws + L.wp_d[l], nullptr, ws + L.gradA, nullptr, c->B, q.C, q.F, q.T, q.Cin, c->conv_mode, st);
}

extern "C" int sed_net_backward(const sed_net_cfg* c, const sed_net_params* p, const sed_net_params* g,
                                const float* x, const float* dlogits, void* workspace, uint64_t seed,
                                const uint64_t* seed_dev, int stage_begin, int stage_end, void* stream, void* aux_stream) {
    SED_REQUIRE(p && g && x && dlogits && workspace, "net_backward: null pointer");
    // cross-stream ordering: ev_dg[l] = the input of block l's BatchNorm backward (the gradient of its pooled output) is
    // complete on the main stream, ev_bn[l] = BatchNorm backward of block l complete (auxiliary stream).  Host objects
    // created on first use, one set per calling thread and device, so concurrent callers never share an event.
    constexpr int kMaxDev = 16;
    static thread_local hipEvent_t ev_all[kMaxDev][2][SED_MAX_CONV] = {};
    // [i]: recurrence of GRU layer i done (main); [SED_MAX_GRU]: the GRU weight gradients of the aux stream done;
    // [SED_MAX_GRU + 1]: data gradient of the top conv block issued (main); [SED_MAX_GRU + 2]: its weight gradient done (aux);
    // [SED_MAX_GRU + 3]: the zero rows of the weight-gradient workspaces cleared (aux); [SED_MAX_GRU + 4]: the fork of the
    // auxiliary stream at the top of the call (main)
    static thread_local hipEvent_t ev_gru_all[kMaxDev][SED_MAX_GRU + 5] = {};
    hipStream_t s_main = as_stream(stream), s_aux = as_stream(aux_stream);
    hipEvent_t* ev_dg = nullptr;
    hipEvent_t* ev_bn = nullptr;
    hipEvent_t* ev_gru = nullptr;
    if (s_aux && s_aux != s_main) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) {
            sed_set_error("net_backward: device index %d outside [0,%d)", dev, kMaxDev);
            return SED_EINVAL;
        }
        ev_dg = ev_all[dev][0];
        ev_bn = ev_all[dev][1];
        ev_gru = ev_gru_all[dev];
        for (int i = 0; i < SED_MAX_GRU + 5; ++i)
            if (!ev_gru[i] && hipEventCreateWithFlags(&ev_gru[i], hipEventDisableTiming) != hipSuccess) {
                sed_set_error("net_backward: hipEventCreate failed");
                return SED_EINVAL;
            }
        for (int l = 0; l < SED_MAX_CONV; ++l)
            if (!ev_dg[l]) {
                if (hipEventCreateWithFlags(&ev_dg[l], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&ev_bn[l], hipEventDisableTiming) != hipSuccess) {
                    sed_set_error("net_backward: hipEventCreate failed");
                    return SED_EINVAL;
                }
            }
    } else {
        s_aux = nullptr;
    }
    Layout L;
    SED_TRY(build_layout(c, 1, &L));
    SED_REQUIRE(stage_begin >= 0 && stage_end <= L.n_conv + 1 && stage_begin < stage_end,
                "net_backward: bad stage range [%d,%d)", stage_begin, stage_end);
    float* ws = (float*)workspace;
    const int B = c->B, M = L.M;

    // The exact-fp32 weight-gradient kernel reads out-of-image rows from a zero-filled row at the start of its workspace.  Nothing
    // writes those bytes, so the backward clears them once — in the call that holds stage 0, on the auxiliary stream, where it
    // costs the critical chain nothing — and passes SED_WGRAD_ZERO_ROW_CLEAN: the memset node in front of each weight gradient
    // is gone (two per step, one of them on the chain).  The same memset clears the arrival counter in front of wgrad_ws.
    // The auxiliary stream is FORKED from the main stream first (event record + wait): under hipGraph capture that is what
    // makes it part of the capture, so the memsets are graph nodes and every replay clears the rows again (round-3 advisor:
    // issued on the un-forked stream they ran once, outside the graph, and every replay trusted bytes nobody re-cleared).
    const int wg_clean = ((L.wgrad_zrow > 0 && s_aux) ? SED_WGRAD_ZERO_ROW_CLEAN : 0) | ((c->flags & SED_NET_DIRECT_CONV) ? SED_WGRAD_DIRECT : 0);
    if (s_aux && stage_begin == 0 && L.n_conv > 1) {
        hipError_t e = hipEventRecord(ev_gru[SED_MAX_GRU + 4], s_main);
        if (e == hipSuccess) e = hipStreamWaitEvent(s_aux, ev_gru[SED_MAX_GRU + 4], 0);
        if (e == hipSuccess) e = hipMemsetAsync(ws + L.wg_arrive, 0, (64 + L.wgrad_zrow) * sizeof(float), s_aux);
        if (e == hipSuccess && L.wgrad_zrow > 0) e = hipMemsetAsync(ws + L.wgrad_ws_aux, 0, L.wgrad_zrow * sizeof(float), s_aux);
        if (e == hipSuccess) e = hipEventRecord(ev_gru[SED_MAX_GRU + 3], s_aux);
        if (e != hipSuccess) {          // fail loudly: a weight gradient that trusts an un-cleared row would be silently wrong
            sed_set_error("net_backward: clearing the weight-gradient zero rows failed: %s", hipGetErrorString(e));
            return (int)e;
        }
    }

    if (stage_begin == 0) {
        // ── dense head ──
        const float* gru_last = ws + L.gout[L.n_gru - 1];
        for (int j = L.n_dense - 1; j >= 0; --j) {
            const int last = (j == L.n_dense - 1);
            const float* xin = (j == 0) ? gru_last : ws + L.act[j - 1];
            float* dy = last ? const_cast<float*>(dlogits) : ws + L.dact[j];   // relu=0 on the last layer: not modified
            float* dx = (j == 0) ? ws + L.dgout[L.n_gru - 1] : ws + L.dact[j - 1];
            SED_REQUIRE(g->dense_w[j] && g->dense_b[j], "net_backward: missing gradient buffers of dense layer %d", j);
            SED_TRY(sed_linear_bwd(xin, p->dense_w[j], last ? nullptr : ws + L.act[j], dy, dx, g->dense_w[j],
                                   g->dense_b[j], ws + L.lin_ws, M, L.dK[j], L.dN[j], !last, stream));
        }
        // ── GRU stack ──
        for (int i = L.n_gru - 1; i >= 0; --i) {
            const int H = L.gr[i].H, K = L.gr[i].in;
            const float* xin = (i == 0) ? ws + L.pooled[L.n_conv - 1] : ws + L.gout[i - 1];
            float* dxin = (i == 0) ? ws + L.gradA : ws + L.dgout[i - 1];
            const float* whh[2] = {p->gru_whh[i][0], p->gru_whh[i][1]};
            float* dgi = ws + L.dgi[i];
            float* dgh = ws + L.dgh[i];
            for (int d = 0; d < 2; ++d)
                SED_REQUIRE(g->gru_wih[i][d] && g->gru_whh[i][d] && g->gru_bih[i][d] && g->gru_bhh[i][d],
                            "net_backward: missing gradient buffers of GRU layer %d dir %d", i, d);
            float* dbih[2] = {g->gru_bih[i][0], g->gru_bih[i][1]};
            float* dbhh[2] = {g->gru_bhh[i][0], g->gru_bhh[i][1]};
            SED_TRY(sed_gru_seq_bwd(ws + L.dgout[i], ws + L.saved[i], whh, dgi, dgh, dbih, dbhh, ws + L.gru_bws, B, L.Tp, H, stream));
            // Critical chain: recurrence(i) -> dX(i) -> recurrence(i-1) ...  The weight gradients hang off it: with an auxiliary
            // stream the small ones (dW_hh of every layer, dW_ih of the layers above the first) run there, beside the next
            // data-gradient GEMM / recurrence (which leaves most CUs idle), with their own split-K scratch.
            // (layer 0's dW_hh instead follows its dW_ih on the main stream: there it fills the wait for the top block's
            // BatchNorm backward; beside the MFMA-bound dX GEMM it ran at a tenth of its rate and slowed that one by 18 %)
            const bool on_aux = s_aux && i > 0;
            void* wst = on_aux ? aux_stream : stream;
            float* wws = ws + (on_aux ? L.gemm_ws_aux : L.gemm_ws);
            if (on_aux) {
                (void)hipEventRecord(ev_gru[i], s_main);
                (void)hipStreamWaitEvent(s_aux, ev_gru[i], 0);
            }
            auto dw_hh = [&]() -> int {      // dW_hh = dgh^T h_prev (block-diagonal over the directions)
                for (int d = 0; d < 2; ++d)
                    SED_TRY(sed_gemm_f32_wgrad(dgh + d * 3 * H, 1, 6 * H, ws + L.saved[i] + ((size_t)d * 5 + 4) * H, 10 * H, 1,
                                               g->gru_whh[i][d], H, 3 * H, H, M, wws, wst));
                return 0;
            };
            if (on_aux || !s_aux) SED_TRY(dw_hh());
            const bool fused = p->gru_wih[i][1] == p->gru_wih[i][0] + (size_t)3 * H * K &&
                               g->gru_wih[i][1] == g->gru_wih[i][0] + (size_t)3 * H * K;
            auto dw_ih = [&](void* st, float* gws) -> int {
                if (fused)
                    return sed_gemm_f32_wgrad(dgi, 1, 6 * H, xin, K, 1, g->gru_wih[i][0], K, 6 * H, K, M, gws, st);
                for (int d = 0; d < 2; ++d)
                    SED_TRY(sed_gemm_f32(dgi + d * 3 * H, 1, 6 * H, xin, K, 1, g->gru_wih[i][d], K, nullptr, 0.f, 3 * H, K, M, st));
                return 0;
            };
            if (on_aux) SED_TRY(dw_ih(aux_stream, wws));
            if (i == 0 && s_aux) (void)hipEventRecord(ev_gru[SED_MAX_GRU], s_aux);      // every weight gradient of the aux stream is issued
            // data gradient first: for layer 0 it is the input of the top conv block's BatchNorm backward, which then runs
            // on the auxiliary stream beside the (independent, MFMA-bound) weight-gradient GEMM
            if (fused) {                     // both directions at once: dx = dgi W_ih (K = 6H), dW_ih = dgi^T x (M = 6H)
                SED_TRY(sed_gemm_f32(dgi, 6 * H, 1, p->gru_wih[i][0], K, 1, dxin, K, nullptr, 0.f, M, K, 6 * H, stream));
            } else {
                for (int d = 0; d < 2; ++d)
                    SED_TRY(sed_gemm_f32(dgi + d * 3 * H, 6 * H, 1, p->gru_wih[i][d], K, 1, dxin, K, nullptr, d ? 1.f : 0.f,
                                         M, K, 3 * H, stream));
            }
            if (i == 0 && s_aux) {
                const int top = L.n_conv - 1;
                (void)hipEventRecord(ev_dg[top], s_main);
                (void)hipStreamWaitEvent(s_aux, ev_dg[top], 0);
                SED_TRY(bn_backward(L, c, p, g, x, ws, seed, seed_dev, top, 3, 1.f, aux_stream));
                (void)hipEventRecord(ev_bn[top], s_aux);
            }
            if (!on_aux) SED_TRY(dw_ih(stream, ws + L.gemm_ws));
            if (s_aux && i == 0) SED_TRY(dw_hh());
        }
        // the stage's gradients are complete on the main stream when it returns (the all-reduce of its bucket is issued there)
        if (s_aux) (void)hipStreamWaitEvent(s_main, ev_gru[SED_MAX_GRU], 0);
    }
    // ── conv blocks, last to first ──
    // Critical chain: dgrad(top) -> BN(top-1) -> dgrad(top-1) -> ... -> dgrad(1) -> BN(0).  The weight gradients hang off
    // it, so they are deferred: stage of block l >= 1 = [BN(l) if not done yet] dgrad(l), BN(l-1); the stage of block 1
    // then issues BN(0) on the auxiliary stream and, beside it on the main stream, the weight gradients of ALL blocks
    // >= 1 (MFMA-bound, ~2.4 ms at config 2: the window BN(0) needs at the low occupancy the weight-gradient kernel
    // leaves it).  BN of blocks 1..top-1 runs alone on the main stream (co-running it with a weight gradient stretched
    // it to the whole window and hid nothing); BN(top) was issued by stage 0 beside the GRU weight-gradient GEMM.
    auto bn_passes = [&](int l, void* st) -> int { return bn_backward(L, c, p, g, x, ws, seed, seed_dev, l, 3, 1.f, st); };
    auto wgrad_on = [&](int l, void* st) -> int {
        const ConvL& q = L.cv[l];
        const float* xin = (l == 0) ? x : ws + L.pooled[l - 1];
        float* scratch = ws + ((l == 0 && L.n_conv > 1) ? L.c1_ws : L.wgrad_ws);
        return sed_conv3x3_wgrad_ex(xin, q.nchw, ws + L.dconv[l], g->conv_w[l], scratch, B, q.Cin, q.F, q.T, q.C, (l > 0) ? (c->conv_mode | wg_clean) : 0, st);
    };
    auto wgrad = [&](int l) -> int { return wgrad_on(l, stream); };
    // an unfused first block (Cin > 2): its HBM-bound weight gradient follows its BatchNorm backward on the same stream, so with
    // an auxiliary stream it hides behind the MFMA weight gradients too instead of trailing them on the main stream
    const bool wg0_with_bn = L.n_conv > 1 && !L.cv[0].fused;
    const int top = L.n_conv - 1;
    // the top block's weight gradient runs on the auxiliary stream (decided by the configuration alone: the stages of one
    // backward may arrive in separate calls)
    // Only the direct kernels: the Winograd kernels (one wave per SIMD, 430-460 registers, 130 KB of LDS) share a CU with nothing of
    // their own size and lose more to a co-running vector pass than that pass gains (config 2: the top block's weight gradient
    // beside the BatchNorm apply pass of the block below took 806 us for both, one after the other 418 + 164) — there the top
    // block's weight gradient follows that pass on the main stream.
    const bool top_wgrad_on_aux = s_aux && top > 1 && (c->flags & SED_NET_DIRECT_CONV);
    const bool top_wgrad_in_stage = s_aux && top > 1 && !top_wgrad_on_aux;
    for (int s = (stage_begin > 1 ? stage_begin : 1); s < stage_end; ++s) {
        const int l = L.n_conv - s;
        const ConvL& q = L.cv[l];
        if (l == top) {                              // BN(top): on the auxiliary stream since stage 0, or here
            if (s_aux) (void)hipStreamWaitEvent(s_main, ev_bn[top], 0);
            else SED_TRY(bn_passes(top, stream));
        } else if (l == 0 && s_aux) {
            (void)hipStreamWaitEvent(s_main, ev_bn[0], 0);
        }
        if (l == 0) {
            if (!q.fused && !wg0_with_bn) SED_TRY(wgrad(0));     // fused block 0: its BN pass already produced every gradient
            continue;
        }
        // (the zero rows were cleared on the auxiliary stream at the top of this call: ordered here, far from the launches
        // that read them — a wait right in front of the last weight gradient let the first block's pass on the auxiliary
        // stream reach the CUs first, see below)
        if (s_aux) (void)hipStreamWaitEvent(s_main, ev_gru[SED_MAX_GRU + 3], 0);
        // data gradient = the same convolution with flipped, transposed taps (+ the BatchNorm-backward sums of the block below)
        SED_TRY(dgrad(L, c, p, x, ws, l, stream));
        if (l > 1) {
            if (top_wgrad_on_aux && l == top) {
                // it starts here, beside BN(l-1) and then the next data gradient (config 2: step -65 us; started any
                // earlier, beside dgrad(top), +80 us)
                (void)hipEventRecord(ev_gru[SED_MAX_GRU + 1], s_main);
                (void)hipStreamWaitEvent(s_aux, ev_gru[SED_MAX_GRU + 1], 0);
                SED_TRY(sed_conv3x3_wgrad_ex(ws + L.pooled[l - 1], q.nchw, ws + L.dconv[l], g->conv_w[l], ws + L.wgrad_ws_aux, B, q.Cin, q.F, q.T, q.C, c->conv_mode | wg_clean, aux_stream));
                (void)hipEventRecord(ev_gru[SED_MAX_GRU + 2], s_aux);
            }
            SED_TRY(bn_passes(l - 1, stream));
            if (top_wgrad_in_stage && l == top) SED_TRY(wgrad(l));      // (its gradients are part of this stage's bucket, as on the auxiliary stream)
            continue;
        }
        // the deferred MFMA weight gradients, main stream (the top block's runs on the auxiliary stream since its own stage)
        int first_wg = 0;
        for (int k = top; k >= 1 && !first_wg; --k)
            if (!(k == top && (top_wgrad_on_aux || top_wgrad_in_stage))) first_wg = k;
        unsigned* arrive = nullptr;
        unsigned gate_target = 0;
        const bool co_run = s_aux && (c->flags & SED_NET_DIRECT_CONV);      // see top_wgrad_on_aux: only the direct kernels share their CUs
        if (co_run && first_wg && !(c->flags & SED_NET_NO_GATE)) {
            const ConvL& w = L.cv[first_wg];
            int n = sed_internal_conv3x3_wgrad_workgroups(B, w.Cin, w.F, w.T, w.C, w.nchw, c->conv_mode);
            static thread_local int n_cu[kMaxDev] = {};
            int dev = 0;
            if (n > 0 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDev) {
                if (!n_cu[dev] && hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu[dev] = 0;
                if (n_cu[dev] > 0) {        // the kernel holds one workgroup per CU: at most that many are resident at once
                    gate_target = (unsigned)(n < n_cu[dev] ? n : n_cu[dev]);
                    arrive = (unsigned*)(ws + L.wg_arrive);
                }
            }
        }
        auto main_wgrads = [&]() -> int {
            for (int k = top; k >= 1; --k) {
                if (k == top && (top_wgrad_on_aux || top_wgrad_in_stage)) continue;
                const ConvL& w = L.cv[k];
                SED_TRY(sed_internal_conv3x3_wgrad(ws + L.pooled[k - 1], w.nchw, ws + L.dconv[k], g->conv_w[k], ws + L.wgrad_ws, B, w.Cin, w.F, w.T, w.C,
                                                   c->conv_mode | wg_clean, k == first_wg ? arrive : nullptr, stream));
            }
            return 0;
        };
        if (s_aux && !co_run) {
            // Winograd kernels: the first block's passes run BEFORE the weight gradients on the main stream (config 2: the 30 us
            // assembly kernel, released beside the weight gradient, found no CU for 0.9 ms and then shared the tail with the slab
            // reduction: both took 110 us)
            SED_TRY(bn_passes(0, stream));
            if (wg0_with_bn) SED_TRY(wgrad_on(0, stream));
            SED_TRY(main_wgrads());
            (void)hipEventRecord(ev_bn[0], s_main);
        } else if (s_aux) {
            // block 0's sums came out of the data gradient just issued: finalising them is one tiny launch (16 workgroups, 10 us
            // alone) that took 0.1-0.33 ms when it started the auxiliary chain BESIDE the MFMA weight gradient; on the main
            // stream, before that kernel is issued, it costs its 10 us and the apply pass starts at once
            const bool fin_on_main = L.cv[0].red_rows > 0;
            if (fin_on_main) SED_TRY(bn_backward(L, c, p, g, x, ws, seed, seed_dev, 0, 1, 1.f, stream));
            (void)hipEventRecord(ev_dg[0], s_main);
            // The first block's passes (auxiliary stream) and the weight gradient (main stream) are released by the same event.
            // The persistent MFMA kernel must hold its CUs FIRST — the passes then move in beside it; the other way round its
            // one-per-CU workgroups (132 KB of LDS, 328 registers per lane) find no CU until the passes' grid is exhausted
            // (config 5: 8.6 -> 16.8 ms for that kernel, 62 -> 69 ms per step).  Round 3 ordered the two with a 20 us sleep;
            // now it is a dependency: the kernel's workgroups count themselves in (`arrive`), and a one-wave gate at the head of
            // the auxiliary chain returns when as many as can be resident are (sed_internal_stream_gate; its timeout only
            // guards against a weight gradient that never starts).  The host enqueues the main-stream kernel first, so under
            // load the gate is never enqueued before the kernel it waits for; SED_NET_AUX_FIRST forces the other order (tests).
            auto aux_chain = [&]() -> int {
                (void)hipStreamWaitEvent(s_aux, ev_dg[0], 0);
                SED_TRY(sed_internal_stream_gate(arrive, gate_target, 2000, aux_stream));
                SED_TRY(bn_backward(L, c, p, g, x, ws, seed, seed_dev, 0, fin_on_main ? 2 : 3, 1.f, aux_stream));
                if (wg0_with_bn) SED_TRY(wgrad_on(0, aux_stream));
                (void)hipEventRecord(ev_bn[0], s_aux);
                return 0;
            };
            if (c->flags & SED_NET_AUX_FIRST) {
                SED_TRY(aux_chain());
                SED_TRY(main_wgrads());
            } else {
                SED_TRY(main_wgrads());
                SED_TRY(aux_chain());
            }
        } else {
            SED_TRY(bn_passes(0, stream));
            if (wg0_with_bn) SED_TRY(wgrad_on(0, stream));
            SED_TRY(main_wgrads());
        }
        if (top_wgrad_on_aux) (void)hipStreamWaitEvent(s_main, ev_gru[SED_MAX_GRU + 2], 0);
    }
    return 0;
}

extern "C" int sed_net_backward_ready_stage(const sed_net_cfg* c, int block) {
    if (!c || block < 0 || block >= c->n_conv) { sed_set_error("net_backward_ready_stage: bad arguments"); return SED_EINVAL; }
    return block == 0 ? c->n_conv : c->n_conv - 1;
}

extern "C" int sed_net_backward_phases(const sed_net_cfg* c, const sed_net_params* p, const sed_net_params* g,
                                       const float* x, const float* dlogits, void* workspace, uint64_t seed,
                                       int phase_begin, int phase_end, float count_scale, void* stream) {
    SED_REQUIRE(p && g && x && dlogits && workspace, "net_backward_phases: null pointer");
    SED_REQUIRE(count_scale >= 1.f, "net_backward_phases: count_scale must be >= 1");
    Layout L;
    SED_TRY(build_layout(c, 1, &L));
    SED_REQUIRE(phase_begin >= 0 && phase_end <= 2 * L.n_conv + 1 && phase_begin < phase_end,
                "net_backward_phases: bad phase range [%d,%d)", phase_begin, phase_end);
    float* ws = (float*)workspace;
    const int B = c->B;
    if (phase_begin == 0) SED_TRY(sed_net_backward(c, p, g, x, dlogits, workspace, seed, nullptr, 0, 1, stream, nullptr));
    for (int ph = (phase_begin > 1 ? phase_begin : 1); ph < phase_end; ++ph) {
        const int k = (ph - 1) >> 1, l = L.n_conv - 1 - k;
        const ConvL& q = L.cv[l];
        if (ph & 1) {                                            // 2k+1: reduction pass
            SED_TRY(bn_backward(L, c, p, g, x, ws, seed, nullptr, l, 1, 1.f, stream));
            continue;
        }
        SED_TRY(bn_backward(L, c, p, g, x, ws, seed, nullptr, l, 2, count_scale, stream));
        if (q.fused) continue;
        const float* xin = (l == 0) ? x : ws + L.pooled[l - 1];
        SED_TRY(sed_conv3x3_wgrad_ex(xin, q.nchw, ws + L.dconv[l], g->conv_w[l], ws + ((l == 0 && L.n_conv > 1) ? L.c1_ws : L.wgrad_ws),
                                     B, q.Cin, q.F, q.T, q.C, (l > 0) ? c->conv_mode : 0, stream));
        if (l > 0) SED_TRY(dgrad(L, c, p, x, ws, l, stream));
    }
    return 0;
}
