/*
 * sedcrnn.h — flat C ABI of libsedcrnn.so (hand-written HIP for gfx950 / MI355X).
 *
 * This is the drop-in boundary of the SEDnet hot path.  The
 * reference (noamzilo/sed-crnn) has no FFI of its own for this path: it reaches
 * native code only through torch.nn modules, so each entry point below cites the
 * reference call site whose ATen kernel it replaces (paths are relative to the
 * reference checkout).  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; all tensors are fp32 device memory unless noted;
 *   - the CALLER owns every buffer (inputs, outputs, workspace); the library never
 *     allocates or frees device memory and keeps no pointer past return;
 *   - every launch is asynchronous on the hipStream_t passed as `stream`
 *     (void*; NULL = the default stream); no entry synchronises the device and the entries are
 *     re-entrant (no mutable process state) — with TWO exceptions: the measurement-only
 *     sed_prof_* group at the end of this header, which is off by default, and a per-thread,
 *     per-device cache of hipEvent_t host objects (at most 15 per calling thread and device,
 *     timing disabled) that sed_net_backward creates on its first call WITH an aux_stream to order
 *     the two streams; they hold no device memory, are never shared between threads, and live
 *     until the process exits (a caller that passes aux_stream = NULL never creates them);
 *   - return value: 0 = ok, <0 = argument / shape error, >0 = hipError_t.  The message
 *     of the last failure on the calling thread is sed_last_error_string().
 *
 * Activation layout inside the conv stack is channels-last: act[b][t][f][c]
 * ("(seq, mel, chan)"), c contiguous.  The network input keeps the reference layout
 * x[b][cin][f][t] (sed.py:105) and the GRU input keeps the reference feature order
 * feat = c*F + f (sed.py:108-110), so reference state_dicts load unchanged.
 */
#ifndef SEDCRNN_H
#define SEDCRNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SED_MAX_CONV 4
#define SED_MAX_GRU 4
#define SED_MAX_DENSE 4

/* ───────────── library ───────────── */
int sed_version(void);                       /* 10000*major + 100*minor + patch */
const char* sed_last_error_string(void);     /* thread-local, never NULL */

/* ───────────── conv 3x3, stride 1, zero pad 1 (sed.py:88,107; crnn_lightning.py:47) ─────────────
 * w_oihw is the reference/PyTorch weight [Cout][Cin][3][3] (kh over mel, kw over time).
 * sed_conv3x3_pack_weights lays it out for the kernels:
 *   wp_fwd  [9][Cout][Cin]  tap-major, ci contiguous           (forward)
 *   wp_dgrad[9][Cin][Cout]  taps flipped, ci/co swapped        (data gradient = conv of dy)
 * either output pointer may be NULL. */
int sed_conv3x3_pack_weights(const float* w_oihw, float* wp_fwd, float* wp_dgrad,
                             int Cout, int Cin, void* stream);

/* Number of per-block partial rows sed_conv3x3_fwd writes into `stat_partials`
 * ([rows][2][Cout]: sum and sum of squares of the outputs, pre-BN, bias included). */
int sed_conv3x3_stat_rows(int B, int Cin, int F, int T, int Cout, int x_is_nchw);

/* y[b][t][f][co] = bias[co] + sum_{kh,kw,ci} w[co][ci][kh][kw] x[.. f+kh-1, t+kw-1 ..]
 * x_is_nchw=1: x is the network input [B][Cin][F][T]; 0: x is channels-last [B][T][F][Cin].
 * wp = wp_fwd from sed_conv3x3_pack_weights ([9][Cout][Cin]).
 * bias may be NULL; stat_partials may be NULL (no statistics). */
int sed_conv3x3_fwd(const float* x, int x_is_nchw, const float* wp, const float* bias,
                    float* y, float* stat_partials,
                    int B, int Cin, int F, int T, int Cout, void* stream);

/* EXPERIMENT, explicit opt-in (never the default, never what bench.py reports as `value`): `mode` 1 runs the MFMA path
 * (Cin%32 == 0, Cout%32 == 0) on a 3-term bf16 split — x*w ~ x_hi*w_hi + x_hi*w_lo + x_lo*w_hi with fp32 accumulation on
 * v_mfma_f32_32x32x16_bf16: 5.3x the matrix rate of the exact-fp32 MFMA at a relative error of ~4e-6 on a K = 1152 sum
 * (exact fp32: 3e-7).  mode 0 = the exact fp32 kernels (identical to the entries above).  Weights for mode m must be
 * packed with sed_conv3x3_pack_weights_ex(..., m); shapes outside the MFMA path fall back to mode 0 in both entries. */
int sed_conv3x3_pack_weights_ex(const float* w_oihw, float* wp_fwd, float* wp_dgrad, int Cout, int Cin, int mode, void* stream);
int sed_conv3x3_fwd_ex(const float* x, int x_is_nchw, const float* wp, const float* bias, float* y, float* stat_partials,
                       int B, int Cin, int F, int T, int Cout, int mode, void* stream);

/* Weight gradient (aten::convolution_backward, reached from loss.backward() sed.py:137).
 * dw_oihw[co][ci][kh][kw] = sum_pos x[pos+tap][ci] * dy[pos][co].  dy is channels-last.
 * workspace: sed_conv3x3_wgrad_workspace_bytes(). Deterministic (fixed-order slab reduce). */
size_t sed_conv3x3_wgrad_workspace_bytes(int B, int Cin, int F, int T, int Cout);
int sed_conv3x3_wgrad(const float* x, int x_is_nchw, const float* dy, float* dw_oihw,
                      void* workspace, int B, int Cin, int F, int T, int Cout, void* stream);

/* mode 1 (EXPERIMENT, as sed_conv3x3_fwd_ex): the MFMA path on the 3-term bf16 split; other shapes run mode 0.
 * mode | SED_WGRAD_ZERO_ROW_CLEAN: the exact-fp32 MFMA kernel reads out-of-image rows from a zero-filled row at the START of
 * the workspace (sed_conv3x3_wgrad_zero_row_bytes(), 0 for shapes that have none), which the entry clears with a memset on
 * `stream` in every call — unless this flag says the caller keeps those bytes zero (nothing ever writes them): a plan that
 * calls the entry every step clears them once, off its critical chain, and saves the memset node in front of each launch.
 * mode | SED_WGRAD_DIRECT: mel extents that cut into 40- or 32-column tiles run the position-contiguous kernel in the Winograd
 * domain by default (F(2x2,3x3): 16 products per 2x2 tile and channel pair instead of 36, dW = G^T dU G applied by the slab
 * reduction; exact-fp32 MFMA arithmetic, see wino.hip); this flag keeps the direct 36-product form (A/B measurements, tests). */
#define SED_WGRAD_ZERO_ROW_CLEAN 0x100
#define SED_WGRAD_DIRECT 0x200
size_t sed_conv3x3_wgrad_zero_row_bytes(int B, int Cin, int F, int T, int Cout);
int sed_conv3x3_wgrad_ex(const float* x, int x_is_nchw, const float* dy, float* dw_oihw,
                         void* workspace, int B, int Cin, int F, int T, int Cout, int mode, void* stream);

/* Inference (run_epoch with optim=None, sed.py:128-141; `model.eval()`): conv3x3 + BatchNorm2d on RUNNING statistics + ReLU +
 * MaxPool2d((1,2)) (sed.py:107) in ONE launch.  BatchNorm is a per-channel affine map there, so it is folded on the way in:
 * sed_conv3x3_pack_weights_bn_folded writes the MFMA fragments of w * scale[co] and bias_folded = bias * scale + shift with
 * scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale; the kernel's epilogue applies ReLU and the time
 * pool to its accumulators and writes only the pooled tensor [B][T/2][F][Cout] (channels-last; a ragged last frame is dropped
 * like nn.MaxPool2d does): in + 0.5 out bytes instead of in + 2.5 out.  x [B][T][F][Cin] channels-last.
 * _supported: the 128-wide exact-fp32 MFMA tile with an even number of time rows (Cin % 32 == 0, Cout % 128 == 0, mel widths
 * whose tile is <= 32 columns: 40 -> 20, 64 -> 32, ...); otherwise use sed_conv3x3_fwd_ex + sed_bn_relu_pool_drop_fwd. */
int sed_conv3x3_bn_relu_pool_eval_supported(int B, int Cin, int F, int T, int Cout);
int sed_conv3x3_pack_weights_bn_folded(const float* w, const float* bias, const float* gamma, const float* beta,
                                       const float* running_mean, const float* running_var, float eps,
                                       float* wf, float* bias_folded, int Cout, int Cin, void* stream);
int sed_conv3x3_bn_relu_pool_eval(const float* x, const float* wp_folded, const float* bias_folded, float* pooled,
                                  int B, int Cin, int F, int T, int Cout, void* stream);
/* Data gradient of conv block l (the convolution of dy with wp_dgrad) FUSED with the reduction pass of the BatchNorm / ReLU /
 * max-pool / dropout backward of block l-1 (sed_bn_relu_pool_drop_bwd_reduce below), for the exact-fp32 MFMA shapes:
 * dy [B][T][F][C] -> dx [B][T][F][Cin] = gradient of block l-1's pooled output, and partials [rows][2][Cin] = per-workgroup
 * (sum g, sum g*xhat) of block l-1 — the input of sed_bn_bwd_finalize — formed in the epilogue from block l-1's own forward
 * OUTPUT `pooled` [B][T][F][Cin] (channels-last): it is > 0 exactly where the gradient passes (element kept by the dropout AND
 * ReLU gate open), g = dx / (1-p) there, and the normalised activation at the arg-max is (pooled (1-p) - beta) / gamma.
 * That recovery divides a rounding error of eps |z| by gamma, so with a stored conv output below (conv_out_below != NULL) the
 * channels with |gamma| < |beta| / 64 (and gamma == 0) contribute 0 to sum g*xhat here and MUST be finalised with
 * sed_bn_bwd_finalize_small_gamma, which recomputes exactly those channels from the conv output; sum g is exact for every
 * channel.  conv_out_below [B][Ty][Fy][Cin] is only tested for NULL here, mean / rstd are unused (kept for the signature) (pool_f, pool_t: block l-1's
 * pool, Ty / pool_t == T, Fy / pool_f == F); conv_out_below may be NULL for a block that stores none (the recomputed first
 * block: sed_conv1_bwd_apply_wgrad then supplies dgamma of those channels).  rows = sed_conv3x3_dgrad_bnred_rows(); 0 = shape not supported (use
 * sed_conv3x3_fwd_ex + sed_bn_relu_pool_drop_bwd_reduce).  Replaces a 1.5x re-read of block l-1's conv output. */
int sed_conv3x3_dgrad_bnred_rows(int B, int C, int F, int T, int Cin);
int sed_conv3x3_dgrad_bnred(const float* dy, const float* wp_dgrad, float* dx, float* partials, const float* pooled,
                            const float* gamma, const float* beta, const float* conv_out_below, const float* mean,
                            const float* rstd, float drop_p, int pool_f, int pool_t, int Fy, int Ty,
                            int B, int C, int F, int T, int Cin, void* stream);
/* ---- Winograd F(2x2, 3x3) form of the 128-input-channel convolutions (wino.hip; reference sed.py:88,107, the same nn.Conv2d) ----
 * Exact-fp32 MFMA arithmetic on 16 transformed products per 2x2 output tile instead of 36 (2.25x fewer MFMAs); the transforms'
 * coefficients are 0, +-1 and 1/2, the result differs from the direct kernels' by a few ulp of the accumulated magnitude.
 * Shapes: Cin == 128, Cout % 64 == 0, F and T even, a tile-row patch that fits the LDS; sed_conv3x3_wino_rows() = the number of
 * statistic / partial rows ([rows][2][Cout], as sed_conv3x3_fwd) or 0 when the shape is not taken.
 * pack_weights: w [Cout][Cin][3][3] -> uf (forward) and ud (data gradient: flipped taps, channel roles swapped), each
 * sed_conv3x3_wino_packed_floats() floats (16 Cin Cout transformed weights in MFMA fragment order + a zero tail the kernel reads
 * its zero padding from); either may be NULL.
 * wino_fwd = sed_conv3x3_fwd on channels-last x; wino_dgrad_bnred = sed_conv3x3_dgrad_bnred (same arguments and outputs). */
int sed_conv3x3_wino_rows(int B, int Cin, int F, int T, int Cout);
/* measurement only (tools/kprobe.py): buf = 4 device uint64 — prologue / main loop / epilogue ticks of the 100 MHz clock summed
 * over the workgroups of every following Winograd forward / data-gradient launch, and the workgroup count; NULL switches it off. */
int sed_conv3x3_wino_phase_ticks(unsigned long long* buf);
size_t sed_conv3x3_wino_packed_floats(int Cout, int Cin);
int sed_conv3x3_wino_pack_weights(const float* w_oihw, float* uf, float* ud, int Cout, int Cin, void* stream);
int sed_conv3x3_wino_fwd(const float* x, const float* uf, const float* bias, float* y, float* stat_partials,
                         int B, int Cin, int F, int T, int Cout, void* stream);
int sed_conv3x3_wino_dgrad_bnred(const float* dy, const float* ud, float* dx, float* partials, const float* pooled,
                                 const float* gamma, const float* beta, const float* conv_out_below, const float* mean,
                                 const float* rstd, float drop_p, int pool_f, int pool_t, int Fy, int Ty,
                                 int B, int C, int F, int T, int Cin, void* stream);
/* inference (= sed_conv3x3_pack_weights_bn_folded + sed_conv3x3_bn_relu_pool_eval): BatchNorm on running statistics folded into the
 * transformed weights and the bias, ReLU + (1,2) time pool in the epilogue (the two time rows of a 2x2 tile are a pooling pair):
 * pooled [B][T/2][F][Cout]; shapes as sed_conv3x3_wino_rows() > 0. */
int sed_conv3x3_wino_pack_weights_bn_folded(const float* w_oihw, const float* bias, const float* gamma, const float* beta,
                                            const float* running_mean, const float* running_var, float eps,
                                            float* uf, float* bias_folded, int Cout, int Cin, void* stream);
int sed_conv3x3_wino_bn_relu_pool_eval(const float* x, const float* uf_folded, const float* bias_folded, float* pooled,
                                       int B, int Cin, int F, int T, int Cout, void* stream);
/* = sed_conv3x3_dgrad_bnred_rg (the first block's weight-gradient sums from the same epilogue); rows = sed_conv3x3_wino_rg_rows() */
int sed_conv3x3_wino_rg_rows(int B, int C, int F, int T, int Cin, int Cin1);
int sed_conv3x3_wino_dgrad_bnred_rg(const float* dy, const float* ud, float* dx, float* partials, const float* pooled,
                                    const float* gamma, const float* beta, const float* mean, const float* rstd, float drop_p,
                                    const float* x1, int Cin1, const unsigned char* argmax_bits, float* rg_partials,
                                    int B, int C, int F, int T, int Cin, void* stream);
/* The same launch when the block below is the recomputed first block with Cin1 = 1 or 2 input channels and pool (1,2)
 * (sed.py:86-92, ch = 1 or 2): the epilogue also forms that block's weight-gradient sums — rg_partials [rows][Cin][1 + 9 Cin1]
 * = (sum g, R_k) per channel and workgroup, R_{(3kh+kw) Cin1 + ci} = sum g~ x[ci][f+kh-1][2t'+sel+kw-1] with sel the arg-max
 * bit — from the network input x1 [B][Cin1][F][2T] and
 * the arg-max bits of sed_conv1_bn_relu_pool_drop_fwd, for sed_conv1_bwd_wgrad_assemble.  Replaces the pass of
 * sed_conv1_bwd_wgrad over dx, the pooled tensor and the bits.  rows = sed_conv3x3_dgrad_bnred_rg_rows(); 0 = not supported. */
int sed_conv3x3_dgrad_bnred_rg_rows(int B, int C, int F, int T, int Cin, int Cin1);
int sed_conv3x3_dgrad_bnred_rg(const float* dy, const float* wp_dgrad, float* dx, float* partials, const float* pooled,
                               const float* gamma, const float* beta, const float* mean, const float* rstd, float drop_p,
                               const float* x1, int Cin1, const unsigned char* argmax_bits, float* rg_partials,
                               int B, int C, int F, int T, int Cin, void* stream);

/* ───────────── BatchNorm2d + ReLU + MaxPool2d + Dropout (sed.py:89-92,107; crnn_lightning.py:48-52) ─────────────
 * Training statistics: reduce the conv partials in a fixed order (double accumulation),
 * write mean/rstd and the fused scale/shift (scale = gamma*rstd, shift = beta - mean*scale),
 * update running_mean / running_var (momentum, unbiased var) like nn.BatchNorm2d.
 * count = B*T*F elements per channel. */
int sed_bn_finalize_train(const float* stat_partials, int rows, int C, double count,
                          const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps,
                          float* mean, float* rstd, float* scale, float* shift, void* stream);
/* The same in two steps, for synchronised BatchNorm across ranks: sums[2C] = (sum x, sum x^2) of this rank's partials;
 * the host all-reduces (SUM) the 2C floats; finalize_from_sums then uses the GLOBAL element count. */
int sed_bn_stat_sums(const float* stat_partials, int rows, int C, float* sums, void* stream);
int sed_bn_finalize_from_sums(const float* sums, int C, double count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps,
                              float* mean, float* rstd, float* scale, float* shift, void* stream);
/* Eval: scale/shift from the running statistics. */
int sed_bn_finalize_eval(const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, int C,
                         float* scale, float* shift, void* stream);

/* out = dropout(maxpool_{pool_f x pool_t}(relu(scale*y + shift))).
 * y [B][T][F][C] channels-last.  out_tcf=0: out [B][T/pt][F/pf][C];
 * out_tcf=1: out [B][T/pt][C][F/pf]  (the GRU feature order c*F'+f, sed.py:108-110).
 * T/pt and F/pf are floor divisions like nn.MaxPool2d's (sed.py:90): a ragged tail is dropped by the pool; it still
 * counts in the batch statistics and, in the backward, receives the statistics terms of the gradient.
 * drop_p=0 or training=0 disables dropout; the keep mask is a counter-based hash of
 * (seed, logical element index) so the backward pass regenerates it. */
int sed_bn_relu_pool_drop_fwd(const float* y, const float* scale, const float* shift, float* out,
                              int B, int T, int F, int C, int pool_f, int pool_t, int out_tcf,
                              float drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream);
/* seed_dev (here and below; may be NULL): device pointer to a per-step salt added to `seed`, so that a captured
 * hipGraph of the step draws fresh masks on every replay (see sed_step_advance). */

/* Backward of the block above (training statistics).  Two passes:
 *  reduce: partials [rows][2][C] of  sum g  and  sum g*xhat  where g is dout routed through
 *          dropout, max-pool arg-max (first maximum wins) and ReLU;
 *  apply : dy = scale*(g - sum_g/N - xhat*sum_gx/N); also dgamma = sum_gx, dbeta = sum_g and
 *          the conv-bias gradient partials sum dy.  Fixed-order reductions. */
/* The routing DECISIONS of the block, as the forward and backward kernels make them (z = y*scale + shift, first maximum of a
 * window wins like max_pool2d, ReLU gate = max > 0; sed.py:107 `pool(torch.relu(bn(conv(x))))`): route [B][T/pt][F/pf][C], one
 * byte per pooled element, 0 = gate closed, 1 + w = window element w = df*pool_t + dt.  Inspection / parity tests: an oracle that
 * routes its gradient with these decisions differs from the kernels by rounding only.  pool_f*pool_t <= 254. */
int sed_bn_relu_pool_route(const float* y, const float* scale, const float* shift, unsigned char* route,
                           int B, int T, int F, int C, int pool_f, int pool_t, void* stream);
int sed_bn_bwd_rows(int B, int T, int pool_t);   /* partial rows written by the two passes below */
int sed_bn_relu_pool_drop_bwd_reduce(const float* y, const float* dout, const float* scale,
                                     const float* shift, const float* mean, const float* rstd,
                                     float* partials, int B, int T, int F, int C, int pool_f,
                                     int pool_t, int out_tcf, float drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream);
int sed_bn_bwd_finalize(const float* partials, int rows, int C, float* sum_g, float* sum_gx,
                        float* dgamma, float* dbeta, void* stream);
/* sed_bn_bwd_finalize for partial rows that come out of sed_conv3x3_dgrad_bnred with a stored conv output below: channels
 * with |gamma| < |beta| / 64 (incl. gamma == 0 with beta > 0) were left out of sum g*xhat there (xhat = (z - beta)/gamma loses
 * eps |beta/gamma|) and are recomputed here from the block's conv output y [B][T][F][C]: g = dpooled/(1-p) where pooled > 0
 * (dpooled, pooled [B][T/pool_t][F/pool_f][C] channels-last), the window searched as the forward searches it (z = fma(y, scale,
 * shift), first maximum; sed.py:107 `pool(relu(bn(.)))`), xhat = (y_max - mean) rstd.  Same launch shape as
 * sed_bn_bwd_finalize; the recomputation is a cold path of the workgroups that own such a channel. */
int sed_bn_bwd_finalize_small_gamma(const float* partials, int rows, int C, float* sum_g, float* sum_gx,
                                    float* dgamma, float* dbeta, const float* dpooled, const float* pooled,
                                    const float* y, const float* gamma, const float* beta, const float* mean,
                                    const float* rstd, const float* scale, const float* shift,
                                    int B, int T, int F, int pool_f, int pool_t, float drop_p, void* stream);
/* The reduce pass of the block that feeds the GRU (out_tcf layout [B][T/pt][C][F/pf]) formed from that block's own pooled
 * OUTPUT and its gradient instead of the conv output: the same partial rows (sed_bn_bwd_rows) for sed_bn_bwd_finalize.
 * pooled > 0 exactly where the gradient passes, and the BatchNorm output at the arg-max is pooled*(1-p), so
 * xhat = (pooled*(1-p) - beta)/gamma; `y` (the conv output), scale and shift are read only for channels with
 * |gamma| < |beta| / 64 (incl. gamma == 0 with beta > 0), whose xhat is taken at the window's arg-max of the conv output.
 * _supported: out_tcf, F/pool_f a multiple of 4 and C*F/pool_f <= 16384; otherwise use sed_bn_relu_pool_drop_bwd_reduce. */
int sed_bn_bwd_reduce_pooled_supported(int F, int C, int pool_f, int pool_t, int out_tcf);
int sed_bn_bwd_reduce_pooled(const float* pooled, const float* dout, const float* gamma, const float* beta,
                             const float* y, const float* mean, const float* rstd, const float* scale, const float* shift,
                             float* partials, int B, int T, int F, int C, int pool_f, int pool_t, int out_tcf, float drop_p, void* stream);
int sed_bn_relu_pool_drop_bwd_apply(const float* y, const float* dout, const float* scale,
                                    const float* shift, const float* mean, const float* rstd,
                                    const float* sum_g, const float* sum_gx, float* dy,
                                    float* dbias_partials, int B, int T, int F, int C, int pool_f,
                                    int pool_t, int out_tcf, float drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream);
/* out[c] = sum_r partials[r][c] in row order (double accumulation). */
int sed_reduce_rows(const float* partials, int rows, int C, int row_stride, float* out, void* stream);

/* ───────────── fused first conv block (conv recomputed, never stored; Cin <= 2) ─────────────
 * The first block of the stack (sed.py:86-92,106-107, ch = 1) expands a few MB of input into the largest tensor of
 * the step.  These four entries replace sed_conv3x3_fwd + the bn_* entries + sed_conv3x3_wgrad for block 1 and
 * recompute conv(x) on the fly, so that tensor never exists in HBM.  x [B][Cin][F][T] (network input layout),
 * wp = [9][C][Cin] from sed_conv3x3_pack_weights, pooled output / dout channels-last [B][T/pt][F/pf][C].
 * rows = sed_conv1_fused_rows(B,T) partial rows for bwd_reduce ([rows][2][C]); the statistics are one row. */
int sed_conv1_fused_supported(int Cin, int F, int T, int C, int pool_f, int pool_t);   /* 1 = shapes accepted */
int sed_conv1_fused_rows(int B, int T);
/* Statistics WITHOUT recomputing the convolution: sum y and sum y^2 of every output channel follow from the first and
 * second moments of the 9*Cin shifted inputs (one pass over x, fp64 quadratic form per channel).  Writes ONE partial
 * row [1][2][C]; workspace >= sed_conv1_stats_workspace_bytes(). */
size_t sed_conv1_stats_workspace_bytes(int B, int Cin, int T);
/* moments (may be NULL): device array of sed_conv1_moments_doubles(Cin) doubles that receives the input moments themselves
 * ([9*Cin] first moments, then the upper triangle of the second moments row by row) for sed_conv1_bwd_wgrad. */
size_t sed_conv1_moments_doubles(int Cin);
int sed_conv1_stats(const float* x, const float* wp, const float* bias, float* stat_partials, void* workspace,
                    int B, int Cin, int F, int T, int C, double* moments, void* stream);
/* argmax_bits (may be NULL; (1,2) pool only): one byte per output channel quad, [B][T/2][F][C/4], bit k = the second time row of
 * the window holds the maximum of channel 4q+k (the first maximum wins a tie): what sed_conv1_bwd_wgrad routes the gradient by. */
int sed_conv1_bn_relu_pool_drop_fwd(const float* x, const float* wp, const float* bias, const float* scale,
                                    const float* shift, float* out, int B, int Cin, int F, int T, int C,
                                    int pool_f, int pool_t, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                                    unsigned char* argmax_bits, void* stream);
/* sed_bn_relu_pool_route for the recomputed first block (the decisions of its passes; same shapes as the forward entry). */
int sed_conv1_route(const float* x, const float* wp, const float* bias, const float* scale, const float* shift,
                    unsigned char* route, int B, int Cin, int F, int T, int C, int pool_f, int pool_t, void* stream);
int sed_conv1_bwd_reduce(const float* x, const float* wp, const float* bias, const float* dout,
                         const float* scale, const float* shift, const float* mean, const float* rstd,
                         float* partials, int B, int Cin, int F, int T, int C, int pool_f, int pool_t,
                         float drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream);
size_t sed_conv1_bwd_apply_workspace_bytes(int B, int Cin, int T, int C);
/* dy is formed on the fly: writes dw_oihw [C][Cin][3][3] and the conv-bias gradient only.
 * gamma / beta / dgamma (all three or none; may be NULL): when given, dgamma[c] is overwritten by this pass's own sum g*xhat
 * for every channel with gamma[c] == 0 and beta[c] > 0 — the one case in which the reduction fused into the data gradient
 * above (sed_conv3x3_dgrad_bnred) cannot form it, because this block stores no conv output. */
int sed_conv1_bwd_apply_wgrad(const float* x, const float* wp, const float* bias, const float* dout,
                              const float* scale, const float* shift, const float* mean, const float* rstd,
                              const float* sum_g, const float* sum_gx, float* dw_oihw, float* dbias,
                              void* workspace, int B, int Cin, int F, int T, int C, int pool_f, int pool_t,
                              float drop_p, uint64_t seed, const uint64_t* seed_dev,
                              const float* gamma, const float* beta, float* dgamma, void* stream);

/* Backward of the recomputed first block WITHOUT recomputing it ((1,2) pool; replaces sed_conv1_bwd_apply_wgrad where
 * sed_conv1_rgrad_supported() and the BatchNorm-backward sums are already known, e.g. from sed_conv3x3_dgrad_bnred):
 *   dW_k = scale [ R_k - (sum_g/N) S1_k - (sum_gx/N) rstd ( b S1_k + sum_k' w_k' G_kk' - mean S1_k ) ],   N = B*T*F,
 * with S1 / G the input moments of sed_conv1_stats and R_k = sum over pooled elements of g * x[arg-max position + tap k], the
 * only sum this pass forms: g = dout / (1-p) where `pooled` (the block's forward output) is > 0, the arg-max row from
 * `argmax_bits`.  No convolution, dropout hash or BatchNorm arithmetic is redone.  Also writes the conv-bias gradient and, given
 * gamma / beta / dgamma, dgamma of channels with gamma == 0 and beta > 0 (see sed_conv1_bwd_apply_wgrad).
 * workspace >= sed_conv1_bwd_wgrad_workspace_bytes(). */
int sed_conv1_rgrad_supported(int Cin, int F, int T, int C, int pool_f, int pool_t);
size_t sed_conv1_bwd_wgrad_workspace_bytes(int B, int Cin, int T, int C);
int sed_conv1_bwd_wgrad(const float* x, const float* dout, const float* pooled, const unsigned char* argmax_bits,
                        const double* moments, const float* wp, const float* bias, const float* mean, const float* rstd,
                        const float* scale, const float* sum_g, const float* sum_gx, float* dw_oihw, float* dbias,
                        void* workspace, int B, int Cin, int F, int T, int C, float drop_p,
                        const float* gamma, const float* beta, float* dgamma, void* stream);
/* its assembling half alone, from partials [rows][C][1 + 9 Cin] formed elsewhere (sed_conv3x3_dgrad_bnred_rg).
 * sum_gx may be NULL (both entries): sum g*xhat is then formed from the block's own sums — rstd (b sum g + sum_k w_k R_k - mean
 * sum g), exact for every gamma, where the value recovered from the pooled output, (z - beta)/gamma, loses eps |beta/gamma| — and
 * also written to dgamma; with sum_gx given (a recomputing reduce pass, or sums all-reduced for synchronised BatchNorm) it is
 * used as is and only gamma == 0 channels take the own value for dgamma. */
int sed_conv1_bwd_wgrad_assemble(const float* partials, int rows, const double* moments, const float* wp, const float* bias,
                                 const float* mean, const float* rstd, const float* scale, const float* sum_g,
                                 const float* sum_gx, float* dw_oihw, float* dbias, int B, int Cin, int F, int T, int C,
                                 const float* gamma, const float* beta, float* dgamma, void* stream);

/* ───────────── dense GEMM on fp32 MFMA (aten::mm/addmm under nn.GRU / nn.Linear, sed.py:101-103) ─────────────
 * C[i][j] = sum_k A(i,k) * B(k,j) (+ bias[j]) (+ beta*C[i][j]), C row-major with leading dim ldc.
 * A(i,k) = A[i*a_si + k*a_sk], B(k,j) = B[k*b_sk + j*b_sj]; for each operand one of the
 * two strides must be 1.  Exact fp32 (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain). */
int sed_gemm_f32(const float* A, long a_si, long a_sk, const float* B, long b_sk, long b_sj,
                 float* C, long ldc, const float* bias, float beta, int M, int N, int K, void* stream);
/* Same product (optional bias, beta = 0) with a scratch buffer: small-output / long-K shapes (the GRU weight
 * gradients, M*N small, K = B*T'; the input projection of a small batch, M = B*T' small, K = C*F') would leave most
 * CUs without a tile, so they are split along K into fixed slices that are summed in slice order (deterministic; the
 * bias is added by the summing pass).  workspace >= sed_gemm_f32_workspace_bytes(M,N,K) (0 = the shape is not split). */
size_t sed_gemm_f32_workspace_bytes(int M, int N, int K);   /* covers sed_gemm_f32_ws and sed_gemm_f32_wgrad */
int sed_gemm_f32_ws(const float* A, long a_si, long a_sk, const float* B, long b_sk, long b_sj,
                    float* C, long ldc, const float* bias, int M, int N, int K, void* workspace, void* stream);

/* Weight-gradient form (dW = dY^T X under loss.backward(), sed.py:137): the same product, beta = 0, no bias, whose K axis is
 * the batch x time axis.  Besides the small-output split above it may be cut into two K-slices when its best tiling
 * gives at most one tile per CU (two co-resident workgroups per CU instead of one).  Deterministic for given (M,N,K). */
int sed_gemm_f32_wgrad(const float* A, long a_si, long a_sk, const float* B, long b_sk, long b_sj,
                       float* C, long ldc, int M, int N, int K, void* workspace, void* stream);

/* Small dense layer y = act(x W^T + b) for the time-distributed head (sed.py:103,112;
 * crnn_lightning.py:63-64,72-73). x [M][K], W [N][K], y [M][N]; relu=1 applies ReLU. */
int sed_linear_fwd(const float* x, const float* W, const float* b, float* y,
                   int M, int K, int N, int relu, void* stream);
/* dy is modified in place when relu=1 (masked by y>0).  dx may be NULL.
 * workspace: sed_linear_bwd_workspace_bytes(). */
size_t sed_linear_bwd_workspace_bytes(int M, int K, int N);
int sed_linear_bwd(const float* x, const float* W, const float* y, float* dy, float* dx,
                   float* dW, float* db, void* workspace, int M, int K, int N, int relu, void* stream);

/* ───────────── bidirectional GRU layer recurrence (nn.GRU, sed.py:101-102,111) ─────────────
 * PyTorch gate convention (r,z,n): n = tanh(gi_n + r*(W_hn h + b_hn)), h' = (1-z)*n + z*h, h0 = 0.
 * gi   [B][T][2][3H]  input projections x W_ih^T + b_ih of both directions (dir 1 = reverse);
 * whh  2 pointers to weight_hh [3H][H]; bhh 2 pointers to bias_hh [3H];
 * out  [B][T][2H]  (forward | reverse), exactly nn.GRU's batch_first output;
 * saved[B][T][2][5][H] = r, z, n, (W_hn h + b_hn), h_prev  (training only, may be NULL). */
size_t sed_gru_seq_workspace_bytes(int H);          /* scratch for the transposed W_hh */
int sed_gru_seq_fwd(const float* gi, const float* const* whh, const float* const* bhh,
                    float* out, float* saved, void* workspace, int B, int T, int H, void* stream);
/* dout [B][T][2H] -> dgi [B][T][2][3H] (grad of gi) and dgh [B][T][2][3H] (grad of W_hh h + b_hh).
 * dbih / dbhh (2 pointers each, or both NULL): bias gradients summed in-kernel (no re-read of dgi/dgh);
 * workspace >= sed_gru_seq_bwd_workspace_bytes(B, H) when they are requested. */
size_t sed_gru_seq_bwd_workspace_bytes(int B, int H);
int sed_gru_seq_bwd(const float* dout, const float* saved, const float* const* whh,
                    float* dgi, float* dgh, float* const* dbih, float* const* dbhh, void* workspace,
                    int B, int T, int H, void* stream);

/* ───────────── loss heads (sed.py:136,160; crnn_lightning.py:27-35) ─────────────
 * kind 0: BCEWithLogits mean; kind 1: focal (alpha, gamma, log(pt+1e-12)); reduction_mean=0 -> sum.
 * loss (1 float), dlogits (may be NULL) = d loss / d logits, probs (may be NULL) = sigmoid(logits). */
int sed_loss_fwd_bwd(const float* logits, const float* targets, int n, int kind, float alpha,
                     float gamma, int reduction_mean, float* loss, float* dlogits, float* probs,
                     void* stream);
int sed_sigmoid(const float* x, float* y, int n, void* stream);
int sed_scale(float* x, long n, float alpha, void* stream);        /* x *= alpha */

/* ───────────── optimiser (torch.optim.Adam sed.py:159; crnn_lightning.py:195-197; clip train_lightning.py:50) ─────────────
 * sq-norm: norm_out[0] = ||g||_2, norm_out[1] = clip coefficient min(1, max_norm/(norm+1e-6))
 * (1 when max_norm <= 0).  workspace >= sed_sqnorm_workspace_bytes(n). Deterministic. */
size_t sed_sqnorm_workspace_bytes(long n);
int sed_grad_norm_clip_coef(const float* g, long n, float max_norm, float* norm_out,
                            void* workspace, void* stream);
/* Adam with coupled L2 weight decay on a flat arena; grads are multiplied by *grad_scale
 * (device pointer, may be NULL = 1) first.  step >= 1. */
int sed_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, const float* grad_scale,
                  const uint64_t* step_state, void* stream);
/* step_state (may be NULL): device pointer to {dropout salt, optimiser step}; when given, the step count is read from
 * step_state[1] on the device instead of `step`.  sed_step_advance increments both words (one tiny launch per fit
 * step): with it the whole fit step is a replayable hipGraph. */
int sed_step_advance(uint64_t* step_state, void* stream);

/* ───────────── log-mel front end (feature.py:55-59 via librosa.stft / filters.mel) ─────────────
 * pcm [n_samples] mono f32 -> out [n_frames][n_mels] = log(mel @ |STFT|^2), n_frames = 1 + n_samples/hop; frames are
 * centred (librosa center=True): frame f covers samples [f*hop - n_fft/2, f*hop + n_fft/2), padded per pad_mode
 * (0 = zeros, librosa >= 0.10 "constant"; 1 = numpy "reflect", older librosa).  n_fft must be 2048.
 * The constant tables travel as ONE blob that the kernel keeps in LDS.  It is built on the HOST (no GPU call) from
 * host arrays: window [n_fft] and melfb [n_mels][n_fft/2+1] (both computed in double by the caller, librosa's Hann /
 * Slaney bank in sed_crnn_amd/feature.py); the builder adds the FFT twiddles and turns the filterbank into its sparse
 * band-major form (any bank whose non-zeros fit the plan: <= 8192 of them, n_mels <= 128).
 *   sed_logmel_tables_bytes  -> size of the blob for this bank (0 = cannot be planned);
 *   sed_logmel_build_tables  -> fills tables_host; the caller copies it to the device once.
 * sed_logmel: `tables` must be a blob written by sed_logmel_build_tables for this n_mels (the library cannot read device
 * memory to check it; only its size is validated); mu/inv_sigma (device, may both be NULL) fuse feature.py:127-129's
 * StandardScaler: (x-mu)*inv_sigma. */
size_t sed_logmel_tables_bytes(const float* melfb_host, int n_fft, int n_mels);
int sed_logmel_build_tables(const float* window_host, const float* melfb_host, int n_fft, int n_mels,
                            void* tables_host, size_t tables_bytes);
int sed_logmel(const float* pcm, long n_samples, const void* tables, size_t tables_bytes,
               const float* mu, const float* inv_sigma, float* out,
               int n_fft, int hop, int n_mels, int pad_mode, void* stream);

/* ───────────── GPU-resident minibatch assembly (sed.py:64-79; decorte_datamodule.py:39-49,77-111; utils.py:15-41) ─────────────
 * mel [N][C*F] (a whole fold, device-resident; channel c = columns [c*F,(c+1)*F)), lab [N][K].
 * starts [B] window starts (clamped to [0, N-L] like the reference's fallback); tmask/fmask [B][n_masks] SpecAugment
 * offsets (-1 = skip; zeroes time_w frames / freq_w mel bins, all channels) or NULL with n_masks = 0.
 * x [B][C][F][L] (the network input layout), y [B][L/pool][K] = max over each group of `pool` frames. */
int sed_window_batch(const float* mel, const float* lab, long N, int C, int F, int K, const int* starts,
                     const int* tmask, const int* fmask, int n_masks, int time_w, int freq_w,
                     float* x, float* y, int B, int L, int pool, void* stream);
/* utils.split_in_seqs + split_multi_channels in one pass: feat [N][C*F] -> out [N/S][C][F][S] (time_last=1, network
 * input) or [N/S][C][S][F] (time_last=0, the utils.py layout); the N %% S remainder is dropped. */
int sed_pack_sequences(const float* feat, long N, int C, int F, int S, int time_last, float* out, void* stream);
/* StandardScaler.fit (feature.py:127-128): per-column mean and population sigma with sklearn's rules — float64
 * accumulation, centred two-pass variance, and scale 1 for a column whose variance is within that algorithm's rounding
 * bound (sklearn's _is_constant_feature: var <= N*eps*var + (N*mean*eps)^2).  x [N][F] f32, any F (256-column chunks); mean / stdv are
 * FLOAT64 device arrays [F], like sklearn's mean_ / scale_ (a float32 mean would already cost 5 % of a sigma on a column
 * whose mean is 1e6 sigmas). */
size_t sed_col_mean_std_workspace_bytes(int F);
int sed_col_mean_std(const float* x, long N, int F, double* mean, double* stdv, void* workspace, void* stream);
/* StandardScaler.transform (feature.py:128-129): out = f32(f32(x - mean[col]) / stdv[col]) (sklearn's two in-place
 * float32 roundings); mean / stdv float64 [F]; out may alias x. */
int sed_col_standardize(const float* x, long N, int F, const double* mean, const double* stdv, float* out, void* stream);

/* Segment-based metric counts (metrics.py:20-68) and the 2x2 confusion counts (crnn_lightning.py:112-119) of thresholded
 * predictions on the device: pred/lab [rows][K] (windows concatenated, K <= 32; labels are 0/1), 1-second blocks of
 * `block` rows.  counts17 (uint64, device):
 * [0..5] frame-wise TP,Nref,Nsys,S,D,I; [6..8] TP,Nref,Nsys over ceil(rows/block) blocks (F1 keeps the partial block);
 * [9..12] S,D,I,Nref over floor(rows/block) blocks (ER drops it); [13..16] element-wise tn,fp,fn,tp with the
 * reference's uint8 truncation of the label (label in [0,1) counts as 0, [1,2) as 1).  Integer arithmetic: exact. */
#define SED_SEGMENT_COUNTS 17
int sed_segment_counts(const float* pred, const float* lab, long rows, int K, int block, float threshold,
                       unsigned long long* counts17, void* stream);

/* ───────────── whole-network plan (TimePooledCRNN.forward sed.py:105-112 / crnn_lightning.py:66-73) ───────────── */
typedef struct sed_net_cfg {
    int B, Cin, F, T;                 /* input x [B][Cin][F][T] */
    int n_conv;
    int C[SED_MAX_CONV];              /* conv channels */
    int pool_f[SED_MAX_CONV], pool_t[SED_MAX_CONV];
    float drop_p[SED_MAX_CONV];       /* dropout after block i (0 = none) */
    int n_gru;
    int H[SED_MAX_GRU];               /* hidden size per direction of each stacked BiGRU layer */
    int n_dense;
    int D[SED_MAX_DENSE];             /* dense sizes; ReLU between layers, last = classes (logits) */
    float bn_eps, bn_momentum;
    int conv_mode;                    /* 0 = exact fp32 (default); 1 = EXPERIMENT: conv forward, data gradient and weight gradient of the
                                         MFMA blocks on the 3-term bf16 split (sed_conv3x3_fwd_ex / sed_conv3x3_wgrad_ex) */
    int flags;                        /* 0 (default) | SED_NET_* below: measurement / test switches of the backward schedule */
} sed_net_cfg;
/* The last backward phase runs the first block's passes (auxiliary stream) BESIDE the deferred MFMA weight gradients (main
 * stream); the persistent MFMA kernel must hold its CUs before the passes move in.  That order is a DEPENDENCY: the kernel's
 * workgroups announce themselves on a counter in the workspace and a one-wave gate at the head of the auxiliary chain waits
 * for them.  SED_NET_AUX_FIRST (tests): the host enqueues the auxiliary chain before the main-stream kernel — the adversarial
 * order, which must give the same gradients at the same speed.  SED_NET_NO_GATE (A/B measurements): no gate. */
#define SED_NET_AUX_FIRST 0x1
#define SED_NET_NO_GATE 0x2
#define SED_NET_DIRECT_CONV 0x4     /* the 128-channel blocks on the direct 3x3 kernels instead of the Winograd form (A/B measurements, tests) */

typedef struct sed_net_params {      /* pointers in the reference's own layouts */
    float* conv_w[SED_MAX_CONV];      /* [C][Cin][3][3] */
    float* conv_b[SED_MAX_CONV];
    float* bn_g[SED_MAX_CONV];
    float* bn_b[SED_MAX_CONV];
    float* bn_rm[SED_MAX_CONV];       /* running stats (params struct only; unused in grads) */
    float* bn_rv[SED_MAX_CONV];
    float* gru_wih[SED_MAX_GRU][2];   /* [3H][in]  (dir 1 = *_reverse) */
    float* gru_whh[SED_MAX_GRU][2];   /* [3H][H] */
    float* gru_bih[SED_MAX_GRU][2];
    float* gru_bhh[SED_MAX_GRU][2];
    float* dense_w[SED_MAX_DENSE];    /* [out][in] */
    float* dense_b[SED_MAX_DENSE];
} sed_net_params;

/* Output time steps / features of the conv stack for cfg (T', F'); returns <0 on a bad cfg. */
int sed_net_out_shape(const sed_net_cfg* cfg, int* Tp, int* Fp);
size_t sed_net_workspace_bytes(const sed_net_cfg* cfg, int training);
/* logits [B][T'][D_last].  training=1: batch statistics, dropout, activations kept in `workspace`
 * for sed_net_backward; running stats updated in place. */
int sed_net_forward(const sed_net_cfg* cfg, const sed_net_params* p, const float* x, float* logits,
                    void* workspace, int training, uint64_t seed, const uint64_t* seed_dev, void* stream);
/* Gradients of every parameter into `g` (same layouts as `p`; written, not accumulated).
 * Stages let the host overlap the gradient all-reduce with the rest of backward:
 *   stage 0 = dense head + GRU stack, stage s>=1 = conv block l = n_conv - s.  Run stages
 *   [stage_begin, stage_end) in increasing order; (0, n_conv+1) = everything.
 * The critical chain of the conv backward is dgrad(top) -> BN(top-1) -> ... -> dgrad(1) -> BN(0); the weight gradients
 * hang off it and are deferred to the stage of block 1, where all of them run back to back.  Hence the gradients of
 * stage 0 are complete on `stream` when stage 0 returns, and those of conv block l when stage
 * sed_net_backward_ready_stage(cfg, l) returns (block 0: the last stage; every other block: the stage of block 1).
 * aux_stream (NULL or == stream: serial): a second hipStream_t for the HBM-bound BatchNorm/ReLU/pool backward passes
 * that have MFMA-bound work to hide behind: BN(top) beside the GRU weight-gradient GEMM (issued by stage 0), BN(0) (for
 * a fused first block: all of block 0) beside the conv weight gradients (issued by the stage of block 1); it also takes
 * the weight gradients that have a latency- or HBM-bound pass of the data-gradient chain to run beside: the small GRU
 * weight-gradient GEMMs of the layers above the first (beside the recurrences) and the top conv block's weight gradient
 * (beside the BatchNorm backward of the block below it).  hipEvents
 * order the two streams and every auxiliary launch is joined back into `stream` before the last stage returns.
 * Stages must therefore be run in order with the same aux_stream for the whole backward. */
int sed_net_backward(const sed_net_cfg* cfg, const sed_net_params* p, const sed_net_params* g,
                     const float* x, const float* dlogits, void* workspace, uint64_t seed, const uint64_t* seed_dev,
                     int stage_begin, int stage_end, void* stream, void* aux_stream);
/* Stage of sed_net_backward after which every gradient of conv block `block` is complete (<0: bad arguments). */
int sed_net_backward_ready_stage(const sed_net_cfg* cfg, int block);

/* Phased execution, for synchronised BatchNorm in data-parallel training: the per-block statistics
 * are the only cross-sample coupling besides the loss mean, so the plan can stop at them.
 *   forward phases : 2l   = conv block l + its statistic sums (training),
 *                    2l+1 = finalise with count*count_scale, normalise/ReLU/pool/dropout,
 *                    2*n_conv = GRU stack + dense head.
 *   backward phases: 0    = dense head + GRU stack,
 *                    2k+1 = BatchNorm-backward sums (sum g, sum g*xhat) of block l = n_conv-1-k,
 *                    2k+2 = rest of block l (apply, weight/bias/data gradients); its gradients are then complete.
 * Between phase 2l and 2l+1 (forward) / 2k+1 and 2k+2 (backward) the host all-reduces (SUM) the workspace region
 * sed_net_sync_region() names; count_scale = number of ranks (1 = unsynchronised).  Single stream, no overlap. */
int sed_net_sync_region(const sed_net_cfg* cfg, int backward, int block, size_t* offset_bytes, size_t* n_floats);
int sed_net_forward_phases(const sed_net_cfg* cfg, const sed_net_params* p, const float* x, float* logits,
                           void* workspace, int training, uint64_t seed, int phase_begin, int phase_end,
                           float count_scale, void* stream);
int sed_net_backward_phases(const sed_net_cfg* cfg, const sed_net_params* p, const sed_net_params* g,
                            const float* x, const float* dlogits, void* workspace, uint64_t seed,
                            int phase_begin, int phase_end, float count_scale, void* stream);

/* Where an intermediate of the plan lives inside `workspace` (offsets are a pure function of cfg and `training`), for hosts
 * that read activations and for parity tests of intermediates.  name / index:
 *   "conv_out"[l] conv output of block l, channels-last [B][T_l][F_l][C] (absent for a recomputed first block);
 *   "pooled"[l] block output [B][T_l/pt][F_l/pf][C] (last block: [B][T'][C][F'], the GRU feature order);
 *   "mean" / "rstd" / "scale" / "shift"[l] the batch statistics and fused BatchNorm coefficients of block l ([C]);
 *   "gi"[i] / "gru_out"[i] input projections [M][2][3H] and outputs [M][2H] of GRU layer i;
 *   training only: "dconv"[l] gradient of block l's conv output, "dgru_out"[i], "grad_act"[0] the buffer that carries the
 *   gradient of the pooled output being back-propagated (reused from block to block), "bn_sums_bwd"[0] (sum g, sum g*xhat),
 *   "wgrad_zero_row"[0|1] the zero-filled rows at the head of the weight-gradient scratch of the main / auxiliary stream
 *   (SED_WGRAD_ZERO_ROW_CLEAN: sed_net_backward re-clears them in every call that holds stage 0).
 * <0: unknown name / index for this plan. */
int sed_net_workspace_region(const sed_net_cfg* cfg, int training, const char* name, int index,
                             size_t* offset_bytes, size_t* n_floats);

/* The ReLU-gate / arg-max decisions (codes of sed_bn_relu_pool_route) of conv block `block` in the last TRAINING forward that
 * ran on `workspace` with input `x`; route: caller-owned [B][T_l/pt][F_l/pf][C] bytes.  Dispatches to sed_bn_relu_pool_route
 * (stored conv output) or sed_conv1_route (recomputed first block: it re-reads p->conv_b, so call this before the optimiser
 * moves the parameters).  For the parity tests of the gradient routing of
 * `loss.backward()` (sed.py:137): see oracle/crnn_ref.py forward_routed. */
int sed_net_routing(const sed_net_cfg* cfg, const sed_net_params* p, const float* x, const void* workspace, int block,
                    unsigned char* route, void* stream);

/* ───────────── in-library kernel timers — MEASUREMENT ONLY, off by default ─────────────
 * The one exception to the conventions at the top of this header: this group keeps process-wide mutable state (the
 * enabled mask and the recorded event pairs, guarded by a mutex) and sed_prof_read SYNCHRONISES (hipEventSynchronize
 * on every recorded pair).  Nothing in the product path enables it; bench.py does, for the `roofline` object.
 * When a tag's bit is set in `tag_mask`, every launch of that kernel family is bracketed by a pair of
 * hipEvents on the launch stream and tagged with its algorithmic work (FLOPs for the MFMA kernels,
 * bytes for the streaming ones).  sed_prof_read waits for the recorded events and returns the totals
 * since the last sed_prof_enable.  With the mask at 0 (the default) every entry is exactly as documented above:
 * no event is recorded and no state is touched. */
enum sed_kernel_tag {
    SED_K_CONV_MFMA_FWD = 0,   /* units: FLOPs */
    SED_K_CONV_SMALL_FWD,      /* units: bytes */
    SED_K_CONV_MFMA_WGRAD,     /* units: FLOPs */
    SED_K_CONV_SMALL_WGRAD,    /* units: bytes */
    SED_K_BN_FWD,              /* units: bytes */
    SED_K_BN_BWD_REDUCE,       /* units: bytes */
    SED_K_BN_BWD_APPLY,        /* units: bytes */
    SED_K_GEMM,                /* units: FLOPs */
    SED_K_GRU_FWD,             /* units: FLOPs */
    SED_K_GRU_BWD,             /* units: FLOPs */
    SED_K_ADAM,                /* units: bytes */
    SED_K_LOGMEL,              /* units: bytes (hop*4 read + n_mels*4 written per frame) */
    SED_K_CONV_MFMA_DGRAD,     /* units: FLOPs — sed_conv3x3_dgrad_bnred (the forward kernel with the BatchNorm-backward epilogue) */
    SED_K_COUNT
};
int sed_prof_enable(unsigned tag_mask);
int sed_prof_read(int tag, double* total_ms, long* launches, double* total_units);
const char* sed_prof_tag_name(int tag);
int sed_prof_tag_count(void);                /* = SED_K_COUNT of the loaded library */

#ifdef __cplusplus
}
#endif
#endif /* SEDCRNN_H */
