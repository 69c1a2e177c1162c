"""Per-kernel Python entry points over the C ABI (torch tensors in, torch tensors out).

These are thin: they allocate outputs/workspaces with torch (device memory plumbing) and call
libsedcrnn.so on torch's current stream.  Used by the parity tests and by feature extraction; the
network itself goes through the whole-network plan (`model.py` -> sed_net_forward/backward).
"""
import ctypes as C

import torch

from ._lib import check, lib, ptr, stream_ptr


def _f32c(t):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "expect contiguous fp32 CUDA tensor"
    return t


def conv3x3_pack(w, mode=0):
    Cout, Cin = w.shape[:2]
    wf = torch.empty(9, Cout, Cin, device=w.device)
    wd = torch.empty(9, Cin, Cout, device=w.device)
    check(lib().sed_conv3x3_pack_weights_ex(ptr(_f32c(w)), ptr(wf), ptr(wd), Cout, Cin, mode, stream_ptr()), "conv3x3_pack")
    return wf, wd


def conv3x3_fwd(x, wp, bias, x_is_nchw, want_stats=True, mode=0):
    """x: [B,Cin,F,T] if x_is_nchw else [B,T,F,Cin]; wp [9,Cout,Cin] -> y [B,T,F,Cout] (+ stat partials)."""
    if x_is_nchw:
        B, Cin, F, T = x.shape
    else:
        B, T, F, Cin = x.shape
    Cout = wp.shape[1]
    y = torch.empty(B, T, F, Cout, device=x.device)
    stat = None
    if want_stats:
        rows = lib().sed_conv3x3_stat_rows(B, Cin, F, T, Cout, int(x_is_nchw))
        stat = torch.zeros(max(rows, 1), 2, Cout, device=x.device)
    check(lib().sed_conv3x3_fwd_ex(ptr(_f32c(x)), int(x_is_nchw), ptr(_f32c(wp)), ptr(bias), ptr(y), ptr(stat),
                                   B, Cin, F, T, Cout, mode, stream_ptr()), "conv3x3_fwd")
    return y, stat


def conv3x3_wino_pack(w):
    """w [Cout,Cin,3,3] -> (uf, ud): the Winograd F(2x2,3x3) transformed weights of the forward and of the data gradient"""
    Cout, Cin = w.shape[:2]
    n = lib().sed_conv3x3_wino_packed_floats(Cout, Cin)
    uf, ud = torch.empty(n, device=w.device), torch.empty(n, device=w.device)
    check(lib().sed_conv3x3_wino_pack_weights(ptr(_f32c(w)), ptr(uf), ptr(ud), Cout, Cin, stream_ptr()), "conv3x3_wino_pack")
    return uf, ud


def conv3x3_wino_fwd(x, up, bias, Cout, want_stats=True):
    """x [B,T,F,128] channels-last, up from conv3x3_wino_pack -> y [B,T,F,Cout] (+ statistic partial rows)"""
    B, T, F, Cin = x.shape
    rows = lib().sed_conv3x3_wino_rows(B, Cin, F, T, Cout)
    if rows <= 0:
        raise ValueError("conv3x3_wino_fwd: shape not supported")
    y = torch.empty(B, T, F, Cout, device=x.device)
    stat = torch.zeros(rows, 2, Cout, device=x.device) if want_stats else None
    check(lib().sed_conv3x3_wino_fwd(ptr(_f32c(x)), ptr(_f32c(up)), ptr(bias), ptr(y), ptr(stat), B, Cin, F, T, Cout, stream_ptr()),
          "conv3x3_wino_fwd")
    return y, stat


def conv3x3_bn_relu_pool_eval(x, w, bias, gamma, beta, running_mean, running_var, eps=1e-5, wino=False):
    """inference: x [B,T,F,Cin] channels-last -> relu(max_pool((1,2))(bn_eval(conv3x3(x)))) [B,T//2,F,Cout] in one launch
    (BatchNorm folded into the packed weights; sed_conv3x3_pack_weights_bn_folded + sed_conv3x3_bn_relu_pool_eval; wino: the
    Winograd forms of both)"""
    B, T, F, Cin = x.shape
    Cout = w.shape[0]
    if wino:
        if lib().sed_conv3x3_wino_rows(B, Cin, F, T, Cout) <= 0:
            raise ValueError("conv3x3_bn_relu_pool_eval: shape not supported by the Winograd kernel")
        uf = torch.empty(lib().sed_conv3x3_wino_packed_floats(Cout, Cin), device=x.device)
        bf = torch.empty(Cout, device=x.device)
        check(lib().sed_conv3x3_wino_pack_weights_bn_folded(ptr(_f32c(w)), ptr(bias), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                                            eps, ptr(uf), ptr(bf), Cout, Cin, stream_ptr()), "conv3x3_wino_pack_weights_bn_folded")
        out = torch.empty(B, T // 2, F, Cout, device=x.device)
        check(lib().sed_conv3x3_wino_bn_relu_pool_eval(ptr(_f32c(x)), ptr(uf), ptr(bf), ptr(out), B, Cin, F, T, Cout, stream_ptr()),
              "conv3x3_wino_bn_relu_pool_eval")
        return out
    if not lib().sed_conv3x3_bn_relu_pool_eval_supported(B, Cin, F, T, Cout):
        raise ValueError("conv3x3_bn_relu_pool_eval: shape not supported")
    wf = torch.empty(9, Cout, Cin, device=x.device)
    bf = torch.empty(Cout, device=x.device)
    check(lib().sed_conv3x3_pack_weights_bn_folded(ptr(_f32c(w)), ptr(bias), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                                   eps, ptr(wf), ptr(bf), Cout, Cin, stream_ptr()), "conv3x3_pack_weights_bn_folded")
    out = torch.empty(B, T // 2, F, Cout, device=x.device)
    check(lib().sed_conv3x3_bn_relu_pool_eval(ptr(_f32c(x)), ptr(wf), ptr(bf), ptr(out), B, Cin, F, T, Cout, stream_ptr()),
          "conv3x3_bn_relu_pool_eval")
    return out


def conv3x3_wgrad(x, dy, x_is_nchw, mode=0):
    if x_is_nchw:
        B, Cin, F, T = x.shape
    else:
        B, T, F, Cin = x.shape
    Cout = dy.shape[-1]
    nbytes = lib().sed_conv3x3_wgrad_workspace_bytes(B, Cin, F, T, Cout)
    ws = torch.empty(nbytes // 4 + 1, device=x.device)
    dw = torch.empty(Cout, Cin, 3, 3, device=x.device)
    check(lib().sed_conv3x3_wgrad_ex(ptr(_f32c(x)), int(x_is_nchw), ptr(_f32c(dy)), ptr(dw), ptr(ws),
                                     B, Cin, F, T, Cout, mode, stream_ptr()), "conv3x3_wgrad")
    return dw


def bn_finalize_train(stat, count, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5):
    rows, _, Cc = stat.shape
    mean, rstd, scale, shift = (torch.empty(Cc, device=stat.device) for _ in range(4))
    check(lib().sed_bn_finalize_train(ptr(stat), rows, Cc, float(count), ptr(gamma), ptr(beta), ptr(running_mean),
                                      ptr(running_var), momentum, eps, ptr(mean), ptr(rstd), ptr(scale), ptr(shift),
                                      stream_ptr()), "bn_finalize_train")
    return mean, rstd, scale, shift


def bn_finalize_eval(gamma, beta, rm, rv, eps=1e-5):
    Cc = gamma.numel()
    scale, shift = torch.empty(Cc, device=gamma.device), torch.empty(Cc, device=gamma.device)
    check(lib().sed_bn_finalize_eval(ptr(gamma), ptr(beta), ptr(rm), ptr(rv), eps, Cc, ptr(scale), ptr(shift),
                                     stream_ptr()), "bn_finalize_eval")
    return scale, shift


def bn_relu_pool_drop_fwd(y, scale, shift, pool_f, pool_t, out_tcf=False, drop_p=0.0, seed=0):
    B, T, F, Cc = y.shape
    Tp, Fp = T // pool_t, F // pool_f
    out = torch.empty((B, Tp, Cc, Fp) if out_tcf else (B, Tp, Fp, Cc), device=y.device)
    check(lib().sed_bn_relu_pool_drop_fwd(ptr(_f32c(y)), ptr(scale), ptr(shift), ptr(out), B, T, F, Cc, pool_f, pool_t,
                                          int(out_tcf), drop_p, seed, None, stream_ptr()), "bn_relu_pool_drop_fwd")
    return out


def bn_relu_pool_drop_bwd(y, dout, scale, shift, mean, rstd, pool_f, pool_t, out_tcf=False, drop_p=0.0, seed=0):
    """-> (dy [B,T,F,C], dgamma, dbeta, dbias)"""
    B, T, F, Cc = y.shape
    rows = lib().sed_bn_bwd_rows(B, T, pool_t)
    part = torch.empty(rows, 2, Cc, device=y.device)
    check(lib().sed_bn_relu_pool_drop_bwd_reduce(ptr(_f32c(y)), ptr(_f32c(dout)), ptr(scale), ptr(shift), ptr(mean),
                                                 ptr(rstd), ptr(part), B, T, F, Cc, pool_f, pool_t, int(out_tcf),
                                                 drop_p, seed, None, stream_ptr()), "bn_bwd_reduce")
    sum_g, sum_gx, dgamma, dbeta = (torch.empty(Cc, device=y.device) for _ in range(4))
    check(lib().sed_bn_bwd_finalize(ptr(part), rows, Cc, ptr(sum_g), ptr(sum_gx), ptr(dgamma), ptr(dbeta),
                                    stream_ptr()), "bn_bwd_finalize")
    dy = torch.empty_like(y)
    dbp = torch.empty(rows, Cc, device=y.device)
    check(lib().sed_bn_relu_pool_drop_bwd_apply(ptr(y), ptr(dout), ptr(scale), ptr(shift), ptr(mean), ptr(rstd),
                                                ptr(sum_g), ptr(sum_gx), ptr(dy), ptr(dbp), B, T, F, Cc, pool_f,
                                                pool_t, int(out_tcf), drop_p, seed, None, stream_ptr()), "bn_bwd_apply")
    dbias = torch.empty(Cc, device=y.device)
    check(lib().sed_reduce_rows(ptr(dbp), rows, Cc, Cc, ptr(dbias), stream_ptr()), "reduce_rows")
    return dy, dgamma, dbeta, dbias


def bn_bwd_sums_from_pooled(pooled, dout, gamma, beta, y, mean, rstd, pool_f, pool_t, drop_p=0.0, scale=None, shift=None):
    """(sum g, sum g*xhat) of the GRU-feeding block from its pooled output [B,Tp,C,Fp] and that tensor's gradient
    (sed_bn_bwd_reduce_pooled + sed_bn_bwd_finalize); y [B,T,F,C] is read for gamma == 0 channels only"""
    B, T, F, Cc = y.shape
    if not lib().sed_bn_bwd_reduce_pooled_supported(F, Cc, pool_f, pool_t, 1):
        raise ValueError("bn_bwd_sums_from_pooled: shape not supported")
    rows = lib().sed_bn_bwd_rows(B, T, pool_t)
    part = torch.empty(rows, 2, Cc, device=y.device)
    if scale is None:                                  # the fused coefficients as sed_bn_finalize_train forms them
        scale = gamma * rstd
        shift = beta - mean * scale
    check(lib().sed_bn_bwd_reduce_pooled(ptr(_f32c(pooled)), ptr(_f32c(dout)), ptr(gamma), ptr(beta), ptr(_f32c(y)), ptr(mean),
                                         ptr(rstd), ptr(scale), ptr(shift), ptr(part), B, T, F, Cc, pool_f, pool_t, 1, drop_p, stream_ptr()),
          "bn_bwd_reduce_pooled")
    sum_g, sum_gx = torch.empty(Cc, device=y.device), torch.empty(Cc, device=y.device)
    check(lib().sed_bn_bwd_finalize(ptr(part), rows, Cc, ptr(sum_g), ptr(sum_gx), None, None, stream_ptr()), "bn_bwd_finalize")
    return sum_g, sum_gx


def conv3x3_dgrad_bnred(dy, wd, pooled, gamma, beta, y_below, mean, rstd, pool_f, pool_t, drop_p=0.0, scale=None, shift=None, wino=False):
    """data gradient of a conv block + the BatchNorm-backward sums of the block below in its epilogue
    -> (dx [B,T,F,Cin], sum_g [Cin], sum_gx [Cin]); dy [B,T,F,C], wd = the dgrad packing [9,Cin,C] (wino: the `ud` of
    conv3x3_wino_pack and the Winograd kernel)"""
    B, T, F, Cc = dy.shape
    Cin = pooled.shape[-1]
    rows = lib().sed_conv3x3_wino_rows(B, Cc, F, T, Cin) if wino else lib().sed_conv3x3_dgrad_bnred_rows(B, Cc, F, T, Cin)
    assert rows > 0, "shape does not take the fused MFMA path"
    dx = torch.empty(B, T, F, Cin, device=dy.device)
    part = torch.empty(rows, 2, Cin, device=dy.device)
    if scale is None:
        scale = gamma * rstd
        shift = beta - mean * scale
    entry = lib().sed_conv3x3_wino_dgrad_bnred if wino else lib().sed_conv3x3_dgrad_bnred
    check(entry(ptr(_f32c(dy)), ptr(_f32c(wd)), ptr(dx), ptr(part), ptr(_f32c(pooled)), ptr(gamma), ptr(beta),
                ptr(_f32c(y_below)), ptr(mean), ptr(rstd), drop_p, pool_f, pool_t, y_below.shape[2],
                y_below.shape[1], B, Cc, F, T, Cin, stream_ptr()), "conv3x3_dgrad_bnred")
    sum_g, sum_gx = torch.empty(Cin, device=dy.device), torch.empty(Cin, device=dy.device)
    # (a stored conv output below: the small-|gamma| channels are finished from it by the finalising kernel)
    check(lib().sed_bn_bwd_finalize_small_gamma(ptr(part), rows, Cin, ptr(sum_g), ptr(sum_gx), None, None, ptr(dx), ptr(pooled),
                                                ptr(y_below), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(scale), ptr(shift),
                                                B, y_below.shape[1], y_below.shape[2], pool_f, pool_t, drop_p, stream_ptr()),
          "bn_bwd_finalize_small_gamma")
    return dx, sum_g, sum_gx


def gemm(A, B, bias=None, out=None, beta=0.0):
    """out = A @ B (+bias) (+beta*out) for 2-D fp32 CUDA tensors with arbitrary (unit-along-one-axis) strides."""
    M, K = A.shape
    K2, N = B.shape
    assert K == K2
    if out is None:
        out = torch.empty(M, N, device=A.device)
    check(lib().sed_gemm_f32(ptr(A), A.stride(0), A.stride(1), ptr(B), B.stride(0), B.stride(1), ptr(out),
                             out.stride(0), ptr(bias), beta, M, N, K, stream_ptr()), "gemm_f32")
    return out


_GEMM_WS = {}


def gemm_ws(A, B, out=None, bias=None, wgrad=False):
    """out = A @ B (+bias) through the split-K capable entries (workspace cached here); wgrad=True: the weight-gradient
    entry, which may also cut a one-tile-per-CU product into two K-slices"""
    M, K = A.shape
    _, N = B.shape
    if out is None:
        out = torch.empty(M, N, device=A.device)
    nb = lib().sed_gemm_f32_workspace_bytes(M, N, K)
    key = (A.device, nb)
    ws = _GEMM_WS.get(key) if nb else None
    if nb and ws is None:
        ws = _GEMM_WS[key] = torch.empty(nb // 4 + 1, device=A.device)
    if wgrad:
        assert bias is None
        check(lib().sed_gemm_f32_wgrad(ptr(A), A.stride(0), A.stride(1), ptr(B), B.stride(0), B.stride(1), ptr(out),
                                       out.stride(0), M, N, K, ptr(ws), stream_ptr()), "gemm_f32_wgrad")
        return out
    check(lib().sed_gemm_f32_ws(ptr(A), A.stride(0), A.stride(1), ptr(B), B.stride(0), B.stride(1), ptr(out),
                                out.stride(0), ptr(bias), M, N, K, ptr(ws), stream_ptr()), "gemm_f32_ws")
    return out


def linear_fwd(x, W, b, relu=False):
    M, K = x.shape
    N = W.shape[0]
    y = torch.empty(M, N, device=x.device)
    check(lib().sed_linear_fwd(ptr(_f32c(x)), ptr(_f32c(W)), ptr(b), ptr(y), M, K, N, int(relu), stream_ptr()), "linear_fwd")
    return y


def linear_bwd(x, W, y, dy, relu=False, need_dx=True):
    M, K = x.shape
    N = W.shape[0]
    ws = torch.empty(lib().sed_linear_bwd_workspace_bytes(M, K, N) // 4 + 1, device=x.device)
    dx = torch.empty_like(x) if need_dx else None
    dW, db = torch.empty_like(W), torch.empty(N, device=x.device)
    dy = dy.clone()
    check(lib().sed_linear_bwd(ptr(x), ptr(W), ptr(y), ptr(dy), ptr(dx), ptr(dW), ptr(db), ptr(ws), M, K, N,
                               int(relu), stream_ptr()), "linear_bwd")
    return dx, dW, db


def _pp(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def gru_seq_fwd(gi, whh, bhh, save=True):
    """gi [B,T,2,3H]; whh/bhh: pairs (forward, reverse) -> out [B,T,2H], saved [B,T,2,5,H]"""
    B, T, _, H3 = gi.shape
    H = H3 // 3
    out = torch.empty(B, T, 2 * H, device=gi.device)
    saved = torch.empty(B, T, 2, 5, H, device=gi.device) if save else None
    ws = torch.empty(lib().sed_gru_seq_workspace_bytes(H) // 4, device=gi.device)
    check(lib().sed_gru_seq_fwd(ptr(_f32c(gi)), _pp(whh), _pp(bhh), ptr(out), ptr(saved), ptr(ws), B, T, H,
                                stream_ptr()), "gru_seq_fwd")
    return out, saved


def gru_seq_bwd(dout, saved, whh, want_bias=False):
    B, T, H2 = dout.shape
    H = H2 // 2
    dgi = torch.empty(B, T, 2, 3 * H, device=dout.device)
    dgh = torch.empty(B, T, 2, 3 * H, device=dout.device)
    if not want_bias:
        check(lib().sed_gru_seq_bwd(ptr(_f32c(dout)), ptr(saved), _pp(whh), ptr(dgi), ptr(dgh), None, None, None,
                                    B, T, H, stream_ptr()), "gru_seq_bwd")
        return dgi, dgh
    dbih = [torch.empty(3 * H, device=dout.device) for _ in range(2)]
    dbhh = [torch.empty(3 * H, device=dout.device) for _ in range(2)]
    ws = torch.empty(lib().sed_gru_seq_bwd_workspace_bytes(B, H) // 4 + 1, device=dout.device)
    check(lib().sed_gru_seq_bwd(ptr(_f32c(dout)), ptr(saved), _pp(whh), ptr(dgi), ptr(dgh), _pp(dbih), _pp(dbhh),
                                ptr(ws), B, T, H, stream_ptr()), "gru_seq_bwd")
    return dgi, dgh, dbih, dbhh


def loss_fwd_bwd(logits, targets, kind="bce", alpha=0.25, gamma=2.0, reduction="mean"):
    """-> (loss scalar tensor, dlogits, probs)"""
    if kind not in ("bce", "focal"):
        raise ValueError(f"loss kind must be 'bce' or 'focal', got {kind!r}")
    if reduction not in ("mean", "sum"):
        raise ValueError(f"reduction must be 'mean' or 'sum' (the fused loss returns a scalar), got {reduction!r}")
    if logits.shape != targets.shape:
        raise ValueError(f"logits {tuple(logits.shape)} and targets {tuple(targets.shape)} differ in shape")
    n = logits.numel()
    loss = torch.empty(1, device=logits.device)
    d = torch.empty_like(logits)
    p = torch.empty_like(logits)
    check(lib().sed_loss_fwd_bwd(ptr(_f32c(logits)), ptr(_f32c(targets)), n, 0 if kind == "bce" else 1, alpha, gamma,
                                 int(reduction == "mean"), ptr(loss), ptr(d), ptr(p), stream_ptr()), "loss_fwd_bwd")
    return loss, d, p


def sigmoid(x):
    y = torch.empty_like(x)
    check(lib().sed_sigmoid(ptr(_f32c(x)), ptr(y), x.numel(), stream_ptr()), "sigmoid")
    return y


def grad_norm_clip_coef(g, max_norm):
    """-> tensor [norm, coef] on device (no host sync)"""
    out = torch.empty(2, device=g.device)
    ws = torch.empty(lib().sed_sqnorm_workspace_bytes(g.numel()) // 4, device=g.device)
    check(lib().sed_grad_norm_clip_coef(ptr(g), g.numel(), float(max_norm), ptr(out), ptr(ws), stream_ptr()), "grad_norm")
    return out


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=None, step_state=None):
    check(lib().sed_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                              ptr(grad_scale), ptr(step_state), stream_ptr()), "adam_step")


def conv1_fused_block(x, w, bias, gamma, beta, pool_f, pool_t, dout=None, drop_p=0.0, seed=0, eps=1e-5, moments_path=False):
    """First conv block through the recompute-fused entries (train statistics).
    x [B,Cin,F,T] -> pooled [B,T/pt,F/pf,C]; with dout also (dw, dbias, dgamma, dbeta).
    moments_path: the weight gradient from the pooled output, the arg-max bits and the input moments (sed_conv1_bwd_wgrad)
    instead of the recomputing apply pass (sed_conv1_bwd_apply_wgrad)."""
    B, Cin, F, T = x.shape
    Cc = w.shape[0]
    L = lib()
    assert L.sed_conv1_fused_supported(Cin, F, T, Cc, pool_f, pool_t) or (moments_path and L.sed_conv1_rgrad_supported(Cin, F, T, Cc, pool_f, pool_t))
    wf, _ = conv3x3_pack(w)
    rows = L.sed_conv1_fused_rows(B, T)
    stat = torch.empty(1, 2, Cc, device=x.device)
    sws = torch.empty(L.sed_conv1_stats_workspace_bytes(B, Cin, T) // 4 + 1, device=x.device)
    mom = torch.empty(L.sed_conv1_moments_doubles(Cin), device=x.device, dtype=torch.float64) if moments_path else None
    check(L.sed_conv1_stats(ptr(_f32c(x)), ptr(wf), ptr(bias), ptr(stat), ptr(sws), B, Cin, F, T, Cc, ptr(mom), stream_ptr()), "conv1_stats")
    rm, rv = torch.zeros(Cc, device=x.device), torch.ones(Cc, device=x.device)
    mean, rstd, scale, shift = bn_finalize_train(stat, B * T * F, gamma, beta, rm, rv, eps=eps)
    out = torch.empty(B, T // pool_t, F // pool_f, Cc, device=x.device)
    bits = torch.empty(out.numel() // 4, device=x.device, dtype=torch.uint8) if moments_path else None
    check(L.sed_conv1_bn_relu_pool_drop_fwd(ptr(x), ptr(wf), ptr(bias), ptr(scale), ptr(shift), ptr(out), B, Cin, F, T, Cc,
                                            pool_f, pool_t, drop_p, seed, None, ptr(bits), stream_ptr()), "conv1_fwd")
    if dout is None:
        return out
    sum_g, sum_gx, dgamma, dbeta = (torch.empty(Cc, device=x.device) for _ in range(4))
    if Cin <= 2:
        part = torch.empty(rows, 2, Cc, device=x.device)
        check(L.sed_conv1_bwd_reduce(ptr(x), ptr(wf), ptr(bias), ptr(_f32c(dout)), ptr(scale), ptr(shift), ptr(mean), ptr(rstd),
                                     ptr(part), B, Cin, F, T, Cc, pool_f, pool_t, drop_p, seed, None, stream_ptr()), "conv1_bwd_reduce")
        check(L.sed_bn_bwd_finalize(ptr(part), rows, Cc, ptr(sum_g), ptr(sum_gx), ptr(dgamma), ptr(dbeta), stream_ptr()), "bn_bwd_finalize")
    else:
        # beyond two input channels there is no recomputing reduce pass (in the network these sums come out of the data gradient
        # of the block above): here the conv output is materialised once and reduced by the stand-alone pass
        y, _ = conv3x3_fwd(x, wf, bias, True, want_stats=False)
        _, dgamma, dbeta, _ = bn_relu_pool_drop_bwd(y, _f32c(dout), scale, shift, mean, rstd, pool_f, pool_t, drop_p=drop_p, seed=seed)
        sum_g, sum_gx = dbeta.clone(), dgamma.clone()
    dw, db = torch.empty_like(w), torch.empty(Cc, device=x.device)
    if moments_path:
        assert L.sed_conv1_rgrad_supported(Cin, F, T, Cc, pool_f, pool_t)
        ws = torch.empty(L.sed_conv1_bwd_wgrad_workspace_bytes(B, Cin, T, Cc) // 4 + 1, device=x.device)
        check(L.sed_conv1_bwd_wgrad(ptr(x), ptr(dout), ptr(out), ptr(bits), ptr(mom), ptr(wf), ptr(bias), ptr(mean), ptr(rstd),
                                    ptr(scale), ptr(sum_g), ptr(sum_gx), ptr(dw), ptr(db), ptr(ws), B, Cin, F, T, Cc, drop_p,
                                    ptr(gamma), ptr(beta), ptr(dgamma), stream_ptr()), "conv1_bwd_wgrad")
        return out, dw, db, dgamma, dbeta
    ws = torch.empty(L.sed_conv1_bwd_apply_workspace_bytes(B, Cin, T, Cc) // 4 + 1, device=x.device)
    check(L.sed_conv1_bwd_apply_wgrad(ptr(x), ptr(wf), ptr(bias), ptr(dout), ptr(scale), ptr(shift), ptr(mean), ptr(rstd),
                                      ptr(sum_g), ptr(sum_gx), ptr(dw), ptr(db), ptr(ws), B, Cin, F, T, Cc, pool_f, pool_t,
                                      drop_p, seed, None, None, None, None, stream_ptr()), "conv1_bwd_apply_wgrad")
    return out, dw, db, dgamma, dbeta
