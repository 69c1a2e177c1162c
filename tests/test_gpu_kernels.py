"""Per-kernel parity: every HIP kernel of libsedcrnn.so (called through the C ABI) against a plain
torch fp32 CPU reference of the same op.  Tolerances are stated per test; fp32 everywhere."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sed_crnn_amd import ops as o
    return o


def g(t):
    return t.cuda().contiguous()


def close(a, b, atol, rtol=1e-4, msg=""):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), atol=atol, rtol=rtol, err_msg=msg)


CONV_CASES = [  # B, Cin, F, T, Cout, nchw
    (2, 1, 40, 16, 8, True), (3, 1, 40, 18, 128, True), (2, 2, 40, 8, 128, True), (2, 4, 128, 8, 128, True),
    (2, 3, 40, 9, 16, True), (1, 4, 40, 6, 32, False),
    (2, 8, 40, 8, 8, False), (2, 16, 40, 10, 16, False),
    (2, 128, 40, 12, 128, False), (1, 128, 40, 7, 128, False), (2, 32, 40, 8, 32, False), (2, 64, 20, 8, 64, False),
    (1, 128, 128, 4, 128, False), (2, 128, 40, 8, 256, False), (2, 128, 128, 3, 128, False),
    # mel axis wider than one block tile / not a multiple of the tile / odd; time extents that leave ragged last tiles
    (1, 128, 128, 11, 128, False), (1, 128, 50, 5, 128, False), (1, 128, 41, 6, 128, False), (1, 128, 200, 2, 128, False),
    (1, 32, 90, 9, 32, False), (1, 64, 130, 3, 64, False), (2, 128, 7, 5, 128, False), (1, 128, 40, 1, 128, False),
    # narrow mel axis of the mel-pooled topologies: tall tiles (up to 35 time rows forward, 20 in the weight gradient)
    (2, 128, 8, 44, 128, False), (1, 128, 4, 75, 128, False), (2, 128, 2, 9, 128, False), (1, 128, 8, 3, 128, False),
    # channel counts off the MFMA path and too wide for one LDS image: the chunked fallback (several launches adding to y)
    (2, 100, 10, 6, 100, False), (1, 36, 12, 5, 132, False), (1, 200, 8, 4, 68, False), (2, 6, 40, 5, 200, True),
]


@pytest.mark.parametrize("B,Cin,Fm,T,Cout,nchw", CONV_CASES)
def test_conv3x3_fwd_stats_dgrad_wgrad(ops, B, Cin, Fm, T, Cout, nchw):
    gen = torch.Generator().manual_seed(B * 1000 + Cin + Cout)
    x = torch.randn(B, Cin, Fm, T, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / np.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=gen)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, b, padding=1)                       # [B,Cout,F,T]
    dy = torch.randn(y_ref.shape, generator=gen)
    y_ref.backward(dy)
    wf, wd = ops.conv3x3_pack(g(w))
    xin = g(x) if nchw else g(x.permute(0, 3, 2, 1))             # channels-last [B,T,F,Cin]
    y, stat = ops.conv3x3_fwd(xin, wf, g(b), nchw)
    y_ref_cl = y_ref.detach().permute(0, 3, 2, 1)                # [B,T,F,Cout]
    close(y, y_ref_cl, atol=2e-5, rtol=1e-4)
    s = stat.sum(0).cpu()
    np.testing.assert_allclose(s[0].numpy(), y_ref_cl.sum((0, 1, 2)).numpy(), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(s[1].numpy(), (y_ref_cl ** 2).sum((0, 1, 2)).numpy(), rtol=1e-4, atol=2e-3)
    # weight gradient
    dy_cl = g(dy.permute(0, 3, 2, 1))
    dw = ops.conv3x3_wgrad(xin, dy_cl, nchw)
    close(dw, wr.grad, atol=2e-4, rtol=2e-4)
    # data gradient = forward kernel on dy with the flipped/transposed pack
    if not nchw:             # the network input needs no gradient; layers >= 2 do
        dx, _ = ops.conv3x3_fwd(dy_cl, wd, None, False, want_stats=False)
        close(dx, xr.grad.permute(0, 3, 2, 1), atol=2e-5, rtol=1e-4)


POOL_CASES = [  # B,T,F,C,pf,pt,tcf
    (2, 8, 40, 8, 1, 2, False), (2, 8, 40, 128, 1, 2, False), (2, 8, 40, 128, 1, 2, True), (3, 4, 40, 16, 1, 2, True),
    (2, 4, 40, 32, 5, 1, False), (2, 4, 8, 32, 2, 1, True), (2, 8, 8, 16, 2, 2, False), (2, 6, 40, 8, 1, 1, True),
    # ragged extents: nn.MaxPool2d floors (sed.py:90), the dropped tail still carries the statistics terms of the gradient
    (2, 9, 40, 16, 1, 2, False), (3, 7, 13, 128, 1, 2, True), (2, 5, 42, 32, 5, 1, False), (2, 11, 9, 16, 2, 4, True),
    (1, 3, 7, 8, 3, 2, False),
]


def _bn_block_ref(y_nchw, gamma, beta, pf, pt, eps=1e-5):
    z = F.batch_norm(y_nchw, None, None, gamma, beta, training=True, eps=eps)
    return F.max_pool2d(torch.relu(z), (pf, pt))


@pytest.mark.parametrize("B,T,Fm,Cc,pf,pt,tcf", POOL_CASES)
def test_bn_relu_pool_fwd_bwd(ops, B, T, Fm, Cc, pf, pt, tcf):
    gen = torch.Generator().manual_seed(T * 100 + Cc + pf)
    y = torch.randn(B, Cc, Fm, T, generator=gen) * 1.5 + 0.3          # NCHW reference layout
    gamma = torch.rand(Cc, generator=gen) + 0.5
    beta = torch.randn(Cc, generator=gen) * 0.2
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    out_ref = _bn_block_ref(yr, gr, br, pf, pt)                       # [B,C,F',T']
    dout = torch.randn(out_ref.shape, generator=gen)
    out_ref.backward(dout)
    y_cl = g(y.permute(0, 3, 2, 1))                                   # [B,T,F,C]
    # statistics from partial sums, as the conv epilogue produces them
    part = torch.stack([y_cl.reshape(-1, Cc).sum(0), (y_cl.reshape(-1, Cc) ** 2).sum(0)]).reshape(1, 2, Cc).contiguous()
    rm, rv = torch.zeros(Cc).cuda(), torch.ones(Cc).cuda()
    mean, rstd, scale, shift = ops.bn_finalize_train(part, B * T * Fm, g(gamma), g(beta), rm, rv)
    close(mean, y.mean((0, 2, 3)), atol=1e-5)
    close(rstd, 1.0 / torch.sqrt(y.var((0, 2, 3), unbiased=False) + 1e-5), atol=1e-5, rtol=1e-4)
    close(rm, 0.1 * y.mean((0, 2, 3)), atol=1e-6)
    close(rv, 0.9 + 0.1 * y.var((0, 2, 3), unbiased=True), atol=1e-5)
    out = ops.bn_relu_pool_drop_fwd(y_cl, scale, shift, pf, pt, out_tcf=tcf)
    ref = out_ref.detach().permute(0, 3, 1, 2) if tcf else out_ref.detach().permute(0, 3, 2, 1)
    close(out, ref, atol=1e-5, rtol=1e-4)
    d_in = dout.permute(0, 3, 1, 2) if tcf else dout.permute(0, 3, 2, 1)
    dy, dgamma, dbeta, dbias = ops.bn_relu_pool_drop_bwd(y_cl, g(d_in), scale, shift, mean, rstd, pf, pt, out_tcf=tcf)
    close(dy, yr.grad.permute(0, 3, 2, 1), atol=2e-5, rtol=1e-3)
    close(dgamma, gr.grad, atol=1e-4, rtol=1e-3)
    close(dbeta, br.grad, atol=1e-4, rtol=1e-3)
    assert float(dbias.abs().max()) < 1e-3                            # sum of BN input grads is 0 up to rounding


@pytest.mark.parametrize("B,Cin,Fm,T,Cc,pf,pt", [(2, 1, 40, 16, 8, 1, 2), (3, 1, 40, 8, 128, 1, 2), (2, 2, 40, 8, 128, 1, 2),
                                                 (2, 2, 40, 8, 32, 5, 1), (2, 1, 8, 12, 16, 2, 2), (1, 1, 40, 4, 64, 1, 1)])
def test_conv1_fused_block_vs_torch(ops, B, Cin, Fm, T, Cc, pf, pt):
    _conv1_fused_case(ops, B, Cin, Fm, T, Cc, pf, pt, False)


@pytest.mark.parametrize("B,Cin,Fm,T,Cc,drop,shift", [(2, 1, 40, 16, 8, 0.0, 0.0), (3, 1, 40, 8, 128, 0.5, 0.0), (2, 2, 40, 8, 128, 0.3, 0.0),
                                                      (4, 1, 40, 64, 128, 0.5, 1.5), (2, 2, 24, 12, 32, 0.0, -2.0), (16, 1, 40, 256, 128, 0.5, 0.0),
                                                      (2, 4, 128, 16, 128, 0.5, 0.0), (3, 3, 40, 8, 32, 0.0, 1.0), (2, 4, 128, 64, 128, 0.5, -1.5)])
def test_conv1_backward_from_pooled_output_bits_and_moments_vs_float64(ops, B, Cin, Fm, T, Cc, drop, shift):
    """sed_conv1_bwd_wgrad: the first block's weight gradient assembled from R_k (the one sum over pooled elements), the input
    moments of the forward statistics pass and the BatchNorm-backward sums — no convolution, hash or BatchNorm arithmetic is
    redone — against a float64 autograd of conv -> BatchNorm(train) -> ReLU -> MaxPool(1,2) -> (the kernel's own dropout
    mask), and against the recomputing apply pass it replaces.  Includes un-standardised input (mean 1.5 / -2: the closed
    form subtracts moment terms), dropout on and off, a gamma == 0 / beta > 0 channel, and the config-1 size."""
    gen = torch.Generator().manual_seed(B * 7 + Cin + Cc + T)
    x = torch.randn(B, Cin, Fm, T, generator=gen) + shift
    w = torch.randn(Cc, Cin, 3, 3, generator=gen) / np.sqrt(9 * Cin)
    bias = torch.randn(Cc, generator=gen) * 0.3
    gamma = torch.rand(Cc, generator=gen) + 0.5
    beta = torch.randn(Cc, generator=gen) * 0.2
    gamma[1], beta[1] = 0.0, 0.4
    seed = 1234
    out = ops.conv1_fused_block(g(x), g(w), g(bias), g(gamma), g(beta), 1, 2, drop_p=drop, seed=seed, moments_path=True)     # [B,T/2,F,C]
    dout = torch.randn(out.shape, generator=gen) * 0.1
    res_m = ops.conv1_fused_block(g(x), g(w), g(bias), g(gamma), g(beta), 1, 2, dout=g(dout), drop_p=drop, seed=seed, moments_path=True)
    both = [("moments", res_m)]
    if Cin <= 2:                      # the recomputing passes exist for one and two input channels only
        res_a = ops.conv1_fused_block(g(x), g(w), g(bias), g(gamma), g(beta), 1, 2, dout=g(dout), drop_p=drop, seed=seed, moments_path=False)
        assert torch.equal(res_m[0], res_a[0])
        both.append(("recompute", res_a))
    # float64 reference with the kernel's own dropout mask (regenerated from the same seed through the stand-alone pass)
    mask = torch.ones(out.shape)
    if drop > 0:
        ones = torch.ones(B, T, Fm, Cc).cuda()
        mask = ops.bn_relu_pool_drop_fwd(ones, torch.ones(Cc).cuda(), torch.zeros(Cc).cuda(), 1, 2, drop_p=drop, seed=seed).cpu()
    xd = x.double()
    wd_ = w.double().requires_grad_(True)
    bd = bias.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    btd = beta.double().requires_grad_(True)
    y = F.conv2d(xd, wd_, bd, padding=1)
    z = F.batch_norm(y, None, None, gd, btd, training=True, eps=1e-5)
    o = F.max_pool2d(torch.relu(z), (1, 2)).permute(0, 3, 2, 1) * mask.double()          # channels-last [B,T/2,F,C]
    o.backward(dout.double())
    wmax = float(wd_.grad.abs().max())
    np.testing.assert_allclose(res_m[0].cpu().numpy(), o.detach().numpy(), atol=3e-5, rtol=1e-4)
    for name, res in both:
        _, dw, db, dgamma, dbeta = res
        np.testing.assert_allclose(dw.cpu().numpy(), wd_.grad.numpy(), atol=2e-5 * wmax + 1e-6, rtol=2e-4, err_msg=name)
        np.testing.assert_allclose(dgamma.cpu().numpy(), gd.grad.numpy(), atol=2e-5 * float(gd.grad.abs().max()) + 1e-6, rtol=2e-4, err_msg=name)
        np.testing.assert_allclose(dbeta.cpu().numpy(), btd.grad.numpy(), atol=2e-5 * float(btd.grad.abs().max()) + 1e-6, rtol=2e-4, err_msg=name)
        assert float(db.abs().max()) < 1e-4 * max(1.0, wmax)                         # analytically zero in front of BatchNorm


def _conv1_fused_case(ops, B, Cin, Fm, T, Cc, pf, pt, moments):
    """first block with the conv recomputed in every pass: same results as conv -> BN(train) -> ReLU -> pool in torch"""
    gen = torch.Generator().manual_seed(B + Cin * 10 + Cc)
    x = torch.randn(B, Cin, Fm, T, generator=gen)
    w = (torch.randn(Cc, Cin, 3, 3, generator=gen) / np.sqrt(9 * Cin)).requires_grad_(True)
    bias = torch.randn(Cc, generator=gen).requires_grad_(True)
    gamma = (torch.rand(Cc, generator=gen) + 0.5).requires_grad_(True)
    beta = (torch.randn(Cc, generator=gen) * 0.2).requires_grad_(True)
    ref = _bn_block_ref(F.conv2d(x, w, bias, padding=1), gamma, beta, pf, pt)       # [B,C,F',T']
    dout = torch.randn(ref.shape, generator=gen)
    ref.backward(dout)
    out, dw, db, dgamma, dbeta = ops.conv1_fused_block(g(x), g(w.detach()), g(bias.detach()), g(gamma.detach()),
                                                       g(beta.detach()), pf, pt, dout=g(dout.permute(0, 3, 2, 1)))
    close(out, ref.detach().permute(0, 3, 2, 1), atol=2e-5, rtol=1e-4)
    close(dw, w.grad, atol=2e-4, rtol=1e-3)
    close(dgamma, gamma.grad, atol=2e-4, rtol=1e-3)
    close(dbeta, beta.grad, atol=2e-4, rtol=1e-3)
    assert float(db.abs().max()) < 1e-3


@pytest.mark.parametrize("cin,mu,sd", [(1, 0.0, 1.0), (1, -3.0, 3.0), (2, 3.0, 3.0), (1, 30.0, 1.0)])
def test_conv1_stats_from_input_moments_vs_float64_sums(ops, cin, mu, sd):
    """round-2 advisor: the first block's batch statistics come from the Gram matrix of the shifted inputs (fp32 per
    workgroup, fp64 across) and a fp64 quadratic form, i.e. E[y^2] - mean^2 — which cancels when the input is NOT standardised
    (raw log-mel energies: mean and sigma of a few units).  Against float64 sums of the float64 convolution, at config-1 size:
    the mean within 1e-6 * sqrt(E[y^2]) and the variance within 1e-5 * E[y^2] — relative to the SECOND MOMENT, which is what
    any fp32 sum-of-squares statistic (the conv epilogue's partials included) can promise — and, for inputs up to mean = sigma
    = 3, the resulting 1/sigma within 1e-4 relative, far inside the 1e-3 budget of the probabilities."""
    from sed_crnn_amd._lib import lib, check, ptr, stream_ptr
    B, Fm, T, Cc = 16, 40, 256, 128
    gen = torch.Generator().manual_seed(int(abs(mu) * 10 + sd + cin))
    x = torch.randn(B, cin, Fm, T, generator=gen) * sd + mu
    w = torch.randn(Cc, cin, 3, 3, generator=gen) / np.sqrt(9 * cin)
    b = torch.randn(Cc, generator=gen)
    y64 = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    n = B * Fm * T
    m64 = y64.mean((0, 2, 3))
    q64 = (y64 ** 2).mean((0, 2, 3))
    v64 = y64.var((0, 2, 3), unbiased=False)
    L = lib()
    assert L.sed_conv1_fused_supported(cin, Fm, T, Cc, 1, 2)
    wf, _ = ops.conv3x3_pack(g(w))
    stat = torch.empty(1, 2, Cc, device="cuda")
    ws = torch.empty(L.sed_conv1_stats_workspace_bytes(B, cin, T) // 4 + 1, device="cuda")
    xg = g(x)
    check(L.sed_conv1_stats(ptr(xg), ptr(wf), ptr(g(b)), ptr(stat), ptr(ws), B, cin, Fm, T, Cc, None, stream_ptr()), "conv1_stats")
    rm, rv = torch.zeros(Cc).cuda(), torch.ones(Cc).cuda()
    mean, rstd, _, _ = ops.bn_finalize_train(stat, n, torch.ones(Cc).cuda(), torch.zeros(Cc).cuda(), rm, rv)
    mean, rstd = mean.cpu().double(), rstd.cpu().double()
    var = 1.0 / rstd ** 2 - 1e-5
    e_m = float(((mean - m64).abs() / q64.sqrt()).max())
    e_v = float(((var - v64).abs() / q64).max())
    e_r = float((rstd * (v64 + 1e-5).sqrt() - 1.0).abs().max())
    print(f"conv1_stats cin={cin} input N({mu},{sd}): mean err / rms {e_m:.1e}, var err / E[y^2] {e_v:.1e}, rstd rel err {e_r:.1e}, "
          f"worst E[y^2]/var {float((q64 / v64).max()):.1f}")
    assert e_m <= 1e-6 and e_v <= 1e-5
    if abs(mu) <= 3.0:
        assert e_r <= 1e-4


@pytest.mark.parametrize("B,Ty,Fy,pf,pt,p", [(3, 16, 40, 1, 2, 0.5), (2, 12, 40, 1, 2, 0.0), (2, 9, 40, 1, 2, 0.5), (1, 8, 41, 2, 2, 0.3),
                                              (2, 6, 128, 1, 2, 0.5), (1, 20, 8, 2, 1, 0.25)])
def test_dgrad_with_fused_bn_backward_reduction_matches_the_two_pass_form(ops, B, Ty, Fy, pf, pt, p):
    """sed_conv3x3_dgrad_bnred: the data gradient of a 128-channel block whose epilogue also forms (sum g, sum g*xhat) of the
    block below from that block's pooled OUTPUT, against the separate passes (sed_conv3x3_fwd_ex on the dgrad weights +
    sed_bn_relu_pool_drop_bwd_reduce, which re-reads the conv output, finds the arg-max and regenerates the dropout mask):
    dx bit for bit (same main loop), the sums to rounding.  Ragged pooling tails, mel pooling, dropout on / off, edge tiles,
    and channels with gamma == 0 (beta > 0: xhat from the conv output at the window's first element; beta <= 0: nothing
    passes) are all in."""
    C = 128
    gen = torch.Generator().manual_seed(B * 100 + Ty + Fy)
    yb = torch.randn(B, Ty, Fy, C, generator=gen).cuda()                        # conv output of the block below (channels-last)
    gamma = (torch.rand(C, generator=gen) + 0.5)
    beta = torch.randn(C, generator=gen) * 0.3
    gamma[3], beta[3] = 0.0, 0.5
    gamma[77], beta[77] = 0.0, -0.2
    gamma[100] = -0.7                                                            # a negative scale flips the arg-max
    # |gamma| << |beta| (round-3 advisor): xhat = (z - beta)/gamma would lose eps |beta/gamma| — these channels must take xhat
    # from the conv output (window searched like the forward does); their sums are held to the SAME tolerance as the others
    gamma[8], beta[8] = 1e-3, 0.5
    gamma[9], beta[9] = -1e-5, 0.4
    gamma[10], beta[10] = 1e-5, 0.6
    gamma[11], beta[11] = 0.02, 0.9                                              # just above the 1/64 switch: the fast path, error bound 4e-6
    gamma, beta = gamma.cuda(), beta.cuda()
    n = B * Ty * Fy
    flat = yb.reshape(-1, C)
    part = torch.stack([flat.sum(0), (flat ** 2).sum(0)]).reshape(1, 2, C).contiguous()
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    mean, rstd, scale, shift = ops.bn_finalize_train(part, n, gamma, beta, rm, rv)
    seed = 99
    pooled = ops.bn_relu_pool_drop_fwd(yb, scale, shift, pf, pt, drop_p=p, seed=seed)        # [B,T,F,C]
    T, Fm = Ty // pt, Fy // pf
    dy = (torch.randn(B, T, Fm, C, generator=gen) * 0.1).cuda()
    w = (torch.randn(C, C, 3, 3, generator=gen) / np.sqrt(9 * C)).cuda()
    _, wd = ops.conv3x3_pack(w)
    dx_ref, _ = ops.conv3x3_fwd(dy, wd, None, False, want_stats=False)
    _, dgamma_ref, dbeta_ref, _ = ops.bn_relu_pool_drop_bwd(yb, dx_ref, scale, shift, mean, rstd, pf, pt, drop_p=p, seed=seed)
    dx, sum_g, sum_gx = ops.conv3x3_dgrad_bnred(dy, wd, pooled, gamma, beta, yb, mean, rstd, pf, pt, drop_p=p, scale=scale, shift=shift)
    assert torch.equal(dx, dx_ref)
    keep_scale = 1.0 / (1.0 - p)
    mag = float((dx_ref.abs() * keep_scale).sum(dim=(0, 1, 2)).max())        # an upper bound of sum |g| per channel
    close(sum_g, dbeta_ref, atol=2e-6 * mag, rtol=1e-4)
    close(sum_gx, dgamma_ref, atol=6e-6 * mag, rtol=1e-4)
    assert float(sum_g[77].abs()) == 0.0 and float(sum_gx[77].abs()) == 0.0      # gamma 0, beta < 0: the gate is shut everywhere
    assert float(sum_gx[3].abs()) > 0.0
    for c in (8, 9, 10, 11):                                                     # the small-gamma channels, individually
        assert abs(float(sum_gx[c] - dgamma_ref[c])) <= 6e-6 * mag + 1e-4 * abs(float(dgamma_ref[c])), (c, float(sum_gx[c]), float(dgamma_ref[c]))
        assert float(sum_gx[c].abs()) > 0.0


@pytest.mark.parametrize("B,Fm,T,Cout", [(2, 40, 16, 128), (1, 40, 128, 128), (3, 40, 6, 128), (2, 20, 10, 64), (1, 10, 64, 128),
                                          (2, 8, 2, 256), (1, 2, 2, 128), (9, 32, 12, 128), (8, 40, 64, 128),
                                          (2, 128, 8, 128), (8, 128, 6, 128), (1, 64, 10, 128), (1, 256, 4, 128)])
def test_conv3x3_winograd_forward_and_data_gradient(ops, B, Fm, T, Cout):
    """The Winograd F(2x2,3x3) form of the 128-channel convolutions (wino.hip) against torch in float64, next to the direct
    exact-fp32 kernel on the same inputs: forward with bias and the statistic partial rows, and the data gradient (same kernel,
    flipped / transposed transformed weights).  Mel widths whose tile rows do not divide the 64-tile blocks (40, 20, 10), blocks
    that end in the middle of a sequence's last tile row, a single tile, batch sizes with and without the XCD-aware order, and
    the wide mel axes (128 bins of config 5, 256) whose tile rows are cut into column groups with a halo between workgroups.  The
    transforms only add and halve, so the error stays within a small multiple of the direct kernel's (asserted: 4x + an ulp term)."""
    torch.manual_seed(B * 7 + Fm + T)
    Cin = 128
    x = torch.randn(B, T, Fm, Cin)
    w = torch.randn(Cout, Cin, 3, 3) / (3.0 * Cin ** 0.5)
    bias = torch.randn(Cout)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 2, 1).double(), w.double(), bias.double(), padding=1).permute(0, 3, 2, 1).contiguous()
    uf, _ = ops.conv3x3_wino_pack(w.cuda())
    y, stat = ops.conv3x3_wino_fwd(x.cuda(), uf, bias.cuda(), Cout)
    wf0, _ = ops.conv3x3_pack(w.cuda())
    y0, _ = ops.conv3x3_fwd(x.cuda(), wf0, bias.cuda(), False)
    scale = float(ref.abs().mean())
    err, err0 = (y.cpu().double() - ref).abs(), (y0.cpu().double() - ref).abs()
    print(f"winograd fwd B={B} F={Fm} T={T}: max err {float(err.max()):.2e} mean {float(err.mean()):.2e} | direct max {float(err0.max()):.2e} "
          f"mean {float(err0.mean()):.2e} | scale {scale:.2e}")
    assert float(err.max()) < 4.0 * float(err0.max()) + 2e-6 * scale and float(err.mean()) < 4.0 * float(err0.mean()) + 2e-7 * scale
    assert float(err.max()) < 2e-5 * scale * 10                      # an absolute bound too: ~1e-5 of the output magnitude
    s = stat.sum(0).cpu().double()
    torch.testing.assert_close(s[0], ref.sum((0, 1, 2)), rtol=1e-4, atol=2e-5 * scale * B * T * Fm)
    torch.testing.assert_close(s[1], (ref * ref).sum((0, 1, 2)), rtol=1e-4, atol=1e-4)
    assert torch.equal(y, ops.conv3x3_wino_fwd(x.cuda(), uf, bias.cuda(), Cout)[0])          # fixed order: run to run identical
    if Cout == Cin:
        dy = torch.randn(B, T, Fm, Cout)
        dref = torch.nn.grad.conv2d_input((B, Cin, Fm, T), w.double(), dy.permute(0, 3, 2, 1).contiguous().double(), padding=1).permute(0, 3, 2, 1)
        _, ud = ops.conv3x3_wino_pack(w.cuda())
        dx, _ = ops.conv3x3_wino_fwd(dy.cuda(), ud, None, Cin, want_stats=False)
        _, wd0 = ops.conv3x3_pack(w.cuda())
        dx0, _ = ops.conv3x3_fwd(dy.cuda(), wd0, None, False, want_stats=False)
        derr, derr0 = (dx.cpu().double() - dref).abs(), (dx0.cpu().double() - dref).abs()
        assert float(derr.max()) < 4.0 * float(derr0.max()) + 2e-6 * float(dref.abs().mean()), (float(derr.max()), float(derr0.max()))


def _wino_sweep_cases(n=28, seed=20260405):
    import random
    r = random.Random(seed)
    out = []
    while len(out) < n:
        B = r.choice([1, 2, 3, 5, 8, 9, 16])
        Fm = r.choice([2, 4, 6, 10, 12, 14, 20, 22, 26, 34, 40, 48, 64, 70, 128])
        T = r.choice([2, 4, 6, 8, 10, 14, 18, 26, 32, 50])
        Cout = r.choice([64, 128, 128, 192])
        if B * Fm * T > 60000:
            continue
        out.append((B, Fm, T, Cout))
    return out


@pytest.mark.parametrize("B,Fm,T,Cout", _wino_sweep_cases())
def test_conv3x3_winograd_seeded_shape_sweep(ops, B, Fm, T, Cout):
    """28 seeded shapes through sed_conv3x3_wino_fwd (every shape the geometry takes must be right; the rest must say so): tile
    rows of 1 to 64 tiles, blocks that straddle tile rows at every phase, column groups (mel 128 and 70: 35 tiles in one group),
    a ragged last block, batch sizes on and off the XCD-aware order, 64 / 128 / 192 output channels — against torch in float64
    with the bound of the fixed-shape test (the direct kernel's own error on the same inputs x 4 + an ulp term)."""
    from sed_crnn_amd._lib import lib
    rows = lib().sed_conv3x3_wino_rows(B, 128, Fm, T, Cout)
    assert rows > 0, "every even (F, T) with 128 input channels and 64-multiples of output channels is taken"
    gen = torch.Generator().manual_seed(B * 131 + Fm * 17 + T)
    x = torch.randn(B, T, Fm, 128, generator=gen)
    w = torch.randn(Cout, 128, 3, 3, generator=gen) / 34.0
    bias = torch.randn(Cout, generator=gen)
    ref = F.conv2d(x.permute(0, 3, 2, 1).double(), w.double(), bias.double(), padding=1).permute(0, 3, 2, 1).contiguous()
    uf, _ = ops.conv3x3_wino_pack(g(w))
    y, stat = ops.conv3x3_wino_fwd(g(x), uf, g(bias), Cout)
    assert stat.shape[0] == rows
    wf0, _ = ops.conv3x3_pack(g(w))
    y0, _ = ops.conv3x3_fwd(g(x), wf0, g(bias), False)
    scale = float(ref.abs().mean())
    err, err0 = (y.cpu().double() - ref).abs(), (y0.cpu().double() - ref).abs()
    assert float(err.max()) < 4.0 * float(err0.max()) + 2e-6 * scale, (float(err.max()), float(err0.max()), scale)
    s = stat.sum(0).cpu().double()
    torch.testing.assert_close(s[0], ref.sum((0, 1, 2)), rtol=1e-4, atol=2e-5 * scale * B * T * Fm)
    torch.testing.assert_close(s[1], (ref * ref).sum((0, 1, 2)), rtol=1e-4, atol=1e-4 * B * T * Fm)


@pytest.mark.parametrize("B,Ty,Fy,pf,pt,p", [(3, 16, 40, 1, 2, 0.5), (2, 12, 40, 1, 2, 0.0), (2, 8, 64, 1, 2, 0.5), (1, 8, 40, 2, 1, 0.25)])
def test_winograd_dgrad_with_fused_bn_backward_reduction(ops, B, Ty, Fy, pf, pt, p):
    """sed_conv3x3_wino_dgrad_bnred against sed_conv3x3_dgrad_bnred (the direct kernel with the same epilogue): dx to the
    Winograd rounding, (sum g, sum g*xhat) of the block below to the tolerance the direct form is held to against the two-pass
    reference — the gate `pooled > 0` comes from the forward's tensor, so the selection is identical."""
    C = 128
    gen = torch.Generator().manual_seed(B * 100 + Ty + Fy)
    yb = torch.randn(B, Ty, Fy, C, generator=gen).cuda()
    gamma = (torch.rand(C, generator=gen) + 0.5)
    beta = torch.randn(C, generator=gen) * 0.3
    gamma[3], beta[3] = 0.0, 0.5
    gamma[100] = -0.7
    gamma[8], beta[8] = 1e-3, 0.5
    gamma, beta = gamma.cuda(), beta.cuda()
    flat = yb.reshape(-1, C)
    part = torch.stack([flat.sum(0), (flat ** 2).sum(0)]).reshape(1, 2, C).contiguous()
    mean, rstd, scale, shift = ops.bn_finalize_train(part, B * Ty * Fy, gamma, beta, torch.zeros(C).cuda(), torch.ones(C).cuda())
    pooled = ops.bn_relu_pool_drop_fwd(yb, scale, shift, pf, pt, drop_p=p, seed=99)
    T, Fm = Ty // pt, Fy // pf
    dy = (torch.randn(B, T, Fm, C, generator=gen) * 0.1).cuda()
    w = (torch.randn(C, C, 3, 3, generator=gen) / np.sqrt(9 * C)).cuda()
    _, wd = ops.conv3x3_pack(w)
    _, ud = ops.conv3x3_wino_pack(w)
    dx0, sg0, sgx0 = ops.conv3x3_dgrad_bnred(dy, wd, pooled, gamma, beta, yb, mean, rstd, pf, pt, drop_p=p, scale=scale, shift=shift)
    dx, sg, sgx = ops.conv3x3_dgrad_bnred(dy, ud, pooled, gamma, beta, yb, mean, rstd, pf, pt, drop_p=p, scale=scale, shift=shift, wino=True)
    close(dx, dx0, atol=2e-6 * float(dx0.abs().mean()) * 10, rtol=1e-5)
    mag = float((dx0.abs() / (1.0 - p)).sum(dim=(0, 1, 2)).max())
    close(sg, sg0, atol=2e-6 * mag, rtol=1e-4)
    close(sgx, sgx0, atol=6e-6 * mag, rtol=1e-4)


@pytest.mark.parametrize("B,Ty,Fy,C,pf,pt,p", [(3, 16, 40, 128, 1, 2, 0.5), (2, 12, 40, 128, 1, 2, 0.0), (2, 9, 40, 128, 1, 2, 0.5),
                                                (1, 8, 128, 128, 1, 2, 0.5), (2, 6, 48, 64, 3, 1, 0.25), (5, 4, 16, 32, 2, 2, 0.3)])
def test_bn_backward_sums_of_the_gru_feeding_block_from_its_pooled_output(ops, B, Ty, Fy, C, pf, pt, p):
    """sed_bn_bwd_reduce_pooled: (sum g, sum g*xhat) of the block that feeds the GRU, formed from its pooled output
    [B,Tp,C,Fp] and that tensor's gradient, against the pass that re-reads the conv output, finds the arg-max and regenerates
    the dropout mask (sed_bn_relu_pool_drop_bwd_reduce): to rounding.  gamma == 0 channels of both beta signs, a negative
    gamma, ragged time tails, mel pooling and config 5's 128 mel bins are in."""
    gen = torch.Generator().manual_seed(B * 100 + Ty + Fy + C)
    yb = torch.randn(B, Ty, Fy, C, generator=gen).cuda()
    gamma = (torch.rand(C, generator=gen) + 0.5)
    beta = torch.randn(C, generator=gen) * 0.3
    gamma[3], beta[3] = 0.0, 0.5
    gamma[17], beta[17] = 0.0, -0.2
    gamma[20] = -0.7
    gamma[8], beta[8] = 1e-3, 0.5                                                # |gamma| << |beta|: xhat from the conv output (see above)
    gamma[9], beta[9] = -1e-5, 0.4
    gamma[10], beta[10] = 1e-5, 0.6
    gamma, beta = gamma.cuda(), beta.cuda()
    flat = yb.reshape(-1, C)
    part = torch.stack([flat.sum(0), (flat ** 2).sum(0)]).reshape(1, 2, C).contiguous()
    mean, rstd, scale, shift = ops.bn_finalize_train(part, B * Ty * Fy, gamma, beta, torch.zeros(C).cuda(), torch.ones(C).cuda())
    seed = 7
    pooled = ops.bn_relu_pool_drop_fwd(yb, scale, shift, pf, pt, out_tcf=True, drop_p=p, seed=seed)       # [B,Tp,C,Fp]
    dout = (torch.randn(pooled.shape, generator=gen) * 0.1).cuda()
    _, dgamma_ref, dbeta_ref, _ = ops.bn_relu_pool_drop_bwd(yb, dout, scale, shift, mean, rstd, pf, pt, out_tcf=True, drop_p=p, seed=seed)
    sum_g, sum_gx = ops.bn_bwd_sums_from_pooled(pooled, dout, gamma, beta, yb, mean, rstd, pf, pt, drop_p=p, scale=scale, shift=shift)
    mag = float((dout.abs() / (1.0 - p)).sum(dim=(0, 1, 3)).max())
    close(sum_g, dbeta_ref, atol=2e-6 * mag, rtol=1e-4)
    close(sum_gx, dgamma_ref, atol=6e-6 * mag, rtol=1e-4)
    assert float(sum_g[17].abs()) == 0.0 and float(sum_gx[17].abs()) == 0.0
    assert float(sum_gx[3].abs()) > 0.0
    for c in (8, 9, 10):
        assert abs(float(sum_gx[c] - dgamma_ref[c])) <= 6e-6 * mag + 1e-4 * abs(float(dgamma_ref[c])), (c, float(sum_gx[c]), float(dgamma_ref[c]))


@pytest.mark.parametrize("B,F,T,p,cin", [(3, 40, 32, 0.5, 1), (2, 40, 20, 0.0, 1), (1, 40, 36, 0.5, 1), (2, 24, 12, 0.3, 1), (2, 128, 16, 0.5, 1),
                                          (3, 40, 32, 0.5, 2), (1, 40, 36, 0.0, 2), (2, 24, 12, 0.3, 2)])
def test_first_block_weight_gradient_sums_from_the_data_gradient_epilogue(ops, B, F, T, p, cin):
    """sed_conv3x3_dgrad_bnred_rg + sed_conv1_bwd_wgrad_assemble against sed_conv3x3_dgrad_bnred + sed_conv1_bwd_wgrad: the
    data gradient of block 2 whose epilogue also forms the 1- or 2-channel first block's weight-gradient sums (sum g, R_k) from
    the network input and the arg-max bits, instead of conv1_rgrad_k's pass over dx / pooled / bits.  dx and the BatchNorm
    partials bit for bit (same instructions), dW / dbias of the first block to rounding.  Edge tiles in time and mel (zero
    padding of the input patch), ragged tile counts, dropout on / off, gamma == 0 of both beta signs, negative gamma."""
    from sed_crnn_amd._lib import lib, ptr, check, stream_ptr
    L = lib()
    C = 128
    gen = torch.Generator().manual_seed(B * 1000 + F * 10 + T + cin)
    g = lambda t: t.cuda()
    x = torch.randn(B, cin, F, T, generator=gen)
    w1 = torch.randn(C, cin, 3, 3, generator=gen) * 0.4
    b1 = torch.randn(C, generator=gen) * 0.1
    gamma = torch.rand(C, generator=gen) + 0.5
    beta = torch.randn(C, generator=gen) * 0.3
    gamma[5], beta[5] = 0.0, 0.4
    gamma[9], beta[9] = 0.0, -0.3
    gamma[64] = -0.8
    w2 = torch.randn(C, C, 3, 3, generator=gen) / np.sqrt(9 * C)
    Tp = T // 2
    if not L.sed_conv3x3_dgrad_bnred_rg_rows(B, C, F, Tp, C, cin):
        pytest.skip("shape does not take the fused path")
    dy = (torch.randn(B, Tp, F, C, generator=gen) * 0.1)
    seed = 11
    # forward of the first block (statistics from the input moments, arg-max bits)
    wf1, _ = ops.conv3x3_pack(g(w1))
    stat = torch.empty(1, 2, C).cuda()
    sws = torch.empty(L.sed_conv1_stats_workspace_bytes(B, cin, T) // 4 + 1).cuda()
    mom = torch.empty(L.sed_conv1_moments_doubles(cin), dtype=torch.float64).cuda()
    xg, b1g, gg, bg = g(x), g(b1), g(gamma), g(beta)
    check(L.sed_conv1_stats(ptr(xg), ptr(wf1), ptr(b1g), ptr(stat), ptr(sws), B, cin, F, T, C, ptr(mom), stream_ptr()), "conv1_stats")
    mean, rstd, scale, shift = ops.bn_finalize_train(stat, B * T * F, gg, bg, torch.zeros(C).cuda(), torch.ones(C).cuda())
    pooled = torch.empty(B, Tp, F, C).cuda()
    bits = torch.empty(pooled.numel() // 4, dtype=torch.uint8).cuda()
    check(L.sed_conv1_bn_relu_pool_drop_fwd(ptr(xg), ptr(wf1), ptr(b1g), ptr(scale), ptr(shift), ptr(pooled), B, cin, F, T, C, 1, 2, p, seed,
                                            None, ptr(bits), stream_ptr()), "conv1_fwd")
    _, wd2 = ops.conv3x3_pack(g(w2))
    dyg = g(dy)

    def finish(part, rows):
        sum_g, sum_gx, dgamma, dbeta = (torch.empty(C).cuda() for _ in range(4))
        check(L.sed_bn_bwd_finalize(ptr(part), rows, C, ptr(sum_g), ptr(sum_gx), ptr(dgamma), ptr(dbeta), stream_ptr()), "fin")
        return sum_g, sum_gx, dgamma, dbeta

    # reference: plain fused data gradient, then the stand-alone pass over dx / pooled / bits
    rows = L.sed_conv3x3_dgrad_bnred_rows(B, C, F, Tp, C)
    dx_a, part_a = torch.empty(B, Tp, F, C).cuda(), torch.empty(rows, 2, C).cuda()
    check(L.sed_conv3x3_dgrad_bnred(ptr(dyg), ptr(wd2), ptr(dx_a), ptr(part_a), ptr(pooled), ptr(gg), ptr(bg), None, ptr(mean), ptr(rstd),
                                    p, 1, 2, F, T, B, C, F, Tp, C, stream_ptr()), "dgrad_bnred")
    sg_a, sgx_a, dgam_a, _ = finish(part_a, rows)
    dw_a, db_a = torch.empty(C, cin, 3, 3).cuda(), torch.empty(C).cuda()
    ws = torch.empty(L.sed_conv1_bwd_wgrad_workspace_bytes(B, cin, T, C) // 4 + 1).cuda()
    check(L.sed_conv1_bwd_wgrad(ptr(xg), ptr(dx_a), ptr(pooled), ptr(bits), ptr(mom), ptr(wf1), ptr(b1g), ptr(mean), ptr(rstd), ptr(scale),
                                ptr(sg_a), ptr(sgx_a), ptr(dw_a), ptr(db_a), ptr(ws), B, cin, F, T, C, p, ptr(gg), ptr(bg), ptr(dgam_a),
                                stream_ptr()), "conv1_bwd_wgrad")
    # fused: the sums come out of the data gradient's epilogue
    rows_b = L.sed_conv3x3_dgrad_bnred_rg_rows(B, C, F, Tp, C, cin)
    assert rows_b == rows
    dx_b, part_b = torch.empty(B, Tp, F, C).cuda(), torch.empty(rows, 2, C).cuda()
    rgp = torch.full((rows, C, 1 + 9 * cin), float("nan")).cuda()
    check(L.sed_conv3x3_dgrad_bnred_rg(ptr(dyg), ptr(wd2), ptr(dx_b), ptr(part_b), ptr(pooled), ptr(gg), ptr(bg), ptr(mean), ptr(rstd), p,
                                       ptr(xg), cin, ptr(bits), ptr(rgp), B, C, F, Tp, C, stream_ptr()), "dgrad_bnred_rg")
    sg_b, sgx_b, dgam_b, _ = finish(part_b, rows)
    dw_b, db_b = torch.empty(C, cin, 3, 3).cuda(), torch.empty(C).cuda()
    check(L.sed_conv1_bwd_wgrad_assemble(ptr(rgp), rows, ptr(mom), ptr(wf1), ptr(b1g), ptr(mean), ptr(rstd), ptr(scale), ptr(sg_b), ptr(sgx_b),
                                         ptr(dw_b), ptr(db_b), B, cin, F, T, C, ptr(gg), ptr(bg), ptr(dgam_b), stream_ptr()), "assemble")
    assert torch.equal(dx_a, dx_b) and torch.equal(part_a, part_b)
    assert bool(torch.isfinite(rgp).all())
    mag = float(dw_a.abs().max())
    close(dw_b, dw_a, atol=2e-5 * mag, rtol=1e-4)
    # the conv-bias gradient in front of a BatchNorm is zero up to rounding (sum g - count * mean g): noise of the size of ulp(sum g)
    close(db_b, db_a, atol=4e-6 * float((sg_a.abs() * scale.abs()).max()) + 1e-6, rtol=1e-4)
    close(dgam_b, dgam_a, atol=2e-5 * float(dgam_a.abs().max()) + 1e-6, rtol=1e-4)
    # the same epilogue behind the Winograd main loop (the plan's default where the shape takes it): dx to the Winograd rounding,
    # the first block's dW / dbias / dgamma to the tolerance above
    rows_w = L.sed_conv3x3_wino_rg_rows(B, C, F, Tp, C, cin)
    if rows_w and Tp % 2 == 0 and F % 2 == 0:
        _, ud2 = ops.conv3x3_wino_pack(g(w2))
        dx_w, part_w = torch.empty(B, Tp, F, C).cuda(), torch.empty(rows_w, 2, C).cuda()
        rgw = torch.full((rows_w, C, 1 + 9 * cin), float("nan")).cuda()
        check(L.sed_conv3x3_wino_dgrad_bnred_rg(ptr(dyg), ptr(ud2), ptr(dx_w), ptr(part_w), ptr(pooled), ptr(gg), ptr(bg), ptr(mean), ptr(rstd), p,
                                                ptr(xg), cin, ptr(bits), ptr(rgw), B, C, F, Tp, C, stream_ptr()), "wino_dgrad_bnred_rg")
        sum_g, sum_gx, dgam_w, _ = (torch.empty(C).cuda() for _ in range(4))
        check(L.sed_bn_bwd_finalize(ptr(part_w), rows_w, C, ptr(sum_g), ptr(sum_gx), ptr(dgam_w), ptr(_), stream_ptr()), "fin")
        dw_w, db_w = torch.empty(C, cin, 3, 3).cuda(), torch.empty(C).cuda()
        check(L.sed_conv1_bwd_wgrad_assemble(ptr(rgw), rows_w, ptr(mom), ptr(wf1), ptr(b1g), ptr(mean), ptr(rstd), ptr(scale), ptr(sum_g), ptr(sum_gx),
                                             ptr(dw_w), ptr(db_w), B, cin, F, T, C, ptr(gg), ptr(bg), ptr(dgam_w), stream_ptr()), "assemble")
        assert bool(torch.isfinite(rgw).all())
        close(dx_w, dx_a, atol=2e-5 * float(dx_a.abs().mean()), rtol=1e-5)
        close(dw_w, dw_a, atol=2e-5 * mag, rtol=1e-4)
        close(db_w, db_a, atol=4e-6 * float((sg_a.abs() * scale.abs()).max()) + 1e-6, rtol=1e-4)
        close(dgam_w, dgam_a, atol=2e-5 * float(dgam_a.abs().max()) + 1e-6, rtol=1e-4)


def test_bn_eval_scale_shift(ops):
    Cc = 16
    gen = torch.Generator().manual_seed(5)
    gamma, beta, rm = (torch.randn(Cc, generator=gen) for _ in range(3))
    rv = torch.rand(Cc, generator=gen) + 0.5
    scale, shift = ops.bn_finalize_eval(g(gamma), g(beta), g(rm), g(rv))
    y = torch.randn(2, Cc, 8, 4, generator=gen)
    ref = F.max_pool2d(torch.relu(F.batch_norm(y, rm, rv, gamma, beta, training=False)), (1, 2))
    out = ops.bn_relu_pool_drop_fwd(g(y.permute(0, 3, 2, 1)), scale, shift, 1, 2)
    close(out, ref.permute(0, 3, 2, 1), atol=1e-5)


@pytest.mark.parametrize("tcf", [False, True])
def test_dropout_mask_statistics_and_backward_consistency(ops, tcf):
    B, T, Fm, Cc, p = 4, 16, 40, 128, 0.5
    y = torch.rand(B, T, Fm, Cc).cuda() + 0.5                          # strictly positive: relu/pool transparent
    scale, shift = torch.ones(Cc).cuda(), torch.zeros(Cc).cuda()
    mean, rstd = torch.zeros(Cc).cuda(), torch.ones(Cc).cuda()
    out = ops.bn_relu_pool_drop_fwd(y, scale, shift, 1, 2, out_tcf=tcf, drop_p=p, seed=1234)
    base = ops.bn_relu_pool_drop_fwd(y, scale, shift, 1, 2, out_tcf=tcf, drop_p=0.0)
    kept = out != 0
    frac = kept.float().mean().item()
    assert abs(frac - (1 - p)) < 0.01, frac
    close(out[kept], (base * 2.0)[kept], atol=1e-6)                    # inverted-dropout scaling 1/(1-p)
    out2 = ops.bn_relu_pool_drop_fwd(y, scale, shift, 1, 2, out_tcf=tcf, drop_p=p, seed=1234)
    assert torch.equal(out, out2)                                      # same seed -> same mask
    out3 = ops.bn_relu_pool_drop_fwd(y, scale, shift, 1, 2, out_tcf=tcf, drop_p=p, seed=1235)
    assert not torch.equal(out, out3)
    # backward regenerates the same mask: with sum_g/sum_gx forced to 0 the routed gradient is dout*mask*2
    dout = torch.ones_like(out)
    from sed_crnn_amd._lib import lib, ptr, stream_ptr, check
    dy = torch.empty_like(y)
    rows = lib().sed_bn_bwd_rows(B, T, 2)
    dbp = torch.empty(rows, Cc).cuda()
    z = torch.zeros(Cc).cuda()
    check(lib().sed_bn_relu_pool_drop_bwd_apply(ptr(y), ptr(dout), ptr(scale), ptr(shift), ptr(mean), ptr(rstd), ptr(z), ptr(z),
                                                ptr(dy), ptr(dbp), B, T, Fm, Cc, 1, 2, int(tcf), p, 1234, None, stream_ptr()))
    routed = dy.reshape(B, T // 2, 2, Fm, Cc).sum(2)                   # one of each time pair carries the gradient
    mask = kept.permute(0, 1, 3, 2) if tcf else kept
    close(routed, mask.float() * 2.0, atol=1e-6)


@pytest.mark.parametrize("pf,pt,Fm,T", [(1, 2, 40, 16), (5, 1, 40, 8), (2, 2, 13, 9), (3, 2, 41, 10), (1, 4, 8, 12), (1, 1, 6, 5)])
def test_routing_codes_are_the_decisions_the_backward_kernels_act_on(ops, pf, pt, Fm, T):
    """sed_bn_relu_pool_route (what model.routing() returns for a stored block and the parity tests inject into the oracle):
    the codes must be (a) the decisions of the APPLY pass — with the statistics terms forced to 0 its output is non-zero
    exactly on the coded element of every open window — and (b) what torch's relu + max_pool2d decide on the same z wherever
    z is not within rounding of a tie.  Ragged mel / time tails (floor pooling), mel pooling, window order (f, t)."""
    from sed_crnn_amd._lib import lib, ptr, stream_ptr, check
    from oracle import crnn_ref
    B, Cc = 3, 16
    gen = torch.Generator().manual_seed(pf * 100 + pt * 10 + T)
    y = torch.randn(B, T, Fm, Cc, generator=gen)
    y[0, 0, 0, :4] = 0.0                                                   # exact ties inside a window: the first element wins
    scale, shift = torch.rand(Cc, generator=gen) + 0.5, torch.randn(Cc, generator=gen) * 0.3
    scale[3] = -scale[3]                                                   # a negative gamma flips the arg-max
    yg, sc, sh = g(y), g(scale), g(shift)
    Tp, Fp = T // pt, Fm // pf
    route = torch.empty(B, Tp, Fp, Cc, dtype=torch.uint8, device="cuda")
    check(lib().sed_bn_relu_pool_route(ptr(yg), ptr(sc), ptr(sh), ptr(route), B, T, Fm, Cc, pf, pt, stream_ptr()))
    # (a) the apply pass of the backward with sum_g = sum_gx = 0 and dout = 1: dy = scale on the routed element, 0 elsewhere
    dout = torch.ones(B, Tp, Fp, Cc, device="cuda")
    dy = torch.empty_like(yg)
    zero, one = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    dbp = torch.empty(lib().sed_bn_bwd_rows(B, T, pt), Cc, device="cuda")
    check(lib().sed_bn_relu_pool_drop_bwd_apply(ptr(yg), ptr(dout), ptr(sc), ptr(sh), ptr(zero), ptr(one), ptr(zero), ptr(zero),
                                                ptr(dy), ptr(dbp), B, T, Fm, Cc, pf, pt, 0, 0.0, 0, None, stream_ptr()))
    R = crnn_ref.route_mask(route.cpu(), pf, pt, Fm, T)                    # [B,C,F,T]
    taken = (dy.cpu() != 0).permute(0, 3, 2, 1)
    assert torch.equal(taken, R.bool()), "route codes differ from where the apply pass sends the gradient"
    # (b) against torch on z = y*scale + shift (float64: no rounding of its own)
    z = (y.double() * scale.double() + shift.double()).permute(0, 3, 2, 1)  # [B,C,F,T]
    n_gate, n_arg, margin = crnn_ref.audit_routes(z, route.cpu(), pf, pt, tol=1e-6)
    assert n_gate == 0 and margin <= 1e-6
    pooled_t = F.max_pool2d(torch.relu(z), (pf, pt))
    pooled_r = crnn_ref.routed_relu_pool(z, R.double(), pf, pt)
    close(pooled_r, pooled_t, atol=1e-6)
    # the forward kernel's output is the routed value too
    out = ops.bn_relu_pool_drop_fwd(yg, sc, sh, pf, pt)
    close(out.permute(0, 3, 2, 1), pooled_t, atol=1e-5)


@pytest.mark.parametrize("cin,pf,pt", [(1, 1, 2), (2, 1, 2), (4, 1, 2), (1, 5, 1), (2, 2, 2)])
def test_routing_codes_of_the_recomputed_first_block(ops, cin, pf, pt):
    """sed_conv1_route: the decisions of the recomputed block's own passes — equal to the arg-max bits + (pooled > 0) that its
    (1,2)-pool backward reads, and to relu + max_pool2d on a float64 conv + BatchNorm away from ties"""
    from sed_crnn_amd._lib import lib, ptr, stream_ptr, check
    from oracle import crnn_ref
    B, Fm, T, Cc = 2, 20, 16, 32
    gen = torch.Generator().manual_seed(cin * 10 + pf + pt)
    x = torch.randn(B, cin, Fm, T, generator=gen)
    w = torch.randn(Cc, cin, 3, 3, generator=gen) * 0.3
    bias, gamma, beta = torch.randn(Cc, generator=gen) * 0.1, torch.rand(Cc, generator=gen) + 0.5, torch.randn(Cc, generator=gen) * 0.2
    xg, bg = g(x), g(bias)
    wf, _ = ops.conv3x3_pack(g(w))
    conv = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    mean, var = conv.mean((0, 2, 3)), conv.var((0, 2, 3), unbiased=False)
    scale = (gamma.double() / torch.sqrt(var + 1e-5)).float()
    shift = (beta.double() - mean * scale.double()).float()
    sc, sh = g(scale), g(shift)
    Tp, Fp = T // pt, Fm // pf
    route = torch.empty(B, Tp, Fp, Cc, dtype=torch.uint8, device="cuda")
    check(lib().sed_conv1_route(ptr(xg), ptr(wf), ptr(bg), ptr(sc), ptr(sh), ptr(route), B, cin, Fm, T, Cc, pf, pt, stream_ptr()))
    z = conv * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    n_gate, n_arg, margin = crnn_ref.audit_routes(z, route.cpu(), pf, pt, tol=2e-5)
    assert n_gate + n_arg <= 2, (n_gate, n_arg)                             # only genuine ties may differ
    if (pf, pt) == (1, 2):
        out = torch.empty(B, Tp, Fp, Cc, device="cuda")
        bits = torch.empty(out.numel() // 4, dtype=torch.uint8, device="cuda")
        check(lib().sed_conv1_bn_relu_pool_drop_fwd(ptr(xg), ptr(wf), ptr(bg), ptr(sc), ptr(sh), ptr(out), B, cin, Fm, T, Cc, pf, pt,
                                                    0.0, 0, None, ptr(bits), stream_ptr()))
        sel = torch.stack([(bits >> k) & 1 for k in range(4)], -1).reshape(B, Tp, Fp, Cc)      # bit k of the quad's byte
        want = torch.where(out > 0, sel + 1, torch.zeros_like(sel))
        assert torch.equal(route, want.to(torch.uint8))


@pytest.mark.parametrize("B,T,Fm,Cin,Cout", [(2, 16, 40, 128, 128), (3, 15, 40, 32, 128), (1, 8, 64, 64, 256), (2, 23, 13, 32, 128),
                                            (1, 4, 41, 128, 128), (2, 64, 20, 128, 128), (1, 31, 56, 32, 128), (2, 17, 70, 32, 128),
                                            (1, 33, 8, 32, 128), (3, 9, 40, 32, 128)])
def test_inference_conv_with_folded_batchnorm_relu_and_pool_in_the_epilogue(ops, B, T, Fm, Cin, Cout):
    """sed_conv3x3_bn_relu_pool_eval (the eval-mode plan's conv blocks >= 1: `pool(relu(bn(conv(x))))` of sed.py:107 under
    model.eval()) against torch: BatchNorm on running statistics folded into weights and bias, ReLU + (1,2) pool in the
    epilogue.  Odd T (floor pooling drops the last frame), mel widths that tile to 20 / 32 / 14 / 28 columns with ragged last
    tiles, one and two 128-channel groups, negative gamma (the fold must scale before the max), T = 2."""
    from sed_crnn_amd._lib import lib
    if not lib().sed_conv3x3_bn_relu_pool_eval_supported(B, Cin, Fm, T, Cout):
        assert (T, Fm) == (9, 40)          # the one case listed for this: its best even-time-row tile scores > 2 % below the free choice
        with pytest.raises(ValueError):
            ops.conv3x3_bn_relu_pool_eval(torch.empty(B, T, Fm, Cin, device="cuda"), *[torch.empty(1, device="cuda")] * 6)
        return
    gen = torch.Generator().manual_seed(B * 1000 + T * 10 + Fm)
    x = torch.randn(B, Cin, Fm, T, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (3.0 * Cin ** 0.5)
    bias, beta, rm = (torch.randn(Cout, generator=gen) * 0.3 for _ in range(3))
    gamma = torch.rand(Cout, generator=gen) + 0.5
    gamma[1], gamma[5] = -0.8, 0.0
    rv = torch.rand(Cout, generator=gen) + 0.5
    ref = F.max_pool2d(torch.relu(F.batch_norm(F.conv2d(x.double(), w.double(), bias.double(), padding=1), rm.double(), rv.double(),
                                               gamma.double(), beta.double(), training=False, eps=1e-5)), (1, 2))
    out = ops.conv3x3_bn_relu_pool_eval(g(x.permute(0, 3, 2, 1)), g(w), g(bias), g(gamma), g(beta), g(rm), g(rv))
    assert out.shape == (B, T // 2, Fm, Cout)
    close(out.permute(0, 3, 2, 1), ref, atol=2e-5 * max(1.0, float(ref.abs().max())), rtol=1e-5)
    out2 = ops.conv3x3_bn_relu_pool_eval(g(x.permute(0, 3, 2, 1)), g(w), g(bias), g(gamma), g(beta), g(rm), g(rv))
    assert torch.equal(out, out2)
    # the Winograd kernel with the same epilogue (the eval plan's default where the shape takes it: 128 input channels, even T and F)
    if lib().sed_conv3x3_wino_rows(B, Cin, Fm, T, Cout) > 0:
        outw = ops.conv3x3_bn_relu_pool_eval(g(x.permute(0, 3, 2, 1)), g(w), g(bias), g(gamma), g(beta), g(rm), g(rv), wino=True)
        close(outw.permute(0, 3, 2, 1), ref, atol=2e-5 * max(1.0, float(ref.abs().max())), rtol=1e-5)


GEMM_CASES = [(64, 64, 64), (130, 70, 50), (4096, 96, 320), (256, 384, 5120), (384, 640, 512), (33, 17, 9), (512, 768, 1024)]


@pytest.mark.parametrize("M,N,K", GEMM_CASES)
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_f32_layouts(ops, M, N, K, ta, tb):
    gen = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=gen)
    Bm = torch.randn(K, N, generator=gen)
    bias = torch.randn(N, generator=gen)
    Ag = g(A.t()).t() if ta else g(A)            # ta: stored [K][M] (contiguous along i)
    Bg = g(Bm.t()).t() if tb else g(Bm)          # tb: stored [N][K] (contiguous along k)
    ref = A.double() @ Bm.double() + bias.double()
    out = ops.gemm(Ag, Bg, bias=g(bias))
    tol = 2e-6 * K ** 0.5 * 10
    close(out, ref.float(), atol=tol * 4, rtol=1e-4)
    out2 = ops.gemm(Ag, Bg, out=out.clone(), beta=1.0)
    close(out2, (ref + A.double() @ Bm.double()).float(), atol=tol * 8, rtol=1e-4)


@pytest.mark.parametrize("M,N,K", [(384, 128, 4096), (768, 256, 4096), (96, 32, 2048), (48, 16, 1024), (130, 70, 1500)])
def test_gemm_split_k_is_exact_sum_and_deterministic(ops, M, N, K):
    gen = torch.Generator().manual_seed(M + K)
    A, Bm = torch.randn(K, M, generator=gen), torch.randn(K, N, generator=gen)      # dW = A^T B, both row-contiguous
    out = ops.gemm_ws(g(A).t(), g(Bm))
    ref = (A.double().t() @ Bm.double()).float()
    close(out, ref, atol=2e-3, rtol=1e-4)
    assert torch.equal(out, ops.gemm_ws(g(A).t(), g(Bm)))


def test_gemm_strided_column_block(ops):
    """the GRU projection writes into a column block of gi [M, 6H] (ldc > N) and reads strided operands"""
    M, K, H = 96, 80, 8
    gen = torch.Generator().manual_seed(1)
    x, w, b = torch.randn(M, K, generator=gen), torch.randn(3 * H, K, generator=gen), torch.randn(3 * H, generator=gen)
    gi = torch.zeros(M, 6 * H).cuda()
    ops.gemm(g(x), g(w).t(), bias=g(b), out=gi[:, 3 * H:])
    close(gi[:, 3 * H:], x @ w.t() + b, atol=1e-4)
    assert float(gi[:, :3 * H].abs().max()) == 0.0


@pytest.mark.parametrize("M,K,N,relu", [(64, 64, 1, False), (100, 16, 8, True), (33, 256, 6, False), (7, 8, 1, True)])
def test_linear_small_fwd_bwd(ops, M, K, N, relu):
    gen = torch.Generator().manual_seed(M + K + N)
    x, W, b = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen) * 0.3, torch.randn(N, generator=gen)
    xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
    y_ref = F.linear(xr, Wr, br)
    if relu:
        y_ref = torch.relu(y_ref)
    dy = torch.randn(M, N, generator=gen)
    y_ref.backward(dy)
    y = ops.linear_fwd(g(x), g(W), g(b), relu)
    close(y, y_ref, atol=1e-5, rtol=1e-4)
    dx, dW, db = ops.linear_bwd(g(x), g(W), y, g(dy), relu)
    close(dx, xr.grad, atol=1e-5, rtol=1e-4)
    close(dW, Wr.grad, atol=1e-4, rtol=1e-4)
    close(db, br.grad, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("B,T,H", [(2, 8, 32), (5, 8, 16), (3, 4, 8), (6, 16, 128), (4, 6, 20), (2, 5, 64), (3, 4, 136)])
def test_gru_recurrence_fwd_bwd(ops, B, T, H):
    In = 24
    torch.manual_seed(H + B)
    ref = torch.nn.GRU(In, H, batch_first=True, bidirectional=True)
    x = torch.randn(B, T, In, requires_grad=True)
    out_ref, _ = ref(x)
    dout = torch.randn(B, T, 2 * H)
    out_ref.backward(dout)
    wih = [ref.weight_ih_l0, ref.weight_ih_l0_reverse]
    whh = [ref.weight_hh_l0, ref.weight_hh_l0_reverse]
    bih = [ref.bias_ih_l0, ref.bias_ih_l0_reverse]
    bhh = [ref.bias_hh_l0, ref.bias_hh_l0_reverse]
    gi = torch.stack([x.detach() @ wih[d].detach().t() + bih[d].detach() for d in range(2)], dim=2)   # [B,T,2,3H]
    whh_g, bhh_g = [g(w.detach()) for w in whh], [g(b.detach()) for b in bhh]
    out, saved = ops.gru_seq_fwd(g(gi), whh_g, bhh_g)
    close(out, out_ref, atol=2e-5, rtol=1e-4)
    dgi, dgh, dbih, dbhh = ops.gru_seq_bwd(g(dout), saved, whh_g, want_bias=True)
    dgi2, dgh2 = ops.gru_seq_bwd(g(dout), saved, whh_g)
    assert torch.equal(dgi, dgi2) and torch.equal(dgh, dgh2)
    for d in range(2):
        np.testing.assert_allclose(dbih[d].cpu().numpy(), bih[d].grad.numpy(), atol=2e-4, rtol=1e-3)
        np.testing.assert_allclose(dbhh[d].cpu().numpy(), bhh[d].grad.numpy(), atol=2e-4, rtol=1e-3)
    dgi_c, dgh_c = dgi.cpu(), dgh.cpu()
    hprev = saved[:, :, :, 4, :].cpu()
    for d in range(2):
        a, c = dgi_c[:, :, d].reshape(-1, 3 * H), dgh_c[:, :, d].reshape(-1, 3 * H)
        np.testing.assert_allclose((a.t() @ x.detach().reshape(-1, In)).numpy(), wih[d].grad.numpy(), atol=2e-4, rtol=1e-3)
        np.testing.assert_allclose((c.t() @ hprev[:, :, d].reshape(-1, H)).numpy(), whh[d].grad.numpy(), atol=2e-4, rtol=1e-3)
        np.testing.assert_allclose(a.sum(0).numpy(), bih[d].grad.numpy(), atol=2e-4, rtol=1e-3)
        np.testing.assert_allclose(c.sum(0).numpy(), bhh[d].grad.numpy(), atol=2e-4, rtol=1e-3)
    dx = sum(dgi_c[:, :, d] @ wih[d].detach() for d in range(2))
    np.testing.assert_allclose(dx.numpy(), x.grad.numpy(), atol=2e-5, rtol=1e-3)


def test_losses_vs_torch(ops):
    gen = torch.Generator().manual_seed(3)
    x = (torch.randn(7, 9, 1, generator=gen) * 3).requires_grad_(True)
    t = (torch.rand(7, 9, 1, generator=gen) > 0.6).float()
    ref = F.binary_cross_entropy_with_logits(x, t)
    ref.backward()
    loss, d, p = ops.loss_fwd_bwd(g(x.detach()), g(t), "bce")
    assert abs(loss.item() - ref.item()) < 1e-6
    close(d, x.grad, atol=1e-7, rtol=1e-4)
    close(p, torch.sigmoid(x.detach()), atol=1e-6)
    for red in ("mean", "sum"):
        x.grad = None
        pr = torch.sigmoid(x)
        pt = torch.where(t == 1, pr, 1 - pr)
        fl = -0.25 * (1 - pt) ** 2.0 * torch.log(pt + 1e-12)
        fl = fl.mean() if red == "mean" else fl.sum()
        fl.backward()
        loss, d, _ = ops.loss_fwd_bwd(g(x.detach()), g(t), "focal", 0.25, 2.0, red)
        assert abs(loss.item() - fl.item()) < 1e-5 * max(1.0, abs(fl.item()))
        close(d, x.grad, atol=1e-6, rtol=1e-3)


def test_adam_and_clip_vs_torch(ops):
    gen = torch.Generator().manual_seed(4)
    n = 10007
    p0 = torch.randn(n, generator=gen)
    for wd, max_norm in ((0.0, None), (1e-4, 1.0)):
        pr = p0.clone().requires_grad_(True)
        opt = torch.optim.Adam([pr], lr=1e-3, weight_decay=wd)
        pg = g(torch.cat([p0, torch.zeros(1)]))[:n]          # arena view (aligned base)
        m, v = torch.zeros_like(pg), torch.zeros_like(pg)
        for step in range(1, 4):
            grad = torch.randn(n, generator=gen) * (0.1 * step)
            pr.grad = grad.clone()
            if max_norm:
                torch.nn.utils.clip_grad_norm_([pr], max_norm)
            opt.step()
            gg = g(torch.cat([grad, torch.zeros(1)]))
            coef = None
            if max_norm:
                nc = ops.grad_norm_clip_coef(gg, max_norm)
                assert abs(nc[0].item() - grad.norm().item()) < 1e-3
                coef = nc[1:2]
            ops.adam_step(pg, gg[:n], m, v, 1e-3, 0.9, 0.999, 1e-8, wd, step, coef)
            close(pg, pr.detach(), atol=2e-6, rtol=1e-5)


def test_logmel_vs_numpy_restatement(ops):
    from oracle import logmel_ref
    from sed_crnn_amd import feature
    rng = np.random.RandomState(0)
    n = 44100 // 2
    tt = np.arange(n) / 44100.0
    y = (0.3 * np.sin(2 * np.pi * 440 * tt) + 0.1 * np.sin(2 * np.pi * 3000 * tt) + 0.05 * rng.randn(n)).astype(np.float32)
    for pad in ("constant", "reflect"):
        ref = logmel_ref.mbe(y, pad_mode=pad)
        out = feature.mbe(torch.from_numpy(y).cuda(), pad_mode=pad).cpu().numpy()
        assert out.shape == ref.shape == (1 + n // 1024, 40)
        np.testing.assert_allclose(out, ref, atol=1e-3, rtol=1e-4)
    mu, sd = logmel_ref.standardize_fit(ref)
    out = feature.mbe(torch.from_numpy(y).cuda(), pad_mode="reflect", mean=torch.from_numpy(mu).float().cuda(),
                      std=torch.from_numpy(sd).float().cuda()).cpu().numpy()
    np.testing.assert_allclose(out, (ref - mu) / sd, atol=2e-3, rtol=1e-3)


@pytest.mark.parametrize("n_mels", [1, 8, 64, 96, 128])
def test_logmel_other_mel_counts_vs_numpy_restatement(ops, n_mels):
    """filterbanks other than the fork's 40 bins (BASELINE config 5 uses 128): librosa's Slaney bank of `n_mels` bins, both
    plans of the mel stage, against the numpy restatement (parity unpinned by the reference like the 40-bin case)"""
    from oracle import logmel_ref
    from sed_crnn_amd import feature
    rng = np.random.RandomState(n_mels)
    n = 30000
    tt = np.arange(n) / 44100.0
    y = (0.2 * np.sin(2 * np.pi * 700 * tt) + 0.1 * np.sin(2 * np.pi * 9000 * tt) + 0.05 * rng.randn(n)).astype(np.float32)
    ref = logmel_ref.mbe(y, n_mels=n_mels, pad_mode="reflect")
    out = feature.mbe(torch.from_numpy(y).cuda(), n_mels=n_mels, pad_mode="reflect").cpu().numpy()
    assert out.shape == ref.shape == (1 + n // 1024, n_mels)
    np.testing.assert_allclose(out, ref, atol=1e-3, rtol=1e-4)


def test_logmel_edges_odd_hop_custom_bank_and_shift_invariance(ops):
    """the cold paths of the log-mel kernel and two properties that hold at any length:
    * signals shorter than one frame, an odd number of frames (the idle half wave), an odd hop (guarded loads);
    * a filterbank that is NOT two-band (overlapping rectangles): the list plan; an all-zero band gives -inf (log 0);
    * shifting the signal by one hop shifts the interior frames by one, bit for bit (same samples, same arithmetic)."""
    from oracle import logmel_ref
    from sed_crnn_amd import feature
    rng = np.random.RandomState(5)
    for n in (1, 700, 1024, 1025, 3 * 1024 + 7):
        y = rng.randn(n).astype(np.float32)
        for pad in ("constant", "reflect"):
            if pad == "reflect" and n < 1025:
                continue                                   # numpy cannot reflect-pad by more than the signal length
            ref = logmel_ref.mbe(y, pad_mode=pad)
            out = feature.mbe(torch.from_numpy(y).cuda(), pad_mode=pad).cpu().numpy()
            assert out.shape == ref.shape == (1 + n // 1024, 40)
            np.testing.assert_allclose(out, ref, atol=1e-3, rtol=1e-4, err_msg=f"n={n} {pad}")
    y = rng.randn(9000).astype(np.float32)
    ref = logmel_ref.mbe(y, hop=999)
    out = feature.mbe(torch.from_numpy(y).cuda(), hop=999).cpu().numpy()
    assert out.shape == ref.shape == (1 + 9000 // 999, 40)
    np.testing.assert_allclose(out, ref, atol=1e-3, rtol=1e-4)
    # custom bank: 12 overlapping rectangular bands of width 200 (three bands per bin) + one empty band
    fb = np.zeros((13, 1025), np.float32)
    for m in range(12):
        fb[m, 70 * m: 70 * m + 200] = 1.0 / (m + 1)
    win = logmel_ref.hann_periodic(2048)
    tb = feature.build_tables(win, fb, "cuda")
    assert int(tb[5]) == 0                                 # header word 5: the list plan was chosen
    assert int(feature._tables(0, 44100, 2048, 40)[5]) == 1          # librosa's bank takes the two-band plan
    y = rng.randn(8 * 1024).astype(np.float32)
    p = logmel_ref.stft_power(y)
    with np.errstate(divide="ignore"):
        ref = np.log(p @ fb.T)
    out = feature.mbe(torch.from_numpy(y).cuda(), tables=tb).cpu().numpy()
    assert np.isneginf(out[:, 12]).all() and np.isneginf(ref[:, 12]).all()
    np.testing.assert_allclose(out[:, :12], ref[:, :12], atol=1e-3, rtol=1e-4)
    # shift invariance on a long signal (2 000 frames)
    y = torch.randn(2000 * 1024 + 1024, generator=torch.Generator().manual_seed(1)).cuda()
    a = feature.mbe(y)
    b = feature.mbe(y[1024:].contiguous())
    assert torch.equal(a[3:1990], b[2:1989])


@pytest.mark.parametrize("width,n_mels", [(60, 40), (120, 40), (100, 80), (64, 128)])
def test_logmel_large_list_plans_run_with_fewer_waves_per_workgroup(ops, width, n_mels):
    """round-2 advisor: sed_logmel_build_tables accepts list plans up to 8 192 non-zeros, but the 12-wave launch only has room
    for ~3 200 beside its FFT scratch; larger plans now run with 8 / 4 / 2 waves per workgroup instead of being refused at
    launch.  Dense-ish banks of 2 400 .. 8 192 non-zeros (overlapping rectangles) against the numpy power spectrum; a
    caller-made blob with a damaged header is refused by feature.mbe."""
    from oracle import logmel_ref
    from sed_crnn_amd import feature
    rng = np.random.RandomState(width + n_mels)
    fb = np.zeros((n_mels, 1025), np.float32)
    step = (1025 - width) // n_mels
    for m in range(n_mels):
        fb[m, step * m: step * m + width] = rng.rand(width).astype(np.float32) + 0.1
    nnz = int(np.count_nonzero(fb))
    tb = feature.build_tables(logmel_ref.hann_periodic(2048), fb, "cuda")
    assert int(tb[5]) == 0 and 2000 < nnz <= 8192
    y = rng.randn(37 * 1024 + 5).astype(np.float32)
    ref = np.log(logmel_ref.stft_power(y) @ fb.T)
    out = feature.mbe(torch.from_numpy(y).cuda(), tables=tb).cpu().numpy()
    assert out.shape == ref.shape
    np.testing.assert_allclose(out, ref, atol=1e-3, rtol=1e-4, err_msg=f"{nnz} non-zeros")
    bad = tb.clone()
    bad[2] = 4096                                           # iters beyond the blob
    with pytest.raises(ValueError, match="bad header"):
        feature.mbe(torch.from_numpy(y).cuda(), tables=bad)
    with pytest.raises(ValueError, match="bad header"):
        feature.mbe(torch.from_numpy(y).cuda(), tables=tb[:-4].contiguous())


@pytest.mark.parametrize("plan", ["two-band (12 waves, PCM prefetched before the mel pass)", "list (8 waves, PCM double-buffered a whole iteration ahead)"])
def test_logmel_steady_state_of_the_persistent_loop_on_a_long_signal(ops, plan):
    """the log-mel kernel is persistent (one workgroup per CU looping over frame pairs) and software-pipelined: the NEXT pair's
    PCM is loaded while the current pair is transformed — into the FFT registers before the mel pass (12 waves per CU) or into
    a second register set a whole iteration ahead (the <= 8-wave launches of large list plans).  Short clips never reach a
    second iteration of that loop; this one runs 13 312 frames (> 2 x 256 workgroups x 12 waves x 2 frames) and compares EVERY
    frame with the numpy restatement (feature.py:55-59; both pad modes' edges included)."""
    from oracle import logmel_ref
    from sed_crnn_amd import feature
    rng = np.random.RandomState(7)
    n = 1024 * 13311 + 517
    tt = np.arange(n, dtype=np.float64) / 44100.0
    y = (0.2 * np.sin(2 * np.pi * (200.0 + 3.0 * tt) * tt) + 0.05 * rng.randn(n)).astype(np.float32)
    yg = torch.from_numpy(y).cuda()
    if plan.startswith("two-band"):
        ref = logmel_ref.mbe(y, pad_mode="reflect")
        out = feature.mbe(yg, pad_mode="reflect").cpu().numpy()
    else:
        fb = np.zeros((40, 1025), np.float32)
        for m in range(40):
            fb[m, 23 * m: 23 * m + 100] = rng.rand(100).astype(np.float32) + 0.1          # 4 000 non-zeros: a list plan that runs 8 waves
        tb = feature.build_tables(logmel_ref.hann_periodic(2048), fb, "cuda")
        assert int(tb[5]) == 0
        ref = np.log(logmel_ref.stft_power(y) @ fb.T)
        out = feature.mbe(yg, tables=tb).cpu().numpy()
    assert out.shape == ref.shape == (13312, 40)
    np.testing.assert_allclose(out, ref, atol=1e-3, rtol=1e-4)
    assert np.array_equal(out, (feature.mbe(yg, pad_mode="reflect") if plan.startswith("two-band") else feature.mbe(yg, tables=tb)).cpu().numpy())


@pytest.mark.parametrize("B,Cin,Fm,T,Cout", [(2, 128, 40, 16, 128), (1, 32, 8, 8, 64), (3, 64, 40, 6, 32), (2, 128, 128, 8, 128)])
def test_conv3x3_bf16x3_experiment_forward_and_dgrad(ops, B, Cin, Fm, T, Cout):
    """the opt-in 3-term bf16-split MFMA path (mode 1): forward, statistics partials and the data gradient (same kernel,
    flipped/transposed weights) against torch fp32 on CPU.  Its error model is ~4e-6 relative to the magnitude of the
    K = 9*Cin sum (exact-fp32 path: 3e-7); the tolerance below is 20x tighter than what one bf16 term alone would give."""
    torch.manual_seed(B * 7 + Cin)
    x = torch.randn(B, T, Fm, Cin)
    w = torch.randn(Cout, Cin, 3, 3) / (3.0 * Cin ** 0.5)
    bias = torch.randn(Cout)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 2, 1), w, bias, padding=1).permute(0, 3, 2, 1).contiguous()   # [B,T,F,Cout]
    wf, wd = ops.conv3x3_pack(w.cuda(), mode=1)
    y, stat = ops.conv3x3_fwd(x.cuda(), wf, bias.cuda(), False, mode=1)
    scale = float(ref.abs().mean())
    err = (y.cpu() - ref).abs()
    assert float(err.max()) < 1e-4 * scale * 10 and float(err.mean()) < 2e-5 * scale, (float(err.max()), float(err.mean()), scale)
    s = stat.sum(0).cpu()
    torch.testing.assert_close(s[0], ref.sum((0, 1, 2)), rtol=1e-3, atol=2e-2)
    torch.testing.assert_close(s[1], (ref * ref).sum((0, 1, 2)), rtol=1e-3, atol=2e-2)
    dy = torch.randn(B, T, Fm, Cout)
    dref = torch.nn.grad.conv2d_input((B, Cin, Fm, T), w, dy.permute(0, 3, 2, 1).contiguous(), padding=1).permute(0, 3, 2, 1)
    dx, _ = ops.conv3x3_fwd(dy.cuda(), wd, None, False, want_stats=False, mode=1)
    derr = (dx.cpu() - dref).abs()
    dscale = float(dref.abs().mean())
    assert float(derr.max()) < 1e-3 * dscale and float(derr.mean()) < 2e-5 * dscale, (float(derr.max()), float(derr.mean()), dscale)
    # and the exact-fp32 path on the same inputs is an order of magnitude closer
    wf0, _ = ops.conv3x3_pack(w.cuda())
    y0, _ = ops.conv3x3_fwd(x.cuda(), wf0, bias.cuda(), False)
    assert float((y0.cpu() - ref).abs().mean()) < 0.3 * float(err.mean()) + 1e-9


@pytest.mark.parametrize("B,Cin,Fm,T,Cout", [(2, 128, 40, 16, 128), (1, 32, 8, 8, 128), (3, 64, 20, 7, 128), (2, 32, 10, 33, 256),
                                             (2, 128, 46, 5, 128), (1, 32, 5, 3, 128)])
def test_conv3x3_bf16x3_experiment_weight_gradient(ops, B, Cin, Fm, T, Cout):
    """the opt-in weight gradient on the 3-term bf16 split (mode 1 of sed_conv3x3_wgrad_ex; positions gathered into the
    MFMA k index with transposed LDS reads) against torch in float64: odd mel widths and time lengths exercise the padded
    k-steps and the clamped halo addresses; the sum runs over B*T*F positions, so the error is measured against the
    magnitude of that sum like the forward's."""
    torch.manual_seed(B * 11 + Cin + T)
    x = torch.randn(B, T, Fm, Cin)
    dy = torch.randn(B, T, Fm, Cout)
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 2, 1).double(), (Cout, Cin, 3, 3), dy.permute(0, 3, 2, 1).double(), padding=1)
    dw = ops.conv3x3_wgrad(x.cuda(), dy.cuda(), False, mode=1).cpu()
    dw0 = ops.conv3x3_wgrad(x.cuda(), dy.cuda(), False).cpu()
    scale = float(ref.abs().mean())
    err, err0 = (dw.double() - ref).abs(), (dw0.double() - ref).abs()
    assert float(err.max()) < 2e-4 * scale and float(err.mean()) < 2e-5 * scale, (float(err.max()), float(err.mean()), scale)
    assert float(err0.mean()) < float(err.mean())                       # the exact path stays the closer one
    assert torch.equal(dw, ops.conv3x3_wgrad(x.cuda(), dy.cuda(), False, mode=1).cpu())      # fixed-order reduction: run to run identical


@pytest.mark.parametrize("mode", [0, 1])
def test_conv3x3_forward_is_bitwise_stable_under_memory_pressure(ops, mode):
    """The forward kernel prefetches weights and halo tiles from inline asm with hand-counted `s_waitcnt vmcnt`.  A count that
    is one too large reads a register before its load has landed, and whether that shows depends on memory latency: so the
    same launch is repeated while a second stream streams 1 GB copies through HBM, and every result must equal the first
    bit for bit (and torch within the usual tolerance)."""
    torch.manual_seed(3)
    B, T, Fm, Cc = 32, 64, 40, 128
    x = torch.randn(B, T, Fm, Cc, device="cuda")
    w = torch.randn(Cc, Cc, 3, 3, device="cuda") / 34.0
    bias = torch.randn(Cc, device="cuda")
    wf, _ = ops.conv3x3_pack(w, mode=mode)                # mode 1: the bf16x3 experiment kernel has the same hand-counted structure
    y0, st0 = ops.conv3x3_fwd(x, wf, bias, False, mode=mode)
    y0, st0 = y0.clone(), st0.clone()
    ref = F.conv2d(x.permute(0, 3, 2, 1), w, bias, padding=1).permute(0, 3, 2, 1)
    close(y0, ref, atol=3e-5 if mode == 0 else 3e-4, rtol=1e-4)
    big = torch.empty(256 << 20, device="cuda")          # 1 GiB
    side = torch.cuda.Stream()
    for it in range(24):
        with torch.cuda.stream(side):
            for _ in range(2):
                big[: 128 << 20].copy_(big[128 << 20:], non_blocking=True)
        y, st = ops.conv3x3_fwd(x, wf, bias, False, mode=mode)
        assert torch.equal(y, y0) and torch.equal(st, st0), it
    torch.cuda.synchronize()


def test_winograd_kernels_are_bitwise_stable_under_memory_pressure(ops):
    """The Winograd kernels leave every wait to hipcc, but their patch arrives by LDS-DMA (counted in vmcnt) under a written-out
    instruction order with one barrier per slice, and the weight gradient stages its tiles through registers two chunks ahead of
    their commit: a read that overtakes its data would depend on memory latency.  Same protocol as above: forward, data gradient
    with the BatchNorm epilogue and weight gradient are repeated while a second stream streams 1 GB copies through HBM, and every
    result must equal the first bit for bit."""
    torch.manual_seed(4)
    B, T, Fm, Cc = 32, 64, 40, 128
    x = torch.randn(B, T, Fm, Cc, device="cuda")
    dy = torch.randn(B, T, Fm, Cc, device="cuda") * 0.1
    w = torch.randn(Cc, Cc, 3, 3, device="cuda") / 34.0
    bias = torch.randn(Cc, device="cuda")
    uf, ud = ops.conv3x3_wino_pack(w)
    gamma, beta = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.2
    yb = torch.randn(B, 2 * T, Fm, Cc, device="cuda")                       # conv output of the block below, pool (1,2)
    flat = yb.reshape(-1, Cc)
    part = torch.stack([flat.sum(0), (flat ** 2).sum(0)]).reshape(1, 2, Cc).contiguous()
    mean, rstd, scale, shift = ops.bn_finalize_train(part, flat.shape[0], gamma, beta, torch.zeros(Cc).cuda(), torch.ones(Cc).cuda())
    pooled = ops.bn_relu_pool_drop_fwd(yb, scale, shift, 1, 2, drop_p=0.5, seed=5)

    def run():
        y, st = ops.conv3x3_wino_fwd(x, uf, bias, Cc)
        dx, sg, sgx = ops.conv3x3_dgrad_bnred(dy, ud, pooled, gamma, beta, yb, mean, rstd, 1, 2, drop_p=0.5, scale=scale, shift=shift, wino=True)
        dw = ops.conv3x3_wgrad(x, dy, False)
        return [t.clone() for t in (y, st, dx, sg, sgx, dw)]

    first = run()
    big = torch.empty(256 << 20, device="cuda")          # 1 GiB
    side = torch.cuda.Stream()
    for it in range(16):
        with torch.cuda.stream(side):
            for _ in range(2):
                big[: 128 << 20].copy_(big[128 << 20:], non_blocking=True)
        for a, b_ in zip(run(), first):
            assert torch.equal(a, b_), it
    torch.cuda.synchronize()
