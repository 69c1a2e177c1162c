"""Oracle parity of a TRAINING step at the BASELINE sizes (round-2 verdict, item 1): the reference's step is
``out = model(xb); loss.backward(); optim.step()`` (sed.py:134-137) with dropout 0.5 after every block and BatchNorm on batch
statistics, so that is what is compared here, at

  * config 1: the reference's own net ``TimePooledCRNN()`` (C=128, BiGRU 2x32) at B=16 x 256 frames — one FULL fit step
    (probabilities, loss, every gradient, the post-Adam state, the BatchNorm running statistics), with a float64 run of
    the oracle as the yardstick for the gradients;
  * config 2 (mono) and config 3 (binaural) at B=128 x 256 frames, BiGRU 2x128 — training forward + backward, every gradient;
  * config 5 at its full per-sample extent (4 channels, 128 mel bins, T=512 -> 64 GRU steps, BiGRU 2x256) with a batch the
    CPU oracle finishes in seconds — training forward + backward, every gradient, float64 yardstick.

Dropout stays ACTIVE (p = 0.5): the keep-masks of the HIP run (a counter hash of seed and element index, never stored) are
regenerated with the stand-alone kernel and handed to the oracle, so both sides drop the same elements although their random
number generators differ.  The ROUTING decisions are handed over the same way (round-3 verdict, item 1): a ReLU gate or a
pooling arg-max is a decision on a BatchNorm output z, two correct fp32 implementations round z differently (~1e-7), at
10^7 .. 10^9 elements a few decisions differ, and one differing gate moves its channel's cancelling sum of gradients by a
whole |g| (measured in round 3: 2 gates + 1 arg-max moved every gradient below them by 3-7e-4).  The plan's decisions
(`model.routing(l)` = sed_net_routing: the codes its backward kernels act on) are injected into the oracle
(oracle.crnn_ref.forward_routed) after `audit_routes` has checked each of them against the oracle's OWN z — a differing
decision must be a tie to 2e-5, so a kernel that routed to a wrong element fails here instead of being followed.  With equal
masks and equal routing the two sides compute the same piecewise-linear function and differ by rounding only, so NO bound
below is widened for near-ties any more.

Bounds (stated here, not tuned per case): probabilities 1e-3 (north star), loss 1e-5; gradients (a) element-wise
|d| <= 1e-4 + 1e-4 max|g| + 1e-2 |g|, (b) relative L2 error per parameter <= REL_L2 against the fp32 oracle, and, where a
float64 run is affordable, (c) no further from float64 than 3x torch-float32's own distance (+ a 2e-6 floor), the rule of
test_single_step_gradients_are_as_close_to_float64_as_torch_float32 — for EVERY parameter."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN, MASK64 = 0x9E3779B97F4A7C15, (1 << 64) - 1
REL_L2 = 5e-5          # measured with injected routing: 4e-6 (configs 2, 3) .. 7e-6 (config 5); before, with the plan's and the
                        # oracle's own decisions differing in a handful of ties: 3e-4 .. 7e-4 under a 5e-3 bound


@pytest.fixture(scope="module")
def sed():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import sed_crnn_amd
    return sed_crnn_amd


def _hip_masks(m, B, F, T, p):
    """the keep-masks of the LAST training forward of ``m`` as NCHW CPU tensors of {0, 1/(1-p)} (block l hashes
    seed + golden*(l+1) over the channels-last pooled index)"""
    from sed_crnn_amd import ops
    masks = []
    for l, (C, (pf, pt)) in enumerate(zip(m.conv_channels, m.pools)):
        ones = torch.ones(B, T, F, C, device="cuda")
        mk = ops.bn_relu_pool_drop_fwd(ones, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), pf, pt, drop_p=p,
                                       seed=(m._seed + GOLDEN * (l + 1)) & MASK64)          # [B,T/pt,F/pf,C]
        masks.append(mk.permute(0, 3, 2, 1).contiguous().cpu())
        keep = float((mk > 0).float().mean())
        assert abs(keep - (1 - p)) < 0.01, (l, keep)
        T, F = T // pt, F // pf
        del ones, mk
    torch.cuda.empty_cache()
    return masks


def _hip_routes(m):
    """the plan's ReLU-gate / arg-max decisions of its last training forward, per conv block (uint8 codes on the CPU)"""
    return [m.routing(l).cpu() for l in range(len(m.conv_channels))]


def _rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _audit_note(audit):
    return "; ".join(f"block {l}: {g} gates + {a} arg-maxima decided differently (all ties, worst margin {w:.1e})"
                     for l, (g, a, w) in enumerate(audit))


def _check_grads(grads_h, g32, tag, g64=None):
    """(a) element-wise; (b) relative L2 per parameter against the fp32 oracle <= REL_L2; (c) with a float64 run: no further
    from float64 than 3x torch-float32's own distance (+ 2e-6) — every parameter, no exemption."""
    worst_l2, worst_ratio, worst_k, rows = 0.0, 0.0, "", []
    for k, gh in grads_h.items():
        g = g32[k]
        gh = gh.detach().cpu()
        gmax = float(g.abs().max())
        err = (gh.double() - g.double()).abs()
        bound = 1e-4 + 1e-4 * gmax + 1e-2 * g.double().abs()
        assert bool((err <= bound).all()), (tag, k, float(err.max()), gmax)
        if k.startswith("convs.") and k.endswith(".bias"):
            continue                      # analytically zero (a bias in front of BatchNorm): rounding noise on both sides
        e = _rel_l2(gh, g)
        if e > worst_l2:
            worst_l2, worst_k = e, k
        assert e <= REL_L2, (tag, k, e)
        if g64 is not None:
            den = float(g64[k].norm()) + 1e-30
            e_h = float((gh.double() - g64[k]).norm()) / den
            e_t = float((g.double() - g64[k]).norm()) / den
            rows.append(f"    {k:28s} HIP {e_h:8.2e}  torch-f32 {e_t:8.2e}")
            worst_ratio = max(worst_ratio, e_h / (e_t + 1e-30))
            assert e_h <= 3.0 * e_t + 2e-6, (tag, k, e_h, e_t)
    print(f"{tag}: worst relative L2 gradient error vs the fp32 oracle {worst_l2:.2e} ({worst_k})"
          + (f"; worst HIP/torch-f32 distance-to-float64 ratio {worst_ratio:.2f}" if g64 is not None else ""))
    if rows:
        print("  distance from the float64 run, relative L2:\n" + "\n".join(rows))


def _train_forward_backward_both(sed, kw, B, T, p, seed, want64):
    """one training forward + BCE + backward on the HIP path (dropout p) and on the oracle under the same masks and routing"""
    from oracle import crnn_ref
    torch.manual_seed(seed)
    ref = crnn_ref.SedNetRef(dropout=0.0, **kw)                     # dropout applied through the injected masks
    m = sed.TimePooledCRNN(dropout=p, **kw)
    m.load_state_dict(ref.state_dict())
    m.cuda().train()
    cin, F = kw.get("in_channels", 1), kw.get("n_mels", 40)
    x, y = crnn_ref.synthetic_batch(B, cin, F, T, T // 8, seed=1234)
    out = m(x.cuda())
    loss = sed.BCEWithLogitsLoss()(out, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    masks, routes = _hip_masks(m, B, F, T, p), _hip_routes(m)
    ref.train()
    t0 = time.time()
    audit = []
    out_r = crnn_ref.forward_routed(ref, x, routes, masks, audit=audit)
    loss_r = crnn_ref.bce_logits(out_r, y)
    loss_r.backward()
    t_oracle = time.time() - t0
    g64 = None
    if want64:
        ref64 = crnn_ref.SedNetRef(dropout=0.0, **kw).double()
        ref64.load_state_dict({k: v.double() if v.dtype.is_floating_point else v for k, v in ref.state_dict().items()})
        # (the running statistics of `ref` have moved by one step; they do not enter a train-mode forward)
        ref64.train()
        o64 = crnn_ref.forward_routed(ref64, x.double(), routes, [mk.double() for mk in masks])
        crnn_ref.bce_logits(o64, y.double()).backward()
        g64 = {k: q.grad for k, q in ref64.named_parameters()}
    dp = float((torch.sigmoid(out).detach().cpu() - torch.sigmoid(out_r).detach()).abs().max())
    print(f"B={B} T={T} {kw}: max |dp| {dp:.2e}, loss {loss.item():.7f} vs {loss_r.item():.7f}, oracle fwd+bwd {t_oracle:.1f} s\n"
          f"  routing: {_audit_note(audit)}")
    assert out.shape == out_r.shape == (B, T // 8, 1)
    assert dp <= 1e-3
    assert abs(loss.item() - loss_r.item()) <= 1e-5
    return m, ref, {k: q.grad for k, q in ref.named_parameters()}, g64


def test_config1_reference_net_full_fit_step_at_B16_T256(sed):
    """config 1 at its own shape: ``TimePooledCRNN()`` defaults (sed.py:82-103) on B=16 x 256 frames, one whole fit step"""
    from oracle import crnn_ref
    B, T, p, lr = 16, 256, 0.5, 1e-3
    torch.manual_seed(0)
    ref = crnn_ref.SedNetRef(conv_channels=128, dropout=0.0)         # dropout through the injected masks
    m = sed.TimePooledCRNN()                                         # the reference signature, defaults (C=128, p=0.5)
    assert m.drops == [0.5, 0.5, 0.5] and m.conv_channels == [128] * 3 and m.gru_hidden == [32, 32]
    m.load_state_dict(ref.state_dict())
    m.cuda()
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    x, y = crnn_ref.synthetic_batch(B, 1, 40, T, T // 8, seed=1234)
    opt = sed.FusedAdam(m.parameters(), lr=lr)
    m.train()
    opt.zero_grad()
    out = m(x.cuda())
    loss = sed.BCEWithLogitsLoss()(out, y.cuda())
    loss.backward()
    grads_h = {k: q.grad.detach().clone() for k, q in m.named_parameters()}
    routes = _hip_routes(m)          # BEFORE the optimiser step: the recomputed first block re-reads the live conv bias
    opt.step()
    torch.cuda.synchronize()
    masks = _hip_masks(m, B, 40, T, p)
    # the oracle's step (fp32) and the float64 yardstick, same masks, same routing decisions
    audit = []
    opt_r = torch.optim.Adam(ref.parameters(), lr=lr)
    loss_r, out_r = crnn_ref.fit_step_with_masks(ref, opt_r, x, y, masks, routes=routes, audit=audit)
    print("config 1 routing:", _audit_note(audit))
    ref64 = crnn_ref.SedNetRef(conv_channels=128, dropout=0.0).double()
    ref64.load_state_dict({k: v.double() if v.dtype.is_floating_point else v for k, v in sd0.items()})
    ref64.train()
    crnn_ref.bce_logits(crnn_ref.forward_routed(ref64, x.double(), routes, [mk.double() for mk in masks]), y.double()).backward()
    dp = float((torch.sigmoid(out).detach().cpu() - torch.sigmoid(out_r)).abs().max())
    print(f"config 1: max |dp| {dp:.2e}, loss {loss.item():.7f} vs oracle {float(loss_r):.7f}")
    assert dp <= 1e-3 and abs(loss.item() - float(loss_r)) <= 1e-5
    g32 = {k: q.grad for k, q in ref.named_parameters()}
    _check_grads(grads_h, g32, "config 1", {k: q.grad for k, q in ref64.named_parameters()})
    # post-Adam state: the first Adam step is -lr * g / (|g| + eps), i.e. -lr * sign(g) wherever |g| >> eps = 1e-8: where both
    # gradients are resolved (|g| > 1e-5) and agree in sign the updated weights agree to rounding; elsewhere they may differ
    # by up to 2 lr.
    sd_h, sd_r = m.state_dict(), ref.state_dict()
    strict_n = total_n = 0
    for k, g in g32.items():
        d = (sd_h[k].cpu().double() - sd_r[k].double()).abs()
        assert float(d.max()) <= 2 * lr + 1e-6, (k, float(d.max()))
        gh = grads_h[k].cpu()
        resolved = (g.abs() > 1e-5) & (gh.abs() > 1e-5) & (torch.sign(g) == torch.sign(gh))
        strict_n += int(resolved.sum())
        total_n += g.numel()
        if resolved.any():
            assert float(d[resolved].max()) <= 2e-6, (k, float(d[resolved].max()))
            moved = (sd_r[k].double() - sd0[k].double()).abs()
            assert float(moved[resolved].min()) >= 0.99 * lr                                # Adam really stepped by lr there
    print(f"config 1: post-Adam weights equal to 2e-6 on {strict_n} of {total_n} coordinates with a resolved gradient sign")
    assert strict_n > 0.05 * total_n
    for k in sd_r:
        if "running" in k:
            np.testing.assert_allclose(sd_h[k].cpu().numpy(), sd_r[k].numpy(), atol=1e-5, rtol=1e-4, err_msg=k)
        if "num_batches_tracked" in k:
            assert int(sd_h[k]) == int(sd_r[k]) == 1


@pytest.mark.parametrize("name,cin", [("config2", 1), ("config3", 2)])
def test_configs_2_and_3_training_forward_backward_at_B128_T256(sed, name, cin):
    """(B=128, 256, 40, C) mono / binaural, 3 x conv128 + BiGRU 2x128, dropout 0.5 active, batch statistics: probabilities,
    loss, every gradient and the running statistics against the oracle at the FULL batch"""
    kw = dict(conv_channels=128, in_channels=cin, n_mels=40, gru_hidden=128)
    m, ref, g32, _ = _train_forward_backward_both(sed, kw, B=128, T=256, p=0.5, seed=2, want64=False)
    _check_grads({k: q.grad for k, q in m.named_parameters()}, g32, name, None)
    sd_h, sd_r = m.state_dict(), ref.state_dict()
    for k in sd_r:
        if "running" in k:
            np.testing.assert_allclose(sd_h[k].cpu().numpy(), sd_r[k].numpy(), atol=1e-5, rtol=1e-4, err_msg=k)
    del m, ref, g32
    torch.cuda.empty_cache()


def test_config5_full_extent_training_forward_backward(sed):
    """4 channels x 128 mel bins x T=512 (64 GRU steps) x BiGRU 2x256 — the stored first block at Cin=4, the mel-tiled conv
    kernels, the H=256 recurrence that streams W_hh, the K = 16384 input projection — at B=3 (the oracle needs seconds),
    dropout 0.5 active, with the float64 yardstick"""
    kw = dict(conv_channels=128, in_channels=4, n_mels=128, gru_hidden=256)
    m, ref, g32, g64 = _train_forward_backward_both(sed, kw, B=3, T=512, p=0.5, seed=5, want64=True)
    _check_grads({k: q.grad for k, q in m.named_parameters()}, g32, "config 5 (B=3)", g64)
    sd_h, sd_r = m.state_dict(), ref.state_dict()
    for k in sd_r:
        if "running" in k:
            np.testing.assert_allclose(sd_h[k].cpu().numpy(), sd_r[k].numpy(), atol=1e-5, rtol=1e-4, err_msg=k)


def test_last_backward_phase_is_ordered_by_a_dependency_not_by_host_timing(sed):
    """round-3 verdict item 2.  In the last backward phase the first block's passes (auxiliary stream) run BESIDE the deferred
    MFMA weight gradient (main stream); the persistent MFMA kernel has to hold its CUs first, or its one-per-CU workgroups
    starve behind the passes' grid (config 5: 8.6 -> 16.8 ms for that kernel, 62 -> 69 ms per step).  Round 3 ordered them with
    a 20 us sleep kernel; now the kernel's workgroups count themselves in and a gate at the head of the auxiliary chain waits
    for them (sed_internal_stream_gate).  Config 5 at its full per-GPU size, both host enqueue orders (SED_NET_AUX_FIRST forces
    the adversarial one): bit-equal gradients and fit-step times within 3 % of each other.
    Since the Winograd kernels this two-stream schedule belongs to the DIRECT plan (SED_NET_DIRECT_CONV, 0x4, set in every run
    below); the default plan runs its conv phase serially on the main stream — its step is reported beside, and its gradients
    must agree with the direct plan's to Winograd rounding."""
    from sed_crnn_amd.trainer import FusedTrainStep
    from oracle import crnn_ref
    B, T = 128, 512
    torch.manual_seed(5)
    m = sed.TimePooledCRNN(conv_channels=128, dropout=0.5, in_channels=4, n_mels=128, gru_hidden=256).cuda()
    x, y = crnn_ref.synthetic_batch(B, 4, 128, T, T // 8, seed=77)
    x, y = x.cuda(), y.cuda()
    st = FusedTrainStep(m, lr=0.0)                      # lr = 0: every step starts from the same weights

    def run(flags, steps=6, warm=3):
        m.plan_flags = flags
        for _ in range(warm):
            st.step(x, y)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        ev[0].record()
        for i in range(steps):
            st.step(x, y)
            ev[i + 1].record()
        m._seed_counter = 10 ** 6                        # the same dropout draw for the gradient comparison
        st.step(x, y)
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
        return ms[len(ms) // 2], m.flat_grads().clone()

    t_main, g_main = run(4)
    t_aux, g_aux = run(4 | 1)                            # SED_NET_AUX_FIRST
    t_nogate, g_nogate = run(4 | 2)                      # SED_NET_NO_GATE (reported only)
    t_race, g_race = run(4 | 3)                          # no gate AND the auxiliary chain enqueued first: the race the gate removes
    t_wino, g_wino = run(0)                              # the default plan (Winograd kernels, serial conv phase)
    m.plan_flags = 0
    print(f"config 5 fit step, direct plan: main-first {t_main:.2f} ms, aux-first {t_aux:.2f} ms; without the gate: main-first {t_nogate:.2f} ms, "
          f"aux-first {t_race:.2f} ms | default (Winograd) plan {t_wino:.2f} ms")
    # two correct fp32 forwards decide a few of the 5e8 ReLU gates / pooling arg-maxes differently on near-ties, and each such
    # decision moves a gradient entry (that is what the routed oracle comparison removes — and each plan passes
    # it on its own, test_training_step_matches_the_routed_oracle...); un-routed, two plans agree to ~1e-4 of the gradient norm
    rel = float((g_wino - g_main).norm() / g_main.norm())
    print(f"direct vs Winograd plan, un-routed: relative L2 of the whole gradient {rel:.2e}")
    assert rel < 5e-4, rel
    assert t_wino < t_main
    assert torch.equal(g_main, g_aux) and torch.equal(g_main, g_nogate) and torch.equal(g_main, g_race)
    assert abs(t_main - t_aux) <= 0.03 * min(t_main, t_aux), (t_main, t_aux)
