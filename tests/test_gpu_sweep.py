"""Seeded sweep of random network shapes, HIP path against the CPU oracle (oracle/crnn_ref.py, pinned by goldens g1-g5):
probabilities, loss, every gradient and the eval forward.  The fixed cases elsewhere pick shapes by hand; this one draws
them (channel counts that do and do not take the MFMA kernels, mel widths that are odd / narrower than a tile / wider
than one, sequence lengths that are not a multiple of the pooling product (floor pooling drops the tail, sed.py:90),
1-3 GRU layers, 1-6 classes, 1-4 conv blocks with time pools of 1, 2 or 4), so a shape-dependent indexing slip in any
kernel shows up as a parity failure rather than in production."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sed():
    import sed_crnn_amd
    return sed_crnn_amd


def _draw(seed):
    r = random.Random(seed)
    pools = tuple(r.choice([1, 2, 2, 2, 4]) for _ in range(r.choice([1, 2, 3, 3, 3, 4])))
    prod = 1
    for p in pools:
        prod *= p
    tp = r.randint(1, 9)                                   # GRU steps
    T = tp * prod + r.randint(0, prod - 1)                 # + a tail the floor pooling drops
    return dict(B=r.choice([1, 2, 3, 5, 8, 17]), Cin=r.choice([1, 1, 2, 3, 4, 6]), F=r.choice([3, 5, 8, 13, 20, 40, 41, 64, 70]),
                T=T, Tp=tp, C=r.choice([4, 8, 16, 32, 32, 64, 128]), H=r.choice([4, 8, 16, 32, 64, 128]),
                L=r.choice([1, 2, 2, 3]), K=r.choice([1, 1, 2, 6]), pools=pools, loss=r.choice(["bce", "bce", "focal"]))


def _cmp(a, b, atol, rtol=0.0, msg=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (msg, a.shape, b.shape)
    err = (a - b).abs()
    bound = atol + rtol * b.abs()
    assert bool((err <= bound).all()), (msg, float(err.max()), float(b.abs().max()))


def hip_routes(m):
    """ReLU and max-pool are discontinuous in their gradient routing: where a BatchNorm output lies within rounding distance
    of 0, or the two largest values of a pooling window within rounding distance of each other, two correct fp32
    implementations can route the gradient differently (torch fp32 against torch fp64 does: one flipped gate moves a conv
    weight gradient by ~1 % of its largest entry at these small sizes).  Rounds 2-3 widened the gradient bound to 5 % of the
    largest entry for every block at or below a near-tie; now the plan's own decisions (``model.routing(l)``, the codes its
    backward kernels act on) are injected into the oracle (oracle.crnn_ref.forward_routed / routed_relu_pool, after
    audit_routes has checked that each differing decision IS a tie), both sides compute the same piecewise-linear function,
    and every gradient keeps the tight bound."""
    return [m.routing(l).cpu() for l in range(len(m.conv_channels))]


ROUTED = {"cases": 0, "with_differing_decisions": 0}


def note_audit(audit):
    """count the cases in which the injected decisions differed from the oracle's own at all (printed by the last sweep test)"""
    ROUTED["cases"] += 1
    ROUTED["with_differing_decisions"] += int(any(g + a for g, a, _ in audit))


@pytest.mark.parametrize("seed", list(range(48)))
def test_random_shape_vs_oracle(sed, seed):
    from oracle import crnn_ref
    c = _draw(seed)
    if c["B"] * c["F"] * c["T"] * c["C"] * max(c["Cin"], c["C"]) > 3e9:      # keep the CPU oracle in seconds
        c["B"] = 2
    torch.manual_seed(1000 + seed)
    kw = dict(conv_channels=c["C"], dropout=0.0, in_channels=c["Cin"], n_mels=c["F"], time_pool=c["pools"],
              gru_hidden=c["H"], gru_layers=c["L"], n_classes=c["K"])
    ref = crnn_ref.SedNetRef(**kw)
    m = sed.TimePooledCRNN(**kw)
    x, y = crnn_ref.synthetic_batch(c["B"], c["Cin"], c["F"], c["T"], c["Tp"], K=c["K"], seed=seed)
    if c["B"] * c["T"] * c["F"] < 4096:      # few samples per channel make BatchNorm's 1/sigma large: keep it well conditioned
        x = x * 3.0
    m.load_state_dict(ref.state_dict())
    m.cuda()
    m.train()
    out = m(x.cuda())
    crit = sed.BCEWithLogitsLoss() if c["loss"] == "bce" else sed.FocalBCELoss()
    lh = crit(out, y.cuda())
    lh.backward()
    ref.train()
    audit = []
    out_r = crnn_ref.forward_routed(ref, x, hip_routes(m), audit=audit)        # the plan's gate / arg-max decisions, audited
    note_audit(audit)
    assert out_r.shape == (c["B"], c["Tp"], c["K"]), (c, out_r.shape)
    lf = crnn_ref.bce_logits if c["loss"] == "bce" else crnn_ref.focal_bce
    lr_ = lf(out_r, y)
    lr_.backward()
    _cmp(torch.sigmoid(out), torch.sigmoid(out_r), atol=1e-3, msg=f"train probabilities {c}")
    assert abs(lh.item() - lr_.item()) < 1e-4, c
    rg = dict(ref.named_parameters())
    _grads_vs(m, lambda k: rg[k].grad, str(c))
    ref.eval()
    m.eval()
    with torch.no_grad():
        _cmp(torch.sigmoid(m(x.cuda())), torch.sigmoid(ref(x)), atol=1e-3, msg=f"eval probabilities {c}")
    # BatchNorm running statistics after the one training forward
    sd, rsd = m.state_dict(), ref.state_dict()
    for k in rsd:
        if "running" in k:
            _cmp(sd[k], rsd[k], atol=1e-5, rtol=1e-4, msg=f"{k} {c}")
        if "num_batches_tracked" in k:
            assert int(sd[k]) == int(rsd[k]) == 1


def _grads_vs(m, want_of, msg):
    """every gradient, one bound: |d| <= 1e-4 + 1e-4 max|g| + 1e-2 |g| (no widening for near-ties: the routing is injected)"""
    for k, p in m.named_parameters():
        g = want_of(k)
        _cmp(p.grad, g, atol=1e-4 + 1e-4 * float(g.abs().max()), rtol=1e-2, msg=f"{k} {msg}")


@pytest.mark.parametrize("seed", list(range(100, 116)))
def test_random_lightning_variant_vs_oracle(sed, seed):
    """crnn_lightning.py:41-73 with drawn widths: two single-layer GRUs of different sizes, dense + ReLU + dense, focal loss."""
    from oracle import crnn_ref
    r = random.Random(seed)
    pools = tuple(r.choice([1, 2, 2, 4]) for _ in range(r.choice([2, 3, 3])))
    prod = 1
    for p in pools:
        prod *= p
    tp = r.randint(1, 10)
    c = dict(B=r.choice([1, 2, 4, 9]), Cin=r.choice([1, 1, 2, 4]), F=r.choice([6, 16, 40, 45]), T=tp * prod + r.randint(0, prod - 1),
             depth=r.choice([4, 8, 16, 32, 64]), g1=r.choice([4, 8, 16, 64]), g2=r.choice([4, 8, 12, 32]), d1=r.choice([1, 3, 8, 16]),
             K=r.choice([1, 1, 3]), pools=pools)
    torch.manual_seed(seed)
    ref = crnn_ref.LightningNetRef(dropout=0.0, in_channels=c["Cin"], n_mels=c["F"], conv_depth=c["depth"], time_pool=pools,
                                   gru1=c["g1"], gru2=c["g2"], dense1=c["d1"], n_classes=c["K"])
    m = sed.LightningTimePooledCRNN(dropout=0.0, in_channels=c["Cin"], n_mels=c["F"], conv_depth=c["depth"], time_pool=pools,
                                    gru1_units=c["g1"], gru2_units=c["g2"], dense1_units=c["d1"], n_classes=c["K"],
                                    seq_len_in=c["T"])
    x, y = crnn_ref.synthetic_batch(c["B"], c["Cin"], c["F"], c["T"], tp, K=c["K"], seed=seed)
    if c["B"] * c["T"] * c["F"] < 4096:
        x = x * 3.0
    m.load_state_dict(ref.state_dict())
    m.cuda()
    m.train()
    out = m(x.cuda())
    lh = sed.FocalBCELoss()(out, y.cuda())
    lh.backward()
    ref.train()
    audit = []
    out_r = crnn_ref.forward_routed(ref, x, hip_routes(m), audit=audit)
    note_audit(audit)
    lr_ = crnn_ref.focal_bce(out_r, y)
    lr_.backward()
    _cmp(torch.sigmoid(out), torch.sigmoid(out_r), atol=1e-3, msg=f"train probabilities {c}")
    assert abs(lh.item() - lr_.item()) < 1e-5, c
    rg = dict(ref.named_parameters())
    _grads_vs(m, lambda k: rg[k].grad, str(c))
    ref.eval()
    m.eval()
    with torch.no_grad():
        _cmp(torch.sigmoid(m(x.cuda())), torch.sigmoid(ref(x)), atol=1e-3, msg=f"eval probabilities {c}")


@pytest.mark.parametrize("seed", list(range(200, 220)))
def test_random_get_model_topology_vs_torch_autograd(sed, seed):
    """`get_model` (README.md:44; no body in the reference, so parity is unpinned by it) with drawn (mel, time) pools per
    block, GRU sizes per layer and dense stacks, against autograd of the same graph assembled from torch.nn.functional:
    mel pooling with ragged mel widths, mixed mel/time pooling, blocks without pooling, up to three dense layers."""
    import torch.nn as nn
    import torch.nn.functional as F
    r = random.Random(seed)
    nb = r.choice([1, 2, 3, 3, 4])
    pools = [r.choice([(1, 1), (1, 2), (2, 1), (2, 2), (5, 1), (3, 2), (1, 4)]) for _ in range(nb)]
    pf_all, pt_all = 1, 1
    for pf, pt in pools:
        pf_all, pt_all = pf_all * pf, pt_all * pt
    Fm = pf_all * r.randint(1, 4) + r.randint(0, pf_all - 1)
    tp = r.randint(1, 8)
    T = tp * pt_all + r.randint(0, pt_all - 1)
    C = r.choice([4, 8, 16, 32, 64, 128])
    hid = [r.choice([4, 8, 16, 32]) for _ in range(r.choice([1, 2, 2, 3]))]
    K = r.choice([1, 2, 6])
    fc = [r.choice([2, 8, 16]) for _ in range(r.choice([0, 1, 1, 2]))] + [K]
    B, cin = r.choice([1, 2, 3, 6]), r.choice([1, 2, 3, 6])
    if B * Fm * T * C * max(cin, C) > 2e9:
        B = 1
    msg = f"pools={pools} F={Fm} T={T} C={C} hid={hid} fc={fc} B={B} cin={cin}"
    torch.manual_seed(seed)
    m = sed.get_model(in_channels=cin, n_mels=Fm, seq_len=T, n_classes=K, conv_channels=C, pools=pools, rnn_hidden=hid,
                      fc=fc, dropout=0.0)
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in m.state_dict().items()}
    m.cuda()
    x = torch.randn(B, cin, Fm, T) * (3.0 if B * T * Fm < 4096 else 1.0)
    y = (torch.rand(B, tp, K) > 0.7).float()
    f_out = Fm
    for pf, _ in pools:
        f_out //= pf
    grus, width = [], C * f_out
    for i, h in enumerate(hid):
        g = nn.GRU(width, h, batch_first=True, bidirectional=True)
        g.load_state_dict({k.split(".", 2)[2]: v.detach() for k, v in sd.items() if k.startswith(f"grus.{i}.")})
        grus.append(g)
        width = 2 * h
    from oracle import crnn_ref
    m.train()
    out = m(x.cuda())
    loss = sed.BCEWithLogitsLoss()(out, y.cuda())
    loss.backward()
    routes, audit = hip_routes(m), []

    def ref_fwd(x):
        h = x
        for l, (pf, pt) in enumerate(pools):
            h = F.conv2d(h, sd[f"convs.{l}.weight"], sd[f"convs.{l}.bias"], padding=1)
            h = F.batch_norm(h, None, None, sd[f"bns.{l}.weight"], sd[f"bns.{l}.bias"], training=True)
            audit.append(crnn_ref.audit_routes(h, routes[l], pf, pt))
            # = F.max_pool2d(torch.relu(h), (pf, pt)) with the plan's (audited) gate / arg-max decisions
            h = crnn_ref.routed_relu_pool(h, crnn_ref.route_mask(routes[l], pf, pt, h.shape[2], h.shape[3]), pf, pt)
        b, c, f, t = h.shape
        h = h.permute(0, 3, 1, 2).reshape(b, t, c * f)
        for g in grus:
            h, _ = g(h)
        for j in range(len(fc)):
            h = F.linear(h, sd[f"fcs.{j}.weight"], sd[f"fcs.{j}.bias"])
            if j + 1 < len(fc):
                h = torch.relu(h)
        return h
    out_r = ref_fwd(x)
    assert out_r.shape == (B, tp, K), msg
    loss_r = F.binary_cross_entropy_with_logits(out_r, y)
    loss_r.backward()
    note_audit(audit)
    _cmp(torch.sigmoid(out), torch.sigmoid(out_r), atol=1e-3, msg=msg)
    assert abs(loss.item() - loss_r.item()) < 1e-4, msg

    def want(k):
        if k.startswith("grus."):
            return dict(grus[int(k.split(".")[1])].named_parameters())[k.split(".", 2)[2]].grad
        return sd[k].grad
    _grads_vs(m, want, msg)


@pytest.mark.parametrize("H,L", [(4, 2), (20, 1), (36, 3), (100, 2), (200, 1), (340, 2)])
def test_unusual_gru_widths_vs_oracle(sed, H, L):
    """hidden sizes off the tuned ones (32 / 128 / 256), up to the largest the recurrence kernel takes (3H <= 1024)"""
    from oracle import crnn_ref
    torch.manual_seed(H)
    kw = dict(conv_channels=8, dropout=0.0, in_channels=1, n_mels=12, gru_hidden=H, gru_layers=L)
    ref = crnn_ref.SedNetRef(**kw)
    m = sed.TimePooledCRNN(**kw)
    x, y = crnn_ref.synthetic_batch(3, 1, 12, 40, 5, seed=H)
    m.load_state_dict(ref.state_dict())
    m.cuda()
    ref.train()
    out_r = ref(x)
    crnn_ref.bce_logits(out_r, y).backward()
    m.train()
    out = m(x.cuda())
    sed.BCEWithLogitsLoss()(out, y.cuda()).backward()
    _cmp(torch.sigmoid(out), torch.sigmoid(out_r), atol=1e-3, msg=f"H={H}")
    rg = dict(ref.named_parameters())
    for k, p in m.named_parameters():
        if k.startswith(("gru.", "fc.")):
            g = rg[k].grad
            _cmp(p.grad, g, atol=1e-4 + 1e-4 * float(g.abs().max()), rtol=1e-2, msg=f"{k} H={H}")


@pytest.mark.parametrize("C,cin", [(4, 1), (12, 2), (20, 1), (36, 3), (100, 1), (132, 2), (256, 1), (512, 4)])
def test_unusual_conv_widths_vs_oracle(sed, C, cin):
    """conv channel counts that are not powers of two (fused first block falls back where its slot layout needs 256 % (C/4) == 0,
    BatchNorm passes with partially filled thread slots), beyond one 128-channel MFMA tile group, and up to 512"""
    from oracle import crnn_ref
    torch.manual_seed(C)
    kw = dict(conv_channels=C, dropout=0.0, in_channels=cin, n_mels=10, gru_hidden=8, gru_layers=1)
    ref = crnn_ref.SedNetRef(**kw)
    m = sed.TimePooledCRNN(**kw)
    x, y = crnn_ref.synthetic_batch(2, cin, 10, 24, 3, seed=C)
    x = x * 3.0
    m.load_state_dict(ref.state_dict())
    m.cuda()
    m.train()
    out = m(x.cuda())
    sed.BCEWithLogitsLoss()(out, y.cuda()).backward()
    ref.train()
    audit = []
    out_r = crnn_ref.forward_routed(ref, x, hip_routes(m), audit=audit)
    note_audit(audit)
    crnn_ref.bce_logits(out_r, y).backward()
    _cmp(torch.sigmoid(out), torch.sigmoid(out_r), atol=1e-3, msg=f"C={C}")
    rg = dict(ref.named_parameters())
    _grads_vs(m, lambda k: rg[k].grad, f"C={C} cin={cin}")
    ref.eval()
    m.eval()
    with torch.no_grad():
        _cmp(torch.sigmoid(m(x.cuda())), torch.sigmoid(ref(x)), atol=1e-3, msg=f"eval C={C}")


def test_routing_injection_tally():
    """how many of the sweep's cases held a decision on which the plan and the oracle's own arithmetic differed (each of
    them audited as a tie): printed so that the log says what the injected routing bought; no bound depends on it"""
    print(f"routing injected in {ROUTED['cases']} sweep cases; in {ROUTED['with_differing_decisions']} of them at least one "
          f"ReLU gate / arg-max was a tie that the two sides had decided differently")


def test_many_random_topologies_run_or_refuse_cleanly(sed):
    """120 drawn topologies (1-4 conv blocks of 4..512 channels, mixed mel/time pools, 1-6 input channels, mel widths 3..160,
    1-3 GRU layers of 4..340 units, 1-3 dense layers, dropout on): a training forward + backward + eval forward each.  Every
    one must either run to finite numbers or be refused with a SedHipError / ValueError that says why — never fault, hang or
    return NaN.  (Parity is the business of the other sweeps; this one hunts launch-configuration slips.)"""
    r = random.Random(4242)
    ran = refused = 0
    for it in range(120):
        nb = r.choice([1, 2, 3, 3, 4])
        pools = [r.choice([(1, 1), (1, 2), (2, 1), (2, 2), (5, 1), (3, 2), (1, 4)]) for _ in range(nb)]
        pf, pt = 1, 1
        for a, b in pools:
            pf, pt = pf * a, pt * b
        Fm = max(1, pf) * r.randint(1, 3) + r.randint(0, 4)
        if r.random() < 0.2:
            Fm = r.choice([96, 128, 160])
        T = pt * r.randint(1, 6) + r.randint(0, pt - 1)
        C = r.choice([4, 8, 12, 16, 32, 36, 64, 100, 128, 256, 512])
        hid = [r.choice([4, 8, 20, 32, 128, 256, 340]) for _ in range(r.choice([1, 2, 3]))]
        K = r.choice([1, 3, 6])
        fc = [r.choice([2, 8, 16]) for _ in range(r.choice([0, 1, 2]))] + [K]
        B, cin = r.choice([1, 2, 5]), r.choice([1, 2, 4, 6])
        if B * Fm * T * C * max(cin, C) > 4e9:
            B, T = 1, pt * 2
        desc = f"#{it} pools={pools} F={Fm} T={T} C={C} hid={hid} fc={fc} B={B} cin={cin}"
        try:
            m = sed.get_model(in_channels=cin, n_mels=Fm, seq_len=T, n_classes=K, conv_channels=C, pools=pools, rnn_hidden=hid,
                              fc=fc, dropout=0.3).cuda()
            x = torch.randn(B, cin, Fm, T, device="cuda")
            m.train()
            out = m(x)
            tp = T // pt
            assert out.shape == (B, tp, K), desc
            y = (torch.rand(B, tp, K, device="cuda") > 0.7).float()
            loss = sed.BCEWithLogitsLoss()(out, y)
            loss.backward()
            assert bool(torch.isfinite(loss)), desc
            for k, p in m.named_parameters():
                assert p.grad is not None and bool(torch.isfinite(p.grad).all()), (desc, k)
            m.eval()
            with torch.no_grad():
                assert bool(torch.isfinite(m(x)).all()), desc
            ran += 1
        except (sed.SedHipError, ValueError) as e:
            assert len(str(e)) > 20, desc              # refused with a reason
            refused += 1
    assert ran >= 100, (ran, refused)                   # the refusals are the LDS-sized corner cases, not the rule
