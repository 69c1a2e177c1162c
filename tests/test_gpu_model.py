"""Module-level parity on the GPU: the HIP network (through sed_net_forward/backward) against
 (a) the golden vectors captured from the imported reference (tests/golden, oracle/make_goldens.py) and
 (b) the CPU oracle (oracle/crnn_ref.py) on identical seeded inputs / weights,
plus size-independent properties at the BASELINE size.  Tolerance for probabilities: 1e-3 (north star);
tighter where stated."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sed():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import sed_crnn_amd
    return sed_crnn_amd


def _sd(d, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith(prefix)}


def _cmp(a, b, atol, rtol=1e-3, msg=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol, err_msg=msg)


def test_g1_reference_sed_net_forward_grads(sed):
    d = load_golden("g1_sed_c8.npz")
    m = sed.TimePooledCRNN(conv_channels=8, dropout=0.0)
    m.load_state_dict(_sd(d))
    m.cuda()
    x, y = torch.from_numpy(d["x"]).cuda(), torch.from_numpy(d["y"]).cuda()
    m.eval()
    _cmp(m(x), d["logits_eval"], atol=1e-4)
    m.train()
    out = m(x)
    loss = sed.BCEWithLogitsLoss()(out, y)
    loss.backward()
    _cmp(out, d["logits_train"], atol=1e-4)
    assert abs(loss.item() - float(d["loss_train"])) < 1e-5
    for k, p in m.named_parameters():
        _cmp(p.grad, d["grad." + k], atol=2e-5, rtol=2e-3, msg=k)
    sd = m.state_dict()
    for k in sd:
        if "running" in k:
            _cmp(sd[k], d["after." + k], atol=1e-5, rtol=1e-4, msg=k)
        if "num_batches" in k:
            assert int(sd[k]) == int(d["after." + k])


def test_g3_fit_loop_trajectory(sed):
    """6 Adam steps through run_epoch (3 batches x 2 epochs) + validation epoch + ER/F1, vs sed.run_epoch."""
    d = load_golden("g3_sed_c8_traj.npz")
    m = sed.TimePooledCRNN(conv_channels=8, dropout=0.0)
    m.load_state_dict(_sd(d, "sd0."))
    m.cuda()
    batches = [(torch.from_numpy(d[f"x{i}"]), torch.from_numpy(d[f"y{i}"])) for i in range(3)]
    opt = sed.FusedAdam(m.parameters(), lr=1e-3)
    crit = sed.BCEWithLogitsLoss()
    losses = [sed.run_epoch(m, batches, crit, opt)[0] for _ in range(2)]
    _cmp(np.asarray(losses), d["train_losses"], atol=1e-3)
    lv, pv, tv = sed.run_epoch(m, batches, crit)
    assert abs(lv - float(d["val_loss"])) < 1e-3
    _cmp(pv, d["val_preds"], atol=1e-3)
    assert pv.dtype == np.float32 and pv.shape == d["val_preds"].shape
    sc = sed.metrics.compute_scores(pv > 0.5, tv, frames_in_1_sec=5)
    assert sc["f1_overall_1sec"] == pytest.approx(float(d["val_f1_1s"]), abs=1e-12)
    assert sc["er_overall_1sec"] == pytest.approx(float(d["val_er_1s"]), abs=1e-12)
    sd = m.state_dict()
    for k, v in _sd(d, "sd6.").items():
        # A conv bias in front of BatchNorm has an analytically ZERO gradient; what both the reference and this
        # build see is rounding noise (~1e-9) that Adam normalises into +-lr steps, so those 8 numbers random-walk
        # differently on any two implementations (and cannot change any output).  Everything else must agree.
        if v.dtype.is_floating_point and not (k.startswith("convs.") and k.endswith(".bias")):
            _cmp(sd[k], v, atol=2e-3, rtol=5e-2, msg=k)


def test_g3_same_trajectory_with_torch_adam(sed):
    """the drop-in autograd path also works with a stock torch optimiser and torch loss"""
    d = load_golden("g3_sed_c8_traj.npz")
    m = sed.TimePooledCRNN(conv_channels=8, dropout=0.0)
    m.load_state_dict(_sd(d, "sd0."))
    m.cuda()
    batches = [(torch.from_numpy(d[f"x{i}"]), torch.from_numpy(d[f"y{i}"])) for i in range(3)]
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    crit = torch.nn.BCEWithLogitsLoss()
    losses = [sed.run_epoch(m, batches, crit, opt)[0] for _ in range(2)]
    _cmp(np.asarray(losses), d["train_losses"], atol=1e-3)


def test_g4_lightning_net_focal(sed):
    d = load_golden("g4_lightning.npz")
    m = sed.LightningTimePooledCRNN(dropout=0.0)
    m.load_state_dict(_sd(d))
    m.cuda()
    assert m.T_out == 8 and m._flat == 640
    x, y = torch.from_numpy(d["x"]).cuda(), torch.from_numpy(d["y"]).cuda()
    m.eval()
    _cmp(m(x), d["logits_eval"], atol=1e-4)
    m.train()
    out = m(x)
    loss = sed.FocalBCELoss()(out, y)
    loss.backward()
    assert abs(loss.item() - float(d["loss_train"])) < 1e-5
    for k, p in m.named_parameters():
        _cmp(p.grad, d["grad." + k], atol=1e-5, rtol=2e-3, msg=k)
    lg, tg = torch.from_numpy(d["focal_logits"]).cuda(), torch.from_numpy(d["focal_targets"]).cuda()
    assert sed.FocalBCELoss()(lg, tg).item() == pytest.approx(float(d["focal_mean"]), rel=1e-5)
    assert sed.FocalBCELoss(reduction="sum")(lg, tg).item() == pytest.approx(float(d["focal_sum"]), rel=1e-5)
    opt = sed.FusedAdam(m.parameters(), lr=float(d["opt_lr"]), weight_decay=float(d["opt_wd"]))
    opt.step()
    sd = m.state_dict()
    for k, v in _sd(d, "sd1.").items():
        if v.dtype.is_floating_point:
            _cmp(sd[k], v, atol=1e-5, rtol=1e-4, msg=k)


def test_g4_crnn_lightning_module_hooks(sed, tmp_path):
    """CRNNLightning mirror: training_step / _aggregate / configure_optimizers against the reference module's outputs"""
    d = load_golden("g4_lightning.npz")
    lm = sed.CRNNLightning(fold_id=1, art_dir=str(tmp_path), dropout=0.0)
    assert list(lm.state_dict().keys())[0] == "model.conv_stack.0.weight"
    lm.model.load_state_dict(_sd(d))
    lm.cuda()
    x, y = torch.from_numpy(d["x"]).cuda(), torch.from_numpy(d["y"]).cuda()
    lm.train()
    loss = lm.training_step((x, y), 0)
    assert abs(loss.item() - float(d["loss_train"])) < 1e-5
    assert lm.logged["train_loss"] is loss
    _cmp(lm._buf["train"]["preds"][0], d["preds_train"], atol=1e-4)
    agg = lm._aggregate("train")
    np.testing.assert_array_equal(agg["cm"], d["agg_cm"])
    got = np.asarray([agg["loss"], agg["f1_frame"], agg["er_frame"], agg["f1_1s"], agg["er_1s"]])
    np.testing.assert_allclose(got, d["agg_vals"], atol=1e-5, equal_nan=True)
    assert lm._buf["train"]["preds"] == []
    cfg = lm.configure_optimizers()
    g0 = cfg["optimizer"].param_groups[0]
    assert g0["lr"] == float(d["opt_lr"]) and g0["weight_decay"] == float(d["opt_wd"])
    assert tuple(g0["betas"]) == tuple(d["opt_betas"]) and g0["eps"] == float(d["opt_eps"])
    sch = cfg["lr_scheduler"]["scheduler"]
    assert sch.factor == float(d["sched_factor"]) and sch.patience == int(d["sched_patience"])
    assert cfg["lr_scheduler"]["monitor"] == str(d["monitor"]) == "val_loss"


def test_fit_lightning_loop_checkpoints_and_early_stop(sed, tmp_path):
    from oracle import crnn_ref
    torch.manual_seed(3)
    lm = sed.CRNNLightning(fold_id=2, art_dir=str(tmp_path / "art"), dropout=0.1)
    data = [crnn_ref.synthetic_batch(8, 1, 40, 64, 8, seed=s) for s in range(3)]
    hist = sed.fit_lightning(lm, data, data[:1], max_epochs=3, early_stop=20, ckpt_dir=str(tmp_path / "ck"))
    assert len(hist) == 3 and all(np.isfinite(h["val_loss"]) for h in hist)
    assert hist[-1]["train_loss"] < hist[0]["train_loss"]
    names = sorted(os.listdir(tmp_path / "ck"))
    assert "last.ckpt" in names and any(n.startswith("epoch000-valer") for n in names) and len(names) == 4
    ck = torch.load(tmp_path / "ck" / "last.ckpt", weights_only=True)
    assert ck["epoch"] == 2 and "model.gru1.weight_ih_l0" in ck["state_dict"]
    assert set(lm.track) >= {"loss_tr", "loss_val", "er_1s_val", "f1_fr_tr"} and len(lm.track["er_1s_val"]) == 3
    assert lm._last_val["cm"].shape == (2, 2) and int(lm._last_val["cm"].sum()) == 8 * 8        # one val batch of 8 x 8 frames
    # early stopping: a validation ER that can never improve (no positives -> nan/inf) stops after `early_stop` epochs
    lm2 = sed.CRNNLightning(fold_id=3, art_dir=str(tmp_path / "art2"), dropout=0.0)
    xz = torch.randn(4, 1, 40, 64)
    hist2 = sed.fit_lightning(lm2, [(xz, torch.zeros(4, 8, 1))], [(xz, torch.zeros(4, 8, 1))], max_epochs=10, early_stop=2)
    assert len(hist2) <= 3


def test_g5_full_width_k1152(sed):
    from oracle import crnn_ref
    d = load_golden("g5_sed_c128.npz")
    ref = crnn_ref.SedNetRef(conv_channels=128, dropout=0.0)
    m = sed.TimePooledCRNN(conv_channels=128, dropout=0.0)
    m.load_state_dict(crnn_ref.rs_state_dict(ref, seed=int(d["weight_seed"])))
    m.cuda()
    x, y = torch.from_numpy(d["x"]).cuda(), torch.from_numpy(d["y"]).cuda()
    m.eval()
    _cmp(m(x), d["logits_eval"], atol=2e-4)
    m.train()
    out = m(x)
    sed.BCEWithLogitsLoss()(out, y).backward()
    _cmp(out, d["logits_train"], atol=2e-4)
    named = dict(m.named_parameters())
    for k in d:
        if k.startswith("grad.") and k[5:] in named:
            _cmp(named[k[5:]].grad, d[k], atol=5e-5, rtol=5e-3, msg=k)
    gw = named["gru.weight_ih_l0"].grad
    _cmp(gw[:4], d["grad.gru.weight_ih_l0.rows0_4"], atol=5e-5, rtol=5e-3)
    _cmp(gw.sum(0), d["grad.gru.weight_ih_l0.colsum"], atol=2e-4, rtol=5e-3)


def _oracle_vs_hip(sed, ref, m, x, y, loss="bce", atol=1e-3):
    """Gradients: rtol 1e-2 / atol 1e-4 for EVERY parameter.  The ReLU-gate / arg-max decisions of the HIP plan
    (``m.routing(l)``) are injected into the oracle (crnn_ref.forward_routed, each differing decision audited as a tie to
    2e-5 of the oracle's own BatchNorm output), so a near-tie that two correct fp32 implementations decide differently no
    longer needs the 5 %-of-the-largest-entry bound rounds 2-3 gave the conv blocks at or below it."""
    from oracle import crnn_ref
    from test_gpu_sweep import hip_routes
    m.load_state_dict(ref.state_dict())
    m.cuda()
    m.train()
    out = m(x.cuda())
    crit = sed.BCEWithLogitsLoss() if loss == "bce" else sed.FocalBCELoss()
    lh = crit(out, y.cuda())
    lh.backward()
    ref.train()
    audit = []
    out_r = crnn_ref.forward_routed(ref, x, hip_routes(m), audit=audit)
    lf = crnn_ref.bce_logits if loss == "bce" else crnn_ref.focal_bce
    lr_ = lf(out_r, y)
    lr_.backward()
    _cmp(torch.sigmoid(out), torch.sigmoid(out_r), atol=atol)
    assert abs(lh.item() - lr_.item()) < 1e-4
    rg = dict(ref.named_parameters())
    for k, p in m.named_parameters():
        _cmp(p.grad, rg[k].grad, atol=1e-4, rtol=1e-2, msg=k)
    ref.eval()
    m.eval()
    with torch.no_grad():
        _cmp(torch.sigmoid(m(x.cuda())), torch.sigmoid(ref(x)), atol=atol)


@pytest.mark.parametrize("cin", [1, 2, 4])
def test_batchnorm_channels_with_a_tiny_gamma_keep_exact_gradients(sed, cin):
    """round-3 advisor: the fused BatchNorm-backward sums recover xhat = (z - beta)/gamma from a block's pooled output, which loses
    eps |beta/gamma| when |gamma| << |beta| (z nearly constant) and does not exist for gamma == 0.  Such channels now take xhat
    from where it is exact — the stored conv output (sed_bn_bwd_finalize_small_gamma for blocks below the top, the cold loop of
    sed_bn_bwd_reduce_pooled for the top block) or, for the recomputed first block, its own tap sums (conv1_wgrad_assemble_k).
    Full-width net (C = 128: every fused path is taken), channels with gamma = 1e-3, -1e-4, 1e-5 and 0 at beta = 0.5 / 0.3 in
    EVERY block, against the oracle routed with the plan's decisions; dgamma of exactly those channels to 2e-4 relative."""
    from oracle import crnn_ref
    from test_gpu_sweep import hip_routes
    torch.manual_seed(31 + cin)
    kw = dict(conv_channels=128, dropout=0.0, in_channels=cin, n_mels=40, gru_hidden=16)
    ref = crnn_ref.SedNetRef(**kw)
    small = {3: (1e-3, 0.5), 5: (-1e-4, 0.5), 7: (0.0, 0.3), 9: (1e-5, 0.5), 64: (2e-3, -0.4)}
    with torch.no_grad():
        for bn in ref.bns:
            for c, (gm, bt) in small.items():
                bn.weight[c], bn.bias[c] = gm, bt
    m = sed.TimePooledCRNN(**kw)
    m.load_state_dict(ref.state_dict())
    m.cuda().train()
    x, y = crnn_ref.synthetic_batch(4, cin, 40, 64, 8, seed=17)
    out = m(x.cuda())
    sed.BCEWithLogitsLoss()(out, y.cuda()).backward()
    ref.train()
    audit = []
    out_r = crnn_ref.forward_routed(ref, x, hip_routes(m), audit=audit)
    crnn_ref.bce_logits(out_r, y).backward()
    _cmp(torch.sigmoid(out), torch.sigmoid(out_r), atol=1e-4)
    rg = dict(ref.named_parameters())
    for k, p in m.named_parameters():
        g = rg[k].grad
        _cmp(p.grad, g, atol=1e-4 + 1e-4 * float(g.abs().max()), rtol=1e-2, msg=k)
    for l in range(3):
        gh, gr = m.bns[l].weight.grad.cpu().double(), ref.bns[l].weight.grad.double()
        scale = float(gr.abs().max())
        for c in small:
            assert abs(float(gh[c] - gr[c])) <= 2e-4 * abs(float(gr[c])) + 2e-6 * scale, (l, c, float(gh[c]), float(gr[c]))
        assert float(gr[3].abs()) > 0                      # the channels are live (their gradient is not trivially 0)


@pytest.mark.parametrize("cin,mel,T,H,C", [(2, 40, 32, 128, 128), (4, 128, 16, 64, 32), (1, 40, 64, 32, 16), (4, 128, 16, 256, 128)])
def test_multichannel_configs_vs_oracle(sed, cin, mel, T, H, C):
    """binaural / 4-channel / 128-mel shapes of BASELINE configs 3 and 5 at sizes the oracle runs in seconds"""
    from oracle import crnn_ref
    torch.manual_seed(7)
    ref = crnn_ref.SedNetRef(conv_channels=C, dropout=0.0, in_channels=cin, n_mels=mel, gru_hidden=H)
    m = sed.TimePooledCRNN(conv_channels=C, dropout=0.0, in_channels=cin, n_mels=mel, gru_hidden=H)
    x, y = crnn_ref.synthetic_batch(3, cin, mel, T, T // 8, seed=5)
    _oracle_vs_hip(sed, ref, m, x, y)


@pytest.mark.parametrize("B,T,C,H", [(1, 8, 8, 8), (3, 24, 8, 16), (2, 40, 16, 8), (5, 72, 8, 32), (7, 8, 128, 128)])
def test_ragged_batch_and_sequence_shapes_vs_oracle(sed, B, T, C, H):
    """smallest legal sequence (T = 8 -> one GRU step), batch 1, batch sizes that do not fill a GRU batch tile or a conv
    time tile, odd numbers of conv tiles"""
    from oracle import crnn_ref
    torch.manual_seed(B * 100 + T)
    ref = crnn_ref.SedNetRef(conv_channels=C, dropout=0.0, gru_hidden=H)
    m = sed.TimePooledCRNN(conv_channels=C, dropout=0.0, gru_hidden=H)
    x, y = crnn_ref.synthetic_batch(B, 1, 40, T, T // 8, seed=B + T)
    if B * T * 40 < 4096:             # tiny batches make BatchNorm's 1/sigma large: keep the comparison well conditioned
        x = x * 3.0
    _oracle_vs_hip(sed, ref, m, x, y)


def test_multiclass_head_and_focal_sum(sed):
    """K = 6 classes (README figure) through the time-pooled topology; label tensor [B,T',6]"""
    from oracle import crnn_ref
    torch.manual_seed(21)
    ref = crnn_ref.SedNetRef(conv_channels=16, dropout=0.0, gru_hidden=16, n_classes=6)
    m = sed.TimePooledCRNN(conv_channels=16, dropout=0.0, gru_hidden=16, n_classes=6)
    x, y = crnn_ref.synthetic_batch(4, 1, 40, 32, 4, K=6, seed=3)
    _oracle_vs_hip(sed, ref, m, x, y)


def test_lightning_variant_vs_oracle_focal(sed):
    from oracle import crnn_ref
    torch.manual_seed(8)
    ref = crnn_ref.LightningNetRef(dropout=0.0)
    m = sed.LightningTimePooledCRNN(dropout=0.0)
    x, y = crnn_ref.synthetic_batch(5, 1, 40, 64, 8, seed=6)
    _oracle_vs_hip(sed, ref, m, x, y, loss="focal")


@pytest.mark.parametrize("cin", [1, 2])
def test_get_model_figure_topology_vs_torch(sed, cin):
    """README-figure SEDnet: mel pooling 5/2/2, no time pooling, 6 classes, dense 16 -> 6 (parity unpinned by the
    reference: there is no code for it; checked against autograd of the same graph assembled from torch.nn.functional):
    train-mode logits, loss and the gradient of EVERY parameter (frequency pooling backward, two-dense head, stacked GRUs
    fed by a [C, F'] = [32, 2] feature map), mono and binaural input."""
    import torch.nn as nn
    import torch.nn.functional as F
    torch.manual_seed(9 + cin)
    pools = [(5, 1), (2, 1), (2, 1)]
    m = sed.get_model(in_channels=cin, n_mels=40, seq_len=16, n_classes=6, conv_channels=32,
                      pools=pools, rnn_hidden=[32, 32], fc=[16, 6], dropout=0.0)
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in m.state_dict().items()}
    m.cuda()
    x = torch.randn(3, cin, 40, 16)
    y = (torch.rand(3, 16, 6) > 0.7).float()
    grus = []
    for i in range(2):
        g = nn.GRU(64 if i == 0 else 64, 32, batch_first=True, bidirectional=True)
        g.load_state_dict({k.split(".", 2)[2]: v.detach() for k, v in sd.items() if k.startswith(f"grus.{i}.")})
        grus.append(g)

    def ref_fwd(x):
        h = x
        for l, (pf, pt) in enumerate(pools):
            h = F.conv2d(h, sd[f"convs.{l}.weight"], sd[f"convs.{l}.bias"], padding=1)
            h = F.batch_norm(h, None, None, sd[f"bns.{l}.weight"], sd[f"bns.{l}.bias"], training=True)
            h = F.max_pool2d(torch.relu(h), (pf, pt))
        b, c, f, t = h.shape
        h = h.permute(0, 3, 1, 2).reshape(b, t, c * f)
        for g in grus:
            h, _ = g(h)
        h = torch.relu(F.linear(h, sd["fcs.0.weight"], sd["fcs.0.bias"]))
        return F.linear(h, sd["fcs.1.weight"], sd["fcs.1.bias"])
    out_r = ref_fwd(x)
    loss_r = F.binary_cross_entropy_with_logits(out_r, y)
    loss_r.backward()
    m.train()
    out = m(x.cuda())
    assert out.shape == (3, 16, 6)
    loss = sed.BCEWithLogitsLoss()(out, y.cuda())
    loss.backward()
    _cmp(out, out_r, atol=2e-4)
    assert abs(loss.item() - loss_r.item()) < 1e-5
    for k, p in m.named_parameters():
        if k.startswith("grus."):
            i, name = int(k.split(".")[1]), k.split(".", 2)[2]
            want = dict(grus[i].named_parameters())[name].grad
        else:
            want = sd[k].grad
        _cmp(p.grad, want, atol=1e-4, rtol=1e-2, msg=k)


def test_dropout_training_forward_is_seeded_and_unbiased(sed):
    torch.manual_seed(11)
    m = sed.TimePooledCRNN(conv_channels=16, dropout=0.5).cuda()
    x = torch.randn(4, 1, 40, 64).cuda()
    m.train()
    a = m(x).detach()
    b = m(x).detach()
    assert not torch.equal(a, b)                      # a fresh mask per step
    m.eval()
    e1, e2 = m(x), m(x)
    assert torch.equal(e1, e2)                        # eval is deterministic


def test_state_dict_roundtrip_and_checkpoint_format(sed, tmp_path):
    torch.manual_seed(12)
    m = sed.TimePooledCRNN(conv_channels=8, dropout=0.0).cuda()
    x = torch.randn(2, 1, 40, 64).cuda()
    m.eval()
    ref_out = m(x)
    path = tmp_path / "best_fold1.pt"
    torch.save(m.state_dict(), path)                  # bare state_dict like reference sed.py:198-199
    sd = torch.load(path, weights_only=True)
    assert list(sd.keys())[:4] == ["convs.0.weight", "convs.0.bias", "convs.1.weight", "convs.1.bias"]
    assert sd["convs.1.weight"].shape == (8, 8, 3, 3) and sd["gru.weight_ih_l0"].shape == (96, 320)
    m2 = sed.TimePooledCRNN(conv_channels=8, dropout=0.0)
    m2.load_state_dict(sd)
    m2.cuda().eval()
    assert torch.equal(m2(x), ref_out)


def test_backward_after_second_forward_raises(sed):
    m = sed.TimePooledCRNN(conv_channels=8, dropout=0.0).cuda().train()
    x = torch.randn(2, 1, 40, 64).cuda()
    out1 = m(x)
    m(x)
    with pytest.raises(RuntimeError, match="overwritten"):
        out1.sum().backward()


def test_bad_inputs_raise(sed):
    m = sed.TimePooledCRNN(conv_channels=8, dropout=0.0).cuda()
    with pytest.raises(ValueError):
        m(torch.zeros(2, 1, 40, 7).cuda())             # shorter than one output frame (T' = floor(7/8) = 0)
    with pytest.raises(ValueError):
        m(torch.zeros(2, 2, 40, 64).cuda())            # wrong channel count
    with pytest.raises(sed.SedHipError):
        m(torch.zeros(2, 1, 40, 64))                   # CPU tensor: no fallback


# ───────────── BASELINE-size properties (config 2: B=128, 1 ch, 40 mel, T=256, C=128, BiGRU 2x128) ─────────────
@pytest.fixture(scope="module")
def cfg2_model(sed):
    torch.manual_seed(0)
    m = sed.TimePooledCRNN(conv_channels=128, dropout=0.5, gru_hidden=128).cuda()
    return m


def test_cfg2_eval_matches_oracle_on_a_slice_and_is_batch_separable(sed, cfg2_model):
    from oracle import crnn_ref
    m = cfg2_model
    x, _ = crnn_ref.synthetic_batch(128, 1, 40, 256, 32, seed=1234)
    m.eval()
    with torch.no_grad():
        full = m(x.cuda())
        # eval mode has no cross-sample coupling: any sub-batch reproduces its rows (to rounding: a 4-sample batch takes
        # the split-K input projection, which sums K in a different order than the full batch's single pass)
        part = m(x[40:44].cuda())
        part32 = m(x[32:64].cuda())                 # same GEMM path as the full batch: bit-identical
    assert full.shape == (128, 32, 1)
    assert torch.equal(full[32:64], part32)
    _cmp(full[40:44], part, atol=2e-6)
    ref = crnn_ref.SedNetRef(conv_channels=128, dropout=0.5, gru_hidden=128)
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    ref.eval()
    with torch.no_grad():
        r = ref(x[40:44])
    _cmp(torch.sigmoid(part), torch.sigmoid(r), atol=1e-3)


def test_cfg2_train_step_is_deterministic_and_descends(sed, cfg2_model):
    from oracle import crnn_ref
    m = cfg2_model
    x, y = crnn_ref.synthetic_batch(128, 1, 40, 256, 32, seed=99)
    x, y = x.cuda(), y.cuda()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    crit = sed.BCEWithLogitsLoss()

    def three_steps():
        m.load_state_dict(sd0)
        m._seed_counter = 0
        opt = sed.FusedAdam(m.parameters(), lr=1e-3)
        m.train()
        ls = []
        for _ in range(3):
            opt.zero_grad()
            loss = crit(m(x), y)
            loss.backward()
            opt.step()
            ls.append(loss.item())
        return ls, m.flat_parameters().clone()
    l1, p1 = three_steps()
    l2, p2 = three_steps()
    assert l1 == l2 and torch.equal(p1, p2)           # bitwise reproducible (fixed-order reductions, seeded dropout)
    assert l1[2] < l1[0]
    assert torch.isfinite(p1).all()


def test_graph_captured_step_equals_eager_and_draws_fresh_dropout(sed):
    """FusedTrainStep(graph=True): one hipGraph per input shape, device-side dropout salt + optimiser step"""
    from oracle import crnn_ref
    from sed_crnn_amd.trainer import FusedTrainStep
    x, y = crnn_ref.synthetic_batch(8, 1, 40, 64, 8, seed=41)
    x, y = x.cuda(), y.cuda()

    def run(graph, dropout, n):
        torch.manual_seed(17)
        m = sed.TimePooledCRNN(conv_channels=16, dropout=dropout, gru_hidden=16).cuda()
        st = FusedTrainStep(m, lr=1e-3, graph=graph, clip_norm=1.0)
        losses = [st.step(x, y)[0].item() for _ in range(n)]
        return losses, m.flat_parameters().clone()
    le, pe = run(False, 0.0, 6)
    lg, pg = run(True, 0.0, 6)                     # 1 eager step, capture, 5 replays: every call is one real step
    np.testing.assert_allclose(lg, le, rtol=1e-5, atol=1e-6)
    sig = (pe - pg).abs() < 5e-5                   # Adam(+-lr) noise entries aside (conv biases before BN)
    assert sig.float().mean() > 0.995
    ld, _ = run(True, 0.5, 8)
    assert len(set(round(v, 6) for v in ld[3:])) > 1 and all(np.isfinite(ld))     # replays see different masks
    # the salt really changes the mask: probabilities of two replays of the same weights/inputs differ under dropout
    torch.manual_seed(3)
    m = sed.TimePooledCRNN(conv_channels=16, dropout=0.5, gru_hidden=16).cuda()
    st = FusedTrainStep(m, lr=0.0, graph=True)
    pr = [st.step(x, y)[1].clone() for _ in range(5)]
    assert not torch.equal(pr[3], pr[4])


def test_dropout_active_parity_with_injected_masks(sed):
    """Training-mode parity WITH dropout (p = 0.5 after every block, sed.py:92,107): the counter-hash masks of the HIP
    run are regenerated with the stand-alone kernel and injected into the torch oracle, so logits and every gradient
    can be compared although the two RNGs differ."""
    import torch.nn.functional as F
    from oracle import crnn_ref
    from sed_crnn_amd import ops
    torch.manual_seed(23)
    C, H, p = 16, 16, 0.5
    ref = crnn_ref.SedNetRef(conv_channels=C, dropout=0.0, gru_hidden=H)          # dropout applied by hand below
    m = sed.TimePooledCRNN(conv_channels=C, dropout=p, gru_hidden=H)
    m.load_state_dict(ref.state_dict())
    m.cuda().train()
    x, y = crnn_ref.synthetic_batch(4, 1, 40, 64, 8, seed=9)
    out = m(x.cuda())
    loss = sed.BCEWithLogitsLoss()(out, y.cuda())
    loss.backward()
    # regenerate the masks: block l uses seed + golden*(l+1) on the channels-last pooled index
    golden, mask64 = 0x9E3779B97F4A7C15, (1 << 64) - 1
    masks, T = [], 64
    for l in range(3):
        ones = torch.ones(4, T, 40, C).cuda()
        mk = ops.bn_relu_pool_drop_fwd(ones, torch.ones(C).cuda(), torch.zeros(C).cuda(), 1, 2, drop_p=p,
                                       seed=(m._seed + golden * (l + 1)) & mask64)          # [B,T/2,F,C] of {0, 2}
        masks.append(mk.permute(0, 3, 2, 1).cpu())                                          # -> NCHW [B,C,F,T/2]
        T //= 2
        assert abs((mk > 0).float().mean().item() - 0.5) < 0.02
    ref.train()
    h = x
    for l in range(3):
        h = F.max_pool2d(torch.relu(ref.bns[l](ref.convs[l](h))), (1, 2)) * masks[l]
    b, c, f, t = h.shape
    h, _ = ref.gru(h.permute(0, 3, 1, 2).reshape(b, t, c * f))
    out_r = ref.fc(h)
    loss_r = crnn_ref.bce_logits(out_r, y)
    loss_r.backward()
    _cmp(out, out_r, atol=2e-4)
    assert abs(loss.item() - loss_r.item()) < 1e-5
    rg = dict(ref.named_parameters())
    for k, q in m.named_parameters():
        _cmp(q.grad, rg[k].grad, atol=1e-4, rtol=1e-2, msg=k)


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("cin,C", [(1, 32), (4, 32)])
def test_staged_backward_completes_each_bucket_slice_when_its_stage_returns(sed, overlap, cin, C):
    """the contract the gradient all-reduce relies on: after stage s of sed_net_backward has been waited for on the main
    stream, the arena slice bucket_slices()[s] already holds its final values (with and without the auxiliary stream,
    fused and stored first block); together the slices cover the whole arena"""
    from oracle import crnn_ref
    from sed_crnn_amd import ops
    torch.manual_seed(13)
    m = sed.TimePooledCRNN(conv_channels=C, dropout=0.5, in_channels=cin, gru_hidden=16).cuda()
    m.overlap_wgrad = overlap
    x, y = crnn_ref.synthetic_batch(6, cin, 40, 32, 4, seed=4)
    x, y = x.cuda(), y.cuda()
    m.train()
    logits = m._run_forward(x, training=True)
    _, dlogits, _ = ops.loss_fwd_bwd(logits, y, "bce", 0.25, 2.0, "mean")
    m._run_backward(x, dlogits)                               # reference: the whole backward at once
    torch.cuda.synchronize()
    want = m.flat_grads().clone()
    slices = m.bucket_slices()
    assert slices[0][0] == 0 and slices[-1][1] == want.numel()
    assert all(a <= b for a, b in slices) and all(slices[i][1] == slices[i + 1][0] or slices[i + 1][0] == slices[i + 1][1]
                                                  for i in range(len(slices) - 1))
    valid = torch.zeros(want.numel(), dtype=torch.bool, device="cuda")      # the arena pads every tensor to 4 floats
    for p_, o in zip(m._arena_params, m._arena_offsets):
        valid[o:o + p_.numel()] = True
    m.flat_grads().fill_(float("nan"))
    for s, (a, b) in enumerate(slices):
        m._run_backward(x, dlogits, s, s + 1)
        torch.cuda.current_stream().synchronize()             # what an all-reduce enqueued on this stream would see
        got = m.flat_grads()[a:b].clone()
        assert torch.equal(got[valid[a:b]], want[a:b][valid[a:b]]), f"stage {s}: slice [{a},{b}) incomplete or different"
    torch.cuda.synchronize()
    assert torch.equal(m.flat_grads()[valid], want[valid])


@pytest.fixture(scope="module")
def forty_steps(sed):
    """The 40-step Adam run of a learnable synthetic task (labels = a threshold on a band of the input; 4 batches of 16
    sequences, lr 2e-3) done three times from the same state: the HIP fused trainer, the fp32 oracle's fit_step, and the
    oracle in FLOAT64 (the yardstick).  Shared by the two trajectory tests below."""
    from oracle import crnn_ref
    from sed_crnn_amd.trainer import FusedTrainStep
    torch.manual_seed(99)
    kw = dict(conv_channels=16, dropout=0.0, gru_hidden=16)
    ref32 = crnn_ref.SedNetRef(**kw)
    ref64 = crnn_ref.SedNetRef(**kw).double()
    ref64.load_state_dict({k: v.double() if v.dtype.is_floating_point else v for k, v in ref32.state_dict().items()})
    m = sed.TimePooledCRNN(**kw)
    m.load_state_dict(ref32.state_dict())
    m.cuda()
    g = torch.Generator().manual_seed(5)
    batches = []
    for _ in range(4):
        x = torch.randn(16, 1, 40, 64, generator=g)
        band = x[:, 0, 8:16, :].mean(1)                                    # [B,T]
        y = (band.reshape(16, 8, 8).amax(2) > 0.45).float().unsqueeze(-1)  # [B,T',1]: learnable, ~55 % positive
        batches.append((x, y))
    o32 = torch.optim.Adam(ref32.parameters(), lr=2e-3)
    o64 = torch.optim.Adam(ref64.parameters(), lr=2e-3)
    step = FusedTrainStep(m, lr=2e-3, loss="bce")
    l32, l64, lh = [], [], []
    for it in range(40):
        x, y = batches[it % 4]
        l32.append(float(crnn_ref.fit_step(ref32, o32, x, y)[0]))
        l64.append(float(crnn_ref.fit_step(ref64, o64, x.double(), y.double())[0]))
        lh.append(step.step(x.cuda(), y.cuda())[0])
    lh = torch.stack([l.reshape(()) for l in lh]).cpu().numpy().astype(np.float64)
    xs = torch.cat([b[0] for b in batches])
    ys = torch.cat([b[1] for b in batches]).numpy()
    for net in (ref32, ref64, m):
        net.eval()
    with torch.no_grad():
        p32 = torch.sigmoid(ref32(xs)).double().numpy()
        p64 = torch.sigmoid(ref64(xs.double())).numpy()
        ph = torch.sigmoid(m(xs.cuda())).cpu().double().numpy()
    return dict(kw=kw, ref32=ref32, m=m, xs=xs, ys=ys, l32=np.asarray(l32), l64=np.asarray(l64), lh=lh, p32=p32, p64=p64, ph=ph)


def test_forty_step_trajectory_tracks_the_oracle_and_scores_agree(sed, forty_steps):
    """40 Adam steps with the fused trainer vs the CPU oracle's fit_step (beyond the 6 steps of golden g3).
    * inference on the TRAINED weights is exact: the oracle's final state loaded into the HIP model gives its frame-wise
      probabilities within 1e-5 and identical ER / F1 at 1 s;
    * the two fp32 trajectories themselves: Adam turns rounding noise into +-lr steps on every coordinate whose gradient is
      near zero (the conv biases in front of BatchNorm are the extreme case), so two fp32 implementations of this loop
      drift apart at ~lr per step and NO fixed bound on their distance means anything (round 2 had literal bounds here
      and widened them when a harmless change of summation order tripped them).  The bound is therefore DERIVED from the
      float64 run: the yardstick test below asserts d(HIP, f64) <= 3 d(torch32, f64) + floor, hence by the triangle
      inequality d(HIP, torch32) <= 4 d(torch32, f64) + floor — asserted here with exactly those constants, nothing tuned."""
    f = forty_steps
    assert f["l32"][-1] < 0.8 * f["l32"][0]                                # it actually learns
    assert f["lh"][-1] < 0.8 * f["lh"][0]
    m2 = sed.TimePooledCRNN(**f["kw"])
    m2.load_state_dict(f["ref32"].state_dict())
    m2.cuda().eval()
    with torch.no_grad():
        p2 = torch.sigmoid(m2(f["xs"].cuda())).cpu().numpy()
    np.testing.assert_allclose(p2, f["p32"], atol=1e-5)
    assert sed.metrics.compute_scores(p2 > 0.5, f["ys"], 5) == sed.metrics.compute_scores(f["p32"] > 0.5, f["ys"], 5)
    e_loss_t = np.abs(f["l32"] - f["l64"]).max()
    e_p_t = np.abs(f["p32"] - f["p64"])
    d_loss = np.abs(f["lh"] - f["l32"]).max()
    d_p = np.abs(f["ph"] - f["p32"])
    print(f"40-step trajectory, HIP vs torch-f32: loss {d_loss:.2e} (allowed {4 * e_loss_t + 1e-4:.2e}), probabilities mean "
          f"{d_p.mean():.2e} (allowed {4 * e_p_t.mean() + 1e-4:.2e}), max {d_p.max():.2e} (allowed {4 * e_p_t.max() + 1e-3:.2e})")
    assert d_loss <= 4.0 * e_loss_t + 1e-4
    assert d_p.mean() <= 4.0 * e_p_t.mean() + 1e-4
    assert d_p.max() <= 4.0 * e_p_t.max() + 1e-3
    # decisions: wherever the float64 run decides by more than the allowed distance, HIP decides the same
    sure = np.abs(f["p64"] - 0.5) > 3.0 * e_p_t.max() + 1e-3
    assert np.array_equal((f["ph"] > 0.5)[sure], (f["p64"] > 0.5)[sure])


def test_long_recording_single_sequence_inference_matches_oracle(sed):
    """streaming-style inference: one 2048-frame sequence (B = 1, 256 GRU steps) through the full-width net in eval mode"""
    from oracle import crnn_ref
    torch.manual_seed(17)
    kw = dict(conv_channels=128, dropout=0.5, gru_hidden=32)
    ref = crnn_ref.SedNetRef(**kw).eval()
    m = sed.TimePooledCRNN(**kw)
    m.load_state_dict(ref.state_dict())
    m.cuda().eval()
    x = torch.randn(1, 1, 40, 2048, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        pr = torch.sigmoid(ref(x))
        ph = torch.sigmoid(m(x.cuda())).cpu()
    assert ph.shape == (1, 256, 1)
    _cmp(ph, pr, atol=1e-4)


# ───────────── BASELINE configs 3 and 5 at their full per-GPU size (properties the domain offers at any size) ─────────────
@pytest.mark.parametrize("name,cfg", [("config3", dict(B=128, Cin=2, F=40, T=256, C=128, H=128)),
                                      ("config5", dict(B=128, Cin=4, F=128, T=512, C=128, H=256))])
def test_full_size_multichannel_configs_finite_separable_deterministic(sed, name, cfg):
    """binaural (B=128,256,40,2) and 4-channel / 128-mel / T=512 / BiGRU 2x256 at B=128 per GPU:
    finite loss and gradients after real fit steps, eval forward of the whole batch == eval forward of its chunks
    bit-for-bit (no cross-sample coupling in eval), and two runs from the same state are bitwise identical."""
    from sed_crnn_amd.trainer import FusedTrainStep
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(cfg["B"], cfg["Cin"], cfg["F"], cfg["T"], generator=g).cuda()
    y = (torch.rand(cfg["B"], cfg["T"] // 8, 1, generator=g) > 0.8).float().cuda()
    m = sed.TimePooledCRNN(conv_channels=cfg["C"], dropout=0.5, in_channels=cfg["Cin"], n_mels=cfg["F"], gru_hidden=cfg["H"]).cuda()
    p0 = m.flat_parameters().clone()
    bufs0 = [b.clone() for b in m.buffers()]

    def two_steps():
        with torch.no_grad():
            m.flat_parameters().copy_(p0)
            for b, b0 in zip(m.buffers(), bufs0):
                b.copy_(b0)
        m._seed_counter = 0
        st = FusedTrainStep(m, lr=1e-3, loss="bce")
        losses = [st.step(x, y)[0].item() for _ in range(2)]
        return losses, m.flat_parameters().clone(), m.flat_grads().clone()
    l1, p1, g1 = two_steps()
    l2, p2, g2 = two_steps()
    assert all(np.isfinite(l1)) and torch.isfinite(p1).all() and torch.isfinite(g1).all()
    assert l1 == l2 and torch.equal(p1, p2) and torch.equal(g1, g2)
    assert float(g1.abs().max()) > 0
    m.eval()
    with torch.no_grad():
        whole = m(x)
        parts = torch.cat([m(x[i:i + 32]) for i in range(0, cfg["B"], 32)])
    assert whole.shape == (cfg["B"], cfg["T"] // 8, 1) and torch.isfinite(whole).all()
    assert torch.equal(whole, parts)
    del m, x, y, p0, p1, p2, g1, g2
    torch.cuda.empty_cache()


def test_forty_step_trajectory_is_as_close_to_float64_as_torch_float32_is(sed, forty_steps):
    """Separates "two fp32 trajectories drift apart" from a real defect (round-1 verdict).  The FLOAT64 run of the same 40
    steps (the oracle net cast to double) is the yardstick.  Freezing the zero-gradient conv biases alone does not stop the
    drift (measured: weights still 2e-3 apart, max |dp| 3e-2), so the drift is Adam's amplification of rounding noise on
    EVERY coordinate with a small gradient, and the question becomes whether the HIP path is any further from the float64
    truth than torch's own float32 path is.  Asserted: per-step losses and final probabilities of the HIP run are within 3x
    torch-float32's own distance from float64 (plus a small floor), i.e. the HIP path is an fp32 implementation of the
    same computation, not a different one."""
    f = forty_steps
    e_loss_t, e_loss_h = np.abs(f["l32"] - f["l64"]).max(), np.abs(f["lh"] - f["l64"]).max()
    e_p_t, e_p_h = np.abs(f["p32"] - f["p64"]), np.abs(f["ph"] - f["p64"])
    print(f"distance from the float64 run after 40 Adam steps: loss torch-f32 {e_loss_t:.2e} / HIP {e_loss_h:.2e}; "
          f"probabilities max torch-f32 {e_p_t.max():.2e} / HIP {e_p_h.max():.2e}, mean {e_p_t.mean():.2e} / {e_p_h.mean():.2e}")
    assert e_loss_h <= 3.0 * e_loss_t + 1e-4
    assert e_p_h.mean() <= 3.0 * e_p_t.mean() + 1e-4
    assert e_p_h.max() <= 3.0 * e_p_t.max() + 1e-3


def test_single_step_gradients_are_as_close_to_float64_as_torch_float32(sed):
    """No chaos in one step: the gradient of every parameter from the HIP path is compared with the FLOAT64 oracle, next to
    torch's own float32 gradient.  The relative L2 error of the HIP gradient must not exceed 3x torch-float32's (plus a
    floor of 2e-6): the first block's statistics from input moments, the fp32-MFMA convolutions and the fixed-order
    reductions are at least as accurate as ATen's float32 kernels."""
    from oracle import crnn_ref
    torch.manual_seed(4)
    kw = dict(conv_channels=32, dropout=0.0, gru_hidden=32)
    ref32 = crnn_ref.SedNetRef(**kw)
    ref64 = crnn_ref.SedNetRef(**kw).double()
    ref64.load_state_dict({k: v.double() if v.dtype.is_floating_point else v for k, v in ref32.state_dict().items()})
    m = sed.TimePooledCRNN(**kw)
    m.load_state_dict(ref32.state_dict())
    m.cuda().train()
    x, y = crnn_ref.synthetic_batch(8, 1, 40, 64, 8, seed=11)
    x = x * 1.5 + 0.3                                   # not perfectly standardised: a mean the statistics must resolve
    for net, xx, yy in ((ref32, x, y), (ref64, x.double(), y.double())):
        net.train()
        crnn_ref.bce_logits(net(xx), yy).backward()
    sed.BCEWithLogitsLoss()(m(x.cuda()), y.cuda()).backward()
    g64 = {k: p.grad for k, p in ref64.named_parameters()}
    g32 = {k: p.grad.double() for k, p in ref32.named_parameters()}
    worst = 0.0
    for k, p in m.named_parameters():
        if k.startswith("convs.") and k.endswith(".bias"):
            continue                                    # analytically zero gradient: both sides hold rounding noise only
        gh = p.grad.cpu().double()
        den = g64[k].norm().item() + 1e-30
        e_h, e_t = (gh - g64[k]).norm().item() / den, (g32[k] - g64[k]).norm().item() / den
        worst = max(worst, e_h / (e_t + 1e-30))
        assert e_h <= 3.0 * e_t + 2e-6, (k, e_h, e_t)
    print(f"single-step gradient error vs float64: worst HIP/torch-f32 ratio over the parameters = {worst:.2f}")


@pytest.mark.parametrize("direct", [False, True])
def test_winograd_and_direct_conv_plans_both_match_golden_g5_and_the_oracle(sed, direct):
    """The 128-channel blocks run as Winograd F(2x2,3x3) by default (forward, data gradient, weight gradient: wino.hip / the
    Winograd form of conv3x3_mfma_wgrad2_k); `plan_flags = SED_NET_DIRECT_CONV` keeps the direct 36-product kernels.  Both plans
    against the SAME references with the SAME tolerances: the full-width golden g5 (K = 1152: logits eval / train and every
    gradient, sed.py:105-112,134-137) and a routed oracle comparison on a 2-channel C = 128 net; and the two plans' gradients
    agree with each other far inside those tolerances."""
    from oracle import crnn_ref
    d = load_golden("g5_sed_c128.npz")
    ref = crnn_ref.SedNetRef(conv_channels=128, dropout=0.0)
    m = sed.TimePooledCRNN(conv_channels=128, dropout=0.0)
    m.plan_flags = 0x4 if direct else 0
    m.load_state_dict(crnn_ref.rs_state_dict(ref, seed=int(d["weight_seed"])))
    m.cuda()
    x, y = torch.from_numpy(d["x"]).cuda(), torch.from_numpy(d["y"]).cuda()
    m.eval()
    _cmp(m(x), d["logits_eval"], atol=2e-4)
    m.train()
    out = m(x)
    sed.BCEWithLogitsLoss()(out, y).backward()
    _cmp(out, d["logits_train"], atol=2e-4)
    named = dict(m.named_parameters())
    for k in d:
        if k.startswith("grad.") and k[5:] in named:
            _cmp(named[k[5:]].grad, d[k], atol=5e-5, rtol=5e-3, msg=k)
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    # the other plan on the same weights and batch
    m2 = sed.TimePooledCRNN(conv_channels=128, dropout=0.0)
    m2.plan_flags = 0 if direct else 0x4
    m2.load_state_dict(crnn_ref.rs_state_dict(ref, seed=int(d["weight_seed"])))
    m2.cuda().train()
    sed.BCEWithLogitsLoss()(m2(x), y).backward()
    for k, p in m2.named_parameters():
        num = float((p.grad - grads[k]).norm()), float(grads[k].norm())
        # (conv biases in front of a BatchNorm have a zero gradient up to rounding: ~1e-8 of noise on both sides)
        assert num[0] <= 2e-5 * num[1] + 1e-7, (k, num)
    torch.manual_seed(7)
    ref2 = crnn_ref.SedNetRef(conv_channels=128, dropout=0.0, in_channels=2, n_mels=40, gru_hidden=128)
    m3 = sed.TimePooledCRNN(conv_channels=128, dropout=0.0, in_channels=2, n_mels=40, gru_hidden=128)
    m3.plan_flags = 0x4 if direct else 0
    x2, y2 = crnn_ref.synthetic_batch(3, 2, 40, 32, 4, seed=5)
    _oracle_vs_hip(sed, ref2, m3, x2, y2)


def test_bf16x3_conv_experiment_stays_within_the_parity_tolerances(sed):
    """model.set_conv_precision("bf16x3") (explicit opt-in, not the default): the full-width golden g5 (K = 1152) and a
    C = 128 oracle comparison with the SAME tolerances as the exact-fp32 path; and the default is unchanged."""
    from oracle import crnn_ref
    d = load_golden("g5_sed_c128.npz")
    ref = crnn_ref.SedNetRef(conv_channels=128, dropout=0.0)
    m = sed.TimePooledCRNN(conv_channels=128, dropout=0.0)
    assert m._cfg(1, 8).conv_mode == 0                                 # exact fp32 unless asked
    m.load_state_dict(crnn_ref.rs_state_dict(ref, seed=int(d["weight_seed"])))
    m.cuda().set_conv_precision("bf16x3")
    assert m._cfg(1, 8).conv_mode == 1
    x, y = torch.from_numpy(d["x"]).cuda(), torch.from_numpy(d["y"]).cuda()
    m.eval()
    _cmp(m(x), d["logits_eval"], atol=2e-4)
    m.train()
    out = m(x)
    sed.BCEWithLogitsLoss()(out, y).backward()
    _cmp(out, d["logits_train"], atol=2e-4)
    named = dict(m.named_parameters())
    for k in d:
        if k.startswith("grad.") and k[5:] in named:
            _cmp(named[k[5:]].grad, d[k], atol=5e-5, rtol=5e-3, msg=k)
    torch.manual_seed(7)
    ref2 = crnn_ref.SedNetRef(conv_channels=128, dropout=0.0, in_channels=2, n_mels=40, gru_hidden=128)
    m2 = sed.TimePooledCRNN(conv_channels=128, dropout=0.0, in_channels=2, n_mels=40, gru_hidden=128).set_conv_precision("bf16x3")
    x2, y2 = crnn_ref.synthetic_batch(3, 2, 40, 32, 4, seed=5)
    _oracle_vs_hip(sed, ref2, m2, x2, y2)
