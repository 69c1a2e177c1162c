"""The CPU oracle (oracle/) against the golden vectors captured from the imported
reference (oracle/make_goldens.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import crnn_ref, metrics_ref


def _sd(d, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith(prefix)}


def test_g1_sed_forward_and_grads():
    d = load_golden("g1_sed_c8.npz")
    m = crnn_ref.SedNetRef(conv_channels=8, dropout=0.0)
    m.load_state_dict(_sd(d))                      # identical keys/shapes as the reference
    x, y = torch.from_numpy(d["x"]), torch.from_numpy(d["y"])
    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(m(x).numpy(), d["logits_eval"], atol=1e-5, rtol=1e-5)
    m.train()
    out = m(x)
    loss = crnn_ref.bce_logits(out, y)
    loss.backward()
    np.testing.assert_allclose(out.detach().numpy(), d["logits_train"], atol=1e-5, rtol=1e-5)
    assert abs(loss.item() - float(d["loss_train"])) < 1e-6
    for k, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), d["grad." + k], atol=2e-6, rtol=1e-4, err_msg=k)
    for k, v in m.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.numpy(), d["after." + k], atol=1e-6, rtol=1e-5, err_msg=k)


def test_g3_adam_trajectory():
    d = load_golden("g3_sed_c8_traj.npz")
    m = crnn_ref.SedNetRef(conv_channels=8, dropout=0.0)
    m.load_state_dict(_sd(d, "sd0."))
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    batches = [(torch.from_numpy(d[f"x{i}"]), torch.from_numpy(d[f"y{i}"])) for i in range(3)]
    losses = []
    for ep in range(2):
        tot = 0.0
        for xb, yb in batches:
            l, _ = crnn_ref.fit_step(m, opt, xb, yb)
            tot += l.item()
        losses.append(tot / 3)
    np.testing.assert_allclose(losses, d["train_losses"], atol=1e-3)
    m.eval()
    with torch.no_grad():
        preds = np.concatenate([torch.sigmoid(m(xb)).numpy() for xb, _ in batches])
    np.testing.assert_allclose(preds, d["val_preds"], atol=1e-3)
    sc = metrics_ref.compute_scores(preds > 0.5, d["val_labels"], 5)
    assert sc["f1_overall_1sec"] == pytest.approx(float(d["val_f1_1s"]), abs=1e-12)
    assert sc["er_overall_1sec"] == pytest.approx(float(d["val_er_1s"]), abs=1e-12)


def test_g4_lightning_net_and_focal():
    d = load_golden("g4_lightning.npz")
    m = crnn_ref.LightningNetRef(dropout=0.0)
    m.load_state_dict(_sd(d))
    x, y = torch.from_numpy(d["x"]), torch.from_numpy(d["y"])
    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(m(x).numpy(), d["logits_eval"], atol=1e-5, rtol=1e-5)
    m.train()
    loss = crnn_ref.focal_bce(m(x), y)
    loss.backward()
    assert abs(loss.item() - float(d["loss_train"])) < 1e-6
    for k, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), d["grad." + k], atol=2e-6, rtol=1e-4, err_msg=k)
    lg, tg = torch.from_numpy(d["focal_logits"]), torch.from_numpy(d["focal_targets"])
    assert crnn_ref.focal_bce(lg, tg).item() == pytest.approx(float(d["focal_mean"]), rel=1e-6)
    assert crnn_ref.focal_bce(lg, tg, reduction="sum").item() == pytest.approx(float(d["focal_sum"]), rel=1e-6)
    # Adam(lr=1e-3, weight_decay=1e-4) step (crnn_lightning.py:195-197)
    opt = torch.optim.Adam(m.parameters(), lr=float(d["opt_lr"]), weight_decay=float(d["opt_wd"]))
    opt.step()
    for k, v in m.state_dict().items():
        if v.dtype.is_floating_point:
            np.testing.assert_allclose(v.numpy(), d["sd1." + k], atol=1e-6, rtol=1e-5, err_msg=k)


def test_g5_full_width_from_seed():
    d = load_golden("g5_sed_c128.npz")
    m = crnn_ref.SedNetRef(conv_channels=128, dropout=0.0)
    m.load_state_dict(crnn_ref.rs_state_dict(m, seed=int(d["weight_seed"])))
    x, y = torch.from_numpy(d["x"]), torch.from_numpy(d["y"])
    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(m(x).numpy(), d["logits_eval"], atol=1e-5, rtol=1e-4)
    m.train()
    out = m(x)
    crnn_ref.bce_logits(out, y).backward()
    np.testing.assert_allclose(out.detach().numpy(), d["logits_train"], atol=1e-5, rtol=1e-4)
    named = dict(m.named_parameters())
    for k in d:
        if k.startswith("grad.") and k[5:] in named:
            np.testing.assert_allclose(named[k[5:]].grad.numpy(), d[k], atol=1e-5, rtol=1e-3, err_msg=k)


def test_g6_metrics_known_answers():
    d = load_golden("g6_metrics.npz")
    p, t = d["p"], d["t"]
    sc = metrics_ref.compute_scores(p > 0.5, t, 5)
    assert sc["f1_overall_1sec"] == float(d["f1_1s"]) == 0.8888888888888887
    assert sc["er_overall_1sec"] == float(d["er_1s"]) == 0.125
    assert metrics_ref.f1_framewise(p > 0.5, t) == float(d["f1_fr"]) == 0.34146341463414626
    assert metrics_ref.er_framewise(p > 0.5, t) == float(d["er_fr"]) == 1.588235294117647
    # uint8 inputs give the same answers
    assert metrics_ref.f1_1sec((p > 0.5).astype(np.uint8), t.astype(np.uint8), 5) == float(d["f1_1s"])
    p6, t6 = d["p6"], d["t6"]
    sc = metrics_ref.compute_scores(p6 > 0.5, t6, 4)
    assert sc["f1_overall_1sec"] == float(d["k6_f1_1s"])
    assert sc["er_overall_1sec"] == float(d["k6_er_1s"])
    assert metrics_ref.f1_framewise((p6 > 0.5).astype(np.uint8), t6.astype(np.uint8)) == float(d["k6_f1_fr"])
    assert metrics_ref.er_framewise((p6 > 0.5).astype(np.uint8), t6.astype(np.uint8)) == float(d["k6_er_fr"])
    z, o = np.zeros((4, 8, 1), np.float32), np.ones((4, 8, 1), np.float32)
    assert metrics_ref.f1_1sec(o, z, 5) == float(d["edge_nref0_nsys_f1"]) == 0.0
    assert np.isinf(metrics_ref.er_1sec(o, z, 5)) and np.isinf(d["edge_nref0_nsys_er"])
    assert np.isnan(metrics_ref.er_1sec(z, z, 5)) and np.isnan(d["edge_allzero_er"])
    assert metrics_ref.f1_1sec(z, z, 5) == float(d["edge_allzero_f1"])


def test_g7_g8_dataset_augment_and_packing_oracle():
    """oracle/data_ref.py against what the imported reference produced (HitWindowDataset items, _spec_augment under a
    seeded np.random, utils.split_in_seqs / split_multi_channels, clean negatives, label pooling)"""
    from oracle import data_ref as R
    d = load_golden("g8_window_aug.npz")
    for i in range(3):
        rs = np.random.RandomState(int(d[f"aug_seed{i}"]))
        t, f = R.draw_spec_masks(rs, 40, 64)
        np.testing.assert_array_equal(R.spec_augment(d[f"aug_in{i}"], t, f), d[f"aug_out{i}"])
    x, y = R.window_item(d["item_mel"], d["item_lab"], int(d["item_start"]), 64, 8)
    np.testing.assert_array_equal(x, d["item_x"])
    np.testing.assert_array_equal(y, d["item_y"])
    x, y = R.window_item(d["item_mel"], d["item_lab"], int(d["item_pos_start"]), 64, 8)
    np.testing.assert_array_equal(x, d["item_pos_x"])
    np.testing.assert_array_equal(y, d["item_pos_y"])
    assert y.max() == 1.0
    np.testing.assert_array_equal(R.split_in_seqs(d["pack_feat"], 16), d["pack_seqs"])
    np.testing.assert_array_equal(R.split_multi_channels(d["pack_seqs"], 2), d["pack_mc"])
    g7 = load_golden("g7_dataset.npz")
    np.testing.assert_array_equal(R.find_clean_negatives(g7["lab"], 64), g7["neg_starts"])
    np.testing.assert_array_equal(R.pool_labels(g7["lab"][60:124], 8), g7["pooled_60"])


def test_routed_oracle_equals_the_plain_oracle_under_its_own_decisions_and_audit_refuses_wrong_ones():
    """oracle.crnn_ref.forward_routed is how the GPU parity tests make the oracle route its gradients with the HIP plan's ReLU /
    arg-max decisions.  It must BE the reference's forward (sed.py:106-112, crnn_lightning.py:66-73): with the decisions torch
    itself takes (first maximum, gate = max > 0) logits and every gradient are bit-identical to the plain forward — both
    reference variants, ragged tails, pools of 1 / 2 / 4.  And `audit_routes` must refuse a decision that is not a tie."""
    import torch.nn.functional as F
    from oracle import crnn_ref
    torch.manual_seed(0)
    for Ref, kw in ((crnn_ref.SedNetRef, dict(conv_channels=8, dropout=0.0, n_mels=13, time_pool=(2, 1, 4), gru_hidden=8)),
                    (crnn_ref.LightningNetRef, dict(dropout=0.0, n_mels=10, time_pool=(2, 2, 2)))):
        m = Ref(**kw)
        m.train()
        x = torch.randn(3, 1, kw["n_mels"], 35)
        blocks, _, _ = crnn_ref._blocks(m)
        routes, h = [], x
        with torch.no_grad():
            for conv, bn, (pf, pt) in blocks:
                z = F.batch_norm(conv(h), None, None, bn.weight, bn.bias, training=True, eps=bn.eps)
                B, C, Fm, T = z.shape
                Fp, Tp = Fm // pf, T // pt
                w = z[:, :, :Fp * pf, :Tp * pt].reshape(B, C, Fp, pf, Tp, pt).permute(0, 1, 2, 4, 3, 5).reshape(B, C, Fp, Tp, pf * pt)
                zmax = w.max(-1).values
                first = (w == zmax.unsqueeze(-1)).float().argmax(-1)
                routes.append(torch.where(zmax > 0, first + 1, torch.zeros_like(first)).to(torch.uint8).permute(0, 3, 2, 1).contiguous())
                h = F.max_pool2d(torch.relu(z), (pf, pt))
        out0 = m(x)
        out0.square().sum().backward()
        g0 = {k: p.grad.clone() for k, p in m.named_parameters()}
        m.zero_grad()
        audit = []
        out1 = crnn_ref.forward_routed(m, x, routes, audit=audit)
        out1.square().sum().backward()
        assert torch.equal(out0, out1), Ref.__name__
        for k, p in m.named_parameters():
            assert torch.equal(g0[k], p.grad), (Ref.__name__, k)
        assert all(a == (0, 0, 0.0) for a in audit), audit
    # a decision that is NOT a tie is refused: route an open window to its smaller element, or close an open gate
    z = torch.tensor([1.0, 0.5, -1.0, -2.0]).reshape(1, 1, 1, 4)
    ok = torch.tensor([1, 0], dtype=torch.uint8).reshape(1, 2, 1, 1)
    assert crnn_ref.audit_routes(z, ok, 1, 2) == (0, 0, 0.0)
    for bad in ([2, 0], [0, 0], [1, 1]):
        with pytest.raises(AssertionError):
            crnn_ref.audit_routes(z, torch.tensor(bad, dtype=torch.uint8).reshape(1, 2, 1, 1), 1, 2)
    # ... while a genuine tie may go either way
    zt = torch.tensor([1.0, 1.0 - 1e-6, 5e-6, -1.0]).reshape(1, 1, 1, 4)
    assert crnn_ref.audit_routes(zt, torch.tensor([2, 0], dtype=torch.uint8).reshape(1, 2, 1, 1), 1, 2)[:2] == (1, 1)
