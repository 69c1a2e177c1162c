#!/usr/bin/env python3
"""Capture golden vectors from the IMPORTED reference.  TEST INFRASTRUCTURE.

Run in the build container only (``/root/reference`` does not exist on the GPU
box):  ``python oracle/make_goldens.py``  ->  tests/golden/*.npz

Only numbers are written (inputs, weights, expected outputs); no reference
source travels.  pytorch_lightning is absent from the image (an ordinary
ModuleNotFoundError, not a denial), so crnn_lightning.py is imported behind a
minimal in-memory stand-in that lives in this script only.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle.crnn_ref import rs_state_dict  # noqa: E402

REF = os.environ.get("SED_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _import_reference():
    os.environ["HOME"] = tempfile.mkdtemp(prefix="sedref_home_")   # sed.py:39-41 makedirs under ~
    sys.path.insert(0, REF)
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        current_epoch = 0

        def save_hyperparameters(self, ignore=()):
            import inspect
            frame = inspect.currentframe().f_back
            args = {k: v for k, v in frame.f_locals.items()
                    if k not in ("self", "__class__") and k not in ignore}
            self.hparams = types.SimpleNamespace(**args)

        def log(self, *a, **k):
            self._logged = getattr(self, "_logged", {})
            self._logged[a[0]] = a[1]

    class LightningDataModule:
        pass

    pl.LightningModule = LightningModule
    pl.LightningDataModule = LightningDataModule
    sys.modules["pytorch_lightning"] = pl
    import sed, metrics, utils, crnn_lightning, decorte_datamodule          # noqa: E401
    return sed, metrics, utils, crnn_lightning, decorte_datamodule


def sd_np(model, prefix="sd."):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def golden_scaler():
    """G9: sklearn.preprocessing.StandardScaler exactly as feature.py:127-129 calls it (fit_transform on the train
    split, transform on the test split), on the seeded matrices of oracle.logmel_ref.scaler_fixture_inputs.  Only the
    seed and sklearn's outputs are stored; scikit-learn is importable in the build container (not on the GPU box)."""
    import sklearn
    from sklearn import preprocessing
    from oracle.logmel_ref import scaler_fixture_inputs
    seed = 77
    xtr, xte = scaler_fixture_inputs(seed)
    scaler = preprocessing.StandardScaler()
    ttr = scaler.fit_transform(xtr.copy())
    tte = scaler.transform(xte.copy())
    d = {"seed": np.int64(seed), "mean_": scaler.mean_, "var_": scaler.var_, "scale_": scaler.scale_,
         "n_samples_seen_": np.int64(scaler.n_samples_seen_), "train_t": ttr, "test_t": tte,
         "sklearn_version": np.asarray(sklearn.__version__)}
    assert ttr.dtype == np.float32 and scaler.mean_.dtype == np.float64
    np.savez_compressed(os.path.join(OUT, "g9_scaler.npz"), **d)
    print("g9_scaler.npz: scale_[3:8] =", scaler.scale_[3:8])


def main():
    if sys.argv[1:] == ["g9"]:               # the scaler golden needs scikit-learn only, not the reference checkout
        os.makedirs(OUT, exist_ok=True)
        return golden_scaler()
    torch.set_num_threads(4)
    torch.use_deterministic_algorithms(False)
    os.makedirs(OUT, exist_ok=True)
    sed, metrics, utils, cl, ddm = _import_reference()
    bce = torch.nn.BCEWithLogitsLoss()

    # ── G1/G2: small sed.py net (C=8), forward (eval + train) and BCE gradients ──
    torch.manual_seed(0)
    m = sed.TimePooledCRNN(conv_channels=8, dropout=0.0)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 1, 40, 64, generator=g)
    y = (torch.rand(4, 8, 1, generator=g) > 0.7).float()
    d = sd_np(m)
    d["x"], d["y"] = x.numpy(), y.numpy()
    m.eval()
    with torch.no_grad():
        d["logits_eval"] = m(x).numpy()
    m.train()
    out = m(x)
    loss = bce(out, y)
    loss.backward()
    d["logits_train"] = out.detach().numpy()
    d["loss_train"] = np.float32(loss.item())
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            d["after." + k] = v.numpy().copy()
    for k, p in m.named_parameters():
        d["grad." + k] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g1_sed_c8.npz"), **d)

    # ── G3: 6-step Adam trajectory through sed.run_epoch (3 batches x 2 epochs) ──
    torch.manual_seed(1)
    m = sed.TimePooledCRNN(conv_channels=8, dropout=0.0)
    d = sd_np(m, "sd0.")
    g = torch.Generator().manual_seed(12)
    batches = []
    for i in range(3):
        xb = torch.randn(4, 1, 40, 64, generator=g)
        yb = (torch.rand(4, 8, 1, generator=g) > 0.7).float()
        batches.append((xb, yb))
        d[f"x{i}"], d[f"y{i}"] = xb.numpy(), yb.numpy()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    losses = []
    for ep in range(2):
        l, preds, labels = sed.run_epoch(m, batches, bce, opt)
        losses.append(l)
    d["train_losses"] = np.asarray(losses, np.float64)
    d["train_preds_last"] = preds
    lv, pv, tv = sed.run_epoch(m, batches, bce)
    d["val_loss"], d["val_preds"], d["val_labels"] = np.float64(lv), pv, tv
    sc = metrics.compute_scores(pv > 0.5, tv, frames_in_1_sec=sed.FPS_OUT)
    d["val_f1_1s"], d["val_er_1s"] = np.float64(sc["f1_overall_1sec"]), np.float64(sc["er_overall_1sec"])
    d.update(sd_np(m, "sd6."))
    np.savez_compressed(os.path.join(OUT, "g3_sed_c8_traj.npz"), **d)

    # ── G4: Lightning net (defaults: C=16, GRU 16/8, dense 8) + focal loss ──
    torch.manual_seed(2)
    lm = cl.CRNNLightning(fold_id=1, art_dir=tempfile.mkdtemp(), dropout=0.0)
    g = torch.Generator().manual_seed(13)
    x = torch.randn(4, 1, 40, 64, generator=g)
    y = (torch.rand(4, 8, 1, generator=g) > 0.7).float()
    d = sd_np(lm.model)
    d["x"], d["y"] = x.numpy(), y.numpy()
    lm.eval()
    with torch.no_grad():
        d["logits_eval"] = lm(x).numpy()
    lm.train()
    loss = lm.training_step((x, y), 0)
    loss.backward()
    d["loss_train"] = np.float32(loss.item())
    d["preds_train"] = lm._buf["train"]["preds"][0].detach().numpy()
    for k, p in lm.model.named_parameters():
        d["grad." + k] = p.grad.numpy().copy()
    cfgd = lm.configure_optimizers()
    o = cfgd["optimizer"]
    d["opt_lr"] = np.float64(o.param_groups[0]["lr"])
    d["opt_wd"] = np.float64(o.param_groups[0]["weight_decay"])
    d["opt_betas"] = np.asarray(o.param_groups[0]["betas"], np.float64)
    d["opt_eps"] = np.float64(o.param_groups[0]["eps"])
    s = cfgd["lr_scheduler"]["scheduler"]
    d["sched_factor"], d["sched_patience"] = np.float64(s.factor), np.int64(s.patience)
    d["monitor"] = np.asarray(cfgd["lr_scheduler"]["monitor"])
    # focal loss on a hand grid, both reductions
    lg = torch.linspace(-6, 6, 25).reshape(5, 5, 1)
    tg = (torch.arange(25).reshape(5, 5, 1) % 3 == 0).float()
    d["focal_logits"], d["focal_targets"] = lg.numpy(), tg.numpy()
    d["focal_mean"] = np.float32(cl.FocalBCELoss()(lg, tg).item())
    d["focal_sum"] = np.float32(cl.FocalBCELoss(reduction="sum")(lg, tg).item())
    # one Adam(wd=1e-4) step on the lightning net
    o.step()
    d.update(sd_np(lm.model, "sd1."))
    # _aggregate dictionary on the collected buffers
    agg = lm._aggregate("train")
    d["agg_cm"] = agg["cm"]
    d["agg_vals"] = np.asarray([agg["loss"], agg["f1_frame"], agg["er_frame"], agg["f1_1s"], agg["er_1s"]], np.float64)
    np.savez_compressed(os.path.join(OUT, "g4_lightning.npz"), **d)

    # ── G5: full-width sed.py net (C=128, K=1152) at tiny B/T; weights from a seed ──
    torch.manual_seed(3)
    m = sed.TimePooledCRNN(conv_channels=128, dropout=0.0)
    m.load_state_dict(rs_state_dict(m, seed=2024))
    g = torch.Generator().manual_seed(14)
    x = torch.randn(2, 1, 40, 32, generator=g)
    y = (torch.rand(2, 4, 1, generator=g) > 0.7).float()
    d = {"weight_seed": np.int64(2024), "x": x.numpy(), "y": y.numpy()}
    m.eval()
    with torch.no_grad():
        d["logits_eval"] = m(x).numpy()
    m.train()
    out = m(x)
    loss = bce(out, y)
    loss.backward()
    d["logits_train"], d["loss_train"] = out.detach().numpy(), np.float32(loss.item())
    for k in ("convs.0.weight", "convs.1.weight", "convs.1.bias", "convs.2.weight", "bns.1.weight", "bns.1.bias",
              "bns.2.weight", "gru.weight_hh_l0", "gru.bias_ih_l1_reverse", "gru.weight_hh_l1_reverse", "fc.weight", "fc.bias"):
        d["grad." + k] = dict(m.named_parameters())[k].grad.numpy().copy()
    gw = m.gru.weight_ih_l0.grad.numpy()
    d["grad.gru.weight_ih_l0.rows0_4"] = gw[:4].copy()
    d["grad.gru.weight_ih_l0.colsum"] = gw.sum(0)
    np.savez_compressed(os.path.join(OUT, "g5_sed_c128.npz"), **d)

    # ── G6: metric known answers (random + edge cases + K=6) ──
    d = {}
    rng = np.random.default_rng(1234)
    p = rng.random((6, 8, 1)).astype(np.float32)
    t = (rng.random((6, 8, 1)) > 0.6).astype(np.float32)
    d["p"], d["t"] = p, t
    sc = metrics.compute_scores(p > 0.5, t, 5)
    d["f1_1s"], d["er_1s"] = np.float64(sc["f1_overall_1sec"]), np.float64(sc["er_overall_1sec"])
    d["f1_fr"] = np.float64(metrics.f1_overall_framewise(p > 0.5, t))
    d["er_fr"] = np.float64(metrics.er_overall_framewise(p > 0.5, t))
    p6 = rng.random((5, 7, 6)).astype(np.float32)
    t6 = (rng.random((5, 7, 6)) > 0.5).astype(np.float32)
    d["p6"], d["t6"] = p6, t6
    sc = metrics.compute_scores(p6 > 0.5, t6, 4)
    d["k6_f1_1s"], d["k6_er_1s"] = np.float64(sc["f1_overall_1sec"]), np.float64(sc["er_overall_1sec"])
    d["k6_f1_fr"] = np.float64(metrics.f1_overall_framewise((p6 > 0.5).astype(np.uint8), t6.astype(np.uint8)))
    d["k6_er_fr"] = np.float64(metrics.er_overall_framewise((p6 > 0.5).astype(np.uint8), t6.astype(np.uint8)))
    z = np.zeros((4, 8, 1), np.float32)
    o1 = np.ones((4, 8, 1), np.float32)
    with np.errstate(all="ignore"):
        d["edge_nref0_nsys_f1"] = np.float64(metrics.f1_overall_1sec(o1, z, 5))
        d["edge_nref0_nsys_er"] = np.float64(metrics.er_overall_1sec(o1, z, 5))
        d["edge_allzero_er"] = np.float64(metrics.er_overall_1sec(z, z, 5))
        d["edge_allzero_f1"] = np.float64(metrics.f1_overall_1sec(z, z, 5))
    np.savez_compressed(os.path.join(OUT, "g6_metrics.npz"), **d)

    # ── G7: dataset helpers (next-row material): label pooling and clean negatives ──
    lab = np.zeros((400, 1), np.float32)
    lab[70:75] = 1
    lab[200:203] = 1
    lab[390:395] = 1
    ds = sed.HitWindowDataset(np.zeros((400, 40), np.float32), lab)
    d = {"lab": lab, "neg_starts": np.asarray(ds.neg_starts, np.int64),
         "pos_frames": np.asarray(ds.pos_frames, np.int64),
         "pooled_60": ds._pool_labels(lab[60:124]), "len": np.int64(len(ds))}
    np.savez_compressed(os.path.join(OUT, "g7_dataset.npz"), **d)
    # ── G8: SpecAugment (decorte_datamodule.py:39-49) under seeded np.random, augmented dataset item, utils packing ──
    d = {}
    rs = np.random.RandomState(5)
    for i, seed in enumerate((3, 17, 101)):
        xw = rs.randn(40, 64).astype(np.float32)
        np.random.seed(seed)
        d[f"aug_in{i}"], d[f"aug_seed{i}"] = xw, np.int64(seed)
        d[f"aug_out{i}"] = ddm._spec_augment(xw.copy())
    mel = rs.randn(300, 40).astype(np.float32)
    lab2 = np.zeros((300, 1), np.float32)
    lab2[100:104] = 1
    ds2 = ddm.HitWindowDataset(mel, lab2, augment=False)
    import random as _random
    _random.seed(9)
    xi, yi = ds2[1]                                   # odd index -> a clean negative start
    _random.seed(9)
    start = ds2._rand_neg()
    d["item_mel"], d["item_lab"], d["item_start"] = mel, lab2, np.int64(start)
    d["item_x"], d["item_y"] = xi.numpy(), yi.numpy()
    _random.seed(11)
    xp, yp = ds2[0]                                   # even index -> a window around a positive frame
    _random.seed(11)
    d["item_pos_start"] = np.int64(ds2._rand_pos())
    d["item_pos_x"], d["item_pos_y"] = xp.numpy(), yp.numpy()
    feat = rs.randn(70, 40 * 2)
    seqs = utils.split_in_seqs(feat, 16)
    d["pack_feat"], d["pack_seqs"] = feat, seqs
    d["pack_mc"] = utils.split_multi_channels(seqs, 2)
    np.savez_compressed(os.path.join(OUT, "g8_window_aug.npz"), **d)
    golden_scaler()
    print("goldens written to", os.path.normpath(OUT))
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f))/1e3:.1f} KB")


if __name__ == "__main__":
    main()
