"""Mirror of the reference LightningModule (crnn_lightning.py:79-200) and of the Trainer settings of
train_lightning.py:27-63, running on the HIP path.

``CRNNLightning(fold_id, art_dir, lr=1e-3, weight_decay=1e-4, dropout=0.4)`` keeps the reference's hooks, log keys
(``train_loss``, ``val_loss``, ``val_er_1s``, ``val_f1_1s``), public dicts (``_buf``, ``track``) and
``configure_optimizers()`` contract.  When ``pytorch_lightning`` is importable the class derives from
``pl.LightningModule`` and can be handed to ``pl.Trainer``; it is absent from this image, so ``fit_lightning`` below
reproduces what the reference's Trainer does for this path (fit/validate epochs, gradient_clip_val=1.0,
ReduceLROnPlateau on ``val_loss``, EarlyStopping(patience) and per-epoch checkpoints named
``epoch{epoch:03d}-valer{val_er_1s:.3f}`` + ``last``), without Lightning.
"""
import os
import types

import numpy as np
import torch
import torch.nn as nn

from . import metrics
from .losses import FocalBCELoss
from .model import LightningTimePooledCRNN
from .optim import FusedAdam

FPS_OUT = 5                                   # train_constants.py:20-21

try:                                          # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:                             # ModuleNotFoundError here
    pl = None

    class _Base(nn.Module):
        """The few LightningModule services the reference module uses."""
        current_epoch = 0

        def save_hyperparameters(self, **kw):
            self.hparams = types.SimpleNamespace(**kw)

        def log(self, name, value, **_):
            if not hasattr(self, "logged"):
                self.logged = {}
            self.logged[name] = value


class CRNNLightning(_Base):
    def __init__(self, fold_id: int, art_dir: str, lr=1e-3, weight_decay=1e-4, dropout=0.4, **model_kw):
        super().__init__()
        if pl is not None:                    # pragma: no cover
            self.save_hyperparameters(ignore=["art_dir"])
        else:
            self.save_hyperparameters(fold_id=fold_id, lr=lr, weight_decay=weight_decay, dropout=dropout)
        self.art_dir = art_dir
        self.model = LightningTimePooledCRNN(dropout, **model_kw)
        self.loss_fn = FocalBCELoss()
        self._buf = {m: {"preds": [], "trues": [], "losses": []} for m in ["train", "val"]}
        self.track = {k: [] for k in [
            "loss_tr", "loss_val", "f1_1s_tr", "f1_1s_val", "er_1s_tr", "er_1s_val",
            "f1_fr_tr", "f1_fr_val", "er_fr_tr", "er_fr_val"]}

    def forward(self, x):
        return self.model(x)

    # ── helpers (crnn_lightning.py:97-129) ──
    def _collect(self, logits, y, loss, mode):
        from . import ops
        self._buf[mode]["preds"].append(ops.sigmoid(logits.detach().contiguous()))     # stays on the device
        self._buf[mode]["trues"].append(y)
        self._buf[mode]["losses"].append(loss.detach())

    def _aggregate(self, mode):
        p_t = torch.cat(self._buf[mode]["preds"])
        t_t = torch.cat(self._buf[mode]["trues"])
        loss = torch.stack([l.reshape(()) for l in self._buf[mode]["losses"]]).mean().item()
        for k in self._buf[mode]:
            self._buf[mode][k].clear()
        p, t = p_t.detach().cpu().numpy(), t_t.detach().cpu().numpy()          # one D2H per epoch
        p_bin, t_bin = (p > 0.5).astype(np.uint8), t.astype(np.uint8)
        tn = np.logical_and(p_bin == 0, t_bin == 0).sum()
        fp = np.logical_and(p_bin == 1, t_bin == 0).sum()
        fn = np.logical_and(p_bin == 0, t_bin == 1).sum()
        tp = np.logical_and(p_bin == 1, t_bin == 1).sum()
        return dict(loss=loss,
                    f1_frame=metrics.f1_overall_framewise(p_bin, t_bin), er_frame=metrics.er_overall_framewise(p_bin, t_bin),
                    f1_1s=metrics.f1_overall_1sec(p_bin, t_bin, FPS_OUT), er_1s=metrics.er_overall_1sec(p_bin, t_bin, FPS_OUT),
                    cm=np.array([[tn, fp], [fn, tp]]))

    def _plot_epoch(self, epoch, tr, val):
        """2x3 dashboard of crnn_lightning.py:131-154 (loss / F1 / ER curves + the two confusion matrices)."""
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:                     # plotting is optional plumbing, never part of the compute path
            return None
        os.makedirs(self.art_dir, exist_ok=True)
        fig, ax = plt.subplots(2, 3, figsize=(14, 6))
        for a, (k, title) in zip([ax[0, 0], ax[0, 1], ax[0, 2], ax[1, 2]],
                                 [("loss", "Focal Loss"), ("f1_1s", "F1 (1 s)"), ("er_1s", "ER (1 s)"), ("f1_fr", "F1 (frame)")]):
            a.plot(self.track[f"{k}_tr"], label="train")
            a.plot(self.track[f"{k}_val"], label="val")
            a.set_title(title); a.set_xlabel("Epoch"); a.grid(); a.legend()
        for a, (m, title) in zip([ax[1, 0], ax[1, 1]], [(tr["cm"], f"Train CM (e{epoch})"), (val["cm"], f"Val CM (e{epoch})")]):
            a.imshow(m, cmap="Blues")
            for i in range(2):
                for j in range(2):
                    a.text(j, i, f"{m[i, j]}", ha="center", va="center")
            a.set_xticks([0, 1]); a.set_yticks([0, 1]); a.set_xlabel("Pred"); a.set_ylabel("True"); a.set_title(title)
        out = os.path.join(self.art_dir, f"metrics_fold{self.hparams.fold_id}.png")
        fig.tight_layout(); fig.savefig(out); plt.close(fig)
        return out

    # ── Lightning hooks (crnn_lightning.py:157-193) ──
    def training_step(self, batch, _):
        x, y = batch
        logits = self(x)
        loss = self.loss_fn(logits, y)
        self._collect(logits, y, loss, "train")
        self.log("train_loss", loss, on_epoch=True, prog_bar=True)
        return loss

    def on_train_epoch_end(self):
        tr = self._aggregate("train")
        self.track["loss_tr"].append(tr["loss"])
        self.track["f1_1s_tr"].append(tr["f1_1s"]); self.track["er_1s_tr"].append(tr["er_1s"])
        self.track["f1_fr_tr"].append(tr["f1_frame"]); self.track["er_fr_tr"].append(tr["er_frame"])
        self._last_train = tr

    def validation_step(self, batch, _):
        x, y = batch
        logits = self(x)
        loss = self.loss_fn(logits, y)
        self._collect(logits, y, loss, "val")
        self.log("val_loss", loss, on_epoch=True, prog_bar=True)

    def on_validation_epoch_end(self):
        val = self._aggregate("val")
        self.track["loss_val"].append(val["loss"])
        self.track["f1_1s_val"].append(val["f1_1s"]); self.track["er_1s_val"].append(val["er_1s"])
        self.track["f1_fr_val"].append(val["f1_frame"]); self.track["er_fr_val"].append(val["er_frame"])
        self.log("val_er_1s", val["er_1s"], prog_bar=True)
        self.log("val_f1_1s", val["f1_1s"], prog_bar=True)
        if not hasattr(self, "_last_train"):
            self._last_train = val.copy()
        self._last_val = val
        self._plot_epoch(self.current_epoch, self._last_train, val)

    def configure_optimizers(self):
        opt = FusedAdam(self.parameters(), lr=self.hparams.lr, weight_decay=self.hparams.weight_decay)
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=.5, patience=10)
        return {"optimizer": opt, "lr_scheduler": {"scheduler": sched, "monitor": "val_loss"}}


def fit_lightning(module, train_loader, val_loader, max_epochs=200, early_stop=20, gradient_clip_val=1.0,
                  ckpt_dir=None, device="cuda"):
    """What ``pl.Trainer(max_epochs=200, devices=1, gradient_clip_val=1.0, callbacks=[ModelCheckpoint(monitor=
    'val_er_1s', save_top_k=-1, save_last=True), EarlyStopping('val_er_1s', patience=20)]).fit`` does for this
    module (train_lightning.py:32-55).  Returns the per-epoch records."""
    module.to(device)
    cfg = module.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    for g in opt.param_groups:
        g["max_grad_norm"] = gradient_clip_val
    best, stale, hist = float("inf"), 0, []
    for epoch in range(max_epochs):
        module.current_epoch = epoch
        module.train()
        for xb, yb in train_loader:
            opt.zero_grad()
            loss = module.training_step((xb.to(device), yb.to(device).float()), 0)
            loss.backward()
            opt.step()
        module.on_train_epoch_end()
        module.eval()
        with torch.no_grad():
            for xb, yb in val_loader:
                module.validation_step((xb.to(device), yb.to(device).float()), 0)
        module.on_validation_epoch_end()
        val = module._last_val
        sched.step(val["loss"])
        rec = dict(epoch=epoch, train_loss=module.track["loss_tr"][-1], val_loss=val["loss"], val_er_1s=val["er_1s"],
                   val_f1_1s=val["f1_1s"], lr=opt.param_groups[0]["lr"])
        hist.append(rec)
        if ckpt_dir:
            os.makedirs(ckpt_dir, exist_ok=True)
            sd = {"state_dict": {"model." + k: v for k, v in module.model.state_dict().items()}, "epoch": epoch}
            torch.save(sd, os.path.join(ckpt_dir, f"epoch{epoch:03d}-valer{val['er_1s']:.3f}.ckpt"))
            torch.save(sd, os.path.join(ckpt_dir, "last.ckpt"))
        if val["er_1s"] < best:
            best, stale = val["er_1s"], 0
        else:
            stale += 1
            if stale >= early_stop:
                break
    return hist
