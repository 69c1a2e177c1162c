"""The reference LightningModule's API surface (crnn_lightning.py:79-200) and the Trainer settings of
train_lightning.py:27-63, on the HIP path.

``CRNNLightning(fold_id, art_dir, lr=1e-3, weight_decay=1e-4, dropout=0.4)`` keeps what callbacks and users bind to:
the hook names, the log keys (``train_loss``, ``val_loss``, ``val_er_1s``, ``val_f1_1s``), the public dicts ``_buf``
and ``track``, the ``_aggregate(mode)`` result keys and the ``configure_optimizers()`` contract.  The bodies are not the
reference's: an epoch is aggregated from 17 integer counts computed on the device (``sed_segment_counts``) instead of
copying every prediction to the host (crnn_lightning.py:102-129), and both step hooks and both epoch-end hooks go
through one helper each.  The matplotlib dashboard (crnn_lightning.py:131-154) is out of scope.

When ``pytorch_lightning`` is importable the class derives from ``pl.LightningModule`` and can be handed to
``pl.Trainer``; it is absent from this image, so ``fit_lightning`` reproduces what the reference's Trainer does for this
path (fit/validate epochs, gradient_clip_val=1.0, ReduceLROnPlateau on ``val_loss``, EarlyStopping(patience) and
per-epoch checkpoints named ``epoch{epoch:03d}-valer{val_er_1s:.3f}`` + ``last``), without Lightning.
"""
import os
import types

import torch
import torch.nn as nn

from . import metrics, ops
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


_MODES = {"train": "tr", "val": "val"}                                                  # _buf key -> track suffix
_TRACKED = {"loss": "loss", "f1_1s": "f1_1s", "er_1s": "er_1s", "f1_frame": "f1_fr", "er_frame": "er_fr"}   # result key -> track stem


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
        self._buf = {mode: dict(preds=[], trues=[], losses=[]) for mode in _MODES}
        self.track = {f"{stem}_{sfx}": [] for stem in _TRACKED.values() for sfx in _MODES.values()}
        self._last = {}

    def forward(self, x):
        return self.model(x)

    # ── one step / one epoch end, shared by the train and val hooks ──
    def _collect(self, logits, y, loss, mode):
        """keep this batch's probabilities, labels and loss ON THE DEVICE until the epoch ends"""
        buf = self._buf[mode]
        buf["preds"].append(ops.sigmoid(logits.detach().contiguous()))
        buf["trues"].append(y)
        buf["losses"].append(loss.detach().reshape(1))

    def _aggregate(self, mode):
        """-> dict(loss, f1_frame, er_frame, f1_1s, er_1s, cm) like crnn_lightning.py:102-129, from device-side counts:
        threshold > 0.5, labels truncated to uint8, 1-second blocks of FPS_OUT rows over the concatenated windows."""
        buf = self._buf[mode]
        counts = metrics.device_counts(torch.cat(buf["preds"]), torch.cat(buf["trues"]), FPS_OUT, 0.5)
        loss = torch.cat(buf["losses"]).mean()
        for lst in buf.values():
            lst.clear()
        s = metrics.scores_from_counts(counts.cpu().tolist())          # 17 integers + 1 float cross PCIe
        return dict(loss=loss.item(), f1_frame=s["f1_overall_framewise"], er_frame=s["er_overall_framewise"],
                    f1_1s=s["f1_overall_1sec"], er_1s=s["er_overall_1sec"], cm=s["cm"])

    def _step(self, batch, mode):
        x, y = batch
        logits = self(x)
        loss = self.loss_fn(logits, y)
        self._collect(logits, y, loss, mode)
        self.log(f"{mode}_loss", loss, on_epoch=True, prog_bar=True)
        return loss

    def _end_epoch(self, mode):
        res = self._aggregate(mode)
        for key, stem in _TRACKED.items():
            self.track[f"{stem}_{_MODES[mode]}"].append(res[key])
        self._last[mode] = res
        return res

    # ── Lightning hooks (names and log keys of crnn_lightning.py:157-193) ──
    def training_step(self, batch, batch_idx):
        return self._step(batch, "train")

    def validation_step(self, batch, batch_idx):
        self._step(batch, "val")

    def on_train_epoch_end(self):
        self._end_epoch("train")

    def on_validation_epoch_end(self):
        val = self._end_epoch("val")
        self.log("val_er_1s", val["er_1s"], prog_bar=True)
        self.log("val_f1_1s", val["f1_1s"], prog_bar=True)
        self._last.setdefault("train", dict(val))          # Lightning's sanity-check validation runs before any training

    @property
    def _last_train(self):
        return self._last["train"]

    @property
    def _last_val(self):
        return self._last["val"]

    def configure_optimizers(self):
        hp = self.hparams
        opt = FusedAdam(self.parameters(), lr=hp.lr, weight_decay=hp.weight_decay)       # coupled L2, like optim.Adam
        plateau = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=10)
        return dict(optimizer=opt, lr_scheduler=dict(scheduler=plateau, monitor="val_loss"))


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
        for i, (xb, yb) in enumerate(train_loader):
            opt.zero_grad()
            loss = module.training_step((xb.to(device), yb.to(device).float()), i)
            loss.backward()
            opt.step()
        module.on_train_epoch_end()
        module.eval()
        with torch.no_grad():
            for i, (xb, yb) in enumerate(val_loader):
                module.validation_step((xb.to(device), yb.to(device).float()), i)
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
