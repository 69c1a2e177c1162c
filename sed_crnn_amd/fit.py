"""Fit-loop mirror of the reference: ``run_epoch`` (sed.py:128-141), the per-fold epoch logic (sed.py:166-202) and the
4-fold driver of ``main()`` (sed.py:144-207).

Same names, arguments and return types; what changes is where time goes.  The reference synchronises twice per step
(``loss.item()`` and ``.cpu().numpy()``, sed.py:138-139) and ships every prediction to the host to score it.  Here an
epoch is tallied on the device (``EpochTally``): per-batch losses, probabilities and labels stay in HBM, the segment
metrics are integer counts produced by one kernel (``sed_segment_counts``), and what crosses PCIe per epoch and split is
one float and 17 integers.  ``run_epoch`` still returns the reference's numpy triple for callers that want it.
"""
import os

import numpy as np
import torch

from . import metrics, ops

FPS_OUT = 5            # reference sed.py:27,32: int(44100/1024)//8
BATCH_SIZE, MAX_EPOCHS, EARLY_STOP = 128, 200, 40          # sed.py:34-36


class EpochTally:
    """What one epoch leaves behind, kept on the device until somebody asks."""

    def __init__(self):
        self.losses, self.probs, self.labels = [], [], []

    def add(self, logits, labels, loss):
        self.losses.append(loss.detach().reshape(1))
        self.probs.append(ops.sigmoid(logits.detach().contiguous()))
        self.labels.append(labels)

    def __len__(self):
        return len(self.losses)

    def mean_loss(self):
        """mean of the per-batch mean losses (sed.py:138,141) — one host sync"""
        if not self.losses:
            raise ValueError("the loader produced no batch (fewer windows than one batch with drop_last?)")
        return torch.cat(self.losses).sum().item() / len(self.losses)

    def tensors(self):
        """(probabilities, labels) of the whole epoch, windows concatenated in visiting order, on the device"""
        if not self.probs:
            raise ValueError("the loader produced no batch (fewer windows than one batch with drop_last?)")
        return torch.cat(self.probs), torch.cat(self.labels)

    def counts(self, frames_in_1_sec=FPS_OUT, threshold=0.5):
        p, t = self.tensors()
        return metrics.device_counts(p, t, frames_in_1_sec, threshold).cpu().tolist()

    def scores(self, frames_in_1_sec=FPS_OUT, threshold=0.5):
        """frame-wise and 1-second F1 / ER (+ confusion matrix) from the device-side counts"""
        return metrics.scores_from_counts(self.counts(frames_in_1_sec, threshold))


def run_epoch_device(model, loader, loss_fn, optim=None, device=None):
    """One pass over ``loader`` (training iff ``optim`` is given) -> EpochTally; no host synchronisation inside."""
    train = optim is not None
    model.train(train)
    device = device or next(model.parameters()).device
    tally = EpochTally()
    for xb, yb in loader:
        xb = xb.to(device, non_blocking=True)
        yb = yb.to(device, non_blocking=True).float()
        if train:
            optim.zero_grad()
            out = model(xb)
            loss = loss_fn(out, yb)
            loss.backward()
            optim.step()
        else:
            with torch.no_grad():
                out = model(xb)
                loss = loss_fn(out, yb)
        tally.add(out, yb, loss)
    return tally


def run_epoch(model, loader, loss_fn, optim=None, device=None):
    """-> (mean of per-batch mean losses, preds [N,T',K] float32 = sigmoid(logits), labels [N,T',K]) like sed.py:128-141"""
    tally = run_epoch_device(model, loader, loss_fn, optim, device)
    p, t = tally.tensors()
    return tally.mean_loss(), p.cpu().numpy(), t.cpu().numpy()


def fit(model, train_loader, val_loader, loss_fn, optim, max_epochs=MAX_EPOCHS, early_stop=EARLY_STOP, fps_out=FPS_OUT,
        on_epoch=None, save_best=None):
    """Epoch logic of reference sed.py:166-202: threshold 0.5, 1-second scores with blocks of ``fps_out`` frames, keep the
    best validation ER (strictly smaller wins; saved as a bare state_dict like sed.py:198-199), stop once
    ``early_stop`` + 1 epochs in a row brought no improvement (``no_imp > EARLY_STOP``, sed.py:200-202)."""
    best_er, best_epoch, no_imp, history = float("inf"), 0, 0, []
    for epoch in range(1, max_epochs + 1):
        tr = run_epoch_device(model, train_loader, loss_fn, optim)
        va = run_epoch_device(model, val_loader, loss_fn)
        tr_s, va_s = tr.scores(fps_out), va.scores(fps_out)
        rec = dict(epoch=epoch, train_loss=tr.mean_loss(), val_loss=va.mean_loss(), train_f1=tr_s["f1_overall_1sec"],
                   val_f1=va_s["f1_overall_1sec"], val_er=va_s["er_overall_1sec"])
        history.append(rec)
        if on_epoch:
            on_epoch(rec)
        if rec["val_er"] < best_er:
            best_er, best_epoch, no_imp = rec["val_er"], epoch, 0
            if save_best:
                torch.save(model.state_dict(), save_best)
        else:
            no_imp += 1
        if no_imp > early_stop:
            break
    return dict(best_er=best_er, best_epoch=best_epoch, history=history)


def fit_folds(cache_dir, art_dir=None, folds=(1, 2, 3, 4), model_factory=None, batch_size=BATCH_SIZE,
              max_epochs=MAX_EPOCHS, early_stop=EARLY_STOP, lr=1e-3, seed=0, device="cuda", on_epoch=None):
    """The 4-fold driver of reference ``main()`` (sed.py:144-207) on the GPU-resident data path.

    Per fold: the fold pack ``mbe_mon_fold{n}.npz`` (feature.py:131-132) goes to the device once, train windows are drawn
    shuffled with ``drop_last`` and validation windows in order (sed.py:153-156), a fresh ``TimePooledCRNN`` is trained
    with Adam(lr) on BCEWithLogits (sed.py:158-160), the best-ER weights are saved as ``best_fold{n}.pt`` under
    ``art_dir`` (sed.py:196-199).  Returns the per-fold results, the list of best error rates and their mean
    (sed.py:204,207).  ``model_factory()`` opens up the network for the configs the reference hard-codes away."""
    from .data import GpuWindowLoader, HitWindowSet, load_fold_npz
    from .losses import BCEWithLogitsLoss
    from .model import TimePooledCRNN
    from .optim import FusedAdam
    if art_dir:
        os.makedirs(art_dir, exist_ok=True)
    results, error_rates = {}, []
    for fold_id in folds:
        fd = load_fold_npz(cache_dir, fold_id)
        train_ds = HitWindowSet(fd["train_x"], fd["train_y"], device=device, seed=seed + 2 * fold_id)
        val_ds = HitWindowSet(fd["val_x"], fd["val_y"], device=device, seed=seed + 2 * fold_id + 1)
        train_ld = GpuWindowLoader(train_ds, batch_size, shuffle=True, drop_last=True)
        val_ld = GpuWindowLoader(val_ds, batch_size, shuffle=False)
        model = (model_factory() if model_factory else TimePooledCRNN()).to(device)
        optim = FusedAdam(model.parameters(), lr=lr)
        path = os.path.join(art_dir, f"best_fold{fold_id}.pt") if art_dir else None
        cb = (lambda rec, f=fold_id: on_epoch(f, rec)) if on_epoch else None
        res = fit(model, train_ld, val_ld, BCEWithLogitsLoss(), optim, max_epochs, early_stop, on_epoch=cb, save_best=path)
        res["checkpoint"] = path
        results[fold_id] = res
        error_rates.append(res["best_er"])
    return dict(folds=results, error_rates=error_rates, mean_er=float(np.mean(error_rates)))
