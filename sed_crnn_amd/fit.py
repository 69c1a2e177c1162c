"""Fit-loop mirror of reference sed.py:128-141 (``run_epoch``) and the epoch logic of sed.py:166-202.

Same signature and return types; what changes is where time goes: losses and predictions are
accumulated on the device and copied to the host ONCE per epoch instead of ``loss.item()`` and
``.cpu().numpy()`` every step (sed.py:138-139), and validation runs without building autograd graphs.
"""
import torch

from . import metrics, ops

FPS_OUT = 5            # reference sed.py:27,32: int(44100/1024)//8


def run_epoch(model, loader, loss_fn, optim=None, device=None):
    """-> (mean of per-batch mean losses, preds [N,T',K] float32 = sigmoid(logits), labels [N,T',K])"""
    train = optim is not None
    model.train() if train else model.eval()
    device = device or next(model.parameters()).device
    losses, preds, labels = [], [], []
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
        losses.append(loss.detach().reshape(1))
        preds.append(ops.sigmoid(out.detach().contiguous()))
        labels.append(yb)
    total = torch.cat(losses).sum().item()                      # the one host sync of the epoch
    return (total / len(losses), torch.cat(preds).cpu().numpy(), torch.cat(labels).cpu().numpy())


def fit(model, train_loader, val_loader, loss_fn, optim, max_epochs=200, early_stop=40, fps_out=FPS_OUT,
        on_epoch=None, save_best=None):
    """Epoch logic of reference sed.py:166-202: threshold 0.5, compute_scores(frames_in_1_sec=5), keep the
    best validation ER, stop after ``early_stop`` non-improving epochs (``no_imp > early_stop``)."""
    best_er, best_epoch, no_imp, history = float("inf"), 0, 0, []
    for epoch in range(1, max_epochs + 1):
        tr_loss, tr_pred, tr_true = run_epoch(model, train_loader, loss_fn, optim)
        va_loss, va_pred, va_true = run_epoch(model, val_loader, loss_fn)
        tr = metrics.compute_scores(tr_pred > 0.5, tr_true, frames_in_1_sec=fps_out)
        va = metrics.compute_scores(va_pred > 0.5, va_true, frames_in_1_sec=fps_out)
        rec = dict(epoch=epoch, train_loss=tr_loss, val_loss=va_loss, train_f1=tr["f1_overall_1sec"],
                   val_f1=va["f1_overall_1sec"], val_er=va["er_overall_1sec"])
        history.append(rec)
        if on_epoch:
            on_epoch(rec)
        if va["er_overall_1sec"] < best_er:
            best_er, best_epoch, no_imp = va["er_overall_1sec"], epoch, 0
            if save_best:
                torch.save(model.state_dict(), save_best)          # bare state_dict, like sed.py:198-199
        else:
            no_imp += 1
        if no_imp > early_stop:
            break
    return dict(best_er=best_er, best_epoch=best_epoch, history=history)
