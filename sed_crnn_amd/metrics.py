"""Segment-based F1 / error-rate, API-compatible with reference metrics.py:20-74 (+ utils.py:4,11-12).

Per-epoch host-side integer work in the reference too (sed.py:175-176; crnn_lightning.py:123-126), so it
stays numpy here; written block-vectorised instead of the reference's Python loop over blocks.  Quirks
kept on purpose: blocks straddle window boundaries, F1 keeps the partial last block
(ceil) while ER drops it (floor), ER has no zero-reference guard.
"""
import numpy as np

eps = np.finfo(float).eps


def reshape_3Dto2D(A):
    return A.reshape(A.shape[0] * A.shape[1], A.shape[2])


def _prep(O, T):
    O, T = np.asarray(O), np.asarray(T)
    if O.ndim == 3:
        O, T = reshape_3Dto2D(O), reshape_3Dto2D(T)
    if O.dtype == bool:
        O = O.astype(np.uint8)
    if T.dtype == bool:
        T = T.astype(np.uint8)
    return O, T


def f1_overall_framewise(O, T):
    O, T = _prep(O, T)
    TP = float(((2 * T - O) == 1).sum())
    Nref, Nsys = float(T.sum()), float(O.sum())
    prec = TP / (Nsys + eps)
    recall = TP / (Nref + eps)
    return 2 * prec * recall / (prec + recall + eps)


def er_overall_framewise(O, T):
    O, T = _prep(O, T)
    FP = np.logical_and(T == 0, O == 1).sum(1)
    FN = np.logical_and(T == 1, O == 0).sum(1)
    S = np.minimum(FP, FN).sum()
    D = np.maximum(0, FN - FP).sum()
    I = np.maximum(0, FP - FN).sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        return (S + D + I) / (T.sum() + 0.0)


def _blocks(A, block, n):
    """max over consecutive row blocks; n blocks (a partial last block is padded with the identity 0)."""
    rows = n * block
    K = A.shape[1]
    buf = np.zeros((rows, K), dtype=np.float64)
    m = min(rows, A.shape[0])
    buf[:m] = A[:m]
    return buf.reshape(n, block, K).max(axis=1) if n else np.zeros((0, K))


def f1_overall_1sec(O, T, block_size):
    O, T = _prep(O, T)
    n = int(np.ceil(O.shape[0] / block_size))
    return f1_overall_framewise(_blocks(O, block_size, n), _blocks(T, block_size, n))


def er_overall_1sec(O, T, block_size):
    O, T = _prep(O, T)
    n = int(O.shape[0] / block_size)
    return er_overall_framewise(_blocks(O, block_size, n), _blocks(T, block_size, n))


def compute_scores(pred, y, frames_in_1_sec=50):
    return {"f1_overall_1sec": f1_overall_1sec(pred, y, frames_in_1_sec),
            "er_overall_1sec": er_overall_1sec(pred, y, frames_in_1_sec)}


N_COUNTS = 17                  # SED_SEGMENT_COUNTS of include/sedcrnn.h


def device_counts(pred, y, frames_in_1_sec=50, threshold=0.5):
    """17 integer counts of the thresholded predictions, computed on the GPU (one kernel, 136-byte result):
    frame-wise TP,Nref,Nsys,S,D,I | TP,Nref,Nsys over ceil blocks | S,D,I,Nref over floor blocks | tn,fp,fn,tp.
    ``pred``/``y`` are device tensors [..., K] of probabilities and 0/1 labels; rows are taken in storage order, i.e. the
    windows of an epoch concatenated (blocks straddle window boundaries exactly as metrics.py:46-68 does)."""
    import torch
    from ._lib import check, lib, ptr, stream_ptr
    p = pred.reshape(-1, pred.shape[-1]).contiguous().float()
    t = y.reshape(-1, y.shape[-1]).contiguous().float()
    if p.shape != t.shape:
        raise ValueError(f"predictions {tuple(pred.shape)} and labels {tuple(y.shape)} differ in shape")
    out = torch.empty(N_COUNTS, dtype=torch.int64, device=p.device)
    check(lib().sed_segment_counts(ptr(p), ptr(t), p.shape[0], p.shape[1], int(frames_in_1_sec), float(threshold), ptr(out),
                                   stream_ptr()), "sed_segment_counts")
    return out


def scores_from_counts(c):
    """the reference's float64 formulas (metrics.py:25-29,43-44) applied to the integer counts; also the 2x2 confusion
    matrix [[tn, fp], [fn, tp]] of crnn_lightning.py:115-119 when the counts carry it"""
    c = [int(v) for v in c]

    def f1(tp, nref, nsys):
        prec, rec = float(tp) / float(nsys + eps), float(tp) / float(nref + eps)
        return 2 * prec * rec / (prec + rec + eps)

    def er(s_, d, i, nref):
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.float64(s_ + d + i) / (np.float64(nref) + 0.0)
    out = {"f1_overall_framewise": f1(c[0], c[1], c[2]), "er_overall_framewise": er(c[3], c[4], c[5], c[1]),
           "f1_overall_1sec": f1(c[6], c[7], c[8]), "er_overall_1sec": er(c[9], c[10], c[11], c[12])}
    if len(c) >= 17:
        out["cm"] = np.array([[c[13], c[14]], [c[15], c[16]]])
    return out


def compute_scores_device(pred, y, frames_in_1_sec=50, threshold=0.5):
    """compute_scores(pred > threshold, y, frames_in_1_sec) for device tensors of PROBABILITIES, without copying the
    predictions to the host (17 integers come back instead)."""
    s = scores_from_counts(device_counts(pred, y, frames_in_1_sec, threshold).cpu().tolist())
    return {"f1_overall_1sec": s["f1_overall_1sec"], "er_overall_1sec": s["er_overall_1sec"]}
