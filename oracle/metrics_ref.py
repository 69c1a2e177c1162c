"""numpy restatement of the segment-based metrics.  TEST INFRASTRUCTURE.

Follows /root/reference/metrics.py:14-74 and utils.py:4,11-12.  Quirks kept
(SURVEY appendix B): blocks are cut across the concatenation of all windows,
F1 keeps the partial last block (ceil, metrics.py:50) while ER drops it (floor,
metrics.py:62), ER has no zero-reference guard (inf / nan), eps = 2.22e-16.

Pinned by tests/golden/g6_metrics.npz (values produced by the imported
reference) and the known answers of SURVEY 8(c).
"""
import numpy as np

EPS = np.finfo(float).eps


def _as2d(a):
    a = np.asarray(a)
    if a.dtype == bool:
        a = a.astype(np.uint8)
    if a.ndim == 3:                                   # utils.py:11-12
        a = a.reshape(a.shape[0] * a.shape[1], a.shape[2])
    return a


def f1_framewise(O, T):
    O, T = _as2d(O), _as2d(T)
    tp = float(((2 * T - O) == 1).sum())
    nref, nsys = float(T.sum()), float(O.sum())
    prec = tp / (nsys + EPS)
    rec = tp / (nref + EPS)
    return 2 * prec * rec / (prec + rec + EPS)


def er_framewise(O, T):
    O, T = _as2d(O), _as2d(T)
    fp = np.logical_and(T == 0, O == 1).sum(1)
    fn = np.logical_and(T == 1, O == 0).sum(1)
    s = np.minimum(fp, fn).sum()
    d = np.maximum(0, fn - fp).sum()
    i = np.maximum(0, fp - fn).sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        return (s + d + i) / (T.sum() + 0.0)


def _block_max(a, block, nblocks):
    out = np.zeros((nblocks, a.shape[1]))
    for k in range(nblocks):
        out[k] = a[k * block:(k + 1) * block].max(axis=0)
    return out


def f1_1sec(O, T, block):
    O, T = _as2d(O), _as2d(T)
    n = int(np.ceil(O.shape[0] / block))
    return f1_framewise(_block_max(O, block, n), _block_max(T, block, n))


def er_1sec(O, T, block):
    O, T = _as2d(O), _as2d(T)
    n = int(O.shape[0] / block)
    return er_framewise(_block_max(O, block, n), _block_max(T, block, n))


def compute_scores(pred, y, frames_in_1_sec=50):
    return {"f1_overall_1sec": f1_1sec(pred, y, frames_in_1_sec),
            "er_overall_1sec": er_1sec(pred, y, frames_in_1_sec)}


def segment_counts(O, T, block):
    """The 17 integers the device kernel hands back instead of the predictions (include/sedcrnn.h SED_SEGMENT_COUNTS), from
    the reference's own intermediate quantities (metrics.py:20-68; confusion counts of crnn_lightning.py:115-119):
    frame-wise TP, Nref, Nsys, S, D, I | TP, Nref, Nsys over ceil blocks | S, D, I, Nref over floor blocks | tn, fp, fn, tp."""
    O, T = _as2d(O).astype(np.int64), _as2d(T).astype(np.int64)

    def f1_counts(o, t):
        return [int(((2 * t - o) == 1).sum()), int(t.sum()), int(o.sum())]

    def er_counts(o, t):
        fp = np.logical_and(t == 0, o == 1).sum(1)
        fn = np.logical_and(t == 1, o == 0).sum(1)
        return [int(np.minimum(fp, fn).sum()), int(np.maximum(0, fn - fp).sum()), int(np.maximum(0, fp - fn).sum())]
    nc, nf = int(np.ceil(O.shape[0] / block)), int(O.shape[0] / block)
    oc, tc = _block_max(O, block, nc).astype(np.int64), _block_max(T, block, nc).astype(np.int64)
    of, tf = _block_max(O, block, nf).astype(np.int64), _block_max(T, block, nf).astype(np.int64)
    cm = [int(((T == 0) & (O == 0)).sum()), int(((T == 0) & (O == 1)).sum()),
          int(((T == 1) & (O == 0)).sum()), int(((T == 1) & (O == 1)).sum())]
    return f1_counts(O, T) + er_counts(O, T) + f1_counts(oc, tc) + er_counts(of, tf) + [int(tf.sum())] + cm
