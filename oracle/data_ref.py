"""numpy restatement of the window dataset / augmentation / packing helpers.  TEST INFRASTRUCTURE.

Follows /root/reference/sed.py:48-79 and decorte_datamodule.py:18-24,39-49,54-111 (HitWindowDataset, clean
negatives, label pooling, SpecAugment) and utils.py:15-41 (split_multi_channels, split_in_seqs).
Pinned by tests/golden/g7_dataset.npz and g8_window_aug.npz (captured from the imported reference).
"""
import numpy as np


def find_clean_negatives(lab, seq_len):
    """window starts whose `seq_len` frames hold no positive frame (np.convolve(mask, ones, 'valid') == 0)"""
    mask = (lab[:, 0] == 1).astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(mask)])
    return np.where(cs[seq_len:] - cs[:-seq_len] == 0)[0]


def pool_labels(lab_win, seq_len_out):
    lab_win = lab_win[:, None] if lab_win.ndim == 1 else lab_win
    return lab_win.reshape(seq_len_out, -1).max(axis=1, keepdims=True)


def draw_spec_masks(rs, n_mels, seq_len, time_w=8, freq_w=8, n_masks=2):
    """the reference's draw order: per mask iteration first the time offset, then the mel offset
    (np.random.randint(0, n - W), high exclusive); -1 = mask skipped because the axis is not longer than W"""
    t, f = [], []
    for _ in range(n_masks):
        t.append(int(rs.randint(0, seq_len - time_w)) if seq_len > time_w else -1)
        f.append(int(rs.randint(0, n_mels - freq_w)) if n_mels > freq_w else -1)
    return t, f


def spec_augment(x, tmask, fmask, time_w=8, freq_w=8):
    """x (F, L): zero `time_w` frames at every tmask offset and `freq_w` mel bins at every fmask offset"""
    x = x.copy()
    for t0, f0 in zip(tmask, fmask):
        if t0 >= 0:
            x[:, t0:t0 + time_w] = 0.0
        if f0 >= 0:
            x[f0:f0 + freq_w, :] = 0.0
    return x


def window_item(mel, lab, start, seq_len, seq_len_out, tmask=None, fmask=None, n_channels=1):
    """one dataset item: x (C, F, L) float32, y (L_out, K) float32"""
    F = mel.shape[1] // n_channels
    win = mel[start:start + seq_len]                                   # (L, C*F)
    x = np.stack([win[:, c * F:(c + 1) * F].T for c in range(n_channels)])
    if tmask is not None:
        x = np.stack([spec_augment(xc, tmask, fmask) for xc in x])
    lw = lab[start:start + seq_len]
    y = np.stack([lw[:, k].reshape(seq_len_out, -1).max(axis=1) for k in range(lw.shape[1])], axis=1)
    return x.astype(np.float32), y.astype(np.float32)


def split_in_seqs(data, subdivs):
    """utils.py:28-41 for 2-D/3-D input: drop the remainder, reshape to (N//s, s, ...)"""
    if data.shape[0] % subdivs:
        data = data[:-(data.shape[0] % subdivs)]
    return data.reshape((data.shape[0] // subdivs, subdivs) + data.shape[1:])


def split_multi_channels(data, num_channels):
    """utils.py:15-25: (N, S, F*C) -> (N, C, S, F) float64, channel c = columns [c*F, (c+1)*F)"""
    n, s, fc = data.shape
    hop = fc // num_channels
    out = np.zeros((n, num_channels, s, hop))
    for c in range(num_channels):
        out[:, c] = data[:, :, c * hop:(c + 1) * hop]
    return out


def rasterize_hits_ref(n_frames, hits, sr=44100, hop=1024):
    """label raster of /root/reference/feature.py:89-93"""
    lbl = np.zeros((n_frames, 1), dtype=np.float32)
    for start, end in hits:
        lbl[int(np.floor(start * sr / hop)):int(np.ceil(end * sr / hop)), 0] = 1.0
    return lbl


def build_fold_packs_ref(per_video):
    """per-fold train / test concatenation + StandardScaler of /root/reference/feature.py:113-129 (numpy; the scaler is the
    sklearn-pinned restatement of oracle.logmel_ref).  -> {fold_number: (X_train, Y_train, X_test, Y_test)}"""
    from .logmel_ref import standardize_apply, standardize_fit
    fold_k = max(v[2] for v in per_video.values()) + 1
    out = {}
    for f in range(fold_k):
        xtr = ytr = xte = yte = None
        for _, (mbe, lbl, fold) in per_video.items():
            if fold == f:
                xte = mbe if xte is None else np.concatenate((xte, mbe), axis=0)
                yte = lbl if yte is None else np.concatenate((yte, lbl), axis=0)
            else:
                xtr = mbe if xtr is None else np.concatenate((xtr, mbe), axis=0)
                ytr = lbl if ytr is None else np.concatenate((ytr, lbl), axis=0)
        mean, scale = standardize_fit(xtr)
        out[f + 1] = (standardize_apply(xtr, mean, scale), ytr, standardize_apply(xte, mean, scale), yte)
    return out
