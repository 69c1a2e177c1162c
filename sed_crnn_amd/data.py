"""GPU-resident data path: what the reference does per sample in Python DataLoader workers.

* ``load_all_npz(folder)``           fold packs ``mbe_mon_fold{1..4}.npz`` (arr_0..arr_3 = X_train, Y_train, X_test,
                                     Y_test; feature.py:131-132, read at sed.py:115-125) — numpy, allow_pickle off.
* ``HitWindowSet``                   the fold stays on the device; windows are gathered by ONE kernel per batch
                                     (sed.py:55-79 / decorte_datamodule.py:54-111: balanced pos/neg starts, label
                                     max-pooling, SpecAugment masks), instead of a Python slice+transpose per sample.
* ``GpuWindowLoader``                epoch iterator with DataLoader(shuffle=True, drop_last=...) semantics over the
                                     2*#positives indices; yields device tensors, plugs into ``run_epoch``.
* ``standard_scaler_fit`` / ``pack_sequences``   feature.py:127-129 and utils.py:15-41 on the device.
Sampling uses a seedable numpy Generator (the reference uses the unseeded ``random`` module: same distribution).
"""
import os

import numpy as np
import torch

from ._lib import check, lib, ptr, stream_ptr

SEQ_LEN_IN, SEQ_LEN_OUT = 64, 8                       # train_constants.py:6-8
TIME_MASK_W, FREQ_MASK_W, MASKS_PER_EX = 8, 8, 2      # train_constants.py:14-16


def load_fold_npz(folder, fold_id):
    """One fold pack ``mbe_mon_fold{n}.npz``: positional arrays arr_0..arr_3 = X_train [N,40], Y_train [N,1], X_test,
    Y_test (written at feature.py:131-132, read at sed.py:119-123)."""
    path = os.path.join(folder, f"mbe_mon_fold{fold_id}.npz")
    with np.load(path, allow_pickle=False) as arr:
        missing = [k for k in ("arr_0", "arr_1", "arr_2", "arr_3") if k not in arr.files]
        if missing:
            raise ValueError(f"{path}: not a fold pack (missing {missing}; feature.py:131-132 saves four positional arrays)")
        fd = {"train_x": arr["arr_0"], "train_y": arr["arr_1"], "val_x": arr["arr_2"], "val_y": arr["arr_3"]}
    for a, b in (("train_x", "train_y"), ("val_x", "val_y")):
        if fd[a].shape[0] != fd[b].shape[0]:
            raise ValueError(f"{path}: {a} has {fd[a].shape[0]} frames but {b} has {fd[b].shape[0]}")
    return fd


def load_all_npz(folder):
    """sed.py:115-125: all four folds, keyed 1..4"""
    return {i: load_fold_npz(folder, i) for i in range(1, 5)}


def find_clean_negatives(lab, seq_len=SEQ_LEN_IN):
    mask = (np.asarray(lab)[:, 0] == 1).astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(mask)])
    return np.where(cs[seq_len:] - cs[:-seq_len] == 0)[0]


class HitWindowSet:
    """Device-resident fold: ``mel`` [N, C*F], ``lab`` [N, K]."""

    def __init__(self, mel, lab, device="cuda", seq_len_in=SEQ_LEN_IN, seq_len_out=SEQ_LEN_OUT, n_channels=1,
                 augment=False, seed=0):
        lab = np.asarray(lab, dtype=np.float32)
        if lab.ndim == 1:
            lab = lab[:, None]
        mel = np.asarray(mel, dtype=np.float32)
        assert mel.shape[0] == lab.shape[0] >= seq_len_in and seq_len_in % seq_len_out == 0
        self.L, self.Lo, self.C = seq_len_in, seq_len_out, n_channels
        self.F = mel.shape[1] // n_channels
        self.K = lab.shape[1]
        self.total_frames = mel.shape[0]
        self.pos_frames = np.where(lab[:, 0] == 1)[0]
        self.neg_starts = find_clean_negatives(lab, seq_len_in)
        self.augment = augment
        self.rng = np.random.default_rng(seed)
        self.mel = torch.from_numpy(mel).to(device)
        self.lab = torch.from_numpy(lab).to(device)

    def __len__(self):
        return 2 * len(self.pos_frames)                       # sed.py:62

    def draw_starts(self, idx):
        """window start per dataset index: even -> around a random positive frame, odd -> a clean negative"""
        idx = np.asarray(idx)
        out = np.empty(len(idx), np.int32)
        ev = idx % 2 == 0
        c = self.rng.choice(self.pos_frames, size=int(ev.sum()))
        a = np.maximum(0, c - self.L + 1)
        b = np.minimum(c, self.total_frames - self.L)
        out[ev] = self.rng.integers(a, b + 1)                 # random.randint(a, b) is inclusive
        out[~ev] = self.rng.choice(self.neg_starts, size=int((~ev).sum()))
        return out

    def draw_masks(self, n):
        """SpecAugment offsets in the reference's draw order (time, then mel, MASKS_PER_EX times); -1 = skipped"""
        t = np.full((n, MASKS_PER_EX), -1, np.int32)
        f = np.full((n, MASKS_PER_EX), -1, np.int32)
        for m in range(MASKS_PER_EX):
            if self.L > TIME_MASK_W:
                t[:, m] = self.rng.integers(0, self.L - TIME_MASK_W, size=n)
            if self.F > FREQ_MASK_W:
                f[:, m] = self.rng.integers(0, self.F - FREQ_MASK_W, size=n)
        return t, f

    def gather(self, starts, tmask=None, fmask=None):
        """-> x [B,C,F,L], y [B,L_out,K] on the device (one kernel launch)"""
        dev = self.mel.device
        starts = torch.as_tensor(np.asarray(starts, np.int32)).to(dev)
        B = starts.numel()
        x = torch.empty(B, self.C, self.F, self.L, device=dev)
        y = torch.empty(B, self.Lo, self.K, device=dev)
        nm = 0
        tm = fm = None
        if tmask is not None:
            tm = torch.as_tensor(np.ascontiguousarray(tmask, np.int32)).to(dev)
            fm = torch.as_tensor(np.ascontiguousarray(fmask, np.int32)).to(dev)
            nm = tm.shape[1]
        check(lib().sed_window_batch(ptr(self.mel), ptr(self.lab), self.total_frames, self.C, self.F, self.K, ptr(starts),
                                     ptr(tm), ptr(fm), nm, TIME_MASK_W, FREQ_MASK_W, ptr(x), ptr(y), B, self.L,
                                     self.L // self.Lo, stream_ptr()), "sed_window_batch")
        return x, y

    def batch(self, idx):
        starts = self.draw_starts(idx)
        if self.augment:
            t, f = self.draw_masks(len(starts))
            return self.gather(starts, t, f)
        return self.gather(starts)


class GpuWindowLoader:
    """``DataLoader(ds, batch_size, shuffle, drop_last)`` semantics (sed.py:153-156) without worker processes."""

    def __init__(self, dataset, batch_size=128, shuffle=True, drop_last=False):
        self.ds, self.bs, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last

    def __len__(self):
        n = len(self.ds)
        return n // self.bs if self.drop_last else -(-n // self.bs)

    def __iter__(self):
        n = len(self.ds)
        order = self.ds.rng.permutation(n) if self.shuffle else np.arange(n)
        for i in range(len(self)):
            yield self.ds.batch(order[i * self.bs:(i + 1) * self.bs])


def standard_scaler_fit(x):
    """per-column (mean, sigma) of a device matrix [N, F] with sklearn StandardScaler's rules: float64 accumulation,
    ddof 0, scale 1 for a (numerically) constant column — pinned by tests/golden/g9_scaler.npz"""
    x = x.contiguous().float()
    N, F = x.shape
    mean, std = (torch.empty(F, device=x.device, dtype=torch.float64) for _ in range(2))      # like sklearn's mean_ / scale_
    ws = torch.empty(lib().sed_col_mean_std_workspace_bytes(F) // 4, device=x.device)
    check(lib().sed_col_mean_std(ptr(x), N, F, ptr(mean), ptr(std), ptr(ws), stream_ptr()), "sed_col_mean_std")
    return mean, std


def standard_scaler_transform(x, mean, std, out=None):
    """(x - mean) / sigma per column on the device (StandardScaler.transform, feature.py:128-129); ``out=x`` works in place"""
    x = x.contiguous().float()
    N, F = x.shape
    out = torch.empty_like(x) if out is None else out
    check(lib().sed_col_standardize(ptr(x), N, F, ptr(mean.to(x.device).contiguous().double()),
                                    ptr(std.to(x.device).contiguous().double()), ptr(out), stream_ptr()), "sed_col_standardize")
    return out


def pack_sequences(feat, seq_len, n_channels=1, time_last=True):
    """utils.split_in_seqs + utils.split_multi_channels: [N, C*F] -> [N//S, C, F, S] (network input) or [N//S, C, S, F]"""
    feat = feat.contiguous().float()
    N, CF = feat.shape
    F = CF // n_channels
    n = N // seq_len
    out = torch.empty((n, n_channels, F, seq_len) if time_last else (n, n_channels, seq_len, F), device=feat.device)
    check(lib().sed_pack_sequences(ptr(feat), N, n_channels, F, seq_len, int(time_last), ptr(out), stream_ptr()), "sed_pack_sequences")
    return out
