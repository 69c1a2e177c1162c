"""Log-mel front end on the GPU: ``_mbe`` of reference feature.py:55-59 as one HIP kernel.

``mbe(y)`` = log(mel_basis @ |STFT(y, n_fft=2048, hop=1024)|^2).T with librosa's defaults (periodic Hann,
center=True, Slaney mel scale and area normalisation, fmin 0, fmax sr/2).  librosa changed its STFT
edge padding from 'reflect' to 'constant' in 0.10, so ``pad_mode`` is explicit.  The constant tables
(window, FFT twiddles, mel basis) are computed once on the host in float64 and cached per device.
feature.py:127-129's StandardScaler can be fused in through ``mean``/``std``.
"""
import functools

import numpy as np
import torch

from ._lib import check, lib, ptr, stream_ptr

SR, NFFT, HOP, NB_MEL = 44_100, 2048, 1024, 40        # reference feature.py:29-32
_LM_OFF_ENT = 8 + 2048 + 2048 + 1028                  # logmel.hip: header, window, inter-pass twiddles, pairing twiddles


def slaney_mel_basis(sr=SR, n_fft=NFFT, n_mels=NB_MEL):
    """librosa.filters.mel(sr, n_fft, n_mels) with its defaults (htk=False, norm='slaney')."""
    f_sp, brk = 200.0 / 3.0, 1000.0
    brk_mel, step = brk / f_sp, np.log(6.4) / 27.0

    def to_mel(f):
        return np.where(f >= brk, brk_mel + np.log(np.maximum(f, brk) / brk) / step, f / f_sp)

    def to_hz(m):
        return np.where(m >= brk_mel, brk * np.exp(step * (m - brk_mel)), f_sp * m)
    edges = to_hz(np.linspace(to_mel(np.float64(0.0)), to_mel(np.float64(sr / 2.0)), n_mels + 2))
    bins = np.arange(n_fft // 2 + 1, dtype=np.float64) * (sr / n_fft)
    lo, ce, hi = edges[:-2, None], edges[1:-1, None], edges[2:, None]
    tri = np.minimum((bins[None, :] - lo) / (ce - lo), (hi - bins[None, :]) / (hi - ce))
    tri = np.maximum(tri, 0.0) * (2.0 / (hi - lo))
    return tri.astype(np.float32)


def hann_periodic(n_fft=NFFT):
    """scipy.signal.get_window('hann', n_fft, fftbins=True) = librosa's default STFT window, float32"""
    n = np.arange(n_fft, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)).astype(np.float32)


def build_tables(window, melfb, device):
    """the kernel's constant blob (window, FFT twiddles, sparse mel plan) for an arbitrary window [n_fft] and filterbank
    [n_mels, n_fft//2+1] (host arrays): built on the host by the library, then copied to ``device``"""
    import ctypes as C
    window = np.ascontiguousarray(window, dtype=np.float32)
    fb = np.ascontiguousarray(melfb, dtype=np.float32)
    n_fft, n_mels = window.shape[0], fb.shape[0]
    if fb.shape != (n_mels, n_fft // 2 + 1):
        raise ValueError(f"filterbank must be [n_mels, {n_fft // 2 + 1}], got {fb.shape}")
    hp = lambda a: C.c_void_p(a.ctypes.data)                                   # noqa: E731  host pointers
    nbytes = lib().sed_logmel_tables_bytes(hp(fb), n_fft, n_mels)
    if nbytes == 0:
        raise ValueError(f"sed_logmel cannot plan this filterbank (n_fft must be {NFFT}, n_mels <= 128, and at most 8192 "
                         f"non-zero weights so that the plan fits LDS beside the FFT scratch; large plans run with fewer waves per CU); got n_fft={n_fft}, n_mels={n_mels}, "
                         f"{int(np.count_nonzero(fb))} non-zeros")
    blob = np.zeros(nbytes // 4, dtype=np.uint32)
    check(lib().sed_logmel_build_tables(hp(window), hp(fb), n_fft, n_mels, hp(blob), nbytes), "sed_logmel_build_tables")
    return torch.from_numpy(blob.view(np.int32)).to(device)


@functools.lru_cache(maxsize=8)
def _tables(device_index, sr, n_fft, n_mels):
    """librosa's defaults (periodic Hann, Slaney bank), cached per (device, configuration)"""
    return build_tables(hann_periodic(n_fft), slaney_mel_basis(sr, n_fft, n_mels), torch.device("cuda", device_index))


_VALIDATED = {}          # (data_ptr, numel, version counter) of a caller's table blob -> its mel count


def _validated_mels(tables):
    """The library cannot read device memory to validate a caller's blob, and the kernel trusts its header: check it here —
    ONCE per blob (keyed by address, size and torch's in-place version counter), not per clip: the check is a blocking
    device-to-host copy (round-3 advisor)."""
    key = (tables.data_ptr(), tables.numel(), tables._version)
    n = _VALIDATED.get(key)
    if n is None:
        hdr = [int(v) & 0xFFFFFFFF for v in tables[:8].cpu().tolist()] if tables.numel() >= 8 else []
        ok = (len(hdr) == 8 and hdr[0] == 0x4C4D3332 and hdr[4] == tables.numel() and 1 <= hdr[1] <= 128 and
              ((hdr[5] == 0 and _LM_OFF_ENT + hdr[2] * 64 + hdr[1] <= hdr[4]) or
               (hdr[5] == 1 and hdr[2] == 33 and _LM_OFF_ENT + 33 * 32 * 4 + hdr[1] * 4 <= hdr[4])))
        if not ok:
            raise ValueError("tables is not a blob written by feature.build_tables / sed_logmel_build_tables (bad header)")
        if len(_VALIDATED) >= 64:
            _VALIDATED.clear()
        n = _VALIDATED[key] = hdr[1]
    return n


def mbe(y, sr=SR, n_fft=NFFT, hop=HOP, n_mels=NB_MEL, pad_mode="constant", mean=None, std=None, tables=None):
    """y: mono float32 PCM CUDA tensor [N] -> [1 + N//hop, n_mels] log-mel energies (natural log, no eps).
    ``tables`` = build_tables(window, melfb, device) replaces librosa's default window / filterbank."""
    if not (isinstance(y, torch.Tensor) and y.is_cuda):
        raise RuntimeError("sed_crnn_amd.feature.mbe needs a CUDA(HIP) tensor; there is no CPU fallback")
    if pad_mode not in ("constant", "reflect"):
        raise ValueError(f"pad_mode must be 'constant' or 'reflect', got {pad_mode!r}")
    y = y.contiguous().float()
    if tables is None:
        tables = _tables(y.device.index or 0, sr, n_fft, n_mels)
    else:
        n_mels = _validated_mels(tables)
    frames = 1 + y.numel() // hop
    out = torch.empty(frames, n_mels, device=y.device)
    inv = None
    if mean is not None:
        mean = mean.to(y.device).float().contiguous()
        inv = (1.0 / std.to(y.device).double()).float().contiguous()
    check(lib().sed_logmel(ptr(y), y.numel(), ptr(tables), tables.numel() * 4, ptr(mean), ptr(inv), ptr(out), n_fft, hop,
                           n_mels, {"constant": 0, "reflect": 1}[pad_mode], stream_ptr()), "sed_logmel")
    return out


# ───────────────────────── feature.py's on-disk formats ─────────────────────────
def rasterize_hits(n_frames, hits, sr=SR, hop=HOP):
    """Frame labels of one recording from its hit intervals in seconds (feature.py:89-93): frames
    [floor(start*sr/hop), ceil(end*sr/hop)) are 1.  -> float32 [n_frames, 1]"""
    lbl = np.zeros((n_frames, 1), dtype=np.float32)
    for start, end in hits:
        s = int(np.floor(start * sr / hop))
        e = int(np.ceil(end * sr / hop))
        lbl[s:e, 0] = 1.0
    return lbl


def save_video_npz(path, mbe_frames, labels):
    """per-recording cache `{base}_mon.npz` (feature.py:72,95): positional arr_0 = features [n, n_mels], arr_1 = labels [n, 1]"""
    np.savez(path, np.asarray(mbe_frames, dtype=np.float32), np.asarray(labels, dtype=np.float32))


def load_video_npz(path):
    with np.load(path, allow_pickle=False) as d:
        if "arr_0" not in d.files or "arr_1" not in d.files:
            raise ValueError(f"{path}: not a per-recording cache (feature.py:95 saves two positional arrays)")
        mbe_frames, labels = d["arr_0"], d["arr_1"]
    if mbe_frames.shape[0] != labels.shape[0]:
        raise ValueError(f"{path}: {mbe_frames.shape[0]} feature frames but {labels.shape[0]} label frames")
    return mbe_frames, labels


def build_fold_packs(per_video, cache_dir, device="cuda"):
    """feature.py:114-132 on the device: for every fold f the recordings of fold f are the test split and all others the
    train split (concatenated in dict order), a StandardScaler is fitted on the train split and applied to both
    (`data.standard_scaler_fit` / `standard_scaler_transform`: sklearn's rules, float64 statistics), and the pack is saved
    as `mbe_mon_fold{f+1}.npz` with the positional arrays X_train, Y_train, X_test, Y_test that sed.py:119-123 reads.
    ``per_video``: {name: (features [n, F] host or device, labels [n, 1], fold_id)}.  Returns the written paths."""
    import os
    from .data import standard_scaler_fit, standard_scaler_transform
    folds = max(v[2] for v in per_video.values()) + 1
    dev = torch.device(device)
    as_dev = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))).to(dev).float()   # noqa: E731
    feats = {k: (as_dev(v[0]), np.asarray(v[1].cpu() if isinstance(v[1], torch.Tensor) else v[1], dtype=np.float32), v[2])
             for k, v in per_video.items()}
    os.makedirs(cache_dir, exist_ok=True)
    paths = []
    for f in range(folds):
        test = [v for v in feats.values() if v[2] == f]
        train = [v for v in feats.values() if v[2] != f]
        if not test or not train:
            raise ValueError(f"fold {f} has {len(test)} test and {len(train)} train recordings")
        x_train, x_test = torch.cat([v[0] for v in train]), torch.cat([v[0] for v in test])
        mean, scale = standard_scaler_fit(x_train)
        x_train = standard_scaler_transform(x_train, mean, scale, out=x_train)
        x_test = standard_scaler_transform(x_test, mean, scale, out=x_test)
        path = os.path.join(cache_dir, f"mbe_mon_fold{f + 1}.npz")
        np.savez(path, x_train.cpu().numpy(), np.concatenate([v[1] for v in train]), x_test.cpu().numpy(),
                 np.concatenate([v[1] for v in test]))
        paths.append(path)
    return paths
