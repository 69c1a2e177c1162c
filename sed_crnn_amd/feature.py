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
                         f"non-zero weights so that the plan fits LDS); got n_fft={n_fft}, n_mels={n_mels}, "
                         f"{int(np.count_nonzero(fb))} non-zeros")
    blob = np.zeros(nbytes // 4, dtype=np.uint32)
    check(lib().sed_logmel_build_tables(hp(window), hp(fb), n_fft, n_mels, hp(blob), nbytes), "sed_logmel_build_tables")
    return torch.from_numpy(blob.view(np.int32)).to(device)


@functools.lru_cache(maxsize=8)
def _tables(device_index, sr, n_fft, n_mels):
    """librosa's defaults (periodic Hann, Slaney bank), cached per (device, configuration)"""
    return build_tables(hann_periodic(n_fft), slaney_mel_basis(sr, n_fft, n_mels), torch.device("cuda", device_index))


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
        n_mels = int(tables[1])
    frames = 1 + y.numel() // hop
    out = torch.empty(frames, n_mels, device=y.device)
    inv = None
    if mean is not None:
        mean = mean.to(y.device).float().contiguous()
        inv = (1.0 / std.to(y.device).double()).float().contiguous()
    check(lib().sed_logmel(ptr(y), y.numel(), ptr(tables), tables.numel() * 4, ptr(mean), ptr(inv), ptr(out), n_fft, hop,
                           n_mels, {"constant": 0, "reflect": 1}[pad_mode], stream_ptr()), "sed_logmel")
    return out
