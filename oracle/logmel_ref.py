"""numpy restatement of the log-mel extractor.  TEST INFRASTRUCTURE.

Follows /root/reference/feature.py:55-59::

    s = librosa.stft(y, n_fft=2048, hop_length=1024)
    power = |s|**2
    mel = librosa.filters.mel(sr=44100, n_fft=2048, n_mels=40)
    return log(mel @ power).T            # (frames, 40), natural log, no epsilon

PARITY UNPINNED: librosa (requirements.txt:4) is not importable in this image
and the reference holds no fixture for feature.py, so this restates librosa's
published defaults: periodic Hann window of n_fft samples, center=True with the
signal padded by n_fft//2 on both sides (``pad_mode`` is explicit because
librosa changed its default from 'reflect' to 'constant' in 0.10), frames
= 1 + len(y)//hop, Slaney mel scale (htk=False), fmin 0, fmax sr/2,
norm='slaney'.
"""
import numpy as np


def hann_periodic(n):
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float32)


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mel = f / f_sp
    min_log_hz, min_log_mel = 1000.0, 1000.0 / f_sp
    logstep = np.log(6.4) / 27.0
    big = f >= min_log_hz
    mel = np.where(big, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mel)
    return mel


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    f = f_sp * m
    min_log_hz, min_log_mel = 1000.0, 1000.0 / f_sp
    logstep = np.log(6.4) / 27.0
    big = m >= min_log_mel
    return np.where(big, min_log_hz * np.exp(logstep * (m - min_log_mel)), f)


def mel_filterbank(sr=44100, n_fft=2048, n_mels=40):
    """[n_mels, 1 + n_fft//2] float32, Slaney scale + Slaney area normalisation."""
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, fftfreqs.size), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def stft_power(y, n_fft=2048, hop=1024, pad_mode="constant"):
    """|STFT|^2, shape [frames, 1+n_fft//2], frames = 1 + len(y)//hop."""
    y = np.asarray(y, dtype=np.float32)
    yp = np.pad(y, n_fft // 2, mode=pad_mode)
    n_frames = 1 + (yp.size - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = yp[idx] * hann_periodic(n_fft)[None, :]
    spec = np.fft.rfft(frames.astype(np.float64), axis=1)
    return (spec.real ** 2 + spec.imag ** 2).astype(np.float32)


def mbe(y, sr=44100, n_fft=2048, hop=1024, n_mels=40, pad_mode="constant"):
    p = stft_power(y, n_fft, hop, pad_mode)                  # [frames, bins]
    m = mel_filterbank(sr, n_fft, n_mels)                    # [mels, bins]
    with np.errstate(divide="ignore"):
        return np.log(p @ m.T)                               # [frames, mels]


def standardize_fit(x):
    """sklearn StandardScaler.fit as feature.py:127-128 uses it -> (mean_, scale_) float64.

    Restates sklearn 1.7 (preprocessing/_data.py StandardScaler.partial_fit, utils/extmath.py _incremental_mean_and_var,
    _is_constant_feature, _handle_zeros_in_scale): float64 accumulators, population variance (ddof 0) from the centred
    second pass  (sum d^2 - (sum d)^2 / n) / n,  and scale 1 for a column whose variance is within the rounding bound of
    that algorithm,  var <= n*eps*var + (n*mean*eps)^2.  Pinned by tests/golden/g9_scaler.npz (generated with the
    installed scikit-learn by oracle/make_goldens.py)."""
    x64 = np.asarray(x, dtype=np.float64)
    n = x64.shape[0]
    mean = x64.sum(axis=0) / n
    d = x64 - mean
    var = ((d * d).sum(axis=0) - d.sum(axis=0) ** 2 / n) / n
    eps = np.finfo(np.float64).eps
    constant = var <= n * eps * var + (n * mean * eps) ** 2
    scale = np.sqrt(var)
    scale[constant] = 1.0
    return mean, scale


def standardize_apply(x, mean, scale):
    """StandardScaler.transform on a float32 matrix: sklearn subtracts and divides IN PLACE, so the float64 statistics
    are applied with a rounding to float32 after each of the two steps."""
    out = np.array(x, dtype=np.float32, copy=True)
    out -= mean
    out /= scale
    return out


def scaler_fixture_inputs(seed=77, n_train=1500, n_test=200, n_cols=40):
    """Seeded (RandomState: a frozen stream) train / test matrices of the g9 golden, float32, with the columns that make
    a scaler interesting: exactly constant (non-zero and zero), constant up to one float32 ulp, mean = 1e6 sigma, tiny
    sigma, and ordinary log-mel-like columns."""
    rs = np.random.RandomState(seed)
    sig = rs.uniform(0.5, 3.0, n_cols)
    mu = rs.uniform(-8.0, 4.0, n_cols)

    def draw(n):
        x = (rs.randn(n, n_cols) * sig + mu).astype(np.float32)
        x[:, 3] = 3.25
        x[:, 4] = 0.0
        x[:, 5] = np.float32(0.1)
        x[n // 2, 5] = np.nextafter(np.float32(0.1), np.float32(1.0))
        x[:, 6] = (1.0e4 + 1.0e-2 * rs.randn(n)).astype(np.float32)
        x[:, 7] = (1.0e-6 * rs.randn(n)).astype(np.float32)
        return x
    return draw(n_train), draw(n_test)
