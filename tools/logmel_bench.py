#!/usr/bin/env python3
"""log-mel kernel: parity against the numpy restatement on a few signal lengths, then the rate on one hour of 44.1 kHz audio
(635 MB of PCM, far beyond the 256 MiB Infinity Cache).  python tools/logmel_bench.py [--seconds 3600]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import logmel_ref
from sed_crnn_amd import feature

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=int, default=3600)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
rng = np.random.RandomState(0)
for n in (1024 * 21 + 300, 1024 * 2, 5000, 44100):
    tt = np.arange(n) / 44100.0
    y = (0.3 * np.sin(2 * np.pi * 440 * tt) + 0.1 * np.sin(2 * np.pi * 3000 * tt) + 0.05 * rng.randn(n)).astype(np.float32)
    for pad in ("constant", "reflect"):
        ref = logmel_ref.mbe(y, pad_mode=pad)
        out = feature.mbe(torch.from_numpy(y).cuda(), pad_mode=pad).cpu().numpy()
        err = np.abs(out - ref)
        print(f"n={n:6d} pad={pad:8s} frames={out.shape[0]:3d} max|d|={err.max():.2e} (worst rel {np.max(err/np.abs(ref)):.2e})", flush=True)
        np.testing.assert_allclose(out, ref, atol=1e-3, rtol=1e-4)
n = 44100 * a.seconds
y = torch.randn(n, device="cuda") * 0.1
feature.mbe(y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    o = feature.mbe(y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
frames = o.shape[0]
print(f"{a.seconds} s of audio: {frames} frames in {dt*1e3:.3f} ms = {frames/dt/1e6:.1f} M frames/s = "
      f"{frames*(1024+40)*4/dt/1e12:.3f} TB/s algorithmic ({frames*(1024+40)*4/dt/8e12*100:.1f} % of 8 TB/s)")
