#!/usr/bin/env python3
"""End-to-end epoch of the reference workflow (sed.py:144-202) on a synthetic fold: device-resident window sampler with
1:1 positive/negative balancing (+ SpecAugment on request), drop-in autograd step of the reference-default net
(C=128, H=32, windows of 64 frames, batch 128), validation pass, device-side 1-second scores.  Prints the whole-epoch
rate next to the bare step rate of the same net on a resident batch: the gap is what sampling, label pooling, metric
accumulation and the host loop cost.
   python tools/epoch_bench.py [--frames 400000] [--epochs 3] [--augment]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sed_crnn_amd as sed
from sed_crnn_amd import data


def synthetic_fold(n_frames, seed):
    rng = np.random.default_rng(seed)
    mel = rng.standard_normal((n_frames, 40)).astype(np.float32)
    lab = np.zeros((n_frames, 1), np.float32)
    for s in rng.integers(0, n_frames - 8, size=n_frames // 400):        # a 2-8 frame hit every ~400 frames, like the fork's data
        lab[s:s + rng.integers(2, 9)] = 1
    return mel, lab


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=400000)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--augment", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    tr = data.HitWindowSet(*synthetic_fold(a.frames, 1), augment=a.augment, seed=1)
    va = data.HitWindowSet(*synthetic_fold(a.frames // 4, 2), seed=2)
    ltr, lva = data.GpuWindowLoader(tr, 128, shuffle=True), data.GpuWindowLoader(va, 128, shuffle=False)
    m = sed.TimePooledCRNN().cuda()                                  # sed.py defaults: C=128, dropout 0.5, GRU 2x32
    opt = sed.FusedAdam(m.parameters(), lr=1e-3)
    crit = sed.BCEWithLogitsLoss()
    print(f"train fold {a.frames} frames -> {len(tr)} windows/epoch ({len(ltr)} batches of 128x64), "
          f"validation {len(va)} windows; augment={a.augment}")
    for ep in range(a.epochs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_tr = sed.run_epoch_device(m, ltr, crit, opt)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        t_va = sed.run_epoch_device(m, lva, crit)
        s = t_va.scores(5)                                           # one small D2H of 17 counters
        t2 = time.perf_counter()
        ftr, fva = len(tr) * 64, len(va) * 64
        print(f"epoch {ep}: train {1e3 * (t1 - t0):8.1f} ms = {ftr / (t1 - t0) / 1e6:5.2f} M frames/s | "
              f"val + scores {1e3 * (t2 - t1):7.1f} ms = {fva / (t2 - t1) / 1e6:5.2f} M frames/s | "
              f"loss {t_tr.mean_loss():.4f} val ER {s['er_overall_1sec']:.3f} F1 {s['f1_overall_1sec']:.3f}")
    # the bare step on a resident batch, same drop-in path
    x, y = tr.batch(np.arange(128))
    m.train()
    for _ in range(5):
        opt.zero_grad(); crit(m(x), y).backward(); opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        opt.zero_grad(); crit(m(x), y).backward(); opt.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(f"bare drop-in step on a resident batch: {1e3 * dt:.3f} ms = {128 * 64 / dt / 1e6:.2f} M frames/s")


if __name__ == "__main__":
    main()
