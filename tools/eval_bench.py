#!/usr/bin/env python3
"""Eval-mode forward (run_epoch with optim=None, sed.py:128-141) of BASELINE config 2 on a resident batch:
python tools/eval_bench.py [--reps 30]   (under rocprofv3 --kernel-trace --stats: the per-kernel split of the inference path)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sed_crnn_amd as sed

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--T", type=int, default=256)
a = ap.parse_args()
torch.manual_seed(0)
m = sed.TimePooledCRNN(conv_channels=128, dropout=0.5, gru_hidden=128).cuda().eval()
x = torch.randn(a.B, 1, 40, a.T).cuda()
with torch.no_grad():
    for _ in range(5):
        m(x)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.reps + 1)]
    ev[0].record()
    for i in range(a.reps):
        m(x)
        ev[i + 1].record()
    torch.cuda.synchronize()
ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.reps))
med = ms[len(ms) // 2]
print(f"eval forward B={a.B} T={a.T}: median {med:.3f} ms  ({a.B * a.T / med / 1e3:.2f} M frames/s), min {ms[0]:.3f}")
