#!/usr/bin/env python3
"""Inference latency of the config-2 net for small batches (serving-style calls): python tools/latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sed_crnn_amd as sed

torch.manual_seed(0)
m = sed.TimePooledCRNN(conv_channels=128, dropout=0.5, gru_hidden=128).cuda().eval()
for B, T in [(1, 256), (1, 2048), (4, 256), (16, 256), (128, 256)]:
    x = torch.randn(B, 1, 40, T).cuda()
    with torch.no_grad():
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            m(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"eval forward B={B:4d} T={T:5d}: {dt*1e3:8.3f} ms   {B*T/dt/1e6:7.2f} M frames/s", flush=True)
