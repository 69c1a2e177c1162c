#!/usr/bin/env python3
"""conv kernel scaling probe: fixed per-block overhead vs per-chunk cost (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sed_crnn_amd import ops
from tools.kbench import timeit

B, T = 128, 64
for Cin in (32, 64, 128, 256, 512):
    x = torch.randn(B, T, 40, Cin, device="cuda")
    w = torch.randn(128, Cin, 3, 3, device="cuda") * 0.03
    wf, _ = ops.conv3x3_pack(w)
    ms = timeit(lambda: ops.conv3x3_fwd(x, wf, None, False, want_stats=False), 10)
    fl = 2 * 9 * Cin * 128 * B * T * 40
    print(f"Cin={Cin:4d}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s   per-chunk {ms/(Cin/32)*1e3:.1f} us")
