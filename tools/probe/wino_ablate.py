#!/usr/bin/env python3
"""Ablation timings of the Winograd forward kernel (needs a -DWN_ABLATION build: SED_CRNN_LIB=tools/probe/libsedcrnn_abl.so).
Results of the ablated launches are wrong on purpose; only the times count."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sed_crnn_amd import ops
from sed_crnn_amd._lib import lib, ptr
from tools.kbench import timeit

B, T = 128, 128
x = torch.randn(B, T, 40, 128, device="cuda")
w = torch.randn(128, 128, 3, 3, device="cuda") * 0.03
bias = torch.randn(128, device="cuda")
uf, _ = ops.conv3x3_wino_pack(w)
y = torch.empty(B, T, 40, 128, device="cuda")
stat = torch.empty(lib().sed_conv3x3_wino_rows(B, 128, 40, T, 128), 2, 128, device="cuda")
from sed_crnn_amd._lib import stream_ptr
run = lambda: lib().sed_conv3x3_wino_fwd(ptr(x), ptr(uf), ptr(bias), ptr(y), ptr(stat), B, 128, 40, T, 128, stream_ptr())
import ctypes as C
lib().sed_conv3x3_wino_ablate.argtypes = [C.c_int]
for name, mask in (("full", 0), ("no transform arithmetic", 1), ("no weight-fragment loads", 2), ("no LDS operand reads (+ no arithmetic)", 4),
                   ("no patch DMA in the loop", 8), ("MFMA + barriers only", 15)):
    lib().sed_conv3x3_wino_ablate(mask)
    ms = timeit(run, 20)
    buf = torch.zeros(4, dtype=torch.int64, device="cuda")
    lib().sed_conv3x3_wino_phase_ticks(ptr(buf))
    run()
    torch.cuda.synchronize()
    lib().sed_conv3x3_wino_phase_ticks(None)
    t = buf.cpu().tolist()
    n = max(t[3], 1)
    print(f"{name:42s} {ms*1e3:7.1f} us   per workgroup: prologue {t[0]/n/100:.2f}  loop {t[1]/n/100:.2f}  epilogue {t[2]/n/100:.2f} us")
lib().sed_conv3x3_wino_ablate(0)
