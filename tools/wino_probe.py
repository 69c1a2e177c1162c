#!/usr/bin/env python3
"""Where a Winograd conv launch spends its time (tuning aid): per-workgroup prologue / main loop / epilogue from s_memrealtime
stamps (sed_conv3x3_wino_phase_ticks), next to the launch time.  python tools/wino_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sed_crnn_amd import ops
from sed_crnn_amd._lib import lib, ptr
from tools.kbench import timeit

B = 128
for T in (128, 64):
    x = torch.randn(B, T, 40, 128, device="cuda")
    w = torch.randn(128, 128, 3, 3, device="cuda") * 0.03
    bias = torch.randn(128, device="cuda")
    uf, _ = ops.conv3x3_wino_pack(w)
    ms = timeit(lambda: ops.conv3x3_wino_fwd(x, uf, bias, 128), 20)
    buf = torch.zeros(4, dtype=torch.int64, device="cuda")
    lib().sed_conv3x3_wino_phase_ticks(ptr(buf))
    ops.conv3x3_wino_fwd(x, uf, bias, 128)
    torch.cuda.synchronize()
    lib().sed_conv3x3_wino_phase_ticks(None)
    t = buf.cpu().tolist()
    n = max(t[3], 1)
    us = [v / n / 100.0 for v in t[:3]]
    per_cu = n / 256.0
    print(f"T={T}: launch {ms*1e3:.0f} us | per workgroup: prologue {us[0]:.2f} us, main loop {us[1]:.2f} us (MFMA-bound 27.3 at 2.4 GHz), "
          f"epilogue {us[2]:.2f} us | {n} workgroups = {per_cu:.1f} per CU -> sum {per_cu*sum(us):.0f} us")
