#!/usr/bin/env python3
"""A/B of the backward schedule on one box: auxiliary stream on / off, per BASELINE config.  python tools/sched_ab.py 2 5"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sed_crnn_amd as sed
from sed_crnn_amd.trainer import FusedTrainStep
from tools.cfg_sweep import CONFIGS

for k in sys.argv[1:]:
    c = CONFIGS[int(k)]
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(c["B"], c["Cin"], c["F"], c["T"], generator=g).cuda()
    y = (torch.rand(c["B"], c["T"] // 8, 1, generator=g) > 0.8).float().cuda()
    m = sed.TimePooledCRNN(conv_channels=c["C"], dropout=0.5, in_channels=c["Cin"], n_mels=c["F"], gru_hidden=c["H"]).cuda()
    st = FusedTrainStep(m, lr=1e-3, loss="bce")
    for rep in range(2):
        for ov in (True, False):
            m.overlap_wgrad = ov
            for _ in range(3):
                st.step(x, y)
            torch.cuda.synchronize()
            n = 20 if int(k) != 5 else 5
            t0 = time.perf_counter()
            for _ in range(n):
                st.step(x, y)
            torch.cuda.synchronize()
            print(f"config {k} aux stream {'on ' if ov else 'off'}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", flush=True)
    del m, st, x, y
    torch.cuda.empty_cache()
