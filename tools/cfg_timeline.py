#!/usr/bin/env python3
"""A few fit steps of one BASELINE config for a rocprofv3 kernel trace (see tools/timeline.py):
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/cfg_timeline.py 5"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sed_crnn_amd as sed
from sed_crnn_amd.trainer import FusedTrainStep
from tools.cfg_sweep import CONFIGS

k = sys.argv[1] if len(sys.argv) > 1 else "5"
c = CONFIGS[k if k == "A" else int(k)]
torch.manual_seed(0)
g = torch.Generator().manual_seed(1234)
x = torch.randn(c["B"], c["Cin"], c["F"], c["T"], generator=g).cuda()
y = (torch.rand(c["B"], c["T"] // 8, 1, generator=g) > 0.8).float().cuda()
m = sed.TimePooledCRNN(conv_channels=c["C"], dropout=0.5, in_channels=c["Cin"], n_mels=c["F"], gru_hidden=c["H"]).cuda()
st = FusedTrainStep(m, lr=1e-3, loss="bce")
for _ in range(6):
    st.step(x, y)
torch.cuda.synchronize()
