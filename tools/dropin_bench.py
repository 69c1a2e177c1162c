#!/usr/bin/env python3
"""Step time of the drop-in autograd path (model(x); loss.backward(); optim.step()) vs the fused trainer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sed_crnn_amd as sed
from sed_crnn_amd.trainer import FusedTrainStep

torch.manual_seed(0)
m = sed.TimePooledCRNN(conv_channels=128, dropout=0.5, gru_hidden=128).cuda()
x = torch.randn(128, 1, 40, 256).cuda()
y = (torch.rand(128, 32, 1) > 0.8).float().cuda()


def run(name, step, n=20):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()          # host enqueue time
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:40s} {1e3*(t2-t0)/n:7.3f} ms/step  (host enqueue {1e3*(t1-t0)/n:6.3f} ms)")


crit = sed.BCEWithLogitsLoss()
for label, opt in (("drop-in: FusedAdam(model.parameters())", sed.FusedAdam(m.parameters(), lr=1e-3)),
                   ("drop-in: FusedAdam(...).attach(model)", sed.FusedAdam(m.parameters(), lr=1e-3).attach(m)),
                   ("drop-in: torch.optim.Adam + torch BCE", torch.optim.Adam(m.parameters(), lr=1e-3))):
    c = torch.nn.BCEWithLogitsLoss() if "torch BCE" in label else crit

    def step():
        m.train()
        opt.zero_grad()
        loss = c(m(x), y)
        loss.backward()
        opt.step()
    run(label, step)
ts = FusedTrainStep(m, lr=1e-3)
run("fused trainer (bench.py path)", lambda: ts.step(x, y))
m.eval()
with torch.no_grad():
    run("eval forward only", lambda: m(x))
