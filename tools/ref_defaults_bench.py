#!/usr/bin/env python3
"""Step time of the reference's own default nets on the drop-in path (documentation aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sed_crnn_amd as sed


def bench(name, m, crit, B=128, T=64):
    m = m.cuda()
    x = torch.randn(B, 1, 40, T).cuda()
    y = (torch.rand(B, T // 8, 1) > 0.8).float().cuda()
    opt = sed.FusedAdam(m.parameters(), lr=1e-3)

    def step():
        m.train(); opt.zero_grad(); loss = crit(m(x), y); loss.backward(); opt.step()
    for _ in range(5):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"{name:55s} {dt*1e3:7.3f} ms/step  {B*T/dt/1e6:6.2f} M frames/s")


torch.manual_seed(0)
bench("sed.py default: TimePooledCRNN() B128 T64 (C128,H32)", sed.TimePooledCRNN(), sed.BCEWithLogitsLoss())
bench("crnn_lightning default: C16, GRU 16/8, focal, B128 T64", sed.LightningTimePooledCRNN(), sed.FocalBCELoss())
bench("config 1 shape: TimePooledCRNN() B16 T256", sed.TimePooledCRNN(), sed.BCEWithLogitsLoss(), B=16, T=256)


def bench_fused(name, m, loss, B, T, graph):
    from sed_crnn_amd.trainer import FusedTrainStep
    m = m.cuda()
    x = torch.randn(B, 1, 40, T).cuda()
    y = (torch.rand(B, T // 8, 1) > 0.8).float().cuda()
    st = FusedTrainStep(m, lr=1e-3, loss=loss, graph=graph)
    for _ in range(5):
        st.step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        st.step(x, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(f"{name:55s} {dt*1e3:7.3f} ms/step  {B*T/dt/1e6:6.2f} M frames/s")


for graph in (False, True):
    tag = "hipGraph replay" if graph else "fused eager    "
    bench_fused(f"[{tag}] sed.py default B128 T64", sed.TimePooledCRNN(), "bce", 128, 64, graph)
    bench_fused(f"[{tag}] lightning default B128 T64", sed.LightningTimePooledCRNN(), "focal", 128, 64, graph)
    bench_fused(f"[{tag}] config 1 B16 T256", sed.TimePooledCRNN(), "bce", 16, 256, graph)
    bench_fused(f"[{tag}] config 2 B128 T256 H128", sed.TimePooledCRNN(gru_hidden=128), "bce", 128, 256, graph)
