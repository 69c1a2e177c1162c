#!/usr/bin/env python3
"""Fit-step time of BASELINE configs 2, 3 and 5 at their full per-GPU sizes on one MI355X, plus two size-independent
checks on the big shapes (eval forward of the whole batch == eval forward of its chunks; finite loss and gradients).
python tools/cfg_sweep.py [--steps N] [--configs 2,3,5]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sed_crnn_amd as sed
from sed_crnn_amd import _lib
from sed_crnn_amd.trainer import FusedTrainStep

# "A" = the README-figure topology: mel pooling 5/2/2, no time pooling -> 256 serial GRU steps, 6 classes
CONFIGS = {
    "A": dict(B=128, Cin=1, F=40, T=256, C=128, H=32, figure=True),
    2: dict(B=128, Cin=1, F=40, T=256, C=128, H=128),
    3: dict(B=128, Cin=2, F=40, T=256, C=128, H=128),
    5: dict(B=128, Cin=4, F=128, T=512, C=128, H=256),
}


def flops_per_frame_train(c):
    T, F, C, H, Cin = c["T"], c["F"], c["C"], c["H"], c["Cin"]
    if c.get("figure"):
        conv = 2 * 9 * C * T * (Cin * 40 + C * 8 + C * 4)
        gru = 2 * 2 * T * 3 * H * (C * 2 + H) + 2 * 2 * T * 3 * H * (2 * H + H)
        return 3.0 * (conv + gru) / T
    conv = 2 * 9 * C * F * (Cin * T + C * T // 2 + C * T // 4)
    Tp = T // 8
    gru = 2 * 2 * Tp * 3 * H * (C * F + H) + 2 * 2 * Tp * 3 * H * (2 * H + H)
    return 3.0 * (conv + gru) / T


def run(k, steps, breakdown=False):
    c = CONFIGS[k]
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(c["B"], c["Cin"], c["F"], c["T"], generator=g).cuda()
    if c.get("figure"):
        m = sed.get_model(in_channels=1, n_mels=40, seq_len=c["T"], n_classes=6, conv_channels=c["C"],
                          pools=[(5, 1), (2, 1), (2, 1)], rnn_hidden=[c["H"], c["H"]], fc=[16, 6]).cuda()
        y = (torch.rand(c["B"], c["T"], 6, generator=g) > 0.8).float().cuda()
    else:
        m = sed.TimePooledCRNN(conv_channels=c["C"], dropout=0.5, in_channels=c["Cin"], n_mels=c["F"], gru_hidden=c["H"]).cuda()
        y = (torch.rand(c["B"], c["T"] // 8, 1, generator=g) > 0.8).float().cuda()
    st = FusedTrainStep(m, lr=1e-3, loss="bce")
    for _ in range(3):
        loss, _ = st.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = st.step(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fr = c["B"] * c["T"] / dt
    finite = bool(torch.isfinite(loss).all() and torch.isfinite(m.flat_grads()).all() and torch.isfinite(m.flat_parameters()).all())
    m.eval()
    with torch.no_grad():
        whole = m(x)
        parts = torch.cat([m(x[i:i + 32]) for i in range(0, c["B"], 32)])
    torch.cuda.synchronize()
    chunk_err = float((whole - parts).abs().max())
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(steps):
            m(x)
    torch.cuda.synchronize()
    de = (time.perf_counter() - t0) / steps
    print(f"config {k} {c}: fit {dt*1e3:8.3f} ms/step  {fr/1e6:6.3f} M frames/s  "
          f"{fr*flops_per_frame_train(c)/1e12:6.1f} TFLOP/s algorithmic | eval fwd {de*1e3:7.3f} ms  {c['B']*c['T']/de/1e6:6.2f} M frames/s | "
          f"loss {loss.item():.5f} finite={finite} | eval whole-vs-chunks max|d| {chunk_err:.2e} | "
          f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    assert finite and chunk_err < 1e-4
    if breakdown:
        import ctypes as C
        lib = _lib.lib()
        lib.sed_prof_enable(0xFFFF)
        for _ in range(2):
            st.step(x, y)
        torch.cuda.synchronize()
        for t in range(11):
            ms, n, u = C.c_double(), C.c_long(), C.c_double()
            lib.sed_prof_read(t, C.byref(ms), C.byref(n), C.byref(u))
            if n.value:
                print(f"    {lib.sed_prof_tag_name(t).decode():24s} {ms.value/2:9.3f} ms/step {n.value//2:4d} launches  "
                      f"{u.value/(ms.value*1e-3)/1e12:8.2f} T(units)/s", flush=True)
        lib.sed_prof_enable(0)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--configs", default="2,3,5")
    ap.add_argument("--breakdown", action="store_true")
    a = ap.parse_args()
    for k in [s if s == "A" else int(s) for s in a.configs.split(",")]:
        run(k, a.steps, a.breakdown)
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
