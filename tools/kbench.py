#!/usr/bin/env python3
"""Single-kernel timings on the GPU (tuning aid): python tools/kbench.py [conv|wgrad|gemm|bn|all] [--iters N]
Shapes are the BASELINE config-2 launches.  Times come from torch events on the launch stream."""
import argparse
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sed_crnn_amd import ops


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = "cuda"
    B = 128
    if a.what in ("conv", "all"):
        for T in (128, 64):
            x = torch.randn(B, T, 40, 128, device=dev)
            w = torch.randn(128, 128, 3, 3, device=dev) * 0.03
            bias = torch.randn(128, device=dev)
            wf, wd = ops.conv3x3_pack(w)
            ms = timeit(lambda: ops.conv3x3_fwd(x, wf, bias, False), a.iters)
            fl = 2 * 9 * 128 * 128 * B * T * 40
            print(f"conv3x3_mfma_fwd  B{B} T{T}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s")
            uf, _ = ops.conv3x3_wino_pack(w)
            ms = timeit(lambda: ops.conv3x3_wino_fwd(x, uf, bias, 128), a.iters)
            print(f"conv3x3 winograd F(2x2,3x3)  B{B} T{T}: {ms:.3f} ms  {fl/ms/1e9:.1f} algorithmic TFLOP/s ({fl/2.25/ms/1e9:.1f} executed)")
            wfb, _ = ops.conv3x3_pack(w, mode=1)
            ms = timeit(lambda: ops.conv3x3_fwd(x, wfb, bias, False, mode=1), a.iters)
            print(f"conv3x3 bf16x3 (experiment)  B{B} T{T}: {ms:.3f} ms  {fl/ms/1e9:.1f} fp32-equivalent TFLOP/s")
    if a.what in ("wgrad", "all"):
        for T in (128, 64):
            x = torch.randn(B, T, 40, 128, device=dev)
            dy = torch.randn(B, T, 40, 128, device=dev)
            ms = timeit(lambda: ops.conv3x3_wgrad(x, dy, False), a.iters)
            fl = 2 * 9 * 128 * 128 * B * T * 40
            print(f"conv3x3_mfma_wgrad B{B} T{T}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s")
            ms = timeit(lambda: ops.conv3x3_wgrad(x, dy, False, mode=1), a.iters)
            print(f"conv3x3 wgrad bf16x3 (experiment) B{B} T{T}: {ms:.3f} ms  {fl/ms/1e9:.1f} fp32-equivalent TFLOP/s")
    if a.what in ("gemm", "all"):
        M, K, H = 4096, 5120, 128
        X = torch.randn(M, K, device=dev)
        W = torch.randn(6 * H, K, device=dev) * 0.01
        dgi = torch.randn(M, 6 * H, device=dev)
        out = torch.empty(M, 6 * H, device=dev)
        dW = torch.empty(6 * H, K, device=dev)
        dX = torch.empty(M, K, device=dev)
        cases = [("fwd  X[4096,5120] @ W^T[5120,768]", lambda: ops.gemm(X, W.t(), out=out), 2 * M * K * 6 * H),
                 ("dW   dgi^T[768,4096] @ X[4096,5120]", lambda: ops.gemm(dgi.t(), X, out=dW), 2 * M * K * 6 * H),
                 ("dW   (weight-gradient entry: two K-slices)", lambda: ops.gemm_ws(dgi.t(), X, out=dW, wgrad=True), 2 * M * K * 6 * H),
                 ("dX   dgi[4096,768] @ W[768,5120]", lambda: ops.gemm(dgi, W, out=dX), 2 * M * K * 6 * H)]
        X1 = torch.randn(M, 256, device=dev)
        W1 = torch.randn(768, 256, device=dev)
        o1 = torch.empty(M, 768, device=dev)
        dW1 = torch.empty(768, 256, device=dev)
        hp = torch.randn(M, 128, device=dev)
        dWh = torch.empty(384, 128, device=dev)
        cases += [("fwd1 X[4096,256] @ W^T[256,768]", lambda: ops.gemm(X1, W1.t(), out=o1), 2 * M * 256 * 768),
                  ("dW1  dgi^T[768,4096] @ X[4096,256]", lambda: ops.gemm_ws(dgi.t(), X1, out=dW1), 2 * M * 256 * 768),
                  ("dWhh dgh^T[384,4096] @ h[4096,128]", lambda: ops.gemm_ws(dgi[:, :384].t(), hp, out=dWh), 2 * M * 384 * 128)]
        for name, fn, fl in cases:
            ms = timeit(fn, a.iters)
            print(f"gemm {name}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s")
    if a.what in ("gru", "all"):
        for H, T in ((128, 32), (256, 64)):
            gi = torch.randn(B, T, 2, 3 * H, device=dev) * 0.5
            whh = [torch.randn(3 * H, H, device=dev) / H ** 0.5 for _ in range(2)]
            bhh = [torch.randn(3 * H, device=dev) * 0.1 for _ in range(2)]
            ms = timeit(lambda: ops.gru_seq_fwd(gi, whh, bhh), a.iters)
            out, saved = ops.gru_seq_fwd(gi, whh, bhh)
            dout = torch.randn_like(out)
            ms2 = timeit(lambda: ops.gru_seq_bwd(dout, saved, whh, want_bias=True), a.iters)
            print(f"gru_seq H{H} T{T}: fwd {ms:.3f} ms ({ms / T * 1e3:.2f} us/step)  bwd {ms2:.3f} ms ({ms2 / T * 1e3:.2f} us/step)")
    if a.what in ("bn", "all"):
        for T in (256, 128, 64):
            y = torch.randn(B, T, 40, 128, device=dev)
            sc, sh = torch.ones(128, device=dev), torch.zeros(128, device=dev)
            mean, rstd = torch.zeros(128, device=dev), torch.ones(128, device=dev)
            ms = timeit(lambda: ops.bn_relu_pool_drop_fwd(y, sc, sh, 1, 2, drop_p=0.5, seed=1), a.iters)
            by = 4 * y.numel() * 1.5
            print(f"bn_relu_pool_drop_fwd T{T}: {ms:.3f} ms  {by/ms/1e9:.2f} TB/s")
            dout = torch.randn(B, T // 2, 40, 128, device=dev)
            ms = timeit(lambda: ops.bn_relu_pool_drop_bwd(y, dout, sc, sh, mean, rstd, 1, 2, drop_p=0.5, seed=1), a.iters)
            print(f"bn bwd (reduce+finalize+apply+rows) T{T}: {ms:.3f} ms  {4*y.numel()*4.0/ms/1e9:.2f} TB/s (alg 4.0 x out)")


if __name__ == "__main__":
    main()
