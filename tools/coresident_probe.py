#!/usr/bin/env python3
"""Does a streaming kernel make progress beside the persistent MFMA weight gradient?  Launches both at once on two streams
and compares the streaming kernel's elapsed time with its stand-alone time (no progress => elapsed ~ wgrad + alone)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sed_crnn_amd import ops

B, T = 128, 128
x = torch.randn(B, T, 40, 128, device="cuda")
dy = torch.randn(B, T, 40, 128, device="cuda")
big = torch.randn(64 * 1024 * 1024, device="cuda")          # 256 MB: read + write = 512 MB
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def ev():
    return torch.cuda.Event(enable_timing=True)


def alone(fn, s):
    a, b = ev(), ev()
    with torch.cuda.stream(s):
        fn(); a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


wg = lambda: ops.conv3x3_wgrad(x, dy, False)
st = lambda: big.mul_(1.0001)
print(f"alone: wgrad {alone(wg, sA):.3f} ms, streaming kernel (512 MB) {alone(st, sB):.3f} ms")
for rep in range(3):
    g = ev(); a1 = ev(); b0 = ev(); b1 = ev()
    torch.cuda.synchronize()
    g.record()
    sA.wait_event(g); sB.wait_event(g)
    with torch.cuda.stream(sA):
        wg(); a1.record()
    with torch.cuda.stream(sB):
        b0.record(); st(); b1.record()
    torch.cuda.synchronize()
    print(f"together: wgrad ends at {g.elapsed_time(a1):.3f} ms, streaming kernel starts {g.elapsed_time(b0):.3f} ends {g.elapsed_time(b1):.3f} ms")
