import sys, os, torch
sys.path.insert(0, '.')
from sed_crnn_amd import ops
dev = 'cuda'
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
B = 128
res = []
for T in (128, 64):
    x = torch.randn(B, T, 40, 128, device=dev)
    w = torch.randn(128, 128, 3, 3, device=dev) * 0.03
    bias = torch.randn(128, device=dev)
    wf, wd = ops.conv3x3_pack(w)
    ms = timeit(lambda: ops.conv3x3_fwd(x, wf, bias, False))
    fl = 2 * 9 * 128 * 128 * B * T * 40
    res.append(f"T{T}: {ms:.3f} ms {fl/ms/1e9:.1f} TF")
print(f"ABL={sys.argv[1]:>3}: " + "   ".join(res), flush=True)
