"""Diagnostic (GPU box): block 1 alone (conv -> batch statistics -> BN/ReLU/pool) on the float64 oracle's block-0 output with
and without dropout zeros in it: conv accuracy, accuracy of the batch mean / rstd, and ReLU-gate disagreements."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
from sed_crnn_amd import ops
from oracle import crnn_ref


def run(p, B=16, T=256, C=128):
    torch.manual_seed(0)
    ref = crnn_ref.SedNetRef(conv_channels=C, dropout=0.0, gru_hidden=32).double()
    ref.train()
    x, _ = crnn_ref.synthetic_batch(B, 1, 40, T, T // 8, seed=1234)
    g = torch.Generator().manual_seed(1)
    h0 = F.max_pool2d(torch.relu(ref.bns[0](ref.convs[0](x.double()))), (1, 2))
    if p > 0:
        h0 = h0 * ((torch.rand(h0.shape, generator=g) >= p).double() / (1 - p))
    h0 = h0.detach().float()                                  # the fp32 input both sides see
    cv64 = ref.convs[1](h0.double()).detach()
    mean64 = cv64.mean((0, 2, 3))
    var64 = cv64.var((0, 2, 3), unbiased=False)
    z64 = ref.bns[1](cv64)
    w, b = ref.convs[1].weight.detach().float(), ref.convs[1].bias.detach().float()
    wf, _ = ops.conv3x3_pack(w.cuda())
    xin = h0.permute(0, 3, 2, 1).contiguous().cuda()           # [B,T/2,F,C]
    y, stat = ops.conv3x3_fwd(xin, wf, b.cuda(), False)
    y64 = cv64.permute(0, 3, 2, 1)
    err = (y.cpu().double() - y64)
    print(f"--- input dropout {p}: conv max err {float(err.abs().max()):.2e}, rms {float(err.pow(2).mean().sqrt()):.2e}, "
          f"per-channel mean of the error / sigma: max {float((err.mean((0, 1, 2)).abs() / var64.sqrt()).max()):.2e}")
    cv32 = F.conv2d(h0, w, b, padding=1).permute(0, 3, 2, 1).double()
    e32 = cv32 - y64
    print(f"    torch fp32 conv: max err {float(e32.abs().max()):.2e}, rms {float(e32.pow(2).mean().sqrt()):.2e}, "
          f"mean of the error / sigma: max {float((e32.mean((0, 1, 2)).abs() / var64.sqrt()).max()):.2e}")
    n = B * (T // 2) * 40
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    mean, rstd, scale, shift = ops.bn_finalize_train(stat, n, torch.ones(C).cuda(), torch.zeros(C).cuda(), rm, rv)
    print(f"    batch mean err / sigma: max {float(((mean.cpu().double() - mean64).abs() / var64.sqrt()).max()):.2e}; "
          f"rstd rel err max {float((rstd.cpu().double() * (var64 + 1e-5).sqrt() - 1).abs().max()):.2e}; |mean|/sigma max {float((mean64.abs() / var64.sqrt()).max()):.2f}")
    zh = (y * scale + shift).cpu().double()
    zr = z64.detach().permute(0, 3, 2, 1)
    flips = ((zh > 0) != (zr > 0))
    print(f"    ReLU gate disagreements with float64: {int(flips.sum())} of {flips.numel()}; max |z| among them {float(zr[flips].abs().max()) if flips.any() else 0:.2e}")
    z32 = F.batch_norm(F.conv2d(h0, w, b, padding=1), None, None, training=True).permute(0, 3, 2, 1)
    f32 = ((z32 > 0) != (zr > 0))
    print(f"    torch fp32 disagreements: {int(f32.sum())}; max |z| among them {float(zr[f32].abs().max()) if f32.any() else 0:.2e}")


if __name__ == "__main__":
    run(0.0)
    run(0.5)
