"""Diagnostic (GPU box): intermediates of the HIP plan against the float64 oracle, dropout in block 0 only."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
import sed_crnn_amd as sed
from sed_crnn_amd import ops
from oracle import crnn_ref
from grad_probe2 import masks_of


def rel(a, b):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def run(drops, overlap, B=16, T=256, C=128, H=32):
    torch.manual_seed(0)
    ref = crnn_ref.SedNetRef(conv_channels=C, dropout=0.0, gru_hidden=H)
    m = sed.TimePooledCRNN(conv_channels=C, dropout=0.5, gru_hidden=H)
    m.drops = list(drops)
    m.overlap_wgrad = overlap
    m.load_state_dict(ref.state_dict())
    m.cuda().train()
    x, y = crnn_ref.synthetic_batch(B, 1, 40, T, T // 8, seed=1234)
    logits = m._run_forward(x.cuda(), training=True)
    _, dlogits, _ = ops.loss_fwd_bwd(logits, y.cuda(), "bce", 0.25, 2.0, "mean")
    masks = masks_of(m, B, 40, T, drops)
    ref64 = crnn_ref.SedNetRef(conv_channels=C, dropout=0.0, gru_hidden=H).double()
    ref64.load_state_dict({k: v.double() if v.dtype.is_floating_point else v for k, v in ref.state_dict().items()})
    ref64.train()
    keep = {}
    h = x.double()
    for l, (conv, bn, mk) in enumerate(zip(ref64.convs, ref64.bns, masks)):
        cv = conv(h)
        cv.retain_grad()
        h = F.max_pool2d(torch.relu(bn(cv)), (1, 2)) * mk.double()
        h.retain_grad()
        keep[l] = (cv, h)
    b, c, f, t = h.shape
    hh, _ = ref64.gru(h.permute(0, 3, 1, 2).reshape(b, t, c * f))
    crnn_ref.bce_logits(ref64.fc(hh), y.double()).backward()
    cl = lambda t_: t_.permute(0, 3, 2, 1)                     # NCHW [B,C,F,T] -> channels-last [B,T,F,C]
    print(f"=== drops {drops} overlap {overlap}")
    print("fwd  pooled[0]", rel(m.workspace_view("pooled", 0), cl(keep[0][1])), " conv_out[1]", rel(m.workspace_view("conv_out", 1), cl(keep[1][0])),
          " pooled[1]", rel(m.workspace_view("pooled", 1), cl(keep[1][1])), " conv_out[2]", rel(m.workspace_view("conv_out", 2), cl(keep[2][0])))
    m._run_backward(None, dlogits, 0, 1)
    torch.cuda.synchronize()
    m._run_backward(None, dlogits, 1, 2)                       # BN(2) joined, dgrad(2), BN(1)
    torch.cuda.synchronize()
    n1 = keep[1][1].numel()
    print("bwd after stage 1: dconv[2]", rel(m.workspace_view("dconv", 2), cl(keep[2][0].grad)),
          " d pooled[1] (grad_act)", rel(m.workspace_view("grad_act")[:n1], cl(keep[1][1].grad)),
          " dconv[1]", rel(m.workspace_view("dconv", 1), cl(keep[1][0].grad)))
    sums = m.workspace_view("bn_sums_bwd").cpu().double()
    print("   sum_g(1) vs float64 dbeta", rel(sums[:C], ref64.bns[1].bias.grad), " sum_gx(1) vs dgamma", rel(sums[C:2 * C], ref64.bns[1].weight.grad))
    e = (m.workspace_view("grad_act")[:n1].cpu().double().reshape(cl(keep[1][1].grad).shape) - cl(keep[1][1].grad))
    print("   d pooled[1] error: max", float(e.abs().max()), " truth max", float(keep[1][1].grad.abs().max()),
          " per-(b) error norms", [f"{float(e[i].norm()):.1e}" for i in range(0, B, 4)],
          " per-t error norms", [f"{float(e[:, i].norm()):.1e}" for i in range(0, e.shape[1], 8)],
          " per-f error norms", [f"{float(e[:, :, i].norm()):.1e}" for i in range(0, 40, 5)])
    m._run_backward(None, dlogits, 2, 4)
    torch.cuda.synchronize()


if __name__ == "__main__":
    run([0.5, 0.0, 0.0], True)
    run([0.5, 0.0, 0.0], False)
    run([0.0, 0.0, 0.0], True)
