"""Diagnostic (GPU box): BN(1) backward sums recomputed three ways from the HIP plan's OWN workspace tensors."""
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


def run(drops, B=16, T=256, C=128, H=32):
    torch.manual_seed(0)
    ref = crnn_ref.SedNetRef(conv_channels=C, dropout=0.0, gru_hidden=H)
    m = sed.TimePooledCRNN(conv_channels=C, dropout=0.5, gru_hidden=H)
    m.drops = list(drops)
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
        z = bn(cv)
        z.retain_grad()
        h = F.max_pool2d(torch.relu(z), (1, 2)) * mk.double()
        h.retain_grad()
        keep[l] = (cv, h, z)
    b, c, f, t = h.shape
    hh, _ = ref64.gru(h.permute(0, 3, 1, 2).reshape(b, t, c * f))
    crnn_ref.bce_logits(ref64.fc(hh), y.double()).backward()
    m._run_backward(None, dlogits, 0, 2)
    torch.cuda.synchronize()
    T1 = T // 2
    n1 = B * (T1 // 2) * 40 * C
    yv = m.workspace_view("conv_out", 1).reshape(B, T1, 40, C).clone()
    dout = m.workspace_view("grad_act")[:n1].reshape(B, T1 // 2, 40, C).clone()
    sc, sh, mu, rs = (m.workspace_view(k, 1).clone() for k in ("scale", "shift", "mean", "rstd"))
    sums = m.workspace_view("bn_sums_bwd").clone().cpu().double()
    truth_b, truth_g = ref64.bns[1].bias.grad, ref64.bns[1].weight.grad
    print(f"=== drops {drops}")
    print("in-net sum_g", rel(sums[:C], truth_b), " sum_gx", rel(sums[C:], truth_g))
    dy, dgamma, dbeta, _ = ops.bn_relu_pool_drop_bwd(yv, dout, sc, sh, mu, rs, 1, 2, drop_p=drops[1], seed=0)
    print("stand-alone kernel on the plan's own tensors: dbeta", rel(dbeta, truth_b), " dgamma", rel(dgamma, truth_g))
    # emulate in double on the plan's own tensors
    yd, dd = yv.cpu().double(), dout.cpu().double()
    z = yd * sc.cpu().double() + sh.cpu().double()                     # [B,T1,F,C]
    zw = z.reshape(B, T1 // 2, 2, 40, C)
    first = zw[:, :, 0] >= zw[:, :, 1]
    best = torch.where(first, zw[:, :, 0], zw[:, :, 1])
    g = dd * (best > 0)
    print("double emulation on the plan's own tensors: dbeta", rel(g.sum((0, 1, 2)), truth_b))
    # the oracle's own dz summed (sanity) and the disagreement pattern
    dz = keep[1][2].grad.permute(0, 3, 2, 1)                            # [B,T1,F,C]
    print("oracle dz summed", rel(dz.sum((0, 1, 2)), truth_b))
    gz = torch.zeros_like(z).reshape(B, T1 // 2, 2, 40, C)
    gz[:, :, 0] = g * first
    gz[:, :, 1] = g * (~first)
    gz = gz.reshape(B, T1, 40, C)
    diff = (gz - dz)
    bad = diff.abs() > 1e-3 * dz.abs().max()
    print("elements where the routed gradient differs from the oracle's dz by > 1e-3 max:", int(bad.sum()), "of", bad.numel())
    if bad.any():
        idx = bad.nonzero()[:10]
        for i in idx:
            i = tuple(int(v) for v in i)
            print("   at", i, " HIP-routed", float(gz[i]), " oracle dz", float(dz[i]), " z(HIP)", float(z[i]),
                  " z(oracle)", float(keep[1][2].permute(0, 3, 2, 1)[i]))


if __name__ == "__main__":
    run([0.5, 0.0, 0.0])
