"""Diagnostic (GPU box): which block's dropout makes block 1's dbeta drift, and whether the stand-alone BatchNorm backward of
block 1 reproduces it when fed the float64 oracle's tensors."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import torch.nn.functional as F
import sed_crnn_amd as sed
from sed_crnn_amd import ops
from oracle import crnn_ref

GOLDEN, MASK64 = 0x9E3779B97F4A7C15, (1 << 64) - 1


def masks_of(m, B, Fm, T, drops):
    out = []
    for l, C in enumerate(m.conv_channels):
        if drops[l] > 0:
            ones = torch.ones(B, T, Fm, C, device="cuda")
            mk = ops.bn_relu_pool_drop_fwd(ones, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), 1, 2,
                                           drop_p=drops[l], seed=(m._seed + GOLDEN * (l + 1)) & MASK64)
            out.append(mk.permute(0, 3, 2, 1).contiguous().cpu())
        else:
            out.append(torch.ones(1))
        T //= 2
    return out


def run(drops, B=16, T=256, C=128, H=32, standalone=False):
    torch.manual_seed(0)
    ref = crnn_ref.SedNetRef(conv_channels=C, dropout=0.0, gru_hidden=H)
    m = sed.TimePooledCRNN(conv_channels=C, dropout=0.5, gru_hidden=H)
    m.drops = list(drops)
    m.load_state_dict(ref.state_dict())
    m.cuda().train()
    x, y = crnn_ref.synthetic_batch(B, 1, 40, T, T // 8, seed=1234)
    out = m(x.cuda())
    sed.BCEWithLogitsLoss()(out, y.cuda()).backward()
    torch.cuda.synchronize()
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
    o64 = ref64.fc(hh)
    crnn_ref.bce_logits(o64, y.double()).backward()
    g64 = {k: q.grad for k, q in ref64.named_parameters()}
    print(f"--- drops {drops}: max |dlogit| {float((out.detach().cpu().double() - o64).abs().max()):.2e}")
    for k, q in m.named_parameters():
        if k.startswith(("bns.", "convs.")) and not k.endswith("s.bias") or k.startswith("bns."):
            den = float(g64[k].norm()) + 1e-30
            print(f"{k:18s} |g64| {den:9.3e}  HIP {float((q.grad.cpu().double() - g64[k]).norm()) / den:9.2e}")
    if standalone:
        l = 1
        cv, hout = keep[l]
        y_cl = cv.detach().float().permute(0, 3, 2, 1).contiguous().cuda()         # [B,T,F,C]
        dout = hout.grad.float().permute(0, 3, 2, 1).contiguous().cuda()
        bn = ref64.bns[l]
        mean = cv.detach().mean((0, 2, 3))
        var = cv.detach().var((0, 2, 3), unbiased=False)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        scale = (bn.weight.detach() * rstd)
        shift = bn.bias.detach() - mean * scale
        f32 = lambda t_: t_.float().cuda()
        dy, dgamma, dbeta, dbias = ops.bn_relu_pool_drop_bwd(y_cl, dout, f32(scale), f32(shift), f32(mean), f32(rstd), 1, 2,
                                                             drop_p=drops[l], seed=(m._seed + GOLDEN * (l + 1)) & MASK64)
        for name, got, want in (("dgamma", dgamma, g64["bns.1.weight"]), ("dbeta", dbeta, g64["bns.1.bias"]),
                                ("dy", dy, cv.grad.permute(0, 3, 2, 1))):
            print(f"stand-alone BN(1) backward on float64 tensors: {name:7s} rel err {float((got.cpu().double() - want).norm() / want.norm()):9.2e}")


if __name__ == "__main__":
    run([0.5, 0.0, 0.0])
    run([0.0, 0.5, 0.0], standalone=True)
    run([0.0, 0.0, 0.5])
    run([0.5, 0.5, 0.5], standalone=True)
