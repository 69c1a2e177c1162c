"""Diagnostic (GPU box): per-parameter distance of the HIP gradients from a float64 oracle run, next to torch-float32's own,
at config 1 (reference net, B=16 x 256 frames), with and without dropout; then block 0 alone fed with the oracle's upstream
gradient (isolates the recomputed first block from the rest of the backward chain)."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import torch.nn.functional as F
import sed_crnn_amd as sed
from sed_crnn_amd import ops
from oracle import crnn_ref
from test_gpu_fullsize import _hip_masks


def run(p, B=16, T=256, C=128, H=32, scale=1.0, shift=0.0):
    torch.manual_seed(0)
    ref = crnn_ref.SedNetRef(conv_channels=C, dropout=0.0, gru_hidden=H)
    m = sed.TimePooledCRNN(conv_channels=C, dropout=p, gru_hidden=H)
    m.load_state_dict(ref.state_dict())
    m.cuda().train()
    x, y = crnn_ref.synthetic_batch(B, 1, 40, T, T // 8, seed=1234)
    x = x * scale + shift
    out = m(x.cuda())
    sed.BCEWithLogitsLoss()(out, y.cuda()).backward()
    torch.cuda.synchronize()
    masks = _hip_masks(m, B, 40, T, p) if p > 0 else [torch.ones(1)] * 3
    ref.train()
    ref64 = crnn_ref.SedNetRef(conv_channels=C, dropout=0.0, gru_hidden=H).double()
    ref64.load_state_dict({k: v.double() if v.dtype.is_floating_point else v for k, v in ref.state_dict().items()})
    ref64.train()
    keep = {}

    def fwd(net, xx, mks, tag):
        h = xx
        for l, (conv, bn, mk) in enumerate(zip(net.convs, net.bns, mks)):
            h = F.max_pool2d(torch.relu(bn(conv(h))), (1, 2)) * mk
            if l == 0:
                h.retain_grad()
                keep[tag] = h
        b, c, f, t = h.shape
        h, _ = net.gru(h.permute(0, 3, 1, 2).reshape(b, t, c * f))
        return net.fc(h)
    crnn_ref.bce_logits(fwd(ref, x, masks, "f32"), y).backward()
    crnn_ref.bce_logits(fwd(ref64, x.double(), [mk.double() for mk in masks], "f64"), y.double()).backward()
    g32 = {k: q.grad for k, q in ref.named_parameters()}
    g64 = {k: q.grad for k, q in ref64.named_parameters()}
    print(f"--- dropout {p}  B={B} T={T} C={C} H={H} input N({shift},{scale})")
    for k, q in m.named_parameters():
        den = float(g64[k].norm()) + 1e-30
        e_h = float((q.grad.cpu().double() - g64[k]).norm()) / den
        e_t = float((g32[k].double() - g64[k]).norm()) / den
        print(f"{k:28s} |g64| {den:9.3e}  HIP {e_h:9.2e}  torch32 {e_t:9.2e}  ratio {e_h / (e_t + 1e-30):8.1f}")
    # block 0 alone: the oracle's float64 upstream gradient (cast to fp32) into the fused first block
    d64 = keep["f64"].grad                                    # [B,C,F,T/2], gradient w.r.t. the block output (after the mask)
    w, b = ref.convs[0].weight.detach(), ref.convs[0].bias.detach()
    ga, be = ref.bns[0].weight.detach(), ref.bns[0].bias.detach()
    GOLDEN, MASK64 = 0x9E3779B97F4A7C15, (1 << 64) - 1
    dout = d64.float().permute(0, 3, 2, 1).contiguous().cuda()
    o, dw, db, dga, dbe = ops.conv1_fused_block(x.cuda().contiguous(), w.cuda(), b.cuda(), ga.cuda(), be.cuda(), 1, 2, dout=dout,
                                                drop_p=p, seed=(m._seed + GOLDEN) & MASK64)
    for name, got, k in (("dw", dw, "convs.0.weight"), ("dgamma", dga, "bns.0.weight"), ("dbeta", dbe, "bns.0.bias")):
        den = float(g64[k].norm()) + 1e-30
        print(f"block 0 alone, float64 upstream: {name:7s} HIP {float((got.cpu().double() - g64[k]).norm()) / den:9.2e}")
    fo = keep["f64"].detach().permute(0, 3, 2, 1)
    print("block 0 alone: forward max err", float((o.cpu().double() - fo).abs().max()))


if __name__ == "__main__":
    run(0.5)
    run(0.0)
    run(0.0, B=8, T=64, C=32, H=32, scale=1.5, shift=0.3)
