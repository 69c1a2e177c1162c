"""Parameterised CPU restatement of the reference networks.  TEST INFRASTRUCTURE.

Follows (does not copy) the reference:
  * SedNetRef        -> /root/reference/sed.py:82-112  (``TimePooledCRNN``)
  * LightningNetRef  -> /root/reference/crnn_lightning.py:41-73
  * focal_bce        -> /root/reference/crnn_lightning.py:27-35
  * fit_step         -> /root/reference/sed.py:134-137 (zero_grad, fwd, loss, bwd, Adam)

Both classes keep the reference's submodule names so ``state_dict()`` keys,
shapes and orders are interchangeable with the reference checkpoints, and add
keyword-only knobs (in_channels, n_mels, hidden sizes ...) for the BASELINE
configs the reference hard-codes away (sed.py:86,95,101).

Pinned by tests/golden/g1..g5 (captured from the imported reference by
oracle/make_goldens.py).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv_block(cin, cout):
    return nn.Conv2d(cin, cout, kernel_size=3, padding=1), nn.BatchNorm2d(cout)


class SedNetRef(nn.Module):
    """sed.py variant: dropout after every conv block, one nn.GRU(num_layers=L), fc."""

    def __init__(self, conv_channels=128, dropout=0.5, *, in_channels=1, n_mels=40,
                 time_pool=(2, 2, 2), gru_hidden=32, gru_layers=2, n_classes=1):
        super().__init__()
        self.convs, self.bns = nn.ModuleList(), nn.ModuleList()
        cin = in_channels
        for _ in time_pool:
            c, b = _conv_block(cin, conv_channels)
            self.convs.append(c)
            self.bns.append(b)
            cin = conv_channels
        self.time_pool = tuple(time_pool)
        self.drop = nn.Dropout(dropout)
        self.flat = conv_channels * n_mels
        self.gru = nn.GRU(self.flat, gru_hidden, num_layers=gru_layers,
                          batch_first=True, bidirectional=True)
        self.fc = nn.Linear(2 * gru_hidden, n_classes)

    def forward(self, x):                       # [B,Cin,F,T]
        for conv, bn, p in zip(self.convs, self.bns, self.time_pool):
            x = self.drop(F.max_pool2d(torch.relu(bn(conv(x))), (1, p)))
        b, c, f, t = x.shape
        x = x.permute(0, 3, 1, 2).reshape(b, t, c * f)   # feature = c*F + f
        x, _ = self.gru(x)
        return self.fc(x)                       # logits [B,T',K]


class LightningNetRef(nn.Module):
    """crnn_lightning.py variant: Sequential conv stack, dropout once, two GRUs, two dense."""

    def __init__(self, dropout=0.4, *, in_channels=1, n_mels=40, conv_depth=16,
                 time_pool=(2, 2, 2), gru1=16, gru2=8, dense1=8, n_classes=1):
        super().__init__()
        layers = []
        cin = in_channels
        for p in time_pool:
            c, b = _conv_block(cin, conv_depth)
            layers += [c, b, nn.ReLU(), nn.MaxPool2d((1, p))]
            cin = conv_depth
        layers.append(nn.Dropout(dropout))
        self.conv_stack = nn.Sequential(*layers)
        self._flat = conv_depth * n_mels
        self.gru1 = nn.GRU(self._flat, gru1, bidirectional=True, batch_first=True)
        self.gru2 = nn.GRU(2 * gru1, gru2, bidirectional=True, batch_first=True)
        self.d1 = nn.Linear(2 * gru2, dense1)
        self.d2 = nn.Linear(dense1, n_classes)

    def forward(self, x):
        x = self.conv_stack(x)
        b, c, f, t = x.shape
        x = x.permute(0, 3, 1, 2).reshape(b, t, c * f)
        x, _ = self.gru1(x)
        x, _ = self.gru2(x)
        return self.d2(torch.relu(self.d1(x)))


def focal_bce(logits, targets, alpha=0.25, gamma=2.0, reduction="mean"):
    """pt = sigma(x) where target==1 else 1-sigma(x); -alpha (1-pt)^gamma log(pt+1e-12)."""
    p = torch.sigmoid(logits)
    pt = torch.where(targets == 1, p, 1 - p)
    loss = -alpha * (1 - pt) ** gamma * torch.log(pt + 1e-12)
    return loss.mean() if reduction == "mean" else loss.sum()


def bce_logits(logits, targets):
    return F.binary_cross_entropy_with_logits(logits, targets)


def forward_with_masks(model, x, masks):
    """SedNetRef.forward (sed.py:106-112) with the Bernoulli draw of nn.Dropout replaced by GIVEN keep-masks
    (one [B,C,F,T_l/p] tensor of {0, 1/(1-p)} per conv block): two implementations with different random number
    generators can then be compared in training mode with dropout ACTIVE.  The model's own nn.Dropout must be p = 0."""
    for conv, bn, p, mk in zip(model.convs, model.bns, model.time_pool, masks):
        x = F.max_pool2d(torch.relu(bn(conv(x))), (1, p)) * mk
    b, c, f, t = x.shape
    x, _ = model.gru(x.permute(0, 3, 1, 2).reshape(b, t, c * f))
    return model.fc(x)


# ── gradient routing with GIVEN decisions ──────────────────────────────────────────────────────────────────────────────
# `pool(torch.relu(bn(conv(x))))` (sed.py:107; crnn_lightning.py:49-50) is piecewise linear: its backward sends the gradient of a
# pooled element to ONE element of the window (the first maximum) if that maximum is > 0.  Which element, and whether the
# gate is open, are DECISIONS on the BatchNorm output z; two correct fp32 implementations round z differently (~1e-7), so at
# 10^7..10^9 elements a few decisions differ, and one differing gate moves its channel's cancelling sum of gradients by a whole
# |g|.  To compare two implementations to ROUNDING, the functions below run the reference's forward with the decisions of the
# other implementation injected (as `forward_with_masks` does for the dropout draw); `audit_routes` checks first that every
# injected decision is one the reference arithmetic could have taken itself (a tie to within `tol`), so a kernel that routed
# to a wrong element is reported, not followed.

def route_mask(code, pf, pt, F, T, dtype=torch.float32):
    """code: uint8 [B,Tp,Fp,C] (0 = ReLU gate closed, 1 + w = the gradient goes to window element w = df*pt + dt, the
    row-major (F, T) window order of nn.MaxPool2d) -> R [B,C,F,T] of {0,1}: 1 on the routed element of every open window,
    0 elsewhere (incl. the ragged tail that floor pooling drops)."""
    B, Tp, Fp, C = code.shape
    cc = code.permute(0, 3, 2, 1)                                   # [B,C,Fp,Tp]
    R = torch.zeros(B, C, F, T, dtype=dtype)
    for df in range(pf):
        for dt in range(pt):
            R[:, :, df:Fp * pf:pf, dt:Tp * pt:pt] = (cc == 1 + df * pt + dt).to(dtype)
    return R


def routed_relu_pool(z, R, pf, pt):
    """relu + max_pool2d((pf,pt)) of z [B,C,F,T] with the window element and gate GIVEN by R (route_mask): the value of the
    routed element (0 for a closed gate); autograd then sends the pooled gradient to exactly that element."""
    B, C, F, T = z.shape
    Fp, Tp = F // pf, T // pt
    return (z * R)[:, :, :Fp * pf, :Tp * pt].reshape(B, C, Fp, pf, Tp, pt).sum((3, 5))      # at most one non-zero per window


def audit_routes(z, code, pf, pt, tol=2e-5):
    """Are the injected decisions ones this arithmetic could have taken?  z: the reference's own BatchNorm output [B,C,F,T].
    Returns (gates that differ, arg-maxima that differ, worst shortfall): a differing gate must have |max z| <= tol and a
    differing arg-max must pick an element within tol of the window maximum, else AssertionError."""
    B, C, F, T = z.shape
    Fp, Tp = F // pf, T // pt
    w = z.detach()[:, :, :Fp * pf, :Tp * pt].reshape(B, C, Fp, pf, Tp, pt).permute(0, 1, 2, 4, 3, 5).reshape(B, C, Fp, Tp, pf * pt)
    zmax = w.max(-1).values
    cc = code.permute(0, 3, 2, 1).long()                             # [B,C,Fp,Tp]
    gate_h = cc > 0
    gate_diff = gate_h != (zmax > 0)
    worst = float(zmax[gate_diff].abs().max()) if bool(gate_diff.any()) else 0.0
    assert worst <= tol, f"a ReLU gate was decided differently where |max z| = {worst:.3e} (not a tie)"
    picked = w.gather(-1, (cc - 1).clamp(min=0).unsqueeze(-1)).squeeze(-1)
    short = torch.where(gate_h, zmax - picked, torch.zeros_like(zmax))
    first = (w == zmax.unsqueeze(-1)).float().argmax(-1)             # first maximum, like max_pool2d
    arg_diff = gate_h & (zmax > 0) & (first != cc - 1)
    ws = float(short.max())
    assert ws <= tol * max(1.0, float(zmax.abs().max())), f"an arg-max was routed to an element {ws:.3e} below the window maximum (not a tie)"
    return int(gate_diff.sum()), int(arg_diff.sum()), max(worst, ws)


def _blocks(model):
    """(conv, bn, (pf, pt)) per block + the trailing dropout module of either reference variant"""
    if hasattr(model, "convs"):
        return [(c, b, (1, p)) for c, b, p in zip(model.convs, model.bns, model.time_pool)], model.drop, True
    mods = list(model.conv_stack)
    blocks = [(mods[i], mods[i + 1], tuple(mods[i + 3].kernel_size)) for i in range(0, len(mods) - 1, 4)]
    return blocks, mods[-1], False


def forward_routed(model, x, routes, masks=None, audit=None):
    """SedNetRef.forward (sed.py:106-112) / LightningNetRef.forward (crnn_lightning.py:66-73) with the ReLU-gate and pooling
    arg-max decisions of every block GIVEN (routes[l]: uint8 codes [B,Tp,Fp,C], see route_mask) and, optionally, the dropout
    draw given too (masks[l], as in forward_with_masks; otherwise the model's own nn.Dropout runs).  `audit`: a list that
    receives audit_routes' result per block."""
    blocks, drop, every = _blocks(model)
    for l, (conv, bn, (pf, pt)) in enumerate(blocks):
        z = bn(conv(x))
        if audit is not None:
            audit.append(audit_routes(z, routes[l], pf, pt))
        x = routed_relu_pool(z, route_mask(routes[l], pf, pt, z.shape[2], z.shape[3], z.dtype), pf, pt)
        if masks is not None:
            x = x * masks[l]
        elif every or l == len(blocks) - 1:
            x = drop(x)
    b, c, f, t = x.shape
    x = x.permute(0, 3, 1, 2).reshape(b, t, c * f)
    if hasattr(model, "gru"):
        x, _ = model.gru(x)
        return model.fc(x)
    x, _ = model.gru1(x)
    x, _ = model.gru2(x)
    return model.d2(torch.relu(model.d1(x)))


def fit_step_with_masks(model, optimizer, x, y, masks, loss_fn=bce_logits, routes=None, audit=None):
    """fit_step (sed.py:134-137) through forward_with_masks (or forward_routed when `routes` are given)"""
    model.train()
    optimizer.zero_grad()
    out = forward_with_masks(model, x, masks) if routes is None else forward_routed(model, x, routes, masks, audit=audit)
    loss = loss_fn(out, y)
    loss.backward()
    optimizer.step()
    return loss.detach(), out.detach()


def fit_step(model, optimizer, x, y, loss_fn=bce_logits, clip_norm=None):
    """One reference fit step (sed.py:134-137; grad clip from train_lightning.py:50)."""
    model.train()
    optimizer.zero_grad()
    out = model(x)
    loss = loss_fn(out, y)
    loss.backward()
    if clip_norm is not None:
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip_norm)
    optimizer.step()
    return loss.detach(), out.detach()


def synthetic_batch(B, Cin, F, T, Tp, K=1, seed=1234):
    """SURVEY 8(d) synthetic inputs: x ~ N(0,1), y = (U > 0.8)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, F, T, generator=g)
    y = (torch.rand(B, Tp, K, generator=g) > 0.8).float()
    return x, y


def rs_state_dict(model, seed, scale=None):
    """Deterministic weights from numpy RandomState, so that only the SEED has to
    travel with a golden fixture (used by make_goldens.py g5 and by the tests)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    ref = model.state_dict()
    sd = {}
    for k, v in ref.items():
        stem = k.rsplit(".", 1)[0]
        is_bn = (stem + ".running_mean") in ref
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros_like(v)
        elif is_bn and (k.endswith("running_var") or k.endswith("weight")):
            sd[k] = torch.from_numpy(rs.uniform(0.5, 1.5, size=tuple(v.shape)).astype(np.float32))
        else:
            fan = v.shape[1:].numel() if v.ndim > 1 else v.numel()
            s = (1.0 / np.sqrt(max(fan, 1))) if scale is None else scale
            sd[k] = torch.from_numpy((rs.uniform(-1, 1, size=tuple(v.shape)) * s).astype(np.float32))
    return sd
