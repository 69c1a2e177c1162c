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


def fit_step_with_masks(model, optimizer, x, y, masks, loss_fn=bce_logits):
    """fit_step (sed.py:134-137) through forward_with_masks"""
    model.train()
    optimizer.zero_grad()
    out = forward_with_masks(model, x, masks)
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
