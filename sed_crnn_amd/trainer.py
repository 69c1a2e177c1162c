"""Fused fit step: zero_grad + forward + loss + backward (+ gradient all-reduce) + Adam, all native.

One step of reference sed.py:134-137 (and of Lightning's training_step + clip + Adam,
crnn_lightning.py:157-163 / train_lightning.py:50) without autograd bookkeeping: the whole-network plan
writes gradients straight into the flat arena, the backward is issued in stages so that each stage's
arena slice (head+GRU first: ~85 % of the bytes) is all-reduced over RCCL while the conv backward is
still running, and the Adam update is one launch over the arena.  No host synchronisation per step.
"""
import torch

from . import ops
from .dist import BucketedAllReduce


class FusedTrainStep:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, loss="bce",
                 focal_alpha=0.25, focal_gamma=2.0, clip_norm=None, process_group=None, distributed=None):
        self.model = model
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.loss_kind, self.alpha, self.gamma = loss, focal_alpha, focal_gamma
        self.clip_norm = clip_norm
        self.t = 0
        p = model.flat_parameters()
        self.m, self.v = torch.zeros_like(p), torch.zeros_like(p)
        if distributed is None:
            distributed = torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size(process_group) > 1
        self.reducer = BucketedAllReduce(model.flat_grads(), model.bucket_slices(), process_group) if distributed else None

    def step(self, x, y):
        """x [B,Cin,F,T], y [B,T',K] on the device -> (loss [1], probs [B,T',K]) device tensors (no sync)."""
        m = self.model
        m.train()
        logits = m._run_forward(x, training=True)
        loss, dlogits, probs = ops.loss_fwd_bwd(logits, y, self.loss_kind, self.alpha, self.gamma, "mean")
        nstage = len(m.conv_channels) + 1
        if self.reducer is None:
            m._run_backward(x, dlogits, 0, nstage)
        else:
            for s in range(nstage):
                m._run_backward(x, dlogits, s, s + 1)
                self.reducer.launch(s)                     # async all-reduce(avg) of this stage's arena slice
            self.reducer.wait_all()
        g = m.flat_grads()
        coef = ops.grad_norm_clip_coef(g, self.clip_norm)[1:2] if self.clip_norm else None
        self.t += 1
        ops.adam_step(m.flat_parameters(), g, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                      self.wd, self.t, coef)
        return loss, probs
