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
                 focal_alpha=0.25, focal_gamma=2.0, clip_norm=None, process_group=None, distributed=None, graph=False):
        self.model = model
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.loss_kind, self.alpha, self.gamma = loss, focal_alpha, focal_gamma
        self.clip_norm = clip_norm
        self.t = 0
        p = self._p = model.flat_parameters()
        self.m, self.v = torch.zeros_like(p), torch.zeros_like(p)
        if distributed is None:
            distributed = torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size(process_group) > 1
        self.reducer = BucketedAllReduce(model.flat_grads(), model.bucket_slices(), process_group) if distributed else None
        # graph=True: the whole step (forward, loss, backward on both streams, Adam) is captured once per input shape into a
        # hipGraph and replayed; the dropout salt and the optimiser step live in a 2-word device state advanced in-graph.
        # Pays off when the step is launch-bound (small nets / batches: ~110 launches); single-process only.
        self.graph = bool(graph) and self.reducer is None
        self._graphs = {}
        self._state = torch.zeros(2, dtype=torch.int64, device=p.device) if self.graph else None
        # measurement aid (bench.py): with time_allreduce set, every data-parallel step brackets reducer.wait_all() with two
        # events on the compute stream; their distance is the part of the gradient all-reduce that the backward did NOT hide
        self.time_allreduce = False
        self.allreduce_events = []

    def step(self, x, y):
        """x [B,Cin,F,T], y [B,T',K] on the device -> (loss [1], probs [B,T',K]) device tensors (no sync)."""
        if self.model.flat_parameters() is not self._p:
            raise RuntimeError("FusedTrainStep: the model's arenas were rebuilt (model.to(...) / copy) after the trainer was "
                               "created; its Adam moments, captured graphs and all-reduce buckets point at the old ones")
        if self.graph:
            return self._step_graph(x, y)
        return self._step_eager(x, y)

    def _step_graph(self, x, y):
        """call 1 for a shape runs eagerly (it also warms up workspaces, LDS attributes and events), call 2 captures the
        step and replays it, later calls only replay: every call is exactly one real fit step"""
        key = (tuple(x.shape), tuple(y.shape))
        ent = self._graphs.get(key)
        if ent is None:
            self._graphs[key] = "warm"
            return self._step_eager(x, y.float(), state=self._state)
        if ent == "warm":
            sx, sy = x.clone(), y.clone().float()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._step_eager(sx, sy, state=self._state)
            self.t -= 1                                          # the capture pass enqueued nothing
            self.model.pin_workspace(sx.shape[0], sx.shape[3], True)   # the graph holds raw pointers into that workspace
            ent = self._graphs[key] = (g, sx, sy, out, self.model._last[1])
        g, sx, sy, out, _ws = ent
        sx.copy_(x)
        sy.copy_(y)
        g.replay()
        self.t += 1
        return out

    def _step_eager(self, x, y, state=None):
        m = self.model
        m.train()
        if state is not None:
            from ._lib import check, lib, ptr, stream_ptr
            check(lib().sed_step_advance(ptr(state), stream_ptr()), "sed_step_advance")
        logits = m._run_forward(x, training=True, step_state=state)
        loss, dlogits, probs = ops.loss_fwd_bwd(logits, y, self.loss_kind, self.alpha, self.gamma, "mean")
        nstage = len(m.conv_channels) + 1
        if self.reducer is None:
            m._run_backward(x, dlogits, 0, nstage)
        else:
            for s in range(nstage):
                m._run_backward(x, dlogits, s, s + 1)
                self.reducer.launch(s)                     # async all-reduce(avg) of this stage's arena slice
            if self.time_allreduce and x.is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()                                # the backward kernels of this rank are done here
                self.reducer.wait_all()
                e1.record()                                # ... and here every slice has arrived
                self.allreduce_events.append((e0, e1))
            else:
                self.reducer.wait_all()
        g = m.flat_grads()
        coef = ops.grad_norm_clip_coef(g, self.clip_norm)[1:2] if self.clip_norm else None
        self.t += 1
        ops.adam_step(m.flat_parameters(), g, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                      self.wd, self.t, coef, step_state=state)
        return loss, probs
