"""Fused Adam on the flat parameter arena (one HIP launch), plus global-norm gradient clipping.

Semantics of torch.optim.Adam as configured by the reference (sed.py:159 ``Adam(lr=1e-3)``;
crnn_lightning.py:195-197 ``Adam(lr, weight_decay=1e-4)`` = coupled L2, not AdamW) and of
``gradient_clip_val=1.0`` (train_lightning.py:50: global L2 norm, coefficient min(1, c/(norm+1e-6))).
The clip coefficient stays on the device (no host sync).
"""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    """``FusedAdam(model.parameters(), lr=1e-3, weight_decay=0)`` — drop-in for torch.optim.Adam.

    When the parameters are exactly the arena views of one HipCRNN (the normal case: ``FusedAdam(model.parameters())``)
    the optimiser attaches to that model: p.grad are bound to the flat gradient arena (``model.bind_flat_grads()``: the
    backward writes them in place and re-binds them after ``zero_grad(set_to_none=True)``), and the update is ONE launch
    over the arena.  There is one moment store: the per-parameter path (used for a step in which some gradient is not
    the arena view, e.g. frozen parameters or hand-set grads) works on views of the same m / v arenas, so switching paths
    between steps never splits the Adam state.  ``max_grad_norm`` fuses clip_grad_norm_ into the step.
    Frozen parameters (``requires_grad_(False)``) keep ``grad is None`` and are skipped exactly like torch.optim.Adam skips
    them (no update, no weight decay); the step then takes the per-parameter path.  One difference from torch remains: the
    bias-correction step count is kept per parameter GROUP, so a parameter whose gradient is None only in SOME steps sees
    the group's count rather than its own."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        super().__init__(params, defaults)
        self._arena = None          # (flat_p, flat_g, m, v) once attached
        self._arena_model = None
        owner = self._single_owner()
        if owner is not None:
            self.attach(owner)

    def _single_owner(self):
        if len(self.param_groups) != 1:
            return None
        ps = self.param_groups[0]["params"]
        from .model import arena_owner
        owners = {id(o): o for o in (arena_owner(p) for p in ps)}
        if len(owners) != 1 or None in owners.values():
            return None
        owner = next(iter(owners.values()))
        if not owner.flat_parameters().is_cuda:
            return None
        return owner if {id(p) for p in ps} == {id(p) for p in owner._arena_params} else None

    def attach(self, model):
        """Use the model's flat arenas (also called automatically, see the class docstring)."""
        p, g = model.flat_parameters(), model.flat_grads()
        self._arena = (p, g, torch.zeros_like(p), torch.zeros_like(p))
        self._arena_model = model
        model.bind_flat_grads()
        for q, o in zip(model._arena_params, model._arena_offsets):          # the single moment store, seen per parameter
            n = q.numel()
            self.state[q]["m"] = self._arena[2][o:o + n].view(q.shape)
            self.state[q]["v"] = self._arena[3][o:o + n].view(q.shape)
        return self

    def load_state_dict(self, state_dict):
        """torch replaces the per-parameter state tensors by fresh ones: copy them back into the one moment store of the
        attached arena and re-view them, so the single-launch path keeps seeing the loaded moments"""
        super().load_state_dict(state_dict)
        if self._arena is not None:
            m = self._arena_model
            for q, o in zip(m._arena_params, m._arena_offsets):
                n = q.numel()
                st = self.state.get(q, {})
                for key, store in (("m", self._arena[2]), ("v", self._arena[3])):
                    view = store[o:o + n].view(q.shape)
                    if key in st and st[key].data_ptr() != view.data_ptr():
                        view.copy_(st[key])
                    self.state[q][key] = view

    def zero_grad(self, set_to_none=True):
        if self._arena_model is not None:
            self._arena_model._arena_dirty = False
        return super().zero_grad(set_to_none)

    def _arena_ok(self):
        """every gradient of the attached model is its arena view -> the whole update is one launch"""
        if self._arena is None:
            return False
        m = self._arena_model
        if m.flat_parameters() is not self._arena[0]:
            raise RuntimeError("FusedAdam: the model's parameter arena was rebuilt after the optimiser was created "
                               "(model.to(...) / copy): create the optimiser after moving the model")
        return all(p.grad is not None and p.grad.data_ptr() == g.data_ptr()
                   for p, g in zip(m._arena_params, m._grad_views))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            lr, (b1, b2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
            mgn = group["max_grad_norm"]
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            group.setdefault("step", 0)
            group["step"] += 1
            t = group["step"]
            if self._arena_ok():
                p, g, m, v = self._arena
                coef = ops.grad_norm_clip_coef(g, mgn)[1:2] if mgn else None
                ops.adam_step(p, g, m, v, lr, b1, b2, eps, wd, t, coef)
                continue
            coef = None
            if mgn:
                flat = torch.cat([p.grad.reshape(-1) for p in ps])
                pad = (-flat.numel()) % 4
                if pad:
                    flat = torch.cat([flat, flat.new_zeros(pad)])
                coef = ops.grad_norm_clip_coef(flat, mgn)[1:2]
            for p in ps:
                st = self.state[p]
                if "m" not in st:
                    st["m"], st["v"] = torch.zeros_like(p), torch.zeros_like(p)
                gr = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adam_step(p.data, gr, st["m"], st["v"], lr, b1, b2, eps, wd, t, coef)
        return loss


def clip_grad_norm_(model, max_norm):
    """Global-norm clip over the model's flat gradient arena; returns the device tensor [norm, coef]."""
    g = model.flat_grads()
    out = ops.grad_norm_clip_coef(g, max_norm)
    g.mul_(out[1])
    return out
