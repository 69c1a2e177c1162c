"""Host-side mirror of the reference model API, executing on libsedcrnn.so (HIP, gfx950).

Mirrors
  * ``TimePooledCRNN(conv_channels=128, dropout=0.5)``      reference sed.py:82-112
  * ``LightningTimePooledCRNN(dropout=0.4)``                 reference crnn_lightning.py:41-73
  * ``get_model(...)``                                        named at reference README.md:44
with identical submodule / state_dict key names, shapes and initialisers, so reference checkpoints
(``best_fold{n}.pt``, sed.py:198-199) load unchanged.  The submodules are parameter holders only: no
torch.nn compute op is ever called; forward/backward go through one autograd.Function over the
whole-network plan (sed_net_forward / sed_net_backward).  There is no CPU or torch fallback: without
the HIP library or a GPU tensor the call raises.
"""
import ctypes as C
import math
import weakref

import torch
import torch.nn as nn

from . import _lib
from ._lib import NetCfg, NetParams, check, lib

_MASK64 = (1 << 64) - 1
_ARENA_OWNERS = weakref.WeakValueDictionary()       # storage address of a parameter arena -> the HipCRNN that owns it


def arena_owner(param):
    """the HipCRNN whose flat arena ``param`` is a view of, or None"""
    try:
        m = _ARENA_OWNERS.get(param.untyped_storage().data_ptr())
    except Exception:
        return None
    if m is not None and any(q is param for q in m._arena_params):
        return m
    return None


# ───────────────────────── parameter holders (same names / init as torch.nn) ─────────────────────────
class ConvParams(nn.Module):
    """nn.Conv2d(cin, cout, 3, padding=1) parameters: weight [cout,cin,3,3], bias [cout]."""

    def __init__(self, cin, cout):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(cin * 9)
        nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, kernel_size=(3, 3), padding=(1, 1) [HIP]"


class BNParams(nn.Module):
    """nn.BatchNorm2d(c) parameters and running statistics."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum} [HIP]"


class GRUParams(nn.Module):
    """nn.GRU(input_size, hidden_size, num_layers, batch_first=True, bidirectional=True) parameters."""

    def __init__(self, input_size, hidden_size, num_layers=1):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        stdv = 1.0 / math.sqrt(hidden_size)
        for k in range(num_layers):
            isz = input_size if k == 0 else 2 * hidden_size
            for sfx in ("", "_reverse"):
                for name, shape in ((f"weight_ih_l{k}{sfx}", (3 * hidden_size, isz)),
                                    (f"weight_hh_l{k}{sfx}", (3 * hidden_size, hidden_size)),
                                    (f"bias_ih_l{k}{sfx}", (3 * hidden_size,)),
                                    (f"bias_hh_l{k}{sfx}", (3 * hidden_size,))):
                    p = nn.Parameter(torch.empty(*shape))
                    nn.init.uniform_(p, -stdv, stdv)
                    self.register_parameter(name, p)

    def layer(self, k, d):
        sfx = "_reverse" if d else ""
        return tuple(getattr(self, f"{n}_l{k}{sfx}") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))

    def extra_repr(self):
        return f"{self.input_size}, {self.hidden_size}, num_layers={self.num_layers}, bidirectional=True [HIP]"


class LinearParams(nn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        self.in_features, self.out_features = fin, fout
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.empty(fout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(fin)
        nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features} [HIP]"


# ───────────────────────── autograd bridge ─────────────────────────
class _NetFn(torch.autograd.Function):
    """forward/backward of the whole network.  The parameters are passed only so that autograd sees the dependency.

    Gradient hand-over, per parameter p with arena view g (the plan always WRITES the arena, it never accumulates):
      * p.grad is already the arena view (``bind_flat_grads()`` / ``FusedAdam.attach``): the gradient is in place, autograd
        gets None (returning g as well would make AccumulateGrad add it to itself: 2g);
      * bound mode and p.grad is None (after ``zero_grad(set_to_none=True)``): p.grad is pointed at the view again;
      * otherwise autograd gets the view and applies torch's own rule (install a copy, or ``p.grad += g``).
    torch's accumulate-until-zero_grad semantics are kept in bound mode too: if a backward already ran since the last
    zero_grad, the previous arena content is added back after the plan has overwritten it."""

    @staticmethod
    def forward(ctx, net, x, *params):
        logits = net._run_forward(x, training=True)
        ctx.net, ctx.ticket = net, net._ticket
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        net = ctx.net
        if ctx.ticket != net._ticket:
            raise RuntimeError("sed_crnn_amd: the activations of this forward were overwritten by a later "
                               "training forward of the same module; call backward() before the next forward()")
        params, views = net._arena_params, net._grad_views
        aliased = [p.grad is not None and p.grad.data_ptr() == g.data_ptr() for p, g in zip(params, views)]
        carry = net._arena_grad.clone() if (net._arena_dirty and any(aliased)) else None
        net._run_backward(None, dlogits.contiguous())
        if carry is not None:                      # second backward without zero_grad: accumulate like torch would
            mask = net._alias_mask(aliased)
            net._arena_grad.add_(carry * mask if mask is not None else carry)
        net._arena_dirty = True
        out = []
        for p, g, al in zip(params, views, aliased):
            if al:
                out.append(None)
            elif not p.requires_grad:                      # frozen: the plan wrote its arena slice, nobody may see it
                out.append(None)
            elif net._bound_grads and p.grad is None:
                p.grad = g
                out.append(None)
            else:
                out.append(g)
        return (None, None) + tuple(out)


class HipCRNN(nn.Module):
    """Base class: holds the architecture spec, the flat parameter/gradient arenas and the workspace."""

    def _init_spec(self, *, in_channels, n_mels, conv_channels, pools, drops, gru_hidden, dense, bn_eps=1e-5,
                   bn_momentum=0.1):
        # the kernels' structural limits, reported when the net is built rather than at its first forward
        if not (1 <= len(conv_channels) == len(pools) == len(drops) <= _lib.SED_MAX_CONV):
            raise ValueError(f"1..{_lib.SED_MAX_CONV} conv blocks with one pool and one dropout rate each, got "
                             f"{len(conv_channels)}/{len(pools)}/{len(drops)}")
        if not (1 <= len(gru_hidden) <= _lib.SED_MAX_GRU and 1 <= len(dense) <= _lib.SED_MAX_DENSE):
            raise ValueError(f"1..{_lib.SED_MAX_GRU} GRU layers and 1..{_lib.SED_MAX_DENSE} dense layers, got "
                             f"{len(gru_hidden)} and {len(dense)}")
        bad = [c for c in conv_channels if c <= 0 or c % 4 or c > 1024]
        if bad:
            raise ValueError(f"conv channel counts must be multiples of 4 up to 1024 (float4 channel vectors), got {bad}")
        bad = [h for h in gru_hidden if h <= 0 or h % 4 or 3 * h > 1024]
        if bad:
            raise ValueError(f"GRU hidden sizes must be multiples of 4 up to 340 (the recurrence keeps 3H gate columns per "
                             f"workgroup), got {bad}")
        if any(not (0.0 <= d < 1.0) for d in drops):
            raise ValueError(f"dropout rates must lie in [0, 1), got {list(drops)}")
        if in_channels < 1 or n_mels < 1 or any(pf < 1 or pt < 1 for pf, pt in pools):
            raise ValueError("in_channels, n_mels and the pool sizes must be positive")
        self.in_channels, self.n_mels = in_channels, n_mels
        self.conv_channels = list(conv_channels)
        self.pools = [tuple(p) for p in pools]          # (pool_f, pool_t) per block
        self.drops = list(drops)
        self.gru_hidden, self.dense = list(gru_hidden), list(dense)
        self.bn_eps, self.bn_momentum = bn_eps, bn_momentum
        f = n_mels
        for pf, _ in self.pools:
            f //= pf
        self.flat_features = conv_channels[-1] * f
        self.time_factor = math.prod(pt for _, pt in self.pools)
        self._last = None
        self.overlap_wgrad = True        # BN backward of block l-1 on an auxiliary stream beside block l's weight gradient
        self._sync_bn, self._sync_group = False, None
        self._aux_stream = None
        self._ticket = 0
        self._bound_grads = False       # p.grad are the arena views (bind_flat_grads / FusedAdam.attach)
        self._arena_dirty = False       # a backward has written the arena since the last zero_grad
        self._ws_pinned = set()         # workspace keys a captured hipGraph points into: never evicted
        self._seed_counter = 0
        self._nbt_pending = 0
        self._ws = {}
        self._structs = None
        self._arena = None

    # subclasses return {role: tensor}; roles: conv_w/conv_b/bn_g/bn_b/bn_rm/bn_rv [l], gru_* [i][d], dense_* [j]
    def _roles(self):
        raise NotImplementedError

    # ── flat arenas (backward-completion order: head+GRU, then conv blocks last -> first) ──
    def _flatten(self):
        r = self._roles()
        order = []
        for j in range(len(self.dense)):
            order += [r["dense_w"][j], r["dense_b"][j]]
        for i in range(len(self.gru_hidden)):          # the two directions of each tensor adjacent: one GEMM for both
            for role in ("gru_wih", "gru_whh", "gru_bih", "gru_bhh"):
                order += [r[role][i][0], r[role][i][1]]
        stage_ends = []
        off = 0
        offs = []
        nconv = len(self.conv_channels)

        def place(ps):
            nonlocal off
            for p in ps:
                offs.append(off)
                off += (p.numel() + 3) // 4 * 4
        place(order)
        stage_ends.append(off)
        for l in range(nconv - 1, -1, -1):              # conv stage s = block l = nconv - s
            ps = [r["conv_w"][l], r["conv_b"][l], r["bn_g"][l], r["bn_b"][l]]
            order += ps
            place(ps)
            stage_ends.append(off)
        dev = order[0].device
        flat = torch.zeros(off, device=dev, dtype=torch.float32)
        flat_g = torch.zeros(off, device=dev, dtype=torch.float32)
        views = []
        with torch.no_grad():
            for p, o in zip(order, offs):
                n = p.numel()
                flat[o:o + n].copy_(p.detach().reshape(-1))
                p.data = flat[o:o + n].view(p.shape)
                views.append(flat_g[o:o + n].view(p.shape))
                p.grad = None
        assert len(order) == len(list(self.parameters())), "every parameter must have a role"
        self._arena, self._arena_grad = flat, flat_g
        _ARENA_OWNERS[flat.untyped_storage().data_ptr()] = self      # lets FusedAdam(model.parameters()) find the arena
        self._arena_params, self._grad_views, self._arena_offsets = order, views, offs
        self._stage_ends = stage_ends          # arena slice [stage_ends[s-1], stage_ends[s]) = backward stage s
        self._structs = None
        self._ws = {}
        self._ws_pinned = set()
        self._arena_dirty = False
        self._last = None
        if self._bound_grads:
            self.bind_flat_grads()

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        if hasattr(self, "_arena_params"):          # parameters were moved one by one: rebuild the arenas
            self._flatten()
        return self

    # ── copy / pickle: ctypes structs, streams and workspaces are per-process; the arenas are rebuilt ──
    def __getstate__(self):
        st = self.__dict__.copy()
        for k in ("_structs", "_ws", "_aux_stream", "_last", "_arena", "_arena_grad", "_arena_params", "_grad_views",
                  "_arena_offsets", "_stage_ends"):
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._structs, self._ws, self._aux_stream, self._arena = None, {}, None, None
        with torch.no_grad():                      # un-alias the (deep-copied / unpickled) parameters, then re-flatten
            for p_ in self.parameters():
                p_.data = p_.data.clone()
        self._flatten()

    def flat_parameters(self):
        return self._arena

    def flat_grads(self):
        return self._arena_grad

    def bucket_slices(self):
        """One arena slice per backward stage: the gradients that are complete when that stage returns (empty for a stage
        that completes none).  Stage 0 = head + GRU; the conv weight gradients are deferred to the stage of block 1
        (``sed_net_backward_ready_stage``), so the slices of blocks >= 1 merge into that stage's slice."""
        b = [0] + self._stage_ends                   # arena order: head+GRU, then conv blocks top -> 0
        n = len(self.conv_channels)
        cfg = self._cfg(1, self.time_factor)
        ready = [lib().sed_net_backward_ready_stage(C.byref(cfg), n - s) for s in range(1, n + 1)]   # per arena stage s
        out = [(b[0], b[1])]
        for st in range(1, n + 1):
            mine = [s for s in range(1, n + 1) if ready[s - 1] == st]
            out.append((b[mine[0]], b[mine[-1] + 1]) if mine else (b[st], b[st]))
        return out

    def bind_flat_grads(self):
        """Point every p.grad at its arena view (what the fused optimiser / all-reduce operate on) and keep it so: after a
        ``zero_grad(set_to_none=True)`` the next backward binds them again."""
        self._bound_grads = True
        for p, g in zip(self._arena_params, self._grad_views):
            p.grad = g if p.requires_grad else None        # a frozen parameter keeps grad None, like under torch autograd

    def zero_grad(self, set_to_none=True):
        self._arena_dirty = False
        return super().zero_grad(set_to_none)

    def _alias_mask(self, aliased):
        """1 on the arena elements whose parameter's .grad is the arena view, 0 elsewhere (None = all of them)"""
        if all(aliased):
            return None
        m = torch.zeros_like(self._arena_grad)
        for p, o, al in zip(self._arena_params, self._arena_offsets, aliased):
            if al:
                m[o:o + p.numel()] = 1.0
        return m

    # ── C structs ──
    def _cfg(self, B, T):
        c = NetCfg()
        c.B, c.Cin, c.F, c.T = B, self.in_channels, self.n_mels, T
        c.n_conv = len(self.conv_channels)
        for l, ch in enumerate(self.conv_channels):
            c.C[l] = ch
            c.pool_f[l], c.pool_t[l] = self.pools[l]
            c.drop_p[l] = float(self.drops[l])
        c.n_gru = len(self.gru_hidden)
        for i, h in enumerate(self.gru_hidden):
            c.H[i] = h
        c.n_dense = len(self.dense)
        for j, d in enumerate(self.dense):
            c.D[j] = d
        c.bn_eps, c.bn_momentum = self.bn_eps, self.bn_momentum
        c.conv_mode = int(getattr(self, "conv_mode", 0))
        c.flags = int(getattr(self, "plan_flags", 0))          # SED_NET_* (tests / A-B measurements of the backward schedule)
        return c

    def set_conv_precision(self, name="f32"):
        """EXPERIMENT, explicit opt-in: ``"bf16x3"`` runs the forward and the data gradient of the 128-channel conv blocks on
        a 3-term bf16 split (fp32 accumulate; relative error ~4e-6 per K = 1152 sum instead of 3e-7); ``"f32"`` (default) is
        the exact-fp32 MFMA path that every parity claim and the bench refer to."""
        self.conv_mode = {"f32": 0, "bf16x3": 1}[name]
        return self

    def _param_structs(self):
        if self._structs is None:
            r = self._roles()
            gmap = {id(p): g for p, g in zip(self._arena_params, self._grad_views)}
            P, G = NetParams(), NetParams()

            def put(field, idx, t, grad=True):
                tgt_p, tgt_g = getattr(P, field), getattr(G, field)
                if isinstance(idx, tuple):
                    tgt_p[idx[0]][idx[1]] = t.data_ptr()
                    tgt_g[idx[0]][idx[1]] = gmap[id(t)].data_ptr() if grad else None
                else:
                    tgt_p[idx] = t.data_ptr()
                    tgt_g[idx] = gmap[id(t)].data_ptr() if grad else None
            for l in range(len(self.conv_channels)):
                put("conv_w", l, r["conv_w"][l]); put("conv_b", l, r["conv_b"][l])
                put("bn_g", l, r["bn_g"][l]); put("bn_b", l, r["bn_b"][l])
                put("bn_rm", l, r["bn_rm"][l], grad=False); put("bn_rv", l, r["bn_rv"][l], grad=False)
            for i in range(len(self.gru_hidden)):
                for d in range(2):
                    put("gru_wih", (i, d), r["gru_wih"][i][d]); put("gru_whh", (i, d), r["gru_whh"][i][d])
                    put("gru_bih", (i, d), r["gru_bih"][i][d]); put("gru_bhh", (i, d), r["gru_bhh"][i][d])
            for j in range(len(self.dense)):
                put("dense_w", j, r["dense_w"][j]); put("dense_b", j, r["dense_b"][j])
            self._structs = (P, G)
        return self._structs

    def _workspace(self, cfg, training):
        key = (cfg.B, cfg.T, bool(training))
        ws = self._ws.get(key)
        if ws is None:
            nbytes = lib().sed_net_workspace_bytes(C.byref(cfg), int(training))
            if nbytes == 0:
                check(-1, "sed_net_workspace_bytes")
            ws = torch.empty(nbytes // 4 + 64, device=self._arena.device, dtype=torch.float32)
            while len(self._ws) - len(self._ws_pinned) >= 4:          # drop the oldest shape no captured graph points into
                victim = next((k for k in self._ws if k not in self._ws_pinned), None)
                if victim is None:
                    break
                if self._last is not None and self._last[1] is self._ws[victim]:
                    self._last = None
                del self._ws[victim]
            self._ws[key] = ws
        return ws

    def pin_workspace(self, B, T, training=True):
        """A captured hipGraph holds raw pointers into the workspace of this shape: keep it for the module's lifetime."""
        self._ws_pinned.add((int(B), int(T), bool(training)))

    def _check_input(self, x):
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise _lib.SedHipError("sed_crnn_amd runs on an MI355X only: input must be a CUDA(HIP) tensor "
                                   "(there is no CPU fallback; the CPU reference lives in oracle/ for tests)")
        if self._arena is None or not self._arena.is_cuda:
            raise _lib.SedHipError("sed_crnn_amd: move the module to the GPU first (model.to('cuda'))")
        if x.dim() != 4 or x.shape[1] != self.in_channels or x.shape[2] != self.n_mels:
            raise ValueError(f"expected input [B,{self.in_channels},{self.n_mels},T], got {tuple(x.shape)}")
        if x.shape[3] < self.time_factor:        # floor pooling like nn.MaxPool2d (sed.py:90): a ragged tail is dropped
            raise ValueError(f"T={x.shape[3]} is shorter than one output frame ({self.time_factor} input frames)")
        if not (x.dtype.is_floating_point or x.dtype in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64)):
            raise ValueError(f"unsupported input dtype {x.dtype}")

    @staticmethod
    def _rank_salt():
        """data-parallel ranks must not drop the same positions of their shards: mix the rank into the dropout seed"""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank() * 0xA24BAED4963EE407
        return 0

    def _graph_seed(self):
        # constant base seed of a captured step; the per-step variation comes from the device-side salt
        return (torch.initial_seed() * 0x9E3779B97F4A7C15 + 0x51ED270B + self._rank_salt()) & _MASK64

    def _next_seed(self):
        self._seed_counter += 1
        return (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._seed_counter * 0xD1B54A32D192ED03 + self._rank_salt()) & _MASK64

    # ── synchronised BatchNorm for data-parallel training ──
    def enable_sync_bn(self, process_group=None, enabled=True):
        """Batch statistics (and their backward sums) over ALL ranks' samples: an N-rank run on a sharded global batch
        then equals the single-device run on that batch (what torch.nn.SyncBatchNorm does for DDP).  Costs one 2C-float
        all-reduce per conv block in forward and one in backward; the plan runs in phases on one stream."""
        self._sync_bn, self._sync_group = bool(enabled), process_group
        return self

    def _sync_world(self):
        import torch.distributed as dist
        if self._sync_bn and dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self._sync_group)
        return 1

    def _allreduce_region(self, cfg, ws, backward, block):
        import torch.distributed as dist
        off, n = C.c_size_t(), C.c_size_t()
        check(lib().sed_net_sync_region(C.byref(cfg), int(backward), block, C.byref(off), C.byref(n)), "sed_net_sync_region")
        dist.all_reduce(ws[off.value // 4: off.value // 4 + n.value], op=dist.ReduceOp.SUM, group=self._sync_group)

    def workspace_view(self, name, index=0):
        """A view of an intermediate of the LAST training forward / backward inside the plan's workspace
        (``sed_net_workspace_region``: "conv_out", "pooled", "mean", "rstd", "scale", "shift", "dconv", "gi", "gru_out",
        "dgru_out", "grad_act", "bn_sums_bwd", "wgrad_zero_row").  Flat fp32; valid until the next forward of that shape."""
        if self._last is None:
            raise RuntimeError("sed_crnn_amd: no training forward to look into")
        cfg, ws, _ = self._last
        off, n = C.c_size_t(), C.c_size_t()
        check(lib().sed_net_workspace_region(C.byref(cfg), 1, name.encode(), int(index), C.byref(off), C.byref(n)),
              "sed_net_workspace_region")
        return ws[off.value // 4: off.value // 4 + n.value]

    def routing(self, block):
        """The ReLU-gate / pooling arg-max decisions of conv block ``block`` in the LAST training forward, exactly as the
        backward kernels take them (``sed_net_routing``): uint8 [B, T_l/pt, F_l/pf, C], 0 = gate closed, 1 + w = the gradient
        goes to window element w = df*pt + dt.  Inspection / parity tests (oracle.crnn_ref.forward_routed).  Call it BEFORE the
        optimiser step: a recomputed first block re-reads its (live) conv bias."""
        if self._last is None:
            raise RuntimeError("sed_crnn_amd: no training forward to look into")
        cfg, ws, x = self._last
        P, _ = self._param_structs()
        T, F = cfg.T, cfg.F
        for l in range(block):
            T, F = T // self.pools[l][1], F // self.pools[l][0]
        pf, pt = self.pools[block]
        route = torch.empty(cfg.B, T // pt, F // pf, self.conv_channels[block], device=x.device, dtype=torch.uint8)
        check(lib().sed_net_routing(C.byref(cfg), C.byref(P), _lib.ptr(x), _lib.ptr(ws), int(block), _lib.ptr(route),
                                    _lib.stream_ptr()), "sed_net_routing")
        return route

    # ── raw plan calls (also used by the fused trainer) ──
    def _run_forward(self, x, training, step_state=None):
        """step_state: optional device tensor {dropout salt, optimiser step} (uint64[2]) read by the kernels instead of a
        host-side seed, which makes the launch sequence replayable as a hipGraph (trainer.FusedTrainStep(graph=True))."""
        self._check_input(x)
        x = x.detach().contiguous().float()        # the kernels read dense fp32 [B,Cin,F,T]; backward re-reads THIS tensor
        B, _, _, T = x.shape
        cfg = self._cfg(B, T)
        P, _ = self._param_structs()
        ws = self._workspace(cfg, training)
        logits = torch.empty(B, T // self.time_factor, self.dense[-1], device=x.device, dtype=torch.float32)
        if training:
            self._ticket += 1
            self._seed = self._next_seed() if step_state is None else self._graph_seed()
            self._nbt_pending += 1
            self._last = (cfg, ws, x)
            self._last_state = step_state
        seed = self._seed if training else 0
        world = self._sync_world() if training else 1
        if world > 1:                                            # phased: all-reduce the statistic sums of every block
            args = (C.byref(cfg), C.byref(P), _lib.ptr(x), _lib.ptr(logits), _lib.ptr(ws), 1, seed)
            n = len(self.conv_channels)
            for l in range(n):
                check(lib().sed_net_forward_phases(*args, 2 * l, 2 * l + 1, float(world), _lib.stream_ptr()), "sed_net_forward_phases")
                self._allreduce_region(cfg, ws, 0, l)
                check(lib().sed_net_forward_phases(*args, 2 * l + 1, 2 * l + 2, float(world), _lib.stream_ptr()), "sed_net_forward_phases")
            check(lib().sed_net_forward_phases(*args, 2 * n, 2 * n + 1, float(world), _lib.stream_ptr()), "sed_net_forward_phases")
            return logits
        check(lib().sed_net_forward(C.byref(cfg), C.byref(P), _lib.ptr(x), _lib.ptr(logits), _lib.ptr(ws),
                                    int(training), seed, _lib.ptr(step_state) if training else None, _lib.stream_ptr()),
              "sed_net_forward")
        return logits

    def _run_backward(self, x, dlogits, stage_begin=0, stage_end=None):
        """``x`` is ignored (kept for call compatibility): the first block re-reads the network input, and it must be
        the converted (contiguous fp32) tensor the forward ran on, which the forward kept."""
        if self._last is None:
            raise RuntimeError("sed_crnn_amd: backward without a training forward (or its workspace was evicted)")
        cfg, ws, x = self._last
        P, G = self._param_structs()
        if stage_end is None:
            stage_end = len(self.conv_channels) + 1
        world = self._sync_world()
        if world > 1:                                            # stage s >= 1 = phases 2(s-1)+1 | all-reduce | 2(s-1)+2
            args = (C.byref(cfg), C.byref(P), C.byref(G), _lib.ptr(x), _lib.ptr(dlogits), _lib.ptr(ws), self._seed)
            n = len(self.conv_channels)
            for s in range(stage_begin, stage_end):
                if s == 0:
                    check(lib().sed_net_backward_phases(*args, 0, 1, float(world), _lib.stream_ptr()), "sed_net_backward_phases")
                    continue
                k = s - 1
                check(lib().sed_net_backward_phases(*args, 2 * k + 1, 2 * k + 2, float(world), _lib.stream_ptr()), "sed_net_backward_phases")
                self._allreduce_region(cfg, ws, 1, n - 1 - k)
                check(lib().sed_net_backward_phases(*args, 2 * k + 2, 2 * k + 3, float(world), _lib.stream_ptr()), "sed_net_backward_phases")
            return
        if getattr(self, "_aux_stream", None) is None or self._aux_stream.device != x.device:
            self._aux_stream = torch.cuda.Stream(device=x.device)
        aux = C.c_void_p(self._aux_stream.cuda_stream) if self.overlap_wgrad else None
        check(lib().sed_net_backward(C.byref(cfg), C.byref(P), C.byref(G), _lib.ptr(x), _lib.ptr(dlogits),
                                     _lib.ptr(ws), self._seed, _lib.ptr(getattr(self, "_last_state", None)), stage_begin,
                                     stage_end, _lib.stream_ptr(), aux), "sed_net_backward")

    # ── nn.Module surface ──
    def forward(self, x):                                   # x [B,Cin,F,T] -> logits [B,T',K]
        if self.training and torch.is_grad_enabled():
            return _NetFn.apply(self, x, *self._arena_params)
        return self._run_forward(x, training=self.training)

    def _flush_nbt(self):
        if self._nbt_pending:
            for m in self.modules():
                if isinstance(m, BNParams):
                    m.num_batches_tracked += self._nbt_pending
            self._nbt_pending = 0

    def state_dict(self, *a, **k):
        self._flush_nbt()
        return super().state_dict(*a, **k)

    def load_state_dict(self, sd, *a, **k):
        self._nbt_pending = 0
        out = super().load_state_dict(sd, *a, **k)     # copies in place: arena views stay valid
        return out


# ───────────────────────── the two reference networks ─────────────────────────
class TimePooledCRNN(HipCRNN):
    """Drop-in for reference sed.py:82-112: ``convs``/``bns`` ModuleLists, ``gru`` (2-layer BiGRU), ``fc``.

    ``TimePooledCRNN(conv_channels=128, dropout=0.5)`` is the reference signature; the keyword-only
    arguments open up what the reference hard-codes (sed.py:86,95,101) for the BASELINE configs.
    """

    def __init__(self, conv_channels=128, dropout=0.5, *, in_channels=1, n_mels=40, time_pool=(2, 2, 2),
                 gru_hidden=32, gru_layers=2, n_classes=1):
        super().__init__()
        n = len(time_pool)
        self._init_spec(in_channels=in_channels, n_mels=n_mels, conv_channels=[conv_channels] * n,
                        pools=[(1, p) for p in time_pool], drops=[dropout] * n,     # dropout after EVERY block
                        gru_hidden=[gru_hidden] * gru_layers, dense=[n_classes])
        self.convs, self.bns = nn.ModuleList(), nn.ModuleList()
        ch = in_channels
        for _ in time_pool:
            self.convs.append(ConvParams(ch, conv_channels))
            self.bns.append(BNParams(conv_channels))
            ch = conv_channels
        self.flat = self.flat_features
        self.gru = GRUParams(self.flat, gru_hidden, gru_layers)
        self.fc = LinearParams(2 * gru_hidden, n_classes)
        self._flatten()

    def _roles(self):
        L = len(self.gru_hidden)
        gl = [[self.gru.layer(k, d) for d in range(2)] for k in range(L)]
        return {
            "conv_w": [c.weight for c in self.convs], "conv_b": [c.bias for c in self.convs],
            "bn_g": [b.weight for b in self.bns], "bn_b": [b.bias for b in self.bns],
            "bn_rm": [b.running_mean for b in self.bns], "bn_rv": [b.running_var for b in self.bns],
            "gru_wih": [[gl[k][d][0] for d in range(2)] for k in range(L)],
            "gru_whh": [[gl[k][d][1] for d in range(2)] for k in range(L)],
            "gru_bih": [[gl[k][d][2] for d in range(2)] for k in range(L)],
            "gru_bhh": [[gl[k][d][3] for d in range(2)] for k in range(L)],
            "dense_w": [self.fc.weight], "dense_b": [self.fc.bias],
        }


class LightningTimePooledCRNN(HipCRNN):
    """Drop-in for reference crnn_lightning.py:41-73: ``conv_stack`` (conv at 0,4,8; BN at 1,5,9; dropout
    once at the end), ``gru1``, ``gru2``, ``d1`` (+ReLU), ``d2``; attributes ``T_out`` and ``_flat``."""

    def __init__(self, dropout=0.4, *, in_channels=1, n_mels=40, conv_depth=16, time_pool=(2, 2, 2),
                 gru1_units=16, gru2_units=8, dense1_units=8, n_classes=1, seq_len_in=64):
        super().__init__()
        n = len(time_pool)
        self._init_spec(in_channels=in_channels, n_mels=n_mels, conv_channels=[conv_depth] * n,
                        pools=[(1, p) for p in time_pool], drops=[0.0] * (n - 1) + [dropout],
                        gru_hidden=[gru1_units, gru2_units], dense=[dense1_units, n_classes])
        self.conv_stack = nn.Sequential()
        ch = in_channels
        for p in time_pool:
            self.conv_stack.append(ConvParams(ch, conv_depth))
            self.conv_stack.append(BNParams(conv_depth))
            self.conv_stack.append(nn.Identity())           # ReLU slot (fused into the HIP kernel)
            self.conv_stack.append(nn.Identity())           # MaxPool2d((1,p)) slot (fused)
            ch = conv_depth
        self.conv_stack.append(nn.Identity())               # Dropout slot (fused)
        self.T_out = seq_len_in // self.time_factor
        self._flat = self.flat_features
        self.gru1 = GRUParams(self._flat, gru1_units)
        self.gru2 = GRUParams(2 * gru1_units, gru2_units)
        self.d1 = LinearParams(2 * gru2_units, dense1_units)
        self.d2 = LinearParams(dense1_units, n_classes)
        self._flatten()

    def _roles(self):
        n = len(self.conv_channels)
        convs = [self.conv_stack[4 * i] for i in range(n)]
        bns = [self.conv_stack[4 * i + 1] for i in range(n)]
        g = [[self.gru1.layer(0, d) for d in range(2)], [self.gru2.layer(0, d) for d in range(2)]]
        return {
            "conv_w": [c.weight for c in convs], "conv_b": [c.bias for c in convs],
            "bn_g": [b.weight for b in bns], "bn_b": [b.bias for b in bns],
            "bn_rm": [b.running_mean for b in bns], "bn_rv": [b.running_var for b in bns],
            "gru_wih": [[g[k][d][0] for d in range(2)] for k in range(2)],
            "gru_whh": [[g[k][d][1] for d in range(2)] for k in range(2)],
            "gru_bih": [[g[k][d][2] for d in range(2)] for k in range(2)],
            "gru_bhh": [[g[k][d][3] for d in range(2)] for k in range(2)],
            "dense_w": [self.d1.weight, self.d2.weight], "dense_b": [self.d1.bias, self.d2.bias],
        }


class SEDNet(HipCRNN):
    """General topology behind ``get_model``: per-block (pool_f, pool_t), stacked BiGRUs, dense head.
    State-dict keys: convs.{l}, bns.{l}, grus.{i}, fcs.{j}."""

    def __init__(self, in_channels, n_mels, conv_channels, pools, rnn_hidden, fc, dropout, dropout_every_block=True):
        super().__init__()
        n = len(pools)
        chans = [conv_channels] * n if isinstance(conv_channels, int) else list(conv_channels)
        drops = [dropout] * n if dropout_every_block else [0.0] * (n - 1) + [dropout]
        self._init_spec(in_channels=in_channels, n_mels=n_mels, conv_channels=chans, pools=pools, drops=drops,
                        gru_hidden=list(rnn_hidden), dense=list(fc))
        self.convs, self.bns = nn.ModuleList(), nn.ModuleList()
        ch = in_channels
        for c in chans:
            self.convs.append(ConvParams(ch, c))
            self.bns.append(BNParams(c))
            ch = c
        self.grus, self.fcs = nn.ModuleList(), nn.ModuleList()
        fin = self.flat_features
        for h in rnn_hidden:
            self.grus.append(GRUParams(fin, h))
            fin = 2 * h
        for d in fc:
            self.fcs.append(LinearParams(fin, d))
            fin = d
        self._flatten()

    def _roles(self):
        g = [[m.layer(0, d) for d in range(2)] for m in self.grus]
        n = len(g)
        return {
            "conv_w": [c.weight for c in self.convs], "conv_b": [c.bias for c in self.convs],
            "bn_g": [b.weight for b in self.bns], "bn_b": [b.bias for b in self.bns],
            "bn_rm": [b.running_mean for b in self.bns], "bn_rv": [b.running_var for b in self.bns],
            "gru_wih": [[g[k][d][0] for d in range(2)] for k in range(n)],
            "gru_whh": [[g[k][d][1] for d in range(2)] for k in range(n)],
            "gru_bih": [[g[k][d][2] for d in range(2)] for k in range(n)],
            "gru_bhh": [[g[k][d][3] for d in range(2)] for k in range(n)],
            "dense_w": [m.weight for m in self.fcs], "dense_b": [m.bias for m in self.fcs],
        }


def get_model(in_channels=1, n_mels=40, seq_len=256, n_classes=1, conv_channels=128,
              pools=((1, 2), (1, 2), (1, 2)), rnn_hidden=(32, 32), fc=None, dropout=0.5,
              dropout_every_block=True):
    """Factory named by the reference README (README.md:44; its body is not in the reference tree).

    ``pools`` are (pool_mel, pool_time) per conv block: the fork's time-pooled net is the default; the
    README figure's original SEDnet is ``pools=[(5,1),(2,1),(2,1)], n_classes=6, fc=[16, 6]``.
    ``fc`` lists the dense sizes after the GRUs (ReLU between them); default ``[n_classes]``.
    ``seq_len`` must cover one output frame and is otherwise free (the kernels take any T; a ragged tail is dropped
    by the pooling like ``nn.MaxPool2d`` does).
    """
    tf = math.prod(p[1] for p in pools)
    if seq_len < tf:
        raise ValueError(f"seq_len={seq_len} is shorter than one output frame (total time pooling {tf})")
    fc = [n_classes] if fc is None else list(fc)
    if fc[-1] != n_classes:
        raise ValueError("the last dense size must equal n_classes")
    return SEDNet(in_channels, n_mels, conv_channels, [tuple(p) for p in pools], list(rnn_hidden), fc, dropout,
                  dropout_every_block)
