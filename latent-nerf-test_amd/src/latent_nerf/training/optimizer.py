"""Fused HIP Adam for the NeRF parameters (betas/eps of the reference's
src/latent_paint/training/trainer.py:93-95: Adam(betas=(0.9, 0.99), eps=1e-15)).

Two launches per step whatever the number of tensors: the hash table (16 B/lane streaming pass that also
refreshes its bf16 shadow) and one multi-tensor launch for every small parameter."""
import ctypes

import torch

from ..raymarching import backend as _b
from ..raymarching.raymarching import _p, _stream


class FusedAdam:
    def __init__(self, param_groups, betas=(0.9, 0.99), eps=1e-15, encoder=None, capturable=False,
                 fuse_table_update=False, mlp=None, tail=None):
        """param_groups: [{'params': [...], 'lr': float}, ...] (as NeRFNetwork.get_params(lr)).
        encoder: the GridEncoder whose `embeddings` are in the groups (for the bf16 shadow refresh).
        fuse_table_update: the hash table's Adam step can be applied by the scatter of the backward pass
        (lnerf_grid_encode_backward_adam) instead of by step(): call arm() right before the step's backward
        (single process, ONE backward per step; set the `grad_scale` attribute first if it is not 1).  A step
        whose backward was not armed takes the ordinary path.
        mlp: the NeRFNetwork whose w1 / w2 / w3 are in the groups (bf16 MLP): the multi-tensor launch then also writes
        the updated weights into the network's bf16 weight fragments (lnerf_adam_step_multi_shadow), and the network's
        forward stops rebuilding them every step (one dispatch less).
        tail (default: on whenever fuse_table_update and mlp are given): an ARMED step is closed by the scatter's own pass
        2 (lnerf_grid_encode_backward_adam_tail: extra workgroups of that launch sum the MLP's gradient slabs, step its
        six tensors straight from the sums, advance the device step counter and leave the scatter's level maxima clean
        for the next step) -- or, when other small parameters must be stepped first, by ONE launch behind their Adam
        launch (lnerf_step_tail: the same slab blocks + tick + clearing; pass 2 finishes its sliced buckets itself
        either way).  The six tensors then never get a `.grad` on armed steps."""
        self.betas, self.eps = betas, eps
        self.encoder = encoder
        self.step_no = 0
        self.grad_scale = 1.0
        self.fused = None
        # capturable: the step counter lives on the device so that a hipGraph of the whole step can be replayed
        self.capturable = capturable
        self.step_dev = None
        self.big = []    # (param, m, v, lr)
        self.small = []
        for group in param_groups:
            for p in group["params"]:
                entry = (p, torch.zeros_like(p), torch.zeros_like(p), float(group["lr"]))
                is_table = encoder is not None and p is encoder.embeddings
                (self.big if (is_table or p.numel() >= (1 << 20)) else self.small).append(entry)
        if len(self.small) > 16:
            raise ValueError("FusedAdam handles at most 16 small tensors per launch")
        n = len(self.small)
        self._pp = (ctypes.c_void_p * n)()
        self._gp = (ctypes.c_void_p * n)()
        self._mp = (ctypes.c_void_p * n)(*[e[1].data_ptr() for e in self.small])
        self._vp = (ctypes.c_void_p * n)(*[e[2].data_ptr() for e in self.small])
        self._n = (ctypes.c_int64 * n)(*[e[0].numel() for e in self.small])
        self._lr = (ctypes.c_float * n)(*[e[3] for e in self.small])
        self.mlp, self._maps, self._map_tensors = None, None, None
        if mlp is not None and getattr(mlp, "precision", None) == "bf16" and self.small:
            dev0 = self.small[0][0].device
            tensors = {}
            for name in ("w1", "w2", "w3"):
                w = getattr(mlp, name)
                # -1 = "no position": the Adam kernel skips negative entries, so a slot the map kernel leaves out can
                # never become a wild store
                tensors[name] = torch.full((w.numel() * 2,), -1, device=dev0, dtype=torch.int32)
            _b.call("lnerf_mlp_fragment_maps", int(mlp.w3.shape[0]), _p(tensors["w1"]), _p(tensors["w2"]),
                    _p(tensors["w3"]), _stream())
            frag_elems = _b.MLP_FRAGMENT_BYTES // 2   # bf16 elements of the fragment image
            for name, tmap in tensors.items():   # (once, at construction: a read-back is fine here)
                lo, hi = int(tmap.min()), int(tmap.max())
                if lo < 0 or hi >= frag_elems:
                    raise _b.LnerfError("lnerf_mlp_fragment_maps: %s has positions outside [0, %d): [%d, %d]"
                                        % (name, frag_elems, lo, hi))
            maps = (ctypes.c_void_p * n)()
            found = 0
            for k, (p, *_r) in enumerate(self.small):
                for name in ("w1", "w2", "w3"):
                    if p is getattr(mlp, name):
                        maps[k] = tensors[name].data_ptr()
                        found += 1
            if found != 3:
                raise ValueError("FusedAdam(mlp=...): w1, w2, w3 of the network must be among the small parameters")
            self.mlp, self._maps, self._map_tensors = mlp, maps, tensors
            mlp._frag_owner = self
        if capturable:
            dev = (self.big + self.small)[0][0].device
            self.step_dev = torch.tensor([1, 0], device=dev, dtype=torch.int32)  # [0]: value used by the NEXT step; [1]: arrival counter of the ticking launch
        if fuse_table_update:
            if encoder is None:
                raise ValueError("fuse_table_update needs the encoder")
            from ..models.encoding import FusedTableUpdate
            for p, m, v, lr in self.big:
                if p is encoder.embeddings:
                    self.fused = FusedTableUpdate(m, v, lr, betas, eps, self)
                    encoder.fused_update = self.fused
            if self.fused is None:
                raise ValueError("fuse_table_update: encoder.embeddings is not among the parameters")
        self._tail = None
        if tail is None:
            tail = self.fused is not None and mlp is not None
        if tail:
            if self.fused is None or mlp is None:
                raise ValueError("FusedAdam(tail=True) needs fuse_table_update=True and mlp=<the network>")
            six = [getattr(mlp, k) for k in ("w1", "b1", "w2", "b2", "w3", "b3")]
            idx = []
            for q in six:
                hit = [k for k, e in enumerate(self.small) if e[0] is q]
                if not hit:
                    raise ValueError("FusedAdam(tail=True): the MLP's six tensors must be among the small parameters")
                idx.append(hit[0])
            lrs = {self.small[k][3] for k in idx}
            if len(lrs) != 1:
                raise ValueError("FusedAdam(tail=True): the MLP's six tensors must share one learning rate")
            self._tail = {"idx": idx, "lr": lrs.pop(), "net": mlp,
                          "p": (ctypes.c_void_p * 6)(), "m": (ctypes.c_void_p * 6)(*[self.small[k][1].data_ptr() for k in idx]),
                          "v": (ctypes.c_void_p * 6)(*[self.small[k][2].data_ptr() for k in idx]),
                          "maps": None if self._map_tensors is None else
                          (ctypes.c_void_p * 3)(*[self._map_tensors[n].data_ptr() for n in ("w1", "w2", "w3")])}
            self.fused.tail = True
            # no other small parameter and a device step counter: the armed scatter closes the step ITSELF (its pass 2 runs
            # the slab sums, the MLP's Adam step and the tick: lnerf_grid_encode_backward_adam_tail); else the tail is the
            # step's last launch (lnerf_step_tail, behind the ordinary Adam launch of the other small parameters)
            self.fused.inline_tail = len(idx) == len(self.small) and self.step_dev is not None

    def arm(self):
        """Let the NEXT backward through the encoder apply the table's Adam step (no-op without fuse_table_update)."""
        if self.fused is not None:
            self.fused.armed = True

    def zero_grad(self):
        """`optimizer.zero_grad()` of src/latent_paint/training/trainer.py:127 (gradients are dropped, not zeroed)."""
        for p, *_ in self.big + self.small:
            p.grad = None
        if self.fused is not None and (self.fused.armed or self.fused.applied):
            raise RuntimeError("FusedAdam.zero_grad(): a fused table update is pending or already applied this step")

    def step(self, grad_scale=None, set_to_none=True, grads=None, row_groups=None):
        """grads: optional {parameter: gradient tensor} overriding `.grad` for the big tensors -- f32, or the bf16
        wire buffer of GradSync (`GradSync.reduced()`), which the Adam kernel then reads directly.
        row_groups: optional {parameter: [(row_lo, row_hi, work[, after]), ...]} -- that parameter is stepped one row range
        at a time, each after `work.wait()` (a pipelined exchange: PendingExchange.table_groups); `work` may be None;
        `after()` is called right behind that range's launch (the sharded optimiser's all-gather of the new shadow rows).
        Ranges need not cover the parameter (a rank of the row-sharded optimiser steps only the rows it owns)."""
        if grad_scale is None:
            grad_scale = self.grad_scale
        elif self.fused is not None and float(grad_scale) != float(self.grad_scale):
            raise RuntimeError("FusedAdam: with fuse_table_update set `optimizer.grad_scale` before backward()")
        self.step_no += 1
        b1, b2 = self.betas
        enc = self.encoder
        for p, m, v, lr in self.big:
            if self.fused is not None and p is enc.embeddings and (self.fused.applied or self.fused.armed):
                # (a captured graph replays the single update it recorded)
                if self.fused.applied != 1 or self.fused.armed or p.grad is not None:
                    raise RuntimeError("FusedAdam: an armed step needs exactly one backward through the encoder "
                                       "(fused updates applied: %d, still armed: %s, extra table gradient: %s)"
                                       % (self.fused.applied, self.fused.armed, p.grad is not None))
                self.fused.applied = 0
                continue
            g = p.grad if grads is None else grads.get(p, p.grad)
            if g is None:
                continue
            if g.dtype not in (torch.float32, torch.bfloat16) or g.numel() != p.numel() or not g.is_contiguous():
                raise ValueError("FusedAdam: gradient must be a contiguous f32 or bf16 tensor of the parameter's size")
            shadow = enc.shadow() if (enc is not None and p is enc.embeddings) else None
            gdt = _b.F32 if g.dtype == torch.float32 else _b.BF16
            ranges = [(0, p.shape[0], None)] if (row_groups is None or p not in row_groups) else row_groups[p]
            for r0, r1, work, *rest in ranges:
                if work is not None:
                    work.wait()        # (orders the current stream behind that group's collective)
                sl = slice(r0, r1)
                _b.call("lnerf_adam_step", _p(p.data[sl]), _p(g[sl]), gdt, _p(m[sl]), _p(v[sl]),
                        None if shadow is None else _p(shadow[sl]), p.data[sl].numel(), lr, b1, b2, self.eps,
                        self.step_no, _p(self.step_dev), float(grad_scale), 0, _stream())
                if rest and rest[0] is not None:
                    rest[0]()
        pend = self.fused.pending_tail if self.fused is not None else None
        if self.fused is not None and self.fused.closed:
            self.fused.closed = False      # the armed backward closed the step: nothing left to launch
        elif pend is not None:
            self._step_tail(pend, float(grad_scale))
        elif self.small:
            for k, (p, m, v, lr) in enumerate(self.small):
                g = p.grad if grads is None else grads.get(p, p.grad)
                if g is None:
                    raise RuntimeError("FusedAdam: parameter %d has no gradient" % k)
                if g.dtype != torch.float32 or g.numel() != p.numel() or not g.is_contiguous():
                    raise ValueError("FusedAdam: the gradient of a small parameter must be a contiguous f32 tensor")
                self._pp[k] = p.data.data_ptr()
                self._gp[k] = g.data_ptr()
            # (with the device counter this launch, the step's last Adam launch, also advances it: LNERF_ADAM_TICK = 2)
            frag = None if self.mlp is None else self.mlp.mlp_workspace(self.small[0][0].device)
            _b.call("lnerf_adam_step_multi_shadow", len(self.small), self._pp, self._gp, self._mp, self._vp, self._n,
                    self._lr, b1, b2, self.eps, self.step_no, _p(self.step_dev), float(grad_scale),
                    2 if self.step_dev is not None else 0, None if frag is None else self._maps,
                    None if frag is None else _p(frag), _stream())
        elif self.step_dev is not None:
            _b.call("lnerf_adam_tick", _p(self.step_dev), _stream())
        # the multi-tensor launch wrote through raw pointers: tell torch, so that anything that cached a function of a
        # small parameter (the network's weight fragments) sees the change; the mirrored fragments are current for
        # exactly these versions.  (The table keeps its own bf16 shadow in step with the same Adam pass: its version
        # counter is what GridEncoder.shadow() compares, and stays.)
        for p, *_ in self.small:
            torch.autograd.graph.increment_version(p)
        if self.mlp is not None:
            self.mlp._frag_versions = self.mlp.weight_versions()
        if set_to_none:
            for p, *_ in self.big + self.small:
                p.grad = None

    def tail_args(self):
        """Host-side pointer arrays of the MLP's six tensors for the tail entry points (refreshed: `p.data` may move)."""
        t = self._tail
        for j, k in enumerate(t["idx"]):
            t["p"][j] = self.small[k][0].data.data_ptr()
        return t

    def _step_tail(self, pend, grad_scale):
        """The armed step's last launch (see __init__): lnerf_step_tail.  Small parameters outside the MLP (a background
        net) take the ordinary multi-tensor launch first; the tail ticks the step counter, so it goes last."""
        t = self._tail
        levels, m_host, variant, wst, mlp_ws, precision, out_dim = pend
        b1, b2 = self.betas
        rest = [k for k in range(len(self.small)) if k not in t["idx"]]
        if rest:
            n = len(rest)
            pp, gp = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
            mp, vp = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
            nn, lr = (ctypes.c_int64 * n)(), (ctypes.c_float * n)()
            for j, k in enumerate(rest):
                p, m, v, lr_k = self.small[k]
                if p.grad is None:
                    raise RuntimeError("FusedAdam: parameter %d has no gradient" % k)
                pp[j], gp[j], mp[j], vp[j] = p.data.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr()
                nn[j], lr[j] = p.numel(), lr_k
            _b.call("lnerf_adam_step_multi_shadow", n, pp, gp, mp, vp, nn, lr, b1, b2, self.eps, self.step_no,
                    _p(self.step_dev), grad_scale, 0, None, None, _stream())
        for j, k in enumerate(t["idx"]):
            t["p"][j] = self.small[k][0].data.data_ptr()
        fu = self.fused
        enc = self.encoder
        # (tick and clearing epilogue need the device counter pair: without it -- capturable=False -- the next scatter
        # call clears its level maxima itself)
        flags = (_b.TAIL_CLEAR_SCATTER | _b.TAIL_TICK) if self.step_dev is not None else 0
        _b.call("lnerf_step_tail", levels.num_levels, levels.level_dim, levels.c_offsets, levels.c_scales, levels.c_res,
                int(m_host), int(variant), _p(wst), wst.numel(), _p(fu.zero), _p(enc.embeddings.data), _p(fu.exp_avg),
                _p(fu.exp_avg_sq), _p(enc.shadow()), fu.lr, _p(mlp_ws), mlp_ws.numel(), int(precision), int(out_dim),
                t["p"], t["m"], t["v"], float(t["lr"]), t["maps"], b1, b2, self.eps, self.step_no, _p(self.step_dev),
                grad_scale, flags, _stream())
        fu.pending_tail = None
        from ..models import encoding as E
        if flags & _b.TAIL_CLEAR_SCATTER:
            E.ws_mark_clean(wst)

    def note_replayed_step(self):
        """A captured graph that contains step() was replayed: the device counter advanced, the host mirror follows
        (checkpoints store it)."""
        self.step_no += 1

    # ---- checkpointing (plain tensors only: loadable with torch.load(weights_only=True))
    def state_dict(self):
        entries = self.big + self.small
        return {"step": self.step_no, "exp_avg": [e[1] for e in entries], "exp_avg_sq": [e[2] for e in entries],
                "lr": [e[3] for e in entries]}

    def load_state_dict(self, state):
        entries = self.big + self.small
        if len(state["exp_avg"]) != len(entries):
            raise ValueError("optimizer state has %d tensors, expected %d" % (len(state["exp_avg"]), len(entries)))
        self.step_no = int(state["step"])
        for (p, m, v, lr), sm, sv in zip(entries, state["exp_avg"], state["exp_avg_sq"]):
            m.copy_(sm)
            v.copy_(sv)
        if self.step_dev is not None:
            self.step_dev[0:1].fill_(self.step_no + 1)
