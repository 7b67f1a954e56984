"""NeRFNetwork: hash-grid encoder + fused sigma/latent MLP (+ background net), HIP-backed.
Counterpart of src/latent_nerf/models/network_grid.py of the absent package (SURVEY.md
Appendix A: sigma_net = MLP(32, 1+C, hidden 64, 3 layers, bias), density blob, bg_net)."""
import math

import torch
import torch.nn as nn

from ..raymarching import backend as _b
from ..raymarching.raymarching import _chk, _p, _stream
from .encoding import GridEncoder
from .nerf_utils import NeRFType
from .renderer import NeRFRenderer

_PREC = {"f32": _b.F32, "bf16": _b.BF16}


class _SigmaLatentMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, xyzs, w1, b1, w2, b2, w3, b3, m_host, m_dev, level_stride, blob_scale, blob_std,
                precision, workspace):
        out_dim = w3.shape[0]
        dev = xyzs.device
        sigmas = torch.empty(level_stride, device=dev, dtype=torch.float32)
        rgbs = torch.empty(level_stride, out_dim - 1, device=dev, dtype=torch.float32)
        fdt = _b.F32 if feat.dtype == torch.float32 else _b.BF16
        _b.call("lnerf_mlp_forward", _chk(feat, "feat", feat.dtype), fdt, int(level_stride), _chk(xyzs, "xyzs"),
                _chk(w1, "w1"), _chk(b1, "b1"), _chk(w2, "w2"), _chk(b2, "b2"), _chk(w3, "w3"), _chk(b3, "b3"),
                out_dim, float(blob_scale), float(blob_std), int(m_host),
                _chk(m_dev, "m_dev", torch.int32, allow_none=True), _p(sigmas), _p(rgbs), precision, _p(workspace),
                0 if workspace is None else workspace.numel(), _stream())
        ctx.save_for_backward(feat, xyzs, w1, b1, w2, b2, w3, b3, sigmas,
                              m_dev if m_dev is not None else torch.empty(0))
        ctx.meta = (m_host, m_dev is not None, level_stride, blob_scale, blob_std, precision, workspace)
        ctx.set_materialize_grads(False)
        return sigmas, rgbs

    @staticmethod
    def backward(ctx, dsigmas, drgbs):
        feat, xyzs, w1, b1, w2, b2, w3, b3, sigmas, m_dev = ctx.saved_tensors
        m_host, has_mdev, level_stride, blob_scale, blob_std, precision, workspace = ctx.meta
        out_dim = w3.shape[0]
        dev = xyzs.device
        dsigmas = torch.zeros_like(sigmas) if dsigmas is None else dsigmas.contiguous()
        drgbs = torch.zeros(level_stride, out_dim - 1, device=dev) if drgbs is None else drgbs.contiguous()
        dfeat = torch.empty(feat.shape, device=dev, dtype=torch.float32)
        grads = [torch.empty_like(t) for t in (w1, b1, w2, b2, w3, b3)]  # overwritten (accumulate = 0)
        need = _b.get_lib().lnerf_mlp_backward_workspace_bytes(out_dim)
        if workspace is None or workspace.numel() < need:
            workspace = torch.empty(need, device=dev, dtype=torch.uint8)
        elif precision == _b.BF16:
            # this node's forward left the weight fragments at the head of the same workspace, and autograd's
            # version check on the saved weights guarantees they have not changed since
            precision |= _b.MLP_FRAGMENTS_READY
        fdt = _b.F32 if feat.dtype == torch.float32 else _b.BF16
        _b.call("lnerf_mlp_backward", _p(feat), fdt, int(level_stride), _p(xyzs), _p(w1), _p(b1), _p(w2), _p(b2),
                _p(w3), _p(b3), out_dim, float(blob_scale), float(blob_std), int(m_host),
                _p(m_dev) if has_mdev else None, _p(sigmas), _chk(dsigmas, "dsigmas"), _chk(drgbs, "drgbs"),
                _p(dfeat), *[_p(g) for g in grads], 0, _p(workspace), workspace.numel(), precision, None, 0, _stream())
        return (dfeat, None, *grads, None, None, None, None, None, None, None)


class _HashMLPField(torch.autograd.Function):
    """sigma, latent = MLP(hash_encode(xyzs)) as ONE autograd node: the level-major feature tensor (f32 or
    bf16) and its f32 gradient stay internal, so autograd never re-casts or copies them.
    forward : gather (lnerf_grid_encode_forward) -> MLP (lnerf_mlp_forward)
    backward: MLP backward (-> dfeat f32, weight grads) -> scatter (lnerf_grid_encode_backward)."""

    @staticmethod
    def forward(ctx, xyzs, table, shadow, w1, b1, w2, b2, w3, b3, encoder, bound, m_host, m_dev, level_stride,
                blob_scale, blob_std, precision, workspace, frag_ready):
        from . import encoding as E
        levels = encoder.levels
        src = table if shadow is None else shadow
        feat_dtype = torch.bfloat16 if precision == _b.BF16 else torch.float32
        feat = E.grid_encode_forward(xyzs, bound, src.detach(), levels, m_host, m_dev, level_stride, None, feat_dtype,
                                     encoder.variant)
        out_dim = w3.shape[0]
        dev = xyzs.device
        sigmas = torch.empty(level_stride, device=dev, dtype=torch.float32)
        rgbs = torch.empty(level_stride, out_dim - 1, device=dev, dtype=torch.float32)
        fdt = _b.F32 if feat.dtype == torch.float32 else _b.BF16
        _b.call("lnerf_mlp_forward", _p(feat), fdt, int(level_stride), _chk(xyzs, "xyzs"), _chk(w1, "w1"),
                _chk(b1, "b1"), _chk(w2, "w2"), _chk(b2, "b2"), _chk(w3, "w3"), _chk(b3, "b3"), out_dim,
                float(blob_scale), float(blob_std), int(m_host), _chk(m_dev, "m_dev", torch.int32, allow_none=True),
                _p(sigmas), _p(rgbs), precision | (_b.MLP_FRAGMENTS_READY if (frag_ready and precision == _b.BF16) else 0),
                _p(workspace), 0 if workspace is None else workspace.numel(), _stream())
        ctx.save_for_backward(xyzs, feat, w1, b1, w2, b2, w3, b3, sigmas,
                              m_dev if m_dev is not None else torch.empty(0))
        ctx.meta = (encoder, bound, m_host, m_dev is not None, level_stride, blob_scale, blob_std, precision, workspace,
                    table.shape)
        ctx.set_materialize_grads(False)
        return sigmas, rgbs

    @staticmethod
    def backward(ctx, dsigmas, drgbs):
        from . import encoding as E
        xyzs, feat, w1, b1, w2, b2, w3, b3, sigmas, m_dev = ctx.saved_tensors
        (encoder, bound, m_host, has_mdev, level_stride, blob_scale, blob_std, precision, workspace,
         tshape) = ctx.meta
        m_dev = m_dev if has_mdev else None
        out_dim = w3.shape[0]
        dev = xyzs.device
        dsigmas = torch.zeros_like(sigmas) if dsigmas is None else dsigmas.contiguous()
        drgbs = torch.zeros(level_stride, out_dim - 1, device=dev) if drgbs is None else drgbs.contiguous()
        dfeat = torch.empty(feat.shape, device=dev, dtype=torch.float32)
        need = _b.get_lib().lnerf_mlp_backward_workspace_bytes(out_dim)
        own_ws = workspace is not None and workspace.numel() >= need
        base_precision = precision
        if not own_ws:
            workspace = torch.empty(need, device=dev, dtype=torch.uint8)
        elif precision == _b.BF16:
            # this node's forward left the weight fragments at the head of the same workspace, and autograd's
            # version check on the saved weights guarantees they have not changed since
            precision |= _b.MLP_FRAGMENTS_READY
        fdt = _b.F32 if feat.dtype == torch.float32 else _b.BF16
        fu = encoder.fused_update
        sv = encoder.scatter_variant
        # TAIL mode (FusedAdam(tail=True), armed, m_host > 0): the slab sum and the Adam step of the six MLP tensors are
        # left to the scatter's own pass 2 (or, with other small parameters around, to ONE launch in optimizer.step():
        # lnerf_step_tail): no weight gradients here
        tail = fu is not None and fu.armed and fu.tail and own_ws and sv >= 2 and m_host > 0
        if tail:
            wst = E.scatter_workspace(encoder.levels, m_host, dev)
            _b.call("lnerf_mlp_backward", _p(feat), fdt, int(level_stride), _p(xyzs), _p(w1), _p(b1), _p(w2), _p(b2),
                    _p(w3), _p(b3), out_dim, float(blob_scale), float(blob_std), int(m_host), _p(m_dev), _p(sigmas),
                    _chk(dsigmas, "dsigmas"), _chk(drgbs, "drgbs"), _p(dfeat), None, None, None, None, None, None, 0,
                    _p(workspace), workspace.numel(), precision | _b.MLP_DEFER_REDUCE, None, 0, _stream())
            fu.take()
            flags = _b.SCATTER_CLEARED if E.ws_is_clean(wst) else 0
            if fu.inline_tail:   # the scatter's pass 2 closes the step: no launch behind it
                E.grid_encode_backward_adam_tail(xyzs, bound, dfeat, encoder, m_host, m_dev, level_stride, sv | flags,
                                                 workspace, base_precision, out_dim)
            else:
                E.grid_encode_backward_adam(xyzs, bound, dfeat, encoder, m_host, m_dev, level_stride, sv | flags)
                fu.pending_tail = (encoder.levels, int(m_host), int(sv), wst, workspace, int(base_precision), int(out_dim))
            return (None,) * 19
        # data parallel, pipelined exchange: the six weight gradients are written straight into their views of the flat
        # bucket the all-reduce sends (no `.grad`, no pack copy: one dispatch less per step)
        sink, direct = encoder.grad_sink, None
        if sink is not None and sink.groups and sink.small_direct:
            views = [sink.small_direct.get(t.data_ptr()) for t in (w1, b1, w2, b2, w3, b3)]
            if all(v is not None and v.shape == t.shape for v, t in zip(views, (w1, b1, w2, b2, w3, b3))):
                direct = views
        grads = direct if direct is not None else [torch.empty_like(t) for t in (w1, b1, w2, b2, w3, b3)]  # overwritten (accumulate = 0)
        # the bucketed scatter that follows needs its level maxima cleared: the MLP's slab-reduction launch does it on the
        # side (one dispatch less per step than the scatter's own fill)
        clear_ptr, clear_bytes = None, 0
        if sv >= 2 and m_host > 0:
            wst = E.scatter_workspace(encoder.levels, m_host, dev)
            clear_bytes = E.scatter_clear_bytes(encoder.levels, m_host)
            clear_ptr, sv = _p(wst), sv | _b.SCATTER_CLEARED
        _b.call("lnerf_mlp_backward", _p(feat), fdt, int(level_stride), _p(xyzs), _p(w1), _p(b1), _p(w2), _p(b2),
                _p(w3), _p(b3), out_dim, float(blob_scale), float(blob_std), int(m_host), _p(m_dev), _p(sigmas),
                _chk(dsigmas, "dsigmas"), _chk(drgbs, "drgbs"), _p(dfeat), *[_p(g) for g in grads], 0, _p(workspace),
                workspace.numel(), precision, clear_ptr, clear_bytes, _stream())
        if fu is not None and fu.take():  # armed: the scatter applies the table's Adam step
            E.grid_encode_backward_adam(xyzs, bound, dfeat, encoder, m_host, m_dev, level_stride, sv)
            dtable = None
        elif encoder.grad_sink is not None:  # data parallel: the gradient goes straight into the bf16 wire buffer
            E.grid_encode_backward_bf16(xyzs, bound, dfeat, encoder, m_host, m_dev, level_stride, sv)
            dtable = None
        else:
            dtable = torch.zeros(tshape, device=dev, dtype=torch.float32)
            E.grid_encode_backward(xyzs, bound, dfeat, encoder.levels, m_host, m_dev, level_stride, dtable, sv)
        if direct is not None:
            sink.small_written = True
            grads = [None] * 6
        return (None, dtable, None, *grads, None, None, None, None, None, None, None, None, None, None)


class NeRFNetwork(NeRFRenderer):
    def __init__(self, cfg, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                 hidden_dim=64, blob_scale=5.0, blob_std=0.2):
        super().__init__(cfg, latent_mode=cfg.nerf_type == NeRFType.latent)
        if hidden_dim != 64 or num_levels * level_dim != 32:
            raise ValueError("the fused HIP MLP is built for 32 -> 64 -> 64 -> 1+C")
        self.img_dims = 3 + 1 if self.latent_mode else 3
        self.blob_scale, self.blob_std = blob_scale, blob_std
        self.precision = cfg.precision("mlp_precision")
        table_dtype = torch.bfloat16 if cfg.precision("table_dtype") == "bf16" else torch.float32
        self.encoder = GridEncoder(num_levels, level_dim, base_resolution, 2048 * self.bound, log2_hashmap_size,
                                   table_dtype=table_dtype, variant=cfg.gather_variant,
                                   scatter_variant=(cfg.scatter_variant if cfg.scatter_variant >= 0
                                                    else (3 if self.precision == "bf16" else 2)),
                                   gridtype=cfg.layout() if hasattr(cfg, "layout") else getattr(cfg, "gridtype", "hash"))
        in_dim, out_dim = self.encoder.out_dim, 1 + self.img_dims
        # nn.Linear default init, kept as bare parameters: the fused kernel takes all six at once
        self.w1 = nn.Parameter(torch.empty(hidden_dim, in_dim))
        self.b1 = nn.Parameter(torch.empty(hidden_dim))
        self.w2 = nn.Parameter(torch.empty(hidden_dim, hidden_dim))
        self.b2 = nn.Parameter(torch.empty(hidden_dim))
        self.w3 = nn.Parameter(torch.empty(out_dim, hidden_dim))
        self.b3 = nn.Parameter(torch.empty(out_dim))
        for w, b in ((self.w1, self.b1), (self.w2, self.b2), (self.w3, self.b3)):
            bound = 1.0 / math.sqrt(w.shape[1])
            nn.init.uniform_(w, -bound, bound)
            nn.init.uniform_(b, -bound, bound)
        self.bg_radius = cfg.bg_radius
        if self.bg_radius > 0:
            self.bg_w1 = nn.Parameter(torch.empty(64, 39))
            self.bg_b1 = nn.Parameter(torch.empty(64))
            self.bg_w2 = nn.Parameter(torch.empty(self.img_dims, 64))
            self.bg_b2 = nn.Parameter(torch.empty(self.img_dims))
            for w, b in ((self.bg_w1, self.bg_b1), (self.bg_w2, self.bg_b2)):
                bound = 1.0 / math.sqrt(w.shape[1])
                nn.init.uniform_(w, -bound, bound)
                nn.init.uniform_(b, -bound, bound)
        self._mlp_ws = None
        # bf16 weight fragments at the head of the MLP workspace: which weights they were built from, in which buffer the
        # full image (constants included) was last built, and the optimiser that keeps them current (FusedAdam(mlp=...))
        self._frag_versions = None
        self._frag_built_in = None
        self._frag_owner = None

    def mlp_workspace(self, device):
        if self._mlp_ws is None or self._mlp_ws.device != device:
            need = _b.get_lib().lnerf_mlp_backward_workspace_bytes(self.w3.shape[0])
            self._mlp_ws = torch.empty(need, device=device, dtype=torch.uint8)
            self._frag_built_in = None
        return self._mlp_ws

    def weight_versions(self):
        return (self.w1._version, self.w2._version, self.w3._version,
                self.w1.data_ptr(), self.w2.data_ptr(), self.w3.data_ptr())

    def fragments_current(self):
        """True when the weight fragments in the workspace equal the weights: only claimed under an optimiser that
        mirrors its updates into them (a replayed hipGraph changes the weights without touching torch's version
        counters; with the mirroring optimiser inside the graph the fragments move with them) and while nothing else
        has modified the weights since (version counters: FusedAdam bumps them, torch's in-place ops do)."""
        return (self.precision == "bf16" and self._frag_owner is not None and self._mlp_ws is not None
                and self._frag_built_in == self._mlp_ws.data_ptr() and self._frag_versions == self.weight_versions())

    MAX_BF16_STRIDE = 1 << 24

    # ---- per-sample field -------------------------------------------------------------
    def field(self, xyzs, m_host, m_dev=None, level_stride=None):
        """xyzs [cap,3] -> sigmas [cap], latents [cap,C] for the first min(m_host, *m_dev) rows."""
        if level_stride is None:
            level_stride = xyzs.shape[0]
        if self.precision == "bf16" and level_stride > self.MAX_BF16_STRIDE and m_dev is None \
                and not torch.is_grad_enabled():
            # the bf16 kernels address features with 32-bit byte offsets (include/lnerf_hip.h: level_stride <= 2^24):
            # an inference batch beyond that is evaluated in pieces (training batches never get near it)
            outs = [self.field(xyzs[s:s + self.MAX_BF16_STRIDE], min(self.MAX_BF16_STRIDE, int(m_host) - s))
                    for s in range(0, int(m_host), self.MAX_BF16_STRIDE)]
            return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
        ws = self.mlp_workspace(xyzs.device)
        enc = self.encoder
        ready = self.fragments_current()
        sigmas, rgbs = _HashMLPField.apply(xyzs, enc.embeddings, enc.shadow(), self.w1, self.b1, self.w2, self.b2,
                                           self.w3, self.b3, enc, self.bound, m_host, m_dev, level_stride,
                                           self.blob_scale, self.blob_std, _PREC[self.precision], ws, ready)
        if not ready and self.precision == "bf16" and int(m_host) > 0:   # that forward built the whole image
            self._frag_built_in = ws.data_ptr()
            self._frag_versions = self.weight_versions()
        if not self.latent_mode:
            rgbs = torch.sigmoid(rgbs)
        return sigmas, rgbs

    def forward(self, x, d=None):
        x = x.reshape(-1, 3).contiguous().float()
        return self.field(x, x.shape[0])

    def density(self, x):
        sigmas, rgbs = self.forward(x)
        return {"sigma": sigmas, "albedo": rgbs}

    def background(self, d):
        from .bg import background_net
        return background_net(d, self.bg_w1, self.bg_b1, self.bg_w2, self.bg_b2)

    def get_params(self, lr):
        params = [{"params": [self.encoder.embeddings], "lr": lr * 10},
                  {"params": [self.w1, self.b1, self.w2, self.b2, self.w3, self.b3], "lr": lr}]
        if self.bg_radius > 0:
            params.append({"params": [self.bg_w1, self.bg_b1, self.bg_w2, self.bg_b2], "lr": lr})
        return params
