"""Multiresolution hash-grid encoder (Instant-NGP), HIP-backed.  Counterpart of the `gridencoder`
extension of the absent src/latent_nerf/models/encoders; the level
table follows the align_corners=False convention recorded there."""
import ctypes
import math

import numpy as np
import torch
import torch.nn as nn

from ..raymarching import backend as _b
from ..raymarching.raymarching import _chk, _p, _stream


class GridLevels:
    """Host-side level table shared with the kernels (offsets in rows, per-level scale, resolution)."""

    def __init__(self, num_levels=16, level_dim=2, base_resolution=16, desired_resolution=2048,
                 log2_hashmap_size=19, gridtype="hash"):
        if gridtype not in ("hash", "tiled", "blocked"):
            raise ValueError("gridtype must be 'hash' (Instant-NGP), 'tiled' (the upstream encoder's other layout: dense "
                             "index wrapped) or 'blocked' (4 x 2 x 2 vertex blocks per 64-byte line)")
        # flag OR-ed into the `variant` of every gather / scatter call (include/lnerf_hip.h LNERF_GRID_BLOCKED / _TILED)
        self.gridtype = gridtype
        self.flag = {"hash": 0, "blocked": _b.GRID_BLOCKED, "tiled": _b.GRID_TILED}[gridtype]
        if level_dim != 2:
            raise ValueError("only level_dim == 2 is built")
        if not (1 <= num_levels <= 32):
            raise ValueError("num_levels must be in [1, 32]")
        self.num_levels, self.level_dim = num_levels, level_dim
        self.base_resolution, self.desired_resolution = base_resolution, desired_resolution
        self.log2_hashmap_size = log2_hashmap_size
        max_params = 2 ** log2_hashmap_size
        pls = 2.0 ** (math.log2(desired_resolution / base_resolution) / (num_levels - 1)) if num_levels > 1 else 1.0
        S = math.log2(pls)
        offsets, scales, ress = [0], [], []
        for l in range(num_levels):
            res_host = int(math.ceil(base_resolution * pls ** l))
            n = min(max_params, (res_host + 1) ** 3)
            n = int(math.ceil(n / 8) * 8)
            offsets.append(offsets[-1] + n)
            scale = float(np.float32(2.0 ** (l * S) * base_resolution - 1.0))
            scales.append(scale)
            ress.append(int(math.ceil(scale)) + 1)
        self.offsets, self.scales, self.resolutions = offsets, scales, ress
        self.n_rows = offsets[-1]
        self.out_dim = num_levels * level_dim
        # ctypes views handed to the C ABI on every call
        self.c_offsets = (ctypes.c_int32 * (num_levels + 1))(*offsets)
        self.c_scales = (ctypes.c_float * num_levels)(*scales)
        self.c_res = (ctypes.c_int32 * num_levels)(*ress)


def grid_encode_forward(xyzs, bound, table, levels: GridLevels, m_host, m_dev, level_stride, out=None,
                        out_dtype=torch.float32, variant=0):
    """xyzs [>=m_host,3] world positions -> level-major features [L, level_stride, 2]."""
    tdt = _b.F32 if table.dtype == torch.float32 else _b.BF16
    if table.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("table must be float32 or bfloat16")
    if out is None:
        out = torch.empty(levels.num_levels, level_stride, 2, device=xyzs.device, dtype=out_dtype)
    odt = _b.F32 if out.dtype == torch.float32 else _b.BF16
    _b.call("lnerf_grid_encode_forward", _chk(xyzs, "xyzs"), float(bound), _chk(table, "table", table.dtype), tdt,
            levels.num_levels, levels.level_dim, levels.c_offsets, levels.c_scales, levels.c_res, int(m_host),
            _chk(m_dev, "m_dev", torch.int32, allow_none=True), int(level_stride), _chk(out, "feat", out.dtype), odt,
            int(variant) | levels.flag, _stream())
    return out


_scatter_ws = {}   # device -> [workspaces, ascending size]


def scatter_workspace(levels: GridLevels, m_host, device):
    """Device scratch of the bucketed scatter: ANY buffer of at least lnerf_grid_encode_backward_workspace_bytes()
    serves (the kernels lay it out from `m_host`, not from its size), so a process keeps one buffer per device that only
    ever grows: a renderer whose sample budget moves up and down in 64 Ki steps (NeRFRenderer.update_sample_budget)
    re-uses the largest one instead of allocating one per distinct capacity.  A captured hipGraph may hold the address
    of a buffer handed out earlier, so outgrown ones are kept, but every growth is by >= 1.5x: the total stays below
    three times the largest request."""
    need = _b.get_lib().lnerf_grid_encode_backward_workspace_bytes(levels.num_levels, levels.c_offsets, int(m_host))
    if need == 0:
        raise _b.LnerfError("bucketed scatter cannot handle this level table; use scatter variant 0/1")
    have = _scatter_ws.setdefault(str(device), [])
    for ws in have:
        if ws.numel() >= need:
            return ws
    size = max(need, (3 * have[-1].numel()) // 2 if have else 0)
    ws = torch.empty(size, device=device, dtype=torch.uint8)
    # header: level maxima and arrival counters start out clean (LNERF_SCATTER_ZERO_HEAD_BYTES; the calls keep them so)
    ws[:min(size, _b.SCATTER_ZERO_HEAD_BYTES)].zero_()
    have.append(ws)
    return ws


# Whether the level maxima at the head of a device's (shared) scatter workspace are known to be ZERO, and a counter of the
# scatter calls issued through it from Python.  The workspace is one buffer per DEVICE, so this is tracked per device, not
# per encoder: a second model -- or a backward through the same net outside the training step -- leaves its maxima in the
# header, and a step that passes LNERF_SCATTER_CLEARED on the strength of its OWN previous call would scale with stale,
# larger maxima (no overflow, but fixed-point precision drops and graph == eager is lost).  A captured hipGraph bakes the
# flag in: whoever replays one compares scatter_workspace_epoch() with the value it noted at capture time and captures
# again when a foreign call went through in between (Trainer.train()).
_ws_state = {}   # str(device) -> [data_ptr of the workspace whose maxima are zero | None, epoch]


def _ws(device):
    return _ws_state.setdefault(str(device), [None, 0])


def ws_mark_dirty(device):
    st = _ws(device)
    st[0], st[1] = None, st[1] + 1


def ws_mark_clean(wst):
    st = _ws(wst.device)
    st[0], st[1] = wst.data_ptr(), st[1] + 1


def ws_is_clean(wst):
    return _ws(wst.device)[0] == wst.data_ptr()


def scatter_workspace_epoch(device):
    """(clean workspace pointer, number of scatter calls issued from Python on this device): changes whenever anybody
    scatters through the device's workspace outside a graph replay."""
    return tuple(_ws(device))


_clear_bytes = {}


def scatter_clear_bytes(levels: GridLevels, m_host):
    """Bytes at the head of the scatter workspace that a call clears (lnerf_grid_scatter_clear_bytes); a caller that
    zeroes them itself passes `variant | backend.SCATTER_CLEARED`."""
    key = (tuple(levels.offsets), int(m_host))
    n = _clear_bytes.get(key)
    if n is None:
        n = int(_b.get_lib().lnerf_grid_scatter_clear_bytes(levels.num_levels, levels.c_offsets, int(m_host)))
        _clear_bytes[key] = n
    return n


def grid_encode_backward(xyzs, bound, dfeat, levels: GridLevels, m_host, m_dev, level_stride, dtable, variant=2):
    """dtable (f32 [rows,2]) += scatter of dfeat (level-major, f32).
    variant 0/1: global float atomics; 2: two-pass bucketed scatter (LDS reduction); 3: the same with packed
    8-byte records (values rounded to 17 mantissa bits)."""
    ws, ws_bytes = None, 0
    if (variant & 0xFF) >= 2:
        wst = scatter_workspace(levels, m_host, xyzs.device)
        ws, ws_bytes = _p(wst), wst.numel()
        ws_mark_dirty(xyzs.device)      # (this call leaves its level maxima in the shared workspace)
    _b.call("lnerf_grid_encode_backward", _chk(xyzs, "xyzs"), float(bound), _chk(dfeat, "dfeat"), _b.F32,
            levels.num_levels, levels.level_dim, levels.c_offsets, levels.c_scales, levels.c_res, int(m_host),
            _chk(m_dev, "m_dev", torch.int32, allow_none=True), int(level_stride), _chk(dtable, "dtable"),
            int(variant) | levels.flag, ws, ws_bytes, _stream())
    return dtable


class FusedTableUpdate:
    """Set on a GridEncoder (`encoder.fused_update`) by FusedAdam(fuse_table_update=True).  When ARMED
    (FusedAdam.arm(), once per step, right before the step's single backward) the scatter of that backward applies
    the table's Adam step itself (lnerf_grid_encode_backward_adam) and the table gets no `.grad`; any other
    backward through the encoder (unarmed) produces the ordinary gradient.  Single GPU only."""

    def __init__(self, exp_avg, exp_avg_sq, lr, betas, eps, optimizer):
        self.exp_avg, self.exp_avg_sq = exp_avg, exp_avg_sq
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        self.optimizer = optimizer          # step number / device step counter live there
        self.zero = torch.zeros_like(exp_avg)  # dtable scratch argument of the fused entry points (never written)
        self.applied = 0                    # fused updates since the last optimizer.step()
        self.armed = False
        # tail mode (FusedAdam(tail=True)): the armed backward leaves the finishing pass of the scatter and the sum of
        # the MLP's gradient slabs to ONE launch in optimizer.step() (lnerf_step_tail), which also steps the MLP's
        # parameters, ticks the step counter and leaves the scatter's level maxima zero for the next step
        self.tail = False
        self.pending_tail = None            # (levels, m_host, variant, scatter workspace, mlp workspace, precision, out_dim)
        # the armed backward closed the step itself (grid_encode_backward_adam_tail): slab sums, the MLP's Adam step, the
        # tick of the step counter all ran inside the scatter's pass 2
        self.closed = False
        self.inline_tail = False            # set by FusedAdam: the tail may run inside the scatter (no other small parameter)

    def take(self):
        """True once per arm(): the caller (a backward pass) then owes the fused update."""
        if self.armed:
            self.armed = False
            return True
        return False


def grid_encode_backward_adam(xyzs, bound, dfeat, encoder, m_host, m_dev, level_stride, variant):
    """Scatter of dfeat fused with the Adam step of encoder.embeddings (see include/lnerf_hip.h)."""
    fu = encoder.fused_update
    levels = encoder.levels
    if (variant & 0xFF) < 2:
        raise _b.LnerfError("the fused table update needs the bucketed scatter (variant 2 or 3)")
    wst = scatter_workspace(levels, m_host, xyzs.device)
    opt = fu.optimizer
    table = encoder.embeddings.data
    shadow = encoder.shadow()
    b1, b2 = fu.betas
    _b.call("lnerf_grid_encode_backward_adam", _chk(xyzs, "xyzs"), float(bound), _chk(dfeat, "dfeat"), _b.F32,
            levels.num_levels, levels.level_dim, levels.c_offsets, levels.c_scales, levels.c_res, int(m_host),
            _chk(m_dev, "m_dev", torch.int32, allow_none=True), int(level_stride), _p(fu.zero), int(variant) | levels.flag,
            _p(wst), wst.numel(), _p(table), _p(fu.exp_avg), _p(fu.exp_avg_sq), _p(shadow), fu.lr, b1, b2, fu.eps,
            opt.step_no + 1, _p(opt.step_dev), float(opt.grad_scale), _stream())
    fu.applied += 1
    ws_mark_dirty(xyzs.device)


def grid_encode_backward_adam_tail(xyzs, bound, dfeat, encoder, m_host, m_dev, level_stride, variant, mlp_ws, precision,
                                   out_dim):
    """The armed scatter that also CLOSES the step (lnerf_grid_encode_backward_adam_tail): the workgroups of its pass 2
    sum the MLP's gradient slabs, step the MLP's six tensors, advance the device step counter and leave the level maxima
    zero for the next step -- no launch behind the scatter.  FusedAdam(tail=True) with no other small parameter."""
    fu = encoder.fused_update
    levels = encoder.levels
    opt = fu.optimizer
    t = opt.tail_args()
    wst = scatter_workspace(levels, m_host, xyzs.device)
    b1, b2 = fu.betas
    _b.call("lnerf_grid_encode_backward_adam_tail", _chk(xyzs, "xyzs"), float(bound), _chk(dfeat, "dfeat"), _b.F32,
            levels.num_levels, levels.level_dim, levels.c_offsets, levels.c_scales, levels.c_res, int(m_host),
            _chk(m_dev, "m_dev", torch.int32, allow_none=True), int(level_stride), _p(fu.zero), int(variant) | levels.flag,
            _p(wst), wst.numel(), _p(encoder.embeddings.data), _p(fu.exp_avg), _p(fu.exp_avg_sq), _p(encoder.shadow()),
            fu.lr, _p(mlp_ws), mlp_ws.numel(), int(precision), int(out_dim), t["p"], t["m"], t["v"], float(t["lr"]),
            t["maps"], b1, b2, fu.eps, opt.step_no + 1, _p(opt.step_dev), float(opt.grad_scale),
            _b.TAIL_TICK | _b.TAIL_CLEAR_SCATTER, _stream())
    fu.applied += 1
    fu.closed = True                 # optimizer.step() has nothing left to launch
    ws_mark_clean(wst)               # (level maxima zero again when the launch ends)


class GradSink:
    """Set on a GridEncoder (`encoder.grad_sink`) by GradSync.attach_sink(): the backward pass WRITES the table
    gradient as bf16 into `wire` (lnerf_grid_encode_backward_bf16) -- the buffer the data-parallel all-reduce sends --
    instead of accumulating an f32 `.grad`: no zero fill, no read-modify-write, no cast.  One backward per step.

    `groups` (list of (level_lo, level_hi)) selects the PIPELINED form: the backward pass only bins the records
    (lnerf_grid_scatter_bin); GradSync.allreduce_pipelined() then sums one level group at a time
    (lnerf_grid_scatter_reduce_bf16) and launches that group's all-reduce while the next group is being summed."""

    def __init__(self, table, groups=None):
        self.wire = torch.zeros(table.shape, device=table.device, dtype=torch.bfloat16)
        self.zero = torch.zeros(table.shape, device=table.device, dtype=torch.float32)  # overflow records only
        self.written = 0
        self.groups = groups
        self.pending = None   # (bound, levels, m_host, level_stride, variant, workspace) of a binned, not yet summed backward
        # {data_ptr of a small parameter: its view of GradSync's flat bucket}: a backward pass may write such a gradient
        # there directly (and sets small_written, for good: replays of a captured backward repeat the write without any
        # Python); None: everything goes through `.grad`
        self.small_direct = None
        self.small_written = False
        self.shard_ranges, self.shard_rest = None, None   # row-sharded optimiser (GradSync._plan_shards)


def level_groups(levels: GridLevels, n_groups=4):
    """Level ranges of roughly equal table rows (the exchange is priced per row): [(lo, hi), ...]."""
    n_groups = max(1, min(int(n_groups), levels.num_levels))
    target = levels.n_rows / n_groups
    out, lo = [], 0
    for l in range(levels.num_levels):
        done_rows = levels.offsets[l + 1]
        if done_rows >= target * (len(out) + 1) - 1e-9 or l == levels.num_levels - 1:
            if len(out) < n_groups - 1 or l == levels.num_levels - 1:
                out.append((lo, l + 1))
                lo = l + 1
    if lo < levels.num_levels:
        out.append((lo, levels.num_levels))
    return out


def grid_encode_backward_bf16(xyzs, bound, dfeat, encoder, m_host, m_dev, level_stride, variant):
    sink = encoder.grad_sink
    levels = encoder.levels
    if variant < 2:
        raise _b.LnerfError("the bf16 gradient output needs the bucketed scatter (variant 2 or 3)")
    wst = scatter_workspace(levels, m_host, xyzs.device)
    ws_mark_dirty(xyzs.device)
    if sink.groups:   # pipelined: pass 1 now, pass 2 per level group inside GradSync.allreduce_pipelined()
        _b.call("lnerf_grid_scatter_bin", _chk(xyzs, "xyzs"), float(bound), _chk(dfeat, "dfeat"), _b.F32,
                levels.num_levels, levels.level_dim, levels.c_offsets, levels.c_scales, levels.c_res, int(m_host),
                _chk(m_dev, "m_dev", torch.int32, allow_none=True), int(level_stride), _p(sink.zero),
                int(variant) | levels.flag, _p(wst), wst.numel(), _stream())
        sink.pending = (float(bound), levels, int(m_host), int(level_stride), int(variant), wst)
        return
    _b.call("lnerf_grid_encode_backward_bf16", _chk(xyzs, "xyzs"), float(bound), _chk(dfeat, "dfeat"), _b.F32,
            levels.num_levels, levels.level_dim, levels.c_offsets, levels.c_scales, levels.c_res, int(m_host),
            _chk(m_dev, "m_dev", torch.int32, allow_none=True), int(level_stride), _p(sink.zero),
            int(variant) | levels.flag, _p(wst), wst.numel(), _p(sink.wire), _stream())
    sink.written += 1


def grid_scatter_reduce_group(sink: GradSink, level_lo, level_hi):
    """Pass 2 (+ finishing pass) of levels [level_lo, level_hi) of the backward that sink.pending describes: writes rows
    offsets[level_lo] .. offsets[level_hi] of sink.wire."""
    if sink.pending is None:
        raise _b.LnerfError("no binned backward pass is pending on this gradient sink")
    bound, levels, m_host, level_stride, variant, wst = sink.pending
    _b.call("lnerf_grid_scatter_reduce_bf16", bound, levels.num_levels, levels.level_dim, levels.c_offsets,
            levels.c_scales, levels.c_res, m_host, level_stride, int(level_lo), int(level_hi), _p(sink.zero), variant,
            _p(wst), wst.numel(), _p(sink.wire), _stream())


class _GridEncode(torch.autograd.Function):
    """feat = encode(xyzs; table).  `table` is the f32 master parameter (gradient target);
    `shadow` an optional bf16 copy that the gather actually reads."""

    @staticmethod
    def forward(ctx, xyzs, table, shadow, levels, bound, m_host, m_dev, level_stride, feat_dtype, variant,
                scatter_variant, encoder=None):
        src = table if shadow is None else shadow
        feat = grid_encode_forward(xyzs, bound, src.detach(), levels, m_host, m_dev, level_stride, None, feat_dtype,
                                   variant)
        ctx.save_for_backward(xyzs, m_dev if m_dev is not None else torch.empty(0))
        ctx.has_mdev = m_dev is not None
        ctx.meta = (levels, bound, m_host, level_stride, scatter_variant, table.shape, table.device)
        ctx.encoder = encoder
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        xyzs, m_dev = ctx.saved_tensors
        levels, bound, m_host, level_stride, variant, shape, dev = ctx.meta
        dfeat = dfeat.contiguous()
        if dfeat.dtype != torch.float32:
            dfeat = dfeat.float()
        enc = ctx.encoder
        if enc is not None and enc.fused_update is not None and enc.fused_update.take():
            grid_encode_backward_adam(xyzs, bound, dfeat, enc, m_host, m_dev if ctx.has_mdev else None, level_stride,
                                      variant)
            return (None,) * 12
        if enc is not None and enc.grad_sink is not None:
            grid_encode_backward_bf16(xyzs, bound, dfeat, enc, m_host, m_dev if ctx.has_mdev else None, level_stride,
                                      variant)
            return (None,) * 12
        dtable = torch.zeros(shape, device=dev, dtype=torch.float32)
        grid_encode_backward(xyzs, bound, dfeat, levels, m_host, m_dev if ctx.has_mdev else None, level_stride,
                             dtable, variant)
        return None, dtable, None, None, None, None, None, None, None, None, None, None


class GridEncoder(nn.Module):
    """embeddings: f32 [n_rows, 2] master table (U(-1e-4, 1e-4) init).  With
    `table_dtype=torch.bfloat16` the gather reads a bf16 shadow that is refreshed lazily whenever
    the master changed (or explicitly by the fused Adam step)."""

    def __init__(self, num_levels=16, level_dim=2, base_resolution=16, desired_resolution=2048,
                 log2_hashmap_size=19, table_dtype=torch.float32, variant=0, scatter_variant=2, gridtype="hash"):
        super().__init__()
        self.levels = GridLevels(num_levels, level_dim, base_resolution, desired_resolution, log2_hashmap_size,
                                 gridtype=gridtype)
        self.out_dim = self.levels.out_dim
        self.variant = variant
        self.scatter_variant = scatter_variant
        self.table_dtype = table_dtype
        self.embeddings = nn.Parameter(torch.empty(self.levels.n_rows, level_dim))
        self.reset_parameters()
        self._shadow = None
        self._shadow_version = -1
        self.fused_update = None  # FusedTableUpdate, installed by FusedAdam(fuse_table_update=True)
        self.grad_sink = None     # GradSink, installed by GradSync.attach_sink() (data parallel, bf16 on the wire)

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def shadow(self):
        if self.table_dtype == torch.float32:
            return None
        ver = self.embeddings._version
        if self._shadow is None or self._shadow.device != self.embeddings.device:
            self._shadow = torch.empty(self.levels.n_rows, 2, device=self.embeddings.device, dtype=torch.bfloat16)
            self._shadow_version = -1
        if self._shadow_version != ver:
            _b.call("lnerf_cast_f32_to_bf16", _p(self.embeddings.data), _p(self._shadow),
                    self.embeddings.numel(), _stream())
            self._shadow_version = ver
        return self._shadow

    def mark_shadow_fresh(self):
        """Called by the fused optimiser step, which rewrites the shadow in the same pass."""
        self._shadow_version = self.embeddings._version

    def encode(self, xyzs, bound, m_host, m_dev=None, level_stride=None, feat_dtype=torch.float32):
        """Level-major features [L, level_stride, 2] (differentiable w.r.t. embeddings)."""
        if level_stride is None:
            level_stride = xyzs.shape[0]
        return _GridEncode.apply(xyzs, self.embeddings, self.shadow(), self.levels, bound, m_host, m_dev, level_stride,
                                 feat_dtype, self.variant, self.scatter_variant, self)

    def forward(self, inputs, bound=1.0):
        """inputs [..., 3] in [-bound, bound] -> [..., L*2] (sample-major view, as the upstream encoder)."""
        prefix = inputs.shape[:-1]
        x = inputs.reshape(-1, 3).contiguous().float()
        M = x.shape[0]
        feat = self.encode(x, bound, M)
        return feat.permute(1, 0, 2).reshape(*prefix, self.out_dim)
