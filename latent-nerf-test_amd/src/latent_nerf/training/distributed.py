"""Data-parallel plumbing of the render path: one process per GPU, one view per rank per step,
gradients summed with one collective exchange per step.

The reference is single-process (no collective anywhere); this is new work that only
uses `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm, "gloo" for the CPU tests).  Every rank
holds a full replica of table + MLP + occupancy grid, renders its own pose, and after
`GradSync.allreduce()` applies the identical optimiser step, so replicas stay bit-identical."""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def force_exchange() -> bool:
    """LNERF_FORCE_DIST=1: run the data-parallel code path -- process group, bf16 gradient sink, pipelined per-group
    all-reduce, row-group Adam -- at ANY world size, one rank included.  One rank on one card is how the RCCL path is
    exercised on a single-GPU box (the collectives are real RCCL calls over a communicator of size 1); it is also a
    same-box A/B of the un-fused step against the fused single-GPU step."""
    import os
    return os.environ.get("LNERF_FORCE_DIST", "0") not in ("", "0")


def exchange_active(group=None) -> bool:
    """True when gradients are exchanged: more than one rank, or a forced exchange on an initialised group."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or force_exchange()


def backend_name(group=None) -> str:
    """"nccl" (= RCCL), "gloo", ... of the initialised process group; "" without one."""
    if not (dist.is_available() and dist.is_initialized()):
        return ""
    return str(dist.get_backend(group))


def init_distributed():
    """Call FIRST in a training process, before anything touches the GPU: binds this rank to its device and, when
    the launcher (`python -m torch.distributed.run --nproc-per-node N ...`) set WORLD_SIZE > 1 (or LNERF_FORCE_DIST=1,
    see force_exchange), joins the process group -- backend "nccl" (= RCCL over xGMI on ROCm), one process per GPU.
    LNERF_DIST_BACKEND=gloo lets several ranks share one card for functional rehearsals (the collectives then stage
    through the host).  Returns (rank, world, device)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("LNERF_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()          # (counting devices does not initialise the GPU)
    if ndev == 0:
        raise RuntimeError("no GPU visible: the render path runs on the HIP library only")
    local_dev = local if backend == "nccl" else local % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if (world > 1 or force_exchange()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # a launcher (torch.distributed.run, training/launch.spawn_ranks) picks the port for a job of several ranks;
            # only a forced one-rank exchange gets here without one: any free port serves
            if world > 1:
                raise RuntimeError("WORLD_SIZE > 1 without MASTER_PORT: start the ranks with a launcher")
            from .launch import free_port
            os.environ["MASTER_PORT"] = str(free_port())
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    return rank, world, dev


def broadcast_parameters(params, src=0, group=None):
    """Replicas start from rank `src`'s values (one flat bucket for the small tensors, big ones on their own)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for p in params:
        dist.broadcast(p.data, src=src, group=group)
        # `p.data` has its own version counter: tell autograd (and whoever caches a function of the parameter -- the bf16
        # table shadow, the MLP weight fragments -- keyed on `p._version`) that the values changed
        torch.autograd.graph.increment_version(p)


def views_for_rank(views_per_step: int, rank: int, world: int) -> List[int]:
    """Indices (within a step's batch of random views) rendered by `rank`: round-robin, so that
    8 views/step on 8 GPUs is 1 view per GPU (BASELINE config 4)."""
    if views_per_step % world != 0:
        raise ValueError("views_per_step (%d) must be a multiple of the world size (%d)" % (views_per_step, world))
    return list(range(rank, views_per_step, world))


def pose_generator(seed: int, step: int, view_index: int) -> torch.Generator:
    """Deterministic per-(step, view) RNG: every rank can reproduce any view's pose without
    communication, and a run with W ranks renders the same set of views as a 1-rank run."""
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1_000_003 + step) * 4099 + view_index)
    return g


_M64 = (1 << 64) - 1


def pose_uniforms(seed: int, step: int, view_index: int, n: int = 4):
    """n uniforms in [0, 1) (24 bits each: exact in float32) of (seed, step, view): the counter-based form of
    pose_generator -- a 64-bit mix (splitmix64's finaliser) of the three counters in plain integer arithmetic, ~2 us
    where seeding a torch.Generator and drawing from it costs ~40 us of the trainer's per-step host time.  Every rank
    can reproduce any view's pose without communication; W ranks render the same set of views as one."""
    out = []
    base = (int(seed) * 0x9E3779B97F4A7C15 + int(step) * 0xD1B54A32D192ED03 + int(view_index) * 0x8CB92BA72F3D8DD7) & _M64
    for j in range(n):
        x = (base + (j + 1) * 0x9E3779B97F4A7C15) & _M64
        x ^= x >> 30
        x = (x * 0xBF58476D1CE4E5B9) & _M64
        x ^= x >> 27
        x = (x * 0x94D049BB133111EB) & _M64
        x ^= x >> 31
        out.append((x >> 40) / 16777216.0)
    return out


class GradSync:
    """Sums gradients across ranks.  `big` tensors (the hash table gradient, 46.7 MiB) are all-reduced
    each as its own bucket; all `small` parameters travel in ONE flat bucket.

    xGMI is point-to-point (7 links per GPU): a few large collectives beat many small ones, hence
    exactly two collectives per step regardless of the number of parameter tensors.

    transport: torch.float32 (exact sum, default) or torch.bfloat16 for the big buckets -- half the bytes
    on the links; the sum is then formed in bf16 by the collective (the usual bf16-gradient trade-off of
    mixed-precision data parallelism).  Small parameters always travel in f32."""

    def __init__(self, big: Iterable[torch.nn.Parameter], small: Iterable[torch.nn.Parameter],
                 group: Optional[dist.ProcessGroup] = None, transport: torch.dtype = torch.float32,
                 shard_optimizer: bool = False):
        """shard_optimizer (pipelined bf16 exchange only, see allreduce_pipelined): the table's optimiser state is
        sharded by rows -- reduce-scatter of the bf16 gradient, the OWNING rank steps its 1/N of the rows, all-gather of
        the bf16 shadow the gather reads.  Same wire bytes as the all-reduce (which is a reduce-scatter + an all-gather
        of the gradient), a 1/N share of the Adam pass's 30 bytes per table entry, and the direct RS + AG shape xGMI's
        point-to-point links favour.  The f32 master table and the moments are then only
        current on their owner: gather_rows() before anything reads them whole (checkpoints)."""
        self.big = list(big)
        self.small = list(small)
        self.group = group
        if transport not in (torch.float32, torch.bfloat16):
            raise ValueError("transport must be float32 or bfloat16")
        self.transport = transport
        self._flat = None
        self._flat_views = []
        self._wire = {}
        self._sinks = {}
        self.shard_optimizer = bool(shard_optimizer)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # gradients are exchanged: more than one rank -- or a forced exchange (the collectives then run over a
        # communicator of size 1; every buffer, launch and wait of the N > 1 step is the same)
        self.active = exchange_active(group)

    def _flat_buffer(self):
        n = sum(p.numel() for p in self.small)
        if self._flat is None or self._flat.numel() != n or self._flat.device != self.small[0].device:
            self._flat = torch.empty(n, device=self.small[0].device, dtype=torch.float32)
            self._flat_views, o = [], 0
            for p in self.small:   # one view per small parameter, in bucket order
                self._flat_views.append(self._flat[o:o + p.numel()].view(p.shape))
                o += p.numel()
        return self._flat

    def _pack_small(self):
        """The small parameters' gradients into the flat bucket: ONE multi-tensor copy (a copy per parameter is a
        dependent ~5 us dispatch each, twice per step) -- and none at all for the tensors whose backward wrote its
        gradient straight into its view of the bucket (GradSink.small_direct: the MLP's six tensors)."""
        flat = self._flat_buffer()
        # (small_written is sticky: a replayed hipGraph writes the views again without running any Python that could
        # raise the flag per step)
        direct = {}
        for sink in self._sinks.values():
            if sink.small_written:
                direct.update(sink.small_direct or {})
        dst, src = [], []
        for p, view in zip(self.small, self._flat_views):
            if p.grad is None:
                if direct.get(p.data_ptr()) is view:
                    continue
                raise RuntimeError("GradSync: a parameter has no gradient on this rank")
            dst.append(view)
            src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        return flat

    def _unpack_small(self):
        torch._foreach_copy_([p.grad for p in self.small], self._flat_views)

    def _wire_buffer(self, p):
        buf = self._wire.get(id(p))
        if buf is None or buf.shape != p.shape or buf.device != p.device:
            buf = torch.empty(p.shape, device=p.device, dtype=self.transport)
            self._wire[id(p)] = buf
        return buf

    def attach_sink(self, encoder, pipeline_groups=0):
        """bf16 transport, more than one rank: let the encoder's backward write the table gradient straight into the
        wire buffer this object all-reduces (GradSink): the table then has no `.grad`, use reduced() for the sums.
        No-op otherwise.  One backward per step.
        pipeline_groups >= 1: the table is exchanged in that many level groups by allreduce_pipelined() (0: one
        collective through allreduce(copy_back=False))."""
        if not self.active or self.transport == torch.float32:
            return None
        from ..models.encoding import GradSink, level_groups
        p = encoder.embeddings
        if not any(p is q for q in self.big):
            raise ValueError("attach_sink: encoder.embeddings is not one of the big buckets")
        sink = GradSink(p.data, groups=level_groups(encoder.levels, pipeline_groups) if pipeline_groups >= 1 else None)
        self._sink_levels = getattr(encoder, "levels", None)
        self._wire[id(p)] = sink.wire
        self._sinks[id(p)] = sink
        if sink.groups and self.small:
            # pipelined form: a backward pass that produces small-parameter gradients itself (the fused MLP) may write
            # them straight into their views of the flat bucket and return no `.grad` for them
            self._flat_buffer()
            sink.small_direct = {q.data_ptr(): v for q, v in zip(self.small, self._flat_views)}
        if self.shard_optimizer:
            if not sink.groups:
                raise ValueError("shard_optimizer needs the pipelined exchange (pipeline_groups >= 1)")
            if encoder.shadow() is None:
                raise ValueError("shard_optimizer needs the bf16 table shadow (what the all-gather distributes)")
            self._shadow_of = encoder.shadow
            self._plan_shards(sink, p)
        encoder.grad_sink = sink
        return sink

    # ---- sharded table optimiser: row ranges
    def _plan_shards(self, sink, table):
        """Exchange ranges of the level groups for the row-sharded optimiser.  A rank's shard must start on a multiple of
        4 rows (16-byte accesses of the Adam kernel on the bf16 gradient), so group boundaries are rounded DOWN to
        multiples of 4 * world -- rows between a rounded boundary and the level boundary travel with the next group,
        whose sums exist by then -- and the last < 4 * world rows of the table stay replicated (all-reduce, stepped by
        every rank).  [(A, B)] per group with (B - A) % (4 * world) == 0, and the replicated remainder [B_last, rows)."""
        offs = self._sink_levels.offsets
        gran = 4 * self.world
        n_rows = table.shape[0]
        cuts = [0] + [(offs[hi] // gran) * gran for _lo, hi in sink.groups]
        cuts[-1] = (n_rows // gran) * gran
        sink.shard_ranges = [(cuts[i], cuts[i + 1]) for i in range(len(sink.groups))]
        sink.shard_rest = (cuts[-1], n_rows)

    def my_rows(self, a, b):
        n = (b - a) // self.world
        return a + self.rank * n, a + (self.rank + 1) * n

    def gather_rows(self, tensors):
        """Sharded optimiser: make `tensors` ([rows, 2] each: the f32 master table, its moments) whole on every rank
        from their owners' rows.  Collective; for checkpoints and end-of-run checks, never on the step's path."""
        if not (self.shard_optimizer and self.active):
            return
        sink = next(iter(self._sinks.values()))
        for t in tensors:
            for a, b in sink.shard_ranges:
                if b > a:
                    lo, hi = self.my_rows(a, b)
                    dist.all_gather_into_tensor(t[a:b], t[lo:hi].clone(), group=self.group)

    def reduced(self):
        """{parameter: tensor holding the reduced gradient} after allreduce(copy_back=False) / allreduce_pipelined():
        big buckets -- the bf16 wire buffers (FusedAdam.step(grads=...) reads them directly), else the `.grad` tensors
        themselves; small parameters -- their views of the flat bucket (no copy back into `.grad`)."""
        out = {p: self._wire.get(id(p), p.grad) if self.transport != torch.float32 and self.active else p.grad
               for p in self.big}
        if self.active and self.small and self._flat is not None:
            out.update({p: v for p, v in zip(self.small, self._flat_views)})
        return out

    def allreduce(self, copy_back=True):
        """After this call every `.grad` holds the SUM over ranks (scale by 1/world in the optimiser).  With the bf16
        transport and copy_back=False the sums of the big buckets stay in the wire buffers (see reduced())."""
        if not self.active:
            return
        pending = []
        for p in self.big:
            sink = self._sinks.get(id(p))
            if sink is not None:  # the backward pass wrote this rank's gradient into the wire buffer itself
                # (no host-side count of backward passes here: a replayed hipGraph writes the buffer without
                # running any Python)
                if copy_back:
                    raise RuntimeError("GradSync: a sinked gradient has no f32 `.grad` to copy back into")
                pending.append((dist.all_reduce(sink.wire, group=self.group, async_op=True), None, None))
                continue
            if p.grad is None:
                raise RuntimeError("GradSync: a bucketed parameter has no gradient on this rank")
            if self.transport == torch.float32:
                pending.append((dist.all_reduce(p.grad, group=self.group, async_op=True), None, None))
            else:
                wire = self._wire_buffer(p)
                wire.copy_(p.grad)  # f32 -> bf16 on the device
                pending.append((dist.all_reduce(wire, group=self.group, async_op=True), wire, p))
        if self.small:
            flat = self._pack_small()
            dist.all_reduce(flat, group=self.group)
            if copy_back:
                self._unpack_small()   # (copy_back=False: the sums stay in the flat bucket, see reduced())
        for h, wire, p in pending:
            h.wait()
            if wire is not None and copy_back:
                p.grad.copy_(wire)  # bf16 -> f32

    def allreduce_pipelined(self):
        """The exchange of a step whose table gradient sits in a pipelined GradSink (attach_sink(pipeline_groups=G)):
        the backward pass has only binned the scatter records; here one level group at a time is summed
        (lnerf_grid_scatter_reduce_bf16) and its all-reduce launched right behind it on the collective's own stream, so
        group g travels over xGMI while group g + 1 is still being summed; the small parameters follow as one flat async
        bucket.  Returns a PendingExchange: `table_groups` = [(row_lo, row_hi, work)] for FusedAdam.step(row_groups=...)
        -- the optimiser waits for a group right before it steps those rows -- and finish_small()."""
        from ..models.encoding import grid_scatter_reduce_group
        table = self.big[0]
        sink = self._sinks.get(id(table))
        if sink is None or not sink.groups:
            raise RuntimeError("allreduce_pipelined() needs attach_sink(encoder, pipeline_groups >= 1)")
        offs = self._sink_levels.offsets
        groups = []
        gathers = []
        for gi, (lo, hi) in enumerate(sink.groups):
            grid_scatter_reduce_group(sink, lo, hi)
            if not self.shard_optimizer:
                rows = sink.wire[offs[lo]:offs[hi]]
                work = dist.all_reduce(rows, group=self.group, async_op=True) if self.active else None
                groups.append((offs[lo], offs[hi], work))
                continue
            # row-sharded optimiser: this rank receives the SUM of its 1/N of the group's rows, steps them, and hands
            # their new bf16 shadow rows to everybody (the all-gather is launched by the optimiser right behind that
            # group's Adam launch: `after`)
            a, b = sink.shard_ranges[gi]
            if b > a:
                r0, r1 = self.my_rows(a, b)
                mine = sink.wire[r0:r1]     # in place: the reduce-scatter's output is this rank's slice of its input
                work = dist.reduce_scatter_tensor(mine, sink.wire[a:b], group=self.group, async_op=True) if self.active else None

                def after(a=a, b=b, r0=r0, r1=r1):
                    sh = self._shadow_of()
                    gathers.append(dist.all_gather_into_tensor(sh[a:b], sh[r0:r1], group=self.group, async_op=True))
                groups.append((r0, r1, work, after))
            if gi == len(sink.groups) - 1 and sink.shard_rest[1] > sink.shard_rest[0]:
                ra, rb = sink.shard_rest   # the table's last few rows: replicated (see _plan_shards)
                work = dist.all_reduce(sink.wire[ra:rb], group=self.group, async_op=True) if self.active else None
                groups.append((ra, rb, work))
        sink.written += 1     # (`pending` stays: a replayed hipGraph bins again without running any Python)
        small_work = None
        if self.small:
            flat = self._pack_small()
            if self.active:
                small_work = dist.all_reduce(flat, group=self.group, async_op=True)
        return PendingExchange(groups, self, small_work, gathers)


class PendingExchange:
    def __init__(self, table_groups, sync, small_work, gathers=None):
        self.table_groups = table_groups
        self._sync, self._work = sync, small_work
        self._gathers = gathers if gathers is not None else []

    def finish_gathers(self):
        """Sharded optimiser: orders the current stream behind the all-gathers of the shadow rows the optimiser step
        launched (call after optimizer.step(): the next gather reads the whole shadow)."""
        for w in self._gathers:
            w.wait()
        del self._gathers[:]

    def finish_small(self, copy_back=False):
        """Orders the current stream behind the flat bucket's all-reduce.  The sums stay in the bucket
        (GradSync.reduced() hands the optimiser views of it); copy_back=True also writes them into the `.grad`s."""
        if self._work is not None:
            self._work.wait()
        if copy_back and self._sync.small:
            self._sync._unpack_small()
        self._work = None
