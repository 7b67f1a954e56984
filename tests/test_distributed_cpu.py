"""World-size-2 gloo tests of the data-parallel host logic (view sharding, flat-bucket gradient
sum, replicas staying identical after the same optimiser step)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from src.latent_nerf.training import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, transport="f32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)  # identical replicas
        table = torch.nn.Parameter(torch.randn(4096, 2))
        small = [torch.nn.Parameter(torch.randn(64, 32)), torch.nn.Parameter(torch.randn(64)),
                 torch.nn.Parameter(torch.randn(5, 64))]
        g = torch.Generator().manual_seed(100 + rank)  # rank-specific gradients (its own view)
        table.grad = torch.randn(4096, 2, generator=g)
        for p in small:
            p.grad = torch.randn(p.shape, generator=g)
        local = [table.grad.clone()] + [p.grad.clone() for p in small]
        sync = D.GradSync([table], small, transport=torch.bfloat16 if transport == "bf16" else torch.float32)
        # first without the copy back: the bf16 sums stay in the wire buffer (what FusedAdam.step(grads=...) reads),
        # the f32 transport reduces in place either way
        keep = table.grad.clone()
        sync.allreduce(copy_back=False)
        red = sync.reduced()[table]
        if transport == "bf16":
            assert red.dtype == torch.bfloat16 and torch.equal(table.grad, keep)   # .grad untouched
            wire_sum = red.float().clone()
        else:
            assert red is table.grad
        table.grad = keep.clone()
        for p, l in zip(small, local[1:]):
            p.grad = l.clone()
        sync.allreduce()
        if transport == "bf16":
            assert torch.equal(table.grad, wire_sum)   # the copy back delivers exactly the wire buffer
        # gather every rank's local gradients to check the sum on rank 0
        gathered = [None] * world
        dist.all_gather_object(gathered, [t.numpy() for t in local])
        if rank == 0:
            for i, p in enumerate([table] + small):
                ref = sum(torch.from_numpy(gathered[r][i]) for r in range(world))
                if transport == "bf16" and i == 0:   # the big bucket travelled (and was summed) in bf16
                    assert torch.allclose(p.grad, ref, rtol=2e-2, atol=2e-2)
                    assert torch.equal(p.grad, p.grad.to(torch.bfloat16).float())
                else:
                    assert torch.allclose(p.grad, ref, atol=1e-6)
        if transport == "bf16":
            # gradient sink: the backward pass writes this rank's table gradient straight into the wire buffer
            # (GridEncoder.grad_sink / lnerf_grid_encode_backward_bf16); emulated here by filling the buffer by hand
            class _Enc:
                pass
            enc = _Enc()
            enc.embeddings = table
            enc.grad_sink = None
            sync2 = D.GradSync([table], small, transport=torch.bfloat16)
            sink = sync2.attach_sink(enc)
            assert enc.grad_sink is sink and sink.wire.dtype == torch.bfloat16 and float(sink.zero.abs().max()) == 0.0
            sink.wire.copy_(local[0])
            for p, l in zip(small, local[1:]):
                p.grad = l.clone()
            keep_grad = table.grad
            table.grad = None                      # a sinked table has no .grad
            sync2.allreduce(copy_back=False)
            assert sync2.reduced()[table] is sink.wire
            # copy_back=False leaves the small `.grad`s alone: their sums sit in the flat bucket, reduced() hands out views
            for p, l in zip(small, local[1:]):
                assert torch.equal(p.grad, l)
                red_p = sync2.reduced()[p]
                assert red_p.shape == p.shape and red_p.data_ptr() != p.grad.data_ptr()
                p.grad.copy_(red_p)                # (what FusedAdam.step(grads=reduced()) reads directly)
            got = [torch.empty_like(sink.wire) for _ in range(world)]
            dist.all_gather(got, sink.wire)
            assert all(torch.equal(got[0], w) for w in got)                      # same sums on every rank
            ref16 = sum(torch.from_numpy(gathered[r][0]).to(torch.bfloat16).float() for r in range(world))
            assert torch.allclose(sink.wire.float(), ref16, rtol=2e-2, atol=2e-2)
            try:
                sync2.allreduce(copy_back=True)
                raise AssertionError("copy_back into a sinked gradient must be refused")
            except RuntimeError:
                pass
            table.grad = keep_grad
        # identical optimiser step on every rank -> replicas stay bit-identical
        opt = torch.optim.Adam([table] + small, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
        for p in [table] + small:
            p.grad.mul_(1.0 / world)
        opt.step()
        flat = torch.cat([p.detach().reshape(-1) for p in [table] + small])
        all_flat = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(all_flat, flat)
        assert all(torch.equal(all_flat[0], f) for f in all_flat)
        out[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["f32", "bf16"])
def test_gradsync_two_ranks_gloo(transport):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out, transport), nprocs=world, join=True)
    assert dict(out) == {0: "ok", 1: "ok"}


def test_view_sharding_and_pose_rng():
    assert D.views_for_rank(8, 3, 8) == [3]
    assert D.views_for_rank(8, 1, 2) == [1, 3, 5, 7]
    all_views = sorted(v for r in range(4) for v in D.views_for_rank(8, r, 4))
    assert all_views == list(range(8))  # a partition: no view rendered twice, none dropped
    with pytest.raises(ValueError):
        D.views_for_rank(6, 0, 4)
    a = torch.rand(3, generator=D.pose_generator(7, 12, 5))
    b = torch.rand(3, generator=D.pose_generator(7, 12, 5))
    c = torch.rand(3, generator=D.pose_generator(7, 12, 6))
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_gradsync_single_process_is_noop():
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    D.GradSync([p], []).allreduce()
    assert torch.equal(p.grad, torch.ones(4))


def _shard_worker(rank, world, port, out):
    """Row-sharded table optimiser (GradSync(shard_optimizer=True)) with the HIP pieces replaced by plain tensor code:
    the per-group reduce launch fills the wire rows with this rank's gradient, the 'optimiser' steps the rows it is
    handed from the reduced gradient and writes the shadow.  Every rank must end with the same, complete shadow --
    the one a replicated optimiser would have produced -- and gather_rows() must make the master whole."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from src.latent_nerf.models import encoding as E

        class _Levels:
            num_levels = 6
            offsets = [0, 40, 112, 312, 824, 1336, 1848 + 8]     # level sizes: multiples of 8 rows, as GridLevels makes them
            n_rows = offsets[-1]

        n_rows = _Levels.n_rows
        torch.manual_seed(0)
        table = torch.nn.Parameter(torch.randn(n_rows, 2))
        master0 = table.detach().clone()
        small = [torch.nn.Parameter(torch.randn(8, 4))]
        shadow = table.detach().to(torch.bfloat16).clone()

        class _Enc:
            levels = _Levels
            embeddings = table
            grad_sink = None

            @staticmethod
            def shadow():
                return shadow

        g = torch.Generator().manual_seed(100 + rank)
        my_grad = torch.randn(n_rows, 2, generator=g).to(torch.bfloat16)

        def fake_reduce(sink, lo, hi):   # pass 2 of levels [lo, hi): this rank's gradient rows into the wire buffer
            a, b = _Levels.offsets[lo], _Levels.offsets[hi]
            sink.wire[a:b] = my_grad[a:b]

        real = E.grid_scatter_reduce_group
        E.grid_scatter_reduce_group = fake_reduce
        try:
            sync = D.GradSync([table], small, transport=torch.bfloat16, shard_optimizer=True)
            sink = sync.attach_sink(_Enc, pipeline_groups=3)
            gran = 4 * world
            assert all((b - a) % gran == 0 and a % gran == 0 for a, b in sink.shard_ranges)
            assert sink.shard_ranges[0][0] == 0 and sink.shard_rest[1] == n_rows and sink.shard_rest[1] - sink.shard_rest[0] < gran
            assert [b for _a, b in sink.shard_ranges[:-1]] == [a for a, _b in sink.shard_ranges[1:]]   # contiguous
            sink.pending = True
            small[0].grad = torch.full((8, 4), float(rank + 1))
            ex = sync.allreduce_pipelined()
            ex.finish_small()
            stepped = torch.zeros(n_rows, dtype=torch.bool)
            for r0, r1, work, *rest in ex.table_groups:          # what FusedAdam.step(row_groups=...) does
                if work is not None:
                    work.wait()
                assert r0 % 4 == 0                                 # 16-byte accesses on the bf16 gradient
                table.data[r0:r1] -= 0.5 * sink.wire[r0:r1].float()
                shadow[r0:r1] = table.data[r0:r1].to(torch.bfloat16)
                stepped[r0:r1] = True
                if rest and rest[0] is not None:
                    rest[0]()
            ex.finish_gathers()
            assert float(sync.reduced()[small[0]].mean()) == sum(range(1, world + 1))
        finally:
            E.grid_scatter_reduce_group = real
        # reference: the replicated optimiser on the bf16 sum of every rank's gradient
        grads = [torch.empty_like(my_grad) for _ in range(world)]
        dist.all_gather(grads, my_grad)
        total = grads[0].clone()
        for x in grads[1:]:
            total = (total.float() + x.float()).to(torch.bfloat16)   # (pairwise bf16 sums: exact order for 2 ranks)
        want_master = master0 - 0.5 * total.float()
        owners = int(stepped.sum())
        assert n_rows // world <= owners <= n_rows // world + 4 * world      # a 1/N share (+ the replicated remainder)
        if world == 2:
            assert torch.equal(shadow, want_master.to(torch.bfloat16))         # complete and identical to the replicated step
        shadows = [torch.empty_like(shadow) for _ in range(world)]
        dist.all_gather(shadows, shadow)
        assert all(torch.equal(shadows[0], s) for s in shadows)
        assert not torch.equal(table.data, want_master) or world == 1         # the master is only current on its owner ...
        sync.gather_rows([table.data])
        masters = [torch.empty_like(table.data) for _ in range(world)]
        dist.all_gather(masters, table.data)
        assert all(torch.equal(masters[0], m) for m in masters)                # ... until gather_rows()
        if world == 2:
            assert torch.equal(table.data, want_master)
        assert torch.equal(table.data.to(torch.bfloat16), shadow)
        out[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_table_optimiser_gloo(world):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_shard_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {r: "ok" for r in range(world)}


def test_shard_plan_on_the_bench_table_for_eight_ranks():
    """Row-shard boundaries of the default table (L = 16, T = 2^19: 6 119 864 rows) for 8 ranks and 4 level groups: every
    exchange range starts and ends on a multiple of 4 x 8 rows (a rank's shard starts on a multiple of 4 rows: 16-byte
    accesses on the bf16 gradient), ranges are contiguous from row 0, a range never reaches past the rows its level group
    (and the groups before it) have summed, and the replicated remainder is shorter than one granule."""
    from oracle import nerf_oracle as O

    class _Sink:
        pass

    lv = O.make_grid_levels()
    assert lv.n_rows == 6119864
    for world in (2, 4, 8, 3):
        sync = D.GradSync([torch.nn.Parameter(torch.zeros(8, 2))], [], transport=torch.bfloat16, shard_optimizer=True)
        sync.world, sync.rank = world, world - 1
        sync._sink_levels = lv
        sink = _Sink()
        # level groups of roughly equal rows, as encoding.level_groups makes them
        target, groups, lo = lv.n_rows / 4, [], 0
        for l in range(16):
            if lv.offsets[l + 1] >= target * (len(groups) + 1) - 1e-9 or l == 15:
                if len(groups) < 3 or l == 15:
                    groups.append((lo, l + 1))
                    lo = l + 1
        sink.groups = groups
        sync._plan_shards(sink, torch.empty(lv.n_rows, 0))
        gran = 4 * world
        prev = 0
        for (a, b), (_glo, ghi) in zip(sink.shard_ranges, groups):
            assert a == prev and a % gran == 0 and b % gran == 0 and b <= lv.offsets[ghi]
            r0, r1 = sync.my_rows(a, b)
            assert r0 % 4 == 0 and (r1 - r0) * world == b - a and a <= r0 < r1 <= b
            prev = b
        ra, rb = sink.shard_rest
        assert ra == prev and rb == lv.n_rows and 0 <= rb - ra < gran
