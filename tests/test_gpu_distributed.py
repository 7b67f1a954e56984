"""Data-parallel trainer on the HIP path, rehearsed on ONE card: two ranks (gloo backend, sharing the GPU) run the real
Trainer for 20 steps -- two occupancy refreshes included (steps 1 and 17) -- exchanging gradients once per step, bf16 on
the wire, the table in pipelined level groups.  Replicas must stay bit-identical: table, MLP, density grid and bitfield.
(RCCL itself cannot run here: one GPU per box.  The collective calls are the same `torch.distributed` calls.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(tmp_path, world, groups, steps=20, precision="bf16", backend="gloo", force=False, save=False, tag="",
         graph_collectives=True, shard=False, views_per_rank=1):
    out = tmp_path / ("w%d_g%d_%s%s" % (world, groups, precision, tag))
    out.mkdir()
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LNERF_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0",
                   LNERF_FORCE_DIST="1" if force else "0",
                   LNERF_TEST_GRAPH_COLLECTIVES="1" if graph_collectives else "0",
                   LNERF_TEST_SHARD="1" if shard else "0", LNERF_TEST_VIEWS_PER_RANK=str(views_per_rank))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(out), str(groups),
                                       str(steps), precision] + (["save"] if save else []), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    res = [json.load(open(out / ("rank%d.json" % r))) for r in range(world)]
    for r in res:
        r["dir"] = str(out)
    return res


@pytest.mark.parametrize("groups", [4, 1])
def test_two_rank_replicas_stay_bit_identical(built_lib, tmp_path, groups):
    res = _run(tmp_path, 2, groups)
    a, b = res
    assert a["pipelined"] and b["pipelined"] and a["steps"] == b["steps"] == 20
    assert a["iter_density"] == b["iter_density"] == 2                     # refreshed at steps 1 and 17
    assert a["finite"] and a["table_moved"] > 0 and a["bits_set"] > 0
    assert a["noise_seed"] != b["noise_seed"]                              # per-rank march jitter
    for key in ("table", "mlp", "density_grid", "bitfield", "mean_density"):
        assert a[key] == b[key], key


def test_row_sharded_table_optimiser_two_ranks(built_lib, tmp_path):
    """optim.shard_table_optimizer: reduce-scatter of the bf16 table gradient, every rank runs Adam on the rows it owns,
    all-gather of the bf16 shadow the gather reads (the direct RS + AG shape).  Two ranks on one
    card (gloo), 20 steps with two occupancy refreshes: the SHADOWS (what renders) are bit-identical on both ranks, the
    f32 masters differ until gathered (each rank only steps its rows) and are bit-identical after the checkpoint's
    gather; the checkpoint file holds that table and its gathered moments.  Everything downstream of the shadow --
    density grid, bitfield -- is identical too.  (RCCL's reduce-scatter cannot run on a one-GPU box: unmeasured.)"""
    a, b = _run(tmp_path, 2, 4, shard=True)
    assert a["sharded"] and b["sharded"] and a["pipelined"] and a["steps"] == 20
    assert a["shadow"] == b["shadow"] and a["shadow"] is not None
    assert a["table_before_gather"] != b["table_before_gather"]      # each master current on its owner's rows only
    for key in ("table", "moments", "mlp", "density_grid", "bitfield"):
        assert a[key] == b[key], key
    assert a["ckpt_table"] == a["table"] and a["ckpt_m"] == a["moments"][0]
    assert a["finite"] and a["table_moved"] > 0 and a["bits_set"] > 0
    # the exchange also runs un-sharded in the same worker (default): same shadows on both ranks there as well
    c, d = _run(tmp_path, 2, 4, tag="_plain")
    assert not c["sharded"] and c["shadow"] == d["shadow"] and c["table"] == d["table"]


def test_two_ranks_two_views_each(built_lib, tmp_path):
    """optim.views_per_step = 4 on two ranks: every rank renders its two views as ONE batch (one backward into the wire
    buffer, one exchange, one optimiser step with grad_scale 1/4); the ranks' view sets partition the step's views and the
    replicas stay bit-identical."""
    a, b = _run(tmp_path, 2, 4, steps=12, views_per_rank=2, tag="_v2")
    assert a["views"] == [0, 2] and b["views"] == [1, 3] and a["pipelined"] and a["steps"] == 12
    for key in ("table", "mlp", "density_grid", "bitfield", "shadow"):
        assert a[key] == b[key], key
    assert a["finite"] and a["table_moved"] > 0
    assert a["graph_stats"]["replayed_steps"] >= 8


def test_f32_exchange_and_single_rank_paths(built_lib, tmp_path):
    """f32 transport (exact-f32 configuration): blocking all-reduce of the f32 `.grad`s; replicas identical too.  And
    one rank through the same worker (no process group): the fused single-GPU step."""
    a, b = _run(tmp_path, 2, 4, steps=6, precision="f32")
    assert not a["pipelined"]
    for key in ("table", "mlp", "density_grid", "bitfield"):
        assert a[key] == b[key], key
    (s,) = _run(tmp_path, 1, 4, steps=6)
    assert not s["pipelined"] and s["finite"] and s["table_moved"] > 0


def test_exchange_path_on_rccl_with_one_rank(built_lib, tmp_path):
    """The N > 1 step on RCCL itself: ONE rank joins a `backend="nccl"` process group (communicator of size 1,
    initialised before any GPU call) and LNERF_FORCE_DIST=1 takes the un-fused path -- bf16 gradient sink written by the
    scatter, table exchanged in 4 pipelined level groups (async all-reduce per group on RCCL's stream), flat f32 bucket
    of small parameters, FusedAdam.step(row_groups=...) -- through the real Trainer.  Twice: with the exchange and the
    optimiser CAPTURED into the step graph (optim.graph_collectives, the default on RCCL: one graph launch per step) and
    with graph / eager exchange / eager optimiser; the two run the same kernels in the same order and must agree bit
    for bit.  Against the fused single-rank trainer on the same seeds the table must agree to the bf16-wire tolerance:
    the only difference is the rounding of the summed table gradient to bf16 (2^-9 relative) before Adam."""
    import numpy as np
    (forced,) = _run(tmp_path, 1, 4, steps=20, backend="nccl", force=True, save=True, tag="_forced")
    (eager_x,) = _run(tmp_path, 1, 4, steps=20, backend="nccl", force=True, save=True, tag="_eagerx",
                      graph_collectives=False)
    (fused,) = _run(tmp_path, 1, 4, steps=20, backend="nccl", force=False, save=True, tag="_fused")
    assert forced["exchange"] and forced["pipelined"] and not fused["exchange"] and not fused["pipelined"]
    assert forced["capture_exchange"] and not eager_x["capture_exchange"] and eager_x["pipelined"]
    for key in ("table", "mlp", "density_grid", "bitfield"):
        assert forced[key] == eager_x[key], key            # captured exchange == eager exchange, bit for bit
    assert forced["steps"] == fused["steps"] == 20 and forced["finite"] and forced["table_moved"] > 0
    assert forced["graph_stats"]["replayed_steps"] >= 15 and fused["graph_stats"]["replayed_steps"] >= 15
    ld = lambda r, name: np.load(os.path.join(r["dir"], name))
    t0 = ld(fused, "table0_rank0.npy")
    # (1) after ONE step: Adam's first step is lr * g / (|g| + eps) = lr * sign(g) for every row with a gradient, and
    # rounding a gradient to bf16 keeps its sign and its zero: the two tables agree to f32 rounding of the step
    a1, b1 = ld(forced, "table1_rank0.npy"), ld(fused, "table1_rank0.npy")
    step = float(np.abs(b1 - t0).max())
    assert step > 0 and float(np.abs(a1 - b1).max()) <= 1e-4 * step, (float(np.abs(a1 - b1).max()), step)
    assert np.array_equal(a1 != t0, b1 != t0)          # the same rows moved
    # (2) after 20 steps (18 of them replays with the eager exchange in between).  Adam normalises every row's step by
    # the row's own gradient history, so a row whose gradient is noise takes full-size steps in a direction the 2^-9
    # perturbation of the wire format can flip: single rows may differ by the whole movement.  The bulk may not:
    a, b = ld(forced, "table_rank0.npy"), ld(fused, "table_rank0.npy")
    moved = np.abs(b - t0)
    sel = moved > 0.25 * float(moved.max())              # rows that moved consistently: their gradients are not noise
    assert int(sel.sum()) > 1000
    rel = np.abs(a - b)[sel] / moved[sel]
    assert float(np.median(rel)) <= 0.02 and float(np.mean(rel)) <= 0.1, (float(np.median(rel)), float(np.mean(rel)))
    w_rel = np.abs(ld(forced, "w2_rank0.npy") - ld(fused, "w2_rank0.npy")).mean() / np.abs(ld(fused, "w2_rank0.npy")).mean()
    assert w_rel <= 0.1, w_rel
