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


def _run(tmp_path, world, groups, steps=20, precision="bf16"):
    out = tmp_path / ("w%d_g%d_%s" % (world, groups, precision))
    out.mkdir()
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LNERF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(out), str(groups),
                                       str(steps), precision], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
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
    return [json.load(open(out / ("rank%d.json" % r))) for r in range(world)]


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


def test_f32_exchange_and_single_rank_paths(built_lib, tmp_path):
    """f32 transport (exact-f32 configuration): blocking all-reduce of the f32 `.grad`s; replicas identical too.  And
    one rank through the same worker (no process group): the fused single-GPU step."""
    a, b = _run(tmp_path, 2, 4, steps=6, precision="f32")
    assert not a["pipelined"]
    for key in ("table", "mlp", "density_grid", "bitfield"):
        assert a[key] == b[key], key
    (s,) = _run(tmp_path, 1, 4, steps=6)
    assert not s["pipelined"] and s["finite"] and s["table_moved"] > 0
