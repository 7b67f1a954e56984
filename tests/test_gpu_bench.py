"""bench.py as the driver runs it, on one card: `--gpus N` without a launcher starts its own ranks (a free rendezvous
port, rank 0's JSON line relayed, worst exit code), the N > 1 step form is chosen by a supervised pre-flight, and a batch
of views per rank goes through the fused captured step.  RCCL itself needs one GPU per rank, so the two-rank run uses
LNERF_DIST_BACKEND=gloo (ranks share the card; the collectives stage through the host) and the RCCL path runs with ONE
rank (--force-dist)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None, timeout=900):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    out, err = p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")
    return p.returncode, out, err


def test_bench_starts_its_own_ranks(built_lib):
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the parent (which never touches the GPU) starts two ranks,
    relays ONE JSON line with n_gpus = 2, and the run's own replica check (rank checksums equal) has passed."""
    rc, out, err = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras"],
                          env={"LNERF_DIST_BACKEND": "gloo"})
    assert rc == 0, err[-3000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 20 and res["warmup"] == 5 and res["scaling"] == "weak"
    assert res["config"]["views_per_step"] == 2 and res["value"] > 0
    assert abs(res["value"] - 2 * 20 / (res["ms_per_step"] * 1e-3 * 20)) < 1e-6 * res["value"]
    assert res["preflight"]["ran"] is False and "gloo" in res["preflight"]["why"]     # nothing to capture on gloo
    assert "starting 2 ranks" in err and "exchange and optimiser eager" in res["launch"]


def test_bench_parent_reports_a_failing_rank(built_lib):
    """A rank that dies takes the job down: the parent kills the others (they would wait at a barrier for ever) and
    exits non-zero, within seconds, without a result line."""
    rc, out, err = _bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-extras",
                           "--tune", "no_such_key=1"], env={"LNERF_DIST_BACKEND": "gloo"}, timeout=600)
    assert rc != 0 and out.strip() == "", (rc, out)


def test_preflight_picks_the_captured_exchange_on_rccl(built_lib):
    """--force-dist: the N > 1 step on a one-rank RCCL communicator.  `--graph-collectives auto` runs the pre-flight
    (a supervised child: eager exchange vs. captured exchange from the same seeded state, bit for bit; replicas
    compared) BEFORE the timed process touches the GPU and captures the exchange on its verdict."""
    rc, out, err = _bench(["--force-dist", "--steps", "12", "--warmup", "4", "--no-cpu-baseline", "--no-extras"])
    assert rc == 0, err[-3000:]
    res = json.loads(out.strip().splitlines()[-1])
    pf = res["preflight"]
    assert pf["ran"] and pf["passed"] and pf["exit_codes"] == [0], pf
    assert "captured ==" in err and "exchange + optimiser captured" in res["launch"]
    assert res["n_gpus"] == 1 and "force-dist" in res["config"]["parallelism"]
    # the depth of the pipelined exchange was timed on the job's own ranks (1 / 2 / 4 / 8 level groups) and the fastest
    # taken: on ONE rank nothing travels, so fewer launches win
    et = res["exchange_tuning"]
    assert set(et["ms_per_step"]) == {"1", "2", "4", "8"} and et["chosen"] in (1, 2)
    assert et["ms_per_step"][str(et["chosen"])] == min(et["ms_per_step"].values())
    assert ("table in %d pipelined level groups" % et["chosen"]) in res["config"]["parallelism"]
    # and the operator's switch: no pre-flight, eager exchange
    rc, out, err = _bench(["--force-dist", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-extras"],
                          env={"LNERF_GRAPH_COLLECTIVES": "0"})
    assert rc == 0, err[-3000:]
    res = json.loads(out.strip().splitlines()[-1])
    assert res["preflight"]["ran"] is False and "eager" in res["launch"]


def test_bench_views_per_rank_goes_through_the_fused_step(built_lib):
    rc, out, err = _bench(["--views-per-rank", "4", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-extras"])
    assert rc == 0, err[-3000:]
    res = json.loads(out.strip().splitlines()[-1])
    assert res["config"]["views_per_rank"] == 4 and res["config"]["views_per_step"] == 4
    assert res["step_tail"] == "inside the scatter's pass 2" and res["launch"] == "hipgraph"
    assert res["roofline"]["samples_per_launch"] > 4 * 300000       # four views' samples in one gather launch
    assert abs(res["value"] - 4e3 / res["ms_per_step"]) < 1e-6 * res["value"]


def test_bench_under_torch_distributed_run(built_lib):
    """The driver's own form for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` -- the ranks arrive with WORLD_SIZE set (no self-launch), join the
    agent's store, and rank 0 prints the one line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LNERF_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10",
                        "--warmup", "3", "--no-cpu-baseline", "--no-extras"], env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    out, err = p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-3000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 10 and res["config"]["views_per_step"] == 2
    assert "starting 2 ranks" not in err


def test_bench_two_ranks_row_sharded_optimiser_and_two_views_each(built_lib):
    """`--shard-optimizer 1 --views-per-rank 2` on two gloo ranks: reduce-scatter of the bf16 gradient, every rank steps the
    rows it owns, all-gather of the shadow; the run's own end-of-run check (masters gathered, checksums of master, MLP and
    shadow equal on both ranks) has passed when the line appears."""
    rc, out, err = _bench(["--gpus", "2", "--steps", "12", "--warmup", "4", "--no-cpu-baseline", "--no-extras",
                           "--shard-optimizer", "1", "--views-per-rank", "2"], env={"LNERF_DIST_BACKEND": "gloo"})
    assert rc == 0, err[-3000:]
    res = json.loads(out.strip().splitlines()[-1])
    assert res["n_gpus"] == 2 and res["config"]["views_per_step"] == 4 and res["config"]["views_per_rank"] == 2
    assert "row-sharded table optimiser" in res["config"]["parallelism"]
    assert abs(res["value"] - 4e3 / res["ms_per_step"]) < 1e-6 * res["value"]
