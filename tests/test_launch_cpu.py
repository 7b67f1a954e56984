"""The rank launcher and the pre-group control plane (src/latent_nerf/training/launch.py) on the CPU: children get a
complete rendezvous environment, rank 0's stdout comes back, the worst exit code wins, a failing or hanging rank takes
the job down instead of leaving the others at a barrier, and `agree()` hands every rank the same list of votes."""
import os
import sys
import time

from src.latent_nerf.training import distributed as D
from src.latent_nerf.training import launch as L


_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py(code):
    return [sys.executable, "-c", code]


def test_spawn_ranks_environment_and_relay():
    code = ("import os, sys; r = os.environ['RANK']; "
            "assert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == r and os.environ['MASTER_ADDR'] == '127.0.0.1'; "
            "assert int(os.environ['MASTER_PORT']) > 1024 and 'TORCHELASTIC_USE_AGENT_STORE' not in os.environ; "
            "print('{\"rank\": %s}' % r)")
    rc, out = L.spawn_ranks(_py(code), 3, timeout_s=60)
    assert rc == 0 and out.strip() == '{"rank": 0}'          # only rank 0's stdout is relayed


def test_spawn_ranks_worst_exit_code_and_group_kill():
    # rank 1 fails at once; ranks 0 and 2 would sleep for a minute: the launcher stops them after its 5 s grace
    code = "import os, sys, time; sys.exit(7) if os.environ['RANK'] == '1' else time.sleep(60)"
    t0 = time.monotonic()
    rc, out = L.spawn_ranks(_py(code), 3, timeout_s=120)
    assert time.monotonic() - t0 < 30
    assert rc >= 7 and out == ""                             # (killed ranks report 128 + signal, the failed one 7)


def test_spawn_ranks_time_limit():
    t0 = time.monotonic()
    rc, _ = L.spawn_ranks(_py("import time; time.sleep(60)"), 2, timeout_s=2)
    assert rc >= 124 and time.monotonic() - t0 < 30


def test_agree_gives_every_rank_the_same_votes():
    # two ranks: store from the environment (rank 0 hosts it), each publishes a vote, both read [vote0, vote1]
    code = ("import os, sys; sys.path[:0] = %r; "
            "from src.latent_nerf.training import launch as L; "
            "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']); "
            "st = L.open_store(r, w, timeout_s=60); "
            "votes = L.agree(st, 'vote', r, w, 10 * (r + 1), timeout_s=60); "
            "assert votes == ['10', '20'], votes; "
            "L.agree(st, 'done', r, w, 1, timeout_s=60); "     # (rank 0 hosts the store: leave together)
            "print('ok')") % ([_ROOT, os.path.join(_ROOT, "latent-nerf-test_amd")],)
    rc, out = L.spawn_ranks(_py(code), 2, timeout_s=120)
    assert rc == 0 and out.strip() == "ok"


def test_run_child_kills_at_the_limit():
    t0 = time.monotonic()
    assert L.run_child(_py("import time; time.sleep(60)"), dict(os.environ), 1.0) == 124
    assert time.monotonic() - t0 < 20
    assert L.run_child(_py("import sys; sys.exit(3)"), dict(os.environ), 30.0) == 3
    assert L.run_child(_py("import time; time.sleep(60)"), dict(os.environ), 30.0, poll=lambda: True) == 124


def test_pose_uniforms_are_a_function_of_seed_step_view():
    a, b = D.pose_uniforms(7, 12, 5), D.pose_uniforms(7, 12, 5)
    assert a == b and len(a) == 4 and all(0.0 <= u < 1.0 for u in a)
    assert all(float(u) == float.fromhex(float(u).hex()) and u * 16777216 == int(u * 16777216) for u in a)   # 24 bits
    assert D.pose_uniforms(7, 12, 6) != a and D.pose_uniforms(7, 13, 5) != a and D.pose_uniforms(8, 12, 5) != a
    # roughly uniform: mean of 4000 draws
    us = [u for s in range(1000) for u in D.pose_uniforms(0, s, 0)]
    assert abs(sum(us) / len(us) - 0.5) < 0.02
    # a run with W ranks renders the same set of views as a 1-rank run
    one = [D.pose_uniforms(3, 9, v) for v in D.views_for_rank(8, 0, 1)]
    many = {v: D.pose_uniforms(3, 9, v) for r in range(4) for v in D.views_for_rank(8, r, 4)}
    assert [many[v] for v in range(8)] == one


def test_store_and_agree_under_torchrun_agent_store(tmp_path):
    """The driver starts N > 1 ranks with `python -m torch.distributed.run`: its agent hosts the c10d store and the ranks
    are clients (TORCHELASTIC_USE_AGENT_STORE).  open_store() must join THAT store, the pre-group exchange must work over
    it, and init_process_group(store=...) must build the group on it -- no second rendezvous, no fixed port."""
    import subprocess
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys\n"
        "sys.path[:0] = %r\n"
        "import torch, torch.distributed as dist\n"
        "from src.latent_nerf.training import launch as L\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ.get('TORCHELASTIC_USE_AGENT_STORE') == 'True'\n"
        "st = L.open_store(r, w, timeout_s=60)\n"
        "if r == 0: st.set('port', str(L.free_port()))\n"
        "port = int(st.get('port').decode())\n"
        "votes = L.agree(st, 'v', r, w, 7 * r, timeout_s=60)\n"
        "assert votes == ['0', '7'], votes\n"
        "dist.init_process_group('gloo', store=st, rank=r, world_size=w)\n"
        "t = torch.tensor([float(r + 1)]); dist.all_reduce(t)\n"
        "assert float(t) == 3.0 and port > 1024\n"
        "dist.barrier(); dist.destroy_process_group()\n"
        "open(os.path.join(%r, 'ok%%d' %% r), 'w').write('ok')\n" % ([_ROOT, os.path.join(_ROOT, "latent-nerf-test_amd")],
                                                                    str(tmp_path)))
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(L.free_port()), str(script)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=180)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-2000:]
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
