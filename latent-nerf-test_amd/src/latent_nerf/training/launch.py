"""Starting and supervising the ranks of a one-node data-parallel job, and the rendezvous pieces around it.

The reference is single-process (every launch line pins one GPU: run_test.sh:10, run_latent_paint.txt:2-14; SURVEY.md
section 2.1), so nothing here mirrors reference code.  One process per GPU, started as plain children of a launcher that
has NOT touched the GPU (on this pool a process that initialised HIP must not exec, and need not: the launcher only
counts devices), each in its own session so that the whole rank can be killed by process group.

    spawn_ranks(argv, n)   N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (a free port),
                           rank 0's stdout relayed, worst exit code returned, every rank killed when one fails or the
                           time limit passes (a rank that died leaves the others at a barrier for ever otherwise)
    open_store()           the c10d store of the job (the launcher's, torchrun's agent store, or one hosted by rank 0):
                           used for small control-plane exchanges BEFORE the process group exists -- and handed to
                           init_process_group, so no second rendezvous happens
    agree(store, ...)      every rank publishes a small value, every rank reads all of them: a collective decision
                           without a collective (the captured-exchange pre-flight of bench.py / scripts)
"""
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def _kill_group(p, sig):
    try:
        os.killpg(p.pid, sig)      # (the child was started with start_new_session: its pid is its process group)
    except (ProcessLookupError, PermissionError):
        pass


def spawn_ranks(argv, n, timeout_s=1200.0, env_extra=None, log=None):
    """Runs `argv` as n ranks on this node.  Returns (worst exit code, rank 0's stdout as str).
    124 = the time limit passed (every rank killed); a rank that fails takes the others down after a 5 s grace."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)      # rank 0 hosts the store itself
        env.update(env_extra or {})
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      stderr=sys.stderr, start_new_session=True))
    t0 = time.monotonic()
    rc, failed_at = 0, None
    out0 = []
    import threading

    def _drain():    # rank 0's stdout must be read while it runs: a full pipe would block it
        for line in procs[0].stdout:
            out0.append(line.decode(errors="replace"))

    th = threading.Thread(target=_drain, daemon=True)
    th.start()
    try:
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            now = time.monotonic()
            bad = [c for c in codes if c not in (None, 0)]
            if bad and failed_at is None:
                failed_at = now
                if log:
                    log("a rank exited with code %s: stopping the others" % bad[0])
            if failed_at is not None and now - failed_at > 5.0:
                break
            if now - t0 > timeout_s:
                rc = 124
                if log:
                    log("time limit of %.0f s passed: killing every rank" % timeout_s)
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                _kill_group(p, signal.SIGTERM)
        t1 = time.monotonic()
        while any(p.poll() is None for p in procs) and time.monotonic() - t1 < 5.0:
            time.sleep(0.05)
        for p in procs:
            if p.poll() is None:
                _kill_group(p, signal.SIGKILL)
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
    th.join(timeout=5)
    for p in procs:
        c = p.returncode
        if c is None:
            c = 125
        if c < 0:                 # killed by a signal: report it the shell's way
            c = 128 - c
        rc = max(rc, c)
    return rc, "".join(out0)


def open_store(rank, world, timeout_s=600):
    """The job's c10d key-value store from the environment (MASTER_ADDR / MASTER_PORT; torchrun's agent store when it
    runs one).  Same rendezvous `init_process_group("env://")` would do -- pass the result as `store=` to it."""
    from datetime import timedelta
    from torch.distributed import rendezvous
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    store, _r, _w = next(rendezvous("env://", rank=rank, world_size=world, timeout=timedelta(seconds=timeout_s)))
    return store


def agree(store, key, rank, world, value, timeout_s=600):
    """Every rank publishes `value` (str) under `key`; returns the list of all ranks' values, the same list on every
    rank: decisions derived from it are taken by all ranks alike."""
    from datetime import timedelta
    store.set("%s/%d" % (key, rank), str(value))
    keys = ["%s/%d" % (key, r) for r in range(world)]
    store.wait(keys, timedelta(seconds=timeout_s))
    return [store.get(k).decode() for k in keys]


def run_child(argv, env, timeout_s, poll=None):
    """One child process (same session: it dies with this rank's process group), waited for with a time limit.
    Returns its exit code; 124 after a kill at the limit.  poll(): optional callable, True = give up now."""
    p = subprocess.Popen(argv, env=env, stdout=sys.stderr, stderr=sys.stderr)
    t0 = time.monotonic()
    while p.poll() is None:
        if time.monotonic() - t0 > timeout_s or (poll is not None and poll()):
            p.kill()
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
            return 124
        time.sleep(0.05)
    return p.returncode
