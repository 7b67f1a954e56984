"""GPU box probe: can torch.distributed's RCCL collectives be captured into a hipGraph (one rank)?
python3 tools/rccl_capture_probe.py  ->  prints one line per case."""
import os
import sys
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
so, se = os.dup(1), None
os.dup2(2, 1)   # RCCL's banner goes to stderr
dist.init_process_group("nccl", device_id=dev)
x = torch.ones(1 << 20, device=dev, dtype=torch.bfloat16)
dist.all_reduce(x)
torch.cuda.synchronize()
os.dup2(so, 1)
print("eager all_reduce ok", float(x.float().sum()), flush=True)

s = torch.cuda.Stream()
a = torch.full((12 << 20,), 1.0, device=dev, dtype=torch.bfloat16)
b = torch.zeros(30000, device=dev)
y = torch.zeros(4, device=dev)
for mode in ("thread_local", "global"):
    for async_op in (False, True):
        try:
            g = torch.cuda.CUDAGraph()
            s.wait_stream(torch.cuda.current_stream())
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s, capture_error_mode=mode):
                a.mul_(2.0)
                works = []
                for k in range(4):
                    sl = a[k * (3 << 20):(k + 1) * (3 << 20)]
                    w = dist.all_reduce(sl, async_op=async_op)
                    works.append(w)
                wb = dist.all_reduce(b, async_op=async_op)
                if async_op:
                    wb.wait()
                    for w in works:
                        w.wait()
                y += a[:4].float()
            a.fill_(1.0)
            y.zero_()
            torch.cuda.synchronize()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 50
            print("capture mode=%s async=%s ok: a[0]=%g y[0]=%g  replay %.1f us" % (mode, async_op, float(a[0]), float(y[0]), dt * 1e6), flush=True)
            del g
        except Exception as e:   # noqa
            print("capture mode=%s async=%s FAILED: %s" % (mode, async_op, str(e).splitlines()[0][:200]), flush=True)
            torch.cuda.synchronize()
dist.destroy_process_group()
print("done", flush=True)
