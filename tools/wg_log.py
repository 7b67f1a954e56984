#!/usr/bin/env python3
"""Per-workgroup wall-clock log of the scatter's pass 2 (diagnostic build, -DLNERF_STAMPS: bash tools/run_bin_stamps.sh):
lifetime of every workgroup, idle time of a CU slot between two workgroups, by level.

    LNERF_HIP_LIB=latent-nerf-test_amd/lib/liblnerf_hip_stamps.so python3 tools/wg_log.py"""
import ctypes, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "latent-nerf-test_amd")):
    sys.path.insert(0, _p)
import torch
import bench
from src.latent_nerf.raymarching import backend as B
from src.latent_nerf.training.optimizer import FusedAdam
dev = torch.device("cuda:0")
net, pose, intr, bg, grad = bench.build(dev, "bf16", 0, 0, "bf16", gridtype=os.environ.get("LNERF_GRIDTYPE", "blocked"))
opt = FusedAdam(net.get_params(1e-7), betas=(0.9, 0.99), eps=1e-15, encoder=net.encoder, capturable=True,
                fuse_table_update=True, mlp=net)
def step():
    out = net.render(None, None, camera=(pose, intr, bench.H, bench.W), bg_color=bg, perturb=True)
    opt.arm(); out["image"].backward(gradient=grad); opt.step()
lib = B.get_lib()
lib.lnerf_debug_wg_log.argtypes = [ctypes.c_void_p, ctypes.c_int]; lib.lnerf_debug_wg_log.restype = ctypes.c_int
N = 4096
buf = (ctypes.c_ulonglong * (4 * N))()
for _ in range(5): step()
torch.cuda.synchronize()
lib.lnerf_debug_wg_log(buf, 4 * N)      # (clears the log: the exit times are running maxima)
step()
torch.cuda.synchronize()
lib.lnerf_debug_wg_log(buf, 4 * N)
rows = []
for i in range(N):
    t0, t1, where, unit = buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]
    if t1 > t0 > 0:
        hw, xcc = where & 0xFFFFFFFF, where >> 32
        cu = ((xcc & 0xF), (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xF)   # xcc, se, sh, cu
        rows.append((t0, t1, cu, int(unit) if unit < 2**31 else int(unit) - 2**32, i))
t_min = min(r[0] for r in rows); t_max = max(r[1] for r in rows)
print("workgroups %d, launch span %.1f us (100 MHz clock)" % (len(rows), (t_max - t_min) / 100.0))
life = sorted((r[1] - r[0]) / 100.0 for r in rows)
q = lambda v, p: v[int(p * (len(v) - 1))]
print("lifetime us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f  mean %.1f" % (life[0], q(life, .1), q(life, .5), q(life, .9), life[-1], sum(life) / len(life)))
by_cu = defaultdict(list)
for r in rows: by_cu[r[2]].append(r)
print("CUs seen %d; workgroups per CU: min %d max %d" % (len(by_cu), min(len(v) for v in by_cu.values()), max(len(v) for v in by_cu.values())))
# per CU: busy = union of lifetimes weighted by concurrency; first entry, last exit
first = sorted((min(r[0] for r in v) - t_min) / 100.0 for v in by_cu.values())
last = sorted((max(r[1] for r in v) - t_min) / 100.0 for v in by_cu.values())
print("first entry on a CU us: median %.1f max %.1f; last exit on a CU us: min %.1f median %.1f max %.1f" % (q(first, .5), first[-1], last[0], q(last, .5), last[-1]))
# concurrency integral per CU: time with 0 / 1 / 2 resident workgroups inside [t_min, t_max]
tot = [0.0, 0.0, 0.0, 0.0]
for v in by_cu.values():
    ev = sorted([(r[0], 1) for r in v] + [(r[1], -1) for r in v])
    cur, prev = 0, t_min
    for t, d in ev:
        tot[min(cur, 3)] += (t - prev) / 100.0; prev = t; cur += d
    tot[0] += (t_max - prev) / 100.0
n = len(by_cu)
print("per CU, mean us with 0 / 1 / 2 / 3+ workgroups resident: %.1f / %.1f / %.1f / %.1f" % tuple(x / n for x in tot))
# lifetime by position in the grid (entry order deciles)
rows.sort(key=lambda r: r[4])
for a in range(0, len(rows), max(1, len(rows) // 12)):
    seg = rows[a:a + max(1, len(rows) // 12)]
    print("  blocks %4d..%4d  units %5d..%5d  entry %.1f..%.1f us  mean life %.1f us" % (seg[0][4], seg[-1][4], seg[0][3], seg[-1][3],
          (min(r[0] for r in seg) - t_min) / 100.0, (max(r[0] for r in seg) - t_min) / 100.0, sum(r[1] - r[0] for r in seg) / len(seg) / 100.0))
# slot turnaround: on each CU, for every workgroup entry after the first two, the time since the latest exit before it
gaps = []
for v in by_cu.values():
    v = sorted(v)
    exits = sorted(r[1] for r in v)
    for k, r in enumerate(v):
        if k < 2: continue
        prev_exits = [e for e in exits if e <= r[0]]
        if prev_exits: gaps.append((r[0] - prev_exits[-1]) / 100.0)
gaps.sort()
print("turnaround (latest wave-0 exit on the CU -> next workgroup's entry) us: p10 %.2f median %.2f p90 %.2f mean %.2f  (n = %d)" % (q(gaps, .1), q(gaps, .5), q(gaps, .9), sum(gaps) / len(gaps), len(gaps)))
# units by lifetime: the longest ones
rows.sort(key=lambda r: r[0] - r[1])
print("longest workgroups (unit, life us, entry us):", [(r[3], round((r[1] - r[0]) / 100.0, 1), round((r[0] - t_min) / 100.0, 1)) for r in rows[:8]])
