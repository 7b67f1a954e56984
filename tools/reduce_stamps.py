#!/usr/bin/env python3
"""Phase breakdown of the fused reduce + Adam pass (k_scatter_reduce, directly finished buckets) from in-kernel
shader-clock stamps: diagnostic build only (bash tools/run_bin_stamps.sh, -DLNERF_STAMPS).

    LNERF_HIP_LIB=latent-nerf-test_amd/lib/liblnerf_hip_stamps.so python3 tools/reduce_stamps.py
Cycles are those of wave 0 of every workgroup; every mark drains the wave's memory counters first."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "latent-nerf-test_amd")):
    sys.path.insert(0, _p)
import torch
import bench
from src.latent_nerf.raymarching import backend as B
from src.latent_nerf.training.optimizer import FusedAdam
dev = torch.device("cuda:0")
net, pose, intr, bg, grad = bench.build(dev, "bf16", 0, 0, "bf16")
opt = FusedAdam(net.get_params(1e-7), betas=(0.9, 0.99), eps=1e-15, encoder=net.encoder, capturable=True, fuse_table_update=True)
def step():
    out = net.render(None, None, camera=(pose, intr, bench.H, bench.W), bg_color=bg, perturb=True)
    opt.arm(); out["image"].backward(gradient=grad); opt.step()
lib = B.get_lib()
lib.lnerf_debug_bin_stamps.argtypes = [ctypes.c_void_p]; lib.lnerf_debug_bin_stamps.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3): step()
torch.cuda.synchronize(); lib.lnerf_debug_bin_stamps(buf)
n = 10
for _ in range(n): step()
torch.cuda.synchronize(); lib.lnerf_debug_bin_stamps(buf)
wgs = buf[15] / n
print(json.dumps({"fused_direct_workgroups_per_launch": wgs,
                  "cycles_per_workgroup": {"10 zero LDS + barrier": buf[10] / n / wgs, "11 record loop (loads + LDS atomics)": buf[11] / n / wgs,
                                           "12 barrier": buf[12] / n / wgs, "13 Adam: loads, math, stores (drained)": buf[13] / n / wgs}}, indent=1))
