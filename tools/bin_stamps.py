#!/usr/bin/env python3
"""Phase breakdown of the binning pass from in-kernel shader-clock stamps (diagnostic build, see tools/run_bin_stamps.sh).

Build first (no GPU needed):   bash tools/run_bin_stamps.sh build
Run on the GPU box:            LNERF_HIP_LIB=latent-nerf-test_amd/lib/liblnerf_hip_stamps.so python3 tools/bin_stamps.py
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "latent-nerf-test_amd")):
    sys.path.insert(0, _p)
import torch  # noqa: E402

PHASES = ["0 loop top", "1 A: cell, rows, runs, emit mask", "2 B: ranking (LDS counters)", "3 barrier 1",
          "4 C+D: reservations + prefetch issued and waited for, values, tile max", "5 E+F: count scan, staging",
          "6 destinations (reservations waited for)", "7 barrier 3", "8 G: copy-out, stores completed (direct levels: stores)"]


def main():
    import bench
    from src.latent_nerf.models import encoding as E
    from src.latent_nerf.raymarching import backend as B
    from src.latent_nerf.raymarching import raymarching as rm
    dev = torch.device("cuda:0")
    net, pose, intr, bg, grad = bench.build(dev, "bf16", 0, 0, "bf16")
    rays_o, rays_d = rm.get_rays(pose, intr, bench.H, bench.W)
    store = {}
    orig = E.grid_encode_backward

    def grab(xyzs_, bound_, dfeat_, *a, **k):
        store["dfeat"] = dfeat_.clone()
        store["xyzs"] = xyzs_
        return orig(xyzs_, bound_, dfeat_, *a, **k)
    E.grid_encode_backward = grab
    out = net.render(rays_o, rays_d, bg_color=bg, perturb=False)
    out["image"].backward(gradient=grad)
    E.grid_encode_backward = orig
    M = int(out["counter"][0])
    cap = net._march.capacity
    m_dev = net._march.counter[0:1]
    levels = net.encoder.levels
    dtable = torch.zeros_like(net.encoder.embeddings.data)
    lib = B.get_lib()
    lib.lnerf_debug_bin_stamps.argtypes = [ctypes.c_void_p]
    lib.lnerf_debug_bin_stamps.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * 16)()
    for _ in range(3):
        E.grid_encode_backward(store["xyzs"], 1.0, store["dfeat"], levels, cap, m_dev, cap, dtable, variant=3)
    torch.cuda.synchronize()
    lib.lnerf_debug_bin_stamps(buf)
    n = 10
    for _ in range(n):
        E.grid_encode_backward(store["xyzs"], 1.0, store["dfeat"], levels, cap, m_dev, cap, dtable, variant=3)
    torch.cuda.synchronize()
    lib.lnerf_debug_bin_stamps(buf)
    tiles = ((M + 511) // 512) * 16
    tot = sum(buf[i] for i in range(len(PHASES)))
    res = {"M": M, "tiles_per_launch": tiles, "cycles_per_tile_total": round(tot / n / tiles, 1), "phases_cycles_per_tile": {}}
    for i, name in enumerate(PHASES):
        res["phases_cycles_per_tile"][name] = round(buf[i] / n / tiles, 1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
