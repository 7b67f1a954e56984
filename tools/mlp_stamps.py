#!/usr/bin/env python3
"""Phase breakdown of the bf16 MLP kernels from in-kernel shader-clock stamps (diagnostic build: tools/run_bin_stamps.sh
build, which compiles every kernel file with -DLNERF_STAMPS).

    LNERF_HIP_LIB=latent-nerf-test_amd/lib/liblnerf_hip_stamps.so python3 tools/mlp_stamps.py
Cycles are those of wave 0 of every workgroup, per 32-sample wave step (forward) / per 128-sample workgroup step
(backward); a phase's time includes whatever it had to wait for."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "latent-nerf-test_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

FWD = ["0 next step's loads issued", "1 layer 1 (8 MFMA) + relu/pack", "2 layer 2 (16 MFMA) + relu/pack",
       "3 layer 3 (4 MFMA) + exp + stores issued"]
BWD = {0: "0 next loads issued, dZ3 fragment", 4: "4 __syncthreads_or", 1: "1 layer 1 + pack", 2: "2 layer 2 + pack",
       5: "5 barrier (images free)", 6: "6 stage-1 images + barrier", 7: "7 dW3 (ld_tr + MFMA)", 8: "8 dA2 chain + mask/pack",
       9: "9 barrier + stage-2 images + barrier", 10: "10 dW2", 11: "11 dA1 chain + mask/pack",
       12: "12 barrier + stage-3 images + barrier", 13: "13 dW1", 14: "14 dX + stores issued"}


def main():
    import bench
    from src.latent_nerf.raymarching import backend as B
    from src.latent_nerf.raymarching import raymarching as rm
    dev = torch.device("cuda:0")
    net, pose, intr, bg, grad = bench.build(dev, "bf16", 0, 0, "bf16")
    rays_o, rays_d = rm.get_rays(pose, intr, bench.H, bench.W)
    lib = B.get_lib()
    lib.lnerf_debug_mlp_stamps.argtypes = [ctypes.c_void_p]
    lib.lnerf_debug_mlp_stamps.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * 32)()

    def step():
        out = net.render(rays_o, rays_d, bg_color=bg, perturb=False)
        out["image"].backward(gradient=grad)
        return out
    for _ in range(3):
        out = step()
    torch.cuda.synchronize()
    lib.lnerf_debug_mlp_stamps(buf)
    n = 10
    for _ in range(n):
        out = step()
    torch.cuda.synchronize()
    lib.lnerf_debug_mlp_stamps(buf)
    M = int(out["counter"][0])
    steps = (M + 127) // 128          # workgroup steps per launch (= wave steps of wave 0)
    res = {"M": M, "steps_per_launch": steps, "forward_cycles_per_step": {}, "backward_cycles_per_step": {}}
    for i, name in enumerate(FWD):
        res["forward_cycles_per_step"][name] = round(buf[i] / n / steps, 1)
    res["forward_cycles_per_step"]["total"] = round(sum(buf[i] for i in range(4)) / n / steps, 1)
    for i in sorted(BWD, key=lambda k: [0, 4, 1, 2, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14].index(k)):
        res["backward_cycles_per_step"][BWD[i]] = round(buf[16 + i] / n / steps, 1)
    res["backward_cycles_per_step"]["total"] = round(sum(buf[16 + i] for i in range(16)) / n / steps, 1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
