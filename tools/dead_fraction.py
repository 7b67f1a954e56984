#!/usr/bin/env python3
"""Share of samples whose upstream gradient is exactly zero on the bench scene (rays past T < 1e-4)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "latent-nerf-test_amd")):
    sys.path.insert(0, _p)
import torch
import bench
from src.latent_nerf.raymarching import raymarching as rm
dev = torch.device("cuda:0")
net, pose, intr, bg, grad = bench.build(dev, "bf16", 0, 0, "bf16")
rays_o, rays_d = rm.get_rays(pose, intr, bench.H, bench.W)
out = net.render(rays_o, rays_d, bg_color=bg, perturb=True)
store = {}
out["sigmas"].register_hook(lambda g: store.__setitem__("ds", g.detach().clone()))
out["image"].backward(gradient=grad)
M = int(out["counter"][0])
ds = store["ds"][:M]
dead = (ds == 0)
# fully dead aligned blocks of 64 / 128 samples
def blocks(n):
    k = M // n
    return float(dead[:k * n].view(k, n).all(1).float().mean())
rays = out["rays"].cpu()
print(json.dumps({"M": M, "dead_fraction": float(dead.float().mean()), "dead_blocks64": blocks(64), "dead_blocks128": blocks(128),
                  "live_rays": int((rays[:, 2] > 0).sum())}))
