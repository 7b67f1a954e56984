#!/usr/bin/env python3
"""Share of samples whose upstream gradient is exactly zero (rays past T < 1e-4) on the bench scene, at the
random-init state and while the bench's optimiser steps move the field (lr as bench.py, or --lr)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "latent-nerf-test_amd")):
    sys.path.insert(0, _p)
import torch  # noqa: E402
import bench  # noqa: E402
from src.latent_nerf.raymarching import raymarching as rm  # noqa: E402
from src.latent_nerf.training.optimizer import FusedAdam  # noqa: E402

lr = float(sys.argv[1]) if len(sys.argv) > 1 else bench.LR
dev = torch.device("cuda:0")
net, pose, intr, bg, grad = bench.build(dev, "bf16", 0, 0, "bf16")
opt = FusedAdam(net.get_params(lr), betas=(0.9, 0.99), eps=1e-15, encoder=net.encoder)
rays_o, rays_d = rm.get_rays(pose, intr, bench.H, bench.W)
out_rows = []
for step in range(0, 2001):
    out = net.render(rays_o, rays_d, bg_color=bg, perturb=True)
    probe = step in (0, 5, 10, 20, 30, 60, 100, 230, 500, 1000, 2000)
    store = {}
    if probe:
        out["sigmas"].register_hook(lambda g: store.__setitem__("ds", g.detach().clone()))
    out["image"].backward(gradient=grad)
    if probe:
        M = int(out["counter"][0])
        ds = store["ds"][:M]
        out_rows.append({"step": step, "M": M, "dead_fraction": round(float((ds == 0).float().mean()), 4),
                         "mean_weights_sum": round(float(out["weights_sum"].mean()), 4)})
    opt.step()
print(json.dumps({"lr": lr, "trace": out_rows}))
