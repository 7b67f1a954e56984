#!/usr/bin/env python3
"""GPU box: training-quality A/B of the table layouts (render.gridtype hash | blocked | tiled) through the real Trainer:
same seeds, same views, same seeded synthetic guidance, 64 x 64 x 4 latents, 128^3 grid, bf16.  Every `--every` steps the
current field is rendered from fixed evaluation poses and compared with the guidance's target latents of those views'
direction buckets (mean squared error: what the guidance pulls towards zero).  One JSON line per layout: the error curve,
the final error, steps per second.

    python tools/ab_layout.py [--steps 500] [--every 50] [--layouts hash,blocked,tiled] [--seeds 0,1]

Decides one thing: whether a layout that renders faster (blocked: one 64-byte line per 4 x 2 x 2 vertex block) also LEARNS
as well as Instant-NGP's vertex hash -- a different collision pattern could cost quality that a kernel timing never shows."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "latent-nerf-test_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402


def run(layout, seed, steps, every, root):
    from src.latent_nerf.configs.train_config import TrainConfig, apply_overrides
    from src.latent_nerf.training.trainer import Trainer
    cfg = apply_overrides(TrainConfig(), {
        "log.exp_name": "ab_%s_%d" % (layout, seed), "log.exp_root": root, "render.train_h": 64, "render.train_w": 64,
        "render.grid_size": 128, "render.eval_h": 64, "render.eval_w": 64, "log.eval_size": 8, "log.full_eval_size": 1,
        "log.save_interval": 10 ** 9, "log.quiet": True, "optim.fp16": True, "optim.seed": seed, "guide.text": "ab",
        "optim.iters": steps, "render.gridtype": layout})
    dev = torch.device("cuda", 0)
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    tr = Trainer(cfg, device=dev)
    tr.full_eval = lambda: None
    val = tr.dataloaders["val"]

    def err():
        tr.nerf.eval()
        tot = 0.0
        with torch.cuda.stream(tr.stream):
            for i in range(len(val)):
                data = val.collate(i)
                pred, _ = tr.eval_render(data)
                tgt = tr.diffusion.targets[int(data["dir"][0])][None]
                tot += float((pred - tgt).pow(2).mean())
        tr.nerf.train()
        return tot / len(val)

    curve = [(0, err())]
    t_train = 0.0
    for s in range(every, steps + 1, every):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train(iters=s)
        torch.cuda.synchronize()
        t_train += time.perf_counter() - t0
        curve.append((s, err()))
    return {"layout": layout, "seed": seed, "steps": steps, "curve": [(s, round(e, 6)) for s, e in curve],
            "final_error": curve[-1][1], "steps_per_s": steps / t_train,
            "occupied_cells": int(tr.nerf.density_bitfield.count_nonzero())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--layouts", default="hash,blocked,tiled")
    ap.add_argument("--seeds", default="0,1")
    a = ap.parse_args()
    root = tempfile.mkdtemp(prefix="lnerf_ab_layout_")
    try:
        for seed in [int(s) for s in a.seeds.split(",")]:
            for layout in a.layouts.split(","):
                print(json.dumps(run(layout, seed, a.steps, a.every, root)), flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
