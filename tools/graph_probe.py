#!/usr/bin/env python3
"""Which part of the step can be captured into a hipGraph?  Each stage runs in its own subprocess
(a failing capture can take the process down).  Usage: python tools/graph_probe.py"""
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGES = ["torch_only", "rays", "march", "march_noperturb", "encode", "mlp", "field", "composite", "render_nograd",
          "render_bwd"]


def run_stage(stage):
    for _p in (ROOT, os.path.join(ROOT, "latent-nerf-test_amd")):
        sys.path.insert(0, _p)
    import torch
    import bench
    from src.latent_nerf.models import encoding as E
    from src.latent_nerf.raymarching import raymarching as rm
    dev = torch.device("cuda:0")
    net, pose, intr, bg, grad = bench.build(dev, "f32", 0, 0)
    H = W = 64
    rays_o, rays_d = rm.get_rays(pose, intr, H, W)
    ro, rd = rays_o.view(-1, 3), rays_d.view(-1, 3)
    nears, fars = rm.near_far_from_aabb(ro, rd, [-1, -1, -1, 1, 1, 1], 0.1)
    march = rm.march_rays_train(ro, rd, 1.0, net.density_bitfield, 1, 128, nears, fars, perturb=False)
    cap = march.capacity
    m_dev = march.counter[0:1]
    noises = torch.rand(H * W, device=dev)

    def fn():
        if stage == "torch_only":
            return torch.rand(1000, device=dev) * 2
        if stage == "rays":
            a, b = rm.get_rays(pose, intr, H, W)
            return rm.near_far_from_aabb(a.view(-1, 3), b.view(-1, 3), [-1, -1, -1, 1, 1, 1], 0.1)
        if stage == "march":
            return rm.march_rays_train(ro, rd, 1.0, net.density_bitfield, 1, 128, nears, fars, perturb=True, out=march).counter
        if stage == "march_noperturb":
            return rm.march_rays_train(ro, rd, 1.0, net.density_bitfield, 1, 128, nears, fars, perturb=False, noises=noises,
                                       out=march).counter
        if stage == "encode":
            return E.grid_encode_forward(march.xyzs, 1.0, net.encoder.embeddings.data, net.encoder.levels, cap, m_dev, cap)
        if stage in ("mlp", "field"):
            with torch.no_grad():
                return net.field(march.xyzs, cap, m_dev, cap)
        if stage == "composite":
            with torch.no_grad():
                s, c = net.field(march.xyzs, cap, m_dev, cap)
                return rm.composite_rays_train(s, c, march.deltas, march.rays, 1e-4, bg)
        if stage == "render_nograd":
            with torch.no_grad():
                return net.render(rays_o, rays_d, bg_color=bg, perturb=True)["image"]
        if stage == "render_bwd":
            out = net.render(rays_o, rays_d, bg_color=bg, perturb=True)
            out["image"].backward(gradient=grad)
            g = net.encoder.embeddings.grad
            for p in net.parameters():
                p.grad = None
            return g
        raise ValueError(stage)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("STAGE %s OK" % stage, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_stage(sys.argv[1])
    else:
        for st in STAGES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), st], capture_output=True, text=True, timeout=300)
            tail = (r.stdout + r.stderr).strip().splitlines()
            ok = any(("STAGE %s OK" % st) in l for l in tail)
            print("%-16s rc=%-4d %s" % (st, r.returncode, "OK" if ok else "FAIL: " + " | ".join(tail[-3:])[:300]), flush=True)
