#!/usr/bin/env python3
"""Scatter (H6) of the bench step, timed per configuration in interleaved rounds (HIP events on the launch stream)
and checked against the global-atomics variant.  Per-kernel times: run under
`rocprofv3 --kernel-trace --stats` (tools/run_scatter_profile.sh).

    python tools/bench_scatter.py [--rounds N] [--check]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "latent-nerf-test_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--configs", default="3:0")   # per_cu:wgs pairs
    ap.add_argument("--tune", default="")            # extra variants: "key=val+key=val,key=val" (comma separates variants)
    args = ap.parse_args()
    import bench
    from src.latent_nerf.models import encoding as E
    from src.latent_nerf.raymarching import backend as B
    from src.latent_nerf.raymarching import raymarching as rm
    dev = torch.device("cuda:0")
    net, pose, intr, bg, grad = bench.build(dev, "bf16", 0, 0, "bf16")
    rays_o, rays_d = rm.get_rays(pose, intr, bench.H, bench.W)
    store = {}
    orig_b, orig_a = E.grid_encode_backward, E.grid_encode_backward_adam

    def grab(xyzs_, bound_, dfeat_, *a, **k):
        store["dfeat"] = dfeat_.clone()
        store["xyzs"] = xyzs_
        return orig_b(xyzs_, bound_, dfeat_, *a, **k)
    E.grid_encode_backward = grab
    out = net.render(rays_o, rays_d, bg_color=bg, perturb=True)
    out["image"].backward(gradient=grad)
    E.grid_encode_backward = orig_b
    M = int(out["counter"][0])
    cap = net._march.capacity
    m_dev = net._march.counter[0:1]
    levels = net.encoder.levels
    xyzs, dfeat = store["xyzs"], store["dfeat"]
    res = {"M": M, "capacity": cap, "build": B.get_lib().lnerf_build_info().decode(),
           "dead_fraction": float((dfeat[:, :M, :] == 0).all(-1).float().mean())}
    dtable = torch.zeros_like(net.encoder.embeddings.data)

    def scatter(variant):
        E.grid_encode_backward(xyzs, 1.0, dfeat, levels, cap, m_dev, cap, dtable, variant=variant)

    if args.check:
        ref = torch.zeros_like(dtable)
        E.grid_encode_backward(xyzs, 1.0, dfeat, levels, cap, m_dev, cap, ref, variant=0)
        chk = {}
        ws = E.scatter_workspace(levels, cap, dev)
        for v in (2, 3):
            for per_cu in (3,):
                B.call("lnerf_set_tuning", b"scatter_bin_per_cu", per_cu)
                ws.fill_(0x7F)       # poison: a reserved record slot that is never written would add ~3e38
                a = torch.zeros_like(dtable)
                E.grid_encode_backward(xyzs, 1.0, dfeat, levels, cap, m_dev, cap, a, variant=v)
                b = torch.zeros_like(dtable)
                E.grid_encode_backward(xyzs, 1.0, dfeat, levels, cap, m_dev, cap, b, variant=v)
                err = float((a - ref).abs().max())
                chk["v%d_percu%d" % (v, per_cu)] = {"max_abs_err": err, "ref_max": float(ref.abs().max()),
                                                     "bitwise_repeatable": bool(torch.equal(a, b)),
                                                     "per_level_err": [float((a - ref)[levels.offsets[l]:levels.offsets[l + 1]].abs().max())
                                                                       for l in range(16)]}
        res["check"] = chk
        B.call("lnerf_set_tuning", b"scatter_bin_per_cu", 3)

    fns = {}
    for c in args.configs.split(","):
        per_cu, wgs = [int(v) for v in c.split(":")]

        def f(per_cu=per_cu, wgs=wgs):
            B.call("lnerf_set_tuning", b"scatter_bin_per_cu", per_cu)
            B.call("lnerf_set_tuning", b"scatter_bin_wgs", wgs)
            scatter(3)
        fns["v3_percu%d_wgs%d" % (per_cu, wgs)] = f
    defaults = {"scatter_compact_max_res": 512, "scatter_skip_zero": 1, "scatter_reduce_threads": 1024}
    for var in [v for v in args.tune.split(",") if v]:
        kv = dict((k, int(x)) for k, x in (t.split("=") for t in var.split("+")))

        def f(kv=kv):
            B.call("lnerf_set_tuning", b"scatter_bin_per_cu", 3)
            B.call("lnerf_set_tuning", b"scatter_bin_wgs", 0)
            for k, x in kv.items():
                B.call("lnerf_set_tuning", k.encode(), x)
            scatter(3)
            for k in kv:
                if k in defaults:
                    B.call("lnerf_set_tuning", k.encode(), defaults[k])
        fns["v3_" + var] = f
    for _ in range(3):
        for f in fns.values():
            f()
    torch.cuda.synchronize()
    samples = {k: [] for k in fns}
    for _ in range(args.rounds):
        for k, f in fns.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            f()
            b.record()
            samples[k].append((a, b))
    torch.cuda.synchronize()
    res["scatter_ms(median,min)"] = {}
    for k, evs in samples.items():
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        res["scatter_ms(median,min)"][k] = (round(ts[len(ts) // 2], 4), round(ts[0], 4))
    B.call("lnerf_set_tuning", b"scatter_bin_per_cu", 3)
    B.call("lnerf_set_tuning", b"scatter_bin_wgs", 0)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
