#!/usr/bin/env python3
"""Kernel-level A/B timings on the bench scene (64x64 rays, 128^3 sphere occupancy, M ~ 4.3e5 samples).

Variants of one op are timed in interleaved rounds inside one process (guide §5.4 rule 24) with HIP
events on the launch stream; prints one JSON object.  Usage (GPU box):
    python tools/microbench.py [gather] [scatter] [mlp]
"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "latent-nerf-test_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402


def timed(fns, rounds=20, warm=3):
    """fns: dict name -> callable.  Returns name -> (median_ms, min_ms)."""
    for _ in range(warm):
        for f in fns.values():
            f()
    torch.cuda.synchronize()
    samples = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            f()
            b.record()
            samples[k].append((a, b))
    torch.cuda.synchronize()
    out = {}
    for k, evs in samples.items():
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        out[k] = (round(ts[len(ts) // 2], 4), round(ts[0], 4))
    return out


def main():
    which = set(sys.argv[1:]) or {"gather", "scatter", "mlp"}
    import bench
    from src.latent_nerf.models import encoding as E
    from src.latent_nerf.raymarching import backend as B
    from src.latent_nerf.raymarching import raymarching as rm
    dev = torch.device("cuda:0")
    net, pose, intr, bg, grad = bench.build(dev, "f32", 1, 0)
    rays_o, rays_d = rm.get_rays(pose, intr, bench.H, bench.W)
    out = net.render(rays_o, rays_d, bg_color=bg, perturb=False)
    M = int(out["counter"][0])
    cap = net._march.capacity
    xyzs = net._march.xyzs
    m_dev = net._march.counter[0:1]
    levels = net.encoder.levels
    table = net.encoder.embeddings.data
    table_bf = table.to(torch.bfloat16)
    res = {"M": M, "capacity": cap}

    if "gather" in which:
        f32o = torch.empty(16, cap, 2, device=dev)
        bf16o = torch.empty(16, cap, 2, device=dev, dtype=torch.bfloat16)
        fns = {}
        for dd, pl in ((512, 1), (512, 2)):
            def mk(tab, out, dd=dd, pl=pl):
                def f():
                    B.call("lnerf_set_tuning", b"gather_dedup_max_res", dd)
                    B.call("lnerf_set_tuning", b"gather_pair_loads", pl)
                    E.grid_encode_forward(xyzs, 1.0, tab, levels, cap, m_dev, cap, out, variant=0)
                return f
            fns["f32tab_bf16out_dedup%d_pairs%d" % (dd, pl)] = mk(table, bf16o)
            fns["bf16tab_bf16out_dedup%d_pairs%d" % (dd, pl)] = mk(table_bf, bf16o)
        for wg in (64, 128, 256, 512):
            def mk2(tab, out, wg=wg):
                def f():
                    B.call("lnerf_set_tuning", b"gather_dedup_max_res", 512)
                    B.call("lnerf_set_tuning", b"gather_pair_loads", 1)
                    B.call("lnerf_set_tuning", b"gather_wgs_per_xcd", wg)
                    E.grid_encode_forward(xyzs, 1.0, tab, levels, cap, m_dev, cap, out, variant=2)
                return f
            fns["bf16tab_bf16out_xcdsets_wg%d" % wg] = mk2(table_bf, bf16o)
            fns["f32tab_bf16out_xcdsets_wg%d" % wg] = mk2(table, bf16o)
        t = timed(fns)
        B.call("lnerf_set_tuning", b"gather_pair_loads", 1)
        res["gather_ms(median,min)"] = t
        bps = lambda k: (1024 if k.startswith("f32tab") else 512) + 12 + (128 if "f32out" in k else 64)
        res["gather_GBps_algorithmic"] = {k: round(M * bps(k) / (v[0] * 1e-3) / 1e9, 1) for k, v in t.items()}
        B.call("lnerf_set_tuning", b"gather_dedup_max_res", 512)

    if "scatter" in which:
        # a REAL dfeat (from a render + backward of the bench step), so that zero gradients appear where they do
        store = {}
        import src.latent_nerf.models.encoding as Emod
        orig = Emod.grid_encode_backward

        def grab(xyzs_, bound_, dfeat_, *a, **k):
            store["dfeat"] = dfeat_.clone()
            return orig(xyzs_, bound_, dfeat_, *a, **k)
        Emod.grid_encode_backward = grab
        out2 = net.render(rays_o, rays_d, bg_color=bg, perturb=False)
        out2["image"].backward(gradient=grad)
        Emod.grid_encode_backward = orig
        dfeat = store["dfeat"]
        res["dfeat_zero_fraction"] = float((dfeat[:, :M, :] == 0).all(-1).float().mean())
        dtable = torch.zeros_like(table)
        fns = {}
        for v in (2, 3):
            def f(v=v):
                E.grid_encode_backward(xyzs, 1.0, dfeat, levels, cap, m_dev, cap, dtable, variant=v)
            fns["variant%d" % v] = f
        t = timed(fns, rounds=10)
        res["scatter_ms(median,min)"] = t

    if "update" in which:
        # the three ways to get from dfeat to an updated table (real dfeat of the bench step, random-init state):
        # fused (N = 1), bf16 gradient sink + Adam reading bf16 (N > 1, bf16 wire), f32 gradient + Adam (N > 1, f32)
        import src.latent_nerf.models.encoding as Emod
        from src.latent_nerf.training.optimizer import FusedAdam
        from src.latent_nerf.raymarching.raymarching import _p, _stream
        store = {}
        orig = Emod.grid_encode_backward

        def grab(xyzs_, bound_, dfeat_, *a, **k):
            store["dfeat"] = dfeat_.clone()
            return orig(xyzs_, bound_, dfeat_, *a, **k)
        Emod.grid_encode_backward = grab
        out2 = net.render(rays_o, rays_d, bg_color=bg, perturb=False)
        out2["image"].backward(gradient=grad)
        Emod.grid_encode_backward = orig
        dfeat = store["dfeat"]
        enc = net.encoder
        emb = enc.embeddings
        opt = FusedAdam([{"params": [emb], "lr": 1e-7}], encoder=enc, fuse_table_update=True)
        m, v = opt.big[0][1], opt.big[0][2]
        sink = Emod.GradSink(emb.data)
        zero = torch.zeros_like(emb.data)

        def fused():
            opt.fused.armed = False
            Emod.grid_encode_backward_adam(xyzs, 1.0, dfeat, enc, cap, m_dev, cap, 3)

        def sink_path():
            enc.grad_sink = sink
            Emod.grid_encode_backward_bf16(xyzs, 1.0, dfeat, enc, cap, m_dev, cap, 3)
            enc.grad_sink = None
            B.call("lnerf_adam_step", _p(emb.data), _p(sink.wire), B.BF16, _p(m), _p(v), None, emb.numel(), 1e-7, 0.9, 0.99,
                   1e-15, 1, None, 1.0, 0, _stream())

        def f32_path():
            zero.zero_()
            Emod.grid_encode_backward(xyzs, 1.0, dfeat, levels, cap, m_dev, cap, zero, variant=3)
            B.call("lnerf_adam_step", _p(emb.data), _p(zero), B.F32, _p(m), _p(v), None, emb.numel(), 1e-7, 0.9, 0.99,
                   1e-15, 1, None, 1.0, 0, _stream())
        res["table_update_ms(median,min)"] = timed({"fused_backward_adam": fused, "bf16_sink_then_adam": sink_path,
                                                    "f32_grad_then_adam": f32_path}, rounds=10)
        enc.fused_update = None

    if "scatter_levels" in which:
        # one launch pair per level (num_levels = 1 slices of the level table): per-level cost under rocprofv3
        import ctypes
        from src.latent_nerf.raymarching.raymarching import _p, _stream
        dfeat = torch.randn(16, cap, 2, device=dev)
        dtable = torch.zeros_like(table)
        ws = E.scatter_workspace(levels, cap, dev)
        per = {}
        for l in range(16):
            offs = (ctypes.c_int32 * 2)(levels.offsets[l], levels.offsets[l + 1])
            sc = (ctypes.c_float * 1)(levels.scales[l])
            rs = (ctypes.c_int32 * 1)(levels.resolutions[l])
            dptr = ctypes.c_void_p(dfeat.data_ptr() + l * cap * 2 * 4)

            def f(offs=offs, sc=sc, rs=rs, dptr=dptr):
                B.call("lnerf_grid_encode_backward", _p(xyzs), 1.0, dptr, B.F32, 1, 2, offs, sc, rs, cap, _p(m_dev), cap,
                       _p(dtable), 3, _p(ws), ws.numel(), _stream())
            per["level%02d" % l] = f
        res["scatter_level_ms(median,min)"] = timed(per, rounds=6, warm=1)

    if "mlp" in which:
        from src.latent_nerf.models.network_grid import _SigmaLatentMLP
        featb = (torch.randn(16, cap, 2, device=dev) * 0.3).to(torch.bfloat16)
        fns = {}
        for wps, nb in ((2, 512), (2, 768), (2, 1024), (4, 1024), (2, 1280)):
            def f(n=nb, wps=wps):
                B.call("lnerf_set_tuning", b"mlp_fwd_blocks", n)
                B.call("lnerf_set_tuning", b"mlp_fwd_wps", wps)
                with torch.no_grad():
                    _SigmaLatentMLP.apply(featb, xyzs, net.w1, net.b1, net.w2, net.b2, net.w3, net.b3, cap, m_dev, cap,
                                          5.0, 0.2, B.BF16, net._mlp_ws)
            fns["mlp_fwd_bf16_wps%d_blocks%d" % (wps, nb)] = f
        res["mlp_ms(median,min)"] = timed(fns)
        B.call("lnerf_set_tuning", b"mlp_fwd_blocks", 768)
        B.call("lnerf_set_tuning", b"mlp_fwd_wps", 2)
        # backward (recomputes the forward): real upstream gradients of the bench step
        import src.latent_nerf.models.network_grid as NG
        store, keep = {}, {}
        orig, orig_chk = B.call, NG._chk

        def spy(name, *a):
            if name == "lnerf_mlp_backward" and "args" not in store:
                store["args"] = list(a)
            return orig(name, *a)

        def chk(t_, name, *a_, **k_):
            keep[name] = t_            # the upstream gradients stay allocated after autograd drops them
            return orig_chk(t_, name, *a_, **k_)
        NG._b.call = spy
        NG._chk = chk
        netb, _, _, bgb, gradb = bench.build(dev, "bf16", 0, 0, "bf16")
        out3 = netb.render(rays_o, rays_d, bg_color=bgb, perturb=False)
        out3["image"].backward(gradient=gradb, retain_graph=True)   # saved feat / sigmas stay allocated
        NG._b.call = orig
        NG._chk = orig_chk
        args = store["args"]
        capb = netb._march.capacity
        mine = [torch.empty(16, capb, 2, device=dev)] + [torch.empty_like(p_) for p_ in
                                                         (netb.w1, netb.b1, netb.w2, netb.b2, netb.w3, netb.b3)]
        for i, t_ in enumerate(mine):
            args[18 + i] = t_.data_ptr()
        Mb = int(out3["counter"][0])
        fns = {}
        for var, nb in ((0, 512), (0, 448), (1, 256)):
            def f(nb=nb, var=var):
                B.call("lnerf_set_tuning", b"mlp_bwd_variant", var)
                B.call("lnerf_set_tuning", b"mlp_bwd_blocks", nb)
                B.call("lnerf_mlp_backward", *args)
            fns["mlp_bwd_bf16_v%d_blocks%d" % (var, nb)] = f
        # the variants against each other: dfeat and the six weight gradients of the same inputs
        outs = {}
        for k, f in fns.items():
            for t_ in mine:
                t_.fill_(float("nan"))
            f()
            torch.cuda.synchronize()
            outs[k] = [mine[0][:, :Mb].clone()] + [t_.clone() for t_ in mine[1:]]
        base = outs["mlp_bwd_bf16_v0_blocks512"]
        res["mlp_bwd_vs_v0(max_abs_diff / max_abs)"] = {
            k: [[float((a_ - b_).abs().max()), float(b_.abs().max())] for a_, b_ in zip(v, base)] for k, v in outs.items()}
        res["mlp_bwd_ms(median,min)"] = timed(fns)
        B.call("lnerf_set_tuning", b"mlp_bwd_blocks", 512)
        B.call("lnerf_set_tuning", b"mlp_bwd_variant", 0)

    print(json.dumps(res))


if __name__ == "__main__":
    main()
