"""Freeze the CPU oracle of the latent-NeRF path (oracle/nerf_oracle.py, rows H1-H11 of SURVEY.md §8) into small
golden files, so that an edit which moves oracle and kernels together no longer goes unnoticed.

    python tests/golden/make_nerf_golden.py        # rewrites tests/golden/nerf_golden_*.npz

The reference holds no source, tests or vectors for this path -- these files pin the
repository's OWN oracle at the revision that generated them, nothing more: parity with the reference stays unpinned.
Both the oracle (tests/test_golden_cpu.py, CPU) and the HIP path (tests/test_gpu_golden.py, `-m gpu`) are compared
with the files.

Two files, sizes after SURVEY.md §8(c)'s suggestion (N = 256 rays, 32^3 grid, T = 2^10):
  nerf_golden_frame.npz  one whole frame, forward + backward + one Adam step: 16 x 16 rays, 32^3 occupancy grid, max_steps 128,
                         L = 16 levels (the fused MLP takes 32 features), F = 2, base 16, T = 2^10, jittered march,
                         random background colours, SDS-like upstream gradient.  Every intermediate is stored.
  nerf_golden_grid.npz   the hash-grid encoder alone with L = 4 levels (base 4, top 32, T = 2^10: two dense, two
                         hashed levels), forward and table gradient, in f32 and with the bf16-rounded table.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import nerf_oracle as O  # noqa: E402

FRAME = os.path.join(HERE, "nerf_golden_frame.npz")
GRID = os.path.join(HERE, "nerf_golden_grid.npz")


def frame_case():
    """Inputs of the whole-frame case (deterministic from the seeds below)."""
    G, HW, log2_T, max_steps = 32, 16, 10, 128     # max_steps 128: step 2 sqrt(3) / 128, ~35 samples per hit ray
    g = torch.Generator().manual_seed(20261004)
    lv = O.make_grid_levels(16, 2, 16, 2048, log2_T)
    table = torch.randn(lv.offsets[-1], 2, generator=g) * 0.1
    params = O.init_mlp_params(32, 64, 5, seed=11)
    grid = O.sphere_density_grid(G=G, radius=0.5)
    bits = O.packbits(grid.reshape(-1), 0.01)
    theta, phi, radius, fovy = 62.0, 25.0, 1.3, 55.0
    f = HW / (2 * math.tan(math.radians(fovy) / 2))
    c2w = O.pose_from_angles(math.radians(theta), math.radians(phi), radius)
    N = HW * HW
    bg = torch.rand(N, 4, generator=g)
    noises = torch.rand(N, generator=g)
    upstream = torch.randn(N, 4, generator=g) * math.sqrt(0.5) * 0.5     # w = sqrt(a)(1 - a), a = .5
    return dict(G=G, HW=HW, log2_T=log2_T, lv=lv, table=table, params=params, grid=grid, bits=bits, c2w=c2w,
                focal=f, bg=bg, noises=noises, upstream=upstream, pose=(theta, phi, radius, fovy), max_steps=max_steps)


def run_frame(case, bf16=False):
    """The oracle's frame on `case`: forward, backward, one Adam step of every parameter."""
    lv, HW, G = case["lv"], case["HW"], case["G"]
    ro, rd = O.get_rays(case["c2w"], case["focal"], case["focal"], HW / 2, HW / 2, HW, HW)
    table = case["table"].clone().requires_grad_()
    params = {k: v.clone().requires_grad_() for k, v in case["params"].items()}
    ref = O.render_frame(ro[0], rd[0], table, params, lv, case["bits"], G=G, noises=case["noises"],
                         bg_color=case["bg"], bf16_mlp=bf16, bf16_table=bf16, max_steps=case["max_steps"])
    ref["feat"].retain_grad()
    ref["sigmas"].retain_grad()
    ref["rgbs"].retain_grad()
    ref["image"].backward(case["upstream"])
    out = {"rays_o": ro[0], "rays_d": rd[0], "nears": ref["nears"], "fars": ref["fars"], "xyzs": ref["xyzs"],
           "dirs": ref["dirs"], "deltas": ref["deltas"], "rays": ref["rays"], "M": torch.tensor(ref["M"]),
           "feat": ref["feat"].detach(), "sigmas": ref["sigmas"].detach(), "rgbs": ref["rgbs"].detach(),
           "image": ref["image"].detach(), "depth": ref["depth"].detach(), "weights_sum": ref["weights_sum"].detach(),
           "dfeat": ref["feat"].grad, "dsigmas": ref["sigmas"].grad, "drgbs": ref["rgbs"].grad, "dtable": table.grad}
    for k, p in params.items():
        out["d" + k] = p.grad
    # one Adam step (betas (0.9, 0.99), eps 1e-15: src/latent_paint/training/trainer.py:93-95); table lr x 10
    z = torch.zeros_like(table)
    p1, m1, v1 = O.adam_step(table.detach(), table.grad, z, z, 1, 1e-2)
    out.update(table_after=p1, table_m=m1, table_v=v1)
    for k, p in params.items():
        q, _, _ = O.adam_step(p.detach(), p.grad, torch.zeros_like(p), torch.zeros_like(p), 1, 1e-3)
        out[k + "_after"] = q
    return out


def grid_case():
    g = torch.Generator().manual_seed(4242)
    lv = O.make_grid_levels(4, 2, 4, 32, 10)
    table = torch.randn(lv.offsets[-1], 2, generator=g) * 0.1
    M = 777
    x = torch.rand(M, 3, generator=g) * 2 - 1
    x[:5] = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0], [0.0, 0.0, 0.0], [1.0, -1.0, 0.5], [-0.25, 0.75, 1.0]])
    x[5:40] = x[5:6] + torch.linspace(0, 0.02, 35)[:, None]      # a run of samples along a short segment (one ray)
    up = torch.randn(M, lv.num_levels * 2, generator=g)
    up[100:140] = 0.0                                             # samples behind a ray's termination: zero gradient
    return dict(lv=lv, table=table, x=x, upstream=up)


def run_grid(case, bf16_table=False):
    lv = case["lv"]
    table = case["table"].clone().requires_grad_()
    tab = table + (O._bf16r(table) - table).detach() if bf16_table else table
    feat = O.grid_encode((case["x"] + 1.0) / 2.0, tab, lv)
    feat.backward(case["upstream"])
    return {"feat": feat.detach(), "dtable": table.grad}


def _np(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def main():
    torch.set_num_threads(1)   # fixed summation order on any host
    c = frame_case()
    blob = {"in_table": c["table"], "in_bits": c["bits"], "in_grid": c["grid"], "in_c2w": c["c2w"],
            "in_focal": torch.tensor(c["focal"]), "in_bg": c["bg"], "in_noises": c["noises"],
            "in_upstream": c["upstream"], "in_pose": torch.tensor(c["pose"]), "in_max_steps": torch.tensor(c["max_steps"]),
            "in_offsets": torch.tensor(c["lv"].offsets), "in_scales": torch.tensor(c["lv"].scales),
            "in_resolutions": torch.tensor(c["lv"].resolutions)}
    for k, v in c["params"].items():
        blob["in_" + k] = v
    for tag, bf in (("f32", False), ("bf16", True)):
        for k, v in run_frame(c, bf16=bf).items():
            if bf and k in ("rays_o", "rays_d", "nears", "fars", "xyzs", "dirs", "deltas", "rays", "M", "dfeat"):
                continue   # discrete part: identical to the f32 run; dfeat: kept once
            blob["%s_%s" % (tag, k)] = v
    # integer known answers: Morton codes and the packed bitfield of a small grid
    coords = torch.tensor([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [3, 5, 7], [31, 31, 31], [1023, 0, 1023]])
    blob["kat_morton_coords"] = coords
    blob["kat_morton_codes"] = O.morton3d(coords)
    np.savez_compressed(FRAME, **_np(blob))
    gc = grid_case()
    gb = {"in_table": gc["table"], "in_x": gc["x"], "in_upstream": gc["upstream"],
          "in_offsets": torch.tensor(gc["lv"].offsets), "in_scales": torch.tensor(gc["lv"].scales),
          "in_resolutions": torch.tensor(gc["lv"].resolutions)}
    for tag, bf in (("f32", False), ("bf16", True)):
        for k, v in run_grid(gc, bf16_table=bf).items():
            gb["%s_%s" % (tag, k)] = v
    np.savez_compressed(GRID, **_np(gb))
    for f in (FRAME, GRID):
        print(f, os.path.getsize(f), "bytes")


if __name__ == "__main__":
    main()
