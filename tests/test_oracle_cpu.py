"""CPU checks of the oracle itself: integer parts against the scalar C restatement
(oracle/bits.c), structural properties of the march / encoding / compositing, and the
hand-derived compositing gradient (the one the HIP backward implements) against autograd."""
import ctypes
import math

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O


def _np_ptr(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def test_morton_roundtrip_and_c_oracle(bits_oracle):
    rng = np.random.RandomState(0)
    coords = rng.randint(0, 1024, size=(5000, 3)).astype(np.int32)
    coords[:4] = [[0, 0, 0], [1, 0, 0], [0, 1, 0], [1023, 1023, 1023]]
    idx = O.morton3d(torch.from_numpy(coords).long())
    assert idx[:3].tolist() == [0, 1, 2]
    out = np.zeros(len(coords), dtype=np.uint32)
    bits_oracle.oracle_morton3d(_np_ptr(coords, ctypes.c_int32), _np_ptr(out, ctypes.c_uint32),
                                ctypes.c_size_t(len(coords)))
    assert np.array_equal(out.astype(np.int64), idx.numpy())
    back = O.morton3d_invert(idx)
    assert torch.equal(back, torch.from_numpy(coords).long())
    inv = np.zeros_like(coords)
    bits_oracle.oracle_morton3d_invert(_np_ptr(out, ctypes.c_uint32), _np_ptr(inv, ctypes.c_int32),
                                       ctypes.c_size_t(len(coords)))
    assert np.array_equal(inv, coords)


def test_packbits_c_oracle(bits_oracle):
    g = torch.rand(4096) * 2
    bits = O.packbits(g, 1.0)
    out = np.zeros(512, dtype=np.uint8)
    arr = g.numpy().copy()
    bits_oracle.oracle_packbits(_np_ptr(arr, ctypes.c_float), ctypes.c_float(1.0), _np_ptr(out, ctypes.c_uint8),
                                ctypes.c_size_t(4096))
    assert np.array_equal(out, bits.numpy())
    assert int(bits[0]) == sum((1 << k) for k in range(8) if g[k] > 1.0)


def test_level_table_matches_survey():
    lv = O.make_grid_levels()
    assert lv.n_rows == 6119864
    assert lv.resolutions[0] == 16 and lv.resolutions[-1] == 2048
    assert [lv.offsets[i + 1] - lv.offsets[i] for i in range(5)] == [4920, 13824, 32768, 85184, 216000]
    assert all(lv.offsets[i + 1] - lv.offsets[i] == 2 ** 19 for i in range(5, 16))


def test_grid_index_c_oracle(bits_oracle):
    lv = O.make_grid_levels()
    rng = np.random.RandomState(1)
    bits_oracle.oracle_grid_index.restype = ctypes.c_uint32
    for l in (0, 3, 4, 5, 9, 15):
        res = lv.resolutions[l]
        hs = lv.offsets[l + 1] - lv.offsets[l]
        verts = rng.randint(0, res + 1, size=(2000, 3)).astype(np.int32)
        got = O.grid_corner_indices(torch.from_numpy(verts).long(), res, hs).numpy()
        out = np.zeros(len(verts), dtype=np.uint32)
        bits_oracle.oracle_grid_indices(_np_ptr(verts, ctypes.c_int32), _np_ptr(out, ctypes.c_uint32),
                                        ctypes.c_size_t(len(verts)), ctypes.c_uint32(res), ctypes.c_uint32(hs))
        assert np.array_equal(out.astype(np.int64), got)
        assert got.max() < hs
        if l < 5:  # dense levels are collision-free
            assert len(np.unique(got)) == len(np.unique(verts, axis=0))


def test_grid_encode_partition_of_unity_and_vertex_values():
    lv = O.make_grid_levels(num_levels=4, base_resolution=4, desired_resolution=32, log2_hashmap_size=10)
    x = torch.rand(500, 3)
    const = torch.full((lv.n_rows, 2), 0.75)
    f = O.grid_encode(x, const, lv)
    assert torch.allclose(f, torch.full_like(f, 0.75), atol=1e-6)
    # a sample exactly on a vertex of a dense level reads that vertex
    table = torch.randn(lv.n_rows, 2)
    l = 0
    res, scale = lv.resolutions[l], lv.scales[l]
    v = torch.tensor([[1, 2, 3]])
    xv = (v.float() - 0.5) / scale
    f = O.grid_encode(xv, table, lv)
    row = O.grid_corner_indices(v, res, lv.offsets[1] - lv.offsets[0]) + lv.offsets[0]
    assert torch.allclose(f[0, :2], table[row[0]], atol=1e-5)


def _small_scene(G=32, N=256, seed=0):
    torch.manual_seed(seed)
    grid = O.sphere_density_grid(G=G, radius=0.5)
    bits = O.packbits(grid.reshape(-1), 0.01)
    H = W = int(math.isqrt(N))
    f = H / (2 * math.tan(math.radians(55) / 2))
    c2w = O.pose_from_angles(math.radians(60), 0.3, 1.25)
    ro, rd = O.get_rays(c2w, f, f, W / 2, H / 2, H, W)
    return grid, bits, ro[0], rd[0]


@pytest.mark.parametrize("dt_gamma", [0.0, 1.0 / 64])
def test_march_properties(dt_gamma):
    G = 32
    grid, bits, ro, rd = _small_scene(G)
    nears, fars = O.near_far_from_aabb(ro, rd, [-1, -1, -1, 1, 1, 1], 0.1)
    max_steps = 256
    xyzs, dirs, deltas, rays, M = O.march_rays_train(ro, rd, nears, fars, bits, 1.0, 1, G, max_steps, dt_gamma,
                                                     torch.rand(ro.shape[0]))
    assert M == xyzs.shape[0] == int(rays[:, 2].sum()) and M > 0
    assert int(rays[:, 2].max()) <= max_steps
    # every sample sits in an occupied cell, inside the sphere's cell hull
    idx = O.march_cell_index(xyzs, deltas[:, 0], 1.0, 1, G)
    assert bool(((bits.long()[idx >> 3] >> (idx & 7)) & 1).all())
    # offsets are the exclusive scan of counts; t increases along each ray; x = o + t d
    cnt = rays[:, 2].long()
    assert torch.equal(rays[:, 1].long(), torch.cumsum(cnt, 0) - cnt)
    for n in torch.nonzero(cnt > 1)[:20, 0].tolist():
        o, c = int(rays[n, 1]), int(rays[n, 2])
        t = deltas[o:o + c, 1]
        assert bool((t[1:] > t[:-1]).all())
        assert torch.allclose(xyzs[o:o + c], ro[n] + t[:, None] * rd[n], atol=1e-6)
        assert bool((t >= nears[n]).all()) and bool((t < fars[n]).all())
    # rays that miss the box have no samples
    assert int(cnt[nears >= fars].sum()) == 0


def test_march_max_steps_cap():
    # bound 2 (two cascades), fully occupied: the lattice has up to 2x max_steps points per ray,
    # so the per-ray sample cap binds
    G = 32
    full = torch.ones(2, G ** 3)
    bits = O.packbits(full.reshape(-1), 0.5)
    _, _, ro, rd = _small_scene(G, N=64)
    nears, fars = O.near_far_from_aabb(ro, rd, [-2, -2, -2, 2, 2, 2], 0.1)
    xyzs, dirs, deltas, rays, M = O.march_rays_train(ro, rd, nears, fars, bits, 2.0, 2, G, 64, 0.0, None)
    assert int(rays[:, 2].max()) == 64 and int(rays[:, 2].min()) > 0
    assert float(xyzs.abs().max()) <= 2.0


def test_composite_matches_sequential_definition_and_manual_gradient():
    torch.manual_seed(3)
    N, C = 7, 4
    cnts = torch.tensor([0, 5, 1, 70, 3, 0, 130])
    offs = torch.cumsum(cnts, 0) - cnts
    M = int(cnts.sum())
    rays = torch.stack([torch.tensor([3, 0, 6, 1, 5, 2, 4]), offs, cnts], -1).int()
    sig = (torch.rand(M) * 30).double().float().requires_grad_()
    rgb = torch.randn(M, C).requires_grad_()
    dl = torch.stack([torch.full((M,), 0.01), torch.rand(M) + 0.2], -1)
    bg = torch.rand(N, C)
    ws, dp, img = O.composite_rays_train(sig, rgb, dl, rays, 1e-4, bg)
    # sequential reference (the loop a per-ray thread would run)
    for r in range(N):
        rid, o, c = [int(v) for v in rays[r]]
        T, w_sum, d_sum, acc = 1.0, 0.0, 0.0, torch.zeros(C)
        for i in range(o, o + c):
            if T < 1e-4:
                break
            a = 1 - math.exp(-float(sig[i]) * float(dl[i, 0]))
            w = a * T
            w_sum += w; d_sum += w * float(dl[i, 1]); acc += w * rgb[i].detach()
            T *= 1 - a
        assert abs(float(ws[rid]) - w_sum) < 1e-5
        assert abs(float(dp[rid]) - d_sum) < 1e-5
        assert torch.allclose(img[rid], acc + (1 - w_sum) * bg[rid], atol=1e-5)
    # manual gradient (formula implemented by csrc/composite.hip) vs autograd
    g_ws, g_dp, g_img = torch.randn(N), torch.randn(N), torch.randn(N, C)
    (ws * g_ws).sum().add((dp * g_dp).sum()).add((img * g_img).sum()).backward()
    dsig = torch.zeros(M); drgb = torch.zeros(M, C)
    for r in range(N):
        rid, o, c = [int(v) for v in rays[r]]
        if c == 0:
            continue
        s, dt, t = sig[o:o + c].detach(), dl[o:o + c, 0], dl[o:o + c, 1]
        tau = s * dt
        excl = torch.cumsum(tau, 0) - tau
        T = torch.exp(-excl)
        keep = T >= 1e-4
        w = torch.where(keep, (1 - torch.exp(-tau)) * T, torch.zeros(1))
        g = g_ws[rid] + g_dp[rid] * t + ((rgb[o:o + c].detach() - bg[rid]) * g_img[rid]).sum(-1)
        total = (g * w).sum()
        pinc = torch.cumsum(g * w, 0)
        dtau = torch.where(keep, g * T * torch.exp(-tau) - (total - pinc), torch.zeros(1))
        dsig[o:o + c] = dt * dtau
        drgb[o:o + c] = w[:, None] * g_img[rid]
    assert torch.allclose(dsig, sig.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(drgb, rgb.grad, rtol=1e-5, atol=1e-7)


def test_render_frame_small_and_grad_flow():
    G = 32
    grid, bits, ro, rd = _small_scene(G, N=64)
    lv = O.make_grid_levels(num_levels=16, base_resolution=4, desired_resolution=64, log2_hashmap_size=10)
    table = (torch.randn(lv.n_rows, 2) * 0.1).requires_grad_()
    mp = {k: v.requires_grad_() for k, v in O.init_mlp_params().items()}
    out = O.render_frame(ro, rd, table, mp, lv, bits, G=G, max_steps=128, bg_color=torch.rand(64, 4))
    assert out["image"].shape == (64, 4) and out["M"] > 0
    out["image"].backward(torch.randn(64, 4))
    assert table.grad.abs().sum() > 0 and all(v.grad is not None for v in mp.values())
    # rays that miss keep exactly the background
    miss = out["rays"][:, 2] == 0
    assert bool((out["weights_sum"][miss] == 0).all())


def test_trunc_exp_and_adam():
    x = torch.tensor([0.0, 10.0, 20.0], requires_grad=True)
    y = O.trunc_exp(x)
    y.sum().backward()
    assert torch.allclose(x.grad, torch.exp(torch.tensor([0.0, 10.0, 15.0])))
    p = torch.randn(100); g = torch.randn(100)
    ref = p.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    m = torch.zeros(100); v = torch.zeros(100); q = p.clone()
    for step in range(1, 4):
        ref.grad = g.clone()
        opt.step()
        q, m, v = O.adam_step(q, g, m, v, step, 1e-2)
    assert torch.allclose(q, ref.detach(), atol=1e-6)
