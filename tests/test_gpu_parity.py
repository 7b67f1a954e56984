"""GPU parity tests: every HIP kernel (through the C ABI via the host package) against the CPU
oracle on the same seeded inputs.  Run with `pytest -m gpu` on an MI355X.

Tolerances (fp32 path, stated per SURVEY.md §7):
  * discrete outputs (Morton, bitfield, ray spans, which lattice points are samples): bit-exact
  * forward floats: rtol 1e-4, atol 1e-5    * gradients: rtol 1e-3, atol 1e-5 (float atomics
    reorder sums); weight gradients that sum 1e5 terms: relative-to-max 1e-4.
"""
import math

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(built_lib):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    from src.latent_nerf.raymarching import backend as B
    B.get_lib()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def _scene(G=32, HW=16, theta=60.0, phi=20.0, radius=1.25, seed=0, bound=1.0, cascade=1, sphere=0.5):
    torch.manual_seed(seed)
    grid = O.density_grid_from_function(lambda x: (x.norm(dim=-1) < sphere).float() * 10.0, G, cascade, bound)
    bits = O.packbits(grid.reshape(-1), 0.01)
    f = HW / (2 * math.tan(math.radians(55) / 2))
    c2w = O.pose_from_angles(math.radians(theta), math.radians(phi), radius)
    ro, rd = O.get_rays(c2w, f, f, HW / 2, HW / 2, HW, HW)
    return grid, bits, c2w, (f, f, HW / 2, HW / 2), ro[0].contiguous(), rd[0].contiguous()


def _close(a, b, rtol=1e-4, atol=1e-5, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), "%s: %d/%d outside tol, max abs err %.3e (ref max %.3e)" % (
        what, int(bad.sum()), bad.numel(), float(err.max()), float(b.abs().max()))


def _close_rel_max(a, b, rel=1e-4, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = float(b.abs().max()) + 1e-30
    err = float((a - b).abs().max())
    assert err <= rel * scale, "%s: max abs err %.3e vs scale %.3e" % (what, err, scale)


# ------------------------------------------------------------------------------ H1 / H2 / H3
def test_get_rays_and_near_far(dev):
    from src.latent_nerf.raymarching import raymarching as rm
    _, _, c2w, intr, ro, rd = _scene(HW=24)
    poses = torch.stack([c2w, O.pose_from_angles(0.3, 4.0, 1.1), O.pose_from_angles(2.5, 1.0, 1.5)]).to(dev)
    go, gd = rm.get_rays(poses, intr, 24, 24)
    ro_all, rd_all = O.get_rays(poses.cpu(), *intr, 24, 24)
    _close(go, ro_all, 0, 1e-7, "rays_o")
    _close(gd, rd_all, 1e-6, 1e-7, "rays_d")
    _close(gd.norm(dim=-1), torch.ones(3, 576), 1e-6, 1e-6, "unit dirs")
    # near/far incl. rays that miss the box and axis-parallel rays
    o = torch.cat([ro_all.reshape(-1, 3), torch.tensor([[0.0, 0.0, -3.0], [5.0, 5.0, 5.0], [0.2, 0.1, 3.0]])])
    d = torch.cat([rd_all.reshape(-1, 3), torch.tensor([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])])
    for aabb, mn in (([-1, -1, -1, 1, 1, 1], 0.1), ([-0.5, -0.25, -0.5, 0.5, 0.75, 0.5], 0.05)):
        n_ref, f_ref = O.near_far_from_aabb(o, d, aabb, mn)
        n, f = rm.near_far_from_aabb(o.to(dev), d.to(dev), aabb, mn)
        _close(n, n_ref, 1e-6, 1e-6, "nears")
        _close(f, f_ref, 1e-6, 1e-6, "fars")
        assert bool((n_ref[-2] == O.FLT_MAX).item()) and float(n[-2]) == float(n_ref[-2])


def test_morton_and_packbits_bit_exact(dev, bits_oracle):
    from src.latent_nerf.raymarching import raymarching as rm
    rng = np.random.RandomState(0)
    coords = torch.from_numpy(rng.randint(0, 1024, size=(100003, 3)).astype(np.int32))
    idx = rm.morton3D(coords.to(dev))
    assert torch.equal(idx.cpu().long(), O.morton3d(coords.long()))
    assert torch.equal(rm.morton3D_invert(idx).cpu(), coords)
    grid = torch.rand(2, 64 ** 3) * 2
    for thresh, mean in ((1.0, None), (1.5, 0.7)):
        md = None if mean is None else torch.tensor([mean], device=dev)
        bits = rm.packbits(grid.to(dev), thresh, None, md)
        ref = O.packbits(grid.reshape(-1), thresh if mean is None else min(thresh, mean))
        assert torch.equal(bits.cpu(), ref)
    assert rm.morton3D(coords[:0].to(dev)).numel() == 0  # empty input


# ------------------------------------------------------------------------------ H4
def _march_both(dev, ro, rd, bits, bound, cascade, G, max_steps, dt_gamma, noises, capacity=None):
    from src.latent_nerf.raymarching import raymarching as rm
    aabb = [-bound] * 3 + [bound] * 3
    nears, fars = O.near_far_from_aabb(ro, rd, aabb, 0.1)
    ref = O.march_rays_train(ro, rd, nears, fars, bits, bound, cascade, G, max_steps, dt_gamma, noises)
    res = rm.march_rays_train(ro.to(dev), rd.to(dev), bound, bits.to(dev), cascade, G, nears.to(dev), fars.to(dev),
                              dt_gamma=dt_gamma, max_steps=max_steps, capacity=capacity,
                              noises=None if noises is None else noises.to(dev))
    # the form that clips against the box inside the march passes (lnerf_march_rays_train_aabb) is the same march
    res2 = rm.march_rays_train(ro.to(dev), rd.to(dev), bound, bits.to(dev), cascade, G, None, None,
                               dt_gamma=dt_gamma, max_steps=max_steps, capacity=capacity,
                               noises=None if noises is None else noises.to(dev), aabb=aabb, min_near=0.1)
    M = int(res.counter[0])
    assert torch.equal(res2.counter.cpu(), res.counter.cpu()) and torch.equal(res2.rays.cpu(), res.rays.cpu())
    assert torch.equal(res2.xyzs[:M].cpu(), res.xyzs[:M].cpu()) and torch.equal(res2.deltas[:M].cpu(), res.deltas[:M].cpu())
    return ref, res


@pytest.mark.parametrize("dt_gamma,perturb", [(0.0, False), (0.0, True), (1.0 / 128, True)])
def test_march_rays_train_bit_exact(dev, dt_gamma, perturb):
    G = 64
    _, bits, _, _, ro, rd = _scene(G=G, HW=32)
    noises = torch.rand(ro.shape[0]) if perturb else None
    (xyzs, dirs, deltas, rays, M), res = _march_both(dev, ro, rd, bits, 1.0, 1, G, 512, dt_gamma, noises)
    cnt = res.counter.cpu()
    assert int(cnt[0]) == M and M > 1000
    assert int(cnt[1]) == int((rays[:, 2] > 0).sum()) and int(cnt[2]) == 0
    assert int(cnt[3]) == M                      # running peak of M (fresh buffers: this march's own)
    assert res.take_peak() == (M, False) and int(res.counter[3]) == 0
    assert torch.equal(res.rays.cpu(), rays)
    assert torch.equal(res.xyzs[:M].cpu(), xyzs)
    assert torch.equal(res.deltas[:M].cpu(), deltas)
    assert torch.equal(res.dirs[:M].cpu(), dirs)


def test_march_counter_based_jitter_matches_oracle_hash(dev):
    """The in-kernel generator (noise_counter form): call k draws u_n = hash(n, seed, k) -- the same samples, bit
    for bit, as the table form fed with the oracle's restatement of the hash; the device counter advances by one
    per call (count and write pass of one call see the same jitter)."""
    from src.latent_nerf.raymarching import raymarching as rm
    G = 64
    _, bits, _, _, ro, rd = _scene(G=G, HW=32)
    N = ro.shape[0]
    seed = 0x5EED
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    aabb = [-1.0] * 3 + [1.0] * 3
    nears, fars = O.near_far_from_aabb(ro, rd, aabb, 0.1)
    args = (ro.to(dev), rd.to(dev), 1.0, bits.to(dev), 1, G, nears.to(dev), fars.to(dev))
    seen = []
    for k in range(3):
        res = rm.march_rays_train(*args, perturb=True, max_steps=512, noise_state=(seed, counter))
        assert int(counter[0]) == k + 1
        u = O.march_noise(N, seed, k)
        assert 0.0 <= float(u.min()) and float(u.max()) < 1.0
        (xyzs, dirs, deltas, rays, M), _ = _march_both(dev, ro, rd, bits, 1.0, 1, G, 512, 0.0, u)
        assert int(res.counter[0]) == M and torch.equal(res.rays.cpu(), rays)
        assert torch.equal(res.xyzs[:M].cpu(), xyzs) and torch.equal(res.deltas[:M].cpu(), deltas)
        seen.append(res.xyzs[:64].cpu().clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])   # fresh jitter every call


def test_march_cascades_cap_and_capacity(dev):
    G = 32
    # two cascades, fully occupied: per-ray cap binds (oracle test_march_max_steps_cap)
    full = torch.ones(2, G ** 3)
    bits = O.packbits(full.reshape(-1), 0.5)
    _, _, _, _, ro, rd = _scene(G=G, HW=12)
    (xyzs, dirs, deltas, rays, M), res = _march_both(dev, ro, rd, bits, 2.0, 2, G, 64, 0.0, None)
    assert int(rays[:, 2].max()) == 64
    assert torch.equal(res.rays.cpu(), rays) and int(res.counter[0]) == M
    assert torch.equal(res.xyzs[:M].cpu(), xyzs) and torch.equal(res.deltas[:M].cpu(), deltas)
    # sphere in a 2-cascade grid (mip level selection)
    _, bits2, _, _, ro, rd = _scene(G=G, HW=16, bound=2.0, cascade=2, sphere=0.8, radius=1.6)
    (xyzs, dirs, deltas, rays, M), res = _march_both(dev, ro, rd, bits2, 2.0, 2, G, 256, 0.0, torch.rand(ro.shape[0]))
    assert M > 0 and torch.equal(res.rays.cpu(), rays) and torch.equal(res.xyzs[:M].cpu(), xyzs)
    # capacity overflow: later rays are dropped, earlier spans stay intact, nothing is written past capacity
    cap = M // 2
    _, res = _march_both(dev, ro, rd, bits2, 2.0, 2, G, 256, 0.0, None, capacity=cap)
    r = res.rays.cpu()
    c = res.counter.cpu()
    assert int(c[2]) > 0 and int(c[0]) <= cap
    assert int(c[3]) == (int(c[0]) | (1 << 30))  # peak word: dropped rays raise bit 30
    kept = r[:, 2] > 0
    assert int((r[kept, 1] + r[kept, 2]).max()) == int(c[0])
    # no occupied cells / all rays missing: zero samples
    empty = torch.zeros(G ** 3 // 8, dtype=torch.uint8)
    (_, _, _, rays0, M0), res0 = _march_both(dev, ro, rd, empty, 1.0, 1, G, 256, 0.0, None)
    assert M0 == 0 and int(res0.counter[0]) == 0 and int(res0.rays[:, 2].sum()) == 0


def test_march_above_8192_rays_takes_the_scan_kernel(dev):
    """Up to 8192 rays the write pass sums the counts before its ray itself (two launches); above, count / scan / write.
    Both forms against the oracle, jitter table and capacity overflow included."""
    G = 32
    _, bits, _, _, ro, rd = _scene(G=G, HW=96)            # 9216 rays
    assert ro.shape[0] > 8192
    noises = torch.rand(ro.shape[0])
    (xyzs, dirs, deltas, rays, M), res = _march_both(dev, ro, rd, bits, 1.0, 1, G, 128, 0.0, noises)
    assert res.counter.numel() == 4                        # (no per-ray scratch in this form)
    assert int(res.counter[0]) == M and M > 10000 and torch.equal(res.rays.cpu(), rays)
    assert torch.equal(res.xyzs[:M].cpu(), xyzs) and torch.equal(res.deltas[:M].cpu(), deltas)
    cap = M // 3
    _, res_c = _march_both(dev, ro, rd, bits, 1.0, 1, G, 128, 0.0, noises, capacity=cap)
    # the same rays in two halves of 4608 (two-launch form): the first half's spans and drops must agree
    h = ro.shape[0] // 2
    _, res_h = _march_both(dev, ro[:h], rd[:h], bits, 1.0, 1, G, 128, 0.0, noises[:h], capacity=cap)
    assert res_h.counter.numel() == 4 + h + (h + 3) // 4
    assert torch.equal(res_h.rays.cpu(), res_c.rays[:h].cpu())
    assert int(res_c.counter[2]) >= int(res_h.counter[2]) > 0 and int(res_c.counter[0]) == int(res_h.counter[0]) <= cap


# ------------------------------------------------------------------------------ H5 / H6
def _rand_points(M, bound=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(M, 3, generator=g) * 2 - 1) * bound
    x[0] = torch.tensor([-bound, -bound, -bound])
    x[1] = torch.tensor([bound, bound, bound])  # the far corner: pos_grid + 1 == resolution
    x[2] = torch.tensor([0.0, 0.0, 0.0])
    return x


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("cfg", ["small", "full"])
def test_grid_encode_forward_backward(dev, variant, cfg):
    from src.latent_nerf.models import encoding as E
    from src.latent_nerf.raymarching import backend as B
    if cfg == "small":
        kw = dict(num_levels=16, base_resolution=4, desired_resolution=128, log2_hashmap_size=12)
        M = 3001
    else:
        kw = dict(num_levels=16, base_resolution=16, desired_resolution=2048, log2_hashmap_size=19)
        M = 20000
    lv = O.make_grid_levels(**kw)
    levels = E.GridLevels(kw["num_levels"], 2, kw["base_resolution"], kw["desired_resolution"],
                          kw["log2_hashmap_size"])
    assert levels.offsets == lv.offsets and levels.resolutions == lv.resolutions
    assert np.allclose(levels.scales, lv.scales, rtol=0, atol=0)
    torch.manual_seed(1)
    table = torch.randn(lv.n_rows, 2) * 0.1
    x = _rand_points(M)
    tref = table.clone().requires_grad_()
    ref = O.grid_encode((x + 1.0) / 2.0, tref, lv)                      # [M, 32]
    stride = M + 37                                                      # level_stride > M on purpose
    m_dev = torch.tensor([M], dtype=torch.int32, device=dev)
    # (gather mapping 1 -- levels pinned to XCDs -- exists in experiment builds only)
    gv = min(variant, 1) if "experiments" in B.get_lib().lnerf_build_info().decode() else 0
    feat = E.grid_encode_forward(x.to(dev), 1.0, table.to(dev), levels, stride, m_dev, stride, variant=gv)
    got = feat[:, :M, :].permute(1, 0, 2).reshape(M, 32)
    _close(got, ref, 1e-4, 1e-6, "features")
    # backward (scatter-add) vs autograd of the oracle
    g = torch.randn(M, 32)
    ref.backward(g)
    dfeat = torch.zeros(kw["num_levels"], stride, 2)
    dfeat[:, :M, :] = g.reshape(M, kw["num_levels"], 2).permute(1, 0, 2)
    dtable = torch.zeros(lv.n_rows, 2, device=dev)
    E.grid_encode_backward(x.to(dev), 1.0, dfeat.to(dev), levels, stride, m_dev, stride, dtable, variant=variant)
    _close(dtable, tref.grad, 1e-3, 1e-5, "dtable")
    # accumulate semantics (+=)
    E.grid_encode_backward(x.to(dev), 1.0, dfeat.to(dev), levels, stride, m_dev, stride, dtable, variant=variant)
    _close(dtable, 2 * tref.grad, 1e-3, 2e-5, "dtable accumulates")


@pytest.mark.parametrize("gridtype", ["blocked", "tiled"])
@pytest.mark.parametrize("table_dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", ["small", "full"])
def test_blocked_and_tiled_layouts_forward_backward(dev, cfg, table_dtype, gridtype):
    """gridtype = "blocked" (LNERF_GRID_BLOCKED: hashed levels keep 4 x 2 x 2 vertex blocks in 16 consecutive rows) and
    gridtype = "tiled" (LNERF_GRID_TILED: the upstream encoder's other layout, the dense index wrapped into the table --
    SURVEY.md Appendix A suggests {hash, tiled} x {16, 19}): the gather with every load path (single / pair / aligned
    quad), the atomic scatter and the bucketed scatter (12- and 8-byte records) against the oracle's restatement of the
    same layout; dense levels unchanged."""
    from src.latent_nerf.models import encoding as E
    if cfg == "small":
        kw = dict(num_levels=16, base_resolution=4, desired_resolution=128, log2_hashmap_size=12)
        M = 3001
    else:
        kw = dict(num_levels=16, base_resolution=16, desired_resolution=2048, log2_hashmap_size=19)
        M = 20000
    lv = O.make_grid_levels(blocked=gridtype == "blocked", tiled=gridtype == "tiled", **kw)
    levels = E.GridLevels(kw["num_levels"], 2, kw["base_resolution"], kw["desired_resolution"], kw["log2_hashmap_size"],
                          gridtype=gridtype)
    torch.manual_seed(5)
    table = torch.randn(lv.n_rows, 2) * 0.1
    if table_dtype == "bf16":
        table = table.to(torch.bfloat16).float()
    x = _rand_points(M)
    tref = table.clone().requires_grad_()
    ref = O.grid_encode((x + 1.0) / 2.0, tref, lv)
    m_dev = torch.tensor([M], dtype=torch.int32, device=dev)
    src = table.to(dev).to(torch.bfloat16) if table_dtype == "bf16" else table.to(dev)
    feat = E.grid_encode_forward(x.to(dev), 1.0, src, levels, M, m_dev, M)
    _close(feat.permute(1, 0, 2).reshape(M, 32), ref, 1e-4, 1e-6, gridtype + " features")
    plain = E.grid_encode_forward(x.to(dev), 1.0, src, E.GridLevels(kw["num_levels"], 2, kw["base_resolution"],
                                                                     kw["desired_resolution"], kw["log2_hashmap_size"]), M, m_dev, M)
    dense = [l for l in range(16) if (lv.resolutions[l] + 1) ** 3 <= lv.offsets[l + 1] - lv.offsets[l]]
    assert dense and torch.equal(feat[dense], plain[dense]) and not torch.equal(feat, plain)
    if table_dtype == "bf16":
        return
    g = torch.randn(M, 32)
    ref.backward(g)
    dfeat = g.reshape(M, 16, 2).permute(1, 0, 2).contiguous().to(dev)
    for variant, atol in ((0, 1e-5), (2, 1e-5), (3, 5e-5)):
        dtable = torch.zeros(lv.n_rows, 2, device=dev)
        E.grid_encode_backward(x.to(dev), 1.0, dfeat, levels, M, m_dev, M, dtable, variant=variant)
        _close(dtable, tref.grad, 1e-3, atol, gridtype + " dtable v%d" % variant)


@pytest.mark.parametrize("variant", [2, 3])
def test_bucketed_scatter_is_bitwise_reproducible(dev, variant):
    """Fixed-point accumulation: the same scatter run twice (different workgroup timing, and a different number of
    slices per coarse bucket when the capacity differs) gives the same bits."""
    from src.latent_nerf.models import encoding as E
    levels = E.GridLevels()
    M = 200000
    g = torch.Generator().manual_seed(9)
    x = ((torch.rand(M, 3, generator=g) * 2 - 1) * 0.999).to(dev)
    dfeat = torch.randn(16, M, 2, generator=g).to(dev)
    outs = []
    for stride in (M, M, M + 4096 * 5):      # a larger capacity -> other slice counts / bucket regions
        d = torch.zeros(levels.n_rows, 2, device=dev)
        df = torch.zeros(16, stride, 2, device=dev)
        df[:, :M] = dfeat
        xx = torch.zeros(stride, 3, device=dev)
        xx[:M] = x
        m_dev = torch.tensor([M], dtype=torch.int32, device=dev)
        E.grid_encode_backward(xx, 1.0, df, levels, stride, m_dev, stride, d, variant=variant)
        outs.append(d)
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("variant", [2, 3])
@pytest.mark.parametrize("step", [0.0009, 0.0034, 0.02])
def test_bucketed_scatter_merges_runs_along_rays(dev, variant, step):
    """Samples ordered along rays (what the march produces): consecutive lanes of a wavefront sit in the same cell for
    runs of 1 ... 64 lanes depending on level and step, the runs cross the 16- and 32-lane rows of the segmented scan
    at arbitrary places, and stretches of exactly-zero gradients (samples behind a ray's termination) fall inside and
    across runs.  The merged sums must equal the oracle's autograd."""
    from src.latent_nerf.models import encoding as E
    levels = E.GridLevels()
    lv = O.make_grid_levels()
    g = torch.Generator().manual_seed(21)
    n_rays, per_ray = 300, 97                       # 97: rays do not start on wavefront boundaries
    o = (torch.rand(n_rays, 1, 3, generator=g) * 2 - 1) * 0.6
    d = torch.nn.functional.normalize(torch.randn(n_rays, 1, 3, generator=g), dim=-1)
    d[::7] = torch.tensor([1.0, 0.0, 0.0])          # axis-parallel rays: the longest runs
    t = torch.arange(per_ray).view(1, per_ray, 1) * step
    x = (o + d * t).clamp(-0.999, 0.999).reshape(-1, 3)
    M = x.shape[0]
    grad = torch.randn(M, 32, generator=g)
    dead = (torch.arange(M) % per_ray) >= torch.randint(20, per_ray + 1, (n_rays,), generator=g).repeat_interleave(per_ray)
    grad[dead] = 0.0                                 # terminated tails: exact zeros
    tref = torch.zeros(lv.n_rows, 2, requires_grad=True)
    O.grid_encode((x + 1) / 2, tref, lv).backward(grad)
    dfeat = grad.reshape(M, 16, 2).permute(1, 0, 2).contiguous().to(dev)
    dtable = torch.zeros(lv.n_rows, 2, device=dev)
    E.grid_encode_backward(x.to(dev), 1.0, dfeat, levels, M, None, M, dtable, variant=variant)
    _close(dtable, tref.grad, 1e-3, 2e-4 if variant == 3 else 5e-5, "run-merged dtable")
    if variant == 2:  # (the 8-byte records quantise to 2^-30 of the level's bound: a 1e-7-weight corner may vanish)
        nz_ref = (tref.grad.abs().sum(-1) > 0)
        assert torch.equal((dtable.abs().sum(-1) > 0).cpu() | ~nz_ref, torch.ones_like(nz_ref))  # no row lost


@pytest.mark.parametrize("variant", [2, 3])
def test_bucketed_scatter_with_every_record_in_a_few_buckets(dev, variant):
    """All samples inside one fine cell: every record of a hashed level lands in <= 8 buckets (whole items are one
    segment).  The item-chunk layout has no per-bucket capacity, so this is the ordinary path -- complete sums, and
    bitwise reproducible like every other input (round 2's layout took a float-atomic fallback here)."""
    from src.latent_nerf.models import encoding as E
    levels = E.GridLevels()
    lv = O.make_grid_levels()
    M = 6000
    torch.manual_seed(11)
    x = torch.tensor([[0.1234, -0.3456, 0.4567]]) + torch.rand(M, 3) * 1e-5
    g = torch.randn(M, 32)
    tref = torch.zeros(lv.n_rows, 2, requires_grad=True)
    O.grid_encode((x + 1) / 2, tref, lv).backward(g)
    dfeat = g.reshape(M, 16, 2).permute(1, 0, 2).contiguous().to(dev)
    dtable = torch.zeros(lv.n_rows, 2, device=dev)
    E.grid_encode_backward(x.to(dev), 1.0, dfeat, levels, M, None, M, dtable, variant=variant)
    _close(dtable, tref.grad, 1e-3, 2e-3, "clustered dtable")  # sums of 6000 terms of O(1)
    nz_ref = (tref.grad.abs().sum(-1) > 0)
    assert torch.equal((dtable.abs().sum(-1) > 0).cpu() | ~nz_ref, torch.ones_like(nz_ref))  # no row lost
    again = torch.zeros(lv.n_rows, 2, device=dev)
    E.grid_encode_backward(x.to(dev), 1.0, dfeat, levels, M, None, M, again, variant=variant)
    assert torch.equal(again, dtable)


def test_grid_encode_bf16_and_properties_full_size(dev):
    from src.latent_nerf.models import encoding as E
    levels = E.GridLevels()
    lv = O.make_grid_levels()
    M = 1 << 19
    x = _rand_points(M, seed=5).to(dev)
    # partition of unity at BASELINE size: a constant table gives constant features on every level
    const = torch.full((levels.n_rows, 2), 0.625, device=dev)
    f = E.grid_encode_forward(x, 1.0, const, levels, M, None, M)
    assert float((f - 0.625).abs().max()) < 1e-6
    # bf16 shadow table / bf16 features vs the oracle evaluated on the bf16-rounded table
    torch.manual_seed(2)
    table = torch.randn(levels.n_rows, 2) * 0.1
    tb = table.to(torch.bfloat16)
    sub = 4096
    ref = O.grid_encode((x[:sub].cpu() + 1) / 2, tb.float(), lv)
    f32o = E.grid_encode_forward(x[:sub].contiguous(), 1.0, tb.to(dev), levels, sub, None, sub)
    _close(f32o.permute(1, 0, 2).reshape(sub, 32), ref, 1e-4, 1e-6, "bf16 table -> f32 feat")
    fb = E.grid_encode_forward(x[:sub].contiguous(), 1.0, tb.to(dev), levels, sub, None, sub,
                               out_dtype=torch.bfloat16)
    _close(fb.float().permute(1, 0, 2).reshape(sub, 32), ref.to(torch.bfloat16).float(), 1e-2, 1e-4, "bf16 feat")
    # linearity of the scatter: backward(a*g1 + g2) == a*backward(g1) + backward(g2) (up to atomics order)
    g1 = torch.randn(16, sub, 2, device=dev)
    g2 = torch.randn(16, sub, 2, device=dev)
    outs = []
    for g in (g1, g2, 3.0 * g1 + g2):
        d = torch.zeros(levels.n_rows, 2, device=dev)
        E.grid_encode_backward(x[:sub].contiguous(), 1.0, g.contiguous(), levels, sub, None, sub, d)
        outs.append(d)
    _close(outs[2], 3.0 * outs[0] + outs[1], 1e-3, 1e-4, "scatter linearity")
    # total gradient mass is conserved level by level: sum(dtable[level]) == sum(dfeat[level])
    for l in (0, 7, 15):
        s = outs[0][levels.offsets[l]:levels.offsets[l + 1]].sum(0)
        _close(s, g1[l].sum(0), 1e-3, 1e-2, "mass level %d" % l)


# ------------------------------------------------------------------------------ H7
def _mlp_inputs(M, seed=0):
    torch.manual_seed(seed)
    feat = torch.randn(M, 32) * 0.5
    xyz = (torch.rand(M, 3) * 2 - 1) * 0.8
    p = O.init_mlp_params(seed=seed)
    return feat, xyz, p


@pytest.mark.parametrize("M", [1, 63, 64, 1000, 70001])
def test_mlp_forward_backward_f32(dev, M):
    from src.latent_nerf.models.network_grid import _SigmaLatentMLP
    from src.latent_nerf.raymarching import backend as B
    feat, xyz, p = _mlp_inputs(M, seed=M)
    pr = {k: v.clone().requires_grad_() for k, v in p.items()}
    fr = feat.clone().requires_grad_()
    s_ref, c_ref = O.sigma_latent_mlp(fr, xyz, pr)
    gs, gc = torch.randn(M) * 0.1, torch.randn(M, 4)
    ((s_ref * gs).sum() + (c_ref * gc).sum()).backward()
    stride = M + 5
    lm = torch.zeros(16, stride, 2)
    lm[:, :M, :] = feat.reshape(M, 16, 2).permute(1, 0, 2)
    lm = lm.to(dev).requires_grad_()
    pg = {k: v.to(dev).requires_grad_() for k, v in p.items()}
    xg = torch.zeros(stride, 3)
    xg[:M] = xyz
    m_dev = torch.tensor([M], dtype=torch.int32, device=dev)
    sig, rgb = _SigmaLatentMLP.apply(lm, xg.to(dev), pg["w1"], pg["b1"], pg["w2"], pg["b2"], pg["w3"], pg["b3"],
                                     stride, m_dev, stride, 5.0, 0.2, B.F32, None)
    _close(sig[:M], s_ref, 1e-4, 1e-5, "sigma")
    _close(rgb[:M], c_ref, 1e-4, 1e-5, "latent")
    gsp, gcp = torch.zeros(stride), torch.zeros(stride, 4)
    gsp[:M], gcp[:M] = gs, gc
    torch.autograd.backward([sig, rgb], [gsp.to(dev), gcp.to(dev)])
    dfe = lm.grad[:, :M, :].permute(1, 0, 2).reshape(M, 32)
    _close(dfe, fr.grad, 1e-3, 1e-5, "dfeat")
    for k in ("w1", "b1", "w2", "b2", "w3", "b3"):
        _close_rel_max(pg[k].grad, pr[k].grad, 1e-4, "d" + k)


@pytest.mark.parametrize("M,feat_bf16", [(1, True), (31, True), (129, False), (5000, True), (70001, True)])
def test_mlp_forward_backward_bf16(dev, M, feat_bf16):
    """bf16 MFMA path vs the oracle with bf16-rounded operands (features, weights, hidden activations)
    and f32 accumulation.  Tolerances: forward rtol 2e-2 / atol 2e-3 (an accumulation-order difference
    can flip the bf16 rounding of a hidden unit); gradients 3e-2 of the largest reference entry
    (the kernel additionally rounds dZ to bf16 before each product)."""
    from src.latent_nerf.models.network_grid import _SigmaLatentMLP
    from src.latent_nerf.raymarching import backend as B
    feat, xyz, p = _mlp_inputs(M, seed=M + 1)
    feat = feat.to(torch.bfloat16).float()  # what the bf16 gather would hand over
    pr = {k: v.clone().requires_grad_() for k, v in p.items()}
    fr = feat.clone().requires_grad_()
    s_ref, c_ref = O.sigma_latent_mlp(fr, xyz, pr, bf16=True)
    gs, gc = torch.randn(M) * 0.1, torch.randn(M, 4)
    ((s_ref * gs).sum() + (c_ref * gc).sum()).backward()
    stride = M + 7
    lm = torch.zeros(16, stride, 2)
    lm[:, :M, :] = feat.reshape(M, 16, 2).permute(1, 0, 2)
    lm = lm.to(dev)
    if feat_bf16:
        lm = lm.to(torch.bfloat16)
    lm.requires_grad_()
    pg = {k: v.to(dev).requires_grad_() for k, v in p.items()}
    xg = torch.zeros(stride, 3)
    xg[:M] = xyz
    m_dev = torch.tensor([M], dtype=torch.int32, device=dev)
    sig, rgb = _SigmaLatentMLP.apply(lm, xg.to(dev), pg["w1"], pg["b1"], pg["w2"], pg["b2"], pg["w3"], pg["b3"],
                                     stride, m_dev, stride, 5.0, 0.2, B.BF16, None)
    _close(sig[:M], s_ref, 2e-2, 2e-3, "sigma bf16")
    _close(rgb[:M], c_ref, 2e-2, 2e-3, "latent bf16")
    gsp, gcp = torch.zeros(stride), torch.zeros(stride, 4)
    gsp[:M], gcp[:M] = gs, gc
    torch.autograd.backward([sig, rgb], [gsp.to(dev), gcp.to(dev)])
    dfe = lm.grad.float()[:, :M, :].permute(1, 0, 2).reshape(M, 32)
    _close_rel_max(dfe, fr.grad, 3e-2, "dfeat bf16")
    for k in ("w1", "b1", "w2", "b2", "w3", "b3"):
        _close_rel_max(pg[k].grad, pr[k].grad, 3e-2, "bf16 d" + k)
    # and the bf16 path stays close to the exact f32 network (sanity on the rounding model)
    s32, c32 = O.sigma_latent_mlp(feat, xyz, p)
    _close(rgb[:M], c32, 5e-2, 2e-2, "latent bf16 vs f32")


def test_occ_sample_matches_oracle(dev):
    """lnerf_occ_sample (device-side steady-state sampling of the occupancy refresh): the cell indices and the jittered
    points against the oracle's restatement, bit for bit; with an empty grid every draw is uniform."""
    from src.latent_nerf.raymarching import backend as B
    G = 32
    G3 = G ** 3
    torch.manual_seed(4)
    grid = torch.where(torch.rand(G3) < 0.07, torch.rand(G3) + 0.01, torch.zeros(G3))
    grid[5] = -1.0                                     # invalid cells are not "occupied"
    n_rand = G3 // 4
    nb = B.get_lib().lnerf_occ_sample_scratch_bytes(G3)
    for cas, bound, level in ((0, 1.0, grid), (1, 2.0, grid), (0, 1.0, torch.zeros(G3))):
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        idx = torch.empty(2 * n_rand, dtype=torch.int32, device=dev)
        xyz = torch.empty(2 * n_rand, 3, device=dev)
        lv = level.to(dev)
        B.call("lnerf_occ_sample", lv.data_ptr(), G3, cas, G, bound, n_rand, 0x5EED, 37, scratch.data_ptr(),
               idx.data_ptr(), xyz.data_ptr(), torch.cuda.current_stream().cuda_stream)
        ref_idx, ref_xyz = O.occ_sample(level, cas, G, bound, n_rand, 0x5EED, 37)
        assert torch.equal(idx.cpu().long(), ref_idx)
        assert torch.equal(xyz.cpu(), ref_xyz)
        # stratified: both halves ascending (neighbouring cells on neighbouring lanes of the density query), the first
        # half exactly one cell out of every aligned group of G3 / n_rand
        assert bool((ref_idx[:n_rand][1:] > ref_idx[:n_rand][:-1]).all())
        assert torch.equal(ref_idx[:n_rand] // (G3 // n_rand), torch.arange(n_rand))
        assert bool((ref_idx[n_rand:][1:] >= ref_idx[n_rand:][:-1]).all())
        if float(level.max()) > 0:
            assert bool((level[ref_idx[n_rand:]] > 0).all())       # the second half sits in occupied cells
            occ = int((level > 0).sum())
            assert len(set(ref_idx[n_rand:].tolist())) == min(occ, n_rand)    # fewer occupied cells than draws: ALL of them
            assert 0.2 < float((ref_idx[:n_rand].float() / G3).mean()) < 0.8


def test_mlp_backward_operand_swap_variant_matches_the_default(dev):
    """`mlp_bwd_variant` 1 (operand-swap form: the activations the weight gradients need are recomputed with the two
    MFMA operands swapped, no LDS transposes, no barriers) against the default backward on the same inputs: the data
    chain is the same arithmetic (dfeat bit-identical), the weight gradients differ by summation order only."""
    from src.latent_nerf.models.network_grid import _SigmaLatentMLP
    from src.latent_nerf.raymarching import backend as B
    if "experiments" not in B.get_lib().lnerf_build_info().decode():
        # the product build leaves the rejected variants out: asking for one must fail loudly
        with pytest.raises(B.LnerfError, match="experiment variant"):
            B.call("lnerf_set_tuning", b"mlp_bwd_variant", 1)
        pytest.skip("operand-swap backward: experiment builds only (LNERF_EXPERIMENTS=1 python latent-nerf-test_amd/build.py)")
    M = 70001
    feat, xyz, p = _mlp_inputs(M, seed=5)
    lm0 = feat.reshape(M, 16, 2).permute(1, 0, 2).contiguous().to(dev).to(torch.bfloat16)
    m_dev = torch.tensor([M], dtype=torch.int32, device=dev)
    gs, gc = (torch.randn(M) * 0.1).to(dev), torch.randn(M, 4).to(dev)
    gs[1000:5000] = 0      # a stretch of dead samples (skipped steps)
    gc[1000:5000] = 0
    res = {}
    try:
        for var in (0, 1):
            B.call("lnerf_set_tuning", b"mlp_bwd_variant", var)
            lm = lm0.clone().requires_grad_()
            pg = {k: v.clone().to(dev).requires_grad_() for k, v in p.items()}
            sig, rgb = _SigmaLatentMLP.apply(lm, xyz.to(dev), pg["w1"], pg["b1"], pg["w2"], pg["b2"], pg["w3"], pg["b3"],
                                             M, m_dev, M, 5.0, 0.2, B.BF16, None)
            torch.autograd.backward([sig, rgb], [gs, gc])
            res[var] = (lm.grad.clone(), {k: pg[k].grad.clone() for k in pg})
    finally:
        B.call("lnerf_set_tuning", b"mlp_bwd_variant", 0)
    assert torch.equal(res[0][0], res[1][0])
    assert float(res[0][0][:, 1000:5000].abs().max()) == 0.0
    for k in res[0][1]:
        a, b = res[0][1][k], res[1][1][k]
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-9, k


def test_mlp_trunc_exp_clamp_and_rgb_mode_shapes(dev):
    from src.latent_nerf.models.network_grid import _SigmaLatentMLP
    from src.latent_nerf.raymarching import backend as B
    # huge pre-activation: forward = exp(x), backward uses exp(min(x, 15))
    M = 64
    feat = torch.zeros(16, M, 2, device=dev)
    p = O.init_mlp_params(out_dim=4)  # C = 3 (rgb mode)
    p["b3"] = torch.tensor([18.0, 0.1, 0.2, 0.3])
    pg = {k: v.to(dev).requires_grad_() for k, v in p.items()}
    xyz = torch.full((M, 3), 5.0, device=dev)  # blob ~ 0
    sig, rgb = _SigmaLatentMLP.apply(feat, xyz, pg["w1"], pg["b1"], pg["w2"], pg["b2"], pg["w3"], pg["b3"], M, None, M,
                                     5.0, 0.2, B.F32, None)
    pr = {k: v.clone().requires_grad_() for k, v in p.items()}
    s_ref, c_ref = O.sigma_latent_mlp(torch.zeros(M, 32), xyz.cpu(), pr)
    assert rgb.shape == (M, 3)
    _close(sig, s_ref, 1e-4, 1e-3, "sigma big")
    sig.sum().backward()
    s_ref.sum().backward()
    _close_rel_max(pg["b3"].grad, pr["b3"].grad, 1e-4, "db3 clamp")
    assert abs(float(pg["b3"].grad[0]) / (M * math.exp(15.0)) - 1.0) < 1e-3


# ------------------------------------------------------------------------------ H8 / H9
@pytest.mark.parametrize("C", [3, 4])
def test_composite_forward_backward(dev, C):
    from src.latent_nerf.raymarching import raymarching as rm
    torch.manual_seed(C)
    cnts = torch.tensor([0, 5, 1, 64, 65, 3, 0, 300, 128, 2])
    N = len(cnts)
    offs = torch.cumsum(cnts, 0) - cnts
    M = int(cnts.sum())
    rays = torch.stack([torch.randperm(N), offs, cnts], -1).int()
    sig = torch.rand(M) * 20
    sig[offs[7]:offs[7] + 300] = torch.rand(300) * 400      # forces the early stop inside a long ray
    rgb = torch.randn(M, C)
    dl = torch.stack([torch.full((M,), 3.4e-3), torch.rand(M) + 0.3], -1)
    bg = torch.rand(N, C)
    sr, rr, br = sig.clone().requires_grad_(), rgb.clone().requires_grad_(), bg.clone().requires_grad_()
    ws_ref, dp_ref, img_ref = O.composite_rays_train(sr, rr, dl, rays, 1e-4, br)
    g_ws, g_dp, g_img = torch.randn(N), torch.randn(N), torch.randn(N, C)
    ((ws_ref * g_ws).sum() + (dp_ref * g_dp).sum() + (img_ref * g_img).sum()).backward()
    sg, rg, bgg = sig.to(dev).requires_grad_(), rgb.to(dev).requires_grad_(), bg.to(dev).requires_grad_()
    ws, dp, img = rm.composite_rays_train(sg, rg, dl.to(dev), rays.to(dev), 1e-4, bgg)
    _close(ws, ws_ref, 1e-4, 1e-6, "weights_sum")
    _close(dp, dp_ref, 1e-4, 1e-6, "depth")
    _close(img, img_ref, 1e-4, 1e-6, "image")
    torch.autograd.backward([ws, dp, img], [g_ws.to(dev), g_dp.to(dev), g_img.to(dev)])
    _close(sg.grad, sr.grad, 1e-3, 1e-5, "dsigma")
    _close(rg.grad, rr.grad, 1e-3, 1e-6, "drgb")
    _close(bgg.grad, br.grad, 1e-4, 1e-6, "dbg")
    # without background / without weights_sum & depth gradients
    ws2, dp2, img2 = rm.composite_rays_train(sig.to(dev), rgb.to(dev), dl.to(dev), rays.to(dev), 1e-4, None)
    ws3, dp3, img3 = O.composite_rays_train(sig, rgb, dl, rays, 1e-4, None)
    _close(img2, img3, 1e-4, 1e-6, "image no bg")
    assert float(ws2.max()) <= 1.0 + 1e-5


# ------------------------------------------------------------------------------ H11, Adam, occupancy
def test_background_net(dev):
    from src.latent_nerf.models.bg import background_net
    torch.manual_seed(0)
    N = 1000
    d = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1)
    p = O.init_bg_params()
    pr = {k: v.clone().requires_grad_() for k, v in p.items()}
    ref = O.bg_mlp(d, pr)
    g = torch.randn(N, 4)
    ref.backward(g)
    pg = {k: v.to(dev).requires_grad_() for k, v in p.items()}
    out = background_net(d.to(dev), pg["w1"], pg["b1"], pg["w2"], pg["b2"])
    _close(out, ref, 1e-4, 1e-5, "bg out")
    out.backward(g.to(dev))
    for k in p:
        _close_rel_max(pg[k].grad, pr[k].grad, 1e-4, "bg d" + k)


def test_adam_step_matches_torch(dev):
    from src.latent_nerf.raymarching import backend as B
    from src.latent_nerf.raymarching.raymarching import _p, _stream
    torch.manual_seed(0)
    n = 100003
    p0 = torch.randn(n)
    ref = p0.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    p, m, v = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    shadow = torch.empty(n, device=dev, dtype=torch.bfloat16)
    for step in range(1, 4):
        g = torch.randn(n)
        ref.grad = g.clone() * 0.5
        opt.step()
        gd = g.to(dev)
        B.call("lnerf_adam_step", _p(p), _p(gd), B.F32, _p(m), _p(v), _p(shadow), n, 1e-2, 0.9, 0.99, 1e-15, step, None,
               0.5, 1, _stream())
        assert float(gd.abs().max()) == 0.0  # gradient cleared in the same pass
    _close(p, ref, 1e-5, 1e-6, "adam params")
    assert torch.equal(shadow.cpu(), p.cpu().to(torch.bfloat16))
    # bf16 gradients (the wire buffer of the data-parallel all-reduce) == the same gradients widened to f32, bit for bit
    pa, ma, va = p.clone(), m.clone(), v.clone()
    pb, mb, vb = p.clone(), m.clone(), v.clone()
    gb = torch.randn(n, device=dev).to(torch.bfloat16)
    gf = gb.float()
    B.call("lnerf_adam_step", _p(pa), _p(gf), B.F32, _p(ma), _p(va), None, n, 1e-2, 0.9, 0.99, 1e-15, 4, None, 0.125, 0,
           _stream())
    B.call("lnerf_adam_step", _p(pb), _p(gb), B.BF16, _p(mb), _p(vb), None, n, 1e-2, 0.9, 0.99, 1e-15, 4, None, 0.125, 1,
           _stream())
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert float(gb.float().abs().max()) == 0.0


@pytest.mark.parametrize("capturable", [False, True])
def test_fused_adam_optimizer_matches_torch(dev, capturable):
    from src.latent_nerf.training.optimizer import FusedAdam
    torch.manual_seed(1)
    shapes = [(1 << 20) + 3, (64, 32), (64,), (5, 64), (5,)]
    ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    groups = [{"params": [ps[0]], "lr": 1e-2}, {"params": ps[1:], "lr": 1e-3}]
    opt = FusedAdam(groups, capturable=capturable)
    topt = torch.optim.Adam([{"params": [ref[0]], "lr": 1e-2}, {"params": ref[1:], "lr": 1e-3}], betas=(0.9, 0.99),
                            eps=1e-15)
    for _ in range(3):
        for p, r in zip(ps, ref):
            g = torch.randn(r.shape)
            r.grad = g * 0.25
            p.grad = g.to(dev)
        topt.step()
        opt.step(grad_scale=0.25)
        assert all(p.grad is None for p in ps)
    for p, r in zip(ps, ref):
        _close(p, r, 1e-5, 1e-6, "fused adam")


def test_occupancy_helpers(dev):
    from src.latent_nerf.raymarching import backend as B
    from src.latent_nerf.raymarching.raymarching import _p, _stream
    G = 16
    idx = torch.randperm(G ** 3)[:1000].int()
    noise = torch.rand(1000, 3)
    ref = O.occupancy_cell_points(idx.long(), 1, G, 2.0, noise)
    idx_d, noise_d = idx.to(dev), noise.to(dev)  # keep the device tensors alive across the raw-pointer call
    xyz = torch.empty(1000, 3, device=dev)
    B.call("lnerf_occ_cell_points", _p(idx_d), 1000, 1, G, 2.0, _p(noise_d), _p(xyz), _stream())
    _close(xyz, ref, 1e-6, 1e-6, "cell points")
    # decayed-max update: cells listed several times (the refresh samples with replacement), invalid cells (-1),
    # zero and negative new densities; the result is order-independent -> bit-exact against the oracle, and the same
    # bits from a shuffled list
    grid = torch.rand(1, G ** 3) * 3
    grid[0, :100] = -1.0
    idx = torch.randint(0, G ** 3, (3000,)).int()
    idx[:50] = idx[50:100]                                     # guaranteed duplicates
    sig = torch.rand(3000) * 5
    sig[::7] = 0.0
    sig[3::11] = -1.0
    ref_g = O.update_density_grid(grid, idx.long(), 0, sig, 0.95)
    outs = []
    for perm in (torch.arange(3000), torch.randperm(3000)):
        gg, sig_d, idx_d = grid.to(dev), sig[perm].to(dev), idx[perm].to(dev)
        cells = torch.zeros(G ** 3, dtype=torch.int32, device=dev)
        B.call("lnerf_occ_update", _p(gg[0]), _p(idx_d), 3000, _p(sig_d), 0.95, _p(cells), _stream())
        assert int(cells.abs().sum()) == 0                         # the scratch is left zero for the next call
        outs.append(gg.cpu())
    assert torch.equal(outs[0], ref_g) and torch.equal(outs[0], outs[1])
    # mean of max(grid, 0) in a fixed summation order: the same bits on every call
    gg = outs[0].to(dev)
    means = []
    for _ in range(3):
        mean = torch.zeros(1, device=dev)
        scratch = torch.zeros(256, device=dev)
        B.call("lnerf_occ_mean", _p(gg), gg.numel(), _p(mean), _p(scratch), _stream())
        means.append(float(mean))
    assert abs(means[0] - float(ref_g.clamp(min=0).mean())) < 1e-5 * max(means[0], 1.0)
    assert means[0] == means[1] == means[2]
    # lnerf_occ_update_mean (apply pass over the CELLS + mean in the same pass): the same grid and the same mean, bit
    # for bit, scratch left zero
    for perm in (torch.arange(3000), torch.randperm(3000)):
        g2, sig_d, idx_d = grid.to(dev), sig[perm].to(dev), idx[perm].to(dev)
        cells = torch.zeros(G ** 3, dtype=torch.int32, device=dev)
        mean2, scratch2 = torch.zeros(1, device=dev), torch.zeros(256, device=dev)
        B.call("lnerf_occ_update_mean", _p(g2[0]), G ** 3, _p(idx_d), 3000, _p(sig_d), 0.95, _p(cells), _p(mean2),
               _p(scratch2), _stream())
        assert int(cells.abs().sum()) == 0
        assert torch.equal(g2.cpu(), ref_g) and float(mean2) == means[0]


def test_scatter_bf16_gradient_output_matches_f32_path(dev):
    """lnerf_grid_encode_backward_bf16 (gradient WRITTEN in the all-reduce's wire format) == the f32 scatter followed
    by a round-to-nearest cast, bit for bit -- for spread-out and for clustered points alike (no order-dependent path is
    left); a second call overwrites (no accumulation) and the f32 scratch argument is never touched."""
    from src.latent_nerf.models import encoding as E
    enc = E.GridEncoder(scatter_variant=3).to(dev)
    levels = enc.levels
    M = 150000
    g = torch.Generator().manual_seed(4)
    x = ((torch.rand(M, 3, generator=g) * 2 - 1) * 0.999).to(dev)
    dfeat = torch.randn(16, M, 2, generator=g).to(dev)
    m_dev = torch.tensor([M], dtype=torch.int32, device=dev)
    ref = torch.zeros(levels.n_rows, 2, device=dev)
    E.grid_encode_backward(x, 1.0, dfeat, levels, M, m_dev, M, ref, variant=3)
    enc.grad_sink = E.GradSink(enc.embeddings.data)
    enc.grad_sink.wire.fill_(7.0)                      # stale content must be overwritten everywhere
    for _ in range(2):
        E.grid_encode_backward_bf16(x, 1.0, dfeat, enc, M, m_dev, M, 3)
    assert torch.equal(enc.grad_sink.wire, ref.to(torch.bfloat16))
    assert float(enc.grad_sink.zero.abs().max()) == 0.0
    # clustered points (every record of a level in a few buckets): the same code path, the same bits
    xc = (torch.tensor([[0.1234, -0.3456, 0.4567]]) + torch.rand(6000, 3, generator=g) * 1e-5).to(dev)
    dc = torch.randn(16, 6000, 2, generator=g).to(dev)
    refc = torch.zeros(levels.n_rows, 2, device=dev)
    E.grid_encode_backward(xc, 1.0, dc, levels, 6000, None, 6000, refc, variant=3)
    E.grid_encode_backward_bf16(xc, 1.0, dc, enc, 6000, None, 6000, 3)
    assert torch.equal(enc.grad_sink.wire, refc.to(torch.bfloat16))
    assert float(enc.grad_sink.zero.abs().max()) == 0.0
