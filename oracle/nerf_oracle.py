"""CPU oracle for the latent-NeRF render hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import it.  The product path (the package under
`latent-nerf-test_amd/`) never imports anything from `oracle/` and fails loudly when the HIP
library is missing.

PARITY UNPINNED for H1-H11: the reference checkout (`/root/reference`) does not contain the
NeRF renderer (`scripts/train_latent_nerf.py:3-4` imports `src.latent_nerf.*`, which is absent;
README.md:152-156 only describes it), has no tests and no golden vectors for it, and pins no
upstream revision (requirements.txt:1-18).  This file is therefore the *normative* statement
of the algorithm the HIP kernels implement: a plain fp32 PyTorch restatement of the published
Instant-NGP multiresolution hash encoding and of the torch-ngp style occupancy-grid ray march
and front-to-back compositing that README.md:163 names as the code's origin.  What the
reference does pin is restated from the files that are present and cited inline:

  * camera placement           src/latent_paint/models/render.py:19-31
  * pose distribution          src/latent_paint/training/views_dataset.py:9-35
  * view-direction bucket      src/utils.py:8-27            (golden: tests/golden/utils_golden.json)
  * image byte conversion      src/utils.py:57-62           (golden: tests/golden/utils_golden.json)
  * renderer->trainer dict     src/latent_paint/models/textured_mesh.py:181-220
  * SDS weighting / hand-off   src/stable_diffusion.py:274,320-321,334;
                               src/latent_paint_mesh/training/trainer.py:657-658
  * optimiser                  src/latent_paint/training/trainer.py:93-95

All arithmetic is float32 on the CPU.  Where the HIP kernels must reproduce a *discrete*
decision bit-for-bit (which lattice points of a ray are occupied) the float operations are
written as separate, un-fused multiplies and adds in a fixed order, and the kernels use the
same order with contraction disabled (see DESIGN.md "Arithmetic contract").
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SQRT3 = 1.7320508075688772
FLT_MAX = 3.4028234663852886e38


# --------------------------------------------------------------------------------------
# src/utils.py counterparts (pinned by golden vectors generated from the reference itself)
# --------------------------------------------------------------------------------------
def get_view_direction(elev: torch.Tensor, azim: torch.Tensor, top=30, front=0, angle=45) -> torch.Tensor:
    """Restates src/utils.py:8-27.  `top`, `front`, `angle` are treated as DEGREES and
    converted here, exactly as the reference does, even though its callers
    (views_dataset.py:12-22) already pass radians -- the double conversion is the
    reference's actual behaviour and the golden vectors capture it."""
    two_pi = 2.0 * np.pi

    def radd(x):
        return np.deg2rad(x) % two_pi

    azim = azim % two_pi
    elev = elev % two_pi
    view = torch.zeros(elev.shape[0], dtype=torch.long)
    view[(radd(front - angle) <= azim) | (azim < radd(front + angle))] = 0
    view[(radd(front + 180 + angle) <= azim) & (azim < radd(front - angle))] = 1
    view[(radd(front + 180 - angle) <= azim) & (azim < radd(front + 180 + angle))] = 2
    view[(radd(front + angle) <= azim) & (azim < radd(front + 180 - angle))] = 3
    view[elev < radd(top)] = 4
    view[elev > radd(180 - top)] = 5
    return view


def tensor2numpy(t: torch.Tensor) -> np.ndarray:
    """Restates src/utils.py:57-62."""
    a = t.detach().cpu().numpy()
    if a.min() < 0:
        a = (a * 0.5) + 0.5
    return (a * 255).astype(np.uint8)


# --------------------------------------------------------------------------------------
# H1  camera pose + ray generation
# --------------------------------------------------------------------------------------
def pose_from_angles(theta: float, phi: float, radius: float, target=(0.0, 0.0, 0.0)) -> torch.Tensor:
    """Camera-to-world [4,4] for elevation-from-+y `theta`, azimuth `phi` (radians), distance
    `radius`.  Eye position follows src/latent_paint/models/render.py:19-23
    (x = r sin(theta) sin(phi), y = r cos(theta), z = r sin(theta) cos(phi)); look-at target and
    world up (+y) follow :25-30.  Columns are (right, down, forward): image x grows to the
    right, image y grows downwards, the camera looks along +z_cam."""
    eye = torch.tensor([radius * math.sin(theta) * math.sin(phi),
                        radius * math.cos(theta),
                        radius * math.sin(theta) * math.cos(phi)], dtype=torch.float64)
    tgt = torch.tensor(target, dtype=torch.float64)
    up = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64)
    fwd = tgt - eye
    fwd = fwd / fwd.norm().clamp_min(1e-20)
    right = torch.linalg.cross(fwd, up)
    if float(right.norm()) < 1e-8:  # looking straight down/up: pick a stable right vector
        right = torch.tensor([1.0, 0.0, 0.0], dtype=torch.float64)
    right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    c2w = torch.eye(4, dtype=torch.float64)
    c2w[:3, 0] = right
    c2w[:3, 1] = down
    c2w[:3, 2] = fwd
    c2w[:3, 3] = eye
    return c2w.to(torch.float32)


def get_rays(c2w: torch.Tensor, fx: float, fy: float, cx: float, cy: float, H: int, W: int
             ) -> Tuple[torch.Tensor, torch.Tensor]:
    """H1.  c2w [B,4,4] -> rays_o, rays_d [B,H*W,3].  Pixel (i=column, j=row) looks through its
    centre: dir_cam = normalise((i+.5-cx)/fx, (j+.5-cy)/fy, 1); rays_d = R dir_cam."""
    c2w = c2w.to(torch.float32)
    if c2w.dim() == 2:
        c2w = c2w[None]
    B = c2w.shape[0]
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                          indexing="ij")
    xs = (i.reshape(-1) + 0.5 - cx) / fx
    ys = (j.reshape(-1) + 0.5 - cy) / fy
    zs = torch.ones_like(xs)
    inv = 1.0 / torch.sqrt(xs * xs + ys * ys + zs * zs)
    d = torch.stack([xs * inv, ys * inv, zs * inv], dim=-1)  # [HW,3]
    R = c2w[:, :3, :3]  # [B,3,3]
    rays_d = (d[None, :, 0:1] * R[:, None, :, 0] + d[None, :, 1:2] * R[:, None, :, 1]
              + d[None, :, 2:3] * R[:, None, :, 2])
    rays_o = c2w[:, None, :3, 3].expand(B, H * W, 3).contiguous()
    return rays_o, rays_d.contiguous()


# --------------------------------------------------------------------------------------
# H2  ray / AABB slab test
# --------------------------------------------------------------------------------------
def near_far_from_aabb(rays_o: torch.Tensor, rays_d: torch.Tensor, aabb, min_near: float
                       ) -> Tuple[torch.Tensor, torch.Tensor]:
    """H2.  rays [N,3]; aabb = (xmin,ymin,zmin,xmax,ymax,zmax).  A miss gives near = far = FLT_MAX."""
    aabb = torch.as_tensor(aabb, dtype=torch.float32)
    rd = 1.0 / rays_d
    t1 = (aabb[:3] - rays_o) * rd
    t2 = (aabb[3:] - rays_o) * rd
    tmin = torch.minimum(t1, t2)
    tmax = torch.maximum(t1, t2)
    near = torch.maximum(torch.maximum(tmin[:, 0], tmin[:, 1]), tmin[:, 2])
    far = torch.minimum(torch.minimum(tmax[:, 0], tmax[:, 1]), tmax[:, 2])
    miss = ~(far >= near)  # also catches NaN from 0*inf
    near = torch.maximum(near, torch.tensor(min_near, dtype=torch.float32))
    miss = miss | ~(far >= near)
    near = torch.where(miss, torch.full_like(near, FLT_MAX), near)
    far = torch.where(miss, torch.full_like(far, FLT_MAX), far)
    return near, far


# --------------------------------------------------------------------------------------
# H3  Morton order + occupancy bitfield
# --------------------------------------------------------------------------------------
def _expand_bits(v: torch.Tensor) -> torch.Tensor:
    m = 0xFFFFFFFF
    v = v.to(torch.int64)
    v = ((v * 0x00010001) & m) & 0xFF0000FF
    v = ((v * 0x00000101) & m) & 0x0F00F00F
    v = ((v * 0x00000011) & m) & 0xC30C30C3
    v = ((v * 0x00000005) & m) & 0x49249249
    return v


def morton3d(coords: torch.Tensor) -> torch.Tensor:
    """coords [...,3] (0 <= c < 1024) -> interleaved index, x in bit 0 (int64)."""
    return _expand_bits(coords[..., 0]) | (_expand_bits(coords[..., 1]) << 1) | (_expand_bits(coords[..., 2]) << 2)


def _compact_bits(x: torch.Tensor) -> torch.Tensor:
    x = x & 0x49249249
    x = (x | (x >> 2)) & 0xC30C30C3
    x = (x | (x >> 4)) & 0x0F00F00F
    x = (x | (x >> 8)) & 0xFF0000FF
    x = (x | (x >> 16)) & 0x0000FFFF
    return x


def morton3d_invert(idx: torch.Tensor) -> torch.Tensor:
    idx = idx.to(torch.int64)
    return torch.stack([_compact_bits(idx), _compact_bits(idx >> 1), _compact_bits(idx >> 2)], dim=-1)


def packbits(grid: torch.Tensor, thresh: float) -> torch.Tensor:
    """grid float32 [n] (n % 8 == 0, Morton order per cascade) -> uint8 [n/8]; bit k of byte b is
    grid[8b+k] > thresh."""
    bits = (grid.reshape(-1, 8) > thresh).to(torch.int64)
    weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int64)
    return (bits * weights).sum(-1).to(torch.uint8)


def density_grid_from_function(fn, G: int, cascade: int, bound: float) -> torch.Tensor:
    """Fill a [cascade, G^3] Morton-ordered density grid from fn(xyz[n,3]) evaluated at the cell
    centres (used to build analytic test scenes such as the SURVEY §8(d) sphere)."""
    idx = torch.arange(G ** 3, dtype=torch.int64)
    coords = morton3d_invert(idx).to(torch.float32)
    out = torch.empty(cascade, G ** 3, dtype=torch.float32)
    for c in range(cascade):
        b = min(2.0 ** c, bound)
        xyz = ((coords + 0.5) / G * 2.0 - 1.0) * b
        out[c] = fn(xyz)
    return out


# --------------------------------------------------------------------------------------
# H4  occupancy-pruned ray march (training)
# --------------------------------------------------------------------------------------
def _frexp_exponent(x: torch.Tensor) -> torch.Tensor:
    """exponent e of frexp: x = m * 2^e with 0.5 <= m < 1 (0 for x == 0)."""
    _, e = torch.frexp(x)
    return e.to(torch.int64)


def march_cell_index(x: torch.Tensor, dt: torch.Tensor, bound: float, cascade: int, G: int) -> torch.Tensor:
    """Bit index (cascade level * G^3 + Morton(cell)) of points x [..,3] already clamped to
    [-bound, bound].  Float ops are un-fused and ordered exactly as the HIP kernel's."""
    if cascade > 1:
        mx = x.abs().amax(-1)
        lvl_pos = _frexp_exponent(mx).clamp(0, cascade - 1)
        lvl_dt = _frexp_exponent(dt * (G * 0.5)).clamp(0, cascade - 1)
        level = torch.maximum(lvl_pos, lvl_dt.expand_as(lvl_pos))
        mip_bound = torch.minimum(torch.pow(torch.tensor(2.0), level.to(torch.float32)),
                                  torch.tensor(float(bound)))
        rb = (1.0 / mip_bound)[..., None]
    else:
        level = torch.zeros(x.shape[:-1], dtype=torch.int64)
        rb = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(min(1.0, float(bound)), dtype=torch.float32)
    u = x * rb
    u = u + 1.0
    u = u * (0.5 * G)
    u = u.clamp(0.0, float(G - 1))
    n = u.to(torch.int64)  # truncation, values are >= 0
    return level * (G ** 3) + morton3d(n)


def march_noise(N: int, seed: int, step: int) -> torch.Tensor:
    """Counter-based jitter u_n = hash(n, seed, step) in [0,1) of lnerf_march_rays_train's `noise_counter` form
    (csrc/rays.hip march_hash_uniform; our own definition -- the upstream draws torch.rand(N)): a 32-bit integer
    finaliser, the top 24 bits scaled by 2^-24."""
    import numpy as np
    M32 = np.uint64(0xFFFFFFFF)
    n = np.arange(N, dtype=np.uint64)
    x = (n * np.uint64(0x9E3779B1) + np.uint64(seed & 0xFFFFFFFF)) & M32
    x ^= (np.uint64(step & 0xFFFFFFFF) * np.uint64(0x85EBCA77)) & M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return torch.from_numpy(((x >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)))


def march_rays_train(rays_o, rays_d, nears, fars, bitfield, bound: float, cascade: int, G: int,
                     max_steps: int = 1024, dt_gamma: float = 0.0, noises: Optional[torch.Tensor] = None):
    """H4.  For every ray walk the step lattice t_0 = near + dt(near)*noise,
    t_{k+1} = t_k + clamp(t_k*dt_gamma, dt_min, dt_max) while t_k < far, and emit a sample
    at every lattice point whose occupancy bit is set, up to `max_steps` samples per ray.
    (Skipping empty voxels, as marching implementations do, never leaves this lattice, so
    the emitted set is the same.)  With dt_gamma == 0 the lattice is t_k = t_0 + k*dt_min.

    Returns xyzs [M,3], dirs [M,3], deltas [M,2] = (dt_k, t_k), rays int32 [N,3] =
    (ray id, offset, count) in ray order, and M."""
    rays_o = rays_o.to(torch.float32)
    rays_d = rays_d.to(torch.float32)
    N = rays_o.shape[0]
    f32 = torch.float32
    dt_min = torch.tensor(2.0 * SQRT3 / max_steps, dtype=f32)
    dt_max = torch.tensor(2.0 * SQRT3 * (2 ** (cascade - 1)) / G, dtype=f32)
    hit = nears < fars
    near = torch.where(hit, nears, torch.zeros_like(nears))
    far = torch.where(hit, fars, torch.zeros_like(fars))
    gam = torch.tensor(dt_gamma, dtype=f32)
    dt0 = torch.clamp(near * gam, dt_min, dt_max)
    if noises is None:
        noises = torch.zeros(N, dtype=f32)
    t0 = near + dt0 * noises.to(f32)
    bits = bitfield.to(torch.int64)

    def occupied(t, dt):
        x = rays_d[:, None, :] * t[..., None]
        x = x + rays_o[:, None, :]
        x = x.clamp(-bound, bound)
        idx = march_cell_index(x, dt, bound, cascade, G)
        return x, ((bits[idx >> 3] >> (idx & 7)) & 1).bool()

    if dt_gamma == 0.0:
        # number of lattice points any ray can have before reaching far
        span = (far - t0).clamp_min(0)
        K = int(torch.ceil(span.max() / dt_min).item()) + 2 if N > 0 else 0
        k = torch.arange(K, dtype=f32)
        t = k[None, :] * dt_min
        t = t + t0[:, None]  # [N,K]
        dt = dt_min.expand(N, K)
    else:
        ts, dts = [], []
        tcur = t0.clone()
        alive = hit.clone()
        while bool((alive & (tcur < far)).any()):
            d = torch.clamp(tcur * gam, dt_min, dt_max)
            ts.append(tcur.clone())
            dts.append(d)
            tcur = tcur + d
            if len(ts) > 1_000_000:
                raise RuntimeError("march lattice did not terminate")
        if ts:
            t = torch.stack(ts, 1)
            dt = torch.stack(dts, 1)
        else:
            t = torch.zeros(N, 0, dtype=f32)
            dt = torch.zeros(N, 0, dtype=f32)
    valid = hit[:, None] & (t < far[:, None])
    if t.shape[1] > 0:
        x, occ = occupied(t, dt)
        occ = occ & valid
        rank = torch.cumsum(occ.to(torch.int64), 1) - occ.to(torch.int64)
        occ = occ & (rank < max_steps)
    else:
        x = torch.zeros(N, 0, 3)
        occ = torch.zeros(N, 0, dtype=torch.bool)
    counts = occ.sum(1).to(torch.int64)
    offsets = torch.cumsum(counts, 0) - counts
    M = int(counts.sum().item())
    sel = occ.reshape(-1).nonzero(as_tuple=False).squeeze(-1)  # row-major => ray order, t order
    ray_of = sel // max(t.shape[1], 1)
    xyzs = x.reshape(-1, 3)[sel]
    dirs = rays_d[ray_of]
    deltas = torch.stack([dt.reshape(-1)[sel], t.reshape(-1)[sel]], -1)
    rays = torch.stack([torch.arange(N, dtype=torch.int64), offsets, counts], -1).to(torch.int32)
    return xyzs, dirs, deltas, rays, M


# --------------------------------------------------------------------------------------
# H5/H6  multiresolution hash grid (forward; backward through autograd)
# --------------------------------------------------------------------------------------
@dataclass
class GridLevels:
    num_levels: int
    level_dim: int
    base_resolution: int
    desired_resolution: int
    log2_hashmap_size: int
    offsets: list          # L+1 entry offsets (in table rows)
    scales: list           # float32 per-level scale  (exp2(l*S)*base - 1)
    resolutions: list      # ceil(scale)+1
    blocked: bool = False  # opt-in layout of the hashed levels (grid_corner_indices)
    tiled: bool = False    # the upstream encoder's `gridtype = "tiled"`: dense index wrapped instead of hashed

    @property
    def n_rows(self) -> int:
        return self.offsets[-1]


def make_grid_levels(num_levels=16, level_dim=2, base_resolution=16, desired_resolution=2048,
                     log2_hashmap_size=19, blocked=False, tiled=False) -> GridLevels:
    """Level table of the Instant-NGP hash grid (align_corners=False convention: a level with
    resolution R stores (R+1)^3 vertices, capped at 2^log2_hashmap_size, rounded up to 8)."""
    max_params = 2 ** log2_hashmap_size
    if num_levels > 1:
        per_level_scale = 2.0 ** (math.log2(desired_resolution / base_resolution) / (num_levels - 1))
    else:
        per_level_scale = 1.0
    S = math.log2(per_level_scale)
    offsets, scales, ress = [0], [], []
    for l in range(num_levels):
        res_host = int(math.ceil(base_resolution * per_level_scale ** l))
        n = min(max_params, (res_host + 1) ** 3)
        n = int(math.ceil(n / 8) * 8)
        offsets.append(offsets[-1] + n)
        scale = float(np.float32(2.0 ** (l * S) * base_resolution - 1.0))
        scales.append(scale)
        ress.append(int(math.ceil(scale)) + 1)
    return GridLevels(num_levels, level_dim, base_resolution, desired_resolution, log2_hashmap_size,
                      offsets, scales, ress, bool(blocked), bool(tiled))


_PRIMES = (1, 2654435761, 805459861)


def grid_corner_indices(pos_grid: torch.Tensor, res: int, hashmap_size: int, blocked: bool = False,
                        tiled: bool = False) -> torch.Tensor:
    """pos_grid int64 [...,3] -> row index inside the level (uint32 arithmetic).
    tiled (the upstream gridencoder's `gridtype = "tiled"`, SURVEY.md Appendix A [UPSTREAM-RECALL]): the dense index
    x + y (res + 1) + z (res + 1)^2, wrapped to 32 bits, modulo the table size -- no hash.
    blocked (our own opt-in variant, include/lnerf_hip.h LNERF_GRID_BLOCKED; dense levels unchanged): on a hashed level
    the lattice is cut into blocks of 4 x 2 x 2 vertices, the block coordinate is hashed, a block's 16 rows are
    consecutive: row = (hash(x >> 2, y >> 1, z >> 1) mod (hashmap_size // 16)) * 16 + (x & 3) + 4 (y & 1) + 8 (z & 1)."""
    m = 0xFFFFFFFF
    if tiled and (res + 1) ** 3 > hashmap_size:
        st = res + 1
        return ((pos_grid[..., 0] + pos_grid[..., 1] * st + pos_grid[..., 2] * st * st) & m) % hashmap_size
    if blocked and (res + 1) ** 3 > hashmap_size:
        x, y, z = pos_grid[..., 0], pos_grid[..., 1], pos_grid[..., 2]
        h = (((x >> 2) * _PRIMES[0]) & m) ^ (((y >> 1) * _PRIMES[1]) & m) ^ (((z >> 1) * _PRIMES[2]) & m)
        return (h % (hashmap_size // 16)) * 16 + ((x & 3) | ((y & 1) << 2) | ((z & 1) << 3))
    stride = 1
    index = torch.zeros(pos_grid.shape[:-1], dtype=torch.int64)
    d = 0
    while d < 3 and stride <= hashmap_size:
        index = (index + pos_grid[..., d] * stride) & m
        stride *= (res + 1)
        d += 1
    if stride > hashmap_size:
        index = ((pos_grid[..., 0] * _PRIMES[0]) & m) ^ ((pos_grid[..., 1] * _PRIMES[1]) & m) \
            ^ ((pos_grid[..., 2] * _PRIMES[2]) & m)
    return index % hashmap_size


def f26_round(v: torch.Tensor) -> torch.Tensor:
    """Value format of the packed 8-byte scatter records (csrc/grid.hip Rec8 / f26_round; our own definition): sign, 8
    exponent and 17 mantissa bits -- the f32 bit pattern plus 0x20 (round to nearest, ties away from zero) with the low
    6 bits dropped.  Relative rounding 2^-18 per addend."""
    bits = v.contiguous().view(torch.int32)
    finite = (bits & 0x7F800000) != 0x7F800000
    bits = torch.where(finite, bits + 0x20, bits) & ~0x3F
    return bits.view(torch.float32)


class _CornerGather(torch.autograd.Function):
    """w * table[idx] of one cell corner.  Backward: the records of the bucketed scatter -- one contribution w * g per
    (sample, corner), rounded to the 26-bit record format when `rec8` (scatter variant 3, the bf16 configuration), summed
    per table row.  (The kernel sums in 64-bit fixed point at 2^-30 of the level's largest value and, on levels whose
    cells hold runs of samples, rounds the run's f32 sum instead of each addend: both below 2^-18 relative.)"""

    @staticmethod
    def forward(ctx, table, idx, w, rec8):
        ctx.save_for_backward(idx, w)
        ctx.meta = (table.shape, rec8)
        return w[:, None] * table[idx]

    @staticmethod
    def backward(ctx, g):
        idx, w = ctx.saved_tensors
        shape, rec8 = ctx.meta
        contrib = w[:, None] * g
        if rec8:
            contrib = f26_round(contrib)
        return torch.zeros(shape, dtype=g.dtype).index_add_(0, idx, contrib), None, None, None


def grid_encode(x01: torch.Tensor, table: torch.Tensor, lv: GridLevels, rec8: bool = False) -> torch.Tensor:
    """H5.  x01 [M,3] in [0,1]; table [n_rows, F] -> features [M, L*F] (level-major columns:
    column l*F+f).  Differentiable w.r.t. `table` (H6 = autograd of the index ops; `rec8`: with the record rounding of
    scatter variant 3, see _CornerGather)."""
    M = x01.shape[0]
    outs = []
    for l in range(lv.num_levels):
        scale = torch.tensor(lv.scales[l], dtype=torch.float32)
        res = lv.resolutions[l]
        hsize = lv.offsets[l + 1] - lv.offsets[l]
        pos = x01 * scale
        pos = pos + 0.5
        pg = torch.floor(pos)
        frac = pos - pg
        pg = pg.to(torch.int64)
        acc = torch.zeros(M, lv.level_dim, dtype=torch.float32)
        for c in range(8):
            bx, by, bz = c & 1, (c >> 1) & 1, (c >> 2) & 1
            wx = frac[:, 0] if bx else 1.0 - frac[:, 0]
            wy = frac[:, 1] if by else 1.0 - frac[:, 1]
            wz = frac[:, 2] if bz else 1.0 - frac[:, 2]
            w = (wx * wy) * wz
            corner = pg + torch.tensor([bx, by, bz], dtype=torch.int64)
            idx = grid_corner_indices(corner, res, hsize, getattr(lv, "blocked", False), getattr(lv, "tiled", False)) \
                + lv.offsets[l]
            acc = acc + (_CornerGather.apply(table, idx, w, True) if rec8 else w[:, None] * table[idx])
        outs.append(acc)
    return torch.cat(outs, dim=-1)


# --------------------------------------------------------------------------------------
# H7  sigma / latent MLP
# --------------------------------------------------------------------------------------
class _TruncExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(max=15.0))


trunc_exp = _TruncExp.apply


def density_blob(x: torch.Tensor, scale: float = 5.0, std: float = 0.2) -> torch.Tensor:
    d2 = (x[..., 0] * x[..., 0] + x[..., 1] * x[..., 1]) + x[..., 2] * x[..., 2]
    sd = torch.tensor(std, dtype=torch.float32)
    denom = (2.0 * sd) * sd  # f32, same op order as the kernel's 2.0f*std*std
    return scale * torch.exp(-d2 / denom)


def _bf16r(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


class _Bf16Linear(torch.autograd.Function):
    """One layer of the bf16 MFMA path (csrc/mlp_bf16.hip), forward AND backward: every MFMA operand is a bf16 value,
    every accumulation is f32.
        forward :  y = r(x) r(W)^T + b
        backward:  the upstream gradient dZ is rounded ONCE (the packed B fragment the kernel builds from it) and that
                   rounded value feeds all three products: dX = r(dZ) r(W),  dW = r(dZ)^T r(x),  db = sum r(dZ)
    (relu's mask is the sign of the rounded activation, i.e. of the activation itself.)"""

    @staticmethod
    def forward(ctx, x, w, b):
        xr, wr = _bf16r(x), _bf16r(w)
        ctx.save_for_backward(xr, wr)
        return xr @ wr.t() + b

    @staticmethod
    def backward(ctx, g):
        xr, wr = ctx.saved_tensors
        gr = _bf16r(g)
        return gr @ wr, gr.t() @ xr, gr.sum(0)


def sigma_latent_mlp(feat: torch.Tensor, xyz: torch.Tensor, params: dict, blob_scale=5.0, blob_std=0.2,
                     bf16: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """H7.  feat [M,32] -> h = W3 relu(W2 relu(W1 feat + b1) + b2) + b3  (32 -> 64 -> 64 -> 1+C);
    sigma = trunc_exp(h[:,0] + blob(xyz)); latent = h[:,1:]  (no squashing in latent mode).
    `bf16=True`: the arithmetic of the bf16 MFMA path in both directions (_Bf16Linear) -- the operands of every product
    (features, weights, hidden activations; in the backward pass the pre-activation gradients) are rounded to bfloat16,
    accumulation is fp32."""
    if bf16:
        lin = _Bf16Linear.apply
    else:
        lin = F.linear
    h = lin(feat, params["w1"], params["b1"])
    h = F.relu(h)
    h = lin(h, params["w2"], params["b2"])
    h = F.relu(h)
    h = lin(h, params["w3"], params["b3"])
    sigma = trunc_exp(h[:, 0] + density_blob(xyz, blob_scale, blob_std))
    return sigma, h[:, 1:]


def init_mlp_params(in_dim=32, hidden=64, out_dim=5, seed=0) -> dict:
    """torch.nn.Linear default init under a seed."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    dims = [(hidden, in_dim), (hidden, hidden), (out_dim, hidden)]
    for i, (o, k) in enumerate(dims, 1):
        bound = 1.0 / math.sqrt(k)
        p[f"w{i}"] = (torch.rand(o, k, generator=g) * 2 - 1) * bound
        p[f"b{i}"] = (torch.rand(o, generator=g) * 2 - 1) * bound
    return p


# --------------------------------------------------------------------------------------
# H8/H9  front-to-back compositing (forward; backward through autograd)
# --------------------------------------------------------------------------------------
def composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh: float = 1e-4, bg_color=None):
    """H8.  sigmas [M], rgbs [M,C], deltas [M,2]=(dt,t), rays int [N,3]=(id,offset,count).
    Per ray, front to back: alpha_i = 1-exp(-sigma_i dt_i), T_i = prod_{j<i}(1-alpha_j),
    w_i = alpha_i T_i, and sample i contributes only while T_i >= T_thresh.
    Returns weights_sum [N], depth [N], image [N,C]  (image += (1-weights_sum) bg if given)."""
    N = rays.shape[0]
    C = rgbs.shape[1]
    ids = rays[:, 0].to(torch.int64)
    offs = rays[:, 1].to(torch.int64)
    cnts = rays[:, 2].to(torch.int64)
    K = int(cnts.max().item()) if N > 0 else 0
    ws = torch.zeros(N, dtype=torch.float32)
    depth = torch.zeros(N, dtype=torch.float32)
    image = torch.zeros(N, C, dtype=torch.float32)
    if K > 0:
        k = torch.arange(K)
        valid = k[None, :] < cnts[:, None]
        idx = (offs[:, None] + k[None, :]).clamp(max=max(sigmas.shape[0] - 1, 0))
        idx = torch.where(valid, idx, torch.zeros_like(idx))
        sg = torch.where(valid, sigmas[idx], torch.zeros(1))
        dt = torch.where(valid, deltas[idx, 0], torch.zeros(1))
        tt = torch.where(valid, deltas[idx, 1], torch.zeros(1))
        tau = sg * dt
        csum = torch.cumsum(tau, 1) - tau  # exclusive
        T = torch.exp(-csum)
        alpha = 1.0 - torch.exp(-tau)
        keep = valid & (T >= T_thresh)
        w = torch.where(keep, alpha * T, torch.zeros(1))
        rgb = torch.where(valid[..., None], rgbs[idx], torch.zeros(1))
        ws_l = w.sum(1)
        depth_l = (w * tt).sum(1)
        img_l = (w[..., None] * rgb).sum(1)
        ws = ws.index_add(0, ids, ws_l)
        depth = depth.index_add(0, ids, depth_l)
        image = image.index_add(0, ids, img_l)
    if bg_color is not None:
        image = image + (1.0 - ws)[:, None] * bg_color
    return ws, depth, image


# --------------------------------------------------------------------------------------
# H10  occupancy grid refresh
# --------------------------------------------------------------------------------------
def occupancy_cell_points(indices: torch.Tensor, cascade_level: int, G: int, bound: float,
                          noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """World positions of (jittered) cell centres for Morton indices of one cascade level.
    noise [n,3] in [0,1) (0.5 = centre)."""
    coords = morton3d_invert(indices).to(torch.float32)
    if noise is None:
        noise = torch.full_like(coords, 0.5)
    b = min(2.0 ** cascade_level, bound)
    u = coords + noise
    u = u * (2.0 / G)
    u = u - 1.0
    return u * b


def occ_hash(i, seed: int, step: int, k: int):
    """32-bit hash of lnerf_occ_sample (csrc/rays.hip occ_hash; our own definition): numpy uint64 arithmetic masked to
    32 bits."""
    import numpy as np
    M32 = np.uint64(0xFFFFFFFF)
    x = (np.asarray(i, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(seed & 0xFFFFFFFF)) & M32
    x ^= ((np.uint64(step & 0xFFFFFFFF) * np.uint64(0x85EBCA77)) + (np.uint64(k) * np.uint64(0xC2B2AE3D))) & M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


def occ_sample(grid_level: torch.Tensor, cascade_level: int, G: int, bound: float, n_rand: int, seed: int, step: int):
    """Steady-state cell sampling of the occupancy refresh (lnerf_occ_sample; our own definition -- the upstream refresh
    draws independently with torch.randint): indices [2*n_rand] = n_rand cells, then n_rand cells of the occupied ones
    (ascending list of cells with grid > 0; all cells when the list is empty), STRATIFIED: draw j of a half takes element
    lo + (h * (hi - lo) >> 32) of its population of n, lo = j n // n_rand, hi = (j + 1) n // n_rand, h = occ_hash(i, seed,
    step, 0) -- same marginal probabilities as independent draws, ascending by construction; and a jittered point in every
    cell (occ_cell_points' formula)."""
    import numpy as np
    n_cells = grid_level.numel()
    i = np.arange(2 * n_rand, dtype=np.uint64)
    h = occ_hash(i, seed, step, 0)
    j = np.where(i < n_rand, i, i - np.uint64(n_rand)).astype(np.uint64)
    occ = torch.nonzero(grid_level > 0).squeeze(-1).numpy().astype(np.uint64)
    second = (i >= n_rand) & (occ.size > 0)
    pop = np.where(second, np.uint64(max(occ.size, 1)), np.uint64(n_cells)).astype(np.uint64)
    lo = j * pop // np.uint64(n_rand)
    hi = (j + np.uint64(1)) * pop // np.uint64(n_rand)
    pick = lo + ((h * (hi - lo)) >> np.uint64(32))
    pick = np.minimum(pick, pop - np.uint64(1))
    idx = pick.copy()
    if occ.size > 0:
        idx[n_rand:] = occ[pick[n_rand:].astype(np.int64)]
    noise = np.stack([(occ_hash(i, seed, step, k) >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
                      for k in (1, 2, 3)], axis=-1)
    indices = torch.from_numpy(idx.astype(np.int64))
    return indices, occupancy_cell_points(indices, cascade_level, G, bound, torch.from_numpy(noise))


def update_density_grid(grid: torch.Tensor, indices: torch.Tensor, cascade_level: int,
                        new_sigma: torch.Tensor, decay: float = 0.95) -> torch.Tensor:
    """grid[c, idx] = max(grid[c, idx]*decay, s) for the sampled cells where both the old value and the new density
    are valid (>= 0).  A cell that is listed several times (the refresh draws its cells with replacement) takes the
    MAXIMUM of its new densities, decayed once: the result does not depend on the order of the list."""
    g = grid.clone()
    lvl = g[cascade_level]
    ok = new_sigma >= 0
    best = torch.full_like(lvl, -1.0)
    best = best.scatter_reduce(0, indices[ok], new_sigma[ok], reduce="amax", include_self=True)
    touched = (best >= 0) & (lvl >= 0)
    g[cascade_level] = torch.where(touched, torch.maximum(lvl * decay, best), lvl)
    return g


# --------------------------------------------------------------------------------------
# H11  background: frequency encoding + small MLP
# --------------------------------------------------------------------------------------
def freq_encode(d: torch.Tensor, degree: int = 6) -> torch.Tensor:
    """[n,3] -> [n, 3 + 3*2*degree]: (d, sin(2^k d), cos(2^k d))_k."""
    out = [d]
    for k in range(degree):
        out.append(torch.sin(d * (2.0 ** k)))
        out.append(torch.cos(d * (2.0 ** k)))
    return torch.cat(out, -1)


def bg_mlp(d: torch.Tensor, params: dict, degree: int = 6) -> torch.Tensor:
    """39 -> 64 -> C, ReLU hidden, linear output."""
    h = F.relu(F.linear(freq_encode(d, degree), params["w1"], params["b1"]))
    return F.linear(h, params["w2"], params["b2"])


def init_bg_params(in_dim=39, hidden=64, out_dim=4, seed=1) -> dict:
    g = torch.Generator().manual_seed(seed)
    p = {}
    for i, (o, k) in enumerate([(hidden, in_dim), (out_dim, hidden)], 1):
        bound = 1.0 / math.sqrt(k)
        p[f"w{i}"] = (torch.rand(o, k, generator=g) * 2 - 1) * bound
        p[f"b{i}"] = (torch.rand(o, generator=g) * 2 - 1) * bound
    return p


# --------------------------------------------------------------------------------------
# H0  whole frame (the CPU baseline `run` of bench.py and the end-to-end parity oracle)
# --------------------------------------------------------------------------------------
def render_frame(rays_o, rays_d, table, mlp_params, lv: GridLevels, bitfield, *, bound=1.0, cascade=1,
                 G=128, min_near=0.1, max_steps=1024, dt_gamma=0.0, noises=None, bg_color=None,
                 T_thresh=1e-4, bf16_mlp=False, bf16_table=False, blob_scale=5.0, blob_std=0.2, rec8=None):
    """rays [N,3] -> {'image' [N,C], 'depth' [N], 'weights_sum' [N], 'xyzs', 'sigmas', ...}.
    Differentiable w.r.t. `table` and the entries of `mlp_params`.
    rec8 (default: follows bf16_mlp, as the renderer's scatter variant does): table gradient with the rounding of the
    packed 8-byte scatter records."""
    if rec8 is None:
        rec8 = bool(bf16_mlp)
    aabb = [-bound, -bound, -bound, bound, bound, bound]
    with torch.no_grad():
        nears, fars = near_far_from_aabb(rays_o, rays_d, aabb, min_near)
        xyzs, dirs, deltas, rays, M = march_rays_train(rays_o, rays_d, nears, fars, bitfield, bound, cascade, G,
                                                       max_steps, dt_gamma, noises)
    x01 = (xyzs + bound) / (2.0 * bound)
    tab = table
    if bf16_table:
        tab = table + (_bf16r(table) - table).detach()  # bf16 shadow, straight-through to the master
    feat = grid_encode(x01, tab, lv, rec8=rec8)
    sigmas, rgbs = sigma_latent_mlp(feat, xyzs, mlp_params, blob_scale, blob_std, bf16=bf16_mlp)
    ws, depth, image = composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh, bg_color)
    return {"image": image, "depth": depth, "weights_sum": ws, "xyzs": xyzs, "dirs": dirs, "deltas": deltas,
            "rays": rays, "sigmas": sigmas, "rgbs": rgbs, "feat": feat, "M": M, "nears": nears, "fars": fars}


def sphere_density_grid(G=128, cascade=1, bound=1.0, radius=0.5, value=10.0) -> torch.Tensor:
    """SURVEY §8(d) synthetic occupancy: density `value` inside ||x|| < radius, 0 outside."""
    return density_grid_from_function(lambda x: (x.norm(dim=-1) < radius).float() * value, G, cascade, bound)


def adam_step(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.99, eps=1e-15):
    """torch.optim.Adam arithmetic (src/latent_paint/training/trainer.py:93-95 settings),
    returns updated (p, m, v)."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    mhat = m / (1 - beta1 ** step)
    vhat = v / (1 - beta2 ** step)
    p = p - lr * mhat / (torch.sqrt(vhat) + eps)
    return p, m, v


def synthetic_guidance(image, targets_rows, dirs, weights, seed: int, step: int, t_lo: int = 20, t_hi: int = 980,
                       noise_scale: float = 0.05):
    """The trainer's seeded synthetic guidance as csrc/guidance.hip computes it (lnerf_synthetic_guidance; our own
    stand-in for `diffusion.train_step` of the reference's src/stable_diffusion.py:248-334 -- weighting form of :274,
    :320-321): image [B, P, C], targets_rows [6, P, C], dirs [B] ->  w(t) (noise_scale z + image - target_dir)  with the
    timestep and the normal deviates from occ_hash-style counters of (seed, step, element): Box-Muller on two 24-bit
    uniforms per channel pair.  float64 transcendental functions here, f32 on the device: compare to ~1e-5."""
    import numpy as np
    B, P, C = image.shape
    n_t = t_hi - t_lo + 1
    h = int(occ_hash(np.array([0xFFFFFFFF], dtype=np.uint64), seed, step, 7)[0])
    t = t_lo + ((h * n_t) >> 32)
    w = float(weights[t])
    r = np.arange(B * P, dtype=np.uint64)
    z = np.zeros((B * P, 4), dtype=np.float64)
    for c in (0, 2):
        u1 = ((occ_hash(r, seed, step, c) >> np.uint64(8)).astype(np.float64) + 1.0) / 16777216.0
        u2 = (occ_hash(r, seed, step, c + 1) >> np.uint64(8)).astype(np.float64) / 16777216.0
        rad = np.sqrt(-2.0 * np.log(u1))
        z[:, c] = rad * np.cos(2.0 * np.pi * u2)
        z[:, c + 1] = rad * np.sin(2.0 * np.pi * u2)
    zt = torch.from_numpy(z[:, :C]).to(torch.float32).reshape(B, P, C)
    tgt = targets_rows[dirs.long().clamp(0, targets_rows.shape[0] - 1)]
    return (zt * noise_scale + (image - tgt)) * w, t
