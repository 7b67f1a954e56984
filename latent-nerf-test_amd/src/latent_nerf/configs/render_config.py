"""Render configuration of the latent-NeRF path.  Field names/defaults follow what the reference
still advertises for the absent package (`--render.nerf_type latent`,
cuda_ray, max_steps 1024, update_extra_interval 16, max_ray_batch 4096, density_thresh 10,
train 64x64, bound 1, dt_gamma 0, min_near 0.1, radius_range (1.0,1.5), fovy_range (40,70)),
plus the MI355X-specific knobs at the bottom."""
from typing import Optional
from dataclasses import dataclass
from typing import Tuple

from ..models.nerf_utils import NeRFType


@dataclass
class RenderConfig:
    # Whether to use the occupancy-grid (HIP) ray marcher
    cuda_ray: bool = True
    # Maximal number of samples per ray
    max_steps: int = 1024
    # Samples per ray of the uniform sampler (non-cuda_ray path)
    num_steps: int = 128
    upsample_steps: int = 0
    # Refresh the occupancy grid every N training steps
    update_extra_interval: int = 16
    # steady-state refreshes draw their cells on the device (lnerf_occ_sample: no host synchronisation); False = the
    # torch.nonzero / randint form
    occ_device_sampling: bool = True
    # Rays per launch at inference
    max_ray_batch: int = 4096
    # Occupancy threshold
    density_thresh: float = 10.0
    train_w: int = 64
    train_h: int = 64
    eval_w: int = 128
    eval_h: int = 128
    jitter_pose: bool = False
    # Scene is assumed inside [-bound, bound]^3
    bound: float = 1.0
    dt_gamma: float = 0.0
    min_near: float = 0.1
    radius_range: Tuple[float, float] = (1.0, 1.5)
    fovy_range: Tuple[float, float] = (40.0, 70.0)
    # fixed training view (theta deg, phi deg, radius, fovy deg) instead of the random pose distribution: benchmarking
    # and debugging only (bench.py's trainer companion uses it to put the trainer on the bench's own view)
    train_pose: Optional[Tuple[float, float, float, float]] = None
    dir_text: bool = True
    angle_overhead: float = 30.0
    angle_front: float = 60.0
    backbone: str = "grid"
    nerf_type: NeRFType = NeRFType.latent
    # > 0 enables the learned background net (bg colour otherwise comes from the caller)
    bg_radius: float = 0.0
    # occupancy grid resolution
    grid_size: int = 128
    # ---- MI355X knobs
    # "f32": exact-f32 MFMA MLP (parity path), "bf16": bf16 MFMA, f32 accumulate.  "auto" (default): decided
    # by the owner -- TrainConfig picks "bf16" when optim.fp16 else "f32"; a bare RenderConfig means "f32"
    mlp_precision: str = "auto"
    # "f32": gather reads the master table, "bf16": gather reads a bf16 shadow (half the bytes); "auto" as above
    table_dtype: str = "auto"
    # layout of the hashed levels of the table: "hash" = Instant-NGP's spatial hash of the vertex; "tiled" = the upstream
    # encoder's other layout (the dense index wrapped into the table: no hash); "blocked" = opt-in
    # variant that hashes 4 x 2 x 2 vertex BLOCKS and keeps a block's 16 rows in one 64-byte line of the bf16 table
    # (2.8 instead of 4.25 cache lines per sample and level in the gather; a different collision pattern, so tables are
    # not interchangeable between the two)
    # "auto" (default): decided by the owner like the precisions -- TrainConfig picks "blocked" with the bf16 table (its
    # 64-byte-line blocks are cut for 4-byte rows; measured: gather 75 -> 67 us at equal or better training error,
    # profiles/r04_ab_layout.jsonl) and "hash" with the f32 parity configuration; a bare RenderConfig means "hash"
    gridtype: str = "auto"
    # workgroup -> (level, tile) mapping of the gather/scatter: 0 = level on grid.y (measured 1.8x faster), 1 = XCD-pinned levels
    gather_variant: int = 0
    # hash-grid backward: 0/1 = global float atomics, 2 = two-pass bucketed scatter (LDS reduction, exact f32
    # records), 3 = the same with packed 8-byte records (values rounded to 17 mantissa bits),
    # -1 = auto: 3 with mlp_precision "bf16", 2 otherwise
    scatter_variant: int = -1
    # jitter of the march start (perturb=True): seed of the in-kernel counter-based generator
    # (lnerf_march_rays_train `noise_counter`: graph-capturable, no host RNG state); None = torch.rand(N) per call
    noise_seed: Optional[int] = 0x5EED
    # sample buffer capacity per view.  0 = automatic: rays * min(max_steps, 256) until the first occupancy refresh
    # has read the march counters back, then 1.5 x the largest sample count seen between refreshes (rounded up to
    # 64 Ki; rays that do not fit are dropped by the march's scan pass and counted, never written out of bounds)
    max_samples: int = 0

    def layout(self) -> str:
        """Resolved `gridtype` ("auto" on a bare RenderConfig = Instant-NGP's vertex hash)."""
        if self.gridtype not in ("auto", "hash", "tiled", "blocked"):
            raise ValueError("render.gridtype must be 'auto', 'hash', 'tiled' or 'blocked' (got %r)" % (self.gridtype,))
        return "hash" if self.gridtype == "auto" else self.gridtype

    def precision(self, name: str) -> str:
        """Resolved value of `mlp_precision` / `table_dtype` ("auto" on a bare RenderConfig = the f32 parity path)."""
        v = getattr(self, name)
        if v not in ("auto", "f32", "bf16"):
            raise ValueError("render.%s must be 'auto', 'f32' or 'bf16' (got %r)" % (name, v))
        return "f32" if v == "auto" else v
