"""Training configuration tree `log / render / optim / guide`, the surface the reference's CLI drives
(`python -m scripts.train_latent_nerf --config_path demo_configs/latent_nerf/lego_man.yaml`, or dotted flags
such as `--log.exp_name x --guide.text "..." --render.nerf_type latent`, README.md:64-69,92-97).

Field names and defaults follow the reference's present Latent-Paint config
(src/latent_paint/configs/train_config.py:7-97) and the NeRF flags its demo configs and README still
advertise for the absent package (demo_configs/latent_nerf/lego_man.yaml:1-10, README.md:140-142:
guide.shape_path, guide.mesh_scale, guide.proximal_surface, optim.lambda_shape, optim.seed, optim.iters).
pyrallis is not installed here, so `load_config()` implements the same two input forms on top of
argparse + yaml; when pyrallis is importable the dataclasses work with `pyrallis.wrap()` unchanged."""
from dataclasses import dataclass, field
from pathlib import Path
from typing import Optional

from ... import config_cli as _cli
from .render_config import RenderConfig


@dataclass
class GuideConfig:
    # Guiding text prompt
    text: str = ""
    # A mesh to be used as a shape prior (sketch-shape guidance)
    shape_path: Optional[str] = None
    # Append direction to text prompts
    append_direction: bool = True
    # A Textual-Inversion concept to use
    concept_name: Optional[str] = None
    # A huggingface diffusion model to use
    diffusion_name: str = "CompVis/stable-diffusion-v1-4"
    # "synthetic": seeded stand-in for the diffusion model (offline); "stable-diffusion": diffusers adapter
    guidance: str = "synthetic"
    # Scale of the mesh that is used as the shape prior
    mesh_scale: float = 0.7
    # Strictness of the shape guidance near the surface
    proximal_surface: float = 0.3


@dataclass
class OptimConfig:
    seed: int = 0
    iters: int = 5000
    lr: float = 1e-3
    # Use amp-style mixed precision (bf16 table shadow + bf16 MFMA MLP)
    fp16: bool = True
    # Start from a checkpoint
    resume: bool = False
    ckpt: Optional[str] = None
    lambda_sparsity: float = 5e-4
    lambda_shape: float = 5e-6
    # views per optimisation step (sharded one-per-GPU in data-parallel runs)
    views_per_step: int = 1
    # single process and one view per step: the hash table's Adam step runs inside the scatter of the backward pass
    fuse_table_update: bool = True
    # data parallel, bf16: level groups the table gradient is exchanged in (each group's all-reduce is launched behind
    # its own sums while the next group is still being summed); 1 = one collective for the whole table
    exchange_groups: int = 4
    # data parallel, bf16: shard the hash table's optimiser by rows -- reduce-scatter of the bf16 gradient, the owning rank
    # runs Adam on its 1/N of the rows, all-gather of the bf16 shadow the gather reads (same wire bytes as the all-reduce,
    # 1/N of the Adam traffic per rank); the f32 master and the moments are gathered for checkpoints only
    shard_table_optimizer: bool = False
    # replay the step from captured hipGraphs (graph F: render / eager guidance / graph B: backward + optimiser); the
    # first steps, and the step after every change of the sample budget, run eagerly.  Any number of views per rank (a
    # step's views are rendered as one batch).
    graph_step: bool = True
    # a guidance object that runs on the device in the renderer's own layout (train_step_image: the synthetic one, ONE HIP
    # launch, counter-based noise) is called that way -- eager steps and captured steps alike -- and sits INSIDE the step
    # graph: one graph launch per step.  False: the reference's call shape `train_step(text_z, latents [B,C,H,W])`
    # (what a real diffusion model gets): graph F / eager guidance / graph B
    graph_guidance: bool = True
    # data parallel on RCCL: the gradient exchange (per-group sums + all-reduces, flat bucket) and the optimiser are
    # captured INTO the step graph (RCCL's collectives are capturable; one graph launch per step on every rank, no eager
    # launches between the ranks' graphs).  "false", or a backend that stages through the host (gloo): graph / eager
    # exchange + optimiser.  "auto" (default): captured on a communicator of ONE rank (LNERF_FORCE_DIST, where tests pin
    # captured == eager bit for bit), NOT with more than one rank -- no recorded multi-rank run has shown the two forms
    # equal yet (parity unpinned at N > 1); "true" / LNERF_GRAPH_COLLECTIVES=1 opt in (bench.py does so after a
    # supervised pre-flight of exactly that comparison on the job's own ranks)
    graph_collectives: str = "auto"


@dataclass
class LogConfig:
    exp_name: str = "default"
    exp_root: Path = Path("experiments/")
    save_interval: int = 100
    eval_only: bool = False
    eval_size: int = 10
    full_eval_size: int = 100
    save_mesh: bool = False
    max_keep_ckpts: int = 2
    # no progress lines on stdout (log.txt in the experiment directory is still written)
    quiet: bool = False
    # evaluation renders go through the guidance model's decoder (vae.decode) instead of the linear latent->RGB preview
    decode_eval: bool = False

    @property
    def exp_dir(self) -> Path:
        return Path(self.exp_root) / self.exp_name


@dataclass
class TrainConfig:
    log: LogConfig = field(default_factory=LogConfig)
    render: RenderConfig = field(default_factory=RenderConfig)
    optim: OptimConfig = field(default_factory=OptimConfig)
    guide: GuideConfig = field(default_factory=GuideConfig)

    def __post_init__(self):
        if self.log.eval_only and (self.optim.ckpt is None and not self.optim.resume):
            self.optim.resume = True  # same rule as src/latent_paint/configs/train_config.py:94-97
        # Precision follows optim.fp16 ONLY where the user left render.mlp_precision / render.table_dtype on "auto":
        # the set of such fields is fixed the first time (and edited by note_explicit), so that re-running this after
        # `--optim.fp16 false` or after an explicit `--render.mlp_precision f32` gives the f32 parity path.
        if not hasattr(self, "_auto_precision"):
            self._auto_precision = {n for n in ("mlp_precision", "table_dtype") if getattr(self.render, n) == "auto"}
        for n in self._auto_precision:
            setattr(self.render, n, "bf16" if self.optim.fp16 else "f32")
        for n in ("mlp_precision", "table_dtype"):
            self.render.precision(n)  # validates
        # the table layout follows the table's dtype where the user left render.gridtype on "auto": the blocked layout
        # with the bf16 shadow (a block = one 64-byte line of 4-byte rows), Instant-NGP's vertex hash with the f32 table
        if not hasattr(self, "_auto_layout"):
            self._auto_layout = self.render.gridtype == "auto"
        if self._auto_layout:
            self.render.gridtype = "blocked" if self.render.table_dtype == "bf16" else "hash"
        self.render.layout()  # validates

    def note_explicit(self, key, value):
        """config_cli.apply_overrides tells us which fields the user set."""
        section, _, name = key.partition(".")
        if section == "render" and name == "gridtype":
            self._auto_layout = value == "auto"
        if section == "render" and name in ("mlp_precision", "table_dtype"):
            if not hasattr(self, "_auto_precision"):
                self._auto_precision = set()
            (self._auto_precision.add if value == "auto" else self._auto_precision.discard)(name)


def apply_overrides(cfg: TrainConfig, flat: dict) -> TrainConfig:
    """flat: {'log.exp_name': 'x', 'render.nerf_type': 'latent', ...}"""
    return _cli.apply_overrides(cfg, flat)


def load_config(argv=None) -> TrainConfig:
    """`--config_path file.yaml` and/or dotted flags `--section.field value`."""
    return _cli.load_config(TrainConfig, argv)
