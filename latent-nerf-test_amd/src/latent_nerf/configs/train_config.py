"""Training configuration tree `log / render / optim / guide`, the surface the reference's CLI drives
(`python -m scripts.train_latent_nerf --config_path demo_configs/latent_nerf/lego_man.yaml`, or dotted flags
such as `--log.exp_name x --guide.text "..." --render.nerf_type latent`, README.md:64-69,92-97).

Field names and defaults follow the reference's present Latent-Paint config
(src/latent_paint/configs/train_config.py:7-97) and the NeRF flags its demo configs and README still
advertise for the absent package (demo_configs/latent_nerf/lego_man.yaml:1-10, README.md:140-142:
guide.shape_path, guide.mesh_scale, guide.proximal_surface, optim.lambda_shape, optim.seed, optim.iters).
pyrallis is not installed here, so `load_config()` implements the same two input forms on top of
argparse + yaml; when pyrallis is importable the dataclasses work with `pyrallis.wrap()` unchanged."""
import argparse
import dataclasses
from dataclasses import dataclass, field
from pathlib import Path
from typing import Optional, Tuple, get_type_hints

import yaml

from ..models.nerf_utils import NeRFType
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
        if self.optim.fp16:
            self.render.mlp_precision = "bf16"
            self.render.table_dtype = "bf16"


def _coerce(value, typ):
    if typ is bool:
        return value if isinstance(value, bool) else str(value).lower() in ("1", "true", "yes", "y")
    if typ is NeRFType:
        return value if isinstance(value, NeRFType) else NeRFType(str(value))
    if typ is Path:
        return Path(value)
    origin = getattr(typ, "__origin__", None)
    if origin is tuple:
        if isinstance(value, str):
            value = [v for v in value.replace("(", "").replace(")", "").split(",") if v.strip()]
        return tuple(float(v) for v in value)
    if typ in (int, float, str):
        return typ(value)
    args = getattr(typ, "__args__", ())
    if type(None) in args:  # Optional[X]
        if value is None or str(value).lower() in ("none", "null"):
            return None
        return _coerce(value, [a for a in args if a is not type(None)][0])
    return value


def apply_overrides(cfg: TrainConfig, flat: dict) -> TrainConfig:
    """flat: {'log.exp_name': 'x', 'render.nerf_type': 'latent', ...}"""
    for key, value in flat.items():
        section, _, name = key.partition(".")
        sub = getattr(cfg, section, None)
        if sub is None or not dataclasses.is_dataclass(sub) or name not in {f.name for f in dataclasses.fields(sub)}:
            raise KeyError("unknown config field %r" % key)
        setattr(sub, name, _coerce(value, get_type_hints(type(sub))[name]))
    cfg.__post_init__()
    return cfg


def load_config(argv=None) -> TrainConfig:
    """`--config_path file.yaml` and/or dotted flags `--section.field value`."""
    ap = argparse.ArgumentParser(add_help=True)
    ap.add_argument("--config_path", default=None)
    args, rest = ap.parse_known_args(argv)
    flat = {}
    if args.config_path:
        doc = yaml.safe_load(open(args.config_path)) or {}
        for section, body in doc.items():
            for name, value in (body or {}).items():
                flat["%s.%s" % (section, name)] = value
    i = 0
    while i < len(rest):
        tok = rest[i]
        if not tok.startswith("--"):
            raise SystemExit("unexpected argument %r" % tok)
        if "=" in tok:
            k, v = tok[2:].split("=", 1)
            i += 1
        else:
            k = tok[2:]
            if i + 1 >= len(rest):
                raise SystemExit("flag %r needs a value" % tok)
            v = rest[i + 1]
            i += 2
        flat[k] = v
    return apply_overrides(TrainConfig(), flat)
