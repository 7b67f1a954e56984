"""Config tree of the Latent-Paint path: the field names and defaults are the reference's CLI/YAML contract
(src/latent_paint/configs/train_config.py:7-97; demo_configs/latent_paint/goldfish.yaml).  Differences, all
deliberate: `guide.texture_resolution` is annotated, so it IS a dataclass field and can be set from the command
line (in the reference it is a bare class attribute, :41); `guide.text` / `guide.shape_path`
/ `log.exp_name` get empty defaults so that the tree can be default-constructed without pyrallis and are checked
by `validate()`; `guide.guidance` selects the offline stand-in for the diffusion model."""
from dataclasses import dataclass, field
from pathlib import Path
from typing import Optional, Tuple

from ... import config_cli as _cli


@dataclass
class RenderConfig:
    # side of the square training render (latent pixels)
    train_grid_size: int = 64
    # side of the square evaluation render (decoded RGB pixels)
    eval_grid_size: int = 512
    radius_range: Tuple[float, float] = (1.0, 1.5)
    # [0, angle_overhead] counts as the overhead view bucket
    angle_overhead: float = 30
    angle_front: float = 70
    # 'texture-mesh' (latent texture) or 'texture-rgb-mesh' (RGB fine-tuning of a trained latent texture)
    backbone: str = "texture-mesh"


@dataclass
class GuideConfig:
    text: str = ""
    # mesh to paint (.obj / .off)
    shape_path: str = ""
    append_direction: bool = True
    concept_name: Optional[str] = None
    diffusion_name: str = "CompVis/stable-diffusion-v1-4"
    # mesh size inside the unit cube, and its lift along +y
    shape_scale: float = 0.6
    dy: float = 0.25
    texture_resolution: int = 128
    # 'nearest' | 'bilinear' | 'bicubic'
    texture_interpolation_mode: str = "nearest"
    # "synthetic": seeded stand-in for the diffusion model (offline); "stable-diffusion": diffusers adapter
    guidance: str = "synthetic"


@dataclass
class OptimConfig:
    seed: int = 0
    iters: int = 5000
    lr: float = 1e-2
    resume: bool = False
    ckpt: Optional[str] = None


@dataclass
class LogConfig:
    exp_name: str = ""
    exp_root: Path = Path("experiments/")
    save_interval: int = 100
    eval_only: bool = False
    eval_size: int = 10
    full_eval_size: int = 100
    save_mesh: bool = True
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
        # evaluation needs weights: without an explicit checkpoint take the experiment's latest one (:94-97)
        if self.log.eval_only and (self.optim.ckpt is None and not self.optim.resume):
            self.optim.resume = True

    def validate(self):
        missing = [n for n, v in (("log.exp_name", self.log.exp_name), ("guide.shape_path", self.guide.shape_path))
                   if not v]
        if missing:
            raise ValueError("required config fields not set: %s" % ", ".join(missing))
        if self.guide.texture_interpolation_mode not in ("nearest", "bilinear", "bicubic"):
            raise ValueError("guide.texture_interpolation_mode must be nearest, bilinear or bicubic")
        return self


def apply_overrides(cfg: TrainConfig, flat: dict) -> TrainConfig:
    return _cli.apply_overrides(cfg, flat)


def load_config(argv=None) -> TrainConfig:
    return _cli.load_config(TrainConfig, argv).validate()
