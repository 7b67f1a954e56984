"""Config tree of the Latent-Paint path.  The field names, types and defaults are the reference's CLI / YAML contract
(src/latent_paint/configs/train_config.py:7-97; demo_configs/latent_paint/goldfish.yaml); here every section is a
table of (field, type, default, help) rows turned into a dataclass by `config_cli.make_section`.  Differences from the
reference, all deliberate: `guide.texture_resolution` is a real (typed) field and can be set from the command line
(in the reference it is a bare class attribute, :41); `guide.text` / `guide.shape_path` /
`log.exp_name` default to "" so that the tree can be built without pyrallis and are checked by `validate()`;
`guide.guidance` selects the offline stand-in for the diffusion model."""
from dataclasses import dataclass, field
from pathlib import Path
from typing import Optional, Tuple

from ... import config_cli as _cli

INTERPOLATION_MODES = ("nearest", "bilinear", "bicubic")

RenderConfig = _cli.make_section("RenderConfig", (
    ("train_grid_size", int, 64, "side of the square training render, in latent pixels"),
    ("eval_grid_size", int, 512, "side of the square evaluation render, in decoded RGB pixels"),
    ("radius_range", Tuple[float, float], (1.0, 1.5), "camera distance is drawn uniformly from this range"),
    ("angle_overhead", float, 30, "elevations in [0, angle_overhead] degrees count as the overhead view bucket"),
    ("angle_front", float, 70, "azimuths within +-angle_front degrees count as the front view bucket"),
    ("backbone", str, "texture-mesh", "'texture-mesh' (latent texture) or 'texture-rgb-mesh' (RGB fine-tuning)"),
), doc="mesh renderer")

GuideConfig = _cli.make_section("GuideConfig", (
    ("text", str, "", "prompt"),
    ("shape_path", str, "", "mesh to paint (.obj / .off)"),
    ("append_direction", bool, True, "append the view bucket (front / side / back / overhead) to the prompt"),
    ("concept_name", Optional[str], None, "Textual-Inversion concept"),
    ("diffusion_name", str, "CompVis/stable-diffusion-v1-4", "diffusion checkpoint (name or local directory)"),
    ("shape_scale", float, 0.6, "size of the mesh inside the unit cube"),
    ("dy", float, 0.25, "lift of the mesh along +y"),
    ("texture_resolution", int, 128, "side of the square latent texture"),
    ("texture_interpolation_mode", str, "nearest", " | ".join(INTERPOLATION_MODES)),
    ("guidance", str, "synthetic", "'synthetic' (seeded offline stand-in) or 'stable-diffusion' (diffusers adapter)"),
), doc="guidance")

OptimConfig = _cli.make_section("OptimConfig", (
    ("seed", int, 0, "experiment seed"),
    ("iters", int, 5000, "optimisation steps"),
    ("lr", float, 1e-2, "Adam learning rate"),
    ("resume", bool, False, "continue from the experiment's latest checkpoint"),
    ("ckpt", Optional[str], None, "explicit checkpoint to load"),
), doc="optimisation")

LogConfig = _cli.make_section("LogConfig", (
    ("exp_name", str, "", "experiment name (directory under exp_root)"),
    ("exp_root", Path, Path("experiments/"), "where experiments live"),
    ("save_interval", int, 100, "steps between checkpoints / evaluation renders"),
    ("eval_only", bool, False, "no training: load a checkpoint and run the full evaluation"),
    ("eval_size", int, 10, "evaluation views during training"),
    ("full_eval_size", int, 100, "evaluation views of the final pass"),
    ("save_mesh", bool, True, "export the textured mesh with the final evaluation"),
    ("max_keep_ckpts", int, 2, "older checkpoints are deleted beyond this count"),
), namespace={"exp_dir": property(lambda self: Path(self.exp_root) / self.exp_name)}, doc="logging and saving")


@dataclass
class TrainConfig:
    log: LogConfig = field(default_factory=LogConfig)
    render: RenderConfig = field(default_factory=RenderConfig)
    optim: OptimConfig = field(default_factory=OptimConfig)
    guide: GuideConfig = field(default_factory=GuideConfig)

    def __post_init__(self):
        # evaluation needs weights: without an explicit checkpoint take the experiment's latest one (:94-97)
        wants_weights = self.log.eval_only and self.optim.ckpt is None
        if wants_weights and not self.optim.resume:
            self.optim.resume = True

    def validate(self):
        unset = [key for key, value in (("log.exp_name", self.log.exp_name), ("guide.shape_path", self.guide.shape_path))
                 if not value]
        if unset:
            raise ValueError("required config fields not set: %s" % ", ".join(unset))
        if self.guide.texture_interpolation_mode not in INTERPOLATION_MODES:
            raise ValueError("guide.texture_interpolation_mode must be one of %s" % ", ".join(INTERPOLATION_MODES))
        return self


def apply_overrides(cfg: TrainConfig, flat: dict) -> TrainConfig:
    return _cli.apply_overrides(cfg, flat)


def load_config(argv=None) -> TrainConfig:
    return _cli.load_config(TrainConfig, argv).validate()
