"""Entry point with the reference's surface (scripts/train_latent_paint.py:1-17 of the reference):

    python -m scripts.train_latent_paint --config_path demo_configs/latent_paint/goldfish.yaml
    python -m scripts.train_latent_paint --log.exp_name goldfish --guide.text "a goldfish" \
        --guide.shape_path shapes/blub.obj --guide.texture_resolution 512

`log.eval_only` runs the final evaluation (circle renders + mesh export) instead of training.  With pyrallis
installed the reference's decorator form works on these dataclasses too; here the same two input forms are
parsed by src.config_cli."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-nerf-test_amd"))

from src.latent_paint.configs.train_config import TrainConfig, load_config  # noqa: E402
from src.latent_paint.training.trainer import Trainer  # noqa: E402


def main(cfg: TrainConfig):
    trainer = Trainer(cfg)
    if cfg.log.eval_only:
        trainer.full_eval()
    else:
        trainer.train()


if __name__ == "__main__":
    main(load_config())
