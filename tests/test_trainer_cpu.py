"""Host-side trainer plumbing that needs no GPU: config surface (YAML + dotted flags as the reference's
CLI takes them), pose sampling distribution, guidance contract, sparsity loss."""
import math
import os

import numpy as np
import pytest
import torch

from src.latent_nerf.configs.train_config import TrainConfig, apply_overrides, load_config
from src.latent_nerf.models.nerf_utils import NeRFType
from src.latent_nerf.training.guidance import SyntheticGuidance, sparsity_loss


def test_config_from_reference_style_inputs(tmp_path):
    # the reference's own demo config (demo_configs/latent_nerf/lego_man.yaml:1-10), restated as data
    y = tmp_path / "lego_man.yaml"
    y.write_text("log:\n  exp_name: 'lego_man'\nguide:\n  text: 'a lego man'\n  shape_path: shapes/teddy.obj\n"
                 "optim:\n  iters: 5000\n  seed: 10\nrender:\n  nerf_type: 'latent'\n")
    cfg = load_config(["--config_path", str(y), "--optim.lr", "0.002", "--render.radius_range", "1.0,1.4",
                       "--log.eval_only", "true"])
    assert cfg.log.exp_name == "lego_man" and cfg.guide.text == "a lego man"
    assert cfg.guide.shape_path == "shapes/teddy.obj" and cfg.optim.seed == 10 and cfg.optim.iters == 5000
    assert cfg.render.nerf_type == NeRFType.latent and cfg.optim.lr == 0.002
    assert cfg.render.radius_range == (1.0, 1.4)
    assert cfg.log.eval_only and cfg.optim.resume  # eval_only without ckpt forces resume (train_config.py:94-97)
    assert cfg.log.exp_dir.name == "lego_man"
    with pytest.raises(KeyError):
        apply_overrides(TrainConfig(), {"render.no_such_field": 1})


def test_default_config_matches_advertised_flags():
    cfg = TrainConfig()
    r = cfg.render
    assert (r.train_h, r.train_w, r.max_steps, r.update_extra_interval, r.max_ray_batch) == (64, 64, 1024, 16, 4096)
    assert r.radius_range == (1.0, 1.5) and r.bound == 1.0 and r.dt_gamma == 0.0 and r.grid_size == 128
    assert cfg.guide.mesh_scale == 0.7 and cfg.optim.iters == 5000 and cfg.log.save_interval == 100
    assert cfg.log.max_keep_ckpts == 2 and cfg.log.eval_size == 10 and cfg.log.full_eval_size == 100


def test_synthetic_guidance_contract():
    g = SyntheticGuidance(torch.device("cpu"), channels=4, size=16, seed=0)
    z = g.get_text_embeds("a lego man, front view")
    lat = torch.zeros(2, 4, 16, 16)
    grad = g.train_step(z, lat, dirs=torch.tensor([0, 3]))
    assert grad.shape == lat.shape and not grad.requires_grad  # same shape as the latents, no autograd
    # following the negative gradient reduces the distance to the view's target
    t = g.targets[torch.tensor([0, 3])]
    d0 = (lat - t).pow(2).mean()
    for _ in range(300):
        lat = lat - 0.5 * g.train_step(z, lat, dirs=torch.tensor([0, 3]))
    assert (lat - t).pow(2).mean() < 0.2 * d0


def test_sparsity_loss_prefers_binary_opacity():
    assert sparsity_loss(torch.tensor([0.0, 1.0, 1.0, 0.0])) < 1e-3
    assert abs(float(sparsity_loss(torch.tensor([0.5, 0.5]))) - 1.0) < 1e-6


def test_precision_follows_fp16_only_where_left_on_auto():
    """`optim.fp16` picks bf16 for the table shadow and the MLP unless the user set `render.mlp_precision` /
    `render.table_dtype` explicitly -- whatever the order of the flags (the f32 parity path must be reachable)."""
    from src.latent_nerf.configs.render_config import RenderConfig
    assert (TrainConfig().render.mlp_precision, TrainConfig().render.table_dtype) == ("bf16", "bf16")
    c = load_config(["--optim.fp16", "false"])
    assert (c.optim.fp16, c.render.mlp_precision, c.render.table_dtype) == (False, "f32", "f32")
    for argv in (["--render.mlp_precision", "f32", "--render.table_dtype", "f32"],
                 ["--render.table_dtype", "f32", "--optim.fp16", "true", "--render.mlp_precision", "f32"]):
        c = load_config(argv)
        assert (c.optim.fp16, c.render.mlp_precision, c.render.table_dtype) == (True, "f32", "f32")
    c = load_config(["--optim.fp16", "true", "--render.mlp_precision", "f32"])
    assert (c.render.mlp_precision, c.render.table_dtype) == ("f32", "bf16")
    c = load_config(["--optim.fp16", "false", "--render.mlp_precision", "bf16"])
    assert (c.render.mlp_precision, c.render.table_dtype) == ("bf16", "f32")
    c = apply_overrides(load_config(["--optim.fp16", "false"]), {"optim.fp16": True})   # re-resolved after a later change
    assert (c.render.mlp_precision, c.render.table_dtype) == ("bf16", "bf16")
    r = RenderConfig()                        # a bare RenderConfig means the f32 parity path
    assert r.mlp_precision == "auto" and r.precision("mlp_precision") == "f32" and r.precision("table_dtype") == "f32"
    with pytest.raises(ValueError):
        load_config(["--render.mlp_precision", "fp8"])
    # the table LAYOUT follows the table's dtype the same way: blocked (64-byte-line blocks of 4-byte rows) with the
    # bf16 shadow, Instant-NGP's vertex hash with the f32 table; an explicit render.gridtype always wins
    assert TrainConfig().render.gridtype == "blocked" and r.gridtype == "auto" and r.layout() == "hash"
    assert load_config(["--optim.fp16", "false"]).render.gridtype == "hash"
    assert load_config(["--render.table_dtype", "f32"]).render.gridtype == "hash"
    assert load_config(["--render.gridtype", "hash"]).render.gridtype == "hash"
    assert load_config(["--render.gridtype", "tiled", "--optim.fp16", "false"]).render.gridtype == "tiled"
    c = apply_overrides(load_config(["--optim.fp16", "false"]), {"optim.fp16": True})
    assert c.render.gridtype == "blocked"
    with pytest.raises(ValueError):
        load_config(["--render.gridtype", "morton"])


def test_write_video_writes_every_frame(tmp_path):
    """mp4 through imageio where it exists, else an animated GIF through Pillow: either way one file, all frames."""
    import numpy as np
    from src.utils import write_video
    frames = [np.full((8, 8, 3), 40 * i, dtype=np.uint8) for i in range(5)]
    out = write_video(tmp_path / "step_00001_rgb", frames)
    assert out.exists() and out.stat().st_size > 0 and out.suffix in (".mp4", ".gif")
    if out.suffix == ".gif":
        from PIL import Image
        assert Image.open(out).n_frames == 5
