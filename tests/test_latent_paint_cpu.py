"""Host-side pieces of the Latent-Paint path and the mesh fixtures, no GPU: the OBJ/OFF readers against the facts
SURVEY.md Appendix C records for the reference's own meshes (tests/golden/shapes/*.obj are those files, committed as
data), the config surface of src/latent_paint/configs/train_config.py, the view sampler of
src/latent_paint/training/views_dataset.py and the built-in UV atlas."""
import math
import os

import numpy as np
import pytest
import torch

from src.latent_nerf.training.shape import load_obj, make_icosphere, normalize_mesh
from src.latent_paint.configs.train_config import TrainConfig, apply_overrides, load_config
from src.latent_paint.models.mesh import Mesh, read_off
from src.latent_paint.models.textured_mesh import per_triangle_atlas
from src.latent_paint.training.views_dataset import ViewsDataset, circle_poses, rand_poses

SHAPES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shapes")


def test_reference_meshes_parse_to_the_surveyed_counts():
    teddy = Mesh(os.path.join(SHAPES, "teddy.obj"))
    assert teddy.vertices.shape == (2892, 3) and teddy.faces.shape == (5760, 3) and teddy.vt.shape == (1118, 2)
    assert int((teddy.ft.min(dim=1)[0] >= 0).sum()) == 1920          # UVs on a third of the faces only
    assert int(teddy.ft.min()) == -1
    assert torch.allclose(teddy.vertices.min(0)[0], torch.tensor([-4.016, -1.209, -2.486]), atol=2e-3)
    assert torch.allclose(teddy.vertices.max(0)[0], torch.tensor([4.297, 10.09, 4.094]), atol=2e-3)
    blub = Mesh(os.path.join(SHAPES, "blub.obj"))
    assert blub.vertices.shape == (7106, 3) and blub.vt.shape == (7317, 2) and blub.faces.shape == (14208, 3)
    assert int(blub.ft.min()) >= 0 and int(blub.ft.max()) == 7316 and int(blub.faces.max()) == 7105
    assert torch.allclose(blub.vertices.min(0)[0], torch.tensor([-0.711, -0.669, -1.911]), atol=2e-3)
    env = Mesh(os.path.join(SHAPES, "env_sphere.obj"))
    assert env.vertices.shape == (2562, 3) and env.faces.shape == (5120, 3) and env.vt is None and env.ft is None
    assert float((env.vertices.norm(dim=1) - 20.0).abs().max()) < 1e-3
    # the generated stand-in (used when the working directory has no shapes/env_sphere.obj) is the same solid
    ev, ef = make_icosphere(4, 20.0)
    assert ev.shape == env.vertices.shape and ef.shape == env.faces.shape
    # the NeRF path's reader (sketch-shape guidance) agrees with the Latent-Paint one
    v2, f2 = load_obj(os.path.join(SHAPES, "teddy.obj"))
    assert torch.equal(v2, teddy.vertices) and torch.equal(f2, teddy.faces)


def test_normalize_mesh_matches_reference_definition():
    """centre on the vertex mean, farthest vertex at `target_scale`, lift by dy (src/latent_paint/models/mesh.py:37-48)."""
    m = Mesh(os.path.join(SHAPES, "blub.obj"))
    n = m.normalize_mesh(inplace=False, target_scale=0.6, dy=0.25)
    assert n is not m and torch.equal(m.vertices, Mesh(os.path.join(SHAPES, "blub.obj")).vertices)
    c = n.vertices - torch.tensor([0.0, 0.25, 0.0])
    assert abs(float(c.norm(dim=1).max()) - 0.6) < 1e-6 and float(c.mean(0).abs().max()) < 1e-6
    v = m.vertices
    want = (v - v.mean(0)) / (v - v.mean(0)).norm(dim=1).max() * 0.6
    want[:, 1] += 0.25
    assert torch.equal(n.vertices, want)
    assert torch.equal(normalize_mesh(v, 0.6, 0.25), want)           # the NeRF path's helper: same arithmetic
    m.normalize_mesh(inplace=True, target_scale=0.6, dy=0.25)
    assert torch.equal(m.vertices, want)


def test_off_reader(tmp_path):
    p = tmp_path / "quad.off"
    p.write_text("OFF\n5 2 0\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n0.5 0.5 1\n4 0 1 2 3\n3 0 1 4\n")
    v, f, vt, ft = read_off(str(p))
    assert v.shape == (5, 3) and vt is None and f.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 4]]
    assert Mesh(str(p)).faces.shape == (3, 3)
    with pytest.raises(ValueError):
        Mesh(str(tmp_path / "mesh.ply"))


def test_per_triangle_atlas_is_a_valid_uv_map():
    for F in (1, 2, 11, 5760):
        vt, ft = per_triangle_atlas(F, "cpu")
        assert vt.shape == (3 * F, 2) and ft.shape == (F, 3) and ft.flatten().tolist() == list(range(3 * F))
        assert float(vt.min()) > 0 and float(vt.max()) < 1
        n = int(math.ceil(math.sqrt((F + 1) // 2)))
        tri = vt[ft] * n                                            # [F,3,2] in cell units
        cell = torch.floor(tri)
        assert bool((cell == cell[:, :1]).all())                    # a chart never leaves its cell
        area = 0.5 * ((tri[:, 1, 0] - tri[:, 0, 0]) * (tri[:, 2, 1] - tri[:, 0, 1])
                      - (tri[:, 2, 0] - tri[:, 0, 0]) * (tri[:, 1, 1] - tri[:, 0, 1])).abs()
        assert float(area.min()) > 0.2
        # the two charts of a cell do not overlap: lower ones stay below the diagonal x + y = 1, upper ones above
        loc = tri - cell
        s = loc.sum(-1)
        assert bool((s[0::2] < 1.0).all()) and (F < 2 or bool((s[1::2] > 1.0).all()))


def test_latent_paint_config_surface(tmp_path):
    cfg = TrainConfig()
    assert (cfg.render.train_grid_size, cfg.render.eval_grid_size, cfg.render.backbone) == (64, 512, "texture-mesh")
    assert cfg.render.radius_range == (1.0, 1.5) and cfg.render.angle_overhead == 30 and cfg.render.angle_front == 70
    g = cfg.guide
    assert (g.shape_scale, g.dy, g.texture_resolution, g.texture_interpolation_mode) == (0.6, 0.25, 128, "nearest")
    assert g.append_direction and g.diffusion_name == "CompVis/stable-diffusion-v1-4"
    assert (cfg.optim.seed, cfg.optim.iters, cfg.optim.lr) == (0, 5000, 1e-2)
    lg = cfg.log
    assert (lg.save_interval, lg.eval_size, lg.full_eval_size, lg.save_mesh, lg.max_keep_ckpts) == (100, 10, 100, True, 2)
    # the reference's own demo config (demo_configs/latent_paint/goldfish.yaml), restated as data
    y = tmp_path / "goldfish.yaml"
    y.write_text("log:\n  exp_name: 'goldfish'\nguide:\n  text: 'a goldfish'\n  shape_path: shapes/blub.obj\n"
                 "render:\n  backbone: 'texture-mesh'\n")
    cfg = load_config(["--config_path", str(y), "--guide.texture_resolution", "512", "--log.eval_only", "true"])
    assert cfg.log.exp_name == "goldfish" and cfg.guide.shape_path == "shapes/blub.obj"
    assert cfg.guide.texture_resolution == 512            # settable from the CLI (annotated here, not in the reference)
    assert cfg.log.eval_only and cfg.optim.resume and cfg.log.exp_dir.name == "goldfish"
    with pytest.raises(ValueError):
        load_config(["--guide.text", "x"])                 # exp_name / shape_path are required
    with pytest.raises(KeyError):
        apply_overrides(TrainConfig(), {"guide.no_such_field": 1})


def test_views_dataset_distribution_and_circle():
    cfg = TrainConfig().render
    g = torch.Generator().manual_seed(0)
    th, ph, rad, dirs = [], [], [], []
    for _ in range(2000):
        d, t, p, r = rand_poses(1, "cpu", radius_range=cfg.radius_range, angle_overhead=cfg.angle_overhead,
                                angle_front=cfg.angle_front, generator=g)
        th.append(t); ph.append(p); rad.append(r); dirs.append(int(d[0]))
    th, ph, rad = np.array(th), np.array(ph), np.array(rad)
    assert 0 <= th.min() and th.max() <= np.deg2rad(150) and abs(th.mean() - np.deg2rad(75)) < 0.05
    assert 0 <= ph.min() and ph.max() < 2 * np.pi and abs(ph.mean() - np.pi) < 0.12
    assert 1.0 <= rad.min() and rad.max() <= 1.5 and abs(rad.mean() - 1.25) < 0.02
    assert set(dirs) <= {0, 1, 2, 3, 4, 5} and {0, 1, 2, 3} <= set(dirs)
    d, t, p, r = circle_poses("cpu", radius=1.8, theta=60, phi=90)
    assert abs(t - math.radians(60)) < 1e-6 and abs(p - math.radians(90)) < 1e-6 and r == 1.8 and int(d[0]) == 3
    val = ViewsDataset(cfg, "cpu", "val", 4).dataloader()
    views = list(val)
    assert len(views) == 4 and [round(math.degrees(v["phi"])) for v in views] == [0, 90, 180, 270]
    assert all(abs(v["radius"] - 1.8) < 1e-9 and set(v) == {"dir", "theta", "phi", "radius"} for v in views)
    train = ViewsDataset(cfg, "cpu", "train", 100, seed=1).dataloader()
    assert len(list(train)) == 100 and train._data.training
