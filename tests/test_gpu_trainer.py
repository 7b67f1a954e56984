"""The training loop on the HIP path actually learns: with the seeded synthetic guidance (a denoising-style
target per view bucket) the rendered latents move toward the target; checkpoints round-trip with the
reference's schema (src/latent_paint/training/trainer.py:288-310)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(built_lib):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    return torch.device("cuda:0")


def _cfg(tmp_path, **over):
    from src.latent_nerf.configs.train_config import TrainConfig, apply_overrides
    flat = {"log.exp_name": "t", "log.exp_root": str(tmp_path), "render.train_h": 32, "render.train_w": 32,
            "render.eval_h": 32, "render.eval_w": 32, "render.grid_size": 64, "optim.iters": 60, "optim.lr": 5e-3,
            "log.save_interval": 30, "log.eval_size": 2, "log.full_eval_size": 3, "optim.fp16": False,
            "guide.text": "a lego man"}
    flat.update(over)
    return apply_overrides(TrainConfig(), flat)


def test_training_reduces_guidance_error_and_checkpoints_roundtrip(dev, tmp_path):
    from src.latent_nerf.training.trainer import Trainer
    cfg = _cfg(tmp_path)
    tr = Trainer(cfg, device=dev)
    ds = tr.dataloaders["val"]

    def err():
        tr.nerf.eval()
        data = ds.collate(0)
        pred, _ = tr.eval_render(data)
        tgt = tr.diffusion.targets[int(data["dir"][0])][None]
        tgt = torch.nn.functional.interpolate(tgt, size=pred.shape[-2:], mode="bilinear", align_corners=False)
        return float((pred - tgt).pow(2).mean())

    tr.nerf.update_extra_state()
    e0 = err()
    tr.train()
    e1 = err()
    assert tr.train_step == 60
    assert e1 < 0.7 * e0, (e0, e1)
    assert int(tr.nerf.density_bitfield.count_nonzero()) > 0 and tr.nerf.iter_density >= 4
    ck = sorted(tr.ckpt_path.glob("*.pth"))
    assert [c.name for c in ck] == ["step_000030.pth", "step_000060.pth"]
    state = torch.load(ck[-1], map_location="cpu", weights_only=True)
    assert set(state) == {"train_step", "checkpoints", "model", "optimizer", "table_layout"} and state["train_step"] == 60
    assert state["table_layout"]["gridtype"] == "hash" and len(state["table_layout"]["offsets"]) == 17
    # resume: same weights, step counter continues at train_step + 1 (reference semantics)
    cfg2 = _cfg(tmp_path, **{"optim.resume": True})
    tr2 = Trainer(cfg2, device=dev)
    assert tr2.train_step == 61
    assert torch.equal(tr2.nerf.encoder.embeddings.detach().cpu(), tr.nerf.encoder.embeddings.detach().cpu())
    assert torch.equal(tr2.nerf.w2.detach().cpu(), tr.nerf.w2.detach().cpu())
    assert torch.equal(tr2.nerf.density_bitfield.cpu(), tr.nerf.density_bitfield.cpu())
    assert tr2.optimizer.step_no == tr.optimizer.step_no
    # max_keep_ckpts = 2: a third checkpoint evicts the oldest
    tr.train_step = 90
    tr.save_checkpoint(full=True)
    assert [c.name for c in sorted(tr.ckpt_path.glob("*.pth"))] == ["step_000060.pth", "step_000090.pth"]
    frames = tr.evaluate(tr.dataloaders["val"], tr.eval_renders_path)
    assert len(frames) == 2 and frames[0].shape == (32, 32, 3) and frames[0].dtype.name == "uint8"
    assert sorted(p.name for p in tr.eval_renders_path.glob("step_00090_*_rgb.png")) == \
        ["step_00090_0000_rgb.png", "step_00090_0001_rgb.png"]
    video = tr.evaluate(tr.dataloaders["val"], tr.final_renders_path, save_as_video=True)   # what full_eval() does
    assert len(video) == 2 and len(list(tr.final_renders_path.glob("step_00090_rgb.*"))) == 1


def test_bf16_training_step_runs(dev, tmp_path):
    from src.latent_nerf.training.trainer import Trainer
    cfg = _cfg(tmp_path, **{"optim.fp16": True, "optim.iters": 20, "log.save_interval": 1000, "log.exp_name": "b"})
    assert cfg.render.mlp_precision == "bf16" and cfg.render.table_dtype == "bf16"
    tr = Trainer(cfg, device=dev)
    tr.train()
    assert tr.train_step == 20 and bool(torch.isfinite(tr.nerf.encoder.embeddings).all())
    assert bool(torch.isfinite(tr.nerf.w1).all())


def test_checkpoint_records_the_table_layout_and_refuses_another(dev, tmp_path):
    """A `blocked` table has the shape of a `hash` table and different contents per row: the checkpoint says which it
    holds, and loading it into a model of the other layout is refused (it would render scrambled features)."""
    from src.latent_nerf.training.trainer import Trainer
    tr = Trainer(_cfg(tmp_path, **{"optim.iters": 2, "log.exp_name": "lay_b", "render.gridtype": "blocked",
                                   "optim.fp16": True, "log.full_eval_size": 1}), device=dev)
    tr.train()
    path = tr.save_checkpoint(full=True)
    other = Trainer(_cfg(tmp_path, **{"optim.iters": 2, "log.exp_name": "lay_h", "optim.fp16": True,
                                      "render.gridtype": "hash"}), device=dev)
    auto = _cfg(tmp_path, **{"optim.fp16": True, "log.exp_name": "lay_a"})
    assert auto.render.gridtype == "blocked"          # bf16 table: the blocked layout is the default ...
    assert _cfg(tmp_path, **{"optim.fp16": False, "log.exp_name": "lay_a2"}).render.gridtype == "hash"   # ... f32: Instant-NGP's
    with pytest.raises(ValueError, match="not interchangeable"):
        other.load_checkpoint(path, model_only=True)
    same = Trainer(_cfg(tmp_path, **{"optim.iters": 2, "log.exp_name": "lay_b2", "render.gridtype": "blocked",
                                     "optim.fp16": True}), device=dev)
    same.load_checkpoint(path, model_only=True)
    assert torch.equal(same.nerf.encoder.embeddings.detach(), tr.nerf.encoder.embeddings.detach())


def test_batched_views_render_like_separate_views(dev):
    """k views handed to the renderer as ONE batch (camera form, B = k): every view's image is bit-identical to that
    view rendered alone (rays, samples, features and MLP outputs are per-ray / per-sample work), and one backward over
    the batch gives the SUM of the separate backward passes: per-sample gradients are the same numbers, the table and
    weight gradients are the same sums in another order (the table's fixed-point sums are exact; what differs is one
    f32 rounding of the total against the sum of k rounded totals)."""
    from src.latent_nerf.configs.render_config import RenderConfig
    from src.latent_nerf.models.network_grid import NeRFNetwork
    from src.latent_nerf.models.nerf_utils import intrinsics_from_fov, pose_from_angles
    torch.manual_seed(3)
    HW, G, k = 32, 64, 3
    cfg = RenderConfig(grid_size=G, train_h=HW, train_w=HW, mlp_precision="f32", table_dtype="f32", noise_seed=None)
    net = NeRFNetwork(cfg, log2_hashmap_size=14)
    net.encoder.embeddings.data.normal_(0, 0.1)
    net = net.to(dev).train()
    net.seed_density_grid(lambda x: (x.norm(dim=-1) < 0.5).float() * 10.0, thresh=0.01)
    # (the last view looks AWAY from the volume: every one of its rays misses the box -- a ragged batch: a view with no
    # sample at all in the middle of the sample region's bookkeeping)
    away = torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 3.0], [0.0, 0.0, 0.0, 1.0]])
    k = k + 1
    poses = torch.stack([pose_from_angles(math.radians(50.0 + 20 * v), math.radians(70.0 * v), 1.2 + 0.1 * v)
                         for v in range(k - 1)] + [away]).to(dev)
    intr = intrinsics_from_fov(55.0, HW, HW)
    bg = torch.rand(k * HW * HW, 4, device=dev)
    g = torch.randn(k, HW * HW, 4, device=dev)
    params = [net.encoder.embeddings, net.w1, net.b1, net.w2, net.b2, net.w3, net.b3]

    def grads():
        out = [p.grad.detach().clone() for p in params]
        for p in params:
            p.grad = None
        return out

    out = net.render(None, None, camera=(poses, intr, HW, HW), bg_color=bg, perturb=False)
    img = out["image"].detach().clone()
    M = int(out["counter"][0])
    assert img.shape == (k, HW * HW, 4) and M > 5000
    out["image"].backward(g)
    gb = grads()
    gs, Ms = None, 0
    for v in range(k):
        o = net.render(None, None, camera=(poses[v:v + 1], intr, HW, HW), bg_color=bg[v * HW * HW:(v + 1) * HW * HW],
                       perturb=False)
        assert torch.equal(o["image"][0], img[v]), v
        Ms += int(o["counter"][0])
        o["image"].backward(g[v:v + 1])
        gv = grads()
        gs = gv if gs is None else [a + b for a, b in zip(gs, gv)]
    assert Ms == M
    assert int(o["counter"][0]) == 0 and torch.equal(img[k - 1], bg[(k - 1) * HW * HW:].reshape(HW * HW, 4))   # pure background
    for a, b in zip(gb, gs):
        scale = float(b.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-5 * scale, (float((a - b).abs().max()), scale)


def test_multi_view_steps_run_fused_and_captured(dev, tmp_path):
    """Two views per step (render.batch_size of the reference's fork,
    /root/reference/src/latent_paint_mesh/configs/train_config.py:32): the step's views are one batch, so the fused table
    update, the closing scatter and the captured step all apply -- no eager accumulate loop.  Graphed == eager, bit for
    bit, as with one view.  And with one view per step the fused table update (default) and the ordinary path give the
    same table, bit for bit (same seeds, same poses)."""
    from src.latent_nerf.training.trainer import Trainer
    tabs = []
    for graph in (True, False):
        cfg = _cfg(tmp_path, **{"optim.views_per_step": 2, "optim.iters": 20, "log.save_interval": 1000, "optim.fp16": True,
                                "log.exp_name": "v2g%d" % graph, "optim.graph_step": graph, "log.full_eval_size": 1})
        torch.manual_seed(7)
        torch.cuda.manual_seed(7)
        tr = Trainer(cfg, device=dev)
        assert tr.optimizer.fused is not None and tr.optimizer.fused.inline_tail and len(tr.views) == 2
        assert abs(tr.optimizer.grad_scale - 0.5) < 1e-12
        before = tr.nerf.encoder.embeddings.detach().clone()
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)
        tr.train()
        assert tr.train_step == 20 and bool(torch.isfinite(tr.nerf.encoder.embeddings).all())
        assert float((tr.nerf.encoder.embeddings.detach() - before).abs().max()) > 0
        if graph:
            assert tr.graph_stats["captures"] >= 1 and tr.graph_stats["replayed_steps"] >= 15, tr.graph_stats
            assert tr._static["poses"].shape == (2, 4, 4)
            assert not torch.equal(tr._static["poses"][0], tr._static["poses"][1])      # two different views
        else:
            assert tr.graph_stats["replayed_steps"] == 0
        tabs.append((tr.nerf.encoder.embeddings.detach().clone(), tr.nerf.w2.detach().clone(), tr.optimizer.step_no))
    assert tabs[0][2] == tabs[1][2] == 20
    assert torch.equal(tabs[0][0], tabs[1][0]) and torch.equal(tabs[0][1], tabs[1][1])
    tabs = []
    for fuse in (True, False):
        c = _cfg(tmp_path, **{"optim.iters": 5, "log.save_interval": 1000, "log.exp_name": "f%d" % fuse,
                              "optim.fuse_table_update": fuse})
        t = Trainer(c, device=dev)
        assert (t.optimizer.fused is not None) == fuse
        t.train()
        tabs.append(t.nerf.encoder.embeddings.detach().clone())
    assert torch.equal(tabs[0], tabs[1])


def test_trainer_recaptures_after_a_foreign_scatter_through_the_shared_workspace(dev, tmp_path):
    """The scatter workspace is one buffer per device; the captured step bakes in "the level maxima are clean".  A
    backward through the net OUTSIDE train() (here: an ordinary, non-closing scatter under the trainer's stream) leaves
    its maxima in the header: the trainer must notice and capture again instead of replaying with stale maxima."""
    from src.latent_nerf.models import encoding as E
    from src.latent_nerf.training.trainer import Trainer
    cfg = _cfg(tmp_path, **{"optim.iters": 8, "log.save_interval": 1000, "optim.fp16": True, "log.exp_name": "fs",
                            "log.full_eval_size": 1})
    tr = Trainer(cfg, device=dev)
    tr.train()
    assert tr.graph_stats["captures"] == 1
    e0 = E.scatter_workspace_epoch(dev)
    assert e0 == tr._gstep_ws and e0[0] is not None          # clean after the closing scatter
    with torch.cuda.stream(tr.stream):
        x = (torch.rand(4096, 3, device=dev) - 0.5)
        s, _ = tr.nerf.field(x.contiguous(), 4096)
        s.sum().backward()                                    # un-armed: ordinary scatter, maxima left behind
        for p in tr.nerf.parameters():
            p.grad = None
    tr.stream.synchronize()
    assert E.scatter_workspace_epoch(dev) != e0 and E.scatter_workspace_epoch(dev)[0] is None
    tr.train(iters=16)
    assert tr.train_step == 16 and tr.graph_stats["captures"] == 2, tr.graph_stats
    assert bool(torch.isfinite(tr.nerf.encoder.embeddings).all())


def test_mesh_winding_distance_and_shape_guidance(dev, tmp_path):
    """csrc/mesh.hip against the float64 oracle on a synthetic closed mesh (5120-face icosphere, the size of
    the reference's shapes/teddy.obj), occupancy seeding from the mesh, and the shape loss driving sigma."""
    import numpy as np
    from oracle import mesh_oracle as MO
    from src.latent_nerf.training import shape as S
    verts, faces = S.make_icosphere(4, 0.6)
    assert faces.shape[0] == 5120
    tris = verts[faces]
    g = torch.Generator().manual_seed(0)
    pts = (torch.rand(300, 3, generator=g) * 2 - 1)
    w = S.mesh_winding_number(pts.to(dev), tris.to(dev)).cpu().numpy()
    d = S.mesh_distance(pts.to(dev), tris.to(dev)).cpu().numpy()
    w_ref = MO.winding_number(pts.numpy(), tris.numpy())
    d_ref = MO.distance(pts.numpy(), tris.numpy())
    assert np.abs(w - w_ref).max() < 2e-3          # f32 sum of 5120 solid angles vs float64
    assert np.abs(d - d_ref).max() < 1e-5
    r = pts.norm(dim=-1).numpy()
    assert np.all(w[r < 0.55] > 0.99) and np.all(np.abs(w[r > 0.65]) < 0.01)   # inside ~1, outside ~0
    assert np.abs(d - np.abs(r - 0.6))[np.abs(r - 0.6) > 0.05].max() < 0.01
    # OBJ round trip through the plain-text reader (quads, negative indices, v/vt/vn tokens)
    obj = tmp_path / "m.obj"
    with open(obj, "w") as f:
        for v in verts.tolist():
            f.write("v %f %f %f\n" % tuple(v))
        for a, b, c in (faces + 1).tolist():
            f.write("f %d//%d %d//%d %d//%d\n" % (a, a, b, b, c, c))
        f.write("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf -4 -3 -2 -1\n")
    v2, f2 = S.load_obj(str(obj))
    assert v2.shape[0] == verts.shape[0] + 4 and f2.shape[0] == faces.shape[0] + 2
    n = S.normalize_mesh(verts * 7 + 3, target_scale=0.7)
    assert abs(float(n.norm(dim=1).max()) - 0.7) < 1e-5 and float(n.mean(0).abs().max()) < 1e-5
    # trainer with guide.shape_path: occupancy seeded from the mesh, shape loss has a gradient
    from src.latent_nerf.training.trainer import Trainer
    obj = tmp_path / "sphere.obj"
    with open(obj, "w") as f:
        for v in verts.tolist():
            f.write("v %f %f %f\n" % tuple(v))
        for a, b, c in (faces + 1).tolist():
            f.write("f %d %d %d\n" % (a, b, c))
    cfg = _cfg(tmp_path, **{"guide.shape_path": str(obj), "optim.lambda_shape": 1e-2, "optim.iters": 5,
                            "log.exp_name": "s", "log.save_interval": 1000, "guide.mesh_scale": 0.6})
    tr = Trainer(cfg, device=dev)
    G = cfg.render.grid_size
    bits = tr.nerf.density_bitfield.cpu()
    frac = float(sum(bin(int(b)).count("1") for b in bits.tolist())) / (G ** 3)
    assert 0.08 < frac < 0.14   # sphere of radius 0.6 in the [-1,1]^3 cube: 4/3 pi 0.6^3 / 8 = 0.113
    tr.nerf.train()
    data = tr.dataloaders["train"].collate(0)
    assert data["rays_o"] is None     # training views hand over the camera; the rays are generated inside the march
    # (on the trainer's stream: autograd pins a parameter's gradient accumulation to the stream of its first backward, and
    # the captured step cannot accumulate on the legacy default stream)
    with torch.cuda.stream(tr.stream):
        out = tr.nerf.render(None, None, camera=data["camera"], perturb=True)
        loss = tr.shape_loss(out["xyzs"], out["sigmas"], out["counter"])
        loss.backward()
    tr.stream.synchronize()
    assert float(loss) > 0 and float(tr.nerf.w3.grad.abs().sum()) > 0
    for p in tr.nerf.parameters():
        p.grad = None
    del out, loss
    tr.train()
    assert tr.train_step == 5


def test_graphed_trainer_matches_eager_trainer_bit_for_bit(dev, tmp_path):
    """`Trainer.train()` with optim.graph_step (graph F: render / eager guidance / graph B: backward + arm + optimiser;
    pose and intrinsics uploaded into static device buffers; occupancy refreshes between replays) against the same
    trainer running every step eagerly: after 20 steps -- two occupancy refreshes, 18 replays -- table, moments, MLP,
    density grid and bitfield are bit-identical, in the f32 and in the bf16 configuration.  (Surface:
    /root/reference/scripts/train_latent_nerf.py:8-14, src/latent_paint/training/trainer.py:113-144.)"""
    from src.latent_nerf.training.trainer import Trainer
    for fp16 in (False, True):
        states = []
        # whole: ONE graph per step (the synthetic guidance is one HIP launch on the render's own layout and sits inside
        # it) against the same guidance called eagerly; split: graph F / eager guidance through the reference's call shape
        # train_step(text_z, latents [B,C,H,W]) / graph B (what a real diffusion model gets) against that loop without
        # graphs.  (The two guidance forms draw their noise from different generators: compared pairwise.)
        for mode, graph, device_guidance in (("whole", True, True), ("eagerdev", False, True), ("split", True, False),
                                             ("eager", False, False)):
            cfg = _cfg(tmp_path, **{"optim.iters": 20, "log.save_interval": 1000, "optim.fp16": fp16,
                                    "log.exp_name": "g%d%s" % (fp16, mode), "optim.graph_step": graph,
                                    "optim.graph_guidance": device_guidance, "log.full_eval_size": 1})
            torch.manual_seed(7)
            torch.cuda.manual_seed(7)
            tr = Trainer(cfg, device=dev)
            torch.manual_seed(11)                 # the guidance's noise / timestep stream
            torch.cuda.manual_seed(11)
            tr.train()
            assert tr.train_step == 20
            if graph:
                assert tr.graph_stats["captures"] >= 1 and tr.graph_stats["replayed_steps"] >= 15, tr.graph_stats
                assert tr._whole == (mode == "whole")
            else:
                assert tr.graph_stats["replayed_steps"] == 0 and tr.graph_stats["eager_steps"] == 20
            opt = tr.optimizer
            states.append({"table": tr.nerf.encoder.embeddings.detach().clone(),
                           "m": opt.big[0][1].clone(), "v": opt.big[0][2].clone(),
                           "w1": tr.nerf.w1.detach().clone(), "w3": tr.nerf.w3.detach().clone(),
                           "b2": tr.nerf.b2.detach().clone(), "grid": tr.nerf.density_grid.clone(),
                           "bits": tr.nerf.density_bitfield.clone(), "step": opt.step_no,
                           "step_dev": int(opt.step_dev[0].item())})
        for a, b in ((states[0], states[1]), (states[2], states[3])):
            assert a["step"] == b["step"] == 20 and a["step_dev"] == b["step_dev"] == 21
            for k in ("table", "m", "v", "w1", "w3", "b2", "grid", "bits"):
                assert torch.equal(a[k], b[k]), (fp16, k, float((a[k].float() - b[k].float()).abs().max()))
        b = states[-1]
        assert float((b["table"] - 0).abs().max()) > 0


def test_graphed_trainer_recaptures_when_the_sample_budget_moves(dev, tmp_path):
    """The refresh re-derives the sample capacity from observed marches; a change means new sample buffers, so the
    trainer drops its graphs, runs one step eagerly and captures again.  Forced here by shrinking the occupied region
    between two stretches of training: the marches of steps 21..32 are small, the refresh of step 33 moves the budget."""
    from src.latent_nerf.training.trainer import Trainer
    cfg = _cfg(tmp_path, **{"optim.iters": 20, "log.save_interval": 1000, "optim.fp16": True, "log.exp_name": "rc",
                            "log.full_eval_size": 1})
    tr = Trainer(cfg, device=dev)
    tr.train()
    assert tr.train_step == 20 and tr.graph_stats["captures"] == 1, tr.graph_stats
    cap0 = tr._gstep_capacity
    with torch.cuda.stream(tr.stream):
        tr.nerf.seed_density_grid(lambda x: (x.norm(dim=-1) < 0.2).float() * 100.0)
        tr.nerf._march.take_peak()      # (forget the peaks of the big marches)
    tr.stream.synchronize()
    tr.train(iters=40)
    assert tr.train_step == 40 and tr.graph_stats["captures"] >= 2, tr.graph_stats
    assert tr._gstep_capacity < cap0, (tr._gstep_capacity, cap0)
    assert tr.graph_stats["replayed_steps"] + tr.graph_stats["eager_steps"] == 40
    assert bool(torch.isfinite(tr.nerf.encoder.embeddings).all())


def test_opacity_entropy_gradient_matches_autograd(dev):
    from src.latent_nerf.training.guidance import sparsity_loss, sparsity_loss_grad
    torch.manual_seed(0)
    ws = torch.rand(4096, device=dev)
    ws[:7] = torch.tensor([0.0, 1.0, 1e-5, 1 - 1e-5, 5e-6, 0.5, 1.5], device=dev)   # clamp edges, outside, centre
    ref = ws.clone().requires_grad_()
    (3e-3 * sparsity_loss(ref)).backward()
    got = sparsity_loss_grad(ws, 3e-3)
    assert float((got - ref.grad).abs().max()) <= 1e-6 * float(ref.grad.abs().max()) + 1e-12
    assert float(got[0]) == 0.0 and float(got[1]) == 0.0 and float(got[6]) == 0.0 and float(got[5]) == 0.0


def test_fused_synthetic_guidance_matches_oracle_and_is_a_counter_function(dev):
    """lnerf_synthetic_guidance (SyntheticGuidance.train_step_image): the guidance gradient and the sparsity gradient of
    one step in ONE launch, in the renderer's image layout.  Against the oracle's restatement (integer hash bit-exact,
    Box-Muller in f32 vs f64: 1e-5), against lnerf_opacity_entropy_grad (bit-exact), and as a FUNCTION of the device
    step counter: same counter -> same values, next counter -> fresh noise and timestep (what a replayed graph relies on)."""
    from oracle import nerf_oracle as O
    from src.latent_nerf.training.guidance import SyntheticGuidance, sparsity_loss_grad
    B, H, W, C = 3, 16, 16, 4
    g = SyntheticGuidance(dev, channels=C, size=H, seed=5)
    torch.manual_seed(0)
    image = torch.randn(B, H * W, C, device=dev)
    ws = torch.rand(B, H * W, device=dev)
    ws[0, :4] = torch.tensor([0.0, 1.0, 1e-6, 0.5], device=dev)
    dirs = torch.tensor([2, 5, 0], device=dev, dtype=torch.int32)
    step_dev = torch.tensor([9, 0], device=dev, dtype=torch.int32)
    gi, gws = g.train_step_image(image, dirs, H, W, step_dev, ws, 5e-4)
    rows = g.targets.permute(0, 2, 3, 1).reshape(6, H * W, C).cpu()
    want, t = O.synthetic_guidance(image.cpu(), rows, dirs.cpu(), g.weights.cpu(), 5, 9, g.min_step, g.max_step, g.noise_scale)
    assert g.min_step <= t <= g.max_step
    assert float((gi.cpu() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    assert torch.equal(gws, sparsity_loss_grad(ws, 5e-4))
    # the noise has unit variance and zero mean (B*H*W*C = 3072 deviates)
    z = (gi.cpu() / float(g.weights[t]) - (image.cpu() - rows[dirs.cpu().long()])) / g.noise_scale
    assert abs(float(z.mean())) < 0.08 and abs(float(z.std()) - 1.0) < 0.08
    again, _ = g.train_step_image(image, dirs, H, W, step_dev, None, 0.0)
    assert torch.equal(again, gi)
    step_dev[0] += 1
    fresh, none = g.train_step_image(image, dirs, H, W, step_dev, None, 0.0)
    assert none is None and not torch.equal(fresh, gi)
