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
            "log.save_interval": 30, "log.eval_size": 2, "optim.fp16": False, "guide.text": "a lego man"}
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
    assert set(state) == {"train_step", "checkpoints", "model", "optimizer"} and state["train_step"] == 60
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
    frames = tr.full_eval() if False else tr.evaluate(tr.dataloaders["val"], tr.eval_renders_path)
    assert len(frames) == 2 and frames[0].shape == (32, 32, 3) and frames[0].dtype.name == "uint8"


def test_bf16_training_step_runs(dev, tmp_path):
    from src.latent_nerf.training.trainer import Trainer
    cfg = _cfg(tmp_path, **{"optim.fp16": True, "optim.iters": 20, "log.save_interval": 1000, "log.exp_name": "b"})
    assert cfg.render.mlp_precision == "bf16" and cfg.render.table_dtype == "bf16"
    tr = Trainer(cfg, device=dev)
    tr.train()
    assert tr.train_step == 20 and bool(torch.isfinite(tr.nerf.encoder.embeddings).all())
    assert bool(torch.isfinite(tr.nerf.w1).all())
