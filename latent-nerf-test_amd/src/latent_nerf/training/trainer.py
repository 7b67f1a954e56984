"""Trainer of the latent-NeRF path: the counterpart of the `src.latent_nerf.training.trainer.Trainer`
the reference's script imports (scripts/train_latent_nerf.py:3-4,8-14) but does not ship.  Structure,
checkpoint schema and evaluation follow the reference's present Latent-Paint trainer
(src/latent_paint/training/trainer.py: __init__ :25-54, train :113-144, evaluate :146-174,
train_render :190-209, load_checkpoint :235-286, save_checkpoint :288-310) with its one known defect
fixed the way its own fork does: the SDS gradient returned by `train_step` is injected with
`pred.backward(gradient=grad)` (src/latent_paint_mesh/training/trainer.py:657-658).

Every render/backward/optimiser op runs on the HIP library; this file is plumbing."""
import json
import math
import os
from pathlib import Path

import numpy as np
import torch

from ...utils import make_path, seed_everything, tensor2numpy, write_video
from ..configs.train_config import TrainConfig
from ..models.network_grid import NeRFNetwork
from . import distributed as D
from .guidance import StableDiffusionGuidance, SyntheticGuidance, sparsity_loss
from .nerf_dataset import NeRFDataset
from .optimizer import FusedAdam

# approximate linear latent -> RGB map used for quick previews (src/latent_paint/models/textured_mesh.py:34-40)
_LATENT_TO_RGB = torch.tensor([[0.298, 0.207, 0.208], [0.187, 0.286, 0.173], [-0.158, 0.189, 0.264],
                               [-0.184, -0.271, -0.473]])


class Trainer:
    def __init__(self, cfg: TrainConfig, device=None, guidance=None):
        """Data parallel: launch one process per GPU (`python -m torch.distributed.run --nproc-per-node N -m
        scripts.train_latent_nerf ...`); scripts/train_latent_nerf.py calls distributed.init_distributed() before
        anything touches the GPU.  Every rank holds a full replica, renders `optim.views_per_step / N` views per
        step and exchanges gradients once per step (GradSync); replicas stay bit-identical (identical initial
        weights, identical reduced gradients, identical Adam step, identically seeded occupancy refresh)."""
        self.cfg = cfg
        self.train_step = 0
        self.rank, self.world = D.world_info()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        seed_everything(cfg.optim.seed)   # model initialisation: the same stream on every rank
        self.exp_path = make_path(cfg.log.exp_dir)
        self.ckpt_path = make_path(self.exp_path / "checkpoints")
        self.train_renders_path = make_path(self.exp_path / "vis" / "train")
        self.eval_renders_path = make_path(self.exp_path / "vis" / "eval")
        self.final_renders_path = make_path(self.exp_path / "results")
        if self.rank == 0:
            with open(self.exp_path / "config.json", "w") as f:
                json.dump(_cfg_to_dict(cfg), f, indent=1, default=str)
        if self.world > 1 and cfg.render.noise_seed is not None:
            # march jitter: a different counter-based stream per rank (poses come from pose_generator(seed, step, view),
            # which every rank can reproduce; jitter and guidance noise must NOT be correlated across ranks)
            cfg.render.noise_seed = (int(cfg.render.noise_seed) + 0x9E3779B1 * self.rank) & 0x7FFFFFFF
        self.nerf = NeRFNetwork(cfg.render).to(self.device)
        D.broadcast_parameters(list(self.nerf.parameters()))          # replicas start from rank 0's weights
        self.nerf.occupancy_generator(seed=cfg.optim.seed)            # the same refresh samples on every rank
        if self.world > 1:                                            # guidance noise / timesteps: per-rank streams
            torch.manual_seed(cfg.optim.seed + 7919 * (self.rank + 1))
            torch.cuda.manual_seed(cfg.optim.seed + 7919 * (self.rank + 1))
        self.diffusion = guidance if guidance is not None else self.init_diffusion()
        self.text_z = self.calc_text_embeddings()
        n_views = len(D.views_for_rank(max(cfg.optim.views_per_step, self.world), self.rank, self.world))
        fuse = bool(cfg.optim.fuse_table_update) and self.world == 1 and n_views == 1
        self.optimizer = FusedAdam(self.nerf.get_params(cfg.optim.lr), betas=(0.9, 0.99), eps=1e-15,
                                   encoder=self.nerf.encoder, fuse_table_update=fuse, mlp=self.nerf)
        small = [p for p in self.nerf.parameters() if p is not self.nerf.encoder.embeddings]
        # exchange: bf16 on the wire with the bf16 configuration (f32 otherwise); with one view per rank and step the
        # backward pass writes the wire buffer itself and the table travels in level groups (pipelined with the sums)
        bf16 = cfg.render.precision("mlp_precision") == "bf16"
        self.grad_sync = D.GradSync([self.nerf.encoder.embeddings], small,
                                    transport=torch.bfloat16 if bf16 else torch.float32)
        self.pipelined = self.world > 1 and bf16 and n_views == 1
        if self.pipelined:
            self.grad_sync.attach_sink(self.nerf.encoder, pipeline_groups=max(1, cfg.optim.exchange_groups))
        self.dataloaders = self.init_dataloaders()
        self.shape_loss = self.init_shape_guidance()
        self.past_checkpoints = []
        if cfg.optim.ckpt is not None:
            self.load_checkpoint(cfg.optim.ckpt, model_only=True)
        if cfg.optim.resume:
            self.load_checkpoint(model_only=False)
        self.log("trainer ready: %d parameters, exp dir %s" % (sum(p.numel() for p in self.nerf.parameters()),
                                                              self.exp_path))

    # ------------------------------------------------------------------ set-up
    def log(self, msg):
        if self.rank == 0:
            print("[trainer] " + msg, flush=True)
            with open(self.exp_path / "log.txt", "a") as f:
                f.write(msg + "\n")

    def init_diffusion(self):
        g = self.cfg.guide
        if g.guidance == "synthetic":
            return SyntheticGuidance(self.device, channels=self.nerf.img_dims, size=self.cfg.render.train_h,
                                     seed=self.cfg.optim.seed)
        return StableDiffusionGuidance(self.device, g.diffusion_name)

    def init_shape_guidance(self):
        """`guide.shape_path` -> mesh occupancy grids + shape loss (sketch-shape guidance)."""
        g = self.cfg.guide
        if not g.shape_path:
            return None
        from .shape import MeshOccupancy, ShapeLoss, load_obj, normalize_mesh
        verts, faces = load_obj(g.shape_path)
        verts = normalize_mesh(verts, target_scale=g.mesh_scale, dy=0.0)
        self.mesh_occ = MeshOccupancy(verts, faces, self.device, bound=self.nerf.bound,
                                      resolution=self.cfg.render.grid_size)
        self.mesh_occ.init_density_grid(self.nerf)
        self.log("shape guidance: %d faces from %s" % (faces.shape[0], g.shape_path))
        return ShapeLoss(self.mesh_occ, proximal_surface=g.proximal_surface)

    def calc_text_embeddings(self):
        """One embedding, or six direction-specific ones (src/latent_paint/training/trainer.py:82-91)."""
        ref_text = self.cfg.guide.text
        if not self.cfg.guide.append_direction:
            return self.diffusion.get_text_embeds(ref_text)
        return [self.diffusion.get_text_embeds("%s, %s view" % (ref_text, d))
                for d in ("front", "side", "back", "side", "overhead", "bottom")]

    def init_dataloaders(self):
        r = self.cfg.render
        return {
            "train": NeRFDataset(r, self.device, "train", r.train_h, r.train_w, 100, seed=self.cfg.optim.seed),
            "val": NeRFDataset(r, self.device, "val", r.eval_h, r.eval_w, self.cfg.log.eval_size),
            "val_large": NeRFDataset(r, self.device, "val", r.eval_h, r.eval_w, self.cfg.log.full_eval_size),
        }

    # ------------------------------------------------------------------ one optimisation step
    def train_render(self, data):
        """Render one view and inject the guidance gradient.  Returns (pred latents [1,C,H,W], loss scalar)."""
        H, W = data["H"], data["W"]
        out = self.nerf.render(data["rays_o"], data["rays_d"], staged=False, perturb=True, bg_color=None,
                               force_all_rays=True, camera=data.get("camera") if data["rays_o"] is None else None)
        pred = out["image"].reshape(1, H, W, -1).permute(0, 3, 1, 2).contiguous()
        dirs = data["dir"]
        text_z = self.text_z[int(dirs[0])] if isinstance(self.text_z, list) else self.text_z
        grad = self.diffusion.train_step(text_z, pred, dirs=dirs) if isinstance(self.diffusion, SyntheticGuidance) \
            else self.diffusion.train_step(text_z, pred)
        loss = torch.zeros((), device=self.device)
        if self.cfg.optim.lambda_sparsity > 0:
            loss = loss + self.cfg.optim.lambda_sparsity * sparsity_loss(out["weights_sum"])
        if self.shape_loss is not None and self.cfg.optim.lambda_shape > 0:
            loss = loss + self.cfg.optim.lambda_shape * self.shape_loss(out["xyzs"], out["sigmas"], out["counter"])
        # SDS: d(loss)/d(pred) = grad (src/latent_paint_mesh/training/trainer.py:657-658); other terms by autograd
        if loss.requires_grad:
            torch.autograd.backward([pred, loss], [grad, torch.ones_like(loss)])
        else:
            pred.backward(gradient=grad)
        return pred, loss

    def train(self, iters=None):
        iters = self.cfg.optim.iters if iters is None else iters
        self.nerf.train()
        views = D.views_for_rank(max(self.cfg.optim.views_per_step, self.world), self.rank, self.world)
        ds = self.dataloaders["train"]
        while self.train_step < iters:
            self.train_step += 1
            if self.nerf.cuda_ray and (self.train_step - 1) % self.cfg.render.update_extra_interval == 0:
                self.nerf.update_extra_state()
            self.optimizer.zero_grad()
            for v in views:
                data = ds.collate(0, generator=D.pose_generator(self.cfg.optim.seed, self.train_step, v))
                if len(views) == 1:
                    self.optimizer.arm()   # one view, one process: the scatter applies the table's Adam step
                self.train_render(data)
            scale = 1.0 / (len(views) * self.world)
            if self.pipelined:
                ex = self.grad_sync.allreduce_pipelined()
                ex.finish_small()
                self.optimizer.step(grad_scale=scale, grads=self.grad_sync.reduced(),
                                    row_groups={self.nerf.encoder.embeddings: ex.table_groups})
            else:
                self.grad_sync.allreduce()
                self.optimizer.step(grad_scale=scale)
            if self.train_step % self.cfg.log.save_interval == 0:
                self.save_checkpoint(full=True)
                self.evaluate(self.dataloaders["val"], self.eval_renders_path)
                self.nerf.train()
        self.log("finished training at step %d" % self.train_step)
        # the reference's trainers close with the full evaluation pass of the last model
        # (src/latent_paint/training/trainer.py:141-144)
        self.log("evaluating the last model...")
        self.full_eval()
        self.nerf.train()

    # ------------------------------------------------------------------ evaluation
    @torch.no_grad()
    def eval_render(self, data):
        H, W = data["H"], data["W"]
        out = self.nerf.render(data["rays_o"], data["rays_d"], staged=True, perturb=False, bg_color=None)
        pred = out["image"].reshape(1, H, W, -1).permute(0, 3, 1, 2).contiguous()
        depth = out["depth"].reshape(1, H, W)
        return pred, depth

    @torch.no_grad()
    def evaluate(self, dataset, save_path: Path, save_as_video=False):
        """One render per evaluation pose: PNG files during training, one video for the final pass (file names of
        src/latent_paint/training/trainer.py:146-172)."""
        from PIL import Image
        self.nerf.eval()
        frames = []
        for i, data in enumerate(dataset):
            pred, depth = self.eval_render(data)
            rgb = self.preview_rgb(pred)
            frames.append(rgb)
            if self.rank == 0 and not save_as_video:
                Image.fromarray(rgb).save(save_path / ("step_%05d_%04d_rgb.png" % (self.train_step, i)))
        if self.rank == 0 and save_as_video and frames:
            write_video(save_path / ("step_%05d_rgb" % self.train_step), frames)
        return frames

    def full_eval(self):
        return self.evaluate(self.dataloaders["val_large"], self.final_renders_path, save_as_video=True)

    def preview_rgb(self, latents):
        """[1,C,H,W] latents -> uint8 [H,W,3] via the linear latent->RGB estimate (no VAE offline)."""
        x = latents[0].permute(1, 2, 0).float().cpu()
        rgb = x @ _LATENT_TO_RGB if x.shape[-1] == 4 else x[..., :3]
        return tensor2numpy(rgb.clamp(-1, 1))

    # ------------------------------------------------------------------ checkpoints
    # schema of src/latent_paint/training/trainer.py:288-310: {'train_step', 'checkpoints', 'model'[, 'optimizer']}
    def save_checkpoint(self, full=False):
        if self.rank != 0:
            return None
        name = "step_%06d" % self.train_step
        state = {"train_step": self.train_step, "checkpoints": self.past_checkpoints,
                 "model": self.nerf.state_dict()}
        if full:
            state["optimizer"] = self.optimizer.state_dict()
        file_path = "%s.pth" % name
        self.past_checkpoints.append(file_path)
        if len(self.past_checkpoints) > self.cfg.log.max_keep_ckpts:
            old = self.ckpt_path / self.past_checkpoints.pop(0)
            old.unlink(missing_ok=True)
        torch.save(state, self.ckpt_path / file_path)
        return self.ckpt_path / file_path

    def load_checkpoint(self, checkpoint=None, model_only=False):
        if checkpoint is None:
            found = sorted(self.ckpt_path.glob("*.pth"))
            if not found:
                self.log("no checkpoint found, model randomly initialized")
                return
            checkpoint = found[-1]
        state = torch.load(checkpoint, map_location=self.device, weights_only=True)
        if "model" not in state:
            self.nerf.load_state_dict(state)
            return
        missing, unexpected = self.nerf.load_state_dict(state["model"], strict=False)
        if missing or unexpected:
            self.log("checkpoint: missing %s unexpected %s" % (missing, unexpected))
        if model_only:
            return
        self.past_checkpoints = list(state["checkpoints"])
        self.train_step = int(state["train_step"]) + 1
        if "optimizer" in state:
            self.optimizer.load_state_dict(state["optimizer"])


def _cfg_to_dict(cfg):
    import dataclasses
    return dataclasses.asdict(cfg)
