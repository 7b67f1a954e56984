"""Trainer of the Latent-Paint path (BASELINE config 5) on the HIP raster kernels: stands in for
src/latent_paint/training/trainer.py (Trainer.__init__ :25-54, init_mesh_model :56-72, train :113-144, evaluate
:146-174, full_eval :176-188, train_render :190-209, eval_render :211-221, load/save_checkpoint :235-310) with the
same public methods, experiment-directory layout and checkpoint schema.

One deliberate difference.  In this fork `StableDiffusion.train_step` RETURNS the SDS gradient
(src/stable_diffusion.py:327-334) and the reference's Latent-Paint trainer never back-propagates it
(:206-209 -- its optimiser steps on empty gradients).  This trainer injects it the way the fork's own live trainer
does, `pred_rgb.backward(gradient=grad)` (src/latent_paint_mesh/training/trainer.py:657-658).

Rendering, its backward and the Adam step run on the HIP library; this file is plumbing."""
import json
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

from ... import config_cli
from ...latent_nerf.training.guidance import StableDiffusionGuidance, SyntheticGuidance, decode_with
from ...latent_nerf.training.optimizer import FusedAdam
from ...utils import make_path, seed_everything, tensor2numpy, write_video
from ..configs.train_config import TrainConfig
from ..models.textured_mesh import TexturedMeshModel
from .views_dataset import ViewsDataset

DIRECTION_WORDS = ("front", "side", "back", "side", "overhead", "bottom")   # index = view bucket (src/utils.py:8-27)


class Trainer:
    def __init__(self, cfg: TrainConfig, device=None, guidance=None):
        self.cfg = cfg
        self.train_step = 0
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("the Latent-Paint render path runs on the HIP library only: no GPU is visible")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        seed_everything(cfg.optim.seed)
        self.exp_path = make_path(Path(cfg.log.exp_dir))
        self.ckpt_path = make_path(self.exp_path / "checkpoints")
        self.train_renders_path = make_path(self.exp_path / "vis" / "train")
        self.eval_renders_path = make_path(self.exp_path / "vis" / "eval")
        self.final_renders_path = make_path(self.exp_path / "results")
        with open(self.exp_path / "config.json", "w") as fh:
            json.dump(config_cli.to_plain_dict(cfg), fh, indent=1)
        self.mesh_model = self.init_mesh_model()
        self.diffusion = guidance if guidance is not None else self.init_diffusion()
        self.text_z = self.calc_text_embeddings()
        self.optimizer = self.init_optimizer()
        self.dataloaders = self.init_dataloaders()
        self.past_checkpoints = []
        if cfg.optim.resume:
            self.load_checkpoint(model_only=False)
        if cfg.optim.ckpt is not None:
            self.load_checkpoint(cfg.optim.ckpt, model_only=True)
        self.log("initialized %s" % cfg.log.exp_name)

    # ------------------------------------------------------------------ set-up
    def log(self, msg):
        print("[latent-paint] " + msg, flush=True)
        with open(self.exp_path / "log.txt", "a") as fh:
            fh.write(msg + "\n")

    def init_mesh_model(self):
        backbones = {"texture-mesh": True, "texture-rgb-mesh": False}   # -> latent_mode
        if self.cfg.render.backbone not in backbones:
            raise NotImplementedError("--backbone %s is not implemented!" % self.cfg.render.backbone)
        model = TexturedMeshModel(self.cfg, device=self.device, render_grid_size=self.cfg.render.train_grid_size,
                                  latent_mode=backbones[self.cfg.render.backbone],
                                  texture_resolution=self.cfg.guide.texture_resolution).to(self.device)
        self.log("loaded %s mesh, #parameters: %d" % (self.cfg.render.backbone,
                                                     sum(p.numel() for p in model.parameters() if p.requires_grad)))
        return model

    def init_diffusion(self):
        g = self.cfg.guide
        if g.guidance == "synthetic":
            return SyntheticGuidance(self.device, channels=4 if self.mesh_model.latent_mode else 3, size=64,
                                     seed=self.cfg.optim.seed)
        return StableDiffusionGuidance(self.device, g.diffusion_name)

    def calc_text_embeddings(self):
        text = self.cfg.guide.text
        if not self.cfg.guide.append_direction:
            return self.diffusion.get_text_embeds(text)
        return [self.diffusion.get_text_embeds("%s, %s view" % (text, word)) for word in DIRECTION_WORDS]

    def init_optimizer(self):
        return FusedAdam([{"params": self.mesh_model.get_params(), "lr": self.cfg.optim.lr}], betas=(0.9, 0.99),
                         eps=1e-15)

    def init_dataloaders(self):
        r, lg = self.cfg.render, self.cfg.log
        return {"train": ViewsDataset(r, self.device, "train", 100, seed=self.cfg.optim.seed).dataloader(),
                "val": ViewsDataset(r, self.device, "val", lg.eval_size).dataloader(),
                "val_large": ViewsDataset(r, self.device, "val", lg.full_eval_size).dataloader()}

    # ------------------------------------------------------------------ optimisation
    def train(self):
        self.log("starting training")
        self.evaluate(self.dataloaders["val"], self.eval_renders_path)     # the initialisation
        self.mesh_model.train()
        rng = np.random.RandomState(self.cfg.optim.seed)
        while self.train_step < self.cfg.optim.iters:
            for data in self.dataloaders["train"]:                            # 100 fresh random views per pass
                if self.train_step >= self.cfg.optim.iters:
                    break
                self.train_step += 1
                self.optimizer.zero_grad()
                pred, _ = self.train_render(data)
                self.optimizer.step()
                if self.train_step % self.cfg.log.save_interval == 0:
                    self.save_checkpoint(full=True)
                    self.evaluate(self.dataloaders["val"], self.eval_renders_path)
                    self.mesh_model.train()
                if rng.uniform(0, 1) < 0.05:
                    self.log_train_renders(pred.detach())
        self.log("finished training, evaluating the last model")
        self.full_eval()

    def train_render(self, data):
        """One view: render, ask the guidance for d(loss)/d(pred) and push it through the render graph."""
        out = self.mesh_model.render(theta=data["theta"], phi=data["phi"], radius=data["radius"])
        pred = out["image"]
        if self.cfg.guide.append_direction:
            text_z = self.text_z[int(data["dir"][0])]
        else:
            text_z = self.text_z
        if isinstance(self.diffusion, SyntheticGuidance):
            grad = self.diffusion.train_step(text_z, pred, dirs=data["dir"])
        else:
            grad = self.diffusion.train_step(text_z, pred)
        pred.backward(gradient=grad)
        return pred, grad

    # ------------------------------------------------------------------ evaluation
    def _decode(self, latents):
        """The guidance model's VAE decoder (src/stable_diffusion.py:462-470) when it has one, else the linear
        latent -> RGB preview: a guidance object without `decode_latents` must not end an evaluation."""
        return decode_with(self.diffusion, latents)

    @torch.no_grad()
    def eval_render(self, data):
        side = self.cfg.render.eval_grid_size
        out = self.mesh_model.render(theta=data["theta"], phi=data["phi"], radius=data["radius"],
                                     decode_func=self._decode, test=True, dims=(side, side))
        as_image = lambda t: t.permute(0, 2, 3, 1).contiguous().clamp(0, 1)
        return as_image(out["image"]), as_image(out["texture_map"])

    @torch.no_grad()
    def evaluate(self, dataloader, save_path: Path, save_as_video=False):
        from PIL import Image
        self.mesh_model.eval()
        save_path.mkdir(exist_ok=True, parents=True)
        frames, texture = [], None
        for i, data in enumerate(dataloader):
            preds, textures = self.eval_render(data)
            frame, texture = tensor2numpy(preds[0]), textures
            if save_as_video:
                frames.append(frame)
            else:
                Image.fromarray(frame).save(save_path / ("step_%05d_%04d_rgb.png" % (self.train_step, i)))
        if texture is not None:   # the texture map is the same for every view
            Image.fromarray(tensor2numpy(texture[0])).save(save_path / ("step_%05d_texture.png" % self.train_step))
        if save_as_video and frames:
            write_video(save_path / ("step_%05d_rgb" % self.train_step), frames)
        return frames

    def full_eval(self):
        self.evaluate(self.dataloaders["val_large"], self.final_renders_path, save_as_video=True)
        if self.cfg.log.save_mesh:
            target = make_path(self.exp_path / "mesh")
            self.mesh_model.export_mesh(target, guidance=self.diffusion)
            self.log("saved mesh to %s" % target)

    @torch.no_grad()
    def log_train_renders(self, preds):
        from PIL import Image
        if self.mesh_model.latent_mode:
            rgb = self._decode(preds).permute(0, 2, 3, 1).contiguous()
        else:
            rgb = preds.permute(0, 2, 3, 1).contiguous().clamp(0, 1)
        Image.fromarray(tensor2numpy(rgb[0])).save(self.train_renders_path / ("step_%05d.jpg" % self.train_step))

    # ------------------------------------------------------------------ checkpoints
    # {'train_step', 'checkpoints': [kept file names], 'model': state_dict[, 'optimizer']} in
    # <exp>/checkpoints/step_%06d.pth, the newest `max_keep_ckpts` kept (:288-310)
    def save_checkpoint(self, full=False):
        state = {"train_step": self.train_step, "checkpoints": self.past_checkpoints}
        if full:
            state["optimizer"] = self.optimizer.state_dict()
        state["model"] = self.mesh_model.state_dict()
        file_name = "step_%06d.pth" % self.train_step
        self.past_checkpoints.append(file_name)
        while len(self.past_checkpoints) > self.cfg.log.max_keep_ckpts:
            (self.ckpt_path / self.past_checkpoints.pop(0)).unlink(missing_ok=True)
        torch.save(state, self.ckpt_path / file_name)
        return self.ckpt_path / file_name

    def _rgb_texture_from_latents(self, latent_texture):
        """Start of the RGB fine-tuning backbone: the decoded latent texture at the texture resolution (:248-253)."""
        side = self.cfg.guide.texture_resolution
        return F.interpolate(self._decode(latent_texture.to(self.device)), (side, side),
                             mode="bilinear", align_corners=False)

    def load_checkpoint(self, checkpoint=None, model_only=False):
        if checkpoint is None:
            found = sorted(self.ckpt_path.glob("*.pth"))
            if not found:
                self.log("no checkpoint found, model randomly initialized")
                return
            checkpoint = found[-1]
        state = torch.load(checkpoint, map_location=self.device, weights_only=True)
        weights = state["model"] if "model" in state else state      # bare state dicts are accepted too
        if not self.mesh_model.latent_mode:
            weights["texture_img_rgb_finetune"] = self._rgb_texture_from_latents(weights["texture_img"])
        if "model" not in state:
            self.mesh_model.load_state_dict(weights)
            return
        missing, unexpected = self.mesh_model.load_state_dict(weights, strict=False)
        if missing or unexpected:
            self.log("checkpoint: missing keys %s, unexpected keys %s" % (missing, unexpected))
        if model_only:
            return
        self.past_checkpoints = list(state["checkpoints"])
        self.train_step = int(state["train_step"]) + 1
        if "optimizer" in state:
            try:
                self.optimizer.load_state_dict(state["optimizer"])
            except (ValueError, RuntimeError, KeyError) as e:
                self.log("failed to load optimizer state: %s" % e)
