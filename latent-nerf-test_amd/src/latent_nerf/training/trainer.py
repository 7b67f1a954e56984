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
from .guidance import (LATENT_TO_RGB, StableDiffusionGuidance, SyntheticGuidance, decode_with, sparsity_loss,
                       sparsity_loss_grad)
from .nerf_dataset import NeRFDataset
from .optimizer import FusedAdam

# approximate linear latent -> RGB map used for quick previews (src/latent_paint/models/textured_mesh.py:34-40)
_LATENT_TO_RGB = torch.tensor(LATENT_TO_RGB)


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
        self.exchange = D.exchange_active()     # more than one rank, or LNERF_FORCE_DIST on one
        fuse = bool(cfg.optim.fuse_table_update) and not self.exchange and n_views == 1
        # capturable: the step counter lives on the device, so that the captured step (optim.graph_step) and the eager
        # step run the very same kernels with the very same bias corrections
        self.optimizer = FusedAdam(self.nerf.get_params(cfg.optim.lr), betas=(0.9, 0.99), eps=1e-15,
                                   encoder=self.nerf.encoder, fuse_table_update=fuse, mlp=self.nerf, capturable=True)
        # the whole loop (eager steps, captures, replays, collectives) runs on ONE non-default stream: autograd pins a
        # parameter's gradient accumulation to the stream it first ran on, and the legacy default stream cannot capture.
        # Code that back-propagates through `self.nerf` outside train() must do so under
        # `torch.cuda.stream(trainer.stream)` (or set optim.graph_step = false).
        self.stream = torch.cuda.Stream(device=self.device)
        self._gstep, self._gstep_capacity, self._static, self._whole = None, None, None, False
        self.graph_stats = {"captures": 0, "replayed_steps": 0, "eager_steps": 0}
        small = [p for p in self.nerf.parameters() if p is not self.nerf.encoder.embeddings]
        # exchange: bf16 on the wire with the bf16 configuration (f32 otherwise); with one view per rank and step the
        # backward pass writes the wire buffer itself and the table travels in level groups (pipelined with the sums)
        bf16 = cfg.render.precision("mlp_precision") == "bf16"
        self.grad_sync = D.GradSync([self.nerf.encoder.embeddings], small,
                                    transport=torch.bfloat16 if bf16 else torch.float32)
        self.pipelined = self.exchange and bf16 and n_views == 1
        # the exchange goes into the captured step where the collectives can be captured (RCCL; not gloo's host staging)
        self.capture_exchange = (self.exchange and bool(getattr(cfg.optim, "graph_collectives", True))
                                 and D.backend_name() == "nccl"
                                 and os.environ.get("LNERF_GRAPH_COLLECTIVES", "1") != "0")
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
            if not getattr(self.cfg.log, "quiet", False):
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
    def _render_train(self, camera):
        out = self.nerf.render(None, None, staged=False, perturb=True, bg_color=None, force_all_rays=True, camera=camera)
        H, W = int(camera[2]), int(camera[3])
        pred = out["image"].reshape(1, H, W, -1).permute(0, 3, 1, 2).contiguous()
        return out, pred

    def _guidance_grad(self, pred, dirs):
        text_z = self.text_z[int(dirs[0])] if isinstance(self.text_z, list) else self.text_z
        if isinstance(self.diffusion, SyntheticGuidance):
            return self.diffusion.train_step(text_z, pred, dirs=dirs)
        return self.diffusion.train_step(text_z, pred)

    def _backward(self, out, pred, grad):
        """SDS: d(loss)/d(pred) = grad (src/latent_paint_mesh/training/trainer.py:657-658).  The sparsity term's gradient
        w.r.t. weights_sum comes from one HIP launch and enters the compositing backward beside it; the shape term (a
        function of the samples) goes through autograd."""
        tensors, grads = [pred], [grad]
        if self.cfg.optim.lambda_sparsity > 0:
            tensors.append(out["weights_sum"])
            grads.append(sparsity_loss_grad(out["weights_sum"], self.cfg.optim.lambda_sparsity))
        if self.shape_loss is not None and self.cfg.optim.lambda_shape > 0:
            loss = self.cfg.optim.lambda_shape * self.shape_loss(out["xyzs"], out["sigmas"], out["counter"])
            tensors.append(loss)
            grads.append(torch.ones_like(loss))
        torch.autograd.backward(tensors, grads)

    def step_loss(self, out):
        """Value of the auxiliary terms of one render (logging only; the step itself never needs it)."""
        loss = torch.zeros((), device=self.device)
        if self.cfg.optim.lambda_sparsity > 0:
            loss = loss + self.cfg.optim.lambda_sparsity * sparsity_loss(out["weights_sum"].detach())
        return loss

    def train_render(self, data):
        """Render one view and inject the guidance gradient.  Returns (pred latents [1,C,H,W], the render's dict)."""
        if data["rays_o"] is None:
            out, pred = self._render_train(data["camera"])
        else:
            H, W = data["H"], data["W"]
            out = self.nerf.render(data["rays_o"], data["rays_d"], staged=False, perturb=True, bg_color=None,
                                   force_all_rays=True)
            pred = out["image"].reshape(1, H, W, -1).permute(0, 3, 1, 2).contiguous()
        grad = self._guidance_grad(pred, data["dir"])
        self._backward(out, pred, grad)
        return pred, out

    def _exchange_and_step(self, n_views):
        scale = 1.0 / (n_views * self.world)
        if self.pipelined:
            ex = self.grad_sync.allreduce_pipelined()
            ex.finish_small()
            self.optimizer.step(grad_scale=scale, grads=self.grad_sync.reduced(),
                                row_groups={self.nerf.encoder.embeddings: ex.table_groups})
        else:
            self.grad_sync.allreduce()
            self.optimizer.step(grad_scale=scale)

    # ---- the captured step (optim.graph_step): graph F (render) / eager guidance / graph B (backward [+ optimiser])
    def _graph_ready(self, n_views):
        r = self.cfg.render
        return (bool(getattr(self.cfg.optim, "graph_step", True)) and n_views == 1 and self.nerf.cuda_ray
                and r.noise_seed is not None and self.nerf.bg_radius <= 0)

    def _capture(self):
        """Capture the step for the current sample capacity.  Runs no kernel: the training state does not advance."""
        from .graph_step import GraphedRenderStep
        r = self.cfg.render
        C, H, W = self.nerf.img_dims, r.train_h, r.train_w
        if self._static is None:
            self._static = {"pose": torch.zeros(1, 4, 4, device=self.device), "intr": torch.zeros(1, 4, device=self.device),
                            "grad": torch.zeros(1, C, H, W, device=self.device),
                            # ring of pinned upload slots: the host runs steps ahead of the GPU, a slot is rewritten
                            # only after the copy that read it has executed (event per slot)
                            # (16 pose + 4 intrinsics + the view bucket as an int32 bit pattern)
                            "host": torch.zeros(64, 21, dtype=torch.float32).pin_memory(), "events": [None] * 64,
                            "cam": torch.zeros(21, device=self.device)}
        st = self._static
        solo = not self.exchange
        inline = self.capture_exchange
        opt = self.optimizer
        keep = (opt.step_no, self.nerf.local_step)

        def forward():
            # (one 80-byte upload per step lands in `cam`; the two views below alias it)
            return self._render_train((st["cam"][:16].view(1, 4, 4), st["cam"][16:20].view(1, 4), H, W))

        def backward(out, pred):
            if solo:
                opt.arm()                 # one view, one process: the scatter applies the table's Adam step
            self._backward(out, pred, st["grad"])
            if solo:
                opt.step(grad_scale=1.0)
            elif inline:
                self._exchange_and_step(1)   # collectives and the optimiser's waits on them are captured

        # a guidance that is itself capturable (the synthetic one) goes INSIDE the graph: one launch per step
        self._whole = bool(getattr(self.diffusion, "capturable", False)) and hasattr(self.diffusion, "train_step_device") \
            and bool(getattr(self.cfg.optim, "graph_guidance", True))
        if self._whole:
            from .graph_step import GraphedWholeStep
            dir_dev = st["cam"][20:21].view(torch.int32)

            def whole():
                out, pred = forward()
                grad = self.diffusion.train_step_device(pred, dir_dev)
                if solo:
                    opt.arm()
                self._backward(out, pred, grad)
                if solo:
                    opt.step(grad_scale=1.0)
                elif inline:
                    self._exchange_and_step(1)
                return out, pred

            build = lambda: GraphedWholeStep(whole, list(self.nerf.parameters()), self.stream)
        else:
            build = lambda: GraphedRenderStep(forward, backward, list(self.nerf.parameters()), self.stream)
        try:
            self._gstep = build()
        except RuntimeError as e:
            opt.step_no, self.nerf.local_step = keep
            if not inline:
                raise
            # a collective the backend cannot capture (every rank takes the same turn): exchange + optimiser stay eager
            self.log("capture with collectives failed (%s): the exchange stays outside the graphs" % str(e).splitlines()[0])
            torch.cuda.synchronize()
            self.capture_exchange = False
            return self._capture()
        opt.step_no, self.nerf.local_step = keep   # host-side counters the captured Python advanced
        self._gstep_capacity = self.nerf._march.capacity
        self.graph_stats["captures"] += 1

    def _graphed_step(self, data):
        st, g = self._static, self._gstep
        slot = self.train_step % 64
        if st["events"][slot] is not None:
            st["events"][slot].synchronize()
        host = st["host"][slot]
        host[:16] = data["pose"].reshape(-1)
        host[16:20] = torch.tensor(data["camera"][1], dtype=torch.float32)
        host[20:21].view(torch.int32)[0] = int(data["dir"][0])
        st["cam"].copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st["events"][slot] = ev
        if self._whole:
            grads = g.replay()
        else:
            out, pred = g.forward()
            st["grad"].copy_(self._guidance_grad(pred, data["dir"]))
            grads = g.backward()
        self.nerf.local_step += 1
        if not self.exchange or self.capture_exchange:
            self.optimizer.note_replayed_step()
        else:   # exchange outside the graphs: the captured backward left this rank's gradients; the rest is eager
            for p, gr in zip(g.params, grads):
                p.grad = gr
            self._exchange_and_step(1)
            for p in g.params:
                p.grad = None
        self.graph_stats["replayed_steps"] += 1

    def train(self, iters=None):
        iters = self.cfg.optim.iters if iters is None else iters
        self.nerf.train()
        views = D.views_for_rank(max(self.cfg.optim.views_per_step, self.world), self.rank, self.world)
        ds = self.dataloaders["train"]
        use_graph = self._graph_ready(len(views))
        eager_left = 2          # eager steps before the first capture (lazy allocations, workspace sizes, autograd streams)
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            while self.train_step < iters:
                self.train_step += 1
                if self.nerf.cuda_ray and (self.train_step - 1) % self.cfg.render.update_extra_interval == 0:
                    self.nerf.update_extra_state()
                    if self._gstep is not None and self.nerf._capacity(*self.nerf._march_key) != self._gstep_capacity:
                        self._gstep, eager_left = None, 1   # the sample budget moved: new buffers, capture again
                if use_graph and self._gstep is None and eager_left <= 0:
                    self._capture()
                if use_graph and self._gstep is not None:
                    data = ds.collate(0, generator=D.pose_generator(self.cfg.optim.seed, self.train_step, views[0]),
                                      device_pose=False)
                    self._graphed_step(data)
                else:
                    eager_left -= 1
                    self.graph_stats["eager_steps"] += 1
                    self.optimizer.zero_grad()
                    for v in views:
                        data = ds.collate(0, generator=D.pose_generator(self.cfg.optim.seed, self.train_step, v))
                        if len(views) == 1:
                            self.optimizer.arm()   # one view, one process: the scatter applies the table's Adam step
                        self.train_render(data)
                    self._exchange_and_step(len(views))
                if self.train_step % self.cfg.log.save_interval == 0:
                    self.save_checkpoint(full=True)
                    self.evaluate(self.dataloaders["val"], self.eval_renders_path)
                    self.nerf.train()
        torch.cuda.current_stream().wait_stream(self.stream)
        self.log("finished training at step %d (%s)" % (self.train_step, self.graph_stats))
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
        """[1,C,H,W] latents -> uint8 [H',W',3].  With `log.decode_eval` the guidance model's decoder turns the latents
        into the image (vae.decode, src/stable_diffusion.py:462-470; a guidance object without a decoder gives the
        linear preview at 8x); default: the linear latent->RGB estimate at the render resolution (no VAE offline)."""
        if getattr(self.cfg.log, "decode_eval", False) and self.nerf.latent_mode:
            rgb = decode_with(self.diffusion, latents.float())[0].permute(1, 2, 0).cpu()
            return tensor2numpy(rgb)
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
