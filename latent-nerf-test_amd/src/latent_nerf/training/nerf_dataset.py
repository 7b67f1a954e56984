"""Camera sampling for the NeRF trainer.  Pose distribution and eval circle follow the reference's
present ViewsDataset (src/latent_paint/training/views_dataset.py:9-35, 38-80: radius ~ U[radius_range],
theta ~ U[0, 150] deg, phi ~ U[0, 360) deg; eval: radius_range[1]*1.2, theta 60 deg, phi = 360 i/size) and
its direction bucket (src/utils.py:8-27, called with already-converted radians exactly as the reference does)."""
import math

import numpy as np
import torch

from ...utils import get_view_direction, view_direction_index
from ..models.nerf_utils import intrinsics_from_fov, pose_from_angles
from ..raymarching import raymarching as rm


class NeRFDataset:
    def __init__(self, cfg, device, type="train", H=64, W=64, size=100, seed=0):
        self.cfg, self.device, self.type = cfg, device, type
        self.H, self.W, self.size = H, W, size
        self.training = type in ("train", "all")
        self.gen = torch.Generator().manual_seed(seed)

    def sample_pose(self, index=0, generator=None, uniforms=None):
        """uniforms: four numbers in [0, 1) to use instead of a draw from `generator` (the trainer's counter-based
        per-(step, view) stream, distributed.pose_uniforms)."""
        cfg = self.cfg
        g = self.gen if generator is None else generator
        if self.training and getattr(cfg, "train_pose", None) is not None:
            th, ph, radius, fov = [float(v) for v in cfg.train_pose]
            theta, phi = math.radians(th), math.radians(ph)
        elif self.training:
            u = uniforms if uniforms is not None else torch.rand(4, generator=g).tolist()
            radius = float(cfg.radius_range[0] + u[0] * (cfg.radius_range[1] - cfg.radius_range[0]))
            theta = float(math.radians(0.0) + u[1] * (math.radians(150.0) - math.radians(0.0)))
            phi = float(u[2] * math.radians(360.0))
            fov = float(cfg.fovy_range[0] + u[3] * (cfg.fovy_range[1] - cfg.fovy_range[0]))
        else:
            radius = cfg.radius_range[1] * 1.2
            theta = math.radians(60.0)
            phi = math.radians((index / self.size) * 360.0)
            fov = 0.5 * (cfg.fovy_range[0] + cfg.fovy_range[1])
        theta = max(theta, 1e-3)
        # (the reference's callers pass already-converted radians for the two cone angles, src/utils.py:8-27 as called
        # from src/latent_paint/training/views_dataset.py:12-22; scalar form of get_view_direction for the one view)
        di = view_direction_index(theta, phi, np.deg2rad(cfg.angle_overhead), np.deg2rad(cfg.angle_front))
        return {"theta": theta, "phi": phi, "radius": radius, "fov": fov, "dir": torch.tensor([di], dtype=torch.long),
                "dir_index": int(di)}

    def collate(self, index=0, generator=None, device_pose=True):
        """device_pose=False: the pose stays on the host (the trainer's captured step uploads pose + intrinsics itself,
        as one copy into the static buffers its graph reads)."""
        p = self.sample_pose(index, generator)
        pose = pose_from_angles(p["theta"], p["phi"], p["radius"])[None]
        if device_pose:
            pose = pose.to(self.device)
        intr = intrinsics_from_fov(p["fov"], self.H, self.W)
        p.update(H=self.H, W=self.W, pose=pose, camera=(pose, intr, self.H, self.W))
        if self.training:
            # training views hand the CAMERA to the renderer: the rays are generated inside the march's count pass
            # (NeRFRenderer.render(camera=...), one dispatch less per view); evaluation keeps explicit rays
            p.update(rays_o=None, rays_d=None)
        else:
            rays_o, rays_d = rm.get_rays(pose, intr, self.H, self.W)
            p.update(rays_o=rays_o, rays_d=rays_d)
        return p

    def __iter__(self):
        for i in range(self.size):
            yield self.collate(i)

    def __len__(self):
        return self.size
