"""Small helpers of the renderer: NeRF type tag, camera poses, ray generation.

Camera convention restates the reference's present Latent-Paint camera
(src/latent_paint/models/render.py:19-31: eye = r (sin th sin ph, cos th, sin th cos ph), look-at
(0, dy, 0), world up +y) and pose distribution (src/latent_paint/training/views_dataset.py:9-35)."""
import math
from enum import Enum

import numpy as np
import torch

from ..raymarching import raymarching as _rm


class NeRFType(Enum):
    latent = "latent"
    rgb = "rgb"
    latent_tune = "latent_tune"


def pose_from_angles(theta, phi, radius, target=(0.0, 0.0, 0.0)):
    """Camera-to-world [4,4] float32 (CPU tensor).  Columns: right, down, forward, eye."""
    eye = np.array([radius * math.sin(theta) * math.sin(phi), radius * math.cos(theta),
                    radius * math.sin(theta) * math.cos(phi)], dtype=np.float64)
    tgt = np.asarray(target, dtype=np.float64)
    up = np.array([0.0, 1.0, 0.0])
    fwd = tgt - eye
    fwd = fwd / max(np.linalg.norm(fwd), 1e-20)
    right = np.cross(fwd, up)
    if np.linalg.norm(right) < 1e-8:
        right = np.array([1.0, 0.0, 0.0])
    right = right / np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, eye
    return torch.from_numpy(c2w.astype(np.float32))


def intrinsics_from_fov(fovy_deg, H, W):
    focal = H / (2.0 * math.tan(math.radians(fovy_deg) / 2.0))
    return (focal, focal, W / 2.0, H / 2.0)


def get_rays(poses, intrinsics, H, W):
    """HIP ray generation: poses [B,4,4] on the GPU -> {'rays_o','rays_d'} [B, H*W, 3]."""
    rays_o, rays_d = _rm.get_rays(poses, intrinsics, H, W)
    return {"rays_o": rays_o, "rays_d": rays_d}
