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


def pose_values(theta, phi, radius, target=(0.0, 0.0, 0.0)):
    """The 16 values (row-major) of the camera-to-world matrix.  Columns: right, down, forward, eye.
    Plain double arithmetic on Python floats (this runs once per view and training step on the host: the numpy form
    of the same formulas -- cross, norm, eye -- cost 0.2 ms per call in array overheads)."""
    st, ct = math.sin(theta), math.cos(theta)
    ex, ey, ez = radius * st * math.sin(phi), radius * ct, radius * st * math.cos(phi)
    fx, fy, fz = target[0] - ex, target[1] - ey, target[2] - ez
    n = max(math.sqrt(fx * fx + fy * fy + fz * fz), 1e-20)
    fx, fy, fz = fx / n, fy / n, fz / n
    # right = fwd x up, up = (0, 1, 0)
    rx, ry, rz = fy * 0.0 - fz * 1.0, fz * 0.0 - fx * 0.0, fx * 1.0 - fy * 0.0
    n = math.sqrt(rx * rx + ry * ry + rz * rz)
    if n < 1e-8:
        rx, ry, rz, n = 1.0, 0.0, 0.0, 1.0
    rx, ry, rz = rx / n, ry / n, rz / n
    # down = fwd x right
    dx, dy, dz = fy * rz - fz * ry, fz * rx - fx * rz, fx * ry - fy * rx
    return (rx, dx, fx, ex, ry, dy, fy, ey, rz, dz, fz, ez, 0.0, 0.0, 0.0, 1.0)


def pose_from_angles(theta, phi, radius, target=(0.0, 0.0, 0.0)):
    """Camera-to-world [4,4] float32 (CPU tensor) of pose_values()."""
    return torch.tensor(pose_values(theta, phi, radius, target), dtype=torch.float32).view(4, 4)


def intrinsics_from_fov(fovy_deg, H, W):
    focal = H / (2.0 * math.tan(math.radians(fovy_deg) / 2.0))
    return (focal, focal, W / 2.0, H / 2.0)


def get_rays(poses, intrinsics, H, W):
    """HIP ray generation: poses [B,4,4] on the GPU -> {'rays_o','rays_d'} [B, H*W, 3]."""
    rays_o, rays_d = _rm.get_rays(poses, intrinsics, H, W)
    return {"rays_o": rays_o, "rays_d": rays_d}
