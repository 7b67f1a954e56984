"""CPU oracle for the Latent-Paint raster path (csrc/raster.hip).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED at the kaolin boundary: the reference calls kaolin (un-pinned git HEAD, setup.sh:3;
call sites src/latent_paint/models/render.py:11,30,39-43,56-64) and holds no tests or vectors for it, and
kaolin is not installed here.  This restates the documented semantics of those ops in plain PyTorch:
perspective projection with fov pi/3, look-at camera, hard z-buffer rasterisation with perspective-correct
barycentric interpolation of per-face-vertex attributes (face_idx = -1 on background), and texture mapping
= F.grid_sample(align_corners=False, padding_mode='border') on (u, 1 - v) after clamping uv to [0, 1]."""
import math

import torch
import torch.nn.functional as F


def camera_from_view(elev, azim, r, look_at_height=0.0):
    """Eye position as src/latent_paint/models/render.py:19-23; returns (rot [3,3] rows = camera axes, pos [3])."""
    pos = torch.tensor([r * math.sin(elev) * math.sin(azim), r * math.cos(elev), r * math.sin(elev) * math.cos(azim)],
                       dtype=torch.float64)
    look = torch.tensor([0.0, look_at_height, 0.0], dtype=torch.float64)
    up = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64)
    z = pos - look
    z = z / z.norm()
    x = torch.linalg.cross(up, z)
    x = x / x.norm()
    y = torch.linalg.cross(z, x)
    return torch.stack([x, y, z]).float(), pos.float()


def prepare_vertices(verts, faces, rot, pos, fov=math.pi / 3):
    cam = (verts - pos) @ rot.T                       # [V,3]
    f = 1.0 / math.tan(fov / 2)
    fv = cam[faces]                                   # [F,3,3]
    z = fv[..., 2]
    xy = fv[..., :2] * f / (-z[..., None])
    return z, xy


def rasterize(H, W, face_z, face_xy):
    """-> face_idx [H*W] (long, -1 background), bary [H*W,3] perspective-correct."""
    j = torch.arange(W, dtype=torch.float32)
    i = torch.arange(H, dtype=torch.float32)
    px = ((2 * j + 1) / W - 1)[None, :].expand(H, W).reshape(-1)
    py = (1 - (2 * i + 1) / H)[:, None].expand(H, W).reshape(-1)
    x0, y0 = face_xy[:, 0, 0][None], face_xy[:, 0, 1][None]
    x1, y1 = face_xy[:, 1, 0][None], face_xy[:, 1, 1][None]
    x2, y2 = face_xy[:, 2, 0][None], face_xy[:, 2, 1][None]
    PX, PY = px[:, None], py[:, None]
    area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0)
    e0 = (x1 - PX) * (y2 - PY) - (x2 - PX) * (y1 - PY)
    e1 = (x2 - PX) * (y0 - PY) - (x0 - PX) * (y2 - PY)
    w0 = e0 / area
    w1 = e1 / area
    w2 = 1 - w0 - w1
    z0, z1, z2 = face_z[:, 0][None], face_z[:, 1][None], face_z[:, 2][None]
    ok = (w0 >= 0) & (w1 >= 0) & (w2 >= 0) & (area != 0) & (z0 < 0) & (z1 < 0) & (z2 < 0)
    q0, q1, q2 = w0 / z0, w1 / z1, w2 / z2
    z = 1.0 / (q0 + q1 + q2)
    z = torch.where(ok, z, torch.full_like(z, -3.0e38))
    best_z, best_f = z.max(dim=1)                     # first max = lowest face index on ties
    hit = best_z > -1.0e38
    idx = torch.where(hit, best_f, torch.full_like(best_f, -1))
    g = best_f[:, None]
    b = torch.stack([torch.gather(q0 * z, 1, g)[:, 0], torch.gather(q1 * z, 1, g)[:, 0],
                     torch.gather(q2 * z, 1, g)[:, 0]], -1)
    b = torch.where(hit[:, None], b, torch.zeros_like(b))
    return idx, b


def interpolate(face_idx, bary, attr):
    """attr [F,3,D] -> [P,D]; differentiable w.r.t. attr."""
    safe = face_idx.clamp(min=0)
    a = attr[safe]                                    # [P,3,D]
    out = (bary[..., None] * a).sum(1)
    return torch.where((face_idx >= 0)[:, None], out, torch.zeros_like(out))


def texture_mapping(uv, tex, mode="nearest"):
    """uv [P,2], tex [1,C,R,R] -> [P,C]  (kaolin texture_mapping semantics)."""
    g = uv.clamp(0, 1) * 2 - 1
    g = torch.stack([g[:, 0], -g[:, 1]], -1).reshape(1, 1, -1, 2)
    out = F.grid_sample(tex, g, mode=mode, align_corners=False, padding_mode="border")
    return out[0, :, 0, :].T
