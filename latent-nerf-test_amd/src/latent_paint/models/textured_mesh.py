"""TexturedMeshModel of the Latent-Paint path on the HIP raster kernels (csrc/raster.hip).

Drop-in for the reference's class of the same name: it is constructed exactly as the reference trainer
constructs it (src/latent_paint/training/trainer.py:59-60: `TexturedMeshModel(cfg, device=...,
render_grid_size=..., latent_mode=..., texture_resolution=...)`, signature src/latent_paint/models/
textured_mesh.py:16-22), exposes the members that trainer touches (`latent_mode`, `get_params()` :114-118,
`render(theta, phi, radius, decode_func, test, dims)` :181-240, `export_mesh(path, guidance)` :120-179) and keeps
the checkpoint keys (`background_sphere_colors`, `texture_img`, `texture_img_rgb_finetune`, :60-79).  What it
computes: a learnable 4-channel latent texture [1,4,R,R] looked up through the mesh's UV map, over a learnable
background = per-face-vertex colours [1,F_env,3,4] of a large sphere around the scene."""
import os
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .mesh import Mesh
from .render import Renderer

# latent -> RGB linear estimate (rows = latent channels), the table of textured_mesh.py:34-40
LATENT_RGB_ROWS = ((0.298, 0.207, 0.208), (0.187, 0.286, 0.173), (-0.158, 0.189, 0.264), (-0.184, -0.271, -0.473))
LATENT_SIDE = 64   # side of the latent image the diffusion model consumes (src/stable_diffusion.py:259)


def per_triangle_atlas(n_faces, device):
    """UV atlas that needs no unwrapping library: the unit square is cut into n x n cells, two triangles per cell
    (lower-left and upper-right half, with a margin so neighbouring charts do not bleed).  Every face gets three
    texture vertices of its own: vt [3F,2], ft [F,3]."""
    n = int(np.ceil(np.sqrt((n_faces + 1) // 2)))
    k = torch.arange(n_faces, device=device)
    cell, upper = k // 2, (k % 2).float()[:, None]
    org = torch.stack([(cell % n).float(), (cell // n).float()], -1)
    lo, hi, m = 0.08, 0.92, 0.06
    lower_tri = torch.tensor([[lo, lo], [hi - m, lo], [lo, hi - m]], device=device)
    upper_tri = torch.tensor([[hi, hi], [lo + m, hi], [hi, lo + m]], device=device)
    corners = lower_tri[None] * (1 - upper[..., None]) + upper_tri[None] * upper[..., None]   # [F,3,2]
    vt = ((org[:, None, :] + corners) / n).reshape(-1, 2)
    ft = torch.arange(3 * n_faces, device=device).reshape(n_faces, 3)
    return vt.float(), ft.long()


class TexturedMeshModel(nn.Module):
    def __init__(self, opt, render_grid_size=64, latent_mode=True, texture_resolution=128,
                 device=torch.device("cpu")):
        super().__init__()
        self.opt = opt
        self.device = torch.device(device)
        self.latent_mode = latent_mode
        self.dy = opt.guide.dy
        self.mesh_scale = opt.guide.shape_scale
        self.texture_resolution = int(texture_resolution)
        self.linear_rgb_estimator = torch.tensor(LATENT_RGB_ROWS, device=self.device)
        self.renderer = Renderer(device=self.device, dim=(render_grid_size, render_grid_size),
                                 interpolation_mode=opt.guide.texture_interpolation_mode)
        self.env_sphere, self.mesh = self.init_meshes()
        self.background_sphere_colors, self.texture_img, self.texture_img_rgb_finetune = self.init_paint()
        self.vt, self.ft = self.init_texture_map()
        # per-face-vertex UVs [1,F,3,2] (kal.ops.mesh.index_vertices_by_faces in the reference, :48-50)
        self.face_attributes = self.vt[self.ft.long()][None].detach()

    # ------------------------------------------------------------------ construction
    def init_meshes(self, env_sphere_path="shapes/env_sphere.obj"):
        """Background sphere + the mesh to paint (scaled into the unit cube, lifted by dy).  The sphere is read from
        the reference's fixture when the working directory has it (`shapes/env_sphere.obj`: 5120 faces, radius 20);
        otherwise the same icosphere is generated."""
        if os.path.exists(env_sphere_path):
            env = Mesh(env_sphere_path, self.device)
        else:
            from ...latent_nerf.training.shape import make_icosphere
            ev, ef = make_icosphere(4, 20.0)
            env = Mesh(vertices=ev, faces=ef, device=self.device)
        shape = Mesh(self.opt.guide.shape_path, self.device)
        shape.normalize_mesh(inplace=True, target_scale=self.mesh_scale, dy=self.dy)
        return env, shape

    def init_paint(self, init_rgb_color=(1.0, 0.0, 0.0)):
        """Learnable state: background colours ~ U(0,1); latent texture = 0.3 x (the latent whose linear RGB
        estimate is `init_rgb_color`, ridge-regularised least squares) + 0.4 x N(0,1); an RGB texture that is only
        used by the 'texture-rgb-mesh' fine-tuning backbone (filled from a checkpoint)."""
        dev, R = self.device, self.texture_resolution
        sky = nn.Parameter(torch.rand(1, self.env_sphere.faces.shape[0], 3, 4, device=dev))
        M = self.linear_rgb_estimator                     # [4,3]: rgb = latent @ M
        ridge = M @ M.T + 1e-2 * torch.eye(4, device=dev)
        seed_latent = torch.linalg.pinv(ridge) @ M @ torch.tensor(init_rgb_color, device=dev)
        latent_tex = nn.Parameter(0.3 * seed_latent.view(1, 4, 1, 1) + 0.4 * torch.randn(1, 4, R, R, device=dev))
        rgb_tex = nn.Parameter(torch.zeros(1, 3, R, R, device=dev))
        return sky, latent_tex, rgb_tex

    def init_texture_map(self):
        """UV map, in the reference's order of preference (:81-109): the mesh's own UVs when every face corner
        has one; else `vt.pth` / `ft.pth` cached in the experiment directory; else a fresh atlas (xatlas when
        importable, as the reference; otherwise the built-in per-triangle atlas), which is then cached."""
        mesh = self.mesh
        if mesh.vt is not None and mesh.ft is not None and mesh.vt.shape[0] > 0 and int(mesh.ft.min()) > -1:
            return mesh.vt.to(self.device), mesh.ft.to(self.device)
        cache = Path(self.opt.log.exp_dir)
        vt_file, ft_file = cache / "vt.pth", cache / "ft.pth"
        if vt_file.exists() and ft_file.exists():
            return (torch.load(vt_file, weights_only=True).to(self.device),
                    torch.load(ft_file, weights_only=True).to(self.device))
        try:
            import xatlas
        except ImportError:
            vt, ft = per_triangle_atlas(mesh.faces.shape[0], self.device)
        else:
            atlas = xatlas.Atlas()
            atlas.add_mesh(mesh.vertices.cpu().numpy(), mesh.faces.int().cpu().numpy())
            chart = xatlas.ChartOptions()
            chart.max_iterations = 4
            atlas.generate(chart_options=chart)
            _, ft_np, vt_np = atlas[0]
            vt = torch.from_numpy(vt_np.astype(np.float32)).to(self.device)
            ft = torch.from_numpy(ft_np.astype(np.int64)).to(self.device)
        cache.mkdir(parents=True, exist_ok=True)
        torch.save(vt.cpu(), vt_file)
        torch.save(ft.cpu(), ft_file)
        return vt, ft

    def forward(self, x):
        raise NotImplementedError("TexturedMeshModel is driven through render()")

    def get_params(self):
        """What the optimiser trains: the background colours plus the texture of the active backbone."""
        texture = self.texture_img if self.latent_mode else self.texture_img_rgb_finetune
        return [self.background_sphere_colors, texture]

    # ------------------------------------------------------------------ rendering
    def _view(self, theta, phi, radius):
        return {"elev": theta, "azim": phi, "radius": radius, "look_at_height": self.dy}

    def render(self, theta, phi, radius, decode_func=None, test=False, dims=None):
        if test:
            return self.render_test(theta, phi, radius, decode_func, dims=dims)
        return self.render_train(theta, phi, radius)

    def render_train(self, theta, phi, radius):
        """-> {'image' [1,C,h,w], 'mask' [1,1,h,w], 'background', 'foreground'}; C = 4 latents (or 3 RGB when
        fine-tuning).  The rasterisation itself carries no gradient: 'image' is differentiable w.r.t. the texture
        (under the mesh) and the background colours (elsewhere)."""
        if self.latent_mode:
            texture, sky_colors = self.texture_img, self.background_sphere_colors
        else:
            texture = self.texture_img_rgb_finetune
            sky_colors = self.background_sphere_colors @ self.linear_rgb_estimator
        view = self._view(theta, phi, radius)
        foreground, coverage = self.renderer.render_single_view_texture(
            self.mesh.vertices, self.mesh.faces, self.face_attributes, texture, **view)
        background, _ = self.renderer.render_single_view(self.env_sphere, sky_colors, **view)
        coverage = coverage.detach()
        image = background * (1 - coverage) + foreground * coverage
        out = {"image": image, "mask": coverage, "background": background, "foreground": foreground}
        if self.latent_mode and coverage.shape[-1] != LATENT_SIDE:
            # the diffusion model takes LATENT_SIDE x LATENT_SIDE latents whatever the render size
            out = {k: F.interpolate(v, (LATENT_SIDE, LATENT_SIDE), mode="bicubic") for k, v in out.items()}
        return out

    def render_test(self, theta, phi, radius, decode_func=None, dims=None):
        """Evaluation view: the latent texture is decoded to RGB first (`decode_func` = the guidance's
        `decode_latents`), then rendered at `dims` on a white background."""
        if self.latent_mode:
            if decode_func is None:
                raise ValueError("decode function was not supplied to decode the latent texture image")
            texture = decode_func(self.texture_img)
        else:
            texture = self.texture_img_rgb_finetune
        image, coverage = self.renderer.render_single_view_texture(
            self.mesh.vertices, self.mesh.faces, self.face_attributes, texture, dims=dims, white_background=True,
            **self._view(theta, phi, radius))
        return {"image": image, "texture_map": texture, "mask": coverage}

    # ------------------------------------------------------------------ export
    @torch.no_grad()
    def export_mesh(self, path, guidance=None):
        """Writes `albedo.png` (the decoded texture), `mesh.obj` (v / vt / f v/vt) and `mesh.mtl` into `path`."""
        from PIL import Image
        path = Path(path)
        path.mkdir(parents=True, exist_ok=True)
        if self.latent_mode:
            if guidance is None:
                raise ValueError("export_mesh needs the guidance model to decode the latent texture")
            from ...latent_nerf.training.guidance import decode_with
            rgb = decode_with(guidance, self.texture_img)
        else:
            rgb = self.texture_img_rgb_finetune
        albedo = (rgb[0].permute(1, 2, 0).clamp(0, 1).cpu().numpy() * 255).astype(np.uint8)
        Image.fromarray(albedo).save(path / "albedo.png")
        v = self.mesh.vertices.cpu().numpy()
        f = self.mesh.faces.cpu().numpy() + 1
        vt = self.vt.cpu().numpy()
        ft = self.ft.cpu().numpy() + 1
        lines = ["mtllib mesh.mtl"]
        lines += ["v %s %s %s" % (p[0], p[1], p[2]) for p in v]
        lines += ["vt %s %s" % (t[0], t[1]) for t in vt]
        lines.append("usemtl mat0")
        lines += ["f %d/%d %d/%d %d/%d" % (a[0], b[0], a[1], b[1], a[2], b[2]) for a, b in zip(f, ft)]
        (path / "mesh.obj").write_text("\n".join(lines) + "\n")
        (path / "mesh.mtl").write_text("newmtl mat0\nKa 1.000000 1.000000 1.000000\nKd 1.000000 1.000000 1.000000\n"
                                       "Ks 0.000000 0.000000 0.000000\nTr 1.000000\nillum 1\nNs 0.000000\n"
                                       "map_Kd albedo.png\n")
