"""TexturedMeshModel of the Latent-Paint path, HIP-backed: counterpart of
src/latent_paint/models/textured_mesh.py (__init__ :16-50, init_paint :60-79, get_params :114-118,
render/render_train/render_test :181-240).  Learnable state: a 4-channel latent texture [1,4,R,R] and the
per-face-vertex colours of a background sphere [1,F_env,3,4]; `render()` returns the dict the trainer
consumes ({'image','mask','background','foreground'}, :220)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .mesh import Mesh
from .render import Renderer


def _icosphere(subdiv, radius):
    from ...latent_nerf.training.shape import make_icosphere
    return make_icosphere(subdiv, radius)


class TexturedMeshModel(nn.Module):
    def __init__(self, shape_path=None, mesh=None, render_grid_size=64, latent_mode=True, texture_resolution=128,
                 shape_scale=0.6, dy=0.25, texture_interpolation_mode="nearest", device=torch.device("cuda")):
        super().__init__()
        self.device = device
        self.latent_mode = latent_mode
        self.dy, self.mesh_scale = dy, shape_scale
        self.texture_resolution = texture_resolution
        self.linear_rgb_estimator = torch.tensor([[0.298, 0.207, 0.208], [0.187, 0.286, 0.173],
                                                  [-0.158, 0.189, 0.264], [-0.184, -0.271, -0.473]]).to(device)
        self.renderer = Renderer(device=device, dim=(render_grid_size, render_grid_size),
                                 interpolation_mode=texture_interpolation_mode)
        # background: the reference loads shapes/env_sphere.obj (V=2562, F=5120, radius 20 = icosphere level 4)
        ev, ef = _icosphere(4, 20.0)
        self.env_sphere = Mesh(vertices=ev, faces=ef, device=device)
        self.mesh = (mesh if mesh is not None else Mesh(shape_path, device)).normalize_mesh(
            inplace=False, target_scale=shape_scale, dy=dy)
        if self.mesh.vt is None or self.mesh.ft is None or int(self.mesh.ft.min()) < 0:
            raise ValueError("mesh has no complete UV map (the reference falls back to xatlas, "
                             "textured_mesh.py:91-108; supply UVs)")
        self.background_sphere_colors = nn.Parameter(torch.rand(1, self.env_sphere.faces.shape[0], 3, 4, device=device))
        A = self.linear_rgb_estimator.T
        init_rgb = torch.tensor([1.0, 0.0, 0.0], device=device)
        init_lat = (torch.pinverse(A.T @ A + 1e-2 * torch.eye(4, device=device)) @ A.T) @ init_rgb
        self.texture_img = nn.Parameter(init_lat[None, :, None, None] * 0.3
                                        + 0.4 * torch.randn(1, 4, texture_resolution, texture_resolution, device=device))
        self.texture_img_rgb_finetune = nn.Parameter(torch.zeros(1, 3, texture_resolution, texture_resolution,
                                                                 device=device))
        self.vt, self.ft = self.mesh.vt, self.mesh.ft
        self.face_attributes = self.vt[self.ft][None].detach()     # index_vertices_by_faces: [1,F,3,2]

    def get_params(self):
        if self.latent_mode:
            return [self.background_sphere_colors, self.texture_img]
        return [self.background_sphere_colors, self.texture_img_rgb_finetune]

    def render(self, theta, phi, radius, decode_func=None, test=False, dims=None):
        if test:
            return self.render_test(theta, phi, radius, decode_func, dims=dims)
        return self.render_train(theta, phi, radius)

    def render_train(self, theta, phi, radius):
        if self.latent_mode:
            texture_img, bg_colors = self.texture_img, self.background_sphere_colors
        else:
            texture_img = self.texture_img_rgb_finetune
            bg_colors = self.background_sphere_colors @ self.linear_rgb_estimator
        pred_features, mask = self.renderer.render_single_view_texture(
            self.mesh.vertices, self.mesh.faces, self.face_attributes, texture_img, elev=theta, azim=phi, radius=radius,
            look_at_height=self.dy)
        pred_back, _ = self.renderer.render_single_view(self.env_sphere, bg_colors, elev=theta, azim=phi, radius=radius,
                                                        look_at_height=self.dy)
        mask = mask.detach()
        pred_map = pred_back * (1 - mask) + pred_features * mask
        if self.latent_mode and mask.shape[-1] != 64:
            mask = F.interpolate(mask, (64, 64), mode="bicubic")
            pred_back = F.interpolate(pred_back, (64, 64), mode="bicubic")
            pred_features = F.interpolate(pred_features, (64, 64), mode="bicubic")
            pred_map = F.interpolate(pred_map, (64, 64), mode="bicubic")
        return {"image": pred_map, "mask": mask, "background": pred_back, "foreground": pred_features}

    def render_test(self, theta, phi, radius, decode_func=None, dims=None):
        if self.latent_mode:
            assert decode_func is not None, "decode function was not supplied to decode the latent texture image"
            texture_img = decode_func(self.texture_img)
        else:
            texture_img = self.texture_img_rgb_finetune
        pred_features, mask = self.renderer.render_single_view_texture(
            self.mesh.vertices, self.mesh.faces, self.face_attributes, texture_img, elev=theta, azim=phi, radius=radius,
            look_at_height=self.dy, dims=dims, white_background=True)
        return {"image": pred_features, "texture_map": texture_img, "mask": mask}
