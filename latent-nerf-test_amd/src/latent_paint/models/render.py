"""HIP-backed counterpart of src/latent_paint/models/render.py (Renderer :5-69): same constructor, camera
convention (:19-31) and the two render entry points (:34-47, :50-69), with the kaolin calls replaced by the
C-ABI raster kernels (csrc/raster.hip)."""
import math

import torch

from ...latent_nerf.raymarching import backend as _b
from ...latent_nerf.raymarching.raymarching import _chk, _p, _stream

_MODES = {"nearest": 0, "bilinear": 1, "bicubic": 2}


class _InterpAttr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, face_idx, bary):
        P, D = face_idx.shape[0], attr.shape[-1]
        attr = attr.contiguous()
        feat = torch.empty(P, D, device=attr.device)
        _b.call("lnerf_interpolate_attributes", _p(face_idx), _p(bary), _chk(attr, "attr"), P, D, _p(feat), _stream())
        ctx.save_for_backward(face_idx, bary)
        ctx.shape = attr.shape
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        face_idx, bary = ctx.saved_tensors
        dattr = torch.zeros(ctx.shape, device=dfeat.device)
        _b.call("lnerf_interpolate_attributes_backward", _p(face_idx), _p(bary), _chk(dfeat.contiguous(), "dfeat"),
                face_idx.shape[0], ctx.shape[-1], _p(dattr), _stream())
        return dattr, None, None


class _TextureMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, uv, face_idx, mode):
        C, R = tex.shape[1], tex.shape[2]
        P = uv.shape[0]
        tex = tex.contiguous()
        out = torch.empty(P, C, device=tex.device)
        _b.call("lnerf_texture_map_forward", _chk(uv, "uv"), _p(face_idx), _chk(tex, "texture"), P, C, R, mode, _p(out),
                _stream())
        ctx.save_for_backward(uv, face_idx)
        ctx.meta = (tex.shape, mode)
        return out

    @staticmethod
    def backward(ctx, dout):
        uv, face_idx = ctx.saved_tensors
        shape, mode = ctx.meta
        dtex = torch.zeros(shape, device=dout.device)
        _b.call("lnerf_texture_map_backward", _p(uv), _p(face_idx), _chk(dout.contiguous(), "dout"), uv.shape[0],
                shape[1], shape[2], mode, _p(dtex), _stream())
        return dtex, None, None, None


class Renderer:
    def __init__(self, device, dim=(224, 224), interpolation_mode="nearest"):
        assert interpolation_mode in ["nearest", "bilinear", "bicubic"], "no interpolation mode %s" % interpolation_mode
        self.device = device
        self.interpolation_mode = interpolation_mode
        self.fov = math.pi / 3                         # kal.render.camera.generate_perspective_projection(np.pi / 3)
        self.dim = dim
        self.background = torch.ones(dim).to(device).float()

    @staticmethod
    def get_camera_from_view(elev, azim, r=3.0, look_at_height=0.0, fov=math.pi / 3):
        """14 floats for the C ABI: rotation rows, eye position, fx, fy."""
        elev, azim = float(elev), float(azim)
        pos = [r * math.sin(elev) * math.sin(azim), r * math.cos(elev), r * math.sin(elev) * math.cos(azim)]
        look = [0.0, look_at_height, 0.0]
        z = [pos[i] - look[i] for i in range(3)]
        n = math.sqrt(sum(c * c for c in z))
        z = [c / n for c in z]
        up = [0.0, 1.0, 0.0]
        x = [up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0]]
        n = math.sqrt(sum(c * c for c in x))
        x = [c / n for c in x]
        y = [z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]]
        f = 1.0 / math.tan(fov / 2)
        import ctypes
        return (ctypes.c_float * 14)(*(x + y + z + pos + [f, f]))

    def _rasterize(self, verts, faces, elev, azim, radius, look_at_height, dims):
        H, W = dims[1], dims[0]
        cam = self.get_camera_from_view(elev, azim, radius, look_at_height, self.fov)
        verts = verts.to(self.device).float().contiguous()
        faces32 = faces.to(self.device).to(torch.int32).contiguous()
        F = faces32.shape[0]
        face_z = torch.empty(F, 3, device=self.device)
        face_xy = torch.empty(F, 3, 2, device=self.device)
        _b.call("lnerf_raster_prepare", _chk(verts, "verts"), verts.shape[0], _chk(faces32, "faces", torch.int32), F, cam,
                _p(face_z), _p(face_xy), _stream())
        face_idx = torch.empty(H * W, device=self.device, dtype=torch.int32)
        bary = torch.empty(H * W, 3, device=self.device)
        _b.call("lnerf_rasterize", H, W, _p(face_z), _p(face_xy), F, _p(face_idx), _p(bary), _stream())
        return face_idx, bary, H, W

    def render_single_view(self, mesh, face_attributes, elev=0, azim=0, radius=2, look_at_height=0.0):
        """face_attributes [1,F,3,D] -> (image [1,D,H,W], mask [1,1,H,W]); differentiable w.r.t. the attributes."""
        face_idx, bary, H, W = self._rasterize(mesh.vertices, mesh.faces, elev, azim, radius, look_at_height, self.dim)
        feat = _InterpAttr.apply(face_attributes[0], face_idx, bary)
        mask = (face_idx > -1).float().reshape(1, H, W, 1)
        return feat.reshape(1, H, W, -1).permute(0, 3, 1, 2), mask.permute(0, 3, 1, 2)

    def render_single_view_texture(self, verts, faces, uv_face_attr, texture_map, elev=0, azim=0, radius=2,
                                   look_at_height=0.0, dims=None, white_background=False):
        dims = self.dim if dims is None else dims
        face_idx, bary, H, W = self._rasterize(verts, faces, elev, azim, radius, look_at_height, dims)
        with torch.no_grad():                          # uv_features.detach() in the reference (:61)
            uv = _InterpAttr.apply(uv_face_attr[0].detach(), face_idx, bary).contiguous()
        image = _TextureMap.apply(texture_map, uv, face_idx, _MODES[self.interpolation_mode])
        mask = (face_idx > -1).float().reshape(1, H, W, 1)
        image = image.reshape(1, H, W, -1) * mask
        if white_background:
            image = image + 1 * (1 - mask)
        return image.permute(0, 3, 1, 2), mask.permute(0, 3, 1, 2)
