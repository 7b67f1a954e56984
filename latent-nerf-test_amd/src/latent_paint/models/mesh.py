"""Mesh container of the Latent-Paint path: kaolin-free counterpart of src/latent_paint/models/mesh.py
(OBJ import :10-17, normalize_mesh :37-48): plain-text OBJ reader with UVs."""
import torch


class Mesh:
    def __init__(self, obj_path=None, device="cpu", vertices=None, faces=None, vt=None, ft=None):
        if obj_path is not None:
            if not str(obj_path).endswith(".obj"):
                raise ValueError("%s extension not implemented in mesh reader." % obj_path)
            vertices, faces, vt, ft = _read_obj(obj_path)
        self.vertices = vertices.to(device).float()
        self.faces = faces.to(device).long()
        self.vt = None if vt is None else vt.to(device).float()
        self.ft = None if ft is None else ft.to(device).long()

    def normalize_mesh(self, inplace=False, target_scale=1, dy=0):
        mesh = self if inplace else Mesh(vertices=self.vertices.clone(), faces=self.faces, vt=self.vt, ft=self.ft,
                                         device=self.vertices.device)
        verts = mesh.vertices
        verts = verts - verts.mean(dim=0)
        verts = verts / torch.max(torch.norm(verts, p=2, dim=1))
        verts = verts * target_scale
        verts[:, 1] += dy
        mesh.vertices = verts
        return mesh


def _read_obj(path):
    v, vt, f, ft = [], [], [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith("v "):
                v.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("vt "):
                vt.append([float(x) for x in line.split()[1:3]])
            elif line.startswith("f "):
                vi, ti = [], []
                for tok in line.split()[1:]:
                    parts = tok.split("/")
                    i = int(parts[0])
                    vi.append(i - 1 if i > 0 else len(v) + i)
                    if len(parts) > 1 and parts[1] != "":
                        t = int(parts[1])
                        ti.append(t - 1 if t > 0 else len(vt) + t)
                    else:
                        ti.append(-1)
                for k in range(1, len(vi) - 1):
                    f.append([vi[0], vi[k], vi[k + 1]])
                    ft.append([ti[0], ti[k], ti[k + 1]])
    V = torch.tensor(v, dtype=torch.float32)
    Fc = torch.tensor(f, dtype=torch.int64)
    VT = torch.tensor(vt, dtype=torch.float32) if vt else None
    FT = torch.tensor(ft, dtype=torch.int64) if vt else None
    return V, Fc, VT, FT
