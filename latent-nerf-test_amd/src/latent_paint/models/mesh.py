"""Triangle-mesh container of the Latent-Paint path, kaolin-free: stands in for src/latent_paint/models/mesh.py
(Mesh(obj_path, device) :7-24 with `.vertices .faces .vt .ft`, normalize_mesh :37-48).  The reference parses
files through kal.io.obj / kal.io.off (:12-17); here a plain-text reader for both formats."""
import torch


def read_obj(path):
    """Wavefront OBJ -> (v [V,3] f32, f [F,3] i64, vt [VT,2] f32 | None, ft [F,3] i64 | None).
    Polygons are fanned into triangles; negative (relative) indices are resolved; a face corner without a
    texture index gets -1 in `ft` (the reference's test `ft.min() > -1`, textured_mesh.py:84-85, then fails
    and the caller falls back to the cached / generated UV atlas)."""
    pos, tex, tri, tri_uv = [], [], [], []
    with open(path) as fh:
        for line in fh:
            tag = line[:2]
            if tag == "v ":
                pos.append([float(x) for x in line.split()[1:4]])
            elif tag == "vt":
                tex.append([float(x) for x in line.split()[1:3]])
            elif tag == "f ":
                corner_v, corner_t = [], []
                for token in line.split()[1:]:
                    ref = token.split("/")
                    k = int(ref[0])
                    corner_v.append(k - 1 if k > 0 else len(pos) + k)
                    if len(ref) > 1 and ref[1]:
                        k = int(ref[1])
                        corner_t.append(k - 1 if k > 0 else len(tex) + k)
                    else:
                        corner_t.append(-1)
                for j in range(1, len(corner_v) - 1):
                    tri.append([corner_v[0], corner_v[j], corner_v[j + 1]])
                    tri_uv.append([corner_t[0], corner_t[j], corner_t[j + 1]])
    if not pos or not tri:
        raise ValueError("%s: no geometry found" % path)
    v = torch.tensor(pos, dtype=torch.float32)
    f = torch.tensor(tri, dtype=torch.int64)
    if not tex:
        return v, f, None, None
    return v, f, torch.tensor(tex, dtype=torch.float32), torch.tensor(tri_uv, dtype=torch.int64)


def read_off(path):
    """Object File Format (the reference accepts '.off' through kal.io.off.import_mesh, mesh.py:16-17)."""
    with open(path) as fh:
        tokens = fh.read().split()
    if not tokens or not tokens[0].startswith("OFF"):
        raise ValueError("%s: not an OFF file" % path)
    head = tokens[0][3:]
    tokens = ([head] if head else []) + tokens[1:]   # "OFF8 12 0" (no separator) occurs in ModelNet files
    nv, nf = int(tokens[0]), int(tokens[1])
    cur = 3
    v = torch.tensor([float(t) for t in tokens[cur:cur + 3 * nv]], dtype=torch.float32).reshape(nv, 3)
    cur += 3 * nv
    tri = []
    for _ in range(nf):
        n = int(tokens[cur])
        idx = [int(t) for t in tokens[cur + 1:cur + 1 + n]]
        cur += 1 + n
        for j in range(1, n - 1):
            tri.append([idx[0], idx[j], idx[j + 1]])
    return v, torch.tensor(tri, dtype=torch.int64), None, None


class Mesh:
    def __init__(self, obj_path=None, device="cpu", *, vertices=None, faces=None, vt=None, ft=None):
        if obj_path is not None:
            name = str(obj_path)
            if ".obj" in name:
                vertices, faces, vt, ft = read_obj(name)
            elif ".off" in name:
                vertices, faces, vt, ft = read_off(name)
            else:
                raise ValueError("%s extension not implemented in mesh reader." % obj_path)
        self.vertices = vertices.to(device).float()
        self.faces = faces.to(device).long()
        self.vt = None if vt is None else vt.to(device).float()
        self.ft = None if ft is None else ft.to(device).long()

    def _copy(self):
        return Mesh(vertices=self.vertices.clone(), faces=self.faces, vt=self.vt, ft=self.ft,
                    device=self.vertices.device)

    def normalize_mesh(self, inplace=False, target_scale=1, dy=0):
        """Centroid to the origin, farthest vertex at distance `target_scale`, then lifted by `dy` along +y."""
        out = self if inplace else self._copy()
        centred = out.vertices - out.vertices.mean(dim=0)
        radius = centred.norm(p=2, dim=1).max()
        placed = (centred / radius) * target_scale   # op order of the reference: bit-equal vertex positions
        placed[:, 1] += dy
        out.vertices = placed
        return out

    def standardize_mesh(self, inplace=False):
        """Centroid to the origin, unit standard deviation of the vertex radii (mesh.py:26-35)."""
        out = self if inplace else self._copy()
        centred = out.vertices - out.vertices.mean(dim=0)
        out.vertices = centred / centred.norm(p=2, dim=1).std()
        return out
