"""CPU oracle for the sketch-shape guidance kernels (csrc/mesh.hip).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference only names the dependency (igl, README.md:119-122) and ships neither the
shape-loss code nor vectors.  This restates the published definitions: generalised winding number as
the sum of triangle solid angles (van Oosterom & Strackee 1983) and the exact point-triangle distance."""
import numpy as np


def winding_number(points: np.ndarray, tris: np.ndarray) -> np.ndarray:
    """points [n,3], tris [F,3,3] -> [n] (float64 arithmetic)."""
    p = points.astype(np.float64)[:, None, :]
    a = tris[None, :, 0, :].astype(np.float64) - p
    b = tris[None, :, 1, :].astype(np.float64) - p
    c = tris[None, :, 2, :].astype(np.float64) - p
    la, lb, lc = np.linalg.norm(a, axis=-1), np.linalg.norm(b, axis=-1), np.linalg.norm(c, axis=-1)
    det = np.einsum("nfi,nfi->nf", a, np.cross(b, c))
    den = la * lb * lc + np.einsum("nfi,nfi->nf", a, b) * lc + np.einsum("nfi,nfi->nf", b, c) * la \
        + np.einsum("nfi,nfi->nf", c, a) * lb
    return (2.0 * np.arctan2(det, den)).sum(-1) / (4.0 * np.pi)


def _closest_on_segment(p, a, b):
    ab = b - a
    t = np.clip(np.einsum("...i,...i->...", p - a, ab) / np.maximum(np.einsum("...i,...i->...", ab, ab), 1e-30), 0, 1)
    return a + t[..., None] * ab


def distance(points: np.ndarray, tris: np.ndarray) -> np.ndarray:
    """Unsigned distance of points [n,3] to the triangle soup [F,3,3] (projection onto the plane if the foot
    is inside the triangle, else the nearest of the three edges)."""
    p = points.astype(np.float64)[:, None, :]
    a, b, c = (tris[None, :, k, :].astype(np.float64) for k in range(3))
    nrm = np.cross(b - a, c - a)
    nn = np.maximum(np.linalg.norm(nrm, axis=-1, keepdims=True), 1e-30)
    nrm = nrm / nn
    dist_plane = np.einsum("nfi,nfi->nf", p - a, nrm)
    foot = p - dist_plane[..., None] * nrm

    def same_side(u, v, w):  # foot on the inner side of edge u->v (w is the opposite vertex)
        return np.einsum("nfi,nfi->nf", np.cross(v - u, foot - u), np.cross(v - u, w - u)) >= 0

    inside = same_side(a, b, c) & same_side(b, c, a) & same_side(c, a, b)
    d_edges = np.minimum.reduce([np.linalg.norm(p - _closest_on_segment(p, u, v), axis=-1)
                                 for u, v in ((a, b), (b, c), (c, a))])
    d = np.where(inside, np.abs(dist_plane), d_edges)
    return d.min(-1)
