"""Camera sampling of the Latent-Paint trainer: stands in for src/latent_paint/training/views_dataset.py
(rand_poses :9-22, circle_poses :25-35, ViewsDataset :38-80).  One view per item, as dicts
{'dir', 'theta', 'phi', 'radius'}; training views are drawn on the fly (radius ~ U[radius_range],
theta ~ U[0, 150] deg, phi ~ U[0, 360) deg), evaluation views go round a circle at theta = 60 deg,
radius = 1.2 x radius_range[1].  The view bucket comes from src.utils.get_view_direction, called the
way the reference calls it (already-converted radians for `angle_overhead` / `angle_front`)."""
import numpy as np
import torch

from ...utils import get_view_direction

THETA_RANGE_DEG = (0.0, 150.0)
PHI_RANGE_DEG = (0.0, 360.0)


def rand_poses(size, device, radius_range=(1.0, 1.5), theta_range=THETA_RANGE_DEG, phi_range=PHI_RANGE_DEG,
               angle_overhead=30.0, angle_front=60.0, generator=None):
    """-> (dirs [size] long, theta, phi, radius as Python floats; size is 1 everywhere in the trainer)."""
    t0, t1 = np.deg2rad(theta_range)
    p0, p1 = np.deg2rad(phi_range)
    u = torch.rand(3, size, generator=generator)
    radius = radius_range[0] + u[0] * (radius_range[1] - radius_range[0])
    thetas = t0 + u[1] * (t1 - t0)
    phis = p0 + u[2] * (p1 - p0)
    dirs = get_view_direction(thetas, phis, np.deg2rad(angle_overhead), np.deg2rad(angle_front))
    return dirs.to(device), thetas.item(), phis.item(), radius.item()


def circle_poses(device, radius=1.25, theta=60.0, phi=0.0, angle_overhead=30.0, angle_front=60.0):
    theta, phi = float(np.deg2rad(theta)), float(np.deg2rad(phi))
    dirs = get_view_direction(torch.tensor([theta]), torch.tensor([phi]), np.deg2rad(angle_overhead),
                              np.deg2rad(angle_front))
    return dirs.to(device), theta, phi, radius


class _ViewLoader:
    """What `ViewsDataset.dataloader()` returns: iterating yields `size` views (a fresh order / fresh random
    poses on every pass, like DataLoader(batch_size=1, shuffle=training)); `_data` is the dataset."""

    def __init__(self, dataset):
        self._data = dataset

    def __len__(self):
        return self._data.size

    def __iter__(self):
        ds = self._data
        order = torch.randperm(ds.size, generator=ds.generator).tolist() if ds.training else range(ds.size)
        for i in order:
            yield ds.collate([i])


class ViewsDataset:
    def __init__(self, cfg, device, type="train", size=100, seed=None):
        self.cfg, self.device, self.type, self.size = cfg, device, type, size
        self.training = type in ("train", "all")
        self.generator = None if seed is None else torch.Generator().manual_seed(seed)

    def collate(self, index):
        cfg = self.cfg
        if self.training:
            dirs, theta, phi, radius = rand_poses(len(index), self.device, radius_range=cfg.radius_range,
                                                  angle_overhead=cfg.angle_overhead, angle_front=cfg.angle_front,
                                                  generator=self.generator)
        else:
            dirs, theta, phi, radius = circle_poses(self.device, radius=cfg.radius_range[1] * 1.2, theta=60,
                                                    phi=(index[0] / self.size) * 360,
                                                    angle_overhead=cfg.angle_overhead, angle_front=cfg.angle_front)
        return {"dir": dirs, "theta": theta, "phi": phi, "radius": radius}

    def dataloader(self):
        return _ViewLoader(self)
