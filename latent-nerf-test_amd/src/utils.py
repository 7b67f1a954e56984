"""Counterpart of the reference's src/utils.py (get_view_direction :8-27, tensor2numpy :57-62,
make_path :64-66, seed_everything :68-73), behaviour pinned by tests/golden/utils_golden.json
(generated from the reference module itself)."""
import os
import random
from pathlib import Path

import numpy as np
import torch

_TWO_PI = 2.0 * np.pi


def _deg_wrapped(x):
    return np.deg2rad(x) % _TWO_PI


def get_view_direction(elev, azim, top=30, front=0, angle=45):
    """6-way view bucket (0 front, 1 left, 2 back, 3 right, 4 overhead, 5 bottom).

    `elev`/`azim` are radians; `top`, `front`, `angle` are interpreted as degrees.  The
    reference's callers pass already-converted radians for `top`/`front`
    (src/latent_paint/training/views_dataset.py:12-22), so the cone is much narrower than the
    config suggests; this mirror keeps that observable behaviour."""
    azim = azim % _TWO_PI
    elev = elev % _TWO_PI
    lo_front, hi_front = _deg_wrapped(front - angle), _deg_wrapped(front + angle)
    lo_back, hi_back = _deg_wrapped(front + 180 - angle), _deg_wrapped(front + 180 + angle)
    view = torch.zeros(elev.shape[0], dtype=torch.long)
    view[(azim >= lo_front) | (azim < hi_front)] = 0
    view[(azim >= hi_back) & (azim < lo_front)] = 1
    view[(azim >= lo_back) & (azim < hi_back)] = 2
    view[(azim >= hi_front) & (azim < lo_back)] = 3
    view[elev < _deg_wrapped(top)] = 4
    view[elev > _deg_wrapped(180 - top)] = 5
    return view


def view_direction_index(elev, azim, top=30, front=0, angle=45) -> int:
    """get_view_direction for ONE view given as Python floats: the same comparisons on the same float32-rounded values
    (a float32 tensor compared with a Python scalar compares in float32), without building one-element tensors -- this
    runs once per training step on the host.  Checked against the tensor form in tests/test_utils_golden.py."""
    f32 = np.float32
    two_pi = f32(_TWO_PI)
    az = np.fmod(f32(azim), two_pi)
    az = az + two_pi if az < 0 else az
    el = np.fmod(f32(elev), two_pi)
    el = el + two_pi if el < 0 else el
    lo_front, hi_front = f32(_deg_wrapped(front - angle)), f32(_deg_wrapped(front + angle))
    lo_back, hi_back = f32(_deg_wrapped(front + 180 - angle)), f32(_deg_wrapped(front + 180 + angle))
    view = 0
    if az >= lo_front or az < hi_front:
        view = 0
    if az >= hi_back and az < lo_front:
        view = 1
    if az >= lo_back and az < hi_back:
        view = 2
    if az >= hi_front and az < lo_back:
        view = 3
    if el < f32(_deg_wrapped(top)):
        view = 4
    if el > f32(_deg_wrapped(180 - top)):
        view = 5
    return view


def tensor2numpy(tensor: torch.Tensor) -> np.ndarray:
    arr = tensor.detach().cpu().numpy()
    if arr.min() < 0:
        arr = arr * 0.5 + 0.5
    return (arr * 255).astype(np.uint8)


def make_path(path: Path) -> Path:
    path.mkdir(exist_ok=True, parents=True)
    return path


def seed_everything(seed: int) -> None:
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def write_video(stem: Path, frames, fps: int = 25) -> Path:
    """uint8 frames [T,H,W,3] -> `<stem>.mp4` through imageio when it is importable (what the reference's trainers call,
    src/latent_paint/training/trainer.py:170-172), else an animated `<stem>.gif` through Pillow.  Returns the file."""
    frames = [np.asarray(f, dtype=np.uint8) for f in frames]
    try:
        import imageio
        out = Path(str(stem) + ".mp4")
        imageio.mimsave(out, np.stack(frames, axis=0), fps=fps, quality=8, macro_block_size=1)
    except ImportError:
        from PIL import Image
        out = Path(str(stem) + ".gif")
        ims = [Image.fromarray(f) for f in frames]
        ims[0].save(out, save_all=True, append_images=ims[1:], duration=int(round(1000 / fps)), loop=0)
    return out
