"""Generate known-answer vectors from the reference's own `src/utils.py`.

Run in the build container only (the reference tree is not on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_utils_golden.py

It imports /root/reference/src/utils.py *by file path* (the only reference module whose
dependencies -- numpy, torch -- are installed here), evaluates `get_view_direction` and
`tensor2numpy` on fixed inputs and writes the inputs + outputs as JSON next to this
script.  The JSON is data (inputs and expected outputs); no reference source is copied.

The calling convention reproduced is the one the reference's own callers use
(src/latent_paint/training/views_dataset.py:9-22): thetas/phis in radians and
`angle_overhead`/`angle_front` *already converted to radians* before being passed as
`top`/`front`, which `get_view_direction` then converts a second time (src/utils.py:10,20-26).
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference/src/utils.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "utils_golden.json")


def main():
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("_ref_utils", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    cases = []
    rng = np.random.RandomState(1234)
    # (a) the SURVEY's hand-picked probes
    phis_deg = [0, 30, 44, 47, 90, 134, 137, 180, 224, 227, 270, 314, 317, 359]
    thetas_deg = [0.2, 0.5, 1, 30, 60, 179, 179.5, 179.9]
    for overhead, front in [(30.0, 70.0), (40.0, 70.0), (30.0, 60.0)]:
        th = np.deg2rad(np.full(len(phis_deg), 90.0)).astype(np.float32)
        ph = np.deg2rad(np.array(phis_deg, dtype=np.float64)).astype(np.float32)
        out = ref.get_view_direction(torch.from_numpy(th), torch.from_numpy(ph),
                                     np.deg2rad(overhead), np.deg2rad(front))
        cases.append(dict(kind="azim_sweep", overhead_deg=overhead, front_deg=front,
                          thetas=th.tolist(), phis=ph.tolist(), expect=out.tolist()))
        th = np.deg2rad(np.array(thetas_deg, dtype=np.float64)).astype(np.float32)
        ph = np.zeros(len(thetas_deg), dtype=np.float32)
        out = ref.get_view_direction(torch.from_numpy(th), torch.from_numpy(ph),
                                     np.deg2rad(overhead), np.deg2rad(front))
        cases.append(dict(kind="elev_sweep", overhead_deg=overhead, front_deg=front,
                          thetas=th.tolist(), phis=ph.tolist(), expect=out.tolist()))
        # (b) random poses from the rand_poses distribution (views_dataset.py:9-22)
        th = np.deg2rad(rng.uniform(0, 150, 256)).astype(np.float32)
        ph = np.deg2rad(rng.uniform(0, 360, 256)).astype(np.float32)
        out = ref.get_view_direction(torch.from_numpy(th), torch.from_numpy(ph),
                                     np.deg2rad(overhead), np.deg2rad(front))
        cases.append(dict(kind="random", overhead_deg=overhead, front_deg=front,
                          thetas=th.tolist(), phis=ph.tolist(), expect=out.tolist()))
    # (c) the default-argument form (top=30, front=0, angle=45 taken as degrees)
    th = np.deg2rad(rng.uniform(0, 180, 128)).astype(np.float32)
    ph = np.deg2rad(rng.uniform(0, 360, 128)).astype(np.float32)
    out = ref.get_view_direction(torch.from_numpy(th), torch.from_numpy(ph))
    cases.append(dict(kind="defaults", thetas=th.tolist(), phis=ph.tolist(), expect=out.tolist()))

    t2n = []
    for vals in ([-1.0, 0.0, 1.0], [0.0, 0.5, 1.0], [-0.5, 0.25, 0.75], [0.1, 0.2, 0.999]):
        out = ref.tensor2numpy(torch.tensor(vals))
        t2n.append(dict(input=vals, expect=[int(v) for v in out.tolist()]))

    with open(OUT, "w") as f:
        json.dump(dict(source="reference src/utils.py (get_view_direction :8-27, tensor2numpy :57-62)",
                       view_direction=cases, tensor2numpy=t2n), f)
    print("wrote", OUT, len(cases), "view cases")


if __name__ == "__main__":
    main()
