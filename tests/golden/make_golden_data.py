#!/usr/bin/env python3
"""Golden vectors for the data side of the path, produced by importing the
reference's captured_data.py IN THIS CONTAINER (its cv2 / h5py / imageio imports are satisfied by empty
stand-in modules: none of the functions exercised here touches them).

    view_schedule.npz   Data.ray_view_generator / silh_view_generator (captured_data.py:61-82) under
                        np.random.seed(s), consumed in the order of one optimisation iteration
                        (optim.py:67-80, 91-95: one refraction view, then eight silhouette views)
    generate_ray.npz    generate_ray (captured_data.py:23-40) for one camera at 17x13

Run:  python tests/golden/make_golden_data.py        (needs /root/reference; the outputs are committed)
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
from drt_amd import views  # noqa: E402


def main():
    for name in ("imageio", "cv2", "h5py"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    import captured_data as cd
    cd.device = "cpu"

    rec = {}
    cases = [("hand", 72, 7), ("hand", 9, 8), ("mouse", 72, 9), ("horse", 36, 10)]
    for k, (name, num_view, seed) in enumerate(cases):
        d = cd.Data.__new__(cd.Data)
        d.name, d.num_view = name, num_view
        np.random.seed(seed)
        ray, silh = d.ray_view_generator(), d.silh_view_generator()
        ray_ids, silh_ids = [], []
        for _ in range(120):
            ray_ids.append(int(next(ray)))
            silh_ids.append([int(next(silh)) for _ in range(8)])
        rec[f"ray_{k}"] = np.array(ray_ids)
        rec[f"silh_{k}"] = np.array(silh_ids)
    rec["names"] = np.array([c[0] for c in cases])
    rec["num_view"] = np.array([c[1] for c in cases])
    rec["seed"] = np.array([c[2] for c in cases])
    np.savez_compressed(os.path.join(OUT, "view_schedule.npz"), **rec)

    cams = views.turntable_cameras(np.array([1.0, -2.0, 3.0]), 200.0, 72, 17, 13)
    R, K, R_inv, K_inv = [np.asarray(m) for m in cams[5]]
    o, d = cd.generate_ray(13, 17, K_inv, R_inv)
    np.savez_compressed(os.path.join(OUT, "generate_ray.npz"), R=R, K=K, R_inv=R_inv, K_inv=K_inv, origin=o.numpy(), dir=d.numpy())
    print("wrote view_schedule.npz, generate_ray.npz")


if __name__ == "__main__":
    main()
