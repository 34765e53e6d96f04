"""Whole reconstruction, end to end -- the reference's ``python optim.py`` (optim.py:173-229).

    python -m drt_amd.reconstruct --name horse [--capture horse.npz] [--res 512] [--passes 20] [--iters 200]

``optimize(HyperParams)`` of the reference builds the scene from ``<data>/<name>_vh.ply``, loads the capture,
runs Pass x Iters iterations with a remesh before every pass and writes ``<result>/<name>_recons.ply``
(optim.py:173-229); its README then measures the result against ``<name>_scan.ply`` with MeshLab.  This
module does the same on the HIP path.  The captures (HDF5) are not distributed with the reference: when
``--capture`` is not given, a synthetic capture is traced through the scanned mesh (or, without a scan,
through a displaced copy of the hull) with the same tuple layout.
"""
from __future__ import annotations

import argparse
import json
import os
import time

import torch

from . import captured_data, diffrender as Render, mesh_io, metrics, optim, views


def run(HyperParams, data_path="./data/", result_path="./result/", capture=None, res=None, fused=True, output=True, device=0, n_views=72):
    name = HyperParams["name"]
    hull_path = os.path.join(data_path, f"{name}_vh.ply")
    scan_path = os.path.join(data_path, f"{name}_scan.ply")
    Render.intIOR = HyperParams["IOR"]
    scene = Render.Scene(hull_path, device)
    scan_scene = Render.Scene(scan_path, device) if os.path.exists(scan_path) else None
    if capture is not None:
        data = captured_data.get_data(HyperParams, path=capture)
    else:
        resx = resy = int(res or 512)
        Render.resx, Render.resy = resx, resy
        gt = scan_scene if scan_scene is not None else Render.Scene(views.displaced_ground_truth(scene.mesh, 0.5, 0), device)
        center, extent = views.mesh_frame(gt.mesh.vertices)
        data = captured_data.SyntheticData(gt, center, extent, resx, resy, num_view=min(HyperParams["num_view"], n_views), n_total=n_views, name=name)
    report = {"name": name, "resx": data.resx, "resy": data.resy, "views": data.n_total, "hull_faces": int(scene.faces.shape[0])}
    if scan_scene is not None:
        report["hull_to_scan"] = metrics.hausdorff(scene, scan_scene)
    t0 = time.time()
    scene, history = optim.optimize(scene, data, HyperParams, output=output, fused=fused)
    torch.cuda.synchronize()
    report["optimize_seconds"] = time.time() - t0
    report["iterations"] = HyperParams["Pass"] * HyperParams["Iters"]
    report["result_faces"] = int(scene.faces.shape[0])
    if scan_scene is not None:
        report["result_to_scan"] = metrics.hausdorff(scene, scan_scene)
    os.makedirs(result_path, exist_ok=True)
    out = os.path.join(result_path, f"{name}_recons.ply")
    scene.mesh.export(out)
    report["result"] = out
    return scene, report


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--name", default=optim.HyperParams["name"])
    ap.add_argument("--data-path", default="./data/")
    ap.add_argument("--result-path", default="./result/")
    ap.add_argument("--capture", default=None, help=".npz / .h5 capture with the reference's datasets (default: synthetic capture)")
    ap.add_argument("--res", type=int, default=512, help="resolution of the synthetic capture")
    ap.add_argument("--passes", type=int, default=optim.HyperParams["Pass"])
    ap.add_argument("--iters", type=int, default=optim.HyperParams["Iters"])
    ap.add_argument("--num-view", type=int, default=optim.HyperParams["num_view"], help="views the refraction loss cycles through")
    ap.add_argument("--views", type=int, default=72, help="views of the synthetic capture (the real captures have 72)")
    ap.add_argument("--ior", type=float, default=optim.HyperParams["IOR"])
    ap.add_argument("--dropin", action="store_true", help="use the reference-shaped (unfused) loss terms")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    import numpy as np
    np.random.seed(a.seed)
    hp = dict(optim.HyperParams, name=a.name, Pass=a.passes, Iters=a.iters, num_view=a.num_view, IOR=a.ior)
    _, report = run(hp, a.data_path, a.result_path, a.capture, a.res, fused=not a.dropin, n_views=a.views)
    print(json.dumps(report))


if __name__ == "__main__":
    main()
