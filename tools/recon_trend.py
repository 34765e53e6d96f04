#!/usr/bin/env python3
"""Vertex-to-scan distance after every pass of the reference recipe on the synthetic horse capture."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drt_amd import captured_data, diffrender as Render, metrics, optim, views
from drt_amd.remesh import Meshlabserver

res = int(os.environ.get("RES", "768"))
np.random.seed(0)
hp = dict(optim.HyperParams, name="horse", Pass=int(os.environ.get("PASSES", "20")))
Render.intIOR = hp["IOR"]; Render.resx = Render.resy = res
scene = Render.Scene("data/horse_vh.ply", 0)
scan = Render.Scene("data/horse_scan.ply", 0)
center, extent = views.mesh_frame(scan.mesh.vertices)
data = captured_data.SyntheticData(scan, center, extent, res, res, num_view=72, name="horse")
print("hull      : faces %6d  mean %.4f rms %.4f max %.3f" % ((scene.faces.shape[0],) + tuple(metrics.hausdorff(scene, scan)[k] for k in ("mean", "rms", "max"))))
from drt_amd.remesh_gpu import GpuMeshlabserver
ml = Meshlabserver() if os.environ.get("REMESH", "gpu") == "host" else GpuMeshlabserver()
print("remesher:", type(ml).__name__)
t0 = time.time()
t_remesh = 0.0
views_ray = views_sil = None
for i_pass in range(hp["Pass"]):
    remesh_len = optim.interp_R(hp["start_len"], hp["end_len"], i_pass, hp["Pass"])
    lr = optim.interp_R(hp["start_lr"], hp["lr_decay"] * hp["start_lr"], i_pass, hp["Pass"])
    torch.cuda.synchronize(); tr = time.time()
    ml.remesh(scene, remesh_len)
    torch.cuda.synchronize(); t_remesh += time.time() - tr
    stepper = optim.FusedIteration(scene, data, hp, lr)
    if views_ray is not None:
        stepper.ray_view, stepper.silh_view = views_ray, views_sil
    views_ray, views_sil = stepper.ray_view, stepper.silh_view
    for it in range(hp["Iters"]):
        total, parts = stepper.step()
    h = metrics.hausdorff(scene, scan)
    print("pass %2d len %5.2f: faces %6d  mean %.4f rms %.4f max %.3f  %s  (%.1f s, remesh %.2f s so far)" % (i_pass, remesh_len, scene.faces.shape[0], h["mean"], h["rms"], h["max"], optim.loss_string(tuple(parts)), time.time() - t0, t_remesh))
