#!/usr/bin/env python3
"""Per-iteration time of the reference's OWN loop shape (optim.py:198-215): one refraction view + 8 silhouette
views + smoothness per iteration, 960x1280 rays, through drt_amd.optim.Loss_calculator."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drt_amd import diffrender as Render, mesh_io, optim as O, views  # noqa: E402

resx, resy, n_views = 1280, 960, int(os.environ.get("VIEWS", "16"))
base = mesh_io.subdivide_midpoint(mesh_io.read_ply("data/horse_vh.ply"))
# smooth a little so that sm_loss is finite (the reference always remeshes first)
V = base.vertices.copy()
e = base.edges
for _ in range(5):
    acc = np.zeros_like(V); cnt = np.zeros(len(V))
    np.add.at(acc, e[:, 0], V[e[:, 1]]); np.add.at(cnt, e[:, 0], 1)
    V = 0.5 * V + 0.5 * acc / cnt[:, None]
mesh = mesh_io.TriMesh(V.astype(np.float32).astype(np.float64), base.faces)
Render.intIOR = 1.4723
Render.resx, Render.resy = resx, resy
center, extent = views.mesh_frame(mesh.vertices)
gt_scene = Render.Scene(views.displaced_ground_truth(mesh, 0.2, 1), 0)
data = O.SyntheticData(gt_scene, center, extent, resx, resy, num_view=n_views, n_total=n_views)
scene = Render.Scene(mesh, 0)
for fused in (False, True):
    hp = dict(O.HyperParams, Pass=1, Iters=1)
    lc = O.Loss_calculator(scene, data, hp, fused=fused)
    init_vertices, parameter, opt = O.setup_opt(scene, 0.05, hp, hook=not fused, fused=fused)     # (fused terms come with the one-kernel limit_hook + SGD)

    def iteration():
        opt.zero_grad()
        scene.update_verticex(init_vertices + parameter)
        loss, parts = lc.all_loss()
        loss.backward()
        opt.step()
        return parts

    for _ in range(5):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        parts = iteration()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"fused={fused}: {1e3 * dt:.3f} ms/iteration  ({resx * resy / dt / 1e6:.1f} M refraction rays/s)  {O.loss_string(parts)}")
    # breakdown
    for name, fn in (("ray", lc.ray_loss), ("vh", lc.vh_loss), ("sm", lc.sm_loss)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            scene.update_verticex(init_vertices + parameter)
            fn().backward()
        torch.cuda.synchronize()
        print(f"    {name}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms (incl. rebuild + backward)")
    if os.environ.get("CPROFILE") and fused:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        for _ in range(20):
            iteration()
        torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)

# the same iteration without the autograd graph around the one-pass terms (optim.FusedIteration)
hp = dict(O.HyperParams, Pass=1, Iters=1)
it = O.FusedIteration(scene, data, hp, 0.05)
for _ in range(5):
    it.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 100
for _ in range(n):
    total, parts = it.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
p = parts.tolist()
print(f"fused, no autograd (FusedIteration): {1e3 * dt:.3f} ms/iteration  ({resx * resy / dt / 1e6:.1f} M refraction rays/s)  ray={p[0]:g} vh={p[1]:g} sm={p[2]:g}")
t0 = time.perf_counter()
for _ in range(n):
    it.step()
host = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
print(f"    host enqueue alone: {1e3 * host:.3f} ms/iteration")
it2 = O.FusedIteration(scene, data, hp, 0.05, concurrent=False)
for _ in range(5):
    it2.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    it2.step()
torch.cuda.synchronize()
print(f"    terms one after the other on one stream: {1e3 * (time.perf_counter() - t0) / n:.3f} ms/iteration")

