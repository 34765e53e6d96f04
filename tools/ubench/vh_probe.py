import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drt_amd import diffrender as Render, mesh_io, optim as O, views
resx, resy, n_views = 1280, 960, 16
base = mesh_io.subdivide_midpoint(mesh_io.read_ply("data/horse_vh.ply"))
V = base.vertices.copy(); e = base.edges
for _ in range(5):
    acc = np.zeros_like(V); cnt = np.zeros(len(V))
    np.add.at(acc, e[:, 0], V[e[:, 1]]); np.add.at(cnt, e[:, 0], 1)
    V = 0.5 * V + 0.5 * acc / cnt[:, None]
mesh = mesh_io.TriMesh(V.astype(np.float32).astype(np.float64), base.faces)
Render.intIOR = 1.4723; Render.resx, Render.resy = resx, resy
center, extent = views.mesh_frame(mesh.vertices)
gt_scene = Render.Scene(views.displaced_ground_truth(mesh, 0.2, 1), 0)
data = O.SyntheticData(gt_scene, center, extent, resx, resy, num_view=n_views, n_total=n_views)
scene = Render.Scene(mesh, 0)
hp = dict(O.HyperParams, Pass=1, Iters=1)
lc = O.Loss_calculator(scene, data, hp, fused=False)
init_vertices, parameter, opt = O.setup_opt(scene, 0.05, hp, hook=True, fused=False)
scene.update_verticex(init_vertices + parameter)
T = np.zeros(6); N = 0
for rep in range(30):
    for vid in range(8):
        _, _, soft_mask, origin, _, camera_M = data.get_view(vid)
        eye = origin[0]
        t0 = time.perf_counter()
        edges = scene.silhouette_edge(eye)
        t1 = time.perf_counter()
        pix, out = scene.primary_visibility(edges, camera_M, eye, detach_depth=True)
        t2 = time.perf_counter()
        image = soft_mask.view((resy, resx))
        term = (image[pix[:, 1], pix[:, 0]] - out).abs().sum()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        term.backward()
        t5 = time.perf_counter()
        torch.cuda.synchronize()
        t6 = time.perf_counter()
        if rep >= 5:
            T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5]; N += 1
print("per view, us: silhouette_edge %.1f  primary_visibility (incl. its sync) %.1f  loss expression %.1f  (drain %.1f)  backward enqueue %.1f (drain %.1f)" % tuple(1e6 * T / N))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for vid in range(8):
        lc._silhouette_term(vid).backward()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=50))
