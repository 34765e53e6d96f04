"""cProfile of the drop-in silhouette term (8 views, forward + backward) at the reference's iteration size: where the HOST time goes."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
init_vertices, parameter, opt = O.setup_opt(scene, 0.05, hp)
def it():
    scene.update_verticex(init_vertices + parameter)
    lc.vh_loss().backward()
for _ in range(5): it()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40): it()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"vh term: host enqueue {1e3*(t1-t0)/40:.3f} ms, with drain {1e3*(t2-t0)/40:.3f} ms per iteration (8 views)")
pr = cProfile.Profile(); pr.enable()
for _ in range(40): it()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
