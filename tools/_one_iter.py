import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from drt_amd import diffrender as Render, mesh_io, optim as O, views
resx, resy = 1280, 960
base = mesh_io.subdivide_midpoint(mesh_io.read_ply("data/horse_vh.ply"))
V = base.vertices.copy(); e = base.edges
for _ in range(5):
    acc = np.zeros_like(V); cnt = np.zeros(len(V))
    np.add.at(acc, e[:, 0], V[e[:, 1]]); np.add.at(cnt, e[:, 0], 1)
    V = 0.5 * V + 0.5 * acc / cnt[:, None]
mesh = mesh_io.TriMesh(V.astype(np.float32).astype(np.float64), base.faces)
Render.intIOR = 1.4723; Render.resx, Render.resy = resx, resy
center, extent = views.mesh_frame(mesh.vertices)
gt = Render.Scene(views.displaced_ground_truth(mesh, 0.2, 1), 0)
data = O.SyntheticData(gt, center, extent, resx, resy, num_view=16, n_total=16)
scene = Render.Scene(mesh, 0)
hp = dict(O.HyperParams, Pass=1, Iters=1)
lc = O.Loss_calculator(scene, data, hp, fused=True)
init_vertices, parameter, opt = O.setup_opt(scene, 0.05, hp, hook=False, fused=True)
def iteration():
    opt.zero_grad(); scene.update_verticex(init_vertices + parameter)
    loss, parts = lc.all_loss(); loss.backward(); opt.step()
for _ in range(5): iteration()
torch.cuda.synchronize()
iteration(); iteration()
torch.cuda.synchronize()
