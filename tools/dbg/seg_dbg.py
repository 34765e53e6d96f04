import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import IOR, data_path
from drt_amd import diffrender as Render, mesh_io, views
RES, NV = 128, 4
Render.intIOR = IOR; Render.resx = Render.resy = RES
mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
c, ext = views.mesh_frame(mesh.vertices)
cams = views.turntable_cameras(c, ext, 8, RES, RES)
rays = [views.generate_ray(RES, RES, cams[k][3], cams[k][2], device="cuda") for k in range(NV)]
o = torch.cat([r[0] for r in rays]).contiguous(); d = torch.cat([r[1] for r in rays]).contiguous()
scene = Render.Scene(mesh, 0)
V = torch.tensor(mesh.vertices, dtype=torch.float64, device="cuda", requires_grad=True)
for step in range(3):
    scene.update_verticex(V)
    oo, od, mk = scene.render_transparent(o, d)
    link = od._drt_link
    vi, nv = link.paths
    torch.cuda.synchronize()
    n = int(nv.item())
    lst = vi[:n].cpu().numpy()
    want = torch.nonzero(mk[:, 0]).squeeze(1).cpu().numpy()
    print(step, "n_valid", n, "mask rows", len(want), "same set", np.array_equal(np.sort(lst), want), "sorted by segment", bool((np.diff(lst // RES**2) >= 0).all()) if n else None)
