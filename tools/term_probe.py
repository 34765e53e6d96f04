"""Per-term loss / gradient of the drop-in methods against the one-pass (fused) kernels at one state of the recorded trajectory."""
import sys; sys.path.insert(0, 'tests')
import numpy as np, torch
import trajectory_case as tc
from conftest import data_path
from drt_amd import mesh_io, diffrender as Render, optim as O
g = tc.load()
hand = mesh_io.read_ply(data_path("hand_vh.ply"))
Render.intIOR = float(g["ior"]); Render.resx = Render.resy = int(g["res"])
Vs = g["vertices"].astype(np.float64)
hp = dict(O.HyperParams, IOR=float(g["ior"]), momentum=float(g["momentum"]), ray_w=float(g["ray_w"]), sm_w=float(g["sm_w"]), vh_w=float(g["vh_w"]))
scene = Render.Scene(mesh_io.TriMesh(Vs, hand.faces), 0); scene.mean_len = float(g["mean_len"])
data = tc.RecordedCapture(g, hand.vertices, "cuda")
V0 = torch.tensor(Vs, device="cuda") + torch.tensor(g["params"][0], device="cuda")
res = {}
for fused in (False, True):
    lc = O.Loss_calculator(scene, data, hp, fused=fused)
    for name, fn in (("ray", lc.ray_loss), ("vh", lc.vh_loss), ("sm", lc.sm_loss)):
        V = V0.clone().requires_grad_(True)
        scene.update_verticex(V)
        l = fn()
        gr, = torch.autograd.grad(l, V)
        res[(name, fused)] = (float(l), gr)
for name in ("ray", "vh", "sm"):
    (la, ga), (lb, gb) = res[(name, False)], res[(name, True)]
    d = (ga - gb).abs()
    i = int(d.max(1).values.argmax())
    print(f"{name}: loss {la!r} vs {lb!r} rel {abs(la-lb)/abs(la):.1e}; grad max|a| {float(ga.abs().max()):.3e} max diff {float(d.max()):.3e} at vertex {i}: {ga[i].tolist()} vs {gb[i].tolist()}")
