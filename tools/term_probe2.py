"""Lockstep: along the drop-in trajectory, the three terms' gradients of the fused kernels against the drop-in ones at EVERY iteration."""
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
lcs = {f: O.Loss_calculator(scene, data, hp, fused=f) for f in (False, True)}
init, par, opt = O.setup_opt(scene, float(g["lr"]), hp)
for it in range(25):
    out = {}
    for fused in (True, False):
        lc = lcs[fused]
        for name, fn in (("ray", lc.ray_loss), ("vh", lc.vh_loss), ("sm", lc.sm_loss)):
            V = (init + par.detach()).requires_grad_(True)
            scene.update_verticex(V)
            l = fn()
            gr, = torch.autograd.grad(l, V)
            out[(name, fused)] = (float(l), gr)
    msg = []
    for name in ("ray", "vh", "sm"):
        (la, ga), (lb, gb) = out[(name, False)], out[(name, True)]
        msg.append(f"{name} dl {abs(la-lb)/max(1e-300,abs(la)):.1e} dg {float((ga-gb).abs().max()):.1e}")
    print(it, " | ".join(msg))
    w = O.loss_weights(hp, data.resy, scene.mean_len)
    opt.zero_grad()
    par.grad = None
    tot = sum(w[k] * out[(n, False)][1] for k, n in enumerate(("ray", "vh", "sm")))
    par.grad = O.limit_hook(tot.clone())
    opt.step()
