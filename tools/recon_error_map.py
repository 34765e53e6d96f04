#!/usr/bin/env python3
"""Where the error of the end-to-end reconstruction sits ("max 5.63 -> 5.31 mm, concavities not recovered -- the reference's
behaviour or the remesher's?").  Runs the reference recipe (20 passes x 200 iterations, remesh before every pass, synthetic 72-view capture of
horse_scan.ply) and then asks, per vertex of the result: how far from the scan, and does ANY completed two-bounce path of ANY of the 72 views
put a gradient on it (|d ray_loss / d vertex| summed over a full sweep)?  Also the scan's own vertices against the result (the direction that
sees unrecovered concavities) and the same numbers for the input hull.
usage (via gpurun): python tools/recon_error_map.py [REMESH=gpu|host|none via env]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drt_amd import captured_data, diffrender as Render, metrics, optim, views
from drt_amd.remesh import Meshlabserver
from drt_amd.remesh_gpu import GpuMeshlabserver

res = int(os.environ.get("RES", "768"))
np.random.seed(0)
hp = dict(optim.HyperParams, name="horse", Pass=int(os.environ.get("PASSES", "20")))
Render.intIOR = hp["IOR"]; Render.resx = Render.resy = res
scene = Render.Scene("data/horse_vh.ply", 0)
scan = Render.Scene("data/horse_scan.ply", 0)
center, extent = views.mesh_frame(scan.mesh.vertices)
data = captured_data.SyntheticData(scan, center, extent, res, res, num_view=72, name="horse")
which = os.environ.get("REMESH", "gpu")
ml = {"host": Meshlabserver, "gpu": GpuMeshlabserver}.get(which, lambda: None)()


def coverage(sc):
    """|d ray_loss / d vertex| summed over all 72 views at the current vertices (one float64 [V] tensor), and the number of completed paths."""
    V = sc.vertices.detach().clone().requires_grad_(True)
    sc.update_verticex(V)
    acc = torch.zeros(V.shape[0], dtype=torch.float64, device=V.device)
    paths = 0
    for k in range(72):
        target, valid, _, origin, ray_dir, _ = data.get_view(k)
        oo, od, mk = sc.render_transparent(origin, ray_dir)
        loss = Render.ray_loss(oo, od, mk, target, valid)
        g, = torch.autograd.grad(loss, V)
        acc += g.norm(dim=1)
        paths += int((mk[:, 0] & valid).sum())
    return acc, paths


def report(tag, sc):
    d = metrics.vertex_to_surface(sc.vertices.detach(), scan)
    back = metrics.vertex_to_surface(scan.vertices.detach(), sc)
    cov, paths = coverage(sc)
    seen = cov > 0
    q = lambda x, p: float(torch.quantile(x, p)) if x.numel() else float("nan")
    print(f"{tag}: {sc.faces.shape[0]} faces; result -> scan mean {d.mean():.4f} p99 {q(d, 0.99):.3f} max {d.max():.3f} | scan -> result mean {back.mean():.4f} p99 {q(back, 0.99):.3f} max {back.max():.3f}")
    print(f"    vertices that get a ray-loss gradient from some view: {int(seen.sum())} of {seen.numel()} ({100.0 * seen.float().mean():.1f} %), {paths} completed + targeted paths over 72 views")
    for name, m in (("with a gradient", seen), ("without", ~seen)):
        if int(m.sum()):
            print(f"    {name:16s}: mean {d[m].mean():.4f} p99 {q(d[m], 0.99):.3f} max {d[m].max():.3f}")
    top = d >= torch.quantile(d, 0.99)
    v = sc.vertices.detach()
    lo, hi = v.min(0).values, v.max(0).values
    rel = ((v[top] - lo) / (hi - lo)).mean(0).tolist()
    print(f"    the worst 1 % ({int(top.sum())} vertices, >= {q(d, 0.99):.2f} mm): {100.0 * seen[top].float().mean():.0f} % of them get a gradient; mean position in the bounding box (x, y, z) = ({rel[0]:.2f}, {rel[1]:.2f}, {rel[2]:.2f})")


report("hull  ", scene)
t0 = time.time()
views_ray = views_sil = None
for i_pass in range(hp["Pass"]):
    remesh_len = optim.interp_R(hp["start_len"], hp["end_len"], i_pass, hp["Pass"])
    lr = optim.interp_R(hp["start_lr"], hp["lr_decay"] * hp["start_lr"], i_pass, hp["Pass"])
    if ml is not None:
        ml.remesh(scene, remesh_len)
    stepper = optim.FusedIteration(scene, data, hp, lr)
    if views_ray is not None:
        stepper.ray_view, stepper.silh_view = views_ray, views_sil
    views_ray, views_sil = stepper.ray_view, stepper.silh_view
    for it in range(hp["Iters"]):
        stepper.step()
torch.cuda.synchronize()
print(f"remesher: {which}; {hp['Pass']} passes x {hp['Iters']} iterations in {time.time() - t0:.1f} s")
report("result", scene)
