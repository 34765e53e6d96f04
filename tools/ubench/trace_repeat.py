#!/usr/bin/env python3
"""How long is ONE closest-hit launch over the refracted rays of a 9-view share when it is repeated back to back (warm clocks, warm
caches), with idle gaps in between, and right after a rebuild?  (Where does the 0.17 ms of `trace2` at 9 views come from?)
usage (via gpurun): python tools/ubench/trace_repeat.py [views]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drt_amd import diffrender as Render, mesh_io, views

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 9
res = 1024
mesh = mesh_io.subdivide_midpoint(mesh_io.read_ply("data/horse_vh.ply"))
Render.intIOR = 1.4723
Render.resx = Render.resy = res
scene = Render.Scene(mesh, 0)
c, ext = views.mesh_frame(mesh.vertices)
cams = views.turntable_cameras(c, ext, 72, res, res)
rays = []
with torch.no_grad():
    for k in range(0, 72, 72 // nv)[:nv] if False else list(range(0, 72, 8))[:nv]:
        o, d = views.generate_ray(res, res, cams[k][3], cams[k][2], device="cuda")
        first, _ = scene.Dintersect(Render.Ray(o, d))
        ok, inside = scene.refract_ray(first)
        r = inside.select(ok)
        rays.append(torch.cat([r.origin.float(), r.direction.float()], 1))
R = torch.cat(rays).contiguous()
print("refracted rays:", len(R))
tr = scene.optix_mesh
ev = [torch.cuda.Event(enable_timing=True) for _ in range(402)]

def timed(n, gap=0.0, rebuild=False):
    ts = []
    for i in range(n):
        if rebuild:
            tr.update_vert_f64(scene.vertices.detach())
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); tr.intersect(R); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
        if gap:
            time.sleep(gap)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]

for _ in range(3):
    tr.intersect(R)
torch.cuda.synchronize()
print("back to back, sync each   : median %.3f min %.3f max %.3f ms" % timed(50))
print("1 ms idle gaps            : median %.3f min %.3f max %.3f ms" % timed(50, 0.001))
print("20 ms idle gaps           : median %.3f min %.3f max %.3f ms" % timed(20, 0.02))
print("after a rebuild each time : median %.3f min %.3f max %.3f ms" % timed(30, 0.0, True))
# a stream of 100 launches without host syncs: per-launch = total / 100
torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100):
    tr.intersect(R)
b.record(); torch.cuda.synchronize()
print("100 launches in a row     : %.3f ms each" % (a.elapsed_time(b) / 100))
for n in (len(R) // 8, len(R) // 64, 4096, 256):
    sub = R[:n].contiguous()
    tr.intersect(sub); torch.cuda.synchronize()
    a.record()
    for _ in range(100):
        tr.intersect(sub)
    b.record(); torch.cuda.synchronize()
    print("  %8d rays: %.3f ms each (100 in a row)" % (n, a.elapsed_time(b) / 100))
os.system("rocm-smi --showclocks 2>/dev/null | grep -iE 'sclk|mclk|fclk' | head -8")
