#!/usr/bin/env python3
"""would re-ordering the refracted-ray list R1 by (direction octant, Morton cell of the origin) make the closest-hit
traversal faster?  Proxy without touching the pipeline: the refracted rays of 36 views of the benchmark mesh (1.5 M rays, the size of
one k_trace launch of the timed step) in (a) the pipeline's order (views one after the other, screen order inside a view -- the pipeline
uses 16x4-pixel tiles, also measured), (b) sorted by octant + 30-bit Morton code of the origin, (c) sorted by Morton code only,
(d) shuffled; each through the same persistent kernel (optix_mesh.intersect), 30 launches in a row.
usage (via gpurun): python tools/ubench/reorder_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drt_amd import diffrender as Render, mesh_io, views

res, nv = 1024, 36
mesh = mesh_io.subdivide_midpoint(mesh_io.read_ply("data/horse_vh.ply"))
Render.intIOR = 1.4723
Render.resx = Render.resy = res
scene = Render.Scene(mesh, 0)
c, ext = views.mesh_frame(mesh.vertices)
cams = views.turntable_cameras(c, ext, 72, res, res)
rays, tile_keys = [], []
with torch.no_grad():
    for v in range(0, 72, 2)[:nv]:
        o, d = views.generate_ray(res, res, cams[v][3], cams[v][2], device="cuda")
        first, _ = scene.Dintersect(Render.Ray(o, d))
        ok, inside = scene.refract_ray(first)
        r = inside.select(ok)
        rays.append(torch.cat([r.origin.float(), r.direction.float()], 1))
        pix = r.ray_ind.to(torch.long)
        y, x = pix // res, pix % res
        tile_keys.append((v * (res // 4) * (res // 16) + (y // 4) * (res // 16) + (x // 16)) * 64 + (y % 4) * 16 + (x % 16))
R = torch.cat(rays).contiguous()
tile = torch.cat(tile_keys)
n = len(R)
lo, hi = R[:, :3].min(0).values, R[:, :3].max(0).values
q = ((R[:, :3] - lo) / (hi - lo) * 1023).clamp(0, 1023).to(torch.long)

def spread(v):
    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
    return v
morton = (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])
octant = ((R[:, 3] < 0).long() << 2) | ((R[:, 4] < 0).long() << 1) | (R[:, 5] < 0).long()
orders = {
    "screen order (row-major inside a view)": torch.arange(n, device="cuda"),
    "16x4-pixel tile order (the pipeline's)": torch.argsort(tile, stable=True),
    "octant, then Morton cell of the origin": torch.argsort((octant << 30) | morton, stable=True),
    "octant + Morton inside runs of 4096 tile-ordered rays": None,
    "Morton cell of the origin only": torch.argsort(morton, stable=True),
    "shuffled": torch.randperm(n, device="cuda"),
}
t_order = orders["16x4-pixel tile order (the pipeline's)"]
blk = torch.arange(n, device="cuda") // 4096
orders["octant + Morton inside runs of 4096 tile-ordered rays"] = t_order[torch.argsort((blk << 34) | ((octant[t_order] << 30) | morton[t_order]), stable=True)]
tr = scene.optix_mesh
ref = None
print(f"{n} refracted rays of {nv} views")
for name, perm in orders.items():
    S = R[perm].contiguous()
    T, ID = tr.intersect(S)
    inv = torch.empty_like(perm); inv[perm] = torch.arange(n, device="cuda")
    if ref is None:
        ref = ID[inv].clone()
    assert torch.equal(ID[inv], ref)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        tr.intersect(S)
    b.record(); torch.cuda.synchronize()
    print(f"  {name:58s} {a.elapsed_time(b) / 30:.3f} ms per launch")
