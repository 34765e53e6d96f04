#!/usr/bin/env python3
"""would LENGTH-binned ray lists (a wave's 64 rays finish together) make the closest-hit traversal of the refracted
rays faster?  Upper bound without touching the pipeline: the refracted rays of 36 views of the benchmark mesh (1.5 M rays = one k_trace
launch of the timed step) with each ray's TRUE node-visit count (a perfect predictor: a -DDRT_PROBE_VISITS build of the library reports
it in place of T), ordered
  (a) in the pipeline's 16x4-pixel tile order,
  (b) inside each of the 8 XCD parts of the list (k_trace cuts the list into 8 contiguous parts) by visit-count bin, longest first, tile
      order inside a bin -- 2 / 4 / 8 / 16 bins at the quantiles, and fully sorted,
  (c) the same with a NOISY predictor (the count of another ray of the same 4x4-pixel neighbourhood),
each through the same persistent kernel (optix_mesh.intersect), 30 launches in a row.
usage (via gpurun): python tools/ubench/length_bin_probe.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
DUMP = "/tmp/length_bin_probe.pt"

if len(sys.argv) > 1 and sys.argv[1] == "visits":
    import torch
    from drt_amd import diffrender as Render, mesh_io, views
    res, nv = 1024, 36
    mesh = mesh_io.subdivide_midpoint(mesh_io.read_ply(os.path.join(ROOT, "data/horse_vh.ply")))
    Render.intIOR = 1.4723
    Render.resx = Render.resy = res
    scene = Render.Scene(mesh, 0)
    c, ext = views.mesh_frame(mesh.vertices)
    dist = float(os.environ.get("PROBE_DISTANCE", "2.5"))
    cams = views.turntable_cameras(c, ext, 72, res, res, distance_factor=dist)
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
    T, ID = scene.optix_mesh.intersect(R)
    torch.cuda.synchronize()
    torch.save({"R": R.cpu(), "tile": tile.cpu(), "visits": T.cpu(), "ID": ID.cpu()}, DUMP)
    sys.exit(0)

from drt_amd import build as B
vis_lib = os.path.join(ROOT, "drt_amd", "_ab", "visits.so")
os.makedirs(os.path.dirname(vis_lib), exist_ok=True)
if not os.path.exists(vis_lib):
    B.build(force=True, out=vis_lib, extra_flags=("-DDRT_PROBE_VISITS=1",))
subprocess.check_call([sys.executable, os.path.abspath(__file__), "visits"], env=dict(os.environ, DRT_HIP_LIB=vis_lib))

import torch
from drt_amd import diffrender as Render, mesh_io
d = torch.load(DUMP)
R, tile, visits, ID0 = d["R"].cuda(), d["tile"].cuda(), d["visits"].cuda(), d["ID"].cuda()
n = len(R)
hit = ID0 >= 0
vis = torch.where(hit, visits, torch.zeros_like(visits)).to(torch.long)      # (a miss reports T = -1: short rays anyway)
q = torch.quantile(vis[hit].float(), torch.tensor([0.1, 0.5, 0.9, 0.99], device="cuda"))
print(f"{n} refracted rays; node visits per ray: mean {vis[hit].float().mean():.1f}, p10 {q[0]:.0f} median {q[1]:.0f} p90 {q[2]:.0f} p99 {q[3]:.0f} max {vis.max()}")
mesh = mesh_io.subdivide_midpoint(mesh_io.read_ply(os.path.join(ROOT, "data/horse_vh.ply")))
Render.intIOR = 1.4723
scene = Render.Scene(mesh, 0)
t_order = torch.argsort(tile, stable=True)
pos = torch.empty(n, dtype=torch.long, device="cuda"); pos[t_order] = torch.arange(n, device="cuda")
part = pos * 8 // n                                   # which XCD part of the tile-ordered list a ray sits in


def binned(pred, n_bins):
    """inside each XCD part: bin by predicted visits (longest first), tile order inside a bin"""
    if n_bins <= 0:
        b = -pred
    else:
        edges = torch.quantile(pred.float(), torch.linspace(0, 1, n_bins + 1, device="cuda")[1:-1])
        b = -(torch.bucketize(pred.float(), edges))
    key = part * (1 << 40) + (b - b.min()) * (1 << 24)
    return t_order[torch.argsort(key[t_order], stable=True)]


# a noisy predictor: the visit count of the next ray in tile order (a neighbour in the 16x4 tile: what last step's count of the SAME pixel
# would roughly be worth after the vertices moved)
nb = vis[t_order].roll(1)
noisy = torch.empty_like(vis); noisy[t_order] = nb
corr = torch.corrcoef(torch.stack([vis.float(), noisy.float()]))[0, 1].item()
orders = {"16x4-pixel tile order (the pipeline's)": t_order}
for nbins in (2, 4, 8, 16, 0):
    orders[f"per XCD part: {nbins if nbins else 'fully sorted'} bins by TRUE visits, longest first"] = binned(vis, nbins)
orders["per XCD part: 8 bins, shortest first"] = t_order[torch.argsort((part * (1 << 40) + torch.bucketize(vis.float(), torch.quantile(vis.float(), torch.linspace(0, 1, 9, device='cuda')[1:-1])) * (1 << 24))[t_order], stable=True)]
for nbins in (4, 8):
    orders[f"per XCD part: {nbins} bins by a NEIGHBOUR's visits (corr {corr:.2f})"] = binned(noisy, nbins)
orders["shuffled"] = torch.randperm(n, device="cuda")
tr = scene.optix_mesh
ref = None
for name, perm in orders.items():
    S = R[perm].contiguous()
    T, ID = tr.intersect(S)
    inv = torch.empty_like(perm); inv[perm] = torch.arange(n, device="cuda")
    if ref is None:
        ref = ID[inv].clone()
        assert torch.equal(ref, ID0)
    assert torch.equal(ID[inv], ref)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            tr.intersect(S)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 30)
    print(f"  {name:75s} {best:.3f} ms per launch (prefilter + k_trace)", flush=True)
