"""drt_tree_mode: updates that keep the tree's topology (mode 1: the LBVH's, for a number of updates; mode 2: a binned-SAH topology built
on the host at update_mesh) must give the results of the full per-update LBVH build -- T and ID bit for bit against the exhaustive test and
the oracle, the rendered outputs exact -- while the vertices move, after the topology has been kept through many updates, and across a
topology change; every tree passes drt_bvh_check."""
import numpy as np
import pytest
import torch

from conftest import IOR, data_path
from drt_amd import mesh_io, views
from oracle import diffrender_oracle as orc

pytestmark = pytest.mark.gpu


def _rays(mesh, res=96, view=7):
    c, ext = views.mesh_frame(mesh.vertices)
    cam = views.turntable_cameras(c, ext, 72, res, res)[view]
    o, d = views.generate_ray(res, res, cam[3], cam[2])
    rng = np.random.default_rng(3)
    inside = np.concatenate([rng.uniform(-0.4, 0.4, (3000, 3)) * ext + c, rng.standard_normal((3000, 3))], 1)
    return np.concatenate([torch.cat([o, d], 1).numpy(), inside]).astype(np.float32)


@pytest.mark.parametrize("mode,every", [(1, 5), (2, 1)])
def test_kept_topologies_trace_like_the_rebuilt_lbvh(mode, every):
    from drt_amd.optix_mesh import optix_mesh
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    F = torch.tensor(mesh.faces, dtype=torch.int32, device="cuda")
    V0 = mesh.vertices.astype(np.float32)
    rays = _rays(mesh)
    R = torch.tensor(rays, device="cuda")
    t = optix_mesh(0)
    t.tree_mode(mode, every)
    t.update_mesh(F, torch.tensor(V0, device="cuda"))
    rng = np.random.default_rng(1)
    heights = set()
    for step in range(12):
        # a drift that adds up (the kept topology ages), plus a stretch that changes the shape of the scene box
        V = (V0 * np.float32(1.0 + 0.03 * step) + rng.standard_normal(V0.shape).astype(np.float32) * np.float32(0.05 * step)).astype(np.float32)
        V[:, 1] *= np.float32(1.0 + 0.05 * step)
        if step % 2:
            t.update_vert(torch.tensor(V, device="cuda"))
        else:
            t.update_vert_f64(torch.tensor(V, device="cuda").double())
        bad, height = t.check()
        assert bad == 0
        heights.add(height)
        T, ID = t.intersect(R)
        Tb, IDb = t.intersect_bruteforce(R)
        assert torch.equal(ID, IDb) and torch.equal(T, Tb)
        if step in (0, 7):
            To, IDo = orc.trace_closest(mesh.faces.astype(np.int32), V, rays)
            assert np.array_equal(ID.cpu().numpy(), IDo) and np.array_equal(T.cpu().numpy(), To)
            assert (IDo >= 0).sum() > 1500
        assert torch.equal(t.intersect_any(R), ID >= 0)
    if mode == 2:
        assert len(heights) == 1          # the host's topology throughout
    # a topology change: the subdivided hull through the same tracer object
    fine = mesh_io.subdivide_midpoint(mesh)
    t.update_mesh(torch.tensor(fine.faces, dtype=torch.int32, device="cuda"), torch.tensor(fine.vertices.astype(np.float32), device="cuda"))
    for step in range(3):
        t.update_vert(torch.tensor((fine.vertices * (1.0 + 0.01 * step)).astype(np.float32), device="cuda"))
        assert t.check()[0] == 0
        T, ID = t.intersect(R)
        Tb, IDb = t.intersect_bruteforce(R)
        assert torch.equal(ID, IDb) and torch.equal(T, Tb)


@pytest.mark.parametrize("mode,every", [(1, 4), (2, 1)])
def test_rendered_outputs_do_not_depend_on_the_tree_mode(mode, every):
    from drt_amd import diffrender as Render
    Render.intIOR = IOR
    Render.resx = Render.resy = 128
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    c, ext = views.mesh_frame(mesh.vertices)
    cams = views.turntable_cameras(c, ext, 8, 128, 128)
    o = torch.cat([views.generate_ray(128, 128, cams[k][3], cams[k][2], device="cuda")[0] for k in range(2)]).contiguous()
    d = torch.cat([views.generate_ray(128, 128, cams[k][3], cams[k][2], device="cuda")[1] for k in range(2)]).contiguous()
    v0 = torch.tensor(mesh.vertices, dtype=torch.float64, device="cuda")
    out = {}
    for m in (0, mode):
        scene = Render.Scene(mesh, 0)
        scene.optix_mesh.tree_mode(m, every)
        if m == 2:
            scene.update_mesh(mesh)                     # (mode 2 takes effect at an update_mesh)
        res = []
        for step in range(6):
            scene.update_verticex(v0 * (1.0 + 0.01 * step) + 0.2 * step)
            oo, od, mk = scene.render_transparent(o, d)
            res.append((oo.clone(), od.clone(), mk.clone(), scene.last_face1.clone(), scene.last_face2.clone()))
            del oo, od, mk
        out[m] = res
    for a, b in zip(out[0], out[mode]):
        assert torch.equal(a[2], b[2]) and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert torch.equal(a[3], b[3]) and torch.equal(a[4][a[2][:, 0]], b[4][b[2][:, 0]])
        assert int(a[2][:, 0].sum()) > 500
