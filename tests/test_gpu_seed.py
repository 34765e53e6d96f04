"""Temporal hit seeds (include/drt_hip.h drt_render_seed, drt_amd/csrc/drt_trace_kernel.h TraceSeed): the traversal of the refracted rays
(reference DiffRender.py:542) starts from the triangle the same pixel's refracted ray left the object through in the previous call.  The
contract is that NOTHING observable changes: outputs, masks and face ids are those of the unseeded call bit for bit -- with the seeds
the loop itself produces over moving vertices, with random and out-of-range seeds, with seeds of another mesh of another size -- and that
the seeds do save node visits when they are right."""
import numpy as np
import pytest
import torch

from conftest import IOR, data_path
from drt_amd import mesh_io, views

pytestmark = pytest.mark.gpu

RES, NV = 128, 3


@pytest.fixture()
def setup():
    from drt_amd import diffrender as Render
    old = (Render.HIT_SEED, Render.RECYCLE_OUTPUTS)
    Render.intIOR = IOR
    Render.resx = Render.resy = RES
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    c, ext = views.mesh_frame(mesh.vertices)
    cams = views.turntable_cameras(c, ext, 8, RES, RES)

    def rays():
        r = [views.generate_ray(RES, RES, cams[k][3], cams[k][2], device="cuda") for k in range(NV)]
        return torch.cat([x[0] for x in r]).contiguous(), torch.cat([x[1] for x in r]).contiguous()
    yield Render, mesh, rays
    Render.HIT_SEED, Render.RECYCLE_OUTPUTS = old


def _moving(mesh, steps, drift=0.004):
    v0 = torch.tensor(mesh.vertices, dtype=torch.float64, device="cuda")
    c = v0.mean(0)
    g = torch.Generator(device="cuda").manual_seed(3)
    out = []
    for k in range(steps):
        out.append(c + (v0 - c) * (1.0 + drift * k) + 0.05 * k + 0.02 * k * torch.randn(v0.shape, generator=g, device="cuda", dtype=torch.float64))
    return out


def _run(Render, mesh, rays, Vs, seeded, poke=None):
    """The sequence of calls; returns per call (out_ori, out_dir, mask, face1, face2) clones.  `poke(k, d)`: edit the seeds before call k."""
    Render.HIT_SEED = seeded
    o, d = rays()
    scene = Render.Scene(mesh, 0)
    out = []
    for k, V in enumerate(Vs):
        if poke is not None:
            poke(k, d)
        scene.update_verticex(V)
        oo, od, mk = scene.render_transparent(o, d)
        out.append((oo.clone(), od.clone(), mk.clone(), scene.last_face1.clone(), scene.last_face2.clone()))
    return out, d, scene


def _same(a, b):
    for k, (x, y) in enumerate(zip(a, b)):
        for j in range(5):
            assert torch.equal(x[j], y[j]), (k, j)


def test_seeded_calls_equal_unseeded_ones_over_moving_vertices(setup):
    Render, mesh, rays = setup
    Vs = _moving(mesh, 7)
    ref, d0, _ = _run(Render, mesh, rays, Vs, False)
    assert getattr(d0, "_drt_seed2", None) is None
    got, d1, _ = _run(Render, mesh, rays, Vs, True)
    _same(ref, got)
    # call 0 establishes the grid verdict (no seeds yet), call 1 creates them, from then on they are last call's exit faces
    seed = d1._drt_seed2
    f2 = got[-1][4]
    hit = got[-1][2][:, 0]
    assert int(hit.sum()) > 500
    # (whole images: the buffer is laid out in 4 x 4-pixel tiles, like the key buffer of the projection pass -- drt_raster.h raster_slot)
    i = torch.arange(len(seed), device="cuda")
    y, x = i // RES, i % RES
    slot = ((((y >> 2) * (RES >> 2) + (x >> 2)) << 4) | ((y & 3) << 2) | (x & 3))
    seed = seed[slot]
    assert torch.equal(seed[hit], f2[hit])
    assert int((seed >= 0).sum()) >= int(hit.sum())            # (refracted rays that hit but whose path died later are seeded too)


def test_any_seed_content_gives_the_same_result(setup):
    Render, mesh, rays = setup
    Vs = _moving(mesh, 6, drift=0.02)
    ref, _, _ = _run(Render, mesh, rays, Vs, False)
    n_f = len(mesh.faces)
    g = torch.Generator(device="cuda").manual_seed(11)

    def poke(k, d):
        seed = getattr(d, "_drt_seed2", None)
        if seed is None:
            return
        if k == 2:      # uniformly random faces: almost all wrong, many of them real hits in front of or behind the true one
            seed.copy_(torch.randint(0, n_f, seed.shape, generator=g, device="cuda", dtype=torch.int32))
        elif k == 3:    # out of range on both sides, INT_MIN / INT_MAX included
            seed.copy_(torch.randint(-4, 5, seed.shape, generator=g, device="cuda", dtype=torch.int32) * (n_f // 2 + 1) + n_f)
            seed[::7] = torch.iinfo(torch.int32).min
            seed[3::7] = torch.iinfo(torch.int32).max
        elif k == 4:    # every pixel seeded with its neighbour's face (a plausible near miss)
            seed.copy_(torch.roll(seed.clamp(0, n_f - 1), 1))
    got, _, _ = _run(Render, mesh, rays, Vs, True, poke)
    _same(ref, got)


def test_seeds_survive_a_topology_change_and_another_size(setup):
    """The ray tensor outlives the mesh: after update_mesh to a mesh with FEWER faces the old ids are partly out of range, partly other
    triangles; after a subdivision they name coarse faces that no longer exist as such."""
    Render, mesh, rays = setup
    small = mesh_io.read_ply(data_path("hand_vh.ply"))
    big = mesh_io.subdivide_midpoint(small)
    V_small = torch.tensor(small.vertices, dtype=torch.float64, device="cuda")
    V_big = torch.tensor(big.vertices, dtype=torch.float64, device="cuda")
    res = {}
    for seeded in (False, True):
        Render.HIT_SEED = seeded
        o, d = rays()
        scene = Render.Scene(big, 0)
        out = []
        for mesh_k, V in ((big, V_big), (big, V_big * 1.001), (small, V_small), (small, V_small * 1.002), (big, V_big), (big, V_big)):
            if len(mesh_k.faces) != scene.faces.shape[0]:
                scene.update_mesh(mesh_k)
            scene.update_verticex(V)
            oo, od, mk = scene.render_transparent(o, d)
            out.append((oo.clone(), od.clone(), mk.clone(), scene.last_face1.clone(), scene.last_face2.clone()))
        res[seeded] = out
    _same(res[False], res[True])


def test_right_seeds_save_node_visits(setup):
    """Same vertices twice: the second call's seeds are exactly right, and the traversal of the refracted rays visits fewer nodes."""
    Render, mesh, rays = setup
    V = torch.tensor(mesh.vertices, dtype=torch.float64, device="cuda")
    visits = {}
    for seeded in (False, True):
        Render.HIT_SEED = seeded
        o, d = rays()
        scene = Render.Scene(mesh, 0)
        for _ in range(3):
            scene.update_verticex(V)
            scene.render_transparent(o, d)
        scene.optix_mesh.profile_enable(2)
        scene.optix_mesh.profile_read()
        scene.update_verticex(V)
        scene.render_transparent(o, d)
        prof = scene.optix_mesh.profile_read()
        ws, ls, lf, mx = scene.optix_mesh.trace_stats()["trace2"]
        scene.optix_mesh.profile_enable(0)
        visits[seeded] = ls / max(1, prof["trace2"][2])
    # (measured: a few per cent -- a refracted ray starts ON the surface, and most of its visits are the boxes around its origin and along
    # the segment up to the exit point, which no bound removes; only the boxes behind the exit point go)
    assert visits[True] < visits[False], visits


def test_fused_loss_with_seeds_equals_without(setup):
    Render, mesh, rays = setup
    Vs = _moving(mesh, 5)
    rng = np.random.default_rng(5)
    n = NV * RES * RES
    c = torch.tensor(mesh.vertices.mean(0))
    sp = torch.tensor(rng.standard_normal((n, 3)) * 40.0, device="cuda") + (c + torch.tensor([0.0, 0.0, 150.0])).cuda()
    valid = torch.tensor(rng.random(n) > 0.1, device="cuda")
    res = {}
    for seeded in (False, True):
        Render.HIT_SEED = seeded
        o, d = rays()
        scene = Render.Scene(mesh, 0)
        out = []
        for V in Vs:
            Vr = V.clone().requires_grad_(True)
            scene.update_verticex(Vr)
            loss = scene.ray_loss_fused(o, d, sp, valid)
            loss.backward()
            out.append((loss.detach().clone(), Vr.grad.clone()))
        res[seeded] = out
    for (la, ga), (lb, gb) in zip(res[False], res[True]):
        # (same paths, same terms; only the order of the float64 atomics differs between two runs)
        assert abs(float(la) - float(lb)) <= 1e-9 * max(1.0, abs(float(la)))
        assert float((ga - gb).abs().max()) <= 1e-9 * max(1.0, float(ga.abs().max()))
