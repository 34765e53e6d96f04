"""Differential fuzzing of boundary B1 (optix_mesh.intersect / intersect_any, reference optix_extend.cpp:29-57): random and
adversarial meshes and rays through the BVH path, against the exhaustive test on the GPU and against oracle/tracer.c -- T and ID
bit for bit.  What is aimed at: the tie rule (equal t -> lowest face id) on shared edges, shared vertices and coplanar duplicates;
degenerate faces; direction components that are exactly zero; origins on the surface; rays inside a triangle's plane; coordinates
from 1e-2 to 1e3; meshes from one triangle to a few thousand (every leaf / collapse shape of a small tree)."""
import numpy as np
import pytest
import torch

from oracle import diffrender_oracle as orc



def _soup(rng, n_tri, scale):
    nv = max(3, int(n_tri * rng.uniform(0.6, 2.0)))
    V = (rng.uniform(-1, 1, (nv, 3)) * scale).astype(np.float32)
    F = rng.integers(0, nv, (n_tri, 3)).astype(np.int32)            # repeated indices happen: degenerate faces
    return V, F


def _grid(rng, n, scale):
    """A wavy height field of 2 n^2 triangles on exactly representable x / y coordinates: rays through grid points and along grid
    lines hit two to six triangles at the same t."""
    xs = (np.arange(n + 1, dtype=np.float32) - n / 2) * np.float32(scale / n)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    Z = (np.float32(0.1 * scale) * np.sin(3.0 * X / scale) * np.cos(2.0 * Y / scale)).astype(np.float32) if rng.random() < 0.7 else np.zeros_like(X)
    V = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)
    idx = lambda i, j: i * (n + 1) + j
    F = []
    for i in range(n):
        for j in range(n):
            F += [[idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)], [idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)]]
    F = np.array(F, np.int32)
    return V, F[rng.permutation(len(F))]                              # face ids in random order: the tie rule is about ids


def _stack(rng, n_tri, scale):
    """Coplanar duplicates and near-duplicates of a few triangles: same t for several faces."""
    base_v, base_f = _soup(rng, max(1, n_tri // 4), scale)
    F = np.concatenate([base_f] * 4)[:max(1, n_tri)]
    return base_v, F[rng.permutation(len(F))].astype(np.int32)


def _rays(rng, V, F, n, scale):
    tri = V[F]                                                       # [F, 3, 3]
    out = []
    # random rays from around the scene
    o = rng.uniform(-2, 2, (n, 3)) * scale
    d = rng.normal(size=(n, 3))
    out.append(np.concatenate([o, d], 1))
    # axis-aligned directions (two components exactly zero) and planar ones (one zero)
    d2 = np.zeros((n, 3)); d2[np.arange(n), rng.integers(0, 3, n)] = rng.choice([-1.0, 1.0], n)
    out.append(np.concatenate([rng.uniform(-1, 1, (n, 3)) * scale, d2], 1))
    d3 = rng.normal(size=(n, 3)); d3[np.arange(n), rng.integers(0, 3, n)] = 0.0
    out.append(np.concatenate([rng.uniform(-1.5, 1.5, (n, 3)) * scale, d3], 1))
    # aimed at vertices, edge midpoints and interior points of random faces (float32 arithmetic: through or next to the feature)
    f = rng.integers(0, len(F), n)
    w = rng.dirichlet([1, 1, 1], n)
    kind = rng.integers(0, 3, n)
    w[kind == 0] = np.eye(3)[rng.integers(0, 3, (kind == 0).sum())]
    e = rng.integers(0, 3, (kind == 1).sum()); we = np.full(((kind == 1).sum(), 3), 0.5); we[np.arange(len(e)), e] = 0.0; w[kind == 1] = we
    target = (tri[f].astype(np.float64) * w[:, :, None]).sum(1)
    o = rng.uniform(-2, 2, (n, 3)) * scale
    out.append(np.concatenate([o, target - o], 1))
    # axis-aligned rays through vertices (grid meshes: exactly through grid points and along grid lines)
    v = V[rng.integers(0, len(V), n)].astype(np.float64)
    ax = rng.integers(0, 3, n)
    o = v.copy(); o[np.arange(n), ax] += rng.choice([-1.0, 1.0], n) * 1.5 * scale
    out.append(np.concatenate([o, v - o], 1))
    # origins ON the surface (t = 0 is not a hit, the far side is) and rays inside a triangle's plane
    o = target
    out.append(np.concatenate([o, rng.normal(size=(n, 3))], 1))
    a = tri[f, 1].astype(np.float64) - tri[f, 0]; b = tri[f, 2].astype(np.float64) - tri[f, 0]
    inplane = a * rng.normal(size=(n, 1)) + b * rng.normal(size=(n, 1))
    out.append(np.concatenate([target - 3.0 * inplane, inplane], 1))
    r = np.concatenate(out).astype(np.float32)
    r[~np.isfinite(r)] = 0.0
    bad = np.abs(r[:, 3:]).sum(1) == 0                               # a zero direction is no ray: give it one
    r[bad, 3] = 1.0
    return np.ascontiguousarray(r)


def scene(seed, n_rays=600):
    """(V float32 [nv,3], F int32 [nf,3], rays float32 [7 n_rays, 6]) of one fuzz case."""
    rng = np.random.default_rng(1000 + seed)
    scale = float(10.0 ** rng.uniform(-2, 3))
    kind = seed % 4
    if kind == 0:
        V, F = _soup(rng, int(rng.choice([1, 2, 3, 5, 9, 17, 64, 257, 1000, 4000])), scale)
    elif kind == 1:
        V, F = _grid(rng, int(rng.choice([1, 2, 5, 16, 40])), scale)
    elif kind == 2:
        V, F = _stack(rng, int(rng.choice([4, 12, 100, 800])), scale)
    else:                                                             # a soup with slivers and far outliers
        V, F = _soup(rng, int(rng.choice([30, 300, 2000])), scale)
        V[rng.integers(0, len(V), max(1, len(V) // 20))] *= 50.0
        V[:, rng.integers(0, 3)] *= 1e-3
    return V, F, _rays(rng, V, F, n_rays, scale), scale


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_random_meshes_and_adversarial_rays_bit_exact(seed):
    from drt_amd.optix_mesh import optix_mesh
    V, F, rays, scale = scene(seed)
    t = optix_mesh(0)
    t.update_mesh(torch.tensor(F, device="cuda"), torch.tensor(V, device="cuda"))
    assert t.check()[0] == 0
    R = torch.tensor(rays, device="cuda")
    T, ID = t.intersect(R)
    Tb, IDb = t.intersect_bruteforce(R)
    assert torch.equal(ID, IDb) and torch.equal(T, Tb)
    To, IDo = orc.trace_closest(F, V, rays, bvh=False)                # the CPU restatement of the same contract (exhaustive)
    assert np.array_equal(ID.cpu().numpy(), IDo) and np.array_equal(T.cpu().numpy(), To)
    assert (IDo >= 0).sum() > 0 or len(F) < 3
    hit = t.intersect_any(R)
    assert torch.equal(hit.bool(), ID >= 0)
    # the same after a refit to moved vertices (update_vert keeps the topology, rebuilds the tree)
    V2 = (V * np.float32(1.25) + np.float32(0.1 * scale)).astype(np.float32)
    t.update_vert(torch.tensor(V2, device="cuda"))
    T2, ID2 = t.intersect(R)
    To2, IDo2 = orc.trace_closest(F, V2, rays, bvh=False)
    assert np.array_equal(ID2.cpu().numpy(), IDo2) and np.array_equal(T2.cpu().numpy(), To2)


@pytest.mark.gpu
@pytest.mark.parametrize("distance_factor", [20.0, 100.0, 1000.0])
def test_far_camera_keeps_its_hits(distance_factor):
    """A camera tens to a thousand extents away (focal length grown with the distance): tree == exhaustive test on the GPU ==
    oracle, and none of the hits of the four inequalities alone is lost to the hit-point condition, whose tolerance grows with the
    ray (drt_tri.h::hit_point_in_box); then the same camera through the whole path (projection pass included)."""
    from conftest import IOR, data_path
    from test_oracle_golden import far_camera_rays
    from drt_amd import diffrender as Render, mesh_io, views
    from drt_amd.optix_mesh import optix_mesh
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    F, V = mesh.faces.astype(np.int32), mesh.vertices.astype(np.float32)
    rays = far_camera_rays(mesh, distance_factor, res=128)
    t = optix_mesh(0)
    t.update_mesh(torch.tensor(F, device="cuda"), torch.tensor(V, device="cuda"))
    R = torch.tensor(rays, device="cuda")
    T, ID = t.intersect(R)
    Tb, IDb = t.intersect_bruteforce(R)
    To, IDo = orc.trace_closest(F, V, rays)
    T0, ID0 = orc.trace_closest(F, V, rays, mt_only=True)
    assert (IDo >= 0).sum() > 1000
    assert np.array_equal(IDo, ID0) and np.array_equal(To, T0)
    assert np.array_equal(IDb.cpu().numpy(), IDo) and np.array_equal(Tb.cpu().numpy(), To)
    assert np.array_equal(ID.cpu().numpy(), IDo) and np.array_equal(T.cpu().numpy(), To)
    # the whole path from the same camera: mask and exit rays against the oracle
    Render.intIOR = IOR
    sc = Render.Scene(mesh, 0)
    o, d = torch.tensor(rays[:, :3], dtype=torch.float64), torch.tensor(rays[:, 3:], dtype=torch.float64)
    d = d / d.norm(dim=1, keepdim=True)
    oo, od, mk = sc.render_transparent(o.cuda(), d.cuda())
    om = orc.Mesh(mesh.faces, torch.tensor(mesh.vertices, dtype=torch.float64))
    ro, rd, rm = orc.render_transparent(om, o, d, IOR)
    assert torch.equal(mk.cpu(), rm) and rm[:, 0].sum() > 300
    assert torch.allclose(od.cpu(), rd, atol=1e-9, rtol=0) and torch.allclose(oo.cpu(), ro, atol=1e-9 * distance_factor, rtol=0)


def _shape(rng, noises=(0.0, 0.05, 0.2)):
    """A random closed surface: a noisy, anisotropically scaled icosphere (star-shaped, with creases and near-tangent facets), sometimes
    two of them merged into one mesh (an inner and an outer shell, or two lobes side by side: rays leave one and enter the other)."""
    from drt_amd import mesh_io
    def one(sub, radius, noise, seed):
        m = mesh_io.icosphere(sub, radius=radius, noise=noise, seed=seed)
        return m.vertices * rng.uniform(0.5, 1.5, 3), m.faces
    V, F = one(int(rng.integers(1, 4)), float(rng.uniform(20, 60)), float(rng.choice(noises)), int(rng.integers(1 << 30)))
    if rng.random() < 0.4:
        V2, F2 = one(int(rng.integers(1, 3)), float(rng.uniform(5, 15)), min(0.1, max(noises)), int(rng.integers(1 << 30)))
        V2 = V2 + (rng.uniform(-1, 1, 3) * rng.choice([3.0, 80.0]))          # inside the first one, or beside it
        V, F = np.concatenate([V, V2]), np.concatenate([F, F2 + len(V)])
    V = V + rng.uniform(-30, 30, 3)
    return mesh_io.TriMesh(V.astype(np.float32).astype(np.float64), F)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_cameras_and_ior_through_the_whole_path(seed):
    """render_transparent + ray_loss + backward on random closed shapes, cameras (far, close, inside the object, non-square images) and
    indices of refraction, against the oracle: mask and both face ids exact, rays to 1e-10, loss to 1e-10, gradient to 1e-9 relative;
    the one-pass loss and a call without the image-size hint (tree for every primary ray) give the same."""
    from conftest import IOR
    from drt_amd import diffrender as Render, views
    rng = np.random.default_rng(7000 + seed)
    mesh = _shape(rng)
    ior = float(rng.choice([IOR, 1.1, 1.33, 2.4]))
    Render.intIOR = ior
    try:
        scene = Render.Scene(mesh, 0)
        w, h = [(64, 64), (96, 48), (48, 80), (128, 64)][seed % 4]
        c, ext = views.mesh_frame(mesh.vertices)
        dist = float(rng.choice([2.5, 1.2, 0.8, 0.3], p=[0.3, 0.4, 0.2, 0.1]))  # 0.8: grazing close-up; 0.3: the camera is inside (or on the edge of) the object
        R, K, Rinv, Kinv = views.turntable_cameras(c, ext, 72, w, h, distance_factor=dist)[int(rng.integers(72))]
        o, d = views.generate_ray(h, w, Kinv, Rinv)
        sp = torch.tensor(rng.standard_normal(o.shape) * 40.0 + np.asarray(c))
        valid = torch.tensor(rng.random(len(o)) > 0.2)
        wgt = torch.tensor(rng.standard_normal(o.shape))
        Vc = torch.tensor(mesh.vertices, dtype=torch.float64, requires_grad=True)
        oo, od, mk, aux = orc.render_transparent(orc.Mesh(mesh.faces, Vc), o, d, ior, return_aux=True)
        ref = orc.ray_loss(oo, od, mk, sp, valid)
        (ref + (od * wgt).sum() + (oo * wgt).sum()).backward()
        for hint in ((w, h), (7, 7)):                                         # projection pass where the rays verify / the tree for every ray
            Render.resx, Render.resy = hint
            V = torch.tensor(mesh.vertices, dtype=torch.float64, device="cuda", requires_grad=True)
            scene.update_verticex(V)
            out_ori, out_dir, mask = scene.render_transparent(o.cuda(), d.cuda())
            assert torch.equal(mask.cpu(), mk)
            assert torch.equal(scene.last_face1.cpu().long(), aux["face1"])
            f2 = aux["face2"].clone(); f2[~mk[:, 0]] = -1
            assert torch.equal(scene.last_face2.cpu().long(), f2)
            torch.testing.assert_close(out_dir.detach().cpu(), od.detach(), rtol=1e-10, atol=1e-11)
            torch.testing.assert_close(out_ori.detach().cpu(), oo.detach(), rtol=1e-10, atol=1e-9)
            loss = Render.ray_loss(out_ori, out_dir, mask, sp.cuda(), valid.cuda())
            assert loss.item() == pytest.approx(ref.item(), rel=1e-10, abs=1e-12)
            (loss + (out_dir * wgt.cuda()).sum() + (out_ori * wgt.cuda()).sum()).backward()
            err = (V.grad.cpu() - Vc.grad).abs().max().item()
            assert err <= 1e-5 and err <= 1e-9 * max(1.0, Vc.grad.abs().max().item()), err
            V2 = V.detach().clone().requires_grad_(True)
            scene.update_verticex(V2)
            lf = scene.ray_loss_fused(o.cuda(), d.cuda(), sp.cuda(), valid.cuda())
            assert lf.item() == pytest.approx(ref.item(), rel=1e-10, abs=1e-12)
    finally:
        Render.intIOR = IOR


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_random_shapes_through_the_silhouette_and_smoothness_terms(seed):
    """silhouette_edge + primary_visibility + |soft mask - 0.5| (reference optim.py:67-80) and -log(1 + cos dihedral) (optim.py:82-89) on
    random closed shapes and cameras against the oracle: the silhouette edge set and the sample pixels exact, the losses to 1e-12, the
    gradients to 1e-8 relative -- through the drop-in methods and through the one-kernel forms."""
    from drt_amd import diffrender as Render, views
    rng = np.random.default_rng(9000 + seed)
    mesh = _shape(rng)
    scene = Render.Scene(mesh, 0)
    res_x, res_y = [(64, 64), (96, 48), (48, 80), (128, 64)][seed % 4]
    Render.resx, Render.resy = res_x, res_y
    c, ext = views.mesh_frame(mesh.vertices)
    dist = float(rng.choice([2.5, 1.2, 0.8]))
    cam_np = views.turntable_cameras(c, ext, 72, res_x, res_y, distance_factor=dist)[int(rng.integers(72))]
    cam = tuple(torch.tensor(np.asarray(m), dtype=torch.float64) for m in cam_np)
    origin3 = cam[2][:3, 3].contiguous()
    soft = torch.tensor(rng.random(res_x * res_y))
    Edges, E2F = scene.Edges.cpu(), scene.E2F.cpu()
    Vc = torch.tensor(mesh.vertices, dtype=torch.float64, requires_grad=True)
    om = orc.Mesh(mesh.faces, Vc)
    sil_ref = orc.silhouette_edges(Vc, Edges, E2F, origin3)
    idx_ref, out_ref = orc.primary_visibility(om, sil_ref, cam, origin3, res_x, res_y, detach_depth=True)
    vh_ref = orc.vh_loss_view(om, Edges, E2F, cam, origin3, soft, res_x, res_y)
    sm_ref = orc.sm_loss(Vc, E2F)
    g_vh, = torch.autograd.grad(vh_ref, Vc, retain_graph=True)
    g_sm, = torch.autograd.grad(sm_ref, Vc)

    V = torch.tensor(mesh.vertices, dtype=torch.float64, device="cuda", requires_grad=True)
    scene.update_verticex(V)
    camd = tuple(m.cuda() for m in cam)
    sil = scene.silhouette_edge(origin3.cuda())
    assert torch.equal(sil.cpu(), sil_ref)
    index, output = scene.primary_visibility(sil, camd, origin3.cuda(), detach_depth=True)
    assert torch.equal(index.cpu(), idx_ref) and torch.equal(output.detach().cpu().double(), out_ref.double())
    vh = (soft.cuda().view((res_y, res_x))[index[:, 1], index[:, 0]] - output).abs().sum()
    assert vh.item() == pytest.approx(vh_ref.item(), rel=1e-12, abs=1e-12)
    sm = (-torch.log(1 + scene.dihedral_angle())).sum()
    assert sm.item() == pytest.approx(sm_ref.item(), rel=1e-10)
    gv, = torch.autograd.grad(vh, V, retain_graph=True, allow_unused=True)
    gs, = torch.autograd.grad(sm, V)
    gv = torch.zeros_like(V) if gv is None else gv
    scale_vh, scale_sm = max(1e-30, g_vh.abs().max().item()), max(1e-30, g_sm.abs().max().item())
    assert (gv.cpu() - g_vh).abs().max().item() <= 1e-8 * scale_vh + 1e-12
    assert (gs.cpu() - g_sm).abs().max().item() <= 1e-8 * scale_sm
    # the one-kernel forms
    V2 = V.detach().clone().requires_grad_(True)
    scene.update_verticex(V2)
    vhf = scene.vh_loss_fused(camd, origin3.cuda(), soft.cuda())
    smf = scene.sm_loss_fused()
    assert vhf.item() == pytest.approx(vh_ref.item(), rel=1e-12, abs=1e-12) and smf.item() == pytest.approx(sm_ref.item(), rel=1e-10)
    (vhf + smf).backward()
    ref = g_vh + g_sm
    assert (V2.grad.cpu() - ref).abs().max().item() <= 1e-8 * max(1e-30, ref.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_random_soups_closest_point(seed):
    """drt_closest_point (the vertex-to-scan metric) on random triangle soups -- degenerate faces, slivers, coplanar duplicates
    included -- against the oracle's independent brute force: distances to 1e-9 of the scene size, the returned point on the returned
    face's plane at that distance.  Points: random, far away, exactly on vertices, on edges, on faces."""
    from drt_amd.optix_mesh import optix_mesh
    rng = np.random.default_rng(4000 + seed)
    scale = float(10.0 ** rng.uniform(-1, 2))
    if seed % 2:
        V, F = _soup(rng, int(rng.choice([1, 2, 8, 60, 500, 3000])), scale)
    else:
        V, F = _stack(rng, int(rng.choice([4, 40, 400])), scale)
    tri = V[F].astype(np.float64)
    n = 300
    f = rng.integers(0, len(F), n)
    w = rng.dirichlet([1, 1, 1], n)
    on_face = (tri[f] * w[:, :, None]).sum(1)
    on_vert = tri[f, rng.integers(0, 3, n)]
    on_edge = 0.5 * (tri[f, 0] + tri[f, 1])
    pts = np.concatenate([rng.uniform(-1.5, 1.5, (n, 3)) * scale, rng.uniform(-40, 40, (n, 3)) * scale, on_face, on_vert, on_edge,
                          on_face + rng.normal(size=(n, 3)) * 1e-4 * scale])
    t = optix_mesh(0)
    t.update_mesh(torch.tensor(F, device="cuda"), torch.tensor(V, device="cuda"))
    dist, face, closest = t.closest_point(torch.tensor(pts, device="cuda"), want_face=True, want_point=True)
    ref_d, _ = orc.point_mesh_distance(pts, V, F)
    size = np.abs(pts).max() + scale
    np.testing.assert_allclose(dist.cpu().numpy(), ref_d, rtol=1e-9, atol=1e-9 * size)
    # the returned point is at the returned distance and no closer to the surface than rounding
    np.testing.assert_allclose(np.linalg.norm(closest.cpu().numpy() - pts, axis=1), dist.cpu().numpy(), rtol=1e-9, atol=1e-9 * size)
    d_back, _ = orc.point_mesh_distance(closest.cpu().numpy(), V, F)
    assert d_back.max() <= 1e-6 * size
    assert int(face.min()) >= 0 and int(face.max()) < len(F)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_interleaved_scenes_and_streams(seed):
    """Three scenes, three caller streams, eighty operations in random order -- vertex updates, drop-in calls of one to three images, one-pass
    losses, backward passes, with the ahead-of-time fills forced on at every size and two sub-batches per call -- each result compared with
    the same operation done alone on a fresh scene: forward outputs bit for bit, losses to 1e-12, gradients to 1e-10.  (The library's internal
    streams, events and workspaces belong to a scene; the caller's stream may be any; nothing may leak from one call into another.)"""
    import os
    import subprocess
    import sys
    code = r"""
import numpy as np, torch, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from conftest import IOR, data_path
from drt_amd import diffrender as Render, mesh_io, views
seed = %d
Render.intIOR = IOR
rng = np.random.default_rng(50 + seed)
meshes = [mesh_io.read_ply(data_path("hand_vh.ply")), mesh_io.icosphere(3, radius=60.0, noise=0.05, seed=3), mesh_io.subdivide_midpoint(mesh_io.icosphere(2, radius=40.0, noise=0.1, seed=5))]
res = 128
Render.resx = Render.resy = res
cams = [views.turntable_cameras(*views.mesh_frame(m.vertices), 72, res, res, distance_factor=1.6) for m in meshes]
def rays(k, ids):
    r = [views.generate_ray(res, res, cams[k][i][3], cams[k][i][2], device="cuda") for i in ids]
    return torch.cat([x[0] for x in r]).contiguous(), torch.cat([x[1] for x in r]).contiguous()
ops = []
for _ in range(80):
    k = int(rng.integers(3)); kind = str(rng.choice(["update", "render", "fused", "render"]))
    ids = [int(i) for i in rng.integers(0, 72, int(rng.integers(1, 4)))]
    ops.append((k, kind, ids, float(rng.uniform(0.97, 1.03)), int(rng.integers(3)), int(rng.integers(1 << 30))))
def run(op, scene, V_state, stream):
    k, kind, ids, scale, _, s2 = op
    with torch.cuda.stream(stream):
        if kind == "update":
            V_state[k] = (torch.tensor(meshes[k].vertices, device="cuda") * scale).requires_grad_(True)
            scene.update_verticex(V_state[k])
            return None
        o, d = rays(k, ids)
        g = torch.Generator(device="cuda").manual_seed(s2)
        sp = torch.randn(o.shape, dtype=torch.float64, device="cuda", generator=g) * 30.0
        valid = torch.rand(len(o), device="cuda", generator=g) < 0.7
        V = V_state[k].detach().clone().requires_grad_(True)
        scene.update_verticex(V)
        if kind == "render":
            oo, od, mk = scene.render_transparent(o, d)
            loss = Render.ray_loss(oo, od, mk, sp, valid)
            loss.backward()
            return [oo.detach().clone(), od.detach().clone(), mk.clone(), loss.detach().clone(), V.grad.clone()]
        loss = scene.ray_loss_fused(o, d, sp, valid)
        loss.backward()
        return [loss.detach().clone(), V.grad.clone()]
# reference: every operation alone, on a fresh scene in its current vertex state, default stream
ref = []
V_state = [torch.tensor(m.vertices, device="cuda").requires_grad_(True) for m in meshes]
for op in ops:
    sc = Render.Scene(meshes[op[0]], 0)
    ref.append(run(op, sc, V_state, torch.cuda.current_stream()))
    torch.cuda.synchronize()
# the interleaved run: one scene per mesh, operations issued on three streams without synchronising in between
scenes = [Render.Scene(m, 0) for m in meshes]
streams = [torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()]
V_state = [torch.tensor(m.vertices, device="cuda").requires_grad_(True) for m in meshes]
got = []
last = [None, None, None]                     # the stream that last touched scene k: a caller orders ITS OWN uses of one scene
for op in ops:
    st = streams[op[4]]
    if last[op[0]] is not None and last[op[0]] is not st:
        st.wait_stream(last[op[0]])
    got.append(run(op, scenes[op[0]], V_state, st))
    last[op[0]] = st
torch.cuda.synchronize()
bad = 0
for a, b, op in zip(ref, got, ops):
    if a is None:
        continue
    if op[1] == "render":
        ok = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        ok = ok and abs(a[3].item() - b[3].item()) <= 1e-12 * max(1.0, abs(a[3].item())) and (a[4] - b[4]).abs().max().item() <= 1e-10 * max(1.0, a[4].abs().max().item())
    else:
        ok = abs(a[0].item() - b[0].item()) <= 1e-12 * max(1.0, abs(a[0].item())) and (a[1] - b[1]).abs().max().item() <= 1e-10 * max(1.0, a[1].abs().max().item())
    bad += not ok
print("operations", len(ops), "mismatches", bad)
sys.exit(1 if bad else 0)
""" % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), seed)
    env = dict(os.environ, DRT_PREFILL_MIN_RAYS="0", DRT_MIN_SUB_LOG2="13")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
