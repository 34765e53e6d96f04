"""Parity of the gfx950 kernels (through the C ABI) with the CPU oracle and the golden
vectors of the reference.  Face ids / masks / T bit-exact; float64 rays within 1e-9;
gradients within 1e-5 absolute (north_star bar), in practice ~1e-12 relative."""
import os

import numpy as np
import pytest
import torch

from conftest import BIG_FIXTURES, HEADLINE_FIXTURE, IOR, data_path, fixture_mesh, fixture_view, golden
from drt_amd import mesh_io, views
from oracle import diffrender_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Render():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from drt_amd import diffrender
    diffrender.intIOR = IOR
    return diffrender


@pytest.fixture(scope="module")
def hand():
    return mesh_io.read_ply(data_path("hand_vh.ply"))


@pytest.fixture(scope="module")
def horse50k():
    return mesh_io.subdivide_midpoint(mesh_io.read_ply(data_path("horse_vh.ply")))


def _tracer(mesh):
    from drt_amd.optix_mesh import optix_mesh
    t = optix_mesh(0)
    t.update_mesh(torch.tensor(mesh.faces, dtype=torch.int32, device="cuda"),
                  torch.tensor(mesh.vertices, dtype=torch.float32, device="cuda"))
    return t


def _camera_rays(mesh, res, view):
    c, ext = views.mesh_frame(mesh.vertices)
    R, K, Rinv, Kinv = views.turntable_cameras(c, ext, 72, res, res)[view]
    o, d = views.generate_ray(res, res, Kinv, Rinv)
    return torch.cat([o.float(), d.float()], 1)


def test_b1_intersect_equals_oracle_bruteforce(hand):
    t = _tracer(hand)
    bad, height = t.check()
    assert bad == 0 and 12 <= height <= 48
    assert 1 <= t.wide_depth <= height and 3 * t.wide_depth <= 64, (t.wide_depth, height)   # 3 postponed children per level fit the stack
    print(f"hand: binary height {height}, wide depth {t.wide_depth}")
    f32 = hand.faces.astype(np.int32); v32 = hand.vertices.astype(np.float32)
    rng = np.random.default_rng(3)
    c, ext = views.mesh_frame(hand.vertices)
    batches = [_camera_rays(hand, 128, v).numpy() for v in (0, 5, 23, 41)]
    o = rng.uniform(-1, 1, (20000, 3)) * ext + c
    d = rng.standard_normal((20000, 3)) * rng.uniform(0.1, 30, (20000, 1))
    d[:300, 0] = 0; d[300:600, 1] = 0; d[600:900, 2] = 0; d[900:1000, :2] = 0
    batches.append(np.concatenate([o, d], 1).astype(np.float32))
    tgt = np.concatenate([hand.vertices, 0.5 * (hand.vertices[hand.faces[:, 0]] + hand.vertices[hand.faces[:, 1]])])
    eye = c + np.array([0, 0, 2.5 * ext])
    batches.append(np.concatenate([np.broadcast_to(eye, tgt.shape), tgt - eye], 1).astype(np.float32))
    for rays in batches:
        T, ID = t.intersect(torch.tensor(rays, device="cuda"))
        To, IDo = orc.trace_closest(f32, v32, rays)
        assert np.array_equal(ID.cpu().numpy(), IDo)
        assert np.array_equal(T.cpu().numpy(), To)
        assert T.is_contiguous() and ID.is_contiguous() and ID.dtype == torch.int32
        hit = t.intersect_any(torch.tensor(rays, device="cuda"))
        assert np.array_equal(hit.cpu().numpy(), IDo >= 0)
        Tb, IDb = t.intersect_bruteforce(torch.tensor(rays, device="cuda"))
        assert np.array_equal(IDb.cpu().numpy(), IDo) and np.array_equal(Tb.cpu().numpy(), To)


def test_b1_edge_cases(hand):
    from drt_amd.optix_mesh import optix_mesh
    t = optix_mesh(0)
    with pytest.raises(AssertionError):
        t.intersect(torch.zeros((1, 6), device="cuda"))
    verts = torch.tensor([[0, 0, 0], [1, 0, 0], [0, 1, 0.]], dtype=torch.float32, device="cuda")
    rays = torch.tensor([[0.2, 0.2, 1, 0, 0, -1], [5, 5, 1, 0, 0, -1], [0.2, 0.2, -1, 0, 0, -1]], dtype=torch.float32, device="cuda")
    for faces in ([[0, 1, 2]], [[0, 1, 2], [0, 1, 2], [0, 1, 2]], [[0, 1, 2], [0, 0, 0]]):
        t.update_mesh(torch.tensor(faces, dtype=torch.int32, device="cuda"), verts)
        assert t.check()[0] == 0
        T, ID = t.intersect(rays)
        assert ID.tolist() == [0, -1, -1] and T.tolist() == [1.0, -1.0, -1.0]
    T, ID = t.intersect(torch.zeros((0, 6), dtype=torch.float32, device="cuda"))      # empty ray set
    assert T.shape == (0,) and ID.shape == (0,)
    t.update_mesh(torch.zeros((0, 3), dtype=torch.int32, device="cuda"), verts)        # empty mesh: all miss
    T, ID = t.intersect(rays)
    assert ID.tolist() == [-1, -1, -1]
    with pytest.raises(RuntimeError):
        t.intersect(rays.double())
    with pytest.raises(RuntimeError):
        t.intersect(rays.cpu())
    with pytest.raises(RuntimeError):
        t.update_vert(torch.zeros((7, 3), dtype=torch.float32, device="cuda"))   # wrong vertex count
    # non-contiguous input is handled (the reference misreads it)
    t.update_mesh(torch.tensor([[0, 1, 2]], dtype=torch.int32, device="cuda"), verts)
    wide = torch.zeros((3, 12), dtype=torch.float32, device="cuda"); wide[:, ::2] = rays
    assert t.intersect(wide[:, ::2])[1].tolist() == [0, -1, -1]


def test_lbvh_rebuild_is_sound_and_stable_under_load(horse50k):
    """Many rebuilds back to back with traversal launches in flight: the inter-workgroup box
    hand-off of the refit (agent-scope release/acquire) must never leave a stale box."""
    t = _tracer(horse50k)
    V = torch.tensor(horse50k.vertices, dtype=torch.float32, device="cuda")
    rays = _camera_rays(horse50k, 512, 7).cuda()
    ref_T, ref_ID = t.intersect(rays)
    order0 = t.sorted_faces().clone()
    assert torch.equal(torch.sort(order0).values, torch.arange(len(horse50k.faces), device="cuda", dtype=torch.int32))
    g = torch.Generator(device="cuda").manual_seed(0)
    for it in range(40):
        # alternate between very different vertex sets: a stale box read from the previous build
        # (same addresses) would then fail the containment check instead of hiding inside the padding
        jitter = V + 1e-3 * torch.randn(V.shape, device="cuda", generator=g)
        if it % 2:
            jitter = 1.37 * jitter + 11.0
        t.update_vert(jitter)
        t.intersect(rays)                       # keep the chip busy between builds
        bad, height = t.check()
        assert bad == 0, f"iteration {it}: {bad} BVH violations"
        assert height <= 48 and 3 * t.wide_depth <= 64
    t.update_vert(V)
    T, ID = t.intersect(rays)
    assert torch.equal(ID, ref_ID) and torch.equal(T, ref_T)
    assert torch.equal(t.sorted_faces(), order0)


@pytest.mark.parametrize("res", [512, 1024])
def test_full_size_traversal_equals_gpu_bruteforce(horse50k, res):
    """BASELINE workload size (50 248 triangles, up to 1024x1024 rays): BVH result == exhaustive test."""
    t = _tracer(horse50k)
    bad, height = t.check()
    assert bad == 0 and 3 * t.wide_depth <= 64
    print(f"horse50k: binary height {height}, wide depth {t.wide_depth}")
    for view in (3, 40):
        rays = _camera_rays(horse50k, res, view).cuda()
        T, ID = t.intersect(rays)
        hit = ID >= 0
        assert 0.01 < hit.float().mean().item() < 0.6
        sel = torch.nonzero(hit).squeeze(1)
        # every hit ray + an equal number of misses, exhaustively
        miss = torch.nonzero(~hit).squeeze(1)[:: max(1, int((~hit).sum()) // max(1, len(sel)))]
        idx = torch.cat([sel, miss])
        Tb, IDb = t.intersect_bruteforce(rays[idx].contiguous())
        assert torch.equal(ID[idx], IDb)
        assert torch.equal(T[idx], Tb)


@pytest.mark.parametrize("name", [f"hand_r{r}_v{v}" for r in (64, 128) for v in (5, 23, 41)] + BIG_FIXTURES)
def test_render_transparent_vs_golden(Render, name):
    """Against the reference's own Python (tests/golden/make_golden.py): hand_vh at 64^2 / 128^2, and the HEADLINE mesh -- horse_vh x4,
    50 248 triangles, one view at 256^2 (DiffRender.py:420-432, optim.py:91-108 on BASELINE.json's workload)."""
    g = golden(name)
    hand = fixture_mesh(g)
    o, d, sp, valid = fixture_view(g)
    scene = Render.Scene(hand, 0)
    V = torch.tensor(hand.vertices, dtype=torch.float64, device="cuda", requires_grad=True)
    scene.update_verticex(V)
    out_ori, out_dir, mask = scene.render_transparent(o.cuda(), d.cuda())
    assert out_ori.dtype == torch.float64 and out_ori.shape == o.shape and mask.dtype == torch.bool and mask.shape == o.shape
    vi = torch.nonzero(mask[:, 0]).squeeze(1).cpu().numpy()
    assert np.array_equal(vi, g["valid_ind"])
    assert torch.equal(mask[:, 0], mask[:, 1]) and torch.equal(mask[:, 0], mask[:, 2])
    f1 = np.full(len(o), -1, np.int64); f1[g["b1_ind"]] = g["b1_face"]
    assert np.array_equal(scene.last_face1.cpu().numpy(), f1)           # hit ids bit-exact
    f2 = np.full(len(o), -1, np.int64); f2[g["b2_ind"]] = g["b2_face"]
    got2 = scene.last_face2.cpu().numpy()
    assert np.array_equal(got2[vi], f2[vi])                             # second-bounce ids of the completed paths
    np.testing.assert_allclose(out_ori[vi].detach().cpu().numpy(), g["out_ori"], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(out_dir[vi].detach().cpu().numpy(), g["out_dir"], rtol=1e-10, atol=1e-11)
    assert float(out_ori.detach()[~mask[:, 0]].abs().sum()) == 0.0
    loss = Render.ray_loss(out_ori, out_dir, mask, sp.cuda(), valid.cuda())
    assert loss.item() == pytest.approx(float(g["ray_loss"]), rel=1e-10)
    g_ray, = torch.autograd.grad(loss, V, retain_graph=True)
    ref = g["grad_ray_loss"]
    assert np.abs(g_ray.cpu().numpy() - ref).max() <= 1e-5
    np.testing.assert_allclose(g_ray.cpu().numpy(), ref, rtol=1e-7, atol=1e-9 * np.abs(ref).max())
    rng = np.random.default_rng(int(g["lin_seed"]))
    w_ori = torch.tensor(rng.standard_normal(o.shape), device="cuda"); w_dir = torch.tensor(rng.standard_normal(o.shape), device="cuda")
    lin = (out_ori * w_ori).sum() + (out_dir * w_dir).sum()
    assert lin.item() == pytest.approx(float(g["lin"]), rel=1e-9)
    g_lin, = torch.autograd.grad(lin, V)
    ref = g["grad_lin"]
    np.testing.assert_allclose(g_lin.cpu().numpy(), ref, rtol=1e-7, atol=1e-9 * np.abs(ref).max())
    # fused one-pass loss + gradient
    V2 = V.detach().clone().requires_grad_(True)
    scene.update_verticex(V2)
    lf = scene.ray_loss_fused(o.cuda(), d.cuda(), sp.cuda(), valid.cuda())
    assert lf.item() == pytest.approx(float(g["ray_loss"]), rel=1e-10)
    (3.0 * lf).backward()
    ref = 3.0 * g["grad_ray_loss"]
    np.testing.assert_allclose(V2.grad.cpu().numpy(), ref, rtol=1e-7, atol=1e-9 * np.abs(ref).max())


def test_render_50k_vs_oracle_sample(Render, horse50k):
    """50k-triangle mesh: the HIP path vs the oracle on a 96x96 view (the oracle is O(rays x faces))."""
    scene = Render.Scene(horse50k, 0)
    c, ext = views.mesh_frame(horse50k.vertices)
    R, K, Rinv, Kinv = views.turntable_cameras(c, ext, 72, 96, 96)[11]
    o, d = views.generate_ray(96, 96, Kinv, Rinv)
    V = torch.tensor(horse50k.vertices, dtype=torch.float64, device="cuda", requires_grad=True)
    scene.update_verticex(V)
    out_ori, out_dir, mask = scene.render_transparent(o.cuda(), d.cuda())
    Vc = torch.tensor(horse50k.vertices, dtype=torch.float64, requires_grad=True)
    oo, od, mk, aux = orc.render_transparent(orc.Mesh(horse50k.faces, Vc), o, d, IOR, return_aux=True)
    assert torch.equal(mask.cpu(), mk)
    assert torch.equal(scene.last_face1.cpu().long(), aux["face1"])
    f2 = aux["face2"].clone(); f2[~mk[:, 0]] = -1
    assert torch.equal(scene.last_face2.cpu().long(), f2)
    assert mk[:, 0].sum() > 200
    torch.testing.assert_close(out_dir.detach().cpu(), od.detach(), rtol=1e-10, atol=1e-11)
    torch.testing.assert_close(out_ori.detach().cpu(), oo.detach(), rtol=1e-10, atol=1e-9)
    w = torch.tensor(np.random.default_rng(5).standard_normal(o.shape))
    ((out_dir * w.cuda()).sum() + (out_ori * w.cuda()).sum()).backward()
    ((od * w).sum() + (oo * w).sum()).backward()
    assert (V.grad.cpu() - Vc.grad).abs().max().item() <= 1e-5 * max(1.0, Vc.grad.abs().max().item())


def test_properties_at_full_size(Render, horse50k):
    """Size-independent properties on 1024x1024 rays x 50k triangles."""
    scene = Render.Scene(horse50k, 0)
    c, ext = views.mesh_frame(horse50k.vertices)
    R, K, Rinv, Kinv = views.turntable_cameras(c, ext, 72, 1024, 1024)[20]
    o, d = views.generate_ray(1024, 1024, Kinv, Rinv, device="cuda")
    V = torch.tensor(horse50k.vertices, dtype=torch.float64, device="cuda", requires_grad=True)
    scene.update_verticex(V)
    out_ori, out_dir, mask = scene.render_transparent(o, d)
    m = mask[:, 0]
    f1, f2 = scene.last_face1, scene.last_face2
    assert 0.005 < m.float().mean().item() < 0.5
    assert torch.all(f2[m] >= 0) and torch.all(f2[~m] == -1) and torch.all(f1[m] >= 0)
    assert torch.all(out_ori[~m] == 0) and torch.all(out_dir[~m] == 0)
    nrm = out_dir.detach()[m].norm(dim=1)
    assert (nrm - 1).abs().max().item() < 1e-12                      # exit directions are unit vectors
    # primary face ids agree with the B1 query of the same rays
    ids, hitted = scene.optix_intersect(Render.Ray(o, d))
    assert torch.equal(hitted, f1 >= 0) and torch.equal(ids[hitted].int(), f1[hitted])
    # exit rays leave the object: re-tracing them hits nothing (that is the occlusion test)
    _, again = scene.optix_intersect(Render.Ray(out_ori.detach()[m], out_dir.detach()[m]))
    assert not again.any()
    # determinism of the forward, run-to-run, and independence from the 8x8-tile ordering hint
    o2, d2, m2 = scene.render_transparent(o, d)
    assert torch.equal(o2, out_ori) and torch.equal(d2, out_dir) and torch.equal(m2, mask)
    saved = (Render.resx, Render.resy)
    for rx, ry in ((1024, 1024), (7, 7)):          # tiled (whole 1024-wide images) vs linear order
        Render.resx, Render.resy = rx, ry
        o3, d3, m3 = scene.render_transparent(o, d)
        assert torch.equal(o3, out_ori) and torch.equal(d3, out_dir) and torch.equal(m3, mask)
        assert torch.equal(scene.last_face1, f1) and torch.equal(scene.last_face2, f2)
    Render.resx, Render.resy = saved
    # linearity of the adjoint in the incoming gradient, and fused == two-pass
    w = torch.randn(o.shape, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    g1, = torch.autograd.grad((out_dir * w).sum(), V, retain_graph=True)
    g2, = torch.autograd.grad((out_dir * (2.5 * w)).sum(), V, retain_graph=True)
    torch.testing.assert_close(g2, 2.5 * g1, rtol=1e-9, atol=1e-9 * g1.abs().max().item())
    sp = out_ori.detach() + 100.0 * out_dir.detach() + 0.3
    valid = m.clone()
    loss = Render.ray_loss(out_ori, out_dir, mask, sp, valid)
    ga, = torch.autograd.grad(loss, V)
    V2 = V.detach().clone().requires_grad_(True)
    scene.update_verticex(V2)
    lf = scene.ray_loss_fused(o, d, sp, valid)
    lf.backward()
    assert lf.item() == pytest.approx(loss.item(), rel=1e-10)
    torch.testing.assert_close(V2.grad, ga, rtol=1e-8, atol=1e-10 * ga.abs().max().item())


# ----------------------------------------------------------------------------- silhouette / smoothness branches
@pytest.mark.parametrize("name", [f"hand_r{r}_v{v}" for r in (64, 128) for v in (5, 23, 41)] + BIG_FIXTURES)
def test_silhouette_branch_vs_golden(Render, name):
    g = golden(name)
    hand = fixture_mesh(g)
    res = int(g["res"])
    Render.resx = Render.resy = res
    o, d, _, _ = fixture_view(g)
    scene = Render.Scene(hand, 0)
    if name not in BIG_FIXTURES:
        topo = golden("hand_topology")
        assert np.array_equal(scene.Edges.cpu().numpy(), topo["Edges"]) and np.array_equal(scene.E2F.cpu().numpy(), topo["E2F"])
        assert scene.mean_len == pytest.approx(float(topo["mean_len"]), rel=1e-14)
    V = torch.tensor(hand.vertices, dtype=torch.float64, device="cuda", requires_grad=True)
    scene.update_verticex(V)
    origin3 = o[0].cuda()
    sil = scene.silhouette_edge(origin3)
    assert np.array_equal(sil.cpu().numpy(), g["sil_edges"])
    cam = tuple(torch.tensor(g[k], dtype=torch.float64, device="cuda") for k in ("R", "K", "Rinv", "Kinv"))
    index, output = scene.primary_visibility(sil, cam, origin3, detach_depth=True)
    assert np.array_equal(index.cpu().numpy(), g["vh_index"])
    assert output.dtype == torch.float32 and np.array_equal(output.detach().cpu().numpy(), g["vh_output"])
    hit = np.zeros(res * res, dtype=np.uint8); hit[g["b1_ind"]] = 1
    soft = torch.tensor(views.process_mask(hit.reshape(res, res)), dtype=torch.float64, device="cuda").reshape(-1)
    vh = (soft.view((res, res))[index[:, 1], index[:, 0]] - output).abs().sum()
    assert vh.item() == pytest.approx(float(g["vh_loss"]), rel=1e-12)
    g_vh, = torch.autograd.grad(vh, V)
    ref = g["grad_vh"]
    assert np.abs(g_vh.cpu().numpy() - ref).max() <= 1e-5
    np.testing.assert_allclose(g_vh.cpu().numpy(), ref, rtol=1e-8, atol=1e-10 * np.abs(ref).max())
    # the dense-coefficient entry point of the C ABI (coef float64 [Es], 0 for dropped edges) gives the same gradient as the row-list one autograd used
    from drt_amd import _lib
    edges_all, flags = scene.Edges.contiguous(), scene.silhouette_edge(origin3)._flags
    n_e = edges_all.shape[0]
    idx2 = torch.empty((n_e, 2), dtype=torch.long, device="cuda"); f_all = torch.full((n_e,), 7.0, dtype=torch.float32, device="cuda")
    keep = torch.full((n_e,), 9, dtype=torch.uint8, device="cuda")
    camp = Render.pack_camera(cam); Vd = V.detach().contiguous(); st = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib().drt_edge_sample_forward(scene.optix_mesh._h, Vd.data_ptr(), edges_all.data_ptr(), n_e, camp.data_ptr(), origin3.data_ptr(),
                                                  idx2.data_ptr(), f_all.data_ptr(), keep.data_ptr(), res, res, flags.data_ptr(), st))
    assert int(keep.max()) <= 1 and float(f_all.abs().max()) <= 1.0              # every row written, dropped ones as zeros
    rows = torch.nonzero(keep).squeeze(1)
    assert torch.equal(idx2[rows], index)
    g_out = -torch.sign(soft.view((res, res))[index[:, 1], index[:, 0]] - output.detach().double())
    coef = torch.zeros(n_e, dtype=torch.float64, device="cuda"); coef[rows] = g_out
    gd = torch.zeros_like(Vd)
    _lib.check(_lib.lib().drt_edge_sample_backward(Vd.data_ptr(), edges_all.data_ptr(), n_e, camp.data_ptr(), f_all.data_ptr(), coef.data_ptr(), 1, gd.data_ptr(), st))
    np.testing.assert_allclose(gd.cpu().numpy(), ref, rtol=1e-8, atol=1e-10 * np.abs(ref).max())
    # fused one-kernel form of the same loss and gradient (no host sync)
    V3 = V.detach().clone().requires_grad_(True)
    scene.update_verticex(V3)
    vhf = scene.vh_loss_fused(cam, origin3, soft)
    assert vhf.item() == pytest.approx(float(g["vh_loss"]), rel=1e-12)
    (0.5 * vhf).backward()
    np.testing.assert_allclose(V3.grad.cpu().numpy(), 0.5 * ref, rtol=1e-8, atol=1e-10 * np.abs(ref).max())
    # render_mask == primary hit flags of the fixture
    rm = scene.render_mask(o.cuda(), d.cuda())
    assert np.array_equal(np.flatnonzero(rm.cpu().numpy() > 0), g["b1_ind"])


def test_dihedral_and_sm_loss_vs_golden(Render, hand):
    g = golden("hand_smooth_sm")
    scene = Render.Scene(data_path("hand_vh.ply"), 0)
    V = torch.tensor(g["vertices"].astype(np.float64), device="cuda", requires_grad=True)
    scene.update_verticex(V)
    cosang = scene.dihedral_angle()
    np.testing.assert_allclose(cosang.detach().cpu().numpy(), g["dihedral_cos"], rtol=1e-10, atol=1e-12)
    sm = (-torch.log(1 + cosang)).sum()
    assert sm.item() == pytest.approx(float(g["sm_loss"]), rel=1e-12)
    g_sm, = torch.autograd.grad(sm, V)
    ref = g["grad_sm"]
    np.testing.assert_allclose(g_sm.cpu().numpy(), ref, rtol=1e-8, atol=1e-10 * np.abs(ref).max())
    smf = scene.sm_loss_fused()
    assert smf.item() == pytest.approx(float(g["sm_loss"]), rel=1e-12)
    g_f, = torch.autograd.grad(2.0 * smf, V)
    np.testing.assert_allclose(g_f.cpu().numpy(), 2.0 * ref, rtol=1e-8, atol=1e-10 * np.abs(ref).max())


@pytest.mark.parametrize("fused", [False, True])
def test_two_optimisation_steps_vs_golden(Render, hand, fused):
    """The reference's Loss_calculator.all_loss + limit_hook + SGD(nesterov) for two iterations
    (golden from the imported optim.py), through drt_amd.optim on the GPU."""
    from drt_amd import optim as O
    g = golden("hand_smooth_steps")
    res = int(g["res"])
    Render.intIOR = float(g["ior"])
    Render.resx = Render.resy = res
    center, extent = views.mesh_frame(hand.vertices)
    cams = views.turntable_cameras(center, extent, 72, res, res)
    Vs = g["vertices"].astype(np.float64)
    scene = Render.Scene(mesh_io.TriMesh(Vs, hand.faces), 0)
    assert scene.mean_len != pytest.approx(float(g["mean_len"]))   # mean_len of the reference run is the RAW hull's
    scene.mean_len = float(g["mean_len"])
    base = orc.Mesh(hand.faces, torch.tensor(Vs))

    class Data:
        resx = resy = res

        def __init__(self):
            self.v = {}
            for k in list(g["ray_views"]) + list(g["sil_views"]):
                k = int(k)
                R, K, Rinv, Kinv = cams[k]
                o, d = views.generate_ray(res, res, Kinv, Rinv)
                rng = np.random.default_rng(100 + k)
                sp = rng.standard_normal((res * res, 3)) * 40.0 + np.asarray(center) + np.array([0.0, 0.0, 150.0])
                valid = rng.random(res * res) > 0.1
                _, hit = orc.intersect_ids(base, o, d)
                soft = torch.tensor(views.process_mask(hit.numpy().reshape(res, res)), dtype=torch.float64).reshape(-1)
                cam = tuple(torch.tensor(a, dtype=torch.float64, device="cuda") for a in (R, K, Rinv, Kinv))
                self.v[k] = (torch.tensor(sp).cuda(), torch.tensor(valid).cuda(), soft.cuda(), o.cuda(), d.cuda(), cam)

        def get_view(self, k):
            return self.v[k]

        def ray_view_generator(self):
            while True:
                for k in g["ray_views"]:
                    yield int(k)

        def silh_view_generator(self):
            while True:
                for k in g["sil_views"]:
                    yield int(k)

    hp = dict(O.HyperParams, IOR=float(g["ior"]), momentum=float(g["momentum"]))
    lc = O.Loss_calculator(scene, Data(), hp, fused=fused)
    init_vertices, parameter, opt = O.setup_opt(scene, float(g["lr"]), hp)
    for it in range(2):
        opt.zero_grad()
        vertices = init_vertices + parameter
        scene.update_verticex(vertices)
        loss, parts = lc.all_loss()
        loss.backward()
        assert loss.item() == pytest.approx(float(g[f"loss{it}"]), rel=1e-10)
        assert O.loss_string(parts) == str(g[f"loss_str{it}"])
        np.testing.assert_allclose(parameter.grad.cpu().numpy(), g[f"grad{it}"], rtol=1e-7, atol=1e-10)
        opt.step()
        np.testing.assert_allclose(parameter.detach().cpu().numpy(), g[f"param{it}"], rtol=1e-7, atol=1e-11)
    assert scene.mesh.vertices.shape == Vs.shape     # lazy device->host copy of the optimised vertices
    np.testing.assert_allclose(scene.mesh.vertices, vertices.detach().cpu().numpy(), rtol=0, atol=0)


def test_optimize_loop_on_synthetic_capture(Render, hand):
    """drt_amd.optim.optimize (the reference's pass/iteration loop) on a synthetic capture: the smoothed hull is
    optimised towards a displaced ground truth; the ray loss must fall and every tensor stay finite."""
    from drt_amd import optim as O
    g = golden("hand_smooth_sm")
    Vs = g["vertices"].astype(np.float64)
    mesh = mesh_io.TriMesh(Vs, hand.faces)
    res = 128
    Render.intIOR = IOR
    Render.resx = Render.resy = res
    center, extent = views.mesh_frame(Vs)
    gt = views.displaced_ground_truth(mesh, sigma=0.15, seed=3)
    scene_gt = Render.Scene(gt, 0)
    data = O.SyntheticData(scene_gt, center, extent, res, res, num_view=8, n_total=8)
    tgt, valid, soft, origin, ray_dir, cam = data.get_view(3)
    assert tgt.shape == (res * res, 3) and valid.dtype == torch.bool and soft.shape == (res * res,)
    assert 0.01 < valid.float().mean().item() < 0.5 and float(soft.min()) >= 0 and float(soft.max()) <= 1
    assert torch.all((tgt[:, 0] != 0) == valid)
    scene = Render.Scene(mesh, 0)
    hp = dict(O.HyperParams, Pass=1, Iters=30, start_lr=0.05, vh_w=2e-3, sm_w=0.08, ray_w=40)

    def mean_ray_loss():
        with torch.no_grad():
            tot = 0.0
            for k in range(8):
                t, v, _, o, d, _ = data.get_view(k)
                oo, od, mk = scene.render_transparent(o, d)
                tot += Render.ray_loss(oo, od, mk, t, v).item()
        return tot / 8

    scene.update_verticex(scene.vertices.detach())
    before = mean_ray_loss()
    scene, hist = O.optimize(scene, data, hp, remesh=None, output=False)
    after = mean_ray_loss()
    assert np.isfinite(before) and np.isfinite(after) and torch.isfinite(scene.vertices).all()
    assert after < 0.8 * before, (before, after)
    out = "/tmp/drt_amd_recons.ply"
    scene.mesh.export(out)
    back = mesh_io.read_ply(out)
    assert back.is_watertight and len(back.faces) == len(hand.faces)
    np.testing.assert_allclose(back.vertices, scene.vertices.detach().cpu().numpy().astype(np.float32), rtol=0, atol=0)


def test_stack_overflow_paths_with_a_tiny_lds_stack(tmp_path):
    """Build the library with a 3-entry LDS traversal stack so that (a) the spilling stack of the B1 / probe
    kernels and (b) the overflow -> k_trace_redo hand-off of the pipeline are exercised on every other ray,
    then repeat the golden and brute-force comparisons in a subprocess that loads that build."""
    import subprocess
    import sys
    from drt_amd import build
    so = str(tmp_path / "libdrt_hip_stack3.so")
    build.build(force=True, out=so, extra_flags=("-DDRT_STACK_FAST=3",))
    sel = ("test_b1_intersect_equals_oracle_bruteforce or test_render_transparent_vs_golden or test_full_size_traversal or test_silhouette_branch_vs_golden or "
           "test_random_meshes_and_adversarial_rays or test_random_shapes_cameras_and_ior")
    fuzz = os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_fuzz.py")
    for extra in ({}, {"DRT_MEGA_MAX_LOG2": "24"}):          # (the staged kernels' redo passes; k_path's redo pass)
        env = dict(os.environ, DRT_HIP_LIB=so, **extra)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), fuzz, "-m", "gpu", "-x", "-q", "-k",
                            sel if not extra else "test_render_transparent_vs_golden or test_two_optimisation_steps or test_random_shapes_cameras_and_ior"],
                           env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        assert " passed" in r.stdout


def test_fused_iteration_without_autograd_equals_the_autograd_loop(Render, hand):
    """optim.FusedIteration (the three one-pass terms written into one buffer, weighted sum, one-kernel limit_hook + SGD; no autograd graph,
    terms on two streams) must walk the SAME trajectory as Loss_calculator(fused=True) + backward() + FusedLimitSGD, and that one the same
    as the drop-in terms with the reference's hook + torch.optim.SGD: same view schedule (seeded generator), first-iteration losses equal to 1e-12,
    parameters and loss histories to 1e-7 after five iterations."""
    from drt_amd import optim as O
    g = golden("hand_smooth_sm")
    Vs = g["vertices"].astype(np.float64)
    mesh = mesh_io.TriMesh(Vs, hand.faces)
    res = 128
    Render.intIOR = IOR
    Render.resx = Render.resy = res
    center, extent = views.mesh_frame(Vs)
    scene_gt = Render.Scene(views.displaced_ground_truth(mesh, sigma=0.15, seed=3), 0)
    hp = dict(O.HyperParams, Pass=1, Iters=5, start_lr=0.05, vh_w=2e-3, sm_w=0.08, ray_w=40, num_view=12)

    def data_for(seed):
        return O.SyntheticData(scene_gt, center, extent, res, res, num_view=12, n_total=12, seed=seed)      # (seeded view schedule)

    def autograd_loop(fused):
        scene = Render.Scene(mesh, 0)
        lc = O.Loss_calculator(scene, data_for(7), hp, fused=fused)
        init_vertices, parameter, opt = O.setup_opt(scene, 0.05, hp, hook=not fused, fused=fused)
        parts_hist = []
        for _ in range(hp["Iters"]):
            opt.zero_grad()
            scene.update_verticex(init_vertices + parameter)
            loss, parts = lc.all_loss()
            loss.backward()
            opt.step()
            parts_hist.append([float(p.detach()) for p in parts] + [float(loss.detach())])
        return parameter.detach().clone(), np.array(parts_hist)

    p_drop, h_drop = autograd_loop(False)
    p_fused, h_fused = autograd_loop(True)
    scene = Render.Scene(mesh, 0)
    it = O.FusedIteration(scene, data_for(7), hp, 0.05)
    h_direct = []
    for _ in range(hp["Iters"]):
        total, parts = it.step()
        h_direct.append(parts.tolist() + [float(total)])
    h_direct = np.array(h_direct)
    scale = p_drop.abs().max().item()
    assert scale > 1e-3
    # (summation orders differ -- float64 atomics, the weighted sum as a matrix product -- and five SGD steps amplify the last bits: the
    # first iteration agrees to 1e-12, the parameters after the fifth to 1e-7)
    np.testing.assert_allclose(h_fused[0], h_drop[0], rtol=1e-12)
    np.testing.assert_allclose(h_direct[0], h_drop[0], rtol=1e-12)
    assert (p_fused - p_drop).abs().max().item() <= 1e-7 * scale, (p_fused - p_drop).abs().max().item() / scale
    assert (it.parameter - p_drop).abs().max().item() <= 1e-7 * scale, (it.parameter - p_drop).abs().max().item() / scale
    np.testing.assert_allclose(h_fused, h_drop, rtol=1e-7)
    np.testing.assert_allclose(h_direct, h_drop, rtol=1e-7)


@pytest.mark.parametrize("flags,extra", [(("-DDRT_CHECK=1",), {}), (("-DDRT_CHECK=1", "-DDRT_STACK_FAST=5"), {}), (("-DDRT_CHECK=1", "-DDRT_STACK_FAST=5"), {"DRT_MEGA_MAX_LOG2": "24"})],
                         ids=["default-stack", "five-entry-stack", "five-entry-stack-one-kernel-path"])
def test_lds_stack_invariants_hold_in_a_checked_build(tmp_path, flags, extra):
    """k_trace's LDS stack has no bound check per push: "a visit that starts with a legal stack stores at most three rows above it"
    (drt_traverse.h FastStack).  A -DDRT_CHECK=1 build ASSERTS that at every store, checks pops against underflow and validates
    poisoned guard rows around each block's stack; the golden, full-size brute-force and monkey (184 k triangles, deepest tree)
    comparisons run against that build in a subprocess, which ends by reading the violation counters: all zero.  Once with the
    default stack and once with a five-entry one (two usable rows + the spare ones: overflows on most rays, so the stores of
    OVERFLOWING visits are what gets checked)."""
    import subprocess
    import sys
    from drt_amd import build
    so = str(tmp_path / "libdrt_hip_check.so")
    build.build(force=True, out=so, extra_flags=flags)
    env = dict(os.environ, DRT_HIP_LIB=so, DRT_EXPECT_CHECKED="1", **extra)
    here = os.path.dirname(os.path.abspath(__file__))
    sel = ("test_b1_intersect_equals_oracle_bruteforce or test_render_transparent_vs_golden or test_full_size_traversal or "
           "test_silhouette_branch_vs_golden or test_two_optimisation_steps or test_random_meshes_and_adversarial_rays or "
           "test_random_shapes_cameras_and_ior or test_zz_check_counters")         # (the fuzz scenes: soups and stacks of duplicates make the deepest, most lopsided trees)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), os.path.join(here, "test_gpu_configs.py"), os.path.join(here, "test_gpu_fuzz.py"),
                        os.path.join(here, "test_gpu_zz_check.py"), "-m", "gpu", "-x", "-q", "-k", sel], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "test_zz_check_counters" not in r.stdout.split("short test summary")[-1]


def test_closest_point_and_hausdorff(Render, hand, horse50k):
    """drt_closest_point: oracle brute force on the hand; size-independent properties on the 50k mesh;
    the acceptance metric of drt_amd.metrics on a displaced copy."""
    from drt_amd import metrics
    rng = np.random.default_rng(5)
    t = _tracer(hand)
    V32 = hand.vertices.astype(np.float32)
    lo, hi = V32.min(0), V32.max(0)
    pts = np.concatenate([rng.uniform(lo - 0.3 * (hi - lo), hi + 0.3 * (hi - lo), size=(1500, 3)),
                          V32[rng.integers(0, len(V32), 200)].astype(np.float64),
                          rng.uniform(-1e4, 1e4, size=(20, 3))])
    dist, face, closest = t.closest_point(torch.tensor(pts, device="cuda"), want_face=True, want_point=True)
    ref_d, _ = orc.point_mesh_distance(pts, V32, hand.faces)
    np.testing.assert_allclose(dist.cpu().numpy(), ref_d, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose((torch.tensor(pts, device="cuda") - closest).norm(dim=1).cpu().numpy(), dist.cpu().numpy(), rtol=1e-12, atol=1e-12)
    assert dist[1500:1700].max().item() < 1e-12 and (face >= 0).all() and (face < len(hand.faces)).all()
    with pytest.raises(RuntimeError):
        t.closest_point(torch.tensor(pts, device="cuda", dtype=torch.float32))

    # 50k triangles: every vertex and every face centroid is on the surface; a point pushed out along
    # the face normal by h from a centroid is at most h away, and exactly h when h is small
    big = _tracer(horse50k)
    Vb = torch.tensor(horse50k.vertices.astype(np.float32), device="cuda").double()
    Fb = torch.tensor(horse50k.faces, device="cuda").long()
    assert big.closest_point(Vb)[0].max().item() < 1e-12
    tri = Vb[Fb]
    cen = tri.mean(1)
    n = torch.linalg.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    n = n / n.norm(dim=1, keepdim=True)
    assert big.closest_point(cen)[0].max().item() < 1e-5
    h = 1e-4
    d_out = big.closest_point(cen + h * n)[0]
    assert d_out.max().item() <= h * (1 + 1e-9) + 1e-6
    assert (d_out - h).abs().median().item() < 1e-9
    # 1-Lipschitz in the query point
    q = torch.tensor(rng.uniform(-150, 150, size=(20000, 3)), device="cuda")
    dq = torch.tensor(rng.normal(0, 1.0, size=(20000, 3)), device="cuda")
    assert ((big.closest_point(q)[0] - big.closest_point(q + dq)[0]).abs() <= dq.norm(dim=1) * (1 + 1e-12) + 1e-12).all()

    # acceptance metric: result = scan displaced along vertex normals by N(0, 0.2)
    scan = Render.Scene(horse50k, 0)
    moved = views.displaced_ground_truth(horse50k, 0.2, 1)
    stats = metrics.hausdorff(moved, scan)
    assert stats["n"] == len(horse50k.vertices) and 0 <= stats["min"] <= stats["mean"] <= stats["rms"] <= stats["max"] < 1.5
    assert 0.05 < stats["mean"] < 0.2                      # E|N(0, 0.2)| = 0.16, less where the surface curves
    sym = metrics.hausdorff(moved, scan, symmetric=True)
    assert sym["forward"] == stats and sym["hausdorff"] >= stats["max"]
    assert metrics.hausdorff(scan, scan)["max"] < 1e-12


def test_full_loop_with_remesh_towards_the_scan(Render):
    """The reference's whole loop (optim.py:190-215): remesh to the pass's target length, optimise, next pass --
    on a synthetic capture traced through the scanned horse, starting from its visual hull; the acceptance
    metric (vertex-to-scan distance, README.md:11) must improve."""
    from drt_amd import metrics, optim as O
    res = 192
    Render.intIOR = IOR
    Render.resx = Render.resy = res
    scan = mesh_io.read_ply(data_path("horse_scan.ply"))
    hull = mesh_io.read_ply(data_path("horse_vh.ply"))
    center, extent = views.mesh_frame(scan.vertices)
    scan_scene = Render.Scene(scan, 0)
    data = O.SyntheticData(scan_scene, center, extent, res, res, num_view=24, n_total=24)
    scene = Render.Scene(hull, 0)
    before = metrics.hausdorff(scene, scan_scene)
    hp = dict(O.HyperParams, Pass=3, Iters=60, start_len=6.0, end_len=4.0, start_lr=0.1, num_view=24)
    scene, _ = O.optimize(scene, data, hp, output=False, fused=True)
    after = metrics.hausdorff(scene, scan_scene)
    m = scene.mesh
    assert m.is_watertight and len(m.faces) != len(hull.faces) and torch.isfinite(scene.vertices).all()
    el = np.linalg.norm(m.vertices[m.edges[:, 0]] - m.vertices[m.edges[:, 1]], axis=1)
    assert 3.0 < el.mean() < 5.0                                     # last pass was remeshed to 4 mm
    assert after["mean"] < before["mean"], (before, after)


def test_large_mesh_uses_the_unfused_sort_and_stays_exact():
    """327 680 triangles: above the 262 144 where the build switches from the fused radix passes to the
    three-kernel passes; the tree must be sound and the traversal must equal the exhaustive test."""
    sphere = mesh_io.icosphere(7, radius=60.0, noise=0.02, seed=4)
    assert len(sphere.faces) == 327680
    t = _tracer(sphere)
    bad, height = t.check()
    assert bad == 0 and 3 * t.wide_depth <= 64, (bad, height, t.wide_depth)
    order = t.sorted_faces().cpu().numpy()
    assert np.array_equal(np.sort(order), np.arange(len(sphere.faces)))
    rays = _camera_rays(sphere, 192, 11).cuda()
    T, ID = t.intersect(rays)
    assert 0.1 < (ID >= 0).float().mean().item() < 0.9
    sel = torch.arange(0, rays.shape[0], 7, device="cuda")
    Tb, IDb = t.intersect_bruteforce(rays[sel].contiguous())
    assert torch.equal(ID[sel], IDb) and torch.equal(T[sel], Tb)
    # and the pipeline on it: exit rays of valid paths are unit vectors, re-tracing them hits nothing
    from drt_amd import diffrender as Render
    Render.intIOR = IOR
    scene = Render.Scene(sphere, 0)
    o, d = rays[:, :3].double().contiguous(), rays[:, 3:].double().contiguous()
    oo, od, m = scene.render_transparent(o, d)
    ok = m[:, 0]
    assert 0.005 < ok.float().mean().item() < 0.9, ok.float().mean().item()
    assert (od[ok].norm(dim=1) - 1).abs().max().item() < 1e-12
    assert not scene.optix_mesh.intersect_any(torch.cat([oo[ok].float(), od[ok].float()], 1)).any()


@pytest.mark.parametrize("name", ["hand_r64_v5", "hand_r128_v41"])
def test_stepwise_methods_vs_golden(Render, hand, name):
    """Scene.Dintersect / refract_ray / trace2 / project_vert (the reference's internal methods, DiffRender.py:481-546)
    bounce by bounce against the reference's own intermediate values."""
    g = golden(name)
    res = int(g["res"])
    Render.resx = Render.resy = res
    Render.intIOR = float(g["ior"])
    o, d, _, _ = fixture_view(g)
    scene = Render.Scene(data_path("hand_vh.ply"), 0)
    V = torch.tensor(hand.vertices, dtype=torch.float64, device="cuda", requires_grad=True)
    scene.update_verticex(V)
    ray = Render.Ray(o.cuda(), d.cuda())
    hit1, hitted = scene.Dintersect(ray)
    assert isinstance(hit1, Render.Intersection) and int(hitted.sum()) == int(g["hit1_count"]) == len(hit1)
    np.testing.assert_array_equal(hit1.ray.ray_ind.cpu().numpy(), g["b1_ind"])
    np.testing.assert_array_equal(hit1.faces_ind.cpu().numpy(), g["b1_face"])
    for key, val in (("b1_u", hit1.u), ("b1_v", hit1.v), ("b1_t", hit1.t)):
        np.testing.assert_allclose(val.detach().cpu().numpy(), g[key], rtol=1e-10, atol=1e-12)
    refracted, inside = scene.refract_ray(hit1)
    np.testing.assert_allclose(hit1.n.detach().cpu().numpy(), g["b1_n"], rtol=1e-10, atol=1e-12)      # flipped in place where leaving
    np.testing.assert_array_equal(refracted.cpu().numpy(), g["b1_refracted"])
    np.testing.assert_allclose(inside.origin.detach().cpu().numpy(), g["b1_new_o"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(inside.direction.detach().cpu().numpy(), g["b1_new_d"], rtol=1e-10, atol=1e-12)
    # both bounces, then the same loss as through the fused pipeline: values and vertex gradient agree
    out = scene.trace2(ray)
    occluded = scene.optix_intersect(out)[1]
    np.testing.assert_array_equal(occluded.cpu().numpy(), g["occluded"])
    keep = torch.logical_not(occluded)
    np.testing.assert_array_equal(out.ray_ind[keep].cpu().numpy(), g["valid_ind"])
    np.testing.assert_allclose(out.origin[keep].detach().cpu().numpy(), g["out_ori"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(out.direction[keep].detach().cpu().numpy(), g["out_dir"], rtol=1e-10, atol=1e-12)
    rng = np.random.default_rng(int(g["lin_seed"]))
    w_ori = torch.tensor(rng.standard_normal((res * res, 3)), device="cuda")
    w_dir = torch.tensor(rng.standard_normal((res * res, 3)), device="cuda")
    idx = out.ray_ind[keep]
    lin = (out.origin[keep] * w_ori[idx]).sum() + (out.direction[keep] * w_dir[idx]).sum()
    assert lin.item() == pytest.approx(float(g["lin"]), rel=1e-10)
    lin.backward()
    ref = g["grad_lin"]
    np.testing.assert_allclose(V.grad.cpu().numpy(), ref, rtol=1e-7, atol=1e-9 * np.abs(ref).max())
    # project_vert: truncation towards zero of K R [V|1]
    cam = tuple(torch.tensor(g[k], dtype=torch.float64, device="cuda") for k in ("R", "K", "Rinv", "Kinv"))
    pix = scene.project_vert(cam, V.detach())
    hom = np.concatenate([hand.vertices, np.ones((len(hand.vertices), 1))], 1)
    c = g["K"] @ (g["R"] @ hom.T)[:3]
    np.testing.assert_array_equal(pix.cpu().numpy(), np.trunc(c[:2] / c[2]).T.astype(np.int64))


def test_pipeline_ragged_and_empty_inputs(Render, hand):
    """render_transparent / ray_loss / the fused loss on ray counts that are not multiples of anything (and zero):
    every ray is independent, so a prefix of a view must give the prefix of the view's result, bit for bit."""
    g = golden("hand_r64_v23")
    res = int(g["res"])
    Render.resx = Render.resy = res
    Render.intIOR = float(g["ior"])
    o, d, sp, valid = fixture_view(g)
    o, d, sp, valid = o.cuda(), d.cuda(), sp.cuda(), valid.cuda()
    scene = Render.Scene(data_path("hand_vh.ply"), 0)
    V = torch.tensor(hand.vertices, dtype=torch.float64, device="cuda")
    scene.update_verticex(V)
    with torch.no_grad():
        full = scene.render_transparent(o, d)
    first_valid = int(torch.nonzero(full[2][:, 0])[0])
    for n in (0, 1, 63, 257, 1000, first_valid + 1, res * res - 1):
        Vn = V.clone().requires_grad_(True)
        scene.update_verticex(Vn)
        oo, od, m = scene.render_transparent(o[:n].contiguous(), d[:n].contiguous())
        assert oo.shape == (n, 3) and od.shape == (n, 3) and m.shape == (n, 3) and m.dtype == torch.bool
        assert torch.equal(oo.detach(), full[0][:n]) and torch.equal(od.detach(), full[1][:n]) and torch.equal(m, full[2][:n])
        loss = Render.ray_loss(oo, od, m, sp[:n].contiguous(), valid[:n].contiguous())
        fused = scene.ray_loss_fused(o[:n].contiguous(), d[:n].contiguous(), sp[:n].contiguous(), valid[:n].contiguous())
        assert fused.item() == pytest.approx(loss.item(), rel=1e-12, abs=1e-300)
        if n == 0 or not bool((m[:, 0] & valid[:n]).any()):
            assert loss.item() == 0.0
        loss.backward()
        grad = Vn.grad.clone()
        assert torch.isfinite(grad).all()
        Vf = V.clone().requires_grad_(True)
        scene.update_verticex(Vf)
        scene.ray_loss_fused(o[:n].contiguous(), d[:n].contiguous(), sp[:n].contiguous(), valid[:n].contiguous()).backward()
        assert torch.allclose(Vf.grad, grad, rtol=1e-9, atol=1e-12 * max(1.0, grad.abs().max().item()))
    # a loss with no target at all
    scene.update_verticex(V.clone().requires_grad_(True))
    oo, od, m = scene.render_transparent(o, d)
    assert Render.ray_loss(oo, od, m, sp, torch.zeros_like(valid)).item() == 0.0


def test_silhouette_loss_edge_cases(Render, hand):
    """Fused silhouette loss over more views than one launch takes (16), with an off-screen camera among them, equals the
    sum of the drop-in per-view terms; zero views and zero query points are fine."""
    res = 96
    Render.resx = Render.resy = res
    scene = Render.Scene(data_path("hand_vh.ply"), 0)
    V = torch.tensor(hand.vertices, dtype=torch.float64, device="cuda")
    c, ext = views.mesh_frame(hand.vertices)
    cams = views.turntable_cameras(c, ext, 19, res, res)
    rng = np.random.default_rng(2)
    vs = []
    for k, cam in enumerate(cams):
        R, K, Rinv, Kinv = [np.array(a) for a in cam]
        if k == 7:                                   # principal point far off: every sample falls outside the image
            K = K.copy(); K[0, 2] += 10 * res; Kinv = np.linalg.inv(K)
        cam_t = tuple(torch.tensor(a, dtype=torch.float64, device="cuda") for a in (R, K, Rinv, Kinv))
        origin3 = torch.tensor(Rinv[:3, 3], dtype=torch.float64, device="cuda")
        soft = torch.tensor(rng.random(res * res), dtype=torch.float64, device="cuda")
        vs.append((cam_t, origin3, soft))
    Va = V.clone().requires_grad_(True)
    scene.update_verticex(Va)
    total = scene.vh_loss_fused_views(vs)
    total.backward()
    Vb = V.clone().requires_grad_(True)
    scene.update_verticex(Vb)
    ref = torch.zeros((), dtype=torch.float64, device="cuda")
    for k, (cam_t, origin3, soft) in enumerate(vs):
        edges = scene.silhouette_edge(origin3)
        pix, out = scene.primary_visibility(edges, cam_t, origin3, detach_depth=True)
        if k == 7:
            assert pix.shape[0] == 0 and out.shape[0] == 0
        ref = ref + (soft.view(res, res)[pix[:, 1], pix[:, 0]] - out).abs().sum()
    ref.backward()
    assert total.item() == pytest.approx(ref.item(), rel=1e-12)
    assert torch.allclose(Va.grad, Vb.grad, rtol=1e-9, atol=1e-12 * Vb.grad.abs().max().item())
    assert scene.vh_loss_fused_views([]).item() == 0.0
    dist, face, closest = scene.optix_mesh.closest_point(torch.zeros((0, 3), dtype=torch.float64, device="cuda"), want_point=True)
    assert dist.shape == (0,) and face.shape == (0,) and closest.shape == (0, 3)


def test_b1_torch_extension_used_like_the_reference(hand, horse50k):
    """drt_amd/optix.so (csrc/optix_hip.cpp, the pybind11 class of optix_extend.cpp:77-83) driven the way
    DiffRender.py does: optix.optix_mesh(device), update_mesh(F int32, V float32), update_vert(V), intersect(Ray [N,6])
    -> [T, ID]; Scene.optix_intersect's `T > 0` (DiffRender.py:386-392).  On torch's current (non-default) stream."""
    import drt_amd.optix as optix
    mesh = optix.optix_mesh(0)
    with pytest.raises(RuntimeError):
        mesh.intersect(torch.zeros((1, 6), device="cuda"))               # update_mesh first
    F = torch.tensor(hand.faces, dtype=torch.int32, device="cuda")
    V = torch.tensor(hand.vertices, dtype=torch.float32, device="cuda")
    mesh.update_mesh(F, V)
    rays = _camera_rays(hand, 128, 5)
    To, IDo = orc.trace_closest(hand.faces.astype(np.int32), hand.vertices.astype(np.float32), rays.numpy())
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        T, ID = mesh.intersect(rays.cuda())
    side.synchronize()
    assert T.dtype == torch.float32 and ID.dtype == torch.int32 and T.is_contiguous() and ID.is_contiguous()
    assert np.array_equal(ID.cpu().numpy(), IDo) and np.array_equal(T.cpu().numpy(), To)
    hitted = T > 0
    assert torch.equal(hitted, ID >= 0)
    # update_vert: same topology, moved vertices -> same answer as a fresh update_mesh
    V2 = V * 1.25 + 3.0
    mesh.update_vert(V2)
    T2, ID2 = mesh.intersect(rays.cuda())
    To2, IDo2 = orc.trace_closest(hand.faces.astype(np.int32), V2.cpu().numpy(), rays.numpy())
    assert np.array_equal(ID2.cpu().numpy(), IDo2) and np.array_equal(T2.cpu().numpy(), To2)
    # validation instead of the reference's silent misreads
    with pytest.raises(RuntimeError):
        mesh.intersect(rays.cuda().double())
    with pytest.raises(RuntimeError):
        mesh.intersect(rays)                                               # CPU tensor
    with pytest.raises(RuntimeError):
        mesh.update_vert(V[:7])
    wide = torch.zeros((len(rays), 12), dtype=torch.float32, device="cuda"); wide[:, ::2] = rays.cuda()
    assert torch.equal(mesh.intersect(wide[:, ::2])[1], ID2)               # non-contiguous input
    T0, ID0 = mesh.intersect(torch.zeros((0, 6), dtype=torch.float32, device="cuda"))
    assert T0.shape == (0,) and ID0.shape == (0,)
    # the 50k workload through the same class == the ctypes class
    mesh.update_mesh(torch.tensor(horse50k.faces, dtype=torch.int32, device="cuda"), torch.tensor(horse50k.vertices, dtype=torch.float32, device="cuda"))
    big = _camera_rays(horse50k, 512, 7).cuda()
    Ta, IDa = mesh.intersect(big)
    Tb, IDb = _tracer(horse50k).intersect(big)
    assert torch.equal(IDa, IDb) and torch.equal(Ta, Tb) and 0.01 < (IDa >= 0).float().mean().item() < 0.6


def test_ray_loss_gradient_as_row_list_equals_the_dense_tensor(Render, hand):
    """ray_loss hands its gradient to render_transparent's backward as a row list (diffrender._GradLink) when out_dir is the
    tensor render_transparent returned; the dense [N,3] form (SPARSE_LOSS_GRAD = False, or a detached / re-wrapped out_dir)
    must give the same vertex gradient, alone, next to other consumers of out_dir, twice on one graph, and with two losses."""
    g = golden("hand_r128_v41")
    o, d, sp, valid = fixture_view(g)
    o, d, sp, valid = o.cuda(), d.cuda(), sp.cuda(), valid.cuda()
    Render.resx = Render.resy = int(g["res"])
    scene = Render.Scene(data_path("hand_vh.ply"), 0)
    w = torch.randn(o.shape, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))

    def grads(sparse):
        Render.SPARSE_LOSS_GRAD = sparse
        try:
            V = torch.tensor(hand.vertices, dtype=torch.float64, device="cuda", requires_grad=True)
            scene.update_verticex(V)
            oo, od, mk = scene.render_transparent(o, d)
            l1 = Render.ray_loss(oo, od, mk, sp, valid)
            l2 = Render.ray_loss(oo, od, mk, sp + 1.0, valid)
            out = [torch.autograd.grad(2.5 * l1, V, retain_graph=True)[0],                        # alone, scaled
                   torch.autograd.grad(l1 + (od * w).sum() + (oo * w).sum(), V, retain_graph=True)[0],   # next to dense consumers
                   torch.autograd.grad(l1, V, retain_graph=True)[0],                              # again on the same graph
                   torch.autograd.grad(l1 + 0.5 * l2, V, retain_graph=True)[0]]                   # two losses on one out_dir
            assert l1.item() == pytest.approx(float(g["ray_loss"]), rel=1e-10)
            if not sparse:
                g_od, = torch.autograd.grad(l1, od)                                               # the dense form can be asked for
                assert g_od.shape == od.shape and int((g_od.abs().sum(dim=1) > 0).sum()) > 100
            return out
        finally:
            Render.SPARSE_LOSS_GRAD = True

    a, b = grads(True), grads(False)
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=1e-11, atol=1e-13 * y.abs().max().item())
    ref = g["grad_ray_loss"]
    np.testing.assert_allclose(a[2].cpu().numpy(), ref, rtol=1e-7, atol=1e-9 * np.abs(ref).max())
    np.testing.assert_allclose(a[0].cpu().numpy(), 2.5 * ref, rtol=1e-7, atol=1e-9 * np.abs(ref).max())


def test_eager_loss_gradient_equals_the_two_pass_form(Render, hand):
    """ray_loss on render_transparent's own outputs leaves d loss / d vertices (unit seed) in its forward pass
    (drt_ray_loss_listed_grad); EAGER_LOSS_GRAD = False, an output that is not the forward's own tensor, or a modified one, take
    the two-pass form (loss pass + recompute in the backward).  Same loss, same vertex gradient; no stash without grad mode."""
    g = golden("hand_r128_v41")
    o, d, sp, valid = fixture_view(g)
    o, d, sp, valid = o.cuda(), d.cuda(), sp.cuda(), valid.cuda()
    Render.resx = Render.resy = int(g["res"])
    scene = Render.Scene(data_path("hand_vh.ply"), 0)

    def run(eager, touch=None):
        Render.EAGER_LOSS_GRAD = eager
        try:
            V = torch.tensor(hand.vertices, dtype=torch.float64, device="cuda", requires_grad=True)
            scene.update_verticex(V)
            oo, od, mk = scene.render_transparent(o, d)
            if touch == "clone_ori":
                oo = oo.clone()                       # not the forward's own out_ori any more
            node = Render.ray_loss(oo, od, mk, sp, valid)
            had_stash = node.grad_fn.stash is not None
            gv, = torch.autograd.grad(1.75 * node, V, retain_graph=True)
            gv2, = torch.autograd.grad(node, V)                     # again on the same graph, another seed
            return float(node.detach()), gv, gv2, had_stash
        finally:
            Render.EAGER_LOSS_GRAD = True

    la, ga, ga2, sa = run(True)
    lb, gb, gb2, sb = run(False)
    lc, gc, gc2, sc = run(True, touch="clone_ori")
    assert sa and not sb and not sc
    assert la == pytest.approx(float(g["ray_loss"]), rel=1e-10) and lb == pytest.approx(la, rel=1e-12) and lc == pytest.approx(la, rel=1e-12)
    for x, y in ((ga, gb), (ga2, gb2), (gc, gb), (gc2, gb2)):
        assert torch.allclose(x, y, rtol=1e-11, atol=1e-13 * y.abs().max().item())
    assert torch.allclose(ga, 1.75 * ga2, rtol=1e-12, atol=0.0)
    ref = g["grad_ray_loss"]
    np.testing.assert_allclose(ga2.cpu().numpy(), ref, rtol=1e-7, atol=1e-9 * np.abs(ref).max())
    with torch.no_grad():                                           # no gradient asked for: the cheap loss pass, no stash
        oo, od, mk = scene.render_transparent(o, d)
        assert float(Render.ray_loss(oo, od, mk, sp, valid)) == pytest.approx(la, rel=1e-12)
    # a caller's OWN out_ori / mask (shifted origins, rows masked out, a row set that has no path): loss and gradient are those of the
    # tensors that were passed -- the forward's stored path must not be substituted for them
    V = torch.tensor(hand.vertices, dtype=torch.float64, device="cuda", requires_grad=True)
    scene.update_verticex(V)
    oo, od, mk = scene.render_transparent(o, d)
    oo2 = oo.detach() + 0.5
    mk2 = mk.clone()
    on = torch.nonzero(mk[:, 0]).squeeze(1)
    off = torch.nonzero(~mk[:, 0]).squeeze(1)
    mk2[on[::3]] = False
    mk2[off[:5]] = True                                             # rows without a completed path: od is zero there, no gradient reaches the mesh
    node = Render.ray_loss(oo2, od, mk2, sp, valid)
    gv, = torch.autograd.grad(node, V, retain_graph=True)
    sel = valid & mk2[:, 0]
    tgt = sp[sel] - oo2[sel]
    tgt = tgt / tgt.norm(dim=1, keepdim=True)
    ref_loss = ((od[sel] - tgt) ** 2).sum()
    gref, = torch.autograd.grad(ref_loss, V)
    assert float(node.detach()) == pytest.approx(float(ref_loss.detach()), rel=1e-12)
    assert torch.allclose(gv, gref, rtol=1e-10, atol=1e-13 * gref.abs().max().item())


def test_outputs_zeroed_ahead_of_time_equal_outputs_filled_in_the_call(Render, hand):
    """PREFILL_NEXT: a trusted-grid call allocates the out_ori / mask of the next call of its size and has them zeroed on the library's idle
    stream (drt_prefill_zero); the next call renders into them without filling them again.  Same outputs as with the fills inside the call,
    every row outside the mask exactly zero, over several steps with moving vertices; a call of another size in between, a scene dropped
    with buffers pending, and a graph capture after eager calls (which must not consume buffers zeroed outside the capture)."""
    g = golden("hand_r128_v41")
    o, d, sp, valid = fixture_view(g)
    o, d = o.cuda(), d.cuda()
    Render.resx = Render.resy = int(g["res"])
    n = o.shape[0]
    V0 = torch.tensor(hand.vertices, dtype=torch.float64, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(5)
    moves = [1e-3 * torch.randn(V0.shape, dtype=torch.float64, device="cuda", generator=gen) for _ in range(5)]
    recycle, Render.RECYCLE_OUTPUTS = Render.RECYCLE_OUTPUTS, False      # (recycled outputs -- tests/test_gpu_recycle.py -- take precedence over buffers zeroed ahead of time)
    try:
        _prefill_body(Render, o, d, n, V0, moves)
    finally:
        Render.RECYCLE_OUTPUTS = recycle


def _prefill_body(Render, o, d, n, V0, moves):
    def run(prefill):
        old = Render.PREFILL_NEXT, Render.PREFILL_MIN_RAYS
        Render.PREFILL_NEXT, Render.PREFILL_MIN_RAYS = prefill, 0
        try:
            scene = Render.Scene(data_path("hand_vh.ply"), 0)
            outs, used = [], []
            for k, mv in enumerate(moves):
                scene.update_verticex(V0 + mv)
                if k == 3:                       # a call of another size in between: the pending buffers are dropped, not used
                    scene.render_transparent(o[: n // 2].contiguous(), d[: n // 2].contiguous())
                pending = getattr(scene.optix_mesh, "_prefilled", None)
                oo, od, mk = scene.render_transparent(o, d)
                used.append(pending is not None and pending[0] == n and oo.data_ptr() == pending[2].data_ptr())
                outs.append((oo.clone(), od.clone(), mk.clone()))
            return outs, used, scene
        finally:
            Render.PREFILL_NEXT, Render.PREFILL_MIN_RAYS = old

    a, used, scene_a = run(True)
    b, unused, _ = run(False)
    assert used[2] and used[4] and not used[0] and not any(unused)      # (the first two calls establish / read back the grid verdict)
    for (oo, od, mk), (oo2, od2, mk2) in zip(a, b):
        assert torch.equal(mk, mk2) and torch.equal(oo, oo2) and torch.equal(od, od2)
        dead = ~mk[:, 0]
        assert int(mk.sum()) > 300 and float(oo[dead].abs().sum()) == 0.0 and float(od[dead].abs().sum()) == 0.0 and not mk[dead].any()
    assert getattr(scene_a.optix_mesh, "_prefilled", None) is not None
    del scene_a                                   # buffers pending: drt_destroy waits for the zeroing before they are released
    torch.cuda.synchronize()
    # capture after eager calls: the captured call allocates and fills its own outputs; replays give the eager result
    Render.PREFILL_NEXT, Render.PREFILL_MIN_RAYS, keep = True, 0, (Render.PREFILL_NEXT, Render.PREFILL_MIN_RAYS)
    try:
        scene = Render.Scene(data_path("hand_vh.ply"), 0)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(3):
                scene.update_verticex(V0)
                ref = scene.render_transparent(o, d)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert getattr(scene.optix_mesh, "_prefilled", None) is not None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            scene.update_verticex(V0)
            got = scene.render_transparent(o, d)
        for _ in range(2):
            got[0].fill_(7.0); got[2].fill_(True)          # what a missing fill inside the graph would leave behind
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(got[2], ref[2]) and torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    finally:
        Render.PREFILL_NEXT, Render.PREFILL_MIN_RAYS = keep


def test_fused_limit_sgd_equals_hook_plus_torch_sgd():
    """drt_amd.optim.FusedLimitSGD (one kernel) against limit_hook + torch.optim.SGD(momentum, nesterov) over five steps with NaN,
    inf and out-of-range gradient entries (reference optim.py:155-171, 215)."""
    from drt_amd import optim as O
    g = torch.Generator(device="cuda").manual_seed(0)
    p0 = torch.randn((1000, 3), dtype=torch.float64, device="cuda", generator=g)
    pa = p0.clone().requires_grad_(True)
    pb = p0.clone().requires_grad_(True)
    ref = torch.optim.SGD([pa], lr=0.1, momentum=0.95, nesterov=True, foreach=False)
    fus = O.FusedLimitSGD(pb, 0.1, 0.95, nesterov=True)
    for it in range(5):
        grad = 3.0 * torch.randn(p0.shape, dtype=torch.float64, device="cuda", generator=g)
        grad[it, 0] = float("nan"); grad[it + 7, 1] = float("inf"); grad[it + 9, 2] = -float("inf")
        pa.grad = O.limit_hook(grad.clone())
        ref.step()
        pb.grad = grad.clone()
        fus.step()
        assert torch.equal(pb.grad, pa.grad)                                   # sanitised in place, like the hook
        torch.testing.assert_close(pb.detach(), pa.detach(), rtol=1e-14, atol=1e-15)
    assert torch.isfinite(pb).all()
