"""The CPU oracle against golden vectors produced by the reference's own Python
(tests/golden/make_golden.py).  Integer results exact, float64 within 1e-12 relative."""
import numpy as np
import pytest
import torch

from conftest import BIG_FIXTURES, HEADLINE_FIXTURE, IOR, data_path, fixture_mesh, fixture_view, golden
from drt_amd import mesh_io, views
from oracle import diffrender_oracle as orc

RTOL = 1e-10   # float64 fields: the oracle is not bit-identical to TorchScript (op order inside cross/norm); 1e-10 covers cancellation in t
FIXTURES = [f"hand_r{r}_v{v}" for r in (64, 128) for v in (5, 23, 41)]


def close(a, b, rtol=RTOL, atol=1e-11):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


@pytest.fixture(scope="module")
def hand():
    return mesh_io.read_ply(data_path("hand_vh.ply"))


def test_topology_tables(hand):
    g = golden("hand_topology")
    # (the product builds these tables with a radix sort in libdrt_hip, tests/test_gpu_topology.py; no CPU path there)
    from drt_amd import diffrender
    with pytest.raises(RuntimeError):
        diffrender.edge_tables(torch.tensor(hand.faces), torch.tensor(hand.vertices))
    edges, e2f, mean_len = mesh_io.edge_tables(hand)
    assert hand.is_watertight
    assert np.array_equal(edges, g["Edges"])
    assert np.array_equal(e2f, g["E2F"])
    assert mean_len == pytest.approx(float(g["mean_len"]), rel=1e-14)
    assert len(hand.faces) == 2 * len(hand.vertices) - 4


def test_unit_tables():
    g = golden("unit_tables")
    wt = orc.refract_dir(torch.tensor(g["wo"]), torch.tensor(g["n"]), torch.tensor(g["eta"]))
    close(wt, g["refract_wt"])
    tir = orc.fresnel_tir(torch.tensor(g["fr_cos"]), torch.tensor(g["fr_etaI"]), torch.tensor(g["fr_etaT"]))
    assert np.array_equal(tir.numpy(), g["fr_tir"])
    u, v, t, n = orc.moller_trumbore(torch.tensor(g["mt_o"]), torch.tensor(g["mt_d"]), torch.tensor(g["mt_tri"]))
    close(u, g["mt_u"]); close(v, g["mt_v"]); close(t, g["mt_t"]); close(n, g["mt_n"])


@pytest.mark.parametrize("name", FIXTURES + BIG_FIXTURES)
def test_render_path(name):
    g = golden(name)
    hand = fixture_mesh(g)            # (the headline fixture: horse_vh x4, 50 248 triangles)
    o, d, sp, valid = fixture_view(g)
    V = torch.tensor(hand.vertices, dtype=torch.float64, requires_grad=True)
    mesh = orc.Mesh(hand.faces, V)
    out_ori, out_dir, mask, aux = orc.render_transparent(mesh, o, d, IOR, return_aux=True)
    b1, b2 = aux["b1"], aux["b2"]
    # bounce 1
    assert np.array_equal(aux["ind1"].numpy(), g["b1_ind"])
    assert np.array_equal(b1["face"].numpy(), g["b1_face"])
    close(b1["u"], g["b1_u"], rtol=1e-9); close(b1["v"], g["b1_v"], rtol=1e-9)
    close(b1["t"], g["b1_t"]); close(b1["n"], g["b1_n"])
    assert np.array_equal(b1["refracted"].numpy(), g["b1_refracted"])
    close(b1["new_o"], g["b1_new_o"]); close(b1["new_d"], g["b1_new_d"])
    # bounce 2 (rays that refracted at bounce 1)
    assert np.array_equal(aux["ind2"].numpy()[b2["hitted"].numpy()], g["b2_ind"])
    assert np.array_equal(b2["hitted"].numpy(), g["b2_hitted"])
    assert np.array_equal(b2["face"].numpy(), g["b2_face"])
    close(b2["t"], g["b2_t"]); close(b2["n"], g["b2_n"])
    assert np.array_equal(b2["refracted"].numpy(), g["b2_refracted"])
    close(b2["new_o"], g["b2_new_o"]); close(b2["new_d"], g["b2_new_d"])
    assert np.array_equal(aux["occluded"].numpy(), g["occluded"])
    # outputs
    vi = torch.nonzero(mask[:, 0]).squeeze(1)
    assert np.array_equal(vi.numpy(), g["valid_ind"])
    assert mask.dtype == torch.bool and mask.shape == o.shape
    assert torch.equal(mask[:, 0], mask[:, 1]) and torch.equal(mask[:, 0], mask[:, 2])
    close(out_ori[vi], g["out_ori"]); close(out_dir[vi], g["out_dir"])
    assert float(out_ori[~mask[:, 0]].abs().sum()) == 0.0 and float(out_dir[~mask[:, 0]].abs().sum()) == 0.0
    # loss and gradients
    loss = orc.ray_loss(out_ori, out_dir, mask, sp, valid)
    assert loss.item() == pytest.approx(float(g["ray_loss"]), rel=1e-12)
    g_ray, = torch.autograd.grad(loss, V, retain_graph=True)
    scale = np.abs(g["grad_ray_loss"]).max()
    close(g_ray, g["grad_ray_loss"], rtol=1e-9, atol=1e-11 * scale)
    rng = np.random.default_rng(int(g["lin_seed"]))
    w_ori = rng.standard_normal(o.shape); w_dir = rng.standard_normal(o.shape)
    lin = (out_ori * torch.tensor(w_ori)).sum() + (out_dir * torch.tensor(w_dir)).sum()
    assert lin.item() == pytest.approx(float(g["lin"]), rel=1e-11)
    g_lin, = torch.autograd.grad(lin, V)
    close(g_lin, g["grad_lin"], rtol=1e-9, atol=1e-11 * np.abs(g["grad_lin"]).max())


@pytest.mark.parametrize("name", FIXTURES + BIG_FIXTURES)
def test_silhouette_branch(name):
    g = golden(name)
    hand = fixture_mesh(g)
    if name in BIG_FIXTURES:      # (edge tables: mesh_io's, pinned to the reference's on the hand hull by test_topology_tables)
        e, e2f, _ = mesh_io.edge_tables(hand)
        topo = {"Edges": e, "E2F": e2f}
    else:
        topo = golden("hand_topology")
    res = int(g["res"])
    o, d, _, _ = fixture_view(g)
    V = torch.tensor(hand.vertices, dtype=torch.float64, requires_grad=True)
    mesh = orc.Mesh(hand.faces, V)
    Edges, E2F = torch.tensor(topo["Edges"]), torch.tensor(topo["E2F"])
    cam = tuple(torch.tensor(g[k]) for k in ("R", "K", "Rinv", "Kinv"))
    sil = orc.silhouette_edges(V, Edges, E2F, o[0])
    assert np.array_equal(sil.numpy(), g["sil_edges"])
    index, output = orc.primary_visibility(mesh, sil, cam, o[0], res, res, detach_depth=True)
    assert np.array_equal(index.numpy(), g["vh_index"])
    assert str(output.dtype) == str(g["vh_output_dtype"]) == "torch.float32"
    close(output, g["vh_output"])
    hit = np.zeros(res * res, dtype=np.uint8)
    hit[g["b1_ind"]] = 1          # the fixture stores the soft mask in float32; rebuild the float64 one
    soft = torch.tensor(views.process_mask(hit.reshape(res, res)), dtype=torch.float64).reshape(-1)
    vh = orc.vh_loss_view(mesh, Edges, E2F, cam, o[0], soft, res, res)
    assert vh.item() == pytest.approx(float(g["vh_loss"]), rel=1e-12)
    g_vh, = torch.autograd.grad(vh, V)
    close(g_vh, g["grad_vh"], rtol=1e-9, atol=1e-11 * np.abs(g["grad_vh"]).max())


def test_process_mask_matches_fixture(hand):
    g = golden("hand_r64_v5")
    hit = np.zeros(64 * 64, dtype=np.uint8)
    hit[g["b1_ind"]] = 1
    soft = views.process_mask(hit.reshape(64, 64))
    np.testing.assert_allclose(soft, g["soft_mask"], rtol=0, atol=1e-7)
    assert np.all(soft[-1] == 0.5) and soft.min() >= 0 and soft.max() <= 1


def test_dihedral_and_sm_loss(hand):
    g = golden("hand_smooth_sm")
    V = torch.tensor(g["vertices"].astype(np.float64), requires_grad=True)
    E2F = torch.tensor(g["E2F"])
    cosang = orc.dihedral_cos(V, E2F)
    close(cosang, g["dihedral_cos"])
    sm = orc.sm_loss(V, E2F)
    assert sm.item() == pytest.approx(float(g["sm_loss"]), rel=1e-12)
    g_sm, = torch.autograd.grad(sm, V)
    close(g_sm, g["grad_sm"], rtol=1e-9, atol=1e-11 * np.abs(g["grad_sm"]).max())


def test_raw_hull_sm_loss_is_infinite(hand):
    """some dihedral cosine of raw hand_vh is exactly -1 -> the reference always remeshes first."""
    _, e2f, _ = mesh_io.edge_tables(hand)
    sm = orc.sm_loss(torch.tensor(hand.vertices), torch.tensor(e2f))
    assert torch.isinf(sm) or torch.isnan(sm)


def test_two_optimisation_steps(hand):
    """all_loss weights + limit_hook + SGD(nesterov) over two iterations (reference optim.py:110-130, 155-171, 198-215)."""
    g = golden("hand_smooth_steps")
    topo = golden("hand_topology")
    res = int(g["res"])
    center, extent = views.mesh_frame(hand.vertices)
    cams = views.turntable_cameras(center, extent, 72, res, res)
    Vs = torch.tensor(g["vertices"].astype(np.float64))
    Edges, E2F = torch.tensor(topo["Edges"]), torch.tensor(topo["E2F"])
    base = orc.Mesh(hand.faces, Vs)

    def view(k):
        R, K, Rinv, Kinv = cams[k]
        o, d = views.generate_ray(res, res, Kinv, Rinv)
        rng = np.random.default_rng(100 + k)
        sp = rng.standard_normal((res * res, 3)) * 40.0 + np.asarray(center) + np.array([0.0, 0.0, 150.0])
        valid = rng.random(res * res) > 0.1
        _, hit = orc.intersect_ids(base, o, d)
        soft = torch.tensor(views.process_mask(hit.numpy().reshape(res, res)), dtype=torch.float64).reshape(-1)
        cam = tuple(torch.tensor(a, dtype=torch.float64) for a in (R, K, Rinv, Kinv))
        return o, d, torch.tensor(sp), torch.tensor(valid), soft, cam

    cache = {k: view(k) for k in list(g["ray_views"]) + list(g["sil_views"])}
    param = torch.zeros_like(Vs)
    buf = None
    for it in range(2):
        p = param.clone().requires_grad_(True)
        V = Vs + p
        mesh = orc.Mesh(hand.faces, V)
        o, d, sp, valid, _, _ = cache[int(g["ray_views"][it])]
        oo, od, mk = orc.render_transparent(mesh, o, d, float(g["ior"]))
        ray = orc.ray_loss(oo, od, mk, sp, valid)
        vh = 0
        for k in g["sil_views"]:
            o_k, _, _, _, soft, cam = cache[int(k)]
            vh = vh + orc.vh_loss_view(mesh, Edges, E2F, cam, o_k[0], soft, res, res)
        sm = orc.sm_loss(V, E2F)
        loss = orc.total_loss(ray, vh, sm, res, float(g["mean_len"]))
        assert loss.item() == pytest.approx(float(g[f"loss{it}"]), rel=1e-11)
        assert f"ray={ray:g} vh={vh:g} sm={sm:g}" == str(g[f"loss_str{it}"])
        grad, = torch.autograd.grad(loss, p)
        grad = orc.limit_grad(grad)
        close(grad, g[f"grad{it}"], rtol=1e-8, atol=1e-12)
        param, buf = orc.sgd_nesterov_step(param, grad, buf, float(g["lr"]), float(g["momentum"]))
        close(param, g[f"param{it}"], rtol=1e-8, atol=1e-13)


def test_tracer_edge_cases():
    # empty ray set, single triangle, miss encoding, t <= 0 rejected, tie -> lowest face id
    faces = np.array([[0, 1, 2]], dtype=np.int32)
    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    T, ID = orc.trace_closest(faces, verts, np.zeros((0, 6), np.float32))
    assert T.shape == (0,) and ID.shape == (0,)
    rays = np.array([[0.2, 0.2, 1, 0, 0, -1], [0.2, 0.2, -1, 0, 0, -1], [2, 2, 1, 0, 0, -1], [0.2, 0.2, 0, 0, 0, -1]], np.float32)
    T, ID = orc.trace_closest(faces, verts, rays)
    assert list(ID) == [0, -1, -1, -1] and T[0] == 1.0 and np.all(T[1:] == -1.0)
    faces2 = np.array([[0, 1, 2], [0, 1, 2]], dtype=np.int32)     # coincident duplicates: equal t
    T, ID = orc.trace_closest(faces2, verts, rays[:1])
    assert ID[0] == 0
    # un-normalised direction: t scales inversely (silhouette probes use such rays, DiffRender.py:222-224)
    T2, _ = orc.trace_closest(faces, verts, np.array([[0.2, 0.2, 1, 0, 0, -4]], np.float32))
    assert T2[0] == 0.25


def far_camera_rays(mesh, distance_factor, res=96, view=3):
    """Pinhole rays from a camera `distance_factor` extents away whose focal length grows with the distance (same image of the object)."""
    c, ext = views.mesh_frame(mesh.vertices)
    cam = views.turntable_cameras(c, ext, 8, res, res, distance_factor=distance_factor, focal_factor=1.1 * distance_factor / 2.5)[view]
    o, d = views.generate_ray(res, res, cam[3], cam[2])
    return np.concatenate([o.numpy(), d.numpy()], 1).astype(np.float32)


@pytest.mark.parametrize("distance_factor", [2.5, 20.0, 100.0, 1000.0])
def test_far_camera_keeps_its_hits(hand, distance_factor):
    """The hit-point condition must not cost a distant camera its legitimate hits: its tolerance grows with the distance between the
    ray's origin and the triangle (oracle/hit_point.h), as the float32 error of the reconstructed point does.  Against the four
    inequalities alone (no fifth condition): same hits, same faces, same t -- with the absolute margin of round 3 five of the 698 hits
    of the 1000-extent camera were dropped (hit points up to 2.2 margins outside their triangle's box)."""
    f32, v32 = hand.faces.astype(np.int32), hand.vertices.astype(np.float32)
    rays = far_camera_rays(hand, distance_factor)
    T, ID = orc.trace_closest(f32, v32, rays)
    T0, ID0 = orc.trace_closest(f32, v32, rays, mt_only=True)
    assert (ID >= 0).sum() > 600
    assert np.array_equal(ID, ID0) and np.array_equal(T, T0)
    Tb, IDb = orc.trace_closest(f32, v32, rays, bvh=True)
    assert np.array_equal(ID, IDb) and np.array_equal(T, Tb)


def test_stepwise_hit_terms_vs_golden(hand):
    """drt_amd.stepwise.hit_point_terms (what Scene.Dintersect computes per hit) on the CPU against the reference's
    own u, v, t, n of the first bounce."""
    from drt_amd import stepwise
    g = golden("hand_r128_v5")
    o, d, _, _ = fixture_view(g)
    ind, face = torch.tensor(g["b1_ind"]), torch.tensor(g["b1_face"])
    tri = torch.tensor(hand.vertices)[torch.tensor(hand.faces)[face]]
    u, v, t, n = stepwise.hit_point_terms(o[ind], d[ind], tri)
    # the fixture's normals are stored after refract_ray flipped the leaving ones: compare up to that sign
    sign = torch.sign((n * torch.tensor(g["b1_n"])).sum(1, keepdim=True))
    for got, key in ((u, "b1_u"), (v, "b1_v"), (t, "b1_t"), (n * sign, "b1_n")):
        np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-10, atol=1e-12)
    assert (u >= -1e-12).all() and (v >= -1e-12).all() and (u + v <= 1 + 1e-12).all() and (t > 0).all()


def test_oracle_bvh_tracer_equals_bruteforce(hand):
    """oracle/bvh_tracer.c (the cpu_baseline's second figure) gives bit-identical T and ID, including rays with zero
    direction components and rays from inside the mesh; and the golden face ids."""
    f32, v32 = hand.faces.astype(np.int32), hand.vertices.astype(np.float32)
    g = golden("hand_r128_v23")
    o, d, _, _ = fixture_view(g)
    cam = np.concatenate([o.numpy(), d.numpy()], 1).astype(np.float32)
    rng = np.random.default_rng(9)
    c, ext = views.mesh_frame(hand.vertices)
    extra = np.concatenate([rng.uniform(-0.6, 0.6, (4000, 3)) * ext + c, rng.standard_normal((4000, 3)) * rng.uniform(0.05, 20, (4000, 1))], 1).astype(np.float32)
    extra[:200, 3] = 0; extra[200:400, 4] = 0; extra[400:600, 5] = 0; extra[600:700, 3:5] = 0
    rays = np.concatenate([cam, extra])
    T0, I0 = orc.trace_closest(f32, v32, rays, bvh=False)
    T1, I1 = orc.trace_closest(f32, v32, rays, bvh=True)
    assert np.array_equal(I0, I1) and np.array_equal(T0, T1)
    hit = np.flatnonzero(I0[:len(cam)] >= 0)
    assert np.array_equal(hit, g["b1_ind"]) and np.array_equal(I1[hit], g["b1_face"])
    assert (I0[len(cam):] >= 0).mean() > 0.1
    Te, Ie = orc.trace_closest(np.zeros((0, 3), np.int32), v32, rays[:5], bvh=True)
    assert (Ie == -1).all() and (Te == -1).all()


def test_degenerate_faces_follow_the_reference(hand):
    """hand_degenerate.npz (a zero-length edge = two zero-area faces, the defect of monkey_vh.ply / dog_vh.ply): a zero-area
    face is never hit, its normal is NaN (reference DiffRender.py:103-104, 149-163), NaN reaches the smoothness loss and
    the vertex gradient, and limit_hook (optim.py:155-162) zeroes it before the SGD step."""
    g = golden("hand_degenerate")
    topo = golden("hand_topology")
    res = int(g["res"])
    Vs = torch.tensor(g["vertices"].astype(np.float64))
    Edges, E2F = torch.tensor(topo["Edges"]), torch.tensor(topo["E2F"])
    assert np.array_equal(g["Edges"], topo["Edges"])
    tri = Vs[torch.tensor(hand.faces)]
    area = torch.linalg.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).norm(dim=1)
    assert np.array_equal(np.flatnonzero(area.numpy() == 0), g["zero_area_faces"])
    center, extent = views.mesh_frame(hand.vertices)
    cams = views.turntable_cameras(center, extent, 72, res, res)
    o, d = views.generate_ray(res, res, g["Kinv"], g["Rinv"])
    rng = np.random.default_rng(int(g["target_seed"]))
    sp = torch.tensor(rng.standard_normal((res * res, 3)) * 40.0 + np.asarray(center) + np.array([0.0, 0.0, 150.0]))
    valid = torch.tensor(rng.random(res * res) > 0.1)
    V = Vs.clone().requires_grad_(True)
    mesh = orc.Mesh(hand.faces, V)
    oo, od, mk, aux = orc.render_transparent(mesh, o, d, float(g["ior"]), return_aux=True)
    assert np.array_equal(torch.nonzero(mk[:, 0]).squeeze(1).numpy(), g["valid_ind"])
    f1 = np.full(res * res, -1, np.int64); f1[g["b1_ind"]] = g["b1_face"]
    assert np.array_equal(aux["face1"].numpy(), f1) and not np.isin(g["zero_area_faces"], f1).any()
    close(od[mk[:, 0]], g["out_dir"]); close(oo[mk[:, 0]], g["out_ori"], atol=1e-9)
    ray = orc.ray_loss(oo, od, mk, sp, valid)
    assert ray.item() == pytest.approx(float(g["ray_loss"]), rel=1e-11)
    g_ray, = torch.autograd.grad(ray, V)
    close(g_ray, g["grad_ray_loss"], rtol=1e-8, atol=1e-11 * np.abs(g["grad_ray_loss"]).max())
    cam = tuple(torch.tensor(g[k]) for k in ("R", "K", "Rinv", "Kinv"))
    sil = orc.silhouette_edges(V, Edges, E2F, o[0])
    assert np.array_equal(sil.numpy(), g["sil_edges"])
    index, output = orc.primary_visibility(mesh, sil, cam, o[0], res, res, detach_depth=True)
    assert np.array_equal(index.numpy(), g["vh_index"])
    hit = np.zeros(res * res, dtype=np.uint8); hit[g["b1_ind"]] = 1
    soft = torch.tensor(views.process_mask(hit.reshape(res, res)), dtype=torch.float64).reshape(-1)
    vh = orc.vh_loss_view(mesh, Edges, E2F, cam, o[0], soft, res, res)
    assert vh.item() == pytest.approx(float(g["vh_loss"]), rel=1e-12)
    g_vh, = torch.autograd.grad(vh, V)
    np.testing.assert_allclose(g_vh.numpy(), g["grad_vh"], rtol=1e-9, atol=1e-11 * np.nanmax(np.abs(g["grad_vh"])), equal_nan=True)
    cosang = orc.dihedral_cos(V, E2F)
    assert np.isnan(g["dihedral_cos"]).sum() == 5
    np.testing.assert_allclose(cosang.detach().numpy(), g["dihedral_cos"], rtol=RTOL, atol=1e-11, equal_nan=True)
    sm = orc.sm_loss(V, E2F)
    assert np.isnan(sm.item()) and np.isnan(float(g["sm_loss"]))
    g_sm, = torch.autograd.grad(sm, V)
    assert np.array_equal(np.isnan(g_sm.numpy()), np.isnan(g["grad_sm"]))
    np.testing.assert_allclose(g_sm.numpy(), g["grad_sm"], rtol=1e-8, atol=1e-10 * np.nanmax(np.abs(g["grad_sm"])), equal_nan=True)
    # the whole iteration: NaN loss, finite clamped gradient, finite parameters
    base = orc.Mesh(hand.faces, Vs)
    p = torch.zeros_like(Vs).requires_grad_(True)
    Vp = Vs + p
    mesh = orc.Mesh(hand.faces, Vp)
    oo, od, mk = orc.render_transparent(mesh, o, d, float(g["ior"]))
    ray = orc.ray_loss(oo, od, mk, sp, valid)
    vh = 0
    for k in g["sil_views"]:
        R, K, Rinv, Kinv = cams[int(k)]
        ok, dk = views.generate_ray(res, res, Kinv, Rinv)
        _, hitk = orc.intersect_ids(base, ok, dk)
        softk = torch.tensor(views.process_mask(hitk.numpy().reshape(res, res)), dtype=torch.float64).reshape(-1)
        camk = tuple(torch.tensor(a, dtype=torch.float64) for a in (R, K, Rinv, Kinv))
        vh = vh + orc.vh_loss_view(mesh, Edges, E2F, camk, ok[0], softk, res, res)
    sm = orc.sm_loss(Vp, E2F)
    loss = orc.total_loss(ray, vh, sm, res, float(g["mean_len"]))
    assert np.isnan(loss.item()) and f"ray={ray:g} vh={vh:g} sm={sm:g}" == str(g["step_loss_str"])
    grad, = torch.autograd.grad(loss, p)
    grad = orc.limit_grad(grad)
    assert torch.isfinite(grad).all()
    close(grad, g["step_grad"], rtol=1e-8, atol=1e-12)
    param, _ = orc.sgd_nesterov_step(torch.zeros_like(Vs), grad, None, float(g["lr"]), float(g["momentum"]))
    close(param, g["step_param"], rtol=1e-8, atol=1e-13)


def test_headline_mesh_trajectory_first_iterations():
    """The first two iterations of the same recorded pass on the HEADLINE mesh (tests/golden/horse50k_trajectory.npz: 50 248 triangles) by
    the oracle (brute-force tracer: about a second per iteration here)."""
    import trajectory_case as tc
    from drt_amd import mesh_io
    g = tc.load("horse50k_trajectory")
    hull = tc.frame_mesh("horse50k_trajectory")
    faces = hull.faces
    assert np.array_equal(np.asarray(g["faces"]), np.asarray(faces, dtype=np.int32))
    data = tc.RecordedCapture(g, hull.vertices, "cpu")
    res = data.resx
    Vs = torch.tensor(g["vertices"].astype(np.float64))
    Edges, E2F, _ = (torch.tensor(np.asarray(a)) if not np.isscalar(a) else a for a in mesh_io.edge_tables(mesh_io.TriMesh(Vs.numpy(), faces)))
    ray_view, silh_view = data.ray_view_generator(), data.silh_view_generator()
    param, buf = torch.zeros_like(Vs), None
    for it in range(2):
        p = param.clone().requires_grad_(True)
        V = Vs + p
        mesh = orc.Mesh(faces, V)
        sp, valid, _, o, d, _ = data.get_view(next(ray_view))
        oo, od, mk = orc.render_transparent(mesh, o, d, float(g["ior"]))
        ray = orc.ray_loss(oo, od, mk, sp, valid)
        vh = 0
        for _ in range(8):
            _, _, soft, o_k, _, cam = data.get_view(next(silh_view))
            vh = vh + orc.vh_loss_view(mesh, Edges, E2F, cam, o_k[0], soft, res, res)
        sm = orc.sm_loss(V, E2F)
        loss = orc.total_loss(ray, vh, sm, res, float(g["mean_len"]))
        assert loss.item() == pytest.approx(float(g["loss"][it]), rel=1e-9), it
        assert f"ray={ray:g} vh={vh:g} sm={sm:g}" == str(g["loss_str"][it]), it
        grad, = torch.autograd.grad(loss, p)
        grad = orc.limit_grad(grad)
        assert grad.abs().max().item() == pytest.approx(float(g["gmax"][it]), rel=1e-7)
        param, buf = orc.sgd_nesterov_step(param, grad, buf, float(g["lr"]), float(g["momentum"]))


def test_long_trajectory_first_iterations(hand):
    """The first 20 iterations of the recorded pass of the reference's loop (tests/golden/hand_trajectory.npz: optim.py:190-215 with the
    reference's own Loss_calculator, 1 stochastic refraction view + 8 silhouette views + smoothness per iteration, limit_hook, nesterov)
    replayed by the oracle: losses of every iteration and the parameter checkpoints at 10 and 20.  (The HIP path replays all 60 on the
    GPU, tests/test_gpu_trajectory.py; the oracle is the checker there and is itself pinned here.)"""
    import trajectory_case as tc
    g = tc.load()
    topo = golden("hand_topology")
    data = tc.RecordedCapture(g, hand.vertices, "cpu")
    res = data.resx
    Vs = torch.tensor(g["vertices"].astype(np.float64))
    Edges, E2F = torch.tensor(topo["Edges"]), torch.tensor(topo["E2F"])
    ray_view, silh_view = data.ray_view_generator(), data.silh_view_generator()
    param, buf = torch.zeros_like(Vs), None
    checkpoints = {int(it): p for it, p in zip(g["param_its"], g["params"])}
    for it in range(20):
        p = param.clone().requires_grad_(True)
        V = Vs + p
        mesh = orc.Mesh(hand.faces, V)
        sp, valid, _, o, d, _ = data.get_view(next(ray_view))
        oo, od, mk = orc.render_transparent(mesh, o, d, float(g["ior"]))
        ray = orc.ray_loss(oo, od, mk, sp, valid)
        vh = 0
        for _ in range(8):
            _, _, soft, o_k, _, cam = data.get_view(next(silh_view))
            vh = vh + orc.vh_loss_view(mesh, Edges, E2F, cam, o_k[0], soft, res, res)
        sm = orc.sm_loss(V, E2F)
        loss = orc.total_loss(ray, vh, sm, res, float(g["mean_len"]))
        assert loss.item() == pytest.approx(float(g["loss"][it]), rel=1e-9), it
        assert f"ray={ray:g} vh={vh:g} sm={sm:g}" == str(g["loss_str"][it]), it
        grad, = torch.autograd.grad(loss, p)
        grad = orc.limit_grad(grad)
        assert grad.abs().max().item() == pytest.approx(float(g["gmax"][it]), rel=1e-7)
        param, buf = orc.sgd_nesterov_step(param, grad, buf, float(g["lr"]), float(g["momentum"]))
        if it + 1 in checkpoints:
            close(param, checkpoints[it + 1], rtol=1e-7, atol=1e-10)
