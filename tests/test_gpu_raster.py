"""Projected primary visibility (drt_amd/csrc/drt_raster.h): the primary face ids of render_transparent must equal the
exhaustive float32 test (the tracer contract, oracle/tracer.c) whether the rays are a pinhole grid (decided by projecting
the triangles), a perturbed grid (every ray falls back to the tree), a mix, a camera inside the object (no projection
bound: whole image falls back) or a low-polygon close-up (triangles that cover thousands of pixels)."""
import numpy as np
import pytest
import torch

from conftest import IOR, data_path
from drt_amd import mesh_io, views

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Render():
    from drt_amd import diffrender
    diffrender.intIOR = IOR
    return diffrender


def _primary_reference(scene, o, d):
    rays = torch.cat([o.float(), d.float()], 1).contiguous()
    T, ID = scene.optix_mesh.intersect_bruteforce(rays)
    return ID


def _check(Render, scene, o, d, res_x, res_y, expect_stage_items=None):
    Render.resx, Render.resy = res_x, res_y
    tr = scene.optix_mesh
    tr.profile_enable(1)
    tr.profile_read()
    with torch.no_grad():
        oo, od, mk = scene.render_transparent(o, d)
    prof = tr.profile_read()
    tr.profile_enable(0)
    f1 = scene.last_face1.clone()
    assert torch.equal(f1, _primary_reference(scene, o, d))
    # and the whole path equals the run without the hint (linear order, tree for every ray)
    Render.resx, Render.resy = 7, 7
    with torch.no_grad():
        oo2, od2, mk2 = scene.render_transparent(o, d)
    assert torch.equal(oo, oo2) and torch.equal(od, od2) and torch.equal(mk, mk2)
    assert torch.equal(scene.last_face1, f1)
    return prof, f1


def test_grid_rays_are_decided_by_projection_and_equal_bruteforce(Render):
    mesh = mesh_io.subdivide_midpoint(mesh_io.read_ply(data_path("horse_vh.ply")))
    scene = Render.Scene(mesh, 0)
    c, ext = views.mesh_frame(mesh.vertices)
    res = 512
    cams = views.turntable_cameras(c, ext, 72, res, res)
    rays = [views.generate_ray(res, res, cams[k][3], cams[k][2], device="cuda") for k in (0, 13, 29, 50)]
    o = torch.cat([r[0] for r in rays]).contiguous(); d = torch.cat([r[1] for r in rays]).contiguous()
    prof, f1 = _check(Render, scene, o, d, res, res)
    assert prof["raster"][1] >= 1 and prof["raster"][2] == len(o)      # the projection pass ran over all four images
    assert prof["trace1"][2] == 0                                       # and no ray needed the tree
    assert 0.01 < (f1 >= 0).float().mean().item() < 0.6
    # a non-square image (the captures are 960x1280 and 1080x1920) and 144 rows of padding-free rectangles
    cams = views.turntable_cameras(c, ext, 72, 320, 192)
    o2, d2 = views.generate_ray(192, 320, cams[7][3], cams[7][2], device="cuda")
    prof, _ = _check(Render, scene, o2, d2, 320, 192)
    assert prof["raster"][2] == len(o2) and prof["trace1"][2] == 0
    # a height that is not a multiple of the 16-row tiles of the projection pass (its last tile row is cut), two images
    cams = views.turntable_cameras(c, ext, 72, 128, 200)
    rays = [views.generate_ray(200, 128, cams[k][3], cams[k][2], device="cuda") for k in (3, 40)]
    o3 = torch.cat([r[0] for r in rays]).contiguous(); d3 = torch.cat([r[1] for r in rays]).contiguous()
    prof, f3 = _check(Render, scene, o3, d3, 128, 200)
    assert prof["raster"][2] == len(o3) and prof["trace1"][2] == 0 and (f3 >= 0).any()


def test_rays_outside_the_grid_fall_back_to_the_tree(Render):
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    scene = Render.Scene(mesh, 0)
    c, ext = views.mesh_frame(mesh.vertices)
    res = 256
    cam = views.turntable_cameras(c, ext, 72, res, res)[31]
    o, d = views.generate_ray(res, res, cam[3], cam[2], device="cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    # (a) a calibrated-camera-like perturbation of every direction (2 pixels of smooth distortion): the four corner rays
    #     still fit a model, the lattice check refuses it, every ray takes the tree
    yy, xx = torch.meshgrid(torch.arange(res, device="cuda", dtype=torch.float64), torch.arange(res, device="cuda", dtype=torch.float64), indexing="ij")
    r2 = ((xx - res / 2) ** 2 + (yy - res / 2) ** 2).reshape(-1, 1) / (res * res)
    d_dist = d + 0.01 * r2 * (1 - 4 * r2) * torch.linalg.cross(d, torch.tensor([[0.0, 1.0, 0.0]], device="cuda", dtype=torch.float64).expand_as(d))
    d_dist = d_dist / d_dist.norm(dim=1, keepdim=True)
    prof, f1 = _check(Render, scene, o, d_dist.contiguous(), res, res)
    assert prof["trace1"][2] > 0 and (f1 >= 0).any()
    # (b) a grid in which a sprinkling of rays is replaced (other origin / other direction): those rays, and only
    #     those, take the tree
    o_mix, d_mix = o.clone(), d.clone()
    lattice = torch.zeros(res, res, dtype=torch.bool, device="cuda")      # the 8x8 sample rays of the model check stay grid rays
    ls = ((res - 1) * torch.arange(8, device="cuda")) // 7
    lattice[ls[:, None], ls[None, :]] = True
    free = ~lattice.reshape(-1)
    pick = (torch.rand(len(o), device="cuda", generator=g) < 0.03) & free
    d_mix[pick] = d[torch.randperm(len(o), device="cuda", generator=g)][pick]
    pick2 = (torch.rand(len(o), device="cuda", generator=g) < 0.01) & free
    o_mix[pick2] = o_mix[pick2] + 3.0
    prof, f1 = _check(Render, scene, o_mix, d_mix, res, res)
    assert 0 < prof["trace1"][2] <= int((pick | pick2).sum())
    # ... and when one of the sampled rays is off, the whole image takes the tree
    d_mix[0] = d[777]
    prof, f1 = _check(Render, scene, o_mix, d_mix, res, res)
    assert prof["trace1"][2] > int((pick | pick2).sum())
    # (c) rays that are no image at all (random order): the model of the "image" does not fit
    perm = torch.randperm(len(o), device="cuda", generator=g)
    prof, _ = _check(Render, scene, o[perm].contiguous(), d[perm].contiguous(), res, res)
    assert prof["trace1"][2] > 0
    # (d) a wrong hint (width x height swapped on a non-square image) is harmless
    cam = views.turntable_cameras(c, ext, 72, 128, 64)[5]
    o3, d3 = views.generate_ray(64, 128, cam[3], cam[2], device="cuda")
    _check(Render, scene, o3, d3, 64, 128)


def test_camera_inside_the_object_and_huge_triangles(Render):
    # camera inside a coarse sphere: triangles behind / through the camera plane have no projection bound
    sphere = mesh_io.icosphere(1, radius=50.0)
    scene = Render.Scene(sphere, 0)
    res = 128
    K = np.array([[0.6 * res, 0, res / 2], [0, 0.6 * res, res / 2], [0, 0, 1.0]])
    Rinv = np.eye(4); Rinv[:3, 3] = [3.0, -2.0, 5.0]
    o, d = views.generate_ray(res, res, np.linalg.inv(K), Rinv, device="cuda")
    prof, f1 = _check(Render, scene, o, d, res, res)
    assert (f1 >= 0).all() and prof["trace1"][2] > 0                   # every ray hits from inside; all took the tree
    # low-polygon close-up from outside: 80 triangles, each covering thousands of pixels (the one-block-per-triangle kernel)
    c, ext = views.mesh_frame(sphere.vertices)
    cam = views.turntable_cameras(c, ext, 72, 256, 256, distance_factor=1.2)[3]
    o, d = views.generate_ray(256, 256, cam[3], cam[2], device="cuda")
    prof, f1 = _check(Render, scene, o, d, 256, 256)
    assert prof["trace1"][2] == 0 and (f1 >= 0).float().mean().item() > 0.3
    # the hand filling a 768^2 image: a third of its triangles cover 50-300 pixel centres (one wave per triangle), three views in one call
    hand = mesh_io.read_ply(data_path("hand_vh.ply"))
    hscene = Render.Scene(hand, 0)
    c, ext = views.mesh_frame(hand.vertices)
    cams = views.turntable_cameras(c, ext, 72, 768, 768, distance_factor=1.1)
    rays = [views.generate_ray(768, 768, cams[k][3], cams[k][2], device="cuda") for k in (2, 31, 55)]
    prof, f1 = _check(Render, hscene, torch.cat([r[0] for r in rays]), torch.cat([r[1] for r in rays]), 768, 768)
    assert prof["trace1"][2] == 0 and (f1 >= 0).float().mean().item() > 0.15
    # a mesh of two triangles, one of them degenerate, and an empty mesh
    flat = mesh_io.TriMesh(np.array([[-30.0, -30, 0], [30, -30, 0], [0, 40, 0]]), np.array([[0, 1, 2], [0, 0, 0]]))
    from drt_amd.optix_mesh import optix_mesh
    t = optix_mesh(0)
    t.update_mesh(torch.tensor(flat.faces, dtype=torch.int32, device="cuda"), torch.tensor(flat.vertices, dtype=torch.float32, device="cuda"))
    T, ID = t.intersect(torch.cat([o.float(), d.float()], 1))
    assert torch.equal(ID, t.intersect_bruteforce(torch.cat([o.float(), d.float()], 1))[1])


def test_key_buffer_is_clean_after_every_call(Render):
    """The per-ray key buffer is reset by the kernel that consumes it: alternating meshes and views on one scene must
    never see a stale key of an earlier call."""
    hand = mesh_io.read_ply(data_path("hand_vh.ply"))
    scene = Render.Scene(hand, 0)
    c, ext = views.mesh_frame(hand.vertices)
    res = 256
    cams = views.turntable_cameras(c, ext, 72, res, res)
    V0 = scene.vertices.detach().clone()
    for it in range(6):
        scene.update_verticex(V0 * (1.0 + 0.07 * (it % 3)) + (it % 2) * 9.0)
        o, d = views.generate_ray(res, res, cams[11 * it][3], cams[11 * it][2], device="cuda")
        Render.resx = Render.resy = res
        with torch.no_grad():
            if it % 2:
                sp = torch.randn(o.shape, dtype=torch.float64, device="cuda")
                valid = torch.rand(len(o), device="cuda") < 0.5
                scene.ray_loss_fused(o, d, sp, valid)               # the fused pass consumes the keys of untargeted pixels too
            else:
                scene.render_transparent(o, d)
                assert torch.equal(scene.last_face1, _primary_reference(scene, o, d))
    scene.update_verticex(V0)
    o, d = views.generate_ray(res, res, cams[40][3], cams[40][2], device="cuda")
    Render.resx = Render.resy = res
    with torch.no_grad():
        scene.render_transparent(o, d)
    assert torch.equal(scene.last_face1, _primary_reference(scene, o, d))


def test_grid_verdict_is_cached_per_ray_tensors_and_invalidated_by_inplace_writes(Render):
    """Scene.render_transparent on the SAME ray tensors again trusts the recorded verdict (no re-verification, rays of
    untouched pixels not even read); results stay equal to the exhaustive test while the mesh moves, and an in-place write
    to either tensor (version counter) makes the next call establish again -- including one that breaks the grid."""
    from drt_amd import diffrender
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    scene = Render.Scene(mesh, 0)
    c, ext = views.mesh_frame(mesh.vertices)
    res = 256
    cams = views.turntable_cameras(c, ext, 72, res, res)
    rays = [views.generate_ray(res, res, cams[k][3], cams[k][2], device="cuda") for k in (4, 22, 47)]
    o = torch.cat([r[0] for r in rays]).contiguous(); d = torch.cat([r[1] for r in rays]).contiguous()
    # image 1 is a distorted grid: it must never be trusted, the other two are
    P = res * res
    d[P:2 * P] = torch.nn.functional.normalize(d[P:2 * P] + 0.003 * torch.sin(torch.arange(P, device="cuda", dtype=torch.float64) / 977.0).unsqueeze(1), dim=1)
    Render.resx = Render.resy = res
    V0 = scene.vertices.detach().clone()
    tr = scene.optix_mesh
    sp = torch.randn(o.shape, dtype=torch.float64, device="cuda"); valid = torch.rand(len(o), device="cuda") < 0.6

    def run(expect_mode):
        if expect_mode is not None:     # (peeking creates the record: only where the test means to)
            assert (diffrender._grid_cache(o, d, len(o), res, res)[0] & 3) == expect_mode
        tr.profile_enable(1); tr.profile_read()
        with torch.no_grad():
            oo, od, mk = scene.render_transparent(o, d)
        prof = tr.profile_read(); tr.profile_enable(0)
        assert torch.equal(scene.last_face1, _primary_reference(scene, o, d))
        return prof, (oo, od, mk)

    # (peeking at the cache state creates it: drop the attribute so that the first render establishes)
    for it in range(4):
        scene.update_verticex(V0 * (1.0 + 0.05 * it) + 2.0 * it)
        if it == 0 and hasattr(d, "_drt_grid"):
            del d._drt_grid
        prof, out = run(None)
        assert prof["trace1"][2] > 0                       # the distorted image takes the tree, every time
        assert prof["trace1"][2] < 0.6 * prof["shade1"][2]  # ... and the two grid images do not
        if it > 0:
            assert (diffrender._grid_cache(o, d, len(o), res, res)[0] & 3) == 2
        # the fused loss on the same tensors shares the verdict
        V = scene.vertices.detach().clone().requires_grad_(True)
        scene.update_verticex(V)
        lf = scene.ray_loss_fused(o, d, sp, valid)
        oo, od, mk = scene.render_transparent(o, d)
        l2 = Render.ray_loss(oo, od, mk, sp, valid)
        assert lf.item() == pytest.approx(l2.item(), rel=1e-12, abs=1e-300)
    # the same values through fresh tensor objects (no cache) and with the cache disabled: identical outputs
    ref = run(None)[1]
    diffrender.GRID_CACHE = False
    try:
        with torch.no_grad():
            again = scene.render_transparent(o, d)
    finally:
        diffrender.GRID_CACHE = True
    with torch.no_grad():
        fresh = scene.render_transparent(o.clone(), d.clone())
    for a, b, c2 in zip(ref, again, fresh):
        assert torch.equal(a, b) and torch.equal(a, c2)
    # in-place writes: image 0 loses a few grid rays (and its trust), image 2 keeps it
    ver = d._version
    d[1000:1010] = d[5000:5010].clone()
    assert d._version > ver and (diffrender._grid_cache(o, d, len(o), res, res)[0] & 3) == 1
    del d._drt_grid
    prof, _ = run(None)
    prof2, _ = run(None)                                   # trusted call after the re-establishment
    # (the establishing call sends image 0's ten foreign rays to the tree; so does a trusting call: an image that is not recorded as
    # all-verified is fitted and verified ray by ray again -- k_check_views -- instead of going to the tree wholesale)
    assert prof2["trace1"][2] == prof["trace1"][2] > 0
    o[2 * P + 5] += 1.0                                    # and an origin write
    prof3, _ = run(None)                                   # establishes again: per-ray verdicts, one more ray for the tree
    assert prof3["trace1"][2] >= prof["trace1"][2]


def test_small_calls_through_the_one_kernel_path_too():
    """DRT_MEGA_MAX_LOG2=n sends sub-batches of at most 2^n camera rays through k_path -- bounce #1, then ONE persistent kernel for the second
    traversal, bounce #2, the occlusion test and the valid-ray list (an option: measured no faster than the staged kernels, DESIGN.md
    section 6).  It must give the reference's results on the golden, oracle, raster-fallback and optimisation-step tests: this re-runs
    those files with every call of the suite taking that path."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, DRT_MEGA_MAX_LOG2="24")
    sel = ("not test_many_sub_batches and not test_small_calls_through and not test_stack_overflow_paths and not test_lds_stack_invariants "
           "and not test_full_loop_with_remesh and not test_optimize_loop and not test_closest_point and not test_large_mesh and not test_b1_ "
           "and not test_profile_select")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_raster.py"), os.path.join(here, "test_gpu_parity.py"),
                        "-m", "gpu", "-x", "-q", "-k", sel], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def _trusted_pair(Render, scene, mesh, res, cam_ids):
    """Ray tensors of `cam_ids` on which two calls have run: the second one has read the verdicts back (DRT_GRID_ALL_VERIFIED)."""
    c, ext = views.mesh_frame(mesh.vertices)
    cams = views.turntable_cameras(c, ext, 72, res, res)
    rays = [views.generate_ray(res, res, cams[k][3], cams[k][2], device="cuda") for k in cam_ids]
    o = torch.cat([r[0] for r in rays]).contiguous(); d = torch.cat([r[1] for r in rays]).contiguous()
    Render.resx = Render.resy = res
    with torch.no_grad():
        for _ in range(2):
            scene.render_transparent(o, d)
    assert d._drt_grid[3][0] is True                       # every image verified in every ray: later calls pass DRT_GRID_ALL_VERIFIED
    return o, d, cams


def test_all_verified_images_can_be_demoted_inside_a_call(Render):
    """DRT_GRID_TRUST | DRT_GRID_ALL_VERIFIED skips the launches that serve rays outside the grid and waits for the tree only in front
    of the second traversal.  The MESH can still take an image's projection bound away in a later call (a vertex reaches the camera
    plane: k_raster demotes the image): its rays must then get their primary hits from the tree (k_gen_late), not an all-miss image."""
    from drt_amd import diffrender
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    scene = Render.Scene(mesh, 0)
    res = 256
    o, d, cams = _trusted_pair(Render, scene, mesh, res, (4, 22, 47))
    V0 = scene.vertices.detach().clone()
    c = V0.mean(dim=0, keepdim=True)
    tr = scene.optix_mesh
    sp = torch.randn(o.shape, dtype=torch.float64, device="cuda") * 40.0; valid = torch.rand(len(o), device="cuda") < 0.6
    for scale, expect_tree in ((1.0, False), (7.0, True), (1.02, False), (7.5, True)):
        scene.update_verticex((V0 - c) * scale + c)        # x7 about the centroid: the cameras (2.5 extents away) are inside the hull
        assert (diffrender._grid_cache(o, d, len(o), res, res)[0] & (3 | 32)) == (2 | 32)
        tr.profile_enable(1); tr.profile_read()
        prof, f1 = _check(Render, scene, o, d, res, res)    # face ids == exhaustive test, outputs == the run without any hint
        Render.resx = Render.resy = res
        assert (prof["trace1"][2] > 0) == expect_tree
        assert (f1 >= 0).any()
        # the one-pass loss on the same tensors takes the same fallback
        Vg = scene.vertices.detach().clone().requires_grad_(True)
        scene.update_verticex(Vg)
        lf = scene.ray_loss_fused(o, d, sp, valid)
        gf, = torch.autograd.grad(lf, Vg)
        oo, od, mk = scene.render_transparent(o, d)
        l2 = Render.ray_loss(oo, od, mk, sp, valid)
        g2, = torch.autograd.grad(l2, Vg)
        assert lf.item() == pytest.approx(l2.item(), rel=1e-12, abs=1e-300)
        assert torch.allclose(gf, g2, rtol=1e-9, atol=1e-12 * max(1e-300, g2.abs().max().item()))


def test_trust_rechecks_the_lattice_of_every_image(Render):
    """A write through `.data` (or a raw pointer) changes rays without bumping the version counter the verdict cache is keyed on.  A
    trusting call re-verifies the 64 lattice rays of every image against the recorded model: wholesale-replaced rays are caught and
    the image is fitted and verified per ray again -- the answer is that of the NEW rays."""
    from drt_amd import diffrender
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    scene = Render.Scene(mesh, 0)
    res = 256
    o, d, cams = _trusted_pair(Render, scene, mesh, res, (4, 22, 47))
    P = res * res
    o_b, d_b = views.generate_ray(res, res, cams[60][3], cams[60][2], device="cuda")
    ver = (o._version, d._version)
    d.data[P:2 * P].copy_(d_b)                              # image 1: another camera's directions ...
    o.data[P:2 * P].copy_(o_b)                              # ... and origin
    d.data[2 * P:3 * P].copy_(torch.nn.functional.normalize(d_b + 0.01 * torch.sin(torch.arange(P, device="cuda", dtype=torch.float64) / 313.0).unsqueeze(1), dim=1))   # image 2: no grid at all
    assert (o._version, d._version) == ver and (diffrender._grid_cache(o, d, len(o), res, res)[0] & 3) == 2
    for _ in range(2):                                      # the call that notices, and the one after it (cache entry now untrusted)
        prof, f1 = _check(Render, scene, o, d, res, res)
        Render.resx = Render.resy = res
        assert prof["trace1"][2] > 0                        # image 2 goes through the tree
        assert (f1[P:2 * P] >= 0).any()
    # an untouched image is still served by the projection pass alone
    with torch.no_grad():
        oo, od, mk = scene.render_transparent(o[:P].clone(), d[:P].clone())
        ref = (oo, od, mk)
        oo, od, mk = scene.render_transparent(o, d)
    assert torch.equal(oo[:P], ref[0]) and torch.equal(od[:P], ref[1]) and torch.equal(mk[:P], ref[2])


def test_sparse_face_ids_leave_results_unchanged(Render):
    """DENSE_FACE_IDS = False (the production setting): face ids are defined for the rays with mask = 1 only; outputs, loss and
    gradient do not change."""
    from drt_amd import diffrender
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    scene = Render.Scene(mesh, 0)
    c, ext = views.mesh_frame(mesh.vertices)
    res = 256
    cam = views.turntable_cameras(c, ext, 72, res, res)[9]
    o, d = views.generate_ray(res, res, cam[3], cam[2], device="cuda")
    Render.resx = Render.resy = res
    sp = torch.randn(o.shape, dtype=torch.float64, device="cuda") * 50; valid = torch.rand(len(o), device="cuda") < 0.7
    outs = []
    for dense in (True, False, False):               # (twice: the second sparse call runs in TRUST mode)
        diffrender.DENSE_FACE_IDS = dense
        V = scene.vertices.detach().clone().requires_grad_(True)
        scene.update_verticex(V)
        oo, od, mk = scene.render_transparent(o, d)
        loss = Render.ray_loss(oo, od, mk, sp, valid)
        loss.backward()
        outs.append((oo.detach(), od.detach(), mk, loss.item(), V.grad.clone(), scene.last_face1.clone(), scene.last_face2.clone()))
    diffrender.DENSE_FACE_IDS = True
    a = outs[0]
    for b in outs[1:]:
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        assert b[3] == pytest.approx(a[3], rel=1e-12) and torch.allclose(a[4], b[4], rtol=1e-11, atol=1e-14 * a[4].abs().max().item())
        m = a[2][:, 0]
        assert torch.equal(a[5][m], b[5][m]) and torch.equal(a[6][m], b[6][m])


@pytest.mark.parametrize("extra", [{}, {"DRT_FILL_OVERLAP": "0"}, {"DRT_MEGA_MAX_LOG2": "24"}, {"DRT_MEGA_MAX_LOG2": "24", "DRT_FILL_OVERLAP": "0"}],
                         ids=["late-fills", "front-fills", "one-kernel-path-late-fills", "one-kernel-path-front-fills"])
def test_many_sub_batches_on_two_streams(extra):
    """Force every call to be cut into sub-batches of one or two images dealt to the two internal streams (the benchmark's 72 x 1024^2
    calls are cut like that; the tests' calls are normally one sub-batch): the projection pass, the verdict cache (indexed by global
    image), the key buffers and the lists of each stream must give the same results.  Re-runs this file and the multi-image tests
    of the other files in a subprocess with the sub-batch knobs set -- with the output fills issued beside the traversal (default: every
    sub-batch's memsets go through the build stream, one after the other) and in front of the projection pass; through the staged kernels
    (the default) and through the one-kernel path (k_path, DRT_MEGA_MAX_LOG2)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, DRT_MIN_SUB_LOG2="16", DRT_CHUNK_LOG2="18", DRT_SUB_PER_STREAM="8", **extra)
    sel = ("test_grid_rays_are_decided or test_rays_outside_the_grid or test_grid_verdict_is_cached or test_key_buffer_is_clean or test_all_verified_images or test_trust_rechecks or "
           "test_properties_at_full_size or test_render_transparent_vs_golden or test_two_optimisation_steps")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_raster.py"), os.path.join(here, "test_gpu_parity.py"),
                        "-m", "gpu", "-x", "-q", "-k", sel], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_axis_aligned_box_seen_along_its_faces(Render):
    """A box with axis-aligned faces and cameras that sit IN the planes of its faces and look along the axes: whole image rows and columns
    of rays run inside a face's plane (direction component exactly zero, origin exactly on the plane) -- the rays for which a float32
    triangle test alone does not define a closest hit (drt_tri.h's hit-point condition does).  Projection pass, tree and exhaustive test
    must agree on every pixel, and the whole path must equal the oracle's."""
    from oracle import diffrender_oracle as orc
    c = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], np.float64) * [40.0, 30.0, 20.0]
    f = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [1, 2, 6], [1, 6, 5], [2, 3, 7], [2, 7, 6], [3, 0, 4], [3, 4, 7]])
    box = mesh_io.TriMesh(c, f)
    for _ in range(3):
        box = mesh_io.subdivide_midpoint(box)                       # 768 faces on exactly representable coordinates
    scene = Render.Scene(box, 0)
    res = 128
    K = np.array([[1.2 * res, 0, res / 2], [0, 1.2 * res, res / 2], [0, 0, 1.0]])        # principal point ON a pixel centre: one row and one column with a zero component
    Kinv = np.linalg.inv(K)
    total_hits = 0
    for eye, fwd, up in (([-200.0, 30.0, 20.0], [1, 0, 0], [0, 0, 1]),     # in the planes y = +30 and z = +20
                         ([40.0, -150.0, -20.0], [0, 1, 0], [0, 0, 1]),     # in the planes x = +40 and z = -20
                         ([0.0, 0.0, 160.0], [0, 0, -1], [0, 1, 0]),        # on two symmetry planes: rays along the grid lines of the top face
                         ([40.0, 30.0, 100.0], [0, 0, -1], [1, 0, 0])):     # straight above a corner
        fwd, up = np.array(fwd, float), np.array(up, float)
        right = np.cross(fwd, up)
        Rinv = np.eye(4); Rinv[:3, 0], Rinv[:3, 1], Rinv[:3, 2], Rinv[:3, 3] = right, up, fwd, eye
        o, d = views.generate_ray(res, res, Kinv, Rinv, device="cuda")
        prof, f1 = _check(Render, scene, o, d, res, res)
        total_hits += int((f1 >= 0).sum())
        with torch.no_grad():
            oo, od, mk = scene.render_transparent(o, d)
        ro, rd, rm = orc.render_transparent(orc.Mesh(box.faces, torch.tensor(box.vertices)), o.cpu(), d.cpu(), IOR)
        assert torch.equal(mk.cpu(), rm)
        np.testing.assert_allclose(oo.cpu().numpy(), ro.numpy(), rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(od.cpu().numpy(), rd.numpy(), rtol=1e-10, atol=1e-11)
    assert total_hits > 1000


@pytest.mark.parametrize("seed", range(6))
def test_random_image_sizes_and_view_counts(Render, seed):
    """Image sizes that are no multiple of anything (the projection pass wants whole 64 x 4 patches: other sizes take the tree, or the
    patch path with cut tiles), one to four images per call, hints that match and hints that do not: always the exhaustive test's ids."""
    rng = np.random.default_rng(300 + seed)
    hand = mesh_io.read_ply(data_path("hand_vh.ply"))
    scene = Render.Scene(hand, 0)
    c, ext = views.mesh_frame(hand.vertices)
    for _ in range(4):
        w = int(rng.choice([rng.integers(5, 200), 64 * rng.integers(1, 4), 4 * rng.integers(2, 50)]))
        h = int(rng.choice([rng.integers(5, 200), 4 * rng.integers(2, 50), 16 * rng.integers(1, 12)]))
        nv = int(rng.integers(1, 5))
        cams = views.turntable_cameras(c, ext, 72, w, h, distance_factor=float(rng.choice([2.5, 1.3])))
        rays = [views.generate_ray(h, w, cams[k][3], cams[k][2], device="cuda") for k in rng.integers(0, 72, nv)]
        o = torch.cat([r[0] for r in rays]).contiguous(); d = torch.cat([r[1] for r in rays]).contiguous()
        _check(Render, scene, o, d, w, h)
        if w != h:                                       # (the same call again would be a trusting one: face ids of dead rays are then not written)
            _check(Render, scene, o, d, h, w)            # a wrong hint


@pytest.mark.parametrize("seed", range(8))
def test_random_intrinsics_and_poses(Render, seed):
    """Pinhole models nobody would build on purpose: unequal and negative focal lengths (mirrored images), skew, a principal point far off
    the image, fields of view from 2 to 150 degrees, cameras rolled about their axis, a few extents away, skimming the surface or inside
    the object, looking away from it; a fine mesh and one whose triangles cover a fifth of the image.  The fitted model either verifies --
    then the projection pass must find the exhaustive test's ids -- or it does not, and the tree answers."""
    rng = np.random.default_rng(500 + seed)
    mesh = mesh_io.read_ply(data_path("hand_vh.ply")) if seed % 2 else mesh_io.icosphere(int(rng.integers(0, 3)), radius=50.0, noise=0.1, seed=seed)
    scene = Render.Scene(mesh, 0)
    c, ext = views.mesh_frame(mesh.vertices)
    c = np.asarray(c, float)
    for _ in range(5):
        w, h = int(rng.choice([64, 128, 192])), int(rng.choice([64, 96, 128]))
        f = w / (2 * np.tan(np.radians(rng.choice([2, 20, 60, 110, 150])) / 2))
        K = np.array([[f * rng.choice([1, -1]), f * rng.choice([0, 0, 0.3]), w / 2 + rng.choice([0, 0, 3 * w]) * rng.choice([-1, 1])],
                      [0, f * rng.uniform(0.5, 2.0) * rng.choice([1, 1, -1]), h / 2 + rng.choice([0, 0.5, -2 * h])], [0, 0, 1.0]])
        # a random rotation (QR of a Gaussian matrix, made proper) and a position at 0.2 - 4 extents, usually looking at the object
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        dist = float(rng.choice([0.2, 0.6, 1.0, 2.5, 4.0])) * float(np.max(ext))
        eye = c + Q[:, 2] * (-dist if rng.random() < 0.8 else dist)
        Rinv = np.eye(4); Rinv[:3, :3] = Q; Rinv[:3, 3] = eye
        o, d = views.generate_ray(h, w, np.linalg.inv(K), Rinv, device="cuda")
        if not bool(torch.isfinite(d).all()):
            continue
        _check(Render, scene, o, d, w, h)


def test_partial_overwrite_behind_the_version_counter_is_caught_by_the_canary(Render):
    """A block of rows of a trusted image rewritten through `.data` (no version bump) and placed so that it misses the fixed 8 x 8 lattice:
    the lattice check alone trusts the image for ever (checked below with the canary compiled out of the call: DRT_GRID_CANARY=0 scene);
    the per-call canary of k_check_views -- 64 more rays per image at pixels that change from call to call -- sees it within a call or
    two, and from then on the image is verified ray by ray: the answer is that of the NEW rays."""
    import os
    from drt_amd import diffrender
    mesh = mesh_io.read_ply(data_path("hand_vh.ply"))
    res = 256
    P = res * res
    lattice_rows = {int(((res - 1) * k) // 7) for k in range(8)}
    y0, y1 = 112, 144                                        # 32 rows of image 1: 12.5 % of its rays, none of them on the lattice (rows 109, 145)
    assert not any(y0 <= r < y1 for r in lattice_rows)

    def run(canary):
        old = os.environ.get("DRT_GRID_CANARY")
        os.environ["DRT_GRID_CANARY"] = "1" if canary else "0"
        try:
            scene = Render.Scene(mesh, 0)                   # (the switch is read when the library creates the scene)
        finally:
            if old is None:
                os.environ.pop("DRT_GRID_CANARY", None)
            else:
                os.environ["DRT_GRID_CANARY"] = old
        o, d, cams = _trusted_pair(Render, scene, mesh, res, (4, 22, 47))
        ver = d._version
        block = d[P + y0 * res:P + y1 * res].clone().view(y1 - y0, res, 3)
        d.data[P + y0 * res:P + y1 * res].copy_(torch.roll(block, 20, dims=1).reshape(-1, 3))      # every ray of those rows looks where the pixel 20 columns away looked
        assert d._version == ver and (diffrender._grid_cache(o, d, len(o), res, res)[0] & 3) == 2
        Render.resx = Render.resy = res
        outs = []
        with torch.no_grad():
            for _ in range(3):
                outs.append([t.clone() for t in scene.render_transparent(o, d)])
            Render.resx = Render.resy = 7                   # the same rays without any whole-image assumption: the tree for every ray
            ref = [t.clone() for t in scene.render_transparent(o.clone(), d.clone())]
        Render.resx = Render.resy = res
        return outs, ref

    outs, ref = run(True)
    # (1 - 0.125)^64 = 2e-4 per call: the first trusting call catches it
    for k, got in enumerate(outs):
        assert all(torch.equal(a, b) for a, b in zip(got, ref)), k
    outs0, ref0 = run(False)
    assert all(torch.equal(a, b) for a, b in zip(ref0, ref))
    assert not all(torch.equal(a, b) for a, b in zip(outs0[-1], ref0)), "the lattice alone was expected to miss this overwrite (the test no longer tests the canary)"
