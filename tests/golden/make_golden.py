#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own Python, imported unmodified.

Run in the build container only (it needs /root/reference, which never travels):

    python tests/golden/make_golden.py

What is imported: /root/reference/DiffRender.py and optim.py (and, transitively,
config.py and captured_data.py) exactly as they are.  Three things they depend on do
not exist in this image and are replaced by stand-ins defined HERE before the import:

  * ``trimesh``   -> drt_amd.mesh_io (PLY load + edge tables; pins the edge order that
                     trimesh leaves implementation-defined),
  * ``imageio`` / ``cv2`` / ``h5py`` -> empty modules (never called on this path),
  * ``torch.utils.cpp_extension.load`` -> returns a module whose ``optix_mesh`` class
    answers ``intersect`` with the brute-force float32 closest hit of oracle/tracer.c
    (OptiX Prime is proprietary and absent, so the face ids pin OUR tracer contract;
    everything after the face id -- all float64 math, losses, gradients -- is the
    reference's own code).

The fixtures hold inputs (camera matrices, seeds, vertex arrays) and the reference's
outputs; they contain no reference source text.
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "tests", "golden")

from drt_amd import mesh_io, views            # noqa: E402
from oracle import diffrender_oracle as orc   # noqa: E402  (only its C tracer is used here)

IOR = 1.4723


# ----------------------------------------------------------------------------- stand-ins
class _StubMesh(mesh_io.TriMesh):
    @property
    def vertex_neighbors(self):
        nb = [set() for _ in range(len(self.vertices))]
        for a, b in self.edges:
            nb[a].add(int(b))
            nb[b].add(int(a))
        return [sorted(s) for s in nb]


def _install_stubs():
    tm = types.ModuleType("trimesh")

    def load(path, process=False):
        m = mesh_io.read_ply(path)
        return _StubMesh(m.vertices, m.faces)

    grouping = types.ModuleType("trimesh.grouping")

    def group_rows(rows, require_count=None):
        assert require_count == 2
        return mesh_io.group_rows_pairs(np.asarray(rows), int(np.asarray(rows).max()) + 1)

    grouping.group_rows = group_rows
    tm.load = load
    tm.grouping = grouping
    sys.modules["trimesh"] = tm
    sys.modules["trimesh.grouping"] = grouping
    for name in ("imageio", "cv2", "h5py"):
        sys.modules[name] = types.ModuleType(name)

    class optix_mesh:
        def __init__(self, device):
            self.F = self.V = None

        def update_mesh(self, F, V):
            self.F, self.V = F, V

        def update_vert(self, V):
            self.V = V

        def intersect(self, Ray):
            T, ID = orc.trace_closest(self.F.numpy(), self.V.numpy(), Ray.detach().numpy())  # the real extension reads raw data pointers
            return [torch.from_numpy(T), torch.from_numpy(ID)]

    ext = types.SimpleNamespace(optix_mesh=optix_mesh)
    import torch.utils.cpp_extension as cpp
    cpp.load = lambda *a, **k: ext


def _import_reference():
    _install_stubs()
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import DiffRender
        import optim
    finally:
        os.chdir(cwd)
    DiffRender.device = "cpu"
    DiffRender.intIOR = IOR
    optim.device = "cpu"
    return DiffRender, optim


def _smooth(mesh, iters=10, lam=0.5):
    """Umbrella smoothing so that no dihedral cosine is -1 (raw hand_vh gives sm_loss = inf)."""
    V = mesh.vertices.copy()
    e = mesh.edges
    for _ in range(iters):
        acc = np.zeros_like(V)
        cnt = np.zeros(len(V))
        np.add.at(acc, e[:, 0], V[e[:, 1]])
        np.add.at(cnt, e[:, 0], 1)
        V = (1 - lam) * V + lam * acc / cnt[:, None]
    return V.astype(np.float32).astype(np.float64)


def _targets(P, center, seed):
    rng = np.random.default_rng(seed)
    sp = rng.standard_normal((P, 3)) * 40.0 + np.asarray(center) + np.array([0.0, 0.0, 150.0])
    valid = rng.random(P) > 0.1
    return sp, valid


# ----------------------------------------------------------------------------- fixtures
def unit_tables(DR):
    rng = np.random.default_rng(7)
    n = 256
    wo = rng.standard_normal((n, 3))
    wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    nn = rng.standard_normal((n, 3))
    nn /= np.linalg.norm(nn, axis=1, keepdims=True)
    nn[:8] = wo[:8]                                  # normal incidence
    nn[8:16] = np.cross(wo[8:16], nn[8:16])          # grazing: cos = 0
    nn[8:16] /= np.linalg.norm(nn[8:16], axis=1, keepdims=True)
    eta = np.where(rng.random(n) > 0.5, 1.00029 / IOR, IOR / 1.00029)
    two, tn, te = (torch.tensor(a) for a in (wo, nn, eta))
    tir_r, wt = DR.Refract(two, tn, te)
    cos = DR.dot(two, tn).clamp(-1, 1).abs()
    etaI = torch.where(te < 1, torch.tensor(1.00029, dtype=torch.float64), torch.tensor(IOR, dtype=torch.float64))
    etaT = torch.where(te < 1, torch.tensor(IOR, dtype=torch.float64), torch.tensor(1.00029, dtype=torch.float64))
    tir_f, Rf = DR.FrDielectric(cos, etaI, etaT)
    o = rng.standard_normal((n, 3)) * 100
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tri = rng.standard_normal((n, 3, 3)) * 20
    u, v, t, nrm = DR.JIT_Dintersect(torch.tensor(o), torch.tensor(d), torch.tensor(tri), torch.tensor(tri))
    np.savez_compressed(os.path.join(OUT, "unit_tables.npz"), wo=wo, n=nn, eta=eta, refract_tir=tir_r.numpy(),
                        refract_wt=wt.numpy(), fr_cos=cos.numpy(), fr_etaI=etaI.numpy(), fr_etaT=etaT.numpy(),
                        fr_tir=tir_f.numpy(), fr_R=Rf.numpy(), mt_o=o, mt_d=d, mt_tri=tri, mt_u=u.numpy(),
                        mt_v=v.numpy(), mt_t=t.numpy(), mt_n=nrm.numpy())


def render_fixture(DR, optim, scene, mesh, center, extent, res, view_id, tag):
    DR.resx = DR.resy = res
    cams = views.turntable_cameras(center, extent, 72, res, res)
    R, K, Rinv, Kinv = cams[view_id]
    origin, ray_dir = views.generate_ray(res, res, Kinv, Rinv)
    P = origin.shape[0]
    V0 = torch.tensor(mesh.vertices, dtype=torch.float64)
    V = V0.clone().requires_grad_(True)
    scene.update_verticex(V)

    rec = {}
    # -- per-bounce internals, by calling the reference's own pieces in trace2's order
    ray = DR.Ray(origin, ray_dir)
    it1, hit1 = scene.Dintersect(ray)
    refr1, ray2 = scene.refract_ray(it1)
    ray2s = ray2.select(refr1)
    it2, hit2 = scene.Dintersect(ray2s)
    refr2, ray3 = scene.refract_ray(it2)
    ray3s = ray3.select(refr2)
    _, occl = scene.optix_intersect(ray3s)
    rec.update(b1_ind=it1.ray.ray_ind.numpy(), b1_face=it1.faces_ind.numpy(), b1_u=it1.u.detach().numpy(),
               b1_v=it1.v.detach().numpy(), b1_t=it1.t.detach().numpy(), b1_n=it1.n.detach().numpy(),
               b1_refracted=refr1.numpy(), b1_new_o=ray2.origin.detach().numpy(), b1_new_d=ray2.direction.detach().numpy(),
               b2_ind=it2.ray.ray_ind.numpy(), b2_hitted=hit2.numpy(), b2_face=it2.faces_ind.numpy(),
               b2_t=it2.t.detach().numpy(), b2_n=it2.n.detach().numpy(), b2_refracted=refr2.numpy(),
               b2_new_o=ray3.origin.detach().numpy(), b2_new_d=ray3.direction.detach().numpy(),
               occluded=occl.numpy(), hit1_count=int(hit1.sum()))

    # -- the public entry point + ray_loss + gradient
    scene.update_verticex(V)
    out_ori, out_dir, mask = scene.render_transparent(origin, ray_dir)
    sp, valid = _targets(P, center, seed=100 + view_id)
    tsp, tvalid = torch.tensor(sp), torch.tensor(valid)
    target = tsp - out_ori.detach()
    target = target / target.norm(dim=1, keepdim=True)
    vm = tvalid * mask[:, 0]
    ray_loss = (out_dir - target)[vm].pow(2).sum()
    g_ray, = torch.autograd.grad(ray_loss, V, retain_graph=True)
    rng = np.random.default_rng(200 + view_id)
    w_ori = rng.standard_normal((P, 3))
    w_dir = rng.standard_normal((P, 3))
    lin = (out_ori * torch.tensor(w_ori)).sum() + (out_dir * torch.tensor(w_dir)).sum()
    g_lin, = torch.autograd.grad(lin, V)
    vi = torch.nonzero(mask[:, 0]).squeeze(1)
    rec.update(valid_ind=vi.numpy(), out_ori=out_ori.detach()[vi].numpy(), out_dir=out_dir.detach()[vi].numpy(),
               ray_loss=ray_loss.item(), grad_ray_loss=g_ray.numpy(), lin_seed=200 + view_id, lin=lin.item(),
               grad_lin=g_lin.numpy(), target_seed=100 + view_id)

    # -- silhouette branch of the same view
    scene.update_verticex(V)
    camera_M = tuple(torch.tensor(a, dtype=torch.float64) for a in (R, K, Rinv, Kinv))
    o3 = origin[0]
    sil = scene.silhouette_edge(o3)
    index, output = scene.primary_visibility(sil, camera_M, o3, detach_depth=True)
    hitmask = np.zeros(P, dtype=np.uint8)
    hitmask[it1.ray.ray_ind.numpy()] = 1
    soft = views.process_mask(hitmask.reshape(res, res))
    tsoft = torch.tensor(soft, dtype=torch.float64).reshape(-1)
    vh = (tsoft.view((res, res))[index[:, 1], index[:, 0]] - output).abs().sum()
    g_vh, = torch.autograd.grad(vh, V)
    rec.update(sil_edges=sil.numpy(), vh_index=index.numpy(), vh_output=output.detach().numpy(),
               vh_output_dtype=str(output.dtype), vh_loss=vh.item(), grad_vh=g_vh.numpy(), soft_mask=soft.astype(np.float32))
    rec.update(res=res, view_id=view_id, R=R, K=K, Rinv=Rinv, Kinv=Kinv, ior=IOR)
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **rec)
    print(tag, "hit1", rec["hit1_count"], "valid", len(vi), "sil", len(sil), "vh idx", len(index),
          "ray_loss", ray_loss.item(), "vh", vh.item())


def smooth_fixture(DR, optim, scene, mesh, center, extent):
    Vs = _smooth(mesh)
    V = torch.tensor(Vs, dtype=torch.float64, requires_grad=True)
    scene.update_verticex(V)
    cosang = scene.dihedral_angle()
    sm = (-torch.log(1 + cosang)).sum()
    g_sm, = torch.autograd.grad(sm, V)
    np.savez_compressed(os.path.join(OUT, "hand_smooth_sm.npz"), vertices=Vs.astype(np.float32), dihedral_cos=cosang.detach().numpy(),
                        sm_loss=sm.item(), grad_sm=g_sm.numpy(), mean_len=scene.mean_len,
                        Edges=scene.Edges.numpy(), E2F=scene.E2F.numpy())
    print("sm", sm.item(), "min cos", cosang.min().item())

    # -- two full optimisation steps through the reference's Loss_calculator / limit_hook / SGD
    res = 64
    DR.resx = DR.resy = res
    cams = views.turntable_cameras(center, extent, 72, res, res)
    sil_ids = list(range(0, 72, 9))

    class FakeData:
        resx = resy = res

        def __init__(self):
            self.cache = {}

        def get_view(self, k):
            if k not in self.cache:
                R, K, Rinv, Kinv = cams[k]
                o, d = views.generate_ray(res, res, Kinv, Rinv)
                P = o.shape[0]
                sp, valid = _targets(P, center, seed=100 + k)
                m0 = orc.Mesh(mesh.faces, torch.tensor(Vs))
                _, hit = orc.intersect_ids(m0, o, d)
                soft = views.process_mask(hit.numpy().reshape(res, res))
                cam = tuple(torch.tensor(a, dtype=torch.float64) for a in (R, K, Rinv, Kinv))
                self.cache[k] = (torch.tensor(sp), torch.tensor(valid), torch.tensor(soft, dtype=torch.float64).reshape(-1), o, d, cam)
            return self.cache[k]

        def ray_view_generator(self):
            while True:
                for k in (5, 23):
                    yield k

        def silh_view_generator(self):
            while True:
                for k in sil_ids:
                    yield k

    HP = dict(ray_w=40, sm_w=0.08, vh_w=2e-3, momentum=0.95)
    data = FakeData()
    lc = optim.Loss_calculator(scene, data, HP)
    init_vertices = torch.tensor(Vs, dtype=torch.float64)
    parameter = torch.zeros(init_vertices.shape, dtype=torch.float64, requires_grad=True)

    def limit_hook(grad):        # the reference's closure (optim.py:155-162) is local to optimize(); same statements
        grad[torch.isnan(grad)] = 0
        grad[grad > 1] = 1
        grad[grad < -1] = -1
        return grad

    parameter.register_hook(limit_hook)
    opt = torch.optim.SGD([parameter], lr=0.1, momentum=HP["momentum"], nesterov=True)
    steps = []
    for it in range(2):
        opt.zero_grad()
        vertices = init_vertices + parameter
        scene.update_verticex(vertices)
        loss, loss_str = lc.all_loss()
        loss.backward()
        steps.append(dict(loss=loss.item(), loss_str=loss_str, grad=parameter.grad.clone().numpy()))
        opt.step()
        steps[-1]["param"] = parameter.detach().clone().numpy()
        print("step", it, loss_str, "LOSS", loss.item())
    np.savez_compressed(os.path.join(OUT, "hand_smooth_steps.npz"), vertices=Vs.astype(np.float32), res=res, ray_views=np.array([5, 23]),
                        sil_views=np.array(sil_ids), lr=0.1, momentum=0.95, ior=IOR, mean_len=scene.mean_len,
                        loss0=steps[0]["loss"], loss1=steps[1]["loss"], grad0=steps[0]["grad"], grad1=steps[1]["grad"],
                        param0=steps[0]["param"], param1=steps[1]["param"], loss_str0=steps[0]["loss_str"], loss_str1=steps[1]["loss_str"])


def trajectory_fixture(DR, optim, mesh, center, extent, iters=60, every=10, res=64, seed=12, tag="hand_trajectory", name="hand", sigma=0.4):
    """One pass of the reference's own loop (optim.py:190-215: update_verticex -> all_loss -> backward -> limit_hook -> SGD
    nesterov), `iters` iterations at the reference's hyper-parameters (config.py:18-39: lr = start_lr of pass 0), with the
    reference's own Loss_calculator and its own stochastic view order: the imported captured_data.Data's
    ray_view_generator / silh_view_generator (captured_data.py:61-82) under np.random.seed(seed), one refraction view and
    eight silhouette views per iteration.  The capture is synthetic (the .h5 captures are not distributed): 72 turntable
    views at res x res of a ground-truth mesh (the smoothed hand hull displaced along its normals), traced by the oracle
    -- inputs, stored in the fixture.  Stored outputs: the realised view schedule, loss / loss string / max |grad| of every
    iteration, the parameter every `every` iterations and at the end."""
    sys.path.insert(0, REF)
    import captured_data as cd
    cd.device = "cpu"
    Vs = _smooth(mesh)
    with __import__("tempfile").TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "hand_smooth.ply")
        mesh_io.write_ply(path, Vs, mesh.faces)
        scene = DR.Scene(path)
    DR.resx = DR.resy = res
    gt = views.displaced_ground_truth(mesh_io.TriMesh(Vs, mesh.faces), sigma=sigma, seed=5)
    gt_mesh = orc.Mesh(gt.faces, torch.tensor(gt.vertices))

    def render_gt(o, d):
        with torch.no_grad():
            return orc.render_transparent(gt_mesh, o, d, IOR)

    def hit_gt(o, d):
        return orc.intersect_ids(gt_mesh, o, d)[1]

    vs = views.make_views(render_gt, hit_gt, center, extent, 72, res, res)
    data = cd.Data.__new__(cd.Data)
    data.name, data.num_view, data.resx, data.resy = name, 72, res, res
    data.Views = vs
    schedule = {"ray": [], "silh": []}
    get_view = data.get_view

    def logged_get_view(k):         # (the order the reference's generators hand the views out in, as consumed by all_loss)
        schedule["cur"].append(int(k))
        return get_view(k)
    data.get_view = logged_get_view

    HP = dict(ray_w=40, sm_w=0.08, vh_w=2e-3, momentum=0.95)
    np.random.seed(seed)
    lc = optim.Loss_calculator(scene, data, HP)
    init_vertices = torch.tensor(Vs, dtype=torch.float64)
    parameter = torch.zeros(init_vertices.shape, dtype=torch.float64, requires_grad=True)

    def limit_hook(grad):        # the reference's closure (optim.py:155-162) is local to optimize(); same statements
        grad[torch.isnan(grad)] = 0
        grad[grad > 1] = 1
        grad[grad < -1] = -1
        return grad

    parameter.register_hook(limit_hook)
    lr = 0.1
    opt = torch.optim.SGD([parameter], lr=lr, momentum=HP["momentum"], nesterov=True)
    rec = dict(loss=[], loss_str=[], gmax=[], params=[], param_its=[])
    for it in range(iters):
        opt.zero_grad()
        vertices = init_vertices + parameter
        scene.update_verticex(vertices)
        schedule["cur"] = []
        loss, loss_str = lc.all_loss()
        loss.backward()
        schedule["ray"].append(schedule["cur"][0])
        schedule["silh"].append(schedule["cur"][1:])
        assert len(schedule["cur"]) == 9
        rec["loss"].append(loss.item()); rec["loss_str"].append(loss_str); rec["gmax"].append(parameter.grad.abs().max().item())
        opt.step()
        if (it + 1) % every == 0 or it + 1 == iters:
            rec["params"].append(parameter.detach().clone().numpy()); rec["param_its"].append(it + 1)
        if it % 10 == 0:
            print("trajectory", it, loss_str, "LOSS", loss.item(), "gmax", rec["gmax"][-1])
    used = sorted(set(schedule["ray"]))
    np.savez_compressed(
        os.path.join(OUT, tag + ".npz"), vertices=Vs.astype(np.float32), faces=np.asarray(mesh.faces, dtype=np.int32), res=res, lr=lr, momentum=HP["momentum"], ior=IOR,
        ray_w=HP["ray_w"], sm_w=HP["sm_w"], vh_w=HP["vh_w"], mean_len=scene.mean_len, seed=seed,
        ray_schedule=np.array(schedule["ray"]), silh_schedule=np.array(schedule["silh"]),
        # inputs of the synthetic capture: targets of the refraction views that were used, soft masks of all 72 views (float32 holds them exactly:
        # process_mask yields multiples of 1/2 clipped EDT values -- checked below)
        ray_views=np.array(used), screen_pixel=np.stack([vs[k][0].numpy() for k in used]),
        soft_mask=np.stack([vs[k][2].numpy() for k in range(72)]),
        loss=np.array(rec["loss"]), loss_str=np.array(rec["loss_str"]), gmax=np.array(rec["gmax"]),
        param_its=np.array(rec["param_its"]), params=np.stack(rec["params"]))
    print("wrote", tag + ".npz: loss", rec["loss"][0], "->", rec["loss"][-1], "distinct ray views", len(used))


def degenerate_fixture(DR, optim, mesh, center, extent):
    """A closed mesh with the defect SURVEY section 4 notes in monkey_vh.ply / dog_vh.ply: a zero-length edge, i.e. two
    zero-area faces (monkey_vh: faces with a duplicated vertex position).  Made from the smoothed hand hull by moving
    one endpoint of edge 100 onto the other (topology untouched, still watertight).  Pins what the reference does
    with it: a zero-area face is never hit (det = 0), its normal n / |n| is NaN (DiffRender.py:103-104, 149-163), so
    its edges drop out of the silhouette test, their dihedral cosines and the smoothness loss are NaN, the NaN
    reaches the vertex gradient and limit_hook (optim.py:155-162) zeroes it before the SGD step."""
    import tempfile
    Vs = _smooth(mesh)
    a, b = (int(x) for x in np.sort(mesh.edges, axis=1)[100])
    Vs[b] = Vs[a]
    tri = Vs[mesh.faces]
    area = np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    assert (area == 0).sum() == 2
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "hand_degenerate.ply")
        mesh_io.write_ply(path, Vs, mesh.faces)
        scene = DR.Scene(path)
    res, view_id = 64, 23
    DR.resx = DR.resy = res
    cams = views.turntable_cameras(center, extent, 72, res, res)
    R, K, Rinv, Kinv = cams[view_id]
    origin, ray_dir = views.generate_ray(res, res, Kinv, Rinv)
    P = origin.shape[0]
    V = torch.tensor(Vs, dtype=torch.float64, requires_grad=True)
    scene.update_verticex(V)
    rec = dict(vertices=Vs.astype(np.float32), collapsed=np.array([a, b]), zero_area_faces=np.flatnonzero(area == 0),
               res=res, view_id=view_id, R=R, K=K, Rinv=Rinv, Kinv=Kinv, ior=IOR, mean_len=scene.mean_len)
    # refraction path of one view (zero-area faces can never be hit)
    out_ori, out_dir, mask = scene.render_transparent(origin, ray_dir)
    sp, valid = _targets(P, center, seed=100 + view_id)
    target = torch.tensor(sp) - out_ori.detach()
    target = target / target.norm(dim=1, keepdim=True)
    vm = torch.tensor(valid) * mask[:, 0]
    ray_loss = (out_dir - target)[vm].pow(2).sum()
    g_ray, = torch.autograd.grad(ray_loss, V)
    vi = torch.nonzero(mask[:, 0]).squeeze(1)
    it1, _ = scene.Dintersect(DR.Ray(origin, ray_dir))
    rec.update(valid_ind=vi.numpy(), out_dir=out_dir.detach()[vi].numpy(), out_ori=out_ori.detach()[vi].numpy(), ray_loss=ray_loss.item(),
               grad_ray_loss=g_ray.numpy(), target_seed=100 + view_id, b1_ind=it1.ray.ray_ind.numpy(), b1_face=it1.faces_ind.numpy())
    # silhouette branch
    scene.update_verticex(V)
    camera_M = tuple(torch.tensor(m, dtype=torch.float64) for m in (R, K, Rinv, Kinv))
    o3 = origin[0]
    sil = scene.silhouette_edge(o3)
    index, output = scene.primary_visibility(sil, camera_M, o3, detach_depth=True)
    hitmask = np.zeros(P, dtype=np.uint8)
    hitmask[it1.ray.ray_ind.numpy()] = 1
    soft = torch.tensor(views.process_mask(hitmask.reshape(res, res)), dtype=torch.float64).reshape(-1)
    vh = (soft.view((res, res))[index[:, 1], index[:, 0]] - output).abs().sum()
    g_vh, = torch.autograd.grad(vh, V)
    rec.update(sil_edges=sil.numpy(), vh_index=index.numpy(), vh_output=output.detach().numpy(), vh_loss=vh.item(), grad_vh=g_vh.numpy())
    # smoothness branch: NaN where a zero-area face is involved
    scene.update_verticex(V)
    cosang = scene.dihedral_angle()
    sm = (-torch.log(1 + cosang)).sum()
    g_sm, = torch.autograd.grad(sm, V)
    rec.update(dihedral_cos=cosang.detach().numpy(), sm_loss=sm.item(), grad_sm=g_sm.numpy(), Edges=scene.Edges.numpy())
    assert np.isnan(rec["dihedral_cos"]).sum() >= 1 and np.isnan(rec["grad_sm"]).any() and np.isnan(sm.item())
    # one whole iteration: all_loss -> backward -> limit_hook (NaN -> 0, clamp) -> SGD(nesterov)
    sil_ids = list(range(0, 72, 9))

    class FakeData:
        resx = resy = res

        def __init__(self):
            self.cache = {}

        def get_view(self, k):
            if k not in self.cache:
                Rk, Kk, Rinvk, Kinvk = cams[k]
                o, d = views.generate_ray(res, res, Kinvk, Rinvk)
                spk, validk = _targets(o.shape[0], center, seed=100 + k)
                _, hit = orc.intersect_ids(orc.Mesh(mesh.faces, torch.tensor(Vs)), o, d)
                softk = views.process_mask(hit.numpy().reshape(res, res))
                cam = tuple(torch.tensor(m, dtype=torch.float64) for m in (Rk, Kk, Rinvk, Kinvk))
                self.cache[k] = (torch.tensor(spk), torch.tensor(validk), torch.tensor(softk, dtype=torch.float64).reshape(-1), o, d, cam)
            return self.cache[k]

        def ray_view_generator(self):
            while True:
                yield view_id

        def silh_view_generator(self):
            while True:
                for k in sil_ids:
                    yield k

    HP = dict(ray_w=40, sm_w=0.08, vh_w=2e-3, momentum=0.95)
    lc = optim.Loss_calculator(scene, FakeData(), HP)
    init_vertices = torch.tensor(Vs, dtype=torch.float64)
    parameter = torch.zeros(init_vertices.shape, dtype=torch.float64, requires_grad=True)

    def limit_hook(grad):        # the reference's closure (optim.py:155-162), same statements
        grad[torch.isnan(grad)] = 0
        grad[grad > 1] = 1
        grad[grad < -1] = -1
        return grad

    parameter.register_hook(limit_hook)
    opt = torch.optim.SGD([parameter], lr=0.1, momentum=HP["momentum"], nesterov=True)
    opt.zero_grad()
    scene.update_verticex(init_vertices + parameter)
    loss, loss_str = lc.all_loss()
    loss.backward()
    grad0 = parameter.grad.clone().numpy()
    opt.step()
    assert np.isnan(loss.item()) and np.isfinite(grad0).all() and np.isfinite(parameter.detach().numpy()).all()
    rec.update(step_loss_is_nan=True, step_loss_str=loss_str, step_grad=grad0, step_param=parameter.detach().numpy(), sil_views=np.array(sil_ids),
               lr=0.1, momentum=0.95)
    np.savez_compressed(os.path.join(OUT, "hand_degenerate.npz"), **rec)
    print("degenerate: zero-area faces", rec["zero_area_faces"], "NaN cos", int(np.isnan(rec["dihedral_cos"]).sum()),
          "NaN grad_sm rows", int(np.isnan(rec["grad_sm"]).any(axis=1).sum()), "sil", len(sil), "vh", vh.item(), "valid", len(vi), loss_str)


def big_mesh_fixture(DR, optim, name, view_id, tag):
    """A BASELINE-size hull (`name`_vh.ply after one midpoint subdivision) through the reference's own Python, one view at 256x256."""
    import hashlib
    import tempfile
    hull = mesh_io.subdivide_midpoint(mesh_io.read_ply(os.path.join(REPO, "data", f"{name}_vh.ply")))
    center, extent = views.mesh_frame(hull.vertices)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, f"{name}_x4.ply")
        mesh_io.write_ply(path, hull.vertices, hull.faces)
        mesh = mesh_io.read_ply(path)            # (float32 on disk: what Scene(path) sees)
        scene = DR.Scene(path)
        render_fixture(DR, optim, scene, mesh, center, extent, 256, view_id, tag)
    f = os.path.join(OUT, tag + ".npz")
    rec = dict(np.load(f))
    rec.update(n_faces=len(mesh.faces), n_vertices=len(mesh.vertices), hull=name,
               mesh_sha256=hashlib.sha256(np.ascontiguousarray(mesh.vertices, np.float64).tobytes() + np.ascontiguousarray(mesh.faces, np.int64).tobytes()).hexdigest())
    np.savez_compressed(f, **rec)
    print(tag, "fixture:", rec["n_faces"], "faces,", os.path.getsize(f), "bytes")


def horse_fixture(DR, optim):
    """The HEADLINE mesh (BASELINE.json's ~50k-triangle workload: horse_vh.ply after one midpoint subdivision = 50 248 triangles,
    25 126 vertices), one view at 256x256, through the reference's own Python: per-bounce ids and terms, outputs, ray_loss and its
    gradient, the silhouette branch.  The subdivided hull is written to a temporary PLY for the reference's Scene(path)."""
    import hashlib
    import tempfile
    hull = mesh_io.subdivide_midpoint(mesh_io.read_ply(os.path.join(REPO, "data", "horse_vh.ply")))
    center, extent = views.mesh_frame(hull.vertices)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "horse_x4.ply")
        mesh_io.write_ply(path, hull.vertices, hull.faces)
        mesh = mesh_io.read_ply(path)            # (float32 on disk: what Scene(path) sees)
        scene = DR.Scene(path)
        render_fixture(DR, optim, scene, mesh, center, extent, 256, 11, "horse50k_r256_v11")
    f = os.path.join(OUT, "horse50k_r256_v11.npz")
    rec = dict(np.load(f))
    rec.update(n_faces=len(mesh.faces), n_vertices=len(mesh.vertices),
               mesh_sha256=hashlib.sha256(np.ascontiguousarray(mesh.vertices, np.float64).tobytes() + np.ascontiguousarray(mesh.faces, np.int64).tobytes()).hexdigest())
    np.savez_compressed(f, **rec)
    print("horse fixture:", rec["n_faces"], "faces,", os.path.getsize(f), "bytes")


def horse_trajectory(DR, optim):
    """The same pass of the reference's loop on the HEADLINE mesh (horse_vh.ply x 4 = 50 248 triangles, smoothed like the hand hull so that
    sm_loss is finite): 40 iterations, checkpoints at 20 and 40."""
    hull = mesh_io.subdivide_midpoint(mesh_io.read_ply(os.path.join(REPO, "data", "horse_vh.ply")))
    center, extent = views.mesh_frame(hull.vertices)
    trajectory_fixture(DR, optim, hull, center, extent, iters=40, every=20, res=64, seed=21, tag="horse50k_trajectory", name="horse", sigma=0.3)


def main():
    torch.manual_seed(0)
    np.random.seed(0)
    os.makedirs(OUT, exist_ok=True)
    DR, optim = _import_reference()
    path = os.path.join(REPO, "data", "hand_vh.ply")
    mesh = mesh_io.read_ply(path)
    center, extent = views.mesh_frame(mesh.vertices)
    only = set(sys.argv[1:])                 # e.g. `make_golden.py degenerate`: regenerate one family of fixtures
    if only == {"degenerate"}:
        return degenerate_fixture(DR, optim, mesh, center, extent)
    if only == {"horse"}:
        return horse_fixture(DR, optim)
    if only == {"trajectory"}:
        return trajectory_fixture(DR, optim, mesh, center, extent)
    if only == {"horse_trajectory"}:
        return horse_trajectory(DR, optim)
    if only == {"mouse"}:       # BASELINE.json configs[2]: mouse_vh.ply subdivided to 36 984 triangles
        return big_mesh_fixture(DR, optim, "mouse", 29, "mouse37k_r256_v29")
    scene = DR.Scene(path)
    np.savez_compressed(os.path.join(OUT, "hand_topology.npz"), Edges=scene.Edges.numpy(), E2F=scene.E2F.numpy(),
                        mean_len=scene.mean_len, n_vertices=len(mesh.vertices), n_faces=len(mesh.faces))
    unit_tables(DR)
    for res in (64, 128):
        for view_id in (5, 23, 41):
            render_fixture(DR, optim, scene, mesh, center, extent, res, view_id, f"hand_r{res}_v{view_id}")
    smooth_fixture(DR, optim, scene, mesh, center, extent)
    trajectory_fixture(DR, optim, mesh, center, extent)
    horse_trajectory(DR, optim)
    degenerate_fixture(DR, optim, mesh, center, extent)
    horse_fixture(DR, optim)
    big_mesh_fixture(DR, optim, "mouse", 29, "mouse37k_r256_v29")


if __name__ == "__main__":
    main()
