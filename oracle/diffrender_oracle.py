"""oracle/diffrender_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (PyTorch float64 on the host, autograd for the gradients) of the
reference's differentiable refraction-tracing path, function by function, on top
of the brute-force float32 tracer in oracle/tracer.c.  It is the checker the HIP
path is compared with; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  Nothing here reads /root/reference.

Pinning: tests/golden/*.npz hold outputs of the reference's own Python
(DiffRender.py / optim.py imported unmodified in the build container with stubs
for trimesh / imageio / the OptiX extension, see tests/golden/make_golden.py);
tests/test_oracle_golden.py checks every function below against them.  The tracer
arithmetic itself (OptiX Prime, closed source) is parity-unpinned -- see tracer.c.

Reference map (file:line in /root/reference):
  trace_closest        <- optix_mesh.intersect            optix_extend.cpp:29-57
  intersect_ids        <- Scene.optix_intersect           DiffRender.py:386-392
  moller_trumbore      <- JIT_Dintersect                  DiffRender.py:64-121
  fresnel_tir          <- FrDielectric                    DiffRender.py:51-61
  refract_dir          <- Refract                         DiffRender.py:35-49
  bounce               <- Scene.Dintersect + refract_ray  DiffRender.py:492-535
  render_transparent   <- Scene.trace2 + render_transparent  DiffRender.py:420-432, 537-546
  ray_loss             <- Loss_calculator.ray_loss        optim.py:91-108
  face_pair_normals    <- edge_face_norm                  DiffRender.py:149-163
  silhouette_edges     <- Scene.silhouette_edge           DiffRender.py:445-457
  EdgeSample / primary_visibility <- primary_edge_sample / Scene.primary_visibility  DiffRender.py:189-267, 459-479
  vh_loss_view         <- body of Loss_calculator.vh_loss optim.py:73-78
  dihedral_cos / sm_loss <- Scene.dihedral_angle / sm_loss  DiffRender.py:440-443, optim.py:82-89
  total_loss           <- Loss_calculator.all_loss weights optim.py:127-129
  limit_grad / sgd_nesterov_step <- limit_hook + torch.optim.SGD(nesterov) optim.py:155-171, 215
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import torch

EXT_IOR = 1.00029          # DiffRender.py:21
F64 = torch.float64

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _tracer_lib():
    """Load (building on demand with gcc) oracle/_build/liboracle_tracer.so."""
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "_build", "liboracle_tracer.so")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("tracer.c", "bvh_tracer.c", "hit_point.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    lib = ctypes.CDLL(so)
    lib.oracle_trace_closest.restype = ctypes.c_int
    lib.oracle_trace_closest.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                         ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    lib.oracle_trace_closest_bvh.restype = ctypes.c_int
    lib.oracle_trace_closest_bvh.argtypes = lib.oracle_trace_closest.argtypes
    lib.oracle_trace_closest_mt_only.restype = ctypes.c_int
    lib.oracle_trace_closest_mt_only.argtypes = lib.oracle_trace_closest.argtypes
    lib.oracle_num_threads.restype = ctypes.c_int
    _LIB = lib
    return lib


def num_threads():
    return int(_tracer_lib().oracle_num_threads())


def set_num_threads(n):
    """OpenMP threads of the C tracers (PyTorch's own pool: torch.set_num_threads)."""
    _tracer_lib().oracle_set_num_threads(int(n))


# --------------------------------------------------------------------------- tracer
TRACER_THREADS = None  # None: whatever OpenMP picks; bench.py sets all cores here and keeps PyTorch's pool small (below)
TORCH_THREADS = None   # restored after every tracer call when TRACER_THREADS is set
USE_BVH = False      # bench.py's second cpu_baseline figure flips this: same contract, tree instead of the loop over every face


def trace_closest(faces_i32, verts_f32, rays_f32, bvh=None, mt_only=False):
    """Closest hit (brute force; oracle/bvh_tracer.c when ``bvh``/USE_BVH). faces i32 [F,3], verts f32 [V,3], rays f32 [N,6] -> (T f32 [N], ID i32 [N]).
    ``mt_only``: the four inequalities without the hit-point condition (not the contract: the far-camera tests' comparison)."""
    faces = np.ascontiguousarray(faces_i32, dtype=np.int32)
    verts = np.ascontiguousarray(verts_f32, dtype=np.float32)
    rays = np.ascontiguousarray(rays_f32, dtype=np.float32)
    n = rays.shape[0]
    T = np.empty(n, dtype=np.float32)
    ID = np.empty(n, dtype=np.int32)
    if n:
        fn = _tracer_lib().oracle_trace_closest_bvh if (USE_BVH if bvh is None else bvh) else _tracer_lib().oracle_trace_closest
        if mt_only:
            fn = _tracer_lib().oracle_trace_closest_mt_only
        if TRACER_THREADS:
            # PyTorch and this library may share one OpenMP runtime (one thread-count setting): the C tracer scales to every
            # core, PyTorch's small float64 ops collapse with 256 threads (30 s instead of 0.14 s for a 512x512 view)
            set_num_threads(TRACER_THREADS)
        rc = fn(faces.ctypes.data, faces.shape[0], verts.ctypes.data, verts.shape[0], rays.ctypes.data, n, T.ctypes.data, ID.ctypes.data)
        if TRACER_THREADS and TORCH_THREADS:
            torch.set_num_threads(TORCH_THREADS)
        if rc != 0:
            raise MemoryError("oracle tracer allocation failed")
    return T, ID


class Mesh:
    """faces i64 [F,3] + the float32 vertex copy handed to the tracer (DiffRender.py:311-313, 379-380)."""

    def __init__(self, faces, vertices):
        self.faces = torch.as_tensor(faces, dtype=torch.long)
        self.set_vertices(vertices)

    def set_vertices(self, vertices):
        self.vertices = vertices                                    # f64 [V,3], may require grad
        self._v32 = vertices.detach().to(torch.float32).numpy()
        self._f32 = self.faces.to(torch.int32).numpy()


def intersect_ids(mesh: Mesh, origin, direction):
    """f64 rays -> f32 -> closest hit -> (face ids i64 [N], hitted bool [N]); hitted = T > 0."""
    rays = torch.cat([origin.detach().to(torch.float32), direction.detach().to(torch.float32)], dim=1).numpy()
    T, ID = trace_closest(mesh._f32, mesh._v32, rays)
    return torch.from_numpy(ID.astype(np.int64)), torch.from_numpy(T > 0)


# --------------------------------------------------------------------------- per-hit float64 math
def _dot(a, b):
    return a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1] + a[:, 2] * b[:, 2]


def _cross(a, b):
    return torch.stack((a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1],
                        a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                        a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]), dim=1)


def _norm(a):
    return a.norm(p=2, dim=1, keepdim=True)


def moller_trumbore(o, d, tri):
    """(u, v, t, unit geometric normal) in float64; differentiable w.r.t. tri [H,3,3]."""
    v0, v1, v2 = tri[:, 0], tri[:, 1], tri[:, 2]
    e1, e2 = v1 - v0, v2 - v0
    p = _cross(d, e2)
    inv_det = 1 / _dot(e1, p)
    s = o - v0
    u = _dot(s, p) * inv_det
    q = _cross(s, e1)
    v = _dot(d, q) * inv_det
    t = _dot(e2, q) * inv_det
    n = _cross(e1, e2)
    n = n / _norm(n)
    return u, v, t, n


def fresnel_tir(cos_i, eta_i, eta_t):
    """Total-internal-reflection flag (the Fresnel term R of the reference is computed and dropped)."""
    sin_i = torch.sqrt((1 - cos_i * cos_i).clamp(0, 1))
    return sin_i * eta_i / eta_t >= 1


def refract_dir(wo, n, eta):
    """The reference's refracted direction (NOT Snell: cosThetaT uses sin2ThetaI, DiffRender.py:42)."""
    eta = eta.view(-1, 1)
    cos_i = _dot(n, wo).view(-1, 1)
    sin2_i = (1 - cos_i * cos_i).clamp(min=0)
    cos_t = torch.sqrt(1 - sin2_i.clamp(max=1))
    wt = eta * -wo + (eta * cos_i - cos_t) * n
    return wt / _norm(wt)


def bounce(mesh: Mesh, o, d, ior_int, ior_ext=EXT_IOR):
    """One Dintersect + refract_ray.  Returns dict with hitted [N], and per-hit (H) tensors."""
    ids, hitted = intersect_ids(mesh, o, d)
    fid = ids[hitted]
    tri = mesh.vertices[mesh.faces[fid]]
    oh, dh = o[hitted], d[hitted]
    u, v, t, n = moller_trumbore(oh, dh, tri)
    wo = -dh
    cos_i = _dot(wo, n).clamp(-1, 1)
    exc = torch.logical_not(cos_i > 0)
    sgn = torch.where(exc, -torch.ones_like(t), torch.ones_like(t))
    eta_i = torch.where(exc, torch.full_like(t, ior_int), torch.full_like(t, ior_ext))
    eta_t = torch.where(exc, torch.full_like(t, ior_ext), torch.full_like(t, ior_int))
    n = n * sgn.view(-1, 1)
    cos_i = cos_i * sgn
    tir = fresnel_tir(cos_i, eta_i, eta_t)
    wt = refract_dir(wo, n, eta_i / eta_t)
    new_o = oh + t.view(-1, 1) * dh
    new_o = new_o + 1e-5 * wt
    return dict(hitted=hitted, face=fid, u=u, v=v, t=t, n=n, refracted=torch.logical_not(tir),
                new_o=new_o, new_d=wt)


def render_transparent(mesh: Mesh, origin, ray_dir, ior_int, ior_ext=EXT_IOR, return_aux=False):
    """origin, ray_dir f64 [P,3] -> out_ori f64 [P,3], out_dir f64 [P,3], mask bool [P,3]."""
    P = ray_dir.shape[0]
    out_ori = torch.zeros(ray_dir.shape, dtype=F64)
    out_dir = torch.zeros(ray_dir.shape, dtype=F64)
    mask = torch.zeros(ray_dir.shape, dtype=torch.bool)
    ind = torch.arange(P)
    b1 = bounce(mesh, origin, ray_dir, ior_int, ior_ext)
    ind1 = ind[b1["hitted"]]
    r1 = b1["refracted"]
    o2, d2, ind2 = b1["new_o"][r1], b1["new_d"][r1], ind1[r1]
    b2 = bounce(mesh, o2, d2, ior_int, ior_ext)
    ind2h = ind2[b2["hitted"]]
    r2 = b2["refracted"]
    o3, d3, ind3 = b2["new_o"][r2], b2["new_d"][r2], ind2h[r2]
    _, occluded = intersect_ids(mesh, o3, d3)
    keep = torch.logical_not(occluded)
    vi = ind3[keep]
    out_ori[vi] = o3[keep]
    out_dir[vi] = d3[keep]
    mask[vi] = True
    if not return_aux:
        return out_ori, out_dir, mask
    face1 = torch.full((P,), -1, dtype=torch.long)
    face2 = torch.full((P,), -1, dtype=torch.long)
    face1[ind1] = b1["face"]
    face2[ind2h] = b2["face"]
    aux = dict(face1=face1, face2=face2, ind1=ind1, ind2=ind2, ind2h=ind2h, ind3=ind3, valid_ind=vi,
               b1=b1, b2=b2, occluded=occluded)
    return out_ori, out_dir, mask, aux


def ray_loss(out_ori, out_dir, mask, screen_pixel, valid):
    target = screen_pixel - out_ori.detach()
    target = target / target.norm(dim=1, keepdim=True)
    diff = out_dir - target
    valid_mask = valid * mask[:, 0]
    return diff[valid_mask].pow(2).sum()


# --------------------------------------------------------------------------- silhouette branch
def face_pair_normals(vertices, E2F):
    out = []
    for k in (0, 1):
        v0, v1, v2 = vertices[E2F[:, k, 0]], vertices[E2F[:, k, 1]], vertices[E2F[:, k, 2]]
        n = _cross(v1 - v0, v2 - v0)
        out.append(n / _norm(n))
    return out


def silhouette_edges(vertices, Edges, E2F, origin3):
    v = vertices.detach()
    n1, n2 = face_pair_normals(v, E2F)
    d1 = _dot(n1, origin3 - v[E2F[:, 0, 0]])
    d2 = _dot(n2, origin3 - v[E2F[:, 1, 0]])
    return Edges[torch.logical_xor(d1 > 0, d2 > 0)]


class EdgeSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, E_pos, mesh, camera_M, origin3):
        R, K, R_inv, K_inv = camera_M
        num = len(E_pos)
        ax, ay, bx, by = E_pos[:, 0, 0], E_pos[:, 0, 1], E_pos[:, 1, 0], E_pos[:, 1, 1]
        mid = torch.stack(((ax + bx) / 2, (ay + by) / 2), dim=1)
        N = torch.stack((ay - by, bx - ax), dim=1)
        Nn = N / N.norm(dim=1, keepdim=True)
        pts = torch.cat((mid + 1 * Nn, mid - 1 * Nn), dim=0).T            # [2,2n]
        W = torch.ones([1, pts.shape[1]], dtype=F64)
        cam = K_inv @ torch.cat([pts, W], dim=0)
        cam = torch.cat([cam, W], dim=0)
        world = (R_inv @ cam)[:3].T
        d = world - origin3.view(-1, 3)
        o = origin3.expand_as(d)
        _, hitted = intersect_ids(mesh, o, d)
        m = torch.zeros(2 * num)
        m[hitted] = 1
        f = m[:num] - m[num:]
        dE = torch.stack((torch.stack((-N[:, 0], -N[:, 0]), dim=1),
                          torch.stack((-N[:, 1], -N[:, 1]), dim=1)), dim=2)   # [n,2(endpoint),2(xy)]
        dE = dE * f.view(-1, 1, 1)
        valid_edge = f.abs() > 1e-5
        index = mid[valid_edge].to(torch.long)
        output = 0.5 * torch.ones(len(index))
        ctx.mark_non_differentiable(index)
        ctx.save_for_backward(dE, valid_edge)
        return index, output

    @staticmethod
    def backward(ctx, grad_index, grad_output):
        dE, valid_edge = ctx.saved_tensors
        dE = dE.clone()
        dE[valid_edge] *= grad_output.view(-1, 1, 1)
        return dE, None, None, None


def project_edges(vertices, sil_edges, camera_M, detach_depth=True):
    R, K, _, _ = camera_M
    V = vertices[sil_edges.reshape(-1)]
    vh = torch.cat([V, torch.ones([V.shape[0], 1], dtype=F64)], dim=1)
    vc = R @ vh.T
    if detach_depth:
        vc = torch.cat([vc[:2], vc[2:3].detach(), vc[3:]], dim=0)
    vc = K @ vc[:3]
    pix = vc[:2] / vc[2]
    return pix.T.reshape(-1, 2, 2)


def primary_visibility(mesh: Mesh, sil_edges, camera_M, origin3, resx, resy, detach_depth=True):
    E_pos = project_edges(mesh.vertices, sil_edges, camera_M, detach_depth)
    index, output = EdgeSample.apply(E_pos, mesh, camera_M, origin3)
    keep = (index[:, 0] < resx - 1) * (index[:, 1] < resy - 1) * (index[:, 0] >= 0) * (index[:, 1] >= 0)
    return index[keep], output[keep]


def vh_loss_view(mesh: Mesh, Edges, E2F, camera_M, origin3, soft_mask, resx, resy):
    sil = silhouette_edges(mesh.vertices, Edges, E2F, origin3)
    index, output = primary_visibility(mesh, sil, camera_M, origin3, resx, resy, detach_depth=True)
    return (soft_mask.view((resy, resx))[index[:, 1], index[:, 0]] - output).abs().sum()


def dihedral_cos(vertices, E2F):
    n1, n2 = face_pair_normals(vertices, E2F)
    return _dot(n1, n2)


def sm_loss(vertices, E2F):
    return (-torch.log(1 + dihedral_cos(vertices, E2F))).sum()


def total_loss(ray, vh, sm, resy, mean_len, ray_w=40.0, vh_w=2e-3, sm_w=0.08):
    return ray_w * 217.5 / resy / resy * ray + vh_w * 217.5 / resy * vh + sm_w * mean_len / 10 * sm


def limit_grad(grad, max_abs=1.0):
    g = grad.clone()
    g[torch.isnan(g)] = 0
    return g.clamp(-max_abs, max_abs)


def sgd_nesterov_step(param, grad, buf, lr, momentum):
    """torch.optim.SGD(nesterov=True, dampening=0, weight_decay=0) update; buf None on the first step."""
    buf = grad.clone() if buf is None else momentum * buf + grad
    return param - lr * (grad + momentum * buf), buf


# --------------------------------------------------------------------------- acceptance metric
def point_mesh_distance(points, verts, faces, chunk=256):
    """Brute-force distance from each point to a triangle mesh: min over faces of
    min(distance to the plane projection when it falls inside the triangle, distance to the three edge
    segments).  Deliberately NOT the Voronoi-region routine of drt_closest.h: an independent formulation
    of the same definition.  The reference has no code for this metric (it shells out to meshlabserver,
    README.md:11), so this row is pinned by the definition only: parity unpinned.
    Returns (dist [N], face [N])."""
    P = np.asarray(points, dtype=np.float64)
    V = np.asarray(verts, dtype=np.float64)
    F = np.asarray(faces)
    a, b, c = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
    n = np.cross(b - a, c - a)
    nn = (n * n).sum(1)

    def seg_d2(p, s0, s1):
        e = s1 - s0
        ee = (e * e).sum(1)
        t = ((p[:, None, :] - s0[None]) * e[None]).sum(2) / np.where(ee > 0, ee, 1.0)[None]
        t = np.clip(t, 0.0, 1.0)
        q = s0[None] + t[..., None] * e[None]
        r = p[:, None, :] - q
        return (r * r).sum(2)

    dist = np.empty(len(P))
    face = np.empty(len(P), dtype=np.int64)
    for s in range(0, len(P), chunk):
        p = P[s:s + chunk]
        d2 = np.minimum(np.minimum(seg_d2(p, a, b), seg_d2(p, b, c)), seg_d2(p, c, a))
        ap = p[:, None, :] - a[None]
        h = (ap * n[None]).sum(2)                                  # signed height * |n|
        with np.errstate(divide="ignore", invalid="ignore"):
            proj = p[:, None, :] - (h / nn[None])[..., None] * n[None]
            inside = np.ones(h.shape, dtype=bool)
            for s0, s1 in ((a, b), (b, c), (c, a)):
                inside &= (np.cross((s1 - s0)[None], proj - s0[None]) * n[None]).sum(2) >= 0
            plane_d2 = h * h / nn[None]
        d2 = np.where(inside & (nn[None] > 0), np.minimum(d2, plane_d2), d2)
        face[s:s + chunk] = d2.argmin(1)
        dist[s:s + chunk] = np.sqrt(d2.min(1))
    return dist, face
