"""The refraction path one step at a time -- the reference's internal Scene methods, for callers and debugging.

    Intersection                              reference DiffRender.py:285-296
    Scene.Dintersect(ray)                     DiffRender.py:492-501   (+ JIT_Dintersect :64-121)
    Scene.refract_ray(intersect)              DiffRender.py:503-535   (+ FrDielectric :51-61, Refract :35-49)
    Scene.trace2(ray)                         DiffRender.py:537-546
    Scene.project_vert(camera_M, V)           DiffRender.py:481-490

`Scene.render_transparent` does NOT go through these: it runs the fused HIP pipeline (one pass, hand-derived
adjoint).  These methods give the same values bounce by bounce as ordinary float64 torch tensors on the GPU,
differentiable by autograd, with face ids from the same HIP tracer (`optix_intersect`).  The reference's quirks
are reproduced: the refracted direction is not Snell's law (DiffRender.py:42: cos(theta_t) is taken from
sin^2(theta_i)), the new origin is offset by 1e-5 along the new direction, `n` is flipped in place when the ray
leaves the object, `project_vert` truncates towards zero.
"""
from __future__ import annotations

import torch


class Intersection:
    """u, v, t [H], unit geometric normal n [H,3], the hit rays and their face ids (reference DiffRender.py:285-296)."""

    def __init__(self, u, v, t, n, ray, faces_ind):
        self.u, self.v, self.t, self.n, self.ray, self.faces_ind = u, v, t, n, ray, faces_ind
        assert len(u) == len(v) == len(t) == len(n) == len(ray) == len(faces_ind)

    def __len__(self):
        return len(self.t)


def _dot(a, b):            # (a0 b0 + a1 b1) + a2 b2: the evaluation order of the reference's `dot`
    return a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1] + a[:, 2] * b[:, 2]


def _cross(a, b):
    return torch.stack((a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1],
                        a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                        a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]), dim=1)


def hit_point_terms(origin, direction, triangles):
    """Moller-Trumbore u, v, t and the unit geometric normal of each (ray, triangle [H,3,3]) pair, float64."""
    v0, v1, v2 = triangles[:, 0], triangles[:, 1], triangles[:, 2]
    e1, e2 = v1 - v0, v2 - v0
    p = _cross(direction, e2)
    inv_det = 1 / _dot(e1, p)
    s = origin - v0
    u = _dot(s, p) * inv_det
    q = _cross(s, e1)
    v = _dot(direction, q) * inv_det
    t = _dot(e2, q) * inv_det
    n = _cross(e1, e2)
    return u, v, t, n / n.norm(p=2, dim=1, keepdim=True)


class StepwiseMixin:
    """Mixed into drt_amd.diffrender.Scene."""

    def Dintersect(self, ray):
        faces_ind, hitted = self.optix_intersect(ray)
        hit_faces = faces_ind[hitted]
        ray_hitted = ray.select(hitted)
        u, v, t, n = hit_point_terms(ray_hitted.origin, ray_hitted.direction, self.vertices[self.faces[hit_faces]])
        return Intersection(u=u, v=v, t=t, n=n, ray=ray_hitted, faces_ind=hit_faces), hitted

    def refract_ray(self, intersect):
        from . import diffrender as Render              # intIOR / extIOR are module globals the caller may overwrite
        t, ray = intersect.t, intersect.ray
        wo = -ray.direction
        cos_i = _dot(wo, intersect.n).clamp(-1, 1)
        leaving = torch.logical_not(cos_i > 0)
        sign = torch.where(leaving, -torch.ones_like(t), torch.ones_like(t))
        eta_i = torch.where(leaving, torch.full_like(t, Render.intIOR), torch.full_like(t, Render.extIOR))
        eta_t = torch.where(leaving, torch.full_like(t, Render.extIOR), torch.full_like(t, Render.intIOR))
        n = intersect.n * sign.view(-1, 1)
        intersect.n = n                                 # the reference flips the stored normal in place
        cos_i = cos_i * sign
        sin_i = torch.sqrt((1 - cos_i * cos_i).clamp(0, 1))
        refracted = torch.logical_not(sin_i * eta_i / eta_t >= 1)        # no total internal reflection
        eta = (eta_i / eta_t).view(-1, 1)
        c = _dot(n, wo).view(-1, 1)
        sin2_i = (1 - c * c).clamp(min=0)
        cos_t = torch.sqrt(1 - sin2_i.clamp(max=1))
        wt = eta * -wo + (eta * c - cos_t) * n
        wt = wt / wt.norm(p=2, dim=1, keepdim=True)
        new_origin = ray.origin + t.view(-1, 1) * ray.direction
        new_origin = new_origin + 1e-5 * wt
        return refracted, type(ray)(new_origin, wt, ray.ray_ind)

    def trace2(self, ray):
        first, _ = self.Dintersect(ray)
        refracted, inside = self.refract_ray(first)
        second, _ = self.Dintersect(inside.select(refracted))
        refracted2, outside = self.refract_ray(second)
        return outside.select(refracted2)

    def project_vert(self, camera_M, V):
        R, K = camera_M[0], camera_M[1]
        hom = torch.cat([V, torch.ones([V.shape[0], 1], dtype=V.dtype, device=V.device)], dim=1)
        cam = K @ (R @ hom.T)[:3]
        return (cam[:2] / cam[2]).to(torch.long).T
