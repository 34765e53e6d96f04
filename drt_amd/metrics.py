"""Reconstruction error against a scanned mesh: the reference's acceptance metric.

The reference reports "average per-vertex distance (Hausdorff Distance)" between its result and the
scanned ground truth ``data/<name>_scan.ply`` and gets it from meshlabserver's Hausdorff-distance filter
(reference README.md:11; the call is not in the repository).  That filter samples the vertices of one
mesh, finds the closest point of the other mesh's surface for each, and prints min / max / mean / RMS.
Here the closest-point query runs on the same HIP LBVH as the ray tracer (``drt_closest_point``).
"""
from __future__ import annotations

import numpy as np
import torch

from . import diffrender as Render
from . import mesh_io


def _as_scene(target, cuda_device=0):
    if isinstance(target, Render.Scene):
        return target
    return Render.Scene(target, cuda_device)


def vertex_to_surface(points, target, want_face=False, want_point=False):
    """Distance of each point (float64 [N,3], GPU tensor or array) to the surface of ``target``
    (a Scene, a TriMesh or a PLY path).  Returns dist [N] (and face ids / closest points when asked)."""
    scene = _as_scene(target)
    dev = scene.vertices.device
    p = torch.as_tensor(np.asarray(points) if not isinstance(points, torch.Tensor) else points, dtype=torch.float64).to(dev).contiguous()
    dist, face, closest = scene.optix_mesh.closest_point(p.detach(), want_face=want_face, want_point=want_point)
    if want_face or want_point:
        return dist, face, closest
    return dist


def distance_stats(dist):
    """min / max / mean / RMS of a distance sample: the four numbers MeshLab's Hausdorff filter reports."""
    d = dist.detach()
    return {"min": d.min().item(), "max": d.max().item(), "mean": d.mean().item(), "rms": d.pow(2).mean().sqrt().item(), "n": d.numel()}


def hausdorff(result, scan, symmetric=False):
    """Vertex-sampled distance of ``result`` to ``scan`` (each a Scene, TriMesh or PLY path).  One-sided
    (result vertices -> scan surface) like the reference's per-vertex error; ``symmetric=True`` also samples
    the scan's vertices against the result and returns both directions plus the larger maximum."""
    def verts(m):
        if isinstance(m, Render.Scene):
            return m.vertices.detach()
        if isinstance(m, str):
            m = mesh_io.read_ply(m)
        return torch.as_tensor(m.vertices, dtype=torch.float64)

    scan_scene = _as_scene(scan)
    fwd = distance_stats(vertex_to_surface(verts(result), scan_scene))
    if not symmetric:
        return fwd
    bwd = distance_stats(vertex_to_surface(verts(scan), _as_scene(result)))
    return {"forward": fwd, "backward": bwd, "hausdorff": max(fwd["max"], bwd["max"])}
