"""Synthetic capture views with the tuple layout of the reference's data layer.

The reference reads 72 calibrated views from an HDF5 capture that is not
distributed with it (reference captured_data.py:84-165, .gitignore:4).  This module
builds views of the same shape -- ``(screen_pixel, valid, mask, origin, ray_dir,
camera_M)`` as returned by ``Data.get_view`` (captured_data.py:44-59) -- from a
turntable of pinhole cameras around the mesh:

* rays exactly as ``generate_ray`` (captured_data.py:23-40): integer pixel
  coordinates, no half-pixel offset, origin = R^-1[:3,3], dir = normalize(R^-1 K^-1 p);
* ``screen_pixel``: where the exit ray of a ground-truth mesh meets a background plane
  behind the object; zeros where the ground-truth path is invalid so that
  ``valid = screen_pixel[:,0] != 0`` as at captured_data.py:106;
* ``mask``: the soft silhouette of ``process_mask`` (captured_data.py:12-20).
"""
from __future__ import annotations

import math

import numpy as np
import torch

F64 = torch.float64


def turntable_cameras(center, extent, n_views, resx, resy, distance_factor=2.5, focal_factor=1.1):
    """List of camera_M = (R 4x4 world->camera, K 3x3, R^-1, K^-1) as float64 numpy arrays."""
    center = np.asarray(center, dtype=np.float64)
    K = np.array([[focal_factor * resx, 0.0, resx / 2.0], [0.0, focal_factor * resx, resy / 2.0], [0.0, 0.0, 1.0]])
    Kinv = np.linalg.inv(K)
    cams = []
    up = np.array([0.0, 1.0, 0.0])
    for k in range(n_views):
        az = 2.0 * math.pi * k / n_views
        eye = center + distance_factor * extent * np.array([math.sin(az), 0.0, math.cos(az)])
        zc = center - eye
        zc /= np.linalg.norm(zc)
        yc = -(up - np.dot(up, zc) * zc)
        yc /= np.linalg.norm(yc)
        xc = np.cross(yc, zc)
        R = np.eye(4)
        R[0, :3], R[1, :3], R[2, :3] = xc, yc, zc
        R[:3, 3] = -R[:3, :3] @ eye
        cams.append((R, K.copy(), np.linalg.inv(R), Kinv.copy()))
    return cams


def generate_ray(resy, resx, K_inverse, R_inverse, device="cpu"):
    """Mirror of reference captured_data.py:23-40; returns (origin [P,3], dir [P,3]) float64, contiguous."""
    K_inverse = torch.as_tensor(K_inverse, device=device, dtype=F64)
    R_inverse = torch.as_tensor(R_inverse, device=device, dtype=F64)
    y_range = torch.arange(0, resy, device=device, dtype=F64)
    x_range = torch.arange(0, resx, device=device, dtype=F64)
    pixely, pixelx = torch.meshgrid(y_range, x_range, indexing="ij")
    pixel = torch.stack([pixelx, pixely, torch.ones_like(pixely)], dim=2).view([-1, 3])
    pixel_p = K_inverse @ pixel.T
    world_p = R_inverse[:3, :3] @ pixel_p + R_inverse[:3, 3:4]
    ray_origin = R_inverse[:3, 3:4]
    ray_dir = (world_p - ray_origin).T
    ray_dir = ray_dir / ray_dir.norm(dim=1, keepdim=True)
    ray_origin = ray_origin.T.expand_as(ray_dir)
    return ray_origin.contiguous(), ray_dir.contiguous()


def process_mask(M):
    """Soft silhouette in [0,1] from a binary mask (reference captured_data.py:12-20; EDT = cv2 DIST_L2 precise)."""
    from scipy.ndimage import distance_transform_edt
    M = (np.asarray(M) > 0).astype(np.uint8)
    dist = distance_transform_edt(M).clip(0, 1) - (distance_transform_edt(1 - M) - 1).clip(0, 1)
    mask = (dist + 1) / 2
    mask[-1] = 0.5
    return mask


def screen_targets(out_ori, out_dir, mask, camera_M, center, extent, plane_factor=1.5):
    """Intersect exit rays with the background plane -> screen_pixel [P,3] (zeros where invalid)."""
    R = torch.as_tensor(camera_M[0], dtype=F64, device=out_ori.device)
    zc = R[2, :3]
    c = torch.as_tensor(center, dtype=F64, device=out_ori.device)
    p0 = c + plane_factor * extent * zc
    denom = out_dir @ zc
    ok = mask[:, 0].bool() & (denom > 1e-6)
    t = ((p0 - out_ori) @ zc) / torch.where(ok, denom, torch.ones_like(denom))
    sp = out_ori + t.unsqueeze(1) * out_dir
    # a target whose x is exactly 0 would read as invalid (captured_data.py:106); nudge it
    sp[:, 0] = torch.where(sp[:, 0] == 0, torch.full_like(sp[:, 0], 1e-12), sp[:, 0])
    return torch.where(ok.unsqueeze(1), sp, torch.zeros_like(sp))


def make_views(render_gt, hit_gt, center, extent, n_views, resx, resy, device="cpu", view_ids=None):
    """Build view tuples.

    render_gt(origin, ray_dir) -> (out_ori, out_dir, mask) traces the GROUND-TRUTH mesh;
    hit_gt(origin, ray_dir) -> bool [P] primary hit flags of the ground-truth mesh (for the
    silhouette mask).  Returns a list of
    (screen_pixel f64 [P,3], valid bool [P], mask f64 [P], origin f64 [P,3], ray_dir f64 [P,3], camera_M).
    """
    cams = turntable_cameras(center, extent, n_views, resx, resy)
    views = []
    ids = range(n_views) if view_ids is None else view_ids
    for k in ids:
        R, K, Rinv, Kinv = cams[k]
        origin, ray_dir = generate_ray(resy, resx, Kinv, Rinv, device=device)
        out_ori, out_dir, m = render_gt(origin, ray_dir)
        sp = screen_targets(out_ori.detach(), out_dir.detach(), m, cams[k], center, extent)
        valid = sp[:, 0] != 0
        hit = hit_gt(origin, ray_dir).view(resy, resx).cpu().numpy()
        soft = torch.as_tensor(process_mask(hit), dtype=F64, device=device).reshape(-1)
        camera_M = tuple(torch.as_tensor(a, dtype=F64, device=device) for a in (R, K, Rinv, Kinv))
        views.append((sp.contiguous(), valid.contiguous(), soft, origin, ray_dir, camera_M))
    return views


def mesh_frame(vertices):
    """(bbox centre, largest bbox extent) of a vertex array."""
    v = np.asarray(vertices, dtype=np.float64)
    lo, hi = v.min(0), v.max(0)
    return 0.5 * (lo + hi), float((hi - lo).max())


def displaced_ground_truth(mesh, sigma=0.5, seed=0):
    """Synthetic 'scan': the mesh displaced along its vertex normals by smooth-ish noise."""
    from . import mesh_io
    rng = np.random.default_rng(seed)
    vn = mesh_io.vertex_normals(mesh)
    disp = sigma * rng.standard_normal(len(mesh.vertices))
    v = (mesh.vertices + vn * disp[:, None]).astype(np.float32).astype(np.float64)
    return mesh_io.TriMesh(v, mesh.faces)
