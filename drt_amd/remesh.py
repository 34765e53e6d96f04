"""Remeshing between optimisation passes -- host-side mirror of the reference's ``Meshlabserver``.

    Meshlabserver().remesh(scene, remesh_len)          reference optim.py:12-52

The reference writes the current mesh to a PLY, runs meshlabserver with the filter "Remeshing: Isotropic
Explicit Remeshing" (3 iterations, TargetLen = remesh_len, CheckSurfDist / MaxSurfDist 1, all five steps
on) and reloads the result with ``scene.update_mesh``.  Here the same algorithm (Botsch & Kobbelt 2004)
runs in-process in the native library (``drt_remesh_isotropic``, drt_amd/csrc/drt_remesh.cpp); the
positions are rounded through float32 like the reference's PLY round trip.  MeshLab's own vertex order and
tie-breaking are not reproducible without MeshLab, so this is behavioural, not bit, parity.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib, mesh_io

SPLIT, COLLAPSE, FLIP, SMOOTH, REPROJECT, CHECK_DIST, ALL = 1, 2, 4, 8, 16, 32, 63


def isotropic_remesh(mesh, target_len, iterations=3, max_surf_dist=1.0, flags=ALL, return_stats=False):
    """TriMesh -> TriMesh with edge lengths around ``target_len`` (closed manifold in, closed manifold out)."""
    V = np.ascontiguousarray(mesh.vertices, dtype=np.float64)
    F = np.ascontiguousarray(mesh.faces, dtype=np.int32)
    if not mesh.is_watertight:
        raise ValueError("isotropic_remesh needs a watertight mesh (the reference asserts it on load, DiffRender.py:305)")
    lib = _lib.lib()
    buf = ctypes.c_void_p()
    rc = lib.drt_remesh_isotropic(V.ctypes.data, len(V), F.ctypes.data, len(F), float(target_len), int(iterations),
                                  float(max_surf_dist), int(flags), ctypes.byref(buf))
    if rc != 0:
        raise _lib.DrtError(f"drt_remesh_isotropic failed with code {rc} (bad mesh or target length)")
    try:
        nv, nf = ctypes.c_int64(), ctypes.c_int64()
        stats = (ctypes.c_int64 * 4)()
        lib.drt_mesh_buf_size(buf, ctypes.byref(nv), ctypes.byref(nf), stats)
        outV = np.empty((nv.value, 3), dtype=np.float64)
        outF = np.empty((nf.value, 3), dtype=np.int32)
        lib.drt_mesh_buf_copy(buf, outV.ctypes.data, outF.ctypes.data)
    finally:
        lib.drt_mesh_buf_free(buf)
    out = mesh_io.TriMesh(outV.astype(np.float32).astype(np.float64), outF.astype(np.int64))
    if return_stats:
        return out, {"split": stats[0], "collapsed": stats[1], "flipped": stats[2], "iterations": stats[3]}
    return out


class Meshlabserver:
    """Same role and call shape as the reference class (optim.py:12-52), without the external program."""

    def __init__(self, iterations=3, max_surf_dist=1.0):
        self.iterations, self.max_surf_dist = iterations, max_surf_dist

    def remesh(self, scene, remesh_len):
        new_mesh = isotropic_remesh(scene.mesh, remesh_len, self.iterations, self.max_surf_dist)
        scene.update_mesh(new_mesh)
        return new_mesh
