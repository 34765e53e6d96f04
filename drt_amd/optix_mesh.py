"""Boundary B1: the reference's native tracer class, backed by the HIP LBVH.

Same four members as the pybind11 class ``optix_mesh`` of the reference
(optix_extend.cpp:77-83):

    optix_mesh(cuda_device)                      optix_extend.cpp:8-12
    update_mesh(F int32 [F,3], V float32 [V,3])  optix_extend.cpp:14-21
    update_vert(V float32 [V,3])                 optix_extend.cpp:23-27
    intersect(Ray float32 [N,6]) -> [T, ID]      optix_extend.cpp:29-57

Differences, all deliberate: inputs are validated (the reference only has C asserts
and misreads non-contiguous tensors), work is enqueued on torch's current stream
without a host sync (the reference's ``execute(0)`` is synchronous), and the returned
``T`` / ``ID`` are owning contiguous tensors (the reference returns strided aliases of
one buffer, ``ID`` through a non-owning ``from_blob``).  A miss has ``T = -1``,
``ID = -1``; callers test ``T > 0`` (reference DiffRender.py:391).
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


# The host side of an iteration is a chain of ~100 small calls, and torch's public accessors are Python: `torch.cuda.current_stream()` builds a
# Stream object (10 us), `with _on(d):` resolves its argument and exchanges the device twice (5-10 us) -- together a quarter of
# the host time of the reference-shaped iteration (tools/vh_host_profile.py).  The raw accessors behind them are used where they exist.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """hipStream_t of torch's current stream on the current device, as a ctypes pointer."""
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _NoContext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_CONTEXT = _NoContext()


def _on(device):
    """``with _on(device):`` -- torch.cuda.device(device), or nothing at all when that device is the current one already (the usual case)."""
    idx = device if isinstance(device, int) else getattr(device, "index", None)
    if idx is not None and _raw_device is not None and _raw_device() == idx:
        return _NO_CONTEXT
    return torch.cuda.device(device)


def _require(t, dtype, cols, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    if t.dim() != 2 or t.size(1) != cols:
        raise RuntimeError(f"{name} must have shape [N,{cols}], got {tuple(t.shape)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (there is no CPU tracer in the product path)")
    return t.contiguous()


class optix_mesh:
    def __init__(self, cuda_device=0):
        if not torch.cuda.is_available():
            raise _lib.DrtError("no GPU visible: drt_amd needs an MI355X (gfx950) device")
        self.device = int(cuda_device)
        h = ctypes.c_void_p()
        with _on(self.device):
            _lib.check(_lib.lib().drt_create(self.device, ctypes.byref(h)))
        self._h = h
        self.builded = False
        self.n_faces = 0
        self.n_verts = 0

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().drt_destroy(h)
            except Exception:
                pass
            self._h = None

    def _check_device(self, t, name):
        if t.device.index != self.device:
            raise RuntimeError(f"{name} is on {t.device}, this tracer is bound to cuda:{self.device}")

    def update_mesh(self, F, V):
        F = _require(F, torch.int32, 3, "F")
        V = _require(V, torch.float32, 3, "V")
        self._check_device(F, "F")
        self._check_device(V, "V")
        with _on(self.device):
            _lib.check(_lib.lib().drt_update_mesh(self._h, F.data_ptr(), F.size(0), V.data_ptr(), V.size(0), _stream()))
        self.n_faces, self.n_verts = F.size(0), V.size(0)
        self.builded = True

    def update_vert(self, V):
        assert self.builded, "update_mesh must be called first"
        V = _require(V, torch.float32, 3, "V")
        self._check_device(V, "V")
        with _on(self.device):
            _lib.check(_lib.lib().drt_update_vert(self._h, V.data_ptr(), V.size(0), _stream()))

    def update_vert_f64(self, V):
        """Fused ``V.detach().to(float32)`` + update_vert (reference DiffRender.py:379-380)."""
        assert self.builded, "update_mesh must be called first"
        V = _require(V.detach(), torch.float64, 3, "V")
        self._check_device(V, "V")
        with _on(self.device):
            _lib.check(_lib.lib().drt_update_vert_f64(self._h, V.data_ptr(), V.size(0), _stream()))

    def intersect(self, Ray):
        assert self.builded, "update_mesh must be called first"
        Ray = _require(Ray, torch.float32, 6, "Ray")
        self._check_device(Ray, "Ray")
        n = Ray.size(0)
        T = torch.empty(n, dtype=torch.float32, device=Ray.device)
        ID = torch.empty(n, dtype=torch.int32, device=Ray.device)
        with _on(self.device):
            _lib.check(_lib.lib().drt_intersect(self._h, Ray.data_ptr(), n, T.data_ptr(), ID.data_ptr(), _stream()))
        return [T, ID]

    # ---- additions beyond the reference class -------------------------------------------
    def intersect_any(self, Ray):
        """Hit flags only (bool [N]); what the occlusion / silhouette callers need."""
        assert self.builded, "update_mesh must be called first"
        Ray = _require(Ray, torch.float32, 6, "Ray")
        self._check_device(Ray, "Ray")
        n = Ray.size(0)
        hit = torch.empty(n, dtype=torch.uint8, device=Ray.device)
        with _on(self.device):
            _lib.check(_lib.lib().drt_intersect_any(self._h, Ray.data_ptr(), n, hit.data_ptr(), _stream()))
        return hit.view(torch.bool)

    def intersect_bruteforce(self, Ray):
        """Same contract as intersect by testing every triangle (diagnostic)."""
        Ray = _require(Ray, torch.float32, 6, "Ray")
        n = Ray.size(0)
        T = torch.empty(n, dtype=torch.float32, device=Ray.device)
        ID = torch.empty(n, dtype=torch.int32, device=Ray.device)
        with _on(self.device):
            _lib.check(_lib.lib().drt_intersect_bruteforce(self._h, Ray.data_ptr(), n, T.data_ptr(), ID.data_ptr(), _stream()))
        return [T, ID]

    def closest_point(self, points, want_face=True, want_point=False):
        """Distance from each point (float64 [N,3]) to the surface: dist float64 [N] (+ face int32 [N], closest float64 [N,3])."""
        assert self.builded, "update_mesh must be called first"
        points = _require(points, torch.float64, 3, "points")
        self._check_device(points, "points")
        n = points.size(0)
        dist = torch.empty(n, dtype=torch.float64, device=points.device)
        face = torch.empty(n, dtype=torch.int32, device=points.device) if want_face else None
        closest = torch.empty(n, 3, dtype=torch.float64, device=points.device) if want_point else None
        with _on(self.device):
            _lib.check(_lib.lib().drt_closest_point(self._h, points.data_ptr(), n, dist.data_ptr(), face.data_ptr() if want_face else None,
                                                    closest.data_ptr() if want_point else None, _stream()))
        return dist, face, closest

    def check(self):
        """(number of BVH containment/link violations, binary tree height); synchronises.  ``self.wide_depth`` = depth of
        the 4-wide tree the traversal walks (a violation is counted when 3 x depth exceeds the traversal stack)."""
        v = ctypes.c_int64()
        hgt = ctypes.c_int32()
        wide = ctypes.c_int32()
        with _on(self.device):
            _lib.check(_lib.lib().drt_bvh_check(self._h, _stream(), ctypes.byref(v), ctypes.byref(hgt), ctypes.byref(wide)))
        self.wide_depth = wide.value
        return v.value, hgt.value

    def tree_mode(self, mode, rebuild_every=1):
        """0: full LBVH build per update (default); 1: topology kept for ``rebuild_every`` updates (refit in between); 2: binned-SAH topology
        from the host at every update_mesh, refit per update_vert (include/drt_hip.h: drt_tree_mode)."""
        _lib.check(_lib.lib().drt_tree_mode(self._h, int(mode), int(rebuild_every)))

    def build_params(self):
        """(lo[3], 1/extent[3], leaf padding) of the scene box the last build derived from the vertices; synchronises."""
        out = (ctypes.c_float * 7)()
        with _on(self.device):
            _lib.check(_lib.lib().drt_build_params(self._h, out, _stream()))
        return list(out)

    def sorted_faces(self):
        out = torch.empty(self.n_faces, dtype=torch.int32, device=f"cuda:{self.device}")
        with _on(self.device):
            _lib.check(_lib.lib().drt_bvh_sorted_faces(self._h, out.data_ptr(), _stream()))
        return out

    STAGES = ("build", "cull", "trace1", "shade1", "trace2", "shade2", "trace3", "finish", "collect", "backward", "loss_bwd_fused", "raster", "fill", "path")

    def profile_enable(self, on=1):
        """1: bracket every pipeline kernel with hipEvents on its launch stream (bench.py's live timing);
        2: also collect traversal statistics (perturbs timing); 3: timers with the pipelines serialised on one internal
        stream (each kernel timed alone); 0: off."""
        with _on(self.device):
            _lib.check(_lib.lib().drt_profile_enable(self._h, int(on)))

    def profile_select(self, stages=None):
        """Time only the named stages (``STAGES`` entries) while the profile is on; None: all of them."""
        mask = 0xFFFFFFFF if stages is None else sum(1 << self.STAGES.index(k) for k in stages)
        _lib.check(_lib.lib().drt_profile_select(self._h, mask))

    def profile_read(self):
        """{stage: (total_ms, launches, items)} since the previous read; synchronises the stream."""
        n = len(self.STAGES)
        ms = (ctypes.c_double * n)()
        launches = (ctypes.c_int64 * n)()
        items = (ctypes.c_int64 * n)()
        with _on(self.device):
            _lib.check(_lib.lib().drt_profile_read(self._h, ms, launches, items))
        return {k: (ms[i], launches[i], items[i]) for i, k in enumerate(self.STAGES)}

    def trace_stats(self):
        """Per k_trace stage: (wave_steps, lane_steps, refills, max_wave_steps) of the last profile_read interval."""
        out = (ctypes.c_int64 * 12)()
        _lib.check(_lib.lib().drt_profile_trace_stats(self._h, out))
        return {f"trace{k + 1}": tuple(out[4 * k:4 * k + 4]) for k in range(3)}
