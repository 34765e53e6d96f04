"""Boundary B2: the ``Scene`` surface of the reference's DiffRender.py, on the HIP path.

``import drt_amd.diffrender as Render`` is a drop-in for the reference's
``import DiffRender as Render`` (optim.py:6) for everything its optimisation loop
touches:

    Scene(mesh_path, cuda_device=0)                   DiffRender.py:298-301
    .update_mesh(path) / .update_verticex(vertices)   DiffRender.py:303-317, 378-384
    .render_transparent(origin, ray_dir)              DiffRender.py:420-432
    .optix_intersect(ray) / .render_mask(...)         DiffRender.py:386-392, 434-438
    .silhouette_edge / .primary_visibility            DiffRender.py:445-479
    .dihedral_angle() / .mean_len / .vertices / .mesh DiffRender.py:440-443, 345
    module globals intIOR, extIOR, resy, resx, device, Float   DiffRender.py:15-21

All per-ray and per-edge work runs in hand-written gfx950 kernels (libdrt_hip.so);
PyTorch only owns the tensors and the autograd plumbing.  Gradients reach
``vertices`` -- hidden in ``self`` by the reference -- through
``torch.autograd.Function``s that take it as an explicit input.
"""
from __future__ import annotations

import collections
import ctypes
import os
import warnings
import weakref

import numpy as np
import torch

from . import _lib, det, mesh_io
from ._util import _f64c, _flag_bytes, _stats, _warn_once, _warned  # noqa: F401
from .optix_mesh import optix_mesh, _stream, _on
from .stepwise import Intersection, StepwiseMixin  # noqa: F401  (Dintersect / refract_ray / trace2 / project_vert)
from .silhouette import (LazyColumn, LazyDiff, LazyGather, LazyIndex, LazyOutput, LazySum, LazyTerm, SampleSet, SilhouetteEdges,  # noqa: F401
                         _Dihedral, _EdgeSample, _SmLossFused, _VhLossFused, _VhTermLazy, force, pack_camera)

def cache_report(reset=False):
    """Counters of the transparent caches of this module: grid verdict cache (`grid_trust` calls that relied on a verdict, `grid_establish`
    calls that read and verified every ray, `grid_off` calls without whole-image hints / with GRID_CACHE off, `grid_unattachable` ray tensors
    that take no attributes), output recycling (`recycle_take` calls that rendered into pooled buffers; `recycle_miss_held` the pool had an
    entry of that size but the caller still holds, or wrote, its outputs; `recycle_miss_empty` nothing pooled yet; `recycle_off_capture` inside
    a graph capture; `recycle_off_small` below RECYCLE_MIN_RAYS), hit seeds (`seed_armed`), and the switches in force."""
    out = dict(_stats)
    out["switches"] = {"GRID_CACHE": GRID_CACHE, "RECYCLE_OUTPUTS": RECYCLE_OUTPUTS, "storage_use_count_available": hasattr(torch._C, "_storage_Use_Count"),
                       "RECYCLE_MIN_RAYS": RECYCLE_MIN_RAYS, "PREFILL_NEXT": PREFILL_NEXT, "HIT_SEED": HIT_SEED, "GRID_CANARY": GRID_CANARY}
    if reset:
        _stats.clear()
    return out


debug = False
resy = 960
resx = 1280
Float = torch.float64
device = "cuda"
extIOR, intIOR = 1.00029, 1.5


class Ray:
    """Ray bundle record (reference DiffRender.py:269-283); ray_ind defaults to arange."""

    def __init__(self, origin, direction, ray_ind=None):
        self.origin = origin
        self.direction = direction
        self.ray_ind = torch.arange(len(origin), device=origin.device) if ray_ind is None else ray_ind
        assert len(self.direction) == len(self.ray_ind)

    def select(self, mask):
        return Ray(self.origin[mask], self.direction[mask], self.ray_ind[mask])

    def __len__(self):
        return len(self.ray_ind)


def _tile_hint(n_rays):
    """(image width, image height) for the pipeline when the ray list is whole images of the module's resx x resy (as
    produced by generate_ray, reference captured_data.py:23-40; the module globals are what optim.py:180-181 sets):
    16x4-pixel tile ray grouping and projected primary visibility.  A hint only -- the library checks every ray against
    the grid it fits and falls back to the tree; (0, 0) = no assumption."""
    if resx >= 64 and resx % 64 == 0 and resy % 4 == 0 and n_rays % (resx * resy) == 0:
        return int(resx), int(resy)
    return 0, 0


DENSE_FACE_IDS = False  # True: Scene.last_face1 / last_face2 hold -1 for every ray without a hit (diagnostics, tests); False: only the entries of
                        # the rays with mask = 1 are defined (all the backward needs), which saves filling 8 bytes per ray
GRID_CACHE = True      # remember, per (origin, ray_dir) tensor pair, that its images verified as pinhole grids (see _grid_cache)
_GRID_BYTES = 104      # DRT_GRID_CACHE_BYTES of include/drt_hip.h


# Temporal hit seeds (include/drt_hip.h drt_render_seed): a ray tensor that comes back (TRUST mode of the grid cache below) carries an
# int32 [N] buffer with the face each pixel's refracted ray left the object through in the previous call on it; the traversal of the
# refracted rays starts from that triangle's distance.  Bit-identical results with any buffer content; 4 B per ray next to the 96 B of the
# ray itself.  Created outside graph captures only (a captured fill would reset the seeds at every replay).
# OFF by default (round 5, measured on MI355X, 72 x 1024^2): right seeds take 5 % of the node visits and 6 % of the vector instructions off
# the traversal of the refracted rays (0.394 -> 0.368 ms per launch with the GPU to itself) -- and nothing off the step (1.80 vs 1.82 ms,
# 7.08 vs 7.10 with the object filling the image), while the launch moves 55 % more HBM bytes (a 4-byte gather and scatter per ray into a
# 300 MB array).  A refracted ray starts ON the surface: its visits are the boxes around its origin and along the chord to the exit point,
# which no distance bound removes.  DRT_HIT_SEED=1 / HIT_SEED = True turns them on (tests/test_gpu_seed.py runs with them on).
HIT_SEED = os.environ.get("DRT_HIT_SEED", "0") != "0"
GRID_CANARY = os.environ.get("DRT_GRID_CANARY", "1") != "0"      # (read by the library itself at scene creation; here for cache_report)


def _hit_seed(ray_dir, n):
    if not HIT_SEED:
        return None
    seed = getattr(ray_dir, "_drt_seed2", None)
    if seed is not None and seed.shape[0] == n and seed.device == ray_dir.device:
        return seed
    if torch.cuda.is_current_stream_capturing():
        return None
    seed = torch.full((n,), -1, dtype=torch.int32, device=ray_dir.device)
    try:
        ray_dir._drt_seed2 = seed
    except Exception:
        return None
    return seed


def _arm_seed(handle, grid, n):
    """Registers the seeds of this call's ray tensor (third entry of _grid_cache's result) with the library: consumed by the next render call."""
    seed = grid[2] if len(grid) > 2 else None
    if seed is not None:
        _stats["seed_armed"] += 1
        _lib.check(_lib.lib().drt_render_seed(handle, seed.data_ptr(), n))


def _grid_cache(origin, ray_dir, n, w, h):
    """(grid_mode, cache tensor or None, hit seeds or None) for a render call on these ray tensors.

    The views of a capture are constants of the optimisation: the same tensor objects come back every iteration.  The
    first call on a pair ESTABLISHES, on the device, which of its images are pinhole ray grids in every single ray
    (drt_raster.h); the record -- a small device buffer -- is kept on the ``ray_dir`` tensor object together with the
    identity of ``origin`` and both tensors' in-place version counters, and later calls TRUST it as long as those
    match: the library then does not re-read the rays of pixels nothing projects onto.  Any in-place write to either
    tensor bumps its version and the next call re-establishes.  ``t.data = ...`` and writes through raw pointers (a DLPack / NumPy
    export of the tensor's memory, a kernel of the caller's own) are not seen by the version counter; what catches them is on the device:
    every trusted call re-verifies, per image, the 8 x 8 lattice of rays its model was fitted on AND 64 more rays at pixels that change from
    call to call (k_check_views' canary), and an image that fails is verified ray by ray from then on.  A partial overwrite of a fraction f
    of an image's rays therefore survives a call with probability (1 - f)^64; a caller who writes single rays behind autograd's back must
    set GRID_CACHE = False (or bump the version: ``t.add_(0)``)."""
    if not GRID_CACHE or w <= 0 or h <= 0 or n == 0:
        _stats["grid_off"] += 1
        return 0, None
    rec = getattr(ray_dir, "_drt_grid", None)
    key = (n, w, h, origin._version, ray_dir._version, origin.data_ptr(), ray_dir.data_ptr())
    if rec is not None and rec[0] == key and rec[1]() is origin:
        if rec[3][0] is None and not torch.cuda.is_current_stream_capturing():
            # first call after the establishing one (not while a graph is being captured: the read-back synchronises): read the verdicts back once (one host sync per ray tensor, not per step).
            # All images verified in all rays -> DRT_GRID_ALL_VERIFIED: the launches that serve other rays are not issued.
            flags = rec[2].view(-1, _GRID_BYTES)[:, 96:104].contiguous().view(torch.int32)
            rec[3][0] = bool((flags != 0).all().item())
        _stats["grid_trust"] += 1
        return 2 | (32 if rec[3][0] else 0), rec[2], _hit_seed(ray_dir, n)
    cache = torch.zeros((n // (w * h)) * _GRID_BYTES, dtype=torch.uint8, device=ray_dir.device)
    try:
        ray_dir._drt_grid = (key, weakref.ref(origin), cache, [None])
    except Exception:           # a tensor that takes no attributes: no cache
        _stats["grid_unattachable"] += 1
        _warn_once("grid_unattachable", "the ray_dir tensor takes no attributes: no grid verdict cache for it (every call verifies every ray)")
        return 0, None
    _stats["grid_establish"] += 1
    return 1, cache


class RayBinding:
    """An explicit handle for the rays of a capture (round 6): ``handle = scene.bind_rays(origin, ray_dir[, screen_pixel, valid])``, then
    ``scene.render_transparent(handle)`` every iteration.  The reference hands its loop fresh copies of constant tensors per call
    (captured_data.py:44-59); the drop-in signature therefore has to RECOGNISE constants (tensor identity, version counters, storage use
    counts: `_grid_cache`, `_OutputPool` below).  A caller that can say so instead gets the same fast path by CONTRACT:

      * the bound tensors are constants -- whether their images are pinhole grids is established on the device by the first call and
        trusted afterwards (still re-checked per call on the 8 x 8 lattice + 64 canary rays per image; an in-place write bumps the version
        counter and re-establishes);
      * the handle OWNS one set of dense outputs: the tensors a call returns -- and the autograd graph behind them -- are valid until the NEXT
        ``render_transparent(handle)``; that call zeroes the rows this one set (55 B per completed path instead of a 51 B-per-ray fill)
        and renders into the same memory.  ``backward()`` of a call whose outputs were recycled raises.  No storage use counts, no
        private torch API, works the same inside a graph capture (replay = recycle);
      * targets bound with the rays are known to be complete before any render call: the loss + gradient pass may start on the first
        sub-batch while the second is still tracing (SPLIT_LOSS) without the identity bookkeeping of `_targets_seen_before`."""

    class Shared:
        """The output set(s) and call counter of one handle -- or of several handles that agree to share them (``shared=``: the views of one
        capture, rendered one per iteration: a call on ANY of them recycles the outputs of the previous call on any of them)."""

        def __init__(self):
            self.sets, self.gen = {}, 0

    def __init__(self, scene, origin, ray_dir, screen_pixel=None, valid=None, shared=None):
        self.scene = scene
        self._shared = shared if shared is not None else RayBinding.Shared()
        self.origin, self.ray_dir = _f64c(origin.detach(), "origin"), _f64c(ray_dir.detach(), "ray_dir")
        if self.origin.shape != self.ray_dir.shape or self.origin.dim() != 2 or self.origin.shape[1] != 3:
            raise RuntimeError("bind_rays: origin and ray_dir must both be float64 [N, 3]")
        self.n = self.origin.shape[0]
        self.targets = (screen_pixel, valid) if screen_pixel is not None else None
        self._cache = self._verdict = self._versions = self._seed = self._hint = None

    def _grid(self):
        """(grid_mode, verdict cache, hit seeds) of the next render call on the bound rays."""
        w, h = _tile_hint(self.n)
        versions = (self.origin._version, self.ray_dir._version, w, h)
        if not GRID_CACHE or w <= 0 or self.n == 0:
            return 0, None
        if self._cache is None or versions != self._versions:
            self._cache = torch.zeros((self.n // (w * h)) * _GRID_BYTES, dtype=torch.uint8, device=self.ray_dir.device)
            self._versions, self._verdict = versions, None
            _stats["grid_establish"] += 1
            return 1, self._cache
        if self._verdict is None and not torch.cuda.is_current_stream_capturing():
            flags = self._cache.view(-1, _GRID_BYTES)[:, 96:104].contiguous().view(torch.int32)
            self._verdict = bool((flags != 0).all().item())          # one host sync per handle, ever
        _stats["grid_trust"] += 1
        if HIT_SEED and self._seed is None and not torch.cuda.is_current_stream_capturing():
            self._seed = torch.full((self.n,), -1, dtype=torch.int32, device=self.ray_dir.device)
        return 2 | (32 if self._verdict else 0), self._cache, (self._seed if HIT_SEED else None)

    def _outputs(self):
        """The handle's output set, in the layout of an _OutputPool entry; zero-filled once."""
        dev, sets = self.origin.device, self._shared.sets
        ent = sets.get((self.n, dev))
        if ent is None:
            n = self.n
            bases = (torch.zeros((n, 3), dtype=torch.float64, device=dev), torch.zeros((n, 3), dtype=torch.float64, device=dev),
                     torch.zeros((n, 3), dtype=torch.uint8, device=dev))
            ent = sets[(n, dev)] = [n, dev, None, bases, None, torch.empty(n, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)]
        return ent

    @property
    def gen(self):
        """render calls made on this handle (or the handles it shares outputs with): what the autograd graph of an EARLIER call checks in its backward"""
        return self._shared.gen

    def release(self):
        self._shared.sets.clear()
        self._shared.gen += 1


# ray_loss's gradient w.r.t. out_dir is zero in all but a few per cent of the rows.  When the out_dir handed to
# ``ray_loss`` is the very tensor ``render_transparent`` returned, the two autograd nodes exchange that gradient as a ROW
# LIST instead of a dense float64 [N,3] tensor (1.8 GB per 72 x 1024^2 step, written once and read once): ray_loss's
# backward queues (rows, targets, scale) on the link and returns a stride-0 zero tensor; render_transparent's backward
# adds the queued rows (drt_render_backward_ray_loss) to whatever dense gradient other consumers of out_dir produced.
# The values are identical.  What it cannot serve is a caller who asks autograd for d loss / d out_dir ITSELF
# (torch.autograd.grad(loss, out_dir), out_dir.retain_grad()): set SPARSE_LOSS_GRAD = False for that.
SPARSE_LOSS_GRAD = True
# On top of that: when ray_loss is given render_transparent's own, untouched (out_ori, out_dir, mask), its forward pass already computes
# d loss / d vertices with a unit seed next to the loss (ONE pass over the completed paths: drt_ray_loss_listed_grad) and the backward pass
# only scales that stash by the incoming gradient -- instead of a loss pass now and a second pass over the same paths (recompute,
# adjoint, scatter) in render_transparent's backward.  Costs the gradient's work to a caller who needs grad mode on but never calls
# backward(); set EAGER_LOSS_GRAD = False for that.
EAGER_LOSS_GRAD = True
# The dense outputs are zeros in all but a few per cent of the rows, and writing those zeros (51 B per ray) is the one HBM-bound stage of a
# render_transparent call.  With PREFILL_NEXT a call in trusted-grid mode allocates the out_ori and mask of the NEXT call of the same size
# right away and has the library zero them on its idle stream (drt_prefill_zero) -- i.e. while the caller's loss, backward and optimiser
# kernels run, which leave the memory system idle -- instead of beside the next call's traversal.  The tensors a call returns are its own
# fresh allocations either way (nothing is ever handed out twice); the price is one extra out_ori + mask (27 B per ray) held between calls.
PREFILL_NEXT = os.environ.get("DRT_PREFILL_NEXT", "1") != "0"
PREFILL_MIN_RAYS = int(os.environ.get("DRT_PREFILL_MIN_RAYS", 1 << 25))          # below this the two extra allocations and calls cost the host more than the earlier fill saves the GPU (18 views x 1024^2: +3 %, 9 views: +4 %)


# RECYCLE_OUTPUTS goes further: a call's outputs are zeros wherever its mask is, and its list of completed paths names exactly the other
# rows.  The library keeps its own reference to the three buffers of a trusted-grid call (`_OutputPool`); once the caller has let go of
# them -- the storages' use counts are back to the pool's own, the version counters unmoved (an in-place write by the caller would
# show), same stream -- the NEXT call of that size zeroes the listed rows (drt_outputs_clean: 51 B per completed path instead of 51 B
# per ray) and renders into the same memory: no fill of the dense outputs at all (72 x 1024^2: 2.38 -> 1.8 ms per step).  What a call
# returns are fresh tensor objects (detached aliases: same storage, same version counter) that nothing else references; a caller who
# keeps its outputs alive simply gets fresh allocations and the fills, as before.  The price in memory: the usual loop rebinds
# `out = render()` only after the call has returned, so the previous outputs are still held when the pool is asked and the pool ping-pongs
# between TWO entries: two output sets plus their two row lists (2 x 55 B per ray) stay resident per scene (`Scene.release_outputs()`
# drops them).  Inside a graph capture a pooled set becomes the graph's own (replay = recycle: `graph_set` in _RenderTransparent.forward).
# (The use count of a storage is read through torch._C._storage_Use_Count, which torch's own CUDA-graph trees rely on; a torch without it
# simply does not recycle.  As with any caching allocator, a caller who used the outputs on ANOTHER stream must have ordered that work in front
# of the stream of its next render call before dropping them.)
RECYCLE_OUTPUTS = os.environ.get("DRT_RECYCLE_OUTPUTS", "1") != "0" and hasattr(torch._C, "_storage_Use_Count")
if os.environ.get("DRT_RECYCLE_OUTPUTS", "1") != "0" and not RECYCLE_OUTPUTS:
    warnings.warn("drt_amd.diffrender: this torch has no torch._C._storage_Use_Count -- render_transparent cannot tell when its previous outputs "
                  "were released and fills fresh ones every call (about a quarter slower at 72 x 1024^2); see cache_report()", RuntimeWarning)
RECYCLE_MIN_RAYS = int(os.environ.get("DRT_RECYCLE_MIN_RAYS", 1 << 22))


def _use_count(t):
    return torch._C._storage_Use_Count(t.untyped_storage()._cdata)


class _OutputPool:
    """The dense outputs of earlier trusted-grid calls, kept for re-use (see RECYCLE_OUTPUTS).  An entry:
    [n, device, stream, (out_ori, out_dir, mask) base tensors, their use counts with nobody else holding them, valid_idx, n_valid]."""
    MAX = 2

    def __init__(self):
        self.entries = []

    def take(self, n, device, stream):
        """An entry of this size whose buffers nobody else references or has written; removed from the pool."""
        for k, e in enumerate(self.entries):
            if e[0] == n and e[1] == device and (stream is None or e[2] == stream) and all(_use_count(t) == c and t._version == 0 for t, c in zip(e[3], e[4])):
                return self.entries.pop(k)
        return None

    def put(self, n, device, stream, bases, counts, valid_idx, n_valid):
        self.entries.append([n, device, stream, bases, counts, valid_idx, n_valid])
        if len(self.entries) > self.MAX:
            self.entries.pop(0)


# SPLIT_LOSS: the loss + gradient pass over the completed paths of the call's FIRST sub-batch starts on the internal stream that produced
# them, beside the tail of the other pipeline, instead of behind the join (drt_ray_loss_listed_grad_split).  For that its two accumulators
# must have been zeroed before the render call was enqueued (render_transparent creates them: `_GradLink.pre`), and the targets must have
# been complete by then -- known when the very same target tensors (object, storage, version) went through ray_loss before.
SPLIT_LOSS = os.environ.get("DRT_SPLIT_LOSS", "1") != "0"
SPLIT_LOSS_MIN_RAYS = int(os.environ.get("DRT_SPLIT_LOSS_MIN_RAYS", 1 << 25))          # below, a call is not cut into sub-batches (DRT_MIN_SUB_LOG2 = 24 per sub-batch); zeroing the accumulators
                                                                                 # in front of the render call anyway (two launches off the tail, two more in front of the fork) measured +1 % at 9 and 18 views
_seen_targets = {}
_render_seq = [0]          # render_transparent calls enqueued so far (any scene): a clock for "was complete before THAT call"


def _targets_seen_before(sp, valid, render_seq):
    """True when these very tensors (same objects, same storage, unchanged version counters) were the targets of a ray_loss that was
    issued BEFORE the render call number ``render_seq`` was enqueued: their content was then complete before that call's fork, which is all
    the internal stream is ordered behind.  (Seen in an earlier ray_loss is not enough: render A, render B, build the targets on the
    stream, ray_loss(A, tgt), ray_loss(B, tgt) -- the second loss would find them registered although they were produced behind B's fork.)
    Registers them either way, with the current clock."""
    ok = True
    for t in (sp, valid):
        rec = _seen_targets.get(id(t))
        if not (rec is not None and rec[0]() is t and rec[1] == t._version and rec[2] == t.data_ptr()):
            ok = False
            if len(_seen_targets) > 4096:
                _seen_targets.clear()
            _seen_targets[id(t)] = (weakref.ref(t), t._version, t.data_ptr(), _render_seq[0])
        elif rec[3] >= render_seq:
            ok = False
    return ok


class _GradLink:
    def __init__(self):
        self.pre = None        # (stash zeros_like(vertices), loss zeros(())) created before the render call was enqueued (SPLIT_LOSS)
        self.seq = 0           # _render_seq of the render call this link belongs to
        self.pending = []
        self._token = None
        self.paths = None
        self.mask = None
        self.out_ori = None
        self.render = None      # (scene handle owner, vertices, origin, ray_dir, face1, face2, (ior_int, ior_ext)) of the forward call
        self.targets = None     # (screen_pixel, valid) bound with the rays (RayBinding): complete before any render call on the handle

    def token(self, n, device):
        if self._token is None or self._token.shape[0] != n:
            self._token = torch.zeros((1, 3), dtype=torch.float64, device=device).expand(n, 3)
        return self._token

    def is_token(self, g):
        t = self._token
        return t is not None and g.data_ptr() == t.data_ptr() and g.stride() == t.stride() and g.shape == t.shape


class _RenderTransparent(torch.autograd.Function):
    """render_transparent as a function of the vertices (the reference's implicit input)."""

    @staticmethod
    def forward(ctx, vertices, origin, ray_dir, scene, ior_int, ior_ext, link, grid=(0, None), binding=None):
        ctx.link = link
        ctx.binding = binding
        v = _f64c(vertices.detach(), "vertices")
        o = _f64c(origin.detach(), "origin")
        d = _f64c(ray_dir.detach(), "ray_dir")
        n = o.shape[0]
        om = scene.optix_mesh            # owns the buffers zeroed ahead of time: its drt_destroy waits for the zeroing before they are released
        capturing = torch.cuda.is_current_stream_capturing()
        need_bwd = ctx.needs_input_grad[0]
        recycle = binding is not None or (RECYCLE_OUTPUTS and n >= RECYCLE_MIN_RAYS)  # (any grid mode: a call that verifies every ray -- no cache, or the
                                                                                  #  establishing one -- then at least does not write the dead rows again)
        # Inside a graph capture: REPLAY = RECYCLE.  A set the eager warm-up calls left in the pool is taken out of it for good and becomes the
        # graph's static outputs; the captured call zeroes the rows of the set's row list and then writes ITS list into those very buffers, so
        # every replay undoes exactly what its predecessor set (the set's state at capture time is that of the eager call that filled it
        # last: consistent with its list, which is what the first replay undoes).  No pooled set -- no warm-up call of this size since the
        # last capture -- and the capture brings fresh outputs and their fills, as before.
        graph_set = None
        pre = getattr(om, "_prefilled", None)
        if capturing:
            # a graph replays THESE launches on THESE buffers: the fills must be part of it, and nothing outside the capture may be waited
            # for inside it -- the buffers zeroed ahead of time stay where they are until the next call outside a capture
            pre = None
        else:
            om._prefilled = None
        stream_id = _stream()
        bases = counts = None
        if not recycle and RECYCLE_OUTPUTS:
            _stats["recycle_off_small"] += 1
        ent = None
        if binding is not None:
            # the handle's own set, by contract (RayBinding): the same path a captured call takes with a pooled set -- this call's list goes where
            # its predecessor's was, the set never enters the pool
            graph_set = ent = binding._outputs()
            bases, counts = ent[3], ent[4]
            binding._shared.gen += 1
            ctx.binding_gen = binding.gen
            _stats["recycle_bound"] += 1
        elif recycle:
            pool = getattr(om, "_out_pool", None)
            if pool is None:
                pool = om._out_pool = _OutputPool()
            # (a capture runs on a stream of its own: any pooled set of this size will do -- the capture is ordered behind the warm-up)
            ent = pool.take(n, o.device, None if capturing else getattr(stream_id, "value", stream_id))
            if capturing:
                _stats["recycle_graph_set" if ent is not None else "recycle_off_capture"] += 1
                if ent is None:
                    recycle = False
                else:
                    graph_set = ent                      # (registered in om._graph_sets once the captured call has gone through)
            else:
                _stats["recycle_take" if ent is not None else ("recycle_miss_held" if any(e[0] == n for e in pool.entries) else "recycle_miss_empty")] += 1
            if ent is not None:
                # the outputs of an earlier call that nobody holds any more: the rows that call set are zeroed (drt_outputs_clean, registered
                # below, right in front of the render call and behind every allocation of this one) and the call renders into the same memory
                bases, counts = ent[3], ent[4]
        took = ent if recycle and bases is not None else None
        if bases is not None:
            if pre is not None:
                with _on(o.device):
                    _lib.check(_lib.lib().drt_prefill_wait(om._h, stream_id))
                pre = None
            out_ori, out_dir, mask = bases
        elif pre is not None and pre[0] == n and pre[1] == o.device:
            out_ori, mask = pre[2], pre[3]
            out_dir = torch.empty((n, 3), dtype=torch.float64, device=o.device)
        else:
            if pre is not None:          # another size: let the zeroing finish (on this stream's timeline) before the memory goes back to the allocator
                with _on(o.device):
                    _lib.check(_lib.lib().drt_prefill_wait(om._h, stream_id))
            out_ori = torch.empty((n, 3), dtype=torch.float64, device=o.device)
            mask = torch.empty((n, 3), dtype=torch.uint8, device=o.device)
            out_dir = torch.empty((n, 3), dtype=torch.float64, device=o.device)
        pre = None
        bases_in = bases
        if recycle and bases is None:
            bases = (out_ori, out_dir, mask)
            counts = tuple(_use_count(t) for t in bases)          # with nobody but these three names holding them
        face1 = torch.empty(n, dtype=torch.int32, device=o.device)
        face2 = torch.empty(n, dtype=torch.int32, device=o.device)
        if (SPLIT_LOSS and need_bwd and EAGER_LOSS_GRAD and link is not None and not capturing and n >= SPLIT_LOSS_MIN_RAYS):
            link.pre = (det.acc(v), det.scalar(o.device))
        _render_seq[0] += 1
        if link is not None:
            link.seq = _render_seq[0]
        want_list = need_bwd or recycle
        if graph_set is not None:
            valid_idx, n_valid = graph_set[5], graph_set[6]           # this call's list goes where its predecessor's was: see above
        else:
            valid_idx = torch.empty(n, dtype=torch.int32, device=o.device) if want_list else None
            n_valid = torch.empty(1, dtype=torch.int64, device=o.device) if want_list else None
        with _on(o.device):
            # The library is handed raw pointers of buffers only this frame keeps alive (`took`: the pooled outputs and their row list):
            # nothing that can raise sits between the registration and the call that consumes it, and a call that fails withdraws them
            # itself (drt_render_forward) -- a request left behind would have the next call zero "rows" of freed memory.
            try:
                if took is not None:
                    _lib.check(_lib.lib().drt_outputs_clean(om._h, bases_in[0].data_ptr(), bases_in[1].data_ptr(), bases_in[2].data_ptr(), n,
                                                            took[5].data_ptr(), took[6].data_ptr(), stream_id))
                _arm_seed(scene.optix_mesh._h, grid, n)
                _lib.check(_lib.lib().drt_render_forward(
                    scene.optix_mesh._h, v.data_ptr(), o.data_ptr(), d.data_ptr(), n, float(ior_int), float(ior_ext),
                    out_ori.data_ptr(), out_dir.data_ptr(), mask.data_ptr(), face1.data_ptr(), face2.data_ptr(),
                    _lib.ptr(valid_idx), _lib.ptr(n_valid), *_tile_hint(n), grid[0] | (0 if DENSE_FACE_IDS else 16), _lib.ptr(grid[1]), stream_id))
            except BaseException:
                _lib.lib().drt_outputs_cancel(om._h)
                if graph_set is not None and binding is None:      # the capture failed: the set goes back to the pool, the caller falls back to eager calls
                    om._out_pool.entries.append(graph_set)
                if binding is not None:
                    binding.release()                    # (whatever state the failed call left its set in: the next call starts from fresh zeros)
                raise
            if graph_set is not None and binding is None:
                sets = getattr(om, "_graph_sets", None)
                if sets is None:
                    sets = om._graph_sets = []
                sets.append(graph_set)                   # alive as long as the scene (release_outputs(graph_sets=True)): the graph's replays read and write these buffers
            # (with recycling on: only for a caller who evidently KEEPS its outputs -- the pool had nothing to offer for this call and for the one
            # before it; the steady state of a loop that drops them never gets here)
            missed, om._recycle_missed = getattr(om, "_recycle_missed", 0), (0 if took is not None or not recycle else getattr(om, "_recycle_missed", 0) + 1)
            if PREFILL_NEXT and (not recycle or (took is None and missed >= 2)) and (grid[0] & 3) == 2 and n >= PREFILL_MIN_RAYS and not capturing and getattr(om, "_prefilled", None) is None:
                # outputs of the next call of this size: allocated now, zeroed on the library's idle stream behind this forward pass
                h = scene.optix_mesh._h
                nxt_ori = torch.empty((n, 3), dtype=torch.float64, device=o.device)
                nxt_mask = torch.empty((n, 3), dtype=torch.uint8, device=o.device)
                _lib.check(_lib.lib().drt_prefill_zero(h, nxt_ori.data_ptr(), nxt_ori.numel() * 8, stream_id))
                _lib.check(_lib.lib().drt_prefill_zero(h, nxt_mask.data_ptr(), nxt_mask.numel(), stream_id))
                om._prefilled = (n, o.device, nxt_ori, nxt_mask)
        if recycle:
            # the pool keeps the base tensors; the caller gets aliases of its own (same storage, same version counter), so that the
            # storages' use counts say when the caller is done with them  (a graph's set stays out of the pool: `om._graph_sets`)
            if graph_set is None:
                om._out_pool.put(n, o.device, getattr(stream_id, "value", stream_id), bases, counts, valid_idx, n_valid)
            out_ori, out_dir, mask = bases[0].detach(), bases[1].detach(), bases[2].detach()
            bases = None
        if not need_bwd:
            valid_idx = n_valid = None
        ctx.scene = scene
        ctx.ior = (float(ior_int), float(ior_ext))
        ctx.save_for_backward(v, o, d, face1, face2, valid_idx, n_valid)
        if link is not None:
            link.paths = (valid_idx, n_valid)        # the rays with mask = 1: ray_loss walks this list instead of all N rays
            link.render = (scene, v, o, d, face1, face2, ctx.ior) if need_bwd else None
        # an output the loss does not use must reach backward() as None, not as a materialised [N,3] float64 zero tensor:
        # the reference's ray_loss detaches out_ori (optim.py:100), and filling 1.8 GB of zeros per step cost 0.3 ms
        ctx.set_materialize_grads(False)
        mask_b = mask.view(torch.bool)
        ctx.mark_non_differentiable(mask_b)
        scene.last_face1, scene.last_face2 = face1, face2
        return out_ori, out_dir, mask_b

    @staticmethod
    def backward(ctx, g_ori, g_dir, g_mask):
        v, o, d, face1, face2, valid_idx, n_valid = ctx.saved_tensors
        link = ctx.link
        if ctx.binding is not None and ctx.binding.gen != ctx.binding_gen and (g_ori is not None or (g_dir is not None and not link.is_token(g_dir)) or any(e[0] is not None for e in link.pending)):
            raise RuntimeError("render_transparent(binding): the outputs of this call (and the list of its completed paths) were recycled by a later "
                               "call on the same RayBinding -- run backward() before the next render_transparent(handle), or render without a handle")
        pending, link.pending = link.pending, []
        if g_dir is not None and link.is_token(g_dir):
            g_dir = None                        # ray_loss's placeholder: its gradient is in `pending`
        # (the usual step -- one ray_loss, eager stash, nothing dense -- is ONE small launch: stash * scale)
        grad_v = None if (g_ori is None and g_dir is None and pending and all(e[0] is None for e in pending)) else det.acc(v)
        h = ctx.scene.optix_mesh._h
        with _on(o.device):
            if g_ori is not None or g_dir is not None:
                g_ori = None if g_ori is None else _f64c(g_ori, "grad_out_ori")
                g_dir = None if g_dir is None else _f64c(g_dir, "grad_out_dir")
                _lib.check(_lib.lib().drt_render_backward(
                    h, v.data_ptr(), o.data_ptr(), d.data_ptr(), o.shape[0], ctx.ior[0], ctx.ior[1],
                    face1.data_ptr(), face2.data_ptr(), _lib.ptr(valid_idx), _lib.ptr(n_valid),
                    _lib.ptr(g_ori), _lib.ptr(g_dir), grad_v.data_ptr(), _stream()))
            for rows, n_rows, sp, scale in pending:
                if rows is None:                # eager entry: n_rows is the unit-seed vertex gradient ray_loss's forward left
                    continue
                _lib.check(_lib.lib().drt_render_backward_ray_loss(
                    h, v.data_ptr(), o.data_ptr(), d.data_ptr(), o.shape[0], ctx.ior[0], ctx.ior[1], face1.data_ptr(), face2.data_ptr(),
                    rows.data_ptr(), n_rows.data_ptr(), sp.data_ptr(), scale.data_ptr(), grad_v.data_ptr(), _stream()))
        if grad_v is not None:
            grad_v = det.value(grad_v, v)          # (deterministic mode: the exact integer sums, rounded once)
        for rows, stash, wide, scale in pending:
            if rows is None:
                if det.SINK is not None and wide is not None:
                    det.SINK.append((wide, scale))        # full_batch_step sums the cells of all calls and ranks exactly, then converts once
                    continue
                grad_v = torch.addcmul(grad_v, stash, scale) if grad_v is not None else stash * scale
        return grad_v, None, None, None, None, None, None, None, None


class _RayLoss(torch.autograd.Function):
    """Loss_calculator.ray_loss (reference optim.py:100-106) in one pass over the rays."""

    @staticmethod
    def forward(ctx, out_ori, out_dir, mask, screen_pixel, valid, link):
        oo = _f64c(out_ori.detach(), "out_ori")
        od = _f64c(out_dir.detach(), "out_dir")
        sp = _f64c(screen_pixel, "screen_pixel")
        n = oo.shape[0]
        m = _flag_bytes(mask, "mask", 3 * n)
        va = _flag_bytes(valid, "valid", n)
        loss = det.scalar(oo.device)
        need = ctx.needs_input_grad[1]
        ctx.link = link if need else None
        g = torch.empty_like(od) if need and link is None else None      # dense d loss / d out_dir only without a link
        ctx.stash = ctx.stash_wide = None
        own = (link is not None and link.paths is not None and link.paths[0] is not None and link.mask is not None
               and link.mask() is mask and mask._version == 0)           # the forward's own mask, untouched: its list of set rows is exact
        eager = (own and need and EAGER_LOSS_GRAD and link.render is not None and link.out_ori is not None and link.out_ori() is out_ori
                 and out_ori._version == 0 and out_dir._version == 0)
        rows = torch.empty(n, dtype=torch.int32, device=oo.device) if need and not eager else None     # (the eager form needs no row list)
        n_rows = torch.zeros(1, dtype=torch.int32, device=oo.device) if need and not eager else None
        with _on(oo.device):
            if eager:
                # loss + unit-seed vertex gradient in one pass over the completed paths (see EAGER_LOSS_GRAD)
                scene, v, o, d, face1, face2, ior = link.render
                pre, link.pre = link.pre, None
                bound = link.targets is not None and link.targets[0] is screen_pixel and link.targets[1] is valid
                early = pre is not None and (bound or _targets_seen_before(screen_pixel, valid, link.seq)) and sp.data_ptr() == screen_pixel.data_ptr() and va.data_ptr() == valid.data_ptr()
                if early:      # accumulators zeroed before the render call: the head of the list can be processed beside the pipelines (SPLIT_LOSS)
                    ctx.stash, loss = pre
                else:
                    ctx.stash = det.acc(v)
                fn = _lib.lib().drt_ray_loss_listed_grad_split if early else _lib.lib().drt_ray_loss_listed_grad
                _lib.check(fn(
                    scene.optix_mesh._h, v.data_ptr(), o.data_ptr(), d.data_ptr(), n, ior[0], ior[1], face1.data_ptr(), face2.data_ptr(),
                    sp.data_ptr(), va.data_ptr(), link.paths[0].data_ptr(), link.paths[1].data_ptr(), loss.data_ptr(), ctx.stash.data_ptr(), _stream()))
                ctx.stash_wide = ctx.stash if ctx.stash.dtype == torch.int64 else None      # (deterministic mode: the cells themselves, for det.SINK)
                ctx.stash = det.value(ctx.stash, v)
            elif own:
                _lib.check(_lib.lib().drt_ray_loss_listed(oo.data_ptr(), od.data_ptr(), sp.data_ptr(), va.data_ptr(), link.paths[0].data_ptr(),
                                                          link.paths[1].data_ptr(), n, loss.data_ptr(), _lib.ptr(rows), _lib.ptr(n_rows), _stream()))
            else:
                _lib.check(_lib.lib().drt_ray_loss(oo.data_ptr(), od.data_ptr(), m.data_ptr(), sp.data_ptr(), va.data_ptr(), n,
                                                   loss.data_ptr(), _lib.ptr(g), _lib.ptr(rows), _lib.ptr(n_rows), _stream()))
        loss = det.value(loss)
        ctx.save_for_backward(g, rows, n_rows, sp if (need and link is not None) else None)
        ctx.applied = None                       # scale already multiplied into the saved rows (see backward)
        ctx.n_rays = n
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        g, rows, n_rows, sp = ctx.saved_tensors
        if ctx.link is not None:                 # row-list hand-off to render_transparent's backward (see _GradLink)
            if ctx.n_rays == 0:
                return None, None, None, None, None, None
            scale = g_loss.detach().to(torch.float64).reshape(1).contiguous()
            if ctx.stash is not None:
                ctx.link.pending.append((None, ctx.stash, ctx.stash_wide, scale))
                return None, ctx.link.token(ctx.n_rays, ctx.stash.device), None, None, None, None
            ctx.link.pending.append((rows, n_rows, sp, scale))
            return None, ctx.link.token(ctx.n_rays, rows.device), None, None, None, None
        if g is None:
            return None, None, None, None, None, None
        if g.numel() == 0:                       # no rays: an empty tensor has no storage to pass down
            return None, g, None, None, None, None
        # scale only the contributing rows (a few % of the rays) instead of streaming the dense tensor again;
        # out_ori is detached in the reference's loss (optim.py:100): no gradient for it
        sc = g_loss.detach().to(torch.float64).reshape(1).contiguous()
        new = sc.clone()
        if ctx.applied is not None:
            # a second backward over the same graph (retain_graph=True): the saved rows already carry the previous
            # incoming gradient, so rescale by the ratio.  Rare path: one host sync to refuse an unrecoverable 0.
            if float(ctx.applied.item()) == 0.0:
                raise RuntimeError("ray_loss: backward was already run with a zero incoming gradient; the saved rows "
                                   "cannot be rescaled -- recompute the loss instead of re-using the graph")
            sc = sc / ctx.applied
        ctx.applied = new
        with _on(g.device):
            _lib.check(_lib.lib().drt_scale_rows3(g.data_ptr(), rows.data_ptr(), n_rows.data_ptr(), sc.data_ptr(), _stream()))
        return None, g, None, None, None, None


class _RenderRayLossFused(torch.autograd.Function):
    """render_transparent + ray_loss + d/d vertices in ONE kernel pass (nothing dense written)."""

    @staticmethod
    def forward(ctx, vertices, origin, ray_dir, screen_pixel, valid, scene, ior_int, ior_ext, grid=(0, None)):
        v = _f64c(vertices.detach(), "vertices")
        o = _f64c(origin, "origin")
        d = _f64c(ray_dir, "ray_dir")
        sp = _f64c(screen_pixel, "screen_pixel")
        va = _flag_bytes(valid, "valid", o.shape[0])
        loss = det.scalar(o.device)
        grad_v = det.acc(v)
        with _on(o.device):
            _arm_seed(scene.optix_mesh._h, grid, o.shape[0])
            _lib.check(_lib.lib().drt_render_ray_loss_fused(
                scene.optix_mesh._h, v.data_ptr(), o.data_ptr(), d.data_ptr(), sp.data_ptr(), va.data_ptr(), o.shape[0],
                float(ior_int), float(ior_ext), loss.data_ptr(), grad_v.data_ptr(), None, *_tile_hint(o.shape[0]), grid[0], _lib.ptr(grid[1]), _stream()))
        ctx.wide = grad_v if grad_v.dtype == torch.int64 else None
        ctx.save_for_backward(det.value(grad_v, v))
        return det.value(loss)

    @staticmethod
    def backward(ctx, g_loss):
        (grad_v,) = ctx.saved_tensors
        if det.SINK is not None and ctx.wide is not None:
            det.SINK.append((ctx.wide, g_loss))
            return None, None, None, None, None, None, None, None, None
        return grad_v * g_loss, None, None, None, None, None, None, None, None


def ray_loss(out_ori, out_dir, mask, screen_pixel, valid):
    """sum over valid & mask rays of |out_dir - normalize(screen_pixel - out_ori.detach())|^2."""
    link = getattr(out_dir, "_drt_link", None) if SPARSE_LOSS_GRAD else None
    if link is not None and not (out_dir.requires_grad and isinstance(out_dir.grad_fn, _RenderTransparent._backward_cls)):
        link = None
    if link is not None:
        # The row-list hand-off RECOMPUTES out_ori / out_dir of the listed rays from the forward's stored path, and lists rays by the
        # forward's own mask: it is only the gradient of THIS call when all three tensors are the forward's own, untouched ones.  A
        # different or modified out_ori / mask (extra rows set, a shifted origin) takes the dense gradient, which reads what was passed.
        own = (link.mask is not None and link.mask() is mask and mask._version == 0
               and link.out_ori is not None and link.out_ori() is out_ori and out_ori._version == 0 and out_dir._version == 0)
        if not own:
            link = None
    return _RayLoss.apply(out_ori, out_dir, mask, screen_pixel, valid, link)


def edge_tables(F, V, want_rows=False, check=True):
    """Edges [E,2], E2F [E,2,3], mean_len (reference DiffRender.py:338-355) on the device of ``F`` / ``V`` -- the reference
    goes through trimesh's host-side ``group_rows`` / ``edges_face``.  One stable radix sort of the 3F directed-edge keys
    in libdrt_hip (``drt_edge_tables``).  Order as pinned in mesh_io.group_rows_pairs: edges ascend by (min vertex, max
    vertex); the first face of a pair is the one with the lower directed-edge row.  Asserts watertightness
    (DiffRender.py:305); that check and ``mean_len`` come back in ONE small device->host copy (a topology change is not
    on the per-iteration path).  ``want_rows``: also the int32 [3F] directed-edge -> unique-edge map.  ``check=False``: no read-back at all
    (``mean_len`` is then None, watertightness is the caller's business: the device remesher's rounds, which keep the mesh closed by
    construction and install the result through ``Scene._set_topology``, which checks)."""
    if not F.is_cuda:
        raise RuntimeError("edge_tables needs GPU tensors (there is no CPU path in the product)")
    Fc = F.to(torch.long).contiguous()
    Vc = _f64c(V.detach(), "V")
    n_f, n_v = Fc.shape[0], Vc.shape[0]
    assert (3 * n_f) % 2 == 0 and n_f > 0, "mesh is not watertight: every edge must be shared by exactly two faces"
    n_e = 3 * n_f // 2
    dev = Fc.device
    Edges = torch.empty((n_e, 2), dtype=torch.long, device=dev)
    E2F = torch.empty((n_e, 2, 3), dtype=torch.long, device=dev)
    rows = torch.empty(3 * n_f, dtype=torch.int32, device=dev)
    out = torch.zeros(2, dtype=torch.float64, device=dev)           # [mean_len, status (int32 in the low bytes of word 1)]
    with _on(dev):
        ws = torch.empty(int(_lib.lib().drt_edge_tables_workspace(n_f)), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib().drt_edge_tables(Fc.data_ptr(), n_f, Vc.data_ptr(), n_v, ws.data_ptr(), Edges.data_ptr(), E2F.data_ptr(),
                                              rows.data_ptr(), out.data_ptr(), out[1:].data_ptr(), _stream()))
    if not check:
        return (Edges, E2F, None, rows) if want_rows else (Edges, E2F, None)
    host = out.cpu()
    status = int(host[1:].view(torch.int32)[0])
    if status & 2:
        raise IndexError(f"faces index vertices outside [0, {n_v}) (torch's indexing, which the reference uses here, raises as well)")
    assert status == 0, "mesh is not watertight: every edge must be shared by exactly two faces"
    mean_len = float(host[0])
    return (Edges, E2F, mean_len, rows) if want_rows else (Edges, E2F, mean_len)


class Scene(StepwiseMixin):
    def __init__(self, mesh_path, cuda_device=0):
        self.cuda_device = int(cuda_device)
        self.optix_mesh = optix_mesh(self.cuda_device)
        self._mesh_stale = False
        self._epoch = 0          # bumped by every change of vertices / mesh: what a lazy result (SampleSet) is tied to
        self.update_mesh(mesh_path)

    # ------------------------------------------------------------------ mesh state
    @property
    def _dev(self):
        return torch.device("cuda", self.cuda_device)

    def update_mesh(self, mesh_path):
        mesh = mesh_path if isinstance(mesh_path, mesh_io.TriMesh) else mesh_io.load(mesh_path, process=False)
        self._mesh = mesh
        self._mesh_stale = False
        self._epoch += 1
        self.vertices = torch.tensor(mesh.vertices, dtype=Float, device=self._dev)
        self.faces = torch.tensor(mesh.faces, dtype=torch.long, device=self._dev)
        self.init_edge()                                  # also the watertightness assert of DiffRender.py:305
        opt_v = self.vertices.detach().to(torch.float32)
        opt_F = self.faces.to(torch.int32)
        self.optix_mesh.update_mesh(opt_F, opt_v)

    def init_edge(self):
        self.Edges, self.E2F, self.mean_len, self._row2edge = edge_tables(self.faces, self.vertices.detach(), want_rows=True)

    def subdivide_midpoint(self, float32_positions=True):
        """One 1 -> 4 midpoint refinement of the current mesh entirely on the device (a level-of-detail step that is pure
        subdivision needs no host remesher): new vertices, faces, edge tables and LBVH.  Same vertex / face order as
        drt_amd.mesh_io.subdivide_midpoint."""
        V = _f64c(self.vertices.detach(), "vertices")
        F = self.faces.contiguous()
        n_v, n_f, n_e = V.shape[0], F.shape[0], self.Edges.shape[0]
        V2 = torch.empty((n_v + n_e, 3), dtype=torch.float64, device=V.device)
        F2 = torch.empty((4 * n_f, 3), dtype=torch.long, device=V.device)
        with _on(V.device):
            _lib.check(_lib.lib().drt_subdivide_midpoint(F.data_ptr(), n_f, V.data_ptr(), n_v, self.Edges.contiguous().data_ptr(), n_e,
                                                         self._row2edge.data_ptr(), int(float32_positions), F2.data_ptr(), V2.data_ptr(), _stream()))
        self._set_topology(V2, F2)

    def _set_topology(self, vertices, faces):
        """Install device-resident vertices / faces as the new mesh (host record synced lazily, see `mesh`)."""
        self._epoch += 1
        self.vertices, self.faces = vertices, faces
        self._mesh = mesh_io.TriMesh(np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64))
        self._mesh_stale = self._faces_stale = True
        self.init_edge()
        self.optix_mesh.update_mesh(self.faces.to(torch.int32), self.vertices.detach().to(torch.float32))

    @property
    def mesh(self):
        """Host-side mesh record; vertex positions are copied back lazily (the reference pays a
        device->host sync on every iteration for this, DiffRender.py:381)."""
        if getattr(self, "_faces_stale", False):
            self._mesh = mesh_io.TriMesh(self.vertices.detach().cpu().numpy(), self.faces.cpu().numpy())
            self._faces_stale = self._mesh_stale = False
        if self._mesh_stale:
            self._mesh.vertices = self.vertices.detach().cpu().numpy()
            self._mesh_stale = False
        return self._mesh

    @property
    def triangles(self):
        return self.vertices[self.faces]

    def update_verticex(self, vertices: torch.Tensor):
        if vertices.shape != self.vertices.shape:
            raise RuntimeError(f"vertices must have shape {tuple(self.vertices.shape)}")
        self._epoch += 1
        self.vertices = vertices
        self.optix_mesh.update_vert_f64(vertices)
        self._mesh_stale = True

    # ------------------------------------------------------------------ tracer access
    def optix_intersect(self, ray: Ray):
        optix_ray = torch.cat([ray.origin.detach().to(torch.float32), ray.direction.detach().to(torch.float32)], dim=1)
        T, faces_ind = self.optix_mesh.intersect(optix_ray)
        return faces_ind.to(torch.long), T > 0

    def render_mask(self, origin, ray_dir):
        optix_ray = torch.cat([origin.detach().to(torch.float32), ray_dir.detach().to(torch.float32)], dim=1)
        return self.optix_mesh.intersect_any(optix_ray).to(Float)

    def release_outputs(self, graph_sets=False):
        """Drops the dense outputs this scene keeps for re-use (RECYCLE_OUTPUTS: up to two sets of 55 B per ray) and the buffers zeroed
        ahead of time (PREFILL_NEXT); the next render call allocates and fills fresh ones.
        ``graph_sets=True`` also drops the output sets that CAPTURED render calls took for good (a graph's replays read and write those
        buffers: they were allocated before the capture, so the graph's own memory pool does not keep them alive) -- only legal once every
        graph captured on this scene has been destroyed; left alone (the default) they live as long as the scene."""
        om = self.optix_mesh
        with _on(self._dev):
            _lib.check(_lib.lib().drt_outputs_cancel(om._h))
            if getattr(om, "_prefilled", None) is not None:
                _lib.check(_lib.lib().drt_prefill_wait(om._h, _stream()))
        om._prefilled = None
        pool = getattr(om, "_out_pool", None)
        if pool is not None:
            pool.entries.clear()
        if graph_sets:
            om._graph_sets = []

    # ------------------------------------------------------------------ refraction path
    def bind_rays(self, origin, ray_dir, screen_pixel=None, valid=None, shared=None):
        """A handle for constant rays (and, optionally, their constant targets): see RayBinding.  ``render_transparent(handle)``."""
        return RayBinding(self, origin, ray_dir, screen_pixel, valid, shared)

    def render_transparent(self, origin, ray_dir=None):
        link = _GradLink()
        binding = origin if isinstance(origin, RayBinding) else None
        if binding is not None:
            origin, ray_dir = binding.origin, binding.ray_dir
            grid = binding._grid()
            link.targets = binding.targets
        else:
            grid = _grid_cache(origin, ray_dir, origin.shape[0], *_tile_hint(origin.shape[0])) if origin.is_contiguous() and ray_dir.is_contiguous() else (0, None)
        out_ori, out_dir, mask = _RenderTransparent.apply(self.vertices, origin, ray_dir, self, intIOR, extIOR, link, grid, binding)
        out_dir._drt_link = link            # lets ray_loss hand its gradient over as a row list (see _GradLink)
        link.mask = weakref.ref(mask)
        link.out_ori = weakref.ref(out_ori)
        return out_ori, out_dir, mask

    def ray_loss_fused(self, origin, ray_dir, screen_pixel, valid):
        """ray_loss of this view without materialising out_ori/out_dir/mask."""
        if isinstance(origin, RayBinding):
            b = origin
            return _RenderRayLossFused.apply(self.vertices, b.origin, b.ray_dir, screen_pixel, valid, self, intIOR, extIOR, b._grid())
        grid = _grid_cache(origin, ray_dir, origin.shape[0], *_tile_hint(origin.shape[0])) if origin.is_contiguous() and ray_dir.is_contiguous() else (0, None)
        return _RenderRayLossFused.apply(self.vertices, origin, ray_dir, screen_pixel, valid, self, intIOR, extIOR, grid)

    # ------------------------------------------------------------------ smoothness branch
    def dihedral_angle(self):
        """cos of the dihedral angle of every unique edge, differentiable w.r.t. vertices (DiffRender.py:440-443)."""
        return _Dihedral.apply(self.vertices, self.E2F)

    def sm_loss_fused(self):
        """sum -log(1 + cos dihedral) (reference optim.py:82-89) with its gradient in one kernel pass."""
        return _SmLossFused.apply(self.vertices, self.E2F)

    # ------------------------------------------------------------------ silhouette branch
    def silhouette_edge(self, origin: torch.Tensor):
        assert origin.dim() == 1
        if LAZY_SILHOUETTE and not torch.cuda.is_current_stream_capturing():
            return SilhouetteEdges(self.Edges, None, scene=self, origin=origin)       # (flags and edge set computed when somebody needs them)
        v = _f64c(self.vertices.detach(), "vertices")
        o = _f64c(origin.detach(), "origin")
        n = self.E2F.shape[0]
        flags = torch.empty(n, dtype=torch.uint8, device=v.device)
        with _on(v.device):
            _lib.check(_lib.lib().drt_silhouette_flags(v.data_ptr(), self.E2F.data_ptr(), n, o.data_ptr(), flags.data_ptr(), _stream()))
        if LAZY_SILHOUETTE:
            return SilhouetteEdges(self.Edges, flags)
        return self.Edges[flags.view(torch.bool)]

    def primary_visibility(self, silhouette_edge, camera_M, origin, detach_depth=False):
        """(index int64 [M,2] (x, y), output float32 [M]) of the in-view silhouette samples (DiffRender.py:459-479)."""
        lazy_ok = LAZY_VISIBILITY and not torch.cuda.is_current_stream_capturing()
        if isinstance(silhouette_edge, SilhouetteEdges) and silhouette_edge._t is None and lazy_ok:
            # NOTHING is computed now: the reference's loop feeds the pair straight into `(mask.view(resy, resx)[index[:, 1], index[:, 0]] -
            # output).abs().sum()` (optim.py:78) and adds the views' terms up (optim.py:72-80) -- SampleSet / LazySum evaluate that sum with one
            # fused launch; anything else that looks at the pair gets the sampling kernel, the compaction and the reference's tensors.
            ss = SampleSet(self, self.vertices, silhouette_edge, camera_M, origin, bool(detach_depth), int(resx), int(resy))
            return LazyIndex(ss), LazyOutput(ss)
        if isinstance(silhouette_edge, SilhouetteEdges):      # (unwrapped here: autograd.Function.apply should see plain tensors / tuples)
            silhouette_edge = (silhouette_edge._edges, silhouette_edge._flags) if silhouette_edge._t is None else silhouette_edge.tensor()
        return _EdgeSample.apply(self.vertices, silhouette_edge, camera_M, origin, self, bool(detach_depth), int(resx), int(resy))

    def vh_loss_fused(self, camera_M, origin, soft_mask):
        """sum |soft_mask[y, x] - 0.5| over the visible silhouette samples of one view and its vertex gradient
        (one view of reference optim.py:73-78) without any host round trip."""
        return self.vh_loss_fused_views([(camera_M, origin, soft_mask)])

    def vh_loss_fused_views(self, views):
        """The same summed over ``views`` = [(camera_M, origin[3], soft_mask), ...] (the whole of optim.py:73-78):
        one autograd node, three kernels per view."""
        flat = []
        for camera_M, origin, soft_mask in views:
            flat += [pack_camera(camera_M), origin, soft_mask]
        return _VhLossFused.apply(self.vertices, self, int(resx), int(resy), True, *flat)


LAZY_SILHOUETTE = True
# LAZY_VISIBILITY: primary_visibility returns stand-ins for (index, output) -- see SampleSet.  They behave as the reference's tensors for
# indexing (incl. assignment), len / shape / dtype / any attribute, operators, torch functions, isinstance(x, torch.Tensor) and
# torch.is_tensor(x); what they cannot do is pass a C++-level tensor check that does not go through __torch_function__ (torch.compile /
# torch.jit tracing of the caller's loss, a custom C++ op taking the pair): such callers set LAZY_VISIBILITY = False.
LAZY_VISIBILITY = os.environ.get("DRT_LAZY_VISIBILITY", "1") != "0"
