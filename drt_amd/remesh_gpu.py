"""Isotropic remeshing between optimisation passes ON THE DEVICE -- the data-parallel form of drt_amd/csrc/drt_remesh.cpp.

    GpuMeshlabserver().remesh(scene, remesh_len)          reference optim.py:12-52 (Meshlabserver.remesh)

The reference shells out to MeshLab's "Remeshing: Isotropic Explicit Remeshing" (3 iterations, TargetLen = remesh_len,
MaxSurfDist 1, refine / collapse / swap / smooth / reproject).  ``drt_amd.remesh`` runs that algorithm (Botsch & Kobbelt 2004)
sequentially on the host; here every step runs on the GPU without the mesh leaving it:

    split     long edges flagged by directed-edge slot, midpoints and the 1 -> 2 / 3 / 4 face patterns for all faces at once (drt_rm_split_*)
    collapse  all short edges evaluated at once (link condition, valences, fold test against the consensus normals, maximum
              length, surface distance of the midpoint and of every surviving face's centroid through the scene's closest-point
              kernel); the survivors claim their two rings by priority (length, index) and the ones that hold every claim are
              applied -- disjoint neighbourhoods commute -- then the rest is evaluated again on the new mesh, until a round
              applies nothing
    flip      the same evaluate / claim / apply rounds on the four vertices of every edge (the face across an edge and the "new edge
              exists already" test come from the vertex -> face lists: no edge table)
    smooth    tangential relaxation of every vertex, faces that would fold take their vertices back (four rounds)
    project   closest point on the INPUT surface (the tree of the scene the mesh came from), same roll-back
    topology  vertex -> face lists by drt_rm_vertex_faces (count / scan / fill / sort each run), once per round: every step works on the
              directed-edge slots 3 f + k of the face array (the lo -> hi slot of an edge speaks for it) and these lists -- no edge table

The geometric decisions, the conflict-free application and the tables are hand-written kernels (csrc/drt_remesh_gpu.hip); torch
provides the prefix sums and the final compactions in between (plumbing).  The result is a closed oriented manifold of the same
genus with edge lengths concentrated around the target, on the input surface, deterministic -- and statistically the mesh the
host version produces (tests/test_gpu_remesh.py holds the two against each other); vertex order and the exact set of operations
differ, as they do between the host version and MeshLab.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from .optix_mesh import _stream
from .remesh import SPLIT, COLLAPSE, FLIP, SMOOTH, REPROJECT, CHECK_DIST, ALL  # noqa: F401

MAX_Q = 24                 # surface-distance queries per collapse candidate (midpoint + surviving faces); more -> the edge is left alone
MAX_ROUNDS = 96            # evaluate / claim / apply rounds per step (a round applies an independent set of the candidates: a few per cent)
SUB_ROUNDS = int(__import__("os").environ.get("DRT_REMESH_SUB_ROUNDS", 3))     # claim / apply pairs per evaluation (one set of edge tables, vertex -> face lists and surface queries
                           # serves several independent sets: candidates whose neighbourhood an earlier pair of the round touched sit out until the next evaluation)
ROUND_BATCH = int(__import__("os").environ.get("DRT_REMESH_ROUND_BATCH", 4))  # rounds enqueued between two read-backs of the step's control block: the DEVICE ends a step
                           # (drt_rm_round_end); the rounds of a batch that come after the end are no-ops (every kernel returns at its first instruction)
TAIL_CUT = int(__import__("os").environ.get("DRT_REMESH_TAIL_CUT", 32))       # a step ends when a round applies less than 1 / TAIL_CUT of what its first round applied
DEBUG = False


def _check(rc):
    _lib.check(rc)


class _Work:
    """The mesh being edited: float64 vertices [V,3], int64 faces [F,3] on the device, and the derived tables."""

    def __init__(self, V, F, surface, max_dist, hint=0.0):
        self.V, self.F = V.contiguous(), F.contiguous()
        self.hint = hint                 # how far from the input surface a vertex is expected to be at most (bounds the projection's search; 0: no bound)
        self.surface, self.max_dist = surface, max_dist
        self.dev = V.device
        self.n_dead_vertices = 0         # vertices the collapses since the last compact() left without a face
        # what a caller can check afterwards: evaluation rounds per step, steps that ran out of rounds with candidates still passing,
        # moves that still folded a face after the last roll-back round (each is then rolled back by more rounds: see move_vertices)
        self.stats = {"collapse_rounds": 0, "flip_rounds": 0, "collapse_unfinished": 0, "flip_unfinished": 0, "move_rounds_max": 0, "move_unresolved": 0}

    def _new_ctl(self):
        """The round control block of a step (drt_rm_round_end): live = 1, everything else 0."""
        ctl = torch.zeros(8, dtype=torch.int32, device=self.dev)
        ctl[0] = 1
        return ctl

    # ---- derived tables
    def csr(self, normals=False, live=None):
        """vertex -> incident faces: vf_start int64 [V+1], vf_face int64 [3F] (ascending face order inside a vertex; faces a collapse round
        killed -- indices -1 -- in nobody's list), by drt_rm_vertex_faces; with ``normals`` also the area-weighted vertex normals."""
        nv, nf = self.V.shape[0], self.F.shape[0]
        vf_start = torch.empty(nv + 1, dtype=torch.long, device=self.dev)
        vf_face = torch.empty(3 * nf, dtype=torch.long, device=self.dev)
        count = torch.empty(nv, dtype=torch.int32, device=self.dev)
        vn = torch.empty_like(self.V) if normals else None
        _check(_lib.lib().drt_rm_vertex_faces(self.F.data_ptr(), nf, nv, count.data_ptr(), vf_start.data_ptr(), vf_face.data_ptr(),
                                              self.V.data_ptr(), _lib.ptr(vn), _lib.ptr(live), _stream()))
        return (vf_start, vf_face, vn) if normals else (vf_start, vf_face)

    def vertex_normals(self, vf_start, vf_face):
        vn = torch.empty_like(self.V)
        _check(_lib.lib().drt_rm_vertex_normals(self.F.data_ptr(), self.V.data_ptr(), vf_start.data_ptr(), vf_face.data_ptr(), self.V.shape[0],
                                                vn.data_ptr(), _stream()))
        return vn

    def near_surface(self, points):
        """bool [n]: within max_dist of the input surface (CheckSurfDist)."""
        if self.surface is None or not np.isfinite(self.max_dist) or len(points) == 0:
            return torch.ones(len(points), dtype=torch.bool, device=self.dev)
        return self.surface.closest_point(points.contiguous(), want_face=False)[0] <= self.max_dist

    # ---- 1. refine
    def split_long_edges(self, max_len):
        """Long edges by directed-edge slot (no edge table): flag, prefix sum = the new vertices' numbers, the midpoint id of every slot on
        both sides of its edge, faces per face, prefix sum, ONE host round trip for both totals (the sizes of the new arrays), write."""
        lib = _lib.lib()
        nv, nf = self.V.shape[0], self.F.shape[0]
        vf_start, vf_face = self.csr()
        flag = torch.empty(3 * nf, dtype=torch.uint8, device=self.dev)
        _check(lib.drt_rm_split_mark(self.F.data_ptr(), nf, self.V.data_ptr(), float(max_len), flag.data_ptr(), _stream()))
        rank = torch.cumsum(flag, 0)                             # (int64)
        mid = torch.full((3 * nf,), -1, dtype=torch.long, device=self.dev)      # (a slot nobody owns -- the hi -> lo side of a boundary edge, were the mesh open -- is left alone)
        count = torch.empty(nf, dtype=torch.long, device=self.dev)
        _check(lib.drt_rm_split_plan(self.F.data_ptr(), nf, vf_start.data_ptr(), vf_face.data_ptr(), flag.data_ptr(), rank.data_ptr(), nv,
                                     mid.data_ptr(), count.data_ptr(), _stream()))
        offset = torch.cumsum(count, 0)
        n_split, n_out = (int(x) for x in torch.stack([rank[-1], offset[-1]]).tolist())
        if n_split == 0:
            return 0
        offset = offset - count
        newV = torch.empty((nv + n_split, 3), dtype=torch.float64, device=self.dev)
        newV[:nv] = self.V
        out = torch.empty((n_out, 3), dtype=torch.long, device=self.dev)
        _check(lib.drt_rm_split_faces(self.F.data_ptr(), nf, mid.data_ptr(), newV.data_ptr(), offset.data_ptr(), out.data_ptr(), _stream()))
        self.V, self.F = newV, out
        return n_split

    # ---- 2. collapse
    def _filter_by_surface(self, ok, nq, q, n_items, max_q, live=None):
        """CheckSurfDist on the device (drt_rm_surface_filter): candidates whose query points leave the input surface lose their `ok`."""
        if self.surface is None or not np.isfinite(self.max_dist) or n_items == 0:
            return
        _check(_lib.lib().drt_rm_surface_filter(self.surface._h, ok.data_ptr(), _lib.ptr(nq), q.data_ptr(), n_items, max_q, float(self.max_dist),
                                                _lib.ptr(live), _stream()))

    def collapse_short_edges(self, min_len, max_len):
        """Rounds of evaluate (every directed-edge slot of the face array at once) -> surface filter -> claim / apply.  Round 6: no candidate
        list, no per-round compaction of the face array (killed faces stay in place with indices -1 until the step is through), and the
        DEVICE decides when the step is over (drt_rm_round_end): the driver enqueues ROUND_BATCH rounds at a time and reads the step's
        control block back once per batch -- two or three host round trips per step, where there were seven per round."""
        lib = _lib.lib()
        nf = self.F.shape[0]
        dev = self.dev
        v_alive = torch.ones(self.V.shape[0], dtype=torch.uint8, device=dev)
        # per-slot workspaces of the step (3 F slots; the face array keeps its size until the compaction at the end)
        E_snap = torch.empty((3 * nf, 2), dtype=torch.long, device=dev)
        length = torch.empty(3 * nf, dtype=torch.float64, device=dev)
        ok = torch.empty(3 * nf, dtype=torch.uint8, device=dev)
        nq = torch.empty(3 * nf, dtype=torch.int32, device=dev)
        q = torch.empty((3 * nf, MAX_Q, 3), dtype=torch.float64, device=dev)
        check_dist = self.surface is not None and np.isfinite(self.max_dist)
        ql_cap = 3 * nf * 8                                      # query points of a round (a candidate that does not fit waits for the next one)
        ql_item = torch.empty(ql_cap, dtype=torch.int32, device=dev) if check_dist else None
        ql_point = torch.empty((ql_cap, 3), dtype=torch.float64, device=dev) if check_dist else None
        ql_count = torch.zeros(1, dtype=torch.int32, device=dev) if check_dist else None
        nv = self.V.shape[0]
        lock = torch.empty(nv, dtype=torch.int64, device=dev)                           # (workspaces: preset by the call)
        dirty = torch.empty(nv, dtype=torch.uint8, device=dev)
        f_alive = torch.ones(nf, dtype=torch.uint8, device=dev)                          # (drt_rm_kill_faces leaves it all ones again)
        ctl = self._new_ctl()
        live, n_done = ctl.data_ptr(), ctl.data_ptr() + 4
        rnd = 0
        while True:
            for _ in range(ROUND_BATCH):                          # rounds enqueued ahead: the device ends the step (drt_rm_round_end)
                vf_start, vf_face, vn = self.csr(normals=True, live=ctl)      # (killed faces hold -1: in nobody's list)
                _check(lib.drt_rm_collapse_eval_all(self.F.data_ptr(), nf, self.V.data_ptr(), vn.data_ptr(), vf_start.data_ptr(), vf_face.data_ptr(),
                                                    float(min_len), float(max_len), MAX_Q, E_snap.data_ptr(), length.data_ptr(), ok.data_ptr(),
                                                    nq.data_ptr(), q.data_ptr(), _lib.ptr(ql_item), _lib.ptr(ql_point), _lib.ptr(ql_count), ql_cap, live, _stream()))
                # CheckSurfDist: the midpoint and the centroid of every face that survives must stay near the input surface
                if check_dist:
                    _check(lib.drt_rm_surface_filter_list(self.surface._h, ok.data_ptr(), ql_item.data_ptr(), ql_point.data_ptr(), ql_count.data_ptr(), ql_cap,
                                                          float(self.max_dist), live, _stream()))
                _check(lib.drt_rm_collapse_apply(None, 3 * nf, ok.data_ptr(), E_snap.data_ptr(), self.F.data_ptr(), self.V.data_ptr(),
                                                 vf_start.data_ptr(), vf_face.data_ptr(), nv, float(min_len), 0x9E3779B9 * (rnd + 1) & 0xFFFFFFFF, rnd, length.data_ptr(),
                                                 lock.data_ptr(), f_alive.data_ptr(), v_alive.data_ptr(), dirty.data_ptr(), SUB_ROUNDS, n_done, live, _stream()))
                _check(lib.drt_rm_kill_faces(self.F.data_ptr(), f_alive.data_ptr(), nf, live, _stream()))
                _check(lib.drt_rm_round_end(ctl.data_ptr(), TAIL_CUT, _stream()))
                rnd += 1
            still_live, done, _, _, ran = ctl.tolist()[:5]        # the batch's one host round trip
            if not still_live or rnd + ROUND_BATCH > MAX_ROUNDS:
                break
        self.stats["collapse_rounds"] += ran
        self.stats["collapse_unfinished"] += int(bool(still_live))        # candidates were still being applied when the rounds ran out
        if done:
            # every collapse killed exactly two faces and one vertex (closed manifold, link condition): the sizes are known, so the compaction
            # needs no read-back
            keep = torch.nonzero_static(self.F[:, 0] >= 0, size=nf - 2 * done).squeeze(1)
            self.F = self.F[keep].contiguous()
            self.n_dead_vertices += done
        return done

    # ---- 3. flip
    def flip_edges(self, max_len):
        lib = _lib.lib()
        done = first = 0
        nv = self.V.shape[0]
        nf = self.F.shape[0]
        n_e = 3 * nf                                              # one candidate per directed-edge slot (the lo -> hi slot of an edge speaks for it)
        lock = torch.empty(nv, dtype=torch.int64, device=self.dev)                       # (workspaces: preset by the first round's call)
        dirty = torch.empty(nv, dtype=torch.uint8, device=self.dev)
        ok = torch.empty(n_e, dtype=torch.uint8, device=self.dev)
        quad = torch.empty((n_e, 6), dtype=torch.long, device=self.dev)
        q = torch.empty((n_e, 3), dtype=torch.float64, device=self.dev)
        ctl = self._new_ctl()
        live, n_done = ctl.data_ptr(), ctl.data_ptr() + 4
        rnd = 0
        while True:
            for _ in range(ROUND_BATCH):
                vf_start, vf_face, vn = self.csr(normals=True, live=ctl)
                _check(lib.drt_rm_flip_eval(self.F.data_ptr(), nf, self.V.data_ptr(), vn.data_ptr(), vf_start.data_ptr(), vf_face.data_ptr(),
                                            float(max_len), ok.data_ptr(), quad.data_ptr(), q.data_ptr(), live, _stream()))
                self._filter_by_surface(ok, None, q, n_e, 1, live=ctl)   # the midpoint of the new edge
                _check(lib.drt_rm_flip_apply(n_e, ok.data_ptr(), quad.data_ptr(), self.F.data_ptr(), nv, rnd, lock.data_ptr(), dirty.data_ptr(), SUB_ROUNDS,
                                             n_done, live, _stream()))
                _check(lib.drt_rm_round_end(ctl.data_ptr(), TAIL_CUT, _stream()))
                rnd += 1
            still_live, done, _, _, ran = ctl.tolist()[:5]        # the batch's one host round trip
            if not still_live or rnd + ROUND_BATCH > MAX_ROUNDS:
                break
        self.stats["flip_rounds"] += ran
        self.stats["flip_unfinished"] += int(bool(still_live))
        return done

    # ---- 4./5. relaxation and projection, with roll-back
    def move_vertices(self, target, vf_start, vf_face):
        lib = _lib.lib()
        old = self.V.clone()
        vn = self.vertex_normals(vf_start, vf_face)
        nf, nv = self.F.shape[0], self.V.shape[0]
        a0 = torch.empty(nf, dtype=torch.float64, device=self.dev)
        _check(lib.drt_rm_face_agreement(self.F.data_ptr(), self.V.data_ptr(), vn.data_ptr(), nf, a0.data_ptr(), _stream()))
        self.V = target.contiguous()
        revert = torch.empty(nv, dtype=torch.uint8, device=self.dev)
        n_bad = torch.zeros(1, dtype=torch.int32, device=self.dev)
        # every round takes the vertices of the faces that fold (or degenerate) back to where they were; that can fold a neighbour, whose
        # vertices go back in the next round.  The cascade ends -- with every vertex back the mesh is the one it started from -- so the loop
        # runs until a round finds nothing (four rounds were the rule before, and what was still folded after them stayed folded)
        prev = -1
        for rnd in range(1, 65):
            _check(lib.drt_rm_move_check(self.F.data_ptr(), self.V.data_ptr(), old.data_ptr(), vn.data_ptr(), a0.data_ptr(), nf, nv,
                                         revert.data_ptr(), n_bad.data_ptr(), _stream()))
            n = int(n_bad.item())
            if n == 0:
                break
            if n == prev:        # nothing left to take back: these faces were degenerate before the move already (not this step's doing)
                self.stats["move_unresolved"] = max(self.stats["move_unresolved"], n)
                break
            prev = n
        self.stats["move_rounds_max"] = max(self.stats["move_rounds_max"], rnd)

    def smooth_tangential(self):
        vf_start, vf_face = self.csr()
        target = torch.empty_like(self.V)
        _check(_lib.lib().drt_rm_smooth_target(self.F.data_ptr(), self.V.data_ptr(), vf_start.data_ptr(), vf_face.data_ptr(), self.V.shape[0],
                                               target.data_ptr(), _stream()))
        self.move_vertices(target, vf_start, vf_face)

    def project_to_surface(self):
        if self.surface is None:
            return
        vf_start, vf_face = self.csr()
        closest = torch.empty_like(self.V)
        _check(_lib.lib().drt_rm_closest_near(self.surface._h, self.V.data_ptr(), self.V.shape[0], float(self.hint), closest.data_ptr(), _stream()))
        self.move_vertices(closest, vf_start, vf_face)

    def compact(self):
        """Drop the vertices no face uses (ascending order kept)."""
        if not self.n_dead_vertices:
            return
        used = torch.zeros(self.V.shape[0], dtype=torch.bool, device=self.dev)
        used[self.F.reshape(-1)] = True
        remap = torch.cumsum(used.long(), 0) - 1
        keep = torch.nonzero_static(used, size=self.V.shape[0] - self.n_dead_vertices).squeeze(1)      # (the collapses' count: no read-back)
        self.V = self.V[keep].contiguous()
        self.F = remap[self.F].contiguous()
        self.n_dead_vertices = 0


def isotropic_remesh_gpu(vertices, faces, target_len, surface=None, iterations=3, max_surf_dist=1.0, flags=ALL, return_stats=False):
    """(vertices float64 [V,3], faces int64 [F,3]) on the device -> the same, re-tessellated to edge lengths around ``target_len``.
    ``surface``: an ``optix_mesh`` holding the INPUT surface (closest-point queries for the projection step and MaxSurfDist); None skips
    both.  Closed manifold in, closed manifold out."""
    if not vertices.is_cuda:
        raise RuntimeError("isotropic_remesh_gpu needs device tensors (drt_amd.remesh is the host version)")
    w = _Work(vertices.detach().to(torch.float64), faces.to(torch.long), surface, float(max_surf_dist) if (flags & CHECK_DIST) and max_surf_dist > 0 else float("inf"), hint=float(target_len))
    min_len, max_len = 0.8 * target_len, 4.0 / 3.0 * target_len
    stats = {"split": 0, "collapsed": 0, "flipped": 0, "iterations": 0}
    with torch.no_grad(), torch.cuda.device(vertices.device):
        # (vertices of the INPUT that no face uses, counted once before anything is enqueued: compact() then knows its sizes from the collapses alone)
        used = torch.zeros(w.V.shape[0], dtype=torch.bool, device=w.dev)
        used[w.F.reshape(-1)] = True
        w.n_dead_vertices = w.V.shape[0] - int(used.sum())
        for _ in range(iterations):
            if flags & SPLIT:
                for _k in range(3):
                    n = w.split_long_edges(max_len)
                    stats["split"] += n
                    if not n:
                        break
            if flags & COLLAPSE:
                stats["collapsed"] += w.collapse_short_edges(min_len, max_len)
            if flags & FLIP:
                stats["flipped"] += w.flip_edges(max_len)
            if flags & SMOOTH:
                w.smooth_tangential()
            if flags & REPROJECT:
                w.project_to_surface()
            w.compact()
            stats["iterations"] += 1
    stats.update(w.stats)
    return (w.V, w.F, stats) if return_stats else (w.V, w.F)


class GpuMeshlabserver:
    """Same role and call shape as the reference class (optim.py:12-52); the mesh stays on the device (``Scene._set_topology``), positions
    are rounded through float32 like the reference's PLY round trip."""

    def __init__(self, iterations=3, max_surf_dist=1.0):
        self.iterations, self.max_surf_dist = iterations, max_surf_dist

    def remesh(self, scene, remesh_len):
        V, F = isotropic_remesh_gpu(scene.vertices.detach(), scene.faces, remesh_len, surface=scene.optix_mesh, iterations=self.iterations,
                                    max_surf_dist=self.max_surf_dist)
        scene._set_topology(V.to(torch.float32).to(torch.float64), F)
        return scene
