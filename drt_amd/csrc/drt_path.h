// drt_path.h -- one camera ray through the two-bounce refraction path, and its adjoint.
//
//   trace_path              <- Scene.trace2 + the occlusion test of Scene.render_transparent
//                              (reference DiffRender.py:537-546, 420-432)
//   path_recompute_backward <- what autograd does for that graph w.r.t. Scene.vertices
// Plain C++ (also compiled by tests/hostsim); the gradient sink is a functor so the GPU
// can use float64 atomics.
#pragma once
#include "drt_shade.h"
#include "drt_traverse.h"

namespace drt {

struct TraceCtx {
    const Node4Q* nodes;
    const TriRec* tris;
    int n_tris;
    int32_t* slow_stack;
};

struct PathCtx {
    TraceCtx tc;
    const int32_t* faces;   // int32 [F,3]
    const double* verts;    // float64 [V,3]
    double ior_int, ior_ext;
};

DRT_HD void load_tri64(const PathCtx& c, int32_t face, d3& v0, d3& v1, d3& v2, int32_t (&vid)[3]) {
    vid[0] = c.faces[3 * (int64_t)face]; vid[1] = c.faces[3 * (int64_t)face + 1]; vid[2] = c.faces[3 * (int64_t)face + 2];
    v0 = load_d3(c.verts, vid[0]); v1 = load_d3(c.verts, vid[1]); v2 = load_d3(c.verts, vid[2]);
}

// Returns true when the path completes; out_o/out_d are then the exit ray.  f1 = face of the
// primary hit (-1 on a miss), f2 = second face, >= 0 only for completed paths.  No tape is
// kept across the traversals (it would pin ~200 VGPRs); the adjoint recomputes both bounces
// from (f1, f2) instead.
DRT_HD bool trace_path(const PathCtx& c, Stack& st, d3 o, d3 d, int32_t& f1, int32_t& f2, d3& out_o, d3& out_d) {
    f2 = -1;
    Hit h = traverse<false>(c.tc.nodes, c.tc.tris, c.tc.n_tris, to_f32(o), to_f32(d), st);
    f1 = h.face;
    if (h.face < 0) return false;
    d3 v0, v1, v2;
    int32_t vid[3];
    Bounce b;
    load_tri64(c, h.face, v0, v1, v2, vid);
    bounce_forward(o, d, v0, v1, v2, c.ior_ext, c.ior_int, b);
    if (b.tir) return false;
    d3 o2 = b.new_o, d2 = b.wt;
    h = traverse<false>(c.tc.nodes, c.tc.tris, c.tc.n_tris, to_f32(o2), to_f32(d2), st);
    if (h.face < 0) return false;
    const int32_t face2 = h.face;
    load_tri64(c, face2, v0, v1, v2, vid);
    bounce_forward(o2, d2, v0, v1, v2, c.ior_ext, c.ior_int, b);
    if (b.tir) return false;
    o2 = b.new_o; d2 = b.wt;
    h = traverse<true>(c.tc.nodes, c.tc.tris, c.tc.n_tris, to_f32(o2), to_f32(d2), st);
    if (h.face >= 0) return false;
    f2 = face2;
    out_o = o2; out_d = d2;
    return true;
}

// Recompute both bounces from the saved face ids, reverse them, hand the six vertex
// gradients to `add(vertex_id, d3)`.
template <typename Add>
DRT_HD void path_recompute_backward(const PathCtx& c, d3 o, d3 d, int32_t f1, int32_t f2, d3 g_ori, d3 g_dir, Add add) {
    d3 v0, v1, v2;
    int32_t vid1[3], vid2[3];
    Bounce b1, b2;
    load_tri64(c, f1, v0, v1, v2, vid1);
    bounce_forward(o, d, v0, v1, v2, c.ior_ext, c.ior_int, b1);
    load_tri64(c, f2, v0, v1, v2, vid2);
    bounce_forward(b1.new_o, b1.wt, v0, v1, v2, c.ior_ext, c.ior_int, b2);
    const d3 z{0.0, 0.0, 0.0};
    d3 ga = z, gb = z, gc = z, g_o, g_d;
    bounce_backward(b2, g_ori, g_dir, ga, gb, gc, g_o, g_d);
    add(vid2[0], ga); add(vid2[1], gb); add(vid2[2], gc);
    ga = z; gb = z; gc = z;
    d3 g_o0, g_d0;
    bounce_backward(b1, g_o, g_d, ga, gb, gc, g_o0, g_d0);
    add(vid1[0], ga); add(vid1[1], gb); add(vid1[2], gc);
}

}  // namespace drt
