// drt_tri.h -- the float32 ray/triangle test that defines the tracer's results.
//
// Replaces the arithmetic the reference leaves to OptiX Prime behind
// optix_mesh::intersect (reference optix_extend.cpp:29-57).  It is the float32
// transcription of the reference's own Moller-Trumbore (JIT_Dintersect,
// reference DiffRender.py:76-91): one rounding per operation, no contraction, so
// the GPU, the host-side unit tests and the CPU oracle agree bit for bit.
//     hit  <=>  u >= 0 && v >= 0 && u + v <= 1 && t > 0  &&  the hit point lies in the triangle's bounding box grown by `margin` (hit_point_in_box)
// (det == 0 gives inf/NaN, which fail the comparisons.)
// The last condition is what makes "closest hit over ALL triangles" something a tree can compute: a ray that runs inside a
// triangle's plane has det = rounding noise, and u, v, t can then pass the first four tests with a "hit point" thousands of box
// paddings away from the triangle (found by tests/test_gpu_fuzz.py: one such ray in a thousand aimed along a triangle's plane).
// An exhaustive test reports that triangle, a tree never visits it.  With the condition every accepted hit lies inside its
// leaf's box -- margin = half the padding of the leaf boxes (drt_lbvh.h: pad_for_extent of the scene box), the other half
// covers the rounding of the slab test -- so tree, projection pass and exhaustive test return the same face by construction.
// For a hit with a well-conditioned t the point is within 1e-6 of the ray's length of the triangle, a twentieth of the margin
// of a scene seen from three extents away.
#pragma once
#include "drt_common.h"

namespace drt {

// Triangle record in BVH leaf order: 48 bytes, three 16-byte loads.
struct TriRec {
    float v0x, v0y, v0z;
    int32_t face;       // original face id
    float e1x, e1y, e1z;
    float margin;       // of the hit-point test (see above): travels with the record, so every consumer applies the same one
    float e2x, e2y, e2z;
    float pad1;
};

DRT_HD TriRec make_tri(f3 a, f3 b, f3 c, int32_t face, float margin) {
    TriRec r;
    r.v0x = a.x; r.v0y = a.y; r.v0z = a.z; r.face = face;
    r.e1x = b.x - a.x; r.e1y = b.y - a.y; r.e1z = b.z - a.z; r.margin = margin;
    r.e2x = c.x - a.x; r.e2y = c.y - a.y; r.e2z = c.z - a.z; r.pad1 = 0.f;
    return r;
}

// The hit point in the triangle's own frame, r = (o - v0) + t d, inside the box of (0, e1, e2) grown by
//     m = max(margin, 2^-18 * max|o - v0|),
// one rounding per operation, like the rest (oracle/hit_point.h is the same expression).  The second term lets the tolerance grow with
// the distance between the ray's origin and the triangle, as the float32 error of the reconstructed point does (about 1e-6 of the
// ray's length for a well-conditioned t): it takes over from 16 scene extents on, so a camera hundreds of extents away -- or a small
// object in a large scene box -- keeps its legitimate hits.  Up to 16 extents an accepted hit lies inside its leaf's box by
// construction (above); beyond, only a ray inside a triangle's plane can still be accepted by the exhaustive test with a point the
// tree's padding does not cover.
DRT_HD bool hit_point_in_box(f3 o, f3 d, float t, f3 v0, f3 e1, f3 e2, float margin) {
    const f3 s = o - v0;
    const float m = fmaxf(margin, fmaxf(fabsf(s.x), fmaxf(fabsf(s.y), fabsf(s.z))) * 0x1p-18f);
    const float rx = s.x + t * d.x;
    bool in = (rx + m >= fminf(0.0f, fminf(e1.x, e2.x))) & (rx - m <= fmaxf(0.0f, fmaxf(e1.x, e2.x)));
    const float ry = s.y + t * d.y;
    in &= (ry + m >= fminf(0.0f, fminf(e1.y, e2.y))) & (ry - m <= fmaxf(0.0f, fmaxf(e1.y, e2.y)));
    const float rz = s.z + t * d.z;
    in &= (rz + m >= fminf(0.0f, fminf(e1.z, e2.z))) & (rz - m <= fmaxf(0.0f, fmaxf(e1.z, e2.z)));
    return in;
}

// The first four conditions alone (the caller applies hit_point_in_box itself: drt_traverse.h).
DRT_HD bool tri_hit_mt(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float& t_out) {
    const f3 p = cross(d, e2);
    const float det = dot(e1, p);
    const float inv = 1.0f / det;
    const f3 s = o - v0;
    const float u = dot(s, p) * inv;
    const f3 q = cross(s, e1);
    const float v = dot(d, q) * inv;
    const float t = dot(e2, q) * inv;
    t_out = t;
    return (u >= 0.0f) & (v >= 0.0f) & (u + v <= 1.0f) & (t > 0.0f);
}

DRT_HD bool tri_hit(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float margin, float& t_out) {
    return tri_hit_mt(o, d, v0, e1, e2, t_out) && hit_point_in_box(o, d, t_out, v0, e1, e2, margin);
}
}  // namespace drt
