// drt_tri.h -- the float32 ray/triangle test that defines the tracer's results.
//
// Replaces the arithmetic the reference leaves to OptiX Prime behind
// optix_mesh::intersect (reference optix_extend.cpp:29-57).  It is the float32
// transcription of the reference's own Moller-Trumbore (JIT_Dintersect,
// reference DiffRender.py:76-91): one rounding per operation, no contraction, so
// the GPU, the host-side unit tests and the CPU oracle agree bit for bit.
//     hit  <=>  u >= 0 && v >= 0 && u + v <= 1 && t > 0
// (det == 0 gives inf/NaN, which fail the comparisons.)
#pragma once
#include "drt_common.h"

namespace drt {

// Triangle record in BVH leaf order: 48 bytes, three 16-byte loads.
struct TriRec {
    float v0x, v0y, v0z;
    int32_t face;       // original face id
    float e1x, e1y, e1z;
    float pad0;
    float e2x, e2y, e2z;
    float pad1;
};

DRT_HD TriRec make_tri(f3 a, f3 b, f3 c, int32_t face) {
    TriRec r;
    r.v0x = a.x; r.v0y = a.y; r.v0z = a.z; r.face = face;
    r.e1x = b.x - a.x; r.e1y = b.y - a.y; r.e1z = b.z - a.z; r.pad0 = 0.f;
    r.e2x = c.x - a.x; r.e2y = c.y - a.y; r.e2z = c.z - a.z; r.pad1 = 0.f;
    return r;
}

// Returns true and sets t on a hit.
DRT_HD bool tri_hit(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float& t_out) {
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

}  // namespace drt
