// drt_traverse.h -- stack-based closest-hit / any-hit traversal of the LBVH.
//
// Replaces `query->execute` of OptiX Prime behind optix_mesh::intersect
// (reference optix_extend.cpp:33-45).  Contract (see oracle/tracer.c): closest
// hit = minimum t over all triangles passing drt_tri.h's test, equal t -> lowest
// face id; miss -> t = -1, face = -1.  The any-hit form answers only "is there a
// hit" (the occlusion test, reference DiffRender.py:426, and the silhouette
// probes, DiffRender.py:224, use nothing else).
//
// Box culling is conservative: slabs are evaluated with fma on padded boxes and a
// box is entered whenever its interval overlaps [0, best_t] (ties included), so the
// set of triangles tested always contains every triangle that could win.
#pragma once
#include "drt_lbvh.h"

namespace drt {

struct alignas(16) F4 {
    float x, y, z, w;
};

struct Hit {
    float t;
    int32_t face;
};

// Traversal stack: `depth_fast` entries in fast memory with a per-lane stride (LDS on
// the GPU), the rest in a per-thread overflow area.  LBVH height is bounded by
// 30 Morton bits + 32 index bits, so 96 entries always suffice.
struct Stack {
    int32_t* fast;      // &fast_mem[lane], entry k at fast[k * stride]
    int stride;
    int depth_fast;
    int32_t* slow;      // per-thread, contiguous
    int sp;
    DRT_HD void push(int32_t v) {
        if (sp < depth_fast) fast[sp * stride] = v; else slow[sp - depth_fast] = v;
        ++sp;
    }
    DRT_HD int32_t pop() {
        --sp;
        return sp < depth_fast ? fast[sp * stride] : slow[sp - depth_fast];
    }
    DRT_HD bool empty() const { return sp == 0; }
};
constexpr int kStackSlow = 72;

// Reciprocal direction for the slab test only.  A zero (or denormal-small) component would
// give inf and then inf - inf = NaN in the fma slab form, so it is replaced by +-2^-80: over
// any t a ray can reach that moves it by far less than the box padding, so culling stays
// conservative.  The triangle test always uses the true direction.
DRT_HD float safe_inv(float d) {
    const float eps = 8.27180613e-25f;   // 2^-80
    return 1.0f / (fabsf(d) > eps ? d : copysignf(eps, d));
}

template <bool ANY>
DRT_HD Hit traverse(const Node* __restrict__ nodes, const TriRec* __restrict__ tris, int n_tris,
                    f3 o, f3 d, Stack& st, uint32_t* visits = nullptr) {
    Hit best{INFINITY, -1};
    if (n_tris <= 0) return Hit{-1.0f, -1};
    const f3 inv{safe_inv(d.x), safe_inv(d.y), safe_inv(d.z)};
    const f3 oi{-o.x * inv.x, -o.y * inv.y, -o.z * inv.z};
    st.sp = 0;
    int32_t cur = 0;
    uint32_t nvis = 0;
    for (;;) {
        if (cur >= 0) {
            const F4* np = reinterpret_cast<const F4*>(nodes + cur);
            const F4 a = np[0], b = np[1], c = np[2], ch = np[3];
            ++nvis;
            // child 0
            float t0 = fmaf(a.x, inv.x, oi.x), t1 = fmaf(a.y, inv.x, oi.x);
            float tmin0 = fminf(t0, t1), tmax0 = fmaxf(t0, t1);
            t0 = fmaf(a.z, inv.y, oi.y); t1 = fmaf(a.w, inv.y, oi.y);
            tmin0 = fmaxf(tmin0, fminf(t0, t1)); tmax0 = fminf(tmax0, fmaxf(t0, t1));
            t0 = fmaf(c.x, inv.z, oi.z); t1 = fmaf(c.y, inv.z, oi.z);
            tmin0 = fmaxf(tmin0, fminf(t0, t1)); tmax0 = fminf(tmax0, fmaxf(t0, t1));
            tmin0 = fmaxf(tmin0, 0.0f); tmax0 = fminf(tmax0, best.t);
            // child 1
            t0 = fmaf(b.x, inv.x, oi.x); t1 = fmaf(b.y, inv.x, oi.x);
            float tmin1 = fminf(t0, t1), tmax1 = fmaxf(t0, t1);
            t0 = fmaf(b.z, inv.y, oi.y); t1 = fmaf(b.w, inv.y, oi.y);
            tmin1 = fmaxf(tmin1, fminf(t0, t1)); tmax1 = fminf(tmax1, fmaxf(t0, t1));
            t0 = fmaf(c.z, inv.z, oi.z); t1 = fmaf(c.w, inv.z, oi.z);
            tmin1 = fmaxf(tmin1, fminf(t0, t1)); tmax1 = fminf(tmax1, fmaxf(t0, t1));
            tmin1 = fmaxf(tmin1, 0.0f); tmax1 = fminf(tmax1, best.t);
            const bool h0 = tmin0 <= tmax0, h1 = tmin1 <= tmax1;
            int32_t c0, c1;
            memcpy(&c0, &ch.x, 4);
            memcpy(&c1, &ch.y, 4);
            if (h0 & h1) {
                const bool swap = tmin1 < tmin0;
                st.push(swap ? c0 : c1);
                cur = swap ? c1 : c0;
                continue;
            }
            if (h0) { cur = c0; continue; }
            if (h1) { cur = c1; continue; }
        } else {
            const F4* tp = reinterpret_cast<const F4*>(tris + (~cur));
            const F4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
            ++nvis;
            float t;
            if (tri_hit(o, d, f3{p0.x, p0.y, p0.z}, f3{p1.x, p1.y, p1.z}, f3{p2.x, p2.y, p2.z}, t)) {
                int32_t face;
                memcpy(&face, &p0.w, 4);
                if (ANY) {
                    if (visits) *visits = nvis;
                    return Hit{t, face};
                }
                if (t < best.t || (t == best.t && face < best.face)) { best.t = t; best.face = face; }
            }
        }
        if (st.empty()) break;
        cur = st.pop();
    }
    if (visits) *visits = nvis;
    if (best.face < 0) best.t = -1.0f;
    return best;
}

}  // namespace drt
