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

// One slab test against child k of a wide node, given its six bounds.
DRT_HD float slab4(float lox, float hix, float loy, float hiy, float loz, float hiz, f3 inv, f3 oi, float best_t, bool& hit) {
    float t0 = fmaf(lox, inv.x, oi.x), t1 = fmaf(hix, inv.x, oi.x);
    float tmin = fminf(t0, t1), tmax = fmaxf(t0, t1);
    t0 = fmaf(loy, inv.y, oi.y); t1 = fmaf(hiy, inv.y, oi.y);
    tmin = fmaxf(tmin, fminf(t0, t1)); tmax = fminf(tmax, fmaxf(t0, t1));
    t0 = fmaf(loz, inv.z, oi.z); t1 = fmaf(hiz, inv.z, oi.z);
    tmin = fmaxf(tmin, fminf(t0, t1)); tmax = fminf(tmax, fmaxf(t0, t1));
    tmin = fmaxf(tmin, 0.0f);
    hit = tmin <= fminf(tmax, best_t);
    return tmin;
}

DRT_HD void sort2(float& ka, int32_t& va, float& kb, int32_t& vb) {
    if (kb < ka) { const float k = ka; ka = kb; kb = k; const int32_t v = va; va = vb; vb = v; }
}

template <bool ANY>
DRT_HD Hit traverse(const Node4* __restrict__ nodes, const TriRec* __restrict__ tris, int n_tris,
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
            const F4 lx = np[0], hx = np[1], ly = np[2], hy = np[3], lz = np[4], hz = np[5], chf = np[6];
            ++nvis;
            int32_t c0, c1, c2, c3;
            memcpy(&c0, &chf.x, 4); memcpy(&c1, &chf.y, 4); memcpy(&c2, &chf.z, 4); memcpy(&c3, &chf.w, 4);
            bool h0, h1, h2, h3;
            float k0 = slab4(lx.x, hx.x, ly.x, hy.x, lz.x, hz.x, inv, oi, best.t, h0);
            float k1 = slab4(lx.y, hx.y, ly.y, hy.y, lz.y, hz.y, inv, oi, best.t, h1);
            float k2 = slab4(lx.z, hx.z, ly.z, hy.z, lz.z, hz.z, inv, oi, best.t, h2);
            float k3 = slab4(lx.w, hx.w, ly.w, hy.w, lz.w, hz.w, inv, oi, best.t, h3);
            h0 &= c0 != kEmptyChild; h1 &= c1 != kEmptyChild; h2 &= c2 != kEmptyChild; h3 &= c3 != kEmptyChild;
            const int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
            if (nh > 0) {
                // misses sort to the end (key = +inf); 5-comparator network; visit nearest, push the rest far-first
                k0 = h0 ? k0 : INFINITY; k1 = h1 ? k1 : INFINITY; k2 = h2 ? k2 : INFINITY; k3 = h3 ? k3 : INFINITY;
                sort2(k0, c0, k1, c1); sort2(k2, c2, k3, c3); sort2(k0, c0, k2, c2); sort2(k1, c1, k3, c3); sort2(k1, c1, k2, c2);
                if (nh > 3) st.push(c3);
                if (nh > 2) st.push(c2);
                if (nh > 1) st.push(c1);
                cur = c0;
                continue;
            }
        } else {
            const int32_t ref = ~cur;
            const int first = ref >> 2, count = (ref & 3) + 1;
            ++nvis;
            for (int j = 0; j < count; ++j) {
                const F4* tp = reinterpret_cast<const F4*>(tris + first + j);
                const F4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
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
        }
        if (st.empty()) break;
        cur = st.pop();
    }
    if (visits) *visits = nvis;
    if (best.face < 0) best.t = -1.0f;
    return best;
}

}  // namespace drt
