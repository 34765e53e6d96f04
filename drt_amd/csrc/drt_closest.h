// drt_closest.h -- closest point of a triangle mesh to a query point, on the same quantised 4-wide
// LBVH the ray traversal reads.
//
// This is the query behind the reference's acceptance metric: "average per-vertex distance (Hausdorff
// Distance)" of the reconstruction against the scanned mesh, which the reference delegates to
// meshlabserver (reference README.md:11; no code in the repository).  SURVEY.md section 8f row 2.
//
// Contract: dist2 = min over faces of |p - closest_on_triangle(p, face)|^2, evaluated in float64 on
// the tracer's float32 vertex positions (promoted exactly).  `face` is a face attaining that minimum
// (which one, when an edge or vertex shared by several faces is the closest feature, is not defined).
// The box bound is evaluated in float64 on the decoded float32 boxes (which contain their triangles,
// padded) with a 1e-12 relative slack, so pruning never discards the minimiser.
#pragma once
#include "drt_lbvh.h"
#include "drt_traverse.h"

namespace drt {

DRT_HD d3 closest_on_segment(d3 p, d3 a, d3 b) {
    const d3 ab = b - a;
    const double ee = dot(ab, ab);
    if (!(ee > 0.0)) return a;
    const double t = fmin(fmax(dot(p - a, ab) / ee, 0.0), 1.0);
    return a + ab * t;
}

// Closest point of triangle abc to p, by the Voronoi region of p (vertex, edge or face region).  A triangle without area -- two equal
// indices, three points on a line -- is the segment (or point) its vertices span: the closest of its three edges.  (The region
// formulas divide by edge lengths and by the area; face [a, a, c] used to come out as NaN and count for nothing, found by
// tests/test_gpu_fuzz.py.  In a closed mesh the edges of such a face belong to its neighbours as well, so nothing changed there.)
DRT_HD d3 closest_on_triangle(d3 p, d3 a, d3 b, d3 c) {
    const d3 ab = b - a, ac = c - a, ap = p - a;
    {
        const d3 n = cross(ab, ac);
        if (!(dot(n, n) > 1e-24 * (dot(ab, ab) * dot(ac, ac)))) {
            const d3 q1 = closest_on_segment(p, a, b), q2 = closest_on_segment(p, a, c), q3 = closest_on_segment(p, b, c);
            const d3 r1 = p - q1, r2 = p - q2, r3 = p - q3;
            const double e1 = dot(r1, r1), e2 = dot(r2, r2), e3 = dot(r3, r3);
            return e1 <= e2 ? (e1 <= e3 ? q1 : q3) : (e2 <= e3 ? q2 : q3);
        }
    }
    const double d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.0 && d2 <= 0.0) return a;
    const d3 bp = p - b;
    const double d3_ = dot(ab, bp), d4 = dot(ac, bp);
    if (d3_ >= 0.0 && d4 <= d3_) return b;
    const double vc = d1 * d4 - d3_ * d2;
    if (vc <= 0.0 && d1 >= 0.0 && d3_ <= 0.0) return a + ab * (d1 / (d1 - d3_));
    const d3 cp = p - c;
    const double d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.0 && d5 <= d6) return c;
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) return a + ac * (d2 / (d2 - d6));
    const double va = d3_ * d6 - d5 * d4;
    if (va <= 0.0 && (d4 - d3_) >= 0.0 && (d5 - d6) >= 0.0) return b + (c - b) * ((d4 - d3_) / ((d4 - d3_) + (d5 - d6)));
    const double denom = 1.0 / (va + vb + vc);
    return a + ab * (vb * denom) + ac * (vc * denom);
}

struct Closest {
    double dist2;
    int32_t face;
    d3 point;
};

DRT_HD double box_dist2(const Box& b, d3 p) {
    const double gx = fmax(fmax((double)b.lox - p.x, p.x - (double)b.hix), 0.0);
    const double gy = fmax(fmax((double)b.loy - p.y, p.y - (double)b.hiy), 0.0);
    const double gz = fmax(fmax((double)b.loz - p.z, p.z - (double)b.hiz), 0.0);
    return (gx * gx + gy * gy) + gz * gz;
}

// (start_dist2: an upper bound the caller believes in -- a triangle must be STRICTLY closer to be reported; face stays -1 when none is, and
// the caller asks again without a bound.  A search that starts bounded opens a fraction of the nodes and, when it finds anything, finds what
// the unbounded one finds: same visiting order, same arithmetic, same winner among ties.)
template <class STACK>
DRT_HD Closest closest_point(const Node4Q* __restrict__ nodes, const TriRec* __restrict__ tris, int n_tris,
                             const int32_t* __restrict__ faces, const float* __restrict__ verts, d3 p, STACK& st, double start_dist2 = INFINITY) {
    Closest best{start_dist2, -1, d3{0.0, 0.0, 0.0}};
    if (n_tris <= 0) return best;
    constexpr double kSlack = 1.0 - 1e-12;
    st.sp = 0;
    int32_t cur = 0;
    for (;;) {
        if (cur >= 0) {
            const Node4Q n = nodes[cur];
            const int32_t ch[4] = {n.child[0], n.child[1], n.child[2], n.child[3]};
            double lb[4];
            int order[4] = {0, 1, 2, 3};
            for (int k = 0; k < 4; ++k) lb[k] = ch[k] == kEmptyChild ? INFINITY : box_dist2(node4q_box(n, k), p);
            // nearest child first: insertion sort of four keys
            for (int i = 1; i < 4; ++i)
                for (int j = i; j > 0 && lb[order[j]] < lb[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
            for (int i = 3; i >= 1; --i)
                if (lb[order[i]] < INFINITY && lb[order[i]] * kSlack <= best.dist2) st.push(ch[order[i]]);
            if (lb[order[0]] < INFINITY && lb[order[0]] * kSlack <= best.dist2) { cur = ch[order[0]]; continue; }
        } else {
            const int32_t ref = ~cur;
            const int first = ref >> kLeafBits, count = (ref & (kLeafMax - 1)) + 1;
            for (int j = 0; j < count; ++j) {
                const int32_t face = tris[first + j].face;
                const int32_t i0 = faces[3 * face], i1 = faces[3 * face + 1], i2 = faces[3 * face + 2];
                const d3 a{verts[3 * i0], verts[3 * i0 + 1], verts[3 * i0 + 2]};
                const d3 b{verts[3 * i1], verts[3 * i1 + 1], verts[3 * i1 + 2]};
                const d3 c{verts[3 * i2], verts[3 * i2 + 1], verts[3 * i2 + 2]};
                const d3 q = closest_on_triangle(p, a, b, c);
                const d3 r = p - q;
                const double dd = dot(r, r);
                if (dd < best.dist2) best = Closest{dd, face, q};      // (NaN -- non-finite vertices -- never wins)
            }
        }
        if (st.empty()) break;
        cur = st.pop();
    }
    return best;
}

// Is any triangle within `radius` of p -- the verdict  sqrt(closest_point(...).dist2) <= radius  without finishing the search: the bound
// starts at the radius (whole subtrees farther away are never opened) and the first triangle inside ends it.  The remesher's
// surface-distance rule asks only this, tens of thousands of times per round; a full closest-point descent opens about five times the
// nodes, and the query is latency-bound (one dependent node load per step).  Same per-triangle arithmetic as closest_point: the two agree
// on every input.
template <class STACK>
DRT_HD bool within_distance(const Node4Q* __restrict__ nodes, const TriRec* __restrict__ tris, int n_tris, const int32_t* __restrict__ faces,
                            const float* __restrict__ verts, d3 p, double radius, STACK& st) {
    if (n_tris <= 0 || !(radius >= 0.0)) return false;
    const double bound = radius * radius * (1.0 + 1e-12);       // (a triangle with sqrt(dd) <= radius has its box bound below this, slack included)
    constexpr double kSlack = 1.0 - 1e-12;
    st.sp = 0;
    int32_t cur = 0;
    for (;;) {
        if (cur >= 0) {
            const Node4Q n = nodes[cur];
            const int32_t ch[4] = {n.child[0], n.child[1], n.child[2], n.child[3]};
            double lb[4];
            int order[4] = {0, 1, 2, 3};
            for (int k = 0; k < 4; ++k) lb[k] = ch[k] == kEmptyChild ? INFINITY : box_dist2(node4q_box(n, k), p);
            for (int i = 1; i < 4; ++i)
                for (int j = i; j > 0 && lb[order[j]] < lb[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
            for (int i = 3; i >= 1; --i)
                if (lb[order[i]] * kSlack <= bound) st.push(ch[order[i]]);           // (an empty child's INFINITY never passes)
            if (lb[order[0]] * kSlack <= bound) { cur = ch[order[0]]; continue; }
        } else {
            const int32_t ref = ~cur;
            const int first = ref >> kLeafBits, count = (ref & (kLeafMax - 1)) + 1;
            for (int j = 0; j < count; ++j) {
                const int32_t face = tris[first + j].face;
                const int32_t i0 = faces[3 * face], i1 = faces[3 * face + 1], i2 = faces[3 * face + 2];
                const d3 a{verts[3 * i0], verts[3 * i0 + 1], verts[3 * i0 + 2]};
                const d3 b{verts[3 * i1], verts[3 * i1 + 1], verts[3 * i1 + 2]};
                const d3 c{verts[3 * i2], verts[3 * i2 + 1], verts[3 * i2 + 2]};
                const d3 r = p - closest_on_triangle(p, a, b, c);
                if (sqrt(dot(r, r)) <= radius) return true;
            }
        }
        if (st.empty()) break;
        cur = st.pop();
    }
    return false;
}

}  // namespace drt
