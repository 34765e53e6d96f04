// drt_lbvh.h -- per-item bodies of the on-GPU LBVH build (Morton code, Karras
// radix-tree hierarchy) and the node layout the traversal reads.
//
// Replaces the acceleration-structure build the reference gets from OptiX Prime
// (`model->setTriangles` + `model->update`, reference optix_extend.cpp:61-67),
// which it re-runs on every `update_vert` (optix_extend.cpp:23-27), i.e. every
// optimisation iteration (reference DiffRender.py:380, optim.py:203).
//
// The hierarchy only decides WHICH triangles a ray is tested against; results are
// defined by drt_tri.h.  Child boxes are padded (see pad_for_extent) so that box
// culling can never drop a triangle the float32 test would accept.
#pragma once
#include "drt_common.h"
#include "drt_tri.h"

namespace drt {

// Inner node, 64 bytes = four 16-byte loads.  Both child boxes live in the parent so
// one fetch decides both descents.  child >= 0: inner node index; child < 0: leaf,
// triangle slot = ~child (slots are in Morton order).
struct Node {
    float c0lox, c0hix, c0loy, c0hiy;
    float c1lox, c1hix, c1loy, c1hiy;
    float c0loz, c0hiz, c1loz, c1hiz;
    int32_t child0, child1, pad0, pad1;
};

struct Box {
    float lox, loy, loz, hix, hiy, hiz;
};

DRT_HD Box box_empty() { return {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}; }
DRT_HD Box box_union(Box a, Box b) {
    return {fminf(a.lox, b.lox), fminf(a.loy, b.loy), fminf(a.loz, b.loz),
            fmaxf(a.hix, b.hix), fmaxf(a.hiy, b.hiy), fmaxf(a.hiz, b.hiz)};
}
DRT_HD Box box_of_tri(f3 a, f3 b, f3 c, float pad) {
    Box r;
    r.lox = fminf(a.x, fminf(b.x, c.x)) - pad; r.hix = fmaxf(a.x, fmaxf(b.x, c.x)) + pad;
    r.loy = fminf(a.y, fminf(b.y, c.y)) - pad; r.hiy = fmaxf(a.y, fmaxf(b.y, c.y)) + pad;
    r.loz = fminf(a.z, fminf(b.z, c.z)) - pad; r.hiz = fmaxf(a.z, fmaxf(b.z, c.z)) + pad;
    return r;
}
DRT_HD bool box_contains(Box outer, Box inner) {
    return outer.lox <= inner.lox && outer.loy <= inner.loy && outer.loz <= inner.loz &&
           outer.hix >= inner.hix && outer.hiy >= inner.hiy && outer.hiz >= inner.hiz;
}

DRT_HD void node_set_child_box(Node& n, int slot, Box b) {
    if (slot == 0) {
        n.c0lox = b.lox; n.c0hix = b.hix; n.c0loy = b.loy; n.c0hiy = b.hiy; n.c0loz = b.loz; n.c0hiz = b.hiz;
    } else {
        n.c1lox = b.lox; n.c1hix = b.hix; n.c1loy = b.loy; n.c1hiy = b.hiy; n.c1loz = b.loz; n.c1hiz = b.hiz;
    }
}
DRT_HD Box node_child_box(const Node& n, int slot) {
    return slot == 0 ? Box{n.c0lox, n.c0loy, n.c0loz, n.c0hix, n.c0hiy, n.c0hiz}
                     : Box{n.c1lox, n.c1loy, n.c1loz, n.c1hix, n.c1hiy, n.c1hiz};
}

// Leaf boxes are grown by this much on every side.  The float32 Moller-Trumbore test
// accepts rays that pass a triangle's exact outline by a rounding-error margin
// (~1e-4 of the scene extent at most for the camera distances used); 2^-13 of the
// largest scene extent (0.024 mm on a 200 mm object) is two orders above that and
// also covers the rounding of the slab arithmetic in the traversal.
DRT_HD float pad_for_extent(float ext) { return ext * (1.0f / 8192.0f); }
// margin of the hit-point test of drt_tri.h: half the leaf padding (the other half is the slab test's)
DRT_HD float hit_margin(float pad) { return 0.5f * pad; }

DRT_HD uint32_t expand_bits10(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// 30-bit Morton code of the triangle centroid inside the scene box [lo, lo + 1/inv_ext].
DRT_HD uint32_t morton30(f3 a, f3 b, f3 c, f3 lo, f3 inv_ext) {
    const float third = 1.0f / 3.0f;
    float x = (((a.x + b.x) + c.x) * third - lo.x) * inv_ext.x;
    float y = (((a.y + b.y) + c.y) * third - lo.y) * inv_ext.y;
    float z = (((a.z + b.z) + c.z) * third - lo.z) * inv_ext.z;
    x = fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f);
    y = fminf(fmaxf(y * 1024.0f, 0.0f), 1023.0f);
    z = fminf(fmaxf(z * 1024.0f, 0.0f), 1023.0f);
    return (expand_bits10((uint32_t)x) << 2) | (expand_bits10((uint32_t)y) << 1) | expand_bits10((uint32_t)z);
}

// Which axis every bit of the 30-bit key splits (most significant first).  The plain x,y,z,x,y,z... interleave cuts each axis
// of the scene box in turn whatever its shape: on a 160 x 210 x 70 mm object the third cut halves the 70 mm side of cells that
// are still 80 x 105 mm across.  Here every bit halves the axis along which the cells are currently LONGEST (ties: x, y, z), so
// the cells stay as cubic as the box allows and long axes get more bits (Vinkler et al., "Extended Morton codes", HPG 2017:
// the axis-order part).  Same radix sort, same Karras hierarchy; the refracted rays of the benchmark need 9 % fewer node
// visits, the exit rays 4 % (tools/bvhq).
struct MortonPlan {
    uint8_t axis[30];    // axis of key bit 29 - k
    uint8_t pos[30];     // which bit of that axis' quantised coordinate
    uint8_t bits[3];     // bits per axis (sum 30)
    uint8_t pad[3];
};
DRT_HD MortonPlan morton_plan(float ex, float ey, float ez) {
    MortonPlan p;
    float e[3] = {ex > 0.f ? ex : 0.f, ey > 0.f ? ey : 0.f, ez > 0.f ? ez : 0.f};
    p.bits[0] = p.bits[1] = p.bits[2] = 0; p.pad[0] = p.pad[1] = p.pad[2] = 0;
    for (int k = 0; k < 30; ++k) {
        int a = -1;                       // (no axis takes more than 20 bits: a needle-shaped box)
        for (int c = 0; c < 3; ++c)
            if (p.bits[c] < 20 && (a < 0 || e[c] > e[a])) a = c;
        p.axis[k] = (uint8_t)a;
        ++p.bits[a];
        e[a] *= 0.5f;
    }
    uint8_t used[3] = {0, 0, 0};
    for (int k = 0; k < 30; ++k) { const int a = p.axis[k]; p.pos[k] = (uint8_t)(p.bits[a] - 1 - used[a]); ++used[a]; }
    return p;
}
// Key of the centre of the triangle's BOUNDING BOX inside the scene box [lo, lo + 1/inv_ext] under `plan`.  (Not the vertex
// centroid: the two triangles of a quad -- the meshes here come from marching-cubes-like extractions and midpoint subdivision --
// have different centroids but the same box centre, get the same key and end up in the same <= 4-triangle leaf.  Measured with
// tools/bvhq on hand / mouse x4 / horse x4 / monkey: 5-8 % fewer node visits on top of the axis order's 1-10 %.)
DRT_HD uint32_t morton_key(f3 a, f3 b, f3 c, f3 lo, f3 inv_ext, const MortonPlan& plan) {
    const float x = (0.5f * (fminf(a.x, fminf(b.x, c.x)) + fmaxf(a.x, fmaxf(b.x, c.x))) - lo.x) * inv_ext.x;
    const float y = (0.5f * (fminf(a.y, fminf(b.y, c.y)) + fmaxf(a.y, fmaxf(b.y, c.y))) - lo.y) * inv_ext.y;
    const float z = (0.5f * (fminf(a.z, fminf(b.z, c.z)) + fmaxf(a.z, fmaxf(b.z, c.z))) - lo.z) * inv_ext.z;
    const float sx = (float)(1u << plan.bits[0]), sy = (float)(1u << plan.bits[1]), sz = (float)(1u << plan.bits[2]);
    const uint32_t qx = (uint32_t)fminf(fmaxf(x * sx, 0.0f), sx - 1.0f);
    const uint32_t qy = (uint32_t)fminf(fmaxf(y * sy, 0.0f), sy - 1.0f);
    const uint32_t qz = (uint32_t)fminf(fmaxf(z * sz, 0.0f), sz - 1.0f);
    uint32_t key = 0;
    for (int k = 0; k < 30; ++k) {
        const uint32_t q = plan.axis[k] == 0 ? qx : (plan.axis[k] == 1 ? qy : qz);
        key = (key << 1) | ((q >> plan.pos[k]) & 1u);
    }
    return key;
}

DRT_HD int clz32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clz((int)v);
#else
    return v ? __builtin_clz(v) : 32;
#endif
}

// Length of the common prefix of (key_i, i) and (key_j, j); -1 when j is out of range.
DRT_HD int lbvh_delta(const uint32_t* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint32_t a = keys[i], b = keys[j];
    if (a != b) return clz32(a ^ b);
    return 32 + clz32((uint32_t)i ^ (uint32_t)j);
}

// Karras 2012 ("Maximizing parallelism in the construction of BVHs, octrees and k-d
// trees"): children of inner node i over n sorted keys.  Returns child encodings
// (>= 0 inner, < 0 leaf ~slot).
DRT_HD void lbvh_children(const uint32_t* keys, int n, int i, int32_t& left, int32_t& right, int32_t& range_lo, int32_t& range_hi) {
    const int d = (lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) / 2;
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    left = (lo == gamma) ? ~gamma : gamma;
    right = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    range_lo = lo;
    range_hi = hi;   // node i covers the sorted triangle slots [lo, hi]
}

// ---- 4-wide tree read by the traversal ------------------------------------------------------
// The binary radix tree is collapsed by skipping every other level (a node at even depth adopts
// its grandchildren) and by turning every subtree of <= kLeafMax triangles into one leaf; such
// a subtree covers a CONTIGUOUS run of Morton-sorted triangle slots, so a leaf is (first, count).
// Fewer, fatter steps: traversal on MI355X is bound by the latency of the dependent node
// fetches, not by ALU or bytes.  One node = 128 bytes = one L2 line, SoA so that each 16-byte
// load brings one bound of all four children.
// Leaf size: ONE triangle.  The best size follows the price of an inner visit: per step on MI355X (horse x4) 8 -> 3.08 ms, 4 -> 2.96,
// 2 -> 2.91, 1 -> 2.92 while an inner visit was ~158 vector instructions (with the round-1 centroid keys 4 was best); at 97 instructions
// a visit is cheaper than the second triangle test a two-triangle leaf makes most rays pay for: k_trace<closest> 0.379 -> 0.364 ms per
// launch, k_trace<any> 0.182 -> 0.171, fused step -2.2 %, drop-in step -0.6 % (mouse -1.5 %, monkey and hand unchanged).  With keys taken
// at the box centre the two halves of a quad still share a key and therefore a parent.  (-DDRT_LEAF_BITS=1, 2: leaves of <= 2, <= 4.)
#ifndef DRT_LEAF_BITS
#define DRT_LEAF_BITS 0
#endif
constexpr int kLeafBits = DRT_LEAF_BITS, kLeafMax = 1 << kLeafBits;

struct alignas(16) Node4 {
    float lox[4], hix[4], loy[4], hiy[4], loz[4], hiz[4];
    int32_t child[4];   // >= 0: Node4 index; < 0: leaf, ~child = (first_slot << kLeafBits) | (count - 1); kEmptyChild: none
    int32_t pad[4];
};
constexpr int32_t kEmptyChild = INT32_MIN;   // flagged in the child reference (an inverted float box still passes a slab test: inf - inf); the QUANTISED form of an empty slot is an inverted byte interval (quantize_axis)

DRT_HD int32_t leaf_ref(int first, int count) { return ~((first << kLeafBits) | (count - 1)); }

DRT_HD bool wide_leaf_of(int32_t c, const int32_t* range_lo, const int32_t* range_hi, int& first, int& count) {
    if (c < 0) { first = ~c; count = 1; return true; }
    const int cnt = range_hi[c] - range_lo[c] + 1;
    if (cnt <= kLeafMax) { first = range_lo[c]; count = cnt; return true; }
    return false;
}

DRT_HD void node4_clear(Node4& o) {
    DRT_UNROLL
    for (int k = 0; k < 4; ++k) {
        o.lox[k] = o.loy[k] = o.loz[k] = INFINITY;
        o.hix[k] = o.hiy[k] = o.hiz[k] = -INFINITY;
        o.child[k] = kEmptyChild;
        o.pad[k] = 0;
    }
}
DRT_HD void node4_set(Node4& o, int k, Box b, int32_t child) {
    o.lox[k] = b.lox; o.hix[k] = b.hix; o.loy[k] = b.loy; o.hiy[k] = b.hiy; o.loz[k] = b.loz; o.hiz[k] = b.hiz;
    o.child[k] = child;
}
DRT_HD Box node4_box(const Node4& o, int k) { return Box{o.lox[k], o.loy[k], o.loz[k], o.hix[k], o.hiy[k], o.hiz[k]}; }

// Stored form of a wide node: 64 bytes = four 16-byte loads instead of seven.  The traversal is bound
// by the number of per-lane vector-memory requests (each lane walks its own node, so every 16-byte
// load of a wave touches 64 different lines), not by ALU: child bounds are therefore quantised to
// 8 bits on the grid (origin, scale) of the node's own box, rounded OUTWARDS, and decoded with spare
// ALU (lo = origin + q * scale).  Boxes only steer the traversal; results are defined by drt_tri.h.
// The three scales are stored multiplied by 2^24 (kQScaleUp): the traversal feeds the bound bytes to v_fma_mix_f32 as float16
// subnormals (byte * 2^-24, exact), one instruction per plane instead of a convert and an fma, and the factor cancels exactly.
struct alignas(16) Node4Q {
    float ox, oy, oz, sx;                     // chunk 0: grid origin, x scale * 2^24
    float sy, sz;                             // chunk 1: y, z scale * 2^24,
    uint32_t qlox, qloy;                      //          byte k of each q word = child k
    uint32_t qloz, qhix, qhiy, qhiz;          // chunk 2
    int32_t child[4];                         // chunk 3: as Node4
};

constexpr float kQScaleUp = 16777216.0f, kQScaleDown = 1.0f / 16777216.0f;   // 2^24, 2^-24
DRT_HD float q_byte(uint32_t word, int k) { return (float)((word >> (8 * k)) & 255u); }
DRT_HD Box node4q_box(const Node4Q& n, int k) {
    const float sx = n.sx * kQScaleDown, sy = n.sy * kQScaleDown, sz = n.sz * kQScaleDown;   // exact
    return Box{fmaf(q_byte(n.qlox, k), sx, n.ox), fmaf(q_byte(n.qloy, k), sy, n.oy), fmaf(q_byte(n.qloz, k), sz, n.oz),
               fmaf(q_byte(n.qhix, k), sx, n.ox), fmaf(q_byte(n.qhiy, k), sy, n.oy), fmaf(q_byte(n.qhiz, k), sz, n.oz)};
}

// One axis: grid (origin strictly below every child's lo, 253 steps up to the largest hi) and the
// outward-rounded bytes of the four children.
DRT_HD void quantize_axis(const float lo[4], const float hi[4], const bool valid[4], float& o, float& s, uint32_t& qlo, uint32_t& qhi) {
    float mn = INFINITY, mx = -INFINITY;
    DRT_UNROLL
    for (int k = 0; k < 4; ++k)
        if (valid[k]) { mn = fminf(mn, lo[k]); mx = fmaxf(mx, hi[k]); }
    if (!(mn <= mx)) { mn = 0.0f; mx = 0.0f; }
    o = mn - (fabsf(mn) * 9.5367431640625e-7f + 1e-30f);
    s = ((mx - o) / 253.0f) * 1.00000095367431640625f;
    if (!(s > 0.0f)) s = 1e-30f;
    qlo = 0; qhi = 0;
    DRT_UNROLL
    for (int k = 0; k < 4; ++k) {
        int a = 255, b = 0;      // no child: an INVERTED interval on every axis -- near plane beyond far plane for any direction, so the slab test itself rejects the slot
        if (valid[k]) {
            a = (int)floorf((lo[k] - o) / s);
            a = a < 0 ? 0 : (a > 255 ? 255 : a);
            while (a > 0 && fmaf((float)a, s, o) > lo[k]) --a;
            b = (int)ceilf((hi[k] - o) / s);
            b = b < 0 ? 0 : (b > 255 ? 255 : b);
            while (b < 255 && fmaf((float)b, s, o) < hi[k]) ++b;
        }
        qlo |= (uint32_t)a << (8 * k);
        qhi |= (uint32_t)b << (8 * k);
    }
}

DRT_HD Node4Q node4_quantize(const Node4& f) {
    Node4Q q;
    bool valid[4];
    DRT_UNROLL
    for (int k = 0; k < 4; ++k) { valid[k] = f.child[k] != kEmptyChild; q.child[k] = f.child[k]; }
    quantize_axis(f.lox, f.hix, valid, q.ox, q.sx, q.qlox, q.qhix);
    quantize_axis(f.loy, f.hiy, valid, q.oy, q.sy, q.qloy, q.qhiy);
    quantize_axis(f.loz, f.hiz, valid, q.oz, q.sz, q.qloz, q.qhiz);
    q.sx *= kQScaleUp; q.sy *= kQScaleUp; q.sz *= kQScaleUp;
    return q;
}

DRT_HD float box_area(const Box& b) {
    const float dx = b.hix - b.lox, dy = b.hiy - b.loy, dz = b.hiz - b.loz;
    return dx * dy + dy * dz + dz * dx;
}

// Wide node rooted at binary node i (more than kLeafMax triangles below it, or the root).  Starts from the
// two binary children and twice opens the open-able child with the LARGEST surface area (surface-area
// heuristic: the box a random ray is most likely to enter is the one worth resolving inside this node), so
// a wide node spans one to three binary levels instead of always exactly two.  Wide nodes keep the index of
// their binary root; any binary node may become one.
#ifndef DRT_GREEDY_COLLAPSE
#define DRT_GREEDY_COLLAPSE 1
#endif
// (The four slots are named members, not arrays, and every pick or placement is a compare-and-select chain: with indexed arrays
// the device kernel kept them in scratch memory.)
// (value selects, field by field: `cond ? box_x : box_y` on structs selects an ADDRESS and keeps both in memory)
DRT_HD int32_t sel(bool c, int32_t x, int32_t y) { return c ? x : y; }
DRT_HD bool sel(bool c, bool x, bool y) { return c ? x : y; }
DRT_HD Box sel(bool c, const Box& x, const Box& y) {
    return Box{c ? x.lox : y.lox, c ? x.loy : y.loy, c ? x.loz : y.loz, c ? x.hix : y.hix, c ? x.hiy : y.hiy, c ? x.hiz : y.hiz};
}
template <class T>
struct Slots4 {
    T a, b, c, d;
    DRT_HD T get(int i) const { return sel(i == 0, a, sel(i == 1, b, sel(i == 2, c, d))); }
    DRT_HD void put(int i, const T& v) {
        a = sel(i == 0, v, a);
        b = sel(i == 1, v, b);
        c = sel(i == 2, v, c);
        d = sel(i == 3, v, d);
    }
};
DRT_HD void collapse4(const Node* bin, const int32_t* range_lo, const int32_t* range_hi, int n_tris, int i, Node4& out) {
    node4_clear(out);
    if (n_tris <= kLeafMax) {   // whole mesh in one leaf under the root
        node4_set(out, 0, box_union(node_child_box(bin[0], 0), node_child_box(bin[0], 1)), leaf_ref(0, n_tris));
        return;
    }
    Slots4<int32_t> ref{0, 0, 0, 0};     // >= 0: binary node to descend into; < 0: finished leaf reference
    Slots4<Box> box{box_empty(), box_empty(), box_empty(), box_empty()};
    Slots4<bool> open{false, false, false, false};
    int k = 2, first, count;
    const Node root = bin[i];
    {
        bool op = !wide_leaf_of(root.child0, range_lo, range_hi, first, count);
        box.a = node_child_box(root, 0); open.a = op; ref.a = op ? root.child0 : leaf_ref(first, count);
        op = !wide_leaf_of(root.child1, range_lo, range_hi, first, count);
        box.b = node_child_box(root, 1); open.b = op; ref.b = op ? root.child1 : leaf_ref(first, count);
    }
    DRT_UNROLL
    for (int round = 0; round < 2; ++round) {
        int pick = -1;
#if DRT_GREEDY_COLLAPSE
        float best = -1.0f;
        DRT_UNROLL
        for (int j = 0; j < 4; ++j)
            if (j < k && open.get(j)) { const float a = box_area(box.get(j)); if (a > best) { best = a; pick = j; } }
#else
        if (open.a && ref.a == root.child0) pick = 0;
        else if (open.b && ref.b == root.child1) pick = 1;
#endif
        if (pick >= 0) {
            const Node opened = bin[ref.get(pick)];
            // the opened child's two children take its slot and the next free one
            bool op = !wide_leaf_of(opened.child0, range_lo, range_hi, first, count);
            box.put(pick, node_child_box(opened, 0)); open.put(pick, op); ref.put(pick, op ? opened.child0 : leaf_ref(first, count));
            op = !wide_leaf_of(opened.child1, range_lo, range_hi, first, count);
            box.put(k, node_child_box(opened, 1)); open.put(k, op); ref.put(k, op ? opened.child1 : leaf_ref(first, count));
            ++k;
        }
    }
    node4_set(out, 0, box.a, ref.a);
    node4_set(out, 1, box.b, ref.b);
    if (k > 2) node4_set(out, 2, box.c, ref.c);
    if (k > 3) node4_set(out, 3, box.d, ref.d);
}

}  // namespace drt
