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
// the GPU), the rest in a per-thread overflow area.  At most three children are postponed per level
// of the wide tree and its depth is bounded by the binary height <= 64 (30 Morton bits + 32 index bits,
// one prefix bit per level), so depth_fast + overflow area = 192 entries always suffice (drt_scene.h).
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
    // Store unconditionally, advance only when `pred`: a dead store above the top of the stack is
    // harmless and keeps the hot loop free of divergent branches.
    DRT_HD void push_if(int32_t v, bool pred) {
        if (sp < depth_fast) fast[sp * stride] = v; else if (pred) slow[sp - depth_fast] = v;
        sp += pred ? 1 : 0;
    }
    DRT_HD void after_pushes() {}
    DRT_HD void reset() { sp = 0; }
    DRT_HD int32_t pop() {
        --sp;
        return sp < depth_fast ? fast[sp * stride] : slow[sp - depth_fast];
    }
    DRT_HD bool empty() const { return sp == 0; }
};
constexpr int kStackSlow = 192;

// Stack of the persistent GPU traversal: fast memory only, no overflow area and therefore no
// divergent slow path (and no flat loads) in the hot loop.  A closest-hit visit issues FOUR stores (one per child slot, the stack
// pointer advancing only past the hit children that are not the nearest: at most three), so four spare entries above `depth` take the
// stores of a visit that starts with a legal stack (sp <= depth: the last store lands at most at depth + 3) without a bound check
// per push (two VALU instructions per push: predicated increment, address = the stack pointer itself); ONE check per visit
// (`after_pushes`) sets `overflow`, the kernel then abandons that ray and a second pass re-traces it with the spilling Stack above.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) int32_t* FastPtr;      // 32-bit LDS address arithmetic
#else
typedef int32_t* FastPtr;
#endif
// -DDRT_CHECK=1 (a debug build of the library, tests/test_gpu_parity.py::test_lds_stack_invariants_hold_in_a_checked_build): the
// argument above -- "a visit that starts with a legal stack stores at most at row depth + 3" -- is ASSERTED at every store, pops are
// checked against underflow, and k_trace surrounds its stack rows with poisoned guard rows it validates before it exits.  Violations
// are counted on the device (drt_trace_kernel.h) and read through drt_check_violations().
#if defined(DRT_CHECK) && defined(__HIP_DEVICE_COMPILE__)
__device__ void drt_check_fail(int what);
#define DRT_DEV_ASSERT(cond, what) do { if (!(cond)) drt_check_fail(what); } while (0)
#else
#define DRT_DEV_ASSERT(cond, what) ((void)0)
#endif
enum { kCheckPushRow = 0, kCheckPopUnderflow = 1, kCheckGuardRow = 2, kCheckVisitStart = 3 };
struct FastStack {
    FastPtr base;       // &fast_mem[lane]; entry k at base[k * stride]; depth + 4 entries allocated
    FastPtr top;        // next free entry (the stack pointer IS the address: no shift-and-or per push)
    int stride;
    int depth;          // usable entries [0, depth); entries depth .. depth + 3 only ever hold the stores of an overflowing visit
    bool overflow;
    DRT_HD void reset() {
        top = base;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(top));       // an address of its own, not "base + offset" re-added at every push
#endif
    }
    DRT_HD void push_if(int32_t v, bool pred) {
        DRT_DEV_ASSERT(top < base + (depth + 4) * stride, kCheckPushRow);        // the store lands inside the rows this lane owns
        *top = v;
        top += pred ? stride : 0;
    }
    DRT_HD void after_pushes() { overflow |= top > base + depth * stride; }
    DRT_HD int32_t pop() {
        DRT_DEV_ASSERT(top > base, kCheckPopUnderflow);
        top -= stride;
        return *top;
    }
    DRT_HD bool empty() const { return top == base; }
};

// Reciprocal direction for the slab test only.  A zero (or denormal-small) component would
// give inf and then inf - inf = NaN in the fma slab form, so it is replaced by +-2^-80: over
// any t a ray can reach that moves it by far less than the box padding, so culling stays
// conservative.  The triangle test always uses the true direction.
DRT_HD float safe_inv(float d) {
    const float eps = 8.27180613e-25f;   // 2^-80
    return 1.0f / (fabsf(d) > eps ? d : copysignf(eps, d));
}

DRT_HD uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// Resumable traversal: the per-ray state lives in registers (+ the lane's stack), one call to
// trav_step visits one node (inner or leaf).  The persistent kernels interleave steps of many rays
// and hand a finished lane a new ray without leaving the loop.
struct TravState {
    f3 o, d, inv, oi;
    bool px, py, pz;      // direction component >= 0: which bound of a child is its near plane (lane masks in scalar registers on the GPU)
    int32_t cur;
    float best_t;
    int32_t best_face;
    int32_t best_slot;    // record of best_face in `tris` (trav_leaf<ANY, true> only: see there)
};

template <class STACK>
DRT_HD void trav_init(TravState& s, STACK& st, f3 o, f3 d) {
    s.o = o; s.d = d;
    s.inv = f3{safe_inv(d.x), safe_inv(d.y), safe_inv(d.z)};
    s.oi = f3{-o.x * s.inv.x, -o.y * s.inv.y, -o.z * s.inv.z};
    s.px = s.inv.x >= 0.0f; s.py = s.inv.y >= 0.0f; s.pz = s.inv.z >= 0.0f;
    s.cur = 0;
    s.best_t = INFINITY;
    s.best_face = -1;
    s.best_slot = -1;
    st.reset();
}

// Pop the next node; returns true when the stack is empty (ray finished).
template <class STACK>
DRT_HD bool trav_pop(TravState& s, STACK& st) {
    if (st.empty()) return true;
    s.cur = st.pop();
    return false;
}

// Slab tests of the four children of a quantised node given its three bound chunks: entry distances
// t[k] (>= 0) and hit flags against [0, best_t].  The decode lo = origin + q * scale is folded into the
// slab: t = q * (scale * inv) + (origin * inv - o * inv).  On the device a bound byte reaches the fma as a float16
// SUBNORMAL (two bytes of a bound word spread into the halves of a register by one v_perm_b32: byte * 2^-24 exactly) and
// v_fma_mix_f32 multiplies it by the node's stored scale * 2^24 times inv: the same product, sum and single rounding as
// the host form fmaf((float)byte, scale * inv, b) -- 12 permutes + 24 mixed fmas per node instead of 24 converts + 24 fmas.
#if defined(__HIP_DEVICE_COMPILE__)
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
struct QPair { half2_t lo, hi; };            // children 0, 1 and 2, 3 of one bound word as float16 subnormals
__device__ __forceinline__ QPair q_spread(uint32_t word) {
    QPair r;
    r.lo = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(0u, word, 0x0c010c00u));
    r.hi = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(0u, word, 0x0c030c02u));
    return r;
}
__device__ __forceinline__ float q_half(const QPair& p, int k) { return (float)(k == 0 ? p.lo.x : k == 1 ? p.lo.y : k == 2 ? p.hi.x : p.hi.y); }
#endif
DRT_HD void slab_node4q(F4 c0, F4 c1, F4 c2, f3 inv, f3 oi, bool px, bool py, bool pz, float best_t, float (&t)[4], bool (&h)[4]) {
    const float ax = c0.w * inv.x, ay = c1.x * inv.y, az = c1.y * inv.z;      // scale * 2^24 * inv
    const float bx = fmaf(c0.x, inv.x, oi.x), by = fmaf(c0.y, inv.y, oi.y), bz = fmaf(c0.z, inv.z, oi.z);
    const uint32_t qlox = f32_bits(c1.z), qloy = f32_bits(c1.w), qloz = f32_bits(c2.x);
    const uint32_t qhix = f32_bits(c2.y), qhiy = f32_bits(c2.z), qhiz = f32_bits(c2.w);
    // the near and far planes of every child follow from the sign of the direction (scales are positive, so the planes run with
    // inv): six selects per node on per-ray flags instead of a min/max pair per child and axis
    const uint32_t nx = px ? qlox : qhix, fx = px ? qhix : qlox;
    const uint32_t ny = py ? qloy : qhiy, fy = py ? qhiy : qloy;
    const uint32_t nz = pz ? qloz : qhiz, fz = pz ? qhiz : qloz;
#if defined(__HIP_DEVICE_COMPILE__)
    const QPair pnx = q_spread(nx), pny = q_spread(ny), pnz = q_spread(nz), pfx = q_spread(fx), pfy = q_spread(fy), pfz = q_spread(fz);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float tn = fmaxf(fmaxf(fmaf(q_half(pnx, k), ax, bx), fmaf(q_half(pny, k), ay, by)), fmaxf(fmaf(q_half(pnz, k), az, bz), 0.0f));
        const float tf = fminf(fminf(fmaf(q_half(pfx, k), ax, bx), fmaf(q_half(pfy, k), ay, by)), fminf(fmaf(q_half(pfz, k), az, bz), best_t));
        t[k] = tn;
        h[k] = tn <= tf;
    }
#else
    const float hx = ax * kQScaleDown, hy = ay * kQScaleDown, hz = az * kQScaleDown;     // exact: = scale * inv as the device multiplies it
    for (int k = 0; k < 4; ++k) {
        const float tn = fmaxf(fmaxf(fmaf(q_byte(nx, k), hx, bx), fmaf(q_byte(ny, k), hy, by)), fmaxf(fmaf(q_byte(nz, k), hz, bz), 0.0f));
        const float tf = fminf(fminf(fmaf(q_byte(fx, k), hx, bx), fmaf(q_byte(fy, k), hy, by)), fminf(fmaf(q_byte(fz, k), hz, bz), best_t));
        t[k] = tn;
        h[k] = tn <= tf;
    }
#endif
}

// Visit the inner node s.cur (>= 0).  Returns true when the ray is finished.  ANY: an occlusion query visits the same set
// of nodes in any order when it misses (nineteen exit rays out of twenty) and needs no face id when it hits, so its children
// are not sorted: the first hit child is next, the others are pushed as they come -- a third fewer instructions per visit.
template <class STACK> DRT_HD void trav_check_visit_start(const STACK&) {}
DRT_HD void trav_check_visit_start(const FastStack& st) { DRT_DEV_ASSERT(st.overflow || st.top <= st.base + st.depth * st.stride, kCheckVisitStart); (void)st; }

template <bool ANY = false, class STACK>
DRT_HD bool trav_inner(const Node4Q* __restrict__ nodes, TravState& s, STACK& st) {
    trav_check_visit_start(st);          // (a no-op unless -DDRT_CHECK: a visit must start with a legal stack for its <= 4 unchecked stores)
    const F4* np = reinterpret_cast<const F4*>(nodes + s.cur);
    const F4 q0 = np[0], q1 = np[1], q2 = np[2];
    struct alignas(16) I4 { int32_t x, y, z, w; };
    const I4 ch = *reinterpret_cast<const I4*>(nodes[s.cur].child);        // one 16-byte load of the child references
    int32_t c0 = ch.x, c1 = ch.y, c2 = ch.z, c3 = ch.w;
    float t[4];
    bool h[4];
    slab_node4q(q0, q1, q2, s.inv, s.oi, s.px, s.py, s.pz, s.best_t, t, h);
    // Empty slots need no test here: their quantised interval is inverted on all three axes (quantize_axis), so tn > tf whenever one of
    // scale * inv is non-zero -- always, short of an underflow on all three axes for directions of absurd length, and then the reference that
    // gets pushed, kEmptyChild, is a "leaf" that trav_leaf skips.
    const bool h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3];
#if defined(__HIP_DEVICE_COMPILE__)
    // keep the load of the child references beside the three bound loads: left alone the compiler sinks it into the "some child was
    // hit" branch, where it becomes a second, dependent memory round trip of the visit
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
#endif
    if (ANY) {
        const bool e1 = h0, e2 = h0 | h1, e3 = e2 | h2;          // "an earlier child was hit"
        if (!(e3 | h3)) return trav_pop(s, st);
        st.push_if(c1, h1 & e1);
        st.push_if(c2, h2 & e2);
        st.push_if(c3, h3 & e3);
        st.after_pushes();
        s.cur = h0 ? c0 : (h1 ? c1 : (h2 ? c2 : c3));
        return false;
    }
    // Only the NEAREST hit child is singled out (ties: the lowest slot); the other hit children are pushed in slot order.  A full
    // sort by entry distance (five compare-exchanges on keys + four slot-to-child selects) saved 0.3 % of the node visits of the
    // benchmark's refracted rays (tools/bvhq) and cost a quarter of the visit's instructions.
    if (h0 | h1 | h2 | h3) {
        // (entry distances are >= 0, so their bit patterns order like unsigned integers: integer minima need no NaN canonicalisation)
        const uint32_t kFar = 0xFFFFFFFFu;     // above every distance; an inline constant (-1) on the GPU
        const uint32_t u0 = h0 ? f32_bits(t[0]) : kFar, u1 = h1 ? f32_bits(t[1]) : kFar, u2 = h2 ? f32_bits(t[2]) : kFar, u3 = h3 ? f32_bits(t[3]) : kFar;
        const uint32_t ua = u0 < u1 ? u0 : u1, ub = u2 < u3 ? u2 : u3, tm = ua < ub ? ua : ub;
        const bool n0 = u0 == tm, n1 = (u1 == tm) & !n0, n2 = (u2 == tm) & !(n0 | n1), n3 = !(n0 | n1 | n2);
        st.push_if(c0, h0 & !n0);
        st.push_if(c1, h1 & !n1);
        st.push_if(c2, h2 & !n2);
        st.push_if(c3, h3 & !n3);
        st.after_pushes();
        s.cur = n0 ? c0 : (n1 ? c1 : (n2 ? c2 : c3));
        return false;
    }
    return trav_pop(s, st);
}

// Test the triangles of the leaf s.cur (< 0).  Returns true when the ray is finished.
// DEFER (the persistent kernel, which has no registers to keep a triangle alive through its own test): candidates are taken on the
// barycentric conditions alone and the record of the best one is remembered; the hit-point condition of drt_tri.h is applied ONCE, to the
// winner, when the ray is done (trav_winner_ok) -- and a ray whose winner fails it is traversed again by the exact form.  That is the
// same result: until the end the bound best_t is never below the final winner's t, so every triangle with an ACCEPTABLE hit at or in front
// of the winner had its box visited and its hit taken; a winner that passes is therefore the closest acceptable hit (lowest id among
// equals), and one that does not is caught.  (A ray without any candidate has pruned nothing by distance: a miss is a miss.)
template <bool ANY, bool DEFER = false, class STACK>
DRT_HD bool trav_leaf(const TriRec* __restrict__ tris, TravState& s, STACK& st) {
    if (s.cur == kEmptyChild) return trav_pop(s, st);      // see trav_inner
    const int32_t ref = ~s.cur;
    const int first = ref >> kLeafBits, count = (ref & (kLeafMax - 1)) + 1;
    for (int j = 0; j < count; ++j) {
        const F4* tp = reinterpret_cast<const F4*>(tris + first + j);
        const F4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
        float t;
        const bool hit = DEFER ? tri_hit_mt(s.o, s.d, f3{p0.x, p0.y, p0.z}, f3{p1.x, p1.y, p1.z}, f3{p2.x, p2.y, p2.z}, t)
                               : tri_hit(s.o, s.d, f3{p0.x, p0.y, p0.z}, f3{p1.x, p1.y, p1.z}, f3{p2.x, p2.y, p2.z}, p1.w, t);
        if (hit) {
            int32_t face;
            memcpy(&face, &p0.w, 4);
            if (ANY || t < s.best_t || (t == s.best_t && face < s.best_face)) {
                s.best_t = t; s.best_face = face;
                if (DEFER) s.best_slot = first + j;
                if (ANY) return true;
            }
        }
    }
    return trav_pop(s, st);
}

// The triangle tests of leaf `ref` alone (no pop): what trav_leaf does before it takes the next node.  For the persistent kernel's
// PARKED leaves (drt_trace_kernel.h): a lane that reaches a leaf parks it and goes on with the next node of its stack; the wave tests
// everybody's parked leaves together.  Returns true when an any-hit query is finished by it.
template <bool ANY, bool DEFER = false>
DRT_HD bool trav_leaf_test(const TriRec* __restrict__ tris, TravState& s, int32_t leaf) {
    if (leaf == kEmptyChild) return false;
    const int32_t ref = ~leaf;
    const int first = ref >> kLeafBits, count = (ref & (kLeafMax - 1)) + 1;
    for (int j = 0; j < count; ++j) {
        const F4* tp = reinterpret_cast<const F4*>(tris + first + j);
        const F4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
        float t;
        const bool hit = DEFER ? tri_hit_mt(s.o, s.d, f3{p0.x, p0.y, p0.z}, f3{p1.x, p1.y, p1.z}, f3{p2.x, p2.y, p2.z}, t)
                               : tri_hit(s.o, s.d, f3{p0.x, p0.y, p0.z}, f3{p1.x, p1.y, p1.z}, f3{p2.x, p2.y, p2.z}, p1.w, t);
        if (hit) {
            int32_t face;
            memcpy(&face, &p0.w, 4);
            if (ANY || t < s.best_t || (t == s.best_t && face < s.best_face)) {
                s.best_t = t; s.best_face = face;
                if (DEFER) s.best_slot = first + j;
                if (ANY) return true;
            }
        }
    }
    return false;
}

// A seed: the record in slot `ts` tested as a leaf would test it under DEFER (drt_trace_kernel.h TraceSeed) before the first visit.
DRT_HD void trav_seed(const TriRec* __restrict__ tris, f3 o, f3 d, int32_t ts, float& best_t, int32_t& best_face, int32_t& best_slot) {
    const F4* tp = reinterpret_cast<const F4*>(tris + ts);
    const F4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
    float t;
    if (tri_hit_mt(o, d, f3{p0.x, p0.y, p0.z}, f3{p1.x, p1.y, p1.z}, f3{p2.x, p2.y, p2.z}, t)) {
        int32_t face;
        memcpy(&face, &p0.w, 4);
        best_t = t; best_face = face; best_slot = ts;
    }
}

// The deferred hit-point condition of a finished ray (trav_leaf<ANY, true>): false = traverse it again with the exact form.
DRT_HD bool trav_winner_ok(const TriRec* __restrict__ tris, const TravState& s) {
    if (s.best_face < 0) return true;
    const F4* tp = reinterpret_cast<const F4*>(tris + s.best_slot);
    const F4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
    return hit_point_in_box(s.o, s.d, s.best_t, f3{p0.x, p0.y, p0.z}, f3{p1.x, p1.y, p1.z}, f3{p2.x, p2.y, p2.z}, p1.w);
}

// One node visit, inner or leaf.  Returns true when the ray is finished (result in best_face /
// best_t; for ANY the first hit found).
template <bool ANY, class STACK>
DRT_HD bool trav_step(const Node4Q* __restrict__ nodes, const TriRec* __restrict__ tris, TravState& s, STACK& st) {
    return s.cur >= 0 ? trav_inner<ANY>(nodes, s, st) : trav_leaf<ANY>(tris, s, st);
}

template <bool ANY>
DRT_HD Hit traverse(const Node4Q* __restrict__ nodes, const TriRec* __restrict__ tris, int n_tris,
                    f3 o, f3 d, Stack& st, uint32_t* visits = nullptr) {
    if (n_tris <= 0) return Hit{-1.0f, -1};
    TravState s;
    trav_init(s, st, o, d);
    uint32_t nvis = 1;
    while (!trav_step<ANY>(nodes, tris, s, st)) ++nvis;
    if (visits) *visits = nvis;
    return Hit{s.best_face < 0 ? -1.0f : s.best_t, s.best_face};
}

}  // namespace drt
