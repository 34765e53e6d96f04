// drt_build.hip -- the on-GPU LBVH build (every update_vert), its diagnostics, and the mesh-update entry points.
#include "drt_device.h"
#include "drt_sort.h"
#include <algorithm>

void scene_free_mesh(drt_scene* s) {
    (void)hipFree(s->faces); (void)hipFree(s->verts); (void)hipFree(s->nodes); (void)hipFree(s->tris);
    (void)hipFree(s->keys[0]); (void)hipFree(s->keys[1]); (void)hipFree(s->idx[0]); (void)hipFree(s->idx[1]);
    (void)hipFree(s->hist); (void)hipFree(s->parent_inner); (void)hipFree(s->parent_leaf); (void)hipFree(s->flags);
    (void)hipFree(s->wide); (void)hipFree(s->range_lo); (void)hipFree(s->range_hi); (void)hipFree(s->tris_flat); (void)hipFree(s->slot_of_face);
    s->slot_of_face = nullptr; s->wide = nullptr; s->range_lo = s->range_hi = nullptr; s->tris_flat = nullptr;
    s->faces = nullptr; s->verts = nullptr; s->nodes = nullptr; s->tris = nullptr;
    s->keys[0] = s->keys[1] = s->idx[0] = s->idx[1] = nullptr;
    s->hist = nullptr; s->parent_inner = s->parent_leaf = nullptr; s->flags = nullptr;
    s->cap_faces = s->cap_verts = 0;
}

// ------------------------------------------------------------------------------------------
// build kernels
// ------------------------------------------------------------------------------------------
// float <-> unsigned with the same order (atomicMin / atomicMax on floats of either sign)
__device__ __forceinline__ uint32_t f32_ordered(float f) { const uint32_t b = __float_as_uint(f); return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__device__ __forceinline__ float f32_unordered(uint32_t u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }
constexpr uint32_t kOrdPosInf = 0xFF800000u, kOrdNegInf = 0x007FFFFFu;      // f32_ordered(+inf), f32_ordered(-inf)

// float64 -> float32 vertices (Scene.update_verticex, reference DiffRender.py:379) AND, in the same pass, the scene box the Morton keys are
// normalised with: every block reduces its elements and folds them into six accumulators, so that the build needs no single-block
// pass over the vertices (k_bounds: 16 us of latency in front of a ~0.15 ms build).  `acc`: the accumulators, found at their identity
// (+inf / -inf): the build that reads them puts them back (k_hierarchy, two launches behind the last reader k_morton), and the next
// update waits for that build (begin_update) -- the reset is part of the enqueued work, not of host state, so a captured step
// replays it (a host-side double buffer did not: every replay folded its vertices into the same half and the box could only grow).
// `hist_zero`: the digit histograms of the fused sort.
__global__ void __launch_bounds__(256) k_cast_verts(const double* __restrict__ v64, float* __restrict__ v32, int64_t n3, uint32_t* acc,
                                                    uint32_t* __restrict__ hist_zero, int hist_entries) {
    __shared__ float red[6][4];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < hist_entries; i += (int64_t)gridDim.x * blockDim.x) hist_zero[i] = 0u;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    // a thread takes whole vertices (three consecutive values), so that its three running bounds are per axis
    const int64_t n = n3 / 3;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
        for (int a = 0; a < 3; ++a) {
            const float f = (float)v64[3 * v + a];
            v32[3 * v + a] = f;
            lo[a] = fminf(lo[a], f); hi[a] = fmaxf(hi[a], f);
        }
    }
    if (!acc) return;
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off >= 1; off >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off)); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) for (int a = 0; a < 3; ++a) { red[a][wave] = lo[a]; red[3 + a][wave] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        float v = red[a][0];
        for (int w = 1; w < 4; ++w) v = a < 3 ? fminf(v, red[a][w]) : fmaxf(v, red[a][w]);
        if (a < 3) { if (v < INFINITY) atomicMin(&acc[a], f32_ordered(v)); }
        else if (v > -INFINITY) atomicMax(&acc[a], f32_ordered(v));
    }
}

// BuildParams from the scene box (what k_bounds' last thread does)
__device__ __forceinline__ void params_from_box(const float* lo, const float* hi, BuildParams* out) {
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    out->lox = lo[0]; out->loy = lo[1]; out->loz = lo[2];
    out->ix = ex > 0.f ? 1.0f / ex : 0.f;
    out->iy = ey > 0.f ? 1.0f / ey : 0.f;
    out->iz = ez > 0.f ? 1.0f / ez : 0.f;
    out->pad = pad_for_extent(fmaxf(ex, fmaxf(ey, ez)));
    out->reserved = 0;
    out->plan = morton_plan(ex, ey, ez);
}

// One block: scene box over all vertices -> Morton normalisation + leaf padding.
__global__ void __launch_bounds__(1024) k_bounds(const float* __restrict__ verts, int64_t n_verts, BuildParams* out,
                                                 uint32_t* __restrict__ hist_zero, int hist_entries) {
    __shared__ float red[6][16];
    for (int i = threadIdx.x; i < hist_entries; i += blockDim.x) hist_zero[i] = 0u;   // digit histograms of the fused sort (below)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    // four vertices = twelve floats = three 16-byte loads per step (x0 y0 z0 x1 | y1 z1 x2 y2 | z2 x3 y3 z3), all independent:
    // one block walking 25 k vertices with dependent 4-byte loads was 20 us of latency
    struct alignas(16) V4 { float x, y, z, w; };
    const V4* v4 = reinterpret_cast<const V4*>(verts);      // hipMalloc'ed: 256-byte aligned
    const int64_t n_quads = n_verts / 4;
    for (int64_t q = threadIdx.x; q < n_quads; q += blockDim.x) {
        const V4 a = v4[3 * q], b = v4[3 * q + 1], c = v4[3 * q + 2];
        lo[0] = fminf(fminf(lo[0], fminf(a.x, a.w)), fminf(b.z, c.y)); hi[0] = fmaxf(fmaxf(hi[0], fmaxf(a.x, a.w)), fmaxf(b.z, c.y));
        lo[1] = fminf(fminf(lo[1], fminf(a.y, b.x)), fminf(b.w, c.z)); hi[1] = fmaxf(fmaxf(hi[1], fmaxf(a.y, b.x)), fmaxf(b.w, c.z));
        lo[2] = fminf(fminf(lo[2], fminf(a.z, b.y)), fminf(c.x, c.w)); hi[2] = fmaxf(fmaxf(hi[2], fmaxf(a.z, b.y)), fmaxf(c.x, c.w));
    }
    for (int64_t i = 4 * n_quads + threadIdx.x; i < n_verts; i += blockDim.x) {
        for (int a = 0; a < 3; ++a) {
            const float v = verts[3 * i + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off >= 1; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        for (int a = 0; a < 3; ++a) { red[a][wave] = lo[a]; red[3 + a][wave] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        for (int w = 1; w < nw; ++w)
            for (int a = 0; a < 3; ++a) {
                red[a][0] = fminf(red[a][0], red[a][w]);
                red[3 + a][0] = fmaxf(red[3 + a][0], red[3 + a][w]);
            }
        const float lo3[3] = {red[0][0], red[1][0], red[2][0]}, hi3[3] = {red[3][0], red[4][0], red[5][0]};
        params_from_box(lo3, hi3, out);
    }
}

__device__ __forceinline__ f3 ld_vert(const float* __restrict__ verts, int32_t i) {
    return f3{verts[3 * (int64_t)i], verts[3 * (int64_t)i + 1], verts[3 * (int64_t)i + 2]};
}

// `acc` non-null: the scene box comes from k_cast_verts' accumulators -- every block derives the BuildParams from them (one thread,
// ~1 us) and block 0 also stores them for the later build kernels; otherwise `bp` was written by k_bounds.
// `drop`: low key bits cleared.  Meshes small enough for the fused sort keep 24 of the 30 key bits (256^3 cells for <= 262 144 triangles
// on a surface: still almost one triangle per occupied cell; equal keys are ordered by face id either way) -- one radix pass less.
__global__ void __launch_bounds__(256) k_morton(const int32_t* __restrict__ faces, const float* __restrict__ verts, int n,
                         BuildParams* bp, const uint32_t* __restrict__ acc, uint32_t* __restrict__ keys, uint32_t* __restrict__ idx,
                         uint32_t* hist0, int tiles, int drop) {
    __shared__ BuildParams s_bp;
    if (acc) {
        if (threadIdx.x == 0) {
            float lo3[3], hi3[3];
            for (int a = 0; a < 3; ++a) { lo3[a] = f32_unordered(acc[a]); hi3[a] = f32_unordered(acc[3 + a]); }
            params_from_box(lo3, hi3, &s_bp);
            if (blockIdx.x == 0) *bp = s_bp;
        }
        __syncthreads();
    }
    const BuildParams* q = acc ? &s_bp : bp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 a = ld_vert(verts, faces[3 * i]), b = ld_vert(verts, faces[3 * i + 1]), c = ld_vert(verts, faces[3 * i + 2]);
    const uint32_t key = morton_key(a, b, c, f3{q->lox, q->loy, q->loz}, f3{q->ix, q->iy, q->iz}, q->plan) & ~((1u << drop) - 1u);
    keys[i] = key;
    idx[i] = (uint32_t)i;
    if (hist0) atomicAdd(&hist0[((key >> drop) & (kRadix - 1)) * tiles + i / kSortTile], 1u);   // first pass of the fused sort
}

// Fused pass for small meshes (tiles <= kSortFusedTiles): the per-tile digit offsets are derived inside the scatter
// from the [kRadix x tiles] histogram (every block redundantly reduces it: a few thousand words), and the scatter
// counts the NEXT pass's histogram while it places the keys (the destination tile is known then).  One launch per
// pass instead of three: a 50 k-triangle sort is bound by launch count, not by work.
constexpr int kSortFusedTiles = 128;
__global__ void __launch_bounds__(kSortBlock) k_sort_pass_fused(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ idx_in,
                                                                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ idx_out, int n, int shift,
                                                                const uint32_t* __restrict__ hist, uint32_t* hist_next, int tiles) {
    constexpr int kWaves = kSortBlock / 64;
    __shared__ uint32_t running[kRadix];
    __shared__ uint32_t wcount[kWaves][kRadix];
    __shared__ uint32_t scan[kRadix];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // offsets of this tile: (keys with a smaller digit) + (keys with this digit in earlier tiles); thread = digit
    uint32_t total = 0, before = 0;
    for (int t = 0; t < tiles; ++t) {
        const uint32_t v = hist[tid * tiles + t];
        before += t < (int)blockIdx.x ? v : 0u;
        total += v;
    }
    scan[tid] = total;
    for (int w = 0; w < kWaves; ++w) wcount[w][tid] = 0;
    __syncthreads();
    for (int off = 1; off < kRadix; off <<= 1) {
        const uint32_t v = tid >= off ? scan[tid - off] : 0u;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    running[tid] = scan[tid] - total + before;
    __syncthreads();
    const int base = blockIdx.x * kSortTile;
    for (int r = 0; r < kSortItems; ++r) {
        const int i = base + r * kSortBlock + tid;
        const bool valid = i < n;
        const uint32_t key = valid ? keys_in[i] : 0u;
        const uint32_t val = valid ? idx_in[i] : 0u;
        const uint32_t digit = (key >> shift) & (kRadix - 1);
        unsigned long long peers = __ballot(valid);
        for (int bit = 0; bit < 8; ++bit) {
            const bool set = (digit >> bit) & 1u;
            const unsigned long long bm = __ballot(valid && set);
            peers &= set ? bm : ~bm;
        }
        const uint32_t rank = __popcll(peers & ((1ull << lane) - 1ull));
        if (valid && rank == 0) wcount[wave][digit] = __popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t pos = running[digit] + rank;
            for (int w = 0; w < wave; ++w) pos += wcount[w][digit];
            keys_out[pos] = key;
            idx_out[pos] = val;
            if (hist_next) atomicAdd(&hist_next[((key >> (shift + 8)) & (kRadix - 1)) * tiles + pos / kSortTile], 1u);
        }
        __syncthreads();
        uint32_t add = 0;
        for (int w = 0; w < kWaves; ++w) { add += wcount[w][tid]; wcount[w][tid] = 0; }
        running[tid] += add;
        __syncthreads();
    }
}

// ---- hierarchy ---------------------------------------------------------------------------
__global__ void k_hierarchy(const uint32_t* __restrict__ keys, int n, Node* __restrict__ nodes,
                            int32_t* __restrict__ parent_inner, int32_t* __restrict__ parent_leaf,
                            uint32_t* __restrict__ flags, int32_t* __restrict__ range_lo, int32_t* __restrict__ range_hi, uint32_t* acc_reset) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) parent_inner[0] = -1;
    if (acc_reset && i < 6) acc_reset[i] = i < 3 ? kOrdPosInf : kOrdNegInf;      // k_cast_verts' scene-box accumulators: read by k_tri_flat and k_morton, both done
    if (n == 1) {   // degenerate: one triangle under a root whose second child is an empty box
        if (i == 0) {
            Node nd;
            node_set_child_box(nd, 0, box_empty());
            node_set_child_box(nd, 1, box_empty());
            nd.child0 = ~0; nd.child1 = ~0; nd.pad0 = nd.pad1 = 0;
            nodes[0] = nd;
            parent_leaf[0] = 0;
            range_lo[0] = 0; range_hi[0] = 0;
            flags[0] = 1;   // the single leaf is the "second" arrival: it stops at the root
        }
        return;
    }
    if (i >= n - 1) return;
    int32_t l, r, lo, hi;
    lbvh_children(keys, n, i, l, r, lo, hi);
    range_lo[i] = lo; range_hi[i] = hi;
    nodes[i].child0 = l; nodes[i].child1 = r; nodes[i].pad0 = 0; nodes[i].pad1 = 0;
    if (l >= 0) parent_inner[l] = i * 2 + 0; else parent_leaf[~l] = i * 2 + 0;
    if (r >= 0) parent_inner[r] = i * 2 + 1; else parent_leaf[~r] = i * 2 + 1;
    flags[i] = 0;
}

// A child box occupies three aligned 8-byte granules of its parent node: (lo.x,hi.x) (lo.y,hi.y) (lo.z,hi.z).
__device__ __forceinline__ unsigned long long* box_granule(Node* nodes, int parent, int slot, int axis) {
    float* f = reinterpret_cast<float*>(nodes + parent);
    const int off = axis == 2 ? 8 + 2 * slot : 4 * slot + 2 * axis;
    return reinterpret_cast<unsigned long long*>(f + off);
}
__device__ __forceinline__ unsigned long long pack2(float a, float b) {
    return (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
}

// One thread per leaf: write the triangle record, then carry boxes towards the root.  The second
// thread to arrive at a node owns it.  Per-XCD L2s and per-CU L1s are not coherent on MI355X, so the
// box hand-off between workgroups goes through agent-scope accesses on BOTH sides: the producer
// writes its three 8-byte granules with relaxed agent-scope atomic stores (write-through, sc1),
// drains them (s_waitcnt vmcnt(0)) and only then bumps the node's counter; the second arriver reads
// the sibling's granules with relaxed agent-scope atomic loads (bypass L1).  No fences: a fence per
// tree level (L2 write-back + L1 invalidate, ~3.5 us) made this kernel 235 us; this form is ~5x
// shorter.  drt_bvh_check verifies every box after the fact (tests run it under load).
__global__ void k_refit(const uint32_t* __restrict__ sorted_idx, const int32_t* __restrict__ faces,
                        const float* __restrict__ verts, int n, BuildParams* bp, TriRec* __restrict__ tris,
                        Node* nodes, const int32_t* __restrict__ parent_inner,
                        const int32_t* __restrict__ parent_leaf, uint32_t* flags, int32_t* __restrict__ slot_of_face) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int32_t face = (int32_t)sorted_idx[k];
    slot_of_face[face] = k;         // where a temporal hit seed (drt_trace_kernel.h TraceSeed) finds this face's record
    const f3 a = ld_vert(verts, faces[3 * face]), b = ld_vert(verts, faces[3 * face + 1]), c = ld_vert(verts, faces[3 * face + 2]);
    tris[k] = make_tri(a, b, c, face, hit_margin(bp->pad));
    Box box = box_of_tri(a, b, c, bp->pad);
    int32_t link = parent_leaf[k];
    while (link >= 0) {
        const int p = link >> 1, slot = link & 1;
        __hip_atomic_store(box_granule(nodes, p, slot, 0), pack2(box.lox, box.hix), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(box_granule(nodes, p, slot, 1), pack2(box.loy, box.hiy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(box_granule(nodes, p, slot, 2), pack2(box.loz, box.hiz), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t old = __hip_atomic_fetch_add(&flags[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0) return;
        const unsigned long long gx = __hip_atomic_load(box_granule(nodes, p, slot ^ 1, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long gy = __hip_atomic_load(box_granule(nodes, p, slot ^ 1, 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long gz = __hip_atomic_load(box_granule(nodes, p, slot ^ 1, 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const Box sib{__uint_as_float((unsigned)gx), __uint_as_float((unsigned)gy), __uint_as_float((unsigned)gz),
                      __uint_as_float((unsigned)(gx >> 32)), __uint_as_float((unsigned)(gy >> 32)), __uint_as_float((unsigned)(gz >> 32))};
        box = box_union(box, sib);
        link = parent_inner[p];
    }
}

// Binary -> 4-wide collapse (drt_lbvh.h): one thread per binary node with more than kLeafMax triangles
// (and the root).  The traversal only ever reaches the wide nodes of EVEN-depth binary nodes (a wide node
// adopts grandchildren), but finding a node's depth means walking its parent links to the root -- a chain
// of ~30 dependent loads that made this kernel 34 us; building the (unreferenced) odd-depth ones too is a
// few microseconds of independent work.  Runs after k_refit (kernel boundary = all boxes visible).
__global__ void k_collapse4(const Node* __restrict__ nodes, const int32_t* __restrict__ parent_inner,
                            const int32_t* __restrict__ range_lo, const int32_t* __restrict__ range_hi, int n,
                            Node4Q* __restrict__ wide) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int inner = n > 1 ? n - 1 : 1;
    if (i >= inner) return;
    if (i != 0 && range_hi[i] - range_lo[i] + 1 <= kLeafMax) return;
    Node4 out;
    collapse4(nodes, range_lo, range_hi, n, i, out);
    wide[i] = node4_quantize(out);
}

// Diagnostic for the wide tree (one thread, depth-first from the root -- only the nodes a traversal can reach):
// every leaf marks its triangle slots and checks its box; out[2] = depth of the wide tree.  A traversal keeps at
// most three postponed children per level, so 3 * depth must fit the spilling stack of k_trace_redo / B1 queries.
__global__ void k_wide_walk(const Node4Q* __restrict__ wide, const TriRec* __restrict__ tris, int n, const BuildParams* __restrict__ bp,
                            uint32_t* seen, unsigned long long* out) {
    if (blockIdx.x != 0 || threadIdx.x != 0 || n <= 0) return;
    constexpr int kCap = 512;
    int32_t node[kCap];
    int16_t level[kCap];
    int sp = 0;
    unsigned long long bad = 0, deepest = 0;
    node[sp] = 0; level[sp] = 1; ++sp;
    while (sp > 0) {
        --sp;
        const Node4Q nd = wide[node[sp]];
        const int lv = level[sp];
        if ((unsigned long long)lv > deepest) deepest = lv;
        for (int k = 0; k < 4; ++k) {
            const int32_t c = nd.child[k];
            if (c == kEmptyChild) continue;
            if (c >= 0) {
                if (sp < kCap) { node[sp] = c; level[sp] = (int16_t)(lv + 1); ++sp; } else ++bad;
                continue;
            }
            const int first = (~c) >> kLeafBits, count = ((~c) & (kLeafMax - 1)) + 1;
            for (int j = first; j < first + count; ++j) {
                if (j < 0 || j >= n) { ++bad; continue; }
                seen[j] += 1u;
                const TriRec t = tris[j];
                const f3 a{t.v0x, t.v0y, t.v0z}, b{t.v0x + t.e1x, t.v0y + t.e1y, t.v0z + t.e1z}, cc{t.v0x + t.e2x, t.v0y + t.e2y, t.v0z + t.e2z};
                if (!box_contains(node4q_box(nd, k), box_of_tri(a, b, cc, 0.5f * bp->pad))) ++bad;
            }
        }
    }
    if (3 * deepest > (unsigned long long)kStackTotal) ++bad;   // would overflow the spilling traversal stack
    out[0] += bad;
    out[2] = deepest;
}
__global__ void k_seen_check(const uint32_t* __restrict__ seen, int n, unsigned long long* violations) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && seen[j] != 1u) atomicAdd(violations, 1ull);
}

// Diagnostic: every ancestor's child box must enclose the padded box of leaf k.
__global__ void k_bvh_check(const TriRec* __restrict__ tris, int n, const BuildParams* __restrict__ bp,
                            const Node* __restrict__ nodes, const int32_t* __restrict__ parent_inner,
                            const int32_t* __restrict__ parent_leaf, unsigned long long* violations) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const TriRec t = tris[k];
    const f3 a{t.v0x, t.v0y, t.v0z};
    // e1/e2 were rounded from b - a; rebuild the leaf box conservatively from a, a+e1, a+e2
    const f3 b{t.v0x + t.e1x, t.v0y + t.e1y, t.v0z + t.e1z}, c{t.v0x + t.e2x, t.v0y + t.e2y, t.v0z + t.e2z};
    const Box leaf = box_of_tri(a, b, c, 0.5f * bp->pad);
    int32_t link = parent_leaf[k];
    int32_t child_expect = ~k;
    unsigned long long bad = 0, depth = 0;
    while (link >= 0) {
        ++depth;
        const int p = link >> 1, slot = link & 1;
        const Node nd = nodes[p];
        if ((slot == 0 ? nd.child0 : nd.child1) != child_expect) ++bad;
        if (!box_contains(node_child_box(nd, slot), leaf)) ++bad;
        child_expect = p;
        link = parent_inner[p];
    }
    if (child_expect != 0 && n > 1) ++bad;   // must end at the root
    if (bad) atomicAdd(violations, bad);
    atomicMax(violations + 1, depth);   // tree height = deepest leaf
}


// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
int ensure_capacity(drt_scene* s, int64_t n_faces, int64_t n_verts) {
    if (n_faces <= s->cap_faces && n_verts <= s->cap_verts) return DRT_OK;
    scene_free_mesh(s);
    const int64_t F = n_faces > 0 ? n_faces : 1, V = n_verts > 0 ? n_verts : 1;
    const int64_t tiles = (F + kSortTile - 1) / kSortTile;
    HIP_TRY(hipMalloc(&s->faces, sizeof(int32_t) * 3 * F));
    HIP_TRY(hipMalloc(&s->verts, sizeof(float) * 3 * V));
    HIP_TRY(hipMalloc(&s->nodes, sizeof(Node) * F));
    HIP_TRY(hipMalloc(&s->wide, sizeof(Node4Q) * F));
    HIP_TRY(hipMalloc(&s->range_lo, sizeof(int32_t) * F));
    HIP_TRY(hipMalloc(&s->range_hi, sizeof(int32_t) * F));
    HIP_TRY(hipMalloc(&s->tris, sizeof(TriRec) * F));
    HIP_TRY(hipMalloc(&s->tris_flat, sizeof(TriRec) * F));
    HIP_TRY(hipMalloc(&s->slot_of_face, sizeof(int32_t) * F));
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(hipMalloc(&s->keys[k], sizeof(uint32_t) * F));
        HIP_TRY(hipMalloc(&s->idx[k], sizeof(uint32_t) * F));
    }
    HIP_TRY(hipMalloc(&s->hist, sizeof(uint32_t) * kRadix * tiles * 4));   // one table per radix pass (fused sort)
    HIP_TRY(hipMalloc(&s->parent_inner, sizeof(int32_t) * F));
    HIP_TRY(hipMalloc(&s->parent_leaf, sizeof(int32_t) * F));
    HIP_TRY(hipMalloc(&s->flags, sizeof(uint32_t) * F));
    s->cap_faces = F;
    s->cap_verts = V;
    return DRT_OK;
}

// `acc`: scene-box accumulators filled by k_cast_verts on the way in (which also zeroed the sort histograms); null: k_bounds does both.
static int rebuild_impl(drt_scene* s, hipStream_t st, const uint32_t* acc) {
    const int n = (int)s->n_faces;
    s->built = true;
    s->order_valid = false;
    if (n == 0) return DRT_OK;
    const int tiles = (n + kSortTile - 1) / kSortTile;
    const bool fused_sort = tiles <= kSortFusedTiles;
    const int table = kRadix * tiles;
    const int drop = fused_sort ? 6 : 0;               // 24-bit keys, three radix passes (see k_morton)
    const int last_pass = fused_sort ? 2 : 3;
    // (without `acc` the scene box -- s->params -- and the zeroed histograms come from k_bounds, launched by rebuild() in front of the fork)
    k_morton<<<(n + 255) / 256, 256, 0, st>>>(s->faces, s->verts, n, s->params, acc, s->keys[0], s->idx[0], fused_sort ? s->hist : nullptr, tiles, drop);
    int cur = 0;
    for (int shift = drop, pass = 0; shift < 30; shift += 8, ++pass) {
        if (fused_sort) {
            k_sort_pass_fused<<<tiles, kSortBlock, 0, st>>>(s->keys[cur], s->idx[cur], s->keys[cur ^ 1], s->idx[cur ^ 1], n, shift,
                                                            s->hist + pass * table, pass < last_pass ? s->hist + (pass + 1) * table : nullptr, tiles);
        } else {
            k_sort_hist<<<tiles, kSortBlock, 0, st>>>(s->keys[cur], n, shift, s->hist, tiles);
            k_sort_scan<<<1, 1024, 0, st>>>(s->hist, kRadix * tiles);
            k_sort_scatter<<<tiles, kSortBlock, 0, st>>>(s->keys[cur], s->idx[cur], s->keys[cur ^ 1], s->idx[cur ^ 1], n, shift, s->hist, tiles);
        }
        cur ^= 1;
    }
    // (four passes end in buffer 0, three in buffer 1)
    const int inner = n > 1 ? n - 1 : 1;
    k_hierarchy<<<(inner + 255) / 256, 256, 0, st>>>(s->keys[cur], n, s->nodes, s->parent_inner, s->parent_leaf, s->flags, s->range_lo, s->range_hi, const_cast<uint32_t*>(acc));
    k_refit<<<(n + 255) / 256, 256, 0, st>>>(s->idx[cur], s->faces, s->verts, n, s->params, s->tris, s->nodes,
                                             s->parent_inner, s->parent_leaf, s->flags, s->slot_of_face);
    k_collapse4<<<(inner + 255) / 256, 256, 0, st>>>(s->nodes, s->parent_inner, s->range_lo, s->range_hi, n, s->wide);
    HIP_TRY(hipGetLastError());
    s->order_valid = true;           // idx[cur] = the face ids in Morton order, kept until the next build starts
    s->sorted_buf = cur;
    return DRT_OK;
}

// ---- updates that keep the topology (drt_scene::tree_mode 1, 2) --------------------------------------------------------------------
// What k_morton (BuildParams from the cast kernel's accumulators) and k_hierarchy (node counters to zero, accumulators back to their
// identity) do for a full build, without keys, sort or hierarchy: the tree's links, leaf order and slot ranges stay those of the last
// full build (or of the host's binned-SAH build), k_refit then carries the new boxes up and k_collapse4 re-quantises the wide nodes.
__global__ void k_refit_prep(uint32_t* __restrict__ flags, int n, BuildParams* bp, uint32_t* acc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int inner = n > 1 ? n - 1 : 1;
    if (i < inner) flags[i] = n == 1 ? 1u : 0u;       // (one triangle: the single leaf is the "second" arrival, as in k_hierarchy)
    if (i == 0 && acc) {
        float lo3[3], hi3[3];
        for (int a = 0; a < 3; ++a) { lo3[a] = f32_unordered(acc[a]); hi3[a] = f32_unordered(acc[3 + a]); }
        BuildParams p;
        params_from_box(lo3, hi3, &p);
        *bp = p;
        for (int a = 0; a < 6; ++a) acc[a] = a < 3 ? kOrdPosInf : kOrdNegInf;
    }
}

static int refit_impl(drt_scene* s, hipStream_t st, const uint32_t* acc) {
    const int n = (int)s->n_faces;
    s->built = true;
    if (n == 0) return DRT_OK;
    const int inner = n > 1 ? n - 1 : 1;
    k_refit_prep<<<(inner + 255) / 256, 256, 0, st>>>(s->flags, n, s->params, const_cast<uint32_t*>(acc));
    k_refit<<<(n + 255) / 256, 256, 0, st>>>(s->idx[s->sorted_buf], s->faces, s->verts, n, s->params, s->tris, s->nodes,
                                             s->parent_inner, s->parent_leaf, s->flags, s->slot_of_face);
    k_collapse4<<<(inner + 255) / 256, 256, 0, st>>>(s->nodes, s->parent_inner, s->range_lo, s->range_hi, n, s->wide);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

// Binned-SAH topology on the HOST (tree_mode 2; once per update_mesh): top-down, 16 bins per axis, cost = area(left) * count(left) +
// area(right) * count(right) over the centroid bins of the three axes, down to single triangles (equal centroids: split in the middle).
// Fills what k_hierarchy fills for the LBVH: child links (>= 0 inner, < 0 leaf ~slot), parents, the slot range of every node and the
// order of the faces along the leaves (the role of the Morton order).  Offline this order of tree needs 11 % fewer node visits per
// refracted ray than the LBVH of the final keys (tools/bvhq).
struct HostBox { float lo[3], hi[3]; };
static inline void hb_reset(HostBox& b) { for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; } }
static inline void hb_grow(HostBox& b, const HostBox& o) { for (int a = 0; a < 3; ++a) { b.lo[a] = fminf(b.lo[a], o.lo[a]); b.hi[a] = fmaxf(b.hi[a], o.hi[a]); } }
static inline float hb_area(const HostBox& b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx >= 0.f ? dx * dy + dy * dz + dz * dx : 0.f;
}
static void build_sah_topology(const float* verts, const int32_t* faces, int n, std::vector<uint32_t>& order, std::vector<Node>& nodes,
                               std::vector<int32_t>& parent_inner, std::vector<int32_t>& parent_leaf, std::vector<int32_t>& range_lo,
                               std::vector<int32_t>& range_hi) {
    const int inner = n > 1 ? n - 1 : 1;
    order.resize(n); nodes.assign(inner, Node{}); parent_inner.assign(inner, -1); parent_leaf.assign(n, 0); range_lo.assign(inner, 0); range_hi.assign(inner, 0);
    std::vector<HostBox> box(n);
    std::vector<float> cen(3 * (size_t)n);
    for (int f = 0; f < n; ++f) {
        HostBox b; hb_reset(b);
        for (int k = 0; k < 3; ++k) {
            const float* v = verts + 3 * (size_t)faces[3 * f + k];
            for (int a = 0; a < 3; ++a) { b.lo[a] = fminf(b.lo[a], v[a]); b.hi[a] = fmaxf(b.hi[a], v[a]); }
        }
        box[f] = b;
        for (int a = 0; a < 3; ++a) cen[3 * (size_t)f + a] = 0.5f * (b.lo[a] + b.hi[a]);
        order[f] = (uint32_t)f;
    }
    if (n == 1) {
        Node nd{}; node_set_child_box(nd, 0, box_empty()); node_set_child_box(nd, 1, box_empty());
        nd.child0 = ~0; nd.child1 = ~0; nodes[0] = nd; parent_leaf[0] = 0; return;
    }
    constexpr int kBins = 16;
    struct Task { int node, lo, hi; };
    std::vector<Task> stack;
    stack.push_back(Task{0, 0, n});
    int next_node = 1;
    while (!stack.empty()) {
        const Task t = stack.back(); stack.pop_back();
        const int cnt = t.hi - t.lo;
        range_lo[t.node] = t.lo; range_hi[t.node] = t.hi - 1;
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = t.lo; i < t.hi; ++i)
            for (int a = 0; a < 3; ++a) { const float c = cen[3 * (size_t)order[i] + a]; clo[a] = fminf(clo[a], c); chi[a] = fmaxf(chi[a], c); }
        int best_axis = -1, best_split = 0;
        float best_cost = INFINITY;
        if (cnt > 2) {
            for (int a = 0; a < 3; ++a) {
                const float ext = chi[a] - clo[a];
                if (!(ext > 0.f)) continue;
                const float scale = kBins / ext;
                HostBox bb[kBins]; int bc[kBins];
                for (int k = 0; k < kBins; ++k) { hb_reset(bb[k]); bc[k] = 0; }
                for (int i = t.lo; i < t.hi; ++i) {
                    const uint32_t f = order[i];
                    int k = (int)((cen[3 * (size_t)f + a] - clo[a]) * scale);
                    k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                    hb_grow(bb[k], box[f]); ++bc[k];
                }
                float right_area[kBins]; int right_cnt[kBins];
                HostBox acc; hb_reset(acc); int c = 0;
                for (int k = kBins - 1; k >= 1; --k) { hb_grow(acc, bb[k]); c += bc[k]; right_area[k] = hb_area(acc); right_cnt[k] = c; }
                hb_reset(acc); c = 0;
                for (int k = 0; k < kBins - 1; ++k) {
                    hb_grow(acc, bb[k]); c += bc[k];
                    if (c == 0 || right_cnt[k + 1] == 0) continue;
                    const float cost = hb_area(acc) * (float)c + right_area[k + 1] * (float)right_cnt[k + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k; }
                }
            }
        }
        int mid;
        if (best_axis >= 0) {
            const float scale = kBins / (chi[best_axis] - clo[best_axis]);
            auto left = [&](uint32_t f) {
                int k = (int)((cen[3 * (size_t)f + best_axis] - clo[best_axis]) * scale);
                k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                return k <= best_split;
            };
            mid = (int)(std::partition(order.begin() + t.lo, order.begin() + t.hi, left) - order.begin());
            if (mid == t.lo || mid == t.hi) mid = t.lo + cnt / 2;
        } else {
            mid = t.lo + cnt / 2;       // two triangles, or all centroids in one point
        }
        Node& nd = nodes[t.node];
        nd.pad0 = nd.pad1 = 0;
        for (int side = 0; side < 2; ++side) {
            const int lo = side == 0 ? t.lo : mid, hi = side == 0 ? mid : t.hi;
            int32_t child;
            if (hi - lo == 1) { child = ~lo; parent_leaf[lo] = t.node * 2 + side; }
            else { child = next_node++; parent_inner[child] = t.node * 2 + side; stack.push_back(Task{child, lo, hi}); }
            (side == 0 ? nd.child0 : nd.child1) = child;
        }
    }
}

// mode 2: the mesh (float32 tracer copy, already enqueued on `st`) comes back to the host once per update_mesh, the topology goes up
static int install_sah_topology(drt_scene* s, hipStream_t st) {
    const int n = (int)s->n_faces;
    if (n == 0) return DRT_OK;
    std::vector<float> hv(3 * (size_t)s->n_verts);
    std::vector<int32_t> hf(3 * (size_t)n);
    HIP_TRY(hipMemcpyAsync(hv.data(), s->verts, sizeof(float) * hv.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hf.data(), s->faces, sizeof(int32_t) * hf.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t k = 0; k < hf.size(); ++k) if (hf[k] < 0 || hf[k] >= s->n_verts) return fail(DRT_E_INVALID, "face index out of range");
    std::vector<uint32_t> order; std::vector<Node> nodes; std::vector<int32_t> pi, pl, rl, rh;
    build_sah_topology(hv.data(), hf.data(), n, order, nodes, pi, pl, rl, rh);
    const int inner = n > 1 ? n - 1 : 1;
    HIP_TRY(hipMemcpyAsync(s->idx[0], order.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->nodes, nodes.data(), sizeof(Node) * inner, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->parent_inner, pi.data(), sizeof(int32_t) * inner, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->parent_leaf, pl.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->range_lo, rl.data(), sizeof(int32_t) * inner, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->range_hi, rh.data(), sizeof(int32_t) * inner, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));      // (the host vectors go out of scope)
    s->sorted_buf = 0;
    s->order_valid = true;
    s->topology_fixed = true;
    return DRT_OK;
}

// Triangle records in FACE order: what the projected primary-visibility pass reads (it needs no tree), so that it can run
// while the tree is still being built.
// `order`: the Morton order of the PREVIOUS build over the same faces (null right after a topology change: face order).
// The records carry their face id, so any order gives the same result; a spatially coherent one lets the wave-aggregated
// tile binning of the projection pass (drt_raster.hip) issue a few list atomics per wave instead of one per triangle --
// and the vertices move little between two steps of an optimisation.
// The records carry the margin of the hit-point test (drt_tri.h), a function of the scene box: from k_cast_verts' accumulators (`acc`,
// float64 vertices) or from the BuildParams k_bounds has just written on the same stream (`bp`, float32 vertices) -- the same number
// k_refit puts into the tree's records a few launches later.
__global__ void k_tri_flat(const int32_t* __restrict__ faces, const float* __restrict__ verts, int n, TriRec* __restrict__ tris,
                           const uint32_t* __restrict__ order, const uint32_t* __restrict__ acc, const BuildParams* __restrict__ bp) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    float pad;
    if (acc) {
        const float ex = f32_unordered(acc[3]) - f32_unordered(acc[0]), ey = f32_unordered(acc[4]) - f32_unordered(acc[1]), ez = f32_unordered(acc[5]) - f32_unordered(acc[2]);
        pad = pad_for_extent(fmaxf(ex, fmaxf(ey, ez)));
    } else {
        pad = bp->pad;
    }
    const int f = order ? (int)order[k] : k;
    tris[k] = make_tri(ld_vert(verts, faces[3 * f]), ld_vert(verts, faces[3 * f + 1]), ld_vert(verts, faces[3 * f + 2]), f, hit_margin(pad));
}

// The LBVH build is ten small dependent launches (~0.17 ms at 50 k triangles, launch-latency bound).  It runs on the
// scene's own stream, after the vertices are in place on the caller's stream, and every consumer of the tree waits for
// `build_done` (wait_build): the next render call's output fills and projection pass, which only need the flat triangle
// records written here, overlap it.
int rebuild(drt_scene* s, hipStream_t st, const uint32_t* acc) {
    const int n = (int)s->n_faces;
    // (the previous build's sorted ids are still in idx[sorted_buf]: this launch precedes the fork of the build stream, whose k_morton /
    // sort passes overwrite them)
    if (n > 0) {
        if (!acc) {      // float32 vertices: the scene box by one block, in front of the records that need it (and of the fork: the build reads it too)
            const int tiles = (n + kSortTile - 1) / kSortTile;
            k_bounds<<<1, 1024, 0, st>>>(s->verts, s->n_verts, s->params, s->hist, tiles <= kSortFusedTiles ? 4 * kRadix * tiles : 0);
        }
        k_tri_flat<<<(n + 255) / 256, 256, 0, st>>>(s->faces, s->verts, n, s->tris_flat, s->order_valid ? s->idx[s->sorted_buf] : nullptr, acc, s->params);
    }
    hipStream_t bs = s->async_build ? s->build_stream : st;
    if (bs != st) {
        HIP_TRY(hipEventRecord(s->build_fork, st));
        HIP_TRY(hipStreamWaitEvent(bs, s->build_fork, 0));
    }
    // full build, or -- a kept topology -- boxes and wide nodes only (tree_mode 1: every `rebuild_every`-th update is a full build again)
    const bool keep = s->tree_mode != 0 && s->order_valid && n > 0 && (s->tree_mode == 2 ? s->topology_fixed : (s->built && s->since_full + 1 < s->rebuild_every));
    int rc;
    { StageTimer t(s, bs, kStageBuild);
      rc = keep ? refit_impl(s, bs, acc) : rebuild_impl(s, bs, acc); }
    s->since_full = keep ? s->since_full + 1 : 0;
    if (bs != st) HIP_TRY(hipEventRecord(s->build_done, bs));
    s->build_pending = bs != st;
    return rc;
}

// Before the caller's stream overwrites the mesh buffers of a build that may still be running.
static int begin_update(drt_scene* s, hipStream_t st) {
    if (s->build_pending) HIP_TRY(hipStreamWaitEvent(st, s->build_done, 0));
    return DRT_OK;
}

// Before `st` reads the tree (nodes, sorted triangle records, sorted order).
int wait_build(drt_scene* s, hipStream_t st) {
    if (s->build_pending) HIP_TRY(hipStreamWaitEvent(st, s->build_done, 0));
    return DRT_OK;
}

extern "C" {

int drt_update_mesh(drt_scene_t* s, const int32_t* d_faces, int64_t n_faces, const float* d_verts, int64_t n_verts, void* stream) {
    CHECK_SCENE(s);
    if (n_faces < 0 || n_verts < 0 || (n_faces && !d_faces) || (n_verts && !d_verts)) return fail(DRT_E_INVALID, "bad mesh arguments");
    if (n_faces > (int64_t)1 << 30) return fail(DRT_E_INVALID, "too many faces");
    hipStream_t st = (hipStream_t)stream;
    int rc = begin_update(s, st);
    if (rc) return rc;
    if (n_faces > s->cap_faces || n_verts > s->cap_verts) HIP_TRY(hipStreamSynchronize(s->build_stream));   // buffers are about to be freed
    rc = ensure_capacity(s, n_faces, n_verts);
    if (rc) return rc;
    s->n_faces = n_faces;
    s->n_verts = n_verts;
    s->order_valid = false;          // new topology: the previous build's order is not a permutation of these faces
    if (n_faces) HIP_TRY(hipMemcpyAsync(s->faces, d_faces, sizeof(int32_t) * 3 * n_faces, hipMemcpyDeviceToDevice, st));
    if (n_verts) HIP_TRY(hipMemcpyAsync(s->verts, d_verts, sizeof(float) * 3 * n_verts, hipMemcpyDeviceToDevice, st));
    s->topology_fixed = false;
    if (s->tree_mode == 2) { rc = install_sah_topology(s, st); if (rc) return rc; }
    return rebuild(s, st, nullptr);
}

int drt_update_vert(drt_scene_t* s, const float* d_verts, int64_t n_verts, void* stream) {
    CHECK_BUILT(s);
    if (n_verts != s->n_verts || (n_verts && !d_verts)) return fail(DRT_E_INVALID, "vertex count %lld != %lld", (long long)n_verts, (long long)s->n_verts);
    hipStream_t st = (hipStream_t)stream;
    int rc = begin_update(s, st);
    if (rc) return rc;
    if (n_verts) HIP_TRY(hipMemcpyAsync(s->verts, d_verts, sizeof(float) * 3 * n_verts, hipMemcpyDeviceToDevice, st));
    return rebuild(s, st, nullptr);
}

int drt_update_vert_f64(drt_scene_t* s, const double* d_verts, int64_t n_verts, void* stream) {
    CHECK_BUILT(s);
    if (n_verts != s->n_verts || (n_verts && !d_verts)) return fail(DRT_E_INVALID, "vertex count %lld != %lld", (long long)n_verts, (long long)s->n_verts);
    hipStream_t st = (hipStream_t)stream;
    int rc = begin_update(s, st);
    if (rc) return rc;
    const uint32_t* acc = nullptr;
    if (n_verts) {
        // the cast also gathers the scene box and zeroes the fused sort's histograms (both otherwise k_bounds' job, a single-block kernel)
        const int tiles = (int)((s->n_faces + kSortTile - 1) / kSortTile);
        const bool fused_sort = s->n_faces > 0 && tiles <= kSortFusedTiles;
        acc = s->n_faces > 0 ? s->bounds_acc : nullptr;       // (no faces: no build that would read the box and put the accumulators back)
        k_cast_verts<<<grid_for(n_verts, 256, 256), 256, 0, st>>>(d_verts, s->verts, 3 * n_verts, const_cast<uint32_t*>(acc),
                                                                 s->hist, fused_sort ? 4 * kRadix * tiles : 0);
    }
    return rebuild(s, st, acc);
}

int drt_bvh_check(drt_scene_t* s, void* stream, int64_t* n_violations, int32_t* height, int32_t* wide_depth) {
    CHECK_BUILT(s);
    hipStream_t st = (hipStream_t)stream;
    { int rc = wait_build(s, st); if (rc) return rc; }
    unsigned long long v[3] = {0, 0, 0};
    if (s->n_faces) {
        HIP_TRY(hipMemsetAsync(s->scratch, 0, 3 * sizeof(unsigned long long), st));
        const int n = (int)s->n_faces;
        k_bvh_check<<<(n + 255) / 256, 256, 0, st>>>(s->tris, n, s->params, s->nodes, s->parent_inner, s->parent_leaf, s->scratch);
        // the refit counters are dead after a build: reuse them as per-slot reference counts
        HIP_TRY(hipMemsetAsync(s->flags, 0, sizeof(uint32_t) * n, st));
        k_wide_walk<<<1, 64, 0, st>>>(s->wide, s->tris, n, s->params, s->flags, s->scratch);
        k_seen_check<<<(n + 255) / 256, 256, 0, st>>>(s->flags, n, s->scratch);
        HIP_TRY(hipMemcpyAsync(v, s->scratch, sizeof(v), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (n_violations) *n_violations = (int64_t)v[0];
    if (height) *height = (int32_t)v[1];
    if (wide_depth) *wide_depth = (int32_t)v[2];
    return DRT_OK;
}

int drt_bvh_sorted_faces(drt_scene_t* s, int32_t* d_order, void* stream) {
    CHECK_BUILT(s);
    if (s->n_faces && !d_order) return fail(DRT_E_INVALID, "d_order is null");
    { int rc = wait_build(s, (hipStream_t)stream); if (rc) return rc; }
    if (s->n_faces) HIP_TRY(hipMemcpyAsync(d_order, s->idx[s->sorted_buf], sizeof(int32_t) * s->n_faces, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DRT_OK;
}

int drt_tree_mode(drt_scene_t* s, int mode, int rebuild_every) {
    CHECK_SCENE(s);
    if (mode < 0 || mode > 2 || rebuild_every < 1) return fail(DRT_E_INVALID, "tree mode 0..2, rebuild_every >= 1");
    s->tree_mode = mode; s->rebuild_every = rebuild_every; s->since_full = 0;
    s->topology_fixed = false;         // (mode 2 takes effect at the next drt_update_mesh; until then updates are full builds)
    return DRT_OK;
}

int drt_build_params(drt_scene_t* s, float* out7, void* stream) {
    CHECK_BUILT(s);
    if (!out7) return fail(DRT_E_INVALID, "out7 is null");
    hipStream_t st = (hipStream_t)stream;
    { int rc = wait_build(s, st); if (rc) return rc; }
    BuildParams bp;
    HIP_TRY(hipMemcpyAsync(&bp, s->params, sizeof(bp), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    out7[0] = bp.lox; out7[1] = bp.loy; out7[2] = bp.loz; out7[3] = bp.ix; out7[4] = bp.iy; out7[5] = bp.iz; out7[6] = bp.pad;
    return DRT_OK;
}

}  // extern "C"
