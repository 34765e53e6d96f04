// drt_kernels.hip -- gfx950 kernels and the C ABI of libdrt_hip.so (include/drt_hip.h).
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
// Wave size is 64 throughout.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/drt_hip.h"
#include "drt_common.h"
#include "drt_closest.h"
#include "drt_edge.h"
#include "drt_lbvh.h"
#include "drt_path.h"
#include "drt_shade.h"
#include "drt_traverse.h"
#include "drt_tri.h"

using namespace drt;

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(DRT_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------------
// scene object
// ------------------------------------------------------------------------------------------
constexpr int kTraceBlock = 128;       // threads per block in traversal kernels (2 waves)
#ifndef DRT_STACK_FAST
#define DRT_STACK_FAST 19
#endif
constexpr int kStackFast = DRT_STACK_FAST;         // LDS stack entries per lane (19.5 KB per 256-thread block -> 8 blocks = 32 waves per CU)
constexpr int kStackSlowDev = 45;      // global overflow entries per thread (LBVH height <= 30 + log2 F <= 64)
constexpr int kTraceGridMax = 4096;    // blocks per traversal launch (persistent, grid-stride)
constexpr int64_t kChunkRays = 1 << 26; // max rays per pipeline pass; bounds the list workspace (96 B per ray of the largest pass)

constexpr int kSortBlock = 256, kSortItems = 8, kSortTile = kSortBlock * kSortItems, kRadix = 256;

// stage ids of drt_profile_read
enum { kStageBuild = 0, kStageCull, kStageTrace1, kStageShade1, kStageTrace2, kStageShade2, kStageTrace3, kStageFinish, kStageCollect, kStageBackward, kStageLossBwdFused, kProfStages };

struct BuildParams {   // written by k_bounds, read by the later build kernels
    float lox, loy, loz;
    float ix, iy, iz;   // 1 / extent per axis (0 extent -> 0)
    float pad;
    int32_t reserved;
};

struct drt_scene {
    int device = 0;
    int64_t n_faces = 0, n_verts = 0;
    int64_t cap_faces = 0, cap_verts = 0;
    int32_t* faces = nullptr;      // [F,3] copy
    float* verts = nullptr;        // [V,3] float32 copy (tracer precision)
    Node* nodes = nullptr;         // [max(F-1,1)] binary radix tree (build intermediate)
    Node4Q* wide = nullptr;        // [max(F-1,1)] 4-wide tree (quantised, 64 B/node) read by the traversal, indexed by binary root
    int32_t *range_lo = nullptr, *range_hi = nullptr;   // sorted-slot range of each binary node
    TriRec* tris = nullptr;        // [F] Morton order
    uint32_t *keys[2] = {nullptr, nullptr}, *idx[2] = {nullptr, nullptr};
    uint32_t* hist = nullptr;      // [kRadix * tiles]
    int32_t *parent_inner = nullptr, *parent_leaf = nullptr;
    uint32_t* flags = nullptr;
    BuildParams* params = nullptr;
    int32_t* slow_stack = nullptr; // [kTraceGridMax * kTraceBlock * kStackSlowDev] (B1 queries and edge probes)
    unsigned long long* scratch = nullptr;  // small counters
    // wavefront-pipeline workspace, sized for one chunk of rays, allocated on first use
    // Pipeline workspaces: one per internal stream.  A call is cut into sub-batches that run on
    // different HIP streams, so that the HBM-bound k_cull of one sub-batch overlaps the latency-bound
    // k_trace of another and the tail of one kernel is filled by the next sub-batch's work.
    struct Sub {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        int32_t* q_idx[3] = {nullptr, nullptr, nullptr};     // ray lists R0..R2: index,
        float* q_ray[3] = {nullptr, nullptr, nullptr};       //   float32 ray [cap,6],
        int32_t* q_face[3] = {nullptr, nullptr, nullptr};    //   traversal result
        int32_t *tmp_face1 = nullptr, *tmp_face2 = nullptr;  // fused path keeps face ids here; backward fallback list
        unsigned* qcount = nullptr;                          // [8] list sizes + redo counts of the sub-batch in flight
        int32_t* redo = nullptr;                             // [cap] rays for k_trace_redo
        int32_t* slow_stack = nullptr;                       // traversal-stack overflow area of this stream's kernels
        int64_t q_cap = 0, fused_cap = 0;
    };
    static constexpr int kMaxSub = 4;
    Sub sub[kMaxSub];
    int n_sub = 2;                 // internal streams in use
    int sub_per_stream = 1;        // sub-batches dealt to each stream (when the call is large enough)
    int64_t min_sub_rays = 1 << 24;   // do not cut a call into sub-batches smaller than this
    hipEvent_t fork_ev = nullptr;
    unsigned* vcount = nullptr;    // [0] valid rays of the whole call, [1] silhouette items of drt_vh_loss_fused
    uint32_t* vh_list = nullptr;   // (view, edge) items of drt_vh_loss_fused: its own buffer, so that the call may run on
    int64_t vh_cap = 0;            //   another stream than a pipeline call (which owns the Sub workspaces)
    // optional per-stage timing (drt_profile_*): hipEvent pairs on the launch stream
    bool prof_on = false;
    bool prof_stats = false;                  // level 2: k_trace also accumulates visit statistics (adds contended atomics)
    bool prof_serial = false;                 // level 3: sub-batches run on ONE internal stream, so that each kernel is timed alone
    std::vector<hipEvent_t> prof_ev;          // pool, used pairwise
    std::vector<int> prof_stage;              // stage id of pair k
    size_t prof_used = 0;                     // events handed out since the last read
    unsigned long long* prof_counts = nullptr;  // device [kProfStages]: queue sizes accumulated per stage
    hipStream_t prof_stream = nullptr;
    int n_cu = 256;
    int grid_trace = 2048;         // resident blocks of the pure-traversal kernels
    int grid_path = 2048;          // resident 256-thread blocks of k_trace
    int64_t trace_stats[12] = {0};  // per k_trace stage: wave-steps, lane-steps, refills, max wave-steps (last profile read)
    int refill_min = 16;           // k_trace refills a wave once this many lanes are idle
    int inner_min = 16;            // k_trace leaves the inner phase once fewer lanes than this are at inner nodes
    int64_t chunk_rays = kChunkRays;

    bool built = false;
};

static void scene_free_mesh(drt_scene* s) {
    (void)hipFree(s->faces); (void)hipFree(s->verts); (void)hipFree(s->nodes); (void)hipFree(s->tris);
    (void)hipFree(s->keys[0]); (void)hipFree(s->keys[1]); (void)hipFree(s->idx[0]); (void)hipFree(s->idx[1]);
    (void)hipFree(s->hist); (void)hipFree(s->parent_inner); (void)hipFree(s->parent_leaf); (void)hipFree(s->flags);
    s->faces = nullptr; s->verts = nullptr; s->nodes = nullptr; s->tris = nullptr;
    s->keys[0] = s->keys[1] = s->idx[0] = s->idx[1] = nullptr;
    s->hist = nullptr; s->parent_inner = s->parent_leaf = nullptr; s->flags = nullptr;
    s->cap_faces = s->cap_verts = 0;
}

// ------------------------------------------------------------------------------------------
// build kernels
// ------------------------------------------------------------------------------------------
__global__ void k_cast_verts(const double* __restrict__ v64, float* __restrict__ v32, int64_t n3) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n3; i += (int64_t)gridDim.x * blockDim.x)
        v32[i] = (float)v64[i];
}

// One block: scene box over all vertices -> Morton normalisation + leaf padding.
__global__ void __launch_bounds__(1024) k_bounds(const float* __restrict__ verts, int64_t n_verts, BuildParams* out,
                                                 uint32_t* __restrict__ hist_zero, int hist_entries) {
    __shared__ float red[6][16];
    for (int i = threadIdx.x; i < hist_entries; i += blockDim.x) hist_zero[i] = 0u;   // digit histograms of the fused sort (below)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = threadIdx.x; i < n_verts; i += blockDim.x) {
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
        const float ex = red[3][0] - red[0][0], ey = red[4][0] - red[1][0], ez = red[5][0] - red[2][0];
        out->lox = red[0][0]; out->loy = red[1][0]; out->loz = red[2][0];
        out->ix = ex > 0.f ? 1.0f / ex : 0.f;
        out->iy = ey > 0.f ? 1.0f / ey : 0.f;
        out->iz = ez > 0.f ? 1.0f / ez : 0.f;
        out->pad = pad_for_extent(fmaxf(ex, fmaxf(ey, ez)));
        out->reserved = 0;
    }
}

__device__ __forceinline__ f3 ld_vert(const float* __restrict__ verts, int32_t i) {
    return f3{verts[3 * (int64_t)i], verts[3 * (int64_t)i + 1], verts[3 * (int64_t)i + 2]};
}

__global__ void k_morton(const int32_t* __restrict__ faces, const float* __restrict__ verts, int n,
                         const BuildParams* __restrict__ bp, uint32_t* __restrict__ keys, uint32_t* __restrict__ idx,
                         uint32_t* hist0, int tiles) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 a = ld_vert(verts, faces[3 * i]), b = ld_vert(verts, faces[3 * i + 1]), c = ld_vert(verts, faces[3 * i + 2]);
    const uint32_t key = morton30(a, b, c, f3{bp->lox, bp->loy, bp->loz}, f3{bp->ix, bp->iy, bp->iz});
    keys[i] = key;
    idx[i] = (uint32_t)i;
    if (hist0) atomicAdd(&hist0[(key & (kRadix - 1)) * tiles + i / kSortTile], 1u);   // first pass of the fused sort
}

// ---- LSD radix sort, 8 bits per pass, stable; three launches per pass -------------------
__global__ void __launch_bounds__(kSortBlock) k_sort_hist(const uint32_t* __restrict__ keys, int n, int shift,
                                                          uint32_t* __restrict__ hist, int tiles) {
    __shared__ uint32_t cnt[kRadix];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * kSortTile;
    for (int r = 0; r < kSortItems; ++r) {
        const int i = base + r * kSortBlock + threadIdx.x;
        if (i < n) atomicAdd(&cnt[(keys[i] >> shift) & (kRadix - 1)], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * tiles + blockIdx.x] = cnt[threadIdx.x];
}

// Exclusive scan of hist[0..total) in place, one block of 1024 threads.
__global__ void __launch_bounds__(1024) k_sort_scan(uint32_t* __restrict__ hist, int total) {
    __shared__ uint32_t part[1024];
    const int chunk = (total + 1023) / 1024;
    const int b = threadIdx.x * chunk, e = min(b + chunk, total);
    uint32_t sum = 0;
    for (int i = b; i < e; ++i) sum += hist[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (int i = b; i < e; ++i) {
        const uint32_t h = hist[i];
        hist[i] = run;
        run += h;
    }
}

__global__ void __launch_bounds__(kSortBlock) k_sort_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ idx_in,
                                                             uint32_t* __restrict__ keys_out, uint32_t* __restrict__ idx_out,
                                                             int n, int shift, const uint32_t* __restrict__ hist, int tiles) {
    constexpr int kWaves = kSortBlock / 64;
    __shared__ uint32_t running[kRadix];
    __shared__ uint32_t wcount[kWaves][kRadix];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    running[tid] = hist[tid * tiles + blockIdx.x];
    for (int w = 0; w < kWaves; ++w) wcount[w][tid] = 0;
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
        }
        __syncthreads();
        uint32_t add = 0;
        for (int w = 0; w < kWaves; ++w) { add += wcount[w][tid]; wcount[w][tid] = 0; }
        running[tid] += add;
        __syncthreads();
    }
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
                            uint32_t* __restrict__ flags, int32_t* __restrict__ range_lo, int32_t* __restrict__ range_hi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) parent_inner[0] = -1;
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
                        const int32_t* __restrict__ parent_leaf, uint32_t* flags) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int32_t face = (int32_t)sorted_idx[k];
    const f3 a = ld_vert(verts, faces[3 * face]), b = ld_vert(verts, faces[3 * face + 1]), c = ld_vert(verts, faces[3 * face + 2]);
    tris[k] = make_tri(a, b, c, face);
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
    if (3 * deepest > (unsigned long long)(kStackFast + kStackSlowDev)) ++bad;   // would overflow the spilling traversal stack
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
// traversal kernels
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ Stack make_stack(int32_t (*lds)[kTraceBlock], const TraceCtx& c) {
    Stack st;
    st.fast = &lds[0][threadIdx.x];
    st.stride = kTraceBlock;
    st.depth_fast = kStackFast;
    st.slow = c.slow_stack + ((int64_t)blockIdx.x * kTraceBlock + threadIdx.x) * kStackSlowDev;
    st.sp = 0;
    return st;
}

template <bool ANY>
__global__ void __launch_bounds__(kTraceBlock) k_intersect(TraceCtx c, const float* __restrict__ rays, int64_t n,
                                                            float* __restrict__ T, int32_t* __restrict__ ID,
                                                            uint8_t* __restrict__ hitflag) {
    __shared__ int32_t lds[kStackFast][kTraceBlock];
    Stack st = make_stack(lds, c);
    for (int64_t i = blockIdx.x * (int64_t)kTraceBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kTraceBlock) {
        const f3 o{rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, d{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]};
        const Hit h = traverse<ANY>(c.nodes, c.tris, c.n_tris, o, d, st);
        if (ANY) {
            hitflag[i] = h.face >= 0 ? 1 : 0;
        } else {
            T[i] = h.t;
            ID[i] = h.face;
        }
    }
}

// Brute force over every triangle with the same test: the GPU-side checker of the traversal.
__global__ void __launch_bounds__(256) k_bruteforce(const TriRec* __restrict__ tris, int n_tris, const float* __restrict__ rays,
                                                    int64_t n, float* __restrict__ T, int32_t* __restrict__ ID) {
    __shared__ TriRec tile[256];
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const bool live = i < n;
    f3 o{0, 0, 0}, d{0, 0, 1};
    if (live) { o = f3{rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}; d = f3{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]}; }
    float best = INFINITY;
    int32_t best_face = -1;
    for (int j0 = 0; j0 < n_tris; j0 += 256) {
        __syncthreads();
        if (j0 + (int)threadIdx.x < n_tris) tile[threadIdx.x] = tris[j0 + threadIdx.x];
        __syncthreads();
        const int m = min(256, n_tris - j0);
        for (int j = 0; j < m; ++j) {
            const TriRec t = tile[j];
            float tt;
            if (tri_hit(o, d, f3{t.v0x, t.v0y, t.v0z}, f3{t.e1x, t.e1y, t.e1z}, f3{t.e2x, t.e2y, t.e2z}, tt)) {
                if (tt < best || (tt == best && t.face < best_face)) { best = tt; best_face = t.face; }
            }
        }
    }
    if (live) { T[i] = best_face >= 0 ? best : -1.0f; ID[i] = best_face; }
}

// ---- the two-bounce refraction path as a compacted wavefront pipeline ---------------------
//
// Only ~5-25 % of camera rays hit the object and the three traversals of a path have very
// different lengths.  One thread per ray start-to-end leaves most lanes idle (measured on the
// first version: 25 % VALU lane utilisation with the SIMDs issue-bound), so the path is cut
// into stages that hand COMPACT ray lists to each other:
//   k_cull     all rays : slab test against the wide root's child boxes; definite miss -> write
//                         zeros (pure HBM streaming); candidate -> R0 (index + float32 ray)
//   k_trace    R0       : closest hit -> R0.face                    (persistent, lanes refilled)
//   k_shade1   R0       : miss -> zeros; hit -> float64 bounce #1; refracted -> R1
//   k_trace    R1       : closest hit -> R1.face
//   k_shade2   R1       : miss/TIR -> zeros; else float64 bounces #1+#2 -> provisional outputs, R2
//   k_trace    R2 (any) : occlusion test of the exit ray -> R2.face
//   k_finish   R2       : occluded -> zeros; survivors -> list of valid rays (for the backward)
// k_trace never diverges on "what kind of ray is this": it only walks the BVH, and a lane whose
// ray ends takes the next ray of its wave's segment at once, so waves stay full.  The float64
// shading runs in the k_shade kernels over dense lists with no traversal in them.
// A list push costs ONE returning atomic per 256-thread block iteration (a single counter word
// sustains only ~90 returning atomics per microsecond on MI355X).
constexpr int kPathBlock = 256;
constexpr int kPathWaves = kPathBlock / 64;

struct RayList {
    int32_t* idx;     // ray index within the chunk
    float* ray;       // [cap,6] float32 origin, direction -- exactly what the tracer sees
    int32_t* face;    // [cap] traversal result
};
struct Pipe {
    RayList r0, r1, r2;
    unsigned* count;   // [0..2] list sizes of the sub-batch in flight, [4..6] rays handed to k_trace_redo per stage
    unsigned* valid;   // number of valid rays of the whole call (shared by all sub-batches)
    int32_t* redo;     // list entries whose traversal overflowed the LDS stack
};

// Block-wide ordered compaction: returns the list slot of this thread's item, or -1.
// Must be reached by every thread of the block (contains barriers).
__device__ __forceinline__ int block_push(bool pred, unsigned* counter, unsigned* s_tmp /* [kPathWaves + 1] */) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(pred);
    if (lane == 0) s_tmp[wave] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int w = 0; w < kPathWaves; ++w) { const unsigned c = s_tmp[w]; s_tmp[w] = tot; tot += c; }
        s_tmp[kPathWaves] = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    const int slot = pred ? (int)(s_tmp[kPathWaves] + s_tmp[wave] + __popcll(m & ((1ull << lane) - 1ull))) : -1;
    __syncthreads();
    return slot;
}

// Staged list append.  A returning atomic on one counter word is served at ~90 per microsecond, and the 2048 resident
// blocks of a shading kernel all arrive at it together: one push per 256-entry block iteration made k_shade1/2 and
// k_finish wait on the counter for more than half of their time.  A block therefore collects its survivors in LDS
// (index + float32 ray) and reserves list space once per ~500-700 of them; the copy-out is fully coalesced.
constexpr int kStageCap = 768;                       // 21.5 KB: six blocks per CU keep their LDS
struct StageMem {
    int32_t idx[kStageCap];
    float ray[kStageCap * 6];
    unsigned n, base, wtot[kPathWaves];
};
__device__ __forceinline__ void stage_init(StageMem& m) {
    if (threadIdx.x == 0) m.n = 0u;
    __syncthreads();
}
// whole block; m.n must be stable (a barrier since its last update)
__device__ __forceinline__ void stage_flush(StageMem& m, const RayList& out, unsigned* counter) {
    const unsigned cnt = m.n;
    if (threadIdx.x == 0) m.base = cnt ? atomicAdd(counter, cnt) : 0u;
    __syncthreads();
    const unsigned base = m.base;
    for (unsigned k = threadIdx.x; k < cnt; k += kPathBlock) out.idx[base + k] = m.idx[k];
    if (out.ray) for (unsigned k = threadIdx.x; k < 6u * cnt; k += kPathBlock) out.ray[6 * (int64_t)base + k] = m.ray[k];
    __syncthreads();
    if (threadIdx.x == 0) m.n = 0u;
    __syncthreads();
}
// whole block, once per block iteration (<= kPathBlock new entries)
__device__ __forceinline__ void stage_push(StageMem& m, bool pred, int32_t i, f3 o, f3 d, const RayList& out, unsigned* counter) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long mask = __ballot(pred);
    if (lane == 0) m.wtot[wave] = (unsigned)__popcll(mask);
    __syncthreads();
    unsigned slot = m.n, tot = 0;
    for (int w = 0; w < kPathWaves; ++w) { const unsigned c = m.wtot[w]; if (w < wave) slot += c; tot += c; }
    if (pred) {
        slot += (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
        m.idx[slot] = i;
        if (out.ray) { float* e = m.ray + 6 * slot; e[0] = o.x; e[1] = o.y; e[2] = o.z; e[3] = d.x; e[4] = d.y; e[5] = d.z; }
    }
    __syncthreads();
    if (threadIdx.x == 0) m.n += tot;
    __syncthreads();
    if (m.n > kStageCap - kPathBlock) stage_flush(m, out, counter);
}


__device__ __forceinline__ Stack make_stack256(int32_t (*lds)[kPathBlock], const TraceCtx& c) {
    Stack st;
    st.fast = &lds[0][threadIdx.x];
    st.stride = kPathBlock;
    st.depth_fast = kStackFast;
    st.slow = c.slow_stack + ((int64_t)blockIdx.x * kPathBlock + threadIdx.x) * kStackSlowDev;
    st.sp = 0;
    return st;
}

__device__ __forceinline__ void write_dead(int64_t i, double* out_ori, double* out_dir, uint8_t* mask, int32_t* face2) {
    const d3 z{0.0, 0.0, 0.0};
    store_d3(out_ori, i, z);
    store_d3(out_dir, i, z);
    mask[3 * i] = 0; mask[3 * i + 1] = 0; mask[3 * i + 2] = 0;
    face2[i] = -1;
}

__device__ __forceinline__ void store_ray32(float* ray, int slot, f3 o, f3 d) {
    float* e = ray + 6 * (int64_t)slot;
    e[0] = o.x; e[1] = o.y; e[2] = o.z; e[3] = d.x; e[4] = d.y; e[5] = d.z;
}

// Conservative "can this ray touch the mesh at all": two levels of the wide tree (the root's
// children, then the children of every inner child the ray enters).  k_cull is HBM-bound, so these
// <= 20 slab tests are free, and every ray they reject is one the traversal stages never see.
__device__ __forceinline__ unsigned hit_mask4(const Node4Q* __restrict__ node, f3 inv, f3 oi) {
    const F4* np = reinterpret_cast<const F4*>(node);
    const int32_t* ch = node->child;
    float t[4];
    bool h[4];
    slab_node4q(np[0], np[1], np[2], inv, oi, INFINITY, t, h);
    return (unsigned)(h[0] & (ch[0] != kEmptyChild)) | ((unsigned)(h[1] & (ch[1] != kEmptyChild)) << 1) |
           ((unsigned)(h[2] & (ch[2] != kEmptyChild)) << 2) | ((unsigned)(h[3] & (ch[3] != kEmptyChild)) << 3);
}

__device__ __forceinline__ bool hits_top_boxes(const Node4Q* __restrict__ nodes, f3 o, f3 d) {
    const f3 inv{safe_inv(d.x), safe_inv(d.y), safe_inv(d.z)};
    const f3 oi{-o.x * inv.x, -o.y * inv.y, -o.z * inv.z};
    const unsigned m = hit_mask4(nodes, inv, oi);
    if (m == 0) return false;
    bool any = false;
    for (int k = 0; k < 4; ++k) {
        if (!((m >> k) & 1u)) continue;
        const int32_t c = nodes[0].child[k];
        if (c < 0) { any = true; continue; }            // a leaf directly under the root
        any |= hit_mask4(nodes + c, inv, oi) != 0;
    }
    return any;
}

// `tile_w` > 0: the rays are rows of an image `tile_w` pixels wide (any number of images of a
// multiple-of-4 height, concatenated).  A block then takes a 64x4 pixel patch per iteration and
// appends its candidates in 16x4-tile order, so that the 64 rays a traversal wave picks up come from
// a compact screen region (same BVH nodes, same depth) instead of a 64x1 strip.
template <bool FUSED>
__global__ void __launch_bounds__(kPathBlock, 8) k_cull(const Node4Q* __restrict__ nodes, int n_tris, const double* __restrict__ origin, const double* __restrict__ dir,
                                                      const uint8_t* __restrict__ valid, int64_t n, double* __restrict__ out_ori,
                                                      double* __restrict__ out_dir, uint8_t* __restrict__ mask,
                                                      int32_t* __restrict__ face1, int32_t* __restrict__ face2, Pipe p, int tile_w) {
    __shared__ unsigned s_tmp[kPathWaves + 1];
    __shared__ uint8_t s_flag[kPathBlock];
    __shared__ int s_slot[kPathBlock];
    const int tid = threadIdx.x;
    // (`nodes` is a __restrict__ parameter of its own, not the TraceCtx struct, so that the compiler can prove the root
    // node is never clobbered: it is then fetched once, through the scalar cache, instead of by four vector loads per ray)
    // patch = 64 pixels wide x 4 rows (every wave reads one full 1536-byte row segment); tiles of 16x4 pixels
    const int vt = ((tid & 63) >> 4) * 64 + (tid >> 6) * 16 + (tid & 15);   // position of this thread's pixel in tile order
    const int64_t patches_per_row = tile_w > 0 ? tile_w / 64 : 1;
    for (int64_t base = blockIdx.x * (int64_t)kPathBlock; base < n; base += (int64_t)gridDim.x * kPathBlock) {
        int64_t i = base + tid;
        if (tile_w > 0) {
            const int64_t patch = base / kPathBlock;
            const int64_t y = 4 * (patch / patches_per_row) + (tid >> 6), x = 64 * (patch % patches_per_row) + (tid & 63);
            i = y * tile_w + x;
        }
        bool cand = false;
        f3 o{0.f, 0.f, 0.f}, d{0.f, 0.f, 1.f};
        if (i < n) {
            // the fused loss ignores pixels without a target (reference optim.py:105): their rays are not traced
            if (!FUSED || valid[i]) {
                o = to_f32(load_d3(origin, i)); d = to_f32(load_d3(dir, i));
                cand = n_tris > 0 && hits_top_boxes(nodes, o, d);
            }
            if (!cand) {
                face1[i] = -1;
                if (!FUSED) write_dead(i, out_ori, out_dir, mask, face2);
            }
        }
        // nine out of ten patches are pure background: one barrier (with an OR-reduction) instead of the six of the push
        if (!__syncthreads_or(cand ? 1 : 0)) continue;
        int slot;
        if (tile_w > 0) {
            s_flag[vt] = cand ? 1 : 0;
            __syncthreads();
            s_slot[tid] = block_push(s_flag[tid] != 0, &p.count[0], s_tmp);   // ranks in tile order
            __syncthreads();
            slot = s_slot[vt];
            __syncthreads();
        } else {
            slot = block_push(cand, &p.count[0], s_tmp);
        }
        if (slot >= 0) { p.r0.idx[slot] = (int32_t)i; store_ray32(p.r0.ray, slot, o, d); }
    }
}

// Persistent traversal over a ray list.  Each wave owns a contiguous segment of the list; a lane
// whose ray finishes takes the segment's next ray (no atomics: the cursor is wave-uniform).
template <bool ANY>
__global__ void __launch_bounds__(kPathBlock, 8) k_trace(TraceCtx c, const float* __restrict__ rays, const unsigned* __restrict__ n_ptr,
                                                       int32_t* __restrict__ out_face, int32_t* __restrict__ redo_list, unsigned* redo_count,
                                                       int refill_min, int inner_min, unsigned long long* stats) {
    __shared__ int32_t lds[kStackFast + 1][kPathBlock];     // + the dump slot of FastStack: 20 x 1 KB x 8 blocks = the CU's 160 KB
    FastStack st;
    st.fast = &lds[0][threadIdx.x]; st.stride = kPathBlock; st.depth = kStackFast; st.sp = 0; st.overflow = false;
    const unsigned n = *n_ptr;
    const int lane = threadIdx.x & 63;
    // Work assignment without atomics, XCD-aware: workgroups are dealt to the 8 XCDs round-robin (block b runs on XCD
    // b % 8), and every XCD has its own L2.  The list -- in tile order, so neighbouring entries walk the same part of the
    // tree -- is therefore cut into 8 contiguous parts, one per XCD, and only WITHIN its part are the groups of 64
    // consecutive rays interleaved over that XCD's waves (wave w owns groups w, w + W, w + 2W, ... of the part: coherent
    // within a group, statistically balanced across waves).  `taken` counts the rays this wave has started.
    constexpr unsigned kXcd = 8;
    const unsigned n_groups = (n + 63u) >> 6;
    const bool split = gridDim.x % kXcd == 0 && n_groups >= 64u * kXcd;
    const unsigned xcd = split ? blockIdx.x % kXcd : 0u, parts = split ? kXcd : 1u;
    const unsigned wave = (split ? blockIdx.x / kXcd : blockIdx.x) * kPathWaves + (threadIdx.x >> 6);
    const unsigned n_waves = (split ? gridDim.x / kXcd : gridDim.x) * kPathWaves;
    const unsigned part_lo = (unsigned)((unsigned long long)n_groups * xcd / parts), part_hi = (unsigned)((unsigned long long)n_groups * (xcd + 1) / parts);
    const unsigned part_groups = part_hi - part_lo;
    const unsigned my_groups = wave < part_groups ? (part_groups - wave + n_waves - 1) / n_waves : 0u;
    const unsigned my_rays = my_groups << 6;      // upper bound; indices >= n are skipped
    unsigned taken = 0;
    int32_t slot = -1;
    TravState s;
    unsigned long long wave_steps = 0, lane_steps = 0, refills = 0;   // wave-uniform diagnostics (scalar registers)
    for (;;) {
        const unsigned long long idle = __ballot(slot < 0);
        if (idle != 0 && taken < my_rays && (__popcll(idle) >= refill_min || idle == ~0ull)) {
            if (slot < 0) {
                const unsigned j = taken + (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
                const unsigned k = ((part_lo + (j >> 6) * n_waves + wave) << 6) | (j & 63u);
                if (j < my_rays && k < n) {
                    const float* e = rays + 6 * (int64_t)k;
                    trav_init(s, st, f3{e[0], e[1], e[2]}, f3{e[3], e[4], e[5]});
                    st.overflow = false;
                    slot = (int32_t)k;
                }
            }
            taken += (unsigned)__popcll(idle);
            ++refills;
        }
        const unsigned long long busy = __ballot(slot >= 0);
        if (busy == 0) break;
        // inner phase ("while-while"): lanes at inner nodes keep descending; lanes that reached a leaf
        // wait, so that the (longer) triangle code runs once for many lanes instead of on every step
        for (;;) {
            const bool at_inner = slot >= 0 && s.cur >= 0;
            const unsigned long long mi = __ballot(at_inner);
            if (mi == 0) break;
            if (__popcll(mi) < inner_min && __ballot(slot >= 0 && s.cur < 0) != 0) break;
            ++wave_steps;
            lane_steps += (unsigned long long)__popcll(mi);
            if (at_inner) {
                const bool done = trav_inner(c.nodes, s, st);
                if (st.overflow) {              // LDS stack exhausted (rare): hand the ray to k_trace_redo
                    redo_list[atomicAdd(redo_count, 1u)] = slot;
                    slot = -1;
                } else if (done) {
                    out_face[slot] = s.best_face;
                    slot = -1;
                }
            }
        }
        // leaf phase
        const bool at_leaf = slot >= 0 && s.cur < 0;
        const unsigned long long ml = __ballot(at_leaf);
        if (ml != 0) {
            ++wave_steps;
            lane_steps += (unsigned long long)__popcll(ml);
            if (at_leaf && trav_leaf<ANY>(c.tris, s, st)) {
                out_face[slot] = s.best_face;
                slot = -1;
            }
        }
    }
    if (stats && lane == 0 && wave_steps) {
        atomicAdd(stats + 0, wave_steps);
        atomicAdd(stats + 1, lane_steps);
        atomicAdd(stats + 2, refills);
        atomicMax(stats + 3, wave_steps);
    }
}

// Second pass for the rays whose traversal overflowed the LDS-only stack of k_trace: one thread per
// ray, spilling stack.  Normally the list is empty and the kernel returns at once.
template <bool ANY>
__global__ void __launch_bounds__(kTraceBlock) k_trace_redo(TraceCtx c, const float* __restrict__ rays, const int32_t* __restrict__ redo_list,
                                                             const unsigned* __restrict__ redo_count, int32_t* __restrict__ out_face) {
    __shared__ int32_t lds[kStackFast][kTraceBlock];
    const unsigned n = *redo_count;
    if (n == 0) return;
    Stack st = make_stack(lds, c);
    for (unsigned k = blockIdx.x * kTraceBlock + threadIdx.x; k < n; k += gridDim.x * kTraceBlock) {
        const int32_t slot = redo_list[k];
        const float* e = rays + 6 * (int64_t)slot;
        out_face[slot] = traverse<ANY>(c.nodes, c.tris, c.n_tris, f3{e[0], e[1], e[2]}, f3{e[3], e[4], e[5]}, st).face;
    }
}

// R0 -> R1: primary hit -> float64 bounce #1 -> refracted ray
template <bool FUSED>
__global__ void __launch_bounds__(kPathBlock) k_shade1(PathCtx c, const double* __restrict__ origin, const double* __restrict__ dir,
                                                        double* __restrict__ out_ori, double* __restrict__ out_dir, uint8_t* __restrict__ mask,
                                                        int32_t* __restrict__ face1, int32_t* __restrict__ face2, Pipe p) {
    __shared__ StageMem stage;
    stage_init(stage);
    const unsigned n0 = p.count[0];
    // each block takes one contiguous run of the list, so that its survivors stay in list (= screen tile) order
    const unsigned per_block = ((n0 + gridDim.x - 1) / gridDim.x + kPathBlock - 1) / kPathBlock * kPathBlock;
    const unsigned first = blockIdx.x * per_block, last = min(n0, first + per_block);
    for (unsigned base = first; base < last; base += kPathBlock) {
        const unsigned k = base + threadIdx.x;
        bool ok = false;
        int64_t i = 0;
        f3 o2{0.f, 0.f, 0.f}, d2{0.f, 0.f, 1.f};
        if (k < n0) {
            i = p.r0.idx[k];
            const int32_t f1 = p.r0.face[k];
            face1[i] = f1;
            if (f1 >= 0) {
                d3 v0, v1, v2;
                int32_t vid[3];
                Bounce b;
                load_tri64(c, f1, v0, v1, v2, vid);
                bounce_forward(load_d3(origin, i), load_d3(dir, i), v0, v1, v2, c.ior_ext, c.ior_int, b);
                ok = !b.tir;
                o2 = to_f32(b.new_o); d2 = to_f32(b.wt);
            }
            if (!ok && !FUSED) write_dead(i, out_ori, out_dir, mask, face2);
        }
        stage_push(stage, ok, (int32_t)i, o2, d2, p.r1, &p.count[1]);
    }
    stage_flush(stage, p.r1, &p.count[1]);
}

// R1 -> R2: second hit -> float64 bounces #1 and #2 -> provisional exit ray
template <bool FUSED>
__global__ void __launch_bounds__(kPathBlock) k_shade2(PathCtx c, const double* __restrict__ origin, const double* __restrict__ dir,
                                                        double* __restrict__ out_ori, double* __restrict__ out_dir, uint8_t* __restrict__ mask,
                                                        const int32_t* __restrict__ face1, int32_t* __restrict__ face2, Pipe p) {
    __shared__ StageMem stage;
    stage_init(stage);
    const unsigned n1 = p.count[1];
    const unsigned per_block = ((n1 + gridDim.x - 1) / gridDim.x + kPathBlock - 1) / kPathBlock * kPathBlock;
    const unsigned first = blockIdx.x * per_block, last = min(n1, first + per_block);
    for (unsigned base = first; base < last; base += kPathBlock) {
        const unsigned k = base + threadIdx.x;
        bool ok = false;
        int64_t i = 0;
        f3 o3{0.f, 0.f, 0.f}, d3f{0.f, 0.f, 1.f};
        if (k < n1) {
            i = p.r1.idx[k];
            const int32_t f2 = p.r1.face[k];
            if (f2 >= 0) {
                d3 v0, v1, v2;
                int32_t vid[3];
                Bounce b;
                load_tri64(c, face1[i], v0, v1, v2, vid);
                bounce_forward(load_d3(origin, i), load_d3(dir, i), v0, v1, v2, c.ior_ext, c.ior_int, b);
                const d3 o2 = b.new_o, d2 = b.wt;
                load_tri64(c, f2, v0, v1, v2, vid);
                bounce_forward(o2, d2, v0, v1, v2, c.ior_ext, c.ior_int, b);
                ok = !b.tir;
                if (ok) {
                    o3 = to_f32(b.new_o); d3f = to_f32(b.wt);
                    face2[i] = f2;
                    if (!FUSED) {
                        store_d3(out_ori, i, b.new_o);
                        store_d3(out_dir, i, b.wt);
                        mask[3 * i] = 1; mask[3 * i + 1] = 1; mask[3 * i + 2] = 1;
                    }
                }
            }
            if (!ok) { if (FUSED) face2[i] = -1; else write_dead(i, out_ori, out_dir, mask, face2); }
        }
        stage_push(stage, ok, (int32_t)i, o3, d3f, p.r2, &p.count[2]);
    }
    stage_flush(stage, p.r2, &p.count[2]);
}

// R2: occluded exit rays die; survivors are appended to the caller's list of valid rays (global index).
__global__ void __launch_bounds__(kPathBlock) k_finish(double* __restrict__ out_ori, double* __restrict__ out_dir, uint8_t* __restrict__ mask,
                                                        int32_t* __restrict__ face2, Pipe p, int64_t chunk_base, int32_t* __restrict__ valid_idx) {
    __shared__ StageMem stage;
    stage_init(stage);
    const RayList out{valid_idx, nullptr, nullptr};              // index-only list
    const unsigned n2 = p.count[2];
    const unsigned per_block = ((n2 + gridDim.x - 1) / gridDim.x + kPathBlock - 1) / kPathBlock * kPathBlock;
    const unsigned first = blockIdx.x * per_block, last = min(n2, first + per_block);
    for (unsigned base = first; base < last; base += kPathBlock) {
        const unsigned k = base + threadIdx.x;
        bool keep = false;
        int64_t i = 0;
        if (k < n2) {
            i = p.r2.idx[k];
            keep = p.r2.face[k] < 0;
            if (!keep) write_dead(i, out_ori, out_dir, mask, face2);
        }
        if (valid_idx) stage_push(stage, keep, (int32_t)(chunk_base + i), f3{0.f, 0.f, 0.f}, f3{0.f, 0.f, 0.f}, out, p.valid);
    }
    if (valid_idx) stage_flush(stage, out, p.valid);
}

__global__ void k_store_count(const unsigned* __restrict__ count, int64_t* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = (int64_t)*count;
}

struct AtomicAdd3 {
    double* g;
    __device__ __forceinline__ void operator()(int32_t v, d3 a) const {
        unsafeAtomicAdd(g + 3 * (int64_t)v + 0, a.x);
        unsafeAtomicAdd(g + 3 * (int64_t)v + 1, a.y);
        unsafeAtomicAdd(g + 3 * (int64_t)v + 2, a.z);
    }
};

// Vertex-gradient accumulation through an LDS hash table.  The float64 scatter is bound by the
// chip's atomic rate (measured 22.6 G global_atomic_add_f64 per second, tools/ubench/atomic_scope.hip,
// independent of scope or per-XCD privatisation), and neighbouring rays hit neighbouring triangles
// that share vertices: a block first sums its contributions per vertex in LDS (ds_add_f64 after a
// compare-and-swap probe on the key) and then issues three global atomics per DISTINCT vertex.
constexpr int kHashBits = 11, kHashSize = 1 << kHashBits;      // 2048 slots: 8 KB keys + 48 KB sums
constexpr int kBwdBatch = 1024;                                 // rays per table fill (6 vertex refs each)

struct HashAdd3 {
    int32_t* keys;      // LDS [kHashSize]
    double* sums;       // LDS [kHashSize * 3]
    double* g;          // global fallback / final target
    __device__ __forceinline__ void operator()(int32_t v, d3 a) const {
        unsigned h = ((unsigned)v * 2654435761u) >> (32 - kHashBits);
#pragma unroll 1
        for (int probe = 0; probe < 24; ++probe) {
            int32_t k = keys[h];
            if (k == -1) k = atomicCAS(&keys[h], -1, v);
            if (k == -1 || k == v) {
                __hip_atomic_fetch_add(&sums[3 * h + 0], a.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&sums[3 * h + 1], a.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&sums[3 * h + 2], a.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return;
            }
            h = (h + 1) & (kHashSize - 1);
        }
        AtomicAdd3{g}(v, a);     // table crowded: straight to memory
    }
};

__device__ __forceinline__ void hash_clear(int32_t* keys, double* sums) {
    for (int i = threadIdx.x; i < kHashSize; i += blockDim.x) keys[i] = -1;
    for (int i = threadIdx.x; i < 3 * kHashSize; i += blockDim.x) sums[i] = 0.0;
    __syncthreads();
}
__device__ __forceinline__ void hash_flush(int32_t* keys, double* sums, double* g) {
    __syncthreads();
    for (int i = threadIdx.x; i < kHashSize; i += blockDim.x) {
        const int32_t v = keys[i];
        if (v >= 0) AtomicAdd3{g}(v, d3{sums[3 * i], sums[3 * i + 1], sums[3 * i + 2]});
    }
    __syncthreads();
}

// Backward without a saved list: compact the rays whose path completed (face2 >= 0).
__global__ void __launch_bounds__(kPathBlock) k_collect_valid(const int32_t* __restrict__ face2, int64_t n, int64_t chunk_base,
                                                               int32_t* __restrict__ list, unsigned* counter) {
    __shared__ unsigned s_tmp[kPathWaves + 1];
    for (int64_t base = blockIdx.x * (int64_t)kPathBlock; base < n; base += (int64_t)gridDim.x * kPathBlock) {
        const int64_t i = base + threadIdx.x;
        const int slot = block_push(i < n && face2[i] >= 0, counter, s_tmp);
        if (slot >= 0) list[slot] = (int32_t)(chunk_base + i);
    }
}

// Backward (full waves over the list of valid rays): recompute both bounces from (face1, face2),
// reverse them, scatter the six vertex gradients.
__global__ void __launch_bounds__(256) k_render_bwd(PathCtx c, const double* __restrict__ origin, const double* __restrict__ dir,
                                                    const int32_t* __restrict__ face1, const int32_t* __restrict__ face2,
                                                    const double* __restrict__ g_out_ori, const double* __restrict__ g_out_dir,
                                                    double* grad_verts, const int32_t* __restrict__ list, const unsigned* __restrict__ n_u32,
                                                    const int64_t* __restrict__ n_i64) {
    __shared__ int32_t hkeys[kHashSize];
    __shared__ double hsums[3 * kHashSize];
    const int64_t n = n_i64 ? *n_i64 : (int64_t)*n_u32;
    const HashAdd3 add{hkeys, hsums, grad_verts};
    for (int64_t base = blockIdx.x * (int64_t)kBwdBatch; base < n; base += (int64_t)gridDim.x * kBwdBatch) {
        hash_clear(hkeys, hsums);
        const int64_t end = base + kBwdBatch < n ? base + kBwdBatch : n;
        for (int64_t k = base + threadIdx.x; k < end; k += blockDim.x) {
            const int64_t i = list[k];
            const d3 z{0.0, 0.0, 0.0};
            const d3 g_ori = g_out_ori ? load_d3(g_out_ori, i) : z;
            const d3 g_dir = g_out_dir ? load_d3(g_out_dir, i) : z;
            path_recompute_backward(c, load_d3(origin, i), load_d3(dir, i), face1[i], face2[i], g_ori, g_dir, add);
        }
        hash_flush(hkeys, hsums, grad_verts);
    }
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ray_loss forward: loss, dense d loss / d out_dir, and (optionally) the list of contributing rays so
// that the backward can rescale only those rows instead of streaming the whole [N,3] tensor again.
// Three things bound the obvious one-ray-per-thread version (tools/ubench/ray_loss_probe.py, 75.5 M rays): 1-byte loads
// of the flags (the vector-memory pipeline is paid per instruction: 0.42 ms for 0.3 GB), 8-byte strided stores of the
// (mostly zero) gradient, and one returning atomic per block iteration on the list counter (a single word sustains
// ~90 of them per microsecond: 3.3 ms when the contributing rays are scattered).  So: a thread takes FOUR consecutive
// rays (one 4-byte load of `valid`, three of `mask`), a wave zero-fills its 6 KB of gradient with lane-consecutive 16-byte
// stores (16-byte stores at a 96-byte lane stride were 3x slower: partial lines) and the block collects row indices in LDS,
// reserving list space with one atomic per ~2000 rows.
struct alignas(16) Dbl2 { double a, b; };
constexpr int kLossRays = 4;                        // rays per thread
constexpr int kLossBuf = 2048;                      // >= 2 x the 1024 rows one block iteration can add
__global__ void __launch_bounds__(kPathBlock) k_ray_loss(const double* __restrict__ out_ori, const double* __restrict__ out_dir,
                                                          const uint8_t* __restrict__ mask, const double* __restrict__ screen_pixel,
                                                          const uint8_t* __restrict__ valid, int64_t n, double* loss,
                                                          double* __restrict__ g_out_dir, int32_t* __restrict__ list, unsigned* list_count) {
    __shared__ unsigned s_tmp[kPathWaves + 2];      // wave totals, [kPathWaves] rows buffered, [kPathWaves + 1] reserved list base
    __shared__ int32_t s_buf[kLossBuf];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long lower = (1ull << lane) - 1ull;
    if (threadIdx.x == 0) s_tmp[kPathWaves] = 0u;
    __syncthreads();
    auto flush = [&]() {                            // reached by the whole block
        const unsigned cnt = s_tmp[kPathWaves];
        if (threadIdx.x == 0) s_tmp[kPathWaves + 1] = cnt ? atomicAdd(list_count, cnt) : 0u;
        __syncthreads();
        const unsigned base = s_tmp[kPathWaves + 1];
        for (unsigned k = threadIdx.x; k < cnt; k += kPathBlock) list[base + k] = s_buf[k];
        __syncthreads();
        if (threadIdx.x == 0) s_tmp[kPathWaves] = 0u;
        __syncthreads();
    };
    double acc = 0.0;
    const int64_t span = (int64_t)kLossRays * kPathBlock;
    for (int64_t base = blockIdx.x * span; base < n; base += (int64_t)gridDim.x * span) {
        const int64_t i0 = base + kLossRays * (int64_t)threadIdx.x;
        bool on[kLossRays] = {false, false, false, false};
        d3 g[kLossRays] = {d3{0.0, 0.0, 0.0}, d3{0.0, 0.0, 0.0}, d3{0.0, 0.0, 0.0}, d3{0.0, 0.0, 0.0}};
        const bool full = i0 + kLossRays - 1 < n;
        if (full) {
            const uint32_t v4 = *reinterpret_cast<const uint32_t*>(valid + i0);
            const uint32_t* mp = reinterpret_cast<const uint32_t*>(mask + 3 * i0);
            const uint32_t m0 = mp[0], m1 = mp[1], m2 = mp[2];
            on[0] = (v4 & 0xFFu) && (m0 & 0xFFu);                 // mask[3 i0]
            on[1] = (v4 & 0xFF00u) && (m0 & 0xFF000000u);         // mask[3 i0 + 3]
            on[2] = (v4 & 0xFF0000u) && (m1 & 0xFF0000u);         // mask[3 i0 + 6]
            on[3] = (v4 & 0xFF000000u) && (m2 & 0xFF00u);         // mask[3 i0 + 9]
        } else {
            for (int k = 0; k < kLossRays; ++k) on[k] = i0 + k < n && valid[i0 + k] && mask[3 * (i0 + k)];
        }
        for (int k = 0; k < kLossRays; ++k)
            if (on[k]) acc += ray_loss_term(load_d3(out_ori, i0 + k), load_d3(out_dir, i0 + k), load_d3(screen_pixel, i0 + k), g[k]);
        if (g_out_dir) {
            const int64_t w0 = i0 - kLossRays * (int64_t)lane;                  // first ray of this wave: 256 rays = 6144 contiguous bytes
            if (w0 + kLossRays * 64 <= n) {
                // zeros for the whole run with fully coalesced 16-byte stores (lane-consecutive), then the few rows that
                // carry a gradient are overwritten -- after the zero stores have been acknowledged (s_waitcnt)
                Dbl2* q = reinterpret_cast<Dbl2*>(g_out_dir + 3 * w0);
                for (int j = 0; j < 6; ++j) q[j * 64 + lane] = Dbl2{0.0, 0.0};
                if (__ballot(on[0] | on[1] | on[2] | on[3]) != 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    for (int k = 0; k < kLossRays; ++k) if (on[k]) store_d3(g_out_dir, i0 + k, g[k]);
                }
            } else {
                for (int k = 0; k < kLossRays; ++k) if (i0 + k < n) store_d3(g_out_dir, i0 + k, g[k]);
            }
        }
        if (list) {
            unsigned long long m[kLossRays];
            unsigned before[kLossRays], tot = 0;
            for (int k = 0; k < kLossRays; ++k) { m[k] = __ballot(on[k]); before[k] = tot; tot += (unsigned)__popcll(m[k]); }
            if (lane == 0) s_tmp[wave] = tot;
            __syncthreads();
            unsigned wbase = s_tmp[kPathWaves], add = 0;
            for (int w = 0; w < kPathWaves; ++w) { const unsigned c = s_tmp[w]; if (w < wave) wbase += c; add += c; }
            if (add) {                               // block-uniform
                for (int k = 0; k < kLossRays; ++k)
                    if (on[k]) s_buf[wbase + before[k] + (unsigned)__popcll(m[k] & lower)] = (int32_t)(i0 + k);
                __syncthreads();
                if (threadIdx.x == 0) s_tmp[kPathWaves] += add;
                __syncthreads();
                if (s_tmp[kPathWaves] > kLossBuf - kLossRays * kPathBlock) flush();
            } else {
                __syncthreads();
            }
        }
    }
    if (list) flush();
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0 && acc != 0.0) unsafeAtomicAdd(loss, acc);
}

// x[list[k], 0..2] *= *scale
__global__ void __launch_bounds__(256) k_scale_rows3(double* __restrict__ x, const int32_t* __restrict__ list, const unsigned* __restrict__ n_ptr,
                                                     const double* __restrict__ scale) {
    const unsigned n = *n_ptr;
    const double sc = *scale;
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const int64_t i = list[k];
        x[3 * i] *= sc; x[3 * i + 1] *= sc; x[3 * i + 2] *= sc;
    }
}

// Fused loss, last stage (full waves over Q2): recompute the path in float64, loss term, adjoint.
__global__ void __launch_bounds__(256) k_loss_bwd_fused(PathCtx c, const double* __restrict__ origin, const double* __restrict__ dir,
                                                        const double* __restrict__ screen_pixel, const int32_t* __restrict__ face1,
                                                        const int32_t* __restrict__ face2, Pipe p, double* loss, double* grad_verts,
                                                        unsigned long long* n_valid) {
    __shared__ int32_t hkeys[kHashSize];
    __shared__ double hsums[3 * kHashSize];
    const unsigned n2 = p.count[2];
    const HashAdd3 add{hkeys, hsums, grad_verts};
    double acc = 0.0;
    unsigned cnt = 0;
    for (unsigned base = blockIdx.x * kBwdBatch; base < n2; base += gridDim.x * kBwdBatch) {
        hash_clear(hkeys, hsums);
        const unsigned end = base + kBwdBatch < n2 ? base + kBwdBatch : n2;
        for (unsigned k = base + threadIdx.x; k < end; k += blockDim.x) {
            if (p.r2.face[k] >= 0) continue;   // occluded exit ray
            const int64_t i = p.r2.idx[k];
            const int32_t f2 = face2[i];
            const d3 o = load_d3(origin, i), d = load_d3(dir, i);
            d3 v0, v1, v2;
            int32_t vid1[3], vid2[3];
            Bounce b1, b2;
            load_tri64(c, face1[i], v0, v1, v2, vid1);
            bounce_forward(o, d, v0, v1, v2, c.ior_ext, c.ior_int, b1);
            load_tri64(c, f2, v0, v1, v2, vid2);
            bounce_forward(b1.new_o, b1.wt, v0, v1, v2, c.ior_ext, c.ior_int, b2);
            d3 g_dir;
            acc += ray_loss_term(b2.new_o, b2.wt, load_d3(screen_pixel, i), g_dir);
            ++cnt;
            const d3 z{0.0, 0.0, 0.0};
            d3 ga = z, gb = z, gc = z, g_o, g_d, g_o0, g_d0;
            bounce_backward(b2, z, g_dir, ga, gb, gc, g_o, g_d);
            add(vid2[0], ga); add(vid2[1], gb); add(vid2[2], gc);
            ga = z; gb = z; gc = z;
            bounce_backward(b1, g_o, g_d, ga, gb, gc, g_o0, g_d0);
            add(vid1[0], ga); add(vid1[1], gb); add(vid1[2], gc);
        }
        hash_flush(hkeys, hsums, grad_verts);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0 && acc != 0.0) unsafeAtomicAdd(loss, acc);
    if (n_valid && cnt) atomicAdd(n_valid, (unsigned long long)cnt);
}

// ---- silhouette and smoothness branches (per unique edge) -------------------------------------
__device__ __forceinline__ void load_face64(const double* __restrict__ verts, const int64_t* __restrict__ f, d3& v0, d3& v1, d3& v2) {
    v0 = load_d3(verts, f[0]); v1 = load_d3(verts, f[1]); v2 = load_d3(verts, f[2]);
}

// cos of the dihedral angle of every edge (reference DiffRender.py:440-443)
__global__ void __launch_bounds__(256) k_dihedral_fwd(const double* __restrict__ verts, const int64_t* __restrict__ e2f, int64_t n,
                                                      double* __restrict__ cos_out) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= n) return;
    d3 v0, v1, v2;
    FaceNormal a, b;
    load_face64(verts, e2f + 6 * e, v0, v1, v2); face_normal(v0, v1, v2, a);
    load_face64(verts, e2f + 6 * e + 3, v0, v1, v2); face_normal(v0, v1, v2, b);
    cos_out[e] = dot(a.n, b.n);
}

// MODE 0: adjoint of k_dihedral_fwd for a given d loss / d cos.
// MODE 1: sm_loss = sum -log(1 + cos) (reference optim.py:82-89) and its vertex gradient in one pass.
template <int MODE>
__global__ void __launch_bounds__(256) k_dihedral_bwd(const double* __restrict__ verts, const int64_t* __restrict__ e2f, int64_t n,
                                                      const double* __restrict__ g_cos, double* loss, double* grad_verts) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    double term = 0.0;
    if (e < n) {
        d3 v0, v1, v2;
        FaceNormal a, b;
        const int64_t* fa = e2f + 6 * e;
        const int64_t* fb = fa + 3;
        load_face64(verts, fa, v0, v1, v2); face_normal(v0, v1, v2, a);
        load_face64(verts, fb, v0, v1, v2); face_normal(v0, v1, v2, b);
        double g;
        if (MODE == 0) {
            g = g_cos[e];
        } else {
            const double c = dot(a.n, b.n);
            term = -log(1.0 + c);
            g = -1.0 / (1.0 + c);
        }
        const d3 z{0.0, 0.0, 0.0};
        d3 g0 = z, g1 = z, g2 = z;
        const AtomicAdd3 add{grad_verts};
        face_normal_backward(a, g * b.n, g0, g1, g2);
        add((int32_t)fa[0], g0); add((int32_t)fa[1], g1); add((int32_t)fa[2], g2);
        g0 = z; g1 = z; g2 = z;
        face_normal_backward(b, g * a.n, g0, g1, g2);
        add((int32_t)fb[0], g0); add((int32_t)fb[1], g1); add((int32_t)fb[2], g2);
    }
    if (MODE == 1) {
        term = wave_sum(term);
        if ((threadIdx.x & 63) == 0 && term != 0.0) unsafeAtomicAdd(loss, term);
    }
}

// silhouette test per unique edge (reference DiffRender.py:445-457)
__global__ void __launch_bounds__(256) k_silhouette_flags(const double* __restrict__ verts, const int64_t* __restrict__ e2f, int64_t n,
                                                          const double* __restrict__ origin3, uint8_t* __restrict__ flags) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const d3 o{origin3[0], origin3[1], origin3[2]};
    d3 a0, b0, v1, v2;
    FaceNormal a, b;
    load_face64(verts, e2f + 6 * e, a0, v1, v2); face_normal(a0, v1, v2, a);
    load_face64(verts, e2f + 6 * e + 3, b0, v1, v2); face_normal(b0, v1, v2, b);
    flags[e] = silhouette_flag(a, a0, b, b0, o) ? 1 : 0;
}

// primary_visibility + primary_edge_sample.forward for every silhouette edge: project the two
// endpoints, probe one pixel either side of the edge midpoint with any-hit rays, f = hit+ - hit-.
__global__ void __launch_bounds__(kTraceBlock) k_edge_sample_fwd(TraceCtx c, const double* __restrict__ verts, const int64_t* __restrict__ edges,
                                                                  int64_t n, const Camera* __restrict__ cam, const double* __restrict__ origin3,
                                                                  int64_t* __restrict__ index, float* __restrict__ f_out) {
    __shared__ int32_t lds[kStackFast][kTraceBlock];
    Stack st = make_stack(lds, c);
    const Camera cm = *cam;
    const d3 o{origin3[0], origin3[1], origin3[2]};
    for (int64_t e = blockIdx.x * (int64_t)kTraceBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kTraceBlock) {
        Projected pa, pb;
        project_endpoint(cm, load_d3(verts, edges[2 * e]), pa);
        project_endpoint(cm, load_d3(verts, edges[2 * e + 1]), pb);
        EdgeSample s;
        edge_sample(cm, pa, pb, o, s);
        const bool hu = traverse<true>(c.nodes, c.tris, c.n_tris, to_f32(o), to_f32(s.dir_up), st).face >= 0;
        const bool hl = traverse<true>(c.nodes, c.tris, c.n_tris, to_f32(o), to_f32(s.dir_lo), st).face >= 0;
        f_out[e] = (hu ? 1.0f : 0.0f) - (hl ? 1.0f : 0.0f);
        index[2 * e] = (int64_t)s.midx;       // truncation toward zero, like Tensor.to(torch.long)
        index[2 * e + 1] = (int64_t)s.midy;
    }
}

// Adjoint: dE_pos[e, endpoint, :] = -N_e * f_e * coef_e for both endpoints (reference
// DiffRender.py:236-242, 263-267), chained through the projection to the two vertices.
__global__ void __launch_bounds__(256) k_edge_sample_bwd(const double* __restrict__ verts, const int64_t* __restrict__ edges, int64_t n,
                                                         const Camera* __restrict__ cam, const float* __restrict__ f,
                                                         const double* __restrict__ coef, int detach_depth, double* grad_verts) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const double w = (double)f[e] * coef[e];
    if (w == 0.0) return;
    const Camera cm = *cam;
    const int64_t ia = edges[2 * e], ib = edges[2 * e + 1];
    Projected pa, pb;
    project_endpoint(cm, load_d3(verts, ia), pa);
    project_endpoint(cm, load_d3(verts, ib), pb);
    const double gx = -(pa.py - pb.py) * w, gy = -(pb.px - pa.px) * w;
    const AtomicAdd3 add{grad_verts};
    add((int32_t)ia, project_endpoint_backward(cm, pa, gx, gy, detach_depth != 0));
    add((int32_t)ib, project_endpoint_backward(cm, pb, gx, gy, detach_depth != 0));
}

// ---- closest point on the mesh (the reference's acceptance metric, README.md:11: vertex-to-surface distance)
__global__ void __launch_bounds__(kTraceBlock) k_closest_point(TraceCtx c, const int32_t* __restrict__ faces, const float* __restrict__ verts,
                                                                const double* __restrict__ points, int64_t n, double* __restrict__ dist,
                                                                int32_t* __restrict__ face, double* __restrict__ closest) {
    __shared__ int32_t lds[kStackFast][kTraceBlock];
    Stack st = make_stack(lds, c);
    for (int64_t i = blockIdx.x * (int64_t)kTraceBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kTraceBlock) {
        const Closest r = closest_point(c.nodes, c.tris, c.n_tris, faces, verts, load_d3(points, i), st);
        dist[i] = sqrt(r.dist2);
        if (face) face[i] = r.face;
        if (closest) store_d3(closest, i, r.point);
    }
}

// ---- fused silhouette loss: Loss_calculator.vh_loss (reference optim.py:73-78) with no host round trip:
// the drop-in methods return dynamically sized tensors (two device->host syncs per view); here the
// silhouette edges of up to kVhViews views are compacted on the device into ONE list and one kernel does
// projection, probe rays, the loss term |soft_mask[y, x] - 0.5| and its vertex gradient.  Views are
// batched because a view has only a few thousand silhouette edges and its probe rays graze the surface:
// a per-view launch is bound by the latency of its longest traversal, not by throughput.
constexpr int kVhViews = 16;
struct VhViews {
    const double* cam[kVhViews];
    const double* origin[kVhViews];
    const double* soft[kVhViews];
};

__global__ void __launch_bounds__(kPathBlock) k_vh_cull(const double* __restrict__ verts, const int64_t* __restrict__ e2f, int64_t n_edges,
                                                         int n_views, VhViews vw, uint32_t* __restrict__ list, unsigned* count) {
    __shared__ unsigned s_tmp[kPathWaves + 1];
    const int64_t n = n_edges * n_views;
    for (int64_t base = blockIdx.x * (int64_t)kPathBlock; base < n; base += (int64_t)gridDim.x * kPathBlock) {
        const int64_t k = base + threadIdx.x;
        bool sil = false;
        if (k < n) {
            const int64_t e = k % n_edges;
            const double* o3 = vw.origin[k / n_edges];
            d3 a0, b0, v1, v2;
            FaceNormal a, b;
            load_face64(verts, e2f + 6 * e, a0, v1, v2); face_normal(a0, v1, v2, a);
            load_face64(verts, e2f + 6 * e + 3, b0, v1, v2); face_normal(b0, v1, v2, b);
            sil = silhouette_flag(a, a0, b, b0, d3{o3[0], o3[1], o3[2]});
        }
        const int slot = block_push(sil, count, s_tmp);
        if (slot >= 0) list[slot] = (uint32_t)k;
    }
}

__global__ void __launch_bounds__(kTraceBlock) k_vh_fused(TraceCtx c, const double* __restrict__ verts, const int64_t* __restrict__ edges,
                                                           uint32_t n_edges, const uint32_t* __restrict__ list, const unsigned* __restrict__ count,
                                                           VhViews vw, int resx, int resy, int detach_depth, double* loss, double* grad_verts) {
    __shared__ int32_t lds[kStackFast][kTraceBlock];
    Stack st = make_stack(lds, c);
    const unsigned n = *count;
    double acc = 0.0;
    // Two lanes per edge, one probe ray each: the rays graze the silhouette and take a few hundred node visits, and with
    // only a few thousand edges per view the kernel lasts as long as its longest lane -- tracing the two probes of an
    // edge one after the other in one lane doubled that.
    constexpr unsigned kPairs = kTraceBlock / 2;
    const int side = threadIdx.x & 1;
    for (unsigned base = blockIdx.x * kPairs; base < n; base += gridDim.x * kPairs) {       // block-uniform trip count (shuffles below)
        const unsigned k = base + (threadIdx.x >> 1);
        const bool live = k < n;
        const uint32_t item = live ? list[k] : 0u, view = item / n_edges;
        const int64_t e = item - view * n_edges;
        const Camera cm = *reinterpret_cast<const Camera*>(vw.cam[view]);
        const double* o3 = vw.origin[view];
        const d3 o{o3[0], o3[1], o3[2]};
        const int64_t ia = edges[2 * e], ib = edges[2 * e + 1];
        Projected pa, pb;
        project_endpoint(cm, load_d3(verts, ia), pa);
        project_endpoint(cm, load_d3(verts, ib), pb);
        EdgeSample s;
        edge_sample(cm, pa, pb, o, s);
        const int mine = live && traverse<true>(c.nodes, c.tris, c.n_tris, to_f32(o), to_f32(side == 0 ? s.dir_up : s.dir_lo), st).face >= 0 ? 1 : 0;
        const int other = __shfl_xor(mine, 1);
        if (!live || side != 0) continue;                         // the even lane of the pair finishes the edge
        const double f = (double)mine - (double)other;           // hit(up) - hit(lo)
        if (f == 0.0) continue;                                   // |f| > 1e-5 (DiffRender.py:244)
        const int64_t x = (int64_t)s.midx, y = (int64_t)s.midy;   // trunc, like Tensor.to(torch.long)
        if (!(x < resx - 1 && y < resy - 1 && x >= 0 && y >= 0)) continue;   // out of view (DiffRender.py:478)
        const double m = vw.soft[view][y * resx + x] - 0.5;       // output is float32 0.5: exact
        acc += fabs(m);
        const double coef = m > 0.0 ? -1.0 : (m < 0.0 ? 1.0 : 0.0);   // d |mask - output| / d output
        const double w = f * coef;
        if (w == 0.0) continue;
        const double gx = -(pa.py - pb.py) * w, gy = -(pb.px - pa.px) * w;
        const AtomicAdd3 add{grad_verts};
        add((int32_t)ia, project_endpoint_backward(cm, pa, gx, gy, detach_depth != 0));
        add((int32_t)ib, project_endpoint_backward(cm, pb, gx, gy, detach_depth != 0));
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0 && acc != 0.0) unsafeAtomicAdd(loss, acc);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int grid_for(int64_t n, int block, int cap) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

static int ensure_capacity(drt_scene* s, int64_t n_faces, int64_t n_verts) {
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

struct StageTimer;
static int rebuild_impl(drt_scene* s, hipStream_t st);

static int rebuild_impl(drt_scene* s, hipStream_t st) {
    const int n = (int)s->n_faces;
    s->built = true;
    if (n == 0) return DRT_OK;
    const int tiles = (n + kSortTile - 1) / kSortTile;
    const bool fused_sort = tiles <= kSortFusedTiles;
    const int table = kRadix * tiles;
    k_bounds<<<1, 1024, 0, st>>>(s->verts, s->n_verts, s->params, s->hist, fused_sort ? 4 * table : 0);
    k_morton<<<(n + 255) / 256, 256, 0, st>>>(s->faces, s->verts, n, s->params, s->keys[0], s->idx[0], fused_sort ? s->hist : nullptr, tiles);
    int cur = 0;
    for (int shift = 0, pass = 0; shift < 30; shift += 8, ++pass) {
        if (fused_sort) {
            k_sort_pass_fused<<<tiles, kSortBlock, 0, st>>>(s->keys[cur], s->idx[cur], s->keys[cur ^ 1], s->idx[cur ^ 1], n, shift,
                                                            s->hist + pass * table, pass < 3 ? s->hist + (pass + 1) * table : nullptr, tiles);
        } else {
            k_sort_hist<<<tiles, kSortBlock, 0, st>>>(s->keys[cur], n, shift, s->hist, tiles);
            k_sort_scan<<<1, 1024, 0, st>>>(s->hist, kRadix * tiles);
            k_sort_scatter<<<tiles, kSortBlock, 0, st>>>(s->keys[cur], s->idx[cur], s->keys[cur ^ 1], s->idx[cur ^ 1], n, shift, s->hist, tiles);
        }
        cur ^= 1;
    }
    // four passes -> result is back in buffer 0
    const int inner = n > 1 ? n - 1 : 1;
    k_hierarchy<<<(inner + 255) / 256, 256, 0, st>>>(s->keys[cur], n, s->nodes, s->parent_inner, s->parent_leaf, s->flags, s->range_lo, s->range_hi);
    k_refit<<<(n + 255) / 256, 256, 0, st>>>(s->idx[cur], s->faces, s->verts, n, s->params, s->tris, s->nodes,
                                             s->parent_inner, s->parent_leaf, s->flags);
    k_collapse4<<<(inner + 255) / 256, 256, 0, st>>>(s->nodes, s->parent_inner, s->range_lo, s->range_hi, n, s->wide);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

static int rebuild(drt_scene* s, hipStream_t st);

static TraceCtx trace_ctx(const drt_scene* s) { return TraceCtx{s->wide, s->tris, (int)s->n_faces, s->slow_stack}; }

#define CHECK_SCENE(s)                                                        \
    do {                                                                      \
        if (!(s)) return fail(DRT_E_INVALID, "null scene");                   \
        HIP_TRY(hipSetDevice((s)->device));                                   \
    } while (0)
#define CHECK_BUILT(s)                                                                          \
    do {                                                                                        \
        CHECK_SCENE(s);                                                                         \
        if (!(s)->built) return fail(DRT_E_INVALID, "no mesh: call drt_update_mesh first");     \
    } while (0)

extern "C" {

const char* drt_last_error(void) { return g_err; }
int drt_version(void) { return 1; }

int drt_create(int device, drt_scene_t** out) {
    if (!out) return fail(DRT_E_INVALID, "out is null");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(DRT_E_INVALID, "device %d out of range (%d visible)", device, count);
    HIP_TRY(hipSetDevice(device));
    drt_scene* s = new (std::nothrow) drt_scene();
    if (!s) return fail(DRT_E_NOMEM, "host allocation failed");
    s->device = device;
    hipError_t e = hipMalloc(&s->params, sizeof(BuildParams));
    if (e == hipSuccess) e = hipMalloc(&s->slow_stack, sizeof(int32_t) * (size_t)kTraceGridMax * kTraceBlock * kStackSlowDev);
    if (e == hipSuccess) e = hipMalloc(&s->scratch, sizeof(unsigned long long) * 8);
    if (e == hipSuccess) e = hipMalloc(&s->vcount, sizeof(unsigned) * 4);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->fork_ev, hipEventDisableTiming);
    if (const char* ev = getenv("DRT_STREAMS")) { const int v = atoi(ev); if (v >= 1 && v <= drt_scene::kMaxSub) s->n_sub = v; }
    if (const char* ev = getenv("DRT_SUB_PER_STREAM")) { const int v = atoi(ev); if (v >= 1 && v <= 16) s->sub_per_stream = v; }
    if (const char* ev = getenv("DRT_MIN_SUB_LOG2")) { const int v = atoi(ev); if (v >= 12 && v <= 30) s->min_sub_rays = (int64_t)1 << v; }
    for (int k = 0; k < s->n_sub && e == hipSuccess; ++k) {
        drt_scene::Sub& w = s->sub[k];
        e = hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&w.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc(&w.qcount, sizeof(unsigned) * 8);
        if (e == hipSuccess) e = hipMalloc(&w.slow_stack, sizeof(int32_t) * (size_t)kTraceGridMax * kTraceBlock * kStackSlowDev);
    }
    if (e == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) s->n_cu = prop.multiProcessorCount;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_intersect<false>, kTraceBlock, 0) != hipSuccess || per_cu < 1) per_cu = 8;
        s->grid_trace = s->n_cu * per_cu;
        if (s->grid_trace > kTraceGridMax) s->grid_trace = kTraceGridMax;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace<false>, kPathBlock, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        s->grid_path = s->n_cu * per_cu;
        if (s->grid_path * 2 > kTraceGridMax) s->grid_path = kTraceGridMax / 2;
        // tuning knobs (measurement only; defaults are the tuned values)
        if (const char* e = getenv("DRT_TRACE_BPC")) { const int v = atoi(e); if (v >= 1 && v * s->n_cu * 2 <= kTraceGridMax) s->grid_path = v * s->n_cu; }
        if (const char* e = getenv("DRT_INNER_MIN")) { const int v = atoi(e); if (v >= 1 && v <= 64) s->inner_min = v; }
        if (const char* e = getenv("DRT_REFILL_MIN")) { const int v = atoi(e); if (v >= 1 && v <= 64) s->refill_min = v; }
        if (const char* e = getenv("DRT_CHUNK_LOG2")) { const int v = atoi(e); if (v >= 16 && v <= 30) s->chunk_rays = (int64_t)1 << v; }
    }
    if (e != hipSuccess) {
        drt_destroy(s);
        return fail(DRT_E_HIP, "hipMalloc: %s", hipGetErrorString(e));
    }
    *out = s;
    return DRT_OK;
}

void drt_destroy(drt_scene_t* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    scene_free_mesh(s);
    (void)hipFree(s->params);
    (void)hipFree(s->slow_stack);
    (void)hipFree(s->scratch);
    for (int j = 0; j < drt_scene::kMaxSub; ++j) {
        drt_scene::Sub& w = s->sub[j];
        for (int k = 0; k < 3; ++k) { (void)hipFree(w.q_idx[k]); (void)hipFree(w.q_ray[k]); (void)hipFree(w.q_face[k]); }
        (void)hipFree(w.tmp_face1); (void)hipFree(w.tmp_face2); (void)hipFree(w.qcount); (void)hipFree(w.slow_stack); (void)hipFree(w.redo);
        if (w.done) (void)hipEventDestroy(w.done);
        if (w.stream) (void)hipStreamDestroy(w.stream);
    }
    (void)hipFree(s->vcount);
    (void)hipFree(s->vh_list);
    if (s->fork_ev) (void)hipEventDestroy(s->fork_ev);
    for (auto& e : s->prof_ev) (void)hipEventDestroy(e);
    (void)hipFree(s->prof_counts);
    delete s;
}

int drt_update_mesh(drt_scene_t* s, const int32_t* d_faces, int64_t n_faces, const float* d_verts, int64_t n_verts, void* stream) {
    CHECK_SCENE(s);
    if (n_faces < 0 || n_verts < 0 || (n_faces && !d_faces) || (n_verts && !d_verts)) return fail(DRT_E_INVALID, "bad mesh arguments");
    if (n_faces > (int64_t)1 << 30) return fail(DRT_E_INVALID, "too many faces");
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_capacity(s, n_faces, n_verts);
    if (rc) return rc;
    s->n_faces = n_faces;
    s->n_verts = n_verts;
    if (n_faces) HIP_TRY(hipMemcpyAsync(s->faces, d_faces, sizeof(int32_t) * 3 * n_faces, hipMemcpyDeviceToDevice, st));
    if (n_verts) HIP_TRY(hipMemcpyAsync(s->verts, d_verts, sizeof(float) * 3 * n_verts, hipMemcpyDeviceToDevice, st));
    return rebuild(s, st);
}

int drt_update_vert(drt_scene_t* s, const float* d_verts, int64_t n_verts, void* stream) {
    CHECK_BUILT(s);
    if (n_verts != s->n_verts || (n_verts && !d_verts)) return fail(DRT_E_INVALID, "vertex count %lld != %lld", (long long)n_verts, (long long)s->n_verts);
    hipStream_t st = (hipStream_t)stream;
    if (n_verts) HIP_TRY(hipMemcpyAsync(s->verts, d_verts, sizeof(float) * 3 * n_verts, hipMemcpyDeviceToDevice, st));
    return rebuild(s, st);
}

int drt_update_vert_f64(drt_scene_t* s, const double* d_verts, int64_t n_verts, void* stream) {
    CHECK_BUILT(s);
    if (n_verts != s->n_verts || (n_verts && !d_verts)) return fail(DRT_E_INVALID, "vertex count %lld != %lld", (long long)n_verts, (long long)s->n_verts);
    hipStream_t st = (hipStream_t)stream;
    if (n_verts) k_cast_verts<<<grid_for(3 * n_verts, 256, 1024), 256, 0, st>>>(d_verts, s->verts, 3 * n_verts);
    return rebuild(s, st);
}

int drt_intersect(drt_scene_t* s, const float* d_rays, int64_t n_rays, float* d_T, int32_t* d_ID, void* stream) {
    CHECK_BUILT(s);
    if (n_rays < 0 || (n_rays && (!d_rays || !d_T || !d_ID))) return fail(DRT_E_INVALID, "bad ray arguments");
    if (n_rays == 0) return DRT_OK;
    k_intersect<false><<<grid_for(n_rays, kTraceBlock, s->grid_trace), kTraceBlock, 0, (hipStream_t)stream>>>(trace_ctx(s), d_rays, n_rays, d_T, d_ID, nullptr);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_intersect_any(drt_scene_t* s, const float* d_rays, int64_t n_rays, uint8_t* d_hit, void* stream) {
    CHECK_BUILT(s);
    if (n_rays < 0 || (n_rays && (!d_rays || !d_hit))) return fail(DRT_E_INVALID, "bad ray arguments");
    if (n_rays == 0) return DRT_OK;
    k_intersect<true><<<grid_for(n_rays, kTraceBlock, s->grid_trace), kTraceBlock, 0, (hipStream_t)stream>>>(trace_ctx(s), d_rays, n_rays, nullptr, nullptr, d_hit);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_intersect_bruteforce(drt_scene_t* s, const float* d_rays, int64_t n_rays, float* d_T, int32_t* d_ID, void* stream) {
    CHECK_BUILT(s);
    if (n_rays < 0 || (n_rays && (!d_rays || !d_T || !d_ID))) return fail(DRT_E_INVALID, "bad ray arguments");
    if (n_rays == 0) return DRT_OK;
    k_bruteforce<<<(unsigned)((n_rays + 255) / 256), 256, 0, (hipStream_t)stream>>>(s->tris, (int)s->n_faces, d_rays, n_rays, d_T, d_ID);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_bvh_check(drt_scene_t* s, void* stream, int64_t* n_violations, int32_t* height, int32_t* wide_depth) {
    CHECK_BUILT(s);
    hipStream_t st = (hipStream_t)stream;
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
    if (s->n_faces) HIP_TRY(hipMemcpyAsync(d_order, s->idx[0], sizeof(int32_t) * s->n_faces, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DRT_OK;
}

static PathCtx path_ctx(const drt_scene* s, const double* d_verts, double ior_int, double ior_ext) {
    return PathCtx{trace_ctx(s), s->faces, d_verts, ior_int, ior_ext};
}

// RAII-ish stage timer: records an event pair around a kernel launch when profiling is on.
struct StageTimer {
    drt_scene* s; hipStream_t st; bool on;
    StageTimer(drt_scene* s_, hipStream_t st_, int stage) : s(s_), st(st_), on(false) {
        if (!s->prof_on || s->prof_used + 2 > s->prof_ev.size()) return;
        on = true;
        s->prof_stage[s->prof_used / 2] = stage;
        if (!s->prof_stream) s->prof_stream = st;
        (void)hipEventRecord(s->prof_ev[s->prof_used], st);
    }
    ~StageTimer() {
        if (!on) return;
        (void)hipEventRecord(s->prof_ev[s->prof_used + 1], st);
        s->prof_used += 2;
    }
};

__global__ void k_prof_counts(const unsigned* __restrict__ qcount, unsigned long long n_rays, unsigned long long* __restrict__ tot, int fused) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // sub-batches on different streams may report concurrently: atomics
    atomicAdd(&tot[kStageCull], n_rays);
    atomicAdd(&tot[kStageTrace1], (unsigned long long)qcount[0]); atomicAdd(&tot[kStageShade1], (unsigned long long)qcount[0]);
    atomicAdd(&tot[kStageTrace2], (unsigned long long)qcount[1]); atomicAdd(&tot[kStageShade2], (unsigned long long)qcount[1]);
    atomicAdd(&tot[kStageTrace3], (unsigned long long)qcount[2]);
    atomicAdd(&tot[fused ? kStageLossBwdFused : kStageFinish], (unsigned long long)qcount[2]);
}
__global__ void k_prof_counts_bwd(const unsigned* __restrict__ vcount, unsigned long long n_rays, unsigned long long* __restrict__ tot) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    tot[kStageCollect] += n_rays;
    tot[kStageBackward] += vcount[0];
}

static int rebuild(drt_scene* s, hipStream_t st) {
    StageTimer t(s, st, kStageBuild);
    return rebuild_impl(s, st);
}

// Queue workspace for one chunk of `n` rays (grown, never shrunk).  Growing frees the old buffers,
// which synchronises the device once; steady-state calls allocate nothing.
static int ensure_queues(drt_scene::Sub& w, int64_t n, bool fused) {
    if (n > w.q_cap) {
        for (int k = 0; k < 3; ++k) {
            (void)hipFree(w.q_idx[k]); (void)hipFree(w.q_ray[k]); (void)hipFree(w.q_face[k]);
            w.q_idx[k] = nullptr; w.q_ray[k] = nullptr; w.q_face[k] = nullptr;
        }
        w.q_cap = 0;
        for (int k = 0; k < 3; ++k) {
            HIP_TRY(hipMalloc(&w.q_idx[k], sizeof(int32_t) * n));
            HIP_TRY(hipMalloc(&w.q_ray[k], sizeof(float) * 6 * n));
            HIP_TRY(hipMalloc(&w.q_face[k], sizeof(int32_t) * n));
        }
        (void)hipFree(w.redo); w.redo = nullptr;
        HIP_TRY(hipMalloc(&w.redo, sizeof(int32_t) * n));
        w.q_cap = n;
    }
    if (fused && n > w.fused_cap) {
        (void)hipFree(w.tmp_face1); (void)hipFree(w.tmp_face2);
        w.tmp_face1 = w.tmp_face2 = nullptr; w.fused_cap = 0;
        HIP_TRY(hipMalloc(&w.tmp_face1, sizeof(int32_t) * n));
        HIP_TRY(hipMalloc(&w.tmp_face2, sizeof(int32_t) * n));
        w.fused_cap = n;
    }
    return DRT_OK;
}

static Pipe pipe_of(const drt_scene* s, const drt_scene::Sub& w) {
    return Pipe{RayList{w.q_idx[0], w.q_ray[0], w.q_face[0]}, RayList{w.q_idx[1], w.q_ray[1], w.q_face[1]},
                RayList{w.q_idx[2], w.q_ray[2], w.q_face[2]}, w.qcount, s->vcount, w.redo};
}

// How a call of n_rays is cut: `size` rays per sub-batch (a multiple of `unit`), `count` sub-batches, dealt
// round-robin to `streams` internal streams.  Sub-batches are at most chunk_rays (workspace bound) and, when
// there is enough work, at least min_sub_rays, so that small calls are not shredded into launch overhead.
struct Plan { int64_t size; int count; int streams; };
static Plan plan_call(const drt_scene* s, int64_t n_rays, int tile_w) {
    int64_t unit = 256;
    if (tile_w >= 64 && tile_w % 64 == 0) unit = 4 * (int64_t)tile_w;       // whole rows of 64x4 patches per sub-batch
    int64_t count = (n_rays + s->chunk_rays - 1) / s->chunk_rays;
    const int64_t by_min = n_rays / s->min_sub_rays;
    const int64_t most = (int64_t)s->n_sub * s->sub_per_stream;
    const int64_t want = by_min < most ? by_min : most;
    if (want > count) count = want;
    if (count < 1) count = 1;
    int64_t size = (n_rays + count - 1) / count;
    size = (size + unit - 1) / unit * unit;
    count = (n_rays + size - 1) / size;
    Plan pl;
    pl.size = size; pl.count = (int)count; pl.streams = count < s->n_sub ? (int)count : s->n_sub;
    if (s->prof_serial) pl.streams = 1;     // measurement mode: the same sub-batches, one after the other on one stream
    return pl;
}

// cull -> trace -> shade1 -> trace -> shade2 -> trace(any) for one sub-batch; the caller appends the last stage.
extern "C++" {
template <bool FUSED>
static void launch_chunk(drt_scene* s, hipStream_t st, const PathCtx& pc, const Pipe& p, const double* o, const double* d, const uint8_t* valid,
                         int64_t n, double* out_ori, double* out_dir, uint8_t* mask, int32_t* face1, int32_t* face2, int tile_w) {
    const int gs = 8 * s->n_cu;   // grid of the streaming / shading kernels
    if (tile_w < 64 || tile_w % 64 != 0 || n % (4 * (int64_t)tile_w) != 0) tile_w = 0;   // not whole 64x4 patches: linear order
    { StageTimer t(s, st, kStageCull);
      k_cull<FUSED><<<grid_for(n, kPathBlock, gs), kPathBlock, 0, st>>>(pc.tc.nodes, pc.tc.n_tris, o, d, valid, n, out_ori, out_dir, mask, face1, face2, p, tile_w); }
    { StageTimer t(s, st, kStageTrace1);
      k_trace<false><<<s->grid_path, kPathBlock, 0, st>>>(pc.tc, p.r0.ray, p.count + 0, p.r0.face, p.redo, p.count + 4, s->refill_min, s->inner_min, s->prof_stats ? s->prof_counts + kProfStages + 4 * 0 : nullptr);
      k_trace_redo<false><<<64, kTraceBlock, 0, st>>>(pc.tc, p.r0.ray, p.redo, p.count + 4, p.r0.face); }
    { StageTimer t(s, st, kStageShade1);
      k_shade1<FUSED><<<gs, kPathBlock, 0, st>>>(pc, o, d, out_ori, out_dir, mask, face1, face2, p); }
    { StageTimer t(s, st, kStageTrace2);
      k_trace<false><<<s->grid_path, kPathBlock, 0, st>>>(pc.tc, p.r1.ray, p.count + 1, p.r1.face, p.redo, p.count + 5, s->refill_min, s->inner_min, s->prof_stats ? s->prof_counts + kProfStages + 4 * 1 : nullptr);
      k_trace_redo<false><<<64, kTraceBlock, 0, st>>>(pc.tc, p.r1.ray, p.redo, p.count + 5, p.r1.face); }
    { StageTimer t(s, st, kStageShade2);
      k_shade2<FUSED><<<gs, kPathBlock, 0, st>>>(pc, o, d, out_ori, out_dir, mask, face1, face2, p); }
    { StageTimer t(s, st, kStageTrace3);
      k_trace<true><<<s->grid_path, kPathBlock, 0, st>>>(pc.tc, p.r2.ray, p.count + 2, p.r2.face, p.redo, p.count + 6, s->refill_min, s->inner_min, s->prof_stats ? s->prof_counts + kProfStages + 4 * 2 : nullptr);
      k_trace_redo<true><<<64, kTraceBlock, 0, st>>>(pc.tc, p.r2.ray, p.redo, p.count + 6, p.r2.face); }
}
}  // extern "C++"

// Fork: the internal streams wait for everything already enqueued on the caller's stream.
static int fork_streams(drt_scene* s, hipStream_t st, int streams) {
    HIP_TRY(hipEventRecord(s->fork_ev, st));
    for (int k = 0; k < streams; ++k) HIP_TRY(hipStreamWaitEvent(s->sub[k].stream, s->fork_ev, 0));
    return DRT_OK;
}
// Join: the caller's stream waits for every internal stream.
static int join_streams(drt_scene* s, hipStream_t st, int streams) {
    for (int k = 0; k < streams; ++k) {
        HIP_TRY(hipEventRecord(s->sub[k].done, s->sub[k].stream));
        HIP_TRY(hipStreamWaitEvent(st, s->sub[k].done, 0));
    }
    return DRT_OK;
}

static PathCtx sub_ctx(const drt_scene* s, const drt_scene::Sub& w, const double* d_verts, double ior_int, double ior_ext) {
    PathCtx pc = path_ctx(s, d_verts, ior_int, ior_ext);
    pc.tc.slow_stack = w.slow_stack;     // concurrent kernels must not share overflow stacks
    return pc;
}

int drt_render_forward(drt_scene_t* s, const double* d_verts, const double* d_origin, const double* d_dir, int64_t n_rays,
                       double ior_int, double ior_ext, double* d_out_ori, double* d_out_dir, uint8_t* d_mask,
                       int32_t* d_face1, int32_t* d_face2, int32_t* d_valid_idx, int64_t* d_n_valid, int tile_w, void* stream) {
    CHECK_BUILT(s);
    if (n_rays < 0 || n_rays > INT32_MAX) return fail(DRT_E_INVALID, "ray count out of range");
    hipStream_t st = (hipStream_t)stream;
    if (n_rays == 0) {
        if (d_n_valid) HIP_TRY(hipMemsetAsync(d_n_valid, 0, sizeof(int64_t), st));
        return DRT_OK;
    }
    if (!d_verts || !d_origin || !d_dir || !d_out_ori || !d_out_dir || !d_mask || !d_face1 || !d_face2) return fail(DRT_E_INVALID, "null pointer argument");
    if ((d_valid_idx == nullptr) != (d_n_valid == nullptr)) return fail(DRT_E_INVALID, "d_valid_idx and d_n_valid go together");
    const Plan pl = plan_call(s, n_rays, tile_w);
    for (int k = 0; k < pl.streams; ++k) { int rc = ensure_queues(s->sub[k], pl.size, false); if (rc) return rc; }
    HIP_TRY(hipMemsetAsync(s->vcount, 0, sizeof(unsigned), st));
    int rc = fork_streams(s, st, pl.streams);
    if (rc) return rc;
    for (int j = 0; j < pl.count; ++j) {
        drt_scene::Sub& w = s->sub[j % pl.streams];
        const int64_t b = j * pl.size;
        const int64_t n = n_rays - b < pl.size ? n_rays - b : pl.size;
        const PathCtx pc = sub_ctx(s, w, d_verts, ior_int, ior_ext);
        const Pipe p = pipe_of(s, w);
        HIP_TRY(hipMemsetAsync(w.qcount, 0, 8 * sizeof(unsigned), w.stream));
        launch_chunk<false>(s, w.stream, pc, p, d_origin + 3 * b, d_dir + 3 * b, nullptr, n, d_out_ori + 3 * b, d_out_dir + 3 * b,
                            d_mask + 3 * b, d_face1 + b, d_face2 + b, tile_w);
        { StageTimer t(s, w.stream, kStageFinish);
          k_finish<<<8 * s->n_cu, kPathBlock, 0, w.stream>>>(d_out_ori + 3 * b, d_out_dir + 3 * b, d_mask + 3 * b, d_face2 + b, p, b, d_valid_idx); }
        if (s->prof_on) k_prof_counts<<<1, 64, 0, w.stream>>>(w.qcount, (unsigned long long)n, s->prof_counts, 0);
    }
    rc = join_streams(s, st, pl.streams);
    if (rc) return rc;
    if (d_n_valid) k_store_count<<<1, 64, 0, st>>>(s->vcount, d_n_valid);
    if (s->prof_on) s->prof_stream = st;
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_render_backward(drt_scene_t* s, const double* d_verts, const double* d_origin, const double* d_dir, int64_t n_rays,
                        double ior_int, double ior_ext, const int32_t* d_face1, const int32_t* d_face2,
                        const int32_t* d_valid_idx, const int64_t* d_n_valid,
                        const double* d_grad_out_ori, const double* d_grad_out_dir, double* d_grad_verts, void* stream) {
    CHECK_BUILT(s);
    if (n_rays < 0 || n_rays > INT32_MAX) return fail(DRT_E_INVALID, "ray count out of range");
    if (n_rays == 0 || (!d_grad_out_ori && !d_grad_out_dir)) return DRT_OK;
    if (!d_verts || !d_origin || !d_dir || !d_face1 || !d_face2 || !d_grad_verts) return fail(DRT_E_INVALID, "null pointer argument");
    if ((d_valid_idx == nullptr) != (d_n_valid == nullptr)) return fail(DRT_E_INVALID, "d_valid_idx and d_n_valid go together");
    hipStream_t st = (hipStream_t)stream;
    const PathCtx pc = path_ctx(s, d_verts, ior_int, ior_ext);
    if (d_valid_idx) {   // the forward's list of completed paths: no pass over the dense arrays at all
        StageTimer t(s, st, kStageBackward);
        k_render_bwd<<<2 * s->n_cu, 256, 0, st>>>(pc, d_origin, d_dir, d_face1, d_face2, d_grad_out_ori, d_grad_out_dir, d_grad_verts,
                                                   d_valid_idx, nullptr, d_n_valid);
    } else {             // no list saved: compact face2 >= 0 first (on the caller's stream, workspace of sub-stream 0)
        drt_scene::Sub& w = s->sub[0];
        const int64_t chunk = n_rays < s->chunk_rays ? n_rays : s->chunk_rays;
        int rc = ensure_queues(w, chunk, false);
        if (rc) return rc;
        for (int64_t b = 0; b < n_rays; b += chunk) {
            const int64_t n = n_rays - b < chunk ? n_rays - b : chunk;
            HIP_TRY(hipMemsetAsync(s->vcount, 0, sizeof(unsigned), st));
            { StageTimer t(s, st, kStageCollect);
              k_collect_valid<<<grid_for(n, kPathBlock, 8 * s->n_cu), kPathBlock, 0, st>>>(d_face2 + b, n, b, w.q_idx[0], s->vcount); }
            { StageTimer t(s, st, kStageBackward);
              k_render_bwd<<<2 * s->n_cu, 256, 0, st>>>(pc, d_origin, d_dir, d_face1, d_face2, d_grad_out_ori, d_grad_out_dir, d_grad_verts,
                                                         w.q_idx[0], s->vcount, nullptr); }
            if (s->prof_on) k_prof_counts_bwd<<<1, 64, 0, st>>>(s->vcount, (unsigned long long)n, s->prof_counts);
        }
    }
    if (s->prof_on) s->prof_stream = st;
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_ray_loss(const double* d_out_ori, const double* d_out_dir, const uint8_t* d_mask, const double* d_screen_pixel,
                 const uint8_t* d_valid, int64_t n_rays, double* d_loss, double* d_grad_out_dir,
                 int32_t* d_list, uint32_t* d_n_list, void* stream) {
    if (n_rays < 0 || n_rays > INT32_MAX) return fail(DRT_E_INVALID, "ray count out of range");
    if (n_rays == 0) return DRT_OK;
    if (!d_out_ori || !d_out_dir || !d_mask || !d_screen_pixel || !d_valid || !d_loss) return fail(DRT_E_INVALID, "null pointer argument");
    if ((d_list == nullptr) != (d_n_list == nullptr)) return fail(DRT_E_INVALID, "d_list and d_n_list go together");
    k_ray_loss<<<grid_for((n_rays + kLossRays - 1) / kLossRays, kPathBlock, 4096), kPathBlock, 0, (hipStream_t)stream>>>(d_out_ori, d_out_dir, d_mask, d_screen_pixel, d_valid, n_rays,
                                                                                          d_loss, d_grad_out_dir, d_list, d_n_list);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_scale_rows3(double* d_x, const int32_t* d_list, const uint32_t* d_n_list, const double* d_scale, void* stream) {
    if (!d_x || !d_list || !d_n_list || !d_scale) return fail(DRT_E_INVALID, "null pointer argument");
    k_scale_rows3<<<1024, 256, 0, (hipStream_t)stream>>>(d_x, d_list, d_n_list, d_scale);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_render_ray_loss_fused(drt_scene_t* s, const double* d_verts, const double* d_origin, const double* d_dir,
                              const double* d_screen_pixel, const uint8_t* d_valid, int64_t n_rays, double ior_int,
                              double ior_ext, double* d_loss, double* d_grad_verts, int64_t* d_n_valid, int tile_w, void* stream) {
    CHECK_BUILT(s);
    if (n_rays < 0 || n_rays > INT32_MAX) return fail(DRT_E_INVALID, "ray count out of range");
    if (n_rays == 0) return DRT_OK;
    if (!d_verts || !d_origin || !d_dir || !d_screen_pixel || !d_valid || !d_loss || !d_grad_verts) return fail(DRT_E_INVALID, "null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    const Plan pl = plan_call(s, n_rays, tile_w);
    for (int k = 0; k < pl.streams; ++k) { int rc = ensure_queues(s->sub[k], pl.size, true); if (rc) return rc; }
    int rc = fork_streams(s, st, pl.streams);
    if (rc) return rc;
    for (int j = 0; j < pl.count; ++j) {
        drt_scene::Sub& w = s->sub[j % pl.streams];
        const int64_t b = j * pl.size;
        const int64_t n = n_rays - b < pl.size ? n_rays - b : pl.size;
        const PathCtx pc = sub_ctx(s, w, d_verts, ior_int, ior_ext);
        const Pipe p = pipe_of(s, w);
        HIP_TRY(hipMemsetAsync(w.qcount, 0, 8 * sizeof(unsigned), w.stream));
        launch_chunk<true>(s, w.stream, pc, p, d_origin + 3 * b, d_dir + 3 * b, d_valid + b, n, nullptr, nullptr, nullptr, w.tmp_face1, w.tmp_face2, tile_w);
        { StageTimer t(s, w.stream, kStageLossBwdFused);
          k_loss_bwd_fused<<<2 * s->n_cu, 256, 0, w.stream>>>(pc, d_origin + 3 * b, d_dir + 3 * b, d_screen_pixel + 3 * b, w.tmp_face1, w.tmp_face2, p,
                                                               d_loss, d_grad_verts, reinterpret_cast<unsigned long long*>(d_n_valid)); }
        if (s->prof_on) k_prof_counts<<<1, 64, 0, w.stream>>>(w.qcount, (unsigned long long)n, s->prof_counts, 1);
    }
    rc = join_streams(s, st, pl.streams);
    if (rc) return rc;
    if (s->prof_on) s->prof_stream = st;
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_profile_enable(drt_scene_t* s, int on) {
    CHECK_SCENE(s);
    if (on && s->prof_ev.empty()) {
        s->prof_ev.resize(8192);
        s->prof_stage.resize(4096);
        for (auto& e : s->prof_ev) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipMalloc(&s->prof_counts, sizeof(unsigned long long) * (kProfStages + 12)));
        HIP_TRY(hipMemset(s->prof_counts, 0, sizeof(unsigned long long) * (kProfStages + 12)));
    }
    s->prof_on = on != 0;
    s->prof_stats = on == 2;
    s->prof_serial = on == 3;
    return DRT_OK;
}

int drt_profile_read(drt_scene_t* s, double* ms_out, int64_t* launches_out, int64_t* items_out) {
    CHECK_SCENE(s);
    if (!ms_out || !launches_out || !items_out) return fail(DRT_E_INVALID, "null pointer argument");
    for (int k = 0; k < kProfStages; ++k) { ms_out[k] = 0.0; launches_out[k] = 0; items_out[k] = 0; }
    if (s->prof_ev.empty()) return DRT_OK;
    HIP_TRY(hipStreamSynchronize(s->prof_stream));
    for (size_t k = 0; k + 1 < s->prof_used; k += 2) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s->prof_ev[k], s->prof_ev[k + 1]));
        const int stg = s->prof_stage[k / 2];
        ms_out[stg] += ms;
        launches_out[stg] += 1;
    }
    unsigned long long h[kProfStages + 12];
    HIP_TRY(hipMemcpy(h, s->prof_counts, sizeof(h), hipMemcpyDeviceToHost));
    for (int k = 0; k < kProfStages; ++k) items_out[k] = (int64_t)h[k];
    for (int k = 0; k < 12; ++k) s->trace_stats[k] = (int64_t)h[kProfStages + k];
    HIP_TRY(hipMemset(s->prof_counts, 0, sizeof(h)));
    s->prof_used = 0;
    return DRT_OK;
}

int drt_profile_trace_stats(drt_scene_t* s, int64_t* out12) {
    CHECK_SCENE(s);
    if (!out12) return fail(DRT_E_INVALID, "null pointer argument");
    for (int k = 0; k < 12; ++k) out12[k] = s->trace_stats[k];
    return DRT_OK;
}

int drt_dihedral_forward(const double* d_verts, const int64_t* d_e2f, int64_t n_edges, double* d_cos, void* stream) {
    if (n_edges < 0) return fail(DRT_E_INVALID, "negative edge count");
    if (n_edges == 0) return DRT_OK;
    if (!d_verts || !d_e2f || !d_cos) return fail(DRT_E_INVALID, "null pointer argument");
    k_dihedral_fwd<<<(unsigned)((n_edges + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_verts, d_e2f, n_edges, d_cos);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_dihedral_backward(const double* d_verts, const int64_t* d_e2f, int64_t n_edges, const double* d_grad_cos,
                          double* d_grad_verts, void* stream) {
    if (n_edges < 0) return fail(DRT_E_INVALID, "negative edge count");
    if (n_edges == 0) return DRT_OK;
    if (!d_verts || !d_e2f || !d_grad_cos || !d_grad_verts) return fail(DRT_E_INVALID, "null pointer argument");
    k_dihedral_bwd<0><<<(unsigned)((n_edges + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_verts, d_e2f, n_edges, d_grad_cos, nullptr, d_grad_verts);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_sm_loss_fused(const double* d_verts, const int64_t* d_e2f, int64_t n_edges, double* d_loss, double* d_grad_verts, void* stream) {
    if (n_edges < 0) return fail(DRT_E_INVALID, "negative edge count");
    if (n_edges == 0) return DRT_OK;
    if (!d_verts || !d_e2f || !d_loss || !d_grad_verts) return fail(DRT_E_INVALID, "null pointer argument");
    k_dihedral_bwd<1><<<(unsigned)((n_edges + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_verts, d_e2f, n_edges, nullptr, d_loss, d_grad_verts);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_silhouette_flags(const double* d_verts, const int64_t* d_e2f, int64_t n_edges, const double* d_origin3, uint8_t* d_flags, void* stream) {
    if (n_edges < 0) return fail(DRT_E_INVALID, "negative edge count");
    if (n_edges == 0) return DRT_OK;
    if (!d_verts || !d_e2f || !d_origin3 || !d_flags) return fail(DRT_E_INVALID, "null pointer argument");
    k_silhouette_flags<<<(unsigned)((n_edges + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_verts, d_e2f, n_edges, d_origin3, d_flags);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_edge_sample_forward(drt_scene_t* s, const double* d_verts, const int64_t* d_edges, int64_t n_edges, const double* d_camera,
                            const double* d_origin3, int64_t* d_index, float* d_f, void* stream) {
    CHECK_BUILT(s);
    if (n_edges < 0) return fail(DRT_E_INVALID, "negative edge count");
    if (n_edges == 0) return DRT_OK;
    if (!d_verts || !d_edges || !d_camera || !d_origin3 || !d_index || !d_f) return fail(DRT_E_INVALID, "null pointer argument");
    k_edge_sample_fwd<<<grid_for(n_edges, kTraceBlock, s->grid_trace), kTraceBlock, 0, (hipStream_t)stream>>>(
        trace_ctx(s), d_verts, d_edges, n_edges, reinterpret_cast<const Camera*>(d_camera), d_origin3, d_index, d_f);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_edge_sample_backward(const double* d_verts, const int64_t* d_edges, int64_t n_edges, const double* d_camera, const float* d_f,
                             const double* d_coef, int detach_depth, double* d_grad_verts, void* stream) {
    if (n_edges < 0) return fail(DRT_E_INVALID, "negative edge count");
    if (n_edges == 0) return DRT_OK;
    if (!d_verts || !d_edges || !d_camera || !d_f || !d_coef || !d_grad_verts) return fail(DRT_E_INVALID, "null pointer argument");
    k_edge_sample_bwd<<<(unsigned)((n_edges + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        d_verts, d_edges, n_edges, reinterpret_cast<const Camera*>(d_camera), d_f, d_coef, detach_depth, d_grad_verts);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_closest_point(drt_scene_t* s, const double* d_points, int64_t n, double* d_dist, int32_t* d_face, double* d_closest, void* stream) {
    CHECK_BUILT(s);
    if (n < 0) return fail(DRT_E_INVALID, "negative point count");
    if (n == 0) return DRT_OK;
    if (!d_points || !d_dist) return fail(DRT_E_INVALID, "null pointer argument");
    k_closest_point<<<grid_for(n, kTraceBlock, s->grid_trace), kTraceBlock, 0, (hipStream_t)stream>>>(trace_ctx(s), s->faces, s->verts, d_points, n, d_dist, d_face, d_closest);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_vh_loss_fused(drt_scene_t* s, const double* d_verts, const int64_t* d_edges, const int64_t* d_e2f, int64_t n_edges,
                      int n_views, const double* const* d_cameras, const double* const* d_origins, const double* const* d_soft_masks,
                      int resx, int resy, int detach_depth, double* d_loss, double* d_grad_verts, void* stream) {
    CHECK_BUILT(s);
    if (n_edges < 0 || n_views < 0 || resx <= 0 || resy <= 0 || n_edges * kVhViews > (int64_t)UINT32_MAX) return fail(DRT_E_INVALID, "bad size argument");
    if (n_edges == 0 || n_views == 0) return DRT_OK;
    if (!d_verts || !d_edges || !d_e2f || !d_cameras || !d_origins || !d_soft_masks || !d_loss || !d_grad_verts) return fail(DRT_E_INVALID, "null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t need = n_edges * std::min(kVhViews, n_views);
    if (need > s->vh_cap) {
        (void)hipFree(s->vh_list); s->vh_list = nullptr; s->vh_cap = 0;
        HIP_TRY(hipMalloc(&s->vh_list, sizeof(uint32_t) * need));
        s->vh_cap = need;
    }
    for (int v0 = 0; v0 < n_views; v0 += kVhViews) {
        const int nv = std::min(kVhViews, n_views - v0);
        VhViews vw{};
        for (int k = 0; k < nv; ++k) {
            if (!d_cameras[v0 + k] || !d_origins[v0 + k] || !d_soft_masks[v0 + k]) return fail(DRT_E_INVALID, "null pointer argument");
            vw.cam[k] = d_cameras[v0 + k]; vw.origin[k] = d_origins[v0 + k]; vw.soft[k] = d_soft_masks[v0 + k];
        }
        HIP_TRY(hipMemsetAsync(s->vcount + 1, 0, sizeof(unsigned), st));
        k_vh_cull<<<grid_for(n_edges * nv, kPathBlock, 4 * s->n_cu), kPathBlock, 0, st>>>(d_verts, d_e2f, n_edges, nv, vw, s->vh_list, s->vcount + 1);
        k_vh_fused<<<4 * s->n_cu, kTraceBlock, 0, st>>>(trace_ctx(s), d_verts, d_edges, (uint32_t)n_edges, s->vh_list, s->vcount + 1,
                                                        vw, resx, resy, detach_depth, d_loss, d_grad_verts);
    }
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

}  // extern "C"
