// drt_scene.h -- what the translation units of libdrt_hip.so share: the scene object behind drt_scene_t, launch
// constants, error plumbing, the per-stage timer.  (drt_build.hip: LBVH build + checks; drt_trace.hip: B1 queries and
// closest point; drt_pipeline.hip: the refraction pipeline, its backward and losses; drt_edges.hip: silhouette and
// smoothness branches; drt_api.hip: create / destroy / profiling.)
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared.  Wave size is 64 throughout.
#pragma once
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
#include "drt_fixed.h"
#include "drt_lbvh.h"
#include "drt_path.h"
#include "drt_raster.h"
#include "drt_shade.h"
#include "drt_traverse.h"
#include "drt_tri.h"

using namespace drt;


// ------------------------------------------------------------------------------------------
// error plumbing (defined in drt_api.hip)
// ------------------------------------------------------------------------------------------
int fail(int code, const char* fmt, ...);
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
// A traversal postpones at most three children per level of the wide tree; a wide node is rooted at a binary node and its
// children are strict binary descendants, so wide depth <= binary height, and a Karras tree over (30-bit key, 32-bit index)
// composites gains at least one prefix bit per level: height <= 64.  3 x 64 entries therefore ALWAYS suffice (drt_traverse.h
// Stack has no bound check); the part beyond the LDS entries lives in a per-thread global area that is only touched on overflow.
static_assert(kStackFast >= 3, "FastStack keeps four spare entries (kStackFast + 1 rows) above its usable depth");
constexpr int kStackTotal = 192;
constexpr int kStackSlowDev = kStackTotal - kStackFast;   // global overflow entries per thread
constexpr int kQCount = 24;            // words of a pipeline's counter block ([16]: k_trace's retired-workgroup counter)
constexpr int kRedoGrid = 64;          // blocks of the one-thread-per-item second-pass kernels (k_path_redo, k_gen_late); sizes the overflow areas
constexpr int kTraceGridMax = 4096;    // blocks per traversal launch (persistent, grid-stride)
constexpr int64_t kChunkRays = 1 << 26; // max rays per pipeline pass; bounds the list workspace (96 B per ray of the largest pass)

constexpr int kSortBlock = 256, kSortItems = 8, kSortTile = kSortBlock * kSortItems, kRadix = 256;

// stage ids of drt_profile_read
enum { kStageBuild = 0, kStageCull, kStageTrace1, kStageShade1, kStageTrace2, kStageShade2, kStageTrace3, kStageFinish, kStageCollect, kStageBackward, kStageLossBwdFused, kStageRaster, kStageFill, kStagePath, kProfStages };
static_assert(kProfStages == DRT_PROFILE_STAGES, "include/drt_hip.h");

struct BuildParams {   // written by k_bounds, read by the later build kernels
    float lox, loy, loz;
    float ix, iy, iz;   // 1 / extent per axis (0 extent -> 0)
    float pad;
    int32_t reserved;
    MortonPlan plan;    // axis of every key bit (drt_lbvh.h)
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
    TriRec* tris_flat = nullptr;   // [F] face order (projected primary visibility: needs no tree)
    int32_t* slot_of_face = nullptr;   // [F] face id -> slot of its record in `tris` (written by k_refit of every build / refit)
    // drt_render_seed: per camera ray, the face id the refracted ray of that pixel hit in an earlier call (TraceSeed); one shot, consumed by
    // the next drt_render_forward / drt_render_ray_loss_fused of exactly `n` rays
    struct Seed { int32_t* face2 = nullptr; int64_t n = 0; } seed;
    hipStream_t build_stream = nullptr;   // the LBVH build runs here, beside the caller's next fills / projection pass
    hipEvent_t build_fork = nullptr, build_done = nullptr;
    bool build_pending = false;    // a build was enqueued on build_stream: consumers of the tree wait for build_done
    bool order_valid = false;      // idx[sorted_buf] holds the Morton order of the last build over the CURRENT faces (k_tri_flat's order)
    int sorted_buf = 0;            // which of keys[] / idx[] the last sort ended in (three radix passes end in 1, four in 0)
    // DRT_TREE (drt_tree_mode): 0 = the LBVH is rebuilt from the Morton sort on every update (the default: BASELINE's "LBVH rebuilt each
    // iteration"); 1 = its TOPOLOGY is kept for `rebuild_every` updates, the updates in between only refit the boxes and re-quantise the
    // wide nodes; 2 = a binned-SAH topology built on the host at every update_mesh, refit-only on every update_vert.
    bool cull_direct = true;       // DRT_CULL_DIRECT=0: bounce #1 as a pass of its own (k_shade1) over list R0, as before round 4
    bool cull_park = true;         // DRT_CULL_PARK=0: k_shade2 recomputes bounce #1 instead of reading the parked refracted ray
    int64_t cull_direct_min_rays = (int64_t)1 << 25;   // sub-batches below this keep the separate pass (DRT_CULL_DIRECT_MIN_LOG2)
    int tree_mode = 0, rebuild_every = 1, since_full = 0;
    bool topology_fixed = false;   // mode 2: the host's topology is installed for the current faces
    uint32_t* bounds_acc = nullptr;  // [6] scene-box accumulators of drt_update_vert_f64's cast kernel (order-preserving uint encodings; put back to +-inf by the build that read them)
    bool async_build = true;       // DRT_ASYNC_BUILD=0: build on the caller's stream
    uint32_t *keys[2] = {nullptr, nullptr}, *idx[2] = {nullptr, nullptr};
    uint32_t* hist = nullptr;      // [kRadix * tiles]
    int32_t *parent_inner = nullptr, *parent_leaf = nullptr;
    uint32_t* flags = nullptr;
    BuildParams* params = nullptr;
    int32_t* slow_stack = nullptr; // [max(grid_trace, 4 n_cu, kRedoGrid) * kTraceBlock * kStackSlowDev] (B1 queries, closest point, edge probes): ensure_slow_stack
    unsigned long long* scratch = nullptr;  // small counters
    int32_t *b1_list = nullptr, *b1_redo = nullptr;   // B1 queries (drt_intersect*): candidate ray numbers, redo list
    unsigned* b1_count = nullptr;           // [0] candidates, [1] redo entries, [2] retired workgroups of k_trace (all zero between queries)
    int64_t b1_cap = 0;
    // wavefront-pipeline workspace, sized for one chunk of rays, allocated on first use
    // Pipeline workspaces: one per internal stream.  A call is cut into sub-batches that run on
    // different HIP streams, so that the HBM-bound k_cull of one sub-batch overlaps the latency-bound
    // k_trace of another and the tail of one kernel is filled by the next sub-batch's work.
    struct Sub {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        hipEvent_t fill_fork = nullptr, fill_join = nullptr; // around the dense-output memsets of a DRT_GRID_TRUST call, issued beside the traversal
        int32_t* q_idx[3] = {nullptr, nullptr, nullptr};     // ray lists R0..R2: index,
        float* q_ray[3] = {nullptr, nullptr, nullptr};       //   float32 ray [cap,6],
        int32_t* q_face[3] = {nullptr, nullptr, nullptr};    //   traversal result
        int32_t *tmp_face1 = nullptr, *tmp_face2 = nullptr;  // fused path keeps face ids here; backward fallback list
        unsigned* qcount = nullptr;                          // [kQCount] list sizes + redo counts of the sub-batch in flight, [8..15]: k_path's cursors
        double* ray64 = nullptr;                             // fused one-kernel path: [2][cap,3] float64 refracted rays between bounce #1 and #2
        int64_t ray64_cap = 0;
        int32_t* redo = nullptr;                             // [cap] rays for the second pass of k_trace (its last workgroup's epilogue)
        int32_t* slow_stack = nullptr;                       // [kRedoGrid * kTraceBlock * kStackSlowDev] overflow area of this stream's second passes
        int64_t q_cap = 0, fused_cap = 0;
        // projected primary visibility (drt_raster.h): per-ray keys (all-empty between calls), one bit per 64 rays that
        // says "some key here was written", the fitted image models, triangles too large for one lane
        unsigned long long* zbuf = nullptr;
        uint32_t* zmask = nullptr;
        int64_t z_cap = 0;
        ViewModel* vmodel = nullptr;
        int vm_cap = 0;
        void* big = nullptr;
        unsigned* big_count = nullptr;
        int32_t* gen_list = nullptr;                         // [q_cap] R0 slots whose ray did not verify as a grid ray: traced like before
    };
    static constexpr unsigned kBigCap = 1u << 20;
    static constexpr int kMaxSub = 4;
    Sub sub[kMaxSub];
    int n_sub = 2;                 // internal streams in use
    int sub_per_stream = 1;        // sub-batches dealt to each stream (when the call is large enough)
    int64_t min_sub_rays = 1 << 24;   // do not cut a call into sub-batches smaller than this
    hipEvent_t fork_ev = nullptr;
    // drt_prefill_zero: dense outputs of the NEXT drt_render_forward zeroed ahead of time on the build stream (idle after a forward's
    // join, i.e. during the caller's loss / backward / optimiser tail); a forward whose out_ori / out_dir / mask IS such a buffer skips that fill
    // drt_outputs_clean: the rows to zero, launched by the next drt_render_forward on the caller's stream BEHIND its fork (idle until the join)
    struct Clean { double* ori = nullptr; double* dir = nullptr; uint8_t* mask = nullptr; int64_t n = 0; const int32_t* rows = nullptr; const int64_t* n_rows = nullptr;
                   bool captured = false; /* requested while a graph was being captured: consumed by the render call of the same capture */ } clean;
    // The list of completed paths of the last drt_render_forward, by sub-batch: sub-batch j appends into the caller's list at its own
    // offset (its first ray's index: a sub-batch cannot complete more paths than it has rays) under its own counter seg_counts[j]; behind
    // the join k_join_lists closes the gaps.  Segment 0 is never moved -- what lets drt_ray_loss_listed_grad_split start on it while the
    // other pipeline is still tracing.
    static constexpr int kMaxSeg = 32;
    unsigned* seg_counts = nullptr;            // device [kMaxSeg]
    struct Segs { const int32_t* list = nullptr; int n = 0; int stream0 = 0; } segs;
    struct Prefill { const void* ptr = nullptr; int64_t bytes = 0; };
    Prefill prefill[3];
    int n_prefill = 0;
    hipEvent_t prefill_fork = nullptr, prefill_done = nullptr, prefill_done_cap = nullptr;   // (_cap: stands in for prefill_done inside a captured render call)
    unsigned* vcount = nullptr;    // [0] valid rays of the whole call, [1] silhouette items of drt_vh_loss_fused, [3] canary counter bumped by every k_check_views launch
    uint32_t* vh_list = nullptr;   // (view, edge) items of drt_vh_loss_fused: its own buffer, so that the call may run on
    int64_t vh_cap = 0;            //   another stream than a pipeline call (which owns the Sub workspaces)
    // optional per-stage timing (drt_profile_*): hipEvent pairs on the launch stream
    bool prof_on = false;
    uint32_t prof_mask = ~0u;                 // stages that are timed while prof_on (drt_profile_select)
    bool prof_stats = false;                  // level 2: k_trace also accumulates visit statistics (adds contended atomics)
    bool prof_serial = false;                 // level 3: sub-batches run on ONE internal stream, so that each kernel is timed alone
    std::vector<hipEvent_t> prof_ev;          // pool, used pairwise
    std::vector<int> prof_stage;              // stage id of pair k
    size_t prof_used = 0;                     // events handed out since the last read
    size_t prof_dropped = 0;                  // stage timings lost since the last read (event pool could not grow): drt_profile_read fails
    unsigned long long* prof_counts = nullptr;  // device [kProfStages]: queue sizes accumulated per stage
    hipStream_t prof_stream = nullptr;
    int n_cu = 256;
    int grid_trace = 2048;         // resident blocks of the pure-traversal kernels
    int grid_path = 2048;          // resident 256-thread blocks of k_trace
    int64_t trace_stats[12] = {0};  // per k_trace stage: wave-steps, lane-steps, refills, max wave-steps (last profile read)
    int grid_mega = 1024;          // resident 256-thread blocks of k_path
    int64_t mega_max_rays = 0;     // sub-batches of at most this many camera rays run their back half as ONE kernel (k_path); 0: never (DRT_MEGA_MAX_LOG2; measured: no gain)
    int shade_min = 8;             // k_path enters a float64 stage once this many lanes wait for one
    int mega_refill_min = 16;      // k_path hands new rays to a wave once this many lanes are idle
    int refill_min = 32;           // k_trace refills a wave once this many lanes are idle
    int inner_min = 24;            // k_trace leaves the inner phase once fewer lanes than this are at inner nodes
    int64_t chunk_rays = kChunkRays;

    bool fill_after_shade1 = true; // the late fills start behind k_shade1 (beside the VALU-bound second traversal only) instead of behind the cull stage: the
                                   // latency-bound first shading then has the memory system to itself (0.19 -> 0.09 ms per launch, step -1 %); DRT_FILL_AFTER_SHADE1=0
    bool fill_overlap = true;      // DRT_FILL_OVERLAP=0: the dense-output memsets of a DRT_GRID_TRUST call stay in front of the projection pass
    bool grid_canary = true;       // DRT_GRID_CANARY=0: a trusted image is re-checked on its fixed 8x8 lattice only (k_check_views)
    unsigned canary_salt = 0;      // changes with every trusted projection pass
    bool seed_tiled = true;        // DRT_SEED_TILED=0: the seed buffer of a whole-image call is indexed by ray number instead of by 4x4-pixel tile
    bool hit_seed = true;          // DRT_HIT_SEED=0: drt_render_seed's seeds are ignored (A/B measurement)
    bool use_raster = true;        // DRT_RASTER=0: every primary ray takes the BVH path (A/B measurement)
    bool built = false;
};


// occupancy of the persistent kernels (defined next to them: drt_trace.hip, drt_pipeline.hip)
int query_blocks_per_cu();
int pipeline_blocks_per_cu();
int mega_blocks_per_cu();

// defined in drt_raster.hip
int ensure_raster(drt_scene* s, drt_scene::Sub& w, int64_t n_rays, int n_views, hipStream_t st);
int launch_raster(drt_scene* s, drt_scene::Sub& w, hipStream_t st, const double* d_origin, const double* d_dir, int n_views, int iw, int ih,
                  ViewModel* trusted);

// defined in drt_api.hip
int ensure_slow_stack(drt_scene* s, hipStream_t st);
bool det_mode();        // drt_deterministic / DRT_DETERMINISTIC: gradient and loss targets are FxCell accumulators (drt_fixed.h)
// a kernel template whose first parameter is the accumulation mode
#define DET_LAUNCH(kern, grid, block, st, ...)                                        \
    do {                                                                              \
        if (det_mode()) kern<true><<<grid, block, 0, st>>>(__VA_ARGS__);              \
        else kern<false><<<grid, block, 0, st>>>(__VA_ARGS__);                        \
    } while (0)
// -DDRT_CHECK=1: the stack-invariant counters of the two translation units that instantiate the traversal kernels
int check_counters_pipeline(unsigned long long* out4);
int check_counters_trace(unsigned long long* out4);

// defined in drt_build.hip
void scene_free_mesh(drt_scene* s);
int ensure_capacity(drt_scene* s, int64_t n_faces, int64_t n_verts);
int rebuild(drt_scene* s, hipStream_t st, const uint32_t* acc);
int wait_build(drt_scene* s, hipStream_t st);

inline int grid_for(int64_t n, int block, int cap) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

inline TraceCtx trace_ctx(const drt_scene* s) { return TraceCtx{s->wide, s->tris, (int)s->n_faces, s->slow_stack}; }
inline PathCtx path_ctx(const drt_scene* s, const double* d_verts, double ior_int, double ior_ext) {
    return PathCtx{trace_ctx(s), s->faces, d_verts, ior_int, ior_ext};
}

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

// RAII-ish stage timer: records an event pair around a kernel launch when profiling is on.
struct StageTimer {
    drt_scene* s; hipStream_t st; bool on;
    StageTimer(drt_scene* s_, hipStream_t st_, int stage) : s(s_), st(st_), on(false) {
        if (!s->prof_on || !((s->prof_mask >> stage) & 1u)) return;
        if (s->prof_used + 2 > s->prof_ev.size() && !grow()) { ++s->prof_dropped; return; }
        on = true;
        s->prof_stage[s->prof_used / 2] = stage;
        if (!s->prof_stream) s->prof_stream = st;
        (void)hipEventRecord(s->prof_ev[s->prof_used], st);
    }
    // the pool grows with the number of launches between two reads (a long --steps run): never silently stops recording
    bool grow() {
        constexpr size_t kMaxEvents = (size_t)1 << 22;
        const size_t want = s->prof_ev.size() ? 2 * s->prof_ev.size() : 8192;
        if (want > kMaxEvents) return false;
        const size_t old = s->prof_ev.size();
        s->prof_ev.resize(want, nullptr);
        s->prof_stage.resize(want / 2, 0);
        for (size_t k = old; k < want; ++k)
            if (hipEventCreate(&s->prof_ev[k]) != hipSuccess) {
                for (size_t j = old; j < k; ++j) (void)hipEventDestroy(s->prof_ev[j]);
                s->prof_ev.resize(old); s->prof_stage.resize(old / 2);
                return false;
            }
        return true;
    }
    ~StageTimer() {
        if (!on) return;
        (void)hipEventRecord(s->prof_ev[s->prof_used + 1], st);
        s->prof_used += 2;
    }
};
