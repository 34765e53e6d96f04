// drt_trace_kernel.h -- the persistent BVH traversal kernel, shared by the refraction pipeline (drt_pipeline.hip:
// compact ray lists R0..R2 -> face per list slot) and by boundary B1 (drt_trace.hip: optix_mesh::intersect of
// reference optix_extend.cpp:29-57 -> T, ID per ray; the any-hit form -> hit flag per ray).
#pragma once
#include "drt_device.h"

// -DDRT_CHECK=1: violation counters of this translation unit's traversal kernels (see drt_traverse.h), [kCheck*]
#if defined(DRT_CHECK)
static __device__ unsigned g_drt_check[4];
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline void drt::drt_check_fail(int what) { atomicAdd(&g_drt_check[what & 3], 1u); }
#endif
static inline int read_check_counters(unsigned long long* out4) {      // host: adds this unit's counters to out4
    unsigned h[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_drt_check), sizeof(h)) != hipSuccess) return DRT_E_HIP;
    for (int k = 0; k < 4; ++k) out4[k] += h[k];
    return DRT_OK;
}
constexpr int kGuardRows = 2;
constexpr int32_t kGuardPoison = 0x5A5A5A5A;
#else
constexpr int kGuardRows = 0;
#endif
#ifndef DRT_LEAF_PARK
#define DRT_LEAF_PARK 1
#endif
constexpr int32_t kDrained = INT32_MIN + 2;       // TravState::cur of a ray whose stack is empty and whose PARKED leaf is all that is left (DRT_LEAF_PARK)
constexpr int32_t kFinished = INT32_MIN + 1;      // TravState::cur of a ray that is done and waits to be emitted (never a node: a leaf reference this large has no slot)

// Where a finished ray's result goes.  Pipeline: out.face[list slot].  B1 (`idx` non-null): the list holds ray numbers;
// the float32 ray is read from rays[idx[slot]] and T / ID (closest) or the hit flag (any) are written at that ray number.
struct TraceOut {
    int32_t* face;          // pipeline: [list size];  B1 closest: ID [N]
    float* t;               // B1 closest: T [N] (miss = -1, like the brute force of oracle/tracer.c)
    uint8_t* flag;          // B1 any-hit: [N]
    const int32_t* idx;     // B1: list slot -> ray number
};

// MODE 0: pipeline list (face per list slot); 1: B1 (T / ID or hit flag per ray number); 2: pipeline, but only the R0
// slots listed in `idx` (the rays the projected primary-visibility pass did not answer): face per R0 slot.
template <bool ANY, int MODE>
__device__ __forceinline__ void trace_emit(const TraceOut& out, int32_t slot, float best_t, int32_t best_face) {
    if (MODE == 0) { out.face[slot] = best_face; return; }
    const int32_t i = out.idx[slot];
    if (MODE == 2) { out.face[i] = best_face; return; }
    if (ANY) { out.flag[i] = best_face >= 0 ? 1 : 0; return; }
    out.t[i] = best_face >= 0 ? best_t : -1.0f;
    out.face[i] = best_face;
}
// Set bits of a wave mask below this lane (v_mbcnt: no (1 << lane) - 1 held in two registers through the kernel -- the traversal loop runs
// at 64 of 64 registers, and loop-invariant values are what the compiler spills first).
__device__ __forceinline__ unsigned lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Temporal hit seeds (pipeline, traversal of the refracted rays; reference DiffRender.py:542 -- the second Dintersect of trace2).  The
// caller is an optimisation loop: between two steps the vertices move by at most lr x clamp (reference optim.py:155-171), so the
// triangle a pixel's refracted ray left the object through last step is almost always the one it leaves through now.  `store` holds, per
// camera ray of the call, the face id the previous call on these rays found (-1: none; anything else out of range is ignored); a lane that
// takes a new ray tests THAT triangle first -- its record of the current build, through `slot_of_face` -- and starts its traversal with
// best_t / best_face / best_slot of that hit instead of +inf: every box behind it is culled from the first visit on.  The result is the
// same bit for bit whatever the seed is: the seed is one of the tree's own triangles, tested by the same expression on the same record as
// its leaf would test it, so the minimum over (t, face) of the candidates is the minimum the unseeded traversal finds; boxes are entered on
// ties (drt_traverse.h), and a winner that fails the deferred hit-point condition goes to the exact second pass as before.  A stale or
// wrong seed only weakens the bound.  The ray's result is written back for the next call.
struct TraceSeed {
    const int32_t* list_idx;        // list slot -> camera-ray index within the sub-batch (RayList::idx)
    int32_t* store;                 // [rays of the sub-batch] face id per camera ray, read at refill, written at emit
    const int32_t* slot_of_face;    // [n_tris] face id -> slot of its record in TraceCtx::tris (k_refit)
    unsigned tile_w;                // > 0: the sub-batch is whole images `tile_w` wide and `store` is laid out in 4 x 4-pixel tiles like the key
                                    // buffer (drt_raster.h raster_slot): the 16 x 4-pixel run of rays a wave picks up from a list in tile order then
                                    // reads and writes four adjacent 64-byte tiles instead of four pieces of four image rows (a quarter of the lines)
};
__device__ __forceinline__ int64_t seed_index(const TraceSeed& sd, int32_t ray) {
    if (sd.tile_w == 0) return ray;
    const unsigned y = (unsigned)ray / sd.tile_w, x = (unsigned)ray - y * sd.tile_w;
    return raster_slot(x, y, sd.tile_w);
}

// instantiated in drt_trace.hip (see there); `sd` only for kTraceClosestSeeded
enum : int { kTraceClosest = 0, kTraceClosestListed = 1, kTraceClosestSeeded = 2, kTraceAny = 3 };
void launch_trace_list(int variant, int grid, hipStream_t st, TraceCtx c, const float* rays, const unsigned* n_ptr, TraceOut out, int32_t* redo_list,
                       unsigned* redo_count, unsigned* done_count, int refill_min, int inner_min, unsigned long long* stats,
                       TraceSeed sd = TraceSeed{nullptr, nullptr, nullptr, 0u});

template <int MODE>
__device__ __forceinline__ const float* trace_ray(const float* __restrict__ rays, const TraceOut& out, unsigned slot) {
    return rays + 6 * (int64_t)(MODE != 0 ? out.idx[slot] : (int32_t)slot);
}

// A ray handed to the second pass (its LDS-only stack overflowed, or its winner failed the deferred hit-point condition).  The list is
// read by ANOTHER workgroup of the same launch (the one that retires last, below), possibly on another XCD, whose L2 is not coherent with
// this one's: the entry is a relaxed agent-scope atomic store (write-through, `sc1`), drained, as in k_refit's box hand-off -- no fences.
__device__ __forceinline__ void trace_redo_push(int32_t* redo_list, unsigned* redo_count, int32_t slot) {
    __hip_atomic_store(&redo_list[atomicAdd(redo_count, 1u)], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // acknowledged before this wave can reach the epilogue's barrier (rare path: the wait costs nothing)
}

// Persistent traversal over a ray list.  Each wave owns a contiguous segment of the list; a lane
// whose ray finishes takes the segment's next ray (no atomics: the cursor is wave-uniform).
// The second pass for the rays on `redo_list` (normally none) is the epilogue of the workgroup that retires LAST (`done_count`, zero at
// launch and put back to zero by that workgroup): one thread per ray with the spilling Stack.  It used to be a launch of its own, an
// empty kernel that -- queued behind this launch while the neighbour pipeline's persistent grid holds every wave slot of the chip -- took
// 0.11 ms to get through the dispatcher in every pipeline of every step (profiles/r03_kernel_summary.txt: 112 us per call, 4 us alone).
template <bool ANY, int MODE, bool SEED = false>
__global__ void __launch_bounds__(kPathBlock, 8) k_trace(TraceCtx c, const float* __restrict__ rays, const unsigned* __restrict__ n_ptr,
                                                       TraceOut out, int32_t* __restrict__ redo_list, unsigned* redo_count, unsigned* done_count,
                                                       int refill_min, int inner_min, unsigned long long* stats, TraceSeed sd = TraceSeed{nullptr, nullptr, nullptr, 0u}) {
    static_assert(!SEED || (!ANY && MODE == 0), "seeds: closest hit over a pipeline list");
    __shared__ int32_t lds[kStackFast + 1 + kGuardRows][kPathBlock];     // 20 x 1 KB x 8 blocks = the CU's 160 KB; the top FOUR rows are FastStack's spare entries
    FastStack st;
    st.base = (drt::FastPtr)&lds[0][threadIdx.x]; st.stride = kPathBlock; st.depth = kStackFast - 3; st.reset(); st.overflow = false;
#if defined(DRT_CHECK)
    for (int g = 0; g < kGuardRows; ++g) lds[kStackFast + 1 + g][threadIdx.x] = kGuardPoison;     // (per-lane columns: no barrier needed)
#endif
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
    unsigned long long wave_steps = 0, lane_steps = 0, leaf_steps = 0;   // wave-uniform diagnostics (scalar registers)
#if defined(DRT_PROBE_VISITS)
    unsigned nvis = 0;          // (tools/ubench/reorder_probe.py: a probe build reports a ray's node visits in place of T)
#endif
    for (;;) {
        // Finished rays leave here, in ONE place between the phases (where only the ray's own state is alive) and only when the wave
        // refills anyway, or is through: the deferred hit-point condition on the winner (trav_leaf<ANY, true>) reads the winner's record
        // again -- a round trip the whole wave would wait for if it were paid whenever some lane finishes -- then the result, or, for a
        // winner that fails it, the exact second pass.
        const bool fin = slot >= 0 && s.cur == kFinished;
        const unsigned long long idle = __ballot(slot < 0 || fin);
        const bool refill = idle != 0 && taken < my_rays && ((int)__popcll(idle) >= refill_min || idle == ~0ull);
        if ((refill || idle == ~0ull) && fin) {
            if (trav_winner_ok(c.tris, s)) {
#if defined(DRT_PROBE_VISITS)
                if (MODE == 1 && !ANY && s.best_face >= 0) s.best_t = (float)nvis;
#endif
                trace_emit<ANY, MODE>(out, slot, s.best_t, s.best_face);
                if (SEED && s.best_face >= 0) sd.store[seed_index(sd, sd.list_idx[slot])] = s.best_face;
            } else {
                trace_redo_push(redo_list, redo_count, slot);
            }
            slot = -1;
        }
        if (refill) {
            if (slot < 0) {
                const unsigned j = taken + lanes_below(idle);
                const unsigned k = ((part_lo + (j >> 6) * n_waves + wave) << 6) | (j & 63u);
                if (j < my_rays && k < n) {
                    const float* e = trace_ray<MODE>(rays, out, k);
                    const f3 ro{e[0], e[1], e[2]}, rd{e[3], e[4], e[5]};
                    // (the seed's triangle test comes BEFORE trav_init: the reciprocal direction and its products are not alive yet, and the
                    // kernel stays within its 64 registers without a spill)
                    float seed_t = INFINITY;
                    int32_t seed_face = -1, seed_slot = -1;
                    if (SEED) {
                        const int32_t f = sd.store[seed_index(sd, sd.list_idx[k])];
                        if ((uint32_t)f < (uint32_t)c.n_tris) trav_seed(c.tris, ro, rd, sd.slot_of_face[f], seed_t, seed_face, seed_slot);
                    }
                    trav_init(s, st, ro, rd);
                    if (SEED) { s.best_t = seed_t; s.best_face = seed_face; s.best_slot = seed_slot; }
                    st.overflow = false;
                    slot = (int32_t)k;
#if defined(DRT_PROBE_VISITS)
                    nvis = 0;
#endif
                    lds[kStackFast][threadIdx.x] = 0;            // (nothing parked: DRT_LEAF_PARK)
                }
            }
            taken += (unsigned)__popcll(idle);
        }
        const unsigned long long busy = __ballot(slot >= 0);
        if (busy == 0) break;
#if DRT_LEAF_PARK
        // "While-while" with PARKED leaves.  A lane that reaches a leaf does not wait for the wave's next triangle step: it parks the leaf
        // (one per lane, in the spare top row of its stack column; 0 = nothing parked) and goes on with the next node of its stack; only a
        // lane that reaches a SECOND leaf, or whose stack is empty, waits.  The triangle step then tests everybody's parked leaf -- also those
        // of lanes that are still descending -- so both kinds of step run with more lanes.  The parked leaf's hit arrives a few visits late
        // as a pruning bound; what is visited in between would have been popped (and visited once) anyway, only its children are not culled
        // yet.  (The spare row is written by a node visit only when that visit overflows the stack, and then the ray is abandoned.)
        for (;;) {
            const bool at_inner = slot >= 0 && s.cur >= 0;
            const unsigned long long mi = __ballot(at_inner);
            if (mi == 0) break;
            if ((int)__popcll(mi) < inner_min && __ballot(slot >= 0 && s.cur < 0 && s.cur != kFinished) != 0) break;
            ++wave_steps;
            lane_steps += (unsigned long long)__popcll(mi);
            if (at_inner) {
#if defined(DRT_PROBE_VISITS)
                ++nvis;
#endif
                const bool done = trav_inner<ANY>(c.nodes, s, st);
                if (st.overflow) {              // LDS stack exhausted (rare): hand the ray to the second pass (epilogue)
                    trace_redo_push(redo_list, redo_count, slot);
                    slot = -1;
                } else if (done) {
                    s.cur = lds[kStackFast][threadIdx.x] != 0 ? kDrained : kFinished;
                } else if (s.cur < 0 && lds[kStackFast][threadIdx.x] == 0) {
                    lds[kStackFast][threadIdx.x] = s.cur;
                    if (trav_pop(s, st)) s.cur = kDrained;
                }
            }
        }
        // triangle step: every parked leaf; then the lanes that wait at a second leaf park that one and move on
        const int32_t pk = slot >= 0 ? lds[kStackFast][threadIdx.x] : 0;
        const unsigned long long mh = __ballot(pk != 0);
        if ((mh | __ballot(slot >= 0 && s.cur < 0 && s.cur != kFinished)) != 0) {
            ++wave_steps; ++leaf_steps;
            lane_steps += (unsigned long long)__popcll(mh);
            if (pk != 0) {
#if defined(DRT_PROBE_VISITS)
                ++nvis;
#endif
                lds[kStackFast][threadIdx.x] = 0;
                if (trav_leaf_test<ANY, true>(c.tris, s, pk)) s.cur = kFinished;      // (any-hit: done)
                else if (s.cur == kDrained) s.cur = kFinished;
            }
            if (slot >= 0 && s.cur < 0 && s.cur != kFinished) {       // a leaf nobody has looked at yet: park it, move on
                lds[kStackFast][threadIdx.x] = s.cur;
                if (trav_pop(s, st)) s.cur = kDrained;
            }
        }
    }
#else
        // inner phase ("while-while"): lanes at inner nodes keep descending; lanes that reached a leaf
        // wait, so that the (longer) triangle code runs once for many lanes instead of on every step
        for (;;) {
            const bool at_inner = slot >= 0 && s.cur >= 0;
            const unsigned long long mi = __ballot(at_inner);
            if (mi == 0) break;
            if ((int)__popcll(mi) < inner_min && __ballot(slot >= 0 && s.cur < 0 && s.cur != kFinished) != 0) break;
            ++wave_steps;
            lane_steps += (unsigned long long)__popcll(mi);
            if (at_inner) {
                const bool done = trav_inner<ANY>(c.nodes, s, st);
                if (st.overflow) {              // LDS stack exhausted (rare): hand the ray to the second pass (epilogue)
                    trace_redo_push(redo_list, redo_count, slot);
                    slot = -1;
                } else if (done) {
                    s.cur = kFinished;          // (emitted at the top of the loop)
                }
            }
        }
        // leaf phase
        const bool at_leaf = slot >= 0 && s.cur < 0 && s.cur != kFinished;
        const unsigned long long ml = __ballot(at_leaf);
        if (ml != 0) {
            ++wave_steps; ++leaf_steps;
            lane_steps += (unsigned long long)__popcll(ml);
            if (at_leaf && trav_leaf<ANY, true>(c.tris, s, st)) s.cur = kFinished;
        }
    }
#endif
#if defined(DRT_CHECK)
    for (int g = 0; g < kGuardRows; ++g) DRT_DEV_ASSERT(lds[kStackFast + 1 + g][threadIdx.x] == kGuardPoison, drt::kCheckGuardRow);
#endif
    if (stats && lane == 0 && wave_steps) {
        atomicAdd(stats + 0, wave_steps);
        atomicAdd(stats + 1, lane_steps);
        atomicAdd(stats + 2, leaf_steps);
        atomicMax(stats + 3, wave_steps);
    }
    // ---- epilogue: the workgroup that retires last runs the second pass.  The barrier's workgroup-scope release waits for this wave's
    // stores (its redo entries among them) to be acknowledged before thread 0 announces the workgroup; row kStackFast of the stack array
    // (a spare row of the FastStack, outside the kStackFast rows the spilling Stack uses) carries the verdict to the other waves.
    __syncthreads();
    if (threadIdx.x == 0) lds[kStackFast][0] = atomicAdd(done_count, 1u) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!lds[kStackFast][0]) return;
    const unsigned n_redo = __hip_atomic_load(redo_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n_redo) {
        Stack sst;
        sst.fast = &lds[0][threadIdx.x]; sst.stride = kPathBlock; sst.depth_fast = kStackFast; sst.sp = 0;
        sst.slow = c.slow_stack + (int64_t)threadIdx.x * kStackSlowDev;      // (the area is sized for kRedoGrid * kTraceBlock >= kPathBlock threads)
        for (unsigned k = threadIdx.x; k < n_redo; k += kPathBlock) {
            const int32_t rs = __hip_atomic_load(&redo_list[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float* e = trace_ray<MODE>(rays, out, (unsigned)rs);
            const Hit h = traverse<ANY>(c.nodes, c.tris, c.n_tris, f3{e[0], e[1], e[2]}, f3{e[3], e[4], e[5]}, sst);
            trace_emit<ANY, MODE>(out, rs, h.t, h.face);       // (a ray of the second pass keeps its old seed: rare, and any seed is a valid one)
        }
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch on this stream
        if (MODE == 1) {      // B1: the query's counters go back to zero here (every workgroup has read them), so the next query needs no memset in front
            __hip_atomic_store(const_cast<unsigned*>(n_ptr), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(redo_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
