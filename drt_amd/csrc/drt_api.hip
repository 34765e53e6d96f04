// drt_api.hip -- scene lifetime, error string, measurement entry points of the C ABI (include/drt_hip.h).
#include "drt_scene.h"

static thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// Overflow area of the one-thread-per-item queries (B1's redo pass, closest point, silhouette probes): [blocks * kTraceBlock *
// kStackSlowDev] with blocks = the largest grid that indexes it.  0.35 GB that only those queries ever touch (and only on overflow), so
// it is allocated by the first of them, not by drt_create: a scene that only renders (a ground-truth scene, most tests) never pays.
// hipMalloc synchronises the device and is illegal while a stream is being captured: a first use under capture is refused with a message
// that says what to do (one eager call of the same entry point before the capture), instead of an opaque capture error.
int ensure_slow_stack(drt_scene* s, hipStream_t st) {
    if (s->slow_stack) return DRT_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(DRT_E_INVALID, "the first silhouette / closest-point / intersect query of a scene allocates its overflow area and cannot run inside a "
                                   "stream capture: issue one such call eagerly before capturing");
    int blocks = s->grid_trace;
    if (blocks < 4 * s->n_cu) blocks = 4 * s->n_cu;
    if (blocks < kRedoGrid) blocks = kRedoGrid;
    HIP_TRY(hipMalloc(&s->slow_stack, sizeof(int32_t) * (size_t)blocks * kTraceBlock * kStackSlowDev));
    return DRT_OK;
}

// drt_deterministic: -1 = not decided yet (DRT_DETERMINISTIC read on first use)
static int g_det = -1;
bool det_mode() {
    if (g_det < 0) { const char* ev = getenv("DRT_DETERMINISTIC"); g_det = ev && atoi(ev) != 0 ? 1 : 0; }
    return g_det != 0;
}
// cells -> float64, correctly rounded (drt_fixed.h)
__global__ void __launch_bounds__(256) k_fx_finalize(const FxCell* __restrict__ cells, int64_t n, double* __restrict__ out, int accumulate) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const FxCell c = cells[i];
        const double v = fx_to_double(Fx128{c.hi, c.lo}, (uint32_t)c.flags);
        out[i] = accumulate ? out[i] + v : v;
    }
}

// dst += src, cell by cell (the sums of several calls, or of several ranks, before ONE conversion)
__global__ void __launch_bounds__(256) k_fx_add(FxCell* __restrict__ dst, const FxCell* __restrict__ src, int64_t n) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const FxCell a = dst[i], b = src[i];
        const Fx128 r = fx_add(Fx128{a.hi, a.lo}, Fx128{b.hi, b.lo});
        dst[i] = FxCell{r.hi, r.lo, a.flags | b.flags};
    }
}
// The exchange format of an all-reduce(SUM) over int64: a cell as four words whose plain per-word sums over up to 2^20 ranks cannot
// overflow -- three limbs of 43 bits (the top one signed) and the three flags as counters in 20-bit digits.  value = l0 + l1 2^43 + l2 2^86.
constexpr int kLimbBits = 43;
__global__ void __launch_bounds__(256) k_fx_to_limbs(const FxCell* __restrict__ cells, int64_t n, int64_t* __restrict__ limbs) {
    constexpr uint64_t kMask = (1ull << kLimbBits) - 1ull;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const FxCell c = cells[i];
        const uint64_t hi = (uint64_t)c.hi;
        limbs[4 * i + 0] = (int64_t)(c.lo & kMask);
        limbs[4 * i + 1] = (int64_t)(((c.lo >> kLimbBits) | (hi << (64 - kLimbBits))) & kMask);
        limbs[4 * i + 2] = c.hi >> (2 * kLimbBits - 64);                      // arithmetic shift: the sign lives here
        limbs[4 * i + 3] = (int64_t)(((c.flags & kFxNaN) ? 1ull : 0ull) | ((c.flags & kFxPosInf) ? 1ull << 20 : 0ull) | ((c.flags & kFxNegInf) ? 1ull << 40 : 0ull));
    }
}
__global__ void __launch_bounds__(256) k_fx_from_limbs(const int64_t* __restrict__ limbs, int64_t n, FxCell* __restrict__ cells) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        // (each limb is a signed 64-bit number now: sign-extend it to 128 bits, shift it into place, add)
        auto wide = [](int64_t v, int shift) {
            Fx128 r{v < 0 ? -1 : 0, (uint64_t)v};
            if (shift >= 64) { r.hi = (int64_t)((uint64_t)v << (shift - 64)); r.lo = 0; }
            else if (shift > 0) { r.hi = (int64_t)(((uint64_t)r.hi << shift) | ((uint64_t)v >> (64 - shift))); r.lo = (uint64_t)v << shift; }
            return r;
        };
        const Fx128 r = fx_add(fx_add(wide(limbs[4 * i], 0), wide(limbs[4 * i + 1], kLimbBits)), wide(limbs[4 * i + 2], 2 * kLimbBits));
        const uint64_t f = (uint64_t)limbs[4 * i + 3];
        const uint64_t flags = ((f & 0xFFFFFull) ? kFxNaN : 0u) | (((f >> 20) & 0xFFFFFull) ? kFxPosInf : 0u) | (((f >> 40) & 0xFFFFFull) ? kFxNegInf : 0u);
        cells[i] = FxCell{r.hi, r.lo, flags};
    }
}

extern "C" {

const char* drt_last_error(void) { return g_err; }
int drt_version(void) { return 2; }

int drt_deterministic(int on) {
    const int was = det_mode() ? 1 : 0;
    if (on >= 0) g_det = on != 0;
    return was;
}
int drt_fx_add(void* d_dst_cells, const void* d_src_cells, int64_t n, void* stream) {
    if (n < 0) return fail(DRT_E_INVALID, "negative size");
    if (n == 0) return DRT_OK;
    if (!d_dst_cells || !d_src_cells) return fail(DRT_E_INVALID, "null pointer argument");
    k_fx_add<<<grid_for(n, 256, 1024), 256, 0, (hipStream_t)stream>>>(static_cast<FxCell*>(d_dst_cells), static_cast<const FxCell*>(d_src_cells), n);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}
int drt_fx_to_limbs(const void* d_cells, int64_t n, int64_t* d_limbs, void* stream) {
    if (n < 0) return fail(DRT_E_INVALID, "negative size");
    if (n == 0) return DRT_OK;
    if (!d_cells || !d_limbs) return fail(DRT_E_INVALID, "null pointer argument");
    k_fx_to_limbs<<<grid_for(n, 256, 1024), 256, 0, (hipStream_t)stream>>>(static_cast<const FxCell*>(d_cells), n, d_limbs);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}
int drt_fx_from_limbs(const int64_t* d_limbs, int64_t n, void* d_cells, void* stream) {
    if (n < 0) return fail(DRT_E_INVALID, "negative size");
    if (n == 0) return DRT_OK;
    if (!d_cells || !d_limbs) return fail(DRT_E_INVALID, "null pointer argument");
    k_fx_from_limbs<<<grid_for(n, 256, 1024), 256, 0, (hipStream_t)stream>>>(d_limbs, n, static_cast<FxCell*>(d_cells));
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}
int drt_fx_finalize(const void* d_cells, int64_t n, double* d_out, int accumulate, void* stream) {
    if (n < 0) return fail(DRT_E_INVALID, "negative size");
    if (n == 0) return DRT_OK;
    if (!d_cells || !d_out) return fail(DRT_E_INVALID, "null pointer argument");
    k_fx_finalize<<<grid_for(n, 256, 1024), 256, 0, (hipStream_t)stream>>>(static_cast<const FxCell*>(d_cells), n, d_out, accumulate);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

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
    if (e == hipSuccess) e = hipMalloc(&s->scratch, sizeof(unsigned long long) * 8);
    if (e == hipSuccess) e = hipMalloc(&s->bounds_acc, sizeof(uint32_t) * 12);
    if (e == hipSuccess) {
        const uint32_t init[12] = {0xFF800000u, 0xFF800000u, 0xFF800000u, 0x007FFFFFu, 0x007FFFFFu, 0x007FFFFFu,
                                   0xFF800000u, 0xFF800000u, 0xFF800000u, 0x007FFFFFu, 0x007FFFFFu, 0x007FFFFFu};     // +inf / -inf in k_cast_verts' ordered encoding
        e = hipMemcpy(s->bounds_acc, init, sizeof(init), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc(&s->vcount, sizeof(unsigned) * 4);
    if (e == hipSuccess) e = hipMemset(s->vcount, 0, sizeof(unsigned) * 4);      // ([3]: the device-side canary counter of k_check_views)
    if (e == hipSuccess) e = hipMalloc(&s->seg_counts, sizeof(unsigned) * drt_scene::kMaxSeg);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->fork_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->build_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->prefill_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->prefill_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->prefill_done_cap, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->build_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->build_done, hipEventDisableTiming);
    if (const char* ev = getenv("DRT_ASYNC_BUILD")) s->async_build = atoi(ev) != 0;
    if (const char* ev = getenv("DRT_CULL_DIRECT")) s->cull_direct = atoi(ev) != 0;
    if (const char* ev = getenv("DRT_CULL_PARK")) s->cull_park = atoi(ev) != 0;
    if (const char* ev = getenv("DRT_CULL_DIRECT_MIN_LOG2")) { const int v = atoi(ev); if (v >= 0 && v <= 40) s->cull_direct_min_rays = (int64_t)1 << v; }
    if (const char* ev = getenv("DRT_TREE")) { const int v = atoi(ev); if (v >= 0 && v <= 2) s->tree_mode = v; }
    if (const char* ev = getenv("DRT_REBUILD_EVERY")) { const int v = atoi(ev); if (v >= 1) s->rebuild_every = v; }
    if (const char* ev = getenv("DRT_STREAMS")) { const int v = atoi(ev); if (v >= 1 && v <= drt_scene::kMaxSub) s->n_sub = v; }
    if (const char* ev = getenv("DRT_RASTER")) s->use_raster = atoi(ev) != 0;
    if (const char* ev = getenv("DRT_HIT_SEED")) s->hit_seed = atoi(ev) != 0;
    if (const char* ev = getenv("DRT_SEED_TILED")) s->seed_tiled = atoi(ev) != 0;
    if (const char* ev = getenv("DRT_GRID_CANARY")) s->grid_canary = atoi(ev) != 0;
    if (const char* ev = getenv("DRT_FILL_OVERLAP")) s->fill_overlap = atoi(ev) != 0;
    if (const char* ev = getenv("DRT_FILL_AFTER_SHADE1")) s->fill_after_shade1 = atoi(ev) != 0;
    if (const char* ev = getenv("DRT_SUB_PER_STREAM")) { const int v = atoi(ev); if (v >= 1 && v <= 16) s->sub_per_stream = v; }
    if (const char* ev = getenv("DRT_MIN_SUB_LOG2")) { const int v = atoi(ev); if (v >= 12 && v <= 30) s->min_sub_rays = (int64_t)1 << v; }
    for (int k = 0; k < s->n_sub && e == hipSuccess; ++k) {
        drt_scene::Sub& w = s->sub[k];
        e = hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&w.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&w.fill_fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&w.fill_join, hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc(&w.qcount, sizeof(unsigned) * kQCount);
        if (e == hipSuccess) e = hipMalloc(&w.slow_stack, sizeof(int32_t) * (size_t)kRedoGrid * kTraceBlock * kStackSlowDev);
    }
    if (e == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) s->n_cu = prop.multiProcessorCount;
        int per_cu;
        per_cu = query_blocks_per_cu();           // drt_trace.hip: resident 128-thread blocks of the B1 query kernels
        s->grid_trace = s->n_cu * per_cu;
        if (s->grid_trace > kTraceGridMax) s->grid_trace = kTraceGridMax;
        per_cu = pipeline_blocks_per_cu();        // drt_pipeline.hip: resident 256-thread blocks of k_trace
        s->grid_path = s->n_cu * per_cu;
        if (s->grid_path * 2 > kTraceGridMax) s->grid_path = kTraceGridMax / 2;
        per_cu = mega_blocks_per_cu();            // drt_pipeline.hip: resident 256-thread blocks of k_path
        s->grid_mega = s->n_cu * per_cu;
        if (const char* e = getenv("DRT_MEGA_BPC")) { const int v = atoi(e); if (v >= 1 && v <= 8) s->grid_mega = v * s->n_cu; }
        if (const char* e = getenv("DRT_MEGA_MAX_LOG2")) { const int v = atoi(e); s->mega_max_rays = v <= 0 ? 0 : (v <= 31 ? (int64_t)1 << v : (int64_t)1 << 31); }
        if (const char* e = getenv("DRT_SHADE_MIN")) { const int v = atoi(e); if (v >= 1 && v <= 64) s->shade_min = v; }
        if (const char* e = getenv("DRT_MEGA_REFILL_MIN")) { const int v = atoi(e); if (v >= 1 && v <= 64) s->mega_refill_min = v; }
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
    if (s->build_stream) (void)hipStreamSynchronize(s->build_stream);
    scene_free_mesh(s);
    if (s->build_fork) (void)hipEventDestroy(s->build_fork);
    if (s->build_done) (void)hipEventDestroy(s->build_done);
    if (s->build_stream) (void)hipStreamDestroy(s->build_stream);
    (void)hipFree(s->params);
    (void)hipFree(s->slow_stack);
    (void)hipFree(s->scratch); (void)hipFree(s->bounds_acc); (void)hipFree(s->seg_counts);
    (void)hipFree(s->b1_list); (void)hipFree(s->b1_redo); (void)hipFree(s->b1_count);
    for (int j = 0; j < drt_scene::kMaxSub; ++j) {
        drt_scene::Sub& w = s->sub[j];
        for (int k = 0; k < 3; ++k) { (void)hipFree(w.q_idx[k]); (void)hipFree(w.q_ray[k]); (void)hipFree(w.q_face[k]); }
        (void)hipFree(w.tmp_face1); (void)hipFree(w.tmp_face2); (void)hipFree(w.qcount); (void)hipFree(w.slow_stack); (void)hipFree(w.redo);
        (void)hipFree(w.zbuf); (void)hipFree(w.zmask); (void)hipFree(w.vmodel); (void)hipFree(w.big); (void)hipFree(w.big_count); (void)hipFree(w.gen_list); (void)hipFree(w.ray64);
        if (w.done) (void)hipEventDestroy(w.done);
        if (w.fill_fork) (void)hipEventDestroy(w.fill_fork);
        if (w.fill_join) (void)hipEventDestroy(w.fill_join);
        if (w.stream) (void)hipStreamDestroy(w.stream);
    }
    (void)hipFree(s->vcount);
    (void)hipFree(s->vh_list);
    if (s->fork_ev) (void)hipEventDestroy(s->fork_ev);
    if (s->prefill_fork) (void)hipEventDestroy(s->prefill_fork);
    if (s->prefill_done) (void)hipEventDestroy(s->prefill_done);
    if (s->prefill_done_cap) (void)hipEventDestroy(s->prefill_done_cap);
    for (auto& e : s->prof_ev) (void)hipEventDestroy(e);
    (void)hipFree(s->prof_counts);
    delete s;
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

int drt_profile_select(drt_scene_t* s, uint32_t stage_mask) {
    CHECK_SCENE(s);
    s->prof_mask = stage_mask;
    return DRT_OK;
}

int drt_profile_read(drt_scene_t* s, double* ms_out, int64_t* launches_out, int64_t* items_out) {
    CHECK_SCENE(s);
    if (!ms_out || !launches_out || !items_out) return fail(DRT_E_INVALID, "null pointer argument");
    for (int k = 0; k < kProfStages; ++k) { ms_out[k] = 0.0; launches_out[k] = 0; items_out[k] = 0; }
    if (s->prof_ev.empty()) return DRT_OK;
    HIP_TRY(hipStreamSynchronize(s->prof_stream));
    if (s->build_stream) HIP_TRY(hipStreamSynchronize(s->build_stream));
    if (s->prof_dropped) {       // the stage times would under-report: say so instead of returning them
        const size_t lost = s->prof_dropped;
        s->prof_dropped = 0; s->prof_used = 0;
        return fail(DRT_E_INVALID, "%zu stage timings were not recorded (event pool exhausted): read the profile more often", lost);
    }
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

int drt_internal_stream(drt_scene_t* s, int which, void** out) {
    if (!s || !out) return fail(DRT_E_INVALID, "null pointer argument");
    if (which == 0) { *out = (void*)s->build_stream; return DRT_OK; }
    if (which >= 1 && which <= s->n_sub) { *out = (void*)s->sub[which - 1].stream; return DRT_OK; }
    return fail(DRT_E_INVALID, "no internal stream %d (0 = build, 1..%d = pipelines)", which, s->n_sub);
}

int drt_check_violations(int64_t* out4) {
    if (!out4) return fail(DRT_E_INVALID, "null pointer argument");
#if defined(DRT_CHECK)
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long tot[4] = {0, 0, 0, 0};
    if (check_counters_pipeline(tot) || check_counters_trace(tot)) return fail(DRT_E_HIP, "could not read the check counters");
    for (int k = 0; k < 4; ++k) out4[k] = (int64_t)tot[k];
#else
    for (int k = 0; k < 4; ++k) out4[k] = -1;
#endif
    return DRT_OK;
}

int drt_profile_trace_stats(drt_scene_t* s, int64_t* out12) {
    CHECK_SCENE(s);
    if (!out12) return fail(DRT_E_INVALID, "null pointer argument");
    for (int k = 0; k < 12; ++k) out12[k] = s->trace_stats[k];
    return DRT_OK;
}


}  // extern "C"
