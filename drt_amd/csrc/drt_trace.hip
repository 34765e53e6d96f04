// drt_trace.hip -- boundary B1 (closest / any hit per ray, the exhaustive diagnostic) and the closest-point query.
#include "drt_device.h"
#include "drt_trace_kernel.h"

// B1 prefilter: one thread per ray.  A ray that misses the top two levels of the wide tree is answered here (most
// camera rays of a view miss the object); the others go, as ray numbers, to the persistent traversal kernel the
// refraction pipeline uses (drt_trace_kernel.h).  HBM-bound: 24 B read + 8 B (closest) or 1 B (any) written per ray.
template <bool ANY>
__global__ void __launch_bounds__(kPathBlock) k_b1_cull(const Node4Q* __restrict__ nodes, int n_tris, const float* __restrict__ rays, int64_t n,
                                                         float* __restrict__ T, int32_t* __restrict__ ID, uint8_t* __restrict__ hitflag,
                                                         int32_t* __restrict__ list, unsigned* count) {
    __shared__ unsigned s_tmp[kPathWaves + 1];
    for (int64_t base = blockIdx.x * (int64_t)kPathBlock; base < n; base += (int64_t)gridDim.x * kPathBlock) {
        const int64_t i = base + threadIdx.x;
        bool cand = false;
        if (i < n) {
            const float* e = rays + 6 * i;
            cand = n_tris > 0 && hits_top_boxes(nodes, f3{e[0], e[1], e[2]}, f3{e[3], e[4], e[5]});
            if (!cand) {
                if (ANY) hitflag[i] = 0; else { T[i] = -1.0f; ID[i] = -1; }
            }
        }
        if (!__syncthreads_or(cand ? 1 : 0)) continue;
        const int slot = block_push(cand, count, s_tmp);
        if (slot >= 0) list[slot] = (int32_t)i;
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
            if (tri_hit(o, d, f3{t.v0x, t.v0y, t.v0z}, f3{t.e1x, t.e1y, t.e1z}, f3{t.e2x, t.e2y, t.e2z}, t.margin, tt)) {
                if (tt < best || (tt == best && t.face < best_face)) { best = tt; best_face = t.face; }
            }
        }
    }
    if (live) { T[i] = best_face >= 0 ? best : -1.0f; ID[i] = best_face; }
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

// The refraction pipeline's traversal launches (drt_pipeline.hip: lists R0 / R1 / R2).  The kernels are instantiated HERE, in the translation
// unit that holds nothing else of the pipeline: k_trace runs at exactly 64 VGPRs and the compiler's register allocation for it was found to
// change with the OTHER kernels of its translation unit (round 6: templating the gradient kernels of drt_pipeline.hip alone moved the seeded
// instantiation's scratch use from 12 to 20 bytes and put a store in its prologue) -- an edit elsewhere must not perturb the hot kernel.
void launch_trace_list(int variant, int grid, hipStream_t st, TraceCtx c, const float* rays, const unsigned* n_ptr, TraceOut out, int32_t* redo_list,
                       unsigned* redo_count, unsigned* done_count, int refill_min, int inner_min, unsigned long long* stats, TraceSeed sd) {
    switch (variant) {
    case kTraceClosest: k_trace<false, 0><<<grid, kPathBlock, 0, st>>>(c, rays, n_ptr, out, redo_list, redo_count, done_count, refill_min, inner_min, stats); break;
    case kTraceClosestListed: k_trace<false, 2><<<grid, kPathBlock, 0, st>>>(c, rays, n_ptr, out, redo_list, redo_count, done_count, refill_min, inner_min, stats); break;
    case kTraceClosestSeeded: k_trace<false, 0, true><<<grid, kPathBlock, 0, st>>>(c, rays, n_ptr, out, redo_list, redo_count, done_count, refill_min, inner_min, stats, sd); break;
    default: k_trace<true, 0><<<grid, kPathBlock, 0, st>>>(c, rays, n_ptr, out, redo_list, redo_count, done_count, refill_min, inner_min, stats); break;
    }
}
int pipeline_blocks_per_cu() {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace<false, 0>, kPathBlock, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    return per_cu;
}

#if defined(DRT_CHECK)
int check_counters_trace(unsigned long long* out4) { return read_check_counters(out4); }
#endif
int query_blocks_per_cu() {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_closest_point, kTraceBlock, 0) != hipSuccess || per_cu < 1) per_cu = 8;
    return per_cu;
}

// prefilter -> persistent traversal -> redo, all on the caller's stream; workspace grown on demand (a scene serves one
// B1 call at a time, like the reference's optix_mesh)
template <bool ANY>
static int b1_query(drt_scene* s, const float* d_rays, int64_t n_rays, float* d_T, int32_t* d_ID, uint8_t* d_hit, hipStream_t st) {
    if (n_rays > INT32_MAX) return fail(DRT_E_INVALID, "ray count out of range");
    { int rc = wait_build(s, st); if (rc) return rc; }
    { int rc = ensure_slow_stack(s, st); if (rc) return rc; }       // (first: it refuses a first use under stream capture with a message)
    if (n_rays > s->b1_cap) {
        (void)hipFree(s->b1_list); (void)hipFree(s->b1_redo);
        s->b1_list = s->b1_redo = nullptr; s->b1_cap = 0;
        HIP_TRY(hipMalloc(&s->b1_list, sizeof(int32_t) * n_rays));
        HIP_TRY(hipMalloc(&s->b1_redo, sizeof(int32_t) * n_rays));
        s->b1_cap = n_rays;
    }
    if (!s->b1_count) {      // zero once: k_trace<ANY, 1>'s last workgroup puts the three counters back to zero at the end of every query
        HIP_TRY(hipMalloc(&s->b1_count, sizeof(unsigned) * 4));
        HIP_TRY(hipMemset(s->b1_count, 0, sizeof(unsigned) * 4));
    }
    const TraceCtx tc = trace_ctx(s);
    const TraceOut out{d_ID, d_T, d_hit, s->b1_list};
    k_b1_cull<ANY><<<grid_for(n_rays, kPathBlock, 8 * s->n_cu), kPathBlock, 0, st>>>(tc.nodes, tc.n_tris, d_rays, n_rays, d_T, d_ID, d_hit, s->b1_list, s->b1_count);
    k_trace<ANY, 1><<<s->grid_path, kPathBlock, 0, st>>>(tc, d_rays, s->b1_count, out, s->b1_redo, s->b1_count + 1, s->b1_count + 2, s->refill_min, s->inner_min, nullptr);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

extern "C" {

int drt_intersect(drt_scene_t* s, const float* d_rays, int64_t n_rays, float* d_T, int32_t* d_ID, void* stream) {
    CHECK_BUILT(s);
    if (n_rays < 0 || (n_rays && (!d_rays || !d_T || !d_ID))) return fail(DRT_E_INVALID, "bad ray arguments");
    if (n_rays == 0) return DRT_OK;
    return b1_query<false>(s, d_rays, n_rays, d_T, d_ID, nullptr, (hipStream_t)stream);
}

int drt_intersect_any(drt_scene_t* s, const float* d_rays, int64_t n_rays, uint8_t* d_hit, void* stream) {
    CHECK_BUILT(s);
    if (n_rays < 0 || (n_rays && (!d_rays || !d_hit))) return fail(DRT_E_INVALID, "bad ray arguments");
    if (n_rays == 0) return DRT_OK;
    return b1_query<true>(s, d_rays, n_rays, nullptr, nullptr, d_hit, (hipStream_t)stream);
}

int drt_intersect_bruteforce(drt_scene_t* s, const float* d_rays, int64_t n_rays, float* d_T, int32_t* d_ID, void* stream) {
    CHECK_BUILT(s);
    if (n_rays < 0 || (n_rays && (!d_rays || !d_T || !d_ID))) return fail(DRT_E_INVALID, "bad ray arguments");
    if (n_rays == 0) return DRT_OK;
    { int rc = wait_build(s, (hipStream_t)stream); if (rc) return rc; }
    k_bruteforce<<<(unsigned)((n_rays + 255) / 256), 256, 0, (hipStream_t)stream>>>(s->tris, (int)s->n_faces, d_rays, n_rays, d_T, d_ID);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_closest_point(drt_scene_t* s, const double* d_points, int64_t n, double* d_dist, int32_t* d_face, double* d_closest, void* stream) {
    CHECK_BUILT(s);
    if (n < 0) return fail(DRT_E_INVALID, "negative point count");
    if (n == 0) return DRT_OK;
    if (!d_points || !d_dist) return fail(DRT_E_INVALID, "null pointer argument");
    { int rc = wait_build(s, (hipStream_t)stream); if (rc) return rc; }
    { int rc = ensure_slow_stack(s, (hipStream_t)stream); if (rc) return rc; }
    k_closest_point<<<grid_for(n, kTraceBlock, s->grid_trace), kTraceBlock, 0, (hipStream_t)stream>>>(trace_ctx(s), s->faces, s->verts, d_points, n, d_dist, d_face, d_closest);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}


}  // extern "C"
