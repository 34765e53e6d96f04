// drt_raster.hip -- kernels of the projected primary-visibility pass (drt_raster.h) and their launcher.
#include "drt_device.h"

// One block per image: lane 0 fits the pinhole model from the four corner rays, then 64 lanes check an 8x8 lattice of
// the image's rays against it; an image that is not a pinhole grid keeps ok = 0 and takes the BVH path entirely
// (k_raster skips it; k_cull verifies every ray once more anyway).
__global__ void __launch_bounds__(64) k_fit_views(const double* __restrict__ origin, const double* __restrict__ dir, int w, int h,
                                                  ViewModel* __restrict__ views) {
    __shared__ ViewModel vm;
    const int64_t base = (int64_t)blockIdx.x * w * h;
    if (threadIdx.x == 0) {
        const int64_t i00 = base, iW0 = base + (w - 1), i0H = base + (int64_t)(h - 1) * w, iWH = i0H + (w - 1);
        if (fit_view_model(load_d3(origin, i00), load_d3(dir, i00), load_d3(dir, iW0), load_d3(dir, i0H), load_d3(dir, iWH),
                           (double)(w - 1), (double)(h - 1), vm))
            vm.ok = 1;
    }
    __syncthreads();
    bool good = vm.ok != 0;
    if (good) {
        const int sx = (int)(threadIdx.x & 7), sy = (int)(threadIdx.x >> 3);
        const int x = (int)(((int64_t)(w - 1) * sx) / 7), y = (int)(((int64_t)(h - 1) * sy) / 7);
        const int64_t i = base + (int64_t)y * w + x;
        good = view_verify(vm, load_d3(origin, i), load_d3(dir, i), (double)x, (double)y);
    }
    const bool all_good = __ballot(good) == ~0ull;
    if (threadIdx.x == 0) {
        vm.ok = all_good ? 1 : 0;
        vm.all = vm.ok;              // k_cull clears it when a ray of the image does not verify
        views[blockIdx.x] = vm;
    }
}

// DRT_GRID_TRUST: the models come from the caller's cache -- but a trusted image is not taken on faith entirely: the 8x8 lattice of
// its rays (the 64 rays k_fit_views samples) is read again and checked against the cached model, so that ray tensors that were
// replaced wholesale behind the cache's back (a write through `.data` or a raw pointer, which no version counter sees) are caught for
// the price of 64 ray loads per image.  An image that fails is re-fitted from its corner rays on the spot and loses `all` -- in the
// working copy AND in the cache -- so that every one of its rays is verified individually (k_cull_listed's general path), in this
// call and in every later one, exactly as in a call without a cache.
// The CANARY (`salt` != 0): besides the fixed lattice every lane re-verifies one more ray of the image at a pseudo-random pixel that changes
// from call to call (a hash of the call's salt, the image and the lane): 64 rays per image and call that a partial overwrite behind the
// version counter's back (a block of rows written through a raw pointer, `t.data[...] = ...`) cannot avoid for long -- a fraction f of
// modified rays survives a call with probability (1 - f)^64 per image.  Same consequence as a lattice mismatch.
__device__ __forceinline__ uint32_t canary_hash(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (c + 0x165667B1u) * 0xC2B2AE35u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
__global__ void __launch_bounds__(64) k_check_views(const double* __restrict__ origin, const double* __restrict__ dir, int w, int h,
                                                    ViewModel* __restrict__ cache, ViewModel* __restrict__ views, unsigned salt, unsigned* __restrict__ salt_ctr) {
    __shared__ ViewModel vm;
    __shared__ unsigned s_ctr;
    const int64_t base = (int64_t)blockIdx.x * w * h;
    if (threadIdx.x == 0) {
        vm = cache[blockIdx.x];
        // The call's salt is a HOST counter: baked into a captured graph, every replay would re-check the same 64 pixels.  A DEVICE counter that
        // every launch bumps is mixed in, so that replays move on like eager calls do (which value a block happens to read does not matter:
        // any pixel is a legitimate canary).
        s_ctr = salt_ctr ? __hip_atomic_load(salt_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (salt_ctr && blockIdx.x == 0) atomicAdd(salt_ctr, 1u);
    }
    __syncthreads();
    if (salt) salt = (salt ^ (s_ctr * 0x9E3779B1u)) | 1u;
    const bool trusted = vm.ok && vm.all;              // (k_store_models stores ok == all)
    const int sx = (int)(threadIdx.x & 7), sy = (int)(threadIdx.x >> 3);
    const int x = (int)(((int64_t)(w - 1) * sx) / 7), y = (int)(((int64_t)(h - 1) * sy) / 7);
    const int64_t i = base + (int64_t)y * w + x;
    bool good = trusted && view_verify(vm, load_d3(origin, i), load_d3(dir, i), (double)x, (double)y);
    if (good && salt) {
        const uint32_t hx = canary_hash(salt, blockIdx.x, threadIdx.x), hy = canary_hash(salt ^ 0xA5A5A5A5u, threadIdx.x, blockIdx.x);
        const int cx = (int)(((uint64_t)hx * (uint64_t)w) >> 32), cy = (int)(((uint64_t)hy * (uint64_t)h) >> 32);
        const int64_t ci = base + (int64_t)cy * w + cx;
        good = view_verify(vm, load_d3(origin, ci), load_d3(dir, ci), (double)cx, (double)cy);
    }
    if (__ballot(good) == ~0ull) {                     // wave-uniform: the usual case
        if (threadIdx.x == 0) views[blockIdx.x] = vm;
        return;
    }
    // not (or no longer) a trusted grid: fit it like a call without a cache would
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t i00 = base, iW0 = base + (w - 1), i0H = base + (int64_t)(h - 1) * w, iWH = i0H + (w - 1);
        if (fit_view_model(load_d3(origin, i00), load_d3(dir, i00), load_d3(dir, iW0), load_d3(dir, i0H), load_d3(dir, iWH),
                           (double)(w - 1), (double)(h - 1), vm))
            vm.ok = 1;
    }
    __syncthreads();
    good = vm.ok != 0 && view_verify(vm, load_d3(origin, i), load_d3(dir, i), (double)x, (double)y);
    const bool all_good = __ballot(good) == ~0ull;
    if (threadIdx.x == 0) {
        vm.ok = all_good ? 1 : 0;
        vm.all = 0;                                    // never trusted again: per-ray verification from now on
        views[blockIdx.x] = vm;
        if (trusted) { cache[blockIdx.x].ok = 0; cache[blockIdx.x].all = 0; }
    }
}

// Test one (triangle, pixel) pair and fold a hit into the pixel's key.  The plain read of the current key may be stale
// (the L1 is not coherent with the atomics at L2) but keys only ever decrease, so a stale value is >= the true one:
// skipping when the new key is not smaller than what was read can never drop a winner.
__device__ __forceinline__ void raster_test(const TriRec& t, f3 o32, const double* __restrict__ dir, int64_t i, int64_t slot,
                                            unsigned long long* __restrict__ zbuf, uint32_t* __restrict__ zmask) {
    const f3 d32 = to_f32(load_d3(dir, i));
    float tt;
    // (tri_hit's conditions with the cheap ones first: the barycentric test, "would this key win", and only then the hit-point test)
    if (!tri_hit_mt(o32, d32, f3{t.v0x, t.v0y, t.v0z}, f3{t.e1x, t.e1y, t.e1z}, f3{t.e2x, t.e2y, t.e2z}, tt)) return;
    const unsigned long long key = raster_key(tt, t.face);
    if (key >= zbuf[slot]) return;
    if (!hit_point_in_box(o32, d32, tt, f3{t.v0x, t.v0y, t.v0z}, f3{t.e1x, t.e1y, t.e1z}, f3{t.e2x, t.e2y, t.e2z}, t.margin)) return;
    atomicMin(&zbuf[slot], key);
    // one bit per 64 consecutive rays: k_cull reads keys only there.  Thousands of hits share a word, and atomics on ONE
    // address are served one at a time: set the bit only when a (possibly stale) read does not show it yet
    const uint32_t bit = 1u << ((i >> 6) & 31);
    if (!(zmask[i >> 11] & bit)) atomicOr(&zmask[i >> 11], bit);
}

struct BigItem { int32_t view, tri, x0, y0, nx, ny; };

// grid (ceil(F / 256), n_views): one thread per (image, triangle in Morton order) projects its triangle (float32 is
// ample: the box is padded by 1/16 pixel) and counts the pixel centres inside the padded box -- 0 for most
// sub-pixel triangles, a handful typically, dozens for a few.  The (triangle, pixel) tests of a wave are then dealt to its
// lanes 64 at a time (wave prefix sum of the counts, owner found by bisection in LDS), so that one fat triangle does not
// hold 63 idle lanes.
struct RasterLane {            // what a lane publishes for the wave: its triangle and its box
    TriRec tri;
    int32_t x0, y0, nx, end;   // `end` = inclusive prefix sum of the counts (first work item NOT of this lane)
};

__global__ void __launch_bounds__(256) k_raster(const TriRec* __restrict__ tris, int n_tris, ViewModel* views,
                                                const double* __restrict__ dir, int w, int h,
                                                unsigned long long* __restrict__ zbuf, uint32_t* __restrict__ zmask,
                                                BigItem* __restrict__ big, unsigned* big_count, unsigned big_cap, int pass) {
    __shared__ RasterLane s_lane[256];
    const int view = blockIdx.y;
    const ViewModel vm = views[view];
    if (!vm.ok) return;                                   // block-uniform
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
    const d3 o{vm.o[0], vm.o[1], vm.o[2]};
    const f3 o32 = to_f32(o);
    int count = 0;
    RasterLane me;
    me.x0 = me.y0 = 0; me.nx = 1;
    if (k < n_tris) {
        me.tri = tris[k];
        // Two launches: triangles facing the camera first (pass 0), the others second (pass 1).  Every triangle is handled in
        // exactly one of them, so the result is the same minimum; but a pixel's closest hit is nearly always a front face,
        // and the second launch sees the finished keys of the first: its read-before-atomic check then skips almost every
        // back face, which cuts the 64-bit atomics -- what bounds this kernel -- by more than half.
        if ((pass == 1) == tri_faces_away(o32, me.tri)) {
            const PixelBox box = project_tri_box(vm, o32, me.tri, w, h);
            if (box.unsafe) {          // the camera plane cuts (or touches) this triangle: no projection bound for this image
                __hip_atomic_store(&views[view].ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (box.x0 <= box.x1 && box.y0 <= box.y1) {
                me.x0 = box.x0; me.y0 = box.y0; me.nx = box.x1 - box.x0 + 1;
                const int ny = box.y1 - box.y0 + 1;
                const int64_t cnt = (int64_t)me.nx * ny;
                if (cnt > kRasterMaxPerLane) {
                    // listed for k_raster_big, one WAVE per entry: in bands of whole rows of at most kRasterBandPixels pixel centres
                    const int rows_per = max(1, kRasterBandPixels / me.nx), n_bands = (ny + rows_per - 1) / rows_per;
                    const unsigned slot = atomicAdd(big_count, (unsigned)n_bands);
                    if (slot <= big_cap && (unsigned)n_bands <= big_cap - slot) {
                        for (int b = 0; b < n_bands; ++b)
                            big[slot + b] = BigItem{view, k, me.x0, me.y0 + b * rows_per, me.nx, min(rows_per, ny - b * rows_per)};
                    } else {
                        __hip_atomic_store(&views[view].ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // list full: BVH path for this image
                        for (unsigned q = slot; q < big_cap; ++q) big[q] = BigItem{view, k, 0, 0, 0, 0};      // (what was reserved below the cap: empty entries)
                    }
                } else {
                    count = (int)cnt;
                }
            }
        }
    }
    // inclusive prefix sum of the counts over the wave
    int end = count;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(end, off);
        if (lane >= off) end += v;
    }
    const int total = __shfl(end, 63);
    if (total == 0) return;                                // wave-uniform: most waves of a view see nothing but background
    me.end = end;
    s_lane[threadIdx.x] = me;                              // read back by the lanes of this wave only: no block barrier needed
    __builtin_amdgcn_wave_barrier();
    const int64_t base = (int64_t)view * w * h;
    for (int item0 = 0; item0 < total; item0 += 64) {
        const int item = item0 + lane;
        if (item >= total) break;
        int lo = 0, hi = 63;                               // first lane whose `end` exceeds item
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_lane[wbase + mid].end > item) hi = mid; else lo = mid + 1;
        }
        const RasterLane& ow = s_lane[wbase + lo];
        const int local = item - (lo == 0 ? 0 : s_lane[wbase + lo - 1].end);
        const int row = (int)(((float)local + 0.5f) / (float)ow.nx);     // exact for these small integers
        const int x = ow.x0 + (local - row * ow.nx), y = ow.y0 + row;
        raster_test(ow.tri, o32, dir, base + (int64_t)y * w + x, raster_slot((unsigned)x, (unsigned)(view * h + y), (unsigned)w), zbuf, zmask);
    }
}

// Triangles whose box holds more pixels than one lane should loop over: one WAVE per list entry (a band of rows of the box).  With the object
// filling the image a third of the camera-facing triangles are such entries, 50-300 pixel centres each: one BLOCK per entry on 2 blocks per
// CU -- the first form, written for the odd large triangle -- kept 512 entries in flight and took 0.4-0.7 ms per launch, twice the main
// kernel; a wave per entry on a full grid keeps 8192.
__global__ void __launch_bounds__(256) k_raster_big(const TriRec* __restrict__ tris, const ViewModel* __restrict__ views,
                                                    const double* __restrict__ dir, int w, int h,
                                                    unsigned long long* __restrict__ zbuf, uint32_t* __restrict__ zmask,
                                                    const BigItem* __restrict__ big, const unsigned* __restrict__ big_count, unsigned big_cap) {
    const unsigned n = min(*big_count, big_cap);          // (an overfull list was cut at an entry boundary: k_raster wrote whole triangles only)
    const unsigned n_waves = gridDim.x * 4u;
    const int lane = threadIdx.x & 63;
    for (unsigned e = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6)); e < n; e += n_waves) {
        const BigItem it = big[e];
        const ViewModel& vm = views[it.view];
        const TriRec t = tris[it.tri];
        const f3 o32 = to_f32(d3{vm.o[0], vm.o[1], vm.o[2]});
        const int64_t base = (int64_t)it.view * w * h;
        const int cnt = it.nx * it.ny;
        for (int p = lane; p < cnt; p += 64) {
            const int row = (int)(((float)p + 0.5f) / (float)it.nx);        // exact for p < 2^22
            int y = it.y0 + row, x = it.x0 + (p - row * it.nx);
            if (x >= it.x0 + it.nx) { x -= it.nx; ++y; } else if (x < it.x0) { x += it.nx; --y; }     // (belt and braces for bands of one very long row)
            raster_test(t, o32, dir, base + (int64_t)y * w + x, raster_slot((unsigned)x, (unsigned)(it.view * h + y), (unsigned)w), zbuf, zmask);
        }
    }
}

__global__ void k_fill_u64(unsigned long long* __restrict__ p, int64_t n, unsigned long long v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// Workspace of the raster pass of one internal stream: keys [cap] (kept all-empty between calls: k_cull resets every
// key it consumes), group bits, image models, the big-triangle list.
int ensure_raster(drt_scene* s, drt_scene::Sub& w, int64_t n_rays, int n_views, hipStream_t st) {
    if (n_rays > w.z_cap) {
        (void)hipFree(w.zbuf); (void)hipFree(w.zmask);
        w.zbuf = nullptr; w.zmask = nullptr; w.z_cap = 0;
        const int64_t words = (n_rays + 2047) / 2048 + 1;
        HIP_TRY(hipMalloc(&w.zbuf, sizeof(unsigned long long) * n_rays));
        HIP_TRY(hipMalloc(&w.zmask, sizeof(uint32_t) * words));
        k_fill_u64<<<4 * s->n_cu, 256, 0, st>>>(w.zbuf, n_rays, kRasterEmpty);
        HIP_TRY(hipMemsetAsync(w.zmask, 0, sizeof(uint32_t) * words, st));
        w.z_cap = n_rays;
    }
    if (n_views > w.vm_cap) {
        (void)hipFree(w.vmodel); w.vmodel = nullptr; w.vm_cap = 0;
        HIP_TRY(hipMalloc(&w.vmodel, sizeof(ViewModel) * n_views));
        w.vm_cap = n_views;
    }
    if (!w.big) {
        HIP_TRY(hipMalloc(&w.big, sizeof(BigItem) * drt_scene::kBigCap));
        HIP_TRY(hipMalloc(&w.big_count, sizeof(unsigned)));
    }
    return DRT_OK;
}

// Fit the image models and rasterise every triangle into the key buffer of `w` (rays [0, n_views * w * h) of the sub-batch).
int launch_raster(drt_scene* s, drt_scene::Sub& w, hipStream_t st, const double* d_origin, const double* d_dir, int n_views, int iw, int ih,
                  ViewModel* trusted) {
    const int n = (int)s->n_faces;
    HIP_TRY(hipMemsetAsync(w.big_count, 0, sizeof(unsigned), st));
    if (trusted) k_check_views<<<n_views, 64, 0, st>>>(d_origin, d_dir, iw, ih, trusted, w.vmodel,           // models of an earlier call: lattice re-checked,
                                                        s->grid_canary ? (++s->canary_salt ? s->canary_salt : ++s->canary_salt) : 0u,    // + 64 rays at this call's random pixels
                                                        s->vcount + 3);                                                                  //   (device counter: moves on under graph replay too)
    else k_fit_views<<<n_views, 64, 0, st>>>(d_origin, d_dir, iw, ih, w.vmodel);
    if (n > 0) {
        for (int pass = 0; pass < 2; ++pass) {
            k_raster<<<dim3((n + 255) / 256, n_views), 256, 0, st>>>(s->tris_flat, n, w.vmodel, d_dir, iw, ih, w.zbuf, w.zmask,
                                                                      reinterpret_cast<BigItem*>(w.big), w.big_count, drt_scene::kBigCap, pass);
            if (pass == 0) {   // the large triangles of the first launch before the second one reads the keys
                k_raster_big<<<8 * s->n_cu, 256, 0, st>>>(s->tris_flat, w.vmodel, d_dir, iw, ih, w.zbuf, w.zmask,
                                                           reinterpret_cast<const BigItem*>(w.big), w.big_count, drt_scene::kBigCap);
                HIP_TRY(hipMemsetAsync(w.big_count, 0, sizeof(unsigned), st));
            }
        }
        k_raster_big<<<8 * s->n_cu, 256, 0, st>>>(s->tris_flat, w.vmodel, d_dir, iw, ih, w.zbuf, w.zmask,
                                                   reinterpret_cast<const BigItem*>(w.big), w.big_count, drt_scene::kBigCap);
    }
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}
