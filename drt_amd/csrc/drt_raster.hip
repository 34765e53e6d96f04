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

// Test one (triangle, pixel) pair and fold a hit into the pixel's key.  The plain read of the current key may be stale
// (the L1 is not coherent with the atomics at L2) but keys only ever decrease, so a stale value is >= the true one:
// skipping when the new key is not smaller than what was read can never drop a winner.
__device__ __forceinline__ void raster_test(const TriRec& t, f3 o32, const double* __restrict__ dir, int64_t i,
                                            unsigned long long* __restrict__ zbuf, uint32_t* __restrict__ zmask) {
    const f3 d32 = to_f32(load_d3(dir, i));
    float tt;
    if (!tri_hit(o32, d32, f3{t.v0x, t.v0y, t.v0z}, f3{t.e1x, t.e1y, t.e1z}, f3{t.e2x, t.e2y, t.e2z}, tt)) return;
    const unsigned long long key = raster_key(tt, t.face);
    if (key >= zbuf[i]) return;
    atomicMin(&zbuf[i], key);
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
                    const unsigned slot = atomicAdd(big_count, 1u);
                    if (slot < big_cap) big[slot] = BigItem{view, k, me.x0, me.y0, me.nx, ny};
                    else __hip_atomic_store(&views[view].ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // list full: BVH path for this image
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
        raster_test(ow.tri, o32, dir, base + (int64_t)y * w + x, zbuf, zmask);
    }
}

// Triangles whose box holds more pixels than one lane should loop over: one block per list entry.
__global__ void __launch_bounds__(256) k_raster_big(const TriRec* __restrict__ tris, const ViewModel* __restrict__ views,
                                                    const double* __restrict__ dir, int w, int h,
                                                    unsigned long long* __restrict__ zbuf, uint32_t* __restrict__ zmask,
                                                    const BigItem* __restrict__ big, const unsigned* __restrict__ big_count, unsigned big_cap) {
    const unsigned n = min(*big_count, big_cap);
    for (unsigned e = blockIdx.x; e < n; e += gridDim.x) {
        const BigItem it = big[e];
        const ViewModel& vm = views[it.view];
        const TriRec t = tris[it.tri];
        const f3 o32 = to_f32(d3{vm.o[0], vm.o[1], vm.o[2]});
        const int64_t base = (int64_t)it.view * w * h;
        const int64_t cnt = (int64_t)it.nx * it.ny;
        for (int64_t p = threadIdx.x; p < cnt; p += 256) {
            const int y = it.y0 + (int)(p / it.nx), x = it.x0 + (int)(p % it.nx);
            raster_test(t, o32, dir, base + (int64_t)y * w + x, zbuf, zmask);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Tiled form of the pass (default; DRT_RASTER_TILES=0 selects k_raster above).  k_raster is bound by its 64-bit atomics --
// every improving (triangle, pixel) hit is one global atomicMin, ~1.3 per covered pixel, and the chip retires ~20 G of them
// per second: 0.20 of its 0.34 ms per 36 images (measured with the atomic compiled out) -- and then by the 24-byte
// direction gathers.  Here every 64 x 16 pixel tile of an image is OWNED by one block: the triangles are first binned to
// the tiles their padded boxes touch (count / scan / fill, wave-aggregated list atomics: a few per wave instead of one per
// triangle), then a block loads its tile's 1024 directions once (row-contiguous float64 -> float32 in LDS), folds its
// triangles' hits into a key tile in LDS (ds_min_u64) and writes the finished keys with plain stores.  Same tests on the
// same float32 rays, same keys, same minimum: only where the minimum is taken changes.  Both facings in one pass (LDS
// atomics are cheap); triangles whose box exceeds kRasterMaxPerLane pixels still go through k_raster_big afterwards.
constexpr int kTileW = 64, kTileH = 16, kTilePix = kTileW * kTileH;

struct TileCtl { uint32_t n_nonempty, total, next, overflow; };

// Aggregated `ctr[tile] += 1` over the lanes of a wave that name the same tile (tile < 0: lane does not take part); returns
// this lane's position within the tile's list segment (FILL) -- one atomic per DISTINCT tile of the wave.
template <bool FILL>
__device__ __forceinline__ uint32_t tile_claim(uint32_t* __restrict__ ctr, int tile, int lane) {
    uint32_t pos = 0;
    bool pending = tile >= 0;
    for (;;) {
        const unsigned long long todo = __ballot(pending);
        if (todo == 0ull) break;
        const int leader = __ffsll((long long)todo) - 1;
        const int lt = __shfl(tile, leader);
        const unsigned long long same = __ballot(pending && tile == lt);
        uint32_t base = 0;
        if (lane == leader) {
            if (FILL) base = atomicAdd(&ctr[lt], (uint32_t)__popcll(same));
            else atomicAdd(&ctr[lt], (uint32_t)__popcll(same));
        }
        if (FILL) base = (uint32_t)__shfl((int)base, leader);
        if (pending && tile == lt) {
            pos = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            pending = false;
        }
    }
    return pos;
}

// grid (ceil(F / 256), n_views).  FILL = false: count the (triangle, tile) pairs per tile, list the large triangles, flag
// images the camera plane cuts.  FILL = true (after k_tile_scan): write the triangle ids into the tiles' list segments.
template <bool FILL>
__global__ void __launch_bounds__(256) k_bin(const TriRec* __restrict__ tris, int n_tris, ViewModel* views, int w, int h, int tiles_x, int tiles_y,
                                             uint32_t* __restrict__ ctr, const uint32_t* __restrict__ off, uint32_t* __restrict__ list, const TileCtl* ctl,
                                             BigItem* __restrict__ big, unsigned* big_count, unsigned big_cap) {
    const int view = blockIdx.y;
    const ViewModel vm = views[view];
    if (!vm.ok) return;                                   // block-uniform
    if (FILL && ctl->overflow) return;
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const f3 o32 = to_f32(d3{vm.o[0], vm.o[1], vm.o[2]});
    int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
    if (k < n_tris) {
        const TriRec t = tris[k];
        const PixelBox box = project_tri_box(vm, o32, t, w, h);
        if (box.unsafe) {              // the camera plane cuts (or touches) this triangle: no projection bound for this image
            if (!FILL) __hip_atomic_store(&views[view].ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (box.x0 <= box.x1 && box.y0 <= box.y1) {
            const int nx = box.x1 - box.x0 + 1, ny = box.y1 - box.y0 + 1;
            if ((int64_t)nx * ny > kRasterMaxPerLane) {
                if (!FILL) {
                    const unsigned slot = atomicAdd(big_count, 1u);
                    if (slot < big_cap) big[slot] = BigItem{view, k, box.x0, box.y0, nx, ny};
                    else __hip_atomic_store(&views[view].ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // list full: BVH path for this image
                }
            } else {
                tx0 = box.x0 / kTileW; tx1 = box.x1 / kTileW; ty0 = box.y0 / kTileH; ty1 = box.y1 / kTileH;
            }
        }
    }
    // a box of <= kRasterMaxPerLane pixels touches at most 2 x 4 tiles; slot j of a lane = its j-th tile in row-major order
    const int ntx = tx1 - tx0 + 1, nt = ntx > 0 ? ntx * (ty1 - ty0 + 1) : 0;
    const int tile_base = view * tiles_x * tiles_y;
    for (int j = 0; __ballot(j < nt) != 0ull; ++j) {
        int tile = -1;
        if (j < nt) tile = tile_base + (ty0 + j / ntx) * tiles_x + tx0 + j % ntx;
        const uint32_t pos = tile_claim<FILL>(ctr, tile, lane);
        if (FILL && tile >= 0) list[off[tile] + pos] = (uint32_t)k;
    }
}

// List segments for the non-empty tiles: any order will do, so a tile takes `count` entries from one running total (one
// atomic per wave: wave prefix sums) and a slot in the compact list of non-empty tiles; the counts move to `cnt_keep` and
// the counters are zeroed again (k_bin<true> uses the same array as its cursors).
__global__ void __launch_bounds__(256) k_tile_alloc(uint32_t* __restrict__ cnt, int n_tiles, uint32_t* __restrict__ off, uint32_t* __restrict__ ids,
                                                     uint32_t* __restrict__ cnt_keep, TileCtl* ctl) {
    const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const uint32_t c = i < n_tiles ? cnt[i] : 0u;
    if (__ballot(c != 0u) == 0ull) return;                 // wave-uniform (the counters of an all-empty wave are zero already)
    uint32_t inc = c;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, o);
        if (lane >= o) inc += v;
    }
    const unsigned long long ne = __ballot(c != 0u);
    uint32_t base = 0, slot = 0;
    if (lane == 63) { base = atomicAdd(&ctl->total, inc); slot = atomicAdd(&ctl->n_nonempty, (uint32_t)__popcll(ne)); }
    base = (uint32_t)__shfl((int)base, 63); slot = (uint32_t)__shfl((int)slot, 63);
    if (i < n_tiles) {
        off[i] = base + inc - c; cnt_keep[i] = c; cnt[i] = 0u;
        if (c) ids[slot + (uint32_t)__popcll(ne & ((1ull << lane) - 1ull))] = (uint32_t)i;
    }
}

// (cannot happen with the capacity ensure_raster allocates -- 8 tiles per pair -- but never index past a buffer:) list
// overflow -> every image takes the BVH path.
__global__ void k_tile_overflow(TileCtl* ctl, ViewModel* views, int n_views, uint32_t list_cap) {
    if (ctl->total <= list_cap) return;
    if (threadIdx.x == 0) { ctl->overflow = 1u; ctl->n_nonempty = 0u; }
    for (int v = threadIdx.x; v < n_views; v += blockDim.x) views[v].ok = 0;
}

struct TileLane { TriRec tri; int32_t x0, y0, nx, end; };      // x0, y0 relative to the tile

__global__ void __launch_bounds__(256) k_tile_raster(const TriRec* __restrict__ tris, const ViewModel* __restrict__ views, const double* __restrict__ dir,
                                                     int w, int h, int tiles_x, int tiles_y, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off,
                                                     const uint32_t* __restrict__ list, const uint32_t* __restrict__ ids, TileCtl* ctl,
                                                     uint32_t* __restrict__ cursor, unsigned long long* __restrict__ zbuf, uint32_t* __restrict__ zmask) {
    __shared__ unsigned long long s_z[kTilePix];
    __shared__ float s_dx[kTilePix], s_dy[kTilePix], s_dz[kTilePix];
    __shared__ TileLane s_lane[256];
    const int tid = threadIdx.x, lane = tid & 63, wbase = tid & ~63;
    const uint32_t n_work = ctl->n_nonempty;
    const int per_view = tiles_x * tiles_y;
    // static dealing (a shared work counter would be one returning atomic per tile on ONE word: ~90 per microsecond)
    for (uint32_t work = blockIdx.x; work < n_work; work += gridDim.x) {
        __syncthreads();                                   // everyone is done with the previous tile's LDS
        const uint32_t id = ids[work];
        const int view = (int)(id / (uint32_t)per_view), tl = (int)(id % (uint32_t)per_view);
        const int px0 = (tl % tiles_x) * kTileW, py0 = (tl / tiles_x) * kTileH;
        const int rows = min(kTileH, h - py0);
        if (tid == 0) cursor[id] = 0u;  // k_bin<true> left the tile's count there: the counters must be zero again for the next call
        const ViewModel vm = views[view];
        if (!vm.ok) continue;           // flagged while it was being binned (a triangle cut by the camera plane, big list full): its lists are
                                        // incomplete -- k_bin<true> skipped it -- and every ray of the image takes the BVH path anyway
        const f3 o32 = to_f32(d3{vm.o[0], vm.o[1], vm.o[2]});
        const int64_t base = (int64_t)view * w * h + (int64_t)py0 * w + px0;       // ray of the tile's first pixel
        // the tile's rays: float32 directions (exactly what the traversal would use) and empty keys
        for (int p = tid; p < kTilePix; p += 256) {
            const int r = p >> 6, c = p & 63;
            s_z[p] = kRasterEmpty;
            if (r < rows) {
                const f3 d = to_f32(load_d3(dir, base + (int64_t)r * w + c));
                s_dx[p] = d.x; s_dy[p] = d.y; s_dz[p] = d.z;
            }
        }
        __syncthreads();
        const uint32_t n_list = cnt[id], first = off[id];
        for (uint32_t j0 = 0; j0 < n_list; j0 += 256) {
            const uint32_t j = j0 + (uint32_t)tid;
            int count = 0;
            TileLane me;
            me.x0 = me.y0 = 0; me.nx = 1;
            if (j < n_list) {
                me.tri = tris[list[first + j]];
                const PixelBox box = project_tri_box(vm, o32, me.tri, w, h);      // the box k_bin saw, clipped to this tile
                const int x0 = max(box.x0, px0), x1 = min(box.x1, px0 + kTileW - 1), y0 = max(box.y0, py0), y1 = min(box.y1, py0 + rows - 1);
                if (x0 <= x1 && y0 <= y1) {
                    me.x0 = x0 - px0; me.y0 = y0 - py0; me.nx = x1 - x0 + 1;
                    count = me.nx * (y1 - y0 + 1);
                }
            }
            int end = count;
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(end, o);
                if (lane >= o) end += v;
            }
            const int total = __shfl(end, 63);
            if (total == 0) continue;                      // wave-uniform
            me.end = end;
            s_lane[tid] = me;                              // read back by the lanes of this wave only
            __builtin_amdgcn_wave_barrier();
            for (int item0 = 0; item0 < total; item0 += 64) {
                const int item = item0 + lane;
                if (item >= total) break;
                int lo = 0, hi = 63;                       // first lane whose `end` exceeds item
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (s_lane[wbase + mid].end > item) hi = mid; else lo = mid + 1;
                }
                const TileLane& ow = s_lane[wbase + lo];
                const int local = item - (lo == 0 ? 0 : s_lane[wbase + lo - 1].end);
                const int row = (int)(((float)local + 0.5f) / (float)ow.nx);     // exact for these small integers
                const int p = ((ow.y0 + row) << 6) + ow.x0 + (local - row * ow.nx);
                float tt;
                if (tri_hit(o32, f3{s_dx[p], s_dy[p], s_dz[p]}, f3{ow.tri.v0x, ow.tri.v0y, ow.tri.v0z}, f3{ow.tri.e1x, ow.tri.e1y, ow.tri.e1z},
                            f3{ow.tri.e2x, ow.tri.e2y, ow.tri.e2z}, tt)) {
                    const unsigned long long key = raster_key(tt, ow.tri.face);
                    if (key < s_z[p]) atomicMin(&s_z[p], key);
                }
            }
            __builtin_amdgcn_wave_barrier();               // before this wave overwrites its s_lane entries
        }
        __syncthreads();
        // finished keys out (plain stores: this block owns the tile); one group bit per row of 64 rays that holds a key
        for (int p = tid; p < kTilePix; p += 256) {
            const int r = p >> 6, c = p & 63;
            const unsigned long long key = s_z[p];
            const bool hit = r < rows && key != kRasterEmpty;
            const int64_t i = base + (int64_t)r * w + c;
            if (hit) zbuf[i] = key;
            if (__ballot(hit) != 0ull && lane == 0) {
                const int64_t i0 = base + (int64_t)r * w;                       // 64-aligned: tile_w is a multiple of 64
                const uint32_t bit = 1u << ((i0 >> 6) & 31);
                if (!(zmask[i0 >> 11] & bit)) atomicOr(&zmask[i0 >> 11], bit);
            }
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

// Workspace of the tiled pass: per-tile counters / offsets / kept counts / ids, the control words, the binned triangle lists
// (a pair touches at most 8 tiles: a box of <= kRasterMaxPerLane pixels spans <= 2 tiles across and <= 4 down).
static int ensure_tiles(drt_scene* s, drt_scene::Sub& w, int n_views, int iw, int ih, hipStream_t st) {
    const int64_t tiles = (int64_t)n_views * (iw / kTileW) * ((ih + kTileH - 1) / kTileH);
    if (tiles > w.tile_cap) {
        (void)hipFree(w.tile_words); w.tile_words = nullptr; w.tile_cap = 0;
        HIP_TRY(hipMalloc(&w.tile_words, sizeof(uint32_t) * (4 * tiles + 4)));
        HIP_TRY(hipMemsetAsync(w.tile_words, 0, sizeof(uint32_t) * (4 * tiles + 4), st));     // counters start at zero (k_tile_scan re-zeroes them)
        w.tile_cap = tiles;
    }
    const int64_t pairs = 8 * (int64_t)n_views * (s->n_faces > 0 ? s->n_faces : 1);
    if (pairs > w.bin_cap) {
        if (pairs > (int64_t)UINT32_MAX) return fail(DRT_E_INVALID, "tile lists: too many (image, triangle) pairs in one sub-batch");
        (void)hipFree(w.bin_list); w.bin_list = nullptr; w.bin_cap = 0;
        HIP_TRY(hipMalloc(&w.bin_list, sizeof(uint32_t) * pairs));
        w.bin_cap = pairs;
    }
    return DRT_OK;
}

// Fit the image models and rasterise every triangle into the key buffer of `w` (rays [0, n_views * w * h) of the sub-batch).
int launch_raster(drt_scene* s, drt_scene::Sub& w, hipStream_t st, const double* d_origin, const double* d_dir, int n_views, int iw, int ih,
                  const ViewModel* trusted) {
    const int n = (int)s->n_faces;
    HIP_TRY(hipMemsetAsync(w.big_count, 0, sizeof(unsigned), st));
    if (trusted) HIP_TRY(hipMemcpyAsync(w.vmodel, trusted, sizeof(ViewModel) * n_views, hipMemcpyDeviceToDevice, st));   // models of an earlier call
    else k_fit_views<<<n_views, 64, 0, st>>>(d_origin, d_dir, iw, ih, w.vmodel);
    if (n > 0 && s->raster_tiles) {
        int rc = ensure_tiles(s, w, n_views, iw, ih, st);
        if (rc) return rc;
        const int tiles_x = iw / kTileW, tiles_y = (ih + kTileH - 1) / kTileH, n_tiles = n_views * tiles_x * tiles_y;
        uint32_t *cnt = w.tile_words, *off = cnt + w.tile_cap, *keep = off + w.tile_cap, *ids = keep + w.tile_cap;
        TileCtl* ctl = reinterpret_cast<TileCtl*>(ids + w.tile_cap);
        BigItem* big = reinterpret_cast<BigItem*>(w.big);
        const dim3 grid((n + 255) / 256, n_views);
        HIP_TRY(hipMemsetAsync(ctl, 0, sizeof(TileCtl), st));
        k_bin<false><<<grid, 256, 0, st>>>(s->tris_flat, n, w.vmodel, iw, ih, tiles_x, tiles_y, cnt, nullptr, nullptr, nullptr, big, w.big_count, drt_scene::kBigCap);
        k_tile_alloc<<<(n_tiles + 255) / 256, 256, 0, st>>>(cnt, n_tiles, off, ids, keep, ctl);
        k_tile_overflow<<<1, 64, 0, st>>>(ctl, w.vmodel, n_views, (uint32_t)w.bin_cap);
        k_bin<true><<<grid, 256, 0, st>>>(s->tris_flat, n, w.vmodel, iw, ih, tiles_x, tiles_y, cnt, off, w.bin_list, ctl, nullptr, nullptr, 0u);
        k_tile_raster<<<4 * s->n_cu, 256, 0, st>>>(s->tris_flat, w.vmodel, d_dir, iw, ih, tiles_x, tiles_y, keep, off, w.bin_list, ids, ctl, cnt, w.zbuf, w.zmask);
        k_raster_big<<<2 * s->n_cu, 256, 0, st>>>(s->tris_flat, w.vmodel, d_dir, iw, ih, w.zbuf, w.zmask, big, w.big_count, drt_scene::kBigCap);
    } else if (n > 0) {
        for (int pass = 0; pass < 2; ++pass) {
            k_raster<<<dim3((n + 255) / 256, n_views), 256, 0, st>>>(s->tris_flat, n, w.vmodel, d_dir, iw, ih, w.zbuf, w.zmask,
                                                                      reinterpret_cast<BigItem*>(w.big), w.big_count, drt_scene::kBigCap, pass);
            if (pass == 0) {   // the large triangles of the first launch before the second one reads the keys
                k_raster_big<<<2 * s->n_cu, 256, 0, st>>>(s->tris_flat, w.vmodel, d_dir, iw, ih, w.zbuf, w.zmask,
                                                           reinterpret_cast<const BigItem*>(w.big), w.big_count, drt_scene::kBigCap);
                HIP_TRY(hipMemsetAsync(w.big_count, 0, sizeof(unsigned), st));
            }
        }
        k_raster_big<<<2 * s->n_cu, 256, 0, st>>>(s->tris_flat, w.vmodel, d_dir, iw, ih, w.zbuf, w.zmask,
                                                   reinterpret_cast<const BigItem*>(w.big), w.big_count, drt_scene::kBigCap);
    }
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}
