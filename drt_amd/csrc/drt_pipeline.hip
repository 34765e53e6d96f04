// drt_pipeline.hip -- the two-bounce refraction path (render_transparent), its backward, ray_loss and the fused loss.
#include "drt_device.h"
#include "drt_trace_kernel.h"

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

struct RayList {
    int32_t* idx;     // ray index within the chunk
    float* ray;       // [cap,6] float32 origin, direction -- exactly what the tracer sees
    int32_t* face;    // [cap] traversal result
};
// where the float64 refracted ray of ray i is parked between bounce #1 (k_shade1 / k_gen_late) and bounce #2 inside k_path: rows of the
// dense outputs themselves (overwritten by the exit ray or by zeros later), or a side buffer in the fused form; null for the staged kernels
struct Ray64 {
    double* o;      // [N,3]
    double* d;      // [N,3]
};
struct Pipe {
    RayList r0, r1, r2;
    unsigned* count;   // [0..2] list sizes of the sub-batch in flight, [4..6] rays handed to the second pass of k_trace per stage, [16] its retired-workgroup counter
    unsigned* valid;   // number of valid rays of the whole call (shared by all sub-batches)
    int32_t* redo;     // list entries whose traversal overflowed the LDS stack
};


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

// `tile_w` > 0: the rays are rows of an image `tile_w` pixels wide (any number of images of a
// multiple-of-4 height, concatenated).  A block then takes a 64x4 pixel patch per iteration and
// appends its candidates in 16x4-tile order, so that the 64 rays a traversal wave picks up come from
// a compact screen region (same BVH nodes, same depth) instead of a 64x1 strip.
// Dead outputs of one wave's 64-aligned run of 64 rays starting at i0: 1536 + 1536 + 192 + 256 + 256 bytes in lane-consecutive
// 16-byte stores (per-ray stores -- 8-byte pieces at a 24-byte stride, single mask bytes -- move the same bytes at half the rate).
template <bool FUSED>
__device__ __forceinline__ void write_dead_row(int64_t i0, int lane, double* out_ori, double* out_dir, uint8_t* mask, int32_t* face1, int32_t* face2) {
    const F4 zero{0.f, 0.f, 0.f, 0.f};
    F4 ones;
    { const int32_t m1 = -1; memcpy(&ones.x, &m1, 4); ones.y = ones.x; ones.z = ones.x; ones.w = ones.x; }
    if (!FUSED) {
        F4* po = reinterpret_cast<F4*>(out_ori + 3 * i0);
        F4* pd = reinterpret_cast<F4*>(out_dir + 3 * i0);
        po[lane] = zero; pd[lane] = zero;
        if (lane < 32) { po[64 + lane] = zero; pd[64 + lane] = zero; }
        if (lane < 12) reinterpret_cast<F4*>(mask + 3 * i0)[lane] = zero;
        if (lane >= 16 && lane < 32) reinterpret_cast<F4*>(face2 + i0)[lane - 16] = ones;
    }
    if (lane >= 32 && lane < 48) reinterpret_cast<F4*>(face1 + i0)[lane - 32] = ones;
}

// Raster mode (`rz.views` non-null; needs tile_w > 0): the primary hit of a ray that verifies as a ray of its image's
// pinhole grid was decided by k_raster (drt_raster.h) -- its key is read here, consumed (reset to empty) and the ray
// goes to R0 with its face already known; only rays that do not verify take the top-box test and are listed in
// `gen_list` for the traversal kernel, exactly as every candidate was before.
struct RasterIn {
    ViewModel* views;                // null: BVH path for every ray
    int mode;                        // DRT_GRID_*: TRUST: images with ok && all need no verification, and no ray loads where no key was written
    unsigned long long* zbuf;
    uint32_t* zmask;
    int32_t* gen_list;               // R0 slots that still need k_trace (count in p.count[3])
    int img_w, img_h;
    int pending;                     // a ray outside the grid is listed with face = kFacePending: its primary hit is found later (k_gen_late behind the
                                     // asynchronous build under DRT_GRID_ALL_VERIFIED; stage T0 of k_path), not by a k_trace launch in front of k_shade1
    int no_tree;                     // DRT_GRID_ALL_VERIFIED: the tree may still be under construction during the cull stage -- no top-box test
};
constexpr int32_t kFacePending = -2; // R0 entry whose primary hit is still to be found by the tree (k_shade1 leaves it alone)

struct CullShared {
    unsigned tmp[kPathWaves + 1];
    uint8_t flag[kPathBlock];
    int slot[kPathBlock];
};

// One patch of 256 rays (must be reached by the whole block: barriers).  `prefilled`: the dense outputs already hold the
// dead values (zeros / -1) everywhere, so dead rays write nothing.
template <bool FUSED>
__device__ __forceinline__ void cull_patch(unsigned patch, bool prefilled, CullShared& sh, const Node4Q* __restrict__ nodes, int n_tris,
                                           const double* __restrict__ origin, const double* __restrict__ dir,
                                           const uint8_t* __restrict__ valid, int64_t n, double* __restrict__ out_ori,
                                           double* __restrict__ out_dir, uint8_t* __restrict__ mask,
                                           int32_t* __restrict__ face1, int32_t* __restrict__ face2, const Pipe& p, int tile_w, const RasterIn& rz) {
    unsigned* s_tmp = sh.tmp;
    uint8_t* s_flag = sh.flag;
    int* s_slot = sh.slot;
    const bool dead_writes = !FUSED && !prefilled;       // (the fused pass reads face ids of listed rays only: nothing to write for dead ones)
    const int tid = threadIdx.x;
    const bool raster = rz.views != nullptr;
    // (`nodes` is a __restrict__ parameter of its own, not the TraceCtx struct, so that the compiler can prove the root
    // node is never clobbered: it is then fetched once, through the scalar cache, instead of by four vector loads per ray)
    // patch = 64 pixels wide x 4 rows (every wave reads one full 1536-byte row segment); tiles of 16x4 pixels
    const int vt = ((tid & 63) >> 4) * 64 + (tid >> 6) * 16 + (tid & 15);   // position of this thread's pixel in tile order
    // (block-uniform 32-bit index arithmetic: n < 2^31)
    const unsigned patches_per_row = tile_w > 0 ? (unsigned)tile_w / 64u : 1u;
    {
        int64_t i = (int64_t)patch * kPathBlock + tid;
        unsigned y = 0, x = 0;
        if (tile_w > 0) {
            const unsigned prow = patch / patches_per_row;
            y = 4u * prow + (unsigned)(tid >> 6); x = 64u * (patch - prow * patches_per_row) + (unsigned)(tid & 63);
            i = (int64_t)y * tile_w + x;
        }
        if (raster && rz.mode == DRT_GRID_TRUST) {
            // A patch of a trusted image on which no projected triangle wrote a key: nothing to decide, nothing to load.
            // The test is block-uniform (every thread reads the same four group bits, which only this block ever clears),
            // so the block skips the rest of the iteration -- barriers included -- together.
            const unsigned view = y / (unsigned)rz.img_h;
            if (rz.views[view].ok && rz.views[view].all) {
                const int64_t row0 = i - (int64_t)(tid >> 6) * tile_w - (tid & 63);      // first ray of the patch
                unsigned any = 0;
                for (int r = 0; r < 4; ++r) {
                    const int64_t g = (row0 + (int64_t)r * tile_w) >> 6;
                    any |= rz.zmask[g >> 5] >> (g & 31);
                }
                if (!(any & 1u)) {
                    if (dead_writes) write_dead_row<FUSED>(i - (tid & 63), tid & 63, out_ori, out_dir, mask, face1, face2);
                    return;
                }
            }
        }
        bool cand = false, generic = false;
        int32_t face = -1;
        unsigned clear_bit = 0;
        f3 o{0.f, 0.f, 0.f}, d{0.f, 0.f, 1.f};
        if (i < n) {
            unsigned long long key = kRasterEmpty;
            bool trusted = false;
            unsigned view = 0;
            if (raster) {
                view = y / (unsigned)rz.img_h;                         // block-uniform: a patch never straddles two images
                trusted = rz.mode == DRT_GRID_TRUST && rz.views[view].ok && rz.views[view].all;
                // the 64 rays of a wave are one 64-aligned run: one group bit decides whether any key was written here
                const unsigned word = rz.zmask[i >> 11], bit = 1u << ((i >> 6) & 31);
                if (word & bit) {
                    const int64_t zs = raster_slot(x, y, (unsigned)tile_w);
                    key = rz.zbuf[zs];
                    if (key != kRasterEmpty) rz.zbuf[zs] = kRasterEmpty;         // consumed: the buffer is empty again for the next call
                    clear_bit = bit;                                             // (cleared after the barrier below: see there)
                }
            }
            if (trusted) {
                // every ray of this image is known to be a grid ray: its primary hit is its key, and a pixel without a
                // key needs nothing from its ray -- no load at all for nine patches out of ten
                if (key != kRasterEmpty && (!FUSED || valid[i])) {
                    o = to_f32(load_d3(origin, i)); d = to_f32(load_d3(dir, i));
                    cand = true;
                    face = (int32_t)(uint32_t)key;
                }
            } else if (!FUSED || valid[i]) {
                // the fused loss ignores pixels without a target (reference optim.py:105): their rays are not traced
                const d3 o64 = load_d3(origin, i), d64 = load_d3(dir, i);
                o = to_f32(o64); d = to_f32(d64);
                bool verified = false;
                if (raster) {
                    ViewModel& vm = rz.views[view];
                    verified = vm.ok && view_verify(vm, o64, d64, (double)x, (double)(y - view * (unsigned)rz.img_h));
                    if (!verified && vm.all) __hip_atomic_store(&vm.all, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (verified) {
                    cand = key != kRasterEmpty;
                    face = (int32_t)(uint32_t)key;
                } else {
                    cand = n_tris > 0 && (rz.no_tree || hits_top_boxes(nodes, o, d));
                    generic = cand;
                }
            } else if (raster && rz.mode == DRT_GRID_ESTABLISH) {
                // (fused, establishing the cache) a pixel without a target is not traced, but its ray still counts for `all`
                ViewModel& vm = rz.views[view];
                if (vm.all && !(vm.ok && view_verify(vm, load_d3(origin, i), load_d3(dir, i), (double)x, (double)(y - view * (unsigned)rz.img_h))))
                    __hip_atomic_store(&vm.all, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // Outputs of the rays that go no further.  A wave whose 64 rays (one 64-aligned run of a row) are ALL dead -- nine
        // out of ten -- writes its 1536 + 1536 + 192 + 256 + 256 bytes with lane-consecutive 16-byte stores; per-ray stores
        // (8-byte pieces at a 24-byte stride, single mask bytes) move the same bytes at half the rate.
        const bool live = i < n;
        if (!dead_writes) {
        } else if (tile_w > 0 && __ballot(cand || !live) == 0ull) {
            write_dead_row<FUSED>(i - (tid & 63), tid & 63, out_ori, out_dir, mask, face1, face2);
        } else if (live && !cand) {
            face1[i] = -1;
            write_dead(i, out_ori, out_dir, mask, face2);
        }
        // nine out of ten patches are pure background: one barrier (with an OR-reduction) instead of the six of the push
        const int any_cand = __syncthreads_or(cand ? 1 : 0);
        // the group bits are cleared only now, after every wave of the block has read all four of them: the waves of a block
        // drift apart over barrier-free iterations, and a wave arriving late must still see what the early one saw
        if (clear_bit && (tid & 63) == 0) atomicAnd(&rz.zmask[i >> 11], ~clear_bit);
        if (!any_cand) return;
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
        if (slot >= 0) {
            p.r0.idx[slot] = (int32_t)i; store_ray32(p.r0.ray, slot, o, d);
            if (raster) p.r0.face[slot] = generic && rz.pending ? kFacePending : face;
        }
        if (raster && __syncthreads_or(generic ? 1 : 0)) {       // rare: rays outside the grid model
            const int g = block_push(generic, &p.count[3], s_tmp);
            if (g >= 0) rz.gen_list[g] = slot;
        }
    }
}

// Every patch of the sub-batch, one short block each (the kernel keeps no state between patches; a pure stream of the dead
// stores runs fastest this way, tools/ubench/dead_fill.hip).
template <bool FUSED>
__global__ void __launch_bounds__(kPathBlock, 8) k_cull(const Node4Q* __restrict__ nodes, int n_tris, const double* __restrict__ origin, const double* __restrict__ dir,
                                                      const uint8_t* __restrict__ valid, int64_t n, double* __restrict__ out_ori,
                                                      double* __restrict__ out_dir, uint8_t* __restrict__ mask,
                                                      int32_t* __restrict__ face1, int32_t* __restrict__ face2, Pipe p, int tile_w, RasterIn rz,
                                                      bool prefilled /* recycled outputs (drt_outputs_clean) + sparse face ids: dead rays write nothing */) {
    __shared__ CullShared sh;
    cull_patch<FUSED>(blockIdx.x, prefilled, sh, nodes, n_tris, origin, dir, valid, n, out_ori, out_dir, mask, face1, face2, p, tile_w, rz);
}

// DRT_GRID_TRUST: only the patches k_patch_list selected -- those on which a projected triangle wrote a key, and all
// patches of images that are not trusted -- with the dense outputs pre-filled by memsets: the other nine tenths of the
// rays are never touched by a thread.  The hits of a trusted patch go to R0 through an LDS stage (index, float32 ray, face)
// that reserves list space once per ~600 entries: one returning atomic per patch -- ~15 000 per launch on ONE counter word,
// served at ~90 per microsecond -- was most of this kernel's time.
struct StageFace {
    StageMem m;
    int32_t face[kStageCap];
};
template <bool FUSED>
__global__ void __launch_bounds__(kPathBlock) k_cull_listed(const uint32_t* __restrict__ patches, const unsigned* __restrict__ n_patches,
                                                           const Node4Q* __restrict__ nodes, int n_tris, const double* __restrict__ origin, const double* __restrict__ dir,
                                                           const uint8_t* __restrict__ valid, int64_t n, double* __restrict__ out_ori,
                                                           double* __restrict__ out_dir, uint8_t* __restrict__ mask,
                                                           int32_t* __restrict__ face1, int32_t* __restrict__ face2, Pipe p, int tile_w, RasterIn rz,
                                                           PathCtx c, bool direct, bool prefilled, Ray64 park) {
    // `direct` (round 4): a pixel of a trusted image with a key does bounce #1 RIGHT HERE -- the float64 ray is being loaded anyway -- and goes
    // straight to list R1; list R0 then only holds what the untrusted images contribute (cull_patch below), and k_shade1, a pass of its own
    // over 1.5 M entries before, has next to nothing to do (it is not even launched when every image of the call is a verified one).  The
    // number of primary hits is still counted (count[17]: the stage statistics and the bench line's `paths` read it).
    __shared__ CullShared sh;
    __shared__ StageFace st;
    stage_init(st.m);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int vt = ((tid & 63) >> 4) * 64 + (tid >> 6) * 16 + (tid & 15);   // position of this thread's pixel in tile order
    const unsigned ppr = (unsigned)tile_w / 64u;
    const RayList dst = direct ? p.r1 : p.r0;
    unsigned* const dst_count = direct ? &p.count[1] : &p.count[0];
    auto flush = [&]() {                                    // whole block; st.m.n stable
        const unsigned cnt = st.m.n;
        if (tid == 0) st.m.base = cnt ? atomicAdd(dst_count, cnt) : 0u;
        __syncthreads();
        const unsigned base = st.m.base;
        for (unsigned k = tid; k < cnt; k += kPathBlock) { dst.idx[base + k] = st.m.idx[k]; if (!direct) dst.face[base + k] = st.face[k]; }
        for (unsigned k = tid; k < 6u * cnt; k += kPathBlock) dst.ray[6 * (int64_t)base + k] = st.m.ray[k];
        __syncthreads();
        if (tid == 0) st.m.n = 0u;
        __syncthreads();
    };
    unsigned n_hits = 0;                                    // primary hits this thread saw (direct mode: they are not listed anywhere)
    const unsigned count = *n_patches;
    for (unsigned k = blockIdx.x; k < count; k += gridDim.x) {
        const unsigned patch = patches[k];
        const unsigned prow = patch / ppr;
        const unsigned y = 4u * prow + (unsigned)wave, x = 64u * (patch - prow * ppr) + (unsigned)lane;
        const unsigned view = y / (unsigned)rz.img_h;
        if (!(rz.views[view].ok && rz.views[view].all)) {   // an image that is not trusted (block-uniform): the general per-ray path
            cull_patch<FUSED>(patch, true, sh, nodes, n_tris, origin, dir, valid, n, out_ori, out_dir, mask, face1, face2, p, tile_w, rz);
            __syncthreads();
            continue;
        }
        const int64_t i = (int64_t)y * tile_w + x;
        unsigned long long key = kRasterEmpty;
        const unsigned word = rz.zmask[i >> 11], bit = 1u << ((i >> 6) & 31);
        if (word & bit) {
            const int64_t zs = raster_slot(x, y, (unsigned)tile_w);
            key = rz.zbuf[zs];
            if (key != kRasterEmpty) rz.zbuf[zs] = kRasterEmpty;           // consumed: the buffer is empty again for the next call
        }
        const bool cand = key != kRasterEmpty && (!FUSED || valid[i]);
        bool push = cand;                                   // contributes a list entry
        f3 o{0.f, 0.f, 0.f}, d{0.f, 0.f, 1.f};
        if (cand) {
            if (direct) {
                // bounce #1 (what k_shade1 does with an R0 entry): float64 Moeller-Trumbore + refraction on the face the projection pass found
                const int32_t f1 = (int32_t)(uint32_t)key;
                face1[i] = f1;
                d3 v0, v1, v2;
                int32_t vid[3];
                Bounce b;
                load_tri64(c, f1, v0, v1, v2, vid);
                bounce_forward(load_d3(origin, i), load_d3(dir, i), v0, v1, v2, c.ior_ext, c.ior_int, b);
                push = !b.tir;
                o = to_f32(b.new_o); d = to_f32(b.wt);
                // (the float64 refracted ray parked in the ray's rows of the dense outputs -- already zeroed, rewritten by k_shade2 whatever
                // becomes of the path -- so that k_shade2 does bounce #2 only: what k_shade1 does for the one-kernel path)
                if (push && park.o) { store_d3(park.o, i, b.new_o); store_d3(park.d, i, b.wt); }
                if (!push && !FUSED) { if (prefilled) face2[i] = -1; else write_dead(i, out_ori, out_dir, mask, face2); }
                ++n_hits;
            } else {
                o = to_f32(load_d3(origin, i)); d = to_f32(load_d3(dir, i));
            }
        }
        // rank of this thread's entry among the patch's entries, in 16x4-tile order
        sh.flag[vt] = push ? 1 : 0;
        __syncthreads();
        if ((word & bit) && lane == 0) atomicAnd(&rz.zmask[i >> 11], ~bit);  // (after the barrier: every wave has read its bit)
        const bool mine = sh.flag[tid] != 0;
        const unsigned long long bm = __ballot(mine);
        if (lane == 0) sh.tmp[wave] = (unsigned)__popcll(bm);
        __syncthreads();
        unsigned before = 0, tot = 0;
        for (int w = 0; w < kPathWaves; ++w) { const unsigned cw = sh.tmp[w]; if (w < wave) before += cw; tot += cw; }
        sh.slot[tid] = mine ? (int)(before + (unsigned)__popcll(bm & ((1ull << lane) - 1ull))) : -1;
        __syncthreads();
        if (push) {
            const unsigned slot = st.m.n + (unsigned)sh.slot[vt];
            st.m.idx[slot] = (int32_t)i;
            float* e = st.m.ray + 6 * slot;
            e[0] = o.x; e[1] = o.y; e[2] = o.z; e[3] = d.x; e[4] = d.y; e[5] = d.z;
            st.face[slot] = (int32_t)(uint32_t)key;
        }
        __syncthreads();
        if (tid == 0) st.m.n += tot;
        __syncthreads();
        if (st.m.n > kStageCap - kPathBlock) flush();
    }
    flush();
    if (direct) {
        unsigned h = n_hits;
        for (int off = 32; off >= 1; off >>= 1) h += __shfl_xor(h, off);
        if (lane == 0 && h) atomicAdd(&p.count[17], h);
    }
}

// Which patches k_cull_listed has to visit: one thread per patch.
__global__ void __launch_bounds__(kPathBlock) k_patch_list(unsigned n_patches, int tile_w, RasterIn rz, uint32_t* __restrict__ list, unsigned* count) {
    __shared__ unsigned s_tmp[kPathWaves + 1];
    const unsigned patch = blockIdx.x * kPathBlock + threadIdx.x;
    bool take = false;
    if (patch < n_patches) {
        const unsigned ppr = (unsigned)tile_w / 64u, prow = patch / ppr;
        const unsigned view = (4u * prow) / (unsigned)rz.img_h;
        take = true;
        if (rz.views[view].ok && rz.views[view].all) {
            const int64_t row0 = (int64_t)(4u * prow) * tile_w + 64u * (patch - prow * ppr);
            unsigned any = 0;
            for (int r = 0; r < 4; ++r) {
                const int64_t g = (row0 + (int64_t)r * tile_w) >> 6;
                any |= rz.zmask[g >> 5] >> (g & 31);
            }
            take = (any & 1u) != 0;
        }
    }
    const int slot = block_push(take, count, s_tmp);
    if (slot >= 0) list[slot] = patch;
}

// R0 -> R1: primary hit -> float64 bounce #1 -> refracted ray
template <bool FUSED>
__global__ void __launch_bounds__(kPathBlock) k_shade1(PathCtx c, const double* __restrict__ origin, const double* __restrict__ dir,
                                                        double* __restrict__ out_ori, double* __restrict__ out_dir, uint8_t* __restrict__ mask,
                                                        int32_t* __restrict__ face1, int32_t* __restrict__ face2, Pipe p,
                                                        bool prefilled /* the dense outputs hold (or are being filled with) the dead values */, Ray64 r64) {
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
            if (f1 != kFacePending) face1[i] = f1;
            if (f1 >= 0) {
                d3 v0, v1, v2;
                int32_t vid[3];
                Bounce b;
                load_tri64(c, f1, v0, v1, v2, vid);
                bounce_forward(load_d3(origin, i), load_d3(dir, i), v0, v1, v2, c.ior_ext, c.ior_int, b);
                ok = !b.tir;
                o2 = to_f32(b.new_o); d2 = to_f32(b.wt);
                if (ok && r64.o) { store_d3(r64.o, i, b.new_o); store_d3(r64.d, i, b.wt); }
            }
            if (!ok && !FUSED && f1 != kFacePending) { if (prefilled) face2[i] = -1; else write_dead(i, out_ori, out_dir, mask, face2); }
        }
        stage_push(stage, ok, (int32_t)i, o2, d2, p.r1, &p.count[1]);
    }
    stage_flush(stage, p.r1, &p.count[1]);
}

// DRT_GRID_ALL_VERIFIED promised that no ray would need the tree for its primary hit, so nothing traced the rays outside the grid
// before k_shade1 -- but an image can still lose that status INSIDE a call: k_raster demotes it when a triangle reaches the camera
// plane (vertices move every iteration) or its large-triangle list overflows, k_check_views when the rays changed behind the cache's
// back.  k_cull_listed then lists every ray of that image in gen_list with face = kFacePending; this kernel -- a few blocks, one thread
// per ray, normally nothing to do -- finds their primary hits in the tree once the build has been waited for, does bounce #1 like
// k_shade1 and appends the refracted rays to R1, in front of the second traversal.  Rare path: simple, not fast.
template <bool FUSED>
__global__ void __launch_bounds__(kTraceBlock) k_gen_late(PathCtx c, const double* __restrict__ origin, const double* __restrict__ dir,
                                                          double* __restrict__ out_ori, double* __restrict__ out_dir, uint8_t* __restrict__ mask,
                                                          int32_t* __restrict__ face1, int32_t* __restrict__ face2, Pipe p,
                                                          const int32_t* __restrict__ gen_list, bool prefilled, Ray64 r64) {
    __shared__ int32_t lds[kStackFast][kTraceBlock];
    const unsigned n = p.count[3];
    if (n == 0) return;
    Stack st = make_stack(lds, c.tc);
    for (unsigned k = blockIdx.x * kTraceBlock + threadIdx.x; k < n; k += gridDim.x * kTraceBlock) {
        const int32_t slot = gen_list[k];
        if (p.r0.face[slot] != kFacePending) continue;
        const int64_t i = p.r0.idx[slot];
        const float* e = p.r0.ray + 6 * (int64_t)slot;
        const Hit h = traverse<false>(c.tc.nodes, c.tc.tris, c.tc.n_tris, f3{e[0], e[1], e[2]}, f3{e[3], e[4], e[5]}, st);
        p.r0.face[slot] = h.face;
        face1[i] = h.face;
        bool ok = false;
        if (h.face >= 0) {
            d3 v0, v1, v2;
            int32_t vid[3];
            Bounce b;
            load_tri64(c, h.face, v0, v1, v2, vid);
            bounce_forward(load_d3(origin, i), load_d3(dir, i), v0, v1, v2, c.ior_ext, c.ior_int, b);
            ok = !b.tir;
            if (ok) {
                const unsigned s1 = atomicAdd(&p.count[1], 1u);
                p.r1.idx[s1] = (int32_t)i;
                store_ray32(p.r1.ray, (int)s1, to_f32(b.new_o), to_f32(b.wt));
                if (r64.o) { store_d3(r64.o, i, b.new_o); store_d3(r64.d, i, b.wt); }
            }
        }
        if (!ok && !FUSED) { if (prefilled) face2[i] = -1; else write_dead(i, out_ori, out_dir, mask, face2); }
    }
}

// R1 -> R2: second hit -> float64 bounces #1 and #2 -> provisional exit ray
template <bool FUSED>
__global__ void __launch_bounds__(kPathBlock) k_shade2(PathCtx c, const double* __restrict__ origin, const double* __restrict__ dir,
                                                        double* __restrict__ out_ori, double* __restrict__ out_dir, uint8_t* __restrict__ mask,
                                                        const int32_t* __restrict__ face1, int32_t* __restrict__ face2, Pipe p,
                                                        bool parked /* rows i of out_ori / out_dir hold the float64 refracted ray of R1's entries (k_cull_listed direct) */) {
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
                d3 o2, d2;
                if (parked) {
                    o2 = load_d3(out_ori, i); d2 = load_d3(out_dir, i);
                } else {
                    load_tri64(c, face1[i], v0, v1, v2, vid);
                    bounce_forward(load_d3(origin, i), load_d3(dir, i), v0, v1, v2, c.ior_ext, c.ior_int, b);
                    o2 = b.new_o; d2 = b.wt;
                }
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

// ---- the back half of the path as ONE persistent kernel (small sub-batches) ----------------------------------------------
//
// Below ~2^24 camera rays a sub-batch has fewer rays in flight than the chip has lanes: every k_trace launch is then as long as
// its slowest wavefront (one 64-ray group per wave, ~0.7 us per dependent load -> slab test -> push step, 150-230 steps), the
// staged pipeline pays that tail twice (refracted rays, exit rays) plus four dependent launches in between, and the SIMDs sit
// idle under both tails.  k_path runs R1 -> outputs in one launch: a LANE carries its ray through
//     T1: closest hit of the refracted ray -> S2: float64 bounce #2, provisional outputs
//     -> T2: occlusion test of the exit ray -> finish (survivor -> list of valid rays; occluded -> dead values)
// and takes another R1 entry when it is through, so the occlusion rays of early finishers are traced under the tail of the
// refracted rays of the others.  Rays are handed out DYNAMICALLY (a cursor per XCD part of the list, bumped by as many entries as a
// wave has idle lanes), so that no wave is left with a second whole group when the others are done.  The traversal phases are
// k_trace's ("while-while": lanes at inner nodes descend together, lanes at leaves wait for a common triangle step; LDS-only stack
// with a redo list); the float64 stages run as wave phases in between, entered when `shade_min` lanes wait for one (or nothing
// else can run).  T2 uses the closest-hit child order (one code path for all traversing lanes) and stops at its first hit.
// Bounce #1 stays a kernel of its own (k_shade1): it needs no tree, so it runs while the asynchronous build finishes, and it parks
// the float64 refracted ray of every R1 entry in `ray64` (rows of the dense outputs themselves, or a side buffer in the fused form)
// so that S2 is ONE bounce.  Results are those of the staged kernels bit for bit (same device functions, same final writes).
// ~2x the registers of k_trace (4 waves per SIMD): chosen for small sub-batches only.
enum : int { kLaneIdle = 0, kLaneT1 = 2, kLaneT1D = 3, kLaneT2 = 4, kLaneT2D = 5 };   // odd = waits for a float64 stage
constexpr int kVBuf = 128;            // finished rays staged per wave before one list reservation

template <class STACK>
__device__ __forceinline__ bool path_leaf(const TriRec* __restrict__ tris, TravState& s, STACK& st, bool any) {
    if (s.cur == kEmptyChild) return trav_pop(s, st);
    const int32_t ref = ~s.cur;
    const int first = ref >> kLeafBits, count = (ref & (kLeafMax - 1)) + 1;
    for (int j = 0; j < count; ++j) {
        const F4* tp = reinterpret_cast<const F4*>(tris + first + j);
        const F4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
        float t;
        if (tri_hit(s.o, s.d, f3{p0.x, p0.y, p0.z}, f3{p1.x, p1.y, p1.z}, f3{p2.x, p2.y, p2.z}, p1.w, t)) {     // (this kernel has the registers)
            int32_t face;
            memcpy(&face, &p0.w, 4);
            if (any) { s.best_t = t; s.best_face = face; return true; }
            if (t < s.best_t || (t == s.best_t && face < s.best_face)) { s.best_t = t; s.best_face = face; }
        }
    }
    return trav_pop(s, st);
}

template <bool FUSED>
__global__ void __launch_bounds__(kPathBlock, 4) k_path(PathCtx c, Ray64 r64, double* __restrict__ out_ori, double* __restrict__ out_dir, uint8_t* __restrict__ mask,
                                                        int32_t* __restrict__ face2, Pipe p, unsigned* cursor /* [8], zero */, int64_t chunk_base,
                                                        int32_t* __restrict__ valid_idx, int refill_min, int inner_min, int shade_min,
                                                        unsigned long long* stats, bool count_items) {
    __shared__ int32_t lds[kStackFast + 1 + kGuardRows][kPathBlock];
    __shared__ int32_t s_vbuf[kPathWaves][kVBuf];
    FastStack st;
    st.base = (drt::FastPtr)&lds[0][threadIdx.x]; st.stride = kPathBlock; st.depth = kStackFast - 3; st.reset(); st.overflow = false;
#if defined(DRT_CHECK)
    for (int g = 0; g < kGuardRows; ++g) lds[kStackFast + 1 + g][threadIdx.x] = kGuardPoison;
#endif
    const unsigned n = p.count[1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long lower = (1ull << lane) - 1ull;
    // R1 (tile order) in 8 contiguous parts, one per XCD (block b runs on XCD b % 8; each XCD has its own L2).  Within a part every wave
    // starts with the 64 entries at 64 x its number; what is left is handed out through the part's cursor.
    constexpr unsigned kXcd = 8;
    const bool split = gridDim.x % kXcd == 0 && n >= 4096u * kXcd;
    const unsigned xcd = split ? blockIdx.x % kXcd : 0u, parts = split ? kXcd : 1u;
    const unsigned wave = (split ? blockIdx.x / kXcd : blockIdx.x) * kPathWaves + (threadIdx.x >> 6);
    const unsigned n_waves = (split ? gridDim.x / kXcd : gridDim.x) * kPathWaves;
    const unsigned part_lo = (unsigned)((unsigned long long)n * xcd / parts), part_hi = (unsigned)((unsigned long long)n * (xcd + 1) / parts);
    // the first share of a wave is static; it shrinks below 64 when the list is short, so that every wave of the grid gets some
    const unsigned part_n = part_hi - part_lo;
    unsigned first = (part_n + n_waves - 1) / n_waves;
    first = first > 64u ? 64u : (first < 8u ? 8u : first);
    unsigned my_next = part_lo + wave * first;         // next entry of the static share
    unsigned my_end = my_next + first;
    if (my_end > part_hi) my_end = part_hi;
    if (my_next > part_hi) my_next = part_hi;
    const unsigned dyn_lo = part_lo + n_waves * first; // entries beyond the static shares
    bool dry = dyn_lo >= part_hi;                      // nothing (left) to hand out dynamically
    // finished rays: the caller's list of valid rays (global index) or, fused, list R2 for k_loss_bwd_fused (index; face < 0 = "not occluded")
    int32_t* const v_out = FUSED ? p.r2.idx : valid_idx;
    unsigned* const v_count = FUSED ? &p.count[2] : p.valid;
    unsigned vn = 0;                                   // entries staged in s_vbuf[wv] (wave-uniform)
    auto v_flush = [&]() {
        if (vn == 0) return;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(v_count, vn);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        for (unsigned k = lane; k < vn; k += 64u) {
            v_out[base + k] = s_vbuf[wv][k];
            if (FUSED) p.r2.face[base + k] = -1;
        }
        vn = 0;
    };
    int stage = kLaneIdle;
    int32_t rk = 0, ri = 0;                            // R1 slot and ray index of this lane's ray
    TravState s;
    s.cur = 0; s.best_face = -1; s.best_t = 0.f;
    unsigned long long wave_steps = 0, lane_steps = 0, leaf_steps = 0;
    unsigned n_t2 = 0;
    for (;;) {
        // ---- refill: idle lanes take R1 entries -- the rest of the static share first, then from the part's cursor
        const unsigned long long idle = __ballot(stage == kLaneIdle);
        if (idle != 0 && (my_next < my_end || !dry) && ((int)__popcll(idle) >= refill_min || idle == ~0ull)) {
            const unsigned want = (unsigned)__popcll(idle);
            unsigned base, got;
            if (my_next < my_end) {
                base = my_next;
                got = my_end - my_next < want ? my_end - my_next : want;
                my_next += got;
            } else {
                unsigned b0 = 0;
                if (lane == 0) b0 = atomicAdd(&cursor[xcd], want);
                b0 = (unsigned)__builtin_amdgcn_readfirstlane((int)b0);
                base = dyn_lo + b0;
                got = base >= part_hi ? 0u : (part_hi - base < want ? part_hi - base : want);
                if (got < want) dry = true;
            }
            const unsigned r = (unsigned)__popcll(idle & lower);
            if (stage == kLaneIdle && r < got) {
                const unsigned k = base + r;
                rk = (int32_t)k;
                ri = p.r1.idx[k];
                const float* e = p.r1.ray + 6 * (int64_t)k;
                trav_init(s, st, f3{e[0], e[1], e[2]}, f3{e[3], e[4], e[5]});
                st.overflow = false;
                stage = kLaneT1;
            }
        }
        if (__ballot(stage != kLaneIdle) == 0) {
            if (my_next < my_end || !dry) continue;    // (all lanes idle: the refill above runs whatever refill_min says)
            break;
        }
        // ---- float64 stages, when enough lanes wait for one or nothing can be traversed meanwhile
        {
            const unsigned long long waiting = __ballot(stage & 1), walking = __ballot(stage != kLaneIdle && !(stage & 1));
            if (waiting != 0 && ((int)__popcll(waiting) >= shade_min || walking == 0)) {
                // finish: the occlusion test of the exit ray is in
                const bool fin = stage == kLaneT2D;
                const bool keep = fin && s.best_face < 0;
                if (fin && !keep) { if (FUSED) face2[ri] = -1; else write_dead(ri, out_ori, out_dir, mask, face2); }
                if (v_out) {
                    const unsigned long long km = __ballot(keep);
                    if (km != 0) {
                        if (keep) s_vbuf[wv][vn + (unsigned)__popcll(km & lower)] = FUSED ? ri : (int32_t)(chunk_base + ri);
                        vn += (unsigned)__popcll(km);
                        __builtin_amdgcn_wave_barrier();
                        if (vn > kVBuf - 64) v_flush();
                    }
                }
                if (fin) stage = kLaneIdle;
                if (stage == kLaneT1D) {
                    // bounce #2 (k_shade2, which recomputes bounce #1 to the same bits): the second hit is in
                    const int32_t f2 = s.best_face;
                    bool ok = false;
                    if (f2 >= 0) {
                        d3 v0, v1, v2;
                        int32_t vid[3];
                        Bounce b;
                        const d3 o2 = load_d3(r64.o, ri), d2 = load_d3(r64.d, ri);
                        load_tri64(c, f2, v0, v1, v2, vid);
                        bounce_forward(o2, d2, v0, v1, v2, c.ior_ext, c.ior_int, b);
                        ok = !b.tir;
                        if (ok) {
                            face2[ri] = f2;
                            if (!FUSED) {
                                store_d3(out_ori, ri, b.new_o);
                                store_d3(out_dir, ri, b.wt);
                                mask[3 * (int64_t)ri] = 1; mask[3 * (int64_t)ri + 1] = 1; mask[3 * (int64_t)ri + 2] = 1;
                            }
                            trav_init(s, st, to_f32(b.new_o), to_f32(b.wt));
                            st.overflow = false;
                            stage = kLaneT2;
                            ++n_t2;
                        }
                    }
                    if (!ok) {
                        if (FUSED) face2[ri] = -1; else write_dead(ri, out_ori, out_dir, mask, face2);
                        stage = kLaneIdle;
                    }
                }
            }
        }
        // ---- traversal phases (k_trace's), for the lanes in T1 / T2
        for (;;) {
            const bool walking = stage != kLaneIdle && !(stage & 1);
            const bool at_inner = walking && s.cur >= 0;
            const unsigned long long mi = __ballot(at_inner);
            if (mi == 0) break;
            if ((int)__popcll(mi) < inner_min && __ballot(walking && s.cur < 0) != 0) break;
            ++wave_steps;
            lane_steps += (unsigned long long)__popcll(mi);
            if (at_inner) {
                const bool done = trav_inner<false>(c.tc.nodes, s, st);
                if (st.overflow) {                 // LDS stack exhausted (rare): the rest of this ray's path is redone by k_path_redo
                    p.redo[atomicAdd(&p.count[5], 1u)] = rk;       // (count[5]: the second traversal's redo counter -- count[4] may hold the first one's)
                    stage = kLaneIdle;
                } else if (done) {
                    stage |= 1;
                }
            }
        }
        {
            const bool at_leaf = stage != kLaneIdle && !(stage & 1) && s.cur < 0;
            const unsigned long long ml = __ballot(at_leaf);
            if (ml != 0) {
                ++wave_steps; ++leaf_steps;
                lane_steps += (unsigned long long)__popcll(ml);
                if (at_leaf && path_leaf(c.tc.tris, s, st, stage == kLaneT2)) stage |= 1;
            }
        }
    }
    v_flush();
#if defined(DRT_CHECK)
    for (int g = 0; g < kGuardRows; ++g) DRT_DEV_ASSERT(lds[kStackFast + 1 + g][threadIdx.x] == kGuardPoison, drt::kCheckGuardRow);
#endif
    if (count_items) {                                  // (profiling only: the size the staged pipeline's list R2 would have)
        const unsigned t2 = (unsigned)wave_sum((double)n_t2);
        if (lane == 0 && t2) atomicAdd(&p.count[6], t2);
    }
    if (stats && lane == 0 && wave_steps) {
        atomicAdd(stats + 0, wave_steps);
        atomicAdd(stats + 1, lane_steps);
        atomicAdd(stats + 2, leaf_steps);
        atomicMax(stats + 3, wave_steps);
    }
}

// The rays whose traversal overflowed the LDS-only stack inside k_path: the rest of the path again from the R1 entry, one thread per
// ray, spilling stack.  Idempotent with whatever the abandoned lane had written (provisional outputs are rewritten or zeroed).  The
// float64 refracted ray is recomputed from the camera ray and the first face (like k_shade2 does): where it was parked -- rows of the
// dense outputs in the drop-in form -- may already hold the provisional EXIT ray of a lane that gave up during the occlusion test.
template <bool FUSED>
__global__ void __launch_bounds__(kTraceBlock) k_path_redo(PathCtx c, const double* __restrict__ origin, const double* __restrict__ dir, const int32_t* __restrict__ face1,
                                                           double* __restrict__ out_ori, double* __restrict__ out_dir, uint8_t* __restrict__ mask,
                                                           int32_t* __restrict__ face2, Pipe p, int64_t chunk_base, int32_t* __restrict__ valid_idx) {
    __shared__ int32_t lds[kStackFast][kTraceBlock];
    const unsigned n = p.count[5];
    if (n == 0) return;
    Stack st = make_stack(lds, c.tc);
    for (unsigned q = blockIdx.x * kTraceBlock + threadIdx.x; q < n; q += gridDim.x * kTraceBlock) {
        const int32_t rk = p.redo[q];
        const int64_t i = p.r1.idx[rk];
        const float* e = p.r1.ray + 6 * (int64_t)rk;
        const int32_t f2 = traverse<false>(c.tc.nodes, c.tc.tris, c.tc.n_tris, f3{e[0], e[1], e[2]}, f3{e[3], e[4], e[5]}, st).face;
        bool alive = false;
        if (f2 >= 0) {
            d3 v0, v1, v2;
            int32_t vid[3];
            Bounce b;
            load_tri64(c, face1[i], v0, v1, v2, vid);
            bounce_forward(load_d3(origin, i), load_d3(dir, i), v0, v1, v2, c.ior_ext, c.ior_int, b);
            const d3 o2 = b.new_o, d2 = b.wt;
            load_tri64(c, f2, v0, v1, v2, vid);
            bounce_forward(o2, d2, v0, v1, v2, c.ior_ext, c.ior_int, b);
            if (!b.tir && traverse<true>(c.tc.nodes, c.tc.tris, c.tc.n_tris, to_f32(b.new_o), to_f32(b.wt), st).face < 0) {
                alive = true;
                face2[i] = f2;
                if (!FUSED) {
                    store_d3(out_ori, i, b.new_o);
                    store_d3(out_dir, i, b.wt);
                    mask[3 * i] = 1; mask[3 * i + 1] = 1; mask[3 * i + 2] = 1;
                }
            }
        }
        if (!alive) {
            if (FUSED) face2[i] = -1; else write_dead(i, out_ori, out_dir, mask, face2);
        } else if (FUSED) {
            const unsigned slot = atomicAdd(&p.count[2], 1u);
            p.r2.idx[slot] = (int32_t)i; p.r2.face[slot] = -1;
        } else if (valid_idx) {
            valid_idx[atomicAdd(p.valid, 1u)] = (int32_t)(chunk_base + i);
        }
    }
}

__global__ void k_store_count(const unsigned* __restrict__ count, int64_t* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = (int64_t)*count;
}

// Behind the join of a call cut into sub-batches: segment j of the list of completed paths (written at offset `src` = the index of the
// sub-batch's first ray) moves down behind the segments before it.  dst <= src always.  The list is a SET (within a segment its order is
// that of the atomics that filled it), so when the two ranges overlap (more than half of the rays of the sub-batches in front complete)
// only the segment's last src - dst entries move, into the hole [dst, src) in front of it: source [dst + len, src + len) and
// destination are disjoint then (dst + len > src), and every block copies its share -- round 4 walked the whole segment with ONE block
// in ascending chunks, tens of milliseconds for an object that fills the frame.  The launch for the last segment also stores the total.
__global__ void __launch_bounds__(256) k_join_lists(int32_t* list, const unsigned* __restrict__ counts, int j, int64_t src, int last,
                                                    unsigned* total_u, int64_t* total_i64) {
    int64_t dst = 0;
    for (int k = 0; k < j; ++k) dst += counts[k];
    const int64_t len = counts[j];
    if (last && blockIdx.x == 0 && threadIdx.x == 0) { *total_u = (unsigned)(dst + len); if (total_i64) *total_i64 = dst + len; }
    if (dst == src || len == 0) return;
    const bool overlap = dst + len > src;
    const int64_t n_move = overlap ? src - dst : len, from = overlap ? dst + len : src;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n_move; k += (int64_t)gridDim.x * blockDim.x) list[dst + k] = list[from + k];
}

// drt_outputs_clean: the rows a render call left non-zero are exactly its list of completed paths; zero them again (51 B per listed row
// instead of 51 B per ray of the call).
__global__ void __launch_bounds__(256) k_unwrite_rows(double* __restrict__ out_ori, double* __restrict__ out_dir, uint8_t* __restrict__ mask,
                                                      const int32_t* __restrict__ rows, const int64_t* __restrict__ n_rows, int64_t n_rays) {
    const int64_t n = *n_rows;
    const d3 z{0.0, 0.0, 0.0};
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = rows[k];
        if (i < 0 || i >= n_rays) continue;
        store_d3(out_ori, i, z);
        store_d3(out_dir, i, z);
        mask[3 * i] = 0; mask[3 * i + 1] = 0; mask[3 * i + 2] = 0;
    }
}


// Vertex-gradient accumulation through an LDS hash table.  The float64 scatter is bound by the
// chip's atomic rate (measured 22.6 G global_atomic_add_f64 per second, tools/ubench/atomic_scope.hip,
// independent of scope or per-XCD privatisation), and neighbouring rays hit neighbouring triangles
// that share vertices: a block first sums its contributions per vertex in LDS (ds_add_f64 after a
// compare-and-swap probe on the key) and then issues three global atomics per DISTINCT vertex.
#ifndef DRT_HASH_BITS
#define DRT_HASH_BITS 11
#endif
#ifndef DRT_BWD_BATCH
#define DRT_BWD_BATCH 1024
#endif
#ifndef DRT_BWD_BPC
#define DRT_BWD_BPC 2
#endif
constexpr int kHashBits = DRT_HASH_BITS, kHashSize = 1 << kHashBits;      // 2048 slots: 8 KB keys + 48 KB sums
constexpr int kBwdBatch = DRT_BWD_BATCH;                                 // rays per table fill (6 vertex refs each)

struct HashAdd3 {
    int32_t* keys;      // LDS [kHashSize]
    double* sums;       // LDS [kHashSize * 3]
    double* g;          // global fallback / final target
    __device__ __forceinline__ void operator()(int32_t v, d3 a) const {
        // slot = low bits of the vertex id: neighbouring slots then hold neighbouring ids, and the flush below walks the
        // table as flat doubles, so that a wave's atomics fall on runs of consecutive addresses -- see hash_flush
        unsigned h = (unsigned)v & (kHashSize - 1);
        // (ids come in runs of neighbours, and so do the occupied slots: a colliding id leaves the run in one odd stride that
        // depends on its high bits instead of walking through it slot by slot)
        const unsigned step = ((((unsigned)v >> kHashBits) << 1) + 97u) | 1u;
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
            h = (h + step) & (kHashSize - 1);
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
    // The chip's atomic units work per cache-line REQUEST, not per lane (tools/ubench/atomic_pattern.hip: 23 G float64 atomics/s on
    // random addresses, 57 G/s when the 64 lanes of an instruction cover consecutive 24-byte rows, 141 G/s on 64 consecutive doubles).
    // With the table indexed by the low bits of the vertex id and the flush walking it as flat doubles (lane j -> component j % 3 of
    // slot j / 3), the vertices of a batch -- a compact patch of the surface, numbered by the mesher with some coherence -- give runs of
    // consecutive addresses: the backward kernel 0.33 -> 0.27 ms against a multiplicative hash flushed slot by slot.
    for (int j = threadIdx.x; j < 3 * kHashSize; j += blockDim.x) {
        const int32_t v = keys[j / 3];
        if (v >= 0) unsafeAtomicAdd(g + 3 * (int64_t)v + (j % 3), sums[j]);
    }
    __syncthreads();
}

// The same table for the deterministic mode (drt_fixed.h): the slots hold 128-bit fixed-point sums -- low and high words as two 64-bit
// integer LDS atomics, the carry owned by the addition that wrapped the low word -- and the flush adds each occupied slot to its FxCell in
// memory with two more.  Integer sums are exact, so WHICH contributions meet in a table (batch composition, probe order, overflow to
// memory) cannot change a bit of the result; the table only cuts the global atomics from 36 per path to 6 per distinct vertex and batch
// (measured round 6, 72 x 1024^2: without it the loss + gradient pass took 4.4 ms instead of 0.3).  Same LDS footprint as the float64
// table: half the slots, twice the bytes per sum.
constexpr int kFxHashSize = kHashSize / 2;
struct FxHashAdd3 {
    int32_t* keys;                    // LDS [kFxHashSize]
    unsigned long long* lo;           // LDS [3 * kFxHashSize]
    unsigned long long* hi;           // LDS [3 * kFxHashSize]
    double* g;                        // the FxCell array in memory
    __device__ __forceinline__ void operator()(int32_t v, d3 a) const {
        Fx128 t[3];
        const double comp[3] = {a.x, a.y, a.z};
        FxCell* cell = reinterpret_cast<FxCell*>(g) + 3 * (int64_t)v;
        bool any = false;
        for (int c = 0; c < 3; ++c) {
            const uint32_t f = fx_from_double(comp[c], t[c]);
            if (f) fx_atomic_add(cell + c, t[c], f);          // (rare: sticky flags go straight to memory; t[c] is zero then)
            any |= (t[c].hi | (int64_t)t[c].lo) != 0;
        }
        if (!any) return;
        unsigned h = (unsigned)v & (kFxHashSize - 1);
        const unsigned step = ((((unsigned)v >> (kHashBits - 1)) << 1) + 97u) | 1u;
#pragma unroll 1
        for (int probe = 0; probe < 24; ++probe) {
            int32_t k = keys[h];
            if (k == -1) k = atomicCAS(&keys[h], -1, v);
            if (k == -1 || k == v) {
                for (int c = 0; c < 3; ++c) {
                    if (!(t[c].hi | (int64_t)t[c].lo)) continue;
                    unsigned long long carry = 0;
                    if (t[c].lo) {
                        const unsigned long long old = atomicAdd(&lo[3 * h + c], (unsigned long long)t[c].lo);
                        carry = old + (unsigned long long)t[c].lo < old ? 1ull : 0ull;
                    }
                    const unsigned long long add_hi = (unsigned long long)t[c].hi + carry;
                    if (add_hi) atomicAdd(&hi[3 * h + c], add_hi);
                }
                return;
            }
            h = (h + step) & (kFxHashSize - 1);
        }
        for (int c = 0; c < 3; ++c) fx_atomic_add(cell + c, t[c], 0u);     // table crowded: straight to memory
    }
};
__device__ __forceinline__ void fx_hash_clear(int32_t* keys, unsigned long long* lo, unsigned long long* hi) {
    for (int i = threadIdx.x; i < kFxHashSize; i += blockDim.x) keys[i] = -1;
    for (int i = threadIdx.x; i < 3 * kFxHashSize; i += blockDim.x) { lo[i] = 0ull; hi[i] = 0ull; }
    __syncthreads();
}
__device__ __forceinline__ void fx_hash_flush(int32_t* keys, unsigned long long* lo, unsigned long long* hi, double* g) {
    __syncthreads();
    for (int j = threadIdx.x; j < 3 * kFxHashSize; j += blockDim.x) {
        const int32_t v = keys[j / 3];
        if (v >= 0) fx_atomic_add(reinterpret_cast<FxCell*>(g) + 3 * (int64_t)v + (j % 3), Fx128{(int64_t)hi[j], (uint64_t)lo[j]}, 0u);
    }
    __syncthreads();
}

// The vertex-gradient sink of the path kernels in the two accumulation modes (drt_device.h GradAdd3) over ONE block of LDS (keys [kHashSize]
// int32, sums [3 kHashSize] float64 -- or, deterministic, half as many slots of two 64-bit words): clear() before a batch, flush() after it.
template <bool DET>
struct PathSink {
    int32_t* keys;
    double* sums;
    double* g;
    __device__ __forceinline__ unsigned long long* lo() const { return reinterpret_cast<unsigned long long*>(sums); }
    __device__ __forceinline__ unsigned long long* hi() const { return reinterpret_cast<unsigned long long*>(sums) + 3 * kFxHashSize; }
    __device__ __forceinline__ void operator()(int32_t v, d3 a) const {
        if (DET) FxHashAdd3{keys, lo(), hi(), g}(v, a); else HashAdd3{keys, sums, g}(v, a);
    }
    __device__ __forceinline__ void clear() const { if (DET) fx_hash_clear(keys, lo(), hi()); else hash_clear(keys, sums); }
    __device__ __forceinline__ void flush() const { if (DET) fx_hash_flush(keys, lo(), hi(), g); else hash_flush(keys, sums, g); }
};

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
template <bool DET>
__global__ void __launch_bounds__(256) k_render_bwd(PathCtx c, const double* __restrict__ origin, const double* __restrict__ dir,
                                                    const int32_t* __restrict__ face1, const int32_t* __restrict__ face2,
                                                    const double* __restrict__ g_out_ori, const double* __restrict__ g_out_dir,
                                                    double* grad_verts, const int32_t* __restrict__ list, const unsigned* __restrict__ n_u32,
                                                    const int64_t* __restrict__ n_i64) {
    __shared__ int32_t hkeys[kHashSize];
    __shared__ double hsums[3 * kHashSize];
    const int64_t n = n_i64 ? *n_i64 : (int64_t)*n_u32;
    const PathSink<DET> add{hkeys, hsums, grad_verts};
    for (int64_t base = blockIdx.x * (int64_t)kBwdBatch; base < n; base += (int64_t)gridDim.x * kBwdBatch) {
        add.clear();
        const int64_t end = base + kBwdBatch < n ? base + kBwdBatch : n;
        for (int64_t k = base + threadIdx.x; k < end; k += blockDim.x) {
            const int64_t i = list[k];
            const d3 z{0.0, 0.0, 0.0};
            const d3 g_ori = g_out_ori ? load_d3(g_out_ori, i) : z;
            const d3 g_dir = g_out_dir ? load_d3(g_out_dir, i) : z;
            path_recompute_backward(c, load_d3(origin, i), load_d3(dir, i), face1[i], face2[i], g_ori, g_dir, add);
        }
        add.flush();
    }
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
template <bool DET>
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
    LossAcc<DET> acc;
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
            if (on[k]) acc.add(ray_loss_term(load_d3(out_ori, i0 + k), load_d3(out_dir, i0 + k), load_d3(screen_pixel, i0 + k), g[k]));
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
    acc.flush(loss);
}

// ray_loss over the forward's list of completed paths (the rays with mask = 1) instead of a pass over all N rays: loss and
// the list of contributing rows (valid target) for drt_render_backward_ray_loss.  O(valid paths), a few per cent of N.
template <bool DET>
__global__ void __launch_bounds__(256) k_ray_loss_listed(const double* __restrict__ out_ori, const double* __restrict__ out_dir,
                                                         const double* __restrict__ screen_pixel, const uint8_t* __restrict__ valid,
                                                         const int32_t* __restrict__ paths, const int64_t* __restrict__ n_paths, double* loss,
                                                         int32_t* __restrict__ rows, unsigned* n_rows) {
    __shared__ unsigned s_tmp[kPathWaves + 1];
    const int64_t n = *n_paths;
    LossAcc<DET> acc;
    for (int64_t base = blockIdx.x * 256ll; base < n; base += (int64_t)gridDim.x * 256) {
        const int64_t k = base + threadIdx.x;
        bool on = false;
        int32_t i = 0;
        if (k < n) {
            i = paths[k];
            on = valid[i] != 0;
            if (on) { d3 g; acc.add(ray_loss_term(load_d3(out_ori, i), load_d3(out_dir, i), load_d3(screen_pixel, i), g)); }
        }
        if (rows) {
            const int slot = block_push(on, n_rows, s_tmp);
            if (slot >= 0) rows[slot] = i;
        }
    }
    acc.flush(loss);
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
template <bool DET>
__global__ void __launch_bounds__(256) k_loss_bwd_fused(PathCtx c, const double* __restrict__ origin, const double* __restrict__ dir,
                                                        const double* __restrict__ screen_pixel, const int32_t* __restrict__ face1,
                                                        const int32_t* __restrict__ face2, Pipe p, double* loss, double* grad_verts,
                                                        unsigned long long* n_valid) {
    __shared__ int32_t hkeys[kHashSize];
    __shared__ double hsums[3 * kHashSize];
    const unsigned n2 = p.count[2];
    const PathSink<DET> add{hkeys, hsums, grad_verts};
    LossAcc<DET> acc;
    unsigned cnt = 0;
    for (unsigned base = blockIdx.x * kBwdBatch; base < n2; base += gridDim.x * kBwdBatch) {
        add.clear();
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
            acc.add(ray_loss_term(b2.new_o, b2.wt, load_d3(screen_pixel, i), g_dir));
            ++cnt;
            const d3 z{0.0, 0.0, 0.0};
            d3 ga = z, gb = z, gc = z, g_o, g_d, g_o0, g_d0;
            bounce_backward(b2, z, g_dir, ga, gb, gc, g_o, g_d);
            add(vid2[0], ga); add(vid2[1], gb); add(vid2[2], gc);
            ga = z; gb = z; gc = z;
            bounce_backward(b1, g_o, g_d, ga, gb, gc, g_o0, g_d0);
            add(vid1[0], ga); add(vid1[1], gb); add(vid1[2], gc);
        }
        add.flush();
    }
    acc.flush(loss);
    if (n_valid && cnt) atomicAdd(n_valid, (unsigned long long)cnt);
}


// Backward of render_transparent + ray_loss for the rows ray_loss reported as contributing (drt_ray_loss's list):
// the loss gradient d loss / d out_dir = 2 (out_dir - target) * scale is recomputed from the path (bit-identical to the
// stored outputs: same code) instead of being read from a dense [N,3] tensor that is zero almost everywhere.
template <bool DET>
__global__ void __launch_bounds__(256) k_render_bwd_rows(PathCtx c, const double* __restrict__ origin, const double* __restrict__ dir,
                                                         const double* __restrict__ screen_pixel, const int32_t* __restrict__ face1,
                                                         const int32_t* __restrict__ face2, const int32_t* __restrict__ rows,
                                                         const unsigned* __restrict__ n_rows, const double* __restrict__ scale, double* grad_verts) {
    __shared__ int32_t hkeys[kHashSize];
    __shared__ double hsums[3 * kHashSize];
    const unsigned n = *n_rows;
    const double sc = *scale;
    const PathSink<DET> add{hkeys, hsums, grad_verts};
    for (unsigned base = blockIdx.x * kBwdBatch; base < n; base += gridDim.x * kBwdBatch) {
        add.clear();
        const unsigned end = base + kBwdBatch < n ? base + kBwdBatch : n;
        for (unsigned k = base + threadIdx.x; k < end; k += blockDim.x) {
            const int64_t i = rows[k];
            d3 v0, v1, v2;
            int32_t vid1[3], vid2[3];
            Bounce b1, b2;
            load_tri64(c, face1[i], v0, v1, v2, vid1);
            bounce_forward(load_d3(origin, i), load_d3(dir, i), v0, v1, v2, c.ior_ext, c.ior_int, b1);
            load_tri64(c, face2[i], v0, v1, v2, vid2);
            bounce_forward(b1.new_o, b1.wt, v0, v1, v2, c.ior_ext, c.ior_int, b2);
            d3 g_dir;
            (void)ray_loss_term(b2.new_o, b2.wt, load_d3(screen_pixel, i), g_dir);
            g_dir = sc * g_dir;
            const d3 z{0.0, 0.0, 0.0};
            d3 ga = z, gb = z, gc = z, g_o, g_d, g_o0, g_d0;
            bounce_backward(b2, z, g_dir, ga, gb, gc, g_o, g_d);
            add(vid2[0], ga); add(vid2[1], gb); add(vid2[2], gc);
            ga = z; gb = z; gc = z;
            bounce_backward(b1, g_o, g_d, ga, gb, gc, g_o0, g_d0);
            add(vid1[0], ga); add(vid1[1], gb); add(vid1[2], gc);
        }
        add.flush();
    }
}

// ray_loss AND its vertex gradient over the forward's list of completed paths, in one pass: what k_ray_loss_listed (loss, row list)
// and, one autograd hop later, k_render_bwd_rows (recompute the listed paths, adjoint, scatter) did in two -- the second already
// recomputed every path and its loss term.  The gradient is accumulated with a UNIT seed into a stash the caller scales by the
// incoming d / d loss when (if) the backward pass arrives; the loss term of a path recomputed from its face ids is the term of
// the stored out_ori / out_dir rows bit for bit (same code).
template <bool DET>
__global__ void __launch_bounds__(256) k_loss_bwd_listed(PathCtx c, const double* __restrict__ origin, const double* __restrict__ dir,
                                                         const double* __restrict__ screen_pixel, const uint8_t* __restrict__ valid,
                                                         const int32_t* __restrict__ face1, const int32_t* __restrict__ face2,
                                                         const int32_t* __restrict__ paths, const int64_t* __restrict__ n_paths,
                                                         const unsigned* __restrict__ first_ptr, const unsigned* __restrict__ n32_ptr,
                                                         double* loss, double* grad_verts) {
    __shared__ int32_t hkeys[kHashSize];
    __shared__ double hsums[3 * kHashSize];
    // the listed range: [*first_ptr (0 without one), *n32_ptr or *n_paths)
    const int64_t first = first_ptr ? (int64_t)*first_ptr : 0;
    paths += first;
    const int64_t n = (n32_ptr ? (int64_t)*n32_ptr : *n_paths) - first;
    const PathSink<DET> add{hkeys, hsums, grad_verts};
    LossAcc<DET> acc;
    for (int64_t base = blockIdx.x * (int64_t)kBwdBatch; base < n; base += (int64_t)gridDim.x * kBwdBatch) {
        add.clear();
        const int64_t end = base + kBwdBatch < n ? base + kBwdBatch : n;
        for (int64_t k = base + threadIdx.x; k < end; k += blockDim.x) {
            const int64_t i = paths[k];
            if (!valid[i]) continue;
            d3 v0, v1, v2;
            int32_t vid1[3], vid2[3];
            Bounce b1, b2;
            load_tri64(c, face1[i], v0, v1, v2, vid1);
            bounce_forward(load_d3(origin, i), load_d3(dir, i), v0, v1, v2, c.ior_ext, c.ior_int, b1);
            load_tri64(c, face2[i], v0, v1, v2, vid2);
            bounce_forward(b1.new_o, b1.wt, v0, v1, v2, c.ior_ext, c.ior_int, b2);
            d3 g_dir;
            acc.add(ray_loss_term(b2.new_o, b2.wt, load_d3(screen_pixel, i), g_dir));
            const d3 z{0.0, 0.0, 0.0};
            d3 ga = z, gb = z, gc = z, g_o, g_d, g_o0, g_d0;
            bounce_backward(b2, z, g_dir, ga, gb, gc, g_o, g_d);
            add(vid2[0], ga); add(vid2[1], gb); add(vid2[2], gc);
            ga = z; gb = z; gc = z;
            bounce_backward(b1, g_o, g_d, ga, gb, gc, g_o0, g_d0);
            add(vid1[0], ga); add(vid1[1], gb); add(vid1[2], gc);
        }
        add.flush();
    }
    acc.flush(loss);
}

__global__ void k_prof_counts(const unsigned* __restrict__ qcount, unsigned long long n_rays, unsigned long long* __restrict__ tot, int fused, int raster, bool mega) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // sub-batches on different streams may report concurrently: atomics
    atomicAdd(&tot[kStageCull], n_rays);
    if (raster) atomicAdd(&tot[kStageRaster], n_rays);
    if (raster && !fused) atomicAdd(&tot[kStageFill], n_rays);
    if (mega) {       // one kernel did the back half: its item count is list R1; the size list R2 would have had was counted inside it
        atomicAdd(&tot[kStageTrace1], (unsigned long long)qcount[raster ? 3 : 0]); atomicAdd(&tot[kStageShade1], (unsigned long long)qcount[0] + qcount[17]);
        atomicAdd(&tot[kStagePath], (unsigned long long)qcount[1]);
        atomicAdd(&tot[kStageTrace2], (unsigned long long)qcount[1]); atomicAdd(&tot[kStageShade2], (unsigned long long)qcount[1]);
        atomicAdd(&tot[kStageTrace3], (unsigned long long)qcount[6]);
        atomicAdd(&tot[fused ? kStageLossBwdFused : kStageFinish], (unsigned long long)qcount[6]);
        return;
    }
    atomicAdd(&tot[kStageTrace1], (unsigned long long)qcount[raster ? 3 : 0]); atomicAdd(&tot[kStageShade1], (unsigned long long)qcount[0] + qcount[17]);
    atomicAdd(&tot[kStageTrace2], (unsigned long long)qcount[1]); atomicAdd(&tot[kStageShade2], (unsigned long long)qcount[1]);
    atomicAdd(&tot[kStageTrace3], (unsigned long long)qcount[2]);
    atomicAdd(&tot[fused ? kStageLossBwdFused : kStageFinish], (unsigned long long)qcount[2]);
}
__global__ void k_prof_counts_bwd(const unsigned* __restrict__ vcount, unsigned long long n_rays, unsigned long long* __restrict__ tot) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    tot[kStageCollect] += n_rays;
    tot[kStageBackward] += vcount[0];
}

// Queue workspace for one chunk of `n` rays (grown, never shrunk).  Growing frees the old buffers,
// which synchronises the device once; steady-state calls allocate nothing.
static int ensure_queues(drt_scene::Sub& w, int64_t n, bool fused, bool mega = false) {
    if (fused && mega && n > w.ray64_cap) {         // fused one-kernel path: float64 refracted rays parked between bounce #1 and #2 (the drop-in form uses rows of its outputs)
        (void)hipFree(w.ray64); w.ray64 = nullptr; w.ray64_cap = 0;
        HIP_TRY(hipMalloc(&w.ray64, sizeof(double) * 6 * n));
        w.ray64_cap = n;
    }
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
        (void)hipFree(w.gen_list); w.gen_list = nullptr;
        HIP_TRY(hipMalloc(&w.gen_list, sizeof(int32_t) * n));
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
static Plan plan_call(const drt_scene* s, int64_t n_rays, int tile_w, int tile_h) {
    int64_t unit = 256;
    if (tile_w >= 64 && tile_w % 64 == 0) unit = 4 * (int64_t)tile_w;       // whole rows of 64x4 patches per sub-batch
    if (unit > 256 && tile_h >= 4 && tile_h % 4 == 0 && n_rays % ((int64_t)tile_w * tile_h) == 0) unit = (int64_t)tile_w * tile_h;   // whole images (projected primary visibility)
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

// ESTABLISH: the fitted model of every image and whether all of its rays verified (k_cull cleared `all` otherwise).  `ok` is
// stored as "all rays verified" too: an image whose projection had no bound in THIS call (camera plane through a
// triangle) verified nothing and is simply never trusted.
__global__ void k_store_models(const ViewModel* __restrict__ work, ViewModel* __restrict__ cache, int n) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    ViewModel m = work[v];
    m.all = (m.ok && m.all) ? 1 : 0;
    m.ok = m.all;
    cache[v] = m;
}

// Whole images of a multiple-of-64 width and multiple-of-4 height: the sub-batch can use the projected primary visibility.
static bool raster_on(const drt_scene* s, int64_t n, int tile_w, int tile_h) {
    if (!s->use_raster || s->n_faces <= 0 || tile_w < 64 || tile_w % 64 != 0 || tile_h < 4 || tile_h % 4 != 0) return false;
    const int64_t image = (int64_t)tile_w * tile_h;
    return n > 0 && n % image == 0 && n / image <= 65535;
}

// cull -> trace -> shade1 -> trace -> shade2 -> trace(any) for one sub-batch; the caller appends the last stage.
// internal bits of launch_chunk's grid_mode (above the public DRT_GRID_* ones): which dense outputs the caller zeroed ahead of time
constexpr int kIntPreOri = 1 << 16, kIntPreDir = 1 << 17, kIntPreMask = 1 << 18, kIntMask = kIntPreOri | kIntPreDir | kIntPreMask;
// `w` non-null and whole images (tile_w x tile_h) of a multiple-of-64 width: primary visibility by projection
// (drt_raster.h), k_trace #1 then only sees the rays that are not grid rays (normally none).
extern "C++" {
template <bool FUSED>
static int launch_chunk(drt_scene* s, drt_scene::Sub& w, hipStream_t st, const PathCtx& pc, const Pipe& p, const double* o, const double* d, const uint8_t* valid,
                        int64_t n, double* out_ori, double* out_dir, uint8_t* mask, int32_t* face1, int32_t* face2, int tile_w, int tile_h,
                        int grid_mode, ViewModel* grid_cache /* models of the images of THIS sub-batch */,
                        hipStream_t aux /* the caller's stream: idle between the fork and the join of a call */, int64_t chunk_base, int32_t* valid_idx,
                        bool* n_finished /* out: the sub-batch ran as ONE kernel (k_path), which also did k_finish's work */,
                        int32_t* seed2 = nullptr /* temporal hit seeds of THIS sub-batch's camera rays (drt_render_seed), or null */) {
    if (n_finished) *n_finished = false;
    const bool sparse_faces = (grid_mode & DRT_GRID_SPARSE_FACES) != 0;
    const bool pre_ori = (grid_mode & kIntPreOri) != 0, pre_dir = (grid_mode & kIntPreDir) != 0, pre_mask = (grid_mode & kIntPreMask) != 0;   // zeroed ahead of time (drt_prefill_zero)
    const bool all_verified = (grid_mode & DRT_GRID_ALL_VERIFIED) != 0 && (grid_mode & 3) == DRT_GRID_TRUST && grid_cache;
    grid_mode &= 3;
    const int gs = 8 * s->n_cu;   // grid of the streaming / shading kernels
    if (tile_w < 64 || tile_w % 64 != 0 || n % (4 * (int64_t)tile_w) != 0) tile_w = 0;   // not whole 64x4 patches: linear order
    RasterIn rz{nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0, 0};
    const bool mega = s->mega_max_rays > 0 && n <= s->mega_max_rays && n_finished != nullptr;
    const Ray64 r64 = !mega ? Ray64{nullptr, nullptr} : (FUSED ? Ray64{w.ray64, w.ray64 + 3 * n} : Ray64{out_ori, out_dir});
    bool late_fill = false;
    // the scene's build stream is idle once the tree is built (before the cull stage): it carries the late fills, so that the
    // library stays within the four hardware queues a process gets by default (more streams would share queues with these)
    hipStream_t fs = s->prof_serial ? st : s->build_stream;
    {   // under stream capture (a whole step recorded as a hipGraph) the fills go to the CALLER's stream -- the origin of the capture, idle
        // between the fork and the join of this call -- instead of the build stream: re-entering the build stream after it has joined
        // faults inside the capture of ROCm 7.2.  (They were issued on the sub-batch's own stream at first, in line with its kernels: a
        // 36-view share then replayed 13 % SLOWER than it ran eagerly, 1.61 vs 1.42 ms.)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (fs != st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) fs = aux ? aux : st;
    }
    const int64_t image = (int64_t)tile_w * tile_h;
    if (raster_on(s, n, tile_w, tile_h)) {
        const int n_views = (int)(n / image);
        int rc = ensure_raster(s, w, n, n_views, st);
        if (rc) return rc;
        if (!grid_cache) grid_mode = DRT_GRID_NONE;
        if (!FUSED && grid_mode == DRT_GRID_TRUST) {
            // dead values into every dense output (plain memsets run at the full write rate); k_cull_listed then only visits
            // the patches that matter.  Nothing writes a row of out_ori / out_dir / mask before k_shade2, so with
            // `fill_overlap` these three memsets are issued later, on the sub-batch's fill stream, beside the VALU-bound
            // traversal of the refracted rays (see below); the face-id arrays (written by k_shade1) are filled here.
            late_fill = s->fill_overlap;
            if (!late_fill) {
                StageTimer t(s, st, kStageFill);
                if (!pre_ori) (void)hipMemsetAsync(out_ori, 0, sizeof(double) * 3 * n, st);
                if (!pre_dir) (void)hipMemsetAsync(out_dir, 0, sizeof(double) * 3 * n, st);
                if (!pre_mask) (void)hipMemsetAsync(mask, 0, 3 * n, st);
            }
            if (!sparse_faces) { (void)hipMemsetAsync(face1, 0xFF, sizeof(int32_t) * n, st); (void)hipMemsetAsync(face2, 0xFF, sizeof(int32_t) * n, st); }
        }
        if (late_fill && mega) {
            // (one-kernel path: the dense rows are written by k_shade1 already -- the float64 refracted rays are parked in them -- so the fills
            // cannot hide beside a later stage on the build stream, which still carries the build at these sizes; they go to the caller's
            // stream, idle until the join, and run beside the projection pass and the cull)
            if (fs != st && aux) fs = aux;
            if (fs != st) { HIP_TRY(hipEventRecord(w.fill_fork, st)); HIP_TRY(hipStreamWaitEvent(fs, w.fill_fork, 0)); }
            { StageTimer tf(s, fs, kStageFill);
              if (!pre_ori) (void)hipMemsetAsync(out_ori, 0, sizeof(double) * 3 * n, fs);
              if (!pre_dir) (void)hipMemsetAsync(out_dir, 0, sizeof(double) * 3 * n, fs);
              if (!pre_mask) (void)hipMemsetAsync(mask, 0, 3 * n, fs); }
            if (fs != st) HIP_TRY(hipEventRecord(w.fill_join, fs));
        }
        StageTimer t(s, st, kStageRaster);
        rc = launch_raster(s, w, st, o, d, n_views, tile_w, tile_h, grid_mode == DRT_GRID_TRUST ? grid_cache : nullptr);
        if (rc) return rc;
        rz = RasterIn{w.vmodel, grid_mode, w.zbuf, w.zmask, w.gen_list, tile_w, tile_h, all_verified ? 1 : 0, all_verified ? 1 : 0};
    }
    // The cull stage reads the tree for rays outside the grid (top-box test, k_trace on the listed slots): wait for the build (it ran
    // beside the projection pass above).  When the caller vouches that every ray of every image is a verified grid ray, nothing before
    // the traversal of the refracted rays touches the tree -- the wait moves there, and at small per-GPU shares (9 views: build 0.19 ms,
    // projection + cull + first shading 0.17 ms) the build leaves the critical path.
    // outputs zeroed ahead of time: the zeroing must have finished before anything writes them -- k_shade2 when the fills of this call are
    // the late ones (k_shade1 then leaves the dense outputs alone), the cull stage otherwise
    const bool pre_any = pre_ori || pre_dir || pre_mask;
    if (pre_any && !late_fill) HIP_TRY(hipStreamWaitEvent(st, s->prefill_done, 0));
    const bool tree_late = all_verified && rz.views != nullptr;
    if (!tree_late) { int rc = wait_build(s, st); if (rc) return rc; }
    // bounce #1 inside the cull of the listed patches (k_cull_listed's `direct`); the one-kernel path parks float64 rays in k_shade1: not there
    // Only for large sub-batches: the listed patches are handled one after the other by each block, a few hits per patch, so the float64
    // bounce runs at low lane use and with the patch loop's barriers around it -- cheaper than a pass of its own over list R0 once the chip
    // is saturated (72 views: -1.3 %, the object filling the image: -4.3 %), dearer when the call is a chain of latencies (36 views in two
    // sub-batches +3.5 %, 9 views +6.5 %).
    const bool direct = s->cull_direct && rz.views && grid_mode == DRT_GRID_TRUST && !mega && n >= s->cull_direct_min_rays;
    // ... and parks the float64 refracted rays in the rows of the dense outputs when those are zeroed already (recycled / zeroed ahead of
    // time: no fill of this call can run over them) and the zeroing is through (prefill_done)
    const bool park = !FUSED && !mega && rz.views && grid_mode == DRT_GRID_TRUST && pre_ori && pre_dir && pre_mask && s->cull_park;      // (with or without `direct`: k_shade1 parks too)
    const Ray64 r64p = park ? Ray64{out_ori, out_dir} : r64;      // (k_shade1 and k_gen_late, which serve the untrusted images, park likewise)
    if (park) HIP_TRY(hipStreamWaitEvent(st, s->prefill_done, 0));
    { StageTimer t(s, st, kStageCull);
      const unsigned n_patches = (unsigned)((n + kPathBlock - 1) / kPathBlock);
      if (rz.views && grid_mode == DRT_GRID_TRUST) {
          // (the dense outputs were pre-filled with the dead values above, before the projection pass)
          uint32_t* list = reinterpret_cast<uint32_t*>(p.redo);          // free until the first k_trace of this sub-batch
          k_patch_list<<<(n_patches + kPathBlock - 1) / kPathBlock, kPathBlock, 0, st>>>(n_patches, tile_w, rz, list, p.count + 7);
          k_cull_listed<FUSED><<<gs, kPathBlock, 0, st>>>(list, p.count + 7, pc.tc.nodes, pc.tc.n_tris, o, d, valid, n, out_ori, out_dir, mask, face1, face2, p, tile_w, rz,
                                                          pc, direct, late_fill, park ? Ray64{out_ori, out_dir} : Ray64{nullptr, nullptr});
      } else {
          // (every ray is read and decided here; with all three outputs already zero -- recycled -- and face ids only promised where mask = 1,
          // the dead ones, nine in ten, need not be written again: 56 B read per ray instead of 56 read + 59 written)
          k_cull<FUSED><<<n_patches, kPathBlock, 0, st>>>(pc.tc.nodes, pc.tc.n_tris, o, d, valid, n, out_ori, out_dir, mask, face1, face2, p, tile_w, rz,
                                                          pre_ori && pre_dir && pre_mask && sparse_faces);
      } }
    // the late fills: issued behind the cull stage (beside k_shade1 and the second traversal) or, DRT_FILL_AFTER_SHADE1, behind k_shade1
    // (beside the second traversal only: the latency-bound first shading then does not share the memory system with 2 GB of memsets)
    auto issue_late_fills = [&]() -> int {
        if (fs != st) { HIP_TRY(hipEventRecord(w.fill_fork, st)); HIP_TRY(hipStreamWaitEvent(fs, w.fill_fork, 0)); }
        { StageTimer tf(s, fs, kStageFill);
          if (!pre_ori) (void)hipMemsetAsync(out_ori, 0, sizeof(double) * 3 * n, fs);
          if (!pre_dir) (void)hipMemsetAsync(out_dir, 0, sizeof(double) * 3 * n, fs);
          if (!pre_mask) (void)hipMemsetAsync(mask, 0, 3 * n, fs); }
        if (fs != st) HIP_TRY(hipEventRecord(w.fill_join, fs));
        return DRT_OK;
    };
    const bool fills_after_shade1 = s->fill_after_shade1 && fs != st;
    if (late_fill && !mega && !fills_after_shade1) { int rc = issue_late_fills(); if (rc) return rc; }      // (StageTimer scopes must not nest: this one follows the cull stage's)
    if (rz.views && grid_mode == DRT_GRID_ESTABLISH)      // what later DRT_GRID_TRUST calls with the same rays may rely on
        k_store_models<<<(int)(n / ((int64_t)tile_w * tile_h) + 63) / 64, 64, 0, st>>>(w.vmodel, grid_cache, (int)(n / ((int64_t)tile_w * tile_h)));
    { StageTimer t(s, st, kStageTrace1);
      if (rz.views && all_verified) {
          // every ray of every image is a verified grid ray (the caller read that off the cache): normally nothing was listed, and what
          // a demotion inside this call did list waits for k_gen_late below
      } else if (rz.views) {       // only the R0 slots listed by k_cull (rays that are not grid rays)
          const TraceOut out{p.r0.face, nullptr, nullptr, w.gen_list};
          launch_trace_list(kTraceClosestListed, s->grid_path, st, pc.tc, p.r0.ray, p.count + 3, out, p.redo, p.count + 4, p.count + 16, s->refill_min, s->inner_min, s->prof_stats ? s->prof_counts + kProfStages + 4 * 0 : nullptr);
      } else {
          launch_trace_list(kTraceClosest, s->grid_path, st, pc.tc, p.r0.ray, p.count + 0, TraceOut{p.r0.face, nullptr, nullptr, nullptr}, p.redo, p.count + 4, p.count + 16, s->refill_min, s->inner_min, s->prof_stats ? s->prof_counts + kProfStages + 4 * 0 : nullptr);
      } }
    if (mega) {     // k_shade1 parks float64 rays in rows of the dense outputs: their zeroing must be through
        if (late_fill && fs != st) HIP_TRY(hipStreamWaitEvent(st, w.fill_join, 0));
        if (pre_any) HIP_TRY(hipStreamWaitEvent(st, s->prefill_done, 0));
    }
    {   // (with `direct` list R0 only holds what untrusted images contributed -- normally nothing: a launch that returns at once)
      StageTimer t(s, st, kStageShade1);
      k_shade1<FUSED><<<gs, kPathBlock, 0, st>>>(pc, o, d, out_ori, out_dir, mask, face1, face2, p, late_fill, r64p); }
    if (late_fill && !mega && fills_after_shade1) { int rc = issue_late_fills(); if (rc) return rc; }
    if (tree_late) {
        int rc = wait_build(s, st); if (rc) return rc;
        StageTimer t(s, st, kStageTrace1);
        k_gen_late<FUSED><<<kRedoGrid, kTraceBlock, 0, st>>>(pc, o, d, out_ori, out_dir, mask, face1, face2, p, w.gen_list, late_fill, r64p);
    }
    if (mega) {
        StageTimer t(s, st, kStagePath);
        k_path<FUSED><<<s->grid_mega, kPathBlock, 0, st>>>(pc, r64, out_ori, out_dir, mask, face2, p, p.count + 8, chunk_base, valid_idx,
                                                           s->mega_refill_min, s->inner_min, s->shade_min, s->prof_stats ? s->prof_counts + kProfStages + 4 * 1 : nullptr, s->prof_on);
        k_path_redo<FUSED><<<kRedoGrid, kTraceBlock, 0, st>>>(pc, o, d, face1, out_ori, out_dir, mask, face2, p, chunk_base, valid_idx);
        *n_finished = true;
        return DRT_OK;
    }
    { StageTimer t(s, st, kStageTrace2);
      unsigned long long* stats = s->prof_stats ? s->prof_counts + kProfStages + 4 * 1 : nullptr;
      if (seed2 && s->hit_seed)       // the refracted rays start from last call's exit triangle of their pixel (TraceSeed)
          launch_trace_list(kTraceClosestSeeded, s->grid_path, st, pc.tc, p.r1.ray, p.count + 1, TraceOut{p.r1.face, nullptr, nullptr, nullptr}, p.redo, p.count + 5, p.count + 16, s->refill_min, s->inner_min, stats,
                                                                        TraceSeed{p.r1.idx, seed2, s->slot_of_face, tile_w > 0 && s->seed_tiled ? (unsigned)tile_w : 0u});
      else
          launch_trace_list(kTraceClosest, s->grid_path, st, pc.tc, p.r1.ray, p.count + 1, TraceOut{p.r1.face, nullptr, nullptr, nullptr}, p.redo, p.count + 5, p.count + 16, s->refill_min, s->inner_min, stats); }
    if (late_fill && fs != st) HIP_TRY(hipStreamWaitEvent(st, w.fill_join, 0));     // k_shade2 is the first kernel that writes rows of the dense outputs
    if (late_fill && pre_any) HIP_TRY(hipStreamWaitEvent(st, s->prefill_done, 0));
    { StageTimer t(s, st, kStageShade2);
      k_shade2<FUSED><<<gs, kPathBlock, 0, st>>>(pc, o, d, out_ori, out_dir, mask, face1, face2, p, park); }
    { StageTimer t(s, st, kStageTrace3);
      launch_trace_list(kTraceAny, s->grid_path, st, pc.tc, p.r2.ray, p.count + 2, TraceOut{p.r2.face, nullptr, nullptr, nullptr}, p.redo, p.count + 6, p.count + 16, s->refill_min, s->inner_min, s->prof_stats ? s->prof_counts + kProfStages + 4 * 2 : nullptr); }
    return DRT_OK;
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

// the cache entries of the images of the sub-batch that starts at ray b
static ViewModel* sub_cache(void* d_grid_cache, int64_t b, int tile_w, int tile_h) {
    if (!d_grid_cache || tile_w <= 0 || tile_h <= 0) return nullptr;
    return static_cast<ViewModel*>(d_grid_cache) + b / ((int64_t)tile_w * tile_h);
}
static PathCtx sub_ctx(const drt_scene* s, const drt_scene::Sub& w, const double* d_verts, double ior_int, double ior_ext) {
    PathCtx pc = path_ctx(s, d_verts, ior_int, ior_ext);
    pc.tc.slow_stack = w.slow_stack;     // concurrent kernels must not share overflow stacks
    return pc;
}
#if defined(DRT_CHECK)
int check_counters_pipeline(unsigned long long* out4) { return read_check_counters(out4); }
#endif
int mega_blocks_per_cu() {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_path<false>, kPathBlock, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    int f = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&f, k_path<true>, kPathBlock, 0) == hipSuccess && f >= 1 && f < per_cu) per_cu = f;
    return per_cu;
}

// The seeds registered for the next render call (drt_render_seed): one shot, and only for a call of exactly that many rays.
static int32_t* take_seed(drt_scene* s, int64_t n_rays) {
    int32_t* p = s->seed.face2 && s->seed.n == n_rays ? s->seed.face2 : nullptr;
    s->seed = drt_scene::Seed{};
    return p;
}

extern "C" {

int drt_render_seed(drt_scene_t* s, int32_t* d_seed_face2, int64_t n_rays) {
    CHECK_SCENE(s);
    if (n_rays < 0 || (d_seed_face2 == nullptr && n_rays != 0)) return fail(DRT_E_INVALID, "bad seed buffer");
    s->seed = drt_scene::Seed{d_seed_face2, n_rays};
    return DRT_OK;
}

static int flush_clean(drt_scene* s, hipStream_t st);
static int render_forward_impl(drt_scene_t* s, const double* d_verts, const double* d_origin, const double* d_dir, int64_t n_rays,
                               double ior_int, double ior_ext, double* d_out_ori, double* d_out_dir, uint8_t* d_mask,
                               int32_t* d_face1, int32_t* d_face2, int32_t* d_valid_idx, int64_t* d_n_valid, int tile_w, int tile_h,
                               int grid_mode, void* d_grid_cache, void* stream);
int drt_render_forward(drt_scene_t* s, const double* d_verts, const double* d_origin, const double* d_dir, int64_t n_rays,
                       double ior_int, double ior_ext, double* d_out_ori, double* d_out_dir, uint8_t* d_mask,
                       int32_t* d_face1, int32_t* d_face2, int32_t* d_valid_idx, int64_t* d_n_valid, int tile_w, int tile_h,
                       int grid_mode, void* d_grid_cache, void* stream) {
    // A call that is being CAPTURED into a graph and was offered its outputs through drt_outputs_clean uses an event of its own for "the rows
    // are zero again": an event recorded inside a capture must not be waited for by eager work afterwards (and vice versa).
    hipStreamCaptureStatus cap0 = hipStreamCaptureStatusNone;
    const bool swap_ev = s && s->clean.rows && s->clean.captured && hipStreamIsCapturing((hipStream_t)stream, &cap0) == hipSuccess && cap0 != hipStreamCaptureStatusNone;
    if (swap_ev) std::swap(s->prefill_done, s->prefill_done_cap);
    const int rc = render_forward_impl(s, d_verts, d_origin, d_dir, n_rays, ior_int, ior_ext, d_out_ori, d_out_dir, d_mask, d_face1, d_face2, d_valid_idx,
                                       d_n_valid, tile_w, tile_h, grid_mode, d_grid_cache, stream);
    if (swap_ev) std::swap(s->prefill_done, s->prefill_done_cap);
    if (rc != DRT_OK && s) {
        // a call that failed must not leave raw pointers of its caller registered: the caller is about to release those buffers, and the
        // next call would zero "rows" of, or skip the fills of, memory that belongs to somebody else by then
        s->clean = drt_scene::Clean{};
        s->n_prefill = 0;
        s->seed = drt_scene::Seed{};
        s->segs = drt_scene::Segs{};
    }
    return rc;
}
static int render_forward_impl(drt_scene_t* s, const double* d_verts, const double* d_origin, const double* d_dir, int64_t n_rays,
                               double ior_int, double ior_ext, double* d_out_ori, double* d_out_dir, uint8_t* d_mask,
                               int32_t* d_face1, int32_t* d_face2, int32_t* d_valid_idx, int64_t* d_n_valid, int tile_w, int tile_h,
                               int grid_mode, void* d_grid_cache, void* stream) {
    CHECK_BUILT(s);
    if (n_rays < 0 || n_rays > INT32_MAX) return fail(DRT_E_INVALID, "ray count out of range");
    hipStream_t st = (hipStream_t)stream;
    if (n_rays == 0) {
        if (d_n_valid) HIP_TRY(hipMemsetAsync(d_n_valid, 0, sizeof(int64_t), st));
        return DRT_OK;
    }
    if (!d_verts || !d_origin || !d_dir || !d_out_ori || !d_out_dir || !d_mask || !d_face1 || !d_face2) return fail(DRT_E_INVALID, "null pointer argument");
    if ((d_valid_idx == nullptr) != (d_n_valid == nullptr)) return fail(DRT_E_INVALID, "d_valid_idx and d_n_valid go together");
    int32_t* const seed2 = take_seed(s, n_rays);
    const Plan pl = plan_call(s, n_rays, tile_w, tile_h);
    for (int k = 0; k < pl.streams; ++k) { int rc = ensure_queues(s->sub[k], pl.size, false); if (rc) return rc; }
    HIP_TRY(hipMemsetAsync(s->vcount, 0, sizeof(unsigned), st));
    // the list of completed paths by sub-batch (drt_scene::segs): only when the call is cut into a few sub-batches
    const bool segmented = d_valid_idx != nullptr && pl.count >= 2 && pl.count <= drt_scene::kMaxSeg;
    if (segmented) HIP_TRY(hipMemsetAsync(s->seg_counts, 0, sizeof(unsigned) * pl.count, st));
    s->segs = drt_scene::Segs{};
    grid_mode &= ~kIntMask;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    // (while a graph is being captured the buffers zeroed ahead of time stay registered: the capturing call brings its own outputs, and
    // nothing outside the capture may be waited for inside it)
    if (s->n_prefill && (cap == hipStreamCaptureStatusNone || (s->clean.rows && s->clean.captured))) {       // buffers zeroed ahead of time (drt_prefill_zero) / offered again (drt_outputs_clean): which of this call's outputs are they?
        int matched = 0;
        for (int k = 0; k < s->n_prefill; ++k) {
            const drt_scene::Prefill& f = s->prefill[k];
            const int before = grid_mode;
            if (f.ptr == d_out_ori && f.bytes == (int64_t)sizeof(double) * 3 * n_rays) grid_mode |= kIntPreOri;
            if (f.ptr == d_out_dir && f.bytes == (int64_t)sizeof(double) * 3 * n_rays) grid_mode |= kIntPreDir;
            if (f.ptr == d_mask && f.bytes == 3 * n_rays) grid_mode |= kIntPreMask;
            matched += grid_mode != before;
        }
        // an entry that is none of this call's outputs is dropped here: whoever writes that buffer next no longer knows about its
        // zeroing, so the zeroing is ordered in front of everything this stream does from now on
        if (matched != s->n_prefill && cap == hipStreamCaptureStatusNone) HIP_TRY(hipStreamWaitEvent(st, s->prefill_done, 0));
        s->n_prefill = 0;     // one shot (launch_chunk waits for prefill_done in front of the first kernel that writes those outputs)
    }
    // drt_outputs_clean's rows.  A trusted-grid call through the projection pass writes no row of the dense outputs before k_shade2, which
    // waits for prefill_done: the zeroing then runs on the caller's stream BEHIND the fork, beside the pipelines' first stages.  Any other
    // call (k_cull writes every row straight away) gets it in front of the fork.
    bool clean_late = s->clean.rows != nullptr && (grid_mode & 3) == DRT_GRID_TRUST && d_grid_cache != nullptr;
    for (int j = 0; j < pl.count && clean_late; ++j) {
        const int64_t b = j * pl.size, n = n_rays - b < pl.size ? n_rays - b : pl.size;
        clean_late = raster_on(s, n, tile_w, tile_h);
    }
    int rc = DRT_OK;
    if (s->clean.rows && !clean_late) { rc = flush_clean(s, st); if (rc) return rc; }
    rc = fork_streams(s, st, pl.streams);
    if (rc) return rc;
    if (s->clean.rows) { rc = flush_clean(s, st); if (rc) return rc; }
    for (int j = 0; j < pl.count; ++j) {
        drt_scene::Sub& w = s->sub[j % pl.streams];
        const int64_t b = j * pl.size;
        const int64_t n = n_rays - b < pl.size ? n_rays - b : pl.size;
        const PathCtx pc = sub_ctx(s, w, d_verts, ior_int, ior_ext);
        Pipe p = pipe_of(s, w);
        int32_t* vlist = d_valid_idx;
        if (segmented) { p.valid = s->seg_counts + j; vlist = d_valid_idx + b; }      // this sub-batch's own segment and counter
        HIP_TRY(hipMemsetAsync(w.qcount, 0, kQCount * sizeof(unsigned), w.stream));
        bool finished = false;
        rc = launch_chunk<false>(s, w, w.stream, pc, p, d_origin + 3 * b, d_dir + 3 * b, nullptr, n, d_out_ori + 3 * b, d_out_dir + 3 * b,
                                 d_mask + 3 * b, d_face1 + b, d_face2 + b, tile_w, tile_h, grid_mode, sub_cache(d_grid_cache, b, tile_w, tile_h),
                                 st, b, vlist, &finished, seed2 ? seed2 + b : nullptr);
        if (rc) return rc;
        if (!finished) {
            StageTimer t(s, w.stream, kStageFinish);
            k_finish<<<8 * s->n_cu, kPathBlock, 0, w.stream>>>(d_out_ori + 3 * b, d_out_dir + 3 * b, d_mask + 3 * b, d_face2 + b, p, b, vlist);
        }
        if (s->prof_on) k_prof_counts<<<1, 64, 0, w.stream>>>(w.qcount, (unsigned long long)n, s->prof_counts, 0, raster_on(s, n, tile_w, tile_h), finished);
    }
    rc = join_streams(s, st, pl.streams);
    if (rc) return rc;
    if (segmented) {
        for (int j = 1; j < pl.count; ++j)
            k_join_lists<<<2 * s->n_cu, 256, 0, st>>>(d_valid_idx, s->seg_counts, j, (int64_t)j * pl.size, j + 1 == pl.count, s->vcount, d_n_valid);
        // (not while a graph is being captured: a later EAGER drt_ray_loss_listed_grad_split on the same list would start on internal stream 0
        // with no ordering against the replayed graph that fills the list)
        if (cap == hipStreamCaptureStatusNone) s->segs = drt_scene::Segs{d_valid_idx, pl.count, 0};      // (sub-batch 0 runs on internal stream 0)
    } else if (d_n_valid) {
        k_store_count<<<1, 64, 0, st>>>(s->vcount, d_n_valid);
    }
    if (s->prof_on) s->prof_stream = st;
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_prefill_zero(drt_scene_t* s, void* d_buf, int64_t bytes, void* stream) {
    CHECK_SCENE(s);
    if (!d_buf || bytes <= 0) return fail(DRT_E_INVALID, "bad buffer");
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return fail(DRT_E_INVALID, "drt_prefill_zero: not while a graph is being captured");
    if (s->n_prefill >= 3) s->n_prefill = 0;      // stale entries of a caller that never rendered into them
    hipStream_t bs = s->build_stream;
    HIP_TRY(hipEventRecord(s->prefill_fork, st));
    HIP_TRY(hipStreamWaitEvent(bs, s->prefill_fork, 0));
    { StageTimer t(s, bs, kStageFill);
      HIP_TRY(hipMemsetAsync(d_buf, 0, (size_t)bytes, bs)); }
    HIP_TRY(hipEventRecord(s->prefill_done, bs));
    s->prefill[s->n_prefill].ptr = d_buf; s->prefill[s->n_prefill].bytes = bytes; ++s->n_prefill;
    return DRT_OK;
}

static int flush_clean(drt_scene* s, hipStream_t st) {
    const drt_scene::Clean c = s->clean;
    s->clean = drt_scene::Clean{};
    if (!c.rows) return DRT_OK;
    { StageTimer t(s, st, kStageFill);
      k_unwrite_rows<<<4 * s->n_cu, 256, 0, st>>>(c.ori, c.dir, c.mask, c.rows, c.n_rows, c.n); }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->prefill_done, st));
    return DRT_OK;
}

int drt_outputs_clean(drt_scene_t* s, double* d_out_ori, double* d_out_dir, uint8_t* d_mask, int64_t n_rays,
                      const int32_t* d_valid_idx, const int64_t* d_n_valid, void* stream) {
    CHECK_SCENE(s);
    if (n_rays <= 0 || !d_out_ori || !d_out_dir || !d_mask || !d_valid_idx || !d_n_valid) return fail(DRT_E_INVALID, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    // While a graph is being captured the request is recorded like any other and the captured drt_render_forward zeroes the listed rows
    // inside the graph.  That is only the caller's intent when the row list IS the list the captured call writes (same d_valid_idx /
    // d_n_valid): every replay then zeroes the rows its predecessor set -- replay = recycle (drt_amd/diffrender.py pins such a set).
    if (capturing && (s->n_prefill || s->clean.rows)) return fail(DRT_E_INVALID, "drt_outputs_clean: zeroings requested outside the capture are still pending");
    if (s->n_prefill) {       // zeroings of OTHER buffers still pending on the build stream: order them in front of this stream, forget the entries
        HIP_TRY(hipStreamWaitEvent(st, s->prefill_done, 0));
        s->n_prefill = 0;
    }
    // The zeroing itself is enqueued by the drt_render_forward that takes these outputs, on its caller's stream BEHIND the fork of the
    // pipelines (that stream is idle until the join, and nothing writes a row of the outputs before k_shade2, which waits for
    // prefill_done): 30 us that no longer sit between the vertex update and the projection pass.
    if (s->clean.rows) { int rc = flush_clean(s, st); if (rc) return rc; }          // an earlier request nobody rendered into: honour it now
    s->clean = drt_scene::Clean{d_out_ori, d_out_dir, d_mask, n_rays, d_valid_idx, d_n_valid, capturing};
    s->prefill[0].ptr = d_out_ori; s->prefill[0].bytes = (int64_t)sizeof(double) * 3 * n_rays;
    s->prefill[1].ptr = d_out_dir; s->prefill[1].bytes = (int64_t)sizeof(double) * 3 * n_rays;
    s->prefill[2].ptr = d_mask;    s->prefill[2].bytes = 3 * n_rays;
    s->n_prefill = 3;
    return DRT_OK;
}

int drt_outputs_cancel(drt_scene_t* s) {
    CHECK_SCENE(s);
    if (s->clean.rows) { s->clean = drt_scene::Clean{}; s->n_prefill = 0; }
    return DRT_OK;
}

int drt_prefill_wait(drt_scene_t* s, void* stream) {
    CHECK_SCENE(s);
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, s->prefill_done, 0));
    s->n_prefill = 0;
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
        DET_LAUNCH(k_render_bwd, DRT_BWD_BPC * s->n_cu, 256, st, pc, d_origin, d_dir, d_face1, d_face2, d_grad_out_ori, d_grad_out_dir, d_grad_verts,
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
              DET_LAUNCH(k_render_bwd, DRT_BWD_BPC * s->n_cu, 256, st, pc, d_origin, d_dir, d_face1, d_face2, d_grad_out_ori, d_grad_out_dir, d_grad_verts,
                                                         w.q_idx[0], s->vcount, nullptr); }
            if (s->prof_on) k_prof_counts_bwd<<<1, 64, 0, st>>>(s->vcount, (unsigned long long)n, s->prof_counts);
        }
    }
    if (s->prof_on) s->prof_stream = st;
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_render_backward_ray_loss(drt_scene_t* s, const double* d_verts, const double* d_origin, const double* d_dir, int64_t n_rays,
                                 double ior_int, double ior_ext, const int32_t* d_face1, const int32_t* d_face2,
                                 const int32_t* d_rows, const uint32_t* d_n_rows, const double* d_screen_pixel, const double* d_scale,
                                 double* d_grad_verts, void* stream) {
    CHECK_BUILT(s);
    if (n_rays < 0 || n_rays > INT32_MAX) return fail(DRT_E_INVALID, "ray count out of range");
    if (n_rays == 0) return DRT_OK;
    if (!d_verts || !d_origin || !d_dir || !d_face1 || !d_face2 || !d_rows || !d_n_rows || !d_screen_pixel || !d_scale || !d_grad_verts)
        return fail(DRT_E_INVALID, "null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    const PathCtx pc = path_ctx(s, d_verts, ior_int, ior_ext);
    { StageTimer t(s, st, kStageBackward);
      DET_LAUNCH(k_render_bwd_rows, DRT_BWD_BPC * s->n_cu, 256, st, pc, d_origin, d_dir, d_screen_pixel, d_face1, d_face2, d_rows, d_n_rows, d_scale, d_grad_verts); }
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
    DET_LAUNCH(k_ray_loss, grid_for((n_rays + kLossRays - 1) / kLossRays, kPathBlock, 4096), kPathBlock, (hipStream_t)stream, d_out_ori, d_out_dir, d_mask, d_screen_pixel, d_valid, n_rays,
                                                                                          d_loss, d_grad_out_dir, d_list, d_n_list);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_ray_loss_listed(const double* d_out_ori, const double* d_out_dir, const double* d_screen_pixel, const uint8_t* d_valid,
                        const int32_t* d_paths, const int64_t* d_n_paths, int64_t n_rays, double* d_loss,
                        int32_t* d_rows, uint32_t* d_n_rows, void* stream) {
    if (n_rays < 0 || n_rays > INT32_MAX) return fail(DRT_E_INVALID, "ray count out of range");
    if (n_rays == 0) return DRT_OK;
    if (!d_out_ori || !d_out_dir || !d_screen_pixel || !d_valid || !d_paths || !d_n_paths || !d_loss) return fail(DRT_E_INVALID, "null pointer argument");
    if ((d_rows == nullptr) != (d_n_rows == nullptr)) return fail(DRT_E_INVALID, "d_rows and d_n_rows go together");
    DET_LAUNCH(k_ray_loss_listed, 1024, 256, (hipStream_t)stream, d_out_ori, d_out_dir, d_screen_pixel, d_valid, d_paths, d_n_paths, d_loss, d_rows, d_n_rows);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_ray_loss_listed_grad(drt_scene_t* s, const double* d_verts, const double* d_origin, const double* d_dir, int64_t n_rays,
                             double ior_int, double ior_ext, const int32_t* d_face1, const int32_t* d_face2,
                             const double* d_screen_pixel, const uint8_t* d_valid, const int32_t* d_paths, const int64_t* d_n_paths,
                             double* d_loss, double* d_grad_verts, void* stream) {
    CHECK_BUILT(s);
    if (n_rays < 0 || n_rays > INT32_MAX) return fail(DRT_E_INVALID, "ray count out of range");
    if (n_rays == 0) return DRT_OK;
    if (!d_verts || !d_origin || !d_dir || !d_face1 || !d_face2 || !d_screen_pixel || !d_valid || !d_paths || !d_n_paths || !d_loss || !d_grad_verts)
        return fail(DRT_E_INVALID, "null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    const PathCtx pc = path_ctx(s, d_verts, ior_int, ior_ext);
    { StageTimer t(s, st, kStageBackward);
      DET_LAUNCH(k_loss_bwd_listed, DRT_BWD_BPC * s->n_cu, 256, st, pc, d_origin, d_dir, d_screen_pixel, d_valid, d_face1, d_face2, d_paths, d_n_paths, nullptr, nullptr, d_loss, d_grad_verts); }
    if (s->prof_on) s->prof_stream = st;
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_ray_loss_listed_grad_split(drt_scene_t* s, const double* d_verts, const double* d_origin, const double* d_dir, int64_t n_rays,
                                   double ior_int, double ior_ext, const int32_t* d_face1, const int32_t* d_face2,
                                   const double* d_screen_pixel, const uint8_t* d_valid, const int32_t* d_paths, const int64_t* d_n_paths,
                                   double* d_loss, double* d_grad_verts, void* stream) {
    CHECK_BUILT(s);
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cs);
    // the list of the last render call, cut into sub-batches on different internal streams, outside a capture: otherwise the plain form
    if (s->segs.list != d_paths || s->segs.n < 2 || cs != hipStreamCaptureStatusNone || n_rays <= 0)
        return drt_ray_loss_listed_grad(s, d_verts, d_origin, d_dir, n_rays, ior_int, ior_ext, d_face1, d_face2, d_screen_pixel, d_valid, d_paths, d_n_paths,
                                        d_loss, d_grad_verts, stream);
    if (n_rays > INT32_MAX) return fail(DRT_E_INVALID, "ray count out of range");
    if (!d_verts || !d_origin || !d_dir || !d_face1 || !d_face2 || !d_screen_pixel || !d_valid || !d_n_paths || !d_loss || !d_grad_verts)
        return fail(DRT_E_INVALID, "null pointer argument");
    const PathCtx pc = path_ctx(s, d_verts, ior_int, ior_ext);
    drt_scene::Sub& w = s->sub[s->segs.stream0];
    // segment 0 (never moved by the join) on the internal stream that produced it: it starts as soon as that pipeline is through, beside
    // the tail of the other one; the rest of the list on the caller's stream, behind the join; the caller's stream then waits for both.
    { StageTimer t(s, w.stream, kStageBackward);
      DET_LAUNCH(k_loss_bwd_listed, DRT_BWD_BPC * s->n_cu, 256, w.stream, pc, d_origin, d_dir, d_screen_pixel, d_valid, d_face1, d_face2, d_paths, nullptr, nullptr,
                                                                  s->seg_counts, d_loss, d_grad_verts); }
    HIP_TRY(hipEventRecord(w.done, w.stream));
    { StageTimer t(s, st, kStageBackward);
      DET_LAUNCH(k_loss_bwd_listed, DRT_BWD_BPC * s->n_cu, 256, st, pc, d_origin, d_dir, d_screen_pixel, d_valid, d_face1, d_face2, d_paths, d_n_paths, s->seg_counts,
                                                           nullptr, d_loss, d_grad_verts); }
    HIP_TRY(hipStreamWaitEvent(st, w.done, 0));
    if (s->prof_on) s->prof_stream = st;
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
                              double ior_ext, double* d_loss, double* d_grad_verts, int64_t* d_n_valid, int tile_w, int tile_h,
                              int grid_mode, void* d_grid_cache, void* stream) {
    CHECK_BUILT(s);
    if (n_rays < 0 || n_rays > INT32_MAX) return fail(DRT_E_INVALID, "ray count out of range");
    if (n_rays == 0) return DRT_OK;
    if (!d_verts || !d_origin || !d_dir || !d_screen_pixel || !d_valid || !d_loss || !d_grad_verts) return fail(DRT_E_INVALID, "null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    int32_t* const seed2 = take_seed(s, n_rays);
    const Plan pl = plan_call(s, n_rays, tile_w, tile_h);
    for (int k = 0; k < pl.streams; ++k) { int rc = ensure_queues(s->sub[k], pl.size, true, s->mega_max_rays > 0 && pl.size <= s->mega_max_rays); if (rc) return rc; }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    if (s->n_prefill && cap == hipStreamCaptureStatusNone) {   // buffers zeroed ahead of time for a drt_render_forward that did not come: forgotten, their zeroing ordered in front of this stream's future
        HIP_TRY(hipStreamWaitEvent(st, s->prefill_done, 0));
        s->n_prefill = 0;
    }
    int rc = fork_streams(s, st, pl.streams);
    if (rc) return rc;
    grid_mode &= ~kIntMask;        // (internal bits: this entry point has no dense outputs that could have been zeroed ahead of time)
    for (int j = 0; j < pl.count; ++j) {
        drt_scene::Sub& w = s->sub[j % pl.streams];
        const int64_t b = j * pl.size;
        const int64_t n = n_rays - b < pl.size ? n_rays - b : pl.size;
        const PathCtx pc = sub_ctx(s, w, d_verts, ior_int, ior_ext);
        const Pipe p = pipe_of(s, w);
        HIP_TRY(hipMemsetAsync(w.qcount, 0, kQCount * sizeof(unsigned), w.stream));
        bool finished = false;
        rc = launch_chunk<true>(s, w, w.stream, pc, p, d_origin + 3 * b, d_dir + 3 * b, d_valid + b, n, nullptr, nullptr, nullptr, w.tmp_face1, w.tmp_face2, tile_w, tile_h, grid_mode, sub_cache(d_grid_cache, b, tile_w, tile_h),
                                st, b, nullptr, &finished, seed2 ? seed2 + b : nullptr);
        if (rc) return rc;
        { StageTimer t(s, w.stream, kStageLossBwdFused);
          DET_LAUNCH(k_loss_bwd_fused, DRT_BWD_BPC * s->n_cu, 256, w.stream, pc, d_origin + 3 * b, d_dir + 3 * b, d_screen_pixel + 3 * b, w.tmp_face1, w.tmp_face2, p,
                                                               d_loss, d_grad_verts, reinterpret_cast<unsigned long long*>(d_n_valid)); }
        if (s->prof_on) k_prof_counts<<<1, 64, 0, w.stream>>>(w.qcount, (unsigned long long)n, s->prof_counts, 1, raster_on(s, n, tile_w, tile_h), finished);
    }
    rc = join_streams(s, st, pl.streams);
    if (rc) return rc;
    if (s->prof_on) s->prof_stream = st;
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}


}  // extern "C"
