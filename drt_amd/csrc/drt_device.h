// drt_device.h -- small device-side helpers shared by the kernels of more than one translation unit.
#pragma once
#include "drt_scene.h"

constexpr int kPathBlock = 256;
constexpr int kPathWaves = kPathBlock / 64;

__device__ __forceinline__ Stack make_stack(int32_t (*lds)[kTraceBlock], const TraceCtx& c) {
    Stack st;
    st.fast = &lds[0][threadIdx.x];
    st.stride = kTraceBlock;
    st.depth_fast = kStackFast;
    st.slow = c.slow_stack + ((int64_t)blockIdx.x * kTraceBlock + threadIdx.x) * kStackSlowDev;
    st.sp = 0;
    return st;
}

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

// Conservative "can this ray touch the mesh at all": two levels of the wide tree (the root's
// children, then the children of every inner child the ray enters).  k_cull is HBM-bound, so these
// <= 20 slab tests are free, and every ray they reject is one the traversal stages never see.
__device__ __forceinline__ unsigned hit_mask4(const Node4Q* __restrict__ node, f3 inv, f3 oi) {
    const F4* np = reinterpret_cast<const F4*>(node);
    const int32_t* ch = node->child;
    float t[4];
    bool h[4];
    slab_node4q(np[0], np[1], np[2], inv, oi, inv.x >= 0.0f, inv.y >= 0.0f, inv.z >= 0.0f, INFINITY, t, h);
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

struct AtomicAdd3 {
    double* g;
    __device__ __forceinline__ void operator()(int32_t v, d3 a) const {
        unsafeAtomicAdd(g + 3 * (int64_t)v + 0, a.x);
        unsafeAtomicAdd(g + 3 * (int64_t)v + 1, a.y);
        unsafeAtomicAdd(g + 3 * (int64_t)v + 2, a.z);
    }
};

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ---- the two accumulation modes of every gradient / loss target (drt_fixed.h; include/drt_hip.h drt_deterministic) --------------------
// DET = false: float64 atomics into double[] (order-dependent at 1e-16 relative).  DET = true: the SAME pointer is an array of FxCell
// -- one 24-byte cell per float64 element, same element numbers -- and contributions are added as 128-bit integers: any order, same bits.
template <bool DET>
struct GradAdd3 {
    double* g;
    __device__ __forceinline__ void operator()(int32_t v, d3 a) const {
        if (DET) {
            FxCell* c = reinterpret_cast<FxCell*>(g) + 3 * (int64_t)v;
            fx_atomic_add(c, a.x); fx_atomic_add(c + 1, a.y); fx_atomic_add(c + 2, a.z);
        } else {
            AtomicAdd3{g}(v, a);
        }
    }
};
// A thread's share of a kernel's scalar loss: summed per thread, then per wave, one atomic per wave.  The fixed-point form is exact at every
// level, so it does not matter which items a thread, or which threads a wave, happened to get (the path lists are SETS: their order varies).
template <bool DET> struct LossAcc;
template <> struct LossAcc<false> {
    double v = 0.0;
    __device__ __forceinline__ void add(double x) { v += x; }
    __device__ __forceinline__ void flush(double* loss) {
        v = wave_sum(v);
        if ((threadIdx.x & 63) == 0 && v != 0.0) unsafeAtomicAdd(loss, v);
    }
};
template <> struct LossAcc<true> {
    FxAcc a;
    __device__ __forceinline__ void add(double x) { a.add(x); }
    __device__ __forceinline__ void flush(double* loss) {
        a = fx_wave_sum(a);
        if ((threadIdx.x & 63) == 0) fx_atomic_add(reinterpret_cast<FxCell*>(loss), a.v, a.flags);
    }
};
