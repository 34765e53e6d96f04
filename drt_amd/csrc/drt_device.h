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
