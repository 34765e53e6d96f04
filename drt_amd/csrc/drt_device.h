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
