// drt_sort.h -- stable LSD radix sort of (32-bit key, 32-bit value) pairs, 8 bits per pass, three launches per pass.
// Shared by the LBVH build (Morton codes, drt_build.hip) and the topology tables (edge keys, drt_topology.hip); the
// kernels are `static` so that each translation unit carries its own copy.
#pragma once
#include "drt_scene.h"

// ---- LSD radix sort, 8 bits per pass, stable; three launches per pass -------------------
static __global__ void __launch_bounds__(kSortBlock) k_sort_hist(const uint32_t* __restrict__ keys, int n, int shift,
                                                          uint32_t* __restrict__ hist, int tiles) {
    __shared__ uint32_t cnt[kRadix];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * kSortTile;
    for (int r = 0; r < kSortItems; ++r) {
        const int i = base + r * kSortBlock + threadIdx.x;
        if (i < n) atomicAdd(&cnt[(keys[i] >> shift) & (kRadix - 1)], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * tiles + blockIdx.x] = cnt[threadIdx.x];
}

// Exclusive scan of hist[0..total) in place, one block of 1024 threads.
static __global__ void __launch_bounds__(1024) k_sort_scan(uint32_t* __restrict__ hist, int total) {
    __shared__ uint32_t part[1024];
    const int chunk = (total + 1023) / 1024;
    const int b = threadIdx.x * chunk, e = min(b + chunk, total);
    uint32_t sum = 0;
    for (int i = b; i < e; ++i) sum += hist[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (int i = b; i < e; ++i) {
        const uint32_t h = hist[i];
        hist[i] = run;
        run += h;
    }
}

static __global__ void __launch_bounds__(kSortBlock) k_sort_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ idx_in,
                                                             uint32_t* __restrict__ keys_out, uint32_t* __restrict__ idx_out,
                                                             int n, int shift, const uint32_t* __restrict__ hist, int tiles) {
    constexpr int kWaves = kSortBlock / 64;
    __shared__ uint32_t running[kRadix];
    __shared__ uint32_t wcount[kWaves][kRadix];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    running[tid] = hist[tid * tiles + blockIdx.x];
    for (int w = 0; w < kWaves; ++w) wcount[w][tid] = 0;
    __syncthreads();
    const int base = blockIdx.x * kSortTile;
    for (int r = 0; r < kSortItems; ++r) {
        const int i = base + r * kSortBlock + tid;
        const bool valid = i < n;
        const uint32_t key = valid ? keys_in[i] : 0u;
        const uint32_t val = valid ? idx_in[i] : 0u;
        const uint32_t digit = (key >> shift) & (kRadix - 1);
        unsigned long long peers = __ballot(valid);
        for (int bit = 0; bit < 8; ++bit) {
            const bool set = (digit >> bit) & 1u;
            const unsigned long long bm = __ballot(valid && set);
            peers &= set ? bm : ~bm;
        }
        const uint32_t rank = __popcll(peers & ((1ull << lane) - 1ull));
        if (valid && rank == 0) wcount[wave][digit] = __popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t pos = running[digit] + rank;
            for (int w = 0; w < wave; ++w) pos += wcount[w][digit];
            keys_out[pos] = key;
            idx_out[pos] = val;
        }
        __syncthreads();
        uint32_t add = 0;
        for (int w = 0; w < kWaves; ++w) { add += wcount[w][tid]; wcount[w][tid] = 0; }
        running[tid] += add;
        __syncthreads();
    }
}


// One stable pass on bits [shift, shift + 8): (keys_in, idx_in) -> (keys_out, idx_out).  `hist` holds kRadix * tiles words.
static inline void radix_pass(const uint32_t* keys_in, const uint32_t* idx_in, uint32_t* keys_out, uint32_t* idx_out, int n, int shift,
                              uint32_t* hist, hipStream_t st) {
    const int tiles = (n + kSortTile - 1) / kSortTile;
    k_sort_hist<<<tiles, kSortBlock, 0, st>>>(keys_in, n, shift, hist, tiles);
    k_sort_scan<<<1, 1024, 0, st>>>(hist, kRadix * tiles);
    k_sort_scatter<<<tiles, kSortBlock, 0, st>>>(keys_in, idx_in, keys_out, idx_out, n, shift, hist, tiles);
}
