// Micro-benchmark: float64 atomic scatter into a small hot array (the vertex-gradient pattern),
// agent scope (one copy) vs per-XCD private copies with workgroup-scope (L2-local) atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int MODE>
__global__ void k_scatter(const int* __restrict__ idx, int n, double* g, int stride, unsigned* xcc_hist) {
    const unsigned x = xcc_id();
    if (threadIdx.x == 0 && xcc_hist) atomicAdd(&xcc_hist[x & 15], 1u);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int v = idx[i];
        if (MODE == 0) {
            unsafeAtomicAdd(g + v, 1.0);
        } else if (MODE == 1) {
            __hip_atomic_fetch_add(g + (size_t)(x & 7) * stride + v, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_fetch_add(g + (size_t)(x & 7) * stride + v, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main() {
    const int V = 75378, n = 6 * 1000 * 1000;
    std::vector<int> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = rand() % V;
    int* d_idx; double* d_g; unsigned* d_hist;
    hipMalloc(&d_idx, sizeof(int) * n); hipMalloc(&d_g, sizeof(double) * 8 * V); hipMalloc(&d_hist, 64);
    hipMemcpy(d_idx, h.data(), sizeof(int) * n, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(d_g, 0, sizeof(double) * 8 * V); hipMemset(d_hist, 0, 64);
            hipEventRecord(e0);
            if (mode == 0) k_scatter<0><<<1024, 256>>>(d_idx, n, d_g, V, d_hist);
            if (mode == 1) k_scatter<1><<<1024, 256>>>(d_idx, n, d_g, V, d_hist);
            if (mode == 2) k_scatter<2><<<1024, 256>>>(d_idx, n, d_g, V, d_hist);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<double> g(8 * V); unsigned hist[16];
            hipMemcpy(g.data(), d_g, sizeof(double) * 8 * V, hipMemcpyDeviceToHost);
            hipMemcpy(hist, d_hist, 64, hipMemcpyDeviceToHost);
            double tot = 0; for (double x : g) tot += x;
            printf("mode %d rep %d: %.3f ms  %.2f G atomics/s  sum=%.0f (expect %d)  xcc blocks:", mode, rep, ms, n / ms / 1e6, tot, n);
            for (int k = 0; k < 8; ++k) printf(" %u", hist[k]);
            printf("\n");
        }
    }
    return 0;
}
