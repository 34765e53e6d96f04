// tools/ubench/atomic_pattern.hip -- does the ADDRESS PATTERN of a wave's float64 atomics matter?  6 M global_atomic_add_f64 into a
// 75 k-entry array: (a) random indices, (b) the same multiset sorted (lane-consecutive addresses, many lanes per cache line),
// (c) random order of 64-entry runs of consecutive indices (each wave instruction touches 4 lines), (d) 3-component rows (stride 24 B).
// build: hipcc --offload-arch=gfx950 -O3 -o atomic_pattern atomic_pattern.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_scatter(const int* __restrict__ idx, int n, double* g) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) unsafeAtomicAdd(g + idx[i], 1.0);
}
__global__ void k_scatter3(const int* __restrict__ idx, int n, double* g) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double* p = g + 3 * (size_t)idx[i];
        unsafeAtomicAdd(p, 1.0); unsafeAtomicAdd(p + 1, 1.0); unsafeAtomicAdd(p + 2, 1.0);
    }
}
int main() {
    const int V = 75378, n = 6 * 1000 * 1000;
    std::vector<int> h(n);
    srand(1);
    int* d_idx; double* d_g;
    hipMalloc(&d_idx, sizeof(int) * n); hipMalloc(&d_g, sizeof(double) * 3 * V);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[5] = {"random", "sorted", "runs of 64 consecutive", "random, 3 components per row", "runs of 64, 3 components per row"};
    for (int mode = 0; mode < 5; ++mode) {
        if (mode == 0 || mode == 3) for (int i = 0; i < n; ++i) h[i] = rand() % (mode == 3 ? V : V);
        if (mode == 1) std::sort(h.begin(), h.end());
        if (mode == 2 || mode == 4) for (int i = 0; i < n; i += 64) { const int b = rand() % (V - 64); for (int k = 0; k < 64 && i + k < n; ++k) h[i + k] = b + k; }
        hipMemcpy(d_idx, h.data(), sizeof(int) * n, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d_g, 0, sizeof(double) * 3 * V);
            hipEventRecord(e0);
            if (mode >= 3) k_scatter3<<<2048, 256>>>(d_idx, n, d_g); else k_scatter<<<2048, 256>>>(d_idx, n, d_g);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-36s %8.3f ms  %8.2f G atomics/s\n", names[mode], ms, (mode >= 3 ? 3.0 : 1.0) * n / ms / 1e6);
        }
    }
    return 0;
}
