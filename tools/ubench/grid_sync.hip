// How long is a grid-wide barrier of a cooperative launch (cooperative_groups::this_grid().sync()) on this chip, by grid shape?
// (Priced for a whole remesh step as ONE persistent kernel: ~14 barriers per evaluate / claim / apply round against ~16 launches.)
// build: hipcc --offload-arch=gfx950 -O3 -o grid_sync grid_sync.hip ; run: ./grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ void k_syncs(int n, int* sink) {
    cg::grid_group g = cg::this_grid();
    int acc = 0;
    for (int i = 0; i < n; ++i) { acc += i; g.sync(); }
    if (blockIdx.x == 0 && threadIdx.x == 0) *sink = acc;
}
// the same barrier hand-made: one atomic counter per generation, spin on a volatile read (needs every block resident: cooperative launch)
__global__ void k_syncs_atomic(int n, unsigned* bar, int* sink) {
    int acc = 0;
    for (int i = 0; i < n; ++i) {
        acc += i;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned target = (unsigned)(i + 1) * gridDim.x;
            atomicAdd(bar, 1u);
            while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *sink = acc;
}
int main() {
    int* sink; hipMalloc(&sink, 4);
    unsigned* bar; hipMalloc(&bar, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int shapes[][2] = {{256, 256}, {256, 1024}, {512, 256}, {1024, 256}, {2048, 256}};
    for (auto& sh : shapes) {
        int blocks = sh[0], threads = sh[1];
        for (int variant = 0; variant < 2; ++variant) {
            float best[2] = {1e9f, 1e9f};
            for (int pass = 0; pass < 2; ++pass) {
                int n = pass == 0 ? 10 : 1010;
                for (int rep = 0; rep < 5; ++rep) {
                    hipMemset(bar, 0, 4);
                    void* args1[] = {&n, &sink};
                    void* args2[] = {&n, &bar, &sink};
                    hipEventRecord(a);
                    hipError_t e = hipLaunchCooperativeKernel(variant == 0 ? (void*)k_syncs : (void*)k_syncs_atomic, dim3(blocks), dim3(threads), variant == 0 ? args1 : args2, 0, 0);
                    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
                    hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b);
                    if (ms < best[pass]) best[pass] = ms;
                }
            }
            printf("%4d blocks x %4d threads, %s: %.2f us per barrier (launch + 10 barriers %.1f us)\n", blocks, threads,
                   variant == 0 ? "cooperative_groups grid.sync()" : "atomic counter + spin       ", 1e3 * (best[1] - best[0]) / 1000.0, 1e3 * best[0]);
        }
    }
    return 0;
}
