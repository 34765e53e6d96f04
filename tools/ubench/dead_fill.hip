// How fast can the dead outputs of k_cull be written?  37.7 M rays: out_ori + out_dir (24 B each), mask (3 B), face1, face2 (4 B each)
// = 59 B/ray = 2.2 GB.  Variants: (a) per 64-ray run, all five arrays by the same wave (what k_cull does), persistent grid;
// (b) the same, one block per 256 rays; (c) one streaming pass per array; (d) torch-like single 2.2 GB stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct alignas(16) F4 { float x, y, z, w; };
__device__ __forceinline__ void row(int64_t i0, int lane, double* oo, double* od, uint8_t* m, int32_t* f1, int32_t* f2) {
    const F4 z{0, 0, 0, 0};
    F4* po = (F4*)(oo + 3 * i0); F4* pd = (F4*)(od + 3 * i0);
    po[lane] = z; pd[lane] = z;
    if (lane < 32) { po[64 + lane] = z; pd[64 + lane] = z; }
    if (lane < 12) ((F4*)(m + 3 * i0))[lane] = z;
    if (lane >= 16 && lane < 32) ((F4*)(f2 + i0))[lane - 16] = z;
    if (lane >= 32 && lane < 48) ((F4*)(f1 + i0))[lane - 32] = z;
}
__global__ void __launch_bounds__(256, 8) k_a(int64_t n, double* oo, double* od, uint8_t* m, int32_t* f1, int32_t* f2) {
    for (int64_t base = blockIdx.x * 256ll; base < n; base += gridDim.x * 256ll) row(base + (threadIdx.x & ~63), threadIdx.x & 63, oo, od, m, f1, f2);
}
__global__ void __launch_bounds__(256) k_c(F4* p, int64_t n16) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n16; i += gridDim.x * 256ll) p[i] = F4{0, 0, 0, 0};
}
int main() {
    const int64_t n = 36ll * 1024 * 1024;
    double *oo, *od; uint8_t* m; int32_t *f1, *f2;
    hipMalloc(&oo, n * 24); hipMalloc(&od, n * 24); hipMalloc(&m, n * 3); hipMalloc(&f1, n * 4); hipMalloc(&f2, n * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](const char* name, auto f) {
        f(); hipDeviceSynchronize(); hipEventRecord(a);
        for (int k = 0; k < 10; ++k) f();
        hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-40s %.3f ms  %.2f TB/s\n", name, ms / 10, 59.0 * n / (ms / 10 * 1e-3) / 1e12);
    };
    time("a: five arrays per wave, 2048 blocks", [&] { k_a<<<2048, 256>>>(n, oo, od, m, f1, f2); });
    time("a: five arrays per wave, 1024 blocks", [&] { k_a<<<1024, 256>>>(n, oo, od, m, f1, f2); });
    time("b: five arrays per wave, block per patch", [&] { k_a<<<(unsigned)(n / 256), 256>>>(n, oo, od, m, f1, f2); });
    time("c: one pass per array, 2048 blocks", [&] { k_c<<<2048, 256>>>((F4*)oo, n * 24 / 16); k_c<<<2048, 256>>>((F4*)od, n * 24 / 16); k_c<<<2048, 256>>>((F4*)m, n * 3 / 16); k_c<<<2048, 256>>>((F4*)f1, n * 4 / 16); k_c<<<2048, 256>>>((F4*)f2, n * 4 / 16); });
    time("c2: one pass per array, 16384 blocks", [&] { k_c<<<16384, 256>>>((F4*)oo, n * 24 / 16); k_c<<<16384, 256>>>((F4*)od, n * 24 / 16); k_c<<<16384, 256>>>((F4*)m, n * 3 / 16); k_c<<<16384, 256>>>((F4*)f1, n * 4 / 16); k_c<<<16384, 256>>>((F4*)f2, n * 4 / 16); });
    time("d: hipMemsetAsync x5", [&] { hipMemsetAsync(oo, 0, n * 24, 0); hipMemsetAsync(od, 0, n * 24, 0); hipMemsetAsync(m, 0, n * 3, 0); hipMemsetAsync(f1, 0xFF, n * 4, 0); hipMemsetAsync(f2, 0xFF, n * 4, 0); });
    return 0;
}
