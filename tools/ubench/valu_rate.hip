// tools/ubench/valu_rate.hip -- what is the issue rate of plain (non-packed) float32 VALU instructions on MI355X?
// 8 waves per SIMD, each a chain-free stream of v_fma_f32 / v_cndmask / v_perm-like work; reports G wave-instructions/s.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void __launch_bounds__(256, 8) k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {        // 8 independent v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (KIND == 1) { // 8 independent v_pk_fma_f32 on pairs (4 instructions x 2 = 8 fmas) -> counted as 4 instructions
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(*(double*)&x0), "+v"(*(double*)&x2), "+v"(*(double*)&x4), "+v"(*(double*)&x6) : "v"(*(double*)&a), "v"(*(double*)&b));
            } else {                // 8 independent v_max_f32 (VOP2)
                asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                             "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int KIND>
void run(const char* name, int per_iter) {
    float* out; hipMalloc(&out, sizeof(float) * 256 * 2048);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 2048;     // 8 blocks per CU: 8 waves per SIMD
    k<KIND><<<blocks, 256>>>(out, 16, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * 4 * iters * 8 * per_iter;
    printf("%-28s %8.3f ms  %8.1f G wave-instr/s  (%.2f cycles per wave-instruction per SIMD at 2.4 GHz)\n", name, ms, winst / ms / 1e6,
           1024.0 * 2.4e9 / (winst / (ms * 1e-3)));
    hipFree(out);
}
int main() {
    run<0>("v_fma_f32", 8);
    run<1>("v_pk_fma_f32 (2 fmas each)", 4);
    run<2>("v_max_f32", 8);
    return 0;
}
