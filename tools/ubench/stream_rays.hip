// Micro-benchmark: what a k_cull-shaped streaming pass can reach on MI355X.  N rays, float64 [N,3] origin and
// direction in (48 B/ray), float64 [N,3] x 2 + 3-byte mask + int32 x 2 out (59 B/ray), no BVH work.
//   R1: read only, per-lane 8-byte loads with a 24-byte stride (how k_cull reads)      -> 48 B/ray
//   R2: read only, 16 bytes per lane, fully coalesced                                  -> 48 B/ray
//   W1: R1 + per-lane strided zero stores (how k_cull writes dead rays)                -> 107 B/ray
//   W2: R1 + full-width 16-byte zero stores                                            -> 107 B/ray
//   W3: write only, full-width                                                         -> 59 B/ray
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

struct alignas(16) Q { unsigned long long a, b; };

template <int MODE>
__global__ void __launch_bounds__(256, 8) k_stream(const double* __restrict__ o, const double* __restrict__ d, int64_t n, double* __restrict__ oo,
                                                   double* __restrict__ od, uint8_t* __restrict__ mask, int32_t* __restrict__ f1, int32_t* __restrict__ f2,
                                                   double* sink) {
    double acc = 0;
    const int lane = threadIdx.x & 63;
    for (int64_t base = blockIdx.x * 256ll; base < n; base += gridDim.x * 256ll) {
        const int64_t i = base + threadIdx.x, i0 = i - lane;
        if (MODE == 0 || MODE == 2 || MODE == 3) {
            acc += o[3 * i] + o[3 * i + 1] + o[3 * i + 2] + d[3 * i] + d[3 * i + 1] + d[3 * i + 2];
        }
        if (MODE == 1) {
            const Q* po = reinterpret_cast<const Q*>(o + 3 * i0);
            const Q* pd = reinterpret_cast<const Q*>(d + 3 * i0);
            Q a = po[lane], b = pd[lane];
            acc += (double)(a.a ^ a.b ^ b.a ^ b.b);
            if (lane < 32) { Q c = po[64 + lane], e = pd[64 + lane]; acc += (double)(c.a ^ e.b); }
        }
        if (MODE == 2) {
            oo[3 * i] = 0; oo[3 * i + 1] = 0; oo[3 * i + 2] = 0; od[3 * i] = 0; od[3 * i + 1] = 0; od[3 * i + 2] = 0;
            mask[3 * i] = 0; mask[3 * i + 1] = 0; mask[3 * i + 2] = 0; f1[i] = -1; f2[i] = -1;
        }
        if (MODE == 3 || MODE == 4) {
            Q* po = reinterpret_cast<Q*>(oo + 3 * i0);
            Q* pd = reinterpret_cast<Q*>(od + 3 * i0);
            po[lane] = Q{0, 0}; pd[lane] = Q{0, 0};
            if (lane < 32) { po[64 + lane] = Q{0, 0}; pd[64 + lane] = Q{0, 0}; }
            if (lane < 12) reinterpret_cast<Q*>(mask + 3 * i0)[lane] = Q{0, 0};
            f1[i] = -1; f2[i] = -1;
        }
    }
    if (acc == 12345.678) *sink = acc;
}

// R4: four rays per thread (all loads issued before the first use), for LOW-occupancy launches: can a streaming pass that
// leaves most wave slots to a co-running latency-bound kernel still reach the HBM rate?
__global__ void __launch_bounds__(256) k_stream4(const double* __restrict__ o, const double* __restrict__ d, int64_t n, double* sink) {
    double acc = 0;
    for (int64_t base = blockIdx.x * 1024ll; base < n; base += gridDim.x * 1024ll) {
        double v[24];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t i = base + r * 256 + threadIdx.x;
            v[6 * r + 0] = o[3 * i]; v[6 * r + 1] = o[3 * i + 1]; v[6 * r + 2] = o[3 * i + 2];
            v[6 * r + 3] = d[3 * i]; v[6 * r + 4] = d[3 * i + 1]; v[6 * r + 5] = d[3 * i + 2];
        }
#pragma unroll
        for (int k = 0; k < 24; ++k) acc += v[k];
    }
    if (acc == 12345.678) *sink = acc;
}

int main() {
    const int64_t n = 36ll * 1024 * 1024;     // one sub-batch of the bench (36 views of 1024 x 1024)
    double *o, *d, *oo, *od, *sink; uint8_t* mask; int32_t *f1, *f2;
    hipMalloc(&o, n * 24); hipMalloc(&d, n * 24); hipMalloc(&oo, n * 24); hipMalloc(&od, n * 24); hipMalloc(&mask, n * 3);
    hipMalloc(&f1, n * 4); hipMalloc(&f2, n * 4); hipMalloc(&sink, 8);
    hipMemset(o, 0, n * 24); hipMemset(d, 0, n * 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[5] = {"R1 strided 8-B reads", "R2 coalesced 16-B reads", "W1 R1 + strided stores", "W2 R1 + 16-B stores", "W3 16-B stores only"};
    const double bytes[5] = {48, 48, 107, 107, 59};
    for (int grid : {2048, 8192, 36 * 4096})
        for (int mode = 0; mode < 5; ++mode)
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k_stream<0><<<grid, 256>>>(o, d, n, oo, od, mask, f1, f2, sink);
                if (mode == 1) k_stream<1><<<grid, 256>>>(o, d, n, oo, od, mask, f1, f2, sink);
                if (mode == 2) k_stream<2><<<grid, 256>>>(o, d, n, oo, od, mask, f1, f2, sink);
                if (mode == 3) k_stream<3><<<grid, 256>>>(o, d, n, oo, od, mask, f1, f2, sink);
                if (mode == 4) k_stream<4><<<grid, 256>>>(o, d, n, oo, od, mask, f1, f2, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("grid %6d  %-26s %7.3f ms  %6.2f TB/s\n", grid, names[mode], ms, bytes[mode] * n / ms / 1e9);
            }
    for (int grid : {256, 512, 1024, 2048})
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            k_stream4<<<grid, 256>>>(o, d, n, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("grid %6d  R4 four rays per thread    %7.3f ms  %6.2f TB/s\n", grid, ms, 48.0 * n / ms / 1e9);
        }
    for (int grid : {256, 512, 1024})
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            k_stream<0><<<grid, 256>>>(o, d, n, oo, od, mask, f1, f2, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("grid %6d  R1 one ray per thread      %7.3f ms  %6.2f TB/s\n", grid, ms, 48.0 * n / ms / 1e9);
        }
    return 0;
}
