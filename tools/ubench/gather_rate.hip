// Micro-benchmark: cost model of per-lane gathers from a cache-resident node array (what bounds k_trace).
// Each lane walks a pseudo-random chain through N 64-byte nodes (3.2 MB, L2-resident) and fetches
//   A: its own node with four 16-byte loads        (4 instructions, 64 distinct lines each)  = k_trace today
//   B: its own node's first 16 bytes only          (1 instruction, 64 distinct lines)
//   C: one 16-byte chunk of the node of its QUAD   (1 instruction, 16 distinct lines; lane c loads chunk c)
//   D: one 16-byte chunk of the node of its OCTET  (1 instruction, 8 distinct lines of 128 B: two 64-B nodes)
//   E: two loads of variant C (two nodes per quad step)
// The next index depends on the loaded data (dependent chain, like a traversal); 8 waves/SIMD hide latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct alignas(16) U4 { unsigned x, y, z, w; };

template <int MODE>
__global__ void __launch_bounds__(256, 8) k_gather(const U4* __restrict__ nodes, unsigned n_nodes, int iters, unsigned* out) {
    const unsigned lane = threadIdx.x & 63;
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_nodes;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            const U4 a = nodes[4 * idx], b = nodes[4 * idx + 1], c = nodes[4 * idx + 2], d = nodes[4 * idx + 3];
            acc += a.x ^ b.y ^ c.z ^ d.w;
            idx = (a.x + b.x + c.x + d.x + acc) % n_nodes;
        } else if (MODE == 1) {
            const U4 a = nodes[4 * idx];
            acc += a.x;
            idx = (a.x + acc) % n_nodes;
        } else if (MODE == 2) {
            const unsigned qidx = __shfl(idx, lane & ~3u);          // the quad's node
            const U4 a = nodes[4 * qidx + (lane & 3u)];
            acc += a.x;
            idx = (a.x + acc) % n_nodes;
        } else if (MODE == 3) {
            const unsigned oidx = __shfl(idx, lane & ~7u) & ~1u;    // the octet's pair of nodes = one 128-byte line
            const U4 a = nodes[4 * oidx + (lane & 7u)];
            acc += a.x;
            idx = (a.x + acc) % n_nodes;
        } else if (MODE == 5) {                                   // two 16-byte loads of the own node
            const U4 a = nodes[4 * idx], b = nodes[4 * idx + 1];
            acc += a.x ^ b.y;
            idx = (a.x + b.x + acc) % n_nodes;
        } else if (MODE == 6) {                                   // three
            const U4 a = nodes[4 * idx], b = nodes[4 * idx + 1], c = nodes[4 * idx + 2];
            acc += a.x ^ b.y ^ c.z;
            idx = (a.x + b.x + c.x + acc) % n_nodes;
        } else if (MODE == 7) {                                   // quad-shared node, 4 x 16 B per lane (all four chunks, rotated)
            const unsigned qidx = __shfl(idx, lane & ~3u);
            const U4 a = nodes[4 * qidx + (lane & 3u)], b = nodes[4 * qidx + ((lane + 1) & 3u)], c = nodes[4 * qidx + ((lane + 2) & 3u)], d = nodes[4 * qidx + ((lane + 3) & 3u)];
            acc += a.x ^ b.y ^ c.z ^ d.w;
            idx = (a.x + b.x + c.x + d.x + acc) % n_nodes;
        } else if (MODE == 8) {                                   // own node via ONE 16-byte load + three 16-byte loads of a wave-uniform node
            const U4 a = nodes[4 * idx];
            const unsigned u = __builtin_amdgcn_readfirstlane(idx);
            const U4 b = nodes[4 * u + 1], c = nodes[4 * u + 2], d = nodes[4 * u + 3];
            acc += a.x ^ b.y ^ c.z ^ d.w;
            idx = (a.x + b.x + c.x + d.x + acc) % n_nodes;
        } else if (MODE == 9) {                                   // four 4-byte loads (one dword of each chunk)
            const unsigned* p = reinterpret_cast<const unsigned*>(nodes + 4 * idx);
            const unsigned a = p[0], b = p[4], c = p[8], d = p[12];
            acc += a ^ b ^ c ^ d;
            idx = (a + b + c + d + acc) % n_nodes;
        } else if (MODE == 10) {                                  // four 8-byte loads
            const uint2* p = reinterpret_cast<const uint2*>(nodes + 4 * idx);
            const uint2 a = p[0], b = p[2], c = p[4], d = p[6];
            acc += a.x ^ b.y ^ c.x ^ d.y;
            idx = (a.x + b.x + c.x + d.x + acc) % n_nodes;
        } else if (MODE == 11) {                                  // eight 16-byte loads: two nodes (a 128-byte line)
            const unsigned i2 = idx & ~1u;
            unsigned s = acc;
            #pragma unroll
            for (int k = 0; k < 8; ++k) { const U4 a = nodes[4 * i2 + k]; s += a.x ^ a.w; }
            acc = s;
            idx = (s + acc) % n_nodes;
        } else {
            const unsigned qidx = __shfl(idx, lane & ~3u);
            const U4 a = nodes[4 * qidx + (lane & 3u)];
            const unsigned q2 = (a.y + acc) % n_nodes;
            const unsigned qidx2 = __shfl(q2, lane & ~3u);
            const U4 b = nodes[4 * qidx2 + (lane & 3u)];
            acc += a.x ^ b.x;
            idx = (a.x + b.x + acc) % n_nodes;
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc + idx;
}

int main(int argc, char** argv) {
    const unsigned n_nodes = argc > 1 ? (unsigned)atoi(argv[1]) : 50000;      // 50000 x 64 B = 3.2 MB: L2-resident, L1 misses
    printf("%u nodes = %.1f KB\n", n_nodes, n_nodes * 64 / 1024.0);
    std::vector<U4> h(4 * n_nodes);
    srand(1);
    for (auto& v : h) v = U4{(unsigned)rand(), (unsigned)rand(), (unsigned)rand(), (unsigned)rand()};
    U4* d; unsigned* o;
    const int blocks = 2048, iters = 400;
    hipMalloc(&d, sizeof(U4) * h.size()); hipMalloc(&o, 4 * blocks * 256);
    hipMemcpy(d, h.data(), sizeof(U4) * h.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[12] = {"A own node, 4 x 16 B", "B own node, 1 x 16 B", "C quad node, 1 x 16 B", "D octet line, 1 x 16 B", "E quad node x 2 (dependent)", "F own node, 2 x 16 B", "G own node, 3 x 16 B", "H quad node, 4 x 16 B", "I own 1 x 16 B + uniform 3 x 16 B", "J own node, 4 x 4 B", "K own node, 4 x 8 B", "L own line, 8 x 16 B"};
    for (int mode = 0; mode < 12; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k_gather<0><<<blocks, 256>>>(d, n_nodes, iters, o);
            if (mode == 1) k_gather<1><<<blocks, 256>>>(d, n_nodes, iters, o);
            if (mode == 2) k_gather<2><<<blocks, 256>>>(d, n_nodes, iters, o);
            if (mode == 3) k_gather<3><<<blocks, 256>>>(d, n_nodes, iters, o);
            if (mode == 4) k_gather<4><<<blocks, 256>>>(d, n_nodes, iters, o);
            if (mode == 5) k_gather<5><<<blocks, 256>>>(d, n_nodes, iters, o);
            if (mode == 6) k_gather<6><<<blocks, 256>>>(d, n_nodes, iters, o);
            if (mode == 7) k_gather<7><<<blocks, 256>>>(d, n_nodes, iters, o);
            if (mode == 8) k_gather<8><<<blocks, 256>>>(d, n_nodes, iters, o);
            if (mode == 9) k_gather<9><<<blocks, 256>>>(d, n_nodes, iters, o);
            if (mode == 10) k_gather<10><<<blocks, 256>>>(d, n_nodes, iters, o);
            if (mode == 11) k_gather<11><<<blocks, 256>>>(d, n_nodes, iters, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double wave_steps = (double)blocks * 4 * iters;
            if (rep) printf("%-30s %8.3f ms  %7.1f ns per wave-step per CU (256 CUs)  %6.2f G lane-steps/s\n", names[mode], ms,
                            ms * 1e6 / (wave_steps / 256.0), wave_steps * 64 / ms / 1e6);
        }
    return 0;
}
