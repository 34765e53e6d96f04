// drt_topology.hip -- what Scene.update_mesh derives from the face list, on the device: the unique-edge tables of
// Scene.init_edge (reference DiffRender.py:338-355: `Edges`, `E2F`, `mean_len`, which the reference gets from trimesh's
// host-side group_rows / edges_face) and the 1 -> 4 midpoint refinement of a level-of-detail step.
//
// Edge tables = one stable sort of the 3F directed-edge rows by (min vertex, max vertex): LSD radix on `hi` then on `lo`
// with the LBVH's sort passes (drt_sort.h).  Order as pinned by drt_amd/mesh_io.group_rows_pairs (and by the golden
// hand_topology.npz): edges ascend by (lo, hi); of the two faces of an edge the one with the lower directed-edge row comes
// first.  Watertightness (DiffRender.py:305) = every key occurs exactly twice; reported through *d_status.
#include "drt_device.h"
#include "drt_sort.h"

namespace {

struct TopoWork {            // carved out of the caller's workspace
    uint32_t *keys[2], *idx[2], *lo, *hist;
    double* partial;         // [kPartials] per-block sums of the directed-edge lengths
};
constexpr int kPartials = 1024;

size_t topo_bytes(int64_t n_faces) {
    const int64_t rows = 3 * (n_faces > 0 ? n_faces : 1);
    const int64_t tiles = (rows + kSortTile - 1) / kSortTile;
    return sizeof(uint32_t) * (size_t)(5 * rows + kRadix * tiles) + sizeof(double) * kPartials + 256;
}

TopoWork carve(void* ws, int64_t n_faces) {
    const int64_t rows = 3 * (n_faces > 0 ? n_faces : 1);
    const int64_t tiles = (rows + kSortTile - 1) / kSortTile;
    TopoWork w;
    char* p = static_cast<char*>(ws);
    w.partial = reinterpret_cast<double*>(p); p += sizeof(double) * kPartials;
    uint32_t* q = reinterpret_cast<uint32_t*>(p);
    w.keys[0] = q; q += rows; w.keys[1] = q; q += rows; w.idx[0] = q; q += rows; w.idx[1] = q; q += rows; w.lo = q; q += rows;
    w.hist = q; (void)tiles;
    return w;
}

}  // namespace

// Row r = 3 f + j is the directed edge (F[f][j], F[f][(j + 1) % 3]) -- trimesh's `edges` order (faces[:, [0,1,1,2,2,0]]).
__global__ void __launch_bounds__(256) k_edge_rows(const int64_t* __restrict__ faces, int64_t n_rows, const double* __restrict__ verts, int64_t n_verts,
                                                   uint32_t* __restrict__ key_hi, uint32_t* __restrict__ key_lo, uint32_t* __restrict__ idx,
                                                   double* __restrict__ partial, int32_t* status) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t r = blockIdx.x * 256ll + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * 256) {
        const int64_t f = r / 3;
        const int j = (int)(r - 3 * f);
        int64_t a = faces[3 * f + j], b = faces[3 * f + (j == 2 ? 0 : j + 1)];
        if (a < 0 || a >= n_verts || b < 0 || b >= n_verts) {     // a face list that points outside the vertex array: reported (status bit 2),
            atomicOr(status, 2);                                  // never dereferenced -- torch's indexing, which this replaces, raises too
            a = b = 0;
        }
        key_lo[r] = (uint32_t)(a < b ? a : b);
        key_hi[r] = (uint32_t)(a < b ? b : a);
        idx[r] = (uint32_t)r;
        const d3 d = load_d3(verts, a) - load_d3(verts, b);
        acc += sqrt((d.x * d.x + d.y * d.y) + d.z * d.z);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);     // fixed order: deterministic
}

__global__ void k_gather_u32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t* __restrict__ dst, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// After the sort: pair e = sorted rows 2e, 2e + 1.  Watertight <=> both rows carry the same (lo, hi) and the pair's key is
// strictly above the previous pair's.  One block also folds the length partials into the mean.
__global__ void __launch_bounds__(256) k_edge_pairs(const int64_t* __restrict__ faces, const uint32_t* __restrict__ sorted_rows,
                                                    int64_t n_edges, int64_t* __restrict__ edges, int64_t* __restrict__ e2f,
                                                    int32_t* __restrict__ row2edge, const double* __restrict__ partial, int n_partial,
                                                    int64_t n_rows, double* __restrict__ mean_len, int32_t* status) {
    auto ends = [&](uint32_t r, int64_t& lo, int64_t& hi) {
        const int64_t f = r / 3;
        const int j = (int)(r - 3 * f);
        const int64_t a = faces[3 * f + j], b = faces[3 * f + (j == 2 ? 0 : j + 1)];
        lo = a < b ? a : b; hi = a < b ? b : a;
    };
    const int64_t e = blockIdx.x * 256ll + threadIdx.x;
    if (e < n_edges) {
        const uint32_t r0 = sorted_rows[2 * e], r1 = sorted_rows[2 * e + 1];
        int64_t lo0, hi0, lo1, hi1;
        ends(r0, lo0, hi0); ends(r1, lo1, hi1);
        bool ok = lo0 == lo1 && hi0 == hi1 && lo0 != hi0;
        if (e > 0) {
            int64_t lop, hip;
            ends(sorted_rows[2 * e - 1], lop, hip);
            ok = ok && (lop < lo0 || (lop == lo0 && hip < hi0));
        }
        if (!ok) atomicOr(status, 1);
        edges[2 * e] = lo0; edges[2 * e + 1] = hi0;
        const int64_t f0 = r0 / 3, f1 = r1 / 3;
        for (int k = 0; k < 3; ++k) { e2f[6 * e + k] = faces[3 * f0 + k]; e2f[6 * e + 3 + k] = faces[3 * f1 + k]; }
        if (row2edge) { row2edge[r0] = (int32_t)e; row2edge[r1] = (int32_t)e; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < n_partial; ++k) s += partial[k];
        *mean_len = n_rows > 0 ? s / (double)n_rows : 0.0;
    }
}

// 1 -> 4 midpoint refinement (drt_amd/mesh_io.subdivide_midpoint on the device): vertex V + e = midpoint of unique edge e
// (optionally rounded through float32 like a PLY round trip); face f -> (v0, m01, m20), (m01, v1, m12), (m20, m12, v2),
// (m01, m12, m20) at rows 4f .. 4f + 3.
__global__ void __launch_bounds__(256) k_subdivide(const int64_t* __restrict__ faces, int64_t n_faces, const double* __restrict__ verts,
                                                   int64_t n_verts, const int64_t* __restrict__ edges, int64_t n_edges,
                                                   const int32_t* __restrict__ row2edge, int round_f32,
                                                   int64_t* __restrict__ faces_out, double* __restrict__ verts_out) {
    const int64_t t = blockIdx.x * 256ll + threadIdx.x;
    if (t < n_verts) store_d3(verts_out, t, load_d3(verts, t));
    if (t < n_edges) {
        const d3 a = load_d3(verts, edges[2 * t]), b = load_d3(verts, edges[2 * t + 1]);
        d3 m{0.5 * (a.x + b.x), 0.5 * (a.y + b.y), 0.5 * (a.z + b.z)};
        if (round_f32) m = to_f64(to_f32(m));
        store_d3(verts_out, n_verts + t, m);
    }
    if (t < n_faces) {
        const int64_t v0 = faces[3 * t], v1 = faces[3 * t + 1], v2 = faces[3 * t + 2];
        const int64_t m0 = n_verts + row2edge[3 * t], m1 = n_verts + row2edge[3 * t + 1], m2 = n_verts + row2edge[3 * t + 2];
        int64_t* o = faces_out + 12 * t;
        o[0] = v0; o[1] = m0; o[2] = m2;
        o[3] = m0; o[4] = v1; o[5] = m1;
        o[6] = m2; o[7] = m1; o[8] = v2;
        o[9] = m0; o[10] = m1; o[11] = m2;
    }
}

extern "C" {

int64_t drt_edge_tables_workspace(int64_t n_faces) { return (int64_t)topo_bytes(n_faces); }

int drt_edge_tables(const int64_t* d_faces, int64_t n_faces, const double* d_verts, int64_t n_verts, void* d_workspace,
                    int64_t* d_edges, int64_t* d_e2f, int32_t* d_row2edge, double* d_mean_len, int32_t* d_status, void* stream) {
    if (n_faces < 0 || n_verts < 0 || n_faces > (int64_t)1 << 29 || n_verts > (int64_t)1 << 31) return fail(DRT_E_INVALID, "mesh size out of range");
    if (!d_mean_len || !d_status) return fail(DRT_E_INVALID, "null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(d_status, 0, sizeof(int32_t), st));
    const int64_t rows = 3 * n_faces, n_edges = rows / 2;
    if (rows == 0) { HIP_TRY(hipMemsetAsync(d_mean_len, 0, sizeof(double), st)); return DRT_OK; }
    if (!d_faces || !d_verts || !d_workspace || !d_edges || !d_e2f) return fail(DRT_E_INVALID, "null pointer argument");
    if (n_verts == 0) return fail(DRT_E_INVALID, "faces without vertices");
    const TopoWork w = carve(d_workspace, n_faces);
    const int blocks = (int)((rows + 255) / 256 < kPartials ? (rows + 255) / 256 : kPartials);
    k_edge_rows<<<blocks, 256, 0, st>>>(d_faces, rows, d_verts, n_verts, w.keys[0], w.lo, w.idx[0], w.partial, d_status);
    int bits = 1;
    while (((int64_t)1 << bits) < n_verts) ++bits;
    int cur = 0;
    for (int shift = 0; shift < bits; shift += 8) {                 // by hi
        radix_pass(w.keys[cur], w.idx[cur], w.keys[cur ^ 1], w.idx[cur ^ 1], (int)rows, shift, w.hist, st);
        cur ^= 1;
    }
    k_gather_u32<<<(unsigned)((rows + 255) / 256), 256, 0, st>>>(w.lo, w.idx[cur], w.keys[cur], rows);
    for (int shift = 0; shift < bits; shift += 8) {                 // then, stably, by lo
        radix_pass(w.keys[cur], w.idx[cur], w.keys[cur ^ 1], w.idx[cur ^ 1], (int)rows, shift, w.hist, st);
        cur ^= 1;
    }
    if (rows % 2) {                                                  // an odd number of directed edges cannot pair up
        k_edge_pairs<<<1, 256, 0, st>>>(d_faces, w.idx[cur], 0, d_edges, d_e2f, d_row2edge, w.partial, blocks, rows, d_mean_len, d_status);
        int32_t one = 1;
        HIP_TRY(hipMemcpyAsync(d_status, &one, sizeof(one), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        return DRT_OK;
    }
    k_edge_pairs<<<(unsigned)((n_edges + 255) / 256), 256, 0, st>>>(d_faces, w.idx[cur], n_edges, d_edges, d_e2f, d_row2edge, w.partial, blocks, rows,
                                                                     d_mean_len, d_status);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_subdivide_midpoint(const int64_t* d_faces, int64_t n_faces, const double* d_verts, int64_t n_verts, const int64_t* d_edges,
                           int64_t n_edges, const int32_t* d_row2edge, int round_f32, int64_t* d_faces_out, double* d_verts_out, void* stream) {
    if (n_faces < 0 || n_verts < 0 || n_edges < 0) return fail(DRT_E_INVALID, "negative size");
    if (n_faces == 0 && n_verts == 0) return DRT_OK;
    if ((n_faces && (!d_faces || !d_row2edge || !d_faces_out)) || (n_verts && (!d_verts || !d_verts_out)) || (n_edges && !d_edges))
        return fail(DRT_E_INVALID, "null pointer argument");
    int64_t m = n_faces > n_verts ? n_faces : n_verts;
    if (n_edges > m) m = n_edges;
    k_subdivide<<<(unsigned)((m + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_faces, n_faces, d_verts, n_verts, d_edges, n_edges, d_row2edge, round_f32,
                                                                               d_faces_out, d_verts_out);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

}  // extern "C"
