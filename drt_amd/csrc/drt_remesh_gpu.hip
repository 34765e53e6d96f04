// drt_remesh_gpu.hip -- the geometric kernels of the isotropic remesher on the device (drt_amd/remesh_gpu.py drives them).
//
// The reference re-tessellates the mesh between passes with MeshLab's "Isotropic Explicit Remeshing" (reference optim.py:12-52);
// csrc/drt_remesh.cpp is the host restatement of that algorithm (Botsch & Kobbelt 2004: split > 4/3 L, collapse < 4/5 L, flip
// towards valence 6, tangential relaxation, projection onto the input surface) with sequential edge operations.  This file is its
// data-parallel form: every edge operation is EVALUATED for all candidates at once against the mesh as it stands; the candidates
// that pass claim the vertices they would write with an atomicMin of their priority (collapses: shorter length class first, a hash
// inside a class; flips: a hash of the directed-edge slot), and those that no higher priority contests anywhere in what they read or write are APPLIED --
// they share no face and do not change each other's premises, so they commute; the driver repeats evaluate / claim / apply on the
// updated mesh until a round applies nothing.  Same acceptance rules as the host version (link condition, valence limits, consensus-normal fold
// test, maximum edge length, MaxSurfDist through closest-point queries on the scene's tree), same split patterns, same
// relaxation and roll-back; a different, but equally legitimate, visiting order -- so the two are compared statistically
// (tests/test_gpu_remesh.py), the host version being the checker.
//
// Layout: faces int64 [F,3] (what drt_edge_tables takes), vertices float64 [V,3]; vertex -> incident faces as a CSR pair
// (vf_start int64 [V+1], vf_face int64 [3F]: face ids grouped by vertex, ascending -- drt_rm_vertex_faces: count, scan, fill, sort each run).
#include "drt_device.h"

namespace {

constexpr int kMaxRing = 32;          // neighbours of one vertex an edge operation looks at (a vertex with more is left alone)

__device__ __forceinline__ double len3(d3 a) { return sqrt(dot(a, a)); }
__device__ __forceinline__ d3 ldv(const double* __restrict__ V, int64_t i) { return load_d3(V, i); }
__device__ __forceinline__ d3 tri_normal(d3 a, d3 b, d3 c) { return cross(b - a, c - a); }

// cosine between a face normal n and the consensus (sum of the area-weighted vertex normals) of its corners; 1 without a consensus
__device__ __forceinline__ double agreement(d3 n, const double* __restrict__ vn, int64_t a, int64_t b, int64_t c) {
    const d3 r = (ldv(vn, a) + ldv(vn, b)) + ldv(vn, c);
    const double ln = len3(n), lr = len3(r);
    return ln > 0 && lr > 0 ? dot(n, r) / (ln * lr) : (ln > 0 ? 1.0 : -1.0);
}
__device__ __forceinline__ bool acceptable(double before, double after) { return after >= 0.3 || after >= before; }

// The ring of a vertex without a list: in a closed oriented manifold the faces around v name each neighbour of v exactly once as "the
// vertex that follows v in the face", so a walk over v's faces IS a walk over its neighbours, and the valence is the number of faces.
// (Round 6: the rings used to be collected into per-thread arrays with a linear-search de-duplication -- 560 bytes of scratch per lane in
// the evaluation and the claim kernels.)
__device__ __forceinline__ int64_t next_in_face(const int64_t* __restrict__ F, int64_t f, int64_t v) {
    const int64_t x = F[3 * f], y = F[3 * f + 1], z = F[3 * f + 2];
    return x == v ? y : (y == v ? z : x);
}
__device__ __forceinline__ bool adjacent(int64_t u, int64_t v, const int64_t* __restrict__ F, const int64_t* __restrict__ vf_start,
                                         const int64_t* __restrict__ vf_face) {
    for (int64_t q = vf_start[v]; q < vf_start[v + 1]; ++q) {
        const int64_t f = vf_face[q];
        if (F[3 * f] == u || F[3 * f + 1] == u || F[3 * f + 2] == u) return true;
    }
    return false;
}

// ---- split -------------------------------------------------------------------------------------------------------------------
// Long edges by directed-edge slot c = 3 f + k (the lo -> hi slot of an edge speaks for it, as in the collapse and the flip): no sorted edge
// table.  k_rm_split_mark flags them; the caller's prefix sum of the flags numbers the new vertices; k_rm_split_assign hands every slot --
// the owner and the slot across the edge, found in the vertex -> face list of the edge's far end -- the midpoint vertex of its edge or -1.
__global__ void k_rm_split_mark(const int64_t* __restrict__ F, int64_t n_faces, const double* __restrict__ V, double max_len, uint8_t* __restrict__ flag) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= 3 * n_faces) return;
    const int64_t f = c / 3;
    const int64_t a = F[c], b = F[3 * f + (int)(c - 3 * f + 1) % 3];
    flag[c] = a >= 0 && a < b && len3(ldv(V, a) - ldv(V, b)) > max_len;
}
__global__ void k_rm_split_assign(const int64_t* __restrict__ F, int64_t n_faces, const int64_t* __restrict__ vf_start, const int64_t* __restrict__ vf_face,
                                  const uint8_t* __restrict__ flag, const int64_t* __restrict__ rank /* inclusive prefix sum of flag */, int64_t n_verts,
                                  int64_t* __restrict__ mid_of_slot) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= 3 * n_faces) return;
    const int64_t f = c / 3;
    const int64_t a = F[c], b = F[3 * f + (int)(c - 3 * f + 1) % 3];
    if (a < 0) { mid_of_slot[c] = -1; return; }
    if (!(a < b)) return;                                      // (written by the owner of the edge)
    const int64_t mid = flag[c] ? n_verts + rank[c] - 1 : -1;
    mid_of_slot[c] = mid;
    for (int64_t s = vf_start[b]; s < vf_start[b + 1]; ++s) {
        const int64_t g = vf_face[s];
        for (int k = 0; k < 3; ++k)
            if (F[3 * g + k] == b && F[3 * g + (k + 1) % 3] == a) { mid_of_slot[3 * g + k] = mid; return; }
    }
}
// the midpoints, written behind the old vertices: (V[lo] + V[hi]) * 0.5, the host version's bits
__global__ void k_rm_split_midpoints(const int64_t* __restrict__ F, int64_t n_faces, const int64_t* __restrict__ mid_of_slot, double* __restrict__ V) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= 3 * n_faces) return;
    const int64_t f = c / 3;
    const int64_t a = F[c], b = F[3 * f + (int)(c - 3 * f + 1) % 3], m = mid_of_slot[c];
    if (a >= 0 && a < b && m >= 0) store_d3(V, m, (ldv(V, a) + ldv(V, b)) * 0.5);
}
// per face: the midpoint vertex of each of its edges (-1: not split) -> number of faces it becomes
__global__ void k_rm_split_count(const int64_t* __restrict__ mid_of_slot, int64_t n_faces, int64_t* __restrict__ count) {
    const int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (f >= n_faces) return;
    int n = 0;
    for (int k = 0; k < 3; ++k) n += mid_of_slot[3 * f + k] >= 0;
    count[f] = n + 1;
}
// the patterns of drt_remesh.cpp::split_long_edges: 1 -> 2 faces, 2 -> 3 with the shorter diagonal, 3 -> 4
__global__ void k_rm_split_faces(const int64_t* __restrict__ F, const int64_t* __restrict__ mid_of_slot,
                                 const double* __restrict__ V /* midpoints already appended */, int64_t n_faces,
                                 const int64_t* __restrict__ offset /* exclusive prefix sum of the counts */, int64_t* __restrict__ out) {
    const int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (f >= n_faces) return;
    const int64_t v[3] = {F[3 * f], F[3 * f + 1], F[3 * f + 2]};
    const int64_t m[3] = {mid_of_slot[3 * f], mid_of_slot[3 * f + 1], mid_of_slot[3 * f + 2]};
    const int n = (m[0] >= 0) + (m[1] >= 0) + (m[2] >= 0);
    int64_t* o = out + 3 * offset[f];
    auto put = [&](int t, int64_t a, int64_t b, int64_t c) { o[3 * t] = a; o[3 * t + 1] = b; o[3 * t + 2] = c; };
    if (n == 0) { put(0, v[0], v[1], v[2]); return; }
    if (n == 3) { put(0, v[0], m[0], m[2]); put(1, m[0], v[1], m[1]); put(2, m[2], m[1], v[2]); put(3, m[0], m[1], m[2]); return; }
    if (n == 1) {
        const int r = m[0] >= 0 ? 0 : (m[1] >= 0 ? 1 : 2);                      // split edge (a, b)
        const int64_t a = v[r], b = v[(r + 1) % 3], c = v[(r + 2) % 3], mab = m[r];
        put(0, a, mab, c); put(1, mab, b, c);
        return;
    }
    const int r = m[0] < 0 ? 1 : (m[1] < 0 ? 2 : 0);                            // the unsplit edge is (c, a); split (a, b) and (b, c)
    const int64_t a = v[r], b = v[(r + 1) % 3], c = v[(r + 2) % 3], mab = m[r], mbc = m[(r + 1) % 3];
    put(0, mab, b, mbc);
    if (len3(ldv(V, a) - ldv(V, mbc)) <= len3(ldv(V, mab) - ldv(V, c))) { put(1, a, mab, mbc); put(2, a, mbc, c); }
    else { put(1, a, mab, c); put(2, mab, mbc, c); }
}

// ---- vertex normals (area-weighted, summed in CSR order: deterministic) ------------------------------------------------------------
__global__ void k_rm_vertex_normals(const int64_t* __restrict__ F, const double* __restrict__ V, const int64_t* __restrict__ vf_start,
                                    const int64_t* __restrict__ vf_face, int64_t n_verts, double* __restrict__ vn) {
    const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (v >= n_verts) return;
    d3 s{0, 0, 0};
    for (int64_t q = vf_start[v]; q < vf_start[v + 1]; ++q) {
        const int64_t f = vf_face[q];
        s += tri_normal(ldv(V, F[3 * f]), ldv(V, F[3 * f + 1]), ldv(V, F[3 * f + 2]));
    }
    store_d3(vn, v, s);
}

// ---- vertex -> incident faces ---------------------------------------------------------------------------------------------------
// Every evaluate / claim / apply round needs this table of the mesh as it stands; a stable sort of the 3 F corner indices (torch.argsort:
// a merge sort of a dozen launches, plus a gather and a search) took 160 us a time, 35 times per remesh call.  Count, scan, fill with an
// atomic cursor, then each vertex puts its handful of faces in ascending order: four small launches, and the same table bit for bit
// (ascending runs are what make the vertex normals -- float64 sums in run order -- and so the whole remesh deterministic).
__global__ void k_rm_vf_count(const int64_t* __restrict__ F, int64_t n_corners, int32_t* count, const int32_t* __restrict__ live) {
    if (live && !*live) return;                   // (the step's rounds are over: drt_rm_round_end)
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= n_corners) return;
    const int64_t v = F[c];
    if (v >= 0) atomicAdd(&count[v], 1);                       // (a killed face holds -1: in nobody's run)
}
// exclusive scan of count[0 .. n) -> start[0 .. n] by ONE workgroup (n = V is tens of thousands): tiles of 8192, eight counts per thread
// (two 16-byte loads), a 32-bit shuffle scan inside each wavefront, the sixteen wavefront totals through LDS, the running total carried
// from tile to tile in 64 bits
__global__ void __launch_bounds__(1024) k_rm_vf_scan(const int32_t* __restrict__ count, int64_t n, int64_t* __restrict__ start, const int32_t* __restrict__ live) {
    if (live && !*live) return;                   // (the step's rounds are over: drt_rm_round_end)
    __shared__ int32_t wsum[2][16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int64_t carry = 0;
    int buf = 0;
    for (int64_t base = 0; base < n; base += 8192, buf ^= 1) {
        const int64_t i = base + 8 * t;
        int32_t c[8];
        if (i + 8 <= n) {
            const int4 lo = *reinterpret_cast<const int4*>(count + i), hi = *reinterpret_cast<const int4*>(count + i + 4);
            c[0] = lo.x; c[1] = lo.y; c[2] = lo.z; c[3] = lo.w; c[4] = hi.x; c[5] = hi.y; c[6] = hi.z; c[7] = hi.w;
        } else {
            for (int k = 0; k < 8; ++k) c[k] = i + k < n ? count[i + k] : 0;
        }
        int32_t s = 0;
        for (int k = 0; k < 8; ++k) s += c[k];            // (a tile holds at most 8192 x 3 F / V corners: far inside 32 bits)
        int32_t x = s;
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        if (lane == 63) wsum[buf][w] = x;
        __syncthreads();                                    // (one barrier per tile: the two halves of wsum alternate)
        int32_t before = 0, total = 0;
        for (int j = 0; j < 16; ++j) { const int32_t q = wsum[buf][j]; total += q; if (j < w) before += q; }
        int64_t run = carry + before + (x - s);
        if (i + 8 <= n) {
            for (int k = 0; k < 8; ++k) { start[i + k] = run; run += c[k]; }
        } else {
            for (int k = 0; k < 8; ++k) if (i + k < n) { start[i + k] = run; run += c[k]; }
        }
        carry += total;
    }
    if (t == 0) start[n] = carry;
}
__global__ void k_rm_vf_fill(const int64_t* __restrict__ F, int64_t n_corners, const int64_t* __restrict__ start, int32_t* count, int64_t* __restrict__ vf_face, const int32_t* __restrict__ live) {
    if (live && !*live) return;                   // (the step's rounds are over: drt_rm_round_end)
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= n_corners) return;
    const int64_t v = F[c];
    if (v >= 0) vf_face[start[v] + (atomicSub(&count[v], 1) - 1)] = c / 3;
}
// each vertex: its run ascending (insertion sort of a handful), then -- when asked -- its area-weighted normal, summed in that order
__global__ void k_rm_vf_sort(const int64_t* __restrict__ F, const int64_t* __restrict__ start, int64_t* __restrict__ vf_face, int64_t n_verts,
                             const double* __restrict__ V, double* __restrict__ vn, const int32_t* __restrict__ live) {
    if (live && !*live) return;                   // (the step's rounds are over: drt_rm_round_end)
    const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (v >= n_verts) return;
    const int64_t lo = start[v], hi = start[v + 1];
    for (int64_t i = lo + 1; i < hi; ++i) {
        const int64_t f = vf_face[i];
        int64_t j = i;
        for (; j > lo && vf_face[j - 1] > f; --j) vf_face[j] = vf_face[j - 1];
        vf_face[j] = f;
    }
    if (!vn) return;
    d3 s{0, 0, 0};
    for (int64_t q = lo; q < hi; ++q) {
        const int64_t f = vf_face[q];
        s += tri_normal(ldv(V, F[3 * f]), ldv(V, F[3 * f + 1]), ldv(V, F[3 * f + 2]));
    }
    store_d3(vn, v, s);
}

// ---- collapse -----------------------------------------------------------------------------------------------------------------
// faces around `v` (other than the two that die with edge (a, b)) must stay valid when v moves to pnew; their centroids after the move
// go to the query list (drt_remesh.cpp::faces_stay_valid)
__device__ bool faces_stay_valid(int64_t v, d3 pnew, int64_t ea, int64_t eb, const int64_t* __restrict__ F, const double* __restrict__ V,
                                 const double* __restrict__ vn, const int64_t* __restrict__ vf_start, const int64_t* __restrict__ vf_face,
                                 double max_len, double* __restrict__ q, int& nq, int max_q) {
    for (int64_t s = vf_start[v]; s < vf_start[v + 1]; ++s) {
        const int64_t f = vf_face[s];
        const int64_t t[3] = {F[3 * f], F[3 * f + 1], F[3 * f + 2]};
        if ((t[0] == ea || t[1] == ea || t[2] == ea) && (t[0] == eb || t[1] == eb || t[2] == eb)) continue;     // dies
        d3 p[3] = {ldv(V, t[0]), ldv(V, t[1]), ldv(V, t[2])};
        const d3 n0 = tri_normal(p[0], p[1], p[2]);
        for (int k = 0; k < 3; ++k) if (t[k] == v) p[k] = pnew;
        const d3 n1 = tri_normal(p[0], p[1], p[2]);
        const double l0 = len3(n0), l1 = len3(n1);
        if (!(l1 > 1e-12 * (1.0 + l0))) return false;                                                            // degenerate
        if (!acceptable(agreement(n0, vn, t[0], t[1], t[2]), agreement(n1, vn, t[0], t[1], t[2]))) return false;   // would fold
        for (int k = 0; k < 3; ++k) if (t[k] != v && len3(p[k] - pnew) > max_len) return false;                  // would need a split again
        if (nq >= max_q) return false;
        const d3 c = ((p[0] + p[1]) + p[2]) * (1.0 / 3.0);
        q[3 * nq] = c.x; q[3 * nq + 1] = c.y; q[3 * nq + 2] = c.z; ++nq;
    }
    return true;
}

// one candidate edge (length < min_len): every check of drt_remesh.cpp::collapse_short_edges but the
// surface distance, whose query points (the midpoint, then the centroids of the surviving faces) it writes to q[c][max_q][3]
__device__ void collapse_eval_one(int64_t c, int64_t a, int64_t b, const int64_t* __restrict__ F,
                                  const double* __restrict__ V, const double* __restrict__ vn, const int64_t* __restrict__ vf_start,
                                  const int64_t* __restrict__ vf_face, double min_len, double max_len, int max_q,
                                  uint8_t* __restrict__ ok, int32_t* __restrict__ n_query, double* __restrict__ q) {
    const d3 pa = ldv(V, a), pb = ldv(V, b);
    if (!(len3(pa - pb) < min_len)) return;
    const int64_t va = vf_start[a + 1] - vf_start[a], vb = vf_start[b + 1] - vf_start[b];
    if (va > kMaxRing || vb > kMaxRing) return;
    int common = 0;
    int64_t opp[2] = {-1, -1};
    for (int64_t s = vf_start[a]; s < vf_start[a + 1]; ++s) {
        const int64_t u = next_in_face(F, vf_face[s], a);
        if (u != b && adjacent(u, b, F, vf_start, vf_face)) { if (common < 2) opp[common] = u; ++common; }
    }
    if (common != 2) return;                                                                        // link condition
    if (vf_start[opp[0] + 1] - vf_start[opp[0]] < 4 || vf_start[opp[1] + 1] - vf_start[opp[1]] < 4) return;   // no valence-3 vertices
    if (va + vb - 4 < 3) return;
    const d3 m = (pa + pb) * 0.5;
    double* qc = q + 3 * (int64_t)max_q * c;
    int nq = 0;
    qc[0] = m.x; qc[1] = m.y; qc[2] = m.z; nq = 1;
    if (!faces_stay_valid(a, m, a, b, F, V, vn, vf_start, vf_face, max_len, qc, nq, max_q)) return;
    if (!faces_stay_valid(b, m, a, b, F, V, vn, vf_start, vf_face, max_len, qc, nq, max_q)) return;
    n_query[c] = nq;
    ok[c] = 1;
}
// collapse_eval_one over EVERY directed-edge slot c = 3 f + k of the face array as it stands (round 6: no candidate list, so no stream compaction and
// no host round trip per round).  Every edge of a closed oriented mesh is the directed edge lo -> hi of exactly one face corner: slots with
// F[c] < F[next] are the unique edges, the others -- and the slots of faces an earlier round killed (indices -1) -- report ok = 0.  Also
// leaves the round's snapshot of the slot's edge (the claim / apply passes rewrite F) and its length (the priority key).
__global__ void k_rm_collapse_eval_all(const int64_t* __restrict__ F, int64_t n_faces, const double* __restrict__ V, const double* __restrict__ vn,
                                       const int64_t* __restrict__ vf_start, const int64_t* __restrict__ vf_face, double min_len, double max_len, int max_q,
                                       int64_t* __restrict__ E_snap, double* __restrict__ length, uint8_t* __restrict__ ok, int32_t* __restrict__ n_query,
                                       double* __restrict__ q, int32_t* __restrict__ ql_item, double* __restrict__ ql_point, unsigned* ql_count, unsigned ql_cap, const int32_t* __restrict__ live) {
    if (live && !*live) return;                   // (the step's rounds are over: drt_rm_round_end)
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= 3 * n_faces) return;
    const int64_t f = c / 3;
    const int k = (int)(c - 3 * f);
    const int64_t a = F[c], b = F[3 * f + (k + 1) % 3];
    ok[c] = 0; n_query[c] = 0;
    E_snap[2 * c] = a; E_snap[2 * c + 1] = b;
    length[c] = 0.0;
    if (a < 0 || !(a < b)) return;
    length[c] = len3(ldv(V, a) - ldv(V, b));
    collapse_eval_one(c, a, b, F, V, vn, vf_start, vf_face, min_len, max_len, max_q, ok, n_query, q);
    // the query points of a candidate that passed go to ONE compact list (drt_rm_surface_filter_list walks it with full wavefronts; a pass
    // over the 3 F x max_q slots with a few busy lanes per wave took 340 us per round, a third of the remesh call).  A full list leaves the
    // candidate to the next round.
    if (ql_item && ok[c]) {
        const unsigned nq = (unsigned)n_query[c], base = atomicAdd(ql_count, nq);
        if (base + nq > ql_cap) {              // does not fit: left to the next round; what it reserved below the cap becomes skipped entries
            ok[c] = 0;
            for (unsigned j = base; j < ql_cap; ++j) ql_item[j] = -1;
            return;
        }
        const double* qc = q + 3 * (int64_t)max_q * c;
        for (unsigned k = 0; k < nq; ++k) {
            ql_item[base + k] = (int32_t)c;
            ql_point[3 * (int64_t)(base + k)] = qc[3 * k]; ql_point[3 * (int64_t)(base + k) + 1] = qc[3 * k + 1]; ql_point[3 * (int64_t)(base + k) + 2] = qc[3 * k + 2];
        }
    }
}
// CheckSurfDist over the compact list of the round's query points (count on the device: no host round trip)
__global__ void __launch_bounds__(kTraceBlock) k_rm_surface_filter_list(TraceCtx c, const int32_t* __restrict__ faces, const float* __restrict__ verts,
                                                                         uint8_t* ok, const int32_t* __restrict__ item, const double* __restrict__ point,
                                                                         const unsigned* __restrict__ count, unsigned cap, double max_dist, const int32_t* __restrict__ live) {
    if (live && !*live) return;                   // (the step's rounds are over: drt_rm_round_end)
    __shared__ int32_t lds[kStackFast][kTraceBlock];
    Stack st = make_stack(lds, c);
    const unsigned n = min(*count, cap);
    for (unsigned i = blockIdx.x * kTraceBlock + threadIdx.x; i < n; i += gridDim.x * kTraceBlock) {
        const int32_t it = item[i];
        if (it < 0 || !ok[it]) continue;
        if (!within_distance(c.nodes, c.tris, c.n_tris, faces, verts, load_d3(point, i), max_dist, st)) ok[it] = 0;
    }
}
// CheckSurfDist for the candidates that passed: item i has n_query[i] (or one) query points in q[i][max_q][3]; a point farther than max_dist
// from the INPUT surface (closest point on the scene's tree, the query of drt_closest_point) takes the item's `ok` back.
__global__ void __launch_bounds__(kTraceBlock) k_rm_surface_filter(TraceCtx c, const int32_t* __restrict__ faces, const float* __restrict__ verts,
                                                                    uint8_t* ok, const int32_t* __restrict__ n_query, const double* __restrict__ q,
                                                                    int64_t n_items, int max_q, double max_dist, const int32_t* __restrict__ live) {
    if (live && !*live) return;                   // (the step's rounds are over: drt_rm_round_end)
    __shared__ int32_t lds[kStackFast][kTraceBlock];
    Stack st = make_stack(lds, c);
    const int64_t n = n_items * max_q;
    for (int64_t i = blockIdx.x * (int64_t)kTraceBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kTraceBlock) {
        const int64_t item = i / max_q;
        const int k = (int)(i - item * max_q);
        if (!ok[item] || k >= (n_query ? n_query[item] : 1)) continue;
        if (!within_distance(c.nodes, c.tris, c.n_tris, faces, verts, load_d3(q, i), max_dist, st)) ok[item] = 0;
    }
}
// the projection step: the closest point of the input surface to every vertex.  The vertices sit within a fraction of an edge length of
// it (they were on it before the relaxation), so the search starts bounded by `hint` and falls back to the unbounded query for the rare
// vertex farther out -- the result is the one drt_closest_point gives, at a fifth of its (latency-bound) time.
__global__ void __launch_bounds__(kTraceBlock) k_rm_closest_near(TraceCtx c, const int32_t* __restrict__ faces, const float* __restrict__ verts,
                                                                  const double* __restrict__ points, int64_t n, double hint2, double* __restrict__ closest) {
    __shared__ int32_t lds[kStackFast][kTraceBlock];
    Stack st = make_stack(lds, c);
    for (int64_t i = blockIdx.x * (int64_t)kTraceBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kTraceBlock) {
        const d3 p = load_d3(points, i);
        Closest r = closest_point(c.nodes, c.tris, c.n_tris, faces, verts, p, st, hint2);
        if (r.face < 0) r = closest_point(c.nodes, c.tris, c.n_tris, faces, verts, p, st);
        store_d3(closest, i, r.point);
    }
}
// the faces a round's collapses killed leave the face array in place: their indices become -1 (the next round's tables skip them; one
// compaction at the end of the step instead of one per round)
__global__ void k_rm_kill_faces(int64_t* __restrict__ F, uint8_t* __restrict__ f_alive, int64_t n_faces, const int32_t* __restrict__ live) {
    if (live && !*live) return;                   // (the step's rounds are over: drt_rm_round_end)
    const int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (f < n_faces && !f_alive[f]) { F[3 * f] = -1; F[3 * f + 1] = -1; F[3 * f + 2] = -1; f_alive[f] = 1; }      // (all ones again for the next round)
}

// Priority of a collapse: the shorter edges first (eight length classes below min_len, as the host version's sweep goes by length), a
// hash inside a class -- a strict (length, index) order leaves only the local minima of a smooth field to win a round, one candidate
// in a hundred -- and the edge index for uniqueness.
__device__ __forceinline__ unsigned long long collapse_key(double l, double min_len, int64_t e, unsigned seed) {
    int cls = (int)(8.0 * l / min_len);
    cls = cls < 0 ? 0 : (cls > 15 ? 15 : cls);
    unsigned h = (unsigned)e * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return ((unsigned long long)cls << 44) | ((unsigned long long)(h & 0xFFFu) << 32) | (unsigned long long)(uint32_t)e;
}
// A collapse WRITES a, b (position, faces) and the two vertices opposite the edge (their valence drops) and READS both rings.  It claims
// what it writes (64-bit atomicMin of its priority) and goes ahead when nobody with a higher priority has claimed anything it reads or
// writes: two collapses that go ahead in one round then share no face, and neither changes a valence or a position the other's checks
// relied on.
template <bool APPLY>
__global__ void k_rm_collapse_claim(const int64_t* __restrict__ cand, int64_t n_cand, const uint8_t* __restrict__ ok, const int64_t* __restrict__ E,
                                    int64_t* F, double* V, const int64_t* __restrict__ vf_start, const int64_t* __restrict__ vf_face, double min_len,
                                    unsigned seed, unsigned generation, uint8_t stamp, const double* __restrict__ length, unsigned long long* lock,
                                    uint8_t* __restrict__ f_alive, uint8_t* __restrict__ v_alive, uint8_t* dirty, int32_t* n_done, const int32_t* __restrict__ live) {
    if (live && !*live) return;                   // (the step's rounds are over: drt_rm_round_end)
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= n_cand || !ok[c]) return;
    const int64_t e = cand ? cand[c] : c, a = E[2 * e], b = E[2 * e + 1];          // (no list: every directed-edge slot is a candidate slot)
    // (from the lengths of the round's start: V changes under APPLY.)  The claim / apply pairs of a step are numbered, and the number sits
    // on top of the key, counting DOWN: a claim of an earlier pair is larger than any of this one, so it reads as "no claim" -- `lock` is
    // preset once per step, not once per pair.
    const unsigned long long key = ((unsigned long long)(0x7FFFu - generation) << 48) | collapse_key(length[e], min_len, e, seed);
    // Sub-rounds (several claim / apply pairs on ONE evaluation and ONE set of tables): a collapse that went ahead marks everything it
    // read or wrote `dirty` (= the round's stamp); a candidate takes part in a later sub-round only while its two vertices and both rings
    // are clean -- then no face, position, normal or valence its evaluation relied on has changed (a face around a clean vertex cannot
    // have been touched: the collapse that touched it would have had that vertex in a ring), so the verdict `ok` still stands.  `dirty` is
    // read in the claim pass and written in the apply pass only: a candidate that was not eligible has left no claim and cannot find its
    // key in `lock`.
    if (!APPLY && (dirty[a] == stamp || dirty[b] == stamp)) return;
    if (APPLY && !(lock[a] == key && lock[b] == key)) return;
    const int64_t a0 = vf_start[a], a1 = vf_start[a + 1], b0 = vf_start[b], b1 = vf_start[b + 1];
    if (!APPLY) {
        for (int64_t s = a0; s < a1; ++s) if (dirty[next_in_face(F, vf_face[s], a)] == stamp) return;
        for (int64_t s = b0; s < b1; ++s) if (dirty[next_in_face(F, vf_face[s], b)] == stamp) return;
        atomicMin(&lock[a], key); atomicMin(&lock[b], key);
        for (int64_t s = a0; s < a1; ++s) {                                          // the two opposite vertices: third corner of the faces on (a, b)
            const int64_t f = vf_face[s];
            const int64_t x = F[3 * f], y = F[3 * f + 1], z = F[3 * f + 2];
            if (x == b || y == b || z == b) atomicMin(&lock[x != a && x != b ? x : (y != a && y != b ? y : z)], key);
        }
        return;
    }
    bool mine = true;
    for (int64_t s = a0; s < a1 && mine; ++s) mine = lock[next_in_face(F, vf_face[s], a)] >= key;
    for (int64_t s = b0; s < b1 && mine; ++s) mine = lock[next_in_face(F, vf_face[s], b)] >= key;
    if (!mine) return;
    // commit: b -> a, a moves to the midpoint, the two shared faces die (every face touched has all its vertices under this claim)
    dirty[a] = stamp; dirty[b] = stamp;
    for (int64_t s = a0; s < a1; ++s) dirty[next_in_face(F, vf_face[s], a)] = stamp;   // (before the faces are rewritten)
    for (int64_t s = b0; s < b1; ++s) dirty[next_in_face(F, vf_face[s], b)] = stamp;
    const d3 m = (ldv(V, a) + ldv(V, b)) * 0.5;
    for (int64_t s = b0; s < b1; ++s) {
        const int64_t f = vf_face[s];
        const bool shared = F[3 * f] == a || F[3 * f + 1] == a || F[3 * f + 2] == a;
        if (shared) { f_alive[f] = 0; continue; }
        for (int k = 0; k < 3; ++k) if (F[3 * f + k] == b) F[3 * f + k] = a;
    }
    v_alive[b] = 0;
    store_d3(V, a, m);
    atomicAdd(n_done, 1);
}

// ---- flip ---------------------------------------------------------------------------------------------------------------------
// one thread per directed-edge slot c = 3 f + k of the face array (as the collapse: the slots with F[c] < F[next] are the unique edges of a
// closed oriented mesh, the others report ok = 0): the rules of drt_remesh.cpp::flip_edges (valence improvement across nearly flat pairs,
// or the repair of a folded pair).  The face across the edge and the "does edge (c, d) exist already" test both come from the vertex ->
// face lists -- a walk over one ring -- so a flip round needs no edge table (round 6: the sorted edge list and the edge -> rows table were
// 350 us of radix and merge sorts per round, most of a round).  Writes the four vertices and two faces of a flip that passes, and the
// midpoint of the new edge as its surface-distance query.
__global__ void k_rm_flip_eval(const int64_t* __restrict__ F, int64_t n_faces, const double* __restrict__ V, const double* __restrict__ vn,
                               const int64_t* __restrict__ vf_start, const int64_t* __restrict__ vf_face, double max_len,
                               uint8_t* __restrict__ ok, int64_t* __restrict__ quad /* [3F,6]: a b c d f1 f2 */, double* __restrict__ q, const int32_t* __restrict__ live) {
    if (live && !*live) return;                   // (the step's rounds are over: drt_rm_round_end)
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= 3 * n_faces) return;
    ok[e] = 0;
    const int64_t f1 = e / 3;
    const int k1 = (int)(e - 3 * f1);
    const int64_t a = F[e], b = F[3 * f1 + (k1 + 1) % 3], c = F[3 * f1 + (k1 + 2) % 3];
    if (a < 0 || !(a < b)) return;
    // the face that holds the edge the other way round (b -> a): one of b's
    int64_t f2 = -1, d = -1;
    for (int64_t s = vf_start[b]; s < vf_start[b + 1] && f2 < 0; ++s) {
        const int64_t f = vf_face[s];
        for (int k = 0; k < 3; ++k)
            if (F[3 * f + k] == b && F[3 * f + (k + 1) % 3] == a) { f2 = f; d = F[3 * f + (k + 2) % 3]; break; }
    }
    if (f2 < 0 || c < 0 || d < 0 || c == d || c == a || c == b || d == a || d == b) return;
    auto val = [&](int64_t v) { return (int)(vf_start[v + 1] - vf_start[v]); };
    const int va = val(a), vb = val(b), vc = val(c), vd = val(d);
    if (va < 4 || vb < 4) return;
    const d3 pa = ldv(V, a), pb = ldv(V, b), pc = ldv(V, c), pd = ldv(V, d);
    const d3 n1 = tri_normal(ldv(V, F[3 * f1]), ldv(V, F[3 * f1 + 1]), ldv(V, F[3 * f1 + 2]));
    const d3 n2 = tri_normal(ldv(V, F[3 * f2]), ldv(V, F[3 * f2 + 1]), ldv(V, F[3 * f2 + 2]));
    const d3 m1 = tri_normal(pc, pa, pd), m2 = tri_normal(pd, pb, pc);
    const double l1 = len3(n1), l2 = len3(n2), k1n = len3(m1), k2n = len3(m2);
    if (!(k1n > 1e-12 * (1.0 + l1)) || !(k2n > 1e-12 * (1.0 + l2))) return;
    const bool folded = dot(n1, n2) < -0.5 * l1 * l2;                      // the pair overlaps itself: repair, whatever the valences
    if (folded) {
        if (dot(m1, m2) < 0.5 * k1n * k2n) return;
        if (agreement(m1, vn, c, a, d) < 0.3 || agreement(m2, vn, d, b, c) < 0.3) return;
    } else {
        const int before = abs(va - 6) + abs(vb - 6) + abs(vc - 6) + abs(vd - 6);
        const int after = abs(va - 7) + abs(vb - 7) + abs(vc - 5) + abs(vd - 5);
        if (after >= before) return;
        if (dot(n1, n2) < 0.94 * l1 * l2) return;                          // only across nearly flat pairs (< 20 degrees)
        if (dot(m1, n1) < 0.5 * k1n * l1 || dot(m1, n2) < 0.5 * k1n * l2 || dot(m2, n1) < 0.5 * k2n * l1 || dot(m2, n2) < 0.5 * k2n * l2) return;
    }
    for (int64_t s = vf_start[c]; s < vf_start[c + 1]; ++s) {              // edge (c, d) exists already: a face of c holds d
        const int64_t f = vf_face[s];
        if (F[3 * f] == d || F[3 * f + 1] == d || F[3 * f + 2] == d) return;
    }
    if (len3(pc - pd) > max_len) return;
    const d3 mid = (pc + pd) * 0.5;
    q[3 * e] = mid.x; q[3 * e + 1] = mid.y; q[3 * e + 2] = mid.z;
    int64_t* o = quad + 6 * e;
    o[0] = a; o[1] = b; o[2] = c; o[3] = d; o[4] = f1; o[5] = f2;
    ok[e] = 1;
}
template <bool APPLY>
__global__ void k_rm_flip_claim(int64_t n_edges, const uint8_t* __restrict__ ok, const int64_t* __restrict__ quad, int64_t* F, unsigned generation, uint8_t stamp,
                                unsigned long long* lock, uint8_t* dirty, int32_t* n_done, const int32_t* __restrict__ live) {
    if (live && !*live) return;                   // (the step's rounds are over: drt_rm_round_end)
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= n_edges || !ok[e]) return;
    const int64_t* o = quad + 6 * e;
    // priority: a bijective hash of the slot (a strict index order would leave only the local minima of a smooth field to win a round),
    // under the number of the claim / apply pair counting down (k_rm_collapse_claim: one preset of `lock` per step)
    unsigned h = (unsigned)e * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned long long key = ((unsigned long long)(0x7FFFu - generation) << 48) | h;
    // (sub-rounds as in k_rm_collapse_claim: a flip that went ahead dirties its four vertices -- their valences and two of their faces
    // changed; a later sub-round admits only quads that are clean, whose evaluation therefore still stands)
    if (!APPLY) {
        for (int k = 0; k < 4; ++k) if (dirty[o[k]] == stamp) return;
        for (int k = 0; k < 4; ++k) atomicMin(&lock[o[k]], key);
        return;
    }
    for (int k = 0; k < 4; ++k) if (lock[o[k]] != key) return;
    const int64_t a = o[0], b = o[1], c = o[2], d = o[3], f1 = o[4], f2 = o[5];
    F[3 * f1] = c; F[3 * f1 + 1] = a; F[3 * f1 + 2] = d;
    F[3 * f2] = d; F[3 * f2 + 1] = b; F[3 * f2 + 2] = c;
    for (int k = 0; k < 4; ++k) dirty[o[k]] = stamp;
    atomicAdd(n_done, 1);
}

// the neighbours of one vertex as a list (the relaxation sums them in ascending order of their ids)
struct Ring {
    int n;
    bool overflow;
    int64_t v[kMaxRing];
    __device__ void add(int64_t u) {
        for (int k = 0; k < n; ++k) if (v[k] == u) return;
        if (n < kMaxRing) v[n++] = u; else overflow = true;
    }
};
__device__ __forceinline__ void collect_ring(int64_t v, const int64_t* __restrict__ F, const int64_t* __restrict__ vf_start,
                                             const int64_t* __restrict__ vf_face, Ring& r) {
    r.n = 0; r.overflow = false;
    for (int64_t q = vf_start[v]; q < vf_start[v + 1]; ++q) {
        const int64_t f = vf_face[q];
        for (int k = 0; k < 3; ++k) { const int64_t u = F[3 * f + k]; if (u != v) r.add(u); }
    }
}

// ---- relaxation / projection with roll-back ---------------------------------------------------------------------------------------
// tangential relaxation target: the ring centroid moved back along the vertex normal (drt_remesh.cpp::smooth_tangential)
__global__ void k_rm_smooth_target(const int64_t* __restrict__ F, const double* __restrict__ V, const int64_t* __restrict__ vf_start,
                                   const int64_t* __restrict__ vf_face, int64_t n_verts, double* __restrict__ target) {
    const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (v >= n_verts) return;
    const d3 pv = ldv(V, v);
    store_d3(target, v, pv);
    if (vf_start[v + 1] == vf_start[v]) return;
    d3 n{0, 0, 0};
    for (int64_t s = vf_start[v]; s < vf_start[v + 1]; ++s) {
        const int64_t f = vf_face[s];
        n += tri_normal(ldv(V, F[3 * f]), ldv(V, F[3 * f + 1]), ldv(V, F[3 * f + 2]));
    }
    const double ln = len3(n);
    if (!(ln > 0)) return;
    n = n * (1.0 / ln);
    Ring r;
    collect_ring(v, F, vf_start, vf_face, r);
    if (r.overflow || r.n == 0) return;
    for (int i = 1; i < r.n; ++i) {                                   // ascending ids: a summation order that does not depend on the adjacency order
        const int64_t x = r.v[i];
        int j = i - 1;
        while (j >= 0 && r.v[j] > x) { r.v[j + 1] = r.v[j]; --j; }
        r.v[j + 1] = x;
    }
    d3 g{0, 0, 0};
    for (int i = 0; i < r.n; ++i) g += ldv(V, r.v[i]);
    g = g * (1.0 / (double)r.n);
    store_d3(target, v, g + n * dot(n, pv - g));
}
// agreement of every face with the consensus before the move
__global__ void k_rm_face_agreement(const int64_t* __restrict__ F, const double* __restrict__ V, const double* __restrict__ vn, int64_t n_faces,
                                    double* __restrict__ a0) {
    const int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (f >= n_faces) return;
    const int64_t a = F[3 * f], b = F[3 * f + 1], c = F[3 * f + 2];
    a0[f] = agreement(tri_normal(ldv(V, a), ldv(V, b), ldv(V, c)), vn, a, b, c);
}
// a face the move degenerated or folded takes its three vertices back (flags; the driver applies them and calls again, four rounds at most)
__global__ void k_rm_move_check(const int64_t* __restrict__ F, const double* __restrict__ V, const double* __restrict__ vn, const double* __restrict__ a0,
                                int64_t n_faces, uint8_t* __restrict__ revert, int32_t* n_bad) {
    const int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (f >= n_faces) return;
    const int64_t a = F[3 * f], b = F[3 * f + 1], c = F[3 * f + 2];
    const d3 n1 = tri_normal(ldv(V, a), ldv(V, b), ldv(V, c));
    if (!(len3(n1) > 0) || !acceptable(a0[f], agreement(n1, vn, a, b, c))) {
        revert[a] = 1; revert[b] = 1; revert[c] = 1;
        atomicAdd(n_bad, 1);
    }
}
__global__ void k_rm_revert(double* __restrict__ V, const double* __restrict__ old, const uint8_t* __restrict__ revert, int64_t n_verts) {
    const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (v < n_verts && revert[v]) store_d3(V, v, ldv(old, v));
}

// The end of an evaluate / claim / apply round, decided ON THE DEVICE (round 6: the driver used to read the round's count back -- one host
// round trip per round, 76 us of an idle GPU each -- to decide whether to go on).  ctl int32 [8]: [0] live, [1] operations applied so far
// (what the apply kernels add to), [2] that count at the end of the previous round, [3] the first round's count, [4] rounds that ran.
// A step ends when a round applies nothing, or less than 1 / tail_cut of what its first round applied (a small mesh -- a first round of less
// than tail_cut operations -- goes on until a round applies nothing: its rounds cost next to nothing); every kernel
// of a later round of the batch the driver enqueued ahead returns at once.
__global__ void k_rm_round_end(int32_t* ctl, int tail_cut) {
    if (!ctl[0]) return;
    const int32_t n = ctl[1] - ctl[2];
    ctl[2] = ctl[1];
    if (ctl[4]++ == 0) ctl[3] = n;
    ctl[0] = n > 0 && n >= ctl[3] / tail_cut ? 1 : 0;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1); }

}  // namespace

extern "C" {

int drt_rm_split_mark(const int64_t* d_faces, int64_t n_faces, const double* d_verts, double max_len, uint8_t* d_flag, void* stream) {
    if (n_faces < 0) return fail(DRT_E_INVALID, "negative face count");
    if (n_faces == 0) return DRT_OK;
    if (!d_faces || !d_verts || !d_flag) return fail(DRT_E_INVALID, "null pointer argument");
    k_rm_split_mark<<<blocks_for(3 * n_faces), 256, 0, (hipStream_t)stream>>>(d_faces, n_faces, d_verts, max_len, d_flag);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_split_plan(const int64_t* d_faces, int64_t n_faces, const int64_t* d_vf_start, const int64_t* d_vf_face, const uint8_t* d_flag,
                      const int64_t* d_rank, int64_t n_verts, int64_t* d_mid_of_slot, int64_t* d_count, void* stream) {
    if (n_faces < 0) return fail(DRT_E_INVALID, "negative face count");
    if (n_faces == 0) return DRT_OK;
    if (!d_faces || !d_vf_start || !d_vf_face || !d_flag || !d_rank || !d_mid_of_slot || !d_count) return fail(DRT_E_INVALID, "null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    k_rm_split_assign<<<blocks_for(3 * n_faces), 256, 0, st>>>(d_faces, n_faces, d_vf_start, d_vf_face, d_flag, d_rank, n_verts, d_mid_of_slot);
    k_rm_split_count<<<blocks_for(n_faces), 256, 0, st>>>(d_mid_of_slot, n_faces, d_count);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_split_faces(const int64_t* d_faces, int64_t n_faces, const int64_t* d_mid_of_slot, double* d_verts, const int64_t* d_offset,
                       int64_t* d_faces_out, void* stream) {
    if (n_faces < 0) return fail(DRT_E_INVALID, "negative face count");
    if (n_faces == 0) return DRT_OK;
    if (!d_faces || !d_mid_of_slot || !d_verts || !d_offset || !d_faces_out) return fail(DRT_E_INVALID, "null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    k_rm_split_midpoints<<<blocks_for(3 * n_faces), 256, 0, st>>>(d_faces, n_faces, d_mid_of_slot, d_verts);
    k_rm_split_faces<<<blocks_for(n_faces), 256, 0, st>>>(d_faces, d_mid_of_slot, d_verts, n_faces, d_offset, d_faces_out);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_vertex_faces(const int64_t* d_faces, int64_t n_faces, int64_t n_verts, int32_t* d_count, int64_t* d_vf_start, int64_t* d_vf_face,
                        const double* d_verts, double* d_vn, const int32_t* d_live, void* stream) {
    if (n_faces < 0 || n_verts <= 0) return fail(DRT_E_INVALID, "bad mesh size");
    if (!d_count || !d_vf_start || !d_vf_face || (n_faces && !d_faces) || (d_vn && !d_verts)) return fail(DRT_E_INVALID, "null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(int32_t) * (size_t)n_verts, st));
    if (n_faces) k_rm_vf_count<<<blocks_for(3 * n_faces), 256, 0, st>>>(d_faces, 3 * n_faces, d_count, d_live);
    k_rm_vf_scan<<<1, 1024, 0, st>>>(d_count, n_verts, d_vf_start, d_live);
    if (n_faces) k_rm_vf_fill<<<blocks_for(3 * n_faces), 256, 0, st>>>(d_faces, 3 * n_faces, d_vf_start, d_count, d_vf_face, d_live);
    k_rm_vf_sort<<<blocks_for(n_verts), 256, 0, st>>>(d_faces, d_vf_start, d_vf_face, n_verts, d_verts, d_vn, d_live);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_vertex_normals(const int64_t* d_faces, const double* d_verts, const int64_t* d_vf_start, const int64_t* d_vf_face, int64_t n_verts,
                          double* d_vn, void* stream) {
    if (n_verts <= 0) return DRT_OK;
    if (!d_faces || !d_verts || !d_vf_start || !d_vf_face || !d_vn) return fail(DRT_E_INVALID, "null pointer argument");
    k_rm_vertex_normals<<<blocks_for(n_verts), 256, 0, (hipStream_t)stream>>>(d_faces, d_verts, d_vf_start, d_vf_face, n_verts, d_vn);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_collapse_eval_all(const int64_t* d_faces, int64_t n_faces, const double* d_verts, const double* d_vn, const int64_t* d_vf_start,
                             const int64_t* d_vf_face, double min_len, double max_len, int max_q, int64_t* d_edge_snap, double* d_length,
                             uint8_t* d_ok, int32_t* d_n_query, double* d_query, int32_t* d_list_item, double* d_list_point, uint32_t* d_list_count,
                             int64_t list_cap, const int32_t* d_live, void* stream) {
    if (n_faces <= 0) return DRT_OK;
    if (!d_faces || !d_verts || !d_vn || !d_vf_start || !d_vf_face || !d_edge_snap || !d_length || !d_ok || !d_n_query || !d_query || max_q < 1)
        return fail(DRT_E_INVALID, "bad argument");
    if (d_list_item && (!d_list_point || !d_list_count || list_cap <= 0 || list_cap > UINT32_MAX / 2)) return fail(DRT_E_INVALID, "bad query list");
    hipStream_t st = (hipStream_t)stream;
    if (d_list_item) HIP_TRY(hipMemsetAsync(d_list_count, 0, sizeof(uint32_t), st));
    k_rm_collapse_eval_all<<<blocks_for(3 * n_faces), 256, 0, st>>>(d_faces, n_faces, d_verts, d_vn, d_vf_start, d_vf_face, min_len, max_len, max_q,
                                                                   d_edge_snap, d_length, d_ok, d_n_query, d_query, d_list_item, d_list_point, d_list_count,
                                                                   (unsigned)(d_list_item ? list_cap : 0), d_live);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_surface_filter_list(drt_scene_t* s, uint8_t* d_ok, const int32_t* d_list_item, const double* d_list_point, const uint32_t* d_list_count,
                               int64_t list_cap, double max_dist, const int32_t* d_live, void* stream) {
    CHECK_BUILT(s);
    if (!d_ok || !d_list_item || !d_list_point || !d_list_count || list_cap <= 0 || list_cap > UINT32_MAX / 2) return fail(DRT_E_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    { int rc = wait_build(s, st); if (rc) return rc; }
    { int rc = ensure_slow_stack(s, st); if (rc) return rc; }
    k_rm_surface_filter_list<<<grid_for(list_cap, kTraceBlock, s->grid_trace), kTraceBlock, 0, st>>>(trace_ctx(s), s->faces, s->verts, d_ok, d_list_item, d_list_point,
                                                                                                      d_list_count, (unsigned)list_cap, max_dist, d_live);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_surface_filter(drt_scene_t* s, uint8_t* d_ok, const int32_t* d_n_query, const double* d_query, int64_t n_items, int max_q, double max_dist,
                          const int32_t* d_live, void* stream) {
    CHECK_BUILT(s);
    if (n_items <= 0) return DRT_OK;
    if (!d_ok || !d_query || max_q < 1) return fail(DRT_E_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    { int rc = wait_build(s, st); if (rc) return rc; }
    { int rc = ensure_slow_stack(s, st); if (rc) return rc; }
    k_rm_surface_filter<<<grid_for(n_items * max_q, kTraceBlock, s->grid_trace), kTraceBlock, 0, st>>>(trace_ctx(s), s->faces, s->verts, d_ok, d_n_query, d_query,
                                                                                                        n_items, max_q, max_dist, d_live);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_closest_near(drt_scene_t* s, const double* d_points, int64_t n, double hint_radius, double* d_closest, void* stream) {
    CHECK_BUILT(s);
    if (n < 0) return fail(DRT_E_INVALID, "negative point count");
    if (n == 0) return DRT_OK;
    if (!d_points || !d_closest) return fail(DRT_E_INVALID, "null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    { int rc = wait_build(s, st); if (rc) return rc; }
    { int rc = ensure_slow_stack(s, st); if (rc) return rc; }
    const double hint2 = hint_radius > 0.0 ? hint_radius * hint_radius : INFINITY;          // (NaN, zero, negative: no hint)
    k_rm_closest_near<<<grid_for(n, kTraceBlock, s->grid_trace), kTraceBlock, 0, st>>>(trace_ctx(s), s->faces, s->verts, d_points, n, hint2, d_closest);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_kill_faces(int64_t* d_faces, uint8_t* d_f_alive, int64_t n_faces, const int32_t* d_live, void* stream) {
    if (n_faces <= 0) return DRT_OK;
    if (!d_faces || !d_f_alive) return fail(DRT_E_INVALID, "null pointer argument");
    k_rm_kill_faces<<<blocks_for(n_faces), 256, 0, (hipStream_t)stream>>>(d_faces, d_f_alive, n_faces, d_live);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_collapse_apply(const int64_t* d_cand, int64_t n_cand, const uint8_t* d_ok, const int64_t* d_edges, int64_t* d_faces, double* d_verts,
                          const int64_t* d_vf_start, const int64_t* d_vf_face, int64_t n_verts, double min_len, uint32_t seed, int round, const double* d_length,
                          uint64_t* d_lock, uint8_t* d_f_alive, uint8_t* d_v_alive, uint8_t* d_dirty, int sub_rounds, int32_t* d_n_done, const int32_t* d_live, void* stream) {
    if (n_cand <= 0) return DRT_OK;
    if (!d_ok || !d_edges || !d_faces || !d_verts || !d_vf_start || !d_vf_face || !d_length || !d_lock || !d_f_alive || !d_v_alive || !d_dirty ||
        !d_n_done || n_verts <= 0 || sub_rounds < 1 || round < 0 || round > 254 || (int64_t)(round + 1) * sub_rounds > 0x7FFF)
        return fail(DRT_E_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* lock = reinterpret_cast<unsigned long long*>(d_lock);
    if (round == 0) {                       // the step's presets: no claim (all ones), nothing dirty, every face alive
        HIP_TRY(hipMemsetAsync(lock, 0xFF, sizeof(unsigned long long) * (size_t)n_verts, st));
        HIP_TRY(hipMemsetAsync(d_dirty, 0, (size_t)n_verts, st));
    }
    const uint8_t stamp = (uint8_t)(round + 1);
    for (int r = 0; r < sub_rounds; ++r) {
        const uint32_t sd = seed + 0x632BE5ABu * (uint32_t)r;
        const unsigned gen = (unsigned)(round * sub_rounds + r);
        k_rm_collapse_claim<false><<<blocks_for(n_cand), 256, 0, st>>>(d_cand, n_cand, d_ok, d_edges, d_faces, d_verts, d_vf_start, d_vf_face, min_len, sd, gen, stamp, d_length, lock, d_f_alive, d_v_alive, d_dirty, d_n_done, d_live);
        k_rm_collapse_claim<true><<<blocks_for(n_cand), 256, 0, st>>>(d_cand, n_cand, d_ok, d_edges, d_faces, d_verts, d_vf_start, d_vf_face, min_len, sd, gen, stamp, d_length, lock, d_f_alive, d_v_alive, d_dirty, d_n_done, d_live);
    }
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_round_end(int32_t* d_ctl, int tail_cut, void* stream) {
    if (!d_ctl || tail_cut < 1) return fail(DRT_E_INVALID, "bad argument");
    k_rm_round_end<<<1, 1, 0, (hipStream_t)stream>>>(d_ctl, tail_cut);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_flip_eval(const int64_t* d_faces, int64_t n_faces, const double* d_verts, const double* d_vn, const int64_t* d_vf_start,
                     const int64_t* d_vf_face, double max_len, uint8_t* d_ok, int64_t* d_quad, double* d_query, const int32_t* d_live, void* stream) {
    if (n_faces <= 0) return DRT_OK;
    if (!d_faces || !d_verts || !d_vn || !d_vf_start || !d_vf_face || !d_ok || !d_quad || !d_query) return fail(DRT_E_INVALID, "null pointer argument");
    k_rm_flip_eval<<<blocks_for(3 * n_faces), 256, 0, (hipStream_t)stream>>>(d_faces, n_faces, d_verts, d_vn, d_vf_start, d_vf_face, max_len, d_ok, d_quad, d_query, d_live);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_flip_apply(int64_t n_items, const uint8_t* d_ok, const int64_t* d_quad, int64_t* d_faces, int64_t n_verts, int round, uint64_t* d_lock,
                      uint8_t* d_dirty, int sub_rounds, int32_t* d_n_done, const int32_t* d_live, void* stream) {
    if (n_items <= 0) return DRT_OK;
    if (!d_ok || !d_quad || !d_faces || !d_lock || !d_dirty || !d_n_done || n_verts <= 0 || sub_rounds < 1 || round < 0 || round > 254 ||
        (int64_t)(round + 1) * sub_rounds > 0x7FFF)
        return fail(DRT_E_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* lock = reinterpret_cast<unsigned long long*>(d_lock);
    if (round == 0) {
        HIP_TRY(hipMemsetAsync(lock, 0xFF, sizeof(unsigned long long) * (size_t)n_verts, st));
        HIP_TRY(hipMemsetAsync(d_dirty, 0, (size_t)n_verts, st));
    }
    const uint8_t stamp = (uint8_t)(round + 1);
    for (int r = 0; r < sub_rounds; ++r) {
        const unsigned gen = (unsigned)(round * sub_rounds + r);
        k_rm_flip_claim<false><<<blocks_for(n_items), 256, 0, st>>>(n_items, d_ok, d_quad, d_faces, gen, stamp, lock, d_dirty, d_n_done, d_live);
        k_rm_flip_claim<true><<<blocks_for(n_items), 256, 0, st>>>(n_items, d_ok, d_quad, d_faces, gen, stamp, lock, d_dirty, d_n_done, d_live);
    }
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_smooth_target(const int64_t* d_faces, const double* d_verts, const int64_t* d_vf_start, const int64_t* d_vf_face, int64_t n_verts,
                         double* d_target, void* stream) {
    if (n_verts <= 0) return DRT_OK;
    if (!d_faces || !d_verts || !d_vf_start || !d_vf_face || !d_target) return fail(DRT_E_INVALID, "null pointer argument");
    k_rm_smooth_target<<<blocks_for(n_verts), 256, 0, (hipStream_t)stream>>>(d_faces, d_verts, d_vf_start, d_vf_face, n_verts, d_target);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_face_agreement(const int64_t* d_faces, const double* d_verts, const double* d_vn, int64_t n_faces, double* d_a0, void* stream) {
    if (n_faces <= 0) return DRT_OK;
    if (!d_faces || !d_verts || !d_vn || !d_a0) return fail(DRT_E_INVALID, "null pointer argument");
    k_rm_face_agreement<<<blocks_for(n_faces), 256, 0, (hipStream_t)stream>>>(d_faces, d_verts, d_vn, n_faces, d_a0);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_rm_move_check(const int64_t* d_faces, double* d_verts, const double* d_old, const double* d_vn, const double* d_a0, int64_t n_faces,
                      int64_t n_verts, uint8_t* d_revert, int32_t* d_n_bad, void* stream) {
    if (n_faces <= 0) return DRT_OK;
    if (!d_faces || !d_verts || !d_old || !d_vn || !d_a0 || !d_revert || !d_n_bad) return fail(DRT_E_INVALID, "null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(d_revert, 0, (size_t)n_verts, st));
    HIP_TRY(hipMemsetAsync(d_n_bad, 0, sizeof(int32_t), st));
    k_rm_move_check<<<blocks_for(n_faces), 256, 0, st>>>(d_faces, d_verts, d_vn, d_a0, n_faces, d_revert, d_n_bad);
    k_rm_revert<<<blocks_for(n_verts), 256, 0, st>>>(d_verts, d_old, d_revert, n_verts);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

}  // extern "C"
