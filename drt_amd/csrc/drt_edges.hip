// drt_edges.hip -- silhouette (visual-hull) and smoothness branches, per unique edge.
#include "drt_device.h"

// ---- silhouette and smoothness branches (per unique edge) -------------------------------------
__device__ __forceinline__ void load_face64(const double* __restrict__ verts, const int64_t* __restrict__ f, d3& v0, d3& v1, d3& v2) {
    v0 = load_d3(verts, f[0]); v1 = load_d3(verts, f[1]); v2 = load_d3(verts, f[2]);
}

// cos of the dihedral angle of every edge (reference DiffRender.py:440-443)
__global__ void __launch_bounds__(256) k_dihedral_fwd(const double* __restrict__ verts, const int64_t* __restrict__ e2f, int64_t n,
                                                      double* __restrict__ cos_out) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= n) return;
    d3 v0, v1, v2;
    FaceNormal a, b;
    load_face64(verts, e2f + 6 * e, v0, v1, v2); face_normal(v0, v1, v2, a);
    load_face64(verts, e2f + 6 * e + 3, v0, v1, v2); face_normal(v0, v1, v2, b);
    cos_out[e] = dot(a.n, b.n);
}

// MODE 0: adjoint of k_dihedral_fwd for a given d loss / d cos.
// MODE 1: sm_loss = sum -log(1 + cos) (reference optim.py:82-89) and its vertex gradient in one pass.
template <bool DET, int MODE>
__global__ void __launch_bounds__(256) k_dihedral_bwd(const double* __restrict__ verts, const int64_t* __restrict__ e2f, int64_t n,
                                                      const double* __restrict__ g_cos, double* loss, double* grad_verts) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    LossAcc<DET> term;
    if (e < n) {
        d3 v0, v1, v2;
        FaceNormal a, b;
        const int64_t* fa = e2f + 6 * e;
        const int64_t* fb = fa + 3;
        load_face64(verts, fa, v0, v1, v2); face_normal(v0, v1, v2, a);
        load_face64(verts, fb, v0, v1, v2); face_normal(v0, v1, v2, b);
        double g;
        if (MODE == 0) {
            g = g_cos[e];
        } else {
            const double c = dot(a.n, b.n);
            term.add(-log(1.0 + c));
            g = -1.0 / (1.0 + c);
        }
        const d3 z{0.0, 0.0, 0.0};
        d3 g0 = z, g1 = z, g2 = z;
        const GradAdd3<DET> add{grad_verts};
        face_normal_backward(a, g * b.n, g0, g1, g2);
        add((int32_t)fa[0], g0); add((int32_t)fa[1], g1); add((int32_t)fa[2], g2);
        g0 = z; g1 = z; g2 = z;
        face_normal_backward(b, g * a.n, g0, g1, g2);
        add((int32_t)fb[0], g0); add((int32_t)fb[1], g1); add((int32_t)fb[2], g2);
    }
    if (MODE == 1) term.flush(loss);
}

// silhouette test per unique edge (reference DiffRender.py:445-457)
__global__ void __launch_bounds__(256) k_silhouette_flags(const double* __restrict__ verts, const int64_t* __restrict__ e2f, int64_t n,
                                                          const double* __restrict__ origin3, uint8_t* __restrict__ flags) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const d3 o{origin3[0], origin3[1], origin3[2]};
    d3 a0, b0, v1, v2;
    FaceNormal a, b;
    load_face64(verts, e2f + 6 * e, a0, v1, v2); face_normal(a0, v1, v2, a);
    load_face64(verts, e2f + 6 * e + 3, b0, v1, v2); face_normal(b0, v1, v2, b);
    flags[e] = silhouette_flag(a, a0, b, b0, o) ? 1 : 0;
}

// primary_visibility + primary_edge_sample.forward for every silhouette edge: project the two
// endpoints, probe one pixel either side of the edge midpoint with any-hit rays, f = hit+ - hit-.
__global__ void __launch_bounds__(kTraceBlock) k_edge_sample_fwd(TraceCtx c, const double* __restrict__ verts, const int64_t* __restrict__ edges,
                                                                  int64_t n, const Camera* __restrict__ cam, const double* __restrict__ origin3,
                                                                  int64_t* __restrict__ index, float* __restrict__ f_out,
                                                                  uint8_t* __restrict__ keep, int resx, int resy, const uint8_t* __restrict__ flags) {
    __shared__ int32_t lds[kStackFast][kTraceBlock];
    Stack st = make_stack(lds, c);
    const Camera cm = *cam;
    const d3 o{origin3[0], origin3[1], origin3[2]};
    // two lanes per edge, one probe ray each (see k_vh_fused: a launch over a few thousand edges lasts as long as its longest lane)
    constexpr int64_t kPairs = kTraceBlock / 2;
    const int side = threadIdx.x & 1;
    for (int64_t base = blockIdx.x * kPairs; base < n; base += (int64_t)gridDim.x * kPairs) {     // block-uniform trip count (shuffle below)
        const int64_t e = base + (threadIdx.x >> 1);
        const bool live = e < n && (!flags || flags[e]);       // (`flags`: the edge list is ALL edges and only the flagged ones are silhouette edges)
        const int64_t el = live ? e : 0;
        Projected pa, pb;
        project_endpoint(cm, load_d3(verts, edges[2 * el]), pa);
        project_endpoint(cm, load_d3(verts, edges[2 * el + 1]), pb);
        EdgeSample s;
        edge_sample(cm, pa, pb, o, s);
        const int mine = live && traverse<true>(c.nodes, c.tris, c.n_tris, to_f32(o), to_f32(side == 0 ? s.dir_up : s.dir_lo), st).face >= 0 ? 1 : 0;
        const int other = __shfl_xor(mine, 1);
        if (side != 0 || e >= n) continue;
        if (!live) {                                  // an unflagged edge of the all-edges form: a dropped row (the caller need not zero anything)
            f_out[e] = 0.0f;
            if (keep) keep[e] = 0;
            continue;
        }
        f_out[e] = (float)mine - (float)other;        // hit(up) - hit(lo)
        const int64_t x = (int64_t)s.midx, y = (int64_t)s.midy;       // truncation toward zero, like Tensor.to(torch.long)
        index[2 * e] = x;
        index[2 * e + 1] = y;
        // |f| > 1e-5 (DiffRender.py:244) and inside the view (DiffRender.py:478): what the caller keeps
        if (keep) keep[e] = (mine != other && x < resx - 1 && y < resy - 1 && x >= 0 && y >= 0) ? 1 : 0;
    }
}

// Adjoint: dE_pos[e, endpoint, :] = -N_e * f_e * coef_e for both endpoints (reference
// DiffRender.py:236-242, 263-267), chained through the projection to the two vertices.
template <bool DET>
__global__ void __launch_bounds__(256) k_edge_sample_bwd(const double* __restrict__ verts, const int64_t* __restrict__ edges, int64_t n,
                                                         const Camera* __restrict__ cam, const float* __restrict__ f,
                                                         const double* __restrict__ coef, int detach_depth, double* grad_verts,
                                                         const double* __restrict__ g32 /* optional device scalar: coef is scaled by (float)*g32 */) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const double w = (double)f[e] * (g32 ? coef[e] * (double)(float)*g32 : coef[e]);
    if (w == 0.0) return;
    const Camera cm = *cam;
    const int64_t ia = edges[2 * e], ib = edges[2 * e + 1];
    Projected pa, pb;
    project_endpoint(cm, load_d3(verts, ia), pa);
    project_endpoint(cm, load_d3(verts, ib), pb);
    const double gx = -(pa.py - pb.py) * w, gy = -(pb.px - pa.px) * w;
    const GradAdd3<DET> add{grad_verts};
    add((int32_t)ia, project_endpoint_backward(cm, pa, gx, gy, detach_depth != 0));
    add((int32_t)ib, project_endpoint_backward(cm, pb, gx, gy, detach_depth != 0));
}

// The loss expression of ONE view over the samples drt_edge_sample_forward left in place (all E rows, `keep` marks the samples):
// sum over kept rows of |soft[y, x] - 0.5| (reference optim.py:78 with `output` = 0.5) and, per row, d term / d output = -sign(soft - 0.5)
// (0 for dropped rows) -- what the drop-in pair's LAZY return values evaluate when the caller writes the reference's expression
// (drt_amd/diffrender.py SampleSet): no compaction, no host round trip.
template <bool DET>
__global__ void __launch_bounds__(256) k_vh_term(const int64_t* __restrict__ index, const uint8_t* __restrict__ keep, int64_t n,
                                                 const double* __restrict__ soft, int resx, double* loss, double* __restrict__ dterm) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    LossAcc<DET> t;
    if (e < n) {
        double dsign = 0.0;
        if (keep[e]) {
            const double m = soft[index[2 * e + 1] * resx + index[2 * e]] - 0.5;
            t.add(fabs(m));
            dsign = m > 0.0 ? -1.0 : (m < 0.0 ? 1.0 : 0.0);
        }
        dterm[e] = dsign;
    }
    t.flush(loss);
}

// ---- fused silhouette loss: Loss_calculator.vh_loss (reference optim.py:73-78) with no host round trip:
// the drop-in methods return dynamically sized tensors (two device->host syncs per view); here the
// silhouette edges of up to kVhViews views are compacted on the device into ONE list and one kernel does
// projection, probe rays, the loss term |soft_mask[y, x] - 0.5| and its vertex gradient.  Views are
// batched because a view has only a few thousand silhouette edges and its probe rays graze the surface:
// a per-view launch is bound by the latency of its longest traversal, not by throughput.
constexpr int kVhViews = 16;
struct VhViews {
    const double* cam[kVhViews];
    const double* origin[kVhViews];
    const double* soft[kVhViews];
};

__global__ void __launch_bounds__(kPathBlock) k_vh_cull(const double* __restrict__ verts, const int64_t* __restrict__ e2f, int64_t n_edges,
                                                         int n_views, VhViews vw, uint32_t* __restrict__ list, unsigned* count) {
    __shared__ unsigned s_tmp[kPathWaves + 1];
    const int64_t n = n_edges * n_views;
    for (int64_t base = blockIdx.x * (int64_t)kPathBlock; base < n; base += (int64_t)gridDim.x * kPathBlock) {
        const int64_t k = base + threadIdx.x;
        bool sil = false;
        if (k < n) {
            const int64_t e = k % n_edges;
            const double* o3 = vw.origin[k / n_edges];
            d3 a0, b0, v1, v2;
            FaceNormal a, b;
            load_face64(verts, e2f + 6 * e, a0, v1, v2); face_normal(a0, v1, v2, a);
            load_face64(verts, e2f + 6 * e + 3, b0, v1, v2); face_normal(b0, v1, v2, b);
            sil = silhouette_flag(a, a0, b, b0, d3{o3[0], o3[1], o3[2]});
        }
        const int slot = block_push(sil, count, s_tmp);
        if (slot >= 0) list[slot] = (uint32_t)k;
    }
}

template <bool DET>
__global__ void __launch_bounds__(kTraceBlock) k_vh_fused(TraceCtx c, const double* __restrict__ verts, const int64_t* __restrict__ edges,
                                                           uint32_t n_edges, const uint32_t* __restrict__ list, const unsigned* __restrict__ count,
                                                           VhViews vw, int resx, int resy, int detach_depth, double* loss, double* grad_verts) {
    __shared__ int32_t lds[kStackFast][kTraceBlock];
    Stack st = make_stack(lds, c);
    const unsigned n = *count;
    LossAcc<DET> acc;
    // Two lanes per edge, one probe ray each: the rays graze the silhouette and take a few hundred node visits, and with
    // only a few thousand edges per view the kernel lasts as long as its longest lane -- tracing the two probes of an
    // edge one after the other in one lane doubled that.
    constexpr unsigned kPairs = kTraceBlock / 2;
    const int side = threadIdx.x & 1;
    for (unsigned base = blockIdx.x * kPairs; base < n; base += gridDim.x * kPairs) {       // block-uniform trip count (shuffles below)
        const unsigned k = base + (threadIdx.x >> 1);
        const bool live = k < n;
        const uint32_t item = live ? list[k] : 0u, view = item / n_edges;
        const int64_t e = item - view * n_edges;
        const Camera cm = *reinterpret_cast<const Camera*>(vw.cam[view]);
        const double* o3 = vw.origin[view];
        const d3 o{o3[0], o3[1], o3[2]};
        const int64_t ia = edges[2 * e], ib = edges[2 * e + 1];
        Projected pa, pb;
        project_endpoint(cm, load_d3(verts, ia), pa);
        project_endpoint(cm, load_d3(verts, ib), pb);
        EdgeSample s;
        edge_sample(cm, pa, pb, o, s);
        const int mine = live && traverse<true>(c.nodes, c.tris, c.n_tris, to_f32(o), to_f32(side == 0 ? s.dir_up : s.dir_lo), st).face >= 0 ? 1 : 0;
        const int other = __shfl_xor(mine, 1);
        if (!live || side != 0) continue;                         // the even lane of the pair finishes the edge
        const double f = (double)mine - (double)other;           // hit(up) - hit(lo)
        if (f == 0.0) continue;                                   // |f| > 1e-5 (DiffRender.py:244)
        const int64_t x = (int64_t)s.midx, y = (int64_t)s.midy;   // trunc, like Tensor.to(torch.long)
        if (!(x < resx - 1 && y < resy - 1 && x >= 0 && y >= 0)) continue;   // out of view (DiffRender.py:478)
        const double m = vw.soft[view][y * resx + x] - 0.5;       // output is float32 0.5: exact
        acc.add(fabs(m));
        const double coef = m > 0.0 ? -1.0 : (m < 0.0 ? 1.0 : 0.0);   // d |mask - output| / d output
        const double w = f * coef;
        if (w == 0.0) continue;
        const double gx = -(pa.py - pb.py) * w, gy = -(pb.px - pa.px) * w;
        const GradAdd3<DET> add{grad_verts};
        add((int32_t)ia, project_endpoint_backward(cm, pa, gx, gy, detach_depth != 0));
        add((int32_t)ib, project_endpoint_backward(cm, pb, gx, gy, detach_depth != 0));
    }
    acc.flush(loss);
}

// limit_hook + one SGD(nesterov) step (reference optim.py:155-171, 215) in one pass over the [V,3] parameter: the gradient is
// sanitised in place (NaN -> 0, clamp to +-max_abs: what the reference's hook does to parameter.grad), the momentum buffer and
// the parameter follow torch.optim.SGD (dampening 0, no weight decay): buf = g on the first step, momentum * buf + g after;
// step = g + momentum * buf with nesterov, buf without; param -= lr * step.  Replaces ~8 elementwise launches per iteration.
// `terms` / `w` / `loss_*` (optional): grad[i] is first formed as w[0] * terms[i] + w[1] * terms[n + i] + w[2] * terms[2 n + i] -- the weighted
// sum of the three loss terms' vertex gradients of reference optim.py:127-129, in that order of operations -- and thread 0 leaves the weighted
// total of the three losses in *loss_total: the all_loss arithmetic of an iteration without its five scalar kernels.
// The weight of the SECOND term's gradient is rounded through float32: the reference's silhouette samples are a float32 tensor
// (`output`, created with torch's default dtype, DiffRender.py:251), so autograd casts d loss / d output = -+w_vh to float32 before
// primary_edge_sample.backward multiplies it in (DiffRender.py:263-267) -- the gradient of that term carries float32(w_vh), the loss itself
// the float64 weight.  Found by the 60-iteration replay of the reference's loop (tests/golden/hand_trajectory.npz): with the float64 weight
// the parameters left the reference's by 3e-11 mm per iteration, with this they stay within 1e-14.
__global__ void __launch_bounds__(256) k_limit_sgd(double* __restrict__ param, double* __restrict__ grad, double* __restrict__ buf, int64_t n,
                                                   double lr, double momentum, int nesterov, int first, double max_abs,
                                                   const double* __restrict__ terms, const double* __restrict__ w, const double* __restrict__ loss_parts,
                                                   double* __restrict__ loss_total) {
    if (loss_total && blockIdx.x == 0 && threadIdx.x == 0) *loss_total = (w[0] * loss_parts[0] + w[1] * loss_parts[1]) + w[2] * loss_parts[2];
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double g = terms ? (w[0] * terms[i] + (double)(float)w[1] * terms[n + i]) + w[2] * terms[2 * n + i] : grad[i];
        if (max_abs > 0.0) {
            g = g != g ? 0.0 : g;
            g = g > max_abs ? max_abs : (g < -max_abs ? -max_abs : g);
            grad[i] = g;
        } else if (terms) {
            grad[i] = g;
        }
        double step = g;
        if (momentum != 0.0) {
            const double b = first ? g : buf[i] * momentum + g;
            buf[i] = b;
            step = nesterov ? g + momentum * b : b;
        }
        param[i] = param[i] + (-lr) * step;
    }
}

// The same for the KEPT rows only: rows[k] = edge row of the k-th kept sample, g[k] = d loss / d output of that sample in the float32 of
// `output` -- what the caller has after its one boolean index; spares it a zero-filled float64 [Es] coefficient vector, a cast and a scatter.
template <bool DET>
__global__ void __launch_bounds__(256) k_edge_sample_bwd_rows(const double* __restrict__ verts, const int64_t* __restrict__ edges, const Camera* __restrict__ cam,
                                                              const float* __restrict__ f, const int64_t* __restrict__ rows, int64_t n_rows,
                                                              const float* __restrict__ g, int detach_depth, double* grad_verts) {
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k >= n_rows) return;
    const int64_t e = rows[k];
    const double w = (double)f[e] * (double)g[k];
    if (w == 0.0) return;
    const Camera cm = *cam;
    const int64_t ia = edges[2 * e], ib = edges[2 * e + 1];
    Projected pa, pb;
    project_endpoint(cm, load_d3(verts, ia), pa);
    project_endpoint(cm, load_d3(verts, ib), pb);
    const double gx = -(pa.py - pb.py) * w, gy = -(pb.px - pa.px) * w;
    const GradAdd3<DET> add{grad_verts};
    add((int32_t)ia, project_endpoint_backward(cm, pa, gx, gy, detach_depth != 0));
    add((int32_t)ib, project_endpoint_backward(cm, pb, gx, gy, detach_depth != 0));
}

extern "C" {

int drt_dihedral_forward(const double* d_verts, const int64_t* d_e2f, int64_t n_edges, double* d_cos, void* stream) {
    if (n_edges < 0) return fail(DRT_E_INVALID, "negative edge count");
    if (n_edges == 0) return DRT_OK;
    if (!d_verts || !d_e2f || !d_cos) return fail(DRT_E_INVALID, "null pointer argument");
    k_dihedral_fwd<<<(unsigned)((n_edges + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_verts, d_e2f, n_edges, d_cos);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_dihedral_backward(const double* d_verts, const int64_t* d_e2f, int64_t n_edges, const double* d_grad_cos,
                          double* d_grad_verts, void* stream) {
    if (n_edges < 0) return fail(DRT_E_INVALID, "negative edge count");
    if (n_edges == 0) return DRT_OK;
    if (!d_verts || !d_e2f || !d_grad_cos || !d_grad_verts) return fail(DRT_E_INVALID, "null pointer argument");
    if (det_mode()) k_dihedral_bwd<true, 0><<<(unsigned)((n_edges + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_verts, d_e2f, n_edges, d_grad_cos, nullptr, d_grad_verts);
    else k_dihedral_bwd<false, 0><<<(unsigned)((n_edges + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_verts, d_e2f, n_edges, d_grad_cos, nullptr, d_grad_verts);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_sm_loss_fused(const double* d_verts, const int64_t* d_e2f, int64_t n_edges, double* d_loss, double* d_grad_verts, void* stream) {
    if (n_edges < 0) return fail(DRT_E_INVALID, "negative edge count");
    if (n_edges == 0) return DRT_OK;
    if (!d_verts || !d_e2f || !d_loss || !d_grad_verts) return fail(DRT_E_INVALID, "null pointer argument");
    if (det_mode()) k_dihedral_bwd<true, 1><<<(unsigned)((n_edges + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_verts, d_e2f, n_edges, nullptr, d_loss, d_grad_verts);
    else k_dihedral_bwd<false, 1><<<(unsigned)((n_edges + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_verts, d_e2f, n_edges, nullptr, d_loss, d_grad_verts);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_silhouette_flags(const double* d_verts, const int64_t* d_e2f, int64_t n_edges, const double* d_origin3, uint8_t* d_flags, void* stream) {
    if (n_edges < 0) return fail(DRT_E_INVALID, "negative edge count");
    if (n_edges == 0) return DRT_OK;
    if (!d_verts || !d_e2f || !d_origin3 || !d_flags) return fail(DRT_E_INVALID, "null pointer argument");
    k_silhouette_flags<<<(unsigned)((n_edges + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_verts, d_e2f, n_edges, d_origin3, d_flags);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_edge_sample_forward(drt_scene_t* s, const double* d_verts, const int64_t* d_edges, int64_t n_edges, const double* d_camera,
                            const double* d_origin3, int64_t* d_index, float* d_f, uint8_t* d_keep, int resx, int resy, const uint8_t* d_flags, void* stream) {
    CHECK_BUILT(s);
    if (n_edges < 0) return fail(DRT_E_INVALID, "negative edge count");
    if (n_edges == 0) return DRT_OK;
    if (!d_verts || !d_edges || !d_camera || !d_origin3 || !d_index || !d_f) return fail(DRT_E_INVALID, "null pointer argument");
    { int rc = wait_build(s, (hipStream_t)stream); if (rc) return rc; }
    { int rc = ensure_slow_stack(s, (hipStream_t)stream); if (rc) return rc; }
    k_edge_sample_fwd<<<grid_for(2 * n_edges, kTraceBlock, s->grid_trace), kTraceBlock, 0, (hipStream_t)stream>>>(
        trace_ctx(s), d_verts, d_edges, n_edges, reinterpret_cast<const Camera*>(d_camera), d_origin3, d_index, d_f, d_keep, resx, resy, d_flags);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_edge_sample_backward_rows(const double* d_verts, const int64_t* d_edges, int64_t n_edges, const double* d_camera, const float* d_f,
                                  const int64_t* d_rows, int64_t n_rows, const float* d_g, int detach_depth, double* d_grad_verts, void* stream) {
    if (n_edges < 0 || n_rows < 0) return fail(DRT_E_INVALID, "negative count");
    if (n_edges == 0 || n_rows == 0) return DRT_OK;
    if (!d_verts || !d_edges || !d_camera || !d_f || !d_rows || !d_g || !d_grad_verts) return fail(DRT_E_INVALID, "null pointer argument");
    DET_LAUNCH(k_edge_sample_bwd_rows, (unsigned)((n_rows + 255) / 256), 256, (hipStream_t)stream, d_verts, d_edges, reinterpret_cast<const Camera*>(d_camera), d_f, d_rows, n_rows, d_g, detach_depth, d_grad_verts);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_edge_sample_backward(const double* d_verts, const int64_t* d_edges, int64_t n_edges, const double* d_camera, const float* d_f,
                             const double* d_coef, int detach_depth, double* d_grad_verts, void* stream) {
    if (n_edges < 0) return fail(DRT_E_INVALID, "negative edge count");
    if (n_edges == 0) return DRT_OK;
    if (!d_verts || !d_edges || !d_camera || !d_f || !d_coef || !d_grad_verts) return fail(DRT_E_INVALID, "null pointer argument");
    DET_LAUNCH(k_edge_sample_bwd, (unsigned)((n_edges + 255) / 256), 256, (hipStream_t)stream, d_verts, d_edges, n_edges, reinterpret_cast<const Camera*>(d_camera), d_f, d_coef, detach_depth, d_grad_verts, nullptr);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_edge_sample_backward_term(const double* d_verts, const int64_t* d_edges, int64_t n_edges, const double* d_camera, const float* d_f,
                                  const double* d_dterm, const double* d_g, int detach_depth, double* d_grad_verts, void* stream) {
    if (n_edges < 0) return fail(DRT_E_INVALID, "negative edge count");
    if (n_edges == 0) return DRT_OK;
    if (!d_verts || !d_edges || !d_camera || !d_f || !d_dterm || !d_g || !d_grad_verts) return fail(DRT_E_INVALID, "null pointer argument");
    DET_LAUNCH(k_edge_sample_bwd, (unsigned)((n_edges + 255) / 256), 256, (hipStream_t)stream, d_verts, d_edges, n_edges, reinterpret_cast<const Camera*>(d_camera), d_f, d_dterm, detach_depth, d_grad_verts, d_g);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_vh_loss_fused(drt_scene_t* s, const double* d_verts, const int64_t* d_edges, const int64_t* d_e2f, int64_t n_edges,
                      int n_views, const double* const* d_cameras, const double* const* d_origins, const double* const* d_soft_masks,
                      int resx, int resy, int detach_depth, double* d_loss, double* d_grad_verts, void* stream) {
    CHECK_BUILT(s);
    if (n_edges < 0 || n_views < 0 || resx <= 0 || resy <= 0 || n_edges * kVhViews > (int64_t)UINT32_MAX) return fail(DRT_E_INVALID, "bad size argument");
    if (n_edges == 0 || n_views == 0) return DRT_OK;
    if (!d_verts || !d_edges || !d_e2f || !d_cameras || !d_origins || !d_soft_masks || !d_loss || !d_grad_verts) return fail(DRT_E_INVALID, "null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    { int rc = ensure_slow_stack(s, (hipStream_t)stream); if (rc) return rc; }
    const int64_t need = n_edges * std::min(kVhViews, n_views);
    if (need > s->vh_cap) {
        (void)hipFree(s->vh_list); s->vh_list = nullptr; s->vh_cap = 0;
        HIP_TRY(hipMalloc(&s->vh_list, sizeof(uint32_t) * need));
        s->vh_cap = need;
    }
    for (int v0 = 0; v0 < n_views; v0 += kVhViews) {
        const int nv = std::min(kVhViews, n_views - v0);
        VhViews vw{};
        for (int k = 0; k < nv; ++k) {
            if (!d_cameras[v0 + k] || !d_origins[v0 + k] || !d_soft_masks[v0 + k]) return fail(DRT_E_INVALID, "null pointer argument");
            vw.cam[k] = d_cameras[v0 + k]; vw.origin[k] = d_origins[v0 + k]; vw.soft[k] = d_soft_masks[v0 + k];
        }
        HIP_TRY(hipMemsetAsync(s->vcount + 1, 0, sizeof(unsigned), st));
        k_vh_cull<<<grid_for(n_edges * nv, kPathBlock, 4 * s->n_cu), kPathBlock, 0, st>>>(d_verts, d_e2f, n_edges, nv, vw, s->vh_list, s->vcount + 1);
        // (the silhouette flags need the vertices only; the probe rays are the first thing that needs the tree: the flags of eight views
        // are found while the asynchronous build finishes)
        { int rc = wait_build(s, st); if (rc) return rc; }
        DET_LAUNCH(k_vh_fused, 4 * s->n_cu, kTraceBlock, st, trace_ctx(s), d_verts, d_edges, (uint32_t)n_edges, s->vh_list, s->vcount + 1,
                                                        vw, resx, resy, detach_depth, d_loss, d_grad_verts);
    }
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}


int drt_limit_sgd_step(double* d_param, double* d_grad, double* d_buf, int64_t n, double lr, double momentum, int nesterov, int first,
                       double max_abs, void* stream) {
    if (n < 0) return fail(DRT_E_INVALID, "negative size");
    if (n == 0) return DRT_OK;
    if (!d_param || !d_grad || (momentum != 0.0 && !d_buf)) return fail(DRT_E_INVALID, "null pointer argument");
    k_limit_sgd<<<grid_for(n, 256, 1024), 256, 0, (hipStream_t)stream>>>(d_param, d_grad, d_buf, n, lr, momentum, nesterov, first, max_abs, nullptr, nullptr, nullptr, nullptr);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_vh_term(const int64_t* d_index, const uint8_t* d_keep, int64_t n_edges, const double* d_soft_mask, int resx, int resy,
                double* d_loss, double* d_dterm, void* stream) {
    if (n_edges < 0 || resx <= 0 || resy <= 0) return fail(DRT_E_INVALID, "bad sizes");
    if (n_edges == 0) return DRT_OK;
    if (!d_index || !d_keep || !d_soft_mask || !d_loss || !d_dterm) return fail(DRT_E_INVALID, "null pointer argument");
    DET_LAUNCH(k_vh_term, (unsigned)((n_edges + 255) / 256), 256, (hipStream_t)stream, d_index, d_keep, n_edges, d_soft_mask, resx, d_loss, d_dterm);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

int drt_limit_sgd_step3(double* d_param, double* d_grad, double* d_buf, int64_t n, double lr, double momentum, int nesterov, int first,
                        double max_abs, const double* d_terms, const double* d_w3, const double* d_loss_parts, double* d_loss_total, void* stream) {
    if (n < 0) return fail(DRT_E_INVALID, "negative size");
    if (n == 0) return DRT_OK;
    if (!d_param || !d_grad || (momentum != 0.0 && !d_buf) || !d_terms || !d_w3 || ((d_loss_parts == nullptr) != (d_loss_total == nullptr)))
        return fail(DRT_E_INVALID, "null pointer argument");
    k_limit_sgd<<<grid_for(n, 256, 1024), 256, 0, (hipStream_t)stream>>>(d_param, d_grad, d_buf, n, lr, momentum, nesterov, first, max_abs, d_terms, d_w3, d_loss_parts, d_loss_total);
    HIP_TRY(hipGetLastError());
    return DRT_OK;
}

}  // extern "C"
