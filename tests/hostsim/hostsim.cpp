// tests/hostsim/hostsim.cpp -- CPU unit-test harness for the device math.  TEST ONLY.
//
// Compiles the plain-C++ per-item headers of drt_amd/csrc (the same code the gfx950
// kernels inline) with g++ and drives them with sequential loops, so the LBVH build
// logic, the traversal and the hand-derived adjoints can be checked against the oracle
// in the GPU-less CI container.  Nothing in drt_amd/ loads this library: the product
// path is the HIP library and fails loudly without it.
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include "../../drt_amd/csrc/drt_closest.h"
#include "../../drt_amd/csrc/drt_edge.h"
#include "../../drt_amd/csrc/drt_fixed.h"
#include "../../drt_amd/csrc/drt_lbvh.h"
#include "../../drt_amd/csrc/drt_path.h"
#include "../../drt_amd/csrc/drt_raster.h"

using namespace drt;

struct HsScene {
    std::vector<int32_t> faces;
    std::vector<float> verts;
    std::vector<Node> nodes;       // binary radix tree (build intermediate)
    std::vector<Node4Q> wide;      // what the traversal reads (quantised)
    std::vector<int32_t> range_lo, range_hi;
    std::vector<TriRec> tris;
    std::vector<int32_t> parent_inner, parent_leaf;
    std::vector<uint32_t> keys, idx;
    float pad = 0.f;
    int height = 0;
};

static f3 vert(const HsScene& s, int32_t i) { return f3{s.verts[3 * i], s.verts[3 * i + 1], s.verts[3 * i + 2]}; }

static void build(HsScene& s) {
    const int n = (int)(s.faces.size() / 3);
    const int64_t nv = (int64_t)s.verts.size() / 3;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < nv; ++i)
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], s.verts[3 * i + a]);
            hi[a] = fmaxf(hi[a], s.verts[3 * i + a]);
        }
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    const f3 inv{ex > 0 ? 1.0f / ex : 0.f, ey > 0 ? 1.0f / ey : 0.f, ez > 0 ? 1.0f / ez : 0.f};
    s.pad = pad_for_extent(fmaxf(ex, fmaxf(ey, ez)));
    std::vector<uint32_t> key(n);
    const MortonPlan plan = morton_plan(ex, ey, ez);
    for (int i = 0; i < n; ++i)
        key[i] = morton_key(vert(s, s.faces[3 * i]), vert(s, s.faces[3 * i + 1]), vert(s, s.faces[3 * i + 2]), f3{lo[0], lo[1], lo[2]}, inv, plan);
    s.idx.resize(n);
    std::iota(s.idx.begin(), s.idx.end(), 0u);
    std::stable_sort(s.idx.begin(), s.idx.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    s.keys.resize(n);
    for (int k = 0; k < n; ++k) s.keys[k] = key[s.idx[k]];
    const int inner = n > 1 ? n - 1 : 1;
    s.nodes.assign(inner, Node{});
    s.parent_inner.assign(inner, -1);
    s.parent_leaf.assign(n, -1);
    s.range_lo.assign(inner, 0);
    s.range_hi.assign(inner, n - 1);
    s.wide.assign(inner, Node4Q{});
    s.tris.resize(n);
    if (n == 0) return;
    if (n == 1) {
        node_set_child_box(s.nodes[0], 0, box_empty());
        node_set_child_box(s.nodes[0], 1, box_empty());
        s.nodes[0].child0 = ~0; s.nodes[0].child1 = ~0;
        s.parent_leaf[0] = 0;
    } else {
        for (int i = 0; i < n - 1; ++i) {
            int32_t l, r;
            lbvh_children(s.keys.data(), n, i, l, r, s.range_lo[i], s.range_hi[i]);
            s.nodes[i].child0 = l; s.nodes[i].child1 = r;
            if (l >= 0) s.parent_inner[l] = i * 2; else s.parent_leaf[~l] = i * 2;
            if (r >= 0) s.parent_inner[r] = i * 2 + 1; else s.parent_leaf[~r] = i * 2 + 1;
        }
    }
    std::vector<uint32_t> flags(inner, n == 1 ? 1u : 0u);
    for (int k = 0; k < n; ++k) {
        const int32_t face = (int32_t)s.idx[k];
        const f3 a = vert(s, s.faces[3 * face]), b = vert(s, s.faces[3 * face + 1]), c = vert(s, s.faces[3 * face + 2]);
        s.tris[k] = make_tri(a, b, c, face, hit_margin(s.pad));
        Box box = box_of_tri(a, b, c, s.pad);
        int32_t link = s.parent_leaf[k];
        while (link >= 0) {
            const int p = link >> 1, slot = link & 1;
            node_set_child_box(s.nodes[p], slot, box);
            if (flags[p]++ == 0) break;
            box = box_union(box, node_child_box(s.nodes[p], slot ^ 1));
            link = s.parent_inner[p];
        }
    }
    for (int k = 0; k < n; ++k) {
        int depth = 0;
        for (int32_t link = s.parent_leaf[k]; link >= 0; link = s.parent_inner[link >> 1]) ++depth;
        s.height = std::max(s.height, depth);
    }
    for (int i = 0; i < inner; ++i) {
        const bool wide_root = i == 0 || s.range_hi[i] - s.range_lo[i] + 1 > kLeafMax;   // any of them may be adopted by a wide parent
        if (wide_root) { Node4 full; collapse4(s.nodes.data(), s.range_lo.data(), s.range_hi.data(), n, i, full); s.wide[i] = node4_quantize(full); }
    }
}

struct HostStack {
    int32_t fast[8];      // deliberately shallow so the overflow path is exercised
    int32_t slow[kStackSlow + 64];
    Stack st;
    HostStack() { st.fast = fast; st.stride = 1; st.depth_fast = 8; st.slow = slow; st.sp = 0; }
};

static PathCtx path_ctx(const HsScene* s, const double* verts64, double ior_int, double ior_ext) {
    return PathCtx{TraceCtx{s->wide.data(), s->tris.data(), (int)s->tris.size(), nullptr}, s->faces.data(), verts64, ior_int, ior_ext};
}

extern "C" {

void* hs_create(const int32_t* faces, int64_t n_faces, const float* verts, int64_t n_verts) {
    HsScene* s = new HsScene();
    s->faces.assign(faces, faces + 3 * n_faces);
    s->verts.assign(verts, verts + 3 * n_verts);
    build(*s);
    return s;
}
void hs_destroy(void* h) { delete (HsScene*)h; }
int hs_height(void* h) { return ((HsScene*)h)->height; }
void hs_sorted_faces(void* h, int32_t* out) {
    HsScene* s = (HsScene*)h;
    for (size_t k = 0; k < s->idx.size(); ++k) out[k] = (int32_t)s->idx[k];
}

// Every ancestor box must enclose each leaf's padded box and the links must be consistent.
int64_t hs_check(void* h) {
    HsScene* s = (HsScene*)h;
    const int n = (int)s->tris.size();
    int64_t bad = 0;
    for (int k = 0; k < n; ++k) {
        const TriRec& t = s->tris[k];
        const f3 a{t.v0x, t.v0y, t.v0z}, b{t.v0x + t.e1x, t.v0y + t.e1y, t.v0z + t.e1z}, c{t.v0x + t.e2x, t.v0y + t.e2y, t.v0z + t.e2z};
        const Box leaf = box_of_tri(a, b, c, 0.5f * s->pad);
        int32_t link = s->parent_leaf[k], expect = ~k;
        while (link >= 0) {
            const int p = link >> 1, slot = link & 1;
            const Node& nd = s->nodes[p];
            if ((slot == 0 ? nd.child0 : nd.child1) != expect) ++bad;
            if (!box_contains(node_child_box(nd, slot), leaf)) ++bad;
            expect = p;
            link = s->parent_inner[p];
        }
        if (n > 1 && expect != 0) ++bad;
    }
    // wide tree: every triangle slot is referenced by exactly one leaf, inside a box that encloses it
    std::vector<int> seen(n, 0);
    std::vector<int32_t> todo;
    if (n > 0) todo.push_back(0);
    while (!todo.empty()) {
        const Node4Q& nd = s->wide[todo.back()];
        todo.pop_back();
        for (int k = 0; k < 4; ++k) {
            const int32_t c = nd.child[k];
            if (c == kEmptyChild) continue;
            if (c >= 0) { todo.push_back(c); continue; }
            const int first = (~c) >> kLeafBits, count = ((~c) & (kLeafMax - 1)) + 1;
            for (int j = first; j < first + count; ++j) {
                if (j < 0 || j >= n) { ++bad; continue; }
                ++seen[j];
                const TriRec& t = s->tris[j];
                const f3 a{t.v0x, t.v0y, t.v0z}, b{t.v0x + t.e1x, t.v0y + t.e1y, t.v0z + t.e1z}, c2{t.v0x + t.e2x, t.v0y + t.e2y, t.v0z + t.e2z};
                if (!box_contains(node4q_box(nd, k), box_of_tri(a, b, c2, 0.5f * s->pad))) ++bad;
            }
        }
    }
    for (int j = 0; j < n; ++j) if (seen[j] != 1) ++bad;
    return bad;
}

void hs_intersect(void* h, const float* rays, int64_t n, float* T, int32_t* ID, int any, uint32_t* visits) {
    HsScene* s = (HsScene*)h;
    HostStack hs;
    for (int64_t i = 0; i < n; ++i) {
        const f3 o{rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, d{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]};
        uint32_t v = 0;
        const Hit r = any ? traverse<true>(s->wide.data(), s->tris.data(), (int)s->tris.size(), o, d, hs.st, &v)
                          : traverse<false>(s->wide.data(), s->tris.data(), (int)s->tris.size(), o, d, hs.st, &v);
        T[i] = r.t;
        ID[i] = r.face;
        if (visits) visits[i] = v;
    }
}

void hs_closest_point(void* h, const double* points, int64_t n, double* dist, int32_t* face, double* closest) {
    HsScene* s = (HsScene*)h;
    HostStack hs;
    for (int64_t i = 0; i < n; ++i) {
        const Closest r = closest_point(s->wide.data(), s->tris.data(), (int)s->tris.size(), s->faces.data(), s->verts.data(), load_d3(points, i), hs.st);
        dist[i] = sqrt(r.dist2);
        face[i] = r.face;
        store_d3(closest, i, r.point);
    }
}

// the remesher's projection query (k_rm_closest_near): bounded first, unbounded when nothing is strictly inside the bound; found[i] = the
// bounded search was enough
void hs_closest_near(void* h, const double* points, const double* hint, int64_t n, double* closest, int32_t* face, uint8_t* found) {
    HsScene* s = (HsScene*)h;
    HostStack hs;
    for (int64_t i = 0; i < n; ++i) {
        const d3 p = load_d3(points, i);
        Closest r = closest_point(s->wide.data(), s->tris.data(), (int)s->tris.size(), s->faces.data(), s->verts.data(), p, hs.st, hint[i] * hint[i]);
        found[i] = r.face >= 0;
        if (r.face < 0) r = closest_point(s->wide.data(), s->tris.data(), (int)s->tris.size(), s->faces.data(), s->verts.data(), p, hs.st);
        face[i] = r.face;
        store_d3(closest, i, r.point);
    }
}

// the remesher's surface-distance verdict (within_distance) for each point against its own radius
void hs_within_distance(void* h, const double* points, const double* radius, int64_t n, uint8_t* within) {
    HsScene* s = (HsScene*)h;
    HostStack hs;
    for (int64_t i = 0; i < n; ++i)
        within[i] = within_distance(s->wide.data(), s->tris.data(), (int)s->tris.size(), s->faces.data(), s->verts.data(), load_d3(points, i), radius[i], hs.st) ? 1 : 0;
}

void hs_render_forward(void* h, const double* verts64, const double* origin, const double* dir, int64_t n, double ior_int,
                       double ior_ext, double* out_ori, double* out_dir, uint8_t* mask, int32_t* face1, int32_t* face2) {
    const PathCtx c = path_ctx((HsScene*)h, verts64, ior_int, ior_ext);
    HostStack hs;
    for (int64_t i = 0; i < n; ++i) {
        int32_t f1, f2;
        d3 eo{0, 0, 0}, ed{0, 0, 0};
        const bool ok = trace_path(c, hs.st, load_d3(origin, i), load_d3(dir, i), f1, f2, eo, ed);
        const d3 z{0, 0, 0};
        store_d3(out_ori, i, ok ? eo : z);
        store_d3(out_dir, i, ok ? ed : z);
        mask[3 * i] = mask[3 * i + 1] = mask[3 * i + 2] = ok ? 1 : 0;
        face1[i] = f1;
        face2[i] = f2;
    }
}

struct HostAdd {
    double* g;
    void operator()(int32_t v, d3 a) const { g[3 * v] += a.x; g[3 * v + 1] += a.y; g[3 * v + 2] += a.z; }
};

void hs_render_backward(void* h, const double* verts64, const double* origin, const double* dir, int64_t n, double ior_int,
                        double ior_ext, const int32_t* face1, const int32_t* face2, const double* g_ori, const double* g_dir,
                        double* grad_verts) {
    const PathCtx c = path_ctx((HsScene*)h, verts64, ior_int, ior_ext);
    const d3 z{0, 0, 0};
    for (int64_t i = 0; i < n; ++i) {
        if (face2[i] < 0) continue;
        path_recompute_backward(c, load_d3(origin, i), load_d3(dir, i), face1[i], face2[i], g_ori ? load_d3(g_ori, i) : z,
                                g_dir ? load_d3(g_dir, i) : z, HostAdd{grad_verts});
    }
}

void hs_ray_loss(const double* out_ori, const double* out_dir, const uint8_t* mask, const double* screen_pixel,
                 const uint8_t* valid, int64_t n, double* loss, double* g_out_dir) {
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        d3 g{0, 0, 0};
        if (valid[i] && mask[3 * i]) acc += ray_loss_term(load_d3(out_ori, i), load_d3(out_dir, i), load_d3(screen_pixel, i), g);
        if (g_out_dir) store_d3(g_out_dir, i, g);
    }
    *loss += acc;
}

void hs_fused(void* h, const double* verts64, const double* origin, const double* dir, const double* screen_pixel,
              const uint8_t* valid, int64_t n, double ior_int, double ior_ext, double* loss, double* grad_verts, int64_t* n_valid) {
    const PathCtx c = path_ctx((HsScene*)h, verts64, ior_int, ior_ext);
    HostStack hs;
    for (int64_t i = 0; i < n; ++i) {
        if (!valid[i]) continue;
        int32_t f1, f2;
        d3 eo, ed, g;
        const d3 o = load_d3(origin, i), d = load_d3(dir, i);
        if (!trace_path(c, hs.st, o, d, f1, f2, eo, ed)) continue;
        *loss += ray_loss_term(eo, ed, load_d3(screen_pixel, i), g);
        if (n_valid) ++*n_valid;
        path_recompute_backward(c, o, d, f1, f2, d3{0, 0, 0}, g, HostAdd{grad_verts});
    }
}

// Single bounce forward + adjoint on explicit triangles: unit test of drt_shade.h.
void hs_bounce(const double* o, const double* d, const double* tri, int64_t n, double ior_int, double ior_ext,
               const double* g_new_o, const double* g_wt, double* new_o, double* wt, uint8_t* tir, double* t_out,
               double* g_tri, double* g_o, double* g_d) {
    for (int64_t i = 0; i < n; ++i) {
        Bounce b;
        bounce_forward(load_d3(o, i), load_d3(d, i), load_d3(tri, 3 * i), load_d3(tri, 3 * i + 1), load_d3(tri, 3 * i + 2), ior_ext, ior_int, b);
        store_d3(new_o, i, b.new_o);
        store_d3(wt, i, b.wt);
        tir[i] = b.tir;
        t_out[i] = b.t;
        d3 ga{0, 0, 0}, gb{0, 0, 0}, gc{0, 0, 0}, go, gd;
        bounce_backward(b, load_d3(g_new_o, i), load_d3(g_wt, i), ga, gb, gc, go, gd);
        store_d3(g_tri, 3 * i, ga); store_d3(g_tri, 3 * i + 1, gb); store_d3(g_tri, 3 * i + 2, gc);
        store_d3(g_o, i, go);
        store_d3(g_d, i, gd);
    }
}

// ---- silhouette / smoothness branches (drt_edge.h) ----
static void face64(const double* verts, const int64_t* f, d3& v0, d3& v1, d3& v2) {
    v0 = load_d3(verts, f[0]); v1 = load_d3(verts, f[1]); v2 = load_d3(verts, f[2]);
}

// mode 0: cos only; mode 1: + adjoint for g_cos; mode 2: sm_loss fused (loss += sum -log(1+cos), grad)
void hs_dihedral(const double* verts, const int64_t* e2f, int64_t n, int mode, double* cos_out, const double* g_cos,
                 double* loss, double* grad_verts) {
    for (int64_t e = 0; e < n; ++e) {
        d3 v0, v1, v2;
        FaceNormal a, b;
        const int64_t* fa = e2f + 6 * e;
        const int64_t* fb = fa + 3;
        face64(verts, fa, v0, v1, v2); face_normal(v0, v1, v2, a);
        face64(verts, fb, v0, v1, v2); face_normal(v0, v1, v2, b);
        const double c = dot(a.n, b.n);
        if (cos_out) cos_out[e] = c;
        if (mode == 0) continue;
        double g = mode == 1 ? g_cos[e] : -1.0 / (1.0 + c);
        if (mode == 2) *loss += -log(1.0 + c);
        const d3 z{0, 0, 0};
        d3 g0 = z, g1 = z, g2 = z;
        const HostAdd add{grad_verts};
        face_normal_backward(a, g * b.n, g0, g1, g2);
        add((int32_t)fa[0], g0); add((int32_t)fa[1], g1); add((int32_t)fa[2], g2);
        g0 = z; g1 = z; g2 = z;
        face_normal_backward(b, g * a.n, g0, g1, g2);
        add((int32_t)fb[0], g0); add((int32_t)fb[1], g1); add((int32_t)fb[2], g2);
    }
}

void hs_silhouette_flags(const double* verts, const int64_t* e2f, int64_t n, const double* origin3, uint8_t* flags) {
    const d3 o{origin3[0], origin3[1], origin3[2]};
    for (int64_t e = 0; e < n; ++e) {
        d3 a0, b0, v1, v2;
        FaceNormal a, b;
        face64(verts, e2f + 6 * e, a0, v1, v2); face_normal(a0, v1, v2, a);
        face64(verts, e2f + 6 * e + 3, b0, v1, v2); face_normal(b0, v1, v2, b);
        flags[e] = silhouette_flag(a, a0, b, b0, o) ? 1 : 0;
    }
}

void hs_edge_sample_forward(void* h, const double* verts, const int64_t* edges, int64_t n, const double* camera50,
                            const double* origin3, int64_t* index, float* f_out) {
    HsScene* s = (HsScene*)h;
    const Camera cm = *reinterpret_cast<const Camera*>(camera50);
    const d3 o{origin3[0], origin3[1], origin3[2]};
    HostStack hs;
    for (int64_t e = 0; e < n; ++e) {
        Projected pa, pb;
        project_endpoint(cm, load_d3(verts, edges[2 * e]), pa);
        project_endpoint(cm, load_d3(verts, edges[2 * e + 1]), pb);
        EdgeSample es;
        edge_sample(cm, pa, pb, o, es);
        const bool hu = traverse<true>(s->wide.data(), s->tris.data(), (int)s->tris.size(), to_f32(o), to_f32(es.dir_up), hs.st).face >= 0;
        const bool hl = traverse<true>(s->wide.data(), s->tris.data(), (int)s->tris.size(), to_f32(o), to_f32(es.dir_lo), hs.st).face >= 0;
        f_out[e] = (hu ? 1.0f : 0.0f) - (hl ? 1.0f : 0.0f);
        index[2 * e] = (int64_t)es.midx;
        index[2 * e + 1] = (int64_t)es.midy;
    }
}

void hs_edge_sample_backward(const double* verts, const int64_t* edges, int64_t n, const double* camera50, const float* f,
                             const double* coef, int detach_depth, double* grad_verts) {
    const Camera cm = *reinterpret_cast<const Camera*>(camera50);
    const HostAdd add{grad_verts};
    for (int64_t e = 0; e < n; ++e) {
        const double w = (double)f[e] * coef[e];
        if (w == 0.0) continue;
        Projected pa, pb;
        project_endpoint(cm, load_d3(verts, edges[2 * e]), pa);
        project_endpoint(cm, load_d3(verts, edges[2 * e + 1]), pb);
        const double gx = -(pa.py - pb.py) * w, gy = -(pb.px - pa.px) * w;
        add((int32_t)edges[2 * e], project_endpoint_backward(cm, pa, gx, gy, detach_depth != 0));
        add((int32_t)edges[2 * e + 1], project_endpoint_backward(cm, pb, gx, gy, detach_depth != 0));
    }
}

// ---- projected primary visibility (drt_raster.h): the per-image model, per-ray verification, projected pixel boxes, and the
// whole decision for a grid of rays -- closest hit per pixel by testing each triangle against the pixels of its padded box only.
int hs_fit_view(const double* origin, const double* dir, int w, int h, double* model14 /* o[3], minv[9], ok, lattice_ok */) {
    ViewModel vm;
    const int64_t i00 = 0, iW0 = w - 1, i0H = (int64_t)(h - 1) * w, iWH = i0H + (w - 1);
    const bool ok = fit_view_model(load_d3(origin, i00), load_d3(dir, i00), load_d3(dir, iW0), load_d3(dir, i0H), load_d3(dir, iWH), (double)(w - 1), (double)(h - 1), vm);
    vm.ok = ok ? 1 : 0;
    bool lattice = ok;
    for (int sy = 0; sy < 8 && lattice; ++sy)
        for (int sx = 0; sx < 8 && lattice; ++sx) {
            const int x = (int)(((int64_t)(w - 1) * sx) / 7), y = (int)(((int64_t)(h - 1) * sy) / 7);
            const int64_t i = (int64_t)y * w + x;
            lattice = view_verify(vm, load_d3(origin, i), load_d3(dir, i), (double)x, (double)y);
        }
    for (int k = 0; k < 3; ++k) model14[k] = vm.o[k];
    for (int k = 0; k < 9; ++k) model14[3 + k] = vm.minv[k];
    model14[12] = vm.ok; model14[13] = lattice ? 1 : 0;
    return ok ? 1 : 0;
}

// every ray verified against the model: flags [w*h]
void hs_verify_rays(const double* model14, const double* origin, const double* dir, int w, int h, uint8_t* flags) {
    ViewModel vm;
    for (int k = 0; k < 3; ++k) vm.o[k] = model14[k];
    for (int k = 0; k < 9; ++k) vm.minv[k] = model14[3 + k];
    vm.ok = 1; vm.all = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int64_t i = (int64_t)y * w + x;
            flags[i] = view_verify(vm, load_d3(origin, i), load_d3(dir, i), (double)x, (double)y) ? 1 : 0;
        }
}

// Primary hits of one image by projection (what k_raster + k_cull decide on the GPU): T, ID as hs_intersect; returns the number
// of (triangle, pixel) tests, or -1 when some triangle has no projection bound (camera plane).
int64_t hs_raster(void* hnd, const double* model14, const double* origin, const double* dir, int w, int h, float* T, int32_t* ID) {
    HsScene* s = (HsScene*)hnd;
    ViewModel vm;
    for (int k = 0; k < 3; ++k) vm.o[k] = model14[k];
    for (int k = 0; k < 9; ++k) vm.minv[k] = model14[3 + k];
    vm.ok = 1; vm.all = 0;
    std::vector<unsigned long long> key((size_t)w * h, kRasterEmpty);
    const d3 o{vm.o[0], vm.o[1], vm.o[2]};
    const f3 o32 = to_f32(o);
    int64_t tests = 0;
    for (const TriRec& t : s->tris) {
        const PixelBox box = project_tri_box(vm, o32, t, w, h);        // the function k_raster calls
        if (box.unsafe) return -1;
        for (int y = box.y0; y <= box.y1; ++y)
            for (int x = box.x0; x <= box.x1; ++x) {
                const int64_t i = (int64_t)y * w + x;
                ++tests;
                float tt;
                if (tri_hit(o32, to_f32(load_d3(dir, i)), f3{t.v0x, t.v0y, t.v0z}, f3{t.e1x, t.e1y, t.e1z}, f3{t.e2x, t.e2y, t.e2z}, t.margin, tt)) {
                    const unsigned long long k = raster_key(tt, t.face);
                    if (k < key[i]) key[i] = k;
                }
            }
    }
    for (int64_t i = 0; i < (int64_t)w * h; ++i) {
        if (key[i] == kRasterEmpty) { T[i] = -1.0f; ID[i] = -1; continue; }
        const uint32_t tb = (uint32_t)(key[i] >> 32);
        memcpy(&T[i], &tb, 4);
        ID[i] = (int32_t)(uint32_t)key[i];
    }
    return tests;
}

// drt_lbvh.h::morton_plan / morton_key: axis[30], pos[30], bits[3] of the plan for a scene box of the given extents, and the key of a
// degenerate "triangle" at point p inside the box [0, ext].
void hs_morton_plan(const float* ext3, uint8_t* axis30, uint8_t* pos30, uint8_t* bits3) {
    const MortonPlan p = morton_plan(ext3[0], ext3[1], ext3[2]);
    for (int k = 0; k < 30; ++k) { axis30[k] = p.axis[k]; pos30[k] = p.pos[k]; }
    for (int a = 0; a < 3; ++a) bits3[a] = p.bits[a];
}
uint32_t hs_morton_key(const float* ext3, const float* p3) {
    const MortonPlan p = morton_plan(ext3[0], ext3[1], ext3[2]);
    const f3 v{p3[0], p3[1], p3[2]};
    const f3 inv{ext3[0] > 0 ? 1.0f / ext3[0] : 0.f, ext3[1] > 0 ? 1.0f / ext3[1] : 0.f, ext3[2] > 0 ? 1.0f / ext3[2] : 0.f};
    return morton_key(v, v, v, f3{0.f, 0.f, 0.f}, inv, p);
}

// drt_fixed.h: float64 -> 128-bit fixed point and back (the deterministic accumulation mode), checked against Python integers
void hs_fx_from(double x, int64_t* hi, uint64_t* lo, uint32_t* flags) { Fx128 r; *flags = fx_from_double(x, r); *hi = r.hi; *lo = r.lo; }
double hs_fx_to(int64_t hi, uint64_t lo, uint32_t flags) { return fx_to_double(Fx128{hi, lo}, flags); }
// the sum of n float64 values the way a kernel accumulates it (any order gives the same cells), converted once
double hs_fx_sum(const double* x, int64_t n) {
    FxAcc a;
    for (int64_t i = 0; i < n; ++i) a.add(x[i]);
    return fx_to_double(a.v, a.flags);
}

}  // extern "C"
