// tools/bvhq/bvhq.cpp -- EXPERIMENT (not product, not test): how many node visits does the traversal of drt_traverse.h
// need on a binary tree given explicitly (e.g. a SAH build made offline) compared with the Morton LBVH the product builds?
// Same refit / collapse / quantisation / traversal headers as the kernels; only the binary topology differs.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "../../drt_amd/csrc/drt_lbvh.h"
#include "../../drt_amd/csrc/drt_traverse.h"

using namespace drt;

struct Tree {
    std::vector<Node> nodes; std::vector<Node4Q> wide; std::vector<int32_t> lo, hi; std::vector<TriRec> tris;
};

static Box refit(Tree& t, int32_t c, const std::vector<Box>& leaf) {      // returns the box of child reference c, fills ranges
    if (c < 0) return leaf[~c];
    Node& n = t.nodes[c];
    const Box a = refit(t, n.child0, leaf), b = refit(t, n.child1, leaf);
    node_set_child_box(t.nodes[c], 0, a); node_set_child_box(t.nodes[c], 1, b);
    auto rlo = [&](int32_t x) { return x < 0 ? ~x : t.lo[x]; };
    auto rhi = [&](int32_t x) { return x < 0 ? ~x : t.hi[x]; };
    t.lo[c] = std::min(rlo(n.child0), rlo(n.child1)); t.hi[c] = std::max(rhi(n.child0), rhi(n.child1));
    return box_union(a, b);
}

static int32_t* g_per_ray = nullptr;
extern "C" {
void bvhq_per_ray(int32_t* out) { g_per_ray = out; }      // optional: visits of every ray of the next calls
// order[k] = face in slot k; child0/child1[i] for inner node i (root 0): >= 0 inner, < 0 leaf ~slot.  When child0 == nullptr the
// tree is the product's: Morton order (order is then an OUTPUT) + Karras hierarchy.
double bvhq_visits(const int32_t* faces, int64_t n_faces, const float* verts, int64_t n_verts, int32_t* order,
                   const int32_t* child0, const int32_t* child1, const float* rays, int64_t n_rays, int32_t* ID_out, double* leaf_visits_out) {
    const int n = (int)n_faces;
    auto V = [&](int32_t i) { return f3{verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]}; };
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n_verts; ++i) for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], verts[3 * i + a]); hi[a] = fmaxf(hi[a], verts[3 * i + a]); }
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    const float pad = pad_for_extent(fmaxf(ex, fmaxf(ey, ez)));
    Tree t;
    t.nodes.assign(n - 1, Node{}); t.lo.assign(n - 1, 0); t.hi.assign(n - 1, 0); t.wide.assign(n - 1, Node4Q{}); t.tris.resize(n);
    if (!child0) {
        const f3 inv{1.0f / ex, 1.0f / ey, 1.0f / ez};
        std::vector<uint32_t> key(n), idx(n), keys(n);
        const MortonPlan plan = morton_plan(ex, ey, ez);
        for (int i = 0; i < n; ++i)
            key[i] = getenv("BVHQ_PLAIN_MORTON") ? morton30(V(faces[3 * i]), V(faces[3 * i + 1]), V(faces[3 * i + 2]), f3{lo[0], lo[1], lo[2]}, inv)
                                                 : morton_key(V(faces[3 * i]), V(faces[3 * i + 1]), V(faces[3 * i + 2]), f3{lo[0], lo[1], lo[2]}, inv, plan);
        std::iota(idx.begin(), idx.end(), 0u);
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        for (int k = 0; k < n; ++k) { keys[k] = key[idx[k]]; order[k] = (int32_t)idx[k]; }
        for (int i = 0; i < n - 1; ++i) { int32_t l, r, a, b; lbvh_children(keys.data(), n, i, l, r, a, b); t.nodes[i].child0 = l; t.nodes[i].child1 = r; }
    } else {
        for (int i = 0; i < n - 1; ++i) { t.nodes[i].child0 = child0[i]; t.nodes[i].child1 = child1[i]; }
    }
    std::vector<Box> leaf(n);
    for (int k = 0; k < n; ++k) {
        const int32_t f = order[k];
        const f3 a = V(faces[3 * f]), b = V(faces[3 * f + 1]), c = V(faces[3 * f + 2]);
        t.tris[k] = make_tri(a, b, c, f, hit_margin(pad));
        leaf[k] = box_of_tri(a, b, c, pad);
    }
    refit(t, 0, leaf);
    for (int i = 0; i < n - 1; ++i)
        if (i == 0 || t.hi[i] - t.lo[i] + 1 > kLeafMax) { Node4 full; collapse4(t.nodes.data(), t.lo.data(), t.hi.data(), n, i, full); t.wide[i] = node4_quantize(full); }
    int32_t fast[8], slow[512];
    Stack st; st.fast = fast; st.stride = 1; st.depth_fast = 8; st.slow = slow; st.sp = 0;
    double total = 0, leafs = 0;
    for (int64_t r = 0; r < n_rays; ++r) {
        const f3 o{rays[6 * r], rays[6 * r + 1], rays[6 * r + 2]}, d{rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]};
        TravState s;
        trav_init(s, st, o, d);
        uint64_t vis = 1, lv = 0;
        for (;;) {
            const bool is_leaf = s.cur < 0;
            lv += is_leaf;
            if (trav_step<false>(t.wide.data(), t.tris.data(), s, st)) break;
            ++vis;
        }
        total += (double)vis; leafs += (double)lv;
        ID_out[r] = s.best_face;
        if (g_per_ray) g_per_ray[r] = (int32_t)vis;
    }
    if (leaf_visits_out) *leaf_visits_out = leafs / (double)n_rays;
    return total / (double)n_rays;
}
}

// ---- width-W variant (float boxes, tool only): what would a BVH8 cost in node visits? ------------------------------------------
// Same Morton order, same binary radix tree, same leaf boxes and triangle test as above; the wide node rooted at a binary node opens
// its largest-area open child until it has W children (the product's collapse4 rule with W = 4), and the traversal takes the nearest
// hit child next and pushes the others in slot order (trav_inner's rule).  Boxes are NOT quantised here: with W = 4 the counts are
// within 1 % of the product's, which is the check that the comparison below means something.
struct WideNode { int n; Box box[16]; int32_t child[16]; };      // child >= 0: wide node index (= its binary root); < 0: ~slot
static void collapse_w(const Tree& t, int W, int i, WideNode& out) {
    int32_t ref[16]; Box box[16]; bool open[16];
    const Node& r = t.nodes[i];
    int k = 2;
    ref[0] = r.child0; box[0] = node_child_box(r, 0); open[0] = r.child0 >= 0;
    ref[1] = r.child1; box[1] = node_child_box(r, 1); open[1] = r.child1 >= 0;
    while (k < W) {
        int pick = -1; float best = -1.0f;
        for (int j = 0; j < k; ++j) if (open[j]) { const float a = box_area(box[j]); if (a > best) { best = a; pick = j; } }
        if (pick < 0) break;
        const Node& o = t.nodes[ref[pick]];
        ref[pick] = o.child0; box[pick] = node_child_box(o, 0); open[pick] = o.child0 >= 0;
        ref[k] = o.child1; box[k] = node_child_box(o, 1); open[k] = o.child1 >= 0;
        ++k;
    }
    out.n = k;
    for (int j = 0; j < k; ++j) { out.box[j] = box[j]; out.child[j] = ref[j]; }
}

extern "C" void bvhq_visits_wide(const int32_t* faces, int64_t n_faces, const float* verts, int64_t n_verts, int W,
                                 const float* rays, int64_t n_rays, int32_t* ID_out, double* stats /* [6] */) {
    const int n = (int)n_faces;
    auto V = [&](int32_t i) { return f3{verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]}; };
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n_verts; ++i) for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], verts[3 * i + a]); hi[a] = fmaxf(hi[a], verts[3 * i + a]); }
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    const float pad = pad_for_extent(fmaxf(ex, fmaxf(ey, ez)));
    Tree t;
    t.nodes.assign(n - 1, Node{}); t.lo.assign(n - 1, 0); t.hi.assign(n - 1, 0); t.tris.resize(n);
    const f3 inv{1.0f / ex, 1.0f / ey, 1.0f / ez};
    std::vector<uint32_t> key(n), idx(n), keys(n);
    const MortonPlan plan = morton_plan(ex, ey, ez);
    for (int i = 0; i < n; ++i) key[i] = morton_key(V(faces[3 * i]), V(faces[3 * i + 1]), V(faces[3 * i + 2]), f3{lo[0], lo[1], lo[2]}, inv, plan);
    std::iota(idx.begin(), idx.end(), 0u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    for (int k = 0; k < n; ++k) keys[k] = key[idx[k]];
    for (int i = 0; i < n - 1; ++i) { int32_t l, r, a, b; lbvh_children(keys.data(), n, i, l, r, a, b); t.nodes[i].child0 = l; t.nodes[i].child1 = r; }
    std::vector<Box> leaf(n);
    for (int k = 0; k < n; ++k) {
        const int32_t f = (int32_t)idx[k];
        const f3 a = V(faces[3 * f]), b = V(faces[3 * f + 1]), c = V(faces[3 * f + 2]);
        t.tris[k] = make_tri(a, b, c, f, hit_margin(pad));
        leaf[k] = box_of_tri(a, b, c, pad);
    }
    refit(t, 0, leaf);
    std::vector<WideNode> wide(n - 1);
    std::vector<char> made(n - 1, 0);
    std::vector<int32_t> todo{0};
    int64_t n_wide = 0; double fill = 0;
    while (!todo.empty()) {
        const int32_t i = todo.back(); todo.pop_back();
        if (made[i]) continue;
        made[i] = 1; collapse_w(t, W, i, wide[i]); ++n_wide; fill += wide[i].n;
        for (int j = 0; j < wide[i].n; ++j) if (wide[i].child[j] >= 0) todo.push_back(wide[i].child[j]);
    }
    double inner = 0, leafs = 0, tests = 0; int64_t max_total = 0, max_inner = 0;
    std::vector<int32_t> stack; stack.reserve(1024);
    for (int64_t r = 0; r < n_rays; ++r) {
        const f3 o{rays[6 * r], rays[6 * r + 1], rays[6 * r + 2]}, d{rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]};
        const f3 iv{safe_inv(d.x), safe_inv(d.y), safe_inv(d.z)};
        float best_t = INFINITY; int32_t best_face = -1;
        int64_t vi = 0, vl = 0;
        stack.clear();
        int32_t cur = 0;
        for (;;) {
            if (cur >= 0) {
                ++vi;
                const WideNode& w = wide[cur];
                tests += w.n;
                float tn[16]; bool h[16]; int near = -1;
                for (int j = 0; j < w.n; ++j) {
                    const Box& b = w.box[j];
                    const float x0 = (b.lox - o.x) * iv.x, x1 = (b.hix - o.x) * iv.x, y0 = (b.loy - o.y) * iv.y, y1 = (b.hiy - o.y) * iv.y,
                                z0 = (b.loz - o.z) * iv.z, z1 = (b.hiz - o.z) * iv.z;
                    const float a = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
                    const float f = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), best_t));
                    tn[j] = a; h[j] = a <= f;
                    if (h[j] && (near < 0 || a < tn[near])) near = j;
                }
                if (near >= 0) {
                    for (int j = 0; j < w.n; ++j) if (h[j] && j != near) stack.push_back(w.child[j]);
                    cur = w.child[near];
                    continue;
                }
            } else {
                ++vl;
                const TriRec& tr = t.tris[~cur];
                float tt;
                if (tri_hit(o, d, f3{tr.v0x, tr.v0y, tr.v0z}, f3{tr.e1x, tr.e1y, tr.e1z}, f3{tr.e2x, tr.e2y, tr.e2z}, tr.margin, tt))
                    if (tt < best_t || (tt == best_t && tr.face < best_face)) { best_t = tt; best_face = tr.face; }
            }
            if (stack.empty()) break;
            cur = stack.back(); stack.pop_back();
        }
        inner += (double)vi; leafs += (double)vl;
        max_total = std::max(max_total, vi + vl); max_inner = std::max(max_inner, vi);
        if (ID_out) ID_out[r] = best_face;
    }
    stats[0] = inner / (double)n_rays; stats[1] = leafs / (double)n_rays; stats[2] = (double)max_inner; stats[3] = (double)max_total;
    stats[4] = fill / (double)n_wide; stats[5] = tests / (double)n_rays;
}
