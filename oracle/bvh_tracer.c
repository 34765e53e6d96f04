/*
 * oracle/bvh_tracer.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The same tracer contract as oracle/tracer.c (closest hit, the float32 Moller-Trumbore transcription, minimum t, equal
 * t -> lowest face id, miss -> T = -1, ID = -1) behind a bounding-volume hierarchy instead of the loop over every face.
 * It exists so that bench.py's cpu_baseline can also quote the reference-shaped CPU path with a REASONABLE tracer, not
 * only with the brute force BASELINE.json's configs[0] names; it shares no code with drt_amd/ (own median-split tree,
 * float64 slab test on boxes padded by 1e-4 of the scene extent) and tests/test_oracle_golden.py checks it against the
 * brute force bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "hit_point.h"
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float lo[3], hi[3]; int32_t left, right, first, count; } node_t;
typedef struct {
    const int32_t *faces; const float *verts;
    int32_t *order; node_t *nodes; int32_t n_nodes;
    float pad, margin;
} bvh_t;

float oracle_hit_margin(const float *verts, int64_t n_verts);      /* tracer.c */

static float centroid(const bvh_t *b, int32_t f, int axis) {
    const int32_t *t = b->faces + 3 * (int64_t)f;
    return b->verts[3 * (int64_t)t[0] + axis] + b->verts[3 * (int64_t)t[1] + axis] + b->verts[3 * (int64_t)t[2] + axis];
}

static int g_axis; static const bvh_t *g_bvh;   /* qsort context (build is single-threaded) */
static int cmp_centroid(const void *x, const void *y) {
    const int32_t a = *(const int32_t *)x, b = *(const int32_t *)y;
    const float ca = centroid(g_bvh, a, g_axis), cb = centroid(g_bvh, b, g_axis);
    return ca < cb ? -1 : (ca > cb ? 1 : (a < b ? -1 : (a > b)));
}

static int32_t build(bvh_t *b, int32_t first, int32_t count) {
    const int32_t id = b->n_nodes++;
    node_t n;
    for (int a = 0; a < 3; ++a) { n.lo[a] = INFINITY; n.hi[a] = -INFINITY; }
    for (int32_t k = first; k < first + count; ++k)
        for (int c = 0; c < 3; ++c) {
            const float *p = b->verts + 3 * (int64_t)b->faces[3 * (int64_t)b->order[k] + c];
            for (int a = 0; a < 3; ++a) { if (p[a] < n.lo[a]) n.lo[a] = p[a]; if (p[a] > n.hi[a]) n.hi[a] = p[a]; }
        }
    for (int a = 0; a < 3; ++a) { n.lo[a] -= b->pad; n.hi[a] += b->pad; }
    n.first = first; n.count = count; n.left = n.right = -1;
    if (count > 4) {
        int axis = 0;
        for (int a = 1; a < 3; ++a) if (n.hi[a] - n.lo[a] > n.hi[axis] - n.lo[axis]) axis = a;
        g_axis = axis; g_bvh = b;
        qsort(b->order + first, (size_t)count, sizeof(int32_t), cmp_centroid);
        const int32_t mid = count / 2;
        n.left = build(b, first, mid);
        n.right = build(b, first + mid, count - mid);
    }
    b->nodes[id] = n;
    return id;
}

/* entry distance of the ray into the padded box, or -1 when it misses [0, best]; float64 arithmetic on float data */
static double slab(const node_t *n, const double o[3], const double inv[3], double best) {
    double t0 = 0.0, t1 = best;
    for (int a = 0; a < 3; ++a) {
        double ta = ((double)n->lo[a] - o[a]) * inv[a], tb = ((double)n->hi[a] - o[a]) * inv[a];
        if (ta > tb) { const double s = ta; ta = tb; tb = s; }
        if (ta != ta || tb != tb) continue;               /* 0 * inf: the ray runs inside this slab's plane */
        if (ta > t0) t0 = ta;
        if (tb < t1) t1 = tb;
    }
    return t0 <= t1 ? t0 : -1.0;
}

static int tri_hit(const bvh_t *b, int32_t f, const float o[3], const float d[3], float *t_out) {
    const int32_t *tv = b->faces + 3 * (int64_t)f;
    const float *v0 = b->verts + 3 * (int64_t)tv[0], *v1 = b->verts + 3 * (int64_t)tv[1], *v2 = b->verts + 3 * (int64_t)tv[2];
    const float e1x = v1[0] - v0[0], e1y = v1[1] - v0[1], e1z = v1[2] - v0[2];
    const float e2x = v2[0] - v0[0], e2y = v2[1] - v0[1], e2z = v2[2] - v0[2];
    const float px = d[1] * e2z - d[2] * e2y, py = d[2] * e2x - d[0] * e2z, pz = d[0] * e2y - d[1] * e2x;
    const float det = (e1x * px + e1y * py) + e1z * pz;
    const float inv = 1.0f / det;
    const float sx = o[0] - v0[0], sy = o[1] - v0[1], sz = o[2] - v0[2];
    const float u = ((sx * px + sy * py) + sz * pz) * inv;
    const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
    const float v = ((d[0] * qx + d[1] * qy) + d[2] * qz) * inv;
    const float t = ((e2x * qx + e2y * qy) + e2z * qz) * inv;
    *t_out = t;
    if (!((u >= 0.0f) & (v >= 0.0f) & (u + v <= 1.0f) & (t > 0.0f))) return 0;
    /* the hit-point test of tracer.c (hit_point.h: same operations, same margin) */
    return oracle_hit_point_in_box(o[0], o[1], o[2], d[0], d[1], d[2], t, v0[0], v0[1], v0[2], e1x, e1y, e1z, e2x, e2y, e2z, b->margin);
}

int oracle_trace_closest_bvh(const int32_t *faces, int64_t n_faces, const float *verts, int64_t n_verts,
                             const float *rays, int64_t n_rays, float *T, int32_t *ID) {
    bvh_t b;
    memset(&b, 0, sizeof b);
    b.faces = faces; b.verts = verts;
    if (n_faces > 0) {
        float lo = INFINITY, hi = -INFINITY;
        for (int64_t i = 0; i < 3 * n_verts; ++i) { if (verts[i] < lo) lo = verts[i]; if (verts[i] > hi) hi = verts[i]; }
        b.pad = 1e-4f * (hi - lo) + 1e-30f;                 /* >= 1.6 x the margin below: an accepted hit point is inside its leaf's box */
        b.margin = oracle_hit_margin(verts, n_verts);
        b.order = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_faces);
        b.nodes = (node_t *)malloc(sizeof(node_t) * (size_t)(2 * n_faces));
        if (!b.order || !b.nodes) { free(b.order); free(b.nodes); return -1; }
        for (int64_t i = 0; i < n_faces; ++i) b.order[i] = (int32_t)i;
        build(&b, 0, (int32_t)n_faces);
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < n_rays; ++r) {
        const float *o = rays + 6 * r, *d = o + 3;
        float best_t = INFINITY;
        int32_t best_f = -1;
        if (n_faces > 0) {
            const double od[3] = {o[0], o[1], o[2]};
            const double inv[3] = {1.0 / (double)d[0], 1.0 / (double)d[1], 1.0 / (double)d[2]};
            int32_t stack[128];
            int sp = 0;
            stack[sp++] = 0;
            while (sp) {
                const node_t *n = b.nodes + stack[--sp];
                /* the bound grows a little with best_t's rounding: ties and near-ties are kept */
                if (slab(n, od, inv, best_t == INFINITY ? INFINITY : (double)best_t * (1.0 + 1e-6) + 1e-30) < 0.0) continue;
                if (n->left < 0) {
                    for (int32_t k = n->first; k < n->first + n->count; ++k) {
                        const int32_t f = b.order[k];
                        float t;
                        if (tri_hit(&b, f, o, d, &t) && (t < best_t || (t == best_t && f < best_f))) { best_t = t; best_f = f; }
                    }
                } else {
                    const double tl = slab(b.nodes + n->left, od, inv, INFINITY), tr = slab(b.nodes + n->right, od, inv, INFINITY);
                    if (tl >= 0.0 && tr >= 0.0) {
                        if (tl < tr) { stack[sp++] = n->right; stack[sp++] = n->left; } else { stack[sp++] = n->left; stack[sp++] = n->right; }
                    } else if (tl >= 0.0) stack[sp++] = n->left;
                    else if (tr >= 0.0) stack[sp++] = n->right;
                }
            }
        }
        T[r] = best_f < 0 ? -1.0f : best_t;
        ID[r] = best_f;
    }
    free(b.order); free(b.nodes);
    return 0;
}

/* thread count of the following OpenMP regions of this library (both tracers) */
void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
