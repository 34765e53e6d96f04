/*
 * oracle/tracer.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the tracer contract behind the reference's native class
 * `optix_mesh` (reference optix_extend.cpp:29-57: closest hit for rays
 * f32 [N,6] -> T f32 [N], ID i32 [N]).  The reference delegates the arithmetic
 * to NVIDIA OptiX Prime 6.5.0 (closed source, pinned by reference README.md:10 and
 * config.py:3-4, absent from /root/reference and from this image), so the
 * ray/triangle arithmetic itself is PARITY UNPINNED against OptiX.  What is pinned
 * is the documented contract (RTP_QUERY_TYPE_CLOSEST, RAY_ORIGIN_DIRECTION i.e.
 * t in (0, inf), HIT_T_TRIID; the caller keeps `T > 0`, reference
 * DiffRender.py:390-392) and the algorithm BASELINE.json configs[0] names:
 * brute-force ray/triangle over every face.
 *
 * The per-triangle arithmetic is the float32 transcription of the reference's own
 * Moller-Trumbore (`JIT_Dintersect`, reference DiffRender.py:64-91), one rounding
 * per operation, no FMA contraction (build with -ffp-contract=off):
 *     e1 = v1 - v0, e2 = v2 - v0, p = d x e2, det = e1 . p, inv = 1 / det,
 *     s = o - v0, u = (s . p) * inv, q = s x e1, v = (d . q) * inv, t = (e2 . q) * inv
 *     dot(a,b) = (a0*b0 + a1*b1) + a2*b2 ; cross(a,b) = (a1*b2 - a2*b1, a2*b0 - a0*b2, a0*b1 - a1*b0)
 *     hit  <=>  u >= 0 && v >= 0 && u + v <= 1 && t > 0      (NaN/inf from det == 0 fail the tests)
 *               && the hit point, in the triangle's frame r = (o - v0) + t*d, lies in the box of (0, e1, e2) grown by
 *                  m = max(margin, 2^-18 * max|o - v0|)                                  (hit_point.h: the exact expression)
 *     margin = 0.5 * (largest extent of the box of ALL vertices) / 8192   (float32; the product's drt_tri.h / drt_lbvh.h state why:
 *     a ray inside a triangle's plane has det = rounding noise and can pass the first four tests anywhere along itself; the fifth
 *     keeps "closest hit over every face" something that does not depend on which faces an acceleration structure visits)
 * The four inequalities run in the vectorised block loop; the fifth is applied in the scalar reduction, only to a candidate that
 * would become the new closest hit (a candidate that fails it is skipped, exactly as if it had failed in the block loop).
 * Closest hit = minimum t; equal t -> lowest face id.  Miss -> T = -1, ID = -1.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "hit_point.h"

#define BLK 512

typedef struct {
    float *v0x, *v0y, *v0z, *e1x, *e1y, *e1z, *e2x, *e2y, *e2z;
} soa_t;

static int soa_alloc(soa_t *s, int64_t n) {
    float *base = (float *)malloc(sizeof(float) * 9 * (size_t)(n + BLK));
    if (!base) return -1;
    s->v0x = base;            s->v0y = base + (n + BLK) * 1; s->v0z = base + (n + BLK) * 2;
    s->e1x = base + (n + BLK) * 3; s->e1y = base + (n + BLK) * 4; s->e1z = base + (n + BLK) * 5;
    s->e2x = base + (n + BLK) * 6; s->e2y = base + (n + BLK) * 7; s->e2z = base + (n + BLK) * 8;
    return 0;
}

/* margin of the hit-point test: half of (largest extent of the box of all vertices) / 8192, in float32 */
float oracle_hit_margin(const float *verts, int64_t n_verts) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n_verts; ++i)
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], verts[3 * i + a]); hi[a] = fmaxf(hi[a], verts[3 * i + a]); }
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    return 0.5f * (fmaxf(ex, fmaxf(ey, ez)) * (1.0f / 8192.0f));
}

/* t of one ray against triangles [j0, j1) where the four inequalities hold, +inf elsewhere (the fifth condition: the caller). */
__attribute__((target_clones("avx2", "default")))
static void block_t(const soa_t *s, int64_t j0, int64_t j1,
                    float ox, float oy, float oz, float dx, float dy, float dz, float *tt) {
    for (int64_t j = j0; j < j1; ++j) {
        const float e1x = s->e1x[j], e1y = s->e1y[j], e1z = s->e1z[j];
        const float e2x = s->e2x[j], e2y = s->e2y[j], e2z = s->e2z[j];
        const float px = dy * e2z - dz * e2y;
        const float py = dz * e2x - dx * e2z;
        const float pz = dx * e2y - dy * e2x;
        const float det = (e1x * px + e1y * py) + e1z * pz;
        const float inv = 1.0f / det;
        const float sx = ox - s->v0x[j], sy = oy - s->v0y[j], sz = oz - s->v0z[j];
        const float u = ((sx * px + sy * py) + sz * pz) * inv;
        const float qx = sy * e1z - sz * e1y;
        const float qy = sz * e1x - sx * e1z;
        const float qz = sx * e1y - sy * e1x;
        const float v = ((dx * qx + dy * qy) + dz * qz) * inv;
        const float t = ((e2x * qx + e2y * qy) + e2z * qz) * inv;
        const int hit = (u >= 0.0f) & (v >= 0.0f) & (u + v <= 1.0f) & (t > 0.0f);
        tt[j - j0] = hit ? t : INFINITY;
    }
}

/*
 * faces i32 [F,3], verts f32 [V,3], rays f32 [N,6] (ox,oy,oz,dx,dy,dz) -> T f32 [N], ID i32 [N].
 * Returns 0, or -1 on allocation failure.
 */
static int trace_closest(const int32_t *faces, int64_t n_faces, const float *verts, int64_t n_verts,
                         const float *rays, int64_t n_rays, float *T, int32_t *ID, int fifth) {
    const float margin = oracle_hit_margin(verts, n_verts);
    soa_t s;
    if (soa_alloc(&s, n_faces) != 0) return -1;
    for (int64_t j = 0; j < n_faces; ++j) {
        const float *a = verts + 3 * (int64_t)faces[3 * j + 0];
        const float *b = verts + 3 * (int64_t)faces[3 * j + 1];
        const float *c = verts + 3 * (int64_t)faces[3 * j + 2];
        s.v0x[j] = a[0]; s.v0y[j] = a[1]; s.v0z[j] = a[2];
        s.e1x[j] = b[0] - a[0]; s.e1y[j] = b[1] - a[1]; s.e1z[j] = b[2] - a[2];
        s.e2x[j] = c[0] - a[0]; s.e2y[j] = c[1] - a[1]; s.e2z[j] = c[2] - a[2];
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n_rays; ++i) {
        const float *r = rays + 6 * i;
        float tt[BLK];
        float best = INFINITY;
        int32_t best_id = -1;
        for (int64_t j0 = 0; j0 < n_faces; j0 += BLK) {
            const int64_t j1 = j0 + BLK < n_faces ? j0 + BLK : n_faces;
            block_t(&s, j0, j1, r[0], r[1], r[2], r[3], r[4], r[5], tt);
            for (int64_t j = j0; j < j1; ++j) {
                if (tt[j - j0] < best &&
                    (!fifth || oracle_hit_point_in_box(r[0], r[1], r[2], r[3], r[4], r[5], tt[j - j0], s.v0x[j], s.v0y[j], s.v0z[j],
                                            s.e1x[j], s.e1y[j], s.e1z[j], s.e2x[j], s.e2y[j], s.e2z[j], margin))) {
                    best = tt[j - j0]; best_id = (int32_t)j;
                }
            }
        }
        T[i] = best_id >= 0 ? best : -1.0f;
        ID[i] = best_id;
    }
    free(s.v0x);
    return 0;
}

int oracle_trace_closest(const int32_t *faces, int64_t n_faces, const float *verts, int64_t n_verts,
                         const float *rays, int64_t n_rays, float *T, int32_t *ID) {
    return trace_closest(faces, n_faces, verts, n_verts, rays, n_rays, T, ID, 1);
}

/* The four inequalities alone, no hit-point condition: NOT the contract -- what the far-camera tests compare it with, to show that
 * the fifth condition keeps every well-conditioned hit of a ray that starts hundreds of extents away. */
int oracle_trace_closest_mt_only(const int32_t *faces, int64_t n_faces, const float *verts, int64_t n_verts,
                                 const float *rays, int64_t n_rays, float *T, int32_t *ID) {
    return trace_closest(faces, n_faces, verts, n_verts, rays, n_rays, T, ID, 0);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
