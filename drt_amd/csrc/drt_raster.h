// drt_raster.h -- primary visibility of pinhole ray grids by projection instead of BVH traversal.
//
// The first traversal of Scene.render_transparent (reference DiffRender.py:537-540: `Dintersect` of the camera rays)
// answers, for every pixel of a view, "closest triangle along this ray".  When the rays of an image ARE a pinhole grid
// -- generate_ray, reference captured_data.py:23-40: origin = R^-1[:3,3] for every pixel, direction ~ M (x, y, 1)^T with
// M = R^-1[:3,:3] K^-1 -- the set of pixels whose ray can hit a triangle is the projection of that triangle, so each
// triangle is tested only against the handful of pixels of its projected bounding box, with the SAME float32
// ray/triangle test (drt_tri.h) on the SAME float32 ray the traversal would use, and the closest hit per pixel is kept
// by a 64-bit atomic minimum of (t bits, face id) -- i.e. the contract of oracle/tracer.c (minimum t, ties -> lowest
// face id) without walking a tree: ~3 tests per (view, triangle) instead of ~14 node visits per candidate ray.
//
// A caller that presents the SAME, unchanged ray tensors again (a capture's views are constants of the optimisation) can
// keep the verdict: a call in DRT_GRID_ESTABLISH mode records the fitted models and, per image, whether every single ray
// verified, in a caller-owned cache; later calls in DRT_GRID_TRUST mode then neither re-fit nor re-read the rays of
// pixels no projected triangle touches (nine out of ten): k_cull writes their zeros without loading anything.
//
// Nothing is assumed about the caller's rays: the model (origin, M^-1) of every image is FITTED on the device from four
// of its rays, every single ray is then VERIFIED against it in k_cull (origin bit-equal, pixel position within
// kRasterVerifyTol), and a ray that does not verify -- calibrated per-pixel rays of a real capture, arbitrary ray
// lists -- takes the BVH path as before.  Conservative by construction: the box is grown by kRasterPad = 1/16 pixel.  What
// it has to cover, in pixels of a 1024-wide image of a ~500 mm distant object: the float32 projection of the vertices
// (~2e-4), the distance a verified ray may be from its pixel centre (kRasterVerifyTol = 1e-3) and how far outside a
// triangle the float32 test can still accept a ray (rounding of dot(s, p) against the pixel pitch, ~3e-4) -- together
// ~1.5e-3 px, ~3e-3 px at 4096 wide: a factor 20-40 below the pad.  (It was 1/4 px at first: the boxes of 2-3 pixel
// triangles then held a third more pixel centres, all of them misses.)
#pragma once
#include "drt_common.h"
#include "drt_tri.h"

namespace drt {

struct ViewModel {
    double o[3];        // common origin of the image's rays
    double minv[9];     // (x, y, 1) ~ minv * direction, row-major
    int32_t ok;         // 1: pinhole grid; 0: take the BVH path for every ray of this image
    int32_t all;        // 1: EVERY ray of the image verified against the model (established by a call in DRT_GRID_ESTABLISH mode)
};

static_assert(sizeof(ViewModel) == 104, "DRT_GRID_CACHE_BYTES of include/drt_hip.h");

constexpr double kRasterPad = 0.0625;         // pixels added on every side of a projected triangle's bounding box
constexpr double kRasterVerifyTol = 1e-3;    // a ray belongs to the grid if its direction projects within this of its pixel
#ifndef DRT_RASTER_MAX_PER_LANE
#define DRT_RASTER_MAX_PER_LANE 48
#endif
constexpr int kRasterMaxPerLane = DRT_RASTER_MAX_PER_LANE;        // larger boxes are handed to k_raster_big (one wave per band of rows of the box)
constexpr int kRasterBandPixels = 1024;      // pixel centres per band (a band is at least one row)

// Solve the model from the rays of the four image corners (unit or not, only their directions matter):
// d(x, y) ~ x m0 + y m1 + m2.  Returns false when the corner directions are degenerate.
DRT_HD bool fit_view_model(d3 o, d3 d00, d3 dW0, d3 d0H, d3 dWH, double wm1, double hm1, ViewModel& vm) {
    // beta dW0 + gamma d0H - eps dWH = d00   (from (W-1) m0 = beta dW0 - d00, (H-1) m1 = gamma d0H - d00, m2 = d00)
    const d3 c0 = dW0, c1 = d0H, c2 = -dWH;
    const double det = dot(c0, cross(c1, c2));
    vm.ok = 0; vm.all = 0;
    vm.o[0] = o.x; vm.o[1] = o.y; vm.o[2] = o.z;
    for (int k = 0; k < 9; ++k) vm.minv[k] = 0.0;
    if (!(fabs(det) > 1e-300) || !(wm1 > 0.0) || !(hm1 > 0.0)) return false;
    const double beta = dot(d00, cross(c1, c2)) / det, gamma = dot(c0, cross(d00, c2)) / det;
    const d3 m0 = (beta * dW0 - d00) / wm1, m1 = (gamma * d0H - d00) / hm1, m2 = d00;
    // inverse of M = [m0 m1 m2] (columns): rows of M^-1 are cross products / det(M)
    const double dm = dot(m0, cross(m1, m2));
    if (!(fabs(dm) > 1e-300)) return false;
    const d3 r0 = cross(m1, m2) / dm, r1 = cross(m2, m0) / dm, r2 = cross(m0, m1) / dm;
    vm.minv[0] = r0.x; vm.minv[1] = r0.y; vm.minv[2] = r0.z;
    vm.minv[3] = r1.x; vm.minv[4] = r1.y; vm.minv[5] = r1.z;
    vm.minv[6] = r2.x; vm.minv[7] = r2.y; vm.minv[8] = r2.z;
    bool fin = true;
    for (int k = 0; k < 9; ++k) fin = fin && (fabs(vm.minv[k]) < 1e300);
    return fin;
}

// (px, py, pz) = minv * v: the pixel position is (px / pz, py / pz), pz > 0 in front of the camera.
DRT_HD d3 view_project(const ViewModel& vm, d3 v) {
    return d3{(vm.minv[0] * v.x + vm.minv[1] * v.y) + vm.minv[2] * v.z,
              (vm.minv[3] * v.x + vm.minv[4] * v.y) + vm.minv[5] * v.z,
              (vm.minv[6] * v.x + vm.minv[7] * v.y) + vm.minv[8] * v.z};
}

// Is (o, d) the ray of pixel (x, y) of this image?
DRT_HD bool view_verify(const ViewModel& vm, d3 o, d3 d, double x, double y) {
    if (!(o.x == vm.o[0] && o.y == vm.o[1] && o.z == vm.o[2])) return false;
    const d3 p = view_project(vm, d);
    if (!(p.z > 0.0)) return false;
    return fabs(p.x - x * p.z) <= kRasterVerifyTol * p.z && fabs(p.y - y * p.z) <= kRasterVerifyTol * p.z;
}

struct PixelBox {
    int x0, x1, y0, y1;     // inclusive; empty when x0 > x1 or y0 > y1
    bool unsafe;            // a vertex at or behind the camera plane (or a non-finite projection): no projection bound
};

// Padded pixel bounding box of the triangle record `t` seen from the image's origin, in float32 (|error| ~1e-4 px at these
// magnitudes; the pad is 1/16 px).  Used by k_raster and by the host-side unit test alike.
DRT_HD PixelBox project_tri_box(const ViewModel& vm, f3 o32, const TriRec& t, int w, int h) {
    PixelBox r{0, -1, 0, -1, false};
    const f3 a{t.v0x - o32.x, t.v0y - o32.y, t.v0z - o32.z};
    const f3 b{a.x + t.e1x, a.y + t.e1y, a.z + t.e1z}, c{a.x + t.e2x, a.y + t.e2y, a.z + t.e2z};
    const float m0 = (float)vm.minv[0], m1 = (float)vm.minv[1], m2 = (float)vm.minv[2], m3 = (float)vm.minv[3], m4 = (float)vm.minv[4],
                m5 = (float)vm.minv[5], m6 = (float)vm.minv[6], m7 = (float)vm.minv[7], m8 = (float)vm.minv[8];
    const float az = fmaf(m6, a.x, fmaf(m7, a.y, m8 * a.z)), bz = fmaf(m6, b.x, fmaf(m7, b.y, m8 * b.z)), cz = fmaf(m6, c.x, fmaf(m7, c.y, m8 * c.z));
    if (!(fminf(az, fminf(bz, cz)) > 1e-20f)) { r.unsafe = true; return r; }      // at or behind the camera plane
    const float ra = 1.0f / az, rb = 1.0f / bz, rc = 1.0f / cz;
    const float ax = fmaf(m0, a.x, fmaf(m1, a.y, m2 * a.z)) * ra, ay = fmaf(m3, a.x, fmaf(m4, a.y, m5 * a.z)) * ra;
    const float bx = fmaf(m0, b.x, fmaf(m1, b.y, m2 * b.z)) * rb, by = fmaf(m3, b.x, fmaf(m4, b.y, m5 * b.z)) * rb;
    const float cx = fmaf(m0, c.x, fmaf(m1, c.y, m2 * c.z)) * rc, cy = fmaf(m3, c.x, fmaf(m4, c.y, m5 * c.z)) * rc;
    const float pad = (float)kRasterPad;
    const float lox = fminf(ax, fminf(bx, cx)) - pad, hix = fmaxf(ax, fmaxf(bx, cx)) + pad;
    const float loy = fminf(ay, fminf(by, cy)) - pad, hiy = fmaxf(ay, fmaxf(by, cy)) + pad;
    if (!(fabsf(lox) < 1e9f && fabsf(hix) < 1e9f && fabsf(loy) < 1e9f && fabsf(hiy) < 1e9f)) { r.unsafe = true; return r; }
    const float fx0 = fmaxf(ceilf(lox), 0.0f), fx1 = fminf(floorf(hix), (float)(w - 1));
    const float fy0 = fmaxf(ceilf(loy), 0.0f), fy1 = fminf(floorf(hiy), (float)(h - 1));
    if (fx0 > fx1 || fy0 > fy1) return r;
    r.x0 = (int)fx0; r.x1 = (int)fx1; r.y0 = (int)fy0; r.y1 = (int)fy1;
    return r;
}

// Camera-facing?  (k_raster handles these in its first launch, the others in the second: see there.)
DRT_HD bool tri_faces_away(f3 o32, const TriRec& t) {
    const f3 a{t.v0x - o32.x, t.v0y - o32.y, t.v0z - o32.z};
    return dot(a, cross(f3{t.e1x, t.e1y, t.e1z}, f3{t.e2x, t.e2y, t.e2z})) > 0.0f;
}

// 64-bit key of a hit: t > 0, so its bit pattern orders like an unsigned integer; equal t -> lowest face id wins.
DRT_HD unsigned long long raster_key(float t, int32_t face) {
    uint32_t tb;
    memcpy(&tb, &t, 4);
    return ((unsigned long long)tb << 32) | (unsigned long long)(uint32_t)face;
}
constexpr unsigned long long kRasterEmpty = ~0ull;

// Where the key of the ray in column x of (sub-batch-global) row gy lives: 4 x 4-pixel tiles, 16 keys = one 128-byte cache line each.
// The chip serves memory requests -- and atomics above all -- per cache line, not per lane (tools/ubench/atomic_pattern.hip); the pixels a
// triangle of one to three pixels touches, and those of its neighbours in the wave, share a tile far more often than a 16-pixel run of
// one image row.  A bijection of [0, rows * w) whenever w and the row count are multiples of 4 (whole 64 x 4 patches: the
// precondition of the projection pass).
DRT_HD int64_t raster_slot(unsigned x, unsigned gy, unsigned w) {
    return (((int64_t)(gy >> 2) * (w >> 2) + (x >> 2)) << 4) | ((gy & 3u) << 2) | (x & 3u);
}

}  // namespace drt
