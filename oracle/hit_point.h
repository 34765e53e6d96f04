/*
 * oracle/hit_point.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The fifth condition of the tracer contract (tracer.c's header; the product states it in drt_amd/csrc/drt_tri.h::hit_point_in_box),
 * in the triangle's own frame and with one rounding per operation (build with -ffp-contract=off), so that both CPU tracers and the
 * device agree bit for bit:
 *     s = o - v0,  r = s + t d,  m = max(margin, 2^-18 * max|s_i|)
 *     r_i + m >= min(0, e1_i, e2_i)  &&  r_i - m <= max(0, e1_i, e2_i)   for i = x, y, z
 * `margin` = half the padding of the product's leaf boxes (2^-14 of the largest extent of the box of all vertices); the second
 * term lets the tolerance grow with the distance between the ray's origin and the triangle, because that is how the float32 error of
 * the reconstructed hit point grows (about 1e-6 of the ray's length for a well-conditioned t): it takes over from 16 extents on,
 * and a camera hundreds of extents away keeps its legitimate hits (tests/test_oracle_golden.py::test_far_camera_keeps_its_hits).
 * Follows the reference's contract at optix_extend.cpp:29-57 only in so far as OptiX documents none: see tracer.c.
 */
#ifndef ORACLE_HIT_POINT_H
#define ORACLE_HIT_POINT_H
#include <math.h>

static inline int oracle_hit_point_in_box(float ox, float oy, float oz, float dx, float dy, float dz, float t,
                                          float v0x, float v0y, float v0z, float e1x, float e1y, float e1z,
                                          float e2x, float e2y, float e2z, float margin) {
    const float sx = ox - v0x, sy = oy - v0y, sz = oz - v0z;
    const float m = fmaxf(margin, fmaxf(fabsf(sx), fmaxf(fabsf(sy), fabsf(sz))) * 0x1p-18f);
    const float rx = sx + t * dx, ry = sy + t * dy, rz = sz + t * dz;
    return (rx + m >= fminf(0.0f, fminf(e1x, e2x))) & (rx - m <= fmaxf(0.0f, fmaxf(e1x, e2x))) &
           (ry + m >= fminf(0.0f, fminf(e1y, e2y))) & (ry - m <= fmaxf(0.0f, fmaxf(e1y, e2y))) &
           (rz + m >= fminf(0.0f, fminf(e1z, e2z))) & (rz - m <= fmaxf(0.0f, fmaxf(e1z, e2z)));
}
#endif
