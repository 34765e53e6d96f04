// drt_edge.h -- per-edge math of the silhouette and smoothness branches (float64), forward
// and hand-derived adjoints.  Plain C++ (also compiled by tests/hostsim).
//
//   edge_face_normals / dihedral_cos   <- edge_face_norm + Scene.dihedral_angle
//                                         (reference DiffRender.py:149-163, 440-443)
//   silhouette_flag                    <- Scene.silhouette_edge (DiffRender.py:445-457)
//   project_endpoint / edge_sample     <- Scene.primary_visibility + primary_edge_sample.forward
//                                         (DiffRender.py:459-479, 189-258)
//   project_endpoint_backward          <- what autograd derives for DiffRender.py:466-474 (depth
//                                         row detached when detach_depth) chained with the custom
//                                         backward of primary_edge_sample (DiffRender.py:263-267)
#pragma once
#include "drt_common.h"

namespace drt {

struct Camera {          // row-major copies of camera_M = (R 4x4, K 3x3, R^-1 4x4, K^-1 3x3)
    double R[16], K[9], Rinv[16], Kinv[9];
};

struct FaceNormal {
    d3 e1, e2, m, n;
    double len;
};

DRT_HD void face_normal(d3 v0, d3 v1, d3 v2, FaceNormal& f) {
    f.e1 = v1 - v0;
    f.e2 = v2 - v0;
    f.m = cross(f.e1, f.e2);
    f.len = sqrt((f.m.x * f.m.x + f.m.y * f.m.y) + f.m.z * f.m.z);
    f.n = f.m / f.len;
}

// adjoint of n = normalize((v1 - v0) x (v2 - v0)); accumulates into g0/g1/g2
DRT_HD void face_normal_backward(const FaceNormal& f, d3 g_n, d3& g0, d3& g1, d3& g2) {
    const d3 g_m = (g_n - dot(f.n, g_n) * f.n) / f.len;
    const d3 g_e1 = cross(f.e2, g_m);
    const d3 g_e2 = cross(g_m, f.e1);
    g1 += g_e1;
    g2 += g_e2;
    g0 -= g_e1 + g_e2;
}

// xor of the facing signs of the two faces of an edge, seen from `origin`
DRT_HD bool silhouette_flag(const FaceNormal& a, d3 a_v0, const FaceNormal& b, d3 b_v0, d3 origin) {
    const double d1 = dot(a.n, origin - a_v0);
    const double d2 = dot(b.n, origin - b_v0);
    return (d1 > 0.0) != (d2 > 0.0);
}

struct Projected {
    double vc0, vc1, vc2;   // camera-space point (rows 0..2 of R [V;1])
    double kc0, kc1, kc2;   // K * vc
    double px, py;
};

DRT_HD void project_endpoint(const Camera& c, d3 v, Projected& p) {
    // R @ [V;1]: ((r0*x + r1*y) + r2*z) + r3 -- the accumulation order of a dense 4-term dot product
    p.vc0 = ((c.R[0] * v.x + c.R[1] * v.y) + c.R[2] * v.z) + c.R[3];
    p.vc1 = ((c.R[4] * v.x + c.R[5] * v.y) + c.R[6] * v.z) + c.R[7];
    p.vc2 = ((c.R[8] * v.x + c.R[9] * v.y) + c.R[10] * v.z) + c.R[11];
    p.kc0 = (c.K[0] * p.vc0 + c.K[1] * p.vc1) + c.K[2] * p.vc2;
    p.kc1 = (c.K[3] * p.vc0 + c.K[4] * p.vc1) + c.K[5] * p.vc2;
    p.kc2 = (c.K[6] * p.vc0 + c.K[7] * p.vc1) + c.K[8] * p.vc2;
    p.px = p.kc0 / p.kc2;
    p.py = p.kc1 / p.kc2;
}

// d(px,py)/dV times (g_px, g_py); the camera-space depth row carries no gradient when detach_depth
DRT_HD d3 project_endpoint_backward(const Camera& c, const Projected& p, double g_px, double g_py, bool detach_depth) {
    const double g_kc0 = g_px / p.kc2;
    const double g_kc1 = g_py / p.kc2;
    const double g_kc2 = -(g_px * p.kc0 + g_py * p.kc1) / (p.kc2 * p.kc2);
    const double g_vc0 = (g_kc0 * c.K[0] + g_kc1 * c.K[3]) + g_kc2 * c.K[6];
    const double g_vc1 = (g_kc0 * c.K[1] + g_kc1 * c.K[4]) + g_kc2 * c.K[7];
    const double g_vc2 = detach_depth ? 0.0 : (g_kc0 * c.K[2] + g_kc1 * c.K[5]) + g_kc2 * c.K[8];
    return d3{(g_vc0 * c.R[0] + g_vc1 * c.R[4]) + g_vc2 * c.R[8],
              (g_vc0 * c.R[1] + g_vc1 * c.R[5]) + g_vc2 * c.R[9],
              (g_vc0 * c.R[2] + g_vc1 * c.R[6]) + g_vc2 * c.R[10]};
}

struct EdgeSample {
    double midx, midy;      // sample point (edge midpoint in pixels)
    double nx, ny;          // un-normalised 2-D edge normal N = (ay - by, bx - ax)
    d3 dir_up, dir_lo;      // world-space probe directions (un-normalised), origin = camera centre
};

// Probe rays one pixel either side of the projected edge (DiffRender.py:202-222).
DRT_HD void edge_sample(const Camera& c, const Projected& a, const Projected& b, d3 origin, EdgeSample& s) {
    s.midx = (a.px + b.px) / 2.0;
    s.midy = (a.py + b.py) / 2.0;
    s.nx = a.py - b.py;
    s.ny = b.px - a.px;
    const double nl = sqrt(s.nx * s.nx + s.ny * s.ny);
    const double ux = s.nx / nl, uy = s.ny / nl;
    for (int k = 0; k < 2; ++k) {
        const double sg = k == 0 ? 1.0 : -1.0;
        const double fx = s.midx + sg * ux, fy = s.midy + sg * uy;
        // K^-1 @ [fx, fy, 1]
        const double cx = (c.Kinv[0] * fx + c.Kinv[1] * fy) + c.Kinv[2];
        const double cy = (c.Kinv[3] * fx + c.Kinv[4] * fy) + c.Kinv[5];
        const double cz = (c.Kinv[6] * fx + c.Kinv[7] * fy) + c.Kinv[8];
        // (R^-1 @ [cx, cy, cz, 1])[:3]
        const d3 w{((c.Rinv[0] * cx + c.Rinv[1] * cy) + c.Rinv[2] * cz) + c.Rinv[3],
                   ((c.Rinv[4] * cx + c.Rinv[5] * cy) + c.Rinv[6] * cz) + c.Rinv[7],
                   ((c.Rinv[8] * cx + c.Rinv[9] * cy) + c.Rinv[10] * cz) + c.Rinv[11]};
        if (k == 0) s.dir_up = w - origin; else s.dir_lo = w - origin;
    }
}

}  // namespace drt
