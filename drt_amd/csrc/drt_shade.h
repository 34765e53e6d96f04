// drt_shade.h -- float64 per-hit math of the refraction path and its hand-derived
// reverse mode.
//
// Forward follows the reference operation by operation (one rounding each, no
// contraction) so results agree with its float64 PyTorch graph to rounding:
//   bounce_forward  = JIT_Dintersect (reference DiffRender.py:64-121: t and the unit
//                     geometric normal; u, v are never consumed downstream)
//                   + Scene.refract_ray (DiffRender.py:503-535: orientation, eta swap,
//                     normal flip, FrDielectric's TIR flag DiffRender.py:51-61, the
//                     reference's Refract DiffRender.py:35-49 -- which is NOT Snell's
//                     law: cosThetaT = sqrt(1 - sin2ThetaI), kept as is -- and the
//                     1e-5 origin offset DiffRender.py:528-532).
// bounce_backward is the adjoint of exactly that graph (what autograd derives for the
// reference), including torch's conventions for clamp (gradient passes where the
// input is inside the closed range) and sqrt / division (no guards: a grazing hit
// yields inf/NaN exactly as the reference does; its limit_hook zeroes NaN later).
#pragma once
#include "drt_common.h"

namespace drt {

struct Bounce {
    // outputs
    d3 new_o, wt;
    bool tir;
    // tape
    d3 d, e1, e2, p, s, q, n0, n, w;
    double inv, t, e2q, len, sg, eta, ci, one_m_ci2, s2, ct, k, wl;
};

DRT_HD void bounce_forward(d3 o, d3 d, d3 v0, d3 v1, d3 v2, double ior_ext, double ior_int, Bounce& b) {
    b.d = d;
    b.e1 = v1 - v0;
    b.e2 = v2 - v0;
    b.p = cross(d, b.e2);
    const double det = dot(b.e1, b.p);
    b.inv = 1.0 / det;
    b.s = o - v0;
    b.q = cross(b.s, b.e1);
    b.e2q = dot(b.e2, b.q);
    b.t = b.e2q * b.inv;
    const d3 m = cross(b.e1, b.e2);
    b.len = sqrt((m.x * m.x + m.y * m.y) + m.z * m.z);
    b.n0 = m / b.len;
    const d3 wo = -d;
    double c = dot(wo, b.n0);
    c = c < -1.0 ? -1.0 : (c > 1.0 ? 1.0 : c);
    const bool entering = c > 0.0;
    double eta_i = ior_ext, eta_t = ior_int;
    b.sg = 1.0;
    if (!entering) {
        eta_i = ior_int; eta_t = ior_ext;
        b.sg = -1.0;
        c = -c;
    }
    b.n = b.sg * b.n0;
    // FrDielectric: only the total-internal-reflection flag survives
    double x = 1.0 - c * c;
    x = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x);
    const double sin_t = sqrt(x) * eta_i / eta_t;
    b.tir = sin_t >= 1.0;
    // Refract
    b.eta = eta_i / eta_t;
    b.ci = dot(b.n, wo);
    b.one_m_ci2 = 1.0 - b.ci * b.ci;
    b.s2 = b.one_m_ci2 < 0.0 ? 0.0 : b.one_m_ci2;
    const double s2c = b.s2 > 1.0 ? 1.0 : b.s2;
    b.ct = sqrt(1.0 - s2c);
    b.k = b.eta * b.ci - b.ct;
    b.w = b.eta * d + b.k * b.n;            // eta * -wo == eta * d
    b.wl = sqrt((b.w.x * b.w.x + b.w.y * b.w.y) + b.w.z * b.w.z);
    b.wt = b.w / b.wl;
    b.new_o = (o + b.t * d) + 1e-5 * b.wt;
}

// Adjoint of bounce_forward.  In: g_new_o, g_wt.  Out: gradients of the three
// vertices (accumulated into gv0/gv1/gv2) and of the incoming ray (g_o, g_d; set).
DRT_HD void bounce_backward(const Bounce& b, d3 g_new_o, d3 g_wt, d3& gv0, d3& gv1, d3& gv2, d3& g_o, d3& g_d) {
    // new_o = o + t*d + 1e-5*wt
    g_o = g_new_o;
    const double g_t = dot(g_new_o, b.d);
    g_d = b.t * g_new_o;
    const d3 G = g_wt + 1e-5 * g_new_o;
    // wt = w / |w|
    const d3 g_w = (G - dot(b.wt, G) * b.wt) / b.wl;
    // w = eta*d + k*n
    g_d += b.eta * g_w;
    const double g_k = dot(g_w, b.n);
    d3 g_n = b.k * g_w;
    // k = eta*ci - ct ; ct = sqrt(1 - min(s2,1)) ; s2 = max(1 - ci^2, 0)
    double g_ci = b.eta * g_k;
    const double g_ct = -g_k;
    const double g_x = g_ct / (2.0 * b.ct);
    const double g_s2 = (b.s2 <= 1.0) ? -g_x : 0.0;
    const double g_in = (b.one_m_ci2 >= 0.0) ? g_s2 : 0.0;
    g_ci += -2.0 * b.ci * g_in;
    // ci = n . wo = -(n . d)
    g_n += (-g_ci) * b.d;
    g_d += (-g_ci) * b.n;
    // n = sg*n0 ; n0 = m/|m| ; m = e1 x e2
    const d3 g_n0 = b.sg * g_n;
    const d3 g_m = (g_n0 - dot(b.n0, g_n0) * b.n0) / b.len;
    d3 g_e1 = cross(b.e2, g_m);
    d3 g_e2 = cross(g_m, b.e1);
    // t = (e2 . q) * inv
    g_e2 += (g_t * b.inv) * b.q;
    const d3 g_q = (g_t * b.inv) * b.e2;
    const double g_inv = g_t * b.e2q;
    // q = s x e1
    const d3 g_s = cross(b.e1, g_q);
    g_e1 += cross(g_q, b.s);
    // s = o - v0
    g_o += g_s;
    // inv = 1/det ; det = e1 . p ; p = d x e2
    const double g_det = -g_inv * b.inv * b.inv;
    g_e1 += g_det * b.p;
    const d3 g_p = g_det * b.e1;
    g_d += cross(b.e2, g_p);
    g_e2 += cross(g_p, b.d);
    gv1 += g_e1;
    gv2 += g_e2;
    gv0 -= (g_s + g_e1) + g_e2;
}

// ray_loss term of one completed path (reference optim.py:100-106):
//   target = normalize(screen_pixel - out_ori.detach()); diff = out_dir - target; loss += |diff|^2
// Returns the term and d loss / d out_dir (no gradient reaches out_ori: it is detached).
DRT_HD double ray_loss_term(d3 out_ori, d3 out_dir, d3 screen_pixel, d3& g_out_dir) {
    const d3 tv = screen_pixel - out_ori;
    const double tl = sqrt((tv.x * tv.x + tv.y * tv.y) + tv.z * tv.z);
    const d3 target = tv / tl;
    const d3 diff = out_dir - target;
    g_out_dir = 2.0 * diff;
    return (diff.x * diff.x + diff.y * diff.y) + diff.z * diff.z;
}

}  // namespace drt
