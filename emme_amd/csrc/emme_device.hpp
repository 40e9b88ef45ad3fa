// emme_device.hpp -- device-side building blocks of the dispersion-matrix fill on gfx950.
//
// What is computed (per matrix entry i<j and velocity moment m): kappa_m(eta_i, eta_j, omega),
// the adaptive Gauss-Kronrod integral over the rotated time contour of the gyrokinetic
// integrand -- reference src/Parameters.cpp:113-184 (integrand :120-176), quadrature
// include/functions.h:181-251,305-331, Bessel helper include/functions.h:381-408.
//
// How it is laid out for CDNA4 (not a translation of the reference's control flow):
//   * one GK rule = one lane group: the 15 (31) nodes of an interval are evaluated by 16
//     (32) adjacent lanes at once, so a wave64 carries 4 (2) integrals; Kronrod/Gauss sums
//     are butterfly all-reduces inside the group and the accept/split decision is
//     group-uniform;
//   * the bisection tree is walked without a stack: (depth, index) is advanced by
//     increment + strip-trailing-zeros, and [l,r] is rebuilt from the path bits with the
//     same (l+r)/2 sequence as the reference, so every abscissa is bit-identical;
//   * everything that depends only on the pair (beta_1, sqrt(b b'), W dx ...) or only on
//     eta (g, b tables in LDS) is hoisted out of the integrand, and algebraic identities
//     remove all but four divisions per evaluation (see integrand()).
#pragma once
#include <hip/hip_runtime.h>

namespace emme {

struct cd {
    double x, y;
};
__device__ __forceinline__ cd mk(double x, double y) { return cd{x, y}; }
__device__ __forceinline__ cd operator+(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd operator-(cd a) { return cd{-a.x, -a.y}; }
__device__ __forceinline__ cd operator*(cd a, cd b) {
    return cd{fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x)};
}
__device__ __forceinline__ cd operator*(double s, cd a) { return cd{s * a.x, s * a.y}; }
__device__ __forceinline__ cd operator*(cd a, double s) { return cd{s * a.x, s * a.y}; }
__device__ __forceinline__ double norm2(cd a) { return fma(a.x, a.x, a.y * a.y); }
__device__ __forceinline__ cd conj(cd a) { return cd{a.x, -a.y}; }
__device__ __forceinline__ cd rcp(cd a) {
    const double d = 1.0 / norm2(a);
    return cd{a.x * d, -(a.y * d)};
}
// multiply by i
__device__ __forceinline__ cd times_i(cd a) { return cd{-a.y, a.x}; }

// Scalars shared by every work item of a launch (passed by value as a kernel argument).
struct DevParams {
    int N;        // grid points
    int dim;      // N (electrostatic, beta_e == 0) or 2N
    int nm;       // integrals per pair: 1 (ES) or 3 (EM)
    int max_sub;  // integration_iteration_limit
    double dx;
    double inv_arc;    // 1 / arc_coeff
    double qR;         // q * R
    double vt;
    double cb;         // (qR / vt) * omega_d_bar            : beta_1   = cb  * (g_i - g_j)
    double cbe;        // (qR / vt) * (omega_d_bar ws_e/ws_i): beta_1_e = cbe * (g_i - g_j)
    double omega_s_i, omega_s_e, eta_i, eta_e, tau;
    double rel_tol;    // integration_precision -> global_rel_tol
    double prec_goal;  // integration_accuracy  -> precision_goal
    double pref;       // qR / (vt sqrt(2 pi)); kappa = -i * pref * integral
    double diag_a;     // 1 + 1/tau
    double diag_d;     // 2 tau / beta_e (EM only)
};

// Per-pair invariants (reference recomputes these inside every integrand call).
struct PairConst {
    double de;     // eta_i - eta_j  (< 0)
    double beta1;  // src/Parameters.cpp:87-90
    double s;      // sqrt(b_i b_j)
    double inv_s;
    double bsum;   // b_i + b_j
    double c_lam;  // lambda = 1 + i c_lam t~   (src/Parameters.cpp:101-106)
    double c_nv;   // norm_vel = c_nv / t~      (src/Parameters.cpp:142)
};

struct OmegaConst {
    cd omega;
    double omi;  // -copysign(1, Re omega), src/Parameters.cpp:121
};

// Miller backward recurrence for the unnormalised I0, I1 of complex z = s / lambda
// (include/functions.h:381-408).  `w` = lambda / s, so the reference's `2n / z * p`
// becomes (2n) * (w * p) with no division.  Returns y0, y1 and mu + y0.
__device__ __forceinline__ void bessel_miller(cd w, double zabs, bool re_z_neg, cd& y0, cd& y1,
                                              cd& mutot) {
    int n = (int)floor(zabs) + 1;
    cd p0 = mk(0.0, 0.0), p1 = mk(1.0, 0.0);
    // threshold evaluated once with p1 = 1: |p0 - 2n/z p1| = 2n/|z|  (functions.h:388-390)
    double test = fmax(sqrt(2.e+7 * (2.0 * n / zabs)), 2.e+7);
    const double test2 = test * test;
    int guard = 0;
    while (norm2(p1) <= test2 && guard < 4096) {  // guard: every wave must terminate
        const cd t = p0 - (2.0 * n) * (w * p1);
        p0 = p1;
        p1 = t;
        ++n;
        ++guard;
    }
    y0 = rcp(p1);
    y1 = mk(0.0, 0.0);
    cd mu = mk(0.0, 0.0);
    for (n--; n > 0; --n) {
        const cd t = (2.0 * n) * (w * y0) + y1;
        y1 = y0;
        y0 = t;
        const double sg = re_z_neg ? (double)(2 - 4 * (n & 1)) : 2.0;  // 2*(1-2(n&1)) or 2
        mu = mu + sg * y1;
    }
    mutot = mu + y0;
}

// F_m(tan x) / cos^2 x for one quadrature abscissa x in (0, pi/2).
//
// Restatement of src/Parameters.cpp:120-176 with
//   e      = exp(-i omi atan u) = (1 - i omi u) / sqrt(1 + u^2),   u = t / arc
//   t~     = t e,   1/t~ = conj(e)/t   (|e| = 1)
//   jacob  = e (1 - i omi u / (1 + u^2))
//   lambda = 1 + i c_lam t~;  2 + i beta_1/norm_vel = 2 lambda  (same c_lam)
//   1/t~ * jacob = (1 - i omi u/(1+u^2)) / t
// and the underflow clamp of :167-173 tested before the Bessel recurrence (its argument
// needs only z = s/lambda), which skips the recurrence and the complex exp where the
// integrand is exactly zero.
__device__ __forceinline__ cd integrand(double x, const DevParams& P, const PairConst& pc,
                                        const OmegaConst& oc, int m) {
    double sx, cx;
    sincos(x, &sx, &cx);
    const double inv_cx = 1.0 / cx;
    const double t = sx * inv_cx;
    const double inv_c2 = inv_cx * inv_cx;
    const double inv_t = cx / sx;

    const double u = t * P.inv_arc;
    const double r2 = 1.0 / fma(u, u, 1.0);
    const double r1 = sqrt(r2);
    const double ou = oc.omi * u;
    const cd e = mk(r1, -(ou * r1));
    const cd taut = t * e;
    const cd jt = mk(inv_t, -(ou * r2 * inv_t));  // jacob / t~

    const cd lam = mk(fma(-pc.c_lam, taut.y, 1.0), pc.c_lam * taut.x);
    const double rl2 = 1.0 / norm2(lam);
    const cd rl = mk(lam.x * rl2, -(lam.y * rl2));  // 1 / lambda
    const cd z = pc.s * rl;

    const cd nv = (pc.c_nv * inv_t) * conj(e);
    const cd nv2 = nv * nv;

    // log_coef, src/Parameters.cpp:154-165
    cd L = (-0.5) * nv2 + times_i(taut * oc.omega);
    L = L + (-0.5 * pc.beta1) * times_i(nv);
    L = L - (0.5 * pc.bsum) * rl;
    const bool zneg = z.x < 0.0;
    const cd arg = zneg ? (L - z) : (L + z);  // log_coef - (Re z<0 ? z : -z)
    if (!(arg.x >= -40.)) {
        // safe_exp clamp (:167-173); NaN also lands here and is propagated below
        if (arg.x < -40.) return mk(0.0, 0.0);
    }
    double sa, ca;
    sincos(arg.y, &sa, &ca);
    const double ea = exp(arg.x);
    const cd sexp = mk(ea * ca, ea * sa);

    cd y0, y1, mutot;
    bessel_miller(pc.inv_s * lam, pc.s * sqrt(rl2), zneg, y0, y1, mutot);

    const cd rl3 = rl * rl * rl;  // lambda^-3 (reference: pow(lambda, -3.) via log/polar)
    const double wsi_eta = P.omega_s_i * P.eta_i;
    cd a0 = oc.omega - P.omega_s_i * mk(fma(P.eta_i, fma(0.5, nv2.x, -1.5), 1.0),
                                         P.eta_i * 0.5 * nv2.y);
    cd i0c = a0 * rl + (wsi_eta * mk(0.5 * pc.bsum - lam.x, -lam.y)) * rl3;
    cd i1c = (-wsi_eta * pc.s) * rl3;

    cd F = jt * sexp;
    if (m == 1)
        F = F * nv;
    else if (m == 2)
        F = F * nv2;
    F = F * (i0c * y0 + i1c * y1);
    F = F * rcp(mutot);
    return inv_c2 * F;
}

// Gauss-Kronrod node tables laid out per lane of a group (centre, +x_1..+x_h, -x_1..-x_h,
// pad).  Values: include/functions.h:93-120 (15 points) and :126-161 (31 points).
struct GkLane {
    double x, wk, wg;
};

}  // namespace emme
