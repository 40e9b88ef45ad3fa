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
// multiply by i
__device__ __forceinline__ cd times_i(cd a) { return cd{-a.y, a.x}; }

// ---- cross-lane sums without LDS traffic (DPP) ------------------------------------------
// One step: v + (v of the lane selected by the DPP control).  Controls used: quad_perm
// [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140:
// after the four of them every lane of a 16-lane row holds the row's sum, and because
// floating-point addition is commutative every lane holds the SAME bits.
// (mov_dpp, not update_dpp(0, ..): every lane is written by these controls, and the "old" operand of the latter
// cost a v_mov_b32 0 per half)
template <int CTRL>
__device__ __forceinline__ double dpp_add_step(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)b, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
    return v + __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// the value of the lane selected by the DPP control (no sum)
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)b, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double row16_sum(double v) {
    v = dpp_add_step<0xB1>(v);
    v = dpp_add_step<0x4E>(v);
    v = dpp_add_step<0x141>(v);
    v = dpp_add_step<0x140>(v);
    return v;
}
// sum over all 64 lanes, in every lane (rows combined in a fixed order)
__device__ __forceinline__ double wave64_sum(double v) {
    v = row16_sum(v);
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long long b = __double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 16 * r);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 16 * r);
        s += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    return s;
}

// 1/x and 1/sqrt(x) for normal-range arguments: hardware seed (v_rcp_f64 / v_rsq_f64) +
// two Newton steps, i.e. the compiler's own division sequence without the range scaling
// and fix-up instructions that exist for subnormal/overflow operands (never met here).
__device__ __forceinline__ double frcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double frsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = fma(0.5 * y, fma(-(x * y), y, 1.0), y);
    y = fma(0.5 * y, fma(-(x * y), y, 1.0), y);
    return y;
}

__device__ __forceinline__ cd rcp(cd a) {
    const double d = frcp(norm2(a));
    return cd{a.x * d, -(a.y * d)};
}

// A floating-point literal pinned to a scalar register pair: the polynomial kernels below
// then compile to one v_fma_f64 per Horner step with the coefficient as an SGPR operand,
// instead of a VGPR copy + v_fmac per step (and ~40 VGPRs of hoisted coefficients).
__device__ __forceinline__ double sreg(double c) {
    asm("" : "+s"(c));
    return c;
}

// sin and cos of a moderate argument (|x| < ~1e9): three-term Cody-Waite reduction by pi/2
// carried by FMAs (the first FMA x - n*C1 is exact by cancellation), then the classical
// degree-13 / degree-14 minimax kernels on [-pi/4, pi/4].  ~1 ulp, no slow path: the
// phase of the integrand's exponent stays far below 1e9 wherever the clamp lets it live.
__device__ __forceinline__ void fsincos(double x, double& s, double& c) {
    const double n = rint(x * sreg(0.63661977236758134308));  // 2/pi
    double r = fma(-n, sreg(1.5707963267948965580e+00), x);
    r = fma(-n, sreg(6.1232339957367660359e-17), r);
    r = fma(-n, sreg(-1.4973849048591698329e-33), r);
    const double z = r * r;
    // sin kernel
    double ps = fma(z, sreg(1.58969099521155010221e-10), sreg(-2.50507602534068634195e-08));
    ps = fma(z, ps, sreg(2.75573137070700676789e-06));
    ps = fma(z, ps, sreg(-1.98412698298579493134e-04));
    ps = fma(z, ps, sreg(8.33333333332248946124e-03));
    ps = fma(z, ps, sreg(-1.66666666666666324348e-01));
    const double sn = fma(r * z, ps, r);
    // cos kernel
    double pc = fma(z, sreg(-1.13596475577881948265e-11), sreg(2.08757232129817482790e-09));
    pc = fma(z, pc, sreg(-2.75573143513906633035e-07));
    pc = fma(z, pc, sreg(2.48015872894767294178e-05));
    pc = fma(z, pc, sreg(-1.38888888888741095749e-03));
    pc = fma(z, pc, sreg(4.16666666666666019037e-02));
    const double hz = 0.5 * z;
    const double wv = 1.0 - hz;
    const double cs = wv + (((1.0 - wv) - hz) + z * (z * pc));
    const int q = (int)n & 3;
    const double s0 = (q & 1) ? cs : sn;
    const double c0 = (q & 1) ? sn : cs;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

// exp(x) for x in the clamp's range [-40, 709]: x = k ln2 + r, degree-13 Taylor polynomial
// on |r| <= ln2/2 (truncation 4e-18) evaluated by Horner, scaled with ldexp.
__device__ __forceinline__ double fexp(double x) {
    const double k = rint(x * sreg(1.4426950408889634074));
    double r = fma(-k, sreg(6.93147180369123816490e-01), x);
    r = fma(-k, sreg(1.90821492927058770002e-10), r);
    double p = sreg(1.6059043836821613e-10);           // 1/13!
    p = fma(p, r, sreg(2.0876756987868100e-09));       // 1/12!
    p = fma(p, r, sreg(2.5052108385441720e-08));       // 1/11!
    p = fma(p, r, sreg(2.7557319223985888e-07));       // 1/10!
    p = fma(p, r, sreg(2.7557319223985893e-06));       // 1/9!
    p = fma(p, r, sreg(2.4801587301587302e-05));       // 1/8!
    p = fma(p, r, sreg(1.9841269841269841e-04));       // 1/7!
    p = fma(p, r, sreg(1.3888888888888889e-03));       // 1/6!
    p = fma(p, r, sreg(8.3333333333333332e-03));       // 1/5!
    p = fma(p, r, sreg(4.1666666666666664e-02));       // 1/4!
    p = fma(p, r, sreg(1.6666666666666666e-01));       // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

// The same two kernels with their coefficients handed in: a caller that evaluates many nodes
// (the unrolled cached fill) loads them ONCE into scalar registers instead of once per use.
struct TransConsts {
    double two_over_pi, pio2_1, pio2_2, pio2_3;
    double s1, s2, s3, s4, s5, s6;
    double c1, c2, c3, c4, c5, c6;
    double log2e, ln2_hi, ln2_lo;
    double e13, e12, e11, e10, e9, e8, e7, e6, e5, e4, e3;
};
__device__ __forceinline__ TransConsts trans_consts() {
    TransConsts k;
    k.two_over_pi = sreg(0.63661977236758134308);
    k.pio2_1 = sreg(1.5707963267948965580e+00);
    k.pio2_2 = sreg(6.1232339957367660359e-17);
    k.pio2_3 = sreg(-1.4973849048591698329e-33);
    k.s1 = sreg(1.58969099521155010221e-10), k.s2 = sreg(-2.50507602534068634195e-08);
    k.s3 = sreg(2.75573137070700676789e-06), k.s4 = sreg(-1.98412698298579493134e-04);
    k.s5 = sreg(8.33333333332248946124e-03), k.s6 = sreg(-1.66666666666666324348e-01);
    k.c1 = sreg(-1.13596475577881948265e-11), k.c2 = sreg(2.08757232129817482790e-09);
    k.c3 = sreg(-2.75573143513906633035e-07), k.c4 = sreg(2.48015872894767294178e-05);
    k.c5 = sreg(-1.38888888888741095749e-03), k.c6 = sreg(4.16666666666666019037e-02);
    k.log2e = sreg(1.4426950408889634074);
    k.ln2_hi = sreg(6.93147180369123816490e-01), k.ln2_lo = sreg(1.90821492927058770002e-10);
    k.e13 = sreg(1.6059043836821613e-10), k.e12 = sreg(2.0876756987868100e-09);
    k.e11 = sreg(2.5052108385441720e-08), k.e10 = sreg(2.7557319223985888e-07);
    k.e9 = sreg(2.7557319223985893e-06), k.e8 = sreg(2.4801587301587302e-05);
    k.e7 = sreg(1.9841269841269841e-04), k.e6 = sreg(1.3888888888888889e-03);
    k.e5 = sreg(8.3333333333333332e-03), k.e4 = sreg(4.1666666666666664e-02);
    k.e3 = sreg(1.6666666666666666e-01);
    return k;
}
// The same coefficients pinned in VECTOR registers: for a kernel that has registers to spare but runs out
// of scalar ones (the cooperative kernel: 124 spilled SGPRs, every eighth instruction a v_readlane /
// v_writelane of the spill lanes, with the coefficients in scalar registers).
__device__ __forceinline__ double vreg(double c) {
    asm("" : "+v"(c));
    return c;
}
__device__ __forceinline__ TransConsts trans_consts_v() {
    TransConsts k = trans_consts();
    double* f = reinterpret_cast<double*>(&k);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(TransConsts) / sizeof(double)); ++i) f[i] = vreg(f[i]);
    return k;
}
__device__ __forceinline__ void fsincos(double x, double& s, double& c, const TransConsts& k) {
    // two-term reduction: the third term of pi/2 (1.5e-33 n) is below 1e-17 for |n| < 6e15
    const double n = rint(x * k.two_over_pi);
    double r = fma(-n, k.pio2_1, x);
    r = fma(-n, k.pio2_2, r);
    const double z = r * r;
    double ps = fma(z, k.s1, k.s2);
    ps = fma(z, ps, k.s3);
    ps = fma(z, ps, k.s4);
    ps = fma(z, ps, k.s5);
    ps = fma(z, ps, k.s6);
    const double sn = fma(r * z, ps, r);
    double pc = fma(z, k.c1, k.c2);
    pc = fma(z, pc, k.c3);
    pc = fma(z, pc, k.c4);
    pc = fma(z, pc, k.c5);
    pc = fma(z, pc, k.c6);
    const double cs = fma(z * z, pc, fma(-0.5, z, 1.0));  // |error| <= 1 ulp of 1
    // quadrant q = n mod 4: (cos, sin)(x) = rotation of (cs, sn) by q quarter turns; the signs go
    // straight into the sign bits
    const unsigned q = (unsigned)(int)n;
    const bool odd = (q & 1u) != 0u;
    const double s0 = odd ? cs : sn;
    const double c0 = odd ? sn : cs;
    const unsigned long long sbit = (unsigned long long)(q & 2u) << 62;
    const unsigned long long cbit = (unsigned long long)((q + 1u) & 2u) << 62;
    s = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(s0) ^ sbit));
    c = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(c0) ^ cbit));
}
__device__ __forceinline__ double fexp(double x, const TransConsts& k) {
    const double kk = rint(x * k.log2e);
    double r = fma(-kk, k.ln2_hi, x);
    r = fma(-kk, k.ln2_lo, r);
    double p = k.e13;
    p = fma(p, r, k.e12);
    p = fma(p, r, k.e11);
    p = fma(p, r, k.e10);
    p = fma(p, r, k.e9);
    p = fma(p, r, k.e8);
    p = fma(p, r, k.e7);
    p = fma(p, r, k.e6);
    p = fma(p, r, k.e5);
    p = fma(p, r, k.e4);
    p = fma(p, r, k.e3);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)kk);
}

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
__device__ __forceinline__ void bessel_miller(cd w, double zabs, double inv_zabs, bool re_z_neg,
                                              cd& y0, cd& y1, cd& mutot) {
    const int n0 = (int)floor(zabs) + 1;
    double tn = 2.0 * n0;  // 2n, carried as a double (no per-step int->double conversion)
    // threshold evaluated once with p1 = 1: |p0 - 2n/z p1| = 2n/|z|  (functions.h:388-390);
    // compared on squares: test^2 = max(2e7 * 2n/|z|, 4e14)
    const double test2 = fmax(2.e+7 * (tn * inv_zabs), 4.e+14);
    // Upward recurrence p_{k+1} = p_{k-1} - (2n/z) p_k until |p| exceeds the threshold.
    // Two steps per trip with the roles of (a, b) swapped, so no register rotation.
    cd a = mk(0.0, 0.0), b = mk(1.0, 0.0);  // (previous, latest)
    int steps = 0;
    for (;;) {
        if (!(norm2(b) <= test2) || steps >= 4096) break;  // cap: every wave must terminate
        cd q = w * b;
        a = mk(fma(-tn, q.x, a.x), fma(-tn, q.y, a.y));  // a is now the latest
        tn += 2.0;
        ++steps;
        if (!(norm2(a) <= test2) || steps >= 4096) {
            b = a;
            break;
        }
        q = w * a;
        b = mk(fma(-tn, q.x, b.x), fma(-tn, q.y, b.y));
        tn += 2.0;
        ++steps;
    }
    // Downward recurrence from n-1 to 1 (functions.h:397-405):
    //   y_new = (2n/z) y0 + y1;  mu += 2 (Re z < 0 ? 1 - 2 (n & 1) : 1) * (old y0)
    int n = n0 + steps - 1;  // first n of the loop
    tn -= 2.0;
    cd u = rcp(b), v = mk(0.0, 0.0);  // (latest y0, previous y1)
    cd mu = mk(0.0, 0.0);
    double sg = re_z_neg ? ((n & 1) ? -2.0 : 2.0) : 2.0;
    const double flip = re_z_neg ? -1.0 : 1.0;
    for (; n >= 2; n -= 2) {
        cd q = w * u;
        v = mk(fma(tn, q.x, v.x), fma(tn, q.y, v.y));  // v is now the latest
        mu = mk(fma(sg, u.x, mu.x), fma(sg, u.y, mu.y));
        sg *= flip;
        tn -= 2.0;
        q = w * v;
        u = mk(fma(tn, q.x, u.x), fma(tn, q.y, u.y));  // u is the latest again
        mu = mk(fma(sg, v.x, mu.x), fma(sg, v.y, mu.y));
        sg *= flip;
        tn -= 2.0;
    }
    if (n == 1) {
        const cd q = w * u;
        const cd t = mk(fma(tn, q.x, v.x), fma(tn, q.y, v.y));
        mu = mk(fma(sg, u.x, mu.x), fma(sg, u.y, mu.y));
        v = u;
        u = t;
    }
    y0 = u;
    y1 = v;
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
// Everything the integrand derives from (t, 1/t, u) and the pair constants.
struct NodeTerms {
    cd e, taut, lam, rl, nv, nv2;
    double r2, ou, rlabs;
};
__device__ __forceinline__ NodeTerms node_terms(double t, double inv_t, double u,
                                                const PairConst& pc, double omi) {
    NodeTerms n;
    const double r1 = frsqrt(fma(u, u, 1.0));
    n.r2 = r1 * r1;
    n.ou = omi * u;
    n.e = mk(r1, -(n.ou * r1));
    n.taut = t * n.e;
    n.lam = mk(fma(-pc.c_lam, n.taut.y, 1.0), pc.c_lam * n.taut.x);
    n.rlabs = frsqrt(norm2(n.lam));  // 1 / |lambda|
    const double rl2 = n.rlabs * n.rlabs;
    n.rl = mk(n.lam.x * rl2, -(n.lam.y * rl2));  // 1 / lambda
    n.nv = (pc.c_nv * inv_t) * conj(n.e);
    n.nv2 = n.nv * n.nv;
    return n;
}

#ifndef EMME_REMAT
#define EMME_REMAT 0
#endif
#ifndef EMME_FAST_TRANSCENDENTALS
#define EMME_FAST_TRANSCENDENTALS 1
#endif

__device__ __forceinline__ cd integrand(double x, const DevParams& P, const PairConst& pc,
                                        const OmegaConst& oc, int m) {
    double sx, cx;
    sincos(x, &sx, &cx);
    const double rsc = frcp(sx * cx);
    const double inv_cx = sx * rsc;
    double t = sx * inv_cx;
    const double inv_c2 = inv_cx * inv_cx;
    double inv_t = cx * (cx * rsc);
    double u = t * P.inv_arc;

    // ---- stage A: exponent of the integrand and the underflow clamp -----------------
    cd arg, w;
    double zabs, inv_zabs;
    bool zneg;
    {
        const NodeTerms n = node_terms(t, inv_t, u, pc, oc.omi);
        const cd z = pc.s * n.rl;
        // log_coef, src/Parameters.cpp:154-165
        cd L = (-0.5) * n.nv2 + times_i(n.taut * oc.omega);
        L = L + (-0.5 * pc.beta1) * times_i(n.nv);
        L = L - (0.5 * pc.bsum) * n.rl;
        zneg = z.x < 0.0;
        arg = zneg ? (L - z) : (L + z);  // log_coef - (Re z<0 ? z : -z)
        if (!(arg.x >= -40.)) {
            // safe_exp clamp (:167-173); NaN falls through and is propagated below
            if (arg.x < -40.) return mk(0.0, 0.0);
        }
        w = pc.inv_s * n.lam;
        zabs = pc.s * n.rlabs;
        inv_zabs = pc.inv_s * (norm2(n.lam) * n.rlabs);
    }

    // ---- stage B: Miller recurrence (the long, data-dependent part) -------------------
    cd y0, y1, mutot;
    bessel_miller(w, zabs, inv_zabs, zneg, y0, y1, mutot);

#if EMME_REMAT
    // Only (t, 1/t, u, 1/cos^2, arg) are meant to stay live across the recurrence; the
    // node terms are recomputed (~45 instructions) instead of being held in ~30 VGPRs.
    asm volatile("" : "+v"(t), "+v"(inv_t), "+v"(u));
#endif
    // ---- stage C: coefficients and assembly of F ---------------------------------------
    const NodeTerms n = node_terms(t, inv_t, u, pc, oc.omi);
    double sa, ca;
#if EMME_FAST_TRANSCENDENTALS
    fsincos(arg.y, sa, ca);
    const double ea = fexp(arg.x);
#else
    sincos(arg.y, &sa, &ca);
    const double ea = exp(arg.x);
#endif
    const cd sexp = mk(ea * ca, ea * sa);

    const cd rl3 = n.rl * n.rl * n.rl;  // lambda^-3 (reference: pow(lambda, -3.) via log/polar)
    const double wsi_eta = P.omega_s_i * P.eta_i;
    const cd a0 = oc.omega - P.omega_s_i * mk(fma(P.eta_i, fma(0.5, n.nv2.x, -1.5), 1.0),
                                               P.eta_i * 0.5 * n.nv2.y);
    const cd i0c = a0 * n.rl + (wsi_eta * mk(0.5 * pc.bsum - n.lam.x, -n.lam.y)) * rl3;
    const cd i1c = (-wsi_eta * pc.s) * rl3;

    cd F = mk(inv_t, -(n.ou * n.r2 * inv_t)) * sexp;  // (jacob / t~) * safe_exp
    if (m == 1)
        F = F * n.nv;
    else if (m == 2)
        F = F * n.nv2;
    F = F * (i0c * y0 + i1c * y1);
    F = F * rcp(mutot);
    return inv_c2 * F;
}

// ---- omega-independent half of the integrand ------------------------------------------
// For fixed pair, moment and contour sense (omi) the integrand at abscissa x is
//     F(omega) = exp(A0 + T omega) (omega Q1 + Q0)      [0 when Re(A0 + T omega) < -40]
// with  T  = i t~,
//       A0 = -nv^2/2 - i beta_1 nv/2 - bsum/(2 lambda) -/+ z          (log_coef without i t~ omega)
//       Q1 = pre * y0/lambda,   Q0 = pre * (C0 y0 + i1c y1),
//       pre = nv^m (jacob/t~) / (mu cos^2 x),
//       C0 = -ws_i (1 + eta_i (nv^2/2 - 3/2))/lambda + ws_i eta_i (bsum/2 - lambda)/lambda^3.
// Everything expensive -- sincos(x), the two rsqrt, the Miller recurrence -- lives here and
// is shared by every omega of a batch (src/Parameters.cpp:120-176 regrouped by omega).
struct NodeData {
    cd A0, T, Q1, Q0;
};
__device__ __forceinline__ NodeData node_data(double x, const DevParams& P, const PairConst& pc,
                                              double omi, int m) {
    double sx, cx;
    sincos(x, &sx, &cx);
    const double rsc = frcp(sx * cx);
    const double inv_cx = sx * rsc;
    const double t = sx * inv_cx;
    const double inv_c2 = inv_cx * inv_cx;
    const double inv_t = cx * (cx * rsc);
    const double u = t * P.inv_arc;
    const NodeTerms n = node_terms(t, inv_t, u, pc, omi);

    const cd z = pc.s * n.rl;
    const bool zneg = z.x < 0.0;
    cd L0 = (-0.5) * n.nv2 + (-0.5 * pc.beta1) * times_i(n.nv);
    L0 = L0 - (0.5 * pc.bsum) * n.rl;

    cd y0, y1, mutot;
    bessel_miller(pc.inv_s * n.lam, pc.s * n.rlabs, pc.inv_s * (norm2(n.lam) * n.rlabs), zneg, y0,
                  y1, mutot);

    const cd rl3 = n.rl * n.rl * n.rl;
    const double wsi_eta = P.omega_s_i * P.eta_i;
    const cd c0 = (-P.omega_s_i) * (mk(fma(P.eta_i, fma(0.5, n.nv2.x, -1.5), 1.0),
                                       P.eta_i * 0.5 * n.nv2.y) * n.rl) +
                  (wsi_eta * mk(0.5 * pc.bsum - n.lam.x, -n.lam.y)) * rl3;
    const cd i1c = (-wsi_eta * pc.s) * rl3;

    cd pre = mk(inv_t, -(n.ou * n.r2 * inv_t));  // jacob / t~
    if (m == 1)
        pre = pre * n.nv;
    else if (m == 2)
        pre = pre * n.nv2;
    pre = (inv_c2 * pre) * rcp(mutot);

    NodeData d;
    d.A0 = zneg ? (L0 - z) : (L0 + z);
    d.T = times_i(n.taut);
    d.Q1 = pre * (n.rl * y0);
    d.Q0 = pre * (c0 * y0 + i1c * y1);
    return d;
}

// The moment factor: F_m = F_0 * norm_vel^m with norm_vel = c_nv * W, W = conj(e) / t (pair-
// independent; src/Parameters.cpp:142, 158-161).  Electromagnetic fills keep ONE record per
// (pair, interval, node) -- the m = 0 one -- and a table of W per (interval, node).
__device__ __forceinline__ cd node_w(double x, const DevParams& P, double omi) {
    double sx, cx;
    sincos(x, &sx, &cx);
    const double rsc = frcp(sx * cx);
    const double inv_cx = sx * rsc;
    const double t = sx * inv_cx;
    const double inv_t = cx * (cx * rsc);
    const double u = t * P.inv_arc;
    const double r1 = frsqrt(fma(u, u, 1.0));
    return mk(inv_t * r1, inv_t * ((omi * u) * r1));  // (1/t) conj(e), e = r1 (1 - i omi u)
}

// omega-dependent half: one complex exponential and three complex products.
__device__ __forceinline__ cd node_eval(const NodeData& d, cd omega) {
    const cd arg = d.A0 + d.T * omega;
    if (!(arg.x >= -40.)) {
        if (arg.x < -40.) return mk(0.0, 0.0);  // safe_exp clamp, src/Parameters.cpp:167-173
    }
    double sa, ca;
    fsincos(arg.y, sa, ca);
    const double ea = fexp(arg.x);
    return mk(ea * ca, ea * sa) * (omega * d.Q1 + d.Q0);
}

__device__ __forceinline__ cd node_eval(const NodeData& d, cd omega, const TransConsts& k) {
    // arg = A0 + T omega and S = omega Q1 + Q0 as fused chains (two FMAs per component)
    const double ax = fma(d.T.x, omega.x, fma(-d.T.y, omega.y, d.A0.x));
    if (!(ax >= -40.)) {
        if (ax < -40.) return mk(0.0, 0.0);  // safe_exp clamp, src/Parameters.cpp:167-173
    }
    const double ay = fma(d.T.x, omega.y, fma(d.T.y, omega.x, d.A0.y));
    double sa, ca;
    fsincos(ay, sa, ca, k);
    const double ea = fexp(ax, k);
    const cd S = mk(fma(omega.x, d.Q1.x, fma(-omega.y, d.Q1.y, d.Q0.x)),
                    fma(omega.x, d.Q1.y, fma(omega.y, d.Q1.x, d.Q0.y)));
    return mk(ea * ca, ea * sa) * S;
}

// Gauss-Kronrod node tables laid out per lane of a group (centre, +x_1..+x_h, -x_1..-x_h,
// pad).  Values: include/functions.h:93-120 (15 points) and :126-161 (31 points).
struct GkLane {
    double x, wk, wg;
};

}  // namespace emme
