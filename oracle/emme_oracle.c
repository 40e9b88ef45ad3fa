/* emme_oracle.c -- TEST INFRASTRUCTURE ONLY (see emme_oracle.h for who may load it).
 *
 * From-scratch C99 restatement of the reference hot path.  Each block cites the
 * reference file:line it follows.  The arithmetic is spelled operation by operation so
 * that it reproduces what libstdc++'s std::complex<double> operators do in the reference
 * binary (component-wise complex*real and complex/real; full complex*complex;
 * complex/complex and real/complex through libgcc's __divdc3; pow(complex,real) through
 * log/polar, /usr/include/c++/11/complex:1028-1039).  Build with -ffp-contract=off and
 * no -march (reference Makefile:13-14 has neither, hence no FMA contraction).
 */
#include "emme_oracle.h"

#include <complex.h>
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef double complex cplx;

/* ---- std::complex<double> operator spellings ------------------------------------ */
static inline cplx c_mk(double re, double im) { return CMPLX(re, im); }
static inline cplx c_scale(cplx z, double d) { return CMPLX(creal(z) * d, cimag(z) * d); }
static inline cplx c_divr(cplx z, double d) { return CMPLX(creal(z) / d, cimag(z) / d); }
static inline cplx c_addr(cplx z, double d) { return CMPLX(creal(z) + d, cimag(z)); }
static inline cplx c_mul(cplx a, cplx b) { return a * b; }            /* inline naive + __muldc3 on NaN */
static inline cplx c_div(cplx a, cplx b) { return a / b; }            /* __divdc3 */
static inline cplx r_div(double x, cplx z) { return CMPLX(x, 0.0) / z; } /* T / complex<T> */
static inline cplx c_neg(cplx z) { return CMPLX(-creal(z), -cimag(z)); }
/* T - complex<T>: libstdc++ builds -y then adds x to the real part */
static inline cplx r_sub(double x, cplx y) { return CMPLX(-creal(y) + x, -cimag(y)); }

/* pow(const complex<T>&, const T&), /usr/include/c++/11/complex:1028-1039 */
static cplx c_pow_real(cplx x, double y) {
    if (cimag(x) == 0.0 && creal(x) > 0.0) return CMPLX(pow(creal(x), y), 0.0);
    cplx t = clog(x);
    double rho = exp(y * creal(t));
    double theta = y * cimag(t);
    return CMPLX(rho * cos(theta), rho * sin(theta)); /* std::polar */
}

/* ---- parameters (src/Parameters.cpp:36-66, 211-223, 395-398) ---------------------- */
static double zero_point(double a) {
    /* src/functions.cpp:32-65 bisection of cos x + a x sin x on [0, pi], tol 1e-9,
     * at most 100 halvings (defaults include/functions.h:496-498) */
    const double tol = 1e-9;
    double lo = 0.0, hi = M_PI, mid = 0.0;
    double flo = cos(lo) + a * lo * sin(lo);
    if (fabs(flo) < tol) return lo;
    if (fabs(cos(hi) + a * hi * sin(hi)) < tol) return hi;
    for (int it = 0; it < 100; ++it) {
        mid = lo + (hi - lo) / 2.0;
        double fm = cos(mid) + a * mid * sin(mid);
        if (fabs(fm) < tol || (hi - lo) / 2.0 < tol) return mid;
        if ((cos(lo) + a * lo * sin(lo)) * fm < 0)
            hi = mid;
        else
            lo = mid;
    }
    return mid;
}

void oracle_params_derive(emme_params_t* p) {
    p->b_theta = p->k_rho * p->k_rho;
    p->alpha = p->q * p->q * p->R * p->beta_e / (p->epsilon_n * p->R) *
               ((1 + p->eta_e) + 1 / p->tau * (1 + p->eta_i));
    p->omega_s_i = -(sqrt(p->b_theta) * p->vt) / (p->epsilon_n * p->R);
    p->omega_s_e = -p->tau * p->omega_s_i;
    p->omega_d_bar = 2.0 * p->epsilon_n * p->omega_s_i * p->omega_d_coeff;
    p->deltap = p->beta_e_p = p->rdeltapp = p->curvature_aver = 0.0;
    p->shat_coeff = 0.0;
    if (p->conf == EMME_CONF_STELLARATOR) {
        p->deltap = -0.25 * p->alpha;
        p->beta_e_p = p->beta_e * (1.0 + p->eta_e) / (p->epsilon_n * p->R);
        p->rdeltapp = (-p->alpha + (2.0 * p->shat - 3) * p->deltap);
        /* `mh / lh` is INTEGER division in the reference (int members) */
        p->curvature_aver = (p->mh / p->lh) * p->r_over_R / (p->q * p->R) * (4.0 - p->shat) +
                            (-p->alpha + 2 * p->shat * p->deltap + 0) / p->R;
    } else if (p->conf == EMME_CONF_CYLINDER) {
        /* src/functions.cpp:72-83 */
        double x0 = zero_point(p->shat);
        double integral = (1.0 + p->shat) * sin(x0) - p->shat * x0 * cos(x0);
        p->shat_coeff = integral / x0;
    }
}

/* ---- grid and weights ------------------------------------------------------------ */
double oracle_grid(double len, unsigned n, double* eta) {
    double dx = (2 * len) / (n - 1); /* include/Grid.h:11 (unsigned n-1 -> double) */
    for (unsigned i = 0; i < n; ++i) eta[i] = -len + i * dx;
    return dx;
}

double oracle_weight(int n, int i, int j) {
    /* src/singularity_handler.cpp:4-20 */
    static const double c[6] = {0.0,
                                2.951388888888883,
                                -2.4305555555555305,
                                4.166666666667441,
                                -0.3472222222224549,
                                1.159722222222284};
    int d = abs(i - j);
    double w = d <= 5 ? c[d] : 1.0;
    if (j == 0 || j == n - 1) w -= 0.5;
    return w;
}

/* ---- geometry: g(eta) and b(eta) ------------------------------------------------- */
static double g_tokamak(const emme_params_t* p, double eta) {
    /* src/Parameters.cpp:76-85; note `3 / 2` is integer 1 there, so the last
     * denominator is pow(eps_r^2 + q^2, 1) */
    return -((p->alpha * eta) / 2.0) + p->shat * p->theta * cos(eta) - p->shat * eta * cos(eta) +
           sin(eta) + p->shat * sin(eta) + 0.25 * p->alpha * sin(2.0 * eta) -
           (1 - p->shat) * p->q * p->epsilon_r /
               pow((pow(p->epsilon_r, 2) + pow(p->q, 2)), 1) * eta;
}

static double g_stellarator(const emme_params_t* p, double eta) {
    /* src/Parameters.cpp:248-393 is a machine-expanded polynomial in lh, mh*q whose
     * ~110 terms collapse, with L = lh - mh q and S = deltap + rdeltapp + deltap*shat,
     * to the closed form below (the common factor 2 (L-1) L^2 (L+1) of numerator and
     * denominator cancels).  Agreement with the expanded expression is tested to
     * <= 2e-13 relative in tests/test_oracle_vs_reference.py. */
    const double lh = p->lh, mq = p->mh * p->q;
    const double L = lh - mq;
    const double S = p->deltap + p->rdeltapp + p->deltap * p->shat;
    const double eh = p->epsilon_h_t;
    const double phi = eta * L - p->alpha_0 * p->mh;
    const double de = eta - p->eta_k;
    double g = 0.5 * eta * (S + p->curvature_aver * p->R);
    g += -p->shat * de * cos(eta) + (1.0 + p->shat) * sin(eta) - 0.25 * S * sin(2.0 * eta);
    g += -eh * p->shat * lh * de * cos(phi) / L + eh * lh * (L + p->shat) * sin(phi) / (L * L);
    g += -0.5 * S * eh * lh * (sin(eta + phi) / (L + 1.0) + sin(eta - phi) / (L - 1.0));
    return g;
}

static double g_taylor(const emme_params_t* p, double eta) {
    /* src/Parameters.cpp:404-436, Pade {3,4} */
    const double a = p->alpha, s = p->shat;
    const double den = 7 + 16 * a + 40 * pow(a, 2) - 28 * s - 80 * a * s + 40 * pow(s, 2);
    const double c3 = -31 - 96 * a - 168 * pow(a, 2) - 560 * pow(a, 3) + 186 * s + 672 * a * s +
                      1680 * pow(a, 2) * s - 504 * pow(s, 2) - 1680 * a * pow(s, 2) +
                      560 * pow(s, 3);
    const double d2 = 3 + 19 * a + 56 * pow(a, 2) - 18 * s - 84 * a * s + 28 * pow(s, 2);
    const double d4 = 11 - 4 * a + 704 * pow(a, 2) - 88 * s - 584 * a * s + 216 * pow(s, 2);
    return (eta + (pow(eta, 3) * c3) / (42. * den)) /
           (1 + (pow(eta, 2) * d2) / (7. * den) + (pow(eta, 4) * d4) / (840. * den));
}

double oracle_g(const emme_params_t* p, double eta) {
    switch (p->conf) {
        case EMME_CONF_TOKAMAK: return g_tokamak(p, eta);
        case EMME_CONF_STELLARATOR: return g_stellarator(p, eta);
        case EMME_CONF_CYLINDER: return eta * p->shat_coeff; /* :400-402 */
        case EMME_CONF_TAYLOR_MD: return g_taylor(p, eta);
        default: return eta; /* cylinder old, :438-440 */
    }
}

double oracle_bi(const emme_params_t* p, double eta) {
    if (p->conf == EMME_CONF_STELLARATOR) {
        /* src/Parameters.cpp:225-232 */
        double sig = p->shat * (eta - p->eta_k) +
                     (p->deltap * (1 + p->shat) + p->rdeltapp) * sin(eta);
        return p->b_theta * (1.0 + pow(sig, 2));
    }
    /* src/Parameters.cpp:97-100 */
    return p->b_theta * (1.0 + pow(p->shat * (eta - p->theta) - p->alpha * sin(eta), 2));
}

/* ---- Miller recurrence for I0, I1 (include/functions.h:381-408) -------------------- */
typedef struct {
    cplx y0, y1, mu, zexp;
} bessel_t;

static bessel_t bessel_alter(cplx z) {
    const double THRESHOLD = 2.e+7;
    int n = (int)(floor(cabs(z)) + 1);
    cplx p0 = 0.0, p1 = 1.0, pt;
    double test = fmax(
        sqrt(THRESHOLD * cabs(p1) * cabs(p0 - c_mul(r_div(2.0 * n, z), p1))), THRESHOLD);
    for (; cabs(p1) <= test; ++n) {
        pt = p0 - c_mul(r_div(2.0 * n, z), p1);
        p0 = p1;
        p1 = pt;
    }
    cplx y0 = r_div(1.0, p1), y1 = 0.0, yt, mu = 0.0;
    const int neg = creal(z) < 0;
    for (n--; n > 0; --n) {
        yt = c_mul(r_div(2. * n, z), y0) + y1;
        y1 = y0;
        y0 = yt;
        mu += c_scale(y1, 2. * (neg ? 1 - 2 * (n & 1) : 1));
    }
    bessel_t r = {y0, y1, mu + y0, neg ? z : c_neg(z)};
    return r;
}

void oracle_bessel(double zre, double zim, double* o) {
    bessel_t b = bessel_alter(c_mk(zre, zim));
    o[0] = creal(b.y0), o[1] = cimag(b.y0);
    o[2] = creal(b.y1), o[3] = cimag(b.y1);
    o[4] = creal(b.mu), o[5] = cimag(b.mu);
    o[6] = creal(b.zexp), o[7] = cimag(b.zexp);
}

/* ---- Gauss-Kronrod tables (values of include/functions.h:93-120, 126-161) ---------- */
static const double GK15_X[8] = {0.,
                                 0.20778495500789847,
                                 0.40584515137739717,
                                 0.58608723546769113,
                                 0.74153118559939444,
                                 0.86486442335976907,
                                 0.94910791234275852,
                                 0.99145537112081264};
static const double GK15_WG[4] = {0.41795918367346939, 0.38183005050511894, 0.27970539148927667,
                                  0.12948496616886969};
static const double GK15_WK[8] = {2.09482141084727828e-01, 2.04432940075298892e-01,
                                  1.90350578064785410e-01, 1.69004726639267903e-01,
                                  1.40653259715525919e-01, 1.04790010322250184e-01,
                                  6.30920926299785533e-02, 2.29353220105292250e-02};
static const double GK31_X[16] = {0.0,
                                  0.1011420669187175,
                                  0.20119409399743452,
                                  0.29918000715316881,
                                  0.39415134707756337,
                                  0.48508186364023968,
                                  0.57097217260853885,
                                  0.65099674129741697,
                                  0.72441773136017005,
                                  0.79041850144246593,
                                  0.84820658341042722,
                                  0.8972645323440819,
                                  0.9372733924007059,
                                  0.96773907567913913,
                                  0.98799251802048543,
                                  0.99800229869339706};
static const double GK31_WG[8] = {0.20257824192556112, 0.19843148532711152, 0.18616100001556193,
                                  0.1662692058169939,  0.1395706779261542,  0.10715922046717143,
                                  0.07036604748810768, 0.030753241996119};
static const double GK31_WK[16] = {
    0.10133000701479155,   0.100769845523875595,  0.099173598721791959,  0.0966427269836236785,
    0.093126598170825321,  0.0885644430562117706, 0.083080502823133021,  0.0768496807577203789,
    0.069854121318728259,  0.0620095678006706403, 0.053481524690928087,  0.0445897513247648766,
    0.035346360791375846,  0.0254608473267153202, 0.0150079473293161225, 0.00537747987292334899};

typedef cplx (*integrand_fn)(double t, void* ctx);

/* diagnostic: number of evaluated intervals per bisection depth (not thread safe; read
 * with oracle_depth_hist after a single-threaded run) */
static long g_depth_hist[64];
void oracle_depth_hist(long* out64, int reset) {
    for (int i = 0; i < 64; ++i) {
        out64[i] = g_depth_hist[i];
        if (reset) g_depth_hist[i] = 0;
    }
}

/* diagnostic: the evaluated intervals of the calling thread's next integrals, in evaluation
 * order, as depth << 56 | index (index = position of the interval among the 2^depth of its depth) */
static __thread long* g_trace_buf = 0;
static __thread long g_trace_cap = 0, g_trace_n = 0;
void oracle_trace_set(long* buf, long cap) {
    g_trace_buf = buf, g_trace_cap = cap, g_trace_n = 0;
}
long oracle_trace_count(void) { return g_trace_n; }

/* x -> f(tan x)/cos^2 x, include/functions.h:313-316 */
static inline cplx mapped(integrand_fn f, void* ctx, double x) {
    const double c = cos(x);
    return c_divr(f(tan(x), ctx), c * c);
}

/* include/functions.h:181-209 + 211-251 + 305-331.  Returns interval count. */
static long gk_adaptive_0_inf(integrand_fn f, void* ctx, double rel_tol, double prec_goal,
                              unsigned long max_sub, unsigned long pts, cplx* result,
                              int* status) {
    const double *X, *WG, *WK;
    int half;
    if (pts == 15) {
        X = GK15_X, WG = GK15_WG, WK = GK15_WK, half = 8;
    } else if (pts == 31) {
        X = GK31_X, WG = GK31_WG, WK = GK31_WK, half = 16;
    } else {
        *status = -1; /* "integration_start_points should be 15 or 31" (:329) */
        *result = 0.0;
        return 0;
    }
    const int gauss_order = (int)((pts - 1) / 2);
    const double a = 0, b = M_PI / 2.0;
    const double inv_scale = 2. / (b - a);
    double abs_tol = 0.0;
    cplx sum = 0.0;
    long nint = 0;
    size_t cap = 64, top = 0;
    double(*stack)[2] = malloc(cap * sizeof *stack);
    stack[top][0] = a, stack[top][1] = b, ++top;
    while (top) {
        --top;
        const double l = stack[top][0], r = stack[top][1];
        const double mid = (r + l) / 2;
        const double scale = (r - l) / 2;
        /* basic rule on [-1,1] */
        cplx f0 = mapped(f, ctx, scale * 0.0 + mid);
        cplx G = (gauss_order & 1) ? c_scale(f0, WG[0]) : 0.0;
        cplx K = c_scale(f0, WK[0]);
        for (int i = 1; i < half; ++i) {
            cplx fp = mapped(f, ctx, scale * X[i] + mid);
            cplx fm = mapped(f, ctx, scale * (-X[i]) + mid);
            cplx fs = fp + fm;
            G += ((gauss_order - i) & 1) ? c_scale(fs, WG[i / 2]) : (cplx)0.0;
            K += c_scale(fs, WK[i]);
        }
        double err = fmax(cabs(K - G), cabs(K) * DBL_EPSILON * 2);
        cplx integral = c_scale(K, scale);
        err = err * scale;
        ++nint;
        {
            int dd = (int)(log2((b - a) / (r - l)) + 0.5);
            if (dd >= 0 && dd < 64) ++g_depth_hist[dd];
            if (g_trace_buf) {
                if (g_trace_n < g_trace_cap && dd >= 0 && dd < 56)
                    g_trace_buf[g_trace_n] = ((long)dd << 56) | (long)(l / (b - a) * ldexp(1.0, dd) + 0.5);
                ++g_trace_n;
            }
        }
        if (fpclassify(abs_tol) == FP_ZERO) abs_tol = cabs(c_scale(integral, rel_tol));
        if (ldexp(scale, (int)max_sub) > 0.99 * (b - a) && err > abs_tol * inv_scale + prec_goal &&
            err > cabs(c_scale(integral, rel_tol)) + prec_goal) {
            if (top + 2 > cap) {
                cap *= 2;
                stack = realloc(stack, cap * sizeof *stack);
            }
            stack[top][0] = mid, stack[top][1] = r, ++top;
            stack[top][0] = l, stack[top][1] = mid, ++top;
        } else {
            sum += integral;
        }
    }
    free(stack);
    *status = 0;
    *result = sum;
    return nint;
}

/* ---- kappa integrand (src/Parameters.cpp:120-176) ---------------------------------- */
typedef struct {
    const emme_params_t* p;
    unsigned m;
    double eta, eta_p;
    cplx omega;
    int recompute;
    /* per-pair hoist (identical values, see oracle_kappa) */
    double beta1, bi, bip;
} kctx_t;

static double beta_1(const emme_params_t* p, double eta, double eta_p) {
    /* src/Parameters.cpp:87-90 */
    return (p->q * p->R) / p->vt * (p->omega_d_bar) * (oracle_g(p, eta) - oracle_g(p, eta_p));
}

static double beta_1_e(const emme_params_t* p, double eta, double eta_p) {
    /* src/Parameters.cpp:92-95 */
    return (p->q * p->R) / p->vt * (p->omega_d_bar * p->omega_s_e / p->omega_s_i) *
           (oracle_g(p, eta) - oracle_g(p, eta_p));
}

static cplx kappa_integrand(double t, void* vctx) {
    const kctx_t* k = vctx;
    const emme_params_t* p = k->p;
    const double eta = k->eta, eta_p = k->eta_p;
    const cplx omega = k->omega;
    const cplx IU = c_mk(0.0, 1.0);

    const double omi = -copysign(1, creal(omega));
    /* exp(-omi * 1.i * atan(t/arc)) */
    const cplx exp_arg = cexp(c_scale(c_scale(IU, -omi), atan(t / p->arc_coeff)));
    const cplx taut = c_scale(exp_arg, t);
    const cplx jacob =
        exp_arg - c_divr(c_scale(c_scale(c_mul(IU, exp_arg), omi), t),
                         p->arc_coeff * (1.0 + pow((t / p->arc_coeff), 2)));

    /* lambda_f_tau, src/Parameters.cpp:101-106 (calls beta_1 again in the reference) */
    const double b1_l = k->recompute ? beta_1(p, eta, eta_p) : k->beta1;
    const cplx lam = c_addr(
        c_scale(c_divr(c_mul(c_scale(IU, 0.5), c_scale(taut, p->vt)), p->q * p->R * (eta - eta_p)),
                b1_l),
        1.0);
    const double bi_eta = k->recompute ? oracle_bi(p, eta) : k->bi;
    const double bi_eta_p = k->recompute ? oracle_bi(p, eta_p) : k->bip;

    const bessel_t bs = bessel_alter(r_div(sqrt(bi_eta * bi_eta_p), lam));

    const cplx lam3i = c_pow_real(lam, -3.);
    const cplx nv = r_div(p->q * p->R * (eta - eta_p), c_scale(taut, p->vt));

    const cplx nv2h = c_mul(c_scale(nv, 0.5), nv); /* 0.5 * nv * nv */
    const cplx i0_coef =
        c_div(omega - c_scale(c_addr(c_scale(c_addr(nv2h, -1.5), p->eta_i), 1.0), p->omega_s_i),
              lam) +
        c_mul(c_scale(r_sub(.5 * (bi_eta + bi_eta_p), lam), p->omega_s_i * p->eta_i), lam3i);
    const cplx i1_coef = c_scale(lam3i, -p->omega_s_i * p->eta_i * sqrt(bi_eta * bi_eta_p));

    const double b1 = k->recompute ? beta_1(p, eta, eta_p) : k->beta1;

    const cplx log_norm_vel = c_mul(c_scale(nv, -0.5), nv);
    const cplx log_i_beta = c_mul(c_scale(c_neg(c_mk(0.0, .5)), b1), nv);
    const cplx log_hf_tau = c_mul(c_mul(IU, taut), omega);
    const cplx log_exp_term = r_div(-(bi_eta + bi_eta_p), c_addr(c_div(c_scale(IU, b1), nv), 2.0));
    const cplx log_coef = log_norm_vel + log_i_beta + log_hf_tau + log_exp_term;

    const cplx ev = log_coef - bs.zexp;
    const cplx sexp = creal(ev) < -40. ? (cplx)0.0 : cexp(ev);

    cplx r = c_div(c_pow_real(nv, (double)k->m), taut);
    r = c_mul(r, jacob);
    r = c_mul(r, sexp);
    r = c_mul(r, c_mul(i0_coef, bs.y0) + c_mul(i1_coef, bs.y1));
    return c_div(r, bs.mu);
}

long oracle_kappa(const emme_params_t* p, unsigned m, double eta, double eta_p, double wre,
                  double wim, int recompute, double* out2) {
    kctx_t k = {p, m, eta, eta_p, c_mk(wre, wim), recompute, 0, 0, 0};
    if (!recompute) {
        k.beta1 = beta_1(p, eta, eta_p);
        k.bi = oracle_bi(p, eta);
        k.bip = oracle_bi(p, eta_p);
    }
    cplx res;
    int status;
    long n = gk_adaptive_0_inf(kappa_integrand, &k, p->integration_precision,
                               p->integration_accuracy, (unsigned long)p->integration_iteration_limit,
                               (unsigned long)p->integration_start_points, &res, &status);
    if (status) {
        out2[0] = out2[1] = NAN;
        return -1;
    }
    /* -i (qR) / (vt sqrt(2 pi)) * result, src/Parameters.cpp:182-183 */
    cplx pref = c_divr(c_scale(c_neg(c_mk(0, 1.0)), p->q * p->R), p->vt * sqrt((2.0 * M_PI)));
    cplx kappa = c_mul(pref, res);
    out2[0] = creal(kappa), out2[1] = cimag(kappa);
    return n;
}

static cplx kappa_e(const emme_params_t* p, unsigned m, double eta, double eta_p, cplx omega,
                    int* bad) {
    /* src/Parameters.cpp:186-209 */
    const double de = eta - eta_p;
    switch (m) {
        case 0: return 0.0;
        case 1: {
            cplx c = c_divr(c_scale(c_neg(c_mk(0.0, 1.0)), p->q * p->R), 2.0 * p->vt * p->tau);
            c = c_mul(c, c_addr(omega, -p->omega_s_e));
            return c_divr(c_scale(c, de), fabs(de));
        }
        case 2: {
            double f = (p->q * p->q * p->R * p->R) / (2.0 * p->vt * p->vt * p->tau) * de / fabs(de);
            cplx a = c_scale(c_mul(omega, c_addr(omega, -p->omega_s_e)), de);
            cplx b = c_scale(c_addr(omega, -(p->omega_s_e * (1.0 + p->eta_e))),
                             beta_1_e(p, eta, eta_p) * p->vt / (p->q * p->R));
            return c_scale(a - b, f);
        }
        default: *bad = 1; return 0.0;
    }
}

int oracle_kappa_e(const emme_params_t* p, unsigned m, double eta, double eta_p, double wre,
                   double wim, double* out2) {
    int bad = 0;
    cplx r = kappa_e(p, m, eta, eta_p, c_mk(wre, wim), &bad);
    out2[0] = creal(r), out2[1] = cimag(r);
    return bad ? -1 : 0;
}

/* ---- quadrature self-test integrand ------------------------------------------------ */
typedef struct {
    cplx a;
    double pw;
} tctx_t;
static cplx test_integrand(double t, void* v) {
    const tctx_t* c = v;
    return c_scale(cexp(c_scale(c->a, t)), pow(t, c->pw));
}
long oracle_integrate_test(double ar, double ai, double pw, double tol, double prec,
                           unsigned long max_sub, unsigned long pts, double* out2) {
    tctx_t c = {c_mk(ar, ai), pw};
    cplx res;
    int st;
    long n = gk_adaptive_0_inf(test_integrand, &c, tol, prec, max_sub, pts, &res, &st);
    out2[0] = creal(res), out2[1] = cimag(res);
    return st ? -1 : n;
}

/* ---- assembly (include/solver.h:417-515) ------------------------------------------- */
typedef struct {
    const emme_params_t* p;
    cplx omega;
    cplx* M;
    long* counts;
    const double* eta;
    double dx;
    unsigned N;
    size_t dim;
    int es, recompute;
    long next; /* shared pair cursor */
    long npairs;
    long intervals;
    int error;
    pthread_mutex_t mu;
} asm_t;

static void pair_from_index(unsigned N, long k, unsigned* pi, unsigned* pj) {
    /* row-major enumeration of i<j */
    unsigned i = 0;
    long rem = k;
    while (rem >= (long)(N - 1 - i)) {
        rem -= (N - 1 - i);
        ++i;
    }
    *pi = i;
    *pj = i + 1 + (unsigned)rem;
}

static void* asm_worker(void* v) {
    asm_t* a = v;
    const emme_params_t* p = a->p;
    const unsigned N = a->N;
    const size_t dim = a->dim;
    long local_int = 0;
    for (;;) {
        long k0, k1;
        pthread_mutex_lock(&a->mu);
        k0 = a->next;
        a->next += 16; /* reference pool hands out batches of 16 (DedicatedThreadPool.h) */
        pthread_mutex_unlock(&a->mu);
        if (k0 >= a->npairs) break;
        k1 = k0 + 16 < a->npairs ? k0 + 16 : a->npairs;
        for (long k = k0; k < k1; ++k) {
            unsigned i, j;
            pair_from_index(N, k, &i, &j);
            const double ea = a->eta[i], eb = a->eta[j];
            double kk[2], ke[2];
            int bad = 0;
            long n0 = oracle_kappa(p, 0, ea, eb, creal(a->omega), cimag(a->omega), a->recompute, kk);
            if (n0 < 0) bad = 1;
            local_int += n0 > 0 ? n0 : 0;
            if (a->counts) a->counts[(size_t)i * N + j] = n0;
            cplx k0v = c_mk(kk[0], kk[1]) + kappa_e(p, 0, ea, eb, a->omega, &bad);
            /* mat(i,j) = -kappa_all(0) * W(i,j) * dx, solver.h:448-451 */
            cplx v = c_scale(c_scale(c_neg(k0v), oracle_weight((int)N, (int)i, (int)j)), a->dx);
            a->M[i * dim + j] = v;
            a->M[j * dim + i] = v;
            if (!a->es) {
                long n1 = oracle_kappa(p, 1, ea, eb, creal(a->omega), cimag(a->omega), a->recompute, kk);
                cplx k1v = c_mk(kk[0], kk[1]) + kappa_e(p, 1, ea, eb, a->omega, &bad);
                long n2 = oracle_kappa(p, 2, ea, eb, creal(a->omega), cimag(a->omega), a->recompute, ke);
                cplx k2v = c_mk(ke[0], ke[1]) + kappa_e(p, 2, ea, eb, a->omega, &bad);
                if (n1 < 0 || n2 < 0) bad = 1;
                local_int += (n1 > 0 ? n1 : 0) + (n2 > 0 ? n2 : 0);
                cplx bb = c_scale(k1v, a->dx), dd = c_scale(k2v, a->dx);
                /* block scatter, solver.h:476-504 */
                a->M[i * dim + j + N] = bb;
                a->M[(i + N) * dim + j + N] = dd;
                a->M[j * dim + i + N] = c_neg(bb);
                a->M[(j + N) * dim + i + N] = dd;
                a->M[(i + N) * dim + j] = c_neg(bb);
                a->M[(j + N) * dim + i] = bb;
            }
            if (bad) a->error = 1;
        }
    }
    pthread_mutex_lock(&a->mu);
    a->intervals += local_int;
    pthread_mutex_unlock(&a->mu);
    return NULL;
}

int oracle_assemble(const emme_params_t* p, double wre, double wim, double* Mout, int nthreads,
                    int recompute, long* counts, long* total_intervals) {
    const unsigned N = (unsigned)p->npoints;
    if (N < 2) return -1;
    asm_t a;
    memset(&a, 0, sizeof a);
    a.p = p;
    a.omega = c_mk(wre, wim);
    a.M = (cplx*)Mout;
    a.counts = counts;
    a.N = N;
    a.es = fpclassify(p->beta_e) == FP_ZERO;
    a.dim = a.es ? N : 2 * (size_t)N;
    a.recompute = recompute;
    a.npairs = (long)N * (N - 1) / 2;
    double* eta = malloc(N * sizeof *eta);
    a.dx = oracle_grid(p->length, N, eta);
    a.eta = eta;
    pthread_mutex_init(&a.mu, NULL);
    if (counts) memset(counts, 0, (size_t)N * N * sizeof *counts);
    for (unsigned i = 0; i < N; ++i) {
        /* diagonal, solver.h:442-443, 465-470 */
        a.M[i * a.dim + i] = (1.0 + 1.0 / p->tau);
        if (!a.es) {
            a.M[i * a.dim + i + N] = 0.0;
            a.M[(i + N) * a.dim + i] = 0.0;
            a.M[(i + N) * a.dim + i + N] = (2.0 * p->tau) / p->beta_e * oracle_bi(p, eta[i]);
        }
    }
    if (nthreads < 1) nthreads = 1;
    pthread_t* th = malloc((size_t)nthreads * sizeof *th);
    for (int t = 1; t < nthreads; ++t) pthread_create(&th[t], NULL, asm_worker, &a);
    asm_worker(&a);
    for (int t = 1; t < nthreads; ++t) pthread_join(th[t], NULL);
    free(th);
    free(eta);
    pthread_mutex_destroy(&a.mu);
    if (total_intervals) *total_intervals = a.intervals;
    return a.error ? -2 : (int)a.dim;
}

/* ---- linear step: X = A^-1 B by partial-pivot LU, tr(X) ----------------------------- */
int oracle_trace_solve(int n, double* Ain, double* Bin, double* tr2) {
    /* Algorithm of src/solver.cpp:14-124 (pivot = argmax |.| down the column, full row
     * swap, factor = A(j,i)/A(i,i), row update), applied to the n right-hand sides of
     * include/solver.h:134-136; then dOmega = -1/trace (solver.h:139). */
    cplx* A = (cplx*)Ain;
    cplx* B = (cplx*)Bin;
    int info = 0;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = cabs(A[(size_t)k * n + k]);
        for (int r = k + 1; r < n; ++r) {
            double v = cabs(A[(size_t)r * n + k]);
            if (v > best) best = v, piv = r;
        }
        if (piv != k) {
            for (int c = 0; c < n; ++c) {
                cplx t = A[(size_t)k * n + c];
                A[(size_t)k * n + c] = A[(size_t)piv * n + c];
                A[(size_t)piv * n + c] = t;
                t = B[(size_t)k * n + c];
                B[(size_t)k * n + c] = B[(size_t)piv * n + c];
                B[(size_t)piv * n + c] = t;
            }
        }
        const cplx d = A[(size_t)k * n + k];
        if (d == 0.0) {
            if (!info) info = k + 1;
            continue;
        }
        for (int r = k + 1; r < n; ++r) {
            const cplx f = A[(size_t)r * n + k] / d;
            if (f == 0.0) continue;
            for (int c = k + 1; c < n; ++c) A[(size_t)r * n + c] -= f * A[(size_t)k * n + c];
            for (int c = 0; c < n; ++c) B[(size_t)r * n + c] -= f * B[(size_t)k * n + c];
        }
    }
    if (info) {
        tr2[0] = tr2[1] = NAN;
        return info;
    }
    /* back substitution, column by column; only the diagonal of X is summed */
    cplx tr = 0.0;
    cplx* x = malloc((size_t)n * sizeof *x);
    for (int c = 0; c < n; ++c) {
        for (int r = n - 1; r >= 0; --r) {
            cplx s = B[(size_t)r * n + c];
            for (int q = r + 1; q < n; ++q) s -= A[(size_t)r * n + q] * x[q];
            x[r] = s / A[(size_t)r * n + r];
            if (r == c) break; /* rows above c are not needed for X(c,c) */
        }
        tr += x[c];
    }
    free(x);
    tr2[0] = creal(tr), tr2[1] = cimag(tr);
    return 0;
}

/* ---- root search (src/main.cpp:19-80, include/solver.h:396-415, 113-160) ------------ */
int oracle_solve_root(const emme_params_t* p, double gre, double gim, int nthreads, int recompute,
                      double* root2, double* iterates, double* M_final, long* total_intervals) {
    const unsigned N = (unsigned)p->npoints;
    const size_t dim = fpclassify(p->beta_e) == FP_ZERO ? N : 2 * (size_t)N;
    const size_t nn = dim * dim;
    cplx* M = malloc(nn * sizeof *M);
    cplx* Mold = malloc(nn * sizeof *M);
    cplx* Mp = malloc(nn * sizeof *M);
    cplx* wa = malloc(nn * sizeof *M);
    long ints = 0, tot = 0;
    int rc = 0, it = 0;
    const cplx g = c_mk(gre, gim);
    /* ctor: omega = 0.99 g; d = 0.01 g; M_old = M(omega); omega += d; M = M(omega) */
    cplx omega = c_scale(g, 0.99);
    cplx domega = c_scale(g, 0.01);
    if (oracle_assemble(p, creal(omega), cimag(omega), (double*)Mold, nthreads, recompute, NULL, &ints) < 0) {
        rc = -2;
        goto done;
    }
    tot += ints;
    omega += domega;
    if (oracle_assemble(p, creal(omega), cimag(omega), (double*)M, nthreads, recompute, NULL, &ints) < 0) {
        rc = -2;
        goto done;
    }
    tot += ints;
    for (size_t k = 0; k < nn; ++k) Mp[k] = (M[k] - Mold[k]) / domega; /* solver.h:54-57 */
    for (int j = 0; j <= p->iteration_step_limit; ++j) {
        memcpy(Mold, M, nn * sizeof *M); /* solver.h:114 */
        memcpy(wa, M, nn * sizeof *M);
        double tr[2];
        int info = oracle_trace_solve((int)dim, (double*)wa, (double*)Mp, tr);
        domega = r_div(-1.0, c_mk(tr[0], tr[1])); /* solver.h:139 */
        omega += domega;
        if (info) {
            rc = -3;
            goto done;
        }
        if (oracle_assemble(p, creal(omega), cimag(omega), (double*)M, nthreads, recompute, NULL, &ints) < 0) {
            rc = -2;
            goto done;
        }
        tot += ints;
        for (size_t k = 0; k < nn; ++k) Mp[k] = (M[k] - Mold[k]) / domega;
        if (iterates) iterates[2 * it] = creal(omega), iterates[2 * it + 1] = cimag(omega);
        ++it;
        /* src/main.cpp:53-56 */
        if (cabs(domega) < cabs(c_scale(omega, p->iteration_precision))) break;
    }
    rc = it;
done:
    root2[0] = creal(omega), root2[1] = cimag(omega);
    if (M_final) memcpy(M_final, M, nn * sizeof *M);
    if (total_intervals) *total_intervals = tot;
    free(M), free(Mold), free(Mp), free(wa);
    return rc;
}
