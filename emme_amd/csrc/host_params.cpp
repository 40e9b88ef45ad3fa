// host_params.cpp -- host side of the parameter model: JSON -> emme_params_t, derived
// scalars, and the per-grid-point tables the device kernels read.
//
// Follows reference src/Parameters.cpp:10-66 (generate + ctor), :76-100 (tokamak g, bi),
// :211-232 (stellarator ctor, sigma, bi), :248-393 (stellarator g), :395-440 (other
// variants), include/Grid.h:7-20 and src/singularity_handler.cpp:3-24.  The tables
// g_i = g(eta_i), b_i = b(eta_i) replace the reference's 4x / 2x re-evaluation per
// integrand call: same values, computed once.
#include <cmath>
#include <cstring>
#include <string>

#include "../../include/emme_hip.h"
#include "host_json.hpp"

namespace emme {
void set_error(const std::string& msg);  // emme_capi.hip
int params_from_value(const JsonValue& root, emme_params_t* out);
}

namespace {

using emme::JsonValue;

// A scan object {head, step, tail} stands for its head (reference src/main.cpp:174-180).
const JsonValue& scalar_of(const JsonValue& root, const char* key) {
    const JsonValue& v = root.at(key);
    return v.is_object() ? v.at("head") : v;
}
double num(const JsonValue& root, const char* key) { return scalar_of(root, key).number(); }

double first_zero_of_drift_potential(double a) {
    // bisection of f(x) = cos x + a x sin x on [0, pi]; reference src/functions.cpp:32-65
    // (tolerance 1e-9, <= 100 halvings: include/functions.h:496-498)
    auto f = [a](double x) { return std::cos(x) + a * x * std::sin(x); };
    const double tol = 1e-9;
    double lo = 0.0, hi = M_PI, mid = 0.0;
    if (std::fabs(f(lo)) < tol) return lo;
    if (std::fabs(f(hi)) < tol) return hi;
    for (int it = 0; it < 100; ++it) {
        mid = lo + (hi - lo) / 2.0;
        const double fm = f(mid);
        if (std::fabs(fm) < tol || (hi - lo) / 2.0 < tol) return mid;
        if (f(lo) * fm < 0)
            hi = mid;
        else
            lo = mid;
    }
    return mid;
}

double g_of_eta(const emme_params_t& p, double eta) {
    switch (p.conf) {
        case EMME_CONF_TOKAMAK: {
            // src/Parameters.cpp:76-85.  The exponent `3 / 2` there is integer 1.
            const double last = (1 - p.shat) * p.q * p.epsilon_r /
                                std::pow((std::pow(p.epsilon_r, 2) + std::pow(p.q, 2)), 1) * eta;
            return -((p.alpha * eta) / 2.0) + p.shat * p.theta * std::cos(eta) -
                   p.shat * eta * std::cos(eta) + std::sin(eta) + p.shat * std::sin(eta) +
                   0.25 * p.alpha * std::sin(2.0 * eta) - last;
        }
        case EMME_CONF_STELLARATOR: {
            // src/Parameters.cpp:248-393 in closed form: with L = lh - mh q and
            // S = deltap + rdeltapp + deltap shat every polynomial coefficient of the
            // expanded expression factors through 2 (L-1) L^2 (L+1), which cancels.
            const double lh = p.lh, L = lh - p.mh * p.q;
            const double S = p.deltap + p.rdeltapp + p.deltap * p.shat;
            const double eh = p.epsilon_h_t, de = eta - p.eta_k;
            const double phi = eta * L - p.alpha_0 * p.mh;
            double g = 0.5 * eta * (S + p.curvature_aver * p.R);
            g += -p.shat * de * std::cos(eta) + (1.0 + p.shat) * std::sin(eta) -
                 0.25 * S * std::sin(2.0 * eta);
            g += -eh * p.shat * lh * de * std::cos(phi) / L +
                 eh * lh * (L + p.shat) * std::sin(phi) / (L * L);
            g += -0.5 * S * eh * lh *
                 (std::sin(eta + phi) / (L + 1.0) + std::sin(eta - phi) / (L - 1.0));
            return g;
        }
        case EMME_CONF_CYLINDER: return eta * p.shat_coeff;  // :400-402
        case EMME_CONF_TAYLOR_MD: {
            // src/Parameters.cpp:404-436, Pade approximant of order {3,4}
            const double a = p.alpha, s = p.shat;
            const double den =
                7 + 16 * a + 40 * std::pow(a, 2) - 28 * s - 80 * a * s + 40 * std::pow(s, 2);
            const double c3 = -31 - 96 * a - 168 * std::pow(a, 2) - 560 * std::pow(a, 3) + 186 * s +
                              672 * a * s + 1680 * std::pow(a, 2) * s - 504 * std::pow(s, 2) -
                              1680 * a * std::pow(s, 2) + 560 * std::pow(s, 3);
            const double d2 =
                3 + 19 * a + 56 * std::pow(a, 2) - 18 * s - 84 * a * s + 28 * std::pow(s, 2);
            const double d4 =
                11 - 4 * a + 704 * std::pow(a, 2) - 88 * s - 584 * a * s + 216 * std::pow(s, 2);
            return (eta + (std::pow(eta, 3) * c3) / (42. * den)) /
                   (1 + (std::pow(eta, 2) * d2) / (7. * den) +
                    (std::pow(eta, 4) * d4) / (840. * den));
        }
        default: return eta;  // "cylinder old", :438-440
    }
}

double b_of_eta(const emme_params_t& p, double eta) {
    if (p.conf == EMME_CONF_STELLARATOR) {
        // src/Parameters.cpp:225-232
        const double sigma =
            p.shat * (eta - p.eta_k) + (p.deltap * (1 + p.shat) + p.rdeltapp) * std::sin(eta);
        return p.b_theta * (1.0 + std::pow(sigma, 2));
    }
    // src/Parameters.cpp:97-100
    return p.b_theta * (1.0 + std::pow(p.shat * (eta - p.theta) - p.alpha * std::sin(eta), 2));
}

}  // namespace

extern "C" {

int emme_params_sizeof(void) { return (int)sizeof(emme_params_t); }

int emme_params_derive(emme_params_t* p) {
    if (!p) return EMME_EINVAL;
    // src/Parameters.cpp:44, 58-64
    p->b_theta = p->k_rho * p->k_rho;
    p->alpha = p->q * p->q * p->R * p->beta_e / (p->epsilon_n * p->R) *
               ((1 + p->eta_e) + 1 / p->tau * (1 + p->eta_i));
    p->omega_s_i = -(std::sqrt(p->b_theta) * p->vt) / (p->epsilon_n * p->R);
    p->omega_s_e = -p->tau * p->omega_s_i;
    p->omega_d_bar = 2.0 * p->epsilon_n * p->omega_s_i * p->omega_d_coeff;
    p->deltap = p->beta_e_p = p->rdeltapp = p->curvature_aver = p->shat_coeff = 0.0;
    if (p->conf == EMME_CONF_STELLARATOR) {
        // src/Parameters.cpp:219-223; `mh / lh` divides two ints there
        if (p->lh == 0) {
            emme::set_error("stellarator: lh must be non-zero");
            return EMME_ECONFIG;
        }
        p->deltap = -0.25 * p->alpha;
        p->beta_e_p = p->beta_e * (1.0 + p->eta_e) / (p->epsilon_n * p->R);
        p->rdeltapp = (-p->alpha + (2.0 * p->shat - 3) * p->deltap);
        p->curvature_aver = (p->mh / p->lh) * p->r_over_R / (p->q * p->R) * (4.0 - p->shat) +
                            (-p->alpha + 2 * p->shat * p->deltap + 0) / p->R;
    } else if (p->conf == EMME_CONF_CYLINDER) {
        // src/Parameters.cpp:395-398 -> src/functions.cpp:72-83
        const double x0 = first_zero_of_drift_potential(p->shat);
        p->shat_coeff = ((1.0 + p->shat) * std::sin(x0) - p->shat * x0 * std::cos(x0)) / x0;
    }
    return EMME_OK;
}

int emme_params_from_json(const char* json_text, emme_params_t* out) {
    if (!json_text || !out) return EMME_EINVAL;
    try {
        const JsonValue root = emme::json_parse(json_text);
        return emme::params_from_value(root, out);
    } catch (const std::exception& e) {
        emme::set_error(e.what());
        return EMME_EJSON;
    }
}

}  // extern "C"

namespace emme {
// Parameters::generate on an already parsed document (used by the scan driver, which edits
// the document between solves)
int params_from_value(const JsonValue& root, emme_params_t* out) {
    try {
        if (!root.is_object()) throw std::runtime_error("top-level JSON value must be an object");
        emme_params_t p;
        std::memset(&p, 0, sizeof p);
        // key order = order in which the reference touches them, so the first missing
        // key reported is the same one (src/main.cpp:23, src/Parameters.cpp:18-66,213-218,
        // src/main.cpp:41)
        p.iteration_precision = num(root, "iteration_precision");
        const std::string& conf = root.at("conf").str();
        if (conf == "tokamak")
            p.conf = EMME_CONF_TOKAMAK;
        else if (conf == "stellarator")
            p.conf = EMME_CONF_STELLARATOR;
        else if (conf == "cylinder")
            p.conf = EMME_CONF_CYLINDER;
        else if (conf == "taloyMagneticDrift")
            p.conf = EMME_CONF_TAYLOR_MD;
        else if (conf == "cylinder old")
            p.conf = EMME_CONF_CYLINDER_OLD;
        else
            throw std::runtime_error("Input configuration not supported yet.");
        p.q = num(root, "q");
        p.shat = num(root, "shat");
        p.tau = num(root, "tau");
        p.epsilon_n = num(root, "epsilon_n");
        p.epsilon_r = num(root, "epsilon_r");
        p.eta_i = num(root, "eta_i");
        p.eta_e = num(root, "eta_e");
        p.k_rho = num(root, "k_rho");
        p.beta_e = num(root, "beta_e");
        p.R = num(root, "R");
        p.vt = num(root, "vt");
        p.omega_d_coeff = num(root, "omega_d_coeff");
        p.length = num(root, "length");
        p.theta = num(root, "theta");
        p.npoints = (int)num(root, "npoints");
        p.iteration_step_limit = (int)num(root, "iteration_step_limit");
        p.integration_precision = num(root, "integration_precision");
        p.integration_accuracy = num(root, "integration_accuracy");
        p.integration_iteration_limit = (int)num(root, "integration_iteration_limit");
        p.integration_start_points = (int)num(root, "integration_start_points");
        p.arc_coeff = num(root, "arc_coeff");
        p.water_bag_weight_vpara = num(root, "water_bag_weight_vpara");
        p.water_bag_weight_vperp = num(root, "water_bag_weight_vperp");
        p.drift_center_transformation_switch =
            scalar_of(root, "drift_center_transformation_switch").boolean() ? 1 : 0;
        if (p.conf == EMME_CONF_STELLARATOR) {
            p.eta_k = num(root, "eta_k");
            p.lh = (int)num(root, "lh");
            p.mh = (int)num(root, "mh");
            p.epsilon_h_t = num(root, "epsilon_h_t");
            p.alpha_0 = num(root, "alpha_0");
            p.r_over_R = num(root, "r_over_R");
        }
        p.iteration_method = root.at("iteration_method").str() == "TraceSecant"
                                 ? EMME_METHOD_TRACE_SECANT
                                 : EMME_METHOD_QR_SECANT;
        if (root.has("initial_guess")) {
            const JsonValue& g = root.at("initial_guess");
            p.initial_guess[0] = g.at((size_t)0).number();
            p.initial_guess[1] = g.at((size_t)1).number();
        }
        int rc = emme_params_derive(&p);
        if (rc) return rc;
        *out = p;
        return EMME_OK;
    } catch (const std::exception& e) {
        emme::set_error(e.what());
        return EMME_EJSON;
    }
}
}  // namespace emme

extern "C" {

double emme_weight(int n, int i, int j) {
    // src/singularity_handler.cpp:4-20: end-corrected weights near the diagonal, 1
    // elsewhere, minus one half on the first and last column
    static const double near_diag[6] = {0.0,
                                        2.951388888888883,
                                        -2.4305555555555305,
                                        4.166666666667441,
                                        -0.3472222222224549,
                                        1.159722222222284};
    const int d = i > j ? i - j : j - i;
    double w = d <= 5 ? near_diag[d] : 1.0;
    if (j == 0 || j == n - 1) w -= 0.5;
    return w;
}

int emme_tables(const emme_params_t* p, double* eta, double* g, double* b, double* dx_out) {
    if (!p || p->npoints < 2) return EMME_EINVAL;
    const unsigned n = (unsigned)p->npoints;
    const double dx = (2 * p->length) / (n - 1);  // include/Grid.h:11
    for (unsigned i = 0; i < n; ++i) {
        const double e = -p->length + i * dx;     // include/Grid.h:13
        if (eta) eta[i] = e;
        if (g) g[i] = g_of_eta(*p, e);
        if (b) b[i] = b_of_eta(*p, e);
    }
    if (dx_out) *dx_out = dx;
    return EMME_OK;
}

}  // extern "C"
