// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" harness that is compiled TOGETHER WITH the untouched reference
// sources where they lie under /root/reference (see oracle/Makefile, target `ref`).
// Only the part of the reference that builds from its own files is used:
//   src/Parameters.cpp, src/JsonParser.cpp, src/functions.cpp,
//   src/singularity_handler.cpp + include/{Parameters,functions,Grid,Matrix}.h
// i.e. the integrand / quadrature / Bessel / weight / grid code.
// include/solver.h (matrixAssembler + Newton) is NOT built: it needs <lapack.h>, which
// this image lacks, and no stand-in header is written for it.  The (i,j) scatter below
// restates include/solver.h:439-511 around the reference's own kappa functions.
//
// Output goes to oracle/_ref/libemme_ref.so (git-ignored, travels with gpurun).
#include <complex>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <atomic>

#include "Grid.h"
#include "JsonParser.h"
#include "Matrix.h"
#include "Parameters.h"
#include "functions.h"
#include "singularity_handler.h"

namespace {
const Parameters* g_para = nullptr;
std::string g_err;
}

extern "C" {

const char* ref_last_error() { return g_err.c_str(); }

// Parameters::generate (src/Parameters.cpp:10-34) on a JSON text.
int ref_open(const char* json_text) {
    try {
        auto v = util::json::parse(std::string(json_text));
        g_para = &Parameters::generate(v);
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        g_para = nullptr;
        return -1;
    }
}

// Derived scalars (src/Parameters.cpp:36-66, 211-223) for cross-checking the host parser.
int ref_params(double* out, int n) {
    if (!g_para) return -1;
    const Parameters& p = *g_para;
    double v[] = {p.q, p.shat, p.tau, p.epsilon_n, p.epsilon_r, p.eta_i, p.eta_e, p.b_theta,
                  p.beta_e, p.R, p.vt, p.omega_d_coeff, p.length, p.theta, (double)p.npoints,
                  (double)p.iteration_step_limit, p.integration_precision,
                  p.integration_accuracy, (double)p.integration_iteration_limit,
                  (double)p.integration_start_points, p.arc_coeff, p.alpha, p.omega_s_i,
                  p.omega_s_e, p.omega_d_bar};
    int m = (int)(sizeof(v) / sizeof(v[0]));
    for (int i = 0; i < n && i < m; ++i) out[i] = v[i];
    return m;
}

double ref_g(double eta) { return g_para->g_integration_f(eta); }
double ref_bi(double eta) { return g_para->bi(eta); }
double ref_beta_1(double eta, double eta_p) { return g_para->beta_1(eta, eta_p); }
double ref_beta_1_e(double eta, double eta_p) { return g_para->beta_1_e(eta, eta_p); }

void ref_kappa(unsigned m, double eta, double eta_p, double wre, double wim, double* out) {
    auto r = g_para->kappa_f_tau(m, eta, eta_p, {wre, wim});
    out[0] = r.real();
    out[1] = r.imag();
}

void ref_kappa_e(unsigned m, double eta, double eta_p, double wre, double wim, double* out) {
    auto r = g_para->kappa_f_tau_e(m, eta, eta_p, {wre, wim});
    out[0] = r.real();
    out[1] = r.imag();
}

// util::bessel_i_alter_helper (include/functions.h:381-408): {y0,y1,mu+y0,-/+z}
void ref_bessel(double zre, double zim, double* out8) {
    auto a = util::bessel_i_alter_helper(std::complex<double>(zre, zim));
    for (int k = 0; k < 4; ++k) {
        out8[2 * k] = a[k].real();
        out8[2 * k + 1] = a[k].imag();
    }
}

// SingularityHandler (src/singularity_handler.cpp:3-24), row-major n*n doubles.
void ref_singularity(int n, double* out) {
    Matrix<double> w = SingularityHandler(n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) out[(size_t)i * n + j] = w(i, j);
}

// Grid<double> (include/Grid.h:7-20)
double ref_grid(double len, unsigned n, double* out) {
    Grid<double> g(len, n);
    for (unsigned i = 0; i < n; ++i) out[i] = g.grid[i];
    return g.dx;
}

// Generic adaptive quadrature front end (include/functions.h:305-331) on a test
// integrand f(t) = exp((ar+i*ai)*t) * t^p, used to pin the GK restatement on its own.
void ref_integrate_test(double ar, double ai, double p, double tol, double prec,
                        unsigned long max_sub, unsigned long pts, double* out) {
    auto f = [&](double t) {
        return std::exp(std::complex<double>(ar, ai) * t) * std::pow(t, p);
    };
    auto r = util::integrate(f, tol, prec, max_sub, pts);
    out[0] = r.real();
    out[1] = r.imag();
}

// (i,j) scatter of include/solver.h:439-511 around the reference kappa functions.
// M is dim*dim complex row-major (dim = N or 2N), written as interleaved doubles.
int ref_assemble(double wre, double wim, double* Mout, int nthreads) {
    if (!g_para) return -1;
    const Parameters& para = *g_para;
    const unsigned N = para.npoints;
    Grid<double> grid(para.length, N);
    Matrix<double> W = SingularityHandler(N);
    const bool es = std::fpclassify(para.beta_e) == FP_ZERO;
    const size_t dim = es ? N : 2 * N;
    auto* M = reinterpret_cast<std::complex<double>*>(Mout);
    const std::complex<double> omega(wre, wim);
    auto kall = [&](unsigned m, double a, double b) {
        return para.kappa_f_tau(m, a, b, omega) + para.kappa_f_tau_e(m, a, b, omega);
    };
    std::vector<std::pair<unsigned, unsigned>> pairs;
    for (unsigned i = 0; i < N; ++i) {
        M[i * dim + i] = 1.0 + 1.0 / para.tau;
        if (!es) {
            M[i * dim + i + N] = 0.0;
            M[(i + N) * dim + i] = 0.0;
            M[(i + N) * dim + i + N] = (2.0 * para.tau) / para.beta_e * para.bi(grid.grid[i]);
        }
        for (unsigned j = i + 1; j < N; ++j) pairs.emplace_back(i, j);
    }
    std::atomic<size_t> next{0};
    std::string err;
    std::atomic<bool> failed{false};
    auto work = [&]() {
        try {
            for (;;) {
                size_t k = next.fetch_add(1);
                if (k >= pairs.size()) break;
                auto [i, j] = pairs[k];
                const double a = grid.grid[i], b = grid.grid[j];
                auto v = -kall(0, a, b) * W(i, j) * grid.dx;
                M[i * dim + j] = v;
                M[j * dim + i] = v;
                if (!es) {
                    auto bb = kall(1, a, b) * grid.dx;
                    auto dd = kall(2, a, b) * grid.dx;
                    M[i * dim + j + N] = bb;
                    M[(i + N) * dim + j + N] = dd;
                    M[j * dim + i + N] = -bb;
                    M[(j + N) * dim + i + N] = dd;
                    M[(i + N) * dim + j] = -bb;
                    M[(j + N) * dim + i] = bb;
                }
            }
        } catch (const std::exception& e) {
            if (!failed.exchange(true)) err = e.what();
        }
    };
    if (nthreads < 1) nthreads = 1;
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (failed) {
        g_err = err;
        return -2;
    }
    return (int)dim;
}

}  // extern "C"
