/* emme_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the reference's dispersion-matrix assembly and
 * trace-secant Newton search.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (emme_amd/) never does.
 *
 * Pinning: every function here is checked against oracle/_ref/libemme_ref.so (the
 * reference's own kappa/quadrature sources compiled unmodified) and against the
 * golden values of SURVEY.md App. B in tests/test_oracle_vs_reference.py and
 * tests/test_golden.py.  The reference's own tests hold no fixture for this path
 * (SURVEY §4), and the LAPACK routine behind its Newton step (zsysv, unpinned module)
 * is cross-checked through SciPy's bundled OpenBLAS in tests/.
 */
#ifndef EMME_ORACLE_H
#define EMME_ORACLE_H

#include "../include/emme_params.h"

#ifdef __cplusplus
extern "C" {
#endif

/* src/Parameters.cpp:36-66, 211-223 (+ Cylinder ctor :395-398) */
void oracle_params_derive(emme_params_t* p);

/* include/Grid.h:7-20; returns dx */
double oracle_grid(double len, unsigned n, double* eta_out);

/* src/singularity_handler.cpp:3-24: weight W(i,j) */
double oracle_weight(int n, int i, int j);

/* g_integration_f / bi, src/Parameters.cpp:76-85, 97-100, 225-232, 248-393, 400-440 */
double oracle_g(const emme_params_t* p, double eta);
double oracle_bi(const emme_params_t* p, double eta);

/* include/functions.h:381-408; out8 = {y0, y1, mu+y0, (Re z<0 ? z : -z)} */
void oracle_bessel(double zre, double zim, double* out8);

/* src/Parameters.cpp:113-184; out2 = kappa; returns number of GK intervals evaluated.
 * recompute != 0 re-evaluates g()/bi() inside every integrand call like the reference
 * does (identical values, reference-like cost); 0 uses the per-pair hoist. */
long oracle_kappa(const emme_params_t* p, unsigned m, double eta, double eta_p, double wre,
                  double wim, int recompute, double* out2);

/* src/Parameters.cpp:186-209 */
int oracle_kappa_e(const emme_params_t* p, unsigned m, double eta, double eta_p, double wre,
                   double wim, double* out2);

/* include/functions.h:305-331 on f(t) = exp((ar+i ai) t) t^p -- pins the quadrature alone */
long oracle_integrate_test(double ar, double ai, double pw, double tol, double prec,
                           unsigned long max_sub, unsigned long pts, double* out2);

/* include/solver.h:417-515.  M: dim*dim complex row-major (interleaved re,im),
 * dim = N (beta_e == 0) or 2N.  counts (optional, N*N longs) receives GK interval
 * counts of the m=0 integral per (i<j) pair.  Returns dim, or <0 on error. */
int oracle_assemble(const emme_params_t* p, double wre, double wim, double* M, int nthreads,
                    int recompute, long* counts, long* total_intervals);

/* One trace-secant linear step (include/solver.h:113-140): X = M^-1 M', returns
 * tr(X) in tr2 via partial-pivot LU (algorithm of the reference's dead code
 * src/solver.cpp:14-124, standing in for LAPACK zsysv).  A and B (n*n complex,
 * row-major) are destroyed.  Returns 0, or k>0 if U(k,k) == 0 (LAPACK convention). */
int oracle_trace_solve(int n, double* A, double* B, double* tr2);

/* Full root search of src/main.cpp:19-80 + include/solver.h:396-415,113-160
 * (TraceSecant).  iterates (optional): 2*(limit+2) doubles; returns number of Newton
 * iterations done (>=1), or <0 on error.  root2 = final omega. M_final (optional). */
int oracle_solve_root(const emme_params_t* p, double gre, double gim, int nthreads,
                      int recompute, double* root2, double* iterates, double* M_final,
                      long* total_intervals);

/* diagnostic: record the intervals the calling thread's integrals evaluate (depth << 56 | index,
 * evaluation order) into buf; buf = NULL switches it off.  oracle_trace_count = intervals seen. */
void oracle_trace_set(long* buf, long cap);
long oracle_trace_count(void);

/* diagnostic histogram of evaluated intervals by bisection depth (single-threaded runs) */
void oracle_depth_hist(long* out64, int reset);

#ifdef __cplusplus
}
#endif
#endif
