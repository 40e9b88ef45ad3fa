/* emme_hip.h -- C ABI of the MI355X (gfx950) dispersion-matrix assembly + eigenvalue
 * search.  Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * The reference (ssskkkky/EMME) has no FFI; these entry points sit behind its two
 * de-facto operator seams and its solve-once driver:
 *
 *   emme_params_from_json   <- util::json::parse + Parameters::generate
 *                              (reference src/JsonParser.cpp:655-669, src/Parameters.cpp:10-66)
 *   emme_assemble_batch     <- EigenSolver<T>::matrixAssembler(matrix_type&)
 *                              (reference include/solver.h:417-515), batched over omega
 *   emme_newton_step_batch  <- EigenSolver<T>::newtonTraceSecantIteration()
 *                              (reference include/solver.h:113-160; LAPACK_zsysv call :134-136)
 *   emme_solve_roots        <- solve_once_eigen loop (reference src/main.cpp:19-80) +
 *                              EigenSolver ctor (include/solver.h:396-415), batched over guesses
 *
 * Conventions: every function returns 0 on success or a negative EMME_E* code (never
 * throws); emme_last_error() gives the text for the calling thread.  Per-item `info`
 * follows LAPACK: 0 ok, k>0 = factor U(k,k) exactly zero (include/solver.h:142-153).
 * Complex numbers are interleaved (re,im) doubles == std::complex<double> == the layout
 * of the reference's Matrix<std::complex<double>> (row-major, include/Matrix.h:43).
 * Buffers are caller-owned and may be host or device pointers (detected with
 * hipPointerGetAttributes); the context owns all device scratch.  Calls on one context
 * must be serialised by the caller; different contexts are independent.
 */
#ifndef EMME_HIP_H
#define EMME_HIP_H

#include <stddef.h>

#include "emme_params.h"

#ifdef __cplusplus
extern "C" {
#endif

#define EMME_OK 0
#define EMME_EINVAL (-1)   /* bad argument                                        */
#define EMME_EJSON (-2)    /* JSON syntax error / missing key / wrong type        */
#define EMME_EDEVICE (-3)  /* no gfx950 device / HIP runtime failure              */
#define EMME_ENOMEM (-4)   /* device or host allocation failed                    */
#define EMME_ECONFIG (-5)  /* unsupported configuration (e.g. start points != 15|31) */
#define EMME_ENUMERIC (-6) /* quadrature depth cap hit or non-finite result       */

typedef struct emme_ctx emme_ctx_t;

/* Per-kernel device timing, filled when profiling is enabled (hipEvents recorded on the
 * context's stream around every launch). */
typedef struct emme_profile {
    double assemble_ms;       /* total time in the main fill kernel (cached / omega-lane /
                                 lanes-are-nodes, whichever the context dispatches)     */
    long assemble_launches;
    double deferred_ms;       /* fill of integrals the cached kernel deferred (work list) */
    long deferred_launches;
    double linstep_ms;        /* total time in the LU + trace kernel                */
    long linstep_launches;
    double other_ms;          /* copies / elementwise kernels                       */
    long other_launches;
    long long gk_intervals;   /* Gauss-Kronrod intervals evaluated (all launches)    */
    long long integrand_evals; /* = intervals * integration_start_points             */
    long long matrices;       /* matrices assembled                                  */
    long long union_rounds;   /* omega-lane kernel: interval rounds walked by lane groups */
    double cache_build_ms;    /* k_node_cache launches (node-record cache, once per context) */
    long cache_build_launches;
    double cache_alloc_ms;    /* host wall time spent allocating the cache buffers (hipMalloc) */
    /* dense (matrix-core) fill: interval rounds of the (16 pairs x 16 omegas) tile tasks */
    long long dense_rounds;   /* served by 48 v_mfma_f64_16x16x4_f64 each                 */
    long long sparse_rounds;  /* served on the vector ALU, one omega column at a time      */
    long long sparse_columns;
    long long tile_tasks;
    double nullspace_ms;      /* nullSpace: factorisation + inverse iteration launches (version 3) */
    long nullspace_launches;
} emme_profile_t;

/* Per-context options (version 3).  emme_options_default() fills in what emme_ctx_create uses; a caller
 * changes what it needs and passes the struct to emme_ctx_create_ex / emme_ctx_set_options.  The EMME_*
 * environment variables of earlier versions remain as DEVELOPER overrides only: they are read once, when a
 * context is created, and win over the struct (DESIGN.md appendix lists them). */
#define EMME_FILL_AUTO 0   /* dense (matrix-core) fill where it applies, else union walk / independent lanes */
#define EMME_FILL_UNION 1  /* never the dense fill: 48-byte records, union-walk kernel (electrostatic GK15)   */
#define EMME_FILL_LANES 2  /* independent lanes on 48-byte records                                            */
typedef struct emme_options {
    int size;                 /* sizeof(emme_options_t) of the caller (set by emme_options_default)      */
    /* ---- HBM node cache ---- */
    double node_cache_gb;     /* budget, both contour classes together (176); 0: never build one         */
    int cache_min_batch;      /* build the cache only for calls with at least this many omegas (8)       */
    int cache_min_depth;      /* shallowest full tree worth caching; 0 = 6 (tiled layout) / 5            */
    /* ---- fill routing (layout options are fixed once the context exists) ---- */
    int fill;                 /* EMME_FILL_*                                                             */
    int phase_table;          /* 1: folded records + per-launch phase table; 0: unfolded records         */
    int em_shared;            /* 1: electromagnetic contexts share one record per pair between moments   */
    int wl_min;               /* smallest uncached batch that takes the omega-lane kernel (4)            */
    int union_sel;            /* union walk: intervals a lane group serves per round (1, 2, 4)           */
    int union_ipg_few, union_few_chunks;  /* union walk: items per lane group in launches of few chunks  */
    int coop_wide_min;        /* deferred-list length from which the one-wave cooperative kernel takes over; -1 never */
    int defer_one_group;      /* 1: deferred integrals by one lane group each                            */
    int dense_min_cols;       /* dense fill: omega columns that must need an interval for the MFMA path   */
    int dense_min_tasks;      /* dense fill: chunk capacity is halved while a launch has fewer tile tasks */
    double dense_cost_ratio;  /* dense fill: a chunk ends where an omega costs less than 1/ratio of its first */
    int dense_wide;           /* dense fill, level lists of 128 instead of 64 entries: 0 = for the omegas of a root search
                                 whose lists overflowed in their previous fill, 1 = always (tests)                */
    int skip_lost;            /* 1: integrals of a matrix that already holds a non-finite entry are skipped */
    /* ---- Newton linear step ---- */
    int lu_split;             /* workgroups per matrix: 0 = by live matrices and order, k = k             */
    int lu_group_min_n;       /* smallest order that takes the grouped trailing updates (256); -1 never   */
    int lu_spin_limit;        /* polls before a hand-over wait of the multi-workgroup LU gives up         */
    int lu_unblocked;         /* 1: the unblocked LU                                                      */
} emme_options_t;

const char* emme_last_error(void);
int emme_params_sizeof(void);
int emme_version(void); /* 3: emme_options_t, emme_ctx_create_ex / _set_options / _get_options, emme_null_vectors_batch,
                           emme_profile_t grew nullspace_*; 2: cache_* fields, emme_comm_*, emme_gather_roots */
void emme_options_default(emme_options_t* opt);

/* JSON text -> raw + derived parameters.  Reproduces the reference parser's grammar
 * (a number token is a float only if it contains '.', else atoi; no string escapes)
 * and its "Failed to accessing key: <k>" errors.  Scan objects {head,step,tail} are
 * replaced by their head (reference src/main.cpp:174-180). */
int emme_params_from_json(const char* json_text, emme_params_t* out);
/* Fill derived members from raw ones (reference src/Parameters.cpp:36-66, 211-223). */
int emme_params_derive(emme_params_t* p);

/* Host-side tables exactly as the device uses them (for tests / tooling):
 * eta[N], g[N] = g_integration_f(eta), b[N] = bi(eta); returns dx in *dx. */
int emme_tables(const emme_params_t* p, double* eta, double* g, double* b, double* dx);
/* SingularityHandler weight W(i,j) (reference src/singularity_handler.cpp:3-24). */
double emme_weight(int n, int i, int j);

/* The Bessel helper alone, evaluated on the device exactly as the fill kernels do (tests and
 * tooling): util::bessel_i_alter_helper (reference include/functions.h:381-408) for n complex
 * arguments z (2n doubles, host); out: 8n doubles = {y0, y1, mu + y0, Re z < 0 ? z : -z} each. */
int emme_bessel_batch(const double* z, int n, double* out);

/* One context per (device, parameter set). device < 0 => current device. */
int emme_ctx_create(const emme_params_t* p, int device, emme_ctx_t** out);
/* The same with options (NULL = defaults).  EMME_EINVAL for a struct of the wrong size or values out of range. */
int emme_ctx_create_ex(const emme_params_t* p, int device, const emme_options_t* opt, emme_ctx_t** out);
/* Change the options of a live context.  Everything takes effect from the next call on, except what fixes the
 * layout of the node cache (fill, phase_table, em_shared): EMME_EINVAL if those differ once a cache exists. */
int emme_ctx_set_options(emme_ctx_t* ctx, const emme_options_t* opt);
int emme_ctx_get_options(const emme_ctx_t* ctx, emme_options_t* opt);
void emme_ctx_destroy(emme_ctx_t* ctx);
/* The node-cache buffers of destroyed contexts (up to ~170 GB) are kept in a process-wide pool
 * and reused by the next context (allocating them costs seconds, a parameter sweep creates one
 * context per parameter set); this returns them to the driver. */
void emme_release_pooled_memory(void);
/* Launch everything on this hipStream_t (e.g. torch's current stream). NULL = default. */
int emme_ctx_set_stream(emme_ctx_t* ctx, void* hip_stream);
int emme_ctx_dim(const emme_ctx_t* ctx); /* N if beta_e == 0 else 2N */
/* Kernel family used by the last fill: 0 lanes-are-nodes, 1 omega-lane, 2 HBM node cache
 * (independent lanes), 3 HBM node cache + phase table, union walk, 4 HBM node cache (tiled) +
 * dense fill on the FP64 matrix cores. */
int emme_ctx_fill_mode(const emme_ctx_t* ctx);
/* GiB of HBM currently held by the node-record cache (0 if none). */
double emme_ctx_node_cache_gib(const emme_ctx_t* ctx);
/* The node cache grows at run time: a fill that had to hand integrals to the from-scratch kernel
 * (their trees leave the cached intervals) makes the next fill cache a subtree around the interval
 * they were missing.  An integral that moves from one kernel to the other keeps its interval count
 * but changes its rounding (summation order, folded amplitudes: ~1e-16 relative, 1e-12 at strongly
 * damped omega), so a root search on a fresh context and on a warm one differ at that level.  The
 * CANONICAL state is the settled one: emme_ctx_cache_settle fills M(omega_b) for the given omegas
 * (host, 2*nbatch doubles; results discarded) until a fill leaves the cache shape unchanged; after
 * it, fills of omegas in the same region are bit-for-bit repeatable.  fills_done (nullable): fills run. */
int emme_ctx_cache_settle(emme_ctx_t* ctx, const double* omega, int nbatch, int* fills_done);
/* Shape of the cache: depth of the fully cached tree (-1 none yet, -2 disabled / does not fit),
 * number of cached subtrees (the fixed one included), GiB held.  Any pointer may be NULL. */
int emme_ctx_cache_state(const emme_ctx_t* ctx, int* full_depth, int* subtrees, double* gib);
int emme_ctx_profile_enable(emme_ctx_t* ctx, int on);
int emme_ctx_profile_read(emme_ctx_t* ctx, emme_profile_t* out, int reset);

/* Fill M(omega_b) for b < nbatch.  omega: 2*nbatch doubles (host).  M: nbatch*dim*dim
 * complex, row-major, host or device.  intervals (optional, host, nbatch long long):
 * GK intervals evaluated per item. */
int emme_assemble_batch(emme_ctx_t* ctx, const double* omega, int nbatch, double* M,
                        long long* intervals);

/* One trace-secant Newton step per item, in place (all arrays host or device, but
 * consistently one of the two):
 *   in : omega[b], M[b] = M(omega[b]), Mp[b] = M'(omega[b])
 *   out: domega[b] = -1/tr(M^-1 M'), omega[b] += domega[b], M[b] = M(new omega),
 *        Mp[b] = (M_new - M_old)/domega, info[b] (LAPACK convention).
 * method: EMME_METHOD_TRACE_SECANT as above; EMME_METHOD_QR_SECANT replaces the step by the
 * reference's newtonQRSecantIteration (include/solver.h:210-383): domega = -R_nn / (Q^H M' v)_n
 * from the column-pivoted QR of M (info[b] = k > 0: R(k,k) == 0, the ztrtrs failure). */
int emme_newton_step_batch(emme_ctx_t* ctx, double* omega, double* domega, int nbatch,
                           double* M, double* Mp, int method, int* info);

/* tr(A_b^-1 B_b) for b < nbatch by partial-pivot LU (A, B: nbatch*n*n complex, destroyed;
 * host or device).  tr: 2*nbatch doubles (host). info: nbatch ints (host): 0, k > 0 = U(k,k)
 * exactly zero (tr[b] = NaN), or EMME_EDEVICE if the cooperating workgroups of that matrix
 * gave up waiting for each other (only when something else holds compute units for seconds;
 * emme_solve_roots repairs that case itself, the step-level calls report it). */
int emme_trace_solve_batch(emme_ctx_t* ctx, int n, int nbatch, double* A, double* B, double* tr,
                           int* info);

/* The QR-secant quotient alone, for any square A_b, B_b (nbatch*n*n complex, row-major, host or
 * device, not modified; n <= 1024): with A P = Q R (Householder QR, LAPACK zgeqp3 pivoting),
 * R11 y = r12, v = P [-y; 1], t = Q^H (B v):  q[b] = t_n / R_nn, so that the reference's step is
 * domega = -1/q (include/solver.h:370).  q: 2*nbatch doubles (host); info: nbatch ints (host). */
int emme_qr_secant_batch(emme_ctx_t* ctx, int n, int nbatch, const double* A, const double* B,
                         double* q, int* info);

/* Batched root search: for each guess g_b run the reference's solve-once sequence
 * (omega=0.99 g; M_old=M(omega); omega+=0.01 g; M; M'; then up to step_limit+1 Newton
 * steps, stopping when |domega| < tol*|omega|).  The step is the one the context's
 * iteration_method names (trace-secant or QR-secant, src/main.cpp:45-49).  All arrays host.
 * roots: 2n doubles; iters: n ints (Newton steps done); info: n ints (0 ok, k>0 singular
 * pivot k, EMME_ENUMERIC if that chain met a non-finite integral / the depth cap).
 * iterates (optional): n*(step_limit+1)*2 doubles, omega after every step, NaN padded. */
int emme_solve_roots(emme_ctx_t* ctx, const double* guesses, int n, double tol, int step_limit,
                     double* roots, int* iters, int* info, double* iterates);
/* Copy M(omega_final) of item b of the last emme_solve_roots call (dim*dim complex). */
int emme_ctx_get_matrix(emme_ctx_t* ctx, int b, double* M_host);

/* nullSpace (reference include/solver.h:58-112): the right singular vector of the smallest
 * singular value of the n x n complex matrix M (row-major), by inverse iteration on M^H M.
 * Same vector as the reference's SVD result up to the arbitrary complex phase. Host pointers. */
int emme_null_vector(const double* M, int n, double* vec /* 2n doubles */);
/* The same on the device, batched (nullspace.hip): ONE partial-pivot LU per matrix (the Newton step's kernels,
 * no right-hand sides) and inverse iteration v <- M^-1 (M^-H v) on its factors until the direction stops
 * moving.  M: nbatch*n*n complex, row-major, host or device, not modified; NULL = the matrices M(omega_final)
 * of the last emme_solve_roots call on this context (then n = emme_ctx_dim, nbatch <= that call's n).
 * vecs: nbatch*n complex (host), unit 2-norm, arbitrary phase.  info (host, nbatch): 0, k > 0 = column k of the
 * factorisation is exactly zero (no vector: NaN), EMME_ENUMERIC = non-finite result.  n <= 2048. */
int emme_null_vectors_batch(emme_ctx_t* ctx, int n, int nbatch, const double* M, double* vecs, int* info);

/* The reference's driver (src/main.cpp:182-338) on an input.json TEXT: one solve, or a
 * parameter scan over every key written {head, step, tail}, with omega continuation.
 * matrix_dir (may be NULL): directory for the raw eigenMatrix .bin files (must exist, like the
 * reference's eigenMatrics/).  *output_text receives the output.json text (release with
 * emme_free).  Per-point failures become {"eigenvalue":"NaN","reason":...} records. */
int emme_run_json(const char* input_text, const char* matrix_dir, char** output_text);
void emme_free(void* p);
/* Values one {head, step, tail:[tail0, tail1]} axis visits, in order (reference
 * src/main.cpp:139-172); turning_flags[k] = 1 where the sweep restarts from the head in the
 * other direction.  Returns the number of values (<= max_values). */
int emme_scan_values(double head, double step, double tail0, double tail1, double* values,
                     int* turning_flags, int max_values);

/* ---- multi-GPU scan: the one collective -------------------------------------------------------
 * The (parameter set, omega guess) items of a scan are independent Newton chains.  The reference
 * walks them sequentially in one process (src/main.cpp:264-324); here item k of n_total goes to
 * rank k mod world (one process per GPU, its own context), nothing is exchanged while iterating,
 * and ONE RCCL all-gather (ncclAllGather over xGMI, 32 B per item) hands every rank all roots.
 * RCCL is bound at run time; a process that never calls these does not load it. */
#define EMME_COMM_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */
typedef struct emme_comm emme_comm_t;
/* Rank 0 makes the id (ncclGetUniqueId) and hands the 128 bytes to the other ranks out of band
 * (a file, a socket, MPI, torch.distributed's store ...). */
int emme_comm_unique_id(unsigned char* id /* EMME_COMM_ID_BYTES */);
/* Collective over all `world` ranks (ncclCommInitRank). device < 0 => current device. */
int emme_comm_create(const unsigned char* id, int rank, int world, int device, emme_comm_t** out);
void emme_comm_destroy(emme_comm_t* comm);
/* Collective.  In: this rank's results in the order of its share (items rank, rank+world, ...),
 * n_local of them; n_local must be the round-robin share of n_total (EMME_EINVAL otherwise,
 * checked before the collective).  Out (host, on every rank): all n_total results in item order.
 * hip_stream: stream for the copies and the collective (NULL = default). */
int emme_gather_roots(emme_comm_t* comm, void* hip_stream, const double* roots /* 2*n_local */,
                      const int* iters, const int* info, int n_local, int n_total,
                      double* roots_all /* 2*n_total */, int* iters_all, int* info_all);
/* 0 if RCCL can be bound in this process.  NOT a collective: every rank asks before emme_comm_create, so that all
 * ranks can agree (e.g. an all-reduce over an existing process group) on the C-ABI gather or on a fall-back. */
int emme_comm_available(void);
/* The slot mapping emme_gather_roots uses, host only (no RCCL, no device): m = emme_gather_slots = ceil(n_total /
 * world) slots of 4 doubles {w_re, w_im, iters, info} per rank; emme_gather_share = items of `rank` in the
 * round-robin deal; emme_gather_pack fills a rank's 4 m doubles (NaN padded; EMME_EINVAL unless n_local is the
 * rank's share); emme_gather_unpack turns the world * 4 m doubles of the all-gather into item order. */
int emme_gather_slots(int n_total, int world);
int emme_gather_share(int n_total, int world, int rank);
int emme_gather_pack(int rank, int world, const double* roots, const int* iters, const int* info, int n_local,
                     int n_total, double* send /* 4 * slots */);
int emme_gather_unpack(int world, int n_total, const double* all /* world * 4 * slots */, double* roots_all,
                       int* iters_all, int* info_all);

#ifdef __cplusplus
}
#endif
#endif /* EMME_HIP_H */
