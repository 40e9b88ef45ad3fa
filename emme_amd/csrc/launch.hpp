// launch.hpp -- host-visible launch descriptors of the device kernels (internal to the .so).
#pragma once
#include <hip/hip_runtime.h>

#include "emme_device.hpp"

namespace emme {

// HBM cache of the omega-independent node records (assemble_cached.hip): the full bisection
// tree down to depth dfull plus up to NODE_CACHE_MAX_SUB full subtrees (root depth rd, root
// path rp, down to depth dd).  Subtree 0 shares the main buffer with the full tree; the others
// are added at run time, each in its own buffer.
constexpr int NODE_CACHE_MAX_SUB = 12;
struct NodeCacheGeom {
    int dfull;
    int nsub;
    int rd[NODE_CACHE_MAX_SUB], dd[NODE_CACHE_MAX_SUB];
    unsigned long long rp[NODE_CACHE_MAX_SUB];
};
struct AssembleLaunch {
    DevParams P;
    int gk_points;       // 15 or 31
    int nbatch;
    int npairs;
    int items_per_group; // scheduling knob: integrals handled per lane group
    const double* tab;   // device: eta[N] | g[N] | b[N]
    const void* pairs;   // device: ushort2[npairs]
    const double* omega; // device: 2*nbatch
    const int* active;   // device or null
    double* M;           // device
    const double* Mold;  // device or null
    double* Mp;          // device or null
    const double* domega; // device (needed when Mold != null)
    unsigned long long* intervals;  // device or null
    int* status;         // device
    unsigned long long* rounds = nullptr;  // device [1], omega-lane kernel diagnostic
    // per-context options (include/emme_hip.h: emme_options_t)
    int skip_lost = 0;         // skip integrals of a matrix whose status flag is already set (Newton loop only)
    int union_sel = 2;         // union walk: intervals served per round
    int union_walk = 1;        // electrostatic GK15 on folded records: union-walk kernel (0: independent lanes)
    int coop_wide_min = 4096;  // deferred list: length from which the one-wave cooperative kernel takes over (-1 never)
    int defer_one_group = 0;   // deferred integrals by one lane group each
    int dense_min_cols = 3;    // dense fill: columns that must need an interval for the MFMA path
};
// lanes-are-nodes kernel; with a node cache (g != null) it reads cached records where they exist
hipError_t launch_assemble(const AssembleLaunch& L, hipStream_t stream, const NodeCacheGeom* g = nullptr,
                           const void* const recs[2] = nullptr,
                           const void* const recs_ext[2][NODE_CACHE_MAX_SUB - 1] = nullptr,
                           const void* const ttab[2] = nullptr, const void* const wtab[2] = nullptr);
// omega-lane form (assemble_wl.hip): the n_act batch items listed in act_idx (device) share
// the omega-independent node data; L.active is ignored.
hipError_t launch_assemble_wl(const AssembleLaunch& L, const int* act_idx, int n_act,
                              hipStream_t stream);

// part -1 = main buffer (full tree + subtree 0), part k >= 0 = run-time subtree k+1
size_t node_cache_bytes(int gk_points, long nitems, const NodeCacheGeom& g, int part);
size_t node_ttab_bytes(int gk_points, int max_intervals);
int node_cache_intervals(const NodeCacheGeom& g);
// wtab != null: electromagnetic SHARED layout -- one record per (pair, interval, node), that of
// moment 0, plus the table of moment factors W (see emme_device.hpp::node_w)
hipError_t launch_node_cache(const AssembleLaunch& L, const NodeCacheGeom& g, int part, double omi,
                             void* recs, void* ttab, void* wtab, double* scale, bool folded,
                             hipStream_t stream);
// electromagnetic fill on the shared layout: a lane walks the three moments of a pair together
hipError_t launch_assemble_cached_em(const AssembleLaunch& L, const NodeCacheGeom& g,
                                     const void* const recs[2],
                                     const void* const recs_ext[2][NODE_CACHE_MAX_SUB - 1],
                                     const void* const ttab[2], const void* const wtab[2],
                                     const double* scale, const void* etab, unsigned long long* worklist,
                                     unsigned int* worklist_count, unsigned long long* defer_info,
                                     const int* act_idx, int n_act, const void* chunks, int nchunks,
                                     hipStream_t stream);
hipError_t launch_assemble_cached(const AssembleLaunch& L, const NodeCacheGeom& g,
                                  const void* const recs[2],
                                  const void* const recs_ext[2][NODE_CACHE_MAX_SUB - 1],
                                  const void* const ttab[2], const double* scale,
                                  const void* etab /*phase table of this launch, or null*/,
                                  unsigned long long* worklist, unsigned int* worklist_count,
                                  unsigned long long* defer_info, const int* act_idx, int n_act,
                                  const void* chunks /*int2[nchunks]: (first, size)*/, int nchunks,
                                  hipStream_t stream);
// exp(T omega) for every cached interval/node and the n_act omegas of a launch: etab is
// [n_intervals][GW][n_act] complex (the records must be in the folded form)
hipError_t launch_phase_table(int gk_points, int n_intervals, const void* const ttab[2],
                              const double* omega, const int* act_idx, int n_act, void* etab,
                              hipStream_t stream);
// integrals deferred by the cached kernels, recomputed by k_assemble_coop (a workgroup each)
hipError_t launch_assemble_list(const AssembleLaunch& L, const unsigned long long* worklist,
                                const unsigned int* count, const NodeCacheGeom* g,
                                const void* const recs[2],
                                const void* const recs_ext[2][NODE_CACHE_MAX_SUB - 1],
                                const void* const ttab[2], const void* const wtab[2], bool folded,
                                hipStream_t stream, bool tiled = false, const unsigned char* const tile_poison[2] = nullptr);

// ---- dense (matrix-core) fill: assemble_dense.hip (electrostatic GK15; electromagnetic GK31) ------
// tiled record layout: see node_cache.hpp (gk_points 15: 8 KB per (tile, interval); 31: 16 KB)
size_t node_cache_bytes_tiled(long npairs, const NodeCacheGeom& g, int part, int gk_points = 15);
// tile_poison [ntiles] (device, zeroed once per contour class): set to 1 for every tile that gets a poisoned block;
// wtab (electromagnetic contexts): the moment factor W per (interval, node lane), written beside ttab
hipError_t launch_node_cache_tiled(const AssembleLaunch& L, const NodeCacheGeom& g, int part, double omi,
                                   void* recs, void* ttab, double* scale, hipStream_t stream,
                                   unsigned char* tile_poison = nullptr, void* wtab = nullptr);
// weighted phase tables of one launch: btab_bytes(cached intervals, chunks of the launch)
size_t btab_bytes(int nslots, int nchunks, int gk_points = 15);
// wmap[position in the omega list] = chunk << 8 | position of that omega in its chunk (its first column / nm)
hipError_t launch_btab(int gk_points, int nm, int nslots, const void* const ttab[2], const void* const wtab[2],
                       const double* omega, const int* act_idx, int n_act, const int* wmap, int nchunks, void* btab,
                       hipStream_t stream);
// act_idx: the launch's omegas, cost-sorted; chunks: int2 (first position, size <= 16 / nm) per chunk.
// stats (nullable): counters (dense rounds, vector rounds, vector columns, tile tasks, ...)
hipError_t launch_assemble_dense(const AssembleLaunch& L, const NodeCacheGeom& g, const void* const recs[2],
                                 const void* const recs_ext[2][NODE_CACHE_MAX_SUB - 1], const double* scale,
                                 const void* btab, unsigned long long* worklist, unsigned int* worklist_count,
                                 unsigned long long* defer_info, const int* act_idx, int n_act,
                                 const void* chunks, int nchunks, unsigned long long* stats, hipStream_t stream,
                                 const unsigned char* const tile_poison[2] = nullptr, int n_wide = 0,
                                 unsigned int* overflow = nullptr);

// tr(A_b^-1 B_b) by partial-pivot LU of the augmented system [A | B]; A, B destroyed.
hipError_t launch_trace_solve(int n, int nbatch, double* A, double* B, const int* active,
                              double* tr /*2*nbatch*/, int* info, hipStream_t stream);

// dst1[b] = src[b] (and dst2[b], if given) for the active n x n matrices of the batch
hipError_t launch_copy_active(int n, int nbatch, const double* src, double* dst1, double* dst2,
                              const int* active, hipStream_t stream);

// Mp[b] = (M[b] - Mold[b]) / domega[b], Mold[b] = M[b], work[b] = M[b] (if given) for the active matrices: the secant
// of a Newton step (include/solver.h:157) and the copies of the next one, in one pass
hipError_t launch_secant_copy(int n, int nbatch, const double* M, double* Mold, double* work, double* Mp,
                              const double* domega, const int* active, hipStream_t stream);

// the same for matrices that are symmetric bit for bit (every fill kernel writes an entry with its mirror): reads the
// upper triangles only; Mold is afterwards valid in its upper triangle only
hipError_t launch_secant_copy_sym(int n, int nbatch, const double* M, double* Mold, double* work, double* Mp,
                                  const double* domega, const int* active, hipStream_t stream);

// Blocked version (linstep_blocked.hip): whole L21 panel in LDS for n <= ~560, in chunks up to 1024.
// nwg workgroups per matrix (1: one does everything; > 1: one factors A, the others carry B's
// columns; `items` = device list of the nitems matrices to work on, null for all of them);
// `scratch` holds trace_solve_blocked_scratch(n, nbatch) bytes.
size_t trace_solve_blocked_lds(int n);
size_t trace_solve_chunked_lds(int n);  // 560 < n <= 1024: panel rows in chunks (helper workgroups required;
                                        // the launcher returns hipErrorNotSupported when it cannot)
size_t trace_solve_blocked_scratch(int n, int nbatch);
hipError_t launch_trace_solve_blocked(int n, int nbatch, double* A, double* B, const int* active,
                                      double* tr, int* info, int nwg, const int* items, int nitems,
                                      void* scratch, hipStream_t stream, int group_min_n = 256, int spin_limit = 0);

// ---- nullSpace (nullspace.hip): factorisation alone + inverse iteration --------------------------------
// P A = L U in place (multipliers left of the pivots, rows never moved), no right-hand sides; for orders whose
// panel fits one workgroup's LDS (hipErrorNotSupported otherwise).  scratch as for launch_trace_solve_blocked.
hipError_t launch_lu_inplace(int n, int nbatch, double* A, const int* items, int nitems, int* info, void* scratch,
                             hipStream_t stream);
int trace_solve_nb();  // rows per panel = rows per row-order snapshot of the blocked factorisations
// row-order snapshots inside a blocked factorisation's scratch: logical row x of matrix b is physical row
// maps[(b ceil(n / nb) + x / nb) n + x]
const int* trace_solve_rowmaps(const void* scratch, int n, int nbatch);
// any order up to ~8000: unblocked, slow; maps [nbatch][n] receives the row order (one snapshot, nb = n)
hipError_t launch_lu_unblocked_inplace(int n, int nbatch, double* A, int* maps, int* info, hipStream_t stream);
size_t null_iterate_lds(int n);
// right singular vector of the smallest singular value of every matrix, from its in-place LU: vecs [nbatch][n]
// complex, info 0 / lu_info's code / EMME_ENUMERIC for a non-finite result
hipError_t launch_null_iterate(int n, const double* A_lu, const int* rowmaps, int nb, const int* items, int nitems,
                               const int* lu_info, double* vecs, int* info, int max_sweeps, hipStream_t stream);

// QR-secant form of the step (linstep_qr.hip, reference include/solver.h:210-383): Wt holds the
// TRANSPOSE of each matrix (destroyed), Mp the secant derivative (read only); writes
// tr = t_n / R_nn so that domega = -1/tr = -R_nn / t_n; info = k > 0 when R11(k,k) == 0.
size_t qr_secant_lds(int n);
hipError_t launch_qr_secant(int n, int nbatch, double* Wt, const double* Mp, const int* active,
                            double* tr, int* info, hipStream_t stream);
// out_b = in_b^T for every active item (n x n complex, row-major)
hipError_t launch_transpose(int n, int nbatch, const double* in, double* out, const int* active,
                            hipStream_t stream);

// domega = -1/tr; omega += domega; iters += 1; active = !(|domega| < tol |omega|) && info == 0
// (include/solver.h:139-140, src/main.cpp:53-56). `iterates` (nullable) records omega.
hipError_t launch_newton_update(int nbatch, const double* tr, double* omega, double* domega,
                                int* active, int* iters, const int* info, double tol,
                                double* iterates, int iter_index, int iter_stride,
                                hipStream_t stream, double* pub_omega = nullptr /* pinned host, all omegas */,
                                const int* lost = nullptr /* status flags: a flagged chain retires with EMME_ENUMERIC */);

// active == 2 ("converged, last step done") -> 0; pub_* (pinned host, nullable): the flags, the interval
// counters and the deferred-integral count for the host to read after its next synchronisation
hipError_t launch_retire(int nbatch, int* active, hipStream_t stream, int* pub_active = nullptr,
                         const unsigned long long* intervals = nullptr, unsigned long long* pub_intervals = nullptr,
                         const unsigned int* deferred = nullptr, unsigned int* pub_deferred = nullptr,
                         unsigned int* overflow = nullptr, unsigned int* pub_overflow = nullptr);
// small integer lists from a pinned host block to device memory: dst1 <- src[0..n1), dst2 <- src[n1..n1+n2)
hipError_t launch_stage_ints(const int* src_pinned, int* dst1, int n1, int* dst2, int n2, hipStream_t stream);

// Mp = (M - Mold) / domega, elementwise (include/solver.h:54-57), for active items.
hipError_t launch_secant(int nbatch, size_t nn, const double* M, const double* Mold,
                         const double* domega, const int* active, double* Mp, hipStream_t stream);

// bessel_miller on n complex arguments (device pointers): out[4n] = y0, y1, mu + y0, -/+ z
hipError_t launch_bessel_probe(const double* z, int n, double* out, hipStream_t stream);

}  // namespace emme
