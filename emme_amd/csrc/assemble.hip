// assemble.hip -- the Grid x Grid complex<double> dispersion-matrix fill on gfx950.
//
// Replaces EigenSolver::matrixAssembler (reference include/solver.h:417-515) and the
// DedicatedThreadPool fan-out behind it (include/DedicatedThreadPool.h:19-270): one launch
// fills M(omega_b) for a whole batch of omega candidates.
//
// Work decomposition: item = (batch b, pair (i<j), moment m).  A lane group of GW = 16
// (GK15) or 32 (GK31) lanes owns one item at a time and evaluates all nodes of one
// quadrature interval per step (see emme_device.hpp).  Groups run a flattened state
// machine -- "evaluate one interval, then accept / split / fetch the next item" -- so a
// group that finishes an integral early starts the next one instead of idling while its
// wave-mates finish theirs.  The pair list is ordered by diagonal offset j-i, which keeps
// the four integrals in flight in a wave at near-identical cost (interval counts are almost
// constant along a diagonal).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "assemble_common.hpp"
#include "launch.hpp"
#include "node_cache.hpp"

namespace emme {

namespace {

struct AsmArgs {
    DevParams P;
    const double* tab;     // eta[N], g[N], b[N]
    const ushort2* pairs;  // (i, j), i < j, ordered by j - i
    int npairs;
    const double2* omega;  // [nbatch]
    const int* active;     // [nbatch] or null: item skipped when 0
    double2* M;            // [nbatch][dim][dim]
    const double2* Mold;   // null, or [nbatch][dim][dim]  -> also write Mp
    double2* Mp;           //   Mp = (M - Mold) / domega   (include/solver.h:54-57)
    const double2* domega; // [nbatch]
    unsigned long long* intervals;  // [nbatch], atomically accumulated (may be null)
    int* status;                    // [nbatch], set non-zero on depth-cap / non-finite
    // LIST mode: explicit (batch index, item) entries instead of the (item, blockIdx.y) grid
    const unsigned long long* worklist;  // entry = batch << 32 | item
    const unsigned int* worklist_count;
    // LIST mode: node-record cache, used for every interval it holds (null = none)
    CacheGeom geom;
    const NodeRec* recs[2];
    const NodeRec* recs_ext[2][NODE_CACHE_MAX_SUB - 1];
    const double2* ttab[2];
    // electromagnetic fills with a shared cache: records are those of moment 0, indexed by pair;
    // moment m multiplies them by (c_nv W)^m with W from wtab (null = one record per moment)
    const double2* wtab[2];
    int folded;  // cached records are in the folded form (exp(A0) inside the amplitudes)
    int tiled;   // the cache is in the tiled layout of the dense fill (node_cache.hpp): `recs` point to doubles
    unsigned int count_lo, count_hi;  // k_assemble_coop: run only if count_lo <= list length < count_hi
    int skip_lost;  // LIST mode: integrals of a matrix whose status flag is already set are skipped
    const unsigned char* tile_poison[2];  // tiled cache: tiles that hold a poisoned (pair, interval) block, per class
};

// Value of the integrand at one quadrature node when a node-record cache may hold the
// interval: take the record (lane = node) wherever there is one and compute the node data on
// the spot only for uncached intervals.  Used for integrals the cached kernel deferred and for
// omegas whose integrals are so long that their serial latency matters more than throughput:
// here all 15 (31) nodes of an interval advance in parallel.
// (the tables of the omega's contour class: a caller whose omega is fixed for many intervals -- the cooperative
// kernel -- reads the three pointers once per integral instead of once per interval, a scalar load and a wait each)
struct ClassTables {
    const NodeRec* recs;
    const double2* ttab;
    const double2* wtab;
    const unsigned char* tile_poison;
};
__device__ __forceinline__ ClassTables class_tables(const AsmArgs& A, int cls) {
    return ClassTables{A.recs[cls], A.ttab[cls], A.wtab[cls], A.tile_poison[cls]};
}
template <int GW>
__device__ __forceinline__ cd node_value(const AsmArgs& A, const ClassTables& ct, int depth, unsigned long long path,
                                         long cache_item, int lane_in_group, double x,
                                         const PairConst& pc, const OmegaConst& oc, int m,
                                         const TransConsts& tc) {
    const int cls = oc.omi > 0.0 ? 0 : 1;
    int which;
    const int cslot = ct.recs ? A.geom.slot(depth, path, which) : -1;
    const NodeRec* buf = cslot < 0 ? nullptr : (which < 0 ? ct.recs : A.recs_ext[cls][which]);
    NodeData d;
    if (buf && A.tiled) {
        // tiled layout (electrostatic GK15, folded): block of the pair's tile, node slot of this lane
        const double* tb = reinterpret_cast<const double*>(buf);
        const long tile = cache_item / TILE_PAIRS;
        const int p = (int)(cache_item - tile * TILE_PAIRS);
        const double* blk =
            which < 0 ? tb + ((size_t)tile * A.geom.ni_main() + cslot) * TILE_BLOCK
                      : tb + ((size_t)tile * A.geom.ni_sub(which + 1) + (cslot - A.geom.base[which + 1])) * TILE_BLOCK;
        const int sn = slotnode_of_lane(lane_in_group);
        const double4 ra = *reinterpret_cast<const double4*>(reinterpret_cast<const double2*>(blk) + tile_index(2 * sn, p));
        const cd q1 = mk(ra.x, ra.y), q0 = mk(ra.z, ra.w);  // (Q1, Q0) of the node: one 32-byte piece
        // a POISONED block (k_node_cache_tiled: the folded amplitude of one of its nodes is not representable;
        // flag in the padding slot, which the group's lane 15 has just read): this interval is evaluated unfolded,
        // from scratch, exactly like an uncached one -- exp(A0 + T omega) with the reference's clamp and, where
        // the reference overflows, its infinity
        if (ct.tile_poison && ct.tile_poison[tile] != 0) {  // (uniform per integral; almost never)
            const double flag = __shfl(ra.x, (threadIdx.x & 63 & ~(GW - 1)) | 15);
            if (flag != 0.0) return node_eval(node_data(x, A.P, pc, oc.omi, m), oc.omega, tc);
        }
        const double2 tt = ct.ttab[(long)cslot * GW + lane_in_group];
        const cd arg = mk(tt.x, tt.y) * oc.omega;
        // (no safe_exp clamp on tiled records, like the dense fill that shares them: node_cache.hpp;
        // exp(T omega) below -745 underflows to an exact 0 by itself)
        double sa, ca;
        fsincos(arg.y, sa, ca, tc);
        const double ea = fexp(arg.x, tc);
        return mk(ea * ca, ea * sa) * (oc.omega * q1 + q0);
    }
    if (buf) {
        const bool shared = ct.wtab != nullptr;
        const long ci = shared ? cache_item / A.P.nm : cache_item;
        const NodeRec rec =
            which < 0 ? buf[(ci * A.geom.ni_main() + cslot) * GW + lane_in_group]
                      : buf[(ci * A.geom.ni_sub(which + 1) + (cslot - A.geom.base[which + 1])) * GW +
                            lane_in_group];
        const double2 tt = ct.ttab[(long)cslot * GW + lane_in_group];
        d.T = mk(tt.x, tt.y);
        d.Q1 = mk(rec.Q1.x, rec.Q1.y);
        d.Q0 = mk(rec.Q0.x, rec.Q0.y);
        if (shared && m > 0) {
            const double2 ww = ct.wtab[(long)cslot * GW + lane_in_group];
            cd nv = pc.c_nv * mk(ww.x, ww.y);
            if (m == 2) nv = nv * nv;
            d.Q1 = d.Q1 * nv;
            d.Q0 = d.Q0 * nv;
        }
        if (A.folded) {
            // folded record: A0 = (|exp A0|^2, Re A0), amplitudes already carry exp(A0); here
            // exp(T omega) is computed on the spot (no phase table in this kernel)
            const cd arg = d.T * oc.omega;
            if (!(rec.A0.y + arg.x >= -40.)) {
                if (rec.A0.y + arg.x < -40.) return mk(0.0, 0.0);  // safe_exp clamp
            }
            double sa, ca;
            fsincos(arg.y, sa, ca, tc);
            const double ea = fexp(arg.x, tc);
            return mk(ea * ca, ea * sa) * (oc.omega * d.Q1 + d.Q0);
        }
        d.A0 = mk(rec.A0.x, rec.A0.y);
    } else {
        d = node_data(x, A.P, pc, oc.omi, m);
    }
    return node_eval(d, oc.omega, tc);
}

template <int PTS, bool LIST>
#ifndef EMME_ASM_MIN_WAVES
#define EMME_ASM_MIN_WAVES 3
#endif
__global__ __launch_bounds__(256, EMME_ASM_MIN_WAVES) void k_assemble(AsmArgs A) {
    constexpr int GW = PTS == 15 ? 16 : 32;
    constexpr int GROUPS_PER_BLOCK = 256 / GW;
    constexpr int MAXD = EMME_MAX_DEPTH;  // bisection depth the LDS interval stack can hold
    extern __shared__ double lds_tab[];  // eta | g | b  (3N doubles) | per-group (mid, r) stack

    const DevParams& P = A.P;
    const TransConsts TC = trans_consts();
    int b = LIST ? 0 : blockIdx.y;
    if (!LIST && A.active && A.active[b] == 0) return;
    const int N = P.N, dim = P.dim;

    for (int k = threadIdx.x; k < 3 * N; k += blockDim.x) lds_tab[k] = A.tab[k];
    __syncthreads();
    const double* eta = lds_tab;
    const double* gtab = lds_tab + N;
    const double* btab = lds_tab + 2 * N;
    // (mid, r) of every split ancestor of the current interval, one stack per lane group.
    // All lanes of a group store the same value to the same slot and later read it back,
    // so plain per-thread program order is enough (no cross-lane hand-off).
    double2* stk = reinterpret_cast<double2*>(lds_tab + 3 * N + (3 * N & 1)) +
                   (threadIdx.x / GW) * MAXD;

    double2* Mb = nullptr;
    const double2* Moldb = nullptr;
    double2* Mpb = nullptr;
    cd rdw = mk(0.0, 0.0);
    OmegaConst oc;
    auto bind_batch = [&]() {  // everything that depends on the batch index b
        Mb = A.M + (size_t)b * dim * dim;
        Moldb = A.Mold ? A.Mold + (size_t)b * dim * dim : nullptr;
        Mpb = A.Mp ? A.Mp + (size_t)b * dim * dim : nullptr;
        if (Moldb) rdw = rcp(mk(A.domega[b].x, A.domega[b].y));
        oc.omega = mk(A.omega[b].x, A.omega[b].y);
        oc.omi = -copysign(1.0, oc.omega.x);
    };
    if (!LIST) bind_batch();

    auto store = [&](int r, int c, cd v) {
        const size_t idx = (size_t)r * dim + c;
        Mb[idx] = make_double2(v.x, v.y);
        if (Moldb) {
            const double2 o = Moldb[idx];
            const cd d = (v - mk(o.x, o.y)) * rdw;
            Mpb[idx] = make_double2(d.x, d.y);
        }
    };

    // diagonal (include/solver.h:442-443, 465-470): block 0 of each batch item
    if (!LIST && blockIdx.x == 0) {
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            store(i, i, mk(P.diag_a, 0.0));
            if (P.nm == 3) {
                store(i, i + N, mk(0.0, 0.0));
                store(i + N, i, mk(0.0, 0.0));
                store(i + N, i + N, mk(P.diag_d * btab[i], 0.0));
            }
        }
    }

    const int lane_in_group = threadIdx.x % GW;
    const int group = blockIdx.x * GROUPS_PER_BLOCK + threadIdx.x / GW;
    const int ngroups = gridDim.x * GROUPS_PER_BLOCK;
    const GkLane gk = gk_lane<PTS>(lane_in_group);

    const double qa = 0.0, qb = M_PI / 2.0;  // include/functions.h:319 / :328
    const double inv_scale = 2. / (qb - qa);
    const int nitems = LIST ? (int)*A.worklist_count : A.npairs * P.nm;

    // ---- group state ------------------------------------------------------------
    int item = group;
    bool live = item < nitems;
    int i = 0, j = 0, m = 0;
    PairConst pc{};
    double dg = 0.0;
    int depth = 0;
    unsigned long long path = 0;  // index of the current interval at this depth
    double l = qa, r = qb;        // the current interval
    double abs_tol = 0.0;
    cd sum = mk(0.0, 0.0);
    unsigned long long my_intervals = 0;
    int item_intervals = 0;  // safety valve: no integral may run away (every wave must exit)
    int bad = 0;

    auto load_item = [&]() {
        int it = item;
        if (LIST) {
            const unsigned long long e = A.worklist[item];
            it = (int)(e & 0xffffffffull);
            b = (int)(e >> 32);
            bind_batch();
        }
        const int p = it / P.nm;
        m = it - p * P.nm;
        const ushort2 ij = A.pairs[p];
        i = ij.x, j = ij.y;
        dg = gtab[i] - gtab[j];
        pc = make_pair_const(P, eta[i], eta[j], btab[i], btab[j], dg);
        depth = 0, path = 0, abs_tol = 0.0;
        l = qa, r = qb;
        item_intervals = 0;
        sum = mk(0.0, 0.0);
    };
    if (live) load_item();

    while (live) {
        const double mid = (r + l) / 2;
        const double scale = (r - l) / 2;
        // abscissa scale * x + mid, rounded like the reference (no FMA contraction)
        const double x = __dadd_rn(__dmul_rn(scale, gk.x), mid);

        const long cache_item = LIST ? (long)(A.worklist[item] & 0xffffffffull) : (long)item;
        const cd f = (LIST || A.recs[oc.omi > 0.0 ? 0 : 1] != nullptr)
                         ? node_value<GW>(A, class_tables(A, oc.omi > 0.0 ? 0 : 1), depth, path, cache_item, lane_in_group, x, pc, oc, m, TC)
                         : integrand(x, P, pc, oc, m);
        const double Kx = group_sum<GW>(gk.wk * f.x), Ky = group_sum<GW>(gk.wk * f.y);
        const double Gx = group_sum<GW>(gk.wg * f.x), Gy = group_sum<GW>(gk.wg * f.y);
        ++my_intervals;
        ++item_intervals;

        // include/functions.h:203-208, 231-233
        const double dKx = Kx - Gx, dKy = Ky - Gy;
        const double absK = sqrt(fma(Kx, Kx, Ky * Ky));
        double err = fmax(sqrt(fma(dKx, dKx, dKy * dKy)), absK * (2.0 * 2.220446049250313e-16));
        const cd integral = mk(Kx * scale, Ky * scale);
        err *= scale;
        const double rel_abs = P.rel_tol * (absK * scale);  // |rel * integral|
        if (abs_tol == 0.0) abs_tol = rel_abs;              // :237-239
        // :240-242.  ldexp(scale, max_sub) > 0.99 (b - a) with scale = (b-a) 2^-(depth+1)
        // (up to rounding far below the 1 % margin) is exactly depth < max_sub.
        bool split = depth < P.max_sub && err > abs_tol * inv_scale + P.prec_goal &&
                     err > rel_abs + P.prec_goal;
        if (split && (depth >= MAXD || item_intervals >= EMME_MAX_INTERVALS)) {  // flag and accept
            split = false;
            bad = 1;
        }
        if (split) {
            // left half first (the reference pushes [mid,r] then [l,mid]); remember the
            // right half's bounds for when the walk comes back up
            stk[depth] = make_double2(mid, r);
            r = mid;
            ++depth;
            path <<= 1;
        } else {
            sum = sum + integral;
            ++path;
            while (depth > 0 && !(path & 1)) {
                path >>= 1;
                --depth;
            }
            if (depth == 0) {
                // integral finished: kappa = -i pref sum (src/Parameters.cpp:182-183)
                cd kap = mk(P.pref * sum.y, -(P.pref * sum.x));
                if (kappa_bad(kap)) bad = 1;
                kap = kap + kappa_e(m, P, pc.de, dg, oc.omega);
                if (lane_in_group == 0) {
                    if (m == 0) {
                        // A_ij = -kappa_all(0) W_ij dx (include/solver.h:448-453)
                        const double w = pair_weight(i, j, N);
                        const cd v = (-(w * P.dx)) * kap;
                        store(i, j, v);
                        store(j, i, v);
                    } else if (m == 1) {
                        // B block and its mirrors (include/solver.h:480-504)
                        const cd v = P.dx * kap;
                        store(i, j + N, v);
                        store(j, i + N, -v);
                        store(i + N, j, -v);
                        store(j + N, i, v);
                    } else {
                        const cd v = P.dx * kap;
                        store(i + N, j + N, v);
                        store(j + N, i + N, v);
                    }
                }
                if (LIST && lane_in_group == 0) {  // the batch index changes with the item
                    if (A.intervals) atomicAdd(&A.intervals[b], (unsigned long long)item_intervals);
                    if (bad) A.status[b] = 1;
                    bad = 0;
                }
                item += ngroups;
                live = item < nitems;
                if (live) load_item();
            } else {
                // right sibling of the ancestor that was split at depth-1
                const double2 pr = stk[depth - 1];
                l = pr.x;
                r = pr.y;
            }
        }
    }

    if (!LIST && lane_in_group == 0) {
        if (A.intervals && my_intervals) atomicAdd(&A.intervals[b], my_intervals);
        if (bad) A.status[b] = 1;
    }
}

// ---- cooperative form for the deferred list ---------------------------------------------------
// The few integrals the cached kernel defers are the long ones (up to thousands of intervals,
// bisection depth ~19): walked by ONE lane group they take milliseconds each and set the
// latency of the whole deferred pass.  Here a whole workgroup owns one integral: its NG lane
// groups pop up to NG intervals per round from a shared LDS stack (the deepest NG, so the
// stack stays bounded by NG * (depth + 1) like a depth-first walk), evaluate them in
// parallel and push the children of the ones that split.  Which group gets which interval and
// where children go is decided by prefix sums, never by timing, so the result is
// deterministic; the accept/split rule of an interval depends only on that interval and on
// abs_tol from the root (include/functions.h:231-247), so the set of intervals -- the tree --
// is exactly the sequential one.  Only the order in which accepted pieces are added differs
// (per-group partial sums, then a fixed-order total): a rounding-level change.
struct CoopEnt {
    double l, r;
    unsigned long long path;
    int depth, pad;
};

template <int PTS, int BT>
__global__ __launch_bounds__(BT) void k_assemble_coop(AsmArgs A) {
    constexpr int GW = PTS == 15 ? 16 : 32;
    constexpr int NG = BT / GW;
    constexpr int CAP = 4 * BT;  // >= NG * (MAXD + 1)
    constexpr int MAXD = EMME_MAX_DEPTH;
    static_assert(CAP >= NG * (MAXD + 1), "stack bound of the deepest-first walk");
    extern __shared__ double lds_tab[];  // eta | g | b (3N doubles) | CoopEnt stack[CAP]
    __shared__ int s_cnt[NG];
    __shared__ double s_part[NG][2];
    __shared__ int s_bad;
    __shared__ int s_lost[2];

    const DevParams& P = A.P;
#ifdef EMME_COOP_SREG_CONSTS
    const TransConsts TC = trans_consts();
#else
    const TransConsts TC = trans_consts_v();  // (vector registers: see trans_consts_v)
#endif
    const int N = P.N, dim = P.dim;
    // (the list length first: both variants are launched for every fill with a grid sized for the longest list, and
    // almost always the list is short or empty -- a workgroup without an item must not stage tables: 0.125 ms per
    // launch, 2.6 ms per bench search, went into exactly that)
    const int nitems = (int)*A.worklist_count;
    if ((unsigned)nitems < A.count_lo || (unsigned)nitems >= A.count_hi) return;  // the other variant's list
    if ((int)blockIdx.x >= nitems) return;
    for (int k = threadIdx.x; k < 3 * N; k += blockDim.x) lds_tab[k] = A.tab[k];
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    const double* eta = lds_tab;
    const double* gtab = lds_tab + N;
    const double* btab = lds_tab + 2 * N;
    CoopEnt* stk = reinterpret_cast<CoopEnt*>(lds_tab + 3 * N + (3 * N & 1));

    const int lane_in_group = threadIdx.x % GW;
    const int grp = threadIdx.x / GW;
    const GkLane gk = gk_lane<PTS>(lane_in_group);
    const double qa = 0.0, qb = M_PI / 2.0;
    const double inv_scale = 2. / (qb - qa);
    int trip = 0;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x, ++trip) {
        const unsigned long long e = A.worklist[item];
        const int it = (int)(e & 0xffffffffull);
        const int b = (int)(e >> 32);
        if (A.skip_lost) {
            // The matrix already holds a non-finite integral (this kernel's own `isbad`, or the cached fill's):
            // its chain retires at the next Newton step whatever else goes into it (the reference's zsysv fails
            // on such a matrix, include/solver.h:142-153) -- the rest of its integrals, thousands of intervals
            // each from scratch, are not worth a cycle.  One thread reads the flag for the workgroup (the slot
            // alternates: the next write to it is two barriers away).
            if (threadIdx.x == 0) s_lost[trip & 1] = A.status[b];
            __syncthreads();
            if (s_lost[trip & 1] != 0) continue;
        }
        OmegaConst oc;
        oc.omega = mk(A.omega[b].x, A.omega[b].y);
        oc.omi = -copysign(1.0, oc.omega.x);
        const int p = it / P.nm;
        const int m = it - p * P.nm;
        const ushort2 ij = A.pairs[p];
        const int i = ij.x, j = ij.y;
        const double dg = gtab[i] - gtab[j];
        const PairConst pc = make_pair_const(P, eta[i], eta[j], btab[i], btab[j], dg);
        const ClassTables ct = class_tables(A, oc.omi > 0.0 ? 0 : 1);

        // one interval: GK estimate, error, accept/split (include/functions.h:203-208, 231-247)
        double abs_tol = 0.0;
        auto evaluate = [&](double l, double r, int depth, unsigned long long path, cd& integral,
                            double& mid) -> bool {
            mid = (r + l) / 2;
            const double scale = (r - l) / 2;
            const double x = __dadd_rn(__dmul_rn(scale, gk.x), mid);
            const cd f = node_value<GW>(A, ct, depth, path, (long)it, lane_in_group, x, pc, oc, m, TC);
            const double Kx = group_sum<GW>(gk.wk * f.x), Ky = group_sum<GW>(gk.wk * f.y);
            const double Gx = group_sum<GW>(gk.wg * f.x), Gy = group_sum<GW>(gk.wg * f.y);
            const double dKx = Kx - Gx, dKy = Ky - Gy;
            const double absK = sqrt(fma(Kx, Kx, Ky * Ky));
            double err = fmax(sqrt(fma(dKx, dKx, dKy * dKy)), absK * (2.0 * 2.220446049250313e-16));
            integral = mk(Kx * scale, Ky * scale);
            err *= scale;
            const double rel_abs = P.rel_tol * (absK * scale);
            if (abs_tol == 0.0) abs_tol = rel_abs;
            return depth < P.max_sub && err > abs_tol * inv_scale + P.prec_goal &&
                   err > rel_abs + P.prec_goal;
        };

        // root: every group evaluates it (identical bits) so that all hold abs_tol
        cd gsum = mk(0.0, 0.0);
        int top = 0, n_intervals = 1, bad = 0;
        {
            cd integral;
            double mid;
            const bool split = evaluate(qa, qb, 0, 0ull, integral, mid);
            if (split) {
                if (threadIdx.x == 0) {
                    stk[0] = CoopEnt{mid, qb, 1ull, 1, 0};
                    stk[1] = CoopEnt{qa, mid, 0ull, 1, 0};
                }
                top = 2;
            } else if (grp == 0) {
                gsum = integral;
            }
        }
        __syncthreads();

        while (top > 0) {  // `top`, `n_intervals`, `bad` are block-uniform
            const int n_take = top < NG ? top : NG;
            const int base = top - n_take;
            bool split = false;
            CoopEnt en{};
            double mid = 0.0;
            if (grp < n_take) {
                en = stk[top - 1 - grp];
                cd integral;
                split = evaluate(en.l, en.r, en.depth, en.path, integral, mid);
                if (split && en.depth >= MAXD) split = false, bad = 1;
                if (!split) gsum = gsum + integral;
            }
            if (lane_in_group == 0) s_cnt[grp] = split ? 2 : 0;
            if (bad && lane_in_group == 0) s_bad = 1;
            __syncthreads();  // all entries read, all counts visible
            int above = 0, total = 0;
#pragma unroll
            for (int h = 0; h < NG; ++h) {
                const int c = s_cnt[h];
                total += c;
                if (h > grp) above += c;
            }
            n_intervals += n_take;
            int ntop = base + total;
            if (ntop > CAP || n_intervals >= EMME_MAX_INTERVALS) {  // flag and stop refining (block-uniform)
                if (threadIdx.x == 0) s_bad = 1;
                ntop = base;
            } else if (split && lane_in_group == 0) {
                // group 0's children end on top, left half above right half
                stk[base + above] = CoopEnt{mid, en.r, 2 * en.path + 1, en.depth + 1, 0};
                stk[base + above + 1] = CoopEnt{en.l, mid, 2 * en.path, en.depth + 1, 0};
            }
            top = ntop;
            __syncthreads();
        }

        // total in fixed group order
        if (lane_in_group == 0) s_part[grp][0] = gsum.x, s_part[grp][1] = gsum.y;
        __syncthreads();
        if (threadIdx.x == 0) {
            cd sum = mk(0.0, 0.0);
            for (int h = 0; h < NG; ++h) sum = sum + mk(s_part[h][0], s_part[h][1]);
            cd kap = mk(P.pref * sum.y, -(P.pref * sum.x));
            int isbad = s_bad;
            if (kappa_bad(kap)) isbad = 1;
            kap = kap + kappa_e(m, P, pc.de, dg, oc.omega);
            double2* Mb = A.M + (size_t)b * dim * dim;
            const double2* Moldb = A.Mold ? A.Mold + (size_t)b * dim * dim : nullptr;
            double2* Mpb = A.Mp ? A.Mp + (size_t)b * dim * dim : nullptr;
            const cd rdw = Moldb ? rcp(mk(A.domega[b].x, A.domega[b].y)) : mk(0.0, 0.0);
            auto store = [&](int rr, int cc, cd v) {
                const size_t idx = (size_t)rr * dim + cc;
                Mb[idx] = make_double2(v.x, v.y);
                if (Moldb) {
                    const double2 o = Moldb[idx];
                    const cd d = (v - mk(o.x, o.y)) * rdw;
                    Mpb[idx] = make_double2(d.x, d.y);
                }
            };
            if (m == 0) {  // include/solver.h:448-453
                const double w = pair_weight(i, j, N);
                const cd v = (-(w * P.dx)) * kap;
                store(i, j, v);
                store(j, i, v);
            } else if (m == 1) {  // include/solver.h:480-504
                const cd v = P.dx * kap;
                store(i, j + N, v);
                store(j, i + N, -v);
                store(i + N, j, -v);
                store(j + N, i, v);
            } else {
                const cd v = P.dx * kap;
                store(i + N, j + N, v);
                store(j + N, i + N, v);
            }
            if (A.intervals) atomicAdd(&A.intervals[b], (unsigned long long)n_intervals);
            if (isbad) A.status[b] = 1;
            s_bad = 0;
        }
        __syncthreads();
    }
}

}  // namespace

hipError_t launch_assemble(const AssembleLaunch& L, hipStream_t stream, const NodeCacheGeom* g,
                           const void* const recs[2],
                           const void* const recs_ext[2][NODE_CACHE_MAX_SUB - 1],
                           const void* const ttab[2], const void* const wtab[2]) {
    AsmArgs A;
    A.tiled = 0;
    A.tile_poison[0] = A.tile_poison[1] = nullptr;
    A.P = L.P;
    A.tab = L.tab;
    A.pairs = (const ushort2*)L.pairs;
    A.npairs = L.npairs;
    A.omega = (const double2*)L.omega;
    A.active = L.active;
    A.M = (double2*)L.M;
    A.Mold = (const double2*)L.Mold;
    A.Mp = (double2*)L.Mp;
    A.domega = (const double2*)L.domega;
    A.intervals = L.intervals;
    A.status = L.status;
    A.skip_lost = L.skip_lost;
    A.worklist = nullptr;
    A.worklist_count = nullptr;
    A.folded = 0;
    A.geom = g ? make_geom(*g) : CacheGeom{};
    for (int c = 0; c < 2; ++c) {
        A.recs[c] = g ? (const NodeRec*)recs[c] : nullptr;
        A.ttab[c] = g ? (const double2*)ttab[c] : nullptr;
        A.wtab[c] = (g && wtab) ? (const double2*)wtab[c] : nullptr;
        for (int k = 0; k < NODE_CACHE_MAX_SUB - 1; ++k)
            A.recs_ext[c][k] = g ? (const NodeRec*)recs_ext[c][k] : nullptr;
    }
    const int gw = L.gk_points == 15 ? 16 : 32;
    const int groups_per_block = 256 / gw;
    const long nitems = (long)L.npairs * L.P.nm;
    // about `items_per_group` integrals per group keeps the tail short without
    // starving the chip when the batch is small
    long want_groups = (nitems + L.items_per_group - 1) / L.items_per_group;
    long gx = (want_groups + groups_per_block - 1) / groups_per_block;
    if (gx < 1) gx = 1;
    if (gx > 65535) gx = 65535;
    dim3 grid((unsigned)gx, (unsigned)L.nbatch), block(256);
    const size_t lds = ((size_t)3 * L.P.N + (3 * L.P.N & 1)) * sizeof(double) +
                       (size_t)groups_per_block * EMME_MAX_DEPTH * sizeof(double2);
    if (L.gk_points == 15)
        hipLaunchKernelGGL((k_assemble<15, false>), grid, block, lds, stream, A);
    else
        hipLaunchKernelGGL((k_assemble<31, false>), grid, block, lds, stream, A);
    return hipGetLastError();
}

// Integrals the cached kernel deferred (they need intervals deeper than the cache holds):
// recomputed whole by the lanes-are-nodes kernel from a device-side list whose length the
// host does not know -- a fixed grid strides over it and exits at once when it is empty.
hipError_t launch_assemble_list(const AssembleLaunch& L, const unsigned long long* worklist,
                                const unsigned int* count, const NodeCacheGeom* g,
                                const void* const recs[2],
                                const void* const recs_ext[2][NODE_CACHE_MAX_SUB - 1],
                                const void* const ttab[2], const void* const wtab[2], bool folded,
                                hipStream_t stream, bool tiled, const unsigned char* const tile_poison[2]) {
    AsmArgs A;
    A.tile_poison[0] = (tiled && tile_poison) ? tile_poison[0] : nullptr;
    A.tile_poison[1] = (tiled && tile_poison) ? tile_poison[1] : nullptr;
    A.tiled = tiled ? 1 : 0;
    A.P = L.P;
    A.tab = L.tab;
    A.pairs = (const ushort2*)L.pairs;
    A.npairs = L.npairs;
    A.omega = (const double2*)L.omega;
    A.active = nullptr;
    A.M = (double2*)L.M;
    A.Mold = (const double2*)L.Mold;
    A.Mp = (double2*)L.Mp;
    A.domega = (const double2*)L.domega;
    A.intervals = L.intervals;
    A.status = L.status;
    A.skip_lost = L.skip_lost;
    A.worklist = worklist;
    A.worklist_count = count;
    A.folded = folded ? 1 : 0;
    A.geom = g ? make_geom(*g) : CacheGeom{};
    for (int c = 0; c < 2; ++c) {
        A.recs[c] = g ? (const NodeRec*)recs[c] : nullptr;
        A.ttab[c] = g ? (const double2*)ttab[c] : nullptr;
        A.wtab[c] = (g && wtab) ? (const double2*)wtab[c] : nullptr;
        for (int k = 0; k < NODE_CACHE_MAX_SUB - 1; ++k)
            A.recs_ext[c][k] = g ? (const NodeRec*)recs_ext[c][k] : nullptr;
    }
    const int gw = L.gk_points == 15 ? 16 : 32;
    const int groups_per_block = 256 / gw;
    dim3 grid(2048), block(256);
    const bool one_group = L.defer_one_group != 0;
    if (!one_group) {
        // a workgroup per integral (see k_assemble_coop): 256 threads for a short list (few, long
        // integrals: as many lane groups per integral as its frontier can feed), one wave for a
        // long one (thousands of integrals, mostly deep and narrow: more of them in flight and
        // wave-level barriers).  The list length is on the device: both variants are launched
        // and the one whose range it is not in returns at once.
        const unsigned int wide_min = (unsigned int)L.coop_wide_min;  // (-1 = never: 0xffffffff)
        const size_t tab_bytes = ((size_t)3 * L.P.N + (3 * L.P.N & 1)) * sizeof(double);
        A.count_lo = 0, A.count_hi = wide_min;
        if (L.gk_points == 15)
            hipLaunchKernelGGL((k_assemble_coop<15, 256>), grid, block, tab_bytes + 4 * 256 * sizeof(CoopEnt), stream, A);
        else
            hipLaunchKernelGGL((k_assemble_coop<31, 256>), grid, block, tab_bytes + 4 * 256 * sizeof(CoopEnt), stream, A);
        if (wide_min != 0xffffffffu) {
            A.count_lo = wide_min, A.count_hi = 0xffffffffu;
            if (L.gk_points == 15)
                hipLaunchKernelGGL((k_assemble_coop<15, 64>), dim3(4096), dim3(64), tab_bytes + 4 * 64 * sizeof(CoopEnt), stream, A);
            else
                hipLaunchKernelGGL((k_assemble_coop<31, 64>), dim3(4096), dim3(64), tab_bytes + 4 * 64 * sizeof(CoopEnt), stream, A);
        }
        return hipGetLastError();
    }
    const size_t lds = ((size_t)3 * L.P.N + (3 * L.P.N & 1)) * sizeof(double) +
                       (size_t)groups_per_block * EMME_MAX_DEPTH * sizeof(double2);
    if (L.gk_points == 15)
        hipLaunchKernelGGL((k_assemble<15, true>), grid, block, lds, stream, A);
    else
        hipLaunchKernelGGL((k_assemble<31, true>), grid, block, lds, stream, A);
    return hipGetLastError();
}


// ---- the Bessel helper alone (tests / tooling) ---------------------------------------------------
// util::bessel_i_alter_helper (include/functions.h:381-408) as the fill kernels evaluate it:
// out = {y0, y1, mu + y0, Re z < 0 ? z : -z} per argument.
namespace {
__global__ void k_bessel_probe(const double2* z, int n, double2* out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const cd zz = mk(z[k].x, z[k].y);
    const double zabs = sqrt(norm2(zz));
    cd y0, y1, mutot;
    bessel_miller(rcp(zz), zabs, 1.0 / zabs, zz.x < 0.0, y0, y1, mutot);
    out[4 * k + 0] = make_double2(y0.x, y0.y);
    out[4 * k + 1] = make_double2(y1.x, y1.y);
    out[4 * k + 2] = make_double2(mutot.x, mutot.y);
    out[4 * k + 3] = zz.x < 0.0 ? make_double2(zz.x, zz.y) : make_double2(-zz.x, -zz.y);
}
}  // namespace
hipError_t launch_bessel_probe(const double* z, int n, double* out, hipStream_t stream) {
    hipLaunchKernelGGL(k_bessel_probe, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, (const double2*)z, n,
                       (double2*)out);
    return hipGetLastError();
}

}  // namespace emme
