// assemble_cached.hip -- batched fill from an HBM-resident cache of the omega-independent
// node data.
//
// emme_device.hpp::node_data shows that, for a pair (i,j), moment m and contour sense, the
// integrand at a quadrature node is exp(A0 + T w)(w Q1 + Q0) with A0, T, Q1, Q0 independent
// of omega -- and of the Newton iteration.  A root search assembles the same parameter set
// hundreds of times (23 launches x 128 omegas in the bench), so those records are computed
// ONCE per context for every interval of the bisection tree down to depth D and kept in HBM
// (12.8 GB for N = 256, D = 8: this is what 288 GB of HBM3E buys).  Building the whole cache
// costs less than one assembly launch of the on-the-fly kernels.
//
// With the records cached nothing ties the omegas of a batch together any more: every LANE
// is an independent worker that owns one omega and walks that omega's own adaptive tree
// (identical decisions to the reference), fetching 15 (31) node records per interval and
// finishing each with one complex exponential and three complex products.  No idle lanes, no
// cross-lane reduction, summation in the reference's order.
//   * k_node_cache   fills the cache: one 16/32-lane group per (item, interval), lane = node.
//   * k_assemble_cached  the fill (electrostatic: one integral per pair).  An integral that
//     needs an interval outside the cache (rare: strongly damped omegas) is handed over whole,
//     through a device-side work list, to the cooperative kernel (assemble.hip,
//     k_assemble_coop: a workgroup per integral), which starts it over.
//   * k_assemble_cached_em  the electromagnetic fill: the three moments of a pair share one
//     record per node and one walk over the union of their trees.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "assemble_common.hpp"
#include "launch.hpp"
#include "node_cache.hpp"

namespace emme {

namespace {

// [l, r] of interval (depth, path) with the reference's midpoint sequence
__device__ __forceinline__ void interval_bounds(int depth, unsigned long long path, double& l,
                                                double& r) {
    l = 0.0;
    r = M_PI / 2.0;
    for (int s = depth - 1; s >= 0; --s) {
        const double mid = (r + l) / 2;
        if ((path >> s) & 1)
            l = mid;
        else
            r = mid;
    }
}

struct CacheArgs {
    DevParams P;
    const double* tab;
    const ushort2* pairs;
    int npairs;
    CacheGeom geom;
    double omi;      // class being built
    NodeRec* recs;   // [nitems][count][GW]: the cached intervals first .. first+count-1
    double2* ttab;   // [NI][GW]  T per (interval, node), all regions
    double2* wtab;   // null, or [NI][GW] moment factor W: records are per PAIR (moment 0)
    double* scale;   // [NI] half-width (r - l)/2 of every cached interval
    int part;          // -1: main buffer (full tree + subtree 0); k >= 0: subtree k+1
    int first, count;  // global slot of the first interval and number of intervals filled
    int folded;        // records in the FOLDED form (see NodeRec in node_cache.hpp)
};

template <int PTS>
__global__ __launch_bounds__(256) void k_node_cache(CacheArgs A) {
    constexpr int GW = PTS == 15 ? 16 : 32;
    constexpr int GROUPS_PER_BLOCK = 256 / GW;
    const DevParams& P = A.P;
    const int N = P.N;
    const int NI = A.count;  // intervals per item filled by this launch
    const int lane = threadIdx.x % GW;
    const int nm_items = A.wtab ? 1 : P.nm;  // shared layout: one item per pair
    const long nitems = (long)A.npairs * nm_items;
    const long total = nitems * NI;
    const GkLane gk = gk_lane<PTS>(lane);
    const double* eta = A.tab;
    const double* gtab = A.tab + N;
    const double* btab = A.tab + 2 * N;
    for (long w = (long)blockIdx.x * GROUPS_PER_BLOCK + threadIdx.x / GW; w < total;
         w += (long)gridDim.x * GROUPS_PER_BLOCK) {
        const long item = w / NI;
        const int idx = (int)(w - item * NI);
        const int p = (int)(item / nm_items);
        const int m = (int)(item - (long)p * nm_items);
        const ushort2 ij = A.pairs[p];
        const int i = ij.x, j = ij.y;
        const PairConst pc = make_pair_const(P, eta[i], eta[j], btab[i], btab[j], gtab[i] - gtab[j]);
        int depth;
        unsigned long long path;
        A.geom.interval(A.part, idx, depth, path);
        double l, r;
        interval_bounds(depth, path, l, r);
        const double mid = (r + l) / 2, scale = (r - l) / 2;
        const double x = __dadd_rn(__dmul_rn(scale, gk.x), mid);
        const NodeData d = node_data(x, P, pc, A.omi, m);
        NodeRec rec;
        if (A.folded) {
            // exp(A0) goes into the amplitudes once, here; what remains per (omega, node) is
            // exp(T omega), which does not depend on the pair and comes from the phase table
            double sa, ca;
            sincos(d.A0.y, &sa, &ca);
            // (no cap on Re A0: where exp(A0) or exp(A0) Q is not representable -- a near-pole of 1/lambda -- the
            // amplitudes are stored non-finite; a node the reference's clamp zeroes stays zero (the fills select),
            // any other use ends in a non-finite integral and flags the matrix instead of a silently wrong value)
            const double ea = exp(d.A0.x);
            const cd ex = mk(ea * ca, ea * sa);
            const cd q1 = ex * d.Q1, q0 = ex * d.Q0;
            rec.A0 = make_double2(exp(fmin(2.0 * d.A0.x, 700.0)), d.A0.x);  // (|exp A0|^2, Re A0)
            rec.Q1 = make_double2(q1.x, q1.y);
            rec.Q0 = make_double2(q0.x, q0.y);
        } else {
            rec.A0 = make_double2(d.A0.x, d.A0.y);
            rec.Q1 = make_double2(d.Q1.x, d.Q1.y);
            rec.Q0 = make_double2(d.Q0.x, d.Q0.y);
        }
        A.recs[w * GW + lane] = rec;
        if (item == 0) {
            A.ttab[(long)(A.first + idx) * GW + lane] = make_double2(d.T.x, d.T.y);
            if (A.wtab) {
                const cd w = node_w(x, P, A.omi);
                A.wtab[(long)(A.first + idx) * GW + lane] = make_double2(w.x, w.y);
            }
            if (lane == 0) A.scale[A.first + idx] = scale;
        }
    }
}

// Phase table: E(interval, node, omega) = exp(T omega) for every cached interval and every omega
// of a launch.  T = i t~ depends on the abscissa only, so this factor of the integrand is the same
// for all N(N-1)/2 pairs: computed once per launch here (7 M complex exponentials for 128
// omegas) instead of once per (pair, node, omega) in the fill (930 M).  Layout
// [interval][node][omega position], so the omegas of a lane group read consecutive entries.
struct PhaseArgs {
    const double2* ttab[2];
    const double2* omega;
    const int* act_idx;
    int n_act;
    int nrows;  // cached intervals x GW
    double2* etab;
};
__global__ __launch_bounds__(256) void k_phase_table(PhaseArgs A) {
    const long total = (long)A.nrows * A.n_act;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / A.n_act;
        const int w = (int)(e - row * A.n_act);
        const double2 om = A.omega[A.act_idx[w]];
        const int cls = -copysign(1.0, om.x) > 0.0 ? 0 : 1;
        double2 ev = make_double2(0.0, 0.0);
        if (A.ttab[cls]) {
            const double2 t = A.ttab[cls][row];
            const double ax = fma(t.x, om.x, -(t.y * om.y)), ay = fma(t.x, om.y, t.y * om.x);
            if (!(ax > 700.0)) {  // (a NaN omega goes through and poisons the integral, which flags it)
                double sa, ca;
                sincos(ay, &sa, &ca);
                const double ea = exp(ax);
                ev = make_double2(ea * ca, ea * sa);
            } else {
                // beyond 1e304 the reference's exp(A0 + T omega) overflows or is about to: NaN, so that an integral
                // which uses this node flags its matrix (EMME_ENUMERIC) instead of dropping the term
                ev = make_double2(__builtin_nan(""), __builtin_nan(""));
            }
        }
        A.etab[e] = ev;
    }
}

struct AsmCachedArgs {
    DevParams P;
    const double* tab;
    const ushort2* pairs;
    int npairs;
    CacheGeom geom;
    const NodeRec* recs[2];   // main part per contour class (omi = +1, -1); null if not built
    const NodeRec* recs_ext[2][NODE_CACHE_MAX_SUB - 1];  // run-time subtrees; null if absent
    const double2* ttab[2];   // [NI][GW] per class
    const double2* wtab[2];   // [NI][GW] per class: moment factor W (shared EM layout only)
    const double2* etab;      // phase table of this launch (records are folded), or null
    const double* scale;      // [NI]
    unsigned long long* worklist;   // deferred integrals: batch << 32 | item
    unsigned long long* defer_info; // depth << 56 | path of the interval each entry was missing
    unsigned int* worklist_count;
    const int* act_idx;      // batch indices, chunk after chunk
    const int2* chunks;      // per omega chunk (blockIdx.y): (first position in act_idx, size <= GW)
    int n_act;
    const double2* omega;
    double2* M;
    const double2* Mold;
    double2* Mp;
    const double2* domega;
    unsigned long long* intervals;
    int* status;
    int skip_lost;  // omegas whose matrix is already flagged (status) are left alone
};

#ifndef EMME_CACHED_MIN_WAVES
#define EMME_CACHED_MIN_WAVES 3
#endif

template <int PTS, bool FOLDED>
__global__ __launch_bounds__(256, EMME_CACHED_MIN_WAVES) void k_assemble_cached(AsmCachedArgs A) {
    constexpr int GW = PTS == 15 ? 16 : 32;
    constexpr int H = (PTS + 1) / 2;
#ifndef EMME_FOLDED_UNROLL
#define EMME_FOLDED_UNROLL 5
#endif
    constexpr int NODE_UNROLL = PTS == 15 ? (FOLDED ? EMME_FOLDED_UNROLL : 5) : 3;  // trips of the 3-node loop unrolled
    constexpr int GROUPS_PER_BLOCK = 256 / GW;
    extern __shared__ double lds_raw[];  // eta | g | b
    __shared__ unsigned long long s_iv[GW];  // interval counts of the chunk's omegas (block_add_intervals)
    if (threadIdx.x < GW) s_iv[threadIdx.x] = 0ull;

    const DevParams& P = A.P;
    const int N = P.N, dim = P.dim;
    for (int k = threadIdx.x; k < 3 * N; k += blockDim.x) lds_raw[k] = A.tab[k];
    __syncthreads();
    const double* eta = lds_raw;
    const double* gtab = lds_raw + N;
    const double* btab = lds_raw + 2 * N;
    const int group_in_block = threadIdx.x / GW;
    const int lane = threadIdx.x % GW;

    // lane -> (omega slot, item stream).  A chunk holds up to GW omegas; when it holds fewer
    // (small batches, the tail of a root search) the spare lanes take further item streams of
    // the same omegas, so a single omega still fills all lanes.
    const int2 chunk = A.chunks[blockIdx.y];
    const int n_in_chunk = chunk.y;
    int n_eff = 1;
    while (n_eff < n_in_chunk) n_eff <<= 1;
    const int nsub = GW / n_eff;
    const int wslot = lane % n_eff, sub = lane / n_eff;
    const bool in_chunk = wslot < n_in_chunk;
    const int wpos = in_chunk ? chunk.x + wslot : 0;  // position in the launch's omega list
    const int b = in_chunk ? A.act_idx[wpos] : 0;
    const bool has_w = in_chunk && !(A.skip_lost && A.status[b] != 0);  // (a lost matrix: see k_assemble_dense)
    cd omega = mk(0.0, 0.0), rdw = mk(0.0, 0.0);
    if (has_w) {
        omega = mk(A.omega[b].x, A.omega[b].y);
        if (A.Mold) rdw = rcp(mk(A.domega[b].x, A.domega[b].y));
    }
    const int cls = -copysign(1.0, omega.x) > 0.0 ? 0 : 1;
    const NodeRec* recs = A.recs[cls];
    const double2* ttab = A.ttab[cls];
    const int NI_MAIN = A.geom.ni_main();
    double2* Mb = A.M + (size_t)b * dim * dim;
    const double2* Moldb = A.Mold ? A.Mold + (size_t)b * dim * dim : nullptr;
    double2* Mpb = A.Mp ? A.Mp + (size_t)b * dim * dim : nullptr;
    auto store = [&](int r, int c, cd v) {
        const size_t idx = (size_t)r * dim + c;
        Mb[idx] = make_double2(v.x, v.y);
        if (Moldb) {
            const double2 o = Moldb[idx];
            const cd d = (v - mk(o.x, o.y)) * rdw;
            Mpb[idx] = make_double2(d.x, d.y);
        }
    };
    if (blockIdx.x == 0 && has_w && sub == 0) {  // diagonal (include/solver.h:442-443, 465-470)
        for (int i = group_in_block; i < N; i += GROUPS_PER_BLOCK) {
            store(i, i, mk(P.diag_a, 0.0));
            if (P.nm == 3) {
                store(i, i + N, mk(0.0, 0.0));
                store(i + N, i, mk(0.0, 0.0));
                store(i + N, i + N, mk(P.diag_d * btab[i], 0.0));
            }
        }
    }

    const double* WK = PTS == 15 ? kWk15 : kWk31;
    const double* WG = PTS == 15 ? kWg15 : kWg31;
    const double inv_scale = 2. / (M_PI / 2.0);
    const int nitems = A.npairs * P.nm;
    const int NI = A.geom.ni();
    const double* scale_tab = lds_raw + 3 * N;  // staged below
    for (int k = threadIdx.x; k < NI; k += blockDim.x) lds_raw[3 * N + k] = A.scale[k];
    __syncthreads();
    const int worker = (blockIdx.x * GROUPS_PER_BLOCK + group_in_block) * nsub + sub;
    const int nworkers = gridDim.x * GROUPS_PER_BLOCK * nsub;

    const TransConsts TC = trans_consts();  // polynomial coefficients, once, in scalar registers
    unsigned long long my_intervals = 0;
    int bad = 0;
    // Item loop: the lanes of a wave take their next items together and walk their own trees
    // until every lane has finished its integral (lanes that finish early wait: the omegas of
    // a chunk are sorted by cost, so their trees have similar sizes); the scatter of the
    // results then runs once for all lanes instead of diverging on every interval.
    for (int item = worker; __ballot(has_w && item < nitems) != 0ull; item += nworkers) {
        const bool mine = has_w && item < nitems;
        int i = 0, j = 0, m = 0;
        if (mine) {
            const int p = item / P.nm;
            m = item - p * P.nm;
            const ushort2 ij = A.pairs[p];
            i = ij.x, j = ij.y;
        }
        int depth = 0;
        unsigned long long path = 0;
        double abs_tol = 0.0;
        cd sum = mk(0.0, 0.0);
        int item_intervals = 0;
        bool walking = mine, deferred = false;

        while (walking) {
            int which;
            const int cslot = A.geom.slot(depth, path, which);
            const NodeRec* ebuf = which >= 0 ? A.recs_ext[cls][which] : recs;
            if (cslot < 0 || ebuf == nullptr) {
                // outside the cache: hand the whole integral to the cooperative kernel
                const unsigned int slot = atomicAdd(A.worklist_count, 1u);
                // bits 58..63: depth, bits 17..57: low path bits (diagnostics only; the list
                // kernel masks them off), bits 32..47 would collide with b, so b sits at 17+24
                A.worklist[slot] = ((unsigned long long)b << 32) | (unsigned int)item;
                A.defer_info[slot] = ((unsigned long long)depth << 56) | ((unsigned long long)cls << 55) |
                                     (path & 0x7fffffffffffffull);  // depth | contour class | path
                deferred = true;
                walking = false;
                continue;
            }
            const double scale = scale_tab[cslot];
            // ---- 15 (31) records of my interval, reference summation order -------------
            // (include/functions.h:186-201: centre, then f(+x_q) + f(-x_q) for q = 1..H-1)
            const NodeRec* rp =
                which < 0 ? recs + ((long)item * NI_MAIN + cslot) * GW
                          : ebuf + ((long)item * A.geom.ni_sub(which + 1) + (cslot - A.geom.base[which + 1])) * GW;
            // per node: T (shared by all omegas) or, folded, this omega's exp(T omega)
            const double2* tp = FOLDED ? A.etab + (long)cslot * GW * A.n_act + wpos : ttab + (long)cslot * GW;
            const int tstride = FOLDED ? A.n_act : 1;
            cd K = mk(0.0, 0.0), G = mk(0.0, 0.0), fplus = mk(0.0, 0.0);
            // visiting order s = 0..PTS-1: centre, +x_1, -x_1, +x_2, -x_2, ...
            auto node_of = [&](int s) { return s == 0 ? 0 : ((s & 1) ? (s + 1) >> 1 : (s >> 1) + H - 1); };
            auto proc = [&](const NodeRec& rec, const double2 tt, int s) {
                cd f;
                if (FOLDED) {
                    // folded record (|exp A0|^2, Re A0, exp(A0) Q1, exp(A0) Q0) and tt = exp(T omega)
                    // from the phase table: safe_exp clamp (src/Parameters.cpp:167-173) as
                    // |exp(A0 + T omega)|^2 < exp(-80), then F = E (omega Q1 + Q0)
                    const bool clamped = fma(tt.x, tt.x, tt.y * tt.y) * rec.A0.x < 1.8048513878454153e-35;
                    const cd S = mk(fma(omega.x, rec.Q1.x, fma(-omega.y, rec.Q1.y, rec.Q0.x)),
                                    fma(omega.x, rec.Q1.y, fma(omega.y, rec.Q1.x, rec.Q0.y)));
                    const cd ef = mk(tt.x, tt.y) * S;
                    f = clamped ? mk(0.0, 0.0) : ef;
                } else {
                    NodeData d;
                    d.A0 = mk(rec.A0.x, rec.A0.y);
                    d.T = mk(tt.x, tt.y);
                    d.Q1 = mk(rec.Q1.x, rec.Q1.y);
                    d.Q0 = mk(rec.Q0.x, rec.Q0.y);
                    f = node_eval(d, omega, TC);
                }
                if (s & 1) {
                    fplus = f;
                } else {
                    const int q = s >> 1;
                    const cd fs = s == 0 ? f : fplus + f;
                    K = K + WK[q] * fs;
                    if ((q & 1) == 0) G = G + WG[q >> 1] * fs;
                }
            };
            // three record buffers rotate so that two record loads are always in flight
            // while a third record is being evaluated; for GK15 the loop is fully unrolled (node
            // order and record offsets become immediates: ~10 % faster; GK31 would spill)
            NodeRec r0 = rp[node_of(0)], r1 = rp[node_of(1)], r2;
            double2 t0 = tp[node_of(0) * tstride], t1 = tp[node_of(1) * tstride], t2;
#pragma unroll NODE_UNROLL
            for (int s = 0; s < PTS; s += 3) {
                if (s + 2 < PTS) r2 = rp[node_of(s + 2)], t2 = tp[node_of(s + 2) * tstride];
                proc(r0, t0, s);
                if (s + 3 < PTS) r0 = rp[node_of(s + 3)], t0 = tp[node_of(s + 3) * tstride];
                if (s + 1 < PTS) proc(r1, t1, s + 1);
                if (s + 4 < PTS) r1 = rp[node_of(s + 4)], t1 = tp[node_of(s + 4) * tstride];
                if (s + 2 < PTS) proc(r2, t2, s + 2);
            }
            ++item_intervals;
            // include/functions.h:203-208, 231-247
            const double dKx = K.x - G.x, dKy = K.y - G.y;
            const double absK = sqrt(fma(K.x, K.x, K.y * K.y));
            double err = fmax(sqrt(fma(dKx, dKx, dKy * dKy)), absK * (2.0 * 2.220446049250313e-16));
            const cd integral = mk(K.x * scale, K.y * scale);
            err *= scale;
            const double rel_abs = P.rel_tol * (absK * scale);
            if (abs_tol == 0.0) abs_tol = rel_abs;
            bool split = depth < P.max_sub && err > abs_tol * inv_scale + P.prec_goal &&
                         err > rel_abs + P.prec_goal;
            if (split && (depth >= EMME_MAX_DEPTH || item_intervals >= EMME_MAX_INTERVALS)) {
                split = false;
                bad = 1;
            }
            if (split) {
                ++depth;
                path <<= 1;
            } else {
                sum = sum + integral;
                ++path;
                // strip trailing zeros: up to the first ancestor that is a left child
                const int tz = min(depth, (int)__builtin_ctzll(path | (1ull << 63)));
                path >>= tz;
                depth -= tz;
                walking = depth != 0;
            }
        }

        // ---- results of this round of items, all lanes together -------------------------
        if (mine && !deferred) {
            my_intervals += (unsigned long long)item_intervals;
            const double dg = gtab[i] - gtab[j], de = eta[i] - eta[j];
            cd kap = mk(P.pref * sum.y, -(P.pref * sum.x));  // -i pref sum, Parameters.cpp:182
            if (kappa_bad(kap)) bad = 1;
            kap = kap + kappa_e(m, P, de, dg, omega);
            if (m == 0) {
                const cd v = (-(pair_weight(i, j, N) * P.dx)) * kap;
                store(i, j, v);
                store(j, i, v);
            } else if (m == 1) {
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
    }

    block_add_intervals(s_iv, A.intervals, wslot, has_w, has_w && group_in_block == 0 && sub == 0, b, my_intervals);
    if (has_w && bad) A.status[b] = 1;
}

// ---- electromagnetic fill on the shared cache layout ---------------------------------------------
// The three integrals of a pair (moments m = 0, 1, 2: blocks A, B, D of the matrix,
// include/solver.h:461-511) differ only by the factor norm_vel^m = (c_nv W)^m in the integrand, and
// their adaptive trees are nearly the same.  A lane therefore walks the UNION of the three trees
// of its (pair, omega): per node one record, one complex exponential, then f1 = f0 nv, f2 = f1 nv;
// each moment keeps its own Kronrod/Gauss sums, its own abs_tol and its own accept/split decisions
// (the reference's, moment by moment), and is evaluated exactly on the intervals of its own tree.
// The walk order is the pre-order of the union: every moment holds the key of the next interval
// it needs, key = position of the left end (path << (56 - depth)) * 64 + depth, and the lane
// always takes the smallest key.  ~2.9x fewer intervals and record loads than three separate walks.
template <int PTS, bool FOLDED>
__global__ __launch_bounds__(256, 2) void k_assemble_cached_em(AsmCachedArgs A) {
    constexpr int GW = PTS == 15 ? 16 : 32;
    constexpr int H = (PTS + 1) / 2;
    constexpr int NODE_UNROLL = PTS == 15 ? 5 : 3;  // trips of the 3-node loop unrolled
    constexpr int GROUPS_PER_BLOCK = 256 / GW;
    constexpr int KD = 56;  // key layout: depth <= KD (>= EMME_MAX_DEPTH + 1)
    extern __shared__ double lds_raw[];  // eta | g | b | scale table
    __shared__ unsigned long long s_iv[GW];  // interval counts of the chunk's omegas (block_add_intervals)
    if (threadIdx.x < GW) s_iv[threadIdx.x] = 0ull;

    const DevParams& P = A.P;
    const int N = P.N, dim = P.dim;
    for (int k = threadIdx.x; k < 3 * N; k += blockDim.x) lds_raw[k] = A.tab[k];
    const int NI = A.geom.ni();
    for (int k = threadIdx.x; k < NI; k += blockDim.x) lds_raw[3 * N + k] = A.scale[k];
    __syncthreads();
    const double* eta = lds_raw;
    const double* gtab = lds_raw + N;
    const double* btab = lds_raw + 2 * N;
    const double* scale_tab = lds_raw + 3 * N;
    const int group_in_block = threadIdx.x / GW;
    const int lane = threadIdx.x % GW;

    const int2 chunk = A.chunks[blockIdx.y];
    const int n_in_chunk = chunk.y;
    int n_eff = 1;
    while (n_eff < n_in_chunk) n_eff <<= 1;
    const int nsub = GW / n_eff;
    const int wslot = lane % n_eff, sub = lane / n_eff;
    const bool in_chunk = wslot < n_in_chunk;
    const int wpos = in_chunk ? chunk.x + wslot : 0;  // position in the launch's omega list
    const int b = in_chunk ? A.act_idx[wpos] : 0;
    const bool has_w = in_chunk && !(A.skip_lost && A.status[b] != 0);  // (a lost matrix: see k_assemble_dense)
    cd omega = mk(0.0, 0.0), rdw = mk(0.0, 0.0);
    if (has_w) {
        omega = mk(A.omega[b].x, A.omega[b].y);
        if (A.Mold) rdw = rcp(mk(A.domega[b].x, A.domega[b].y));
    }
    const int cls = -copysign(1.0, omega.x) > 0.0 ? 0 : 1;
    const NodeRec* recs = A.recs[cls];
    const double2* ttab = A.ttab[cls];
    const double2* wtab = A.wtab[cls];
    const int NI_MAIN = A.geom.ni_main();
    double2* Mb = A.M + (size_t)b * dim * dim;
    const double2* Moldb = A.Mold ? A.Mold + (size_t)b * dim * dim : nullptr;
    double2* Mpb = A.Mp ? A.Mp + (size_t)b * dim * dim : nullptr;
    auto store = [&](int r, int c, cd v) {
#ifdef EMME_EM_NO_STORE  // timing experiment only (results are garbage): what the scattered 16-byte stores cost
        if (v.x != 1.2345e300) return;
#endif
        const size_t idx = (size_t)r * dim + c;
        Mb[idx] = make_double2(v.x, v.y);
        if (Moldb) {
            const double2 o = Moldb[idx];
            const cd d = (v - mk(o.x, o.y)) * rdw;
            Mpb[idx] = make_double2(d.x, d.y);
        }
    };
    if (blockIdx.x == 0 && has_w && sub == 0) {  // diagonal (include/solver.h:465-470)
        for (int i = group_in_block; i < N; i += GROUPS_PER_BLOCK) {
            store(i, i, mk(P.diag_a, 0.0));
            store(i, i + N, mk(0.0, 0.0));
            store(i + N, i, mk(0.0, 0.0));
            store(i + N, i + N, mk(P.diag_d * btab[i], 0.0));
        }
    }

    const double* WK = PTS == 15 ? kWk15 : kWk31;
    const double* WG = PTS == 15 ? kWg15 : kWg31;
    const double inv_scale = 2. / (M_PI / 2.0);
    const int nitems = A.npairs;  // an item is a pair: its three moments travel together
    const int worker = (blockIdx.x * GROUPS_PER_BLOCK + group_in_block) * nsub + sub;
    const int nworkers = gridDim.x * GROUPS_PER_BLOCK * nsub;
    auto make_key = [](int depth, unsigned long long path) -> unsigned long long {
        return ((path << (KD - depth)) << 6) | (unsigned long long)depth;
    };
    const unsigned long long DONE = ~0ull;

    const TransConsts TC = trans_consts();  // polynomial coefficients, once, in scalar registers
    unsigned long long my_intervals = 0;
    int bad = 0;
    for (int item = worker; __ballot(has_w && item < nitems) != 0ull; item += nworkers) {
        const bool mine = has_w && item < nitems;
        int i = 0, j = 0;
        double c_nv = 0.0;
        if (mine) {
            const ushort2 ij = A.pairs[item];
            i = ij.x, j = ij.y;
            c_nv = (P.qR * (eta[i] - eta[j])) / P.vt;  // as make_pair_const
        }
        unsigned long long key[3];
        double abs_tol[3];
        cd sum[3];
        int count[3];
        bool deferred[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            key[m] = mine ? 0ull : DONE;  // root: depth 0, path 0
            abs_tol[m] = 0.0, sum[m] = mk(0.0, 0.0), count[m] = 0, deferred[m] = false;
        }

        for (;;) {
            unsigned long long cur = key[0] < key[1] ? key[0] : key[1];
            cur = key[2] < cur ? key[2] : cur;
            if (cur == DONE) break;
            const int depth = (int)(cur & 63ull);
            const unsigned long long path = (cur >> 6) >> (KD - depth);
            int which;
            const int cslot = A.geom.slot(depth, path, which);
            const NodeRec* ebuf = which >= 0 ? A.recs_ext[cls][which] : recs;
            if (cslot < 0 || ebuf == nullptr) {
                // outside the cache: the moments that need this interval go, whole, to the
                // cooperative kernel; the others carry on
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    if (key[m] == cur) {
                        const unsigned int slot = atomicAdd(A.worklist_count, 1u);
                        A.worklist[slot] = ((unsigned long long)b << 32) | (unsigned int)(item * 3 + m);
                        A.defer_info[slot] = ((unsigned long long)depth << 56) | ((unsigned long long)cls << 55) |
                                     (path & 0x7fffffffffffffull);  // depth | contour class | path
                        deferred[m] = true;
                        key[m] = DONE;
                    }
                }
                continue;
            }
            const double scale = scale_tab[cslot];
            const long off = (long)cslot * GW;
            const NodeRec* rp =
                which < 0 ? recs + ((long)item * NI_MAIN + cslot) * GW
                          : ebuf + ((long)item * A.geom.ni_sub(which + 1) + (cslot - A.geom.base[which + 1])) * GW;
            const double2* tp = FOLDED ? A.etab + off * A.n_act + wpos : ttab + off;
            const int tstride = FOLDED ? A.n_act : 1;
            const double2* wp = wtab + off;
            cd K[3], G[3], fplus[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) K[m] = G[m] = fplus[m] = mk(0.0, 0.0);
            // visiting order s = 0..PTS-1: centre, +x_1, -x_1, +x_2, -x_2, ... (the reference's
            // summation order, include/functions.h:186-201)
            auto node_of = [&](int s) { return s == 0 ? 0 : ((s & 1) ? (s + 1) >> 1 : (s >> 1) + H - 1); };
            auto proc = [&](const NodeRec& rec, const double2 tt, const double2 ww, int s) {
                cd f[3];
                if (FOLDED) {
                    const bool clamped = fma(tt.x, tt.x, tt.y * tt.y) * rec.A0.x < 1.8048513878454153e-35;
                    const cd S = mk(fma(omega.x, rec.Q1.x, fma(-omega.y, rec.Q1.y, rec.Q0.x)),
                                    fma(omega.x, rec.Q1.y, fma(omega.y, rec.Q1.x, rec.Q0.y)));
                    const cd ef = mk(tt.x, tt.y) * S;
                    f[0] = clamped ? mk(0.0, 0.0) : ef;
                } else {
                    NodeData d;
                    d.A0 = mk(rec.A0.x, rec.A0.y);
                    d.T = mk(tt.x, tt.y);
                    d.Q1 = mk(rec.Q1.x, rec.Q1.y);
                    d.Q0 = mk(rec.Q0.x, rec.Q0.y);
                    f[0] = node_eval(d, omega, TC);
                }
                const cd nv = c_nv * mk(ww.x, ww.y);
                f[1] = f[0] * nv;
                f[2] = f[1] * nv;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    if (s & 1) {
                        fplus[m] = f[m];
                    } else {
                        const int q = s >> 1;
                        const cd fs = s == 0 ? f[m] : fplus[m] + f[m];
                        K[m] = K[m] + WK[q] * fs;
                        if ((q & 1) == 0) G[m] = G[m] + WG[q >> 1] * fs;
                    }
                }
            };
            // two record loads in flight while a third record is being evaluated
            NodeRec r0 = rp[node_of(0)], r1 = rp[node_of(1)], r2;
            double2 t0 = tp[node_of(0) * tstride], t1 = tp[node_of(1) * tstride], t2;
            double2 w0 = wp[node_of(0)], w1 = wp[node_of(1)], w2;
#pragma unroll NODE_UNROLL
            for (int s = 0; s < PTS; s += 3) {
                if (s + 2 < PTS) r2 = rp[node_of(s + 2)], t2 = tp[node_of(s + 2) * tstride], w2 = wp[node_of(s + 2)];
                proc(r0, t0, w0, s);
                if (s + 3 < PTS) r0 = rp[node_of(s + 3)], t0 = tp[node_of(s + 3) * tstride], w0 = wp[node_of(s + 3)];
                if (s + 1 < PTS) proc(r1, t1, w1, s + 1);
                if (s + 4 < PTS) r1 = rp[node_of(s + 4)], t1 = tp[node_of(s + 4) * tstride], w1 = wp[node_of(s + 4)];
                if (s + 2 < PTS) proc(r2, t2, w2, s + 2);
            }
            // per moment: include/functions.h:203-208, 231-247
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if (key[m] != cur) continue;
                ++count[m];
                const double dKx = K[m].x - G[m].x, dKy = K[m].y - G[m].y;
                const double absK = sqrt(fma(K[m].x, K[m].x, K[m].y * K[m].y));
                double err = fmax(sqrt(fma(dKx, dKx, dKy * dKy)), absK * (2.0 * 2.220446049250313e-16));
                const cd integral = mk(K[m].x * scale, K[m].y * scale);
                err *= scale;
                const double rel_abs = P.rel_tol * (absK * scale);
                if (abs_tol[m] == 0.0) abs_tol[m] = rel_abs;
                bool split = depth < P.max_sub && err > abs_tol[m] * inv_scale + P.prec_goal &&
                             err > rel_abs + P.prec_goal;
                if (split && (depth >= EMME_MAX_DEPTH || count[m] >= EMME_MAX_INTERVALS)) {
                    split = false;
                    bad = 1;
                }
                if (split) {
                    key[m] = make_key(depth + 1, path << 1);
                } else {
                    sum[m] = sum[m] + integral;
                    unsigned long long p2 = path + 1;
                    const int tz = min(depth, (int)__builtin_ctzll(p2 | (1ull << 63)));
                    p2 >>= tz;
                    const int d2 = depth - tz;
                    key[m] = d2 == 0 ? DONE : make_key(d2, p2);
                }
            }
        }

        // ---- results: blocks A (m = 0), B and its mirrors (m = 1), D (m = 2) --------------------
        if (mine) {
            const double dg = gtab[i] - gtab[j], de = eta[i] - eta[j];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if (deferred[m]) continue;
                my_intervals += (unsigned long long)count[m];
                cd kap = mk(P.pref * sum[m].y, -(P.pref * sum[m].x));  // -i pref sum, Parameters.cpp:182
                if (kappa_bad(kap)) bad = 1;
                kap = kap + kappa_e(m, P, de, dg, omega);
                if (m == 0) {
                    const cd v = (-(pair_weight(i, j, N) * P.dx)) * kap;
                    store(i, j, v);
                    store(j, i, v);
                } else if (m == 1) {
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
        }
    }

    block_add_intervals(s_iv, A.intervals, wslot, has_w, has_w && group_in_block == 0 && sub == 0, b, my_intervals);
    if (has_w && bad) A.status[b] = 1;
}

// ---- union walk on folded records + phase table (electrostatic, GK15) ------------------------------
// The independent-lane kernel above issues 60 loads per lane and interval at per-lane addresses
// (a wave-level load touches ~21 L1 lines): once the phase table has taken the transcendental
// work away, that memory-pipeline work is what bounds it.  Here the 16 lanes of a group stay on
// ONE interval at a time -- the union of their omegas' trees in pre-order, like k_assemble_wl --
//   phase 1  lane = node : one coalesced load of the interval's 16 records (768 B) into LDS;
//   phase 2  lane = omega: every lane whose own tree contains the interval walks the 15 nodes,
//            records broadcast from LDS, exp(T omega) from the phase table (16 consecutive
//            entries per node for the group), ~10 flops per node; own Kronrod/Gauss sums, own
//            abs_tol, own accept/split decision (the reference's, omega by omega).
// Lanes whose tree does not contain the interval sit the round out (cheap now: a round is
// mostly memory).  Each lane holds the pre-order key of the next interval it needs; the group
// takes the minimum (DPP row reduction).  Intervals outside the cache defer the integrals that
// need them to the cooperative kernel.
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_min_step_u64(unsigned long long x) {
    const unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o < x ? o : x;
}
__device__ __forceinline__ unsigned long long row16_min_u64(unsigned long long v) {
    v = dpp_min_step_u64<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_min_step_u64<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_min_step_u64<0x141>(v);  // row_half_mirror
    v = dpp_min_step_u64<0x140>(v);  // row_mirror
    return v;
}

template <int PTS, int NSEL>
__global__ __launch_bounds__(256, 4) void k_assemble_union(AsmCachedArgs A) {
    static_assert(PTS == 15, "lane groups of 16: GK15");
    static_assert(NSEL >= 1 && NSEL <= 4, "intervals served per round");
    constexpr int GW = 16;
    constexpr int H = (PTS + 1) / 2;
    constexpr int GROUPS_PER_BLOCK = 256 / GW;
    constexpr int KD = 56;
    extern __shared__ double lds_raw[];  // eta | g | b | per-group record slots
    __shared__ unsigned long long s_iv[GW];  // interval counts of the chunk's omegas (block_add_intervals)
    if (threadIdx.x < GW) s_iv[threadIdx.x] = 0ull;

    const DevParams& P = A.P;
    const int N = P.N, dim = P.dim;
    for (int k = threadIdx.x; k < 3 * N; k += blockDim.x) lds_raw[k] = A.tab[k];
    __syncthreads();
    const double* eta = lds_raw;
    const double* gtab = lds_raw + N;
    const int group_in_block = threadIdx.x / GW;
    const int lane = threadIdx.x % GW;
    double2* slots = reinterpret_cast<double2*>(lds_raw + 3 * N + ((3 * N) & 1)) + group_in_block * (NSEL * GW * 3);

    const int2 chunk = A.chunks[blockIdx.y];
    const bool in_chunk = lane < chunk.y;
    const int wpos = in_chunk ? chunk.x + lane : 0;
    const int b = in_chunk ? A.act_idx[wpos] : 0;
    const bool has_w = in_chunk && !(A.skip_lost && A.status[b] != 0);  // (a lost matrix: see k_assemble_dense)
    cd omega = mk(0.0, 0.0), rdw = mk(0.0, 0.0);
    if (has_w) {
        omega = mk(A.omega[b].x, A.omega[b].y);
        if (A.Mold) rdw = rcp(mk(A.domega[b].x, A.domega[b].y));
    }
    const int cls = -copysign(1.0, omega.x) > 0.0 ? 0 : 1;
    double2* Mb = A.M + (size_t)b * dim * dim;
    const double2* Moldb = A.Mold ? A.Mold + (size_t)b * dim * dim : nullptr;
    double2* Mpb = A.Mp ? A.Mp + (size_t)b * dim * dim : nullptr;
    auto store = [&](int r, int c, cd v) {
        const size_t idx = (size_t)r * dim + c;
        Mb[idx] = make_double2(v.x, v.y);
        if (Moldb) {
            const double2 o = Moldb[idx];
            const cd d = (v - mk(o.x, o.y)) * rdw;
            Mpb[idx] = make_double2(d.x, d.y);
        }
    };
    if (blockIdx.x == 0 && has_w)  // diagonal (include/solver.h:442-443)
        for (int i = group_in_block; i < N; i += GROUPS_PER_BLOCK) store(i, i, mk(P.diag_a, 0.0));

    const double* WK = kWk15;
    const double* WG = kWg15;
    const double inv_scale = 2. / (M_PI / 2.0);
    const int nitems = A.npairs;
    const int ngroups = gridDim.x * GROUPS_PER_BLOCK;
    auto make_key = [](int depth, unsigned long long path) -> unsigned long long {
        return ((path << (KD - depth)) << 6) | (unsigned long long)depth;
    };
    const unsigned long long DONE = ~0ull;
    unsigned long long my_intervals = 0;
    int bad = 0;
    for (int item = blockIdx.x * GROUPS_PER_BLOCK + group_in_block; item < nitems; item += ngroups) {
        const ushort2 ij = A.pairs[item];
        const int i = ij.x, j = ij.y;
        unsigned long long key = has_w ? 0ull : DONE;  // root: depth 0, path 0
        double abs_tol = 0.0;
        cd sum = mk(0.0, 0.0);
        int count = 0;
        bool deferred = false;
        for (;;) {
            // The NSEL smallest distinct keys of the group are served in one round (a group whose
            // omegas sit on both sides of the imaginary axis walks contour class 0 first: bit 63
            // of the group key, the records differ).  With NSEL = 1 this is the plain union walk;
            // NSEL > 1 lets lanes whose trees have diverged advance in the same round -- each
            // lane still needs nothing but its own next key.
            const unsigned long long mykey = key == DONE ? DONE : (key | ((unsigned long long)cls << 63));
            unsigned long long ksel[NSEL];
            ksel[0] = row16_min_u64(mykey);
            if (ksel[0] == DONE) break;  // group-uniform
#pragma unroll
            for (int q = 1; q < NSEL; ++q)
                ksel[q] = ksel[q - 1] == DONE ? DONE : row16_min_u64(mykey > ksel[q - 1] ? mykey : DONE);
            int sel = -1;  // which of the served intervals is this lane's
#pragma unroll
            for (int q = 0; q < NSEL; ++q)
                if (mykey == ksel[q] && mykey != DONE) sel = q;
            int my_cslot = -1;
            // ---- phase 1: lane = node, one coalesced 768-byte read per served interval ----------
#pragma unroll
            for (int q = 0; q < NSEL; ++q) {
                if (ksel[q] == DONE) continue;  // group-uniform
                const int ccls = (int)(ksel[q] >> 63);
                const unsigned long long cur = ksel[q] & ~(1ull << 63);
                const int depth = (int)(cur & 63ull);
                const unsigned long long path = (cur >> 6) >> (KD - depth);
                int which;
                const int cslot = A.geom.slot(depth, path, which);
                const NodeRec* ebuf = which >= 0 ? A.recs_ext[ccls][which] : A.recs[ccls];
                if (cslot < 0 || ebuf == nullptr) {
                    if (sel == q) {  // outside the cache: this omega's integral goes to the cooperative kernel
                        const unsigned int slot = atomicAdd(A.worklist_count, 1u);
                        A.worklist[slot] = ((unsigned long long)b << 32) | (unsigned int)item;
                        A.defer_info[slot] = ((unsigned long long)depth << 56) | ((unsigned long long)cls << 55) |
                                             (path & 0x7fffffffffffffull);
                        deferred = true;
                        key = DONE;
                        sel = -1;
                    }
                    continue;
                }
                const NodeRec* rp =
                    which < 0 ? ebuf + ((long)item * A.geom.ni_main() + cslot) * GW
                              : ebuf + ((long)item * A.geom.ni_sub(which + 1) + (cslot - A.geom.base[which + 1])) * GW;
                const NodeRec rec = rp[lane];
                double2* s = slots + (q * GW + lane) * 3;
                s[0] = rec.A0, s[1] = rec.Q1, s[2] = rec.Q0;
                if (sel == q) my_cslot = cslot;
            }
            // the slots are produced and consumed inside one wave: LDS operations of a wave
            // complete in issue order; the fences keep the compiler from moving them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- phase 2: lane = omega -----------------------------------------------------------
            if (sel >= 0) {
                const int cslot = my_cslot;
                const int depth = (int)(key & 63ull);
                const unsigned long long path = (key >> 6) >> (KD - depth);
                const double2* myslots = slots + sel * (GW * 3);
                const double2* tp = A.etab + (long)cslot * GW * A.n_act + wpos;
                auto node_of = [&](int s) { return s == 0 ? 0 : ((s & 1) ? (s + 1) >> 1 : (s >> 1) + H - 1); };
                double2 ev[PTS];
#pragma unroll
                for (int s = 0; s < PTS; ++s) ev[s] = tp[(long)node_of(s) * A.n_act];
                cd K = mk(0.0, 0.0), G = mk(0.0, 0.0), fplus = mk(0.0, 0.0);
#pragma unroll
                for (int s = 0; s < PTS; ++s) {
                    const double2* r = myslots + node_of(s) * 3;
                    const double2 a0 = r[0], q1 = r[1], q0 = r[2];
                    const double2 tt = ev[s];
                    // safe_exp clamp (src/Parameters.cpp:167-173) as |exp(A0 + T omega)|^2 < exp(-80)
                    const bool clamped = fma(tt.x, tt.x, tt.y * tt.y) * a0.x < 1.8048513878454153e-35;
                    const cd S = mk(fma(omega.x, q1.x, fma(-omega.y, q1.y, q0.x)),
                                    fma(omega.x, q1.y, fma(omega.y, q1.x, q0.y)));
                    const cd ef = mk(tt.x, tt.y) * S;
                    const cd f = clamped ? mk(0.0, 0.0) : ef;
                    if (s & 1) {
                        fplus = f;
                    } else {
                        const int q = s >> 1;
                        const cd fs = s == 0 ? f : fplus + f;
                        K = K + WK[q] * fs;
                        if ((q & 1) == 0) G = G + WG[q >> 1] * fs;
                    }
                }
                ++count;
                // include/functions.h:203-208, 231-247
                const double scale = A.scale[cslot];  // (r - l)/2 of the interval: small table, L2-resident
                const double dKx = K.x - G.x, dKy = K.y - G.y;
                const double absK = sqrt(fma(K.x, K.x, K.y * K.y));
                double err = fmax(sqrt(fma(dKx, dKx, dKy * dKy)), absK * (2.0 * 2.220446049250313e-16));
                const cd integral = mk(K.x * scale, K.y * scale);
                err *= scale;
                const double rel_abs = P.rel_tol * (absK * scale);
                if (abs_tol == 0.0) abs_tol = rel_abs;
                bool split = depth < P.max_sub && err > abs_tol * inv_scale + P.prec_goal &&
                             err > rel_abs + P.prec_goal;
                if (split && (depth >= EMME_MAX_DEPTH || count >= EMME_MAX_INTERVALS)) {
                    split = false;
                    bad = 1;
                }
                if (split) {
                    key = make_key(depth + 1, path << 1);
                } else {
                    sum = sum + integral;
                    unsigned long long p2 = path + 1;
                    const int tz = min(depth, (int)__builtin_ctzll(p2 | (1ull << 63)));
                    p2 >>= tz;
                    const int d2 = depth - tz;
                    key = d2 == 0 ? DONE : make_key(d2, p2);
                }
            }
            // phase-2 reads are done before the next round overwrites the slots
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (has_w && !deferred) {
            my_intervals += (unsigned long long)count;
            const double dg = gtab[i] - gtab[j], de = eta[i] - eta[j];
            cd kap = mk(P.pref * sum.y, -(P.pref * sum.x));  // -i pref sum, Parameters.cpp:182
            if (kappa_bad(kap)) bad = 1;
            kap = kap + kappa_e(0, P, de, dg, omega);
            const cd v = (-(pair_weight(i, j, N) * P.dx)) * kap;
            store(i, j, v);
            store(j, i, v);
        }
    }
    block_add_intervals(s_iv, A.intervals, lane, has_w, has_w && group_in_block == 0, b, my_intervals);
    if (has_w && bad) A.status[b] = 1;
}

}  // namespace

// part -1 = main buffer (full tree + subtree 0), part k >= 0 = run-time subtree k+1
size_t node_cache_bytes(int gk_points, long nitems, const NodeCacheGeom& g, int part) {
    const int gw = gk_points == 15 ? 16 : 32;
    const CacheGeom c = make_geom(g);
    const int ni = part < 0 ? c.ni_main() : c.ni_sub(part + 1);
    return (size_t)nitems * (size_t)ni * gw * sizeof(NodeRec);
}
int node_cache_intervals(const NodeCacheGeom& g) { return make_geom(g).ni(); }
size_t node_ttab_bytes(int gk_points, int max_intervals) {
    return (size_t)max_intervals * (gk_points == 15 ? 16 : 32) * sizeof(double2);
}

hipError_t launch_node_cache(const AssembleLaunch& L, const NodeCacheGeom& g, int part, double omi,
                             void* recs, void* ttab, void* wtab, double* scale, bool folded,
                             hipStream_t stream) {
    CacheArgs A;
    A.P = L.P;
    A.tab = L.tab;
    A.pairs = (const ushort2*)L.pairs;
    A.npairs = L.npairs;
    A.geom = make_geom(g);
    A.omi = omi;
    A.recs = (NodeRec*)recs;
    A.ttab = (double2*)ttab;
    A.wtab = (double2*)wtab;
    A.scale = scale;
    A.folded = folded ? 1 : 0;
    A.part = part;
    A.first = part < 0 ? 0 : A.geom.base[part + 1];
    A.count = part < 0 ? A.geom.ni_main() : A.geom.ni_sub(part + 1);
    if (A.count == 0) return hipSuccess;
    dim3 grid(256 * 32), block(256);
    if (L.gk_points == 15)
        hipLaunchKernelGGL(k_node_cache<15>, grid, block, 0, stream, A);
    else
        hipLaunchKernelGGL(k_node_cache<31>, grid, block, 0, stream, A);
    return hipGetLastError();
}

hipError_t launch_assemble_cached(const AssembleLaunch& L, const NodeCacheGeom& g,
                                  const void* const recs[2],
                                  const void* const recs_ext[2][NODE_CACHE_MAX_SUB - 1],
                                  const void* const ttab[2], const double* scale, const void* etab,
                                  unsigned long long* worklist, unsigned int* worklist_count,
                                  unsigned long long* defer_info, const int* act_idx, int n_act,
                                  const void* chunks, int nchunks, hipStream_t stream) {
    AsmCachedArgs A;
    A.chunks = (const int2*)chunks;
    A.P = L.P;
    A.tab = L.tab;
    A.pairs = (const ushort2*)L.pairs;
    A.npairs = L.npairs;
    A.geom = make_geom(g);
    for (int c = 0; c < 2; ++c) {
        A.recs[c] = (const NodeRec*)recs[c];
        for (int k = 0; k < NODE_CACHE_MAX_SUB - 1; ++k) A.recs_ext[c][k] = (const NodeRec*)recs_ext[c][k];
        A.ttab[c] = (const double2*)ttab[c];
        A.wtab[c] = nullptr;
    }
    A.scale = scale;
    A.etab = (const double2*)etab;
    A.worklist = worklist;
    A.worklist_count = worklist_count;
    A.defer_info = defer_info;
    A.act_idx = act_idx;
    A.n_act = n_act;
    A.omega = (const double2*)L.omega;
    A.M = (double2*)L.M;
    A.Mold = (const double2*)L.Mold;
    A.Mp = (double2*)L.Mp;
    A.domega = (const double2*)L.domega;
    A.intervals = L.intervals;
    A.status = L.status;
    A.skip_lost = L.skip_lost;
    const int gw = L.gk_points == 15 ? 16 : 32;
    const int groups_per_block = 256 / gw;
    const long nitems = (long)L.npairs * L.P.nm;
    long want_groups = (nitems + L.items_per_group - 1) / L.items_per_group;
    long gx = (want_groups + groups_per_block - 1) / groups_per_block;
    if (gx < 1) gx = 1;
    if (gx > 65535) gx = 65535;
    dim3 grid((unsigned)gx, (unsigned)nchunks), block(256);
    const size_t lds = ((size_t)3 * L.P.N + (size_t)A.geom.ni()) * sizeof(double);
    // electrostatic GK15 on folded records: the union-walk kernel (EMME_UNION=0: independent lanes)
    const bool union_walk = L.union_walk != 0;
    if (L.gk_points == 15 && etab && union_walk && L.P.nm == 1) {
        const long ug = (nitems + L.items_per_group - 1) / L.items_per_group;
        long ugx = (ug + 15) / 16;
        if (ugx < 1) ugx = 1;
        if (ugx > 65535) ugx = 65535;
        const size_t n0 = (size_t)3 * L.P.N;
        // intervals served per round (EMME_UNION_SEL, default 2): see the kernel
        const int nsel = L.union_sel;
        const size_t ulds0 = (n0 + (n0 & 1)) * sizeof(double);
        const size_t slot_bytes = (size_t)16 * 16 * 3 * sizeof(double2);
        if (nsel <= 1)
            hipLaunchKernelGGL((k_assemble_union<15, 1>), dim3((unsigned)ugx, (unsigned)nchunks), block, ulds0 + slot_bytes, stream, A);
        else if (nsel == 2)
            hipLaunchKernelGGL((k_assemble_union<15, 2>), dim3((unsigned)ugx, (unsigned)nchunks), block, ulds0 + 2 * slot_bytes, stream, A);
        else
            hipLaunchKernelGGL((k_assemble_union<15, 4>), dim3((unsigned)ugx, (unsigned)nchunks), block, ulds0 + 4 * slot_bytes, stream, A);
    } else if (L.gk_points == 15 && etab)
        hipLaunchKernelGGL((k_assemble_cached<15, true>), grid, block, lds, stream, A);
    else if (L.gk_points == 15)
        hipLaunchKernelGGL((k_assemble_cached<15, false>), grid, block, lds, stream, A);
    else if (etab)
        hipLaunchKernelGGL((k_assemble_cached<31, true>), grid, block, lds, stream, A);
    else
        hipLaunchKernelGGL((k_assemble_cached<31, false>), grid, block, lds, stream, A);
    return hipGetLastError();
}

hipError_t launch_assemble_cached_em(const AssembleLaunch& L, const NodeCacheGeom& g,
                                  const void* const recs[2],
                                  const void* const recs_ext[2][NODE_CACHE_MAX_SUB - 1],
                                  const void* const ttab[2], const void* const wtab[2], const double* scale,
                                  const void* etab, unsigned long long* worklist, unsigned int* worklist_count,
                                  unsigned long long* defer_info, const int* act_idx, int n_act,
                                  const void* chunks, int nchunks, hipStream_t stream) {
    AsmCachedArgs A;
    A.chunks = (const int2*)chunks;
    A.P = L.P;
    A.tab = L.tab;
    A.pairs = (const ushort2*)L.pairs;
    A.npairs = L.npairs;
    A.geom = make_geom(g);
    for (int c = 0; c < 2; ++c) {
        A.recs[c] = (const NodeRec*)recs[c];
        for (int k = 0; k < NODE_CACHE_MAX_SUB - 1; ++k) A.recs_ext[c][k] = (const NodeRec*)recs_ext[c][k];
        A.ttab[c] = (const double2*)ttab[c];
        A.wtab[c] = (const double2*)wtab[c];
    }
    A.scale = scale;
    A.etab = (const double2*)etab;
    A.worklist = worklist;
    A.worklist_count = worklist_count;
    A.defer_info = defer_info;
    A.act_idx = act_idx;
    A.n_act = n_act;
    A.omega = (const double2*)L.omega;
    A.M = (double2*)L.M;
    A.Mold = (const double2*)L.Mold;
    A.Mp = (double2*)L.Mp;
    A.domega = (const double2*)L.domega;
    A.intervals = L.intervals;
    A.status = L.status;
    A.skip_lost = L.skip_lost;
    const int gw = L.gk_points == 15 ? 16 : 32;
    const int groups_per_block = 256 / gw;
    const long nitems = (long)L.npairs;  // an item is a pair
    long want_groups = (nitems + L.items_per_group - 1) / L.items_per_group;
    long gx = (want_groups + groups_per_block - 1) / groups_per_block;
    if (gx < 1) gx = 1;
    if (gx > 65535) gx = 65535;
    dim3 grid((unsigned)gx, (unsigned)nchunks), block(256);
    const size_t lds = ((size_t)3 * L.P.N + (size_t)A.geom.ni()) * sizeof(double);
    if (L.gk_points == 15 && etab)
        hipLaunchKernelGGL((k_assemble_cached_em<15, true>), grid, block, lds, stream, A);
    else if (L.gk_points == 15)
        hipLaunchKernelGGL((k_assemble_cached_em<15, false>), grid, block, lds, stream, A);
    else if (etab)
        hipLaunchKernelGGL((k_assemble_cached_em<31, true>), grid, block, lds, stream, A);
    else
        hipLaunchKernelGGL((k_assemble_cached_em<31, false>), grid, block, lds, stream, A);
    return hipGetLastError();
}

hipError_t launch_phase_table(int gk_points, int n_intervals, const void* const ttab[2],
                              const double* omega, const int* act_idx, int n_act, void* etab,
                              hipStream_t stream) {
    PhaseArgs A;
    A.ttab[0] = (const double2*)ttab[0], A.ttab[1] = (const double2*)ttab[1];
    A.omega = (const double2*)omega;
    A.act_idx = act_idx;
    A.n_act = n_act;
    A.nrows = n_intervals * (gk_points == 15 ? 16 : 32);
    A.etab = (double2*)etab;
    const long total = (long)A.nrows * n_act;
    long blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_phase_table, dim3((unsigned)blocks), dim3(256), 0, stream, A);
    return hipGetLastError();
}

}  // namespace emme
