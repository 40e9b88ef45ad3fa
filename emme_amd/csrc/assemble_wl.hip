// assemble_wl.hip -- batched fill, "omega-lane" form: the kernel of choice when several
// omega candidates are assembled for the same parameter set (the root-search batch).
//
// Observation (emme_device.hpp::node_data): for a fixed pair (i,j), moment m and contour
// sense, the integrand at a quadrature node is  exp(A0 + T w)(w Q1 + Q0)  where A0, T, Q1, Q0
// do not depend on omega -- and they contain everything expensive (the Miller recurrence for
// I0/I1, sincos of the abscissa, two rsqrt).  The reference recomputes all of it for every
// omega (src/Parameters.cpp:113-184 is called once per matrix entry per omega).
//
// Mapping.  A lane group of GW = 16 (GK15) / 32 (GK31) lanes owns one (pair, moment) item and
// GW omegas of the batch at a time, and alternates two phases per quadrature interval:
//   phase 1  lane = node : lane r evaluates the omega-independent NodeData of node r and
//            parks it in LDS (64 B per node);
//   phase 2  lane = omega: every lane walks the 15 (31) nodes in the reference's summation
//            order, finishing F with its own omega (one complex exp + three complex products
//            per node), and accumulates its private Kronrod / Gauss sums -- no cross-lane
//            reduction, and the accept/split decision is per lane.
// Each omega keeps its own adaptive tree (same decisions as the reference); the group walks
// the UNION of the trees depth-first: an interval is evaluated when at least one lane's next
// node is that interval, and lanes whose tree does not contain it sit the round out.
#include <hip/hip_runtime.h>

#include "assemble_common.hpp"
#include "launch.hpp"

namespace emme {

namespace {

struct AsmWlArgs {
    DevParams P;
    const double* tab;     // eta[N] | g[N] | b[N]
    const ushort2* pairs;  // (i, j), i < j, ordered by j - i
    int npairs;
    const int* act_idx;    // [n_act] batch index of every omega handled by this launch
    int n_act;
    const double2* omega;  // [nbatch]
    double2* M;            // [nbatch][dim][dim]
    const double2* Mold;   // null, or [nbatch][dim][dim] -> also write Mp = (M - Mold)/domega
    double2* Mp;
    const double2* domega;
    unsigned long long* intervals;  // [nbatch]
    int* status;                    // [nbatch]
    unsigned long long* rounds;     // [1] interval rounds walked by all groups (diagnostic)
};

#ifndef EMME_WL_MIN_WAVES
#define EMME_WL_MIN_WAVES 3
#endif

template <int PTS>
__global__ __launch_bounds__(256, EMME_WL_MIN_WAVES) void k_assemble_wl(AsmWlArgs A) {
    constexpr int GW = PTS == 15 ? 16 : 32;
    constexpr int H = (PTS + 1) / 2;
    constexpr int GROUPS_PER_BLOCK = 256 / GW;
    constexpr int GROUPS_PER_WAVE = 64 / GW;
    constexpr int MAXD = EMME_MAX_DEPTH;
    extern __shared__ double lds_raw[];  // tables | interval stacks | node slots
    __shared__ unsigned long long s_iv[GW];  // interval counts of the chunk's omegas (block_add_intervals)
    if (threadIdx.x < GW) s_iv[threadIdx.x] = 0ull;

    const DevParams& P = A.P;
    const TransConsts TC = trans_consts();
    const int N = P.N, dim = P.dim;
    for (int k = threadIdx.x; k < 3 * N; k += blockDim.x) lds_raw[k] = A.tab[k];
    __syncthreads();
    const double* eta = lds_raw;
    const double* gtab = lds_raw + N;
    const double* btab = lds_raw + 2 * N;
    const int group_in_block = threadIdx.x / GW;
    const int lane = threadIdx.x % GW;
    double2* stk = reinterpret_cast<double2*>(lds_raw + 3 * N + (3 * N & 1)) + group_in_block * MAXD;
    double2* slots = reinterpret_cast<double2*>(lds_raw + 3 * N + (3 * N & 1)) +
                     GROUPS_PER_BLOCK * MAXD + group_in_block * (GW * 4);  // 4 double2 per node

    // this lane's omega (phase 2 identity)
    const int slot_w = blockIdx.y * GW + lane;
    const bool has_w = slot_w < A.n_act;
    const int b = has_w ? A.act_idx[slot_w] : 0;
    cd omega = mk(0.0, 0.0);
    cd rdw = mk(0.0, 0.0);
    if (has_w) {
        omega = mk(A.omega[b].x, A.omega[b].y);
        if (A.Mold) rdw = rcp(mk(A.domega[b].x, A.domega[b].y));
    }
    const double my_omi = -copysign(1.0, omega.x);
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

    // diagonal (include/solver.h:442-443, 465-470): first block of every omega chunk
    if (blockIdx.x == 0 && has_w) {
        for (int i = group_in_block; i < N; i += GROUPS_PER_BLOCK) {
            store(i, i, mk(P.diag_a, 0.0));
            if (P.nm == 3) {
                store(i, i + N, mk(0.0, 0.0));
                store(i + N, i, mk(0.0, 0.0));
                store(i + N, i + N, mk(P.diag_d * btab[i], 0.0));
            }
        }
    }

    // wave-level helper: does any lane of MY group satisfy pred?
    const unsigned long long gmask =
        (GW == 64 ? ~0ull : ((1ull << GW) - 1ull)) << (((threadIdx.x & 63) / GW) * GW);
    auto group_any = [&](bool pred) -> bool { return (__ballot(pred) & gmask) != 0ull; };
    (void)GROUPS_PER_WAVE;

    const GkLane gk = gk_lane<PTS>(lane);  // phase-1 identity: node `lane`
    const double* WK = PTS == 15 ? kWk15 : kWk31;
    const double* WG = PTS == 15 ? kWg15 : kWg31;

    const double qa = 0.0, qb = M_PI / 2.0;
    const double inv_scale = 2. / (qb - qa);
    const int nitems = A.npairs * P.nm;
    const int group = blockIdx.x * GROUPS_PER_BLOCK + group_in_block;
    const int ngroups = gridDim.x * GROUPS_PER_BLOCK;

    // ---- group state (replicated in every lane of the group) ------------------------
    int item = group;
    bool live = item < nitems;
    int i = 0, j = 0, m = 0;
    PairConst pc{};
    double dg = 0.0;
    int depth = 0;
    unsigned long long path = 0;
    double l = qa, r = qb;
    // ---- lane state (phase 2: this lane's omega) -------------------------------------
    int my_depth = 0;               // next node of this omega's own tree
    unsigned long long my_path = 0;
    bool my_done = true;
    double abs_tol = 0.0;
    cd sum = mk(0.0, 0.0);
    unsigned long long my_intervals = 0;
    int item_intervals = 0;
    int bad = 0;
    unsigned long long my_rounds = 0;

    auto load_item = [&]() {
        const int p = item / P.nm;
        m = item - p * P.nm;
        const ushort2 ij = A.pairs[p];
        i = ij.x, j = ij.y;
        dg = gtab[i] - gtab[j];
        pc = make_pair_const(P, eta[i], eta[j], btab[i], btab[j], dg);
        depth = 0, path = 0, l = qa, r = qb;
        my_depth = 0, my_path = 0, my_done = !has_w;
        abs_tol = 0.0, item_intervals = 0;
        sum = mk(0.0, 0.0);
    };
    if (live) load_item();

    while (live) {
        const double mid = (r + l) / 2;
        const double scale = (r - l) / 2;
        const bool need = !my_done && my_depth == depth && my_path == path;
        ++my_rounds;
        bool my_split = false;

#pragma unroll 1
        for (int cls = 0; cls < 2; ++cls) {
            const double omi = cls == 0 ? 1.0 : -1.0;
            const bool mine = need && my_omi == omi;
            if (!group_any(mine)) continue;

            // ---- phase 1: lane = node ------------------------------------------------
            {
                const double x = __dadd_rn(__dmul_rn(scale, gk.x), mid);
                const NodeData d = node_data(x, P, pc, omi, m);
                double2* s = slots + lane * 4;
                s[0] = make_double2(d.A0.x, d.A0.y);
                s[1] = make_double2(d.T.x, d.T.y);
                s[2] = make_double2(d.Q1.x, d.Q1.y);
                s[3] = make_double2(d.Q0.x, d.Q0.y);
            }
            // the slots are produced and consumed inside one wave: LDS operations of a wave
            // complete in issue order; the fence keeps the compiler from moving them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ---- phase 2: lane = omega -------------------------------------------------
            if (mine) {
                auto eval = [&](int node) -> cd {
                    const double2* s = slots + node * 4;
                    NodeData d;
                    d.A0 = mk(s[0].x, s[0].y);
                    d.T = mk(s[1].x, s[1].y);
                    d.Q1 = mk(s[2].x, s[2].y);
                    d.Q0 = mk(s[3].x, s[3].y);
                    return node_eval(d, omega, TC);
                };
                // include/functions.h:186-201: centre, then f(+x_i) + f(-x_i) for i = 1..
                const cd f0 = eval(0);
                cd K = WK[0] * f0;
                cd G = WG[0] * f0;
#pragma unroll 1
                for (int q = 1; q < H; ++q) {
                    const cd f = eval(q) + eval(q + H - 1);
                    K = K + WK[q] * f;
                    if ((q & 1) == 0) G = G + WG[q >> 1] * f;
                }
                ++my_intervals;
                ++item_intervals;
                // include/functions.h:203-208, 231-247
                const double dKx = K.x - G.x, dKy = K.y - G.y;
                const double absK = sqrt(fma(K.x, K.x, K.y * K.y));
                double err = fmax(sqrt(fma(dKx, dKx, dKy * dKy)), absK * (2.0 * 2.220446049250313e-16));
                const cd integral = mk(K.x * scale, K.y * scale);
                err *= scale;
                const double rel_abs = P.rel_tol * (absK * scale);
                if (abs_tol == 0.0) abs_tol = rel_abs;
                my_split = depth < P.max_sub && err > abs_tol * inv_scale + P.prec_goal &&
                           err > rel_abs + P.prec_goal;
                if (my_split && (depth >= MAXD || item_intervals >= EMME_MAX_INTERVALS)) {
                    my_split = false;
                    bad = 1;
                }
                if (!my_split) sum = sum + integral;
            }
            // phase-2 reads must finish before the next class / round overwrites the slots
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        // ---- successor of the current interval in pre-order (group-uniform) -------------
        unsigned long long spath = path + 1;
        int sdepth = depth;
        while (sdepth > 0 && !(spath & 1)) {
            spath >>= 1;
            --sdepth;
        }
        // ---- lane bookkeeping ------------------------------------------------------------
        if (need) {
            if (my_split) {
                my_depth = depth + 1;
                my_path = path << 1;
            } else if (sdepth == 0) {
                // this omega's integral is complete: kappa = -i pref sum, + kappa_e, scatter
                my_done = true;
                cd kap = mk(P.pref * sum.y, -(P.pref * sum.x));
                if (kappa_bad(kap)) bad = 1;
                kap = kap + kappa_e(m, P, pc.de, dg, omega);
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
            } else {
                my_depth = sdepth;
                my_path = spath;
            }
        }
        // ---- group walk over the union tree ----------------------------------------------
        if (group_any(my_split)) {
            stk[depth] = make_double2(mid, r);  // bounds of the right half, for the way back
            r = mid;
            ++depth;
            path <<= 1;
        } else if (sdepth == 0) {
            item += ngroups;
            live = item < nitems;
            if (live) load_item();
        } else {
            depth = sdepth;
            path = spath;
            const double2 pr = stk[depth - 1];
            l = pr.x;
            r = pr.y;
        }
    }

    if (A.rounds && lane == 0 && my_rounds) atomicAdd(A.rounds, my_rounds);
    block_add_intervals(s_iv, A.intervals, lane, has_w, has_w && group_in_block == 0, b, my_intervals);
    if (has_w && bad) A.status[b] = 1;
}

}  // namespace

hipError_t launch_assemble_wl(const AssembleLaunch& L, const int* act_idx, int n_act,
                              hipStream_t stream) {
    AsmWlArgs A;
    A.P = L.P;
    A.tab = L.tab;
    A.pairs = (const ushort2*)L.pairs;
    A.npairs = L.npairs;
    A.act_idx = act_idx;
    A.n_act = n_act;
    A.omega = (const double2*)L.omega;
    A.M = (double2*)L.M;
    A.Mold = (const double2*)L.Mold;
    A.Mp = (double2*)L.Mp;
    A.domega = (const double2*)L.domega;
    A.intervals = L.intervals;
    A.status = L.status;
    A.rounds = L.rounds;
    const int gw = L.gk_points == 15 ? 16 : 32;
    const int groups_per_block = 256 / gw;
    const int chunks = (n_act + gw - 1) / gw;
    const long nitems = (long)L.npairs * L.P.nm;
    long want_groups = (nitems + L.items_per_group - 1) / L.items_per_group;
    long gx = (want_groups + groups_per_block - 1) / groups_per_block;
    if (gx < 1) gx = 1;
    if (gx > 65535) gx = 65535;
    dim3 grid((unsigned)gx, (unsigned)chunks), block(256);
    const size_t lds = ((size_t)3 * L.P.N + (3 * L.P.N & 1)) * sizeof(double) +
                       (size_t)groups_per_block * EMME_MAX_DEPTH * sizeof(double2) +
                       (size_t)256 * 4 * sizeof(double2);
    if (L.gk_points == 15)
        hipLaunchKernelGGL(k_assemble_wl<15>, grid, block, lds, stream, A);
    else
        hipLaunchKernelGGL(k_assemble_wl<31>, grid, block, lds, stream, A);
    return hipGetLastError();
}

}  // namespace emme
