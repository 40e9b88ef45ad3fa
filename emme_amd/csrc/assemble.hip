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

#include "emme_device.hpp"
#include "launch.hpp"

namespace emme {

namespace {

// per-lane node tables: lane r of a group -> (signed abscissa, Kronrod weight, Gauss weight)
__device__ const double kX15[8] = {0.,
                                   0.20778495500789847,
                                   0.40584515137739717,
                                   0.58608723546769113,
                                   0.74153118559939444,
                                   0.86486442335976907,
                                   0.94910791234275852,
                                   0.99145537112081264};
__device__ const double kWg15[4] = {0.41795918367346939, 0.38183005050511894,
                                    0.27970539148927667, 0.12948496616886969};
__device__ const double kWk15[8] = {2.09482141084727828e-01, 2.04432940075298892e-01,
                                    1.90350578064785410e-01, 1.69004726639267903e-01,
                                    1.40653259715525919e-01, 1.04790010322250184e-01,
                                    6.30920926299785533e-02, 2.29353220105292250e-02};
__device__ const double kX31[16] = {0.0,
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
__device__ const double kWg31[8] = {0.20257824192556112, 0.19843148532711152,
                                    0.18616100001556193, 0.1662692058169939,
                                    0.1395706779261542,  0.10715922046717143,
                                    0.07036604748810768, 0.030753241996119};
__device__ const double kWk31[16] = {
    0.10133000701479155,   0.100769845523875595,  0.099173598721791959,  0.0966427269836236785,
    0.093126598170825321,  0.0885644430562117706, 0.083080502823133021,  0.0768496807577203789,
    0.069854121318728259,  0.0620095678006706403, 0.053481524690928087,  0.0445897513247648766,
    0.035346360791375846,  0.0254608473267153202, 0.0150079473293161225, 0.00537747987292334899};

template <int PTS>
__device__ __forceinline__ GkLane gk_lane(int r) {
    constexpr int H = (PTS + 1) / 2;  // 8 or 16 (centre + H-1 pairs)
    const double* X = PTS == 15 ? kX15 : kX31;
    const double* WK = PTS == 15 ? kWk15 : kWk31;
    const double* WG = PTS == 15 ? kWg15 : kWg31;
    GkLane g;
    if (r >= PTS) {  // padding lane: evaluates the centre again with zero weight
        g.x = 0.0, g.wk = 0.0, g.wg = 0.0;
        return g;
    }
    const int i = r < H ? r : r - (H - 1);  // node index 0..H-1
    g.x = r < H ? X[i] : -X[i];
    g.wk = WK[i];
    // Gauss nodes of the embedded rule: the centre and the even Kronrod nodes
    // (include/functions.h:190-199; both embedded orders, 7 and 15, are odd)
    g.wg = (i % 2 == 0) ? WG[i / 2] : 0.0;
    return g;
}

template <int GW>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int off = 1; off < GW; off <<= 1) v += __shfl_xor(v, off, GW);
    return v;
}

// Adiabatic-electron closed forms kappa_e (src/Parameters.cpp:186-209).
__device__ __forceinline__ cd kappa_e(int m, const DevParams& P, double de, double dg, cd omega) {
    if (m == 1) {
        // -i qR/(2 vt tau) (omega - ws_e) sgn(de)
        const double c = P.qR / (2.0 * P.vt * P.tau) * (de / fabs(de));
        const cd a = mk(omega.x - P.omega_s_e, omega.y);
        return mk(c * a.y, -(c * a.x));
    }
    if (m == 2) {
        const double f = (P.qR * P.qR) / (2.0 * P.vt * P.vt * P.tau) * de / fabs(de);
        const cd wa = mk(omega.x - P.omega_s_e, omega.y);
        const cd a = de * (omega * wa);
        const double b1e = P.cbe * dg;
        const cd b = (b1e * P.vt / P.qR) * mk(omega.x - P.omega_s_e * (1.0 + P.eta_e), omega.y);
        return f * (a - b);
    }
    return mk(0.0, 0.0);
}

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
};

template <int PTS>
__global__ __launch_bounds__(256) void k_assemble(AsmArgs A) {
    constexpr int GW = PTS == 15 ? 16 : 32;
    constexpr int GROUPS_PER_BLOCK = 256 / GW;
    extern __shared__ double lds_tab[];  // eta | g | b  (3N doubles)

    const DevParams& P = A.P;
    const int b = blockIdx.y;
    if (A.active && A.active[b] == 0) return;
    const int N = P.N, dim = P.dim;

    for (int k = threadIdx.x; k < 3 * N; k += blockDim.x) lds_tab[k] = A.tab[k];
    __syncthreads();
    const double* eta = lds_tab;
    const double* gtab = lds_tab + N;
    const double* btab = lds_tab + 2 * N;

    double2* Mb = A.M + (size_t)b * dim * dim;
    const double2* Moldb = A.Mold ? A.Mold + (size_t)b * dim * dim : nullptr;
    double2* Mpb = A.Mp ? A.Mp + (size_t)b * dim * dim : nullptr;
    cd rdw = mk(0.0, 0.0);
    if (Moldb) rdw = rcp(mk(A.domega[b].x, A.domega[b].y));

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
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            store(i, i, mk(P.diag_a, 0.0));
            if (P.nm == 3) {
                store(i, i + N, mk(0.0, 0.0));
                store(i + N, i, mk(0.0, 0.0));
                store(i + N, i + N, mk(P.diag_d * btab[i], 0.0));
            }
        }
    }

    OmegaConst oc;
    oc.omega = mk(A.omega[b].x, A.omega[b].y);
    oc.omi = -copysign(1.0, oc.omega.x);

    const int lane_in_group = threadIdx.x % GW;
    const int group = blockIdx.x * GROUPS_PER_BLOCK + threadIdx.x / GW;
    const int ngroups = gridDim.x * GROUPS_PER_BLOCK;
    const GkLane gk = gk_lane<PTS>(lane_in_group);

    const double qa = 0.0, qb = M_PI / 2.0;  // include/functions.h:319 / :328
    const double inv_scale = 2. / (qb - qa);
    const int nitems = A.npairs * P.nm;

    // ---- group state ------------------------------------------------------------
    int item = group;
    bool live = item < nitems;
    int i = 0, j = 0, m = 0;
    PairConst pc{};
    double dg = 0.0;
    int depth = 0;
    unsigned long long path = 0;  // index of the current interval at this depth
    double abs_tol = 0.0;
    cd sum = mk(0.0, 0.0);
    unsigned long long my_intervals = 0;
    int item_intervals = 0;  // safety valve: no integral may run away (every wave must exit)
    int bad = 0;

    auto load_item = [&]() {
        const int p = item / P.nm;
        m = item - p * P.nm;
        const ushort2 ij = A.pairs[p];
        i = ij.x, j = ij.y;
        const double bi = btab[i], bj = btab[j];
        dg = gtab[i] - gtab[j];
        pc.de = eta[i] - eta[j];
        pc.beta1 = P.cb * dg;
        pc.s = sqrt(bi * bj);
        pc.inv_s = 1.0 / pc.s;
        pc.bsum = bi + bj;
        const double qRd = P.qR * pc.de;
        pc.c_lam = 0.5 * P.vt / qRd * pc.beta1;
        pc.c_nv = qRd / P.vt;
        depth = 0, path = 0, abs_tol = 0.0;
        item_intervals = 0;
        sum = mk(0.0, 0.0);
    };
    if (live) load_item();

    while (live) {
        // [l, r] of node (depth, path): the reference's (l+r)/2 bisection sequence
        double l = qa, r = qb;
        for (int s = depth - 1; s >= 0; --s) {
            const double mid_s = (r + l) / 2;
            if ((path >> s) & 1)
                l = mid_s;
            else
                r = mid_s;
        }
        const double mid = (r + l) / 2;
        const double scale = (r - l) / 2;
        // abscissa scale * x + mid, rounded like the reference (no FMA contraction)
        const double x = __dadd_rn(__dmul_rn(scale, gk.x), mid);

        const cd f = integrand(x, P, pc, oc, m);
        const double Kx = group_sum<GW>(gk.wk * f.x), Ky = group_sum<GW>(gk.wk * f.y);
        const double Gx = group_sum<GW>(gk.wg * f.x), Gy = group_sum<GW>(gk.wg * f.y);
        ++my_intervals;
        ++item_intervals;

        // include/functions.h:203-208, 231-233
        double err = fmax(hypot(Kx - Gx, Ky - Gy), hypot(Kx, Ky) * (2.0 * 2.220446049250313e-16));
        const cd integral = mk(Kx * scale, Ky * scale);
        err *= scale;
        const double rel_abs = hypot(P.rel_tol * integral.x, P.rel_tol * integral.y);
        if (abs_tol == 0.0) abs_tol = rel_abs;  // :237-239
        // :240-242; ldexp(scale, max_sub) > 0.99 (b - a)
        bool split = ldexp(scale, P.max_sub) > 0.99 * (qb - qa) &&
                     err > abs_tol * inv_scale + P.prec_goal && err > rel_abs + P.prec_goal;
        if (split && (depth >= 62 || item_intervals >= (1 << 18))) {  // flag and accept
            split = false;
            bad = 1;
        }
        if (split) {
            ++depth;
            path <<= 1;  // left half first (the reference pushes [mid,r] then [l,mid])
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
                if (!(isfinite(kap.x) && isfinite(kap.y))) bad = 1;
                kap = kap + kappa_e(m, P, pc.de, dg, oc.omega);
                if (lane_in_group == 0) {
                    if (m == 0) {
                        // A_ij = -kappa_all(0) W_ij dx (include/solver.h:448-453)
                        double w = (j - i) <= 5
                                       ? (j - i == 1   ? 2.951388888888883
                                          : j - i == 2 ? -2.4305555555555305
                                          : j - i == 3 ? 4.166666666667441
                                          : j - i == 4 ? -0.3472222222224549
                                                       : 1.159722222222284)
                                       : 1.0;
                        if (j == N - 1) w -= 0.5;  // src/singularity_handler.cpp:18 (j>i>=0)
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
                item += ngroups;
                live = item < nitems;
                if (live) load_item();
            }
        }
    }

    if (lane_in_group == 0) {
        if (A.intervals && my_intervals) atomicAdd(&A.intervals[b], my_intervals);
        if (bad) A.status[b] = 1;
    }
}

}  // namespace

hipError_t launch_assemble(const AssembleLaunch& L, hipStream_t stream) {
    AsmArgs A;
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
    const size_t lds = (size_t)3 * L.P.N * sizeof(double);
    if (L.gk_points == 15)
        hipLaunchKernelGGL(k_assemble<15>, grid, block, lds, stream, A);
    else
        hipLaunchKernelGGL(k_assemble<31>, grid, block, lds, stream, A);
    return hipGetLastError();
}

}  // namespace emme
