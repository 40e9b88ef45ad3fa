// assemble_common.hpp -- pieces shared by the two fill kernels (assemble.hip: lanes = nodes,
// one omega per lane group; assemble_wl.hip: lanes = omegas sharing the omega-independent
// node data).
#pragma once
#include <hip/hip_runtime.h>

#include "emme_device.hpp"

namespace emme {
namespace {

// Caps of the adaptive quadrature that the reference does not have.  Its only limit is the depth
// integration_iteration_limit (`ldexp(scale, max_sub) > 0.99 (b - a)`, include/functions.h:240),
// 100 in the shipped inputs; every fill kernel here additionally stops refining at depth
// EMME_MAX_DEPTH (the interval is then pi/2 * 2^-40 = 1.4e-12 wide: a finite integrand of this
// family is resolved long before -- deepest tree met in the tests and the bench: 21 -- and a NaN
// never splits, its comparisons being false) or after EMME_MAX_INTERVALS intervals of one
// integral, and flags the matrix (status -> EMME_ENUMERIC) instead of walking on.  The same two
// numbers in every kernel, so that the flag does not depend on which kernel served the integral.
constexpr int EMME_MAX_DEPTH = 40;
constexpr int EMME_MAX_INTERVALS = 1 << 18;
// Largest modulus of an integral the device path stands for.  The accept/split rule takes |K| and |K - G| as
// sqrt(x^2 + y^2) and the Newton step's pivot search compares |x|^2: both overflow beyond 1.3e154, so a matrix
// with a larger entry would be walked on corrupted error estimates and could not be factored anyway.  The
// reference (std::abs = hypot, LAPACK's scaled arithmetic) goes on to 1e308 -- where, on the headline lattice,
// its own intermediates overflow first (chain 80's iterate -0.0055-0.734i: entries up to 3e202, two of them
// inf; its zsysv then fails, include/solver.h:142-153).  Such a matrix is flagged (status -> EMME_ENUMERIC).
constexpr double EMME_MAX_ENTRY = 1e150;
__device__ __forceinline__ bool kappa_bad(cd k) { return !(fabs(k.x) < EMME_MAX_ENTRY && fabs(k.y) < EMME_MAX_ENTRY); }

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

// all-reduce over a lane group of 16 (one DPP row) or 32 (two rows) lanes
template <int GW>
__device__ __forceinline__ double group_sum(double v) {
    v = row16_sum(v);
    if (GW == 32) v += __shfl_xor(v, 16);
    return v;
}

// Interval counters (one per batch item) leave a workgroup once per omega slot: the lanes add theirs up in
// LDS, one designated lane per slot (`flusher`) carries the sum to memory after a barrier.  Every workgroup
// of a fill launch adds to the same <= 128 counters: at one global atomic per lane the dense fill spent 10 %
// of a launch waiting for them.  s_iv: LDS, one word per slot, zeroed before the kernel's
// first barrier.
__device__ __forceinline__ void block_add_intervals(unsigned long long* s_iv, unsigned long long* intervals,
                                                    int slot, bool has_w, bool flusher, int b,
                                                    unsigned long long mine) {
    if (!intervals) return;  // (uniform)
    if (has_w && mine) atomicAdd(&s_iv[slot], mine);
    __syncthreads();
    if (flusher) {
        const unsigned long long v = s_iv[slot];
        if (v) atomicAdd(&intervals[b], v);
    }
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


// SingularityHandler weight for i < j (src/singularity_handler.cpp:4-20): end-corrected
// band near the diagonal, 1 elsewhere, minus one half on the last column.
__device__ __forceinline__ double pair_weight(int i, int j, int N) {
    const int d = j - i;
    double w = d <= 5 ? (d == 1   ? 2.951388888888883
                         : d == 2 ? -2.4305555555555305
                         : d == 3 ? 4.166666666667441
                         : d == 4 ? -0.3472222222224549
                                  : 1.159722222222284)
                      : 1.0;
    if (j == N - 1) w -= 0.5;
    return w;
}

__device__ __forceinline__ PairConst make_pair_const(const DevParams& P, double eta_i, double eta_j,
                                                     double bi, double bj, double dg) {
    PairConst pc;
    pc.de = eta_i - eta_j;
    pc.beta1 = P.cb * dg;
    pc.s = sqrt(bi * bj);
    pc.inv_s = 1.0 / pc.s;
    pc.bsum = bi + bj;
    const double qRd = P.qR * pc.de;
    pc.c_lam = 0.5 * P.vt / qRd * pc.beta1;
    pc.c_nv = qRd / P.vt;
    return pc;
}

}  // namespace
}  // namespace emme
