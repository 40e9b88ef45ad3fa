// assemble_dense.hip -- the electrostatic GK15 fill as small complex GEMMs on the FP64 matrix cores.
//
// With the node records folded (emme_device.hpp::node_data, assemble_cached.hip) the Kronrod and
// Gauss sums of one quadrature interval are, for pair p and omega w,
//     K[p, w] = sum_n  wk_n E[n, w] (w Q1[p, n] + Q0[p, n])  =  sum_k Q[p, k] BK[k, w]
//     G[p, w] =                                                  sum_k Q[p, k] BG[k, w]   (k < 16)
// with Q[p, 2n] = exp(A0) Q1, Q[p, 2n+1] = exp(A0) Q0 (omega-independent, HBM node cache) and
// BK[2n, w] = wk_n w E[n, w], BK[2n+1, w] = wk_n E[n, w], E = exp(T_n w) (pair-independent, one
// table per launch): a complex 16 x 16 x 32 GEMM per (16 pairs, 16 omegas, interval).  One wave owns
// a TILE of 16 consecutive pairs x one chunk of 16 cost-sorted omegas = 256 integrals
// (include/solver.h:446-455 fills them one task each) and walks the UNION of their adaptive trees in
// pre-order, one interval per round:
//   dense round  (the interval is in the trees of >= 3 omega columns): 48 v_mfma_f64_16x16x4_f64 --
//                 32 for K (k = 32), 16 for G (k = 16) -- fed by 40 coalesced 512-byte loads, no
//                 per-element address or key arithmetic at all;
//   sparse round (1-2 columns: a chain that has wandered to a damped omega refines where nobody
//                 else does): lane = node, DPP row sums, 16 pairs of one omega at a time;
// then every one of the 256 elements that owns the interval takes ITS OWN accept/split decision
// with its own abs_tol (include/functions.h:231-247), so the set of intervals of every integral --
// and its interval count -- is the reference's.  What changes is rounding-level only: the order of
// the node sums, and the safe_exp clamp (src/Parameters.cpp:167-173), which the GEMM cannot apply
// per (pair, node, omega): clamped terms are < e^-40 of their coefficient and enter at <= 4e-14
// absolute (records that are not finite are zeroed when the cache is built).  For that reason the
// dense path serves only inputs whose absolute quadrature goal (integration_accuracy) is >= 1e-9;
// tighter ones keep the exact union kernel.
//
// Intervals outside the cache defer the integrals that need them to k_assemble_coop, as before.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "assemble_common.hpp"
#include "launch.hpp"
#include "node_cache.hpp"

namespace emme {

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void interval_bounds_d(int depth, unsigned long long path, double& l, double& r) {
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

struct TiledCacheArgs {
    DevParams P;
    const double* tab;
    const ushort2* pairs;
    int npairs;
    CacheGeom geom;
    double omi;
    double* recs;    // [ntiles][count][TILE_BLOCK]
    double2* ttab;   // [NI][16]  T per (interval, node lane)
    double* scale;   // [NI]
    int part, first, count;
};

// One 16-lane group per (pair, interval), lane = node (gk_lane<15>): computes the folded record and
// scatters it into the tile block.
__global__ __launch_bounds__(256) void k_node_cache_tiled(TiledCacheArgs A) {
    constexpr int GW = 16, GROUPS_PER_BLOCK = 256 / GW;
    const DevParams& P = A.P;
    const int N = P.N, NI = A.count;
    const int lane = threadIdx.x % GW;
    const int ntiles = (A.npairs + TILE_PAIRS - 1) / TILE_PAIRS;
    const long total = (long)ntiles * TILE_PAIRS * NI;
    const GkLane gk = gk_lane<15>(lane);
    const double* eta = A.tab;
    const double* gtab = A.tab + N;
    const double* btab = A.tab + 2 * N;
    const int sn = slotnode_of_lane(lane);
    for (long w = (long)blockIdx.x * GROUPS_PER_BLOCK + threadIdx.x / GW; w < total;
         w += (long)gridDim.x * GROUPS_PER_BLOCK) {
        // w = (tile * NI + idx) * 16 + p: the 16 pairs of a (tile, interval) block are neighbours
        const int p = (int)(w % TILE_PAIRS);
        const long ti = w / TILE_PAIRS;
        const int idx = (int)(ti % NI);
        const long tile = ti / NI;
        const long item = tile * TILE_PAIRS + p;
        double* blk = A.recs + ti * TILE_BLOCK;
        cd q1 = mk(0.0, 0.0), q0 = mk(0.0, 0.0);
        double rea0 = -1.0e300;  // "always clamped"
        int depth;
        unsigned long long path;
        A.geom.interval(A.part, idx, depth, path);
        double l, r;
        interval_bounds_d(depth, path, l, r);
        const double mid = (r + l) / 2, scale = (r - l) / 2;
        const double x = __dadd_rn(__dmul_rn(scale, gk.x), mid);
        if (item < A.npairs || item == 0) {
            const ushort2 ij = A.pairs[item < A.npairs ? item : 0];
            const int i = ij.x, j = ij.y;
            const PairConst pc = make_pair_const(P, eta[i], eta[j], btab[i], btab[j], gtab[i] - gtab[j]);
            const NodeData d = node_data(x, P, pc, A.omi, 0);
            if (item < A.npairs && lane < 15) {
                double sa, ca;
                sincos(d.A0.y, &sa, &ca);
                const double ea = exp(fmin(d.A0.x, 700.0));
                const cd ex = mk(ea * ca, ea * sa);
                q1 = ex * d.Q1, q0 = ex * d.Q0;
                rea0 = d.A0.x;
                if (!(isfinite(q1.x) && isfinite(q1.y) && isfinite(q0.x) && isfinite(q0.y))) {
                    // exp(A0) = 0 against an overflowing amplitude: the reference's clamp makes this
                    // node contribute exactly 0 for every omega the integrand is finite for
                    q1 = mk(0.0, 0.0), q0 = mk(0.0, 0.0), rea0 = -1.0e300;
                }
            }
            if (item == 0) {
                A.ttab[(long)(A.first + idx) * GW + lane] = make_double2(d.T.x, d.T.y);
                if (lane == 0) A.scale[A.first + idx] = scale;
            }
        }
        blk[(2 * sn) * 16 + p] = q1.x;
        blk[512 + (2 * sn) * 16 + p] = q1.y;
        blk[(2 * sn + 1) * 16 + p] = q0.x;
        blk[512 + (2 * sn + 1) * 16 + p] = q0.y;
        blk[1024 + lane * 16 + p] = rea0;
    }
}

// Weighted phase tables of one launch (see node_cache.hpp: BTAB_BLOCK).
struct BtabArgs {
    const double2* ttab[2];
    const double2* omega;
    const int* act_idx;
    int n_act, nchunks, nslots;
    double* btab;
};
__global__ __launch_bounds__(256) void k_btab(BtabArgs A) {
    const long total = (long)A.nslots * 16 * A.n_act;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        // e = (slot * 16 + lane) * n_act + wpos: neighbouring threads = neighbouring omega columns of one
        // table row, so every store instruction writes whole 128-byte row segments
        const long row = e / A.n_act;
        const int wpos = (int)(e - row * A.n_act);
        const int lane = (int)(row & 15);
        const int slot = (int)(row >> 4);
        const double2 om = A.omega[A.act_idx[wpos]];
        const int cls = -copysign(1.0, om.x) > 0.0 ? 0 : 1;
        cd ev = mk(0.0, 0.0);
        if (A.ttab[cls] && lane < 15) {
            const double2 t = A.ttab[cls][(long)slot * 16 + lane];
            const double ax = fma(t.x, om.x, -(t.y * om.y)), ay = fma(t.x, om.y, t.y * om.x);
            if (!(ax > 700.0)) {  // beyond: not representable (never met where the integrand lives);
                double sa, ca;    // a NaN omega goes through and poisons its own column only
                sincos(ay, &sa, &ca);
                const double ea = exp(ax);
                ev = mk(ea * ca, ea * sa);
            }
        }
        const GkLane gk = gk_lane<15>(lane);
        const int sn = slotnode_of_lane(lane);
        const cd we = mk(om.x, om.y) * ev;
        double* blk = A.btab + ((size_t)slot * A.nchunks + (wpos >> 4)) * BTAB_BLOCK;
        const int col = wpos & 15;
        blk[(2 * sn) * 16 + col] = gk.wk * we.x;
        blk[512 + (2 * sn) * 16 + col] = gk.wk * we.y;
        blk[(2 * sn + 1) * 16 + col] = gk.wk * ev.x;
        blk[512 + (2 * sn + 1) * 16 + col] = gk.wk * ev.y;
        if (sn < 8) {  // rows 0..15 of the Gauss table (sn 7 is a Kronrod-only node: wg = 0)
            blk[1024 + (2 * sn) * 16 + col] = gk.wg * we.x;
            blk[1280 + (2 * sn) * 16 + col] = gk.wg * we.y;
            blk[1024 + (2 * sn + 1) * 16 + col] = gk.wg * ev.x;
            blk[1280 + (2 * sn + 1) * 16 + col] = gk.wg * ev.y;
        }
    }
}

struct DenseArgs {
    DevParams P;
    const ushort2* pairs;
    int npairs;
    CacheGeom geom;
    const double* recs[2];
    const double* recs_ext[2][NODE_CACHE_MAX_SUB - 1];
    const double* btab;
    const double* scale;
    unsigned long long* worklist;
    unsigned long long* defer_info;
    unsigned int* worklist_count;
    const int* act_idx;
    int n_act, nchunks;
    const double2* omega;
    double2* M;
    const double2* Mold;
    double2* Mp;
    const double2* domega;
    unsigned long long* intervals;
    int* status;
    unsigned long long* stats;  // [0] dense rounds, [1] sparse rounds, [2] sparse columns, [3] tile tasks
    int dense_min_cols;         // columns that must need an interval for the MFMA path
};

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_min_u64(unsigned long long x) {
    const unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o < x ? o : x;
}
// minimum over the whole wave, as a wave-uniform (scalar) value
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    v = dpp_min_u64<0xB1>(v);
    v = dpp_min_u64<0x4E>(v);
    v = dpp_min_u64<0x141>(v);
    v = dpp_min_u64<0x140>(v);
    unsigned long long m = ~0ull;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 16 * r);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 16 * r);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        m = o < m ? o : m;
    }
    return m;
}

#ifndef EMME_DENSE_MIN_WAVES
#define EMME_DENSE_MIN_WAVES 2
#endif

__global__ __launch_bounds__(256, EMME_DENSE_MIN_WAVES) void k_assemble_dense(DenseArgs A) {
    constexpr int KD = 56;
    const DevParams& P = A.P;
    const int N = P.N, dim = P.dim;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, rho = lane >> 4;
    // XCD-aware task order: consecutive blocks go round-robin over the 8 XCDs (block L and L + 8
    // share one), so the chunk index cycles fastest WITHIN an XCD: the waves that read a tile's
    // records for its different omega chunks run on one XCD at about the same time and share its L2
    const int ntiles = (A.npairs + TILE_PAIRS - 1) / TILE_PAIRS;
    const int ntg = (ntiles + 3) / 4;  // tile groups: 4 tiles (one per wave) per workgroup
    const int L = blockIdx.x, xcd = L & 7, q = L >> 3;
    const int chunk = q % A.nchunks;
    const int tg = (q / A.nchunks) * 8 + xcd;
    if (tg >= ntg) return;
    const int tile = tg * 4 + wave;
    if (tile >= ntiles) return;

    const int n_in_chunk = min(16, A.n_act - 16 * chunk);
    const bool has_w = col < n_in_chunk;
    const int wpos = 16 * chunk + (has_w ? col : 0);
    const int b = A.act_idx[wpos];
    cd omega = mk(A.omega[b].x, A.omega[b].y), rdw = mk(0.0, 0.0);
    if (A.Mold) rdw = rcp(mk(A.domega[b].x, A.domega[b].y));
    const unsigned long long cls = -copysign(1.0, omega.x) > 0.0 ? 0ull : 1ull;
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
    if (tile == 0 && has_w)  // diagonal (include/solver.h:442-443)
        for (int i = rho; i < N; i += 4) store(i, i, mk(P.diag_a, 0.0));

    const double inv_scale = 2. / (M_PI / 2.0);
    const unsigned long long DONE = ~0ull;
    auto make_key = [](int depth, unsigned long long path) -> unsigned long long {
        return ((path << (KD - depth)) << 6) | (unsigned long long)depth;
    };

    // ---- the wave's 256 integrals: element r of this lane = (pair tile*16 + rho + 4 r, omega col) -----
    unsigned long long key[4];
    double abs_tol[4];
    cd sum[4];
    int count[4];
    bool deferred[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int pidx = tile * TILE_PAIRS + rho + 4 * r;
        key[r] = (has_w && pidx < A.npairs) ? (cls << 63) : DONE;  // root: depth 0, path 0
        abs_tol[r] = 0.0, sum[r] = mk(0.0, 0.0), count[r] = 0, deferred[r] = false;
    }
    unsigned int n_dense = 0, n_sparse = 0, n_cols = 0;
    int bad = 0;
#ifdef EMME_DENSE_STAMPS  // diagnostic build: where a round spends its cycles (never in the product build)
    unsigned long long cyc_sel = 0, cyc_dense = 0, cyc_sparse = 0, cyc_dec = 0;
    const unsigned long long t_task = __builtin_amdgcn_s_memtime();
#define STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define STAMP(v)
#endif

    for (;;) {
        STAMP(ts0);
        unsigned long long mine = key[0] < key[1] ? key[0] : key[1];
        const unsigned long long m23 = key[2] < key[3] ? key[2] : key[3];
        mine = m23 < mine ? m23 : mine;
        const unsigned long long cur = wave_min_u64(mine);  // wave-uniform: lives in scalar registers
        if (cur == DONE) break;
        const int ccls = (int)(cur >> 63);
        const unsigned long long ck = cur & ~(1ull << 63);
        const int depth = (int)(ck & 63ull);
        const unsigned long long path = (ck >> 6) >> (KD - depth);
        bool match[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) match[r] = key[r] == cur;
        int which;
        const int cslot = A.geom.slot(depth, path, which);
        const double* ebuf = which >= 0 ? A.recs_ext[ccls][which] : A.recs[ccls];
        if (cslot < 0 || ebuf == nullptr) {
            // outside the cache: the integrals that need this interval go, whole, to the cooperative kernel
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (match[r]) {
                    const unsigned int slot = atomicAdd(A.worklist_count, 1u);
                    A.worklist[slot] = ((unsigned long long)b << 32) | (unsigned int)(tile * TILE_PAIRS + rho + 4 * r);
                    A.defer_info[slot] = ((unsigned long long)depth << 56) | ((unsigned long long)ccls << 55) |
                                         (path & 0x7fffffffffffffull);
                    deferred[r] = true;
                    key[r] = DONE;
                }
            }
            continue;
        }
        const double* ablk =
            which < 0 ? ebuf + ((size_t)tile * A.geom.ni_main() + cslot) * TILE_BLOCK
                      : ebuf + ((size_t)tile * A.geom.ni_sub(which + 1) + (cslot - A.geom.base[which + 1])) * TILE_BLOCK;
        const double* bblk = A.btab + ((size_t)cslot * A.nchunks + chunk) * BTAB_BLOCK;
        // which omega columns own this interval
        const unsigned long long need = __ballot(match[0] || match[1] || match[2] || match[3]);
        unsigned int colmask = (unsigned int)((need | (need >> 16) | (need >> 32) | (need >> 48)) & 0xffffull);
        v4d Kre = {0.0, 0.0, 0.0, 0.0}, Kim = Kre, Gre = Kre, Gim = Kre;
        STAMP(ts1);
        const bool dense_round = __popc(colmask) >= A.dense_min_cols;
        if (dense_round) {
            // ---- dense round: K = Q BK (k = 32), G = Q[:, 0:16] BG on the matrix cores ----------------
            // operand maps (one f64 per lane): A[p = lane & 15][k = 4 ks + (lane >> 4)] = ablk[k * 16 + p],
            // B[k][w = lane & 15] = bblk[k * 16 + w]: both are base + 64 ks + lane, 512 contiguous bytes
            ++n_dense;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const double are = ablk[64 * ks + lane], aim = ablk[512 + 64 * ks + lane];
                const double bre = bblk[64 * ks + lane], bim = bblk[512 + 64 * ks + lane];
                Kre = __builtin_amdgcn_mfma_f64_16x16x4f64(are, bre, Kre, 0, 0, 0);
                Kim = __builtin_amdgcn_mfma_f64_16x16x4f64(are, bim, Kim, 0, 0, 0);
                Kre = __builtin_amdgcn_mfma_f64_16x16x4f64(aim, -bim, Kre, 0, 0, 0);
                Kim = __builtin_amdgcn_mfma_f64_16x16x4f64(aim, bre, Kim, 0, 0, 0);
                if (ks < 4) {
                    const double gre = bblk[1024 + 64 * ks + lane], gim = bblk[1280 + 64 * ks + lane];
                    Gre = __builtin_amdgcn_mfma_f64_16x16x4f64(are, gre, Gre, 0, 0, 0);
                    Gim = __builtin_amdgcn_mfma_f64_16x16x4f64(are, gim, Gim, 0, 0, 0);
                    Gre = __builtin_amdgcn_mfma_f64_16x16x4f64(aim, -gim, Gre, 0, 0, 0);
                    Gim = __builtin_amdgcn_mfma_f64_16x16x4f64(aim, gre, Gim, 0, 0, 0);
                }
            }
        } else {
            // ---- sparse round: lane = node (sn = lane & 15), row rho takes pair rho + 4 r; one omega
            // column at a time, 16-lane DPP row sums; the sums land in the owner lane of each element
            ++n_sparse;
            const int sn = col;
            cd q1[4], q0[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = rho + 4 * r;
                q1[r] = mk(ablk[(2 * sn) * 16 + p], ablk[512 + (2 * sn) * 16 + p]);
                q0[r] = mk(ablk[(2 * sn + 1) * 16 + p], ablk[512 + (2 * sn + 1) * 16 + p]);
            }
            while (colmask) {
                const int c = __builtin_ctz(colmask);
                colmask &= colmask - 1;
                ++n_cols;
                const cd bk1 = mk(bblk[(2 * sn) * 16 + c], bblk[512 + (2 * sn) * 16 + c]);
                const cd bk0 = mk(bblk[(2 * sn + 1) * 16 + c], bblk[512 + (2 * sn + 1) * 16 + c]);
                cd bg1 = mk(0.0, 0.0), bg0 = mk(0.0, 0.0);
                if (sn < 8) {
                    bg1 = mk(bblk[1024 + (2 * sn) * 16 + c], bblk[1280 + (2 * sn) * 16 + c]);
                    bg0 = mk(bblk[1024 + (2 * sn + 1) * 16 + c], bblk[1280 + (2 * sn + 1) * 16 + c]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const cd fk = q1[r] * bk1 + q0[r] * bk0;
                    const cd fg = q1[r] * bg1 + q0[r] * bg0;
                    const double kx = row16_sum(fk.x), ky = row16_sum(fk.y);
                    const double gx = row16_sum(fg.x), gy = row16_sum(fg.y);
                    if (col == c) Kre[r] = kx, Kim[r] = ky, Gre[r] = gx, Gim[r] = gy;
                }
            }
        }
#ifdef EMME_DENSE_STAMPS
        // (make the stamp wait for the sums: a dependent scalar read of one accumulator lane)
        const double probe = Kre[0] + Gim[3];
        int stamp_dep;
        asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(stamp_dep) : "v"(__double2loint(probe)));
#endif
        STAMP(ts2);
        // ---- every element that owns the interval decides for itself (include/functions.h:203-208,
        // 231-247); depth, path and the half-width are wave-uniform, so both possible next keys are too
        const double scale = A.scale[cslot];
        const unsigned long long key_split = make_key(depth + 1, path << 1) | ((unsigned long long)ccls << 63);
        unsigned long long p2 = path + 1;
        const int tz = min(depth, (int)__builtin_ctzll(p2 | (1ull << 63)));
        p2 >>= tz;
        const int d2 = depth - tz;
        const unsigned long long key_next = d2 == 0 ? DONE : (make_key(d2, p2) | ((unsigned long long)ccls << 63));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (__ballot(match[r]) == 0ull) continue;  // wave-uniform
            if (match[r]) {
                ++count[r];
                const double kx = Kre[r], ky = Kim[r];
                const double dKx = kx - Gre[r], dKy = ky - Gim[r];
                const double absK = sqrt(fma(kx, kx, ky * ky));
                double err = fmax(sqrt(fma(dKx, dKx, dKy * dKy)), absK * (2.0 * 2.220446049250313e-16));
                err *= scale;
                const double rel_abs = P.rel_tol * (absK * scale);
                if (abs_tol[r] == 0.0) abs_tol[r] = rel_abs;
                bool split = depth < P.max_sub && err > abs_tol[r] * inv_scale + P.prec_goal &&
                             err > rel_abs + P.prec_goal;
                if (split && (depth >= EMME_MAX_DEPTH || count[r] >= EMME_MAX_INTERVALS)) {
                    split = false;
                    bad = 1;
                }
                if (split) {
                    key[r] = key_split;
                } else {
                    sum[r] = sum[r] + mk(kx * scale, ky * scale);
                    key[r] = key_next;
                }
            }
        }
#ifdef EMME_DENSE_STAMPS
        {
            int stamp_dep2;
            asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(stamp_dep2) : "v"((int)(unsigned)key[0]));
            STAMP(ts3);
            cyc_sel += ts1 - ts0;
            if (dense_round) cyc_dense += ts2 - ts1; else cyc_sparse += ts2 - ts1;
            cyc_dec += ts3 - ts2;
        }
#endif
    }

    // ---- results (include/solver.h:448-455: mat(i,j) = -kappa W_ij dx, mirrored) ---------------------
    unsigned long long my_intervals = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int pidx = tile * TILE_PAIRS + rho + 4 * r;
        if (has_w && pidx < A.npairs && !deferred[r]) {
            my_intervals += (unsigned long long)count[r];
            const ushort2 ij = A.pairs[pidx];
            const int i = ij.x, j = ij.y;
            const cd kap = mk(P.pref * sum[r].y, -(P.pref * sum[r].x));  // -i pref sum, Parameters.cpp:182
            if (!(isfinite(kap.x) && isfinite(kap.y))) bad = 1;
            // (the adiabatic-electron term kappa_e is zero for moment 0, src/Parameters.cpp:191-193)
            const cd v = (-(pair_weight(i, j, N) * P.dx)) * kap;
            store(i, j, v);
            store(j, i, v);
        }
    }
    if (has_w) {
        if (A.intervals && my_intervals) atomicAdd(&A.intervals[b], my_intervals);
        if (bad) A.status[b] = 1;
    }
#ifdef EMME_DENSE_STAMPS
    if (A.stats && lane == 0) {
        atomicAdd(&A.stats[4], cyc_sel);
        atomicAdd(&A.stats[5], cyc_dense);
        atomicAdd(&A.stats[6], cyc_sparse);
        atomicAdd(&A.stats[7], cyc_dec);
        atomicAdd(&A.stats[8], __builtin_amdgcn_s_memtime() - t_task);
        atomicMax(&A.stats[9], __builtin_amdgcn_s_memtime() - t_task);
    }
#endif
    if (A.stats && lane == 0) {
        atomicAdd(&A.stats[0], (unsigned long long)n_dense);
        atomicAdd(&A.stats[1], (unsigned long long)n_sparse);
        atomicAdd(&A.stats[2], (unsigned long long)n_cols);
        atomicAdd(&A.stats[3], 1ull);
    }
}

}  // namespace

size_t node_cache_bytes_tiled(long npairs, const NodeCacheGeom& g, int part) {
    const CacheGeom c = make_geom(g);
    const int ni = part < 0 ? c.ni_main() : c.ni_sub(part + 1);
    const size_t ntiles = (size_t)((npairs + TILE_PAIRS - 1) / TILE_PAIRS);
    return ntiles * (size_t)ni * TILE_BLOCK * sizeof(double);
}

size_t btab_bytes(int nslots, int nchunks) { return (size_t)nslots * nchunks * BTAB_BLOCK * sizeof(double); }

hipError_t launch_node_cache_tiled(const AssembleLaunch& L, const NodeCacheGeom& g, int part, double omi, void* recs,
                                   void* ttab, double* scale, hipStream_t stream) {
    TiledCacheArgs A;
    A.P = L.P;
    A.tab = L.tab;
    A.pairs = (const ushort2*)L.pairs;
    A.npairs = L.npairs;
    A.geom = make_geom(g);
    A.omi = omi;
    A.recs = (double*)recs;
    A.ttab = (double2*)ttab;
    A.scale = scale;
    A.part = part;
    A.first = part < 0 ? 0 : A.geom.base[part + 1];
    A.count = part < 0 ? A.geom.ni_main() : A.geom.ni_sub(part + 1);
    if (A.count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_node_cache_tiled, dim3(256 * 32), dim3(256), 0, stream, A);
    return hipGetLastError();
}

hipError_t launch_btab(int nslots, const void* const ttab[2], const double* omega, const int* act_idx, int n_act,
                       void* btab, hipStream_t stream) {
    BtabArgs A;
    A.ttab[0] = (const double2*)ttab[0], A.ttab[1] = (const double2*)ttab[1];
    A.omega = (const double2*)omega;
    A.act_idx = act_idx;
    A.n_act = n_act;
    A.nchunks = (n_act + 15) / 16;
    A.nslots = nslots;
    A.btab = (double*)btab;
    const long total = (long)nslots * 16 * n_act;
    long blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_btab, dim3((unsigned)blocks), dim3(256), 0, stream, A);
    return hipGetLastError();
}

hipError_t launch_assemble_dense(const AssembleLaunch& L, const NodeCacheGeom& g, const void* const recs[2],
                                 const void* const recs_ext[2][NODE_CACHE_MAX_SUB - 1], const double* scale,
                                 const void* btab, unsigned long long* worklist, unsigned int* worklist_count,
                                 unsigned long long* defer_info, const int* act_idx, int n_act,
                                 unsigned long long* stats, hipStream_t stream) {
    DenseArgs A;
    A.P = L.P;
    A.pairs = (const ushort2*)L.pairs;
    A.npairs = L.npairs;
    A.geom = make_geom(g);
    for (int c = 0; c < 2; ++c) {
        A.recs[c] = (const double*)recs[c];
        for (int k = 0; k < NODE_CACHE_MAX_SUB - 1; ++k) A.recs_ext[c][k] = (const double*)recs_ext[c][k];
    }
    A.btab = (const double*)btab;
    A.scale = scale;
    A.worklist = worklist;
    A.worklist_count = worklist_count;
    A.defer_info = defer_info;
    A.act_idx = act_idx;
    A.n_act = n_act;
    A.nchunks = (n_act + 15) / 16;
    A.omega = (const double2*)L.omega;
    A.M = (double2*)L.M;
    A.Mold = (const double2*)L.Mold;
    A.Mp = (double2*)L.Mp;
    A.domega = (const double2*)L.domega;
    A.intervals = L.intervals;
    A.status = L.status;
    A.stats = stats;
    const char* e = std::getenv("EMME_DENSE_MIN_COLS");
    A.dense_min_cols = e ? std::atoi(e) : 3;
    const int ntiles = (L.npairs + TILE_PAIRS - 1) / TILE_PAIRS;
    const int ntg = (ntiles + 3) / 4;
    const long blocks = (long)((ntg + 7) / 8) * A.nchunks * 8;
    hipLaunchKernelGGL(k_assemble_dense, dim3((unsigned)blocks), dim3(256), 0, stream, A);
    return hipGetLastError();
}

}  // namespace emme
