// assemble_dense.hip -- the electrostatic GK15 fill as small complex GEMMs on the FP64 matrix cores.
//
// With the node records folded (emme_device.hpp::node_data, assemble_cached.hip) the Kronrod and
// Gauss sums of one quadrature interval are, for pair p and omega w,
//     K[p, w] = sum_n  wk_n E[n, w] (w Q1[p, n] + Q0[p, n])  =  sum_k Q[p, k] BK[k, w]
//     G[p, w] =                                                  sum_k Q[p, k] BG[k, w]   (k < 16)
// with Q[p, 2n] = exp(A0) Q1, Q[p, 2n+1] = exp(A0) Q0 (omega-independent, HBM node cache) and
// BK[2n, w] = wk_n w E[n, w], BK[2n+1, w] = wk_n E[n, w], E = exp(T_n w) (pair-independent, one
// table per launch): a complex 16 x 16 x 32 GEMM per (16 pairs, 16 omegas, interval).  One wave owns
// a TILE of 16 consecutive pairs x one chunk of <= 16 cost-sorted omegas = 256 integrals
// (include/solver.h:446-455 fills them one task each) and walks the UNION of their adaptive trees
// level by level (see k_assemble_dense), one interval per round:
//   dense round  (the interval is in the trees of >= 3 omega columns): 48 v_mfma_f64_16x16x4_f64 --
//                 32 for K (k = 32), 16 for G (k = 16; the Gauss operand is the Kronrod operand scaled
//                 by wg/wk, so G needs no table of its own) -- fed by 16 coalesced 1-KB loads, no
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
    double2* ttab;   // [NI][GW]  T per (interval, node lane)
    double2* wtab;   // [NI][GW]  W per (interval, node lane): the moment factor of electromagnetic fills (null: none)
    double* scale;   // [NI]
    int part, first, count;
    unsigned char* tile_poison;  // [ntiles] of this contour class: 1 if any (pair, interval) block of the tile is poisoned
};

// One lane group (16 lanes for GK15, 32 for GK31) per (pair, interval), lane = node (gk_lane<PTS>): computes the
// folded record and scatters it into the tile block.
template <int PTS>
__global__ __launch_bounds__(256) void k_node_cache_tiled(TiledCacheArgs A) {
    constexpr int GW = tile_slots(PTS), GROUPS_PER_BLOCK = 256 / GW, TB = tile_block_doubles(PTS);
    const DevParams& P = A.P;
    const int N = P.N, NI = A.count;
    const int lane = threadIdx.x % GW;
    const int ntiles = (A.npairs + TILE_PAIRS - 1) / TILE_PAIRS;
    const long total = (long)ntiles * TILE_PAIRS * NI;
    const GkLane gk = gk_lane<PTS>(lane);
    const double* eta = A.tab;
    const double* gtab = A.tab + N;
    const double* btab = A.tab + 2 * N;
    const int sn = slotnode_of_lane_t<PTS>(lane);
    for (long w = (long)blockIdx.x * GROUPS_PER_BLOCK + threadIdx.x / GW; w < total;
         w += (long)gridDim.x * GROUPS_PER_BLOCK) {
        // w = (tile * NI + idx) * 16 + p: the 16 pairs of a (tile, interval) block are neighbours
        const int p = (int)(w % TILE_PAIRS);
        const long ti = w / TILE_PAIRS;
        const int idx = (int)(ti % NI);
        const long tile = ti / NI;
        const long item = tile * TILE_PAIRS + p;
        double* blk = A.recs + ti * TB;
        cd q1 = mk(0.0, 0.0), q0 = mk(0.0, 0.0);
        bool over = false;
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
            if (item < A.npairs && lane < PTS) {
                double sa, ca;
                sincos(d.A0.y, &sa, &ca);
                const double ea = exp(d.A0.x);
                const cd ex = mk(ea * ca, ea * sa);
                q1 = ex * d.Q1, q0 = ex * d.Q0;
                if (!(isfinite(q1.x) && isfinite(q1.y) && isfinite(q0.x) && isfinite(q0.y))) {
                    // Re A0 << 0: exp(A0) = 0 against an overflowing amplitude -- the reference's clamp
                    // (Re(A0 + T omega) < -40, src/Parameters.cpp:167-173) makes this node contribute exactly 0 for
                    // every omega the integrand is finite for.  Otherwise the folded amplitude itself is not
                    // representable (a near-pole of 1/lambda: Re A0 > 709, or exp(A0) Q overflows): the block
                    // is POISONED below -- whoever needs this (pair, interval) evaluates it unfolded, from scratch.
                    over = d.A0.x > -700.0;
                    q1 = mk(0.0, 0.0), q0 = mk(0.0, 0.0);
                }
            }
            if (item == 0) {
                A.ttab[(long)(A.first + idx) * GW + lane] = make_double2(d.T.x, d.T.y);
                if (A.wtab) {
                    const cd wv = node_w(x, P, A.omi);
                    A.wtab[(long)(A.first + idx) * GW + lane] = make_double2(wv.x, wv.y);
                }
                if (lane == 0) A.scale[A.first + idx] = scale;
            }
        }
        // poisoned (pair, interval): all of its records are zeroed (the GEMMs of the tile's other pairs stay finite)
        // and the padding slot [sn = GW - 1][p][0] carries the flag -- it multiplies a zero row of the phase block
        const bool poisoned = GW == 16 ? ((__ballot(over) >> (threadIdx.x & 48)) & 0xffffull) != 0ull
                                       : ((__ballot(over) >> (threadIdx.x & 32)) & 0xffffffffull) != 0ull;
        if (poisoned) {
            q1 = mk(lane == GW - 1 ? 1.0 : 0.0, 0.0), q0 = mk(0.0, 0.0);
            if (lane == GW - 1 && A.tile_poison) A.tile_poison[tile] = 1;
        }
        double2* q = reinterpret_cast<double2*>(blk);
        q[tile_index(2 * sn, p)] = make_double2(q1.x, q1.y);
        q[tile_index(2 * sn + 1, p)] = make_double2(q0.x, q0.y);
    }
}

// Weighted phase tables of one launch (see node_cache.hpp: BTAB_BLOCK).
struct BtabArgs {
    const double2* ttab[2];
    const double2* wtab[2];  // electromagnetic fills: W per (interval, node lane)
    const double2* omega;
    const int* act_idx;
    const int* wmap;  // per position of the launch's omega list: chunk << 8 | column
    int n_act, nchunks, nslots;
    double* btab;
};
// Electromagnetic fills (NM = 3): the moment factor of F_m = F_0 (c_nv W)^m has a pair-independent part, W^m, which
// belongs to the phase operand: omega position w of a chunk owns the three columns 3 w + m with wk E W^m (the
// real factor c_nv^m of the pair is applied to the sums before the decisions, k_assemble_dense).
template <int PTS, int NM>
__global__ __launch_bounds__(256) void k_btab(BtabArgs A) {
    constexpr int GW = tile_slots(PTS), BT = btab_block_doubles(PTS);
    const long total = (long)A.nslots * GW * A.n_act;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        // e = (slot * 16 + lane) * n_act + wpos: neighbouring threads = neighbouring omega columns of one
        // table row, so every store instruction writes whole 128-byte row segments
        const long row = e / A.n_act;
        const int wpos = (int)(e - row * A.n_act);
        const int lane = (int)(row % GW);
        const int slot = (int)(row / GW);
        const double2 om = A.omega[A.act_idx[wpos]];
        const int cls = -copysign(1.0, om.x) > 0.0 ? 0 : 1;
        cd ev = mk(0.0, 0.0);
        if (A.ttab[cls] && lane < PTS) {
            const double2 t = A.ttab[cls][(long)slot * GW + lane];
            const double ax = fma(t.x, om.x, -(t.y * om.y)), ay = fma(t.x, om.y, t.y * om.x);
            if (!(ax > 700.0)) {  // (a NaN omega goes through and poisons its own column only)
                double sa, ca;
                sincos(ay, &sa, &ca);
                const double ea = exp(ax);
                ev = mk(ea * ca, ea * sa);
            } else {
                // exp(T omega) beyond 1e304: the reference's exp(A0 + T omega) overflows here or is about to
                // (Re A0 = O(1) wherever t is large enough for this).  NaN: an integral of this omega that
                // uses the node ends non-finite and flags its matrix (EMME_ENUMERIC) instead of dropping the term
                ev = mk(__builtin_nan(""), __builtin_nan(""));
            }
        }
        const GkLane gk = gk_lane<PTS>(lane);
        const int sn = slotnode_of_lane_t<PTS>(lane);
        const int wm = A.wmap[wpos];
        double* blk = A.btab + ((size_t)slot * A.nchunks + (wm >> 8)) * BT;
        const int col = NM * (wm & 255);
        double2* bk = reinterpret_cast<double2*>(blk);
        cd bv = mk(gk.wk * ev.x, gk.wk * ev.y);
        bk[sn * 16 + col] = make_double2(bv.x, bv.y);
        if (NM > 1) {
            cd wv = mk(0.0, 0.0);
            if (A.wtab[cls] && lane < PTS) {
                const double2 w2 = A.wtab[cls][(long)slot * GW + lane];
                wv = mk(w2.x, w2.y);
            }
#pragma unroll
            for (int m = 1; m < NM; ++m) {
                bv = bv * wv;
                bk[sn * 16 + col + m] = make_double2(bv.x, bv.y);
            }
        }
    }
}

// (wg / wk) of node slot sn: the Gauss rule's weight relative to the Kronrod weight; 0 for Kronrod-only nodes
template <int PTS>
__device__ __forceinline__ double gauss_ratio(int sn) {
    if (sn >= (PTS - 1) / 2) return 0.0;          // (7 / 15 Gauss nodes, slots 0 ..)
    const int q = sn == 0 ? 0 : ((sn + 1) & ~1);  // slots (1,2) (3,4) (5,6) .. are nodes +-x2, +-x4, +-x6 ..
    return PTS == 15 ? kWg15[q >> 1] / kWk15[q] : kWg31[q >> 1] / kWk31[q];
}

struct DenseArgs {
    DevParams P;
    const double* tab;  // eta | g | b (electromagnetic fills: c_nv, kappa_e and the D diagonal)
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
    const int2* chunks;  // (first position, size <= 16) of every omega chunk
    int n_act, nchunks;
    const double2* omega;
    double2* M;
    const double2* Mold;
    double2* Mp;
    const double2* domega;
    unsigned long long* intervals;
    int* status;
    unsigned long long* stats;  // [0] dense rounds, [1] sparse rounds, [2] sparse columns, [3] tile tasks
    const unsigned char* tile_poison[2];  // per contour class: tiles that hold a poisoned (pair, interval) block
    int dense_min_cols;         // columns that must need an interval for the MFMA path
    int skip_lost;              // columns whose matrix is already flagged (status) are left alone
    int chunk0;                 // first chunk of this launch
    int nch_launch;             // chunks this launch serves (chunk0 .. chunk0 + nch_launch - 1)
    int tile_major;             // task order: the chunks of a tile group back to back on one XCD (else chunk-major)
    unsigned int* overflow;     // [nbatch] integrals handed over because a level list was full (null: not counted)
};

// ---- the fill: level by level ----------------------------------------------------------------------
// (A first version walked the union tree in pre-order like k_assemble_union, one interval per round
// chosen by a wave-wide key minimum: every round was a chain of exposed latencies and the tasks of the
// expensive omega chunk, interleaved with the others, finished last: 128 ms per bench search against 66
// for this form.)  The 256 integrals of a wave advance one bisection LEVEL at a time: the intervals of the current level that
// any element needs form a list of at most 64 entries (entry e lives in lane e of a register pair, read
// with v_readlane), every element holds a 64-bit mask of the entries it needs, and an entry that some
// element splits appends its two children to the next level's list.  Entries of a level are independent
// of each other, there is no key arithmetic and no minimum search, and the accepted pieces are added
// level by level instead of left to right (a rounding-level change, like the cooperative kernel's).
// An element whose split does not fit the next list (more than 64 intervals of one level in a tile: in
// the bench, one strongly damped omega per search whose trees also leave the cached depth)
// is handed, whole, to the cooperative kernel.
__device__ __forceinline__ double fsqrt_pos(double x) {
    // sqrt for the error estimates: hardware reciprocal-square-root seed + two Newton steps (<= 1 ulp
    // for normal arguments), 0 for 0 and NaN for NaN
    const double y = x * frsqrt(x);
    return x > 0.0 ? y : x;
}

#ifndef EMME_DENSE_MIN_WAVES
#define EMME_DENSE_MIN_WAVES 2
#endif

// LW: words of a level list -- 64 LW entries per level.  LW = 1 is the kernel of every ordinary omega.  The trees of
// an omega far below the real axis (Im omega = -6.7: 317 intervals per integral, 78 of them on one level) do not fit
// 64 entries: their elements used to be handed, one by one, to the cooperative kernel -- all 32 640 integrals of the
// omega, every Newton step (5 ms per step for two such chains).  The host sends the chunks of such omegas to the
// LW = 2 build (128 entries, 128-bit element masks: 20 more vector registers, a launch of its own).
//
// PTS = 31: 32 node slots, a complex 16 x 16 x 64 GEMM per round (two batches of eight k-steps; the embedded 15-point
// Gauss rule lives in the first).  NM = 3 (electromagnetic, include/solver.h:461-511): the three velocity moments
// F_m = F_0 (c_nv W)^m of a pair share the records; W^m sits in the phase operand (k_btab), so a chunk is <= 5 omegas x 3
// moments = 15 COLUMNS (column 3 w + m) of the same GEMM, every (pair, omega, moment) integral decides for itself as
// before, and the pair's real factor c_nv^m multiplies its sums before the decision (k_assemble_cached_em walks the
// union of the three moments' trees per lane; here they are three columns of the tile's union).
template <int LW, int PTS, int NM>
__global__ __launch_bounds__(256, EMME_DENSE_MIN_WAVES) void k_assemble_dense(DenseArgs A) {
    constexpr int NS = tile_slots(PTS), KS = NS / 2, GKS = KS / 2;  // node slots, k-steps of K, k-steps that feed G too
    constexpr int TB = tile_block_doubles(PTS), BT = btab_block_doubles(PTS);
    const DevParams& P = A.P;
    const int N = P.N, dim = P.dim;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, rho = lane >> 4;
    const int wcol = NM == 1 ? col : col / NM;      // omega position of this lane's column in its chunk
    const int mom = NM == 1 ? 0 : col - wcol * NM;  // its velocity moment
    // task order: the chunks of the cost-sorted omega list hold the most expensive omegas first, and
    // their tasks are the longest: chunk-major, so that they all start at once and the cheap ones fill in
    const int ntiles = (A.npairs + TILE_PAIRS - 1) / TILE_PAIRS;
    const int ntg = (ntiles + 3) / 4;  // tile groups: 4 tiles (one per wave) per workgroup
    int chunk = A.chunk0 + blockIdx.x / ntg;
    int tile = (blockIdx.x - (chunk - A.chunk0) * ntg) * 4 + wave;
    if (NM > 1 || A.tile_major) {
        // Electromagnetic launches have 3.2 times the chunks (5 omegas each) and 16-KB record blocks: in chunk-major
        // order every chunk fetched every block of its tiles from HBM again (26 chunks x 390 MB per launch).  Here
        // the chunks of a tile group run back to back ON ONE XCD (workgroup ids go round the eight XCDs), whose L2
        // then serves the tile's blocks to all but the first.
        const int nch = A.nch_launch;
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int tg = (k / nch) * 8 + xcd;
        chunk = A.chunk0 + k % nch;
        tile = tg * 4 + wave;
    }
    // Counters leave the workgroup once: its waves add them up in LDS and the last one to finish carries the
    // sums to memory.  (Every wave of a launch adds to the same <= 128 interval counters and four round
    // counters: at one global atomic per lane -- 1.2 million per launch -- the launch waited for them, 10 % of
    // its time; one more per wave cost another 3 %.)
    __shared__ unsigned long long s_iv[16];
    __shared__ unsigned int s_st[4];
    __shared__ int s_arrived;
    if (threadIdx.x < 16) s_iv[threadIdx.x] = 0ull;
    if (threadIdx.x < 4) s_st[threadIdx.x] = 0u;
    if (threadIdx.x == 0) s_arrived = 0;
    __syncthreads();
    if (tile >= ntiles) return;
    const int waves_here = min(4, ntiles - (tile - wave));  // waves of this workgroup that own a tile

    const int2 ch = A.chunks[chunk];
    const bool in_chunk = wcol < ch.y;
    const int wpos = ch.x + (in_chunk ? wcol : 0);
    const int b = A.act_idx[wpos];
    // A matrix that already holds a non-finite integral is lost (its chain retires at the next Newton step, as the
    // reference's does: include/solver.h:142-153): nobody works on it any more.  (The four lanes of a column read
    // the flag in one instruction: they agree.)
    const bool has_w = in_chunk && !(A.skip_lost && A.status[b] != 0);
    const int cls = -copysign(1.0, A.omega[b].x) > 0.0 ? 0 : 1;
    // (everything the stores need is fetched again after the walk: nothing of it stays live in the loop)
    auto store = [&](int r, int c, cd v, cd rdw) {
#ifdef EMME_EXP_UPPER_ONLY  // timing experiment: what the mirrored half of the stores costs
        if (r > c) return;
#endif
        const size_t idx = (size_t)b * dim * dim + (size_t)r * dim + c;
        A.M[idx] = make_double2(v.x, v.y);
        if (A.Mold) {
            const double2 o = A.Mold[idx];
            const cd d = (v - mk(o.x, o.y)) * rdw;
            A.Mp[idx] = make_double2(d.x, d.y);
        }
    };
    if (tile == 0 && has_w && mom == 0) {  // diagonal (include/solver.h:442-443; electromagnetic: 465-470)
        const cd rdw0 = A.Mold ? rcp(mk(A.domega[b].x, A.domega[b].y)) : mk(0.0, 0.0);
        for (int i = rho; i < N; i += 4) {
            store(i, i, mk(P.diag_a, 0.0), rdw0);
            if (NM > 1) {
                store(i, i + N, mk(0.0, 0.0), rdw0);
                store(i + N, i, mk(0.0, 0.0), rdw0);
                store(i + N, i + N, mk(P.diag_d * A.tab[2 * N + i], 0.0), rdw0);
            }
        }
    }
    // electromagnetic: c_nv of the tile's 16 pairs (emme_device.hpp: norm_vel = c_nv W; as make_pair_const)
    __shared__ double s_cnv[4][16];
    if (NM > 1 && lane < 16) {
        const int pidx = tile * TILE_PAIRS + lane;
        double v = 0.0;
        if (pidx < A.npairs) {
            const ushort2 ij = A.pairs[pidx];
            v = (P.qR * (A.tab[ij.x] - A.tab[ij.y])) / P.vt;
        }
        s_cnv[wave][lane] = v;
    }
    // c_nv^m of pair slot q of this tile (1 for electrostatic fills); m is the column's moment
    auto moment_factor = [&](int q, int m) -> double {
        if (NM == 1) return 1.0;
        const double cv = s_cnv[wave][q];
        return m == 0 ? 1.0 : (m == 1 ? cv : cv * cv);
    };

    const double inv_scale = 2. / (M_PI / 2.0);
    // (wg / wk) of this lane's rows: as MFMA A operand (row 4 ks + (lane >> 4), ks < 4) and as node lane & 15
    double grat[GKS];
#pragma unroll
    for (int ks = 0; ks < GKS; ++ks) grat[ks] = gauss_ratio<PTS>((4 * ks + (lane >> 4)) >> 1);
    const double grat_node = gauss_ratio<PTS>(col);  // (node slots 16 .. 31 of GK31 are Kronrod-only)
    const int loff = tile_index(lane >> 4, lane & 15);  // this lane's element of an MFMA operand load, k-step 0
    const int eoff = (lane >> 5) * 16 + (lane & 15);    // the same for the phase block: node 2 ks + (rho >> 1)
    const double2 omw = A.omega[b];                     // this lane's column omega
    // The cache geometry, one subtree per lane (lanes 0 .. nsub-1): the slot look-up of an entry is then a few
    // vector instructions, a ballot and v_readlane's.  (The scalar loop over the subtrees read rd / dd / rp /
    // base, the buffer pointer and the block counts from the kernel arguments: up to a dozen DEPENDENT scalar
    // loads per entry below the full tree -- 700 cycles of a vector round's 4 000.)
    const int g_dfull = A.geom.dfull;
    const bool g_on = lane < A.geom.nsub;
    const int gk = g_on ? lane : 0;
    const int g_rd = A.geom.rd[gk], g_dd = A.geom.dd[gk], g_base = A.geom.base[gk];
    const unsigned long long g_rp = A.geom.rp[gk];
    // records of subtree 0 live behind the full tree in the main buffer; the others have buffers of their own
    const double* g_ptr0 = gk == 0 ? A.recs[0] : A.recs_ext[0][gk - 1];
    const double* g_ptr1 = gk == 0 ? A.recs[1] : A.recs_ext[1][gk - 1];
    const unsigned long long g_blk_main = (unsigned long long)tile * (unsigned long long)A.geom.ni_main();
    // block number of the subtree's first interval for this wave's tile in its buffer
    const unsigned long long g_blk0 =
        gk == 0 ? g_blk_main + (unsigned long long)g_base : (unsigned long long)tile * (unsigned long long)((2 << (g_dd - g_rd)) - 1);
    auto lane_ptr = [&](const double* p, int k) -> const double* {
        const unsigned long long bits = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, k);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), k);
        return reinterpret_cast<const double*>(((unsigned long long)hi << 32) | lo);
    };
    // ---- the wave's 256 integrals: element r of this lane = (pair tile*16 + rho + 4 r, omega col) -----
    unsigned long long mcur[4][LW], mnext[4][LW];  // entries of the current / next level this element needs
    // per-element accumulators live in LDS (touched only by their owner lane, only when the element owns
    // the entry): 28 VGPRs less per lane, which is what lets a third wave onto the SIMD
    __shared__ double s_sumx[4][4][64], s_sumy[4][4][64], s_abstol[4][4][64];
    __shared__ int s_count[4][4][64];
    bool deferred[4], alive[4];
    // level lists: entry e of a level = (contour class << 62 | path) in lane e of (E_lo, E_hi)
    unsigned int ecur_lo[LW], ecur_hi[LW], enext_lo[LW], enext_hi[LW];
#pragma unroll
    for (int q = 0; q < LW; ++q) ecur_lo[q] = 0, ecur_hi[q] = 0, enext_lo[q] = 0, enext_hi[q] = 0;
    int n_cur = 0;
    {
        // level 0: the root interval, once per contour class present among this chunk's omegas
        const unsigned long long c0 = __ballot(has_w && cls == 0), c1 = __ballot(has_w && cls == 1);
        int e_of_cls[2] = {-1, -1};
        if (c0) e_of_cls[0] = n_cur++;
        if (c1) e_of_cls[1] = n_cur++;
        if (c1) {
            ecur_hi[0] = lane == e_of_cls[1] ? (1u << 30) : ecur_hi[0];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pidx = tile * TILE_PAIRS + rho + 4 * r;
            alive[r] = has_w && pidx < A.npairs;
#pragma unroll
            for (int q = 0; q < LW; ++q) mcur[r][q] = 0ull, mnext[r][q] = 0ull;
            mcur[r][0] = alive[r] ? (1ull << e_of_cls[cls]) : 0ull;
            s_abstol[wave][r][lane] = 0.0, s_sumx[wave][r][lane] = 0.0, s_sumy[wave][r][lane] = 0.0;
            s_count[wave][r][lane] = 0, deferred[r] = false;
        }
    }
    unsigned int n_dense = 0, n_sparse = 0, n_cols = 0;
    int bad = 0;

#ifdef EMME_DENSE_STAMPS  // diagnostic build: where an entry spends its cycles (never in the product build)
    unsigned long long cyc_sel = 0, cyc_dense = 0, cyc_sparse = 0, cyc_dec = 0;
    const unsigned long long t_task = __builtin_amdgcn_s_memtime();
#define STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define STAMP(v)
#endif
    auto defer = [&](int r, int depth, int ccls, unsigned long long path) {
        const unsigned int slot = atomicAdd(A.worklist_count, 1u);
        A.worklist[slot] = ((unsigned long long)b << 32) | (unsigned int)((tile * TILE_PAIRS + rho + 4 * r) * NM + mom);
        A.defer_info[slot] = ((unsigned long long)depth << 56) | ((unsigned long long)ccls << 55) | (path & 0x7fffffffffffffull);
        deferred[r] = true, alive[r] = false;
#pragma unroll
        for (int q = 0; q < LW; ++q) mcur[r][q] = 0ull, mnext[r][q] = 0ull;
    };

    // Poisoned blocks (k_node_cache_tiled: a folded amplitude that is not representable -- a handful of pairs with a
    // near-pole of 1/lambda; 4 of the 2 040 tiles of the bench grid, contour class Re omega > 0 only) are not looked
    // for in the rounds: a tile that holds one hands ALL its integrals of that contour class to the cooperative
    // kernel right here, which evaluates poisoned blocks unfolded, from scratch.  (Looking for the flag in the rounds
    // -- the padding row of every block -- cost 7 % of the fill: 17 more vector registers, 36.1 -> 38.9 ms per search.)
    if (has_w && A.tile_poison[cls] && A.tile_poison[cls][tile] != 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (alive[r]) defer(r, 0, cls, 0ull);
    }

    for (int depth = 0; n_cur > 0; ++depth) {
        int n_next = 0;
        for (int e = 0; e < n_cur; ++e) {
            STAMP(ts0);
            // (entry e: lane e & 63 of word e >> 6 -- e is wave-uniform)
            const unsigned int elo = (unsigned)__builtin_amdgcn_readlane((int)(LW > 1 && e >= 64 ? ecur_lo[LW - 1] : ecur_lo[0]), e & 63);
            const unsigned int ehi = (unsigned)__builtin_amdgcn_readlane((int)(LW > 1 && e >= 64 ? ecur_hi[LW - 1] : ecur_hi[0]), e & 63);
            const int ccls = (int)(ehi >> 30);
            const unsigned long long path = (((unsigned long long)(ehi & 0x3fffffffu)) << 32) | elo;
            bool match[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) match[r] = (((LW > 1 && e >= 64 ? mcur[r][LW - 1] : mcur[r][0]) >> (e & 63)) & 1ull) != 0ull;
            const unsigned long long need = __ballot(match[0] || match[1] || match[2] || match[3]);
            if (need == 0ull) continue;  // (its owners were deferred meanwhile)
            // slot of the interval in the cache and its record block for this tile (CacheGeom::slot, one subtree
            // per lane: see g_rd above)
            int cslot;
            unsigned long long blk;
            const double* ebuf;
            if (depth <= g_dfull) {
                cslot = (1 << depth) - 1 + (int)path;
                blk = g_blk_main + (unsigned long long)cslot;
                ebuf = lane_ptr(ccls ? g_ptr1 : g_ptr0, 0);
            } else {
                const int sd = (depth - g_rd) & 63;
                const bool hit = g_on && depth <= g_dd && depth >= g_rd && (path >> sd) == g_rp;
                const unsigned long long hb = __ballot(hit);
                const int k = hb ? __builtin_ctzll(hb) : 0;  // (the first subtree that holds it, as the scalar loop)
                const int rel = (int)((1u << sd) - 1u) + (int)(unsigned)(path & ((1ull << sd) - 1ull));
                cslot = hb ? __builtin_amdgcn_readlane(g_base + rel, k) : -1;
                const unsigned long long bl = g_blk0 + (unsigned long long)rel;
                blk = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bl >> 32), k) << 32) |
                      (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bl, k);
                ebuf = lane_ptr(ccls ? g_ptr1 : g_ptr0, k);
            }
            if (cslot < 0 || ebuf == nullptr) {
                // outside the cache: the integrals that need this interval go, whole, to the cooperative kernel
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (match[r]) defer(r, depth, ccls, path);
                continue;
            }
            const double* ablk = ebuf + blk * TB;
            const double* bblk = A.btab + ((size_t)cslot * A.nchunks + chunk) * BT;
            const double2* a2 = reinterpret_cast<const double2*>(ablk);
            const double2* b2 = reinterpret_cast<const double2*>(bblk);
            unsigned int colmask = (unsigned int)((need | (need >> 16) | (need >> 32) | (need >> 48)) & 0xffffull);
            v4d Kre = {0.0, 0.0, 0.0, 0.0}, Kim = Kre, Gre = Kre, Gim = Kre;
            const bool dense_round = __popc(colmask) >= A.dense_min_cols;
            STAMP(ts1);
#ifdef EMME_DENSE_STAMPS
            unsigned long long ts1x = ts1;
#endif
            if (dense_round) {
                ++n_dense;
                v4d K2re = {0.0, 0.0, 0.0, 0.0}, K2im = K2re, G2re = K2re, G2im = K2re;
                // k-steps 0..3 feed K and G (the embedded Gauss rule lives in rows 0..15), 4..7 only K; all 16
                // operand loads are issued before the first MFMA (one exposed latency per entry, not four)
                // operand maps: A[p = lane & 15][k = 4 ks + (lane >> 4)] = a2[tile_index(k, p)], B[k][w] likewise:
                // both are base + 64 ks + loff in (re, im) pairs -- ONE coalesced 1-KB load each
#pragma unroll
                for (int h = 0; h < KS / 8; ++h) {  // (GK31: two batches of eight k-steps, the Gauss rule in the first)
                    double2 av[8], ev[8];
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) av[ks] = a2[64 * (8 * h + ks) + loff], ev[ks] = b2[32 * (8 * h + ks) + eoff];
#ifndef EMME_DENSE_NO_SCHED_BARRIER
                    __builtin_amdgcn_sched_barrier(0);  // (the loads stay ahead of the first MFMA whatever else the scheduler weighs)
#endif
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        // (B rows 4 ks + rho belong to node 2 ks + (rho >> 1): row rho even = omega E', odd = E')
                        const double2 a = av[ks], ep = ev[ks];
                        const double2 bk = (rho & 1) ? ep : make_double2(fma(omw.x, ep.x, -(omw.y * ep.y)), fma(omw.x, ep.y, omw.y * ep.x));
                        // (eight independent accumulation chains instead of four: K's a.x and a.y products apart)
                        Kre = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bk.x, Kre, 0, 0, 0);
                        Kim = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bk.y, Kim, 0, 0, 0);
                        K2re = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, -bk.y, K2re, 0, 0, 0);
                        K2im = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bk.x, K2im, 0, 0, 0);
                        if (8 * h + ks < GKS) {  // G = sum_k (rho_k Q[p][k]) BK[k][w]: the A operand scaled, the same B
                            const double gx = a.x * grat[(8 * h + ks) % GKS], gy = a.y * grat[(8 * h + ks) % GKS];
                            Gre = __builtin_amdgcn_mfma_f64_16x16x4f64(gx, bk.x, Gre, 0, 0, 0);
                            Gim = __builtin_amdgcn_mfma_f64_16x16x4f64(gx, bk.y, Gim, 0, 0, 0);
                            G2re = __builtin_amdgcn_mfma_f64_16x16x4f64(gy, -bk.y, G2re, 0, 0, 0);
                            G2im = __builtin_amdgcn_mfma_f64_16x16x4f64(gy, bk.x, G2im, 0, 0, 0);
                        }
                    }
                }
                Kre += K2re, Kim += K2im, Gre += G2re, Gim += G2im;
            }
            // ---- every element that owns the interval decides for itself (include/functions.h:203-208,
            // 231-247); an entry somebody splits puts its two children on the next level's list
            const double scale = A.scale[cslot];
            bool split[4] = {false, false, false, false};
            // one decision: sums (kx, ky) / (gx, gy) of the element whose accumulators are slot [r][owner]
            auto decide = [&](int r, int owner, double kx, double ky, double gx, double gy, int& flag_bad) -> bool {
                const int cnt = s_count[wave][r][owner] + 1;
                s_count[wave][r][owner] = cnt;
                const double dKx = kx - gx, dKy = ky - gy;
                const double absK = fsqrt_pos(fma(kx, kx, ky * ky));
                double err = fmax(fsqrt_pos(fma(dKx, dKx, dKy * dKy)), absK * (2.0 * 2.220446049250313e-16));
                err *= scale;
                const double rel_abs = P.rel_tol * (absK * scale);
                double at = s_abstol[wave][r][owner];
                if (at == 0.0) {
                    at = rel_abs;
                    s_abstol[wave][r][owner] = at;
                }
                bool sp = depth < P.max_sub && err > at * inv_scale + P.prec_goal && err > rel_abs + P.prec_goal;
                if (sp && (depth >= EMME_MAX_DEPTH || cnt >= EMME_MAX_INTERVALS)) {
                    sp = false;
                    flag_bad = 1;
                }
                if (!sp) {
                    s_sumx[wave][r][owner] += kx * scale;
                    s_sumy[wave][r][owner] += ky * scale;
                }
                return sp;
            };
            if (!dense_round) {
                // ---- vector round (one or two omega columns own the interval: a chain that wandered to a
                // damped omega refines where nobody else does): lane = node (sn = lane & 15), row rho takes
                // pair rho + 4 r; one omega column at a time, 16-lane DPP reductions (all four pairs at once).
                // (A lane = pair form with the decisions taken by 16 lanes on the LDS accumulators needs a third
                // of the instructions and was 40 % SLOWER: one dependent chain per column -- loads, shuffles
                // through LDS, square roots -- instead of four.)
                ++n_sparse;
                const int sn = col;
                unsigned long long mb[4];  // who owns the entry, per element slot
#pragma unroll
                for (int r = 0; r < 4; ++r) mb[r] = __ballot(match[r]);
                while (colmask) {
                    const int c = __builtin_ctz(colmask);
                    colmask &= colmask - 1;
                    ++n_cols;
                    double mkx = 0.0, mky = 0.0, mgx = 0.0, mgy = 0.0;
                    // (omega of column c: it lives in lane c; c is wave-uniform, so v_readlane, not a bpermute through LDS)
                    auto lane_value = [&](double v) -> double {
                        const long long bits = __double_as_longlong(v);
                        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, c);
                        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), c);
                        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
                    };
                    const double wcx = lane_value(omw.x), wcy = lane_value(omw.y);
                    // the four pairs' node products, then ONE reduction per quantity for all four: two halving
                    // exchanges (row_mirror, row_half_mirror: a lane keeps half of what it holds and adds the
                    // partner's copy of it), then two butterfly steps inside the quad -- 5 additions per quantity
                    // instead of 16.  The lane takes its pairs in the order r ^ m, m = 2 (lane >= 8) + (lane / 4 odd),
                    // so that what it keeps is always its slots 0, 1 (then 0) and what the partner wants its slots
                    // 3, 2 (row_mirror flips both bits of m; then slot 1: row_half_mirror flips the low one): no
                    // selects.  Lanes 4 q .. 4 q + 3 end up with the sums of pair rho + 4 q.
                    const int pmask = (col >= 8 ? 2 : 0) | ((col >> 2) & 1);
                    double pkx[4], pky[4], pgx[4], pgy[4];
#pragma unroll
                    for (int h = 0; h < NS / 16; ++h) {  // (GK31: a lane takes node slots col and col + 16)
                        const int snh = sn + 16 * h;
                        const double2 ep = b2[snh * 16 + c];  // E' of the node; BK0 = E', BK1 = omega_c E'
                        const cd bk0 = mk(ep.x, ep.y);
                        const cd bk1 = mk(fma(wcx, ep.x, -(wcy * ep.y)), fma(wcx, ep.y, wcy * ep.x));
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int p = rho + 4 * (r ^ pmask);
                            const double4 ra = *reinterpret_cast<const double4*>(a2 + tile_index(2 * snh, p));  // (Q1, Q0): 32 bytes
                            const cd q1 = mk(ra.x, ra.y), q0 = mk(ra.z, ra.w);
                            // fk = q1 bk1 + q0 bk0, one multiplication and three FMAs per component
                            const cd fk = mk(fma(q1.x, bk1.x, fma(-q1.y, bk1.y, fma(q0.x, bk0.x, -(q0.y * bk0.y)))),
                                             fma(q1.x, bk1.y, fma(q1.y, bk1.x, fma(q0.x, bk0.y, q0.y * bk0.x))));
                            if (h == 0) {
                                const cd fg = grat_node * fk;  // the Gauss rule's term of this node: (wg / wk) times the Kronrod one
                                pkx[r] = fk.x, pky[r] = fk.y, pgx[r] = fg.x, pgy[r] = fg.y;
                            } else {
                                pkx[r] += fk.x, pky[r] += fk.y;
                            }
                        }
                    }
                    auto mv_reduce = [&](const double (&v)[4]) -> double {
                        const double w0 = v[0] + dpp_mov<0x140>(v[3]), w1 = v[1] + dpp_mov<0x140>(v[2]);
                        double x = w0 + dpp_mov<0x141>(w1);
                        x = dpp_add_step<0xB1>(x);
                        return dpp_add_step<0x4E>(x);
                    };
                    mkx = mv_reduce(pkx), mky = mv_reduce(pky), mgx = mv_reduce(pgx), mgy = mv_reduce(pgy);
#ifdef EMME_DENSE_STAMPS
                    {
                        int dep;
                        asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(dep) : "v"(__double2loint(mkx + mgy)));
                        const unsigned long long tsx = __builtin_amdgcn_s_memtime();
                        cyc_sparse += tsx - ts1;
                        ts1x = tsx;
                    }
#endif
                    // the 16 elements of this column decide in ONE pass: lane (col = 4 q, rho) takes element
                    // (pair rho + 4 q, column c), whose state is slot q of the owner lane c + 16 rho; the verdicts
                    // go back to the owners as a ballot
                    const int owner = c + 16 * rho;
                    const int dq = col >> 2;              // the pair slot this lane's sums belong to
                    const bool decider = (col & 3) == 0;  // (one lane of the quad decides)
                    constexpr int DSTRIDE = 4;
                    const unsigned long long mbq = dq == 0 ? mb[0] : dq == 1 ? mb[1] : dq == 2 ? mb[2] : mb[3];
                    int qbad = 0;
                    bool sp = false;
                    if (decider && ((mbq >> owner) & 1ull)) {
                        // (column c is wave-uniform, and so is its moment; the decider lane's pair slot is rho + 4 dq)
                        const double cf = moment_factor(rho + 4 * dq, NM == 1 ? 0 : c % NM);
                        sp = NM == 1 ? decide(dq, owner, mkx, mky, mgx, mgy, qbad)
                                     : decide(dq, owner, mkx * cf, mky * cf, mgx * cf, mgy * cf, qbad);
                    }
                    const unsigned long long sb = __ballot(sp), bb = __ballot(qbad != 0);
                    if (col == c) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) split[r] = ((sb >> (DSTRIDE * r + 16 * rho)) & 1ull) != 0ull;
                        if ((bb >> (16 * rho)) & 0xffffull) bad = 1;
                    }
                }
            }
#ifdef EMME_DENSE_STAMPS
            {  // (make the stamp wait for the sums)
                int dep;
                asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(dep) : "v"(__double2loint(Kre[0] + Gim[3])));
            }
#endif
            STAMP(ts2);
            if (dense_round) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (__ballot(match[r]) == 0ull) continue;  // wave-uniform
                    if (match[r]) {
                        if (NM == 1) {
                            split[r] = decide(r, lane, Kre[r], Kim[r], Gre[r], Gim[r], bad);
                        } else {
                            const double cf = moment_factor(rho + 4 * r, mom);
                            split[r] = decide(r, lane, Kre[r] * cf, Kim[r] * cf, Gre[r] * cf, Gim[r] * cf, bad);
                        }
                    }
                }
            }
            if (__ballot(split[0] || split[1] || split[2] || split[3]) != 0ull) {
                if (n_next + 2 <= 64 * LW) {
                    const unsigned long long c0 = path << 1;
                    const unsigned int hi = ((unsigned)ccls << 30) | (unsigned)(c0 >> 32);
                    // (values and positions are wave-uniform: a lane-select writes lane n_next / n_next + 1)
                    // (n_next is even: both children land in the same word)
                    const int nl = n_next & 63;
                    if (LW > 1 && n_next >= 64) {
                        enext_lo[LW - 1] = lane == nl ? (unsigned)c0 : (lane == nl + 1 ? (unsigned)(c0 | 1ull) : enext_lo[LW - 1]);
                        enext_hi[LW - 1] = (lane == nl || lane == nl + 1) ? hi : enext_hi[LW - 1];
                    } else {
                        enext_lo[0] = lane == nl ? (unsigned)c0 : (lane == nl + 1 ? (unsigned)(c0 | 1ull) : enext_lo[0]);
                        enext_hi[0] = (lane == nl || lane == nl + 1) ? hi : enext_hi[0];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (split[r]) {
                            if (LW > 1 && n_next >= 64)
                                mnext[r][LW - 1] |= 3ull << nl;
                            else
                                mnext[r][0] |= 3ull << nl;
                        }
                    n_next += 2;
                } else {
                    // the next level's list is full: these integrals restart in the cooperative kernel
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (split[r]) {
                            defer(r, depth, ccls, path);
                            if (A.stats) atomicAdd(&A.stats[10], 1ull);
                            if (A.overflow) atomicAdd(&A.overflow[b], 1u);  // (the host widens this omega's lists next time)
                        }
                }
            }
#ifdef EMME_DENSE_STAMPS
            {
                int dep2;
                asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(dep2) : "v"((int)(unsigned)mnext[0][0]));
                STAMP(ts3);
                cyc_sel += ts1 - ts0;
                if (dense_round) cyc_dense += ts2 - ts1, cyc_dec += ts3 - ts2;
                else cyc_dec += ts3 - ts1x;  // (vector rounds: the decisions start where the last column's sums ended)
            }
#endif
        }
#pragma unroll
        for (int q = 0; q < LW; ++q) ecur_lo[q] = enext_lo[q], ecur_hi[q] = enext_hi[q];
        n_cur = n_next;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < LW; ++q) mcur[r][q] = mnext[r][q], mnext[r][q] = 0ull;
    }

    // ---- results (include/solver.h:448-455: mat(i,j) = -kappa W_ij dx, mirrored) ---------------------
    unsigned long long my_intervals = 0;
    const cd rdw = A.Mold ? rcp(mk(A.domega[b].x, A.domega[b].y)) : mk(0.0, 0.0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int pidx = tile * TILE_PAIRS + rho + 4 * r;
        if (has_w && pidx < A.npairs && !deferred[r]) {
            my_intervals += (unsigned long long)s_count[wave][r][lane];
            const ushort2 ij = A.pairs[pidx];
            const int i = ij.x, j = ij.y;
            const cd sm = mk(s_sumx[wave][r][lane], s_sumy[wave][r][lane]);
            cd kap = mk(P.pref * sm.y, -(P.pref * sm.x));  // -i pref sum, Parameters.cpp:182
            if (kappa_bad(kap)) bad = 1;
            if (NM == 1) {
                const cd v = (-(pair_weight(i, j, N) * P.dx)) * kap;
                store(i, j, v, rdw);
                store(j, i, v, rdw);
            } else {
                // blocks A (m = 0), B and its mirrors (m = 1), D (m = 2): include/solver.h:472-509
                const double de = A.tab[i] - A.tab[j], dg = A.tab[N + i] - A.tab[N + j];
                kap = kap + kappa_e(mom, P, de, dg, mk(omw.x, omw.y));
                if (mom == 0) {
                    const cd v = (-(pair_weight(i, j, N) * P.dx)) * kap;
                    store(i, j, v, rdw);
                    store(j, i, v, rdw);
                } else if (mom == 1) {
                    const cd v = P.dx * kap;
                    store(i, j + N, v, rdw);
                    store(j, i + N, -v, rdw);
                    store(i + N, j, -v, rdw);
                    store(j + N, i, v, rdw);
                } else {
                    const cd v = P.dx * kap;
                    store(i + N, j + N, v, rdw);
                    store(j + N, i + N, v, rdw);
                }
            }
        }
    }
    // interval count of this wave's 16 pairs per omega column: the four row lanes of a column, then the
    // workgroup's sum in LDS
    my_intervals += __shfl_xor(my_intervals, 16);
    my_intervals += __shfl_xor(my_intervals, 32);
    if (has_w) {
        if (my_intervals && rho == 0) atomicAdd(&s_iv[col], my_intervals);
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
        if (tile < 8192) atomicAdd(&A.stats[16 + tile], __builtin_amdgcn_s_memtime() - t_task);
    }
#endif
    if (lane == 0) {
        atomicAdd(&s_st[0], n_dense);
        atomicAdd(&s_st[1], n_sparse);
        atomicAdd(&s_st[2], n_cols);
        atomicAdd(&s_st[3], 1u);
    }
    __threadfence_block();
    int arrived = 0;
    if (lane == 0) arrived = atomicAdd(&s_arrived, 1) + 1;  // (LDS operations of a wave are performed in order)
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived == waves_here) {
        // the last wave of the workgroup: the sums go out (the lanes of row 0 hold the columns' items)
        __threadfence_block();
        if (lane < 16 && has_w && A.intervals && s_iv[lane] != 0ull) atomicAdd(&A.intervals[b], s_iv[lane]);
        if (A.stats && lane < 4) atomicAdd(&A.stats[lane], (unsigned long long)s_st[lane]);
    }
}

}  // namespace

size_t node_cache_bytes_tiled(long npairs, const NodeCacheGeom& g, int part, int gk_points) {
    const CacheGeom c = make_geom(g);
    const int ni = part < 0 ? c.ni_main() : c.ni_sub(part + 1);
    const size_t ntiles = (size_t)((npairs + TILE_PAIRS - 1) / TILE_PAIRS);
    return ntiles * (size_t)ni * tile_block_doubles(gk_points) * sizeof(double);
}

size_t btab_bytes(int nslots, int nchunks, int gk_points) {
    return (size_t)nslots * nchunks * btab_block_doubles(gk_points) * sizeof(double);
}

hipError_t launch_node_cache_tiled(const AssembleLaunch& L, const NodeCacheGeom& g, int part, double omi, void* recs,
                                   void* ttab, double* scale, hipStream_t stream, unsigned char* tile_poison, void* wtab) {
    TiledCacheArgs A;
    A.tile_poison = tile_poison;
    A.P = L.P;
    A.tab = L.tab;
    A.pairs = (const ushort2*)L.pairs;
    A.npairs = L.npairs;
    A.geom = make_geom(g);
    A.omi = omi;
    A.recs = (double*)recs;
    A.ttab = (double2*)ttab;
    A.wtab = (double2*)wtab;
    A.scale = scale;
    A.part = part;
    A.first = part < 0 ? 0 : A.geom.base[part + 1];
    A.count = part < 0 ? A.geom.ni_main() : A.geom.ni_sub(part + 1);
    if (A.count == 0) return hipSuccess;
    if (L.gk_points == 15)
        hipLaunchKernelGGL(k_node_cache_tiled<15>, dim3(256 * 32), dim3(256), 0, stream, A);
    else
        hipLaunchKernelGGL(k_node_cache_tiled<31>, dim3(256 * 32), dim3(256), 0, stream, A);
    return hipGetLastError();
}

hipError_t launch_btab(int gk_points, int nm, int nslots, const void* const ttab[2], const void* const wtab[2],
                       const double* omega, const int* act_idx, int n_act, const int* wmap, int nchunks, void* btab,
                       hipStream_t stream) {
    BtabArgs A;
    A.ttab[0] = (const double2*)ttab[0], A.ttab[1] = (const double2*)ttab[1];
    A.wtab[0] = wtab ? (const double2*)wtab[0] : nullptr, A.wtab[1] = wtab ? (const double2*)wtab[1] : nullptr;
    A.omega = (const double2*)omega;
    A.act_idx = act_idx;
    A.wmap = wmap;
    A.n_act = n_act;
    A.nchunks = nchunks;
    A.nslots = nslots;
    A.btab = (double*)btab;
    const long total = (long)nslots * tile_slots(gk_points) * n_act;
    long blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    if (blocks < 1) blocks = 1;
    if (gk_points == 15 && nm == 1)
        hipLaunchKernelGGL((k_btab<15, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, A);
    else if (gk_points == 31 && nm == 3)
        hipLaunchKernelGGL((k_btab<31, 3>), dim3((unsigned)blocks), dim3(256), 0, stream, A);
    else if (gk_points == 15 && nm == 3)
        hipLaunchKernelGGL((k_btab<15, 3>), dim3((unsigned)blocks), dim3(256), 0, stream, A);
    else if (gk_points == 31 && nm == 1)
        hipLaunchKernelGGL((k_btab<31, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, A);
    else
        return hipErrorNotSupported;
    return hipGetLastError();
}

hipError_t launch_assemble_dense(const AssembleLaunch& L, const NodeCacheGeom& g, const void* const recs[2],
                                 const void* const recs_ext[2][NODE_CACHE_MAX_SUB - 1], const double* scale,
                                 const void* btab, unsigned long long* worklist, unsigned int* worklist_count,
                                 unsigned long long* defer_info, const int* act_idx, int n_act,
                                 const void* chunks, int nchunks, unsigned long long* stats, hipStream_t stream,
                                 const unsigned char* const tile_poison[2], int n_wide, unsigned int* overflow) {
    DenseArgs A;
    A.tile_poison[0] = tile_poison ? tile_poison[0] : nullptr;
    A.tile_poison[1] = tile_poison ? tile_poison[1] : nullptr;
    A.P = L.P;
    A.tab = L.tab;
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
    A.chunks = (const int2*)chunks;
    A.nchunks = nchunks;
    A.omega = (const double2*)L.omega;
    A.M = (double2*)L.M;
    A.Mold = (const double2*)L.Mold;
    A.Mp = (double2*)L.Mp;
    A.domega = (const double2*)L.domega;
    A.intervals = L.intervals;
    A.status = L.status;
    A.stats = stats;
    A.dense_min_cols = L.dense_min_cols;
    A.skip_lost = L.skip_lost;
    A.chunk0 = 0;
    A.nch_launch = nchunks;
    A.tile_major = 0;
    A.overflow = overflow;
    const int ntiles = (L.npairs + TILE_PAIRS - 1) / TILE_PAIRS;
    const int ntg = (ntiles + 3) / 4;
    const int nm = L.P.dim == L.P.N ? 1 : 3;
    if (nm == 3) {  // electromagnetic (64-entry level lists only; tile-major, XCD-aware order)
        const dim3 grid((unsigned)((long)((ntg + 7) / 8) * 8 * A.nchunks));
        if (L.gk_points == 31)
            hipLaunchKernelGGL((k_assemble_dense<1, 31, 3>), grid, dim3(256), 0, stream, A);
        else
            hipLaunchKernelGGL((k_assemble_dense<1, 15, 3>), grid, dim3(256), 0, stream, A);
        return hipGetLastError();
    }
    if (L.gk_points == 31) {  // electrostatic GK31 (64-entry level lists only)
        hipLaunchKernelGGL((k_assemble_dense<1, 31, 1>), dim3((unsigned)((long)ntg * A.nchunks)), dim3(256), 0, stream, A);
        return hipGetLastError();
    }
    // the first n_wide chunks (omegas whose level lists overflowed last time) through the 128-entry build
    if (n_wide > A.nchunks) n_wide = A.nchunks;
    // (experiment, EMME_EXP_TILE_MAJOR = n: the electromagnetic task order for electrostatic launches of >= n chunks
    // too.  For ALL launches it is 57 against 35 ms per bench search: the late launches end on their expensive chunk)
    static const int tm_min = std::getenv("EMME_EXP_TILE_MAJOR") ? std::atoi(std::getenv("EMME_EXP_TILE_MAJOR")) : 0;
    A.tile_major = tm_min > 0 && A.nchunks - n_wide >= tm_min;
    const long ntg_l = A.tile_major ? (long)((ntg + 7) / 8) * 8 : ntg;
    if (n_wide > 0) {
        const int tm = A.tile_major;
        A.nch_launch = n_wide, A.tile_major = 0;
        hipLaunchKernelGGL((k_assemble_dense<2, 15, 1>), dim3((unsigned)((long)ntg * n_wide)), dim3(256), 0, stream, A);
        A.tile_major = tm;
    }
    if (A.nchunks > n_wide) {
        A.chunk0 = n_wide;
        A.nch_launch = A.nchunks - n_wide;
        hipLaunchKernelGGL((k_assemble_dense<1, 15, 1>), dim3((unsigned)(ntg_l * (A.nchunks - n_wide))), dim3(256), 0, stream, A);
    }
    return hipGetLastError();
}

}  // namespace emme
