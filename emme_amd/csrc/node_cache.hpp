// node_cache.hpp -- layout of the HBM cache of omega-independent node records, shared by the
// kernel that fills it / reads it (assemble_cached.hip) and by the list-mode fill kernel
// (assemble.hip), which uses cached records wherever they exist.
#pragma once
#include <hip/hip_runtime.h>

#include "launch.hpp"

namespace emme {
namespace {

struct NodeRec {  // 48 bytes per (item, interval, node); T = i t~ does not depend on the pair
    double2 A0, Q1, Q0;  // and lives in a small table shared by all items (L2-resident)
};

// Which intervals are cached.  Adaptive trees of this integrand are shallow almost everywhere;
// damped omegas force narrow, deep refinements, mostly towards t -> infinity (x -> pi/2).  The
// cache therefore holds the FULL tree down to depth `dfull` plus a list of full SUBTREES, each
// given by its root (depth rd, path rp) and the depth dd it reaches.  Subtree 0 is fixed (under
// the rightmost depth-5 node) and lives with the full tree in the main buffer; further
// subtrees are added by the host at run time around intervals that integrals were found to
// need (see emme_capi.hip), each in a buffer of its own.
struct CacheGeom {
    int dfull;
    int nsub;
    int dmax;  // deepest cached level (max of dfull and the dd's): nothing below it needs a look at the subtrees
    int rd[NODE_CACHE_MAX_SUB], dd[NODE_CACHE_MAX_SUB], base[NODE_CACHE_MAX_SUB];
    unsigned long long rp[NODE_CACHE_MAX_SUB];
    __host__ __device__ int ni_full() const { return (2 << dfull) - 1; }
    __host__ __device__ int ni_sub(int k) const { return (2 << (dd[k] - rd[k])) - 1; }
    __host__ __device__ int ni_main() const { return ni_full() + (nsub > 0 ? ni_sub(0) : 0); }
    __host__ __device__ int ni() const { return nsub > 0 ? base[nsub - 1] + ni_sub(nsub - 1) : ni_full(); }
    // record slot of interval (depth, path), or -1 if it is not cached; *which = -1 for the
    // main buffer, k-1 for the extension buffer of subtree k >= 1
    __device__ int slot(int depth, unsigned long long path, int& which) const {
        which = -1;
        if (depth <= dfull) return (1 << depth) - 1 + (int)path;
        if (depth > dmax) return -1;
        for (int k = 0; k < nsub; ++k) {
            if (depth <= dd[k] && depth >= rd[k]) {
                const int sd = depth - rd[k];
                if ((path >> sd) == rp[k]) {
                    which = k - 1;
                    return base[k] + (1 << sd) - 1 + (int)(path & ((1ull << sd) - 1ull));
                }
            }
        }
        return -1;
    }
    // inverse of slot() for the builder: part -1 = main buffer, part k >= 0 = subtree k+1
    __device__ void interval(int part, int rel, int& depth, unsigned long long& path) const {
        int k = part + 1;
        if (part < 0) {
            if (rel < ni_full()) {
                depth = 31 - __clz(rel + 1);
                path = (unsigned long long)(rel + 1) - (1ull << depth);
                return;
            }
            rel -= ni_full();
            k = 0;
        }
        const int sd = 31 - __clz(rel + 1);
        depth = rd[k] + sd;
        path = (rp[k] << sd) | ((unsigned long long)(rel + 1) - (1ull << sd));
    }
};


// ---- tiled layout (electrostatic GK15, dense fill: assemble_dense.hip) ---------------------------
// The records of TILE_PAIRS = 16 consecutive pairs of the pair list and one interval form one
// block of TILE_BLOCK doubles, laid out as the A operand of v_mfma_f64_16x16x4_f64 wants it:
//     Q[sn = 0..15][p = 0..15][which = 0, 1] as (re, im) pairs   (row k = 2 sn + which of the GEMM)
// k = 2 sn + which: sn = node slot in GAUSS-FIRST order (centre, +-x2, +-x4, +-x6, then +-x1, +-x3,
// +-x5, +-x7; sn 15 = zero padding), which = 0: exp(A0) Q1, 1: exp(A0) Q0 (the folded amplitudes),
// so that rows 0..13 are the embedded Gauss rule's and K = Q . BK, G = Q[0:16] . BG are plain
// complex GEMMs against per-launch tables of weighted phases (k_btab).  Re A0 is not kept: neither the
// dense fill nor the cooperative kernel applies the safe_exp clamp to tiled records (the clamped tails are
// <= 4e-14 absolute, assemble_dense.hip), and records that are not finite are zeroed when they are built.
// GK31 (round 3: the electromagnetic dense fill) has 32 node slots -- 15 Gauss nodes first, the 16 Kronrod-only
// ones behind them, slot 31 padding -- in the same [sn][p][which] order: 16 KB per (tile, interval).
constexpr int TILE_PAIRS = 16;
constexpr int TILE_BLOCK = 2 * 32 * 16;  // doubles: 8 KB per (tile, interval), GK15
__host__ __device__ constexpr int tile_slots(int pts) { return pts == 15 ? 16 : 32; }
__host__ __device__ constexpr int tile_block_doubles(int pts) { return 4 * tile_slots(pts) * 16; }
__host__ __device__ constexpr int btab_block_doubles(int pts) { return 2 * tile_slots(pts) * 16; }
// node slot of lane r of a lane group (gk_lane<PTS>(r): centre, +x_1 .. +x_{H-1}, -x_1 .. -x_{H-1}, padding)
template <int PTS>
__host__ __device__ inline int slotnode_of_lane_t(int lane) {
    constexpr int H = (PTS + 1) / 2;
    if (lane >= PTS) return tile_slots(PTS) - 1;
    const int q = lane < H ? lane : lane - (H - 1), neg = lane >= H ? 1 : 0;
    if (q == 0) return 0;
    return (q & 1) ? (H - 1) + (q - 1) + neg : q - 1 + neg;
}
__host__ __device__ inline int slotnode_of_lane(int lane) { return slotnode_of_lane_t<15>(lane); }
// per-launch table of weighted phases, one block per (interval slot, omega chunk of 16):
//     E'[sn][column omega] = wk_sn exp(T_sn omega)   as (re, im) pairs, 4 KB.
// The GEMM's B rows of node sn are 2 sn: omega E' and 2 sn + 1: E'; the fill forms the first from the second
// in registers (the table is read by every round of every tile: half the bytes of keeping both rows).  The
// embedded Gauss rule needs no table of its own either: its row k is (wg / wk)_k times BK's, a per-row
// constant that the fill applies to the A operand.
constexpr int BTAB_BLOCK = 2 * 16 * 16;  // doubles: 4 KB
// element (row k, column / pair j) of a record or phase block, in (re, im) pairs.  The two rows of a node sit
// side by side: the vector rounds and the cooperative kernel read a node's (Q1, Q0) as ONE 32-byte piece,
// and the 64 lanes of an MFMA operand load (k = 4 ks + (lane >> 4), j = lane & 15) still cover 1 KB exactly.
__host__ __device__ inline int tile_index(int k, int j) { return ((((k >> 1) << 4) + j) << 1) | (k & 1); }

inline CacheGeom make_geom(const NodeCacheGeom& g) {
    CacheGeom c;
    c.dfull = g.dfull;
    c.nsub = g.nsub;
    c.dmax = g.dfull;
    for (int k = 0; k < g.nsub && k < NODE_CACHE_MAX_SUB; ++k) c.dmax = g.dd[k] > c.dmax ? g.dd[k] : c.dmax;
    int base = c.ni_full();
    for (int k = 0; k < NODE_CACHE_MAX_SUB; ++k) {
        c.rd[k] = k < g.nsub ? g.rd[k] : 0;
        c.dd[k] = k < g.nsub ? g.dd[k] : 0;
        c.rp[k] = k < g.nsub ? g.rp[k] : 0;
        c.base[k] = base;
        if (k < g.nsub) base += c.ni_sub(k);
    }
    return c;
}


}  // namespace
}  // namespace emme
