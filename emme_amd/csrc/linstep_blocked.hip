// linstep_blocked.hip -- blocked form of the Newton linear step tr(M^-1 M') (see linstep.hip
// for the unblocked reference version and the algorithmic notes).
//
// One workgroup (1024 threads) per batch item -- or several, see "several workgroups per matrix"
// below; n <= ~560 with the whole L21 panel in LDS, up to 1024 in chunks (CHUNK) -- block size NB = 16:
//   panel    the NB current columns of every remaining row live in REGISTERS of the row's
//            owner thread; partial pivoting = block argmax per column, the winner publishes
//            its row through LDS; no row is ever moved in memory (a row map keeps the pivot
//            order).  Multipliers go to global memory only for helper workgroups.
//   trailing T1: each lane owns one column J of the augmented matrix [A | B], loads the NB
//            pivot-row entries of that column, finishes them with the NB x NB unit-lower block
//            (forward substitution in registers) and stores them: rows of U / of L^-1 P B.
//            T2: the rows below the block, X -= L21 * U12, as 16 x 16 complex tiles on the
//            FP64 matrix cores (mfma_update below; L21 in LDS in pivot order): the trailing
//            matrix is read and written once per NB columns.
//   back     the truncated back substitution (X(c,c) needs rows c..n-1 of column c only) in
//            the same shape: per block of NB rows a register triangular solve per column,
//            then the rows above are updated by the same MFMA tile routine with the U column
//            block in LDS (tiles right of the diagonal are skipped).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "emme_device.hpp"
#include "launch.hpp"

namespace emme {

namespace {

constexpr int BT = 1024;     // threads per workgroup
constexpr int BW = BT / 64;  // waves
constexpr int NB = 16;
static_assert(BW == 16, "the cross-wave pivot reduction reads one candidate per lane of a 16-lane row");
constexpr int LS = NB + 1;  // row stride of the L21 panel in LDS (double2 units)
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ cd ldg(const double2* p) {
    const double2 v = *p;
    return mk(v.x, v.y);
}
__device__ __forceinline__ void stg(double2* p, cd v) { *p = make_double2(v.x, v.y); }

// max over the 16 lanes of a DPP row, result in every lane of the row (no LDS traffic): quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror, row_mirror; 32-bit keys (the pivot search): one v_max_u32 with a DPP
// operand per step
template <int CTRL>
__device__ __forceinline__ unsigned int dpp_max32_step(unsigned int v) {
    const unsigned int o = (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned int row16_max32(unsigned int v) {
    v = dpp_max32_step<0xB1>(v);
    v = dpp_max32_step<0x4E>(v);
    v = dpp_max32_step<0x141>(v);
    v = dpp_max32_step<0x140>(v);
    return v;
}
__device__ __forceinline__ unsigned int wave_max32(unsigned int v) {
    v = row16_max32(v);
    unsigned int m = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned int o = (unsigned)__builtin_amdgcn_readlane((int)v, 16 * r);
        m = o > m ? o : m;
    }
    return m;
}
// acc - f p with four fused multiply-adds (the panels' rank-1 updates; the trailing updates accumulate fused
// products on the matrix cores anyway)
__device__ __forceinline__ cd cmsub(cd acc, cd f, double2 p) {
    return cd{__builtin_fma(f.y, p.y, __builtin_fma(-f.x, p.x, acc.x)), __builtin_fma(-f.y, p.x, __builtin_fma(-f.x, p.y, acc.y))};
}
#ifndef EMME_LU_SOLVE_BATCH
#define EMME_LU_SOLVE_BATCH 4
#endif
constexpr int SOLVE_BATCH = EMME_LU_SOLVE_BATCH;  // (a power of two)
// Unit-lower solve with L11 (T1: the NB pivot rows of one column, in registers).  The uniform test per row
// also keeps the LDS reads of one row together and the rows apart (all 120 hoisted do not fit in registers).
__device__ __forceinline__ void t1_solve(cd (&u)[NB], const double2* L11, int nbk) {
#pragma unroll
    for (int kk = 1; kk < NB; ++kk) {
        if (kk < nbk) {
#pragma unroll
            for (int c = 0; c < kk; ++c) u[kk] = cmsub(u[kk], u[c], L11[kk * NB + c]);
        }
    }
}
// Upper-triangular solve of one column of the truncated back substitution against the LDS copy of U11
// (reciprocal diagonal; zero outside the block's nbk x nbk triangle, so that a short last block needs no
// test: its extra rows stay zero).  Rows below the column index c are not needed for X(c,c) and are zeroed
// so that they drop out of every later sum; X(c,c) itself goes to diag.
__device__ __forceinline__ void back_solve(cd (&x)[NB], const double2* L11, int k0, int c, bool okc, double2* diag) {
#pragma unroll
    for (int kk = NB - 1; kk >= 0; --kk) {
        cd sx = x[kk];
#pragma unroll
        for (int qq = kk + 1; qq < NB; ++qq) {
            sx = cmsub(sx, x[qq], L11[kk * NB + qq]);
            if (((qq - kk) & (SOLVE_BATCH - 1)) == 0) __builtin_amdgcn_sched_barrier(0);
        }
        const double2 rd = L11[kk * NB + kk];
        sx = sx * mk(rd.x, rd.y);
        __builtin_amdgcn_sched_barrier(0);
        x[kk] = (k0 + kk >= c) ? sx : mk(0.0, 0.0);
        if (okc && k0 + kk == c) diag[c] = make_double2(sx.x, sx.y);
    }
}
// pivot candidate of a row: the high word of |x|^2 (sign 0, 11 exponent and 10 mantissa bits: the modulus to
// 2^-11) with its 10 lowest bits replaced by 1023 - slot, so that a plain 32-bit maximum picks the largest
// modulus and, among equals, the first row.  0 = not a candidate / exactly zero; an |x|^2 below 2^-1012 counts
// as zero, a NaN or infinity as the largest (and is then reported as a singular column).
__device__ __forceinline__ unsigned int pivot_key(double n2, int slot) {
    const unsigned int hi = (unsigned int)__double2hiint(n2);
    return (hi & ~1023u) ? ((hi & ~1023u) | (unsigned int)(1023 - slot)) : 0u;
}
// the candidate that won: usable as a pivot? (positive, finite)
__device__ __forceinline__ bool pivot_ok(unsigned int key) { return key != 0u && key < 0x7ff00000u; }

struct BlkShared {
    unsigned int s_key[2][BW];  // per-wave pivot candidates (columns alternate: one barrier per column)
    double s_val[BW];
    int s_idx[BW];
    int info;
    int go;
};

// C(rows, cols) -= A(rows, 0:nbk) * B(0:nbk, cols) for complex matrices on the FP64 matrix cores:
// one 16 x 16 tile per wave and trip, v_mfma_f64_16x16x4_f64, four k-steps, and per k-step four
// products for the complex multiply-subtract
//     Cre += (-Are) Bre + Aim Bim,   Cim += (-Are) Bim + (-Aim) Bre.
// Operand maps (one f64 per lane): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15];
// C/D: column lane & 15, row (lane >> 4) + 4 r.
//   C rows : logical rows crow0 .. crow0 + nrows - 1 (physical row = rowmap[.])
//   A      : LDS `panel`, row (logical row - crow0), rows LS entries apart
//   B rows : logical rows brow0 .. brow0 + nbk - 1, read from memory once per column tile
//   cols   : columns col0 .. col0 + ncols - 1 of the augmented matrix [a | bb]
//   LOWER  : only tiles that contain an entry with (row - crow0) + 16 rt_shift >= (col - col0) are touched
// The inner loop is branch-free: out-of-range rows / columns are clamped to valid addresses for
// the loads (their operands are zeroed, their results never stored), so a tile is 4 LDS index
// reads, 4 LDS operand reads, 4 loads, 16 MFMAs, 4 stores.
template <bool LOWER>
__device__ __forceinline__ void mfma_update(int n, double2* a, double2* bb, const int* rowmap,
                                            const double2* panel, int crow0, int nrows, int brow0,
                                            int nbk, int col0, int ncols, int wave, int lane,
                                            int rt_shift = 0) {
    const int nrt = (nrows + 15) >> 4, nct = (ncols + 15) >> 4;
    const int i16 = lane & 15, kq = lane >> 4;
    int cur_ct = -1;
    double bre[4], bim[4];
    double2* colbase = a;  // &[a | bb](row 0, this lane's column)
    bool okc = false;
    for (int tile = wave; tile < nrt * nct; tile += BW) {
        const int ct = tile / nrt, rt = tile - ct * nrt;
        if (LOWER && ct > rt + rt_shift) continue;  // wave-uniform (rt_shift: row tiles the C rows start below the columns)
        if (ct != cur_ct) {              // wave-uniform
            cur_ct = ct;
            const int J = col0 + ct * 16 + i16;
            okc = J < col0 + ncols;
            const int Jc = okc ? J : col0 + ncols - 1;
            colbase = Jc < n ? a + Jc : bb + (Jc - n);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int k = 4 * ks + kq;
                const int kc = k < nbk ? k : nbk - 1;
                const double2 u = colbase[(size_t)rowmap[brow0 + kc] * n];
                const bool ok = okc && k < nbk;
                bre[ks] = ok ? u.x : 0.0, bim[ks] = ok ? u.y : 0.0;
            }
        }
        // this lane's four C rows (C/D map: row kq + 4 r)
        double2* px[4];
        bool okr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = rt * 16 + kq + 4 * r;
            okr[r] = okc && rr < nrows;
            const int rc = rr < nrows ? rr : nrows - 1;
            px[r] = colbase + (size_t)rowmap[crow0 + rc] * n;
        }
        // (all four loads first, into registers of their own: unpacked one by one into the accumulator
        // vectors they shared temporaries, and each load waited for the one before it)
        d4 cre, cim;
        double2 cx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) cx[r] = *px[r];
        asm volatile("" : "+v"(cx[0].x), "+v"(cx[0].y), "+v"(cx[1].x), "+v"(cx[1].y), "+v"(cx[2].x), "+v"(cx[2].y), "+v"(cx[3].x), "+v"(cx[3].y));
#pragma unroll
        for (int r = 0; r < 4; ++r) cre[r] = cx[r].x, cim[r] = cx[r].y;
        const int ri = rt * 16 + i16;
        const int ric = ri < nrows ? ri : nrows - 1;
        double nare[4], aim[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = 4 * ks + kq;  // < NB <= LS: inside the padded LDS row
            const double2 l = panel[ric * LS + k];
            const bool ok = ri < nrows && k < nbk;
            nare[ks] = ok ? -l.x : 0.0, aim[ks] = ok ? l.y : 0.0;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            cre = __builtin_amdgcn_mfma_f64_16x16x4f64(nare[ks], bre[ks], cre, 0, 0, 0);
            cim = __builtin_amdgcn_mfma_f64_16x16x4f64(nare[ks], bim[ks], cim, 0, 0, 0);
            cre = __builtin_amdgcn_mfma_f64_16x16x4f64(aim[ks], bim[ks], cre, 0, 0, 0);
            cim = __builtin_amdgcn_mfma_f64_16x16x4f64(-aim[ks], bre[ks], cim, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (okr[r]) *px[r] = make_double2(cre[r], cim[r]);
    }
}


// T1 of the trailing update: finish the NB pivot rows of columns Jlo .. Jhi-1 of [a | bb] (unit-
// lower solve with L11, one column per lane in registers) and store them: they are rows of U
// and of L^-1 P B.
__device__ __forceinline__ void pivot_rows_update(int n, double2* a, double2* bb, const int* rowmap,
                                                  const double2* L11, int k0, int nbk, int Jlo,
                                                  int Jhi, int wave, int lane) {
    const int nchunks = (Jhi - Jlo + 63) / 64;
    for (int q = wave; q < nchunks; q += BW) {
        const int J = Jlo + q * 64 + lane;
        if (J < Jhi) {
            double2* col = J < n ? a + J : bb + (J - n);
            cd u[NB];
#pragma unroll
            for (int kk = 0; kk < NB; ++kk)
                u[kk] = kk < nbk ? ldg(col + (size_t)rowmap[k0 + kk] * n) : mk(0.0, 0.0);
            t1_solve(u, L11, nbk);
#pragma unroll
            for (int kk = 0; kk < NB; ++kk)
                if (kk < nbk) stg(col + (size_t)rowmap[k0 + kk] * n, u[kk]);
        }
    }
}

// Delayed ("grouped") trailing update: C(rows, cols) -= L(rows, k0 : k0+nk) * U(k0 : k0+nk, cols) for a
// GROUP of up to GP = 4 panels (nk <= 64) in ONE pass over C -- the same MFMA sequence per element as
// four mfma_update passes (same bits), a quarter of the reads and writes of the trailing matrix, which
// is what a launch of 128 matrices of order 512 is bound by (1 GB of [M | M'], streamed from HBM per
// panel: 34 GB per launch).  The multipliers are read from A (in-place LU layout) into LDS in chunks
// of `rc` rows, GLS entries apart; the U rows (B operand) come from memory per tile and k-block.
#ifndef EMME_LU_GP
#define EMME_LU_GP 4
#endif
constexpr int GP = EMME_LU_GP, GK = GP * NB, GLS = GK + 1;
// LOWER (the truncated back substitution): only tiles on or left of the diagonal of the (rows, cols) range
// are touched, and the group's panels are applied last to first, the order of the per-panel sweep.
template <bool LOWER = false>
__device__ __forceinline__ void mfma_update_grouped(int n, double2* a, double2* bb, const int* rowmap,
                                                    double2* buf, int rc, int crow0, int nrows, int k0,
                                                    int nk, int col0, int ncols, int tid, int wave, int lane) {
    const int i16 = lane & 15, kq = lane >> 4;
    const int nct = (ncols + 15) >> 4;
    for (int r0 = 0; r0 < nrows; r0 += rc) {
        const int nr = min(rc, nrows - r0);
        for (int e = tid; e < nr * GK; e += BT) {
            const int r = e / GK, c = e % GK;
            buf[r * GLS + c] = c < nk ? a[(size_t)rowmap[crow0 + r0 + r] * n + k0 + c] : make_double2(0.0, 0.0);
        }
        __syncthreads();
        const int nrt = (nr + 15) >> 4;
        if (!LOWER) {
        // forward sweep: TWO row tiles per wave and trip: they share the loads of the B operand (the U rows, 4 loads per 16
        // k-steps), which is what the pass waits for -- per element the same MFMA sequence as one tile at a time
        const int nrp = (nrt + 1) >> 1;
        for (int tile = wave; tile < nrp * nct; tile += BW) {
            const int ct = tile / nrp, rp = tile - ct * nrp;
            const int rt0 = 2 * rp, rt1 = 2 * rp + 1;
            const bool on0 = !(LOWER && ct > rt0 + (r0 >> 4));                 // wave-uniform
            const bool on1 = rt1 < nrt && !(LOWER && ct > rt1 + (r0 >> 4));    // wave-uniform
            if (!on0 && !on1) continue;
            const int J = col0 + ct * 16 + i16;
            const bool okc = J < col0 + ncols;
            const int Jc = okc ? J : col0 + ncols - 1;
            double2* colbase = Jc < n ? a + Jc : bb + (Jc - n);
            int poff[2][4];  // row offsets of the C entries (elements: n^2 < 2^31)
            bool okr[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = (2 * rp + h) * 16 + kq + 4 * r;
                    okr[h][r] = (h ? on1 : on0) && okc && rr < nr;
                    const int rcl = rr < nr ? rr : nr - 1;
                    poff[h][r] = rowmap[crow0 + r0 + rcl] * n;
                }
            }
            d4 cre[2], cim[2];
            double2 cx[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) cx[h][r] = colbase[poff[h][r]];
            asm volatile("" : "+v"(cx[0][0].x), "+v"(cx[0][0].y), "+v"(cx[0][1].x), "+v"(cx[0][1].y), "+v"(cx[0][2].x), "+v"(cx[0][2].y), "+v"(cx[0][3].x), "+v"(cx[0][3].y));
            asm volatile("" : "+v"(cx[1][0].x), "+v"(cx[1][0].y), "+v"(cx[1][1].x), "+v"(cx[1][1].y), "+v"(cx[1][2].x), "+v"(cx[1][2].y), "+v"(cx[1][3].x), "+v"(cx[1][3].y));
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) cre[h][r] = cx[h][r].x, cim[h][r] = cx[h][r].y;
            const int ri0 = rt0 * 16 + i16, ri1 = rt1 * 16 + i16;
            const int ric0 = ri0 < nr ? ri0 : nr - 1, ric1 = ri1 < nr ? ri1 : nr - 1;
            const int nq = (nk + NB - 1) / NB;
            for (int qq = 0; qq < nq; ++qq) {
                const int q = LOWER ? nq - 1 - qq : qq;
                double nare0[4], aim0[4], nare1[4], aim1[4], bre[4], bim[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int k = q * NB + 4 * ks + kq;
                    const int kc = k < nk ? k : nk - 1;
                    const double2 u = colbase[(size_t)rowmap[k0 + kc] * n];
                    const bool okb = okc && k < nk;
                    bre[ks] = okb ? u.x : 0.0, bim[ks] = okb ? u.y : 0.0;
                    const double2 l0 = buf[ric0 * GLS + kc], l1 = buf[ric1 * GLS + kc];
                    const bool oka0 = ri0 < nr && k < nk, oka1 = ri1 < nr && k < nk;
                    nare0[ks] = oka0 ? -l0.x : 0.0, aim0[ks] = oka0 ? l0.y : 0.0;
                    nare1[ks] = oka1 ? -l1.x : 0.0, aim1[ks] = oka1 ? l1.y : 0.0;
                }
                if (on0) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        cre[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(nare0[ks], bre[ks], cre[0], 0, 0, 0);
                        cim[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(nare0[ks], bim[ks], cim[0], 0, 0, 0);
                        cre[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aim0[ks], bim[ks], cre[0], 0, 0, 0);
                        cim[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-aim0[ks], bre[ks], cim[0], 0, 0, 0);
                    }
                }
                if (on1) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        cre[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(nare1[ks], bre[ks], cre[1], 0, 0, 0);
                        cim[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(nare1[ks], bim[ks], cim[1], 0, 0, 0);
                        cre[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(aim1[ks], bim[ks], cre[1], 0, 0, 0);
                        cim[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-aim1[ks], bre[ks], cim[1], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (okr[h][r]) colbase[poff[h][r]] = make_double2(cre[h][r], cim[h][r]);
        }
        } else
        for (int tile = wave; tile < nrt * nct; tile += BW) {
            const int ct = tile / nrt, rt = tile - ct * nrt;
            if (LOWER && ct > rt + (r0 >> 4)) continue;  // wave-uniform
            const int J = col0 + ct * 16 + i16;
            const bool okc = J < col0 + ncols;
            const int Jc = okc ? J : col0 + ncols - 1;
            double2* colbase = Jc < n ? a + Jc : bb + (Jc - n);
            double2* px[4];
            bool okr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = rt * 16 + kq + 4 * r;
                okr[r] = okc && rr < nr;
                const int rcl = rr < nr ? rr : nr - 1;
                px[r] = colbase + (size_t)rowmap[crow0 + r0 + rcl] * n;
            }
            d4 cre, cim;
            double2 cx[4];  // (all four loads first: see mfma_update)
#pragma unroll
            for (int r = 0; r < 4; ++r) cx[r] = *px[r];
            asm volatile("" : "+v"(cx[0].x), "+v"(cx[0].y), "+v"(cx[1].x), "+v"(cx[1].y), "+v"(cx[2].x), "+v"(cx[2].y), "+v"(cx[3].x), "+v"(cx[3].y));
#pragma unroll
            for (int r = 0; r < 4; ++r) cre[r] = cx[r].x, cim[r] = cx[r].y;
            const int ri = rt * 16 + i16;
            const int ric = ri < nr ? ri : nr - 1;
            const int nq = (nk + NB - 1) / NB;
            for (int qq = 0; qq < nq; ++qq) {
                const int q = LOWER ? nq - 1 - qq : qq;
                double nare[4], aim[4], bre[4], bim[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int k = q * NB + 4 * ks + kq;
                    const int kc = k < nk ? k : nk - 1;
                    const double2 u = colbase[(size_t)rowmap[k0 + kc] * n];
                    const bool okb = okc && k < nk;
                    bre[ks] = okb ? u.x : 0.0, bim[ks] = okb ? u.y : 0.0;
                    const double2 l = buf[ric * GLS + kc];
                    const bool oka = ri < nr && k < nk;
                    nare[ks] = oka ? -l.x : 0.0, aim[ks] = oka ? l.y : 0.0;
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    cre = __builtin_amdgcn_mfma_f64_16x16x4f64(nare[ks], bre[ks], cre, 0, 0, 0);
                    cim = __builtin_amdgcn_mfma_f64_16x16x4f64(nare[ks], bim[ks], cim, 0, 0, 0);
                    cre = __builtin_amdgcn_mfma_f64_16x16x4f64(aim[ks], bim[ks], cre, 0, 0, 0);
                    cim = __builtin_amdgcn_mfma_f64_16x16x4f64(-aim[ks], bre[ks], cim, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (okr[r]) *px[r] = make_double2(cre[r], cim[r]);
        }
        __syncthreads();
    }
}

#ifdef EMME_LU_STAMPS  // diagnostic build: where the roles of a matrix spend their time (never in the product build)
__device__ unsigned long long g_lu_stamps[32];
#define LU_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define LU_ADD(slot, t0, t1) do { if (threadIdx.x == 0) atomicAdd(&g_lu_stamps[(slot)], (t1) - (t0)); } while (0)
#else
#define LU_T(v)
#define LU_ADD(slot, t0, t1)
#endif
// Arguments of a __noinline__ device function arrive in vector registers: the compiler must take them for
// per-lane values and turns every branch and address that depends on them into exec-mask code (380
// s_and_saveexec in factor_panel alone).  They are wave-uniform here: say so.
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* uniform(T* p) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    // ... and they point to GLOBAL memory (rebuilt as an address-space-1 pointer): otherwise every access
    // through them is a FLAT one, which counts in lgkmcnt as well as vmcnt -- each wait for an LDS read then
    // also waits for every load in flight
    typedef __attribute__((address_space(1))) T GT;
    GT* g = (GT*)(((unsigned long long)hi << 32) | lo);
    return (T*)g;
}

// The panels of a group (first column g0, ng <= GK columns) are factored and published and their row
// order is in the row map: apply them to columns Jlo .. Jhi-1 of [A | B] -- per panel the pivot rows
// (T1) and the group's later pivot rows, then ONE pass over the rows below the group with all of the
// group's multipliers.  Not inlined: the panel loop of the kernel lives on a 128-register budget of its
// own (inlined, this costs it 230 more spilled registers).  The LDS areas are the kernel's (dynamic LDS:
// rowmap | physrow | pivof | L11 | NB spare entries | panel).
__device__ __noinline__ void apply_group(int n_, double2* a_, double2* bb_, int g0_, int ng_, int Jlo_, int Jhi_) {
    const int n = uniform(n_), g0 = uniform(g0_), ng = uniform(ng_), Jlo = uniform(Jlo_), Jhi = uniform(Jhi_);
    double2* a = uniform(a_);
    double2* bb = uniform(bb_);
    extern __shared__ double2 lds2[];
    if (Jhi <= Jlo) return;  // uniform
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* rowmap = reinterpret_cast<int*>(lds2);
    double2* L11 = lds2 + (3 * n * (int)sizeof(int) + 15) / 16;
    double2* panel = L11 + NB * NB + NB;
    for (int k0 = g0; k0 < g0 + ng; k0 += NB) {
        const int nbk = min(NB, g0 + ng - k0);
        const int nin = g0 + ng - (k0 + nbk);  // pivot rows of the group's later panels
        LU_T(ta0);
        for (int e = tid; e < NB * NB; e += BT) {
            const int kk = e / NB, c = e % NB;
            L11[e] = (kk < nbk && c < kk) ? a[(size_t)rowmap[k0 + kk] * n + k0 + c] : make_double2(0.0, 0.0);
        }
        for (int e = tid; e < nin * NB; e += BT) {
            const int r = e / NB, c = e % NB;
            panel[r * LS + c] = c < nbk ? a[(size_t)rowmap[k0 + nbk + r] * n + k0 + c] : make_double2(0.0, 0.0);
        }
        __syncthreads();
        LU_T(ta1);
        LU_ADD(12, ta0, ta1);
        pivot_rows_update(n, a, bb, rowmap, L11, k0, nbk, Jlo, Jhi, wave, lane);
        __syncthreads();
        LU_T(ta2);
        LU_ADD(13, ta1, ta2);
        if (nin > 0) mfma_update<false>(n, a, bb, rowmap, panel, k0 + nbk, nin, k0, nbk, Jlo, Jhi - Jlo, wave, lane);
        __syncthreads();
        LU_T(ta3);
        LU_ADD(14, ta2, ta3);
    }
    const int below = n - (g0 + ng);
    if (below > 0) {
        LU_T(ta4);
        const int rc = max(16, (n * LS / GLS) / 16 * 16);  // rows of multipliers the panel area holds
        mfma_update_grouped<false>(n, a, bb, rowmap, panel, rc, g0 + ng, below, g0, ng, Jlo, Jhi - Jlo, tid, wave, lane);
        LU_T(ta5);
        LU_ADD(15, ta4, ta5);
    }
}

// ---- several workgroups per matrix ---------------------------------------------------------
// With fewer matrices than compute units a launch of one workgroup per matrix leaves CUs idle, so
// a matrix can be given nwg = 1 + S workgroups ("roles"; role = blockIdx.x / nitems, so that every
// role-0 workgroup is dispatched before any other):
//   forward  role 0 factors A (panel + trailing A columns) and publishes, per panel, the
//            multipliers (in A below the diagonal), the row order (a snapshot of the row map) and
//            a counter; roles 1..S each own a range of B's columns and apply the published panels
//            to it (producer -> consumer only: role 0 never waits in this phase).
//            Look-ahead (4 or more workgroups): role 0 applies panel j only to block j+1, the
//            next panel's columns; roles 1..na (na = 1, or 2 from 6 workgroups on: columns left /
//            right of 0.65 n) carry the rest of A's trailing columns and count their finished
//            steps; role 0 waits for that block's helper to have finished step j-1 before it
//            puts panel j there.  The other roles share B.
//   back     when U and L^-1 P B are complete every role takes a range of columns of the
//            truncated back substitution, cut so that the ranges cost the same.
//   trace    X(c,c) goes to a scratch vector and the last workgroup to arrive adds it up in a
//            fixed order: the result does not depend on nwg, bit for bit.
// Cross-workgroup hand-over: workgroup barrier, then ONE thread's agent-scope release store /
// acquire load (they write back / invalidate this XCD's L2: the per-XCD L2s are not coherent
// with each other otherwise), then a barrier again on the consumer side.
// Waits are bounded: a time-out retires the matrix with info = EMME_EDEVICE instead of hanging.
struct SplitCtl {
    int nwg;
    int na;            // look-ahead: workgroups that carry A's trailing columns (0: none, 1 .. 5)
    int a_cut[4];      // first column of A-helper 2, 3, .. (equal shares of the trailing work)
    int spin_limit;    // polls before a hand-over wait gives up (EMME_LU_SPIN_LIMIT; tests use 1)
    int nitems;        // matrices of this launch: role = blockIdx.x / nitems
    const int* items;  // their indices in the batch (null: 0 .. nitems-1).  A dense list, so that
                       // the workgroups spread evenly over the XCDs (block i runs on XCD i % 8)
    int* flags;     // [nbatch][8]: panels published | helpers finished | arrivals | steps done by A-helper 1 .. 5
    int* rowmaps;   // [nbatch][nblk][n]  (nwg > 1 only)
    double2* diag;  // [nbatch][n]
};
constexpr int ABORT = 1 << 30;
constexpr int SPIN_LIMIT = 16000000;  // about 4 s
constexpr int INFO_TIMEOUT = -3;  // EMME_EDEVICE

__device__ __forceinline__ int spin_ge(int* p, int v, int limit) {
    // relaxed polling (an acquire load would invalidate the XCD's L2 on every trip), one
    // acquire fence once the value is there
    for (int it = 0; it < limit; ++it) {
        const int x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (x >= v) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            return x;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    return -1;
}

// CHUNK (with SPLIT only): matrices whose L21 panel does not fit in LDS (560 < n <= 1024).  The
// multipliers then go from the panel registers straight to global memory and every trailing /
// back-substitution update walks the rows in chunks of RC, fetching that chunk's panel rows.
constexpr int RC = 512;
template <bool SPLIT, bool CHUNK = false>
__global__ __launch_bounds__(BT) void k_trace_solve_blocked(int n, int nbatch, double2* A, double2* B,
                                                            const int* active, double2* tr_out,
                                                            int* info_out, SplitCtl ctl) {
    // dynamic LDS: rowmap[n] | physrow[n] | pivof[n] (ints) | L11[NB][NB] | NB spare entries |
    //              panel[n][NB] (L21 in the forward phase, U column block in the back phase; during a
    //              panel's column loop its first 8 KB hold the waves' candidate pivot rows)
    extern __shared__ double2 lds2[];
    __shared__ BlkShared sh;

    static_assert(SPLIT || !CHUNK, "the chunked panel needs the multipliers in global memory");
    const int role = SPLIT ? blockIdx.x / ctl.nitems : 0;
    int b = blockIdx.x - role * ctl.nitems;
    if (SPLIT && ctl.items) b = ctl.items[b];
    if (active && active[b] == 0) return;
    const int nwg = SPLIT ? ctl.nwg : 1, S = nwg - 1;
    // look-ahead (4 or more workgroups): role 0 applies a panel only to the NEXT panel's columns,
    // roles 1..na to the rest of A (column ranges of equal work, cut by the host), so that A's
    // trailing update leaves the factoring critical path
    const int na = SPLIT ? ctl.na : 0;
    const bool la = na > 0;
    int* flag_pub = ctl.flags + 8 * b;  // the hand-over state is touched for SPLIT only
    int* snap = ctl.rowmaps + (size_t)b * ((n + NB - 1) / NB) * n;
    double2* a = A + (size_t)b * n * n;
    double2* bb = B + (size_t)b * n * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    int* rowmap = reinterpret_cast<int*>(lds2);  // logical position -> physical row
    double2* L11 = lds2 + (3 * n * (int)sizeof(int) + 15) / 16;
    double2* panel = L11 + NB * NB + NB;
    double2* cand = panel;  // [2][BW][NB] candidate pivot rows of the panel loop (the area is free during it)

    for (int r = tid; r < n; r += BT) rowmap[r] = r;
    if (tid == 0) sh.info = 0;
    __syncthreads();

    // column J of the augmented matrix: J < n -> A(:, J), else B(:, J - n)
    auto elem = [&](int prow_, int J) -> double2* {
        return J < n ? a + (size_t)prow_ * n + J : bb + (size_t)prow_ * n + (J - n);
    };

    // Retire the matrix for every workgroup working on it: ABORT goes into all eight hand-over
    // words with a MAXIMUM (publications are maxima too, so a later publication cannot erase it,
    // and the arrival counters only grow), so whichever word a workgroup waits on, it leaves at once.
    auto abort_all = [&]() {
        for (int q = 0; q < 8; ++q)
            __hip_atomic_fetch_max(flag_pub + q, ABORT, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    };
    // whole-workgroup wait for *flag >= v; false on abort / time-out (the matrix is retired)
    auto wg_wait = [&](int* flag, int v) -> bool {
        if (tid == 0) sh.go = spin_ge(flag, v, ctl.spin_limit);
        __syncthreads();
        const int g = sh.go;
        __syncthreads();
        if (g < 0) {
            if (tid == 0) {
                tr_out[b] = make_double2(__builtin_nan(""), __builtin_nan(""));
                info_out[b] = INFO_TIMEOUT;
                abort_all();
            }
            return false;
        }
        // thread 0's acquire fence invalidated this CU's L1 / this XCD's L2 before the barrier
        return g < ABORT;
    };
    // 16-aligned cuts of the column range [0, n)
    auto cut16 = [&](double frac) -> int {
        const int c = ((int)(frac * n) + 8) / 16 * 16;
        return c < n ? c : n;
    };

    // X(rows below block k0, columns Jlo ..) -= L21 * U12 with the panel rows fetched from A in
    // chunks of RC rows (CHUNK builds; the row map already holds the new order)
    auto t2_chunked = [&](int k0, int nbk, int nrem, int Jlo, int ncols, int wave_, int lane_) {
        for (int r0 = 0; r0 < nrem - nbk; r0 += RC) {
            const int nr = min(RC, nrem - nbk - r0);
            for (int e = tid; e < nr * NB; e += BT) {
                const int r = e / NB, c = e % NB;
                panel[r * LS + c] = c < nbk ? a[(size_t)rowmap[k0 + nbk + r0 + r] * n + k0 + c]
                                            : make_double2(0.0, 0.0);
            }
            __syncthreads();
            mfma_update<false>(n, a, bb, rowmap, panel, k0 + nbk + r0, nr, k0, nbk, Jlo, ncols, wave_, lane_);
            __syncthreads();
        }
    };

    // ================= forward elimination, NB columns per step =======================
    if (SPLIT && role > 0) {
        // ---- roles 1..S: apply the published panels to B's columns f0 .. f1-1 ---------------
        // (with look-ahead, roles 1..na instead carry A's trailing columns beyond the next panel
        // and report every finished step in flags[3], flags[4]; the others share B)
        const bool a_helper = role <= na;
        const int a_lo = (a_helper && role > 1) ? ctl.a_cut[role - 2] : 0;
        const int a_hi = (a_helper && role < na) ? ctl.a_cut[role - 1] : n;
        const int hB = S - na, rB = role - na - 1;  // B helpers / this one's rank
        const int f0 = a_helper ? 0 : cut16((double)rB / hB);
        const int f1 = a_helper ? 0 : (rB == hB - 1 ? n : cut16((double)(rB + 1) / hB));
        for (int k0 = 0, kblk = 0; k0 < n; k0 += NB, ++kblk) {
            const int nbk = min(NB, n - k0);
            const int nrem = n - k0;
            if (!wg_wait(flag_pub, kblk + 1)) return;
            for (int r = tid; r < nrem; r += BT) rowmap[k0 + r] = snap[(size_t)kblk * n + k0 + r];
            __syncthreads();
            // this role's columns of [A | B] at this step (an A-helper whose columns are all
            // factored by now only keeps its row map up to date)
            const int Jhi = a_helper ? a_hi : n + f1;
            const int Jlo = a_helper ? min(Jhi, max(a_lo, k0 + nbk + NB)) : n + f0;
            if (Jhi > Jlo) {  // uniform
                for (int e = tid; e < NB * NB; e += BT) {
                    const int kk = e / NB, c = e % NB;
                    L11[e] = (kk < nbk && c < kk) ? a[(size_t)rowmap[k0 + kk] * n + k0 + c]
                                                  : make_double2(0.0, 0.0);
                }
                if (!CHUNK) {
                    for (int e = tid; e < (nrem - nbk) * NB; e += BT) {
                        const int r = e / NB, c = e % NB;
                        panel[r * LS + c] = c < nbk ? a[(size_t)rowmap[k0 + nbk + r] * n + k0 + c]
                                                    : make_double2(0.0, 0.0);
                    }
                }
                __syncthreads();
                pivot_rows_update(n, a, bb, rowmap, L11, k0, nbk, Jlo, Jhi, wave, lane);
                __syncthreads();
                if (CHUNK)
                    t2_chunked(k0, nbk, nrem, Jlo, Jhi - Jlo, wave, lane);
                else
                    mfma_update<false>(n, a, bb, rowmap, panel, k0 + nbk, nrem - nbk, k0, nbk, Jlo, Jhi - Jlo, wave, lane);
            }
            __syncthreads();  // (every thread's stores are performed; thread 0 releases them)
            if (a_helper && tid == 0)
                __hip_atomic_fetch_max(flag_pub + 2 + role, kblk + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int nbk = min(NB, n - k0);
        const int nrem = n - k0;  // rows still in play; slot t <-> logical position k0 + t
        // ---- panel: row of slot `tid` in registers -------------------------------------
        cd pr[NB];
        int myrow = -1;
        int mypiv = -1;
        bool singular = false;
        if (tid < nrem) {
            myrow = rowmap[k0 + tid];
            if (nbk == NB) {  // (uniform; one batch of loads instead of a test per column)
#pragma unroll
                for (int c = 0; c < NB; ++c) pr[c] = ldg(&a[(size_t)myrow * n + k0 + c]);
            } else {
#pragma unroll
                for (int c = 0; c < NB; ++c)
                    pr[c] = c < nbk ? ldg(&a[(size_t)myrow * n + k0 + c]) : mk(0.0, 0.0);
            }
        }
#pragma unroll
        for (int kk = 0; kk < NB; ++kk) {
            if (kk < nbk) {  // uniform
                // pivot search: max modulus among rows not yet used in this block.  Candidates
                // are 32-bit keys (pivot_key: the high word of |x|^2 with its 10 lowest bits
                // replaced by 1023 - slot), so that a plain integer maximum picks the largest
                // modulus (to 2^-11) and, among equals, the first row.  Wave maximum by DPP.
                unsigned int key = 0;
                if (tid < nrem && mypiv < 0) key = pivot_key(norm2(pr[kk]), tid);
                const unsigned int wkey = wave_max32(key);
                // ONE barrier per column: every wave's candidate row goes to LDS with its key (keys carry
                // the slot: one lane per wave matches), and the winner's is read back from its wave's place
                double2* cw = cand + ((kk & 1) * BW + wave) * NB;
                if (key == wkey && key != 0u) {
#pragma unroll
                    for (int c = kk; c < NB; ++c) cw[c] = make_double2(pr[c].x, pr[c].y);
                }
                if (lane == 0) sh.s_key[kk & 1][wave] = wkey;
                __syncthreads();
                key = row16_max32(sh.s_key[kk & 1][lane & 15]);  // BW = 16 waves: one candidate per lane
                int pt = 1023 - (int)(key & 1023u);
                if (!pivot_ok(key)) {  // exactly singular (or NaN) column
                    if (tid == 0 && !singular) sh.info = k0 + kk + 1;
                    singular = true;
                    pt = -1;
                }
                if (pt >= 0) {
                    const double2* prow = cand + ((kk & 1) * BW + (pt >> 6)) * NB;
                    if (tid == pt) {
                        mypiv = kk;
#pragma unroll
                        for (int c = 0; c < kk; ++c)  // left of kk: the pivot row's multipliers
                            L11[kk * NB + c] = make_double2(pr[c].x, pr[c].y);
                    }
                    if (tid < nrem && mypiv < 0) {
                        const cd f = pr[kk] * rcp(mk(prow[kk].x, prow[kk].y));
                        pr[kk] = f;  // multiplier L(row, k0+kk)
#pragma unroll
                        for (int c = kk + 1; c < NB; ++c)  // (columns >= nbk of a short last panel hold zeros: no test)
                            pr[c] = cmsub(pr[c], f, prow[c]);
                    }
                }
            }
        }
        if (singular) break;  // uniform: every thread made the same reduction

        // ---- publish the panel: U11 -> global, new row order, multipliers -> LDS -------------
        // (pivots per wave, for the ranks below)
        const unsigned long long pbal = __ballot(tid < nrem && mypiv >= 0);
        if ((tid & 63) == 0) sh.s_idx[tid >> 6] = __popcll(pbal);
        // A pivot row goes back to A in place, whole: U from its pivot on, its multipliers left of it (the
        // helpers read them there).  ONE masked region of 16 stores -- with a test per column (c >= mypiv)
        // every store sat behind a wait for the one before it, 2.8 us per panel.
        if (tid < nrem && mypiv >= 0) {
            if (nbk == NB) {
#pragma unroll
                for (int c = 0; c < NB; ++c) stg(&a[(size_t)myrow * n + k0 + c], pr[c]);
            } else {
#pragma unroll
                for (int c = 0; c < NB; ++c)
                    if (c < nbk) stg(&a[(size_t)myrow * n + k0 + c], pr[c]);
            }
        }
        __syncthreads();
        // new row order: the nbk pivots first (in pivot order), then the others, stable; the
        // multipliers of a non-pivot row go to LDS at its NEW position (the trailing update
        // walks rows in that order), rows LS = NB + 1 entries apart (bank spread for the
        // 16-row operand reads of the MFMA tiles)
        if (tid < nrem) {
            int pos;
            if (mypiv >= 0) {
                pos = mypiv;
            } else {
                // rank among non-pivot slots = tid - #pivots before tid (lower waves' counts + this wave's lanes below)
                int before = __popcll(pbal & ((1ull << (tid & 63)) - 1ull));
                for (int w = 0; w < (tid >> 6); ++w) before += sh.s_idx[w];
                pos = nbk + (tid - before);
                if (CHUNK) {  // straight to A: the panel does not fit in LDS
#pragma unroll
                    for (int c = 0; c < NB; ++c)
                        if (c < nbk) stg(&a[(size_t)myrow * n + k0 + c], pr[c]);
                } else {
#pragma unroll
                    for (int c = 0; c < NB; ++c)
                        panel[(pos - nbk) * LS + c] = make_double2(pr[c].x, pr[c].y);
                }
            }
            rowmap[k0 + pos] = myrow;
        }
        __syncthreads();

        if (SPLIT) {
            // helpers read the multipliers from A (the usual in-place LU layout: L below / left
            // of the pivots; copied here from LDS) and the row order from a snapshot
            const int kblk = k0 / NB;
            if (tid < nrem) snap[(size_t)kblk * n + k0 + tid] = rowmap[k0 + tid];
            if (!CHUNK) {  // the other rows' multipliers: from LDS, 16 lanes per row
                for (int e = tid; e < (nrem - nbk) * NB; e += BT) {
                    const int r = e / NB, c = e % NB;
                    if (c < nbk) a[(size_t)rowmap[k0 + nbk + r] * n + k0 + c] = panel[r * LS + c];
                }
            }
            __syncthreads();  // every thread's stores are performed; thread 0 releases them
            if (tid == 0)  // (a maximum, not a store: an ABORT already there must survive)
                __hip_atomic_fetch_max(flag_pub, kblk + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- trailing update, one column per lane -------------------------------------------
        // (thread coordinates recomputed from an opaque copy: whatever the update derives from
        // them must not stay live across the panel loop, which needs the registers itself)
        int tid_t = tid;
        if (SPLIT) asm volatile("" : "+v"(tid_t));  // (the one-workgroup build is better off without)
        const int lane = tid_t & 63, wave = SPLIT ? __builtin_amdgcn_readfirstlane(tid_t >> 6) : tid_t >> 6;
        const int J0 = k0 + nbk;              // first trailing column of A
        if (SPLIT && la && k0 > 0 && J0 < n) {
            // panel j goes onto block j+1 after that block's A-helper has put panels 0 .. j-1 there
            int owner = 0;
            for (int h = 0; h + 1 < na; ++h) owner += J0 >= ctl.a_cut[h];
            if (!wg_wait(flag_pub + 3 + owner, k0 / NB)) return;
        }
        // trailing A columns (+ all of B without helpers; the next panel's only with look-ahead)
        const int Jend = SPLIT ? (la ? min(n, J0 + NB) : n) : 2 * n;
        const int ncols = Jend - J0;
        const int nchunks = (ncols + 63) / 64;
        // T1: finish the NB pivot rows of every column (unit-lower solve with L11 in
        //     registers) and store them: they are rows of U and of L^-1 P B.
        for (int q = wave; q < nchunks; q += BW) {
            const int J = J0 + q * 64 + lane;
            if (J < Jend) {
                cd u[NB];
#pragma unroll
                for (int kk = 0; kk < NB; ++kk)
                    u[kk] = kk < nbk ? ldg(elem(rowmap[k0 + kk], J)) : mk(0.0, 0.0);
                t1_solve(u, L11, nbk);
#pragma unroll
                for (int kk = 0; kk < NB; ++kk)
                    if (kk < nbk) stg(elem(rowmap[k0 + kk], J), u[kk]);
            }
        }
        __syncthreads();
        // T2: X(rows below the block, trailing columns) -= L21 * U12 on the matrix cores
        if (CHUNK) {
            if (ncols > 0) t2_chunked(k0, nbk, nrem, J0, ncols, wave, lane);
        } else if (!SPLIT || ncols > 0)
            mfma_update<false>(n, a, bb, rowmap, panel, k0 + nbk, nrem - nbk, k0, nbk, J0, ncols, wave, lane);
        __syncthreads();
    }

    __syncthreads();  // sh.info (written by thread 0 on a singular column) is visible
    if (sh.info != 0) {  // role 0 only
        if (tid == 0) {
            tr_out[b] = make_double2(__builtin_nan(""), __builtin_nan(""));
            info_out[b] = sh.info;
            if (SPLIT) abort_all();
        }
        return;
    }
    int c0 = 0, c1 = n;  // this role's columns of the back substitution
    if (SPLIT) {
        // L^-1 P B is complete when every helper has finished its columns
        if (role > 0) {
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(flag_pub + 1, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!wg_wait(flag_pub + 1, S)) return;
        // column c costs (n - c)^2 / 2: equal shares of the sum
        c0 = role == 0 ? 0 : cut16(1.0 - cbrt(1.0 - (double)role / nwg));
        c1 = role == nwg - 1 ? n : cut16(1.0 - cbrt(1.0 - (double)(role + 1) / nwg));
    }

    // ================= truncated back substitution, NB rows per step ====================
    // rowmap is now the full pivot order: logical row r = physical row rowmap[r];
    // U(r, c) = a[rowmap[r]][c] for c >= r, and C = L^-1 P B sits in bb[rowmap[r]][:].
    double2* diag = ctl.diag + (size_t)b * n;
    const int nblk = (n + NB - 1) / NB;
    for (int kb = nblk - 1; kb >= 0; --kb) {
        const int k0 = kb * NB;
        const int nbk = min(NB, n - k0);
        const int live = min(c1, k0 + nbk);  // columns c0 .. live-1 of C are still needed
        if (live <= c0) break;               // uniform
        // U11 (upper NB x NB block, with reciprocal diagonal) and the U column block above it
        for (int e = tid; e < NB * NB; e += BT) {
            const int kk = e / NB, c = e % NB;
            double2 v = make_double2(0.0, 0.0);
            if (kk < nbk && c < nbk && c >= kk) {
                cd uv = ldg(&a[(size_t)rowmap[k0 + kk] * n + k0 + c]);
                if (c == kk) uv = rcp(uv);
                v = make_double2(uv.x, uv.y);
            }
            L11[e] = v;
        }
        if (!CHUNK) {
            for (int e = tid; e < (k0 - c0) * NB; e += BT) {  // rows c0 .. k0-1 (none if k0 <= c0)
                const int rr = e / NB, c = e % NB;
                panel[rr * LS + c] = c < nbk ? *(&a[(size_t)rowmap[c0 + rr] * n + k0 + c]) : make_double2(0.0, 0.0);
            }
        }
        __syncthreads();
        const int ncols = live - c0;
        const int nchunks = (ncols + 63) / 64;
        for (int q = wave; q < nchunks; q += BW) {
            const int c = c0 + q * 64 + lane;
            const bool okc = c < live;
            cd x[NB];
#pragma unroll
            for (int kk = 0; kk < NB; ++kk) x[kk] = mk(0.0, 0.0);
            if (okc) {
                if (nbk == NB) {  // (uniform: one batch of loads instead of a test per row)
#pragma unroll
                    for (int kk = 0; kk < NB; ++kk) x[kk] = ldg(&bb[(size_t)__builtin_amdgcn_readfirstlane(rowmap[k0 + kk]) * n + c]);
                } else {
#pragma unroll
                    for (int kk = 0; kk < NB; ++kk)
                        if (kk < nbk) x[kk] = ldg(&bb[(size_t)rowmap[k0 + kk] * n + c]);
                }
            }
            back_solve(x, L11, k0, c, okc, diag);  // upper-triangular solve in registers
            // the solved block replaces C's block rows: it is the B operand of the update below
            if (okc) {
                if (nbk == NB) {
#pragma unroll
                    for (int kk = 0; kk < NB; ++kk) stg(&bb[(size_t)__builtin_amdgcn_readfirstlane(rowmap[k0 + kk]) * n + c], x[kk]);
                } else {
#pragma unroll
                    for (int kk = 0; kk < NB; ++kk)
                        if (kk < nbk) stg(&bb[(size_t)rowmap[k0 + kk] * n + c], x[kk]);
                }
            }
        }
        __syncthreads();
        // rows above the block: C(rr, c) -= sum_k U(rr, k0+k) X(k0+k, c), needed for rr >= c only
        // (column tiles right of the row tile are skipped)
        if (CHUNK) {
            // the U column block in chunks of RC rows (c0 and RC are multiples of 16)
            for (int r0 = c0; r0 < k0; r0 += RC) {
                const int nr = min(RC, k0 - r0);
                for (int e = tid; e < nr * NB; e += BT) {
                    const int rr = e / NB, c = e % NB;
                    panel[rr * LS + c] = c < nbk ? a[(size_t)rowmap[r0 + rr] * n + k0 + c] : make_double2(0.0, 0.0);
                }
                __syncthreads();
                mfma_update<true>(n, a, bb, rowmap, panel, r0, nr, k0, nbk, n + c0, ncols, wave, lane, (r0 - c0) / 16);
                __syncthreads();
            }
        } else if (k0 > c0)
            mfma_update<true>(n, a, bb, rowmap, panel, c0, k0 - c0, k0, nbk, n + c0, ncols, wave, lane);
        __syncthreads();
    }
    // trace = sum of X(c,c) in a fixed order, by the last workgroup of this matrix to get here
    __syncthreads();
    if (SPLIT) {
        if (tid == 0)
            sh.go = __hip_atomic_fetch_add(flag_pub + 2, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (sh.go != nwg - 1) return;
    }
    if (wave == 0) {
        cd t = mk(0.0, 0.0);
        for (int c = lane; c < n; c += 64) {
            const double2 v = diag[c];
            t = t + mk(v.x, v.y);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            t.x += __shfl_xor(t.x, off);
            t.y += __shfl_xor(t.y, off);
        }
        if (lane == 0) {
            tr_out[b] = make_double2(t.x, t.y);
            info_out[b] = 0;
        }
    }
}

// The panel step of k_trace_solve_blocked as a function of its own (for k_trace_solve_grouped): factor the NB
// columns from k0 on (rows in registers, pivot search, multipliers), publish U11, the new row order,
// the multipliers (LDS panel and, in place, A) and the panel counter.  Returns 0, or LAPACK's info of an
// exactly singular column (nothing is published then).  Not inlined, so that its register allocation is
// its own whatever else the grouped kernel calls (inlined next to the grouped update the panel rows end up in scratch).
// publish_ = 0: the panel counter is left alone (the helpers of the grouped kernel wait for whole GROUPS of panels: an
// agent-scope release is a write-back of the XCD's whole L2 -- buffer_wbl2 -- and only the group's last panel needs one:
// n = 512, 128 matrices: 4.83 -> 4.74 ms; n = 256: 1.01 -> 0.96).
// (Measured and not kept: hand-overs WITHOUT the L2 write-back / invalidate when both roles of a matrix sit on one XCD
// -- they do, checked through HW_REG_XCC_ID with the role stride padded to a multiple of 8; producer s_waitcnt vmcnt(0),
// consumer buffer_inv sc0 -- correct and no faster: the trailing passes stream 4-8 MB per matrix through a 4-MB L2.)
__device__ __noinline__ int factor_panel(int n_, double2* a_, int k0_, int* snap_, int* flag_pub_, int publish_) {
    const int n = uniform(n_), k0 = uniform(k0_), publish = uniform(publish_);
    double2* a = uniform(a_);
    int* snap = uniform(snap_);
    int* flag_pub = uniform(flag_pub_);
    extern __shared__ double2 lds2[];
    __shared__ BlkShared shp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* rowmap = reinterpret_cast<int*>(lds2);
    double2* L11 = lds2 + (3 * n * (int)sizeof(int) + 15) / 16;
    double2* panel = L11 + NB * NB + NB;
    double2* cand = panel;  // [2][BW][NB] candidate pivot rows: the panel area is free until the multipliers go there
    const int nbk = min(NB, n - k0);
    const int nrem = n - k0;
    cd pr[NB];
    int myrow = -1;
    int mypiv = -1;
    int sing_info = 0;  // (uniform: every thread makes the same reductions)
    LU_T(tf0);
    if (tid < nrem) {
        myrow = rowmap[k0 + tid];
        if (nbk == NB) {  // (uniform; one batch of loads instead of a test per column)
#pragma unroll
            for (int c = 0; c < NB; ++c) pr[c] = ldg(&a[(size_t)myrow * n + k0 + c]);
        } else {
#pragma unroll
            for (int c = 0; c < NB; ++c)
                pr[c] = c < nbk ? ldg(&a[(size_t)myrow * n + k0 + c]) : mk(0.0, 0.0);
        }
    }
#ifdef EMME_LU_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    LU_T(tf1);
    LU_ADD(8, tf0, tf1);
#pragma unroll
    for (int kk = 0; kk < NB; ++kk) {
        if (kk < nbk) {  // uniform
            // pivot search: max modulus among rows not yet used in this block.  Candidates
            // are 32-bit keys (pivot_key: the high word of |x|^2 with its 10 lowest bits
            // replaced by 1023 - slot), so that a plain integer maximum picks the largest
            // modulus (to 2^-11) and, among equals, the first row.  Wave maximum by DPP.
            unsigned int key = 0;
            if (tid < nrem && mypiv < 0) key = pivot_key(norm2(pr[kk]), tid);
            const unsigned int wkey = wave_max32(key);
            // ONE barrier per column: every wave's candidate row goes to LDS with its key (see the kernel)
            double2* cw = cand + ((kk & 1) * BW + wave) * NB;
            if (key == wkey && key != 0u) {
#pragma unroll
                for (int c = kk; c < NB; ++c) cw[c] = make_double2(pr[c].x, pr[c].y);
            }
            if (lane == 0) shp.s_key[kk & 1][wave] = wkey;
            __syncthreads();
            key = row16_max32(shp.s_key[kk & 1][lane & 15]);  // BW = 16 waves: one candidate per lane
            int pt = 1023 - (int)(key & 1023u);
            if (!pivot_ok(key)) {  // exactly singular (or NaN) column
                if (sing_info == 0) sing_info = k0 + kk + 1;
                pt = -1;
            }
            if (pt >= 0) {
                const double2* prow = cand + ((kk & 1) * BW + (pt >> 6)) * NB;
                if (tid == pt) {
                    mypiv = kk;
#pragma unroll
                    for (int c = 0; c < kk; ++c)  // left of kk: the pivot row's multipliers
                        L11[kk * NB + c] = make_double2(pr[c].x, pr[c].y);
                }
                if (tid < nrem && mypiv < 0) {
                    const cd f = pr[kk] * rcp(mk(prow[kk].x, prow[kk].y));
                    pr[kk] = f;  // multiplier L(row, k0+kk)
#pragma unroll
                    for (int c = kk + 1; c < NB; ++c)  // (columns >= nbk of a short last panel hold zeros: no test)
                        pr[c] = cmsub(pr[c], f, prow[c]);
                }
            }
        }
    }
    if (sing_info != 0) return sing_info;
    LU_T(tf2);
    LU_ADD(9, tf1, tf2);

    // ---- publish the panel: U11 -> global, new row order, multipliers -> LDS -------------
    // (pivots per wave, for the ranks below)
    const unsigned long long pbal = __ballot(tid < nrem && mypiv >= 0);
    if (lane == 0) shp.s_idx[wave] = __popcll(pbal);
    // a pivot row goes back to A in place, whole (see the kernel): U from the pivot on, its multipliers left of it
    if (tid < nrem && mypiv >= 0) {
        if (nbk == NB) {
#pragma unroll
            for (int c = 0; c < NB; ++c) stg(&a[(size_t)myrow * n + k0 + c], pr[c]);
        } else {
#pragma unroll
            for (int c = 0; c < NB; ++c)
                if (c < nbk) stg(&a[(size_t)myrow * n + k0 + c], pr[c]);
        }
    }
    __syncthreads();
    // new row order: the nbk pivots first (in pivot order), then the others, stable; the
    // multipliers of a non-pivot row go to LDS at its NEW position (the trailing update
    // walks rows in that order), rows LS = NB + 1 entries apart (bank spread for the
    // 16-row operand reads of the MFMA tiles)
    if (tid < nrem) {
        int pos;
        if (mypiv >= 0) {
            pos = mypiv;
        } else {
            // rank among non-pivot slots = tid - #pivots before tid (lower waves' counts + this wave's lanes below)
            int before = __popcll(pbal & ((1ull << lane) - 1ull));
            for (int w = 0; w < wave; ++w) before += shp.s_idx[w];
            pos = nbk + (tid - before);
#pragma unroll
            for (int c = 0; c < NB; ++c)
                panel[(pos - nbk) * LS + c] = make_double2(pr[c].x, pr[c].y);
        }
        rowmap[k0 + pos] = myrow;
    }
    __syncthreads();
    LU_T(tf3);
    LU_ADD(10, tf2, tf3);

    {
        // helpers read the multipliers from A (the usual in-place LU layout: L below / left
        // of the pivots; copied here from LDS) and the row order from a snapshot
        const int kblk = k0 / NB;
        if (tid < nrem) snap[(size_t)kblk * n + k0 + tid] = rowmap[k0 + tid];
        for (int e = tid; e < (nrem - nbk) * NB; e += BT) {  // the other rows' multipliers: from LDS, 16 lanes per row
            const int r = e / NB, c = e % NB;
            if (c < nbk) a[(size_t)rowmap[k0 + nbk + r] * n + k0 + c] = panel[r * LS + c];
        }
        __syncthreads();  // every thread's stores are performed; thread 0 releases them
        if (tid == 0 && publish)  // (a maximum, not a store: an ABORT already there must survive)
            __hip_atomic_fetch_max(flag_pub, kblk + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    LU_T(tf4);
    LU_ADD(11, tf3, tf4);
    return 0;
}

// The truncated back substitution of k_trace_solve_blocked for columns c0 .. c1-1 (see there), as a function,
// with the same grouping as the forward sweep: a block of NB rows is solved and applied to the rows above it
// INSIDE its group of GP blocks only; the rows above the group get the whole group in one pass
// (mfma_update_grouped<LOWER>, panels last to first: the order, and the bits, of the block-by-block sweep).
__device__ __noinline__ void back_substitute(int n_, double2* a_, double2* bb_, double2* diag_, int c0_, int c1_) {
    const int n = uniform(n_), c0 = uniform(c0_), c1 = uniform(c1_);
    double2* a = uniform(a_);
    double2* bb = uniform(bb_);
    double2* diag = uniform(diag_);
    extern __shared__ double2 lds2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int* rowmap = reinterpret_cast<const int*>(lds2);
    double2* L11 = lds2 + (3 * n * (int)sizeof(int) + 15) / 16;
    double2* panel = L11 + NB * NB + NB;
    const int nblk = (n + NB - 1) / NB;
    for (int g0 = (n - 1) / GK * GK; g0 >= 0; g0 -= GK) {
        const int ng = min(GK, n - g0);
        if (min(c1, g0 + ng) <= c0) break;  // uniform: none of this role's columns reach down to these rows
        const int top = max(c0, g0);        // first row of the group this role still needs
        for (int kb = min(nblk, (g0 + ng + NB - 1) / NB) - 1; kb * NB >= g0; --kb) {
            const int k0 = kb * NB;
            const int nbk = min(NB, n - k0);
            const int live = min(c1, k0 + nbk);  // columns c0 .. live-1 of C are still needed
            if (live <= c0) break;               // uniform
            // U11 (upper NB x NB block, with reciprocal diagonal) and the U column block above it, inside the group
            for (int e = tid; e < NB * NB; e += BT) {
                const int kk = e / NB, c = e % NB;
                double2 v = make_double2(0.0, 0.0);
                if (kk < nbk && c < nbk && c >= kk) {
                    cd uv = ldg(&a[(size_t)rowmap[k0 + kk] * n + k0 + c]);
                    if (c == kk) uv = rcp(uv);
                    v = make_double2(uv.x, uv.y);
                }
                L11[e] = v;
            }
            for (int e = tid; e < (k0 - top) * NB; e += BT) {  // rows top .. k0-1 (none if k0 <= top)
                const int rr = e / NB, c = e % NB;
                panel[rr * LS + c] = c < nbk ? a[(size_t)rowmap[top + rr] * n + k0 + c] : make_double2(0.0, 0.0);
            }
            __syncthreads();
            const int ncols = live - c0;
            const int nchunks = (ncols + 63) / 64;
            for (int q = wave; q < nchunks; q += BW) {
                const int c = c0 + q * 64 + lane;
                const bool okc = c < live;
                cd x[NB];
#pragma unroll
                for (int kk = 0; kk < NB; ++kk) x[kk] = mk(0.0, 0.0);
                if (okc) {
                    if (nbk == NB) {  // (uniform: one batch of loads instead of a test per row)
#pragma unroll
                        for (int kk = 0; kk < NB; ++kk) x[kk] = ldg(&bb[(size_t)__builtin_amdgcn_readfirstlane(rowmap[k0 + kk]) * n + c]);
                    } else {
#pragma unroll
                        for (int kk = 0; kk < NB; ++kk)
                            if (kk < nbk) x[kk] = ldg(&bb[(size_t)rowmap[k0 + kk] * n + c]);
                    }
                }
                back_solve(x, L11, k0, c, okc, diag);  // upper-triangular solve in registers
                // the solved block replaces C's block rows: it is the B operand of the updates below
                if (okc) {
                    if (nbk == NB) {
#pragma unroll
                        for (int kk = 0; kk < NB; ++kk) stg(&bb[(size_t)__builtin_amdgcn_readfirstlane(rowmap[k0 + kk]) * n + c], x[kk]);
                    } else {
#pragma unroll
                        for (int kk = 0; kk < NB; ++kk)
                            if (kk < nbk) stg(&bb[(size_t)rowmap[k0 + kk] * n + c], x[kk]);
                    }
                }
            }
            __syncthreads();
            // rows above the block, inside the group: C(rr, c) -= sum_k U(rr, k0+k) X(k0+k, c), needed for rr >= c
            // only (column tiles right of the row tile are skipped)
            if (k0 > top)
                mfma_update<true>(n, a, bb, rowmap, panel, top, k0 - top, k0, nbk, n + c0, ncols, wave, lane, (top - c0) / 16);
            __syncthreads();
        }
        // rows above the group: all of the group's blocks in one pass
        if (g0 > c0) {
            const int rc = max(16, (n * LS / GLS) / 16 * 16);
            mfma_update_grouped<true>(n, a, bb, rowmap, panel, rc, c0, g0 - c0, g0, ng, n + c0, min(c1, g0) - c0, tid, wave, lane);
        }
    }
}

// k_trace_solve_blocked<true> (no look-ahead, whole L21 panel in LDS) with the trailing matrix updated once
// per GROUP of GP panels instead of once per panel: role 0 factors a panel (factor_panel), applies it
// to the rest of its group's columns only, and at the end of a group puts the whole group onto A's
// columns behind it in one pass (apply_group); the helpers wait for a whole group and do the same for
// their columns of B.  The same operations on every element in the same order -- the same bits -- with
// a quarter of the passes over the trailing matrix.  The kernel itself is a driver: panel, grouped
// update and back substitution are functions with register allocations of their own.
__global__ __launch_bounds__(BT) void k_trace_solve_grouped(int n, int nbatch, double2* A, double2* B,
                                                            const int* active, double2* tr_out,
                                                            int* info_out, SplitCtl ctl) {
    extern __shared__ double2 lds2[];
    __shared__ BlkShared sh;
    const int role = blockIdx.x / ctl.nitems;
    int b = blockIdx.x - role * ctl.nitems;
    if (ctl.items) b = ctl.items[b];
    if (active && active[b] == 0) return;
    const int nwg = ctl.nwg, S = nwg - 1;
    int* flag_pub = ctl.flags + 8 * b;
    int* snap = ctl.rowmaps + (size_t)b * ((n + NB - 1) / NB) * n;
    double2* a = A + (size_t)b * n * n;
    double2* bb = B + (size_t)b * n * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* rowmap = reinterpret_cast<int*>(lds2);
    double2* L11 = lds2 + (3 * n * (int)sizeof(int) + 15) / 16;
    double2* panel = L11 + NB * NB + NB;

    for (int r = tid; r < n; r += BT) rowmap[r] = r;
    if (tid == 0) sh.info = 0;
    __syncthreads();

    auto abort_all = [&]() {
        for (int q = 0; q < 8; ++q)
            __hip_atomic_fetch_max(flag_pub + q, ABORT, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto wg_wait = [&](int* flag, int v) -> bool {
        if (tid == 0) sh.go = spin_ge(flag, v, ctl.spin_limit);
        __syncthreads();
        const int g = sh.go;
        __syncthreads();
        if (g < 0) {
            if (tid == 0) {
                tr_out[b] = make_double2(__builtin_nan(""), __builtin_nan(""));
                info_out[b] = INFO_TIMEOUT;
                abort_all();
            }
            return false;
        }
        return g < ABORT;
    };
    auto cut16 = [&](double frac) -> int {
        const int c = ((int)(frac * n) + 8) / 16 * 16;
        return c < n ? c : n;
    };

    if (role > 0) {
        // ---- helpers: columns f0 .. f1-1 of B, a group of panels at a time ---------------------------
        const int f0 = cut16((double)(role - 1) / S);
        const int f1 = role == S ? n : cut16((double)role / S);
        for (int g0 = 0; g0 < n; g0 += GK) {
            const int ng = min(GK, n - g0);
            const int npan = (ng + NB - 1) / NB;
            LU_T(tw0);
            if (!wg_wait(flag_pub, g0 / NB + npan)) return;
            LU_T(tw1);
            LU_ADD(16, tw0, tw1);
            // row order after the group's last panel: position x was fixed by panel (x - g0) / NB
            for (int x = g0 + tid; x < n; x += BT) {
                const int pq = min(npan - 1, (x - g0) / NB);
                rowmap[x] = snap[(size_t)(g0 / NB + pq) * n + x];
            }
            __syncthreads();
            apply_group(n, a, bb, g0, ng, n + f0, n + f1);
            LU_T(tw2);
            LU_ADD(17, tw1, tw2);
        }
    } else {
        for (int k0 = 0; k0 < n; k0 += NB) {
            const int nbk = min(NB, n - k0);
            LU_T(tp0);
            // (the helpers wait for whole groups: only a group's last panel is released)
            const int last_of_group = k0 + nbk == min(n, (k0 / GK) * GK + GK);
            const int inf = factor_panel(n, a, k0, snap, flag_pub, last_of_group);
            if (inf != 0) {  // uniform
                if (tid == 0) sh.info = inf;
                break;
            }
            LU_T(tp1);
            LU_ADD(0, tp0, tp1);
            // the rest of this group's columns: pivot rows (T1), then the rows below (T2)
            const int J0 = k0 + nbk;
            const int gend = min(n, (k0 / GK) * GK + GK);
            if (gend > J0) {
                pivot_rows_update(n, a, bb, rowmap, L11, k0, nbk, J0, gend, wave, lane);
                __syncthreads();
                mfma_update<false>(n, a, bb, rowmap, panel, k0 + nbk, n - k0 - nbk, k0, nbk, J0, gend - J0, wave, lane);
                __syncthreads();
            }
            LU_T(tp3);
            LU_ADD(2, tp1, tp3);
            if (J0 == gend && gend < n) {
                // the group is complete: its panels go onto A's columns behind it in one pass
                const int g0 = (k0 / GK) * GK;
                apply_group(n, a, bb, g0, gend - g0, gend, n);
                LU_T(tp4);
                LU_ADD(3, tp3, tp4);
            }
        }
    }
    LU_T(tb0);
    __syncthreads();  // sh.info is visible
    if (sh.info != 0) {  // role 0 only
        if (tid == 0) {
            tr_out[b] = make_double2(__builtin_nan(""), __builtin_nan(""));
            info_out[b] = sh.info;
            abort_all();
        }
        return;
    }
    // L^-1 P B is complete when every helper has finished its columns
    if (role > 0) {
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(flag_pub + 1, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!wg_wait(flag_pub + 1, S)) return;
    LU_T(tb1);
    LU_ADD(role == 0 ? 4 : 18, tb0, tb1);
    const int c0 = role == 0 ? 0 : cut16(1.0 - cbrt(1.0 - (double)role / nwg));
    const int c1 = role == nwg - 1 ? n : cut16(1.0 - cbrt(1.0 - (double)(role + 1) / nwg));
    double2* diag = ctl.diag + (size_t)b * n;
    back_substitute(n, a, bb, diag, c0, c1);
    __syncthreads();
    LU_T(tb2);
    LU_ADD(role == 0 ? 5 : 19, tb0, tb2);
    LU_ADD(role == 0 ? 6 : 20, 0ull, 1ull);
    if (tid == 0) sh.go = __hip_atomic_fetch_add(flag_pub + 2, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (sh.go != nwg - 1) return;
    if (wave == 0) {
        cd t = mk(0.0, 0.0);
        for (int c = lane; c < n; c += 64) {
            const double2 v = diag[c];
            t = t + mk(v.x, v.y);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            t.x += __shfl_xor(t.x, off);
            t.y += __shfl_xor(t.y, off);
        }
        if (lane == 0) {
            tr_out[b] = make_double2(t.x, t.y);
            info_out[b] = 0;
        }
    }
}

// The factorisation alone (nullSpace, nullspace.hip): role 0's forward sweep of k_trace_solve_grouped on A only --
// factor_panel, the rest of the group's columns, apply_group behind the group -- with no right-hand sides, no
// helper workgroups and therefore no waits.  Leaves P A = L U in place (multipliers left of the pivots, every
// row where it was) and the row order in the panel snapshots: logical row x = physical row snap[(x / NB) n + x].
__global__ __launch_bounds__(BT) void k_lu_inplace(int n, double2* A, const int* items, int* info_out, SplitCtl ctl) {
    extern __shared__ double2 lds2[];
    __shared__ int s_info;
    const int b = items ? items[blockIdx.x] : blockIdx.x;
    int* flag_pub = ctl.flags + 8 * b;
    int* snap = ctl.rowmaps + (size_t)b * ((n + NB - 1) / NB) * n;
    double2* a = A + (size_t)b * n * n;
    double2* bnone = a;  // (no right-hand sides: every column range below ends at n, the pointer is never followed)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* rowmap = reinterpret_cast<int*>(lds2);
    double2* L11 = lds2 + (3 * n * (int)sizeof(int) + 15) / 16;
    double2* panel = L11 + NB * NB + NB;
    for (int r = tid; r < n; r += BT) rowmap[r] = r;
    if (tid == 0) s_info = 0;
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int nbk = min(NB, n - k0);
        const int inf = factor_panel(n, a, k0, snap, flag_pub, 0);  // (no helpers: nothing to release)
        if (inf != 0) {  // uniform
            if (tid == 0) s_info = inf;
            break;
        }
        const int J0 = k0 + nbk;
        const int gend = min(n, (k0 / GK) * GK + GK);
        if (gend > J0) {
            pivot_rows_update(n, a, bnone, rowmap, L11, k0, nbk, J0, gend, wave, lane);
            __syncthreads();
            mfma_update<false>(n, a, bnone, rowmap, panel, k0 + nbk, n - k0 - nbk, k0, nbk, J0, gend - J0, wave, lane);
            __syncthreads();
        }
        if (J0 == gend && gend < n) {
            const int g0 = (k0 / GK) * GK;
            apply_group(n, a, bnone, g0, gend - g0, gend, n);
        }
    }
    __syncthreads();
    if (tid == 0) info_out[b] = s_info;
}

}  // namespace

int trace_solve_nb() { return NB; }

// P A = L U in place for the matrices listed in `items` (device list of nitems batch indices, null = 0 .. nitems-1),
// no right-hand sides: for nullSpace.  n must fit the one-workgroup panel (trace_solve_blocked_lds(n) <= 150 KB);
// scratch as for launch_trace_solve_blocked (row orders: trace_solve_rowmaps()).
hipError_t launch_lu_inplace(int n, int nbatch, double* A, const int* items, int nitems, int* info, void* scratch,
                             hipStream_t stream) {
    const size_t lds = trace_solve_blocked_lds(n);
    if (lds > 150 * 1024) return hipErrorNotSupported;
    if (lds > 48 * 1024) {  // (beyond the default dynamic-LDS limit only)
        const hipError_t ea = hipFuncSetAttribute((const void*)k_lu_inplace, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return ea;
    }
    SplitCtl ctl{};
    ctl.nwg = 1;
    ctl.spin_limit = 1;
    ctl.items = items;
    ctl.nitems = nitems;
    ctl.diag = (double2*)scratch;
    ctl.flags = (int*)(ctl.diag + (size_t)nbatch * n);
    ctl.rowmaps = ctl.flags + 8 * (size_t)nbatch;
    hipLaunchKernelGGL(k_lu_inplace, dim3(nitems), dim3(BT), lds, stream, n, (double2*)A, items, info, ctl);
    return hipGetLastError();
}

// where the row orders of a factorisation live inside `scratch` (see trace_solve_blocked_scratch): per matrix
// ceil(n / NB) snapshots of n ints; logical row x of matrix b = physical row maps[(b nblk + x / NB) n + x]
const int* trace_solve_rowmaps(const void* scratch, int n, int nbatch) {
    const double2* diag = (const double2*)scratch;
    const int* flags = (const int*)(diag + (size_t)nbatch * n);
    return flags + 8 * (size_t)nbatch;
}

// LDS of the chunked build (560 < n <= 1024): the panel holds RC rows at a time
size_t trace_solve_chunked_lds(int n) {
    const int rows = n < RC ? n : RC;
    return (((size_t)3 * n * sizeof(int) + 15) / 16 + NB * NB + NB + std::max((size_t)rows * LS, (size_t)2 * BW * NB)) * sizeof(double2);
}

size_t trace_solve_blocked_lds(int n) {
    return (((size_t)3 * n * sizeof(int) + 15) / 16 + NB * NB + NB + std::max((size_t)n * LS, (size_t)2 * BW * NB)) * sizeof(double2);
}

size_t trace_solve_blocked_scratch(int n, int nbatch) {
    const size_t nblk = (size_t)(n + NB - 1) / NB;
    return (size_t)nbatch * (8 * sizeof(int) + nblk * n * sizeof(int) + (size_t)n * sizeof(double2)) + 256;
}

#ifdef EMME_LU_STAMPS
static void lu_stamps_report(hipStream_t stream) {
    if (!std::getenv("EMME_DEBUG_STAMPS")) return;
    (void)hipStreamSynchronize(stream);
    unsigned long long h[32] = {0};
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lu_stamps), sizeof h);
    const double r0 = h[6] ? 1.0 / (100.0 * h[6]) : 0.0, r1 = h[20] ? 1.0 / (100.0 * h[20]) : 0.0;  // s_memtime: 100 MHz
    std::fprintf(stderr, "[lu stamps, us per workgroup] role 0 (%llu): panel %.0f publish %.0f trailing %.0f group %.0f wait-helpers %.0f back+wait %.0f | "
                 "helpers (%llu): wait %.0f apply %.0f wait-all %.0f back+wait %.0f\n", h[6], h[0] * r0, h[1] * r0, h[2] * r0, h[3] * r0, h[4] * r0, h[5] * r0,
                 h[20], h[16] * r1, h[17] * r1, h[18] * r1, h[19] * r1);
    {
        const double ra = (h[6] + h[20]) ? 1.0 / (100.0 * (h[6] + h[20])) : 0.0;
        std::fprintf(stderr, "[lu stamps] apply_group (all roles): stage %.0f T1 %.0f in-group MFMA %.0f grouped MFMA %.0f\n", h[12] * ra, h[13] * ra, h[14] * ra, h[15] * ra);
    }
    std::fprintf(stderr, "[lu stamps] panel: load rows %.0f columns %.0f rank + multipliers to LDS %.0f copy to A + release %.0f\n", h[8] * r0, h[9] * r0,
                 h[10] * r0, h[11] * r0);
    unsigned long long z[32] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lu_stamps), z, sizeof z);
}
#else
static void lu_stamps_report(hipStream_t) {}
#endif

hipError_t launch_trace_solve_blocked(int n, int nbatch, double* A, double* B, const int* active,
                                      double* tr, int* info, int nwg, const int* items, int nitems,
                                      void* scratch, hipStream_t stream, int group_min_n, int spin_limit) {
    // matrices whose whole L21 panel does not fit in LDS go through the chunked build, which
    // exists only with helper workgroups (the multipliers must be in global memory anyway)
    const bool chunk = trace_solve_blocked_lds(n) > 150 * 1024;
    if (chunk && (n > BT || nwg < 2)) return hipErrorNotSupported;
    const size_t lds = chunk ? trace_solve_chunked_lds(n) : trace_solve_blocked_lds(n);
    static thread_local int attr_dev = -1;  // (function attributes are per device)
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    if (attr_dev != cur_dev) {
        (void)hipFuncSetAttribute((const void*)k_trace_solve_blocked<false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        (void)hipFuncSetAttribute((const void*)k_trace_solve_blocked<true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        (void)hipFuncSetAttribute((const void*)k_trace_solve_blocked<true, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        (void)hipFuncSetAttribute((const void*)k_trace_solve_grouped,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        attr_dev = cur_dev;
    }
    if (nwg < 1) nwg = 1;
    // scratch: diag | flags | row-map snapshots
    SplitCtl ctl;
    ctl.nwg = nwg;
    // A-helpers: about 0.4 of the helpers (A's trailing update is n^3/3, B's n^3/2), at most 5
    // (one progress counter each); their column ranges get equal shares of the work
    // w(x) = x^2/2 - x^3/6 of the columns left of x n
    ctl.na = nwg >= 4 ? std::max(1, std::min(5, (int)(0.4 * (nwg - 1) + 0.5))) : 0;
    for (int i = 0; i < 4; ++i) {
        ctl.a_cut[i] = n;
        if (i + 1 < ctl.na) {
            const double target = (double)(i + 1) / (3.0 * ctl.na);
            double lo = 0.0, hi = 1.0;
            for (int it = 0; it < 40; ++it) {
                const double x = 0.5 * (lo + hi);
                (x * x / 2 - x * x * x / 6 < target ? lo : hi) = x;
            }
            ctl.a_cut[i] = std::min(n, ((int)(lo * n) + 8) / 16 * 16);
        }
    }
    ctl.spin_limit = spin_limit > 0 ? spin_limit : SPIN_LIMIT;
    // grouped trailing updates (a quarter of the passes over the matrix): without look-ahead, from the
    // order at which it pays (n = 256, 128 matrices: 1.44 -> 1.34 ms; EMME_LU_GROUP=0 switches it off, =n sets the order)
    if (group_min_n <= 0) group_min_n = 1 << 30;  // (options: lu_group_min_n, -1 = never)
    const bool group = !chunk && nwg > 1 && ctl.na == 0 && n >= group_min_n;
    ctl.items = items;
    ctl.nitems = items ? nitems : nbatch;
    ctl.diag = (double2*)scratch;
    ctl.flags = (int*)(ctl.diag + (size_t)nbatch * n);
    ctl.rowmaps = ctl.flags + 8 * (size_t)nbatch;
    if (nwg > 1) {
        hipError_t e = hipMemsetAsync(ctl.flags, 0, 8 * sizeof(int) * (size_t)nbatch, stream);
        if (e != hipSuccess) return e;
    }
    if (nwg > 1) {
        // The roles of a matrix wait for each other, so the whole grid has to be resident at
        // once: workgroups the device can hold (occupancy x compute units) >= grid, else one
        // workgroup per matrix does the job.  (A cooperative launch would make the same check;
        // it is not used because its extra queue crashes rocprofv3 at process exit.)
        // capacity of the current device for this build and LDS size (per host thread: contexts
        // on different devices may be driven from different threads)
        static thread_local size_t cap_lds = ~(size_t)0;
        static thread_local int cap_dev = -1;
        static thread_local long cap = 0;
        static thread_local int cap_chunk = -1;
        const int variant = chunk ? 1 : group ? 2 : 0;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = -1;
        if (dev < 0 || dev != cap_dev || lds != cap_lds || variant != cap_chunk) {
            int per_cu = 0, ncu = 0;
            hipError_t eo = chunk   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_solve_blocked<true, true>, BT, lds)
                            : group ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_solve_grouped, BT, lds)
                                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_solve_blocked<true>, BT, lds);
            if (dev < 0 || eo != hipSuccess ||
                hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
                (void)hipGetLastError();
                per_cu = 0;
            }
            cap = (long)per_cu * ncu, cap_dev = dev, cap_lds = lds, cap_chunk = variant;
        }
        if (cap >= (long)ctl.nitems * nwg) {
            if (chunk)
                hipLaunchKernelGGL((k_trace_solve_blocked<true, true>), dim3(ctl.nitems * nwg), dim3(BT), lds, stream,
                                   n, nbatch, (double2*)A, (double2*)B, active, (double2*)tr, info, ctl);
            else if (group)
                hipLaunchKernelGGL(k_trace_solve_grouped, dim3(ctl.nitems * nwg), dim3(BT), lds, stream,
                                   n, nbatch, (double2*)A, (double2*)B, active, (double2*)tr, info, ctl);
            else
                hipLaunchKernelGGL(k_trace_solve_blocked<true>, dim3(ctl.nitems * nwg), dim3(BT), lds, stream,
                                   n, nbatch, (double2*)A, (double2*)B, active, (double2*)tr, info, ctl);
            lu_stamps_report(stream);
            return hipGetLastError();
        }
        if (chunk) return hipErrorNotSupported;
        ctl.nwg = 1;
    }
    hipLaunchKernelGGL(k_trace_solve_blocked<false>, dim3(nbatch), dim3(BT), lds, stream, n, nbatch,
                       (double2*)A, (double2*)B, active, (double2*)tr, info, ctl);
    return hipGetLastError();
}

}  // namespace emme
