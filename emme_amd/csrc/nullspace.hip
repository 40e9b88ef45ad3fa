// nullspace.hip -- EigenSolver::nullSpace (reference include/solver.h:58-112, called once per solve by
// src/main.cpp:70-75) on the device, batched over the roots of a search.
//
// The reference takes the last row of V^H of a full SVD (LAPACK zgesdd), i.e. the right singular vector of
// the smallest singular value of M(omega_root).  Here: inverse iteration on M^H M, applied as
// v <- M^-1 (M^-H v) through ONE partial-pivot LU of M itself (M^H M is never formed, so the factorisation
// sees cond(M), not its square); the convergence factor per sweep is (s_n / s_n-1)^2 -- <= 1e-8 at a
// converged root -- and the sweeps stop when the direction has stopped moving.  Same vector as the SVD's up
// to the arbitrary complex phase LAPACK would return.  No symmetry of M is assumed: M^-H goes through the
// transposed triangular solves U^H, L^H of the same factors.
//
// The factorisation is the Newton step's (linstep_blocked.hip): k_lu_inplace for orders whose panel fits one
// workgroup's LDS, the chunked multi-workgroup kernel above that; both leave P M = L U in place with the row
// order in panel snapshots.  k_null_iterate then runs the four triangular solves of a sweep with one
// 1024-thread workgroup per matrix: the non-transposed ones in dot form (a wave per row of a 16-row block,
// rows are contiguous), the transposed ones in axpy form (a thread per column, the 16 rows of the block are
// contiguous across threads), the 16 x 16 triangles by one wave with the block's entries in registers.
#include <hip/hip_runtime.h>

#include <cmath>
#include <type_traits>

#include "emme_device.hpp"
#include "launch.hpp"

namespace emme {

namespace {

constexpr int NT = 512;  // (256 vector registers per thread: the chain wave holds two 16 x 16 coefficient rows)
constexpr int NW = NT / 64;
constexpr int TB = 16;  // rows per block of the triangular solves

__device__ __forceinline__ cd ld2(const double2* p) {
    const double2 v = *p;
    return mk(v.x, v.y);
}
__device__ __forceinline__ cd conj_(cd a) { return mk(a.x, -a.y); }
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

struct NullArgs {
    int n, nb, nblk;       // order; rows per panel snapshot and snapshots per matrix of the factorisation
    const double2* A;      // [nbatch][n][n]  P M = L U in place
    const int* maps;       // [nbatch][nblk][n] row-order snapshots
    const int* items;      // batch indices of this launch (null: 0 .. grid-1)
    const int* lu_info;    // per batch item: 0, or the column at which the factorisation stopped
    double2* vecs;         // [nbatch][n]
    int* info;             // [nbatch]
    int max_sweeps;
};

__global__ __launch_bounds__(NT) void k_null_iterate(NullArgs P) {
    extern __shared__ double2 sm[];
    __shared__ double s_red[2 * NW + 2];
    const int n = P.n;
    double2* v = sm;       // the iterate / right-hand side
    double2* t = sm + n;   // work vector
    double2* o = sm + 2 * (size_t)n;  // the direction before the sweep
    int* rowmap = reinterpret_cast<int*>(sm + 3 * (size_t)n);
    const int b = P.items ? P.items[blockIdx.x] : blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double2* out = P.vecs + (size_t)b * n;
    if (P.lu_info && P.lu_info[b] != 0) {  // no factorisation (exactly singular column / time-out): no vector
        for (int i = tid; i < n; i += NT) out[i] = make_double2(__builtin_nan(""), __builtin_nan(""));
        if (tid == 0) P.info[b] = P.lu_info[b];
        return;
    }
    const double2* a = P.A + (size_t)b * n * n;
    const int* maps = P.maps + (size_t)b * P.nblk * n;
    for (int x = tid; x < n; x += NT) rowmap[x] = maps[(size_t)(x / P.nb) * n + x];
    // the start vector of the host version (host_driver.cpp): no symmetry that M's null vector could be orthogonal to
    for (int i = tid; i < n; i += NT) v[i] = make_double2(1.0 + 0.37 * sin(1.0 + i), 0.21 * cos(2.0 * i));
    __syncthreads();

    // v <- v / |v|; returns |<old, v>| for the caller's `old` (per-thread elements i = tid + k NT)
    auto normalise = [&]() {
        double s = 0.0;
        for (int i = tid; i < n; i += NT) s += v[i].x * v[i].x + v[i].y * v[i].y;
        s = wave_sum(s);
        if (lane == 0) s_red[wave] = s;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += s_red[w];
        const double r = 1.0 / sqrt(tot);
        for (int i = tid; i < n; i += NT) v[i] = make_double2(v[i].x * r, v[i].y * r);
        __syncthreads();
    };
    // ---- the triangular solves, pipelined over blocks of TB = 16 unknowns ------------------------------------------------
    // One right-hand side is a chain of n dependent steps whatever one does; what can be taken OFF that chain is
    // everything else.  Per stage (one block of 16 unknowns) the workgroup splits into
    //   wave 0   the chain: rhs of the block's 16 equations (already updated with all blocks but the previous one),
    //            minus the previous block's coupling (16 x 16, the previous solution still in its registers, read with
    //            v_readlane), then the 16 x 16 triangle -- 32 readlane steps, operands from LDS;
    //   wave 1   stages the NEXT stage's two 16 x 16 coefficient blocks from the factors (global memory) into LDS;
    //   others   apply the PREVIOUS block's solution to all equations beyond the next block;
    // and ONE barrier ends the stage.  (The first version had wave 0 wait for the other waves' dots and the other
    // waves wait for wave 0's triangle, two barriers and a global-memory latency per block on the chain: 0.3 ms per
    // sweep at n = 256, 19 ms for the 128 matrices of the headline search, eight of which are not singular and take
    // all 60 sweeps.)
    // coef(i, j) = coefficient of unknown j in equation i:  M(i, j) = a[rowmap[i] n + j] of the factor, or conj M(j, i)
    // for the transposed systems; LOWER = unit diagonal (L, L^H), else divide by coef(i, i) (U, U^H).
    double2* bA = sm + 3 * (size_t)n + ((size_t)n * sizeof(int) + 15) / 16;  // [2][TB][TB + 1] coupling with the previous block
    double2* bD = bA + 2 * TB * (TB + 1);                                     // [2][TB][TB + 1] the block's own triangle
    const int B = (n + TB - 1) / TB;
    auto readlane_c = [&](cd x, int k) -> cd {
        const long long bx = __double_as_longlong(x.x), by = __double_as_longlong(x.y);
        const unsigned xl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bx, k), xh = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bx >> 32), k);
        const unsigned yl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)by, k), yh = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(by >> 32), k);
        return mk(__longlong_as_double((long long)(((unsigned long long)xh << 32) | xl)), __longlong_as_double((long long)(((unsigned long long)yh << 32) | yl)));
    };
    // x: right-hand side on entry, solution on exit (LDS, indexed by equation = unknown)
    auto solve_tri = [&](auto trans_tag, auto lower_tag, double2* x) {
        constexpr bool TRANS = decltype(trans_tag)::value, LOWER = decltype(lower_tag)::value;
        constexpr bool FWD = LOWER != TRANS;
        auto blk0 = [&](int s) { return (FWD ? s : B - 1 - s) * TB; };   // first unknown of the block of stage s
        auto blkn = [&](int s) { return min(TB, n - blk0(s)); };
        // stage the coefficient blocks of stage s (equations of block s x unknowns of block s - 1 / of block s)
        auto stage_blocks = [&](int s) {
            const int r1 = blk0(s), n1 = blkn(s);
            const int rp = s > 0 ? blk0(s - 1) : 0, np = s > 0 ? blkn(s - 1) : 0;
            double2* dA = bA + (s & 1) * TB * (TB + 1);
            double2* dD = bD + (s & 1) * TB * (TB + 1);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int e = it * 64 + lane;
                // (contiguous in memory along the fast index: columns of a row for the plain systems, rows for the transposed)
                const int l = TRANS ? (e & 15) : (e >> 4), kk = TRANS ? (e >> 4) : (e & 15);
                double2 va = make_double2(0.0, 0.0), vd = make_double2(0.0, 0.0);
                if (l < n1 && kk < n1) {
                    vd = TRANS ? a[(size_t)rowmap[r1 + kk] * n + r1 + l] : a[(size_t)rowmap[r1 + l] * n + r1 + kk];
                    if (TRANS) vd.y = -vd.y;
                }
                if (l < n1 && kk < np) {
                    va = TRANS ? a[(size_t)rowmap[rp + kk] * n + r1 + l] : a[(size_t)rowmap[r1 + l] * n + rp + kk];
                    if (TRANS) va.y = -va.y;
                }
                dA[l * (TB + 1) + kk] = va, dD[l * (TB + 1) + kk] = vd;
            }
        };
        if (wave == 1) stage_blocks(0);
        __syncthreads();
        cd xprev = mk(0.0, 0.0);  // wave 0: the previous block's solution, unknown kk in lane kk
        for (int s = 0; s < B; ++s) {
            const int r0 = blk0(s), nbk = blkn(s);
            if (wave == 0) {
                const double2* dA = bA + (s & 1) * TB * (TB + 1) + (lane & 15) * (TB + 1);
                const double2* dD = bD + (s & 1) * TB * (TB + 1) + (lane & 15) * (TB + 1);
                const bool on = lane < nbk;
                cd sv = on ? mk(x[r0 + lane].x, x[r0 + lane].y) : mk(0.0, 0.0);
                cd ca[TB], cdg[TB];
#pragma unroll
                for (int kk = 0; kk < TB; ++kk) ca[kk] = mk(dA[kk].x, dA[kk].y), cdg[kk] = mk(dD[kk].x, dD[kk].y);
                if (s > 0) {
#pragma unroll
                    for (int kk = 0; kk < TB; ++kk) sv = sv - ca[kk] * readlane_c(xprev, kk);  // (rows beyond the block hold zeros)
                }
#pragma unroll
                for (int q = 0; q < TB; ++q) {
                    const int kk = FWD ? q : TB - 1 - q;
                    if (kk < nbk) {  // uniform
                        if (!LOWER && lane == kk) sv = sv * rcp(cdg[kk]);
                        const cd xk = readlane_c(sv, kk);
                        if (on && (FWD ? lane > kk : lane < kk)) sv = sv - cdg[kk] * xk;
                    }
                }
                if (on) x[r0 + lane] = make_double2(sv.x, sv.y);
                xprev = sv;
            } else if (wave == 1) {
                if (s + 1 < B) stage_blocks(s + 1);
            } else if (s > 0) {
                // apply the previous block's solution (final in x since the last barrier) to the equations beyond this block
                const int rp = blk0(s - 1), np = blkn(s - 1);
                const int lo = FWD ? r0 + nbk : 0, hi = FWD ? n : r0;  // equations [lo, hi)
                if (!TRANS) {
                    // 16 contiguous coefficients per equation: four equations per wave and trip, 16-lane sums
                    const int kk = lane & 15;
                    const cd xk = kk < np ? mk(x[rp + kk].x, x[rp + kk].y) : mk(0.0, 0.0);
                    for (int i = lo + (wave - 2) * 4 + (lane >> 4); i < hi; i += (NW - 2) * 4) {
                        const cd c = kk < np ? ld2(&a[(size_t)rowmap[i] * n + rp + kk]) : mk(0.0, 0.0);
                        const cd pr = c * xk;
                        const double sx = row16_sum(pr.x), sy = row16_sum(pr.y);
                        if (kk == 0) x[i] = make_double2(x[i].x - sx, x[i].y - sy);
                    }
                } else {
                    // coefficients of one unknown are contiguous along the equations: an equation per thread
                    for (int i = lo + (wave - 2) * 64 + lane; i < hi; i += (NW - 2) * 64) {
                        cd acc = mk(x[i].x, x[i].y);
#pragma unroll
                        for (int kk = 0; kk < TB; ++kk)
                            if (kk < np) acc = acc - conj_(ld2(&a[(size_t)rowmap[rp + kk] * n + i])) * mk(x[rp + kk].x, x[rp + kk].y);
                        x[i] = make_double2(acc.x, acc.y);
                    }
                }
            }
            __syncthreads();
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    // z = M^-H b:  U^H w = b, L^H z' = w, z = P^T z'
    auto solve_h = [&]() {
        solve_tri(T_{}, F_{}, v);
        solve_tri(T_{}, T_{}, v);
        for (int r = tid; r < n; r += NT) t[rowmap[r]] = v[r];
        __syncthreads();
        for (int r = tid; r < n; r += NT) v[r] = t[r];
        __syncthreads();
    };
    // y = M^-1 b:  c = P b, L c' = c, U y = c'
    auto solve_n = [&]() {
        for (int r = tid; r < n; r += NT) t[r] = v[rowmap[r]];
        __syncthreads();
        solve_tri(F_{}, T_{}, t);
        solve_tri(F_{}, F_{}, t);
        for (int r = tid; r < n; r += NT) v[r] = t[r];
        __syncthreads();
    };

    normalise();
    int sweeps = 0;
    for (int it = 0; it < P.max_sweeps; ++it) {
        for (int i = tid; i < n; i += NT) o[i] = v[i];  // (each thread reads back only what it wrote)
        solve_h();
        normalise();
        solve_n();
        normalise();
        ++sweeps;
        // |<old, v>| -> 1 when the direction has stopped moving
        cd ov = mk(0.0, 0.0);
        for (int i = tid; i < n; i += NT) ov = ov + conj_(mk(o[i].x, o[i].y)) * mk(v[i].x, v[i].y);
        ov.x = wave_sum(ov.x), ov.y = wave_sum(ov.y);
        if (lane == 0) s_red[wave] = ov.x, s_red[NW + wave] = ov.y;
        __syncthreads();
        double ox = 0.0, oy = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) ox += s_red[w], oy += s_red[NW + w];
        __syncthreads();
        const double overlap = sqrt(ox * ox + oy * oy);
        if (it >= 1 && !(fabs(1.0 - overlap) > 1e-14)) break;  // uniform (a NaN ends the loop too)
    }
    bool finite = true;
    for (int i = tid; i < n; i += NT) {
        out[i] = v[i];
        finite = finite && isfinite(v[i].x) && isfinite(v[i].y);
    }
    const int bad = __syncthreads_or(!finite);
    if (tid == 0) P.info[b] = bad ? -6 /* EMME_ENUMERIC */ : 0;
    (void)sweeps;
}

// Orders above the blocked factorisations' reach (n > 1024): plain right-looking partial-pivot LU in place, one
// workgroup per matrix, the pivot row staged in LDS, rows never moved (row map in LDS, written out as ONE
// snapshot: nb = n).  O(n) barriers per column block of one -- slow, and only there so that every order the
// Newton step accepts has a device nullSpace too.
__global__ __launch_bounds__(NT) void k_lu_unblocked_inplace(int n, double2* A, int* maps, int* info_out) {
    extern __shared__ double2 sm[];
    __shared__ double s_val[NW];
    __shared__ int s_idx[NW];
    __shared__ int s_piv, s_info;
    double2* prow = sm;
    int* rowmap = reinterpret_cast<int*>(sm + n);
    const int b = blockIdx.x;
    double2* a = A + (size_t)b * n * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int r = tid; r < n; r += NT) rowmap[r] = r;
    if (tid == 0) s_info = 0;
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        double best = -1.0;
        int brow = n;
        for (int r = k + tid; r < n; r += NT) {
            const double v = norm2(ld2(&a[(size_t)rowmap[r] * n + k]));
            if (v > best) best = v, brow = r;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(best, off);
            const int oi = __shfl_xor(brow, off);
            if (ov > best || (ov == best && oi < brow)) best = ov, brow = oi;
        }
        if (lane == 0) s_val[wave] = best, s_idx[wave] = brow;
        __syncthreads();
        if (tid == 0) {
            double bv = s_val[0];
            int bi = s_idx[0];
            for (int w = 1; w < NW; ++w)
                if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < bi)) bv = s_val[w], bi = s_idx[w];
            if (!(bv > 0.0) || !isfinite(bv)) {
                if (s_info == 0) s_info = k + 1;
                bi = -1;
            } else {
                const int tmp = rowmap[k];
                rowmap[k] = rowmap[bi], rowmap[bi] = tmp;
            }
            s_piv = bi;
        }
        __syncthreads();
        if (s_piv < 0) break;  // uniform
        const double2* pr = a + (size_t)rowmap[k] * n;
        for (int c = k + tid; c < n; c += NT) prow[c] = pr[c];
        __syncthreads();
        const cd rp = rcp(mk(prow[k].x, prow[k].y));
        for (int r = k + 1 + wave; r < n; r += NW) {
            double2* ar = a + (size_t)rowmap[r] * n;
            const cd f = ld2(&ar[k]) * rp;
            if (lane == 0) ar[k] = make_double2(f.x, f.y);
            for (int c = k + 1 + lane; c < n; c += 64) {
                const cd u = ld2(&ar[c]) - f * mk(prow[c].x, prow[c].y);
                ar[c] = make_double2(u.x, u.y);
            }
        }
        __syncthreads();
    }
    __syncthreads();
    for (int r = tid; r < n; r += NT) maps[(size_t)b * n + r] = rowmap[r];
    if (tid == 0) info_out[b] = s_info;
}

}  // namespace

size_t null_iterate_lds(int n) {
    return (size_t)3 * n * sizeof(double2) + (((size_t)n * sizeof(int) + 15) / 16) * 16 + (size_t)4 * TB * (TB + 1) * sizeof(double2);
}

hipError_t launch_lu_unblocked_inplace(int n, int nbatch, double* A, int* maps, int* info, hipStream_t stream) {
    const size_t lds = (size_t)n * sizeof(double2) + (size_t)n * sizeof(int);
    if (lds > 150 * 1024) return hipErrorNotSupported;
    if (lds > 48 * 1024) {  // (beyond the default dynamic-LDS limit only)
        const hipError_t ea = hipFuncSetAttribute((const void*)k_lu_unblocked_inplace, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return ea;
    }
    hipLaunchKernelGGL(k_lu_unblocked_inplace, dim3(nbatch), dim3(NT), lds, stream, n, (double2*)A, maps, info);
    return hipGetLastError();
}

hipError_t launch_null_iterate(int n, const double* A_lu, const int* rowmaps, int nb, const int* items, int nitems,
                               const int* lu_info, double* vecs, int* info, int max_sweeps, hipStream_t stream) {
    NullArgs P;
    P.n = n, P.nb = nb, P.nblk = (n + nb - 1) / nb;
    P.A = (const double2*)A_lu;
    P.maps = rowmaps;
    P.items = items;
    P.lu_info = lu_info;
    P.vecs = (double2*)vecs;
    P.info = info;
    P.max_sweeps = max_sweeps;
    const size_t lds = null_iterate_lds(n);
    if (lds > 150 * 1024) return hipErrorNotSupported;
    if (lds > 48 * 1024) {  // (beyond the default dynamic-LDS limit only)
        const hipError_t ea = hipFuncSetAttribute((const void*)k_null_iterate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return ea;
    }
    hipLaunchKernelGGL(k_null_iterate, dim3(nitems), dim3(NT), lds, stream, P);
    return hipGetLastError();
}

}  // namespace emme
