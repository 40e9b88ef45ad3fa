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

#include "emme_device.hpp"
#include "launch.hpp"

namespace emme {

namespace {

constexpr int NT = 1024;
constexpr int NW = NT / 64;
constexpr int TB = 16;  // rows per block of the triangular solves

__device__ __forceinline__ cd ld2(const double2* p) {
    const double2 v = *p;
    return mk(v.x, v.y);
}
__device__ __forceinline__ cd conj_(cd a) { return mk(a.x, -a.y); }
__device__ __forceinline__ cd shfl_c(cd v, int src) { return mk(__shfl(v.x, src), __shfl(v.y, src)); }
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
    // the 16 x 16 blocks of the triangles: lane l of wave 0 owns element k0 + l
    // ---- z = M^-H b:  U^H w = b (forward, axpy form), L^H z' = w (backward, axpy form), z = P^T z' -------------
    auto solve_h = [&]() {
        for (int k0 = 0; k0 < n; k0 += TB) {
            const int nbk = min(TB, n - k0);
            if (wave == 0) {
                cd u[TB];  // u[kk] = U(k0 + kk, k0 + lane)
#pragma unroll
                for (int kk = 0; kk < TB; ++kk)
                    u[kk] = (kk < nbk && lane < nbk) ? ld2(&a[(size_t)rowmap[k0 + kk] * n + k0 + lane]) : mk(1.0, 0.0);
                cd w = lane < nbk ? mk(v[k0 + lane].x, v[k0 + lane].y) : mk(0.0, 0.0);
#pragma unroll
                for (int kk = 0; kk < TB; ++kk) {
                    if (kk < nbk) {
                        if (lane == kk) w = w * rcp(conj_(u[kk]));
                        const cd wk = shfl_c(w, kk);
                        if (lane > kk) w = w - conj_(u[kk]) * wk;
                    }
                }
                if (lane < nbk) v[k0 + lane] = make_double2(w.x, w.y);
            }
            __syncthreads();
            for (int r = k0 + nbk + tid; r < n; r += NT) {
                cd acc = mk(v[r].x, v[r].y);
#pragma unroll
                for (int kk = 0; kk < TB; ++kk)
                    if (kk < nbk) acc = acc - conj_(ld2(&a[(size_t)rowmap[k0 + kk] * n + r])) * mk(v[k0 + kk].x, v[k0 + kk].y);
                v[r] = make_double2(acc.x, acc.y);
            }
            __syncthreads();
        }
        for (int k0 = (n - 1) / TB * TB; k0 >= 0; k0 -= TB) {
            const int nbk = min(TB, n - k0);
            if (wave == 0) {
                cd l[TB];  // l[kk] = L(k0 + kk, k0 + lane), lane < kk
#pragma unroll
                for (int kk = 0; kk < TB; ++kk)
                    l[kk] = (kk < nbk && lane < kk) ? ld2(&a[(size_t)rowmap[k0 + kk] * n + k0 + lane]) : mk(0.0, 0.0);
                cd z = lane < nbk ? mk(v[k0 + lane].x, v[k0 + lane].y) : mk(0.0, 0.0);
#pragma unroll
                for (int kk = TB - 1; kk >= 1; --kk) {
                    if (kk < nbk) {
                        const cd zk = shfl_c(z, kk);
                        if (lane < kk) z = z - conj_(l[kk]) * zk;
                    }
                }
                if (lane < nbk) v[k0 + lane] = make_double2(z.x, z.y);
            }
            __syncthreads();
            for (int r = tid; r < k0; r += NT) {
                cd acc = mk(v[r].x, v[r].y);
#pragma unroll
                for (int kk = 0; kk < TB; ++kk)
                    if (kk < nbk) acc = acc - conj_(ld2(&a[(size_t)rowmap[k0 + kk] * n + r])) * mk(v[k0 + kk].x, v[k0 + kk].y);
                v[r] = make_double2(acc.x, acc.y);
            }
            __syncthreads();
        }
        for (int r = tid; r < n; r += NT) t[rowmap[r]] = v[r];
        __syncthreads();
        for (int r = tid; r < n; r += NT) v[r] = t[r];
        __syncthreads();
    };
    // ---- y = M^-1 b:  c = P b, L c' = c (forward, dot form), U y = c' (backward, dot form) --------------------------
    auto solve_n = [&]() {
        for (int r = tid; r < n; r += NT) t[r] = v[rowmap[r]];
        __syncthreads();
        for (int r0 = 0; r0 < n; r0 += TB) {
            const int nbk = min(TB, n - r0);
            if (wave < nbk && r0 > 0) {
                const double2* row = a + (size_t)rowmap[r0 + wave] * n;
                cd s = mk(0.0, 0.0);
                for (int k = lane; k < r0; k += 64) s = s + ld2(&row[k]) * mk(t[k].x, t[k].y);
                s.x = wave_sum(s.x), s.y = wave_sum(s.y);
                if (lane == 0) t[r0 + wave] = make_double2(t[r0 + wave].x - s.x, t[r0 + wave].y - s.y);
            }
            __syncthreads();
            if (wave == 0) {
                cd l[TB];  // l[kk] = L(r0 + lane, r0 + kk), kk < lane
#pragma unroll
                for (int kk = 0; kk < TB; ++kk)
                    l[kk] = (lane < nbk && kk < lane) ? ld2(&a[(size_t)rowmap[r0 + lane] * n + r0 + kk]) : mk(0.0, 0.0);
                cd c = lane < nbk ? mk(t[r0 + lane].x, t[r0 + lane].y) : mk(0.0, 0.0);
#pragma unroll
                for (int kk = 0; kk < TB - 1; ++kk) {
                    const cd ck = shfl_c(c, kk);
                    if (lane > kk) c = c - l[kk] * ck;
                }
                if (lane < nbk) t[r0 + lane] = make_double2(c.x, c.y);
            }
            __syncthreads();
        }
        for (int r0 = (n - 1) / TB * TB; r0 >= 0; r0 -= TB) {
            const int nbk = min(TB, n - r0);
            if (wave < nbk && r0 + nbk < n) {
                const double2* row = a + (size_t)rowmap[r0 + wave] * n;
                cd s = mk(0.0, 0.0);
                for (int k = r0 + nbk + lane; k < n; k += 64) s = s + ld2(&row[k]) * mk(t[k].x, t[k].y);
                s.x = wave_sum(s.x), s.y = wave_sum(s.y);
                if (lane == 0) t[r0 + wave] = make_double2(t[r0 + wave].x - s.x, t[r0 + wave].y - s.y);
            }
            __syncthreads();
            if (wave == 0) {
                cd u[TB];  // u[kk] = U(r0 + lane, r0 + kk), kk >= lane
#pragma unroll
                for (int kk = 0; kk < TB; ++kk)
                    u[kk] = (lane < nbk && kk < nbk && kk >= lane) ? ld2(&a[(size_t)rowmap[r0 + lane] * n + r0 + kk]) : mk(1.0, 0.0);
                cd y = lane < nbk ? mk(t[r0 + lane].x, t[r0 + lane].y) : mk(0.0, 0.0);
#pragma unroll
                for (int kk = TB - 1; kk >= 0; --kk) {
                    if (kk < nbk) {
                        if (lane == kk) y = y * rcp(u[kk]);
                        const cd yk = shfl_c(y, kk);
                        if (lane < kk) y = y - u[kk] * yk;
                    }
                }
                if (lane < nbk) t[r0 + lane] = make_double2(y.x, y.y);
            }
            __syncthreads();
        }
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

size_t null_iterate_lds(int n) { return (size_t)3 * n * sizeof(double2) + (size_t)n * sizeof(int); }

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
