// linstep.hip -- the Newton linear step of the eigenvalue search, batched over omega.
//
// The reference calls LAPACK zsysv with n right-hand sides and then uses only
// trace(M^-1 M') (include/solver.h:134-140).  Here each batch item is one workgroup that
// runs a partial-pivot Gaussian elimination on the augmented system [M | M'] (the
// pivoting / row-update scheme of the reference's own LU, src/solver.cpp:14-85) and then a
// *truncated* back substitution: X(c,c) only needs rows c..n-1 of column c, which cuts the
// triangular solve from n^3/2 to n^3/6 complex multiply-adds.
//
// Data stay in global memory (L2 / Infinity-Cache resident for the batch); the pivot row
// and the current solution row are staged in LDS each step so that the rank-1 update reads
// them at LDS rate and streams its own rows with coalesced 16-byte accesses.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "emme_device.hpp"
#include "launch.hpp"

namespace emme {

namespace {

constexpr int LU_THREADS = 1024;
constexpr int LU_WAVES = LU_THREADS / 64;

__device__ __forceinline__ cd ld(const double2* p) {
    const double2 v = *p;
    return mk(v.x, v.y);
}
__device__ __forceinline__ void st(double2* p, cd v) { *p = make_double2(v.x, v.y); }

__global__ __launch_bounds__(LU_THREADS) void k_trace_solve(int n, double2* A, double2* B,
                                                           const int* active, double2* tr_out,
                                                           int* info_out) {
    extern __shared__ double2 lds_row[];  // 2n entries: pivot row of [A | B], later x row
    __shared__ double s_val[LU_WAVES];
    __shared__ int s_idx[LU_WAVES];
    __shared__ int s_piv;
    __shared__ int s_info;
    __shared__ double s_tr[2];

    const int b = blockIdx.x;
    if (active && active[b] == 0) return;
    double2* a = A + (size_t)b * n * n;
    double2* bb = B + (size_t)b * n * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (tid == 0) s_info = 0, s_tr[0] = 0.0, s_tr[1] = 0.0;
    __syncthreads();

    // ---- forward elimination with partial pivoting on [A | B] --------------------
    for (int k = 0; k < n; ++k) {
        // pivot = first row of maximal modulus in column k, rows k..n-1
        double best = -1.0;
        int brow = n;
        for (int r = k + tid; r < n; r += LU_THREADS) {
            const double v = norm2(ld(&a[(size_t)r * n + k]));
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
            for (int w = 1; w < LU_WAVES; ++w)
                if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < bi)) bv = s_val[w], bi = s_idx[w];
            // exactly singular (or NaN) column: LAPACK-style info = k+1, first occurrence
            if (!(bv > 0.0)) {
                if (s_info == 0) s_info = k + 1;
                bi = -1;
            }
            s_piv = bi;
        }
        __syncthreads();
        const int piv = s_piv;
        if (piv < 0) continue;  // uniform: nothing to eliminate with

        // swap rows k <-> piv over the columns still in play and stage the pivot row in LDS
        for (int c = tid; c < 2 * n; c += LU_THREADS) {
            if (c < n) {
                if (c < k) continue;
                const cd pv = ld(&a[(size_t)piv * n + c]);
                if (piv != k) {
                    const cd top = ld(&a[(size_t)k * n + c]);
                    st(&a[(size_t)k * n + c], pv);
                    st(&a[(size_t)piv * n + c], top);
                }
                lds_row[c] = make_double2(pv.x, pv.y);
            } else {
                const int cc = c - n;
                const cd pv = ld(&bb[(size_t)piv * n + cc]);
                if (piv != k) {
                    const cd top = ld(&bb[(size_t)k * n + cc]);
                    st(&bb[(size_t)k * n + cc], pv);
                    st(&bb[(size_t)piv * n + cc], top);
                }
                lds_row[c] = make_double2(pv.x, pv.y);
            }
        }
        __syncthreads();

        const cd rp = rcp(mk(lds_row[k].x, lds_row[k].y));
        for (int r = k + 1 + wave; r < n; r += LU_WAVES) {
            const cd f = ld(&a[(size_t)r * n + k]) * rp;  // factor = A(r,k)/A(k,k)
            if (f.x == 0.0 && f.y == 0.0) continue;
            double2* ar = a + (size_t)r * n;
            double2* br = bb + (size_t)r * n;
            for (int c = k + 1 + lane; c < n; c += 64) {
                const double2 p = lds_row[c];
                st(&ar[c], ld(&ar[c]) - f * mk(p.x, p.y));
            }
            for (int c = lane; c < n; c += 64) {
                const double2 p = lds_row[n + c];
                st(&br[c], ld(&br[c]) - f * mk(p.x, p.y));
            }
        }
        __syncthreads();
    }

    if (s_info != 0) {
        if (tid == 0) {
            tr_out[b] = make_double2(__builtin_nan(""), __builtin_nan(""));
            info_out[b] = s_info;
        }
        return;
    }

    // ---- truncated back substitution: X(c,c) needs only rows c..n-1 of column c ----
    for (int r = n - 1; r >= 0; --r) {
        const cd ru = rcp(ld(&a[(size_t)r * n + r]));
        for (int c = tid; c <= r; c += LU_THREADS) {
            const cd x = ld(&bb[(size_t)r * n + c]) * ru;
            lds_row[c] = make_double2(x.x, x.y);
            if (c == r) s_tr[0] += x.x, s_tr[1] += x.y;  // one thread per step
        }
        __syncthreads();
        for (int rr = wave; rr < r; rr += LU_WAVES) {
            const cd u = ld(&a[(size_t)rr * n + r]);
            double2* brr = bb + (size_t)rr * n;
            for (int c = lane; c <= rr; c += 64) {
                const double2 x = lds_row[c];
                st(&brr[c], ld(&brr[c]) - u * mk(x.x, x.y));
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        tr_out[b] = make_double2(s_tr[0], s_tr[1]);
        info_out[b] = 0;
    }
}

__global__ void k_newton_update(int nbatch, const double2* tr, double2* omega, double2* domega,
                                int* active, int* iters, const int* info, double tol,
                                double2* iterates, int iter_index, int iter_stride, double2* pub_omega,
                                int* info_w, const int* lost) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbatch) return;
    if (active && active[b] == 0) {
        if (pub_omega) pub_omega[b] = omega[b];
        return;
    }
    // d_eigen_value = -1 / trace; eigen_value += d (include/solver.h:139-140)
    const cd t = mk(tr[b].x, tr[b].y);
    cd d = -rcp(t);
    cd w = mk(omega[b].x, omega[b].y);
    w = w + d;
    // a chain whose matrix holds a non-finite integral (or met the quadrature caps) is lost: the fill stopped
    // working on that matrix when the flag went up, so what the factorisation saw is unspecified -- the chain
    // retires here, at the step the reference's zsysv fails at (include/solver.h:142-153), with EMME_ENUMERIC
    const bool is_lost = lost && active && lost[b] != 0;
    if (is_lost) {
        d = mk(__builtin_nan(""), __builtin_nan(""));
        w = d;
        info_w[b] = -6;
    }
    omega[b] = make_double2(w.x, w.y);
    domega[b] = make_double2(d.x, d.y);
    if (iters) iters[b] += 1;
    if (iterates) iterates[(size_t)b * iter_stride + iter_index] = make_double2(w.x, w.y);
    // the host needs the new omegas (contour classes) before it launches the fill: written straight
    // into its pinned memory instead of a device-to-host copy of their own
    if (pub_omega) pub_omega[b] = make_double2(w.x, w.y);
    if (active) {
        // stop when |d| < |tol * omega| (src/main.cpp:53-56) or on a failed factorisation.
        // The reassembly at the new omega still happens this step (solver.h:157), so the
        // flag only takes effect from the next step on: encoded as 2 = "last step".
        const bool conv = hypot(d.x, d.y) < hypot(tol * w.x, tol * w.y);
        const bool fail = is_lost || (info && info[b] != 0);
        if (fail)
            active[b] = 0;
        else if (conv || !(isfinite(w.x) && isfinite(w.y)))
            active[b] = 2;
    }
}

// active == 2 ("converged, last step done") -> 0; and what the host reads after the next step's
// synchronisation -- the active flags, the interval counters and the number of integrals the fill
// deferred -- goes straight into its pinned memory (one kernel instead of three small copies)
__global__ void k_retire(int nbatch, int* active, int* pub_active, const unsigned long long* intervals,
                         unsigned long long* pub_intervals, const unsigned int* deferred, unsigned int* pub_deferred,
                         unsigned int* overflow, unsigned int* pub_overflow) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0 && pub_deferred) *pub_deferred = deferred ? *deferred : 0u;
    if (b >= nbatch) return;
    if (overflow) {  // integrals of this item that left the dense fill because a level list was full: published, reset
        if (pub_overflow) pub_overflow[b] = overflow[b];
        overflow[b] = 0u;
    }
    int a = active[b];
    if (a == 2) active[b] = a = 0;
    if (pub_active) pub_active[b] = a;
    if (pub_intervals) pub_intervals[b] = intervals[b];
}

// dst1[0..n1) = src[0..n1), dst2[0..n2) = src[n1..n1+n2): the small per-launch lists (omega order,
// chunks, live matrices) go from the host's pinned staging block to device memory in one kernel
__global__ void k_stage_ints(const int* src, int* dst1, int n1, int* dst2, int n2) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n1 + n2; k += gridDim.x * blockDim.x) {
        const int v = src[k];
        if (k < n1)
            dst1[k] = v;
        else
            dst2[k - n1] = v;
    }
}

__global__ void k_secant(size_t nn, const double2* M, const double2* Mold, const double2* domega,
                         const int* active, double2* Mp) {
    const int b = blockIdx.y;
    if (active && active[b] == 0) return;
    const cd rdw = rcp(mk(domega[b].x, domega[b].y));
    const size_t base = (size_t)b * nn;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < nn;
         k += (size_t)gridDim.x * blockDim.x) {
        const double2 m = M[base + k], o = Mold[base + k];
        const cd d = mk(m.x - o.x, m.y - o.y) * rdw;
        Mp[base + k] = make_double2(d.x, d.y);
    }
}

// dst1[b] = src[b] (and dst2[b] = src[b] if dst2) for the active matrices of the batch only: the
// Newton loop's "previous matrix" and LU work copies, which shrink with the live chains
__global__ __launch_bounds__(256) void k_copy_active(long elems, const double2* src, double2* dst1,
                                                     double2* dst2, const int* active) {
    const int b = blockIdx.y;
    if (active && active[b] == 0) return;
    const double2* s = src + (size_t)b * elems;
    double2* d1 = dst1 + (size_t)b * elems;
    double2* d2 = dst2 ? dst2 + (size_t)b * elems : nullptr;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < elems; e += (long)gridDim.x * blockDim.x) {
        const double2 v = s[e];
        d1[e] = v;
        if (d2) d2[e] = v;
    }
}

// The secant of a Newton step for the active matrices, fused with the two copies that follow it:
//   Mp = (M - Mold) / domega  (include/solver.h:157),  Mold = M,  work = M (if given)
// in ONE coalesced pass.  The fill kernels used to do the secant entry by entry where they store their integrals
// -- lanes are (pair, omega), so every Mold read and Mp write was a 16-byte access with a line to itself: at
// dim 512 (BASELINE configs[3]) those scattered accesses were 2.0 of a fill launch's 2.9 ms.
__global__ __launch_bounds__(256) void k_secant_copy(long elems, const double2* M, double2* Mold, double2* work,
                                                     double2* Mp, const double2* domega, const int* active) {
    const int b = blockIdx.y;
    if (active && active[b] == 0) return;
    const cd rdw = rcp(mk(domega[b].x, domega[b].y));
    const double2* s = M + (size_t)b * elems;
    double2* o = Mold + (size_t)b * elems;
    double2* p = Mp + (size_t)b * elems;
    double2* w = work ? work + (size_t)b * elems : nullptr;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < elems; e += (long)gridDim.x * blockDim.x) {
        const double2 v = s[e], vo = o[e];
        const cd d = (mk(v.x, v.y) - mk(vo.x, vo.y)) * rdw;
        p[e] = make_double2(d.x, d.y);
        o[e] = v;
        if (w) w[e] = v;
    }
}

// The same pass on SYMMETRIC matrices (include/solver.h:453, 472-509: every entry is written with its mirror, so M and
// M_old are symmetric bit for bit): only the tiles on and above the diagonal of M and M_old are read, the secant and the
// work copy are written with their mirror images (transposed through LDS, both coalesced), and M_old is kept in its
// upper triangle only (nothing else reads it).  2 x 1/2 matrices in, 2 1/2 out instead of 2 in, 3 out.
constexpr int ST = 32;  // tile edge
__global__ __launch_bounds__(256) void k_secant_copy_sym(int n, const double2* M, double2* Mold, double2* work,
                                                         double2* Mp, const double2* domega, const int* active) {
    const int b = blockIdx.y;
    if (active && active[b] == 0) return;
    __shared__ double2 s_v[ST][ST + 1], s_d[ST][ST + 1];
    // blockIdx.x -> tile (I, J), I <= J, of the nt x nt tile grid: row I holds nt - I tiles
    const int nt = (n + ST - 1) / ST;
    int I = 0, rem = blockIdx.x;
    while (rem >= nt - I) rem -= nt - I, ++I;
    const int J = I + rem;
    const cd rdw = rcp(mk(domega[b].x, domega[b].y));
    const size_t base = (size_t)b * n * n;
    const int tx = threadIdx.x & (ST - 1), ty = threadIdx.x / ST;  // 32 x 8 threads, 4 rows each
    for (int rr = ty; rr < ST; rr += 256 / ST) {
        const int r = I * ST + rr, c = J * ST + tx;
        double2 v = make_double2(0.0, 0.0), d2 = v;
        if (r < n && c < n && (I != J || rr <= tx)) {
            const size_t idx = base + (size_t)r * n + c;
            v = M[idx];
            const double2 vo = Mold[idx];
            const cd d = (mk(v.x, v.y) - mk(vo.x, vo.y)) * rdw;
            d2 = make_double2(d.x, d.y);
            Mold[idx] = v;
        }
        s_v[rr][tx] = v, s_d[rr][tx] = d2;
    }
    __syncthreads();
    // tile (I, J) itself (a diagonal tile takes its lower half from the transposed upper half) ...
    for (int rr = ty; rr < ST; rr += 256 / ST) {
        const int r = I * ST + rr, c = J * ST + tx;
        if (r < n && c < n) {
            const bool lower = I == J && rr > tx;
            const size_t idx = base + (size_t)r * n + c;
            Mp[idx] = lower ? s_d[tx][rr] : s_d[rr][tx];
            if (work) work[idx] = lower ? s_v[tx][rr] : s_v[rr][tx];
        }
    }
    // ... and its mirror image (J, I)
    if (I != J) {
        for (int rr = ty; rr < ST; rr += 256 / ST) {
            const int r = J * ST + rr, c = I * ST + tx;
            if (r < n && c < n) {
                const size_t idx = base + (size_t)r * n + c;
                Mp[idx] = s_d[tx][rr];
                if (work) work[idx] = s_v[tx][rr];
            }
        }
    }
}

}  // namespace

hipError_t launch_secant_copy_sym(int n, int nbatch, const double* M, double* Mold, double* work, double* Mp,
                                  const double* domega, const int* active, hipStream_t stream) {
    const int nt = (n + ST - 1) / ST;
    hipLaunchKernelGGL(k_secant_copy_sym, dim3((unsigned)(nt * (nt + 1) / 2), nbatch), dim3(256), 0, stream, n,
                       (const double2*)M, (double2*)Mold, (double2*)work, (double2*)Mp, (const double2*)domega, active);
    return hipGetLastError();
}

hipError_t launch_secant_copy(int n, int nbatch, const double* M, double* Mold, double* work, double* Mp,
                              const double* domega, const int* active, hipStream_t stream) {
    const long elems = (long)n * n;
    const unsigned gx = (unsigned)std::min<long>(64, (elems + 255) / 256);
    hipLaunchKernelGGL(k_secant_copy, dim3(gx, nbatch), dim3(256), 0, stream, elems, (const double2*)M, (double2*)Mold,
                       (double2*)work, (double2*)Mp, (const double2*)domega, active);
    return hipGetLastError();
}

hipError_t launch_trace_solve(int n, int nbatch, double* A, double* B, const int* active,
                              double* tr, int* info, hipStream_t stream) {
    const size_t lds = (size_t)2 * n * sizeof(double2);
    hipLaunchKernelGGL(k_trace_solve, dim3(nbatch), dim3(LU_THREADS), lds, stream, n,
                       (double2*)A, (double2*)B, active, (double2*)tr, info);
    return hipGetLastError();
}

hipError_t launch_newton_update(int nbatch, const double* tr, double* omega, double* domega,
                                int* active, int* iters, const int* info, double tol,
                                double* iterates, int iter_index, int iter_stride,
                                hipStream_t stream, double* pub_omega, const int* lost) {
    hipLaunchKernelGGL(k_newton_update, dim3((nbatch + 63) / 64), dim3(64), 0, stream, nbatch,
                       (const double2*)tr, (double2*)omega, (double2*)domega, active, iters, info,
                       tol, (double2*)iterates, iter_index, iter_stride, (double2*)pub_omega,
                       const_cast<int*>(info), lost);
    return hipGetLastError();
}

hipError_t launch_retire(int nbatch, int* active, hipStream_t stream, int* pub_active,
                         const unsigned long long* intervals, unsigned long long* pub_intervals,
                         const unsigned int* deferred, unsigned int* pub_deferred, unsigned int* overflow,
                         unsigned int* pub_overflow) {
    hipLaunchKernelGGL(k_retire, dim3((nbatch + 63) / 64), dim3(64), 0, stream, nbatch, active, pub_active,
                       intervals, pub_intervals, deferred, pub_deferred, overflow, pub_overflow);
    return hipGetLastError();
}

hipError_t launch_stage_ints(const int* src_pinned, int* dst1, int n1, int* dst2, int n2, hipStream_t stream) {
    if (n1 + n2 <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_stage_ints, dim3((n1 + n2 + 255) / 256), dim3(256), 0, stream, src_pinned, dst1, n1, dst2, n2);
    return hipGetLastError();
}

hipError_t launch_secant(int nbatch, size_t nn, const double* M, const double* Mold,
                         const double* domega, const int* active, double* Mp,
                         hipStream_t stream) {
    unsigned gx = (unsigned)((nn + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_secant, dim3(gx, nbatch), dim3(256), 0, stream, nn, (const double2*)M,
                       (const double2*)Mold, (const double2*)domega, active, (double2*)Mp);
    return hipGetLastError();
}

hipError_t launch_copy_active(int n, int nbatch, const double* src, double* dst1, double* dst2,
                              const int* active, hipStream_t stream) {
    const long elems = (long)n * n;
    const unsigned gx = (unsigned)std::min<long>(64, (elems + 255) / 256);
    hipLaunchKernelGGL(k_copy_active, dim3(gx, nbatch), dim3(256), 0, stream, elems, (const double2*)src,
                       (double2*)dst1, (double2*)dst2, active);
    return hipGetLastError();
}

}  // namespace emme
