// linstep_qr.hip -- the QR-secant form of the Newton linear step
// (reference include/solver.h:210-383, `iteration_method` != "TraceSecant").
//
// The reference calls LAPACK: zgeqp3 (A P = Q R, column pivoting), ztrtrs (R11 y = r12),
// builds v = P [-y; 1], t = M' v, zunmqr (t <- Q^H t) and sets d omega = -R_nn / t_n.
// Here one workgroup (1024 threads; 512 above n = 512) does all of it for one batch item:
//
//   storage   the input is the TRANSPOSE of M (the reference makes the same copy,
//             include/solver.h:240-243), so column j of the factored matrix is the
//             contiguous memory row j.  Columns are never swapped: perm[] maps pivot
//             position -> memory row, exactly the bookkeeping zlaqp2 does with jpvt.
//   pivoting  zlaqp2's rule: pivot = first maximum of the partial column norms vn1;
//             after every reflector vn1 is down-dated with the LAPACK safeguard
//             (recompute when the estimate lost half its digits, tol3z = sqrt(eps)).
//   reflector zlarfg's (beta, tau, v) for the pivot column; v is kept in LDS and, scaled,
//             below the diagonal in memory (needed again for Q^H t).
//   update    every remaining column is one contiguous row: a wave loads it once, takes
//             s = v^H a with a butterfly, applies a -= conj(tau) s v and stores it -- the
//             trailing matrix is streamed exactly once per reflector (this is a BLAS-2
//             factorisation, bounded by Infinity-Cache/L2 bandwidth: n^3/3 * 32 B per item).
//   tail      wave 0 keeps the right-hand side / t in registers and runs the triangular
//             solve and the n reflector applications with shuffles only (no barriers);
//             M' v is a row-per-wave matrix-vector product in between.
//
// Output is t_n / R_nn, i.e. the value whose negative reciprocal is d omega, so the caller's
// update kernel is the same as for the trace form.
#include <hip/hip_runtime.h>

#include "emme_device.hpp"
#include "launch.hpp"

namespace emme {

namespace {

constexpr int QW_MAX = 16;  // waves per workgroup, at most

__device__ __forceinline__ cd ldq(const double2* p) {
    const double2 v = *p;
    return mk(v.x, v.y);
}
__device__ __forceinline__ void stq(double2* p, cd v) { *p = make_double2(v.x, v.y); }

__device__ __forceinline__ double wave_sum(double v) { return wave64_sum(v); }

// plain complex quotient (Smith-free textbook form is what the BLAS kernels use)
__device__ __forceinline__ cd cquot(cd a, cd b) {
    const double d = b.x * b.x + b.y * b.y;
    return mk((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}

struct QrShared {
    double s_val[QW_MAX];
    int s_idx[QW_MAX];
    double s_sum[QW_MAX];
    double alpha[2];
    int info;
};

// element i = lane + 64 q of a register-resident vector, broadcast to the whole wave
template <int PER>
__device__ __forceinline__ cd pick(const cd (&v)[PER], int i) {
    const int q = i >> 6;
    cd r = mk(0.0, 0.0);
#pragma unroll
    for (int qq = 0; qq < PER; ++qq)
        if (qq == q) r = v[qq];
    return mk(__shfl(r.x, i & 63), __shfl(r.y, i & 63));
}

// PER = ceil(n_max / 64) elements of a column per lane; RB columns per wave in flight;
// QT threads (the n_max = 1024 variant runs 512 threads so that a wave may hold 256 VGPRs)
template <int PER, int RB, int QT>
__global__ __launch_bounds__(QT) void k_qr_secant(int n, double2* Wt, const double2* Mp,
                                                  const int* active, double2* tr_out,
                                                  int* info_out) {
    // dynamic LDS: vn1[n] | vn2[n] (double) | hv[n] | tau[n] | vf[n] | tv[n] (double2) | perm[n] (int)
    extern __shared__ double2 qlds[];
    __shared__ QrShared sh;

    constexpr int QW = QT / 64;
    constexpr int EPT = (64 * PER + QT - 1) / QT;  // column elements per thread in the panel steps
    constexpr bool PF = PER <= 8;  // tail loops prefetch the next column while registers allow
    const int b = blockIdx.x;
    if (active && active[b] == 0) return;
    double2* w = Wt + (size_t)b * n * n;
    const double2* mp = Mp + (size_t)b * n * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    double* vn1 = reinterpret_cast<double*>(qlds);
    double* vn2 = vn1 + n;
    double2* hv = qlds + n;  // 2n doubles = n double2
    double2* tau = hv + n;
    double2* vf = tau + n;
    double2* tv = vf + n;
    int* perm = reinterpret_cast<int*>(tv + n);

    const double tol3z = 1.0536712127723509e-08;  // sqrt(dlamch('Epsilon')) = sqrt(2^-53)

    // ---- initial column norms (zgeqp3: vn1 = vn2 = dznrm2 of every column) -----------------
    for (int r = wave; r < n; r += QW) {
        double ss = 0.0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = lane + 64 * q;
            if (i < n) ss += norm2(ldq(&w[(size_t)r * n + i]));
        }
        ss = wave_sum(ss);
        if (lane == 0) vn1[r] = vn2[r] = sqrt(ss);
    }
    for (int r = tid; r < n; r += QT) perm[r] = r;
    if (tid == 0) sh.info = 0;
    __syncthreads();

    // =================== Householder QR with column pivoting ==============================
    for (int k = 0; k < n; ++k) {
        const int nrem = n - k;
        // ---- pivot: first maximum of vn1[k..n) ----------------------------------------------
        {
            double best = -1.0;
            int bidx = n;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int t = tid + e * QT;
                if (t < nrem) {
                    const double v = vn1[k + t];
                    if (v > best) best = v, bidx = k + t;
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double ov = __shfl_xor(best, off);
                const int oi = __shfl_xor(bidx, off);
                if (ov > best || (ov == best && oi < bidx)) best = ov, bidx = oi;
            }
            if (lane == 0) sh.s_val[wave] = best, sh.s_idx[wave] = bidx;
        }
        __syncthreads();  // A
        int pvt;
        {
            double bv = sh.s_val[0];
            int bi = sh.s_idx[0];
#pragma unroll
            for (int ww = 1; ww < QW; ++ww) {
                const double ov = sh.s_val[ww];
                const int oi = sh.s_idx[ww];
                if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
            }
            pvt = (bi >= k && bi < n) ? bi : k;  // all-NaN norms: keep the column in place
        }
        const int prow = perm[pvt];
        const int krow = perm[k];
        const double vn1k = vn1[k], vn2k = vn2[k];
        // ---- reflector for the pivot column (zlarfg) --------------------------------------------
        cd x[EPT];
        {
            double ss = 0.0;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int t = tid + e * QT;
                x[e] = mk(0.0, 0.0);
                if (t < nrem) {
                    x[e] = ldq(&w[(size_t)prow * n + k + t]);
                    if (t > 0) ss += norm2(x[e]);
                }
            }
            ss = wave_sum(ss);
            if (lane == 0) sh.s_sum[wave] = ss;
            if (tid == 0) sh.alpha[0] = x[0].x, sh.alpha[1] = x[0].y;
        }
        __syncthreads();  // B  (everyone has read perm/vn1 of the two positions)
        if (tid == 0) {
            perm[k] = prow;
            perm[pvt] = krow;
            vn1[pvt] = vn1k;
            vn2[pvt] = vn2k;
        }
        double xn2 = 0.0;
#pragma unroll
        for (int ww = 0; ww < QW; ++ww) xn2 += sh.s_sum[ww];
        const cd alpha = mk(sh.alpha[0], sh.alpha[1]);
        cd tk, scale;
        double beta;
        if (xn2 == 0.0 && alpha.y == 0.0) {
            tk = mk(0.0, 0.0);
            scale = mk(1.0, 0.0);
            beta = alpha.x;
        } else {
            const double nrm = sqrt(fma(alpha.x, alpha.x, fma(alpha.y, alpha.y, xn2)));
            beta = alpha.x >= 0.0 ? -nrm : nrm;
            tk = mk((beta - alpha.x) / beta, -alpha.y / beta);
            scale = cquot(mk(1.0, 0.0), mk(alpha.x - beta, alpha.y));
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int t = tid + e * QT;
            if (t < nrem) {
                const cd hvi = t == 0 ? mk(1.0, 0.0) : x[e] * scale;
                hv[k + t] = make_double2(hvi.x, hvi.y);
                // R(k,k) = beta on the diagonal, the scaled reflector below it
                stq(&w[(size_t)prow * n + k + t], t == 0 ? mk(beta, 0.0) : hvi);
            }
        }
        if (tid == 0) {
            tau[k] = make_double2(tk.x, tk.y);
            // ztrtrs: an exactly zero diagonal entry of R11 (include/solver.h:309-316)
            if (k < n - 1 && beta == 0.0 && sh.info == 0) sh.info = k + 1;
        }
        __syncthreads();  // C
        // ---- apply H^H to the remaining columns, down-date their norms --------------------------
        const cd ctau = conj(tk);
        const int q0 = k >> 6;
        cd hvr[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = lane + 64 * q;
            hvr[q] = (q >= q0 && i >= k && i < n) ? mk(hv[i].x, hv[i].y) : mk(0.0, 0.0);
        }
        for (int c0 = k + 1 + wave * RB; c0 < n; c0 += QW * RB) {
            cd a[RB][PER];
            int row[RB];
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int c = c0 + j;
                row[j] = c < n ? perm[c] : -1;
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const int i = lane + 64 * q;
                    a[j][q] = (row[j] >= 0 && q >= q0 && i >= k && i < n)
                                  ? ldq(&w[(size_t)row[j] * n + i])
                                  : mk(0.0, 0.0);
                }
            }
            double sr[RB], si[RB];
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                sr[j] = 0.0, si[j] = 0.0;
#pragma unroll
                for (int q = 0; q < PER; ++q) {  // conj(v) * a
                    sr[j] = fma(hvr[q].x, a[j][q].x, fma(hvr[q].y, a[j][q].y, sr[j]));
                    si[j] = fma(hvr[q].x, a[j][q].y, fma(-hvr[q].y, a[j][q].x, si[j]));
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    sr[j] += __shfl_xor(sr[j], off);
                    si[j] += __shfl_xor(si[j], off);
                }
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                if (row[j] < 0) continue;  // wave-uniform
                const int c = c0 + j;
                const cd f = ctau * mk(sr[j], si[j]);
                double rest = 0.0;  // |a(k+1.., c)|^2 after the update
                cd akc = mk(0.0, 0.0);
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const int i = lane + 64 * q;
                    if (q >= q0 && i >= k && i < n) {
                        const cd nv = a[j][q] - f * hvr[q];
                        stq(&w[(size_t)row[j] * n + i], nv);
                        if (i == k) akc = nv;
                        else rest += norm2(nv);
                    }
                }
                // zlaqp2 norm down-date (wave-uniform decisions; lane 0 stores)
                const double v1 = vn1[c], v2 = vn2[c];
                if (v1 != 0.0) {
                    const double ak = sqrt(wave_sum(norm2(akc)));  // only one lane is non-zero
                    double temp = ak / v1;
                    temp = fmax(0.0, (1.0 + temp) * (1.0 - temp));
                    const double ratio = v1 / v2;
                    const double temp2 = temp * (ratio * ratio);
                    if (temp2 <= tol3z) {
                        const double nr = sqrt(wave_sum(rest));
                        if (lane == 0) vn1[c] = nr, vn2[c] = nr;
                    } else if (lane == 0) {
                        vn1[c] = v1 * sqrt(temp);
                    }
                }
            }
        }
        __syncthreads();  // D
    }

    const int info = sh.info;
    if (info != 0) {
        if (tid == 0) {
            tr_out[b] = make_double2(__builtin_nan(""), __builtin_nan(""));
            info_out[b] = info;
        }
        return;
    }

    // =================== R11 y = r12 (ztrtrs), wave 0, right-hand side in registers ==========
    if (wave == 0) {
        const int last = perm[n - 1];
        cd rhs[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = lane + 64 * q;
            rhs[q] = i < n - 1 ? ldq(&w[(size_t)last * n + i]) : mk(0.0, 0.0);
        }
        cd col[PER], nxt[PER];
        auto load_col = [&](cd (&dst)[PER], int c) {
            const int r = perm[c];
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int i = lane + 64 * q;
                dst[q] = (q <= (c >> 6) && i <= c) ? ldq(&w[(size_t)r * n + i]) : mk(0.0, 0.0);
            }
        };
        if (PF && n >= 2) load_col(col, n - 2);
        for (int c = n - 2; c >= 0; --c) {
            if (!PF) load_col(col, c);
            else if (c > 0) load_col(nxt, c - 1);
            const cd diag = pick<PER>(col, c);
            const cd rc = pick<PER>(rhs, c);
            const cd y = cquot(rc, diag);
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int i = lane + 64 * q;
                if (i < c) rhs[q] = rhs[q] - y * col[q];
                else if (i == c) rhs[q] = y;
            }
            if (PF) {
#pragma unroll
                for (int q = 0; q < PER; ++q) col[q] = nxt[q];
            }
        }
        // v = P [-y; 1]  (include/solver.h:329-333)
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = lane + 64 * q;
            if (i < n - 1) vf[perm[i]] = make_double2(-rhs[q].x, -rhs[q].y);
            else if (i == n - 1) vf[perm[i]] = make_double2(1.0, 0.0);
        }
    }
    __syncthreads();

    // =================== t = M' v  (include/solver.h:340-346), a row per wave ================
    {
        cd vr[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = lane + 64 * q;
            vr[q] = i < n ? mk(vf[i].x, vf[i].y) : mk(0.0, 0.0);
        }
        for (int r0 = wave * RB; r0 < n; r0 += QW * RB) {
            double sr[RB], si[RB];
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                sr[j] = 0.0, si[j] = 0.0;
                const int r = r0 + j;
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const int i = lane + 64 * q;
                    if (r < n && i < n) {
                        const cd m = ldq(&mp[(size_t)r * n + i]);
                        sr[j] = fma(m.x, vr[q].x, fma(-m.y, vr[q].y, sr[j]));
                        si[j] = fma(m.x, vr[q].y, fma(m.y, vr[q].x, si[j]));
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const double tr_ = wave_sum(sr[j]), ti_ = wave_sum(si[j]);
                if (lane == 0 && r0 + j < n) tv[r0 + j] = make_double2(tr_, ti_);
            }
        }
    }
    __syncthreads();

    // =================== t <- Q^H t (zunmqr 'L','C'), wave 0 =================================
    if (wave == 0) {
        cd t[PER], vcur[PER], vnxt[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = lane + 64 * q;
            t[q] = i < n ? mk(tv[i].x, tv[i].y) : mk(0.0, 0.0);
        }
        auto load_v = [&](cd (&dst)[PER], int kk) {
            const int r = perm[kk];
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int i = lane + 64 * q;
                dst[q] = (i > kk && i < n) ? ldq(&w[(size_t)r * n + i])
                                           : (i == kk ? mk(1.0, 0.0) : mk(0.0, 0.0));
            }
        };
        if (PF) load_v(vcur, 0);
        for (int kk = 0; kk < n; ++kk) {
            if (!PF) load_v(vcur, kk);
            else if (kk + 1 < n) load_v(vnxt, kk + 1);
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                sr = fma(vcur[q].x, t[q].x, fma(vcur[q].y, t[q].y, sr));
                si = fma(vcur[q].x, t[q].y, fma(-vcur[q].y, t[q].x, si));
            }
            sr = wave_sum(sr), si = wave_sum(si);
            const cd f = conj(mk(tau[kk].x, tau[kk].y)) * mk(sr, si);
#pragma unroll
            for (int q = 0; q < PER; ++q) t[q] = t[q] - f * vcur[q];
            if (PF) {
#pragma unroll
                for (int q = 0; q < PER; ++q) vcur[q] = vnxt[q];
            }
        }
        const cd tn = pick<PER>(t, n - 1);
        const double rnn = w[(size_t)perm[n - 1] * n + (n - 1)].x;
        if (lane == 0) {
            // d omega = -R_nn / t_n (include/solver.h:370)  <=>  "trace" = t_n / R_nn
            tr_out[b] = make_double2(tn.x / rnn, tn.y / rnn);
            info_out[b] = 0;
        }
    }
}

// out[b][j][i] = in[b][i][j]
__global__ __launch_bounds__(256) void k_transpose(int n, const double2* in, double2* out,
                                                   const int* active) {
    __shared__ double2 tile[16][17];
    const int b = blockIdx.z;
    if (active && active[b] == 0) return;
    const double2* src = in + (size_t)b * n * n;
    double2* dst = out + (size_t)b * n * n;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
    if (i < n && j < n) tile[ty][tx] = src[(size_t)i * n + j];
    __syncthreads();
    const int oj = blockIdx.x * 16 + ty, oi = blockIdx.y * 16 + tx;
    if (oj < n && oi < n) dst[(size_t)oj * n + oi] = tile[tx][ty];
}

template <int PER, int RB, int QT>
hipError_t launch_qr(int n, int nbatch, double* Wt, const double* Mp, const int* active,
                     double* tr, int* info, hipStream_t stream) {
    const size_t lds = qr_secant_lds(n);
    static thread_local int attr_dev = -1;  // (function attributes are per device)
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    if (attr_dev != cur_dev) {
        (void)hipFuncSetAttribute((const void*)k_qr_secant<PER, RB, QT>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        attr_dev = cur_dev;
    }
    hipLaunchKernelGGL((k_qr_secant<PER, RB, QT>), dim3(nbatch), dim3(QT), lds, stream, n,
                       (double2*)Wt, (const double2*)Mp, active, (double2*)tr, info);
    return hipGetLastError();
}

}  // namespace

size_t qr_secant_lds(int n) {
    return (size_t)n * (2 * sizeof(double) + 4 * sizeof(double2) + sizeof(int));
}

hipError_t launch_transpose(int n, int nbatch, const double* in, double* out, const int* active,
                            hipStream_t stream) {
    const int t = (n + 15) / 16;
    hipLaunchKernelGGL(k_transpose, dim3(t, t, nbatch), dim3(256), 0, stream, n,
                       (const double2*)in, (double2*)out, active);
    return hipGetLastError();
}

hipError_t launch_qr_secant(int n, int nbatch, double* Wt, const double* Mp, const int* active,
                            double* tr, int* info, hipStream_t stream) {
    if (n < 1 || n > 1024) return hipErrorInvalidValue;
    if (n <= 256) return launch_qr<4, 2, 1024>(n, nbatch, Wt, Mp, active, tr, info, stream);
    if (n <= 512) return launch_qr<8, 2, 512>(n, nbatch, Wt, Mp, active, tr, info, stream);
    return launch_qr<16, 1, 512>(n, nbatch, Wt, Mp, active, tr, info, stream);
}

}  // namespace emme
