// Issue rate of v_mfma_f64_16x16x4_f64 and of v_fma_f64 on one CU (development micro-benchmark).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void k_mfma(double* out, int iters, unsigned long long* ticks) {
    d4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

template <int CHAINS>
__global__ void k_fma(double* out, int iters, unsigned long long* ticks) {
    double acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = c;
    double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-4;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = fma(acc[c], a, b);
    }
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

int main() {
    double* out;
    unsigned long long* ticks;
    hipMalloc(&out, 1024 * 8 * 1024);
    hipMallocManaged(&ticks, 8);
    const int iters = 20000;
    for (int threads : {256, 512, 1024}) {
        for (int chains : {1, 2, 8}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (chains == 1) hipLaunchKernelGGL(k_mfma<1>, dim3(1), dim3(threads), 0, 0, out, iters, ticks);
                if (chains == 2) hipLaunchKernelGGL(k_mfma<2>, dim3(1), dim3(threads), 0, 0, out, iters, ticks);
                if (chains == 8) hipLaunchKernelGGL(k_mfma<8>, dim3(1), dim3(threads), 0, 0, out, iters, ticks);
                hipDeviceSynchronize();
            }
            const double ns = *ticks * 10.0;  // 100 MHz counter
            const double n_per_simd = (double)iters * chains * (threads / 64) / 4.0;
            printf("mfma_f64_16x16x4: %4d threads, %d chains/wave: %.1f ns per MFMA per SIMD (%.1f GFLOP/s per CU)\n",
                   threads, chains, ns / n_per_simd, 2048.0 * n_per_simd * 4 / ns);
        }
    }
    for (int threads : {256, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_fma<8>, dim3(1), dim3(threads), 0, 0, out, iters, ticks);
            hipDeviceSynchronize();
        }
        const double ns = *ticks * 10.0;
        const double n_per_simd = (double)iters * 8 * (threads / 64) / 4.0;
        printf("v_fma_f64: %4d threads, 8 chains: %.2f ns per wave-instruction per SIMD (%.1f GFLOP/s per CU)\n", threads,
               ns / n_per_simd, 128.0 * n_per_simd * 4 / ns);
    }
    return 0;
}
