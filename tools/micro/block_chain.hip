// block_chain.hip -- how fast can a wave follow a chain of dependent 8-KB block reads (the shape of a vector
// round of k_assemble_dense: the next block's address is known only after the current one was summed), as a
// function of the table size and of the waves per SIMD?  Development micro-benchmark.
//   hipcc --offload-arch=gfx950 -O3 -o build/block_chain block_chain.hip && build/block_chain
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            std::exit(1);                                                             \
        }                                                                             \
    } while (0)

// each wave: `steps` dependent reads of one 8-KB block (64 lanes x 8 x 16 B), block index from a hash of
// the previous block's content (all zeros -> the hash of the step counter: the dependence is real, the
// values are not)
template <int WAVES_PER_BLOCK>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_chain(const double2* tab, unsigned long long nblocks,
                                                               int steps, int local, unsigned long long* out) {
    const int lane = threadIdx.x & 63;
    const unsigned long long wid = (unsigned long long)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    unsigned long long h = wid * 0x9E3779B97F4A7C15ull + 12345ull;
    // `local` > 0: the wave stays inside a window of that many blocks (a tile's 4-MB slice of the cache)
    const unsigned long long base = local > 0 ? (wid * (unsigned long long)local) % (nblocks - local) : 0ull;
    const unsigned long long span = local > 0 ? (unsigned long long)local : nblocks;
    double acc = 0.0;
    for (int s = 0; s < steps; ++s) {
        h = h * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned long long blk = base + (h >> 20) % span;
        const double2* p = tab + blk * 512 + lane;  // 512 double2 = 8 KB
        double2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[64 * k];
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += v[k].x + v[k].y;
        acc += t;
        // the next address depends on the data (wave-uniform: first lane's sum)
        const long long bits = __double_as_longlong(t);
        h += (unsigned long long)__builtin_amdgcn_readfirstlane((int)bits);
    }
    if (lane == 0) out[wid] = (unsigned long long)__double_as_longlong(acc) + h;
}

int main(int argc, char** argv) {
    const int steps = 2000;
    unsigned long long* out;
    CHECK(hipMalloc(&out, sizeof(unsigned long long) * (1 << 20)));
    const double sizes_gib[] = {0.25, 2.0, 16.0, 64.0};
    for (double gib : sizes_gib) {
        const unsigned long long bytes = (unsigned long long)(gib * (1ull << 30));
        double2* tab;
        if (hipMalloc(&tab, bytes) != hipSuccess) {
            std::printf("%.2f GiB: allocation failed\n", gib);
            continue;
        }
        CHECK(hipMemset(tab, 0, bytes));
        const unsigned long long nblocks = bytes / 8192;
        for (int local : {0, 512}) {
            for (int wps : {1, 2, 3, 4, 8}) {  // waves per SIMD: 256 CUs x 4 SIMDs
                const int waves = 1024 * wps;
                hipEvent_t e0, e1;
                CHECK(hipEventCreate(&e0));
                CHECK(hipEventCreate(&e1));
                hipLaunchKernelGGL(k_chain<4>, dim3(waves / 4), dim3(256), 0, 0, tab, nblocks, 50, local, out);
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k_chain<4>, dim3(waves / 4), dim3(256), 0, 0, tab, nblocks, steps, local, out);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0.f;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double per_step_us = ms * 1e3 / steps;
                const double tbps = (double)waves * steps * 8192.0 / (ms * 1e-3) / 1e12;
                std::printf("table %6.2f GiB  %s  %d waves/SIMD: %.2f us per dependent block read, %.2f TB/s\n", gib,
                            local ? "4-MB window per wave" : "whole table        ", wps, per_step_us, tbps);
            }
        }
        CHECK(hipFree(tab));
    }
    return 0;
}
