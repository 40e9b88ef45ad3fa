// What a large device allocation costs on this machine, and whether it can hide behind running kernels.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/build/alloc_probe tools/micro/alloc_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); (void)hipGetLastError(); } } while (0)
__global__ void spin(double* out, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.0000001;
    for (int i = 0; i < iters; ++i) a = a * b + 1e-9;
    if (a == 12345.0) out[0] = a;
}
__global__ void touch(char* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4096; i += (size_t)gridDim.x * blockDim.x) p[i * 4096] = 1;
}
int main() {
    const size_t G = 1ull << 30;
    CK(hipSetDevice(0));
    double* dummy; CK(hipMalloc(&dummy, 4096));
    for (size_t gib : {1, 8, 16}) {
        void* p = nullptr;
        double t0 = now(); CK(hipMalloc(&p, gib * G)); double t1 = now();
        touch<<<1024, 256>>>((char*)p, gib * G); CK(hipDeviceSynchronize()); double t2 = now();
        CK(hipFree(p)); double t3 = now();
        printf("hipMalloc %2zu GiB: %.1f ms (%.1f ms/GiB); first touch %.1f ms; hipFree %.1f ms\n", gib, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / gib, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
        t0 = now(); CK(hipMalloc(&p, gib * G)); t1 = now(); CK(hipFree(p));
        printf("   again: %.1f ms\n", (t1 - t0) * 1e3);
    }
    {   // stream-ordered pool
        hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0));
        unsigned long long thr = ~0ull; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
        void* p = nullptr;
        double t0 = now(); CK(hipMallocAsync(&p, 8 * G, 0)); CK(hipStreamSynchronize(0)); double t1 = now();
        CK(hipFreeAsync(p, 0)); CK(hipStreamSynchronize(0)); double t2 = now();
        CK(hipMallocAsync(&p, 8 * G, 0)); CK(hipStreamSynchronize(0)); double t3 = now();
        CK(hipFreeAsync(p, 0)); CK(hipStreamSynchronize(0));
        printf("hipMallocAsync 8 GiB: %.1f ms; free %.1f ms; again from the pool %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
        size_t z = 0; CK(hipMemPoolTrimTo(pool, z));
    }
    {   // virtual memory API, 2 MiB-granular physical chunks of 1 GiB
        hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        void* va = nullptr; double t0 = now(); CK(hipMemAddressReserve(&va, 8 * G, gran, nullptr, 0)); double t1 = now();
        std::vector<hipMemGenericAllocationHandle_t> hs(8);
        for (int k = 0; k < 8; ++k) { CK(hipMemCreate(&hs[k], G, &prop, 0)); CK(hipMemMap((char*)va + k * G, G, 0, hs[k], 0)); }
        hipMemAccessDesc ad{}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, 8 * G, &ad, 1)); double t2 = now();
        touch<<<1024, 256>>>((char*)va, 8 * G); CK(hipDeviceSynchronize()); double t3 = now();
        printf("VMM: granularity %zu; reserve %.2f ms; create+map+access 8 x 1 GiB %.1f ms; touch %.1f ms\n", gran, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
        CK(hipMemUnmap(va, 8 * G)); for (auto h : hs) CK(hipMemRelease(h)); CK(hipMemAddressFree(va, 8 * G));
    }
    {   // kernels on the main thread while another thread allocates
        std::atomic<int> done{0};
        double talloc = 0;
        hipStream_t s; CK(hipStreamCreate(&s));
        spin<<<1024, 256, 0, s>>>(dummy, 20000); CK(hipStreamSynchronize(s));
        double k0 = now(); for (int i = 0; i < 20; ++i) spin<<<1024, 256, 0, s>>>(dummy, 20000); CK(hipStreamSynchronize(s)); double k1 = now();
        printf("kernel alone: %.3f ms each\n", (k1 - k0) * 1e3 / 20);
        void* p = nullptr;
        std::thread th([&] { CK(hipSetDevice(0)); double t0 = now(); CK(hipMalloc(&p, 16 * G)); talloc = now() - t0; done = 1; });
        int n = 0; double w0 = now(), worst = 0;
        while (!done) { double a = now(); spin<<<1024, 256, 0, s>>>(dummy, 20000); CK(hipStreamSynchronize(s)); double d = now() - a; if (d > worst) worst = d; ++n; }
        double w1 = now(); th.join();
        printf("during hipMalloc 16 GiB on another thread (%.1f ms): %d kernels, %.3f ms each, worst %.3f ms\n", talloc * 1e3, n, (w1 - w0) * 1e3 / (n ? n : 1), worst * 1e3);
        CK(hipFree(p));
        // two allocating threads
        void *pa = nullptr, *pb = nullptr; double t0 = now();
        std::thread ta([&] { CK(hipSetDevice(0)); CK(hipMalloc(&pa, 8 * G)); }), tb([&] { CK(hipSetDevice(0)); CK(hipMalloc(&pb, 8 * G)); });
        ta.join(); tb.join(); double t1 = now();
        printf("two threads x hipMalloc 8 GiB: %.1f ms\n", (t1 - t0) * 1e3);
        CK(hipFree(pa)); CK(hipFree(pb));
    }
    return 0;
}
