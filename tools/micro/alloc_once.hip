// One large device allocation, timed; optionally written to; then the process exits (its memory goes back to the driver).
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/build/alloc_once tools/micro/alloc_once.hip
// Run twice in a row, and again after a pause: whether the SECOND process waits for the driver to wipe what the first released.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(char* p, size_t n) {
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += (size_t)gridDim.x * blockDim.x * 16)
        *reinterpret_cast<int4*>(p + i) = make_int4(1, 2, 3, 4);
}
int main(int argc, char** argv) {
    const size_t gib = argc > 1 ? atoi(argv[1]) : 64;
    const int write = argc > 2 ? atoi(argv[2]) : 0;
    hipSetDevice(0);
    void* d = nullptr; hipMalloc(&d, 4096);  // runtime up
    void* p = nullptr;
    double t0 = now();
    hipError_t e = hipMalloc(&p, gib << 30);
    double t1 = now();
    if (e == hipSuccess && write) { fill<<<4096, 256>>>((char*)p, gib << 30); hipDeviceSynchronize(); }
    double t2 = now();
    printf("hipMalloc %zu GiB: %.1f ms (%s)%s", gib, (t1 - t0) * 1e3, hipGetErrorString(e), write ? "" : "\n");
    if (write) printf(", written in %.1f ms\n", (t2 - t1) * 1e3);
    return 0;
}
