#!/usr/bin/env python3
"""Does a slow hipMalloc on one host thread hold up GPU work issued from another?  (No torch.)
A thread allocates `gib` GiB through the HIP runtime while the main thread keeps issuing small memsets + synchronisations
and records their latencies.  Run it right after a profiler pass of the bench workload (when allocations have been seen to
take seconds):  python tools/micro/bg_alloc_probe.py [gib] [malloc]"""
import ctypes, sys, threading, time
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 72
assert hip.hipSetDevice(0) == 0
small = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(small), ctypes.c_size_t(1 << 20)) == 0
stream = ctypes.c_void_p(); assert hip.hipStreamCreate(ctypes.byref(stream)) == 0
with_malloc = len(sys.argv) > 2 and sys.argv[2] == "malloc"   # the main thread also allocates and frees 1 MiB per trip
def op():
    t = time.perf_counter()
    hip.hipMemsetAsync(small, 0, ctypes.c_size_t(1 << 20), stream)
    hip.hipStreamSynchronize(stream)
    if with_malloc:
        q = ctypes.c_void_p()
        hip.hipMalloc(ctypes.byref(q), ctypes.c_size_t(1 << 20))
        hip.hipFree(q)
    return time.perf_counter() - t
for _ in range(100): op()
base = sorted(op() for _ in range(200))
res = {}
def alloc():
    hip.hipSetDevice(0)
    ps = []
    t = time.perf_counter()
    for k in range(7):  # like the node cache: seven allocations
        p = ctypes.c_void_p()
        rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t((gib << 30) // 7))
        ps.append(p)
    res["alloc_s"] = time.perf_counter() - t
    res["rc"] = rc
th = threading.Thread(target=alloc); t0 = time.perf_counter(); th.start()
lat = []
while th.is_alive():
    lat.append(op())
th.join()
lat.sort()
print(f"allocation of {gib} GiB in 7 pieces on a second thread: {res['alloc_s'] * 1e3:.1f} ms (rc {res['rc']}); main thread meanwhile: {len(lat)} memset+sync, "
      f"median {lat[len(lat) // 2] * 1e6 if lat else 0:.0f} us, worst {lat[-1] * 1e6 if lat else 0:.0f} us (alone: median {base[100] * 1e6:.0f} us, worst {base[-1] * 1e6:.0f} us)")
