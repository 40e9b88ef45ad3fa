#!/usr/bin/env python3
"""Why a 16 GiB hipMalloc takes 0.3 ms in a plain process and ~0.5 s once torch.cuda is initialised:
time the same allocation (through the HIP runtime the process has loaded) after each stage of torch's start-up."""
import ctypes, sys, time
import torch
hip = ctypes.CDLL([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0])
def t_alloc(label, gib=16):
    p = ctypes.c_void_p()
    t = time.time(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(gib << 30)); dt = time.time() - t
    t = time.time(); hip.hipFree(p); df = time.time() - t
    print(f"{label:50s} hipMalloc {gib} GiB rc={rc}: {dt * 1e3:8.1f} ms, hipFree {df * 1e3:.1f} ms", flush=True)
hip.hipSetDevice(0)
t_alloc("after import torch")
torch.cuda.init()
t_alloc("after torch.cuda.init()")
x = torch.empty(16, device="cuda:0")
t_alloc("after torch.empty on the device")
x.zero_(); torch.cuda.synchronize()
t_alloc("after one torch kernel")
y = torch.randn(256, 256, device="cuda:0") @ torch.randn(256, 256, device="cuda:0"); torch.cuda.synchronize()
t_alloc("after a matmul (hipBLASLt loaded)")
t_alloc("again")
s = torch.cuda.Stream(); e = torch.cuda.Event(enable_timing=True)
t_alloc("after a stream and an event")
