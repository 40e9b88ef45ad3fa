#!/usr/bin/env python3
"""Micro-benchmark of the batched Newton linear step alone (development tool)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import emme_amd, torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 128
kind = sys.argv[3] if len(sys.argv) > 3 else "lu"   # "lu" (trace form) or "qr"
rng = np.random.default_rng(0)
A = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n)); A = A + np.transpose(A, (0, 2, 1)) + 4 * np.eye(n)
B = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n))
ctx = emme_amd.Context(emme_amd.params_from_dict(bench.workload_dict(16)))
dA = torch.from_numpy(A).cuda(); dB = torch.from_numpy(B).cuda()
lib = ctx.lib
tr = np.zeros(nb, dtype=np.complex128); info = np.zeros(nb, dtype=np.int32)
def run():
    a = dA.clone(); b = dB.clone(); torch.cuda.synchronize()
    t = time.perf_counter()
    if kind == "qr":
        rc = lib.emme_qr_secant_batch(ctx.h, n, nb, a.data_ptr(), b.data_ptr(), tr.ctypes.data, info.ctypes.data)
    else:
        rc = lib.emme_trace_solve_batch(ctx.h, n, nb, a.data_ptr(), b.data_ptr(), tr.ctypes.data, info.ctypes.data)
    torch.cuda.synchronize()
    return time.perf_counter() - t, rc
run()
ts = [run()[0] for _ in range(5)]
if kind == "qr":
    want = tr[0]  # (the QR quotient is checked against the LAPACK sequence in tests/test_gpu_parity.py)
else:
    want = np.trace(np.linalg.solve(A[0], B[0]))
    allw = np.trace(np.linalg.solve(A, B), axis1=1, axis2=2)
    print(f"   all {nb} matrices: max rel err {np.max(np.abs(tr - allw) / np.abs(allw)):.1e}, info max {info.max()} min {info.min()}, split {os.environ.get('EMME_LU_SPLIT', 'auto')}")
print(f"{kind} n={n} batch={nb}: {min(ts)*1e3:.2f} ms (median {np.median(ts)*1e3:.2f}); check rel err {abs(tr[0]-want)/abs(want):.1e}")
