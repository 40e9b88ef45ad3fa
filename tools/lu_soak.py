#!/usr/bin/env python3
"""Soak of the multi-workgroup LU: many launches, every one compared bit for bit with the
one-workgroup result (development tool; a race in the hand-over would show as a mismatch)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import emme_amd
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = emme_amd.Context(emme_amd.params_from_dict(bench.workload_dict(16)))
rng = np.random.default_rng(2024)
bad = 0
total = 0
t0 = time.time()
# (2 and 3 workgroups per matrix at n >= 256 = the grouped kernel, k_trace_solve_grouped; the others the
# per-panel kernel with look-ahead)
for n, nb, splits in [(256, 128, ["2"]), (256, 64, ["4", "3"]), (256, 20, ["4", "8"]), (200, 36, ["5", "7"]),
                      (512, 32, ["8", "6"]), (512, 64, ["4", "2", "3"]), (512, 128, ["2"]), (448, 40, ["2", "3"]),
                      (144, 50, ["5"]), (384, 24, ["8", "2"])]:
    for r in range(reps):
        A = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n))
        A = A + np.transpose(A, (0, 2, 1)) + 0.5 * n ** 0.5 * np.eye(n)
        B = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n))
        ctx.set_options(lu_split=1)
        tr1, i1 = ctx.trace_solve(A, B)
        for sp in splits:
            ctx.set_options(lu_split=int(sp))
            trs, i_s = ctx.trace_solve(A, B)
            total += 1
            if not (np.array_equal(i1, i_s) and np.array_equal(tr1.view(np.float64), trs.view(np.float64))):
                bad += 1
                print("MISMATCH", n, nb, sp, r, "info", np.unique(i_s), "max diff", np.nanmax(np.abs(tr1 - trs)))
    print(f"n={n} nb={nb} splits={splits}: {reps} repetitions done, mismatches so far {bad}, {time.time()-t0:.0f} s", flush=True)
print("launches compared:", total, "mismatches:", bad)
sys.exit(1 if bad else 0)
