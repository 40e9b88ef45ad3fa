#!/usr/bin/env python3
"""Per-launch profile of root-search batches (development tool; the program the PMC passes run):
  python tools/iter_profile.py [nrep] [config]     config 3 (default): the bench workload; 4: BASELINE configs[3]
  (stellarator EM, K = 8 fixed steps, share 0); 5: BASELINE configs[4] (N = 512, the first k_rho of share 0).
One preparing search + nrep timed ones = nrep + 1 searches in a PMC pass."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, emme_amd
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kw = {}
if cfg == 3:
    d, g = bench.workload_dict(256), bench.lattice(1, 0)
elif cfg == 4:
    d, g, kw = dict(bench.STELLARATOR, npoints=256), bench.lattice_cfg4(0), {"step_limit": 7, "tol": 0.0}
else:
    krs, g = bench.sweep_cfg5(0)
    d = bench.workload_dict(512, k_rho=float(krs[0]))
p = emme_amd.params_from_dict(d)
ctx = emme_amd.Context(p)
ctx.solve_roots(g, **kw)
ctx.profile(True)
ctx.profile_read(reset=True)
for it in range(nrep):
    t = time.time()
    roots, iters, info = ctx.solve_roots(g, **kw)
    print("wall", time.time() - t)
pr = ctx.profile_read()
print("kernel", ctx.fill_kernel(), "rounds", pr.union_rounds, "element-intervals", pr.gk_intervals)
if pr.tile_tasks:
    print(f"dense rounds {pr.dense_rounds} sparse rounds {pr.sparse_rounds} (columns {pr.sparse_columns}) tile tasks {pr.tile_tasks}; "
          f"element-intervals per round {pr.gk_intervals / max(pr.union_rounds, 1):.1f} of 256")
print("asm ms", pr.assemble_ms, pr.assemble_launches, "deferred", pr.deferred_ms, "lin", pr.linstep_ms, "other", pr.other_ms, "evals", pr.integrand_evals)
print("points", int(iters[info == 0].sum()), "failed", int((info != 0).sum()))
