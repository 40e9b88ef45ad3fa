#!/usr/bin/env python3
"""Per-launch profile of one root-search batch (development tool; the program the PMC passes run)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, emme_amd
d = bench.workload_dict(256)
p = emme_amd.params_from_dict(d)
g = bench.lattice(1, 0)
ctx = emme_amd.Context(p)
ctx.solve_roots(g)
ctx.profile(True)
ctx.profile_read(reset=True)
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for it in range(nrep):
    t = time.time()
    roots, iters, info = ctx.solve_roots(g)
    print("wall", time.time() - t)
pr = ctx.profile_read()
print("kernel", ctx.fill_kernel(), "rounds", pr.union_rounds, "element-intervals", pr.gk_intervals)
if pr.tile_tasks:
    print(f"dense rounds {pr.dense_rounds} sparse rounds {pr.sparse_rounds} (columns {pr.sparse_columns}) tile tasks {pr.tile_tasks}; "
          f"element-intervals per round {pr.gk_intervals / max(pr.union_rounds, 1):.1f} of 256")
print("asm ms", pr.assemble_ms, pr.assemble_launches, "deferred", pr.deferred_ms, "lin", pr.linstep_ms, "other", pr.other_ms, "evals", pr.integrand_evals)
print("points", int(iters[info == 0].sum()), "failed", int((info != 0).sum()))
