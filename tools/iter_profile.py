#!/usr/bin/env python3
"""Per-launch profile of one root-search batch (development tool)."""
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
for it in range(1):
    t = time.time()
    roots, iters, info = ctx.solve_roots(g)
    print("wall", time.time() - t)
pr = ctx.profile_read()
print("rounds", pr.union_rounds, "lane-intervals", pr.gk_intervals, "fill", pr.gk_intervals / (16.0 * max(pr.union_rounds, 1)))
print("asm ms", pr.assemble_ms, pr.assemble_launches, "lin", pr.linstep_ms, "evals", pr.integrand_evals)
