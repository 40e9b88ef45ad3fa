#!/usr/bin/env python3
"""Stellarator EM (GK31, 3 moments, dim = 2N) spot benchmark: one root search batch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, emme_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
d = dict(bench.STELLARATOR, npoints=n, iteration_step_limit=30)
p = emme_amd.params_from_dict(d)
g = (np.linspace(-1.8, -1.5, nb) + 1j * np.linspace(2.3, 2.6, nb))
ctx = emme_amd.Context(p)
t = time.time(); roots, iters, info = ctx.solve_roots(g); t1 = time.time() - t
ctx.profile(True); ctx.profile_read(reset=True)
t = time.time(); roots, iters, info = ctx.solve_roots(g); t2 = time.time() - t
pr = ctx.profile_read()
print(f"N={n} dim={ctx.dim} batch={nb}: first {t1:.2f}s, second {t2:.2f}s; omega-points {iters.sum()} -> {iters.sum()/t2:.1f}/s; "
      f"fill {pr.assemble_ms:.1f} ms + deferred {pr.deferred_ms:.1f} ms, LU {pr.linstep_ms:.1f} ms, other {pr.other_ms:.1f}; cache {ctx.node_cache_gib():.1f} GiB; kernel {ctx.fill_kernel()}")
print("iters", iters[:16], "info", info[:16]); print("roots", roots[:4])
