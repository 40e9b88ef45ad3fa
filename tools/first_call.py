#!/usr/bin/env python3
"""Cold-start cost of a context: creation, first root search (node-cache build), second one
(development tool; the second context of the run shows the effect of the buffer pool)."""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import emme_amd
import bench
for n in (256, 512):
    g = (np.linspace(-1.0, -0.5, 8)[None, :] + 1j * np.linspace(0.1, 0.4, 4)[:, None]).reshape(-1)
    t = time.perf_counter(); ctx = emme_amd.Context(emme_amd.params_from_dict(bench.workload_dict(n, k_rho=0.33))); t1 = time.perf_counter() - t
    t = time.perf_counter(); ctx.solve_roots(g); t2 = time.perf_counter() - t
    t = time.perf_counter(); ctx.solve_roots(g); t3 = time.perf_counter() - t
    ctx.profile(True); ctx.profile_read(reset=True)
    print(f"N={n}: ctx create {t1*1e3:.0f} ms, first solve (cache build) {t2*1e3:.0f} ms, second {t3*1e3:.0f} ms, cache {ctx.node_cache_gib():.1f} GiB")
    t = time.perf_counter(); ctx.close(); print(f"   close {1e3*(time.perf_counter()-t):.0f} ms")
