#!/usr/bin/env python3
"""Where does the cached fill stop fitting?  Root searches at growing grid sizes with the default
node-cache budget: cache geometry chosen, GiB held, cold and warm time, omega-points/s, and the
same without the cache (the fallback).  Development / reporting tool."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, emme_amd

sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024]
for n in sizes:
    nb = 128 if n <= 256 else (32 if n <= 512 else 16)
    g = bench.lattice(1, 0)[:: 128 // nb][:nb]
    for cache in ("default", "0"):
        opts = {"node_cache_gb": 0.0} if cache == "0" else {}
        emme_amd.release_pooled_memory()
        t0 = time.perf_counter()
        with emme_amd.Context(emme_amd.params_from_dict(bench.workload_dict(n)), **opts) as ctx:
            r, it, inf = ctx.solve_roots(g)
            t1 = time.perf_counter()
            r, it, inf = ctx.solve_roots(g)
            t2 = time.perf_counter()
            depth, sub, gib = ctx.cache_state()
            print(json.dumps({"npoints": n, "chains": nb, "cache_budget": cache, "fill_kernel": ctx.fill_kernel(),
                              "cache_full_depth": depth, "cache_subtrees": sub, "cache_gib": round(gib, 2),
                              "first_call_s": round(t1 - t0, 3), "second_call_s": round(t2 - t1, 3),
                              "omega_points_per_s_warm": round(float(it[inf == 0].sum()) / (t2 - t1), 1),
                              "omega_points_per_s_cold": round(float(it[inf == 0].sum()) / (t1 - t0), 1),
                              "info0": int((inf == 0).sum())}), flush=True)
