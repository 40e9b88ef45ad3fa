#!/usr/bin/env python3
"""Where does a (k_rho, omega) sweep point of BASELINE configs[4] spend its time?  A fresh context per k_rho
(N = 512, 32 guesses): context creation, the first root search with its cache build, destruction."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, emme_amd, torch
krs, g = bench.sweep_cfg5(0)
for rep in range(2):
    for kr in krs[:2]:
        t0 = time.perf_counter()
        ctx = emme_amd.Context(emme_amd.params_from_dict(bench.workload_dict(512, k_rho=float(kr))))
        ctx.profile(True)
        t1 = time.perf_counter()
        r, it, inf = ctx.solve_roots(g)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        p = ctx.profile_read(reset=True)
        ctx.close() if hasattr(ctx, "close") else ctx.__exit__(None, None, None)
        t3 = time.perf_counter()
        print(f"k_rho {kr:.3f}: create {1e3*(t1-t0):.1f} ms, search {1e3*(t2-t1):.1f} ms (cache build {p.cache_build_ms:.1f} in {p.cache_build_launches} launches, "
              f"alloc {p.cache_alloc_ms:.1f}, fill {p.assemble_ms:.1f}, deferred {p.deferred_ms:.1f}, lu {p.linstep_ms:.1f}, other {p.other_ms:.1f}), destroy {1e3*(t3-t2):.1f} ms; "
              f"{int(it[inf==0].sum())} omega-points", flush=True)
