#!/usr/bin/env python3
"""Run the bench's 128-chain root search N times on one settled context and compare every run bit for bit
with the first (roots, iteration counts, info codes, interval counts): a race in a kernel's hand-overs or
counters would show as a difference.  Development tool."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, emme_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = bench.lattice(1, 0)
with emme_amd.Context(emme_amd.params_from_dict(bench.workload_dict(256))) as ctx:
    ctx.profile(True)
    for _ in range(6):
        ctx.solve_roots(g)  # (the cache settles)
    ctx.profile_read(reset=True)
    ref = None
    bad = 0
    for k in range(n):
        r, it, inf = ctx.solve_roots(g)
        p = ctx.profile_read(reset=True)
        cur = (r.view(np.float64).copy(), it.copy(), inf.copy(), p.gk_intervals)
        if ref is None:
            ref = cur
        else:
            same = all(np.array_equal(a, b, equal_nan=True) if isinstance(a, np.ndarray) else a == b for a, b in zip(ref, cur))
            if not same:
                bad += 1
                print("run", k, "differs: intervals", cur[3], "vs", ref[3], "roots max diff", np.nanmax(np.abs(cur[0] - ref[0])))
    print(f"{n} searches on a settled context: {bad} differ from the first; {ref[3]} intervals per search")
sys.exit(1 if bad else 0)
