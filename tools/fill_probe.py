#!/usr/bin/env python3
"""Which omegas set the duration of a late fill launch? (development tool)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, emme_amd, torch
d = bench.workload_dict(256)
p = emme_amd.params_from_dict(d)
g = bench.lattice(1, 0)
ctx = emme_amd.Context(p)
ctx.solve_roots(g)
roots, iters, info, its = ctx.solve_roots(g, want_iterates=True)
step = 10
live = np.flatnonzero(iters > step)
w = its[live, step]
print("live at step", step, ":", len(live))
buf = torch.zeros((len(g), ctx.dim, ctx.dim), dtype=torch.complex128, device="cuda")
def timed(omegas, label):
    omegas = np.asarray(omegas)
    ctx.assemble(omegas, out_device_ptr=buf.data_ptr())
    ctx.profile(True); ctx.profile_read(reset=True)
    iv = ctx.assemble(omegas, out_device_ptr=buf.data_ptr(), want_intervals=True)
    pr = ctx.profile_read(); ctx.profile(False)
    print(f"{label:40s} n={len(omegas):3d} fill {pr.assemble_ms:7.3f} ms deferred {pr.deferred_ms:6.3f} ms; intervals/omega min {iv.min()} median {int(np.median(iv))} max {iv.max()}")
    return iv
iv = timed(w, "all live omegas")
order = np.argsort(iv)
print("omegas sorted by cost:", [(complex(np.round(w[k], 3)), int(iv[k])) for k in order[-6:]])
timed(w[order[:-1]], "without the most expensive")
timed(w[order[:-3]], "without the 3 most expensive")
timed(w[order[:16]], "the 16 cheapest")
timed(np.repeat(w[order[-1]], 16), "16 copies of the most expensive")
timed(np.repeat(w[order[0]], 16), "16 copies of the cheapest")
timed(np.repeat(w[order[len(order)//2]], 16), "16 copies of the median")
timed(g[:16] * 0.99, "16 lattice guesses")
timed(g * 0.99, "128 lattice guesses")
