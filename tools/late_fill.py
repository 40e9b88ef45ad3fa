#!/usr/bin/env python3
"""Where does the time of a late-iteration fill go?  (development tool)
Runs the bench root search once with iterates, takes the omegas that are still iterating at
step 12 and times fills of sub-sets of them."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, emme_amd, torch
d = bench.workload_dict(256)
ctx = emme_amd.Context(emme_amd.params_from_dict(d))
g = bench.lattice(1, 0)
roots, iters, info, its = ctx.solve_roots(g, want_iterates=True)
late = [b for b in range(len(g)) if iters[b] > 13]
ws = np.array([its[b][12] for b in late])
buf = torch.zeros((len(ws), ctx.dim, ctx.dim), dtype=torch.complex128, device="cuda")
def t_fill(w):
    ctx.assemble(w, out_device_ptr=buf.data_ptr())
    torch.cuda.synchronize(); t = time.perf_counter()
    iv = ctx.assemble(w, out_device_ptr=buf.data_ptr())
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3, iv
ms, iv = t_fill(ws)
print(f"{len(ws)} late omegas together: {ms:.2f} ms, {iv.sum()/1e6:.1f}M intervals")
order = np.argsort(-iv)
for k in order[:6]:
    m1, i1 = t_fill(ws[k:k + 1])
    print(f"  omega {ws[k]:.4f}: alone {m1:.2f} ms, {i1[0]/1e6:.2f}M intervals ({i1[0]/32640:.0f}/entry)")
cheap = order[6:]
if len(cheap):
    m2, i2 = t_fill(ws[cheap])
    print(f"  the other {len(cheap)}: {m2:.2f} ms, {i2.sum()/1e6:.1f}M intervals")
m3, i3 = t_fill(ws[order[:2]])
print(f"  heaviest two together: {m3:.2f} ms")
