#!/usr/bin/env python3
"""Development probe (GPU box): matrices of the dense and the union fill against the oracle at the
omegas where chain 30 of the cfg3 lattice leaves the reference's path."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, emme_amd
from oracle.binding import Oracle
orc = Oracle(); d = bench.workload_dict(256); po = orc.params(d)
ws = np.array([0.06302406 - 0.00126148j, 0.04680305 + 0.00355542j, -0.57163319 - 0.17391373j, -0.8 + 0.25j,
               0.05 + 0.01j, 0.06 - 0.002j, 0.03 + 0.001j, 0.1 + 0.02j, 0.2 - 0.01j])
Mo = [orc.assemble(po, complex(w), 16) for w in ws]
g = bench.lattice(1, 0)
for name, env in (("dense", "1"), ("union", "0")):
    os.environ["EMME_DENSE"] = env
    with emme_amd.Context(emme_amd.params_from_dict(d)) as ctx:
        M, iv = ctx.assemble(ws, want_intervals=True)
        for k, w in enumerate(ws):
            e = np.abs(M[k] - Mo[k][0])
            i, j = np.unravel_index(e.argmax(), e.shape)
            print(f"{name} w={w}: max|dM|/max|M| = {e.max() / np.abs(Mo[k][0]).max():.3e} at ({i},{j}) |M_ij|={abs(Mo[k][0][i, j]):.3e}, "
                  f"intervals {iv[k]} vs oracle {Mo[k][1]}", flush=True)
        roots, iters, info, its = ctx.solve_roots(g[[30, 46, 47]], want_iterates=True)
        print(name, "chain 30 iterates", its[0, :12])
        print(name, "roots", roots, iters, info)
