#!/usr/bin/env python3
"""The electromagnetic fill alone (BASELINE configs[3]: stellarator, N = 256, dim 512, GK31): plain assembly of
128 omegas -- the lattice guesses of share 0, or with `newton` their first Newton iterates (deeper trees) -- kernel
time per launch from the library's profile, default fill against the independent-lane kernel, entries compared.
python tools/em_fill_probe.py [nrep] [iterate index 0..8]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, emme_amd
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 5
p = emme_amd.params_from_dict(dict(bench.STELLARATOR, npoints=256))
g = bench.lattice_cfg4(0)
res = {}
for name, opts in (("default", {}), ("lanes", dict(fill=emme_amd.FILL_LANES))):
    ctx = emme_amd.Context(p, **opts)
    w = g
    if len(sys.argv) > 2 and sys.argv[2] == "damped":
        # 128 omegas around the reference's own step-7 iterate of tests/golden/cfg4_k8_n256.npz, chain 0 (no root search here)
        z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cfg4_k8_n256.npz"))
        w = z["iterates"][0, 7] + (g - g.mean()) * 0.5
    elif len(sys.argv) > 2:
        _, _, _, its = ctx.solve_roots(g, tol=0.0, step_limit=7, want_iterates=True)
        w = its[:, int(sys.argv[2])]
        print("omega[0]", w[0], "spread", np.abs(w - w.mean()).max())
    M, iv = ctx.assemble(w, want_intervals=True)
    ctx.profile(True); ctx.profile_read(reset=True)
    for _ in range(nrep):
        M2, iv2 = ctx.assemble(w, want_intervals=True)
    pr = ctx.profile_read()
    print(f"{name}: kernel {ctx.fill_kernel_symbol()}: {pr.assemble_ms / max(pr.assemble_launches, 1):.3f} ms per launch "
          f"({pr.assemble_launches} launches), deferred {pr.deferred_ms / nrep:.3f} ms, other {pr.other_ms / nrep:.3f} ms; "
          f"intervals per integral {iv2.sum() / (len(g) * 3 * 256 * 255 / 2):.2f}; repeat diff {np.abs(M2 - M).max():.1e}; "
          f"dense rounds {pr.dense_rounds} sparse {pr.sparse_rounds} tasks {pr.tile_tasks}", flush=True)
    res[name] = (M2, iv2)
    del ctx
(Ma, iva), (Mb, ivb) = res["default"], res["lanes"]
print("interval counts equal:", bool(np.array_equal(iva, ivb)), "max |dM| / max|M| per matrix:",
      float((np.abs(Ma - Mb).max(axis=(1, 2)) / np.abs(Mb).max(axis=(1, 2))).max()))
