import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, emme_amd
z = np.load("tests/golden/cfg3_damped.npz")
w = z["omegas"][-1:]
d = bench.workload_dict(256)
for opts in (dict(cache_min_batch=1), dict(node_cache_gb=0.0, wl_min=100000), dict(cache_min_batch=1, fill=emme_amd.FILL_UNION)):
    with emme_amd.Context(emme_amd.params_from_dict(d), **opts) as ctx:
        for rep in range(3):
            M = np.zeros((1, 256, 256), dtype=np.complex128)
            iv = np.zeros(1, dtype=np.int64)
            rc = ctx.lib.emme_assemble_batch(ctx.h, w.ctypes.data, 1, M.ctypes.data, iv.ctypes.data)
            print(opts, "rep", rep, "rc", rc, ctx.fill_kernel(), "intervals", iv[0], "want", z["intervals"][-1], "M[56,84]", M[0, 56, 84], "M[171,199]", M[0, 171, 199],
                  "nonfinite", (~np.isfinite(M)).sum(), "max finite", np.abs(M[np.isfinite(M)]).max(), flush=True)
