import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, emme_amd
d = bench.workload_dict(256)
p = emme_amd.params_from_dict(d)
g = bench.lattice(1, 0)
ctx = emme_amd.Context(p)
ctx.solve_roots(g)
os.environ["EMME_LU_SPLIT"] = "1"
r1, it1, info1 = ctx.solve_roots(g)
os.environ.pop("EMME_LU_SPLIT")
ctx.profile(True); ctx.profile_read(reset=True)
t = time.time(); r2, it2, info2 = ctx.solve_roots(g); print("wall", time.time() - t)
pr = ctx.profile_read(); print("lin ms", pr.linstep_ms)
print("info split1", np.unique(info1, return_counts=True))
print("info auto  ", np.unique(info2, return_counts=True))
print("iters equal", np.array_equal(it1, it2), "roots bitwise equal", np.array_equal(np.asarray(r1).view(np.float64), np.asarray(r2).view(np.float64), equal_nan=True))
