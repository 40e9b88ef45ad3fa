import sys, numpy as np, time
sys.path.insert(0,'/root/repo')
import emme_amd, bench
d = bench.workload_dict(256)
p = emme_amd.params_from_dict(d)
g = bench.lattice(1,0)
ctx = emme_amd.Context(p)
ctx.profile(True)
t=time.time()
roots, iters, info, its = ctx.solve_roots(g, want_iterates=True)
print("time", time.time()-t)
pr = ctx.profile_read()
print("asm ms", pr.assemble_ms, pr.assemble_launches, "lin ms", pr.linstep_ms, pr.linstep_launches, "other", pr.other_ms, "intervals", pr.gk_intervals)
np.set_printoptions(linewidth=200, precision=5)
for b in range(len(g)):
    print(b, g[b], iters[b], info[b], roots[b])
