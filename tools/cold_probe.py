#!/usr/bin/env python3
"""Cold cost of a context: first root search of the bench workload in a fresh process, with the library's own
allocation timing (EMME_DEBUG=1 prints every node-cache allocation).  python tools/cold_probe.py [plain|torch|cuda]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "plain"   # plain | torch (import only) | cuda (torch.cuda.init() first)
if mode in ("torch", "cuda"):
    import torch
    if mode == "cuda":
        torch.cuda.init(); torch.zeros(1, device="cuda:0")
t0 = time.time()
import bench, emme_amd
t1 = time.time()
p = emme_amd.params_from_dict(bench.workload_dict(256))
g = bench.lattice(1, 0)
ctx = emme_amd.Context(p)
t2 = time.time()
ctx.profile(True)
roots, iters, info = ctx.solve_roots(g)
t3 = time.time()
pr = ctx.profile_read(reset=True)
print(mode, [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1])
print(f"import {t1 - t0:.2f} s, context {t2 - t1:.3f} s, first search {t3 - t2:.3f} s "
      f"(cache alloc {pr.cache_alloc_ms:.0f} ms, cache build {pr.cache_build_ms:.0f} ms, {ctx.node_cache_gib():.1f} GiB)")
later = []
for k in range(3):
    t = time.time(); ctx.solve_roots(g); later.append(time.time() - t); print(f"search {k + 2}: {later[-1]:.3f} s, {ctx.node_cache_gib():.1f} GiB")
if "--json" in sys.argv:  # (bench.py runs this file as a child process, before it touches the GPU itself, and reads this line)
    import json
    print(json.dumps({"mode": mode, "context_create_s": t2 - t1, "first_search_s": t3 - t2, "node_cache_alloc_ms": pr.cache_alloc_ms,
                      "node_cache_build_ms": pr.cache_build_ms, "node_cache_gib": ctx.node_cache_gib(),
                      "first_search_omega_points": int(iters[info == 0].sum()), "later_search_s": min(later)}), flush=True)
    # Leave WITHOUT giving the 72 GiB back call by call: the process that allocates right after one that hipFree'd tens of
    # GiB waits seconds inside hipMalloc (tools/micro/bg_alloc_probe.py run in turns with this file: 3.8 s after this
    # file's orderly exit, 2 ms after its own exit without frees) -- and that next process is the bench itself.
    os._exit(0)
