#!/usr/bin/env python3
"""What the round-robin deal of the scan does to the per-rank time -- measured on ONE GPU, rank share after rank
share (no multi-GPU node is available to the builder; the driver measures the real curve as SCALE_rNN.json).

For world = 1, 2, 4, 8 the weak-scaling lattice of bench.py (128 guesses per rank: Re w in linspace(-1.2,-0.4,16) x
Im w in linspace(0.05,0.40, 8 world), item k -> rank k mod world) is solved share by share on a settled context;
the same for the 8 shares of BASELINE configs[3] (--config 4) and configs[4] (--config 5).  Output (JSON, one
object): per world the per-share milliseconds, omega-points, max/mean of the time = the efficiency a perfectly
overlapped N-GPU run would lose to imbalance alone (predicted weak-scaling efficiency = T(world 1) / max share time).

  python tools/scaling_prediction.py > profiles/r03_scaling_prediction.json
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, emme_amd


def timed(ctx, g, reps=3, **kw):
    for _ in range(3):  # cache grown and settled for this share
        ctx.solve_roots(g, **kw)
    best, pts = 1e9, 0
    for _ in range(reps):
        t = time.perf_counter()
        r, it, inf = ctx.solve_roots(g, **kw)
        best = min(best, time.perf_counter() - t)
        pts = int(it[inf == 0].sum())
    return best * 1e3, pts


out = {"note": __doc__.split("\n\n")[0], "device": "1 x MI355X, shares run one after the other"}
d = bench.workload_dict(256)
res = {}
for world in (1, 2, 4, 8):
    ms, pts = [], []
    with emme_amd.Context(emme_amd.params_from_dict(d)) as ctx:
        for rank in range(world):
            t, p = timed(ctx, bench.lattice(world, rank))
            ms.append(round(t, 2)), pts.append(p)
    res[str(world)] = {"share_ms": ms, "share_omega_points": pts, "max_over_mean": round(max(ms) / np.mean(ms), 4),
                       "omega_points_per_s_if_perfectly_overlapped": round(sum(pts) / (max(ms) * 1e-3), 1)}
t1 = res["1"]["share_ms"][0]
for w in res:
    res[w]["predicted_weak_scaling_efficiency"] = round(t1 / max(res[w]["share_ms"]), 4)
out["config3_headline_lattice"] = res
if "--all" in sys.argv:
    d4 = dict(bench.STELLARATOR, npoints=256)
    ms, pts = [], []
    with emme_amd.Context(emme_amd.params_from_dict(d4)) as ctx:
        for share in range(8):
            t, p = timed(ctx, bench.lattice_cfg4(share), reps=2, step_limit=7, tol=0.0)
            ms.append(round(t, 2)), pts.append(p)
    out["config4_shares"] = {"share_ms": ms, "share_omega_points": pts, "max_over_mean": round(max(ms) / np.mean(ms), 4)}
    ms, pts = [], []
    for share in [0] + list(range(8)):  # (share 0 once untimed first: the process's buffer pool changes size class here)
        krs, g = bench.sweep_cfg5(share)
        if len(ms) == 0 and share == 0 and not pts and not globals().get("_warmed"):
            _warmed = True
            for kr in krs:
                with emme_amd.Context(emme_amd.params_from_dict(bench.workload_dict(512, k_rho=float(kr)))) as ctx:
                    ctx.solve_roots(g)
            continue
        t0 = time.perf_counter()
        p = 0
        for kr in krs:
            with emme_amd.Context(emme_amd.params_from_dict(bench.workload_dict(512, k_rho=float(kr)))) as ctx:
                r, it, inf = ctx.solve_roots(g)
                p += int(it[inf == 0].sum())
        ms.append(round((time.perf_counter() - t0) * 1e3, 1)), pts.append(p)
    out["config5_shares"] = {"share_ms": ms, "share_omega_points": pts, "max_over_mean": round(max(ms) / np.mean(ms), 4),
                             "note": "a fresh context per k_rho, as bench.py --config 5 times it"}
print(json.dumps(out, indent=1))
