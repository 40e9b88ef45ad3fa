#!/usr/bin/env python3
"""The five BASELINE.json configurations on ONE MI355X (for the 8-GPU ones: one GPU's share of
the sharded work list), with the checks that exist for each.  Development / reporting tool:
  python tools/run_configs.py            # prints one JSON line per configuration
configs[2] is the bench line (bench.py); the others are parity cases measured here for the record
(DESIGN.md section 8)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import emme_amd  # noqa: E402

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def timed_solve(ctx, guesses, **kw):
    ctx.solve_roots(guesses, **kw)  # builds the node cache, warms up
    t = time.perf_counter()
    roots, iters, info = ctx.solve_roots(guesses, **kw)
    return roots, iters, info, time.perf_counter() - t


def main():
    sv = json.load(open(os.path.join(G, "survey_appendix_b.json")))
    out = []
    # configs[0]: 64-point grid, single guess (CPU plumbing case in the reference) -- on the GPU
    with emme_amd.Context(emme_amd.params_from_dict(bench.workload_dict(64))) as ctx:
        roots, iters, info, dt = timed_solve(ctx, [-0.8 + 0.25j])
        want = complex(-0.67067782097052198, 0.27077138768282322)  # SURVEY App. B (complete reference)
        out.append({"config": 0, "what": "N=64, one root", "seconds": dt, "root_err_vs_reference": abs(roots[0] - want),
                    "iterations": int(iters[0])})
    # configs[1]: 256-point grid, single root
    with emme_amd.Context(emme_amd.params_from_dict(bench.workload_dict(256))) as ctx:
        roots, iters, info, dt = timed_solve(ctx, [-0.8 + 0.25j])
        want = complex(*sv["n256"]["iterates"][-1])
        out.append({"config": 1, "what": "N=256, one root", "seconds": dt, "root_err_vs_reference": abs(roots[0] - want),
                    "iterations": int(iters[0]), "omega_points_per_s": float(iters[0]) / dt})
        # configs[2]: 128-guess lattice (the bench workload; see bench.py for the full line)
        g = bench.lattice(1, 0)
        roots, iters, info, dt = timed_solve(ctx, g)
        out.append({"config": 2, "what": "N=256, 128-guess lattice", "seconds": dt, "omega_points_per_s": float(iters.sum()) / dt,
                    "converged": int((info == 0).sum()), "chains": len(g)})
    # configs[3]: stellarator EM, N=256 (dim 512), 32x32 lattice around (-1.656, 2.490): one GPU's 128
    # guesses of the round-robin deal, fixed work K = 8 Newton steps per guess
    d = dict(bench.STELLARATOR, npoints=256)
    re, im = np.linspace(-1.756, -1.556, 32), np.linspace(2.39, 2.59, 32)
    lat = (re[None, :] + 1j * im[:, None]).reshape(-1)[0::8]
    with emme_amd.Context(emme_amd.params_from_dict(d)) as ctx:
        roots, iters, info, dt = timed_solve(ctx, lat, step_limit=7, tol=0.0)
        out.append({"config": 3, "what": "stellarator EM N=256 (dim 512), 128 of 1024 guesses, 8 Newton steps each",
                    "seconds": dt, "omega_points_per_s": float(iters.sum()) / dt, "steps": int(iters.sum()),
                    "finite": int(np.isfinite(roots).sum()), "node_cache_gib": ctx.node_cache_gib()})
    # configs[4]: N=512, (k_rho, omega) sweep 32 x 32: one GPU's share = 4 k_rho values x 32 guesses
    krs = np.linspace(0.2, 0.5, 32)[0::8]
    guesses = np.linspace(-1.0, -0.5, 8)[None, :] + 1j * np.linspace(0.1, 0.4, 4)[:, None]
    guesses = guesses.reshape(-1)
    t_all, pts, conv = 0.0, 0, 0
    for kr in krs:
        with emme_amd.Context(emme_amd.params_from_dict(bench.workload_dict(512, k_rho=float(kr)))) as ctx:
            roots, iters, info, dt = timed_solve(ctx, guesses)
            t_all += dt
            pts += int(iters.sum())
            conv += int((info == 0).sum())
    out.append({"config": 4, "what": "N=512 ES, 4 of 32 k_rho values x 32 guesses (one context per k_rho; context "
                "creation and cache build not timed)", "seconds": t_all, "omega_points_per_s": pts / t_all,
                "converged": conv, "chains": 4 * len(guesses)})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
