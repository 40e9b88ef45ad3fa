#!/usr/bin/env python3
"""Golden Newton chains at N = 512 (BASELINE configs[4], the (k_rho, omega) sweep: bench.py --config 5):
tokamak ES, npoints = 512, omega_d_coeff = 1.01, k_rho = 0.2 (first value of the sweep) and 0.5 (last), the
first and the last guess of the sweep's 32-guess lattice each -- every iterate, computed like
make_golden_cfg3.py: reference kappa sources (oracle/_ref) for the fill, LAPACK zsysv for the step.
Run in the build container (about 10 minutes on 8 cores):
  python tests/golden/make_golden_cfg5.py   ->  tests/golden/cfg5_chains.npz
  k_rho[4], guesses[4], iterates[4, 22] (NaN padded), iters[4], roots[4], converged[4], info[4]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle.binding import Reference, example_tokamak  # noqa: E402
from make_golden_cfg3 import solve  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg5_chains.npz")


def main():
    g_all = (np.linspace(-1.0, -0.5, 8)[None, :] + 1j * np.linspace(0.1, 0.4, 4)[:, None]).reshape(-1)
    cases = [(0.2, g_all[0]), (0.2, g_all[-1]), (0.5, g_all[0]), (0.5, g_all[-1])]
    kr = np.array([c[0] for c in cases])
    g = np.array([c[1] for c in cases])
    iterates = np.full((len(cases), 22), np.nan + 1j * np.nan)
    iters = np.zeros(len(cases), dtype=np.int32)
    roots = np.full(len(cases), np.nan + 1j * np.nan)
    conv = np.zeros(len(cases), dtype=np.int32)
    info = np.zeros(len(cases), dtype=np.int32)
    ref = Reference()
    t00 = time.time()
    for b, (k, guess) in enumerate(cases):
        d = example_tokamak(npoints=512, omega_d_coeff=1.01, k_rho=float(k))
        ref.open_dict(d)
        its, ok, inf = solve(ref, d["npoints"], d["iteration_precision"], d["iteration_step_limit"], os.cpu_count(), guess)
        iterates[b, :len(its)] = its
        iters[b], roots[b], conv[b], info[b] = len(its), its[-1], ok, inf
        print(f"case {b}: k_rho {k} guess {guess:.4f} -> {its[-1]:.12f} in {len(its)} steps, converged={ok} info={inf} "
              f"({time.time() - t00:.0f} s)", flush=True)
        np.savez_compressed(OUT, k_rho=kr, guesses=g, iterates=iterates, iters=iters, roots=roots, converged=conv, info=info)


if __name__ == "__main__":
    main()
