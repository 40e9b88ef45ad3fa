#!/usr/bin/env python3
"""Golden Newton chains of the HEADLINE workload (BASELINE configs[2] / bench.py): tokamak ES,
npoints=256, omega_d_coeff=1.01, the 128-guess lattice Re w in linspace(-1.2,-0.4,16) x
Im w in linspace(0.05,0.40,8) -- every chain, every iterate.

Computed in the build container from the reference's own kappa sources (oracle/_ref, built by
oracle/Makefile from /root/reference) for the fill, and the LAPACK routine the reference calls for
the Newton step (zsysv "Upper", include/solver.h:134-136, through SciPy's OpenBLAS), following
EigenSolver's constructor (include/solver.h:396-415), newtonTraceSecantIteration (:113-160) and
the solve-once loop (src/main.cpp:43-57) literally.

Run:  python tests/golden/make_golden_cfg3.py [first last]   (about 35 min on 8 cores for all 128)
      python tests/golden/make_golden_cfg3.py perturb        (second pass, same cost: every chain again
        from guess * (1 + 1e-13): `roots_perturbed`, `iters_perturbed` = how far the REFERENCE's own root
        moves under a last-digit change of its input, i.e. which chains carry a root that any two
        correct fp64 implementations can be asked to share to 1e-9)
Output: tests/golden/cfg3_chains.npz  (inputs + expected outputs only)
  guesses[128], iterates[128, 22] (omega after each Newton step, NaN padded), iters[128],
  roots[128], converged[128], info[128] (zsysv info of the failing step, 0 otherwise)
Partial runs are merged into the existing file.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Reference, example_tokamak  # noqa: E402
from scipy.linalg.lapack import zsysv  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg3_chains.npz")


def lattice():
    re = np.linspace(-1.2, -0.4, 16)
    im = np.linspace(0.05, 0.40, 8)
    return (re[None, :] + 1j * im[:, None]).reshape(-1).copy()


def solve(ref, n, tol, limit, cores, g):
    """One chain: (iterates list, converged, zsysv info)."""
    w = 0.99 * g                            # include/solver.h:401
    dw = 0.01 * g                           # :402
    Mold = ref.assemble(n, complex(w), cores)
    w = w + dw                              # :412
    M = ref.assemble(n, complex(w), cores)
    Mp = (M - Mold) / dw                    # :414
    its, ok, inf = [], 0, 0
    for _ in range(limit + 1):              # src/main.cpp:43
        Mold = M
        _, _, x, inf = zsysv(M.copy(), Mp, lower=0)   # include/solver.h:134-136
        dw = -1.0 / np.trace(x)             # :139
        w = w + dw                          # :140
        its.append(w)
        if inf != 0:                        # :142-153 (thrown after omega moved)
            break
        M = ref.assemble(n, complex(w), cores)        # :157
        Mp = (M - Mold) / dw                # :159
        if abs(dw) < abs(tol * w):          # src/main.cpp:53-56
            ok = 1
            break
    return its, ok, inf


def perturb_pass():
    d = example_tokamak(npoints=256, omega_d_coeff=1.01)
    n, tol, limit = d["npoints"], d["iteration_precision"], d["iteration_step_limit"]
    ref = Reference()
    ref.open_dict(d)
    z = dict(np.load(OUT))
    g = z["guesses"]
    rp = z.get("roots_perturbed", np.full(128, np.nan + 1j * np.nan)).copy()
    ip = z.get("iters_perturbed", np.zeros(128, dtype=np.int32)).copy()
    dp = z.get("done_perturbed", np.zeros(128, dtype=np.int32)).copy()
    t00 = time.time()
    for b in range(128):
        if dp[b] or not z["converged"][b]:
            continue
        its, ok, inf = solve(ref, n, tol, limit, os.cpu_count(), g[b] * (1.0 + 1e-13))
        rp[b], ip[b], dp[b] = its[-1], len(its) if ok else -len(its), 1
        print(f"chain {b:3d}: root moves by {abs(rp[b] - z['roots'][b]):.3e} ({ip[b]} vs {z['iters'][b]} steps, "
              f"total {time.time() - t00:.0f} s)", flush=True)
        z["roots_perturbed"], z["iters_perturbed"], z["done_perturbed"] = rp, ip, dp
        np.savez_compressed(OUT, **z)
    print("done")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "perturb":
        return perturb_pass()
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    d = example_tokamak(npoints=256, omega_d_coeff=1.01)
    n, tol, limit = d["npoints"], d["iteration_precision"], d["iteration_step_limit"]
    cores = os.cpu_count()
    ref = Reference()
    ref.open_dict(d)
    g = lattice()
    if os.path.exists(OUT):
        z = np.load(OUT)
        iterates, iters, roots = z["iterates"].copy(), z["iters"].copy(), z["roots"].copy()
        conv, info_a, done = z["converged"].copy(), z["info"].copy(), z["done"].copy()
    else:
        iterates = np.full((128, limit + 2), np.nan + 1j * np.nan)
        iters = np.zeros(128, dtype=np.int32)
        roots = np.full(128, np.nan + 1j * np.nan)
        conv = np.zeros(128, dtype=np.int32)
        info_a = np.zeros(128, dtype=np.int32)
        done = np.zeros(128, dtype=np.int32)
    t00 = time.time()
    for b in range(first, last):
        if done[b]:
            continue
        t0 = time.time()
        its, ok, inf = solve(ref, n, tol, limit, cores, g[b])
        k, w = len(its), its[-1]
        iterates[b, :k] = its
        iters[b], roots[b], conv[b], info_a[b], done[b] = k, w, ok, inf, 1
        print(f"chain {b:3d} guess {g[b]:.4f} -> {w:.12f} in {k} steps, converged={ok} info={inf} "
              f"({time.time() - t0:.1f} s, total {time.time() - t00:.0f} s)", flush=True)
        np.savez_compressed(OUT, guesses=g, iterates=iterates, iters=iters, roots=roots,
                            converged=conv, info=info_a, done=done)
    print("done")


if __name__ == "__main__":
    main()
