#!/usr/bin/env python3
"""Checksums of full-size N=512 tokamak matrices (BASELINE configs[4]: the 512-point grid of the
(k_rho, omega) sweep) from the reference's own kappa sources (oracle/_ref), at two k_rho values of the
sweep and omegas on both sides of Im omega = 0.  Run in the build container:
  python tests/golden/make_golden_n512.py   ->  tests/golden/matrix_checksums_n512.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Reference, example_tokamak  # noqa: E402

ref = Reference()
rng = np.random.default_rng(5)
idx = [(int(a), int(b)) for a, b in zip(rng.integers(0, 512, 40), rng.integers(0, 512, 40))]
out = {}
for tag, kr, w in (("kr0.2", 0.2, -0.75 + 0.25j), ("kr0.5", 0.5, -0.6 - 0.15j)):
    d = example_tokamak(npoints=512, k_rho=kr)
    ref.open_dict(d)
    M = ref.assemble(512, w, os.cpu_count())
    out[tag] = {"k_rho": kr, "omega": [w.real, w.imag], "sum": [M.sum().real, M.sum().imag],
                "fro": float(np.linalg.norm(M)), "max_abs": float(np.abs(M).max()),
                "row_abs_sums_first8": np.abs(M).sum(axis=1)[:8].tolist(),
                "entries": [[i, j, M[i, j].real, M[i, j].imag] for i, j in idx]}
    print(tag, out[tag]["sum"], out[tag]["fro"], flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "matrix_checksums_n512.json"), "w"), indent=1)
