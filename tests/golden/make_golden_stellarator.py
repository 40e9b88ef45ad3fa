#!/usr/bin/env python3
"""Golden data for BASELINE configs[3] (stellarator, electromagnetic, GK31), computed in the build
container with the reference's own kappa sources (oracle/_ref):

  matrix_checksums_stellarator.json  -- full-size N=256 (dim 512) assemblies at two omegas, the shipped
      guess (Im omega > 0, ~1 interval per integral) and a damped omega the N=128 chain of SURVEY.md
      App. B ends on (Im omega < 0, ~13 intervals per integral): sum, Frobenius norm, sampled entries of
      all four blocks, row sums -- the 4 MiB matrices themselves are too big to commit.
  stellarator_k8.npz -- FIXED-WORK chains: K = 8 trace-secant Newton steps per guess at
      N = 32 and N = 48 (LAPACK zsysv), guesses on both sides of Im omega = 0, every iterate.

Run:  python tests/golden/make_golden_stellarator.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Reference, example_stellarator  # noqa: E402
from scipy.linalg.lapack import zsysv  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
ref = Reference()
cores = os.cpu_count()

# ---- K = 8 fixed-step chains --------------------------------------------------------------------
K = 8
cases = {}
for n, guesses in ((32, [-1.656 + 2.490j, -0.9 - 0.04j, -0.75 - 0.37j, -1.2 + 0.5j]),
                   (48, [-1.656 + 2.490j, -0.85 - 0.30j])):
    d = example_stellarator(npoints=n)
    ref.open_dict(d)
    its = np.full((len(guesses), K), np.nan + 1j * np.nan)
    for b, g in enumerate(guesses):
        w, dw = 0.99 * g, 0.01 * g
        Mold = ref.assemble(2 * n, complex(w), cores)
        w = w + dw
        M = ref.assemble(2 * n, complex(w), cores)
        Mp = (M - Mold) / dw
        for k in range(K):
            Mold = M
            _, _, x, info = zsysv(M.copy(), Mp, lower=0)
            assert info == 0
            dw = -1.0 / np.trace(x)
            w = w + dw
            M = ref.assemble(2 * n, complex(w), cores)
            Mp = (M - Mold) / dw
            its[b, k] = w
        print(n, g, "->", its[b], flush=True)
    cases[f"n{n}_guesses"] = np.array(guesses)
    cases[f"n{n}_iterates"] = its
np.savez_compressed(os.path.join(OUT, "stellarator_k8.npz"), **cases)

# ---- full-size checksums ------------------------------------------------------------------------
d = example_stellarator(npoints=256)
ref.open_dict(d)
chk = {}
rng = np.random.default_rng(4)
idx = [(int(a), int(b)) for a, b in zip(rng.integers(0, 512, 40), rng.integers(0, 512, 40))]
for tag, w in (("guess", -1.656 + 2.490j), ("damped", -0.855574 - 0.320125j)):
    M = ref.assemble(512, w, cores)
    chk[tag] = {"omega": [w.real, w.imag], "sum": [M.sum().real, M.sum().imag], "fro": float(np.linalg.norm(M)),
                "max_abs": float(np.abs(M).max()),
                "block_sums": {k: [v.real, v.imag] for k, v in
                               (("A", M[:256, :256].sum()), ("B", M[:256, 256:].sum()),
                                ("C", M[256:, :256].sum()), ("D", M[256:, 256:].sum()))},
                "row_abs_sums_first8": np.abs(M).sum(axis=1)[:8].tolist(),
                "row_abs_sums_last8": np.abs(M).sum(axis=1)[-8:].tolist(),
                "entries": [[i, j, M[i, j].real, M[i, j].imag] for i, j in idx]}
    print(tag, chk[tag]["sum"], chk[tag]["fro"], flush=True)
with open(os.path.join(OUT, "matrix_checksums_stellarator.json"), "w") as f:
    json.dump(chk, f, indent=1)
print("done")
