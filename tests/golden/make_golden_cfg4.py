#!/usr/bin/env python3
"""Golden FIXED-WORK chains at full size for BASELINE configs[3] (bench.py --config 4): stellarator,
electromagnetic, GK31, N = 256 (dim 512), K = 8 trace-secant Newton steps from three guesses of the 32 x 32
lattice around the shipped initial guess (two corners and the centre), every iterate -- reference kappa
sources (oracle/_ref) for the fill, LAPACK zsysv for the step, like make_golden_stellarator.py at N = 32/48.
Run in the build container (a few minutes on 8 cores):
  python tests/golden/make_golden_cfg4.py   ->  tests/golden/cfg4_k8_n256.npz  (guesses[3], iterates[3, 8])"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Reference, example_stellarator  # noqa: E402
from scipy.linalg.lapack import zsysv  # noqa: E402

K, n = 8, 256
re, im = np.linspace(-1.756, -1.556, 32), np.linspace(2.39, 2.59, 32)
guesses = np.array([re[0] + 1j * im[0], re[15] + 1j * im[16], re[31] + 1j * im[31]])
ref = Reference()
ref.open_dict(example_stellarator(npoints=n))
cores = os.cpu_count()
its = np.full((len(guesses), K), np.nan + 1j * np.nan)
t0 = time.time()
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
    print(g, "->", its[b], f"({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg4_k8_n256.npz"), guesses=guesses, iterates=its)
