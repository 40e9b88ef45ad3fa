#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's own compiled
kappa/quadrature sources (oracle/_ref/libemme_ref.so, built by oracle/Makefile from
/root/reference -- the reference's tests hold no fixture for this path).

Run in the build container only:  python tests/golden/make_golden.py
Outputs are plain data (inputs + expected outputs); no reference source text is stored.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Reference, example_stellarator, example_tokamak, json_text  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
ref = Reference()
rng = np.random.default_rng(20250808)


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print("wrote", name, {k: np.shape(v) for k, v in arrs.items()})


# ---- Bessel helper (include/functions.h:381-408) -----------------------------------
z = np.concatenate([
    (rng.normal(size=200) + 1j * rng.normal(size=200)) * 0.3,
    (rng.normal(size=200) + 1j * rng.normal(size=200)) * 3.0,
    (rng.normal(size=100) + 1j * rng.normal(size=100)) * 25.0,
    np.array([1e-3 + 0j, -1e-3 + 1e-4j, 24.9 + 0.1j, -7.5 - 7.5j, 0.999999 + 0j, 1.0 + 0j]),
])
out = np.array([ref.bessel(complex(v)) for v in z])
save("bessel.npz", z=z, out=out)

# ---- adaptive quadrature alone (include/functions.h:305-331) -------------------------
cases = []
for a in [(-1 + 2j), (-0.5 - 3j), (-2 + 0.1j), (-0.05 + 1j), (-3 + 0j)]:
    for pw in [0.0, 1.0, 2.5]:
        for pts in (15, 31):
            for tol, prec in [(1e-6, 1e-6), (1e-5, 1e-2), (1e-11, 1e-12)]:
                val = ref.integrate_test(a, pw, tol, prec, 100, pts)
                cases.append([a.real, a.imag, pw, pts, tol, prec, 100, val.real, val.imag])
save("integrate.npz", cases=np.array(cases))

# ---- weights + grid ------------------------------------------------------------------
save("weights_grid.npz", w12=ref.weights(12), w5=ref.weights(5), w64=ref.weights(64),
     eta33=ref.grid(7.5, 33)[0], dx33=ref.grid(7.5, 33)[1],
     eta64=ref.grid(20.0, 64)[0], dx64=ref.grid(20.0, 64)[1])

# ---- geometry tables g(eta), b(eta) and derived parameters for all five `conf` ------
inputs = {
    "tokamak": example_tokamak(npoints=24),
    "tokamak_em": example_tokamak(npoints=24, beta_e=0.01, epsilon_r=0.1, theta=0.3),
    "stellarator": example_stellarator(npoints=24),
    "cylinder": example_tokamak(npoints=24, conf="cylinder"),
    "taylor": example_tokamak(npoints=24, conf="taloyMagneticDrift", beta_e=0.004),
    "cylinder_old": example_tokamak(npoints=24, conf="cylinder old"),
}
geo = {}
for name, d in inputs.items():
    ref.open_dict(d)
    eta, dx = ref.grid(d["length"], d["npoints"])
    geo[name + "_eta"] = eta
    geo[name + "_g"] = np.array([ref.g(e) for e in eta])
    geo[name + "_b"] = np.array([ref.bi(e) for e in eta])
    pr = ref.params()
    geo[name + "_params"] = np.array([pr[k] for k in Reference.PARAM_NAMES])
save("geometry.npz", **geo)
with open(os.path.join(OUT, "inputs.json"), "w") as f:
    json.dump({"param_names": Reference.PARAM_NAMES, "inputs": inputs}, f, indent=1)

# ---- kappa samples -------------------------------------------------------------------
def kappa_samples(d, omegas, ms, nsamp):
    ref.open_dict(d)
    eta, _ = ref.grid(d["length"], d["npoints"])
    n = d["npoints"]
    rows = []
    for w in omegas:
        for _ in range(nsamp):
            i = int(rng.integers(0, n - 1))
            j = int(rng.integers(i + 1, n))
            for m in ms:
                k = ref.kappa(m, eta[i], eta[j], w)
                ke = ref.kappa_e(m, eta[i], eta[j], w)
                rows.append([m, i, j, eta[i], eta[j], w.real, w.imag, k.real, k.imag, ke.real, ke.imag])
    return np.array(rows)

save("kappa_tokamak.npz",
     rows=kappa_samples(example_tokamak(npoints=64), [-0.8 + 0.25j, 0.5 + 0.1j, -0.6 - 0.21j], [0], 60))
save("kappa_tokamak_tight.npz",
     rows=kappa_samples(example_tokamak(npoints=64, integration_precision=1e-11,
                                        integration_accuracy=1e-12), [-0.8 + 0.25j], [0], 40))
save("kappa_stellarator.npz",
     rows=kappa_samples(example_stellarator(npoints=32), [-1.656 + 2.49j, -0.85 - 0.32j], [0, 1, 2], 25))
save("kappa_tokamak_em.npz",
     rows=kappa_samples(inputs["tokamak_em"], [-0.8 + 0.25j], [0, 1, 2], 30))

# ---- whole matrices (the reference kappa functions under the scatter of
# include/solver.h:439-511, see oracle/ref_harness.cpp::ref_assemble) -------------------
mats = {}
d = example_tokamak(npoints=16)
ref.open_dict(d)
for tag, w in [("a", -0.8 + 0.25j), ("b", 0.45 + 0.05j)]:
    mats["tok16_w" + tag] = np.array([w])
    mats["tok16_M" + tag] = ref.assemble(16, w)
d = example_stellarator(npoints=8)
ref.open_dict(d)
mats["stel8_w"] = np.array([-1.656 + 2.49j])
mats["stel8_M"] = ref.assemble(16, -1.656 + 2.49j)
d = inputs["tokamak_em"]
d12 = dict(d, npoints=12)
ref.open_dict(d12)
mats["tokem12_w"] = np.array([-0.8 + 0.25j])
mats["tokem12_M"] = ref.assemble(24, -0.8 + 0.25j)
save("matrices.npz", **mats)

# checksums of full-size matrices (too big to store): N=64 and N=256 tokamak at the guess
chk = {}
for n in (64, 256):
    d = example_tokamak(npoints=n)
    ref.open_dict(d)
    M = ref.assemble(n, -0.8 * 0.99 + 0.01 * -0.8 + 1j * (0.25 * 0.99 + 0.01 * 0.25))
    chk[str(n)] = {"omega": [-0.8 * 0.99 + 0.01 * -0.8, 0.25 * 0.99 + 0.01 * 0.25],
                   "sum": [M.sum().real, M.sum().imag], "fro": float(np.linalg.norm(M)),
                   "m01": [M[0, 1].real, M[0, 1].imag],
                   "row_abs_sums_first8": np.abs(M).sum(axis=1)[:8].tolist(),
                   "diag_offsets": {str(k): [complex(M[5, 5 + k]).real, complex(M[5, 5 + k]).imag]
                                    for k in (1, 2, 6, 30)}}
with open(os.path.join(OUT, "matrix_checksums.json"), "w") as f:
    json.dump(chk, f, indent=1)

# ---- parser quirks: what the reference's JSON reader makes of odd number spellings -----
quirk_text = json_text(example_tokamak()).replace('"integration_precision": 1.0e-06',
                                                  '"integration_precision": 1e-6')
quirk_text = quirk_text.replace('"npoints": 64', '"npoints": 48.0')
quirk_text = quirk_text.replace('"arc_coeff": 100.0', '"arc_coeff": 1.e2')
quirk_text = quirk_text.replace('"theta": 0.0', '"theta": -.25')
ref.open(quirk_text)
pr = ref.params()
with open(os.path.join(OUT, "parser_quirks.json"), "w") as f:
    json.dump({"text": quirk_text, "expected": pr}, f, indent=1)
print("done")
