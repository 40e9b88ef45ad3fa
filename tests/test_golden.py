"""The oracle (oracle/emme_oracle.c) against the committed golden vectors.

Fixtures under tests/golden/*.npz were produced by tests/golden/make_golden.py from the
reference's own compiled kappa sources; survey_appendix_b.json holds numbers printed by the
complete reference binary during the survey.  These tests need neither /root/reference nor
a GPU, so they pin the oracle on the GPU box as well.
"""
import json
import os

import numpy as np
import pytest

from oracle.binding import example_stellarator, example_tokamak

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_bessel_bit_exact(oracle):
    f = load("bessel.npz")
    for z, want in zip(f["z"], f["out"]):
        got = oracle.bessel(complex(z))
        assert np.array_equal(got, want), z


def test_quadrature_bit_exact(oracle):
    for ar, ai, pw, pts, tol, prec, ms, vr, vi in load("integrate.npz")["cases"]:
        got, n = oracle.integrate_test(complex(ar, ai), pw, tol, prec, int(ms), int(pts))
        assert got == complex(vr, vi)
        assert n >= 1


def test_weights_and_grid(oracle):
    f = load("weights_grid.npz")
    for n in (5, 12, 64):
        assert np.array_equal(oracle.weights(n), f[f"w{n}"])
    eta, dx = oracle.grid(7.5, 33)
    assert np.array_equal(eta, f["eta33"]) and dx == float(f["dx33"])
    eta, dx = oracle.grid(20.0, 64)
    assert np.array_equal(eta, f["eta64"]) and dx == float(f["dx64"])


def test_geometry_tables_and_derived_params(oracle):
    f = load("geometry.npz")
    meta = json.load(open(os.path.join(G, "inputs.json")))
    for name, d in meta["inputs"].items():
        p = oracle.params(d)
        want = dict(zip(meta["param_names"], f[name + "_params"]))
        for k, v in want.items():
            assert getattr(p, k) == v, (name, k)
        eta = f[name + "_eta"]
        g = np.array([oracle.g(p, e) for e in eta])
        b = np.array([oracle.bi(p, e) for e in eta])
        assert np.array_equal(b, f[name + "_b"]), name
        if name == "stellarator":
            # closed form of the machine-expanded expression (see emme_oracle.c)
            assert np.abs(g - f[name + "_g"]).max() <= 2e-13 * np.abs(f[name + "_g"]).max()
        else:
            assert np.array_equal(g, f[name + "_g"]), name


@pytest.mark.parametrize("fixture,make,exact", [
    ("kappa_tokamak.npz", lambda: example_tokamak(npoints=64), True),
    ("kappa_tokamak_tight.npz", lambda: example_tokamak(npoints=64, integration_precision=1e-11,
                                                         integration_accuracy=1e-12), True),
    ("kappa_stellarator.npz", lambda: example_stellarator(npoints=32), False),
])
def test_kappa(oracle, fixture, make, exact):
    p = oracle.params(make())
    for m, i, j, e1, e2, wr, wi, kr, ki, ker, kei in load(fixture)["rows"]:
        w = complex(wr, wi)
        k, nint = oracle.kappa(p, int(m), e1, e2, w)
        ke = oracle.kappa_e(p, int(m), e1, e2, w)
        if exact:
            assert k == complex(kr, ki) and ke == complex(ker, kei)
            k2, _ = oracle.kappa(p, int(m), e1, e2, w, recompute=1)
            assert k2 == k
        else:  # stellarator g differs from the expanded form in the last bits
            assert abs(k - complex(kr, ki)) <= 1e-12 * max(1.0, abs(complex(kr, ki)))
            assert abs(ke - complex(ker, kei)) <= 1e-12 * max(1.0, abs(complex(ker, kei)))


def test_kappa_tokamak_em(oracle):
    meta = json.load(open(os.path.join(G, "inputs.json")))
    p = oracle.params(meta["inputs"]["tokamak_em"])
    for m, i, j, e1, e2, wr, wi, kr, ki, ker, kei in load("kappa_tokamak_em.npz")["rows"]:
        w = complex(wr, wi)
        k, _ = oracle.kappa(p, int(m), e1, e2, w)
        assert k == complex(kr, ki)
        assert oracle.kappa_e(p, int(m), e1, e2, w) == complex(ker, kei)


def test_whole_matrices(oracle):
    f = load("matrices.npz")
    p = oracle.params(example_tokamak(npoints=16))
    for tag in "ab":
        M, _ = oracle.assemble(p, complex(f["tok16_w" + tag][0]), nthreads=2)
        assert np.array_equal(M, f["tok16_M" + tag])
    p = oracle.params(example_stellarator(npoints=8))
    M, _ = oracle.assemble(p, complex(f["stel8_w"][0]), nthreads=2)
    assert np.abs(M - f["stel8_M"]).max() <= 1e-12 * np.abs(f["stel8_M"]).max()
    meta = json.load(open(os.path.join(G, "inputs.json")))
    p = oracle.params(dict(meta["inputs"]["tokamak_em"], npoints=12))
    M, _ = oracle.assemble(p, complex(f["tokem12_w"][0]), nthreads=2)
    assert np.array_equal(M, f["tokem12_M"])
    # block structure of the EM matrix (include/solver.h:492-504)
    n = 12
    A, B, Cb, D = M[:n, :n], M[:n, n:], M[n:, :n], M[n:, n:]
    assert np.array_equal(A, A.T) and np.array_equal(D, D.T)
    assert np.array_equal(B, -B.T) and np.array_equal(Cb, -B)


def test_matrix_checksum_n64_and_survey_golden(oracle):
    chk = json.load(open(os.path.join(G, "matrix_checksums.json")))["64"]
    sv = json.load(open(os.path.join(G, "survey_appendix_b.json")))["n64"]
    p = oracle.params(example_tokamak(npoints=64))
    M, _ = oracle.assemble(p, complex(*chk["omega"]))
    assert [M.sum().real, M.sum().imag] == chk["sum"]
    assert [M[0, 1].real, M[0, 1].imag] == chk["m01"] == sv["M_guess_01"]
    assert abs(M.sum() - complex(*sv["M_guess_sum"])) < 1e-11
    assert abs(np.linalg.norm(M) - sv["M_guess_fro"]) < 1e-12


def test_newton_iterates_match_complete_reference_n64(oracle):
    """Trace-secant chain with our partial-pivot LU vs the reference's zsysv chain."""
    sv = json.load(open(os.path.join(G, "survey_appendix_b.json")))["n64"]
    p = oracle.params(example_tokamak(npoints=64))
    root, its, Mf, _ = oracle.solve_root(p, -0.8 + 0.25j, want_matrix=True)
    want = np.array([complex(*z) for z in sv["iterates"]])
    assert len(its) == len(want)
    assert np.abs(its - want).max() < 1e-12
    assert abs(Mf.sum() - complex(*sv["M_final_sum"])) < 1e-10
    assert abs(np.linalg.norm(Mf) - sv["M_final_fro"]) < 1e-11


def test_trace_solve_against_lapack_zsysv(oracle):
    """The LAPACK routine the reference calls (include/solver.h:134-136), via SciPy."""
    from scipy.linalg.lapack import zsysv
    p = oracle.params(example_tokamak(npoints=24))
    g = -0.8 + 0.25j
    M0, _ = oracle.assemble(p, 0.99 * g, nthreads=2)
    M1, _ = oracle.assemble(p, 0.99 * g + 0.01 * g, nthreads=2)
    Mp = (M1 - M0) / (0.01 * g)
    tr, info = oracle.trace_solve(M1, Mp)
    _, _, x, linfo = zsysv(M1, Mp, lower=0)
    assert info == 0 and linfo == 0
    assert abs(tr - np.trace(x)) <= 1e-11 * abs(tr)


def test_trace_solve_singular(oracle):
    A = np.eye(6, dtype=np.complex128)
    A[:, 2] = 0
    tr, info = oracle.trace_solve(A, np.eye(6))
    assert info == 3 and np.isnan(tr.real)
