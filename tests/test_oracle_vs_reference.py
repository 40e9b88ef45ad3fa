"""Oracle vs the reference's own compiled sources (oracle/_ref), wider random sampling.
Skipped where _ref was not built (it is built wherever /root/reference exists and travels
to the GPU box as a built file)."""
import numpy as np

from oracle.binding import example_stellarator, example_tokamak


def test_bessel_random(oracle, reference):
    rng = np.random.default_rng(1)
    for s in (0.05, 1.0, 8.0, 40.0):
        for _ in range(300):
            z = complex(*(rng.normal(size=2) * s))
            assert np.array_equal(oracle.bessel(z), reference.bessel(z))


def test_kappa_random_pairs_tokamak(oracle, reference):
    d = example_tokamak(npoints=96, omega_d_coeff=0.71, theta=0.2, epsilon_r=0.05)
    reference.open_dict(d)
    p = oracle.params(d)
    eta, _ = oracle.grid(p.length, p.npoints)
    rng = np.random.default_rng(2)
    for _ in range(120):
        i = int(rng.integers(0, 95))
        j = int(rng.integers(i + 1, 96))
        w = complex(rng.uniform(-1.2, 1.2), rng.uniform(-0.3, 0.5))
        k, _ = oracle.kappa(p, 0, eta[i], eta[j], w)
        assert k == reference.kappa(0, eta[i], eta[j], w)


def test_whole_matrix_bit_exact_n48(oracle, reference):
    d = example_tokamak(npoints=48)
    reference.open_dict(d)
    p = oracle.params(d)
    w = -0.8 + 0.25j
    M, _ = oracle.assemble(p, w)
    assert np.array_equal(M, reference.assemble(48, w))


def test_stellarator_matrix(oracle, reference):
    d = example_stellarator(npoints=12)
    reference.open_dict(d)
    p = oracle.params(d)
    w = -0.9 + 0.4j
    M, _ = oracle.assemble(p, w)
    Mr = reference.assemble(24, w)
    assert np.abs(M - Mr).max() <= 1e-12 * np.abs(Mr).max()


def test_bad_start_points_error(oracle, reference):
    d = example_tokamak(npoints=8, integration_start_points=21)
    p = oracle.params(d)
    k, n = oracle.kappa(p, 0, -1.0, 1.0, -0.8 + 0.25j)
    assert n == -1 and np.isnan(k.real)  # reference throws (include/functions.h:329)


def test_qr_secant_restatement_matches_its_closed_form():
    """Oracle.qr_secant (the reference's zgeqp3/ztrtrs/zunmqr sequence, include/solver.h:210-383)
    against the algebraic identity it implements: with p the last pivot column and v the vector
    with v[p] = 1 and A v parallel to the last column of Q, d omega = -1 / (A^-1 B v)[p]."""
    import numpy as np
    from scipy.linalg import qr
    from oracle.binding import Oracle
    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 48):
        A = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
        B = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
        dw, info = Oracle.qr_secant(A, B)
        assert info == 0
        p = qr(A, pivoting=True)[2][-1]
        e = np.zeros(n)
        e[p] = 1.0
        v = np.linalg.solve(A.conj().T @ A, e)
        v = v / v[p]
        want = -1.0 / np.linalg.solve(A, B @ v)[p]
        assert abs(dw - want) <= 1e-9 * abs(want)


def test_qr_and_trace_variants_reach_the_same_root(oracle):
    """both Newton variants converge to the same root (checked there on the
    complete reference at N=64: (-0.670678, 0.270771); the restatement must land there too)."""
    from oracle.binding import example_tokamak
    d = example_tokamak(npoints=64)
    p = oracle.params(d)
    g = complex(*d["initial_guess"])
    w_tr = oracle.solve_root(p, g)[0]
    w_qr, its = oracle.solve_root_qr(p, g)
    assert len(its) <= p.iteration_step_limit + 1
    assert abs(w_qr - w_tr) <= 1e-5 * abs(w_tr)
    assert abs(w_qr - complex(-0.670678, 0.270771)) <= 2e-6
