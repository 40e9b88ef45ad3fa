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
