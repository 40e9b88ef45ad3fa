"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (fp64, stated per SURVEY §8d): matrix entries 1e-10 * max|M| (observed ~1e-14:
the device uses FMA contraction, reciprocal-multiply complex division and butterfly sums,
the adaptive bisection tree is identical); Newton iterates / roots 1e-9 absolute.
"""
import numpy as np
import pytest

from oracle.binding import example_stellarator, example_tokamak

pytestmark = pytest.mark.gpu

TOL_M = 1e-10
TOL_W = 1e-9


def _ctx(emme, d):
    return emme.Context(emme.params_from_dict(d))


@pytest.mark.parametrize("n", [16, 64])
def test_assemble_tokamak_es(emme, oracle, n):
    d = example_tokamak(npoints=n)
    po = oracle.params(d)
    omegas = np.array([-0.8 + 0.25j, -0.792 + 0.2475j, 0.6 + 0.1j, -0.3 - 0.05j])
    with _ctx(emme, d) as ctx:
        M, iv = ctx.assemble(omegas, want_intervals=True)
    for b, w in enumerate(omegas):
        Mo, tot = oracle.assemble(po, complex(w))
        scale = np.abs(Mo).max()
        assert np.abs(M[b] - Mo).max() <= TOL_M * scale, (b, np.abs(M[b] - Mo).max() / scale)
        assert iv[b] == tot  # same bisection tree, interval for interval


def test_assemble_stellarator_em(emme, oracle):
    d = example_stellarator(npoints=16)
    po = oracle.params(d)
    omegas = np.array([-1.656 + 2.49j, -0.9 + 0.4j])
    with _ctx(emme, d) as ctx:
        assert ctx.dim == 32
        M, iv = ctx.assemble(omegas, want_intervals=True)
    for b, w in enumerate(omegas):
        Mo, tot = oracle.assemble(po, complex(w))
        scale = np.abs(Mo).max()
        assert np.abs(M[b] - Mo).max() <= TOL_M * scale
        assert iv[b] == tot


def test_trace_solve_matches_oracle_and_lapack(emme, oracle):
    rng = np.random.default_rng(7)
    n, nb = 48, 5
    A = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n))
    A = A + np.transpose(A, (0, 2, 1))  # complex symmetric like M
    B = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n))
    with _ctx(emme, example_tokamak(npoints=16)) as ctx:
        tr, info = ctx.trace_solve(A, B)
    assert (info == 0).all()
    for b in range(nb):
        t_or, i_or = oracle.trace_solve(A[b], B[b])
        t_np = np.trace(np.linalg.solve(A[b], B[b]))
        assert i_or == 0
        assert abs(tr[b] - t_or) <= 1e-10 * abs(t_or)
        assert abs(tr[b] - t_np) <= 1e-9 * abs(t_np)


def test_trace_solve_singular_info(emme):
    n = 8
    A = np.eye(n, dtype=np.complex128)
    A[3, 3] = 0.0
    A[:, 3] = 0.0
    B = np.eye(n, dtype=np.complex128)
    with _ctx(emme, example_tokamak(npoints=16)) as ctx:
        tr, info = ctx.trace_solve(A, B)
    assert info[0] == 4  # LAPACK convention: U(4,4) exactly zero
    assert np.isnan(tr[0].real)


def test_newton_step_matches_oracle(emme, oracle):
    d = example_tokamak(npoints=32)
    po = oracle.params(d)
    g = -0.8 + 0.25j
    w0, dw = 0.99 * g, 0.01 * g
    M0, _ = oracle.assemble(po, w0)
    M1, _ = oracle.assemble(po, w0 + dw)
    Mp = (M1 - M0) / dw
    tr, info = oracle.trace_solve(M1, Mp)
    d_or = -1.0 / tr
    w_or = (w0 + dw) + d_or
    Mn, _ = oracle.assemble(po, w_or)
    with _ctx(emme, d) as ctx:
        w, dwn, M, Mpn, info = ctx.newton_step([w0 + dw], M1[None], Mp[None])
    assert info[0] == 0
    assert abs(w[0] - w_or) <= TOL_W
    assert abs(dwn[0] - d_or) <= TOL_W
    assert np.abs(M[0] - Mn).max() <= 1e-8 * np.abs(Mn).max()
    assert np.abs(Mpn[0] - (Mn - M1) / d_or).max() <= 1e-6 * np.abs(Mpn[0]).max()


def test_solve_roots_tokamak_matches_oracle(emme, oracle):
    d = example_tokamak(npoints=64)
    po = oracle.params(d)
    guesses = np.array([-0.8 + 0.25j, -0.7 + 0.3j, -0.9 + 0.2j])
    with _ctx(emme, d) as ctx:
        roots, iters, info, its = ctx.solve_roots(guesses, want_iterates=True)
        Mf = ctx.final_matrix(0)
    assert (info == 0).all()
    for b, g in enumerate(guesses):
        r_or, its_or, Mo, _ = oracle.solve_root(po, complex(g), want_matrix=(b == 0))
        assert iters[b] == len(its_or)
        assert np.abs(its[b, :len(its_or)] - its_or).max() <= TOL_W
        assert abs(roots[b] - r_or) <= TOL_W
        if b == 0:
            assert np.abs(Mf - Mo).max() <= 1e-8 * np.abs(Mo).max()
    # golden (compiled reference incl. LAPACK zsysv), N=64
    assert abs(roots[0] - complex(-0.67067782097052198, 0.27077138768282322)) <= 1e-9
