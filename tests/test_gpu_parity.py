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


def _ctx(emme, d, **options):
    return emme.Context(emme.params_from_dict(d), **options)


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


# ---- committed golden fixtures (reference's own compiled kappa sources) -------------------
import json
import os

G = os.path.join(os.path.dirname(__file__), "golden")


def test_golden_matrices_on_device(emme):
    f = np.load(os.path.join(G, "matrices.npz"), allow_pickle=False)
    meta = json.load(open(os.path.join(G, "inputs.json")))
    with _ctx(emme, example_tokamak(npoints=16)) as ctx:
        M = ctx.assemble([complex(f["tok16_wa"][0]), complex(f["tok16_wb"][0])])
    for k, tag in enumerate("ab"):
        want = f["tok16_M" + tag]
        assert np.abs(M[k] - want).max() <= TOL_M * np.abs(want).max()
    with _ctx(emme, example_stellarator(npoints=8)) as ctx:
        M = ctx.assemble([complex(f["stel8_w"][0])])
    assert np.abs(M[0] - f["stel8_M"]).max() <= TOL_M * np.abs(f["stel8_M"]).max()
    with _ctx(emme, dict(meta["inputs"]["tokamak_em"], npoints=12)) as ctx:
        M = ctx.assemble([complex(f["tokem12_w"][0])])
    assert np.abs(M[0] - f["tokem12_M"]).max() <= TOL_M * np.abs(f["tokem12_M"]).max()


def test_full_size_n256_against_reference_checksums_and_golden_root(emme):
    """BASELINE configs[1]: 256-point grid, single root.  Whole-matrix checksums come from
    the reference's compiled kappa sources (tests/golden/matrix_checksums.json); iterates and
    root from the complete reference run of SURVEY.md App. B."""
    chk = json.load(open(os.path.join(G, "matrix_checksums.json")))["256"]
    sv = json.load(open(os.path.join(G, "survey_appendix_b.json")))["n256"]
    with _ctx(emme, example_tokamak(npoints=256)) as ctx:
        M = ctx.assemble([complex(*chk["omega"])])[0]
        assert abs(M.sum() - complex(*chk["sum"])) <= 1e-10 * abs(complex(*chk["sum"]))
        assert abs(np.linalg.norm(M) - chk["fro"]) <= 1e-11 * chk["fro"]
        assert abs(M[0, 1] - complex(*chk["m01"])) <= 1e-13
        assert np.abs(np.abs(M).sum(axis=1)[:8] - chk["row_abs_sums_first8"]).max() <= 1e-11
        for k, v in chk["diag_offsets"].items():
            assert abs(M[5, 5 + int(k)] - complex(*v)) <= 1e-13
        assert np.array_equal(M, M.T)  # exactly symmetric: both halves come from one integral
        roots, iters, info, its = ctx.solve_roots([-0.8 + 0.25j], want_iterates=True)
        Mf = ctx.final_matrix(0)
    want = np.array([complex(*z) for z in sv["iterates"]])
    assert iters[0] == len(want) and info[0] == 0
    assert np.abs(its[0, :len(want)] - want).max() <= TOL_W
    assert abs(Mf.sum() - complex(*sv["M_final_sum"])) <= 1e-8
    # size-independent property: M(w_root) is numerically singular
    s = np.linalg.svd(Mf, compute_uv=False)
    assert s[-1] / s[0] < 1e-6


def test_golden_scan_roots_n64(emme):
    """11-point omega_d_coeff scan of the shipped example (6-digit prints of the reference,
    SURVEY.md App. B), each from the reference's continuation guess."""
    sv = json.load(open(os.path.join(G, "survey_appendix_b.json")))["n64_omega_d_coeff_scan_6digits"]
    guess = -0.8 + 0.25j
    for key in ["1.01", "0.91", "0.81", "0.71", "0.61", "0.51", "0.41", "0.31", "0.21", "0.11", "0.01"]:
        with _ctx(emme, example_tokamak(npoints=64, omega_d_coeff=float(key))) as ctx:
            roots, iters, info = ctx.solve_roots([guess])
        assert info[0] == 0
        want = complex(*sv[key])
        assert abs(roots[0] - want) <= 6e-6 * max(1.0, abs(want)), (key, roots[0], want)
        guess = complex(roots[0])  # omega continuation (src/main.cpp:78)


def test_stellarator_root_search_matches_oracle(emme, oracle):
    d = example_stellarator(npoints=16, iteration_step_limit=30)
    po = oracle.params(d)
    g = -1.656 + 2.49j
    r_or, its_or, _, _ = oracle.solve_root(po, g)
    with _ctx(emme, d) as ctx:
        roots, iters, info, its = ctx.solve_roots([g], want_iterates=True)
    assert info[0] == 0 and iters[0] == len(its_or)
    assert np.abs(its[0, :len(its_or)] - its_or).max() <= 1e-8
    assert abs(roots[0] - r_or) <= 1e-8


@pytest.mark.parametrize("n", [2, 3, 5, 33])
def test_small_and_odd_grids(emme, oracle, n):
    d = example_tokamak(npoints=n)
    po = oracle.params(d)
    w = -0.8 + 0.25j
    with _ctx(emme, d) as ctx:
        M = ctx.assemble([w])[0]
    Mo, _ = oracle.assemble(po, w, nthreads=2)
    assert np.abs(M - Mo).max() <= TOL_M * np.abs(Mo).max()


def test_batch_items_are_independent(emme):
    """Ragged batches: an item's matrix does not depend on what else is in the batch."""
    ws = np.array([-0.8 + 0.25j, 0.3 + 0.2j, -0.5 - 0.1j, -1.1 + 0.05j, 0.9 + 0.4j])
    with _ctx(emme, example_tokamak(npoints=24)) as ctx:
        Mall = ctx.assemble(ws)
        for k in (0, 3):
            # a single item goes through the lanes-are-nodes kernel, a batch through the
            # omega-lane kernel: same tree, last-bit differences only
            one = ctx.assemble(ws[k:k + 1])[0]
            assert np.abs(one - Mall[k]).max() <= 1e-13 * np.abs(one).max()
        assert np.array_equal(ctx.assemble(ws[::-1])[::-1], Mall)  # order inside a batch
        r5, i5, f5 = ctx.solve_roots(ws)
        r1, i1, f1 = ctx.solve_roots(ws[2:3])
    assert abs(r1[0] - r5[2]) <= 1e-11 and i1[0] == i5[2]


def test_tight_tolerance_input(emme, oracle):
    """1e-11/1e-12 quadrature tolerances: deep trees, no room for branch flips."""
    d = example_tokamak(npoints=12, integration_precision=1e-11, integration_accuracy=1e-12)
    po = oracle.params(d)
    w = -0.8 + 0.25j
    with _ctx(emme, d) as ctx:
        M, iv = ctx.assemble([w], want_intervals=True)
    Mo, tot = oracle.assemble(po, w, nthreads=2)
    assert iv[0] == tot
    assert np.abs(M[0] - Mo).max() <= 1e-11 * np.abs(Mo).max()


def test_positive_real_omega_uses_other_contour(emme, oracle):
    """omi = -sign(Re w) flips the contour rotation (src/Parameters.cpp:121)."""
    d = example_tokamak(npoints=20)
    po = oracle.params(d)
    for w in (0.7 + 0.2j, -0.7 + 0.2j, 0.0 + 0.3j):
        with _ctx(emme, d) as ctx:
            M = ctx.assemble([w])[0]
        Mo, _ = oracle.assemble(po, w, nthreads=2)
        assert np.abs(M - Mo).max() <= TOL_M * np.abs(Mo).max()


def test_device_resident_buffers_and_stream(emme):
    """The boundary takes device pointers and a caller stream (torch is only plumbing)."""
    import torch
    d = example_tokamak(npoints=16)
    with _ctx(emme, d) as ctx:
        host = ctx.assemble([-0.8 + 0.25j])[0]
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            ctx.set_stream(s.cuda_stream)
            buf = torch.zeros((1, 16, 16), dtype=torch.complex128, device="cuda")
            ctx.assemble([-0.8 + 0.25j], out_device_ptr=buf.data_ptr())
            s.synchronize()
        assert np.array_equal(buf.cpu().numpy()[0], host)


@pytest.mark.parametrize("n", [5, 16, 33, 100, 256, 512, 600, 777, 1024, 1100])
def test_trace_solve_sizes_against_lapack(emme, n):
    """Blocked LU (NB=16) incl. ragged last block, vs numpy (LAPACK zgesv); above n = 560 the
    chunked multi-workgroup build (panel rows in chunks of 512), above 1024 the unblocked kernel."""
    rng = np.random.default_rng(n)
    nb = 3
    A = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n))
    A = A + np.transpose(A, (0, 2, 1)) + 0.5 * n ** 0.5 * np.eye(n)
    B = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n))
    with _ctx(emme, example_tokamak(npoints=16)) as ctx:
        tr, info = ctx.trace_solve(A, B)
    assert (info == 0).all()
    for b in range(nb):
        want = np.trace(np.linalg.solve(A[b], B[b]))
        assert abs(tr[b] - want) <= 1e-10 * max(1.0, abs(want)), (n, b, tr[b], want)


@pytest.mark.parametrize("n,nwg", [(256, 2), (256, 4), (200, 3), (130, 2), (37, 2), (512, 4), (16, 8),
                                   (256, 6), (200, 7), (512, 8), (100, 5), (600, 2), (777, 5), (1024, 8),
                                   (512, 12), (300, 16), (900, 13)])
def test_trace_solve_several_workgroups_per_matrix(emme, n, nwg):
    """The LU with 1 + S workgroups per matrix (role 0 factors A, the others carry B's columns,
    all share the back substitution; from 4 workgroups on with look-ahead: one or two of them
    carry A's trailing columns): bit-identical to the one-workgroup launch, incl. ragged
    sizes, a singular neighbour and sizes below one panel per helper."""
    rng = np.random.default_rng(1000 * n + nwg)
    nb = 5
    A = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n))
    A = A + np.transpose(A, (0, 2, 1)) + 0.5 * n ** 0.5 * np.eye(n)
    B = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n))
    A[3, :, n // 2] = 0.0  # matrix 3 is exactly singular at column n/2 + 1
    with _ctx(emme, example_tokamak(npoints=16)) as ctx:
        ctx.set_options(lu_split=1)
        tr1, info1 = ctx.trace_solve(A, B)
        ctx.set_options(lu_split=nwg)
        trs, infos = ctx.trace_solve(A, B)
    assert info1[3] == n // 2 + 1 and np.array_equal(info1, infos)
    assert np.isnan(trs[3].real) and np.isnan(tr1[3].real)
    ok = np.arange(nb) != 3
    assert (info1[ok] == 0).all()
    if n <= 560:
        assert np.array_equal(tr1[ok].view(np.float64), trs[ok].view(np.float64))  # bit for bit
    else:  # one workgroup per matrix means the unblocked kernel up there: same numbers, other rounding
        assert np.abs(tr1[ok] - trs[ok]).max() <= 1e-10 * np.abs(tr1[ok]).max()
    for b in np.flatnonzero(ok):
        want = np.trace(np.linalg.solve(A[b], B[b]))
        assert abs(trs[b] - want) <= 1e-10 * max(1.0, abs(want)), (n, b, trs[b], want)


@pytest.mark.parametrize("n,nwg", [(512, 2), (400, 2), (448, 3), (130, 2), (70, 2), (64, 2), (37, 3), (520, 2)])
def test_trace_solve_grouped_trailing_updates_give_the_same_bits(emme, n, nwg):
    """The delayed trailing update (four panels per pass over the trailing matrix, 2-3 workgroups per
    matrix without look-ahead; default from n = 384, forced here for every n): bit-identical to the
    one-workgroup launch -- full and ragged groups, a group that is the whole matrix, a singular
    neighbour -- and switched off it is the per-panel kernel again."""
    rng = np.random.default_rng(77 * n + nwg)
    nb = 4
    A = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n))
    A = A + np.transpose(A, (0, 2, 1)) + 0.5 * n ** 0.5 * np.eye(n)
    B = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n))
    A[2, :, (2 * n) // 3] = 0.0  # matrix 2 is exactly singular
    with _ctx(emme, example_tokamak(npoints=16)) as ctx:
        ctx.set_options(lu_split=1)
        tr1, info1 = ctx.trace_solve(A, B)
        ctx.set_options(lu_split=nwg)
        ctx.set_options(lu_group_min_n=16)
        trg, infog = ctx.trace_solve(A, B)
        ctx.set_options(lu_group_min_n=-1)
        tr0, info0 = ctx.trace_solve(A, B)
    assert info1[2] == (2 * n) // 3 + 1 and np.array_equal(info1, infog) and np.array_equal(info1, info0)
    ok = np.arange(nb) != 2
    assert (info1[ok] == 0).all() and np.isnan(trg[2].real)
    assert np.array_equal(tr1[ok].view(np.float64), trg[ok].view(np.float64))
    assert np.array_equal(tr1[ok].view(np.float64), tr0[ok].view(np.float64))
    for b in np.flatnonzero(ok):
        want = np.trace(np.linalg.solve(A[b], B[b]))
        assert abs(trg[b] - want) <= 1e-10 * max(1.0, abs(want)), (n, b, trg[b], want)


def test_trace_solve_chunked_panel_is_independent_of_workgroups(emme):
    """n = 1024 (L21 panel in chunks of 512 rows): 2, 4, 8 and 16 workgroups per matrix give the
    same bits (with / without look-ahead, one to five A-helpers)."""
    rng = np.random.default_rng(99)
    n, nb = 1024, 3
    A = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n)) + 0.5 * n ** 0.5 * np.eye(n)
    B = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n))
    res = []
    with _ctx(emme, example_tokamak(npoints=16)) as ctx:
        for nwg in ("2", "4", "8", "16"):
            ctx.set_options(lu_split=int(nwg))
            res.append(ctx.trace_solve(A, B))
    for tr, info in res:
        assert (info == 0).all()
        assert np.array_equal(tr.view(np.float64), res[0][0].view(np.float64))
    want = np.trace(np.linalg.solve(A[0], B[0]))
    assert abs(res[0][0][0] - want) <= 1e-10 * abs(want)


def test_trace_solve_split_falls_back_when_the_grid_cannot_be_resident(emme):
    """More workgroups than the device holds at once would wait for each other forever: the
    launcher must notice and use one workgroup per matrix (same bits, no time-out codes)."""
    rng = np.random.default_rng(5)
    n, nb = 32, 100  # 100 matrices x 4 workgroups > 256 compute units x 1
    A = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n)) + 3.0 * np.eye(n)
    B = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n))
    with _ctx(emme, example_tokamak(npoints=16)) as ctx:
        ctx.set_options(lu_split=1)
        tr1, info1 = ctx.trace_solve(A, B)
        ctx.set_options(lu_split=4)
        tr4, info4 = ctx.trace_solve(A, B)
    assert (info1 == 0).all() and (info4 == 0).all()
    assert np.array_equal(tr1.view(np.float64), tr4.view(np.float64))


def test_lu_hand_over_time_out_is_bounded_and_repaired(emme):
    """A helper workgroup that gives up waiting (here: after ONE poll) retires its matrix with
    EMME_EDEVICE instead of hanging; a direct call reports that per item, a root search repeats
    itself with one workgroup per matrix and returns the same bits as if it had never split."""
    rng = np.random.default_rng(11)
    n, nb = 64, 6
    A = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n)) + 4.0 * np.eye(n)
    B = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n))
    d = example_tokamak(npoints=64)
    guesses = np.array([-0.8 + 0.25j, -0.7 + 0.3j, -0.9 + 0.2j])
    with _ctx(emme, d) as ctx:
        for _ in range(3):  # node cache built and settled
            ctx.solve_roots(guesses)
        ctx.set_options(lu_split=1)
        tr1, info1 = ctx.trace_solve(A, B)
        r1, it1, i1 = ctx.solve_roots(guesses)
        ctx.set_options(lu_split=3)
        ctx.set_options(lu_spin_limit=1)
        tr3, info3 = ctx.trace_solve(A, B)
        assert set(np.unique(info3)) <= {0, -3}  # EMME_EDEVICE
        ok = info3 == 0
        assert np.array_equal(tr1[ok].view(np.float64), tr3[ok].view(np.float64))
        assert np.isnan(tr3[~ok].real).all()
        r3, it3, i3 = ctx.solve_roots(guesses)
    assert (i1 == 0).all() and np.array_equal(i1, i3) and np.array_equal(it1, it3)
    assert np.array_equal(r1.view(np.float64), r3.view(np.float64))


@pytest.mark.parametrize("nwg", [2, 3])
def test_root_search_is_independent_of_lu_workgroups(emme, nwg):
    """Whole Newton searches with 1 and with several LU workgroups per matrix: same iterates,
    same roots, bit for bit (chains retire at different steps, so the launches see dense lists
    of live matrices of changing length)."""
    d = example_tokamak(npoints=64)
    guesses = np.array([-0.8 + 0.25j, -0.7 + 0.3j, -0.9 + 0.2j, -0.6 + 0.1j, -1.0 + 0.35j])
    with _ctx(emme, d) as ctx:
        for _ in range(3):  # builds the node cache and lets it settle: later fills are identical
            ctx.solve_roots(guesses)
        ctx.set_options(lu_split=1)
        r1, it1, info1, its1 = ctx.solve_roots(guesses, want_iterates=True)
        ctx.set_options(lu_split=nwg)
        r2, it2, info2, its2 = ctx.solve_roots(guesses, want_iterates=True)
    assert np.array_equal(it1, it2) and np.array_equal(info1, info2)
    assert np.array_equal(its1.view(np.float64), its2.view(np.float64), equal_nan=True)
    assert np.array_equal(r1.view(np.float64), r2.view(np.float64), equal_nan=True)
    assert len(set(it1.tolist())) > 1  # the chains did retire at different steps


# ---- every fill kernel, forced through the context's options (emme_options_t) ---------------------
def _modes(emme):
    U, L = emme.FILL_UNION, emme.FILL_LANES
    return {
        # HBM node cache, folded records + phase table; electrostatic GK15: dense fill on the tiled layout, EM /
        # GK31: independent lanes (the defaults); uncached integrals via the work list
        "cached": dict(node_cache_gb=8.0, wl_min=1),
        # electrostatic GK15 cases: every round on the vector path, every round on the matrix cores, one omega
        # per chunk, a tiny cache, and the union-walk kernel on the per-pair layout instead
        "cached-dense-all-sparse": dict(node_cache_gb=8.0, wl_min=1, dense_min_cols=17),
        "cached-dense-all-mfma": dict(node_cache_gb=8.0, wl_min=1, dense_min_cols=1),
        "cached-dense-narrow": dict(node_cache_gb=8.0, wl_min=1, dense_min_tasks=100000000),
        "cached-dense-tiny": dict(node_cache_gb=0.002, wl_min=1, cache_min_depth=1),
        # 128-entry level lists (the build omegas with very wide trees are sent to)
        "cached-dense-wide": dict(node_cache_gb=8.0, wl_min=1, dense_wide=1),
        "cached-union": dict(node_cache_gb=8.0, wl_min=1, fill=U),
        "cached-union-tiny": dict(node_cache_gb=0.002, wl_min=1, fill=U, cache_min_depth=1),
        # the same with the independent-lane kernel for every case
        "cached-independent": dict(node_cache_gb=8.0, wl_min=1, fill=L),
        # unfolded records, exp(A0 + T omega) evaluated per (pair, node, omega) in the fill
        "cached-unfolded": dict(node_cache_gb=8.0, wl_min=1, phase_table=0),
        # electromagnetic cases with one record per moment instead of the shared layout
        "cached-em-per-moment": dict(node_cache_gb=8.0, wl_min=1, em_shared=0),
        # cache too small for anything but the shallowest tree: most integrals are deferred
        "cached-tiny": dict(node_cache_gb=0.002, wl_min=1, cache_min_depth=1),
        "cached-tiny-independent": dict(node_cache_gb=0.002, wl_min=1, fill=L, cache_min_depth=1),
        # no cache: omega-lane kernel (node data shared on the fly inside a lane group)
        "omega-lane": dict(node_cache_gb=0.0, wl_min=1),
        # no cache, lanes-are-nodes kernel only
        "nodes": dict(node_cache_gb=0.0, wl_min=100000),
    }


KERNEL_MODES = ["cached", "cached-dense-all-sparse", "cached-dense-all-mfma", "cached-dense-narrow", "cached-dense-tiny",
                "cached-dense-wide",
                "cached-union", "cached-union-tiny", "cached-independent", "cached-unfolded", "cached-em-per-moment",
                "cached-tiny", "cached-tiny-independent", "omega-lane", "nodes"]


@pytest.mark.parametrize("mode", list(KERNEL_MODES))
def test_every_fill_kernel_matches_oracle(emme, oracle, mode):
    opts = _modes(emme)[mode]
    cases = [
        (example_tokamak(npoints=40), [-0.8 + 0.25j, -0.6 - 0.21j, 0.5 + 0.1j, -0.142 - 1.469j, 0.153 - 0.316j]),
        (example_stellarator(npoints=10), [-1.656 + 2.49j, -0.85 - 0.32j, 0.4 - 0.2j]),
    ]
    for d, ws in cases:
        po = oracle.params(d)
        with _ctx(emme, d, **opts) as ctx:
            M, iv = ctx.assemble(ws, want_intervals=True)
            M1, iv1 = ctx.assemble(ws[1:2], want_intervals=True)
        assert np.abs(M1[0] - M[1]).max() <= 1e-13 * np.abs(M[1]).max() and iv1[0] == iv[1]
        for k, w in enumerate(ws):
            Mo, tot = oracle.assemble(po, complex(w))
            assert iv[k] == tot, (mode, d["conf"], w)
            # (-0.142-1.469j) is a strongly damped point a wandering Newton chain visits:
            # entries are ~1e38 and each is the remainder of integrand values ~1e8 times
            # larger, so BOTH implementations carry ~1e-8 relative rounding there; the tree
            # (interval count) must still be identical
            tol = 1e-6 if w == -0.142 - 1.469j else TOL_M
            assert np.abs(M[k] - Mo).max() <= tol * np.abs(Mo).max(), (mode, d["conf"], w)


def test_union_fill_bits_do_not_depend_on_intervals_per_round(emme):
    """The union-walk kernel serving one or two intervals per round (and two or three items per
    lane group): a lane needs only its own next key, so every omega's matrix and interval count
    are the same bit for bit -- also for omegas whose trees barely overlap."""
    # the union-walk kernel itself (the default for these inputs is the dense fill)
    opts = dict(node_cache_gb=8.0, wl_min=1, fill=emme.FILL_UNION)
    d = example_tokamak(npoints=48)
    rng = np.random.default_rng(3)
    ws = np.concatenate([rng.uniform(-1.2, -0.4, 20) + 1j * rng.uniform(0.05, 0.4, 20),
                         [-0.142 - 1.469j, 0.153 - 0.316j, 4.591 - 3.987j, -0.35 - 0.788j, 0.6 + 0.1j]])
    with _ctx(emme, d, **opts) as ctx:
        # builds the cache and lets it grow to its final shape (a fill that deferred integrals
        # makes the next one cache a subtree around them, at most NODE_CACHE_MAX_SUB times; an
        # integral that moves from the cooperative kernel to the cached path changes its rounding)
        ctx.cache_settle(ws)
        ctx.set_options(union_sel=1)
        M1, iv1 = ctx.assemble(ws, want_intervals=True)
        ctx.set_options(union_sel=2)
        M2, iv2 = ctx.assemble(ws, want_intervals=True)
        ctx.set_options(union_ipg_few=3)
        M3, iv3 = ctx.assemble(ws, want_intervals=True)
        assert ctx.fill_kernel().startswith("k_assemble_union")
    assert np.array_equal(iv1, iv2) and np.array_equal(iv1, iv3)
    assert np.array_equal(M1.view(np.float64), M2.view(np.float64), equal_nan=True)
    assert np.array_equal(M1.view(np.float64), M3.view(np.float64), equal_nan=True)


@pytest.mark.parametrize("mode", ["cached", "cached-independent", "cached-unfolded", "omega-lane", "nodes"])
def test_root_search_same_in_every_kernel_mode(emme, oracle, mode):
    opts = _modes(emme)[mode]
    d = example_tokamak(npoints=32)
    po = oracle.params(d)
    guesses = np.array([-0.8 + 0.25j, -0.7 + 0.3j, -0.9 + 0.2j, -0.5 + 0.1j, 0.6 + 0.2j])
    with _ctx(emme, d, **opts) as ctx:
        roots, iters, info = ctx.solve_roots(guesses)
    for b in (0, 3):
        r_or, its_or, _, _ = oracle.solve_root(po, complex(guesses[b]))
        assert iters[b] == len(its_or) and abs(roots[b] - r_or) <= TOL_W


# ---- the reference's driver (src/main.cpp) through emme_run_json --------------------------
def test_run_json_scan_with_continuation(emme, oracle, tmp_path):
    d = example_tokamak(npoints=24, omega_d_coeff={"head": 1.01, "tail": [0.81, 1.01], "step": 0.1})
    out = emme.run_json(emme.json_text(d), str(tmp_path))
    unit = out["result"]["omega_d_coeff"]
    assert unit["scan_key"] == "omega_d_coeff"
    assert np.allclose(unit["scan_values"], [1.01, 0.91, 0.81])
    guess = -0.8 + 0.25j
    for rec, val in zip(unit["scan_result"], unit["scan_values"]):
        po = oracle.params(example_tokamak(npoints=24, omega_d_coeff=val))
        r_or, its, Mo, _ = oracle.solve_root(po, guess, want_matrix=True)
        got = complex(*rec["eigenvalue"])
        assert abs(got - r_or) <= 2e-6 * abs(r_or)  # text output carries 6 significant digits
        assert rec["scan_value"] == pytest.approx(val)
        # raw matrix file: dim^2 complex128, row-major (src/main.cpp:61-63)
        M = np.fromfile(rec["eigenMatrix"], dtype=np.complex128).reshape(24, 24)
        assert np.abs(M - Mo).max() <= 1e-8 * np.abs(Mo).max()
        # eigenvector = null vector of M(root), up to a phase
        v = np.array([complex(*z) for z in rec["eigenvector"]])
        want = np.linalg.svd(Mo)[2][-1].conj()
        assert abs(abs(np.vdot(want, v)) - 1.0) < 1e-4
        guess = r_or  # omega continuation (src/main.cpp:78)
    assert "run_time" in out and out["input"]["npoints"] == 24


def test_run_json_single_and_error_records(emme, tmp_path):
    d = example_tokamak(npoints=16)
    out = emme.run_json(emme.json_text(d), str(tmp_path))
    rec = out["result"]["(None)"]["scan_result"][0]
    assert len(rec["eigenvector"]) == 16 and os.path.exists(os.path.join(str(tmp_path), "eigenMatrix.bin"))
    # a failing scan point becomes a NaN record with the reason (src/main.cpp:311-318)
    bad = example_tokamak(npoints=16, integration_start_points=21,
                          omega_d_coeff={"head": 1.01, "tail": [0.91, 1.01], "step": 0.1})
    out = emme.run_json(emme.json_text(bad), None)
    recs = out["result"]["omega_d_coeff"]["scan_result"]
    assert len(recs) == 2 and all(r["eigenvalue"] == "NaN" for r in recs)
    assert recs[0]["reason"] == "integration_start_points should be 15 or 31"
    with pytest.raises(emme.EmmeError) as e:
        emme.run_json(emme.json_text(dict(d, method="PIC")), None)
    assert "Method 'PIC' is not supported" in e.value.reason


# ---- QR-secant step (include/solver.h:210-383) ---------------------------------------------
def _near_singular(rng, n, gap):
    """Random complex matrix whose smallest singular value is `gap` times the others."""
    a = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    u, s, vh = np.linalg.svd(a)
    s[-1] = gap * s[0]
    return (u * s) @ vh


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 100, 256, 300, 512, 640])
def test_qr_secant_quotient_matches_lapack_sequence(emme, oracle, n):
    """Householder QR with zlaqp2's pivot rule + ztrtrs + zunmqr on the device vs the same
    LAPACK calls on the host (oracle.qr_secant); all three register-width variants of the
    kernel (n <= 256, <= 512, <= 1024), generic and nearly singular (the Newton regime)."""
    rng = np.random.default_rng(100 + n)
    nb = 4 if n <= 300 else 2
    A = np.stack([_near_singular(rng, n, 1e-7) if b % 2 else
                  rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n)) for b in range(nb)])
    B = rng.normal(size=(nb, n, n)) + 1j * rng.normal(size=(nb, n, n))
    with _ctx(emme, example_tokamak(npoints=16)) as ctx:
        q, info = ctx.qr_secant(A, B)
    assert (info == 0).all()
    for b in range(nb):
        dw, i_or = oracle.qr_secant(A[b], B[b])
        assert i_or == 0
        assert abs(-1.0 / q[b] - dw) <= 1e-9 * abs(dw), (n, b, -1.0 / q[b], dw)


def test_qr_secant_zero_diagonal_info(emme, oracle):
    """Two zero columns: R11 gets an exactly zero diagonal entry -> the ztrtrs failure
    (include/solver.h:309-316) is reported per item; the other items are unaffected."""
    rng = np.random.default_rng(5)
    n = 12
    A = rng.normal(size=(3, n, n)) + 1j * rng.normal(size=(3, n, n))
    B = rng.normal(size=(3, n, n)) + 1j * rng.normal(size=(3, n, n))
    A[1][:, 3] = 0.0
    A[1][:, 7] = 0.0
    with _ctx(emme, example_tokamak(npoints=16)) as ctx:
        q, info = ctx.qr_secant(A, B)
    _, i_or = oracle.qr_secant(A[1], B[1])
    assert info[1] == i_or == n - 1 and np.isnan(q[1])
    for b in (0, 2):
        dw, _ = oracle.qr_secant(A[b], B[b])
        assert info[b] == 0 and abs(-1.0 / q[b] - dw) <= 1e-9 * abs(dw)


@pytest.mark.parametrize("em", [False, True])
def test_newton_step_qr_matches_oracle(emme, oracle, em):
    d = example_stellarator(npoints=12) if em else example_tokamak(npoints=24)
    po = oracle.params(d)
    g = (-1.656 + 2.49j) if em else (-0.8 + 0.25j)
    w0, dw0 = 0.99 * g, 0.01 * g
    m_old, _ = oracle.assemble(po, w0)
    m, _ = oracle.assemble(po, w0 + dw0)
    mp = (m - m_old) / dw0
    dw_or, info_or = oracle.qr_secant(m, mp)
    m_new, _ = oracle.assemble(po, w0 + dw0 + dw_or)
    with _ctx(emme, d) as ctx:
        w, dw, M, Mp, info = ctx.newton_step([w0 + dw0], m[None], mp[None], method=1)
    assert info[0] == info_or == 0
    assert abs(dw[0] - dw_or) <= TOL_W * abs(dw_or)
    assert abs(w[0] - (w0 + dw0 + dw_or)) <= TOL_W
    scale = np.abs(m_new).max()
    assert np.abs(M[0] - m_new).max() <= 1e-7 * scale  # omega differs in the last digits
    assert np.abs(Mp[0] - (m_new - m) / dw_or).max() <= 1e-6 * np.abs(Mp[0]).max()


def test_solve_roots_qr_method(emme, oracle):
    """iteration_method != "TraceSecant" selects the QR step (src/main.cpp:45-49); both
    variants reach the same root."""
    d = example_tokamak(npoints=32, iteration_method="QRSecant")
    po = oracle.params(d)
    guesses = [-0.8 + 0.25j, -0.7 + 0.3j]
    with _ctx(emme, d) as ctx:
        roots, iters, info, its = ctx.solve_roots(guesses, want_iterates=True)
    with _ctx(emme, dict(d, iteration_method="TraceSecant")) as ctx:
        roots_tr, _, _ = ctx.solve_roots(guesses)
    assert (info == 0).all()
    for b, g in enumerate(guesses):
        w_or, its_or = oracle.solve_root_qr(po, g)
        assert iters[b] == len(its_or)
        assert np.abs(its[b][:len(its_or)] - its_or).max() <= 1e-8
        assert abs(roots[b] - w_or) <= TOL_W
        assert abs(roots[b] - roots_tr[b]) <= 1e-5 * abs(roots[b])


def test_run_json_qr_method(emme, oracle, tmp_path):
    d = example_tokamak(npoints=32, iteration_method="QR")
    out = emme.run_json(emme.json_text(d), str(tmp_path))
    ev = out["result"]["(None)"]["scan_result"][0]["eigenvalue"]
    w_or, _ = oracle.solve_root_qr(oracle.params(d), complex(*d["initial_guess"]))
    assert abs(complex(ev[0], ev[1]) - w_or) <= 2e-6 * abs(w_or)  # 6 significant digits


# ---- maximum sizes: properties that need no oracle run (BASELINE configs[3] and [4]) ----------
@pytest.mark.parametrize("case", ["es512", "em256"])
def test_full_size_newton_step_properties(emme, oracle, case):
    """N=512 electrostatic (dim 512, 130 816 pairs) and N=256 electromagnetic stellarator
    (dim 512, GK31, three moments): structure of M, secant M', and BOTH Newton steps recomputed
    on the host from the device's own matrices (numpy solve / the LAPACK QR sequence)."""
    if case == "es512":
        d, g = example_tokamak(npoints=512), -0.8 + 0.25j
    else:
        d, g = example_stellarator(npoints=256), -1.656 + 2.49j
    w0, w1 = 0.99 * g, g
    with _ctx(emme, d) as ctx:
        n = ctx.dim
        assert n == 512
        M01 = ctx.assemble([w0, w1])
        M0, M1 = M01[0], M01[1]
        Mp = (M1 - M0) / (w1 - w0)
        w_tr, dw_tr, Mn, Mpn, info = ctx.newton_step([w1], M1[None], Mp[None], method=0)
        w_qr, dw_qr, _, _, info_qr = ctx.newton_step([w1], M1[None], Mp[None], method=1)
        Mchk = ctx.assemble([w_tr[0]])[0]
    N = n if case == "es512" else n // 2
    if case == "es512":
        assert np.array_equal(M1, M1.T)  # both halves of every pair come from one integral
        assert np.all(M1.diagonal() == M1[0, 0])
    else:
        A, B, C, D = M1[:N, :N], M1[:N, N:], M1[N:, :N], M1[N:, N:]
        assert np.array_equal(A, A.T) and np.array_equal(D, D.T)  # include/solver.h:461-511
        assert np.array_equal(C, -B) and np.array_equal(B, -B.T)  # B_ij = -B_ji, C = -B
    assert info[0] == 0 and info_qr[0] == 0
    want_tr = -1.0 / np.trace(np.linalg.solve(M1, Mp))
    assert abs(dw_tr[0] - want_tr) <= 1e-9 * abs(want_tr)
    want_qr, _ = oracle.qr_secant(M1, Mp)
    # the QR quotient divides by the last component of Q^H M' v: far from a root its conditioning
    # (R11 of a 512 x 512 matrix whose blocks differ by orders of magnitude) costs ~7 digits
    assert abs(dw_qr[0] - want_qr) <= 1e-7 * abs(want_qr)
    # the step's own output matrices: M(new omega) and the new secant
    scale = np.abs(Mchk).max()
    assert np.abs(Mn[0] - Mchk).max() <= 1e-12 * scale
    assert np.abs(Mpn[0] - (Mn[0] - M1) / dw_tr[0]).max() <= 1e-9 * np.abs(Mpn[0]).max()


def test_degenerate_batches_and_bad_arguments(emme, oracle):
    """Empty / single-item / zero-step-limit calls and argument errors at the ABI (no crash, the
    documented codes)."""
    import ctypes as C
    d = example_tokamak(npoints=16)
    po = oracle.params(d)
    with _ctx(emme, d) as ctx:
        lib, h = ctx.lib, ctx.h
        w = np.array([-0.8 + 0.25j])
        out = np.zeros(2)
        it = np.zeros(1, dtype=np.int32)
        inf = np.zeros(1, dtype=np.int32)
        # empty batch, null pointers, negative limit
        assert lib.emme_solve_roots(h, w.ctypes.data, 0, 1e-6, 20, out.ctypes.data, it.ctypes.data, inf.ctypes.data, None) == -1
        assert lib.emme_solve_roots(h, None, 1, 1e-6, 20, out.ctypes.data, it.ctypes.data, inf.ctypes.data, None) == -1
        assert lib.emme_solve_roots(h, w.ctypes.data, 1, 1e-6, -1, out.ctypes.data, it.ctypes.data, inf.ctypes.data, None) == -1
        assert lib.emme_assemble_batch(h, w.ctypes.data, 0, out.ctypes.data, None) == -1
        assert lib.emme_ctx_get_matrix(h, 5, out.ctypes.data) == -1  # no such chain
        # unknown iteration method
        M = np.zeros((1, 16, 16), dtype=np.complex128)
        assert lib.emme_newton_step_batch(h, w.ctypes.data, out.ctypes.data, 1, M.ctypes.data, M.ctypes.data, 7,
                                          inf.ctypes.data) == -1
        # step limit 0: exactly one Newton step (src/main.cpp:43 runs j = 0..limit), a single chain
        roots, iters, info, its = ctx.solve_roots([-0.8 + 0.25j], step_limit=0, want_iterates=True)
        r_or, its_or, _, _ = oracle.solve_root(po, -0.8 + 0.25j)
        assert iters[0] == 1 and abs(roots[0] - its_or[0]) <= TOL_W and abs(its[0, 0] - its_or[0]) <= TOL_W


def test_cache_policy_and_buffer_pool(emme, oracle):
    """Default policy: a call with fewer than 8 omegas does not build the node cache (a single
    root of a parameter scan would pay seconds of allocation for nothing), a larger one does and
    later small calls then use it; the big buffers of a destroyed context are reused by the next."""
    d = example_tokamak(npoints=24)
    po = oracle.params(d)
    ws = np.array([-0.8 + 0.25j - 0.01j * k for k in range(9)])
    Mo, _ = oracle.assemble(po, complex(ws[0]))
    for _ in range(2):  # the second context takes its buffers from the pool
        with _ctx(emme, d, cache_min_batch=8, node_cache_gb=4.0) as ctx:
            M1 = ctx.assemble(ws[:2])
            assert "cache" not in ctx.fill_kernel() and ctx.node_cache_gib() == 0.0
            M9 = ctx.assemble(ws)
            assert "cache" in ctx.fill_kernel() and ctx.node_cache_gib() > 0.0
            M2 = ctx.assemble(ws[:2])
            assert "cache" in ctx.fill_kernel()
            for M in (M1[0], M9[0], M2[0]):
                assert np.abs(M - Mo).max() <= TOL_M * np.abs(Mo).max()
    emme.release_pooled_memory()


def test_electrostatic_gk31_matches_oracle(emme, oracle):
    """integration_start_points = 31 on the electrostatic path (lane groups of 32: the
    independent-lane cached kernel on folded records), including both contour classes."""
    d = example_tokamak(npoints=20, integration_start_points=31)
    po = oracle.params(d)
    ws = np.array([-0.8 + 0.25j, -0.6 - 0.21j, 0.5 + 0.1j, -0.3 - 0.05j, -0.7 + 0.3j, 0.2 - 0.4j,
                   -0.9 + 0.1j, -0.5 + 0.2j, -0.65 + 0.27j])
    with _ctx(emme, d) as ctx:
        M, iv = ctx.assemble(ws, want_intervals=True)
        assert "cache" in ctx.fill_kernel()
        roots, iters, info = ctx.solve_roots(ws[[0, 4]])
    for k, w in enumerate(ws):
        Mo, tot = oracle.assemble(po, complex(w))
        assert iv[k] == tot
        assert np.abs(M[k] - Mo).max() <= TOL_M * np.abs(Mo).max()
    for k, g in enumerate(ws[[0, 4]]):
        r_or = oracle.solve_root(po, complex(g))[0]
        # on this coarse grid one of the chains ends on a damped root (Im w < 0), where M is badly
        # conditioned: agreement to the solver's own stopping tolerance, not to TOL_W
        assert info[k] == 0 and abs(roots[k] - r_or) <= 1e-6 * abs(r_or)
