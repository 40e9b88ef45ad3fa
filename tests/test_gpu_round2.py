"""GPU parity, round 2: the headline shape (BASELINE configs[2]: N=256, the 128-guess lattice,
default kernel routing, fresh context) against chains computed with the reference's own kappa
sources + LAPACK zsysv; the remaining geometry variants; the Bessel helper alone; the failure
records of the scan driver; the RCCL gather through the C ABI.
"""
import json
import os

import numpy as np
import pytest

from oracle.binding import example_stellarator, example_tokamak

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_M = 1e-10
TOL_W = 1e-9


def _ctx(emme, d, **options):
    return emme.Context(emme.params_from_dict(d), **options)


# ---- a8: util::bessel_i_alter_helper on the device, alone (include/functions.h:381-408) -----------
def test_bessel_helper_on_device_matches_reference_vectors(emme):
    f = np.load(os.path.join(G, "bessel.npz"))  # 506 vectors from the reference's own function
    z, want = f["z"], f["out"]
    got = emme.bessel(z)
    # {y0, y1, mu + y0} are UNNORMALISED Miller values: compare each relative to its own size
    for c in range(3):
        rel = np.abs(got[:, c] - want[:, c]) / np.abs(want[:, c])
        assert rel.max() <= 1e-11, (c, rel.max(), z[rel.argmax()])
    assert np.array_equal(got[:, 3], want[:, 3])  # -/+ z: exact
    # what the integrand uses: the normalised ratios y0 / (mu + y0), y1 / (mu + y0)
    for c in range(2):
        r_got, r_want = got[:, c] / got[:, 2], want[:, c] / want[:, 2]
        assert (np.abs(r_got - r_want) <= 1e-12 * np.abs(r_want).max()).all()


# ---- f4: the three remaining `conf` variants through the fill kernels (src/Parameters.cpp:395-440) ---
@pytest.mark.parametrize("name", ["cylinder", "taylor", "cylinder_old"])
def test_assemble_other_geometry_variants(emme, oracle, name):
    d = json.load(open(os.path.join(G, "inputs.json")))["inputs"][name]
    po = oracle.params(d)
    ws = np.array([-0.8 + 0.25j, -0.55 - 0.12j, 0.4 + 0.15j])
    with _ctx(emme, d) as ctx:
        M, iv = ctx.assemble(ws, want_intervals=True)
        roots, iters, info = ctx.solve_roots(ws[:1])
    for k, w in enumerate(ws):
        Mo, tot = oracle.assemble(po, complex(w))
        assert iv[k] == tot, (name, w)
        assert np.abs(M[k] - Mo).max() <= TOL_M * np.abs(Mo).max(), (name, w)
    r_or, its, _, _ = oracle.solve_root(po, complex(ws[0]))
    # (coarse 24-point grids: a chain may end on a damped root where M is badly conditioned, so the
    # root is held to the solver's own stopping tolerance here; TOL_W cases are in test_gpu_parity.py)
    assert info[0] == 0 and iters[0] == len(its) and abs(roots[0] - r_or) <= 1e-6 * abs(r_or)


# ---- the headline shape ---------------------------------------------------------------------------
def _cfg3():
    import bench
    return bench.workload_dict(256), bench.lattice(1, 0, 128)


def test_cfg3_headline_shape_matches_reference_chains(emme):
    """bench.py's workload exactly (N=256, 128-guess lattice, default routing: node cache, union
    kernel on cost-sorted chunks, multi-workgroup LU, deferred pass), on a FRESH context, against
    tests/golden/cfg3_chains.npz = every chain computed in the build container with the reference's
    kappa sources (oracle/_ref) and LAPACK zsysv (make_golden_cfg3.py): iteration counts, every
    iterate and the root of every chain the reference converges, and a second search on the now
    warm context giving the same answers."""
    z = np.load(os.path.join(G, "cfg3_chains.npz"))
    done = z["done"].astype(bool)
    assert done.sum() >= 8
    d, g = _cfg3()
    assert np.array_equal(g, z["guesses"])
    with _ctx(emme, d, cache_min_batch=8) as ctx:  # (the library's own default, not the tests')
        roots, iters, info, its = ctx.solve_roots(g, want_iterates=True)
        assert ctx.fill_kernel().startswith("k_assemble_union") or ctx.fill_kernel().startswith("k_assemble_dense")
        roots2, iters2, info2 = ctx.solve_roots(g)
    if os.environ.get("EMME_TEST_DUMP"):  # development aid: keep what the device produced
        np.savez_compressed(os.environ["EMME_TEST_DUMP"], roots=roots, iters=iters, info=info, iterates=its,
                            roots2=roots2, iters2=iters2, info2=info2)
    conv = done & (z["converged"] == 1)
    assert conv.sum() >= 0.7 * done.sum()
    # Which roots can two correct fp64 implementations be asked to share to 1e-9?  Those the REFERENCE
    # itself reproduces when its input changes in the last digit (second pass of make_golden_cfg3.py:
    # guess * (1 + 1e-13)).  One chain of this lattice (#30) jumps to Re omega > 0, where matrix
    # entries are ~1e57 and the remainder of integrand values 1e16 larger -- pure rounding noise in the
    # reference too (tests/analysis/folded_probe.py) -- and ends on a "root" that moves by ~1e-7.
    sens = np.abs(z["roots_perturbed"] - z["roots"]) / np.abs(z["roots"])
    stable = conv & (z["done_perturbed"] == 1) & (sens <= 1e-10) & (z["iters_perturbed"] == z["iters"])
    assert stable.sum() >= 100
    n_cmp = 0
    for b in np.nonzero(conv)[0]:
        k = int(z["iters"][b])
        assert info[b] == 0, (b, g[b], info[b])
        assert iters[b] == k, (b, g[b], iters[b], k)
        want = z["iterates"][b, :k]
        if not stable[b]:
            # same path as long as the reference's own path is reproducible, same verdict at the end
            assert abs(roots[b] - z["roots"][b]) <= max(1e-9, 1e3 * sens[b]) * abs(z["roots"][b]), (b, g[b], sens[b])
            continue
        # every iterate: chains that visit strongly damped omegas on their way amplify rounding there
        # (M is conditioned to ~1e-8 at those points), so the intermediate iterates are held
        # to 1e-6 relative and the ROOT to 1e-9
        assert np.abs(its[b, :k] - want).max() <= 1e-6 * np.abs(want).max(), (b, g[b])
        assert abs(roots[b] - z["roots"][b]) <= TOL_W * abs(z["roots"][b]), (b, g[b], roots[b], z["roots"][b])
        assert abs(roots2[b] - z["roots"][b]) <= TOL_W * abs(z["roots"][b]) and iters2[b] == k
        n_cmp += 1
    assert n_cmp >= 100
    # the one chain the reference abandons (zsysv reports an exactly singular factor at step 3 from
    # guess -1.2+0.3i: M holds infinities at omega = -0.0055-0.734i) fails here too, at the same step
    for b in np.nonzero(done & (z["info"] != 0))[0]:
        assert info[b] != 0 and iters[b] == z["iters"][b], (b, g[b], info[b], iters[b])
    # Chains the reference does not converge within its 21 steps wander through strongly damped
    # omegas, where a Newton step amplifies last-bit differences of M (conditioned to ~1e-8 there): no
    # late-iterate parity exists for them, in either direction.  What is pinned: the chain follows the
    # reference's up to and including the first iterate that leaves the well-conditioned region
    # (Re omega < -0.2, Im omega > -0.5): the step computed FROM such an omega is the first one that
    # amplifies (near Re omega = 0 the entries grow to 1e50 and beyond, below Im omega = -0.5 to 1e38).
    for b in np.nonzero(done & (z["converged"] == 0) & (z["info"] == 0))[0]:
        want = z["iterates"][b, :int(z["iters"][b])]
        rough = np.nonzero(~((want.real < -0.2) & (want.imag > -0.5)))[0]
        k = int(rough[0]) + 1 if rough.size else len(want)
        k = min(k, int(iters[b]), len(want))
        assert np.abs(its[b, :k] - want[:k]).max() <= TOL_W * np.abs(want[:k]).max(), (b, g[b], k)


def test_run_json_marks_enumeric_failures_as_nan_records(emme):
    """A chain that ends with info < 0 (non-finite integral / quadrature cap: EMME_ENUMERIC) is a
    failed scan point: {"eigenvalue": "NaN", "reason": ...} and no continuation from its omega
    (src/main.cpp:300-318) -- not an eigenvalue record.  omega = 0 makes the secant step 0/0."""
    bad = example_tokamak(npoints=16, initial_guess=[0.0, 0.0],
                          omega_d_coeff={"head": 1.01, "tail": [0.91, 1.01], "step": 0.1})
    out = emme.run_json(emme.json_text(bad), None)
    recs = out["result"]["omega_d_coeff"]["scan_result"]
    assert len(recs) == 2
    for r in recs:
        assert r["eigenvalue"] == "NaN" and "EMME_ENUMERIC" in r["reason"], r


def test_rccl_gather_through_c_abi_single_rank(emme):
    """emme_comm_* / emme_gather_roots with a world of one on the real GPU: ncclCommInitRank,
    ncclAllGather and the item-order unpacking all run (ordering over ranks: tests/test_scan_gloo.py)."""
    uid = emme.comm_unique_id()
    assert len(uid) == 128
    comm = emme.Comm(uid, 0, 1)
    roots = np.array([1 + 2j, -3.5 + 0.25j, 0.125 - 7j])
    ra, ia, fa = comm.gather_roots(roots, [4, 5, 21], [0, 0, -6], 3)
    assert np.array_equal(ra, roots) and list(ia) == [4, 5, 21] and list(fa) == [0, 0, -6]
    with pytest.raises(emme.EmmeError):
        comm.gather_roots(roots, [4, 5, 21], [0, 0, -6], 5)  # not this rank's share: refused before the collective
    comm.close()


def test_cache_settle_is_the_canonical_state(emme, oracle):
    """emme_ctx_cache_settle grows the node cache to its final shape for a set of omegas; from then
    on fills are bit-for-bit repeatable, and a second context settled the same way gives the same
    bits (fresh-vs-warm differences exist only while the cache is still growing)."""
    d = example_tokamak(npoints=48)
    ws = np.array([-0.8 + 0.25j, -0.6 - 0.21j, -0.142 - 1.469j, 0.153 - 0.316j, 4.591 - 3.987j,
                   -0.35 - 0.788j, 0.6 + 0.1j, -0.7 + 0.3j, -0.9 + 0.1j])
    po = oracle.params(d)
    mats = []
    for _ in range(2):
        with _ctx(emme, d, node_cache_gb=8.0) as ctx:
            depth0, sub0, gib0 = ctx.cache_state()
            assert depth0 == -1 and gib0 == 0.0
            fills = ctx.cache_settle(ws)
            depth, sub, gib = ctx.cache_state()
            assert 2 <= fills <= 14 and depth >= 3 and sub >= 1 and gib > 0.0
            ctx.cache_settle(ws)
            assert ctx.cache_state()[1:] == (sub, gib)  # nothing left to grow
            M1, iv1 = ctx.assemble(ws, want_intervals=True)
            M2, iv2 = ctx.assemble(ws, want_intervals=True)
            assert np.array_equal(M1.view(np.float64), M2.view(np.float64), equal_nan=True) and np.array_equal(iv1, iv2)
            mats.append(M1)
    assert np.array_equal(mats[0].view(np.float64), mats[1].view(np.float64), equal_nan=True)
    for k, w in enumerate(ws):
        Mo, tot = oracle.assemble(po, complex(w))
        assert iv1[k] == tot
        if w.imag < -2.0:
            continue  # entries ~1e24 that are remainders of far larger integrand values: only the tree is pinned
        tol = 1e-6 if w.imag < -0.5 else TOL_M  # strongly damped: both sides carry ~1e-8
        assert np.abs(mats[0][k] - Mo).max() <= tol * np.abs(Mo).max(), w


# ---- BASELINE configs[3]: stellarator, electromagnetic, GK31 ----------------------------------------
def test_stellarator_k8_fixed_work_step_by_step(emme):
    """the reference's own chain does not converge at configs[3], so the configuration is
    scored as FIXED WORK -- K = 8 trace-secant Newton steps per guess -- compared step by step with
    chains computed from the reference's kappa sources + LAPACK zsysv (make_golden_stellarator.py),
    for guesses on both sides of Im omega = 0 (13x deeper trees below)."""
    z = np.load(os.path.join(G, "stellarator_k8.npz"))
    for n in (32, 48):
        g, want = z[f"n{n}_guesses"], z[f"n{n}_iterates"]
        with _ctx(emme, example_stellarator(npoints=n)) as ctx:
            roots, iters, info, its = ctx.solve_roots(g, tol=0.0, step_limit=7, want_iterates=True)
        assert (iters == 8).all() and (info == 0).all()
        for b in range(len(g)):
            err = np.abs(its[b, :8] - want[b]) / np.abs(want[b])
            # loose quadrature goal (1e-2) makes M only piecewise smooth; steps are compared to 1e-8
            assert err.max() <= 1e-8, (n, g[b], err)


def test_cfg4_full_size_k8_chains_match_reference_chains(emme):
    """BASELINE configs[3] as bench.py --config 4 runs it, at FULL size: N = 256 (dim 512), electromagnetic,
    GK31, K = 8 fixed Newton steps from two corners and the centre of the 32 x 32 guess lattice, every iterate
    against chains from the reference's kappa sources + LAPACK zsysv (make_golden_cfg4.py)."""
    z = np.load(os.path.join(G, "cfg4_k8_n256.npz"))
    g, want = z["guesses"], z["iterates"]
    with _ctx(emme, example_stellarator(npoints=256)) as ctx:
        roots, iters, info, its = ctx.solve_roots(g, tol=0.0, step_limit=7, want_iterates=True)
    assert (iters == 8).all() and (info == 0).all()
    for b in range(len(g)):
        err = np.abs(its[b, :8] - want[b]) / np.abs(want[b])
        assert err.max() <= 1e-8, (g[b], err)


def test_stellarator_full_size_assembly_against_reference_checksums(emme):
    """N=256 (dim 512) electromagnetic matrices at the shipped guess and at a damped omega against
    checksums of the reference's own kappa sources: every block, sampled entries, row sums."""
    chk = json.load(open(os.path.join(G, "matrix_checksums_stellarator.json")))
    d = example_stellarator(npoints=256)
    ws = np.array([complex(*chk[t]["omega"]) for t in ("guess", "damped")])
    with _ctx(emme, d) as ctx:
        M = ctx.assemble(ws)
    for k, t in enumerate(("guess", "damped")):
        c = chk[t]
        scale = c["max_abs"]
        assert abs(np.linalg.norm(M[k]) - c["fro"]) <= 1e-10 * c["fro"]
        blocks = {"A": M[k][:256, :256], "B": M[k][:256, 256:], "C": M[k][256:, :256], "D": M[k][256:, 256:]}
        for name, blk in blocks.items():
            assert abs(blk.sum() - complex(*c["block_sums"][name])) <= 1e-9 * scale * 256, (t, name)
        for i, j, re, im in c["entries"]:
            assert abs(M[k][int(i), int(j)] - complex(re, im)) <= TOL_M * scale, (t, i, j)
        rows = np.abs(M[k]).sum(axis=1)
        assert np.allclose(rows[:8], c["row_abs_sums_first8"], rtol=1e-10, atol=0)
        assert np.allclose(rows[-8:], c["row_abs_sums_last8"], rtol=1e-10, atol=0)


# ---- BASELINE configs[4]: the 512-point grid ----------------------------------------------------------
def test_n512_full_size_assembly_against_reference_checksums(emme):
    """N=512 electrostatic matrices at two k_rho values of the sweep (omegas on both sides of
    Im omega = 0) against checksums of the reference's own kappa sources."""
    import bench
    chk = json.load(open(os.path.join(G, "matrix_checksums_n512.json")))
    for tag, c in chk.items():
        w = complex(*c["omega"])
        with _ctx(emme, bench.workload_dict(512, k_rho=c["k_rho"], omega_d_coeff=1.01)) as ctx:
            M = ctx.assemble([w])[0]
        scale = c["max_abs"]
        assert abs(np.linalg.norm(M) - c["fro"]) <= 1e-10 * c["fro"], tag
        assert abs(M.sum() - complex(*c["sum"])) <= 1e-9 * scale * 512, tag
        for i, j, re, im in c["entries"]:
            assert abs(M[int(i), int(j)] - complex(re, im)) <= TOL_M * scale, (tag, i, j)
        assert np.allclose(np.abs(M).sum(axis=1)[:8], c["row_abs_sums_first8"], rtol=1e-10, atol=0)


@pytest.mark.gpu
def test_cfg5_n512_chains_match_reference_chains(emme):
    """BASELINE configs[4] (N = 512, the (k_rho, omega) sweep of bench.py --config 5): four chains -- first
    and last k_rho of the sweep, first and last guess of its lattice -- against the golden chains from the
    reference's own kappa sources + zsysv (tests/golden/make_golden_cfg5.py): iteration counts, every
    iterate, the roots."""
    z = np.load(os.path.join(G, "cfg5_chains.npz"))
    n_conv = 0
    for kr in np.unique(z["k_rho"]):
        sel = np.nonzero(z["k_rho"] == kr)[0]
        d = example_tokamak(npoints=512, omega_d_coeff=1.01, k_rho=float(kr))
        with _ctx(emme, d) as ctx:
            roots, iters, info, its = ctx.solve_roots(z["guesses"][sel], want_iterates=True)
        for q, b in enumerate(sel):
            k = int(z["iters"][b])
            want = z["iterates"][b, :k]
            if z["converged"][b]:
                n_conv += 1
                assert info[q] == 0 and iters[q] == k, (kr, z["guesses"][b], info[q], iters[q], k)
                got = its[q, :k]
                assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max(), (kr, np.abs(got - want).max())
                assert abs(roots[q] - z["roots"][b]) <= 1e-9 * abs(z["roots"][b]), (kr, roots[q], z["roots"][b])
            else:  # the reference does not converge either: the first iterates must still agree
                m = min(3, k)
                assert np.abs(its[q, :m] - want[:m]).max() <= 1e-6 * np.abs(want[:m]).max()
    assert n_conv >= 2
