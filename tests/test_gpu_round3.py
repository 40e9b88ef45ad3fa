"""Round-3 GPU tests (pytest -m gpu), all through the C ABI:
  * the fill at FULL size (N = 256) on the default (dense, matrix-core) path where the headline's wandering
    chains go -- strongly damped omegas, the Re omega > 0 contour class, |omega| up to 6 -- against checksums and
    Gauss-Kronrod interval counts of the reference's own kappa sources (tests/golden/cfg3_damped.npz), with the
    tolerance taken from the reference's own sensitivity;
  * the dense fill's unclamped tails (src/Parameters.cpp:167-173), entry by entry, on pairs where most nodes are
    clamped, against the exact union kernel and the oracle;
  * nullSpace on the device, batched (include/solver.h:58-112);
  * per-context options through the ABI (emme_options_t) instead of environment variables;
  * lost matrices (a non-finite integral) are left alone by the fill and retire their chain.
"""
import ctypes
import os
import sys

import numpy as np
import pytest

from oracle.binding import example_stellarator, example_tokamak

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from integrand_np import PairNodes  # noqa: E402

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_M = 1e-10


def _ctx(emme, d, **options):
    return emme.Context(emme.params_from_dict(d), **options)


# ---- 1. the fill where the wandering chains go, full size, default path ---------------------------------------
def test_damped_omegas_full_size_match_reference_checksums(emme):
    """36 omegas taken verbatim from the reference's own iterates of the 15 non-converging headline chains, of
    chain 30 (Re omega > 0) and chain 80 (abandoned by the reference): ONE batch on a settled DEFAULT context at
    N = 256 (tiled node cache with run-time growth, dense fill, cooperative deferred pass).
    Pinned: the total Gauss-Kronrod interval count of every matrix (= the reference's trees, integral by
    integral summed), and Frobenius norm, sum, row sums and 40 entries within 10x the spread the REFERENCE's own
    matrix shows when omega changes in its last digits (fixture: M(omega (1 + 1e-13)) - M(omega), both from
    oracle/_ref) -- floor TOL_M = 1e-10 of max|M|.  make_golden_cfg3_damped.py."""
    import bench
    z = np.load(os.path.join(G, "cfg3_damped.npz"))
    assert z["done"].all() and z["oracle_bits_equal"].all()
    fin = z["nonfinite"] == 0
    ws = z["omegas"][fin]
    assert len(ws) >= 30 and (ws.imag < -0.5).sum() >= 8 and (ws.real > 0).sum() >= 8
    eij = z["eij"]
    d = bench.workload_dict(256)
    with _ctx(emme, d, cache_min_batch=8) as ctx:
        fills = ctx.cache_settle(ws)
        assert ctx.fill_kernel().startswith("k_assemble_dense") and fills >= 2
        M, iv = ctx.assemble(ws, want_intervals=True)
        M2, iv2 = ctx.assemble(ws, want_intervals=True)
        # the one omega whose matrix holds infinities in the reference too (chain 80's second iterate)
        wbad = z["omegas"][~fin]
        rc = ctx.assemble_rc(wbad)
    assert rc == -6  # EMME_ENUMERIC: non-finite integral, as in the reference's matrix (nonfinite > 0 in the fixture)
    assert np.array_equal(iv, iv2) and np.array_equal(M.view(np.float64), M2.view(np.float64), equal_nan=True)
    worst = 0.0
    for k, kk in enumerate(np.nonzero(fin)[0]):
        assert iv[k] == z["intervals"][kk], (k, ws[k], iv[k], z["intervals"][kk])
        mx = z["maxabs"][kk]
        tol = max(TOL_M * mx, 10.0 * z["spread_max"][kk])
        e_ent = np.abs(M[k][eij[:, 0], eij[:, 1]] - z["entries"][kk]).max()
        e_row = np.abs(M[k].sum(axis=1) - z["rowsum"][kk]).max()
        e_fro = abs(np.sqrt((np.abs(M[k]) ** 2).sum()) - z["fro"][kk])
        e_sum = abs(M[k].sum() - z["sum"][kk])
        assert e_ent <= tol, (ws[k], e_ent, tol)
        assert e_row <= 16 * tol and e_fro <= 16 * tol and e_sum <= 256 * tol, (ws[k], e_row, e_fro, e_sum, tol)
        assert np.abs(M[k] - M[k].T).max() == 0.0  # include/solver.h:453: mat(j,i) = mat(i,j)
        worst = max(worst, e_ent / tol)
    print(f"damped omegas: worst entry error / tolerance = {worst:.3g}")


# ---- 1b. the electromagnetic dense fill, full size ------------------------------------------------------------
def test_em_dense_fill_full_size_matches_reference_checksums(emme):
    """BASELINE configs[3] (stellarator, electromagnetic, GK31, N = 256, dim 512) at nine omegas taken verbatim from
    the reference's own K = 8 chains (cfg4_k8_n256.npz), six of them at Im omega <= 0 where the three moments' trees
    are 5-11 intervals deep per integral: ONE batch on a default context -> the electromagnetic dense fill
    (k_assemble_dense<1, 31, 3>: three moments as columns of a 16 x 16 x 64 GEMM, tiled GK31 records).
    Pinned against oracle/_ref checksums (make_golden_cfg4_damped.py): Frobenius norm, sum, the 512 row sums and 40
    entries from all four blocks within max(1e-12 max|M|, 10 x the reference's own spread under omega (1 + 1e-13));
    the total interval count of every matrix against the C restatement's (last-bit differences from _ref on these
    matrices, `oracle_bits_equal` = 0: the counts are the restatement's); and against the independent-lane kernel
    (k_assemble_cached_em) entry by entry with equal counts."""
    from oracle.binding import example_stellarator
    z = np.load(os.path.join(G, "cfg4_damped.npz"))
    assert z["done"].all() and (z["nonfinite"] == 0).all()
    ws, eij = z["omegas"], z["eij"]
    assert (ws.imag <= 0).sum() >= 5
    d = example_stellarator(npoints=256)
    with _ctx(emme, d) as ctx:
        M, iv = ctx.assemble(ws, want_intervals=True)
        M2, iv2 = ctx.assemble(ws, want_intervals=True)
        assert ctx.fill_kernel_symbol() == "k_assemble_dense<1, 31, 3>"
    with _ctx(emme, d, fill=emme.FILL_LANES) as ctx:
        Ml, ivl = ctx.assemble(ws, want_intervals=True)
        assert ctx.fill_kernel_symbol().startswith("k_assemble_cached_em<31")
    assert np.array_equal(iv, iv2) and np.array_equal(M.view(np.float64), M2.view(np.float64))
    assert np.array_equal(iv, ivl)
    worst = 0.0
    for k, w in enumerate(ws):
        assert iv[k] == z["intervals"][k], (w, iv[k], z["intervals"][k])
        mx = z["maxabs"][k]
        tol = max(1e-12 * mx, 10.0 * z["spread_max"][k])
        e_ent = np.abs(M[k][eij[:, 0], eij[:, 1]] - z["entries"][k]).max()
        e_row = np.abs(M[k].sum(axis=1) - z["rowsum"][k]).max()
        e_fro = abs(np.sqrt((np.abs(M[k]) ** 2).sum()) - z["fro"][k])
        e_sum = abs(M[k].sum() - z["sum"][k])
        assert e_ent <= tol, (w, e_ent, tol)
        assert e_row <= 16 * tol and e_fro <= 16 * tol and e_sum <= 512 * tol, (w, e_row, e_fro, e_sum, tol)
        assert np.abs(M[k] - Ml[k]).max() <= 1e-13 * mx
        assert np.abs(M[k] - M[k].T).max() == 0.0  # include/solver.h:472-509: every block is written with its mirror
        worst = max(worst, e_ent / tol)
    print(f"EM dense fill, 9 omegas at dim 512: worst entry error / tolerance = {worst:.3g}; intervals per integral "
          f"{iv.min() / 97920:.2f} .. {iv.max() / 97920:.2f}")


def test_em_dense_fill_random_omegas_counts_and_conditioning(emme, oracle):
    """Random omegas (unstable, damped, Re omega > 0, near the real axis) on stellarator grids of two sizes through the
    electromagnetic dense fill and through the independent-lane kernel: interval counts equal item by item.  Entries:
    where the matrix is well conditioned the two kernels agree to 1e-12 of max|M|; at damped omegas (Im omega ~ -0.8:
    entries ~1e9 that are remainders of larger integrand values) the ORACLE's own matrix moves by up to 2e-5 of max|M|
    under omega (1 +- 1e-13), and that spread is the bar -- checked against the oracle, counts included, for the most
    damped omegas of the sample: within 10x the spread (observed 1x; the two GPU kernels differ from each other by
    1e-2 of it)."""
    from oracle.binding import example_stellarator
    rng = np.random.default_rng(7)
    for N, nb in ((40, 28), (72, 20)):
        d = example_stellarator(npoints=N)
        ws = rng.uniform(-2.5, 1.0, nb) + 1j * rng.uniform(-1.0, 3.0, nb)
        ws[rng.random(nb) < 0.2] *= 0.1
        with _ctx(emme, d) as ctx:
            Md, ivd = ctx.assemble(ws, want_intervals=True)
            assert ctx.fill_kernel_symbol() == "k_assemble_dense<1, 31, 3>"
        with _ctx(emme, d, fill=emme.FILL_LANES) as ctx:
            Ml, ivl = ctx.assemble(ws, want_intervals=True)
        assert np.array_equal(ivd, ivl)
        mx = np.abs(Ml).max(axis=(1, 2))
        rel = np.abs(Md - Ml).max(axis=(1, 2)) / mx
        well = ws.imag > 0.05
        assert (rel[well] <= 1e-12).all(), (N, ws[well][np.argmax(rel[well])], rel[well].max())
        if N == 40:
            po = oracle.params(d)
            worst = 0.0
            for k in np.argsort(ws.imag)[:5]:
                w = complex(ws[k])
                Mo, tot = oracle.assemble(po, w)
                spread = max(np.abs(oracle.assemble(po, w * (1 + 1e-13))[0] - Mo).max(),
                             np.abs(oracle.assemble(po, w * (1 - 1e-13))[0] - Mo).max())
                assert ivd[k] == tot, (w, ivd[k], tot)
                tol = max(1e-12 * np.abs(Mo).max(), 10.0 * spread)
                err = np.abs(Md[k] - Mo).max()
                assert err <= tol, (w, err, tol, spread)
                worst = max(worst, err / tol)
            print(f"EM dense fill at the 5 most damped of {nb} random omegas (N = 40): worst error / (10 x oracle spread) = {worst:.3g}")


# ---- 2. the dense fill's unclamped tails, entry by entry ----------------------------------------------------
def test_dense_fill_clamped_tails_entry_by_entry(emme, oracle):
    """safe_exp (src/Parameters.cpp:167-173) zeroes a node when Re(A0 + T omega) < -40.  The dense fill cannot
    apply that per (pair, node, omega) inside a GEMM and carries the tails (|term| < e^-40 of its coefficient).
    Constructed case: the 210 pairs of an N = 40 grid with |i - j| >= 20, where >= 80 % of the level-4 nodes are
    clamped (checked with the numpy restatement of the split) -- dense (default) vs the exact union kernel vs the
    oracle, ENTRY BY ENTRY: interval counts equal, and every such entry within
        1e-13 + 1e-10 |entry| + 10 x (how far the ORACLE's own entry moves when omega changes by 1e-13 relative)
    of the oracle's (the last term is the entry's own rounding noise: these far entries are small remainders of
    larger integrand values; at strongly damped omegas it reaches 1e-3 relative, which is why those are pinned
    by checksums with the reference's spread in test_damped_omegas_full_size_match_reference_checksums instead)."""
    d = example_tokamak(npoints=40)
    po = oracle.params(d)
    eta, _ = oracle.grid(d["length"], 40)
    ws = np.array([-0.8 + 0.25j, -0.6 - 0.21j, 0.5 + 0.1j, -0.7 - 0.1j, -0.75 + 0.3j, -0.9 + 0.12j, 0.3 - 0.3j, -0.5 + 0.05j])
    far = [(i, j) for i in range(40) for j in range(i + 1, 40) if j - i >= 20]
    for w in ws[:4]:
        omi = -np.copysign(1.0, w.real)
        fr = [PairNodes(oracle, po, eta[i], eta[j], omi).clamped_fraction(complex(w)) for i, j in far[::23]]
        assert min(fr) >= 0.8, (w, min(fr))
    with _ctx(emme, d, node_cache_gb=8.0) as ctx:
        Md, ivd = ctx.assemble(ws, want_intervals=True)
        assert ctx.fill_kernel().startswith("k_assemble_dense")
    with _ctx(emme, d, node_cache_gb=8.0, fill=emme.FILL_UNION) as ctx:
        Mu, ivu = ctx.assemble(ws, want_intervals=True)
        assert ctx.fill_kernel().startswith("k_assemble_union")
    assert np.array_equal(ivd, ivu)
    I, J = np.array(far).T
    worst = 0.0
    for k, w in enumerate(ws):
        Mo, tot = oracle.assemble(po, complex(w))
        assert ivd[k] == tot
        noise = np.maximum(np.abs(oracle.assemble(po, complex(w) * (1 + 1e-13))[0] - Mo),
                           np.abs(oracle.assemble(po, complex(w) * (1 - 1e-13))[0] - Mo))[I, J]
        eo, ed, eu = Mo[I, J], Md[k][I, J], Mu[k][I, J]
        tol = 1e-13 + 1e-10 * np.abs(eo) + 10.0 * noise
        assert np.all(np.abs(eu - eo) <= tol), (w, (np.abs(eu - eo) / tol).max())
        assert np.all(np.abs(ed - eo) <= tol), (w, (np.abs(ed - eo) / tol).max())
        worst = max(worst, (np.abs(ed - eo) / tol).max())
    print(f"dense vs oracle on {len(far)} far pairs x {len(ws)} omegas: worst error / tolerance {worst:.3g}")


# ---- 3. nullSpace on the device, batched ----------------------------------------------------------------
def _overlap(v, w):
    return abs(np.vdot(v, w)) / (np.linalg.norm(v) * np.linalg.norm(w))


def _svd_null(M):
    return np.linalg.svd(M)[2][-1].conj()  # include/solver.h:106-108: last row of V^H, conjugated


@pytest.mark.parametrize("n", [5, 37, 256, 300, 512, 600, 1024, 1100])
def test_null_vectors_batch_general_matrices(emme, n):
    """emme_null_vectors_batch on NON-symmetric nearly singular matrices (transposed solves, no symmetry
    assumed): n <= 525 the one-workgroup factorisation, up to 1024 the chunked multi-workgroup one, above the
    unblocked one; against the constructed right singular vector."""
    rng = np.random.default_rng(n)
    nb = 3 if n >= 512 else 6
    Ms, want = [], []
    for b in range(nb):
        U = np.linalg.qr(rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n)))[0]
        V = np.linalg.qr(rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n)))[0]
        s = np.sort(rng.uniform(0.3, 3.0, n))[::-1].copy()
        s[-1] = 10.0 ** (-4 - 2 * b)
        Ms.append((U * s) @ V.conj().T)
        want.append(V[:, -1])
    with _ctx(emme, example_tokamak(npoints=16)) as ctx:
        v, info = ctx.null_vectors(np.array(Ms))
    assert (info == 0).all()
    for b in range(nb):
        assert abs(np.linalg.norm(v[b]) - 1.0) <= 1e-12
        assert 1.0 - _overlap(v[b], want[b]) <= 1e-10, (n, b, 1.0 - _overlap(v[b], want[b]))


def test_null_vectors_of_the_headline_roots(emme, oracle):
    """128 eigenvectors at N = 256 in ONE call after the headline root search (M = NULL: the matrices M(omega_root)
    the search left on the device): against numpy's SVD (the reference's zgesdd route, include/solver.h:58-112) of
    those matrices for every converged chain, and of the ORACLE's matrix for a sample; time reported."""
    import bench
    d, g = bench.workload_dict(256), bench.lattice(1, 0, 128)
    po = oracle.params(d)
    with _ctx(emme, d, cache_min_batch=8) as ctx:
        roots, iters, info = ctx.solve_roots(g)
        ctx.profile(True)
        ctx.profile_read(reset=True)
        v, vinfo = ctx.null_vectors(None, nbatch=len(g))
        pr = ctx.profile_read()
        ok = np.nonzero(info == 0)[0]
        mats = {b: ctx.final_matrix(b) for b in ok}
    assert len(ok) >= 100 and (vinfo[ok] == 0).all()
    # The comparison needs a vector the SVD itself determines: sigma_n-1 well above the rounding level of sigma_1
    # (eps * sigma_1).  Five of the lattice's chains end where matrix entries are 1e50 .. 1e77 (the wandering chains
    # of test_damped_omegas...; chain 30's Re omega > 0 "root"): there LAPACK's own last singular vector is noise.
    worst, n_cmp = 0.0, 0
    for b in ok:
        sv = np.linalg.svd(mats[b], compute_uv=False)
        if sv[-2] < 1e-9 * sv[0]:
            continue
        worst = max(worst, 1.0 - _overlap(v[b], _svd_null(mats[b])))
        n_cmp += 1
    assert n_cmp >= 100 and worst <= 1e-8, (n_cmp, worst)
    for b in [b for b in ok if iters[b] <= 6][::40][:3]:
        Mo, _ = oracle.assemble(po, complex(roots[b]))
        assert 1.0 - _overlap(v[b], _svd_null(Mo)) <= 1e-8
    print(f"128 null vectors at n=256: {pr.nullspace_ms:.2f} ms in {pr.nullspace_launches} launch spans; worst 1 - overlap {worst:.1e}")
    assert pr.nullspace_ms <= 10.0


def test_null_vector_em_dim_1024_and_run_json(emme):
    """The largest order of SURVEY 8 (stellarator EM, N = 512: dim 1024) through the chunked factorisation, and the
    driver's eigenvector (emme_run_json -> device nullSpace) against the SVD of the matrix it wrote."""
    d = example_stellarator(npoints=512)
    with _ctx(emme, d) as ctx:
        M = ctx.assemble([-1.656 + 2.49j])[0]
        v, info = ctx.null_vectors(M)
    assert info[0] == 0 and 1.0 - _overlap(v[0], _svd_null(M)) <= 1e-8
    import tempfile
    dd = example_tokamak(npoints=48, initial_guess=[-0.8, 0.25])
    with tempfile.TemporaryDirectory() as tmp:
        out = emme.run_json(emme.json_text(dd), tmp)
        rec = out["result"]["(None)"]["scan_result"][0]
        Mf = np.fromfile(os.path.join(tmp, "eigenMatrix.bin"), dtype=np.complex128).reshape(48, 48)
    vec = np.array([complex(a, b) for a, b in rec["eigenvector"]])
    # (text output carries 6 significant digits, src/JsonParser.cpp:227)
    assert 1.0 - _overlap(vec, _svd_null(Mf)) <= 1e-9


def test_null_vectors_report_an_exactly_singular_factorisation(emme):
    n = 64
    rng = np.random.default_rng(1)
    A = rng.normal(size=(2, n, n)) + 1j * rng.normal(size=(2, n, n))
    A[1, :, 10] = 0.0
    with _ctx(emme, example_tokamak(npoints=16)) as ctx:
        v, info = ctx.null_vectors(A)
        with pytest.raises(emme.EmmeError):
            ctx.null_vectors(None, nbatch=1)  # no root search on this context yet
    assert info[0] == 0 and info[1] == 11 and np.isnan(v[1]).all()


# ---- 4. options through the ABI ---------------------------------------------------------------------------
def test_options_through_the_abi(emme, oracle):
    lib = emme.load()
    p = emme.params_from_dict(example_tokamak(npoints=24))
    h = ctypes.c_void_p()
    o = emme.default_options()
    assert o.size == ctypes.sizeof(emme.Options) and o.node_cache_gb == 176.0 and o.skip_lost == 1
    bad = emme.default_options()
    bad.size = 12
    assert lib.emme_ctx_create_ex(ctypes.byref(p), -1, ctypes.byref(bad), ctypes.byref(h)) == -1
    bad = emme.default_options(union_sel=3)
    assert lib.emme_ctx_create_ex(ctypes.byref(p), -1, ctypes.byref(bad), ctypes.byref(h)) == -1
    ws = np.array([-0.8 + 0.25j - 0.01j * k for k in range(9)])
    with emme.Context(p, node_cache_gb=0.0, wl_min=1) as ctx:  # no cache: omega-lane kernel
        Mn = ctx.assemble(ws)
        assert "cache" not in ctx.fill_kernel() and ctx.node_cache_gib() == 0.0
        ctx.set_options(node_cache_gb=4.0)  # a budget afterwards: the cache is decided again
        Mc = ctx.assemble(ws)
        assert "dense" in ctx.fill_kernel() and ctx.node_cache_gib() > 0.0
        assert ctx.options().node_cache_gb == 4.0
        with pytest.raises(emme.EmmeError):
            ctx.set_options(fill=emme.FILL_UNION)  # the layout is fixed once the cache exists
        ctx.set_options(lu_split=1, dense_min_cols=17)
        Mv = ctx.assemble(ws)
    assert np.abs(Mn - Mc).max() <= TOL_M * np.abs(Mn).max() and np.abs(Mv - Mc).max() <= TOL_M * np.abs(Mn).max()


# ---- 5. lost matrices ----------------------------------------------------------------------------------------
def test_lost_matrix_is_left_alone_and_its_chain_retires(emme):
    """Chain 80 of the headline lattice (guess -1.2+0.3i) jumps to -0.0055-0.734i, where integrals overflow: the
    reference's matrix holds infinities and its zsysv fails at the next step (include/solver.h:142-153).  With
    skip_lost (default) the fill stops working on that matrix at the first non-finite integral -- it cost 8 ms
    of every 65 ms search before -- and the chain retires at the same step with EMME_ENUMERIC; without it, the
    old behaviour (whole matrix filled, the LU reports a zero pivot).  Every other chain: same root and step count either way."""
    import bench
    d = bench.workload_dict(256)
    g = bench.lattice(1, 0, 128)[72:88]  # 16 chains around #80
    res = {}
    for skip in (1, 0):
        with _ctx(emme, d, cache_min_batch=8, skip_lost=skip) as ctx:
            ctx.solve_roots(g)
            ctx.solve_roots(g)  # (cache grown: the third search is the settled one)
            ctx.profile(True)
            ctx.profile_read(reset=True)
            r, it, inf = ctx.solve_roots(g)
            pr = ctx.profile_read()
            res[skip] = (r, it, inf, pr.deferred_ms, pr.gk_intervals)
    (r1, it1, inf1, t1, iv1), (r0, it0, inf0, t0, iv0) = res[1], res[0]
    b = 8  # chain 80
    assert inf1[b] == -6 and inf0[b] > 0 and it1[b] == it0[b] == 3
    keep = np.arange(len(g)) != b
    assert np.array_equal(inf1[keep], inf0[keep]) and np.array_equal(it1[keep], it0[keep])
    # (not bit for bit: an omega's rounding depends on its chunk mates -- vector or MFMA rounds -- and the chunks are
    # cut by cost, which the lost matrix no longer contributes to)
    assert np.abs(r1[keep] - r0[keep]).max() <= 1e-12
    print(f"deferred pass per search: {t1:.2f} ms with skip_lost, {t0:.2f} ms without; intervals {iv1} vs {iv0}")
    assert iv1 < iv0 and t1 < 0.6 * t0


def test_wide_level_lists_for_omegas_far_below_the_axis(emme, oracle):
    """Im omega = -6.7 (where one chain of the 4-GPU weak-scaling lattice ends): 317 intervals per integral, 78 of them
    on one bisection level -- more than the dense fill's 64-entry level lists hold.  In a root search the omega's
    integrals leave the dense fill in its first fill (cooperative kernel) and take the 128-entry build from then on;
    forced here for a plain assembly (dense_wide = 1) and compared with the 64-entry build (which hands them over)
    and the oracle: same interval counts, same entries."""
    d = example_tokamak(npoints=48)
    po = oracle.params(d)
    ws = np.array([-2.6977 - 6.7337j, -1.9019 - 3.7784j, -0.8 + 0.25j])
    with _ctx(emme, d, node_cache_gb=8.0, dense_wide=1) as ctx:
        ctx.cache_settle(ws)
        Mw, ivw = ctx.assemble(ws, want_intervals=True)
    with _ctx(emme, d, node_cache_gb=8.0) as ctx:
        ctx.cache_settle(ws)
        Mn, ivn = ctx.assemble(ws, want_intervals=True)
    assert np.array_equal(ivw, ivn)
    for k, w in enumerate(ws):
        Mo, tot = oracle.assemble(po, complex(w))
        noise = np.abs(oracle.assemble(po, complex(w) * (1 + 1e-13))[0] - Mo).max()
        tol = max(TOL_M * np.abs(Mo).max(), 10.0 * noise)
        assert ivw[k] == tot
        assert np.abs(Mw[k] - Mo).max() <= tol and np.abs(Mn[k] - Mo).max() <= tol, (w, np.abs(Mw[k] - Mo).max(), tol)


# ---- a contour class with only a few omegas does not get a cache of its own ----------------------------------
def test_minority_contour_class_goes_uncached(emme, oracle):
    """A batch whose Re omega > 0 side holds less than a sixteenth of the omegas, on a context without a cache for that
    class: the majority goes through the cached (dense) fill, the minority through the omega-lane kernel in the same
    call -- no cache is built for it (the context holds one class's buffers only), every matrix and interval count
    equals the oracle's; when the minority grows its cache is built as before."""
    d = example_tokamak(npoints=40)
    po = oracle.params(d)
    rng = np.random.default_rng(11)
    ws = rng.uniform(-1.2, -0.4, 32) + 1j * rng.uniform(0.05, 0.4, 32)
    ws[5] = 0.5 + 0.1j
    ws[20] = 0.153 - 0.316j
    with _ctx(emme, d, node_cache_gb=8.0) as ctx:
        M, iv = ctx.assemble(ws, want_intervals=True)
        assert ctx.fill_kernel().startswith("k_assemble_dense")
        one = ctx.node_cache_gib()
        M2, iv2 = ctx.assemble(ws, want_intervals=True)
        assert ctx.node_cache_gib() == one and np.array_equal(iv, iv2)
        # a quarter of the batch on the other side: now that class is cached too
        ws3 = ws.copy()
        ws3[:8] = 0.3 + 0.05j * np.arange(1, 9)
        ctx.assemble(ws3)
        assert ctx.node_cache_gib() > 1.5 * one
    for k in (0, 5, 20, 31):
        Mo, tot = oracle.assemble(po, complex(ws[k]))
        assert iv[k] == tot, (k, ws[k], iv[k], tot)
        assert np.abs(M[k] - Mo).max() <= TOL_M * np.abs(Mo).max(), (k, ws[k])


# ---- the dense fill serves both quadrature orders, electrostatic and electromagnetic ------------------------------
@pytest.mark.parametrize("pts,em", [(15, False), (31, False), (15, True), (31, True)])
def test_dense_fill_every_shape_matches_oracle(emme, oracle, pts, em):
    """k_assemble_dense<1, PTS, NM> for PTS in {15, 31} x NM in {1, 3} (BASELINE's configurations use (15, 1) and
    (31, 3)): the default path of every such context; matrices and interval counts against the oracle, both contour
    classes, a damped omega included."""
    from oracle.binding import example_stellarator
    d = dict(example_stellarator(npoints=14), integration_start_points=pts) if em else example_tokamak(npoints=28, integration_start_points=pts)
    po = oracle.params(d)
    ws = np.array([-0.8 + 0.25j, -0.6 - 0.21j, 0.5 + 0.1j, -1.656 + 2.49j, 0.153 - 0.316j, -0.7 + 0.3j, -0.9 + 0.1j, -0.5 + 0.2j,
                   -0.65 + 0.27j, -1.0 + 0.05j])
    with _ctx(emme, d, node_cache_gb=8.0) as ctx:
        M, iv = ctx.assemble(ws, want_intervals=True)
        assert ctx.fill_kernel_symbol() == "k_assemble_dense<1, %d, %d>" % (pts, 3 if em else 1)
        M2, iv2 = ctx.assemble(ws[3:5], want_intervals=True)
    assert np.array_equal(iv2, iv[3:5])
    for k, w in enumerate(ws):
        Mo, tot = oracle.assemble(po, complex(w))
        assert iv[k] == tot, (pts, em, w, iv[k], tot)
        assert np.abs(M[k] - Mo).max() <= TOL_M * np.abs(Mo).max(), (pts, em, w)
