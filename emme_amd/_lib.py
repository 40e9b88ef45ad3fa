"""ctypes binding of include/emme_hip.h.  Names mirror the reference's objects:
Params <- Parameters (include/Parameters.h), Context <- EigenSolver (include/solver.h:44-516).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class EmmeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[emme {code}] {msg}")
        self.code = code
        self.reason = msg


def lib_path() -> str:
    # EMME_LIB lets a developer A/B a differently-built library; the default is the in-tree one
    return os.environ.get("EMME_LIB") or os.path.join(HERE, "libemme_hip.so")


class Params(C.Structure):
    """emme_params_t (include/emme_params.h)."""

    _fields_ = [
        ("conf", C.c_int), ("iteration_method", C.c_int),
        ("q", C.c_double), ("shat", C.c_double), ("tau", C.c_double),
        ("epsilon_n", C.c_double), ("epsilon_r", C.c_double),
        ("eta_i", C.c_double), ("eta_e", C.c_double), ("k_rho", C.c_double),
        ("beta_e", C.c_double), ("R", C.c_double), ("vt", C.c_double),
        ("omega_d_coeff", C.c_double), ("length", C.c_double), ("theta", C.c_double),
        ("npoints", C.c_int), ("iteration_step_limit", C.c_int),
        ("integration_precision", C.c_double), ("integration_accuracy", C.c_double),
        ("integration_iteration_limit", C.c_int), ("integration_start_points", C.c_int),
        ("arc_coeff", C.c_double),
        ("water_bag_weight_vpara", C.c_double), ("water_bag_weight_vperp", C.c_double),
        ("drift_center_transformation_switch", C.c_int),
        ("iteration_precision", C.c_double),
        ("initial_guess", C.c_double * 2),
        ("eta_k", C.c_double), ("lh", C.c_int), ("mh", C.c_int),
        ("epsilon_h_t", C.c_double), ("alpha_0", C.c_double), ("r_over_R", C.c_double),
        ("b_theta", C.c_double), ("alpha", C.c_double), ("omega_s_i", C.c_double),
        ("omega_s_e", C.c_double), ("omega_d_bar", C.c_double),
        ("deltap", C.c_double), ("beta_e_p", C.c_double), ("rdeltapp", C.c_double),
        ("curvature_aver", C.c_double), ("shat_coeff", C.c_double),
    ]

    @property
    def dim(self) -> int:
        return self.npoints if self.beta_e == 0.0 else 2 * self.npoints


class Profile(C.Structure):
    """emme_profile_t."""

    _fields_ = [
        ("assemble_ms", C.c_double), ("assemble_launches", C.c_long),
        ("deferred_ms", C.c_double), ("deferred_launches", C.c_long),
        ("linstep_ms", C.c_double), ("linstep_launches", C.c_long),
        ("other_ms", C.c_double), ("other_launches", C.c_long),
        ("gk_intervals", C.c_longlong), ("integrand_evals", C.c_longlong),
        ("matrices", C.c_longlong), ("union_rounds", C.c_longlong),
        ("cache_build_ms", C.c_double), ("cache_build_launches", C.c_long),
        ("cache_alloc_ms", C.c_double),
        ("dense_rounds", C.c_longlong), ("sparse_rounds", C.c_longlong),
        ("sparse_columns", C.c_longlong), ("tile_tasks", C.c_longlong),
        ("nullspace_ms", C.c_double), ("nullspace_launches", C.c_long),
    ]


FILL_AUTO, FILL_UNION, FILL_LANES = 0, 1, 2


class Options(C.Structure):
    """emme_options_t: per-context options (cache budget, fill routing, LU split ...)."""

    _fields_ = [
        ("size", C.c_int),
        ("node_cache_gb", C.c_double), ("cache_min_batch", C.c_int), ("cache_min_depth", C.c_int),
        ("fill", C.c_int), ("phase_table", C.c_int), ("em_shared", C.c_int), ("wl_min", C.c_int),
        ("union_sel", C.c_int), ("union_ipg_few", C.c_int), ("union_few_chunks", C.c_int),
        ("coop_wide_min", C.c_int), ("defer_one_group", C.c_int),
        ("dense_min_cols", C.c_int), ("dense_min_tasks", C.c_int), ("dense_cost_ratio", C.c_double),
        ("dense_wide", C.c_int),
        ("skip_lost", C.c_int),
        ("lu_split", C.c_int), ("lu_group_min_n", C.c_int), ("lu_spin_limit", C.c_int),
        ("lu_unblocked", C.c_int),
    ]


# options every new Context starts from (the tests build their node cache for a handful of omegas)
_DEFAULT_OPTIONS: dict = {}


def set_default_options(**kw) -> None:
    """Python-side defaults merged into every Context created afterwards (explicit arguments win)."""
    _DEFAULT_OPTIONS.clear()
    _DEFAULT_OPTIONS.update(kw)


def default_options(**kw) -> Options:
    o = Options()
    load().emme_options_default(C.byref(o))
    for k, v in {**_DEFAULT_OPTIONS, **kw}.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown option {k!r}")
        setattr(o, k, v)
    return o


def load():
    """Load libemme_hip.so or raise: the product path never falls back to a CPU path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise EmmeError(-3, f"{path} is missing: build it with __graft_entry__.build() "
                            "(make -C emme_amd/csrc); there is no CPU fallback")
    lib = C.CDLL(path)
    P = C.c_void_p
    PP = C.POINTER(Params)
    lib.emme_last_error.restype = C.c_char_p
    lib.emme_params_from_json.argtypes = [C.c_char_p, PP]
    lib.emme_params_derive.argtypes = [PP]
    lib.emme_tables.argtypes = [PP, P, P, P, P]
    lib.emme_weight.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.emme_weight.restype = C.c_double
    lib.emme_bessel_batch.argtypes = [P, C.c_int, P]
    lib.emme_ctx_create.argtypes = [PP, C.c_int, C.POINTER(P)]
    lib.emme_options_default.argtypes = [C.POINTER(Options)]
    lib.emme_options_default.restype = None
    lib.emme_ctx_create_ex.argtypes = [PP, C.c_int, C.POINTER(Options), C.POINTER(P)]
    lib.emme_ctx_set_options.argtypes = [P, C.POINTER(Options)]
    lib.emme_ctx_get_options.argtypes = [P, C.POINTER(Options)]
    lib.emme_null_vectors_batch.argtypes = [P, C.c_int, C.c_int, P, P, P]
    lib.emme_ctx_destroy.argtypes = [P]
    lib.emme_ctx_destroy.restype = None
    lib.emme_release_pooled_memory.argtypes = []
    lib.emme_release_pooled_memory.restype = None
    lib.emme_ctx_set_stream.argtypes = [P, P]
    lib.emme_ctx_dim.argtypes = [P]
    lib.emme_ctx_fill_mode.argtypes = [P]
    lib.emme_ctx_node_cache_gib.argtypes = [P]
    lib.emme_ctx_node_cache_gib.restype = C.c_double
    lib.emme_ctx_cache_settle.argtypes = [P, P, C.c_int, P]
    lib.emme_ctx_cache_state.argtypes = [P, P, P, P]
    lib.emme_ctx_profile_enable.argtypes = [P, C.c_int]
    lib.emme_ctx_profile_read.argtypes = [P, C.POINTER(Profile), C.c_int]
    lib.emme_assemble_batch.argtypes = [P, P, C.c_int, P, P]
    lib.emme_trace_solve_batch.argtypes = [P, C.c_int, C.c_int, P, P, P, P]
    lib.emme_qr_secant_batch.argtypes = [P, C.c_int, C.c_int, P, P, P, P]
    lib.emme_newton_step_batch.argtypes = [P, P, P, C.c_int, P, P, C.c_int, P]
    lib.emme_solve_roots.argtypes = [P, P, C.c_int, C.c_double, C.c_int, P, P, P, P]
    lib.emme_ctx_get_matrix.argtypes = [P, C.c_int, P]
    lib.emme_null_vector.argtypes = [P, C.c_int, P]
    lib.emme_run_json.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    lib.emme_free.argtypes = [C.c_void_p]
    lib.emme_free.restype = None
    lib.emme_scan_values.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, P, P, C.c_int]
    lib.emme_comm_unique_id.argtypes = [P]
    lib.emme_comm_create.argtypes = [P, C.c_int, C.c_int, C.c_int, C.POINTER(P)]
    lib.emme_comm_destroy.argtypes = [P]
    lib.emme_comm_destroy.restype = None
    lib.emme_gather_roots.argtypes = [P, P, P, P, P, C.c_int, C.c_int, P, P, P]
    lib.emme_comm_available.argtypes = []
    lib.emme_gather_slots.argtypes = [C.c_int, C.c_int]
    lib.emme_gather_share.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.emme_gather_pack.argtypes = [C.c_int, C.c_int, P, P, P, C.c_int, C.c_int, P]
    lib.emme_gather_unpack.argtypes = [C.c_int, C.c_int, P, P, P, P]
    _LIB = lib
    return lib


def _check(rc):
    if rc != 0:
        raise EmmeError(rc, load().emme_last_error().decode(errors="replace"))


def _fnum(x) -> str:
    if isinstance(x, bool):
        return "true" if x else "false"
    if isinstance(x, int):
        return str(x)
    s = repr(float(x))
    if "e" in s:
        m, e = s.split("e")
        if "." not in m:
            m += ".0"
        return m + "e" + e
    return s


def json_text(d: dict) -> str:
    """Serialise a dict so that the reference grammar (a number is a float only if it
    contains '.') reads back the intended values."""
    items = []
    for k, v in d.items():
        if isinstance(v, str):
            items.append(f'"{k}": "{v}"')
        elif isinstance(v, (list, tuple)):
            items.append(f'"{k}": [' + ", ".join(_fnum(x) for x in v) + "]")
        elif isinstance(v, dict):
            items.append(f'"{k}": ' + json_text(v))
        else:
            items.append(f'"{k}": {_fnum(v)}')
    return "{" + ", ".join(items) + "}"


def params_from_json(text: str) -> Params:
    """Parameters::generate on a JSON text (reference src/Parameters.cpp:10-66)."""
    p = Params()
    _check(load().emme_params_from_json(text.encode(), C.byref(p)))
    return p


def params_from_dict(d: dict) -> Params:
    return params_from_json(json_text(d))


def tables(p: Params):
    n = p.npoints
    eta, g, b = np.zeros(n), np.zeros(n), np.zeros(n)
    dx = C.c_double(0)
    _check(load().emme_tables(C.byref(p), eta.ctypes.data, g.ctypes.data, b.ctypes.data,
                              C.addressof(dx)))
    return eta, g, b, dx.value


def weight(n, i, j) -> float:
    return load().emme_weight(n, i, j)


def bessel(z) -> np.ndarray:
    """util::bessel_i_alter_helper (include/functions.h:381-408) on the device: [n, 4] complex."""
    z = np.ascontiguousarray(np.atleast_1d(z), dtype=np.complex128)
    out = np.zeros((len(z), 4), dtype=np.complex128)
    _check(load().emme_bessel_batch(z.ctypes.data, len(z), out.ctypes.data))
    return out


def null_vector(M) -> np.ndarray:
    """nullSpace (reference include/solver.h:58-112) of a complex symmetric matrix."""
    M = np.ascontiguousarray(M, dtype=np.complex128)
    v = np.zeros(M.shape[0], dtype=np.complex128)
    _check(load().emme_null_vector(M.ctypes.data, M.shape[0], v.ctypes.data))
    return v


def scan_values(head, step, tail):
    """The values (and turning flags) one {head, step, tail} axis visits (src/main.cpp:139-172)."""
    if isinstance(tail, (list, tuple)):
        t0, t1 = tail
    else:
        t0, t1 = tail, head + 0.5 * np.copysign(step, head - tail)
    vals = np.zeros(100000)
    turn = np.zeros(100000, dtype=np.int32)
    n = load().emme_scan_values(head, step, t0, t1, vals.ctypes.data, turn.ctypes.data, len(vals))
    return vals[:n].copy(), turn[:n].copy()


def run_json(text: str, matrix_dir: str | None = None) -> dict:
    """The reference's main() on an input.json text (src/main.cpp:182-338); returns output.json."""
    import json
    out = C.c_void_p()
    rc = load().emme_run_json(text.encode(), matrix_dir.encode() if matrix_dir else None, C.byref(out))
    _check(rc)
    try:
        return json.loads(C.string_at(out).decode())
    finally:
        load().emme_free(out)


def release_pooled_memory() -> None:
    """Give the node-cache buffers kept from destroyed contexts back to the driver."""
    load().emme_release_pooled_memory()


COMM_ID_BYTES = 128


def comm_available() -> bool:
    """Can RCCL be bound in this process?  Not a collective (see emme_comm_available)."""
    return load().emme_comm_available() == 0


def gather_pack(rank: int, world: int, roots, iters, info, n_total: int) -> np.ndarray:
    """emme_gather_pack: this rank's 4 m doubles of the all-gather (host only)."""
    lib = load()
    r = np.ascontiguousarray(roots, dtype=np.complex128)
    it = np.ascontiguousarray(iters, dtype=np.int32)
    inf = np.ascontiguousarray(info, dtype=np.int32)
    m = lib.emme_gather_slots(n_total, world)
    if m < 0:
        _check(m)
    send = np.zeros(4 * m)
    _check(lib.emme_gather_pack(rank, world, r.ctypes.data, it.ctypes.data, inf.ctypes.data, len(r), n_total,
                                send.ctypes.data))
    return send


def gather_unpack(world: int, n_total: int, allbuf):
    """emme_gather_unpack: the all-gather's world * 4 m doubles in item order (host only)."""
    a = np.ascontiguousarray(allbuf, dtype=np.float64)
    ra = np.zeros(n_total, dtype=np.complex128)
    ia = np.zeros(n_total, dtype=np.int32)
    fa = np.zeros(n_total, dtype=np.int32)
    _check(load().emme_gather_unpack(world, n_total, a.ctypes.data, ra.ctypes.data, ia.ctypes.data, fa.ctypes.data))
    return ra, ia, fa


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the C ABI (rank 0 calls it and hands the bytes to the others)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load().emme_comm_unique_id(buf))
    return buf.raw


class Comm:
    """RCCL communicator of the scan's one collective (emme_comm_* in include/emme_hip.h)."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int = -1):
        assert len(unique_id) == COMM_ID_BYTES
        self.lib = load()
        self.rank, self.world = rank, world
        h = C.c_void_p()
        _check(self.lib.emme_comm_create(unique_id, rank, world, device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.emme_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gather_roots(self, roots, iters, info, n_total: int, stream_handle: int = 0):
        """ONE ncclAllGather of {w_re, w_im, iters, info}; returns all n_total results in item
        order (item k was solved by rank k mod world) on every rank."""
        r = np.ascontiguousarray(roots, dtype=np.complex128)
        it = np.ascontiguousarray(iters, dtype=np.int32)
        inf = np.ascontiguousarray(info, dtype=np.int32)
        ra = np.zeros(n_total, dtype=np.complex128)
        ia = np.zeros(n_total, dtype=np.int32)
        fa = np.zeros(n_total, dtype=np.int32)
        _check(self.lib.emme_gather_roots(self.h, C.c_void_p(stream_handle), r.ctypes.data, it.ctypes.data,
                                          inf.ctypes.data, len(r), n_total, ra.ctypes.data, ia.ctypes.data,
                                          fa.ctypes.data))
        return ra, ia, fa


def _c128(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.complex128)
    return a if shape is None else a.reshape(shape)


class Context:
    """One (device, parameter set): owns device tables and batch scratch."""

    def __init__(self, params: Params, device: int = -1, **options):
        """options: fields of emme_options_t (node_cache_gb, cache_min_batch, fill, lu_split, ...)."""
        self.lib = load()
        self.params = params
        h = C.c_void_p()
        o = default_options(**options)
        _check(self.lib.emme_ctx_create_ex(C.byref(params), device, C.byref(o), C.byref(h)))
        self.h = h
        self.dim = self.lib.emme_ctx_dim(h)

    def options(self) -> Options:
        o = Options()
        _check(self.lib.emme_ctx_get_options(self.h, C.byref(o)))
        return o

    def set_options(self, **kw) -> None:
        o = self.options()
        for k, v in kw.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown option {k!r}")
            setattr(o, k, v)
        _check(self.lib.emme_ctx_set_options(self.h, C.byref(o)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.emme_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream_handle: int):
        _check(self.lib.emme_ctx_set_stream(self.h, C.c_void_p(stream_handle)))

    FILL_KERNELS = {0: "k_assemble (lanes=nodes)", 1: "k_assemble_wl (omega-lane)",
                    2: "k_assemble_cached (HBM node cache)",
                    3: "k_assemble_union (HBM node cache + phase table)",
                    4: "k_assemble_dense (tiled HBM node cache + weighted phase tables, FP64 matrix cores)"}

    def fill_kernel(self) -> str:
        return self.FILL_KERNELS.get(self.lib.emme_ctx_fill_mode(self.h), "none yet")

    def fill_kernel_symbol(self) -> str:
        """Name of the last fill's main kernel as rocprofv3 prints it (key of profiles/*_pmc_summary.json)."""
        mode = self.lib.emme_ctx_fill_mode(self.h)
        pts = self.params.integration_start_points
        em = self.params.beta_e != 0.0
        o = self.options()
        folded = "true" if o.phase_table else "false"
        if mode == 4:
            return "k_assemble_dense<1, %d, %d>" % (pts, 3 if em else 1)
        if mode == 3:
            return "k_assemble_union<15, %d>" % o.union_sel
        if mode == 2:
            return f"k_assemble_cached_em<{pts}, {folded}>" if em else f"k_assemble_cached<{pts}, {folded}>"
        if mode == 1:
            return f"k_assemble_wl<{pts}>"
        return f"k_assemble<{pts}, false>"

    def node_cache_gib(self) -> float:
        return self.lib.emme_ctx_node_cache_gib(self.h)

    def cache_settle(self, omegas) -> int:
        """Grow the node cache to its final shape for these omegas (the canonical state); returns
        the number of fills it took."""
        w = _c128(np.atleast_1d(omegas))
        n = C.c_int(0)
        _check(self.lib.emme_ctx_cache_settle(self.h, w.ctypes.data, w.shape[0], C.byref(n)))
        return n.value

    def cache_state(self):
        """(depth of the fully cached tree, cached subtrees, GiB held)."""
        d, k, g = C.c_int(0), C.c_int(0), C.c_double(0)
        _check(self.lib.emme_ctx_cache_state(self.h, C.byref(d), C.byref(k), C.byref(g)))
        return d.value, k.value, g.value

    def profile(self, on=True):
        _check(self.lib.emme_ctx_profile_enable(self.h, int(on)))

    def profile_read(self, reset=False) -> Profile:
        pr = Profile()
        _check(self.lib.emme_ctx_profile_read(self.h, C.byref(pr), int(reset)))
        return pr

    # matrixAssembler (include/solver.h:417-515), batched over omega
    def assemble(self, omegas, out_device_ptr: int | None = None, want_intervals=False):
        w = _c128(np.atleast_1d(omegas))
        nb = w.shape[0]
        iv = np.zeros(nb, dtype=np.int64)
        if out_device_ptr is None:
            M = np.zeros((nb, self.dim, self.dim), dtype=np.complex128)
            _check(self.lib.emme_assemble_batch(self.h, w.ctypes.data, nb, M.ctypes.data,
                                                iv.ctypes.data))
            return (M, iv) if want_intervals else M
        _check(self.lib.emme_assemble_batch(self.h, w.ctypes.data, nb,
                                            C.c_void_p(out_device_ptr), iv.ctypes.data))
        return iv

    def assemble_rc(self, omegas) -> int:
        """emme_assemble_batch's return code alone (0, or EMME_ENUMERIC when a matrix holds a non-finite integral)."""
        w = _c128(np.atleast_1d(omegas))
        M = np.zeros((w.shape[0], self.dim, self.dim), dtype=np.complex128)
        return self.lib.emme_assemble_batch(self.h, w.ctypes.data, w.shape[0], M.ctypes.data, None)

    def trace_solve(self, A, B):
        A = _c128(A).copy()
        B = _c128(B).copy()
        if A.ndim == 2:
            A, B = A[None], B[None]
        nb, n, _ = A.shape
        tr = np.zeros(nb, dtype=np.complex128)
        info = np.zeros(nb, dtype=np.int32)
        _check(self.lib.emme_trace_solve_batch(self.h, n, nb, A.ctypes.data, B.ctypes.data,
                                               tr.ctypes.data, info.ctypes.data))
        return tr, info

    # the linear algebra of newtonQRSecantIteration (include/solver.h:244-370): q with
    # domega = -1/q, for arbitrary square A (= M) and B (= M')
    def qr_secant(self, A, B):
        A = _c128(A)
        B = _c128(B)
        if A.ndim == 2:
            A, B = A[None], B[None]
        A, B = np.ascontiguousarray(A), np.ascontiguousarray(B)
        nb, n, _ = A.shape
        q = np.zeros(nb, dtype=np.complex128)
        info = np.zeros(nb, dtype=np.int32)
        _check(self.lib.emme_qr_secant_batch(self.h, n, nb, A.ctypes.data, B.ctypes.data,
                                             q.ctypes.data, info.ctypes.data))
        return q, info

    # newtonTraceSecantIteration / newtonQRSecantIteration (include/solver.h:113-160, 210-383),
    # batched; method: 0 trace-secant, 1 QR-secant
    def newton_step(self, omegas, M, Mp, method=0):
        w = _c128(np.atleast_1d(omegas)).copy()
        nb = w.shape[0]
        M = _c128(M, (nb, self.dim, self.dim)).copy()
        Mp = _c128(Mp, (nb, self.dim, self.dim)).copy()
        dw = np.zeros(nb, dtype=np.complex128)
        info = np.zeros(nb, dtype=np.int32)
        _check(self.lib.emme_newton_step_batch(self.h, w.ctypes.data, dw.ctypes.data, nb,
                                               M.ctypes.data, Mp.ctypes.data, method,
                                               info.ctypes.data))
        return w, dw, M, Mp, info

    # solve_once_eigen (src/main.cpp:19-80), batched over initial guesses
    def solve_roots(self, guesses, tol=None, step_limit=None, want_iterates=False):
        g = _c128(np.atleast_1d(guesses))
        n = g.shape[0]
        tol = self.params.iteration_precision if tol is None else tol
        step_limit = self.params.iteration_step_limit if step_limit is None else step_limit
        roots = np.zeros(n, dtype=np.complex128)
        iters = np.zeros(n, dtype=np.int32)
        info = np.zeros(n, dtype=np.int32)
        its = np.zeros((n, step_limit + 1), dtype=np.complex128) if want_iterates else None
        _check(self.lib.emme_solve_roots(self.h, g.ctypes.data, n, tol, step_limit,
                                         roots.ctypes.data, iters.ctypes.data, info.ctypes.data,
                                         its.ctypes.data if want_iterates else None))
        return (roots, iters, info, its) if want_iterates else (roots, iters, info)

    def null_vectors(self, M=None, nbatch=None):
        """nullSpace (include/solver.h:58-112) on the device, batched: right singular vector of the smallest
        singular value of every matrix.  M = None: the matrices M(omega_final) of the last solve_roots call.
        Returns (vectors [nbatch, n], info [nbatch])."""
        if M is None:
            n, nb, ptr = self.dim, nbatch, None
        else:
            M = _c128(M)
            if M.ndim == 2:
                M = M[None]
            M = np.ascontiguousarray(M)
            nb, n, ptr = M.shape[0], M.shape[1], M.ctypes.data
        v = np.zeros((nb, n), dtype=np.complex128)
        info = np.zeros(nb, dtype=np.int32)
        _check(self.lib.emme_null_vectors_batch(self.h, n, nb, ptr, v.ctypes.data, info.ctypes.data))
        return v, info

    def final_matrix(self, b=0):
        M = np.zeros((self.dim, self.dim), dtype=np.complex128)
        _check(self.lib.emme_ctx_get_matrix(self.h, b, M.ctypes.data))
        return M
