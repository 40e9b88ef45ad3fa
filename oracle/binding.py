"""ctypes bindings for the CHECKERS under oracle/ -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (emme_amd/) never does.

  * liboracle.so          plain-C restatement (oracle/emme_oracle.c)
  * _ref/libemme_ref.so   the reference's own kappa/quadrature sources, compiled unmodified
                          by oracle/Makefile (present only if it was built in the container
                          that has /root/reference; it travels to the GPU box as a built file)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

CONF = {"tokamak": 0, "stellarator": 1, "cylinder": 2, "taloyMagneticDrift": 3, "cylinder old": 4}


class Params(C.Structure):
    """Mirror of include/emme_params.h::emme_params_t (field order matters)."""

    _fields_ = [
        ("conf", C.c_int),
        ("iteration_method", C.c_int),
        ("q", C.c_double), ("shat", C.c_double), ("tau", C.c_double),
        ("epsilon_n", C.c_double), ("epsilon_r", C.c_double),
        ("eta_i", C.c_double), ("eta_e", C.c_double),
        ("k_rho", C.c_double),
        ("beta_e", C.c_double), ("R", C.c_double), ("vt", C.c_double),
        ("omega_d_coeff", C.c_double), ("length", C.c_double), ("theta", C.c_double),
        ("npoints", C.c_int),
        ("iteration_step_limit", C.c_int),
        ("integration_precision", C.c_double),
        ("integration_accuracy", C.c_double),
        ("integration_iteration_limit", C.c_int),
        ("integration_start_points", C.c_int),
        ("arc_coeff", C.c_double),
        ("water_bag_weight_vpara", C.c_double), ("water_bag_weight_vperp", C.c_double),
        ("drift_center_transformation_switch", C.c_int),
        ("iteration_precision", C.c_double),
        ("initial_guess", C.c_double * 2),
        ("eta_k", C.c_double),
        ("lh", C.c_int), ("mh", C.c_int),
        ("epsilon_h_t", C.c_double), ("alpha_0", C.c_double), ("r_over_R", C.c_double),
        ("b_theta", C.c_double),
        ("alpha", C.c_double), ("omega_s_i", C.c_double), ("omega_s_e", C.c_double),
        ("omega_d_bar", C.c_double),
        ("deltap", C.c_double), ("beta_e_p", C.c_double), ("rdeltapp", C.c_double),
        ("curvature_aver", C.c_double),
        ("shat_coeff", C.c_double),
    ]


_RAW_DOUBLE = ["q", "shat", "tau", "epsilon_n", "epsilon_r", "eta_i", "eta_e", "k_rho", "beta_e",
               "R", "vt", "omega_d_coeff", "length", "theta", "integration_precision",
               "integration_accuracy", "arc_coeff", "water_bag_weight_vpara",
               "water_bag_weight_vperp", "iteration_precision"]
_RAW_INT = ["npoints", "iteration_step_limit", "integration_iteration_limit",
            "integration_start_points"]
_STELL_DOUBLE = ["eta_k", "epsilon_h_t", "alpha_0", "r_over_R"]
_STELL_INT = ["lh", "mh"]


def params_from_dict(d: dict) -> Params:
    """Raw (underived) parameter block from an input dict with ordinary JSON semantics."""
    p = Params()
    p.conf = CONF[d["conf"]]
    p.iteration_method = 0 if d.get("iteration_method", "TraceSecant") == "TraceSecant" else 1
    for k in _RAW_DOUBLE:
        setattr(p, k, float(d[k]))
    for k in _RAW_INT:
        setattr(p, k, int(d[k]))
    p.drift_center_transformation_switch = int(bool(d["drift_center_transformation_switch"]))
    g = d.get("initial_guess", [0.0, 0.0])
    p.initial_guess[0], p.initial_guess[1] = float(g[0]), float(g[1])
    if p.conf == CONF["stellarator"]:
        for k in _STELL_DOUBLE:
            setattr(p, k, float(d[k]))
        for k in _STELL_INT:
            setattr(p, k, int(d[k]))
    return p


def _fnum(x) -> str:
    if isinstance(x, bool):
        return "true" if x else "false"
    if isinstance(x, int):
        return str(x)
    s = repr(float(x))
    if "e" in s:  # the reference lexer calls a token FLOAT only if it contains '.'
        m, e = s.split("e")
        if "." not in m:
            m += ".0"
        return m + "e" + e
    return s


def json_text(d: dict) -> str:
    """JSON text that the reference's parser reads with the intended values
    (src/JsonParser.cpp:440: a number without '.' is an INTEGER and goes through atoi)."""
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


def build(force: bool = False) -> None:
    """Compile liboracle.so (always possible) and _ref (only where /root/reference exists)."""
    if force or not os.path.exists(os.path.join(HERE, "liboracle.so")) or \
            os.path.getmtime(os.path.join(HERE, "liboracle.so")) < os.path.getmtime(
                os.path.join(HERE, "emme_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])
    if os.path.isdir("/root/reference/src") and (
            force or not os.path.exists(os.path.join(HERE, "_ref", "libemme_ref.so"))):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_D, _I, _L, _P = C.c_double, C.c_int, C.c_long, C.c_void_p


class Oracle:
    def __init__(self):
        build()
        self.lib = lib = C.CDLL(os.path.join(HERE, "liboracle.so"))
        PP = C.POINTER(Params)
        lib.oracle_params_derive.argtypes = [PP]
        lib.oracle_grid.argtypes = [_D, C.c_uint, _P]
        lib.oracle_grid.restype = _D
        lib.oracle_weight.argtypes = [_I, _I, _I]
        lib.oracle_weight.restype = _D
        lib.oracle_g.argtypes = [PP, _D]
        lib.oracle_g.restype = _D
        lib.oracle_bi.argtypes = [PP, _D]
        lib.oracle_bi.restype = _D
        lib.oracle_bessel.argtypes = [_D, _D, _P]
        lib.oracle_kappa.argtypes = [PP, C.c_uint, _D, _D, _D, _D, _I, _P]
        lib.oracle_kappa.restype = _L
        lib.oracle_kappa_e.argtypes = [PP, C.c_uint, _D, _D, _D, _D, _P]
        lib.oracle_integrate_test.argtypes = [_D, _D, _D, _D, _D, C.c_ulong, C.c_ulong, _P]
        lib.oracle_integrate_test.restype = _L
        lib.oracle_assemble.argtypes = [PP, _D, _D, _P, _I, _I, _P, _P]
        lib.oracle_trace_solve.argtypes = [_I, _P, _P, _P]
        lib.oracle_solve_root.argtypes = [PP, _D, _D, _I, _I, _P, _P, _P, _P]

    def params(self, d: dict) -> Params:
        p = params_from_dict(d)
        self.lib.oracle_params_derive(C.byref(p))
        return p

    def grid(self, length, n):
        eta = np.zeros(n)
        dx = self.lib.oracle_grid(length, n, eta.ctypes.data)
        return eta, dx

    def weights(self, n):
        return np.array([[self.lib.oracle_weight(n, i, j) for j in range(n)] for i in range(n)])

    def g(self, p, eta):
        return self.lib.oracle_g(C.byref(p), eta)

    def bi(self, p, eta):
        return self.lib.oracle_bi(C.byref(p), eta)

    def bessel(self, z: complex):
        o = np.zeros(8)
        self.lib.oracle_bessel(z.real, z.imag, o.ctypes.data)
        return o.view(np.complex128)

    def kappa(self, p, m, eta, eta_p, omega: complex, recompute=0):
        o = np.zeros(2)
        n = self.lib.oracle_kappa(C.byref(p), m, eta, eta_p, omega.real, omega.imag, recompute,
                                  o.ctypes.data)
        return complex(o[0], o[1]), n

    def kappa_e(self, p, m, eta, eta_p, omega: complex):
        o = np.zeros(2)
        self.lib.oracle_kappa_e(C.byref(p), m, eta, eta_p, omega.real, omega.imag, o.ctypes.data)
        return complex(o[0], o[1])

    def integrate_test(self, a: complex, pw, tol, prec, max_sub, pts):
        o = np.zeros(2)
        n = self.lib.oracle_integrate_test(a.real, a.imag, pw, tol, prec, max_sub, pts,
                                           o.ctypes.data)
        return complex(o[0], o[1]), n

    def dim(self, p):
        return p.npoints if p.beta_e == 0.0 else 2 * p.npoints

    def assemble(self, p, omega: complex, nthreads=0, recompute=0, want_counts=False):
        nthreads = nthreads or os.cpu_count()
        dim = self.dim(p)
        M = np.zeros((dim, dim), dtype=np.complex128)
        counts = np.zeros((p.npoints, p.npoints), dtype=np.int64) if want_counts else None
        tot = C.c_long(0)
        rc = self.lib.oracle_assemble(C.byref(p), omega.real, omega.imag, M.ctypes.data, nthreads,
                                      recompute, counts.ctypes.data if want_counts else None,
                                      C.byref(tot))
        if rc < 0:
            raise RuntimeError(f"oracle_assemble failed rc={rc}")
        return (M, counts, tot.value) if want_counts else (M, tot.value)

    def trace_solve(self, A, B):
        A = np.array(A, dtype=np.complex128, order="C")
        B = np.array(B, dtype=np.complex128, order="C")
        tr = np.zeros(2)
        info = self.lib.oracle_trace_solve(A.shape[0], A.ctypes.data, B.ctypes.data,
                                           tr.ctypes.data)
        return complex(tr[0], tr[1]), info

    # newtonQRSecantIteration's linear algebra (include/solver.h:210-383): the same LAPACK
    # routines in the same order, through SciPy's LAPACK.  Returns (domega, info) with the
    # reference's failure convention (info > 0: ztrtrs met a zero diagonal, :309-316).
    # Parity at this boundary is unpinned in the sense of SURVEY.md §8(c): the reference holds
    # no vectors for it and links whatever LAPACK the site provides.
    @staticmethod
    def qr_secant(A, B):
        from scipy.linalg import lapack
        A = np.array(A, dtype=np.complex128)
        B = np.array(B, dtype=np.complex128)
        n = A.shape[0]
        # :240-251  column-major copy of M, zgeqp3 with all columns free
        qr, jpvt, tau, _, info = lapack.zgeqp3(np.asfortranarray(A))
        if info != 0:
            raise RuntimeError("QR factorization with pivoting failed")
        # :289-307  R11 y = r12
        if n > 1:
            y, info = lapack.ztrtrs(np.asfortranarray(np.triu(qr[:n - 1, :n - 1])),
                                    qr[:n - 1, n - 1].copy(), lower=0, trans=0, unitdiag=0)
            if info != 0:
                return complex("nan"), int(info)
        else:
            y = np.zeros(0, dtype=np.complex128)
        # :329-333  v = P [-y; 1]
        v = np.zeros(n, dtype=np.complex128)
        v[jpvt[:n - 1] - 1] = -y
        v[jpvt[n - 1] - 1] = 1.0
        # :340-346  t = M' v, summed in index order like the reference's loop
        t = np.zeros(n, dtype=np.complex128)
        for j in range(n):
            t += B[:, j] * v[j]
        # :361-363  t <- Q^H t
        cq, _, info = lapack.zunmqr("L", "C", qr, tau, np.asfortranarray(t.reshape(n, 1)), n * n)
        if info != 0:
            raise RuntimeError("Q application failed")
        return -qr[n - 1, n - 1] / cq[n - 1, 0], 0  # :370

    # solve_once_eigen (src/main.cpp:19-80) with the QR-secant step, at Python level on top of
    # the C restatement's fill.  Returns (root, iterates).
    def solve_root_qr(self, p, guess: complex, nthreads=0):
        w = 0.99 * guess  # EigenSolver ctor, include/solver.h:396-415
        dw = 0.01 * guess
        m_old, _ = self.assemble(p, w, nthreads)
        w = w + dw
        m, _ = self.assemble(p, w, nthreads)
        mp = (m - m_old) / dw
        its = []
        for _ in range(p.iteration_step_limit + 1):  # src/main.cpp:43-57
            m_old = m
            dw, info = self.qr_secant(m, mp)
            if info != 0:
                return complex("nan"), np.array(its)
            w = w + dw
            m, _ = self.assemble(p, w, nthreads)
            mp = (m - m_old) / dw
            its.append(w)
            if abs(dw) < p.iteration_precision * abs(w):
                break
        return w, np.array(its)

    def solve_root(self, p, guess: complex, nthreads=0, recompute=0, want_matrix=False):
        nthreads = nthreads or os.cpu_count()
        root = np.zeros(2)
        its = np.zeros(2 * (p.iteration_step_limit + 2))
        dim = self.dim(p)
        Mf = np.zeros((dim, dim), dtype=np.complex128) if want_matrix else None
        tot = C.c_long(0)
        n = self.lib.oracle_solve_root(C.byref(p), guess.real, guess.imag, nthreads, recompute,
                                       root.ctypes.data, its.ctypes.data,
                                       Mf.ctypes.data if want_matrix else None, C.byref(tot))
        if n < 0:
            raise RuntimeError(f"oracle_solve_root failed rc={n}")
        iters = its[:2 * n].view(np.complex128).copy()
        return complex(root[0], root[1]), iters, Mf, tot.value


class Reference:
    """The reference's own compiled kappa path (oracle/_ref). None if not built."""

    @staticmethod
    def available() -> bool:
        build()
        return os.path.exists(os.path.join(HERE, "_ref", "libemme_ref.so"))

    def __init__(self):
        build()
        self.lib = lib = C.CDLL(os.path.join(HERE, "_ref", "libemme_ref.so"))
        lib.ref_last_error.restype = C.c_char_p
        lib.ref_open.argtypes = [C.c_char_p]
        lib.ref_params.argtypes = [_P, _I]
        for f in ("ref_g", "ref_bi"):
            getattr(lib, f).argtypes = [_D]
            getattr(lib, f).restype = _D
        for f in ("ref_beta_1", "ref_beta_1_e"):
            getattr(lib, f).argtypes = [_D, _D]
            getattr(lib, f).restype = _D
        lib.ref_kappa.argtypes = [C.c_uint, _D, _D, _D, _D, _P]
        lib.ref_kappa_e.argtypes = [C.c_uint, _D, _D, _D, _D, _P]
        lib.ref_bessel.argtypes = [_D, _D, _P]
        lib.ref_singularity.argtypes = [_I, _P]
        lib.ref_grid.argtypes = [_D, C.c_uint, _P]
        lib.ref_grid.restype = _D
        lib.ref_integrate_test.argtypes = [_D, _D, _D, _D, _D, C.c_ulong, C.c_ulong, _P]
        lib.ref_assemble.argtypes = [_D, _D, _P, _I]

    PARAM_NAMES = ["q", "shat", "tau", "epsilon_n", "epsilon_r", "eta_i", "eta_e", "b_theta",
                   "beta_e", "R", "vt", "omega_d_coeff", "length", "theta", "npoints",
                   "iteration_step_limit", "integration_precision", "integration_accuracy",
                   "integration_iteration_limit", "integration_start_points", "arc_coeff",
                   "alpha", "omega_s_i", "omega_s_e", "omega_d_bar"]

    def open(self, text: str):
        if self.lib.ref_open(text.encode()) != 0:
            raise RuntimeError(self.lib.ref_last_error().decode())

    def open_dict(self, d: dict):
        self.open(json_text(d))

    def params(self) -> dict:
        v = np.zeros(len(self.PARAM_NAMES))
        self.lib.ref_params(v.ctypes.data, len(v))
        return dict(zip(self.PARAM_NAMES, v))

    def g(self, eta):
        return self.lib.ref_g(eta)

    def bi(self, eta):
        return self.lib.ref_bi(eta)

    def kappa(self, m, eta, eta_p, omega: complex):
        o = np.zeros(2)
        self.lib.ref_kappa(m, eta, eta_p, omega.real, omega.imag, o.ctypes.data)
        return complex(o[0], o[1])

    def kappa_e(self, m, eta, eta_p, omega: complex):
        o = np.zeros(2)
        self.lib.ref_kappa_e(m, eta, eta_p, omega.real, omega.imag, o.ctypes.data)
        return complex(o[0], o[1])

    def bessel(self, z: complex):
        o = np.zeros(8)
        self.lib.ref_bessel(z.real, z.imag, o.ctypes.data)
        return o.view(np.complex128)

    def weights(self, n):
        w = np.zeros((n, n))
        self.lib.ref_singularity(n, w.ctypes.data)
        return w

    def grid(self, length, n):
        eta = np.zeros(n)
        dx = self.lib.ref_grid(length, n, eta.ctypes.data)
        return eta, dx

    def integrate_test(self, a: complex, pw, tol, prec, max_sub, pts):
        o = np.zeros(2)
        self.lib.ref_integrate_test(a.real, a.imag, pw, tol, prec, max_sub, pts, o.ctypes.data)
        return complex(o[0], o[1])

    def assemble(self, dim, omega: complex, nthreads=0):
        nthreads = nthreads or os.cpu_count()
        M = np.zeros((dim, dim), dtype=np.complex128)
        rc = self.lib.ref_assemble(omega.real, omega.imag, M.ctypes.data, nthreads)
        if rc < 0:
            raise RuntimeError(self.lib.ref_last_error().decode())
        return M


# --- the two shipped example inputs, as data (values of input-example.json:1-37 and
# input-stellarator-example.json:1-33; edits per SURVEY.md App. C / §8(d)) ---------------
def example_tokamak(**over) -> dict:
    d = {
        "conf": "tokamak", "method": "eigen", "q": 1.4, "shat": 0.78, "tau": 1.0,
        "epsilon_n": 0.45, "epsilon_r": 0.0, "eta_i": 3.13, "eta_e": 3.13, "k_rho": 0.3182,
        "beta_e": 0.0, "R": 1.0, "vt": 1.0, "omega_d_coeff": 1.01, "length": 20.0, "theta": 0.0,
        "npoints": 64, "iteration_step_limit": 20, "initial_guess": [-0.8, 0.25],
        "integration_precision": 1.0e-6, "integration_accuracy": 1.0e-6,
        "integration_iteration_limit": 100, "integration_start_points": 15, "arc_coeff": 100.0,
        "iteration_precision": 1.0e-6, "iteration_method": "TraceSecant",
        "water_bag_weight_vpara": 1.0, "water_bag_weight_vperp": 1.0,
        "drift_center_transformation_switch": True,
    }
    d.update(over)
    return d


def example_stellarator(**over) -> dict:
    d = {
        "conf": "stellarator", "q": 2.0, "shat": -1.0, "tau": 1.0, "epsilon_n": 0.3,
        "eta_i": 3.0, "eta_e": 3.0, "k_rho": 0.247487, "beta_e": 0.02, "R": 1.0, "vt": 1.0,
        "length": 10.0, "theta": 0.0, "npoints": 32, "iteration_step_limit": 100,
        "integration_precision": 1.0e-5, "integration_accuracy": 1.0e-2,
        "integration_iteration_limit": 20, "integration_start_points": 31, "arc_coeff": 100.0,
        "eta_k": 0.0, "lh": 2, "mh": 10, "epsilon_h_t": 1.0, "alpha_0": 0.0, "r_over_R": 0.1,
        "initial_guess": [-1.656, 2.490], "iteration_precision": 1.0e-6,
        # the 7 keys the current code requires but the shipped file lacks:
        "method": "eigen", "iteration_method": "TraceSecant", "epsilon_r": 0.0,
        "omega_d_coeff": 1.0, "water_bag_weight_vpara": 1.0, "water_bag_weight_vperp": 1.0,
        "drift_center_transformation_switch": True,
    }
    d.update(over)
    return d
