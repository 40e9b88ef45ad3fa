#!/usr/bin/env python3
"""CPU emulation (numpy) of the device's omega-split integrand for ONE pair: direct vs folded
evaluation against the oracle's kappa -- where does the 5e-4 discrepancy at small Re omega > 0 come from?"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from oracle.binding import Oracle
orc = Oracle(); lib = orc.lib
lib.oracle_trace_set.argtypes = [C.c_void_p, C.c_long]; lib.oracle_trace_count.restype = C.c_long
d = bench.workload_dict(256); po = orc.params(d); N = 256
eta, dx = orc.grid(d["length"], N)
i, j = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 226
w = complex(sys.argv[3]) if len(sys.argv) > 3 else 0.05 + 0.01j
buf = np.zeros(8192, dtype=np.int64)
lib.oracle_trace_set(buf.ctypes.data, len(buf))
kap, n = orc.kappa(po, 0, eta[i], eta[j], w)
cnt = lib.oracle_trace_count(); lib.oracle_trace_set(None, 0)
print("oracle kappa", kap, "intervals", n)
gi, gj = orc.g(po, eta[i]), orc.g(po, eta[j]); bi, bj = orc.bi(po, eta[i]), orc.bi(po, eta[j])
qR, vt = po.q * po.R, po.vt
cb = qR / vt * po.omega_d_bar
de = eta[i] - eta[j]; beta1 = cb * (gi - gj); s = np.sqrt(bi * bj); bsum = bi + bj
c_lam = 0.5 * vt / (qR * de) * beta1; c_nv = qR * de / vt
omi = -np.copysign(1.0, w.real); inv_arc = 1.0 / po.arc_coeff
X = np.array([0., 0.20778495500789847, 0.40584515137739717, 0.58608723546769113, 0.74153118559939444, 0.86486442335976907, 0.94910791234275852, 0.99145537112081264])
WK = np.array([2.09482141084727828e-01, 2.04432940075298892e-01, 1.90350578064785410e-01, 1.69004726639267903e-01, 1.40653259715525919e-01, 1.04790010322250184e-01, 6.30920926299785533e-02, 2.29353220105292250e-02])

def bessel(z):
    n = int(np.floor(abs(z))) + 1
    p0, p1 = 0j, 1 + 0j
    test = max(np.sqrt(2e7 * abs(p1) * abs(p0 - 2 * n / z * p1)), 2e7)
    while abs(p1) <= test:
        p0, p1 = p1, p0 - 2.0 * n / z * p1
        n += 1
    y0, y1, mu = 1 / p1, 0j, 0j
    n -= 1
    while n > 0:
        y0, y1 = 2 * n / z * y0 + y1, y0
        mu += 2 * ((1 - 2 * (n & 1)) if z.real < 0 else 1) * y1
        n -= 1
    return y0, y1, mu + y0, (z if z.real < 0 else -z)

def node(x):
    t = np.tan(x); inv_c2 = 1 / np.cos(x) ** 2; u = t * inv_arc
    r1 = 1 / np.sqrt(1 + u * u); e = complex(r1, -omi * u * r1); taut = t * e
    lam = 1 + 1j * c_lam * taut; rl = 1 / lam; nv = c_nv / t * np.conj(e); nv2 = nv * nv
    z = s * rl
    L0 = -0.5 * nv2 - 0.5j * beta1 * nv - 0.5 * bsum * rl
    y0, y1, mut, z4 = bessel(z)
    rl3 = rl ** 3; wsi = po.omega_s_i; eta_i = po.eta_i
    c0 = -wsi * (1 + eta_i * (0.5 * nv2 - 1.5)) * rl + wsi * eta_i * (0.5 * bsum - lam) * rl3
    i1c = -wsi * eta_i * s * rl3
    pre = complex(1 / t, -(omi * u) * r1 * r1 / t) * inv_c2 / mut
    A0 = L0 - z4
    return A0, 1j * taut, pre * rl * y0, pre * (c0 * y0 + i1c * y1)

tot_d = tot_f = 0j
worst = (0, None)
for k in buf[:cnt]:
    depth, idx = int(k >> 56), int(k & ((1 << 56) - 1))
    l, r = 0.0, np.pi / 2
    for sft in range(depth - 1, -1, -1):
        mid = (r + l) / 2
        if (idx >> sft) & 1: l = mid
        else: r = mid
    mid, scale = (r + l) / 2, (r - l) / 2
    Kd = Kf = 0j
    for q in range(8):
        for sg in ((1,) if q == 0 else (1, -1)):
            A0, T, Q1, Q0 = node(scale * (sg * X[q]) + mid)
            v = A0 + T * w
            fd = 0 if v.real < -40 else np.exp(v) * (w * Q1 + Q0)
            ea = np.exp(min(A0.real, 700.0)) * complex(np.cos(A0.imag), np.sin(A0.imag))
            E = np.exp(T * w)
            a0 = np.exp(min(2 * A0.real, 700.0))
            ff = 0 if abs(E) ** 2 * a0 < 1.8048513878454153e-35 else E * (w * (ea * Q1) + ea * Q0)
            Kd += WK[q] * fd; Kf += WK[q] * ff
            if abs(fd - ff) > worst[0]: worst = (abs(fd - ff), (depth, idx, q, sg, A0, T * w, fd, ff))
    # which intervals were accepted? (leaf = no child in the trace)
    child = ((depth + 1) << 56) | (idx << 1)
    if child not in set(buf[:cnt].tolist()):
        tot_d += Kd * scale; tot_f += Kf * scale
pref = qR / (vt * np.sqrt(2 * np.pi))
kd, kf = -1j * pref * tot_d, -1j * pref * tot_f
print("emulated direct ", kd, "rel err vs oracle", abs(kd - kap) / abs(kap))
print("emulated folded ", kf, "rel err vs oracle", abs(kf - kap) / abs(kap))
print("worst node diff", worst)
