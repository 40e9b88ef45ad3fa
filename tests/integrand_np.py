"""numpy restatement of the omega-split of the integrand (emme_amd/csrc/emme_device.hpp::node_data;
reference src/Parameters.cpp:120-176): for one pair and one abscissa x of the mapped integral,
F(omega) = exp(A0 + T omega) (omega Q1 + Q0), with the reference's safe_exp clamp at Re(A0 + T omega) < -40
(:167-173).  TEST INFRASTRUCTURE: used to CONSTRUCT cases (which nodes of which intervals are clamped), never
as an oracle for values -- those come from oracle/ and tests/golden/."""
import numpy as np

X15 = np.array([0., 0.20778495500789847, 0.40584515137739717, 0.58608723546769113, 0.74153118559939444,
                0.86486442335976907, 0.94910791234275852, 0.99145537112081264])


def _bessel_z4(z):
    """only the last output of util::bessel_i_alter_helper (include/functions.h:407): Re z < 0 ? z : -z"""
    return z if z.real < 0 else -z


class PairNodes:
    def __init__(self, orc, po, eta_i, eta_j, omi):
        gi, gj = orc.g(po, eta_i), orc.g(po, eta_j)
        bi, bj = orc.bi(po, eta_i), orc.bi(po, eta_j)
        qR, vt = po.q * po.R, po.vt
        self.de = eta_i - eta_j
        self.beta1 = qR / vt * po.omega_d_bar * (gi - gj)
        self.s = np.sqrt(bi * bj)
        self.bsum = bi + bj
        self.c_lam = 0.5 * vt / (qR * self.de) * self.beta1
        self.c_nv = qR * self.de / vt
        self.omi = omi
        self.inv_arc = 1.0 / po.arc_coeff

    def a0_t(self, x):
        """(A0, T) at abscissa x in (0, pi/2): Re(A0 + T omega) is the argument of the reference's safe_exp."""
        t = np.tan(x)
        u = t * self.inv_arc
        r1 = 1 / np.sqrt(1 + u * u)
        e = complex(r1, -self.omi * u * r1)
        taut = t * e
        lam = 1 + 1j * self.c_lam * taut
        rl = 1 / lam
        nv = self.c_nv / t * np.conj(e)
        L0 = -0.5 * nv * nv - 0.5j * self.beta1 * nv - 0.5 * self.bsum * rl
        return L0 - _bessel_z4(self.s * rl), 1j * taut

    def clamped_fraction(self, omega, depth=4):
        """fraction of the 15 x 2^depth Kronrod nodes of bisection level `depth` that safe_exp zeroes"""
        n, hit = 0, 0
        for k in range(1 << depth):
            l, r = np.pi / 2 * k / (1 << depth), np.pi / 2 * (k + 1) / (1 << depth)
            mid, scale = (r + l) / 2, (r - l) / 2
            for q in range(8):
                for sg in ((1,) if q == 0 else (1, -1)):
                    a0, t = self.a0_t(scale * sg * X15[q] + mid)
                    n += 1
                    hit += (a0 + t * omega).real < -40.0
        return hit / n
