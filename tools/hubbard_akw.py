#!/usr/bin/env python3
"""Single-particle (photoemission) spectral function A(q, w) = sum_n |<n; N-1, q| c_{q,up} |psi0>|^2 delta(w - (E0 - E_n)) of a
Hubbard cluster, computed sector by sector on one GPU: ground state in its momentum sector (locate_E0_lanczos: Lanczos +
CG), c_{q,up}|psi0> into the (N_up - 1, N_dn) sector at momentum q (qbh_mopr_c_hubrepr_dev), "dnmcs" Lanczos there
(measure_repr_dynamic, src/model.cc:1896-1935).  Prints the weight per momentum, the sum rule and the lowest poles.

    python tools/hubbard_akw.py 4 4 8 8           # the 4x4 cluster at half filling (C3's family): 16 sectors of 5.6e6 states
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import quantum_basis_amd as q  # noqa: E402
from quantum_basis_amd import lattices  # noqa: E402


def main():
    Lx, Ly, nu, nd = (int(a) for a in sys.argv[1:5])
    k0 = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (0, 0)
    n = Lx * Ly
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    ch0 = lattices.characters(shifts, k0, (Lx, Ly))
    t0 = time.time()
    A0 = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, ch0)
    res = q.locate_E0_lanczos(A0, nev=1, ncv=1)
    dim0 = A0.info().ncols
    vphi = q.DeviceVec(A0, dim0)
    vphi.upload(res.eigenvecs)
    print("ground state of the k=%s sector (dim %d): E0 = %.12f, %d Lanczos + %d CG steps, %.1f s" %
          (k0, dim0, res.E0, res.steps["E0"], res.steps["V0"], time.time() - t0), flush=True)
    maxit = 300
    total = 0.0
    for qx in range(Lx):
        for qy in range(Ly):
            qv = (qx, qy)
            chq = lattices.characters(shifts, ((k0[0] + qx) % Lx, (k0[1] + qy) % Ly), (Lx, Ly))
            coef = np.array([np.exp(-2j * np.pi * (qx * (s % Lx) / Lx + qy * (s // Lx) / Ly)) for s in range(n)]) / np.sqrt(n)
            t1 = time.time()
            B = q.csr_mat.hubbard_repr(n, nu - 1, nd, bonds, perms, chq)
            m, norm, hess = q.measure_full_dynamic_dev(
                B, lambda dst: q.moprXvec_c_hubrepr(n, nu, nd, 0, -1, perms, ch0, chq, coef, vphi.ptr, dst), maxit)
            total += norm ** 2
            poles = ""
            if m > 1:
                ritz, s = q.hess_eigen(hess, maxit, m, "sr")
                wts = np.array(s).reshape(m, m, order="F")[0, :] ** 2          # |<phi|n>|^2 from the first components
                top = np.argsort(-wts)[:3]
                poles = "  strongest poles (w = E0 - E_n, weight): " + ", ".join("(%.4f, %.4f)" % (res.E0 - ritz[i], norm ** 2 * wts[i]) for i in sorted(top))
            print("q=%s  dim %d  weight <n_q> = %.6f  %d steps  %.2f s%s" % (qv, B.info().ncols, norm ** 2, m, time.time() - t1, poles), flush=True)
            B.destroy()
    print("sum rule: sum_q <c+_q c_q> = %.10f (N_up = %d)" % (total, nu))
    vphi.free()
    A0.destroy()


if __name__ == "__main__":
    main()
