#!/usr/bin/env python3
"""Ground-state energies of Hubbard momentum sectors on one GPU (qbh_gen_hubbard_repr + the device Lanczos driver).

    python tools/hubbard_sectors.py 4 4 8 8                 # all inequivalent momenta of the 4x4 torus at half filling
    python tools/hubbard_sectors.py 4 5 8 8 0 0             # one sector: 4x5, 8+8 electrons, k = (0,0): dim 7.9e8
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import quantum_basis_amd as q  # noqa: E402
from quantum_basis_amd import lattices  # noqa: E402


def main():
    mf = "--mf" in sys.argv                     # matrix-free sector operator (qbh_mf_hubbard_repr) instead of the stored CSR
    if mf:
        sys.argv.remove("--mf")
    Lx, Ly, nu, nd = (int(a) for a in sys.argv[1:5])
    n = Lx * Ly
    if len(sys.argv) > 6:
        ks = [(int(sys.argv[5]), int(sys.argv[6]))]
    else:
        ks = [(kx, ky) for kx in range(Lx // 2 + 1) for ky in range(Ly // 2 + 1)]
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    for k in ks:
        chars = lattices.characters(shifts, k, (Lx, Ly))
        t0 = time.time()
        A = (q.csr_mat.hubbard_repr_mf if mf else q.csr_mat.hubbard_repr)(n, nu, nd, bonds, perms, chars, t=1.0, U=1.1)
        i = A.info()
        t1 = time.time()
        maxit = 1000
        hess = np.zeros(2 * maxit)
        if mf and np.abs(np.imag(np.asarray(chars))).max() < 1e-14:
            # real sector: the Lanczos vectors are kept as packed doubles by the caller (two slots of dim doubles)
            import ctypes
            dv = A.vec(1)
            q._lib.check(q._lib.lib().qbh_vec_randomize_real(A.handle, dv.ptr, ctypes.c_uint32(7)), "qbh_vec_randomize_real")
            m = q.lanczos_real(0, maxit - 1, maxit, A, dv, hess)
        else:
            dv = A.vec(3)
            A.randomize(dv.at(0), 7)
            m = q.lanczos(0, maxit - 1, maxit, i.ncols, A, None, hess, "sr_val0", device_v=dv)
        ritz, _ = q.hess_eigen(hess, maxit, m, "sr")
        t2 = time.time()
        print(("matrix-free " if mf else "") + "k=%s dim %d nnz %d (%.1f GB, value_dict %d) build %.1f s; lanczos %d steps in %.1f s (%.1f ms/step); E0 = %.12f" %
              (k, i.ncols, i.nnz, i.bytes_matrix * 1e-9, i.value_dict, t1 - t0, m, t2 - t1, 1e3 * (t2 - t1) / max(1, m), ritz[0]), flush=True)
        dv.free()
        A.destroy()


if __name__ == "__main__":
    main()
