#!/usr/bin/env python3
"""Randomised campaign over the ALL-REAL drivers on one GPU (round 6, outside the GPU tier): qbh_lanczos_real_dev / qbh_eigenvec_cg_real_dev
(the caller's vectors are packed doubles; stored operator with value codes + real fast path, or matrix-free) on real operators on random
bond graphs against the COMPLEX interface on the same operator (coefficients 1e-9, step counts +-1 / +-2, E0, eigenvector) and dense
diagonalisation; continuation in two pieces at a random step.
usage: python tools/r6/fuzz_real.py [cases=150] [seed=1]"""
import ctypes as C
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import quantum_basis_amd as q  # noqa: E402
from quantum_basis_amd import _lib  # noqa: E402
import fastham  # noqa: E402


def main():
    kv = dict(a.split("=", 1) for a in sys.argv[1:])
    cases, seed = int(kv.get("cases", 150)), int(kv.get("seed", 1))
    rng = np.random.default_rng(seed)
    fails, done, t0, n_mf = [], 0, time.time(), 0
    while done < cases:
        n = int(rng.integers(6, 11))
        nb = int(rng.integers(n, 2 * n + 1))
        bonds = []
        while len(bonds) < nb:
            a, b = int(rng.integers(n)), int(rng.integers(n))
            if a != b:
                bonds.append((a, b))
        mf = bool(rng.integers(2))
        if int(rng.integers(3)) == 0:
            nd = int(rng.integers(2, n - 1))
            if not 200 <= math.comb(n, nd) <= 6000:
                continue
            J = float(rng.choice([1.0, -0.7]))
            H = fastham.heisenberg_full(n, nd, bonds, J=J)
            A = q.csr_mat.heisenberg(n, nd, bonds, J=J, matrix_free=mf, opts=q.make_opts(sector_cut=-1))
            tag = "heisenberg n %d nd %d J %g mf %d bonds %s" % (n, nd, J, mf, bonds)
        else:
            nu, nd = int(rng.integers(1, n)), int(rng.integers(1, n))
            if not 200 <= math.comb(n, nu) * math.comb(n, nd) <= 6000:
                continue
            U = float(rng.choice([1.1, 4.0]))
            H = fastham.hubbard_full(n, nu, nd, bonds, t=1.0, U=U)
            A = q.csr_mat.hubbard(n, nu, nd, bonds, t=1.0, U=U, matrix_free=mf)
            tag = "hubbard n %d nu %d nd %d U %g mf %d bonds %s" % (n, nu, nd, U, mf, bonds)
        try:
            dim, maxit = A.dim, 1000
            w = np.linalg.eigvalsh(H.toarray())
            ref = q.locate_E0_lanczos(A, nev=1, ncv=1, maxit=maxit)
            assert abs(ref.E0 - w[0]) <= 1e-9 * max(abs(w[0]), 1.0), ("complex interface E0", ref.E0, w[0])
            buf = A.vec(2)                                             # 2 dim complex128 = 4 dim doubles: v, r, p, pp
            at = lambda j: C.c_void_p(buf.ptr.value + 8 * dim * j)       # noqa: E731
            _lib.check(_lib.lib().qbh_vec_randomize_real(A.handle, at(0), C.c_uint32(1)), "qbh_vec_randomize_real")
            hess = np.zeros(2 * maxit)
            lan = type("V", (), {"ptr": at(0)})()
            cut = int(rng.integers(2, max(3, ref.steps["E0"] - 1)))
            m1 = q.lanczos_real(0, cut, maxit, A, lan, hess)
            m = m1 if m1 < cut else q.lanczos_real(m1, maxit - 1 - m1, maxit, A, lan, hess, state=q.lanczos_real.last["state"])
            ritz, _ = q.hess_eigen(hess, maxit, m, "sr")
            hc = ref.hessenberg_E0
            k = min(m, ref.steps["E0"], 15)
            bo = hc[1:ref.steps["E0"]]
            well = not (bo < 1e-6 * max(bo.max(), 1.0)).any()
            assert abs(ritz[0] - ref.E0) <= 1e-10 * max(abs(ref.E0), 1.0), ("E0", ritz[0], ref.E0)
            if well:
                assert abs(m - ref.steps["E0"]) <= 1, ("steps", m, ref.steps["E0"])
                sc = max(np.abs(hc[maxit:maxit + k]).max(), 1.0)
                assert np.allclose(hess[maxit:maxit + k], hc[maxit:maxit + k], rtol=0, atol=1e-9 * sc) and np.allclose(hess[1:k], hc[1:k], rtol=0, atol=1e-9 * sc), "a_j, b_j"
            _lib.check(_lib.lib().qbh_vec_randomize_real(A.handle, at(0), C.c_uint32(1)), "qbh_vec_randomize_real")
            mcg, accu = q.eigenvec_CG_real(maxit, 0, A, ritz[0], at(0), at(1), at(2), at(3))
            assert accu < 2e-12, ("CG", accu)
            vec = buf.download(0, dim // 2 + (dim & 1)).view(np.float64)[:dim]
            assert abs(np.linalg.norm(vec) - 1.0) < 1e-9, "norm"
            assert np.abs(H @ vec - ritz[0] * vec).max() < 1e-7, "eigenvector residual"
            if w[1] - w[0] > 1e-6 and well:
                assert abs(mcg - ref.steps["V0"]) <= max(3, ref.steps["V0"] // 20), ("CG steps", mcg, ref.steps["V0"])      # (the last steps of a long CG run hang on residuals near 2e-12)
                assert abs(abs(np.dot(vec, ref.eigenvecs.real)) - 1.0) < 1e-7, "the complex interface's eigenvector"
            st = A.stats()
            assert st.n_spmv_real > 0, "the all-real kernels were not used"
            n_mf += int(mf)
            buf.free()
            A.destroy()
        except Exception as e:      # noqa: BLE001
            fails.append((tag, repr(e)[:300]))
            print("FAIL", tag, "::", repr(e)[:300], flush=True)
        done += 1
    print("fuzz_real: %d cases (%d matrix-free), %d failures, %.0f s (seed %d)" % (done, n_mf, len(fails), time.time() - t0, seed))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
