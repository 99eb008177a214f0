#!/usr/bin/env python3
"""Randomised campaign over the Hubbard momentum-sector operators on one GPU (round 6, outside the GPU tier): chains and tori with random
fillings, momenta and U -- qbh_gen_hubbard_repr (stored, with and without value codes) against the explicit projection of the full-basis
operator (tests/test_gpu_hubrepr.py::_sector_reference), and qbh_mf_hubbard_repr (matrix-free, rows orbit by orbit or not) against the
stored sector on random vectors translated at the seams.
usage: python tools/r6/fuzz_hubrepr.py [cases=80] [seed=1]"""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import quantum_basis_amd as q  # noqa: E402
from quantum_basis_amd import lattices  # noqa: E402
import test_gpu_hubrepr as T  # noqa: E402


def main():
    kv = dict(a.split("=", 1) for a in sys.argv[1:])
    cases, seed = int(kv.get("cases", 80)), int(kv.get("seed", 1))
    rng = np.random.default_rng(seed)
    fails, done, t0, n_mf = [], 0, time.time(), 0
    while done < cases:
        Lx, Ly = [(4, 1), (5, 1), (6, 1), (7, 1), (2, 2), (3, 2), (4, 2), (3, 3)][int(rng.integers(8))]
        n = Lx * Ly
        nu, nd = int(rng.integers(0, n + 1)), int(rng.integers(0, n + 1))
        if nu + nd == 0 or math.comb(n, nu) * math.comb(n, nd) > 2500:
            continue
        k = (int(rng.integers(Lx)), int(rng.integers(Ly)))
        t, U = float(rng.choice([1.0, 0.7])), float(rng.choice([0.0, 1.1, 4.0]))
        bonds = lattices.chain(Lx) if Ly == 1 else lattices.square(Lx, Ly)
        perms, shifts = lattices.translations(Lx, Ly)
        chars = lattices.characters(shifts, k, (Lx, Ly))
        tag = "%dx%d nu %d nd %d k %s t %g U %g" % (Lx, Ly, nu, nd, k, t, U)
        try:
            reps, alive, Hk = T._sector_reference(n, nu, nd, T._hubbard_terms(bonds, t), U, perms, chars)
            if len(reps) < 2:
                continue
            A = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, chars, t=t, U=U, opts=q.make_opts(value_dict=int(rng.integers(2))))
            M = T._dense(A)
            assert M.shape == Hk.shape, ("shape", M.shape, Hk.shape)
            assert np.abs(M - Hk).max() < 1e-12, ("entries", np.abs(M - Hk).max())
            dim = M.shape[0]
            if nu > 0 and nd > 0 and dim >= 4:
                so = int(rng.integers(2))
                F = q.csr_mat.hubbard_repr_mf(n, nu, nd, bonds, perms, chars, t=t, U=U, opts=q.make_opts(sector_orbit=so))
                assert F.info().ncols == dim, ("matrix-free dim", F.info().ncols, dim)
                x = rng.normal(size=dim)                     # the matrix-free sector operator works on packed-real or complex vectors: use its MultMv seam
                xc = x.astype(np.complex128)
                y = np.empty(dim, dtype=np.complex128)
                try:
                    F.MultMv(xc, y)
                    want = Hk @ xc
                    assert np.abs(y - want).max() <= 1e-11 * max(np.abs(want).max(), 1.0), ("matrix-free MultMv", np.abs(y - want).max())
                    n_mf += 1
                except q._lib.QbhError as e:                 # momenta with complex characters are refused by the real-valued kernel: loudly
                    assert "real" in str(e).lower() or "unsupported" in str(e).lower() or "not supported" in str(e).lower(), str(e)
                F.destroy()
            A.destroy()
        except Exception as e:      # noqa: BLE001
            fails.append((tag, repr(e)[:300]))
            print("FAIL", tag, "::", repr(e)[:300], flush=True)
        done += 1
    print("fuzz_hubrepr: %d cases (%d with the matrix-free sector operator), %d failures, %.0f s (seed %d)" % (done, n_mf, len(fails), time.time() - t0, seed))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
