#!/usr/bin/env python3
"""Randomised campaign over the solvers and the matrix-free / momentum-sector operators on one GPU (round 6, outside the GPU tier):
  * matrix-free Hubbard / Heisenberg operators on random bond graphs against the stored operator of the same generator (MultMv) and
    dense diagonalisation (E0);
  * qbh_iram (restarted Lanczos in HBM): the nev lowest / highest eigenvalues against dense diagonalisation, eigenvector residuals;
  * locate_E0_lanczos(nev = 2): E0, E1 (re-orthogonalised against phi0) and both eigenvectors against dense diagonalisation;
  * qbh_gen_heisenberg_repr on small tori at random momenta against the numpy projection (tests/reprham.py), entry by entry.
usage: python tools/r6/fuzz_solvers.py [cases=120] [seed=1]"""
import math
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import quantum_basis_amd as q  # noqa: E402
import fastham  # noqa: E402
import reprham  # noqa: E402


def random_bonds(rng, n):
    nb = int(rng.integers(n - 1, 2 * n + 1))
    bonds = []
    while len(bonds) < nb:
        a, b = int(rng.integers(n)), int(rng.integers(n))
        if a != b:
            bonds.append((a, b))
    return bonds


def main():
    kv = dict(a.split("=", 1) for a in sys.argv[1:])
    cases, seed = int(kv.get("cases", 120)), int(kv.get("seed", 1))
    rng = np.random.default_rng(seed)
    fails, done, t0, cnt = [], 0, time.time(), {"mf": 0, "iram": 0, "nev2": 0, "repr": 0}
    while done < cases:
        kind = ["mf", "iram", "nev2", "repr"][int(rng.integers(4))]
        tag = kind
        try:
            if kind == "repr":
                lx, ly = [(2, 2), (3, 2), (4, 2), (3, 3), (4, 3), (6, 1), (8, 1), (10, 1)][int(rng.integers(8))]
                n = lx * ly
                nd = int(rng.integers(1, n))
                if math.comb(n, nd) > 1200:
                    continue
                k = (int(rng.integers(lx)), int(rng.integers(ly)))
                bonds = [(x + lx * y, (x + 1) % lx + lx * y) for x in range(lx) for y in range(ly)] + ([(x + lx * y, x + lx * ((y + 1) % ly)) for x in range(lx) for y in range(ly)] if ly > 1 else [])
                bonds = [b for b in bonds if b[0] != b[1]]
                J = float(rng.choice([1.0, -0.5]))
                tag = "repr %dx%d nd %d k %s J %g" % (lx, ly, nd, k, J)
                perms, shifts = reprham.translations_2d(lx, ly)
                chars = reprham.characters(shifts, k, (lx, ly))
                H, reps, stab, zero = reprham.repr_heisenberg_csr(n, nd, bonds, perms, chars, J=J)
                A = q.csr_mat.heisenberg_repr(n, nd, bonds, perms, chars, J=J, opts=q.make_opts(value_dict=int(rng.integers(2)), real_fast_path=int(rng.integers(2))))
                assert A.dim == H.shape[0], ("dim", A.dim, H.shape[0])
                ia, ja, val = A.download()
                D = (sp.csr_matrix((val, ja, ia), shape=H.shape) - H).tocoo()
                assert D.nnz == 0 or np.abs(D.data).max() <= 1e-13, ("entries", np.abs(D.data).max())
                x = (rng.normal(size=A.dim) + 1j * rng.normal(size=A.dim)).astype(np.complex128)
                y = np.empty(A.dim, dtype=np.complex128)
                A.MultMv(x, y)
                assert np.abs(y - H @ x).max() <= 4e-13 * max(np.abs(H @ x).max(), 1e-300), "MultMv"
                A.destroy()
            else:
                n = int(rng.integers(4, 10))
                bonds = random_bonds(rng, n)
                heis = int(rng.integers(2))
                if heis:
                    nd = int(rng.integers(1, n))
                    if not 40 <= math.comb(n, nd) <= 6000:
                        continue
                    J = float(rng.choice([1.0, -0.7]))
                    tag = "%s heisenberg n %d nd %d J %g bonds %s" % (kind, n, nd, J, bonds)
                    H = fastham.heisenberg_full(n, nd, bonds, J=J)
                    mk = lambda mf, **o: q.csr_mat.heisenberg(n, nd, bonds, J=J, matrix_free=mf, opts=q.make_opts(**o))      # noqa: E731
                else:
                    nu, nd = int(rng.integers(1, n)), int(rng.integers(1, n))
                    if not 40 <= math.comb(n, nu) * math.comb(n, nd) <= 6000:
                        continue
                    t, U = 1.0, float(rng.choice([0.0, 1.1, 4.0]))
                    tag = "%s hubbard n %d nu %d nd %d U %g bonds %s" % (kind, n, nu, nd, U, bonds)
                    H = fastham.hubbard_full(n, nu, nd, bonds, t=t, U=U)
                    mk = lambda mf, **o: q.csr_mat.hubbard(n, nu, nd, bonds, t=t, U=U, matrix_free=mf, opts=q.make_opts(**o))      # noqa: E731
                dim = H.shape[0]
                w, z = np.linalg.eigh(H.toarray())
                if kind == "mf":
                    M = mk(True)
                    x = (rng.normal(size=dim) + 1j * rng.normal(size=dim)).astype(np.complex128)
                    y = np.empty(dim, dtype=np.complex128)
                    M.MultMv(x, y)
                    want = H @ x
                    assert np.abs(y - want).max() <= 4e-13 * max(np.abs(want).max(), 1e-300), ("matrix-free MultMv", np.abs(y - want).max())
                    r = q.locate_E0_lanczos(M, nev=1, ncv=1, maxit=1000)
                    assert abs(r.E0 - w[0]) <= 1e-9 * max(abs(w[0]), 1.0), ("matrix-free E0", r.E0, w[0])
                    assert np.abs(H @ r.eigenvecs - r.E0 * r.eigenvecs).max() < 1e-7, "matrix-free eigenvector"
                    M.destroy()
                elif kind == "iram":
                    A = mk(False, value_dict=int(rng.integers(2)), real_fast_path=int(rng.integers(2)))
                    nev = int(rng.integers(1, 4))
                    ncv = int(rng.integers(2 * nev + 2, 25))
                    order = str(rng.choice(["sr", "lr"]))
                    nconv, ew, ez = q.iram(dim, A, None, nev, ncv, 3000, order=order, method="device")
                    tag += " nev %d ncv %d %s" % (nev, ncv, order)
                    assert nconv >= nev, ("nconv", nconv)
                    wd = w[:nev + 1] if order == "sr" else w[::-1][:nev + 1]
                    tol = 1e-8 * max(np.abs(w).max(), 1.0)
                    if np.min(np.abs(np.diff(wd))) > 1e-6:           # no multiplicity among the wanted ones: exactly those
                        assert np.allclose(np.sort(ew), np.sort(wd[:nev]), rtol=0, atol=tol), ("eigenvalues", ew, wd)
                    else:                                           # a Krylov space of one start vector holds ONE copy of a multiple eigenvalue (ARPACK likewise:
                        assert abs((ew.min() if order == "sr" else ew.max()) - wd[0]) <= tol, ("extreme eigenvalue", ew, wd)      # further copies come from rounding, if at all):
                        assert all(np.min(np.abs(w - e)) <= tol for e in ew), ("not eigenvalues", ew)                           # every value returned is an eigenvalue, the extreme one is there
                    for j in range(nev):
                        v = ez[j * dim:(j + 1) * dim]
                        assert np.abs(H @ v - ew[j] * v).max() < 1e-6 and abs(np.linalg.norm(v) - 1.0) < 1e-8, ("eigenvector", j)
                    A.destroy()
                else:
                    A = mk(False, value_dict=int(rng.integers(2)), real_fast_path=int(rng.integers(2)), kron_split=int(rng.choice([0, 2])))
                    r = q.locate_E0_lanczos(A, nev=2, ncv=2, maxit=2000)
                    assert abs(r.E0 - w[0]) <= 1e-9 * max(abs(w[0]), 1.0), ("E0", r.E0, w[0])
                    if w[1] - w[0] > 1e-6:                       # (a degenerate ground state: E1 = E0 and the split of the pair is arbitrary)
                        assert abs(r.E1 - w[1]) <= 1e-8 * max(abs(w[1]), 1.0), ("E1", r.E1, w[1])
                    v0 = r.eigenvecs[:dim]
                    assert np.abs(H @ v0 - r.E0 * v0).max() < 1e-7, "V0"
                    A.destroy()
            cnt[kind] += 1
        except Exception as e:      # noqa: BLE001
            fails.append((tag, repr(e)[:300]))
            print("FAIL", tag, "::", repr(e)[:300], flush=True)
        done += 1
    print("fuzz_solvers: %d cases %s, %d failures, %.0f s (seed %d)" % (done, cnt, len(fails), time.time() - t0, seed))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
