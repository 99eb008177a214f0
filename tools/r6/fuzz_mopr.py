#!/usr/bin/env python3
"""Randomised campaign over the operator x vector step of the dynamical correlations (qbh_mopr_terms_dev) on one GPU (round 6,
outside the GPU tier): RANDOM term lists -- products of 1..4 local factors with random complex coefficients, every term leading to the same
target sector -- on spin sectors (S+, S-, Sz) and on two-species fermion sectors (c+, c, n per site and species), against dense operators
on the full product / Fock space (Kronecker products; Jordan-Wigner signs: tests/test_gpu_mopr.py).
usage: python tools/r6/fuzz_mopr.py [cases=200] [seed=1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import quantum_basis_amd as q  # noqa: E402
from quantum_basis_amd import lattices  # noqa: E402
import test_gpu_mopr as T  # noqa: E402

_spin_ops, _ferm_ops = {}, {}


def main():
    kv = dict(a.split("=", 1) for a in sys.argv[1:])
    cases, seed = int(kv.get("cases", 200)), int(kv.get("seed", 1))
    rng = np.random.default_rng(seed)
    fails, done, t0, cnt = [], 0, time.time(), {"spin": 0, "fermion": 0}
    coef = lambda: complex(rng.normal(), rng.normal() if rng.integers(2) else 0.0)      # noqa: E731
    while done < cases:
        tag = ""
        try:
            if int(rng.integers(2)):
                n = int(rng.integers(4, 10))
                nd = int(rng.integers(0, n + 1))
                delta = int(rng.integers(-1, 2))
                if not 0 <= nd + delta <= n:
                    continue
                if n not in _spin_ops:
                    _spin_ops[n] = T._dense_spin_ops(n)
                ops = _spin_ops[n]
                terms = []
                for _ in range(int(rng.integers(1, 6))):
                    fs, d = [], 0
                    for _ in range(int(rng.integers(0, 4))):
                        nm = ["S+", "S-", "Sz"][int(rng.integers(3))]
                        fs.append((nm, int(rng.integers(n))))
                        d += {"S+": -1, "S-": 1, "Sz": 0}[nm]
                    while d != delta:                                   # factors that bring the term to the common target sector
                        nm = "S-" if d < delta else "S+"
                        fs.insert(int(rng.integers(len(fs) + 1)), (nm, int(rng.integers(n))))
                        d += 1 if nm == "S-" else -1
                    if not fs:
                        fs = [("Sz", int(rng.integers(n)))]
                    terms.append((coef(), fs))
                tag = "spin n %d nd %d delta %d terms %s" % (n, nd, delta, terms)
                dense = sum(c * np.linalg.multi_dot([ops[f] for f in fs] + [np.eye(1 << n)]) for c, fs in terms)
                old, new = T._patterns(n, nd), T._patterns(n, nd + delta)
                x = (rng.normal(size=len(old)) + 1j * rng.normal(size=len(old))).astype(np.complex128)
                want = dense[np.ix_(new, old)] @ x
                A = q.csr_mat.heisenberg(n, max(1, min(nd, n - 1)), lattices.chain(n))
                vx, vy = q.DeviceVec(A, len(old)), q.DeviceVec(A, len(new))
                vx.upload(x)
                assert q.moprXvec_terms("spin", n, nd, 0, terms, vx.ptr, vy.ptr) == len(new), "dim"
                got = vy.download()
                assert np.abs(got - want).max() <= 1e-12 * max(np.abs(want).max(), 1.0), ("spin", np.abs(got - want).max())
                vx.free(), vy.free()
                A.destroy()
                cnt["spin"] += 1
            else:
                n = int(rng.integers(2, 6))
                nu, nd = int(rng.integers(0, n + 1)), int(rng.integers(0, n + 1))
                dnu, dnd = int(rng.integers(-1, 2)), int(rng.integers(-1, 2))
                if not (0 <= nu + dnu <= n and 0 <= nd + dnd <= n):
                    continue
                if n not in _ferm_ops:
                    _ferm_ops[n] = T._dense_fermion_ops(2 * n)
                ops = _ferm_ops[n]
                terms = []
                for _ in range(int(rng.integers(1, 6))):
                    fs, d = [], [0, 0]
                    for _ in range(int(rng.integers(0, 4))):
                        nm, s, sp = ["c+", "c", "n"][int(rng.integers(3))], int(rng.integers(n)), int(rng.integers(2))
                        fs.append((nm, s, sp))
                        d[sp] += {"c+": 1, "c": -1, "n": 0}[nm]
                    for sp, tgt in ((0, dnu), (1, dnd)):
                        while d[sp] != tgt:
                            nm = "c+" if d[sp] < tgt else "c"
                            fs.insert(int(rng.integers(len(fs) + 1)), (nm, int(rng.integers(n)), sp))
                            d[sp] += 1 if nm == "c+" else -1
                    if not fs:
                        fs = [("n", int(rng.integers(n)), int(rng.integers(2)))]
                    terms.append((coef(), fs))
                tag = "fermion n %d nu %d nd %d -> (%+d, %+d) terms %s" % (n, nu, nd, dnu, dnd, terms)
                dense = sum(c * np.linalg.multi_dot([ops[(nm, s + sp * n)] for nm, s, sp in fs] + [np.eye(1 << (2 * n))]) for c, fs in terms)
                pu, pd_, pu2, pd2 = T._patterns(n, nu), T._patterns(n, nd), T._patterns(n, nu + dnu), T._patterns(n, nd + dnd)
                old = np.array([int(u) | (int(d) << n) for u in pu for d in pd_])
                new = np.array([int(u) | (int(d) << n) for u in pu2 for d in pd2])
                x = (rng.normal(size=len(old)) + 1j * rng.normal(size=len(old))).astype(np.complex128)
                want = dense[np.ix_(new, old)] @ x
                A = q.csr_mat.hubbard(max(n, 2), 1, 1, lattices.chain(max(n, 2)))          # (any handle: it lends its device and stream)
                vx, vy = q.DeviceVec(A, len(old)), q.DeviceVec(A, len(new))
                vx.upload(x)
                assert q.moprXvec_terms("fermion", n, nu, nd, terms, vx.ptr, vy.ptr) == len(new), "dim"
                got = vy.download()
                assert np.abs(got - want).max() <= 1e-12 * max(np.abs(want).max(), 1.0), ("fermion", np.abs(got - want).max())
                vx.free(), vy.free()
                A.destroy()
                cnt["fermion"] += 1
        except Exception as e:      # noqa: BLE001
            fails.append((tag, repr(e)[:300]))
            print("FAIL", tag[:500], "::", repr(e)[:300], flush=True)
        done += 1
    print("fuzz_mopr: %d cases %s, %d failures, %.0f s (seed %d)" % (done, cnt, len(fails), time.time() - t0, seed))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
