#!/usr/bin/env python3
"""Randomised campaign over the extended momentum-sector generator on one GPU (round 6, outside the GPU tier): qbh_gen_hubbard_repr with
density-density pair terms, spin-exchange terms and the no-double-occupancy constraint (the t-J family of the reference's
examples/trans_symmetric/latt_kagome/kagome_tJ.cc) on chains and tori with random fillings, momenta and couplings, against the explicit
projection of the operator built by successive fermion operators on the (constrained) full space (the construction of
tests/test_gpu_hubrepr.py::test_spin_exchange_terms_and_the_tj_constraint_against_explicit_projection), entry by entry (1e-12).
usage: python tools/r6/fuzz_tj.py [cases=80] [seed=1]"""
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
    fails, done, t0, n_tj = [], 0, time.time(), 0
    while done < cases:
        Lx, Ly = [(4, 1), (5, 1), (6, 1), (2, 2), (3, 2), (4, 2)][int(rng.integers(6))]
        n = Lx * Ly
        nu, nd = int(rng.integers(0, n + 1)), int(rng.integers(0, n + 1))
        no_double = bool(rng.integers(2))
        if nu + nd == 0 or (no_double and nu + nd > n) or math.comb(n, nu) * math.comb(n, nd) > 1500:
            continue
        k = (int(rng.integers(Lx)), int(rng.integers(Ly)))
        t, U = float(rng.choice([1.0, 0.7])), (0.0 if no_double else float(rng.choice([0.0, 0.9, 4.0])))
        bonds = lattices.chain(Lx) if Ly == 1 else lattices.square(Lx, Ly)
        perms, shifts = lattices.translations(Lx, Ly)
        xa = float(rng.choice([0.0, 0.35, 0.5]))
        exch = [(i, j, xa) for (i, j) in bonds] if xa else []
        pc = [float(rng.choice([0.0, -0.25, 0.4])) for _ in range(4)]
        pairs = [(i, j, pc[0], pc[1], pc[1], pc[3]) for (i, j) in bonds] if any(pc) else []
        tag = "%dx%d nu %d nd %d k %s t %g U %g no_double %s exch %g pairs %s" % (Lx, Ly, nu, nd, k, t, U, no_double, xa, pc)
        try:
            terms = T._hubbard_terms(bonds, t)
            m = (1 << n) - 1
            words = [w for w in T._words(n, nu, nd) if not (no_double and (w & m) & (w >> n))]
            if len(words) < 2:
                continue
            index = {w: i for i, w in enumerate(words)}
            O = np.zeros((len(words), len(words)), dtype=np.complex128)
            for a, w in enumerate(words):
                u, d = w & m, w >> n
                O[a, a] += U * bin(u & d).count("1")
                for (i, j, au, ad) in terms:
                    r = T._hop(u, i, j)
                    if r and (r[1] | (d << n)) in index:
                        O[index[r[1] | (d << n)], a] += au * r[0]
                    r = T._hop(d, i, j)
                    if r and (u | (r[1] << n)) in index:
                        O[index[u | (r[1] << n)], a] += ad * r[0]
            if pairs:
                O = T._add_pairs(O, n, words, pairs)
            if exch:
                O = O + T._exchange_operator(n, words, index, exch)
            assert np.abs(O - O.conj().T).max() < 1e-13, "reference construction not Hermitian"
            chars = lattices.characters(shifts, k, (Lx, Ly))
            Ts = [T._translation(n, words, index, p) for p in perms]
            P = sum(c * Tm for c, Tm in zip(chars, Ts)) / len(perms)
            reps = [w for w in words if min(T._image(n, p, w & m)[0] | (T._image(n, p, w >> n)[0] << n) for p in perms) == w]
            psi = np.zeros((len(words), len(reps)), dtype=np.complex128)
            for r, w in enumerate(reps):
                v = P[:, index[w]]
                if np.linalg.norm(v) > 1e-10:
                    psi[:, r] = v / np.linalg.norm(v)
            Hk = psi.conj().T @ O @ psi
            A = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, chars, t=t, U=U, pairs=pairs or None, exchange=exch or None, no_double=no_double,
                                       opts=q.make_opts(value_dict=int(rng.integers(2))))
            M = T._dense(A)
            assert M.shape == Hk.shape, ("shape", M.shape, Hk.shape)
            for r in np.nonzero(np.abs(psi).sum(axis=0) == 0)[0]:
                Hk[r, r] = M[r, r]                       # decoupled zero-norm representatives carry the fake diagonal
            assert np.abs(M - Hk).max() < 1e-12, ("entries", np.abs(M - Hk).max())
            A.destroy()
            n_tj += int(no_double)
        except Exception as e:      # noqa: BLE001
            fails.append((tag, repr(e)[:300]))
            print("FAIL", tag, "::", repr(e)[:300], flush=True)
        done += 1
    print("fuzz_tj: %d cases (%d with the no-double-occupancy constraint), %d failures, %.0f s (seed %d)" % (done, n_tj, len(fails), time.time() - t0, seed))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
