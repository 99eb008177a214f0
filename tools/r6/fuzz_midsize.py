#!/usr/bin/env python3
"""Randomised campaign at MID size on one GPU (round 6, outside the GPU tier): operators of 1e5 .. 6e6 dimensions (up to ~2e8 nonzeros) on
random bond graphs of 10..16 sites -- where launches span many wave blocks, chunks and XCD queues, which the small campaigns never reach --
the STORED operator under random options (split in place / sliced / padded, 2-byte columns, value codes, real fast path, static walks,
partition order, Heisenberg cut) against the MATRIX-FREE operator of the same model (an independent code path: hop tables, no matrix):
y = Hx on a random complex vector (1e-12 of |y|), <x, Hx> real, and the Lanczos ground-state energy of both (1e-10).
usage: python tools/r6/fuzz_midsize.py [cases=40] [seed=1] [lo=1e5] [hi=6e6]"""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import quantum_basis_amd as q  # noqa: E402


def main():
    kv = dict(a.split("=", 1) for a in sys.argv[1:])
    cases, seed = int(kv.get("cases", 40)), int(kv.get("seed", 1))
    lo, hi = float(kv.get("lo", 1e5)), float(kv.get("hi", 6e6))
    rng = np.random.default_rng(seed)
    fails, done, t0, n_split, nnz_max = [], 0, time.time(), 0, 0
    while done < cases:
        n = int(rng.integers(10, 19))
        nb = int(rng.integers(n, 2 * n + 1))
        bonds = []
        while len(bonds) < nb:
            a, b = int(rng.integers(n)), int(rng.integers(n))
            if a != b:
                bonds.append((a, b))
        heis = int(rng.integers(3)) == 0
        o = dict(kron_split=int(rng.choice([0, 2, 2])), kron_sliced=int(rng.integers(3)), kron_cols16=int(rng.integers(2)), value_dict=int(rng.choice([0, 0, 1])),
                 real_fast_path=int(rng.choice([0, 0, 1])), deterministic=int(rng.integers(2)))
        if heis:
            nd = int(rng.integers(2, n - 1))
            dim = math.comb(n, nd)
            if not lo <= dim <= hi:
                continue
            J = float(rng.choice([1.0, -0.7]))
            o["sector_cut"] = int(rng.choice([0, -1]))
            tag = "heisenberg n %d nd %d J %g bonds %s %s" % (n, nd, J, bonds, o)
            mk = lambda mf, **oo: q.csr_mat.heisenberg(n, nd, bonds, J=J, matrix_free=mf, opts=q.make_opts(**oo))      # noqa: E731
        else:
            nu, nd = int(rng.integers(1, n)), int(rng.integers(1, n))
            dim = math.comb(n, nu) * math.comb(n, nd)
            if not lo <= dim <= hi:
                continue
            U = float(rng.choice([0.0, 1.1, 4.0]))
            parts = int(rng.choice([0, 0, 3, 8]))
            if parts:
                o["major_partition"] = parts
            tag = "hubbard n %d nu %d nd %d U %g bonds %s %s" % (n, nu, nd, U, bonds, o)
            mk = lambda mf, **oo: q.csr_mat.hubbard(n, nu, nd, bonds, t=1.0, U=U, matrix_free=mf, opts=q.make_opts(**oo))      # noqa: E731
        try:
            A = mk(False, **o)
            info = A.info()
            n_split += int(info.kron_minor > 0)
            nnz_max = max(nnz_max, int(info.nnz))
            M = mk(True)
            x = (rng.normal(size=dim) + 1j * rng.normal(size=dim)).astype(np.complex128)
            if info.major_partition > 1:              # the stored operator's HOST vectors are in its partition order: permute x for it
                S = math.comb(n, nd)
                mo = A.major_order(math.comb(n, nu)).astype(np.int64)
                perm = (mo[:, None] * S + np.arange(S)[None, :]).ravel()
            else:
                perm = None
            ya, ym = np.empty(dim, dtype=np.complex128), np.empty(dim, dtype=np.complex128)
            A.MultMv(x if perm is None else np.ascontiguousarray(x[perm]), ya)
            M.MultMv(x, ym)
            if perm is not None:
                ya_g = np.empty_like(ya)
                ya_g[perm] = ya
                ya = ya_g
            scale = max(np.abs(ym).max(), 1e-300)
            assert np.abs(ya - ym).max() <= 1e-12 * scale, ("y = Hx", np.abs(ya - ym).max() / scale)
            assert abs(np.vdot(x, ya).imag) <= 1e-9 * abs(np.vdot(x, ya)), "<x, Hx> not real"
            ea = q.locate_E0_lanczos(A, nev=1, ncv=0, maxit=600).E0
            em = q.locate_E0_lanczos(M, nev=1, ncv=0, maxit=600).E0
            assert abs(ea - em) <= 1e-10 * max(abs(em), 1.0), ("E0", ea, em)
            A.destroy()
            M.destroy()
        except Exception as e:      # noqa: BLE001
            fails.append((tag, repr(e)[:300]))
            print("FAIL", tag, "::", repr(e)[:300], flush=True)
        done += 1
    print("fuzz_midsize: %d cases (%d split in place, largest %.2e nonzeros), %d failures, %.0f s (seed %d)" % (done, n_split, nnz_max, len(fails), time.time() - t0, seed))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
