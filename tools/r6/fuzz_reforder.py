#!/usr/bin/env python3
"""Randomised campaign over the reference-order pass and the sharded sector generators on one GPU (round 6, outside the GPU tier):
  * qbh_csr_reference_order: a device-generated Hubbard operator (random square lattice, fillings, U) permuted
    on the device into the REFERENCE's Lin order and fermion convention must equal, entry by entry, the numpy re-derivation of the
    reference's host pipeline (tests/refham.py) -- and handing THAT back through qbh_csr_create with the basis found by the library must
    reproduce the same MultMv;
  * qbh_gen_heisenberg_repr row shards: the shards (r, P) of a momentum sector, P = 2..5, stacked, are the whole sector's rows.
usage: python tools/r6/fuzz_reforder.py [cases=100] [seed=1]"""
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
import refham  # noqa: E402
import reprham  # noqa: E402
import test_gpu_reforder as T  # noqa: E402


def main():
    kv = dict(a.split("=", 1) for a in sys.argv[1:])
    cases, seed = int(kv.get("cases", 100)), int(kv.get("seed", 1))
    rng = np.random.default_rng(seed)
    fails, done, t0, cnt = [], 0, time.time(), {"reforder": 0, "repr_shards": 0}
    while done < cases:
        tag = ""
        try:
            if int(rng.integers(3)) < 2:
                lx, ly = [(2, 2), (3, 2), (4, 2), (3, 3), (5, 2)][int(rng.integers(5))]
                n = lx * ly
                nu, nd = int(rng.integers(1, n)), int(rng.integers(1, n))
                if math.comb(n, nu) * math.comb(n, nd) > 20000:
                    continue
                U = float(rng.choice([0.0, 1.1, 4.0]))
                o = dict(value_dict=0, real_fast_path=0, kron_split=0)            # (the pass takes an unsplit complex128 source and says so otherwise)
                tag = "reforder %dx%d nu %d nd %d U %g %s" % (lx, ly, nu, nd, U, o)
                G = q.csr_mat.hubbard(n, nu, nd, lattices.square(lx, ly), t=1.0, U=U, opts=q.make_opts(**o))
                R = G.reference_order(1, n, nu, nd)
                dim, ia, ja, val, _ = refham.hubbard_csr(lx, ly, nu, nd, t=1.0, U=U)
                F = T._full(dim, ia, ja, val)
                ria, rja, rval = R.download()
                assert R.dim == dim and np.array_equal(ria, F.indptr) and np.array_equal(rja, F.indices), "pattern"
                assert np.abs(rval - F.data).max() < 1e-13, ("values", np.abs(rval - F.data).max())
                # ... and back through the host seam: the library finds the basis by itself and gives the same product
                B = q.csr_mat(dim, ia, ja, val, sym=True, opts=q.make_opts(value_dict=0, real_fast_path=0, kron_split=2))
                x = (rng.normal(size=dim) + 1j * rng.normal(size=dim)).astype(np.complex128)
                yr, yb = np.empty(dim, dtype=np.complex128), np.empty(dim, dtype=np.complex128)
                R.MultMv(x, yr)
                B.MultMv(x, yb)
                assert np.abs(yr - yb).max() <= 4e-13 * max(np.abs(yr).max(), 1e-300), "host seam"
                for A in (G, R, B):
                    A.destroy()
                cnt["reforder"] += 1
            else:
                lx, ly = [(3, 2), (4, 2), (3, 3), (4, 3), (6, 1), (8, 1), (10, 1), (12, 1)][int(rng.integers(8))]
                n = lx * ly
                nd = int(rng.integers(1, n))
                if math.comb(n, nd) > 5000:
                    continue
                k = (int(rng.integers(lx)), int(rng.integers(ly)))
                bonds = [(x + lx * y, (x + 1) % lx + lx * y) for x in range(lx) for y in range(ly)] + ([(x + lx * y, x + lx * ((y + 1) % ly)) for x in range(lx) for y in range(ly)] if ly > 1 else [])
                bonds = [b for b in bonds if b[0] != b[1]]
                P = int(rng.integers(2, 6))
                tag = "repr shards %dx%d nd %d k %s P %d" % (lx, ly, nd, k, P)
                perms, shifts = reprham.translations_2d(lx, ly)
                chars = reprham.characters(shifts, k, (lx, ly))
                W = q.csr_mat.heisenberg_repr(n, nd, bonds, perms, chars, opts=q.make_opts(value_dict=0))
                wia, wja, wval = W.download()
                dim = W.dim
                if dim < P * P:                  # (uniform blocks of ceil(dim / P) rows: a smaller sector leaves the last rank without rows, which the generator refuses loudly)
                    W.destroy()
                    continue
                r_at = 0
                for r in range(P):
                    S = q.csr_mat.heisenberg_repr(n, nd, bonds, perms, chars, shard=(r, P), opts=q.make_opts(value_dict=int(rng.integers(2))))
                    i = S.info()
                    assert i.ncols == dim and i.row_offset == r_at, ("shard rows", r, i.row_offset, r_at)
                    sia, sja, sval = S.download()
                    a, b = wia[r_at], wia[r_at + i.nrows]
                    assert np.array_equal(sia, wia[r_at:r_at + i.nrows + 1] - a) and np.array_equal(sja, wja[a:b]) and np.abs(sval - wval[a:b]).max() <= 1e-14, ("shard entries", r)
                    r_at += int(i.nrows)
                    S.destroy()
                assert r_at == dim, "shards do not tile the rows"
                W.destroy()
                cnt["repr_shards"] += 1
        except Exception as e:      # noqa: BLE001
            fails.append((tag, repr(e)[:300]))
            print("FAIL", tag, "::", repr(e)[:300], flush=True)
        done += 1
    print("fuzz_reforder: %d cases %s, %d failures, %.0f s (seed %d)" % (done, cnt, len(fails), time.time() - t0, seed))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
