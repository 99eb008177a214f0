#!/usr/bin/env python3
"""Randomised campaign over the HOST-ARRAY seam on one GPU (round 6, outside the GPU tier): Fermi-Hubbard operators in the REFERENCE's
own order (Lin order, site-ordered operators, Hermitian-upper int64 CSR: what csr_mat(lil_mat&) hands over, src/sparse.cc:202-260),
real or gauge-transformed to complex values, through qbh_csr_create with random options (basis named / found by the library / neither,
forced split, 2-byte or int32 columns, value codes, real fast path, static or dynamic walks) against the oracle ON THE SAME ARRAYS:
MultMv, MultMv2, a device SpMV with (alpha, beta, gamma) and its fused reductions, the download of the caller's rows, and the Lanczos
ground-state energy against dense diagonalisation.
usage: python tools/r6/fuzz_hostcsr.py [cases=60] [seed=1]"""
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
from oracle import qb_oracle as qo  # noqa: E402
import refham  # noqa: E402


def main():
    kv = dict(a.split("=", 1) for a in sys.argv[1:])
    cases, seed = int(kv.get("cases", 60)), int(kv.get("seed", 1))
    rng = np.random.default_rng(seed)
    fails, done, t0, n_split, n_internal, n_decoy = [], 0, time.time(), 0, 0, 0
    while done < cases:
        lx, ly = [(2, 2), (3, 2), (4, 2), (3, 3)][rng.integers(4)]
        n = lx * ly
        nu, nd = int(rng.integers(1, n)), int(rng.integers(1, n))
        dim0 = math.comb(n, nu) * math.comb(n, nd)
        if dim0 > 12000 or dim0 < 4:
            continue
        U = float(rng.choice([0.0, 1.1, 4.0]))
        dim, ia, ja, val, _ = refham.hubbard_csr(lx, ly, nu, nd, t=1.0, U=U)
        decoy = int(rng.integers(4)) == 0                # a matrix of the same (colliding) dimension WITHOUT the product structure: must stay as given
        if decoy:
            import scipy.sparse as sp
            R = sp.random(dim, dim, density=min(0.5, 8.0 / dim), random_state=int(rng.integers(1 << 30)), format="csr", dtype=np.float64)
            R = R + 1j * sp.random(dim, dim, density=min(0.5, 4.0 / dim), random_state=int(rng.integers(1 << 30)), format="csr", dtype=np.float64)
            Hd = sp.csr_matrix(R + R.conj().T + sp.diags(rng.normal(size=dim)))
            Hd = sp.csr_matrix(sp.triu(Hd) + sp.diags(np.full(dim, 1e-300)))          # the reference stores every diagonal entry
            Hd.sort_indices()
            ia, ja, val = Hd.indptr.astype(np.int64), Hd.indices.astype(np.int64), Hd.data.astype(np.complex128)
            val[np.repeat(np.arange(dim), np.diff(ia)) == ja] = val[np.repeat(np.arange(dim), np.diff(ia)) == ja].real
            n_decoy += 1
        gauge = int(rng.integers(2))
        if gauge:
            ph = np.exp(2j * np.pi * rng.random(dim))
            rows = np.repeat(np.arange(dim), np.diff(ia))
            val = val * ph[rows] * np.conj(ph[ja])
            val[rows == ja] = val[rows == ja].real
        val = np.ascontiguousarray(val, dtype=np.complex128)
        O = qo.Csr(dim, ia, ja, val, True)
        hint = int(rng.choice([0, 0, 1, 1, 2]))                      # 0 nothing said, 1 basis named, 2 detection off and nothing named
        o = dict(kron_split=int(rng.choice([0, 2, 2, 2])), value_dict=int(rng.integers(2)), real_fast_path=int(rng.integers(2)), kron_cols16=int(rng.integers(2)),
                 deterministic=int(rng.integers(2)), basis_detect=0 if hint == 2 else 1)
        if hint == 1:
            o.update(basis_kind=_lib.BASIS_REF_FERMION2, n_sites=n, n_up=nu, n_dn=nd)
        tag = "%dx%d nu %d nd %d U %g gauge %d decoy %d %s" % (lx, ly, nu, nd, U, gauge, int(decoy), o)
        try:
            A = q.csr_mat(dim, ia, ja, val, sym=True, opts=q.make_opts(**o))
            info = A.info()
            if decoy:
                assert info.kron_minor == 0 and info.basis_internal == 0, ("a matrix without the structure was permuted / split", info.kron_minor, info.basis_internal)
            n_split += int(info.kron_minor > 0)
            n_internal += int(info.basis_internal != 0)
            x = (rng.normal(size=dim) + 1j * rng.normal(size=dim)).astype(np.complex128)
            y0 = (rng.normal(size=dim) + 1j * rng.normal(size=dim)).astype(np.complex128)
            want = O.multmv(x)
            scale = max(np.abs(want).max(), 1e-300)
            y = np.empty(dim, dtype=np.complex128)
            A.MultMv(x, y)
            assert np.abs(y - want).max() <= 4e-13 * scale, ("MultMv", np.abs(y - want).max() / scale)
            y = y0.copy()
            A.MultMv2(x, y)
            assert np.abs(y - (y0 + want)).max() <= 4e-13 * max(scale, np.abs(y0).max()), "MultMv2"
            # the caller's rows back, both triangles
            fia, fja, fval = A.download()
            rows = np.repeat(np.arange(dim), np.diff(ia))
            off = ja > rows
            er, ec, ev = np.concatenate([rows, ja[off]]), np.concatenate([ja, rows[off]]), np.concatenate([val, np.conj(val[off])])
            order = np.lexsort((ec, er))
            assert np.array_equal(fja, ec[order]) and np.array_equal(fval, ev[order]), "download"
            # Lanczos from the reference's start vector: E0 against dense diagonalisation
            if dim >= 40:
                import scipy.sparse as sp
                H = sp.csr_matrix((fval, fja, fia), shape=(dim, dim))
                e0 = float(np.linalg.eigvalsh(H.toarray())[0]) if dim <= 2500 else float(sp.linalg.eigsh(H, k=1, which="SA", tol=1e-13)[0][0])
                r = q.locate_E0_lanczos(A, nev=1, ncv=1, maxit=1000)
                assert abs(r.E0 - e0) <= 1e-9 * max(abs(e0), 1.0), ("E0", r.E0, e0)
                assert np.abs(O.multmv(r.eigenvecs) - r.E0 * r.eigenvecs).max() < 1e-7, "eigenvector residual"
            A.destroy()
        except Exception as e:      # noqa: BLE001
            fails.append((tag, repr(e)[:300]))
            print("FAIL", tag, "::", repr(e)[:300], flush=True)
        done += 1
    print("fuzz_hostcsr: %d cases (%d split in place, %d held in an internal order, %d decoys of a colliding dimension), %d failures, %.0f s (seed %d)" % (done, n_split, n_internal, n_decoy, len(fails), time.time() - t0, seed))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
