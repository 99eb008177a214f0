#!/usr/bin/env python3
"""Randomised campaign over the device generators and the stored forms on one GPU (round 6, outside the GPU tier): Fermi-Hubbard and
Heisenberg operators on RANDOM bond graphs (4..10 sites, repeated bonds allowed), random fillings, random options (split in place or not,
far part sliced / padded / row-major, 2-byte or int32 columns, value codes, real fast path, static or dynamic walks, the up
configurations in a partition order) against an independent numpy / scipy assembly (tests/fastham.py): the stored entries (through the
major-index map where the order is partitioned), device SpMVs with random (alpha, beta, gamma) and their fused reductions, MultMv, and
the Lanczos ground-state energy against dense diagonalisation.
usage: python tools/r6/fuzz_gen.py [cases=200] [seed=1] [edge=0]"""
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


def main():
    kv = dict(a.split("=", 1) for a in sys.argv[1:])
    cases, seed = int(kv.get("cases", 200)), int(kv.get("seed", 1))
    edge = int(kv.get("edge", 0))          # 1: degenerate shapes too -- 2..10 sites, EMPTY or FULL species (one configuration: minor or major size 1), dim down to 1
    rng = np.random.default_rng(seed)
    fails, done, t0, n_split, n_part, n_heis = [], 0, time.time(), 0, 0, 0
    while done < cases:
        n = int(rng.integers(2 if edge else 4, 11))
        nb = int(rng.integers(max(1, n - 1), 2 * n + 1))
        bonds = []
        while len(bonds) < nb:
            a, b = int(rng.integers(n)), int(rng.integers(n))
            if a != b:
                bonds.append((a, b))
        heis = int(rng.integers(4)) == 0
        o = dict(kron_split=int(rng.choice([0, 2, 2])), kron_sliced=int(rng.integers(3)), kron_cols16=int(rng.integers(2)), value_dict=int(rng.choice([0, 0, 1])),
                 real_fast_path=int(rng.choice([0, 0, 1])), deterministic=int(rng.integers(2)))
        try:
            if heis:
                nd = int(rng.integers(0, n + 1)) if edge else int(rng.integers(1, n))
                if math.comb(n, nd) > 20000:
                    continue
                J = float(rng.choice([1.0, -0.7, 2.5]))
                tag = "heisenberg n %d nd %d J %g bonds %s %s" % (n, nd, J, bonds, o)
                H = fastham.heisenberg_full(n, nd, bonds, J=J)
                A = q.csr_mat.heisenberg(n, nd, bonds, J=J, opts=q.make_opts(sector_cut=int(rng.choice([0, -1])), **o))
                n_heis += 1
                perm = None
            else:
                nu, nd = (int(rng.integers(0, n + 1)), int(rng.integers(0, n + 1))) if edge else (int(rng.integers(1, n)), int(rng.integers(1, n)))
                NU, S = math.comb(n, nu), math.comb(n, nd)
                if NU * S > 20000:
                    continue
                t, U = float(rng.choice([1.0, 0.6])), float(rng.choice([0.0, 1.1, 4.0]))
                parts = int(rng.choice([0, 0, 2, 3, 5]))
                if parts > NU:
                    parts = 0
                tag = "hubbard n %d nu %d nd %d t %g U %g parts %d bonds %s %s" % (n, nu, nd, t, U, parts, bonds, o)
                H = fastham.hubbard_full(n, nu, nd, bonds, t=t, U=U)
                A = q.csr_mat.hubbard(n, nu, nd, bonds, t=t, U=U, opts=q.make_opts(major_partition=parts, **o))
                perm = None
                if A.info().major_partition > 1:
                    mo = A.major_order(NU).astype(np.int64)               # operator's major index i holds the generator's major mo[i]
                    perm = (mo[:, None] * S + np.arange(S)[None, :]).ravel()
                    n_part += 1
            info = A.info()
            dim = A.dim
            assert dim == H.shape[0], "dim"
            n_split += int(info.kron_minor > 0)
            Hp = H if perm is None else H[perm][:, perm]
            Hp = sp.csr_matrix(Hp)
            Hp.sort_indices()
            if info.basis_internal == 0:                                 # (a cut Heisenberg sector holds class-major rows: compared through the seams below)
                ia, ja, val = A.download()
                Hd = sp.csr_matrix((val, ja, ia), shape=(dim, dim))
                D = (Hd - Hp).tocoo()
                assert D.nnz == 0 or np.abs(D.data).max() <= 1e-14, ("stored entries", np.abs(D.data).max())
                assert len(ja) >= Hp.nnz, "explicit diagonal"
            x = (rng.normal(size=dim) + 1j * rng.normal(size=dim)).astype(np.complex128)
            y0 = (rng.normal(size=dim) + 1j * rng.normal(size=dim)).astype(np.complex128)
            Hs = Hp if info.basis_internal == 0 else sp.csr_matrix(H)    # host vectors are always the caller's (generator) order
            want = Hs @ x
            scale = max(np.abs(want).max(), 1e-300)
            y = np.empty(dim, dtype=np.complex128)
            A.MultMv(x, y)
            assert np.abs(y - want).max() <= 4e-13 * scale, ("MultMv", np.abs(y - want).max() / scale)
            if info.basis_internal == 0:
                alpha, beta, gamma = float(rng.normal()), float(rng.choice([0.0, 1.0, -0.3])), float(rng.choice([0.0, 0.25]))
                v = A.vec(2)
                v.upload(x, 0)
                v.upload(y0, dim)
                xy, yy = A.spmv(v.at(0), v.at(dim), alpha, beta, gamma, want_red=True)
                yd = v.download(dim, dim)
                v.free()
                ref = alpha * want + beta * y0 + gamma * x
                assert np.abs(yd - ref).max() <= 4e-13 * max(np.abs(ref).max(), scale), "device SpMV"
                assert abs(xy - np.vdot(x, ref)) <= 1e-11 * max(abs(np.vdot(x, ref)), 1.0) and abs(yy - np.vdot(ref, ref).real) <= 1e-11 * max(np.vdot(ref, ref).real, 1.0), "fused reductions"
            if dim >= 40:
                e0 = float(np.linalg.eigvalsh(H.toarray())[0]) if dim <= 2500 else float(sp.linalg.eigsh(sp.csr_matrix(H), k=1, which="SA", tol=1e-13)[0][0])
                r = q.locate_E0_lanczos(A, nev=1, ncv=0, maxit=1000)
                assert abs(r.E0 - e0) <= 1e-9 * max(abs(e0), 1.0), ("E0", r.E0, e0)
            A.destroy()
        except Exception as e:      # noqa: BLE001
            fails.append((tag, repr(e)[:300]))
            print("FAIL", tag, "::", repr(e)[:300], flush=True)
        done += 1
    print("fuzz_gen: %d cases (%d Heisenberg, %d split in place, %d in a partition order), %d failures, %.0f s (seed %d)" % (done, n_heis, n_split, n_part, len(fails), time.time() - t0, seed))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
