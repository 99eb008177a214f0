#!/usr/bin/env python3
"""Parity of the matrix-free sector operator at a size where the whole stored operator does not fit: one row shard of the
stored sector CSR (qbh_gen_hubbard_repr, shard = (rank, world)) and the matrix-free operator (qbh_mf_hubbard_repr) are built
side by side, applied to the same full-length random vector, and compared on the shard's rows.

    python tools/sector_mf_vs_shard.py 4 5 10 10 0 4      # BASELINE configs[3]: rows of shard 0 of 4, 4.3e8 of 1.7e9 rows
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import quantum_basis_amd as q  # noqa: E402
from quantum_basis_amd import lattices  # noqa: E402


def main():
    Lx, Ly, nu, nd, rank, world = (int(a) for a in sys.argv[1:7])
    k = (int(sys.argv[7]), int(sys.argv[8])) if len(sys.argv) > 8 else (0, 0)
    n = Lx * Ly
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    chars = lattices.characters(shifts, k, (Lx, Ly))
    t0 = time.time()
    M = q.csr_mat.hubbard_repr_mf(n, nu, nd, bonds, perms, chars, t=1.0, U=1.1)
    t1 = time.time()
    S = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, chars, t=1.0, U=1.1, shard=(rank, world))
    t2 = time.time()
    i = S.info()
    dim = M.info().ncols
    assert i.ncols == dim
    print("sector k=%s dim %d; matrix-free operator built in %.1f s; stored shard rows [%d, %d), nnz %d, built in %.1f s" %
          (k, dim, t1 - t0, i.row_offset, i.row_offset + i.nrows, i.nnz, t2 - t1), flush=True)
    w = M.vec(2)                                           # x and y of the matrix-free operator, in ITS row order (orbit by orbit)
    M.randomize(w.at(0), 11)
    M.spmv(w.at(0), w.at(dim))
    v = q.engine.DeviceVec(S, 2 * dim)                     # the same two vectors in the caller's order, owned by the stored shard's handle
    M.from_internal(v.at(0), w.at(0))
    M.from_internal(v.at(dim), w.at(dim))
    M.sync()
    w.free()
    ys = S.vec()
    S.spmv(v.at(0), ys.ptr)                                # unsharded convention: x is the full vector
    S.sync()
    # compare on the shard's rows, block by block through the host
    worst, scale = 0.0, 0.0
    step = 1 << 24
    for off in range(0, i.nrows, step):
        cnt = min(step, i.nrows - off)
        a = ys.download(off, cnt)
        b = v.download(dim + i.row_offset + off, cnt)
        worst = max(worst, float(np.abs(a - b).max()))
        scale = max(scale, float(np.abs(a).max()))
    print("max |y_shard - y_mf| over %d rows = %.3e (max |y| = %.3e)" % (i.nrows, worst, scale))
    ok = worst <= 1e-12 * scale
    print("PARITY", "OK" if ok else "FAILED")
    v.free()
    ys.free()
    S.destroy()
    M.destroy()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
