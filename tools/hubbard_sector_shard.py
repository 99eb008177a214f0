#!/usr/bin/env python3
"""One row shard of a Hubbard momentum sector on one GPU: build it (qbh_gen_hubbard_repr with shard = (rank, world)) and time
its SpMV against a full-length x -- the per-rank compute of a row-sharded run, measured without the other ranks.

    python tools/hubbard_sector_shard.py 4 5 10 10 0 4      # BASELINE configs[3]: 4x5 at half filling, rank 0 of 4

4x5 at half filling has 3.4e10 basis states (no stored operator fits any node); its k = (0,0) sector has 1.7e9
representatives and ~7e10 nonzeros, i.e. ~105 GB per rank as column indices + 2-byte value codes on 4 GPUs."""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
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
    A = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, chars, t=1.0, U=1.1, shard=(rank, world), opts=q.make_opts(profile=1))
    t1 = time.time()
    i = A.info()
    print("sector k=%s of %dx%d (%d up, %d down): dim %d, this shard rows [%d, %d) = %d rows, nnz %d, value_dict %d" %
          (k, Lx, Ly, nu, nd, i.ncols, i.row_offset, i.row_offset + i.nrows, i.nrows, i.nnz, i.value_dict), flush=True)
    print("built in %.1f s; device bytes of the operator %.2f GB (%.2f B/nnz)" % (t1 - t0, i.bytes_matrix * 1e-9, i.bytes_matrix / max(1, i.nnz)),
          flush=True)
    x = q.DeviceVec(A, i.ncols)
    y = A.vec()
    off = 0
    seed = 1
    while off < i.ncols:                                   # fill the full-length x block by block
        if off + i.nrows <= i.ncols:
            A.randomize(x.at(off), seed)
            off += i.nrows
        else:
            A.randomize(x.at(i.ncols - i.nrows), seed)
            off = i.ncols
        seed += 1
    for _ in range(2):
        A.spmv(x.at(0), y.ptr)
    A.stats(reset=True)
    reps = 5
    for _ in range(reps):
        A.spmv(x.at(0), y.ptr)
    A.sync()
    s = A.stats()
    ms = s.ms_spmv / max(1, s.n_spmv)
    print("SpMV of the shard against the full x (complex128 vectors): %.2f ms per launch, %.1f G nonzeros/s, %.0f GB/s of format bytes" %
          (ms, i.nnz / ms * 1e-6, i.bytes_matrix / ms * 1e-6), flush=True)
    print("|y| = %.12e" % A.nrm2(y.ptr))
    x.free()
    y.free()
    A.destroy()


if __name__ == "__main__":
    main()
