#!/usr/bin/env python3
"""How many DISTINCT 128-byte lines of x does a stretch of consecutive CSR entries touch?  (measurement tool for the
'column-segmented storage per row block' idea: sorting a block's entries by column can save at most the repeats INSIDE the block.)
Builds one row shard of a benchmark operator on the device, downloads its columns and counts distinct lines (column >> 3 for
complex128 x) per block of E consecutive entries.  Usage: python tools/line_reuse.py <workload> [world] [rank]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import quantum_basis_amd as q  # noqa: E402
from quantum_basis_amd import lattices  # noqa: E402


def main():
    name = sys.argv[1]
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    rank = int(sys.argv[3]) if len(sys.argv) > 3 else world // 2
    W = bench.workloads()[name]
    opts = q.make_opts(value_dict=0, real_fast_path=0)
    if W["kind"] == "heisenberg_repr":
        perms, shifts = lattices.translations(*W["trans"])
        chars = lattices.characters(shifts, W["k"], W["trans"])
        A = q.csr_mat.heisenberg_repr(W["n_sites"], W["n_dn"], W["bonds"], perms, chars, J=W["J"], shard=(rank, world), opts=opts)
    elif W["kind"] == "heisenberg":
        dim = bench.dim_of(W)
        A = bench.build_operator(W, (rank * dim // world, (rank + 1) * dim // world), opts)
    else:
        dim = bench.dim_of(W)
        A = bench.build_operator(W, (rank * dim // world, (rank + 1) * dim // world), opts)
    info = A.info()
    ia, ja, _ = A.download()
    ja = np.asarray(ja, dtype=np.int64)
    nnz = ja.size
    print("%s shard %d/%d: rows %d of %d, nnz %d (%.1f per row)" % (name, rank, world, info.nrows, info.ncols, nnz, nnz / max(1, info.nrows)))
    lines = ja >> 3
    print("  distinct x lines in the whole shard: %d = %.3f per entry (x bytes if each were fetched once: %.2f x the shard's 16 B/row)"
          % (np.unique(lines).size, np.unique(lines).size / nnz, np.unique(lines).size * 128 / (info.nrows * 16.0)))
    for E in (64, 512, 4096, 32768, 262144, 2097152):
        nb = nnz // E
        if nb == 0:
            break
        sample = np.linspace(0, nb - 1, min(nb, 2000)).astype(np.int64)
        d = np.array([np.unique(lines[b * E:(b + 1) * E]).size for b in sample], dtype=np.float64)
        print("  blocks of %8d consecutive entries: %.3f distinct lines per entry (min %.3f, max %.3f)" % (E, d.mean() / E, d.min() / E, d.max() / E))
    A.destroy()


if __name__ == "__main__":
    main()
