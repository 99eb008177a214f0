#!/usr/bin/env python3
"""Balance the row shards of a Hubbard momentum sector BY MEASURED TIME, rank by rank on one GPU: round 0 builds every shard on
the uniform cuts and times its SpMV against a full-length x; every further round rebuilds them on the cuts
quantum_basis_amd.dist.rebalance_cuts derives from the previous round's times (qbh_gen_hubbard_repr_cuts).

    python tools/sector_balance.py 4 5 10 10 4 [rounds=2]      # BASELINE configs[3]: 4x5 at half filling, 4 ranks
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import quantum_basis_amd as q  # noqa: E402
from quantum_basis_amd import dist, lattices  # noqa: E402


def time_shard(n, nu, nd, bonds, perms, chars, rank, world, cuts):
    t0 = time.time()
    A = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, chars, t=1.0, U=1.1, shard=(rank, world), opts=q.make_opts(profile=1),
                               row_cuts=cuts)
    i = A.info()
    x = q.DeviceVec(A, i.ncols)
    y = A.vec()
    off, seed = 0, 1
    while off < i.ncols:
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
    for _ in range(4):
        A.spmv(x.at(0), y.ptr)
    A.sync()
    s = A.stats()
    ms = s.ms_spmv / max(1, s.n_spmv)
    out = dict(rows=(int(i.row_offset), int(i.row_offset + i.nrows)), nnz=int(i.nnz), ms=ms, dim=int(i.ncols), build_s=time.time() - t0)
    x.free()
    y.free()
    A.destroy()
    return out


def main():
    Lx, Ly, nu, nd, world = (int(a) for a in sys.argv[1:6])
    rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 2
    n = Lx * Ly
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    chars = lattices.characters(shifts, (0, 0), (Lx, Ly))
    cuts = None
    for rnd in range(rounds + 1):
        res = [time_shard(n, nu, nd, bonds, perms, chars, r, world, cuts) for r in range(world)]
        ms = np.array([r["ms"] for r in res])
        cur = np.array([res[0]["rows"][0]] + [r["rows"][1] for r in res], dtype=np.int64)
        print("round %d (%s cuts): rows per rank %s" % (rnd, "uniform" if cuts is None else "rebalanced", list(np.diff(cur))))
        print("   nnz per rank   %s" % [r["nnz"] for r in res])
        print("   ms per SpMV    %s   max/min %.3f, spread around the mean +%.1f %% / -%.1f %%" %
              (["%.2f" % m for m in ms], ms.max() / ms.min(), 100 * (ms.max() / ms.mean() - 1), 100 * (1 - ms.min() / ms.mean())), flush=True)
        cuts = dist.rebalance_cuts(cur, ms)


if __name__ == "__main__":
    main()
