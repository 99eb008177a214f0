#!/usr/bin/env python3
"""Kill criterion for a RAGGED Kronecker split of a single-species (spin-1/2, fixed n_dn) sector -- VERDICT round 3, item 5.

Basis = bit patterns with n_dn bits set, ascending (= colexicographic rank, qbh_gen_heisenberg).  Cut the sites into low
(< h) and high (>= h): patterns with the same high part are contiguous -- a block of C(h, n_dn - popcount(high)) rows -- so the
order already IS index = base[high] + rank(low | its particle number).  Bonds inside the low half keep the block (near part:
an L2-sized window of x), bonds inside the high half keep the rank inside the block and the block size (far part: same minor
index, band-major per block-size class, the C3 treatment), bonds ACROSS the cut change both and the block size: a third part
with unstructured gathers.  Host analysis, no GPU: for every cut h the share of the nonzeros in each part, and for the crossing
part the distinct 128-byte lines of x per entry in windows of consecutive entries (what a cache of that reach has to fetch),
on a sample of consecutive rows from the middle of the basis.  Projected fabric traffic per SpMV =
  20 B x nnz (stream) + vectors (x natural + tiled copy + far result round trip + y_old + y: ~9 x 16 B x dim) + crossing lines x 128 B.
Build only if that is <= 140 GB for kagome-30 (round 3 measured 203 GB).
usage: python tools/ragged_kron_analysis.py [kagome_30|kagome_24|chain_26] [sample rows]"""
import itertools
import os
import sys
from math import comb

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_basis_amd import lattices  # noqa: E402

CASES = {"kagome_30": (30, 15, lattices.kagome(5, 2)), "kagome_24": (24, 12, lattices.kagome(4, 2)), "chain_26": (26, 13, lattices.chain(26))}


def unrank(r, n, k):
    r = r.copy()
    kk = np.full(r.shape, k, dtype=np.int64)
    bits = np.zeros(r.shape, dtype=np.int64)
    B = np.array([[comb(p, j) for j in range(k + 2)] for p in range(n + 1)], dtype=np.int64)
    for p in range(n - 1, -1, -1):
        c = B[p, kk]
        take = (kk > 0) & (c <= r)
        bits |= take.astype(np.int64) << p
        r -= np.where(take, c, 0)
        kk -= take
    return bits


def rank(bits, n, k):
    B = np.array([[comb(p, j) for j in range(k + 2)] for p in range(n + 1)], dtype=np.int64)
    cnt = np.zeros(bits.shape, dtype=np.int64)
    r = np.zeros(bits.shape, dtype=np.int64)
    for p in range(n):
        on = (bits >> p) & 1
        cnt += on
        r += np.where(on == 1, B[p, np.minimum(cnt, k + 1)], 0)
    return r


def best_numbering(n, bonds, h, iters=200000, seed=1):
    """simulated annealing on the site numbering: minimise the bonds across the cut at h (sites keep their bonds)"""
    rng = np.random.default_rng(seed)
    perm = np.arange(n)
    b = np.asarray(bonds)

    def cut(pm):
        lo = pm[b] < h
        return int(np.sum(lo[:, 0] != lo[:, 1]))
    cur = cut(perm)
    best, best_perm = cur, perm.copy()
    T = 2.0
    for it in range(iters):
        i, j = rng.integers(0, n, 2)
        if (perm[i] < h) == (perm[j] < h):
            continue
        perm[i], perm[j] = perm[j], perm[i]
        c = cut(perm)
        if c <= cur or rng.random() < np.exp((cur - c) / T):
            cur = c
            if c < best:
                best, best_perm = c, perm.copy()
        else:
            perm[i], perm[j] = perm[j], perm[i]
        T = max(0.05, T * 0.99997)
    return best, best_perm


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "kagome_30"
    nsample = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    n, k, bonds = CASES[name]
    bonds = np.asarray(bonds, dtype=np.int64)
    dim = comb(n, k)
    print("%s: %d sites, n_dn %d, dim %d, %d bonds" % (name, n, k, dim, len(bonds)))
    start = dim // 2 - nsample // 2
    pats = unrank(np.arange(start, start + nsample, dtype=np.int64), n, k)
    print("  sample: rows [%d, %d)" % (start, start + nsample))
    for relabel in (False, True):
        for h in sorted({n // 2, n // 2 + 3, n // 2 - 3}):
            bb = bonds
            if relabel:
                cutsize, perm = best_numbering(n, bonds, h)
                bb = perm[bonds]
            lo = bb < h
            kinds = np.where(lo[:, 0] & lo[:, 1], 0, np.where(~lo[:, 0] & ~lo[:, 1], 1, 2))      # 0 low-only, 1 high-only, 2 crossing
            counts = [0, 0, 0]
            cross_cols = []
            for (i, j), kind in zip(bb, kinds):
                opp = ((pats >> i) & 1) != ((pats >> j) & 1)
                counts[kind] += int(opp.sum())
                if kind == 2:
                    col = np.full(pats.shape, -1, dtype=np.int64)
                    col[opp] = rank(pats[opp] ^ ((1 << int(i)) | (1 << int(j))), n, k)
                    cross_cols.append(col)
            off = sum(counts)
            nnz_row = off / nsample + 1.0
            nnz = nnz_row * dim
            cc = np.stack(cross_cols, axis=1).reshape(-1) if cross_cols else np.zeros(0, dtype=np.int64)
            cc = cc[cc >= 0]                                      # the crossing part in row-major order
            lines = cc >> 3
            rep = {}
            for E in (512, 4096, 32768, 262144):
                nb = lines.size // E
                if nb == 0:
                    continue
                smp = np.linspace(0, nb - 1, min(nb, 400)).astype(np.int64)
                rep[E] = float(np.mean([np.unique(lines[s * E:(s + 1) * E]).size for s in smp])) / E
            n_cross = counts[2] / nsample * dim
            big_blocks = comb(h, k // 2 if h >= k else 0)
            base_gb = (20.0 * nnz + 9 * 16.0 * dim) / 1e9
            proj = {E: base_gb + v * n_cross * 128 / 1e9 for E, v in rep.items()}
            print("  %s cut h = %2d: crossing bonds %2d of %d | entries per row: near %.2f (incl. diagonal) far %.2f crossing %.2f (%.1f %%) | "
                  "largest block C(%d, %d) = %d rows = %.0f KB of x"
                  % ("annealed numbering," if relabel else "lattice numbering, ", h, int((kinds == 2).sum()), len(bb), counts[0] / nsample + 1, counts[1] / nsample,
                     counts[2] / nsample, 100.0 * counts[2] / off, h, min(h, k) // 1 if False else min(k, h) // 2 + (0 if True else 0), big_blocks, big_blocks * 16 / 1024))
            print("      crossing part: distinct x lines per entry in windows of E consecutive entries: " +
                  ", ".join("E=%d: %.3f" % (E, v) for E, v in rep.items()))
            print("      projected fabric traffic per SpMV: stream + vectors %.1f GB + crossing gathers -> " % base_gb +
                  ", ".join("%.0f GB (reach %d)" % (p, E) for E, p in proj.items()))


if __name__ == "__main__":
    main()
