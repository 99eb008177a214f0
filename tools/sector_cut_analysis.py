#!/usr/bin/env python3
"""Kill criterion for a site-cut (ragged Kronecker) split INSIDE a momentum sector -- VERDICT round 3, item 6 (C5: triangular 6x6,
Sz = 0, k = (1, 0): 776 GB per SpMV = 2.6 x algorithmic, fabric-saturated at 6.25 TB/s).

In a translation-symmetric sector the basis is the orbit REPRESENTATIVES (smallest bit pattern of each orbit, ascending) and a
bond flip is followed by re-canonicalisation: the flipped pattern is translated to ITS smallest form (src/model.cc:687-836).  A
site cut gives the entry a block structure only when that translation is the identity -- the flipped pattern is already the
representative -- AND the bond does not cross the cut.  Host analysis on a random sample of representatives: the share of the
off-diagonal entries that (a) need no translation, (b) need none and stay inside the low / the high half of a cut at h sites.
With s = the structured share, the gather traffic of the rest stays at today's 0.26 lines per entry (profiles/r3_lab/
line_reuse.txt): traffic >= 295 GB + (1 - s) x 481 GB; <= 500 GB needs s >= 0.57.
usage: python tools/sector_cut_analysis.py [samples]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_basis_amd import lattices  # noqa: E402


def main():
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    n, k = 36, 18
    bonds = np.asarray(lattices.triangular(6, 6), dtype=np.int64)
    perms, shifts = lattices.translations(6, 6)
    perms = np.asarray(perms, dtype=np.int64)                     # perms[g][site] = image
    rng = np.random.default_rng(7)

    def translate(p, g):
        out = np.zeros_like(p)
        for s in range(n):
            out |= ((p >> s) & 1) << int(perms[g][s])
        return out

    def canon(p):
        best = p.copy()
        which = np.zeros(p.shape, dtype=np.int64)
        for g in range(1, len(perms)):
            t = translate(p, g)
            better = t < best
            best = np.where(better, t, best)
            which = np.where(better, g, which)
        return best, which

    # random patterns with k bits -> their representatives (orbits of size 36 dominate: a fair sample of the basis)
    pats = np.zeros(ns, dtype=np.int64)
    for i in range(ns):
        pats[i] = np.sum(1 << rng.choice(n, k, replace=False).astype(np.int64))
    reps, _ = canon(pats)
    reps = np.unique(reps)
    print("triangular 6x6, Sz = 0: %d sampled representatives, %d bonds, %d translations" % (reps.size, len(bonds), len(perms)))
    tot = ident = 0
    stay = {h: 0 for h in (12, 18, 24)}
    for (i, j) in bonds:
        opp = ((reps >> i) & 1) != ((reps >> j) & 1)
        q = reps[opp] ^ ((1 << int(i)) | (1 << int(j)))
        _, g = canon(q)
        tot += q.size
        isid = g == 0
        ident += int(isid.sum())
        for h in stay:
            if (i < h) == (j < h):
                stay[h] += int(isid.sum())
    print("  off-diagonal entries per representative: %.2f" % (tot / reps.size))
    print("  flipped pattern is already the representative (no translation): %.1f %%" % (100.0 * ident / tot))
    for h, v in stay.items():
        s = v / tot
        print("  ... and the bond does not cross a cut at %d low sites: %.1f %%  -> projected traffic >= %.0f GB (target <= 500 GB, now 776 GB)"
              % (h, 100.0 * s, 295 + (1 - s) * 481))


if __name__ == "__main__":
    main()
