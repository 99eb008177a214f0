#!/usr/bin/env python3
"""How much of the gathered x does a rank of a row-sharded two-species (Hubbard) operator actually read?  
The far part of a shard of whole major indices (up configurations) gathers, for every own major index u, the majors u' = hop(u):
the rank needs the union of the neighbours of its block in the up-hop graph (all minor indices of each: the far entries keep the
minor index).  Received volume per rank = |N(block) \\ block| * S elements, against (NU - |block|) * S of the all-gather.
Orders compared: (i) the generator's order (ascending bit pattern = colexicographic rank), cut into P consecutive blocks as even
as they come (dist.kron_row_cuts); (ii) recursive spectral bisection of the hop graph (Fiedler vector of each part, split at the
median), eigenvalues being invariant under the permutation; (iii) greedy graph-growing refinement of (ii) (one Kernighan-Lin
style pass moving boundary nodes while the cut volume drops and the parts stay within 1 % of equal).
usage: python tools/needed_columns.py  ->  profiles/r6_lab/needed_columns.txt"""
import itertools
import sys
from math import comb

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, ".")
from quantum_basis_amd import lattices  # noqa: E402


def hop_graph(n_sites, n_up, bonds):
    cfgs = [sum(1 << i for i in c) for c in itertools.combinations(range(n_sites), n_up)]
    cfgs.sort()
    idx = {c: i for i, c in enumerate(cfgs)}
    rows, cols = [], []
    for i, c in enumerate(cfgs):
        for (a, b) in bonds:
            ba, bb = (c >> a) & 1, (c >> b) & 1
            if ba != bb:
                rows.append(i)
                cols.append(idx[c ^ ((1 << a) | (1 << b))])
    n = len(cfgs)
    G = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n))
    G.data[:] = 1.0
    return G


def needed(G, part, P):
    out = []
    for q in range(P):
        own = part == q
        nb = (G[own].sum(axis=0).A1 > 0) & ~own
        out.append((int(own.sum()), int(nb.sum())))
    return out


def bisect(G, nodes, P, part, base):
    if P == 1:
        part[nodes] = base
        return
    sub = G[nodes][:, nodes]
    deg = np.asarray(sub.sum(axis=1)).ravel()
    L = sp.diags(deg) - sub
    if len(nodes) <= 4000:
        w, v = np.linalg.eigh(L.toarray())
        f = v[:, 1]
    else:
        # Fiedler vector by power iteration on (c I - L) with the constant vector projected out
        c = 2.0 * deg.max()
        rng = np.random.default_rng(0)
        f = rng.normal(size=len(nodes))
        for _ in range(400):
            f -= f.mean()
            f = c * f - L @ f
            f /= np.linalg.norm(f)
    order = np.argsort(f)
    half = (len(nodes) * (P // 2)) // P
    bisect(G, nodes[order[:half]], P // 2, part, base)
    bisect(G, nodes[order[half:]], P - P // 2, part, base + P // 2)


def refine(G, part, P, passes=3):
    """move a node to the neighbouring part that lowers the total needed volume, parts within 1 % of equal"""
    n = G.shape[0]
    cap = int(np.ceil(n / P * 1.01))
    Gc = G.tocsr()
    for _ in range(passes):
        moved = 0
        sizes = np.bincount(part, minlength=P)
        for u in np.random.default_rng(1).permutation(n):
            nb = Gc.indices[Gc.indptr[u]:Gc.indptr[u + 1]]
            cnt = np.bincount(part[nb], minlength=P)
            best = int(np.argmax(cnt))
            if best != part[u] and cnt[best] > cnt[part[u]] and sizes[best] < cap:
                sizes[part[u]] -= 1
                sizes[best] += 1
                part[u] = best
                moved += 1
        if moved == 0:
            break
    return part


def main():
    out = []
    for name, (lx, ly, nu) in {"hubbard_4x4_half (C3)": (4, 4, 8), "hubbard_4x5 N=5 (C4 substitute)": (4, 5, 5)}.items():
        n = lx * ly
        G = hop_graph(n, nu, lattices.square(lx, ly))
        NU = G.shape[0]
        out.append("%s: %d up configurations, %d hops per configuration on average" % (name, NU, G.nnz // NU))
        for P in (2, 4, 8):
            lex = np.zeros(NU, dtype=np.int64)
            for q in range(P):
                lex[(q * NU) // P:((q + 1) * NU) // P] = q
            spec = np.zeros(NU, dtype=np.int64)
            bisect(G, np.arange(NU), P, spec, 0)
            ref = refine(G, spec.copy(), P)
            for label, part in (("generator order, consecutive blocks", lex), ("recursive spectral bisection", spec), ("  + greedy refinement", ref)):
                nd = needed(G, part, P)
                recv_all = [NU - o for o, _ in nd]
                frac = [b / r for (o, b), r in zip(nd, recv_all)]
                out.append("  P = %d  %-38s needed / all-gather per rank: min %.3f  mean %.3f  max %.3f   (block sizes %d..%d)"
                           % (P, label, min(frac), float(np.mean(frac)), max(frac), min(o for o, _ in nd), max(o for o, _ in nd)))
    txt = "\n".join(out)
    print(txt)
    open("profiles/r6_lab/needed_columns.txt", "w").write(__doc__ + "\n" + txt + "\n")


if __name__ == "__main__":
    main()
