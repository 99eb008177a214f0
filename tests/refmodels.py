"""Three more model families of the reference's example programs, assembled independently in numpy (test
infrastructure).  They reach the hot path as plain CSR matrices, like everything the reference's host code builds:

  spin1_chain          examples/trans_absent/latt_chain/chain_Heisenberg_spin_one.cc    L=10, Sz=0; E0, E1 asserted (:96-97)
  bose_hubbard_3x3     examples/trans_absent/latt_square/square_Bose_Hubbard.cc         9 bosons, Nmax=2; E0 (:100)
  spinless_honeycomb   examples/trans_absent/latt_honeycomb/honeycomb_Spinless_Fermion.cc  3x2 cells, N=4, full storage; E0 (:129)

Each returns (dim, ia, ja, val, sym) in the reference's CSR conventions (diagonal always stored, columns ascending)."""
import itertools

import numpy as np
import scipy.sparse as sp

KNOWN = {
    "spin1_chain": {"E0": -14.09412995, "E1": -13.569322, "tol": 1e-8},
    "bose_hubbard_3x3": {"E0": -25.81136094, "tol": 1e-8},
    "spinless_honeycomb": {"E0": -28.60363167, "tol": 1e-8},
}


def _to_csr(rows, cols, vals, dim, upper):
    M = sp.coo_matrix((vals, (rows, cols)), shape=(dim, dim), dtype=np.complex128).tocsr()
    M.sum_duplicates()
    M = M + sp.diags(np.full(dim, 1e-300))           # the reference stores every diagonal entry (src/qbasis.h:930)
    if upper:
        M = sp.triu(M, format="csr")
    M = sp.csr_matrix(M)
    M.sort_indices()
    return dim, M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.astype(np.complex128), bool(upper)


def spin1_chain(L=10, J=1.0, upper=True):
    """H = J sum_i S_i . S_{i+1}, spin 1, periodic, total Sz = 0.  Local states m = +1, 0, -1."""
    states = [s for s in itertools.product((1, 0, -1), repeat=L) if sum(s) == 0]
    index = {s: i for i, s in enumerate(states)}
    rows, cols, vals = [], [], []
    for i, s in enumerate(states):
        dg = 0.0
        for a in range(L):
            b = (a + 1) % L
            dg += J * s[a] * s[b]
            # (S+_a S-_b + S-_a S+_b) / 2 ; <m+1|S+|m> = sqrt(2 - m(m+1)) = sqrt(2) for m = 0, -1
            for da, db in ((1, -1), (-1, 1)):
                ma, mb = s[a] + da, s[b] + db
                if abs(ma) <= 1 and abs(mb) <= 1:
                    t = list(s)
                    t[a], t[b] = ma, mb
                    rows.append(index[tuple(t)])
                    cols.append(i)
                    vals.append(0.5 * J * 2.0)        # sqrt(2) * sqrt(2)
        rows.append(i)
        cols.append(i)
        vals.append(dg)
    return _to_csr(rows, cols, vals, len(states), upper)


def bose_hubbard_3x3(t=1.0, U=1.1, Lx=3, Ly=3, N=9, nmax=2, upper=True):
    """H = -t sum_<ij> (b+_i b_j + h.c.) + U/2 sum n(n-1), at most nmax bosons per site (truncated b matrices)."""
    ns = Lx * Ly
    site = lambda x, y: (x % Lx) + Lx * (y % Ly)
    bonds = []
    for x in range(Lx):
        for y in range(Ly):
            bonds.append((site(x, y), site(x + 1, y)))
            bonds.append((site(x, y), site(x, y + 1)))
    states = [s for s in itertools.product(range(nmax + 1), repeat=ns) if sum(s) == N]
    index = {s: i for i, s in enumerate(states)}
    rows, cols, vals = [], [], []
    for i, s in enumerate(states):
        rows.append(i)
        cols.append(i)
        vals.append(0.5 * U * sum(n * (n - 1) for n in s))
        for a, b in bonds:
            for src, dst in ((a, b), (b, a)):          # b+_dst b_src
                if s[src] > 0 and s[dst] < nmax:
                    tt = list(s)
                    amp = np.sqrt(s[src]) * np.sqrt(s[dst] + 1)
                    tt[src] -= 1
                    tt[dst] += 1
                    rows.append(index[tuple(tt)])
                    cols.append(i)
                    vals.append(-t * amp)
    return _to_csr(rows, cols, vals, len(states), upper)


def spinless_honeycomb(t=1.0, V1=4.0, Lx=3, Ly=2, N=None, upper=False):
    """H = sum_<ij> [-t (c+_i c_j + h.c.) + V1 n_i n_j - V1/2 (n_i + n_j)], honeycomb Lx x Ly cells, periodic.
    The example generates FULL storage (generate_Ham_sparse_full(0, false)); Jordan-Wigner order = site number."""
    N = Lx * Ly - 2 if N is None else N
    ns = 2 * Lx * Ly
    site = lambda x, y, sub: sub + 2 * ((x % Lx) + Lx * (y % Ly))
    bonds = []
    for x in range(Lx):
        for y in range(Ly):
            a = site(x, y, 0)
            for b in (site(x, y, 1), site(x - 1, y, 1), site(x - 1, y - 1, 1)):
                bonds.append((a, b))
    states = [sum(1 << p for p in c) for c in itertools.combinations(range(ns), N)]
    states.sort()
    index = {s: i for i, s in enumerate(states)}
    rows, cols, vals = [], [], []
    for i, s in enumerate(states):
        dg = 0.0
        for a, b in bonds:
            na, nb = (s >> a) & 1, (s >> b) & 1
            dg += V1 * na * nb - 0.5 * V1 * (na + nb)
            for src, dst in ((a, b), (b, a)):          # c+_dst c_src
                if (s >> src) & 1 and not (s >> dst) & 1:
                    lo, hi = min(src, dst), max(src, dst)
                    between = bin(s & (((1 << hi) - 1) & ~((1 << (lo + 1)) - 1))).count("1")
                    rows.append(index[s ^ (1 << src) ^ (1 << dst)])
                    cols.append(i)
                    vals.append(-t * (-1.0) ** between)
        rows.append(i)
        cols.append(i)
        vals.append(dg)
    return _to_csr(rows, cols, vals, len(states), upper)


CASES = {"spin1_chain": spin1_chain, "bose_hubbard_3x3": bose_hubbard_3x3, "spinless_honeycomb": spinless_honeycomb}
