"""Small-size Hamiltonians in the REFERENCE's basis order and sign conventions (test helper).

numpy re-derivation of what the reference's host pipeline produces for its own
test/example models, so that the oracle and the HIP path can be checked against
the reference's known answers and the survey's 17-digit vectors:

* bit packing        src/basis.cc:178-204   (site s -> bits [s*bps,(s+1)*bps))
* local states       src/basis.cc:52-83     (spin-1/2: 0=up 1=dn; electron: 0,up,dn,updn)
* basis order        src/basis.cc:1144-1190 (sort by (sub_b, sub_a): odd sites, then even sites)
* matrix elements    src/model.cc:649-679   (H[i][j] += conj(c); upper triangle keeps i<=j)
* lil_mat::add       src/sparse.cc:58-82    (drop |v|<1e-14, merge duplicates, diagonal always stored)
* fermion sign       src/basis.cc:2650-2664 (# fermions on lower sites; intra-site sign in local matrix)
"""
import itertools

import numpy as np
import scipy.sparse as sp

SPARSE_PRECISION = 1e-14


def _extract_sub(states, nsites, bps, parity):
    """Compact the sites of one parity (0: even -> sub_a, 1: odd -> sub_b) into an integer."""
    out = np.zeros_like(states)
    mask = (1 << bps) - 1
    k = 0
    for s in range(parity, nsites, 2):
        out |= ((states >> (s * bps)) & mask) << (k * bps)
        k += 1
    return out


def lin_order(states, nsites, bps):
    """Sort basis states in the reference's Lin-table order (src/basis.cc:1144-1190)."""
    states = np.asarray(states, dtype=np.int64)
    sub_a = _extract_sub(states, nsites, bps, 0)
    sub_b = _extract_sub(states, nsites, bps, 1)
    order = np.lexsort((sub_a, sub_b))
    return states[order]


def _index_of(sorted_states_by_value, perm, targets):
    pos = np.searchsorted(sorted_states_by_value, targets)
    pos = np.clip(pos, 0, len(sorted_states_by_value) - 1)
    ok = sorted_states_by_value[pos] == targets
    return perm[pos], ok


def _assemble(dim, rows, cols, vals, upper):
    rows = np.concatenate(rows + [np.arange(dim)])
    cols = np.concatenate(cols + [np.arange(dim)])
    vals = np.concatenate(vals + [np.zeros(dim, dtype=np.complex128)])   # diagonal always present
    if upper:
        keep = rows <= cols
        rows, cols, vals = rows[keep], cols[keep], vals[keep]
    A = sp.coo_matrix((vals, (rows, cols)), shape=(dim, dim)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    # drop cancelled off-diagonal entries, keep the diagonal (src/sparse.cc:72-77)
    coo = A.tocoo()
    keep = (np.abs(coo.data) >= SPARSE_PRECISION) | (coo.row == coo.col)
    A = sp.coo_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=(dim, dim)).tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.complex128)


def bit_patterns(nbits, k):
    """All nbits-bit integers with k bits set, ascending (vectorised: patterns(n, k) = patterns(n-1, k) followed by
    patterns(n-1, k-1) with bit n-1 set)."""
    table = {}

    def rec(n, j):
        if j < 0 or j > n:
            return np.zeros(0, dtype=np.int64)
        if j == 0:
            return np.zeros(1, dtype=np.int64)
        if (n, j) not in table:
            table[(n, j)] = np.concatenate([rec(n - 1, j), rec(n - 1, j - 1) | (np.int64(1) << (n - 1))])
        return table[(n, j)]

    return rec(nbits, k)


def spin_half_basis(nsites, n_dn=None):
    """All spin-1/2 states (bit=1: down), optionally with a fixed number of down spins, Lin order."""
    if n_dn is None:
        states = np.arange(1 << nsites, dtype=np.int64)
    else:
        states = bit_patterns(nsites, n_dn)
    return lin_order(states, nsites, 1)


def heisenberg_csr(nsites, bonds, J=1.0, n_dn=None, upper=True, basis=None):
    """H = J sum_<ij> [ (S+_i S-_j + S-_i S+_j)/2 + Sz_i Sz_j ], reference order.

    bonds: list of (i, j) with multiplicity.  Returns (dim, ia, ja, val, basis)."""
    if basis is None:
        basis = spin_half_basis(nsites, n_dn)
    dim = len(basis)
    perm = np.argsort(basis, kind="stable")
    sorted_states = basis[perm]
    rows, cols, vals = [], [], []
    diag = np.zeros(dim)
    all_rows = np.arange(dim)
    for (i, j) in bonds:
        bi = (basis >> i) & 1
        bj = (basis >> j) & 1
        differ = bi != bj
        diag += np.where(differ, -0.25 * J, 0.25 * J)
        tgt = basis[differ] ^ ((1 << i) | (1 << j))
        idx, ok = _index_of(sorted_states, perm, tgt)
        assert ok.all()
        rows.append(all_rows[differ])
        cols.append(idx)
        vals.append(np.full(idx.shape, 0.5 * J, dtype=np.complex128))
    rows.append(all_rows)
    cols.append(all_rows)
    vals.append(diag.astype(np.complex128))
    ia, ja, val = _assemble(dim, rows, cols, vals, upper)
    return dim, ia, ja, val, basis


def chain_bonds(L, pbc=True):
    return [(x, (x + 1) % L) for x in range(L if pbc else L - 1)]


def square_bonds(Lx, Ly, pbc=True):
    """Bonds exactly as the example loops add them (examples/.../square_Fermi_Hubbard.cc:47-93):
    for every (x,y): (x,y)-(x+1,y) and (x,y)-(x,y+1); with Ly=2 and PBC the y-bond is added twice.
    Site numbering: site = x + Lx*y (src/lattice.cc:546-582 with dim_spec = x)."""
    bonds = []
    for x in range(Lx):
        for y in range(Ly):
            s = x + Lx * y
            if pbc or x < Lx - 1:
                bonds.append((s, (x + 1) % Lx + Lx * y))
            if pbc or y < Ly - 1:
                bonds.append((s, x + Lx * ((y + 1) % Ly)))
    return bonds


def electron_basis(nsites, n_up, n_dn):
    """Electron states, 2 bits per site (bit0: up, bit1: down), fixed particle numbers, Lin order."""
    def spread(x):                        # bit s -> bit 2s
        out = np.zeros_like(x)
        for s_ in range(nsites):
            out |= ((x >> s_) & 1) << (2 * s_)
        return out
    ups = spread(bit_patterns(nsites, n_up))
    dns = spread(bit_patterns(nsites, n_dn)) << 1
    states = (ups[:, None] | dns[None, :]).ravel()
    return lin_order(states, nsites, 2)


def _popcount(x):
    x = x.astype(np.uint64)
    c = np.zeros(x.shape, dtype=np.int64)
    while True:
        nz = x != 0
        if not nz.any():
            break
        c += (x & np.uint64(1)).astype(np.int64)
        x = x >> np.uint64(1)
    return c


def hubbard_csr(Lx, Ly, n_up, n_dn, t=1.0, U=1.1, upper=True, pbc=True):
    """Fermi-Hubbard on the square lattice in the reference's conventions.

    H = -t sum_<ij>,s (c+_is c_js + h.c.) + U sum_i n_iup n_idn   (examples/.../square_Fermi_Hubbard.cc)."""
    nsites = Lx * Ly
    basis = electron_basis(nsites, n_up, n_dn)
    dim = len(basis)
    perm = np.argsort(basis, kind="stable")
    sorted_states = basis[perm]
    all_rows = np.arange(dim)
    rows, cols, vals = [], [], []
    # diagonal: U * number of doubly occupied sites
    dbl = np.zeros(dim)
    for s in range(nsites):
        dbl += ((basis >> (2 * s)) & 3) == 3
    rows.append(all_rows)
    cols.append(all_rows)
    vals.append((U * dbl).astype(np.complex128))
    for (i, j) in square_bonds(Lx, Ly, pbc):
        for spin in (0, 1):
            for (a, b) in ((2 * i + spin, 2 * j + spin), (2 * j + spin, 2 * i + spin)):
                # term -t c+_a c_b applied to |row>: needs b occupied, a empty
                can = (((basis >> b) & 1) == 1) & (((basis >> a) & 1) == 0)
                src = basis[can]
                tgt = (src ^ (1 << b)) | (1 << a)
                lo, hi = min(a, b), max(a, b)
                between = ((1 << hi) - 1) & ~((1 << (lo + 1)) - 1)
                sign = 1 - 2 * (_popcount(src & between) & 1)
                idx, ok = _index_of(sorted_states, perm, tgt)
                assert ok.all()
                rows.append(all_rows[can])
                cols.append(idx)
                vals.append((-t * sign).astype(np.complex128))
    ia, ja, val = _assemble(dim, rows, cols, vals, upper)
    return dim, ia, ja, val, basis


def csr_checksums(ia, ja, val):
    """The permutation-sensitive checksums of SURVEY.md Appendix E."""
    k = np.arange(len(ja), dtype=np.int64)
    return dict(sum_ia=int(ia.sum()), sum_ja_w=int((ja * (k % 7 + 1)).sum()),
                sum_val=complex(val.sum()), sum_abs=float(np.abs(val).sum()))
