"""Independent numpy assembly of the benchmark Hamiltonians in the DEVICE generators' basis
order (include/qbhip.h: qbh_gen_hubbard / qbh_gen_heisenberg) -- test helper used to check
the device-built CSR entry by entry at small sizes."""
import itertools

import numpy as np
import scipy.sparse as sp


def _configs(L, n):
    c = [sum(1 << s for s in comb) for comb in itertools.combinations(range(L), n)]
    return np.array(sorted(c), dtype=np.int64)


def _merge(bonds):
    w = {}
    for (a, b) in bonds:
        key = (min(a, b), max(a, b))
        w[key] = w.get(key, 0.0) + 1.0
    return sorted(w.items())


def _hop_matrix(L, n, bonds, t):
    cfg = _configs(L, n)
    idx = {int(c): i for i, c in enumerate(cfg)}
    rows, cols, vals = [], [], []
    for i, c in enumerate(cfg):
        c = int(c)
        for (s0, s1), w in _merge(bonds):
            for (a, b) in ((s0, s1), (s1, s0)):
                if (c >> a) & 1 and not (c >> b) & 1:
                    nc = (c ^ (1 << a)) | (1 << b)
                    lo, hi = min(a, b), max(a, b)
                    between = ((1 << hi) - 1) & ~((1 << (lo + 1)) - 1)
                    sign = -1.0 if bin(c & between).count("1") & 1 else 1.0
                    rows.append(i)
                    cols.append(idx[nc])
                    vals.append(-t * w * sign)
    n_c = len(cfg)
    return cfg, sp.coo_matrix((vals, (rows, cols)), shape=(n_c, n_c)).tocsr()


def hubbard_full(L, n_up, n_dn, bonds, t=1.0, U=1.1):
    """index = rank(up)*C(L,n_dn) + rank(dn); operator order: all up, then all down."""
    cu, Tu = _hop_matrix(L, n_up, bonds, t)
    cd, Td = _hop_matrix(L, n_dn, bonds, t)
    Nu, Nd = len(cu), len(cd)
    dbl = np.array([[bin(int(a) & int(b)).count("1") for b in cd] for a in cu], dtype=np.float64).ravel()
    H = sp.kron(Tu, sp.identity(Nd), format="csr") + sp.kron(sp.identity(Nu), Td, format="csr")
    H = (H + sp.diags(U * dbl)).tocsr()
    # the diagonal is always stored (zero included): add explicit entries
    H = _with_full_diagonal(H, U * dbl)
    return H


def _with_full_diagonal(H, diag):
    H = H.tocoo()
    off = H.row != H.col
    n = H.shape[0]
    rows = np.concatenate([H.row[off], np.arange(n)])
    cols = np.concatenate([H.col[off], np.arange(n)])
    vals = np.concatenate([H.data[off], diag])
    keep = (np.abs(vals) >= 1e-14) | (rows == cols)
    A = sp.csr_matrix((vals[keep], (rows[keep], cols[keep])), shape=(n, n))
    # scipy drops nothing here: explicit zeros on the diagonal survive csr_matrix construction
    A.sort_indices()
    return A


def heisenberg_full(L, n_dn, bonds, J=1.0):
    """index = colex rank of the down-spin pattern (ascending integer)."""
    cfg = _configs(L, n_dn)
    idx = {int(c): i for i, c in enumerate(cfg)}
    n = len(cfg)
    rows, cols, vals = [], [], []
    diag = np.zeros(n)
    mb = _merge(bonds)
    for i, c in enumerate(cfg):
        c = int(c)
        dg = 0.0
        for (a, b), w in mb:
            if ((c >> a) ^ (c >> b)) & 1:
                dg -= 0.25 * J * w
                rows.append(i)
                cols.append(idx[c ^ (1 << a) ^ (1 << b)])
                vals.append(0.5 * J * w)
            else:
                dg += 0.25 * J * w
        diag[i] = dg
    H = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    return _with_full_diagonal(H, diag)


def to_ref_csr(H):
    """scipy CSR -> (dim, ia int64, ja int64, val complex128)."""
    H = H.tocsr()
    H.sort_indices()
    return H.shape[0], H.indptr.astype(np.int64), H.indices.astype(np.int64), H.data.astype(np.complex128)
