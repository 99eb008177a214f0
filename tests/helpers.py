"""Shared fixtures-as-functions for the parity tests (test infrastructure)."""
import functools
import json
import os

import numpy as np

import fastham
import refham
from quantum_basis_amd import lattices

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def known():
    return json.load(open(os.path.join(GOLDEN, "reference_known_answers.json")))


def probe():
    return json.load(open(os.path.join(GOLDEN, "survey_probe_goldens.json")))


@functools.lru_cache(maxsize=None)
def case(name):
    """-> (dim, ia, ja, val, sym) in the reference host layout (upper triangle unless noted)."""
    if name == "chain16_sz0":            # BASELINE config C1, reference Lin order
        d, ia, ja, val, _ = refham.heisenberg_csr(16, refham.chain_bonds(16), n_dn=8)
        return d, ia, ja, val, True
    if name == "chain16_full":           # src/main_test.cc test 1
        d, ia, ja, val, _ = refham.heisenberg_csr(16, refham.chain_bonds(16), n_dn=None)
        return d, ia, ja, val, True
    if name == "chain12_sz0":
        d, ia, ja, val, _ = refham.heisenberg_csr(12, refham.chain_bonds(12), n_dn=6)
        return d, ia, ja, val, True
    if name == "hubbard_4x2":            # reference order + sign convention
        d, ia, ja, val, _ = refham.hubbard_csr(4, 2, 4, 4)
        return d, ia, ja, val, True
    if name == "kagome_12":
        d, ia, ja, val, _ = refham.heisenberg_csr(12, lattices.kagome(2, 2), n_dn=6)
        return d, ia, ja, val, True
    if name == "triangular_4x4":
        d, ia, ja, val, _ = refham.heisenberg_csr(16, lattices.triangular(4, 4), n_dn=8)
        return d, ia, ja, val, True
    if name == "hubbard_4x2_fast_full":  # device-generator order, full storage
        d, ia, ja, val = fastham.to_ref_csr(fastham.hubbard_full(8, 4, 4, lattices.square(4, 2)))
        return d, ia, ja, val, False
    if name.startswith("chain16_k"):     # translation-symmetric sector, complex phases, full storage
        k = int(name[len("chain16_k"):])
        d, ia, ja, val = momentum_chain_csr(16, 8, k)
        return d, ia, ja, val, False
    raise KeyError(name)


def momentum_chain_csr(L, n_dn, k, J=1.0):
    """Heisenberg chain in the momentum-k sector (translation-symmetric 'repr' basis).

    |a(k)> = N_a^{-1/2} sum_r e^{-i 2 pi k r / L} T^r |a>, representatives a = smallest
    integer of the orbit; H_ab = sum_bonds h * sqrt(R_a / R_b) * e^{+- i 2 pi k l / L}
    (the same structure as generate_Ham_sparse_repr, src/model.cc:687-836: complex phases
    times sqrt(nu_i/nu_j)).  Returns a FULL-storage complex Hermitian CSR."""
    import itertools

    import scipy.sparse as sp
    mask = (1 << L) - 1

    def rot(s, r):
        return ((s << r) | (s >> (L - r))) & mask if r else s

    reps, period = {}, {}
    for comb in itertools.combinations(range(L), n_dn):
        s = sum(1 << i for i in comb)
        orbit = [rot(s, r) for r in range(L)]
        m = min(orbit)
        if s != m:
            continue
        R = orbit[1:].index(s) + 1 if s in orbit[1:] else L
        if (k * R) % L != 0:        # zero-norm representative for this momentum
            continue
        reps[s] = len(reps)
        period[s] = R
    dim = len(reps)
    rows, cols, vals = [], [], []
    diag = np.zeros(dim)
    for a, ia_ in reps.items():
        for x in range(L):
            y = (x + 1) % L
            if ((a >> x) ^ (a >> y)) & 1:
                diag[ia_] -= 0.25 * J
                b = a ^ (1 << x) ^ (1 << y)
                orbit = [rot(b, r) for r in range(L)]
                rb = min(orbit)
                if rb not in reps:
                    continue
                l = orbit.index(rb)          # T^l b = rep
                amp = 0.5 * J * np.sqrt(period[a] / period[rb]) * np.exp(2j * np.pi * k * l / L)
                rows.append(reps[rb])
                cols.append(ia_)
                vals.append(amp)
            else:
                diag[ia_] += 0.25 * J
    H = sp.coo_matrix((vals, (rows, cols)), shape=(dim, dim), dtype=np.complex128).tocsr()
    H = H + sp.diags(diag.astype(np.complex128))
    H = H.tocsr()
    H.sum_duplicates()
    H.sort_indices()
    # make sure every diagonal entry is stored
    H = (H + sp.diags(np.full(dim, 1e-300))).tocsr()
    H.sort_indices()
    return dim, H.indptr.astype(np.int64), H.indices.astype(np.int64), H.data.astype(np.complex128)


def gauge(dim, ia, ja, val, seed=7):
    """H -> D H D^+ with random unit phases D: a genuinely complex Hermitian matrix with the
    same spectrum (exercises the complex arithmetic of every kernel)."""
    rng = np.random.default_rng(seed)
    ph = np.exp(2j * np.pi * rng.random(dim))
    rows = np.repeat(np.arange(dim), np.diff(ia))
    return val * ph[rows] * np.conj(ph[ja]), ph


def expect_sz_sz(vec, basis_bits, i, j):
    """<Sz_i Sz_j> for a spin-1/2 state given in the bit basis (bit=1: down)."""
    si = 0.5 - ((basis_bits >> i) & 1)
    sj = 0.5 - ((basis_bits >> j) & 1)
    return float(np.sum(np.abs(vec) ** 2 * si * sj))


def expect_sp_sm(vec, basis_bits, i, j):
    """<S+_i S-_j>: flips site j up->down and site i down->up."""
    perm = np.argsort(basis_bits, kind="stable")
    sorted_bits = basis_bits[perm]
    ok = (((basis_bits >> j) & 1) == 0) & (((basis_bits >> i) & 1) == 1)
    tgt = basis_bits[ok] ^ ((1 << i) | (1 << j))
    pos = perm[np.searchsorted(sorted_bits, tgt)]
    return complex(np.sum(np.conj(vec[pos]) * vec[ok]))


def tj_chain_csr(L=12, n_elec=8, n_up=4, t=1.0, J=1.0):
    """t-J chain of src/main_test.cc:113-211 (PBC): H = -t sum P c+_is c_js P + h.c. + J sum (S_i.S_j - n_i n_j / 4),
    local states 0 = empty, 1 = up, 2 = down, fermion order by site.  Full-storage CSR in the reference layout."""
    import itertools

    import scipy.sparse as sp
    states = []
    for occ in itertools.combinations(range(L), n_elec):
        for ups in itertools.combinations(occ, n_up):
            s = [0] * L
            for i in occ:
                s[i] = 1 if i in ups else 2
            states.append(tuple(s))
    index = {s: i for i, s in enumerate(states)}
    rows, cols, vals = [], [], []
    diag = np.zeros(len(states))
    for a, s in enumerate(states):
        for i in range(L):
            j = (i + 1) % L
            si, sj = s[i], s[j]
            if si and sj:
                if si != sj:
                    diag[a] += -0.5 * J                  # J (-1/4) - J/4
                    ns = list(s)
                    ns[i], ns[j] = sj, si                 # (S+_i S-_j + h.c.) / 2
                    rows.append(index[tuple(ns)]), cols.append(a), vals.append(0.5 * J)
            elif si or sj:                                # one electron hops across the bond
                src, dst = (i, j) if si else (j, i)
                ns = list(s)
                ns[dst], ns[src] = s[src], 0
                lo, hi = min(src, dst), max(src, dst)
                nbetween = sum(1 for q in range(lo + 1, hi) if s[q])
                sign = -1.0 if nbetween % 2 else 1.0
                rows.append(index[tuple(ns)]), cols.append(a), vals.append(-t * sign)
    n = len(states)
    H = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr() + sp.diags(diag + 1e-300)
    H = H.tocsr()
    H.sum_duplicates()
    H.sort_indices()
    return n, H.indptr.astype(np.int64), H.indices.astype(np.int64), H.data.astype(np.complex128)
