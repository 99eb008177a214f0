"""numpy reference of the translation-symmetric ('repr') Heisenberg sector Hamiltonian in the convention of the
device generator qbh_gen_heisenberg_repr (test helper).

Basis: ALL orbit representatives (smallest bit pattern of each translation orbit) of the fixed-n_dn sector, ascending;
like the reference (src/model.cc:735-740) a representative whose norm vanishes at this momentum stays in the basis as a
decoupled row carrying only the fake diagonal fake_pos + i/dim.
Matrix element (row a, column b):  sum over bond terms taking |a> to c = l.b of  h * conj(chi(g*)) * sqrt(|S_b| / |S_a|),
g* the translation with g* c = b, chi the momentum character, S the stabiliser (src/model.cc:808-814: phase times
sqrt(nu_i/nu_j))."""
import itertools

import numpy as np
import scipy.sparse as sp


def translations_2d(Lx, Ly, n_sub=1, site=None):
    """All Lx*Ly translations of a periodic Lx x Ly (x n_sub) cluster as site permutations, with their shifts."""
    if site is None:
        def site(x, y, s):
            return s + n_sub * (x + Lx * y)
    perms, shifts = [], []
    for tx in range(Lx):
        for ty in range(Ly):
            p = np.zeros(Lx * Ly * n_sub, dtype=np.int32)
            for x in range(Lx):
                for y in range(Ly):
                    for s in range(n_sub):
                        p[site(x, y, s)] = site((x + tx) % Lx, (y + ty) % Ly, s)
            perms.append(p)
            shifts.append((tx, ty))
    return np.array(perms, dtype=np.int32), shifts


def characters(shifts, k, L):
    """chi_k(g) = exp(-2 pi i sum_d k_d t_d / L_d)."""
    return np.array([np.exp(-2j * np.pi * sum(kd * td / Ld for kd, td, Ld in zip(k, t, L))) for t in shifts])


def apply_perm(s, perm):
    out = 0
    for i, p in enumerate(perm):
        if (s >> i) & 1:
            out |= 1 << int(p)
    return out


def repr_basis(n_sites, n_dn, perms, chars):
    reps, stab, zero = [], [], []
    for comb in itertools.combinations(range(n_sites), n_dn):
        s = sum(1 << i for i in comb)
        imgs = [apply_perm(s, p) for p in perms]
        if min(imgs) != s:
            continue
        S = [g for g, t in enumerate(imgs) if t == s]
        sigma = sum(chars[g] for g in S)
        reps.append(s)
        stab.append(len(S))
        zero.append(abs(sigma) < 1e-10)
    order = np.argsort(reps)
    return (np.array(reps, dtype=np.int64)[order], np.array(stab)[order], np.array(zero)[order])


def repr_heisenberg_csr(n_sites, n_dn, bonds, perms, chars, J=1.0, fake_pos=100.0):
    reps, stab, zero = repr_basis(n_sites, n_dn, perms, chars)
    dim = len(reps)
    index = {int(s): i for i, s in enumerate(reps)}
    w = {}
    for (a, b) in bonds:
        key = (min(a, b), max(a, b))
        w[key] = w.get(key, 0.0) + 1.0
    rows, cols, vals = [], [], []
    for i, a in enumerate(reps):
        a = int(a)
        if zero[i]:
            rows.append(i), cols.append(i), vals.append(fake_pos + i / dim)
            continue
        acc = {i: 0.0 + 0.0j}
        for (x, y), wt in sorted(w.items()):
            if ((a >> x) ^ (a >> y)) & 1:
                acc[i] -= 0.25 * J * wt
                c = a ^ (1 << x) ^ (1 << y)
                imgs = [apply_perm(c, p) for p in perms]
                g = int(np.argmin(imgs))
                j = index[imgs[g]]
                if zero[j]:
                    continue
                acc[j] = acc.get(j, 0.0) + 0.5 * J * wt * np.conj(chars[g]) * np.sqrt(stab[j] / stab[i])
            else:
                acc[i] += 0.25 * J * wt
        for j, v in acc.items():
            if j == i or abs(v) >= 1e-14:
                rows.append(i), cols.append(j), vals.append(v)
    H = sp.coo_matrix((np.array(vals, dtype=np.complex128), (rows, cols)), shape=(dim, dim)).tocsr()
    H.sort_indices()
    return H, reps, stab, zero
