"""Lattice helpers used by the generators and benchmarks (host logic, no GPU)."""
import itertools

import numpy as np

from quantum_basis_amd import lattices


def _e0(bonds, n, ndn):
    states = [sum(1 << i for i in c) for c in itertools.combinations(range(n), ndn)]
    idx = {s: i for i, s in enumerate(states)}
    H = np.zeros((len(states),) * 2)
    for i, s in enumerate(states):
        for a, b in bonds:
            if ((s >> a) ^ (s >> b)) & 1:
                H[i, i] -= 0.25
                H[idx[s ^ (1 << a) ^ (1 << b)], i] += 0.5
            else:
                H[i, i] += 0.25
    return np.linalg.eigvalsh(H)[0]


def test_kagome_torus_matches_rectangular_helper_and_reference_answer():
    # examples/trans_absent/latt_kagome/kagome_Heisenberg_spin_half.cc:175: E0 = -5.444875217 on the 2x2 torus
    e_rect = _e0(lattices.kagome(2, 2), 12, 6)
    e_tor = _e0(lattices.kagome_torus((2, 0), (0, 2)), 12, 6)
    assert abs(e_rect + 5.444875217) < 1e-8 and abs(e_tor - e_rect) < 1e-12


def test_kagome_36a_cluster_shape():
    b = lattices.kagome_torus((4, 2), (2, 4))
    assert len(b) == 72 and len(set(b)) == 72
    deg = np.bincount(np.array(b).ravel(), minlength=36)
    assert deg.min() == deg.max() == 4
    # corner-sharing triangles: every site belongs to exactly two of the 24 triangles
    nb = {i: set() for i in range(36)}
    for x, y in b:
        nb[x].add(y)
        nb[y].add(x)
    tri = {tuple(sorted((i, j, k))) for i in range(36) for j in nb[i] for k in nb[i] & nb[j]}
    assert len(tri) == 24
    assert all(sum(i in t for t in tri) == 2 for i in range(36))


def _invariant_under(bonds, perms):
    und = sorted(tuple(sorted(b)) for b in bonds)
    for p in perms:
        if sorted(tuple(sorted((p[a], p[b]))) for a, b in bonds) != und:
            return False
    return True


def test_translations_map_every_bond_list_onto_itself():
    """the site permutations handed to the sector generators are symmetries of the bond lists they are used with"""
    for name, bonds, args in [("square 4x2", lattices.square(4, 2), (4, 2)), ("square 4x5", lattices.square(4, 5), (4, 5)),
                              ("triangular 6x6", lattices.triangular(6, 6), (6, 6)), ("chain 6", lattices.chain(6), (6, 1))]:
        perms, shifts = lattices.translations(*args)
        assert len(perms) == args[0] * args[1] and perms[0] == list(range(len(perms[0]))), name
        assert _invariant_under(bonds, perms), name
    hb = lattices.honeycomb(3, 2)
    perms, shifts = lattices.translations(3, 2, n_sub=2)
    assert len(hb) == 18 and np.bincount(np.array(hb).ravel(), minlength=12).tolist() == [3] * 12
    assert all(a % 2 == 0 and b % 2 == 1 for a, b in hb)          # bipartite: sublattice 0 -- sublattice 1
    assert _invariant_under(hb, perms)
    # characters: a one-dimensional representation of the translation group
    ch = lattices.characters(shifts, (1, 1), (3, 2))
    for (t1, c1) in zip(shifts, ch):
        for (t2, c2) in zip(shifts, ch):
            t12 = ((t1[0] + t2[0]) % 3, (t1[1] + t2[1]) % 2)
            assert abs(ch[shifts.index(t12)] - c1 * c2) < 1e-14
