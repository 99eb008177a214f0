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
