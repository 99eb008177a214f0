"""Bond lists of the benchmark lattices (host-side input of the measurement harness).

Each function returns a list of (i, j) site pairs WITH multiplicity, exactly as the
reference's example programs add terms in their loops; site numbering is a free choice
(the spectrum is invariant under site relabelling).
"""


def chain(L, pbc=True):
    """examples/trans_absent/latt_chain/chain_Heisenberg_spin_half.cc:44-58"""
    return [(x, (x + 1) % L) for x in range(L if pbc else L - 1)]


def square(Lx, Ly, pbc=True):
    """examples/trans_absent/latt_square/square_Fermi_Hubbard.cc:47-93 (site = x + Lx*y).
    With Ly == 2 and PBC the y-bond appears twice, as in the reference's loop."""
    b = []
    for x in range(Lx):
        for y in range(Ly):
            s = x + Lx * y
            if pbc or x < Lx - 1:
                b.append((s, (x + 1) % Lx + Lx * y))
            if pbc or y < Ly - 1:
                b.append((s, x + Lx * ((y + 1) % Ly)))
    return b


def honeycomb(Lx, Ly):
    """examples/trans_symmetric/latt_honeycomb/honeycomb_Spinless_Fermion.cc:51-99 (PBC): site = sub + 2*(x + Lx*y); the
    sublattice-0 site of cell (x, y) is bonded to the sublattice-1 sites of cells (x, y), (x-1, y) and (x, y-1)."""
    b = []
    for x in range(Lx):
        for y in range(Ly):
            i = 2 * (x + Lx * y)
            for (cx, cy) in ((x, y), ((x - 1) % Lx, y), (x, (y - 1) % Ly)):
                b.append((i, 1 + 2 * (cx + Lx * cy)))
    return b


def triangular(Lx, Ly):
    """examples/trans_absent/latt_triangular/triangular_Heisenberg_spin_half.cc:50-86 (PBC):
    neighbours (m+1,n), (m+1,n+1), (m,n+1)."""
    b = []
    for m in range(Lx):
        for n in range(Ly):
            s = m + Lx * n
            b.append((s, (m + 1) % Lx + Lx * n))
            b.append((s, (m + 1) % Lx + Lx * ((n + 1) % Ly)))
            b.append((s, m + Lx * ((n + 1) % Ly)))
    return b


def kagome(Lx, Ly):
    """examples/trans_absent/latt_kagome/kagome_Heisenberg_spin_half.cc:69-152 (PBC):
    per unit cell (m,n), sublattices 0,1,2: 0-2(m+1,n), 0-2, 1-0(m,n+1), 1-0, 2-1(m-1,n-1), 2-1."""
    def site(m, n, sub):
        # x slowest: the two halves of the bit pattern are the two halves of the cluster, so only the bonds
        # across the cut (and the periodic wrap) flip one spin in each half -- better x-gather locality
        return sub + 3 * ((n % Ly) + Ly * (m % Lx))
    b = []
    for m in range(Lx):
        for n in range(Ly):
            i0, i1, i2 = site(m, n, 0), site(m, n, 1), site(m, n, 2)
            b.append((i0, site(m + 1, n, 2)))
            b.append((i0, i2))
            b.append((i1, site(m, n + 1, 0)))
            b.append((i1, i0))
            b.append((i2, site(m - 1, n - 1, 1)))
            b.append((i2, i1))
    return b


def kagome_torus(T1, T2):
    """The same kagome bond pattern on the torus spanned by T1 = (a, b), T2 = (c, d) (in unit cells): a*d - b*c cells.
    In this bond convention the two cell vectors are 120 degrees apart, so the six-fold symmetric 12-cell (36-site)
    cluster of the literature is T1 = (4, 2), T2 = (2, 4)."""
    (a, b), (c, d) = T1, T2
    det = a * d - b * c
    assert det > 0
    cells = {}
    for m in range(abs(det) + 1):
        for n in range(abs(det) + 1):
            key = ((d * m - c * n) % det, (-b * m + a * n) % det)
            if key not in cells:
                cells[key] = len(cells)
    assert len(cells) == det

    def site(m, n, sub):
        return sub + 3 * cells[((d * m - c * n) % det, (-b * m + a * n) % det)]
    bonds = set()
    for m in range(abs(det) + 1):
        for n in range(abs(det) + 1):
            i0, i1, i2 = site(m, n, 0), site(m, n, 1), site(m, n, 2)
            for p, q2 in ((i0, site(m + 1, n, 2)), (i0, i2), (i1, site(m, n + 1, 0)), (i1, i0), (i2, site(m - 1, n - 1, 1)), (i2, i1)):
                bonds.add((min(p, q2), max(p, q2)))
    assert len(bonds) == 6 * det
    return sorted(bonds)


# Site relabelling of the 36-site kagome cluster kagome_torus((4, 2), (2, 4)) found by simulated annealing on the bond
# list (tools/relabel_sites.py): 16 instead of 24 bonds cross the cut between the lower and the upper 18 sites, and only 4
# instead of 16 bonds join a site >= 18 to a site < 12.  In the colexicographic basis of qbh_mf_heisenberg a bond's
# highest site decides how far a flipped pattern's rank moves, so this numbering should keep more gathers inside
# cache-sized windows (the spectrum does not depend on the numbering).  MEASURED (round 2, dim 9.08e9): no effect -- 3.04
# instead of 3.08 TB per apply, 1.00 s either way; see DESIGN.md 4.6b.  Kept for the record, not used by default.
KAGOME36A_LOCAL = [22, 27, 30, 20, 8, 13, 1, 16, 11, 24, 28, 33, 35, 14, 25, 0, 6, 7, 2, 21, 12, 26, 31, 29, 17, 4, 5, 15, 23, 19,
                   32, 18, 34, 9, 3, 10]


def relabel(bonds, perm):
    """The same bond list with site s renamed perm[s]."""
    return sorted((min(perm[a], perm[b]), max(perm[a], perm[b])) for a, b in bonds)


def translations(Lx, Ly=1, n_sub=1, site=None):
    """All Lx*Ly translations of a periodic cluster as site permutations (first = identity) and their shifts.
    `site(x, y, sub)` is the numbering used for the bonds (default sub + n_sub*(x + Lx*y))."""
    if site is None:
        def site(x, y, s):
            return s + n_sub * (x + Lx * y)
    perms, shifts = [], []
    for tx in range(Lx):
        for ty in range(Ly):
            p = [0] * (Lx * Ly * n_sub)
            for x in range(Lx):
                for y in range(Ly):
                    for s in range(n_sub):
                        p[site(x, y, s)] = site((x + tx) % Lx, (y + ty) % Ly, s)
            perms.append(p)
            shifts.append((tx, ty))
    return perms, shifts


def characters(shifts, k, L):
    """Momentum characters chi_k(g) = exp(-2 pi i sum_d k_d t_d / L_d) for momentum indices k on an L = (Lx, Ly) cluster."""
    import cmath
    import math
    from fractions import Fraction
    exact = {Fraction(0): 1.0 + 0.0j, Fraction(1, 4): 0.0 - 1.0j, Fraction(1, 2): -1.0 + 0.0j, Fraction(3, 4): 0.0 + 1.0j}
    out = []
    for t in shifts:
        f = sum(Fraction(int(kd) * int(td), int(Ld)) for kd, td, Ld in zip(k, t, L)) % 1
        # multiples of pi/2 exactly: a sector at k = 0 or pi is then a REAL operator (packed-double Lanczos vectors)
        out.append(exact[f] if f in exact else cmath.exp(-2j * math.pi * float(f)))
    return out
