"""the operator-apply step model<T>::moprXvec_full (src/model.cc:1468-1538) on the device, against an
independent numpy operator-apply, and measure_full_dynamic (src/model.cc:1696-1712) end to end against the oracle's
"dnmcs" coefficients."""
import math

import numpy as np
import pytest

import refham
import quantum_basis_amd as q
from quantum_basis_amd import lattices
from oracle import qb_oracle as qo

pytestmark = pytest.mark.gpu


def _patterns(n_sites, n_dn):
    return refham.bit_patterns(n_sites, n_dn)              # ascending == colexicographic rank order of the generator


def _np_spin_apply(n_sites, n_dn_old, kind, coef, x):
    """vec_new = sum_s coef_s O_s |x> by explicit state bookkeeping (dictionary lookups), 0 = up, 1 = down"""
    old = _patterns(n_sites, n_dn_old)
    new = _patterns(n_sites, n_dn_old - kind)
    index = {int(p): i for i, p in enumerate(new)}
    y = np.zeros(len(new), dtype=np.complex128)
    for j, p in enumerate(old):
        p = int(p)
        for s in range(n_sites):
            bit = (p >> s) & 1
            if kind == 0:
                y[index[p]] += coef[s] * (-0.5 if bit else 0.5) * x[j]
            elif kind == -1 and bit == 0:                  # S^-: up -> down
                y[index[p | (1 << s)]] += coef[s] * x[j]
            elif kind == +1 and bit == 1:                  # S^+: down -> up
                y[index[p & ~(1 << s)]] += coef[s] * x[j]
    return y


@pytest.mark.parametrize("n_sites,n_dn", [(10, 5), (12, 3), (13, 9)])
def test_spin_operators_match_numpy(n_sites, n_dn):
    rng = np.random.default_rng(n_sites)
    coef = np.exp(2j * np.pi * 3 * np.arange(n_sites) / n_sites) / np.sqrt(n_sites) * (1 + 0.1 * rng.normal(size=n_sites))
    A = q.csr_mat.heisenberg(n_sites, n_dn, lattices.chain(n_sites))       # only a handle for device vectors
    d_old = math.comb(n_sites, n_dn)
    x = (rng.normal(size=d_old) + 1j * rng.normal(size=d_old)).astype(np.complex128)
    vx = q.DeviceVec(A, d_old)
    vx.upload(x)
    for kind in (0, -1, +1):
        d_new = math.comb(n_sites, n_dn - kind)
        vy = q.DeviceVec(A, d_new)
        q.moprXvec_spin(n_sites, n_dn, kind, coef, vx.ptr, vy.ptr)
        want = _np_spin_apply(n_sites, n_dn, kind, coef, x)
        got = vy.download()
        assert np.abs(got - want).max() <= 1e-14 * max(np.abs(want).max(), 1.0), kind
        vy.free()
    vx.free()
    A.destroy()


def test_onebody_operator_reproduces_the_hopping_part_of_the_hubbard_matrix():
    """sum over bonds, directions and spins of -t c+_a c_b must equal (H at U = 0) x, and with densities added
    U-independent pieces; this pins the fermion sign convention of the operator-apply to the generator's."""
    n_sites, n_up, n_dn = 8, 3, 4
    bonds = np.asarray(lattices.square(4, 2)).reshape(-1, 2)
    H0 = q.csr_mat.hubbard(n_sites, n_up, n_dn, bonds, t=1.0, U=0.0)
    d = H0.dim
    rng = np.random.default_rng(5)
    x = (rng.normal(size=d) + 1j * rng.normal(size=d)).astype(np.complex128)
    want = np.empty_like(x)
    H0.MultMv(x, want)
    terms = []
    for (a, b) in bonds:
        for spin in (0, 1):
            terms += [(int(a), int(b), spin, -1.0), (int(b), int(a), spin, -1.0)]
    v = q.DeviceVec(H0, 2 * d)
    v.upload(x)
    q.moprXvec_onebody(n_sites, n_up, n_dn, terms, v.at(0), v.at(d))
    got = v.download(d, d)
    assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()
    # densities: n_q = sum_s c_s (n_s,up + n_s,dn) is diagonal; total particle number is (n_up + n_dn) * x
    q.moprXvec_onebody(n_sites, n_up, n_dn, [(s, s, sp, 1.0) for s in range(n_sites) for sp in (0, 1)], v.at(0), v.at(d))
    assert np.abs(v.download(d, d) - (n_up + n_dn) * x).max() <= 1e-13 * np.abs(x).max()
    # a complex-weighted current-like operator against numpy bookkeeping on the product basis
    cu, cd = _patterns(n_sites, n_up), _patterns(n_sites, n_dn)
    iu = {int(c): i for i, c in enumerate(cu)}
    idn = {int(c): i for i, c in enumerate(cd)}
    w = 0.3 - 0.7j
    a, b = 6, 1
    ref = np.zeros(d, dtype=np.complex128)
    for u, c in enumerate(cu):
        c = int(c)
        if (c >> b) & 1 and not (c >> a) & 1:                 # c+_a c_b on the up species
            nc = (c ^ (1 << b)) | (1 << a)
            between = sum((c >> s) & 1 for s in range(min(a, b) + 1, max(a, b)))
            for dd in range(len(cd)):
                ref[iu[nc] * len(cd) + dd] += w * (-1) ** between * x[u * len(cd) + dd]
    q.moprXvec_onebody(n_sites, n_up, n_dn, [(a, b, 0, w)], v.at(0), v.at(d))
    assert np.abs(v.download(d, d) - ref).max() <= 1e-14 * np.abs(x).max()
    assert idn                                                 # (down-species map unused above: up operator only)
    v.free()
    H0.destroy()


def test_measure_full_dynamic_end_to_end_against_the_oracle():
    """S^z_q and S^-_q spectral functions of the Heisenberg chain L = 16: ground state on the device, A_q |phi0> on the
    device, "dnmcs" Lanczos on the device; the continued-fraction coefficients against the oracle fed with the numpy
    operator-apply of the same ground state."""
    L, n_dn, maxit = 16, 8, 200
    bonds = lattices.chain(L)
    A = q.csr_mat.heisenberg(L, n_dn, bonds)
    res = q.locate_E0_lanczos(A, nev=1, ncv=1)
    phi = res.eigenvecs
    assert abs(res.E0 - (-7.1422963606168)) < 1e-9            # BASELINE C1 known answer
    vphi = q.DeviceVec(A, A.dim)
    vphi.upload(phi)
    qk = 5
    coef = np.exp(2j * np.pi * qk * np.arange(L) / L) / np.sqrt(L)
    for kind, n_new in ((0, n_dn), (-1, n_dn + 1)):
        B = A if kind == 0 else q.csr_mat.heisenberg(L, n_new, bonds)
        m, norm, hess = q.measure_full_dynamic_dev(B, lambda dst: q.moprXvec_spin(L, n_dn, kind, coef, vphi.ptr, dst), maxit)
        # oracle: same operator in the same basis (downloaded), start vector from the numpy operator-apply
        ia, ja, val = B.download()
        O = qo.Csr(B.dim, ia, ja.astype(np.int64), val, False)
        y = _np_spin_apply(L, n_dn, kind, coef, phi)
        nrm = np.linalg.norm(y)
        assert abs(norm - nrm) <= 1e-12 * nrm
        v = np.zeros(2 * B.dim, dtype=np.complex128)
        v[:B.dim] = y / nrm
        ho = np.zeros(2 * maxit)
        mo = qo.lanczos(0, maxit - 1, maxit, O, v, ho, "dnmcs")[0]
        assert abs(m - mo) <= 2 and min(m, mo) >= 100
        # the leading coefficients agree to rounding; later ones are individually sensitive to rounding once Ritz values
        # have converged (Lanczos loses orthogonality), so the physical output is compared instead: the dynamical
        # correlation function as the continued fraction of the coefficients (docs/Manual: eq. of the dnmcs section)
        k = 12
        assert np.allclose(hess[maxit:maxit + k], ho[maxit:maxit + k], rtol=1e-9, atol=1e-11)
        assert np.allclose(hess[1:k + 1], ho[1:k + 1], rtol=1e-9, atol=1e-11)

        def green(h, mm, z):
            g = 0.0
            for j in range(mm - 1, -1, -1):
                g = 1.0 / (z - h[maxit + j] - (h[j + 1] ** 2) * g)
            return g

        for w in np.linspace(0.0, 4.0, 9):
            z = res.E0 + w + 0.1j
            g1, g2 = norm ** 2 * green(hess, min(m, mo) - 1, z), nrm ** 2 * green(ho, min(m, mo) - 1, z)
            assert abs(g1 - g2) <= 1e-8 * max(abs(g2), 1e-3), (kind, w)
        if B is not A:
            B.destroy()
    # sum rule: sum_q <phi| S^z_-q S^z_q |phi> over all q equals sum_s <(S^z_s)^2> = L/4
    total = 0.0
    vy = q.DeviceVec(A, A.dim)
    for kq in range(L):
        c = np.exp(2j * np.pi * kq * np.arange(L) / L) / np.sqrt(L)
        q.moprXvec_spin(L, n_dn, 0, c, vphi.ptr, vy.ptr)
        total += A.nrm2(vy.ptr) ** 2
    assert abs(total - L / 4) < 1e-10
    vy.free()
    vphi.free()
    A.destroy()


def _momentum_states(n_sites, n_dn, perms, chars, conj):
    """Explicit momentum states as columns in the FULL n_dn basis: |a, k> = (1/sqrt(|G||S_a|)) sum_g chi_k(g)^(*) T_g |a> for
    every representative a (ascending); zero columns for representatives whose norm vanishes.  Returns (P, reps, zero)."""
    import reprham
    full = _patterns(n_sites, n_dn)
    index = {int(p): i for i, p in enumerate(full)}
    reps, stab, zero = reprham.repr_basis(n_sites, n_dn, perms, chars)
    G = len(perms)
    P = np.zeros((len(full), len(reps)), dtype=np.complex128)
    for j, a in enumerate(reps):
        if zero[j]:
            continue
        for g in range(G):
            t = reprham.apply_perm(int(a), perms[g])
            P[index[t], j] += (np.conj(chars[g]) if conj else chars[g]) / np.sqrt(G * stab[j])
    return P, reps, zero


def test_sz_q_between_momentum_sectors_matches_the_explicit_projection():
    """moprXvec_repr for S^z_q (src/model.cc:1715-1846): device result against P_new^+ A_q P_old built from explicit momentum
    states in the full basis.  The convention of those states is first pinned against the device-assembled sector
    Hamiltonian (P^+ H_full P = H_repr), so the check does not depend on a hand-derived formula."""
    import reprham
    n_sites, n_dn, L = 12, 6, (12, 1)
    bonds = lattices.chain(n_sites)
    perms, shifts = lattices.translations(12, 1)
    perms = np.asarray(perms)
    k_old, qk = 2, 3
    k_new = (k_old + qk) % n_sites
    ch_old = np.asarray(lattices.characters(shifts, (k_old, 0), L))
    ch_new = np.asarray(lattices.characters(shifts, (k_new, 0), L))
    # full-basis Hamiltonian in the pattern-ascending basis (device generator order)
    F = q.csr_mat.heisenberg(n_sites, n_dn, bonds)
    ia, ja, val = F.download()
    import scipy.sparse as sp
    Hfull = sp.csr_matrix((val, ja, ia), shape=(F.dim, F.dim)).toarray()
    A_old = q.csr_mat.heisenberg_repr(n_sites, n_dn, bonds, perms, ch_old, opts=q.make_opts(value_dict=0))
    ia, ja, val = A_old.download()
    Hrepr = sp.csr_matrix((val, ja, ia), shape=(A_old.dim, A_old.dim)).toarray()
    conj = None
    for c in (False, True):
        P, reps, zero = _momentum_states(n_sites, n_dn, perms, ch_old, c)
        M = P.conj().T @ Hfull @ P
        ok = ~zero
        if np.abs(M[np.ix_(ok, ok)] - Hrepr[np.ix_(ok, ok)]).max() < 1e-12:
            conj = c
    assert conj is not None                                   # one of the two conventions reproduces the sector operator
    P_old, reps, zero_old = _momentum_states(n_sites, n_dn, perms, ch_old, conj)
    P_new, _, zero_new = _momentum_states(n_sites, n_dn, perms, ch_new, conj)
    # A_q = sum_s c_s S^z_s with c_s = exp(+-i q r_s)/sqrt(N): the sign that maps k_old to k_new in this convention
    full = _patterns(n_sites, n_dn)
    rng = np.random.default_rng(3)
    x = (rng.normal(size=len(reps)) + 1j * rng.normal(size=len(reps))) * (~zero_old)
    found = False
    for sign in (+1, -1):
        coef = np.exp(sign * 2j * np.pi * qk * np.arange(n_sites) / n_sites) / np.sqrt(n_sites)
        sz = np.array([[(-0.5 if (int(p) >> s) & 1 else 0.5) for s in range(n_sites)] for p in full])
        Aq = sz @ coef                                        # diagonal of A_q in the full basis
        phi = Aq * (P_old @ x)
        want = P_new.conj().T @ phi
        # completeness: A_q maps the k_old subspace into the k_new subspace only for the right sign
        if abs(np.linalg.norm(want) - np.linalg.norm(phi)) > 1e-10 * max(np.linalg.norm(phi), 1e-30):
            continue
        found = True
        vx = q.DeviceVec(A_old, len(reps))
        vy = q.DeviceVec(A_old, len(reps))
        vx.upload(x)
        dim = q.moprXvec_sz_repr(n_sites, n_dn, perms, ch_new, coef, vx.ptr, vy.ptr)
        got = vy.download()
        assert dim == len(reps) == A_old.dim
        assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()
        assert np.all(got[zero_new] == 0)
        vx.free()
        vy.free()
    assert found
    F.destroy()
    A_old.destroy()


@pytest.mark.parametrize("kind", [-1, +1])
def test_spin_flip_operators_between_momentum_sectors_match_the_explicit_projection(kind):
    """The off-diagonal branch of moprXvec_repr (src/model.cc:1760-1830): S^-_q / S^+_q from (n_dn, k) to (n_dn -+ 1... , k + q)
    against P_new^+ A_q P_old with explicit momentum states; convention pinned by P^+ H P = H_repr as above."""
    import scipy.sparse as sp
    n_sites, n_dn, L = 12, 5, (12, 1)
    n_new = n_dn - kind
    bonds = lattices.chain(n_sites)
    perms, shifts = lattices.translations(12, 1)
    perms = np.asarray(perms)
    k_old, qk = 1, 4
    ch_old = np.asarray(lattices.characters(shifts, (k_old, 0), L))
    F = q.csr_mat.heisenberg(n_sites, n_dn, bonds)
    ia, ja, val = F.download()
    Hfull = sp.csr_matrix((val, ja, ia), shape=(F.dim, F.dim)).toarray()
    A_old = q.csr_mat.heisenberg_repr(n_sites, n_dn, bonds, perms, ch_old, opts=q.make_opts(value_dict=0))
    ia, ja, val = A_old.download()
    Hrepr = sp.csr_matrix((val, ja, ia), shape=(A_old.dim, A_old.dim)).toarray()
    conj = None
    for c in (False, True):
        P, reps, zero = _momentum_states(n_sites, n_dn, perms, ch_old, c)
        ok = ~zero
        if np.abs((P.conj().T @ Hfull @ P)[np.ix_(ok, ok)] - Hrepr[np.ix_(ok, ok)]).max() < 1e-12:
            conj = c
    assert conj is not None
    P_old, reps_old, zero_old = _momentum_states(n_sites, n_dn, perms, ch_old, conj)
    rng = np.random.default_rng(11 + kind)
    x = (rng.normal(size=len(reps_old)) + 1j * rng.normal(size=len(reps_old))) * (~zero_old)
    found = False
    for sign in (+1, -1):
        k_new = (k_old + sign * qk) % n_sites
        ch_new = np.asarray(lattices.characters(shifts, (k_new, 0), L))
        P_new, reps_new, zero_new = _momentum_states(n_sites, n_new, perms, ch_new, conj)
        for csign in (+1, -1):
            coef = np.exp(csign * 2j * np.pi * qk * np.arange(n_sites) / n_sites) / np.sqrt(n_sites)
            phi = _np_spin_apply(n_sites, n_dn, kind, coef, P_old @ x)
            want = P_new.conj().T @ phi
            if abs(np.linalg.norm(want) - np.linalg.norm(phi)) > 1e-10 * max(np.linalg.norm(phi), 1e-30):
                continue                                  # this (coef sign, target momentum) pair does not match: A_q |k> is not in k_new
            found = True
            vx = q.DeviceVec(A_old, len(reps_old))
            vy = q.DeviceVec(A_old, len(reps_new))
            vx.upload(x)
            d0, d1 = q.moprXvec_flip_repr(n_sites, n_dn, kind, perms, ch_old, ch_new, coef, vx.ptr, vy.ptr)
            got = vy.download()
            assert (d0, d1) == (len(reps_old), len(reps_new))
            assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max(), (kind, sign, csign)
            assert np.all(got[zero_new] == 0)
            vx.free()
            vy.free()
    assert found
    F.destroy()
    A_old.destroy()


def test_measure_full_static_reproduces_the_references_asserted_correlators():
    """src/main_test.cc:94-108: <Sz0 Sz1>, <Sz0 Sz2>, <S+0 S-1> in the ground state of the L = 16 chain, here with the ground
    state, the operator products and the inner product all on the device (the ground state lies in the Sz = 0 sector)."""
    import helpers
    k = helpers.known()["chain16_full"]
    L, n_dn = 16, 8
    A = q.csr_mat.heisenberg(L, n_dn, lattices.chain(L))
    res = q.locate_E0_lanczos(A, nev=1, ncv=1)
    assert abs(res.E0 - k["E0"]) < k["tol"]
    v = q.DeviceVec(A, A.dim)
    v.upload(res.eigenvecs)

    def at(site):
        c = np.zeros(L, dtype=np.complex128)
        c[site] = 1.0
        return c
    m1 = q.measure_full_static_spin_dev(A, L, n_dn, v.ptr, [(0, at(0)), (0, at(1))])
    m2 = q.measure_full_static_spin_dev(A, L, n_dn, v.ptr, [(0, at(0)), (0, at(2))])
    m3 = q.measure_full_static_spin_dev(A, L, n_dn, v.ptr, [(+1, at(0)), (-1, at(1))])        # S+_0 S-_1: S-_1 acts first
    assert abs(m1 - k["Sz0Sz1"]) < k["tol"] and abs(m1.imag) < 1e-12
    assert abs(m2 - k["Sz0Sz2"]) < k["tol"]
    assert abs(m3 - k["Sp0Sm1"]) < k["tol"]
    with pytest.raises(ValueError):
        q.measure_full_static_spin_dev(A, L, n_dn, v.ptr, [(-1, at(1))])
    v.free()
    A.destroy()


# ---- the general term list (qbh_mopr_terms_dev): any mopr as a sum of ordered products of elementary site operators ----
def _dense_spin_ops(n):
    """S^z_s, S^+_s, S^-_s on the 2^n product space; state index = bit pattern, bit s = 1: site s is DOWN"""
    sz, sp = np.diag([0.5, -0.5]), np.array([[0.0, 1.0], [0.0, 0.0]])           # local index 0 = up, 1 = down; S^+ |down> = |up>
    ops = {}
    for s in range(n):
        for name, m in (("Sz", sz), ("S+", sp), ("S-", sp.T)):
            full = np.array([[1.0]])
            for k in reversed(range(n)):                                          # site n-1 is the most significant bit
                full = np.kron(full, m if k == s else np.eye(2))
            ops[(name, s)] = full
    return ops


def _dense_fermion_ops(n_orb):
    """c+_o, c_o, n_o by Jordan-Wigner on 2^n_orb states; |w> = prod_{o ascending} c+_o |0>, so c+_o |w> = (-1)^{occupied below o} |w + o>"""
    dim = 1 << n_orb
    ops = {}
    for o in range(n_orb):
        cd = np.zeros((dim, dim))
        for w in range(dim):
            if not (w >> o) & 1:
                cd[w | (1 << o), w] = -1.0 if bin(w & ((1 << o) - 1)).count("1") & 1 else 1.0
        ops[("c+", o)], ops[("c", o)], ops[("n", o)] = cd, cd.T.copy(), cd @ cd.T
    return ops


@pytest.mark.parametrize("delta", [0, 1, -1])
def test_general_term_list_on_a_spin_sector_matches_dense_operators(delta):
    n, nd = 9, 4
    rng = np.random.default_rng(11 + delta)
    ops = _dense_spin_ops(n)
    tail = {0: [], 1: [("S-", 6)], -1: [("S+", 2)]}[delta]                          # S^- adds a down spin, S^+ removes one
    terms = [(0.7 + 0.2j, [("S+", 2), ("S-", 5), ("Sz", 1)] + tail), (-1.3j, [("S-", 0), ("S+", 7)] + tail), (0.4, [("Sz", 3), ("Sz", 3)] + tail),
             (1.1, tail + [("Sz", 8)]), (0.25 - 0.5j, [("S+", 4), ("S-", 4), ("S-", 1), ("S+", 0)] + tail)]
    dense = sum(c * np.linalg.multi_dot([ops[f] for f in fs] + [np.eye(1 << n)]) for c, fs in terms)
    old, new = _patterns(n, nd), _patterns(n, nd + delta)
    x = (rng.normal(size=len(old)) + 1j * rng.normal(size=len(old))).astype(np.complex128)
    want = (dense[np.ix_(new, old)] @ x)
    A = q.csr_mat.heisenberg(n, nd, lattices.chain(n))
    vx, vy = q.DeviceVec(A, len(old)), q.DeviceVec(A, len(new))
    vx.upload(x)
    dim_new = q.moprXvec_terms("spin", n, nd, 0, terms, vx.ptr, vy.ptr)
    assert dim_new == len(new)
    got = vy.download()
    assert np.abs(got - want).max() <= 1e-13 * max(np.abs(want).max(), 1.0)
    # the special-purpose entry point is one instance of the general one
    if delta:
        coef = np.exp(2j * np.pi * np.arange(n) / n)
        q.moprXvec_spin(n, nd, -delta, coef, vx.ptr, vy.ptr)
        a = vy.download()
        q.moprXvec_terms("spin", n, nd, 0, [(coef[s], [("S-" if delta > 0 else "S+", s)]) for s in range(n)], vx.ptr, vy.ptr)
        assert np.abs(vy.download() - a).max() <= 1e-13
    vx.free(), vy.free()
    A.destroy()


@pytest.mark.parametrize("case", ["number_conserving", "spin_flip", "remove_up", "add_pair"])
def test_general_term_list_on_two_species_fermions_matches_jordan_wigner(case):
    """Fermion signs against an independent Jordan-Wigner construction on the 2^(2 n_sites) Fock space: the convention is the
    generators' (all up operators left of all down operators, sites ascending inside a species)."""
    n, nu, nd = 5, 2, 3
    ops = _dense_fermion_ops(2 * n)
    F = lambda name, s, sp: (name, s + sp * n)                                       # dense orbital index
    if case == "number_conserving":
        terms = [(0.8, [("c+", 1, 0), ("c", 3, 0)]), (-0.6j, [("c+", 4, 1), ("c", 0, 1)]), (1.5, [("n", 2, 0), ("n", 2, 1)]),
                 (0.3 + 0.1j, [("c+", 0, 0), ("c+", 1, 1), ("c", 3, 1), ("c", 2, 0)]), (2.0, [("c", 1, 0), ("c+", 1, 0)])]
        dnu, dnd = 0, 0
    elif case == "spin_flip":                   # S^+_s = c+_{s,up} c_{s,down} and a hop that flips the spin: (n_up, n_dn) -> (n_up + 1, n_dn - 1)
        terms = [(1.0, [("c+", s, 0), ("c", s, 1)]) for s in range(n)] + [(0.5j, [("c+", 0, 0), ("c", 4, 1), ("n", 2, 1)])]
        dnu, dnd = 1, -1
    elif case == "remove_up":                   # the photo-emission operator c_{q,up}
        terms = [(np.exp(2j * np.pi * s / n), [("c", s, 0)]) for s in range(n)]
        dnu, dnd = -1, 0
    else:                                       # a pair creation c+_{s,up} c+_{s,down}
        terms = [(1.0 + 0.5 * s, [("c+", s, 0), ("c+", s, 1)]) for s in range(n)]
        dnu, dnd = 1, 1
    dense = sum(c * np.linalg.multi_dot([ops[F(*f)] for f in fs] + [np.eye(1 << (2 * n))]) for c, fs in terms)
    pu, pd_ = _patterns(n, nu), _patterns(n, nd)
    pu2, pd2 = _patterns(n, nu + dnu), _patterns(n, nd + dnd)
    old = np.array([int(u) | (int(d) << n) for u in pu for d in pd_])                # index = rank(up) * C(n, n_dn) + rank(down)
    new = np.array([int(u) | (int(d) << n) for u in pu2 for d in pd2])
    rng = np.random.default_rng(3)
    x = (rng.normal(size=len(old)) + 1j * rng.normal(size=len(old))).astype(np.complex128)
    want = dense[np.ix_(new, old)] @ x
    A = q.csr_mat.hubbard(n, nu, nd, lattices.chain(n))
    vx, vy = q.DeviceVec(A, len(old)), q.DeviceVec(A, len(new))
    vx.upload(x)
    assert q.moprXvec_terms("fermion", n, nu, nd, terms, vx.ptr, vy.ptr) == len(new)
    assert np.abs(vy.download() - want).max() <= 1e-13 * max(np.abs(want).max(), 1.0)
    if case == "number_conserving":             # ... and agrees with the one-body entry point on its own ground
        one = [(1, 3, 0, 0.8), (4, 0, 1, -0.6j), (2, 2, 1, 0.25)]
        q.moprXvec_onebody(n, nu, nd, one, vx.ptr, vy.ptr)
        a = vy.download()
        q.moprXvec_terms("fermion", n, nu, nd, [(w, [("c+", a_, sp), ("c", b_, sp)] if a_ != b_ else [("n", a_, sp)]) for a_, b_, sp, w in one], vx.ptr, vy.ptr)
        assert np.abs(vy.download() - a).max() <= 1e-13
    vx.free(), vy.free()
    A.destroy()


def test_general_term_list_refuses_terms_that_leave_different_sectors():
    from quantum_basis_amd import _lib
    A = q.csr_mat.hubbard(4, 2, 2, lattices.chain(4))
    v = q.DeviceVec(A, 2 * A.dim)
    with pytest.raises(_lib.QbhError) as e:
        q.moprXvec_terms("fermion", 4, 2, 2, [(1.0, [("c", 0, 0)]), (1.0, [("c", 0, 1)])], v.at(0), v.at(A.dim))
    assert e.value.code == -1
    v.free()
    A.destroy()
