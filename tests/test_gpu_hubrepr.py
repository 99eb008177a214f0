"""qbh_gen_hubbard_repr: the Hubbard family in translation-symmetric (momentum) sectors, assembled on the device --
counterpart of model::enumerate_basis_repr + generate_Ham_sparse_repr (src/model.cc:687-836) on the path of
examples/trans_symmetric/latt_square/square_Fermi_Hubbard.cc.

Pinned three ways:
  * entry by entry against an explicit projection in numpy: the full fermionic operator (operator order all-up then
    all-down, hop signs from the occupied sites in between), the translation operators WITH their fermion signs, and
    the momentum states |a,k> = (|G||S_a|)^(-1/2) sum_g chi_k(g) T_g |a> built from them;
  * against the reference's own asserted numbers: the eight sector ground-state energies of the 4x2 torus with 4+4
    electrons (square_Fermi_Hubbard.cc:146-153) and <c+_{1,up} c_{5,up}> = 0.3957690742 in the k = (0,0) ground state
    (:182, through the translation-averaged operator of model::measure_repr_static, src/model.cc:1874-1888);
  * row shards of a sector against the whole.
"""
import itertools

import numpy as np
import pytest

import quantum_basis_amd as q
from quantum_basis_amd import lattices

pytestmark = pytest.mark.gpu


def _words(n, nu, nd):
    """all (u, d) occupation words u | d << n, ascending"""
    ups = [sum(1 << i for i in c) for c in itertools.combinations(range(n), nu)]
    dns = [sum(1 << i for i in c) for c in itertools.combinations(range(n), nd)]
    return sorted(u | (d << n) for u in ups for d in dns)


def _hop(occ, i, j):
    """c+_i c_j on one species' occupation word: (sign, new word) or None"""
    if not (occ >> j) & 1 or ((occ >> i) & 1 and i != j):
        return None
    if i == j:
        return 1, occ
    lo, hi = min(i, j), max(i, j)
    between = ((1 << hi) - 1) & ~((2 << lo) - 1)
    return (-1) ** bin(occ & between).count("1"), occ ^ (1 << i) ^ (1 << j)


def _full_operator(n, words, index, terms, U):
    """dense matrix O[b][a] of sum_t amp_up c+_i c_j (up) + amp_dn c+_i c_j (dn) + U sum n_up n_dn"""
    m = (1 << n) - 1
    O = np.zeros((len(words), len(words)), dtype=np.complex128)
    for a, w in enumerate(words):
        u, d = w & m, w >> n
        O[a, a] += U * bin(u & d).count("1")
        for (i, j, au, ad) in terms:
            if au != 0:
                r = _hop(u, i, j)
                if r:
                    O[index[r[1] | (d << n)], a] += au * r[0]
            if ad != 0:
                r = _hop(d, i, j)
                if r:
                    O[index[u | (r[1] << n)], a] += ad * r[0]
    return O


def _image(n, perm, occ):
    """image word and the parity of sorting the images of the occupied sites"""
    imgs = [perm[i] for i in range(n) if (occ >> i) & 1]
    inv = sum(1 for x in range(len(imgs)) for y in range(x + 1, len(imgs)) if imgs[x] > imgs[y])
    return sum(1 << p for p in imgs), (-1) ** inv


def _translation(n, words, index, perm):
    m = (1 << n) - 1
    T = np.zeros((len(words), len(words)))
    for a, w in enumerate(words):
        iu, su = _image(n, perm, w & m)
        idn, sd = _image(n, perm, w >> n)
        T[index[iu | (idn << n)], a] = su * sd
    return T


def _sector_reference(n, nu, nd, terms, U, perms, chars, fake_pos=100.0):
    """(representatives ascending, explicit sector matrix with the generator's conventions)"""
    words = _words(n, nu, nd)
    index = {w: i for i, w in enumerate(words)}
    O = _full_operator(n, words, index, terms, U)
    Ts = [_translation(n, words, index, p) for p in perms]
    m = (1 << n) - 1
    reps = []
    for w in words:
        imgs = [_image(n, p, w & m)[0] | (_image(n, p, w >> n)[0] << n) for p in perms]
        if min(imgs) == w:
            reps.append(w)
    P = sum(c * T for c, T in zip(chars, Ts)) / len(perms)
    dim = len(reps)
    psi = np.zeros((len(words), dim), dtype=np.complex128)
    alive = np.zeros(dim, dtype=bool)
    for r, w in enumerate(reps):
        v = P[:, index[w]]
        nv = np.linalg.norm(v)
        if nv > 1e-10:
            psi[:, r] = v / nv
            alive[r] = True
    Hk = psi.conj().T @ O @ psi
    for r in range(dim):
        if not alive[r]:
            Hk[r, :] = 0
            Hk[:, r] = 0
            Hk[r, r] = fake_pos + r / dim
    return reps, alive, Hk


def _dense(A):
    ia, ja, val = A.download()
    dim = A.info().ncols
    M = np.zeros((len(ia) - 1, dim), dtype=np.complex128)
    for r in range(len(ia) - 1):
        M[r, ja[ia[r]:ia[r + 1]]] = val[ia[r]:ia[r + 1]]
    return M


def _hubbard_terms(bonds, t):
    out = []
    for (i, j) in bonds:
        out += [(i, j, -t, -t), (j, i, -t, -t)]
    return out


CASES = [
    # (Lx, Ly, n_up, n_dn): odd and even particle numbers (the sign of a cyclic shift depends on them), a stabilised
    # sector with zero-norm representatives, a two-dimensional torus with the doubled y bond
    (6, 1, 3, 3),
    (6, 1, 2, 3),
    (4, 1, 2, 2),
    (3, 2, 2, 3),
    (4, 2, 2, 1),
    (2, 2, 2, 2),
    (5, 1, 2, 0),          # one species only (spinless fermions)
    (4, 2, 0, 3),
]


@pytest.mark.parametrize("Lx,Ly,nu,nd", CASES)
def test_sector_matrix_matches_explicit_projection(Lx, Ly, nu, nd):
    n = Lx * Ly
    bonds = lattices.chain(Lx) if Ly == 1 else lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    t, U = 1.0, 1.1
    total = 0
    for k in itertools.product(range(Lx), range(Ly)):
        chars = lattices.characters(shifts, k, (Lx, Ly))
        reps, alive, Hk = _sector_reference(n, nu, nd, _hubbard_terms(bonds, t), U, perms, chars)
        A = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, chars, t=t, U=U, opts=q.make_opts(value_dict=0))
        M = _dense(A)
        assert M.shape == Hk.shape, (k, M.shape, Hk.shape)
        assert np.abs(M - Hk).max() < 1e-12, (k, np.abs(M - Hk).max())
        assert np.abs(M - M.conj().T).max() < 1e-13
        total += int(alive.sum())
        B = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, chars, t=t, U=U)        # value codes emitted by the generator
        assert B.info().value_dict > 0 and np.array_equal(_dense(B), M)
        B.destroy()
        A.destroy()
    # the momentum sectors together span the whole fixed-particle-number space
    assert total == len(_words(n, nu, nd))


def test_non_hermitian_one_body_operator():
    """a directed, translation-averaged hop (what measure_repr_static builds) is filled as O[a][b], not as its adjoint"""
    Lx, Ly, nu, nd = 4, 2, 2, 2
    n = Lx * Ly
    perms, shifts = lattices.translations(Lx, Ly)
    terms = [(p[1], p[4], 1.0 / len(perms) * (1 + 0.5j), 0.25 / len(perms)) for p in perms]
    for k in [(0, 0), (1, 0), (2, 1)]:
        chars = lattices.characters(shifts, k, (Lx, Ly))
        reps, alive, Ok = _sector_reference(n, nu, nd, terms, 0.0, perms, chars, fake_pos=7.0)
        A = q.csr_mat.hubbard_repr(n, nu, nd, None, perms, chars, U=0.0, terms=terms, fake_pos=7.0, opts=q.make_opts(value_dict=0))
        M = _dense(A)
        assert np.abs(M - Ok).max() < 1e-12
        A.destroy()


def test_reference_asserted_sector_energies_and_correlator():
    """examples/trans_symmetric/latt_square/square_Fermi_Hubbard.cc:146-153 and :182"""
    Lx, Ly, nu, nd = 4, 2, 4, 4
    n = Lx * Ly
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    want = {(0, 0): -14.07605866, (0, 1): -10.50470669, (1, 0): -12.16861094, (1, 1): -12.19847764,
            (2, 0): -10.54300366, (2, 1): -14.03137587, (3, 0): -12.16861094, (3, 1): -12.19847764}
    dims = 0
    for k, e_ref in want.items():
        chars = lattices.characters(shifts, k, (Lx, Ly))
        A = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, chars, t=1.0, U=1.1)
        dim = A.info().ncols
        M = _dense(A)
        ev = np.linalg.eigvalsh(M)
        assert abs(ev[0] - e_ref) < 1e-8, (k, ev[0], e_ref)
        dims += int(np.sum(np.abs(np.diag(M)) < 50.0))
        if k == (0, 0):
            w, vec = np.linalg.eigh(M)
            psi = np.ascontiguousarray(vec[:, 0])
            # O_t = (1/N) sum_R c+_{R(1),up} c_{R(5),up}
            terms = [(p[1], p[5], 1.0 / len(perms), 0.0) for p in perms]
            Ot = q.csr_mat.hubbard_repr(n, nu, nd, None, perms, chars, U=0.0, terms=terms, fake_pos=0.0)
            y = np.zeros(dim, dtype=np.complex128)
            Ot.MultMv(psi, y)
            m1 = np.vdot(psi, y)
            assert abs(m1 - 0.3957690742) < 1e-8, m1
            Ot.destroy()
            # and the device Lanczos driver on the sector operator, as locate_E0_lanczos(which_sym::repr) runs it
            e0 = _lanczos_e0(A, dim)
            assert abs(e0 - e_ref) < 1e-8, (e0, e_ref)
        A.destroy()
    assert dims == 4900          # C(8,4)^2 states in all sectors together


def _lanczos_e0(A, dim, maxit=300):
    dv = A.vec(3)
    A.randomize(dv.at(0), 7)
    hess = np.zeros(2 * maxit)
    m = q.lanczos(0, maxit - 1, maxit, dim, A, None, hess, "sr_val0", device_v=dv)
    ritz, _ = q.hess_eigen(hess, maxit, m, "sr")
    dv.free()
    return ritz[0]


def test_row_shards_of_a_sector():
    Lx, Ly, nu, nd = 4, 2, 3, 2
    n = Lx * Ly
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    chars = lattices.characters(shifts, (1, 1), (Lx, Ly))
    A = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, chars, opts=q.make_opts(value_dict=0))
    whole = _dense(A)
    parts = [q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, chars, shard=(r, 3), opts=q.make_opts(value_dict=0)) for r in range(3)]
    stacked = np.vstack([_dense(p) for p in parts])
    assert stacked.shape == whole.shape and np.abs(stacked - whole).max() == 0.0
    for p in parts:
        p.destroy()
    A.destroy()


def test_rejects_bad_arguments():
    perms, shifts = lattices.translations(4, 1)
    chars = lattices.characters(shifts, (0, 0), (4, 1))
    with pytest.raises(q._lib.QbhError):
        q.csr_mat.hubbard_repr(4, 2, 2, [(0, 7)], perms, chars)                       # site outside the lattice
    bad = [list(p) for p in perms]
    bad[1][0] = bad[1][1]
    with pytest.raises(q._lib.QbhError):
        q.csr_mat.hubbard_repr(4, 2, 2, lattices.chain(4), bad, chars)                # not a permutation


def test_c3_ground_state_through_its_momentum_sectors_full_size():
    """BASELINE configs[2] (Hubbard 4x4, half filling, dim 165,636,900) reached the way the reference reaches larger
    clusters: through translation-symmetric sectors (16 x smaller: 10,353,252 representatives each).  Cross-path pin at
    full size: the k = (0,0) sector reproduces the ground-state energy of the full-basis operator (matrix-free,
    packed-double Lanczos) to 1e-11 relative, no sector lies below it, and momenta related by the lattice's symmetries
    are degenerate."""
    Lx = Ly = 4
    n = 16
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    full = q.csr_mat.hubbard(n, 8, 8, bonds, t=1.0, U=1.1, matrix_free=True)
    maxit = 600
    v = full.vec(1)
    q._lib.check(q._lib.lib().qbh_vec_randomize_real(full.handle, v.ptr, 5), "qbh_vec_randomize_real")
    h = np.zeros(2 * maxit)
    m = q.lanczos_real(0, maxit - 1, maxit, full, v, h)
    e_full = q.hess_eigen(h, maxit, m, "sr")[0][0]
    v.free()
    full.destroy()
    e = {}
    for k in [(0, 0), (1, 0), (0, 1), (3, 0), (2, 0), (1, 1), (2, 1), (2, 2)]:
        chars = lattices.characters(shifts, k, (Lx, Ly))
        A = q.csr_mat.hubbard_repr(n, 8, 8, bonds, perms, chars, t=1.0, U=1.1)
        i = A.info()
        assert i.ncols == 10353252 and i.nrows == i.ncols
        e[k] = _lanczos_e0(A, i.ncols, maxit=800)
        A.destroy()
    assert abs(e[(0, 0)] - e_full) < 1e-11 * abs(e_full), (e[(0, 0)], e_full)
    assert all(x > e_full - 1e-9 for x in e.values()), e
    assert abs(e[(1, 0)] - e[(0, 1)]) < 1e-9 and abs(e[(1, 0)] - e[(3, 0)]) < 1e-9      # x <-> y, k <-> -k
    assert abs(e[(2, 0)] - e[(1, 1)]) < 1e-9                                              # the 4x4 torus is the hypercube


def test_c4_substitute_ground_state_through_its_momentum_sectors_full_size():
    """Hubbard 4x5 with 5+5 electrons (the C4 substitute of test_gpu_configs, dim 240,374,016): the lowest of the nine
    inequivalent momentum sectors (12,018,806 representatives each) is the full-basis ground state (matrix-free,
    packed-double Lanczos), and no sector lies below it."""
    Lx, Ly, n = 4, 5, 20
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    full = q.csr_mat.hubbard(n, 5, 5, bonds, t=1.0, U=1.1, matrix_free=True)
    maxit = 800
    v = full.vec(1)
    q._lib.check(q._lib.lib().qbh_vec_randomize_real(full.handle, v.ptr, 5), "qbh_vec_randomize_real")
    h = np.zeros(2 * maxit)
    m = q.lanczos_real(0, maxit - 1, maxit, full, v, h)
    e_full = q.hess_eigen(h, maxit, m, "sr")[0][0]
    v.free()
    full.destroy()
    e = {}
    dims = set()
    for k in [(kx, ky) for kx in range(3) for ky in range(3)]:
        chars = lattices.characters(shifts, k, (Lx, Ly))
        A = q.csr_mat.hubbard_repr(n, 5, 5, bonds, perms, chars, t=1.0, U=1.1)
        dims.add(A.info().ncols)
        e[k] = _lanczos_e0(A, A.info().ncols, maxit=800)
        A.destroy()
    assert len(dims) == 1                                  # all representatives are kept in every sector
    lowest = min(e.values())
    assert abs(lowest - e_full) < 1e-11 * abs(e_full), (e, e_full)
    assert all(x > e_full - 1e-9 for x in e.values())


def test_density_and_sz_fourier_components_between_sectors():
    """qbh_mopr_diag_hubrepr_dev = moprXvec_repr for N_q and S^z_q: against the explicit matrix <b,k+q| O |a,k> built from
    the projected momentum states."""
    Lx, Ly, nu, nd = 4, 2, 3, 2
    n = Lx * Ly
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    terms = _hubbard_terms(bonds, 1.0)
    words = _words(n, nu, nd)
    index = {w: i for i, w in enumerate(words)}
    m = (1 << n) - 1
    rng = np.random.default_rng(5)
    for k_old, qv, spin in [((0, 0), (1, 0), False), ((1, 1), (2, 1), True), ((3, 0), (1, 1), False), ((2, 0), (0, 0), True)]:
        k_new = ((k_old[0] + qv[0]) % Lx, (k_old[1] + qv[1]) % Ly)
        ch_old = lattices.characters(shifts, k_old, (Lx, Ly))
        ch_new = lattices.characters(shifts, k_new, (Lx, Ly))
        # e^{-i q.r_s}: translating the site by t multiplies the coefficient by chi_q(t), so the target character is chi_k * chi_q
        phase = np.array([np.exp(-2j * np.pi * (qv[0] * (s % Lx) / Lx + qv[1] * (s // Lx) / Ly)) for s in range(n)])
        cu, cd = (0.5 * phase, -0.5 * phase) if spin else (phase, phase)
        O = np.zeros((len(words), len(words)), dtype=np.complex128)
        for a, w in enumerate(words):
            u, d = w & m, w >> n
            O[a, a] = sum(cu[s] for s in range(n) if (u >> s) & 1) + sum(cd[s] for s in range(n) if (d >> s) & 1)

        def states(chars):
            Ts = [_translation(n, words, index, p) for p in perms]
            P = sum(c * T for c, T in zip(chars, Ts)) / len(perms)
            reps = [w for w in words if min(_image(n, p, w & m)[0] | (_image(n, p, w >> n)[0] << n) for p in perms) == w]
            psi = np.zeros((len(words), len(reps)), dtype=np.complex128)
            for r, w in enumerate(reps):
                v = P[:, index[w]]
                if np.linalg.norm(v) > 1e-10:
                    psi[:, r] = v / np.linalg.norm(v)
            return psi
        psi_old, psi_new = states(ch_old), states(ch_new)
        Okk = psi_new.conj().T @ O @ psi_old
        dim = psi_old.shape[1]
        x = rng.standard_normal(dim) + 1j * rng.standard_normal(dim)
        x[np.abs(psi_old).sum(axis=0) == 0] = 0.0          # no weight on zero-norm representatives of the old sector
        A = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, ch_new)
        assert A.info().ncols == dim
        dv = A.vec(2)
        dv.upload(x, 0)
        got_dim = q.moprXvec_diag_hubrepr(n, nu, nd, perms, ch_new, cu, cd, dv.at(0), dv.at(dim))
        assert got_dim == dim
        y = dv.download(dim, dim)
        assert np.abs(y - Okk @ x).max() < 1e-12, (k_old, qv, np.abs(y - Okk @ x).max())
        dv.free()
        A.destroy()
    # coefficients that do not transform with a character are refused
    A = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, lattices.characters(shifts, (0, 0), (Lx, Ly)))
    dv = A.vec(2)
    bad = np.arange(1, n + 1).astype(np.complex128)
    with pytest.raises(q._lib.QbhError):
        q.moprXvec_diag_hubrepr(n, nu, nd, perms, lattices.characters(shifts, (1, 0), (Lx, Ly)), bad, bad, dv.at(0), dv.at(A.dim))
    dv.free()
    A.destroy()


def test_density_structure_factor_in_sectors_end_to_end():
    """measure_repr_dynamic (src/model.cc:1896-1935) for the Hubbard family: ground state of the k = (0,0) sector of the 4x2
    torus (4+4 electrons, the reference's example), N_q |psi0> on the device, lanczos("dnmcs") on the k + q sector operator;
    the continued fraction of the coefficients against the resolvent from a dense diagonalisation of that sector."""
    Lx, Ly, nu, nd = 4, 2, 4, 4
    n = Lx * Ly
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    ch0 = lattices.characters(shifts, (0, 0), (Lx, Ly))
    A0 = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, ch0)
    M0 = _dense(A0)
    w0, v0 = np.linalg.eigh(M0)
    assert abs(w0[0] + 14.07605866) < 1e-8
    psi0 = np.ascontiguousarray(v0[:, 0])
    dim = A0.info().ncols
    vphi = q.DeviceVec(A0, dim)
    vphi.upload(psi0)
    maxit = 200
    for qv in [(1, 0), (2, 1)]:
        chq = lattices.characters(shifts, qv, (Lx, Ly))
        phase = np.array([np.exp(-2j * np.pi * (qv[0] * (s % Lx) / Lx + qv[1] * (s // Lx) / Ly)) for s in range(n)]) / np.sqrt(n)
        B = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, chq)
        m, norm, hess = q.measure_full_dynamic_dev(
            B, lambda dst: q.moprXvec_diag_hubrepr(n, nu, nd, perms, chq, phase, phase, vphi.ptr, dst), maxit)
        Mq = _dense(B)
        wq, vq = np.linalg.eigh(Mq)
        # N_q psi0 in the new sector's basis: diagonal on the shared representative list
        tmp = q.DeviceVec(B, dim)
        q.moprXvec_diag_hubrepr(n, nu, nd, perms, chq, phase, phase, vphi.ptr, tmp.ptr)
        phi = tmp.download(0, dim)
        tmp.free()
        assert abs(np.linalg.norm(phi) - norm) < 1e-12
        weights = np.abs(vq.conj().T @ phi) ** 2
        assert weights[wq > 50.0].sum() < 1e-24          # nothing on the decoupled zero-norm rows

        def green(h, mm, z):
            g = 0.0
            for j in range(mm - 1, -1, -1):
                g = 1.0 / (z - h[maxit + j] - (h[j + 1] ** 2) * g)
            return g

        for w in np.linspace(0.0, 6.0, 13):
            z = w0[0] + w + 0.1j
            g_dev = norm ** 2 * green(hess, m - 1, z)
            g_ref = np.sum(weights / (z - wq))
            assert abs(g_dev - g_ref) <= 1e-8 * max(abs(g_ref), 1e-3), (qv, w, g_dev, g_ref)
        B.destroy()
    vphi.free()
    A0.destroy()


def _opstring(w, n):
    """the word's operator string: all up (ascending site), then all down"""
    return [(0, i) for i in range(n) if (w >> i) & 1] + [(1, i) for i in range(n) if (w >> (n + i)) & 1]


def _apply_fermion(w, n, s, sp, create):
    """c_{s,sp} / c^dag_{s,sp} on the word by anticommuting through its operator string: (sign, new word) or None"""
    ops = _opstring(w, n)
    if create:
        if (sp, s) in ops:
            return None
        p = sum(1 for o in ops if o < (sp, s))
        return (-1) ** p, w | (1 << (s + n * sp))
    if (sp, s) not in ops:
        return None
    return (-1) ** ops.index((sp, s)), w & ~(1 << (s + n * sp))


def _momentum_states(n, nu, nd, perms, chars):
    words = _words(n, nu, nd)
    index = {w: i for i, w in enumerate(words)}
    m = (1 << n) - 1
    Ts = [_translation(n, words, index, p) for p in perms]
    P = sum(c * T for c, T in zip(chars, Ts)) / len(perms)
    reps = [w for w in words if min(_image(n, p, w & m)[0] | (_image(n, p, w >> n)[0] << n) for p in perms) == w]
    psi = np.zeros((len(words), len(reps)), dtype=np.complex128)
    for r, w in enumerate(reps):
        v = P[:, index[w]]
        if np.linalg.norm(v) > 1e-10:
            psi[:, r] = v / np.linalg.norm(v)
    return words, index, psi


def test_single_fermion_operators_between_sectors():
    """qbh_mopr_c_hubrepr_dev (c_q / c^dag_q, both species) against the explicit matrix <b, k+q; N -/+ 1| O |a, k; N> built by
    anticommuting through operator strings and projecting onto momentum states."""
    Lx, Ly, nu, nd = 4, 2, 3, 2
    n = Lx * Ly
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    rng = np.random.default_rng(11)
    for species, kind, k_old, qv in [(0, -1, (0, 0), (1, 0)), (1, -1, (1, 1), (2, 1)), (0, +1, (3, 0), (1, 1)), (1, +1, (2, 1), (0, 0)),
                                     (0, -1, (2, 0), (2, 0))]:
        k_new = ((k_old[0] + qv[0]) % Lx, (k_old[1] + qv[1]) % Ly)
        ch_old = lattices.characters(shifts, k_old, (Lx, Ly))
        ch_new = lattices.characters(shifts, k_new, (Lx, Ly))
        coef = np.array([np.exp(-2j * np.pi * (qv[0] * (s % Lx) / Lx + qv[1] * (s // Lx) / Ly)) for s in range(n)]) / np.sqrt(n)
        nu2, nd2 = nu + (kind if species == 0 else 0), nd + (kind if species == 1 else 0)
        w_old, i_old, psi_old = _momentum_states(n, nu, nd, perms, ch_old)
        w_new, i_new, psi_new = _momentum_states(n, nu2, nd2, perms, ch_new)
        O = np.zeros((len(w_new), len(w_old)), dtype=np.complex128)
        for a, w in enumerate(w_old):
            for s_ in range(n):
                r = _apply_fermion(w, n, s_, species, kind > 0)
                if r:
                    O[i_new[r[1]], a] += coef[s_] * r[0]
        Okk = psi_new.conj().T @ O @ psi_old
        d_old, d_new = psi_old.shape[1], psi_new.shape[1]
        x = rng.standard_normal(d_old) + 1j * rng.standard_normal(d_old)
        x[np.abs(psi_old).sum(axis=0) == 0] = 0.0
        A = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, ch_old)
        B = q.csr_mat.hubbard_repr(n, nu2, nd2, bonds, perms, ch_new)
        assert A.info().ncols == d_old and B.info().ncols == d_new
        vx, vy = q.DeviceVec(A, d_old), q.DeviceVec(B, d_new)
        vx.upload(x)
        got = q.moprXvec_c_hubrepr(n, nu, nd, species, kind, perms, ch_old, ch_new, coef, vx.ptr, vy.ptr)
        assert got == (d_old, d_new)
        y = vy.download(0, d_new)
        assert np.abs(y - Okk @ x).max() < 1e-12, (species, kind, k_old, qv, np.abs(y - Okk @ x).max())
        vx.free()
        vy.free()
        A.destroy()
        B.destroy()


def test_photoemission_sum_rule_in_sectors():
    """sum_q | c_{q,up} |psi0> |^2 = N_up for the ground state of the reference's 4x2 example (k = (0,0), 4+4 electrons)"""
    Lx, Ly, nu, nd = 4, 2, 4, 4
    n = Lx * Ly
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    ch0 = lattices.characters(shifts, (0, 0), (Lx, Ly))
    A0 = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, ch0)
    w0, v0 = np.linalg.eigh(_dense(A0))
    dim = A0.info().ncols
    vphi = q.DeviceVec(A0, dim)
    vphi.upload(np.ascontiguousarray(v0[:, 0]))
    total = {0: 0.0, 1: 0.0}
    for species in (0, 1):
        for qv in [(kx, ky) for kx in range(Lx) for ky in range(Ly)]:
            chq = lattices.characters(shifts, qv, (Lx, Ly))
            coef = np.array([np.exp(-2j * np.pi * (qv[0] * (s % Lx) / Lx + qv[1] * (s // Lx) / Ly)) for s in range(n)]) / np.sqrt(n)
            B = q.csr_mat.hubbard_repr(n, nu - (species == 0), nd - (species == 1), bonds, perms, chq)
            vy = q.DeviceVec(B, B.info().ncols)
            q.moprXvec_c_hubrepr(n, nu, nd, species, -1, perms, ch0, chq, coef, vphi.ptr, vy.ptr)
            total[species] += B.nrm2(vy.ptr) ** 2
            vy.free()
            B.destroy()
    assert abs(total[0] - nu) < 1e-10 and abs(total[1] - nd) < 1e-10, total
    vphi.free()
    A0.destroy()


def test_photoemission_sum_rule_full_size_c3():
    """The same chain at full size: ground state of C3 in its k = (0,0) sector (10,353,252 representatives; Lanczos + CG on
    the device), c_{q,up}|psi0> into each of the 16 (7 up, 8 down) sectors (9,202,050 representatives):
    sum_q |c_q psi0|^2 = N_up = 8, momenta related by the lattice symmetries carry equal weight, and the momentum
    distribution is exactly 1/2 on the Fermi surface |kx| + |ky| = pi (particle-hole symmetry of the half-filled bipartite
    cluster)."""
    Lx = Ly = 4
    n, nu, nd = 16, 8, 8
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    ch0 = lattices.characters(shifts, (0, 0), (Lx, Ly))
    A0 = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, ch0)
    res = q.locate_E0_lanczos(A0, nev=1, ncv=1)
    assert abs(res.E0 + 20.497352266554) < 1e-9
    dim0 = A0.info().ncols
    vphi = q.DeviceVec(A0, dim0)
    vphi.upload(res.eigenvecs)
    wt = {}
    for qv in [(kx, ky) for kx in range(Lx) for ky in range(Ly)]:
        chq = lattices.characters(shifts, qv, (Lx, Ly))
        coef = np.array([np.exp(-2j * np.pi * (qv[0] * (s % Lx) / Lx + qv[1] * (s // Lx) / Ly)) for s in range(n)]) / np.sqrt(n)
        vy = q.DeviceVec(A0, dim0)                       # at least as long as the smaller target sector
        d_old, d_new = q.moprXvec_c_hubrepr(n, nu, nd, 0, -1, perms, ch0, chq, coef, vphi.ptr, vy.ptr)
        assert d_old == dim0 and d_new == 9202050
        y = vy.download(0, d_new)
        wt[qv] = float(np.vdot(y, y).real)
        vy.free()
    assert abs(sum(wt.values()) - nu) < 1e-9, sum(wt.values())
    assert abs(wt[(1, 0)] - wt[(0, 1)]) < 1e-9 and abs(wt[(1, 0)] - wt[(3, 0)]) < 1e-9 and abs(wt[(1, 2)] - wt[(2, 1)]) < 1e-9
    for qv in [(2, 0), (0, 2), (1, 1), (1, 3), (3, 1), (3, 3)]:
        assert abs(wt[qv] - 0.5) < 1e-9, (qv, wt[qv])
    assert abs(wt[(0, 0)] + wt[(2, 2)] - 1.0) < 1e-9          # n(k) + n(k + (pi,pi)) = 1
    vphi.free()
    A0.destroy()


def _add_pairs(O, n, words, pairs):
    m = (1 << n) - 1
    for a, w in enumerate(words):
        u, d = w & m, w >> n
        for (i, j, vuu, vud, vdu, vdd) in pairs:
            iu, idn, ju, jd = (u >> i) & 1, (d >> i) & 1, (u >> j) & 1, (d >> j) & 1
            O[a, a] += vuu * iu * ju + vud * iu * jd + vdu * idn * ju + vdd * idn * jd
    return O


def test_extended_hubbard_density_density_terms():
    """density-density terms v n_{i,s} n_{j,s'} (extended Hubbard) against the explicit projection"""
    Lx, Ly, nu, nd = 3, 2, 2, 2
    n = Lx * Ly
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    pairs = [(i, j, 0.7, 0.3, 0.3, 0.7) for (i, j) in bonds]
    terms = _hubbard_terms(bonds, 1.0)
    words = _words(n, nu, nd)
    index = {w: i for i, w in enumerate(words)}
    for k in [(0, 0), (1, 0), (2, 1)]:
        chars = lattices.characters(shifts, k, (Lx, Ly))
        O = _add_pairs(_full_operator(n, words, index, terms, 1.1), n, words, pairs)
        _, _, psi = _momentum_states(n, nu, nd, perms, chars)
        Hk = psi.conj().T @ O @ psi
        dead = np.abs(psi).sum(axis=0) == 0
        A = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, chars, t=1.0, U=1.1, pairs=pairs, opts=q.make_opts(value_dict=0))
        M = _dense(A)
        for r in np.nonzero(dead)[0]:
            Hk[r, r] = M[r, r]                             # the fake diagonal of a zero-norm representative
        assert np.abs(M - Hk).max() < 1e-12
        A.destroy()


def test_reference_asserted_spinless_fermion_honeycomb_energies():
    """examples/trans_symmetric/latt_honeycomb/honeycomb_Spinless_Fermion.cc:146-151: t-V model (t = 1, V1 = 4) on the 3x2
    honeycomb torus (12 sites, two per cell), 4 fermions, all six momenta -- the generator with one species, a two-site
    unit cell, density-density and number terms."""
    Lx, Ly, t, V1 = 3, 2, 1.0, 4.0
    n = 2 * Lx * Ly
    bonds = lattices.honeycomb(Lx, Ly)
    assert len(bonds) == 18 and len(set(bonds)) == 18
    perms, shifts = lattices.translations(Lx, Ly, n_sub=2)
    terms, pairs = [], []
    for (i, j) in bonds:
        terms += [(i, j, -t, 0.0), (j, i, -t, 0.0), (i, i, -0.5 * V1, 0.0), (j, j, -0.5 * V1, 0.0)]
        pairs.append((i, j, V1, 0.0, 0.0, 0.0))
    want = [-28.60363167, -28.27163215, -28.60363167, -28.27163215, -28.60363167, -28.27163215]
    got = []
    for m in range(Lx):
        for nn in range(Ly):
            chars = lattices.characters(shifts, (m, nn), (Lx, Ly))
            A = q.csr_mat.hubbard_repr(n, 4, 0, None, perms, chars, U=0.0, terms=terms, pairs=pairs)     # N = Lx*Ly - 2 (:17)
            got.append(np.linalg.eigvalsh(_dense(A))[0])
            if (m, nn) == (0, 0):
                assert abs(_lanczos_e0(A, A.info().ncols) - want[0]) < 1e-8      # and through the device Lanczos driver
            A.destroy()
    assert np.abs(np.array(got) - np.array(want)).max() < 1e-8, got
    # with the trivial group the "sector" is the full fixed-N basis: examples/trans_absent/latt_honeycomb/...:129
    A = q.csr_mat.hubbard_repr(n, 4, 0, None, [list(range(n))], [1.0], U=0.0, terms=terms, pairs=pairs)
    assert A.info().ncols == 495 and abs(np.linalg.eigvalsh(_dense(A))[0] + 28.60363167) < 1e-8
    A.destroy()


def _exchange_operator(n, words, index, exch):
    """xa (S+_i S-_j + S-_i S+_j) by four successive fermion operators on the operator string"""
    O = np.zeros((len(words), len(words)), dtype=np.complex128)
    for a, w in enumerate(words):
        for (i, j, xa) in exch:
            for (p, m_) in ((i, j), (j, i)):                # S+_p S-_m = c+_{p,up} c_{p,dn} c+_{m,dn} c_{m,up}
                sign, cur = 1, w
                for (site, sp, create) in ((m_, 0, False), (m_, 1, True), (p, 1, False), (p, 0, True)):
                    r = _apply_fermion(cur, n, site, sp, create)
                    if r is None:
                        cur = None
                        break
                    sign *= r[0]
                    cur = r[1]
                if cur is not None and cur in index:
                    O[index[cur], a] += xa * sign
    return O


def test_spin_exchange_terms_and_the_tj_constraint_against_explicit_projection():
    rng = np.random.default_rng(3)
    for (Lx, Ly, nu, nd, no_double) in [(3, 2, 2, 2, False), (6, 1, 3, 2, False), (6, 1, 2, 2, True), (3, 2, 2, 1, True), (4, 2, 3, 3, True)]:
        n = Lx * Ly
        bonds = lattices.chain(Lx) if Ly == 1 else lattices.square(Lx, Ly)
        perms, shifts = lattices.translations(Lx, Ly)
        exch = [(i, j, 0.5 * (1.0 + 0.1 * b)) for b, (i, j) in enumerate(bonds[:1])] * 0 + [(i, j, 0.35) for (i, j) in bonds]
        pairs = [(i, j, 0.0, -0.25, -0.25, 0.0) for (i, j) in bonds]
        terms = _hubbard_terms(bonds, 1.0)
        words = [w for w in _words(n, nu, nd) if not (no_double and (w & ((1 << n) - 1)) & (w >> n))]
        index = {w: i for i, w in enumerate(words)}
        m = (1 << n) - 1
        # the hopping restricted to the (possibly constrained) space
        O = np.zeros((len(words), len(words)), dtype=np.complex128)
        for a, w in enumerate(words):
            u, d = w & m, w >> n
            O[a, a] += (0.0 if no_double else 0.9) * bin(u & d).count("1")
            for (i, j, au, ad) in terms:
                r = _hop(u, i, j)
                if r and (r[1] | (d << n)) in index:
                    O[index[r[1] | (d << n)], a] += au * r[0]
                r = _hop(d, i, j)
                if r and (u | (r[1] << n)) in index:
                    O[index[u | (r[1] << n)], a] += ad * r[0]
        O = _add_pairs(O, n, words, pairs) + _exchange_operator(n, words, index, exch)
        assert np.abs(O - O.conj().T).max() < 1e-14
        for k in [(0, 0), (1, 0), (Lx - 1, Ly - 1)]:
            chars = lattices.characters(shifts, k, (Lx, Ly))
            Ts = [_translation(n, words, index, p) for p in perms]
            P = sum(c * T for c, T in zip(chars, Ts)) / len(perms)
            reps = [w for w in words if min(_image(n, p, w & m)[0] | (_image(n, p, w >> n)[0] << n) for p in perms) == w]
            psi = np.zeros((len(words), len(reps)), dtype=np.complex128)
            for r, w in enumerate(reps):
                v = P[:, index[w]]
                if np.linalg.norm(v) > 1e-10:
                    psi[:, r] = v / np.linalg.norm(v)
            Hk = psi.conj().T @ O @ psi
            A = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, chars, t=1.0, U=0.0 if no_double else 0.9, pairs=pairs, exchange=exch,
                                       no_double=no_double, opts=q.make_opts(value_dict=0))
            M = _dense(A)
            assert M.shape == Hk.shape, (Lx, Ly, nu, nd, no_double, k, M.shape, Hk.shape)
            for r in np.nonzero(np.abs(psi).sum(axis=0) == 0)[0]:
                Hk[r, r] = M[r, r]
            assert np.abs(M - Hk).max() < 1e-12, (Lx, Ly, nu, nd, no_double, k, np.abs(M - Hk).max())
            A.destroy()


def test_reference_asserted_tj_energies():
    """examples/trans_symmetric/latt_kagome/kagome_tJ.cc:237-240 (2x2 kagome torus, 12 sites, 4 up + 4 down, t = J = 1: the four
    momentum sectors), its trans_absent twin (:232, trivial group) and examples/trans_absent/latt_chain/chain_tJ.cc:100-101
    (L = 12 ring, 4 up + 4 down: doubly degenerate ground state, found here in two momentum sectors)."""
    kb = lattices.kagome(2, 2)
    perms, shifts = lattices.translations(2, 2, n_sub=3)
    want = {(0, 0): -15.41931496, (0, 1): -14.40277723, (1, 0): -14.40277723, (1, 1): -14.40277723}
    total = 0
    for k, e_ref in want.items():
        A = q.csr_mat.tj_repr(12, 4, 4, kb, perms, lattices.characters(shifts, k, (2, 2)), t=1.0, J=1.0)
        e0 = _lanczos_e0(A, A.info().ncols, maxit=600)
        assert abs(e0 - e_ref) < 1e-8, (k, e0)
        ia, ja, val = A.download()
        rows = np.repeat(np.arange(len(ia) - 1), np.diff(ia))
        total += int(np.sum(np.abs(val[rows == ja]) < 50.0))          # representatives with non-zero norm at this momentum
        A.destroy()
    from math import comb
    assert total == comb(12, 4) * comb(8, 4)               # 4 up on 12 sites, 4 down on the remaining 8
    A = q.csr_mat.tj_repr(12, 4, 4, kb, [list(range(12))], [1.0], t=1.0, J=1.0)
    assert A.info().ncols == comb(12, 4) * comb(8, 4)
    assert abs(_lanczos_e0(A, A.info().ncols, maxit=600) + 15.41931496) < 1e-8
    A.destroy()
    # the ring
    cb = lattices.chain(12)
    perms, shifts = lattices.translations(12, 1)
    e = []
    for kx in range(12):
        A = q.csr_mat.tj_repr(12, 4, 4, cb, perms, lattices.characters(shifts, (kx, 0), (12, 1)), t=1.0, J=1.0)
        e.append(_lanczos_e0(A, A.info().ncols, maxit=600))
        A.destroy()
    e = sorted(e)
    assert abs(e[0] + 9.762087307) < 1e-8 and abs(e[1] + 9.762087307) < 1e-8, e[:3]


def test_no_device_memory_is_leaked_by_the_sector_entry_points():
    import torch
    Lx, Ly, nu, nd = 4, 2, 3, 3
    n = Lx * Ly
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    ch0 = lattices.characters(shifts, (0, 0), (Lx, Ly))
    ch1 = lattices.characters(shifts, (1, 0), (Lx, Ly))
    phase = np.array([np.exp(-2j * np.pi * (s % Lx) / Lx) for s in range(n)])

    def cycle():
        A = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, ch0, pairs=[(i, j, 0.1, 0.2, 0.2, 0.1) for (i, j) in bonds],
                                   exchange=[(i, j, 0.3) for (i, j) in bonds])
        B = q.csr_mat.tj_repr(n, nu, nd, bonds, perms, ch1, shard=(1, 2))
        M = q.csr_mat.hubbard_repr_mf(n, nu, nd, bonds, perms, ch1, pairs=[(i, j, 0.1, 0.2, 0.2, 0.1) for (i, j) in bonds])
        w = M.vec(2)
        M.randomize(w.at(0), 2)
        M.spmv(w.at(0), w.at(M.dim), want_red=True)
        w.free()
        M.destroy()
        dim = A.info().ncols
        v = A.vec(2)
        A.randomize(v.at(0), 1)
        q.moprXvec_diag_hubrepr(n, nu, nd, perms, ch1, phase, phase, v.at(0), v.at(dim))
        q.moprXvec_c_hubrepr(n, nu, nd, 0, -1, perms, ch0, ch1, phase, v.at(0), v.at(dim))
        v.free()
        A.destroy()
        B.destroy()
    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(40):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (8 << 20), (free0 - free1)


MF_CASES = [
    # (Lx, Ly, n_up, n_dn): clusters with many stabilised down blocks (4x2), with none (5x1 with 2 down: prime length), two
    # dimensions, pairs of very different species counts
    (4, 2, 4, 4), (4, 2, 3, 2), (5, 1, 2, 2), (6, 1, 3, 3), (3, 2, 2, 3), (4, 3, 3, 2), (4, 2, 1, 0), (4, 2, 0, 2),
]


@pytest.mark.parametrize("Lx,Ly,nu,nd", MF_CASES)
def test_matrix_free_sector_operator_equals_the_stored_one(Lx, Ly, nu, nd):
    """qbh_mf_hubbard_repr (block tables + stored remainder) against qbh_gen_hubbard_repr (stored CSR of the same sector):
    y = alpha H x + beta y + gamma x on random complex vectors for every momentum, and the packed-real Lanczos path."""
    n = Lx * Ly
    bonds = lattices.chain(Lx) if Ly == 1 else lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    pairs = [(i, j, 0.2, 0.1, 0.1, 0.2) for (i, j) in bonds]
    rng = np.random.default_rng(17)
    for k in itertools.product(range(Lx), range(Ly)):
        chars = lattices.characters(shifts, k, (Lx, Ly))
        A = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, chars, t=1.0, U=1.3, pairs=pairs)
        # the two row orders of the matrix-free form: orbit by orbit of the up patterns (default; device vectors in that order,
        # host vectors translated at the seams) and ascending (the rank-table kernel)
        M = q.csr_mat.hubbard_repr_mf(n, nu, nd, bonds, perms, chars, t=1.0, U=1.3, pairs=pairs)
        M0 = q.csr_mat.hubbard_repr_mf(n, nu, nd, bonds, perms, chars, t=1.0, U=1.3, pairs=pairs, opts=q.make_opts(sector_orbit=0))
        dim = A.info().ncols
        assert M.info().ncols == dim and M0.info().ncols == dim
        assert M.info().basis_internal == q._lib.BASIS_SECTOR_ORBIT and M0.info().basis_internal == 0
        x = rng.standard_normal(dim) + 1j * rng.standard_normal(dim)
        y0 = rng.standard_normal(dim) + 1j * rng.standard_normal(dim)
        out = []
        for op in (A, M, M0):
            v = op.vec(2)
            v.upload(x, 0)
            v.upload(y0, dim)
            red = op.spmv(v.at(0), v.at(dim), 0.7, -0.4, 0.25, want_red=True)
            out.append((v.download(dim, dim), red))
            v.free()
        ref = out[0]
        scale = np.abs(ref[0]).max()
        for got in out[1:]:
            assert np.abs(got[0] - ref[0]).max() < 1e-12 * scale, (k, np.abs(got[0] - ref[0]).max())
            assert abs(got[1][0] - ref[1][0]) < 1e-10 * abs(ref[1][0]) and abs(got[1][1] - ref[1][1]) < 1e-10 * ref[1][1]
        # device vectors: the caller's order <-> the operator's own
        v, w = A.vec(2), M.vec(2)
        v.upload(x, 0)
        M.to_internal(w.at(0), v.at(0))
        M.spmv(w.at(0), w.at(dim))
        M.from_internal(v.at(dim), w.at(dim))
        M.sync()                                            # every handle has its own stream
        A.spmv(v.at(0), v.at(dim), 1.0, -1.0, 0.0)          # H x - (H x through the matrix-free form)
        assert A.nrm2(v.at(dim)) < 1e-12 * scale * np.sqrt(dim)
        v.free()
        w.free()
        if dim > 40:
            e_a, e_m, e_0 = _lanczos_e0(A, dim), _lanczos_e0(M, dim), _lanczos_e0(M0, dim)
            assert abs(e_a - e_m) < 1e-10 * max(1.0, abs(e_a)), (k, e_a, e_m)
            assert abs(e_a - e_0) < 1e-10 * max(1.0, abs(e_a)), (k, e_a, e_0)
        A.destroy()
        M.destroy()
        M0.destroy()


def test_matrix_free_sector_operator_with_a_non_abelian_group():
    """The orbit order rests on the group's composition table (position of g(e(u0)) inside the orbit of u0): a ring of 6 sites with
    its DIHEDRAL group (6 rotations + 6 reflections, g1 g2 != g2 g1) in its two one-dimensional representations with real
    characters -- the library must take the orbit form (its own position-by-position check at build passes) and reproduce the
    stored sector operator."""
    L, nu, nd = 6, 3, 2
    bonds = lattices.chain(L)
    rot = [[(s + t) % L for s in range(L)] for t in range(L)]
    ref = [[(t - s) % L for s in range(L)] for t in range(L)]
    perms = rot + ref
    rng = np.random.default_rng(5)
    for chars in ([1.0 + 0j] * (2 * L), [1.0 + 0j] * L + [-1.0 + 0j] * L):
        A = q.csr_mat.hubbard_repr(L, nu, nd, bonds, perms, chars, t=1.0, U=1.3)
        M = q.csr_mat.hubbard_repr_mf(L, nu, nd, bonds, perms, chars, t=1.0, U=1.3)
        dim = A.info().ncols
        assert dim > 0 and M.info().ncols == dim and M.info().basis_internal == q._lib.BASIS_SECTOR_ORBIT
        x = rng.standard_normal(dim) + 1j * rng.standard_normal(dim)
        out = []
        for op in (A, M):
            v = op.vec(2)
            v.upload(x, 0)
            op.spmv(v.at(0), v.at(dim))
            out.append(v.download(dim, dim))
            v.free()
        assert np.abs(out[1] - out[0]).max() < 1e-12 * np.abs(out[0]).max()
        A.destroy()
        M.destroy()


def test_matrix_free_sector_operator_at_scale():
    """4x5 with 6+6 electrons, k = (0,0) (75,117,600 representatives): the matrix-free sector operator (0.3 GB) against the
    stored one (14 GB) -- the same y on a random vector, the same ground-state energy through the packed-double Lanczos
    interface, and a complex sector through the ordinary driver."""
    import ctypes
    Lx, Ly, n = 4, 5, 20
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    for k, real in [((0, 0), True), ((1, 2), False)]:
        chars = lattices.characters(shifts, k, (Lx, Ly))
        A = q.csr_mat.hubbard_repr(n, 6, 6, bonds, perms, chars)
        M = q.csr_mat.hubbard_repr_mf(n, 6, 6, bonds, perms, chars)
        dim = A.info().ncols
        assert M.info().ncols == dim == 75117600
        assert M.info().basis_internal == q._lib.BASIS_SECTOR_ORBIT
        v, w = A.vec(3), M.vec(2)
        A.randomize(v.at(0), 9)
        A.spmv(v.at(0), v.at(dim))
        A.sync()
        M.to_internal(w.at(0), v.at(0))                     # M keeps its device vectors orbit by orbit
        M.spmv(w.at(0), w.at(dim))
        M.from_internal(v.at(2 * dim), w.at(dim))
        M.sync()
        hx = A.nrm2(v.at(dim))
        assert np.sqrt(A.axpy_norm(-1.0, v.at(dim), v.at(2 * dim))) <= 1e-12 * hx
        M.randomize(w.at(0), 9)                             # the same stream by the caller's element number
        M.from_internal(w.at(dim), w.at(0))
        M.sync()
        assert np.sqrt(A.axpy_norm(-1.0, v.at(0), w.at(dim))) <= 1e-14
        v.free()
        w.free()
        e_a = _lanczos_e0(A, dim, maxit=600)
        A.destroy()
        if real:
            maxit = 600
            dv = M.vec(1)
            q._lib.check(q._lib.lib().qbh_vec_randomize_real(M.handle, dv.ptr, ctypes.c_uint32(7)), "qbh_vec_randomize_real")
            hess = np.zeros(2 * maxit)
            m = q.lanczos_real(0, maxit - 1, maxit, M, dv, hess)
            e_m = q.hess_eigen(hess, maxit, m, "sr")[0][0]
            dv.free()
        else:
            e_m = _lanczos_e0(M, dim, maxit=600)
        M.destroy()
        assert abs(e_a - e_m) < 1e-11 * abs(e_a), (k, e_a, e_m)


def test_matrix_free_sector_operator_refuses_what_it_cannot_do():
    perms, shifts = lattices.translations(4, 2)
    chars = lattices.characters(shifts, (0, 0), (4, 2))
    M = q.csr_mat.hubbard_repr_mf(8, 3, 3, lattices.square(4, 2), perms, chars)
    assert M.info().kernel == q._lib.KERNEL_MATRIX_FREE
    with pytest.raises(q._lib.QbhError):
        M.download()                                        # no stored CSR
    with pytest.raises(q._lib.QbhError):                    # complex up-species amplitudes are not covered by the block tables
        q.csr_mat.hubbard_repr_mf(8, 3, 3, None, perms, chars, terms=[(0, 1, 1j, 1.0), (1, 0, -1j, 1.0)])
    with pytest.raises(q._lib.QbhError):
        q.csr_mat.hubbard_repr_mf(25, 3, 3, lattices.square(5, 5), *[lattices.translations(5, 5)[0], lattices.characters(lattices.translations(5, 5)[1], (0, 0), (5, 5))])
    M.destroy()


def test_two_body_diagonal_observables_in_a_sector_against_the_full_basis():
    """model::measure_repr_static (src/model.cc:1860-1891) for two-site diagonal observables: <n_i n_j>, the double occupancy
    and <S^z_i S^z_j> of the Hubbard 4x2 ground state, measured in its momentum sector through the translation-averaged
    operator, against the same expectation values taken from the ground state of the FULL basis (dense, numpy)."""
    Lx, Ly, nu, nd = 4, 2, 4, 4
    n = Lx * Ly
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    words = _words(n, nu, nd)
    index = {w: a for a, w in enumerate(words)}
    terms = []
    for (i, j) in bonds:
        terms += [(i, j, -1.0, -1.0), (j, i, -1.0, -1.0)]
    H = _full_operator(n, words, index, terms, 1.1)
    # the two lowest states of the full basis (a dense eigh of the 4900 x 4900 matrix took 30-50 s of the GPU tier's budget)
    import scipy.sparse as _sp
    import scipy.sparse.linalg as _spl
    w_full, v_full = _spl.eigsh(_sp.csr_matrix(H), k=2, which="SA", tol=1e-13)
    order = np.argsort(w_full)
    w_full, v_full = w_full[order], v_full[:, order]
    assert abs(w_full[0] + 14.07605866) < 1e-8 and w_full[1] - w_full[0] > 1e-3          # unique ground state, k = (0,0)
    p = np.abs(v_full[:, 0]) ** 2
    m = (1 << n) - 1
    up = np.array([[(w & m) >> s & 1 for s in range(n)] for w in words], dtype=np.float64)
    dn = np.array([[(w >> n) >> s & 1 for s in range(n)] for w in words], dtype=np.float64)
    chars = lattices.characters(shifts, (0, 0), (Lx, Ly))
    A = q.csr_mat.hubbard_repr(n, nu, nd, bonds, perms, chars, t=1.0, U=1.1)
    dim = A.info().ncols
    res = q.locate_E0_lanczos(A)                                    # E0 and the eigenvector (CG) on the device
    assert abs(res.E0 - w_full[0]) < 1e-9
    psi = A.vec(1)
    psi.upload(res.eigenvecs, 0)
    for (i, j) in [(0, 1), (0, 5), (2, 7), (3, 3)]:
        nn_full = float(p @ ((up[:, i] + dn[:, i]) * (up[:, j] + dn[:, j])))
        szsz_full = float(p @ (0.25 * (up[:, i] - dn[:, i]) * (up[:, j] - dn[:, j])))
        nn = q.measure_repr_static_hubbard(n, nu, nd, perms, chars, psi.ptr, two_body=[(i, j, 1.0, 1.0, 1.0, 1.0)])
        szsz = q.measure_repr_static_hubbard(n, nu, nd, perms, chars, psi.ptr, two_body=[(i, j, 0.25, -0.25, -0.25, 0.25)])
        assert abs(nn - nn_full) < 1e-8 and abs(nn.imag) < 1e-12, (i, j, nn, nn_full)
        assert abs(szsz - szsz_full) < 1e-8, (i, j, szsz, szsz_full)
    docc_full = float(p @ (up[:, 2] * dn[:, 2]))
    docc = q.measure_repr_static_hubbard(n, nu, nd, perms, chars, psi.ptr, two_body=[(2, 2, 0.0, 1.0, 0.0, 0.0)])
    assert abs(docc - docc_full) < 1e-8 and 0.0 < docc.real < 0.25
    # the reference's own asserted one-body correlator through the same entry point (square_Fermi_Hubbard.cc:182)
    hop = q.measure_repr_static_hubbard(n, nu, nd, perms, chars, psi.ptr, one_body=[(1, 5, 1.0, 0.0)])
    assert abs(hop - 0.3957690742) < 1e-8
    # The same measurement on the eigenvector of the MATRIX-FREE sector operator, which keeps its device vectors orbit by orbit
    # (qbh_opts.sector_orbit, qbh_csr_info.basis_internal != 0): a handle-ordered device vector must be named with its handle
    # (`source`), and then gives the same numbers; handed over bare it is a different vector (nothing used to check).
    M = q.csr_mat.hubbard_repr_mf(n, nu, nd, bonds, perms, chars, t=1.0, U=1.1)
    assert M.info().basis_internal != 0
    maxit = 400
    vm = M.vec(2)
    M.randomize(vm.at(0), 1)
    hess = np.zeros(2 * maxit)
    mm = q.lanczos(0, maxit - 1, maxit, dim, M, None, hess, "sr_val0", device_v=vm)
    e0m = q.hess_eigen(hess, maxit, mm, "sr")[0][0]
    assert abs(e0m - w_full[0]) < 1e-9
    cg = M.vec(4)
    M.randomize(cg.at(0), 1)
    q.eigenvec_CG(dim, maxit, 0, M, e0m, cg.at(0), cg.at(dim), cg.at(2 * dim), cg.at(3 * dim), device=True)
    docc_m = q.measure_repr_static_hubbard(n, nu, nd, perms, chars, cg.at(0), two_body=[(2, 2, 0.0, 1.0, 0.0, 0.0)], source=M)
    hop_m = q.measure_repr_static_hubbard(n, nu, nd, perms, chars, cg.at(0), one_body=[(1, 5, 1.0, 0.0)], source=M)
    assert abs(docc_m - docc_full) < 1e-8 and abs(hop_m - 0.3957690742) < 1e-8
    hop_bare = q.measure_repr_static_hubbard(n, nu, nd, perms, chars, cg.at(0), one_body=[(1, 5, 1.0, 0.0)])
    assert abs(hop_bare - 0.3957690742) > 1e-4                      # the orbit-ordered vector is NOT the generator-ordered one
    vm.free()
    cg.free()
    M.destroy()
    psi.free()
    A.destroy()
