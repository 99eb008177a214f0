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
