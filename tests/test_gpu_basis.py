"""qbh_opts.basis_kind = QBH_BASIS_REF_FERMION2: host arrays in the REFERENCE's own order (Lin order of src/basis.cc:1144-1190,
operators ordered by site, Hermitian-upper int64 CSR exactly as model::generate_Ham_sparse_full hands over, src/model.cc:649-679)
are held species-major inside the library so that the Kronecker split applies; vectors are permuted / sign-flipped at the seams.
Callers must see the reference's order throughout: every result is compared with the oracle ON THE SAME HOST ARRAYS, with the
internal permutation active."""
import math

import numpy as np
import pytest

import quantum_basis_amd as q
from quantum_basis_amd import _lib
from oracle import qb_oracle as qo

import refham

pytestmark = pytest.mark.gpu
PLAIN = dict(value_dict=0, real_fast_path=0)


def _rand(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.normal(size=n) + 1j * rng.normal(size=n)).astype(np.complex128)


def _hint(n, nu, nd, **kw):
    return q.make_opts(basis_kind=_lib.BASIS_REF_FERMION2, n_sites=n, n_up=nu, n_dn=nd, **kw)


# (the 12-site operator once: its oracle Lanczos + CG on the host is 15 s of the GPU tier's budget per case)
@pytest.mark.parametrize("ly,nu,nd,hinted", [(2, 4, 4, 1), (2, 4, 4, 0), (2, 3, 5, 1), (2, 3, 5, 0), (3, 6, 6, 0)])
def test_reference_ordered_host_arrays_run_on_the_split_with_the_hint_and_without(ly, nu, nd, hinted):
    """hinted = 0: NOTHING is said about the basis -- what the reference's csr_mat(lil_mat&) (src/sparse.cc:202-260) can say.  The
    library looks for a two-species basis of this dimension by itself (qbh_opts.basis_detect; kron_split = 2 lifts the 1e8-nonzero
    threshold for the test), finds one under which the operator has the product structure and runs the same split path."""
    n = 4 * ly
    dim, ia, ja, val, _ = refham.hubbard_csr(4, ly, nu, nd, t=1.0, U=1.1)          # Hermitian-upper, reference order
    O = qo.Csr(dim, ia, ja, val, True)
    A = q.csr_mat(dim, ia, ja, val, sym=True, opts=_hint(n, nu, nd, kron_split=2, **PLAIN) if hinted else q.make_opts(kron_split=2, **PLAIN))
    info = A.info()
    assert info.basis_internal == _lib.BASIS_REF_FERMION2 and info.kron_minor == math.comb(n, nd) and info.kron_inplace == 1
    assert info.basis_detected == (0 if hinted else 1) and info.basis_n_sites == n
    assert math.comb(n, info.basis_n_up) * math.comb(n, info.basis_n_dn) == dim
    assert (info.basis_detect_ms > 0.0) == (not hinted) and info.basis_detect_ms < info.create_ms + 1e-9
    # MultMv / MultMv2 (src/sparse.cc:262-297) through the host seam, caller's order in and out
    x, y0 = _rand(dim, 1), _rand(dim, 2)
    want = O.multmv(x)
    y = np.empty(dim, dtype=np.complex128)
    A.MultMv(x, y)
    assert np.abs(y - want).max() <= 2e-13 * np.abs(want).max()
    y = y0.copy()
    A.MultMv2(x, y)
    assert np.abs(y - (y0 + want)).max() <= 2e-13 * np.abs(want).max()
    # lanczos (src/lanczos.cc:134-266) from the reference's start vector: a_j, b_j and the step count of the oracle
    maxit = 1000
    v = np.zeros(2 * dim, dtype=np.complex128)
    v[:dim] = qo.vec_randomize(dim, 1)
    vo = v.copy()
    hess, hess_o = np.zeros(2 * maxit), np.zeros(2 * maxit)
    m = q.lanczos(0, maxit - 1, maxit, dim, A, v, hess, "sr_val0")
    mo, _, _ = qo.lanczos(0, maxit - 1, maxit, O, vo, hess_o, "sr_val0")
    assert abs(m - mo) <= 1
    assert np.allclose(hess[maxit:maxit + 20], hess_o[maxit:maxit + 20], rtol=1e-9) and np.allclose(hess[1:21], hess_o[1:21], rtol=1e-9)
    for j in (0, 1):                                                  # the two vectors handed back are the oracle's, in the caller's order
        assert abs(abs(np.vdot(v[j * dim:(j + 1) * dim], vo[j * dim:(j + 1) * dim])) - 1.0) < 1e-3      # (rounding differences grow along the recurrence)
    # the device start vector is the same Lehmer stream in the CALLER's element order
    assert np.allclose(q.vec_randomize(A, seed=1), qo.vec_randomize(dim, 1), rtol=1e-13, atol=0)
    # locate_E0_lanczos: E0 and the eigenvector, in the caller's order
    r = q.locate_E0_lanczos(A, nev=1, ncv=1, maxit=1000)
    if dim < 100000:
        ro = qo.locate_E0_lanczos(O, nev=1, ncv=1, maxit=1000)
        assert abs(r.E0 - ro["E0"]) <= 1e-11 * abs(ro["E0"])
        assert abs(abs(np.vdot(r.eigenvecs, ro["eigenvecs"])) - 1.0) < 1e-8
    else:           # 12 sites: the oracle's Lanczos above already gave E0 (its CG on the host is another 8 s of the GPU tier's budget); the eigenvector is checked by its residual
        e0_o = qo.hess_eigen(hess_o, maxit, mo, "sr")[0][0]
        assert abs(r.E0 - e0_o) <= 1e-11 * abs(e0_o)
    assert np.abs(O.multmv(r.eigenvecs) - r.E0 * r.eigenvecs).max() < 1e-7
    # qbh_csr_download gives the CALLER's rows back (through the map: H_caller[r, c] = s_r s_c H_internal[g(r), g(c)]): the full-storage
    # form of the arrays that went in, entry by entry -- and row ranges of it
    fia, fja, fval = A.download()
    rows = np.repeat(np.arange(dim), np.diff(ia))
    off = ja > rows                                                    # the stored strict upper triangle, mirrored (explicit zeros kept)
    er, ec, ev = np.concatenate([rows, ja[off]]), np.concatenate([ja, rows[off]]), np.concatenate([val, np.conj(val[off])])
    order = np.lexsort((ec, er))
    assert np.array_equal(fia, np.concatenate([[0], np.cumsum(np.bincount(er, minlength=dim))]))
    assert np.array_equal(fja, ec[order]) and np.array_equal(fval, ev[order])
    r0, r1 = dim // 3, dim // 3 + 29
    sia, sja, sval = A.download(r0, r1)
    assert np.array_equal(sia, fia[r0:r1 + 1] - fia[r0]) and np.array_equal(sja, fja[fia[r0]:fia[r1]]) and np.array_equal(sval, fval[fia[r0]:fia[r1]])
    A.destroy()


def test_a_hint_that_does_not_describe_the_matrix_changes_nothing():
    """8 sites, 3 up + 5 down.  (5, 3) has the same dimension and the same minor size but sorts the wrong particles into the
    major index: the permuted operator has no product structure, which the device check sees; (4, 4) has another dimension.
    Both leave the operator exactly as given -- unpermuted, unsplit, correct."""
    n, nu, nd = 8, 3, 5
    dim, ia, ja, val, _ = refham.hubbard_csr(4, 2, nu, nd, t=1.0, U=1.1)
    O = qo.Csr(dim, ia, ja, val, True)
    x = _rand(dim, 4)
    want = O.multmv(x)
    for hint in [(n, 5, 3), (n, 4, 4), (10, 3, 5)]:
        A = q.csr_mat(dim, ia, ja, val, sym=True, opts=_hint(*hint, **PLAIN))
        info = A.info()
        assert info.basis_internal == 0 and info.kron_minor == 0, hint
        y = np.empty(dim, dtype=np.complex128)
        A.MultMv(x, y)
        assert np.abs(y - want).max() <= 2e-13 * np.abs(want).max()
        fia, fja, fval = A.download()                                  # still the caller's rows (both triangles)
        assert fia[-1] == 2 * ia[-1] - dim
        A.destroy()


def test_a_matrix_of_a_colliding_dimension_without_the_structure_stays_as_given():
    """dim 4900 = C(8, 4)^2 -- but a random Hermitian band matrix: the search tries (8, 4, 4), the pre-check finds an entry that
    changes both indices after a handful of rows, nothing is permuted (the arrays download exactly as they went in) and the
    operator is right.  So is a Hubbard matrix whose caller turned the search off."""
    dim = 4900
    rng = np.random.default_rng(7)
    import scipy.sparse as sp
    B = sp.random(dim, dim, density=2e-3, random_state=rng, format="csr", dtype=np.float64)
    H = (B + B.T + sp.diags(rng.normal(size=dim))).tocsr().astype(np.complex128)
    H.sort_indices()
    ia, ja, val = H.indptr.astype(np.int64), H.indices.astype(np.int64), H.data
    A = q.csr_mat(dim, ia, ja, val, sym=False, opts=q.make_opts(kron_split=2, **PLAIN))
    info = A.info()
    assert info.basis_internal == 0 and info.basis_detected == 0 and info.kron_minor == 0 and info.basis_detect_ms > 0.0
    fia, fja, fval = A.download()
    assert np.array_equal(fia, ia) and np.array_equal(fja, ja) and np.array_equal(fval, val)
    x = _rand(dim, 8)
    y = np.empty(dim, dtype=np.complex128)
    A.MultMv(x, y)
    want = H @ x
    assert np.abs(y - want).max() <= 2e-13 * np.abs(want).max()
    A.destroy()
    d2_, ia2, ja2, val2, _ = refham.hubbard_csr(4, 2, 4, 4, t=1.0, U=1.1)
    A = q.csr_mat(d2_, ia2, ja2, val2, sym=True, opts=q.make_opts(kron_split=2, basis_detect=0, **PLAIN))
    assert A.info().basis_internal == 0 and A.info().basis_detect_ms == 0.0
    A.destroy()


def test_default_format_finds_the_basis_too():
    """The options the unchanged host code really passes: none (value codes + real fast path on).  The search runs before the
    value dictionary is built; the coded split then applies to the permuted operator."""
    n, nu, nd = 12, 6, 6
    dim, ia, ja, val, _ = refham.hubbard_csr(4, 3, nu, nd, t=1.0, U=1.1)
    A = q.csr_mat(dim, ia, ja, val, sym=True, opts=q.make_opts(kron_split=2))
    info = A.info()
    assert info.basis_detected == 1 and info.basis_internal == _lib.BASIS_REF_FERMION2 and info.value_dict > 0 and info.kron_minor == math.comb(n, nd)
    r = q.locate_E0_lanczos(A, nev=1, ncv=1, maxit=1000)
    ro = qo.locate_E0_lanczos(qo.Csr(dim, ia, ja, val, True), nev=1, ncv=1, maxit=1000)
    assert abs(r.E0 - ro["E0"]) <= 1e-11 * abs(ro["E0"]) and abs(abs(np.vdot(r.eigenvecs, ro["eigenvecs"])) - 1.0) < 1e-8
    A.destroy()


def test_set_basis_on_an_existing_operator():
    """qbh_csr_set_basis: the same declaration for an operator that already exists (bench.py --order reference uses it on the
    device-permuted C3).  Hubbard 4x3 in the reference's order, made on the device."""
    from quantum_basis_amd import lattices
    n, nu, nd = 12, 6, 6
    G = q.csr_mat.hubbard(n, nu, nd, lattices.square(4, 3), t=1.0, U=1.1, opts=q.make_opts(kron_split=0, **PLAIN))
    R = G.reference_order(1, n, nu, nd, opts=q.make_opts(kron_split=0, **PLAIN))
    e_ref = q.locate_E0_lanczos(R).E0
    assert R.info().kron_minor == 0
    R2 = G.reference_order(1, n, nu, nd, opts=q.make_opts(kron_split=0, **PLAIN))
    assert R2.set_basis(_lib.BASIS_DETECT, 0, 0, 0)                        # the library's own search on an existing operator
    i2 = R2.info()
    assert i2.basis_internal == 1 and i2.basis_detected == 1 and (i2.basis_n_sites, i2.basis_n_up, i2.basis_n_dn) == (n, nu, nd)
    assert abs(q.locate_E0_lanczos(R2).E0 - e_ref) <= 1e-11 * abs(e_ref)
    R2.destroy()
    assert R.set_basis(_lib.BASIS_REF_FERMION2, n, nu, nd)                 # (kron_split 0 -> 1 by the call; 4x3 is below the size 1 splits)
    info = R.info()
    assert info.basis_internal == 1 and info.kron_minor in (0, math.comb(n, nd))
    r = q.locate_E0_lanczos(R)
    assert abs(r.E0 - e_ref) <= 1e-11 * abs(e_ref)
    # the eigenvector comes back in the reference's order: it is the generator's eigenvector permuted and sign-flipped
    g = q.locate_E0_lanczos(G)
    dim, ia, ja, val, _ = refham.hubbard_csr(4, 3, nu, nd, t=1.0, U=1.1)
    assert np.abs(qo.Csr(dim, ia, ja, val, True).multmv(r.eigenvecs) - r.E0 * r.eigenvecs).max() < 1e-7
    assert abs(g.E0 - r.E0) <= 1e-11 * abs(r.E0)
    G.destroy()
    R.destroy()
