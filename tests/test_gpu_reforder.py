"""qbh_csr_reference_order: a device-generated operator permuted ON THE DEVICE into the reference's Lin order and fermion
convention (src/basis.cc:1144-1190, src/model.cc:665-670, src/basis.cc:2650-2664) must be, entry by entry, the matrix the
numpy re-derivation of the reference's host pipeline (tests/refham.py, pinned by the survey's index checksums of the
reference's own matrices) assembles -- and must have the spectrum of the operator it came from."""
import numpy as np
import pytest
import scipy.sparse as sp

import quantum_basis_amd as q
from quantum_basis_amd import lattices

import refham

pytestmark = pytest.mark.gpu
PLAIN = dict(value_dict=0, real_fast_path=0)


class _Full:
    """Both triangles of a Hermitian-upper CSR, every stored entry kept (the reference stores the diagonal even when it is
    zero, src/sparse.cc:45-54; a scipy sum would drop those)."""

    def __init__(self, dim, ia, ja, val):
        rows = np.repeat(np.arange(dim, dtype=np.int64), np.diff(ia))
        off = ja > rows
        r = np.concatenate([rows, ja[off]])
        c = np.concatenate([ja, rows[off]])
        v = np.concatenate([val, np.conj(val[off])])
        order = np.lexsort((c, r))
        self.indices, self.data = c[order], v[order]
        self.indptr = np.zeros(dim + 1, dtype=np.int64)
        np.cumsum(np.bincount(r, minlength=dim), out=self.indptr[1:])
        self.nnz = len(self.indices)


def _full(dim, ia, ja, val):
    return _Full(dim, ia, ja, val)


@pytest.mark.parametrize("ly", [2, 3])
def test_hubbard_in_reference_order_entry_by_entry(ly):
    n = 4 * ly
    G = q.csr_mat.hubbard(n, n // 2, n // 2, lattices.square(4, ly), t=1.0, U=1.1, opts=q.make_opts(**PLAIN))
    R = G.reference_order(1, n, n // 2, n // 2)
    dim, ia, ja, val, _ = refham.hubbard_csr(4, ly, n // 2, n // 2, t=1.0, U=1.1)
    F = _full(dim, ia, ja, val)
    ria, rja, rval = R.download()
    assert R.dim == dim and ria[-1] == F.nnz
    assert np.array_equal(ria, F.indptr) and np.array_equal(rja, F.indices)          # pattern and column order exactly
    assert np.abs(rval - F.data).max() < 1e-13
    if ly == 2:          # the survey's checksums of the reference's own matrix, on the upper triangle
        rows = np.repeat(np.arange(dim), np.diff(ria))
        keep = rja >= rows
        uia = np.zeros(dim + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows[keep], minlength=dim), out=uia[1:])
        cs = refham.csr_checksums(uia, rja[keep].astype(np.int64), rval[keep])
        assert cs["sum_ia"] == 104757230 and cs["sum_ja_w"] == 419558689
    e_g = q.locate_E0_lanczos(G).E0
    e_r = q.locate_E0_lanczos(R).E0
    assert abs(e_g - e_r) < 1e-11 * abs(e_g)
    G.destroy()
    R.destroy()


def test_heisenberg_in_reference_order_entry_by_entry():
    L = 16
    G = q.csr_mat.heisenberg(L, L // 2, lattices.chain(L), J=1.0, opts=q.make_opts(**PLAIN))
    R = G.reference_order(0, L, 0, L // 2)
    dim, ia, ja, val, _ = refham.heisenberg_csr(L, refham.chain_bonds(L), J=1.0, n_dn=L // 2)
    F = _full(dim, ia, ja, val)
    ria, rja, rval = R.download()
    assert np.array_equal(ria, F.indptr) and np.array_equal(rja, F.indices) and np.abs(rval - F.data).max() < 1e-14
    rows = np.repeat(np.arange(dim), np.diff(ria))
    keep = rja >= rows
    uia = np.zeros(dim + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows[keep], minlength=dim), out=uia[1:])
    cs = refham.csr_checksums(uia, rja[keep].astype(np.int64), rval[keep])
    assert (cs["sum_ia"], cs["sum_ja_w"]) == (483762205, 1934832532)
    # the survey's y = Hx on vec_randomize(1) in the reference order
    x = R.vec(2)
    R.randomize(x.at(0), 1)
    R.spmv(x.at(0), x.at(dim), 1.0, 0.0, 0.0)
    y = x.download(dim, 3)
    assert np.allclose(y.real, [0.067126123683761876, 0.030954161484316456, -0.00367655572169668], atol=1e-14)
    x.free()
    G.destroy()
    R.destroy()


def test_reference_order_refuses_what_it_cannot_do():
    G = q.csr_mat.heisenberg(12, 6, lattices.chain(12), J=1.0)            # default options: dictionary-coded values
    with pytest.raises(q._lib.QbhError):
        G.reference_order(0, 12, 0, 6)
    G.destroy()
    G = q.csr_mat.heisenberg(12, 6, lattices.chain(12), J=1.0, opts=q.make_opts(**PLAIN))
    with pytest.raises(q._lib.QbhError):
        G.reference_order(0, 14, 0, 7)                                   # another basis
    G.destroy()


def test_reference_order_at_headline_size_beside_its_source():
    """C3 (dim 165,636,900, nnz 5.82e9) through the path of `bench.py --order reference`: the source operator is created for the
    row kernel only (no split copy, nothing timed), 116 GB permuted on the device beside it.  The permuted operator is the same
    operator in another basis: same nonzero count, Hermitian, and the ground-state energy of the matrix-free operator."""
    from quantum_basis_amd import _lib
    import ctypes as C
    bonds = lattices.square(4, 4)
    src = q.make_opts(spmv_kernel=_lib.KERNEL_ROWS, kron_split=0, **PLAIN)
    G = q.csr_mat.hubbard(16, 8, 8, bonds, t=1.0, U=1.1, opts=src)
    assert G.info().kron_minor == 0
    R = G.reference_order(1, 16, 8, 8, opts=q.make_opts(**PLAIN))
    nnz = G.nnz
    G.destroy()
    assert R.nnz == nnz and R.info().kron_minor == 0              # no product structure in the reference's order
    n = R.dim
    v = R.vec(4)
    R.randomize(v.at(0), 21)
    R.randomize(v.at(n), 22)
    R.axpy_norm(0.5j, v.at(n), v.at(0))
    R.randomize(v.at(n), 23)
    R.spmv(v.at(0), v.at(2 * n))
    R.spmv(v.at(n), v.at(3 * n))
    lhs, rhs = R.dotc(v.at(0), v.at(3 * n)), R.dotc(v.at(2 * n), v.at(n))          # <x, H y> = <H x, y>
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), 1.0)
    v.free()

    def first_steps():
        vv = R.vec(2)
        R.randomize(vv.at(0), 1)                                   # the Lehmer stream in the CALLER's element order, whatever the internal one
        hess = np.zeros(2 * 64)
        m = q.lanczos(0, 12, 64, n, R, None, hess, "dnmcs", device_v=vv)
        vv.free()
        assert m == 12
        return hess[64:76].copy(), hess[1:13].copy()
    a_plain, b_plain = first_steps()
    # ... and the same reference-ordered operator once its basis is NAMED (qbh_csr_set_basis, what bench.py --order reference does):
    # held species-major internally, split in place -- the recurrence must not notice (a_j, b_j of the first 12 steps)
    assert R.set_basis(_lib.BASIS_REF_FERMION2, 16, 8, 8)
    info = R.info()
    assert info.basis_internal == _lib.BASIS_REF_FERMION2 and info.kron_minor == 12870 and info.kron_inplace == 1 and info.kron_sliced == 1
    a_split, b_split = first_steps()
    assert np.allclose(a_split, a_plain, rtol=1e-10, atol=1e-10) and np.allclose(b_split, b_plain, rtol=1e-10, atol=1e-10)
    e_ref = q.locate_E0_lanczos(R, nev=1, ncv=0, maxit=1000).E0
    R.destroy()
    M = q.csr_mat.hubbard(16, 8, 8, bonds, t=1.0, U=1.1, matrix_free=True)
    w = M.vec(1)
    _lib.check(_lib.lib().qbh_vec_randomize_real(M.handle, w.ptr, C.c_uint32(1)), "qbh_vec_randomize_real")
    maxit = 400
    hess = np.zeros(2 * maxit)
    m = q.lanczos_real(0, maxit - 1, maxit, M, w, hess)
    e_mf = q.hess_eigen(hess, maxit, m, "sr")[0][0]
    w.free()
    M.destroy()
    assert abs(e_ref - e_mf) <= 1e-10 * abs(e_mf)
