"""The entry point the unchanged reference host code binds to -- qbh_csr_create / qbh_csr_create_rows on the host CSR
that model::generate_Ham_sparse_full assembles (src/model.cc:619-685: int64 ia/ja, complex128 val, Hermitian-upper,
reference Lin order j = Ja[i_a] + Jb[i_b]) -- at dim 853,776 (Fermi-Hubbard 4x3, half filling, t = 1, U = 1.1),
against the oracle on the same host arrays."""
import numpy as np
import pytest
import scipy.sparse as sp

import helpers  # noqa: F401  (puts tests/ helpers on the path)
import refham
import quantum_basis_amd as q
from quantum_basis_amd import _lib
from oracle import qb_oracle as qo

pytestmark = pytest.mark.gpu

_CACHE = {}


def _hubbard_4x3():
    if "h" not in _CACHE:
        d, ia, ja, val, _ = refham.hubbard_csr(4, 3, 6, 6)
        _CACHE["h"] = (d, ia, ja, val)
    return _CACHE["h"]


def _full_scipy(d, ia, ja, val):
    # explicit zeros on the diagonal must survive (the reference stores every diagonal entry, src/qbasis.h:930), so the
    # mirror is built by concatenating coordinates, not by a sparse addition (scipy drops stored zeros there)
    rows = np.repeat(np.arange(d, dtype=np.int64), np.diff(ia))
    off = rows != ja
    r = np.concatenate([rows, ja[off]])
    c = np.concatenate([ja, rows[off]])
    v = np.concatenate([val, np.conj(val[off])])
    order = np.lexsort((c, r))
    r, c, v = r[order], c[order], v[order]
    indptr = np.zeros(d + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=d), out=indptr[1:])
    F = sp.csr_matrix((v, c, indptr), shape=(d, d))
    return F


def test_host_upper_csr_at_dim_8e5_against_the_oracle():
    d, ia, ja, val = _hubbard_4x3()
    assert d == 853776 and len(ja) > 1.2e7
    O = qo.Csr(d, ia, ja, val, True)                     # Hermitian-upper product of the oracle (src/sparse.cc:269-285)
    x = qo.vec_randomize(d, 1) + 1j * qo.vec_randomize(d, 5)
    want = O.multmv(x)
    scale = np.abs(want).max()
    F = _full_scipy(d, ia, ja, val)
    for vd, rfp in ((0, 0), (1, 1)):                     # north-star format, and the coded default
        A = q.csr_mat(d, ia, ja, val, sym=True, opts=q.make_opts(value_dict=vd, real_fast_path=rfp))
        info = A.info()
        assert info.nnz == F.nnz and info.create_ms > 0 and info.create_bytes_in == len(ja) * 24 + (d + 1) * 8
        y = np.empty_like(x)
        A.MultMv(x, y)
        assert np.abs(y - want).max() <= 1e-13 * scale
        y2 = want.copy()
        A.MultMv2(x, y2)                                  # y += Hx
        assert np.abs(y2 - 2 * want).max() <= 2e-13 * scale
        # the expansion on the device is exact: same pattern, columns ascending, same bits, conj mirrored
        fia, fja, fval = A.download()
        assert np.array_equal(fia, F.indptr) and np.array_equal(fja, F.indices)
        assert np.array_equal(fval, F.data)
        A.destroy()
    # full storage from the host (the reference's upper_triangle = false): same operator
    B = q.csr_mat(d, F.indptr.astype(np.int64), F.indices.astype(np.int64), F.data, sym=False,
                  opts=q.make_opts(value_dict=0))
    y = np.empty_like(x)
    B.MultMv(x, y)
    assert np.abs(y - want).max() <= 1e-13 * scale
    B.destroy()


def test_ground_state_energy_of_the_host_csr_matches_the_oracle():
    d, ia, ja, val = _hubbard_4x3()
    A = q.csr_mat(d, ia, ja, val, sym=True)
    res = q.locate_E0_lanczos(A, nev=1, ncv=0)
    O = qo.Csr(d, ia, ja, val, True)
    v = np.zeros(2 * d, dtype=np.complex128)
    v[:d] = qo.vec_randomize(d, 1)
    hess = np.zeros(2000)
    m = qo.lanczos(0, 999, 1000, O, v, hess, "sr_val0")[0]
    ritz, _ = qo.hess_eigen(hess, 1000, m, "sr")
    assert abs(res.E0 - ritz[0]) <= 1e-10 * abs(ritz[0])
    assert abs(res.steps["E0"] - m) <= 1
    A.destroy()


def test_row_blocks_created_from_the_same_host_arrays_tile_the_operator():
    d, ia, ja, val = _hubbard_4x3()
    F = _full_scipy(d, ia, ja, val)
    cuts = q.balanced_row_cuts(d, ia, ja, True, 3)
    assert cuts[0] == 0 and cuts[-1] == d and np.all(np.diff(cuts) > 0)
    per = np.diff(F.indptr[cuts])
    assert per.max() - per.min() <= 2 * np.diff(F.indptr).max()          # balanced to within a row or two
    x = qo.vec_randomize(d, 3) + 0j
    want = F @ x
    for (r0, r1) in zip(cuts[:-1], cuts[1:]):
        A = q.csr_mat(d, ia, ja, val, sym=True, rows=(int(r0), int(r1)), opts=q.make_opts(value_dict=0))
        info = A.info()
        assert (info.nrows, info.ncols, info.row_offset) == (r1 - r0, d, r0)
        fia, fja, fval = A.download()
        lo, hi = F.indptr[r0], F.indptr[r1]
        assert np.array_equal(fia, F.indptr[r0:r1 + 1] - lo)
        assert np.array_equal(fja, F.indices[lo:hi]) and np.array_equal(fval, F.data[lo:hi])
        dv = A.vec(1)                                     # shard-local y; x is the full vector without a communicator
        xv = q.DeviceVec(A, d)
        xv.upload(x)
        A.spmv(xv.ptr, dv.ptr, 1.0, 0.0, 0.0)
        y = dv.download()
        assert np.abs(y - want[r0:r1]).max() <= 1e-13 * np.abs(want).max()
        dv.free()
        xv.free()
        A.destroy()


def test_invalid_host_arrays_are_rejected_at_scale():
    d, ia, ja, val = _hubbard_4x3()
    bad = ja.copy()
    bad[len(bad) // 2] = d                                # column out of range
    with pytest.raises(_lib.QbhError) as e:
        q.csr_mat(d, ia, bad, val, sym=True)
    assert e.value.code == -1
    F = _full_scipy(d, ia, ja, val)
    fv = F.data.copy()
    r = d // 2
    k = next(p for p in range(F.indptr[r], F.indptr[r + 1]) if F.indices[p] != r)
    fv[k] += 1e-9                                         # breaks Hermiticity above sparse_precision
    with pytest.raises(_lib.QbhError) as e:
        q.csr_mat(d, F.indptr.astype(np.int64), F.indices.astype(np.int64), fv, sym=False)
    assert e.value.code == -5


def test_chunk_boundaries_of_the_staged_upload(monkeypatch):
    """The host arrays are streamed in chunks that start and end inside rows; with a 1,000-nonzero chunk every kind of
    boundary occurs thousands of times: the expansion must not depend on the chunk size."""
    d, ia, ja, val, sym = helpers.case("hubbard_4x2")
    A = q.csr_mat(d, ia, ja, val, sym=sym, opts=q.make_opts(value_dict=0))
    want = A.download()
    A.destroy()
    for chunk in ("1000", "1024", "77777"):
        monkeypatch.setenv("QBH_DEBUG", "create_chunk=" + chunk)        # a measurement knob, not a form: the debug list
        B = q.csr_mat(d, ia, ja, val, sym=sym, opts=q.make_opts(value_dict=0))
        got = B.download()
        assert all(np.array_equal(a, b) for a, b in zip(want, got)), chunk
        r0, r1 = d // 3, d // 3 + 1234
        C = q.csr_mat(d, ia, ja, val, sym=sym, rows=(r0, r1), opts=q.make_opts(value_dict=0))
        cia, cja, cval = C.download()
        lo, hi = want[0][r0], want[0][r1]
        assert np.array_equal(cia, want[0][r0:r1 + 1] - lo) and np.array_equal(cja, want[1][lo:hi]) and np.array_equal(cval, want[2][lo:hi])
        B.destroy()
        C.destroy()


def test_reference_ordered_host_csr_at_dim_1e7_through_qbh_csr_create():
    """The same entry at dim 10,400,600 (Heisenberg chain L = 26, Sz = 0; 1.46e8 nonzeros in full storage): host arrays exactly
    as the unchanged reference host code hands them over -- Hermitian-upper, int64 ia / ja, complex128, the reference's Lin order
    (device generator + qbh_csr_reference_order + download: tests/test_gpu_reforder.py checks that pass entry by entry) -- through
    qbh_csr_create, against the oracle's Hermitian-upper product on the SAME host arrays."""
    from quantum_basis_amd import lattices
    L = 26
    src = q.make_opts(value_dict=0, real_fast_path=0, spmv_kernel=_lib.KERNEL_ROWS, kron_split=0)
    G = q.csr_mat.heisenberg(L, L // 2, lattices.chain(L), J=1.0, opts=src)
    R = G.reference_order(0, L, 0, L // 2, opts=src)
    G.destroy()
    d = R.dim
    ia, ja, val = R.download()
    R.destroy()
    assert d == 10400600
    rows = np.repeat(np.arange(d, dtype=np.int64), np.diff(ia))
    keep = ja >= rows
    uia = np.zeros(d + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows[keep], minlength=d), out=uia[1:])
    uja, uval = ja[keep].astype(np.int64), val[keep]
    nnz_full = len(ja)
    del rows, keep, ia, ja, val
    O = qo.Csr(d, uia, uja, uval, True)
    x = qo.vec_randomize(d, 1) + 1j * qo.vec_randomize(d, 5)
    want = O.multmv(x)
    scale = np.abs(want).max()
    for vd, rfp in ((0, 0), (1, 1)):
        A = q.csr_mat(d, uia, uja, uval, sym=True, opts=q.make_opts(value_dict=vd, real_fast_path=rfp))
        info = A.info()
        assert info.nnz == nnz_full and info.create_bytes_in == len(uja) * 24 + (d + 1) * 8
        assert info.create_bytes_in / (info.create_ms * 1e-3) > 1e9          # validation + upload + expansion: GB/s, not MB/s
        y = np.empty_like(x)
        A.MultMv(x, y)
        assert np.abs(y - want).max() <= 1e-13 * scale
        A.destroy()
