"""Randomised parity: Hermitian CSR matrices of many shapes (short / long / empty rows, tiny and odd
dimensions, real and complex values, upper-triangle and full storage) through every SpMV kernel and
both value codings, against scipy on the host; plus Lanczos ground-state energies against dense
diagonalisation."""
import numpy as np
import pytest
import scipy.sparse as sp

import quantum_basis_amd as q
from quantum_basis_amd import _lib

pytestmark = pytest.mark.gpu


def _random_hermitian(n, density, seed, complex_vals=True, few_values=False, long_row=False, empty_rows=0):
    rng = np.random.default_rng(seed)
    nnz_target = max(1, int(density * n * n))
    r = rng.integers(0, n, nnz_target)
    c = rng.integers(0, n, nnz_target)
    if few_values:
        v = rng.choice(np.array([0.5, -1.0, 0.25 + 0.75j, 2.0]), nnz_target)
    else:
        v = rng.normal(size=nnz_target) + (1j * rng.normal(size=nnz_target) if complex_vals else 0.0)
    M = sp.coo_matrix((v, (r, c)), shape=(n, n), dtype=np.complex128).tocsr()
    if long_row and n > 8:
        M = M.tolil()
        M[n // 3, :] = rng.normal(size=n) + 1j * rng.normal(size=n)
        M = M.tocsr()
    M = M + M.conj().T                                   # Hermitian
    if empty_rows and n > 2 * empty_rows + 2:
        keep = np.ones(n)
        keep[rng.choice(n, empty_rows, replace=False)] = 0.0
        D = sp.diags(keep)
        M = D @ M @ D
    diag = rng.choice(np.array([0.0, 1.1, 2.2]), n) if few_values else rng.normal(size=n)
    M = M + sp.diags(diag.astype(np.complex128))         # real diagonal, every entry stored
    M = sp.csr_matrix(M)
    M.sum_duplicates()
    M.sort_indices()
    return M


def _to_ref(M, upper):
    A = sp.triu(M, format="csr") if upper else M
    A = sp.csr_matrix(A)
    A.sort_indices()
    # the reference always stores the diagonal (src/qbasis.h:930): add explicit zeros where missing
    n = M.shape[0]
    A = (A + sp.diags(np.full(n, 1e-300))).tocsr()
    A.sort_indices()
    return n, A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.complex128)


SHAPES = [
    dict(n=1, density=1.0), dict(n=2, density=0.5), dict(n=7, density=0.4), dict(n=63, density=0.2),
    dict(n=257, density=0.05), dict(n=1000, density=0.004), dict(n=1000, density=0.05, few_values=True),
    dict(n=3001, density=0.002, long_row=True), dict(n=2049, density=0.003, empty_rows=40),
    dict(n=5000, density=0.0006, complex_vals=False), dict(n=4097, density=0.01, few_values=True, empty_rows=10),
]


@pytest.mark.parametrize("shape", range(len(SHAPES)))
@pytest.mark.parametrize("upper", [True, False])
def test_random_matrices_all_kernels(shape, upper):
    kw = SHAPES[shape]
    M = _random_hermitian(seed=100 + shape, **kw)
    n, ia, ja, val = _to_ref(M, upper)
    rng = np.random.default_rng(shape)
    x = (rng.normal(size=n) + 1j * rng.normal(size=n)).astype(np.complex128)
    y0 = (rng.normal(size=n) + 1j * rng.normal(size=n)).astype(np.complex128)
    want = M @ x
    scale = max(np.abs(want).max(), 1e-300)
    for kernel in (_lib.KERNEL_ROWS, _lib.KERNEL_STREAM, _lib.KERNEL_VECTOR, _lib.KERNEL_WAVE):
        for vd in (0, 1):
            for npb in (0, 1024):
                A = q.csr_mat(n, ia, ja, val, sym=upper, opts=q.make_opts(spmv_kernel=kernel, value_dict=vd, nnz_per_block=npb))
                y = np.empty(n, dtype=np.complex128)
                A.MultMv(x, y)
                assert np.abs(y - want).max() <= 2e-13 * scale, (kernel, vd, npb)
                y2 = y0.copy()
                A.MultMv2(x, y2)
                assert np.abs(y2 - (y0 + want)).max() <= 2e-13 * max(scale, np.abs(y0).max())
                if kw.get("few_values") and vd:
                    assert 0 < A.info().value_dict <= 256
                A.destroy()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_matrix_ground_state_against_dense(seed):
    M = _random_hermitian(400, 0.02, 500 + seed)
    n, ia, ja, val = _to_ref(M, True)
    w = np.linalg.eigvalsh(M.toarray())
    A = q.csr_mat(n, ia, ja, val, sym=True)
    res = q.locate_E0_lanczos(A, nev=1, ncv=1)
    assert abs(res.E0 - w[0]) <= 1e-10 * max(abs(w[0]), 1.0)
    v = res.eigenvecs
    assert np.linalg.norm(M @ v - res.E0 * v) < 1e-8
    nconv, wi, _ = q.iram(n, A, None, 3, 12, 300, "sr")
    assert nconv == 3 and np.allclose(wi, w[:3], atol=1e-9)
    nconv, wl, _ = q.iram(n, A, None, 2, 10, 300, "lr")
    assert np.allclose(wl, w[::-1][:2], atol=1e-9)


@pytest.mark.parametrize("n", [401, 1000, 3001])
def test_real_symmetric_matrices_take_the_all_real_solvers(n):
    """Real symmetric operators (odd and even dimension) with the reference's real start vector: Lanczos (E0, E1),
    CG and IRAM keep their vectors as packed doubles; results against the dense spectrum."""
    M = _random_hermitian(n, 8.0 / n, 900 + n, complex_vals=False)
    nn, ia, ja, val = _to_ref(M, True)
    w, U = np.linalg.eigh(M.toarray().real)
    A = q.csr_mat(nn, ia, ja, val, sym=True)
    res = q.locate_E0_lanczos(A, nev=2, ncv=2, maxit=1000)
    assert abs(res.E0 - w[0]) <= 1e-10 * max(abs(w[0]), 1.0) and abs(res.E1 - w[1]) < 1e-8
    v0, v1 = res.eigenvecs[:n], res.eigenvecs[n:]
    assert np.abs(v0.imag).max() == 0.0 and abs(np.linalg.norm(v0) - 1.0) < 1e-12
    assert abs(abs(np.vdot(v0, U[:, 0])) - 1.0) < 1e-8 and abs(abs(np.vdot(v1, U[:, 1])) - 1.0) < 1e-6
    st = A.stats()
    assert st.n_spmv_real == st.n_spmv > 0
    nconv, wi, z = q.iram(n, A, None, 3, 12, 400, "sr")
    assert nconv == 3 and np.allclose(wi, w[:3], atol=1e-9)
    Z = z.reshape(3, n)
    assert np.abs(Z.imag).max() == 0.0
    for j in range(3):
        assert np.linalg.norm(M @ Z[j] - wi[j] * Z[j]) < 1e-8
    assert np.allclose(Z.conj() @ Z.T, np.eye(3), atol=1e-10)
    nconv, wl, _ = q.iram(n, A, None, 2, 10, 400, "lr")
    assert np.allclose(wl, w[::-1][:2], atol=1e-9)
