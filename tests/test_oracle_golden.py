"""Pins the CPU oracle (oracle/qb_oracle.c) before anything trusts it:
 (a) the known answers the reference's own tests/examples assert (1e-8), and
 (b) the 17-digit vectors the survey captured from the reference's lanczos()/csr_mat.
CPU only."""
import numpy as np
import pytest

import helpers
import refham
from oracle import qb_oracle as qo


def _csr(name):
    d, ia, ja, val, sym = helpers.case(name)
    return qo.Csr(d, ia, ja, val, sym)


def test_vec_randomize_matches_reference_stream():
    g = helpers.probe()["chain16_sz0"]
    x = qo.vec_randomize(g["dim"], 1)
    assert np.allclose(x[:3].real, g["x0_2"], rtol=1e-14, atol=0)   # the norm differs from MKL dznrm2 in the last bits
    assert np.all(x.imag == 0.0)
    assert abs(np.linalg.norm(x) - 1.0) < 1e-14
    # seed 0: constant vector (src/miscellaneous.cc:374-376)
    x0 = qo.vec_randomize(10, 0)
    assert np.allclose(x0, np.sqrt(0.1))
    # the Lehmer stream itself: minstd_rand0's 10000th draw from seed 1 is 1043618065
    s = 1
    for _ in range(10000):
        s = s * 16807 % 2147483647
    assert s == 1043618065


@pytest.mark.parametrize("name", ["chain16_sz0", "hubbard_4x2"])
def test_reference_order_csr_checksums(name):
    """The numpy re-derivation reproduces the reference's CSR bit for bit (indices) / 1e-12 (values)."""
    g = helpers.probe()[name]
    d, ia, ja, val, sym = helpers.case(name)
    cs = refham.csr_checksums(ia, ja, val)
    assert d == g["dim"] and len(ja) == g["nnz_upper"]
    assert cs["sum_ia"] == g["sum_ia"] and cs["sum_ja_w"] == g["sum_ja_w"]
    assert abs(cs["sum_val"] - g["sum_val"]) < 1e-8 and abs(cs["sum_abs"] - g["sum_abs"]) < 1e-8


@pytest.mark.parametrize("name", ["chain16_sz0", "hubbard_4x2"])
def test_multmv_against_reference_vectors(name):
    g = helpers.probe()[name]
    A = _csr(name)
    x = qo.vec_randomize(A.dim, 1)
    y = A.multmv(x)
    assert np.allclose(y[:3].real, g["y0_2"], rtol=1e-13, atol=1e-16)
    assert abs(y.sum().real - g["sum_y"]) < 1e-13 * A.dim ** 0.5
    assert abs(np.linalg.norm(y) - g["norm_y"]) < 1e-13
    # Hermitian-upper and full storage agree; MultMv2 accumulates
    Af = A.expand_full()
    assert Af.nnz == 2 * A.nnz - A.dim
    assert np.allclose(Af.multmv(x), y, rtol=0, atol=1e-14)
    y2 = y.copy()
    A.multmv2(x, y2)
    assert np.allclose(y2, 2 * y, rtol=0, atol=1e-14)


@pytest.mark.parametrize("name", ["chain16_sz0", "hubbard_4x2"])
def test_lanczos_coefficients_against_reference(name):
    g = helpers.probe()[name]
    A = _csr(name)
    r = qo.locate_E0_lanczos(A, ncv=0)
    h = r["hess0"]
    maxit = 1000
    assert r["m_E0"] == g["lanczos_m"]
    assert abs(r["E0"] - g["E0"]) <= 1e-10 * abs(g["E0"])
    assert np.allclose(h[maxit:maxit + 10], g["a0_9"], rtol=1e-11, atol=0)
    assert np.allclose(h[1:11], g["b1_10"], rtol=1e-11, atol=0)
    assert abs(r["accuracy_E0"] - g["final_accuracy"]) < 5e-14
    row4 = r["log_E0"][0]
    assert row4["k"] == 4
    got = [row4["ritz"][0], row4["ritz"][1], row4["ritz"][2], row4["ritz"][3], row4["a"], row4["b"], row4["accuracy"]]
    assert np.allclose(got, g["log_row_k4"][1:], rtol=2e-9)


def test_hubbard_cg_steps_and_known_answers():
    g = helpers.probe()["hubbard_4x2"]
    k = helpers.known()["hubbard_4x2"]
    A = _csr("hubbard_4x2")
    r = qo.locate_E0_lanczos(A, nev=1, ncv=1)
    assert r["m_V0"] == g["cg_steps"]
    assert r["accu_V0"] < qo.LANCZOS_PRECISION
    assert abs(r["E0"] - k["E0"]) < k["tol"]
    v = r["eigenvecs"]
    assert abs(np.vdot(v, A.multmv(v)).real - k["E0"]) < k["tol"]


def test_main_test_chain_heisenberg():
    """src/main_test.cc test 1: E0 and three correlators of the L=16 chain (pins the CG eigenvector)."""
    k = helpers.known()["chain16_full"]
    g = helpers.probe()["chain16_full"]
    A = _csr("chain16_full")
    r = qo.locate_E0_lanczos(A, nev=1, ncv=1)
    assert abs(r["E0"] - k["E0"]) < k["tol"]
    assert r["m_E0"] == g["lanczos_m"]
    assert abs(r["m_V0"] - g["cg_steps"]) <= 1
    assert np.allclose(r["log_V0"][:3], g["cg_resid_1_3"], rtol=1e-8)
    last = r["log_E0"][-1]
    assert np.allclose(last["ritz"], g["log_row_last"][1:5], atol=5e-9)
    basis = refham.spin_half_basis(16, None)
    v = r["eigenvecs"]
    assert abs(helpers.expect_sz_sz(v, basis, 0, 1) - k["Sz0Sz1"]) < k["tol"]
    assert abs(helpers.expect_sz_sz(v, basis, 0, 2) - k["Sz0Sz2"]) < k["tol"]
    assert abs(helpers.expect_sp_sm(v, basis, 0, 1).real - k["Sp0Sm1"]) < k["tol"]


@pytest.mark.parametrize("name", ["kagome_12", "triangular_4x4"])
def test_example_ground_state_energies(name):
    k = helpers.known()[name]
    r = qo.locate_E0_lanczos(_csr(name), ncv=0)
    assert abs(r["E0"] - k["E0"]) < k["tol"]


@pytest.mark.parametrize("name", ["bose_hubbard_3x3", "spinless_honeycomb", "spin1_chain"])
def test_more_example_model_families(name):
    """Spin-1 chain (E0 and E1: the two-state path the example runs, locate_E0_lanczos(full, 2, 2)), truncated
    Bose-Hubbard and spinless fermions on the honeycomb lattice (FULL storage, as the example generates it)."""
    import refmodels
    k = refmodels.KNOWN[name]
    d, ia, ja, val, sym = refmodels.CASES[name]()
    assert sym == (name != "spinless_honeycomb")
    A = qo.Csr(d, ia, ja, val, sym)
    if "E1" in k:
        r = qo.locate_E0_lanczos(A, nev=2, ncv=2)
        assert abs(r["E1"] - k["E1"]) < k["tol"]
    else:
        r = qo.locate_E0_lanczos(A, ncv=0)
    assert abs(r["E0"] - k["E0"]) < k["tol"]


@pytest.mark.parametrize("k", [0, 1, 3, 8, 13])
def test_momentum_sector_energies_complex_phases(k):
    """examples/trans_symmetric/latt_chain/...:102-117: per-momentum E0, complex Hermitian CSR."""
    ans = helpers.known()["chain16_momentum"]
    A = _csr("chain16_k%d" % k)
    if k not in (0, 8):
        assert np.abs(A.val.imag).max() > 0.1
    r = qo.locate_E0_lanczos(A, ncv=0)
    assert abs(r["E0"] - ans["E0_k"][k]) < ans["tol"]


def test_second_state_and_reorthogonalisation():
    """sr_val1 path (src/lanczos.cc:218-226): E1 of the chain equals the k=8 sector minimum."""
    ans = helpers.known()["chain16_momentum"]
    A = _csr("chain16_sz0")
    r = qo.locate_E0_lanczos(A, nev=2, ncv=1)
    assert abs(r["E1"] - ans["E0_k"][8]) < 1e-7
    assert r["gap"] > 0.27


def test_hess_eigen_orders_and_dense_agreement():
    rng = np.random.default_rng(3)
    maxit, m = 40, 17
    h = np.zeros(2 * maxit)
    h[maxit:maxit + m] = rng.normal(size=m)
    h[1:m] = rng.uniform(0.5, 2.0, size=m - 1)
    T = np.diag(h[maxit:maxit + m]) + np.diag(h[1:m], 1) + np.diag(h[1:m], -1)
    w = np.linalg.eigvalsh(T)
    for order, want in (("sr", np.sort(w)), ("lr", np.sort(w)[::-1]),
                        ("sm", w[np.argsort(np.abs(w))]), ("lm", w[np.argsort(-np.abs(w))])):
        ritz, s = qo.hess_eigen(h, maxit, m, order)
        assert np.allclose(ritz, want, atol=1e-13)
        S = s.reshape(m, m).T      # columns are eigenvectors
        assert np.allclose(T @ S, S * ritz, atol=1e-12)


def test_to_dense_and_dense_spectrum():
    d, ia, ja, val, sym = helpers.case("chain12_sz0")
    A = qo.Csr(d, ia, ja, val, sym)
    D = A.to_dense()
    assert np.allclose(D, D.conj().T)
    w = np.linalg.eigvalsh(D)
    r = qo.locate_E0_lanczos(A, ncv=0)
    assert abs(r["E0"] - w[0]) < 1e-11


def test_gauged_complex_matrix_same_energy():
    d, ia, ja, val, sym = helpers.case("hubbard_4x2")
    valc, _ = helpers.gauge(d, ia, ja, val)
    assert np.abs(valc.imag).max() > 0.1
    r = qo.locate_E0_lanczos(qo.Csr(d, ia, ja, valc, True), ncv=0)
    assert abs(r["E0"] - helpers.known()["hubbard_4x2"]["E0"]) < 1e-8


@pytest.mark.parametrize("name", ["chain16_sz0", "hubbard_4x2", "kagome_12", "chain16_k3", "hubbard_4x2_fast_full"])
def test_oracle_spmv_against_the_real_mkl(name):
    """The reference delegates y += Hx to Intel MKL's mkl_sparse_z_mv (src/sparse.cc:287).  When an MKL
    runtime exists in the image, call exactly that (ILP64, zero-based 4-array CSR, HERMITIAN/UPPER or
    GENERAL descriptor, alpha = beta = 1) and compare with the oracle's restatement."""
    from oracle import mkl_ref
    if mkl_ref.load() is None:
        pytest.skip("no MKL runtime in this image")
    d, ia, ja, val, sym = helpers.case(name)
    M = mkl_ref.MklCsr(d, ia, ja, val, sym)
    O = qo.Csr(d, ia, ja, val, sym)
    rng = np.random.default_rng(4)
    x = (rng.normal(size=d) + 1j * rng.normal(size=d)).astype(np.complex128)
    y0 = (rng.normal(size=d) + 1j * rng.normal(size=d)).astype(np.complex128)
    ym, yo = y0.copy(), y0.copy()
    M.multmv2(x, ym)
    O.multmv2(x, yo)
    assert np.abs(ym - yo).max() <= 1e-14 * np.abs(yo).max()
    if name == "chain16_sz0":            # and MKL itself reproduces the vectors the survey captured
        g = helpers.probe()[name]
        y = M.multmv(qo.vec_randomize(d, 1))
        assert np.allclose(y[:3].real, g["y0_2"], rtol=1e-13) and abs(np.linalg.norm(y) - g["norm_y"]) < 1e-13
