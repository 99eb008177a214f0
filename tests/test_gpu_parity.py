"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerances: the path is complex128 floating point; per-element SpMV / BLAS-1 results must
agree with the oracle to 1e-13 relative (summation order differs, nothing else), Lanczos
coefficients to 1e-10 for the first steps, ground-state energies to 1e-10 relative
(BASELINE.json north_star), step counts to +-1."""
import ctypes as C

import numpy as np
import pytest

import fastham
import helpers
import quantum_basis_amd as q
from quantum_basis_amd import _lib, lattices
from oracle import qb_oracle as qo

pytestmark = pytest.mark.gpu

SPMV_RTOL = 1e-13
E0_RTOL = 1e-10

CASES = ["chain16_sz0", "hubbard_4x2", "kagome_12", "hubbard_4x2_fast_full", "chain16_k3", "chain12_sz0"]


def _both(name, **opts):
    d, ia, ja, val, sym = helpers.case(name)
    return q.csr_mat(d, ia, ja, val, sym, opts=q.make_opts(**opts)), qo.Csr(d, ia, ja, val, sym)


def _rand(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.normal(size=n) + 1j * rng.normal(size=n)).astype(np.complex128)


def _close(a, b, rtol=SPMV_RTOL):
    scale = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() <= rtol * scale * 8


def test_native_library_is_loaded_and_sees_the_gpu():
    assert _lib.lib().qbh_device_count() >= 1
    assert _lib.SO_PATH.endswith("quantum_basis_amd/libqbhip.so")


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("kernel,vd", [(_lib.KERNEL_STREAM, 0), (_lib.KERNEL_VECTOR, 0), (_lib.KERNEL_ROWS, 0), (_lib.KERNEL_ROWS, 1), (_lib.KERNEL_STREAM, 1),
                                       (_lib.KERNEL_WAVE, 0), (_lib.KERNEL_AUTO, 0)])
def test_multmv_and_multmv2_host_seam(name, kernel, vd):
    A, O = _both(name, spmv_kernel=kernel, value_dict=vd)
    x = _rand(A.dim, 1)
    y = np.empty(A.dim, dtype=np.complex128)
    A.MultMv(x, y)
    want = O.multmv(x)
    assert _close(y, want)
    y0 = _rand(A.dim, 2)
    y2 = y0.copy()
    A.MultMv2(x, y2)                       # accumulates (src/sparse.cc:262)
    assert _close(y2, y0 + want)
    info = A.info()
    assert info.nnz == O.expand_full().nnz
    assert info.bytes_algorithmic == info.nnz * 20 + (A.dim + 1) * 8 + A.dim * 32


@pytest.mark.parametrize("kernel", [_lib.KERNEL_STREAM, _lib.KERNEL_ROWS, _lib.KERNEL_WAVE])
@pytest.mark.parametrize("npb,swz", [(1024, 1), (2048, 0), (4096, 2), (2048, 2)])
def test_kernel_geometries(npb, swz, kernel):
    A, O = _both("chain16_sz0", nnz_per_block=npb, xcd_swizzle=swz, spmv_kernel=kernel)
    x = _rand(A.dim, 5)
    y = np.empty_like(x)
    A.MultMv(x, y)
    assert _close(y, O.multmv(x))


def test_survey_vectors_chain16():
    """y = H x for x = vec_randomize(seed=1) against the vectors captured from the reference."""
    g = helpers.probe()["chain16_sz0"]
    A, _ = _both("chain16_sz0")
    x = q.vec_randomize(A, seed=1)
    assert np.allclose(x[:3].real, g["x0_2"], rtol=1e-14, atol=0) and np.all(x.imag == 0)
    y = np.empty_like(x)
    A.MultMv(x, y)
    assert np.allclose(y[:3].real, g["y0_2"], rtol=1e-13)
    assert abs(np.linalg.norm(y) - g["norm_y"]) < 1e-13
    assert abs(y.sum().real - g["sum_y"]) < 1e-12


def test_vec_randomize_same_lehmer_stream():
    """Same minstd_rand0 draws element by element; only the norm's summation order differs."""
    A, _ = _both("hubbard_4x2")
    x = q.vec_randomize(A, seed=1)
    xo = qo.vec_randomize(A.dim, 1)
    assert np.allclose(x, xo, rtol=1e-14, atol=0)
    ratio = x.real / xo.real
    assert ratio.max() - ratio.min() < 1e-15          # one common scale factor: identical draws
    x8 = q.vec_randomize(A, seed=8)
    assert np.allclose(x8, qo.vec_randomize(A.dim, 8), rtol=1e-14, atol=0)
    x0 = q.vec_randomize(A, seed=0)
    assert np.allclose(x0, 1.0 / np.sqrt(A.dim))


def test_fused_spmv_epilogue_and_reductions():
    """y <- alpha*Hx + beta*y + gamma*x with <x,y> and |y|^2 produced by the same launch."""
    A, O = _both("chain16_k3")
    n = A.dim
    x, y0 = _rand(n, 3), _rand(n, 4)
    v = A.vec(2)
    for alpha, beta, gamma in [(1.0, 0.0, 0.0), (1.0, 1.0, 0.0), (0.7, -1.3, 0.0), (1.0, 0.0, -2.5), (-1.0, 0.0, 3.25)]:
        v.upload(x, 0)
        v.upload(y0, n)
        dot, nrm2 = A.spmv(v.at(0), v.at(n), alpha, beta, gamma, want_red=True)
        got = v.download(n, n)
        want = alpha * O.multmv(x) + beta * y0 + gamma * x
        assert _close(got, want)
        assert abs(dot - np.vdot(x, want)) <= 1e-12 * abs(np.vdot(x, want)) + 1e-12
        assert abs(nrm2 - np.vdot(want, want).real) <= 1e-12 * nrm2
    v.free()


def test_blas1_building_blocks():
    A, _ = _both("hubbard_4x2")
    n = A.dim
    x, y = _rand(n, 6), _rand(n, 7)
    v = A.vec(2)
    v.upload(x, 0)
    v.upload(y, n)
    assert abs(A.dotc(v.at(0), v.at(n)) - np.vdot(x, y)) < 1e-11
    assert abs(A.nrm2(v.at(0)) - np.linalg.norm(x)) < 1e-11
    alpha = 0.3 - 0.8j
    nrm2 = A.axpy_norm(alpha, v.at(0), v.at(n))
    want = y + alpha * x
    assert _close(v.download(n, n), want)
    assert abs(nrm2 - np.vdot(want, want).real) < 1e-10
    A.scal(0.125, v.at(0))
    assert _close(v.download(0, n), 0.125 * x)
    v.free()


@pytest.mark.parametrize("name", ["chain16_sz0", "hubbard_4x2"])
def test_lanczos_coefficients_and_steps(name):
    """lanczos(..., "sr_val0"): a[], b[] against the oracle AND the reference-captured values."""
    g = helpers.probe()[name]
    A, O = _both(name)
    maxit = 1000
    dim = A.dim
    v = np.zeros(2 * dim, dtype=np.complex128)
    v[:dim] = qo.vec_randomize(dim, 1)
    vo = v.copy()
    hess, hess_o = np.zeros(2 * maxit), np.zeros(2 * maxit)
    m = q.lanczos(0, maxit - 1, maxit, dim, A, v, hess, "sr_val0")
    mo, rows_o, _ = qo.lanczos(0, maxit - 1, maxit, O, vo, hess_o, "sr_val0")
    assert abs(m - mo) <= 1 and abs(m - g["lanczos_m"]) <= 1
    assert np.allclose(hess[maxit:maxit + 10], g["a0_9"], rtol=1e-10)
    assert np.allclose(hess[1:11], g["b1_10"], rtol=1e-10)
    assert np.allclose(hess[maxit:maxit + 20], hess_o[maxit:maxit + 20], rtol=1e-9)
    ritz, s = q.hess_eigen(hess, maxit, m, "sr")
    assert abs(ritz[0] - g["E0"]) <= E0_RTOL * abs(g["E0"])
    rows = q.lanczos.last["log"]
    assert rows[0]["k"] == 4 and len(rows) == m - 3
    assert np.allclose(rows[0]["ritz"], g["log_row_k4"][1:5], rtol=2e-9)
    assert q.lanczos.last["n_matvec"] == m
    # exit contract (src/qbasis.h:1056-1058): v holds the last two normalised Lanczos vectors
    for j in (0, 1):
        assert abs(np.linalg.norm(v[j * dim:(j + 1) * dim]) - 1.0) < 1e-12
    assert abs(np.vdot(v[:dim], v[dim:])) < 1e-6


def test_lanczos_continuation_matches_single_run():
    """lanczos(k, np): 'on entry, assuming k steps of Lanczos already performed' (src/qbasis.h:1030)."""
    A, _ = _both("kagome_12")
    maxit, dim = 200, A.dim
    v1 = np.zeros(2 * dim, dtype=np.complex128)
    v1[:dim] = qo.vec_randomize(dim, 1)
    v2 = v1.copy()
    h1, h2 = np.zeros(2 * maxit), np.zeros(2 * maxit)
    m1 = q.lanczos(0, 30, maxit, dim, A, v1, h1, "dnmcs")
    m2 = q.lanczos(0, 12, maxit, dim, A, v2, h2, "dnmcs")
    assert m2 == 12
    m2 = q.lanczos(12, 18, maxit, dim, A, v2, h2, "dnmcs")
    assert m1 == m2 == 30
    assert np.allclose(h1, h2, rtol=1e-9, atol=1e-12)
    assert _close(v1, v2, rtol=1e-9)


def test_lanczos_rejects_unnormalised_start_and_bad_purpose():
    A, _ = _both("kagome_12")
    dim = A.dim
    v = np.ones(2 * dim, dtype=np.complex128)
    with pytest.raises(_lib.QbhError) as e:
        q.lanczos(0, 10, 100, dim, A, v, np.zeros(200), "sr_val0")     # assert at src/lanczos.cc:166
    assert e.value.code == -7
    v[:dim] /= np.linalg.norm(v[:dim])
    with pytest.raises(_lib.QbhError) as e:
        q.lanczos(0, 10, 100, dim, A, v, np.zeros(200), "iram")
    assert e.value.code == -9
    with pytest.raises(_lib.QbhError) as e:
        q.lanczos(0, 100, 100, dim, A, v, np.zeros(200), "sr_val0")    # assert(mm < maxit)
    assert e.value.code == -1


@pytest.mark.parametrize("name", ["hubbard_4x2", "chain16_k3"])
def test_eigenvec_cg_against_oracle(name):
    A, O = _both(name)
    dim, maxit = A.dim, 1000
    E0 = qo.locate_E0_lanczos(O, ncv=0)["E0"]
    bufs = [np.zeros(dim, dtype=np.complex128) for _ in range(8)]
    v, r, p, pp, vo, ro, po, ppo = bufs
    v[:] = qo.vec_randomize(dim, 1)
    vo[:] = v
    m, accu = q.eigenvec_CG(dim, maxit, 0, A, E0, v, r, p, pp)
    mo, accu_o, rl_o = qo.eigenvec_cg(maxit, O, E0, vo, ro, po, ppo)
    assert abs(m - mo) <= 2 and accu < q.lanczos_precision
    rl = np.array(q.eigenvec_CG.last["resid"])
    assert np.allclose(rl[:10], rl_o[:10], rtol=1e-8)
    assert abs(abs(np.vdot(v, vo)) - 1.0) < 1e-9             # same eigenvector up to phase
    assert abs(np.vdot(v, O.multmv(v)).real - E0) < 1e-10 * abs(E0)


def test_locate_E0_lanczos_main_test_known_answers():
    """src/main_test.cc test 1 on the GPU path: E0 and the three correlators."""
    k = helpers.known()["chain16_full"]
    d, ia, ja, val, sym = helpers.case("chain16_full")
    A = q.csr_mat(d, ia, ja, val, sym)
    res = q.locate_E0_lanczos(A, nev=1, ncv=1)
    assert abs(res.E0 - k["E0"]) < k["tol"]
    assert abs(res.steps["E0"] - 68) <= 1 and abs(res.steps["V0"] - 71) <= 2
    import refham
    basis = refham.spin_half_basis(16, None)
    vec = res.eigenvecs
    assert abs(helpers.expect_sz_sz(vec, basis, 0, 1) - k["Sz0Sz1"]) < k["tol"]
    assert abs(helpers.expect_sz_sz(vec, basis, 0, 2) - k["Sz0Sz2"]) < k["tol"]
    assert abs(helpers.expect_sp_sm(vec, basis, 0, 1).real - k["Sp0Sm1"]) < k["tol"]


def test_locate_E0_lanczos_two_states_against_oracle():
    """E0 -> V0 -> E1 (re-orthogonalised against phi0) -> V1, all four stages."""
    A, O = _both("chain16_sz0")
    res = q.locate_E0_lanczos(A, nev=2, ncv=2)
    ro = qo.locate_E0_lanczos(O, nev=2, ncv=2)
    assert abs(res.E0 - ro["E0"]) <= E0_RTOL * abs(ro["E0"])
    assert abs(res.E1 - ro["E1"]) <= 1e-8
    assert abs(res.E1 - helpers.known()["chain16_momentum"]["E0_k"][8]) < 1e-7
    assert res.nconv == 2 and res.eigenvecs.size == 2 * A.dim
    v0, v1 = res.eigenvecs[:A.dim], res.eigenvecs[A.dim:]
    assert abs(np.vdot(v0, v1)) < 1e-6
    assert abs(np.vdot(v1, O.multmv(v1)).real - res.E1) < 1e-8


@pytest.mark.parametrize("k", [1, 8, 13])
def test_momentum_sectors_complex_phases_known_answers(k):
    ans = helpers.known()["chain16_momentum"]
    A, _ = _both("chain16_k%d" % k)
    res = q.locate_E0_lanczos(A, nev=1, ncv=0)
    assert abs(res.E0 - ans["E0_k"][k]) < ans["tol"]


@pytest.mark.parametrize("name", ["bose_hubbard_3x3", "spinless_honeycomb", "spin1_chain"])
def test_more_example_model_families(name):
    """Three more of the reference's example programs on the GPU path: the spin-1 chain runs all four stages
    (E0 -> V0 -> E1 -> V1, examples/trans_absent/latt_chain/chain_Heisenberg_spin_one.cc:92-97), the honeycomb
    spinless-fermion model arrives in FULL storage (generate_Ham_sparse_full(0, false)), Bose-Hubbard has
    sqrt(n) amplitudes (many distinct values)."""
    import refmodels
    k = refmodels.KNOWN[name]
    d, ia, ja, val, sym = refmodels.CASES[name]()
    A = q.csr_mat(d, ia, ja, val, sym)
    O = qo.Csr(d, ia, ja, val, sym)
    x = _rand(d, 21)
    y = np.empty_like(x)
    A.MultMv(x, y)
    assert _close(y, O.multmv(x))
    if "E1" in k:
        res = q.locate_E0_lanczos(A, nev=2, ncv=2)
        ro = qo.locate_E0_lanczos(O, nev=2, ncv=2)
        assert abs(res.E1 - k["E1"]) < k["tol"] and abs(res.E1 - ro["E1"]) < 1e-8
        v0, v1 = res.eigenvecs[:d], res.eigenvecs[d:]
        assert abs(np.vdot(v0, v1)) < 1e-8
        assert np.linalg.norm(O.multmv(v1) - res.E1 * v1) < 1e-6
    else:
        res = q.locate_E0_lanczos(A, nev=1, ncv=1)
        ro = qo.locate_E0_lanczos(O, nev=1, ncv=1)
    assert abs(res.E0 - k["E0"]) < k["tol"] and abs(res.E0 - ro["E0"]) <= E0_RTOL * abs(ro["E0"])
    assert abs(res.steps["E0"] - ro["m_E0"]) <= 2
    v0 = res.eigenvecs[:d]
    assert np.linalg.norm(O.multmv(v0) - res.E0 * v0) < 1e-7


def test_gauged_complex_hubbard_energy():
    d, ia, ja, val, sym = helpers.case("hubbard_4x2")
    valc, _ = helpers.gauge(d, ia, ja, val)
    A = q.csr_mat(d, ia, ja, valc, True)
    res = q.locate_E0_lanczos(A, nev=1, ncv=0)
    assert abs(res.E0 - helpers.known()["hubbard_4x2"]["E0"]) < 1e-8


def test_iram_reverse_communication_matvec():
    """locate_E0_iram, literal path: ARPACK drives csr_mat.MultMv on the device (src/lanczos.cc:473-477)."""
    A, O = _both("hubbard_4x2")
    res = q.locate_E0_iram(A, nev=2, ncv=8, method="arpack")
    assert abs(res.E0 - helpers.known()["hubbard_4x2"]["E0"]) < 1e-8
    assert abs(res.eigenvals[1] - helpers.probe()["hubbard_4x2"]["ritz1"]) < 1e-8
    v0 = res.eigenvecs[:A.dim]
    assert np.linalg.norm(O.multmv(v0) - res.E0 * v0) < 1e-8
    # dense fall-back for dim <= 30 (src/lanczos.cc:508-542)
    dd, ia, ja, val, _ = __import__("refham").heisenberg_csr(4, [(0, 1), (1, 2), (2, 3), (3, 0)], n_dn=2)
    S = q.csr_mat(dd, ia, ja, val, True)
    nconv, w, z = q.iram(dd, S, None, 2, 4, 100, "sr")
    assert nconv == 2 and abs(w[0] + 2.0) < 1e-12


def test_iram_device_resident_against_arpack_and_known_answers():
    """qbh_iram (Krylov basis in HBM) against ARPACK's eigenvalues and the reference's asserts."""
    A, O = _both("hubbard_4x2")
    nconv, w, z = q.iram(A.dim, A, None, 4, 12, 400, "sr", method="device")
    _, wa, _ = q.iram_arpack(A.dim, A, None, 4, 12, 400, "sr")
    assert nconv == 4
    assert np.allclose(w, wa, rtol=0, atol=1e-9)
    assert abs(w[0] - helpers.known()["hubbard_4x2"]["E0"]) < 1e-8
    assert abs(w[0] - helpers.probe()["hubbard_4x2"]["E0"]) <= 1e-10 * abs(w[0])
    Z = z.reshape(4, A.dim)
    for j in range(4):                                   # residuals and orthonormality
        assert np.linalg.norm(O.multmv(Z[j]) - w[j] * Z[j]) < 1e-9
    assert np.allclose(Z.conj() @ Z.T, np.eye(4), atol=1e-10)
    # largest eigenvalues ("lr", locate_Emax_iram)
    dense = np.linalg.eigvalsh(O.to_dense()) if A.dim <= 5000 else None
    nconv, wl, _ = q.iram(A.dim, A, None, 2, 8, 400, "lr", method="device")
    assert nconv == 2 and np.allclose(wl, dense[::-1][:2], atol=1e-9)
    assert np.allclose(w, dense[:4], atol=1e-9)


def test_iram_on_an_operator_whose_spectrum_lies_to_one_side_of_zero():
    """Found by tools/r6/fuzz_solvers.py (round 6): Hubbard on a random 8-site bond graph with 7 + 7 electrons (every state has six or seven
    doubly occupied sites: spectrum [1.20, 12.97]).  qbh_iram re-orthogonalised a second time only when the first Gram-Schmidt pass had removed
    more than 98 % of |w|^2; here a pass removes 86-97 %, the basis lost its orthogonality over the restarts and Ritz values such as -226 or
    +75.9 came back as converged.  With ARPACK's own criterion (a second pass when more than half is removed) the eigenvalues are the
    dense ones for every shift of the spectrum."""
    import scipy.sparse as sp
    import fastham
    bonds = [(4, 1), (1, 0), (7, 6), (3, 5), (3, 4), (0, 2), (7, 2), (7, 2), (0, 4), (3, 7), (0, 6)]
    H0 = fastham.hubbard_full(8, 7, 7, bonds, t=1.0, U=1.1)
    for shift in (0.0, -20.0):
        H = (H0 + shift * sp.identity(H0.shape[0])).tocsr()
        H.sort_indices()
        w = np.linalg.eigvalsh(H.toarray())
        dim, ia, ja, val = fastham.to_ref_csr(H)
        for vd, rf in ((0, 0), (1, 1)):
            A = q.csr_mat(dim, ia, ja, val, sym=False, opts=q.make_opts(value_dict=vd, real_fast_path=rf))
            for order, want in (("sr", w[0]), ("lr", w[-1])):
                for ncv in (6, 14):
                    nconv, ew, ez = q.iram(dim, A, None, 1, ncv, 3000, order=order, method="device")
                    v = ez[:dim]
                    assert nconv == 1 and abs(ew[0] - want) < 1e-10 * max(abs(want), 1.0), (shift, vd, order, ncv, ew, want)
                    assert np.abs(H @ v - ew[0] * v).max() < 1e-10 and abs(np.linalg.norm(v) - 1.0) < 1e-12
            A.destroy()


def test_iram_device_resident_with_a_large_basis():
    """the reference's largest calls: iram(20, 30) (examples/trans_symmetric/latt_square/square_Kondo.cc:172) and
    iram(30, 40) (src/model.cc:2211) -- 40 basis vectors in HBM, eigenvalues against the dense spectrum"""
    A, O = _both("hubbard_4x2")
    dense = np.linalg.eigvalsh(O.to_dense())
    for nev, ncv in ((20, 30), (30, 40), (10, 64)):
        nconv, w, z = q.iram(A.dim, A, None, nev, ncv, 2000, "sr", method="device")
        assert nconv == nev and np.allclose(w, dense[:nev], atol=1e-8), (nev, ncv, np.abs(w - dense[:nev]).max())
        Z = z.reshape(nev, A.dim)
        assert np.allclose(Z.conj() @ Z.T, np.eye(nev), atol=1e-8)
        for j in (0, nev - 1):
            assert np.linalg.norm(O.multmv(Z[j]) - w[j] * Z[j]) < 1e-7


def test_iram_main_test_tj_chain_degenerate_pair():
    """src/main_test.cc:113-211: locate_E0_iram(full, 4, 8) on the t-J chain, E0 = E1 = -9.762087307."""
    k = helpers.known()["tJ_chain12"]
    d, ia, ja, val = helpers.tj_chain_csr()
    assert d == 34650
    A = q.csr_mat(d, ia, ja, val, sym=False)
    res = q.locate_E0_iram(A, nev=4, ncv=8)              # device-resident
    assert abs(res.eigenvals[0] - k["E0"]) < k["tol"] and abs(res.eigenvals[1] - k["E1"]) < k["tol"]
    res2 = q.locate_E0_iram(A, nev=4, ncv=8, method="arpack")
    assert abs(res2.eigenvals[0] - k["E0"]) < k["tol"] and abs(res2.eigenvals[1] - k["E1"]) < k["tol"]
    assert np.allclose(res.eigenvals, res2.eigenvals, atol=1e-8)
    # the two degenerate eigenvectors span the same plane in both solvers
    Zd = res.eigenvecs.reshape(4, d)[:2]
    Za = np.linalg.qr(res2.eigenvecs.reshape(4, d)[:2].T)[0].T       # ARPACK's pair is not exactly orthonormal
    assert np.allclose(Zd.conj() @ Zd.T, np.eye(2), atol=1e-10)       # the device pair is
    sv = np.linalg.svd(Zd.conj() @ Za.T, compute_uv=False)
    assert np.all(np.abs(sv - 1.0) < 1e-6)


def test_iram_complex_momentum_sector():
    ans = helpers.known()["chain16_momentum"]
    A, _ = _both("chain16_k3")
    nconv, w, _ = q.iram(A.dim, A, None, 2, 8, 400, "sr")
    assert abs(w[0] - ans["E0_k"][3]) < ans["tol"]


def test_value_dictionary_is_exact():
    A, O = _both("hubbard_4x2", value_dict=1)
    assert 0 < A.info().value_dict <= 256
    x = _rand(A.dim, 9)
    y = np.empty_like(x)
    A.MultMv(x, y)
    assert _close(y, O.multmv(x))
    # a genuinely complex sector: coded and uncoded operators hold the same bits
    B, OB = _both("chain16_k3", value_dict=1)
    assert 0 < B.info().value_dict <= 256
    y = np.empty(B.dim, dtype=np.complex128)
    xb = _rand(B.dim, 10)
    B.MultMv(xb, y)
    assert _close(y, OB.multmv(xb))
    C2, _ = _both("chain16_k3", value_dict=0)
    assert C2.info().value_dict == 0
    y2 = np.empty_like(y)
    C2.MultMv(xb, y2)
    assert np.abs(y2 - y).max() <= 1e-13 * np.abs(y).max()             # same products, different summation order
    _, _, vb = B.download()
    _, _, vc = C2.download()
    assert np.array_equal(vb.view(np.uint64), vc.view(np.uint64))


@pytest.mark.parametrize("n_values,real", [(300, False), (1024, True), (1025, False), (5000, False), (70000, False)])
def test_two_byte_value_codes(n_values, real):
    """Matrices with 257..65536 distinct values are stored with 2-byte codes (dictionary in LDS up to 1024
    entries, read through the caches beyond); results carry the same bits as the uncoded operator."""
    import scipy.sparse as sp
    rng = np.random.default_rng(n_values)
    n, per_row = 6000, 24
    pool = rng.normal(size=n_values) + (0 if real else 1j) * rng.normal(size=n_values)
    rows = np.repeat(np.arange(n), per_row)
    cols = rng.integers(0, n, size=n * per_row)
    keep = rows < cols
    U = sp.csr_matrix((pool[rng.integers(0, n_values, size=keep.sum())], (rows[keep], cols[keep])), shape=(n, n), dtype=np.complex128)
    U.sum_duplicates()
    U.data = pool[np.arange(U.nnz) % n_values]                      # duplicates summed: re-draw from the pool
    dg = sp.diags(pool.real[np.arange(n) % n_values]).astype(np.complex128)
    M = (U + U.getH() + dg).tocsr()
    M.sort_indices()
    ia, ja, val = M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.astype(np.complex128)
    distinct = len(np.unique(val))
    A = q.csr_mat(n, ia, ja, val, sym=False, opts=q.make_opts(value_dict=1))
    P = q.csr_mat(n, ia, ja, val, sym=False, opts=q.make_opts(value_dict=0))
    if distinct <= 65536:
        assert A.info().value_dict == distinct
    else:
        assert A.info().value_dict == 0
    one_byte = q.csr_mat(n, ia, ja, val, sym=False, opts=q.make_opts(value_dict=2))
    assert one_byte.info().value_dict == 0                           # value_dict=2: one-byte codes or nothing
    one_byte.destroy()
    x = _rand(n, 3)
    if real:
        x = x.real.astype(np.complex128)
    ya, yp = np.empty_like(x), np.empty_like(x)
    A.MultMv(x, ya)
    P.MultMv(x, yp)
    assert np.abs(ya - M @ x).max() <= 1e-12 * np.abs(ya).max()
    # the coded kernel stages 4096 nonzeros per workgroup, the uncoded one 2048 with 4 lanes per row: the
    # summation order differs, the values multiplied do not
    assert np.abs(ya - yp).max() <= 1e-13 * np.abs(ya).max()
    _, ja_d, va = A.download()
    assert np.array_equal(ja_d, ja) and np.array_equal(va.view(np.uint64), val.view(np.uint64))
    ra = q.locate_E0_lanczos(A, nev=1, ncv=1, maxit=600)
    rp = q.locate_E0_lanczos(P, nev=1, ncv=1, maxit=600)
    assert abs(ra.E0 - rp.E0) <= 1e-10 * abs(rp.E0)
    if real and distinct <= 65536:
        assert A.stats().n_spmv_real > 0                              # real fast path also with 2-byte codes


def test_ragged_and_degenerate_shapes():
    """Empty rows, a row longer than the LDS tile, 1x1, and very short rows."""
    rng = np.random.default_rng(5)
    n = 3000
    import scipy.sparse as sp
    M = sp.random(n, n, density=0.002, random_state=5, format="lil", dtype=np.float64)
    M[7, :] = rng.normal(size=n)                      # one dense row (3000 nnz > 2048)
    M[100:140, :] = 0                                 # empty rows
    M = M.tocsr()
    M = M + M.T
    Mc = sp.csr_matrix(M, dtype=np.complex128)
    Mc.sort_indices()
    ia, ja, val = Mc.indptr.astype(np.int64), Mc.indices.astype(np.int64), Mc.data.astype(np.complex128)
    for kernel in (_lib.KERNEL_STREAM, _lib.KERNEL_VECTOR, _lib.KERNEL_ROWS, _lib.KERNEL_WAVE):
        A = q.csr_mat(n, ia, ja, val, sym=False, opts=q.make_opts(spmv_kernel=kernel, nnz_per_block=1024))
        x = _rand(n, 12)
        y = np.empty_like(x)
        A.MultMv(x, y)
        assert _close(y, Mc @ x, rtol=1e-12)
        y2 = np.ones(n, dtype=np.complex128)
        A.MultMv2(x, y2)
        assert _close(y2, 1.0 + Mc @ x, rtol=1e-12)
    one = q.csr_mat(1, [0, 1], [0], [2.5], sym=True)
    y = np.empty(1, dtype=np.complex128)
    one.MultMv(np.array([2.0 + 1j]), y)
    assert abs(y[0] - (5.0 + 2.5j)) < 1e-15


def test_device_generators_match_numpy_assembly():
    H = fastham.hubbard_full(8, 4, 4, lattices.square(4, 2))
    A = q.csr_mat.hubbard(8, 4, 4, lattices.square(4, 2), t=1.0, U=1.1)
    ia, ja, val = A.download()
    assert A.dim == H.shape[0] and A.nnz == H.nnz
    assert np.array_equal(ia, H.indptr) and np.array_equal(ja, H.indices)
    assert np.array_equal(val, H.data.astype(np.complex128))
    res = q.locate_E0_lanczos(A, nev=1, ncv=0)
    assert abs(res.E0 - helpers.known()["hubbard_4x2"]["E0"]) < 1e-8
    # unequal fillings, 3x3 lattice (odd sizes)
    H = fastham.hubbard_full(9, 4, 3, lattices.square(3, 3), t=0.7, U=4.0)
    A = q.csr_mat.hubbard(9, 4, 3, lattices.square(3, 3), t=0.7, U=4.0)
    ia, ja, val = A.download()
    assert np.array_equal(ia, H.indptr) and np.array_equal(ja, H.indices) and np.allclose(val, H.data, rtol=0, atol=0)
    for (L, ndn, bonds) in [(12, 6, lattices.kagome(2, 2)), (16, 8, lattices.triangular(4, 4)), (14, 5, lattices.chain(14))]:
        H = fastham.heisenberg_full(L, ndn, bonds)
        A = q.csr_mat.heisenberg(L, ndn, bonds, J=1.0)
        ia, ja, val = A.download()
        assert np.array_equal(ia, H.indptr) and np.array_equal(ja, H.indices)
        assert np.array_equal(val, H.data.astype(np.complex128))
    res = q.locate_E0_lanczos(q.csr_mat.heisenberg(12, 6, lattices.kagome(2, 2)), nev=1, ncv=0)
    assert abs(res.E0 - helpers.known()["kagome_12"]["E0"]) < 1e-8


def test_row_shards_of_the_generator_tile_the_operator():
    """Rows [r0,r1) built independently equal the slice of the full operator (multi-GPU row blocks)."""
    bonds = lattices.square(4, 2)
    full = q.csr_mat.hubbard(8, 4, 4, bonds)
    ia, ja, val = full.download()
    dim = full.dim
    cut = [0, 1700, 3333, dim]
    x = _rand(dim, 21)
    xv = full.vec()
    xv.upload(x)
    yfull = full.vec()
    full.spmv(xv.ptr, yfull.ptr)
    want = yfull.download()
    for r0, r1 in zip(cut[:-1], cut[1:]):
        sh = q.csr_mat.hubbard(8, 4, 4, bonds, rows=(r0, r1))
        sia, sja, sval = sh.download()
        assert sh.dim == r1 - r0 and sh.ncols == dim and sh.row_offset == r0
        assert np.array_equal(sia, ia[r0:r1 + 1] - ia[r0])
        assert np.array_equal(sja, ja[ia[r0]:ia[r1]]) and np.array_equal(sval, val[ia[r0]:ia[r1]])
        ys = sh.vec()
        full.sync()
        sh.spmv(xv.ptr, ys.ptr)          # unsharded call convention: x is the full-length vector
        sh.sync()
        assert _close(ys.download(), want[r0:r1])


def test_medium_size_properties_chain22():
    """A size the oracle still finishes in seconds (dim 705,432): E0 vs the survey's reference value,
    Hermiticity <x,Hy> = <Hx,y>, and linearity of the device operator."""
    g = helpers.probe()["chain22_sz0"]
    A = q.csr_mat.heisenberg(22, 11, lattices.chain(22))
    assert A.dim == g["dim"] and A.nnz == g["nnz_full"]
    n = A.dim
    v = A.vec(4)
    x, y = _rand(n, 31), _rand(n, 32)
    v.upload(x, 0)
    v.upload(y, n)
    A.spmv(v.at(0), v.at(2 * n))
    A.spmv(v.at(n), v.at(3 * n))
    lhs = A.dotc(v.at(0), v.at(3 * n))
    rhs = A.dotc(v.at(2 * n), v.at(n))
    assert abs(lhs - rhs) < 1e-9 * abs(lhs)
    hx, hy = v.download(2 * n, n), v.download(3 * n, n)
    v.upload(2.0 * x - 0.5j * y, 0)
    A.spmv(v.at(0), v.at(n))
    assert _close(v.download(n, n), 2.0 * hx - 0.5j * hy, rtol=1e-12)
    v.free()
    res = q.locate_E0_lanczos(A, nev=1, ncv=0)
    assert abs(res.E0 - g["E0"]) <= E0_RTOL * abs(g["E0"])
    assert abs(res.steps["E0"] - g["lanczos_m"]) <= 3


def test_measure_full_dynamic_and_log_writer(tmp_path):
    """The device part of model::measure_full_dynamic (src/model.cc:1696-1712: normalise A_q|phi>, then
    lanczos(..., "dnmcs")) and the log_Lanczos_<purpose>.txt writer (src/lanczos.cc:102-128)."""
    A, O = _both("kagome_12")
    dim, maxit = A.dim, 60
    vec = _rand(dim, 41) * 3.0
    m, norm, hess = q.measure_full_dynamic(A, vec, maxit)
    assert m == maxit - 1 and abs(norm - np.linalg.norm(vec)) < 1e-12
    vo = np.zeros(2 * dim, dtype=np.complex128)
    vo[:dim] = vec / np.linalg.norm(vec)
    ho = np.zeros(2 * maxit)
    mo, _, _ = qo.lanczos(0, maxit - 1, maxit, O, vo, ho, "dnmcs")
    assert mo == m
    assert np.allclose(hess[maxit:maxit + 25], ho[maxit:maxit + 25], rtol=1e-8)
    assert np.allclose(hess[1:26], ho[1:26], rtol=1e-8)
    assert q.measure_full_dynamic(A, np.zeros(dim, dtype=np.complex128), maxit)[0] == 0      # norm < lanczos_precision: no run
    v = np.zeros(2 * dim, dtype=np.complex128)
    v[:dim] = qo.vec_randomize(dim, 1)
    h = np.zeros(2 * 1000)
    q.lanczos(0, 999, 1000, dim, A, v, h, "sr_val0")
    rows = q.lanczos.last["log"]
    f = str(tmp_path / "log_Lanczos_sr_val0.txt")
    q.write_lanczos_log(rows, f)
    lines = open(f).read().splitlines()
    assert len(lines) == 3 * len(rows) and lines[0].split()[0] == "#(1)" and lines[1].split()[0] == "Iter(k)"
    first = lines[2].split()
    assert int(first[0]) == 4 and abs(float(first[1]) - rows[0]["ritz"][0]) < 1e-8 * abs(rows[0]["ritz"][0])


# 4x5 with N_dn = 6: 38760 down configurations (310 KB per row) take the windowed variant of the row-staged kernel
@pytest.mark.parametrize("geom", [(8, 4, 4, "4x2", 1.0, 1.1), (9, 4, 3, "3x3", 0.7, 4.0), (12, 6, 6, "4x3", 1.0, 1.1),
                                  (20, 1, 6, "4x5", 1.0, 2.0)])
def test_matrix_free_hubbard_equals_csr(geom):
    """qbh_mf_hubbard: the operator applied from the hop tables is the CSR operator."""
    L, nu, nd, shape, t, U = geom
    bonds = {"4x2": lattices.square(4, 2), "3x3": lattices.square(3, 3), "4x3": lattices.square(4, 3),
             "4x5": lattices.square(4, 5)}[shape]
    A = q.csr_mat.hubbard(L, nu, nd, bonds, t=t, U=U)
    M = q.csr_mat.hubbard(L, nu, nd, bonds, t=t, U=U, matrix_free=True)
    assert M.dim == A.dim and M.nnz == A.nnz and M.info().kernel == _lib.KERNEL_MATRIX_FREE
    n = A.dim
    x, y0 = _rand(n, 51), _rand(n, 52)
    va, vm = A.vec(2), M.vec(2)
    for alpha, beta, gamma in [(1.0, 0.0, 0.0), (1.0, 1.0, 0.0), (0.6, -1.2, 0.0), (1.0, 0.0, -3.0)]:
        for v in (va, vm):
            v.upload(x, 0)
            v.upload(y0, n)
        da, na = A.spmv(va.at(0), va.at(n), alpha, beta, gamma, want_red=True)
        dm, nm = M.spmv(vm.at(0), vm.at(n), alpha, beta, gamma, want_red=True)
        assert _close(vm.download(n, n), va.download(n, n))
        assert abs(da - dm) <= 1e-12 * max(abs(da), 1.0) and abs(na - nm) <= 1e-12 * na
    # host-vector seam, Lanczos (real fast path), IRAM
    y = np.empty(n, dtype=np.complex128)
    M.MultMv(x, y)
    ya = np.empty(n, dtype=np.complex128)
    A.MultMv(x, ya)
    assert _close(y, ya)
    ra, rm = q.locate_E0_lanczos(A, nev=1, ncv=1), q.locate_E0_lanczos(M, nev=1, ncv=1)
    assert abs(ra.E0 - rm.E0) <= 1e-11 * abs(ra.E0) and abs(ra.steps["E0"] - rm.steps["E0"]) <= 1
    if shape == "4x5":        # one up electron: the ground level may be degenerate, check the eigenpair instead
        hv = np.empty(n, dtype=np.complex128)
        A.MultMv(rm.eigenvecs, hv)
        assert np.linalg.norm(hv - rm.E0 * rm.eigenvecs) < 1e-8
    else:
        assert abs(abs(np.vdot(ra.eigenvecs, rm.eigenvecs)) - 1.0) < 1e-8
    assert M.stats().n_spmv_real > 0
    nconv, w, _ = q.iram(n, M, None, 2, 8, 300, "sr")
    assert abs(w[0] - ra.E0) < 1e-9
    with pytest.raises(_lib.QbhError):
        M.download()
    if shape == "4x2":
        assert abs(rm.E0 - helpers.known()["hubbard_4x2"]["E0"]) < 1e-8


@pytest.mark.parametrize("case", ["chain16_k3", "chain16_k0", "tri4x4_k01", "tri4x4_k00", "tri4x4_k12", "kagome12_k10"])
def test_device_repr_generator_matches_numpy_and_reference_answers(case):
    """qbh_gen_heisenberg_repr (counterpart of generate_Ham_sparse_repr) against an independent numpy assembly in
    the same convention (entry by entry), the reference's per-momentum known answers and the survey's structural
    pins (dim 822, zero-norm representatives 0 / 22 / 6, nnz_upper)."""
    import reprham
    if case.startswith("chain16"):
        L, k = 16, int(case[-1])
        n, ndn, bonds = 16, 8, lattices.chain(16)
        perms, shifts = lattices.translations(16)
        chars = lattices.characters(shifts, (k, 0), (16, 1))
        want = helpers.known()["chain16_momentum"]["E0_k"][k]
        pins = None
    elif case.startswith("tri4x4"):
        k = (int(case[-2]), int(case[-1]))
        n, ndn, bonds = 16, 8, lattices.triangular(4, 4)
        perms, shifts = lattices.translations(4, 4)
        chars = lattices.characters(shifts, k, (4, 4))
        want = {(0, 0): -8.555514918, (0, 1): -8.002263841, (1, 2): -7.588987242}[k]   # examples/trans_symmetric/latt_triangular/...:135-139
        pins = {(0, 0): (822, 0, 10986), (0, 1): (822, 22, 10738), (1, 2): (822, 22, None)}[k]
    else:
        n, ndn, bonds = 12, 6, lattices.kagome(2, 2)
        perms, shifts = lattices.translations(2, 2, n_sub=3, site=lambda x, y, s: s + 3 * (y + 2 * x))
        chars = lattices.characters(shifts, (1, 0), (2, 2))
        want, pins = None, None
    H, reps, stab, zero = reprham.repr_heisenberg_csr(n, ndn, bonds, np.array(perms), np.array(chars))
    A = q.csr_mat.heisenberg_repr(n, ndn, bonds, perms, chars)
    ia, ja, val = A.download()
    assert A.dim == H.shape[0] and A.nnz == H.nnz
    assert np.array_equal(ia, H.indptr) and np.array_equal(ja, H.indices)
    assert np.abs(val - H.data).max() < 1e-14
    if pins:
        assert A.dim == pins[0] and int(zero.sum()) == pins[1]
        if pins[2]:
            assert (A.nnz + A.dim) // 2 == pins[2]
    dense = np.linalg.eigvalsh(H.toarray())
    res = q.locate_E0_lanczos(A, nev=1, ncv=1)
    assert abs(res.E0 - dense[0]) < 1e-9
    if want is not None:
        assert abs(res.E0 - want) < 1e-8
    if case in ("chain16_k3", "tri4x4_k01", "tri4x4_k12"):
        assert np.abs(val.imag).max() > 0.05                     # genuinely complex sector
    v = res.eigenvecs
    assert np.linalg.norm(H @ v - res.E0 * v) < 1e-8
    nconv, w, _ = q.iram(A.dim, A, None, 2, 8, 300, "sr")
    assert np.allclose(w, dense[:2], atol=1e-9)
    # the directly-coded sector (codes emitted by the generator, no complex128 array) and the uncoded one hold
    # the same bits; row shards named by (rank, world) tile the sector
    coded = A.info().value_dict
    B = q.csr_mat.heisenberg_repr(n, ndn, bonds, perms, chars, opts=q.make_opts(value_dict=0))
    assert B.info().value_dict == 0
    _, jb, vb = B.download()
    assert np.array_equal(jb, ja) and np.array_equal(vb.view(np.uint64), val.view(np.uint64))
    if case != "kagome12_k10":
        assert 0 < coded <= 256
    parts = [q.csr_mat.heisenberg_repr(n, ndn, bonds, perms, chars, shard=(r, 3)) for r in range(3)]
    rows = 0
    for r, P in enumerate(parts):
        i = P.info()
        assert i.ncols == A.dim and i.row_offset == rows
        pa, pj, pv = P.download()
        assert np.array_equal(pa + ia[rows], ia[rows:rows + i.nrows + 1])
        assert np.array_equal(pj, ja[ia[rows]:ia[rows + i.nrows]])
        assert np.array_equal(pv.view(np.uint64), val[ia[rows]:ia[rows + i.nrows]].view(np.uint64))
        rows += i.nrows
    assert rows == A.dim


@pytest.mark.parametrize("geom", [("chain", 16, 8), ("kagome12", 12, 6), ("tri4x4", 16, 7), ("chain", 22, 11), ("tri6x6", 36, 2)])
def test_matrix_free_heisenberg_equals_csr(geom):
    """qbh_mf_heisenberg: rows unranked, bonds flipped and re-ranked on the fly; same operator as the device-built CSR
    (complex x, fused epilogue and reductions, host seam, Lanczos in the all-real path, IRAM)."""
    name, L, ndn = geom
    bonds = {"chain": lattices.chain(L), "kagome12": lattices.kagome(2, 2), "tri4x4": lattices.triangular(4, 4),
             "tri6x6": lattices.triangular(6, 6)}[name]
    A = q.csr_mat.heisenberg(L, ndn, bonds, J=1.0)
    M = q.csr_mat.heisenberg(L, ndn, bonds, J=1.0, matrix_free=True)
    assert M.dim == A.dim and M.nnz == A.nnz and M.info().kernel == _lib.KERNEL_MATRIX_FREE
    n = A.dim
    x, y0 = _rand(n, 61), _rand(n, 62)
    va, vm = A.vec(2), M.vec(2)
    for alpha, beta, gamma in [(1.0, 0.0, 0.0), (1.0, 1.0, 0.0), (0.6, -1.2, 0.0), (1.0, 0.0, -3.0)]:
        for v in (va, vm):
            v.upload(x, 0)
            v.upload(y0, n)
        da, na = A.spmv(va.at(0), va.at(n), alpha, beta, gamma, want_red=True)
        dm, nm = M.spmv(vm.at(0), vm.at(n), alpha, beta, gamma, want_red=True)
        assert _close(vm.download(n, n), va.download(n, n))
        assert abs(da - dm) <= 1e-12 * max(abs(da), 1.0) and abs(na - nm) <= 1e-12 * na
    ra, rm = q.locate_E0_lanczos(A, nev=1, ncv=1), q.locate_E0_lanczos(M, nev=1, ncv=1)
    assert abs(ra.E0 - rm.E0) <= 1e-11 * abs(ra.E0) and abs(ra.steps["E0"] - rm.steps["E0"]) <= 1
    hv = np.empty(n, dtype=np.complex128)
    A.MultMv(rm.eigenvecs, hv)
    assert np.linalg.norm(hv - rm.E0 * rm.eigenvecs) < 1e-8
    assert M.stats().n_spmv_real > 0
    nconv, w, _ = q.iram(n, M, None, 1, 16, 300, "sr")
    assert abs(w[0] - ra.E0) < 1e-9
    if name == "kagome12":
        assert abs(rm.E0 - helpers.known()["kagome_12"]["E0"]) < 1e-8
    if name == "chain" and L == 16:
        assert abs(rm.E0 - helpers.known()["chain16_full"]["E0"]) < 1e-8      # the Sz = 0 sector holds the ground state
    with pytest.raises(_lib.QbhError):
        M.download()


@pytest.mark.parametrize("mf", [False, True])
def test_lanczos_on_real_packed_vectors(mf):
    """qbh_lanczos_real_dev / qbh_vec_randomize_real: the caller's vectors are packed doubles (nothing complex is
    allocated); same coefficients, step count and E0 as the complex interface, continuation included."""
    import ctypes as C
    bonds = lattices.kagome(2, 2)
    A = q.csr_mat.heisenberg(12, 6, bonds, J=1.0, matrix_free=mf)
    n, maxit = A.dim, 300
    # complex interface
    vc = A.vec(2)
    A.randomize(vc.at(0), 1)
    hc = np.zeros(2 * maxit)
    mc = q.lanczos(0, maxit - 1, maxit, n, A, None, hc, "sr_val0", device_v=vc)
    # packed doubles: n complex = 2n doubles = the two slots
    vr = A.vec(1)
    _lib.check(_lib.lib().qbh_vec_randomize_real(A.handle, vr.ptr, C.c_uint32(1)), "qbh_vec_randomize_real")
    x0 = vr.download(0, n // 2).view(np.float64)                              # n doubles = the first slot
    assert np.allclose(x0, qo.vec_randomize(n, 1).real, rtol=1e-13, atol=0)
    hr = np.zeros(2 * maxit)
    m1 = q.lanczos_real(0, 40, maxit, A, vr, hr)                              # 40 steps, then continue to convergence
    assert m1 == 40
    m2 = q.lanczos_real(m1, maxit - 1 - m1, maxit, A, vr, hr, state=q.lanczos_real.last["state"])
    assert abs(m2 - mc) <= 1
    assert np.allclose(hr[maxit:maxit + 30], hc[maxit:maxit + 30], rtol=1e-9, atol=1e-11)     # a_0 .. a_29
    assert np.allclose(hr[1:31], hc[1:31], rtol=1e-9, atol=1e-11)                               # b_1 .. b_30
    e_r, _ = q.hess_eigen(hr, maxit, m2, "sr")
    assert abs(e_r[0] - helpers.known()["kagome_12"]["E0"]) < 1e-8
    st = A.stats()
    assert st.n_spmv_real > 0


def test_lanczos_single_step_calls():
    """k + np <= 2: the driver's Ritz scratch must not depend on the step count (bench.py --warmup 1 does this)."""
    A, O = _both("hubbard_4x2")
    n, maxit = A.dim, 40
    v = A.vec(2)
    A.randomize(v.at(0), 1)
    hess = np.zeros(2 * maxit)
    # (k = 0, np = 1) runs TWO steps, as the reference's do-while does after the bootstrap step (src/lanczos.cc:167-193)
    k = q.lanczos(0, 1, maxit, n, A, None, hess, "sr_val0", device_v=v)
    assert k == 2
    for step in (1, 1, 1):
        k2 = q.lanczos(k, step, maxit, n, A, None, hess, "sr_val0", device_v=v)
        assert k2 == k + step
        k = k2
    ref = np.zeros(2 * maxit)
    vr = qo.vec_randomize(n, 1)
    vv = np.zeros(2 * n, dtype=np.complex128)
    vv[:n] = vr
    m = qo.lanczos(0, 5, maxit, O, vv, ref, "sr_val0")[0]
    assert m == 5 and np.allclose(hess[maxit:maxit + 5], ref[maxit:maxit + 5], rtol=1e-11) and np.allclose(hess[1:6], ref[1:6], rtol=1e-11)


@pytest.mark.parametrize("mf", [False, True])
def test_cg_on_real_packed_vectors(mf):
    """qbh_eigenvec_cg_real_dev after qbh_lanczos_real_dev: the whole E0 -> V0 pipeline without a complex vector;
    the eigenvector equals the one of the complex interface."""
    import ctypes as C
    bonds = lattices.square(4, 2)
    A = q.csr_mat.hubbard(8, 4, 4, bonds, t=1.0, U=1.1, matrix_free=mf)
    n, maxit = A.dim, 400
    ref = q.locate_E0_lanczos(A, nev=1, ncv=1, maxit=maxit)
    buf = A.vec(2)                                             # 2n complex128 = 4n doubles: v, r, p, pp
    at = lambda j: C.c_void_p(buf.ptr.value + 8 * n * j)
    _lib.check(_lib.lib().qbh_vec_randomize_real(A.handle, at(0), C.c_uint32(1)), "qbh_vec_randomize_real")
    hess = np.zeros(2 * maxit)
    lan = type("V", (), {"ptr": at(0)})()                      # slots 0, 1 of the buffer are the two Lanczos vectors
    m = q.lanczos_real(0, maxit - 1, maxit, A, lan, hess)
    ritz, _ = q.hess_eigen(hess, maxit, m, "sr")
    assert abs(ritz[0] - ref.E0) <= 1e-11 * abs(ref.E0) and abs(m - ref.steps["E0"]) <= 1
    _lib.check(_lib.lib().qbh_vec_randomize_real(A.handle, at(0), C.c_uint32(1)), "qbh_vec_randomize_real")   # CG start vector, :1205
    mcg, accu = q.eigenvec_CG_real(maxit, 0, A, ritz[0], at(0), at(1), at(2), at(3))
    assert accu < 2e-12 and abs(mcg - ref.steps["V0"]) <= 2
    vec = buf.download(0, n // 2).view(np.float64)             # n doubles
    assert abs(np.linalg.norm(vec) - 1.0) < 1e-10
    assert abs(abs(np.dot(vec, ref.eigenvecs.real)) - 1.0) < 1e-8
    assert abs(ritz[0] - helpers.known()["hubbard_4x2"]["E0"]) < 1e-8


def test_energy_scale_matches_the_oracle_recurrence():
    """energy_scale<T,MAT> (src/kpm.cc:45-88): iters - 1 Lanczos steps without a stop rule, bounds from the extreme Ritz values."""
    d, ia, ja, val, sym = helpers.case("hubbard_4x2")
    A = q.csr_mat(d, ia, ja, val, sym)
    iters, extend = 64, 0.1
    lo, hi = q.energy_scale(d, A, extend=extend, iters=iters)
    O = qo.Csr(d, ia, ja, val, sym)
    v = np.zeros(2 * d, dtype=np.complex128)
    v[:d] = qo.vec_randomize(d, 1)
    h = np.zeros(2 * iters)
    m = qo.lanczos(0, iters - 1, iters, O, v, h, "dnmcs")[0]
    ritz, _ = qo.hess_eigen(h, iters, m, "sr")
    assert m == iters - 1
    lo_o, hi_o = ritz[0] - extend * (ritz[m - 1] - ritz[0]), ritz[m - 1] + extend * (ritz[m - 1] - ritz[0])
    assert abs(lo - lo_o) < 1e-9 * abs(lo_o) and abs(hi - hi_o) < 1e-9 * abs(hi_o)
    w = np.linalg.eigvalsh(O.to_dense())
    assert lo < w[0] and hi > w[-1]                     # the bounds bracket the spectrum
    A.destroy()
