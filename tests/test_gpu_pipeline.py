"""The pipelined Lanczos loop (qbh_opts.lanczos_pipeline, round 6): step m + 1 is enqueued before step m's two scalars have been
read back, the coefficients of the three-term step live on the device and the one speculative step is discarded on
convergence / breakdown.  Contract: m, a[], b[], the log rows, the stop step and the two vectors returned are EXACTLY those of
the unpipelined loop (src/lanczos.cc:193-264) -- compared bit for bit where the kernels' reductions are run-to-run
reproducible, and against the CPU oracle as every other driver test."""
import numpy as np
import pytest

import helpers
import quantum_basis_amd as q
from quantum_basis_amd import _lib, lattices
from oracle import qb_oracle as qo

pytestmark = pytest.mark.gpu
PLAIN = dict(value_dict=0, real_fast_path=0)


class _Mode:
    """One operator in one of the two modes (the device generators do not promise the same entry order inside a row from one
    handle to the next, so bit-for-bit comparisons are made on the SAME handle)."""

    def __init__(self, A, pipe):
        self.A, self.pipe, self.dim = A, pipe, A.dim


def _pair(make):
    A = make(1)
    return _Mode(A, 1), _Mode(A, 0)


def _run(A, k, nsteps, maxit, v, hess, purpose="sr_val0"):
    if isinstance(A, _Mode):
        A.A.set_option("lanczos_pipeline", A.pipe)
        A = A.A
    m = q.lanczos(k, nsteps, maxit, A.dim, A, v, hess, purpose)
    return m, dict(q.lanczos.last)


def _hubbard(shape, **o):
    lx, ly, nu, nd = shape
    return lambda pipe: q.csr_mat.hubbard(lx * ly, nu, nd, lattices.square(lx, ly), t=1.0, U=4.0,
                                          opts=q.make_opts(lanczos_pipeline=pipe, deterministic=1, **PLAIN, **o))


def _case(name, **o):
    d, ia, ja, val, sym = helpers.case(name)
    return lambda pipe: q.csr_mat(d, ia, ja, val, sym, opts=q.make_opts(lanczos_pipeline=pipe, deterministic=1, **PLAIN, **o))


MAKERS = {
    "kron_sliced_4x3": _hubbard((4, 3, 6, 6), kron_split=2),                       # the headline form: split in place, 2-byte columns
    "kron_int32cols_4x3": _hubbard((4, 3, 5, 7), kron_split=2, kron_cols16=0),
    "kron_narrow_band_3x3": _hubbard((3, 3, 4, 5), kron_split=2),                  # S % 8 != 0: cross rows, element-wise edge of the tiled copy
    "wave_unsplit_4x3": _hubbard((4, 3, 6, 6), kron_split=0, spmv_kernel=_lib.KERNEL_WAVE),
    "rows_complex_chain16_k3": _case("chain16_k3", spmv_kernel=_lib.KERNEL_ROWS),
    "stream_complex_chain16_k3": _case("chain16_k3", spmv_kernel=_lib.KERNEL_STREAM),
    "vector_kagome12": _case("kagome_12", spmv_kernel=_lib.KERNEL_VECTOR),
}


@pytest.mark.parametrize("name", sorted(MAKERS))
def test_pipelined_loop_is_the_unpipelined_loop_bit_for_bit(name):
    P, U = _pair(MAKERS[name])
    dim, maxit = P.dim, 600
    outs = []
    x0 = qo.vec_randomize(dim, 1)           # once: the oracle normalises with an OpenMP reduction, whose last bit varies from call to call
    for A in (P, U):
        v = np.zeros(2 * dim, dtype=np.complex128)
        v[:dim] = x0
        hess = np.zeros(2 * maxit)
        m, last = _run(A, 0, maxit - 1, maxit, v, hess)
        outs.append((m, hess, v, last))
    (mp, hp, vp, lp), (mu, hu, vu, lu) = outs
    assert mp == mu and 4 < mp < maxit - 1                                # both converged, at the same step
    assert np.array_equal(hp, hu)                                         # a_j, b_j: the same numbers
    assert np.array_equal(vp.view(np.float64), vu.view(np.float64))       # v_{m-1}, v_m: the speculative step destroyed nothing
    assert lp["n_matvec"] == lu["n_matvec"] == mp                         # ... and is not counted
    assert lp["state"] == lu["state"]
    assert [r["k"] for r in lp["log"]] == [r["k"] for r in lu["log"]]
    assert all(np.array_equal(a["ritz"], b["ritz"]) and a["accuracy"] == b["accuracy"] for a, b in zip(lp["log"], lu["log"]))
    for j in (0, 1):
        assert abs(np.linalg.norm(vp[j * dim:(j + 1) * dim]) - 1.0) < 1e-12
    P.A.destroy()


@pytest.mark.parametrize("name", ["kron_sliced_4x3", "rows_complex_chain16_k3"])
def test_pipelined_loop_against_the_oracle(name):
    P = MAKERS[name](1)
    ia, ja, val = P.download()
    O = qo.Csr(P.dim, ia, ja.astype(np.int64), val, False)
    dim, maxit = P.dim, 600
    v = np.zeros(2 * dim, dtype=np.complex128)
    v[:dim] = qo.vec_randomize(dim, 1)
    vo = v.copy()
    hess, hess_o = np.zeros(2 * maxit), np.zeros(2 * maxit)
    m, _ = _run(P, 0, maxit - 1, maxit, v, hess)
    mo = qo.lanczos(0, maxit - 1, maxit, O, vo, hess_o, "sr_val0")[0]
    assert abs(m - mo) <= 1
    assert np.allclose(hess[maxit:maxit + 20], hess_o[maxit:maxit + 20], rtol=1e-9)
    assert np.allclose(hess[1:21], hess_o[1:21], rtol=1e-9)
    e, eo = q.hess_eigen(hess, maxit, m, "sr")[0][0], q.hess_eigen(hess_o, maxit, mo, "sr")[0][0]
    assert abs(e - eo) <= 1e-10 * abs(eo)
    P.destroy()


@pytest.mark.parametrize("name", ["kron_sliced_4x3", "wave_unsplit_4x3", "stream_complex_chain16_k3"])
@pytest.mark.parametrize("cuts", [(1, 1, 1, 1, 25), (12, 18), (2, 7, 3, 18), (5, 5, 5, 5, 5, 5)])
def test_continuation_and_every_exit_position_of_the_rotation(name, cuts):
    """lanczos(k, np) in pieces (src/qbasis.h:1030): every piece leaves the last two vectors in the caller's slots k % 2 and
    (k - 1) % 2 whatever buffer of the three-buffer rotation they were computed in (the cuts cover all residues mod 6)."""
    P, U = _pair(MAKERS[name])
    dim, maxit, total = P.dim, 200, 30
    v0 = np.zeros(2 * dim, dtype=np.complex128)
    v0[:dim] = qo.vec_randomize(dim, 1)
    vs, hs = v0.copy(), np.zeros(2 * maxit)
    assert _run(U, 0, total, maxit, vs, hs, "dnmcs")[0] == total          # single unpipelined run
    for A in (P, U):
        v, h = v0.copy(), np.zeros(2 * maxit)
        k = 0
        for c in cuts:
            m, _ = _run(A, k, c, maxit, v, h, "dnmcs")
            assert m == max(k + c, 2)                                       # (k = 0, np = 1) runs two steps, as the reference's do-while
            k = m
            # the vectors handed back are v_{k-1}, v_k: normalised, and they continue the recurrence exactly
            for j in (0, 1):
                assert abs(np.linalg.norm(v[j * dim:(j + 1) * dim]) - 1.0) < 1e-12
        assert k == total
        assert np.allclose(h, hs, rtol=1e-9, atol=1e-12)
        assert np.abs(v - vs).max() < 1e-8
        if A is P:
            vpieces, hpieces = v, h
    assert np.array_equal(hpieces, h) and np.array_equal(vpieces.view(np.float64), v.view(np.float64))   # pieces: pipelined == unpipelined
    P.A.destroy()


def test_breakdown_discards_the_speculative_step():
    """b_m < precision (src/lanczos.cc:216): the Krylov space of a 2 x 2-block operator is exhausted after two vectors; the
    speculative step m + 1 (coefficient 1 / b_m, inf or huge) must leave m, a, b and the vectors alone."""
    n = 64
    ia = np.arange(n + 1, dtype=np.int64) * 2
    ja = np.empty(2 * n, dtype=np.int64)
    val = np.empty(2 * n, dtype=np.complex128)
    for r in range(n):                       # H = 1_{n/2} (x) [[1, 2], [2, -1]]: every Krylov space has dimension <= 2
        p = r ^ 1
        lo, hi = min(r, p), max(r, p)
        ja[2 * r], ja[2 * r + 1] = lo, hi
        d = 1.0 if r % 2 == 0 else -1.0
        val[2 * r], val[2 * r + 1] = (d, 2.0) if r == lo else (2.0, d)
    res = []
    A = q.csr_mat(n, ia, ja, val, False, opts=q.make_opts(deterministic=1, **PLAIN))
    for pipe in (1, 0):
        A.set_option("lanczos_pipeline", pipe)
        v = np.zeros(2 * n, dtype=np.complex128)
        v[:n:2] = 1.0 / np.sqrt(n / 2)       # the same vector in every block
        h = np.zeros(2 * 40)
        m, last = _run(A, 0, 30, 40, v, h)
        res.append((m, h, v, last["n_matvec"]))
    A.destroy()
    assert res[0][0] == res[1][0] == 2 and res[0][3] == res[1][3] == 2
    assert np.array_equal(res[0][1], res[1][1])
    # (u_2 / b_2 is 0 / 0 or rounding noise over its own norm, in the reference as here: whatever it is, it is the same)
    assert np.array_equal(res[0][2].view(np.float64), res[1][2].view(np.float64), equal_nan=True)
    assert abs(res[0][1][2]) < 1e-12 and abs(res[0][1][40] - 1.0) < 1e-12        # b_2 = 0, a_0 = <v|H|v> = 1


def test_profile_events_under_the_pipeline_count_every_step_once():
    """qbh_opts.profile = 1: the per-SpMV events are queued behind the steps in flight, not waited for (that would serialise
    the loop again); the discarded speculative SpMV is neither counted nor timed."""
    lx, ly, nu, nd = 4, 3, 6, 6
    A = q.csr_mat.hubbard(lx * ly, nu, nd, lattices.square(lx, ly), t=1.0, U=4.0, opts=q.make_opts(kron_split=2, profile=1, **PLAIN))
    v = A.vec(2)
    A.randomize(v.at(0), 1)
    maxit = 400
    hess = np.zeros(2 * maxit)
    A.stats(reset=True)
    m = q.lanczos(0, maxit - 1, maxit, A.dim, A, None, hess, "sr_val0", device_v=v)
    st = A.stats()
    assert st.n_spmv == m == q.lanczos.last["n_matvec"] and m < maxit - 1
    assert st.ms_spmv > 0.0 and 0.0 < st.ms_spmv_min <= st.ms_spmv / m
    # a fixed number of steps: nothing speculative goes out beyond the last one
    A.randomize(v.at(0), 1)
    A.stats(reset=True)
    assert q.lanczos(0, 10, maxit, A.dim, A, None, hess, "dnmcs", device_v=v) == 10
    assert A.stats().n_spmv == 10
    v.free()
    A.destroy()


@pytest.mark.parametrize("name", ["kron_sliced_4x3", "rows_complex_chain16_k3", "wave_unsplit_4x3"])
def test_cg_with_the_dot_product_left_on_the_device_is_the_same_cg(name):
    """eigenvec_CG (src/lanczos.cc:319-331): delta = <p, pp> stays on the device and the update pass forms alpha = gamma^2 / delta
    itself (one host synchronisation per step instead of two) -- in the host expression's arithmetic, operation by operation: the
    residual history, the step count and the vectors are bit for bit those of the loop that reads delta back."""
    A = MAKERS[name](1)
    dim, maxit = A.dim, 1000
    x0 = qo.vec_randomize(dim, 1)
    hess = np.zeros(2 * maxit)
    v = np.zeros(2 * dim, dtype=np.complex128)
    v[:dim] = x0
    m = q.lanczos(0, maxit - 1, maxit, dim, A, v, hess, "sr_val0")
    e0 = q.hess_eigen(hess, maxit, m, "sr")[0][0]
    outs = []
    for pipe in (1, 0):
        A.set_option("lanczos_pipeline", pipe)
        vs = [x0.copy()] + [np.zeros(dim, dtype=np.complex128) for _ in range(3)]
        mcg, accu = q.eigenvec_CG(dim, maxit, 0, A, e0, *vs)
        outs.append((mcg, accu, list(q.eigenvec_CG.last["resid"]), vs))
    (m1, a1, r1, v1), (m0, a0, r0, v0) = outs
    assert m1 == m0 and 5 < m1 < maxit and a1 == a0 and r1 == r0
    assert all(np.array_equal(x.view(np.float64), y.view(np.float64)) for x, y in zip(v1, v0))
    y = np.empty(dim, dtype=np.complex128)
    A.MultMv(v1[0], y)
    assert np.linalg.norm(y - e0 * v1[0]) < 1e-8
    A.destroy()
