"""The Kronecker split of a SINGLE-SPECIES sector (qbh_opts.basis_kind = QBH_BASIS_SPIN_SECTOR): the sites are cut into a low and
a high half, the operator is held class-major internally (class = particles among the high sites), bonds inside the low half
are the near part, bonds inside the high half the far part (band-major per class), bonds across the cut a third, unstructured
part.  Callers keep the generator's order (vectors are translated at upload / download / randomize); every result must equal the
unsplit operator's and the oracle's."""
import numpy as np
import pytest

import quantum_basis_amd as q
from quantum_basis_amd import _lib, lattices
from oracle import qb_oracle as qo

pytestmark = pytest.mark.gpu
PLAIN = dict(value_dict=0, real_fast_path=0)


def _rand(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.normal(size=n) + 1j * rng.normal(size=n)).astype(np.complex128)


CASES = {
    "chain16": (16, 8, lattices.chain(16)),
    "kagome12": (12, 6, lattices.kagome(2, 2)),
    "triangular16": (16, 8, lattices.triangular(4, 4)),
    "chain18_n7": (18, 7, lattices.chain(18)),
    "kagome18": (18, 9, lattices.kagome(3, 2)),
}


@pytest.mark.parametrize("cross_in_near", [1, 0])
@pytest.mark.parametrize("name,h", [("chain16", 8), ("chain16", 10), ("kagome12", 6), ("triangular16", 8), ("chain18_n7", 9), ("kagome18", 9), ("kagome18", 11)])
def test_cut_sector_equals_the_unsplit_operator(name, h, cross_in_near):
    # the entries across the cut: inside the near part (default: two passes) or a third pass of their own
    n, k, bonds = CASES[name]
    P = q.csr_mat.heisenberg(n, k, bonds, J=1.0, opts=q.make_opts(kron_split=0, **PLAIN))
    ia, ja, val = P.download()
    dim = P.dim
    O = qo.Csr(dim, ia, ja.astype(np.int64), val, False)
    K = q.csr_mat.heisenberg(n, k, bonds, J=1.0, opts=q.make_opts(kron_split=2, basis_kind=_lib.BASIS_SPIN_SECTOR, n_sites=n, n_up=h, n_dn=k, kron_cross_in_near=cross_in_near, **PLAIN))
    info = K.info()
    assert info.basis_internal == _lib.BASIS_SPIN_SECTOR and info.nnz == ia[-1]
    crossing = np.mean([(a < h) != (b < h) for a, b in bonds])          # share of the bonds across the cut
    if crossing < 0.3:
        assert info.kron_classes > 1 and info.kron_inplace == 1 and info.kron_sliced == 1
        assert 0 < info.kron_far_nnz < info.nnz and (info.kron_cross_nnz == 0 if cross_in_near else 0 < info.kron_cross_nnz < 0.4 * info.nnz)
    elif crossing > 0.5:
        assert info.kron_classes == 0           # mostly unstructured: permuted, not split -- and still right (below)
    kia, kja, kval = K.download()                                    # the CALLER's rows again, through the map (bit for bit)
    assert np.array_equal(kia, ia) and np.array_equal(kja, ja) and np.array_equal(kval.view(np.float64), val.view(np.float64))
    x, y0 = _rand(dim, 1), _rand(dim, 2)
    want = O.multmv(x)
    scale = np.abs(want).max()
    v = K.vec(2)
    for alpha, beta, gamma in [(1.0, 0.0, 0.0), (0.7, -0.3, 0.25), (-1.0, 1.0, 1.5)]:
        v.upload(x, 0)
        v.upload(y0, dim)
        xy, yy = K.spmv(v.at(0), v.at(dim), alpha, beta, gamma, want_red=True)
        ref = alpha * want + beta * y0 + gamma * x
        assert np.abs(v.download(dim, dim) - ref).max() <= 2e-13 * max(scale, 1.0)
        assert abs(xy - np.vdot(x, ref)) <= 1e-11 * max(abs(np.vdot(x, ref)), 1.0)
        assert abs(yy - np.vdot(ref, ref).real) <= 1e-11 * np.vdot(ref, ref).real
    v.free()
    y = np.empty(dim, dtype=np.complex128)
    K.MultMv(x, y)
    assert np.abs(y - want).max() <= 2e-13 * scale
    # start vector, Lanczos + CG through the three passes: the unsplit operator's answers, in the caller's order
    assert np.allclose(q.vec_randomize(K, seed=1), qo.vec_randomize(dim, 1), rtol=1e-13, atol=0)
    rk, rp = q.locate_E0_lanczos(K), q.locate_E0_lanczos(P)
    assert abs(rk.E0 - rp.E0) <= 1e-11 * abs(rp.E0) and abs(rk.steps["E0"] - rp.steps["E0"]) <= 1
    assert np.abs(O.multmv(rk.eigenvecs) - rk.E0 * rk.eigenvecs).max() < 1e-7
    K.destroy()
    P.destroy()


def test_a_useless_cut_leaves_the_operator_unsplit_but_correct():
    """A cut with most bonds across it (every second site low on a chain: all bonds cross): the third part would be most of the
    operator, so no split is made -- the operator is still permuted, and still right."""
    n, k = 16, 8
    bonds = [(i, (i + 8) % 16) for i in range(16)]              # every bond joins the low half to the high half
    P = q.csr_mat.heisenberg(n, k, bonds, J=1.0, opts=q.make_opts(kron_split=0, **PLAIN))
    ia, ja, val = P.download()
    O = qo.Csr(P.dim, ia, ja.astype(np.int64), val, False)
    K = q.csr_mat.heisenberg(n, k, bonds, J=1.0, opts=q.make_opts(kron_split=2, basis_kind=_lib.BASIS_SPIN_SECTOR, n_sites=n, n_up=8, n_dn=k, **PLAIN))
    assert K.info().kron_classes == 0 and K.info().kron_minor == 0
    x = _rand(P.dim, 3)
    y = np.empty(P.dim, dtype=np.complex128)
    K.MultMv(x, y)
    assert np.abs(y - O.multmv(x)).max() <= 2e-13 * np.abs(y).max()
    K.destroy()
    P.destroy()


@pytest.mark.parametrize("cross_in_near", [1, 0])
def test_cut_sector_under_a_communicator_is_merged_back_and_stays_right(cross_in_near):
    """Only the one-class split has an exchange format of its own: a cut sector that gets a communicator (here the native
    one-rank RCCL communicator) is merged back into a CSR -- near, far and cross parts, row by row (kron_restore) -- and keeps
    its internal order and its vector translation."""
    from quantum_basis_amd import dist as qdist
    n, k, bonds = CASES["kagome18"]
    P = q.csr_mat.heisenberg(n, k, bonds, J=1.0, opts=q.make_opts(kron_split=0, **PLAIN))
    ia, ja, val = P.download()
    dim = P.dim
    O = qo.Csr(dim, ia, ja.astype(np.int64), val, False)
    K = q.csr_mat.heisenberg(n, k, bonds, J=1.0, opts=q.make_opts(kron_split=2, basis_kind=_lib.BASIS_SPIN_SECTOR, n_sites=n, n_up=9, n_dn=k, kron_cross_in_near=cross_in_near, **PLAIN))
    assert K.info().kron_classes > 1
    x = _rand(dim, 5)
    want = O.multmv(x)
    qdist.NativeComm(dim, rank=0, world=1).attach(K)
    info = K.info()
    assert info.kron_classes == 0 and info.kron_minor == 0 and info.basis_internal == _lib.BASIS_SPIN_SECTOR
    v = K.vec(2)
    v.upload(x, 0)
    K.spmv(v.at(0), v.at(dim))
    assert np.abs(v.download(dim, dim) - want).max() <= 2e-13 * np.abs(want).max()
    v.free()
    rk, rp = q.locate_E0_lanczos(K), q.locate_E0_lanczos(P)
    assert abs(rk.E0 - rp.E0) <= 1e-11 * abs(rp.E0)
    assert np.abs(O.multmv(rk.eigenvecs) - rk.E0 * rk.eigenvecs).max() < 1e-7
    K.destroy()
    P.destroy()


def _auto_cut(n, k, bonds):
    """the rule of qbh_opts.sector_cut = 0 (qbh_gen_heisenberg): feasible cut with the fewest bonds across it, nearest n / 2 among equals"""
    import math
    best, best_cross = -1, 1 << 30
    part_min = max(4, n // 4)
    for c in range(part_min, n - part_min + 1):
        if n - c > 24:
            continue
        p_min, p_max = max(0, k - c), min(n - c, k)
        if p_max <= p_min or p_max - p_min + 1 > 24:
            continue
        if any(math.comb(n - c, p) * 128.0 > 2.5e6 or math.comb(c, k - p) * 16.0 > 4.0e6 for p in range(p_min, p_max + 1)):
            continue
        cross = sum((min(a, b) < c) != (max(a, b) < c) for a, b in bonds)
        if cross < best_cross or (cross == best_cross and abs(2 * c - n) < abs(2 * best - n)):
            best, best_cross = c, cross
    return best


@pytest.mark.parametrize("name", ["kagome18", "chain18_n7", "triangular16"])
def test_the_generator_picks_the_cut_by_itself(name):
    """qbh_opts.sector_cut = 0 (default): a whole complex128 Heisenberg sector large enough for the split (kron_split = 2 here: any size;
    1: from 1e8 nonzeros, where kagome-30 picks h = 18) is held as a cut sector without the caller naming a basis; -1 keeps the rows in
    ascending pattern order; > 0 names the cut.  Host vectors are the caller's in every case (translated at the seams)."""
    n, k, bonds = CASES[name]
    bl = [tuple(b) for b in np.asarray(bonds).reshape(-1, 2)]
    P = q.csr_mat.heisenberg(n, k, bonds, J=1.0, opts=q.make_opts(kron_split=2, sector_cut=-1, **PLAIN))
    assert P.info().basis_internal == 0
    K = q.csr_mat.heisenberg(n, k, bonds, J=1.0, opts=q.make_opts(kron_split=2, **PLAIN))
    ik = K.info()
    h = _auto_cut(n, k, bl)
    assert h > 0 and ik.basis_internal == _lib.BASIS_SPIN_SECTOR and ik.basis_n_up == h and ik.basis_n_sites == n and ik.basis_n_dn == k
    E = q.csr_mat.heisenberg(n, k, bonds, J=1.0, opts=q.make_opts(kron_split=2, sector_cut=h - 1, **PLAIN))
    assert E.info().basis_n_up == h - 1
    x = _rand(P.dim, 3)
    ys = []
    for A in (P, K, E):
        y = np.empty(P.dim, dtype=np.complex128)
        A.MultMv(x, y)
        ys.append(y)
    scale = np.abs(ys[0]).max()
    assert np.abs(ys[1] - ys[0]).max() <= 2e-13 * scale and np.abs(ys[2] - ys[0]).max() <= 2e-13 * scale
    rk, rp = q.locate_E0_lanczos(K), q.locate_E0_lanczos(P)
    assert abs(rk.E0 - rp.E0) <= 1e-11 * abs(rp.E0) and abs(abs(np.vdot(rk.eigenvecs, rp.eigenvecs)) - 1.0) < 1e-8
    # the default format (coded values, real fast path) is not touched by the option
    D = q.csr_mat.heisenberg(n, k, bonds, J=1.0, opts=q.make_opts(kron_split=2))
    assert D.info().basis_internal == 0
    for A in (P, K, E, D):
        A.destroy()


def test_a_cut_sector_with_an_empty_part():
    """Found by tools/r6/fuzz_gen.py (round 6): 9 sites on a random bond graph, 3 down spins -- the cut the generator picks leaves one kind
    of rows EMPTY in a class, and the scan of a zero-length count array launched no workgroups (an error that stayed behind as HIP's "last
    error" and failed the next checked launch).  The operator must be created and equal the uncut one."""
    bonds = [(0, 7), (3, 8), (4, 8), (7, 8), (5, 2), (4, 7), (5, 7), (3, 2), (3, 1), (4, 7), (8, 6), (8, 1), (8, 4), (7, 3), (0, 4), (2, 1), (0, 6)]
    o = dict(kron_split=2, kron_sliced=2, kron_cols16=1, value_dict=0, real_fast_path=0, deterministic=1)
    K = q.csr_mat.heisenberg(9, 3, bonds, J=2.5, opts=q.make_opts(sector_cut=0, **o))
    P = q.csr_mat.heisenberg(9, 3, bonds, J=2.5, opts=q.make_opts(sector_cut=-1, **o))
    assert K.info().basis_internal == _lib.BASIS_SPIN_SECTOR and P.info().basis_internal == 0 and K.dim == P.dim == 84
    x = _rand(P.dim, 5)
    yk, yp = np.empty(P.dim, dtype=np.complex128), np.empty(P.dim, dtype=np.complex128)
    K.MultMv(x, yk)
    P.MultMv(x, yp)
    assert np.abs(yk - yp).max() <= 2e-13 * np.abs(yp).max()
    ia, ja, val = P.download()
    import scipy.sparse as sp
    e0 = np.linalg.eigvalsh(sp.csr_matrix((val, ja, ia), shape=(84, 84)).toarray())[0]
    assert abs(q.locate_E0_lanczos(K, nev=1, ncv=0).E0 - e0) <= 1e-10 * abs(e0)
    K.destroy()
    P.destroy()
