"""The Kronecker split of a product-basis operator (qbh_opts.kron_split): H = H_near + H_far with the far part stored
band-major over the minor index and applied from a tiled copy of x.  Same values, same 20 B per nonzero, re-ordered IN PLACE
(no CSR beside it: qbh_csr_download merges the parts back); every result must equal the unsplit operator's (and the oracle's)
up to summation order."""
import math

import numpy as np
import pytest

import quantum_basis_amd as q
from quantum_basis_amd import lattices
from oracle import qb_oracle as qo

pytestmark = pytest.mark.gpu
PLAIN = dict(value_dict=0, real_fast_path=0)


def _rand(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.normal(size=n) + 1j * rng.normal(size=n)).astype(np.complex128)


@pytest.mark.parametrize("sliced", [0, 1, 2])
# the last two shapes have 4 / 8 far entries per row: 128 / 64 rows per 512-slot block, so a run of four blocks overflows the
# wave's row buffer (mid-run flush, groups cut at the flush added atomically)
@pytest.mark.parametrize("shape", [(4, 2, 4, 4), (4, 3, 6, 6), (4, 3, 5, 7), (4, 3, 7, 2), (3, 3, 4, 5), (4, 4, 1, 3), (4, 4, 2, 2)])
def test_split_operator_equals_the_unsplit_one(shape, sliced):
    # far part: 0 = row-major inside the bands, 1 = sliced in groups of 8 rows where that costs < 1/8 padding, 2 = sliced
    # even where the minor size is not a multiple of 8 and the groups straddle bands (padding, same results)
    lx, ly, nu, nd = shape
    n = lx * ly
    bonds = lattices.square(lx, ly)
    K = q.csr_mat.hubbard(n, nu, nd, bonds, t=1.0, U=1.1, opts=q.make_opts(kron_split=2, kron_sliced=sliced, **PLAIN))     # kron_split 2: always split
    P = q.csr_mat.hubbard(n, nu, nd, bonds, t=1.0, U=1.1, opts=q.make_opts(kron_split=0, **PLAIN))
    ik, ip = K.info(), P.info()
    assert ik.kron_minor == ik.ncols // int(round(ik.ncols / ik.kron_minor)) > 0 and ip.kron_minor == 0
    assert ik.kron_band in (2, 4, 8) and 0 < ik.kron_far_nnz < ik.nnz and ik.nnz == ip.nnz
    # rows of the narrow last band (S % 8 != 0) keep their far entries in the near part: a product operator never pads
    assert ik.kron_sliced == (1 if sliced and ik.kron_band == 8 else 0) and ik.kron_inplace == 1
    assert ik.bytes_matrix <= ip.bytes_matrix + 64 * ik.nrows + 65536           # no second copy of the matrix
    ia, ja, val = K.download()                                     # the parts merged back on the device ...
    pia, pja, pval = P.download()
    assert np.array_equal(ia, pia) and np.array_equal(ja, pja) and np.array_equal(val.view(np.float64), pval.view(np.float64))   # ... bit for bit
    r0, r1 = K.dim // 3, K.dim // 3 + 17                           # a row range that starts inside a band group
    sia, sja, sval = K.download(r0, r1)
    assert np.array_equal(sia, ia[r0:r1 + 1] - ia[r0]) and np.array_equal(sja, ja[ia[r0]:ia[r1]]) and np.array_equal(sval, val[ia[r0]:ia[r1]])
    O = qo.Csr(K.dim, ia, ja.astype(np.int64), val, False)
    x, y0 = _rand(K.dim, 1), _rand(K.dim, 2)
    want = O.multmv(x)
    scale = np.abs(want).max()
    for alpha, beta, gamma in [(1.0, 0.0, 0.0), (1.0, 1.0, 0.0), (0.7, -0.3, 0.25), (-1.0, 0.0, 1.5)]:
        ys = []
        for A in (K, P):
            v = A.vec(2)
            v.upload(x, 0)
            v.upload(y0, A.dim)
            xy, yy = A.spmv(v.at(0), v.at(A.dim), alpha, beta, gamma, want_red=True)        # <x, y_new>, |y_new|^2
            y = v.download(A.dim, A.dim)
            v.free()
            ref = alpha * want + beta * y0 + gamma * x
            assert np.abs(y - ref).max() <= 2e-13 * max(scale, 1.0)
            assert abs(xy - np.vdot(x, ref)) <= 1e-11 * max(abs(np.vdot(x, ref)), 1.0)
            assert abs(yy - np.vdot(ref, ref).real) <= 1e-11 * np.vdot(ref, ref).real
            ys.append(y)
        assert np.abs(ys[0] - ys[1]).max() <= 2e-13 * max(scale, 1.0)
    # the host seam of the reference (MultMv / MultMv2) and the solvers run on the split operator as on any other
    y = np.empty(K.dim, dtype=np.complex128)
    K.MultMv(x, y)
    assert np.abs(y - want).max() <= 2e-13 * scale
    if K.dim > 200:
        ek, ep = q.locate_E0_lanczos(K), q.locate_E0_lanczos(P)
        assert abs(ek.E0 - ep.E0) <= 1e-11 * abs(ep.E0)                             # Lanczos + CG eigenvector (nev = 1)
        assert abs(abs(np.vdot(ek.eigenvecs, ep.eigenvecs)) - 1.0) < 1e-8
    K.destroy()
    P.destroy()


@pytest.mark.parametrize("deterministic", [1, 0])
@pytest.mark.parametrize("shape", [(4, 3, 6, 6), (4, 3, 5, 7), (4, 4, 2, 2), (4, 2, 4, 4)])
def test_two_byte_columns_are_the_same_operator_bit_for_bit(shape, deterministic):
    """qbh_opts.kron_cols16: the parts of the split keep 2-byte columns relative to a base named by the block descriptor.  Same
    values in the same order through the same passes, so y, the fused reductions and the Lanczos coefficients must equal the
    int32 form's BIT FOR BIT (the atomics of the sliced far pass add two addends: order-independent), and qbh_csr_download must
    re-derive the int32 columns exactly."""
    lx, ly, nu, nd = shape
    n = lx * ly
    bonds = lattices.square(lx, ly)
    mk = lambda c16: q.csr_mat.hubbard(n, nu, nd, bonds, t=1.0, U=1.1,
                                       opts=q.make_opts(kron_split=2, deterministic=deterministic, kron_cols16=c16, **PLAIN))
    K16, K32 = mk(1), mk(0)
    i16, i32 = K16.info(), K32.info()
    assert i32.kron_cols16 == 0 and i16.kron_sliced == i32.kron_sliced
    assert i16.kron_cols16 == (3 if i16.kron_sliced else 1)            # the far part only in its sliced layout
    assert i16.bytes_matrix < i32.bytes_matrix
    a, b = K16.download(), K32.download()
    for u, v in zip(a, b):
        assert np.array_equal(u.view(np.float64) if u.dtype == np.complex128 else u, v.view(np.float64) if v.dtype == np.complex128 else v)
    r0, r1 = K16.dim // 3, K16.dim // 3 + 29
    for u, v in zip(K16.download(r0, r1), K32.download(r0, r1)):
        assert np.array_equal(u, v)
    x, y0 = _rand(K16.dim, 4), _rand(K16.dim, 5)
    for alpha, beta, gamma in [(1.0, 0.0, 0.0), (0.7, -0.3, 0.25)]:
        out = []
        for A in (K16, K32):
            v = A.vec(2)
            v.upload(x, 0)
            v.upload(y0, A.dim)
            red = A.spmv(v.at(0), v.at(A.dim), alpha, beta, gamma, want_red=True)
            out.append((v.download(A.dim, A.dim), red))
            v.free()
        assert np.array_equal(out[0][0].view(np.float64), out[1][0].view(np.float64))
        if deterministic:
            assert out[0][1] == out[1][1]
    if deterministic:
        h = []
        for A in (K16, K32):
            r = q.locate_E0_lanczos(A, nev=1, ncv=0, maxit=300)
            h.append(r)
        assert h[0].E0 == h[1].E0 and h[0].steps == h[1].steps
    K16.destroy()
    K32.destroy()


def test_two_byte_columns_are_taken_part_by_part_where_they_fit():
    """18 sites, 1 up + 9 down electrons: S = C(18, 9) = 48620 -- a near column relative to the block's first row can reach
    2 S > 65535, so the near part keeps int32 columns; the major count 18 fits easily, so the far part is converted
    (qbh_csr_info.kron_cols16 == 2).  Mixed forms must apply and download like the all-int32 operator."""
    n, nu, nd = 18, 1, 9
    bonds = lattices.chain(n)
    mk = lambda c16: q.csr_mat.hubbard(n, nu, nd, bonds, t=1.0, U=1.1, opts=q.make_opts(kron_split=2, deterministic=1, kron_cols16=c16, **PLAIN))
    K16, K32 = mk(1), mk(0)
    assert K16.info().kron_minor == math.comb(18, 9) and K16.info().kron_sliced == 1
    assert K16.info().kron_cols16 == 2 and K32.info().kron_cols16 == 0
    for u, v in zip(K16.download(), K32.download()):
        assert np.array_equal(u, v)
    x = _rand(K16.dim, 21)
    out = []
    for A in (K16, K32):
        v = A.vec(2)
        v.upload(x, 0)
        red = A.spmv(v.at(0), v.at(A.dim), 1.0, 0.0, 0.0, want_red=True)
        out.append((v.download(A.dim, A.dim), red))
        v.free()
    assert np.array_equal(out[0][0].view(np.float64), out[1][0].view(np.float64)) and out[0][1] == out[1][1]
    K16.destroy()
    K32.destroy()


def test_two_byte_columns_survive_a_one_rank_communicator_and_a_restore():
    """A whole operator in 2-byte columns under a 1-rank communicator (hooks run, nothing moves) and merged back into a CSR
    (kron_restore through a communicator that cannot exchange tiled blocks is covered in test_gpu_dist)."""
    n, nu, nd = 12, 6, 6
    bonds = lattices.square(4, 3)
    K = q.csr_mat.hubbard(n, nu, nd, bonds, opts=q.make_opts(kron_split=2, **PLAIN))
    assert K.info().kron_cols16 == 3
    x = _rand(K.dim, 11)
    y = np.empty(K.dim, dtype=np.complex128)
    K.MultMv(x, y)
    ia, ja, val = K.download()
    want = qo.Csr(K.dim, ia, ja.astype(np.int64), val, False).multmv(x)
    assert np.abs(y - want).max() <= 2e-13 * np.abs(want).max()
    K.destroy()


def test_split_from_host_arrays_needs_the_hint_and_is_verified():
    """An operator created from host arrays is split only when the caller names the minor size -- and only when EVERY entry
    keeps the major or the minor index: a wrong hint leaves the operator unsplit (checked on the device), never wrong."""
    n, nu, nd = 8, 3, 5
    G = q.csr_mat.hubbard(n, nu, nd, lattices.square(4, 2), opts=q.make_opts(kron_split=0, **PLAIN))
    ia, ja, val = G.download()
    dim = G.dim
    G.destroy()
    S = 56                                                          # C(8, 5)
    x = _rand(dim, 3)
    want = qo.Csr(dim, ia, ja.astype(np.int64), val, False).multmv(x)
    for minor, expect in [(0, 0), (S, S), (28, 0), (7, 0)]:
        A = q.csr_mat(dim, ia, ja.astype(np.int64), val, sym=False, opts=q.make_opts(kron_minor=minor, kron_split=2, **PLAIN))
        assert A.info().kron_minor == expect, minor
        y = np.empty(dim, dtype=np.complex128)
        A.MultMv(x, y)
        assert np.abs(y - want).max() <= 2e-13 * np.abs(want).max()
        A.destroy()


@pytest.mark.parametrize("drop_mod,expect_sliced", [(29, 1), (3, 0)])
def test_ragged_far_rows_pad_their_groups_or_fall_back_to_plain_rows(drop_mod, expect_sliced):
    """An operator that keeps the product structure but whose far rows differ INSIDE a group of 8 (here: the up-hops of the minor
    indices divisible by drop_mod removed -- a correlated hopping): the sliced far part pads such groups (value 0, the row's own
    tiled column) and then needs arrays of its own; with more than 1/8 padding the far part stays in plain tiled rows.  Both
    must apply and download exactly like the CSR they came from."""
    n, nu, nd = 8, 4, 4
    G = q.csr_mat.hubbard(n, nu, nd, lattices.square(4, 2), opts=q.make_opts(kron_split=0, **PLAIN))
    ia, ja, val = G.download()
    dim, S = G.dim, 70
    G.destroy()
    rows = np.repeat(np.arange(dim), np.diff(ia))
    far = (rows // S) != (ja // S)
    keep = ~(far & ((rows % S) % drop_mod == 0))                   # depends on the minor index only: stays Hermitian
    nia = np.zeros(dim + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows[keep], minlength=dim), out=nia[1:])
    nja, nval = ja[keep].astype(np.int64), val[keep]
    O = qo.Csr(dim, nia, nja, nval, False)
    A = q.csr_mat(dim, nia, nja, nval, sym=False, opts=q.make_opts(kron_minor=S, kron_split=2, **PLAIN))
    info = A.info()
    assert info.kron_minor == S and info.kron_inplace == 1 and info.kron_sliced == expect_sliced
    x = _rand(dim, 9)
    y = np.empty(dim, dtype=np.complex128)
    A.MultMv(x, y)
    want = O.multmv(x)
    assert np.abs(y - want).max() <= 2e-13 * np.abs(want).max()
    dia, dja, dval = A.download()
    assert np.array_equal(dia, nia) and np.array_equal(dja, nja) and np.array_equal(dval, nval)
    e = q.locate_E0_lanczos(A).E0
    eo = qo.locate_E0_lanczos(O, nev=1, ncv=1, maxit=600)["E0"]
    assert abs(e - eo) <= 1e-10 * abs(eo)
    A.destroy()


def test_split_is_skipped_where_it_does_not_apply():
    n = 8
    bonds = lattices.square(4, 2)
    assert q.csr_mat.hubbard(n, 4, 4, bonds).info().kron_minor == 0                                   # coded values: row kernel
    assert q.csr_mat.hubbard(n, 4, 4, bonds, rows=(0, 2000), opts=q.make_opts(**PLAIN)).info().kron_minor == 0   # a row shard that cuts a major index
    assert q.csr_mat.hubbard(n, 4, 4, bonds, opts=q.make_opts(value_dict=0)).info().kron_minor == 0             # real fast path wanted: the row kernel's real gather needs the CSR
    assert q.csr_mat.heisenberg(12, 6, lattices.chain(12), opts=q.make_opts(**PLAIN)).info().kron_minor == 0     # no product basis


@pytest.mark.parametrize("shape,majors", [((4, 2, 4, 4), (20, 50)), ((4, 3, 6, 6), (300, 620)), ((4, 3, 5, 7), (0, 100)), ((3, 3, 4, 5), (100, 126))])
def test_row_shard_of_whole_major_indices_is_split_in_place(shape, majors):
    """a row shard made of whole major indices keeps the two-part form (near columns all locally owned; far columns in
    the tiled order of the FULL x).  Driven with the full-length x and no communicator it must give the oracle's rows; downloaded
    it must be the rows of the unsplit operator, bit for bit."""
    lx, ly, nu, nd = shape
    n = lx * ly
    bonds = lattices.square(lx, ly)
    P = q.csr_mat.hubbard(n, nu, nd, bonds, t=1.0, U=1.1, opts=q.make_opts(kron_split=0, **PLAIN))
    ia, ja, val = P.download()
    dim = P.dim
    P.destroy()
    import math
    S = math.comb(n, nd)
    r0, r1 = majors[0] * S, majors[1] * S
    Sh = q.csr_mat.hubbard(n, nu, nd, bonds, t=1.0, U=1.1, rows=(r0, r1), opts=q.make_opts(kron_split=2, **PLAIN))
    info = Sh.info()
    assert info.kron_minor == S and info.kron_inplace == 1 and info.nrows == r1 - r0 and info.ncols == dim
    sia, sja, sval = Sh.download()
    assert np.array_equal(sia, ia[r0:r1 + 1] - ia[r0]) and np.array_equal(sja, ja[ia[r0]:ia[r1]]) and np.array_equal(sval, val[ia[r0]:ia[r1]])
    O = qo.Csr(dim, ia, ja.astype(np.int64), val, False)
    x, y0 = _rand(dim, 5), _rand(r1 - r0, 6)
    want = O.multmv(x)[r0:r1]
    xv, yv = q.engine.DeviceVec(Sh, dim), Sh.vec()
    xv.upload(x)
    for alpha, beta, gamma in [(1.0, 0.0, 0.0), (0.7, -0.3, 0.25)]:
        yv.upload(y0)
        xy, yy = Sh.spmv(xv.ptr, yv.ptr, alpha, beta, gamma, want_red=True)
        ref = alpha * want + beta * y0 + gamma * x[r0:r1]
        assert np.abs(yv.download() - ref).max() <= 2e-13 * max(np.abs(want).max(), 1.0)
        assert abs(xy - np.vdot(x[r0:r1], ref)) <= 1e-11 * max(abs(np.vdot(x[r0:r1], ref)), 1.0)
        assert abs(yy - np.vdot(ref, ref).real) <= 1e-11 * np.vdot(ref, ref).real
    xv.free()
    yv.free()
    Sh.destroy()


@pytest.mark.parametrize("deterministic", [1, 0])
def test_split_operator_gives_bit_identical_lanczos_coefficients_from_run_to_run(deterministic):
    """Two solves give the same hessenberg array bit for bit and the same step count (the stop rule of src/lanczos.cc:228-245
    consumes exactly these scalars).  Since round 5 that holds for the split operator WITH the dynamic walk as well -- and so also
    without qbh_opts.deterministic: y never depended on which wavefront computes a row, and the fused reductions are collected
    per chunk of blocks and added in a fixed order (k_spmv_wave2 CHUNKRED + k_reduce_chunks).  Hubbard 4x3, split; the four runs
    (two per option value) must also agree with each other."""
    bonds = lattices.square(4, 3)
    out = []
    for _ in range(3):
        K = q.csr_mat.hubbard(12, 6, 6, bonds, t=1.0, U=1.1, opts=q.make_opts(deterministic=deterministic, kron_split=2, **PLAIN))
        assert K.info().kron_inplace == 1 and K.info().tuned == -1
        r = q.locate_E0_lanczos(K, nev=1, ncv=0, maxit=400)
        out.append((r.hessenberg_E0.copy(), r.steps["E0"], r.E0))
        K.destroy()
    for o in out[1:]:
        assert out[0][1] == o[1] and np.array_equal(out[0][0], o[0]) and out[0][2] == o[2]
    # the static walk (qbh_opts.wave_walk = 2) is another summation order: same E0 to rounding
    K = q.csr_mat.hubbard(12, 6, 6, bonds, t=1.0, U=1.1, opts=q.make_opts(wave_walk=2, kron_split=2, **PLAIN))
    r = q.locate_E0_lanczos(K, nev=1, ncv=0, maxit=400)
    K.destroy()
    assert abs(r.E0 - out[0][2]) <= 1e-12 * abs(r.E0) and abs(r.steps["E0"] - out[0][1]) <= 1


def test_deterministic_option_on_the_coded_split_of_the_default_format():
    """the same promise for the library's default format (value codes, packed-double vectors) through its sliced split
    (qbh_kronc.hip): the near pass takes major indices in static turns instead of drawing them from a counter, so the partial sums
    of the fused reductions are formed in the same order -- bit-identical hessenberg arrays and step counts; without the option the
    results agree to rounding only.  Hubbard 4x3 on the triangular lattice (36 bonds), kron_split = 2."""
    bonds = lattices.triangular(4, 3)
    out = []
    for _ in range(2):
        K = q.csr_mat.hubbard(12, 6, 6, bonds, t=1.0, U=1.1, opts=q.make_opts(deterministic=1, kron_split=2))
        ik = K.info()
        assert ik.value_dict > 0 and ik.kron_minor == 924 and ik.kron_sliced == 1 and ik.kron_band == 16 and ik.kron_inplace == 0
        r = q.locate_E0_lanczos(K, nev=1, ncv=0, maxit=400)
        assert K.stats().n_spmv_real > 0
        out.append((r.hessenberg_E0.copy(), r.steps["E0"], r.E0))
        K.destroy()
    assert out[0][1] == out[1][1] and np.array_equal(out[0][0], out[1][0]) and out[0][2] == out[1][2]
    P = q.csr_mat.hubbard(12, 6, 6, bonds, t=1.0, U=1.1, opts=q.make_opts(kron_split=0))
    rp = q.locate_E0_lanczos(P, nev=1, ncv=0, maxit=400)
    P.destroy()
    assert abs(rp.E0 - out[0][2]) <= 1e-12 * abs(rp.E0) and abs(rp.steps["E0"] - out[0][1]) <= 1


@pytest.mark.parametrize("variant", ["kronecker_sum", "far_sees_minor", "near_sees_major", "both"])
def test_coded_split_recognises_only_what_is_there(variant):
    """The recognitions of the coded split (far part T (x) 1, near part 1 (x) T' + D) are decided by the ENTRIES, not by the model:
    the Hubbard 4x2 operator from host arrays in the default format (value codes, real fast path) -- as it is, with the far
    entries of the minor indices divisible by 3 negated (still Hermitian: a far entry keeps the minor index; the far part is
    no longer T (x) 1), with the off-diagonal near entries of every fifth major index negated (the near part is no longer
    1 (x) T' + D), and with both.  Each must give the lowest eigenvalue of ITS matrix (scipy eigsh on the same arrays) and its
    eigenvector; qbh_csr_info shows which far form was taken."""
    n, nu, nd = 8, 4, 4
    G = q.csr_mat.hubbard(n, nu, nd, lattices.square(4, 2), t=1.0, U=1.3, opts=q.make_opts(kron_split=0, **PLAIN))
    ia, ja, val = G.download()
    dim, S = G.dim, 70
    G.destroy()
    rows = np.repeat(np.arange(dim), np.diff(ia))
    val = val.copy()
    far = (rows // S) != (ja // S)
    near_off = (~far) & (rows != ja)
    if variant in ("far_sees_minor", "both"):
        val[far & ((rows % S) % 3 == 0)] *= -1.0
    if variant in ("near_sees_major", "both"):
        val[near_off & ((rows // S) % 5 == 0)] *= -1.0
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    M = sp.csr_matrix((val.real, ja, ia), shape=(dim, dim))
    assert abs(M - M.T).max() == 0.0
    w, v = spl.eigsh(M, k=1, which="SA", tol=1e-13)                 # (a dense eigh of 4900 x 4900 costs 7 s per variant)
    A = q.csr_mat(dim, ia, ja.astype(np.int64), val, sym=False, opts=q.make_opts(kron_minor=S, kron_split=2))
    info = A.info()
    assert info.value_dict > 0 and info.kron_minor == S and info.kron_sliced == 1 and info.kron_band == 16
    nnz_far = int(far.sum())
    if variant in ("kronecker_sum", "near_sees_major"):
        assert info.kron_far_nnz <= 70 * 16                        # T alone: at most 16 entries for each of the 70 major indices
    else:
        assert info.kron_far_nnz >= nnz_far                        # every far entry stored (plus padding)
    r = q.locate_E0_lanczos(A, nev=1, ncv=1, maxit=600)
    assert A.stats().n_spmv_real > 0
    assert abs(r.E0 - w[0]) <= 1e-11 * abs(w[0])
    assert abs(abs(np.vdot(r.eigenvecs, v[:, 0])) - 1.0) < 1e-7
    A.destroy()


def test_headline_operator_split_and_sliced_equals_the_matrix_free_operator_at_full_size():
    """BASELINE configs[2] (C3, dim 165,636,900, nnz 5.82e9) exactly as bench.py's headline applies it: complex128 CSR, Kronecker
    split with the sliced far part.  The oracle cannot run at this size; the independent path is the matrix-free operator (no
    stored matrix at all).  Lanczos form y <- alpha H x + beta y + gamma x with the fused reductions, complex x."""
    import math
    n_sites, nu, nd = 16, 8, 8
    bonds = lattices.square(4, 4)
    K = q.csr_mat.hubbard(n_sites, nu, nd, bonds, t=1.0, U=1.1, opts=q.make_opts(kron_split=2, **PLAIN))
    info = K.info()
    S = math.comb(16, 8)
    assert K.dim == S * S and info.kron_minor == S and info.kron_band == 8 and info.kron_sliced == 1
    assert info.kernel == q._lib.KERNEL_WAVE and 0 < info.kron_far_nnz < info.nnz
    M = q.csr_mat.hubbard(n_sites, nu, nd, bonds, t=1.0, U=1.1, matrix_free=True)
    assert M.nnz == K.nnz
    n = K.dim
    v = K.vec(4)
    K.randomize(v.at(0), 11)
    K.randomize(v.at(n), 12)
    K.axpy_norm(0.6j, v.at(n), v.at(0))                             # a genuinely complex x
    K.randomize(v.at(n), 13)                                        # y_old
    # same y_old for both operators
    K.spmv(v.at(n), v.at(2 * n), 0.0, 0.0, 1.0)                     # slot 2 <- y_old  (alpha = beta = 0, gamma = 1: a copy)
    K.spmv(v.at(n), v.at(3 * n), 0.0, 0.0, 1.0)                     # slot 3 <- y_old
    alpha, beta, gamma = 0.7, -0.3, 0.25
    xy_k, yy_k = K.spmv(v.at(0), v.at(2 * n), alpha, beta, gamma, want_red=True)
    K.sync()
    xy_m, yy_m = M.spmv(v.at(0), v.at(3 * n), alpha, beta, gamma, want_red=True)
    M.sync()
    hy = math.sqrt(yy_m)
    assert abs(yy_k - yy_m) <= 1e-12 * yy_m and abs(xy_k - xy_m) <= 1e-12 * max(abs(xy_m), hy)
    assert np.sqrt(K.axpy_norm(-1.0, v.at(2 * n), v.at(3 * n))) <= 1e-13 * hy
    v.free()
    M.destroy()
    K.destroy()


def test_split_replaces_the_csr_no_second_copy_at_headline_size():
    """Round 3 kept the split BESIDE the CSR (C3: ~232 of 288 GB, released under memory pressure).  It now replaces it: the
    operator holds nnz * 20 B + row pointers + two vectors (~125 GB at C3), a 24-vector Krylov basis (64 GB) fits beside it and
    the split stays.  The merged-back rows equal the generator's own (a second, unsplit operator made from the same rows)."""
    import torch
    n_sites, nu, nd = 16, 8, 8
    bonds = lattices.square(4, 4)
    free0 = torch.cuda.mem_get_info()[0]
    K = q.csr_mat.hubbard(n_sites, nu, nd, bonds, t=1.0, U=1.1, opts=q.make_opts(kron_split=1, **PLAIN))
    info = K.info()
    assert info.kron_minor > 0 and info.kron_inplace == 1 and info.kron_sliced == 1
    n = K.dim
    v = K.vec(3)
    K.randomize(v.at(0), 31)
    K.spmv(v.at(0), v.at(n))
    K.sync()
    used = free0 - torch.cuda.mem_get_info()[0]
    # matrix (20 B per nonzero, once) + ia, ia_n (16 B per row) + tiled x, far sums (32) + the 3 vectors (48) + descriptors and slack:
    # ~125 GB for the operator itself (round 3: ~232 GB)
    assert used <= info.nnz * 20 + 104 * n + (2 << 30), used
    basis = [K.vec(1) for _ in range(24)]                           # 24 x 2.65 GB
    assert K.info().kron_minor > 0
    K.spmv(v.at(0), v.at(2 * n))
    K.sync()
    hx = K.nrm2(v.at(n))
    assert hx > 0 and np.sqrt(K.axpy_norm(-1.0, v.at(n), v.at(2 * n))) == 0.0     # same launches, same walk-independent sums
    for b in basis:
        b.free()
    v.free()
    # rows of a middle major index, merged back, against an unsplit shard of the same rows
    S = info.kron_minor
    r0 = 6001 * S
    P = q.csr_mat.hubbard(n_sites, nu, nd, bonds, t=1.0, U=1.1, rows=(r0, r0 + S), opts=q.make_opts(kron_split=0, **PLAIN))
    pia, pja, pval = P.download(0, S)
    kia, kja, kval = K.download(r0, r0 + S)
    assert np.array_equal(kia, pia) and np.array_equal(kja, pja) and np.array_equal(kval.view(np.float64), pval.view(np.float64))
    P.destroy()
    K.destroy()


def test_headline_lattice_at_U4_reproduces_the_published_ground_state_energy():
    """An external pin at C3's size (dim 165,636,900): the 4x4 periodic Hubbard model at half filling and U = 4t is the standard
    exact-diagonalisation benchmark, E0 = -13.6219 t, E0 / N = -0.8514 (Fano, Ortolani, Parola, PRB 42, 6877 (1990); Dagotto et
    al., PRB 45, 10741 (1992)).  The matrix-free operator and the stored complex128 CSR applied through the Kronecker split with
    the sliced far part (the headline path) must both give it, and agree with each other to rounding."""
    import ctypes as C
    from quantum_basis_amd import _lib
    bonds = lattices.square(4, 4)
    M = q.csr_mat.hubbard(16, 8, 8, bonds, t=1.0, U=4.0, matrix_free=True)
    w = M.vec(1)
    _lib.check(_lib.lib().qbh_vec_randomize_real(M.handle, w.ptr, C.c_uint32(1)), "qbh_vec_randomize_real")
    maxit = 600
    hess = np.zeros(2 * maxit)
    m = q.lanczos_real(0, maxit - 1, maxit, M, w, hess)
    e_mf = q.hess_eigen(hess, maxit, m, "sr")[0][0]
    w.free()
    M.destroy()
    assert abs(e_mf - (-13.6219)) <= 1e-4 and abs(e_mf / 16 - (-0.8514)) <= 1e-4
    K = q.csr_mat.hubbard(16, 8, 8, bonds, t=1.0, U=4.0, opts=q.make_opts(kron_split=2, **PLAIN))
    assert K.info().kron_sliced == 1
    e_k = q.locate_E0_lanczos(K, nev=1, ncv=0, maxit=1000).E0
    K.destroy()
    assert abs(e_k - e_mf) <= 1e-11 * abs(e_mf)


@pytest.mark.parametrize("form", ["1", "2", "2g", "2t"])
@pytest.mark.parametrize("shape", [(4, 2, 4, 4), (4, 3, 6, 6), (4, 3, 5, 7), (4, 4, 2, 2), (4, 3, 6, 6, "tri"), (4, 4, 1, 7, "tri")])
def test_coded_real_form_of_the_split(shape, form):
    """The library's default form of a real operator (dictionary-coded values, packed-double Lanczos vectors) through the split
    for the row kernel (qbh_opts.kron_coded = 1: near launch in natural order, far launch with tiled rows and columns accumulating at
    orig(row); kron_coded = 2: both parts sliced in groups of 16 rows, far pass gathering whole lines of the tiled x, near pass
    gathering from the block of x held in LDS -- qbh_kronc.hip; "2g": with the far part in its general form, not recognised as
    T (x) 1): same E0, same step count, same eigenvector as the unsplit coded operator and as the oracle's operator."""
    lx, ly, nu, nd = shape[:4]
    n = lx * ly
    # triangular: 3 n bonds -- rows of more than 32 entries per part (what a lane's two loads cover: the scalar tail runs); 4x4
    # with 1 + 7 electrons: S = 11440, 715 groups per major index (a second round of group pointers in the near pass), 91 KB of LDS
    bonds = lattices.triangular(lx, ly) if len(shape) > 4 else lattices.square(lx, ly)
    P = q.csr_mat.hubbard(n, nu, nd, bonds, t=1.0, U=1.1, opts=q.make_opts(kron_split=0))      # the default format (coded + real fast path), unsplit
    assert P.info().kron_minor == 0 and P.info().value_dict > 0
    # "2g": the general form of both parts (a Hubbard operator is T (x) 1 + 1 (x) T' + D and would be kept as T, T', D)
    # "2t" (the default): both parts recognised -> the row-staged table kernel applies T, T' and the diagonal codes (minor size >= 256)
    K = q.csr_mat.hubbard(n, nu, nd, bonds, t=1.0, U=1.1, opts=q.make_opts(kron_coded=int(form[0]), kron_uniform={"2g": 0, "2t": 7}.get(form, 3)))
    assert K.info().kron_table_kernel == (1 if form == "2t" and math.comb(n, nd) >= 256 else 0)
    ik = K.info()
    assert ik.kron_minor > 0 and ik.kron_band in (2, 4, 8, 16) and 0 < ik.kron_far_nnz < ik.nnz and ik.value_dict > 0
    assert ik.kron_sliced == (0 if form == "1" else 1) and (form == "1" or ik.kron_band == 16)
    if form in ("2", "2t"):          # T alone: one short row per major index instead of one per row
        assert ik.kron_far_nnz <= 40 * int(round(ik.nrows / ik.kron_minor))
    rk, rp = q.locate_E0_lanczos(K), q.locate_E0_lanczos(P)
    assert abs(rk.E0 - rp.E0) <= 1e-12 * abs(rp.E0) and abs(rk.steps["E0"] - rp.steps["E0"]) <= 1
    assert abs(abs(np.vdot(rk.eigenvecs, rp.eigenvecs)) - 1.0) < 1e-8
    sk, sp = K.stats(), P.stats()
    assert sk.n_spmv_real > 0 and abs(sk.n_spmv_real - sp.n_spmv_real) <= 2   # both ran the all-real form (step counts may differ by one)
    ia, ja, val = P.download()
    O = qo.Csr(P.dim, ia, ja.astype(np.int64), val, False)
    x = rk.eigenvecs
    assert np.abs(O.multmv(x) - rk.E0 * x).max() < 1e-7
    K.destroy()
    P.destroy()
