"""The BASELINE configurations that round 1 only ran through bench.py, inside the GPU test tier at FULL size:

  C2  Heisenberg kagome 30 sites, Sz = 0                       dim 155,117,520   nnz 4.97e9
  C4  Fermi-Hubbard 4x5, N_up = N_dn = 5 (half filling is 3.4e10-dim: SURVEY 8d)  dim 240,374,016  nnz 7.83e9
      + the same lattice at N_up = N_dn = 3 row-sharded over 4 ranks (the way C4 is meant to run)
  C5  Heisenberg triangular 6x6, Sz = 0, momentum sector k = (1,0) (complex) and k = (0,0)   dim 2.52e8   nnz 1.43e10

The oracle cannot run at these sizes, so parity is carried by size-independent properties (Hermiticity, linearity, exact
structure counts), by AGREEMENT OF INDEPENDENT PATHS on the same operator (stored CSR vs matrix-free operator, Lanczos vs
the device-resident restarted Lanczos qbh_iram, complex vs packed-double vectors) and, for the triangular lattice, by the
literature value of the 36-site ground-state energy reached through the k = (0,0) sector."""
import ctypes as C
import math
import socket
import tempfile

import numpy as np
import pytest

import quantum_basis_amd as q
from quantum_basis_amd import _lib, lattices

pytestmark = pytest.mark.gpu


def _herm_lin(A, complex_x=False):
    """<x, H y> = <H x, y> and H(2x - 0.5i y) = 2 Hx - 0.5i Hy on random vectors; returns |Hx|"""
    n = A.dim
    v = A.vec(5)
    A.randomize(v.at(0), 1)
    A.randomize(v.at(n), 2)
    if complex_x:
        A.axpy_norm(0.75j, v.at(n), v.at(0))
    A.spmv(v.at(0), v.at(2 * n))
    A.spmv(v.at(n), v.at(3 * n))
    lhs, rhs = A.dotc(v.at(0), v.at(3 * n)), A.dotc(v.at(2 * n), v.at(n))
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), 1e-3)
    hx = A.nrm2(v.at(2 * n))
    A.spmv(v.at(0), v.at(4 * n), 0.0, 0.0, 0.0)
    A.axpy_norm(2.0, v.at(0), v.at(4 * n))
    A.axpy_norm(-0.5j, v.at(n), v.at(4 * n))
    A.spmv(v.at(4 * n), v.at(0))
    A.axpy_norm(-2.0, v.at(2 * n), v.at(0))
    assert np.sqrt(A.axpy_norm(0.5j, v.at(3 * n), v.at(0))) <= 1e-12 * hx
    v.free()
    return hx


def _same_y(A, B, seed=7):
    n = A.dim
    v = A.vec(3)
    A.randomize(v.at(0), seed)
    A.spmv(v.at(0), v.at(n))
    A.sync()
    B.spmv(v.at(0), v.at(2 * n))
    B.sync()
    hx = A.nrm2(v.at(n))
    assert np.sqrt(A.axpy_norm(-1.0, v.at(n), v.at(2 * n))) <= 1e-13 * hx
    v.free()


def _packed_lanczos_e0(M, maxit=600):
    v = M.vec(1)
    _lib.check(_lib.lib().qbh_vec_randomize_real(M.handle, v.ptr, C.c_uint32(1)), "qbh_vec_randomize_real")
    hess = np.zeros(2 * maxit)
    m = q.lanczos_real(0, maxit - 1, maxit, M, v, hess)
    v.free()
    return q.hess_eigen(hess, maxit, m, "sr")[0][0], m


def test_c2_kagome_30_full_size():
    bonds = lattices.kagome(5, 2)
    A = q.csr_mat.heisenberg(30, 15, bonds, J=1.0)
    n = A.dim
    assert n == math.comb(30, 15) == 155117520
    # every bond flips C(28, 14) states in each direction: nnz = dim + 2 * n_bonds * C(28, 14)
    nb = len(np.asarray(bonds).reshape(-1, 2))
    assert nb == 60 and A.nnz == n + 2 * nb * math.comb(28, 14)
    _herm_lin(A)
    M = q.csr_mat.heisenberg(30, 15, bonds, J=1.0, matrix_free=True)
    _same_y(A, M)
    # E0 by three independent paths: stored CSR + complex driver, matrix-free + packed-double vectors, qbh_iram
    r = q.locate_E0_lanczos(A, nev=1, ncv=0, maxit=800)
    e_mf, m_mf = _packed_lanczos_e0(M, 800)
    assert abs(r.E0 - e_mf) <= 1e-12 * abs(e_mf) and abs(r.steps["E0"] - m_mf) <= 1
    nconv, w, _ = q.iram(n, A, None, 1, 24, 300, "sr")
    assert nconv >= 1 and abs(w[0] - r.E0) <= 1e-10 * abs(r.E0)
    # bracket: each bond's S.S >= -3/4, and the Neel-type product state |up/down> bound is not needed: E0 < 0
    assert -0.75 * nb <= r.E0 < -0.25 * nb * 0.5
    A.destroy()
    M.destroy()


def test_c4_substitute_hubbard_4x5_n5_full_size():
    bonds = lattices.square(4, 5)
    A = q.csr_mat.hubbard(20, 5, 5, bonds, t=1.0, U=1.1)
    n = A.dim
    nc = math.comb(20, 5)
    assert n == nc * nc == 240374016
    nb = len(np.asarray(bonds).reshape(-1, 2))
    assert nb == 40 and A.nnz == n + 2 * (2 * nb * math.comb(18, 4)) * nc      # hops per species x the other species' configurations
    assert 0 < A.info().value_dict <= 256
    _herm_lin(A)
    M = q.csr_mat.hubbard(20, 5, 5, bonds, t=1.0, U=1.1, matrix_free=True)
    _same_y(A, M)
    r = q.locate_E0_lanczos(A, nev=1, ncv=0, maxit=800)
    e_mf, m_mf = _packed_lanczos_e0(M, 800)
    assert abs(r.E0 - e_mf) <= 1e-12 * abs(e_mf) and abs(r.steps["E0"] - m_mf) <= 1
    nconv, w, _ = q.iram(n, M, None, 1, 24, 300, "sr")
    assert nconv >= 1 and abs(w[0] - r.E0) <= 1e-10 * abs(r.E0)
    # U*D >= 0: E0 >= E0(U = 0) = 2 * (sum of the 5 lowest levels of -2t(cos kx + cos ky) on 4x5)
    eps = sorted(-2.0 * (math.cos(2 * math.pi * a / 4) + math.cos(2 * math.pi * b / 5)) for a in range(4) for b in range(5))
    assert 2 * sum(eps[:5]) - 1e-9 <= r.E0 <= 2 * sum(eps[:5]) + 1.1 * 20 * (5 / 20) ** 2 + 1e-9
    A.destroy()
    M.destroy()


def test_c4_substitute_in_the_north_star_format_split_in_place():
    """The same operator as complex128 CSR (157 GB: 7.83e9 x 20 B), split in place with 2-byte columns (S = C(20, 5) = 15504): the
    form bench.py --workload hubbard_4x5_n5 times.  y = Hx against the matrix-free operator at full size, E0 against its Lanczos."""
    bonds = lattices.square(4, 5)
    A = q.csr_mat.hubbard(20, 5, 5, bonds, t=1.0, U=1.1, opts=q.make_opts(value_dict=0, real_fast_path=0))
    info = A.info()
    assert info.kron_minor == math.comb(20, 5) and info.kron_inplace == 1 and info.kron_sliced == 1 and info.kron_cols16 == 3
    # 16 B of values + 2 B of columns per nonzero (+ row / group pointers and descriptors): 18 B, not the 20 B of the int32 form
    assert info.value_dict == 0 and info.nnz * 18 <= info.bytes_matrix < info.nnz * 18 + 64 * info.nrows
    M = q.csr_mat.hubbard(20, 5, 5, bonds, t=1.0, U=1.1, matrix_free=True)
    _same_y(A, M)
    r = q.locate_E0_lanczos(A, nev=1, ncv=0, maxit=800)
    e_mf, m_mf = _packed_lanczos_e0(M, 800)
    assert abs(r.E0 - e_mf) <= 1e-12 * abs(e_mf) and abs(r.steps["E0"] - m_mf) <= 1
    A.destroy()
    M.destroy()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_c4_family_row_sharded_over_four_ranks():
    """Hubbard 4x5, N_up = N_dn = 3 (dim 1,299,600) on 4 ranks that share this box's GPU and exchange through gloo:
    the sharded stored-CSR path gives the single-GPU and the matrix-free answer."""
    import torch.multiprocessing as mp
    import dist_worker
    world = 4
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(dist_worker.gpu_sharded_solver, args=(world, _free_port(), "gloo", tmp, False, "4x5n3"), nprocs=world, join=True)
        res = np.load(tmp + "/res.npy")
        vec = np.concatenate([np.load(tmp + "/vec_%d.npy" % r) for r in range(world)])
    bonds = lattices.square(4, 5)
    A = q.csr_mat.hubbard(20, 3, 3, bonds)
    assert A.dim == math.comb(20, 3) ** 2 == 1299600
    ref = q.locate_E0_lanczos(A, nev=2, ncv=1, maxit=400)
    M = q.csr_mat.hubbard(20, 3, 3, bonds, matrix_free=True)
    e_mf, _ = _packed_lanczos_e0(M, 400)
    assert abs(res[0] - ref.E0) <= 1e-10 * abs(ref.E0) and abs(res[0] - e_mf) <= 1e-10 * abs(e_mf)
    assert abs(res[1] - ref.E1) < 1e-8 and abs(res[2] - ref.steps["E0"]) <= 1
    assert abs(abs(np.vdot(vec, ref.eigenvecs)) - 1.0) < 1e-8
    A.destroy()
    M.destroy()


def _sector(k):
    perms, shifts = lattices.translations(6, 6)
    return q.csr_mat.heisenberg_repr(36, 18, lattices.triangular(6, 6), perms, lattices.characters(shifts, k, (6, 6)))


def test_c5_triangular_6x6_sz0_momentum_sector_full_size():
    """k = (1,0): dim 252,091,362, nnz 1.4253e10, genuinely complex values and vectors (2-byte value codes)."""
    A = _sector((1, 0))
    n = A.dim
    assert n == 252091362 and A.nnz == 14253402130
    assert 256 < A.info().value_dict <= 65536
    # structure of a momentum sector: decoupled zero-norm rows carry only the fake diagonal 100 + i/dim (src/model.cc:737-740)
    ia, ja, val = A.download(0, 200000)
    lens = np.diff(ia)
    single = np.flatnonzero(lens == 1)
    assert np.all(ja[ia[single]] == single)
    fake = val[ia[single]]
    is_fake = fake.real >= 99.0
    # the row index i enters as a REAL offset: diagonal = fake_pos + i / dim
    assert is_fake.any() and np.allclose(fake[is_fake].real, 100.0 + single[is_fake] / n, rtol=1e-14, atol=0) and np.all(fake[is_fake].imag == 0)
    assert np.abs(val.imag).max() > 0.05
    _herm_lin(A, complex_x=True)
    # 60 Lanczos steps on the complex sector (to convergence: ~180 steps = 25 s of the GPU tier's budget for an energy nothing pins;
    # the lowest Ritz value of a Krylov space is a rigorous upper bound of the sector's minimum, which lies above the ground state).
    # (The device IRAM against Lanczos on a momentum sector is checked at dim 1.5e8 in test_gpu_fullsize and at small sizes in
    # test_gpu_parity.)
    maxit = 64
    v, hess = A.vec(2), np.zeros(2 * maxit)
    A.randomize(v.at(0), 1)
    m = q.lanczos(0, 60, maxit, n, A, None, hess, "sr_val0", device_v=v)
    v.free()
    assert 59 <= m <= 61
    e_k10 = q.hess_eigen(hess, maxit, m, "sr")[0][0]
    assert -0.75 * 108 <= e_k10 < 0.0
    A.destroy()
    # k = (0,0) holds the ground state of the 36-site triangular antiferromagnet: E0/N = -0.5603734 (Bernu, Lecheminant,
    # Lhuillier, Pierre, PRB 50, 10048 (1994)); the full Sz = 0 sector (dim 9.08e9, matrix-free, round 1) gave -20.173442240311
    B = _sector((0, 0))
    assert B.dim == n
    rb = q.locate_E0_lanczos(B, nev=1, ncv=0, maxit=600)
    assert abs(rb.E0 / 36 - (-0.5603734)) < 5e-8
    assert abs(rb.E0 - (-20.173442240311)) < 1e-9
    assert rb.E0 < e_k10                                   # every Ritz value of the k = (1,0) sector lies above the ground state
    B.destroy()


def test_c4_as_written_half_filling_sector_full_size():
    """BASELINE configs[3] AS WRITTEN: Fermi-Hubbard 4x5 at half filling (N_up = N_dn = 10, 3.4e10 basis states) through its
    k = (0,0) momentum sector, 1,706,742,160 representatives, on ONE GPU in matrix-free form (qbh_mf_hubbard_repr).
      * y = Hx of the matrix-free operator against ONE stored row shard of the same sector (qbh_gen_hubbard_repr, shard 0 of
        4: 4.3e8 rows, 1.7e10 nonzeros as columns + 1-byte codes) on the shard's rows, to 1e-13 |y|;
      * the packed-double Lanczos to E0, which must equal the round-2 builder run (profiles/r2_sectors/
        c4_hubbard_4x5_half_all_sectors_ONE_gpu_matrix_free.txt) to 1e-11, lie below the ground-state energy recorded there
        for every other sector class, and inside the two rigorous bounds that need no computation of ours:
        2 E_free <= E0 <= 2 E_free + U N_up N_dn / N   (U >= 0; the paramagnetic plane-wave Slater determinant)."""
    Lx, Ly, n, nu = 4, 5, 20, 10
    bonds = lattices.square(Lx, Ly)
    perms, shifts = lattices.translations(Lx, Ly)
    chars = lattices.characters(shifts, (0, 0), (Lx, Ly))
    M = q.csr_mat.hubbard_repr_mf(n, nu, nu, bonds, perms, chars, t=1.0, U=1.1)
    dim = M.info().ncols
    assert dim == 1706742160
    S = q.csr_mat.hubbard_repr(n, nu, nu, bonds, perms, chars, t=1.0, U=1.1, shard=(0, 4))
    i = S.info()
    assert i.ncols == dim and i.row_offset == 0 and i.nrows == 426685540
    assert M.info().basis_internal == _lib.BASIS_SECTOR_ORBIT                    # rows orbit by orbit of the up patterns
    v, c = M.vec(2), M.vec(1)
    M.randomize(v.at(0), 11)
    M.spmv(v.at(0), v.at(dim))
    M.sync()
    ys = S.vec()
    M.from_internal(c.ptr, v.at(0))              # the stored shard has the caller's (ascending) order
    M.sync()                                     # every handle works on its own stream
    S.spmv(c.ptr, ys.ptr)                        # unsharded convention: x is the full-length vector
    S.sync()
    hy = S.nrm2(ys.ptr)
    assert hy > 0.0
    M.from_internal(c.ptr, v.at(dim))
    M.sync()
    assert np.sqrt(S.axpy_norm(-1.0, ys.ptr, c.ptr)) <= 1e-13 * hy               # rows [0, nrows) of the matrix-free result
    ys.free()
    c.free()
    v.free()
    S.destroy()
    e0, m = _packed_lanczos_e0(M, maxit=600)
    M.destroy()
    recorded = -27.029212101320                  # k = (0,0), 195 steps (round 2, same operator, same start vector)
    assert abs(e0 - recorded) <= 1e-11 * abs(recorded), (e0, m)
    others = {(2, 0): -26.652254639705, (0, 1): -26.993024675302, (0, 2): -26.623567801498, (1, 0): -26.431798738331,
              (2, 1): -26.337053630786, (2, 2): -26.674629340360, (1, 1): -26.823077903905, (1, 2): -26.811730670669}
    assert all(e0 < e for e in others.values())
    # one-particle levels of the 4x5 torus with the bond list actually used (H = -t sum over bonds of c+c + h.c.)
    h1 = np.zeros((n, n))
    for a, b in bonds:
        h1[a, b] -= 1.0
        h1[b, a] -= 1.0
    e_free = 2.0 * np.sort(np.linalg.eigvalsh(h1))[:nu].sum()
    assert e_free - 1e-9 <= e0 <= e_free + 1.1 * nu * nu / n + 1e-9, (e_free, e0)
