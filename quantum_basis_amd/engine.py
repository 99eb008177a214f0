"""Host-side mirror of the reference's operator / solver interface for the hot path.

Same names, argument meaning and error behaviour as the reference so that the
parity tests read like the reference's own tests:

    csr_mat.MultMv / MultMv2          src/sparse.cc:262-297
    lanczos(k, np, maxit, ...)        src/lanczos.cc:134-266   (sr_val0, sr_val1, dnmcs)
    eigenvec_CG(...)                  src/lanczos.cc:281-341
    hess_eigen(...)                   src/lanczos.cc:355-390
    iram(...)                         src/lanczos.cc:497-603   (ARPACK reverse communication)
    vec_randomize(n, seed)            src/miscellaneous.cc:371-386
    locate_E0_lanczos / locate_E0_iram  src/model.cc:1123-1366 (drivers; here free functions on a csr_mat)

All numerics run in libqbhip.so on the GPU; numpy arrays are only the host
containers the reference keeps in std::vector<T>.  Nothing here calls oracle/.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib

lanczos_precision = 2e-12      # src/miscellaneous.cc:47
sparse_precision = 1e-14       # src/miscellaneous.cc:46
machine_prec = np.finfo(np.float64).eps


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _cvec(a, name):
    if not (isinstance(a, np.ndarray) and a.dtype == np.complex128 and a.flags.c_contiguous):
        raise ValueError("%s must be a C-contiguous complex128 numpy array" % name)
    return a


def make_opts(device=-1, stream=None, spmv_kernel=_lib.KERNEL_AUTO, nnz_per_block=0, xcd_swizzle=2,
              value_dict=1, profile=0, check_hermitian=1, real_fast_path=1, kron_split=1, kron_minor=0, deterministic=0,
              basis_kind=0, n_sites=0, n_up=0, n_dn=0, **more):
    """qbh_opts with the library's defaults; `more` names any further field of the struct (include/qbhip.h), e.g. kron_cols16=0."""
    o = _lib.Opts()
    lib().qbh_opts_default(C.byref(o))
    o.device = device
    o.stream = stream
    o.spmv_kernel = spmv_kernel
    o.nnz_per_block = nnz_per_block
    o.xcd_swizzle = xcd_swizzle
    o.value_dict = value_dict
    o.profile = profile
    o.check_hermitian = check_hermitian
    o.real_fast_path = real_fast_path
    o.kron_split = kron_split
    o.kron_minor = kron_minor
    o.deterministic = deterministic
    o.basis_kind = basis_kind
    o.n_sites, o.n_up, o.n_dn = n_sites, n_up, n_dn
    known = {f[0] for f in _lib.Opts._fields_}
    for k, v in more.items():
        if k not in known:
            raise TypeError("make_opts: qbh_opts has no field %r" % k)
        setattr(o, k, v)
    return o


class DeviceVec:
    """n complex128 in HBM (the callers' std::vector<T>, src/model.cc:1158-1164)."""

    def __init__(self, mat, n):
        self.mat = mat
        self.n = int(n)
        self.ptr = C.c_void_p()
        check(lib().qbh_vec_alloc(C.byref(self.ptr), C.c_int64(self.n)), "qbh_vec_alloc")

    def at(self, offset):
        """Device address of element `offset` (vectors are packed back to back like v + j*dim)."""
        return C.c_void_p(self.ptr.value + 16 * int(offset))

    def upload(self, h, offset=0):
        h = np.ascontiguousarray(h, dtype=np.complex128)
        check(lib().qbh_vec_upload(self.mat.handle, self.at(offset), _p(h), C.c_int64(h.size)), "qbh_vec_upload")

    def download(self, offset=0, n=None):
        n = self.n - offset if n is None else n
        h = np.empty(n, dtype=np.complex128)
        check(lib().qbh_vec_download(self.mat.handle, _p(h), self.at(offset), C.c_int64(n)), "qbh_vec_download")
        return h

    def free(self):
        if self.ptr:
            lib().qbh_vec_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class csr_mat:
    """Device-resident counterpart of csr_mat<std::complex<double>> (src/qbasis.h:976-1021).

    Built from the reference's host layout: zero-based ia[dim+1], ja[nnz] (int64),
    val[nnz] (complex128); sym=True means only the upper triangle is stored.
    """

    def __init__(self, dim, ia, ja, val, sym=True, opts=None, _handle=None, rows=None):
        """rows=(r0, r1): build only that row block of the operator from the same host arrays (qbh_csr_create_rows,
        one rank of a row-sharded run); None: the whole operator (qbh_csr_create)."""
        _lib.require_gpu()
        self.handle = C.c_void_p()
        self._opts = opts if opts is not None else make_opts()
        if _handle is not None:
            self.handle = _handle
            self.ia = self.ja = self.val = None
            self.sym = False
        else:
            self.ia = np.ascontiguousarray(ia, dtype=np.int64)
            self.ja = np.ascontiguousarray(ja, dtype=np.int64)
            self.val = np.ascontiguousarray(val, dtype=np.complex128)
            self.sym = bool(sym)
            nnz = int(self.ia[-1]) if self.ia.size else 0
            if rows is None:
                check(lib().qbh_csr_create(C.byref(self.handle), C.c_int64(dim), C.c_int64(nnz), int(self.sym),
                                           _p(self.ia), _p(self.ja), _p(self.val), C.byref(self._opts)),
                      "qbh_csr_create")
            else:
                check(lib().qbh_csr_create_rows(C.byref(self.handle), C.c_int64(dim), C.c_int64(nnz), int(self.sym),
                                                _p(self.ia), _p(self.ja), _p(self.val), C.c_int64(rows[0]),
                                                C.c_int64(rows[1]), C.byref(self._opts)), "qbh_csr_create_rows")
        info = self.info()
        self.dim = info.nrows          # shard-local length (== global dim when unsharded)
        self.ncols = info.ncols
        self.row_offset = info.row_offset
        self.nnz = info.nnz
        self._comm = None

    # ---- generators (measurement harness) ------------------------------------------------
    @classmethod
    def hubbard(cls, n_sites, n_up, n_dn, bonds, t=1.0, U=1.1, rows=None, opts=None, matrix_free=False):
        """Device-built Fermi-Hubbard operator: CSR in HBM (default) or, with matrix_free=True, applied from
        the hop tables without a stored matrix (qbh_mf_hubbard)."""
        _lib.require_gpu()
        opts = opts if opts is not None else make_opts()
        b = np.ascontiguousarray(np.asarray(bonds, dtype=np.int32).reshape(-1, 2))
        r0, r1 = (0, -1) if rows is None else rows
        h = C.c_void_p()
        fn = lib().qbh_mf_hubbard if matrix_free else lib().qbh_gen_hubbard
        check(fn(C.byref(h), n_sites, n_up, n_dn, len(b), _p(b), t, U, C.c_int64(r0),
                 C.c_int64(r1), C.byref(opts)), "qbh_mf_hubbard" if matrix_free else "qbh_gen_hubbard")
        return cls(0, None, None, None, opts=opts, _handle=h)

    @classmethod
    def heisenberg(cls, n_sites, n_dn, bonds, J=1.0, rows=None, opts=None, matrix_free=False):
        """Spin-1/2 Heisenberg sector with n_dn down spins, assembled on the device (qbh_gen_heisenberg) or applied
        without a stored matrix (matrix_free=True: qbh_mf_heisenberg, same basis)."""
        _lib.require_gpu()
        opts = opts if opts is not None else make_opts()
        b = np.ascontiguousarray(np.asarray(bonds, dtype=np.int32).reshape(-1, 2))
        r0, r1 = (0, -1) if rows is None else rows
        h = C.c_void_p()
        fn = lib().qbh_mf_heisenberg if matrix_free else lib().qbh_gen_heisenberg
        check(fn(C.byref(h), n_sites, n_dn, len(b), _p(b), J, C.c_int64(r0), C.c_int64(r1), C.byref(opts)),
              "qbh_mf_heisenberg" if matrix_free else "qbh_gen_heisenberg")
        return cls(0, None, None, None, opts=opts, _handle=h)

    @classmethod
    def heisenberg_repr(cls, n_sites, n_dn, bonds, perms, chars, J=1.0, fake_pos=100.0, shard=(0, 1), opts=None, row_cuts=None):
        """Translation-symmetric sector assembled on the device (qbh_gen_heisenberg_repr, counterpart of
        model::generate_Ham_sparse_repr): perms[g] = site images under translation g (g = 0 identity), chars[g] =
        momentum character chi_k(g).  shard = (rank, world) selects a uniform row block of the sector (its dimension
        is only known after enumeration: read it from .info().ncols); row_cuts = [0, ..., dim] (world + 1 entries) replaces
        the uniform blocks by the caller's (qbh_gen_heisenberg_repr_cuts; see dist.rebalance_cuts)."""
        _lib.require_gpu()
        opts = opts if opts is not None else make_opts()
        b = np.ascontiguousarray(np.asarray(bonds, dtype=np.int32).reshape(-1, 2))
        p = np.ascontiguousarray(np.asarray(perms, dtype=np.int32))
        c = np.ascontiguousarray(np.asarray(chars, dtype=np.complex128))
        assert p.shape == (len(c), n_sites)
        h = C.c_void_p()
        dim = C.c_int64(0)
        cuts = None if row_cuts is None else np.ascontiguousarray(row_cuts, dtype=np.int64)
        assert cuts is None or cuts.size == int(shard[1]) + 1
        check(lib().qbh_gen_heisenberg_repr_cuts(C.byref(h), n_sites, n_dn, len(b), _p(b), J, len(c), _p(p), _p(c), fake_pos,
                                                 int(shard[0]), int(shard[1]), _p(cuts) if cuts is not None else None,
                                                 C.byref(dim), C.byref(opts)),
              "qbh_gen_heisenberg_repr")
        return cls(0, None, None, None, opts=opts, _handle=h)

    @classmethod
    def hubbard_repr(cls, n_sites, n_up, n_dn, bonds, perms, chars, t=1.0, U=1.1, fake_pos=100.0, shard=(0, 1), opts=None,
                     terms=None, pairs=None, exchange=None, no_double=False, row_cuts=None):
        """Hubbard family in a translation-symmetric sector, assembled on the device (qbh_gen_hubbard_repr; counterpart of
        model::enumerate_basis_repr + generate_Ham_sparse_repr for the reference's
        examples/trans_symmetric/latt_square/square_Fermi_Hubbard.cc).  Default operator: -t sum_<ij>,sigma (c+_i c_j + h.c.)
        + U sum_i n_up n_dn over `bonds` (a bond listed twice counts twice, as in the reference's 4x2 torus).  terms =
        [(i, j, amp_up, amp_dn), ...] replaces the hopping part by explicit directed one-body terms amp * c+_i c_j (they
        must form a translation-invariant operator; U still applies -- pass U=0 for a pure one-body observable).  pairs =
        [(i, j, v_uu, v_ud, v_du, v_dd), ...] adds density-density terms v * n_{i,s} n_{j,s'} (extended Hubbard; spinless t-V
        with n_dn = 0).  exchange = [(i, j, a), ...] adds a * (S+_i S-_j + S-_i S+_j); no_double=True removes the words
        with doubly occupied sites and projects the hopping (t-J model: see tj_repr)."""
        _lib.require_gpu()
        opts = opts if opts is not None else make_opts()
        exchange = exchange or []
        xs = np.ascontiguousarray(np.array([[a[0], a[1]] for a in exchange], dtype=np.int32).reshape(-1, 2))
        xa = np.ascontiguousarray(np.array([a[2] for a in exchange], dtype=np.float64))
        if terms is None:
            terms = []
            for (i, j) in np.asarray(bonds, dtype=np.int64).reshape(-1, 2):
                terms.append((int(i), int(j), -t, -t))
                terms.append((int(j), int(i), -t, -t))
        pairs = pairs or []
        psites = np.ascontiguousarray(np.array([[a[0], a[1]] for a in pairs], dtype=np.int32).reshape(-1, 2))
        pv = np.ascontiguousarray(np.array([a[2:6] for a in pairs], dtype=np.float64).reshape(-1, 4))
        sites = np.ascontiguousarray(np.array([[a[0], a[1]] for a in terms], dtype=np.int32).reshape(-1, 2))
        aup = np.ascontiguousarray(np.array([a[2] for a in terms], dtype=np.complex128))
        adn = np.ascontiguousarray(np.array([a[3] for a in terms], dtype=np.complex128))
        p = np.ascontiguousarray(np.asarray(perms, dtype=np.int32))
        c = np.ascontiguousarray(np.asarray(chars, dtype=np.complex128))
        assert p.shape == (len(c), n_sites)
        h = C.c_void_p()
        dim = C.c_int64(0)
        cuts = None if row_cuts is None else np.ascontiguousarray(row_cuts, dtype=np.int64)      # see heisenberg_repr
        assert cuts is None or cuts.size == int(shard[1]) + 1
        check(lib().qbh_gen_hubbard_repr_cuts(C.byref(h), n_sites, n_up, n_dn, len(terms), _p(sites), _p(aup), _p(adn), float(U),
                                              len(pairs), _p(psites) if pairs else None, _p(pv) if pairs else None,
                                              len(exchange), _p(xs) if exchange else None, _p(xa) if exchange else None,
                                              int(bool(no_double)), len(c),
                                              _p(p), _p(c), fake_pos, int(shard[0]), int(shard[1]),
                                              _p(cuts) if cuts is not None else None, C.byref(dim), C.byref(opts)),
              "qbh_gen_hubbard_repr")
        return cls(0, None, None, None, opts=opts, _handle=h)

    @classmethod
    def hubbard_repr_mf(cls, n_sites, n_up, n_dn, bonds, perms, chars, t=1.0, U=1.1, fake_pos=100.0, opts=None, terms=None, pairs=None):
        """The sector operator of hubbard_repr in matrix-free form with a small stored remainder (qbh_mf_hubbard_repr): same
        basis, same matrix, ~1000 x less memory -- one GPU holds 4x5 at half filling."""
        _lib.require_gpu()
        opts = opts if opts is not None else make_opts()
        if terms is None:
            terms = []
            for (i, j) in np.asarray(bonds, dtype=np.int64).reshape(-1, 2):
                terms.append((int(i), int(j), -t, -t))
                terms.append((int(j), int(i), -t, -t))
        pairs = pairs or []
        psites = np.ascontiguousarray(np.array([[a[0], a[1]] for a in pairs], dtype=np.int32).reshape(-1, 2))
        pv = np.ascontiguousarray(np.array([a[2:6] for a in pairs], dtype=np.float64).reshape(-1, 4))
        sites = np.ascontiguousarray(np.array([[a[0], a[1]] for a in terms], dtype=np.int32).reshape(-1, 2))
        aup = np.ascontiguousarray(np.array([a[2] for a in terms], dtype=np.complex128))
        adn = np.ascontiguousarray(np.array([a[3] for a in terms], dtype=np.complex128))
        p = np.ascontiguousarray(np.asarray(perms, dtype=np.int32))
        c = np.ascontiguousarray(np.asarray(chars, dtype=np.complex128))
        assert p.shape == (len(c), n_sites)
        h = C.c_void_p()
        dim = C.c_int64(0)
        check(lib().qbh_mf_hubbard_repr(C.byref(h), n_sites, n_up, n_dn, len(terms), _p(sites), _p(aup), _p(adn), float(U),
                                        len(pairs), _p(psites) if pairs else None, _p(pv) if pairs else None, len(c), _p(p), _p(c),
                                        fake_pos, C.byref(dim), C.byref(opts)), "qbh_mf_hubbard_repr")
        return cls(0, None, None, None, opts=opts, _handle=h)

    @classmethod
    def tj_repr(cls, n_sites, n_up, n_dn, bonds, perms, chars, t=1.0, J=1.0, **kw):
        """t-J model in a momentum sector: -t P sum (c+_i c_j + h.c.) P + J sum (S_i.S_j - n_i n_j / 4) without doubly
        occupied sites (examples/trans_symmetric/latt_kagome/kagome_tJ.cc:98-104)."""
        b = [(int(i), int(j)) for (i, j) in np.asarray(bonds, dtype=np.int64).reshape(-1, 2)]
        pairs = [(i, j, 0.0, -0.5 * J, -0.5 * J, 0.0) for (i, j) in b]            # J Sz Sz - J n n / 4
        exch = [(i, j, 0.5 * J) for (i, j) in b]
        return cls.hubbard_repr(n_sites, n_up, n_dn, b, perms, chars, t=t, U=0.0, pairs=pairs, exchange=exch, no_double=True, **kw)

    # ---- reference interface -------------------------------------------------------------
    def dimension(self):
        return self.dim

    def info(self):
        i = _lib.CsrInfo()
        check(lib().qbh_csr_get_info(self.handle, C.byref(i)), "qbh_csr_get_info")
        return i

    def MultMv2(self, x, y):
        """y = H * x + y (host vectors; src/sparse.cc:262-289)."""
        _cvec(x, "x"), _cvec(y, "y")
        check(lib().qbh_multmv2(self.handle, _p(x), _p(y)), "qbh_multmv2")

    def MultMv(self, x, y):
        """y = H * x (host vectors; src/sparse.cc:291-297)."""
        _cvec(x, "x"), _cvec(y, "y")
        check(lib().qbh_multmv(self.handle, _p(x), _p(y)), "qbh_multmv")

    def to_dense(self):
        """Column-major dense copy from the host arrays (src/sparse.cc:299-315); only used for dim <= 30."""
        if self.ia is None:
            raise RuntimeError("to_dense needs the host CSR arrays")
        d = np.zeros((self.dim, self.dim), dtype=np.complex128, order="F")
        for row in range(self.dim):
            for pt in range(self.ia[row], self.ia[row + 1]):
                col = self.ja[pt]
                d[row, col] = self.val[pt]
                if self.sym and row != col:
                    d[col, row] = np.conj(self.val[pt])
        return d

    def destroy(self):
        if self.handle:
            lib().qbh_csr_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # ---- device-level helpers ------------------------------------------------------------
    def vec(self, nvec=1):
        return DeviceVec(self, nvec * self.dim)

    def spmv(self, d_x, d_y, alpha=1.0, beta=0.0, gamma=0.0, want_red=False):
        red = (C.c_double * 3)()
        check(lib().qbh_spmv_dev(self.handle, d_x, d_y, alpha, beta, gamma, red if want_red else None),
              "qbh_spmv_dev")
        return (complex(red[0], red[1]), red[2]) if want_red else None

    def dotc(self, d_x, d_y):
        r = (C.c_double * 2)()
        check(lib().qbh_dotc_dev(self.handle, d_x, d_y, r), "qbh_dotc_dev")
        return complex(r[0], r[1])

    def axpy_norm(self, alpha, d_x, d_y):
        r = C.c_double()
        a = complex(alpha)
        check(lib().qbh_axpy_norm_dev(self.handle, _lib.Z(a.real, a.imag), d_x, d_y, C.byref(r)),
              "qbh_axpy_norm_dev")
        return r.value

    def scal(self, a, d_x):
        check(lib().qbh_scal_dev(self.handle, float(a), d_x), "qbh_scal_dev")

    def nrm2(self, d_x):
        r = C.c_double()
        check(lib().qbh_nrm2_dev(self.handle, d_x, C.byref(r)), "qbh_nrm2_dev")
        return r.value

    def randomize(self, d_x, seed):
        check(lib().qbh_vec_randomize(self.handle, d_x, C.c_uint32(seed)), "qbh_vec_randomize")

    def to_internal(self, d_dst, d_src):
        """d_dst (this operator's internal order) <- d_src (the caller's order): device vectors of an operator with
        info().basis_internal != 0 are in the internal order (qbh_vec_to_internal); a plain copy otherwise."""
        check(lib().qbh_vec_to_internal(self.handle, d_dst, d_src), "qbh_vec_to_internal")

    def from_internal(self, d_dst, d_src):
        check(lib().qbh_vec_from_internal(self.handle, d_dst, d_src), "qbh_vec_from_internal")

    def sync(self):
        check(lib().qbh_sync(self.handle), "qbh_sync")

    def major_order(self, n_major=None):
        """qbh_csr_major_order: for an operator generated with qbh_opts.major_partition, the generator's (ascending pattern) major index
        of every major index of this operator.  n_major: the number of major indices (up configurations) -- taken from the split's
        minor size where the operator is split in place, needed from the caller where it is not."""
        n = int(n_major) if n_major else (int(self.info().ncols // self.info().kron_minor) if self.info().kron_minor else 0)
        if n == 0:
            raise ValueError("major_order: the operator is not split in place: name n_major")
        out = np.empty(n, dtype=np.int32)
        check(lib().qbh_csr_major_order(self.handle, _p(out), C.c_int64(n)), "qbh_csr_major_order")
        return out

    def set_option(self, name, value):
        """qbh_csr_set_option: lanczos_pipeline / profile / tile_fold / comm_reserve on an existing operator."""
        check(lib().qbh_csr_set_option(self.handle, name.encode(), int(value)), "qbh_csr_set_option")

    def stats(self, reset=False):
        s = _lib.Stats()
        check(lib().qbh_get_stats(self.handle, C.byref(s), int(reset)), "qbh_get_stats")
        return s

    def reference_order(self, kind, n_sites, n_up, n_dn, opts=None):
        """The same operator in the reference's basis order and fermion convention, permuted on the device
        (qbh_csr_reference_order; kind 0 = spin-1/2 basis of csr_mat.heisenberg, 1 = electron basis of csr_mat.hubbard;
        src/basis.cc:1144-1190, src/model.cc:665-670).  Needs complex128 values (value_dict = 0)."""
        h = C.c_void_p()
        o = opts if opts is not None else make_opts(value_dict=0, real_fast_path=0)
        check(lib().qbh_csr_reference_order(C.byref(h), self.handle, int(kind), int(n_sites), int(n_up), int(n_dn), C.byref(o)),
              "qbh_csr_reference_order")
        return csr_mat(self.dim, None, None, None, _handle=h)

    def set_basis(self, basis_kind, n_sites, n_up, n_dn):
        """Tell an existing plain CSR what its index means (qbh_csr_set_basis; _lib.BASIS_REF_FERMION2 = the reference's order of a
        two-species fermion basis): it is then held species-major internally (Kronecker split), vectors are translated at
        upload / download / randomize.  Returns True when the hint described the matrix."""
        check(lib().qbh_csr_set_basis(self.handle, int(basis_kind), int(n_sites), int(n_up), int(n_dn)), "qbh_csr_set_basis")
        return self.info().basis_internal != 0

    def download(self, r0=0, r1=None, values=True):
        """Copy rows [r0, r1) of the device CSR back (tests / CPU-baseline sample)."""
        r1 = self.dim if r1 is None else r1
        ia = np.empty(r1 - r0 + 1, dtype=np.int64)
        check(lib().qbh_csr_download(self.handle, C.c_int64(r0), C.c_int64(r1), _p(ia), None, None),
              "qbh_csr_download")
        nnz = int(ia[-1])
        ja = np.empty(nnz, dtype=np.int32)
        val = np.empty(nnz, dtype=np.complex128) if values else None
        check(lib().qbh_csr_download(self.handle, C.c_int64(r0), C.c_int64(r1), None, _p(ja),
                                     _p(val) if values else None), "qbh_csr_download")
        return ia, ja, val


# ------------------------------------------------------------------------------------------
def balanced_row_cuts(dim, ia, ja, sym, nranks):
    """Row cuts of nranks blocks balanced by the nonzeros of the full operator (qbh_balanced_row_cuts)."""
    ia = np.ascontiguousarray(ia, dtype=np.int64)
    ja = np.ascontiguousarray(ja, dtype=np.int64)
    cuts = np.zeros(nranks + 1, dtype=np.int64)
    check(lib().qbh_balanced_row_cuts(C.c_int64(dim), C.c_int64(int(ia[-1])), int(bool(sym)), _p(ia), _p(ja), int(nranks),
                                      _p(cuts)), "qbh_balanced_row_cuts")
    return cuts


def vec_randomize(mat, n=None, seed=1):
    """Host start vector produced on the device (src/miscellaneous.cc:371-386)."""
    dv = mat.vec()
    mat.randomize(dv.ptr, seed)
    out = dv.download()
    dv.free()
    return out


def hess_eigen(hessenberg, maxit, m, order="sr"):
    """src/lanczos.cc:355-390: returns (ritz[m], s[m*m] column-major)."""
    h = np.ascontiguousarray(hessenberg, dtype=np.float64)
    ritz = np.empty(m)
    s = np.empty(m * m)
    check(lib().qbh_hess_eigen(_p(h), C.c_int64(maxit), C.c_int64(m), order.encode(), _p(ritz), _p(s)),
          "qbh_hess_eigen")
    return ritz, s


def _solver_info(maxit, want_log=True, want_resid=False):
    info = _lib.SolverInfo()
    keep = []
    if want_log:
        buf = (_lib.LanczosRow * int(maxit))()
        info.log = C.cast(buf, C.POINTER(_lib.LanczosRow))
        info.log_cap = int(maxit)
        keep.append(buf)
    if want_resid:
        rl = (C.c_double * (int(maxit) + 1))()
        info.cg_resid = C.cast(rl, C.POINTER(C.c_double))
        keep.append(rl)
    return info, keep


def _rows(info):
    return [dict(k=info.log[i].k, ritz=list(info.log[i].ritz), a=info.log[i].a_km1, b=info.log[i].b_k,
                 accuracy=info.log[i].accuracy, accu_E0=info.log[i].accu_E0, accu_E1=info.log[i].accu_E1)
            for i in range(info.log_len)]


def lanczos(k, np_steps, maxit, dim, mat, v, hessenberg, purpose, device_v=None, state=None):
    """lanczos<T,MAT> (src/lanczos.cc:134): returns m; v and hessenberg are updated in place.

    v is a host array (2*dim, 3*dim for sr_val1) unless device_v (a DeviceVec) is given, in
    which case the vectors stay in HBM (level-2 seam).  Extra outputs on lanczos.last."""
    if dim != mat.dim:
        raise ValueError("dim mismatch")
    assert hessenberg.dtype == np.float64 and hessenberg.size >= 2 * maxit
    m = C.c_int64(0)
    info, keep = _solver_info(maxit)
    if state is not None:                      # resume exactly where a checkpoint left the stop test
        info.resume = 1
        info.cnt_accuE0 = int(state["cnt_accuE0"])
        info.accuracy, info.theta0_prev, info.theta1_prev = state["accuracy"], state["theta0_prev"], state["theta1_prev"]
    if device_v is not None:
        rc = lib().qbh_lanczos_dev(mat.handle, k, np_steps, maxit, C.byref(m), device_v.ptr, _p(hessenberg),
                                   purpose.encode(), C.byref(info))
    else:
        _cvec(v, "v")
        rc = lib().qbh_lanczos(mat.handle, k, np_steps, maxit, C.byref(m), _p(v), _p(hessenberg),
                               purpose.encode(), C.byref(info))
    check(rc, "qbh_lanczos")
    lanczos.last = dict(log=_rows(info), n_matvec=info.n_matvec, n_reorth=info.n_reorth,
                        ms_total=info.ms_total, ms_spmv=info.ms_spmv,
                        state=dict(cnt_accuE0=int(info.cnt_accuE0), accuracy=info.accuracy,
                                   theta0_prev=info.theta0_prev, theta1_prev=info.theta1_prev))
    return m.value


lanczos.last = {}


def lanczos_real(k, np_steps, maxit, mat, device_v, hessenberg, purpose="sr_val0", state=None):
    """qbh_lanczos_real_dev: the recurrence on vectors stored as packed doubles (device_v: a DeviceVec whose memory is
    read as two slots of mat.dim doubles).  Real operator, one GPU; see include/qbhip.h."""
    assert hessenberg.dtype == np.float64 and hessenberg.size >= 2 * maxit
    m = C.c_int64(0)
    info, keep = _solver_info(maxit)
    if state is not None:
        info.resume = 1
        info.cnt_accuE0 = int(state["cnt_accuE0"])
        info.accuracy, info.theta0_prev, info.theta1_prev = state["accuracy"], state["theta0_prev"], state["theta1_prev"]
    check(lib().qbh_lanczos_real_dev(mat.handle, k, np_steps, maxit, C.byref(m), device_v.ptr, _p(hessenberg),
                                     purpose.encode(), C.byref(info)), "qbh_lanczos_real_dev")
    lanczos_real.last = dict(log=_rows(info), n_matvec=info.n_matvec, ms_total=info.ms_total, ms_spmv=info.ms_spmv,
                             state=dict(cnt_accuE0=int(info.cnt_accuE0), accuracy=info.accuracy,
                                        theta0_prev=info.theta0_prev, theta1_prev=info.theta1_prev))
    return m.value


lanczos_real.last = {}


def eigenvec_CG(dim, maxit, m, mat, E0, v, r, p, pp, device=False):
    """eigenvec_CG<T,MAT> (src/lanczos.cc:281): returns (m, accu); v, r, p, pp updated in place."""
    if dim != mat.dim:
        raise ValueError("dim mismatch")
    mm = C.c_int64(m)
    accu = C.c_double(0.0)
    info, keep = _solver_info(maxit, want_log=False, want_resid=True)
    if device:
        rc = lib().qbh_eigenvec_cg_dev(mat.handle, maxit, C.byref(mm), float(np.real(E0)), C.byref(accu),
                                       v, r, p, pp, C.byref(info))
    else:
        for a in (v, r, p, pp):
            _cvec(a, "vector")
        rc = lib().qbh_eigenvec_cg(mat.handle, maxit, C.byref(mm), float(np.real(E0)), C.byref(accu),
                                   _p(v), _p(r), _p(p), _p(pp), C.byref(info))
    check(rc, "qbh_eigenvec_cg")
    eigenvec_CG.last = dict(resid=[info.cg_resid[i] for i in range(1, mm.value + 1)],
                            n_matvec=info.n_matvec, ms_total=info.ms_total)
    return mm.value, accu.value


eigenvec_CG.last = {}


def eigenvec_CG_real(maxit, m, mat, E0, v, r, p, pp):
    """qbh_eigenvec_cg_real_dev: eigenvec_CG on four device vectors of mat.dim PACKED DOUBLES (device addresses);
    returns (m, accu)."""
    mm = C.c_int64(m)
    accu = C.c_double(0.0)
    info, keep = _solver_info(maxit, want_log=False, want_resid=True)
    check(lib().qbh_eigenvec_cg_real_dev(mat.handle, maxit, C.byref(mm), float(np.real(E0)), C.byref(accu), v, r, p, pp,
                                         C.byref(info)), "qbh_eigenvec_cg_real_dev")
    eigenvec_CG_real.last = dict(resid=[info.cg_resid[i] for i in range(1, mm.value + 1)], n_matvec=info.n_matvec,
                                 ms_total=info.ms_total)
    return mm.value, accu.value


eigenvec_CG_real.last = {}


def _iram_checks(dim, nev, maxit, order):
    if nev <= 0 or nev >= dim - 1:
        raise ValueError("0 < nev < N-1 should be satisfied.")          # src/lanczos.cc:502
    if maxit < 20:
        raise ValueError("maxit should not be smaller than 20!")        # src/lanczos.cc:504
    orderC = order.upper()
    if orderC not in ("SR", "SA", "LR", "LA", "SM", "LM"):
        raise ValueError("Invalid argument orderC.")
    key = {"SR": lambda e: e, "SA": lambda e: e, "LR": lambda e: -e, "LA": lambda e: -e,
           "SM": lambda e: np.abs(e), "LM": lambda e: -np.abs(e)}[orderC]
    return orderC, key


def iram(dim, mat, v0, nev, ncv, maxit, order="sr", method="auto", seed=1):
    """iram<T,MAT> (src/lanczos.cc:497-603).  Returns (nconv, eigenvals[nev], eigenvecs[nev*dim]).

    method "device" (default where applicable: order sr/lr, ncv <= 64): the restarted Lanczos process
    runs entirely in HBM (qbh_iram, thick restart == implicit restart for a Hermitian operator).
    method "arpack": the literal reverse-communication loop of call_arpack (src/lanczos.cc:472-477) with
    ARPACK's znaupd/zneupd (scipy's bundled copy; mode 1, bmat='I', tol=0, info=0 so v0 is ignored) and
    csr_mat.MultMv on the device as the matvec -- the vectors cross PCIe on every call."""
    orderC, key = _iram_checks(dim, nev, maxit, order)
    if dim <= 30:                                                        # :508-542 dense fall-back
        w, z = np.linalg.eigh(mat.to_dense())
        idx = np.argsort(key(w), kind="stable")[:nev]
        return nev, w[idx].copy(), np.concatenate([z[:, j] for j in idx])
    if method == "auto":
        method = "device" if (orderC in ("SR", "SA", "LR", "LA") and ncv <= 64) else "arpack"
    if method == "device":
        nconv = C.c_int64(0)
        w = np.zeros(nev)
        z = np.zeros(nev * mat.dim, dtype=np.complex128)
        info, keep = _solver_info(maxit, want_log=False)
        check(lib().qbh_iram(mat.handle, nev, ncv, maxit, orderC.lower().encode(), 0.0, C.c_uint32(seed),
                             C.byref(nconv), _p(w), _p(z), C.byref(info)), "qbh_iram")
        iram.last = dict(n_matvec=info.n_matvec, restarts=info.n_reorth, ms_total=info.ms_total)
        if nconv.value <= 0:
            raise RuntimeError("nconv == 0...")                          # src/lanczos.cc:566
        return nconv.value, w, z
    return iram_arpack(dim, mat, v0, nev, ncv, maxit, order)


iram.last = {}


def iram_arpack(dim, mat, v0, nev, ncv, maxit, order="sr"):
    """ARPACK by reverse communication over the host-vector seam (csr_mat.MultMv)."""
    orderC, key = _iram_checks(dim, nev, maxit, order)
    from scipy.sparse.linalg import LinearOperator, eigs

    count = [0]

    def matvec(x):
        x = np.ascontiguousarray(x, dtype=np.complex128).reshape(-1)
        y = np.empty(dim, dtype=np.complex128)
        mat.MultMv(x, y)
        count[0] += 1
        return y

    which = {"SA": "SR", "LA": "LR"}.get(orderC, orderC)
    op = LinearOperator((dim, dim), matvec=matvec, dtype=np.complex128)
    w, z = eigs(op, k=nev, ncv=ncv, which=which, maxiter=maxit, tol=0)
    if np.max(np.abs(w.imag)) > lanczos_precision:                       # :487-492
        raise RuntimeError("eigenvalue should be real.")
    w = w.real
    idx = np.argsort(key(w), kind="stable")
    iram.last = dict(n_matvec=count[0])
    return len(idx), w[idx].copy(), np.concatenate([z[:, j] for j in idx])


def measure_full_dynamic(mat, vec_new, maxit):
    """The device part of model<T>::measure_full_dynamic (src/model.cc:1696-1712): given
    vec_new = A_q |phi> (built by the host's moprXvec_full), returns (m, norm, hessenberg) of the
    "dnmcs" Lanczos run whose continued fraction is the dynamical correlation function."""
    dim = mat.dim
    v = np.zeros(2 * dim, dtype=np.complex128)
    v[:dim] = vec_new
    norm = float(np.linalg.norm(v[:dim]))
    hessenberg = np.zeros(2 * maxit)
    if abs(norm) < lanczos_precision:
        return 0, norm, hessenberg
    v[:dim] /= norm
    m = lanczos(0, maxit - 1, maxit, dim, mat, v, hessenberg, "dnmcs")
    return m, norm, hessenberg


def energy_scale(dim, mat, extend=0.1, iters=128, seed=1):
    """energy_scale<T,MAT> (src/kpm.cc:45-88): spectral bounds for the kernel polynomial method from `iters - 1` Lanczos
    steps without a stop rule ("dnmcs" recurrence), lo / hi = extreme Ritz values widened by extend * (hi - lo)."""
    if dim != mat.dim:
        raise ValueError("dim mismatch")
    mm = iters - 1
    v = mat.vec(2)
    try:
        mat.randomize(v.at(0), seed)                                      # :60
        hess = np.zeros(2 * iters)
        m = lanczos(0, mm, iters, dim, mat, None, hess, "dnmcs", device_v=v)   # :62-76
        ritz, _ = hess_eigen(hess, iters, m, "sr")                        # :77
    finally:
        v.free()
    lo, hi = float(ritz[0]), float(ritz[m - 1])
    slack = extend * (hi - lo)
    return lo - slack, hi + slack


def moprXvec_spin(n_sites, n_dn_old, kind, coef, d_vec_old, d_vec_new, stream=None):
    """moprXvec_full (src/model.cc:1468-1538) on the device for S^z_q (kind 0), S^-_q (kind -1) and S^+_q (kind +1) on a
    spin-1/2 sector with n_dn_old down spins; d_vec_old / d_vec_new are device addresses (DeviceVec.at(...))."""
    c = np.ascontiguousarray(coef, dtype=np.complex128)
    assert c.size == n_sites
    check(lib().qbh_mopr_spin_dev(n_sites, n_dn_old, kind, _p(c), d_vec_old, d_vec_new, stream), "qbh_mopr_spin_dev")


def moprXvec_onebody(n_sites, n_up, n_dn, terms, d_vec_old, d_vec_new, stream=None):
    """A = sum_k w_k c+_{a_k, s_k} c_{b_k, s_k} applied on the device in the basis of qbh_gen_hubbard;
    terms: iterable of (a, b, spin, w)."""
    terms = list(terms)
    a = np.array([t[0] for t in terms], dtype=np.int32)
    b = np.array([t[1] for t in terms], dtype=np.int32)
    sp = np.array([t[2] for t in terms], dtype=np.int32)
    w = np.array([t[3] for t in terms], dtype=np.complex128)
    check(lib().qbh_mopr_onebody_dev(n_sites, n_up, n_dn, len(terms), _p(a), _p(b), _p(sp), _p(w), d_vec_old, d_vec_new, stream),
          "qbh_mopr_onebody_dev")


SPIN_OPS = {"Sz": 0, "S+": 1, "S-": 2}
FERMION_OPS = {"n": 0, "c+": 1, "c": 2}


def moprXvec_terms(family, n_sites, n_a_old, n_b_old, terms, d_vec_old, d_vec_new, stream=None):
    """The general moprXvec_full (src/model.cc:1468-1538; qbh_mopr_terms_dev): A = sum_t coef_t * (ordered product of elementary
    site operators), applied on the device in the generators' basis order.  terms: iterable of (coef, factors), the factors from
    LEFT to right; family "spin": factors ("Sz" | "S+" | "S-", site) on a sector with n_a_old down spins; family "fermion":
    factors ("n" | "c+" | "c", site, species) with species 0 = up, 1 = down on the (n_a_old, n_b_old) sector.  Returns the
    dimension of the target sector (every product must change the particle numbers by the same amount)."""
    fam = {"spin": 0, "fermion": 1}[family]
    names = SPIN_OPS if fam == 0 else FERMION_OPS
    terms = list(terms)
    ptr, kind, site, spec, coef = [0], [], [], [], []
    for c, factors in terms:
        coef.append(complex(c))
        for f in factors:
            kind.append(names[f[0]])
            site.append(int(f[1]))
            spec.append(int(f[2]) if fam == 1 else 0)
        ptr.append(len(kind))
    ptr, kind, site, spec = (np.asarray(a, dtype=np.int32) for a in (ptr, kind, site, spec))
    coef = np.asarray(coef, dtype=np.complex128)
    dim_new = C.c_int64(0)
    check(lib().qbh_mopr_terms_dev(fam, n_sites, n_a_old, n_b_old, len(terms), _p(ptr), _p(kind), _p(site), _p(spec), _p(coef), d_vec_old, d_vec_new,
                                   C.byref(dim_new), stream), "qbh_mopr_terms_dev")
    return dim_new.value


def moprXvec_sz_repr(n_sites, n_dn, perms, chars_new, coef, d_vec_old, d_vec_new):
    """moprXvec_repr (src/model.cc:1715-1846) for S^z_q between momentum sectors of qbh_gen_heisenberg_repr; returns the
    number of representatives."""
    p = np.ascontiguousarray(np.asarray(perms, dtype=np.int32))
    ch = np.ascontiguousarray(np.asarray(chars_new, dtype=np.complex128))
    c = np.ascontiguousarray(coef, dtype=np.complex128)
    assert p.shape == (len(ch), n_sites) and c.size == n_sites
    dim = C.c_int64(0)
    check(lib().qbh_mopr_sz_repr_dev(n_sites, n_dn, len(ch), _p(p), _p(ch), _p(c), d_vec_old, d_vec_new, C.byref(dim)),
          "qbh_mopr_sz_repr_dev")
    return dim.value


def moprXvec_diag_hubrepr(n_sites, n_up, n_dn, perms, chars_new, coef_up, coef_dn, d_vec_old, d_vec_new):
    """moprXvec_repr (src/model.cc:1715-1846) for sum_s (coef_up[s] n_{s,up} + coef_dn[s] n_{s,dn}) between momentum sectors
    of qbh_gen_hubbard_repr (density N_q, S^z_q); chars_new = characters of the target momentum.  Returns the number of
    representatives."""
    p = np.ascontiguousarray(np.asarray(perms, dtype=np.int32))
    ch = np.ascontiguousarray(np.asarray(chars_new, dtype=np.complex128))
    cu = np.ascontiguousarray(coef_up, dtype=np.complex128)
    cd = np.ascontiguousarray(coef_dn, dtype=np.complex128)
    assert p.shape == (len(ch), n_sites) and cu.size == n_sites and cd.size == n_sites
    dim = C.c_int64(0)
    check(lib().qbh_mopr_diag_hubrepr_dev(n_sites, n_up, n_dn, len(ch), _p(p), _p(ch), _p(cu), _p(cd), d_vec_old, d_vec_new,
                                          C.byref(dim)), "qbh_mopr_diag_hubrepr_dev")
    return dim.value


def moprXvec_c_hubrepr(n_sites, n_up_old, n_dn_old, species, kind, perms, chars_old, chars_new, coef, d_vec_old, d_vec_new):
    """moprXvec_repr for sum_s coef[s] c_{s,sigma} (kind -1) / c^dag_{s,sigma} (kind +1), species 0 up / 1 down, between
    momentum sectors of qbh_gen_hubbard_repr with one particle less / more; returns (dim_old, dim_new)."""
    p = np.ascontiguousarray(np.asarray(perms, dtype=np.int32))
    co = np.ascontiguousarray(np.asarray(chars_old, dtype=np.complex128))
    cn = np.ascontiguousarray(np.asarray(chars_new, dtype=np.complex128))
    c = np.ascontiguousarray(coef, dtype=np.complex128)
    assert p.shape == (len(co), n_sites) and len(cn) == len(co) and c.size == n_sites
    d0, d1 = C.c_int64(0), C.c_int64(0)
    check(lib().qbh_mopr_c_hubrepr_dev(n_sites, n_up_old, n_dn_old, int(species), int(kind), len(co), _p(p), _p(co), _p(cn), _p(c),
                                       d_vec_old, d_vec_new, C.byref(d0), C.byref(d1)), "qbh_mopr_c_hubrepr_dev")
    return d0.value, d1.value


def moprXvec_flip_repr(n_sites, n_dn_old, kind, perms, chars_old, chars_new, coef, d_vec_old, d_vec_new):
    """moprXvec_repr for S^-_q (kind -1) / S^+_q (kind +1) between momentum sectors; returns (dim_old, dim_new)."""
    p = np.ascontiguousarray(np.asarray(perms, dtype=np.int32))
    co = np.ascontiguousarray(np.asarray(chars_old, dtype=np.complex128))
    cn = np.ascontiguousarray(np.asarray(chars_new, dtype=np.complex128))
    c = np.ascontiguousarray(coef, dtype=np.complex128)
    assert p.shape == (len(co), n_sites) and len(cn) == len(co) and c.size == n_sites
    d0, d1 = C.c_int64(0), C.c_int64(0)
    check(lib().qbh_mopr_flip_repr_dev(n_sites, n_dn_old, kind, len(co), _p(p), _p(co), _p(cn), _p(c), d_vec_old, d_vec_new,
                                       C.byref(d0), C.byref(d1)), "qbh_mopr_flip_repr_dev")
    return d0.value, d1.value


def measure_repr_static_hubbard(n_sites, n_up, n_dn, perms, chars, d_psi, one_body=(), two_body=(), spin_exchange=(), opts=None, source=None):
    """model::measure_repr_static (src/model.cc:1860-1891) for the two-species fermion family in a momentum sector:
    <psi| O_t |psi> with O_t = (1/N) sum_R T(R) O T(-R), the translation average of
        O = sum (a_up c+_{i,up} c_{j,up} + a_dn c+_{i,dn} c_{j,dn})                        one_body  = [(i, j, a_up, a_dn), ...]
          + sum (v_uu n_{i,up} n_{j,up} + v_ud n_{i,up} n_{j,dn} + v_du n_{i,dn} n_{j,up} + v_dd n_{i,dn} n_{j,dn})
                                                                                         two_body  = [(i, j, v_uu, v_ud, v_du, v_dd), ...]
          + sum a (S+_i S-_j + S-_i S+_j)                                                  spin_exchange = [(i, j, a), ...]
    e.g. the density-density correlator <n_i n_j> (two_body = [(i, j, 1, 1, 1, 1)]), the double occupancy
    (two_body = [(i, i, 0, 1, 0, 0)]) or <S^z_i S^z_j> (two_body = [(i, j, .25, -.25, -.25, .25)]); with spin_exchange a/2 = 1/2
    on top of the last one <S_i . S_j>.  The reference transforms the operator with every translation plan and applies the
    averaged mopr through moprXvec_repr; here the averaged operator is assembled as a sector operator on the device
    (qbh_gen_hubbard_repr: same rows, same representatives as the Hamiltonian of the sector) and applied once.
    d_psi: device vector of the sector IN THE ORDER OF THE GENERATORS (ascending representatives) -- the order of a stored
    hubbard_repr operator's device vectors and of every host vector.  A device vector of a handle that keeps another order
    internally (info().basis_internal != 0: the matrix-free sector operator with qbh_opts.sector_orbit, an operator whose basis
    was named or detected) must be passed together with that handle as `source`: it is translated first (qbh_vec_from_internal).
    Without `source` a handle-ordered vector would give a silently wrong number, so name the handle whenever there is one."""
    conv = None
    if source is not None and source.info().basis_internal != 0:
        conv = source.vec(1)
        source.from_internal(conv.ptr, d_psi)
        source.sync()
        d_psi = conv.ptr
    perms = np.asarray(perms, dtype=np.int64)
    w = 1.0 / len(perms)
    terms, pairs, exch = [], [], []
    for g in perms:                                   # O -> T(R) O T(-R): every site index through the translation plan
        terms += [(int(g[i]), int(g[j]), au * w, ad * w) for (i, j, au, ad) in one_body]
        pairs += [(int(g[i]), int(g[j]), vuu * w, vud * w, vdu * w, vdd * w) for (i, j, vuu, vud, vdu, vdd) in two_body]
        exch += [(int(g[i]), int(g[j]), a * w) for (i, j, a) in spin_exchange]
    Ot = csr_mat.hubbard_repr(n_sites, n_up, n_dn, None, perms, chars, U=0.0, terms=terms, pairs=pairs or None,
                              exchange=exch or None, fake_pos=0.0, opts=opts)
    y = Ot.vec(1)
    Ot.spmv(d_psi, y.ptr, 1.0, 0.0, 0.0)
    val = Ot.dotc(d_psi, y.ptr)
    y.free()
    Ot.destroy()
    if conv is not None:
        conv.free()
    return val


def measure_full_static_spin_dev(mat, n_sites, n_dn, d_phi, ops):
    """model<T>::measure_full_static (src/model.cc:1660-1694) for a product of spin operators on a fixed-N_dn sector, on the
    device: <phi| O_1 O_2 ... O_k |phi> with O_j = (kind_j, coef_j) as in moprXvec_spin, applied right to left (the
    rightmost operator acts first, src/basis.cc:2767); the product must return to the sector of phi.  mat: any operator
    handle of that sector (used for the reduction); d_phi: device address of the state."""
    import math
    cur_n, src, bufs = n_dn, d_phi, []
    try:
        if mat.info().basis_internal != 0:
            # the handle keeps its device vectors in another order than the generators' (a cut sector, a named / detected basis):
            # the mopr helpers work in the generators' order, so phi is translated first (qbh_vec_from_internal)
            conv = mat.vec(1)
            bufs.append(conv)
            mat.from_internal(conv.ptr, d_phi)
            mat.sync()
            d_phi = src = conv.ptr
        for kind, coef in reversed(list(ops)):
            new_n = cur_n - kind
            dst = DeviceVec(mat, math.comb(n_sites, new_n))
            bufs.append(dst)
            moprXvec_spin(n_sites, cur_n, kind, coef, src, dst.ptr)
            cur_n, src = new_n, dst.ptr
        if cur_n != n_dn:
            raise ValueError("the operator product does not return to the sector of phi")
        return mat.dotc(d_phi, src)
    finally:
        for b in bufs:
            b.free()


def measure_full_dynamic_dev(mat_new, apply_mopr, maxit):
    """model<T>::measure_full_dynamic (src/model.cc:1696-1712) end to end in HBM: apply_mopr(d_vec_new) writes
    A_q |phi> into the first slot of a two-slot device vector of the target sector (moprXvec_spin / moprXvec_onebody on
    the resident ground state), then norm, normalisation and the "dnmcs" Lanczos run on mat_new.
    Returns (m, norm, hessenberg)."""
    dim = mat_new.dim
    v = mat_new.vec(2)
    try:
        if mat_new.info().basis_internal != 0:          # apply_mopr writes in the generators' order: translated into the handle's
            tmp = mat_new.vec(1)
            try:
                apply_mopr(tmp.ptr)
                mat_new.sync()
                mat_new.to_internal(v.at(0), tmp.ptr)
                mat_new.sync()
            finally:
                tmp.free()
        else:
            apply_mopr(v.at(0))
        norm = mat_new.nrm2(v.at(0))                               # sqrt(<phi| Aq^+ Aq |phi>)
        hessenberg = np.zeros(2 * maxit)
        if abs(norm) < lanczos_precision:
            return 0, norm, hessenberg
        mat_new.scal(1.0 / norm, v.at(0))
        m = lanczos(0, maxit - 1, maxit, dim, mat_new, None, hessenberg, "dnmcs", device_v=v)
        return m, norm, hessenberg
    finally:
        v.free()


def write_lanczos_log(rows, filename):
    """Append the rows returned by lanczos() in the format of log_Lanczos_srval (src/lanczos.cc:102-128)."""
    head1 = "".join("%20s" % s for s in ("#(1)", "(2)", "(3)", "(4)", "(5)", "(6)", "(7)", "(8)", "(9)", "(10)"))
    head2 = "".join("%20s" % s for s in ("Iter(k)", "Ritz[0]", "Ritz[1]", "Ritz[2]", "Ritz[3]", "a[k-1]", "b[k]",
                                         "accuracy", "accu_E0", "accu_E1"))
    with open(filename, "a") as f:
        for r in rows:
            vals = [r["k"]] + list(r["ritz"]) + [r["a"], r["b"], r["accuracy"], r["accu_E0"], r["accu_E1"]]
            f.write(head1 + "\n" + head2 + "\n")
            f.write("%20d" % vals[0] + "".join("%20.10g" % x for x in vals[1:]) + "\n")


class E0Result:
    """The fields model<T> exposes after locate_E0_* (src/qbasis.h: eigenvals_full, eigenvecs_full, E0, E1, gap, nconv)."""

    def __init__(self):
        self.eigenvals = []
        self.eigenvecs = None
        self.E0 = self.E1 = self.gap = None
        self.nconv = 0
        self.steps = {}


def locate_E0_lanczos(mat, nev=1, ncv=1, maxit=1000):
    """Work-alike of model<T>::locate_E0_lanczos (src/model.cc:1123-1316) for the CSR branch:
    E0 (Lanczos) -> V0 (CG) -> E1 (Lanczos, re-orthogonalised against phi0) -> V1 (CG),
    with all vectors resident in HBM for the whole call."""
    assert 0 < nev <= 2 and nev - 1 <= ncv <= nev
    dim = mat.dim
    seed = 1
    res = E0Result()
    nv = 5 if (ncv == 2) else (4 if ncv > 0 else 2)
    v = mat.vec(nv)
    hess = np.zeros(2 * maxit)
    try:
        mat.randomize(v.at(0), seed)                                     # :1165
        m = lanczos(0, maxit - 1, maxit, dim, mat, None, hess, "sr_val0", device_v=v)   # :1177
        ritz, s = hess_eigen(hess, maxit, m, "sr")
        res.eigenvals = [ritz[0]]
        res.E0 = ritz[0]
        res.steps["E0"] = m
        res.steps["E0_accuracy"] = abs(hess[m] * s[m - 1])
        res.hessenberg_E0 = hess.copy()
        if ncv == 0:
            return res
        mat.randomize(v.at(2 * dim), seed)                               # :1209
        mcg, accu = eigenvec_CG(dim, maxit, 0, mat, res.E0, v.at(2 * dim), v.at(0), v.at(dim), v.at(3 * dim),
                                device=True)
        if not accu < lanczos_precision:
            raise RuntimeError("CG did not converge (accuracy %g)" % accu)   # assert at :1221
        res.steps["V0"] = mcg
        res.nconv = 1
        if nev == 2:
            mat.randomize(v.at(0), seed)                                 # :1237-1241
            alpha = mat.dotc(v.at(2 * dim), v.at(0))
            nrm = np.sqrt(mat.axpy_norm(-alpha, v.at(2 * dim), v.at(0)))
            mat.scal(1.0 / nrm, v.at(0))
            m1 = lanczos(0, maxit - 1, maxit, dim, mat, None, hess, "sr_val1", device_v=v)
            ritz, s = hess_eigen(hess, maxit, m1, "sr")
            res.eigenvals.append(ritz[0])
            res.E1 = ritz[0]
            res.gap = res.E1 - res.E0
            res.steps["E1"] = m1
        if ncv == 1:
            res.eigenvecs = v.download(2 * dim, dim)                     # :1267-1273
            return res
        mat.randomize(v.at(3 * dim), seed + 7)                           # :1280
        mcg1, accu1 = eigenvec_CG(dim, maxit, 0, mat, res.E1, v.at(3 * dim), v.at(0), v.at(dim), v.at(4 * dim),
                                  device=True)
        res.steps["V1"] = mcg1
        res.nconv = 2
        if res.gap < lanczos_precision:                                  # :1301-1310
            alpha = mat.dotc(v.at(2 * dim), v.at(3 * dim))
            nrm = np.sqrt(mat.axpy_norm(-alpha, v.at(2 * dim), v.at(3 * dim)))
            mat.scal(1.0 / nrm, v.at(3 * dim))
        res.eigenvecs = v.download(2 * dim, 2 * dim)
        return res
    finally:
        v.free()


def locate_E0_iram(mat, nev=2, ncv=6, maxit=0, method="auto", order="sr"):
    """Work-alike of model<T>::locate_E0_iram (src/model.cc:1319-1366); order="lr" gives
    locate_Emax_iram (src/model.cc:1369-1422)."""
    assert nev > 0 and ncv > nev + 1
    if maxit <= 0:
        maxit = nev * 100
    res = E0Result()
    v0 = None                       # ARPACK info = 0: the reference's v0 is never read (src/lanczos.cc:470)
    nconv, w, z = iram(mat.dim, mat, v0, nev, ncv, maxit, order, method=method)
    res.nconv = nconv
    res.eigenvals = list(w)
    res.eigenvecs = z
    res.E0 = w[0]
    if nconv > 1:
        res.gap = w[1] - w[0]
    return res
