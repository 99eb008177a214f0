"""ctypes binding of oracle/libqb_oracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The C file restates the reference's CPU algorithm (see qb_oracle.h for the
file:line map).  This module only marshals numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libqb_oracle.so")

LANCZOS_PRECISION = 2e-12      # src/miscellaneous.cc:47
SPARSE_PRECISION = 1e-14       # src/miscellaneous.cc:46


def build(force=False):
    """Compile the oracle with gcc (make -C oracle)."""
    src = os.path.join(_HERE, "qb_oracle.c")
    if force or (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class _CSR(C.Structure):
    _fields_ = [("dim", C.c_int64), ("nnz", C.c_int64), ("sym", C.c_int),
                ("ia", C.c_void_p), ("ja", C.c_void_p), ("val", C.c_void_p)]


class _Log(C.Structure):
    _fields_ = [("k", C.c_int64), ("ritz", C.c_double * 4), ("a_km1", C.c_double),
                ("b_k", C.c_double), ("accuracy", C.c_double), ("accu_E0", C.c_double),
                ("accu_E1", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.qbo_nrm2.restype = C.c_double
        L.qbo_expand_upper.restype = C.c_int64
        L.qbo_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Csr:
    """Host CSR in the reference layout (src/qbasis.h:976-1021): int64 ia/ja, complex128 val."""

    def __init__(self, dim, ia, ja, val, sym):
        self.dim = int(dim)
        self.ia = np.ascontiguousarray(ia, dtype=np.int64)
        self.ja = np.ascontiguousarray(ja, dtype=np.int64)
        self.val = np.ascontiguousarray(val, dtype=np.complex128)
        self.sym = bool(sym)
        self.nnz = int(self.ia[-1])
        assert self.ia.shape == (self.dim + 1,)
        assert self.ja.shape == (self.nnz,) and self.val.shape == (self.nnz,)
        self._c = _CSR(self.dim, self.nnz, int(self.sym), _p(self.ia).value, _p(self.ja).value,
                       _p(self.val).value)

    def ref(self):
        return C.byref(self._c)

    def multmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128)
        y = np.empty(self.dim, dtype=np.complex128)
        lib().qbo_multmv(self.ref(), _p(x), _p(y))
        return y

    def multmv2(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.complex128)
        assert y.dtype == np.complex128 and y.flags.c_contiguous
        lib().qbo_multmv2(self.ref(), _p(x), _p(y))
        return y

    def expand_full(self):
        """Upper-triangle storage -> both triangles (columns ascending)."""
        if not self.sym:
            return self
        ia = np.empty(self.dim + 1, dtype=np.int64)
        nnz = lib().qbo_expand_upper(self.ref(), _p(ia), None, None)
        ja = np.empty(nnz, dtype=np.int64)
        val = np.empty(nnz, dtype=np.complex128)
        lib().qbo_expand_upper(self.ref(), _p(ia), _p(ja), _p(val))
        return Csr(self.dim, ia, ja, val, False)

    def to_dense(self):
        d = np.empty((self.dim, self.dim), dtype=np.complex128, order="F")
        lib().qbo_to_dense(self.ref(), _p(d))
        return d


def vec_randomize(n, seed):
    x = np.empty(n, dtype=np.complex128)
    lib().qbo_vec_randomize(C.c_int64(n), _p(x), C.c_uint32(seed))
    return x


def nrm2(x):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    return lib().qbo_nrm2(C.c_int64(x.size), _p(x))


def dotc(x, y):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    y = np.ascontiguousarray(y, dtype=np.complex128)
    r = np.empty(2)
    lib().qbo_dotc(C.c_int64(x.size), _p(x), _p(y), _p(r))
    return complex(r[0], r[1])


def hess_eigen(hessenberg, maxit, m, order="sr"):
    h = np.ascontiguousarray(hessenberg, dtype=np.float64)
    ritz = np.empty(m)
    s = np.empty(m * m)
    info = lib().qbo_hess_eigen(_p(h), C.c_int64(maxit), C.c_int64(m), order.encode(), _p(ritz), _p(s))
    if info:
        raise RuntimeError("tridiagonal QL did not converge")
    return ritz, s


def lanczos(k, np_steps, maxit, csr, v, hessenberg, purpose):
    """Mirror of lanczos<T,MAT> (src/lanczos.cc:134).  v, hessenberg updated in place.

    Returns (m, log_rows, n_reorth); log_rows is a list of dicts with the
    columns of log_Lanczos_<purpose>.txt."""
    assert v.dtype == np.complex128 and v.flags.c_contiguous
    assert hessenberg.dtype == np.float64 and hessenberg.size >= 2 * maxit
    m = C.c_int64(0)
    nlog = C.c_int64(0)
    nre = C.c_int64(0)
    log = (_Log * int(maxit))()
    rc = lib().qbo_lanczos(C.c_int64(k), C.c_int64(np_steps), C.c_int64(maxit), C.byref(m),
                           C.c_int64(csr.dim), csr.ref(), _p(v), _p(hessenberg), purpose.encode(),
                           log, C.byref(nlog), C.byref(nre))
    if rc:
        raise RuntimeError("qbo_lanczos failed: %d" % rc)
    rows = [dict(k=r.k, ritz=list(r.ritz), a=r.a_km1, b=r.b_k, accuracy=r.accuracy,
                 accu_E0=r.accu_E0, accu_E1=r.accu_E1) for r in log[:nlog.value]]
    return m.value, rows, nre.value


def eigenvec_cg(maxit, csr, E0, v, r, p, pp, m0=0):
    """Mirror of eigenvec_CG (src/lanczos.cc:281).  Returns (m, accu, residual_log)."""
    for a in (v, r, p, pp):
        assert a.dtype == np.complex128 and a.flags.c_contiguous
    m = C.c_int64(m0)
    accu = C.c_double(0.0)
    rl = np.zeros(maxit + 1)
    rc = lib().qbo_eigenvec_cg(C.c_int64(csr.dim), C.c_int64(maxit), C.byref(m), csr.ref(),
                               C.c_double(E0), C.byref(accu), _p(v), _p(r), _p(p), _p(pp), _p(rl))
    if rc:
        raise RuntimeError("qbo_eigenvec_cg failed: %d" % rc)
    return m.value, accu.value, rl[1:m.value + 1]


def locate_E0_lanczos(csr, nev=1, ncv=1, maxit=1000):
    """Work-alike of model::locate_E0_lanczos (src/model.cc:1123-1316) on the oracle.

    Returns dict(E0, E1, gap, m_E0, m_V0, m_E1, m_V1, eigenvecs, hess0)."""
    assert 0 < nev <= 2 and nev - 1 <= ncv <= nev
    dim = csr.dim
    seed = 1
    out = {}
    v = np.zeros((4 if ncv > 0 else 2) * dim, dtype=np.complex128)
    v[:dim] = vec_randomize(dim, seed)
    hess = np.zeros(2 * maxit)
    m, rows, _ = lanczos(0, maxit - 1, maxit, csr, v, hess, "sr_val0")
    ritz, s = hess_eigen(hess, maxit, m, "sr")
    out.update(E0=ritz[0], m_E0=m, log_E0=rows, hess0=hess.copy(),
               accuracy_E0=abs(hess[m] * s[m - 1]))
    if ncv == 0:
        return out
    v[2 * dim:3 * dim] = vec_randomize(dim, seed)
    mcg, accu, rl = eigenvec_cg(maxit, csr, out["E0"], v[2 * dim:3 * dim], v[:dim], v[dim:2 * dim],
                                v[3 * dim:4 * dim])
    out.update(m_V0=mcg, accu_V0=accu, log_V0=rl)
    if nev == 2:
        v[:dim] = vec_randomize(dim, seed)
        phi0 = v[2 * dim:3 * dim]
        alpha = dotc(phi0, v[:dim])
        v[:dim] -= alpha * phi0
        v[:dim] /= nrm2(v[:dim])
        hess[:] = 0.0   # the reference reuses the array; entries are overwritten before use
        m1, rows1, nre = lanczos(0, maxit - 1, maxit, csr, v, hess, "sr_val1")
        ritz, s = hess_eigen(hess, maxit, m1, "sr")
        out.update(E1=ritz[0], gap=ritz[0] - out["E0"], m_E1=m1, n_reorth=nre)
    if ncv == 1:
        out["eigenvecs"] = v[2 * dim:3 * dim].copy()
        return out
    v = np.concatenate([v, np.zeros(dim, dtype=np.complex128)])
    v[3 * dim:4 * dim] = vec_randomize(dim, seed + 7)
    mcg1, accu1, _ = eigenvec_cg(maxit, csr, out["E1"], v[3 * dim:4 * dim], v[:dim], v[dim:2 * dim],
                                 v[4 * dim:5 * dim])
    ev = v[2 * dim:4 * dim].copy()
    if out["gap"] < LANCZOS_PRECISION:
        alpha = dotc(ev[:dim], ev[dim:])
        ev[dim:] -= alpha * ev[:dim]
        ev[dim:] /= nrm2(ev[dim:])
    out.update(m_V1=mcg1, accu_V1=accu1, eigenvecs=ev)
    return out


def first_touch(a):
    """Copy of `a` whose pages were first touched by the OpenMP threads (static slices): on a multi-socket host the
    array is then spread over all memory controllers instead of sitting on one NUMA node (CPU baseline only)."""
    a = np.ascontiguousarray(a)
    out = np.empty_like(a)
    lib().qbo_first_touch_copy(C.c_int64(a.nbytes), _p(a), _p(out))
    return out


def num_threads():
    return lib().qbo_num_threads()
