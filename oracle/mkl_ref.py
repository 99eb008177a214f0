"""The reference's actual third-party SpMV -- Intel MKL's mkl_sparse_z_mv -- called directly through
ctypes when an MKL runtime is present in the image (/opt/conda/lib/libmkl_rt.so*).  TEST
INFRASTRUCTURE ONLY: used to cross-check the oracle's restatement of that call (tests/) and as an
extra reported CPU baseline (bench.py).  Nothing here builds or wraps the reference's sources; the
calls mirror src/sparse.cc:23-40 (mkl_sparse_z_create_csr, 4-array form, zero based, ILP64) and
src/sparse.cc:262-289 (mkl_sparse_z_mv with descr = {HERMITIAN|GENERAL, UPPER, NON_UNIT}, alpha = beta = 1).
"""
import ctypes as C
import glob
import os

import numpy as np

SPARSE_OPERATION_NON_TRANSPOSE = 10
SPARSE_MATRIX_TYPE_GENERAL = 20
SPARSE_MATRIX_TYPE_HERMITIAN = 22
SPARSE_FILL_MODE_UPPER = 41
SPARSE_DIAG_NON_UNIT = 50
SPARSE_INDEX_BASE_ZERO = 0


class _Descr(C.Structure):
    _fields_ = [("type", C.c_int), ("mode", C.c_int), ("diag", C.c_int)]


class _Z(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


_mkl = None


def load(threads=None, threading_layer="gnu"):
    """dlopen libmkl_rt with the ILP64 interface (the reference builds with -DMKL_ILP64). Returns None if absent."""
    global _mkl
    if _mkl is not None:
        return _mkl
    cands = sorted(glob.glob("/opt/conda/lib/libmkl_rt.so*")) + ["libmkl_rt.so.2", "libmkl_rt.so.1", "libmkl_rt.so"]
    lib = None
    for c in cands:
        try:
            lib = C.CDLL(c, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    if lib is None:
        return None
    lib.MKL_Set_Interface_Layer(C.c_int(1))                       # MKL_INTERFACE_ILP64
    lib.MKL_Set_Threading_Layer(C.c_int({"seq": 1, "gnu": 3, "tbb": 4, "intel": 0}[threading_layer]))
    if threads:
        lib.MKL_Set_Num_Threads(C.c_int(int(threads)))
    lib.mkl_sparse_z_create_csr.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mkl_sparse_z_mv.argtypes = [C.c_int, _Z, C.c_void_p, _Descr, C.c_void_p, _Z, C.c_void_p]
    lib.mkl_sparse_destroy.argtypes = [C.c_void_p]
    lib.MKL_Get_Max_Threads.restype = C.c_int
    _mkl = lib
    return lib


class MklCsr:
    """csr_mat<complex<double>> as the reference hands it to MKL (src/sparse.cc:258)."""

    def __init__(self, dim, ia, ja, val, sym, ncols=None):
        lib = load()
        if lib is None:
            raise RuntimeError("no MKL runtime in this image")
        self.lib = lib
        self.dim = int(dim)
        self.ncols = int(ncols) if ncols is not None else int(dim)
        self.sym = bool(sym)
        self.ia = np.ascontiguousarray(ia, dtype=np.int64)
        self.ja = np.ascontiguousarray(ja, dtype=np.int64)
        self.val = np.ascontiguousarray(val, dtype=np.complex128)
        self.handle = C.c_void_p()
        st = lib.mkl_sparse_z_create_csr(C.byref(self.handle), SPARSE_INDEX_BASE_ZERO, self.dim, self.ncols,
                                         self.ia.ctypes.data, self.ia.ctypes.data + 8, self.ja.ctypes.data,
                                         self.val.ctypes.data)
        if st != 0:
            raise RuntimeError("create_handle failed (%d)" % st)
        self.descr = _Descr(SPARSE_MATRIX_TYPE_HERMITIAN if self.sym else SPARSE_MATRIX_TYPE_GENERAL,
                            SPARSE_FILL_MODE_UPPER, SPARSE_DIAG_NON_UNIT)

    def multmv2(self, x, y):
        """y += H x, exactly the call of src/sparse.cc:287."""
        st = self.lib.mkl_sparse_z_mv(SPARSE_OPERATION_NON_TRANSPOSE, _Z(1.0, 0.0), self.handle, self.descr,
                                      x.ctypes.data, _Z(1.0, 0.0), y.ctypes.data)
        if st != 0:
            raise RuntimeError("matrix-vector product failed. (%d)" % st)
        return y

    def multmv(self, x):
        y = np.zeros(self.dim, dtype=np.complex128)
        return self.multmv2(np.ascontiguousarray(x, dtype=np.complex128), y)

    def threads(self):
        return int(self.lib.MKL_Get_Max_Threads())

    def __del__(self):
        try:
            if self.handle:
                self.lib.mkl_sparse_destroy(self.handle)
        except Exception:
            pass
