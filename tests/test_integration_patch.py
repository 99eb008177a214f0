"""integration/qbhip_reference.patch applies cleanly to the unchanged reference (build container only: the GPU box has
no /root/reference, and nothing of the reference is committed)."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
PATCH = os.path.join(ROOT, "integration", "qbhip_reference.patch")


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference sources are only present in the build container")
def test_patch_applies_to_the_unchanged_reference():
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "src"))
        for f in ("sparse.cc", "lanczos.cc", "qbasis.h"):
            shutil.copy(os.path.join(REF, f), os.path.join(tmp, "src", f))
        before_h = open(os.path.join(tmp, "src", "qbasis.h")).read()
        dry = subprocess.run(["patch", "--dry-run", "-p1", "-i", PATCH], cwd=tmp, capture_output=True, text=True)
        assert dry.returncode == 0, dry.stdout + dry.stderr
        assert "FAILED" not in dry.stdout and "fuzz" not in dry.stdout
        real = subprocess.run(["patch", "-p1", "-i", PATCH], cwd=tmp, capture_output=True, text=True)
        assert real.returncode == 0, real.stdout + real.stderr
        sp = open(os.path.join(tmp, "src", "sparse.cc")).read()
        lz = open(os.path.join(tmp, "src", "lanczos.cc")).read()
        assert open(os.path.join(tmp, "src", "qbasis.h")).read() == before_h            # the public header is untouched
        # every binding of INTEGRATION.md is in place, and no MKL sparse call is left on the complex path
        for needle in ("qbh_csr_create(", "qbh_csr_destroy(", "qbh_multmv2(", "qbh_multmv("):
            assert needle in sp
        assert sp.count("create_handle(&handle, dim, nnz, sym, ia, ja, val)") == 2
        assert "mkl_sparse_z_mv" not in sp and "mkl_sparse_z_create_csr" not in sp
        for needle in ("qbh_lanczos(", "qbh_lanczos_ckpt(", "qbh_eigenvec_cg("):
            assert needle in lz
        # the explicit instantiations for model<T> as MAT are still there (not accelerated, must still link)
        assert "const model<std::complex<double>> &mat, std::complex<double> v[]," in lz


def test_patch_only_touches_the_two_translation_units():
    files = [ln.split()[1] for ln in open(PATCH) if ln.startswith("+++ ")]
    assert sorted(os.path.basename(f) for f in files) == ["lanczos.cc", "sparse.cc"]
    assert all("qbasis.h" not in ln for ln in open(PATCH) if ln.startswith(("+++", "---")))


# --------------------------------------------------------------------------------------------------------------------
# Compile acceptance (SURVEY 8b "Consequence for testing"): the patched translation units must COMPILE against the
# unchanged qbasis.h and import the C ABI instead of MKL's sparse calls.  The image has no mkl.h / arpack.hpp, so the
# test writes prototype-only stand-ins into a scratch directory (declarations, no code of the
# reference, nothing committed) -- enough for `g++ -c`, which is all that is asserted here.
_SHIM_MKL_H = r"""
#pragma once
#include <complex>
typedef long long MKL_INT;
typedef long long MKL_INT64;
extern "C" {
typedef enum { CblasRowMajor = 101, CblasColMajor = 102 } CBLAS_LAYOUT;
typedef enum { CblasNoTrans = 111, CblasTrans = 112, CblasConjTrans = 113 } CBLAS_TRANSPOSE;
void cblas_daxpy(MKL_INT, double, const double *, MKL_INT, double *, MKL_INT);
void cblas_zaxpy(MKL_INT, const void *, const void *, MKL_INT, void *, MKL_INT);
void cblas_dcopy(MKL_INT, const double *, MKL_INT, double *, MKL_INT);
void cblas_zcopy(MKL_INT, const void *, MKL_INT, void *, MKL_INT);
void cblas_dscal(MKL_INT, double, double *, MKL_INT);
void cblas_zscal(MKL_INT, const void *, void *, MKL_INT);
double cblas_dnrm2(MKL_INT, const double *, MKL_INT);
double cblas_dznrm2(MKL_INT, const void *, MKL_INT);
double cblas_ddot(MKL_INT, const double *, MKL_INT, const double *, MKL_INT);
void cblas_zdotc_sub(MKL_INT, const void *, MKL_INT, const void *, MKL_INT, void *);
void cblas_dgemm(CBLAS_LAYOUT, CBLAS_TRANSPOSE, CBLAS_TRANSPOSE, MKL_INT, MKL_INT, MKL_INT, double, const double *, MKL_INT,
                 const double *, MKL_INT, double, double *, MKL_INT);
void cblas_zgemm(CBLAS_LAYOUT, CBLAS_TRANSPOSE, CBLAS_TRANSPOSE, MKL_INT, MKL_INT, MKL_INT, const void *, const void *, MKL_INT,
                 const void *, MKL_INT, const void *, void *, MKL_INT);
#define LAPACK_ROW_MAJOR 101
#define LAPACK_COL_MAJOR 102
MKL_INT LAPACKE_dsyevd(int, char, char, MKL_INT, double *, MKL_INT, double *);
MKL_INT LAPACKE_zheevd(int, char, char, MKL_INT, std::complex<double> *, MKL_INT, double *);
MKL_INT LAPACKE_dstedc(int, char, MKL_INT, double *, double *, double *, MKL_INT);
MKL_INT LAPACKE_dgesv(int, MKL_INT, MKL_INT, double *, MKL_INT, MKL_INT *, double *, MKL_INT);
typedef enum { SPARSE_STATUS_SUCCESS = 0, SPARSE_STATUS_NOT_INITIALIZED = 1, SPARSE_STATUS_ALLOC_FAILED = 2,
               SPARSE_STATUS_INVALID_VALUE = 3, SPARSE_STATUS_EXECUTION_FAILED = 4 } sparse_status_t;
typedef enum { SPARSE_OPERATION_NON_TRANSPOSE = 10, SPARSE_OPERATION_TRANSPOSE = 11 } sparse_operation_t;
typedef enum { SPARSE_MATRIX_TYPE_GENERAL = 20, SPARSE_MATRIX_TYPE_SYMMETRIC = 21, SPARSE_MATRIX_TYPE_HERMITIAN = 22 } sparse_matrix_type_t;
typedef enum { SPARSE_INDEX_BASE_ZERO = 0, SPARSE_INDEX_BASE_ONE = 1 } sparse_index_base_t;
typedef enum { SPARSE_FILL_MODE_LOWER = 40, SPARSE_FILL_MODE_UPPER = 41 } sparse_fill_mode_t;
typedef enum { SPARSE_DIAG_NON_UNIT = 50, SPARSE_DIAG_UNIT = 51 } sparse_diag_type_t;
struct matrix_descr { sparse_matrix_type_t type; sparse_fill_mode_t mode; sparse_diag_type_t diag; };
struct sparse_matrix;
typedef struct sparse_matrix *sparse_matrix_t;
sparse_status_t mkl_sparse_d_create_csr(sparse_matrix_t *, sparse_index_base_t, MKL_INT, MKL_INT, MKL_INT *, MKL_INT *, MKL_INT *, double *);
sparse_status_t mkl_sparse_z_create_csr(sparse_matrix_t *, sparse_index_base_t, MKL_INT, MKL_INT, MKL_INT *, MKL_INT *, MKL_INT *,
                                        std::complex<double> *);
sparse_status_t mkl_sparse_d_mv(sparse_operation_t, double, const sparse_matrix_t, struct matrix_descr, const double *, double, double *);
sparse_status_t mkl_sparse_z_mv(sparse_operation_t, std::complex<double>, const sparse_matrix_t, struct matrix_descr,
                                const std::complex<double> *, std::complex<double>, std::complex<double> *);
sparse_status_t mkl_sparse_destroy(sparse_matrix_t);
void feastinit(MKL_INT *);
void zfeast_hcsrev(const char *, const MKL_INT *, const std::complex<double> *, const MKL_INT *, const MKL_INT *, MKL_INT *, double *,
                   MKL_INT *, const double *, const double *, MKL_INT *, double *, std::complex<double> *, MKL_INT *, double *, MKL_INT *);
}
"""

_SHIM_ARPACK_HPP = r"""
#pragma once
typedef long long a_int;
namespace arpack {
enum class which { largest_algebraic, smallest_algebraic, largest_magnitude, smallest_magnitude, largest_real, smallest_real,
                   largest_imaginary, smallest_imaginary, both_ends };
enum class bmat { identity, generalized };
enum class howmny { ritz_vectors, schur_vectors, ritz_specified };
template <typename... A> void saupd(A &&...);
template <typename... A> void seupd(A &&...);
template <typename... A> void naupd(A &&...);
template <typename... A> void neupd(A &&...);
}
"""


def _patched_tree(tmp):
    os.makedirs(os.path.join(tmp, "src"))
    for f in ("sparse.cc", "lanczos.cc", "qbasis.h"):
        shutil.copy(os.path.join(REF, f), os.path.join(tmp, "src", f))
    real = subprocess.run(["patch", "-p1", "-s", "-i", PATCH], cwd=tmp, capture_output=True, text=True)
    assert real.returncode == 0, real.stdout + real.stderr
    shim = os.path.join(tmp, "shim")
    os.makedirs(os.path.join(shim, "arpack-ng"))
    open(os.path.join(shim, "mkl.h"), "w").write(_SHIM_MKL_H)
    open(os.path.join(shim, "arpack-ng", "arpack.hpp"), "w").write(_SHIM_ARPACK_HPP)
    return shim


def _undefined(obj):
    out = subprocess.run(["nm", "-u", "-C", obj], capture_output=True, text=True, check=True).stdout
    return out


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference sources are only present in the build container")
@pytest.mark.parametrize("device_iram", [False, True])
def test_patched_translation_units_compile_and_import_the_c_abi(device_iram):
    with tempfile.TemporaryDirectory() as tmp:
        shim = _patched_tree(tmp)
        objs = {}
        for tu in ("sparse", "lanczos"):
            obj = os.path.join(tmp, tu + ".o")
            cmd = ["g++", "-std=c++17", "-O1", "-w", "-DMKL_ILP64", "-fopenmp", "-I", shim, "-I", os.path.join(ROOT, "include"),
                   "-c", os.path.join(tmp, "src", tu + ".cc"), "-o", obj]
            if device_iram:
                cmd.insert(1, "-DQBHIP_DEVICE_IRAM")
            p = subprocess.run(cmd, capture_output=True, text=True)
            assert p.returncode == 0, p.stderr[-4000:]
            objs[tu] = _undefined(obj)
        sp, lz = objs["sparse"], objs["lanczos"]
        # sparse.o: the complex operator lives on the GPU -- created, applied and destroyed through the C ABI ...
        for sym in ("qbh_csr_create", "qbh_csr_destroy", "qbh_multmv2", "qbh_multmv", "qbh_last_error"):
            assert sym in sp, sym
        # ... and no complex MKL sparse call is left (the double overloads stay: model<double> is not instantiated upstream)
        assert "mkl_sparse_z_mv" not in sp and "mkl_sparse_z_create_csr" not in sp
        # lanczos.o: the csr_mat<complex> solvers are the fused device solvers
        for sym in ("qbh_lanczos", "qbh_lanczos_ckpt", "qbh_eigenvec_cg"):
            assert sym in lz, sym
        assert ("qbh_iram" in lz) == device_iram
        # the matrix-free instantiations (MAT = model<T>) still come from the reference's own generic code
        defined = subprocess.run(["nm", "-C", "--defined-only", os.path.join(tmp, "lanczos.o")], capture_output=True, text=True).stdout
        assert "qbasis::model<std::complex<double> >" in defined and "void qbasis::lanczos<" in defined
        assert "void qbasis::iram<std::complex<double>, qbasis::csr_mat<std::complex<double> > >" in defined
