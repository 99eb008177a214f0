"""CPU checks of the C-ABI library: it loads, exports every symbol include/qbhip.h declares,
its host-only pieces (qbh_hess_eigen, argument checking) behave, and the compute entry points
fail loudly instead of falling back when there is no GPU.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import quantum_basis_amd as q
from quantum_basis_amd import _lib
from oracle import qb_oracle as qo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "qbhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qbh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == declared
    assert L.qbh_version() == 601


def test_struct_layouts_match_header():
    """sizeof / offsetof as the C compiler sees include/qbhip.h against the ctypes mirrors in _lib.py"""
    import subprocess
    import tempfile
    src = r"""
#include <stdio.h>
#include <stddef.h>
#include "qbhip.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(qbh_z), sizeof(qbh_lanczos_row), sizeof(qbh_opts), sizeof(qbh_stats),
           sizeof(qbh_csr_info), sizeof(qbh_solver_info), offsetof(qbh_opts, kron_split), offsetof(qbh_opts, kron_minor),
           offsetof(qbh_csr_info, kron_minor), offsetof(qbh_csr_info, kron_band), offsetof(qbh_csr_info, kron_sliced));
    printf("%zu %zu %zu %zu %zu\n", offsetof(qbh_opts, kron_cols16), offsetof(qbh_opts, gather_parts), offsetof(qbh_opts, basis_detect),
           offsetof(qbh_csr_info, kron_cols16), offsetof(qbh_csr_info, kron_table_kernel));
    printf("%zu\n", offsetof(qbh_opts, sector_orbit));
    return 0;
}
"""
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(tmp, "t.c"), "-o", os.path.join(tmp, "t")])
        got = [int(t) for t in subprocess.check_output([os.path.join(tmp, "t")], text=True).split()]
    want = [C.sizeof(_lib.Z), C.sizeof(_lib.LanczosRow), C.sizeof(_lib.Opts), C.sizeof(_lib.Stats), C.sizeof(_lib.CsrInfo),
            C.sizeof(_lib.SolverInfo), _lib.Opts.kron_split.offset, _lib.Opts.kron_minor.offset, _lib.CsrInfo.kron_minor.offset,
            _lib.CsrInfo.kron_band.offset, _lib.CsrInfo.kron_sliced.offset,
            _lib.Opts.kron_cols16.offset, _lib.Opts.gather_parts.offset, _lib.Opts.basis_detect.offset, _lib.CsrInfo.kron_cols16.offset,
            _lib.CsrInfo.kron_table_kernel.offset, _lib.Opts.sector_orbit.offset]
    assert got == want


def test_process_wide_default_options_for_a_host_whose_constructor_carries_none():
    """qbh_opts_set_default: what a NULL `opts` means for the reference's csr_mat(lil_mat&): the host
    names its basis once, qbh_opts_default hands it back, NULL restores the built-in defaults."""
    L = _lib.lib()
    o = _lib.Opts()
    L.qbh_opts_default(C.byref(o))
    assert (o.basis_kind, o.deterministic, o.kron_split, o.value_dict, o.real_fast_path) == (0, 0, 1, 1, 1)
    # the form switches that were environment variables up to ABI 400: their documented defaults
    assert (o.kron_cols16, o.kron_sliced, o.kron_band, o.kron_cross_in_near, o.kron_coded, o.kron_uniform, o.gather_parts, o.wave_walk, o.tile_fold,
            o.autotune, o.shard_split, o.real_forms, o.basis_detect, o.sector_orbit) == (1, 1, 0, 1, -1, 7, 0, -1, 1, 1, 1, 7, 1, 1)
    o.basis_kind, o.n_sites, o.n_up, o.n_dn = _lib.BASIS_REF_FERMION2, 16, 8, 8
    o.device, o.stream = 3, 0x1234                       # a default names no device and no stream: not kept
    L.qbh_opts_set_default(C.byref(o))
    try:
        o2 = _lib.Opts()
        L.qbh_opts_default(C.byref(o2))
        assert (o2.basis_kind, o2.n_sites, o2.n_up, o2.n_dn, o2.kron_split) == (_lib.BASIS_REF_FERMION2, 16, 8, 8, 1)
        assert o2.device == -1 and not o2.stream
    finally:
        L.qbh_opts_set_default(None)
    L.qbh_opts_default(C.byref(o))
    assert (o.basis_kind, o.n_sites) == (0, 0)


def test_strerror_and_no_device_is_loud():
    L = _lib.lib()
    assert L.qbh_strerror(0) == b"success"
    assert b"no CPU fallback" in L.qbh_strerror(-2)
    if L.qbh_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.QbhError) as e:
        q.csr_mat(2, [0, 1, 2], [0, 1], [1.0, 1.0], sym=True)
    assert e.value.code == -2
    # the raw C entry point refuses as well (argument validation runs first, then the device check)
    h = C.c_void_p()
    ia = np.array([0, 1, 2], dtype=np.int64)
    ja = np.array([0, 1], dtype=np.int64)
    val = np.ones(2, dtype=np.complex128)
    rc = L.qbh_csr_create(C.byref(h), 2, 2, 1, ia.ctypes.data, ja.ctypes.data, val.ctypes.data, None)
    assert rc == -2 and not h.value
    p = C.c_void_p()
    assert L.qbh_vec_alloc(C.byref(p), 10) == -2


def test_create_rejects_bad_csr_before_touching_the_device():
    L = _lib.lib()
    h = C.c_void_p()
    ia = np.array([0, 1, 3], dtype=np.int64)
    ja = np.array([0, 0, 1], dtype=np.int64)       # row 1 holds column 0 < row: invalid for sym_upper
    val = np.ones(3, dtype=np.complex128)
    assert L.qbh_csr_create(C.byref(h), 2, 3, 1, ia.ctypes.data, ja.ctypes.data, val.ctypes.data, None) == -1
    assert b"bad column" in L.qbh_last_error()
    assert L.qbh_csr_create(C.byref(h), 2, 2, 1, ia.ctypes.data, ja.ctypes.data, val.ctypes.data, None) == -1
    assert L.qbh_csr_create(C.byref(h), 0, 0, 1, None, None, None, None) == -1
    # full storage that is not Hermitian: the reference exits with code 99 (src/sparse.cc:251)
    ia = np.array([0, 2, 4], dtype=np.int64)
    ja = np.array([0, 1, 0, 1], dtype=np.int64)
    val = np.array([1, 2 + 1j, 2 + 1j, 1], dtype=np.complex128)
    assert L.qbh_csr_create(C.byref(h), 2, 4, 0, ia.ctypes.data, ja.ctypes.data, val.ctypes.data, None) == -5


def test_hess_eigen_host_against_oracle_and_numpy():
    rng = np.random.default_rng(11)
    maxit = 64
    for m in (1, 2, 5, 33, 63):
        h = np.zeros(2 * maxit)
        h[maxit:maxit + m] = rng.normal(size=m)
        h[1:m] = rng.uniform(0.1, 3.0, size=max(m - 1, 0))
        for order in ("sr", "lr", "sm", "lm"):
            ritz, s = q.hess_eigen(h, maxit, m, order)
            ritz_o, s_o = qo.hess_eigen(h, maxit, m, order)
            assert np.allclose(ritz, ritz_o, atol=1e-13)
            S, So = s.reshape(m, m), s_o.reshape(m, m)
            # eigenvectors agree up to sign
            assert np.allclose(np.abs(np.sum(S * So, axis=1)), 1.0, atol=1e-10)
        T = np.diag(h[maxit:maxit + m]) + np.diag(h[1:m], 1) + np.diag(h[1:m], -1)
        assert np.allclose(q.hess_eigen(h, maxit, m, "sr")[0], np.linalg.eigvalsh(T), atol=1e-13)
    with pytest.raises(_lib.QbhError):
        q.hess_eigen(np.zeros(20), 10, 10, "sr")       # assert(m > 0 && m < maxit), src/lanczos.cc:358
    with pytest.raises(_lib.QbhError):
        q.hess_eigen(np.zeros(20), 10, 3, "xx")


def test_iram_argument_checks_mirror_reference():
    class Fake:
        dim = 100
    with pytest.raises(ValueError):
        q.iram(100, Fake(), None, 0, 6, 100)           # src/lanczos.cc:502
    with pytest.raises(ValueError):
        q.iram(100, Fake(), None, 2, 6, 10)            # src/lanczos.cc:504
    with pytest.raises(ValueError):
        q.iram(100, Fake(), None, 2, 6, 100, "zz")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "quantum_basis_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "qb_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f


def test_no_environment_variable_selects_a_form():
    """Switch hygiene (round 5): the library reads exactly three environment variables -- QBH_DEBUG (one key=value list of
    measurement / tracing knobs), QBH_HOST_THREADS, QBH_RCCL_LIB -- and tests never steer it through the environment of a form."""
    import glob
    import re
    names = set()
    for f in glob.glob(os.path.join(ROOT, "quantum_basis_amd", "csrc", "*.*")):
        names |= set(re.findall(r'getenv\("([A-Z_0-9]+)"\)', open(f).read()))
    assert names == {"QBH_DEBUG", "QBH_HOST_THREADS", "QBH_RCCL_LIB"}, names
    allowed = {"QBH_DEBUG", "QBH_RCCL_LIB", "QBH_DIST_BACKEND", "QBH_HOST_THREADS", "QBH_PY_HOOKS", "QBH_WORKLOAD", "QBH_OK"}
    for f in glob.glob(os.path.join(ROOT, "tests", "*.py")):
        if os.path.basename(f) == "test_abi.py":
            continue
        used = set(re.findall(r'(?:setenv|environ)[^\n]*?"(QBH_[A-Z_0-9]+)"', open(f).read()))
        assert used <= allowed, (f, used - allowed)
