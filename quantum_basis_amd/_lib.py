"""ctypes binding of libqbhip.so (the C ABI declared in include/qbhip.h).

The library is the product; this module only loads it and declares signatures.
There is no CPU fallback: if the shared object is missing, or no HIP device is
visible, every compute entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libqbhip.so")
if os.environ.get("QBHIP_LIBRARY"):      # e.g. a host-AddressSanitizer build of the same sources (tools/asan_build/)
    SO_PATH = os.path.abspath(os.environ["QBHIP_LIBRARY"])

QBH_OK = 0
KERNEL_AUTO, KERNEL_STREAM, KERNEL_VECTOR, KERNEL_ROWS, KERNEL_MATRIX_FREE, KERNEL_WAVE = 0, 1, 2, 3, 4, 5
BASIS_NONE, BASIS_REF_FERMION2, BASIS_SPIN_SECTOR = 0, 1, 2
BASIS_SECTOR_ORBIT = 3    # qbh_csr_info.basis_internal of a matrix-free sector operator held orbit by orbit (qbh_opts.sector_orbit)
BASIS_DETECT = -1         # qbh_csr_set_basis only: the library's own search (qbh_opts.basis_detect)


class QbhError(RuntimeError):
    def __init__(self, code, where, detail):
        self.code = code
        super().__init__("%s failed: %s (%d)%s" % (where, _strerror(code), code,
                                                   (": " + detail) if detail else ""))


class Z(C.Structure):
    """qbh_z: one complex128 passed by value."""
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


class Opts(C.Structure):
    _fields_ = [("device", C.c_int), ("stream", C.c_void_p), ("spmv_kernel", C.c_int),
                ("nnz_per_block", C.c_int), ("xcd_swizzle", C.c_int), ("value_dict", C.c_int),
                ("profile", C.c_int), ("check_hermitian", C.c_int), ("real_fast_path", C.c_int),
                ("kron_split", C.c_int), ("kron_minor", C.c_int64), ("deterministic", C.c_int), ("basis_kind", C.c_int),
                ("n_sites", C.c_int), ("n_up", C.c_int), ("n_dn", C.c_int), ("kron_cols16", C.c_int),
                ("kron_sliced", C.c_int), ("kron_band", C.c_int), ("kron_cross_in_near", C.c_int), ("kron_coded", C.c_int),
                ("kron_uniform", C.c_int), ("gather_parts", C.c_int), ("wave_walk", C.c_int), ("tile_fold", C.c_int),
                ("autotune", C.c_int), ("shard_split", C.c_int), ("real_forms", C.c_int), ("basis_detect", C.c_int),
                ("sector_orbit", C.c_int), ("lanczos_pipeline", C.c_int), ("real_wire", C.c_int), ("sparse_gather", C.c_int), ("major_partition", C.c_int), ("sector_cut", C.c_int), ("comm_reserve", C.c_int)]


class CsrInfo(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("row_offset", C.c_int64),
                ("nnz", C.c_int64), ("n_blocks", C.c_int64), ("bytes_matrix", C.c_int64),
                ("bytes_algorithmic", C.c_int64), ("kernel", C.c_int), ("value_dict", C.c_int),
                ("device", C.c_int), ("stream", C.c_void_p), ("create_ms", C.c_double),
                ("create_bytes_in", C.c_int64), ("kron_minor", C.c_int64), ("kron_far_nnz", C.c_int64), ("kron_band", C.c_int), ("kron_sliced", C.c_int),
                ("kron_inplace", C.c_int), ("tuned", C.c_int), ("tune_ms_rows", C.c_double), ("tune_ms_wave", C.c_double),
                ("basis_internal", C.c_int), ("kron_classes", C.c_int), ("kron_cross_nnz", C.c_int64), ("gather_parts", C.c_int),
                ("kron_cols16", C.c_int), ("basis_detected", C.c_int), ("basis_n_sites", C.c_int), ("basis_n_up", C.c_int),
                ("basis_n_dn", C.c_int), ("basis_detect_ms", C.c_double), ("kron_table_kernel", C.c_int), ("wire_element_bytes", C.c_int), ("major_partition", C.c_int), ("gather_sparse", C.c_int), ("gather_needed_frac", C.c_double)]


class LanczosRow(C.Structure):
    _fields_ = [("k", C.c_int64), ("ritz", C.c_double * 4), ("a_km1", C.c_double),
                ("b_k", C.c_double), ("accuracy", C.c_double), ("accu_E0", C.c_double),
                ("accu_E1", C.c_double)]


class SolverInfo(C.Structure):
    _fields_ = [("log", C.POINTER(LanczosRow)), ("log_cap", C.c_int64), ("log_len", C.c_int64),
                ("n_matvec", C.c_int64), ("n_reorth", C.c_int64), ("ms_total", C.c_double),
                ("ms_spmv", C.c_double), ("cg_resid", C.POINTER(C.c_double)),
                ("resume", C.c_int64), ("cnt_accuE0", C.c_int64), ("accuracy", C.c_double),
                ("theta0_prev", C.c_double), ("theta1_prev", C.c_double)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)       # (ctx, packed)
ALLWAIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)


PART_BEGIN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64))
PART_WAIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
PART_BEGIN_W_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_int)


class Comm(C.Structure):
    _fields_ = [("rank", C.c_int), ("nranks", C.c_int), ("nblk", C.c_int64),
                ("d_xsend", C.c_void_p), ("d_xfull", C.c_void_p), ("d_scal", C.c_void_p),
                ("d_xfull_r", C.c_void_p),
                ("ctx", C.c_void_p), ("allgather_x", ALLGATHER_FN), ("allreduce_sum", ALLREDUCE_FN),
                ("allgather_begin", ALLGATHER_FN), ("allgather_wait", ALLWAIT_FN), ("row_cuts", C.c_void_p),
                ("allgather_part_begin", PART_BEGIN_FN), ("allgather_part_wait", PART_WAIT_FN),      # optional: the gather in parts
                ("allgather_part_begin_w", PART_BEGIN_W_FN), ("exchange_v", C.c_void_p)]                               # (exchange_v: native communicator only)                                            # ... with the wire format named (ABI 600)


class Stats(C.Structure):
    _fields_ = [("n_spmv", C.c_int64), ("ms_spmv", C.c_double), ("ms_spmv_min", C.c_double),
                ("n_gather", C.c_int64), ("ms_gather", C.c_double), ("n_spmv_real", C.c_int64)]


# every symbol include/qbhip.h declares (tests check that the .so exports all of them)
EXPORTS = [
    "qbh_version", "qbh_device_count", "qbh_strerror", "qbh_last_error", "qbh_opts_default", "qbh_opts_set_default",
    "qbh_csr_create", "qbh_csr_create_rows", "qbh_balanced_row_cuts", "qbh_csr_create_device", "qbh_csr_destroy", "qbh_csr_get_info",
    "qbh_multmv", "qbh_multmv2",
    "qbh_vec_alloc", "qbh_vec_free", "qbh_vec_upload", "qbh_vec_download", "qbh_vec_zero", "qbh_vec_to_internal", "qbh_vec_from_internal",
    "qbh_vec_randomize",
    "qbh_spmv_dev", "qbh_dotc_dev", "qbh_axpy_norm_dev", "qbh_scal_dev", "qbh_nrm2_dev",
    "qbh_lanczos", "qbh_lanczos_dev", "qbh_lanczos_real_dev", "qbh_vec_randomize_real", "qbh_eigenvec_cg_real_dev", "qbh_eigenvec_cg", "qbh_eigenvec_cg_dev", "qbh_hess_eigen", "qbh_iram",
    "qbh_mopr_spin_dev", "qbh_mopr_onebody_dev", "qbh_mopr_terms_dev", "qbh_mopr_sz_repr_dev", "qbh_mopr_flip_repr_dev",
    "qbh_crc32", "qbh_vec_disk_write", "qbh_vec_disk_read", "qbh_ckpt_lanczos_update", "qbh_ckpt_lanczos_init", "qbh_lanczos_ckpt",
    "qbh_ckpt_cg_update", "qbh_ckpt_cg_init", "qbh_ckpt_cg_clean", "qbh_eigenvec_cg_ckpt",
    "qbh_csr_set_comm", "qbh_rccl_unique_id", "qbh_comm_create_rccl", "qbh_comm_destroy", "qbh_get_stats", "qbh_sync", "qbh_csr_set_option", "qbh_csr_major_order",
    "qbh_gen_hubbard", "qbh_mf_hubbard", "qbh_gen_heisenberg", "qbh_mf_heisenberg", "qbh_gen_heisenberg_repr", "qbh_gen_hubbard_repr", "qbh_gen_heisenberg_repr_cuts", "qbh_gen_hubbard_repr_cuts", "qbh_mf_hubbard_repr", "qbh_mopr_diag_hubrepr_dev", "qbh_mopr_c_hubrepr_dev", "qbh_csr_download", "qbh_csr_reference_order", "qbh_csr_set_basis",
]

_lib = None


def _strerror(code):
    try:
        return lib().qbh_strerror(code).decode()
    except Exception:
        return "error"


def lib():
    """Load libqbhip.so; raise loudly if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError("libqbhip.so not found at %s -- build it with "
                          "`python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % SO_PATH)
    # torch bundles its own libamdhip64.so.7; import it first so both share one HIP runtime
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
    L.qbh_strerror.restype = C.c_char_p
    L.qbh_last_error.restype = C.c_char_p
    L.qbh_csr_destroy.restype = None
    L.qbh_opts_default.restype = None
    L.qbh_opts_set_default.restype = None
    vp, i64, dbl = C.c_void_p, C.c_int64, C.c_double
    L.qbh_csr_create.argtypes = [C.POINTER(vp), i64, i64, C.c_int, vp, vp, vp, C.POINTER(Opts)]
    L.qbh_csr_create_rows.argtypes = [C.POINTER(vp), i64, i64, C.c_int, vp, vp, vp, i64, i64, C.POINTER(Opts)]
    L.qbh_balanced_row_cuts.argtypes = [i64, i64, C.c_int, vp, vp, C.c_int, vp]
    L.qbh_csr_create_device.argtypes = [C.POINTER(vp), i64, i64, i64, i64, vp, vp, vp, C.c_int,
                                        C.POINTER(Opts)]
    L.qbh_csr_destroy.argtypes = [vp]
    L.qbh_csr_get_info.argtypes = [vp, C.POINTER(CsrInfo)]
    L.qbh_multmv.argtypes = [vp, vp, vp]
    L.qbh_multmv2.argtypes = [vp, vp, vp]
    L.qbh_vec_alloc.argtypes = [C.POINTER(vp), i64]
    L.qbh_vec_free.argtypes = [vp]
    L.qbh_vec_upload.argtypes = [vp, vp, vp, i64]
    L.qbh_vec_download.argtypes = [vp, vp, vp, i64]
    L.qbh_vec_zero.argtypes = [vp, vp, i64]
    L.qbh_vec_to_internal.argtypes = [vp, vp, vp]
    L.qbh_vec_from_internal.argtypes = [vp, vp, vp]
    L.qbh_vec_randomize.argtypes = [vp, vp, C.c_uint32]
    L.qbh_spmv_dev.argtypes = [vp, vp, vp, dbl, dbl, dbl, vp]
    L.qbh_dotc_dev.argtypes = [vp, vp, vp, vp]
    L.qbh_axpy_norm_dev.argtypes = [vp, Z, vp, vp, vp]
    L.qbh_scal_dev.argtypes = [vp, dbl, vp]
    L.qbh_nrm2_dev.argtypes = [vp, vp, vp]
    L.qbh_lanczos.argtypes = [vp, i64, i64, i64, C.POINTER(i64), vp, vp, C.c_char_p,
                              C.POINTER(SolverInfo)]
    L.qbh_lanczos_dev.argtypes = L.qbh_lanczos.argtypes
    L.qbh_lanczos_real_dev.argtypes = L.qbh_lanczos.argtypes
    L.qbh_vec_randomize_real.argtypes = [vp, vp, C.c_uint32]
    L.qbh_eigenvec_cg.argtypes = [vp, i64, C.POINTER(i64), dbl, C.POINTER(dbl), vp, vp, vp, vp,
                                  C.POINTER(SolverInfo)]
    L.qbh_eigenvec_cg_dev.argtypes = L.qbh_eigenvec_cg.argtypes
    L.qbh_eigenvec_cg_real_dev.argtypes = L.qbh_eigenvec_cg.argtypes
    L.qbh_hess_eigen.argtypes = [vp, i64, i64, C.c_char_p, vp, vp]
    L.qbh_iram.argtypes = [vp, i64, i64, i64, C.c_char_p, dbl, C.c_uint32, C.POINTER(i64), vp, vp, C.POINTER(SolverInfo)]
    L.qbh_csr_set_comm.argtypes = [vp, C.POINTER(Comm)]
    L.qbh_mopr_spin_dev.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    L.qbh_mopr_sz_repr_dev.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.POINTER(i64)]
    L.qbh_mopr_flip_repr_dev.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.POINTER(i64), C.POINTER(i64)]
    L.qbh_mopr_onebody_dev.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.qbh_mopr_terms_dev.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.POINTER(i64), vp]
    L.qbh_crc32.argtypes = [C.c_uint32, vp, i64]
    L.qbh_crc32.restype = C.c_uint32
    L.qbh_vec_disk_write.argtypes = [C.c_char_p, i64, C.c_int, vp]
    L.qbh_vec_disk_read.argtypes = [C.c_char_p, i64, C.c_int, vp]
    L.qbh_ckpt_lanczos_update.argtypes = [C.c_char_p, i64, i64, i64, C.c_int, dbl, dbl, dbl, vp, vp, C.c_char_p]
    L.qbh_ckpt_lanczos_init.argtypes = [C.c_char_p, C.POINTER(i64), i64, i64, C.POINTER(C.c_int), C.POINTER(dbl), C.POINTER(dbl),
                                        C.POINTER(dbl), vp, vp, C.c_char_p]
    L.qbh_lanczos_ckpt.argtypes = [vp, i64, C.POINTER(i64), vp, vp, C.c_char_p, i64, i64, C.c_char_p, C.POINTER(C.c_int),
                                   C.POINTER(SolverInfo)]
    L.qbh_ckpt_cg_update.argtypes = [C.c_char_p, i64, i64, vp, vp, vp]
    L.qbh_ckpt_cg_init.argtypes = [C.c_char_p, C.POINTER(i64), i64, i64, vp, vp, vp]
    L.qbh_ckpt_cg_clean.argtypes = [C.c_char_p]
    L.qbh_eigenvec_cg_ckpt.argtypes = [vp, i64, C.POINTER(i64), dbl, C.POINTER(dbl), vp, vp, vp, vp, i64, i64, C.c_char_p, C.POINTER(C.c_int),
                                       C.POINTER(SolverInfo)]
    L.qbh_rccl_unique_id.argtypes = [vp]
    L.qbh_comm_create_rccl.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.qbh_comm_destroy.argtypes = [vp]
    L.qbh_get_stats.argtypes = [vp, C.POINTER(Stats), C.c_int]
    L.qbh_sync.argtypes = [vp]
    L.qbh_csr_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.qbh_csr_major_order.argtypes = [vp, vp, i64]
    L.qbh_gen_hubbard.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, vp, dbl, dbl,
                                  i64, i64, C.POINTER(Opts)]
    L.qbh_mf_hubbard.argtypes = L.qbh_gen_hubbard.argtypes
    L.qbh_gen_heisenberg.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, vp, dbl, i64, i64,
                                     C.POINTER(Opts)]
    L.qbh_mf_heisenberg.argtypes = L.qbh_gen_heisenberg.argtypes
    L.qbh_gen_heisenberg_repr.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, vp, dbl, C.c_int, vp, vp, dbl,
                                          C.c_int, C.c_int, C.POINTER(i64), C.POINTER(Opts)]
    L.qbh_gen_hubbard_repr.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, dbl, C.c_int, vp, vp,
                                       C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, dbl, C.c_int, C.c_int, C.POINTER(i64),
                                       C.POINTER(Opts)]
    L.qbh_gen_heisenberg_repr_cuts.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, vp, dbl, C.c_int, vp, vp, dbl,
                                               C.c_int, C.c_int, vp, C.POINTER(i64), C.POINTER(Opts)]
    L.qbh_gen_hubbard_repr_cuts.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, dbl, C.c_int, vp, vp,
                                            C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, dbl, C.c_int, C.c_int, vp, C.POINTER(i64),
                                            C.POINTER(Opts)]
    L.qbh_mf_hubbard_repr.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, dbl, C.c_int, vp, vp,
                                      C.c_int, vp, vp, dbl, C.POINTER(i64), C.POINTER(Opts)]
    L.qbh_mopr_diag_hubrepr_dev.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.POINTER(i64)]
    L.qbh_mopr_c_hubrepr_dev.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp,
                                         C.POINTER(i64), C.POINTER(i64)]
    L.qbh_csr_download.argtypes = [vp, i64, i64, vp, vp, vp]
    L.qbh_csr_reference_order.argtypes = [C.POINTER(vp), vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    _lib = L
    return L


def check(rc, where):
    if rc != QBH_OK:
        raise QbhError(rc, where, lib().qbh_last_error().decode())


def require_gpu():
    """Fail loudly when the HIP path cannot run (no silent CPU route exists)."""
    n = lib().qbh_device_count()
    if n <= 0:
        raise QbhError(-2, "qbh_device_count", "no MI355X / HIP device visible")
    return n
