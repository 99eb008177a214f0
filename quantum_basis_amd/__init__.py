"""quantum_basis_amd -- MI355X-native engine for the sparse Hamiltonian x vector hot path of
wztzjhn/quantum_basis (CSR SpMV inside Lanczos / CG / ARPACK-IRAM).

The product is libqbhip.so (HIP, gfx950) behind the C ABI of include/qbhip.h; this package is
the host-side mirror of the reference's operator/solver interface plus the ctypes loader.
"""
from . import _lib  # noqa: F401
from .engine import (csr_mat, DeviceVec, lanczos, lanczos_real, eigenvec_CG, eigenvec_CG_real, hess_eigen, iram, iram_arpack, vec_randomize,  # noqa: F401
                     locate_E0_lanczos, locate_E0_iram, measure_full_dynamic, write_lanczos_log, make_opts, lanczos_precision, sparse_precision,
                     balanced_row_cuts, moprXvec_spin, moprXvec_onebody, moprXvec_terms, moprXvec_sz_repr, moprXvec_flip_repr, moprXvec_diag_hubrepr, moprXvec_c_hubrepr, measure_full_dynamic_dev,
                     measure_full_static_spin_dev, measure_repr_static_hubbard, energy_scale)
