#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-vector seam qbh_multmv / qbh_multmv2 (level-1 seam of INTEGRATION.md): what the unchanged
reference pays per csr_mat::MultMv call when only src/sparse.cc is patched."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402


def main():
    import quantum_basis_amd as q
    name = sys.argv[1] if len(sys.argv) > 1 else "kagome_30"
    W = bench.workloads()[name]
    A = bench.build_operator(W, None, q.make_opts(profile=1))
    n = A.dim
    x = np.full(n, 1.0 / np.sqrt(n), dtype=np.complex128)
    y = np.zeros(n, dtype=np.complex128)
    A.MultMv(x, y)
    A.stats(reset=True)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        A.MultMv(x, y)
    t = (time.perf_counter() - t0) / reps
    st = A.stats()
    print("%s dim %d: qbh_multmv %.1f ms per call (kernel %.1f ms) = %.1f GB/s of host vector traffic (2 x 16 x dim bytes)"
          % (name, n, 1e3 * t, st.ms_spmv / max(st.n_spmv, 1), 32.0 * n / t / 1e9))
    t0 = time.perf_counter()
    for _ in range(reps):
        A.MultMv2(x, y)
    t = (time.perf_counter() - t0) / reps
    print("%s dim %d: qbh_multmv2 %.1f ms per call = %.1f GB/s (3 x 16 x dim bytes)" % (name, n, 1e3 * t, 48.0 * n / t / 1e9))


if __name__ == "__main__":
    main()
