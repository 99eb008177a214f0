#!/usr/bin/env python3
"""The host-vector seam at C3 (what the default patch's ARPACK loop pays per matvec): qbh_multmv (y = H x: x up, y down) and
qbh_multmv2 (y += H x: x and y up, y down) on pageable host arrays, wall time per call.  usage: python tools/pcie_seam.py [workload]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import quantum_basis_amd as q  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "hubbard_4x4_half"
W = bench.workloads()[name]
A = bench.build_operator(W, (0, -1), q.make_opts(value_dict=0, real_fast_path=0))
n = A.dim
rng = np.random.default_rng(1)
x = rng.normal(size=n).astype(np.complex128)
y = np.zeros(n, dtype=np.complex128)
out = {"tool": "tools/pcie_seam.py", "workload": name, "dim": int(n), "vector_GB": round(16 * n / 1e9, 3)}
for label, fn in (("qbh_multmv (x up, y down)", A.MultMv), ("qbh_multmv2 (x and y up, y down)", A.MultMv2)):
    fn(x, y)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        fn(x, y)
        ts.append(time.perf_counter() - t0)
    out[label] = {"ms_per_call_min": round(1e3 * min(ts), 2), "ms_per_call_median": round(1e3 * sorted(ts)[len(ts) // 2], 2)}
print(json.dumps(out))
A.destroy()
