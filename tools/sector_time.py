#!/usr/bin/env python3
"""Per-apply time of the matrix-free momentum-sector operator (qbh_mf_hubbard_repr) in its two row orders: orbit by orbit of the
up patterns (qbh_opts.sector_orbit = 1, k_mf_sector_orb) and ascending (0, k_mf_sector with the rank tables).  Real sectors run
the packed-double Lanczos (what the C4 solve runs), complex ones the ordinary device driver.
usage: python tools/sector_time.py [workload ...]   (bench.py workloads of kind hubbard_repr_mf)   ORBIT=0,1  STEPS=12"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import quantum_basis_amd as q  # noqa: E402
from quantum_basis_amd import _lib, lattices  # noqa: E402


def main():
    names = sys.argv[1:] or ["hubbard_4x5_n6_k00_mf", "hubbard_4x5_n8_k20_mf", "hubbard_4x5_half_k00_mf"]
    forms = [int(t) for t in os.environ.get("ORBIT", "1,0").split(",")]
    steps = int(os.environ.get("STEPS", "12"))
    for name in names:
        W = bench.workloads()[name]
        for orb in forms:
            t0 = time.time()
            M = bench.build_operator(W, None, q.make_opts(profile=1, sector_orbit=orb))
            build_s = time.time() - t0
            info = M.info()
            dim = int(info.ncols)
            real = all(abs(complex(c).imag) < 1e-14 for c in np.asarray(lattices.characters(lattices.translations(*W["trans"])[1], W["k"], W["trans"])).ravel())
            hess = np.zeros(2 * (steps + 1))
            if real:
                v = M.vec(1)
                _lib.check(_lib.lib().qbh_vec_randomize_real(M.handle, v.ptr, C.c_uint32(1)), "qbh_vec_randomize_real")
                M.stats(reset=True)
                m = q.lanczos_real(0, steps, steps + 1, M, v, hess)
            else:
                v = M.vec(3)
                M.randomize(v.at(0), 1)
                M.stats(reset=True)
                m = q.lanczos(0, steps, steps + 1, dim, M, None, hess, "sr_val0", device_v=v)
            M.sync()
            st = M.stats()
            rec = {"workload": name, "sector_orbit": orb, "basis_internal": int(info.basis_internal), "dim": dim, "real_vectors": bool(real),
                   "build_s": round(build_s, 2), "steps": int(m), "n_spmv": int(st.n_spmv), "ms_per_apply": round(st.ms_spmv / max(1, st.n_spmv), 3),
                   "ms_min": round(st.ms_spmv_min, 3), "a0": float(hess[0]), "b1": float(hess[steps + 1 + 1]),
                   "table_bytes": int(info.bytes_matrix)}
            print(json.dumps(rec), flush=True)
            v.free()
            M.destroy()


if __name__ == "__main__":
    main()
