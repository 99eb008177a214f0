#!/usr/bin/env python3
"""What the gather in parts costs when there is nobody to wait for: C3 on ONE GPU under a one-rank native RCCL communicator
(every piece of the gather is the rank's own block, a device copy), Lanczos steps with the far pass launched as 1 / 4 / 8 band
ranges.  The difference between the lines is the price of splitting the far pass (tails of the shorter launches); what the
parts buy -- the far pass of the first ranges running under the rest of the wire time -- needs a node."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import quantum_basis_amd as q  # noqa: E402
from quantum_basis_amd import dist as qdist  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "hubbard_4x4_half"
    W = bench.workloads()[name]
    dim = bench.dim_of(W)
    A = bench.build_operator(W, (0, dim), q.make_opts(value_dict=0, real_fast_path=0))
    info = A.info()
    steps = 30
    ref = None
    for parts in [1, 4, 8, 1, 4]:
        os.environ["QBH_GATHER_PARTS"] = str(parts)
        comm = qdist.NativeComm(dim, rank=0, world=1).attach(A)
        best = None
        for rep in range(3):
            v = A.vec(2)
            A.randomize(v.at(0), 1)
            hess = np.zeros(2 * 64)
            A.sync()
            t0 = time.time()
            m = q.lanczos(0, steps, 64, dim, A, None, hess, "dnmcs", device_v=v)
            A.sync()
            dt = (time.time() - t0) / m * 1e3
            best = dt if best is None else min(best, dt)
            v.free()
        ab = np.concatenate([hess[64:64 + steps], hess[1:1 + steps]])
        if ref is None:
            ref = ab
        rec = {"workload": name, "gather_parts": parts, "ms_per_step": round(best, 3), "kron_minor": int(info.kron_minor),
               "max_rel_dev_of_a_b_from_the_single_gather": float(np.max(np.abs(ab - ref) / np.maximum(np.abs(ref), 1e-300)))}
        print(json.dumps(rec), flush=True)
        comm.detach(A)
    A.destroy()


if __name__ == "__main__":
    main()
