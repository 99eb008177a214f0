#!/usr/bin/env python3
"""Does the idle gap of a host synchronisation cost the next kernels time?  C3, complex128, Kronecker split in place: SpMVs issued
back to back (one synchronisation at the end) against one synchronisation per SpMV, against a synchronisation + a host sleep."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import quantum_basis_amd as q  # noqa: E402

W = bench.workloads()[sys.argv[1] if len(sys.argv) > 1 else "hubbard_4x4_half"]
A = bench.build_operator(W, (0, bench.dim_of(W)), q.make_opts(value_dict=0, real_fast_path=0, profile=0))
n = A.dim
v = A.vec(2)
A.randomize(v.at(0), 1)
A.randomize(v.at(n), 2)


def loop(reps, sync_each, sleep_s=0.0, red=False):
    A.sync()
    t0 = time.perf_counter()
    slept = 0.0
    for _ in range(reps):
        A.spmv(v.at(0), v.at(n), 1.0, -0.3, 0.0, want_red=red)
        if sync_each:
            A.sync()
        if sleep_s:
            time.sleep(sleep_s)
            slept += sleep_s
    A.sync()
    return 1e3 * (time.perf_counter() - t0 - slept) / reps


for _ in range(3):
    loop(3, True)
for rnd in range(3):
    print("round %d: back to back %.3f ms | sync each %.3f ms | sync + 2 ms sleep %.3f ms | sync + 20 ms sleep %.3f ms | reductions (sync each) %.3f ms"
          % (rnd, loop(20, False), loop(20, True), loop(20, True, 0.002), loop(10, True, 0.02), loop(20, False, 0.0, True)), flush=True)
