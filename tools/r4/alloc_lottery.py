#!/usr/bin/env python3
"""Is the +-7 % spread of the two passes a draw per PROCESS or per ALLOCATION?  One process creates the
headline operator several times (destroy, create again: new hipMalloc calls, possibly other physical pages) and times the SpMV
of every instance; between instances the vectors stay where they are.  A per-allocation draw shows up as a spread between
instances of one process comparable to the spread between processes; a per-process draw as instances that agree."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import quantum_basis_amd as q  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "hubbard_4x4_half"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    W = bench.workloads()[name]
    dim = bench.dim_of(W)
    keep = []
    for r in range(rounds):
        A = bench.build_operator(W, (0, dim), q.make_opts(value_dict=0, real_fast_path=0, profile=1))
        v = A.vec(2)
        A.randomize(v.at(0), 1)
        A.randomize(v.at(dim), 2)
        for _ in range(3):
            A.spmv(v.at(0), v.at(dim), 1.0, -0.3, 0.0, want_red=True)
        A.stats(reset=True)
        reps = 12
        for _ in range(reps):
            A.spmv(v.at(0), v.at(dim), 1.0, -0.3, 0.0, want_red=True)
        A.sync()
        st = A.stats()
        print(json.dumps({"instance": r, "ms_spmv_mean": round(st.ms_spmv / max(1, st.n_spmv), 3), "ms_spmv_min": round(st.ms_spmv_min, 3),
                          "kron_minor": int(A.info().kron_minor)}), flush=True)
        v.free()
        if r % 2 == 0 and len(keep) < 1 and os.environ.get("LOTTERY_HOLD"):
            keep.append(A)           # hold one instance: the next one cannot land on the same pages
        else:
            A.destroy()
    for A in keep:
        A.destroy()


if __name__ == "__main__":
    main()
