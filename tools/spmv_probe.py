#!/usr/bin/env python3
"""Kernel-only timing probe: build a benchmark operator on the device and time qbh_spmv_dev
launches with the library's HIP events.  Used for A/B experiments (env QBH_DEBUG etc.)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="hubbard_4x4_half")
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--npb", type=int, default=0)
    ap.add_argument("--no-swizzle", action="store_true")
    ap.add_argument("--swizzle", type=int, default=2)
    ap.add_argument("--value-dict", type=int, default=0)
    ap.add_argument("--beta", type=float, default=-0.5)
    ap.add_argument("--red", type=int, default=1)
    args = ap.parse_args()
    import quantum_basis_amd as q
    W = bench.workloads()[args.workload]
    dim = bench.dim_of(W)
    opts = q.make_opts(spmv_kernel=args.kernel, nnz_per_block=args.npb, xcd_swizzle=0 if args.no_swizzle else args.swizzle,
                       value_dict=args.value_dict, profile=1)
    A = bench.build_operator(W, None, opts)
    info = A.info()
    v = A.vec(2)
    A.randomize(v.at(0), 1)
    A.randomize(v.at(dim), 2)
    A.spmv(v.at(0), v.at(dim), 1.0, args.beta, 0.0, want_red=bool(args.red))
    A.stats(reset=True)
    for _ in range(args.reps):
        A.spmv(v.at(0), v.at(dim), 1.0, args.beta, 0.0, want_red=bool(args.red))
    A.nrm2(v.at(0))      # drains the stream so the last event pair is harvested
    st = A.stats()
    ms = st.ms_spmv / max(st.n_spmv, 1)
    alg = info.bytes_algorithmic
    print(json.dumps({"workload": args.workload, "dim": dim, "nnz": info.nnz, "ms": round(ms, 4), "ms_min": round(st.ms_spmv_min, 4),
                      "alg_GBps": round(alg / ms / 1e6, 1), "frac": round(alg / ms / 1e6 / 8000.0, 4),
                      "debug": os.environ.get("QBH_DEBUG", "0"), "tpr": os.environ.get("QBH_TPR", "auto"), "npb": args.npb, "kernel": info.kernel,
                      "dict": info.value_dict}))


if __name__ == "__main__":
    main()
