#!/usr/bin/env python3
"""IRAM path, two ways, on the same device-built operator:
 (a) qbh_iram: restarted Lanczos with the Krylov basis resident in HBM;
 (b) ARPACK (scipy's bundled znaupd/zneupd) by reverse communication over the host-vector seam
     qbh_multmv -- the literal structure of call_arpack (src/lanczos.cc:438-495): x and y cross PCIe
     on every matvec and ARPACK's own basis updates run on one host core."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="hubbard_4x3_half")
    ap.add_argument("--nev", type=int, default=2)
    ap.add_argument("--ncv", type=int, default=8)
    ap.add_argument("--skip-arpack", action="store_true")
    args = ap.parse_args()
    import quantum_basis_amd as q
    W = bench.workloads()[args.workload]
    dim = bench.dim_of(W)
    A = bench.build_operator(W, None, q.make_opts(value_dict=1, profile=1))
    maxit = 100 * args.nev
    out = {"workload": args.workload, "dim": dim, "nnz": A.nnz, "nev": args.nev, "ncv": args.ncv}
    t0 = time.perf_counter()
    nconv, w, _ = q.iram(dim, A, None, args.nev, args.ncv, maxit, "sr", method="device")
    out["device"] = {"s": round(time.perf_counter() - t0, 4), "nconv": nconv, "eigenvals": [float(x) for x in w], **q.iram.last}
    if not args.skip_arpack:
        t0 = time.perf_counter()
        nconv, wa, _ = q.iram_arpack(dim, A, None, args.nev, args.ncv, maxit, "sr")
        out["arpack_seam"] = {"s": round(time.perf_counter() - t0, 4), "nconv": nconv, "eigenvals": [float(x) for x in wa], **q.iram.last}
        out["max_abs_diff"] = float(np.max(np.abs(np.array(w) - np.array(wa))))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
