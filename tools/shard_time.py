#!/usr/bin/env python3
"""Per-rank SpMV time of row shards of a product-basis operator on ONE GPU (SURVEY 8e without the node): shard q of P -- whole
major indices (dist.kron_row_cuts), split in place like the unsharded operator -- driven with the full-length x and no
communicator.  What a rank of a P-GPU run executes per SpMV is the near pass on its own block + the far pass on the gathered
tiled x (+ the light combine pass); here the far pass reads a tiled copy of the full x made on this GPU (k_kron_tile over the
WHOLE vector, 5.3 GB at C3 -- a P-rank run tiles only its own block and receives the rest).  Up to round 4 a product switch
(QBH_KRON_REUSE_TILE) let this tool skip that copy; the switch is gone from the library (a caller's promise about x is not a
product feature), so the timing INCLUDES the copy and the record carries its cost estimate beside it: dim * 32 B at the
5.3 TB/s k_kron_tile8 was measured at (profiles/r4_lab/fold_per_kernel.txt: 1.0 ms at C3).  usage: python tools/shard_time.py [workload] [P ...]"""
import json
import os
import sys
from math import comb

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import quantum_basis_amd as q  # noqa: E402
from quantum_basis_amd import dist as qdist  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "hubbard_4x4_half"
    worlds = [int(a) for a in sys.argv[2:]] or [2, 4, 8]
    W = bench.workloads()[name]
    dim = bench.dim_of(W)
    S = comb(W["n_sites"], W["n_dn"])
    out = []
    for P in worlds:
        cuts = qdist.kron_row_cuts(dim, S, P)
        for rank in sorted({0, P // 2}):
            r0, r1 = int(cuts[rank]), int(cuts[rank + 1])
            A = bench.build_operator(W, (r0, r1), q.make_opts(value_dict=0, real_fast_path=0, profile=1))
            info = A.info()
            xv, yv = q.engine.DeviceVec(A, dim), A.vec()
            A.randomize(yv.at(0), 3)
            # a full-length random x: randomize works on shard-local vectors, so fill it block by block through y
            import numpy as np
            rng = np.random.default_rng(1)
            blk = 1 << 24
            for o in range(0, dim, blk):
                m = min(blk, dim - o)
                h = (rng.normal(size=m) + 1j * rng.normal(size=m)).astype(np.complex128)
                import ctypes as C
                q._lib.check(q._lib.lib().qbh_vec_upload(A.handle, xv.at(o), h.ctypes.data_as(C.c_void_p), C.c_int64(m)), "upload")
            for _ in range(3):
                A.spmv(xv.ptr, yv.ptr, 1.0, -0.3, 0.0, want_red=True)
            A.stats(reset=True)
            reps = 10
            for _ in range(reps):
                A.spmv(xv.ptr, yv.ptr, 1.0, -0.3, 0.0, want_red=True)
            A.sync()
            st = A.stats()
            ms = st.ms_spmv / max(1, st.n_spmv)
            b_alg = info.nnz * 20 + (info.nrows + 1) * 8 + dim * 16 + info.nrows * 16
            rec = {"workload": name, "P": P, "rank": rank, "rows": int(info.nrows), "nnz": int(info.nnz), "kron_minor": int(info.kron_minor),
                   "ms_spmv": round(ms, 3), "tile_of_full_x_in_the_timing": True, "tile_of_full_x_ms_estimate": round(dim * 32 / 5.3e9, 3),
                   "ms_spmv_minus_tile_estimate": round(ms - dim * 32 / 5.3e9, 3),
                   "algorithmic_bytes": int(b_alg), "frac": round(b_alg / (ms * 1e-3) / 8e12, 4),
                   "frac_minus_tile_estimate": round(b_alg / ((ms - dim * 32 / 5.3e9) * 1e-3) / 8e12, 4)}
            print(json.dumps(rec), flush=True)
            out.append(rec)
            xv.free()
            yv.free()
            A.destroy()
    return out


if __name__ == "__main__":
    main()
