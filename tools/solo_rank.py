#!/usr/bin/env python3
"""One rank of a P-rank row-sharded run, ALONE on this GPU, with its peers modelled by the librccl stand-in's solo mode
(tests/stub_rccl/rccl_stub.cpp, QBH_STUB_SOLO = GB/s per link): the rank builds its shard (whole major indices, split in place,
2-byte columns in both parts), attaches the library's NATIVE communicator (qbh_comm_create_rccl: side stream, events, the gather
in parts) and runs the Lanczos loop.  Every receive is a device copy of as many bytes followed by a hold of bytes / link rate on
the communicator's side stream -- no device synchronisation anywhere -- so the step time shows what the near pass hides of the
gather and what the far pass waits for: the TIMING path of SURVEY 8(e), rehearsed before a multi-GPU node runs it.  The numbers
the peers "send" are the rank's own block, so E0 means nothing here (parity of the sharded path: tests/test_gpu_native_ranks.py).

usage: QBH_RCCL_LIB=tests/stub_rccl/librccl_stub.so QBH_STUB_SOLO=50 python tools/solo_rank.py [workload] P rank [key=value ...]
       steps=20 warmup=4 parts=0 realwire=1 pipeline=1 sparse=1 partition=1 reserve=0
       QBH_STUB_SOLO_KERNEL=W:rccl holds the side stream with a KERNEL of W workgroups of RCCL's own footprint instead of a host function"""
import ctypes as C
import json
import os
import sys
import time
from math import comb

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import quantum_basis_amd as q  # noqa: E402
from quantum_basis_amd import _lib, dist as qdist  # noqa: E402
from quantum_basis_amd._lib import check, lib  # noqa: E402


def main():
    name = sys.argv[1]
    P, rank = int(sys.argv[2]), int(sys.argv[3])
    kv = dict(a.split("=", 1) for a in sys.argv[4:])
    steps, warmup = int(kv.get("steps", 20)), int(kv.get("warmup", 4))
    parts, realwire, pipeline, sparse = int(kv.get("parts", 0)), int(kv.get("realwire", 1)), int(kv.get("pipeline", 1)), int(kv.get("sparse", 1))
    partition = int(kv.get("partition", 1))
    reserve = int(kv.get("reserve", 0))            # qbh_opts.comm_reserve (0: the library's default, -1: none)
    split = int(kv.get("split", 1))                # 0: plain row shards (locally-owned / remote columns), the form of operators without a product basis
    if not os.environ.get("QBH_STUB_SOLO") or not os.environ.get("QBH_RCCL_LIB"):
        raise SystemExit("needs QBH_RCCL_LIB=<librccl_stub.so> and QBH_STUB_SOLO=<GB/s per link>")
    W = bench.workloads()[name]
    dim = bench.dim_of(W)
    S = comb(W["n_sites"], W["n_dn"])
    cuts = qdist.kron_row_cuts(dim, S, P)
    r0, r1 = int(cuts[rank]), int(cuts[rank + 1])
    opts = q.make_opts(value_dict=0, real_fast_path=0, profile=1, gather_parts=parts, real_wire=realwire, lanczos_pipeline=pipeline, sparse_gather=sparse, major_partition=P if partition else 0, comm_reserve=reserve,
                       kron_split=1 if split else 0)
    t0 = time.time()
    A = bench.build_operator(W, (r0, r1), opts)
    info = A.info()
    uid = np.zeros(128, dtype=np.uint8)
    check(lib().qbh_rccl_unique_id(uid.ctypes.data_as(C.c_void_p)), "qbh_rccl_unique_id")
    c = np.asarray(cuts, dtype=np.int64)
    uniform = all(int(c[k + 1] - c[k]) == int(c[1] - c[0]) for k in range(P))
    check(lib().qbh_comm_create_rccl(A.handle, uid.ctypes.data_as(C.c_void_p), rank, P, None if uniform else c.ctypes.data_as(C.c_void_p)),
          "qbh_comm_create_rccl")
    build_s = time.time() - t0
    n = A.dim
    v = A.vec(2)
    A.randomize(v.at(0), 1)
    maxit = steps + warmup + 16
    hess = np.zeros(2 * maxit)
    k = q.lanczos(0, max(warmup, 2), maxit, n, A, None, hess, "dnmcs", device_v=v)
    A.stats(reset=True)
    A.sync()
    t1 = time.perf_counter()
    k2 = q.lanczos(k, steps, maxit, n, A, None, hess, "dnmcs", device_v=v)
    A.sync()
    el = time.perf_counter() - t1
    st = A.stats()
    inf = A.info()
    nst = k2 - k
    ms_step = 1e3 * el / max(nst, 1)
    ms_spmv = st.ms_spmv / max(st.n_spmv, 1)
    ms_gather = st.ms_gather / max(st.n_gather, 1)
    elem = int(inf.wire_element_bytes)
    block_bytes = elem * max(int(c[q_ + 1] - c[q_]) for q_ in range(P)) * (float(inf.gather_needed_frac) if inf.gather_sparse else 1.0)   # (the mean share; the model holds the stream for the LONGEST piece)
    rate = float(os.environ["QBH_STUB_SOLO"])
    b_alg = info.nnz * 20 + (info.nrows + 1) * 8 + dim * 16 + info.nrows * 16
    print(json.dumps({
        "tool": "tools/solo_rank.py", "workload": name, "ranks": P, "rank": rank, "rows": int(info.nrows), "nnz": int(info.nnz), "steps": int(nst),
        "columns": {0: "int32", 1: "near 2-byte, far int32", 2: "near int32, far 2-byte", 3: "2-byte in both parts"}[int(inf.kron_cols16)],
        "gather_parts": int(inf.gather_parts), "element_bytes": elem, "gather_needed_frac": round(float(inf.gather_needed_frac), 4), "personalised_exchange": bool(inf.gather_sparse), "major_partition": int(inf.major_partition), "lanczos_pipeline": pipeline, "comm_reserve": reserve, "split": split,
        "link_model": {"GBps_per_link": rate, "latency_us": float(os.environ.get("QBH_STUB_LATENCY_US", 20)),
                       "modelled_ms_per_gather (longest block / link rate + latency per part)": round(block_bytes / rate / 1e6 + 0.02 * max(int(inf.gather_parts), 1), 3)},
        "ms_per_step": round(ms_step, 4), "ms_spmv_kernels (near + far + place + combine, event-timed on the operator's stream; the far pass's wait for its pieces is inside)": round(ms_spmv, 4),
        "ms_gather (event-timed on the side stream)": round(ms_gather, 4),
        "shard_spmv_roofline_frac_on_step (algorithmic bytes of the shard / step time)": round(b_alg / ms_step / 1e6 / bench.HBM_PEAK_GBPS, 4),
        "build_s": round(build_s, 2)}))
    v.free()
    check(lib().qbh_comm_destroy(A.handle), "qbh_comm_destroy")
    A.destroy()


if __name__ == "__main__":
    main()
