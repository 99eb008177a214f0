#!/usr/bin/env python3
"""What does a split shard report after the NATIVE communicator is attached the way bench.py does it (torch rendezvous over gloo, the
library's exchange on QBH_RCCL_LIB)?  usage: torchrun --nproc-per-node N tools/r5/parts_probe.py [workload] [gather_parts]"""
import os
import sys
from math import comb

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

import bench
import quantum_basis_amd as q
from quantum_basis_amd import dist as qdist


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "hubbard_4x5_n4"
    parts = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    split = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    more = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[4:]}          # further qbh_opts fields, key=value
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo")
    W = bench.workloads()[name]
    dim = bench.dim_of(W)
    cuts = qdist.kron_row_cuts(dim, comb(W["n_sites"], W["n_dn"]), world)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        opts = q.make_opts(device=0, stream=stream.cuda_stream, value_dict=0, real_fast_path=0, profile=1, gather_parts=parts, kron_split=split, **more)
        A = bench.build_operator(W, (int(cuts[rank]), int(cuts[rank + 1])), opts)
        i0 = A.info()
        print("%s P=%d rank %d rows [%d, %d) before: kron_minor %d sliced %d cols16 %d nnz %d" % (name, world, rank, cuts[rank], cuts[rank + 1], i0.kron_minor, i0.kron_sliced, i0.kron_cols16, i0.nnz), flush=True)
        qdist.NativeComm(dim, rank=rank, world=world, cuts=cuts).attach(A)
        i1 = A.info()
        print("rank %d after : kron_minor %d gather_parts %d" % (rank, i1.kron_minor, i1.gather_parts), flush=True)
        maxit = int(os.environ.get("PROBE_MAXIT", "600"))
        r = q.locate_E0_lanczos(A, nev=1, ncv=0, maxit=maxit) if maxit >= 600 else None
        if r is None:                      # a few steps only: does the exchange path run at all?
            import numpy as np
            v = A.vec(2)
            A.randomize(v.at(0), 1)
            hess = np.zeros(2 * maxit)
            m = q.lanczos(0, maxit - 1, maxit, A.dim, A, None, hess, "sr_val0", device_v=v)
            print("rank %d ran %d steps, a0 %.12f b1 %.12f" % (rank, m, hess[maxit], hess[1]), flush=True)
            v.free()
            dist.barrier()
            dist.destroy_process_group()
            return
        print("rank %d E0 %.12f steps %d" % (rank, r.E0, r.steps["E0"]), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
