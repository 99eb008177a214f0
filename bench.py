#!/usr/bin/env python3
"""bench.py -- Lanczos iterations/s + CSR-SpMV GB/s (vs the HBM roofline) on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1
the driver launches it under torch.distributed.run, one rank per GPU.  W untimed warm-up
Lanczos steps, then EXACTLY K timed steps bracketed by barrier + synchronize, max over ranks,
rank 0 prints ONE JSON line.

A "step" is one full iteration of the reference's Lanczos loop (src/lanczos.cc:193-264):
v_m = -b v_{m-2} + H v_{m-1}, a = Re<v_{m-1}, v_m>, v_m -= a v_{m-1}, b = |v_m|, v_m /= b, and
the host Ritz solve + stop test ("sr_val0").  The Hamiltonian is assembled on the device
(synthetic: no dataset exists for this path), resident in HBM before the timed region.

Scaling is STRONG: the same operator is row-sharded over the N GPUs.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def workloads():
    from quantum_basis_amd import lattices
    return {
        # BASELINE.json configs[2] / SURVEY 8(d) C3: the >=1e8-dim Hubbard the metric is quoted on
        "hubbard_4x4_half": dict(kind="hubbard", n_sites=16, n_up=8, n_dn=8, bonds=lattices.square(4, 4), t=1.0, U=1.1),
        # SURVEY 8(d) C4 substitute (4x5, N_up = N_dn = 5; half filling is 3.4e10-dim)
        "hubbard_4x5_n5": dict(kind="hubbard", n_sites=20, n_up=5, n_dn=5, bonds=lattices.square(4, 5), t=1.0, U=1.1),
        # beyond what a stored CSR can hold on one GPU (nnz ~ 6e10): matrix-free only (--matrix-free)
        "hubbard_4x5_n6": dict(kind="hubbard", n_sites=20, n_up=6, n_dn=6, bonds=lattices.square(4, 5), t=1.0, U=1.1),
        "hubbard_4x3_half": dict(kind="hubbard", n_sites=12, n_up=6, n_dn=6, bonds=lattices.square(4, 3), t=1.0, U=1.1),
        "hubbard_4x2_half": dict(kind="hubbard", n_sites=8, n_up=4, n_dn=4, bonds=lattices.square(4, 2), t=1.0, U=1.1),
        # BASELINE.json configs[1] / C2
        "kagome_30": dict(kind="heisenberg", n_sites=30, n_dn=15, bonds=lattices.kagome(5, 2), J=1.0),
        # BASELINE.json configs[1]: the 36-site kagome torus (4 x 3 cells).  Sz = 0 has dim 9,075,135,300: no CSR can be
        # stored; it runs matrix-free with real-packed vectors (tools/big_lanczos.py kagome36).  n_dn = 9 is the same lattice at dim 9.4e7.
        "kagome_36": dict(kind="heisenberg", n_sites=36, n_dn=18, bonds=lattices.kagome(4, 3), J=1.0, packed_real=True),
        "kagome_36a": dict(kind="heisenberg", n_sites=36, n_dn=18, bonds=lattices.kagome_torus((4, 2), (2, 4)), J=1.0, packed_real=True),
        "triangular_36": dict(kind="heisenberg", n_sites=36, n_dn=18, bonds=lattices.triangular(6, 6), J=1.0, packed_real=True),
        "hubbard_4x5_n7": dict(kind="hubbard", n_sites=20, n_up=7, n_dn=7, bonds=lattices.square(4, 5), t=1.0, U=1.1, packed_real=True),
        "hubbard_4x5_n8": dict(kind="hubbard", n_sites=20, n_up=8, n_dn=8, bonds=lattices.square(4, 5), t=1.0, U=1.1, packed_real=True),
        "kagome_36_n9": dict(kind="heisenberg", n_sites=36, n_dn=9, bonds=lattices.kagome(4, 3), J=1.0),
        "kagome_24": dict(kind="heisenberg", n_sites=24, n_dn=12, bonds=lattices.kagome(4, 2), J=1.0),
        "chain_22": dict(kind="heisenberg", n_sites=22, n_dn=11, bonds=lattices.chain(22), J=1.0),
        # BASELINE.json configs[4] family: triangular 6x6, translation-symmetric sector k = (1,0), complex phases.
        # Sz = 0 (n_dn = 18, dim ~2.5e8, 285 GB of complex128 CSR) is the 8-GPU case; these fit one GPU.
        "triangular_6x6_k10_n12": dict(kind="heisenberg_repr", n_sites=36, n_dn=12, bonds=lattices.triangular(6, 6), J=1.0,
                                       trans=(6, 6), k=(1, 0)),
        "triangular_6x6_k10_sz0": dict(kind="heisenberg_repr", n_sites=36, n_dn=18, bonds=lattices.triangular(6, 6), J=1.0,
                                       trans=(6, 6), k=(1, 0)),
        "triangular_6x6_k10_n15": dict(kind="heisenberg_repr", n_sites=36, n_dn=15, bonds=lattices.triangular(6, 6), J=1.0,
                                       trans=(6, 6), k=(1, 0)),
        "triangular_4x4_k01": dict(kind="heisenberg_repr", n_sites=16, n_dn=8, bonds=lattices.triangular(4, 4), J=1.0,
                                   trans=(4, 4), k=(0, 1)),
    }


def traffic_of(key):
    """HBM bytes per launch measured in separate rocprofv3 --pmc passes (profiles/traffic.json), or None"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[key]["hbm_bytes"]
    except Exception:
        return None


def dim_of(w):
    from math import comb
    if w["kind"] == "hubbard":
        return comb(w["n_sites"], w["n_up"]) * comb(w["n_sites"], w["n_dn"])
    if w["kind"] == "heisenberg_repr":
        return None                      # known only after the representatives have been enumerated
    return comb(w["n_sites"], w["n_dn"])


def build_operator(w, rows, opts, matrix_free=False, shard=(0, 1)):
    import quantum_basis_amd as q
    if w["kind"] == "hubbard":
        return q.csr_mat.hubbard(w["n_sites"], w["n_up"], w["n_dn"], w["bonds"], t=w["t"], U=w["U"], rows=rows, opts=opts,
                                 matrix_free=matrix_free)
    if w["kind"] == "heisenberg_repr":
        from quantum_basis_amd import lattices
        perms, shifts = lattices.translations(*w["trans"])
        return q.csr_mat.heisenberg_repr(w["n_sites"], w["n_dn"], w["bonds"], perms, lattices.characters(shifts, w["k"], w["trans"]),
                                         J=w["J"], shard=shard, opts=opts)
    return q.csr_mat.heisenberg(w["n_sites"], w["n_dn"], w["bonds"], J=w["J"], rows=rows, opts=opts, matrix_free=matrix_free)


def cpu_baseline(A, dim, budget_rows):
    """Port baseline: the oracle's OpenMP CSR SpMV + BLAS-1 (oracle/qb_oracle.c) on the host
    cores, timed on a bounded slab of the SAME operator (first R rows, full-length x), scaled
    by dim/R to one Lanczos iteration."""
    from oracle import qb_oracle as qo
    R = int(min(A.dim, budget_rows))
    ia, ja, val = A.download(0, R)
    slab = qo.Csr.__new__(qo.Csr)
    # the slab is R x dim (rectangular): build the ctypes view by hand
    slab.dim, slab.ia, slab.ja, slab.val, slab.sym = R, ia, ja.astype(np.int64), val, False
    slab.nnz = int(ia[-1])
    slab._c = qo._CSR(R, slab.nnz, 0, slab.ia.ctypes.data, slab.ja.ctypes.data, slab.val.ctypes.data)
    x = qo.vec_randomize(dim, 1)
    y = np.zeros(R, dtype=np.complex128)
    L = qo.lib()
    import ctypes as C
    reps, t_spmv, t_blas = 0, 0.0, 0.0
    t_end = time.time() + 12.0
    while reps < 3 or (time.time() < t_end and reps < 50):
        t0 = time.perf_counter()
        L.qbo_multmv2(slab.ref(), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
        t1 = time.perf_counter()
        # the BLAS-1 of one Lanczos step on the same R elements: scale, dot, axpy, nrm2, scale
        xs = x[:R]
        y *= -0.5
        a = qo.dotc(xs, y).real
        y -= a * xs
        nrm = qo.nrm2(y)
        y *= 1.0 / max(nrm, 1e-300)
        t2 = time.perf_counter()
        if reps > 0:             # first pass warms the page cache / threads
            t_spmv += t1 - t0
            t_blas += t2 - t1
        reps += 1
    n = reps - 1
    scale = dim / R
    ms_spmv = 1e3 * t_spmv / n * scale
    ms_iter = 1e3 * (t_spmv + t_blas) / n * scale
    bytes_alg = slab.nnz * 20 + (R + 1) * 8 + R * 32
    mkl = None
    try:        # the reference's own SpMV library, when the image has it: mkl_sparse_z_mv on the same slab
        from oracle import mkl_ref
        if mkl_ref.load() is not None:
            M = mkl_ref.MklCsr(R, slab.ia, slab.ja, slab.val, False, ncols=dim)
            ym = np.zeros(R, dtype=np.complex128)
            M.multmv2(x, ym)
            t0 = time.perf_counter()
            nm = 0
            while nm < 3 or (time.perf_counter() - t0 < 4.0 and nm < 30):
                M.multmv2(x, ym)
                nm += 1
            tm = (time.perf_counter() - t0) / nm
            mkl = {"spmv_ms_scaled": round(1e3 * tm * scale, 3), "spmv_GBps": round(bytes_alg / tm / 1e9, 3),
                   "threads": M.threads(), "call": "mkl_sparse_z_mv, GENERAL descriptor, full storage, no mkl_sparse_optimize (src/sparse.cc:262-289)"}
    except Exception as e:
        mkl = {"error": repr(e)}
    return {"value": round(1e3 / ms_iter, 4), "unit": "lanczos_iters/s", "cores": qo.num_threads(), "kind": "port", "mkl": mkl,
            "sample": "first %d of %d rows (%d nnz) of the same operator, full-length x, %d timed passes of "
                      "oracle qbo_multmv2 (OpenMP, full storage) + the step's BLAS-1; scaled by dim/rows"
                      % (R, dim, slab.nnz, n),
            "spmv_ms_scaled": round(ms_spmv, 3), "spmv_GBps": round(bytes_alg / (t_spmv / n) / 1e9, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("QBH_WORKLOAD", "hubbard_4x4_half"))
    ap.add_argument("--kernel", type=int, default=0, help="0 auto (rows), 1 stream, 2 vector, 3 rows")
    ap.add_argument("--npb", type=int, default=0)
    ap.add_argument("--swizzle", type=int, default=2)
    ap.add_argument("--value-dict", type=int, default=1, help="1: dictionary-code the value stream when <=256 distinct values (lossless)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-converge", action="store_true", help="skip the untimed run to convergence (E0)")
    ap.add_argument("--cpu-rows", type=int, default=2_000_000)
    ap.add_argument("--matrix-free", action="store_true", help="use the matrix-free Hubbard operator as THE operator (not the CSR north-star path)")
    ap.add_argument("--no-matrix-free", action="store_true", help="skip the extra measurement of the matrix-free Hubbard operator")
    ap.add_argument("--packed-real", action="store_true", help="matrix-free operator + Lanczos vectors as packed doubles (qbh_lanczos_real_dev); "
                    "forced for the workloads whose complex vectors do not fit one GPU")
    ap.add_argument("--converge", action="store_true", help="run to convergence also for the dim > 1e9 packed-real workloads")
    ap.add_argument("--no-plain", action="store_true", help="skip the extra (untimed-region) measurement of the uncoded complex128 kernel")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import quantum_basis_amd as q
    from quantum_basis_amd import _lib, dist as qdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
        args.gpus = world
    _lib.require_gpu()                      # no CPU fallback: fail loudly
    # QBH_DIST_BACKEND=gloo lets several ranks share one GPU on a single-GPU test rig (RCCL refuses that)
    backend = os.environ.get("QBH_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    local_rank = dev_index
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    W = workloads()[args.workload]
    packed_real = bool(args.packed_real or W.get("packed_real"))
    if packed_real:
        if world > 1:
            raise SystemExit("--packed-real is a single-GPU mode")
        args.matrix_free = True
        args.no_plain = args.no_matrix_free = args.no_cpu_baseline = True
        if W.get("packed_real") and not args.converge:
            args.no_converge = True          # hundreds of ~1 s steps: tools/big_lanczos.py does that run (logs under profiles/)
    dim = dim_of(W)
    if dim is None:
        r0, r1 = 0, -1                   # the generator shards by (rank, world) itself
    else:
        nblk, ranges = qdist.row_partition(dim, world)
        r0, r1 = ranges[rank]
    stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(stream):
        opts = q.make_opts(device=local_rank, stream=stream.cuda_stream, spmv_kernel=args.kernel,
                           nnz_per_block=args.npb, xcd_swizzle=args.swizzle,
                           value_dict=args.value_dict, profile=1)
        t_gen = time.time()
        A = build_operator(W, (r0, r1), opts, matrix_free=args.matrix_free, shard=(rank, world))
        torch.cuda.synchronize()
        t_gen = time.time() - t_gen
        info = A.info()
        if dim is None:
            dim = int(info.ncols)
        if world > 1:
            comm = qdist.ShardComm(dim, rank=rank, world=world, device=device, stream=stream).attach(A)

        def allreduce_host(vals, op):
            t = torch.tensor(vals, dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            if world > 1:
                dist.all_reduce(t, op=op)
            return [float(z) for z in t.tolist()]

        nnz_total = int(allreduce_host([float(info.nnz)], dist.ReduceOp.SUM)[0])

        # lanczos() cannot make a single step from k = 0 (the reference's do-while runs once more after the bootstrap step,
        # src/lanczos.cc:167-193), so the recurrence is always started by at least two untimed steps: the timed region
        # then is EXACTLY K steps
        K, Wm = args.steps, max(args.warmup, 2)
        maxit = max(K + Wm + 16, 64)
        n = A.dim
        v = A.vec(1 if packed_real else 2)           # packed doubles: n complex128 = the two slots of n doubles
        hess = np.zeros(2 * maxit)

        def fresh_start(seed):
            if packed_real:
                import ctypes
                _lib.check(_lib.lib().qbh_vec_randomize_real(A.handle, v.ptr, ctypes.c_uint32(seed)), "qbh_vec_randomize_real")
            else:
                A.randomize(v.at(0), seed)
            hess[:] = 0.0
            return 0

        def lanczos_call(k0, nsteps, mx, hs):
            if packed_real:
                return q.lanczos_real(k0, nsteps, mx, A, v, hs)
            return q.lanczos(k0, nsteps, mx, n, A, None, hs, "sr_val0", device_v=v)

        def run_steps(k, nsteps, seed):
            """Advance nsteps Lanczos steps (restarting from a new start vector if the stop rule
            fires first); returns (k, seed, steps actually done)."""
            left, total = nsteps, 0
            while left > 0:
                m = lanczos_call(k, left, maxit, hess)
                left -= m - k
                total += m - k
                k = m
                if left > 0:          # converged / broke down before np steps: new Krylov space
                    seed += 1
                    k = fresh_start(seed)
            return k, seed, total

        seed = 1
        k = fresh_start(seed)
        if Wm > 0:
            k, seed, _ = run_steps(k, Wm, seed)
        A.stats(reset=True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k, seed, K_done = run_steps(k, K, seed)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        st = A.stats()
        elapsed, ms_spmv = allreduce_host([elapsed, st.ms_spmv / max(st.n_spmv, 1)], dist.ReduceOp.MAX)

        # untimed: run the same solver to convergence for E0 (parity across N and vs the small-size oracle tests)
        e0 = steps_e0 = None
        if not args.no_converge:
            maxit2 = 1000
            hess2 = np.zeros(2 * maxit2)
            fresh_start(1)
            m = lanczos_call(0, maxit2 - 1, maxit2, hess2)
            ritz, _ = q.hess_eigen(hess2, maxit2, m, "sr")
            e0, steps_e0 = float(ritz[0]), int(m)

    # algorithmic bytes of ONE SpMV launch on this rank: nnz*(16+4) + (rows+1)*8 + x once + y once
    bytes_launch = info.nnz * 20 + (info.nrows + 1) * 8 + (dim if world > 1 else info.nrows) * 16 + info.nrows * 16
    achieved = bytes_launch / (ms_spmv * 1e-3) / 1e9 if ms_spmv > 0 else 0.0
    # HBM traffic of one launch: measured in separate rocprofv3 --pmc passes (tools/profile_bench.sh), kept in
    # profiles/traffic.json and quoted only when it was taken on this exact workload / kernel / value coding
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        key = "%s|%s|%s" % (args.workload, {1: "stream", 2: "vector", 3: "rows", 4: "matrix_free"}[info.kernel], "dict" if info.value_dict else "plain")
        if st.n_spmv_real > 0:
            key += "|real"
        if world == 1 and key in tj:
            traffic = tj[key]["hbm_bytes"]
    except Exception:
        traffic = None
    out = {
        "metric": "lanczos_iters_per_sec", "value": round(K_done / elapsed, 4), "unit": "lanczos_iters/s",
        "n_gpus": world, "steps": K_done, "warmup": Wm, "ms_per_step": round(1e3 * elapsed / K_done, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "complex128 (f64)",
        "data": "synthetic", "config": {"workload": args.workload, "dim": dim, "nnz_full": nnz_total,
                                         "rows_per_gpu": info.nrows, "parallelism": "row-shard x%d" % world,
                                         "kernel": {1: "stream", 2: "vector", 3: "rows", 4: "matrix_free"}[info.kernel], "value_dict": info.value_dict,
                                         "real_gather": bool(st.n_spmv_real > 0),
                                         "build_s": round(t_gen, 3)},
        "roofline": {"bound": "hbm", "kernel": {1: "k_spmv_stream", 2: "k_spmv_vector", 3: "k_spmv_rows", 4: "k_mf_hubbard"}[info.kernel],
                     "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "bytes_per_launch": bytes_launch, "ms_per_launch": round(ms_spmv, 4), "launches": int(st.n_spmv),
                     "note": "achieved = ALGORITHMIC bytes (SURVEY 8d: nnz*20 + rows*40) / kernel time; the kernel moves fewer "
                             "bytes than that (traffic) because values are dictionary-coded (1 or 2 B instead of 16 B, lossless) and, for a "
                             "real operator and real vectors, x is gathered as 8-byte real parts (bit-identical result)"},
        "e0": e0, "lanczos_steps_to_converge": steps_e0,
    }
    if packed_real:
        out["dtype"] = "f64 (real operator, Lanczos vectors stored as packed doubles)"
        out["config"]["vectors"] = "2 x %.1f GB packed doubles (qbh_lanczos_real_dev)" % (dim * 8e-9)
    if args.matrix_free:
        out["config"]["kernel"] = "matrix_free"
        out["roofline"]["kernel"] = "k_mf_hubbard" if W["kind"] == "hubbard" else "k_mf_heis"
        out["roofline"]["note"] = ("MATRIX-FREE operator (qbh_mf_hubbard / qbh_mf_heisenberg, SURVEY 8f-1): no CSR is stored; achieved = bytes the CSR of the "
                                   "same operator would move per SpMV / kernel time -- not the north-star CSR measurement")
    if world == 1 and info.value_dict and not args.no_plain:
        # transparency: the same SpMV with the value stream left as complex128 (16 B/nnz), measured after the
        # timed region on a second copy of the operator (it needs the full 20 B/nnz in HBM)
        try:
            with torch.cuda.stream(stream):
                P = build_operator(W, (r0, r1), shard=(rank, world), opts=q.make_opts(device=local_rank, stream=stream.cuda_stream,
                                                             value_dict=0, xcd_swizzle=args.swizzle, profile=1))
                pv = P.vec(2)
                P.randomize(pv.at(0), 1)
                P.spmv(pv.at(0), pv.at(n), 1.0, 0.0, 0.0, want_red=True)
                P.stats(reset=True)
                for _ in range(5):
                    P.spmv(pv.at(0), pv.at(n), 1.0, -0.5, 0.0, want_red=True)
                ps = P.stats()
                pms = ps.ms_spmv / max(ps.n_spmv, 1)
                out["roofline_plain_values"] = {"kernel": "k_spmv_rows (complex128 values)", "ms_per_launch": round(pms, 4),
                                                "achieved": round(bytes_launch / pms / 1e6, 2), "unit": "GB/s",
                                                "frac": round(bytes_launch / pms / 1e6 / HBM_PEAK_GBPS, 4), "launches": int(ps.n_spmv),
                                                "traffic": traffic_of("%s|rows|plain" % args.workload)}
                pv.free()
                P.destroy()
        except Exception as e:
            out["roofline_plain_values"] = {"error": repr(e)}
    if world == 1 and W["kind"] in ("hubbard", "heisenberg") and not args.no_matrix_free and not args.matrix_free:
        # SURVEY 8f-1 (next row, NOT the north-star CSR path): the same operator applied from the hop tables without a
        # stored matrix, same solver code; measured after the timed region, same step definition
        try:
            with torch.cuda.stream(stream):
                M = build_operator(W, (0, -1), q.make_opts(device=local_rank, stream=stream.cuda_stream, profile=1),
                                   matrix_free=True)
                mv = M.vec(2)
                mh = np.zeros(2 * maxit)
                M.randomize(mv.at(0), 1)
                mk = q.lanczos(0, max(Wm, 2), maxit, n, M, None, mh, "sr_val0", device_v=mv)
                M.stats(reset=True)
                torch.cuda.synchronize()
                tm0 = time.perf_counter()
                mk2 = q.lanczos(mk, K, maxit, n, M, None, mh, "sr_val0", device_v=mv)
                torch.cuda.synchronize()
                tm = time.perf_counter() - tm0
                ms_ = M.stats()
                mms = ms_.ms_spmv / max(ms_.n_spmv, 1)
                out["matrix_free_" + W["kind"]] = {"lanczos_iters_per_s": round((mk2 - mk) / tm, 4), "steps": int(mk2 - mk),
                                              "spmv_ms_per_launch": round(mms, 4),
                                              "equivalent_csr_GBps": round(bytes_launch / mms / 1e6, 2),
                                              "table_bytes": int(M.info().bytes_matrix),
                                              "kernel": ("k_mf_hubbard_row" if ms_.n_spmv_real > 0 else "k_mf_hubbard") if W["kind"] == "hubbard" else "k_mf_heis",
                                              "traffic": traffic_of("%s|matrix_free|plain%s" % (args.workload, "|real" if ms_.n_spmv_real > 0 else "")),
                                              "note": "no stored matrix; not the CSR north-star path; traffic = HBM bytes per "
                                                      "apply from the rocprofv3 --pmc passes under profiles/"}
                mv.free()
                M.destroy()
        except Exception as e:
            out["matrix_free_" + W["kind"]] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(A, dim, args.cpu_rows)
        except Exception as e:      # the baseline is reported, never required
            out["cpu_baseline"] = {"value": None, "unit": "lanczos_iters/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
