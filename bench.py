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

What the blocks of the JSON line are measured on:
  value / roofline   THE NORTH-STAR FORMAT (default --format complex128): CSR with complex128 values (16 B) + int32
                     columns, complex128 vectors, complex arithmetic -- the format SURVEY 8(d)'s algorithmic bytes
                     (nnz*20 + rows*40) describe, semantics of csr_mat::MultMv2 (src/sparse.cc:262-289).
                     At N = 1 the headline is the MEDIAN of --processes fresh child processes (each builds its own operator
                     and times W + exactly K steps); min / median / max are printed in "processes".
  fast_path          the library's default for a real operator (lossless, bit-identical results): 1-byte value codes,
                     vectors and gathers as packed doubles.  Its roofline fraction is defined on the bytes the form that
                     runs must MOVE (coded_format_roofline): the sliced coded split of a recognised Kronecker sum streams
                     almost no matrix, so bytes of a CSR it does not read are reported as *_equivalent_GBps, never as frac.
  locate_E0          the user call locate_E0_lanczos(nev=1, ncv=1) = Lanczos + CG eigenvector (src/model.cc:1123-1316,
                     src/lanczos.cc:281-341), timed end to end, with the CG step's own bytes and fraction.
  matrix_free_*      SURVEY 8f-1, no stored matrix (not the CSR path).
  cpu_baseline       the oracle port and the reference's own MKL SpMV on the host cores (reported, not a target).

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
KERNEL_NAME = {1: "k_spmv_stream", 2: "k_spmv_vector", 3: "k_spmv_rows", 4: "matrix_free", 5: "k_spmv_wave"}
KERNEL_KEY = {1: "stream", 2: "vector", 3: "rows", 4: "matrix_free", 5: "wave"}


def workloads():
    from quantum_basis_amd import lattices
    W = {
        # BASELINE.json configs[2] / SURVEY 8(d) C3: the >=1e8-dim Hubbard the metric is quoted on
        "hubbard_4x4_half": dict(kind="hubbard", n_sites=16, n_up=8, n_dn=8, bonds=lattices.square(4, 4), t=1.0, U=1.1),
        # SURVEY 8(d) C4 substitute (4x5, N_up = N_dn = 5; half filling is 3.4e10-dim)
        "hubbard_4x5_n5": dict(kind="hubbard", n_sites=20, n_up=5, n_dn=5, bonds=lattices.square(4, 5), t=1.0, U=1.1),
        # beyond what a stored CSR can hold on one GPU (nnz ~ 6e10): matrix-free only (--matrix-free)
        "hubbard_4x5_n6": dict(kind="hubbard", n_sites=20, n_up=6, n_dn=6, bonds=lattices.square(4, 5), t=1.0, U=1.1),
        "hubbard_4x5_n4": dict(kind="hubbard", n_sites=20, n_up=4, n_dn=4, bonds=lattices.square(4, 5), t=1.0, U=1.1),
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
        "chain_24": dict(kind="heisenberg", n_sites=24, n_dn=12, bonds=lattices.chain(24), J=1.0),
        "chain_26": dict(kind="heisenberg", n_sites=26, n_dn=13, bonds=lattices.chain(26), J=1.0),
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
        # The Hubbard family through momentum sectors (qbh_gen_hubbard_repr).  BASELINE configs[3] AS WRITTEN -- 4x5 at half
        # filling, 3.4e10 basis states -- is its k = (0,0) sector: 1,706,742,160 representatives, 7.4e10 nonzeros, ~100 GB per
        # GPU as columns + 1-byte value codes on FOUR GPUs (python bench.py --gpus 4 --workload hubbard_4x5_half_k00;
        # per-rank shards generated and timed on one GPU: profiles/r2_sectors/).
        "hubbard_4x5_half_k00": dict(kind="hubbard_repr", n_sites=20, n_up=10, n_dn=10, bonds=lattices.square(4, 5), t=1.0, U=1.1,
                                     trans=(4, 5), k=(0, 0), format="fast"),
        # 4x5 with 8+8 electrons: the ground-state sector k = (pi,0), 171 GB on one GPU (full basis: dim 1.59e10)
        "hubbard_4x5_n8_k20": dict(kind="hubbard_repr", n_sites=20, n_up=8, n_dn=8, bonds=lattices.square(4, 5), t=1.0, U=1.1,
                                   trans=(4, 5), k=(2, 0), format="fast"),
        "hubbard_4x5_n6_k00": dict(kind="hubbard_repr", n_sites=20, n_up=6, n_dn=6, bonds=lattices.square(4, 5), t=1.0, U=1.1,
                                   trans=(4, 5), k=(0, 0)),
        # ... and the same sector on ONE GPU in matrix-free form (qbh_mf_hubbard_repr: 3.5 GB instead of 364 GB)
        "hubbard_4x5_half_k00_mf": dict(kind="hubbard_repr_mf", n_sites=20, n_up=10, n_dn=10, bonds=lattices.square(4, 5), t=1.0, U=1.1,
                                        trans=(4, 5), k=(0, 0), packed_real=True),
        "hubbard_4x5_n8_k20_mf": dict(kind="hubbard_repr_mf", n_sites=20, n_up=8, n_dn=8, bonds=lattices.square(4, 5), t=1.0, U=1.1,
                                      trans=(4, 5), k=(2, 0), packed_real=True),
        "hubbard_4x5_n6_k00_mf": dict(kind="hubbard_repr_mf", n_sites=20, n_up=6, n_dn=6, bonds=lattices.square(4, 5), t=1.0, U=1.1,
                                      trans=(4, 5), k=(0, 0), packed_real=True),
        "hubbard_4x4_half_k00": dict(kind="hubbard_repr", n_sites=16, n_up=8, n_dn=8, bonds=lattices.square(4, 4), t=1.0, U=1.1,
                                     trans=(4, 4), k=(0, 0)),
    }
    # test rigs (tools/r6/fuzz_ranks_mid.py): further workloads as JSON in the environment, {"name": {"kind": "hubbard", "n_sites": ..., "bonds": [[a, b], ...], ...}}
    extra = os.environ.get("QBH_WORKLOAD_JSON")
    if extra:
        for name, spec in json.loads(extra).items():
            spec = dict(spec)
            spec["bonds"] = [tuple(b) for b in spec["bonds"]]
            W[name] = spec
    return W


def traffic_of(key):
    """(HBM bytes per launch, source file) measured in separate rocprofv3 --pmc passes and kept in profiles/traffic.json;
    (None, None) when that exact workload / kernel / value coding was never profiled.  NOT measured in this run."""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[key]
        TRAFFIC_STAMP[key] = e.get("kernel_sources_sha16")
        return e["hbm_bytes"], e.get("source")
    except Exception:
        return None, None


TRAFFIC_STAMP = {}


def traffic_stamp(key):
    """{'traffic_sources_sha16': hash of the kernel sources the PMC passes ran on, 'traffic_stale': True when the sources of this
    run differ (or the entry predates the stamps): the quoted traffic is then a number about OTHER code}"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import src_hash
        now = src_hash.kernel_sources_sha16()
    except Exception:
        now = None
    was = TRAFFIC_STAMP.get(key)
    return {"traffic_sources_sha16": was, "sources_sha16": now, "traffic_stale": (was is None or now is None or was != now)}


def coded_format_roofline(info, ms_spmv, code_w, coded, real_used, traffic, tsrc, launches, survey_bytes, kernel_name):
    """Roofline block of an operator held in the library's DEFAULT format (value codes and / or packed-double vectors).
    The fraction is defined on the bytes the form that RUNS has to move, so it cannot exceed 1:
      * plain coded CSR: nnz * (4 + code bytes) + row pointers + x once + y once;
      * the sliced coded split with the recognised T (x) 1 + 1 (x) T' + D structure (qbh_kronc.hip) streams almost no matrix:
        x, old y, new y (3 vector passes), the tiled copy of x written and read (2), the far row sums written and read (2), one
        diagonal code per row, and the stored T / T' entries (3 B each) -- C3: ~9.5 GB, not the 33 GB of the coded CSR.
    What a CSR of the operator would have moved (the format's bytes, SURVEY 8(d)'s bytes) over the same time is given as
    *_equivalent_GBps: a speed-up figure, NOT a fraction of any roofline."""
    vec_b = 8 if real_used else 16
    fmt_bytes = info.nnz * (4 + (code_w if coded else 16)) + (info.nrows + 1) * 8 + info.nrows * 2 * vec_b
    kronc = bool(info.kron_minor and info.kron_sliced and coded and real_used)      # the sliced coded split (qbh_kronc.hip)
    uniform = kronc and info.kron_far_nnz * 8 < info.nnz                             # T kept once: far entries stored << nnz
    table = bool(getattr(info, "kron_table_kernel", 0)) and real_used
    if table:
        moved = info.nrows * (3 * vec_b + 1) + int(info.kron_far_nnz) * 3
        definition = ("recognised form T (x) 1 + 1 (x) T' + D applied by the row-staged table kernel: rows * (3 * %d + 1) [x, old y, new y, "
                      "diagonal code] + the T / T' tables: the COMPULSORY bytes; the kernel re-reads neighbour rows on top (traffic)" % vec_b)
    elif uniform:
        moved = info.nrows * (7 * vec_b + 1) + int(info.kron_far_nnz) * 3
        definition = ("recognised form T (x) 1 + 1 (x) T' + D: rows * (7 * %d + 1) [x, old y, new y, tiled x written + read, far sums "
                      "written + read, diagonal code] + stored T entries * 3" % vec_b)
    elif kronc:
        moved = info.nnz * 3 + info.nrows * 7 * vec_b
        definition = "sliced coded split, general form: nnz * (2 + 1) + rows * 7 * %d" % vec_b
    else:
        moved = fmt_bytes
        definition = "this format's bytes: nnz*(4 + %d) + (rows+1)*8 + rows*%d" % (code_w if coded else 16, 2 * vec_b)
    return {"bound": "hbm", "kernel": "k_mf_hubbard_row on T, T' and one diagonal code per row (kronc_table_route)" if table else
            "k_kronc_far + k_kronc_near (tiled x written by the producer pass)" if kronc else kernel_name,
            "achieved": round(moved / ms_spmv / 1e6, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(moved / ms_spmv / 1e6 / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "traffic_ratio": (round(traffic / moved, 3) if traffic else None), "traffic_source": tsrc,
            "bytes_per_launch": int(moved), "ms_per_launch": round(ms_spmv, 4), "launches": launches, "bytes_definition": definition,
            "format_bytes_per_launch": int(fmt_bytes), "format_equivalent_GBps": round(fmt_bytes / ms_spmv / 1e6, 2),
            "survey_8d_bytes_per_launch": int(survey_bytes), "survey_8d_equivalent_GBps": round(survey_bytes / ms_spmv / 1e6, 2),
            "note": "frac is on the bytes the running form must move; the *_equivalent_GBps figures divide bytes the kernel does NOT move "
                    "by its time -- speed-ups over a CSR sweep, not roofline fractions"
                    + ("; the table kernel re-reads the ~17 neighbour rows of every up-configuration from HBM (traffic_ratio says how often): on "
                       "its ACTUAL traffic it runs close to the fabric ceiling (C3: 31.8 GB in 4.7 ms = 6.8 TB/s), so its lever is reuse of those "
                       "rows, not a faster pass (DESIGN-history 5.0d item 5)" if table else
                       "; the two passes are bound by L2 line requests and LDS gathers, not by HBM bandwidth (DESIGN-history 4.1g)" if kronc else "")}


def dim_of(w):
    from math import comb
    if w["kind"] == "hubbard":
        return comb(w["n_sites"], w["n_up"]) * comb(w["n_sites"], w["n_dn"])
    if w["kind"] in ("heisenberg_repr", "hubbard_repr", "hubbard_repr_mf"):
        return None                      # known only after the representatives have been enumerated
    return comb(w["n_sites"], w["n_dn"])


def build_operator(w, rows, opts, matrix_free=False, shard=(0, 1)):
    import quantum_basis_amd as q
    if w["kind"] == "hubbard":
        return q.csr_mat.hubbard(w["n_sites"], w["n_up"], w["n_dn"], w["bonds"], t=w["t"], U=w["U"], rows=rows, opts=opts,
                                 matrix_free=matrix_free)
    if w["kind"] == "hubbard_repr_mf":
        from quantum_basis_amd import lattices
        perms, shifts = lattices.translations(*w["trans"])
        return q.csr_mat.hubbard_repr_mf(w["n_sites"], w["n_up"], w["n_dn"], w["bonds"], perms, lattices.characters(shifts, w["k"], w["trans"]),
                                         t=w["t"], U=w["U"], opts=opts)
    if w["kind"] == "hubbard_repr":
        from quantum_basis_amd import lattices
        perms, shifts = lattices.translations(*w["trans"])
        return q.csr_mat.hubbard_repr(w["n_sites"], w["n_up"], w["n_dn"], w["bonds"], perms, lattices.characters(shifts, w["k"], w["trans"]),
                                      t=w["t"], U=w["U"], shard=shard, opts=opts)
    if w["kind"] == "heisenberg_repr":
        from quantum_basis_amd import lattices
        perms, shifts = lattices.translations(*w["trans"])
        return q.csr_mat.heisenberg_repr(w["n_sites"], w["n_dn"], w["bonds"], perms, lattices.characters(shifts, w["k"], w["trans"]),
                                         J=w["J"], shard=shard, opts=opts)
    return q.csr_mat.heisenberg(w["n_sites"], w["n_dn"], w["bonds"], J=w["J"], rows=rows, opts=opts, matrix_free=matrix_free)


def _reforder_args(w):
    if w["kind"] == "hubbard":
        return 1, w["n_sites"], w["n_up"], w["n_dn"]
    if w["kind"] == "heisenberg":
        return 0, w["n_sites"], 0, w["n_dn"]
    raise SystemExit("reference order is defined for the hubbard and heisenberg (full-basis) workloads")


def reference_order_host_csr(name, w, q, stream):
    """Host CSR exactly as the unchanged reference host code hands it over (src/model.cc:619-685): Hermitian-upper, int64
    ia/ja, complex128 values, basis in the reference's Lin order j = Ja[i_a] + Jb[i_b] (src/model.cc:665-670,
    src/basis.cc:1144-1190).  Produced by the device generator + the device permutation qbh_csr_reference_order (checked entry
    by entry against the numpy re-derivation of the reference's pipeline in tests/test_gpu_reforder.py) and downloaded."""
    o = q.make_opts(stream=stream.cuda_stream, value_dict=0, real_fast_path=0, spmv_kernel=q._lib.KERNEL_ROWS, kron_split=0)
    G = build_operator(w, (0, -1), o)
    R = G.reference_order(*_reforder_args(w), opts=o)
    G.destroy()
    d = R.dim
    ia, ja, val = R.download()
    R.destroy()
    rows = np.repeat(np.arange(d, dtype=np.int64), np.diff(ia))
    keep = ja >= rows
    uia = np.zeros(d + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows[keep], minlength=d), out=uia[1:])
    return d, uia, ja[keep].astype(np.int64), val[keep]


# ------------------------------------------------------------------------------------------ CPU baseline ----
def _slab_sample(A, dim, budget_rows, n_slabs=16):
    """budget_rows rows of the operator as n_slabs slabs of consecutive rows spread evenly over the WHOLE operator
    (a rectangular R x dim CSR).  Returns (ia, ja(int64), val, row_index)."""
    R = int(min(A.dim, budget_rows))
    n_slabs = max(1, min(n_slabs, R))
    per = R // n_slabs
    ias, jas, vals, rows = [np.zeros(1, dtype=np.int64)], [], [], []
    base = 0
    for s in range(n_slabs):
        r0 = int(s * (A.dim - per) // max(n_slabs - 1, 1)) if n_slabs > 1 else 0
        ia, ja, val = A.download(r0, r0 + per)
        ias.append(ia[1:] + base)
        base += int(ia[-1])
        jas.append(ja.astype(np.int64))
        vals.append(val)
        rows.append(np.arange(r0, r0 + per, dtype=np.int64))
    return np.concatenate(ias), np.concatenate(jas), np.concatenate(vals), np.concatenate(rows)


def _time_loop(fn, min_reps, budget_s, max_reps):
    fn()                                   # warm (threads, page cache)
    t0 = time.perf_counter()
    n = 0
    while n < min_reps or (time.perf_counter() - t0 < budget_s and n < max_reps):
        fn()
        n += 1
    return (time.perf_counter() - t0) / n, n


def cpu_baseline(A, dim, budget_rows, W, q, stream):
    """CPU legs, all on the GPU box's host cores, all on bounded samples:
    (1) port: the oracle's OpenMP full-storage SpMV + the step's BLAS-1 on 16 row slabs spread over the whole operator
        (full-length x, NUMA first-touch by the OpenMP threads), scaled by dim/rows to one Lanczos iteration;
    (2) the reference's own library call mkl_sparse_z_mv on the same slab (GENERAL descriptor);
    (3) a mid-size operator of the same family run IN FULL on the CPU: Lanczos to convergence with the oracle
        (E0_cpu -> e0_rel_err_vs_cpu against the GPU path on the same operator), mkl_sparse_z_mv with the reference's
        default HERMITIAN-upper descriptor (src/sparse.cc:269-285) beside GENERAL, and scipy's ARPACK eigs() wall time
        beside qbh_iram."""
    import ctypes as C
    from oracle import qb_oracle as qo
    L = qo.lib()
    ia, ja, val, rows = _slab_sample(A, dim, budget_rows)
    R = len(rows)
    ia, ja, val = qo.first_touch(ia), qo.first_touch(ja), qo.first_touch(val)
    slab = qo.Csr.__new__(qo.Csr)
    slab.dim, slab.ia, slab.ja, slab.val, slab.sym = R, ia, ja, val, False      # rectangular R x dim view
    slab.nnz = int(ia[-1])
    slab._c = qo._CSR(R, slab.nnz, 0, slab.ia.ctypes.data, slab.ja.ctypes.data, slab.val.ctypes.data)
    x = qo.first_touch(qo.vec_randomize(dim, 1))
    y = qo.first_touch(np.zeros(R, dtype=np.complex128))
    xs = qo.first_touch(x[:R].copy())
    t = {"spmv": 0.0, "blas": 0.0, "n": 0}

    def step():
        t0 = time.perf_counter()
        L.qbo_multmv2(slab.ref(), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
        t1 = time.perf_counter()
        # the BLAS-1 of one Lanczos step on the same R elements: scale, dot, axpy, nrm2, scale
        np.multiply(y, -0.5, out=y)
        a = qo.dotc(xs, y).real
        y.__isub__(a * xs)
        nrm = qo.nrm2(y)
        np.multiply(y, 1.0 / max(nrm, 1e-300), out=y)
        t2 = time.perf_counter()
        t["spmv"] += t1 - t0
        t["blas"] += t2 - t1
        t["n"] += 1

    step()
    t.update(spmv=0.0, blas=0.0, n=0)
    t_end = time.time() + 8.0
    while t["n"] < 3 or (time.time() < t_end and t["n"] < 50):
        step()
    n = t["n"]
    scale = dim / R
    ms_spmv = 1e3 * t["spmv"] / n * scale
    ms_iter = 1e3 * (t["spmv"] + t["blas"]) / n * scale
    bytes_alg = slab.nnz * 20 + (R + 1) * 8 + R * 32
    out = {"value": round(1e3 / ms_iter, 4), "unit": "lanczos_iters/s", "cores": qo.num_threads(), "kind": "port",
           "sample": "%d rows in 16 slabs spread over all %d rows (%d nnz) of the same operator, full-length x, arrays "
                     "first-touched by the OpenMP threads; %d timed passes of oracle qbo_multmv2 (OpenMP, full storage) + the "
                     "step's BLAS-1; scaled by dim/rows" % (R, dim, slab.nnz, n),
           "spmv_ms_scaled": round(ms_spmv, 3), "spmv_GBps": round(bytes_alg / (t["spmv"] / n) / 1e9, 3)}
    try:        # the reference's own SpMV library, when the image has it: mkl_sparse_z_mv on the same slab
        from oracle import mkl_ref
        if mkl_ref.load() is not None:
            M = mkl_ref.MklCsr(R, slab.ia, slab.ja, slab.val, False, ncols=dim)
            ym = qo.first_touch(np.zeros(R, dtype=np.complex128))
            tm, nm = _time_loop(lambda: M.multmv2(x, ym), 3, 3.0, 30)
            out["mkl"] = {"spmv_ms_scaled": round(1e3 * tm * scale, 3), "spmv_GBps": round(bytes_alg / tm / 1e9, 3),
                          "threads": M.threads(),
                          "call": "mkl_sparse_z_mv, GENERAL descriptor, full storage, no mkl_sparse_optimize (src/sparse.cc:262-289)"}
    except Exception as e:
        out["mkl"] = {"error": repr(e)}
    try:        # the reference's DEFAULT storage at headline scale: mkl_sparse_z_mv with the HERMITIAN-upper descriptor
        from oracle import mkl_ref
        if mkl_ref.load() is not None:
            out["mkl_hermitian_upper"] = _mkl_hermitian_block(A, dim, budget_rows, qo, mkl_ref)
    except Exception as e:
        out["mkl_hermitian_upper"] = {"error": repr(e)}
    try:
        out["midsize"] = cpu_midsize_full_run(W, q, stream)
    except Exception as e:
        out["midsize"] = {"error": repr(e)}
    return out


def _mkl_hermitian_block(A, dim, budget_rows, qo, mkl_ref):
    """mkl_sparse_z_mv with descr {HERMITIAN, UPPER, NON_UNIT} -- the reference's default call (src/sparse.cc:269-285) -- on a
    SQUARE sample of the headline operator: the principal block of `budget_rows` consecutive rows in the middle of the
    operator (entries whose column leaves the block are dropped, the upper triangle of the rest is what the reference would
    store).  Scaled to the whole operator by the nonzeros the block represents: full-storage nnz of the operator / full-storage
    nnz of the block."""
    Rb = int(min(dim, budget_rows))
    r0 = int((dim - Rb) // 2)
    ia, ja, val = A.download(r0, r0 + Rb)
    rows = np.repeat(np.arange(Rb, dtype=np.int64), np.diff(ia))
    c = ja.astype(np.int64) - r0
    inside = (c >= 0) & (c < Rb)
    nnz_block_full = int(inside.sum())
    keep = inside & (c >= rows)
    uia = np.zeros(Rb + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows[keep], minlength=Rb), out=uia[1:])
    uia, uja, uval = qo.first_touch(uia), qo.first_touch(c[keep]), qo.first_touch(val[keep])
    M = mkl_ref.MklCsr(Rb, uia, uja, uval, True)
    x = qo.first_touch(qo.vec_randomize(Rb, 1))
    y = qo.first_touch(np.zeros(Rb, dtype=np.complex128))
    tm, nm = _time_loop(lambda: M.multmv2(x, y), 3, 3.0, 30)
    nnz_full = int(A.info().nnz)
    scale = nnz_full / max(nnz_block_full, 1)
    b_alg = nnz_block_full * 20 + (Rb + 1) * 8 + Rb * 32
    return {"spmv_ms_scaled": round(1e3 * tm * scale, 3), "spmv_GBps": round(b_alg / tm / 1e9, 3), "threads": M.threads(),
            "sample": "principal block of %d rows [%d, %d): %d of its %d full-storage nonzeros stay inside the block, %d stored "
                      "(upper triangle); %d timed passes; scaled by nnz_full(operator) / nnz_full(block) = %.1f"
                      % (Rb, r0, r0 + Rb, nnz_block_full, int(ia[-1]), int(uia[-1]), nm, scale),
            "call": "mkl_sparse_z_mv, descr {HERMITIAN, UPPER, NON_UNIT}, int64 indices: the reference default (src/sparse.cc:269-285)"}


def cpu_midsize_full_run(W, q, stream):
    """A whole (not sampled) operator of the same family small enough for the CPU: the reference pipeline end to end on
    the host (oracle Lanczos, MKL both descriptors, scipy ARPACK) beside the GPU path on the SAME matrix."""
    import scipy.sparse as sp
    import torch
    from oracle import qb_oracle as qo
    name = "hubbard_4x3_half" if W["kind"] == "hubbard" else "chain_22"
    Wm = workloads()[name]
    with torch.cuda.stream(stream):
        G = build_operator(Wm, (0, -1), q.make_opts(stream=stream.cuda_stream, value_dict=0, real_fast_path=0))
        d = G.dim
        fia, fja, fval = G.download()
        Fm = sp.csr_matrix((fval, fja, fia), shape=(d, d))
        rows = np.repeat(np.arange(d), np.diff(fia))
        keep = fja >= rows
        uia = np.zeros(d + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows[keep], minlength=d), out=uia[1:])
        uja, uval = fja[keep].astype(np.int64), fval[keep]
        res = {"workload": name, "dim": int(d), "nnz_full": int(fia[-1]), "nnz_upper": int(uia[-1])}
        # GPU: Lanczos to convergence on this operator (north-star format)
        maxit = 1000
        hess = np.zeros(2 * maxit)
        v = G.vec(2)
        G.randomize(v.at(0), 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m_gpu = q.lanczos(0, maxit - 1, maxit, d, G, None, hess, "sr_val0", device_v=v)
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t0
        e0_gpu = float(q.hess_eigen(hess, maxit, m_gpu, "sr")[0][0])
        v.free()
        # CPU: the same solve with the oracle on the reference's default storage (Hermitian upper)
        O = qo.Csr(d, qo.first_touch(uia), qo.first_touch(uja), qo.first_touch(uval), True)
        vc = np.zeros(2 * d, dtype=np.complex128)
        vc[:d] = qo.vec_randomize(d, 1)
        hc = np.zeros(2 * maxit)
        t0 = time.perf_counter()
        m_cpu = qo.lanczos(0, maxit - 1, maxit, O, vc, hc, "sr_val0")[0]
        t_cpu = time.perf_counter() - t0
        e0_cpu = float(qo.hess_eigen(hc, maxit, m_cpu, "sr")[0][0])
        res.update(gpu_lanczos_steps=int(m_gpu), gpu_s=round(t_gpu, 4), gpu_iters_per_s=round(m_gpu / t_gpu, 2),
                   cpu_lanczos_steps=int(m_cpu), cpu_s=round(t_cpu, 3), cpu_iters_per_s=round(m_cpu / t_cpu, 3),
                   cpu_threads=qo.num_threads(), e0_gpu=e0_gpu, e0_cpu=e0_cpu,
                   e0_rel_err_vs_cpu=abs(e0_gpu - e0_cpu) / abs(e0_cpu))
        # the user call end to end on both sides: locate_E0_lanczos(nev = 1, ncv = 1) = Lanczos + CG eigenvector
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rg = q.locate_E0_lanczos(G, nev=1, ncv=1, maxit=maxit)
            torch.cuda.synchronize()
            tg_ = time.perf_counter() - t0
            t0 = time.perf_counter()
            rc_ = qo.locate_E0_lanczos(O, nev=1, ncv=1, maxit=maxit)
            tc_ = time.perf_counter() - t0
            res["locate_E0"] = {"workload": name, "gpu_s": round(tg_, 4), "cpu_s": round(tc_, 3), "cpu_threads": qo.num_threads(), "cpu_kind": "port (oracle, Hermitian-upper storage)",
                                "gpu_steps": {k_: int(v_) for k_, v_ in rg.steps.items() if k_ in ("E0", "V0")},
                                "cpu_steps": {k_: int(v_) for k_, v_ in rc_.get("steps", {}).items() if k_ in ("E0", "V0")},
                                "e0_rel_diff": abs(rg.E0 - rc_["E0"]) / abs(rc_["E0"]),
                                "eigenvector_overlap": float(abs(np.vdot(rg.eigenvecs, rc_["eigenvecs"])))}
        except Exception as e:
            res["locate_E0"] = {"error": repr(e)}
        b_spmv = int(fia[-1]) * 20 + (d + 1) * 8 + d * 32
        x = qo.first_touch(qo.vec_randomize(d, 2))
        try:
            from oracle import mkl_ref
            if mkl_ref.load() is not None:
                y = qo.first_touch(np.zeros(d, dtype=np.complex128))
                MU = mkl_ref.MklCsr(d, O.ia, O.ja, O.val, True)
                tu, _ = _time_loop(lambda: MU.multmv2(x, y), 3, 2.0, 40)
                MG = mkl_ref.MklCsr(d, qo.first_touch(fia), qo.first_touch(fja.astype(np.int64)), qo.first_touch(fval), False)
                tg, _ = _time_loop(lambda: MG.multmv2(x, y), 3, 2.0, 40)
                res["mkl_hermitian_upper"] = {"spmv_ms": round(1e3 * tu, 3), "spmv_GBps": round(b_spmv / tu / 1e9, 2),
                                              "call": "mkl_sparse_z_mv, descr {HERMITIAN, UPPER, NON_UNIT}: the reference default (src/sparse.cc:269-285)"}
                res["mkl_general"] = {"spmv_ms": round(1e3 * tg, 3), "spmv_GBps": round(b_spmv / tg / 1e9, 2), "threads": MG.threads()}
        except Exception as e:
            res["mkl_hermitian_upper"] = {"error": repr(e)}
        # IRAM: scipy's bundled ARPACK (znaupd/zneupd, mode 1, which='SR', tol=0) on the host vs qbh_iram on the device
        try:
            from scipy.sparse.linalg import eigs
            t0 = time.perf_counter()
            w_cpu = eigs(Fm, k=2, which="SR", ncv=20, tol=0, return_eigenvectors=False)
            t_eigs = time.perf_counter() - t0
            w_cpu = np.sort(w_cpu.real)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, w_gpu, _ = q.iram(d, G, None, 2, 20, 300, "sr")
            torch.cuda.synchronize()
            t_iram = time.perf_counter() - t0
            res["iram"] = {"scipy_eigs_s": round(t_eigs, 3), "qbh_iram_s": round(t_iram, 4), "nev": 2, "ncv": 20,
                           "eig_rel_diff": float(np.max(np.abs(np.sort(w_gpu) - w_cpu) / np.abs(w_cpu)))}
        except Exception as e:
            res["iram"] = {"error": repr(e)}
        G.destroy()
    return res


def bare_spmv_block(A, b_spmv, reps=10):
    """qbh_spmv_dev on a vector the library did not produce -- what an ARPACK reverse-communication caller gets (src/lanczos.cc:393-495):
    no pass of a driver has written the tiled copy of x, so the SpMV of a split operator makes it itself (k_kron_tile8) inside
    the timed launch.  The headline's ms_per_launch is the SpMV inside qbh_lanczos_dev, where the producer pass writes that copy."""
    v = A.vec(2)
    try:
        A.randomize(v.at(0), 5)
        for _ in range(2):
            A.spmv(v.at(0), v.at(A.dim))
        A.sync()
        A.stats(reset=True)
        for _ in range(reps):
            A.spmv(v.at(0), v.at(A.dim))
        A.sync()
        st = A.stats()
        ms = st.ms_spmv / max(1, st.n_spmv)
        return {"call": "qbh_spmv_dev(x, y) on a fresh x (y = H x, no reductions)", "launches": int(st.n_spmv), "ms_per_launch": round(ms, 4),
                "frac": round(b_spmv / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "bytes_per_launch": int(b_spmv),
                "note": "includes the tiled copy of x the split operator needs (one more pass over x: read 16 B, write 16 B per element), "
                        "which inside the solvers is written by the pass that produces x"}
    finally:
        v.free()


def locate_e0_block(A, q, torch, b_spmv, maxit=1000):
    """The call a user of the reference makes: model::locate_E0_lanczos(nev = 1, ncv = 1) (src/model.cc:1123-1316) = Lanczos to
    convergence ("sr_val0", src/lanczos.cc:134-266) + the CG eigenvector (src/lanczos.cc:281-341), every vector resident in HBM.
    Timed stage by stage with the start vectors of the reference (seed 1).  A CG step is one SpMV in the form
    pp = (H - E0) p with <p, pp> fused into its epilogue, k_cg_update (v += a p, r -= a pp, |r|^2: 4 reads + 2 writes) and
    k_xpby (p = r + b p: 2 reads + 1 write, + the tiled copy of p for the split operator's next SpMV)."""
    dim = A.dim
    v = A.vec(4)
    hess = np.zeros(2 * maxit)
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        A.randomize(v.at(0), 1)
        m = q.lanczos(0, maxit - 1, maxit, dim, A, None, hess, "sr_val0", device_v=v)
        e0 = float(q.hess_eigen(hess, maxit, m, "sr")[0][0])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        A.randomize(v.at(2 * dim), 1)
        A.stats(reset=True)
        mcg, accu = q.eigenvec_CG(dim, maxit, 0, A, e0, v.at(2 * dim), v.at(0), v.at(dim), v.at(3 * dim), device=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        st = A.stats()
        # the eigenvector's residual, one more SpMV: |H v - E0 v|
        red = A.spmv(v.at(2 * dim), v.at(0), 1.0, 0.0, -e0, want_red=True)
        resid = float(np.sqrt(max(red[1], 0.0)))
    finally:
        v.free()
    ms_cg = 1e3 * (t2 - t1) / max(mcg, 1)
    b_alg = b_spmv + 144 * dim            # 1 SpMV + nine 16-byte-per-element vector passes
    b_fused = b_spmv + 160 * dim          # what the three fused launches move: + the tiled copy of the new p
    return {"call": "locate_E0_lanczos(nev=1, ncv=1): Lanczos to convergence + CG eigenvector (src/model.cc:1123-1316, src/lanczos.cc:281-341)",
            "seconds_total": round(t2 - t0, 4), "lanczos_s": round(t1 - t0, 4), "lanczos_steps": int(m), "lanczos_ms_per_step": round(1e3 * (t1 - t0) / max(m, 1), 4),
            "cg_s": round(t2 - t1, 4), "cg_steps": int(mcg), "cg_ms_per_step": round(ms_cg, 4), "cg_accuracy": float(accu),
            "cg_spmv_ms_per_launch": round(st.ms_spmv / max(st.n_spmv, 1), 4),
            "cg_bytes_per_step_algorithmic": int(b_alg), "cg_bytes_per_step_fused_kernels": int(b_fused),
            "cg_roofline": {"bound": "hbm", "achieved": round(b_alg / ms_cg / 1e6, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": round(b_alg / ms_cg / 1e6 / HBM_PEAK_GBPS, 4),
                            "bytes_definition": "B_spmv + 144 * dim per CG step (SURVEY a8: 1 SpMV + 9 BLAS-1 passes of 16 B per element), over the WALL time of a step"},
            "e0": e0, "eigenvector_residual_norm": resid}


# ------------------------------------------------------------------------------------------------- main ----
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("QBH_WORKLOAD", "hubbard_4x4_half"))
    ap.add_argument("--format", default=None, choices=["complex128", "fast"],
                    help="what the HEADLINE is timed on: complex128 = north-star format (complex128 CSR values, complex vectors); "
                         "fast = the library default (value dictionary + real fast path where the operator allows)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto (rows), 1 stream, 2 vector, 3 rows")
    ap.add_argument("--npb", type=int, default=0)
    ap.add_argument("--swizzle", type=int, default=2)
    ap.add_argument("--value-dict", type=int, default=None, help="override --format: 1 dictionary-code the value stream (lossless), 0 keep complex128")
    ap.add_argument("--real-fast-path", type=int, default=None, help="override --format: 1 allow 8-byte real gathers / packed-double vectors")
    ap.add_argument("--host-csr", default=None, choices=[None, "reference-order"],
                    help="assemble the operator on the HOST in the reference's Lin order (Hermitian-upper int64 CSR) and hand it to "
                         "qbh_csr_create, as the unchanged reference host code would (workloads up to a few 1e7 nnz)")
    ap.add_argument("--order", default="generator", choices=["generator", "reference"],
                    help="reference: permute the generated operator ON THE DEVICE into the reference's Lin order and fermion convention "
                         "(qbh_csr_reference_order; src/basis.cc:1144-1190) before timing -- the order the unchanged host code hands over")
    ap.add_argument("--no-basis-hint", action="store_true",
                    help="with --order reference / --host-csr: do NOT tell the library what the reference-ordered index means (qbh_opts.basis_kind); "
                         "default is to name the basis, so the operator is held species-major internally and the Kronecker split applies")
    ap.add_argument("--site-cut", type=int, default=0,
                    help="heisenberg (single-species) workloads, complex128 format (qbh_opts.sector_cut): the sites are cut into this many LOW "
                         "sites and the rest, the operator is held class-major internally and split into near (low-half bonds) / far (high-half "
                         "bonds) / cross parts; 0 (default) = the library picks the cut, -1 = never (rows in ascending pattern order)")
    ap.add_argument("--cols16", type=int, default=1, help="qbh_opts.kron_cols16: 1 (library default) the parts of a split operator keep 2-byte columns, 0 int32 columns")
    ap.add_argument("--deterministic", action="store_true", help="qbh_opts.deterministic: static walks, nothing timed at creation (bit-identical a_j / b_j from run to run)")
    ap.add_argument("--no-pipeline", action="store_true", help="qbh_opts.lanczos_pipeline = 0: one host synchronisation per Lanczos step (the loop of ABI <= 501), for A/B runs")
    ap.add_argument("--no-sparse-gather", action="store_true", help="N > 1: qbh_opts.sparse_gather = 0 (every rank's whole tiled block travels to everybody)")
    ap.add_argument("--force-split", action="store_true", help="qbh_opts.kron_split = 2: split a product-basis operator in place even below the 1e8 nonzeros at which the library does it by itself (test rigs)")
    ap.add_argument("--comm-reserve", type=int, default=0, help="N > 1, split shards (qbh_opts.comm_reserve): workgroups the persistent passes leave out of their grids so that "
                    "RCCL's own kernels find a place beside them; 0 (library default) 64, -1 none")
    ap.add_argument("--no-reserve-calibration", action="store_true", help="N > 1: keep qbh_opts.comm_reserve at the library's default instead of trying 64 / 128 / 32 in a few untimed steps")
    ap.add_argument("--no-partition", action="store_true", help="N > 1, hubbard workloads: keep the up configurations in ascending pattern order (qbh_opts.major_partition = 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-converge", action="store_true", help="skip the untimed run to convergence (E0)")
    ap.add_argument("--cpu-rows", type=int, default=2_000_000)
    ap.add_argument("--matrix-free", action="store_true", help="use the matrix-free operator as THE operator (not the CSR north-star path)")
    ap.add_argument("--no-matrix-free", action="store_true", help="skip the extra measurement of the matrix-free operator")
    ap.add_argument("--packed-real", action="store_true", help="matrix-free operator + Lanczos vectors as packed doubles (qbh_lanczos_real_dev); "
                    "forced for the workloads whose complex vectors do not fit one GPU")
    ap.add_argument("--converge", action="store_true", help="run to convergence also for the dim > 1e9 packed-real workloads")
    ap.add_argument("--no-fast-path", action="store_true", help="skip the extra measurement of the coded / real fast path")
    ap.add_argument("--no-plain", action="store_true", help="(kept for older command lines; the uncoded kernel is the headline now)")
    ap.add_argument("--processes", type=int, default=3,
                    help="N = 1 only: the headline (W warm-up + exactly K timed steps on a freshly created operator) is measured in this many "
                         "FRESH child processes, started one after the other before this process touches the GPU; the line reports min / "
                         "median / max and the headline value / roofline.frac are the MEDIAN process's (physical placement differs from "
                         "process to process: DESIGN-history 5.0c).  1: this process only")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-locate", action="store_true", help="skip the timing of the user call locate_E0_lanczos(nev=1, ncv=1) = Lanczos + CG eigenvector")
    args = ap.parse_args()
    if args.child:
        args.no_cpu_baseline = args.no_fast_path = args.no_matrix_free = args.no_converge = args.no_locate = True
        args.processes = 1
    if args.format is None:                  # workloads whose complex128 CSR cannot be stored name their own default
        args.format = workloads()[args.workload].get("format", "complex128")
    fmt_fast = args.format == "fast"
    value_dict = args.value_dict if args.value_dict is not None else (1 if fmt_fast else 0)
    real_fp = args.real_fast_path if args.real_fast_path is not None else (1 if fmt_fast else 0)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N rank processes ourselves -- FRESH processes, from a parent
        # that has made no GPU call (never re-exec a process that has touched the GPU) -- relay rank 0's line and their exit code
        import socket
        import subprocess
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        pr = subprocess.run(cmd, env=env)
        raise SystemExit(pr.returncode)

    import torch
    import torch.distributed as dist
    import quantum_basis_amd as q
    from quantum_basis_amd import _lib, dist as qdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    children, child_failures = [], []
    if args.processes > 1 and ("rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROFILER_") or k.startswith("ROCPROF_") for k in os.environ)):
        # under rocprofv3 this process has already been GPU-initialised by the profiler's preload and its children would inherit
        # the preload: one process, so that kernel statistics and PMC means describe exactly one run
        print("bench.py: profiler preload detected: --processes 1", file=sys.stderr)
        args.processes = 1
    if world == 1 and args.processes > 1 and not args.child and not args.host_csr and not args.packed_real and not workloads()[args.workload].get("packed_real"):
        # fresh processes FIRST: this process has made no GPU call yet (torch.cuda.device_count() does not initialise the GPU
        # on this image), so each child sees the device as the driver's own process would
        import subprocess
        argv = [a for a in sys.argv[1:]]
        for i in range(args.processes):
            pr = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv + ["--child"], capture_output=True, text=True)
            line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
            if pr.returncode != 0 or not line:
                # a child that fails is reported, not fatal: the line then rests on the processes that ran (or on this one alone)
                child_failures.append({"process": i, "rc": pr.returncode, "stderr_tail": pr.stderr[-600:]})
                print("bench.py: WARNING child process %d failed (rc %d): %s" % (i, pr.returncode, pr.stderr[-400:]), file=sys.stderr)
                continue
            children.append(json.loads(line[-1]))
    _lib.require_gpu()                      # no CPU fallback: fail loudly
    # QBH_DIST_BACKEND=gloo lets several ranks share one GPU on a single-GPU test rig (RCCL refuses that)
    backend = os.environ.get("QBH_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    local_rank = dev_index
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL's send / receive kernel runs one workgroup per channel and every one of them needs a place beside the persistent passes
        # (qbh_opts.comm_reserve, default 64 workgroups): keep its grid within that (a host's own setting wins)
        os.environ.setdefault("NCCL_MAX_P2P_NCHANNELS", "64")
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
        args.no_fast_path = args.no_matrix_free = args.no_cpu_baseline = True
        if W.get("packed_real") and not args.converge:
            args.no_converge = True          # hundreds of ~1 s steps: tools/big_lanczos.py does that run (logs under profiles/)
    if args.matrix_free:
        real_fp = 1                          # the matrix-free operators are real; their row-staged kernel is the real form
    dim = dim_of(W)
    row_cuts = None
    if dim is None:
        r0, r1 = 0, -1                   # the generator shards by (rank, world) itself
    elif world > 1 and W["kind"] == "hubbard" and not args.matrix_free and not args.host_csr and value_dict == 0:
        # complex128 CSR of a product-basis operator: shards of WHOLE major (up-configuration) indices keep the Kronecker
        # split in place and exchange the tiled copies of their blocks (dist.kron_row_cuts)
        from math import comb
        row_cuts = qdist.kron_row_cuts(dim, comb(W["n_sites"], W["n_dn"]), world)
        r0, r1 = int(row_cuts[rank]), int(row_cuts[rank + 1])
    else:
        nblk, ranges = qdist.row_partition(dim, world)
        r0, r1 = ranges[rank]
    stream = torch.cuda.Stream(device=device)
    K, Wm = args.steps, max(args.warmup, 2)
    maxit = max(K + Wm + 16, 64)

    def allreduce_host(vals, op):
        t = torch.tensor(vals, dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t, op=op)
        return [float(z) for z in t.tolist()]

    def timed_lanczos(A, packed):
        """Wm untimed + exactly K timed Lanczos steps on operator A.  lanczos() cannot make a single step from k = 0 (the
        reference's do-while runs once more after the bootstrap step, src/lanczos.cc:167-193), so the recurrence is always
        started by at least two untimed steps.  Returns dict(elapsed, steps, ms_spmv, n_spmv, n_real)."""
        n = A.dim
        v = A.vec(1 if packed else 2)           # packed doubles: n complex128 = the two slots of n doubles
        hess = np.zeros(2 * maxit)

        def fresh_start(seed):
            if packed:
                import ctypes
                _lib.check(_lib.lib().qbh_vec_randomize_real(A.handle, v.ptr, ctypes.c_uint32(seed)), "qbh_vec_randomize_real")
            else:
                A.randomize(v.at(0), seed)
            hess[:] = 0.0
            return 0

        def call(k0, nsteps, mx, hs):
            if packed:
                return q.lanczos_real(k0, nsteps, mx, A, v, hs)
            return q.lanczos(k0, nsteps, mx, n, A, None, hs, "sr_val0", device_v=v)

        def run_steps(k, nsteps, seed):
            left, total = nsteps, 0
            while left > 0:
                m = call(k, left, maxit, hess)
                left -= m - k
                total += m - k
                k = m
                if left > 0:          # converged / broke down before np steps: new Krylov space
                    seed += 1
                    k = fresh_start(seed)
            return k, seed, total

        seed = 1
        k = fresh_start(seed)
        k, seed, _ = run_steps(k, Wm, seed)
        A.stats(reset=True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k, seed, done = run_steps(k, K, seed)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        st = A.stats()
        mine = [st.ms_spmv / max(st.n_spmv, 1), st.ms_gather / max(st.n_gather, 1)]
        per_rank = None
        if world > 1:            # every rank's own kernel and exchange time (the headline takes the slowest rank)
            slots = [0.0] * (2 * world)
            slots[2 * rank], slots[2 * rank + 1] = mine
            slots = allreduce_host(slots, dist.ReduceOp.SUM)
            per_rank = [(slots[2 * r], slots[2 * r + 1]) for r in range(world)]
        elapsed, ms_spmv, ms_gather = allreduce_host([elapsed] + mine, dist.ReduceOp.MAX)
        e0 = steps_e0 = None
        if not args.no_converge:      # untimed: the same solver to convergence (E0 parity across N / formats / vs the oracle tests)
            maxit2 = 1000
            hess2 = np.zeros(2 * maxit2)
            fresh_start(1)
            m = call(0, maxit2 - 1, maxit2, hess2)
            ritz, _ = q.hess_eigen(hess2, maxit2, m, "sr")
            e0, steps_e0 = float(ritz[0]), int(m)
        v.free()
        return dict(elapsed=elapsed, steps=done, ms_spmv=ms_spmv, n_spmv=int(st.n_spmv), n_real=int(st.n_spmv_real), e0=e0,
                    steps_e0=steps_e0, ms_gather=ms_gather, n_gather=int(st.n_gather), per_rank=per_rank)

    create = None
    with torch.cuda.stream(stream):
        opts = q.make_opts(device=local_rank, stream=stream.cuda_stream, spmv_kernel=args.kernel,
                           nnz_per_block=args.npb, xcd_swizzle=args.swizzle,
                           value_dict=value_dict, real_fast_path=real_fp, profile=1, deterministic=1 if args.deterministic else 0,
                           kron_cols16=args.cols16, lanczos_pipeline=0 if args.no_pipeline else 1, sparse_gather=0 if args.no_sparse_gather else 1,
                           comm_reserve=args.comm_reserve)
        if world > 1 and W["kind"] == "hubbard" and not args.matrix_free and not args.host_csr and value_dict == 0 and not args.no_partition:
            # the up configurations in the order of a recursive bisection of the hop graph into `world` parts: every rank's far part then reads
            # far fewer of its peers' major indices, which is what the personalised exchange carries (qbh_opts.major_partition)
            opts.major_partition = world
        if args.force_split:
            opts.kron_split = 2
        opts.sector_cut = args.site_cut if world == 1 else -1          # qbh_opts.sector_cut: 0 = the library picks the cut of a heisenberg sector, -1 never
        t_gen = time.time()
        hint = (not args.no_basis_hint) and W["kind"] == "hubbard" and world == 1 and value_dict == 0
        if args.host_csr:
            hd, hia, hja, hval = reference_order_host_csr(args.workload, W, q, stream)
            t_host = time.time() - t_gen
            assert hd == dim
            t_gen = time.time()
            if hint:            # what the unchanged host code would say once (qbh_opts_set_default)
                opts.basis_kind, opts.n_sites, opts.n_up, opts.n_dn = q._lib.BASIS_REF_FERMION2, W["n_sites"], W["n_up"], W["n_dn"]
            A = q.csr_mat(dim, hia, hja, hval, sym=True, opts=opts, rows=None if world == 1 else (r0, r1))
            ci = A.info()
            create = {"create_s": round(ci.create_ms * 1e-3, 4), "input_bytes": int(ci.create_bytes_in),
                      "create_GBps_of_input": round(ci.create_bytes_in / ci.create_ms / 1e6, 2),
                      "host_arrays_s (device generator + device permutation + download, not the product)": round(t_host, 2),
                      "entry": "qbh_csr_create" if world == 1 else "qbh_csr_create_rows", "storage": "Hermitian-upper, int64 ia/ja, complex128",
                      "order": "reference Lin order (src/model.cc:665-670)", "nnz_upper": int(hia[-1]),
                      "basis_hint": ("qbh_opts.basis_kind = QBH_BASIS_REF_FERMION2" if hint else None), "basis_hint_accepted": bool(ci.basis_internal)}
        elif args.order == "reference":
            if world != 1 or args.matrix_free or value_dict != 0:
                raise SystemExit("--order reference: one GPU, stored operator, --format complex128")
            # the source of the permutation is never applied: row kernel geometry only, no second (split) copy, nothing timed
            src_opts = q.make_opts(device=local_rank, stream=stream.cuda_stream, spmv_kernel=q._lib.KERNEL_ROWS, value_dict=0,
                                   real_fast_path=0, kron_split=0)
            G = build_operator(W, (r0, r1), src_opts)
            # created as a plain CSR in the reference's order (three copies of a 116 GB matrix do not fit: the source of the
            # permutation must go first); then EITHER the caller names the basis OR -- --no-basis-hint, what the unchanged host
            # code can say: nothing -- the library's own search runs (qbh_opts.basis_detect; the same code qbh_csr_create runs)
            detect = W["kind"] == "hubbard" and value_dict == 0 and not hint
            if hint or detect:
                opts.kron_split = 0
                opts.basis_detect = 0
            A = G.reference_order(*_reforder_args(W), opts=opts)
            G.destroy()
            if hint or detect:
                torch.cuda.synchronize()
                t_b = time.time()
                if hint:
                    named = A.set_basis(q._lib.BASIS_REF_FERMION2, W["n_sites"], W["n_up"], W["n_dn"])
                else:
                    named = A.set_basis(q._lib.BASIS_DETECT, 0, 0, 0)
                torch.cuda.synchronize()
                bi = A.info()
                create = {"basis_hint": ("qbh_csr_set_basis(QBH_BASIS_REF_FERMION2, %d, %d, %d)" % (W["n_sites"], W["n_up"], W["n_dn"])) if hint else
                          "none: the library searched the two-species bases of this dimension by itself (qbh_opts.basis_detect)",
                          "basis_hint_accepted": bool(named), "basis_detected": bool(bi.basis_detected),
                          "basis_found": [int(bi.basis_n_sites), int(bi.basis_n_up), int(bi.basis_n_dn)] if named else None,
                          "detect_ms (candidates tried through the one-pass pre-check, incl. the permutation of the one that verified)": round(bi.basis_detect_ms, 1),
                          "set_basis_s": round(time.time() - t_b, 3)}
        else:
            A = build_operator(W, (r0, r1), opts, matrix_free=args.matrix_free, shard=(rank, world))
        torch.cuda.synchronize()
        t_gen = time.time() - t_gen
        info = A.info()
        if dim is None:
            dim = int(info.ncols)
        exchange_kind = None
        if world > 1:
            # QBH_RCCL_LIB (the test-only librccl stub, tests/stub_rccl/) puts the library's own communicator under ranks that
            # share one GPU: torch's rendezvous then runs over gloo, the exchange still through qbh_comm_create_rccl
            want_native = (backend == "nccl" or bool(os.environ.get("QBH_RCCL_LIB"))) and not os.environ.get("QBH_PY_HOOKS")
            native_err = None
            if want_native:
                # the library's own RCCL communicator (qbh_comm_create_rccl): no Python in the SpMV loop.  Every rank must
                # end up on the same path, so the outcome is agreed on before anything is exchanged through it.
                try:
                    comm = qdist.NativeComm(dim, rank=rank, world=world, cuts=row_cuts).attach(A)
                    ok = 1.0
                except Exception as e:          # e.g. librccl not loadable: all ranks fall back together
                    comm, ok, native_err = None, 0.0, repr(e)
                if allreduce_host([ok], dist.ReduceOp.MIN)[0] < 1.0:
                    if comm is not None:
                        comm.detach(A)
                    want_native = False
            if want_native:
                exchange_kind = "native RCCL (qbh_comm_create_rccl)"
            else:
                comm = qdist.ShardComm(dim, rank=rank, world=world, device=device, stream=stream, cuts=row_cuts).attach(A)  # noqa: F841
                exchange_kind = "torch.distributed hooks (%s)" % backend + (" [native communicator failed: %s]" % native_err if native_err else "")
        nnz_total = int(allreduce_host([float(info.nnz)], dist.ReduceOp.SUM)[0])
        reserve_cal = None
        if world > 1 and info.kron_minor and args.comm_reserve == 0 and not args.no_reserve_calibration:
            # qbh_opts.comm_reserve: how many workgroups the persistent passes must leave out for RCCL's kernels depends on how many
            # channels RCCL opens on THIS node, which no API tells: a few untimed Lanczos steps with 64 / 128 / 32 left out, max over
            # ranks, the fastest stays (ties: 64).  Through the stand-in (ranks sharing a GPU) the answer means nothing and harms nothing.
            cal_v = A.vec(2)
            cal_h = np.zeros(2 * 64)
            A.randomize(cal_v.at(0), 1)
            kc = q.lanczos(0, 2, 64, A.dim, A, None, cal_h, "sr_val0", device_v=cal_v)
            timing = {}
            for cand in (64, 128, 32):
                A.set_option("comm_reserve", cand)
                kc = q.lanczos(kc, 1, 64, A.dim, A, None, cal_h, "sr_val0", device_v=cal_v)
                torch.cuda.synchronize()
                dist.barrier()
                tc = time.perf_counter()
                kc = q.lanczos(kc, 4, 64, A.dim, A, None, cal_h, "sr_val0", device_v=cal_v)
                torch.cuda.synchronize()
                timing[cand] = allreduce_host([(time.perf_counter() - tc) * 250.0], dist.ReduceOp.MAX)[0]       # ms per step, slowest rank
            cal_v.free()
            best = 64                                    # the library's default unless another is more than 1 % faster
            for cand in (128, 32):
                if timing[cand] < 0.99 * timing[best]:
                    best = cand
            A.set_option("comm_reserve", best)
            reserve_cal = {"ms_per_step_by_workgroups_left_out": {str(c): round(t, 4) for c, t in timing.items()}, "chosen": best}
        head = timed_lanczos(A, packed_real)

    # algorithmic bytes of ONE SpMV launch on this rank: nnz*(16+4) + (rows+1)*8 + x once + y once
    bytes_launch = info.nnz * 20 + (info.nrows + 1) * 8 + (dim if world > 1 else info.nrows) * 16 + info.nrows * 16
    ms_spmv = head["ms_spmv"]
    achieved = bytes_launch / (ms_spmv * 1e-3) / 1e9 if ms_spmv > 0 else 0.0
    coded = bool(info.value_dict)
    real_used = head["n_real"] > 0
    code_w = 0 if not coded else (1 if info.value_dict <= 256 else 2)
    tkey = "%s|%s|%s" % (args.workload, KERNEL_KEY[info.kernel], "dict" if coded else "plain") + ("|real" if real_used else "")
    if info.kron_minor:
        tkey += ("|kron_sliced" if info.kron_sliced else "|kron") + ("|inplace" if info.kron_inplace else "") + ("|cut%d" % int(info.basis_n_up) if info.kron_classes > 1 else "")
        tkey += "|c16" if info.kron_cols16 == 3 else ""           # 2-byte columns in both parts: another stream, another traffic entry
        tkey += "|table" if info.kron_table_kernel else ""        # the table route moves other bytes than the sliced passes
    tkey += "|reforder" if args.order == "reference" else ""
    traffic, tsrc = traffic_of(tkey) if world == 1 and not args.host_csr else (None, None)
    if coded or real_used:
        roof = coded_format_roofline(info, ms_spmv, code_w, coded, real_used, traffic, tsrc, head["n_spmv"], bytes_launch, KERNEL_NAME[info.kernel])
        dtype = "f64 real (1-byte value codes, packed-double vectors; bit-identical to complex128)" if real_used else \
                "complex128 vectors, %d-byte value codes" % code_w
    else:
        roof = {"bound": "hbm", "kernel": ("k_spmv_wave2 (Kronecker split in place: far + near launches)" if info.kron_minor else KERNEL_NAME[info.kernel]),
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_ratio": (round(traffic / bytes_launch, 3) if traffic else None), "traffic_source": tsrc,
                "bytes_per_launch": bytes_launch, "ms_per_launch": round(ms_spmv, 4), "launches": head["n_spmv"],
                "bytes_definition": "SURVEY 8(d) algorithmic bytes: nnz*(16+4) + (rows+1)*8 + rows*16 (x once) + rows*16 (y once)",
                "note": "traffic (HBM bytes per launch from separate rocprofv3 --pmc passes, NOT measured in this run; see traffic_source) "
                        "exceeds the algorithmic bytes by the x gathers that miss the caches"}
        dtype = "complex128"
    out = {
        "metric": "lanczos_iters_per_sec", "value": round(head["steps"] / head["elapsed"], 4), "unit": "lanczos_iters/s",
        "n_gpus": world, "steps": head["steps"], "warmup": Wm, "ms_per_step": round(1e3 * head["elapsed"] / head["steps"], 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype,
        "data": "synthetic", "config": {"workload": args.workload, "dim": dim, "nnz_full": nnz_total,
                                         "rows_per_gpu": info.nrows, "parallelism": "row-shard x%d" % world,
                                         "exchange": exchange_kind,
                                         "kernel": KERNEL_KEY[info.kernel], "format": ("complex128 CSR values + %s, complex128 vectors" % ("2-byte columns held (int32 counted in the roofline bytes)" if (info.kron_minor and int(info.kron_cols16)) else "int32 columns"))
                                         if not (coded or real_used) else "value codes %s, real fast path %s" % (coded, real_used),
                                         "value_dict": info.value_dict, "real_gather": real_used, "deterministic": bool(args.deterministic),
                                         "kron_split": ({"minor": int(info.kron_minor), "band": int(info.kron_band), "far_nnz": int(info.kron_far_nnz), "far_sliced": bool(info.kron_sliced),
                                                         "columns": {0: "int32", 1: "near part 2-byte, far part int32", 2: "near part int32, far part 2-byte",
                                                                     3: "2-byte in both parts (relative to the wave block's base; qbh_csr_download re-derives int32)"}[int(info.kron_cols16)],
                                                         "launches_per_spmv": ("k_kron_tile_re + k_kronc_far (sliced, whole lines of the tiled x) + k_kronc_near (sliced, x block in LDS, epilogue)"
                                                                               if (info.value_dict and info.kron_sliced) else
                                                                               "k_kron_tile_re + k_spmv_rows (near) + k_spmv_rows (far, tiled rows and columns; QBH_KRON_CODED=1)"
                                                                               if info.value_dict else
                                                                               ("k_zero_cut_groups + k_spmv_wave2<.,3> (far, sliced) + k_spmv_wave2<.,2> (near); the tiled copy of x is written by the "
                                                                                "pass that produces x (k_axpy_norm_tile8), k_kron_tile only in front of a driver's first step"
                                                                                if world == 1 else
                                                                                "tiled block (8-byte real parts when the solve is real: qbh_opts.real_wire) -> all-gather in parts || k_spmv_wave2<.,1> (near, own x) ; per part: k_kron_place (gathered blocks -> tiled order of the whole x) + k_spmv_wave2<.,3> (far, the one-GPU kernel, 2-byte columns) ; k_kron_combine")
                                                                               if info.kron_sliced else "k_kron_tile + k_spmv_wave2<.,0> (far) + k_spmv_wave2<.,2> (near)"),
                                                         "in_place": bool(info.kron_inplace)}
                                                        if info.kron_minor else None),
                                         "basis_internal": ({1: "species-major (index = up * C(n, n_dn) + down), vectors translated at the seams",
                                                             2: "class-major cut sector (%d low sites, %d classes, cross part %d nonzeros), vectors translated at the seams"
                                                                % (int(info.basis_n_up), info.kron_classes, info.kron_cross_nnz),
                                                             3: "the rows of every down block orbit by orbit of the up patterns (qbh_opts.sector_orbit); vectors translated at the seams"}
                                                            .get(info.basis_internal)),
                                         "operator_source": "host CSR in reference order through qbh_csr_create" if args.host_csr else
                                         "device generator, permuted on the device into the reference's Lin order and fermion convention "
                                         "(qbh_csr_reference_order)" if args.order == "reference" else "device generator",
                                         "build_s": round(t_gen, 3)},
        "roofline": roof, "e0": head["e0"], "lanczos_steps_to_converge": head["steps_e0"],
    }
    if roof.get("traffic") is not None:
        roof.update(traffic_stamp(tkey))
        if roof["traffic_stale"]:
            print("bench.py: WARNING roofline.traffic was measured on other kernel sources (%s, now %s): re-run tools/profile_bench.sh"
                  % (roof["traffic_sources_sha16"], roof["sources_sha16"]), file=sys.stderr)
    if child_failures and not children:
        out["processes"] = {"n": 0, "failed": child_failures, "what": "every fresh child process failed: the line is this process's own measurement"}
    if children:
        # The headline is the MEDIAN of the fresh processes (each: its own operator, W warm-up + exactly K timed steps); the
        # spread is printed beside it.  What this process measured afterwards on its own operator is listed, not used.
        runs = sorted(children, key=lambda c: c["roofline"]["ms_per_launch"])
        med = runs[len(runs) // 2]
        mine = {"ms_per_launch": roof["ms_per_launch"], "frac": roof["frac"], "value": out["value"], "ms_per_step": out["ms_per_step"]}
        out["processes"] = {
            "n": len(runs), "requested": args.processes, "incomplete": len(runs) < args.processes,
            "what": "fresh child processes run one after the other BEFORE this process made any GPU call; each builds its own operator "
                    "and times W warm-up + exactly K Lanczos steps; headline value / ms_per_step / roofline = the median process",
            "ms_per_launch": [c["roofline"]["ms_per_launch"] for c in runs], "frac": [c["roofline"]["frac"] for c in runs],
            "value": [c["value"] for c in runs], "ms_per_step": [c["ms_per_step"] for c in runs],
            "frac_min": runs[-1]["roofline"]["frac"], "frac_median": med["roofline"]["frac"], "frac_max": runs[0]["roofline"]["frac"],
            "this_process_after_the_children": mine}
        if child_failures:
            out["processes"]["failed"] = child_failures
        for k in ("value", "steps", "ms_per_step"):
            out[k] = med[k]
        for k in ("achieved", "frac", "ms_per_launch", "launches", "traffic_ratio"):
            if k in med["roofline"]:
                roof[k] = med["roofline"][k]
        roof["frac_min"], roof["frac_max"] = runs[-1]["roofline"]["frac"], runs[0]["roofline"]["frac"]
    if world > 1:
        # link bytes per GPU reported separately from the HBM bytes.  ms_per_gather is the event-timed duration of
        # the all-gather on RCCL's side stream (native communicator), max over ranks; it overlaps the locally-owned columns.
        elem = int(A.info().wire_element_bytes) or (8 if real_used else 16)      # what the last gather carried (qbh_csr_info.wire_element_bytes)
        sparse_x, need_frac = bool(A.info().gather_sparse), float(A.info().gather_needed_frac)
        out["exchange"] = {"bytes_received_per_gpu_per_spmv": int(elem * dim * (world - 1) / world * (need_frac if sparse_x else 1.0)), "element_bytes": elem,
                           # personalised exchange (qbh_opts.sparse_gather): every rank sends each peer only the major indices that peer's far / cross
                           # entries read; needed_frac_rank0 = that share of the all-gather for THIS rank (it differs from rank to rank)
                           "personalised": sparse_x, "needed_frac_rank0": round(need_frac, 4), "major_partition": int(A.info().major_partition),
                           "ms_per_gather": round(head["ms_gather"], 4) if head["ms_gather"] > 0 else None,
                           "gathers": head["n_gather"],
                           # > 1: the tiled blocks travel as that many band ranges and the far pass follows range by range; the
                           # per-rank ms_spmv then contains whatever the far pass waited for the later ranges
                           "gather_parts": int(A.info().gather_parts),
                           # what the ranks agreed on when the communicator was attached (qbh_csr_set_comm is collective): split shards
                           # exchanging the tiled copies of their blocks, or the plain blocks of unsplit shards
                           "tiled_blocks_of_split_shards": bool(A.info().kron_minor),
                           # qbh_opts.comm_reserve: workgroups the persistent passes of a split shard leave out of their grids (room for RCCL's kernels)
                           "comm_reserve_workgroups": ((reserve_cal["chosen"] if reserve_cal else 0 if args.comm_reserve < 0 else (args.comm_reserve // 8) * 8 or 64) if A.info().kron_minor else None),
                           "comm_reserve_calibration": reserve_cal,
                           "allreduce": "<= 3 doubles per reduction point"}
        # per rank: SpMV kernel ms (both parts of a split shard), gather ms on the side stream, how much of the gather the
        # locally-owned columns hide, and the rank's own roofline on ITS algorithmic bytes (local nnz, rows, the whole x)
        loc = [0.0] * (2 * world)
        loc[2 * rank], loc[2 * rank + 1] = float(info.nnz), float(info.nrows)
        loc = allreduce_host(loc, dist.ReduceOp.SUM)
        step_ms = 1e3 * head["elapsed"] / max(head["steps"], 1)
        ranks = []
        for r in range(world):
            ms_k, ms_g = head["per_rank"][r]
            b_r = loc[2 * r] * 20 + (loc[2 * r + 1] + 1) * 8 + dim * 16 + loc[2 * r + 1] * 16
            exposed = max(0.0, step_ms - ms_k - 96.0 * loc[2 * r + 1] / 6.0e9)      # what the step waits beyond kernel + BLAS-1 at ~6 TB/s
            ranks.append({"rank": r, "rows": int(loc[2 * r + 1]), "nnz": int(loc[2 * r]), "ms_spmv": round(ms_k, 4),
                          "ms_gather": round(ms_g, 4) if ms_g > 0 else None,
                          "gather_hidden_frac": (round(max(0.0, min(1.0, 1.0 - exposed / ms_g)), 3) if ms_g > 0 else None),
                          "roofline_frac": round(b_r / max(ms_k, 1e-9) / 1e6 / HBM_PEAK_GBPS, 4)})
        out["per_rank"] = ranks
        ms_all = [r_["ms_spmv"] for r_ in ranks]
        out["exchange"]["spmv_ms_imbalance_max_over_min"] = round(max(ms_all) / max(min(ms_all), 1e-9), 3)
    if create:
        out["create"] = create
    if packed_real:
        out["dtype"] = "f64 (real operator, Lanczos vectors stored as packed doubles)"
        out["config"]["vectors"] = "2 x %.1f GB packed doubles (qbh_lanczos_real_dev)" % (dim * 8e-9)
    if args.matrix_free or W["kind"].endswith("_mf"):
        # A matrix-free operator streams no matrix: its algorithmic bytes are the VECTORS of y <- a H x + b y (x read once,
        # old y read once, new y written once) -- never the bytes a CSR of the operator would move.
        vb = 8 if (real_used or packed_real) else 16
        vec_bytes = int(info.nrows) * vb * 3
        mf_key = "%s|matrix_free|plain" % args.workload + ("|real" if vb == 8 else "")
        mtr, msrc = traffic_of(mf_key) if world == 1 else (None, None)
        out["config"]["kernel"] = "matrix_free"
        out["roofline"] = {"bound": "hbm", "kernel": {"hubbard": "k_mf_hubbard",
                                                        "hubbard_repr_mf": "k_mf_sector_orb + k_sec_remainder + k_sec_reduce (rows orbit by orbit)"
                                                        if int(info.basis_internal) == 3 else "k_mf_sector + k_sec_remainder + k_sec_reduce"}.get(W["kind"], "k_mf_heis"),
                           "achieved": round(vec_bytes / ms_spmv / 1e6, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": round(vec_bytes / ms_spmv / 1e6 / HBM_PEAK_GBPS, 4), "traffic": mtr,
                           "traffic_ratio": (round(mtr / vec_bytes, 2) if mtr else None), "traffic_source": msrc,
                           "bytes_per_launch": vec_bytes, "ms_per_launch": round(ms_spmv, 4), "launches": head["n_spmv"],
                           "bytes_definition": "matrix-free operator: vectors only, rows * %d B * 3 (x once, old y once, new y once); "
                                               "the hop / block tables are a few MB" % vb,
                           "csr_equivalent_GBps": round(achieved, 2),
                           "note": "MATRIX-FREE operator (SURVEY 8f-1): no CSR is stored.  frac is on the vector bytes; the traffic "
                                   "ratio says how far the gathers are from touching every vector line once.  csr_equivalent_GBps "
                                   "(bytes a CSR of the same operator would move / kernel time) is informational, not a roofline"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.matrix_free:
        try:
            out["cpu_baseline"] = cpu_baseline(A, dim, args.cpu_rows, W, q, stream)
            mid = out["cpu_baseline"].get("midsize", {})
            if "e0_rel_err_vs_cpu" in mid:
                out["e0_rel_err_vs_cpu"] = mid["e0_rel_err_vs_cpu"]
        except Exception as e:      # the baseline is reported, never required
            out["cpu_baseline"] = {"value": None, "unit": "lanczos_iters/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0 and world == 1 and not args.no_locate and not args.matrix_free and not packed_real and not (coded or real_used):
        try:
            with torch.cuda.stream(stream):
                out["bare_spmv"] = bare_spmv_block(A, bytes_launch)
        except Exception as e:
            out["bare_spmv"] = {"failed": repr(e)}
        try:
            with torch.cuda.stream(stream):
                out["locate_E0"] = locate_e0_block(A, q, torch, bytes_launch)
                mid = out.get("cpu_baseline", {}).get("midsize", {})
                if "locate_E0" in mid:          # the same call on the mid-size operator, GPU beside the CPU oracle
                    out["locate_E0"]["midsize"] = mid["locate_E0"]
        except Exception as e:
            out["locate_E0"] = {"error": repr(e)}
    n = A.dim
    if world == 1:
        A.destroy()          # the extra blocks below build their own operators: give the HBM back first (C4 substitute: 157 GB)
        torch.cuda.synchronize()
    if world == 1 and not (coded or real_used) and not args.no_fast_path and not args.matrix_free and not args.host_csr and args.order != "reference":
        # the library's default path for this operator (lossless value codes; real operator + real vectors -> packed doubles):
        # same step definition, same K, its own roofline on its own format's bytes
        try:
            with torch.cuda.stream(stream):
                F = build_operator(W, (r0, r1), q.make_opts(device=local_rank, stream=stream.cuda_stream, xcd_swizzle=args.swizzle, profile=1),
                                   shard=(rank, world))
                fi = F.info()
                fp = timed_lanczos(F, False)
                f_coded = bool(fi.value_dict)
                f_real = fp["n_real"] > 0
                f_cw = 0 if not f_coded else (1 if fi.value_dict <= 256 else 2)
                f_kronc = bool(fi.kron_minor > 0 and fi.kron_sliced and f_coded and f_real)      # the sliced coded split (qbh_kronc.hip)
                fkey = "%s|%s|%s" % (args.workload, KERNEL_KEY[fi.kernel], "dict" if f_coded else "plain") + ("|real" if f_real else "") + ("|kron_sliced" if f_kronc else "") + \
                       ("|table" if fi.kron_table_kernel else "")
                ftr, fsrc = traffic_of(fkey)
                froof = coded_format_roofline(fi, fp["ms_spmv"], f_cw, f_coded, f_real, ftr, fsrc, fp["n_spmv"],
                                              fi.nnz * 20 + (fi.nrows + 1) * 8 + fi.nrows * 32, KERNEL_NAME[fi.kernel])
                if ftr is not None:
                    froof.update(traffic_stamp(fkey))
                out["fast_path"] = {
                    "value": round(fp["steps"] / fp["elapsed"], 4), "unit": "lanczos_iters/s", "steps": fp["steps"],
                    "ms_per_step": round(1e3 * fp["elapsed"] / fp["steps"], 4),
                    "dtype": "f64 real (1-byte value codes, packed-double vectors; bit-identical to complex128)" if f_real
                             else "complex128 vectors, %d-byte value codes" % f_cw,
                    "value_dict": fi.value_dict, "real_gather": f_real, "e0": fp["e0"],
                    "e0_rel_diff_vs_complex128": (abs(fp["e0"] - head["e0"]) / abs(head["e0"])) if (fp["e0"] is not None and head["e0"]) else None,
                    "kron_split": ({"minor": int(fi.kron_minor), "band": int(fi.kron_band),
                                    "form": "recognised as T (x) 1 + 1 (x) T' + D: applied by the row-staged table kernel (qbh_csr_info.kron_table_kernel)"
                                    if fi.kron_table_kernel else "both parts sliced in groups of 16 rows, near x block in LDS",
                                    "stored_far_entries": int(fi.kron_far_nnz)} if f_kronc else None),
                    "roofline": froof,
                    "note": "lossless: values are dictionary-coded, and a real operator applied to real vectors gathers 8-byte real parts"}
                F.destroy()
        except Exception as e:
            out["fast_path"] = {"error": repr(e)}
    if world == 1 and W["kind"] in ("hubbard", "heisenberg") and not args.no_matrix_free and not args.matrix_free and not args.host_csr and args.order != "reference":
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
                mtr, msrc = traffic_of("%s|matrix_free|plain%s" % (args.workload, "|real" if ms_.n_spmv_real > 0 else ""))
                out["matrix_free_" + W["kind"]] = {"lanczos_iters_per_s": round((mk2 - mk) / tm, 4), "steps": int(mk2 - mk),
                                                   "spmv_ms_per_launch": round(mms, 4),
                                                   "dtype": "f64 real (packed-double vectors)" if ms_.n_spmv_real > 0 else "complex128",
                                                   "table_bytes": int(M.info().bytes_matrix),
                                                   "kernel": ("k_mf_hubbard_row" if ms_.n_spmv_real > 0 else "k_mf_hubbard") if W["kind"] == "hubbard" else "k_mf_heis",
                                                   "traffic": mtr, "traffic_source": msrc,
                                                   "note": "no stored matrix; not the CSR north-star path"}
                mv.free()
                M.destroy()
        except Exception as e:
            out["matrix_free_" + W["kind"]] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
