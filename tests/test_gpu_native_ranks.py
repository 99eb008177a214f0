"""SURVEY 8(e) through the library's NATIVE communicator with N > 1 ranks before a multi-GPU node ever runs it: N C++ host
processes (tests/cxx/sharded_main.cpp) share this box's one GPU and exchange through tests/stub_rccl/librccl_stub.so -- a
test-only stand-in for librccl (real RCCL refuses two ranks on one device) selected with QBH_RCCL_LIB -- so that
qbh_comm_create_rccl, the collective qbh_csr_set_comm, ncclAllGather for uniform blocks, the grouped ncclSend / ncclRecv
all-gather-v for ragged ones, the gather in band ranges with its per-part events, the tiled exchange of split shards, the fall
back when one rank cannot split, and the all-reduce of the Lanczos / CG scalars all execute with real peers.  The stub FAILS
where real RCCL would hang (a receive no send matches, element counts that differ).  Every rank must report the one-rank
E0, a_j / b_j, step counts and its slice of the one-rank eigenvector.  (The matvec served: src/sparse.cc:262-289.)"""
import json
import math
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import quantum_basis_amd as q
from quantum_basis_amd import lattices
from test_cxx_adaptor import ROOT, _build
from test_rccl_stub import STUB, _stub

pytestmark = pytest.mark.gpu
S_MINOR = 70            # Hubbard 4x2, 4 + 4 electrons in the generator's species-major order: index = up * 70 + down, dim 4900


@pytest.fixture(scope="module")
def rig():
    _stub()             # builds the stub when the .so is not there
    with tempfile.TemporaryDirectory() as tmp:
        exe = _build(tmp, "sharded_main")
        G = q.csr_mat.hubbard(8, 4, 4, lattices.square(4, 2), t=1.0, U=1.1, opts=q.make_opts(kron_split=0, value_dict=0, real_fast_path=0))
        ia, ja, val = G.download()
        dim = G.dim
        G.destroy()
        path = os.path.join(tmp, "csr.bin")
        with open(path, "wb") as f:
            np.array([dim, len(ja), 0], dtype=np.int64).tofile(f)
            ia.astype(np.int64).tofile(f), ja.astype(np.int64).tofile(f), val.tofile(f)
        yield {"tmp": tmp, "exe": exe, "csr": path, "dim": dim, "ref": {}}


def _run(rig, nranks, args, tag, csr=None):
    uid = os.path.join(rig["tmp"], "uid_%s.bin" % tag)
    dump = os.path.join(rig["tmp"], "dump_%s" % tag)
    env = dict(os.environ, QBH_RCCL_LIB=STUB, TMPDIR=rig["tmp"], HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([rig["exe"], csr or rig["csr"], str(r), str(nranks), uid] + args + ["dump=" + dump], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, env=env) for r in range(nranks)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for p2 in procs:
                p2.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    res = []
    for r in range(nranks):
        line = [ln for ln in outs[r].splitlines() if ln.startswith("OK ")]
        assert len(line) == 1, outs[r]
        tok = line[0].split()
        raw = np.fromfile("%s.%d.bin" % (dump, r), dtype=np.uint8)
        m, mcg = np.frombuffer(raw[:16].tobytes(), dtype=np.int64)
        e0 = np.frombuffer(raw[16:24].tobytes(), dtype=np.float64)[0]
        ab = np.frombuffer(raw[24:24 + 16 * m].tobytes(), dtype=np.float64)
        vec = np.frombuffer(raw[24 + 16 * m:].tobytes(), dtype=np.complex128)
        res.append({"r0": int(tok[3]), "r1": int(tok[4]), "m": int(m), "mcg": int(mcg), "E0": float(e0), "a": ab[:m], "b": ab[m:], "vec": vec,
                    "accu": float(tok[8]), "nrm": float(tok[9]), "kron": int(tok[11]), "parts": int(tok[13]), "cols16": int(tok[15]), "wire": int(tok[17]), "need": float(tok[19]), "sparse": int(tok[21])})
    return res


def _reference(rig, plain):
    """the ONE-rank answer of the same program (same format), also through the stub"""
    key = "plain" if plain else "default"
    if key not in rig["ref"]:
        rig["ref"][key] = _run(rig, 1, ["plain=1"] if plain else [], "ref_" + key)[0]
    return rig["ref"][key]


def _check(rig, res, ref, nranks, dim=None, e0=-14.076058658879278, kmax=40):
    assert res[0]["r0"] == 0 and res[-1]["r1"] == (dim or rig["dim"]) and all(res[i]["r1"] == res[i + 1]["r0"] for i in range(nranks - 1))
    assert abs(ref["E0"] - e0) < 1e-9                                 # SURVEY App. E (the rig's operator) / dense diagonalisation (others)
    full = np.concatenate([r["vec"] for r in res])
    for r in res:
        assert abs(r["E0"] - ref["E0"]) <= 1e-12 * abs(ref["E0"])
        assert abs(r["m"] - ref["m"]) <= 1 and abs(r["mcg"] - ref["mcg"]) <= 2
        k = min(r["m"], ref["m"], kmax)                               # beyond ~40 steps rounding differences are amplified by the recurrence
        assert np.allclose(r["a"][:k], ref["a"][:k], rtol=0, atol=1e-9) and np.allclose(r["b"][:k], ref["b"][:k], rtol=0, atol=1e-9)
        assert r["accu"] < 2e-12 and abs(r["nrm"] - 1.0) < 1e-10
        assert all(np.array_equal(r["a"], res[0]["a"]) for r in res)  # the scalars are all-reduced: every rank holds the same bits
    assert abs(abs(np.vdot(full, ref["vec"])) - 1.0) < 1e-8           # the slices ARE the one-rank eigenvector


# a covering set of (ranks, parts, wire, personalised): every pair of values of two different options meets at least once (the full
# 36-case product took 90 s of the GPU tier for one operator shape; the shapes below buy more)
@pytest.mark.parametrize("nranks,parts,realwire,sparse", [(2, 1, 1, 1), (2, 4, 0, 1), (2, 7, 1, 0), (2, 4, 0, 0),
                                                          (3, 1, 0, 1), (3, 4, 1, 1), (3, 7, 0, 0), (3, 1, 1, 0),
                                                          (4, 1, 0, 0), (4, 4, 1, 0), (4, 7, 1, 1), (4, 7, 0, 1)])
def test_split_shards_exchange_tiled_blocks_over_the_native_communicator(rig, nranks, parts, realwire, sparse):
    """complex128 shards of whole major indices, each split in place with 2-byte columns in BOTH parts (the one-GPU kernel: the far
    columns index the tiled order of the whole vector, the gathered blocks are moved there piece by piece), the TILED block of
    every rank on the wire -- as 8-byte real parts (qbh_opts.real_wire: real operator, real Lanczos / CG vectors) or as complex128
    elements, whole (all-gather) or only what each peer reads (qbh_opts.sparse_gather: the personalised exchange) -- in 1 / 4 / 7 band ranges; 2 ranks: uniform blocks, 3 and 4 ranks: ragged (70 major indices)."""
    res = _run(rig, nranks, ["plain=1", "kron=%d" % S_MINOR, "parts=%d" % parts, "realwire=%d" % realwire, "sparse=%d" % sparse] + (["uniform"] if nranks == 2 else []),
               "kron_%d_%d_%d_%d" % (nranks, parts, realwire, sparse))
    assert all(r["kron"] == S_MINOR and r["parts"] == parts and r["cols16"] == 3 and r["wire"] == (8 if realwire else 16) for r in res), res
    # only the major indices a shard's far / cross entries read are moved into its tiled x: on the 4 x 2 lattice every up configuration
    # hops into every rank's range, so the share stays high here (C3 with 8 ranks: 0.42-0.68, tools/needed_columns.py)
    assert all(0.3 < r["need"] <= 1.0 for r in res), [r["need"] for r in res]
    # sparse = 1: the exchange is personalised -- every rank sends each peer only the major indices that peer reads (packed per
    # destination, same band ranges), the receiver moves them to their place; sparse = 0: whole blocks to everybody
    assert all(r["sparse"] == sparse for r in res), res
    _check(rig, res, _reference(rig, True), nranks)


# (lx, ly, n_up, n_dn): minor sizes 56 (whole bands only: no cross part), 28 (last band of 4), 6 (narrower than one band: stays unsplit, plain exchange), 126 with 126 major indices, and 6 / 8 major indices over 3 / 4 / 5 ranks (ranks that own ONE major index).  Not more than 5 rank processes
# on the one GPU: from 6 on their queues outnumber the hardware's and every stream synchronisation of the stand-in waits for a
# scheduling quantum (the same 8-major case: 1 s with 5 ranks, 28-45 s with 6, 15 s with 7)
OTHER_SHAPES = {"s56": (4, 2, 4, 3), "s28": (4, 2, 5, 2), "s6": (3, 2, 3, 1), "s126": (3, 3, 4, 4), "few_majors": (3, 2, 1, 3), "few8": (4, 2, 1, 4)}


@pytest.fixture(scope="module")
def other_ops(rig):
    ops = {}
    for name, (lx, ly, nu, nd) in OTHER_SHAPES.items():
        n = lx * ly
        G = q.csr_mat.hubbard(n, nu, nd, lattices.square(lx, ly), t=1.0, U=1.1, opts=q.make_opts(kron_split=0, value_dict=0, real_fast_path=0))
        ia, ja, val = G.download()
        dim, S = G.dim, math.comb(n, nd)
        G.destroy()
        path = os.path.join(rig["tmp"], "csr_%s.bin" % name)
        with open(path, "wb") as f:
            np.array([dim, len(ja), 0], dtype=np.int64).tofile(f)
            ia.astype(np.int64).tofile(f), ja.astype(np.int64).tofile(f), val.tofile(f)
        import scipy.sparse as sp
        H = sp.csr_matrix((val, ja, ia), shape=(dim, dim))
        e0 = float(np.linalg.eigvalsh(H.toarray())[0]) if dim <= 4000 else float(sp.linalg.eigsh(H, k=1, which="SA", tol=1e-13)[0][0])
        ops[name] = {"csr": path, "dim": dim, "S": S, "e0": e0, "ref": None}
    return ops


@pytest.mark.parametrize("shape,nranks,parts,realwire,sparse", [
    ("s56", 2, 4, 1, 1), ("s56", 3, 1, 0, 1), ("s56", 4, 7, 1, 0),
    ("s28", 2, 1, 0, 0), ("s28", 3, 4, 1, 1), ("s28", 4, 4, 0, 1),
    ("s6", 2, 4, 1, 1), ("s6", 3, 1, 1, 0), ("s6", 4, 2, 0, 1),
    ("s126", 3, 4, 1, 1), ("s126", 4, 7, 0, 0),
    ("few_majors", 3, 4, 1, 1), ("few_majors", 4, 1, 0, 1), ("few8", 5, 1, 0, 1)])
def test_other_shapes_of_split_shards_over_the_native_communicator(rig, other_ops, shape, nranks, parts, realwire, sparse):
    """The exchange forms of the sharded headline on operators whose minor size is a whole number of bands (no cross part), ends in a
    band of 4, is narrower than a band, and on shards of one or two major indices: every rank must report the one-rank run's E0 (which
    must be the dense ground-state energy), its a_j / b_j and its slice of the eigenvector."""
    op = other_ops[shape]
    if op["ref"] is None:
        op["ref"] = _run(rig, 1, ["plain=1"], "oref_" + shape, csr=op["csr"])[0]
    res = _run(rig, nranks, ["plain=1", "kron=%d" % op["S"], "parts=%d" % parts, "realwire=%d" % realwire, "sparse=%d" % sparse],
               "o_%s_%d_%d_%d_%d" % (shape, nranks, parts, realwire, sparse), csr=op["csr"])
    if shape == "s6":               # narrower than one band: the library does not split such an operator -- every rank on the plain exchange
        assert all(r["kron"] == 0 and r["parts"] == 1 for r in res), [(r["kron"], r["parts"], r["wire"]) for r in res]
    else:
        assert all(r["kron"] == op["S"] and r["wire"] == (8 if realwire else 16) for r in res), [(r["kron"], r["parts"], r["wire"]) for r in res]
    _check(rig, res, op["ref"], nranks, dim=op["dim"], e0=op["e0"], kmax=20 if op["dim"] < 1000 else 40)     # (dim 120: the Krylov space is exhausted early)


@pytest.mark.parametrize("shape,nranks", [("s126", 3), ("s56", 2)])
def test_static_walks_under_ranks_with_workgroups_left_out(rig, other_ops, shape, nranks):
    """qbh_opts.deterministic = 1 (static walks: every workgroup's blocks follow from its index and the grid size) on split shards under a
    communicator, whose persistent passes run on REDUCED grids (qbh_opts.comm_reserve): the grids must stay multiples of 8 or blocks
    are visited twice.  Same E0, coefficients and eigenvector as the one-rank run."""
    op = other_ops[shape]
    if op["ref"] is None:
        op["ref"] = _run(rig, 1, ["plain=1"], "oref_" + shape, csr=op["csr"])[0]
    res = _run(rig, nranks, ["plain=1", "kron=%d" % op["S"], "parts=4", "det=1"], "det_%s_%d" % (shape, nranks), csr=op["csr"])
    assert all(r["kron"] == op["S"] for r in res), [(r["kron"], r["parts"]) for r in res]
    _check(rig, res, op["ref"], nranks, dim=op["dim"], e0=op["e0"])


@pytest.mark.parametrize("nranks,unsplit", [(2, 1), (3, 0)])
def test_one_unsplit_rank_makes_every_rank_fall_back_together(rig, nranks, unsplit):
    """One rank keeps its shard unsplit: the collective qbh_csr_set_comm must end with EVERY rank on the plain exchange (the
    split ones merged back into a CSR), not with mismatched message sizes."""
    res = _run(rig, nranks, ["plain=1", "kron=%d" % S_MINOR, "parts=4", "unsplit=%d" % unsplit], "mixed_%d" % nranks)
    assert all(r["kron"] == 0 and r["parts"] == 1 for r in res), res
    _check(rig, res, _reference(rig, True), nranks)


@pytest.mark.parametrize("nranks", [2, 3, 4])
@pytest.mark.parametrize("cuts", ["uniform", "ragged"])
@pytest.mark.parametrize("plain", [0, 1])
def test_plain_shards_over_the_native_communicator(rig, nranks, cuts, plain):
    """Unsplit row shards (locally-owned / remote column split): the library's default format -- coded values, REAL wire format,
    8 bytes per element through d_xfull_r -- and the complex128 format; ncclAllGather for uniform blocks (4900 / 3 is not whole:
    falls to the ragged path by itself) and the grouped send / receive all-gather-v for nnz-balanced cuts."""
    if cuts == "uniform" and rig["dim"] % nranks:
        pytest.skip("4900 rows do not divide by %d" % nranks)
    res = _run(rig, nranks, (["plain=1"] if plain else []) + (["uniform"] if cuts == "uniform" else []), "plain_%d_%s_%d" % (nranks, cuts, plain))
    _check(rig, res, _reference(rig, bool(plain)), nranks)


def test_bench_two_ranks_on_one_gpu_through_the_native_communicator(rig):
    """bench.py --gpus 2 as the driver launches it, two ranks sharing the GPU (gloo for torch's own rendezvous, the library's
    exchange through qbh_comm_create_rccl on the stub): the N > 1 line's schema -- per_rank timing, ms_gather, hidden fraction --
    is populated and E0 equals the one-rank value."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, QBH_RCCL_LIB=STUB, QBH_DIST_BACKEND="gloo", TMPDIR=rig["tmp"], HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "5", "--warmup", "2", "--workload", "hubbard_4x3_half", "--no-cpu-baseline", "--no-matrix-free", "--no-fast-path", "--processes", "1"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert one.returncode == 0, one.stdout + one.stderr
    ref = json.loads([ln for ln in one.stdout.strip().splitlines() if ln.startswith("{")][-1])
    many = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, capture_output=True, text=True,
                          env=env, cwd=ROOT, timeout=900)
    assert many.returncode == 0, many.stdout + many.stderr
    got = json.loads([ln for ln in many.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert got["n_gpus"] == 2 and "native RCCL" in got["config"]["exchange"], got["config"]
    assert abs(got["e0"] - ref["e0"]) <= 1e-10 * abs(ref["e0"])
    pr = got["per_rank"]
    assert len(pr) == 2 and all(set(("rank", "rows", "nnz", "ms_spmv", "ms_gather", "gather_hidden_frac", "roofline_frac")) <= set(p) for p in pr), pr
    assert all(p["ms_spmv"] > 0 and p["rows"] > 0 for p in pr) and sum(p["rows"] for p in pr) == got["config"]["dim"], pr
    # qbh_opts.comm_reserve: a few untimed steps with 64 / 128 / 32 workgroups left out of the persistent passes, the ranks agree on one
    # (split shards only -- this 12-site operator is below the size the library splits by itself: the field is there and empty; C3 through
    # the same path: profiles/r6_bench/ranks_stub/*_abi601.json)
    cal = got["exchange"]["comm_reserve_calibration"]
    if got["config"].get("kron_split"):
        assert set(cal["ms_per_step_by_workgroups_left_out"]) == {"64", "128", "32"} and cal["chosen"] in (64, 128, 32)
        assert got["exchange"]["comm_reserve_workgroups"] == cal["chosen"]
    else:
        assert cal is None and got["exchange"]["comm_reserve_workgroups"] is None


@pytest.mark.parametrize("kron", [0, 1])
@pytest.mark.parametrize("nranks", [2, 3])
def test_pipelined_lanczos_loop_under_a_communicator_is_the_unpipelined_one(rig, nranks, kron):
    """qbh_opts.lanczos_pipeline under a communicator: <u, w> and |w'|^2 are all-reduced in stream order between the same launches
    and step m + 1 goes out before step m has been read back; m, a_j, b_j on every rank are bit for bit those of the loop that
    synchronises with the host twice per step (the stand-in sums in rank order, as every deterministic all-reduce does)."""
    extra = ["plain=1"] + (["kron=%d" % S_MINOR, "parts=4"] if kron else [])
    p1 = _run(rig, nranks, extra + ["pipeline=1"], "pl1_%d_%d" % (nranks, kron))
    p0 = _run(rig, nranks, extra + ["pipeline=0"], "pl0_%d_%d" % (nranks, kron))
    for a, b in zip(p1, p0):
        assert a["m"] == b["m"] and np.array_equal(a["a"], b["a"]) and np.array_equal(a["b"], b["b"]) and a["E0"] == b["E0"]
    _check(rig, p1, _reference(rig, True), nranks)


@pytest.mark.parametrize("kron", [0, 1])
def test_sharded_lanczos_run_interrupted_and_resumed_from_rank_checkpoints(rig, kron):
    """qbh_lanczos_ckpt on row shards (collective): two ranks, stopped after 20 steps -- every rank's slice sits in
    <dir>/shard<r>of2/ under the reference's own file names (src/ckpt.cc:178-297) with shard-local lengths -- then two NEW
    processes resume from the files and converge to the one-rank E0; the first 20 coefficients come from disk.  Then a torn
    update on ONE rank (first marker, new data complete, no second marker -- the crash window between the ranks' commits):
    the ranks settle on the new step together.  Plain shards and split shards (tiled exchange, real wire)."""
    import shutil
    import struct
    ck = os.path.join(rig["tmp"], "ck_shards_%d" % kron)
    shutil.rmtree(ck, ignore_errors=True)
    extra = ["plain=1"] + (["kron=%d" % S_MINOR, "uniform"] if kron else [])
    ref = _reference(rig, True)
    r1 = _run(rig, 2, extra + ["ckpt=" + ck, "every=7", "maxsteps=20"], "ck1_%d" % kron)
    assert all(r["m"] == 20 and r["mcg"] == 0 for r in r1)                               # (mcg slot: converged flag)
    for r, res in enumerate(r1):
        d = os.path.join(ck, "shard%dof2" % r)
        assert sorted(os.listdir(d)) == ["HessenbergA.dat", "HessenbergB.dat", "lanczosV19.dat", "lanczosV20.dat", "lczs_mlns.dat"]
        n = res["r1"] - res["r0"]
        assert os.path.getsize(os.path.join(d, "lanczosV20.dat")) == 8 + 16 * n + 4      # the rank's slice, vec_disk_write format
        assert struct.unpack("<q", open(os.path.join(d, "lanczosV20.dat"), "rb").read(8))[0] == n
        assert os.path.getsize(os.path.join(d, "HessenbergA.dat")) == 8 + 8 * 20 + 4
    assert np.array_equal(r1[0]["a"], r1[1]["a"])
    # the crash window: rank 1 is put back into "new data written, second marker missing" for a step-21 update while rank 0 holds 21 committed
    r21 = _run(rig, 2, extra + ["ckpt=" + ck, "every=1", "maxsteps=1"], "ck21_%d" % kron)
    assert all(r["m"] == 21 for r in r21)
    d1 = os.path.join(ck, "shard1of2")
    for nm in ("HessenbergA.dat", "HessenbergB.dat", "lczs_mlns.dat"):
        os.rename(os.path.join(d1, nm), os.path.join(d1, nm + ".new"))
    open(os.path.join(d1, "lczs_updt.Qckpt1"), "wb").write(struct.pack("<q", 21))
    r2 = _run(rig, 2, extra + ["ckpt=" + ck, "every=50"], "ck2_%d" % kron)
    assert all(r["mcg"] == 1 for r in r2)                                                  # converged
    for r in r2:
        assert abs(r["m"] - ref["m"]) <= 1 and abs(r["E0"] - ref["E0"]) <= 1e-11 * abs(ref["E0"])
        assert np.array_equal(r["a"][:20], r1[0]["a"][:20]) and np.array_equal(r["b"][:20], r1[0]["b"][:20])      # from disk, bit for bit
        assert np.array_equal(r["a"][:21], r21[0]["a"][:21])


def test_bench_starts_its_own_ranks_when_no_launcher_did(rig):
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the way the driver runs N = 1): the parent starts the rank processes itself
    before it touches the GPU, relays rank 0's line and the ranks' exit code; --gpus 1 is unchanged."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(QBH_RCCL_LIB=STUB, QBH_DIST_BACKEND="gloo", TMPDIR=rig["tmp"], HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "4", "--warmup", "2", "--workload", "hubbard_4x3_half", "--no-cpu-baseline", "--no-matrix-free", "--no-fast-path", "--processes", "1", "--force-split"]
    many = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert many.returncode == 0, many.stdout + many.stderr
    lines = [ln for ln in many.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, many.stdout                                    # ONE JSON line, rank 0's
    got = json.loads(lines[0])
    assert got["n_gpus"] == 2 and got["steps"] == 4 and len(got["per_rank"]) == 2 and "native RCCL" in got["config"]["exchange"], got
    assert abs(got["e0"] + 16.879382788684) < 1e-9
    # split shards (--force-split): the comm_reserve calibration ran and the ranks agreed on one value
    cal = got["exchange"]["comm_reserve_calibration"]
    assert got["config"]["kron_split"] and set(cal["ms_per_step_by_workgroups_left_out"]) == {"64", "128", "32"}
    assert cal["chosen"] in (64, 128, 32) and got["exchange"]["comm_reserve_workgroups"] == cal["chosen"]
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "hubbard_4x3_half", "--packed-real"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert bad.returncode != 0                                             # the ranks' failure is the parent's exit code


@pytest.mark.parametrize("nranks", [2, 3])
def test_python_hosts_keep_their_split_shards_under_the_native_communicator(rig, nranks):
    """The Python route of bench.py (quantum_basis_amd.dist.NativeComm on a torch stream, torch's own rendezvous over gloo):
    split shards must STAY split after the collective attach, with the default 4 gather parts, and give the one-rank E0.
    (Round 5's first C3 run through the stand-in fell back to the plain exchange now and then: qbh_comm_create_rccl zeroed its
    buffers with hipMemset on the null stream, which is not ordered with a non-blocking host stream -- the zeroing could land
    after the agreement's own data.  tools/r5/parts_probe.py is the probe that showed it.)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, QBH_RCCL_LIB=STUB, TMPDIR=rig["tmp"], HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "r5", "parts_probe.py"), "hubbard_4x3_half", "0", "2"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    import re
    after = re.findall(r"rank \d+ after : kron_minor \d+ gather_parts \d+", p.stdout)
    e0 = [float(t) for t in re.findall(r"rank \d+ E0 (-?[0-9.]+) steps", p.stdout)]      # (the ranks' lines can run into one another)
    assert len(after) == nranks and all("kron_minor 924 gather_parts 4" in ln for ln in after), p.stdout
    assert len(e0) == nranks and all(abs(e + 16.879382788684) < 1e-9 for e in e0), p.stdout


def test_headline_operator_on_four_ranks_through_the_native_communicator(rig):
    """C3 itself (dim 165,636,900) as four ragged row shards of whole major indices on this one GPU, each split in place with 2-byte
    near columns, the tiled blocks exchanged through qbh_comm_create_rccl in 4 gather parts: the first Lanczos coefficients must be
    the one-rank values on every rank.  This is the run that found two bugs no small test had seen (round 5): buffers zeroed on the
    null stream, and the 2-byte columns of the up-to-7 entries a near block loads in front of its first one being decoded with the
    wrong base -- a read past the end of x on the ranks whose vectors ended at an allocation boundary."""
    import re
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, QBH_RCCL_LIB=STUB, TMPDIR=rig["tmp"], HSA_ENABLE_IPC_MODE_LEGACY="0", PROBE_MAXIT="6")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "r5", "parts_probe.py"), "hubbard_4x4_half", "0", "1"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    after = re.findall(r"rank \d+ after : kron_minor (\d+) gather_parts (\d+)", p.stdout)
    ran = re.findall(r"rank \d+ ran (\d+) steps, a0 (-?[0-9.]+) b1 (-?[0-9.]+)", p.stdout)
    assert len(after) == 4 and all(a == ("12870", "4") for a in after), p.stdout
    assert len(ran) == 4 and all(r[0] == "5" for r in ran), p.stdout
    # the one-rank coefficients of the same start vector (seed 1): tests/test_gpu_kron.py pins them through E0; here to 1e-9
    assert all(abs(float(r[1]) - 4.399209342849) < 1e-9 and abs(float(r[2]) - 5.951874544884) < 1e-9 for r in ran), ran


@pytest.mark.parametrize("model", ["", "8:rccl"])
def test_solo_rank_timing_model_runs(rig, model):
    """tools/solo_rank.py: one rank of two alone on the GPU, its peer modelled by the stand-in's solo mode -- the hold of the side stream as a
    host function or (QBH_STUB_SOLO_KERNEL=W:rccl) as a kernel with the register / LDS footprint of RCCL's own, which is what showed that the
    persistent passes must leave room (qbh_opts.comm_reserve; profiles/r6_bench/solo_rank/occupancy_SUMMARY.txt).  The numbers of a 12-site
    operator mean nothing; the tool and both hold paths must run and report the fields the summaries are made of."""
    env = dict(os.environ, QBH_RCCL_LIB=STUB, QBH_STUB_SOLO="50", TMPDIR=rig["tmp"], PYTHONPATH=ROOT)
    if model:
        env["QBH_STUB_SOLO_KERNEL"] = model
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "solo_rank.py"), "hubbard_4x3_half", "2", "0", "steps=6", "warmup=2", "parts=4"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["ranks"] == 2 and d["rank"] == 0 and d["steps"] == 6 and d["ms_per_step"] > 0 and d["comm_reserve"] == 0
    assert d["link_model"]["GBps_per_link"] == 50.0
