"""tests/stub_rccl/librccl_stub.so is what lets N ranks on ONE GPU run the library's native communicator
(qbh_comm_create_rccl, tests/test_gpu_native_ranks.py).  A checker has to be checked: here its own protocol runs in the build
container on host buffers (the stub copies with memcpy when no HIP device is visible) -- all-gather, all-reduce, the grouped
send / receive all-gather-v with ragged blocks, and the failures it must report where real RCCL would hang or corrupt."""
import ctypes as C
import multiprocessing as mp
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_DIR = os.path.join(ROOT, "tests", "stub_rccl")
STUB = os.path.join(STUB_DIR, "librccl_stub.so")

NCCL_F64, NCCL_SUM = 8, 0          # ncclFloat64 / ncclSum of <rccl/rccl.h>


def _stub():
    if not os.path.exists(STUB):
        subprocess.check_call(["make", "-C", STUB_DIR])
    L = C.CDLL(STUB)
    L.ncclGetErrorString.restype = C.c_char_p
    return L


class _Uid(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _rank(rank, world, uid_bytes, scenario, q):
    try:
        L = _stub()
        uid = _Uid()
        C.memmove(C.byref(uid), uid_bytes, 128)
        comm = C.c_void_p()
        rc = L.ncclCommInitRank(C.byref(comm), world, uid, rank)
        assert rc == 0, L.ncclGetErrorString(rc)
        out = {}
        # all-gather of equal blocks
        mine = np.arange(5, dtype=np.float64) + 100.0 * rank
        full = np.zeros(5 * world)
        rc = L.ncclAllGather(mine.ctypes.data_as(C.c_void_p), full.ctypes.data_as(C.c_void_p), C.c_size_t(5), NCCL_F64, comm, None)
        out["allgather"] = (rc, full.copy())
        # all-reduce in place
        s = np.array([1.0 + rank, 10.0 * rank, -2.5])
        rc = L.ncclAllReduce(s.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), C.c_size_t(3), NCCL_F64, NCCL_SUM, comm, None)
        out["allreduce"] = (rc, s.copy())
        # all-gather-v: ragged blocks, one send + one receive per peer in one group (qbh_comm.cpp enqueue_gather)
        lens = [3 + 2 * r for r in range(world)]
        if scenario == "mismatch" and rank == 1:
            lens[0] += 1                                  # rank 1 expects one element more from rank 0 than rank 0 sends
        offs = np.concatenate([[0], np.cumsum(lens)])
        blk = np.full(lens[rank], float(rank + 1))
        recv = np.zeros(int(offs[-1]))
        assert L.ncclGroupStart() == 0
        for peer in range(world):
            if peer == rank:
                continue
            assert L.ncclRecv(C.c_void_p(recv.ctypes.data + 8 * int(offs[peer])), C.c_size_t(lens[peer]), NCCL_F64, peer, comm, None) == 0
            if not (scenario == "missing_send" and rank == 0 and peer == 1):
                assert L.ncclSend(blk.ctypes.data_as(C.c_void_p), C.c_size_t(lens[rank]), NCCL_F64, peer, comm, None) == 0
        rc = L.ncclGroupEnd()
        recv[int(offs[rank]):int(offs[rank + 1])] = blk
        out["gatherv"] = (rc, recv.copy(), L.ncclGetErrorString(rc).decode())
        L.ncclCommDestroy(comm)
        q.put((rank, out))
    except Exception as e:          # noqa: BLE001
        q.put((rank, {"error": repr(e)}))


def _run(world, scenario):
    L = _stub()
    uid = _Uid()
    with tempfile.TemporaryDirectory() as tmp:
        old = os.environ.get("TMPDIR")
        os.environ["TMPDIR"] = tmp
        try:
            assert L.ncclGetUniqueId(C.byref(uid)) == 0
            ctx = mp.get_context("spawn")
            q = ctx.Queue()
            ps = [ctx.Process(target=_rank, args=(r, world, bytes(uid), scenario, q)) for r in range(world)]
            for p in ps:
                p.start()
            res = dict(q.get(timeout=180) for _ in ps)
            for p in ps:
                p.join(timeout=30)
        finally:
            if old is None:
                os.environ.pop("TMPDIR", None)
            else:
                os.environ["TMPDIR"] = old
    return res


@pytest.mark.parametrize("world", [2, 3])
def test_stub_collectives_and_grouped_send_recv(world):
    res = _run(world, "ok")
    lens = [3 + 2 * r for r in range(world)]
    for r in range(world):
        out = res[r]
        assert "error" not in out, out
        rc, full = out["allgather"]
        assert rc == 0 and np.array_equal(full, np.concatenate([np.arange(5) + 100.0 * k for k in range(world)]))
        rc, s = out["allreduce"]
        assert rc == 0 and np.allclose(s, [sum(1.0 + k for k in range(world)), sum(10.0 * k for k in range(world)), -2.5 * world])
        rc, recv, err = out["gatherv"]
        assert rc == 0, err
        assert np.array_equal(recv, np.concatenate([np.full(lens[k], k + 1.0) for k in range(world)]))
    # every rank holds the same all-reduce bits (summed in rank order)
    assert all(np.array_equal(res[0]["allreduce"][1], res[r]["allreduce"][1]) for r in range(world))


@pytest.mark.parametrize("scenario,needle", [("mismatch", "bytes"), ("missing_send", "no send matches")])
def test_stub_reports_what_real_rccl_would_hang_on(scenario, needle):
    """A receive whose element count differs from the matched send, or a receive nobody sends to: RCCL hangs or corrupts;
    the stub makes EVERY rank's call fail (the advisor's round-4 finding on diverging gather-part counts is exactly this)."""
    res = _run(2, scenario)
    rcs = [res[r]["gatherv"][0] for r in range(2)]
    assert all(rc != 0 for rc in rcs), res
    assert any(needle in res[r]["gatherv"][2] for r in range(2)), [res[r]["gatherv"][2] for r in range(2)]
