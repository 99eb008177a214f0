"""The C++ adaptor header (include/qbhip_qbasis.hpp) compiles with plain g++, links against
libqbhip.so, and (on the GPU box) reproduces the oracle's numbers from a C++ host program."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import helpers
from oracle import qb_oracle as qo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp, name="adaptor_main"):
    exe = os.path.join(tmp, name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx", name + ".cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "quantum_basis_amd"), "-lqbhip",
                           "-Wl,-rpath," + os.path.join(ROOT, "quantum_basis_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _dump(tmp, name):
    d, ia, ja, val, sym = helpers.case(name)
    x = qo.vec_randomize(d, 1)
    path = os.path.join(tmp, "csr.bin")
    with open(path, "wb") as f:
        np.array([d, len(ja), int(sym)], dtype=np.int64).tofile(f)
        ia.tofile(f), ja.tofile(f), val.tofile(f), x.tofile(f)
    return path, qo.Csr(d, ia, ja, val, sym), x


def test_adaptor_compiles_links_and_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    with tempfile.TemporaryDirectory() as tmp:
        exe = _build(tmp)
        path, _, _ = _dump(tmp, "chain12_sz0")
        p = subprocess.run([exe, path], capture_output=True, text=True)
        assert p.returncode == 3 and "no HIP device" in p.stdout     # std::runtime_error, no CPU fallback


@pytest.mark.gpu
def test_adaptor_cxx_host_program_matches_oracle():
    with tempfile.TemporaryDirectory() as tmp:
        exe = _build(tmp)
        path, O, x = _dump(tmp, "hubbard_4x2")
        p = subprocess.run([exe, path], capture_output=True, text=True)
        assert p.returncode == 0, p.stdout + p.stderr
        tok = p.stdout.splitlines()[0].split()
        assert tok[0] == "OK"
        sr, si, sr2, m, E0, mcg, accu, d = [float(t) for t in tok[1:]]
        y = O.multmv(x)
        assert abs(sr - y.sum().real) < 1e-11 and abs(si - y.sum().imag) < 1e-11
        assert abs(sr2 - 2 * y.sum().real) < 1e-11          # MultMv2 accumulated onto MultMv's result
        assert abs(m - 82) <= 1 and abs(E0 + 14.076058658879278) < 1e-9
        assert abs(mcg - 83) <= 2 and accu < 2e-12 and d < 1e-9
        drv = p.stdout.splitlines()[1].split()
        assert drv[0] == "DRV"
        e0, e1, s_e0, s_v0, nconv_l, w0, w1, w2, nconv_i, ov = [float(t) for t in drv[1:]]
        dense = np.linalg.eigvalsh(O.to_dense())
        assert abs(e0 - dense[0]) < 1e-9 and abs(e1 - dense[1]) < 1e-7 and nconv_l == 1
        assert abs(s_e0 - 82) <= 1 and abs(s_v0 - 83) <= 2
        assert nconv_i == 3 and np.allclose([w0, w1, w2], dense[:3], atol=1e-9)
        assert abs(ov - 1.0) < 1e-8


def test_sharded_host_program_compiles_and_links():
    with tempfile.TemporaryDirectory() as tmp:
        assert os.path.exists(_build(tmp, "sharded_main"))


@pytest.mark.gpu
@pytest.mark.parametrize("force_ragged", [0, 1])
def test_sharded_cxx_host_rank_over_native_rccl(force_ragged):
    """A C++ host process as one rank of the row-sharded run: qbh_balanced_row_cuts -> qbh_csr_create_rows ->
    qbh_comm_create_rccl (1 rank on this one-GPU box; QBH_DEBUG=force_ragged=1 takes the send/recv all-gather-v of
    ragged partitions) -> qbh_lanczos_dev / qbh_eigenvec_cg_dev.  No Python anywhere in the SpMV loop."""
    with tempfile.TemporaryDirectory() as tmp:
        exe = _build(tmp, "sharded_main")
        path, O, x = _dump(tmp, "hubbard_4x2")
        env = dict(os.environ, QBH_DEBUG="force_ragged=%d" % force_ragged)
        p = subprocess.run([exe, path, "0", "1", os.path.join(tmp, "uid.bin")], capture_output=True, text=True, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("OK ")]      # RCCL prints a version banner first
        assert len(line) == 1, p.stdout
        tok = line[0].split()
        assert tok[0] == "OK" and tok[1:5] == ["0", "1", "0", "4900"]
        m, E0, mcg, accu, nrm = int(tok[5]), float(tok[6]), int(tok[7]), float(tok[8]), float(tok[9])
        assert abs(m - 82) <= 1 and abs(E0 + 14.076058658879278) < 1e-9
        assert abs(mcg - 83) <= 2 and accu < 2e-12 and abs(nrm - 1.0) < 1e-10


def _visible_gpus():
    import torch
    return torch.cuda.device_count()          # does not initialise the GPU on this image


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 4])
@pytest.mark.parametrize("cuts", ["ragged", "uniform"])
def test_sharded_cxx_host_ranks_over_native_rccl_on_real_gpus(nranks, cuts):
    """SURVEY 8(e) on hardware: N C++ host processes, one per GPU, row blocks of the same host CSR, the native RCCL
    communicator over xGMI (ncclAllGather for uniform blocks, grouped ncclSend/ncclRecv for nnz-balanced ones) and the
    all-reduce of the Lanczos scalars.  Skipped -- visibly -- on a box with fewer GPUs; every rank must report the
    single-GPU answer."""
    if _visible_gpus() < nranks:
        pytest.skip("needs %d GPUs, %d visible" % (nranks, _visible_gpus()))
    with tempfile.TemporaryDirectory() as tmp:
        exe = _build(tmp, "sharded_main")
        path, O, x = _dump(tmp, "hubbard_4x2")
        uid = os.path.join(tmp, "uid.bin")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([exe, path, str(r), str(nranks), uid] + (["uniform"] if cuts == "uniform" else []),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(nranks)]
        outs = []
        for p in procs:
            try:
                out, _ = p.communicate(timeout=300)
            except subprocess.TimeoutExpired:
                for q_ in procs:
                    q_.kill()
                raise
            outs.append(out)
            assert p.returncode == 0, out
        covered = []
        for r, out in enumerate(outs):
            line = [ln for ln in out.splitlines() if ln.startswith("OK ")]
            assert len(line) == 1, out
            tok = line[0].split()
            assert tok[1:3] == [str(r), str(nranks)]
            covered.append((int(tok[3]), int(tok[4])))
            m, E0, mcg, accu, nrm = int(tok[5]), float(tok[6]), int(tok[7]), float(tok[8]), float(tok[9])
            assert abs(m - 82) <= 1 and abs(E0 + 14.076058658879278) < 1e-9          # SURVEY App. E, every rank
            assert abs(mcg - 83) <= 2 and accu < 2e-12 and abs(nrm - 1.0) < 1e-10
        assert covered[0][0] == 0 and covered[-1][1] == 4900 and all(covered[i][1] == covered[i + 1][0] for i in range(nranks - 1))
        if cuts == "uniform":
            assert all(b - a == 4900 // nranks for a, b in covered)


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 4])
def test_bench_under_torch_distributed_run_uses_the_native_communicator(nranks):
    """The driver's own launch line (python -m torch.distributed.run ... bench.py --gpus N) on a multi-GPU box: the JSON
    line must say the exchange ran on the native RCCL communicator and E0 must equal the one-rank value."""
    if _visible_gpus() < nranks:
        pytest.skip("needs %d GPUs, %d visible" % (nranks, _visible_gpus()))
    import json
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "5", "--warmup", "2", "--workload", "hubbard_4x3_half", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True,
                         env=env, cwd=ROOT, timeout=600)
    assert one.returncode == 0, one.stdout + one.stderr
    ref = json.loads(one.stdout.strip().splitlines()[-1])
    many = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
                           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                           "--gpus", str(nranks)] + common, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert many.returncode == 0, many.stdout + many.stderr
    got = json.loads([ln for ln in many.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert got["n_gpus"] == nranks and "native RCCL" in got["config"]["exchange"], got["config"]
    assert abs(got["e0"] - ref["e0"]) <= 1e-10 * abs(ref["e0"])


@pytest.mark.gpu
def test_cxx_program_following_the_reference_example_reproduces_its_asserted_sector_energies():
    """tests/cxx/sectors_main.cpp = examples/trans_symmetric/latt_square/square_Fermi_Hubbard.cc through the C++ header:
    eight momentum sectors assembled on the device, locate_E0_lanczos in each, the reference's asserts (:146-153)."""
    with tempfile.TemporaryDirectory() as tmp:
        exe = _build(tmp, "sectors_main")
        p = subprocess.run([exe], capture_output=True, text=True)
        assert p.returncode == 0, p.stdout + p.stderr
        last = [ln for ln in p.stdout.splitlines() if ln.startswith("OK ")][-1]
        assert float(last.split()[1]) < 1e-8
        assert p.stdout.count("k=(") == 8
