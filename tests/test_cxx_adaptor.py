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
    qbh_comm_create_rccl (1 rank on this one-GPU box; QBH_COMM_FORCE_RAGGED=1 takes the grouped-broadcast gather of
    ragged partitions) -> qbh_lanczos_dev / qbh_eigenvec_cg_dev.  No Python anywhere in the SpMV loop."""
    with tempfile.TemporaryDirectory() as tmp:
        exe = _build(tmp, "sharded_main")
        path, O, x = _dump(tmp, "hubbard_4x2")
        env = dict(os.environ, QBH_COMM_FORCE_RAGGED=str(force_ragged))
        p = subprocess.run([exe, path, "0", "1", os.path.join(tmp, "uid.bin")], capture_output=True, text=True, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("OK ")]      # RCCL prints a version banner first
        assert len(line) == 1, p.stdout
        tok = line[0].split()
        assert tok[0] == "OK" and tok[1:5] == ["0", "1", "0", "4900"]
        m, E0, mcg, accu, nrm = int(tok[5]), float(tok[6]), int(tok[7]), float(tok[8]), float(tok[9])
        assert abs(m - 82) <= 1 and abs(E0 + 14.076058658879278) < 1e-9
        assert abs(mcg - 83) <= 2 and accu < 2e-12 and abs(nrm - 1.0) < 1e-10


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
