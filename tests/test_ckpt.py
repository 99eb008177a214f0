"""Checkpoint files in the reference's on-disk format (src/miscellaneous.cc:391-469, src/ckpt.cc:178-297)
and exact resume of the device Lanczos run."""
import os
import struct
import zlib

import numpy as np
import pytest

from quantum_basis_amd import ckpt


def test_vec_disk_format_is_the_references(tmp_path):
    x = (np.arange(70001) * 0.25 - 3j * np.arange(70001)).astype(np.complex128)     # > 1 MiB: several CRC chunks
    f = str(tmp_path / "lanczosV7.dat")
    assert ckpt.vec_disk_write(f, x) == 0
    raw = open(f, "rb").read()
    # int64 n | payload | CRC-32 over (n, payload); boost::crc_32_type == zlib's CRC-32
    assert len(raw) == 8 + 16 * x.size + 4
    assert struct.unpack("<q", raw[:8])[0] == x.size
    assert raw[8:-4] == x.tobytes()
    assert struct.unpack("<I", raw[-4:])[0] == zlib.crc32(raw[:-4]) & 0xFFFFFFFF
    assert np.array_equal(ckpt.vec_disk_read(f, x.size, np.complex128), x)
    # the reference returns 1 on: missing file, wrong size, wrong n, wrong checksum
    assert ckpt.vec_disk_read(str(tmp_path / "nope.dat"), x.size, np.complex128) is None
    assert ckpt.vec_disk_read(f, x.size - 1, np.complex128) is None
    bad = bytearray(raw)
    bad[100] ^= 0x40
    open(f, "wb").write(bytes(bad))
    assert ckpt.vec_disk_read(f, x.size, np.complex128) is None
    d = np.linspace(0, 1, 17)
    g = str(tmp_path / "HessenbergA.dat")
    ckpt.vec_disk_write(g, d)
    assert os.path.getsize(g) == 8 + 8 * 17 + 4 and np.array_equal(ckpt.vec_disk_read(g, 17, np.float64), d)


def test_checkpoint_directory_protocol(tmp_path):
    d = str(tmp_path / "out_Qckpt")
    dim, maxit = 50, 40
    rng = np.random.default_rng(0)
    hess = rng.normal(size=2 * maxit)
    v = (rng.normal(size=2 * dim) + 1j * rng.normal(size=2 * dim)).astype(np.complex128)
    st = dict(cnt_accuE0=3, accuracy=1.5e-7, theta0_prev=-4.25, theta1_prev=-3.5)
    assert ckpt.ckpt_lanczos_init(maxit, dim, "sr_val0", d) is None
    ckpt.ckpt_lanczos_update(6, maxit, dim, st, v, hess, "sr_val0", directory=d)
    ckpt.ckpt_lanczos_update(7, maxit, dim, st, v, hess, "sr_val0", directory=d)
    names = sorted(os.listdir(d))
    assert names == ["HessenbergA.dat", "HessenbergB.dat", "lanczosV6.dat", "lanczosV7.dat", "lczs_mlns.dat"]
    ck = ckpt.ckpt_lanczos_init(maxit, dim, "sr_val0", d)
    assert ck["k"] == 7 and ck["state"] == st
    assert np.array_equal(ck["hessenberg"][maxit:maxit + 7], hess[maxit:maxit + 7])
    assert np.array_equal(ck["hessenberg"][:8], hess[:8])
    assert np.array_equal(ck["v_pair"], v)
    # lczs_mlns.dat layout: int cnt, double accuracy, double theta0_prev, double theta1_prev (src/ckpt.cc:252-257)
    assert struct.unpack("<iddd", open(os.path.join(d, "lczs_mlns.dat"), "rb").read()) == (3, 1.5e-7, -4.25, -3.5)


def _snapshot(d, dim, maxit, seed, m):
    rng = np.random.default_rng(seed)
    hess = rng.normal(size=2 * maxit)
    v = (rng.normal(size=2 * dim) + 1j * rng.normal(size=2 * dim)).astype(np.complex128)
    st = dict(cnt_accuE0=seed, accuracy=1e-7 * seed, theta0_prev=-4.0 - seed, theta1_prev=-3.0 - seed)
    return hess, v, st


def test_torn_update_is_rewound_or_finished_like_the_reference(tmp_path):
    """src/ckpt.cc:40-100: marker Qckpt1 alone -> rewind one step (old data stays valid); Qckpt1 + Qckpt2 -> every new
    file was complete, finish the renames and the clean-up."""
    dim, maxit = 30, 40
    # --- interrupted BEFORE the second marker: the update to step 8 is discarded, step 7 is resumed
    d = str(tmp_path / "a")
    h6, v6, s6 = _snapshot(d, dim, maxit, 1, 6)
    ckpt.ckpt_lanczos_update(6, maxit, dim, s6, v6, h6, "sr_val0", directory=d)
    h7, v7, s7 = _snapshot(d, dim, maxit, 2, 7)
    ckpt.ckpt_lanczos_update(7, maxit, dim, s7, v7, h7, "sr_val0", directory=d)
    h8, v8, s8 = _snapshot(d, dim, maxit, 3, 8)
    open(os.path.join(d, "lczs_updt.Qckpt1"), "wb").write(struct.pack("<q", 8))
    ckpt.vec_disk_write(os.path.join(d, "HessenbergA.dat.new"), h8[maxit:maxit + 8])
    ckpt.vec_disk_write(os.path.join(d, "HessenbergB.dat.new"), h8[:9])
    ckpt.vec_disk_write(os.path.join(d, "lanczosV8.dat"), v8[:dim])
    open(os.path.join(d, "lczs_mlns.dat.new"), "wb").write(b"torn")           # half-written
    ck = ckpt.ckpt_lanczos_init(maxit, dim, "sr_val0", d)
    assert ck["k"] == 7 and ck["state"] == s7
    assert np.array_equal(ck["hessenberg"][maxit:maxit + 7], h7[maxit:maxit + 7]) and np.array_equal(ck["v_pair"], v7)
    assert sorted(os.listdir(d)) == ["HessenbergA.dat", "HessenbergB.dat", "lanczosV6.dat", "lanczosV7.dat", "lczs_mlns.dat"]
    # --- interrupted AFTER the second marker, in the middle of the clean-up: step 8 is complete and is used
    d = str(tmp_path / "b")
    ckpt.ckpt_lanczos_update(6, maxit, dim, s6, v6, h6, "sr_val0", directory=d)
    ckpt.ckpt_lanczos_update(7, maxit, dim, s7, v7, h7, "sr_val0", directory=d)
    v78 = np.concatenate([v8[:dim], v7[dim:]])                                # slot 0 = v[8], slot 1 = v[7]
    open(os.path.join(d, "lczs_updt.Qckpt1"), "wb").write(struct.pack("<q", 8))
    ckpt.vec_disk_write(os.path.join(d, "HessenbergA.dat.new"), h8[maxit:maxit + 8])
    ckpt.vec_disk_write(os.path.join(d, "HessenbergB.dat.new"), h8[:9])
    ckpt.vec_disk_write(os.path.join(d, "lanczosV8.dat"), v78[:dim])
    open(os.path.join(d, "lczs_mlns.dat.new"), "wb").write(struct.pack("<iddd", 3, s8["accuracy"], s8["theta0_prev"], s8["theta1_prev"]))
    open(os.path.join(d, "lczs_updt.Qckpt2"), "wb").write(struct.pack("<q", 8))
    os.remove(os.path.join(d, "HessenbergA.dat"))                             # the clean-up had started
    ck = ckpt.ckpt_lanczos_init(maxit, dim, "sr_val0", d)
    assert ck["k"] == 8 and ck["state"] == s8
    assert np.array_equal(ck["hessenberg"][:9], h8[:9]) and np.array_equal(ck["v_pair"], v78)
    assert sorted(os.listdir(d)) == ["HessenbergA.dat", "HessenbergB.dat", "lanczosV7.dat", "lanczosV8.dat", "lczs_mlns.dat"]


@pytest.mark.parametrize("every", [1, 2, 3])
@pytest.mark.parametrize("native", [False, True])
def test_torn_chunked_update_rewinds_to_the_committed_step(tmp_path, every, native):
    """Updates are `every` steps apart (the reference's are one).  A crash before the second marker of the update to step
    m_old + every must resume at m_old whatever the torn update had already written: for every = 2 that is the file
    lanczosV<m_old+1>, which made the consecutive run end one step past the committed Hessenberg data (advisor, round 2)."""
    dim, maxit, m_old = 24, 60, 10
    d = str(tmp_path / "ck")
    init = ckpt.native_ckpt_init if native else ckpt.ckpt_lanczos_init
    update = ckpt.native_ckpt_update if native else ckpt.ckpt_lanczos_update
    h0, v0, s0 = _snapshot(d, dim, maxit, 4, m_old)
    update(m_old, maxit, dim, s0, v0, h0, "sr_val0", directory=d)
    committed = {n: open(os.path.join(d, n), "rb").read() for n in os.listdir(d)}
    m = m_old + every
    h1, v1, s1 = _snapshot(d, dim, maxit, 5, m)
    # the torn update: first marker, the new Hessenberg files, both vectors under their final names, then the crash
    open(os.path.join(d, "lczs_updt.Qckpt1"), "wb").write(struct.pack("<q", m))
    ckpt.vec_disk_write(os.path.join(d, "HessenbergA.dat.new"), h1[maxit:maxit + m])
    ckpt.vec_disk_write(os.path.join(d, "HessenbergB.dat.new"), h1[:m + 1])
    if every > 1:              # with every == 1 V(m-1) IS the committed V(m_old): written via a temporary name, renamed
        ckpt.vec_disk_write(os.path.join(d, "lanczosV%d.dat" % (m - 1)), v1[((m - 1) % 2) * dim:((m - 1) % 2 + 1) * dim])
    ckpt.vec_disk_write(os.path.join(d, "lanczosV%d.dat.tmp" % m), v1[(m % 2) * dim:(m % 2 + 1) * dim])     # interrupted mid-write
    ck = init(maxit, dim, "sr_val0", d)
    assert ck is not None and ck["k"] == m_old and ck["state"] == s0
    assert np.array_equal(ck["v_pair"], v0) and np.array_equal(ck["hessenberg"][:m_old + 1], h0[:m_old + 1])
    assert {n: open(os.path.join(d, n), "rb").read() for n in os.listdir(d)} == committed


@pytest.mark.parametrize("native", [False, True])
def test_vectors_of_the_committed_step_are_never_rewritten_in_place(tmp_path, native):
    """every = 1: V(m-1) of the new update is V(m_old) of the committed checkpoint.  It is replaced by rename, so a
    reader (or a crash) never sees a partly written file under the final name: the inode changes, the bytes do not."""
    dim, maxit = 16, 40
    d = str(tmp_path / "ck")
    update = ckpt.native_ckpt_update if native else ckpt.ckpt_lanczos_update
    h0, v0, s0 = _snapshot(d, dim, maxit, 7, 5)
    update(5, maxit, dim, s0, v0, h0, "sr_val0", directory=d)
    f = os.path.join(d, "lanczosV5.dat")
    before, ino = open(f, "rb").read(), os.stat(f).st_ino
    h1, v1, s1 = _snapshot(d, dim, maxit, 8, 6)
    v1[dim:2 * dim] = v0[dim:2 * dim]                 # slot 1 = v[5], unchanged between the two steps
    update(6, maxit, dim, s1, v1, h1, "sr_val0", directory=d)
    assert open(f, "rb").read() == before and os.stat(f).st_ino != ino
    assert not [n for n in os.listdir(d) if n.endswith(".tmp")]


def test_stale_files_of_an_earlier_run_never_mix_with_a_new_one(tmp_path):
    d = str(tmp_path / "out_Qckpt")
    dim, maxit = 20, 40
    h, v, st = _snapshot(d, dim, maxit, 5, 30)
    ckpt.ckpt_lanczos_update(30, maxit, dim, st, v, h, "sr_val0", directory=d)       # earlier run stopped at step 30
    # a restart from scratch purges everything ...
    ckpt.ckpt_purge(d)
    assert os.listdir(d) == []
    # ... and an update at a lower step removes stale higher indices and rewrites V(m-1) even if a file of that name exists
    ckpt.ckpt_lanczos_update(30, maxit, dim, st, v, h, "sr_val0", directory=d)
    open(os.path.join(d, "lanczosV9.dat"), "wb").write(b"stale vector of another run")
    h2, v2, st2 = _snapshot(d, dim, maxit, 6, 10)
    ckpt.ckpt_lanczos_update(10, maxit, dim, st2, v2, h2, "sr_val0", directory=d)
    assert sorted(n for n in os.listdir(d) if n.startswith("lanczosV")) == ["lanczosV10.dat", "lanczosV9.dat"]
    ck = ckpt.ckpt_lanczos_init(maxit, dim, "sr_val0", d)
    assert ck["k"] == 10 and np.array_equal(ck["v_pair"], v2)


def test_native_checkpoint_code_is_file_compatible_with_the_python_mirror(tmp_path):
    """qbh_ckpt.cpp (own table CRC-32) against ckpt.py (zlib): each side reads what the other wrote; the CRC matches the
    published check value of CRC-32/ISO-HDLC ("123456789" -> 0xCBF43926), which neither implementation was derived from."""
    import ctypes as C
    from quantum_basis_amd import _lib
    L = _lib.lib()
    assert L.qbh_crc32(0, b"123456789", 9) == 0xCBF43926 == zlib.crc32(b"123456789")
    assert L.qbh_crc32(L.qbh_crc32(0, b"1234", 4), b"56789", 5) == 0xCBF43926          # running form
    x = (np.arange(70001) * 0.5 + 2j * np.arange(70001)).astype(np.complex128)
    f1, f2 = str(tmp_path / "a.dat"), str(tmp_path / "b.dat")
    ckpt.native_vec_disk_write(f1, x)
    ckpt.vec_disk_write(f2, x)
    assert open(f1, "rb").read() == open(f2, "rb").read()
    assert np.array_equal(ckpt.vec_disk_read(f1, x.size, np.complex128), x)
    assert np.array_equal(ckpt.native_vec_disk_read(f2, x.size, np.complex128), x)
    assert ckpt.native_vec_disk_read(f2, x.size - 1, np.complex128) is None
    bad = bytearray(open(f2, "rb").read())
    bad[1000] ^= 1
    open(f2, "wb").write(bytes(bad))
    assert ckpt.native_vec_disk_read(f2, x.size, np.complex128) is None
    # directory protocol: written natively, read by the Python mirror, and the other way round
    dim, maxit = 30, 40
    h7, v7, s7 = _snapshot(None, dim, maxit, 2, 7)
    d1, d2 = str(tmp_path / "n"), str(tmp_path / "p")
    ckpt.native_ckpt_update(7, maxit, dim, s7, v7, h7, "sr_val0", directory=d1)
    ckpt.ckpt_lanczos_update(7, maxit, dim, s7, v7, h7, "sr_val0", directory=d2)
    assert sorted(os.listdir(d1)) == sorted(os.listdir(d2)) == ["HessenbergA.dat", "HessenbergB.dat", "lanczosV6.dat", "lanczosV7.dat", "lczs_mlns.dat"]
    for n in os.listdir(d1):
        assert open(os.path.join(d1, n), "rb").read() == open(os.path.join(d2, n), "rb").read(), n
    a, b = ckpt.ckpt_lanczos_init(maxit, dim, "sr_val0", d1), ckpt.native_ckpt_init(maxit, dim, "sr_val0", d2)
    assert a["k"] == b["k"] == 7 and a["state"] == b["state"] == s7
    assert np.array_equal(a["v_pair"], v7) and np.array_equal(b["v_pair"], v7)
    assert np.array_equal(a["hessenberg"][:8], b["hessenberg"][:8]) and np.array_equal(a["hessenberg"][maxit:maxit + 7], b["hessenberg"][maxit:maxit + 7])
    # the native init finishes / rewinds a torn update exactly like the Python mirror (src/ckpt.cc:40-100)
    h8, v8, s8 = _snapshot(None, dim, maxit, 3, 8)
    open(os.path.join(d2, "lczs_updt.Qckpt1"), "wb").write(struct.pack("<q", 8))
    ckpt.vec_disk_write(os.path.join(d2, "HessenbergA.dat.new"), h8[maxit:maxit + 8])
    ckpt.vec_disk_write(os.path.join(d2, "lanczosV8.dat"), v8[:dim])
    b = ckpt.native_ckpt_init(maxit, dim, "sr_val0", d2)                      # no second marker: rewound to step 7
    assert b["k"] == 7 and np.array_equal(b["v_pair"], v7)
    assert sorted(os.listdir(d2)) == ["HessenbergA.dat", "HessenbergB.dat", "lanczosV6.dat", "lanczosV7.dat", "lczs_mlns.dat"]
    assert ckpt.native_ckpt_init(maxit, dim, "sr_val0", str(tmp_path / "missing")) is None


@pytest.mark.gpu
def test_native_checkpointed_run_interrupted_and_resumed(tmp_path):
    """qbh_lanczos_ckpt (C ABI): 30 steps, "crash", resume from the files in a new operator -- and a run started by the
    Python mirror is continued by the native code."""
    import helpers
    import quantum_basis_amd as q
    d, ia, ja, val, sym = helpers.case("hubbard_4x2")
    g = helpers.probe()["hubbard_4x2"]
    maxit = 1000
    for starter in ("native", "python"):
        ckdir = str(tmp_path / ("ck_" + starter))
        A = q.csr_mat(d, ia, ja, val, sym)
        if starter == "native":
            m1, hess1, _, conv1 = ckpt.native_lanczos_checkpointed(A, maxit, "sr_val0", every=10, directory=ckdir, max_steps=30)
        else:
            m1, hess1, _, conv1 = ckpt.lanczos_checkpointed(A, maxit, "sr_val0", every=10, directory=ckdir, max_steps=30)
        assert m1 == 30 and not conv1
        A.destroy()
        B = q.csr_mat(d, ia, ja, val, sym)
        m2, hess2, v_pair, conv2 = ckpt.native_lanczos_checkpointed(B, maxit, "sr_val0", every=25, directory=ckdir)
        assert conv2 and abs(m2 - g["lanczos_m"]) <= 1
        ritz, _ = q.hess_eigen(hess2, maxit, m2, "sr")
        assert abs(ritz[0] - g["E0"]) <= 1e-10 * abs(g["E0"])
        assert np.array_equal(hess2[maxit:maxit + 30], hess1[maxit:maxit + 30])       # the first 30 steps came from disk
        assert abs(np.linalg.norm(v_pair[:d]) - 1.0) < 1e-12
        B.destroy()


@pytest.mark.gpu
def test_native_checkpoint_of_the_first_excited_state_run(tmp_path):
    """purpose "sr_val1" (src/model.cc:1233-1265): three live vectors, phi0 travels through lanczosY0.dat; interrupted after 20
    steps and resumed by a new operator object, E1 equals the uncheckpointed two-state solve."""
    import helpers
    import quantum_basis_amd as q
    d, ia, ja, val, sym = helpers.case("hubbard_4x2")
    maxit = 600
    A = q.csr_mat(d, ia, ja, val, sym)
    ref = q.locate_E0_lanczos(A, nev=2, ncv=1, maxit=maxit)
    phi0 = ref.eigenvecs[:d]
    v0 = q.vec_randomize(A, seed=1)
    v0 = v0 - np.vdot(phi0, v0) * phi0
    v0 /= np.linalg.norm(v0)
    ckdir = str(tmp_path / "ck1")
    m1, _, _, conv1 = ckpt.native_lanczos_checkpointed(A, maxit, "sr_val1", every=10, directory=ckdir, v0=v0, phi0=phi0, max_steps=20)
    assert m1 == 20 and not conv1
    assert os.path.exists(os.path.join(ckdir, "lanczosY0.dat"))
    assert np.array_equal(ckpt.vec_disk_read(os.path.join(ckdir, "lanczosY0.dat"), d, np.complex128), phi0)
    A.destroy()
    B = q.csr_mat(d, ia, ja, val, sym)
    m2, hess2, _, conv2 = ckpt.native_lanczos_checkpointed(B, maxit, "sr_val1", every=50, directory=ckdir, v0=np.zeros(d), phi0=np.zeros(d))
    assert conv2 and abs(m2 - ref.steps["E1"]) <= 2
    ritz, _ = q.hess_eigen(hess2, maxit, m2, "sr")
    assert abs(ritz[0] - ref.E1) < 1e-9
    B.destroy()


@pytest.mark.gpu
def test_interrupted_run_resumes_to_the_same_answer(tmp_path):
    import helpers
    import quantum_basis_amd as q
    d, ia, ja, val, sym = helpers.case("hubbard_4x2")
    g = helpers.probe()["hubbard_4x2"]
    maxit = 1000
    ckdir = str(tmp_path / "out_Qckpt")
    A = q.csr_mat(d, ia, ja, val, sym)
    m1, hess1, _, conv1 = ckpt.lanczos_checkpointed(A, maxit, "sr_val0", every=10, directory=ckdir, max_steps=30)
    assert m1 == 30 and not conv1
    A.destroy()
    B = q.csr_mat(d, ia, ja, val, sym)                    # "new process": everything comes back from the files
    m2, hess2, v_pair, conv2 = ckpt.lanczos_checkpointed(B, maxit, "sr_val0", every=25, directory=ckdir)
    assert conv2 and abs(m2 - g["lanczos_m"]) <= 1
    ritz, _ = q.hess_eigen(hess2, maxit, m2, "sr")
    assert abs(ritz[0] - g["E0"]) <= 1e-10 * abs(g["E0"])
    assert np.array_equal(hess2[maxit:maxit + 30], hess1[maxit:maxit + 30])       # the first 30 steps came from disk
    # uninterrupted reference run on the same device path
    v = np.zeros(2 * d, dtype=np.complex128)
    v[:d] = q.vec_randomize(B, seed=1)
    h0 = np.zeros(2 * maxit)
    m0 = q.lanczos(0, maxit - 1, maxit, d, B, v, h0, "sr_val0")
    assert abs(m0 - m2) <= 1
    assert np.allclose(h0[maxit:maxit + 40], hess2[maxit:maxit + 40], rtol=1e-9)
    assert abs(np.linalg.norm(v_pair[:d]) - 1.0) < 1e-12
