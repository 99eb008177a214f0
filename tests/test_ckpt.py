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


# ---- the CG half of the protocol (src/ckpt.cc:344-517) -------------------------------------------------------------
def _cg_native(d, m, dim, seed):
    import ctypes as C
    from quantum_basis_amd._lib import check, lib
    rng = np.random.default_rng(seed)
    vs = [(rng.normal(size=dim) + 1j * rng.normal(size=dim)).astype(np.complex128) for _ in range(3)]
    check(lib().qbh_ckpt_cg_update(d.encode(), m, dim, *[x.ctypes.data_as(C.c_void_p) for x in vs]), "qbh_ckpt_cg_update")
    return vs


def _cg_init(d, maxit, dim):
    import ctypes as C
    from quantum_basis_amd._lib import check, lib
    m = C.c_int64(-1)
    vs = [np.zeros(dim, dtype=np.complex128) for _ in range(3)]
    check(lib().qbh_ckpt_cg_init(d.encode(), C.byref(m), maxit, dim, *[x.ctypes.data_as(C.c_void_p) for x in vs]), "qbh_ckpt_cg_init")
    return m.value, vs


def test_cg_checkpoint_files_and_markers_are_the_references(tmp_path):
    """ckpt_CG_update / ckpt_CG_init / ckpt_CG_clean: CG_V<m>.dat, CG_R<m>.dat, CG_P<m>.dat in vec_disk_write format, the step
    before removed after the second marker, both markers gone at the end (src/ckpt.cc:438-478)."""
    d = str(tmp_path / "out_Qckpt")
    dim, maxit = 37, 100
    assert _cg_init(d, maxit, dim)[0] == 0                                   # no directory: start from scratch
    v5 = _cg_native(d, 5, dim, 1)
    assert sorted(os.listdir(d)) == ["CG_P5.dat", "CG_R5.dat", "CG_V5.dat"]
    raw = open(os.path.join(d, "CG_R5.dat"), "rb").read()                    # int64 n | payload | CRC-32 (src/miscellaneous.cc:439-469)
    assert struct.unpack("<q", raw[:8])[0] == dim and raw[8:-4] == v5[1].tobytes() and struct.unpack("<I", raw[-4:])[0] == zlib.crc32(raw[:-4]) & 0xFFFFFFFF
    v6 = _cg_native(d, 6, dim, 2)
    assert sorted(os.listdir(d)) == ["CG_P6.dat", "CG_R6.dat", "CG_V6.dat"]  # :470-475
    m, got = _cg_init(d, maxit, dim)
    assert m == 6 and all(np.array_equal(a, b) for a, b in zip(got, v6))
    # the Python mirror's reader accepts the files (one format)
    assert np.array_equal(ckpt.vec_disk_read(os.path.join(d, "CG_V6.dat"), dim, np.complex128), v6[0])
    # torn update, first marker only (:393-405): the new step's files go, the step before is what is loaded
    open(os.path.join(d, "CG_updt.Qckpt1"), "wb").write(struct.pack("<q", 7))
    ckpt.vec_disk_write(os.path.join(d, "CG_V7.dat"), v5[0])
    ckpt.vec_disk_write(os.path.join(d, "CG_R7.dat"), v5[1])                 # (P7 never written)
    m, got = _cg_init(d, maxit, dim)
    assert m == 6 and np.array_equal(got[2], v6[2]) and sorted(os.listdir(d)) == ["CG_P6.dat", "CG_R6.dat", "CG_V6.dat"]
    # interrupted after the second marker (:382-392): the new data is complete, the old step goes
    for w, x in zip("VRP", v5):
        ckpt.vec_disk_write(os.path.join(d, "CG_%s7.dat" % w), x)
    open(os.path.join(d, "CG_updt.Qckpt1"), "wb").write(struct.pack("<q", 7))
    open(os.path.join(d, "CG_updt.Qckpt2"), "wb").write(struct.pack("<q", 7))
    m, got = _cg_init(d, maxit, dim)
    assert m == 7 and np.array_equal(got[0], v5[0]) and sorted(os.listdir(d)) == ["CG_P7.dat", "CG_R7.dat", "CG_V7.dat"]
    # a corrupted vector: nothing is loaded (the reference asserts)
    raw = bytearray(open(os.path.join(d, "CG_P7.dat"), "rb").read())
    raw[40] ^= 1
    open(os.path.join(d, "CG_P7.dat"), "wb").write(bytes(raw))
    assert _cg_init(d, maxit, dim)[0] == 0
    from quantum_basis_amd._lib import check, lib
    check(lib().qbh_ckpt_cg_clean(d.encode()), "qbh_ckpt_cg_clean")          # :480-517
    assert os.listdir(d) == []


@pytest.mark.gpu
def test_cg_eigenvector_run_interrupted_and_resumed(tmp_path):
    """qbh_eigenvec_cg_ckpt (eigenvec_CG with enable_ckpt, src/lanczos.cc:281-341): stopped after 25 steps, the files on disk are the
    reference's, a NEW operator resumes from them and ends where the uninterrupted run ends; the residuals of log_CG.txt are
    returned for the steps each call made."""
    import helpers
    import quantum_basis_amd as q
    d, ia, ja, val, sym = helpers.case("hubbard_4x2")
    g = helpers.probe()["hubbard_4x2"]
    maxit = 1000
    A = q.csr_mat(d, ia, ja, val, sym)
    E0 = q.locate_E0_lanczos(A, nev=1, ncv=0, maxit=maxit).E0 if False else g["E0"]
    v0 = q.vec_randomize(A, seed=1)
    # uninterrupted device run (no checkpoints)
    vs = [v0.copy()] + [np.zeros(d, dtype=np.complex128) for _ in range(3)]
    m_ref, accu_ref = q.eigenvec_CG(d, maxit, 0, A, E0, *vs)
    res_ref = list(q.eigenvec_CG.last["resid"])
    ck = str(tmp_path / "out_Qckpt")
    m1, accu1, _, conv1, resid1 = ckpt.native_cg_checkpointed(A, maxit, E0, v0, every=10, directory=ck, max_steps=25)
    assert m1 == 25 and not conv1
    assert sorted(os.listdir(ck)) == ["CG_P25.dat", "CG_R25.dat", "CG_V25.dat"]
    assert np.allclose(resid1[1:26], res_ref[:25], rtol=1e-6)
    A.destroy()
    B = q.csr_mat(d, ia, ja, val, sym)
    m2, accu2, v, conv2, resid2 = ckpt.native_cg_checkpointed(B, maxit, E0, np.zeros(d), every=40, directory=ck)      # v0 ignored: resumed
    assert conv2 and abs(m2 - m_ref) <= 3 and accu2 < 2e-12
    assert resid2[25] == 0.0 and resid2[26] > 0.0                             # only the steps of this call
    assert abs(np.linalg.norm(v) - 1.0) < 1e-10 and abs(abs(np.vdot(v, vs[0])) - 1.0) < 1e-9       # the same eigenvector
    y = np.empty(d, dtype=np.complex128)
    B.MultMv(v, y)
    assert np.linalg.norm(y - E0 * v) < 1e-9
    log = str(tmp_path / "log_CG.txt")
    ckpt.append_log_cg(resid1, 0, m1, log)
    ckpt.append_log_cg(resid2, m1, m2, log)
    rows = [ln.split() for ln in open(log)]
    assert [int(r[0]) for r in rows] == list(range(1, m2 + 1)) and all(len(ln) == 41 for ln in open(log))      # setw(20) x 2 + newline
    B.destroy()
