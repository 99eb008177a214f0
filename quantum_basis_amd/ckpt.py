"""Checkpoint / resume of the Lanczos run in the reference's on-disk format.

File-compatible with src/ckpt.cc + src/miscellaneous.cc:439-469 for the "val" purposes:

  out_Qckpt/HessenbergA.dat   vec_disk format, m doubles      a[0..m)
  out_Qckpt/HessenbergB.dat   vec_disk format, m+1 doubles    b[0..m]
  out_Qckpt/lanczosV<m-1>.dat, lanczosV<m>.dat   vec_disk format, dim complex128 each
  out_Qckpt/lanczosY0.dat     (sr_val1 only) phi0
  out_Qckpt/lczs_mlns.dat     int32 cnt_accuE0, double accuracy, theta0_prev, theta1_prev
  out_Qckpt/lczs_updt.Qckpt1/2   two-phase commit markers holding the step number (int64)

vec_disk format: int64 n | n * sizeof(T) payload | CRC-32 (reflected 0xEDB88320, i.e. zlib's) of
the 8 header bytes followed by the payload, stored as uint32.

The device loop is not interrupted per step: `lanczos_checkpointed` advances the run in chunks with
the continuation form lanczos(k, np, ...) (src/qbasis.h:1030), and after each chunk downloads the two
live vectors and commits a checkpoint with the same *.new + marker + rename protocol.
"""
import os
import struct
import zlib

import numpy as np

CKPT_DIR = "out_Qckpt"


def vec_disk_write(filename, x):
    """src/miscellaneous.cc:439-469."""
    x = np.ascontiguousarray(x)
    head = struct.pack("<q", x.size)
    crc = zlib.crc32(head)
    payload = x.tobytes()
    for off in range(0, len(payload), 1 << 20):               # 1 MiB chunks like the reference
        crc = zlib.crc32(payload[off:off + (1 << 20)], crc)
    with open(filename, "wb") as f:
        f.write(head)
        f.write(payload)
        f.write(struct.pack("<I", crc & 0xFFFFFFFF))
    return 0


def vec_disk_read(filename, n, dtype):
    """src/miscellaneous.cc:391-436: returns the array, or None where the reference returns 1."""
    dtype = np.dtype(dtype)
    if not os.path.exists(filename):
        return None
    if os.path.getsize(filename) != 8 + dtype.itemsize * n + 4:
        return None
    with open(filename, "rb") as f:
        head = f.read(8)
        if struct.unpack("<q", head)[0] != n:
            return None
        payload = f.read(dtype.itemsize * n)
        (crc_file,) = struct.unpack("<I", f.read(4))
    crc = zlib.crc32(payload, zlib.crc32(head)) & 0xFFFFFFFF
    if crc != crc_file:
        return None
    return np.frombuffer(payload, dtype=dtype).copy()


def _p(d, name):
    return os.path.join(d, name)


def ckpt_lanczos_update(m, maxit, dim, state, v_pair, hessenberg, purpose, phi0=None, directory=CKPT_DIR):
    """ckpt_lanczos_update for the "val" purposes (src/ckpt.cc:178-297).  v_pair: host array holding
    v[m-1], v[m] in the reference's slots ((j%2)*dim)."""
    os.makedirs(directory, exist_ok=True)
    for mk in ("lczs_updt.Qckpt1", "lczs_updt.Qckpt2"):
        if os.path.exists(_p(directory, mk)):
            os.remove(_p(directory, mk))
    with open(_p(directory, "lczs_updt.Qckpt1"), "wb") as f:
        f.write(struct.pack("<q", m))
    vec_disk_write(_p(directory, "HessenbergA.dat.new"), hessenberg[maxit:maxit + m])
    vec_disk_write(_p(directory, "HessenbergB.dat.new"), hessenberg[:m + 1])
    # the reference skips V(m-1) when the file exists (it was written by the previous update of the same run);
    # here updates are `every` steps apart, so an existing V(m-1) may be a stale file and is rewritten -- but with
    # every == 1 it also belongs to the last COMMITTED checkpoint, which must stay readable until Qckpt2 exists:
    # temporary name + atomic rename (old or new content, never a torn file)
    for k in ((m - 1, m) if m > 0 else (m,)):
        s = (k % 2) * dim
        fin = _p(directory, "lanczosV%d.dat" % k)
        vec_disk_write(fin + ".tmp", v_pair[s:s + dim])
        os.replace(fin + ".tmp", fin)
    if "val0" not in purpose and phi0 is not None:
        vec_disk_write(_p(directory, "lanczosY0.dat.new"), phi0)
    with open(_p(directory, "lczs_mlns.dat.new"), "wb") as f:
        f.write(struct.pack("<iddd", int(state["cnt_accuE0"]), state["accuracy"], state["theta0_prev"], state["theta1_prev"]))
    with open(_p(directory, "lczs_updt.Qckpt2"), "wb") as f:      # before / after this point: old / new data
        f.write(struct.pack("<q", m))
    for name in ("HessenbergA.dat", "HessenbergB.dat", "lczs_mlns.dat"):
        if os.path.exists(_p(directory, name)):
            os.remove(_p(directory, name))
    for name in os.listdir(directory):
        if name.startswith("lanczosV") and name.endswith(".dat"):
            k = int(name[len("lanczosV"):-4])
            if k < m - 1 or k > m:          # older steps (reference) and stale higher indices of an earlier run
                os.remove(_p(directory, name))
    os.replace(_p(directory, "HessenbergA.dat.new"), _p(directory, "HessenbergA.dat"))
    os.replace(_p(directory, "HessenbergB.dat.new"), _p(directory, "HessenbergB.dat"))
    if os.path.exists(_p(directory, "lanczosY0.dat.new")):
        os.replace(_p(directory, "lanczosY0.dat.new"), _p(directory, "lanczosY0.dat"))
    os.replace(_p(directory, "lczs_mlns.dat.new"), _p(directory, "lczs_mlns.dat"))
    os.remove(_p(directory, "lczs_updt.Qckpt1"))
    os.remove(_p(directory, "lczs_updt.Qckpt2"))


def _rm(path):
    if os.path.exists(path):
        os.remove(path)


def _lanczos_vec_indices(directory):
    return sorted(int(n[len("lanczosV"):-4]) for n in os.listdir(directory)
                  if n.startswith("lanczosV") and n.endswith(".dat") and n[len("lanczosV"):-4].isdigit())


def ckpt_purge(directory=CKPT_DIR):
    """Start from scratch: no file of an earlier run (other operator, other start vector, higher step index)
    may survive, or a later resume would pair it with fresh data."""
    if not os.path.isdir(directory):
        return
    for name in os.listdir(directory):
        if (name.startswith("lanczosV") or name.startswith("lanczosY") or name.startswith("Hessenberg")
                or name.startswith("lczs_")):
            os.remove(_p(directory, name))


def _recover_torn_update(directory, purpose):
    """The two branches of src/ckpt.cc:40-100 for an update that was interrupted (marker lczs_updt.Qckpt1 of the
    right size present).  Qckpt2 present: every new file was completely written -> finish the renames and the
    clean-up.  Otherwise rewind to the last COMMITTED step (not k-1: updates are `every` steps apart here)."""
    mk1, mk2 = _p(directory, "lczs_updt.Qckpt1"), _p(directory, "lczs_updt.Qckpt2")
    if not (os.path.exists(mk1) and os.path.getsize(mk1) == 8):
        _rm(mk1)
        _rm(mk2)
        return
    (k,) = struct.unpack("<q", open(mk1, "rb").read(8))
    renames = ("HessenbergA.dat", "HessenbergB.dat", "lanczosY0.dat", "lanczosY1.dat", "lczs_mlns.dat")
    if os.path.exists(mk2):                                           # src/ckpt.cc:50-79
        for name in renames:
            if os.path.exists(_p(directory, name + ".new")):
                _rm(_p(directory, name))
                os.replace(_p(directory, name + ".new"), _p(directory, name))
        if purpose != "iram":
            for kk in _lanczos_vec_indices(directory):
                if kk < k - 1 or kk > k:
                    os.remove(_p(directory, "lanczosV%d.dat" % kk))
        _rm(mk1)
        _rm(mk2)
    else:
        # src/ckpt.cc:80-97 rewinds ONE step because the reference's updates are one step apart.  Here they are
        # `every` steps apart: the committed step is the one the old HessenbergA.dat was written for (its int64
        # header); only its two vectors may survive (a torn 2-step update has already written V(m_old+1)).
        m_old = -1
        ha = _p(directory, "HessenbergA.dat")
        if os.path.exists(ha) and os.path.getsize(ha) >= 8:
            (m_old,) = struct.unpack("<q", open(ha, "rb").read(8))
        for name in renames:
            _rm(_p(directory, name + ".new"))
        for kk in _lanczos_vec_indices(directory):
            if m_old < 1 or kk not in (m_old - 1, m_old):
                os.remove(_p(directory, "lanczosV%d.dat" % kk))
        _rm(mk1)
    for name in os.listdir(directory):                                # temporary names of an interrupted vector write
        if name.endswith(".tmp"):
            os.remove(_p(directory, name))


def ckpt_lanczos_init(maxit, dim, purpose, directory=CKPT_DIR):
    """ckpt_lanczos_init for the "val" purposes (src/ckpt.cc:38-176): returns None when there is no usable
    checkpoint, else dict(k, state, v_pair, hessenberg, phi0).  An interrupted update is finished as the reference
    does, or rewound to the last committed step (the reference's "one step back" generalised to updates that are
    `every` steps apart); where the reference asserts on unreadable files this returns None."""
    if not os.path.isdir(directory):
        return None
    _recover_torn_update(directory, purpose)
    ks = _lanczos_vec_indices(directory)
    if not ks:
        return None
    # src/ckpt.cc:101-111: first existing index, then the end of the consecutive run that starts there
    m = ks[0]
    while m + 1 in ks:
        m += 1
    if m == 0 or (m - 1) not in ks:
        return None
    a = vec_disk_read(_p(directory, "HessenbergA.dat"), m, np.float64)
    b = vec_disk_read(_p(directory, "HessenbergB.dat"), m + 1, np.float64)
    v1 = vec_disk_read(_p(directory, "lanczosV%d.dat" % (m - 1)), dim, np.complex128)
    v2 = vec_disk_read(_p(directory, "lanczosV%d.dat" % m), dim, np.complex128)
    if a is None or b is None or v1 is None or v2 is None or not os.path.exists(_p(directory, "lczs_mlns.dat")):
        return None
    cnt, accuracy, t0, t1 = struct.unpack("<iddd", open(_p(directory, "lczs_mlns.dat"), "rb").read(28))
    hess = np.zeros(2 * maxit)
    hess[maxit:maxit + m] = a
    hess[:m + 1] = b
    v_pair = np.zeros(2 * dim, dtype=np.complex128)
    v_pair[((m - 1) % 2) * dim:((m - 1) % 2 + 1) * dim] = v1
    v_pair[(m % 2) * dim:(m % 2 + 1) * dim] = v2
    phi0 = None
    if "val0" not in purpose:
        phi0 = vec_disk_read(_p(directory, "lanczosY0.dat"), dim, np.complex128)
        if phi0 is None:
            return None
    return dict(k=m, state=dict(cnt_accuE0=cnt, accuracy=accuracy, theta0_prev=t0, theta1_prev=t1),
                v_pair=v_pair, hessenberg=hess, phi0=phi0)


def lanczos_checkpointed(mat, maxit, purpose="sr_val0", every=50, directory=CKPT_DIR, v0=None, phi0=None,
                         max_steps=None):
    """Run (or resume) lanczos(0, maxit-1, ...) with a checkpoint every `every` steps.
    Returns (m, hessenberg, v_pair, converged)."""
    from . import engine
    dim = mat.dim
    nvec = 3 if "val1" in purpose else 2
    ck = ckpt_lanczos_init(maxit, dim, purpose, directory)
    dv = mat.vec(nvec)
    if ck is None:
        ckpt_purge(directory)
        hess = np.zeros(2 * maxit)
        k, state = 0, None
        if v0 is None:
            mat.randomize(dv.at(0), 1)
        else:
            dv.upload(v0, 0)
        if nvec == 3:
            dv.upload(phi0, 2 * dim)
    else:
        hess, k, state = ck["hessenberg"], ck["k"], ck["state"]
        dv.upload(ck["v_pair"], 0)
        if nvec == 3:
            dv.upload(ck["phi0"] if ck["phi0"] is not None else phi0, 2 * dim)
    done_steps, converged, m = 0, False, k
    try:
        while m < maxit - 1:
            np_steps = min(every, maxit - 1 - m)
            if max_steps is not None:
                np_steps = min(np_steps, max_steps - done_steps)
                if np_steps <= 0:
                    break
            m_new = engine.lanczos(m, np_steps, maxit, dim, mat, None, hess, purpose, device_v=dv, state=state)
            state = engine.lanczos.last["state"]
            done_steps += m_new - m
            stopped_early = m_new < m + np_steps
            m = m_new
            v_pair = dv.download(0, 2 * dim)
            ckpt_lanczos_update(m, maxit, dim, state, v_pair, hess, purpose,
                                phi0=dv.download(2 * dim, dim) if nvec == 3 else None, directory=directory)
            if stopped_early or (state["cnt_accuE0"] > 15 and state["accuracy"] < engine.lanczos_precision):
                converged = True
                break
        return m, hess, dv.download(0, 2 * dim), converged
    finally:
        dv.free()


# ---- the same protocol inside libqbhip.so (qbh_ckpt.cpp): what a C++ host calls -------------------------------------
def native_vec_disk_write(filename, x):
    import ctypes as C
    from ._lib import check, lib
    x = np.ascontiguousarray(x)
    check(lib().qbh_vec_disk_write(filename.encode(), C.c_int64(x.size), int(x.dtype.itemsize), x.ctypes.data_as(C.c_void_p)),
          "qbh_vec_disk_write")
    return 0


def native_vec_disk_read(filename, n, dtype):
    import ctypes as C
    from ._lib import lib
    out = np.empty(n, dtype=np.dtype(dtype))
    rc = lib().qbh_vec_disk_read(filename.encode(), C.c_int64(n), int(out.dtype.itemsize), out.ctypes.data_as(C.c_void_p))
    return out if rc == 0 else None


def native_ckpt_update(m, maxit, dim, state, v, hessenberg, purpose, directory=CKPT_DIR):
    import ctypes as C
    from ._lib import check, lib
    v = np.ascontiguousarray(v, dtype=np.complex128)
    h = np.ascontiguousarray(hessenberg, dtype=np.float64)
    check(lib().qbh_ckpt_lanczos_update(directory.encode(), m, maxit, dim, int(state["cnt_accuE0"]), float(state["accuracy"]),
                                        float(state["theta0_prev"]), float(state["theta1_prev"]), v.ctypes.data_as(C.c_void_p),
                                        h.ctypes.data_as(C.c_void_p), purpose.encode()), "qbh_ckpt_lanczos_update")


def native_ckpt_init(maxit, dim, purpose, directory=CKPT_DIR):
    """qbh_ckpt_lanczos_init: same return convention as ckpt_lanczos_init above (None = start from scratch)."""
    import ctypes as C
    from ._lib import check, lib
    nvec = 2 if "val0" in purpose else 3
    v = np.zeros(nvec * dim, dtype=np.complex128)
    h = np.zeros(2 * maxit)
    k, cnt = C.c_int64(0), C.c_int(0)
    acc, t0, t1 = C.c_double(0), C.c_double(0), C.c_double(0)
    check(lib().qbh_ckpt_lanczos_init(directory.encode(), C.byref(k), maxit, dim, C.byref(cnt), C.byref(acc), C.byref(t0), C.byref(t1),
                                      v.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), purpose.encode()),
          "qbh_ckpt_lanczos_init")
    if k.value == 0:
        return None
    return dict(k=k.value, state=dict(cnt_accuE0=cnt.value, accuracy=acc.value, theta0_prev=t0.value, theta1_prev=t1.value),
                v_pair=v[:2 * dim].copy(), hessenberg=h, phi0=v[2 * dim:].copy() if nvec == 3 else None)


def native_lanczos_checkpointed(mat, maxit, purpose="sr_val0", every=50, directory=CKPT_DIR, v0=None, phi0=None, max_steps=0):
    """qbh_lanczos_ckpt: the whole checkpointed run inside the library.  Returns (m, hessenberg, v_pair, converged)."""
    import ctypes as C
    from ._lib import SolverInfo, check, lib
    from . import engine
    dim = mat.dim
    nvec = 3 if "val1" in purpose else 2
    v = np.zeros(nvec * dim, dtype=np.complex128)
    v[:dim] = engine.vec_randomize(mat, seed=1) if v0 is None else v0
    if nvec == 3:
        v[2 * dim:] = phi0
    hess = np.zeros(2 * maxit)
    m, conv = C.c_int64(0), C.c_int(0)
    info = SolverInfo()
    os.makedirs(directory, exist_ok=True)
    check(lib().qbh_lanczos_ckpt(mat.handle, maxit, C.byref(m), v.ctypes.data_as(C.c_void_p), hess.ctypes.data_as(C.c_void_p),
                                 purpose.encode(), every, max_steps, directory.encode(), C.byref(conv), C.byref(info)), "qbh_lanczos_ckpt")
    return m.value, hess, v[:2 * dim].copy(), bool(conv.value)


def native_cg_checkpointed(mat, maxit, E0, v0, every=20, directory=CKPT_DIR, max_steps=0):
    """qbh_eigenvec_cg_ckpt: eigenvec_CG with enable_ckpt = true (src/lanczos.cc:281-341, src/ckpt.cc:344-517) inside the library.
    Returns (m, accu, v, converged, resid) -- resid[j] = the residual of step j for the steps made by THIS call (log_CG.txt rows)."""
    import ctypes as C
    from ._lib import SolverInfo, check, lib
    dim = mat.dim
    vs = [np.ascontiguousarray(v0, dtype=np.complex128).copy()] + [np.zeros(dim, dtype=np.complex128) for _ in range(3)]
    m, conv, accu = C.c_int64(0), C.c_int(0), C.c_double(0.0)
    info = SolverInfo()
    resid = np.zeros(maxit + 2)
    info.cg_resid = resid.ctypes.data_as(C.POINTER(C.c_double))
    os.makedirs(directory, exist_ok=True)
    check(lib().qbh_eigenvec_cg_ckpt(mat.handle, maxit, C.byref(m), float(np.real(E0)), C.byref(accu), *[x.ctypes.data_as(C.c_void_p) for x in vs],
                                     every, max_steps, directory.encode(), C.byref(conv), C.byref(info)), "qbh_eigenvec_cg_ckpt")
    return m.value, accu.value, vs[0], bool(conv.value), resid


def append_log_cg(resid, m_from, m_to, filename="log_CG.txt"):
    """The rows eigenvec_CG appends to log_CG.txt (src/lanczos.cc:308-311,334-337: setprecision(10), two columns of width 20),
    read by the reference's python/lanczos_plotCG.py."""
    with open(filename, "a") as f:
        for j in range(m_from + 1, m_to + 1):
            f.write("%20d%20.10g\n" % (j, resid[j]))
