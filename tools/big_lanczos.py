#!/usr/bin/env python3
"""Ground-state energy of sectors whose COMPLEX Lanczos vectors would not fit one GPU: matrix-free operator + Lanczos on
vectors stored as packed doubles (qbh_lanczos_real_dev), nothing else in HBM.

  kagome36   BASELINE.json configs[1]: spin-1/2 Heisenberg, 36-site kagome torus (4 x 3 cells), Sz = 0:
             dim = C(36,18) = 9,075,135,300 (no CSR can be stored: 3.4e11 nonzeros; complex vectors: 2 x 145 GB)
  kagome36a  the six-fold symmetric 36-site kagome cluster of the literature (torus (4,2) x (2,4) in unit cells)
  triangular36  the 6 x 6 triangular torus of BASELINE.json configs[4] WITHOUT the translation symmetry, Sz = 0 (same dim);
             its E0 must be the minimum over the momentum sectors (literature: E0/N = -0.5604 for N = 36)
  hubbard4x5 BASELINE.json configs[3] family: Fermi-Hubbard 4 x 5, t = 1, U = 1.1, N_up = N_dn = n (n = 7: dim 6.0e9)

usage: python tools/big_lanczos.py kagome36 [n_dn=18] [max_steps=400] [chunk=25]
       python tools/big_lanczos.py hubbard4x5 [n=7] [max_steps=400] [chunk=25] [vector]
"vector": also the ground-state vector by eigenvec_CG on packed doubles (four vectors must fit: 4 x 8 x dim bytes)."""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import quantum_basis_amd as q  # noqa: E402
from quantum_basis_amd import _lib, lattices  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "kagome36"
npart = int(sys.argv[2]) if len(sys.argv) > 2 else (7 if model == "hubbard4x5" else 18)
max_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 25
want_vector = len(sys.argv) > 5 and sys.argv[5] == "vector"
t0 = time.time()
if model in ("kagome36", "kagome36a", "triangular36"):
    bonds = {"kagome36": lambda: lattices.kagome(4, 3), "kagome36a": lambda: lattices.kagome_torus((4, 2), (2, 4)),
             "triangular36": lambda: lattices.triangular(6, 6)}[model]()
    n_sites = 36
    import os
    if model == "kagome36a" and os.environ.get("QBH_RELABEL", "0") != "0":
        bonds = lattices.relabel(bonds, lattices.KAGOME36A_LOCAL)          # experiment: site numbering with a smaller cut, same spectrum (no gain measured: DESIGN.md 4.6b)
    A = q.csr_mat.heisenberg(36, npart, bonds, J=1.0, matrix_free=True, opts=q.make_opts(profile=1))
else:
    bonds, n_sites = lattices.square(4, 5), 20
    A = q.csr_mat.hubbard(20, npart, npart, bonds, t=1.0, U=1.1, matrix_free=True, opts=q.make_opts(profile=1))
n = A.dim
print("%s: dim %d, %d bonds, equivalent CSR nnz %d (%.2f TB as complex128 CSR); vectors: 2 x %.1f GB as doubles"
      % (model, n, len(bonds), A.nnz, A.nnz * 20e-12, n * 8e-9), flush=True)
v = A.vec(2 if want_vector else 1)             # n complex128 = 2n doubles = the two Lanczos slots (4n: v, r, p, pp of CG)
_lib.check(_lib.lib().qbh_vec_randomize_real(A.handle, v.ptr, C.c_uint32(1)), "qbh_vec_randomize_real")
A.sync()
print("start vector %.1f s" % (time.time() - t0), flush=True)
maxit = max_steps + 2
hess = np.zeros(2 * maxit)
k, state, t1, ritz = 0, None, time.time(), [0.0]
while k < max_steps:
    want = min(chunk, max_steps - k)
    m = q.lanczos_real(k, want, maxit, A, v, hess, state=state)
    last = q.lanczos_real.last
    state = last["state"]
    ritz, _ = q.hess_eigen(hess, maxit, m, "sr")
    st = A.stats()
    print("step %4d  E0 = %.12f  E0/site = %.10f  accuracy %.3e  %.3f s/step (SpMV %.1f ms)"
          % (m, ritz[0], ritz[0] / n_sites, state["accuracy"], last["ms_total"] * 1e-3 / max(m - k, 1),
             st.ms_spmv / max(st.n_spmv, 1)), flush=True)
    done = m < k + want                        # the stop rule fired
    k = m
    if done:
        break
print("done: %d Lanczos steps, E0 = %.12f, %.1f s" % (k, ritz[0], time.time() - t1), flush=True)
if want_vector:
    at = lambda j: C.c_void_p(v.ptr.value + 8 * n * j)
    _lib.check(_lib.lib().qbh_vec_randomize_real(A.handle, at(0), C.c_uint32(1)), "qbh_vec_randomize_real")   # src/model.cc:1205
    t2 = time.time()
    mcg, accu = q.eigenvec_CG_real(1000, 0, A, ritz[0], at(0), at(1), at(2), at(3))
    print("ground-state vector: %d CG steps, |(H - E0) v| = %.3e, %.1f s (%.3f s/step)"
          % (mcg, accu, time.time() - t2, (time.time() - t2) / max(mcg, 1)), flush=True)
