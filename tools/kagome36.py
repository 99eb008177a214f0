#!/usr/bin/env python3
"""BASELINE.json configs[1]: spin-1/2 Heisenberg model on the 36-site kagome torus (4 x 3 cells), Sz = 0:
dim = C(36,18) = 9,075,135,300.  No CSR can be stored (3.4e11 nonzeros) and two complex Lanczos vectors would need
290 GB; matrix-free (qbh_mf_heisenberg) with the vectors as packed doubles (qbh_lanczos_real_dev) the ground-state energy
runs on ONE MI355X in 145 GB.   usage: python tools/kagome36.py [n_dn=18] [max_steps=400] [chunk=25]"""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import quantum_basis_amd as q
from quantum_basis_amd import _lib, lattices

n_dn = int(sys.argv[1]) if len(sys.argv) > 1 else 18
max_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 25
bonds = lattices.kagome(4, 3)
t0 = time.time()
A = q.csr_mat.heisenberg(36, n_dn, bonds, J=1.0, matrix_free=True, opts=q.make_opts(profile=1))
n = A.dim
print("dim %d, %d bonds, equivalent CSR nnz %d (%.1f TB as complex128 CSR); vectors: 2 x %.1f GB as doubles"
      % (n, len(bonds), A.nnz, A.nnz * 20e-12, n * 8e-9), flush=True)
v = A.vec(1)                                   # n complex128 = 2n doubles = the two Lanczos slots
_lib.check(_lib.lib().qbh_vec_randomize_real(A.handle, v.ptr, C.c_uint32(1)), "qbh_vec_randomize_real")
A.sync()
print("start vector %.1f s" % (time.time() - t0), flush=True)
maxit = max_steps + 2
hess = np.zeros(2 * maxit)
k, state, t1 = 0, None, time.time()
while k < max_steps:
    m = q.lanczos_real(k, min(chunk, max_steps - k), maxit, A, v, hess, state=state)
    last = q.lanczos_real.last
    state = last["state"]
    ritz, _ = q.hess_eigen(hess, maxit, m, "sr")
    st = A.stats()
    print("step %4d  E0 = %.12f  E0/N = %.10f  accuracy %.3e  %.3f s/step (SpMV %.1f ms)"
          % (m, ritz[0], ritz[0] / 36, state["accuracy"], last["ms_total"] * 1e-3 / max(m - k, 1), st.ms_spmv / max(st.n_spmv, 1)), flush=True)
    if m < k + min(chunk, max_steps - k):      # the stop rule fired
        k = m
        break
    k = m
print("done: %d Lanczos steps, E0 = %.12f, %.1f s" % (k, ritz[0], time.time() - t1), flush=True)
