#!/usr/bin/env python3
"""Wall time of qbh_iram (thick-restart Lanczos, basis in HBM) on the C3 operator.  usage: tools/iram_time.py nev ncv"""
import sys
import time

sys.path.insert(0, ".")
import quantum_basis_amd as q
from quantum_basis_amd import lattices

nev, ncv = int(sys.argv[1]), int(sys.argv[2])
A = q.csr_mat.hubbard(16, 8, 8, lattices.square(4, 4), t=1.0, U=1.1)
A.sync()
A.stats(reset=True)
t0 = time.time()
nconv, w, _ = q.iram(A.dim, A, None, nev, ncv, 300, "sr", want_vectors=False) if "want_vectors" in q.iram.__code__.co_varnames else q.iram(A.dim, A, None, nev, ncv, 300, "sr")
t1 = time.time()
st = A.stats()
print("iram nev %d ncv %d: %.2f s, nconv %d, %d matvecs (%d real), eigenvalues %s" % (nev, ncv, t1 - t0, nconv, st.n_spmv, st.n_spmv_real, [float("%.12f" % x) for x in w]))
