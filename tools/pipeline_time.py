#!/usr/bin/env python3
"""Wall time of the whole locate_E0_lanczos pipeline (E0 by Lanczos, ground-state vector by CG; src/model.cc:1123-1316)
on the C3 operator, stored CSR and matrix-free.  usage: python tools/pipeline_time.py [nev]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import quantum_basis_amd as q
from quantum_basis_amd import lattices

nev = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bonds = lattices.square(4, 4)
for name, mf in (("csr", False), ("matrix_free", True)):
    t0 = time.time()
    A = q.csr_mat.hubbard(16, 8, 8, bonds, t=1.0, U=1.1, matrix_free=mf)
    A.sync()
    t1 = time.time()
    res = q.locate_E0_lanczos(A, nev=nev, ncv=nev)
    t2 = time.time()
    print("%-12s build %.2f s, locate_E0_lanczos(nev=%d) %.2f s: E0 %.12f%s steps %s |v0| %.15f"
          % (name, t1 - t0, nev, t2 - t1, res.E0, (" E1 %.12f" % res.E1) if nev > 1 else "",
             {k: int(v) for k, v in res.steps.items() if k in ("E0", "V0", "E1", "V1")}, np.linalg.norm(res.eigenvecs[:A.dim])),
          flush=True)
    A.destroy()
