import sys, time, itertools
import numpy as np
sys.path.insert(0, "/root/repo")
import quantum_basis_amd as q
from quantum_basis_amd import lattices
Lx = Ly = 4
n = 16
bonds = lattices.square(Lx, Ly)
perms, shifts = lattices.translations(Lx, Ly)
res = {}
for k in [(0, 0), (1, 0), (2, 0), (1, 1), (2, 1), (2, 2)]:
    chars = lattices.characters(shifts, k, (Lx, Ly))
    t0 = time.time()
    A = q.csr_mat.hubbard_repr(n, 8, 8, bonds, perms, chars, t=1.0, U=1.1)
    i = A.info()
    t1 = time.time()
    maxit = 600
    dv = A.vec(3)
    A.randomize(dv.at(0), 7)
    hess = np.zeros(2 * maxit)
    m = q.lanczos(0, maxit - 1, maxit, i.ncols, A, None, hess, "sr_val0", device_v=dv)
    ritz, _ = q.hess_eigen(hess, maxit, m, "sr")
    t2 = time.time()
    print(k, "dim", i.ncols, "nnz", i.nnz, "build %.2f s" % (t1 - t0), "lanczos %d steps %.2f s" % (m, t2 - t1), "E0 %.12f" % ritz[0], flush=True)
    dv.free(); A.destroy()
