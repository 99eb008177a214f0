"""qbh_opts.major_partition (qbh_gen_hubbard): the up configurations -- the MAJOR index of the product basis -- ordered so that P
consecutive blocks of them are the parts of a recursive spectral bisection of the up-hop graph.  The operator is the same operator
in a permuted basis (entry by entry), its spectrum is unchanged, qbh_vec_randomize gives the same physical vector, and a rank's
far part reads far fewer of the other parts' major indices -- what the personalised exchange of row shards then carries
(src/sparse.cc:262-289 is the matvec served)."""
import numpy as np
import pytest
import scipy.sparse as sp

import quantum_basis_amd as q
from quantum_basis_amd import lattices

pytestmark = pytest.mark.gpu
PLAIN = dict(value_dict=0, real_fast_path=0)


def _needed(M, S, parts):
    """per part: share of the OTHER parts' major indices that the part's far entries (major index changes) read"""
    NU = M.shape[0] // S
    coo = M.tocoo()
    ru, cu_ = coo.row // S, coo.col // S
    far = ru != cu_
    G = sp.csr_matrix((np.ones(far.sum()), (ru[far], cu_[far])), shape=(NU, NU))
    out = []
    for p in range(parts):
        lo, hi = p * NU // parts, (p + 1) * NU // parts
        nb = np.asarray(G[lo:hi].sum(axis=0)).ravel() > 0
        nb[lo:hi] = False
        out.append(nb.sum() / (NU - (hi - lo)))
    return out


@pytest.mark.parametrize("shape,parts", [((4, 3, 6, 6), 4), ((4, 3, 5, 7), 2), ((4, 4, 3, 3), 8), ((3, 3, 4, 5), 3)])
def test_partition_order_is_the_same_operator_in_a_permuted_basis(shape, parts):
    lx, ly, nu, nd = shape
    n = lx * ly
    bonds = lattices.square(lx, ly)
    N = q.csr_mat.hubbard(n, nu, nd, bonds, t=1.0, U=1.3, opts=q.make_opts(kron_split=2, **PLAIN))
    P = q.csr_mat.hubbard(n, nu, nd, bonds, t=1.0, U=1.3, opts=q.make_opts(kron_split=2, major_partition=parts, **PLAIN))
    assert N.info().major_partition == 0 and P.info().major_partition == parts
    S = int(P.info().kron_minor)
    inv = P.major_order()                                   # new major index -> generator's
    NU = P.dim // S
    assert sorted(inv.tolist()) == list(range(NU))
    with pytest.raises(Exception):
        N.major_order()
    ia, ja, va = N.download()
    pa, pj, pv = P.download()
    Mn = sp.csr_matrix((va, ja.astype(np.int64), ia), shape=(N.dim, N.dim))
    Mp = sp.csr_matrix((pv, pj.astype(np.int64), pa), shape=(P.dim, P.dim))
    old = (inv.astype(np.int64)[:, None] * S + np.arange(S)[None, :]).ravel()       # row r' of P = row old[r'] of N
    ref = Mn[old][:, old]
    assert abs(Mp - ref).max() == 0.0                        # the same numbers, entry by entry
    assert all(np.all(np.diff(pj[pa[r]:pa[r + 1]]) > 0) for r in range(0, P.dim, max(1, P.dim // 997)))      # rows stay column-sorted
    # the same physical start vector, the same Lanczos run (up to the summation order of the permuted rows)
    xn, xp = q.vec_randomize(N, seed=1), q.vec_randomize(P, seed=1)
    assert np.abs(xp - xn[old]).max() < 1e-15
    rn, rp = q.locate_E0_lanczos(N), q.locate_E0_lanczos(P)
    assert abs(rn.E0 - rp.E0) <= 1e-11 * abs(rn.E0) and abs(rn.steps["E0"] - rp.steps["E0"]) <= 1
    assert abs(abs(np.vdot(rp.eigenvecs, rn.eigenvecs[old])) - 1.0) < 1e-8
    # what it is for: the parts read less of one another
    need_n, need_p = _needed(Mn, S, parts), _needed(Mp, S, parts)
    assert max(need_p) < max(need_n) and np.mean(need_p) < 0.85 * np.mean(need_n), (need_n, need_p)
    N.destroy()
    P.destroy()


def test_partition_order_is_deterministic():
    """every rank computes the order for itself: two operators (here: one whole, one a row shard) must agree on it bit for bit"""
    bonds = lattices.square(4, 3)
    A = q.csr_mat.hubbard(12, 6, 6, bonds, t=1.0, U=1.3, opts=q.make_opts(kron_split=2, major_partition=4, **PLAIN))
    S = int(A.info().kron_minor)
    B = q.csr_mat.hubbard(12, 6, 6, bonds, t=1.0, U=1.3, rows=(231 * S, 462 * S), opts=q.make_opts(kron_split=2, major_partition=4, **PLAIN))
    assert np.array_equal(A.major_order(), B.major_order())
    ia, ja, va = A.download(231 * S, 462 * S)
    ib, jb, vb = B.download()
    assert np.array_equal(ia, ib) and np.array_equal(ja, jb) and np.array_equal(va, vb)
    A.destroy()
    B.destroy()
