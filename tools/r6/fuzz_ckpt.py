#!/usr/bin/env python3
"""Randomised campaign over the checkpointed solvers on one GPU (round 6, outside the GPU tier): random operators (Hubbard / Heisenberg on
random bond graphs), the checkpointed Lanczos run (qbh_lanczos_ckpt) interrupted at a random step -- once, twice or three times, with
random chunk lengths, each piece in a NEW operator handle -- must end at the step (+-1), the early coefficients (1e-9) and the E0 (1e-10)
of the uninterrupted checkpointed run (the stop test's state travels through the files), and the checkpointed CG (qbh_eigenvec_cg_ckpt), interrupted likewise, at the same
eigenvector (overlap, residual) in the same number of steps (+-3).
usage: python tools/r6/fuzz_ckpt.py [cases=60] [seed=1]"""
import math
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import quantum_basis_amd as q  # noqa: E402
from quantum_basis_amd import ckpt  # noqa: E402
import fastham  # noqa: E402


def main():
    kv = dict(a.split("=", 1) for a in sys.argv[1:])
    cases, seed = int(kv.get("cases", 60)), int(kv.get("seed", 1))
    rng = np.random.default_rng(seed)
    fails, done, t0 = [], 0, time.time()
    tmp = tempfile.mkdtemp()
    while done < cases:
        n = int(rng.integers(6, 11))
        nb = int(rng.integers(n, 2 * n + 1))
        bonds = []
        while len(bonds) < nb:
            a, b = int(rng.integers(n)), int(rng.integers(n))
            if a != b:
                bonds.append((a, b))
        if int(rng.integers(3)) == 0:
            nd = int(rng.integers(2, n - 1))
            if not 300 <= math.comb(n, nd) <= 6000:
                continue
            H = fastham.heisenberg_full(n, nd, bonds, J=1.0)
            tag = "heisenberg n %d nd %d" % (n, nd)
        else:
            nu, nd = int(rng.integers(1, n)), int(rng.integers(1, n))
            if not 300 <= math.comb(n, nu) * math.comb(n, nd) <= 6000:
                continue
            H = fastham.hubbard_full(n, nu, nd, bonds, t=1.0, U=float(rng.choice([1.1, 4.0])))
            tag = "hubbard n %d nu %d nd %d" % (n, nu, nd)
        dim, ia, ja, val = fastham.to_ref_csr(sp.csr_matrix(H))
        o = dict(value_dict=int(rng.integers(2)), real_fast_path=int(rng.integers(2)), lanczos_pipeline=int(rng.integers(2)))
        mk = lambda: q.csr_mat(dim, ia, ja, val, sym=False, opts=q.make_opts(**o))      # noqa: E731
        maxit = 1000
        tag += " bonds %s %s" % (bonds, o)
        try:
            d0 = os.path.join(tmp, "c%d_ref" % done)
            A = mk()
            m_ref, hess_ref, v_ref, conv_ref = ckpt.native_lanczos_checkpointed(A, maxit, "sr_val0", every=int(rng.integers(5, 60)), directory=d0)
            assert conv_ref, "reference run did not converge"
            e0 = q.hess_eigen(hess_ref, maxit, m_ref, "sr")[0][0]
            A.destroy()
            d1 = os.path.join(tmp, "c%d_cut" % done)
            cuts = sorted(set(int(c) for c in rng.integers(1, max(2, m_ref), size=int(rng.integers(1, 4)))))
            tag += " lanczos m %d cuts %s" % (m_ref, cuts)
            prev, m, conv = 0, 0, False
            for c in cuts:
                A = mk()
                m, hess, v, conv = ckpt.native_lanczos_checkpointed(A, maxit, "sr_val0", every=int(rng.integers(1, 40)), directory=d1, max_steps=c - prev)
                A.destroy()
                prev = m
                if conv:
                    break
            if not conv:
                A = mk()
                m, hess, v, conv = ckpt.native_lanczos_checkpointed(A, maxit, "sr_val0", every=int(rng.integers(5, 60)), directory=d1)
                A.destroy()
            # (a commit normalises the two vectors it writes; the loop in between carries them unnormalised with the scale folded into the next
            # coefficients -- runs cut at different steps therefore differ in the last bits, which the recurrence amplifies: early coefficients to
            # 1e-9, the step count within one, E0 to 1e-10)
            assert conv and abs(m - m_ref) <= 1, ("steps", m, m_ref)
            k = min(m, m_ref, 20)
            sc = max(np.abs(hess_ref[maxit:maxit + k]).max(), 1.0)
            assert np.allclose(hess[maxit:maxit + k], hess_ref[maxit:maxit + k], rtol=0, atol=1e-9 * sc) and np.allclose(hess[1:k], hess_ref[1:k], rtol=0, atol=1e-9 * sc), "early coefficients"
            e0_cut = q.hess_eigen(hess, maxit, m, "sr")[0][0]
            assert abs(e0_cut - e0) <= 1e-10 * max(abs(e0), 1.0), ("E0", e0_cut, e0)
            # CG
            A = mk()
            v0 = q.vec_randomize(A, seed=1)
            d2 = os.path.join(tmp, "c%d_cg_ref" % done)
            mc_ref, accu_ref, vcg_ref, cconv, _ = ckpt.native_cg_checkpointed(A, maxit, e0, v0, every=int(rng.integers(5, 40)), directory=d2)
            A.destroy()
            assert cconv and accu_ref < 2e-12, ("CG reference", cconv, accu_ref)
            d3 = os.path.join(tmp, "c%d_cg_cut" % done)
            cut = int(rng.integers(1, max(2, mc_ref)))
            A = mk()
            m1, _, _, c1, _ = ckpt.native_cg_checkpointed(A, maxit, e0, v0, every=int(rng.integers(1, 30)), directory=d3, max_steps=cut)
            A.destroy()
            A = mk()
            m2, accu2, vcg, c2, _ = ckpt.native_cg_checkpointed(A, maxit, e0, np.zeros(dim), every=int(rng.integers(5, 40)), directory=d3)
            A.destroy()
            assert c2 and accu2 < 2e-12 and abs(m2 - mc_ref) <= 3, ("CG", m1, m2, mc_ref, accu2)
            w = np.linalg.eigvalsh(H.toarray())[:2] if dim <= 2500 else None
            if w is None or w[1] - w[0] > 1e-6:
                assert abs(abs(np.vdot(vcg, vcg_ref)) - 1.0) < 1e-8, ("CG eigenvector", abs(np.vdot(vcg, vcg_ref)))
            for d in (d0, d1, d2, d3):
                shutil.rmtree(d, ignore_errors=True)
        except Exception as e:      # noqa: BLE001
            fails.append((tag, repr(e)[:300]))
            print("FAIL", tag, "::", repr(e)[:300], flush=True)
        done += 1
    shutil.rmtree(tmp, ignore_errors=True)
    print("fuzz_ckpt: %d cases, %d failures, %.0f s (seed %d)" % (done, len(fails), time.time() - t0, seed))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
