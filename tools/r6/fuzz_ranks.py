#!/usr/bin/env python3
"""Randomised campaign over the sharded exchange forms (round 6, outside the GPU tier): random two-species operators (lattice, fillings, real or gauge-transformed to complex values) x
2..5 rank processes x parts x wire format x personalised or not x static / dynamic walks, through tests/cxx/sharded_main.cpp and the
librccl stand-in on one GPU.  Every rank must report the dense ground-state energy (1e-9), a converged eigenvector (CG residual
< 2e-12), the one-rank run's first Lanczos coefficients, and the ranks' slices must tile the rows.
usage: python tools/r6/fuzz_ranks.py [cases=60] [seed=1]"""
import math
import os
import sys
import tempfile
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import quantum_basis_amd as q  # noqa: E402
from quantum_basis_amd import lattices  # noqa: E402
import test_gpu_native_ranks as T  # noqa: E402
from test_cxx_adaptor import _build  # noqa: E402
from test_rccl_stub import _stub  # noqa: E402


def main():
    kv = dict(a.split("=", 1) for a in sys.argv[1:])
    cases, seed = int(kv.get("cases", 60)), int(kv.get("seed", 1))
    rng = np.random.default_rng(seed)
    _stub()
    fails, done, t0, n_split, n_sparse, n_gauge = [], 0, time.time(), 0, 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        rig = {"tmp": tmp, "exe": _build(tmp, "sharded_main"), "csr": None, "dim": 0, "ref": {}}
        ops = {}
        while done < cases:
            lx, ly = [(2, 2), (3, 2), (4, 2), (3, 3), (5, 2)][rng.integers(5)]
            n = lx * ly
            nu, nd = int(rng.integers(1, n)), int(rng.integers(1, n))
            NU, S = math.comb(n, nu), math.comb(n, nd)
            if NU * S > 30000 or NU < 2:
                continue
            gauge = int(rng.integers(2))                 # 1: H -> D H D^+ with a random diagonal unitary D: complex values, same spectrum, same structure
            key = (lx, ly, nu, nd, gauge)
            if key not in ops:
                G = q.csr_mat.hubbard(n, nu, nd, lattices.square(lx, ly), t=1.0, U=float(rng.choice([0.0, 1.1, 4.0])),
                                      opts=q.make_opts(kron_split=0, value_dict=0, real_fast_path=0))
                ia, ja, val = G.download()
                dim = G.dim
                G.destroy()
                if gauge:
                    ph = np.exp(2j * np.pi * rng.random(dim))
                    rows = np.repeat(np.arange(dim), np.diff(ia))
                    val = val * ph[rows] * np.conj(ph[ja])
                    val[rows == ja] = val[rows == ja].real           # the diagonal stays exactly real
                path = os.path.join(tmp, "csr_%d_%d_%d_%d_%d.bin" % key)
                with open(path, "wb") as f:
                    np.array([dim, len(ja), 0], dtype=np.int64).tofile(f)
                    ia.astype(np.int64).tofile(f), ja.astype(np.int64).tofile(f), val.tofile(f)
                H = sp.csr_matrix((val, ja, ia), shape=(dim, dim))
                w = np.linalg.eigvalsh(H.toarray()) if dim <= 3000 else sp.linalg.eigsh(H, k=2, which="SA", tol=1e-13)[0]
                ops[key] = {"csr": path, "dim": dim, "S": S, "NU": NU, "e0": float(np.min(w)), "gap": float(np.sort(w)[1] - np.min(w)), "ref": None}
            op = ops[key]
            nranks = int(rng.integers(2, 6))
            if nranks > op["NU"]:
                continue
            parts, realwire, sparse, det, pipeline = int(rng.choice([1, 2, 4, 7])), int(rng.integers(2)), int(rng.integers(2)), int(rng.integers(2)), int(rng.integers(2))
            args = ["plain=1", "kron=%d" % op["S"], "parts=%d" % parts, "realwire=%d" % realwire, "sparse=%d" % sparse, "det=%d" % det, "pipeline=%d" % pipeline]
            tag = "f%d" % done
            try:
                if op["ref"] is None:
                    op["ref"] = T._run(rig, 1, ["plain=1"], "ref_" + tag, csr=op["csr"])[0]
                ref = op["ref"]
                res = T._run(rig, nranks, args, tag, csr=op["csr"])
                if gauge and all(r["kron"] > 0 for r in res):
                    assert all(r["wire"] == 16 for r in res), ("a complex operator must travel as complex elements", [r["wire"] for r in res])
                    n_gauge += 1
                n_split += int(all(r["kron"] > 0 for r in res))
                n_sparse += int(all(r["sparse"] > 0 for r in res))
                assert res[0]["r0"] == 0 and res[-1]["r1"] == op["dim"] and all(res[i]["r1"] == res[i + 1]["r0"] for i in range(nranks - 1)), "rows"
                assert abs(ref["E0"] - op["e0"]) < 1e-9, ("one-rank E0", ref["E0"], op["e0"])
                for r in res:
                    assert abs(r["E0"] - op["e0"]) < 1e-9, ("E0", r["E0"], op["e0"])
                    assert r["accu"] < 2e-12 and abs(r["nrm"] - 1.0) < 1e-10, ("CG", r["accu"], r["nrm"])
                    k = min(r["m"], ref["m"], 8)
                    assert np.allclose(r["a"][:k], ref["a"][:k], rtol=0, atol=1e-9) and np.allclose(r["b"][:k], ref["b"][:k], rtol=0, atol=1e-9), "a, b"
                    assert np.array_equal(r["a"], res[0]["a"]), "ranks disagree on a"
                if op["gap"] > 1e-6:                       # a non-degenerate ground state: the slices are the one-rank eigenvector
                    full = np.concatenate([r["vec"] for r in res])
                    assert abs(abs(np.vdot(full, ref["vec"])) - 1.0) < 1e-7, ("overlap", abs(np.vdot(full, ref["vec"])))
            except Exception as e:      # noqa: BLE001
                fails.append((key, nranks, args, repr(e)[:300]))
                print("FAIL", key, "dim", op["dim"], "S", op["S"], "NU", op["NU"], "ranks", nranks, " ".join(args), "::", repr(e)[:300], flush=True)
            done += 1
    print("fuzz_ranks: %d cases (%d on split shards, %d with the personalised exchange, %d complex-valued split operators), %d distinct operators, %d failures, %.0f s (seed %d)" %
          (done, n_split, n_sparse, n_gauge, len(ops), len(fails), time.time() - t0, seed))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
