#!/usr/bin/env python3
"""Randomised campaign over the sharded headline path at MID size (round 6, outside the GPU tier): bench.py --gpus N (N = 2..5 rank processes
sharing this GPU through the librccl stand-in; gloo for torch's own rendezvous) on Hubbard operators of 1e5 .. 3e6 dimensions on random
bond graphs -- split shards of whole major indices in the PARTITION order of the up configurations, 2-byte columns in both parts,
personalised exchange in parts, real wire, the comm_reserve calibration -- against bench.py --gpus 1 on the same operator: the converged
Lanczos E0 (1e-10) and its step count (+-1).
usage: python tools/r6/fuzz_ranks_mid.py [cases=12] [seed=1]"""
import json
import math
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(env, n, extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "custom", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-fast-path",
           "--no-matrix-free", "--no-locate", "--processes", "1", "--force-split"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        raise RuntimeError("bench.py --gpus %d failed (rc %d): %s" % (n, p.returncode, (p.stdout + p.stderr)[-600:]))
    return json.loads(lines[-1])


def main():
    kv = dict(a.split("=", 1) for a in sys.argv[1:])
    cases, seed = int(kv.get("cases", 12)), int(kv.get("seed", 1))
    rng = np.random.default_rng(seed)
    stub = os.path.join(ROOT, "tests", "stub_rccl", "librccl_stub.so")
    fails, done, t0 = [], 0, time.time()
    with tempfile.TemporaryDirectory() as tmp:
        while done < cases:
            n = int(rng.integers(9, 15))
            nu, nd = int(rng.integers(2, n - 1)), int(rng.integers(2, n - 1))
            NU, S = math.comb(n, nu), math.comb(n, nd)
            if not 1e5 <= NU * S <= 3e6 or NU < 16:
                continue
            nb = int(rng.integers(n, 2 * n + 1))
            bonds = []
            while len(bonds) < nb:
                a, b = int(rng.integers(n)), int(rng.integers(n))
                if a != b:
                    bonds.append([a, b])
            spec = {"kind": "hubbard", "n_sites": n, "n_up": nu, "n_dn": nd, "bonds": bonds, "t": 1.0, "U": float(rng.choice([1.1, 4.0]))}
            ranks = int(rng.integers(2, 6))
            extra = [] if rng.integers(2) else ["--no-sparse-gather"]
            if rng.integers(3) == 0:
                extra.append("--no-partition")
            tag = "n %d nu %d nd %d (dim %d, S %d) U %g ranks %d %s bonds %s" % (n, nu, nd, NU * S, S, spec["U"], ranks, extra, bonds)
            env = dict(os.environ, QBH_RCCL_LIB=stub, QBH_DIST_BACKEND="gloo", TMPDIR=tmp, HSA_ENABLE_IPC_MODE_LEGACY="0", QBH_WORKLOAD_JSON=json.dumps({"custom": spec}))
            for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            try:
                one = run(env, 1, [])
                many = run(env, ranks, extra)
                assert many["n_gpus"] == ranks and "native RCCL" in many["config"]["exchange"], many["config"].get("exchange")
                assert many["config"].get("kron_split"), "shards not split"
                assert abs(many["e0"] - one["e0"]) <= 1e-10 * max(abs(one["e0"]), 1.0), ("E0", many["e0"], one["e0"])
                assert abs(many["lanczos_steps_to_converge"] - one["lanczos_steps_to_converge"]) <= 1, ("steps", many["lanczos_steps_to_converge"], one["lanczos_steps_to_converge"])
                x = many["exchange"]
                print("ok", tag.split(" bonds")[0], "| e0 %.10f steps %d | personalised %s needed %.2f partition %s parts %s reserve %s" %
                      (many["e0"], many["lanczos_steps_to_converge"], x.get("personalised"), x.get("needed_frac_rank0") or 0.0, x.get("major_partition"), x.get("gather_parts"), x.get("comm_reserve_workgroups")), flush=True)
            except Exception as e:      # noqa: BLE001
                fails.append((tag, repr(e)[:400]))
                print("FAIL", tag, "::", repr(e)[:400], flush=True)
            done += 1
    print("fuzz_ranks_mid: %d cases, %d failures, %.0f s (seed %d)" % (done, len(fails), time.time() - t0, seed))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
