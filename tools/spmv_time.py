#!/usr/bin/env python3
"""Time the bare SpMV launch (y = alpha*H*x + beta*y + gamma*x, the Lanczos form) of one benchmark operator under several
configurations.  A configuration is a string of key=value tokens: a lower-case key is a field of qbh_opts (kron_cols16=0,
deterministic=1, wave_walk=2 ...; site_cut=18 names the QBH_BASIS_SPIN_SECTOR cut of a Heisenberg workload), an upper-case key
an environment variable (QBH_DEBUG=grid=512,no_far_align=1: the library re-reads the debug list when its text changes).
Usage: python tools/spmv_time.py hubbard_4x4_half "" "kron_cols16=0" "QBH_DEBUG=no_far_align=1" ...
An empty string is the default configuration.  Prints ms per launch from the library's own HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import quantum_basis_amd as q  # noqa: E402


def main():
    name = sys.argv[1]
    cfgs = sys.argv[2:] or [""]
    W = bench.workloads()[name]
    fmt = os.environ.get("SPMV_FORMAT", "complex128")
    rounds = int(os.environ.get("SPMV_ROUNDS", "1"))      # cycle through the configurations this many times: min / median per configuration
    results = {}
    for cfg in cfgs * rounds:
        saved = {}
        kw = {}
        for kv in cfg.split():
            k, v = kv.split("=", 1)
            if k.isupper():
                saved[k] = os.environ.get(k)
                os.environ[k] = v
            elif k == "site_cut":
                kw.update(basis_kind=q._lib.BASIS_SPIN_SECTOR, n_sites=W["n_sites"], n_up=int(v), n_dn=W["n_dn"])
            else:
                kw[k] = int(v)
        opts = q.make_opts(profile=1, value_dict=0 if fmt == "complex128" else 1, real_fast_path=0 if fmt == "complex128" else 1, **kw)
        A = bench.build_operator(W, (0, bench.dim_of(W)), opts)
        v = A.vec(2)
        A.randomize(v.at(0), 1)
        A.randomize(v.at(A.dim), 2)
        reps = int(os.environ.get("SPMV_REPS", "8"))
        for _ in range(2):
            A.spmv(v.at(0), v.at(A.dim), 1.0, -0.3, 0.0, want_red=True)
        A.stats(reset=True)
        for _ in range(reps):
            A.spmv(v.at(0), v.at(A.dim), 1.0, -0.3, 0.0, want_red=True)
        A.sync()
        s = A.stats()
        print("%-70s %8.3f ms/launch (%d launches)" % (cfg or "(default)", s.ms_spmv / max(1, s.n_spmv), s.n_spmv), flush=True)
        results.setdefault(cfg, []).append(s.ms_spmv / max(1, s.n_spmv))
        v.free()
        A.destroy()
        for k, old in saved.items():
            if old is None:
                del os.environ[k]
            else:
                os.environ[k] = old
    if rounds > 1:
        for cfg, r in results.items():
            r = sorted(r)
            print("%-70s min %8.3f  median %8.3f ms/launch over %d rounds" % (cfg or "(default)", r[0], r[len(r) // 2], len(r)), flush=True)


if __name__ == "__main__":
    main()
