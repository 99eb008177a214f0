#!/usr/bin/env python3
"""HBM traffic of ONE SpMV from rocprofv3 --pmc passes (sqlite rocpd output under <root>/g*/): every kernel that is dispatched
(about) once per SpMV -- the far / near passes of a Kronecker split, k_zero_cut_groups, k_kron_tile when it is not folded into
the producer, k_kron_combine, or the single launch of an unsplit operator -- mean counter value per dispatch, summed.
Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE / WRITE_SIZE are KiB, and on gfx950
FETCH_SIZE counts wide coalesced reads at half their size: read bytes = 2 * FETCH_SIZE * 1024 (cross-checked with
TCC_EA0_RDREQ_128B * 128 where that pass exists).  Writes the entry bench.py reads from profiles/traffic.json, stamped with
the hash of the kernel sources it was measured on (tools/src_hash.py): bench.py reports when the sources have changed since.
usage: traffic_entry.py <prof_root> <traffic key> <source label> <out.json>"""
import glob
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import src_hash  # noqa: E402

SPMV_LIKE = ("k_spmv_", "k_zero_cut_groups", "k_kron_tile", "k_kron_combine", "k_mf_", "k_kronc_far", "k_kronc_near", "k_sec_remainder", "k_sec_reduce")


def main():
    root, key, label, out = sys.argv[1:5]
    per = {}                      # kernel -> counter -> (n, mean)
    for db_path in sorted(glob.glob(root + "/g*/*results.db")):
        db = sqlite3.connect(db_path)
        q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
        for k, c, n, v in db.execute(q):
            name = k.replace("(anonymous namespace)::", "").split("(")[0]
            if any(t in name for t in SPMV_LIKE):
                per.setdefault(name, {})[c] = (n, v)
    if not per:
        raise SystemExit("no SpMV kernels in " + root)
    # launches of ONE SpMV: the SpMV kernels proper set the count; a helper kernel counts with as many launches per SpMV as it has
    # (the tiled copy of a cut sector is one launch per class), and not at all when it runs less often than the SpMV (the
    # tile copy in front of a driver's first step)
    ref = max(max(n for n, _ in cs.values()) for name, cs in per.items() if "k_spmv_" in name or "k_mf_" in name or "k_kronc_near" in name)
    kernels, read, write, rd128 = {}, 0.0, 0.0, 0.0
    for name, cs in sorted(per.items()):
        n = max(n for n, _ in cs.values())
        if n < 0.9 * ref or "FETCH_SIZE" not in cs:
            continue
        mult = max(1, int(round(n / ref)))
        r = 2.0 * cs["FETCH_SIZE"][1] * 1024.0 * mult
        w = cs.get("WRITE_SIZE", (0, 0.0))[1] * 1024.0 * mult
        kernels[name.replace("void ", "").replace("qbh::", "")] = {"dispatches": n, "launches_per_spmv": mult, "read_bytes": r, "write_bytes": w,
                                                                       "TCC_EA0_RDREQ_128B_x128": cs.get("TCC_EA0_RDREQ_128B_sum", (0, 0.0))[1] * 128.0 * mult}
        read += r
        write += w
        rd128 += cs.get("TCC_EA0_RDREQ_128B_sum", (0, 0.0))[1] * 128.0 * mult
    entry = {key: {"kernel": " + ".join(kernels), "read_bytes": read, "write_bytes": write, "hbm_bytes": read + write,
                   "check_TCC_EA0_RDREQ_128B_x128": rd128, "per_kernel": kernels, "source": label,
                   "kernel_sources_sha16": src_hash.kernel_sources_sha16()}}
    json.dump(entry, open(out, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
