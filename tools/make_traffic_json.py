#!/usr/bin/env python3
"""profiles/<tag>_pmc.txt (rocprofv3 --pmc passes) -> profiles/traffic.json, the per-launch HBM traffic
bench.py reports in roofline.traffic.  Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM):
FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their size,
so read bytes = 2 * FETCH_SIZE * 1024 (cross-checked against TCC_EA0_RDREQ_128B * 128)."""
import json
import re
import sys


def parse(path, want=None):
    """want: substring the kernel line must contain (e.g. the template argument list) when the file holds several kernels"""
    vals = {}
    for line in open(path):
        if want is not None and want not in line:
            continue
        m = re.search(r"(k_\w+|_spmv_\w+)(?:<([^>]*)>)?\S*\s+(\w+)\s+n=\d+\s+mean=([0-9.e+]+)", line)
        if m:
            vals["kernel"] = m.group(1)
            vals["template"] = m.group(2) or ""
            vals[m.group(3)] = float(m.group(4))
    return vals


def main():
    out = {}
    for spec in sys.argv[1:]:
        key, path = spec.split("=")
        want = None
        if "@" in path:                       # key=file@substring-of-the-kernel-line
            path, want = path.split("@", 1)
        v = parse(path, want)
        read = 2.0 * v["FETCH_SIZE"] * 1024.0
        write = v["WRITE_SIZE"] * 1024.0
        out[key] = {"kernel": v["kernel"] + ("<" + v["template"] + ">" if v["template"] else ""), "read_bytes": read, "write_bytes": write,
                    "hbm_bytes": read + write, "FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
                    "check_TCC_EA0_RDREQ_128B_x128": v.get("TCC_EA0_RDREQ_128B_sum", 0.0) * 128.0, "source": path}
    try:
        old = json.load(open("profiles/traffic.json"))
    except Exception:
        old = {}
    old.update(out)
    out = old
    json.dump(out, open("profiles/traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
