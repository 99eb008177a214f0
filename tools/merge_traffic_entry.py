#!/usr/bin/env python3
"""Merge one entry written by tools/traffic_entry.py (gpurun_out/<tag>_traffic_entry.json) into profiles/traffic.json.
usage: python tools/merge_traffic_entry.py gpurun_out/r6_c3_traffic_entry.json"""
import json
import sys

new = json.load(open(sys.argv[1]))
path = "profiles/traffic.json"
cur = json.load(open(path))
cur.update(new)
json.dump(cur, open(path, "w"), indent=1)
print("merged", list(new))
