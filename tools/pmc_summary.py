#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (sqlite rocpd output) per kernel: mean counter value per dispatch."""
import glob
import sqlite3
import sys


def main():
    root = sys.argv[1]
    like = sys.argv[2] if len(sys.argv) > 2 else "%spmv%"
    for db_path in sorted(glob.glob(root + "/g*/*results.db")):
        db = sqlite3.connect(db_path)
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
             "where kernel_name like ? group by kernel_name, counter_name")
        for k, c, n, v in db.execute(q, (like,)):
            print("%-44s %-36s n=%-3d mean=%.6g" % (k.replace("(anonymous namespace)::", "").split("(")[0][-44:], c, n, v))


if __name__ == "__main__":
    main()
