#!/usr/bin/env python3
"""Text summary (per-kernel calls / total / average duration) of a rocprofv3 --kernel-trace --stats run."""
import glob
import sqlite3
import sys


def main():
    root, cmd = sys.argv[1], sys.argv[2]
    print("# rocprofv3 --kernel-trace --stats -- %s" % cmd)
    print("# durations in milliseconds")
    print("%-86s %8s %14s %12s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for db_path in sorted(glob.glob(root + "/*results.db")):
        db = sqlite3.connect(db_path)
        for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            print("%-86s %8d %14.1f %12.1f %7.2f" % (name[:86], calls, total / 1e3, avg / 1e3, pct))


if __name__ == "__main__":
    main()
