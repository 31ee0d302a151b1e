#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) results.db: per-kernel calls / total / average, from the top_kernels view.
usage: rocpd_summary.py <dir-or-db> [header line]"""
import glob
import os
import sqlite3
import sys


def main():
    path = sys.argv[1]
    dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    if not dbs:
        sys.exit("no .db under " + path)
    if len(sys.argv) > 2:
        print("# " + sys.argv[2])
    print("# durations in microseconds; source: top_kernels view of the rocpd results.db")
    print("%-60s %6s %16s %14s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for db in dbs:
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute("pragma table_info(top_kernels)")]
        rows = con.execute("select * from top_kernels").fetchall()
        ix = {c: i for i, c in enumerate(cols)}
        name = ix.get("name", 0)
        calls = ix.get("total_calls", ix.get("calls", 1))
        tot = ix.get("total_duration", ix.get("total_duration (us)", 2))
        pct = ix.get("percentage", ix.get("percent", None))
        scale = 1.0
        # top_kernels is in nanoseconds in some builds: normalise by the column name when present
        for c in cols:
            if "ns" in c.lower() and "dur" in c.lower():
                scale = 1e-3
        for r in sorted(rows, key=lambda r: -float(r[tot])):
            t = float(r[tot]) * scale
            n = int(r[calls])
            print("%-60s %6d %16d %14d %8.2f" % (str(r[name])[:58], n, t, t / max(1, n), float(r[pct]) if pct is not None else 0.0))


if __name__ == "__main__":
    main()
