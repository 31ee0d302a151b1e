#!/usr/bin/env python
"""Sum rocprofv3 --pmc counters per kernel from the rocpd sqlite output.
usage: rocpd_counters.py <dir-or-db> [kernel-name-substring]"""
import glob
import os
import sqlite3
import sys


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    for db in dbs:
        con = sqlite3.connect(db)
        views = [r[0] for r in con.execute("select name from sqlite_master where type in ('view','table')")]
        v = [x for x in views if x.startswith("counters_collection")]
        if not v:
            continue
        cols = [r[1] for r in con.execute("pragma table_info(%s)" % v[0])]
        kn = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
        cn = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
        cv = "value" if "value" in cols else [c for c in cols if "value" in c][0]
        q = "select %s, %s, sum(%s), count(*) from %s group by 1, 2" % (kn, cn, cv, v[0])
        agg = {}
        for k, c, s, n in con.execute(q):
            k = str(k).split("(")[0]
            if flt and flt not in k:
                continue
            agg.setdefault(k, {})[c] = (s, n)
        for k in sorted(agg):
            print(k)
            for c in sorted(agg[k]):
                print("    %-28s %18.0f  (%d samples)" % (c, agg[k][c][0], agg[k][c][1]))


if __name__ == "__main__":
    main()
