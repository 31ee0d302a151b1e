#!/usr/bin/env python
"""Timeline of the LAST pipeline step in a rocprofv3 --kernel-trace results.db: every kernel with its start (us after the step's first
kernel), duration and queue, gaps between consecutive kernels, and per queue the busy time -- what the step's wall-clock time is made of
when the kernels are short (small batches: chains of lone wavefronts, launch gaps, host read-backs).
usage: kernel_timeline.py <dir-or-db> [first-kernel-regex of a step, default cand_minimizer_kernel] [min us to print, default 30]"""
import glob
import os
import re
import sqlite3
import sys


def main():
    path = sys.argv[1]
    first = re.compile(sys.argv[2] if len(sys.argv) > 2 else "cand_minimizer_kernel")
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
    dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    if not dbs:
        sys.exit("no .db under " + path)
    con = sqlite3.connect(dbs[-1])
    views = [r[0] for r in con.execute("select name from sqlite_master where type in ('view','table')")]
    src = None
    for v in views:
        cols = [r[1] for r in con.execute("pragma table_info(%s)" % v)]
        if v == "kernels" or (src is None and "start" in cols and "end" in cols and "name" in cols and "kernel" in v.lower()):
            src, scols = v, cols
            if v == "kernels":
                break
    if src is None:
        sys.exit("no kernel view; have: " + ", ".join(views))
    qcol = "queue_id" if "queue_id" in scols else ("stream_id" if "stream_id" in scols else None)
    rows = con.execute("select name, start, end%s from %s order by start" % (", " + qcol if qcol else "", src)).fetchall()
    starts = [i for i, r in enumerate(rows) if first.search(str(r[0]))]
    if not starts:
        sys.exit("no kernel matches the step's first kernel")
    rows = rows[starts[-1]:]
    t0 = rows[0][1]
    print("# source view %s, %d kernels in the last step; times in us after the step's first kernel" % (src, len(rows)))
    busy_end, busy, qbusy, small_n, small_us = t0, 0.0, {}, 0, 0.0
    prev_end = t0
    for r in rows:
        name, s, e = str(r[0]), r[1], r[2]
        q = r[3] if qcol else 0
        d = (e - s) / 1e3
        qbusy[q] = qbusy.get(q, 0.0) + d
        if e > busy_end:
            busy += (e - max(s, busy_end)) / 1e3
            busy_end = e
        gap = (s - prev_end) / 1e3
        prev_end = max(prev_end, e)
        if d >= min_us or gap >= min_us:
            if small_n:
                print("%10s %9s        (%d kernels below %.0f us: %.0f us in all)" % ("", "", small_n, min_us, small_us))
                small_n, small_us = 0, 0.0
            print("%10.0f %9.0f  q%-3s %s%s" % ((s - t0) / 1e3, d, q, re.sub(r"\(.*", "", name)[:60], "   <- %.0f us after the last kernel ended" % gap if gap >= min_us else ""))
        else:
            small_n += 1
            small_us += d
    wall = (prev_end - t0) / 1e3
    print("# step: %.0f us from the first kernel's start to the last kernel's end; some kernel running %.0f us (%.1f %%); per queue busy us: %s" %
          (wall, busy, 100.0 * busy / wall, {k: round(v) for k, v in sorted(qbusy.items())}))


if __name__ == "__main__":
    main()
