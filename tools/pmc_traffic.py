#!/usr/bin/env python
"""Merge two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace) into
profiles/pmc_traffic.json + a text table.  Values are KiB (MI355X_MICROARCH.md, HBM section): bytes = KiB * 1024;
on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, other widths are uncalibrated, so both the raw figure
and the x2-on-reads upper bound are kept.
usage: pmc_traffic.py <fetch_dir> <write_dir> <out_json> <out_txt> [header]"""
import glob
import json
import os
import sqlite3
import sys


def per_kernel(path, counter):
    out = {}
    for db in sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        views = [r[0] for r in con.execute("select name from sqlite_master where type in ('view','table')")]
        v = [x for x in views if x.startswith("counters_collection")]
        if not v:
            continue
        cols = [r[1] for r in con.execute("pragma table_info(%s)" % v[0])]
        kn = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
        cn = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
        did = [c for c in cols if c in ("dispatch_id", "dispatch_handle", "id")]
        q = "select %s, sum(value), count(distinct %s) from %s where %s = ? group by 1" % (kn, did[0] if did else kn, v[0], cn)
        for k, s, n in con.execute(q, (counter,)):
            k = str(k).split("(")[0].replace("void ", "")
            a, b = out.get(k, (0.0, 0))
            out[k] = (a + float(s), b + int(n))
    return out


def main():
    fdir, wdir, oj, ot = sys.argv[1:5]
    hdr = sys.argv[5] if len(sys.argv) > 5 else ""
    f = per_kernel(fdir, "FETCH_SIZE")
    w = per_kernel(wdir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fk, n1 = f.get(k, (0.0, 0))
        wk, n2 = w.get(k, (0.0, 0))
        n = max(n1, n2, 1)
        res[k] = {"launches": n, "fetch_kib": fk, "write_kib": wk, "bytes_per_launch": int((fk + wk) * 1024 / n),
                  "bytes_per_launch_fetch_x2": int((2 * fk + wk) * 1024 / n)}
    json.dump(res, open(oj, "w"), indent=1, sort_keys=True)
    with open(ot, "w") as o:
        if hdr:
            o.write("# " + hdr + "\n")
        o.write("# values are KiB summed over the launches of the run; bytes/launch(raw) = (FETCH + WRITE) * 1024 / launches\n")
        o.write("# NOTE (MI355X_MICROARCH.md, HBM section): gfx950 FETCH_SIZE under-reports wide (16 B/lane) coalesced reads by 2x and is\n")
        o.write("#       uncalibrated for the byte / dword accesses of these kernels: raw values listed, x2-on-reads is an upper bound.\n")
        o.write("%-36s %8s %16s %16s %18s %18s\n" % ("kernel", "launches", "FETCH_KiB", "WRITE_KiB", "bytes/launch(raw)", "bytes/launch(x2rd)"))
        for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["fetch_kib"] + kv[1]["write_kib"])):
            if v["fetch_kib"] + v["write_kib"] < 1024:
                continue
            o.write("%-36s %8d %16d %16d %18d %18d\n" % (k[:36], v["launches"], v["fetch_kib"], v["write_kib"], v["bytes_per_launch"],
                                                        v["bytes_per_launch_fetch_x2"]))


if __name__ == "__main__":
    main()
