#!/usr/bin/env python
"""Stage 3.1 on the GPU at scale: all-vs-all seeding (hite_seed_allvsall) + FMEA (hite_fmea_chain) on the synthetic
genome of bench.py.  Prints one JSON line (not the driver's bench contract: bench.py measures BASELINE.json's metric,
this is the companion measurement of the coarse stage quoted in DESIGN.md).
usage: bench_coarse.py [--genome-mbp 1000] [--n-tir 2500] [--n-ltr 2500] [--repeat 2]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import hite_amd
    from hite_amd import synth

    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=int, default=1000)
    ap.add_argument("--n-tir", type=int, default=2500)
    ap.add_argument("--n-ltr", type=int, default=2500)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--skip-gap", type=int, default=4000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    G = a.genome_mbp * 1_000_000
    w = synth.make_workload(genome_bp=G, n_tir=a.n_tir, n_ltr=a.n_ltr, cands_per_family=1, device=dev)
    ctx = hite_amd.Context(0)
    ctx.genome_pack_dev(w["genome"].data_ptr(), w["contig_off"])
    t0 = time.perf_counter()
    ctx.copy_index_build()
    torch.cuda.synchronize()
    t_index = time.perf_counter() - t0
    times = []
    for _ in range(a.repeat):
        t0 = time.perf_counter()
        tab = ctx.seed_allvsall(seg_len=1_000_000, max_anchors=6_000_000_000, cap=1 << 26)
        times.append(time.perf_counter() - t0)
    sc, so = ctx.seed_segments(1_000_000)
    t0 = time.perf_counter()
    oc, os_, oe = ctx.fmea_chain(tab["qseg"], tab["sseg"], tab["qs"], tab["qe"], tab["ss"], tab["se"], sc, so, a.skip_gap, 30000)
    t_fmea = time.perf_counter() - t0
    seeds, anchors, clusters, records = tab["stats"]
    # how many planted families are represented by at least one interval that covers >= 70 % of one of their copies
    print(json.dumps({"genome_bp": G, "families": a.n_tir + a.n_ltr, "segments": int(len(sc)), "index_s": round(t_index, 3),
                      "seed_allvsall_s": round(min(times), 3), "seeds": seeds, "anchors": anchors, "clusters": clusters,
                      "hsp_records": records, "anchors_per_s": round(anchors / min(times)), "fmea_s": round(t_fmea, 3),
                      "repeat_intervals": int(len(oc)), "hsp_per_s_fmea": round(records / max(t_fmea, 1e-9))}))


if __name__ == "__main__":
    main()
