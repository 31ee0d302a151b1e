#!/usr/bin/env python
"""Why the star alignment uses mismatch 1 / gap 3: TE calls of the oracle chain (flank windows -> star alignment ->
remove_sparse_col -> judge_boundary_v5) on a small synthetic TIR workload, with the pairwise alignments computed by the
full-matrix programme of oracle/hite_oracle_nw.c under different (mismatch, gap) costs.  Every candidate is a planted TIR
element with boundaries perturbed by up to 30 bp, so every call is a true positive; the alignment costs only change how
often the homology boundary and the TSD are found.  CPU only (test tooling, not part of the product path).

    python tools/align_cost_sweep.py [--families 60] [--genome-mbp 20]  > profiles/r02_align_cost_sweep.txt
"""
import argparse
import collections
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
import oracle_pipeline as OP  # noqa: E402
from hite_amd import synth  # noqa: E402

u8p, u16p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16)


def nw_ops(a, b, mis, gap):
    ops = np.zeros(len(a) + 1, np.uint16)
    d = O.lib().orc_nw_pair_cost(a.ctypes.data_as(u8p), len(a), b.ctypes.data_as(u8p), len(b), mis, gap, ops.ctypes.data_as(u16p))
    assert d >= 0
    return ops[:len(a)]


def star(wins, mis, gap):
    """star alignment with the layout rules of oracle/hite_oracle_msa.c (insertion blocks left-justified)"""
    ws = [np.frombuffer(w.encode(), np.uint8).copy() for w in wins]
    a, m = ws[0], len(ws[0])
    allops = [None] + [nw_ops(a, b, mis, gap) for b in ws[1:]]
    ins = np.zeros((len(ws), m + 1), int)
    for r in range(1, len(ws)):
        o, nxt = allops[r], 0
        for p in range(m + 1):
            q = int(o[p] & 0x7FFF) if p < m else len(ws[r])
            ins[r, p] = q - nxt
            if p < m:
                nxt = q if (o[p] >> 15) else q + 1
    insmax = ins.max(axis=0)
    bstart = np.concatenate([[0], np.cumsum(insmax + np.concatenate([np.ones(m, int), [0]]))])
    out = np.full((len(ws), int(bstart[-1])), ord("-"), np.uint8)
    for p in range(m):
        out[0, bstart[p] + insmax[p]] = a[p]
    for r in range(1, len(ws)):
        o, b, nxt = allops[r], ws[r], 0
        for p in range(m + 1):
            q = int(o[p] & 0x7FFF) if p < m else len(b)
            out[r, bstart[p]:bstart[p] + (q - nxt)] = b[nxt:q]
            if p < m:
                if o[p] >> 15:
                    nxt = q
                else:
                    out[r, bstart[p] + insmax[p]] = b[q]
                    nxt = q + 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--families", type=int, default=60)
    ap.add_argument("--genome-mbp", type=int, default=20)
    ap.add_argument("--max-window", type=int, default=1400)
    ap.add_argument("--max-rows", type=int, default=40)
    args = ap.parse_args()
    w = synth.make_workload(genome_bp=args.genome_mbp * 1_000_000, n_tir=args.families, n_ltr=0, cands_per_family=3, seed=78, chrom_bp=10_000_000)
    g, co = w["genome"], w["contig_off"]
    contigs = [bytes(g[co[i]:co[i + 1]]) for i in range(len(co) - 1)]
    cands = [bytes(w["cands"][w["cand_off"][i]:w["cand_off"][i + 1]]).decode() for i in range(len(w["cand_off"]) - 1)]
    tab = O.find_copies(contigs, cands)
    schemes = [("1/1", 1, 1), ("4/5", 4, 5), ("2/3", 2, 3), ("1/2", 1, 2), ("1/3", 1, 3), ("1/4", 1, 4)]
    res = collections.Counter()
    for cand, cp in zip(cands, tab):
        full = []
        for (cc, s, e, mn, _an) in cp:
            wv, _tv = O.flank_window(contigs[cc], s, e, "-" if mn else "+", 50)
            if wv is not None:
                full.append(wv)
        if not full or len(full[0]) > args.max_window:
            continue
        keep = OP.select_rows([len(x) for x in full])
        wins = [full[i] for i in keep][:args.max_rows]
        for name, mis, gap in schemes:
            msa = star(wins, mis, gap)
            kc = O.sparse_cols(msa).astype(bool)
            r, _b = O.judge("tir", np.ascontiguousarray(msa[:, kc]), cand, 1)
            res[name] += bool(r[0])
        res["n"] += 1
    print("# TE calls of %d planted TIR candidates (<= %d bp windows, <= %d rows), oracle chain, full-matrix alignments" % (res["n"], args.max_window, args.max_rows))
    print("# mismatch/gap  calls")
    for name, _m, _g in schemes:
        print("  %-12s %d" % (name, res[name]))


if __name__ == "__main__":
    main()
