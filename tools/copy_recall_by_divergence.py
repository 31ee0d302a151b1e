#!/usr/bin/env python
"""TEST INFRASTRUCTURE (CPU; uses the copy finder's twin, oracle/hite_oracle_copies.c).  Recall of the copy finder against the
planted full-length copies, by the divergence between candidate and copy (each is cut from / is a copy that sits up to 15 % from its
family consensus, so pairs are up to 30 % apart).  Where the recall goes, in numbers, for lead 1 of DESIGN.md section 9.

    python tools/copy_recall_by_divergence.py [genome Mbp, default 20] > profiles/rNN_copy_recall_by_divergence.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from hite_amd import synth  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    w = synth.make_workload(genome_bp=mbp * 1_000_000, n_tir=5 * mbp, n_ltr=0, cands_per_family=10, seed=20250927 + 2, device=torch.device("cpu"))
    genome = w["genome"].numpy()
    coff = np.asarray(w["contig_off"], dtype=np.int64)
    contigs = [genome[coff[i]:coff[i + 1]].tobytes() for i in range(len(coff) - 1)]
    n = len(w["cand_off"]) - 1
    rng = np.random.default_rng(5)
    pick = rng.permutation(n)[:1500]
    cands = [bytes(w["cands"][w["cand_off"][c]:w["cand_off"][c + 1]]) for c in pick]
    tab = O.find_copies(contigs, cands)
    p = w["planted"]
    order = np.argsort(p["family"], kind="stable")
    fam_sorted = p["family"][order]
    edges = [0.0, 0.05, 0.10, 0.15, 0.20, 0.25, 0.301]
    tot = np.zeros(len(edges) - 1, dtype=np.int64)
    hit = np.zeros(len(edges) - 1, dtype=np.int64)
    # pairs the reference's own two 0.95 filters (Util.py:8008-8022) can accept at all: candidate and copy within 3 % of each other
    # in length (the generator moves candidate ends by up to 30 bases: on a 260-base element that alone is more than 5 %)
    tot_e = np.zeros(len(edges) - 1, dtype=np.int64)
    hit_e = np.zeros(len(edges) - 1, dtype=np.int64)
    for c, copies in zip(pick, tab):
        fam = int(w["family"][c])
        lo, hi = np.searchsorted(fam_sorted, [fam, fam + 1])
        fc = np.array([x[0] for x in copies], dtype=np.int64)
        fs = np.array([x[1] - 1 for x in copies], dtype=np.int64)
        fe = np.array([x[2] for x in copies], dtype=np.int64)
        fm = np.array([bool(x[3]) for x in copies], dtype=bool)
        for i in order[lo:hi]:
            if not p["full"][i]:
                continue
            s, e = int(p["start"][i]), int(p["start"][i] + p["length"][i])
            d = float(w["cand_div"][c] + p["div"][i])
            b = int(np.searchsorted(edges, d, "right")) - 1
            b = min(max(b, 0), len(tot) - 1)
            tot[b] += 1
            Lc = int(w["cand_off"][c + 1] - w["cand_off"][c])
            elig = abs((e - s) - Lc) <= 0.03 * Lc
            tot_e[b] += elig
            if len(fc):
                ov = np.minimum(fe, e) - np.maximum(fs, s)
                ok = bool(((fc == p["contig"][i]) & (fm == bool(p["minus"][i])) & (ov >= 0.8 * (e - s))).any())
                hit[b] += ok
                hit_e[b] += ok and elig
    print("# tools/copy_recall_by_divergence.py %d -- %d Mbp synthetic genome, %d TIR families, 1500 random candidates; the copy finder's CPU twin" % (mbp, mbp, 5 * mbp))
    print("# (HIP == twin record for record: tests/test_gpu_parity.py, bench verify.copy_tables).  Pair divergence = substitutions of the candidate's")
    print("# source copy + of the planted copy, each from the family consensus (plus 1 % indels each).")
    print("# expected intact 15-mers: a 15-mer survives with (1 - d)^15; a 2.5 kb element has ~450 minimizers (w = 10)")
    print("%-14s %10s %10s %8s   %10s %10s %8s   %s" % ("divergence", "pairs", "found", "recall", "lengths", "found", "recall", "(1-d)^15 at the bin's middle"))
    print("%-14s %10s %10s %8s   %10s" % ("", "", "", "", "within 3 %"))
    for k in range(len(tot)):
        mid = 0.5 * (edges[k] + min(edges[k + 1], 0.30))
        print("%4.2f - %4.2f   %10d %10d %8.3f   %10d %10d %8.3f   %.4f" % (edges[k], min(edges[k + 1], 0.30), tot[k], hit[k], hit[k] / max(1, tot[k]),
                                                                     tot_e[k], hit_e[k], hit_e[k] / max(1, tot_e[k]), (1 - mid) ** 15))
    print("%-14s %10d %10d %8.3f   %10d %10d %8.3f" % ("all", tot.sum(), hit.sum(), hit.sum() / max(1, tot.sum()), tot_e.sum(), hit_e.sum(),
                                                        hit_e.sum() / max(1, tot_e.sum())))


if __name__ == "__main__":
    main()
