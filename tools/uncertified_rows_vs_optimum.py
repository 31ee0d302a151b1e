#!/usr/bin/env python
"""TEST INFRASTRUCTURE (CPU; oracle twins).  How good are the rows the banded aligner could NOT certify?  For a sample of
(centre window, copy window) pairs of a C3-like workload -- the copy finder's twin gives the copies, windows carry 50 flanking
bases as in the pipeline -- every pair is aligned with the product's schedule (orc_align_pair, band cap 8 words) and, when no band
gave a certificate, with the band-free definition (orc_nw_pair): is the banded cost the optimum, are the ops the canonical
optimal ops?

    python tools/uncertified_rows_vs_optimum.py [genome Mbp, default 20] > profiles/rNN_uncertified_rows_vs_optimum.txt
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

COMP = bytes.maketrans(b"ACGTN", b"TGCAN")


def main():
    mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    w = synth.make_workload(genome_bp=mbp * 1_000_000, n_tir=int(2.5 * mbp), n_ltr=int(2.5 * mbp), cands_per_family=10, seed=20250927 + 3,
                            device=torch.device("cpu"))
    genome = w["genome"].numpy()
    coff = np.asarray(w["contig_off"], dtype=np.int64)
    contigs = [genome[coff[i]:coff[i + 1]].tobytes() for i in range(len(coff) - 1)]
    n = len(w["cand_off"]) - 1
    rng = np.random.default_rng(11)
    pick = rng.permutation(n)[:int(os.environ.get("UNC_CANDS", "160"))]
    cands = [bytes(w["cands"][w["cand_off"][c]:w["cand_off"][c + 1]]) for c in pick]
    tab = O.find_copies(contigs, cands)
    flank = 50
    pairs = cert = unc = unc_opt_cost = unc_opt_ops = 0
    excess = []
    by_level = {}
    for copies in tab:
        wins = []
        for (ci, s1, e1, minus, _a) in copies[:100]:
            lo, hi = s1 - 1 - flank, e1 + flank
            if lo < 0 or hi > len(contigs[ci]) or hi - lo < 100 or hi - lo > 32767:
                continue
            s = contigs[ci][lo:hi]
            wins.append(s.translate(COMP)[::-1] if minus else s)
        if len(wins) < 2:
            continue
        centre = wins[0]
        for row in wins[1:21]:
            ops, info = O.align_pair(centre, row, 8)
            if ops is None:
                continue
            pairs += 1
            by_level[info["nw"]] = by_level.get(info["nw"], 0) + 1
            if info["cert"]:
                cert += 1
                continue
            unc += 1
            ops_ref, d = O.nw_pair(centre, row)
            unc_opt_cost += info["U"] == d
            unc_opt_ops += bool(info["U"] == d and np.array_equal(ops, ops_ref))
            if info["U"] != d:
                excess.append((info["U"] - d, d, len(centre), len(row)))
    print("# tools/uncertified_rows_vs_optimum.py %d -- %d Mbp synthetic genome (TIR + LTR families as in C3), %d random candidates, up to 20 copy windows" % (mbp, mbp, len(pick)))
    print("# each against the first (centre) window; orc_align_pair with band cap 8 (the product's default), orc_nw_pair = the band-free definition")
    print("pairs aligned                                  %6d   (band words of the run kept: %s)" % (pairs, ", ".join("%d: %d" % kv for kv in sorted(by_level.items()))))
    print("certified (provably the definition's alignment) %6d   %.3f" % (cert, cert / max(1, pairs)))
    print("not certified                                  %6d   %.3f" % (unc, unc / max(1, pairs)))
    print("  of those: banded cost == optimal cost        %6d   %.3f" % (unc_opt_cost, unc_opt_cost / max(1, unc)))
    print("  of those: ops == the canonical optimal ops   %6d   %.3f" % (unc_opt_ops, unc_opt_ops / max(1, unc)))
    if excess:
        ex = np.array([e[0] for e in excess], dtype=np.float64)
        rel = np.array([e[0] / max(1, e[1]) for e in excess])
        print("  the others: cost above the optimum           median %d, 90 %% %d, max %d  (relative: median %.3f, max %.3f)"
              % (np.median(ex), np.quantile(ex, 0.9), ex.max(), np.median(rel), rel.max()))
    tot_ok = cert + unc_opt_ops
    print("rows that ARE the definition's alignment (certified or not): %d of %d = %.3f" % (tot_ok, pairs, tot_ok / max(1, pairs)))


if __name__ == "__main__":
    main()
