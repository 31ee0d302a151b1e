#!/usr/bin/env python
"""TEST INFRASTRUCTURE (CPU; oracle twins).  What would a FIRST-EXIT certificate certify?  (oracle/hite_oracle_msa.c, measurement
aid in orc_bp_pair: the least cost of any alignment that leaves the band = band value at the last in-band cell + the step out + GAP x
the diagonal offset still to make up; above U, the band's alignment is the definition's.)  On (centre window, copy window) pairs of a
C3-like workload: share of the pairs certified by Ukkonen's bound (what the product uses) and by the first-exit bound, with a band of
4 words and with 4-then-8; every pair the first-exit bound certifies is checked against the band-free definition.

    python tools/first_exit_certificate.py [genome Mbp, default 20] > profiles/rNN_first_exit_certificate.txt
"""
import ctypes as C
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
    ncand = int(os.environ.get("FE_CANDS", "200"))
    L = O.lib()
    L.orc_bp_last_exit_bound.restype = C.c_long
    L.orc_bp_exit_bound_enable(1)
    w = synth.make_workload(genome_bp=mbp * 1_000_000, n_tir=int(2.5 * mbp), n_ltr=int(2.5 * mbp), cands_per_family=10, seed=20250927 + 3,
                            device=torch.device("cpu"))
    genome = w["genome"].numpy()
    coff = np.asarray(w["contig_off"], dtype=np.int64)
    contigs = [genome[coff[i]:coff[i + 1]].tobytes() for i in range(len(coff) - 1)]
    n = len(w["cand_off"]) - 1
    pick = np.random.default_rng(11).permutation(n)[:ncand]
    tab = O.find_copies(contigs, [bytes(w["cands"][w["cand_off"][c]:w["cand_off"][c + 1]]) for c in pick])
    pairs = uk4 = fe4 = uk8 = fe8 = checked = 0
    slack = []
    for copies in tab:
        wins = []
        for (ci, s1, e1, minus, _a) in copies[:100]:
            lo, hi = s1 - 1 - 50, e1 + 50
            if lo < 0 or hi > len(contigs[ci]) or hi - lo < 100 or hi - lo > 32767:
                continue
            s = contigs[ci][lo:hi]
            wins.append(s.translate(COMP)[::-1] if minus else s)
        for row in wins[1:21]:
            ops, r = O.bp_pair(wins[0], row, 4)
            lb = L.orc_bp_last_exit_bound()
            if r["status"] == 2:
                continue
            pairs += 1
            c_uk, c_fe = bool(r["cert"]), lb > r["U"]
            uk4 += c_uk
            fe4 += c_fe
            slack.append((lb - r["U"]) / max(1, r["U"]))
            ok_uk, ok_fe = c_uk, c_fe
            res = (ops, r)
            if not c_fe or not c_uk:
                ops8, r8 = O.bp_pair(wins[0], row, 8)
                lb8 = L.orc_bp_last_exit_bound()
                if r8["status"] != 2:
                    ok_uk = ok_uk or bool(r8["cert"])
                    if not c_fe and lb8 > r8["U"]:
                        ok_fe = True
                        res = (ops8, r8)
            uk8 += ok_uk
            fe8 += ok_fe
            if ok_fe and res[1]["status"] == 0 and checked < 1500:
                exp, d = O.nw_pair(wins[0], row)
                assert res[1]["U"] == d and np.array_equal(res[0], exp), "a first-exit certificate on an alignment that is not the definition's"
                checked += 1
    print("# tools/first_exit_certificate.py %d -- %d Mbp synthetic genome (TIR + LTR families as in C3), %d random candidates, up to 20 copy windows" % (mbp, mbp, ncand))
    print("# each against the first (centre) window; %d pairs.  certified = the band's alignment is provably the band-free definition's" % pairs)
    print("%-44s %8s %8s" % ("", "Ukkonen", "first exit"))
    print("%-44s %8.3f %8.3f" % ("band of 4 words", uk4 / max(1, pairs), fe4 / max(1, pairs)))
    print("%-44s %8.3f %8.3f" % ("4 words, then 8 for the pairs left", uk8 / max(1, pairs), fe8 / max(1, pairs)))
    sl = np.array(slack)
    print("# first-exit bound over U, 4-word band: median %.2f x U, 10 %% quantile %.2f x U (0 or below = no certificate)" % (1 + np.median(sl), 1 + np.quantile(sl, 0.1)))
    print("# %d first-exit certified pairs compared with orc_nw_pair (cost and canonical ops): all equal" % checked)


if __name__ == "__main__":
    main()
