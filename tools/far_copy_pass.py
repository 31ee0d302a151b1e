#!/usr/bin/env python
"""TEST INFRASTRUCTURE (CPU; the copy finder's twin + the oracle chain).  VERDICT round 4, item 7 / DESIGN.md section 9 lead 1: does a second,
shorter-k seed pass for candidates with few copies lift the TE calls?  The twin carries the pass as a MEASUREMENT AID (orc_find_copies_far:
candidates that come out of the (10, 15) search with fewer than `far_min` copies are searched again in an (8, 13) minimizer index of the same
genome, the larger table stands; the product has no such pass).  For far_min in 0 (off) / 10 / 30 / 100 / all: the recall of the planted
full-length copies at 20-25 % pair divergence (tools/copy_recall_by_divergence.py's table), the copies found, and what the oracle chain
(tests/oracle_pipeline.py: flank windows, star alignment, judge_boundary_v5) calls on the same candidates with each copy table.

    python tools/far_copy_pass.py [genome Mbp, default 20] [candidates, default 600] > profiles/rNN_far_copy_pass.txt
"""
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from hite_amd import synth  # noqa: E402
import oracle_lib as O  # noqa: E402
import oracle_pipeline as OP  # noqa: E402

ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)
_G = {}


def _judge(args):
    cand, copies = args
    return OP.fine_stage_candidate("tir", cand, copies, _G["contigs"], plant=1)


def _end(ref, cons):
    from test_gpu_scale import _find

    so = _find(ref[:60], cons[:16])
    if so is None:
        o2 = _find(cons[:60], ref[:16])
        so = -o2 if o2 is not None else None
    eo = _find(ref[-60:][::-1], cons[-16:][::-1])
    if eo is None:
        o2 = _find(cons[-60:][::-1], ref[-16:][::-1])
        eo = -o2 if o2 is not None else None
    return so, eo


def main():
    mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ncand = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    seed = 20250927 + 2
    w = synth.make_workload(genome_bp=mbp * 1_000_000, n_tir=5 * mbp, n_ltr=0, cands_per_family=10, seed=seed, device=torch.device("cpu"))
    fams = synth.make_families(np.random.default_rng(seed), 5 * mbp, 0)
    genome = w["genome"].numpy()
    coff = np.asarray(w["contig_off"], dtype=np.int64)
    contigs = [genome[coff[i]:coff[i + 1]].tobytes() for i in range(len(coff) - 1)]
    _G["contigs"] = contigs
    n = len(w["cand_off"]) - 1
    pick = np.random.default_rng(5).permutation(n)[:ncand]
    cands = [bytes(w["cands"][w["cand_off"][c]:w["cand_off"][c + 1]]) for c in pick]
    p = w["planted"]
    order = np.argsort(p["family"], kind="stable")
    fam_sorted = p["family"][order]
    print("# tools/far_copy_pass.py %d %d -- %d Mbp synthetic genome, %d TIR families, %d candidates (ends moved by up to 30 bp); copy finder's CPU twin" % (mbp, ncand, mbp, 5 * mbp, len(cands)))
    print("# with its measurement-aid far pass ((8, 13) minimizers for candidates with < far_min copies) + the oracle chain on every candidate")
    print("%-8s %9s %9s %9s %12s %12s %10s %12s %12s" % ("far_min", "searched", "replaced", "copies", "recall 20-25", "recall 25-30", "TE calls", "ends exact", "within 3 bp"))
    O.find_copies_far(0)
    base = O.find_copies(contigs, cands)
    for far in (0, 10, 30, 100, 1 << 20):
        O.find_copies_far(far)
        t0 = time.time()
        tab = O.find_copies(contigs, cands)
        dt = time.time() - t0
        searched = sum(len(t) < far for t in base)
        replaced = sum(len(a) != len(b) for a, b in zip(tab, base))
        tot = np.zeros(2, dtype=np.int64)
        hit = np.zeros(2, dtype=np.int64)
        for c, copies in zip(pick, tab):
            fam = int(w["family"][c])
            lo, hi = np.searchsorted(fam_sorted, [fam, fam + 1])
            fc = np.array([x[0] for x in copies], dtype=np.int64)
            fs = np.array([x[1] - 1 for x in copies], dtype=np.int64)
            fe = np.array([x[2] for x in copies], dtype=np.int64)
            fm = np.array([bool(x[3]) for x in copies], dtype=bool)
            Lc = int(w["cand_off"][c + 1] - w["cand_off"][c])
            for i in order[lo:hi]:
                s, e = int(p["start"][i]), int(p["start"][i] + p["length"][i])
                d = float(w["cand_div"][c] + p["div"][i])
                if not p["full"][i] or d < 0.20 or abs((e - s) - Lc) > 0.03 * Lc:
                    continue
                b = 0 if d < 0.25 else 1
                tot[b] += 1
                if len(fc):
                    ov = np.minimum(fe, e) - np.maximum(fs, s)
                    hit[b] += bool(((fc == p["contig"][i]) & (fm == bool(p["minus"][i])) & (ov >= 0.8 * (e - s))).any())
        with Pool(min(8, os.cpu_count() or 1)) as pool:
            res = pool.map(_judge, [(c.decode(), [x[:4] for x in t]) for c, t in zip(cands, tab)], chunksize=4)
        te = exact = near = 0
        for c, r in zip(pick, res):
            if not r[0]:
                continue
            te += 1
            cons = np.frombuffer(r[2].encode(), dtype=np.uint8)
            ref = ASCII[fams[int(w["family"][c])]["cons"]]
            if len(cons) < 40 or len(ref) < 100:
                continue
            so, eo = _end(ref, cons)
            exact += so == 0 and eo == 0
            near += so is not None and eo is not None and abs(so) <= 3 and abs(eo) <= 3
        print("%-8s %9d %9d %9d %12.3f %12.3f %10d %12d %12d   (twin %.1f s)" % ("all" if far > 100000 else str(far), searched, replaced, sum(len(t) for t in tab),
                                                                      hit[0] / max(1, tot[0]), hit[1] / max(1, tot[1]), te, exact, near, dt))
        sys.stdout.flush()
    O.find_copies_far(0)


if __name__ == "__main__":
    main()
