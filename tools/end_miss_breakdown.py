#!/usr/bin/env python
"""Why do TE calls miss the planted ends?    Config C2 (100 Mbp, 500 TIR families, ~5000 candidates
cut from planted copies with their boundaries moved by up to 30 bp): the GPU fine stage calls ~2/3 of them TE, and about half
of those calls have both consensus ends exactly on the planted element.  For a sample of the calls that miss, the oracle chain
(tests/oracle_pipeline.py: the CPU restatement of the same stage, which the GPU matches call for call) is re-run with one
ingredient replaced at a time:
    found  x banded   the step as it runs (copy table of the GPU's finder, star alignment with the default band schedule)
    truth  x banded   the generator's copy table (every planted copy of the family, ends within 2 bp) instead of the finder's
    found  x wide     exact_cap 32: nearly every row certified = the optimal alignment of the definition (oracle/hite_oracle_nw.c)
    truth  x wide     both
A miss that only `truth` repairs is the copy finder's (recall / intervals), one that only `wide` repairs the aligner's band,
one that needs both is shared, one that nothing repairs is judge_boundary_v5's own answer on this family (or the input).
usage (GPU box): python tools/end_miss_breakdown.py [sample] > profiles/r04_end_miss_breakdown.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)
_G = {}


def end_offsets(cons, ref):
    """(start offset, end offset) of a consensus against the family consensus: 0 = on the planted end, > 0 inside the element,
    < 0 outside; None when an end cannot be placed"""
    from test_gpu_scale import _find

    if len(cons) < 40 or len(ref) < 100:
        return None, None
    so = _find(ref[:60], cons[:16])
    if so is None:
        o2 = _find(cons[:60], ref[:16])
        so = -o2 if o2 is not None else None
    eo = _find(ref[-60:][::-1], cons[-16:][::-1])
    if eo is None:
        o2 = _find(cons[-60:][::-1], ref[-16:][::-1])
        eo = -o2 if o2 is not None else None
    return so, eo


def _init(path, glen):
    wv = dict(np.load(path + ".npz", allow_pickle=False))
    wv["genome"] = np.memmap(path + ".genome", dtype=np.uint8, mode="r", shape=(glen,))
    co = wv["contig_off"]
    _G["wv"] = wv
    _G["contigs"] = {ci: wv["genome"][co[ci]:co[ci + 1]].tobytes() for ci in range(len(co) - 1)}


def _variant(c, table, cap):
    import oracle_lib as O
    import oracle_pipeline as OP

    wv = _G["wv"]
    cf = wv[table + "_copy_first"]
    a, b = int(cf[c]), int(cf[c + 1])
    copies = [(int(wv[table + "_contig"][i]), int(wv[table + "_start1"][i]), int(wv[table + "_end1"][i]), int(wv[table + "_minus"][i])) for i in range(a, b)]
    cand = wv["cands"][wv["cand_off"][c]:wv["cand_off"][c + 1]].tobytes().decode()
    prev = O.set_align_exact(cap)
    try:
        res = OP.fine_stage_candidate("tir", cand, copies, _G["contigs"], plant=1)
    finally:
        O.set_align_exact(prev)
    ref = ASCII[wv["fam_cons"][wv["fam_off"][wv["family"][c]]:wv["fam_off"][wv["family"][c] + 1]]]
    if not res[0]:
        return (0, None, None, len(copies))
    so, eo = end_offsets(np.frombuffer(res[2].encode(), dtype=np.uint8), ref)
    return (1, so, eo, len(copies))


def _job(c):
    return c, {(t, cap): _variant(c, t, cap) for t in ("found", "truth") for cap in (8, 32)}


def main():
    import multiprocessing as mp

    from hite_amd import synth
    from test_gpu_scale import run_fine

    sample = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    seed = 20250927 + 2
    t0 = time.time()
    R = run_fine(100, 500, 0, seed)
    w, f, calls = R["w"], R["found"], R["calls"]
    fams = synth.make_families(np.random.default_rng(seed), 500, 0)
    fam_off = np.zeros(len(fams) + 1, dtype=np.int64)
    np.cumsum([len(x["cons"]) for x in fams], out=fam_off[1:])
    fam_cons = np.concatenate([np.asarray(x["cons"], dtype=np.uint8) for x in fams])
    n = R["n"]
    status = {}
    for c in range(n):
        r = calls[c]
        if not r["is_te"]:
            status[c] = "not_te"
            continue
        cons = R["cons"][r["cons_off"]:r["cons_off"] + r["cons_len"]]
        so, eo = end_offsets(cons, ASCII[fams[int(w["family"][c])]["cons"]])
        status[c] = "unplaced" if so is None or eo is None else ("exact" if so == 0 and eo == 0 else "miss")
    count = {k: sum(1 for v in status.values() if v == k) for k in ("not_te", "unplaced", "exact", "miss")}
    rng = np.random.default_rng(7)
    miss = [c for c in range(n) if status[c] == "miss"]
    pick = sorted(int(x) for x in rng.permutation(miss)[:sample])
    nte = [c for c in range(n) if status[c] == "not_te"]
    pick_nte = sorted(int(x) for x in rng.permutation(nte)[:sample // 2])
    base = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", "hite_miss_%d" % os.getpid())
    g = R["genome"]
    np.asarray(g).tofile(base + ".genome")
    np.savez(base + ".npz", contig_off=np.asarray(w["contig_off"]), cands=np.asarray(w["cands"]), cand_off=np.asarray(w["cand_off"]),
             family=np.asarray(w["family"]), fam_cons=fam_cons, fam_off=fam_off,
             found_copy_first=f["copy_first"], found_contig=f["contig"], found_start1=f["start1"], found_end1=f["end1"], found_minus=f["minus"],
             truth_copy_first=np.asarray(w["copy_first"]), truth_contig=np.asarray(w["contig"]), truth_start1=np.asarray(w["start1"]),
             truth_end1=np.asarray(w["end1"]), truth_minus=np.asarray(w["minus"]))
    try:
        with mp.get_context("spawn").Pool(min(40, os.cpu_count() or 1), initializer=_init, initargs=(base, int(len(g)))) as pool:
            res = dict(pool.map(_job, pick + pick_nte, chunksize=4))
    finally:
        for suf in (".genome", ".npz"):
            try:
                os.remove(base + suf)
            except OSError:
                pass
    R["ctx"].close()

    def ok(v):
        return v[0] == 1 and v[1] == 0 and v[2] == 0

    print("# tools/end_miss_breakdown.py -- config C2 (100 Mbp, 500 TIR families, %d candidates, boundaries moved by up to 30 bp); MI355X + %d host processes, %.0f s" %
          (n, min(40, os.cpu_count() or 1), time.time() - t0))
    print("GPU fine stage: TE calls %d of %d; both ends exactly on the planted element %d, ends placed but not both exact %d, an end not placeable %d; not TE %d" %
          (count["exact"] + count["miss"] + count["unplaced"], n, count["exact"], count["miss"], count["unplaced"], count["not_te"]))
    same = sum(1 for c in pick if res[c][("found", 8)][:3] == (1,) + end_offsets(
        R["cons"][calls[c]["cons_off"]:calls[c]["cons_off"] + calls[c]["cons_len"]], ASCII[fams[int(w["family"][c])]["cons"]]))
    print("\nsample: %d of the %d calls with a missed end (the oracle chain on found x banded reproduces the GPU's ends for %d of them)" % (len(pick), len(miss), same))
    rows = {"repaired by the truth copy table alone": 0, "repaired by wide bands alone": 0, "repaired by either one": 0,
            "repaired only by both together": 0, "not repaired by either (judge_boundary_v5's answer on these rows)": 0}
    detail = {"truth x banded exact": 0, "found x wide exact": 0, "truth x wide exact": 0, "truth x banded: no longer TE": 0}
    off_hist = {}
    for c in pick:
        v = res[c]
        b_, c_, d_ = ok(v[("truth", 8)]), ok(v[("found", 32)]), ok(v[("truth", 32)])
        detail["truth x banded exact"] += b_
        detail["found x wide exact"] += c_
        detail["truth x wide exact"] += d_
        detail["truth x banded: no longer TE"] += v[("truth", 8)][0] == 0
        if b_ and c_:
            rows["repaired by either one"] += 1
        elif b_:
            rows["repaired by the truth copy table alone"] += 1
        elif c_:
            rows["repaired by wide bands alone"] += 1
        elif d_:
            rows["repaired only by both together"] += 1
        else:
            rows["not repaired by either (judge_boundary_v5's answer on these rows)"] += 1
        a = v[("found", 8)]
        key = (None if a[1] is None else max(-9, min(9, a[1])), None if a[2] is None else max(-9, min(9, a[2])))
        off_hist[key] = off_hist.get(key, 0) + 1
    for k, val in rows.items():
        print("  %-70s %5d  %5.1f %%" % (k, val, 100.0 * val / max(1, len(pick))))
    print("  (" + "; ".join("%s %d" % kv for kv in detail.items()) + ")")
    top = sorted(off_hist.items(), key=lambda kv: -kv[1])[:12]
    print("  most frequent (start offset, end offset) of the missed calls, + = consensus starts / ends inside the element, clipped to +-9: " +
          ", ".join("%s x%d" % (k, v_) for k, v_ in top))
    cop = [res[c][("found", 8)][3] for c in pick]
    tru = [res[c][("truth", 8)][3] for c in pick]
    print("  copies per candidate in the sample: found median %d, truth median %d" % (int(np.median(cop)), int(np.median(tru))))
    print("\nsample: %d of the %d candidates NOT called TE" % (len(pick_nte), len(nte)))
    r2 = {"TE with the truth copy table (banded)": 0, "TE with wide bands (found copies)": 0, "TE with both": 0, "never TE": 0}
    for c in pick_nte:
        v = res[c]
        r2["TE with the truth copy table (banded)"] += v[("truth", 8)][0]
        r2["TE with wide bands (found copies)"] += v[("found", 32)][0]
        r2["TE with both"] += v[("truth", 32)][0]
        r2["never TE"] += not (v[("truth", 8)][0] or v[("found", 32)][0] or v[("truth", 32)][0])
    for k, val in r2.items():
        print("  %-70s %5d  %5.1f %%" % (k, val, 100.0 * val / max(1, len(pick_nte))))


if __name__ == "__main__":
    main()
