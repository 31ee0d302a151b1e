"""Full-size GPU tests on the generator of the bench (hite_amd/synth.py): BASELINE.json config C2 (100 Mbp, 500 TIR families,
~5k candidates) through coarse + fine, and config C3 (1 Gbp, 2.5k TIR + 2.5k LTR families, 50k candidates) through the fine
stage.  Parity: random candidates re-judged by the oracle chain (tests/oracle_pipeline.py) on the copy table the GPU found.
Ground truth (the planted copies): recall / precision of the copy finder, recovery of the families by the coarse stage,
boundaries of the TE calls -- thresholds set from the measured values of round 2 (noted beside each assert)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)


def run_fine(mbp, n_tir, n_ltr, seed, te_types=("tir",), **workload_kw):
    import torch

    import hite_amd
    from hite_amd import synth
    from hite_amd._lib import CALL_DTYPE

    dev = torch.device("cuda", 0)
    w = synth.make_workload(genome_bp=mbp * 1_000_000, n_tir=n_tir, n_ltr=n_ltr, cands_per_family=10, seed=seed, device=dev, **workload_kw)
    ctx = hite_amd.Context(0)
    ctx.genome_pack_dev(w["genome"].data_ptr(), w["contig_off"])
    ctx.copy_index_build()
    n = len(w["cand_off"]) - 1
    nbytes = int(w["cand_off"][-1])
    d_cand = torch.from_numpy(np.concatenate([w["cands"], np.zeros(64, np.uint8)])).to(dev)
    d_off = torch.from_numpy(np.ascontiguousarray(w["cand_off"])).to(dev)
    d_calls = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    cap = nbytes + 200 * n + 4096
    d_cons = torch.zeros(cap + 64, dtype=torch.uint8, device=dev)
    ctx.align_stats(reset=True)
    nc, p_cf, p_ct, p_s1, p_e1, p_mn, _an = ctx.find_copies_dev(n, d_cand.data_ptr(), d_off.data_ptr(), nbytes)
    # the clip words of the records (aligned intervals, the default): the rows are padded by them; zero with HITE_COPY_INTERVAL=whole (HITE_TEST_NO_CLIP=1,
    # tools/copy_interval_modes.py: no clip pointer, as for a copy table in the reference's own form -- the library estimates the words)
    p_cl = 0 if os.environ.get("HITE_TEST_NO_CLIP") == "1" else ctx.copy_clips_dev()
    if not p_cl and nc > 0:     # (an EXTERNAL table: a copy of the start array)
        ext_s1 = torch.from_numpy(ctx.download(p_s1, nc, np.int64)).to(dev)
        p_s1 = ext_s1.data_ptr()
    # the same candidates and copy table judged as Helitron / non-LTR as well (judge_Helitron_transposons.py:86-97,
    # judge_Non_LTR_transposons.py:48-51 run the same flank_region_align_v5 with another TE_type)
    other = {}
    for te in te_types:
        if te == "tir":
            continue
        stats = ctx.flank_region_align_dev(te, 1, n, d_cand.data_ptr(), d_off.data_ptr(), p_cf, nc, p_ct, p_s1, p_e1, p_mn, 50,
                                           d_calls.data_ptr(), d_cons.data_ptr(), cap, d_clip=p_cl)
        torch.cuda.synchronize()
        other[te] = (d_calls.cpu().numpy().view(CALL_DTYPE).copy(), d_cons.cpu().numpy().copy())
    ctx.align_stats(reset=True)
    stats = ctx.flank_region_align_dev("tir", 1, n, d_cand.data_ptr(), d_off.data_ptr(), p_cf, nc, p_ct, p_s1, p_e1, p_mn, 50,
                                       d_calls.data_ptr(), d_cons.data_ptr(), cap, d_clip=p_cl)
    torch.cuda.synchronize()
    found = dict(copy_first=ctx.download(p_cf, n + 1, np.int32), contig=ctx.download(p_ct, nc, np.int32),
                 start1=ctx.download(p_s1, nc, np.int64), end1=ctx.download(p_e1, nc, np.int64), minus=ctx.download(p_mn, nc, np.uint8),
                 clip=ctx.download(p_cl, nc, np.uint32) if p_cl else np.zeros(nc, dtype=np.uint32))
    return dict(w=w, ctx=ctx, n=n, calls=d_calls.cpu().numpy().view(CALL_DTYPE).copy(), cons=d_cons.cpu().numpy(), found=found,
                other=other, stats=stats, align=ctx.align_stats(), genome=w["genome"].cpu().numpy(), seed=seed, n_tir=n_tir, n_ltr=n_ltr)


def _oracle_worker(job):
    """spawned (never forked: the parent holds a HIP context): the oracle chain on a slice of the candidates; the genome is a
    memory-mapped file, the candidates and the copy table an .npz beside it"""
    path, genome_len, te_type, cands = job[:4]
    want_anchors = len(job) > 4 and job[4]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_pipeline as OP

    z = dict(np.load(path + ".npz"))
    genome = np.memmap(path + ".genome", dtype=np.uint8, mode="r", shape=(genome_len,))
    co = z["contig_off"]
    contigs = {ci: genome[co[ci]:co[ci + 1]] for ci in range(len(co) - 1)}
    out = []
    for c in cands:
        a, b = int(z["copy_first"][c]), int(z["copy_first"][c + 1])
        copies = [(int(z["contig"][i]), int(z["start1"][i]), int(z["end1"][i]), int(z["minus"][i]), 0, int(z["clip"][i]) if "clip" in z else 0)
                  for i in range(a, b)]
        cand = z["cands"][z["cand_off"][c]:z["cand_off"][c + 1]].tobytes().decode()
        if want_anchors:      # how the two 20-base anchors sit in every alignment the chain judged (oracle_pipeline.anchor_class)
            msas = []
            exp = OP.fine_stage_candidate(te_type, cand, copies, contigs, plant=1, keep_msa=msas)
            out.append((int(c), exp, [OP.anchor_class(cand, m_) for m_ in msas]))
        else:
            out.append((int(c), OP.fine_stage_candidate(te_type, cand, copies, contigs, plant=1)))
    return out


def oracle_check(R, count, seed, te_type="tir", workers=None, anchors=None):
    """re-judge `count` random candidates with the oracle chain on the copy table the GPU found, on min(40, cores) spawned worker
    processes (as bench.py's cpu_baseline leg does); anchors (a dict, optional): filled with the count of judged alignments per
    anchor class"""
    import multiprocessing as mp
    import tempfile

    w, f = R["w"], R["found"]
    info_names = {0: "", 1: "nb", 2: "fl1", 3: "EXC"}
    calls_all, cons_all = (R["calls"], R["cons"]) if te_type == "tir" else R["other"][te_type]
    picks = [int(c) for c in np.random.default_rng(seed).permutation(R["n"])[:count]]
    workers = workers or max(1, min(40, os.cpu_count() or 1, (len(picks) + 7) // 8))
    if "oracle_files" not in R:
        d = tempfile.mkdtemp(prefix="hite_scale_")
        path = os.path.join(d, "w")
        np.asarray(R["genome"], dtype=np.uint8).tofile(path + ".genome")
        np.savez(path + ".npz", contig_off=np.asarray(w["contig_off"]), cands=w["cands"], cand_off=w["cand_off"], copy_first=f["copy_first"],
                 contig=f["contig"], start1=f["start1"], end1=f["end1"], minus=f["minus"], clip=f["clip"])
        R["oracle_files"] = (d, path)
    path = R["oracle_files"][1]
    jobs = [(path, int(len(R["genome"])), te_type, picks[k::workers], anchors is not None) for k in range(workers)]
    if workers == 1:
        res = [_oracle_worker(jobs[0])]
    else:
        with mp.get_context("spawn").Pool(workers) as pool:
            res = pool.map(_oracle_worker, jobs)
    bad, n_te = [], 0
    for part in res:
        for rec in part:
            c, exp = rec[0], rec[1]
            if anchors is not None:
                for k in rec[2]:
                    anchors[k] = anchors.get(k, 0) + 1
            r = calls_all[c]
            got = [bool(r["is_te"]), info_names[int(r["info"])],
                   cons_all[r["cons_off"]:r["cons_off"] + r["cons_len"]].tobytes().decode() if r["is_te"] else "", int(r["row_num"])]
            n_te += got[0]
            if got != exp:
                bad.append(int(c))
    return bad, n_te


def _release_oracle_files(R):
    import shutil

    if "oracle_files" in R:
        shutil.rmtree(R["oracle_files"][0], ignore_errors=True)


def _find(hay, needle, maxmm=3):
    """offset of the best occurrence of needle in hay with <= maxmm mismatches, or None"""
    k = len(needle)
    if len(hay) < k:
        return None
    win = np.lib.stride_tricks.sliding_window_view(hay, k)
    mm = (win != needle).sum(axis=1)
    o = int(np.argmin(mm))
    return o if mm[o] <= maxmm else None


def boundary_stats(R, limit=4000):
    """TE calls on candidates of TIR families: how far the ends of the consensus are from the ends of the planted element.
    -> (TIR candidates, judged TE, checked, both ends exact, both ends within 3 bp)"""
    from hite_amd import synth

    fams = synth.make_families(np.random.default_rng(R["seed"]), R["n_tir"], R["n_ltr"])
    w = R["w"]
    tir_cand = ~w["fam_is_ltr"][w["family"]]
    calls = R["calls"]
    idx = np.flatnonzero(tir_cand & (calls["is_te"] != 0))
    called = len(idx)
    exact = near = checked = 0
    for c in idx[:limit]:
        r = calls[c]
        cons = R["cons"][r["cons_off"]:r["cons_off"] + r["cons_len"]]
        ref = ASCII[fams[int(w["family"][c])]["cons"]]
        if len(cons) < 40 or len(ref) < 100:
            continue
        checked += 1
        so = _find(ref[:60], cons[:16])               # consensus starts `so` bases inside the element ...
        if so is None:
            o2 = _find(cons[:60], ref[:16])           # ... or -so bases before it
            so = -o2 if o2 is not None else None
        eo = _find(ref[-60:][::-1], cons[-16:][::-1])
        if eo is None:
            o2 = _find(cons[-60:][::-1], ref[-16:][::-1])
            eo = -o2 if o2 is not None else None
        if so is None or eo is None:
            continue
        exact += so == 0 and eo == 0
        near += abs(so) <= 3 and abs(eo) <= 3
    return int(tir_cand.sum()), called, checked, exact, near


def copy_recall_precision(R, sample=1500):
    w, f, p = R["w"], R["found"], R["w"]["planted"]
    rng = np.random.default_rng(5)
    order = np.argsort(p["family"], kind="stable")
    fam_sorted = p["family"][order]
    tot_truth = hit_truth = tot_found = ok_found = near_truth = near_hit = 0
    for c in rng.permutation(R["n"])[:sample]:
        fam = int(w["family"][c])
        lo, hi = np.searchsorted(fam_sorted, [fam, fam + 1])
        idx = order[lo:hi]
        a, b = int(f["copy_first"][c]), int(f["copy_first"][c + 1])
        fc, fs, fe, fm = f["contig"][a:b], f["start1"][a:b] - 1, f["end1"][a:b], f["minus"][a:b].astype(bool)
        matched_found = np.zeros(b - a, dtype=bool)
        for i in idx:
            s, e = int(p["start"][i]), int(p["start"][i] + p["length"][i])
            ov = np.minimum(fe, e) - np.maximum(fs, s)
            m = (fc == p["contig"][i]) & (fm == p["minus"][i]) & (ov >= 0.8 * (e - s))
            if p["full"][i]:
                tot_truth += 1
                hit_truth += bool(m.any())
                if w["cand_div"][c] + p["div"][i] <= 0.15:      # pairs within 15 % of each other
                    near_truth += 1
                    near_hit += bool(m.any())
            matched_found |= (fc == p["contig"][i]) & (ov >= 0.5 * np.minimum(fe - fs, e - s))
        tot_found += b - a
        ok_found += int(matched_found.sum())
    return hit_truth / max(1, tot_truth), ok_found / max(1, tot_found), tot_truth, tot_found, near_hit / max(1, near_truth), near_truth


@pytest.fixture(scope="module")
def c2():
    R = run_fine(100, 500, 0, 20250927 + 2, te_types=("tir", "helitron", "non_ltr"))
    yield R
    _release_oracle_files(R)
    R["ctx"].close()


def test_c2_fine_stage_matches_oracle_chain(c2):
    """EVERY candidate of the C2 batch (5 000) re-judged by the oracle chain on the copy table the GPU found"""
    bad, n_te = oracle_check(c2, c2["n"], 1)
    assert c2["n"] == 5000 and bad == [] and n_te >= 3000
    st = c2["align"]
    assert st["dropped"] == 0 and st["pairs"] > 50_000
    assert st["certified"] >= 0.80 * st["pairs"]            # measured r02: 0.89 (exact_cap 8)


def test_c2_anchor_matches_with_an_edit_on_an_end_base_are_rare(c2):
    """The one class of anchor matches where the real `fuzzysearch` package could report another start / end than the definition
    the goldens pin (oracle/stubs.py; Util.py:9173-9174): an edit on the first or last base of a match.  Of the
    alignments the chain judges for 1 500 C2 candidates it stays below 1 % (measured: 5 of 832 on C3's sample, 0.6 %)."""
    anchors = {}
    bad, _n_te = oracle_check(c2, 1500, 7, anchors=anchors)
    total = sum(anchors.values())
    print("C2, anchor matches of %d judged alignments: %s" % (total, anchors))
    assert bad == [] and total >= 1000
    assert anchors.get("end", 0) < 0.01 * total


def test_c2_whole_candidate_intervals(c2):
    """C2 in the other interval mode (HITE_COPY_INTERVAL=whole: the whole-candidate intervals of rounds 2-4) beside the default -- copy
    records in the reference's coordinates (reference_start + 1 .. reference_end, Util.py:8026) with the rows padded by the clipped
    candidate bases (hite_flank_region_align_clip_dev).  The default mode is usable at size: hardly any wide fall-back (8 633 with bare
    aligned windows, round 4), TE calls not behind the whole-candidate mode's; 1 000 random candidates of the whole-candidate run
    re-judged by the oracle chain (test_c2_fine_stage_matches_oracle_chain re-judges all 5 000 of the default run)."""
    from hite_amd import _lib as hl

    te = int((c2["calls"]["is_te"] != 0).sum())
    n_tir, called, checked, exact, near = boundary_stats(c2)
    st = c2["align"]
    print("C2, reference coordinates + padded rows (default): %d copies (%d with a clip); TE calls %d; of %d checked: both ends exact %d, "
          "within 3 bp %d; wide fall-backs %d, dropped %d, certified %d of %d pairs"
          % (len(c2["found"]["contig"]), int((c2["found"]["clip"] != 0).sum()), te, checked, exact, near, st["fallback"], st["dropped"],
             st["certified"], st["pairs"]))
    assert (c2["found"]["clip"] != 0).sum() > 10000
    assert st["fallback"] < 400 and st["dropped"] == 0
    os.environ["HITE_COPY_INTERVAL"] = "whole"
    hl.load().hite_copy_config(-1)
    try:
        R = run_fine(100, 500, 0, c2["seed"])
    finally:
        os.environ.pop("HITE_COPY_INTERVAL", None)
        hl.load().hite_copy_config(-1)
    try:
        te_w = int((R["calls"]["is_te"] != 0).sum())
        _n, _called, checked_w, exact_w, near_w = boundary_stats(R)
        print("C2, whole-candidate intervals: %d copies; TE calls %d; of %d checked: both ends exact %d, within 3 bp %d; wide fall-backs %d"
              % (len(R["found"]["contig"]), te_w, checked_w, exact_w, near_w, R["align"]["fallback"]))
        assert (R["found"]["clip"] != 0).sum() == 0
        assert te >= 0.95 * te_w
        bad, _n = oracle_check(R, 1000, 3)
        assert bad == []
    finally:
        _release_oracle_files(R)
        R["ctx"].close()


def test_c2_wider_bands_never_lower_a_cost(c2):
    """A re-run in a wider band can only lower a pair's cost, and a certified cost is the optimum.  Equal cost sums over all pairs at
    exact_cap 0 / 8 / 16 therefore mean that the 4-word band already had the optimal cost for every pair the 16-word schedule
    certifies (profiles/r04_uncertified_rows_vs_optimum.txt: the certificate is conservative, not the band);
    the calls are the same."""
    import torch
    from hite_amd._lib import CALL_DTYPE
    ctx, w, n = c2["ctx"], c2["w"], c2["n"]
    dev = torch.device("cuda", 0)
    nbytes = int(w["cand_off"][-1])
    d_cand = torch.from_numpy(np.concatenate([w["cands"], np.zeros(64, np.uint8)])).to(dev)
    d_off = torch.from_numpy(np.ascontiguousarray(w["cand_off"])).to(dev)
    d_calls = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    cap = nbytes + 200 * n + 4096
    d_cons = torch.zeros(cap + 64, dtype=torch.uint8, device=dev)
    seen = {8: (c2["align"], c2["calls"])}
    try:
        for exact in (0, 16):
            ctx.align_config(exact)
            nc, p_cf, p_ct, p_s1, p_e1, p_mn, _an = ctx.find_copies_dev(n, d_cand.data_ptr(), d_off.data_ptr(), nbytes)
            ctx.align_stats(reset=True)
            ctx.flank_region_align_dev("tir", 1, n, d_cand.data_ptr(), d_off.data_ptr(), p_cf, nc, p_ct, p_s1, p_e1, p_mn, 50,
                                       d_calls.data_ptr(), d_cons.data_ptr(), cap, d_clip=ctx.copy_clips_dev())
            torch.cuda.synchronize()
            seen[exact] = (ctx.align_stats(), d_calls.cpu().numpy().view(CALL_DTYPE).copy())
    finally:
        ctx.align_config(8)
    st = {k: v[0] for k, v in seen.items()}
    print("cost sums at exact_cap 0 / 8 / 16: %d / %d / %d; certified %d / %d / %d of %d pairs"
          % (st[0]["cost"], st[8]["cost"], st[16]["cost"], st[0]["certified"], st[8]["certified"], st[16]["certified"], st[8]["pairs"]))
    assert st[0]["pairs"] == st[8]["pairs"] == st[16]["pairs"] and st[0]["dropped"] == st[16]["dropped"] == 0
    assert st[0]["certified"] < st[8]["certified"] < st[16]["certified"]
    assert st[0]["cost"] == st[8]["cost"] == st[16]["cost"]
    for exact in (0, 16):
        assert np.array_equal(seen[exact][1]["is_te"], seen[8][1]["is_te"]) and np.array_equal(seen[exact][1]["cons_len"], seen[8][1]["cons_len"])


@pytest.mark.parametrize("te_type", ["helitron", "non_ltr"])
def test_c2_fine_stage_other_types_match_oracle_chain(c2, te_type):
    """the fused pipeline with TE_type Helitron / non-LTR (what judge_Helitron/Non_LTR_transposons.py run) on the C2 batch: the
    same 1 000 random candidates through the oracle chain (the families are TIR elements: mostly rejections, each one compared)"""
    bad, n_te = oracle_check(c2, 1000, 3, te_type)
    assert bad == []


def test_c2_copy_finder_recall_precision(c2):
    recall, precision, n_truth, n_found, near_recall, n_near = copy_recall_precision(c2)
    print("copy finder: recall %.4f of %d planted full-length copies (%.4f of the %d within 15 %% of the candidate), precision %.4f of %d copies found"
          % (recall, n_truth, near_recall, n_near, precision, n_found))
    assert n_truth > 10_000 and n_found > 10_000 and n_near > 3_000
    # round 3 (chains extended base by base to the candidate ends + the reference's two 95 % filters, Util.py:8008-8022):
    # measured 0.923 / 0.838 / 1.000 (profiles/r03_scale_tests.txt); round 2 ("anchor span >= 80 %"): 0.833 / 0.622 / 1.000
    assert near_recall >= 0.90
    assert recall >= 0.75                                   # over all pairs, up to 30 % apart: (w=10, k=15) minimizers lose the far ones
    assert precision >= 0.97


def test_c2_te_calls_have_the_planted_boundaries(c2):
    n_tir_cand, called, checked, exact, near = boundary_stats(c2)
    print("fine stage: %d TIR candidates (boundaries off by up to 30 bp on input), %d judged TE; of %d checked: both ends exact %d, within 3 bp %d"
          % (n_tir_cand, called, checked, exact, near))
    assert called >= 0.58 * n_tir_cand                      # measured r03: 0.661 (r02: 0.585 -- more copies per candidate found)
    assert exact >= 0.45 * checked and near >= 0.72 * checked   # measured r03: 0.535 / 0.816 (r02: 0.48 / 0.74; judge_boundary_v5's own TSD / homology choices)


def test_c2_coarse_stage_recovers_the_families(c2):
    """stage 3.1 on the same genome (all-vs-all seeding + FMEA, GPU): planted families with >= 3 full copies come out as intervals"""
    w, ctx, p = c2["w"], c2["ctx"], c2["w"]["planted"]
    sc, so = ctx.seed_segments(1_000_000)
    (oc, os_, oe), st = ctx.coarse_stage_dev(1_000_000, sc, so, 2000, 30000)
    assert len(oc) > 1000
    key = oc.astype(np.int64) << 40
    order = np.argsort(key + os_)
    oc, os_, oe = oc[order], os_[order], oe[order]
    skey = key[order] + os_
    rec = {}
    for i in np.flatnonzero(p["full"]):
        fam = int(p["family"][i])
        s, e = int(p["start"][i]), int(p["start"][i] + p["length"][i])
        k = (int(p["contig"][i]) << 40)
        lo = np.searchsorted(skey, k + max(0, s - 30000))
        hi = np.searchsorted(skey, k + e)
        hit = False
        for q in range(lo, hi):
            ov = min(int(oe[q]), e) - max(int(os_[q]) - 1, s)
            if ov >= 0.8 * (e - s) and (int(oe[q]) - int(os_[q])) <= 1.5 * (e - s) + 100:
                hit = True
                break
        a, b = rec.get(fam, (0, 0))
        rec[fam] = (a + 1, b + hit)
    multi = [f for f, (a, _b) in rec.items() if a >= 3]
    got = sum(1 for f in multi if rec[f][1] >= 1)
    print("coarse stage: %d intervals; %d of %d families with >= 3 full copies recovered" % (len(oc), got, len(multi)))
    assert len(multi) >= 400 and got >= 0.95 * len(multi)   # measured r02: 489 of 489


def test_c3_fine_stage_matches_oracle_chain():
    R = run_fine(1000, 2500, 2500, 20250927 + 3)
    try:
        bad, n_te = oracle_check(R, 5000, 2)
        assert bad == [] and n_te >= 3000
        n_tir_cand, called, checked, exact, near = boundary_stats(R)
        print("C3: %d TIR candidates, %d judged TE; of %d checked: both ends exact %d, within 3 bp %d; %d TE calls in all" %
              (n_tir_cand, called, checked, exact, near, int((R["calls"]["is_te"] != 0).sum())))
        assert called >= 0.58 * n_tir_cand and exact >= 0.45 * checked and near >= 0.72 * checked    # measured r03: 0.673 / 0.556 / 0.826
        st = R["align"]
        assert st["dropped"] == 0 and st["certified"] >= 0.65 * st["pairs"]   # measured r02: 0.74 (exact_cap 8; 0.90 with 16)
    finally:
        _release_oracle_files(R)
        R["ctx"].close()


def test_c5_two_population_genomes_merge_into_one_library(tmp_path):
    """BASELINE.json configs[4] (panHiTE) at its configured genome size on the one GPU there is: two 300 Mbp genomes drawn from
    a shared family pool (70 % of the families each, as bench.py --config C5 draws them) go through copy finding + the fine
    stage one after the other; their TE libraries are concatenated and merged by deredundant_for_LTR_v5
    (pan_remove_redundancy.py:16-46, Util.py:12202-12337).
      (i)  parity of the merge: the clusters of the WHOLE library are those the same host code finds when every device stage
           is replaced by its CPU twin (tests/oracle_ctx.py), and on a sub-library (the members of 150 random clusters + 100
           unclustered sequences) the two runs write byte-identical .tmp.cons and .cons files;
      (ii) the result is non-redundant: a planted family that reached the merged library comes out as ONE record."""
    _c5_merge(tmp_path, 2, whole_library_twin=True, min_once=0.80)


def test_c5_eight_population_genomes_merge_into_one_library(tmp_path):
    """config C5 at its configured SIZE: eight 300 Mbp genomes one after another on the one GPU, then the real merge of eight
    libraries (~57 000 sequences: deredundant_for_LTR_v5 takes its blocked all-vs-all path, Util.py:12202-12337).  Parity on a
    sub-library (members of 150 random clusters of the whole-library run + 100 unclustered sequences: the twin-driven host code
    writes the same files), and the whole library comes out non-redundant."""
    _c5_merge(tmp_path, 8, whole_library_twin=False, min_once=0.90)       # (measured: 1 390 of 1 496 = 0.929)


def _c5_merge(tmp_path, n_genomes, whole_library_twin, min_once):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hite_amd import util
    from oracle_ctx import OracleCtx

    seed = 20250927 + 5
    names, seqs, fam_of = [], {}, {}
    ctx = None
    for g in range(n_genomes):
        R = run_fine(300, 750, 750, seed + 1009 * (g + 1), family_seed=seed, family_keep=0.7)
        try:
            calls, cons, w = R["calls"], R["cons"], R["w"]
            assert 5_000 < R["n"] < 20_000
            for c in np.flatnonzero(calls["is_te"] != 0):
                r = calls[c]
                nm = "G%d-TE_%d#Unknown" % (g, int(c))
                names.append(nm)
                seqs[nm] = cons[r["cons_off"]:r["cons_off"] + r["cons_len"]].tobytes().decode()
                fam_of[nm] = int(w["family"][c])
        finally:
            if g + 1 < n_genomes:
                R["ctx"].close()
            else:
                ctx = R["ctx"]
            del R
    try:
        assert len(names) > 4_000 * n_genomes
        merged = str(tmp_path / "merged.fa")
        util.store_fasta({n: seqs[n] for n in names}, merged)
        st_gpu, st_cpu = {}, {}
        out = util.deredundant_for_LTR_v5(merged, str(tmp_path), 1, "terminal", 0.95, 0, ctx=ctx, stages=st_gpu)
        final_names, _final = util.read_fasta(merged + ".cons")
        # (i) whole library: hits and clusters through the twins
        twin_in = str(tmp_path / "twin_merged.fa")
        util.store_fasta({n: seqs[n] for n in names}, twin_in)
        # (the twin run of the whole library stops after the clusters: its alignments are compared on the sub-library below)
        class StopAfterClusters(Exception):
            pass

        class ClustersOnly(OracleCtx):
            def star_msa(self, *a, **k):
                raise StopAfterClusters()

        if whole_library_twin:
            try:
                util.deredundant_for_LTR_v5(twin_in, str(tmp_path), 1, "terminal", 0.95, 0, ctx=ClustersOnly(), stages=st_cpu)
            except StopAfterClusters:
                pass
            assert st_cpu["hits"] == st_gpu["hits"] and st_cpu["clusters"] == st_gpu["clusters"]
        clusters = st_gpu["clusters"]
        rng = np.random.default_rng(11)
        pick = [clusters[i] for i in rng.permutation(len(clusters))[:150]]
        in_cluster = set(n for cl in clusters for n in cl)
        loose = [n for n in names if n not in in_cluster]
        sub = [n for cl in pick for n in cl] + [loose[i] for i in rng.permutation(len(loose))[:100]]
        sub_set = set(sub)
        sub = [n for n in names if n in sub_set]          # library order
        outs = []
        for tag, cx in (("gpu", ctx), ("cpu", OracleCtx())):
            path = str(tmp_path / ("sub_%s.fa" % tag))
            util.store_fasta({n: seqs[n] for n in sub}, path)
            util.deredundant_for_LTR_v5(path, str(tmp_path), 1, "terminal", 0.95, 0, ctx=cx)
            outs.append((open(path + ".tmp.cons").read(), open(path + ".cons").read()))
        assert outs[0] == outs[1]
        assert len(util.read_fasta(str(tmp_path / "sub_gpu.fa.cons"))[0]) < len(sub)
        # (ii) non-redundant: records per planted family in the merged library
        per_fam = {}
        for n in final_names:
            per_fam[fam_of[n]] = per_fam.get(fam_of[n], 0) + 1
        fams_in = set(fam_of[n] for n in names)
        once = sum(1 for f in fams_in if per_fam.get(f, 0) == 1)
        lost = sum(1 for f in fams_in if per_fam.get(f, 0) == 0)
        more = sorted((per_fam[f] for f in fams_in if per_fam.get(f, 0) > 1), reverse=True)
        print("C5 merge: %d sequences of %d genomes (%d families) -> %d clusters -> %d records; families with exactly one record %d (%.3f), "
              "with none %d, with more %d (largest %s); %d hits" % (len(names), n_genomes, len(fams_in), len(clusters), len(final_names), once,
                                                                     once / max(1, len(fams_in)), lost, len(more), more[:5], st_gpu["hits"]))
        assert os.path.exists(out) and len(final_names) < 0.3 * len(names)
        assert lost == 0
        assert once >= min_once * len(fams_in)
    finally:
        if ctx is not None:
            ctx.close()


def test_bench_strong_scaling_line_two_ranks_functional():
    """The N > 1 path of bench.py as the driver launches it (--gpus 2: one C2 batch sharded over the ranks by cost, the 32-byte calls
    all-gathered and put back in candidate order, the weak form as a side block), run FUNCTIONALLY on this one-GPU box: the two ranks
    share the GPU and the collectives move host copies over gloo (HITE_BENCH_BACKEND=gloo; RCCL refuses two ranks on one device).
    Checks what a node will not be asked twice: the line is printed, it is the strong form, the merged calls are the per-rank calls,
    and the re-judged sample has no mismatch."""
    import json
    import subprocess

    env = dict(os.environ)
    env["HITE_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C2", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--no-coarse", "--no-modes", "--verify", "16"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["unit"] == "candidates/s" and d["value"] > 0
    assert "not a measurement" in d["data"]
    assert d["config"]["candidates_per_gpu"] == 2500 and d["weak"]["candidates_per_gpu"] == 5000 and d["weak"]["value"] > 0
    assert d["verify"]["mismatches"] == 0 and d["verify"]["checked"] == 16


def _bench_two_ranks(extra, timeout=900):
    import json
    import subprocess

    env = dict(os.environ)
    env["HITE_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "not a measurement" in d["data"]
    return d


def test_bench_c5_line_two_ranks_functional():
    """config C5 as a node runs it -- one population genome per rank, the per-rank libraries all-gathered (padded consensus pools +
    lengths) and merged on rank 0 -- with two ranks sharing this GPU over gloo: the line is printed, both genomes' libraries reach
    the merge, and the re-judged sample of rank 0 has no mismatch (genomes of 30 Mbp: a functional run)"""
    d = _bench_two_ranks(["--config", "C5", "--genome-mbp", "30", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--verify", "16"])
    cfg = d["config"]
    assert d["scaling"] == "weak" and cfg["genomes"] == 2 and cfg["library_sequences_in"] > 600
    assert 0 < cfg["library_sequences_final"] < cfg["library_sequences_in"]
    assert d["verify"]["mismatches"] == 0


def test_bench_coarse_line_two_ranks_functional():
    """stage 3.1's companion line with two ranks (replicas: every rank its own genome, no collective on the data path), the ranks
    sharing this GPU over gloo"""
    d = _bench_two_ranks(["--stage", "coarse", "--genome-mbp", "50", "--steps", "1", "--warmup", "2", "--no-cpu-baseline"])
    assert d["unit"] == "Mbp/s" and d["config"]["hsp_records"] > 0 and d["config"]["repeat_intervals"] > 0
