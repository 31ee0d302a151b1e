#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.json.gz by calling the REFERENCE's
own Python (imported from /root/reference via oracle/ref_harness.py) on seeded synthetic
inputs from tests/casegen.py.  Run in the build container only:

    PYTHONHASHSEED=0 python oracle/gen_golden.py

The fixtures hold inputs and the reference's outputs (data only; no reference source).
fuzzysearch / Levenshtein are the documented restatements of oracle/stubs.py
("parity unpinned" at that third-party boundary, see that file).
"""
import gzip
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import casegen  # noqa: E402
import ref_harness  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def dump(name, obj):
    path = os.path.join(GOLD, name + ".json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(obj, separators=(",", ":"), sort_keys=True).encode())
    print("wrote", path, os.path.getsize(path), "bytes")


def write_fasta(path, names, seqs):
    with open(path, "w") as f:
        for n, s in zip(names, seqs):
            f.write(">" + n + "\n" + s + "\n")


def gen_fmea(U, tmp):
    cases = []
    specs = [
        dict(seed=1, n_seg=2, n_fam=3, noise=5, frag=(1, 1)),
        dict(seed=2, n_seg=3, n_fam=6, noise=20, frag=(1, 4)),
        dict(seed=3, n_seg=2, n_fam=4, noise=10, frag=(2, 5), dup=15),
        dict(seed=4, n_seg=4, n_fam=10, noise=40, frag=(1, 3), chroms=("chrX",)),
        dict(seed=5, n_seg=1, n_fam=5, noise=0, frag=(3, 6), copies=(5, 14)),
        dict(seed=6, n_seg=3, n_fam=8, noise=100, frag=(1, 2)),
    ]
    for sp in specs:
        for skip_gap, max_len in ((2000, 30000), (150, 5000)):
            rows = casegen.make_hsp_table(**sp)
            p = os.path.join(tmp, "fmea_%d_%d.out" % (sp["seed"], skip_gap))
            with open(p, "w") as f:
                f.writelines(casegen.hsp_to_blast6_lines(rows))
            pkl = U.get_longest_repeats_v4(p, skip_gap, max_len, 0)
            res = U.load_from_file(pkl)
            cases.append(dict(rows=rows, skip_gap=skip_gap, max_len=max_len, expected=list(res.keys())))
    # hand-made edge cases: chain break at skip_gap +-1, 10-bp rounding collisions
    q, s = "chr1$0", "chr1$1000000"
    hand = []
    for gap in (1999, 2000, 2001):
        hand.append([(q, s, 100, 400, 5000, 5300), (q, s, 400 + gap, 800 + gap, 5300 + 10, 5700 + 10)])
        hand.append([(q, s, 100, 400, 5000, 5300), (q, s, 410, 800, 5300 + gap, 5700 + gap)])
        hand.append([(q, s, 100, 400, 9000, 8700), (q, s, 410, 800, 8700 - gap, 8300 - gap)])
    hand.append([(q, s, 101, 400, 5001, 5300), (q, "chr2$0", 105, 398, 77001, 77294), (q, "chr2$0", 2105, 2398, 97001, 97294),
                 (q, "chr2$0", 2111, 2391, 197001, 197281)])
    hand.append([(q, s, 100, 100 + 78, 5000, 5078), (q, s, 1000, 1000 + 79, 15000, 15079), (q, s, 3000, 3000 + 80, 25000, 25080)])
    hand.append([(q, q, 100, 900, 20100, 20900), (q, q, 20100, 20900, 100, 900), (q, q, 150, 880, 50150, 50880),
                 (q, q, 100, 900, 100, 900)])
    for rows in hand:
        p = os.path.join(tmp, "fmea_hand.out")
        with open(p, "w") as f:
            f.writelines(casegen.hsp_to_blast6_lines(rows))
        pkl = U.get_longest_repeats_v4(p, 2000, 30000, 0)
        cases.append(dict(rows=rows, skip_gap=2000, max_len=30000, expected=list(U.load_from_file(pkl).keys())))
    dump("fmea", cases)
    # stress case: ~30 k lines; stored as the generator parameters + sha256 of the ordered interval names
    import hashlib
    sp = dict(seed=77, n_seg=4, n_fam=220, noise=1800, frag=(1, 4), copies=(2, 12))
    rows = casegen.make_hsp_table(**sp)
    p = os.path.join(tmp, "fmea_stress.out")
    with open(p, "w") as f:
        f.writelines(casegen.hsp_to_blast6_lines(rows))
    names = list(U.load_from_file(U.get_longest_repeats_v4(p, 2000, 30000, 0)).keys())
    dump("fmea_stress", dict(params=sp, lines=len(rows), intervals=len(names), skip_gap=2000, max_len=30000,
                             sha256=hashlib.sha256("\n".join(names).encode()).hexdigest()))


def run_msa_case(U, tmp, case):
    raw = os.path.join(tmp, "aln.fa")
    write_fasta(raw, case["names"], case["seqs"])
    clean = U.remove_sparse_col_in_align_file(raw)
    cn, cc = U.read_fasta(clean)
    out = dict(case)
    out["clean"] = [cc[n] for n in cn]
    fn = {"tir": U.judge_boundary_v5, "helitron": U.judge_boundary_v6, "non_ltr": U.judge_boundary_v9}[case["te_type"]]
    try:
        is_te, info, cons, rn = fn(case["cand"], clean, 0, case["te_type"], case["plant"], "cons")
        out["expected"] = [bool(is_te), info, cons, int(rn)]
    except Exception as e:  # the reference raises on a few degenerate inputs
        out["expected"] = ["EXC", type(e).__name__]
    return out


def gen_judge(U, tmp):
    for te_type, n, seed0 in (("tir", 60, 11), ("non_ltr", 40, 12), ("helitron", 40, 13)):
        cases = []
        for p in casegen.msa_param_grid(te_type, n, seed0):
            c = casegen.make_msa_case(**p)
            for plant in ((1, 0) if te_type == "tir" and p["seed"] % 3 == 0 else (1,)):
                c2 = dict(c)
                c2["plant"] = plant
                cases.append(run_msa_case(U, tmp, c2))
        # large ones: 101-row cap and wide matrices
        big = casegen.make_msa_case(seed=seed0 * 7, te_type=te_type, rows=110, te_len=900, div=0.1, ins_cols=10,
                                    trunc_rows=4, shift_l=6, shift_r=-4, tsd_len=8, tsd_frac=0.9)
        cases.append(run_msa_case(U, tmp, big))
        # round 3: cases aimed at the rare exits -- positives with plant 0 and 1, 'nb', 'fl1', inputs the reference raises on,
        # the 100-row cap (appended: the cases above keep their indices)
        for c in casegen.msa_outcome_cases(te_type, seed0 + 10):
            cases.append(run_msa_case(U, tmp, c))
        stats = {}
        for c in cases:
            k = str(c["expected"][:2])
            stats[k] = stats.get(k, 0) + 1
        print(te_type, "outcomes:", stats)
        dump("judge_" + te_type, cases)


def gen_judge_edge(U, tmp):
    """round 4: the lines of the judges and of the boundary searches that tools/ref_line_coverage.py showed unreached"""
    cases = [run_msa_case(U, tmp, c) for c in casegen.msa_edge_cases(61)]
    stats = {}
    for c in cases:
        k = c["te_type"] + " " + str(c["expected"][:2])
        stats[k] = stats.get(k, 0) + 1
    print("judge_edge outcomes:", stats)
    dump("judge_edge", cases)
    # the searches called directly next to the alignment's edges and on alignments too narrow for a window
    rng = np.random.default_rng(62)
    out = []
    for ci, c in enumerate(cases):
        homolog = c["aim"].startswith("homologous flanks")
        if ci % 3 and not homolog:
            continue
        mat = [list(s) for s in c["clean"]]
        R, C = len(mat), len(mat[0])
        if C < 2:
            continue
        thr = 0.95 if R <= 2 else (0.9 if R <= 5 else float(rng.choice([0.7, 0.8])))
        for side in ("start", "end"):
            near = [0, 3, 9, 10, 11, C - 12, C - 11, C - 10, C - 4, C - 1, int(rng.integers(0, C))]
            if homolog:      # homology that runs to the alignment's edge: a new boundary within 10 columns of it cannot be judged
                near += list(range(12, 24, 2)) + list(range(C - 24, C - 12, 2))
            for pos in sorted(set(near)):
                if pos < 0 or pos >= C:
                    continue
                v3 = U.search_boundary_homo_v3(int(R / 2), pos, mat, R, C, side, thr, 0, 20, 10)
                v4 = U.search_boundary_homo_v4(int(R / 2), pos, mat, R, C, side, thr, thr - 0.05, thr, 0, 20, 10)
                out.append(dict(seqs=c["clean"], pos=pos, side=side, thr=thr, v3=int(v3), v4=[bool(v4[0]), int(v4[1])]))
    # alignments of 5 .. 30 columns
    for W in (5, 9, 10, 12, 19, 20, 25, 30):
        for R in (2, 4, 9):
            rows = []
            base = casegen.rand_seq(rng, W)
            for r in range(R):
                rows.append(casegen.mutate(rng, base, 0.1 if r else 0.0))
            mat = [list(s) for s in rows]
            thr = 0.95 if R <= 2 else (0.9 if R <= 5 else 0.7)
            for side in ("start", "end"):
                for pos in sorted(set([0, W // 2, W - 1])):
                    v3 = U.search_boundary_homo_v3(int(R / 2), pos, mat, R, W, side, thr, 0, 20, 10)
                    v4 = U.search_boundary_homo_v4(int(R / 2), pos, mat, R, W, side, thr, thr - 0.05, thr, 0, 20, 10)
                    out.append(dict(seqs=rows, pos=pos, side=side, thr=thr, v3=int(v3), v4=[bool(v4[0]), int(v4[1])]))
    print("boundary_search_edge:", len(out), "cases; v3 found", sum(c["v3"] != -1 for c in out), "; v4 valid", sum(c["v4"][0] for c in out))
    dump("boundary_search_edge", out)


def gen_boundary_search(U):
    """search_boundary_homo_v3 / v4 and calculate_window_homology called directly."""
    rng = np.random.default_rng(77)
    cases = []
    for i, p in enumerate(casegen.msa_param_grid("tir", 40, 21)):
        c = casegen.make_msa_case(**p)
        mat = [list(s) for s in c["seqs"]]
        R, C = len(mat), len(mat[0])
        thr = 0.95 if R <= 2 else (0.9 if R <= 5 else float(rng.choice([0.7, 0.8])))
        for side in ("start", "end"):
            base = 50 if side == "start" else C - 51
            pos = int(np.clip(base + int(rng.integers(-30, 31)), 0, C - 1))
            v3 = U.search_boundary_homo_v3(int(R / 2), pos, mat, R, C, side, thr, 0, 20, 10)
            v4 = U.search_boundary_homo_v4(int(R / 2), pos, mat, R, C, side, thr, thr - 0.05, thr, 0, 20, 10)
            cases.append(dict(seqs=c["seqs"], pos=pos, side=side, thr=thr, v3=int(v3), v4=[bool(v4[0]), int(v4[1])]))
    dump("boundary_search", cases)

    # threshold ties: R=10 / 20 rows with columns at exactly 6/10, 7/10, 8/10, 17/20
    ties = []
    for R, k, thr in ((10, 6, 0.7), (10, 7, 0.7), (10, 7, 0.8), (10, 8, 0.8), (20, 17, 0.95), (20, 14, 0.8),
                      (20, 14, 0.7), (10, 9, 0.9), (10, 8, 0.9), (3, 2, 0.7)):
        W = 12
        rows = []
        for r in range(R):
            rows.append("".join("A" if r < k else "CGT"[(r + c) % 3] for c in range(W)))
        mat = [list(s) for s in rows]
        res = U.calculate_window_homology(mat, list(range(0, W)), thr)
        res2 = U.calculate_window_homology(mat, list(range(W - 1, 1, -1)), thr)
        ties.append(dict(seqs=rows, thr=thr, fwd=int(res), rev=int(res2)))
    dump("thr_ties", ties)

    # round 3: the same binary64 ties reached THROUGH the searches (so that the GPU path, which has no entry for a single
    # window, is pinned on them as well): 30 unrelated columns, then columns whose majority is exactly k/R, (k-1)/R or
    # (k+1)/R of the rows; window means of 20 / 10 such ratios against thr, single ratios against thr - 0.1
    rng = np.random.default_rng(78)
    ts = []
    for R, k, thr in ((10, 6, 0.7), (10, 7, 0.7), (10, 7, 0.8), (10, 8, 0.8), (20, 19, 0.95), (20, 14, 0.8), (20, 14, 0.7),
                      (10, 9, 0.9), (10, 8, 0.9), (5, 4, 0.9), (4, 3, 0.9), (20, 16, 0.8), (20, 12, 0.7), (30, 21, 0.7),
                      (30, 24, 0.8), (50, 35, 0.7), (100, 70, 0.7), (100, 80, 0.8), (101, 71, 0.7), (7, 5, 0.7)):
        for mix in (0, 1, 2):
            C = 120
            cols = []
            for c in range(C):
                if c < 30 or c >= C - 30:
                    cols.append([casegen.BASES[int(x)] for x in rng.integers(0, 4, size=R)])
                else:
                    kk = k if mix == 0 else int(k + rng.choice([-1, 0, 0, 1] if mix == 1 else [-1, 0, 1]))
                    kk = max(1, min(R, kk))
                    cols.append(["A" if r < kk else "CGT"[(r + c) % 3] for r in range(R)])
            rows = ["".join(cols[c][r] for c in range(C)) for r in range(R)]
            mat = [list(x) for x in rows]
            for side, pos in (("start", 20), ("end", C - 21), ("start", 35), ("end", C - 36)):
                v3 = U.search_boundary_homo_v3(int(R / 2), pos, mat, R, C, side, thr, 0, 20, 10)
                v4 = U.search_boundary_homo_v4(int(R / 2), pos, mat, R, C, side, thr, thr - 0.05, thr, 0, 20, 10)
                ts.append(dict(seqs=rows, pos=pos, side=side, thr=thr, v3=int(v3), v4=[bool(v4[0]), int(v4[1])]))
    print("thr_ties_search: v3 found", sum(c["v3"] != -1 for c in ts), "of", len(ts), "; v4 valid", sum(c["v4"][0] for c in ts))
    dump("thr_ties_search", ts)


def gen_tsd(U):
    rng = np.random.default_rng(5)
    cases = []
    for i in range(300):
        k = int(rng.choice([0, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12]))
        L = int(rng.integers(30, 90))
        core = casegen.rand_seq(rng, L)
        mode = rng.random()
        if k == 2 and mode < 0.5:
            tsd = "TA"
        elif k == 2:
            tsd = casegen.rand_seq(rng, 2)
            core = "CCC" + core[3:-3] + "GGG"
        elif k == 3:
            tsd = str(rng.choice(["TAA", "TTA", casegen.rand_seq(rng, 3)]))
            if mode < 0.4:
                core = "CACTA" + core[5:-5] + "TAGTG"
            elif mode < 0.6:
                core = "CACTG" + core[5:-5] + "CAGTG"
        elif k == 4:
            tsd = str(rng.choice(["TTAA", casegen.rand_seq(rng, 4)]))
        else:
            tsd = casegen.rand_seq(rng, k)
        rt = tsd
        if k >= 8 and mode < 0.3:
            j = int(rng.integers(0, k))
            rt = tsd[:j] + casegen.BASES[(casegen.BASES.index(tsd[j]) + 1) % 4] + tsd[j + 1:]
        left = casegen.rand_seq(rng, int(rng.integers(0, 15))) + tsd
        right = rt + casegen.rand_seq(rng, int(rng.integers(0, 15)))
        s = list(left + core + right)
        # sprinkle gaps
        for _ in range(int(rng.integers(0, 8))):
            p = int(rng.integers(0, len(s) + 1))
            s[p:p] = ["-"] * int(rng.integers(1, 4))
        s = "".join(s)
        # boundary columns = first / last base of core in gapped coordinates
        ung = -1
        bs = be = None
        for col, ch in enumerate(s):
            if ch != "-":
                ung += 1
                if ung == len(left):
                    bs = col
                if ung == len(left) + L - 1:
                    be = col
        bs += int(rng.choice([0, 0, 0, 1, -1]))
        be += int(rng.choice([0, 0, 0, 1, -1]))
        bs = max(0, bs)
        be = min(len(s) - 1, be)
        for plant in (0, 1):
            l, r = U.TSDsearch_v5(s, bs, be, plant)
            cases.append(dict(seq=s, start=bs, end=be, plant=plant, left=l, right=r))
    dump("tsd_search", cases)


def gen_tir_kmer(U):
    cases = []
    i = 0
    for te_len in (120, 300, 1500):
        for tsd_len in (2, 3, 4, 5, 6, 8, 9, 10, 11):
            for off_l, off_r in ((0, 0), (7, -3), (-20, 25), (40, 40)):
                i += 1
                seq, flank = casegen.make_tir_candidate(1000 + i, te_len=te_len, tsd_len=tsd_len, off_l=off_l,
                                                        off_r=off_r, with_n=(i % 7 == 0))
                for plant in (0, 1):
                    res = U.search_confident_tir_v4(seq, flank + 1, len(seq) - flank, flank, "q%d" % i, plant)
                    # canonical order: the reference's order among equal distances depends on
                    # PYTHONHASHSEED, and its -C_{i} index is the rank in that order; fixtures keep the multiset
                    # {(tsd, distance, sequence)} sorted canonically.
                    items = []
                    for name, s in res.items():
                        parts = name.split("-")
                        tsd = [p for p in parts if p.startswith("tsd_")][0][4:]
                        dist = int([p for p in parts if p.startswith("distance_")][0][9:])
                        items.append([dist, tsd, s])
                    items.sort()
                    cases.append(dict(seq=seq, flank=flank, plant=plant, name="q%d" % i, n=len(res), items=items))
    dump("tir_kmer", cases)

    # round 4 (tools/ref_line_coverage.py): candidates shorter than two flanks (the right window starts before the sequence, TSD
    # positions beyond its end), N N as a "TSD" on both sides, elements that start with TATATATA / ATATATAT
    rng = np.random.default_rng(4100)
    edge = []

    def run(seq, tag):
        for plant in (0, 1):
            res = U.search_confident_tir_v4(seq, 51, len(seq) - 50, 50, tag, plant)
            items = []
            for name, s in res.items():
                parts = name.split("-")
                tsd = [p for p in parts if p.startswith("tsd_")][0][4:]
                dist = int([p for p in parts if p.startswith("distance_")][0][9:])
                items.append([dist, tsd, s])
            items.sort()
            edge.append(dict(seq=seq, flank=50, plant=plant, name=tag, n=len(res), items=items))

    for L in (51, 52, 60, 75, 99, 100, 101, 102, 103, 110, 125, 150):      # (a flanked candidate is never shorter than one flank + 1)
        for rep in range(2):
            tsd = casegen.rand_seq(rng, int(rng.choice([2, 3, 4, 5, 8])))
            core = casegen.rand_seq(rng, max(1, L // 3))
            s = casegen.rand_seq(rng, L)
            a = max(0, L // 3 - len(tsd))
            s = (s[:a] + tsd + core + tsd + s)[:L]
            run(s, "short%d_%d" % (L, rep))
    for i in range(12):
        seq, _f = casegen.make_tir_candidate(4200 + i, te_len=200, tsd_len=[2, 3, 4, 8][i % 4], off_l=0, off_r=0)
        a = 50 - 2 - (i % 5)
        b = len(seq) - 50 + (i % 4)
        seq = seq[:a] + "NN" + seq[a + 2:b] + "NN" + seq[b + 2:]
        if i % 3 == 0:      # a longer run on both sides: NNN, NNNN as k-mers too
            seq = seq[:a - 2] + "NNNN" + seq[a + 2:b] + "NNNN" + seq[b + 4:]
        run(seq, "nn%d" % i)
    for i in range(12):
        seq, _f = casegen.make_tir_candidate(4300 + i, te_len=180, tsd_len=[2, 3, 4, 5, 8, 9][i % 6], off_l=0, off_r=0)
        head = "TATATATA" if i % 2 == 0 else "ATATATAT"
        seq = seq[:50] + head + seq[58:]        # (make_tir_candidate cuts the sequence so that the element starts at column 50)
        run(seq, "tata%d" % i)
    print("tir_kmer_edge:", len(edge), "cases,", sum(c["n"] > 0 for c in edge), "with TSDs")
    dump("tir_kmer_edge", edge)


def gen_gather(U, tmp):
    cases = []
    for seed in (1, 2, 3):
        names, seqs = casegen.make_genome(seed)
        ref = os.path.join(tmp, "genome_%d.fa" % seed)
        write_fasta(ref, names, seqs)
        copies = casegen.make_copies(seed, names, seqs)
        cand = os.path.join(tmp, "cand_%d.fa" % seed)
        write_fasta(cand, list(copies.keys()), ["ACGT" * 30 for _ in copies])
        captured = {}

        def fake_copies(query_path, reference, temp_dir, max_copy_num, threads, _c=copies):
            os.makedirs(temp_dir, exist_ok=True)
            return _c

        def fake_members(task, temp_dir, subset_script_path, plant, TE_type, debug, result_type, _cap=captured):
            (query_name, cur_seq, trunc_member_file, extend_member_file) = task
            ent = {}
            n, c = U.read_fasta(extend_member_file)
            ent["extend"] = [[x, c[x]] for x in n]
            if trunc_member_file is not None:
                n, c = U.read_fasta(trunc_member_file)
                ent["trunc"] = [[x, c[x]] for x in n]
            else:
                ent["trunc"] = None
            _cap[query_name] = ent
            return (None, None, "", 0, extend_member_file)

        saved = (U.get_full_length_copies_minimap2, U.run_find_members_v8, U.ProcessPoolExecutor, U.as_completed)
        U.get_full_length_copies_minimap2 = fake_copies
        U.run_find_members_v8 = fake_members
        U.ProcessPoolExecutor = ref_harness.SyncExecutor
        U.as_completed = lambda fs: fs
        try:
            log = type("L", (), {"logger": type("LL", (), {"info": staticmethod(lambda *a: None)})})()
            U.flank_region_align_v5(cand, os.path.join(tmp, "real.fa"), 50, ref, None, "other", tmp, 1, 0, log,
                                    "", 1, 0, 0, os.path.join(tmp, "low.fa"))
        finally:
            (U.get_full_length_copies_minimap2, U.run_find_members_v8, U.ProcessPoolExecutor, U.as_completed) = saved
        cases.append(dict(names=names, seqs=seqs, flank=50,
                          copies={k: [list(t) for t in v] for k, v in copies.items()}, expected=captured))

        # flanking_seq (a-6)
        rng = np.random.default_rng(seed)
        rep_names = []
        for _ in range(40):
            ci = int(rng.integers(0, len(names)))
            n = len(seqs[ci])
            L = int(rng.integers(80, 3000))
            st = int(rng.choice([0, 5, 49, 50, 51, n - L - 60, n - L - 50, n - L - 10, n - L, int(rng.integers(0, n - L))]))
            st = max(0, st)
            rep_names.append("%s:%d-%d" % (names[ci], st, st + L))
        rep_names = list(dict.fromkeys(rep_names))
        lr = os.path.join(tmp, "lr.fa")
        write_fasta(lr, rep_names, [seqs[names.index(x.split(":")[0])][int(x.split(":")[1].split("-")[0]):int(x.split("-")[1])] for x in rep_names])
        out = os.path.join(tmp, "lr.flanked.fa")
        U.flanking_seq(lr, out, ref, 50)
        n, c = U.read_fasta(out)
        cases[-1]["flanking_in"] = rep_names
        cases[-1]["flanking_out"] = [[x, c[x]] for x in n]
    dump("gather", cases)


def gen_tails(U):
    rng = np.random.default_rng(9)
    cases = []
    for i in range(200):
        L = int(rng.integers(5, 80))
        s = casegen.rand_seq(rng, L)
        m = rng.random()
        if m < 0.3:
            s = s + "A" * int(rng.integers(4, 12)) + casegen.rand_seq(rng, int(rng.integers(0, 6)))
        elif m < 0.6:
            u = casegen.rand_seq(rng, int(rng.integers(2, 7)))
            s = s + u * int(rng.integers(3, 7)) + casegen.rand_seq(rng, int(rng.integers(0, 4)))
        a = U.find_tail_polyA(s)
        b = U.find_longest_tandem_repeat_tail(s)
        cases.append(dict(seq=s, polyA=int(a[0]), tandem=int(b[0])))
    dump("tails", cases)


def gen_host(U):
    """host-side glue of the stage wrappers: get_short_tir_contigs (Util.py:7297), filter_dup_itr_v3 (:2791),
    split_and_store_sequences grouping (:4987), getReverseSequence (:1635)"""
    rng = np.random.default_rng(909)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    short_cases = []
    for plant in (0, 1):
        contigs = {}
        for q in range(60):
            body = casegen.rand_seq(rng, int(rng.integers(40, 300)))
            kind = int(rng.integers(0, 6))
            head = casegen.rand_seq(rng, 5)
            if kind == 0:
                head = "CACTA" if rng.random() < 0.5 else "CACTG"
            if kind == 1:
                head = "CCC" + casegen.rand_seq(rng, 2)
            tail = "".join(comp[c] for c in reversed(head)) if kind != 5 else casegen.rand_seq(rng, 5)
            if kind == 4:
                body = body + casegen.rand_seq(rng, 4000)   # hAT length limit
            if kind == 3:
                tail = casegen.rand_seq(rng, 2) + "GGG"      # CCC ... GGG with different inner bases
                head = "CCC" + head[3:]
            seq = head + body + tail
            if rng.random() < 0.1:
                seq = seq[:2] + "N" + seq[3:]
            tsd = casegen.rand_seq(rng, int(rng.choice([2, 3, 4, 8, 9, 10, 11, 12])))
            contigs["chr1:%d-%d-C_%d-tsd_%s-distance_%d" % (q * 100, q * 100 + len(seq), q % 7, tsd, int(rng.integers(0, 40)))] = seq
        # round 4 (Util.py:7329): CACTA / CACTG ends with a 3-base TSD (kept for plants only), and the same ends with a 2-base one
        for q, (head, tsd) in enumerate((("CACTA", "TAA"), ("CACTG", "GCA"), ("CACTA", "TA"), ("CACTG", "TTAA"))):
            seq = head + "ACGGTCATTG" * 9 + "".join(comp[c] for c in reversed(head))
            contigs["chr2:%d-%d-C_0-tsd_%s-distance_3" % (q * 100, q * 100 + len(seq), tsd)] = seq
        got = U.get_short_tir_contigs(dict(contigs), plant)
        short_cases.append({"plant": plant, "names": list(contigs.keys()), "seqs": list(contigs.values()), "kept": list(got.keys())})
    dup_cases = []
    for q in range(40):
        n = int(rng.integers(1, 8))
        names, seqs, lens = [], [], {}
        for i in range(n):
            nm = "chr2:%d-%d-C_%d-tsd_%s-distance_%d" % (q, q + 500, i, casegen.rand_seq(rng, int(rng.integers(2, 12))), int(rng.integers(0, 6)))
            names.append(nm)
            seqs.append(casegen.rand_seq(rng, int(rng.choice([120, 400, 29999, 30000], p=[0.48, 0.48, 0.02, 0.02]))))
            if rng.random() < 0.6:
                lens[nm] = int(rng.integers(5, 40))
        res = U.filter_dup_itr_v3(dict(zip(names, seqs)), dict(lens))
        dup_cases.append({"names": names, "seqs": seqs, "tir_len": lens, "out_names": list(res.keys()), "out_seqs": list(res.values())})
    split_cases = []
    for thr in (100, 1000, 5000):
        names = ["s%d" % i for i in range(int(rng.integers(1, 30)))]
        lens = [int(rng.integers(1, 1500)) for _ in names]
        with tempfile.TemporaryDirectory() as d:
            files = U.split_and_store_sequences(names, {n: "A" * l for n, l in zip(names, lens)}, d, thr)
            groups = [U.read_fasta(f[0])[0] for f in files]
        split_cases.append({"names": names, "lens": lens, "thr": thr, "groups": groups})
    # round 4 (Util.py:5008-5011): sequences left over after the last full file, an exact fit, one sequence below the threshold
    for names, lens, thr in ((["a", "b", "c"], [400, 700, 50], 1000), (["a", "b", "c", "d"], [500, 500, 999, 1], 1000), (["only"], [12], 1000),
                             (["a", "b", "c", "d", "e"], [10, 20, 30, 40, 50], 10_000)):
        with tempfile.TemporaryDirectory() as d:
            files = U.split_and_store_sequences(names, {n: "A" * l for n, l in zip(names, lens)}, d, thr)
            groups = [U.read_fasta(f[0])[0] for f in files]
        split_cases.append({"names": names, "lens": lens, "thr": thr, "groups": groups})
    # on-disk format helpers: rename_fasta (:7500), rename_reference (:7517), lib_add_prefix (:11559),
    # file_exist (:2831), update_prev_TE (:6378) -- file text in, file text out
    fmt = {}
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "in.fa")
        text = ">a#DNA/hAT extra words\nACGTacgtNN\nGGGG\n>b\nTTTT\n>c#x#LTR/Gypsy\nCCCC\n>empty\n>d\tdescr\nAAAA\n"
        open(src, "w").write(text)
        out = os.path.join(d, "out.fa")
        U.rename_fasta(src, out, "TIR_0")
        fmt["rename_fasta"] = {"in": text, "header": "TIR_0", "out": open(out).read()}
        ref_out, cmap = os.path.join(d, "ref.fa"), os.path.join(d, "map.txt")
        U.rename_reference(src, ref_out, cmap)
        fmt["rename_reference"] = {"in": text, "out": open(ref_out).read(), "map": open(cmap).read()}
        lib = os.path.join(d, "lib.fa")
        open(lib, "w").write(text)
        U.lib_add_prefix(lib, "genomeA")
        fmt["lib_add_prefix"] = {"in": text, "prefix": "genomeA", "out": open(lib).read()}
        fe = []
        for name, body in (("x.fa", ">a\nACGT\n"), ("y.fa", ">a\n"), ("z.fa", ""), ("t.txt", "# only a comment\n\n"), ("u.txt", "#c\nline\n"),
                           ("v.list", "   \n")):
            pth = os.path.join(d, name)
            open(pth, "w").write(body)
            fe.append([name, body, bool(U.file_exist(pth))])
        os.makedirs(os.path.join(d, "emptydir")); os.makedirs(os.path.join(d, "fulldir")); open(os.path.join(d, "fulldir", "f"), "w").write("1")
        fmt["file_exist"] = {"files": fe, "emptydir": bool(U.file_exist(os.path.join(d, "emptydir"))),
                             "fulldir": bool(U.file_exist(os.path.join(d, "fulldir"))), "missing": bool(U.file_exist(os.path.join(d, "nope")))}
        prev, cur = os.path.join(d, "prev_TE.fa"), os.path.join(d, "cur.fa")
        open(prev, "w").write(">p\nAAAA\n")
        open(cur, "w").write(">q\nCCCC")
        U.update_prev_TE(prev, cur)
        U.update_prev_TE(prev, os.path.join(d, "absent.fa"))
        fmt["update_prev_TE"] = {"prev": ">p\nAAAA\n", "cur": ">q\nCCCC", "out": open(prev).read()}
    rev = [casegen.rand_seq(rng, 30) for _ in range(5)] + ["ACGTNRYacgt-", ""]
    dump("host_formats", fmt)
    # round 6: read_fasta itself (:1650: headers without sequence, blank sequence lines, text before the first header, a name twice)
    # and the block grouping of split_genome_chunks.py (split_chromosomes :10252, split_dict_into_blocks :10276)
    fa_cases = []
    with tempfile.TemporaryDirectory() as d:
        for i, text in enumerate([">a desc\nacgt\nNN\n>b\t1\n\n>c\n>d\nTT\n", "junk\n>x\nAC\n>x\nGG\n", "", ">only\n", "\n\n>z \nac gt\n",
                                  ">e\n \n>f\nA\n", ">g#DNA/x#TIR  w\nAC\n\nGT\n"]):
            pth = os.path.join(d, "r%d.fa" % i)
            open(pth, "w").write(text)
            names, contigs = U.read_fasta(pth)
            fa_cases.append({"text": text, "names": names, "contigs": contigs, "exists": bool(U.file_exist(pth))})
        names, contigs = U.read_fasta(os.path.join(d, "absent.fa"))
        fa_cases.append({"text": None, "names": names, "contigs": contigs, "exists": False})
    rng2 = np.random.default_rng(4242)
    blk_cases = []
    for _ in range(40):
        n = int(rng2.integers(0, 12))
        lens = [int(rng2.integers(0, 50)) for _ in range(n)]
        threads, chunk = int(rng2.integers(1, 7)), int(rng2.choice([7, 20, 1000]))
        cd = {"c%d" % i: "A" * l for i, l in enumerate(lens)}
        parts = U.split_chromosomes(dict(cd), chunk)
        blocks = U.split_dict_into_blocks(dict(cd), threads, chunk)
        blk_cases.append({"lens": lens, "threads": threads, "chunk": chunk, "parts": [[k, len(v)] for k, v in parts.items()],
                          "blocks": [[[k, len(v)] for k, v in b.items()] for b in blocks]})
    dump("host_fasta", {"read_fasta": fa_cases, "blocks": blk_cases})
    dump("host_glue", {"short_tir": short_cases, "filter_dup": dup_cases, "split": split_cases,
                       "revcomp": [[s, U.getReverseSequence(s)] for s in rev]})


def gen_ltr_frame(tmp):
    """FiLTR flank-frame voting (bin/FiLTR-main/src/Util.py:9175 judge_right_frame_LTR, :9327 judge_left_frame_LTR):
    matrix file rows 'left_frame\tright_frame' -> (is_ltr, new_boundary) for both sides"""
    F = ref_harness.load_filtr_util()
    rng = np.random.default_rng(4242)
    cases = []
    for ci in range(140):
        R = int(rng.choice([1, 2, 3, 5, 6, 10, 11, 24, 51, 80]))
        flank = int(rng.choice([30, 50, 100]))
        # the homologous region reaches `hl` columns into the left frame (from its right end) and `hr` into the right frame
        hl = int(rng.choice([0, 0, 3, 6, 15, 25, flank]))
        hr = int(rng.choice([0, 0, 5, 19, 22, 30, flank]))
        div = float(rng.choice([0.0, 0.05, 0.12, 0.3]))
        cons_l, cons_r = casegen.rand_seq(rng, flank), casegen.rand_seq(rng, flank)
        rows = []
        for r in range(R):
            L = list(casegen.rand_seq(rng, flank))
            Rr = list(casegen.rand_seq(rng, flank))
            for k in range(hl):
                if rng.random() >= div:
                    L[flank - 1 - k] = cons_l[flank - 1 - k]
            for k in range(hr):
                if rng.random() >= div:
                    Rr[k] = cons_r[k]
            L, Rr = "".join(L), "".join(Rr)
            x = rng.random()
            if x < 0.08:
                L = "-" * flank
            elif x < 0.16:
                Rr = "-" * flank
            elif x < 0.22:
                cut = int(rng.integers(1, flank))
                L = "-" * cut + L[cut:]
                Rr = Rr[:flank - cut] + "-" * cut
            elif x < 0.26:
                L = L[:5] + "N" + L[6:]
            rows.append((L, Rr))
        path = os.path.join(tmp, "m%d.matrix" % ci)
        with open(path, "w") as f:
            for L, Rr in rows:
                f.write(L + "\t" + Rr + "\n")
        win = int(rng.choice([20, 20, 20, 10]))
        lt = F.judge_left_frame_LTR(path, flank, sliding_window_size=win)
        rt = F.judge_right_frame_LTR(path, flank, sliding_window_size=win)
        cases.append({"left": [r[0] for r in rows], "right": [r[1] for r in rows], "flank": flank, "window": win,
                      "left_out": [bool(lt[0]), int(lt[1])], "right_out": [bool(rt[0]), int(rt[1])]})
    dump("ltr_frame", cases)


def gen_ltr_both_ends(tmp):
    """FiLTR get_both_ends_frame (bin/FiLTR-main/src/Util.py:1401): aligned copies + terminal sequence -> `.matrix` frames and
    full-length rows (anchor search with the harness' find_near_matches, its own sparse-column rule)"""
    F = ref_harness.load_filtr_util()
    rng = np.random.default_rng(1401)
    cases = []
    for ci in range(70):
        R = int(rng.choice([1, 2, 3, 4, 7, 12, 30]))
        L = int(rng.choice([45, 120, 400, 900]))
        flank = int(rng.choice([20, 50, 100]))
        outer = int(rng.choice([0, 10, flank, flank + 30]))
        elem = casegen.rand_seq(rng, L)
        consensus = casegen.rand_seq(rng, outer) + elem + casegen.rand_seq(rng, outer)
        # column plan: every consensus position is a column; sparse insertion columns are sprinkled in
        plan = []
        for i in range(len(consensus)):
            if rng.random() < 0.06:
                plan.extend([("ins", None)] * int(rng.integers(1, 4)))
            plan.append(("pos", i))
        kind = ci % 10
        rows = []
        for r in range(R):
            lo = 0 if rng.random() < 0.7 else int(rng.integers(0, outer + 25))
            hi = len(consensus) if rng.random() < 0.7 else len(consensus) - int(rng.integers(0, outer + 25))
            div = float(rng.choice([0.0, 0.02, 0.08]))
            if kind == 3 and r == 0:
                div = 0.5                                         # the first row loses its anchors: the second one decides
            if kind == 7:
                div = 0.6                                         # nobody has the anchors
            row = []
            for (k, i) in plan:
                if k == "ins":
                    row.append(casegen.rand_seq(rng, 1) if rng.random() < (0.15 if kind != 5 else 0.55) else "-")
                elif i < lo or i >= hi or rng.random() < 0.04:
                    row.append("-")
                else:
                    row.append(consensus[i] if rng.random() >= div else casegen.rand_seq(rng, 1))
            rows.append("".join(row))
        cur = elem if kind != 2 else casegen.mutate(rng, elem, 0.01)
        af = os.path.join(tmp, "be_%d.maf.fa" % ci)
        with open(af, "w") as fh:
            for r, row in enumerate(rows):
                fh.write(">copy%d\n%s\n" % (r, row))
        od, fd = os.path.join(tmp, "be_out_%d" % ci), os.path.join(tmp, "be_full_%d" % ci)
        os.makedirs(od, exist_ok=True); os.makedirs(fd, exist_ok=True)
        m1, m2 = F.get_both_ends_frame("q", cur, af, od, fd, flank, 0)
        if m1 is None:
            cases.append({"rows": rows, "cur": cur, "flank": flank, "frames": None, "full": None})
        else:
            frames = [ln.rstrip("\n").split("\t") for ln in open(m1)]
            full = [ln.rstrip("\n") for ln in open(m2)]
            cases.append({"rows": rows, "cur": cur, "flank": flank, "frames": frames, "full": full})
    dump("ltr_both_ends", cases)


def gen_nonltr_prep(U):
    """search_polyA_TSD (Util.py:10915): flanked repeat -> (found_TSD, TSD_seq, non_ltr_seq)"""
    rng = np.random.default_rng(515)
    cases = []
    for ci in range(260):
        flank = 50
        L = int(rng.choice([60, 150, 400, 900, 3000]))
        body = casegen.rand_seq(rng, L)
        kind = int(rng.integers(0, 8))
        tsd = casegen.rand_seq(rng, int(rng.integers(8, 21)))
        left, right = casegen.rand_seq(rng, flank), casegen.rand_seq(rng, flank)
        if kind in (0, 1, 2):                      # polyA tail at the 3' end, TSD on both sides
            tail = "A" * int(rng.integers(5, 18)) if kind != 2 else "".join(["CA", "TTG", "GAAT"][int(rng.integers(0, 3))] for _ in range(6))
            t2 = tsd if rng.random() < 0.7 else casegen.mutate(rng, tsd, 0.08)
            left = left[:flank - len(tsd) - int(rng.integers(0, 6))] + tsd
            left = (casegen.rand_seq(rng, flank) + left)[-flank:]
            seq = left + body + tail + t2 + right
            seq = seq[:flank + L + len(tail) + flank]
        elif kind in (3, 4):                       # polyT head (minus strand element)
            head = "T" * int(rng.integers(5, 18))
            seq = (left + tsd)[-flank:] + head + body + tsd + right
            seq = seq[:len(seq) - len(tsd)] if rng.random() < 0.3 else seq
        elif kind == 5:
            seq = left + body + right
        elif kind == 6:                            # short sequence: windows clamp at both ends
            seq = casegen.rand_seq(rng, int(rng.integers(20, 90)))[:40] + "A" * 9 + casegen.rand_seq(rng, 12)
        else:
            seq = left + body[:L // 2] + "N" * 3 + "AAAAAAAA" + body[L // 2:] + "AAAAAAAAAA" + tsd[:4] + "N" + tsd[5:] + right
        found, tsd_seq, nl = U.search_polyA_TSD(seq, flank, 25, list(range(8, 21)))
        cases.append({"seq": seq, "flank": flank, "found": bool(found), "tsd": tsd_seq, "non_ltr": nl})
    dump("nonltr_prep", cases)


def gen_query_copies(U, tmp):
    """get_query_copies (Util.py:6828) on synthetic blast6-like HSP tables of TE queries against chromosomes"""
    rng = np.random.default_rng(6828)
    cases = []
    for ci in range(40):
        nq, ns = int(rng.integers(1, 6)), int(rng.integers(1, 4))
        qlen = [int(rng.integers(200, 3000)) for _ in range(nq)]
        slen = [int(rng.integers(20_000, 200_000)) for _ in range(ns)]
        rows = []
        many = ci % 13 == 5
        scov = 0.02 if ci % 7 == 3 else 0.0
        for q in range(nq):
            for _copy in range(int(rng.integers(105, 140)) if (many and q == 0) else int(rng.integers(1, 14))):
                s = int(rng.integers(0, ns))
                rev = rng.random() < 0.5
                pos = int(rng.integers(100, slen[s] - 2 * qlen[q] - 100))
                cov = float(rng.choice([1.0, 1.0, 0.97, 0.6]))
                span = int(qlen[q] * cov)
                q0 = int(rng.integers(1, qlen[q] - span + 2))
                nfr = int(rng.integers(1, 5))
                cuts = sorted(set([0, span] + [int(x) for x in rng.integers(20, max(21, span - 20), size=nfr - 1)]))
                shift = 0
                for i in range(len(cuts) - 1):
                    a, b = cuts[i], cuts[i + 1]
                    gapq = int(rng.choice([0, 0, 5, 150, 199, 200, 260])) if i else 0
                    shift += int(rng.choice([0, 0, 3, -3, 120, 199, 200, 230])) if i else 0
                    fs, fe = q0 + a + (5 if gapq else 0), q0 + b - 1
                    if fe <= fs:
                        continue
                    if not rev:
                        ss_, se_ = pos + a + shift, pos + b - 1 + shift
                    else:
                        ss_, se_ = pos + span - a + shift, pos + span - b + 1 + shift
                    ident = float(rng.choice([100.0, 98.5, 91.25]))
                    rows.append((q, s, fs, fe, ss_, se_, ident))
                    if rng.random() < 0.08:
                        rows.append((q, s, fs, fe, ss_, se_, ident))              # exact duplicate line
                    if rng.random() < 0.08:
                        rows.append((q, s, fs + 3, fe, ss_ + (3 if not rev else -3), se_, 97.0))   # overlapping HSP
        order = rng.permutation(len(rows))
        rows = [rows[i] for i in order]
        qnames = ["TE_%d" % q for q in range(nq)]
        snames = ["chr%d" % s for s in range(ns)]
        recs = {}
        for (q, s, a, b, c, d, idt) in rows:
            recs.setdefault(qnames[q], {}).setdefault(snames[s], []).append((a, b, c, d, idt))
        qcov = float(rng.choice([0.95, 0.8, 0.5]))
        qc = {qnames[q]: "A" * qlen[q] for q in range(nq)}
        if scov > 0:        # the reference's subject coverage divides by the length of the SUBJECT contig (Util.py:7010)
            spath = os.path.join(tmp, "qc_subj_%d.fa" % ci)
            with open(spath, "w") as fh:
                for s_ in range(ns):
                    fh.write(">%s\n%s\n" % (snames[s_], "A" * slen[s_]))
        else:
            spath = None
        res = U.get_query_copies(list(recs.items()), qc, spath, qcov, scov)
        cases.append({"rows": [[int(x) for x in r[:6]] + [float(r[6])] for r in rows], "qlen": qlen, "slen": slen, "qcov": qcov, "scov": scov,
                      "out": {k: [[c[0], int(c[1]), int(c[2]), int(c[3]), c[4]] for c in v] for k, v in res.items()}})
    dump("query_copies", cases)


def _te_vs_genome_rows(rng, nq, ns, qlen, slen, many_first=False):
    """blast6-like HSPs of TE queries against chromosomes: copies cut into fragments with gaps / shifts around the thresholds"""
    rows = []
    for q in range(nq):
        for _copy in range(int(rng.integers(105, 125)) if (many_first and q == 0) else int(rng.integers(1, 12))):
            s = int(rng.integers(0, ns))
            rev = rng.random() < 0.5
            pos = int(rng.integers(100, slen[s] - 2 * qlen[q] - 100))
            cov = float(rng.choice([1.0, 1.0, 0.97, 0.94, 0.6]))
            span = int(qlen[q] * cov)
            q0 = int(rng.integers(1, qlen[q] - span + 2))
            nfr = int(rng.integers(1, 5))
            cuts = sorted(set([0, span] + [int(x) for x in rng.integers(20, max(21, span - 20), size=nfr - 1)]))
            shift = 0
            for i in range(len(cuts) - 1):
                a, b = cuts[i], cuts[i + 1]
                gapq = int(rng.choice([0, 0, 5, 150, 300])) if i else 0
                shift += int(rng.choice([0, 0, 3, -3, 120, 400, 1200])) if i else 0
                fs, fe = q0 + a + (5 if gapq else 0), q0 + b - 1
                if fe <= fs:
                    continue
                if not rev:
                    ss_, se_ = pos + a + shift, pos + b - 1 + shift
                else:
                    ss_, se_ = pos + span - a + shift, pos + span - b + 1 + shift
                rows.append((q, s, fs, fe, ss_, se_))
                if rng.random() < 0.08:
                    rows.append((q, s, fs, fe, ss_, se_))                      # exact duplicate line
                if rng.random() < 0.08:
                    rows.append((q, s, fs + 3, fe, ss_ + (3 if not rev else -3), se_))   # overlapping HSP
    order = rng.permutation(len(rows))
    return [rows[i] for i in order]


def gen_chain_variants(U, tmp):
    """The chaining variants next to get_longest_repeats_v4 / get_query_copies:
    FMEA (Util.py:10452), get_full_length_copies_from_blastn_v1 (:5907), generate_full_length_out_v1 (:6288; what
    mask_genome_intactTE reads) and multiple_alignment_blast_and_get_copies_v1 (:7179, its blastn call answered from
    prepared tables), each run by the reference on synthetic blast6 tables."""
    import shutil as _sh

    rng = np.random.default_rng(10452)
    out = {"fmea": [], "full_length": [], "multi_blast": []}
    # ---- FMEA: a library against itself (names coincide: exact self hits are skipped), fixed gaps
    for ci in range(16):
        n = int(rng.integers(2, 7))
        L = [int(rng.integers(300, 4000)) for _ in range(n)]
        rows = []
        for _ in range(int(rng.integers(4, 40))):
            q, s_ = int(rng.integers(0, n)), int(rng.integers(0, n))
            span = int(rng.integers(60, min(L[q], L[s_]) - 20))
            qa, sa = int(rng.integers(1, L[q] - span + 1)), int(rng.integers(1, L[s_] - span + 1))
            cuts = sorted(set([0, span] + [int(x) for x in rng.integers(10, max(11, span - 10), size=int(rng.integers(0, 3)))]))
            rev = rng.random() < 0.4
            for i in range(len(cuts) - 1):
                a, b = cuts[i], cuts[i + 1] - 1
                if b <= a:
                    continue
                g = int(rng.choice([0, 0, 7, 40]))
                if not rev:
                    rows.append((q, s_, qa + a + g, qa + b, sa + a + g, sa + b))
                else:
                    rows.append((q, s_, qa + a + g, qa + b, sa + span - a - g, sa + span - b))
        for q in range(n):
            if rng.random() < 0.7:
                rows.append((q, q, 1, L[q], 1, L[q]))            # the self hit of a library-vs-itself search
        rows = [rows[i] for i in rng.permutation(len(rows))]
        names = ["LTR_%d" % i for i in range(n)]
        p = os.path.join(tmp, "cv_fmea_%d.out" % ci)
        with open(p, "w") as f:
            f.writelines(casegen.hsp_to_blast6_lines([(names[q], names[s_], a, b, c, d) for (q, s_, a, b, c, d) in rows]))
        gap = int(rng.choice([1000, 50, 8]))
        res = U.FMEA(p, gap)
        out["fmea"].append({"rows": [list(r) for r in rows], "names": names, "gap": gap,
                            # (lists of [key, value]: dump() sorts dict keys, the insertion order of the reference's dicts is part of the contract)
                            "out": [[k, [[t[0], int(t[1]), int(t[2]), t[3], int(t[4]), int(t[5])] for t in v]] for k, v in res.items()]})
    # ---- full-length copies of a TE library in a genome, and the sets mask_genome_intactTE reads
    for ci in range(24):
        nq, ns = int(rng.integers(1, 6)), int(rng.integers(1, 4))
        qlen = [int(rng.integers(200, 3000)) for _ in range(nq)]
        slen = [int(rng.integers(20_000, 200_000)) for _ in range(ns)]
        rows = _te_vs_genome_rows(rng, nq, ns, qlen, slen)
        if ci % 4 == 2 and qlen[0] >= 320:
            # round 4 (tools/ref_line_coverage.py, Util.py:6079 / 6096): two short fragments of the two ENDS of a query next to each other
            # in the subject, both strands -- the query gap reaches skip_gap = 0.95 x the query, the extension loop breaks
            Lq = qlen[0]
            rows += [(0, 0, 1, 9, 5001, 5009), (0, 0, Lq - 8, Lq, 5030, 5038), (0, 0, 1, 9, 9038, 9030), (0, 0, Lq - 8, Lq, 9009, 9001)]
        qnames = ["TE_%d#%s" % (q, ["DNA/hAT", "LTR/Gypsy", "Unknown"][q % 3]) if q % 2 else "Helitron_%d" % q for q in range(nq)]
        snames = ["chr%d" % s_ for s_ in range(ns)]
        thr = float(rng.choice([0.95, 0.95, 0.8]))
        lib = os.path.join(tmp, "cv_lib_%d.fa" % ci)
        ref = os.path.join(tmp, "cv_ref_%d.fa" % ci)
        # one query of the table is missing from the library every few cases (the reference skips it)
        drop = int(rng.integers(0, nq)) if (ci % 5 == 4 and nq > 1) else -1
        write_fasta(lib, [qnames[q] for q in range(nq) if q != drop], ["ACGT" * (qlen[q] // 4) + "A" * (qlen[q] % 4) for q in range(nq) if q != drop])
        gen = np.random.default_rng(1000 + ci)
        refseqs = ["".join("ACGT"[i] for i in gen.integers(0, 4, slen[s_])) for s_ in range(ns)]
        write_fasta(ref, snames, refseqs)
        p = os.path.join(tmp, "cv_fl_%d.out" % ci)
        lines = casegen.hsp_to_blast6_lines([(qnames[q], snames[s_], a, b, c, d) for (q, s_, a, b, c, d) in rows])
        if ci % 6 == 1:
            lines.insert(0, "# a comment line\n")
        with open(p, "w") as f:
            f.writelines(lines)
        search_struct = ci % 4 == 3
        fl, ffl = U.get_full_length_copies_from_blastn_v1(lib, ref, p, tmp, 1, 20, thr, search_struct, "")
        p2 = p + ".copy"
        _sh.copy(p, p2)
        cat = "Total" if ci % 3 else "DNA"
        if cat != "Total" and any("#" not in qn for qn in qnames):
            cat = "Total"                     # (the reference indexes the class after '#': names without one would raise)
        files = U.generate_full_length_out_v1(p2, lib, ref, os.path.join(tmp, "cv_w_%d" % ci), "", thr, cat, debug=0)
        sets = [sorted([list(t) for t in U.load_from_file(fp)]) for fp in files]
        out["full_length"].append({"rows": [list(r) for r in rows], "qnames": qnames, "snames": snames, "qlen": qlen, "slen": slen, "thr": thr,
                                   "drop": drop, "comment": ci % 6 == 1, "search_struct": search_struct, "ref_seed": 1000 + ci,
                                   "copies": [[k, [[kk, vv] for kk, vv in v.items()]] for k, v in fl.items()],
                                   "flank_copies": [[k, [[kk, vv] for kk, vv in v.items()]] for k, v in ffl.items()],
                                   "category": cat, "out_files": [os.path.basename(fp) for fp in files], "out_sets": sets})
    # ---- multiple_alignment_blast_and_get_copies_v1: the blastn of each chromosome file answered from a prepared table
    real_system, real_listdir = os.system, os.listdir
    for ci in range(6):
        nq, ns = int(rng.integers(2, 6)), int(rng.integers(2, 5))
        qlen = [int(rng.integers(200, 2000)) for _ in range(nq)]
        slen = [int(rng.integers(60_000, 200_000)) for _ in range(ns)]
        rows = _te_vs_genome_rows(rng, nq, ns, qlen, slen, many_first=ci % 2 == 0)
        if ci == 0:
            # round 4 (Util.py:7206): 60 full-length single-HSP copies of query 0 per chromosome file -- it reaches 100 copies after
            # the second file and leaves the query file before the third is searched
            for s_ in range(ns):
                rows += [(0, s_, 1, qlen[0], 1000 + 700 * k_, 1000 + 700 * k_ + qlen[0] - 1) for k_ in range(60) if 1000 + 700 * k_ + qlen[0] < slen[s_]]
        qnames = ["TE_%d" % q for q in range(nq)]
        files = ["chr%d.fa" % s_ for s_ in range(ns)] + ["chr0.fa.nhr", "notes.txt"]
        files = [files[i] for i in rng.permutation(len(files))]
        d = os.path.join(tmp, "cv_mb_%d" % ci)
        os.makedirs(d, exist_ok=True)
        qpath = os.path.join(d, "q.fa")
        write_fasta(qpath, qnames, ["A" * L for L in qlen])
        refdir = os.path.join(d, "ref")
        os.makedirs(refdir, exist_ok=True)
        tables = {}
        for s_ in range(ns):
            tables["chr%d.fa" % s_] = [(qnames[q], "chr%d" % s2, a, b, c, e) for (q, s2, a, b, c, e) in rows if s2 == s_]
        if ci == 1:     # a line whose subject IS the query (the library sequence among the subjects): get_copies_v1 skips it (Util.py:7045-7046)
            tables["chr0.fa"].insert(len(tables["chr0.fa"]) // 2, (qnames[0], qnames[0], 1, qlen[0], 1, qlen[0]))
        calls = []

        def fake_system(cmd, _tables=tables, _calls=calls):
            if cmd.startswith("blastn "):
                db = cmd.split(" -db ")[1].split(" ")[0]
                outp = cmd.split(" > ")[1].strip()
                qp = cmd.split(" -query ")[1].split(" ")[0]
                live = set(U.read_fasta(qp)[0])
                _calls.append(os.path.basename(db))
                with open(outp, "w") as f:
                    f.writelines(casegen.hsp_to_blast6_lines([r for r in _tables[os.path.basename(db)] if r[0] in live]))
                return 0
            return real_system(cmd)

        os.system = fake_system
        os.listdir = lambda pth, _files=files, _refdir=refdir: list(_files) if pth == _refdir else real_listdir(pth)
        try:
            res = U.multiple_alignment_blast_and_get_copies_v1((qpath, refdir, os.path.join(d, "b.out")))
        finally:
            os.system, os.listdir = real_system, real_listdir
        out["multi_blast"].append({"qnames": qnames, "qlen": qlen, "files": files,
                                   "tables": {k: [list(r) for r in v] for k, v in tables.items()}, "blast_calls": calls,
                                   "left_in_query_file": U.read_fasta(qpath)[0],
                                   "out": [[k, [[c[0], int(c[1]), int(c[2]), int(c[3]), c[4]] for c in v]] for k, v in res.items()]})
    dump("chain_variants", out)


def gen_lib_dedup(U, tmp):
    """panHiTE library de-duplication: process_blast_results_in_chunks -> process_chunk (extend_fragments) ->
    cluster_sequences_from_chunks, and cons_from_mafft_v1, run through the reference on synthetic all-vs-all tables"""
    import shutil
    rng = np.random.default_rng(12202)
    cases = []
    for ci in range(36):
        nseq = int(rng.integers(2, 12))
        lens = [int(rng.integers(150, 2500)) for _ in range(nseq)]
        thr = float(rng.choice([0.95, 0.8, 0.9]))
        rows = []
        for _ in range(int(rng.integers(1, 60))):
            q, s = int(rng.integers(0, nseq)), int(rng.integers(0, nseq))
            rev = rng.random() < 0.4
            span = max(30, int(min(lens[q], lens[s]) * float(rng.choice([1.0, 0.97, 0.9, 0.5, 0.2]))))
            span = min(span, lens[q], lens[s])
            q0 = int(rng.integers(1, lens[q] - span + 2)); s0 = int(rng.integers(1, lens[s] - span + 2))
            cuts = sorted(set([0, span] + [int(x) for x in rng.integers(5, max(6, span - 5), size=int(rng.integers(0, 4)))]))
            for i in range(len(cuts) - 1):
                a, b = cuts[i], cuts[i + 1]
                jq = int(rng.choice([0, 0, 3, 20, 60])) if i else 0
                js = int(rng.choice([0, 0, -2, 4, 30, 90])) if i else 0
                fs, fe = q0 + a + jq, q0 + b - 1
                if fe < fs:
                    continue
                if not rev:
                    ss_, se_ = s0 + a + js, s0 + b - 1
                else:
                    ss_, se_ = s0 + span - a - 1 - js, s0 + span - b
                if ss_ < 1 or se_ < 1:
                    continue
                rows.append((q, s, fs, fe, ss_, se_))
                if rng.random() < 0.06:
                    rows.append((q, s, fs, fe, ss_, se_))
            if rng.random() < 0.3:
                rows.append((q, q, 1, lens[q], 1, lens[q]))                      # self hit (skipped, counts for chunking)
            if rng.random() < 0.1:
                rows.append((q, q, 1, 80, lens[q] - 79, lens[q]))               # internal repeat of the same sequence
        rows = [rows[i] for i in rng.permutation(len(rows))]
        chunk_size = int(rng.choice([5_000_000, 7, 13])) if ci % 3 else 5_000_000
        names = ["seq_%d" % i for i in range(nseq)]
        work = os.path.join(tmp, "lib_%d" % ci)
        os.makedirs(work, exist_ok=True)
        bl = os.path.join(work, "all.out")
        with open(bl, "w") as fh:
            for (q, s, a, b, c, d) in rows:
                fh.write("%s\t%s\t95.0\t%d\t0\t0\t%d\t%d\t%d\t%d\t1e-20\t200\n" % (names[q], names[s], b - a + 1, a, b, c, d))
        if ci % 4 == 1:       # a chunk directory left by an earlier run is removed first (Util.py:12163-12164): a stale file must not come back
            stale = os.path.join(work, "LTR_query_records_t_chunks")
            os.makedirs(stale, exist_ok=True)
            with open(os.path.join(stale, "chunk_999.pkl"), "wb") as fh:
                fh.write(b"stale")
        files = U.process_blast_results_in_chunks(bl, work, "t", chunk_size=chunk_size)
        assert not any(f.endswith("chunk_999.pkl") for f in files) and not os.path.exists(os.path.join(work, "LTR_query_records_t_chunks", "chunk_999.pkl"))
        qlens = {names[i]: lens[i] for i in range(nseq)}
        lr_dir = os.path.join(work, "lr")
        os.makedirs(lr_dir, exist_ok=True)
        lr_files, recs = [], []
        for k, f in enumerate(files):                                            # FMEA_new1_parallel_large, one job per file, in order
            idx, res = U.process_chunk(qlens, [f], thr, k)
            lr_files.append(U.save_data_in_chunks(res, lr_dir, idx))
            for qn, lst in res.items():
                for r in lst:
                    recs.append([k, int(r[0][4:]), int(r[1]), int(r[2]), int(r[3][4:]), int(r[4]), int(r[5])])
        contigs = {names[i]: "A" * lens[i] for i in range(nseq)}
        clusters = U.cluster_sequences_from_chunks(lr_files, contigs, thr)
        cases.append({"rows": [list(map(int, r)) for r in rows], "lens": lens, "thr": thr, "chunk_size": chunk_size, "recs": recs,
                      "clusters": [sorted(int(x[4:]) for x in cl) for cl in clusters]})
        shutil.rmtree(work)
    cons = []
    for ci in range(30):
        R = int(rng.integers(1, 14)); L = int(rng.integers(5, 300))
        base = casegen.rand_seq(rng, L)
        mat = []
        for r in range(R):
            row = list(base if rng.random() < 0.8 else casegen.rand_seq(rng, L))
            for c in range(L):
                x = rng.random()
                if x < 0.25:
                    row[c] = "-"
                elif x < 0.32:
                    row[c] = "ACGTNacgt"[int(rng.integers(0, 9))]
            mat.append("".join(row))
        af = os.path.join(tmp, "cons_%d.maf.fa" % ci)
        with open(af, "w") as fh:
            for r, row in enumerate(mat):
                fh.write(">r%d\n%s\n" % (r, row))
        cons.append({"rows": mat, "cons": U.cons_from_mafft_v1(af)})
    dump("lib_dedup", {"chain": cases, "cons": cons})


def gen_cons_v1(U, tmp):
    """generate_cons_v1 (Util.py:12457-12498) with its two external tools replaced inside the harness: `mafft` by the oracle's
    star alignment (centre = the longest sequence, rows in input order), `Ninja` by a cluster file FABRICATED per case.
    What the fixture pins is everything HiTE owns around them: read_Ninja_clusters, the sub-cluster files, the
    second alignment, cons_from_mafft_v1, the naming by the last member, the fall-back to the original sequences."""
    import oracle_lib as O

    rng = np.random.default_rng(1212)
    cases = []
    state = {"ninja": None, "first_call": True}

    def fake_system(cmd):
        if "mafft " in cmd:
            left, out = cmd.rsplit(">", 1)
            src = left.split()[-1]
            names, contigs = U.read_fasta(src)
            seqs = [contigs[n] for n in names]
            centre = max(range(len(seqs)), key=lambda i: (len(seqs[i]), -i))
            order = [centre] + [i for i in range(len(seqs)) if i != centre]
            m, kept = O.star_msa([seqs[i] for i in order], rows=True)
            back = {i: k for k, i in enumerate(order)}
            with open(out.strip(), "w") as f:
                if m is None or kept != len(seqs):
                    # unrelated sequences in one file (the whole-cluster alignment, which only Ninja reads): any valid alignment
                    # will do for a fabricated Ninja -- left-justified rows
                    assert state["first_call"], "a sub-cluster of the harness cases must align completely"
                    W = max(len(x) for x in seqs)
                    for n, x in zip(names, seqs):
                        f.write(">" + n + "\n" + x + "-" * (W - len(x)) + "\n")
                else:
                    for i, n in enumerate(names):
                        f.write(">" + n + "\n" + bytes(m[back[i]]).decode() + "\n")
            state["first_call"] = False
            return 0
        if cmd.startswith("Ninja "):
            toks = cmd.split()
            out = toks[toks.index("--out") + 1]
            with open(out, "w") as f:
                for cid, members in state["ninja"].items():
                    for n in members:
                        f.write("%d\t%s\n" % (cid, n))
            return 0
        return 0

    real_system = U.os.system
    U.os.system = fake_system
    try:
        for ci in range(14):
            nfam = int(rng.integers(1, 4))
            recs, assign = [], {}
            for f in range(nfam):
                cons = casegen.rand_seq(rng, int(rng.integers(150, 700)))
                for k in range(int(rng.integers(1, 6))):
                    sq = casegen.mutate(rng, cons, float(rng.uniform(0.0, 0.06)))
                    if rng.random() < 0.3:      # a few bases missing at one end
                        cut = int(rng.integers(1, 12))
                        sq = sq[cut:] if rng.random() < 0.5 else sq[:-cut]
                    name = "G%d-fam%d_%d#LTR/Gypsy" % (k, f, ci)
                    recs.append((name, sq))
                    assign.setdefault(f, []).append(name)
            order = rng.permutation(len(recs))
            recs = [recs[i] for i in order]
            pos = {n: i for i, (n, _s) in enumerate(recs)}
            ninja = {cid: sorted(members, key=lambda n: pos[n]) for cid, members in assign.items()}
            cdir = os.path.join(tmp, "cons_v1_%d" % ci)
            os.makedirs(cdir, exist_ok=True)
            path = os.path.join(cdir, "0.fa")
            write_fasta(path, [n for n, _s in recs], [sq for _n, sq in recs])
            state["ninja"] = ninja
            if ci % 4 == 3 and nfam == 1 and len(recs) >= 4:
                # Ninja may also split one family: two sub-clusters of the same family
                half = len(recs) // 2
                ninja = {0: [n for n, _s in recs[:half]], 1: [n for n, _s in recs[half:]]}
                state["ninja"] = ninja
            state["first_call"] = True
            got = U.generate_cons_v1(0, path, cdir, 1)
            cases.append(dict(names=[n for n, _s in recs], seqs=[sq for _n, sq in recs], ninja={str(k): v for k, v in ninja.items()},
                              expected={k: v for k, v in got.items()}))
        # Ninja writes no cluster at all: "no reliable consensus, the original sequences instead" (Util.py:12495-12498)
        recs = [("G%d-fam0_e#LTR/Copia" % k, casegen.rand_seq(rng, 180 + 40 * k)) for k in range(3)]
        cdir = os.path.join(tmp, "cons_v1_empty")
        os.makedirs(cdir, exist_ok=True)
        path = os.path.join(cdir, "0.fa")
        write_fasta(path, [n for n, _s in recs], [sq for _n, sq in recs])
        state["ninja"] = {}
        state["first_call"] = True
        got = U.generate_cons_v1(0, path, cdir, 1)
        assert got == dict(recs)
        cases.append(dict(names=[n for n, _s in recs], seqs=[sq for _n, sq in recs], ninja={}, expected={k: v for k, v in got.items()}))
    finally:
        U.os.system = real_system
    print("cons_v1: %d cases, %d consensus sequences" % (len(cases), sum(len(c["expected"]) for c in cases)))
    dump("cons_v1", cases)


def gen_trf_mask(tmp):
    """TRF 4.09 itself -- the binary the reference bundles (tools/trf409.linux64) run with the reference's command line
    (Util.py:2859) -- on seeded sequences with planted tandem arrays: the fixture holds the sequences, the planted intervals
    and TRF's .mask output as intervals.  The build's masker is MEASURED against it (it is not a restatement of TRF)."""
    import re
    import shutil
    import subprocess

    exe = os.path.join(tmp, "trf")
    shutil.copyfile(os.path.join(ref_harness.REFERENCE_ROOT, "tools", "trf409.linux64"), exe)
    os.chmod(exe, 0o755)
    cases = []
    for seed in (101, 102, 103):
        seq, planted = casegen.make_tandem_case(seed)
        d = os.path.join(tmp, "trf_%d" % seed)
        os.makedirs(d)
        with open(os.path.join(d, "x.fa"), "w") as f:
            f.write(">s\n" + seq + "\n")
        subprocess.run("cd %s && %s x.fa 2 7 7 80 10 50 500 -f -d -m -h > /dev/null 2>&1" % (d, exe), shell=True, check=False)
        masked = open(os.path.join(d, "x.fa.2.7.7.80.10.50.500.mask")).read().split("\n", 1)[1].replace("\n", "")
        assert len(masked) == len(seq)
        iv = [[m.start(), m.end()] for m in re.finditer("N+", masked)]
        cases.append(dict(seed=seed, seq=seq, planted=planted, trf_masked=iv))
        print("trf_mask seed %d: %d bases, %d planted arrays, TRF masks %d intervals / %d bases" %
              (seed, len(seq), len(planted), len(iv), sum(b - a for a, b in iv)))
    dump("trf_mask", cases)


def gen_ready_for_msa(tmp):
    """tools/ready_for_MSA.sh <file> 100 100 (is_TE_from_align_file, Util.py:10410) run as the reference runs it, under the POSIX
    locale of the reference's container.  Its one external call, `samtools faidx <file>`, only writes the .fai index (name,
    length, offset, line bases, line width -- the documented format); samtools is absent from this image, so a ten-line writer
    of that file stands on the PATH under its name.  Everything the fixture pins is the script's own: `sort -nk 2 -r` (equal
    lengths fall to the reverse byte order of the line), head, `grep -a -A 1 -f` (input order kept)."""
    import subprocess

    bindir = os.path.join(tmp, "fakebin")
    os.makedirs(bindir, exist_ok=True)
    shim = os.path.join(bindir, "samtools")
    with open(shim, "w") as f:
        f.write("#!/usr/bin/env python3\nimport sys\nassert sys.argv[1] == 'faidx'\npath = sys.argv[2]\nout = open(path + '.fai', 'w')\n"
                "data = open(path, 'rb').read()\npos = 0\nlines = data.split(b'\\n')\ni = 0\n"
                "while i + 1 < len(lines):\n    h, s = lines[i], lines[i + 1]\n    if h.startswith(b'>'):\n"
                "        name = h[1:].split()[0].decode()\n        off = pos + len(h) + 1\n"
                "        out.write('%s\\t%d\\t%d\\t%d\\t%d\\n' % (name, len(s), off, len(s), len(s) + 1))\n"
                "    pos += len(h) + 1 + len(s) + 1\n    i += 2\nout.close()\n")
    os.chmod(shim, 0o755)
    rng = np.random.default_rng(4410)
    contig_sets = (["chr1", "chr2", "chr10"], ["scaffold_3", "Chr1", "chr1_random", "chr1"], ["1", "10", "2", "X"])
    cases = []
    for ci in range(10):
        cn = contig_sets[ci % 3]
        n = int(rng.choice([90, 101, 130, 260, 300]))
        names, lens, seen = [], [], set()
        while len(names) < n:
            s = int(rng.choice([7, 95, 99, 100, 1000, 9999, 10000, 123456, int(rng.integers(1, 2_000_000))]))
            ln = 1000 if ci % 2 == 0 else int(rng.choice([150, 150, 151, 400, 1000, 1000, int(rng.integers(100, 3000))]))
            nm = "%s:%d-%d(%s)" % (cn[int(rng.integers(0, len(cn)))], s, s + ln - 101, "+-"[int(rng.integers(0, 2))])
            if nm in seen:
                continue
            seen.add(nm)
            names.append(nm)
            lens.append(ln)
        d = os.path.join(tmp, "rfm_%d" % ci)
        os.makedirs(d)
        fa = os.path.join(d, "members.fa")
        with open(fa, "w") as f:
            for nm, ln in zip(names, lens):
                f.write(">" + nm + "\n" + "ACGT" * (ln // 4) + "A" * (ln % 4) + "\n")
        env = dict(os.environ, PATH=bindir + ":" + os.environ["PATH"], LC_ALL="C")
        subprocess.run(["sh", os.path.join(ref_harness.REFERENCE_ROOT, "tools", "ready_for_MSA.sh"), fa, "100", "100"], cwd=d, env=env, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        picked = [ln_[1:].strip() for ln_ in open(fa + ".rdmSubset.fa") if ln_.startswith(">")]
        cases.append(dict(names=names, lens=lens, selected=picked))
        print("ready_for_MSA case %d: %d members -> %d kept" % (ci, n, len(picked)))
    dump("ready_for_msa", cases)


def gen_split_chunks(U, tmp):
    """module/split_genome_chunks.py run as a script (runpy) on small genomes: the reference FASTA is rewritten upper-case in
    place (convertToUpperCase_v1), cut into chr$offset segments (multi_line) and grouped into genome.cut{i}.fa by FASTA-text
    bytes; ref_chr/ref_block_{i}.fa by split_dict_into_blocks.  makeblastdb is absent here (its call fails silently)."""
    import runpy

    script = os.path.join(ref_harness.REFERENCE_ROOT, "module", "split_genome_chunks.py")
    cases = []
    rng = np.random.default_rng(4711)
    for ci, (seg_len, chunk_mb, lens) in enumerate([(1000, 0.005, [3500, 1200, 999, 1000, 2001]), (700, 0.002, [5000]),
                                                    (1000, 400, [1500, 800]), (500, 0.0011, [499, 500, 501, 1, 2600])]):
        d = os.path.join(tmp, "split_%d" % ci)
        os.makedirs(d)
        ref = os.path.join(d, "genome.fa")
        text = []
        for k, L in enumerate(lens):
            seq = casegen.rand_seq(rng, L)
            if k % 2 == 0:
                seq = seq.lower() if k % 4 == 0 else seq[:L // 2] + seq[L // 2:].lower()
            text.append(">Chr%d some description %d\n" % (k + 1, k))
            w = int(rng.integers(50, 90))
            text += [seq[p:p + w] + "\n" for p in range(0, L, w)]
        text = "".join(text)
        with open(ref, "w") as f:
            f.write(text)
        argv = sys.argv
        sys.argv = [script, "-g", ref, "--tmp_output_dir", d, "--chrom_seg_length", str(seg_len), "--chunk_size", str(chunk_mb)]
        try:
            runpy.run_path(script, run_name="__main__")
        finally:
            sys.argv = argv
        files = {}
        for fn in sorted(os.listdir(d)):
            if fn.startswith("genome.cut") and fn.endswith(".fa"):
                files[fn] = open(os.path.join(d, fn)).read()
        for fn in sorted(os.listdir(os.path.join(d, "ref_chr"))):
            if fn.endswith(".fa"):
                files["ref_chr/" + fn] = open(os.path.join(d, "ref_chr", fn)).read()
        files["genome.fa"] = open(ref).read()
        cases.append(dict(input=text, chrom_seg_length=seg_len, chunk_size=chunk_mb, files=files))
    dump("split_chunks", cases)


def gen_bucketing(U, tmp):
    """the collection loop of flank_region_align_v5 (Util.py:8159-8194, 8282-8287) with run_find_members_v8 replaced by a table
    of result tuples: which consensus goes to real_TEs, which to all_low_copy, which is dropped (TG...CA), per TE type.  The
    low-copy rescue (itrsearch / blastx domains, external tools) is stubbed to rescue nothing, as when those tools are absent."""
    rng = np.random.default_rng(8159)
    cases = []
    for te_type in ("tir", "helitron", "non_ltr", "other"):
        for rep in range(3):
            names, seqs = casegen.make_genome(10 + rep)
            ref = os.path.join(tmp, "bg_%s_%d.fa" % (te_type, rep))
            write_fasta(ref, names, seqs)
            table = {}
            copies = {}
            for q in range(40):
                qn = "q%d-C_%d" % (q, rep)
                body = casegen.rand_seq(rng, int(rng.integers(90, 400)))
                kind = int(rng.integers(0, 5))
                if kind == 0:
                    body = "TG" + body[2:-2] + "CA"
                if kind == 1:
                    body = "TG" + body[2:]
                cc = int(rng.choice([0, 1, 2, 3, 5, 6, 7, 40, 100]))
                info = str(rng.choice(["", "", "", "nb", "fl1", "copy_num:3"]))
                is_te = info in ("", "copy_num:3") and rng.random() < 0.8
                table[qn] = (qn if is_te else None, body if is_te else None, info, cc)
                copies[qn] = [(names[0], 100, 400, 301, "+")]
            cand = os.path.join(tmp, "bc_%s_%d.fa" % (te_type, rep))
            write_fasta(cand, list(table.keys()), ["ACGT" * 30 for _ in table])

            def fake_copies(query_path, reference, temp_dir, max_copy_num, threads, _c=copies):
                os.makedirs(temp_dir, exist_ok=True)
                return _c

            def fake_members(task, temp_dir, subset_script_path, plant, TE_type, debug, result_type, _t=table):
                (query_name, cur_seq, trunc_member_file, extend_member_file) = task
                r = _t[query_name]
                return (r[0], r[1], r[2], r[3], extend_member_file)

            def fake_remove_no_tirs(low_copy_path, plant, TRsearch_dir, low_copy_dir):
                w, n = os.path.join(low_copy_dir, "with_tir.fa"), os.path.join(low_copy_dir, "no_tir.fa")
                open(w, "w").close(); open(n, "w").close()
                return w, n

            def fake_domain(path, db, output_table, threads, temp_dir):
                with open(output_table, "w") as f:
                    f.write("#h1\n#h2\n")

            saved = (U.get_full_length_copies_minimap2, U.run_find_members_v8, U.ProcessPoolExecutor, U.as_completed, U.remove_no_tirs,
                     U.get_domain_info)
            U.get_full_length_copies_minimap2 = fake_copies
            U.run_find_members_v8 = fake_members
            U.ProcessPoolExecutor = ref_harness.SyncExecutor
            U.as_completed = lambda fs: fs
            U.remove_no_tirs = fake_remove_no_tirs
            U.get_domain_info = fake_domain
            real = os.path.join(tmp, "real_%s_%d.fa" % (te_type, rep))
            low = os.path.join(tmp, "low_%s_%d.fa" % (te_type, rep))
            with open(low, "w") as f:
                f.write(">earlier\nACGT\n")
            try:
                log = type("L", (), {"logger": type("LL", (), {"info": staticmethod(lambda *a: None), "debug": staticmethod(lambda *a: None)})})()
                U.flank_region_align_v5(cand, real, 50, ref, None, te_type, os.path.join(tmp, "bw_%s_%d" % (te_type, rep)), 1, 0, log,
                                        "", 1, 0, 0, low)
            finally:
                (U.get_full_length_copies_minimap2, U.run_find_members_v8, U.ProcessPoolExecutor, U.as_completed, U.remove_no_tirs,
                 U.get_domain_info) = saved
            rn, rc = U.read_fasta(real)
            cases.append(dict(te_type=te_type, table=[[k, v[0], v[1], v[2], v[3]] for k, v in table.items()],
                              real=[[x, rc[x]] for x in rn], low_text=open(low).read()))
    dump("bucketing", cases)



def _itrsearch_dir(tmp):
    """the ELF the reference bundles (tools/itrsearch), copied and made executable: /root/reference is read-only"""
    import shutil

    d = os.path.join(tmp, "trsearch")
    if not os.path.exists(os.path.join(d, "itrsearch")):
        os.makedirs(d, exist_ok=True)
        shutil.copyfile(os.path.join(ref_harness.REFERENCE_ROOT, "tools", "itrsearch"), os.path.join(d, "itrsearch"))
        os.chmod(os.path.join(d, "itrsearch"), 0o755)
    return d


def gen_itr_search(U, tmp):
    """`itrsearch -i 0.7 -l 7` ITSELF -- the ELF the reference bundles, through the reference's own wrapper run_itrsearch
    (Util.py:216-224) -- on seeded records: which of them it writes to <input>.itr and the "Length itr=" of their headers (what
    search_confident_tir_batch_v1 reads, Util.py:6587-6596).  `pairs`: first 40 + last 40 records; `whole`: long sequences as
    remove_no_tirs submits them.  Then the reference functions around the tool, run with the tool: search_confident_tir_batch_v1
    (Util.py:6533-6628; its pick among variants of equal distance depends on PYTHONHASHSEED, the test allows any of them) and
    remove_no_tirs (Util.py:13897-13920)."""
    trs = _itrsearch_dir(tmp)

    def run(seqs, tag):
        d = os.path.join(tmp, "itr_" + tag)
        os.makedirs(d, exist_ok=True)
        fa = os.path.join(d, tag + ".fa")
        write_fasta(fa, ["s%d" % k for k in range(len(seqs))], seqs)
        out, _log = U.run_itrsearch(trs, fa, d)
        found = {}
        for line in open(out):
            if line.startswith(">"):
                found[int(line[2:].split(" ")[0])] = int(line.split("Length itr=")[1])
        return [[1, found[k]] if k in found else [0, -1] for k in range(len(seqs))]

    pairs = casegen.make_itr_cases(5101, 3000)
    whole = casegen.make_itr_cases(5102, 300, long_=True)
    obj = dict(pairs=dict(seqs=pairs, res=run(pairs, "pairs")), whole=dict(seqs=whole, res=run(whole, "whole")))
    print("itr_search: pairs %d found of %d, whole %d of %d" % (sum(r[0] for r in obj["pairs"]["res"]), len(pairs),
                                                                 sum(r[0] for r in obj["whole"]["res"]), len(whole)))
    batches = []
    for bi, (seed, plant) in enumerate(((5201, 1), (5202, 0), (5203, 1))):
        names, seqs = casegen.make_tir_batch(seed, n=70)
        d = os.path.join(tmp, "itr_batch_%d" % bi)
        os.makedirs(d, exist_ok=True)
        fa = os.path.join(d, "split.fa")
        write_fasta(fa, names, seqs)
        res = U.search_confident_tir_batch_v1(fa, 50, d, trs, 0, plant)
        batches.append(dict(names=names, seqs=seqs, plant=plant, flank=50, out=[[k, v] for k, v in res.items()]))
        print("itr_search batch %d: %d candidates -> %d with a TIR variant" % (bi, len(names), len(res)))
    obj["batches"] = batches
    rescue = []
    for bi, (seed, plant) in enumerate(((5301, 1), (5302, 0))):
        rng = np.random.default_rng(seed)
        seqs = casegen.make_itr_cases(seed, 80, long_=True) + casegen.make_itr_cases(seed + 10, 40)
        names = []
        for k, s_ in enumerate(seqs):
            tl = int(rng.choice([2, 3, 8, 9, 10]))
            names.append("N_%d-tir_%d-tsd_%s" % (k, int(rng.integers(0, 30)), casegen.rand_seq(rng, tl)))
            r = rng.random()
            if r < 0.15:      # short-TIR signatures (get_short_tir_contigs, Util.py:7297): kept without the tool
                head = "CACTA" if (tl == 3 and r < 0.08) else s_[:5].replace("N", "A")
                seqs[k] = head + s_[5:-5] + casegen.revcomp(head)
            elif r < 0.2:
                seqs[k] = "CCC" + s_[3:-3] + "GGG"
        d = os.path.join(tmp, "itr_rescue_%d" % bi)
        os.makedirs(d, exist_ok=True)
        fa = os.path.join(d, "low_copy.fa")
        write_fasta(fa, names, seqs)
        with_tir, no_tir = U.remove_no_tirs(fa, plant, trs, d)
        wn, wc = U.read_fasta(with_tir)
        nn, nc = U.read_fasta(no_tir)
        rescue.append(dict(names=names, seqs=seqs, plant=plant, with_tir=[[k, wc[k]] for k in wn], no_tir=[[k, nc[k]] for k in nn]))
        print("itr_search rescue %d: %d low-copy sequences -> %d with a TIR, %d without" % (bi, len(names), len(wn), len(nn)))
    obj["rescue"] = rescue
    dump("itr_search", obj)



def gen_low_copy_rescue(U, tmp):
    """The low-copy recall at the end of flank_region_align_v5 (Util.py:8196-8287) run by the REFERENCE with its tools: `trf` =
    the TRF 4.09 it bundles, `itrsearch` = the ELF it bundles (through remove_no_tirs), get_domain_info / multiple_alignment_blastx_v1
    (Util.py:4571-4612, 1006-1262) as they are, with `blastx` (NCBI BLAST+, absent from the image) answered by a shim that prints a
    FABRICATED `-outfmt 6` table for the sequences of its query file, and `makeblastdb` a no-op.  run_find_members_v8 is a table of
    result tuples (as in `bucketing`).  The fixture holds the tuples, the protein libraries, the fabricated blastx lines and what
    the reference wrote: real_TEs, all_low_copy and the domain table."""
    import json as _json
    import shutil

    root = os.path.join(tmp, "lcr_root")
    os.makedirs(os.path.join(root, "tools"), exist_ok=True)
    os.makedirs(os.path.join(root, "library"), exist_ok=True)
    shutil.copyfile(os.path.join(_itrsearch_dir(tmp), "itrsearch"), os.path.join(root, "tools", "itrsearch"))
    os.chmod(os.path.join(root, "tools", "itrsearch"), 0o755)
    bindir = os.path.join(tmp, "lcr_bin")
    os.makedirs(bindir, exist_ok=True)
    shutil.copyfile(os.path.join(ref_harness.REFERENCE_ROOT, "tools", "trf409.linux64"), os.path.join(bindir, "trf"))
    os.chmod(os.path.join(bindir, "trf"), 0o755)
    with open(os.path.join(bindir, "makeblastdb"), "w") as f:
        f.write("#!/bin/sh\nexit 0\n")
    with open(os.path.join(bindir, "blastx"), "w") as f:
        f.write("#!/usr/bin/env python3\nimport json, os, sys\nq = sys.argv[sys.argv.index('-query') + 1]\n"
                "tab = json.load(open(os.environ['HITE_FAKE_BLASTX']))\n"
                "for line in open(q):\n    if line.startswith('>'):\n        for r in tab.get(line[1:].strip(), []):\n            sys.stdout.write(r)\n")
    for x in ("makeblastdb", "blastx"):
        os.chmod(os.path.join(bindir, x), 0o755)
    rng = np.random.default_rng(8215)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    libs = {}
    for fn in ("TIRPeps.lib", "HelitronPeps.lib", "non_LTR.lib"):
        prot = {"%s_p%d#DNA" % (fn.split(".")[0], k): "".join(aa[int(x)] for x in rng.integers(0, 20, size=int(rng.integers(120, 600)))) for k in range(6)}
        write_fasta(os.path.join(root, "library", fn), list(prot.keys()), list(prot.values()))
        libs[fn] = [[k, v] for k, v in prot.items()]
    cases = []
    for te_type, fn in (("tir", "TIRPeps.lib"), ("helitron", "HelitronPeps.lib"), ("non_ltr", "non_LTR.lib")):
        for rep in range(2):
            prot = dict(libs[fn])
            pnames = list(prot.keys())
            names, seqs = casegen.make_genome(40 + rep)
            ref = os.path.join(tmp, "lg_%s_%d.fa" % (te_type, rep))
            write_fasta(ref, names, seqs)
            table, copies, blast = {}, {}, {}
            for q in range(48):
                body = casegen.rand_seq(rng, int(rng.integers(600, 2600)))
                kind = int(rng.integers(0, 6))
                tsd = casegen.rand_seq(rng, int(rng.choice([2, 3, 8, 9, 10])))
                if te_type == "tir" and kind in (0, 1):          # a terminal inverted repeat itrsearch finds
                    t = casegen.rand_seq(rng, int(rng.integers(12, 40)))
                    body = t + body[len(t):-len(t)] + casegen.revcomp(casegen.mutate(rng, t, 0.05))
                if te_type == "tir" and kind == 2 and len(tsd) in (8, 9, 10):   # short-TIR signature (get_short_tir_contigs)
                    body = body[:-5] + casegen.revcomp(body[:5])
                qn = ("N_%d-tir_%d-tsd_%s" % (q, int(rng.integers(0, 30)), tsd)) if te_type == "tir" else "N_%d" % q
                cc = int(rng.choice([1, 2, 3, 4, 5, 6, 9, 40]))
                table[qn] = (qn, body, "", cc)
                copies[qn] = [(names[0], 100, 400, 301, "+")]
                if kind >= 3 or rng.random() < 0.3:              # fabricated blastx fragments against one or two proteins
                    lines = []
                    for pn in [pnames[int(rng.integers(0, len(pnames)))] for _ in range(int(rng.integers(1, 3)))]:
                        plen = len(prot[pn])
                        cover = float(rng.choice([0.4, 0.8, 0.94, 0.96, 1.0]))
                        nfrag = int(rng.integers(1, 5))
                        span = max(nfrag, int(plen * cover))
                        cuts = sorted(set([0, span] + [int(x) for x in rng.integers(1, max(2, span), size=nfrag - 1)]))
                        p0 = 1 + int(rng.integers(0, max(1, plen - span)))
                        q0 = int(rng.integers(1, max(2, len(body) - 3 * span - 400)))
                        minus = rng.random() < 0.4
                        frs = []
                        for a, b in zip(cuts[:-1], cuts[1:]):
                            if b - a < 2:
                                continue
                            gapq = int(rng.choice([0, 0, 20, 60, 150]))
                            gaps = int(rng.choice([0, 0, 5, 20, 40]))
                            qs_, qe_ = q0 + 3 * a + gapq, q0 + 3 * b + gapq - 1
                            ss_, se_ = p0 + a + gaps, min(plen, p0 + b - 1 + gaps)
                            if minus:
                                qs_, qe_ = len(body) - qs_, len(body) - qe_
                            frs.append((qs_, qe_, ss_, se_))
                        if rng.random() < 0.3:
                            frs.reverse()
                        for (qs_, qe_, ss_, se_) in frs:
                            lines.append("%s\t%s\t%.2f\t%d\t0\t0\t%d\t%d\t%d\t%d\t1e-30\t200\n" % (qn, pn, 60 + 30 * rng.random(), se_ - ss_ + 1, qs_, qe_, ss_, se_))
                    blast[qn] = lines
            cand = os.path.join(tmp, "lc_%s_%d.fa" % (te_type, rep))
            write_fasta(cand, list(table.keys()), ["ACGT" * 30 for _ in table])
            fake = os.path.join(tmp, "lcr_blastx_%s_%d.json" % (te_type, rep))
            with open(fake, "w") as f:
                _json.dump(blast, f)

            def fake_copies(query_path, reference, temp_dir, max_copy_num, threads, _c=copies):
                os.makedirs(temp_dir, exist_ok=True)
                return _c

            def fake_members(task, temp_dir, subset_script_path, plant, TE_type, debug, result_type, _t=table):
                (query_name, cur_seq, trunc_member_file, extend_member_file) = task
                r = _t[query_name]
                return (r[0], r[1], r[2], r[3], extend_member_file)

            saved = (U.get_full_length_copies_minimap2, U.run_find_members_v8, U.ProcessPoolExecutor, U.as_completed, U.cur_dir)
            saved_env = (os.environ.get("PATH"), os.environ.get("HITE_FAKE_BLASTX"))
            U.get_full_length_copies_minimap2 = fake_copies
            U.run_find_members_v8 = fake_members
            U.ProcessPoolExecutor = ref_harness.SyncExecutor
            U.as_completed = lambda fs: fs
            U.cur_dir = root
            os.environ["PATH"] = bindir + os.pathsep + saved_env[0]
            os.environ["HITE_FAKE_BLASTX"] = fake
            real = os.path.join(tmp, "lreal_%s_%d.fa" % (te_type, rep))
            low = os.path.join(tmp, "llow_%s_%d.fa" % (te_type, rep))
            work = os.path.join(tmp, "lw_%s_%d" % (te_type, rep))
            open(low, "w").close()
            try:
                log = type("L", (), {"logger": type("LL", (), {"info": staticmethod(lambda *a: None), "debug": staticmethod(lambda *a: None)})})()
                U.flank_region_align_v5(cand, real, 50, ref, None, te_type, work, 1, 0, log, "", rep, 1, 0, low)     # (debug = 1: the working directory stays)
            finally:
                (U.get_full_length_copies_minimap2, U.run_find_members_v8, U.ProcessPoolExecutor, U.as_completed, U.cur_dir) = saved
                os.environ["PATH"] = saved_env[0]
                if saved_env[1] is None:
                    os.environ.pop("HITE_FAKE_BLASTX", None)
            lcd = os.path.join(work, "%s_copies_0_0" % te_type, "low_copy_itr")
            tables = [x for x in sorted(os.listdir(lcd)) if x.endswith("_domain")]
            assert len(tables) == 1, tables
            masked = [x for x in os.listdir(lcd) if x.endswith(".mask")]
            assert masked, "TRF did not run"
            mn, mc = U.read_fasta(os.path.join(lcd, masked[0]))
            rn, rc = U.read_fasta(real)
            cases.append(dict(te_type=te_type, plant=rep, table=[[k, v[0], v[1], v[2], v[3]] for k, v in table.items()], library=libs[fn],
                              blastx=blast, real=[[x, rc[x]] for x in rn], low_text=open(low).read(), domain_table=open(os.path.join(lcd, tables[0])).read(),
                              trf_changed=[k for k in mn if mc[k] != table[k][1]]))
            print("low_copy_rescue %s %d: %d candidates -> %d real TEs, %d low copy; %d domain rows; TRF changed %d" %
                  (te_type, rep, len(table), len(rn), open(low).read().count(">"), open(os.path.join(lcd, tables[0])).read().count("\n") - 2,
                   len(cases[-1]["trf_changed"])))
    dump("low_copy_rescue", cases)


def main():
    assert os.environ.get("PYTHONHASHSEED") == "0", "run with PYTHONHASHSEED=0"
    U = ref_harness.load_reference_util()
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["fmea", "judge", "search", "tsd", "kmer", "gather", "tails", "host", "ltr", "nonltr", "qcopies", "libdedup", "bothends", "split", "bucketing", "consv1", "trf", "rfm", "chainvar", "edge", "itr", "lcr"]
    with tempfile.TemporaryDirectory() as tmp:
        if "fmea" in which:
            gen_fmea(U, tmp)
        if "judge" in which:
            gen_judge(U, tmp)
        if "search" in which:
            gen_boundary_search(U)
        if "tsd" in which:
            gen_tsd(U)
        if "kmer" in which:
            gen_tir_kmer(U)
        if "gather" in which:
            gen_gather(U, tmp)
        if "tails" in which:
            gen_tails(U)
        if "host" in which:
            gen_host(U)
        if "ltr" in which:
            gen_ltr_frame(tmp)
        if "nonltr" in which:
            gen_nonltr_prep(U)
        if "qcopies" in which:
            gen_query_copies(U, tmp)
        if "libdedup" in which:
            gen_lib_dedup(U, tmp)
        if "bothends" in which:
            gen_ltr_both_ends(tmp)
        if "split" in which:
            gen_split_chunks(U, tmp)
        if "bucketing" in which:
            gen_bucketing(U, tmp)
        if "consv1" in which:
            gen_cons_v1(U, tmp)
        if "trf" in which:
            gen_trf_mask(tmp)
        if "rfm" in which:
            gen_ready_for_msa(tmp)
        if "chainvar" in which:
            gen_chain_variants(U, tmp)
        if "edge" in which:
            gen_judge_edge(U, tmp)
        if "itr" in which:
            gen_itr_search(U, tmp)
        if "lcr" in which:
            gen_low_copy_rescue(U, tmp)


if __name__ == "__main__":
    main()
