"""The CPU oracle (oracle/*.c) must reproduce every golden vector generated from the
reference's own Python (oracle/gen_golden.py).  CPU only."""
import os
import numpy as np
import pytest

import casegen
import oracle_lib as O
from conftest import load_golden


def test_fmea_golden():
    for case in load_golden("fmea"):
        h = O.hsp_arrays([tuple(r) for r in case["rows"]])
        got = O.fmea(h, case["skip_gap"], case["max_len"])
        assert got == case["expected"]


def test_sparse_cols_golden():
    n = 0
    for name in ("judge_tir", "judge_non_ltr", "judge_helitron"):
        for case in load_golden(name):
            msa = O.msa_array(case["seqs"])
            keep = O.sparse_cols(msa).astype(bool)
            clean = ["".join(chr(c) for c in row[keep]) for row in msa]
            assert clean == case["clean"]
            n += 1
    assert n > 100


@pytest.mark.parametrize("name,te_type", [("judge_tir", "tir"), ("judge_non_ltr", "non_ltr"), ("judge_helitron", "helitron")])
def test_judge_golden(name, te_type):
    ntrue = 0
    for i, case in enumerate(load_golden(name)):
        msa = O.msa_array(case["clean"])
        got, _ = O.judge(te_type, msa, case["cand"], case["plant"])
        exp = case["expected"]
        if exp[0] == "EXC":
            assert got[0] == "EXC", (i, got, exp)
        else:
            assert got == exp, (i, got, exp)
            ntrue += bool(exp[0])
    assert ntrue >= 2


def test_judge_edge_golden():
    """round 4: the cases aimed at the reference lines the earlier fixtures never reached (tools/ref_line_coverage.py): anchors
    and homology next to the alignment's edges, no flank at all, '-' majorities inside the element, the TA / TAA / TTAA trims"""
    cases = load_golden("judge_edge")
    regen = casegen.msa_edge_cases(61)
    assert len(cases) == len(regen)
    seen = {}
    for i, (case, r) in enumerate(zip(cases, regen)):
        assert case["seqs"] == r["seqs"] and case["cand"] == r["cand"] and case["aim"] == r["aim"]      # the fixture's inputs are the generator's
        msa = O.msa_array(case["seqs"])
        keep = O.sparse_cols(msa).astype(bool)
        assert ["".join(chr(c) for c in row[keep]) for row in msa] == case["clean"], i
        got, _ = O.judge(case["te_type"], O.msa_array(case["clean"]), case["cand"], case["plant"])
        exp = case["expected"]
        if exp[0] == "EXC":
            assert got[0] == "EXC", (i, got, exp)
        else:
            assert got == exp, (i, case["aim"], got, exp)
        seen[(case["te_type"], bool(exp[0] is True))] = seen.get((case["te_type"], bool(exp[0] is True)), 0) + 1
    assert all(seen.get((t, v), 0) >= 5 for t in ("tir", "non_ltr", "helitron") for v in (True, False)), seen


def test_boundary_search_edge_golden():
    cases = load_golden("boundary_search_edge")
    assert len(cases) > 1500 and sum(c["v3"] != -1 for c in cases) > 300 and sum(c["v3"] == -1 for c in cases) > 300
    for i, case in enumerate(cases):
        msa = O.msa_array(case["seqs"])
        thr = case["thr"]
        assert O.search_v3(msa, case["pos"], case["side"], thr) == case["v3"], i
        v, b = O.search_v4(msa, case["pos"], case["side"], thr, thr - 0.05, thr)
        assert [v, b] == case["v4"], i


def test_boundary_search_golden():
    for i, case in enumerate(load_golden("boundary_search")):
        msa = O.msa_array(case["seqs"])
        thr = case["thr"]
        assert O.search_v3(msa, case["pos"], case["side"], thr) == case["v3"], i
        v, b = O.search_v4(msa, case["pos"], case["side"], thr, thr - 0.05, thr)
        assert [v, b] == case["v4"], i


def test_threshold_ties_golden():
    for case in load_golden("thr_ties"):
        msa = O.msa_array(case["seqs"])
        W = msa.shape[1]
        assert O.window_homology(msa, 0, W, 1, case["thr"]) == case["fwd"]
        assert O.window_homology(msa, W - 1, W - 2, -1, case["thr"]) == case["rev"]


def test_threshold_ties_through_search_golden():
    """the binary64 ties of thr / thr - 0.1 reached through search_boundary_homo_v3 / _v4 (240 reference runs)"""
    cases = load_golden("thr_ties_search")
    assert sum(c["v3"] != -1 for c in cases) > 50 and sum(c["v3"] == -1 for c in cases) > 50
    for i, case in enumerate(cases):
        msa = O.msa_array(case["seqs"])
        thr = case["thr"]
        assert O.search_v3(msa, case["pos"], case["side"], thr) == case["v3"], i
        v, b = O.search_v4(msa, case["pos"], case["side"], thr, thr - 0.05, thr)
        assert [v, b] == case["v4"], i


def test_tsd_search_golden():
    for case in load_golden("tsd_search"):
        got = O.tsd_search_v5(case["seq"], case["start"], case["end"], case["plant"])
        assert got == (case["left"], case["right"])


def test_tails_golden():
    for case in load_golden("tails"):
        assert O.find_tail_polyA(case["seq"]) == case["polyA"]
        assert O.find_tandem_tail(case["seq"]) == case["tandem"]


def check_tir_items(items, case):
    """The reference keeps the top 100 by distance; which of the candidates tied at the cut
    distance survive depends on PYTHONHASHSEED (set iteration, Util.py:7741/:7807), so at a
    full cut only the strictly-closer part is pinned."""
    exp = case["items"]
    assert len(items) == len(exp) == case["n"]
    if case["n"] < 100:
        assert items == exp
    else:
        dcut = exp[-1][0]
        assert [x for x in items if x[0] < dcut] == [x for x in exp if x[0] < dcut]
        assert all(x[0] == dcut for x in items if x[0] >= dcut)


def test_tir_kmer_golden():
    for case in load_golden("tir_kmer"):
        seq, flank = case["seq"], case["flank"]
        recs = O.tir_kmer(seq, flank + 1, len(seq) - flank, flank, case["plant"])
        items = sorted([d, seq[ts - k:ts], seq[ts:te + 1]] for (k, ts, te, d) in recs)
        check_tir_items(items, case)


def test_tir_kmer_edge_golden():
    """round 4: candidates barely longer than their flanks (windows that run past the sequence), 'NN' as a k-mer on both sides, elements
    that start with TATATATA / ATATATAT (tools/ref_line_coverage.py: the lines of search_confident_tir_v4 the first fixture missed)"""
    cases = load_golden("tir_kmer_edge")
    assert len(cases) >= 90 and sum(c["n"] > 0 for c in cases) >= 40 and sum(c["n"] == 0 for c in cases) >= 20
    for case in cases:
        seq, flank = case["seq"], case["flank"]
        recs = O.tir_kmer(seq, flank + 1, len(seq) - flank, flank, case["plant"])
        items = sorted([d, seq[ts - k:ts], seq[ts:te + 1]] for (k, ts, te, d) in recs)
        check_tir_items(items, case)


def test_gather_golden():
    for case in load_golden("gather"):
        contigs = dict(zip(case["names"], case["seqs"]))
        for q, copies in case["copies"].items():
            ext, trunc = [], []
            seen = {}
            for (chrom, s, e, _alen, strand) in copies:
                w, t = O.flank_window(contigs[chrom], s, e, strand, case["flank"])
                if w is None:
                    continue
                name = "%s:%d-%d(%s)" % (chrom, s, e, strand)
                seen[name] = (w, t)  # dict semantics: later duplicate names overwrite in place
            ext = [[k, v[0]] for k, v in seen.items()]
            trunc = [[k, v[1]] for k, v in seen.items() if v[1] is not None]
            exp = case["expected"].get(q)
            if exp is None:
                assert ext == []
                continue
            assert ext == exp["extend"]
            assert (trunc or None) == exp["trunc"]
        # flanking_seq
        for name, (oname, oseq) in zip(case["flanking_in"], case["flanking_out"]):
            chrom, pos = name.split(":")
            s, e = map(int, pos.split("-"))
            lo, hi, ns, ne = O.flanking_seq(s, e, len(contigs[chrom]), 50)
            assert "%s:%d-%d" % (chrom, ns, ne) == oname
            assert contigs[chrom][lo:hi] == oseq


def test_seed_allvsall_twin_recovers_planted_pairs():
    """own-design stage (blastn stand-in): planted copies that differ by <= ~8 % (any two copies of a 2 % family, the
    unmutated copy against the others of an 8 % family) are joined by HSPs that cover most of the query copy, on the right
    strand; records respect the blast6 conventions and segment borders"""
    import synth_small

    g = synth_small.make(31, n_fam=10, n_chr=2, chr_len=260_000)
    seg_len = 100_000
    h = O.seed_allvsall(g["contigs"], seg_len=seg_len)
    n = len(h["qseg"])
    assert n > 50
    assert (h["qs"] >= 1).all() and (h["qe"] <= seg_len).all() and (h["qs"] <= h["qe"]).all()
    assert (np.minimum(h["ss"], h["se"]) >= 1).all() and (np.maximum(h["ss"], h["se"]) <= seg_len).all()
    key = h["qseg"].astype(np.int64) * 65536 + h["sseg"]
    assert (np.diff(key) >= 0).all()
    # genome coordinates of every record
    qchr, schr = h["seg_chrom"][h["qseg"]], h["seg_chrom"][h["sseg"]]
    qa, qb = h["seg_off"][h["qseg"]] + h["qs"], h["seg_off"][h["qseg"]] + h["qe"]
    sa = h["seg_off"][h["sseg"]] + np.minimum(h["ss"], h["se"])
    sb = h["seg_off"][h["sseg"]] + np.maximum(h["ss"], h["se"])
    minus = h["ss"] > h["se"]
    found = total = 0
    for fam, div in zip(g["truth"], g["divs"]):
        if div > 0.08 or len(fam) < 2:
            continue
        for i, (c1, a1, b1, m1) in enumerate(fam[:6]):
            for j, (c2, a2, b2, m2) in enumerate(fam[:6]):
                if i == j or (div > 0.02 and i != 0 and j != 0):
                    continue
                total += 1
                sel = (qchr == c1) & (schr == c2) & (qa < b1) & (qb > a1) & (sa < b2) & (sb > a2) & (minus == (m1 != m2))
                cov = 0
                if sel.any():
                    cov = int((np.minimum(qb[sel], b1) - np.maximum(qa[sel], a1) + 1).sum())
                found += cov >= 0.6 * (b1 - a1 + 1)
    assert total > 20 and found >= 0.9 * total, (found, total)


def test_host_glue_golden():
    """host-side glue of the stage wrappers against the reference's outputs: terminal-structure shortcuts,
    min-distance variant, query-file grouping, reverse complement"""
    import sys

    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    from hite_amd import util

    g = load_golden("host_glue")
    for c in g["short_tir"]:
        got = util.get_short_tir_contigs(dict(zip(c["names"], c["seqs"])), c["plant"])
        assert list(got.keys()) == c["kept"]
    assert sum(len(c["kept"]) for c in g["short_tir"]) > 10
    for c in g["filter_dup"]:
        got = util.filter_dup_itr_v3(dict(zip(c["names"], c["seqs"])), c["tir_len"])
        assert list(got.keys()) == c["out_names"] and list(got.values()) == c["out_seqs"]
    for c in g["split"]:
        assert util.split_and_store_sequences(c["names"], {n: "A" * l for n, l in zip(c["names"], c["lens"])}, c["thr"]) == c["groups"]
    for s, r in g["revcomp"]:
        assert util.getReverseSequence(s) == r


def test_ltr_frame_golden():
    """FiLTR flank-frame vote (judge_left/right_frame_LTR) restatement vs the reference's outputs"""
    g = load_golden("ltr_frame")
    seen = set()
    for c in g:
        assert list(O.ltr_frame(c["left"], c["flank"], c["window"], "left")) == c["left_out"]
        assert list(O.ltr_frame(c["right"], c["flank"], c["window"], "right")) == c["right_out"]
        seen.add((tuple(c["left_out"])[0], c["left_out"][1] >= 0, c["right_out"][0], c["right_out"][1] >= 0))
    assert len(seen) >= 5


def test_nonltr_prep_golden():
    """search_polyA_TSD restatement (polyA / polyT / tandem tails + 8-20 bp TSD near the 5' end) vs the reference"""
    g = load_golden("nonltr_prep")
    kinds = set()
    for c in g:
        got = O.search_polyA_TSD(c["seq"], c["flank"], 25)
        assert list(got) == [c["found"], c["tsd"], c["non_ltr"]], (c["seq"][:60], got, (c["found"], c["tsd"], c["non_ltr"][:40]))
        kinds.add((c["found"], bool(c["non_ltr"])))
    assert len(kinds) == 3


def _query_copies_case(c):
    """golden case -> (rows with dense first-appearance query ids, qlen in that order, expected per query)"""
    order, rows = {}, []
    for r in c["rows"]:
        q = order.setdefault(r[0], len(order))
        rows.append((q, r[1], r[2], r[3], r[4], r[5], r[6]))
    qlen = [0] * len(order)
    for orig, q in order.items():
        qlen[q] = c["qlen"][orig]
    exp = [[(int(x[0][3:]), x[1], x[2], x[3], x[4]) for x in c["out"].get("TE_%d" % orig, [])] for orig, q in sorted(order.items(), key=lambda kv: kv[1])]
    return rows, qlen, exp


def test_query_copies_golden():
    """get_query_copies restatement (blastn-route copy clustering) vs the reference's outputs"""
    g = load_golden("query_copies")
    total = 0
    for c in g:
        rows, qlen, exp = _query_copies_case(c)
        assert O.query_copies(rows, qlen, c["slen"], c["qcov"], c["scov"]) == exp
        total += sum(len(e) for e in exp)
    assert total > 100


def test_lib_dedup_golden():
    """panHiTE library de-duplication (f-3): chain records, greedy clusters and majority consensus vs the reference's outputs"""
    g = load_golden("lib_dedup")
    nrec = ncl = 0
    for c in g["chain"]:
        recs = O.lib_chain(c["rows"], c["lens"], c["thr"], c["chunk_size"])
        assert recs == c["recs"]
        cl = O.lib_cluster(recs, c["lens"], c["thr"])
        assert [sorted(x) for x in cl] == c["clusters"]
        nrec += len(recs); ncl += len(cl)
    assert nrec > 300 and ncl > 50
    for c in g["cons"]:
        assert O.cons_majority([r.upper() for r in c["rows"]]) == c["cons"]


def test_ltr_both_ends_golden():
    """FiLTR get_both_ends_frame (anchors, its sparse-column rule, `.matrix` frames, full-length rows) vs the reference"""
    g = load_golden("ltr_both_ends")
    found = 0
    for c in g:
        rows = [r.upper() for r in c["rows"]]
        got = O.ltr_both_ends(rows, c["cur"], c["flank"])
        if c["frames"] is None:
            assert got is None
            continue
        assert got is not None and not isinstance(got, int)
        assert got[0] == c["frames"] and got[1] == c["full"]
        found += 1
    assert found > 40 and found < len(g)


def test_fmea_stress_hash():
    """30 k-line HSP table: the ordered interval names of the reference are pinned by their sha256"""
    import hashlib

    g = load_golden("fmea_stress")
    rows = casegen.make_hsp_table(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in g["params"].items()})
    assert len(rows) == g["lines"]
    names = O.fmea(O.hsp_arrays([tuple(r) for r in rows]), g["skip_gap"], g["max_len"])
    assert len(names) == g["intervals"]
    assert hashlib.sha256("\n".join(names).encode()).hexdigest() == g["sha256"]


def test_host_formats_golden(tmp_path):
    """on-disk format helpers vs the reference's outputs: rename_fasta, rename_reference, lib_add_prefix,
    file_exist, update_prev_TE"""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from hite_amd import util

    g = load_golden("host_formats")
    src = tmp_path / "in.fa"
    c = g["rename_fasta"]
    src.write_text(c["in"])
    util.rename_fasta(str(src), str(tmp_path / "out.fa"), c["header"])
    assert (tmp_path / "out.fa").read_text() == c["out"]
    c = g["rename_reference"]
    util.rename_reference(str(src), str(tmp_path / "ref.fa"), str(tmp_path / "map.txt"))
    assert (tmp_path / "ref.fa").read_text() == c["out"] and (tmp_path / "map.txt").read_text() == c["map"]
    c = g["lib_add_prefix"]
    lib = tmp_path / "lib.fa"
    lib.write_text(c["in"])
    assert util.lib_add_prefix(str(lib), c["prefix"]) == str(lib) and lib.read_text() == c["out"]
    c = g["file_exist"]
    for name, body, exp in c["files"]:
        p = tmp_path / name
        p.write_text(body)
        assert util.file_exist(str(p)) == exp, name
    (tmp_path / "emptydir").mkdir(); (tmp_path / "fulldir").mkdir(); (tmp_path / "fulldir" / "f").write_text("1")
    assert util.file_exist(str(tmp_path / "emptydir")) == c["emptydir"] and util.file_exist(str(tmp_path / "fulldir")) == c["fulldir"]
    assert util.file_exist(str(tmp_path / "nope")) == c["missing"]
    c = g["update_prev_TE"]
    prev, cur = tmp_path / "prev_TE.fa", tmp_path / "cur.fa"
    prev.write_text(c["prev"]); cur.write_text(c["cur"])
    util.update_prev_TE(str(prev), str(cur))
    util.update_prev_TE(str(prev), str(tmp_path / "absent.fa"))
    assert prev.read_text() == c["out"]


def test_host_fasta_golden(tmp_path):
    """read_fasta (Util.py:1650) on the cases that make it differ from a naive reader -- headers without sequence, blank sequence
    lines, text before the first header, a name twice, a missing file --, file_exist on the same files, and the block grouping of
    split_genome_chunks.py (split_chromosomes :10252, split_dict_into_blocks :10276) on 40 random length lists: the reference's own
    results (round 6: the host helpers were re-written away from the reference's statement order; this pins them)"""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from hite_amd import util

    g = load_golden("host_fasta")
    for i, c in enumerate(g["read_fasta"]):
        p = tmp_path / ("r%d.fa" % i)
        if c["text"] is not None:
            p.write_text(c["text"])
        names, contigs = util.read_fasta(str(p))
        assert names == c["names"] and contigs == c["contigs"], i
        assert util.file_exist(str(p)) == c["exists"], i
    for c in g["blocks"]:
        cd = {"c%d" % i: "A" * l for i, l in enumerate(c["lens"])}
        parts = util.split_chromosomes(dict(cd), c["chunk"])
        assert [[k, len(v)] for k, v in parts.items()] == c["parts"]
        blocks = util.split_dict_into_blocks(dict(cd), c["threads"], c["chunk"])
        assert [[[k, len(v)] for k, v in b.items()] for b in blocks] == c["blocks"], c


def test_split_genome_chunks_golden(tmp_path):
    """f-1: the drop-in of module/split_genome_chunks.py writes byte-identical genome.cut{i}.fa / ref_chr/ref_block_{i}.fa and
    rewrites the genome like the reference (upper case, one line per contig, names cut at the first blank)"""
    import subprocess
    import sys as _sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for ci, case in enumerate(load_golden("split_chunks")):
        d = tmp_path / ("c%d" % ci)
        d.mkdir()
        ref = d / "genome.fa"
        ref.write_text(case["input"])
        rc = subprocess.run([_sys.executable, os.path.join(root, "hite_amd", "scripts", "split_genome_chunks.py"), "-g", str(ref),
                             "--tmp_output_dir", str(d), "--chrom_seg_length", str(case["chrom_seg_length"]), "--chunk_size",
                             str(case["chunk_size"])], capture_output=True, text=True)
        assert rc.returncode == 0, rc.stderr[-1500:]
        got = {}
        for fn in sorted(os.listdir(d)):
            if fn.startswith("genome.cut") and fn.endswith(".fa"):
                got[fn] = (d / fn).read_text()
        for fn in sorted(os.listdir(d / "ref_chr")):
            if fn.endswith(".fa"):
                got["ref_chr/" + fn] = (d / "ref_chr" / fn).read_text()
        got["genome.fa"] = ref.read_text()
        assert got == case["files"]


def test_result_bucketing_golden():
    """a-22: which consensus is a real TE, which goes to the low-copy file, which is dropped -- the reference's collection loop
    (Util.py:8159-8194, 8282-8287) run on tables of result tuples"""
    from hite_amd import util

    n_real = n_low = 0
    for case in load_golden("bucketing"):
        true_tes, low = util.bucket_results(case["te_type"], [(r[1], r[2], r[3], r[4]) for r in case["table"]])
        assert [[k, v] for k, v in true_tes.items()] == case["real"]
        exp_low = case["low_text"].split(">earlier\nACGT\n", 1)[1]
        assert "".join(">%s\n%s\n" % (k, v) for k, v in low.items()) == exp_low
        n_real += len(true_tes)
        n_low += len(low)
    assert n_real > 50 and n_low > 30


def test_low_copy_tir_recall_by_structure(tmp_path):
    """rescue_low_copy (Util.py:8196-8213 + remove_no_tirs :13897): low-copy TIR candidates with a short-TIR signature are real
    TEs, the others are when the terminal-inverted-repeat search (the in-tree stage where the reference runs itrsearch) finds
    one; Helitron / non-LTR candidates stay low copy (the blastx domain recall is external)"""
    from hite_amd import util

    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    core = casegen.rand_seq(np.random.default_rng(8196), 400)
    head = "GGGTC"
    tail = "".join(comp[c] for c in reversed(head))
    low = {"N_1-C_0-tsd_ACGTACGT-distance_0": head + core + tail,        # 8-bp TSD (hAT), first 5 = revcomp(last 5), < 4 kb
           "N_2-C_0-tsd_ACGTACGT-distance_0": "TTTTT" + core + "CCCCC",  # no terminal inverted repeat
           "N_3-C_0-tsd_AC-distance_0": head + core + tail}              # 2-bp TSD: no short-TIR family
    assert util.get_short_tir_contigs(low, 1).keys() == {"N_1-C_0-tsd_ACGTACGT-distance_0"}
    from oracle_ctx import OracleCtx
    # (the tandem-repeat step in front of the recall needs the GPU masker or trf: switched off here, this test is about the glue;
    # the terminal-inverted-repeat search answers from its twin)
    rescued, still = util.rescue_low_copy("tir", low, 1, str(tmp_path / "lc"), tandem_masker=lambda names, contigs: dict(contigs), ctx=OracleCtx())
    with_itr = {n for n, r in zip(low, O.itr_search(list(low.values()), 0)) if r[5]}
    assert "N_1-C_0-tsd_ACGTACGT-distance_0" in rescued
    assert set(rescued) == {"N_1-C_0-tsd_ACGTACGT-distance_0"} | (with_itr - {"N_1-C_0-tsd_ACGTACGT-distance_0"})
    assert set(rescued) | set(still) == set(low) and not set(rescued) & set(still)
    assert "N_2-C_0-tsd_ACGTACGT-distance_0" in still     # random core, TTTTT ... CCCCC: nothing inverted at its ends
    for te in ("helitron", "non_ltr"):
        r2, s2 = util.rescue_low_copy(te, low, 1, str(tmp_path / te))
        assert r2 == {} and s2 == low


def test_generate_cons_v1_golden():
    """generate_cons_v1 (Util.py:12457) run by the reference with `mafft` = the oracle's star alignment and a fabricated Ninja
    file: the oracle pieces (star alignment around the longest member, strict-majority consensus) chained the same way give
    the reference's dict -- naming by the last member of each sub-cluster included"""
    cases = load_golden("cons_v1")
    assert len(cases) >= 10 and sum(len(c["expected"]) for c in cases) >= 20
    for ci, c in enumerate(cases):
        seq_of = dict(zip(c["names"], c["seqs"]))
        got = {}
        for k in c["ninja"]:
            members = c["ninja"][k]
            seqs = [seq_of[n] for n in members]
            centre = max(range(len(seqs)), key=lambda i: (len(seqs[i]), -i))
            order = [centre] + [i for i in range(len(seqs)) if i != centre]
            m, kept = O.star_msa([seqs[i] for i in order], rows=True)
            assert kept == len(seqs)
            back = {i: q for q, i in enumerate(order)}
            rows = [bytes(m[back[i]]).decode() for i in range(len(seqs))]
            got[members[-1]] = O.cons_majority(rows)
        if not c["ninja"]:          # no cluster came out of Ninja: the original sequences (Util.py:12495-12498)
            got = seq_of
        assert got == c["expected"], ci


def test_ready_for_msa_golden():
    """tools/ready_for_MSA.sh <members.fa> 100 100 as the reference runs it (is_TE_from_align_file, Util.py:10410): the rows the
    script keeps == oracle_pipeline.select_rows with the window names -- the 100 longest, equal lengths in descending byte
    order of the name, output in input order"""
    import oracle_pipeline as OP

    cases = load_golden("ready_for_msa")
    assert len(cases) == 10
    ties_decided = 0
    for ci, c in enumerate(cases):
        keep = OP.select_rows(c["lens"], c["names"])
        assert [c["names"][i] for i in keep] == c["selected"], ci
        if len(c["names"]) > 100:
            cut = sorted(c["lens"], reverse=True)[99]
            ties_decided += sum(1 for x in c["lens"] if x == cut) > sum(1 for i in keep if c["lens"][i] == cut)
    assert ties_decided >= 6          # the cut falls inside a run of equal lengths in most cases: the tie rule is what is pinned


def test_stretch_hits_looks_at_the_bases():
    """util._stretch_hits (host arithmetic of the library merge): a hit grows over the overhang only by the bases that align
    -- the same core with unrelated 8 % termini must stay below a 0.95 coverage rule, copies of one family reach the ends, on
    either strand"""
    import numpy as np
    from hite_amd import util as U

    rng = np.random.default_rng(7)
    rnd = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))  # noqa: E731
    rc = lambda x: x[::-1].translate(str.maketrans("ACGT", "TGCA"))  # noqa: E731

    def mutate(x, frac):
        a = list(x)
        for i in rng.choice(len(a), int(len(a) * frac), replace=False):
            a[i] = "ACGT"[("ACGT".index(a[i]) + 1 + int(rng.integers(0, 3))) % 4]
        return "".join(a)

    for trial in range(20):
        core = rnd(800)
        a = rnd(80) + core + rnd(80)
        b = rnd(80) + core + rnd(80)          # same core, different termini
        c = mutate(a, 0.10)                   # a copy of a, 10 % substitutions
        seqs = [a, b, c, rc(c)]
        lens = [len(x) for x in seqs]
        # hits over the core only (what the anchors of the seeding stage span), 1-based inclusive; the last one on the minus strand
        qs, qe, ss, se = U._stretch_hits([0, 0, 0], [1, 2, 3], [81, 81, 81], [880, 880, 880], [81, 81, 880], [880, 880, 81], lens, seqs)
        assert (qe[0] - qs[0] + 1) / 960 < 0.95 and abs(se[0] - ss[0]) + 1 == qe[0] - qs[0] + 1, (trial, qs, qe)
        assert (qe[1] - qs[1] + 1) / 960 >= 0.97 and ss[1] == qs[1] and se[1] == qe[1], (trial, qs, qe, ss, se)
        assert (qe[2] - qs[2] + 1) / 960 >= 0.97 and ss[2] == 961 - qs[2] and se[2] == 961 - qe[2], (trial, qs, qe, ss, se)
    # an overhang beyond the reach (a tenth of the shorter sequence, at least 30) is left alone
    a = rnd(1000)
    qs, qe, ss, se = U._stretch_hits([0], [1], [201], [800], [201], [800], [1000, 1000], [a, a])
    assert (int(qs[0]), int(qe[0])) == (201, 800)


def test_library_merge_host_code_through_the_twins(tmp_path):
    """deredundant_for_LTR_v5's HOST code (hits -> stretch -> chains -> clusters -> sub-clusters -> consensus -> cd-hit stand-in)
    driven with every device stage answered by its CPU twin (tests/oracle_ctx.py) -- the second leg of the C5 parity test
    (tests/test_gpu_scale.py), checked here where there is no GPU: planted families collapse, singletons pass"""
    from hite_amd import util
    from oracle_ctx import OracleCtx

    rng = np.random.default_rng(515)
    recs, fams = [], []
    for f in range(6):
        cons = casegen.rand_seq(rng, int(rng.integers(400, 1500)))
        fams.append(cons)
        for g in range(int(rng.integers(3, 7))):
            recs.append(("G%d-fam%d#DNA/hAT" % (g, f), casegen.mutate(rng, cons, float(rng.uniform(0.01, 0.04)))))
    for k in range(5):
        recs.append(("single%d#Unknown" % k, casegen.rand_seq(rng, int(rng.integers(300, 900)))))
    merged = str(tmp_path / "m.fa")
    util.store_fasta(dict(recs), merged)
    st = {}
    out = util.deredundant_for_LTR_v5(merged, str(tmp_path), 1, "terminal", 0.95, 0, ctx=OracleCtx(), stages=st)
    assert out == merged + ".tmp.cons" and st["hits"] > 50
    assert sum(len(c) > 1 for c in st["clusters"]) == 6           # (a member the greedy clustering leaves out falls to the cd-hit stand-in)
    names, seqs = util.read_fasta(merged + ".cons")
    assert sum(n.startswith("single") for n in names) == 5 and len(names) == 11
    for f, cons in enumerate(fams):
        mine = [n for n in names if "-fam%d#" % f in n]
        assert len(mine) == 1 and abs(len(seqs[mine[0]]) - len(cons)) <= 0.02 * len(cons)


def test_chain_variants_golden_through_the_twins(tmp_path):
    """FMEA (Util.py:10452), get_full_length_copies_from_blastn_v1 (:5907), generate_full_length_out_v1 (:6288) and
    multiple_alignment_blast_and_get_copies_v1 (:7179): the product's host mirrors with the chaining answered by the CPU twin
    (orc_chain_all / orc_query_copies) reproduce what the reference computed (tests/golden/chain_variants.json.gz)"""
    import chain_variant_cases
    from hite_amd import util
    from oracle_ctx import OracleCtx

    n_chains, n_copies = chain_variant_cases.check_all(util, OracleCtx(), str(tmp_path))
    assert n_chains > 100 and n_copies > 50


def test_itr_search_twin_equals_the_tool():
    """oracle/hite_oracle_itr.c (read from the disassembly of tools/itrsearch) against the tool's own output on 3 300 seeded records:
    which records it writes to <input>.itr and the "Length itr=" of their headers (run_itrsearch, Util.py:216-224)"""
    import itr_cases

    assert itr_cases.check_tool_records(lambda seqs, e: O.itr_search(seqs, e)) > 1500


def test_itr_filter_host_mirrors_golden():
    """the product's host mirrors around the filter -- search_confident_tir_batch_v1 (Util.py:6533-6628) and remove_no_tirs
    (Util.py:13897-13920) -- with the twins behind them, against the reference's own runs WITH the tool"""
    import itr_cases
    from hite_amd import util
    from oracle_ctx import OracleCtx

    ctx = OracleCtx()
    n_q, n_drop = itr_cases.check_batches(util, ctx)
    assert n_drop > n_q
    itr_cases.check_rescue(util, ctx)


def test_low_copy_rescue_golden(tmp_path, monkeypatch):
    """the low-copy recall (Util.py:8196-8287) against the reference's own run with its tools (TRF 4.09, itrsearch, get_domain_info over
    a fabricated blastx table): real_TEs, all_low_copy and the domain table.  Here: bucket_results + rescue_low_copy with the
    terminal-inverted-repeat search answering from its twin and `blastx` = the same fabricated table (a shim on PATH: blastx itself is
    an external search in both builds); TRF changed none of the fixture's sequences, so the tandem step is the identity"""
    import itr_cases
    from hite_amd import util
    from oracle_ctx import OracleCtx

    itr_cases.check_low_copy_rescue(util, OracleCtx(), tmp_path, monkeypatch)


def test_column_vote_oracle_is_the_plain_count():
    """orc_column_vote (read off the oracle's col_base_map) == counting the six symbols column by column"""
    for case in load_golden("judge_tir")[:30]:
        if not case["clean"] or not case["clean"][0]:
            continue
        m = O.msa_array(case["clean"])
        exp = np.stack([(m == ord(ch)).sum(axis=0) for ch in "ACGTN-"], axis=1)
        assert np.array_equal(O.column_vote(m), exp)


def test_twin_far_pass_is_off_by_default_and_only_adds():
    """orc_find_copies_far is a MEASUREMENT AID of the twin (tools/far_copy_pass.py, profiles/r05_far_copy_pass.txt): the product has no
    far pass, so the twin's default must be the plain (10, 15) search; switched on, a candidate's table is replaced only by a larger one."""
    import synth_small

    g = synth_small.make(23, n_fam=12)
    base = O.find_copies(g["contigs"], g["cands"])
    try:
        O.find_copies_far(1 << 20)
        far = O.find_copies(g["contigs"], g["cands"])
    finally:
        O.find_copies_far(0)
    assert O.find_copies(g["contigs"], g["cands"]) == base
    assert all(len(b) >= len(a) for a, b in zip(base, far))
    assert all(a == b for a, b in zip(base, far) if len(a) == len(b))


def test_twin_padded_rows_and_the_aligned_interval_mode():
    """HITE_ROW_PAD through the twins (the GPU tests pin HIP == these): a padded row leaves the star alignment without its pads, and
    on records in the reference's coordinates (aligned intervals + clip words) the oracle chain with padded rows calls at least as
    many candidates TE as with the bare aligned windows (where every row is globally aligned to a centre it covers only in part)."""
    import oracle_pipeline as OP
    import synth_small

    rows = ["ACGTTGCAAGGCTTAACCGGTTAAGCAT" * 6, "." * 17 + ("ACGTTGCAAGGCTTAACCGGTTAAGCAT" * 6)[17:150] + "." * 18, "." * 168]
    m = O.star_msa(rows)
    assert m.shape[0] == 3 and not (m == ord(".")).any()
    assert [bytes(r[r != ord("-")]) for r in m] == [w.strip(".").encode() for w in rows]
    assert bytes(m[1][:17]) == b"-" * 17 and bytes(m[1][-18:]) == b"-" * 18 and bytes(m[2]) == b"-" * m.shape[1]
    # the other kind of pad byte: the centre's own bases in lower case match what they face -- same alignment, 35 less cost
    low = rows[0][:17].lower() + rows[1].strip(".") + rows[0][150:].lower()
    assert len(low) == len(rows[0]) and np.array_equal(O.star_msa([rows[0], low]), m[:2])
    ops_dot, d_dot = O.nw_pair(rows[0], rows[1])
    ops_low, d_low = O.nw_pair(rows[0], low)
    assert d_dot - d_low == 35 and np.array_equal(ops_dot, ops_low)
    g = synth_small.make(23, n_fam=24)
    tab = O.find_copies(g["contigs"], g["cands"], clips=True)         # the default: records in the reference's coordinates
    try:
        O.find_copies_config(False)
        whole = O.find_copies(g["contigs"], g["cands"], clips=True)
    finally:
        O.find_copies_config(None)
    assert all(cp[5] == 0 for t in whole for cp in t) and sum(cp[5] != 0 for t in tab for cp in t) > 20
    te = lambda table: sum(bool(OP.fine_stage_candidate("tir", c, t, g["contigs"], plant=1)[0]) for c, t in zip(g["cands"], table))  # noqa: E731
    n_pad, n_bare, n_whole = te(tab), te([[cp[:4] for cp in t] for t in tab]), te(whole)
    assert n_pad >= n_bare and n_pad >= 0.9 * n_whole, (n_pad, n_bare, n_whole)
