"""TEST INFRASTRUCTURE: checks of the terminal-inverted-repeat stage (itrsearch, Util.py:216) against tests/golden/itr_search.json.gz
(the tool's own output + the reference functions around it), written once and run twice: with the CPU twin behind the product's
host mirrors (tests/oracle_ctx.py, CPU suite) and with the HIP library (GPU suite)."""
import numpy as np

from conftest import load_golden


def check_tool_records(itr_search):
    """itr_search(seqs, end_len) -> int32 [n, 8] must report what the tool reported: presence in <input>.itr and "Length itr=";
    no record ever takes the path on which the tool's stale state would matter (flags 0)"""
    g = load_golden("itr_search")
    n = 0
    for part in ("pairs", "whole"):
        seqs, res = g[part]["seqs"], g[part]["res"]
        out = itr_search(seqs, 0)
        assert out.shape == (len(seqs), 8)
        assert [[int(r[5]), int(r[6]) if r[5] else -1] for r in out] == res, part
        assert int(out[:, 7].max()) == 0
        n += int(out[:, 5].sum())
        if part == "pairs":
            # first 40 + last 40 composed by the stage == the composed record handed over whole
            short = [s for s in seqs if len(s) > 80][:200] + seqs[:200]
            a = itr_search(short, 40)
            b = itr_search([s[:40] + s[-40:] for s in short], 0)
            assert np.array_equal(a, b)
    assert n > 1500
    return n


def _parse_final(name):
    q, rest = name.split("-tir_")
    tl, tsd = rest.split("-tsd_")
    return q, int(tl), tsd


def check_batches(util, ctx):
    """search_confident_tir_batch_v1 (Util.py:6533-6628) as the reference ran it WITH the tool: the same queries survive, and the
    reference's pick for each query is one of the variants the product keeps at the smallest distance (the reference's order among
    equal distances is PYTHONHASHSEED's, the product's is canonical), with the same TIR length, TSD and sequence"""
    g = load_golden("itr_search")
    n_q = n_drop = 0
    for b in g["batches"]:
        contigs = dict(zip(b["names"], b["seqs"]))
        kept, tir_len = util.tir_variants_with_structure(b["names"], contigs, b["flank"], b["plant"], ctx=ctx)
        groups = {}
        for v, s in kept.items():
            groups.setdefault(v.split("-C_")[0], []).append((int(v.split("-distance_")[1]), v, s))
        ref = {}
        for name, seq in b["out"]:
            q, tl, tsd = _parse_final(name)
            ref[q] = (tl, tsd, seq)
        assert set(ref) == {q for q, vs in groups.items() if any(len(s) < 30000 for _d, _v, s in vs)}
        for q, (tl, tsd, seq) in ref.items():
            dmin = min(d for d, _v, _s in groups[q])
            ties = {(tir_len.get(v, 0), v.split("-tsd_")[1].split("-")[0], s) for d, v, s in groups[q] if d == dmin}
            assert (tl, tsd, seq) in ties, (q, tl, tsd, sorted(t[:2] for t in ties))
        got = util.search_confident_tir_batch_v1(b["names"], contigs, b["flank"], b["plant"], ctx=ctx)
        assert {_parse_final(k)[0] for k in got} == set(ref)
        # the filter drops variants: every k-mer TSD variant of the batch minus those kept
        n_all = sum(len(r) for r in ctx.tsd_kmer([contigs[x] for x in b["names"] if "NNNNNNNNNN" not in contigs[x]], flank=b["flank"], plant=b["plant"]))
        n_drop += n_all - len(kept)
        n_q += len(ref)
    assert n_q > 80 and n_drop > 1000
    return n_q, n_drop


def check_rescue(util, ctx):
    """remove_no_tirs (Util.py:13897-13920) as the reference ran it with the tool: both output files, in order"""
    g = load_golden("itr_search")
    n = 0
    for r in g["rescue"]:
        contigs = dict(zip(r["names"], r["seqs"]))
        with_tir, no_tir = util.remove_no_tirs(contigs, r["plant"], ctx=ctx)
        assert [[k, v] for k, v in with_tir.items()] == r["with_tir"]
        assert [[k, v] for k, v in no_tir.items()] == r["no_tir"]
        n += len(with_tir)
    assert n > 100
    return n


def check_low_copy_rescue(util, ctx, tmp_path, monkeypatch):
    """bucket_results + rescue_low_copy against tests/golden/low_copy_rescue.json.gz (the reference's run of Util.py:8196-8287 with TRF,
    itrsearch and get_domain_info over a fabricated blastx table); `blastx` is a shim on PATH printing the same fabricated table"""
    import json
    import os
    import stat

    bindir = tmp_path / "bin"
    bindir.mkdir()
    (bindir / "makeblastdb").write_text("#!/bin/sh\nexit 0\n")
    (bindir / "blastx").write_text("#!/usr/bin/env python3\nimport json, os, sys\nq = sys.argv[sys.argv.index('-query') + 1]\n"
                                   "tab = json.load(open(os.environ['HITE_FAKE_BLASTX']))\n"
                                   "for line in open(q):\n    if line.startswith('>'):\n        for r in tab.get(line[1:].strip(), []):\n            sys.stdout.write(r)\n")
    for x in ("makeblastdb", "blastx"):
        os.chmod(str(bindir / x), os.stat(str(bindir / x)).st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ["PATH"])
    n_dom = n_itr = 0
    for ci, case in enumerate(load_golden("low_copy_rescue")):
        assert case["trf_changed"] == []
        te = case["te_type"]
        lib_dir = tmp_path / ("lib%d" % ci)
        lib_dir.mkdir()
        fn = {"tir": "TIRPeps.lib", "helitron": "HelitronPeps.lib", "non_ltr": "non_LTR.lib"}[te]
        (lib_dir / fn).write_text("".join(">%s\n%s\n" % (k, v) for k, v in case["library"]))
        fake = tmp_path / ("blastx%d.json" % ci)
        fake.write_text(json.dumps(case["blastx"]))
        monkeypatch.setenv("HITE_FAKE_BLASTX", str(fake))
        true_tes, low = util.bucket_results(te, [(r[1], r[2], r[3], r[4]) for r in case["table"]])
        work = str(tmp_path / ("w%d" % ci))
        rescued, still = util.rescue_low_copy(te, low, case["plant"], work, tandem_masker=lambda names, contigs: dict(contigs), ctx=ctx,
                                              library_dir=str(lib_dir), threads=1)
        true_tes.update(rescued)
        assert [[k, v] for k, v in true_tes.items()] == case["real"], (ci, te)
        assert "".join(">%s\n%s\n" % (k, v) for k, v in still.items()) == case["low_text"]
        tab = [x for x in os.listdir(work) if x.endswith("_domain") and os.path.isfile(os.path.join(work, x))]
        assert len(tab) == 1 and open(os.path.join(work, tab[0])).read() == case["domain_table"]
        n_dom += case["domain_table"].count("\n") - 2
        n_itr += sum(1 for k in rescued if te == "tir")
    assert n_dom > 80 and n_itr > 10
