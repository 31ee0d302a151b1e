"""shared by the CPU test (host code + twins) and the GPU test (host code + HIP): runs hite_amd/util.py's mirrors of FMEA,
get_full_length_copies_from_blastn_v1, generate_full_length_out_v1 and multiple_alignment_blast_and_get_copies_v1 on the inputs
of tests/golden/chain_variants.json.gz (made by the reference, oracle/gen_golden.py gen_chain_variants) and compares"""
import os
import pickle

import numpy as np

import casegen
from conftest import load_golden


def _write_fasta(path, names, seqs):
    with open(path, "w") as f:
        for n, s in zip(names, seqs):
            f.write(">%s\n%s\n" % (n, s))


def check_all(util, ctx, tmp):
    g = load_golden("chain_variants")
    n_chains = n_copies = 0
    for ci, c in enumerate(g["fmea"]):
        p = os.path.join(tmp, "fmea_%d.out" % ci)
        names = c["names"]
        with open(p, "w") as f:
            f.writelines(casegen.hsp_to_blast6_lines([(names[q], names[s], a, b, cc, d) for (q, s, a, b, cc, d) in c["rows"]]))
        got = util.FMEA(p, c["gap"], ctx=ctx)
        got = [[k, [[t[0], int(t[1]), int(t[2]), t[3], int(t[4]), int(t[5])] for t in v]] for k, v in got.items()]
        assert got == c["out"], ("fmea", ci)                # (keys in the reference's insertion order)
        n_chains += sum(len(v) for _k, v in got)
    for ci, c in enumerate(g["full_length"]):
        qnames, snames, qlen, slen = c["qnames"], c["snames"], c["qlen"], c["slen"]
        lib, ref = os.path.join(tmp, "lib_%d.fa" % ci), os.path.join(tmp, "ref_%d.fa" % ci)
        keep = [q for q in range(len(qnames)) if q != c["drop"]]
        _write_fasta(lib, [qnames[q] for q in keep], ["ACGT" * (qlen[q] // 4) + "A" * (qlen[q] % 4) for q in keep])
        gen = np.random.default_rng(c["ref_seed"])
        _write_fasta(ref, snames, ["".join("ACGT"[i] for i in gen.integers(0, 4, L)) for L in slen])
        p = os.path.join(tmp, "fl_%d.out" % ci)
        lines = casegen.hsp_to_blast6_lines([(qnames[q], snames[s], a, b, cc, d) for (q, s, a, b, cc, d) in c["rows"]])
        if c["comment"]:
            lines.insert(0, "# a comment line\n")
        with open(p, "w") as f:
            f.writelines(lines)
        fl, ffl = util.get_full_length_copies_from_blastn_v1(lib, ref, p, tmp, 1, 20, c["thr"], c["search_struct"], "", ctx=ctx)
        assert [[k, [[kk, vv] for kk, vv in v.items()]] for k, v in fl.items()] == c["copies"], ("full_length", ci)
        assert [[k, [[kk, vv] for kk, vv in v.items()]] for k, v in ffl.items()] == c["flank_copies"], ("flank", ci)
        n_copies += sum(len(v) for v in fl.values())
        p2 = p + ".copy"
        with open(p2, "w") as f:
            f.writelines(lines)
        files = util.generate_full_length_out_v1(p2, lib, ref, os.path.join(tmp, "w_%d" % ci), "", c["thr"], c["category"], debug=0, ctx=ctx)
        assert [os.path.basename(x) for x in files] == c["out_files"] and not os.path.exists(p2)
        assert [sorted([list(t) for t in pickle.load(open(x, "rb"))]) for x in files] == c["out_sets"], ("out_sets", ci)
    for ci, c in enumerate(g["multi_blast"]):
        d = os.path.join(tmp, "mb_%d" % ci)
        refdir = os.path.join(d, "ref")
        os.makedirs(refdir, exist_ok=True)
        qpath = os.path.join(d, "q.fa")
        _write_fasta(qpath, c["qnames"], ["A" * L for L in c["qlen"]])
        calls = []

        def align_fn(chr_path, query_path, out_path, _c=c, _calls=calls):
            live = set(util.read_fasta(query_path)[0])
            _calls.append(os.path.basename(chr_path))
            with open(out_path, "w") as f:
                f.writelines(casegen.hsp_to_blast6_lines([tuple(r) for r in _c["tables"][os.path.basename(chr_path)] if r[0] in live]))

        real_listdir = os.listdir
        os.listdir = lambda pth, _f=c["files"], _r=refdir: list(_f) if pth == _r else real_listdir(pth)
        try:
            got = util.multiple_alignment_blast_and_get_copies_v1((qpath, refdir, os.path.join(d, "b.out")), align_fn=align_fn, ctx=ctx)
        finally:
            os.listdir = real_listdir
        got = [[k, [[x[0], int(x[1]), int(x[2]), int(x[3]), x[4]] for x in v]] for k, v in got.items()]
        assert got == c["out"] and calls == c["blast_calls"] and util.read_fasta(qpath)[0] == c["left_in_query_file"], ("multi_blast", ci)
    return n_chains, n_copies
