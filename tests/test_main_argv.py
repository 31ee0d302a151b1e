"""C1 plumbing from the real caller: tests/golden/main_argv.json holds the command lines /root/reference/main.py itself builds for
the stages of the dynamic-boundary path (main.py:479-482, 520-532, 545-562, 580-594, 612-628; recorded by oracle/gen_main_argv.py,
which runs main.py with os.system replaced by a recorder).  CPU: every recorded argv parses with the matching drop-in's own parser
and lands in the right fields.  GPU: split -> coarse -> TIR -> Helitron -> non-LTR on a miniature genome through exactly those argv."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SCRIPTS = os.path.join(ROOT, "hite_amd", "scripts")


def _runs():
    with open(os.path.join(HERE, "golden", "main_argv.json")) as f:
        return json.load(f)


def _parser(script):
    if SCRIPTS not in sys.path:
        sys.path.insert(0, SCRIPTS)
    spec = importlib.util.spec_from_file_location("dropin_" + script[:-3], os.path.join(SCRIPTS, script))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build_parser()


def test_reference_argv_parse_with_the_dropin_parsers():
    runs = _runs()
    assert [r["tag"] for r in runs] == ["defaults", "animal_tir_debug"]
    seen = set()
    for r in runs:
        plant = "0" if "--plant" in r["extra_args"] else "1"
        debug = 1 if "--debug" in r["extra_args"] else 0
        for c in r["commands"]:
            a = _parser(c[0]).parse_args(c[1:])         # an option the drop-in does not know would exit here
            seen.add(c[0])
            if c[0] == "split_genome_chunks.py":
                assert a.g == "{OUT}/genome.fa.clean" and a.tmp_output_dir == "{OUT}" and int(a.chrom_seg_length) == 1_000_000 and float(a.chunk_size) == 400
                continue
            i = a.ref_index
            assert i in ("0", "1") and a.r == "{OUT}/genome.fa.clean" and a.tmp_output_dir == "{OUT}" and a.work_dir == "{WORK}"
            assert a.recover == 0 and a.debug == debug and a.flanking_len == 50
            if c[0] == "coarse_boundary.py":
                assert a.g == "{OUT}/genome.cut%s.fa" % i and a.prev_TE == "{OUT}/prev_TE.fa" and a.hsp is None
                # 60 kb genome: get_fixed_extend_base_threshold's smallest class; threads = --thread 12 minus main.py's reserve of 4
                assert a.fixed_extend_base_threshold == 2000 and a.max_repeat_len == 30000 and a.thread == 8
            else:
                assert a.seqs == "{OUT}/longest_repeats_%s.flanked.fa" % i and a.t == 8 and a.min_TE_len == 80
                assert a.split_ref_dir == "{OUT}/ref_chr" and a.prev_TE == "{OUT}/prev_TE.fa"
                if c[0] == "judge_TIR_transposons.py":
                    assert str(a.plant) == plant and a.all_low_copy_tir == "{OUT}/tir_low_copy.fa"
                elif c[0] == "judge_Helitron_transposons.py":
                    assert a.all_low_copy_helitron == "{OUT}/helitron_low_copy.fa" and a.candidates is None
                else:
                    assert a.all_low_copy_non_ltr == "{OUT}/non_ltr_low_copy.fa" and a.is_denovo_nonltr == 1 and a.candidates is None
    assert seen == {"split_genome_chunks.py", "coarse_boundary.py", "judge_TIR_transposons.py", "judge_Helitron_transposons.py", "judge_Non_LTR_transposons.py"}


@pytest.mark.gpu
def test_stage_chain_through_the_reference_argv(tmp_path):
    """the stages of step 3 run one after the other with the argv main.py builds (chunk 0 of the `defaults` run; {OUT} / {WORK} =
    a scratch directory, genome.fa.clean = a miniature genome with planted TIR families): every stage ends with its output file
    in place -- the success test of the reference (Util.py:2831) -- the TIR library holds planted families, prev_TE.fa grows"""
    import synth_small
    from hite_amd import util

    g = synth_small.make(29, n_fam=12, n_chr=2, chr_len=180_000)
    out, work = tmp_path / "out", tmp_path / "work"
    out.mkdir(); work.mkdir()
    (out / "genome.fa.clean").write_text("".join(">chr%d\n%s\n" % (i + 1, s) for i, s in enumerate(g["contigs"])))
    # what HelitronScanner / EAHelitron (external, out of scope) would leave for the Helitron stage: a candidate consensus file
    (out / "candidate_helitron_0.cons.fa").write_text("".join(">h%d\n%s\n" % (i, s) for i, s in enumerate(g["cands"][:6])))
    cmds = [c for c in _runs()[0]["commands"] if "--ref_index" not in c or c[c.index("--ref_index") + 1] == "0"]
    assert [c[0] for c in cmds] == ["split_genome_chunks.py", "coarse_boundary.py", "judge_TIR_transposons.py", "judge_Helitron_transposons.py",
                                    "judge_Non_LTR_transposons.py"]
    expect = {"split_genome_chunks.py": "genome.cut0.fa", "coarse_boundary.py": "longest_repeats_0.flanked.fa", "judge_TIR_transposons.py": "confident_tir_0.fa",
              "judge_Helitron_transposons.py": "confident_helitron_0.fa", "judge_Non_LTR_transposons.py": "confident_non_ltr_0.fa"}
    for c in cmds:
        argv = [a.replace("{OUT}", str(out)).replace("{WORK}", str(work)) for a in c[1:]]
        rc = subprocess.run([sys.executable, os.path.join(SCRIPTS, c[0])] + argv, capture_output=True, text=True)
        assert rc.returncode == 0, (c[0], rc.stderr[-2000:])
        assert (out / expect[c[0]]).exists(), c[0]
    names, _ = util.read_fasta(str(out / "longest_repeats_0.flanked.fa"))
    assert len(names) >= 10
    tn, tc = util.read_fasta(str(out / "confident_tir_0.fa"))
    assert len(tn) >= 2 and all(n.startswith("genome-TIR_0_") for n in tn)
    genome_text = "".join(g["contigs"])
    planted = 0
    for n in tn:        # a TIR library entry is a consensus: close to one planted element (within 3 %)
        sq = tc[n]
        planted += any(abs(len(sq) - (b - a + 1)) <= 0.1 * len(sq) for fam in g["truth"] for (c_, a, b, _m) in fam[:1])
    assert planted >= 2
    pn, _pc = util.read_fasta(str(out / "prev_TE.fa"))
    assert len(pn) >= len(tn)
