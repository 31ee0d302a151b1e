#!/usr/bin/env python
"""Drop-in for HiTE's module/judge_Helitron_transposons.py (same argv and output files,
/root/reference/module/judge_Helitron_transposons.py:20-144): <tmp_output_dir>/confident_helitron_{i}.fa.

Candidates: the reference obtains them from HelitronScanner / EAHelitron on the flanked repeats (external Java / Perl
tools, :33-76).  Here `--candidates <fa>` (extension of this build) passes that candidate consensus file; without it
the script looks for <tmp_output_dir>/candidate_helitron_{i}.cons.fa, the file the reference's own front half writes.
GPU: three refinement iterations of flank_region_align_v5 with judge_boundary_v6 (copy finding, window gather, star
alignment, sparse columns, mode-anchored boundary search, ATC / CTRRT motif rules)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import _stage  # noqa: E402
from _stage import util  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="run HiTE Helitron module on the MI355X path")
    p.add_argument("--seqs"); p.add_argument("-t", type=int, default=1); p.add_argument("--tmp_output_dir")
    p.add_argument("--HSDIR", default=None); p.add_argument("--HSJAR", default=None); p.add_argument("--sh_dir", default=None)
    p.add_argument("--member_script_path", default=None); p.add_argument("--subset_script_path", default=None)
    p.add_argument("--flanking_len", type=int, default=50); p.add_argument("--ref_index", default="0")
    p.add_argument("--recover", type=int, default=0); p.add_argument("--debug", type=int, default=0)
    p.add_argument("-r"); p.add_argument("--split_ref_dir", default=None); p.add_argument("--prev_TE", default=None)
    p.add_argument("--all_low_copy_helitron", default=None); p.add_argument("--min_TE_len", type=int, default=80)
    p.add_argument("-w", "--work_dir", default="/tmp")
    p.add_argument("--candidates", default=None, help="candidate Helitron consensus FASTA -- extension of this build")
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    out_dir = os.path.abspath(a.tmp_output_dir or os.getcwd())
    os.makedirs(out_dir, exist_ok=True)
    final = os.path.join(out_dir, "confident_helitron_%s.fa" % a.ref_index)
    if a.recover and os.path.exists(final) and util.read_fasta(final)[0]:
        return 0
    cand = a.candidates or os.path.join(out_dir, "candidate_helitron_%s.cons.fa" % a.ref_index)
    if not os.path.exists(cand):
        sys.stderr.write("judge_Helitron_transposons (MI355X path): no candidate file (%s); HelitronScanner / EAHelitron are external\n" % cand)
        return 2
    low = a.all_low_copy_helitron or os.path.join(out_dir, "helitron_low_copy.fa")
    util.set_reference(a.r)
    last = _stage.refine("helitron", cand, out_dir, "confident_helitron", a.ref_index, a.r, a.split_ref_dir, a.t, 1, a.debug, low, 3)
    cons = last.replace(".fa", ".cons.fa")
    _stage.run_cd_hit(last, cons, a.t)
    names, contigs = util.read_fasta(cons)
    util.store_fasta({n.split("#")[0]: contigs[n] for n in names}, cons)   # headers lose their '#...' part (:127)
    _stage.finish(cons, final, "Helitron", a.ref_index, a.r, a.min_TE_len, a.prev_TE)
    return 0


if __name__ == "__main__":
    sys.exit(main())
