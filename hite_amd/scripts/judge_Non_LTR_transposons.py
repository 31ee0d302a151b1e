#!/usr/bin/env python
"""Drop-in for HiTE's module/judge_Non_LTR_transposons.py (same argv and output files,
/root/reference/module/judge_Non_LTR_transposons.py:16-144): <tmp_output_dir>/confident_non_ltr_{i}.fa.

Candidates: as in the reference they come from the flanked repeats (`--seqs`): polyA/T or tandem tail + TSD structures
(get_candidate_non_LTR -> search_polyA_TSD, on the GPU), SINE class first, then LINE.  The rescue of LINEs without a TSD by
protein-domain search (blastx, external) is not part of this build.  `--candidates <fa>` (extension) passes a ready
candidate file instead.
GPU: one pass of flank_region_align_v5 with judge_boundary_v9 (homology boundaries, polyA / tandem tail within 10 columns
of the 3' boundary, 8-20 bp TSD with <= 1 edit upstream of the 5' boundary)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import _stage  # noqa: E402
from _stage import util  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="run HiTE non-LTR module on the MI355X path")
    p.add_argument("--seqs"); p.add_argument("-t", type=int, default=1); p.add_argument("--subset_script_path", default=None)
    p.add_argument("--tmp_output_dir"); p.add_argument("--library_dir", default=None); p.add_argument("--recover", type=int, default=0)
    p.add_argument("--plant", type=int, default=1); p.add_argument("--debug", type=int, default=0)
    p.add_argument("--flanking_len", type=int, default=50); p.add_argument("--ref_index", default="0")
    p.add_argument("--is_denovo_nonltr", type=int, default=1); p.add_argument("-r"); p.add_argument("--split_ref_dir", default=None)
    p.add_argument("--prev_TE", default=None); p.add_argument("--all_low_copy_non_ltr", default=None)
    p.add_argument("--min_TE_len", type=int, default=80); p.add_argument("-w", "--work_dir", default="/tmp")
    p.add_argument("--candidates", default=None, help="candidate non-LTR FASTA -- extension of this build")
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    out_dir = os.path.abspath(a.tmp_output_dir or os.getcwd())
    os.makedirs(out_dir, exist_ok=True)
    final = os.path.join(out_dir, "confident_non_ltr_%s.fa" % a.ref_index)
    if a.recover and os.path.exists(final) and util.read_fasta(final)[0]:
        return 0
    if not a.is_denovo_nonltr:
        util.store_fasta({}, final)
        return 0
    cand = a.candidates
    if cand is None:
        cand = os.path.join(out_dir, "candidate_non_ltr_%s.fa" % a.ref_index)
        sine, line = util.get_candidate_non_LTR(a.seqs, a.flanking_len)
        merged = dict(sine)
        merged.update(line)          # `cat SINE > candidates; cat LINE >> candidates` (judge_Non_LTR_transposons.py:37-38)
        util.store_fasta(merged, cand)
    low = a.all_low_copy_non_ltr or os.path.join(out_dir, "non_ltr_low_copy.fa")
    util.set_reference(a.r)
    cons_in = cand + ".cons"
    _stage.run_cd_hit(cand, cons_in, a.t)
    last = _stage.refine("non_ltr", cons_in, out_dir, "confident_non_ltr", a.ref_index, a.r, a.split_ref_dir, a.t, a.plant, a.debug, low, 1)
    cons = last + ".cons"
    _stage.run_cd_hit(last, cons, a.t)
    _stage.finish(cons, final, "Non_LTR", a.ref_index, a.r, a.min_TE_len, a.prev_TE)
    return 0


if __name__ == "__main__":
    sys.exit(main())
