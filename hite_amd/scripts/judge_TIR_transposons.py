#!/usr/bin/env python
"""Drop-in for HiTE's module/judge_TIR_transposons.py (stage 3.2): same argv, same files
(/root/reference/module/judge_TIR_transposons.py:114-143, 63-111, 16-61).

  --seqs longest_repeats_{i}.flanked.fa  ->  <tmp_output_dir>/confident_tir_{i}.fa   (+ tir_low_copy.fa, prev_TE append)

What runs on the GPU: the k-mer TSD seed search (search_confident_tir_v4), copy finding, flank-window
gather, star alignment, sparse-column removal and judge_boundary_v5, three refinement iterations.
The terminal-inverted-repeat filter the reference runs as `tools/itrsearch -i 0.7 -l 7` (Util.py:216) is an in-tree GPU stage
(hite_itr_search, pinned to the tool's own output): it is never skipped.  `cd-hit-est` (judge_TIR_transposons.py:87) is called
when it is installed; otherwise the in-tree clustering stand-in runs (_stage.run_cd_hit) -- no step passes candidates through."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import _stage  # noqa: E402
from _stage import util  # noqa: E402


def tsd_variants(flanked_path, flanking_len, plant, work_dir):
    """multi_process_tsd_v1 (Util.py:6630) without the per-file process pool: one batched call"""
    names, contigs = util.read_fasta(flanked_path)
    return util.search_confident_tir_batch_v1(names, contigs, flanking_len, plant, work_dir)


def build_parser():
    p = argparse.ArgumentParser(description="run HiTE TIR module on the MI355X path")
    p.add_argument("--seqs"); p.add_argument("-t", type=int, default=1); p.add_argument("--tmp_output_dir")
    p.add_argument("--tandem_region_cutoff", default="0.5"); p.add_argument("--ref_index", default="0")
    p.add_argument("--plant", type=int, default=1); p.add_argument("--flanking_len", type=int, default=50)
    p.add_argument("--recover", type=int, default=0); p.add_argument("--debug", type=int, default=0)
    p.add_argument("-r"); p.add_argument("--split_ref_dir", default=None); p.add_argument("--prev_TE", default=None)
    p.add_argument("--all_low_copy_tir", default=None); p.add_argument("--min_TE_len", type=int, default=80)
    p.add_argument("-w", "--work_dir", default="/tmp")
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    out_dir = os.path.abspath(a.tmp_output_dir or os.getcwd())
    os.makedirs(out_dir, exist_ok=True)
    ref_index, flank = a.ref_index, 50  # the reference pins flanking_len to 50 here (judge_TIR_transposons.py:21)
    final = os.path.join(out_dir, "confident_tir_%s.fa" % ref_index)
    if a.recover and os.path.exists(final) and util.read_fasta(final)[0]:
        return 0
    low = a.all_low_copy_tir or os.path.join(out_dir, "tir_low_copy.fa")
    util.set_reference(a.r)
    tsd_path = os.path.join(out_dir, "tir_tsd_%s.fa" % ref_index)
    util.store_fasta(tsd_variants(a.seqs, flank, a.plant, out_dir), tsd_path)
    cons_path = tsd_path + ".cons"
    _stage.run_cd_hit(tsd_path, cons_path, a.t)
    last = _stage.refine("tir", cons_path, out_dir, "confident_tir", ref_index, a.r, a.split_ref_dir, a.t, a.plant, a.debug, low, 3)
    _stage.finish(last, final, "TIR", ref_index, a.r, a.min_TE_len, a.prev_TE)
    return 0


if __name__ == "__main__":
    sys.exit(main())
