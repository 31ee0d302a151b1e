#!/usr/bin/env python
"""Drop-in for HiTE's module/coarse_boundary.py (stage 3.1): same argv and output files
(/root/reference/module/coarse_boundary.py:36-60):

  <tmp_output_dir>/longest_repeats_{i}.fa, longest_repeats_{i}.flanked.fa

GPU: the all-vs-all search of the chunk's 1 Mbp segments against each other (the build's own stage where the
reference runs `blastn`, Util.py:4068 -- hite_seed_allvsall), FMEA chaining + de-duplication
(get_longest_repeats_v4) and the flank gather.  `--hsp <blast6 file(s)>` (an extension of this build) takes
the HSP tables from real blastn runs instead (what sequence2sequenceBlastn writes, one file per query FASTA as
process_blast_alignments concatenates them, Util.py:4750-4769).  As in the reference the chunk is first masked: tandem
repeats by `trf` (Util.py:2855; external tool, used when installed, a warning otherwise), full-length copies of the TEs of
--prev_TE by mask_genome_intactTE (Util.py:6389), so that a later chunk does not re-discover what an earlier one found."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from hite_amd import util  # noqa: E402


def fmea_file(ctx, path, skip_gap, max_len):
    """one call of get_longest_repeats_v4 (Util.py:4122): blast6 text -> ['chr:start-end', ...] in dict order"""
    segs, chroms, seg_chrom, seg_off = {}, {}, [], []
    cols = [[], [], [], [], [], []]

    def seg(name):
        if name not in segs:
            c, off = name.split("$")
            chroms.setdefault(c, len(chroms))
            segs[name] = len(segs)
            seg_chrom.append(chroms[c])
            seg_off.append(int(off))
        return segs[name]

    with open(path) as f:
        for line in f:
            p = line.split("\t")
            if len(p) < 10:
                continue
            for k, v in zip(range(6), (seg(p[0]), seg(p[1]), int(p[6]), int(p[7]), int(p[8]), int(p[9]))):
                cols[k].append(v)
    if not cols[0]:
        return []
    oc, os_, oe = ctx.fmea_chain(cols[0], cols[1], cols[2], cols[3], cols[4], cols[5], seg_chrom, seg_off, skip_gap, max_len)
    inv = {v: k for k, v in chroms.items()}
    return ["%s:%d-%d" % (inv[c], s, e) for c, s, e in zip(oc, os_, oe)]


def build_parser():
    p = argparse.ArgumentParser(description="run HiTE De novo TE searching on the MI355X path")
    p.add_argument("-g"); p.add_argument("--prev_TE", default=None)
    p.add_argument("--fixed_extend_base_threshold", type=int, default=4000); p.add_argument("--max_repeat_len", type=int, default=30000)
    p.add_argument("--thread", type=int, default=1); p.add_argument("--flanking_len", type=int, default=50)
    p.add_argument("--tandem_region_cutoff", default="0.5"); p.add_argument("--ref_index", default="0")
    p.add_argument("-r"); p.add_argument("--tmp_output_dir"); p.add_argument("--recover", type=int, default=0)
    p.add_argument("--debug", type=int, default=0); p.add_argument("-w", "--work_dir", default="/tmp")
    p.add_argument("--hsp", nargs="+", default=None, help="blast6 HSP tables (one per query file) -- extension of this build")
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    out_dir = os.path.abspath(a.tmp_output_dir or os.getcwd())
    os.makedirs(out_dir, exist_ok=True)
    lr = os.path.join(out_dir, "longest_repeats_%s.fa" % a.ref_index)
    fl = os.path.join(out_dir, "longest_repeats_%s.flanked.fa" % a.ref_index)
    if a.recover and os.path.exists(lr) and os.path.exists(fl) and util.read_fasta(fl)[0]:
        return 0
    if not a.hsp:
        tmp = lr + ".tmp"
        util.determine_repeat_boundary_v5(a.g, tmp, a.prev_TE, a.fixed_extend_base_threshold, a.max_repeat_len, out_dir, a.thread, a.ref_index,
                                          a.r, a.debug)
        os.replace(tmp, lr)
        util.flanking_seq(lr, fl + ".tmp", a.r, a.flanking_len)
        os.replace(fl + ".tmp", fl)
        return 0
    ctx = util.set_reference(a.r)
    _names, ref = util.read_fasta(a.r)
    final = {}
    for path in a.hsp:  # generate_final_result (Util.py:4783): union by name, first occurrence keeps its position
        for name in fmea_file(ctx, path, a.fixed_extend_base_threshold, a.max_repeat_len):
            if name not in final:
                c, pos = name.split(":")
                s, e = map(int, pos.split("-"))
                final[name] = ref[c][s:e]
    tmp = lr + ".tmp"
    util.store_fasta(final, tmp)
    os.replace(tmp, lr)
    util.flanking_seq(lr, fl + ".tmp", a.r, a.flanking_len)
    os.replace(fl + ".tmp", fl)
    return 0


if __name__ == "__main__":
    sys.exit(main())
