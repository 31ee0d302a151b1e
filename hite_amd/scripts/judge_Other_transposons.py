#!/usr/bin/env python
"""Drop-in for HiTE's module/judge_Other_transposons.py (same argv and output file,
/root/reference/module/judge_Other_transposons.py:34-86): <tmp_output_dir>/confident_other.fa.

Homology search of the curated non-LTR library against the genome: the reference runs blastn (`multi_process_align_and_get_copies`,
query_coverage 0.95) and keeps, per library entry, its longest genomic copy (>= 100 bp, inside the contig), named
`chr:start-end#Class`, then drops entries shorter than min_TE_len and renames to Homology_Non_LTR_<i>#Class (rename_fasta,
Util.py:7500).  Here the copies come from the build's copy finder on the resident packed genome (the stage that stands where
the reference aligns library / candidates to the genome); selection, naming and the file contract are the reference's.
The library file: `--lib <fa>` (extension of this build), else $HITE_LIBRARY_DIR/non_LTR.lib, else <HiTE>/library/non_LTR.lib
next to this package (the data file HiTE ships; it is not part of this build)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import _stage  # noqa: E402
from _stage import util  # noqa: E402


def longest_copies(all_copies, ref_contigs):
    """judge_Other_transposons.py:50-80 -- {library name: copies} -> {chr:s-e#Class: sequence} (longest valid copy of each)"""
    out = {}
    for query_name, copies in all_copies.items():
        parts = str(query_name).split("#")
        class_name = parts[1] if len(parts) == 2 else None
        max_len, max_name, max_seq = 0, None, None
        for copy in copies:
            ref_name, s, e = copy[0], int(copy[1]), int(copy[2])
            if s - 1 < 0 or e > len(ref_contigs[ref_name]):
                continue
            seq = ref_contigs[ref_name][s - 1:e]
            if len(seq) < 100:
                continue
            new_name = "%s:%d-%d" % (ref_name, s, e) + ("#" + class_name if class_name is not None else "")
            if len(seq) > max_len:
                max_name, max_seq, max_len = new_name, seq, len(seq)
        if max_name is not None:
            out[max_name] = max_seq
    return out


def main():
    p = argparse.ArgumentParser(description="run HiTE homology Non_LTR module on the MI355X path")
    p.add_argument("-t", type=int, default=1); p.add_argument("--tmp_output_dir"); p.add_argument("--recover", type=int, default=0)
    p.add_argument("-r"); p.add_argument("--min_TE_len", type=int, default=80); p.add_argument("-w", "--work_dir", default="/tmp")
    p.add_argument("--lib", default=None, help="non-LTR library FASTA -- extension of this build")
    a = p.parse_args()
    out_dir = os.path.abspath(a.tmp_output_dir or os.getcwd())
    os.makedirs(out_dir, exist_ok=True)
    final = os.path.join(out_dir, "confident_other.fa")
    if a.recover and os.path.exists(final) and os.path.getsize(final) > 0:
        return 0
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    lib = a.lib or os.path.join(os.environ.get("HITE_LIBRARY_DIR", os.path.join(here, "library")), "non_LTR.lib")
    if not os.path.exists(lib):
        sys.stderr.write("judge_Other_transposons (MI355X path): library %s not found (pass --lib or set HITE_LIBRARY_DIR)\n" % lib)
        return 2
    reference = os.path.realpath(a.r)
    util.set_reference(reference)
    all_copies = util.get_full_length_copies_minimap2(lib, reference)
    _names, ref_contigs = util.read_fasta(reference)
    best = longest_copies(all_copies, ref_contigs)
    kept = {n: s for n, s in best.items() if len(s) >= a.min_TE_len}
    renamed = {}
    for i, (n, s) in enumerate(kept.items()):                        # rename_fasta(..., 'Homology_Non_LTR')
        parts = n.split("#")
        renamed["Homology_Non_LTR_%d" % i + ("#" + parts[-1] if len(parts) >= 2 else "")] = s
    tmp = final + ".tmp"
    util.store_fasta(renamed, tmp)
    os.replace(tmp, final)
    return 0


if __name__ == "__main__":
    sys.exit(main())
