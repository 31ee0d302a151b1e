#!/usr/bin/env python
"""Drop-in for HiTE's module/pan_remove_redundancy.py (the library merge of panHiTE, BASELINE.json config 5): same argv
(/root/reference/module/pan_remove_redundancy.py:50-80), same output <output_dir>/panTE.fa.

  --merge_te_file <all genomes' TE libraries concatenated>  ->  LTR internal sequences and the rest are de-duplicated
  separately (coverage 0.8 / 0.95, deredundant_for_LTR_v5) and concatenated.

GPU: library-vs-library seeding, fragment chaining, star alignment of every cluster, majority consensus.  External tools of the
reference's path: blastn and mafft are replaced by the build's own stages, Ninja's sub-clustering is not made, cd-hit-est runs
when installed."""
import argparse
import os
import shutil
import sys
import uuid

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from hite_amd import util  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="panHiTE remove redundancy.")
    p.add_argument("--merge_te_file"); p.add_argument("--threads", type=int, default=1)
    p.add_argument("--output_dir", nargs="?", default=os.getcwd()); p.add_argument("-w", "--work_dir", nargs="?", default="/tmp")
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    out_dir = os.path.abspath(a.output_dir)
    os.makedirs(out_dir, exist_ok=True)
    tmp = os.path.join(os.path.abspath(a.work_dir), "pan_remove_redundancy_" + str(uuid.uuid4()))
    os.makedirs(tmp)
    try:
        other, internal = util.split_internal_out(a.merge_te_file, tmp)
        util.deredundant_for_LTR_v5(other, tmp, a.threads, "terminal", 0.95, 0)
        util.deredundant_for_LTR_v5(internal, tmp, a.threads, "internal", 0.8, 0)
        pan = os.path.join(out_dir, "panTE.fa")
        with open(pan + ".tmp", "w") as f:
            for part in (other + ".cons", internal + ".cons"):
                with open(part) as g:
                    f.write(g.read())
        os.replace(pan + ".tmp", pan)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
