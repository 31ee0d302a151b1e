#!/usr/bin/env python
"""Drop-in for HiTE's module/split_genome_chunks.py (same argv, same files: /root/reference/module/split_genome_chunks.py:11-88):

  <tmp_output_dir>/genome.cut{i}.fa   'chr$offset' segments of --chrom_seg_length bases, a new chunk every --chunk_size MiB
                                       of FASTA text;  <tmp_output_dir>/ref_chr/ref_block_{i}.fa

Pure file formatting (no GPU work): it exists so that the chunk files the GPU stages read are byte-identical to the
reference's (tests/golden/split_chunks.json.gz)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from hite_amd import util  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="run HiTE split genome chunks...")
    p.add_argument("-g"); p.add_argument("--tmp_output_dir", default=None)
    p.add_argument("--chrom_seg_length"); p.add_argument("--chunk_size")
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    util.split_genome_chunks(a.g, a.tmp_output_dir or os.getcwd(), int(a.chrom_seg_length), float(a.chunk_size))
    return 0


if __name__ == "__main__":
    sys.exit(main())
