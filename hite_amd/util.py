"""Host-side mirror of the reference's function interface for the dynamic-boundary path
(/root/reference/module/Util.py).  Same names, argument meaning and file contracts; the
arithmetic runs in libhite_gpu.so through `hite_amd._lib` (no CPU fallback).

Third-party tools the reference shells out to: `minimap2` copy finding is replaced by the build's own minimizer-based
finder (get_full_length_copies_minimap2; `all_copies=` / `copy_finder=` of flank_region_align_v5 override it); `itrsearch`
(run_itrsearch, Util.py:216) is an in-tree GPU stage pinned to the tool's own output; `trf` and `cd-hit-est` are called where the
reference calls them when they are installed (the build's own masker / clustering otherwise); the low-copy recall of TIR
candidates by structure (Util.py:8196-8213: short-TIR signatures + terminal inverted repeats) is reproduced, and so is the
decision of the recall by protein domains (:8215-8276) on a blastx domain table; blastx itself stays external.
"""
import itertools
import os
import re
import shutil
import subprocess
import sys

import numpy as np

from ._lib import Context

_CTX = None
_PACKED = {"path": None, "names": None, "lens": None}


def get_ctx(device=0):
    global _CTX
    if _CTX is None or not getattr(_CTX, "h", None):      # none yet, or closed by its previous owner
        _CTX = Context(device)
    return _CTX


# ---- FASTA (Util.py:1650-1670, 1983-1988) ---------------------------------------------------------
def read_fasta(fasta_path):
    """-> (names in file order, {name: sequence}): the name is the header up to the first blank or tab, the sequence upper-case;
    a record without sequence characters is skipped, text before the first header ignored, a missing file gives empty results"""
    names, seqs = [], {}
    if not os.path.exists(fasta_path):
        return names, seqs
    current = ""
    with open(fasta_path, "r") as handle:
        # runs of header lines / of sequence lines: of several headers in a row only the last can own sequence
        for is_header, run in itertools.groupby(handle, key=lambda ln: ln.startswith(">")):
            if is_header:
                for ln in run:
                    current = ln.strip()[1:].split(" ")[0].split("\t")[0]
                continue
            sequence = "".join(ln.strip().upper() for ln in run)
            if current != "" and sequence != "":
                seqs[current] = sequence
                names.append(current)
            # (a second run of sequence lines cannot follow without a header in between: groupby merges neighbours)
    return names, seqs


def store_fasta(contigs, file_path):
    with open(file_path, "w") as f:
        for name, seq in contigs.items():
            f.write(">" + name + "\n" + seq + "\n")


def rename_fasta(input, output, header="N"):
    """Util.py:7500 -- names become <header>_<i>, a '#class' suffix (text after the last '#') is kept"""
    names, contigs = read_fasta(input)
    with open(output, "w") as out:
        for i, old in enumerate(names):
            _head, sep, cls = str(old).rpartition("#")
            out.write(">%s_%d%s\n%s\n" % (header, i, sep + cls if sep else "", contigs[old]))


def rename_reference(input, output, chr_name_map):
    """Util.py:7517 -- contigs renamed chr_<i>; the map file holds 'new\told' lines"""
    names, contigs = read_fasta(input)
    chr_name_dict = {}
    with open(output, "w") as f_save:
        for ref_index, name in enumerate(names):
            new_name = "chr_" + str(ref_index)
            f_save.write(">" + new_name + "\n" + contigs[name] + "\n")
            chr_name_dict[new_name] = name
    with open(chr_name_map, "w") as f_save:
        for new_name, old in chr_name_dict.items():
            f_save.write(new_name + "\t" + old + "\n")


def lib_add_prefix(HiTE_lib, prefix):
    """Util.py:11559 -- every name gets '<prefix>-' in front, in place"""
    lib_names, lib_contigs = read_fasta(HiTE_lib)
    store_fasta({prefix + "-" + name: lib_contigs[name] for name in lib_names}, HiTE_lib)
    return HiTE_lib


def convertToUpperCase_v1(reference):
    """Util.py:1521 -- the genome FASTA is rewritten in place: names cut at the first blank, sequences upper-case on one line"""
    names, contigs = [], {}
    with open(reference, "r") as f_r:
        name, seq = "", []
        for line in f_r:
            if line.startswith(">"):
                if name != "" and seq:
                    contigs[name] = "".join(seq)
                    names.append(name)
                name, seq = line.strip()[1:].split(" ")[0], []
            else:
                seq.append(line.strip().upper())
        contigs[name] = "".join(seq)
        names.append(name)
    with open(reference, "w") as f_save:
        for name in names:
            f_save.write(">" + name + "\n" + contigs[name] + "\n")
    return reference


def genome_segments(reference, chrom_seg_length):
    """multi_line (Util.py:1801) + the naming of split_genome_chunks.py:41-52 without the temporary file: yields
    ('chr$offset', segment, len of the '>chr\\toffset\\tsegment' text line) in genome order"""
    names, contigs = read_fasta(reference)
    for name in names:
        contig = contigs[name]
        for start in range(0, len(contig), chrom_seg_length):
            seg = contig[start:start + chrom_seg_length]
            yield name + "$" + str(start), seg, len(">" + name + "\t" + str(start) + "\t" + seg)


def split_chromosomes(chromosomes_dict, max_length=200_000_000):
    """Util.py:10252 -- sequences longer than max_length become <name>_part<k> pieces"""
    out = {}
    for chrom, sequence in chromosomes_dict.items():
        if len(sequence) <= max_length:
            out[chrom] = sequence
            continue
        out.update(("%s_part%d" % (chrom, k + 1), sequence[o:o + max_length]) for k, o in enumerate(range(0, len(sequence), max_length)))
    return out


def split_dict_into_blocks(chromosomes_dict, threads, chunk_size):
    """Util.py:10276 -- consecutive sequences are grouped until a block holds total / threads bases: on the prefix sums of the
    lengths, a block that starts at sequence `first` ends at the first sequence where the sum since `first` reaches the target"""
    items = list(split_chromosomes(chromosomes_dict, max_length=chunk_size).items())
    ends = np.cumsum([len(seq) for _name, seq in items], dtype=np.int64)
    target = int(ends[-1]) // threads if len(items) else 0
    blocks, first = [], 0
    while first < len(items):
        before = int(ends[first - 1]) if first else 0
        last = int(np.searchsorted(ends, before + target, side="left"))       # first index with ends[last] - before >= target
        last = max(first, min(last, len(items) - 1))
        blocks.append(dict(items[first:last + 1]))
        first = last + 1
    return blocks


def split_genome_chunks(reference, tmp_output_dir, chrom_seg_length, chunk_size_mb):
    """module/split_genome_chunks.py:27-88: genome.cut{i}.fa (chr$offset segments, a new chunk whenever the FASTA-text bytes
    read reach chunk_size) and ref_chr/ref_block_{i}.fa; returns the chunk paths"""
    chunk_size = int(float(chunk_size_mb) * 1024 * 1024)
    tmp_output_dir = os.path.abspath(tmp_output_dir)
    os.makedirs(tmp_output_dir, exist_ok=True)
    convertToUpperCase_v1(reference)
    cut_references, cur, cur_base_num, ref_index = [], {}, 0, 0
    for seg_name, seg, line_len in genome_segments(reference, int(chrom_seg_length)):
        cur[seg_name] = seg
        cur_base_num += line_len
        if cur_base_num >= chunk_size:
            path = os.path.join(tmp_output_dir, "genome.cut%d.fa" % ref_index)
            store_fasta(cur, path)
            cut_references.append(path)
            cur, cur_base_num, ref_index = {}, 0, ref_index + 1
    if cur:
        path = os.path.join(tmp_output_dir, "genome.cut%d.fa" % ref_index)
        store_fasta(cur, path)
        cut_references.append(path)
    _names, ref_contigs = read_fasta(reference)
    split_ref_dir = os.path.join(tmp_output_dir, "ref_chr")
    shutil.rmtree(split_ref_dir, ignore_errors=True)
    os.makedirs(split_ref_dir)
    for i, block in enumerate(split_dict_into_blocks(ref_contigs, 100, chunk_size)):
        chr_path = os.path.join(split_ref_dir, "ref_block_%d.fa" % i)
        store_fasta(block, chr_path)
        if shutil.which("makeblastdb"):     # only the blastn route of the reference reads these databases
            subprocess.run("makeblastdb -in %s -dbtype nucl > /dev/null 2>&1" % chr_path, shell=True, check=False)
    return cut_references


def file_exist(result_path):
    """Util.py:2831 -- the reference's 'did this stage succeed' test: a FASTA with at least one record, any other file with
    a non-comment non-blank line, a non-empty directory"""
    if os.path.isdir(result_path):
        with os.scandir(result_path) as entries:
            return next(entries, None) is not None
    if not os.path.isfile(result_path) or os.path.getsize(result_path) == 0:
        return False
    if result_path.endswith((".fa", ".fasta")):
        return bool(read_fasta(result_path)[1])
    with open(result_path, "r") as handle:
        return any(ln.strip() != "" and not ln.startswith("#") for ln in handle)


def update_prev_TE(prev_TE, cur_file):
    """Util.py:6378 -- append cur_file (+ newline) to prev_TE under a lock file (several stage scripts share prev_TE)"""
    if not os.path.exists(cur_file):
        print("Warning: %s not found, skipping" % cur_file)
        return
    import fcntl

    with open(prev_TE + ".lock", "w") as lock_f:
        fcntl.flock(lock_f, fcntl.LOCK_EX)
        try:
            with open(prev_TE, "a+") as target_f, open(cur_file, "r") as source_f:
                target_f.write(source_f.read() + "\n")
        finally:
            fcntl.flock(lock_f, fcntl.LOCK_UN)


def set_reference(reference, device=0):
    """pack the genome into HBM once per process (replaces the per-stage read_fasta(reference))"""
    ctx = get_ctx(device)
    if _PACKED["path"] != os.path.abspath(reference):
        names, contigs = read_fasta(reference)
        ctx.genome_pack([contigs[n] for n in names])
        ctx.set_contig_order(names)       # length ties between alignment rows fall to the row NAME (tools/ready_for_MSA.sh)
        ctx.release_copy_index()          # the minimizer index belongs to the genome that was packed before
        _PACKED.update(path=os.path.abspath(reference), names={n: i for i, n in enumerate(names)},
                       lens=[len(contigs[n]) for n in names])
    return ctx


def _read_alignment(align_file):
    names, contigs = read_fasta(align_file)
    if not names:
        return names, None
    rows = [contigs[n] for n in names]
    width = len(rows[0])
    if any(len(r) != width for r in rows):
        raise ValueError("alignment rows differ in length: " + align_file)
    return names, np.frombuffer("".join(rows).encode(), dtype=np.uint8).reshape(len(rows), width).copy()


# ---- a-14 .. a-21 ------------------------------------------------------------------------------------
def remove_sparse_col_in_align_file(align_file):
    """Util.py:10344-10405: writes <align_file>.clean.fa and returns its path"""
    names, m = _read_alignment(align_file)
    clean = get_ctx().sparse_cols([m])[0]
    out = align_file + ".clean.fa"
    with open(out, "w") as f:
        for n, r in zip(names, clean):
            f.write(">" + n + "\n" + r.tobytes().decode() + "\n")
    return out


def _judge(te_type, cur_seq, align_file, plant, result_type):
    if result_type != "cons":
        raise NotImplementedError("only result_type='cons' (the only value the reference's callers pass)")
    names, m = _read_alignment(align_file)
    if m is None:
        return False, "nb", "", 0
    r = get_ctx().judge(te_type, [m], [cur_seq], plant=int(plant))[0]
    if r[1] == "EXC":
        raise RuntimeError("the reference raises on this input (empty candidate / no full-length row / index error)")
    return r[0], r[1], r[2], r[3]


def judge_boundary_v5(cur_seq, align_file, debug, TE_type, plant, result_type):
    """Util.py:9145-9480 (TIR)"""
    return _judge("tir", cur_seq, align_file, plant, result_type)


def judge_boundary_v6(cur_seq, align_file, debug, TE_type, plant, result_type):
    """Util.py:9821-10159 (Helitron)"""
    return _judge("helitron", cur_seq, align_file, plant, result_type)


def judge_boundary_v9(cur_seq, align_file, debug, TE_type, plant, result_type):
    """Util.py:9483-9720 (non-LTR)"""
    return _judge("non_ltr", cur_seq, align_file, plant, result_type)


def TSDsearch_v5(raw_align_seq, cur_boundary_start, cur_boundary_end, plant):
    """Util.py:2460-2492"""
    l, r, _k = get_ctx().tsd_search([raw_align_seq], [cur_boundary_start], [cur_boundary_end], int(plant))[0]
    return l, r


def search_boundary_homo_v3(valid_col_threshold, pos, matrix, row_num, col_num, type, homo_threshold, debug,
                            int_sliding_window_size, out_sliding_window_size):
    """Util.py:8887-9143; `matrix` = list of rows (lists/strings of single characters)"""
    m = np.frombuffer("".join("".join(r) for r in matrix).encode(), dtype=np.uint8).reshape(row_num, col_num).copy()
    b, _ = get_ctx().boundary_search([m], [pos], [type], [homo_threshold], variant=3, win_in=int_sliding_window_size,
                                     win_out=out_sliding_window_size)
    return int(b[0])


def search_boundary_homo_v4(valid_col_threshold, pos, matrix, row_num, col_num, type, homo_threshold, int_homo_threshold,
                            out_homo_threshold, debug, int_sliding_window_size, out_sliding_window_size):
    """Util.py:8556-8824"""
    m = np.frombuffer("".join("".join(r) for r in matrix).encode(), dtype=np.uint8).reshape(row_num, col_num).copy()
    b, v = get_ctx().boundary_search([m], [pos], [type], [homo_threshold], variant=4, int_thr=[int_homo_threshold],
                                     out_thr=[out_homo_threshold], win_in=int_sliding_window_size,
                                     win_out=out_sliding_window_size)
    return bool(v[0]), int(b[0])


# ---- a-6 ---------------------------------------------------------------------------------------------
def _read_matrix(matrix_file, column):
    rows = []
    with open(matrix_file) as f:
        for line in f:
            rows.append(line.replace("\n", "").split("\t")[column])
    return rows


def get_both_ends_frame(query_name, cur_seq, align_file, output_dir, full_length_output_dir, flanking_len, debug, device=0):
    """bin/FiLTR-main/src/Util.py:1401 -- aligned copies of one LTR terminal -> <output_dir>/<query>.matrix
    ('left_frame\\tright_frame' per copy) and <full_length_output_dir>/<query>.matrix; (None, None) without a boundary"""
    align_names, align_contigs = read_fasta(align_file)
    if not align_names:
        return None, None
    res = get_ctx(device).ltr_both_ends([[align_contigs[n] for n in align_names]], [cur_seq], flanking_len)[0]
    if res is None:
        return None, None
    frames, full, _ns, _ne = res
    start_align_file = os.path.join(output_dir, query_name + ".matrix")
    with open(start_align_file, "w") as f_save:
        for left, right in frames:
            f_save.write(left + "\t" + right + "\n")
    full_length_align_file = os.path.join(full_length_output_dir, query_name + ".matrix")
    with open(full_length_align_file, "w") as f_save:
        for row in full:
            f_save.write(row + "\n")
    return start_align_file, full_length_align_file


def judge_left_frame_LTR(matrix_file, flanking_len, sliding_window_size=20):
    """bin/FiLTR-main/src/Util.py:9327: (is_ltr, new_boundary_start) from the left frames of the '.matrix' file"""
    rows = _read_matrix(matrix_file, 0)
    return (True, -1) if len(rows) <= 1 else get_ctx().ltr_frame([rows], flanking_len, sliding_window_size, "left")[0]


def judge_right_frame_LTR(matrix_file, flanking_len, sliding_window_size=20):
    """bin/FiLTR-main/src/Util.py:9175: (is_ltr, new_boundary_end) from the right frames of the '.matrix' file"""
    rows = _read_matrix(matrix_file, 1)
    return (True, -1) if len(rows) <= 1 else get_ctx().ltr_frame([rows], flanking_len, sliding_window_size, "right")[0]


_COMP = {"A": "T", "T": "A", "C": "G", "G": "C"}


def getReverseSequence(sequence):
    """reverse complement, every non-ACGT symbol -> N (Util.py:1635)"""
    return "".join(_COMP.get(b, "N") for b in reversed(sequence))


def get_short_tir_contigs(cur_itr_contigs, plant):
    """terminal-structure shortcuts of Util.py:7297-7334: variants whose first 5 bases equal the reverse complement of
    their last 5 are kept without itrsearch when the TSD length in the name says hAT (8, < 4 kb), Mutator (9-11) or --
    plants, CACTA/CACTG start -- CACTA (3); so are variants framed by CCC ... GGG."""
    keep = {}
    for name, seq in cur_itr_contigs.items():
        tsd_len = len(name.split("-tsd_")[1].split("-")[0]) if "-tsd_" in name else 0
        head5, tail5 = seq[:5], getReverseSequence(seq[len(seq) - 5:])
        head3, tail3 = seq[:3], getReverseSequence(seq[len(seq) - 3:])
        if head5 == tail5:
            if (tsd_len == 8 and len(seq) < 4000) or 9 <= tsd_len <= 11 or \
                    (plant == 1 and tsd_len == 3 and head5 in ("CACTA", "CACTG")):
                keep[name] = seq
        elif head3 == tail3 == "CCC":
            keep[name] = seq
    return keep


def filter_dup_itr_v3(cur_copies_out_contigs, TIR_len_dict):
    """Util.py:2791-2812: of the variants of one query keep the one with the smallest '-distance_' (first wins ties),
    renamed '<query>-tir_<len>-tsd_<seq>'; >= 30 kb is dropped."""
    best, best_d = "", 100000
    for name in cur_copies_out_contigs:
        d = int(name.split("-distance_")[1])
        if d < best_d:
            best, best_d = name, d
    if not best:
        return {}
    query, rest = best.split("-C_")[0], best.split("-C_")[1]
    tsd = rest.split("-tsd_")[1].split("-")[0]
    seq = cur_copies_out_contigs[best]
    return {"%s-tir_%d-tsd_%s" % (query, TIR_len_dict.get(best, 0), tsd): seq} if len(seq) < 30000 else {}


def run_itrsearch(contigs, end_len=40, device=0, ctx=None):
    """run_itrsearch (Util.py:216-224: `tools/itrsearch -i 0.7 -l 7 <fasta>`) on {name: sequence}: the in-tree stage
    (hite_itr_search; defined by oracle/hite_oracle_itr.c, pinned to the tool's own output).  end_len = 40: the records are the
    first 40 + last 40 bases (Util.py:6564, 6577); end_len = 0: whole sequences (remove_no_tirs, Util.py:13907).
    -> (names the tool writes to <input>.itr, in input order; {name: the "Length itr=" of its header})."""
    names = list(contigs.keys())
    res = (ctx or get_ctx(device)).itr_search([contigs[n] for n in names], end_len=end_len, min_identity=0.7, min_len=7)
    found = [n for n, r in zip(names, res) if r[5]]
    return found, {n: int(r[6]) for n, r in zip(names, res) if r[5]}


def tir_variants_with_structure(names, contigs, flanking_len, plant, device=0, ctx=None):
    """the first two thirds of search_confident_tir_batch_v1 (Util.py:6533-6604): k-mer TSD variants on the GPU
    (search_confident_tir_v4, names '<q>-C_<i>-tsd_<kmer>-distance_<d>' in the canonical order (distance, start, end, k)), the
    terminal-structure shortcuts (get_short_tir_contigs), the terminal-inverted-repeat filter for the rest (a variant itrsearch
    does not report is dropped, Util.py:6598-6600)  -> ({variant: sequence} that stay, {variant: TIR length})."""
    ctx = ctx or get_ctx(device)
    names = [n for n in names if "NNNNNNNNNN" not in contigs[n]]
    recs = ctx.tsd_kmer([contigs[n] for n in names], flank=flanking_len, plant=plant)
    variants = {}
    for n, rr in zip(names, recs):
        seq = contigs[n]
        for i, (k, ts, te, d) in enumerate(rr):
            variants["%s-C_%d-tsd_%s-distance_%d" % (n, i, seq[ts - k:ts], d)] = seq[ts:te + 1]
    short = get_short_tir_contigs(variants, plant)
    rest = {n: s for n, s in variants.items() if n not in short}
    tir_len = {}
    _f, lens = run_itrsearch(short, 40, ctx=ctx) if short else ([], {})
    found, lens_rest = run_itrsearch(rest, 40, ctx=ctx) if rest else ([], {})
    kept = {n: rest[n] for n in found}
    tir_len.update(lens_rest)
    tir_len.update(lens)        # (the reference reads the short set's lengths last, Util.py:6593-6596)
    kept.update(short)
    return kept, tir_len


def search_confident_tir_batch_v1(names, contigs, flanking_len, plant, work_dir=None, device=0, ctx=None):
    """search_confident_tir_batch_v1 (Util.py:6533-6628) for one batch: the variants that keep a terminal structure
    (tir_variants_with_structure), then one variant per query (filter_dup_itr_v3: the smallest '-distance_')."""
    kept, tir_len = tir_variants_with_structure(names, contigs, flanking_len, plant, device, ctx)
    groups = {}
    for n, s in kept.items():
        groups.setdefault(n.split("-C_")[0], {})[n] = s
    out = {}
    for q in groups:
        out.update(filter_dup_itr_v3(groups[q], tir_len))
    return out


def remove_no_tirs(tir_contigs, plant, device=0, ctx=None):
    """remove_no_tirs (Util.py:13897-13920) on {name: sequence}: sequences with a short-TIR signature stay, the rest is
    searched whole for a terminal inverted repeat (itrsearch -i 0.7 -l 7)  -> ({with a TIR}, {without}); the first in the
    reference's order (the tool's output order, then the short-TIR set)."""
    short = get_short_tir_contigs(tir_contigs, plant)
    rest = {n: s for n, s in tir_contigs.items() if n not in short}
    found, _lens = run_itrsearch(rest, 0, device, ctx) if rest else ([], {})
    with_tir = {n: rest[n] for n in found}
    with_tir.update(short)
    return with_tir, {n: s for n, s in tir_contigs.items() if n not in with_tir}


def search_polyA_TSD_batch(seqs, flanking_len=50, end_5_window_size=25, device=0):
    """search_polyA_TSD (Util.py:10915) for a batch of flanked repeats -> [(found_TSD, TSD_seq, non_ltr_seq)]"""
    res = get_ctx(device).nonltr_prep(seqs, flanking_len, end_5_window_size)
    out = []
    for s, (found, direct, ts, tn, lo, hi) in zip(seqs, res):
        nl = s[lo:hi] if direct else ""
        if direct == 2:
            nl = getReverseSequence(nl)
        out.append((bool(found), s[ts:ts + tn] if found else "", nl))
    return out


def get_candidate_non_LTR(longest_repeats_flanked_path, flanking_len=50, device=0):
    """get_candidate_non_LTR (Util.py:11009-11045): flanked repeats of 100-700 bp (SINE) / 700-8000 bp (LINE) with a polyA/T
    or tandem tail and a TSD -> ({name\tTSD:seq: element}, {...}); the element must fall into the same length class."""
    names, contigs = read_fasta(longest_repeats_flanked_path)
    sine = [n for n in names if 100 <= len(contigs[n]) - 2 * flanking_len <= 700]
    line = [n for n in names if 700 < len(contigs[n]) - 2 * flanking_len <= 8000]
    res = search_polyA_TSD_batch([contigs[n] for n in sine + line], flanking_len, 25, device)
    cand_sine, cand_line = {}, {}
    for i, n in enumerate(sine + line):
        found, tsd, seq = res[i]
        if not found:
            continue
        if i < len(sine) and 100 <= len(seq) <= 700:
            cand_sine[n + "\tTSD:" + tsd] = seq
        elif i >= len(sine) and 700 < len(seq) <= 8000:
            cand_line[n + "\tTSD:" + tsd] = seq
    return cand_sine, cand_line


def get_query_copies(cur_segments, query_contigs, subject_path, query_coverage, subject_coverage, query_fixed_extend_base_threshold=200,
                     subject_fixed_extend_base_threshold=200, max_copy_num=100, device=0, ctx=None):
    """Util.py:6828 -- cur_segments = [(query_name, {subject_name: [(q_start, q_end, s_start, s_end, identity)]})] ->
    {query_name: [(subject_name, start, end, chain_len, '+'/'-')]}.  One device call for all queries."""
    subject_contigs = read_fasta(subject_path)[1] if subject_coverage > 0 else {}
    qnames, snames, sidx = [], [], {}
    qid, sid, qs, qe, ss, se, ident = [], [], [], [], [], [], []
    for query_name, subject_dict in cur_segments:
        q = len(qnames)
        qnames.append(query_name)
        for subject_name, subject_pos in subject_dict.items():
            s = sidx.setdefault(subject_name, len(snames))
            if s == len(snames):
                snames.append(subject_name)
            for (a, b, c, d, idt) in subject_pos:
                qid.append(q); sid.append(s); qs.append(a); qe.append(b); ss.append(c); se.append(d); ident.append(idt)
    all_copies = {}
    if not qnames:
        return all_copies
    qlen = [len(query_contigs[name]) for name in qnames]
    slen = [len(subject_contigs[name]) for name in snames] if subject_coverage > 0 else None
    res = (ctx or get_ctx(device)).query_copies(qid, sid, qs, qe, ss, se, ident, qlen, slen, ns=max(1, len(snames)), qcov=query_coverage,
                                       scov=subject_coverage, qthr=query_fixed_extend_base_threshold,
                                       sthr=subject_fixed_extend_base_threshold, max_copy=max_copy_num)
    for q, name in enumerate(qnames):
        all_copies[name] = [(snames[c[0]], c[1], c[2], c[3], c[4]) for c in res[q]]
    return all_copies


def get_copies_v1(blastnResults_path, query_path, subject_path, query_coverage=0.95, subject_coverage=0, device=0, ctx=None):
    """Util.py:7032 -- blast6 table -> {query: copies}; lines with query == subject name are skipped"""
    query_records = {}
    with open(blastnResults_path) as f_r:
        for line in f_r:
            parts = line.split("\t")
            if len(parts) < 10 or parts[0] == parts[1]:
                continue
            query_records.setdefault(parts[0], {}).setdefault(parts[1], []).append(
                (int(parts[6]), int(parts[7]), int(parts[8]), int(parts[9]), float(parts[2])))
    _names, query_contigs = read_fasta(query_path)
    return get_query_copies(list(query_records.items()), query_contigs, subject_path, query_coverage, subject_coverage, device=device, ctx=ctx)


def _chain_all_by_name(query_records, gap_of, ctx):
    """query_records: {query: {subject: [(q_start, q_end, s_start, s_end), ...]}} in the reference's insertion order;
    gap_of(query) -> its skip_gap (real).  -> {query: [(q_start, q_end, length, s_start, s_end, |s span|, subject, extend_num)]}:
    the reference's longest_queries tuples, per query, in its order (hite_chain_all)"""
    import math

    qnames = list(query_records.keys())
    snames, sidx = [], {}
    qid, sid, qs, qe, ss, se = [], [], [], [], [], []
    for qi, q in enumerate(qnames):
        for sname, frags in query_records[q].items():
            si = sidx.setdefault(sname, len(sidx))
            if si == len(snames):
                snames.append(sname)
            for f in frags:
                qid.append(qi); sid.append(si); qs.append(f[0]); qe.append(f[1]); ss.append(f[2]); se.append(f[3])
    out = {q: [] for q in qnames}
    if not qid:
        return out
    # an integer distance is below the real skip_gap exactly when it is below its ceiling
    gaps = [int(math.ceil(gap_of(q))) for q in qnames]
    chains = ctx.chain_all(qid, sid, qs, qe, ss, se, len(qnames), len(snames), gaps)
    for q, lst in zip(qnames, chains):
        for (si, a, b, c, d, nx) in lst:
            out[q].append((a, b, (b - a) if nx else abs(b - a), c, d, abs(d - c), snames[si], nx))
    return out


def FMEA(blastn2Results_path, fixed_extend_base_threshold, device=0, ctx=None):
    """FMEA (Util.py:10452-10645), same arguments: blast6 table -> {query_name: [(query_name, q_start - 1, q_end, subject_name,
    s_start - 1, s_end), ...]} -- every chain of get_longest_repeats_v4's chaining core with a fixed skip_gap, no
    de-duplication (deredundant_for_LTR, Util.py:10750).  A zero-length HSP raises ZeroDivisionError (the reference divides
    by the fragment lengths, :10550-10554, wherever such a fragment meets another one; here: always)."""
    query_records = {}
    with open(blastn2Results_path) as f_r:
        for line in f_r:
            parts = line.split("\t")
            query_name, subject_name = parts[0], parts[1]
            float(parts[2]); int(parts[3])                     # (identity, alignment length: parsed, unused -- a malformed line raises as in the reference)
            q_start, q_end, s_start, s_end = int(parts[6]), int(parts[7]), int(parts[8]), int(parts[9])
            if query_name == subject_name and q_start == s_start and q_end == s_end:
                continue
            if q_start == q_end or s_start == s_end:
                raise ZeroDivisionError("zero-length HSP")
            query_records.setdefault(query_name, {}).setdefault(subject_name, []).append((q_start, q_end, s_start, s_end))
    if ctx is None:
        ctx = get_ctx(device)
    chains = _chain_all_by_name(query_records, lambda _q: fixed_extend_base_threshold, ctx)
    return {q: [(q, r[0] - 1, r[1], r[6], r[3] - 1, r[4]) for r in lst] for q, lst in chains.items()}


def get_full_length_copies_from_blastn_v1(TE_lib, reference, blastn_out, tmp_output_dir, threads, divergence_threshold,
                                          full_length_threshold, search_struct, tools_dir, device=0, ctx=None):
    """get_full_length_copies_from_blastn_v1 (Util.py:5907-6135), same arguments: blast6 table of a TE library against a genome
    -> ({query: {'chr:start-end': sequence | '1'}}, the same with flanks): the chains (skip_gap = len(query) * threshold) that
    cover >= threshold of their query; names lose their '#class' suffix; with search_struct the sequences (flank 5 for
    'Helitron' names, else 50) come from `reference`, else '1' as in the reference."""
    ref_names, ref_contigs = read_fasta(reference) if search_struct else ([], {})
    query_names, query_contigs = read_fasta(TE_lib)
    query_contigs = {name.split("#")[0]: query_contigs[name] for name in query_names}
    query_records = {}
    with open(blastn_out) as f_r:
        for line in f_r:
            if line.startswith("#"):
                continue
            info_parts = line.split("\t")
            query_name = info_parts[0].split("#")[0]
            query_records.setdefault(query_name, {}).setdefault(info_parts[1], []).append(
                (int(info_parts[6]), int(info_parts[7]), int(info_parts[8]), int(info_parts[9])))
    known = {q: v for q, v in query_records.items() if q in query_contigs}       # (`continue` at :5943: no entry in the results)
    if ctx is None:
        ctx = get_ctx(device)
    chains = _chain_all_by_name(known, lambda q: len(query_contigs[q]) * full_length_threshold, ctx)
    full_length_copies, flank_full_length_copies = {}, {}
    for query_name in query_records.keys():
        if query_name not in query_contigs:
            continue
        query_len = len(query_contigs[query_name])
        flanking_len = 5 if "Helitron" in str(query_name) else 50
        query_copies, flank_query_copies = {}, {}
        for repeat in chains[query_name]:
            if repeat[2] < full_length_threshold * query_len:
                continue
            subject_name = repeat[6]
            if repeat[3] > repeat[4]:
                start, end = repeat[4] - 1, repeat[3]
            else:
                start, end = repeat[3] - 1, repeat[4]
            subject_pos = subject_name + ":" + str(start) + "-" + str(end)
            if search_struct:
                subject_seq = ref_contigs[subject_name][start:end]
                flank_subject_seq = ref_contigs[subject_name][start - flanking_len:end + flanking_len]
            else:
                subject_seq = flank_subject_seq = "1"
            if float(repeat[2]) / query_len >= full_length_threshold:
                query_copies[subject_pos] = subject_seq
                flank_query_copies[subject_pos] = flank_subject_seq
        full_length_copies[query_name] = query_copies
        flank_full_length_copies[query_name] = flank_query_copies
    return full_length_copies, flank_full_length_copies


def generate_full_length_out_v1(BlastnOut, TE_lib, reference, tmp_output_dir, tools_dir, full_length_threshold, category, debug=0,
                                device=0, ctx=None):
    """generate_full_length_out_v1 (Util.py:6288-6316), same arguments: the full-length copies of every library sequence as
    pickled sets of (query_name, chr_name, chr_start, chr_end) (1-based, inclusive), one file per batch of 500 queries --
    what mask_genome_intactTE (:6406-6417) reads.  category 'Total' keeps every line of the table, another value those
    whose query class (after '#') contains it (filter_out_by_category :5014).  The structure search of the reference's
    get_structure_info_v1 is switched off in this call there as well (search_struct = False)."""
    import pickle

    os.makedirs(tmp_output_dir, exist_ok=True)
    filter_tmp_out = os.path.join(tmp_output_dir, "tmp.out")
    shutil.copy(BlastnOut, filter_tmp_out)
    if category != "Total":
        kept = [line for line in open(filter_tmp_out) if category in line.split("\t")[0].split("#")[1]]
        filter_tmp_out = os.path.join(tmp_output_dir, "tmp.filter.out")
        with open(filter_tmp_out, "w") as f_save:
            f_save.writelines(kept)
    if os.path.exists(BlastnOut) and debug != 1:
        os.remove(BlastnOut)
    full_length_copies, _flank = get_full_length_copies_from_blastn_v1(TE_lib, reference, filter_tmp_out, tmp_output_dir, 1, 20,
                                                                       full_length_threshold, False, tools_dir, device=device, ctx=ctx)
    if os.path.exists(filter_tmp_out) and debug != 1:
        os.remove(filter_tmp_out)
    cluster_dir = os.path.join(tmp_output_dir, "cluster")
    shutil.rmtree(cluster_dir, ignore_errors=True)
    os.makedirs(cluster_dir)
    all_query_names = list(full_length_copies.keys())
    output_files = []
    for batch_start in range(0, len(all_query_names), 500):
        lines = set()
        for query_name in all_query_names[batch_start:batch_start + 500]:
            for chr_pos in full_length_copies[query_name].keys():
                chr_name, span = chr_pos.split(":")[0], chr_pos.split(":")[1].split("-")
                lines.add((str(query_name), chr_name, int(span[0]) + 1, int(span[1])))
        output_file = os.path.join(cluster_dir, "result_%d.pkl" % batch_start)
        with open(output_file, "wb") as f:
            pickle.dump(lines, f)
        output_files.append(output_file)
    return output_files


def multiple_alignment_blast_and_get_copies_v1(repeats_path, align_fn=None, device=0, ctx=None):
    """multiple_alignment_blast_and_get_copies_v1 (Util.py:7179-7210), same argument (query FASTA, directory of per-chromosome
    '.fa' databases, scratch blast6 path): the chromosomes are searched one after the other (os.listdir order), the copies of
    every query accumulate (get_copies_v1), a query that has reached 100 copies leaves the query file.  align_fn(chr_path,
    query_path, out_path) writes the blast6 table of one chromosome: by default `blastn -evalue 1e-20 -outfmt 6` as in the
    reference when it is installed (it is external), else the build's copy finder, one line per copy."""
    split_repeats_path, split_ref_dir, blastn2Results_path = repeats_path[0], repeats_path[1], repeats_path[2]
    if os.path.exists(blastn2Results_path):
        os.remove(blastn2Results_path)
    if align_fn is None:
        align_fn = _blastn_one_chromosome if shutil.which("blastn") else (lambda c, q, o: _copies_as_blast6(c, q, o, device))
    all_copies = {}
    repeat_names, repeat_contigs = read_fasta(split_repeats_path)
    remain_contigs = repeat_contigs
    for chr_name in os.listdir(split_ref_dir):
        if len(remain_contigs) > 0:
            if not str(chr_name).endswith(".fa"):
                continue
            align_fn(split_ref_dir + "/" + chr_name, split_repeats_path, blastn2Results_path)
            cur_all_copies = get_copies_v1(blastn2Results_path, split_repeats_path, "", device=device, ctx=ctx)
            for query_name in cur_all_copies.keys():
                update_copy_list = all_copies.get(query_name, []) + cur_all_copies[query_name]
                all_copies[query_name] = update_copy_list
                if len(update_copy_list) >= 100:
                    del repeat_contigs[query_name]
            remain_contigs = repeat_contigs
            store_fasta(remain_contigs, split_repeats_path)
    return all_copies


def _blastn_one_chromosome(chr_path, query_path, out_path):
    subprocess.run("blastn -db %s -num_threads 1 -query %s -evalue 1e-20 -outfmt 6 > %s" % (chr_path, query_path, out_path), shell=True, check=False)


def _copies_as_blast6(chr_path, query_path, out_path, device=0):
    """the build's stand-in for one blastn run: the copies hite_find_copies reports for the queries in the sequences of
    chr_path, one blast6 line per copy (the whole query against the copy's interval)"""
    names, contigs = read_fasta(chr_path)
    qnames, queries = read_fasta(query_path)
    ctx = get_ctx(device)
    ctx.genome_pack([contigs[n].upper() for n in names])
    ctx.release_copy_index()
    _PACKED["path"] = None
    # (the index of this chunk serves this one query set -- the caller masks the chunk next: built for these queries only)
    tab = ctx.find_copies([queries[q].upper() for q in qnames], restricted=True) if qnames and names else []
    with open(out_path, "w") as f:
        for q, copies in zip(qnames, tab):
            L = len(queries[q])
            for (c, s1, e1, minus, _anch) in copies:
                a, b = (e1, s1) if minus else (s1, e1)
                f.write("%s\t%s\t%.3f\t%d\t0\t0\t1\t%d\t%d\t%d\t1e-50\t%.1f\n" % (q, names[c], 95.0, L, L, a, b, 2.0 * L))


def lib_longest_repeats(blastnResults_path, redundant_ltr, coverage_threshold, chunk_size=5_000_000, device=0):
    """process_blast_results_in_chunks + FMEA_new1_parallel_large (Util.py:12146, 12006) in one device call ->
    [{query_name: [(query_name, q_start-1, q_end, subject_name, s_start-1, s_end)]}] one dict per chunk, in chunk order"""
    names, contigs = read_fasta(redundant_ltr)
    idx = {name: i for i, name in enumerate(names)}
    qid, sid, qs, qe, ss, se = [], [], [], [], [], []
    with open(blastnResults_path) as f_r:
        for line in f_r:
            parts = line.rstrip("\n").split("\t")
            qid.append(idx[parts[0]]); sid.append(idx[parts[1]])
            qs.append(int(parts[6])); qe.append(int(parts[7])); ss.append(int(parts[8])); se.append(int(parts[9]))
    lens = [len(contigs[name]) for name in names]
    recs = get_ctx(device).lib_chain(qid, sid, qs, qe, ss, se, lens, coverage_threshold, chunk_size)
    chunks = []
    for (ch, q, a, b, s, c, d) in recs:
        while len(chunks) <= ch:
            chunks.append({})
        chunks[ch].setdefault(names[q], []).append((names[q], a, b, names[s], c, d))
    return chunks


def cluster_sequences_from_chunks(longest_repeats_chunks, contigs, coverage_threshold, device=0):
    """Util.py:12067 -- chunks as lib_longest_repeats returns them (the reference reads the same dicts from JSON files) ->
    list of clusters; each cluster is a list (query first) where the reference builds a set"""
    names = list(contigs.keys())
    idx = {name: i for i, name in enumerate(names)}
    recs = []
    for ch, chunk in enumerate(longest_repeats_chunks):
        for query_name, lst in chunk.items():
            for r in lst:
                recs.append((ch, idx[query_name], r[1], r[2], idx[r[3]], r[4], r[5]))
    lens = [len(contigs[name]) for name in names]
    return [[names[i] for i in cl] for cl in get_ctx(device).lib_cluster(recs, lens, coverage_threshold)]


def cons_from_mafft_v1(align_file, device=0):
    """Util.py:12515 -- strict-majority consensus of an aligned FASTA (None for an empty file)"""
    align_names, align_contigs = read_fasta(align_file)
    if len(align_names) <= 0:
        return None
    return get_ctx(device).msa_consensus([[align_contigs[name] for name in align_names]])[0]


def split_internal_out(merge_te_file, output_dir):
    """Util.py:14577 -- LTR internal sequences ('-int#' in the name) and the rest of a merged library go to two files"""
    names, contigs = read_fasta(merge_te_file)
    other_path, internal_path = os.path.join(output_dir, "merged_other.fa"), os.path.join(output_dir, "merged_internal.fa")
    store_fasta({n: contigs[n] for n in names if "-int#" not in n}, other_path)
    store_fasta({n: contigs[n] for n in names if "-int#" in n}, internal_path)
    return other_path, internal_path


def read_Ninja_clusters(cluster_file):
    """Util.py:12500 -- '<cluster id>\t<sequence name>' per line -> {cluster id: [names in file order]}"""
    clusters = {}
    with open(cluster_file) as f_r:
        for line in f_r:
            parts = line.rstrip("\n").split("\t")
            if len(parts) < 2:
                continue
            clusters.setdefault(int(parts[0]), []).append(parts[1])
    return clusters


CLUSTER_CLEAN_THRESHOLD = 10000   # "mafft cannot take too many rows": clusters are kept within this size (Util.py:12254)
NINJA_CUTOFF = 0.2        # generate_cons_v1 runs `Ninja --cluster_cutoff 0.2` (Util.py:12470)
STAR_MAX_LEN = 32767      # longest window the star aligner takes (include/hite_gpu.h)
SEED_MAX_SEGMENTS = 65000  # sequences per all-vs-all call (the seeding stage addresses < 65535 segments)


def ninja_stand_in(rows):
    """The build's stand-in for Ninja's re-clustering of an aligned cluster (Util.py:12468-12474; Ninja is an external tool,
    absent: PARITY UNPINNED).  Leader clustering on the aligned rows: a row joins the first earlier leader it differs from
    in <= 20 % of the columns where either has a base (a base against a gap is a difference; Ninja cuts its
    neighbour-joining tree at distance 0.2), else it becomes a leader.  rows: 2-D uint8 alignment -> list of lists of row indices (sub-clusters in order of their leaders)."""
    rows = np.asarray(rows)
    R, C = rows.shape if rows.ndim == 2 else (0, 0)
    base = rows != 45
    lead_rows = np.empty((0, C), dtype=rows.dtype)     # the leaders' rows, grown in blocks: one vectorised compare per row
    lead_base = np.empty((0, C), dtype=bool)
    n_lead, members = 0, []
    for r in range(R):
        k = -1
        if n_lead:
            # a base against a gap counts like a base against another base (the star alignment threads a short unrelated
            # member through the centre with gaps wherever that makes bases agree: its aligned bases alone look similar)
            either = lead_base[:n_lead] | base[r]
            n = either.sum(axis=1)
            diff = ((lead_rows[:n_lead] != rows[r]) & either).sum(axis=1)
            ok = np.nonzero((n > 0) & (diff <= NINJA_CUTOFF * n))[0]
            if len(ok):
                k = int(ok[0])
        if k >= 0:
            members[k].append(r)
            continue
        if n_lead == len(lead_rows):
            grow = max(16, n_lead)
            lead_rows = np.concatenate([lead_rows, np.empty((grow, C), dtype=rows.dtype)])
            lead_base = np.concatenate([lead_base, np.zeros((grow, C), dtype=bool)])
        lead_rows[n_lead], lead_base[n_lead] = rows[r], base[r]
        n_lead += 1
        members.append([r])
    return members


def _star_align_clusters(ctx, clusters):
    """clusters: list of lists of (name, sequence).  Every cluster is aligned as a star around its LONGEST member (ties: the
    first), rows returned in the cluster's own order -- the stand-in for `mafft --preservecase` (Util.py:12464, 12490; mafft
    compares case-insensitively: sequences are upper-cased first).  -> per cluster (rows: {name: aligned row}, dropped: [names
    the aligner could not place (shorter than half the centre, or an insertion / deletion beyond its widest band)])."""
    groups, orders = [], []
    for cl in clusters:
        seqs = [sq.upper() for _n, sq in cl]
        centre = max(range(len(seqs)), key=lambda i: (len(seqs[i]), -i))
        order = [centre] + [i for i in range(len(seqs)) if i != centre]
        orders.append(order)
        groups.append([seqs[i] for i in order])
    aligned, info = ctx.star_msa(groups, info=True) if groups else ([], [])
    out = []
    for cl, order, m, inf in zip(clusters, orders, aligned, info):
        rows, dropped = {}, []
        if m is None:
            out.append(({}, [n for n, _s in cl]))
            continue
        k = 0
        by_member = {}
        for pos, i in enumerate(order):
            if pos == 0 or int(inf[pos][2]) == 0:
                by_member[i] = bytes(m[k]).decode()
                k += 1
            else:
                dropped.append(cl[i][0])
        for i in range(len(cl)):               # back to the cluster's own order
            if i in by_member:
                rows[cl[i][0]] = by_member[i]
        out.append((rows, dropped))
    return out


def _generate_cons_batch(ctx, clusters, ninja=None):
    """generate_cons_v1 (Util.py:12457-12498) for a batch of clusters, each a list of (name, sequence) in file order:
    alignment -> sub-clusters (Ninja in the reference, ninja_stand_in here; `ninja` = per cluster {id: [names]} overrides it,
    which is how the goldens pin everything around the external tool) -> alignment of every sub-cluster -> strict-majority
    consensus (cons_from_mafft_v1) named after the sub-cluster's LAST member.  A cluster that yields no consensus returns
    its sequences unchanged; members the aligner dropped pass unchanged as well.  -> per cluster {name: sequence}."""
    need_first = [ci for ci in range(len(clusters)) if ninja is None or ninja[ci] is None]
    first = dict(zip(need_first, _star_align_clusters(ctx, [clusters[ci] for ci in need_first])))
    subs, owner = [], []
    passed = [dict() for _ in clusters]
    for ci, cl in enumerate(clusters):
        seq_of = dict(cl)
        if ci in first:
            rows, dropped = first[ci]
            for n in dropped:                # not placed in the cluster's alignment: passes unchanged
                passed[ci][n] = seq_of[n]
            names = [n for n, _s in cl if n in rows]
            if not names:
                continue
            mat = np.frombuffer("".join(rows[n] for n in names).encode(), dtype=np.uint8).reshape(len(names), -1)
            parts = [[names[r] for r in grp] for grp in ninja_stand_in(mat)]
        else:
            parts = [list(ninja[ci][k]) for k in ninja[ci]]
        for part in parts:
            if part:
                subs.append([(n, seq_of[n]) for n in part])
                owner.append(ci)
    second = _star_align_clusters(ctx, subs)
    als, who = [], []
    for si, (sub, (rows, dropped)) in enumerate(zip(subs, second)):
        for n in dropped:
            passed[owner[si]][n] = dict(sub)[n]
        kept = [n for n, _s in sub if n in rows]
        if kept:
            als.append([rows[n] for n in kept])
            who.append((owner[si], kept[-1]))
    cons = ctx.msa_consensus(als) if als else []
    out = [dict() for _ in clusters]
    for (ci, last), c in zip(who, cons):
        out[ci][last] = c
    for ci, cl in enumerate(clusters):
        if not out[ci]:                      # no reliable consensus: the original sequences (Util.py:12495-12498)
            out[ci] = dict(cl)
        else:
            out[ci].update(passed[ci])
    return out


def generate_cons_v1(cluster_id, cur_cluster_path, cluster_dir, threads, device=0, ninja_clusters=None):
    """generate_cons_v1 (Util.py:12457), same arguments: the FASTA of one cluster -> {name: consensus}.  ninja_clusters: the
    parsed output of Ninja ({id: [names]}, read_Ninja_clusters) when the caller has one; else the build's stand-in."""
    names, contigs = read_fasta(cur_cluster_path)
    if not names:
        return {}
    return _generate_cons_batch(get_ctx(device), [[(n, contigs[n]) for n in names]], [ninja_clusters])[0]


def _library_hits(ctx, names, contigs):
    """all-vs-all of a library with itself where the reference runs blastn (multi_process_align, Util.py:12209): the library is
    packed as a genome (one contig per sequence) and searched by hite_seed_allvsall, in blocks when it has more sequences
    than one call addresses (every pair of blocks once: the blocks of a pair are packed together).
    -> (q, s, qs, qe, ss, se) arrays over the indices of `names`."""
    n = len(names)
    half = SEED_MAX_SEGMENTS // 2
    blocks = [list(range(a, min(n, a + half))) for a in range(0, n, half)] if n > SEED_MAX_SEGMENTS else [list(range(n))]
    parts = []
    pairs = [(i, j) for i in range(len(blocks)) for j in range(i, len(blocks))] if len(blocks) > 1 else [(0, 0)]
    for (i, j) in pairs:
        ids = blocks[i] + (blocks[j] if j != i else [])
        ctx.genome_pack([contigs[names[t]].upper() for t in ids])
        ctx.release_copy_index()
        _PACKED["path"] = None
        lens = [len(contigs[names[t]]) for t in ids]
        tab = ctx.seed_allvsall(seg_len=max(lens))
        gi = np.asarray(ids, dtype=np.int64)
        q, s2 = gi[np.asarray(tab["qseg"], dtype=np.int64)], gi[np.asarray(tab["sseg"], dtype=np.int64)]
        keep = np.ones(len(q), dtype=bool)
        if j != i:          # a mixed pair contributes the hits BETWEEN its two blocks only
            inb = np.zeros(n, dtype=bool)
            inb[blocks[i]] = True
            keep = inb[q] != inb[s2]
        parts.append(tuple(np.asarray(x)[keep] for x in (q, s2, tab["qs"], tab["qe"], tab["ss"], tab["se"])))
    return tuple(np.concatenate([p[k] for p in parts]) for k in range(6))


_COMP_U8 = np.full(256, ord("N"), dtype=np.uint8)
for _a, _b in zip(b"ACGT", b"TGCA"):
    _COMP_U8[_a] = _b


def _stretch_hits(q, s, qs, qe, ss, se, lens, seqs):
    """The seeding stage reports a hit from its first to its last anchor; blastn extends an alignment to the ends of the
    sequences when they keep matching.  A hit whose ends lie within a short overhang of the sequence ends (on both sequences;
    <= 30 bases or a tenth of the shorter sequence) is therefore stretched along its diagonal -- BY THE BASES THAT ALIGN: the
    overhang is compared base by base (match +1, mismatch -2 as megablast scores, no gaps) and the hit grows to the best-scoring prefix of it,
    as an ungapped X-drop extension would leave it.  Copies of one family (<= 15 % apart) gain nearly the whole overhang and
    pass a 0.95 coverage rule they would miss by the bases outside their anchors; unrelated termini (a quarter of the bases
    agree by chance) gain a handful of bases at most and stay below it, as they do under blastn / cd-hit-est.
    seqs: the upper-case sequences behind the ids (bytes / str)."""
    L = np.asarray(lens, dtype=np.int64)
    q, s = np.asarray(q, dtype=np.int64), np.asarray(s, dtype=np.int64)
    qs, qe, ss, se = (np.asarray(x, dtype=np.int64).copy() for x in (qs, qe, ss, se))
    if len(q) == 0:
        return qs, qe, ss, se
    off = np.zeros(len(L) + 1, dtype=np.int64)
    np.cumsum(L, out=off[1:])
    buf = np.frombuffer(b"".join(x.encode() if isinstance(x, str) else bytes(x) for x in seqs), dtype=np.uint8)
    fwd = ss <= se
    left = np.minimum(qs - 1, np.where(fwd, ss - 1, L[s] - ss))
    right = np.minimum(L[q] - qe, np.where(fwd, L[s] - se, se - 1))
    reach = np.maximum(30, np.minimum(L[q], L[s]) // 10)
    left = np.where(left <= reach, left, 0)
    right = np.where(right <= reach, right, 0)

    def aligned(room, q0, qstep, s0, sstep):
        """room[h] candidate bases; base j of hit h: query q0 + qstep * j, subject s0 + sstep * j (0-based, inside the
        sequence); -> the length of the best-scoring prefix"""
        out = np.zeros(len(room), dtype=np.int64)
        idx = np.nonzero(room > 0)[0]
        CH = 4096
        for a in range(0, len(idx), CH):
            h = idx[a:a + CH]
            w = int(room[h].max())
            j = np.arange(w, dtype=np.int64)[None, :]
            ok = j < room[h][:, None]
            qi = off[q[h]][:, None] + np.where(ok, q0[h][:, None] + qstep * j, 0)
            si = off[s[h]][:, None] + np.where(ok, s0[h][:, None] + sstep[h][:, None] * j, 0)
            qb, sb = buf[qi], buf[si]
            sb = np.where(fwd[h][:, None], sb, _COMP_U8[sb])
            sc = np.where(ok, np.where((qb == sb) & (qb != ord("N")), 1, -2), -(1 << 20)).cumsum(axis=1)
            best = sc.argmax(axis=1)
            out[h] = np.where(sc[np.arange(len(h)), best] > 0, best + 1, 0)
        return out

    one = np.where(fwd, 1, -1).astype(np.int64)
    # to the left of the hit: query bases qs-2, qs-3, ... (0-based); subject ss-2, ss-3, ... (forward) / ss, ss+1, ... (reverse)
    left = aligned(left, qs - 2, -1, np.where(fwd, ss - 2, ss), -one)
    # to the right: query qe, qe+1, ...; subject se, se+1, ... (forward) / se-2, se-3, ... (reverse)
    right = aligned(right, qe, 1, np.where(fwd, se, se - 2), one)
    qs -= left; qe += right
    ss = np.where(fwd, ss - left, ss + left)
    se = np.where(fwd, se + right, se - right)
    return qs, qe, ss, se


def deredundant_for_LTR_v5(redundant_ltr, work_dir, threads, type, coverage_threshold, debug, device=0, ctx=None, stages=None):
    """deredundant_for_LTR_v5 (Util.py:12202-12337, the library de-duplication of panHiTE, config C5), same arguments.
    Reference: blastn all-vs-all of the library -> chunked fragment chaining (process_blast_results_in_chunks +
    FMEA_new1_parallel_large) -> greedy clusters (cluster_sequences_from_chunks) -> per cluster generate_cons_v1 (mafft ->
    Ninja sub-clusters -> mafft -> cons_from_mafft_v1) -> cd-hit-est.  Here: the library is packed as a genome and searched
    against itself by hite_seed_allvsall (where the reference runs blastn; libraries of >= 65 000 sequences in blocks),
    chaining / clustering / consensus are the pinned device stages (hite_lib_chain, hite_lib_cluster, hite_msa_consensus),
    the alignments are star alignments (where the reference runs mafft), the sub-clusters come from ninja_stand_in (where it
    runs Ninja); a cluster above 10 000 members is cut in file order into pieces of <= 10 000, the fall-back the reference
    itself takes when its cd-hit-est pre-reduction does not get a cluster below that size (:12252-12299; the pre-reduction
    itself needs the external tool); cd-hit-est after the consensus step runs when it is installed.  Sequences longer than
    the aligner's 32 767-base windows pass unclustered.
    Writes <redundant_ltr>.tmp.cons and <redundant_ltr>.cons, returns the former like the reference.
    ctx: the context to run on (default: the process-wide one of `device`); stages: a dict that receives the intermediate
    results ('hits', 'clusters': lists of names) -- what the parity tests compare between two runs."""
    names, contigs = read_fasta(redundant_ltr)
    cons_path, final_path = redundant_ltr + ".tmp.cons", redundant_ltr + ".cons"
    if not names:
        store_fasta({}, cons_path)
        store_fasta({}, final_path)
        return cons_path
    if ctx is None:
        ctx = get_ctx(device)
    work = [n for n in names if 0 < len(contigs[n]) <= STAR_MAX_LEN]       # the rest passes unchanged
    all_cons, clustered = {}, set()
    if work:
        q, s, qs, qe, ss, se = _library_hits(ctx, work, contigs)
        lens = [len(contigs[n]) for n in work]
        qs, qe, ss, se = _stretch_hits(q, s, qs, qe, ss, se, lens, [contigs[n].upper() for n in work])
        recs = ctx.lib_chain(q, s, qs, qe, ss, se, lens, coverage_threshold, 5_000_000)
        clusters = [cl for cl in ctx.lib_cluster(recs, lens, coverage_threshold) if len(cl) >= 1]
        clusters = [cl[a:a + CLUSTER_CLEAN_THRESHOLD] for cl in clusters for a in range(0, len(cl), CLUSTER_CLEAN_THRESHOLD)]
        if stages is not None:
            stages["hits"] = len(q)
            stages["clusters"] = [[work[i] for i in cl] for cl in clusters]
        batch = [[(work[i], contigs[work[i]]) for i in cl] for cl in clusters]
        for cl, cons in zip(clusters, _generate_cons_batch(ctx, batch) if batch else []):
            clustered.update(work[i] for i in cl)
            all_cons.update(cons)
    for n in names:      # sequences outside every cluster (and the over-long ones) pass unchanged
        if n not in clustered:
            all_cons[n] = contigs[n]
    store_fasta(all_cons, cons_path)
    if shutil.which("cd-hit-est"):
        subprocess.run("cd-hit-est -aS 0.95 -aL 0.95 -c %s -G 0 -g 1 -A 80 -i %s -o %s -T 0 -M 0 > /dev/null 2>&1" %
                       (coverage_threshold, cons_path, final_path), shell=True, check=False)
    else:
        remove_redundant_sequences(cons_path, final_path, 0.95, 0.95, device=device, ctx=ctx)     # the build's stand-in, never a plain copy
    return cons_path


def remove_redundant_sequences(inp, outp, aS=0.95, aL=0.95, device=0, ctx=None):
    """The build's stand-in for `cd-hit-est -aS 0.95 -aL 0.95 -c <c> -G 0 -g 1 -A 80 -i inp -o outp` (judge_TIR_transposons.py:87,
    Util.py:12330; cd-hit-est is an external tool: PARITY UNPINNED), used when it is not installed -- the step is never skipped.
    Greedy incremental clustering in cd-hit's order (longest first, ties in input order): a sequence joins the first longer
    representative that a chain of library-vs-library hits (hite_seed_allvsall + hite_lib_chain, the stages of the library
    merge) covers to >= aS of the shorter and >= aL of the longer sequence; otherwise it becomes a representative.  The
    representatives are written longest first, as cd-hit-est writes them.  Identity is not computed: hits are runs of shared
    15-base minimizers, which sequences below ~85 % identity hardly have (cd-hit's -c 0.8 / 0.95 asks for less / more)."""
    names, contigs = read_fasta(inp)
    work = [n for n in names if len(contigs[n]) > 0]         # (nothing is aligned here: no length limit)
    drop = set()
    if len(work) > 1:
        if ctx is None:
            ctx = get_ctx(device)
        q, s_, qs, qe, ss, se = _library_hits(ctx, work, contigs)
        lens = [len(contigs[n]) for n in work]
        qs, qe, ss, se = _stretch_hits(q, s_, qs, qe, ss, se, lens, [contigs[n].upper() for n in work])
        recs = ctx.lib_chain(q, s_, qs, qe, ss, se, lens, min(aS, aL), 5_000_000)
        covered = {}
        for (_ch, qi, a, b, si, c, d) in recs:
            if qi == si:
                continue
            cq, cs = (b - a) / lens[qi], abs(d - c) / lens[si]
            short_cov, long_cov = (cq, cs) if lens[qi] <= lens[si] else (cs, cq)
            if short_cov >= aS and long_cov >= aL:
                covered.setdefault(qi, set()).add(si)
                covered.setdefault(si, set()).add(qi)
        order = sorted(range(len(work)), key=lambda i: (-lens[i], i))
        reps = set()
        for i in order:
            if covered.get(i, set()) & reps:
                drop.add(work[i])
            else:
                reps.add(i)
    keep = [n for n in names if n not in drop]
    keep.sort(key=lambda n: -len(contigs[n]))          # (stable: ties stay in input order)
    store_fasta({n: contigs[n] for n in keep}, outp)
    return outp


def mask_genome_intactTE(TE_lib, genome_path, work_dir=None, thread=1, ref_index=0, debug=0, device=0):
    """mask_genome_intactTE (Util.py:6389-6431), step for step: the library is searched in the chunk (the reference runs
    minimap2 and converts its PAF to blast6, :6396-6398; here the build's copy finder, one blast6 line per copy:
    _copies_as_blast6), generate_full_length_out_v1 (:6406) turns the table into the full-length copies (coverage >= 0.95 of
    the library sequence), and those intervals become N -- in <genome_path>.masked and in the resident genome."""
    masked = genome_path + ".masked"
    names, contigs = read_fasta(genome_path)
    te_names, tes = read_fasta(TE_lib) if TE_lib is not None and os.path.exists(TE_lib) else ([], {})
    if not te_names:
        store_fasta(contigs, masked)
        return masked
    import pickle
    import tempfile

    ctx = get_ctx(device)
    own_dir = work_dir is None
    work = tempfile.mkdtemp(prefix="hite_mask_") if own_dir else work_dir
    os.makedirs(work, exist_ok=True)
    tmp_blast_dir = work + "/mask_tmp_" + str(ref_index)
    lib_out = work + "/prev_TE_" + str(ref_index) + ".out"
    _copies_as_blast6(genome_path, TE_lib, lib_out, device)         # (packs the chunk: it is the resident genome from here on)
    output_files = generate_full_length_out_v1(lib_out, TE_lib, genome_path, tmp_blast_dir, "", 0.95, "Total", debug=debug, device=device)
    idx = {n: i for i, n in enumerate(names)}
    cc, ss, ee = [], [], []
    for output_file in output_files:
        with open(output_file, "rb") as f:
            for _query_name, chr_name, chr_start, chr_end in sorted(pickle.load(f)):
                cc.append(idx[chr_name]); ss.append(chr_start); ee.append(chr_end)
    ctx.genome_mask(cc, ss, ee)
    arrs = [np.frombuffer(contigs[n].encode(), dtype=np.uint8).copy() for n in names]
    for c, s1, e1 in zip(cc, ss, ee):
        arrs[c][max(0, s1 - 1):e1] = ord("N")
    store_fasta({n: a.tobytes().decode() for n, a in zip(names, arrs)}, masked)
    if debug != 1:
        shutil.rmtree(tmp_blast_dir, ignore_errors=True)
        if own_dir:
            shutil.rmtree(work, ignore_errors=True)
    return masked


def split_and_store_sequences(names, contigs, base_threshold):
    """grouping rule of split_and_store_sequences (/root/reference/module/Util.py:4987-5012) without the files:
    consecutive sequences are collected until their total reaches base_threshold -> list of name lists (the reference's
    {i}_target.fa query / target files)."""
    groups, cur, count = [], [], 0
    for name in names:
        cur.append(name)
        count += len(contigs[name])
        if count >= base_threshold:
            groups.append(cur)
            cur, count = [], 0
    if cur:
        groups.append(cur)
    return groups


def run_remove_TR(target_file, trf_dir):
    """Util.py:2855-2874 -- `trf <file> 2 7 7 80 10 50 500 -f -d -m -h` in trf_dir; returns the .mask FASTA (tandem repeats -> N)"""
    os.makedirs(trf_dir, exist_ok=True)
    subprocess.run("cd %s && trf %s 2 7 7 80 10 50 500 -f -d -m -h > /dev/null 2>&1" % (trf_dir, target_file), shell=True, check=False)
    return os.path.join(trf_dir, os.path.basename(target_file) + ".2.7.7.80.10.50.500.mask")


def mask_tandem_repeats(names, contigs, device=0, max_period=500):
    """The build's own tandem-repeat masker (hite_tr_mask; definition oracle/hite_oracle_trf.c) where the reference runs TRF:
    {name: sequence} -> {name: sequence with tandem repeats as N}.  The resident genome becomes these sequences."""
    if not names or not any(len(contigs[n]) for n in names):     # an empty chunk stays an empty file, as in the reference
        return {n: contigs[n] for n in names}
    ctx = get_ctx(device)
    ctx.genome_pack([contigs[n] for n in names])
    ctx.release_copy_index()
    _PACKED["path"] = None
    m = ctx.tr_mask(max_period)
    out, pos = {}, 0
    for n in names:
        a = np.frombuffer(contigs[n].encode(), dtype=np.uint8).copy()
        a[m[pos:pos + len(a)]] = ord("N")
        out[n] = a.tobytes().decode()
        pos += len(a)
    return out


def filter_tandem_repeats(repeat_names, repeat_contigs, tmp_output_dir, ref_index, threads, device=0):
    """Util.py:4672-4697 -- the chunk is cut into files of >= 100 kb (split_and_store_sequences), every file goes through TRF,
    the masked files are concatenated (in file order: the canonical replacement of the reference's as_completed order) into
    filter_tandem_{ref_index}.fa.  `trf` is an external tool: when it is installed (and HITE_TR_MASKER is not
    "gpu") it is called exactly as the reference calls it; otherwise the chunk is masked by the build's own GPU masker
    (mask_tandem_repeats: match 2 / edit 5 -- TRF's 2 / 7 / 7 calibrated for neighbour-against-neighbour scoring --, periods <= 500, score >= 50) -- the chunk never passes unmasked."""
    out = os.path.join(tmp_output_dir, "filter_tandem_%s.fa" % ref_index)
    if shutil.which("trf") is None or os.environ.get("HITE_TR_MASKER", "") == "gpu":
        store_fasta(mask_tandem_repeats(repeat_names, repeat_contigs, device=device), out)
        return out
    tmp_dir = os.path.join(tmp_output_dir, "trf_filter_%s" % ref_index)
    os.makedirs(tmp_dir, exist_ok=True)
    jobs = []
    for i, group in enumerate(split_and_store_sequences(repeat_names, repeat_contigs, 100000)):
        sub = os.path.join(tmp_dir, str(i))
        os.makedirs(sub, exist_ok=True)
        target = os.path.join(sub, "%d_target.fa" % i)
        store_fasta({n: repeat_contigs[n] for n in group}, target)
        jobs.append((target, os.path.join(sub, "%d_trf" % i)))
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=max(1, int(threads))) as ex:     # the workers only wait for `trf` processes
        masked = list(ex.map(lambda j: run_remove_TR(*j), jobs))
    with open(out, "w") as f:
        for (target, _d), m in zip(jobs, masked):
            with open(m if os.path.exists(m) else target) as g:
                f.write(g.read())
    return out


def determine_repeat_boundary_v5(repeats_path, longest_repeats_path, prev_TE, fixed_extend_base_threshold, max_single_repeat_len,
                                 tmp_output_dir, threads, ref_index, reference, debug, device=0, ctx=None):
    """determine_repeat_boundary_v5 (Util.py:4637-4670), same positional arguments: TRF masking of the chunk
    (filter_tandem_repeats), N-masking of the full-length copies of the TEs found so far (mask_genome_intactTE with prev_TE),
    process_blast_alignments (:4724) + get_longest_repeats_v4 per query file + generate_final_result (:4783).
    repeats_path = the chunk FASTA of 'chr$offset' segments; the all-vs-all stage is hite_seed_allvsall (the build's blastn
    stand-in), FMEA runs per query file as in the reference (each call has its own first-come de-duplication), results are
    unioned by name in query-file order (the canonical replacement of the reference's as_completed order)."""
    from . import dist as hd

    # One code path for one GPU and for the ranks of a node (hite_amd/dist.py).  Under torchrun every rank is called with the same
    # arguments and shares tmp_output_dir: the steps that WRITE files (tandem masking, prev_TE masking, the result, the clean-up)
    # run on rank 0 only, the others wait at a barrier and read what it wrote; every rank seeds its share of the all-vs-all
    # stage, the HSP records go to the owners of the query files, the interval lists are gathered.
    multi, rank, world = hd.process_group_state()

    def barrier():
        if multi and world > 1:
            hd.dist.barrier()

    os.makedirs(tmp_output_dir, exist_ok=True)
    ctx = ctx or get_ctx(device)
    repeat_names, repeat_contigs = read_fasta(repeats_path)
    if not repeat_names:
        if rank == 0:
            store_fasta({}, longest_repeats_path)
        barrier()
        return longest_repeats_path
    filter_tandem_file = os.path.join(tmp_output_dir, "filter_tandem_%s.fa" % ref_index)
    masked_file_path = filter_tandem_file + ".masked"
    if rank == 0:
        filter_tandem_file = filter_tandem_repeats(repeat_names, repeat_contigs, tmp_output_dir, ref_index, threads, device=device)
        masked_file_path = mask_genome_intactTE(prev_TE, filter_tandem_file, tmp_output_dir, threads, ref_index, debug=debug, device=device)
    barrier()
    names, contigs = read_fasta(masked_file_path)
    ctx.genome_pack([contigs[n] for n in names])
    ctx.release_copy_index()
    _PACKED["path"] = None   # the reference genome has to be packed again by whoever needs it next
    seg_len = max(len(contigs[n]) for n in names)
    chroms, seg_chrom, seg_off = {}, [], []
    for n in names:
        c, off = n.split("$")
        chroms.setdefault(c, len(chroms))
        seg_chrom.append(chroms[c])
        seg_off.append(int(off))
    inv = {v: k for k, v in chroms.items()}
    # query files of >= 1 Mbp as in the reference: FMEA (with its own first-come de-duplication) runs per query file, the results
    # are unioned by name in file order
    oc, os_, oe = hd.coarse_stage_sharded(ctx, max(seg_len, 1), fixed_extend_base_threshold, max_single_repeat_len, seg_table=(seg_chrom, seg_off))
    if rank == 0:
        final = [("%s:%d-%d" % (inv[int(c)], a_, b_), inv[int(c)], int(a_), int(b_)) for c, a_, b_ in zip(oc, os_, oe)]
        _rn, ref = read_fasta(reference)
        store_fasta({name: ref[c][a_:b_] for name, c, a_, b_ in final}, longest_repeats_path)
    barrier()                 # nobody is still reading the masked chunk, and the result is there for every rank
    if rank == 0 and not debug:      # cleanup_temp_files (Util.py:4797)
        shutil.rmtree(os.path.join(tmp_output_dir, "trf_filter_%s" % ref_index), ignore_errors=True)
        for p_ in (filter_tandem_file, masked_file_path):
            if os.path.exists(p_):
                os.remove(p_)
    return longest_repeats_path


def flanking_seq(longest_repeats_path, longest_repeats_flanked_path, reference, flanking_len):
    """Util.py:4614-4634: `chr:start-end` (0-based half-open) -> `chr:(s+1-flank)-(e+flank)` with the
    window clamped into the contig; bases come from the packed genome."""
    seq_names, _ = read_fasta(longest_repeats_path)
    ctx = set_reference(reference)
    idx, lens = _PACKED["names"], _PACKED["lens"]
    contig, s1, e1, new_names = [], [], [], []
    for name in seq_names:
        ref_name, pos = name.split(":")
        a, b = pos.split("-")
        ref_start, ref_end = int(a) + 1, int(b)
        clen = lens[idx[ref_name]]
        if ref_start - 1 - flanking_len < 0:
            ref_start = flanking_len + 1
        if ref_end + flanking_len > clen:
            ref_end = clen - flanking_len
        new_names.append(ref_name + ":" + str(ref_start - flanking_len) + "-" + str(ref_end + flanking_len))
        contig.append(idx[ref_name])
        s1.append(ref_start - flanking_len)       # 1-based inclusive window, gathered with flank 0
        e1.append(ref_end + flanking_len)
    wins, _ = ctx.flank_gather(contig, s1, e1, [0] * len(contig), flank=0)
    flanked = {}
    host_ref = None
    for n, w, (c, a, b) in zip(new_names, wins, zip(contig, s1, e1)):
        if w is None:   # the gather applies the 100-bp rule of the copy windows (Util.py:8116); flanking_seq has no minimum
            if host_ref is None:
                rn, rc_ = read_fasta(reference)
                host_ref = [rc_[x] for x in rn]
            flanked[n] = host_ref[c][max(0, a - 1):max(0, b)]
        else:
            flanked[n] = w.decode()
    store_fasta(flanked, longest_repeats_flanked_path)


# ---- the fine stage (Util.py:8032-8287) -----------------------------------------------------------------
def get_full_length_copies_minimap2(query_path, reference, temp_dir=None, max_copy_num=100, threads=1, device=0):
    """Same arguments as the reference's function (Util.py:7933, which runs minimap2): {query: [(chr, start1, end1, length,
    '+'/'-'), ...]} from the build's minimizer-based copy finder on the resident genome (packed from `reference`).  The finder
    keeps up to 300 copies per query (the reference's own -N of the masking step); the <= 100 rows of an alignment are chosen
    later by the longest-100 rule (ready_for_MSA.sh 100 100), so max_copy_num / temp_dir / threads have nothing to steer here."""
    ctx = set_reference(reference, device)
    names, contigs = read_fasta(query_path)
    tab = ctx.find_copies([contigs[n] for n in names], clips=True)
    rev = {v: k for k, v in _PACKED["names"].items()}
    # (a 6th field beside the reference's five: the clip word of the record -- zero unless the records are aligned intervals,
    # hite_copy_config(1) -- which flank_region_align_v5 hands to the star alignment)
    return {n: [(rev[c], s_, e_, e_ - s_ + 1, "-" if m_ else "+", cl_) for (c, s_, e_, m_, _an, cl_) in t] for n, t in zip(names, tab) if t}


def flank_region_align_v5(candidate_sequence_path, real_TEs, flanking_len, reference, split_ref_dir, TE_type, tmp_output_dir,
                          threads, ref_index, log, subset_script_path, plant, debug, iter_num, all_low_copy,
                          result_type="cons", all_copies=None, copy_finder=None):
    """Same contract and positional arguments as the reference (Util.py:8032): reads the candidate FASTA, writes `real_TEs` and
    appends the low-copy elements to `all_low_copy`.  The copies come from the built-in copy finder (get_full_length_copies_minimap2,
    where the reference runs minimap2, Util.py:8076); `all_copies` ({query: [(chr, start1, end1, aligned_len, '+'/'-'), ...]},
    what get_full_length_copies_minimap2 returns) or `copy_finder(candidate_path, reference)` override it; one batched GPU call
    replaces the per-candidate process pool."""
    if result_type != "cons":
        raise NotImplementedError("only result_type='cons'")
    names, contigs = read_fasta(candidate_sequence_path)
    ctx = set_reference(reference)          # before the copy finder: it searches whatever genome is resident
    if all_copies is None:
        all_copies = (copy_finder or get_full_length_copies_minimap2)(candidate_sequence_path, reference)
    idx = _PACKED["names"]
    qnames = [q for q in all_copies.keys() if q in contigs]  # reference iterates all_copies (Util.py:8095)
    cands = [contigs[q] for q in qnames]
    copies = []
    for q in qnames:
        seen, lst = {}, []
        for cp in all_copies[q]:
            key = (cp[0], int(cp[1]), int(cp[2]), cp[4])  # dict semantics of copy_contigs[new_name] (Util.py:8110-8114)
            if key in seen:
                continue
            seen[key] = 1
            lst.append((idx[cp[0]], int(cp[1]), int(cp[2]), 1 if cp[4] == "-" else 0, 0, int(cp[5]) if len(cp) > 5 else 0))
        copies.append(lst)
    res, _stats = ctx.flank_region_align(TE_type, cands, copies, plant=int(plant), flank=int(flanking_len)) if qnames else ([], None)
    true_tes, low_copy = bucket_results(TE_type, [(q if is_te else None, cons if is_te else None, info, copy_count)
                                                  for q, (is_te, info, cons, copy_count, _bs, _be) in zip(qnames, res)])
    rescued, low_copy = rescue_low_copy(TE_type, low_copy, plant, os.path.join(tmp_output_dir, "low_copy_%s_%s" % (TE_type, ref_index)),
                                        threads=max(1, int(threads)))
    true_tes.update(rescued)
    store_fasta(true_tes, real_TEs)
    with open(all_low_copy, "a") as f:
        for q, s in low_copy.items():
            f.write(">" + q + "\n" + s + "\n")
    return true_tes, low_copy


def rescue_low_copy(TE_type, low_copy, plant, work_dir, tandem_masker=None, ctx=None, library_dir=None, threads=1):
    """The recall of low-copy elements (Util.py:8196-8276): the low-copy sequences go through TRF (tandem repeats -> N) when `trf`
    is installed and through the build's own masker otherwise (`tandem_masker(names, contigs) -> contigs` overrides it).  TIR
    stage: those with a short-TIR signature (get_short_tir_contigs) or a terminal inverted repeat (remove_no_tirs: the in-tree
    stage where the reference runs `itrsearch -i 0.7 -l 7`) are real TEs, with their masked sequence as the tool writes it; the
    others, and the low-copy Helitron / non-LTR candidates, are searched for intact protein domains (get_domain_info: blastx
    against <library_dir>/TIRPeps.lib | HelitronPeps.lib | non_LTR.lib, a hit over >= 95 % of a protein recalls the element with
    its unmasked sequence).  library_dir defaults to $HITE_LIBRARY_DIR, then <HiTE>/library beside the package (as scripts/judge_Other_transposons.py); blastx is an external tool: when it or the library is
    missing that recall finds nothing and the stage log says so.  -> (rescued, still low copy), both in the reference's order."""
    if not low_copy or TE_type not in _PROTEIN_LIB:
        return {}, dict(low_copy)
    # the reference always looks in <HiTE>/library (Util.py:8215-8230: cur_dir + '/library/...'); here: the argument, then
    # $HITE_LIBRARY_DIR, then <HiTE>/library beside the package (as scripts/judge_Other_transposons.py does)
    library_dir = library_dir or os.environ.get("HITE_LIBRARY_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "library")
    lib = os.path.join(library_dir, _PROTEIN_LIB[TE_type]) if library_dir else None
    can_search = shutil.which("blastx") is not None and lib is not None and os.path.exists(lib)
    if not can_search:
        # (said once per stage call, in the stage's stderr log: a user must know what this run cannot recall)
        sys.stderr.write("[hite_amd] %d low-copy %s candidate%s: blastx or the protein library (%s) is not there, nothing is recalled by its "
                         "protein domains (Util.py:8215-8276)%s\n" % (len(low_copy), TE_type, "" if len(low_copy) == 1 else "s",
                                                                      lib or "set HITE_LIBRARY_DIR", "" if TE_type == "tir" else
                                                                      ": low-copy %s elements stay in the low-copy file" % TE_type))
        if TE_type != "tir":
            return {}, dict(low_copy)
    os.makedirs(work_dir, exist_ok=True)
    masked = dict(low_copy)
    if shutil.which("trf") is not None and tandem_masker is None:
        path = os.path.join(work_dir, "low_copy.fa")
        store_fasta(low_copy, path)
        m = run_remove_TR(path, work_dir)
        if os.path.exists(m):
            _n, mc = read_fasta(m)
            masked = {n: mc[n] for n in low_copy if n in mc}
    else:
        # the build's own masker (the resident genome becomes these sequences; whoever needs the reference next packs it again)
        masked = (tandem_masker or mask_tandem_repeats)(list(low_copy.keys()), low_copy)
    rescued = {}
    to_search = masked
    if TE_type == "tir":
        rescued, to_search = remove_no_tirs(masked, plant, ctx=ctx)
    if to_search and can_search:
        cons = os.path.join(work_dir, "low_copy.%s.fa" % ("no_tir" if TE_type == "tir" else "masked"))
        table = cons + "." + TE_type + "_domain"
        store_fasta(to_search, cons)
        if get_domain_info(cons, lib, table, threads, os.path.join(work_dir, TE_type + "_domain")):
            for n in intact_domain_names(table, lib):
                if n in low_copy:
                    rescued[n] = low_copy[n]
    return rescued, {n: s_ for n, s_ in low_copy.items() if n not in rescued}


# ---- low-copy rescue by intact protein domains (Util.py:8215-8276; get_domain_info :4571-4612, multiple_alignment_blastx_v1 :1006-1262)
def _chain_domain_fragments(frags, thr):
    """one (query, protein) pair of multiple_alignment_blastx_v1 (Util.py:1055-1207): blastx fragments (q_start, q_end, s_start,
    s_end) -> [(q_start, q_end, length, s_start, s_end, s_length, extensions)] , one per cluster, in cluster order.  Forward
    fragments (q_start <= q_end) ordered by the protein interval, reverse ones by descending query position; a fragment joins the
    running cluster when it starts less than `thr` behind the end of any member (newest first); inside a cluster (ordered by the
    protein interval again) every unvisited fragment grows a chain over the later ones that advance on the protein, keep the
    strand, start < thr beyond the chain's query end and < thr / 3 beyond its protein end; the longest chain of a cluster stays
    (the first among equals).  Fragments are keyed by value: equal tuples share their visited mark, as in the reference's dict."""
    fwd = sorted((f for f in frags if not f[0] > f[1]), key=lambda x: (x[2], x[3]))
    rev = sorted((f for f in frags if f[0] > f[1]), key=lambda x: (-x[0], -x[1]))
    clusters = []
    for group, sign in ((fwd, 1), (rev, -1)):
        cur = None
        for k, f in enumerate(group):
            if k == 0:
                cur = [f]
                clusters.append(cur)
                continue
            near = any((f[0] - m[1] if sign > 0 else m[1] - f[0]) < thr for m in reversed(cur))
            if near:
                cur.append(f)
            else:
                cur = [f]
                clusters.append(cur)
    out = []
    for cl in clusters:
        cl = sorted(cl, key=lambda x: (x[2], x[3]))
        best, visited = None, set()
        for i, origin in enumerate(cl):
            if origin in visited:
                continue
            qs, qe, ss, se = origin
            length, n_ext = abs(qe - qs), 0
            visited.add(origin)
            for ext in cl[i + 1:]:
                if ext in visited or not ext[3] > se:
                    continue
                if qs < qe and ext[0] < ext[1]:
                    if ext[1] > qe:
                        if ext[0] - qe < thr and ext[2] - se < thr / 3:
                            qe, ss, se = ext[1], min(ss, ext[2]), ext[3]
                            length, n_ext = qe - qs, n_ext + 1
                            visited.add(ext)
                        elif ext[0] - qe >= thr:
                            break
                elif qs > qe and ext[0] > ext[1]:
                    if ext[1] < qe:
                        if qe - ext[0] < thr and ext[2] - se < thr / 3:
                            qe, ss, se = ext[1], min(ss, ext[2]), ext[3]
                            length, n_ext = qs - qe, n_ext + 1
                            visited.add(ext)
                        elif qe - ext[0] >= thr:
                            break
            if best is None or length > best[2]:
                best = (qs, qe, length, ss, se, se - ss, n_ext)
        if best is not None:
            out.append(best)
    return out


def _domain_overlap(pre, cur):
    """the overlap arithmetic of the table writer (Util.py:1226-1246), both intervals put in ascending order first"""
    ps, pe = (pre[0], pre[1]) if pre[0] <= pre[1] else (pre[1], pre[0])
    cs, ce = (cur[0], cur[1]) if cur[0] <= cur[1] else (cur[1], cur[0])
    if ps <= ce <= pe:
        return ce - ps if cs <= ps else ce - cs
    if ce > pe and ps <= cs <= pe:
        return pe - cs
    return 0


def blastx_domain_table(blastx_out, merge_distance=100):
    """the part of multiple_alignment_blastx_v1 (Util.py:1017-1262) behind the blastx call: `-outfmt 6` lines -> the rows of the
    domain table [(TE, protein, TE_start, TE_end, protein_start, protein_end)]: per (TE, protein) the chained fragments, per TE
    the chains by length (longest first), one dropped when more than half of it lies inside a longer one that stayed"""
    records = {}
    with open(blastx_out) as f:
        for line in f:
            parts = line.split("\t")
            if len(parts) < 10:
                continue
            records.setdefault(parts[0], {}).setdefault(parts[1], []).append((int(parts[6]), int(parts[7]), int(parts[8]), int(parts[9])))
    rows = []
    for te, subjects in records.items():
        chains = []
        for protein, frags in subjects.items():
            chains.extend(c + (protein,) for c in _chain_domain_fragments(frags, merge_distance))
        chains.sort(key=lambda x: -x[2])
        kept = []
        for c in chains:
            if all(not float(_domain_overlap(k_, c) / c[2]) > 0.5 for k_ in kept):
                kept.append(c)
        rows.extend((te, c[7], c[0], c[1], c[3], c[4]) for c in kept)
    return rows


def pet_partitions(items, partitions):
    """PET / divided_array (Util.py:1771-1798) on (name, sequence) pairs: longest first (stable), dealt to the partitions in rounds
    that take alternately from the front and from the back of that order"""
    order = sorted(items, key=lambda x: len(x[1]), reverse=True)
    parts = [[] for _ in range(partitions)]
    i, j, k, front = 0, len(order) - 1, 0, True
    while i <= j:
        if front:
            parts[k % partitions].append(order[i])
            i += 1
        else:
            parts[k % partitions].append(order[j])
            j -= 1
        k += 1
        if k % partitions == 0:
            front = not front
    return parts


def get_domain_info(cons, lib, output_table, threads, temp_dir):
    """get_domain_info (Util.py:4571-4612), same arguments: `blastx -evalue 1e-20 -outfmt 6` of the sequences of `cons`
    (in `threads` partitions dealt as the reference's PET deals them) against the protein library `lib`, the fragments chained per
    protein (blastx_domain_table), written as the reference's table: a header line, an empty line, then
    TE \t protein \t TE_start \t TE_end \t protein_start \t protein_end.  blastx (NCBI BLAST+) is an external search and
    stays one: without it the table holds the header only and a warning says so.  -> True when the search ran."""
    os.makedirs(temp_dir, exist_ok=True)
    names, contigs = read_fasta(cons)
    ran = False
    rows = []
    if names and shutil.which("blastx") is not None and lib is not None and os.path.exists(lib):
        if not all(os.path.exists(lib + ext) for ext in (".phr", ".pin", ".psq")) and shutil.which("makeblastdb"):
            subprocess.run("cd %s && makeblastdb -dbtype prot -in %s > /dev/null 2>&1" % (os.path.dirname(lib) or ".", lib), shell=True, check=False)
        for pi, part in enumerate(pet_partitions([(n, contigs[n]) for n in names], max(1, int(threads)))):
            if not part:
                continue
            query, out = os.path.join(temp_dir, "%d.fa" % pi), os.path.join(temp_dir, "%d.out" % pi)
            store_fasta(dict(part), query)
            subprocess.run("blastx -db %s -num_threads 1 -evalue 1e-20 -query %s -outfmt 6 > %s" % (lib, query, out), shell=True, check=False)
            if os.path.exists(out):        # (the reference concatenates the partitions' tables as they complete; here: in partition order)
                rows.extend(blastx_domain_table(out, 100))
        ran = True
    elif names:
        sys.stderr.write("[hite_amd] blastx or the protein library %s not found: no low-copy element is recalled by its protein domains\n" % lib)
    with open(output_table, "w") as f:
        f.write("TE_name\tdomain_name\tTE_start\tTE_end\tdomain_start\tdomain_end\n\n")
        for r in rows:
            f.write("\t".join(str(x) for x in r) + "\n")
    return ran


def intact_domain_names(output_table, protein_lib):
    """the decision of the domain recall (Util.py:8221-8234): TEs with a protein hit spanning >= 95 % of the protein"""
    _pn, proteins = read_fasta(protein_lib)
    keep = []
    with open(output_table) as f:
        for i, line in enumerate(f):
            if i < 2:
                continue
            parts = line.split("\t")
            if len(parts) < 6:
                continue
            if float(abs(int(parts[5]) - int(parts[4]))) / len(proteins[parts[1]]) >= 0.95 and parts[0] not in keep:
                keep.append(parts[0])
    return keep


_PROTEIN_LIB = {"tir": "TIRPeps.lib", "helitron": "HelitronPeps.lib", "non_ltr": "non_LTR.lib"}


def bucket_results(TE_type, results):
    """the collection loop of flank_region_align_v5 (Util.py:8159-8194, 8282-8287) on (cur_name, cur_seq, info, copy_count)
    tuples (cur_name None = not a TE): TIR / Helitron / non-LTR consensi that start with TG and end with CA are dropped (LTR
    ends), those with copy_count <= 5 (TIR, non-LTR) or <= 2 (Helitron) go to the low-copy bucket, the rest are real TEs.
    -> (true_tes, low_copy) in result order (the recall of low-copy elements follows in rescue_low_copy)."""
    true_tes, low_copy = {}, {}
    thr = 5 if TE_type in ("tir", "non_ltr") else 2
    for cur_name, cur_seq, _info, copy_count in results:
        if cur_name is None:
            continue
        if TE_type in ("tir", "helitron", "non_ltr"):
            if cur_seq.startswith("TG") and cur_seq.endswith("CA"):
                continue
            if copy_count <= thr:
                low_copy[cur_name] = cur_seq
            else:
                true_tes[cur_name] = cur_seq
        else:
            true_tes[cur_name] = cur_seq
    return true_tes, low_copy


def valid_filename(query_name):
    """Util.py:8127"""
    return re.sub(r'[<>:"/\\|?*]', "-", query_name)
