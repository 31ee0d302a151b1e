"""Shared plumbing of the drop-in stage scripts (judge_TIR / judge_Helitron / judge_Non_LTR_transposons.py):
copy finder on the resident genome, cd-hit-est if installed, the refinement loop around flank_region_align_v5,
the final rename / prefix / prev_TE update (rename_fasta Util.py:7500, lib_add_prefix :11559, update_prev_TE)."""
import os
import shutil
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from hite_amd import util  # noqa: E402


def run_cd_hit(inp, outp, threads):
    """cd-hit-est -aS .95 -aL .95 -c .8 -G 0 -g 1 -A 80 (judge_TIR_transposons.py:87); when it is not installed the build's own
    stand-in removes the redundant sequences (util.remove_redundant_sequences) -- the step is never skipped"""
    if shutil.which("cd-hit-est") is None:
        util.remove_redundant_sequences(inp, outp, 0.95, 0.95)
        return
    subprocess.run("cd-hit-est -aS 0.95 -aL 0.95 -c 0.8 -G 0 -g 1 -A 80 -i %s -o %s -T %d -M 0 > /dev/null 2>&1" % (inp, outp, threads),
                   shell=True, check=False)


def refine(te_type, first_input, out_dir, stem, ref_index, reference, split_ref_dir, threads, plant, debug, low_copy, iters, flank=50):
    """iters rounds of flank_region_align_v5, each fed with the consensus of the previous one -> path of the last output"""
    cur = first_input
    for it in range(iters):
        nxt = os.path.join(out_dir, "%s_%s.r%d.fa" % (stem, ref_index, it))
        util.flank_region_align_v5(cur, nxt, flank, reference, split_ref_dir, te_type, out_dir, threads, ref_index, None, "", plant, debug,
                                   it, low_copy)
        cur = nxt
    return cur


def finish(result_path, final_path, label, ref_index, reference, min_len, prev_TE):
    """length filter, rename_fasta to <label>_<i>_<n> (a '#class' suffix survives, Util.py:7500), lib_add_prefix with the genome
    prefix (:11559), atomic publish (success == the file exists, :2831), update_prev_TE under its lock (:6378) -- the tail of
    judge_TIR / judge_Helitron / judge_Non_LTR_transposons.py"""
    names, contigs = util.read_fasta(result_path)
    tmp = final_path + ".tmp"
    util.store_fasta({n: contigs[n] for n in names if len(contigs[n]) >= min_len}, tmp + ".len")
    util.rename_fasta(tmp + ".len", tmp, "%s_%s" % (label, ref_index))
    os.remove(tmp + ".len")
    util.lib_add_prefix(tmp, os.path.basename(reference).split(".")[0])
    os.replace(tmp, final_path)
    if prev_TE:
        util.update_prev_TE(prev_TE, final_path)
    return util.read_fasta(final_path)[1]
