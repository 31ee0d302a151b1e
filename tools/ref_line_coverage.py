#!/usr/bin/env python
"""TEST INFRASTRUCTURE, build container only (needs /root/reference): which lines of the reference's functions on the path do the
golden generators actually execute?  Runs oracle/gen_golden.py's generators with the fixture writer switched off and a line tracer
on the reference's module/Util.py (and FiLTR's src/Util.py), then prints executed / executable lines per function and the line
numbers never reached -- the places where "the oracle equals the reference on the goldens" says nothing yet.

    PYTHONHASHSEED=0 python tools/ref_line_coverage.py [generator names ...] > profiles/rNN_reference_line_coverage.txt
"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_golden  # noqa: E402
import ref_harness  # noqa: E402

# the functions of SURVEY.md section 8(a) that are HiTE's own Python (third-party tools are not traceable)
WATCH = ["get_longest_repeats_v4", "FMEA", "get_full_length_copies_from_blastn_v1", "generate_full_length_out_v1", "flanking_seq",
         "search_confident_tir_v4", "search_confident_tir_batch_v1", "get_query_copies", "get_copies_v1", "remove_sparse_col_in_align_file",
         "judge_boundary_v5", "judge_boundary_v6", "judge_boundary_v9", "search_boundary_homo_v3", "search_boundary_homo_v4",
         "get_boundary_ungap_str", "TSDsearch_v5", "is_TE_from_align_file", "generate_cons_v1", "split_and_store_sequences",
         "get_short_tir_contigs", "filter_dup_itr_v3", "FMEA_new1_parallel_large", "process_blast_results_in_chunks",
         "multiple_alignment_blast_and_get_copies_v1", "get_domain_info", "judge_both_ends_frame_v1", "judge_left_frame_LTR",
         "judge_right_frame_LTR", "filter_ltr_by_flank_seq_v2", "get_non_empty_seq", "map_fragment",
         # the pieces the generators call directly (f-1 ... f-4, a-22 and the library merge)
         "flank_region_align_v5", "calculate_window_homology", "search_polyA_TSD", "find_tail_polyA", "find_longest_tandem_repeat_tail",
         "process_chunk", "extend_fragments", "cluster_sequences_from_chunks", "cons_from_mafft_v1", "save_data_in_chunks", "rename_fasta",
         "rename_reference", "lib_add_prefix", "file_exist", "update_prev_TE", "getReverseSequence", "get_both_ends_frame",
         "most_common_element", "read_Ninja_clusters", "generate_both_ends_frame_for_intactLTR", "get_LTR_seq_from_scn"]


def code_lines(code):
    out = set()
    for _s, _e, ln in code.co_lines():
        if ln is not None:
            out.add(ln)
    for c in code.co_consts:
        if hasattr(c, "co_lines"):
            out |= code_lines(c)
    out.discard(code.co_firstlineno)        # the def line itself runs at import
    return out


def main():
    assert os.environ.get("PYTHONHASHSEED") == "0", "run with PYTHONHASHSEED=0 (the generators' inputs are then the fixtures')"
    U = ref_harness.load_reference_util()
    files = {os.path.realpath(U.__file__)}
    hit = {}

    def local(frame, event, arg):
        if event == "line":
            hit.setdefault(frame.f_code.co_filename, set()).add(frame.f_lineno)
        return local

    def tracer(frame, event, arg):
        if event == "call" and os.path.realpath(frame.f_code.co_filename) in files:
            return local
        return None

    gen_golden.dump = lambda name, obj: None          # fixtures stay as they are
    try:
        F = ref_harness.load_filtr_util()
        files.add(os.path.realpath(F.__file__))
    except Exception:
        F = None
    sys.argv = [sys.argv[0]] + sys.argv[1:]
    threading.settrace(tracer)
    sys.settrace(tracer)
    try:
        gen_golden.main()
    finally:
        sys.settrace(None)
        threading.settrace(None)
    allhit = set()
    for fn, s in hit.items():
        if os.path.realpath(fn) == os.path.realpath(U.__file__):
            allhit |= s
    fhit = set()
    if F is not None:
        for fn, s in hit.items():
            if os.path.realpath(fn) == os.path.realpath(F.__file__):
                fhit |= s
    print("# lines of the reference's own functions executed while the golden fixtures are generated (tools/ref_line_coverage.py)")
    print("# function (file:first line)                       executed / executable    lines never reached")
    tot_e = tot_x = 0
    for mod, hs, tag in ((U, allhit, "module/Util.py"), (F, fhit, "FiLTR src/Util.py")):
        if mod is None:
            continue
        for name in WATCH:
            f = getattr(mod, name, None)
            if f is None or not hasattr(f, "__code__"):
                continue
            lines = code_lines(f.__code__)
            ex = lines & hs
            if not ex:
                continue                              # not a function the fixtures go through in this module
            miss = sorted(lines - hs)
            tot_e += len(ex); tot_x += len(lines)
            print("%-52s %4d / %4d  %5.1f %%   %s" % ("%s (%s:%d)" % (name, tag, f.__code__.co_firstlineno), len(ex), len(lines),
                                                     100.0 * len(ex) / max(1, len(lines)), " ".join(map(str, miss)) if miss else "-"))
    print("# total %d / %d = %.1f %%" % (tot_e, tot_x, 100.0 * tot_e / max(1, tot_x)))
    never = [n for n in WATCH if not any(hasattr(getattr(mod, n, None), "__code__") and (code_lines(getattr(mod, n).__code__) & hs)
                                         for mod, hs in ((U, allhit), (F, fhit)) if mod is not None)]
    print("# watched but never entered by a generator (wrappers around the above, or run in worker processes the tracer does not follow): "
          + ", ".join(never))


if __name__ == "__main__":
    main()
