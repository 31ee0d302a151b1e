"""TEST INFRASTRUCTURE: the fine stage chained on the CPU from oracle pieces, following
flank_region_align_v5 / run_find_members_v8 / is_TE_from_align_file (Util.py:8095-8147,
10407-10449).  Used by the parity tests and by bench.py's cpu_baseline leg."""
import numpy as np

import oracle_lib as O

MAXROWS = 100


def window_name(copy, contig_names=None):
    """the FASTA name of a copy's window (Util.py:8110); without contig names: the index, zero-padded (byte order = index order)"""
    ci, s, e, mn = copy[:4]
    return "%s:%d-%d(%s)" % (contig_names[ci] if contig_names is not None else "%09d" % ci, s, e, "-" if mn else "+")


def select_rows(lens, names=None):
    """indices of the windows kept for alignment: all if <= 100, else the 100 longest -- `sort -nk 2 -r` on the .fai leaves equal
    lengths in REVERSE byte order of the line, i.e. of the window name (POSIX locale; pinned by tests/golden/ready_for_msa.json.gz,
    the reference's tools/ready_for_MSA.sh run by oracle/gen_golden.py) -- in input order (grep keeps the file's order).
    names = None: ties in input order (callers without names)."""
    idx = list(range(len(lens)))
    if len(idx) <= MAXROWS:
        return idx
    if names is None:
        order = sorted(idx, key=lambda i: (-lens[i], i))[:MAXROWS]
    else:
        # descending name; the same name twice (the same interval twice): input order
        pos = {i: r for r, i in enumerate(sorted(idx, key=lambda i: (names[i].encode(), -i), reverse=True))}
        order = sorted(idx, key=lambda i: (-lens[i], pos[i]))[:MAXROWS]
    return sorted(order)


def judge_windows(te_type, cand, windows, plant, names=None):
    keep = select_rows([len(w) for w in windows], names)
    wins = [windows[i] for i in keep]
    m = O.star_msa(wins)
    if m is None:
        return ["EXC", 0], (-1, -1)
    kc = O.sparse_cols(m).astype(bool)
    res, b = O.judge(te_type, np.ascontiguousarray(m[:, kc]), cand, plant)
    return res, b


def fine_stage_candidate(te_type, cand, copies, contigs, plant=1, flank=50, contig_names=None):
    """copies: (contig_index, start1, end1, minus) -> [is_TE, info, cons, row_num]"""
    full, trunc, fn, tn = [], [], [], []
    for cp in copies:
        (ci, s, e, mn) = cp[:4]
        w, t = O.flank_window(contigs[ci], s, e, "-" if mn else "+", flank)
        if w is None:
            continue
        full.append(w)
        fn.append(window_name(cp, contig_names))
        if t is not None:
            trunc.append(t)
            tn.append(fn[-1])
    if not full:
        return [False, "", "", 0]
    if trunc:
        res, _ = judge_windows(te_type, cand, trunc, plant, tn)
        if res[0] == "EXC" or not res[0]:
            return res if res[0] != "EXC" else [False, "EXC", "", 0]
    res, _ = judge_windows(te_type, cand, full, plant, fn)
    if res[0] == "EXC":
        return [False, "EXC", "", 0]
    return res
