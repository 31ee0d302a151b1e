"""TEST INFRASTRUCTURE: the fine stage chained on the CPU from oracle pieces, following
flank_region_align_v5 / run_find_members_v8 / is_TE_from_align_file (Util.py:8095-8147,
10407-10449).  Used by the parity tests and by bench.py's cpu_baseline leg."""
import numpy as np

import oracle_lib as O

MAXROWS = 100


def select_rows(lens):
    """indices of the windows kept for alignment: all if <= 100, else the 100 longest
    (ties: input order), in input order (tools/ready_for_MSA.sh <f> 100 100)."""
    idx = list(range(len(lens)))
    if len(idx) <= MAXROWS:
        return idx
    order = sorted(idx, key=lambda i: (-lens[i], i))[:MAXROWS]
    return sorted(order)


def judge_windows(te_type, cand, windows, plant):
    keep = select_rows([len(w) for w in windows])
    wins = [windows[i] for i in keep]
    m = O.star_msa(wins)
    if m is None:
        return ["EXC", 0], (-1, -1)
    kc = O.sparse_cols(m).astype(bool)
    res, b = O.judge(te_type, np.ascontiguousarray(m[:, kc]), cand, plant)
    return res, b


def fine_stage_candidate(te_type, cand, copies, contigs, plant=1, flank=50):
    """copies: (contig_index, start1, end1, minus) -> [is_TE, info, cons, row_num]"""
    full, trunc = [], []
    for (ci, s, e, mn) in copies:
        w, t = O.flank_window(contigs[ci], s, e, "-" if mn else "+", flank)
        if w is None:
            continue
        full.append(w)
        if t is not None:
            trunc.append(t)
    if not full:
        return [False, "", "", 0]
    if trunc:
        res, _ = judge_windows(te_type, cand, trunc, plant)
        if res[0] == "EXC" or not res[0]:
            return res if res[0] != "EXC" else [False, "EXC", "", 0]
    res, _ = judge_windows(te_type, cand, full, plant)
    if res[0] == "EXC":
        return [False, "EXC", "", 0]
    return res
