"""TEST INFRASTRUCTURE: the fine stage chained on the CPU from oracle pieces, following
flank_region_align_v5 / run_find_members_v8 / is_TE_from_align_file (Util.py:8095-8147,
10407-10449).  Used by the parity tests and by bench.py's cpu_baseline leg."""
import numpy as np

import oracle_lib as O

MAXROWS = 100
ROW_PAD = "."      # HITE_ROW_PAD (include/hite_gpu.h)


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


def judge_windows(te_type, cand, windows, plant, names=None, keep_msa=None, lens=None):
    """lens: the lengths the row selection goes by when they are not the windows' own (padded rows: the genome window's length)"""
    keep = select_rows(lens if lens is not None else [len(w) for w in windows], names)
    wins = [windows[i] for i in keep]
    m = O.star_msa(wins)
    if m is None:
        return ["EXC", 0], (-1, -1)
    kc = O.sparse_cols(m).astype(bool)
    clean = np.ascontiguousarray(m[:, kc])
    if keep_msa is not None:
        keep_msa.append(clean)
    res, b = O.judge(te_type, clean, cand, plant)
    return res, b


def anchor_class(cand, msa):
    """How the two 20-base anchors of judge_boundary_v5 / v9 (Util.py:9158-9181) sit in the first row that carries both:
    'none' (no row has both), 'exact' (both found without an edit), 'interior' (edits, but the first and the last base of each
    match agree with the pattern), 'end' (an edit on the first or last base of a match: the class where the real
    fuzzysearch package may report another start / end than the definition in oracle/stubs.py -- SURVEY.md 8c)."""
    import ctypes as C

    cb = cand.encode() if isinstance(cand, str) else bytes(cand)
    p1, p2 = cb[:20], cb[-20:]
    L = O.lib()
    out = (C.c_int * 4)()
    for r in range(msa.shape[0]):
        ung = msa[r][msa[r] != 45].tobytes()
        t = np.frombuffer(ung, dtype=np.uint8)
        g1 = L.orc_find_near_matches(O._ptr(O._u8(p1), O.u8p), len(p1), O._ptr(t, O.u8p), len(ung), 2, out)
        a = (out[0], out[1])
        if g1 <= 0:
            continue
        g2 = L.orc_find_near_matches(O._ptr(O._u8(p2), O.u8p), len(p2), O._ptr(t, O.u8p), len(ung), 2, out)
        b = (out[2], out[3])
        if g2 <= 0:
            continue
        ma, mb = ung[a[0]:a[1]], ung[b[0]:b[1]]
        if ma == p1 and mb == p2:
            return "exact"
        if ma[:1] != p1[:1] or ma[-1:] != p1[-1:] or mb[:1] != p2[:1] or mb[-1:] != p2[-1:]:
            return "end"
        return "interior"
    return "none"


_COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def _interval(contig, s, e, minus):
    """the bases of a record's interval (1-based, inclusive, clamped to the contig) read on the record's strand; not A / C / G / T -> N"""
    cb = contig.encode() if isinstance(contig, str) else bytes(contig)
    iv = bytes(c if c in b"ACGT" else 78 for c in cb[max(0, s - 1):max(0, min(len(cb), e))].upper())
    return iv.translate(_COMP)[::-1] if minus else iv


def _front_back(clip, minus):
    """clip word (left | right << 16 in the orientation of the genome) -> (in front of, behind) the window as it is read (a minus
    copy's window is reverse-complemented)"""
    a, b = clip & 0xffff, clip >> 16
    return (b, a) if minus else (a, b)


def _padded(w, clip, minus, centre, centre_clip=0, centre_minus=False):
    """the window of a copy record in the reference's coordinates, padded by the candidate bases its record leaves out LESS what the
    centre's own record leaves out on that side (pad_lengths, hite_pipeline.hip): pad byte i in front = the CENTRE's base i, pad byte j
    of the b behind = the centre's base m - b + j, in lower case (HITE_IS_ROW_PAD: they match the centre positions they face and leave
    the alignment as gaps of the row); ROW_PAD where the centre has no such position"""
    a, b = _front_back(clip, minus)
    a0, b0 = _front_back(centre_clip, centre_minus)
    a, b = max(0, a - a0), max(0, b - b0)
    m = len(centre)
    front = "".join(centre[i].lower() if i < m else ROW_PAD for i in range(a))
    back = "".join(centre[m - b + j].lower() if 0 <= m - b + j < m else ROW_PAD for j in range(b))
    return front + w + back


def fine_stage_candidate(te_type, cand, copies, contigs, plant=1, flank=50, contig_names=None, keep_msa=None):
    """copies: (contig_index, start1, end1, minus[, anchors, clip]) -> [is_TE, info, cons, row_num]; keep_msa: a list that receives
    the cleaned alignment of every pass that was judged.  clip (find_copies(..., clips=True); non-zero only for records in the
    reference's coordinates): hite_flank_region_align_clip -- the rows (never the centre, the first row kept) are padded by the clipped
    bases (_padded); the rows are chosen by the length of the genome window; the first500 + last500 form is cut from the padded window.
    A tuple WITHOUT the clip field (the reference's own 5-tuple): the clip word is estimated as hite_flank_region_align does (O.clip_probe)"""
    full, trunc = [], []            # (window, name, clip, minus) per pass, in input order
    for cp in copies:
        (ci, s, e, mn) = cp[:4]
        w, t = O.flank_window(contigs[ci], s, e, "-" if mn else "+", flank)
        if w is None:
            continue
        if len(cp) > 5:
            clip = int(cp[5])
        else:
            # a record without a clip word (the reference's own tuples): estimated from the sequences, hite_flank_region_align's rule
            pr = O.clip_probe(cand, _interval(contigs[ci], s, e, mn))
            clip = ((pr >> 16) | ((pr & 0xffff) << 16)) if mn else pr      # (kept in the orientation of the genome)
        rec = (w, window_name(cp, contig_names), clip, bool(mn))
        full.append(rec)
        if t is not None:
            trunc.append(rec)
    if not full:
        return [False, "", "", 0]

    def run(recs, cut):
        lens = [1000 if cut else len(r[0]) for r in recs]
        names = [r[1] for r in recs]
        if not any(r[2] for r in recs):
            wins = [r[0][:500] + r[0][-500:] if cut else r[0] for r in recs]
            return judge_windows(te_type, cand, wins, plant, names, keep_msa, lens=lens)
        keep = select_rows(lens, names)
        centre, _n0, clip0, mn0 = recs[keep[0]]
        wins = []
        for k, i in enumerate(keep):
            w, _nm, clip, mn = recs[i]
            if k > 0 and clip:
                w = _padded(w, clip, mn, centre, clip0, mn0)
            wins.append(w[:500] + w[-500:] if cut else w)
        return judge_windows(te_type, cand, wins, plant, None, keep_msa)       # (the rows are chosen: <= 100 windows)

    if trunc:
        res, _ = run(trunc, True)
        if res[0] == "EXC" or not res[0]:
            return res if res[0] != "EXC" else [False, "EXC", "", 0]
    res, _ = run(full, False)
    if res[0] == "EXC":
        return [False, "EXC", "", 0]
    return res
