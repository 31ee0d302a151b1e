"""Seeded synthetic case generators shared by oracle/gen_golden.py (which feeds them to
the reference's Python) and by the parity tests (which feed the same generators, at
other seeds/sizes, to the C oracle and the HIP path).

Pure numpy/python; no reference code, no oracle code.
"""
import numpy as np

BASES = "ACGT"
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def revcomp(s):
    return "".join(_COMP.get(c, "N") for c in reversed(s))


def rand_seq(rng, n):
    return "".join(BASES[i] for i in rng.integers(0, 4, size=n))


def mutate(rng, s, sub_rate):
    out = list(s)
    for i in range(len(out)):
        if rng.random() < sub_rate:
            out[i] = BASES[(BASES.index(out[i]) + 1 + int(rng.integers(0, 3))) % 4] if out[i] in BASES else "A"
    return "".join(out)


# ----------------------------------------------------------------------------------
# multiple-alignment matrices (input of remove_sparse_col / judge_boundary_v5/v6/v9)
# ----------------------------------------------------------------------------------
def make_msa_case(seed, te_type="tir", rows=12, te_len=200, flank=50, div=0.08,
                  row_gap_rate=0.01, ins_cols=3, trunc_rows=1, shift_l=0, shift_r=0,
                  tsd_len=8, tsd_frac=0.8, noise_rows=0, homolog_flank_l=0,
                  homolog_flank_r=0, long_gap_rows=0, start_motif=None, end_motif=None):
    """Build a gapped rows x cols alignment of `rows` copies of one synthetic TE family
    with +-`flank` bp of (mostly) non-homologous flanks, plus the candidate sequence
    (row 0's element, boundaries shifted by shift_l/shift_r: positive = candidate longer).

    Returns dict(names, seqs (gapped, equal length), cand, te_type, plant).
    """
    rng = np.random.default_rng(seed)
    cons = rand_seq(rng, te_len)
    if te_type == "tir":
        tir = rand_seq(rng, 12)
        # avoid TG..CA and TATATATA starts
        if tir.startswith("TG"):
            tir = "CA" + tir[2:]
        cons = tir + cons[12:-12] + revcomp(tir)
    elif te_type == "helitron":
        cons = "TC" + cons[2:-4] + str(rng.choice(["CTAG", "CTAA", "CTGG", "CTGA"]))
    elif te_type == "non_ltr":
        cons = cons[:-14] + "A" * 14
    if start_motif:
        cons = start_motif + cons[len(start_motif):]
    if end_motif:
        cons = cons[:-len(end_motif)] + end_motif
    L = len(cons)
    hom_l = rand_seq(rng, homolog_flank_l)
    hom_r = rand_seq(rng, homolog_flank_r)
    # per-row ungapped pieces: left flank, element (with deletions marked), right flank
    row_left, row_te, row_right = [], [], []
    for r in range(rows):
        te = mutate(rng, cons, div if r > 0 else 0.0)
        lf = rand_seq(rng, flank)
        rf = rand_seq(rng, flank)
        if homolog_flank_l:
            lf = lf[: flank - homolog_flank_l] + mutate(rng, hom_l, div)
        if homolog_flank_r:
            rf = mutate(rng, hom_r, div) + rf[homolog_flank_r:]
        if te_type == "tir" and rng.random() < tsd_frac and tsd_len > 0:
            if tsd_len == 2:
                tsd = "TA"
            elif tsd_len == 4:
                tsd = "TTAA"
            elif tsd_len == 3:
                tsd = "TAA"
            else:
                tsd = rand_seq(rng, tsd_len)
            lf = lf[: flank - tsd_len] + tsd
            rf = tsd + rf[tsd_len:]
        elif te_type == "helitron":
            lf = lf[:-1] + ("A" if rng.random() < 0.9 else "C")
            rf = ("T" if rng.random() < 0.9 else "G") + rf[1:]
        elif te_type == "non_ltr" and rng.random() < tsd_frac:
            k = int(rng.integers(8, 21))
            tsd = rand_seq(rng, k)
            while tsd.endswith("A") or tsd.startswith("A"):
                tsd = rand_seq(rng, k)
            lf = lf[: flank - k] + tsd
            rf = tsd + rf[k:]
        row_left.append(lf)
        row_te.append(te)
        row_right.append(rf)
    # column model: every ungapped position of [left|te|right] is a column; add
    # insertion columns (bases present in a single row) and per-row deletions.
    total = flank + L + flank
    ins_after = sorted(set(int(x) for x in rng.integers(5, total - 5, size=ins_cols))) if ins_cols else []
    mat = []
    for r in range(rows):
        s = row_left[r] + row_te[r] + row_right[r]
        s = list(s)
        # deletions inside the element
        i = flank + 25
        while i < flank + L - 25:
            if r > 0 and rng.random() < row_gap_rate:
                glen = int(rng.integers(1, 6))
                for j in range(i, min(i + glen, flank + L - 25)):
                    s[j] = "-"
                i += glen
            i += 1
        mat.append(s)
    # truncated rows: leading part missing (gaps) through the left boundary
    for t in range(trunc_rows):
        r = rows - 1 - t
        if r <= 0:
            break
        cut = flank + int(rng.integers(15, max(16, L // 3)))
        for j in range(cut):
            mat[r][j] = "-"
    for t in range(long_gap_rows):
        r = 1 + t
        if r >= rows:
            break
        a = flank + L // 3
        b = flank + 2 * L // 3
        for j in range(a, b):
            mat[r][j] = "-"
    # insertion columns
    for pos in reversed(ins_after):
        owner = int(rng.integers(0, rows))
        ilen = int(rng.integers(1, 4))
        for r in range(rows):
            ins = list(rand_seq(rng, ilen)) if r == owner else ["-"] * ilen
            mat[r][pos:pos] = ins
    seqs = ["".join(m) for m in mat]
    # noise rows: unrelated sequence, same width
    width = len(seqs[0])
    for t in range(noise_rows):
        seqs.append(rand_seq(rng, width))
    names = ["chr%d:%d-%d(%s)" % (i % 5, 1000 + 37 * i, 1000 + 37 * i + L - 1, "+-"[i % 2])
             for i in range(len(seqs))]
    # candidate = row-0 element with shifted boundaries (ungapped coordinates of row 0)
    row0 = row_left[0] + row_te[0] + row_right[0]
    cs = max(0, flank - shift_l)
    ce = min(len(row0), flank + L + shift_r)
    cand = row0[cs:ce]
    return {"names": names, "seqs": seqs, "cand": cand, "te_type": te_type, "plant": 1,
            "true_start_col_ungapped": flank, "true_len": L}


def msa_param_grid(te_type, n, seed0):
    """n varied parameter sets for make_msa_case (deterministic)."""
    rng = np.random.default_rng(seed0)
    out = []
    for i in range(n):
        rows = int(rng.choice([2, 3, 5, 6, 8, 12, 20, 30, 60, 104]))
        p = dict(seed=seed0 * 1000 + i, te_type=te_type, rows=rows,
                 te_len=int(rng.choice([90, 120, 200, 350, 600, 900])),
                 div=float(rng.choice([0.0, 0.03, 0.08, 0.15, 0.25])),
                 row_gap_rate=float(rng.choice([0.0, 0.005, 0.02])),
                 ins_cols=int(rng.choice([0, 2, 6])),
                 trunc_rows=int(rng.choice([0, 1, 3])) if rows > 4 else 0,
                 shift_l=int(rng.choice([0, 0, 7, -5, 20, -12, 3])),
                 shift_r=int(rng.choice([0, 0, 7, -5, 20, -12, 3])),
                 tsd_len=int(rng.choice([0, 2, 3, 4, 5, 8, 9, 11])),
                 tsd_frac=float(rng.choice([0.0, 0.5, 1.0])),
                 noise_rows=int(rng.choice([0, 0, 1, 4])),
                 homolog_flank_l=int(rng.choice([0, 0, 0, 15, 45])),
                 homolog_flank_r=int(rng.choice([0, 0, 0, 15, 45])),
                 long_gap_rows=int(rng.choice([0, 0, 2])))
        out.append(p)
    return out


# ----------------------------------------------------------------------------------
# blast6-style HSP tables (input of get_longest_repeats_v4)
# ----------------------------------------------------------------------------------
def make_hsp_table(seed, n_seg=4, n_fam=6, copies=(2, 9), seg_len=1_000_000, noise=20,
                   frag=(1, 4), dup=0, chroms=("chr1", "chr2")):
    """HSP tuples (qname, sname, qs, qe, ss, se) resembling blastn -outfmt 6 between
    1 Mbp genome segments named chr$offset, 1-based inclusive, reverse hits have ss>se."""
    rng = np.random.default_rng(seed)
    segs = []
    for c in chroms:
        for k in range(n_seg):
            segs.append((c, k * seg_len))
    copies_by_fam = []
    for f in range(n_fam):
        flen = int(rng.integers(150, 9000))
        nc = int(rng.integers(copies[0], copies[1] + 1))
        cl = []
        for _ in range(nc):
            sg = segs[int(rng.integers(0, len(segs)))]
            pos = int(rng.integers(1000, seg_len - flen - 1000))
            strand = int(rng.integers(0, 2))
            cl.append((sg, pos, strand))
        copies_by_fam.append((flen, cl))
    rows = []
    for flen, cl in copies_by_fam:
        for a in range(len(cl)):
            for b in range(len(cl)):
                if a == b:
                    continue
                (sa, pa, sta), (sb, pb, stb) = cl[a], cl[b]
                nfr = int(rng.integers(frag[0], frag[1] + 1))
                cuts = sorted(set([0, flen] + [int(x) for x in rng.integers(30, flen - 30, size=nfr - 1)])) if flen > 80 else [0, flen]
                for i in range(len(cuts) - 1):
                    x0, x1 = cuts[i], cuts[i + 1]
                    g0 = int(rng.integers(0, 12))
                    g1 = int(rng.integers(0, 12))
                    if x1 - g1 - (x0 + g0) < 20:
                        continue
                    # copy A forward coordinates of the fragment
                    if sta == 0:
                        qs, qe = pa + x0 + g0, pa + x1 - g1 - 1
                    else:
                        qs, qe = pa + (flen - x1) + g1, pa + (flen - x0) - g0 - 1
                    if stb == 0:
                        ts, te = pb + x0 + g0, pb + x1 - g1 - 1
                    else:
                        ts, te = pb + (flen - x1) + g1, pb + (flen - x0) - g0 - 1
                    jitter = int(rng.integers(-3, 4))
                    ts += jitter
                    te += jitter
                    if sta != stb:
                        ss, se = te, ts
                    else:
                        ss, se = ts, te
                    rows.append(("%s$%d" % sa, "%s$%d" % sb, qs, qe, ss, se))
    for _ in range(noise):
        sa = segs[int(rng.integers(0, len(segs)))]
        sb = segs[int(rng.integers(0, len(segs)))]
        ln = int(rng.integers(30, 400))
        qs = int(rng.integers(1, seg_len - ln))
        ss = int(rng.integers(1, seg_len - ln))
        if rng.random() < 0.5:
            rows.append(("%s$%d" % sa, "%s$%d" % sb, qs, qs + ln, ss, ss + ln + int(rng.integers(-2, 3))))
        else:
            rows.append(("%s$%d" % sa, "%s$%d" % sb, qs, qs + ln, ss + ln, ss))
    # exact self hits (skipped by the reference) and duplicates
    for sg in segs[:2]:
        rows.append(("%s$%d" % sg, "%s$%d" % sg, 1, seg_len, 1, seg_len))
    for _ in range(dup):
        rows.append(rows[int(rng.integers(0, len(rows)))])
    perm = rng.permutation(len(rows))
    return [rows[i] for i in perm]


def hsp_to_blast6_lines(rows):
    return ["%s\t%s\t%.3f\t%d\t0\t0\t%d\t%d\t%d\t%d\t1e-50\t%.1f\n" %
            (q, s, 95.0, abs(qe - qs) + 1, qs, qe, ss, se, 2.0 * (abs(qe - qs) + 1))
            for (q, s, qs, qe, ss, se) in rows]


# ----------------------------------------------------------------------------------
# flanked candidates (input of search_confident_tir_v4) and genomes/copies (gather)
# ----------------------------------------------------------------------------------
def make_tir_candidate(seed, te_len=300, flank=50, tsd_len=8, off_l=0, off_r=0, with_n=False):
    rng = np.random.default_rng(seed)
    tsd = {2: "TA", 4: "TTAA"}.get(tsd_len, rand_seq(rng, tsd_len))
    tir = rand_seq(rng, 15)
    te = tir + rand_seq(rng, te_len - 30) + revcomp(tir)
    left = rand_seq(rng, flank + 30)
    right = rand_seq(rng, flank + 30)
    seq = left + tsd + te + tsd + right
    if with_n:
        seq = seq[:20] + "NNNN" + seq[24:]
    true_start = len(left) + len(tsd)  # 0-based first base of the element
    true_end = true_start + te_len - 1
    # the flanked candidate is cut so that the raw boundaries sit `flank` from each end
    cs = true_start + off_l - flank
    ce = true_end + off_r + flank + 1
    cs = max(0, cs)
    ce = min(len(seq), ce)
    return seq[cs:ce], flank


def make_genome(seed, n_chr=3, chr_len=(20000, 60000), n_frac=0.002, other_frac=0.0005):
    rng = np.random.default_rng(seed)
    names, seqs = [], []
    for i in range(n_chr):
        n = int(rng.integers(chr_len[0], chr_len[1]))
        a = rng.integers(0, 4, size=n)
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[a].copy()
        # runs of N and a few IUPAC codes
        for _ in range(max(1, int(n * n_frac / 20))):
            p = int(rng.integers(0, n - 40))
            s[p:p + int(rng.integers(1, 40))] = ord("N")
        for _ in range(int(n * other_frac)):
            s[int(rng.integers(0, n))] = ord("RYKM"[int(rng.integers(0, 4))])
        names.append("chr%d" % (i + 1))
        seqs.append(s.tobytes().decode())
    return names, seqs


def make_copies(seed, names, seqs, n_cand=6, per_cand=(1, 12), length=(60, 2500), flank=50):
    """{query: [(chr, start1, end1, aln_len, strand)]} as get_copies_minimap2 returns them
    (1-based inclusive), including copies that run off the contig ends."""
    rng = np.random.default_rng(seed)
    out = {}
    for q in range(n_cand):
        lst = []
        for _ in range(int(rng.integers(per_cand[0], per_cand[1] + 1))):
            ci = int(rng.integers(0, len(names)))
            L = int(rng.integers(length[0], length[1]))
            n = len(seqs[ci])
            mode = rng.random()
            if mode < 0.08:
                st = int(rng.integers(1, flank + 2))
            elif mode < 0.16:
                st = n - L - int(rng.integers(0, flank + 1))
            else:
                st = int(rng.integers(1, max(2, n - L)))
            st = max(1, st)
            en = min(n, st + L - 1)
            lst.append((names[ci], st, en, en - st + 1, "+-"[int(rng.integers(0, 2))]))
        out["cand_%d" % q] = lst
    return out


# ----------------------------------------------------------------------------------
# alignment cases aimed at the rare outcomes of judge_boundary_v5/v6/v9: positives, 'nb' (no anchor),
# 'fl1' (one full-length row), and inputs on which the reference raises
# ----------------------------------------------------------------------------------
def msa_outcome_cases(te_type, seed0):
    """Deterministic list of cases (dicts as make_msa_case returns, `plant` set) whose outcomes cover
    every exit of the three judges: >= 12 positives (plant 0 and 1), >= 6 'nb', >= 6 'fl1'-shaped inputs
    (for Helitron, which has no such exit, they pin whatever v6 does with one start/end row) and
    degenerate inputs (empty candidate, one-column rows, truncated rows in front of the 100-row cap)."""
    rng = np.random.default_rng(seed0)
    out = []

    def base(i, **kw):
        p = dict(seed=seed0 * 100 + i, te_type=te_type, rows=int(rng.choice([3, 5, 8, 12, 20, 30, 64, 70])),
                 te_len=int(rng.choice([120, 200, 350, 700])), div=float(rng.choice([0.0, 0.03, 0.08])),
                 row_gap_rate=float(rng.choice([0.0, 0.005])), ins_cols=int(rng.choice([0, 2])), trunc_rows=0,
                 shift_l=0, shift_r=0, tsd_len=int(rng.choice([2, 3, 4, 5, 8, 9, 11])), tsd_frac=1.0)
        p.update(kw)
        return make_msa_case(**p)

    # positives, both values of `plant`
    for i in range(14):
        c = base(i)
        c["plant"] = i % 2
        c["aim"] = "positive"
        out.append(c)
    # 'nb': the candidate is not in any row (unrelated, or both ends rewritten)
    for i in range(6):
        c = base(20 + i)
        r2 = np.random.default_rng(seed0 * 7 + i)
        if i % 2 == 0:
            c["cand"] = rand_seq(r2, len(c["cand"]))
        else:
            c["cand"] = rand_seq(r2, 20) + c["cand"][20:-20] + rand_seq(r2, 20)
        c["plant"] = i % 2
        c["aim"] = "nb"
        out.append(c)
    # 'fl1': only the first row has bases at both anchors (a single row; or the other rows lost one end each -- half of
    # them the left one, half the right one, so that no column becomes sparse)
    for i in range(6):
        rows = 1 if i < 2 else int(rng.choice([3, 7, 13]))
        c = base(30 + i, rows=rows, trunc_rows=0, ins_cols=0)
        if rows > 1:
            W = len(c["seqs"][0])
            cut = 50 + 15 + 3 * i
            seqs = [c["seqs"][0]]
            for r, s in enumerate(c["seqs"][1:]):
                seqs.append("-" * cut + s[cut:] if r % 2 == 0 else s[:W - cut] + "-" * cut)
            c["seqs"] = seqs
        c["plant"] = i % 2
        c["aim"] = "fl1"
        out.append(c)
    # degenerate inputs
    c = base(40)
    c["cand"] = ""
    c["aim"] = "empty candidate"
    out.append(c)
    c = base(41, rows=4, te_len=90)
    c["cand"] = c["cand"][:12]           # shorter than the 20-bp anchors: both anchors are the whole candidate
    c["aim"] = "short candidate"
    out.append(c)
    c = base(42, rows=3, te_len=90)
    c["seqs"] = [s[:1] for s in c["seqs"]]   # one-column alignment
    c["aim"] = "one column"
    out.append(c)
    # the 100-row cap of the row sets: 101 rows without a left end in front of 110 full-length rows (columns stay dense)
    def renamed(c):
        c["names"] = ["chr%d:%d-%d(%s)" % (i % 5, 1000 + 37 * i, 1000 + 37 * i + 199, "+-"[i % 2]) for i in range(len(c["seqs"]))]
        return c
    c = base(43, rows=110, te_len=200, ins_cols=0, div=0.03)
    c["seqs"] = ["-" * 90 + s[90:] for s in c["seqs"][1:102]] + c["seqs"]
    c["aim"] = "101 rows without a left end, then the full-length rows"
    out.append(renamed(c))
    c = base(44, rows=110, te_len=200, ins_cols=0, div=0.03)
    c["seqs"] = c["seqs"][:1] + ["-" * 90 + s[90:] for s in c["seqs"][1:102]] + c["seqs"][1:]
    c["aim"] = "anchor row first, then 101 rows without a left end"
    out.append(renamed(c))
    W = len(c["seqs"][0])
    c = base(45, rows=110, te_len=200, ins_cols=0, div=0.03)
    W = len(c["seqs"][0])
    c["seqs"] = [s[:W - 90] + "-" * 90 for s in c["seqs"][1:102]] + c["seqs"]
    c["aim"] = "101 rows without a right end, then the full-length rows"
    out.append(renamed(c))
    for c in out:
        c.setdefault("plant", 1)
    return out


def msa_edge_cases(seed0):
    """Round 4: cases aimed at the lines of judge_boundary_v5 / v6 / v9 and search_boundary_homo_v3 / v4 that the fixtures of rounds
    1-3 never reached (tools/ref_line_coverage.py): anchors within 10 columns of the alignment's edges, homology that runs to the
    edge, fewer than 10 homologous columns left for a window, a column of the element whose majority is '-' among the full-length
    rows, the TA / TAA / TTAA trims at both ends, a homology boundary outside the columns that are dense among the full-length
    rows.  Every case carries te_type, plant and `aim`."""
    out = []

    def add(aim, te_type, plant=1, **kw):
        c = make_msa_case(te_type=te_type, **kw)
        c["plant"] = plant
        c["aim"] = aim
        out.append(c)
        return c

    i = 0
    for te_type in ("tir", "non_ltr", "helitron"):
        tf = 1.0
        for flank in (3, 6, 9, 10, 11, 12, 20):
            for rows in (4, 12):
                i += 1
                # (the non-LTR target-site duplication, 8-20 bases, does not fit the shorter flanks)
                add("flank of %d columns" % flank, te_type, plant=i % 2, seed=seed0 * 100 + i, rows=rows, te_len=160, flank=flank, div=0.05,
                    row_gap_rate=0.0, ins_cols=0, trunc_rows=0, tsd_len=2, tsd_frac=0.0 if te_type == "non_ltr" and flank < 20 else 1.0)
        # no flank at all: the candidate's ends are the alignment's first and last column
        for rows in (3, 9):
            i += 1
            c = add("no flank", te_type, plant=i % 2, seed=seed0 * 100 + i, rows=rows, te_len=150, flank=4, div=0.04, row_gap_rate=0.0, ins_cols=0,
                    trunc_rows=0, tsd_len=2, tsd_frac=0.0)
            c["seqs"] = [x[4:-4] for x in c["seqs"]]
        for hl, hr in ((50, 0), (0, 50), (50, 50), (44, 0), (0, 44), (41, 47)):
            i += 1
            add("homologous flanks %d / %d of 50" % (hl, hr), te_type, plant=i % 2, seed=seed0 * 100 + i, rows=10, te_len=200, div=0.04,
                row_gap_rate=0.0, ins_cols=0, trunc_rows=0, tsd_len=4, tsd_frac=tf, homolog_flank_l=hl, homolog_flank_r=hr)
        for rows, tr, lg in ((12, 6, 4), (16, 6, 7), (9, 4, 3)):
            i += 1
            add("middle third of the element: '-' in most full-length rows", te_type, plant=1, seed=seed0 * 100 + i, rows=rows, te_len=240, div=0.04,
                row_gap_rate=0.0, ins_cols=0, trunc_rows=tr, tsd_len=8, tsd_frac=tf, long_gap_rows=lg)
        # leading / trailing flank columns that are dense in the alignment but '-' in most full-length rows, homology reaching into them
        for hl, cut, side in ((38, 16, "l"), (36, 20, "l"), (38, 16, "r"), (30, 26, "l")):
            i += 1
            c = add("homology starts inside flank columns that most full-length rows lack (%s)" % side, te_type, plant=1, seed=seed0 * 100 + i, rows=12,
                    te_len=200, div=0.04, row_gap_rate=0.0, ins_cols=0, trunc_rows=0, tsd_len=0, tsd_frac=0.0,
                    homolog_flank_l=hl if side == "l" else 0, homolog_flank_r=hl if side == "r" else 0)
            W = len(c["seqs"][0])
            seqs = list(c["seqs"])
            for r in range(1, 8):
                seqs[r] = "-" * cut + seqs[r][cut:] if side == "l" else seqs[r][:W - cut] + "-" * cut
            # six rows that lack the OTHER end (not full length) keep those columns dense in the alignment
            extra = [(s[:W - 90] + "-" * 90) if side == "l" else ("-" * 90 + s[90:]) for s in c["seqs"][1:7]]
            c["seqs"] = seqs + extra
            c["names"] = ["chr%d:%d-%d(%s)" % (k % 5, 1000 + 37 * k, 1000 + 37 * k + 199, "+-"[k % 2]) for k in range(len(c["seqs"]))]
    # the terminal trims of judge_boundary_v5: A / TA / AA / TAA / TTA / TTAA at the start, T / TT / TA / TAA / TTA / TTAA at the end
    for sm, em in (("TTAA", "TTAA"), ("TAA", "TTA"), ("TA", "TA"), ("AA", "TT"), ("A", "T"), ("TTA", "TAA"), (None, "TTAA"), ("TTAA", None)):
        for tl in (2, 3, 4, 8):
            i += 1
            add("ends %s ... %s" % (sm, em), "tir", plant=i % 2, seed=seed0 * 100 + i, rows=10, te_len=180, div=0.03, row_gap_rate=0.0, ins_cols=0,
                trunc_rows=0, tsd_len=tl, tsd_frac=1.0, start_motif=sm, end_motif=em)
    return out


# ----------------------------------------------------------------------------------
# sequences with planted tandem arrays (input of the tandem-repeat masker / of TRF)
# ----------------------------------------------------------------------------------
def make_tandem_case(seed, G=120_000, n_arr=70):
    """-> (sequence, [[start, end, period, copies, substitution rate per copy, indel rate]]): random background with tandem
    arrays of period 1 .. 500, 1.6 .. 25 copies, 0-15 % substitutions and 0-2 % indels per copy, 300-1500 bases apart"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    seq = rng.choice(acgt, size=G)
    pos, planted = 3000, []
    while pos < G - 14000 and len(planted) < n_arr:
        p = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 9, 12, 15, 21, 33, 48, 77, 120, 180, 260, 390, 500]))
        copies = float(rng.choice([1.6, 2.0, 2.5, 3, 4, 6, 10, 25]))
        div = float(rng.choice([0, 0, 0.03, 0.08, 0.15]))
        ind = float(rng.choice([0, 0, 0.005, 0.02]))
        L = max(int(p * copies), int(rng.integers(20, 70)) if p < 8 else 0)
        unit = rng.choice(acgt, size=p)
        arr = []
        while len(arr) < L:
            for ch in unit:
                x = rng.random()
                if x < ind / 2:
                    continue
                if x < ind:
                    arr.append(int(rng.choice(acgt)))
                arr.append(int(ch) if rng.random() >= div else int(rng.choice(acgt)))
        seq[pos:pos + L] = arr[:L]
        planted.append([pos, pos + L, p, copies, div, ind])
        pos += L + int(rng.integers(300, 1500))
    return seq.tobytes().decode(), planted


def _mut_indel(rng, s, psub, pindel):
    out = []
    for ch in s:
        r = rng.random()
        if r < psub:
            out.append("ACGT"[int(rng.integers(0, 4))])
        elif r < psub + pindel / 2:
            continue
        elif r < psub + pindel:
            out.append(ch)
            out.append("ACGT"[int(rng.integers(0, 4))])
        else:
            out.append(ch)
    return "".join(out)


def make_itr_cases(seed, n, long_=False):
    """inputs of the terminal-inverted-repeat filter (itrsearch, Util.py:216): records as search_confident_tir_batch_v1 hands them
    over (80 bases and shorter: first 40 + last 40) or, long_=True, whole low-copy sequences as remove_no_tirs does (up to ~2 kb,
    tandem-masked stretches of N).  Terminal inverted repeats of 5-40 (long: 10-700) bases with 0-30 % substitutions, indels,
    a few bases of overhang on either end; a fifth of the records carries none."""
    rng = np.random.default_rng(seed)
    seqs = []
    for _ in range(n):
        if long_:
            t = rand_seq(rng, int(rng.integers(10, 700)))
            a = _mut_indel(rng, t, float(rng.choice([0, .05, .1, .2, .35])), float(rng.choice([0, .01, .03])))
            if rng.random() < 0.25:
                a = rand_seq(rng, len(a))                   # no inverted repeat at all
            s = (rand_seq(rng, int(rng.choice([0, 0, 1, 3]))) + t + rand_seq(rng, int(rng.integers(0, 900))) + revcomp(a) +
                 rand_seq(rng, int(rng.choice([0, 0, 2]))))
            if rng.random() < 0.3:
                p = int(rng.integers(0, len(s)))
                q = min(len(s), p + int(rng.integers(1, 60)))
                s = s[:p] + "N" * (q - p) + s[q:]
        elif rng.random() < 0.2:
            s = rand_seq(rng, int(rng.integers(10, 90)))
        else:
            t = rand_seq(rng, int(rng.integers(5, 41)))
            a = _mut_indel(rng, t, float(rng.choice([0, .05, .1, .2, .3])), float(rng.choice([0, 0, .03, .08])))
            s = (rand_seq(rng, int(rng.choice([0, 0, 0, 1, 2, 3]))) + t + rand_seq(rng, int(rng.integers(0, 60))) + revcomp(a) +
                 rand_seq(rng, int(rng.choice([0, 0, 0, 1, 2, 3]))))
            if rng.random() < 0.8:
                s = s[:40] + s[-40:]
            if rng.random() < 0.1:
                b = list(s)
                for _k in range(int(rng.integers(1, 5))):
                    b[int(rng.integers(0, len(b)))] = "N"
                s = "".join(b)
        seqs.append(s)
    return seqs


def make_tir_batch(seed, n=60, flank=50):
    """flanked candidates for search_confident_tir_batch_v1 (Util.py:6533): elements with / without terminal inverted repeats of
    varying quality, TSDs of every length the k-mer search knows, boundaries off by a few bases, some with runs of N"""
    rng = np.random.default_rng(seed)
    names, seqs = [], []
    for q in range(n):
        te_len = int(rng.choice([90, 160, 400, 1200, 4500]))
        tsd_len = int(rng.choice([2, 3, 4, 5, 6, 8, 9, 10, 11]))
        tsd = {2: "TA", 4: "TTAA"}.get(tsd_len, rand_seq(rng, tsd_len))
        kind = rng.random()
        if kind < 0.25:
            te = rand_seq(rng, te_len)                      # no terminal structure
        else:
            tl = int(rng.choice([5, 8, 12, 20, 35]))
            tir = rand_seq(rng, tl)
            if kind < 0.35:
                tir = "CACTA" + tir[5:] if tl >= 5 else tir
            other = _mut_indel(rng, tir, float(rng.choice([0, 0, .1, .25])), float(rng.choice([0, 0, .05])))
            te = tir + rand_seq(rng, max(10, te_len - len(tir) - len(other))) + revcomp(other)
        if kind > 0.93:
            te = "CCC" + te[3:-3] + "GGG"
        left, right = rand_seq(rng, flank + 30), rand_seq(rng, flank + 30)
        full = left + tsd + te + tsd + right
        s0 = len(left) + len(tsd)
        off_l, off_r = int(rng.integers(-6, 7)), int(rng.integers(-6, 7))
        cs, ce = max(0, s0 + off_l - flank), min(len(full), s0 + len(te) + off_r + flank)
        cand = full[cs:ce]
        if q % 17 == 5:
            cand = cand[:70] + "N" * 10 + cand[80:]        # the batch function skips these (Util.py:6542)
        elif q % 11 == 3:
            cand = cand[:flank + 12] + "NN" + cand[flank + 14:]
        names.append("N_%d" % q)
        seqs.append(cand)
    return names, seqs
