"""ctypes binding of oracle/libhite_oracle.so -- the CHECKER.  Imported only by tests,
__graft_entry__.smoke() and bench.py's cpu_baseline leg (never by hite_amd/)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.environ.get("HITE_ORACLE_SO") or os.path.join(ORACLE_DIR, "libhite_oracle.so")     # (override: a gcov build, tools/oracle_line_coverage.sh)

ORC_EXC = -1000
_lib = None

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
ip = C.POINTER(C.c_int)


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in sorted(os.listdir(ORACLE_DIR)) if f.endswith(".c")]
        if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs):
            build()
        _lib = C.CDLL(SO)
        _lib.orc_flank_window.restype = C.c_int64
        _lib.orc_find_copies.restype = C.c_int64
    return _lib


def _u8(b):
    if isinstance(b, np.ndarray):        # (a slice of a memory-mapped genome: no copy)
        return np.ascontiguousarray(b, dtype=np.uint8)
    a = np.frombuffer(b if isinstance(b, (bytes, bytearray)) else b.encode(), dtype=np.uint8)
    return np.ascontiguousarray(a)


def _ptr(a, t):
    return a.ctypes.data_as(t)


def msa_array(seqs):
    R = len(seqs)
    Cn = len(seqs[0])
    assert all(len(s) == Cn for s in seqs)
    return np.frombuffer("".join(seqs).encode(), dtype=np.uint8).reshape(R, Cn).copy()


def sparse_cols(msa):
    R, Cn = msa.shape
    keep = np.zeros(Cn, dtype=np.uint8)
    rc = lib().orc_sparse_cols(_ptr(msa, u8p), R, Cn, _ptr(keep, u8p))
    assert rc == 0
    return keep


def column_vote(msa):
    """per-column counts of A, C, G, T, N, '-' from the oracle's col_base_map (Util.py:9251-9266)"""
    R, Cn = msa.shape
    out = np.zeros((Cn, 6), dtype=np.int32)
    rc = lib().orc_column_vote(_ptr(np.ascontiguousarray(msa), u8p), R, Cn, _ptr(out, i32p))
    assert rc == 0, rc
    return out


def search_v3(msa, pos, side, thr, win_in=20, win_out=10):
    R, Cn = msa.shape
    return lib().orc_search_v3(_ptr(msa, u8p), R, Cn, int(pos), 0 if side == "start" else 1, C.c_double(thr), win_in, win_out)


def search_v4(msa, pos, side, thr, int_thr, out_thr, win_in=20, win_out=10):
    R, Cn = msa.shape
    valid = C.c_int(0)
    b = lib().orc_search_v4(_ptr(msa, u8p), R, Cn, int(pos), 0 if side == "start" else 1, C.c_double(thr),
                            C.c_double(int_thr), C.c_double(out_thr), win_in, win_out, C.byref(valid))
    return bool(valid.value), b


def window_homology(msa, first, n, step, thr):
    R, Cn = msa.shape
    return lib().orc_window_homology(_ptr(msa, u8p), R, Cn, first, n, step, C.c_double(thr))


INFO = {0: "", 1: "nb", 2: "fl1"}


def judge(te_type, msa, cand, plant):
    """-> [is_TE, info, cons, row_num] or ['EXC'] ; plus (bstart, bend)"""
    R, Cn = msa.shape
    fn = {"tir": lib().orc_judge_v5, "helitron": lib().orc_judge_v6, "non_ltr": lib().orc_judge_v9}[te_type]
    cb = _u8(cand)
    cons = np.zeros(Cn + 8, dtype=np.uint8)
    clen = C.c_int(0)
    info = C.c_int(0)
    rn = C.c_int(0)
    bounds = (C.c_int * 2)(-1, -1)
    rc = fn(_ptr(msa, u8p), R, Cn, _ptr(cb, u8p), len(cb), int(plant), _ptr(cons, u8p), Cn + 8, C.byref(clen),
            C.byref(info), C.byref(rn), bounds)
    if rc < 0:
        return ["EXC", rc], (-1, -1)
    return [bool(rc), INFO[info.value], cons[:clen.value].tobytes().decode(), rn.value], (bounds[0], bounds[1])


def tsd_search_v5(row, bs, be, plant):
    rb = _u8(row)
    l = np.zeros(16, dtype=np.uint8)
    r = np.zeros(16, dtype=np.uint8)
    k = lib().orc_tsd_search_v5(_ptr(rb, u8p), len(rb), bs, be, plant, _ptr(l, u8p), _ptr(r, u8p))
    if k < 0:
        return None
    return l[:k].tobytes().decode(), r[:k].tobytes().decode()


def find_tail_polyA(s):
    b = _u8(s)
    return lib().orc_find_tail_polyA(_ptr(b, u8p), len(b))


def find_tandem_tail(s):
    b = _u8(s)
    return lib().orc_find_tandem_tail(_ptr(b, u8p), len(b))


def hsp_arrays(rows):
    """rows of (qname, sname, qs, qe, ss, se) -> dict of numpy arrays + segment tables"""
    segs = {}
    chroms = {}
    seg_chrom, seg_off = [], []

    def seg_id(name):
        if name not in segs:
            c, off = name.split("$")
            if c not in chroms:
                chroms[c] = len(chroms)
            segs[name] = len(segs)
            seg_chrom.append(chroms[c])
            seg_off.append(int(off))
        return segs[name]

    q = np.array([seg_id(r[0]) for r in rows], dtype=np.int32)
    s = np.array([seg_id(r[1]) for r in rows], dtype=np.int32)
    arr = lambda k: np.array([r[k] for r in rows], dtype=np.int64)  # noqa: E731
    return dict(qseg=q, sseg=s, qs=arr(2), qe=arr(3), ss=arr(4), se=arr(5),
                seg_chrom=np.array(seg_chrom, dtype=np.int32), seg_off=np.array(seg_off, dtype=np.int64),
                chrom_names=list(chroms.keys()))


def fmea(h, skip_gap, max_len):
    n = len(h["qseg"])
    cap = max(16, n + 16)
    oc = np.zeros(cap, dtype=np.int32)
    os_ = np.zeros(cap, dtype=np.int64)
    oe = np.zeros(cap, dtype=np.int64)
    rc = lib().orc_fmea(n, _ptr(h["qseg"], i32p), _ptr(h["sseg"], i32p), _ptr(h["qs"], i64p), _ptr(h["qe"], i64p),
                        _ptr(h["ss"], i64p), _ptr(h["se"], i64p), len(h["seg_chrom"]), _ptr(h["seg_chrom"], i32p),
                        _ptr(h["seg_off"], i64p), C.c_int64(skip_gap), C.c_int64(max_len), cap, _ptr(oc, i32p),
                        _ptr(os_, i64p), _ptr(oe, i64p))
    assert rc >= 0, rc
    return ["%s:%d-%d" % (h["chrom_names"][oc[i]], os_[i], oe[i]) for i in range(rc)]


def flank_window(contig_bytes, start1, end1, strand, flank=50):
    cb = _u8(contig_bytes)
    n = max(0, end1 - start1 + 1 + 2 * flank)
    out = np.zeros(n + 8, dtype=np.uint8)
    tr = np.zeros(1000, dtype=np.uint8)
    tl = C.c_int64(0)
    L = lib().orc_flank_window(_ptr(cb, u8p), C.c_int64(len(cb)), C.c_int64(start1), C.c_int64(end1),
                               1 if strand == "-" else 0, C.c_int64(flank), _ptr(out, u8p), _ptr(tr, u8p), C.byref(tl))
    if L == 0:
        return None, None
    return out[:L].tobytes().decode(), (tr[:tl.value].tobytes().decode() if tl.value else None)


def flanking_seq(s, e, clen, flank=50):
    lo, hi, ns, ne = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    lib().orc_flanking_seq(C.c_int64(s), C.c_int64(e), C.c_int64(clen), C.c_int64(flank), C.byref(lo), C.byref(hi),
                           C.byref(ns), C.byref(ne))
    return lo.value, hi.value, ns.value, ne.value


def tir_kmer(seq, raw_start, raw_end, dist, plant):
    b = _u8(seq)
    cap = 128
    k = np.zeros(cap, dtype=np.int32)
    ts = np.zeros(cap, dtype=np.int64)
    te = np.zeros(cap, dtype=np.int64)
    d = np.zeros(cap, dtype=np.int64)
    m = lib().orc_tir_kmer(_ptr(b, u8p), C.c_int64(len(b)), C.c_int64(raw_start), C.c_int64(raw_end), C.c_int64(dist),
                           int(plant), cap, _ptr(k, i32p), _ptr(ts, i64p), _ptr(te, i64p), _ptr(d, i64p))
    assert m >= 0, m
    return [(int(k[i]), int(ts[i]), int(te[i]), int(d[i])) for i in range(m)]


def star_msa(windows, rows=False):
    """windows: list of bytes/str (row 0 = centre) -> 2-D uint8 alignment (this build's mafft stand-in: every row aligned
    to the centre by the optimal unit-cost global alignment, see oracle/hite_oracle_nw.c; rows that cannot be aligned
    are dropped).  rows=True also returns the number of rows kept."""
    wb = [w.encode() if isinstance(w, str) else bytes(w) for w in windows]
    off = np.zeros(len(wb) + 1, dtype=np.int64)
    np.cumsum([len(w) for w in wb], out=off[1:])
    buf = np.frombuffer(b"".join(wb), dtype=np.uint8)
    cols, kept = C.c_int(0), C.c_int(0)
    rc = lib().orc_star_msa2(_ptr(buf, u8p), _ptr(off, i64p), len(wb), C.byref(cols), C.byref(kept), None, C.c_int64(0))
    if rc != 0:
        return (None, 0) if rows else None
    out = np.zeros((kept.value, cols.value), dtype=np.uint8)
    rc = lib().orc_star_msa2(_ptr(buf, u8p), _ptr(off, i64p), len(wb), C.byref(cols), C.byref(kept), _ptr(out, u8p),
                             C.c_int64(out.size))
    assert rc == 0, rc
    return (out, kept.value) if rows else out


def set_align_exact(cap):
    """exact-mode cap of the twin's pair schedule (0 fast, 8 / 16 / 32); returns the previous value"""
    prev = lib().orc_msa_get_exact()
    lib().orc_msa_set_exact(int(cap))
    return prev


def _seq(x):
    return np.frombuffer(x.encode() if isinstance(x, str) else bytes(x), dtype=np.uint8).copy()


def nw_pair(a, b):
    """THE definition (oracle/hite_oracle_nw.c): -> (ops uint16[m], distance)"""
    a, b = _seq(a), _seq(b)
    ops = np.zeros(len(a) + 1, dtype=np.uint16)
    d = lib().orc_nw_pair(_ptr(a, u8p), len(a), _ptr(b, u8p), len(b), _ptr(ops, C.POINTER(C.c_uint16)))
    assert d >= 0, d
    return ops[:len(a)], d


def nw_distance(a, b):
    a, b = _seq(a), _seq(b)
    d = lib().orc_nw_distance(_ptr(a, u8p), len(a), _ptr(b, u8p), len(b))
    assert d >= 0, d
    return d


def ops_cost(a, b, ops):
    a, b = _seq(a), _seq(b)
    ops = np.ascontiguousarray(ops, dtype=np.uint16)
    return lib().orc_ops_cost(_ptr(a, u8p), len(a), _ptr(b, u8p), len(b), _ptr(ops, C.POINTER(C.c_uint16)))


def bp_pair(a, b, nw=4, full=False):
    """one run of the banded bit-parallel aligner: -> (ops, dict(U, cert, status, kstar))"""
    a, b = _seq(a), _seq(b)
    ops = np.zeros(len(a) + 1, dtype=np.uint16)
    out = np.zeros(4, dtype=np.int32)
    rc = lib().orc_bp_pair(_ptr(a, u8p), len(a), _ptr(b, u8p), len(b), int(nw), 1 if full else 0,
                           _ptr(ops, C.POINTER(C.c_uint16)), _ptr(out, i32p))
    assert rc == 0, rc
    return ops[:len(a)], dict(U=int(out[0]), cert=int(out[1]), status=int(out[2]), kstar=int(out[3]))


def align_pair(a, b, exact_cap=8):
    """the product's schedule for one pair: -> (ops or None if the row is dropped, dict(U, cert, status, kstar, nw))"""
    a, b = _seq(a), _seq(b)
    ops = np.zeros(len(a) + 1, dtype=np.uint16)
    out = np.zeros(5, dtype=np.int32)
    rc = lib().orc_align_pair(_ptr(a, u8p), len(a), _ptr(b, u8p), len(b), int(exact_cap), _ptr(ops, C.POINTER(C.c_uint16)),
                              _ptr(out, i32p))
    assert rc >= 0, rc
    info = dict(U=int(out[0]), cert=int(out[1]), status=int(out[2]), kstar=int(out[3]), nw=int(out[4]))
    return (None if rc else ops[:len(a)]), info


def find_copies(contigs, cands, clips=False):
    """this build's minimap2 stand-in: -> per candidate list of (contig, start1, end1, minus, anchors); clips=True: + the clip word
    of the record (clipped candidate bases left | right << 16; zero in the whole-candidate mode, find_copies_config(False)), as Context.find_copies"""
    gb = [c.encode() if isinstance(c, str) else bytes(c) for c in contigs]
    coff = np.zeros(len(gb) + 1, dtype=np.int64)
    np.cumsum([len(c) for c in gb], out=coff[1:])
    gbuf = np.frombuffer(b"".join(gb), dtype=np.uint8)
    cb = [c.encode() if isinstance(c, str) else bytes(c) for c in cands]
    qoff = np.zeros(len(cb) + 1, dtype=np.int64)
    np.cumsum([len(c) for c in cb], out=qoff[1:])
    qbuf = np.frombuffer(b"".join(cb) + b"\0", dtype=np.uint8)
    cap = 300 * len(cb) + 16
    cf = np.zeros(len(cb) + 1, dtype=np.int32)
    ct = np.zeros(cap, dtype=np.int32)
    s1 = np.zeros(cap, dtype=np.int64)
    e1 = np.zeros(cap, dtype=np.int64)
    mn = np.zeros(cap, dtype=np.uint8)
    an = np.zeros(cap, dtype=np.int32)
    n = lib().orc_find_copies(_ptr(gbuf, u8p), _ptr(coff, i64p), len(gb), _ptr(qbuf, u8p), _ptr(qoff, i64p), len(cb), C.c_int64(cap),
                              _ptr(cf, i32p), _ptr(ct, i32p), _ptr(s1, i64p), _ptr(e1, i64p), _ptr(mn, u8p), _ptr(an, i32p))
    assert n >= 0, n
    if clips:
        cl, cr = np.zeros(n + 1, dtype=np.int32), np.zeros(n + 1, dtype=np.int32)
        lib().orc_find_copies_clips.restype = C.c_int64
        assert lib().orc_find_copies_clips(C.c_int64(n), _ptr(cl, i32p), _ptr(cr, i32p)) == n
        return [[(int(ct[i]), int(s1[i]), int(e1[i]), int(mn[i]), int(an[i]), int(cl[i]) | (int(cr[i]) << 16)) for i in range(cf[c], cf[c + 1])]
                for c in range(len(cb))]
    return [[(int(ct[i]), int(s1[i]), int(e1[i]), int(mn[i]), int(an[i])) for i in range(cf[c], cf[c + 1])] for c in range(len(cb))]


def clip_probe(cand, interval_seq):
    """clip word (left | right << 16, in the CANDIDATE's orientation) estimated for a copy record without one: twin of clip_probe_kernel"""
    q = np.frombuffer(cand.encode() if isinstance(cand, str) else bytes(cand), dtype=np.uint8)
    y = np.frombuffer(interval_seq.encode() if isinstance(interval_seq, str) else bytes(interval_seq), dtype=np.uint8)
    lib().orc_clip_probe.restype = C.c_uint32
    return int(lib().orc_clip_probe(_ptr(q, u8p), C.c_int64(len(q)), _ptr(y, u8p), C.c_int64(len(y))))


def find_copies_config(aligned_interval):
    """interval mode of the twin's records (mirror of hite_copy_config): True = the aligned interval (the default), False = the whole
    candidate, None = back to the default"""
    lib().orc_find_copies_config(1 if aligned_interval is None else int(bool(aligned_interval)))


def find_copies_far(min_copies):
    """far pass of the twin (mirror of hite_copy_far_pass): candidates with fewer copies are searched again in the (8, 13) index; 0 = off"""
    lib().orc_find_copies_far(int(min_copies))


def seed_allvsall(contigs, seg_len=1_000_000):
    """this build's blastn stand-in (twin): -> dict(qseg, sseg, qs, qe, ss, se) + the segment table"""
    gb = [c.encode() if isinstance(c, str) else bytes(c) for c in contigs]
    coff = np.zeros(len(gb) + 1, dtype=np.int64)
    np.cumsum([len(c) for c in gb], out=coff[1:])
    gbuf = np.frombuffer(b"".join(gb) + b"\0", dtype=np.uint8)
    L = lib()
    L.orc_seed_allvsall.restype = C.c_int64
    cap = 1 << 16
    while True:
        qseg = np.zeros(cap, dtype=np.int32); sseg = np.zeros(cap, dtype=np.int32)
        qs = np.zeros(cap, dtype=np.int64); qe = np.zeros(cap, dtype=np.int64)
        ss = np.zeros(cap, dtype=np.int64); se = np.zeros(cap, dtype=np.int64)
        n = L.orc_seed_allvsall(_ptr(gbuf, u8p), _ptr(coff, i64p), len(gb), C.c_int64(seg_len), C.c_int64(cap), _ptr(qseg, i32p),
                                _ptr(sseg, i32p), _ptr(qs, i64p), _ptr(qe, i64p), _ptr(ss, i64p), _ptr(se, i64p))
        if n == -1001:
            cap *= 4
            continue
        assert n >= 0, n
        break
    nseg = L.orc_seed_segments(_ptr(coff, i64p), len(gb), C.c_int64(seg_len), None, None, 0)
    sc = np.zeros(nseg, dtype=np.int32)
    so = np.zeros(nseg, dtype=np.int64)
    L.orc_seed_segments(_ptr(coff, i64p), len(gb), C.c_int64(seg_len), _ptr(sc, i32p), _ptr(so, i64p), nseg)
    return {"qseg": qseg[:n].copy(), "sseg": sseg[:n].copy(), "qs": qs[:n].copy(), "qe": qe[:n].copy(), "ss": ss[:n].copy(),
            "se": se[:n].copy(), "seg_chrom": sc, "seg_off": so}


def ltr_frame(rows, flank, window, side):
    """FiLTR flank-frame vote: rows = frames of one side -> (is_ltr, boundary)"""
    R = len(rows)
    Cn = len(rows[0]) if R else 0
    m = np.frombuffer("".join(rows).encode(), dtype=np.uint8).copy() if R else np.zeros(1, np.uint8)
    b = C.c_int(-1)
    ok = lib().orc_ltr_frame(_ptr(m, u8p), R, Cn, int(flank), int(window), 0 if side == "left" else 1, C.byref(b))
    return bool(ok), int(b.value)


def search_polyA_TSD(seq, flank=50, win5=25):
    """non-LTR candidate preparation (search_polyA_TSD, Util.py:10915) -> (found_TSD, TSD_seq, non_ltr_seq)"""
    b = seq.encode() if isinstance(seq, str) else bytes(seq)
    buf = np.frombuffer(b + b"\0", dtype=np.uint8)
    out = np.zeros(6, dtype=np.int64)
    L = lib()
    L.orc_search_polyA_TSD.restype = None
    L.orc_search_polyA_TSD(_ptr(buf, u8p), C.c_int64(len(b)), int(flank), int(win5), _ptr(out, i64p))
    found, direct, ts, tn, lo, hi = (int(x) for x in out)
    s = b.decode()
    nl = s[lo:hi] if direct else ""
    if direct == 2:
        comp = {"A": "T", "T": "A", "C": "G", "G": "C"}
        nl = "".join(comp.get(c, "N") for c in reversed(nl))
    return bool(found), s[ts:ts + tn] if found else "", nl


def query_copies(rows, qlen, slen, qcov, scov=0.0, qthr=200, sthr=200, max_copy=100):
    """get_query_copies: rows = (query id, subject id, qs, qe, ss, se, identity) in file order, query ids dense by first
    appearance -> per query list of (subject id, start, end, chain length, '+'/'-')"""
    n = len(rows)
    qid = np.array([r[0] for r in rows], dtype=np.int32); sid = np.array([r[1] for r in rows], dtype=np.int32)
    a = lambda k: np.array([r[k] for r in rows], dtype=np.int64)  # noqa: E731
    qs, qe, ss, se = a(2), a(3), a(4), a(5)
    idt = np.array([r[6] for r in rows], dtype=np.float64)
    ql = np.array(qlen, dtype=np.int64); sl = np.array(slen, dtype=np.int64)
    nq = len(qlen)
    cap = n + 16
    cf = np.zeros(nq + 1, dtype=np.int32)
    osid = np.zeros(cap, dtype=np.int32); os_ = np.zeros(cap, dtype=np.int64); oe = np.zeros(cap, dtype=np.int64)
    ol = np.zeros(cap, dtype=np.int64); om = np.zeros(cap, dtype=np.uint8)
    L = lib()
    L.orc_query_copies.restype = C.c_int64
    dp = C.POINTER(C.c_double)
    tot = L.orc_query_copies(C.c_int64(n), _ptr(qid, i32p), _ptr(sid, i32p), _ptr(qs, i64p), _ptr(qe, i64p), _ptr(ss, i64p), _ptr(se, i64p),
                             idt.ctypes.data_as(dp), nq, _ptr(ql, i64p), len(slen), _ptr(sl, i64p), C.c_double(qcov), C.c_double(scov),
                             C.c_int64(qthr), C.c_int64(sthr), int(max_copy), C.c_int64(cap), _ptr(cf, i32p), _ptr(osid, i32p),
                             _ptr(os_, i64p), _ptr(oe, i64p), _ptr(ol, i64p), _ptr(om, u8p))
    assert tot >= 0, tot
    return [[(int(osid[i]), int(os_[i]), int(oe[i]), int(ol[i]), "-" if om[i] else "+") for i in range(cf[q], cf[q + 1])] for q in range(nq)]


def chain_all(qid, sid, qs, qe, ss, se, nq, ns, qgap):
    """orc_chain_all -> per query list of (subject id, q_start, q_end, s_start, s_end, extend_num), as Context.chain_all"""
    n = len(qid)
    a32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)  # noqa: E731
    a64 = lambda x: np.ascontiguousarray(x, dtype=np.int64)  # noqa: E731
    qid, sid, qs, qe, ss, se, gap = a32(qid), a32(sid), a64(qs), a64(qe), a64(ss), a64(se), a64(qgap)
    cap = n + 16
    oq, os_, onx = (np.zeros(cap, dtype=np.int32) for _ in range(3))
    oqs, oqe, oss, ose = (np.zeros(cap, dtype=np.int64) for _ in range(4))
    L = lib()
    L.orc_chain_all.restype = C.c_int64
    k = L.orc_chain_all(C.c_int64(n), _ptr(qid, i32p), _ptr(sid, i32p), _ptr(qs, i64p), _ptr(qe, i64p), _ptr(ss, i64p), _ptr(se, i64p),
                        int(nq), int(ns), _ptr(gap, i64p), C.c_int64(cap), _ptr(oq, i32p), _ptr(oqs, i64p), _ptr(oqe, i64p), _ptr(os_, i32p),
                        _ptr(oss, i64p), _ptr(ose, i64p), _ptr(onx, i32p))
    assert k >= 0, k
    out = [[] for _ in range(nq)]
    for i in range(k):
        out[int(oq[i])].append((int(os_[i]), int(oqs[i]), int(oqe[i]), int(oss[i]), int(ose[i]), int(onx[i])))
    return out


def lib_chain(rows, lens, thr, chunk_size=0):
    """process_blast_results_in_chunks + process_chunk: rows (q, s, qs, qe, ss, se) -> records [chunk, q, qs-1, qe, s, ss-1, se]"""
    n = len(rows)
    a32 = lambda k: np.array([r[k] for r in rows], dtype=np.int32)  # noqa: E731
    a64 = lambda k: np.array([r[k] for r in rows], dtype=np.int64)  # noqa: E731
    qid, sid, qs, qe, ss, se = a32(0), a32(1), a64(2), a64(3), a64(4), a64(5)
    sl = np.array(lens, dtype=np.int64)
    cap = n + 16
    oc, oq, os_ = (np.zeros(cap, dtype=np.int32) for _ in range(3))
    oqs, oqe, oss, ose = (np.zeros(cap, dtype=np.int64) for _ in range(4))
    L = lib()
    L.orc_lib_chain.restype = C.c_int64
    k = L.orc_lib_chain(C.c_int64(n), _ptr(qid, i32p), _ptr(sid, i32p), _ptr(qs, i64p), _ptr(qe, i64p), _ptr(ss, i64p), _ptr(se, i64p),
                        len(lens), _ptr(sl, i64p), C.c_double(thr), C.c_int64(chunk_size), C.c_int64(cap), _ptr(oc, i32p), _ptr(oq, i32p),
                        _ptr(oqs, i64p), _ptr(oqe, i64p), _ptr(os_, i32p), _ptr(oss, i64p), _ptr(ose, i64p))
    assert k >= 0, k
    return [[int(oc[i]), int(oq[i]), int(oqs[i]), int(oqe[i]), int(os_[i]), int(oss[i]), int(ose[i])] for i in range(k)]


def lib_cluster(recs, lens, thr):
    """cluster_sequences_from_chunks on lib_chain records -> list of clusters (query first, then subjects in order added)"""
    n = len(recs)
    a32 = lambda k: np.array([r[k] for r in recs], dtype=np.int32)  # noqa: E731
    a64 = lambda k: np.array([r[k] for r in recs], dtype=np.int64)  # noqa: E731
    ch, q, qs, qe, s, ss, se = a32(0), a32(1), a64(2), a64(3), a32(4), a64(5), a64(6)
    sl = np.array(lens, dtype=np.int64)
    capc, capm = n + 2, 2 * n + 2
    cf = np.zeros(capc + 1, dtype=np.int64)
    mem = np.zeros(capm, dtype=np.int32)
    L = lib()
    L.orc_lib_cluster.restype = C.c_int64
    k = L.orc_lib_cluster(C.c_int64(n), _ptr(ch, i32p), _ptr(q, i32p), _ptr(qs, i64p), _ptr(qe, i64p), _ptr(s, i32p), _ptr(ss, i64p),
                          _ptr(se, i64p), len(lens), _ptr(sl, i64p), C.c_double(thr), C.c_int64(capc), C.c_int64(capm), _ptr(cf, i64p),
                          _ptr(mem, i32p))
    assert k >= 0, k
    return [[int(x) for x in mem[cf[c]:cf[c + 1]]] for c in range(k)]


def cons_majority(rows):
    """cons_from_mafft_v1 on equal-length aligned rows -> consensus string"""
    R, cols = len(rows), len(rows[0])
    mat = np.frombuffer("".join(rows).encode(), dtype=np.uint8).copy()
    out = np.zeros(cols + 1, dtype=np.uint8)
    L = lib()
    L.orc_cons_majority.restype = C.c_int64
    k = L.orc_cons_majority(R, C.c_int64(cols), _ptr(mat, u8p), _ptr(out, u8p))
    return out[:k].tobytes().decode()


def ltr_both_ends(rows, cur, flank):
    """FiLTR get_both_ends_frame on aligned rows -> (frames [(left, right)], full rows, new_start, new_end) or None"""
    R, Cc = len(rows), len(rows[0])
    mat = np.frombuffer("".join(rows).encode(), dtype=np.uint8).copy()
    cb = np.frombuffer(cur.encode(), dtype=np.uint8).copy()
    fr = np.zeros(R * 2 * flank + 16, dtype=np.uint8)
    stride = 2 * flank + Cc
    fu = np.zeros(R * stride + 16, dtype=np.uint8)
    fc, ns, ne = C.c_int(0), C.c_int(0), C.c_int(0)
    rc = lib().orc_ltr_both_ends(_ptr(mat, u8p), R, Cc, _ptr(cb, u8p), len(cur), flank, _ptr(fr, u8p), _ptr(fu, u8p), C.byref(fc),
                                 C.byref(ns), C.byref(ne))
    if rc != 0:
        return None if rc == 1 else rc
    frames = [[fr[r * 2 * flank:r * 2 * flank + flank].tobytes().decode(), fr[r * 2 * flank + flank:(r + 1) * 2 * flank].tobytes().decode()]
              for r in range(R)]
    full = [fu[r * stride:r * stride + fc.value].tobytes().decode() for r in range(R)]
    return frames, full, ns.value, ne.value


def itr_search(seqs, end_len=40, min_id=0.7, min_len=7, match=10, mismatch=16, gap_open=32, gap_extend=32, max_len=500):
    """itrsearch (third-party ELF the reference bundles, Util.py:216-224) as restated in oracle/hite_oracle_itr.c ->
    int32 [n, 8]: score, end1, end2, matches, aligned, found, header "Length itr=", flags"""
    bufs = [_u8(s) for s in seqs]
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(b) for b in bufs])
    cat = np.ascontiguousarray(np.concatenate(bufs + [np.zeros(16, np.uint8)]))
    out = np.zeros((len(seqs), 8), dtype=np.int32)
    rc = lib().orc_itr_search(len(seqs), _ptr(cat, u8p), _ptr(off, i64p), int(end_len), C.c_double(min_id), int(min_len), int(match),
                              int(mismatch), int(gap_open), int(gap_extend), int(max_len), _ptr(out, i32p))
    assert rc == 0
    return out
