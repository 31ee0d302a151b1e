"""ctypes binding of libhite_gpu.so (include/hite_gpu.h).  This is the stub a HiTE maintainer
would add next to module/Util.py.  There is NO CPU fallback: if the HIP
library is missing or no GPU is present every call raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libhite_gpu.so")

HITE_TE = {"tir": 0, "helitron": 1, "non_ltr": 2}
INFO = {0: "", 1: "nb", 2: "fl1", 3: "EXC"}


class HiteError(RuntimeError):
    pass


class HiteCall(C.Structure):
    _fields_ = [("is_te", C.c_int32), ("info", C.c_int32), ("row_num", C.c_int32), ("bstart", C.c_int32),
                ("bend", C.c_int32), ("cons_len", C.c_int32), ("cons_off", C.c_int64)]


CALL_DTYPE = np.dtype([("is_te", "<i4"), ("info", "<i4"), ("row_num", "<i4"), ("bstart", "<i4"), ("bend", "<i4"),
                       ("cons_len", "<i4"), ("cons_off", "<i8")])
assert CALL_DTYPE.itemsize == 32 and C.sizeof(HiteCall) == 32

_lib = None


def load():
    """Load libhite_gpu.so; raises HiteError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise HiteError("libhite_gpu.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(or make -C hite_amd/csrc); there is no CPU fallback")
        lib = C.CDLL(SO_PATH)
        lib.hite_last_error.restype = C.c_char_p
        lib.hite_genome_bases.restype = C.c_int64
        lib.hite_host_free.restype = None
        lib.hite_host_free.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _arr(x, dtype):
    return np.ascontiguousarray(x, dtype=dtype)


class Context:
    """One per process/GPU.  Owns the resident 2-bit genome and device scratch."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.hite_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise HiteError("hite_ctx_create(device=%d) failed: %d (no usable GPU?)" % (device, rc))
        self.h = h
        self.device = device

    def copy_config(self, aligned_interval):
        """which interval the copy records of THIS context carry (include/hite_gpu.h hite_copy_config_ctx): True = the aligned interval
        as get_copies_minimap2 reports it (Util.py:8026; the default), False = the whole candidate (rounds 2-4), None = follow the
        process-wide setting again (hite_copy_config; HITE_COPY_INTERVAL = aligned | whole, else the default)"""
        self._check(self.lib.hite_copy_config_ctx(self.h, -1 if aligned_interval is None else int(bool(aligned_interval))), "hite_copy_config_ctx")

    def release_copy_index(self):
        """drops the minimizer index of the packed genome (device memory is freed); the next copy / seeding call rebuilds it"""
        if getattr(self, "_copy_state", None) is not None and getattr(self, "h", None):
            self.lib.hite_copy_index_release(self._copy_state)
        self._copy_state = None

    def close(self):
        self.release_copy_index()
        if getattr(self, "_pipe_state", None) is not None and getattr(self, "h", None):
            self.lib.hite_pipeline_release(self._pipe_state)
            self._pipe_state = C.c_void_p(None)
        if getattr(self, "h", None):
            self.lib.hite_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.hite_last_error(self.h)
            raise HiteError("%s failed: %d %s" % (what, rc, (msg or b"").decode(errors="replace")))

    # ---- genome ---------------------------------------------------------------------------
    def genome_pack(self, seqs):
        """seqs: list of str/bytes contigs (upper-case)."""
        bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        off = np.zeros(len(bs) + 1, dtype=np.int64)
        np.cumsum([len(b) for b in bs], out=off[1:])
        buf = np.frombuffer(b"".join(bs), dtype=np.uint8)
        self._check(self.lib.hite_genome_pack(self.h, _p(buf), _p(off), len(bs)), "hite_genome_pack")
        self.contig_len = np.diff(off)

    def set_contig_order(self, names):
        """contig names (in packing order) -> the byte order of "<name>:" that breaks length ties between alignment rows
        (tools/ready_for_MSA.sh); None clears it (contigs then compare by index)"""
        if names is None:
            self._check(self.lib.hite_set_contig_order(self.h, None, 0), "hite_set_contig_order")
            return
        keyed = sorted(range(len(names)), key=lambda i: (names[i] + ":").encode())
        rank = np.zeros(len(names), dtype=np.int32)
        rank[keyed] = np.arange(len(names), dtype=np.int32)
        self._check(self.lib.hite_set_contig_order(self.h, _p(rank), len(names)), "hite_set_contig_order")

    def genome_pack_dev(self, d_ptr, contig_off, stream=0):
        off = _arr(contig_off, np.int64)
        self._check(self.lib.hite_genome_pack_dev(self.h, C.c_void_p(d_ptr), _p(off), len(off) - 1, C.c_void_p(stream)),
                    "hite_genome_pack_dev")
        self.contig_len = np.diff(off)

    # ---- flank windows --------------------------------------------------------------------
    def flank_gather(self, contig, start1, end1, minus, flank=50):
        """-> (windows: list[bytes|None], trunc: list[bytes|None]) following Util.py:8095-8124"""
        n = len(contig)
        contig = _arr(contig, np.int32)
        start1 = _arr(start1, np.int64)
        end1 = _arr(end1, np.int64)
        minus = _arr(minus, np.uint8)
        ln = np.zeros(n, dtype=np.int64)
        tl = np.zeros(n, dtype=np.int64)
        self._check(self.lib.hite_flank_sizes(self.h, C.c_int64(n), _p(contig), _p(start1), _p(end1), int(flank), _p(ln), _p(tl)),
                    "hite_flank_sizes")
        pad = (ln + 15) // 16 * 16
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(pad, out=off[1:])
        tpad = (tl + 15) // 16 * 16
        toff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(tpad, out=toff[1:])
        out = np.zeros(int(off[-1]) + 16, dtype=np.uint8)
        tout = np.zeros(int(toff[-1]) + 16, dtype=np.uint8)
        self._check(self.lib.hite_flank_gather(self.h, C.c_int64(n), _p(contig), _p(start1), _p(end1), _p(minus), int(flank),
                                               _p(off), _p(out), _p(toff), _p(tout)), "hite_flank_gather")
        wins = [out[off[i]:off[i] + ln[i]].tobytes() if ln[i] else None for i in range(n)]
        tr = [tout[toff[i]:toff[i] + tl[i]].tobytes() if tl[i] else None for i in range(n)]
        return wins, tr

    # ---- alignments -----------------------------------------------------------------------
    @staticmethod
    def pack_msas(msas):
        """msas: list of 2-D uint8 arrays -> (bytes, msa_off, rows, cols)"""
        rows = np.array([m.shape[0] for m in msas], dtype=np.int32)
        cols = np.array([m.shape[1] for m in msas], dtype=np.int32)
        sizes = rows.astype(np.int64) * cols
        pad = (sizes + 15) // 16 * 16
        off = np.zeros(len(msas) + 1, dtype=np.int64)
        np.cumsum(pad, out=off[1:])
        buf = np.full(int(off[-1]) + 16, ord("-"), dtype=np.uint8)
        for i, m in enumerate(msas):
            buf[off[i]:off[i] + sizes[i]] = np.ascontiguousarray(m, dtype=np.uint8).reshape(-1)
        return buf, off[:-1].copy(), rows, cols

    def sparse_cols(self, msas):
        buf, off, rows, cols = self.pack_msas(msas)
        out = np.zeros_like(buf)
        nc = np.zeros(len(msas), dtype=np.int32)
        self._check(self.lib.hite_sparse_cols(self.h, len(msas), _p(buf), _p(off), _p(rows), _p(cols), _p(out), _p(nc)),
                    "hite_sparse_cols")
        return [out[off[i]:off[i] + int(rows[i]) * int(nc[i])].reshape(int(rows[i]), int(nc[i])).copy() for i in range(len(msas))]

    def column_vote(self, msas):
        buf, off, rows, cols = self.pack_msas(msas)
        coff = np.zeros(len(msas) + 1, dtype=np.int64)
        np.cumsum(cols, out=coff[1:])
        counts = np.zeros((int(coff[-1]), 6), dtype=np.int32)
        self._check(self.lib.hite_column_vote(self.h, len(msas), _p(buf), _p(off), _p(rows), _p(cols), _p(coff), _p(counts)),
                    "hite_column_vote")
        return [counts[coff[i]:coff[i + 1]] for i in range(len(msas))]

    def boundary_search(self, msas, pos, side, thr, variant=3, int_thr=None, out_thr=None, win_in=20, win_out=10):
        buf, off, rows, cols = self.pack_msas(msas)
        n = len(msas)
        pos = _arr(pos, np.int32)
        side = _arr([0 if s in (0, "start") else 1 for s in side], np.int32)
        thr = _arr(thr, np.float64)
        it = _arr(int_thr, np.float64) if int_thr is not None else None
        ot = _arr(out_thr, np.float64) if out_thr is not None else None
        b = np.zeros(n, dtype=np.int32)
        v = np.zeros(n, dtype=np.int32)
        self._check(self.lib.hite_boundary_search(self.h, n, _p(buf), _p(off), _p(rows), _p(cols), _p(pos), _p(side), _p(thr),
                                                  _p(it), _p(ot), int(variant), win_in, win_out, _p(b), _p(v)),
                    "hite_boundary_search")
        return b, v

    def judge(self, te_type, msas, cands, plant=1):
        """-> list of (is_TE, info, cons, row_num, bstart, bend) -- the tuple judge_boundary_v5/v6/v9 return"""
        buf, off, rows, cols = self.pack_msas(msas)
        n = len(msas)
        cb = [c.encode() if isinstance(c, str) else bytes(c) for c in cands]
        coff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(c) for c in cb], out=coff[1:])
        cbuf = np.frombuffer(b"".join(cb) + b"\0" * 16, dtype=np.uint8)
        calls = np.zeros(n, dtype=CALL_DTYPE)
        cons = np.zeros(int(cols.astype(np.int64).sum()) + 8 * n + 16, dtype=np.uint8)
        self._check(self.lib.hite_judge(self.h, HITE_TE[te_type], int(plant), n, _p(buf), _p(off), _p(rows), _p(cols),
                                        _p(cbuf), _p(coff), _p(calls), _p(cons)), "hite_judge")
        out = []
        for i in range(n):
            c = calls[i]
            s = cons[c["cons_off"]:c["cons_off"] + c["cons_len"]].tobytes().decode() if c["is_te"] else ""
            out.append((bool(c["is_te"]), INFO[int(c["info"])], s, int(c["row_num"]), int(c["bstart"]), int(c["bend"])))
        return out

    def tsd_search(self, rows_, bstart, bend, plant=1):
        rb = [r.encode() if isinstance(r, str) else bytes(r) for r in rows_]
        n = len(rb)
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(r) for r in rb], out=off[1:])
        buf = np.frombuffer(b"".join(rb) + b"\0" * 16, dtype=np.uint8)
        ln = np.zeros(n, dtype=np.int32)
        lo = np.zeros((n, 16), dtype=np.uint8)
        ro = np.zeros((n, 16), dtype=np.uint8)
        self._check(self.lib.hite_tsd_search(self.h, n, _p(buf), _p(off), _p(_arr(bstart, np.int32)), _p(_arr(bend, np.int32)),
                                             int(plant), _p(ln), _p(lo), _p(ro)), "hite_tsd_search")
        return [(lo[i, :max(ln[i], 0)].tobytes().decode(), ro[i, :max(ln[i], 0)].tobytes().decode(), int(ln[i])) for i in range(n)]

    # ---- star alignment (stage where the reference calls mafft) -------------------------------
    def align_config(self, exact_cap):
        """exact_cap: 0 = fast (band of 128 centre rows only); 8 / 16 / 32 = widest band (x 32 rows) tried for an optimality
        certificate (default 8 or $HITE_ALIGN_EXACT)"""
        self._check(self.lib.hite_align_config(self.h, int(exact_cap)), "hite_align_config")

    def align_lanes(self, min_cols):
        """which pairs the lane-parallel kernels take (hite_align_lanes): -1 = by the size of the call (default or
        $HITE_ALIGN_LANES), >= 0 = pairs of at least this many columns (0: all), -2 = none.  Never changes a result."""
        self._check(self.lib.hite_align_lanes(self.h, int(min_cols)), "hite_align_lanes")

    def align_stats(self, reset=False):
        """-> dict(pairs, certified, wide, fallback, dropped, cost, columns, exact_cap) accumulated since the last reset"""
        o = np.zeros(8, dtype=np.int64)
        self._check(self.lib.hite_align_stats(self.h, _p(o), 1 if reset else 0), "hite_align_stats")
        return dict(zip(("pairs", "certified", "wide", "fallback", "dropped", "cost", "columns", "exact_cap"), (int(x) for x in o)))

    def star_msa(self, groups, sparse=False, info=False):
        """groups: list of lists of windows (bytes/str), first window of each group = centre.
        -> list of 2-D uint8 alignments (None where the alignment failed); rows that cannot be aligned are dropped.
        sparse=True: the fused path, sparse columns (remove_sparse_col_in_align_file) already removed.
        info=True: also, per group, an int32 array (windows x 5): cost U, certified, status, k*, band words (| 0x100 fall-back)."""
        flat = [w.encode() if isinstance(w, str) else bytes(w) for g in groups for w in g]
        n = len(groups)
        row_first = np.zeros(n + 1, dtype=np.int32)
        np.cumsum([len(g) for g in groups], out=row_first[1:])
        off = np.zeros(len(flat) + 1, dtype=np.int64)
        np.cumsum([len(w) for w in flat], out=off[1:])
        buf = np.frombuffer(b"".join(flat) + b"\0" * 16, dtype=np.uint8)
        cols = np.zeros(n, dtype=np.int32)
        rows = np.zeros(n, dtype=np.int32)
        inf = np.zeros((len(flat), 5), dtype=np.int32)
        moff = np.zeros(max(n, 1), dtype=np.int64)
        if n == 0:
            return ([], []) if info else []
        # ONE call: the library allocates the result (the sizes-then-fill protocol of hite_star_msa ran the alignment twice)
        ptr = C.POINTER(C.c_uint8)()
        nbytes = C.c_int64(0)
        self._check(self.lib.hite_star_msa_once(self.h, n, _p(buf), _p(off), _p(row_first), 1 if sparse else 0, _p(cols), _p(rows),
                                                _p(inf) if info else None, C.byref(ptr), _p(moff), C.byref(nbytes)), "hite_star_msa_once")
        try:
            out = np.ctypeslib.as_array(ptr, shape=(max(int(nbytes.value), 1),)).copy() if nbytes.value > 0 else np.zeros(1, dtype=np.uint8)
        finally:
            self.lib.hite_host_free(C.cast(ptr, C.c_void_p))
        res = []
        for i in range(n):
            if cols[i] <= 0:
                res.append(None)
            else:
                res.append(out[moff[i]:moff[i] + int(rows[i]) * cols[i]].reshape(int(rows[i]), int(cols[i])).copy())
        if info:
            return res, [inf[row_first[i]:row_first[i + 1]].copy() for i in range(n)]
        return res

    # ---- the fine stage in one call (body of flank_region_align_v5 after copy finding) -----------
    def flank_region_align(self, te_type, cands, copies, plant=1, flank=50):
        """cands: list of candidate sequences; copies: list (per candidate) of
        (contig_index, start1, end1, minus) tuples.  Needs genome_pack() first.
        Tuples as find_copies(..., clips=True) returns them -- (contig, start1, end1, minus, anchors, clip) -- hand their clip
        words on (hite_flank_region_align_clip: rows padded by the clipped candidate bases); shorter tuples: no pads.
        -> (list of (is_TE, info, cons, row_num, bstart, bend), stats)"""
        n = len(cands)
        cb = [c.encode() if isinstance(c, str) else bytes(c) for c in cands]
        coff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(c) for c in cb], out=coff[1:])
        cbuf = np.frombuffer(b"".join(cb) + b"\0" * 16, dtype=np.uint8)
        cf = np.zeros(n + 1, dtype=np.int32)
        np.cumsum([len(c) for c in copies], out=cf[1:])
        flat = [t for c in copies for t in c]
        contig = _arr([t[0] for t in flat], np.int32)
        s1 = _arr([t[1] for t in flat], np.int64)
        e1 = _arr([t[2] for t in flat], np.int64)
        mn = _arr([t[3] for t in flat], np.uint8)
        # the clip words travel as the 6th field of the tuples (find_copies(clips=True)); tuples WITHOUT it -- the reference's own
        # (chr, start, end, length, strand) records -- get theirs estimated by the library (hite_clip_probe's rule)
        has = [len(t) > 5 for t in flat]
        if any(has) and not all(has):
            raise ValueError("flank_region_align: copy tuples with and without a clip field in one table")
        clip = _arr([t[5] for t in flat], np.uint32) if flat and all(has) else None
        calls = np.zeros(n, dtype=CALL_DTYPE)
        cap = int(coff[-1]) + (2 * flank + 64) * n + 4096
        stats = np.zeros(12, dtype=np.int64)
        for _attempt in range(2):
            cons = np.zeros(cap + 16, dtype=np.uint8)
            rc = self.lib.hite_flank_region_align_clip(self.h, HITE_TE[te_type], int(plant), n, _p(cbuf), _p(coff), _p(cf),
                                                       C.c_int64(len(flat)), _p(contig), _p(s1), _p(e1), _p(mn),
                                                       _p(clip) if clip is not None else None, int(flank),
                                                       _p(calls), _p(cons), C.c_int64(cap), _p(stats))
            if rc == -4 and stats[10] > cap:     # HITE_ECAP: the consensus pool was too small; stats[10] says how much is needed
                cap = int(stats[10]) + 4096
                continue
            break
        self._check(rc, "hite_flank_region_align")
        out = []
        for i in range(n):
            c = calls[i]
            s = cons[c["cons_off"]:c["cons_off"] + c["cons_len"]].tobytes().decode() if c["is_te"] else ""
            out.append((bool(c["is_te"]), INFO[int(c["info"])], s, int(c["row_num"]), int(c["bstart"]), int(c["bend"])))
        return out, stats

    def clip_probe(self, cands, copies):
        """the clip words hite_flank_region_align estimates for a table without them (hite_clip_probe): copies as for
        flank_region_align -> per candidate list of words (left | right << 16, in the orientation of the genome)"""
        n = len(cands)
        cb = [c.encode() if isinstance(c, str) else bytes(c) for c in cands]
        coff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(c) for c in cb], out=coff[1:])
        cbuf = np.frombuffer(b"".join(cb) + b"\0" * 16, dtype=np.uint8)
        cf = np.zeros(n + 1, dtype=np.int32)
        np.cumsum([len(c) for c in copies], out=cf[1:])
        flat = [t for c in copies for t in c]
        out = np.zeros(len(flat) + 1, dtype=np.uint32)
        self._check(self.lib.hite_clip_probe(self.h, n, _p(cbuf), _p(coff), _p(cf), C.c_int64(len(flat)), _p(_arr([t[0] for t in flat], np.int32)),
                                             _p(_arr([t[1] for t in flat], np.int64)), _p(_arr([t[2] for t in flat], np.int64)),
                                             _p(_arr([t[3] for t in flat], np.uint8)), _p(out)), "hite_clip_probe")
        return [[int(x) for x in out[cf[i]:cf[i + 1]]] for i in range(n)]

    # device-resident variant: every argument is a raw device pointer (e.g. torch tensor .data_ptr())
    def flank_region_align_dev(self, te_type, plant, n_cand, d_cand, d_cand_off, d_copy_first, n_copies, d_contig, d_start1,
                               d_end1, d_minus, flank, d_calls, d_cons, cons_cap, stream=0, d_clip=0):
        """d_clip: device pointer of the records' clip words (copy_clips_dev()), 0: the library estimates them (hite_clip_probe's rule)"""
        if not hasattr(self, "_pipe_state"):
            self._pipe_state = C.c_void_p(None)
        stats = np.zeros(12, dtype=np.int64)
        v = C.c_void_p
        rc = self.lib.hite_flank_region_align_clip_dev(self.h, C.byref(self._pipe_state), HITE_TE[te_type], int(plant), int(n_cand),
                                                       v(d_cand), v(d_cand_off), v(d_copy_first), C.c_int64(n_copies), v(d_contig),
                                                       v(d_start1), v(d_end1), v(d_minus), v(d_clip or None), int(flank), v(d_calls),
                                                       v(d_cons), C.c_int64(cons_cap), _p(stats), v(stream))
        self._check(rc, "hite_flank_region_align_clip_dev")
        return stats

    def profile(self, on=None, reset=False):
        """per-stage HIP-event timings recorded by the library on its launch stream"""
        if reset:
            self.lib.hite_profile_reset(self.h)
        if on is not None:
            self.lib.hite_profile_enable(self.h, 1 if on else 0)
        out = {}
        name = C.create_string_buffer(32)
        ms = C.c_double(0)
        cnt = C.c_int64(0)
        for i in range(self.lib.hite_profile_count(self.h)):
            if self.lib.hite_profile_get(self.h, i, name, C.byref(ms), C.byref(cnt)) == 0:
                out[name.value.decode()] = (ms.value, cnt.value)
        return out

    # ---- k-mer TSD seed matching (search_confident_tir_v4, Util.py:7734) -----------------------------
    def tsd_kmer(self, seqs, flank=50, plant=1):
        """-> per candidate list of (tsd_len, tir_start, tir_end, distance), canonical order, <= 100"""
        sb = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        n = len(sb)
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(s) for s in sb], out=off[1:])
        buf = np.frombuffer(b"".join(sb) + b"\0" * 16, dtype=np.uint8)
        rec = np.zeros((n, 100, 4), dtype=np.int32)
        cnt = np.zeros(n, dtype=np.int32)
        self._check(self.lib.hite_tsd_kmer(self.h, n, _p(buf), _p(off), int(flank), int(plant), _p(rec), _p(cnt)), "hite_tsd_kmer")
        return [[tuple(int(x) for x in rec[i, j]) for j in range(max(cnt[i], 0))] for i in range(n)]

    # ---- merge of the call records on a caller-owned RCCL communicator (no reference counterpart) ------------
    def allgather_records(self, nccl_comm, d_send, d_recv, bytes_per_rank, stream=None):
        """one ncclAllGather of bytes_per_rank bytes per rank (device pointers; nccl_comm = the caller's ncclComm_t as an int / c_void_p)"""
        self._check(self.lib.hite_allgather_records(self.h, C.c_void_p(nccl_comm), C.c_void_p(d_send), C.c_void_p(d_recv), C.c_int64(bytes_per_rank),
                                                    C.c_void_p(stream or 0)), "hite_allgather_records")

    # ---- terminal inverted repeats (run_itrsearch, Util.py:216: tools/itrsearch -i 0.7 -l 7) --------------------
    def itr_search(self, seqs, end_len=40, min_identity=0.7, min_len=7, match=10, mismatch=16, gap_open=32, gap_extend=32):
        """-> int32 [n, 8]: score, end1, end2, equal bases, aligned columns, found, "Length itr=", flags.  end_len > 0: the record
        is s[:end_len] + s[-end_len:] (search_confident_tir_batch_v1, Util.py:6564); end_len = 0: the whole sequence (remove_no_tirs)"""
        sb = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        n = len(sb)
        out = np.zeros((n, 8), dtype=np.int32)
        if n == 0:
            return out
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(s) for s in sb], out=off[1:])
        buf = np.frombuffer(b"".join(sb) + b"\0" * 16, dtype=np.uint8)
        self._check(self.lib.hite_itr_search(self.h, C.c_int64(n), _p(buf), _p(off), int(end_len), C.c_double(min_identity), int(min_len),
                                             int(match), int(mismatch), int(gap_open), int(gap_extend), _p(out)), "hite_itr_search")
        return out

    # ---- FMEA (get_longest_repeats_v4 + process_all_seqs, Util.py:4122) --------------------------------
    def fmea_chain(self, qseg, sseg, qs, qe, ss, se, seg_chrom, seg_off, skip_gap, max_len):
        """-> (chrom ids, starts, ends) of the longest_repeats keys, in the reference's insertion order"""
        n = len(qseg)
        a = lambda x, t: _arr(x, t)  # noqa: E731
        qseg, sseg = a(qseg, np.int32), a(sseg, np.int32)
        qs, qe, ss, se = a(qs, np.int64), a(qe, np.int64), a(ss, np.int64), a(se, np.int64)
        seg_chrom, seg_off = a(seg_chrom, np.int32), a(seg_off, np.int64)
        cap = max(16, n + 16)
        oc = np.zeros(cap, dtype=np.int32)
        os_ = np.zeros(cap, dtype=np.int64)
        oe = np.zeros(cap, dtype=np.int64)
        nout = C.c_int64(0)
        self._check(self.lib.hite_fmea_chain(self.h, C.c_int64(n), _p(qseg), _p(sseg), _p(qs), _p(qe), _p(ss), _p(se), len(seg_chrom),
                                             _p(seg_chrom), _p(seg_off), C.c_int64(skip_gap), C.c_int64(max_len), C.c_int64(cap),
                                             _p(oc), _p(os_), _p(oe), C.byref(nout)), "hite_fmea_chain")
        k = nout.value
        return oc[:k].copy(), os_[:k].copy(), oe[:k].copy()

    # ---- copy finding (stage where the reference calls minimap2, Util.py:7933) ---------------------------
    FIND_COPIES_BATCH = 1 << 18

    def find_copies_table(self, cands, restricted=False, clips=False):
        """the copy table as arrays: (copy_first int32[n + 1], contig, start1, end1, minus, anchors) -- the copies of candidate c are
        rows copy_first[c] .. copy_first[c + 1]; needs genome_pack() first; at most FIND_COPIES_BATCH candidates.
        clips=True: a 7th array, uint32 per record: the candidate bases the end extensions clipped, left | right << 16 in the
        orientation of the genome (hite_copy_clips) -- zero when the records carry the whole-candidate interval
        (hite_copy_config(0)); flank_region_align pads the rows of the star alignment with them.
        restricted=True: the caller uses the index of this genome for THIS candidate set only (the masking step of stage 3.1):
        hite_find_copies_restricted builds the index from just the genome minimizers the candidates look up -- same table,
        a fraction of the build; any later use of the handle rebuilds the full index by itself."""
        if getattr(self, "_copy_state", None) is None:
            self._copy_state = C.c_void_p(None)
        cb = [c.encode() if isinstance(c, str) else bytes(c) for c in cands]
        n = len(cb)
        if n > self.FIND_COPIES_BATCH:
            raise ValueError("find_copies_table: %d candidates in one call (limit %d)" % (n, self.FIND_COPIES_BATCH))
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(c) for c in cb], out=off[1:])
        buf = np.frombuffer(b"".join(cb) + b"\0" * 16, dtype=np.uint8)
        cap = 300 * n + 16
        cf = np.zeros(n + 1, dtype=np.int32)
        ct = np.zeros(cap, dtype=np.int32)
        s1 = np.zeros(cap, dtype=np.int64)
        e1 = np.zeros(cap, dtype=np.int64)
        mn = np.zeros(cap, dtype=np.uint8)
        an = np.zeros(cap, dtype=np.int32)
        nout = C.c_int64(0)
        fn = self.lib.hite_find_copies_restricted if restricted else self.lib.hite_find_copies
        self._check(fn(self.h, C.byref(self._copy_state), n, _p(buf), _p(off), C.c_int64(cap), _p(cf), _p(ct), _p(s1), _p(e1), _p(mn),
                       _p(an), C.byref(nout)), "hite_find_copies_restricted" if restricted else "hite_find_copies")
        k = int(cf[n])
        if clips:
            cl = np.zeros(k + 1, dtype=np.uint32)
            self._check(self.lib.hite_copy_clips(self._copy_state, C.c_int64(k + 1), _p(cl)), "hite_copy_clips")
            return cf, ct[:k], s1[:k], e1[:k], mn[:k], an[:k], cl[:k]
        return cf, ct[:k], s1[:k], e1[:k], mn[:k], an[:k]

    def find_copies(self, cands, restricted=False, clips=False):
        """-> per candidate list of (contig, start1, end1, minus, anchors); needs genome_pack() first (restricted, clips: see
        find_copies_table; clips=True appends the packed clip word to every tuple: flank_region_align reads it from there)"""
        if len(cands) > self.FIND_COPIES_BATCH:    # one device call handles < 2^19 candidates: larger libraries go in batches
            out = []
            for k in range(0, len(cands), self.FIND_COPIES_BATCH):     # (batches share one full index)
                out += self.find_copies(cands[k:k + self.FIND_COPIES_BATCH], clips=clips)
            return out
        tab = self.find_copies_table(cands, restricted, clips)
        cf, ct, s1, e1, mn, an = tab[:6]
        rows = list(zip(ct.tolist(), s1.tolist(), e1.tolist(), mn.tolist(), an.tolist(), *([tab[6].tolist()] if clips else [])))
        cf = cf.tolist()
        return [rows[cf[c]:cf[c + 1]] for c in range(len(cands))]

    def copy_index_build(self, stream=0, fresh=False):
        """fresh: the minimizer tiles the handle keeps from its last build are dropped first (hite_copy_index_forget)"""
        if getattr(self, "_copy_state", None) is None:
            self._copy_state = C.c_void_p(None)
        if fresh:
            self.lib.hite_copy_index_forget(self._copy_state)
        self._check(self.lib.hite_copy_index_build(self.h, C.byref(self._copy_state), C.c_void_p(stream)), "hite_copy_index_build")

    def find_copies_dev(self, n_cand, d_cand, d_cand_off, cand_bytes, stream=0):
        """device-resident: -> (n_copies, d_copy_first, d_contig, d_start1, d_end1, d_minus, d_anchors) raw device pointers.
        PRECONDITION (include/hite_gpu.h): the buffer at d_cand is readable for 16 bytes beyond cand_bytes -- allocate
        cand_bytes + 16 (the end extension fetches candidate words ahead of use)."""
        v = C.c_void_p
        outs = [v() for _ in range(6)]
        n = C.c_int64(0)
        rc = self.lib.hite_find_copies_dev(self.h, self._copy_state, int(n_cand), v(d_cand), v(d_cand_off), C.c_int64(cand_bytes),
                                           C.byref(outs[0]), C.byref(n), C.byref(outs[1]), C.byref(outs[2]), C.byref(outs[3]),
                                           C.byref(outs[4]), C.byref(outs[5]), v(stream))
        self._check(rc, "hite_find_copies_dev")
        return (n.value,) + tuple(o.value or 0 for o in outs)

    def copy_clips_dev(self):
        """raw device pointer of the clip words of the last find_copies_dev call's records (0: none) -- the d_clip of flank_region_align_dev"""
        d = C.c_void_p()
        n = C.c_int64(0)
        self._check(self.lib.hite_copy_clips_dev(self._copy_state, C.byref(d), C.byref(n)), "hite_copy_clips_dev")
        return d.value or 0

    def query_copies(self, qid, sid, qs, qe, ss, se, ident, qlen, slen=None, ns=None, qcov=0.95, scov=0.0, qthr=200, sthr=200, max_copy=100):
        """get_query_copies on an HSP table in file order -> per query list of (subject id, start, end, chain length, '+'/'-')"""
        n = len(qid)
        a = lambda x, t: _arr(x, t)  # noqa: E731
        qid, sid = a(qid, np.int32), a(sid, np.int32)
        qs, qe, ss, se = a(qs, np.int64), a(qe, np.int64), a(ss, np.int64), a(se, np.int64)
        ident = a(ident, np.float64) if ident is not None else None
        qlen = a(qlen, np.int64)
        nq = len(qlen)
        slen = a(slen, np.int64) if slen is not None else None
        if ns is None:
            ns = len(slen) if slen is not None else (int(sid.max()) + 1 if n else 1)
        cap = (max_copy + 1) * nq + 16
        cf = np.zeros(nq + 1, dtype=np.int64)
        osid = np.zeros(cap, dtype=np.int32)
        os_, oe, ol = (np.zeros(cap, dtype=np.int64) for _ in range(3))
        om = np.zeros(cap, dtype=np.uint8)
        nout = C.c_int64(0)
        self._check(self.lib.hite_query_copies(self.h, C.c_int64(n), _p(qid), _p(sid), _p(qs), _p(qe), _p(ss), _p(se),
                                               _p(ident) if ident is not None else None, nq, _p(qlen), int(ns),
                                               _p(slen) if slen is not None else None, C.c_double(qcov), C.c_double(scov), C.c_int64(qthr),
                                               C.c_int64(sthr), int(max_copy), C.c_int64(cap), _p(cf), _p(osid), _p(os_), _p(oe), _p(ol),
                                               _p(om), C.byref(nout)), "hite_query_copies")
        return [[(int(osid[i]), int(os_[i]), int(oe[i]), int(ol[i]), "-" if om[i] else "+") for i in range(cf[q], cf[q + 1])]
                for q in range(nq)]

    def chain_all(self, qid, sid, qs, qe, ss, se, nq, ns, qgap):
        """every chain of every cluster (hite_chain_all: the core of FMEA / get_full_length_copies_from_blastn_v1) ->
        per query list of (subject id, q_start, q_end, s_start, s_end, extend_num) in the reference's order"""
        n = len(qid)
        qid, sid = _arr(qid, np.int32), _arr(sid, np.int32)
        qs, qe, ss, se = _arr(qs, np.int64), _arr(qe, np.int64), _arr(ss, np.int64), _arr(se, np.int64)
        gap = _arr(qgap, np.int64)
        cap = n + 16
        cf = np.zeros(nq + 1, dtype=np.int64)
        osid, onext = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.int32)
        oqs, oqe, oss, ose = (np.zeros(cap, dtype=np.int64) for _ in range(4))
        tot = C.c_int64(0)
        self._check(self.lib.hite_chain_all(self.h, C.c_int64(n), _p(qid), _p(sid), _p(qs), _p(qe), _p(ss), _p(se), int(nq), int(ns), _p(gap),
                                            C.c_int64(cap), _p(cf), _p(osid), _p(oqs), _p(oqe), _p(oss), _p(ose), _p(onext), C.byref(tot)),
                    "hite_chain_all")
        return [[(int(osid[i]), int(oqs[i]), int(oqe[i]), int(oss[i]), int(ose[i]), int(onext[i])) for i in range(cf[q], cf[q + 1])]
                for q in range(nq)]

    def lib_chain(self, qid, sid, qs, qe, ss, se, seq_len, threshold, chunk_size=0):
        """library-vs-itself HSP table -> chain records [chunk, q, qs-1, qe, s, ss-1, se] in the reference's chunk-file order"""
        n = len(qid)
        a = lambda x, t: _arr(x, t)  # noqa: E731
        qid, sid = a(qid, np.int32), a(sid, np.int32)
        qs, qe, ss, se = a(qs, np.int64), a(qe, np.int64), a(ss, np.int64), a(se, np.int64)
        sl = a(seq_len, np.int64)
        cap = n + 16
        oc, oq, os_ = (np.zeros(cap, dtype=np.int32) for _ in range(3))
        oqs, oqe, oss, ose = (np.zeros(cap, dtype=np.int64) for _ in range(4))
        nout = C.c_int64(0)
        self._check(self.lib.hite_lib_chain(self.h, C.c_int64(n), _p(qid), _p(sid), _p(qs), _p(qe), _p(ss), _p(se), len(sl), _p(sl),
                                            C.c_double(threshold), C.c_int64(chunk_size), C.c_int64(cap), _p(oc), _p(oq), _p(oqs), _p(oqe),
                                            _p(os_), _p(oss), _p(ose), C.byref(nout)), "hite_lib_chain")
        k = nout.value
        return [[int(oc[i]), int(oq[i]), int(oqs[i]), int(oqe[i]), int(os_[i]), int(oss[i]), int(ose[i])] for i in range(k)]

    def lib_cluster(self, recs, seq_len, threshold):
        """chain records -> clusters (lists of sequence ids: the query, then the subjects in the order they joined)"""
        n = len(recs)
        col = lambda k, t: _arr([r[k] for r in recs], t)  # noqa: E731
        ch, q, qs, qe, s, ss, se = col(0, np.int32), col(1, np.int32), col(2, np.int64), col(3, np.int64), col(4, np.int32), col(5, np.int64), col(6, np.int64)
        sl = _arr(seq_len, np.int64)
        capc, capm = n + 2, 2 * n + 2
        cf = np.zeros(capc + 1, dtype=np.int64)
        mem = np.zeros(capm, dtype=np.int32)
        ncl = C.c_int64(0)
        self._check(self.lib.hite_lib_cluster(C.c_int64(n), _p(ch), _p(q), _p(qs), _p(qe), _p(s), _p(ss), _p(se), len(sl), _p(sl),
                                              C.c_double(threshold), C.c_int64(capc), C.c_int64(capm), _p(cf), _p(mem), C.byref(ncl)),
                    "hite_lib_cluster")
        return [[int(x) for x in mem[cf[c]:cf[c + 1]]] for c in range(ncl.value)]

    def msa_consensus(self, alignments):
        """batch of alignments (each a list of equal-length byte strings) -> list of consensus strings (cons_from_mafft_v1)"""
        nmat = len(alignments)
        if nmat == 0:
            return []
        rows = np.array([len(al) for al in alignments], dtype=np.int32)
        cols = np.array([len(al[0]) for al in alignments], dtype=np.int64)
        flat = []
        for al in alignments:
            for r in al:
                rb = r.encode() if isinstance(r, str) else bytes(r)
                if len(rb) != len(al[0]):
                    raise ValueError("ragged alignment")
                flat.append(rb)
        buf = np.frombuffer(b"".join(flat) + b"\0" * 16, dtype=np.uint8)
        moff = np.zeros(nmat + 1, dtype=np.int64)
        np.cumsum(rows.astype(np.int64) * cols, out=moff[1:])
        ooff = np.zeros(nmat + 1, dtype=np.int64)
        np.cumsum(cols, out=ooff[1:])
        cons = np.zeros(int(ooff[-1]) + 16, dtype=np.uint8)
        clen = np.zeros(nmat, dtype=np.int64)
        self._check(self.lib.hite_msa_consensus(self.h, nmat, _p(rows), _p(cols), _p(moff), _p(buf), _p(ooff), _p(cons), _p(clen)),
                    "hite_msa_consensus")
        return [cons[ooff[a]:ooff[a] + clen[a]].tobytes().decode("latin-1") for a in range(nmat)]

    def ltr_both_ends(self, alignments, cur_seqs, flank):
        """FiLTR get_both_ends_frame on a batch: alignments = lists of equal-length rows, cur_seqs = terminal sequences ->
        per alignment None (boundary not found) or (frames [(left, right)], full rows, new_start, new_end)"""
        n = len(alignments)
        if n == 0:
            return []
        rows = np.array([len(al) for al in alignments], dtype=np.int32)
        cols = np.array([len(al[0]) if al else 0 for al in alignments], dtype=np.int32)
        flat = []
        for al in alignments:
            for r in al:
                rb = r.encode() if isinstance(r, str) else bytes(r)
                if len(rb) != len(al[0]):
                    raise ValueError("ragged alignment")
                flat.append(rb)
        buf = np.frombuffer(b"".join(flat) + b"\0" * 16, dtype=np.uint8)
        moff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(rows.astype(np.int64) * cols, out=moff[1:])
        cb = [c.encode() if isinstance(c, str) else bytes(c) for c in cur_seqs]
        coff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(c) for c in cb], out=coff[1:])
        cbuf = np.frombuffer(b"".join(cb) + b"\0" * 16, dtype=np.uint8)
        froff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(rows.astype(np.int64) * 2 * flank, out=froff[1:])
        fuoff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(rows.astype(np.int64) * (2 * flank + cols.astype(np.int64)), out=fuoff[1:])
        frames = np.zeros(int(froff[-1]) + 16, dtype=np.uint8)
        full = np.zeros(int(fuoff[-1]) + 16, dtype=np.uint8)
        fcols = np.zeros(n, dtype=np.int32)
        npos = np.zeros(2 * n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        self._check(self.lib.hite_ltr_both_ends(self.h, n, _p(buf), _p(moff), _p(rows), _p(cols), _p(cbuf), _p(coff), int(flank), _p(frames),
                                                _p(froff), _p(full), _p(fuoff), _p(fcols), _p(npos), _p(status)), "hite_ltr_both_ends")
        out = []
        for a in range(n):
            if status[a] != 0:
                out.append(None)
                continue
            R, stride = int(rows[a]), 2 * flank + int(cols[a])
            fr = [(frames[froff[a] + r * 2 * flank:froff[a] + r * 2 * flank + flank].tobytes().decode("latin-1"),
                   frames[froff[a] + r * 2 * flank + flank:froff[a] + (r + 1) * 2 * flank].tobytes().decode("latin-1")) for r in range(R)]
            fu = [full[fuoff[a] + r * stride:fuoff[a] + r * stride + fcols[a]].tobytes().decode("latin-1") for r in range(R)]
            out.append((fr, fu, int(npos[2 * a]), int(npos[2 * a + 1])))
        return out

    def nonltr_prep(self, seqs, flank=50, win5=25):
        """search_polyA_TSD on a batch -> [(found_TSD, direct, tsd_start, tsd_len, lo, hi)]"""
        sb = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        n = len(sb)
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(s) for s in sb], out=off[1:])
        buf = np.frombuffer(b"".join(sb) + b"\0" * 16, dtype=np.uint8)
        out = np.zeros(6 * max(1, n), dtype=np.int64)
        self._check(self.lib.hite_nonltr_prep(self.h, n, _p(buf), _p(off), int(flank), int(win5), _p(out)), "hite_nonltr_prep")
        return [tuple(int(x) for x in out[6 * i:6 * i + 6]) for i in range(n)]

    def ltr_frame(self, matrices, flank, window=20, side="left"):
        """FiLTR flank-frame vote on a batch: matrices = list of lists of equal-length frame strings -> [(is_ltr, boundary)]"""
        n = len(matrices)
        rows = np.array([len(m) for m in matrices], dtype=np.int32)
        cols = np.array([len(m[0]) if m else 0 for m in matrices], dtype=np.int32)
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(rows.astype(np.int64) * cols, out=off[1:])
        buf = np.frombuffer(("".join("".join(m) for m in matrices)).encode() + b"\0" * 16, dtype=np.uint8)
        ok = np.zeros(n, dtype=np.int32)
        b = np.zeros(n, dtype=np.int32)
        self._check(self.lib.hite_ltr_frame(self.h, n, _p(buf), _p(off), _p(rows), _p(cols), int(flank), int(window),
                                            0 if side == "left" else 1, _p(ok), _p(b)), "hite_ltr_frame")
        return [(bool(o), int(x)) for o, x in zip(ok, b)]

    def genome_mask(self, contig, start1, end1):
        """N-mask the 1-based inclusive intervals of the resident genome (mask_genome_intactTE)"""
        c, a, b = _arr(contig, np.int32), _arr(start1, np.int64), _arr(end1, np.int64)
        self._check(self.lib.hite_genome_mask(self.h, C.c_int64(len(c)), _p(c), _p(a), _p(b)), "hite_genome_mask")

    def seed_segments(self, seg_len=1_000_000):
        """-> (seg_chrom int32[nseg], seg_off int64[nseg]): the 'chr$offset' segment table of the packed genome"""
        n = C.c_int32(0)
        self._check(self.lib.hite_seed_segments(self.h, C.c_int64(seg_len), 0, None, None, C.byref(n)), "hite_seed_segments")
        sc = np.zeros(max(1, n.value), dtype=np.int32)
        so = np.zeros(max(1, n.value), dtype=np.int64)
        self._check(self.lib.hite_seed_segments(self.h, C.c_int64(seg_len), n.value, _p(sc), _p(so), C.byref(n)), "hite_seed_segments")
        return sc[:n.value], so[:n.value]

    def seed_shard(self, rank, world):
        """the seeding calls of this context compute rank's share of `world` (hite_seed_shard); world <= 1: everything"""
        self._check(self.lib.hite_seed_shard(self.h, int(rank), int(world)), "hite_seed_shard")

    def _seed_allvsall_once(self, seg_len, max_anchors, cap):
        """one hite_seed_allvsall call -> (table or None, anchors): None when the call's anchors exceed what one call sorts"""
        if getattr(self, "_copy_state", None) is None:
            self._copy_state = C.c_void_p(None)
        stats = (C.c_int64 * 4)()
        n = C.c_int64(0)
        cap = int(cap) if cap is not None else 1 << 20
        while True:
            qseg = np.empty(cap, dtype=np.int32); sseg = np.empty(cap, dtype=np.int32)
            qs = np.empty(cap, dtype=np.int64); qe = np.empty(cap, dtype=np.int64)
            ss = np.empty(cap, dtype=np.int64); se = np.empty(cap, dtype=np.int64)
            rc = self.lib.hite_seed_allvsall(self.h, C.byref(self._copy_state), C.c_int64(seg_len), C.c_int64(max_anchors), C.c_int64(cap),
                                             _p(qseg), _p(sseg), _p(qs), _p(qe), _p(ss), _p(se), C.byref(n), stats)
            if rc == -4 and n.value > cap:   # HITE_ECAP with the needed size known: retry once with room
                cap = n.value + 16
                continue
            if rc == -4 and (int(stats[1]) > max_anchors or int(stats[1]) >= 0xffffffff):
                return None, int(stats[1])
            self._check(rc, "hite_seed_allvsall")
            break
        k = n.value
        return {"qseg": qseg[:k].copy(), "sseg": sseg[:k].copy(), "qs": qs[:k].copy(), "qe": qe[:k].copy(), "ss": ss[:k].copy(),
                "se": se[:k].copy(), "stats": tuple(int(x) for x in stats)}, int(stats[1])

    def seed_allvsall(self, seg_len=1_000_000, max_anchors=2_000_000_000, cap=None):
        """all-vs-all seeding of the packed genome -> dict(qseg, sseg, qs, qe, ss, se, stats): the HSP table fmea_chain takes.
        A search with more anchors than one call sorts (the merged library of eight population genomes: dozens of near-identical
        sequences per family, 5 x 10^9 anchors) runs in SHARES, one after the other -- the (strand, diagonal) ranges of hite_seed_shard,
        whose union, put in share order and sorted stably by (query segment, subject segment), is the whole table record for record
        (what hite_amd.dist does across ranks)."""
        tab, anchors = self._seed_allvsall_once(seg_len, max_anchors, cap)
        if tab is not None:
            return tab
        world = 2
        while world * max_anchors < 2 * anchors:
            world *= 2
        while True:
            parts = []
            try:
                for r in range(world):
                    self.seed_shard(r, world)
                    t, _a = self._seed_allvsall_once(seg_len, max_anchors, cap)
                    if t is None:
                        break
                    parts.append(t)
            finally:
                self.seed_shard(0, 0)
            if len(parts) == world:
                break
            world *= 2
            if world > 256:
                raise HiteError("hite_seed_allvsall: %d anchors do not fit %d shares" % (anchors, world // 2))
        cat = {k: np.concatenate([t[k] for t in parts]) for k in ("qseg", "sseg", "qs", "qe", "ss", "se")}
        order = np.lexsort((cat["sseg"], cat["qseg"]))          # (stable: equal keys stay in share order)
        out = {k: v[order] for k, v in cat.items()}
        st = np.asarray([t["stats"] for t in parts], dtype=np.int64)
        out["stats"] = (int(st[0, 0]), int(st[:, 1].sum()), int(st[:, 2].sum()), int(st[:, 3].sum()))
        out["shares"] = world
        return out

    def coarse_stage_dev(self, seg_len, seg_chrom, seg_off, skip_gap, max_len, max_anchors=8_000_000_000):
        """all-vs-all seeding + FMEA with the HSP table kept on the device -> ((chrom ids, starts, ends), seeding stats)"""
        if getattr(self, "_copy_state", None) is None:
            self._copy_state = C.c_void_p(None)
        v = C.c_void_p
        ptr = [v() for _ in range(6)]
        stats = (C.c_int64 * 4)()
        n = C.c_int64(0)
        self._check(self.lib.hite_seed_allvsall_dev(self.h, C.byref(self._copy_state), C.c_int64(seg_len), C.c_int64(max_anchors),
                                                    *[C.byref(x) for x in ptr], C.byref(n), stats), "hite_seed_allvsall_dev")
        sc, so = _arr(seg_chrom, np.int32), _arr(seg_off, np.int64)
        cap = max(16, n.value + 16)
        oc = np.empty(cap, dtype=np.int32); os_ = np.empty(cap, dtype=np.int64); oe = np.empty(cap, dtype=np.int64)
        nout = C.c_int64(0)
        if n.value:
            self._check(self.lib.hite_fmea_chain_dev(self.h, n, *ptr, len(sc), _p(sc), _p(so), C.c_int64(skip_gap), C.c_int64(max_len),
                                                     C.c_int64(cap), _p(oc), _p(os_), _p(oe), C.byref(nout)), "hite_fmea_chain_dev")
        k = nout.value
        return (oc[:k].copy(), os_[:k].copy(), oe[:k].copy()), tuple(int(x) for x in stats)

    def copy_stats(self):
        """sizes of the last find_copies call: (candidate minimizers, index hits, diagonal clusters, copies before the cap)"""
        out = (C.c_int64 * 4)()
        self._check(self.lib.hite_copy_stats(self._copy_state, out), "hite_copy_stats")
        return tuple(int(x) for x in out)

    def tr_mask(self, max_period=500):
        """tandem repeats of the resident genome -> N (in place, for every later stage); returns a bool array over the
        concatenated contigs (True = masked) -- the build's stage where the reference runs TRF (Util.py:2855)"""
        G = int(self.lib.hite_genome_bases(self.h))
        bits = np.zeros((G + 31) // 32 + 1, dtype=np.uint32)
        n = C.c_int64(0)
        self._check(self.lib.hite_tr_mask(self.h, int(max_period), _p(bits), C.byref(n)), "hite_tr_mask")
        m = np.unpackbits(bits.view(np.uint8), bitorder="little")[:G].astype(bool)
        assert int(m.sum()) == n.value
        return m

    def tr_mask_dev(self, max_period=500):
        """tr_mask without the bit map on the host: the resident genome is masked in place -> number of masked bases"""
        n = C.c_int64(0)
        self._check(self.lib.hite_tr_mask(self.h, int(max_period), None, C.byref(n)), "hite_tr_mask")
        return int(n.value)

    def flanking_seq_dev(self, chrom, start0, end0, flank, contig_len=None):
        """generate_final_result + flanking_seq (Util.py:4783, 4614) for intervals (contig id, 0-based start, end) of the resident
        genome: the window of each interval with `flank` bases either side, clamped into the contig as the reference clamps it
        -> total bytes gathered (hite_flank_gather: coordinates up, sequences down)"""
        c = _arr(chrom, np.int32)
        if len(c) == 0:
            return 0
        if contig_len is None:
            contig_len = self.contig_len
        clen = np.asarray(contig_len, dtype=np.int64)[c]
        s1 = _arr(start0, np.int64) + 1
        e1 = _arr(end0, np.int64).copy()
        s1 = np.where(s1 - 1 - flank < 0, flank + 1, s1)
        e1 = np.where(e1 + flank > clen, clen - flank, e1)
        n = len(c)
        ws, we, mn = np.ascontiguousarray(s1 - flank), np.ascontiguousarray(e1 + flank), np.zeros(n, dtype=np.uint8)
        ln, tl = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
        self._check(self.lib.hite_flank_sizes(self.h, C.c_int64(n), _p(c), _p(ws), _p(we), 0, _p(ln), _p(tl)), "hite_flank_sizes")
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum((ln + 15) // 16 * 16, out=off[1:])
        # flanking_seq wants the whole window only (the first500 + last500 form belongs to the copy windows of the fine stage); the
        # host buffer is kept between calls: fresh pages would be faulted in under the copy
        need = int(off[-1]) + 16
        if getattr(self, "_flank_out", None) is None or len(self._flank_out) < need:
            self._flank_out = np.empty(need + need // 8, dtype=np.uint8)
        out = self._flank_out
        self._check(self.lib.hite_flank_gather(self.h, C.c_int64(n), _p(c), _p(ws), _p(we), _p(mn), 0, _p(off), _p(out), None, None),
                    "hite_flank_gather")
        self.flank_windows = (out, off, ln)      # (bytes, 16-byte aligned offsets, lengths) of the last call
        return int(ln.sum())

    def copy_stats_ext(self):
        """copy_stats() + (chains with a long end to extend, the other chains, DP columns of the end extension, 0)"""
        out = (C.c_int64 * 8)()
        self._check(self.lib.hite_copy_stats_ext(self._copy_state, out), "hite_copy_stats_ext")
        return tuple(int(x) for x in out)

    def download(self, d_ptr, count, dtype):
        """numpy copy of `count` elements of `dtype` at raw device pointer d_ptr"""
        out = np.zeros(int(count), dtype=dtype)
        if count:
            self._check(self.lib.hite_memcpy_d2h(_p(out), C.c_void_p(d_ptr), C.c_int64(out.nbytes)), "hite_memcpy_d2h")
        return out
