"""TEST INFRASTRUCTURE: an object with the method names of hite_amd.Context that answers from the CPU twins / oracle
(tests/oracle_lib.py over oracle/*.c).  It lets a test drive the product's HOST glue (hite_amd/util.py: library merge, ...)
a second time with every device stage replaced by its twin, so that whole multi-stage results can be compared -- the
product itself never sees this file (tests/test_abi.py: nothing under hite_amd/ imports the oracle)."""
import numpy as np

import oracle_lib as O


class OracleCtx:
    def __init__(self):
        self.contigs = []
        self.h = 1           # (hite_amd.util.get_ctx keeps a context whose handle is alive)
        self.device = 0

    # ---- residency ------------------------------------------------------------------------------
    def genome_pack(self, contigs):
        self.contigs = [c.encode() if isinstance(c, str) else bytes(c) for c in contigs]
        self.contig_len = np.asarray([len(c) for c in self.contigs], dtype=np.int64)

    def release_copy_index(self):
        pass

    def tr_mask(self, max_period=500):
        """as Context.tr_mask: the twin's tandem mask over the concatenated contigs; the resident genome carries it from here on"""
        import test_trmask

        m = test_trmask.twin_mask(self.contigs, max_period)
        pos, out = 0, []
        for c in self.contigs:
            a = np.frombuffer(c, dtype=np.uint8).copy()
            a[m[pos:pos + len(a)]] = ord("N")
            out.append(a.tobytes())
            pos += len(a)
        self.contigs = out
        return m

    # ---- stages ---------------------------------------------------------------------------------
    def itr_search(self, seqs, end_len=40, min_identity=0.7, min_len=7, match=10, mismatch=16, gap_open=32, gap_extend=32):
        return O.itr_search(seqs, end_len, min_identity, min_len, match, mismatch, gap_open, gap_extend)

    def tsd_kmer(self, seqs, flank=50, plant=1):
        return [O.tir_kmer(s, flank + 1, len(s) - flank, flank, plant) for s in seqs]

    def seed_shard(self, rank, world):
        self._shard = (int(rank), int(world))

    def seed_allvsall(self, seg_len=1_000_000, max_anchors=None, cap=None):
        r, w = getattr(self, "_shard", (0, 0))
        O.lib().orc_seed_shard(r, w)
        try:
            t = O.seed_allvsall(self.contigs, seg_len)
        finally:
            O.lib().orc_seed_shard(0, 0)
        return {k: t[k] for k in ("qseg", "sseg", "qs", "qe", "ss", "se")}

    def seed_segments(self, seg_len=1_000_000):
        import numpy as _np
        sc, so = [], []
        for ci, c in enumerate(self.contigs):
            nseg = max(1, (len(c) + seg_len - 1) // seg_len)
            for k in range(nseg):
                sc.append(ci); so.append(k * seg_len)
        return _np.asarray(sc, dtype=_np.int32), _np.asarray(so, dtype=_np.int64)

    def fmea_chain(self, qseg, sseg, qs, qe, ss, se, seg_chrom, seg_off, skip_gap, max_len):
        import numpy as _np
        nchrom = int(max(seg_chrom)) + 1 if len(seg_chrom) else 1
        i32 = lambda x: _np.ascontiguousarray(x, dtype=_np.int32)  # noqa: E731
        i64 = lambda x: _np.ascontiguousarray(x, dtype=_np.int64)  # noqa: E731
        h = {"qseg": i32(qseg), "sseg": i32(sseg), "qs": i64(qs), "qe": i64(qe), "ss": i64(ss), "se": i64(se), "seg_chrom": _np.asarray(seg_chrom, dtype=_np.int32), "seg_off": _np.asarray(seg_off, dtype=_np.int64),
             "chrom_names": ["c%d" % i for i in range(nchrom)]}
        names = O.fmea(h, skip_gap, max_len)
        oc, os_, oe = [], [], []
        for nm in names:
            c, pos = nm.split(":")
            a, b = pos.split("-")
            oc.append(int(c[1:])); os_.append(int(a)); oe.append(int(b))
        return _np.asarray(oc, dtype=_np.int32), _np.asarray(os_, dtype=_np.int64), _np.asarray(oe, dtype=_np.int64)

    def query_copies(self, qid, sid, qs, qe, ss, se, ident, qlen, slen=None, ns=None, qcov=0.95, scov=0.0, qthr=200, sthr=200, max_copy=100):
        n = len(qid)
        if ns is None:
            ns = len(slen) if slen is not None else (max(sid) + 1 if n else 1)
        rows = [(int(qid[i]), int(sid[i]), int(qs[i]), int(qe[i]), int(ss[i]), int(se[i]), float(ident[i]) if ident is not None else 0.0) for i in range(n)]
        return O.query_copies(rows, list(qlen), list(slen) if slen is not None else [1] * int(ns), qcov, scov, qthr, sthr, max_copy)

    def chain_all(self, qid, sid, qs, qe, ss, se, nq, ns, qgap):
        return O.chain_all(qid, sid, qs, qe, ss, se, nq, ns, qgap)

    def lib_chain(self, qid, sid, qs, qe, ss, se, seq_len, threshold, chunk_size=0):
        rows = list(zip((int(x) for x in qid), (int(x) for x in sid), (int(x) for x in qs), (int(x) for x in qe), (int(x) for x in ss),
                        (int(x) for x in se)))
        return O.lib_chain(rows, list(seq_len), threshold, chunk_size)

    def lib_cluster(self, recs, seq_len, threshold):
        return O.lib_cluster(recs, list(seq_len), threshold)

    def star_msa(self, groups, sparse=False, info=False):
        """as Context.star_msa (not sparse); the per-window status of `info` (column 2: 0 = aligned) is recovered from the rows
        of the twin's alignment: they come in input order, a dropped window has no row (equal windows share their fate)"""
        assert not sparse
        res, infos = [], []
        for g in groups:
            wb = [w.encode() if isinstance(w, str) else bytes(w) for w in g]
            m, kept = O.star_msa(wb, rows=True)
            inf = np.zeros((len(wb), 5), dtype=np.int32)
            if m is None:
                res.append(None)
                inf[1:, 2] = 1
            else:
                k = 1
                for i in range(1, len(wb)):
                    if k < kept and m[k][m[k] != 45].tobytes() == wb[i]:
                        k += 1
                    else:
                        inf[i, 2] = 1
                assert k == kept, (k, kept)
                res.append(m)
            infos.append(inf)
        return (res, infos) if info else res

    def msa_consensus(self, alignments):
        return [O.cons_majority([r if isinstance(r, str) else bytes(r).decode() for r in al]) for al in alignments]
