/*
 * hite_oracle_lib.c -- TEST INFRASTRUCTURE ONLY (see hite_oracle.c header).
 *
 * CPU restatement of the library de-duplication arithmetic HiTE owns (panHiTE merge):
 *   orc_lib_chain     process_blast_results_in_chunks + process_chunk + extend_fragments
 *                     /root/reference/module/Util.py:12146-12200, 11958-12003, 11869-11944
 *   orc_lib_cluster   cluster_sequences_from_chunks + is_above_coverage_threshold   Util.py:12067-12115, 12053-12064
 *   orc_cons_majority cons_from_mafft_v1                                              Util.py:12515-12566
 * Pinned against tests/golden/lib_dedup.json.gz (outputs of the reference's Python, oracle/gen_golden.py).
 * The external steps between them (blastn, mafft, Ninja, cd-hit-est) are not restated.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_ECAP (-1001)
#define ORC_EINVAL (-1002)

typedef struct { int64_t qs, qe, ss, se, ord; } lc_frag;
static int lc_cmp_fwd(const void *a, const void *b) {   /* sort(key = (x[2], x[3])), stable  :11880 */
    const lc_frag *x = (const lc_frag *)a, *y = (const lc_frag *)b;
    if (x->ss != y->ss) return x->ss < y->ss ? -1 : 1;
    if (x->se != y->se) return x->se < y->se ? -1 : 1;
    return x->ord < y->ord ? -1 : (x->ord > y->ord);
}
static int lc_cmp_rev(const void *a, const void *b) {   /* sort(key = (-x[2], -x[3])), stable  :11882 */
    const lc_frag *x = (const lc_frag *)a, *y = (const lc_frag *)b;
    if (x->ss != y->ss) return x->ss > y->ss ? -1 : 1;
    if (x->se != y->se) return x->se > y->se ? -1 : 1;
    return x->ord < y->ord ? -1 : (x->ord > y->ord);
}

/* extend_fragments on one (query, subject, strand) list; returns the number of long fragments written to lf[] */
static int64_t lc_extend(lc_frag *fr, int64_t k, double skip_gap, int fwd, lc_frag *lf) {
    qsort(fr, k, sizeof(lc_frag), fwd ? lc_cmp_fwd : lc_cmp_rev);
    int64_t nl = 0;
    for (int64_t t = 0; t < k; t++) {
        const lc_frag c = fr[t];
        int upd = 0;
        for (int64_t u = nl - 1; u >= 0; u--) {              /* reversed(frag_index_array) */
            lc_frag *p = &lf[u];
            if (fwd) {
                if ((double)(c.ss - p->se) >= skip_gap) break;
                if ((double)(c.qs - p->qe) < skip_gap && c.qe > p->qe && c.se > p->se) {
                    if (c.qs < p->qs) p->qs = c.qs;
                    p->qe = c.qe;
                    if (c.ss < p->ss) p->ss = c.ss;
                    p->se = c.se;
                    upd = 1;
                }
            } else {
                if ((double)(p->se - c.ss) >= skip_gap) break;
                if ((double)(c.qs - p->qe) < skip_gap && c.qe > p->qe && c.se < p->se) {
                    p->qe = c.qe;
                    if (c.ss > p->ss) p->ss = c.ss;
                    p->se = c.se;
                    upd = 1;
                }
            }
        }
        if (!upd) lf[nl++] = c;
    }
    return nl;
}

/*
 * n blast6 lines in file order (dense ids into seq_len[nseq]); chunk_size <= 0: one chunk.  A line with
 * qid == sid, qs == ss, qe == se is skipped; its number still counts, but it never closes a chunk (:12183-12196).
 * Output records in the order of the reference's chunk files: chunk, query by first appearance in the chunk, subject by
 * first appearance, forward then reverse, long fragments in creation order:
 *   (o_chunk, o_q, o_qs = q_start - 1, o_qe, o_s, o_ss = s_start - 1, o_se)        :11996-11999
 * Returns the number of records or ORC_ECAP.
 */
int64_t orc_lib_chain(int64_t n, const int32_t *qid, const int32_t *sid, const int64_t *qs, const int64_t *qe, const int64_t *ss,
                      const int64_t *se, int nseq, const int64_t *seq_len, double threshold, int64_t chunk_size, int64_t cap,
                      int32_t *o_chunk, int32_t *o_q, int64_t *o_qs, int64_t *o_qe, int32_t *o_s, int64_t *o_ss, int64_t *o_se) {
    int64_t nout = 0;
    if (chunk_size <= 0) chunk_size = n > 0 ? n : 1;
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    lc_frag *fr = (lc_frag *)malloc(sizeof(lc_frag) * (size_t)(n + 1));
    lc_frag *lf = (lc_frag *)malloc(sizeof(lc_frag) * (size_t)(n + 1));
    int32_t *qorder = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nseq + 1)), *sorder = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nseq + 1));
    uint8_t *qseen = (uint8_t *)malloc((size_t)nseq + 1), *sseen = (uint8_t *)malloc((size_t)nseq + 1);
    int rc = 0;
    /* a chunk is closed by a KEPT line whose 1-based number is a multiple of chunk_size (the skipped self hits `continue`
     * past the boundary test, :12183-12196) */
    for (int64_t c0 = 0, chunk = 0; c0 < n && !rc; chunk++) {
        int64_t c1 = c0;
        while (c1 < n) {
            const int self = qid[c1] == sid[c1] && qs[c1] == ss[c1] && qe[c1] == se[c1];
            c1++;
            if (!self && c1 % chunk_size == 0) break;
        }
        int nqo = 0;
        memset(qseen, 0, (size_t)nseq + 1);
        for (int64_t i = c0; i < c1; i++) {
            if (qid[i] == sid[i] && qs[i] == ss[i] && qe[i] == se[i]) continue;
            if (!qseen[qid[i]]) { qseen[qid[i]] = 1; qorder[nqo++] = qid[i]; }
        }
        for (int qo = 0; qo < nqo && !rc; qo++) {
            const int q = qorder[qo];
            const double skip_gap = (double)seq_len[q] * (1 - threshold);             /* :11978 */
            int64_t m = 0;
            int nso = 0;
            memset(sseen, 0, (size_t)nseq + 1);
            for (int64_t i = c0; i < c1; i++) {
                if (qid[i] != q || (qid[i] == sid[i] && qs[i] == ss[i] && qe[i] == se[i])) continue;
                idx[m++] = i;
                if (!sseen[sid[i]]) { sseen[sid[i]] = 1; sorder[nso++] = sid[i]; }
            }
            for (int so = 0; so < nso && !rc; so++) {
                const int s = sorder[so];
                for (int fwd = 1; fwd >= 0 && !rc; fwd--) {
                    int64_t k = 0;
                    for (int64_t t = 0; t < m; t++) {
                        const int64_t i = idx[t];
                        if (sid[i] != s || (ss[i] <= se[i]) != fwd) continue;       /* pos[2] <= pos[3] -> forward  :11985 */
                        fr[k].qs = qs[i]; fr[k].qe = qe[i]; fr[k].ss = ss[i]; fr[k].se = se[i]; fr[k].ord = k; k++;
                    }
                    const int64_t nl = lc_extend(fr, k, skip_gap, fwd, lf);
                    for (int64_t u = 0; u < nl; u++) {
                        if (nout >= cap) { rc = 1; break; }
                        o_chunk[nout] = (int32_t)chunk; o_q[nout] = q; o_qs[nout] = lf[u].qs - 1; o_qe[nout] = lf[u].qe; o_s[nout] = s;
                        o_ss[nout] = lf[u].ss - 1; o_se[nout] = lf[u].se;
                        nout++;
                    }
                }
            }
        }
        c0 = c1;
    }
    free(idx); free(fr); free(lf); free(qorder); free(sorder); free(qseen); free(sseen);
    return rc ? ORC_ECAP : nout;
}

/*
 * Greedy star clustering over the chain records (in the order orc_lib_chain emits them; a new (chunk, query) run starts
 * wherever chunk or query changes).  cl_first[ncl + 1] into members[]; a cluster lists its query first, then the subjects
 * in the order they were added (the reference holds a Python set: membership is what is pinned).  Returns ncl or ORC_ECAP.
 */
int64_t orc_lib_cluster(int64_t nrec, const int32_t *chunk, const int32_t *q, const int64_t *qs, const int64_t *qe, const int32_t *s,
                        const int64_t *ss, const int64_t *se, int nseq, const int64_t *seq_len, double threshold, int64_t cap_cl,
                        int64_t cap_mem, int64_t *cl_first, int32_t *members) {
    uint8_t *red = (uint8_t *)calloc((size_t)nseq + 1, 1);
    int64_t ncl = 0, nm = 0;
    int rc = 0;
    cl_first[0] = 0;
    for (int64_t i = 0; i < nrec && !rc;) {
        int64_t j = i;
        while (j < nrec && chunk[j] == chunk[i] && q[j] == q[i]) j++;
        const int query = q[i];
        if (!red[query]) {                                                            /* :12084 */
            if (ncl >= cap_cl || nm >= cap_mem) { rc = 1; break; }
            members[nm++] = query;
            for (int64_t t = i; t < j; t++) {
                const int sub = s[t];
                if (red[sub]) continue;                                               /* :12098 */
                int64_t ql = qe[t] - qs[t], sl = se[t] - ss[t];
                if (ql < 0) ql = -ql;
                if (sl < 0) sl = -sl;
                if ((double)ql / (double)seq_len[query] >= threshold || (double)sl / (double)seq_len[sub] >= threshold) {   /* :12064 */
                    red[sub] = 1;
                    if (sub != query) {
                        if (nm >= cap_mem) { rc = 1; break; }
                        members[nm++] = sub;
                    }
                }
            }
            cl_first[++ncl] = nm;
        }
        i = j;
    }
    free(red);
    return rc ? ORC_ECAP : ncl;
}

/* cons_from_mafft_v1: per column the most frequent non-gap character if its count > R / 2 (integer division), else the
 * column is dropped.  mat row-major R x cols; returns the consensus length. */
int64_t orc_cons_majority(int R, int64_t cols, const uint8_t *mat, uint8_t *cons) {
    int64_t n = 0;
    for (int64_t c = 0; c < cols; c++) {
        int cnt[256];
        int first[256];
        memset(cnt, 0, sizeof cnt);
        for (int r = 0; r < R; r++) { uint8_t ch = mat[(int64_t)r * cols + c]; if (!cnt[ch]) first[ch] = r; cnt[ch]++; }
        int best = 0, bch = -1;
        /* dict order = first appearance down the column; strictly greater replaces */
        for (int ch = 0; ch < 256; ch++) {
            if (ch == '-' || !cnt[ch]) continue;
            if (cnt[ch] > best || (cnt[ch] == best && first[ch] < first[bch])) { best = cnt[ch]; bch = ch; }
        }
        if (best > R / 2 && bch >= 0) cons[n++] = (uint8_t)bch;
    }
    return n;
}
