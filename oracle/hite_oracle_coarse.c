/*
 * hite_oracle_coarse.c -- TEST INFRASTRUCTURE ONLY (see hite_oracle.c header).
 *
 * CPU restatement of the coarse-stage / candidate-prep arithmetic HiTE owns:
 *   orc_fmea          get_longest_repeats_v4 + process_all_seqs   Util.py:4122-4400, 4529-4569
 *   orc_flank_window  flank-window gather inside flank_region_align_v5  Util.py:8095-8124
 *   orc_flanking_seq  flanking_seq coordinates                    Util.py:4614-4634
 *   orc_tir_kmer      search_confident_tir_v4 (k-mer TSD seeds)   Util.py:7734-7845
 * Pinned against tests/golden/ fmea, gather, tir_kmer fixtures (generated from the
 * reference's Python by oracle/gen_golden.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EXC (-1000)
#define ORC_ECAP (-1001)
#define ORC_EINVAL (-1002)

typedef struct { int64_t qs, qe, ss, se; int32_t idx; } hsp_t;

/* stable merge sort on hsp_t with comparator */
typedef int (*hsp_cmp)(const hsp_t *, const hsp_t *);
static void msort(hsp_t *a, hsp_t *tmp, int n, hsp_cmp cmp) {
    if (n < 2) return;
    int h = n / 2;
    msort(a, tmp, h, cmp);
    msort(a + h, tmp, n - h, cmp);
    int i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = cmp(&a[j], &a[i]) < 0 ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, sizeof(hsp_t) * n);
}
static int cmp_fwd(const hsp_t *a, const hsp_t *b) { /* key (ss, se)   :4173 */
    if (a->ss != b->ss) return a->ss < b->ss ? -1 : 1;
    if (a->se != b->se) return a->se < b->se ? -1 : 1;
    return 0;
}
static int cmp_rev(const hsp_t *a, const hsp_t *b) { /* key (-ss, -se) :4174 */
    if (a->ss != b->ss) return a->ss > b->ss ? -1 : 1;
    if (a->se != b->se) return a->se > b->se ? -1 : 1;
    return 0;
}
static int cmp_q(const hsp_t *a, const hsp_t *b) { /* key (qs, qe)   :4231 */
    if (a->qs != b->qs) return a->qs < b->qs ? -1 : 1;
    if (a->qe != b->qe) return a->qe < b->qe ? -1 : 1;
    return 0;
}

typedef struct { int64_t qs, qe, ss, se; int32_t sseg; int32_t next; /* HSPs the chain took in beyond its first (cur_extend_num) */ } chain_t;

typedef struct { chain_t *v; int n, cap; } chainvec;
static void cv_push(chainvec *c, chain_t x) {
    if (c->n == c->cap) { c->cap = c->cap ? c->cap * 2 : 256; c->v = (chain_t *)realloc(c->v, sizeof(chain_t) * c->cap); }
    c->v[c->n++] = x;
}

/* chain one cluster (already holds HSPs of one strand)  :4229-4319 */
static void chain_cluster(hsp_t *cl, hsp_t *tmp, int n, int64_t gap, int32_t sseg, chainvec *out) {
    msort(cl, tmp, n, cmp_q);
    /* visited is keyed by the HSP tuple: identical tuples alias (:4235-4257) */
    int *canon = (int *)malloc(sizeof(int) * n);
    char *vis = (char *)calloc(n, 1);
    for (int i = 0; i < n; i++) {
        canon[i] = i;
        for (int j = i - 1; j >= 0 && cl[j].qs == cl[i].qs && cl[j].qe == cl[i].qe; j--)
            if (cl[j].ss == cl[i].ss && cl[j].se == cl[i].se) canon[i] = canon[j];
    }
    for (int i = 0; i < n; i++) {
        if (vis[canon[i]]) continue;
        int64_t pqs = cl[i].qs, pqe = cl[i].qe, pss = cl[i].ss, pse = cl[i].se;
        int32_t next = 0;
        vis[canon[i]] = 1;
        for (int j = i + 1; j < n; j++) {
            if (vis[canon[j]]) continue;
            int64_t cqs = cl[j].qs, cqe = cl[j].qe, css = cl[j].ss, cse = cl[j].se;
            if (cqe > pqe) {
                if (pss < pse && css < cse) {
                    if (cse > pse) {
                        if (cqs - pqe < gap && cqe > pqe && css - pse < gap) {
                            pqe = cqe; pss = pss < css ? pss : css; pse = cse; next++;
                            vis[canon[j]] = 1;
                        } else if (cqs - pqe >= gap) break;
                    }
                } else if (pss > pse && css > cse) {
                    if (cse < pse) {
                        if (cqs - pqe < gap && cqe > pqe && pse - css < gap) {
                            pqe = cqe; pss = pss > css ? pss : css; pse = cse; next++;
                            vis[canon[j]] = 1;
                        } else if (cqs - pqe >= gap) break;
                    }
                }
            }
        }
        chain_t c = {pqs, pqe, pss, pse, sseg, next};
        cv_push(out, c);
    }
    free(canon); free(vis);
}

/* cluster one strand list (sorted) and chain each cluster  :4176-4227 */
static void cluster_and_chain(hsp_t *v, int n, int rev, int64_t gap, int32_t sseg, hsp_t *tmp, chainvec *out) {
    if (n == 0) return;
    int start = 0; /* current cluster = v[start .. k) ; members stay in list order */
    for (int k = 1; k <= n; k++) {
        int closed = 0;
        if (k < n) {
            for (int e = k - 1; e >= start; e--) {
                int64_t d = rev ? v[e].se - v[k].ss : v[k].ss - v[e].se;
                if (d < gap && v[k].qe > v[e].qe) { closed = 1; break; }
            }
        }
        if (!closed) {
            int m = k - start;
            hsp_t *cl = (hsp_t *)malloc(sizeof(hsp_t) * m);
            memcpy(cl, v + start, sizeof(hsp_t) * m);
            chain_cluster(cl, tmp, m, gap, sseg, out);
            free(cl);
            start = k;
        }
    }
}

/* seen-set of (chrom, a, b) keys  :4155, :4370-4390 */
typedef struct { int64_t *k; char *used; size_t cap; } kset;
static size_t khash(int64_t c, int64_t a, int64_t b) {
    uint64_t h = (uint64_t)c * 0x9E3779B97F4A7C15ull;
    h ^= (uint64_t)a + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h ^= (uint64_t)b + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    return (size_t)(h ^ (h >> 29));
}
static int ks_has(const kset *s, int64_t c, int64_t a, int64_t b) {
    size_t i = khash(c, a, b) & (s->cap - 1);
    while (s->used[i]) {
        if (s->k[3 * i] == c && s->k[3 * i + 1] == a && s->k[3 * i + 2] == b) return 1;
        i = (i + 1) & (s->cap - 1);
    }
    return 0;
}
static void ks_add(kset *s, int64_t c, int64_t a, int64_t b) {
    size_t i = khash(c, a, b) & (s->cap - 1);
    while (s->used[i]) {
        if (s->k[3 * i] == c && s->k[3 * i + 1] == a && s->k[3 * i + 2] == b) return;
        i = (i + 1) & (s->cap - 1);
    }
    s->used[i] = 1; s->k[3 * i] = c; s->k[3 * i + 1] = a; s->k[3 * i + 2] = b;
}
static int64_t fl10(int64_t x) { /* python (x // 10) * 10 (floor)  :4566-4569 */
    int64_t q = x / 10;
    if (x % 10 != 0 && x < 0) q--;
    return q * 10;
}

typedef struct { int64_t s, e; int ord; } iv_t;

/*
 * n HSPs in file order.  qseg/sseg = segment ids (name 'chr$offset'); seg_chrom / seg_off give
 * the chromosome id and the offset of each segment.  Output = keys of the reference's
 * longest_repeats dict in insertion order (chrom id, start, end) ['chr:start-end'].
 * Returns number of intervals, or <0.
 */
int orc_fmea(int n, const int32_t *qseg, const int32_t *sseg, const int64_t *qs, const int64_t *qe,
             const int64_t *ss, const int64_t *se, int nseg, const int32_t *seg_chrom, const int64_t *seg_off,
             int64_t skip_gap, int64_t max_len, int cap, int32_t *out_chrom, int64_t *out_start, int64_t *out_end) {
    if (n < 0 || nseg <= 0) return ORC_EINVAL;
    for (int i = 0; i < n; i++) {
        if (qseg[i] < 0 || qseg[i] >= nseg || sseg[i] < 0 || sseg[i] >= nseg) return ORC_EINVAL;
        /* the reference divides by |qe-qs| and |se-ss| (:4270-4274): zero-length HSPs raise */
        if (qs[i] == qe[i] || ss[i] == se[i]) return ORC_EINVAL;
    }
    /* order: queries by first appearance, subjects (within query) by first appearance, then file order */
    int *first_q = (int *)malloc(sizeof(int) * nseg);
    for (int i = 0; i < nseg; i++) first_q[i] = -1;
    int *keep = (int *)malloc(sizeof(int) * (n + 1)), nk = 0;
    for (int i = 0; i < n; i++) {
        if (qseg[i] == sseg[i] && qs[i] == ss[i] && qe[i] == se[i]) continue;  /* :4138 */
        keep[nk++] = i;
        if (first_q[qseg[i]] < 0) first_q[qseg[i]] = i;
    }
    /* query list in first-appearance order */
    int *qorder = (int *)malloc(sizeof(int) * nseg), nq = 0;
    {
        char *seenq = (char *)calloc(nseg, 1);
        for (int t = 0; t < nk; t++) { int q = qseg[keep[t]]; if (!seenq[q]) { seenq[q] = 1; qorder[nq++] = q; } }
        free(seenq);
    }
    /* bucket HSP indices per query (file order) */
    int *qcount = (int *)calloc(nseg + 1, sizeof(int));
    for (int t = 0; t < nk; t++) qcount[qseg[keep[t]] + 1]++;
    for (int i = 0; i < nseg; i++) qcount[i + 1] += qcount[i];
    int *qfill = (int *)malloc(sizeof(int) * nseg);
    memcpy(qfill, qcount, sizeof(int) * nseg);
    int *byq = (int *)malloc(sizeof(int) * (nk + 1));
    for (int t = 0; t < nk; t++) byq[qfill[qseg[keep[t]]]++] = keep[t];

    kset seen;
    seen.cap = 64;
    while (seen.cap < (size_t)nk * 16 + 64) seen.cap <<= 1;
    seen.k = (int64_t *)malloc(sizeof(int64_t) * 3 * seen.cap);
    seen.used = (char *)calloc(seen.cap, 1);

    int n_out = 0, rc = 0;
    int *sfirst = (int *)malloc(sizeof(int) * nseg);
    int *sorder = (int *)malloc(sizeof(int) * nseg);
    int *scount = (int *)malloc(sizeof(int) * (nseg + 1));
    hsp_t *fwd = (hsp_t *)malloc(sizeof(hsp_t) * (nk + 1)), *rev = (hsp_t *)malloc(sizeof(hsp_t) * (nk + 1));
    hsp_t *tmp = (hsp_t *)malloc(sizeof(hsp_t) * (nk + 1));
    chainvec chains = {0, 0, 0};
    for (int i = 0; i < nseg; i++) sfirst[i] = -1;

    for (int qi = 0; qi < nq && rc == 0; qi++) {
        int q = qorder[qi];
        int b0 = qcount[q], b1 = qcount[q + 1];
        int ns = 0;
        for (int t = b0; t < b1; t++) { int s = sseg[byq[t]]; if (sfirst[s] < 0) { sfirst[s] = t; sorder[ns++] = s; } }
        chains.n = 0;
        for (int si = 0; si < ns; si++) {
            int s = sorder[si];
            int nf = 0, nr = 0;
            for (int t = b0; t < b1; t++) {
                int i = byq[t];
                if (sseg[i] != s) continue;
                hsp_t h = {qs[i], qe[i], ss[i], se[i], i};
                if (h.ss > h.se) rev[nr++] = h; else fwd[nf++] = h;           /* :4168-4172 */
            }
            msort(fwd, tmp, nf, cmp_fwd);
            msort(rev, tmp, nr, cmp_rev);
            cluster_and_chain(fwd, nf, 0, skip_gap, s, tmp, &chains);
            cluster_and_chain(rev, nr, 1, skip_gap, s, tmp, &chains);
        }
        for (int si = 0; si < ns; si++) sfirst[sorder[si]] = -1;
        /* :4324-4390 */
        int64_t qoff = seg_off[q];
        int32_t qchr = seg_chrom[q];
        iv_t *names = (iv_t *)malloc(sizeof(iv_t) * (chains.n + 1));
        int nn = 0;
        for (int c = 0; c < chains.n; c++) {
            chain_t *r = &chains.v[c];
            int32_t schr = seg_chrom[r->sseg];
            int64_t soff = seg_off[r->sseg];
            int64_t sst = soff + r->ss - 1, sen = soff + r->se;
            int64_t s1 = fl10(sst), s2 = s1 + 10, e1 = fl10(sen), e2 = e1 + 10;
            int64_t qst = qoff + r->qs - 1, qen = qoff + r->qe;
            int64_t qlen = r->qe - (r->qs - 1); if (qlen < 0) qlen = -qlen;
            int64_t a1 = fl10(qst), a2 = a1 + 10, b1_ = fl10(qen), b2 = b1_ + 10;
            if (!ks_has(&seen, schr, s1, e1) && !ks_has(&seen, schr, s1, e2) && !ks_has(&seen, schr, s2, e1) &&
                !ks_has(&seen, schr, s2, e2) && !ks_has(&seen, qchr, a1, b1_) && !ks_has(&seen, qchr, a1, b2) &&
                !ks_has(&seen, qchr, a2, b1_) && !ks_has(&seen, qchr, a2, b2)) {
                if (qlen >= 80 && qlen < max_len) { names[nn].s = qst; names[nn].e = qen; names[nn].ord = nn; nn++; }
            }
            ks_add(&seen, schr, s1, e1); ks_add(&seen, schr, s1, e2); ks_add(&seen, schr, s2, e1); ks_add(&seen, schr, s2, e2);
            ks_add(&seen, qchr, a1, b1_); ks_add(&seen, qchr, a1, b2); ks_add(&seen, qchr, a2, b1_); ks_add(&seen, qchr, a2, b2);
        }
        /* process_all_seqs / process_seq_group  :4529-4563: stable sort by length descending */
        for (int i = 1; i < nn; i++) {
            iv_t x = names[i];
            int j = i - 1;
            while (j >= 0 && (names[j].e - names[j].s) < (x.e - x.s)) { names[j + 1] = names[j]; j--; }
            names[j + 1] = x;
        }
        char *kp = (char *)malloc(nn + 1);
        memset(kp, 1, nn + 1);
        for (int i = 0; i < nn; i++) {
            if (!kp[i]) continue;
            for (int j = i + 1; j < nn; j++) {
                if (!kp[j]) continue;
                int64_t lo = names[i].s > names[j].s ? names[i].s : names[j].s;
                int64_t hi = names[i].e < names[j].e ? names[i].e : names[j].e;
                int64_t ov = hi - lo; if (ov < 0) ov = 0;
                if ((double)ov / (double)(names[j].e - names[j].s) >= 0.95) kp[j] = 0;
            }
        }
        for (int i = 0; i < nn; i++) {
            if (!kp[i]) continue;
            if (n_out >= cap) { rc = ORC_ECAP; break; }
            out_chrom[n_out] = qchr; out_start[n_out] = names[i].s; out_end[n_out] = names[i].e; n_out++;
        }
        free(kp); free(names);
    }
    free(chains.v); free(fwd); free(rev); free(tmp); free(sfirst); free(sorder); free(scount);
    free(seen.k); free(seen.used); free(byq); free(qfill); free(qcount); free(qorder); free(keep); free(first_q);
    return rc < 0 ? rc : n_out;
}

/* ------------------------------------------------------------------------------- */
/* flank-window gather  Util.py:8095-8124                                          */
/* copy = (start1, end1) 1-based inclusive on a contig of length clen (ASCII).      */
/* Returns window length written to out (0 = copy skipped), trunc_out gets the      */
/* first500+last500 form when the window is > 1000 (else *trunc_len = 0).           */
/* ------------------------------------------------------------------------------- */
/* ---------------------------------------------------------------------------------------------
 * Every chain of every cluster: the chaining core that FMEA (Util.py:10452-10645) and get_full_length_copies_from_blastn_v1
 * (:5907-6105) share with get_longest_repeats_v4, without that function's de-duplication.
 * n HSPs in file order (the caller has already dropped the lines its function skips), query id in [0, nq), subject id in
 * [0, ns), 1-based inclusive coordinates.  qgap[q] = the query's skip_gap (FMEA: the same for all; full-length copies:
 * ceil(len(query) * threshold) -- an integer d is < a real g exactly when d < ceil(g)).  Queries in order of first appearance,
 * subjects of a query in order of first appearance, forward clusters before reverse ones, chains in creation order: the
 * order of the reference's longest_queries lists.  Output per chain: query, (prev_query_start, prev_query_end,
 * prev_subject_start, prev_subject_end) as the reference holds them, subject, cur_extend_num.  Returns the number of chains
 * (ORC_ECAP if more than cap). */
int64_t orc_chain_all(int64_t n, const int32_t *qid, const int32_t *sid, const int64_t *qs, const int64_t *qe, const int64_t *ss,
                      const int64_t *se, int32_t nq, int32_t ns, const int64_t *qgap, int64_t cap, int32_t *o_q, int64_t *o_qs,
                      int64_t *o_qe, int32_t *o_s, int64_t *o_ss, int64_t *o_se, int32_t *o_next) {
    if (n < 0 || nq <= 0 || ns <= 0) return ORC_EINVAL;
    for (int64_t i = 0; i < n; i++) if (qid[i] < 0 || qid[i] >= nq || sid[i] < 0 || sid[i] >= ns) return ORC_EINVAL;
    int *qorder = (int *)malloc(sizeof(int) * nq), nqs = 0;
    char *seenq = (char *)calloc(nq, 1);
    int *qcount = (int *)calloc((size_t)nq + 1, sizeof(int));
    for (int64_t i = 0; i < n; i++) { if (!seenq[qid[i]]) { seenq[qid[i]] = 1; qorder[nqs++] = qid[i]; } qcount[qid[i] + 1]++; }
    for (int q = 0; q < nq; q++) qcount[q + 1] += qcount[q];
    int *qfill = (int *)malloc(sizeof(int) * nq);
    memcpy(qfill, qcount, sizeof(int) * nq);
    int *byq = (int *)malloc(sizeof(int) * ((size_t)n + 1));
    for (int64_t i = 0; i < n; i++) byq[qfill[qid[i]]++] = (int)i;
    int *sfirst = (int *)malloc(sizeof(int) * ns), *sorder = (int *)malloc(sizeof(int) * ns);
    for (int s = 0; s < ns; s++) sfirst[s] = -1;
    hsp_t *fwd = (hsp_t *)malloc(sizeof(hsp_t) * ((size_t)n + 1)), *rev = (hsp_t *)malloc(sizeof(hsp_t) * ((size_t)n + 1));
    hsp_t *tmp = (hsp_t *)malloc(sizeof(hsp_t) * ((size_t)n + 1));
    chainvec chains = {0, 0, 0};
    int64_t nout = 0;
    int over = 0;
    for (int qi = 0; qi < nqs; qi++) {
        const int q = qorder[qi];
        const int b0 = qcount[q], b1 = qcount[q + 1];
        int nsub = 0;
        for (int t = b0; t < b1; t++) { const int s = sid[byq[t]]; if (sfirst[s] < 0) { sfirst[s] = t; sorder[nsub++] = s; } }
        chains.n = 0;
        for (int k = 0; k < nsub; k++) {
            const int s = sorder[k];
            int nf = 0, nr = 0;
            for (int t = b0; t < b1; t++) {
                const int h = byq[t];
                if (sid[h] != s) continue;
                hsp_t x = {qs[h], qe[h], ss[h], se[h], h};
                if (x.ss > x.se) rev[nr++] = x; else fwd[nf++] = x;
            }
            msort(fwd, tmp, nf, cmp_fwd);
            msort(rev, tmp, nr, cmp_rev);
            cluster_and_chain(fwd, nf, 0, qgap[q], s, tmp, &chains);
            cluster_and_chain(rev, nr, 1, qgap[q], s, tmp, &chains);
        }
        for (int k = 0; k < nsub; k++) sfirst[sorder[k]] = -1;
        for (int c = 0; c < chains.n; c++) {
            if (nout < cap) {
                o_q[nout] = q; o_qs[nout] = chains.v[c].qs; o_qe[nout] = chains.v[c].qe; o_s[nout] = chains.v[c].sseg;
                o_ss[nout] = chains.v[c].ss; o_se[nout] = chains.v[c].se; o_next[nout] = chains.v[c].next;
            } else over = 1;
            nout++;
        }
    }
    free(qorder); free(seenq); free(qcount); free(qfill); free(byq); free(sfirst); free(sorder); free(fwd); free(rev); free(tmp); free(chains.v);
    return over ? ORC_ECAP : nout;
}

int64_t orc_flank_window(const uint8_t *contig, int64_t clen, int64_t start1, int64_t end1, int minus, int64_t flank,
                         uint8_t *out, uint8_t *trunc_out, int64_t *trunc_len) {
    *trunc_len = 0;
    if (start1 - 1 - flank < 0 || end1 + flank > clen) return 0;             /* :8103 */
    int64_t lo = start1 - 1 - flank, hi = end1 + flank;
    if (hi < lo) hi = lo;
    int64_t n = hi - lo;
    if (!minus) {
        for (int64_t i = 0; i < n; i++) out[i] = contig[lo + i];
    } else {                                                                  /* getReverseSequence :1635 */
        for (int64_t i = 0; i < n; i++) {
            uint8_t c = contig[hi - 1 - i];
            out[i] = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'N';
        }
    }
    if (n < 100) return 0;                                                    /* :8108 */
    if (n > 1000) {                                                           /* :8117-8119 */
        memcpy(trunc_out, out, 500);
        memcpy(trunc_out + 500, out + n - 500, 500);
        *trunc_len = 1000;
    }
    return n;
}

/* flanking_seq  Util.py:4614-4634: name 'chr:s-e' (0-based half-open) -> slice [lo,hi) and the
 * 1-based name coordinates (name_s, name_e) of the flanked record */
void orc_flanking_seq(int64_t s, int64_t e, int64_t clen, int64_t flank, int64_t *lo, int64_t *hi, int64_t *name_s,
                      int64_t *name_e) {
    int64_t rs = s + 1, re = e;
    if (rs - 1 - flank < 0) rs = flank + 1;
    if (re + flank > clen) re = clen - flank;
    int64_t a = rs - 1 - flank, b = re + flank;
    /* python slice semantics for negative / out-of-range */
    if (a < 0) { a += clen; if (a < 0) a = 0; }
    if (b < 0) { b += clen; if (b < 0) b = 0; }
    if (a > clen) a = clen;
    if (b > clen) b = clen;
    if (b < a) b = a;
    *lo = a; *hi = b; *name_s = rs - flank; *name_e = re + flank;
}

/* ------------------------------------------------------------------------------- */
/* search_confident_tir_v4  Util.py:7734-7845                                      */
/* seq = flanked candidate; raw_start/raw_end 1-based as the caller passes them     */
/* (flank+1, len-flank), dist = tsd_search_distance.                                */
/* Output: up to cap records (tsd_len, tir_start, tir_end, distance) 0-based        */
/* inclusive, in CANONICAL order (distance, tir_start, tir_end, tsd_len) after all  */
/* the reference's filters, truncated to the top 100 by distance.  The reference's   */
/* order among equal distances is PYTHONHASHSEED-dependent (set iteration, :7741,    */
/* :7807), see SURVEY.md a-9.  Returns count.                                       */
/* ------------------------------------------------------------------------------- */
typedef struct { int k; int64_t ts, te, d; } tsdrec;
static int tsd_cmp(const void *a, const void *b) {
    const tsdrec *x = (const tsdrec *)a, *y = (const tsdrec *)b;
    if (x->d != y->d) return x->d < y->d ? -1 : 1;
    if (x->ts != y->ts) return x->ts < y->ts ? -1 : 1;
    if (x->te != y->te) return x->te < y->te ? -1 : 1;
    return x->k - y->k;
}

int orc_tir_kmer(const uint8_t *seq, int64_t n, int64_t raw_start, int64_t raw_end, int64_t dist, int plant, int cap,
                 int32_t *out_k, int64_t *out_ts, int64_t *out_te, int64_t *out_d) {
    static const int KS[9] = {2, 3, 4, 5, 6, 8, 9, 10, 11};
    raw_start -= 1; raw_end -= 1;
    int64_t ls = raw_start - dist; if (ls < 0) ls = 0;
    int64_t le = raw_start + dist + 1; if (le > n) le = n; if (le < ls) le = ls;
    int64_t rs = raw_end - dist; if (rs < 0) rs = 0;
    int64_t re = raw_end + dist + 1; if (re > n) re = n; if (re < rs) re = rs;
    /* python slices with negative ends cannot occur here: raw_* >= 0 in every caller */
    if (raw_start < 0 || raw_end < 0) return ORC_EINVAL;
    int64_t llen = le - ls, rlen = re - rs;
    tsdrec *rec = (tsdrec *)malloc(sizeof(tsdrec) * (size_t)(9 * (rlen + 1) + 1));
    int nrec = 0;
    for (int ki = 0; ki < 9; ki++) {
        int k = KS[ki];
        /* dict semantics: for every distinct right k-mer that also occurs on the left,
         * left_pos = left occurrence closest to raw_start (first wins ties, :7776),
         * right_pos evolves while scanning (:7788-7794) and a record is emitted at EVERY
         * right occurrence with the right_pos current at that moment (:7796-7804). */
        for (int64_t i = 0; i + k <= rlen; i++) {
            const uint8_t *rk = seq + rs + i;
            int64_t cur_pos = rs + i - 1;
            if (cur_pos < 0 || cur_pos > n - 1) continue;
            /* left_pos for this k-mer */
            int have_l = 0; int64_t lpos = 0;
            for (int64_t j = 0; j + k <= llen; j++) {
                int64_t lp = ls + j + k;
                if (lp < 0 || lp > n - 1) continue;
                if (memcmp(seq + ls + j, rk, k) != 0) continue;
                if (!have_l) { have_l = 1; lpos = lp; }
                else {
                    int64_t a = lp - raw_start, b = lpos - raw_start;
                    if (a < 0) a = -a;
                    if (b < 0) b = -b;
                    if (a < b) lpos = lp;
                }
            }
            if (!have_l) continue;
            /* right_pos = best among right occurrences 0..i of this k-mer (first wins ties) */
            int have_r = 0; int64_t rpos = 0;
            for (int64_t j = 0; j <= i; j++) {
                int64_t rp = rs + j - 1;
                if (rp < 0 || rp > n - 1) continue;
                if (memcmp(seq + rs + j, rk, k) != 0) continue;
                if (!have_r) { have_r = 1; rpos = rp; }
                else {
                    int64_t a = rp - raw_end, b = rpos - raw_end;
                    if (a < 0) a = -a;
                    if (b < 0) b = -b;
                    if (a < b) rpos = rp;
                }
            }
            int64_t ts = lpos, te = rpos;
            int ok;
            if (k != 2 && k != 4) ok = 1;
            else if (k == 4) ok = memcmp(rk, "TTAA", 4) == 0;
            else {
                ok = memcmp(rk, "TA", 2) == 0;
                if (!ok && plant == 0) {
                    /* first_3bp = seq[ts:ts+3]; last_3bp = seq[te-2:te+1] (python slices) */
                    int f = ts + 3 <= n && memcmp(seq + ts, "CCC", 3) == 0;
                    int l = te - 2 >= 0 && te + 1 <= n && memcmp(seq + te - 2, "GGG", 3) == 0;
                    ok = f && l;
                }
            }
            if (!ok) continue;
            /* set semantics: (kmer, ts, te) unique */
            int dup = 0;
            for (int t = 0; t < nrec; t++)
                if (rec[t].k == k && rec[t].ts == ts && rec[t].te == te &&
                    memcmp(seq + rs + (rec[t].d), rk, k) == 0) { dup = 1; break; }
            if (dup) continue;
            rec[nrec].k = k; rec[nrec].ts = ts; rec[nrec].te = te; rec[nrec].d = i; /* d holds the right index for now */
            nrec++;
        }
    }
    /* filters :7810-7831 and distance */
    int m = 0;
    for (int t = 0; t < nrec; t++) {
        int k = rec[t].k;
        const uint8_t *km = seq + rs + rec[t].d;
        int hasNN = 0;
        for (int i = 0; i + 1 < k; i++) if (km[i] == 'N' && km[i + 1] == 'N') hasNN = 1;
        if (hasNN) continue;
        int64_t ts = rec[t].ts, te = rec[t].te;
        int64_t a = ts, b = te + 1; /* tir_seq = seq[ts:te+1] */
        if (b > n) b = n;
        int64_t L = b - a; if (L < 0) L = 0;
        if (L < 100) continue;
        const uint8_t *tir = seq + a;
        if (tir[0] == 'T' && tir[1] == 'G' && tir[L - 2] == 'C' && tir[L - 1] == 'A') continue;
        if (L >= 8 && (memcmp(tir, "TATATATA", 8) == 0 || memcmp(tir, "ATATATAT", 8) == 0)) continue;
        int64_t d1 = ts - raw_start, d2 = te - raw_end;
        if (d1 < 0) d1 = -d1;
        if (d2 < 0) d2 = -d2;
        rec[m].k = k; rec[m].ts = ts; rec[m].te = te; rec[m].d = d1 + d2;
        m++;
    }
    qsort(rec, m, sizeof(tsdrec), tsd_cmp);
    if (m > 100) m = 100;
    if (m > cap) { free(rec); return ORC_ECAP; }
    for (int t = 0; t < m; t++) { out_k[t] = rec[t].k; out_ts[t] = rec[t].ts; out_te[t] = rec[t].te; out_d[t] = rec[t].d; }
    free(rec);
    return m;
}

/* ===================================================================================================
 * get_query_copies  /root/reference/module/Util.py:6828-7030 (input built by get_copies_v1 :7032-7060):
 * copy clustering on a blast6 HSP table (the blastn route of copy finding).
 * HSPs arrive in file order with dense ids: qid = rank of first appearance of the query name, sid = id of the
 * subject name; the subject dict of a query iterates in first-appearance order.
 * =================================================================================================== */
typedef struct { int64_t qs, qe, ss, se; double id; int64_t ord; } qc_frag;
static int qc_cmp_fwd(const void *a, const void *b) {   /* sort(key = (x[2], x[3])), stable */
    const qc_frag *x = (const qc_frag *)a, *y = (const qc_frag *)b;
    if (x->ss != y->ss) return x->ss < y->ss ? -1 : 1;
    if (x->se != y->se) return x->se < y->se ? -1 : 1;
    return x->ord < y->ord ? -1 : (x->ord > y->ord);
}
static int qc_cmp_rev(const void *a, const void *b) {   /* sort(key = (-x[2], -x[3])), stable */
    const qc_frag *x = (const qc_frag *)a, *y = (const qc_frag *)b;
    if (x->ss != y->ss) return x->ss > y->ss ? -1 : 1;
    if (x->se != y->se) return x->se > y->se ? -1 : 1;
    return x->ord < y->ord ? -1 : (x->ord > y->ord);
}
static int qc_cmp_q(const void *a, const void *b) {     /* sort(key = (x[0], x[1])), stable */
    const qc_frag *x = (const qc_frag *)a, *y = (const qc_frag *)b;
    if (x->qs != y->qs) return x->qs < y->qs ? -1 : 1;
    if (x->qe != y->qe) return x->qe < y->qe ? -1 : 1;
    return x->ord < y->ord ? -1 : (x->ord > y->ord);
}
static int qc_same(const qc_frag *a, const qc_frag *b) {
    return a->qs == b->qs && a->qe == b->qe && a->ss == b->ss && a->se == b->se && a->id == b->id;
}
typedef struct { int64_t qstart, qend, qlen, sstart, send, slen; int32_t sid; int64_t ord; } qc_long;
static int qc_cmp_long(const void *a, const void *b) {  /* sort(key = -x[2]), stable */
    const qc_long *x = (const qc_long *)a, *y = (const qc_long *)b;
    if (x->qlen != y->qlen) return x->qlen > y->qlen ? -1 : 1;
    return x->ord < y->ord ? -1 : (x->ord > y->ord);
}

/* one cluster (already sorted by (qs, qe)): the longest chain, Util.py:6900-6990.  returns 0 if the cluster is empty */
static int qc_cluster_best(qc_frag *cl, int n, int rev, int64_t qthr, int64_t sthr, qc_long *out) {
    if (n <= 0) return 0;
    uint8_t *vis = (uint8_t *)calloc(n, 1);
    int64_t best_len = -1;
    for (int i = 0; i < n; i++) {
        /* visited_frag is keyed by the tuple VALUE: a duplicate of a visited fragment is visited */
        int v = vis[i];
        for (int t = 0; !v && t < n; t++) if (vis[t] && qc_same(&cl[t], &cl[i])) v = 1;
        if (v) continue;
        int64_t cur_len = cl[i].qe - cl[i].qs + 1, lqs = cl[i].qs, lqe = cl[i].qe, lss = cl[i].ss, lse = cl[i].se;
        vis[i] = 1;
        for (int j = i + 1; j < n; j++) {
            int vj = vis[j];
            for (int t = 0; !vj && t < n; t++) if (vis[t] && qc_same(&cl[t], &cl[j])) vj = 1;
            if (vj) continue;
            const qc_frag *e = &cl[j];
            if (e->qe > lqe) {
                if (lss < lse && e->ss < e->se) {                     /* forward */
                    if (e->se > lse) {
                        if (e->qs - lqe < qthr && e->ss - lse < sthr) {
                            lqe = e->qe; lss = lss < e->ss ? lss : e->ss; lse = e->se; cur_len = lqe - lqs; vis[j] = 1;
                        } else if (e->qs - lqe >= qthr) break;
                    }
                } else if (lss > lse && e->ss > e->se) {              /* reverse */
                    if (e->se < lse) {
                        if (e->qs - lqe < qthr && lse - e->ss < sthr) {
                            lqe = e->qe; lss = lss > e->ss ? lss : e->ss; lse = e->se; cur_len = lqe - lqs; vis[j] = 1;
                        } else if (e->qs - lqe >= qthr) break;
                    }
                }
            }
        }
        if (cur_len > best_len) {
            best_len = cur_len;
            out->qstart = lqs; out->qend = lqe; out->qlen = cur_len; out->sstart = lss; out->send = lse;
            out->slen = (lse > lss ? lse - lss : lss - lse) + 1;
        }
    }
    free(vis);
    (void)rev;
    return best_len != -1;
}

/*
 * n HSPs in file order; qid / sid dense ids; qlen[nq], slen[ns] (slen may be NULL when scov <= 0).
 * Output CSR copy_first[nq + 1] into (sid, start, end (start <= end), chain length, minus); returns total or -1001 (cap).
 */
int64_t orc_query_copies(int64_t n, const int32_t *qid, const int32_t *sid, const int64_t *qs, const int64_t *qe, const int64_t *ss,
                         const int64_t *se, const double *ident, int nq, const int64_t *qlen, int ns, const int64_t *slen,
                         double qcov, double scov, int64_t qthr, int64_t sthr, int max_copy, int64_t cap, int32_t *copy_first,
                         int32_t *o_sid, int64_t *o_s, int64_t *o_e, int64_t *o_len, uint8_t *o_minus) {
    int64_t nout = 0;
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    qc_frag *fr = (qc_frag *)malloc(sizeof(qc_frag) * (n + 1));
    qc_long *lq = (qc_long *)malloc(sizeof(qc_long) * (n + 1));
    int32_t *sorder = (int32_t *)malloc(sizeof(int32_t) * (ns + 1));
    uint8_t *sseen = (uint8_t *)malloc(ns + 1);
    copy_first[0] = 0;
    for (int q = 0; q < nq; q++) {
        /* HSPs of this query in file order, subjects in first-appearance order */
        int64_t m = 0;
        for (int64_t i = 0; i < n; i++) if (qid[i] == q) idx[m++] = i;
        int nso = 0;
        for (int s = 0; s < ns; s++) sseen[s] = 0;
        for (int64_t t = 0; t < m; t++) { int s = sid[idx[t]]; if (!sseen[s]) { sseen[s] = 1; sorder[nso++] = s; } }
        int64_t nl = 0;
        for (int so = 0; so < nso; so++) {
            const int s = sorder[so];
            for (int rev = 0; rev < 2; rev++) {
                int64_t k = 0;
                for (int64_t t = 0; t < m; t++) {
                    const int64_t i = idx[t];
                    if (sid[i] != s) continue;
                    if ((ss[i] > se[i]) != rev) continue;          /* pos_item[2] > pos_item[3] -> reverse */
                    fr[k].qs = qs[i]; fr[k].qe = qe[i]; fr[k].ss = ss[i]; fr[k].se = se[i]; fr[k].id = ident[i]; fr[k].ord = k; k++;
                }
                if (k == 0) continue;
                qsort(fr, k, sizeof(qc_frag), rev ? qc_cmp_rev : qc_cmp_fwd);
                /* closed clusters: a fragment joins the current cluster if ANY member is close on the subject and ends
                 * earlier on the query (:6856-6893) */
                int64_t cs = 0;   /* start of the current cluster */
                for (int64_t t = 0; t <= k; t++) {
                    int close_ = 0;
                    if (t < k && t > cs)
                        for (int64_t u = t - 1; u >= cs && !close_; u--)
                            close_ = rev ? (fr[u].se - fr[t].ss < sthr && fr[t].qe > fr[u].qe) : (fr[t].ss - fr[u].se < sthr && fr[t].qe > fr[u].qe);
                    if (t == k || (t > cs && !close_)) {
                        /* cluster [cs, t): sort by (qs, qe), pick the longest chain */
                        for (int64_t u = cs; u < t; u++) fr[u].ord = u;
                        qsort(fr + cs, t - cs, sizeof(qc_frag), qc_cmp_q);
                        qc_long L;
                        if (qc_cluster_best(fr + cs, (int)(t - cs), rev, qthr, sthr, &L)) { L.sid = s; L.ord = nl; lq[nl++] = L; }
                        cs = t;
                    }
                }
            }
        }
        qsort(lq, nl, sizeof(qc_long), qc_cmp_long);
        int64_t first = nout;
        for (int64_t t = 0; t < nl; t++) {
            if (nout - first > max_copy) break;                       /* len(copies) > max_copy_num */
            int64_t a = lq[t].sstart, b = lq[t].send;
            int minus = 0;
            if (a > b) { int64_t x = a; a = b; b = x; minus = 1; }
            int dup = 0;
            for (int64_t u = first; u < nout && !dup; u++) dup = o_sid[u] == lq[t].sid && o_s[u] == a && o_e[u] == b;
            int ok = (double)lq[t].qlen / (double)qlen[q] >= qcov && !dup;
            if (scov > 0) ok = ok && (double)lq[t].slen / (double)slen[lq[t].sid] >= scov;
            if (!ok) continue;
            if (nout >= cap) { free(idx); free(fr); free(lq); free(sorder); free(sseen); return -1001; }
            o_sid[nout] = lq[t].sid; o_s[nout] = a; o_e[nout] = b; o_len[nout] = lq[t].qlen; o_minus[nout] = (uint8_t)minus;
            nout++;
        }
        copy_first[q + 1] = (int32_t)nout;
    }
    free(idx); free(fr); free(lq); free(sorder); free(sseen);
    return nout;
}
