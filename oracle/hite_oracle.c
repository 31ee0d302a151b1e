/*
 * hite_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the arithmetic HiTE itself owns on the
 * dynamic-boundary hot path.  It exists to CHECK the HIP path (tests/, smoke(), and the
 * cpu_baseline leg of bench.py); nothing in hite_amd/ may call, link or import it.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  It is pinned against tests/golden/ (json.gz fixtures), which were produced by
 * importing the reference's module/Util.py in the build container
 * (oracle/gen_golden.py).  Third-party arithmetic absent from the reference tree
 * (fuzzysearch.find_near_matches, Levenshtein.distance) follows the published
 * definition written down in oracle/stubs.py: PARITY UNPINNED at that boundary.
 *
 * Conventions: an alignment is R rows x C columns of bytes, row-major; '-' is the gap.
 * Return value < 0 is an error: ORC_EXC means "the reference raises a Python exception
 * on this input".
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EXC (-1000)
#define ORC_ECAP (-1001)
#define ORC_EINVAL (-1002)
#define MAXSYM 16

/* ------------------------------------------------------------------------------- */
/* small helpers                                                                   */
/* ------------------------------------------------------------------------------- */
static inline int64_t i64min(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int64_t i64max(int64_t a, int64_t b) { return a > b ? a : b; }
static inline int64_t i64abs(int64_t a) { return a < 0 ? -a : a; }

/* Python s[a:b] on a length-n sequence: normalise to 0 <= lo <= hi <= n */
static void py_slice(int64_t a, int64_t b, int64_t n, int64_t *lo, int64_t *hi) {
    if (a < 0) { a += n; if (a < 0) a = 0; }
    if (b < 0) { b += n; if (b < 0) b = 0; }
    if (a > n) a = n;
    if (b > n) b = n;
    if (b < a) b = a;
    *lo = a; *hi = b;
}

static uint8_t comp_base(uint8_t c) { /* getReverseSequence  Util.py:1635-1648 */
    switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; default: return 'N'; }
}

/* Levenshtein.distance: plain unit-cost edit distance (call site Util.py:9403) */
static int lev_dist(const uint8_t *a, int n, const uint8_t *b, int m) {
    int *prev = (int *)malloc(sizeof(int) * (m + 1) * 2), *cur = prev + m + 1, *t;
    for (int j = 0; j <= m; j++) prev[j] = j;
    for (int i = 1; i <= n; i++) {
        cur[0] = i;
        for (int j = 1; j <= m; j++) {
            int c = prev[j - 1] + (a[i - 1] != b[j - 1]);
            int d = prev[j] + 1, e = cur[j - 1] + 1;
            cur[j] = c < d ? (c < e ? c : e) : (d < e ? d : e);
        }
        t = prev; prev = cur; cur = t;
    }
    int r = prev[m];
    free(prev < cur ? prev : cur);
    return r;
}

/* find_near_matches(pattern, text, max_l_dist=k) -- definition in oracle/stubs.py.
 * Returns the number of groups; first group's best (start,end) in *fs,*fe and the last
 * group's best in *ls,*le.  (Callers only ever use [0].start and [-1].end.) */
static int fnm(const uint8_t *p, int m, const uint8_t *t, int n, int k, int *fs, int *fe, int *ls, int *le) {
    if (m <= 0) return ORC_EXC; /* ValueError: empty subsequence */
    int ngroups = 0;
    int gopen = 0, gend = -1, bs = 0, be = 0, bd = 0;
    int W = m + k;
    int *prev = (int *)malloc(sizeof(int) * (W + 1) * 2), *cur = prev + W + 1, *base = prev, *tt;
    for (int s = 0; s < n; s++) {
        int w = W < n - s ? W : n - s;
        if (w < m - k || w <= 0) continue;
        prev = base; cur = base + W + 1;
        for (int j = 0; j <= w; j++) prev[j] = j;
        for (int i = 1; i <= m; i++) {
            cur[0] = i;
            uint8_t pc = p[i - 1];
            for (int j = 1; j <= w; j++) {
                int c = prev[j - 1] + (pc != t[s + j - 1]);
                int d = prev[j] + 1, e = cur[j - 1] + 1;
                cur[j] = c < d ? (c < e ? c : e) : (d < e ? d : e);
            }
            tt = prev; prev = cur; cur = tt;
        }
        int L0 = m - k > 1 ? m - k : 1;
        for (int L = L0; L <= w; L++) {
            if (prev[L] > k) continue;
            int ms = s, me = s + L, md = prev[L];
            if (gopen && ms < gend) {
                if (me > gend) gend = me;
                /* best = min (dist, -(len), start) */
                if (md < bd || (md == bd && (me - ms) > (be - bs))) { bs = ms; be = me; bd = md; }
            } else {
                if (gopen) {
                    if (ngroups == 0) { *fs = bs; *fe = be; }
                    *ls = bs; *le = be;
                    ngroups++;
                }
                gopen = 1; gend = me; bs = ms; be = me; bd = md;
            }
        }
    }
    if (gopen) {
        if (ngroups == 0) { *fs = bs; *fe = be; }
        *ls = bs; *le = be;
        ngroups++;
    }
    free(base);
    return ngroups;
}

int orc_find_near_matches(const uint8_t *p, int m, const uint8_t *t, int n, int k, int *out4) {
    return fnm(p, m, t, n, k, out4, out4 + 1, out4 + 2, out4 + 3);
}

/* ------------------------------------------------------------------------------- */
/* column histogram with first-appearance order (col_base_map)                     */
/* Util.py:9251-9266 (same code :8568 :8900 :9670 :10051 :10359)                   */
/* ------------------------------------------------------------------------------- */
typedef struct {
    int nsym;
    uint8_t sym[MAXSYM];
    int cnt[MAXSYM];
    int gap; /* count of '-' */
} colmap_t;

/* rows: array of R pointers to C-byte rows */
static int build_colmap(const uint8_t **rows, int R, int C, colmap_t *cm) {
    for (int c = 0; c < C; c++) {
        colmap_t *m = &cm[c];
        m->nsym = 0; m->gap = 0;
        for (int r = 0; r < R; r++) {
            uint8_t b = rows[r][c];
            int k;
            for (k = 0; k < m->nsym; k++) if (m->sym[k] == b) break;
            if (k == m->nsym) {
                if (m->nsym == MAXSYM) return ORC_ECAP;
                m->sym[k] = b; m->cnt[k] = 0; m->nsym++;
            }
            m->cnt[k]++;
            if (b == '-') m->gap++;
        }
        /* '-' appended with count 0 if absent: never wins a (count > best) scan, only matters
         * as "exists"; nothing to do. */
    }
    return 0;
}

/* the column histogram as the GPU entry hite_column_vote reports it: counts of A, C, G, T, N, '-' per column, read off the
 * oracle's col_base_map (build_colmap above, Util.py:9251-9266); any other symbol is an error (the product folds it to N on entry) */
int orc_column_vote(const uint8_t *msa, int R, int C, int32_t *out /* C x 6 */) {
    if (R <= 0 || C <= 0) return ORC_EINVAL;
    const uint8_t **rows = (const uint8_t **)malloc(sizeof(uint8_t *) * (size_t)R);
    colmap_t *cm = (colmap_t *)malloc(sizeof(colmap_t) * (size_t)C);
    if (!rows || !cm) { free(rows); free(cm); return ORC_ECAP; }
    for (int r = 0; r < R; r++) rows[r] = msa + (size_t)r * C;
    int rc = build_colmap(rows, R, C, cm);
    for (int c = 0; c < C && rc == 0; c++) {
        for (int k = 0; k < 6; k++) out[(size_t)c * 6 + k] = 0;
        for (int k = 0; k < cm[c].nsym; k++) {
            const char *pos = strchr("ACGTN-", cm[c].sym[k]);
            if (!pos || !cm[c].sym[k]) { rc = ORC_EINVAL; break; }
            out[(size_t)c * 6 + (pos - "ACGTN-")] = cm[c].cnt[k];
        }
    }
    free(rows); free(cm);
    return rc;
}

/* ------------------------------------------------------------------------------- */
/* remove_sparse_col_in_align_file  Util.py:10344-10405                            */
/* keep[c] = 1 if the column survives                                              */
/* ------------------------------------------------------------------------------- */
int orc_sparse_cols(const uint8_t *msa, int R, int C, uint8_t *keep) {
    if (R <= 0 || C <= 0) return ORC_EXC;
    for (int c = 0; c < C; c++) {
        int gap = 0;
        for (int r = 0; r < R; r++) gap += msa[(size_t)r * C + c] == '-';
        /* :10384-10391 column 0 and the last column are never dropped; float compare
         * gap_num > row_num / 2 */
        if (c == 0 || c == C - 1) keep[c] = 1;
        else keep[c] = !((double)gap > (double)R / 2.0);
    }
    return 0;
}

/* ------------------------------------------------------------------------------- */
/* calculate_window_homology  Util.py:8827-8885                                    */
/* window columns are first, first+step, ... (n of them)                           */
/* ------------------------------------------------------------------------------- */
static int window_homology(const uint8_t **rows, int R, int first, int n, int step, double thr) {
    if (n <= 0) return ORC_EXC; /* ZeroDivisionError :8877 (cannot happen from the callers) */
    int *valid = (int *)malloc(sizeof(int) * R);
    int nv = 0;
    for (int r = 0; r < R; r++) {
        int g = 0;
        for (int i = 0, c = first; i < n; i++, c += step) g += rows[r][c] == '-';
        if ((double)g <= (double)n / 2.0) valid[nv++] = r;
    }
    if (nv < 2) { free(valid); return -1; }
    double total = 0.0;
    int first_cand = -1;
    double lim = thr - 0.1;
    for (int i = 0, c = first; i < n; i++, c += step) {
        int cnt[256];
        memset(cnt, 0, sizeof cnt);
        for (int v = 0; v < nv; v++) cnt[rows[valid[v]][c]]++;
        int best = 0;
        for (int b = 0; b < 256; b++) if (b != '-' && cnt[b] > best) best = cnt[b];
        double ratio = best ? (double)best / (double)nv : 0.0;
        if (ratio >= lim && first_cand == -1) first_cand = c;
        total += ratio;
    }
    free(valid);
    double avg = total / (double)n;
    return avg >= thr ? first_cand : -1;
}

int orc_window_homology(const uint8_t *msa, int R, int C, int first, int n, int step, double thr) {
    const uint8_t **rows = (const uint8_t **)malloc(sizeof(void *) * R);
    for (int r = 0; r < R; r++) rows[r] = msa + (size_t)r * C;
    int res = window_homology(rows, R, first, n, step, thr);
    free(rows);
    return res;
}

/* collect up to 100 valid columns scanning from `from` by `dir` while cond holds.
 * mode 0: c < C/2 (float)   mode 1: c >= 0   mode 2: c < C   mode 3: c >= C/2 (float) */
static int scan_valid(const colmap_t *cm, int C, int vthr, int from, int dir, int mode, int *cols) {
    int n = 0, c = from;
    for (;;) {
        if (n >= 100) break;
        int ok;
        switch (mode) {
            case 0: ok = (double)c < (double)C / 2.0; break;
            case 1: ok = c >= 0; break;
            case 2: ok = c < C; break;
            default: ok = (double)c >= (double)C / 2.0; break;
        }
        if (!ok) break;
        if (c < 0 || c >= C) return ORC_EXC; /* KeyError in col_base_map */
        if (cm[c].gap <= vthr) cols[n++] = c;
        c += dir;
    }
    return n;
}

static void reverse_int(int *a, int n) {
    for (int i = 0, j = n - 1; i < j; i++, j--) { int t = a[i]; a[i] = a[j]; a[j] = t; }
}

/* ------------------------------------------------------------------------------- */
/* search_boundary_homo_v3  Util.py:8887-9143                                      */
/* side: 0 = 'start', 1 = 'end'                                                    */
/* ------------------------------------------------------------------------------- */
static int search_v3(const uint8_t **rows, const colmap_t *cm, int R, int C, int vthr, int pos, int side,
                     double thr, int win_in, int win_out) {
    int cols[100];
    int n, ws, nb, cur;
    if (side == 0) {
        n = scan_valid(cm, C, vthr, pos, +1, 0, cols);                      /* :8922-8947 */
        if (n < 0) return n;
        cur = pos; nb = -1;
        ws = n < win_in ? n : win_in;
        if (ws < 10) cur = -1;
        else {
            for (int i = 0; i + ws <= n; i++) {                               /* :8958-8965 */
                int f = cols[i], l = cols[i + ws - 1];
                nb = window_homology(rows, R, f, l - f + 1, +1, thr);
                if (nb != -1) break;
            }
            cur = nb;                                                         /* :8969 */
        }
        n = scan_valid(cm, C, vthr, cur, -1, 1, cols);                      /* :8971-8999 */
        if (n < 0) return n;
        reverse_int(cols, n);
        nb = -1;
        ws = n < win_out ? n : win_out;
        if (ws < 10) cur = -1;
        else {
            for (int i = 0; i + ws <= n; i++) {
                int f = cols[i], l = cols[i + ws - 1];
                nb = window_homology(rows, R, f, l - f + 1, +1, thr);
                if (nb != -1) break;
            }
            if (nb != -1) {
                if (nb < 10) cur = -1;                                        /* :9020 */
                else cur = nb;
            }
        }
        return cur;
    } else {
        n = scan_valid(cm, C, vthr, pos, +1, 2, cols);                      /* :9034-9060 */
        if (n < 0) return n;
        cur = pos;
        reverse_int(cols, n);
        nb = -1;
        ws = n < win_out ? n : win_out;
        if (ws < 10) cur = -1;
        else {
            for (int i = 0; i + ws <= n; i++) {
                int f = cols[i], l = cols[i + ws - 1];
                /* range(first, last + 1, -1): first, first-1, ..., last+2   :9076 */
                int cnt = f - l - 1;
                nb = window_homology(rows, R, f, cnt, -1, thr);
                if (nb < -1) return nb;
                if (nb != -1) break;
            }
            if (nb != -1) {
                if (C - nb < 10) cur = -1;                                    /* :9082 */
                else cur = nb;
            }
        }
        n = scan_valid(cm, C, vthr, cur, -1, 3, cols);                      /* :9091-9120 */
        if (n < 0) return n;
        nb = -1;
        ws = n < win_in ? n : win_in;
        if (ws < 10) cur = -1;
        else {
            for (int i = 0; i + ws <= n; i++) {
                int f = cols[i], l = cols[i + ws - 1];
                int cnt = f - l - 1;                                          /* :9134 */
                nb = window_homology(rows, R, f, cnt, -1, thr);
                if (nb < -1) return nb;
                if (nb != -1) break;
            }
            cur = nb;                                                         /* :9141 */
        }
        return cur;
    }
}

int orc_search_v3(const uint8_t *msa, int R, int C, int pos, int side, double thr, int win_in, int win_out) {
    const uint8_t **rows = (const uint8_t **)malloc(sizeof(void *) * R);
    for (int r = 0; r < R; r++) rows[r] = msa + (size_t)r * C;
    colmap_t *cm = (colmap_t *)malloc(sizeof(colmap_t) * C);
    int rc = build_colmap(rows, R, C, cm);
    if (rc == 0) rc = search_v3(rows, cm, R, C, R / 2, pos, side, thr, win_in, win_out);
    free(cm); free(rows);
    return rc;
}

/* per-column "max_homo_ratio" exactly as the scan loops store it (Util.py:8604-8613):
 * iterate symbols in first-appearance order, skip '-', ratio = cnt / R, track max, stop
 * after the first symbol whose ratio >= thr. */
static double stored_max_ratio(const colmap_t *m, int R, double thr) {
    double mx = 0.0;
    for (int k = 0; k < m->nsym; k++) {
        if (m->sym[k] == '-') continue;
        double r = (double)m->cnt[k] / (double)R;
        if (r > mx) mx = r;
        if (r >= thr) break;
    }
    return mx;
}

/* ------------------------------------------------------------------------------- */
/* search_boundary_homo_v4  Util.py:8556-8824 (Helitron)                           */
/* returns boundary, *valid = first tuple element                                  */
/* ------------------------------------------------------------------------------- */
static int search_v4(const uint8_t **rows, const colmap_t *cm, int R, int C, int vthr, int pos, int side,
                     double thr, double int_thr, double out_thr, int win_in, int win_out, int *valid) {
    int cols[100];
    int n, ws, nb, cur;
    if (side == 0) {
        n = scan_valid(cm, C, vthr, pos, +1, 0, cols);
        if (n < 0) return n;
        cur = pos; nb = -1;
        ws = n < win_in ? n : win_in;
        if (ws < 10) cur = -1;
        else {
            for (int i = 0; i + ws <= n; i++) {
                int f = cols[i], l = cols[i + ws - 1];
                nb = window_homology(rows, R, f, l - f + 1, +1, thr);
                if (nb != -1) break;
            }
            if (nb != cur && nb != -1) cur = nb;                              /* :8634-8637 */
        }
        n = scan_valid(cm, C, vthr, cur - 1, -1, 1, cols);                  /* :8639 */
        if (n < 0) return n;
        reverse_int(cols, n);
        nb = -1;
        ws = n < win_out ? n : win_out;
        if (ws < 10) cur = -1;
        else {
            for (int i = 0; i + ws <= n; i++) {
                int f = cols[i], l = cols[i + ws - 1];
                nb = window_homology(rows, R, f, l - f + 1, +1, thr);
                if (nb != -1) break;
            }
            if (nb != -1) {
                if (nb < 10) cur = -1;
                else cur = nb;
            }
        }
        *valid = 1;
        return cur;
    } else {
        n = scan_valid(cm, C, vthr, pos + 1, +1, 2, cols);                  /* :8704 */
        if (n < 0) return n;
        cur = pos; nb = -1;
        ws = n < win_out ? n : win_out;
        if (ws < 10) { *valid = 0; return -1; }
        {
            double s = 0.0;                                                   /* :8748-8752 (i == 0 only) */
            for (int i = 0; i < ws; i++) s += stored_max_ratio(&cm[cols[i]], R, thr);
            s = s / (double)ws;
            if (s >= out_thr) nb = cols[ws - 1];
        }
        if (nb != -1) {
            if (C - nb < 10) { *valid = 0; return -1; }
            else if (nb != cur) { *valid = 0; return -1; }
        }
        n = scan_valid(cm, C, vthr, pos, -1, 3, cols);                      /* :8769-8798 */
        if (n < 0) return n;
        nb = -1;
        ws = n < win_in ? n : win_in;
        if (ws < 10) { *valid = 0; return -1; }
        {
            double s = 0.0;
            for (int i = 0; i < ws; i++) s += stored_max_ratio(&cm[cols[i]], R, thr);
            s = s / (double)ws;
            if (s < int_thr) nb = cols[ws - 1];
        }
        if (nb != pos && nb != -1) { *valid = 0; return -1; }
        *valid = 1;
        return pos;
    }
}

int orc_search_v4(const uint8_t *msa, int R, int C, int pos, int side, double thr, double int_thr, double out_thr,
                  int win_in, int win_out, int *valid) {
    const uint8_t **rows = (const uint8_t **)malloc(sizeof(void *) * R);
    for (int r = 0; r < R; r++) rows[r] = msa + (size_t)r * C;
    colmap_t *cm = (colmap_t *)malloc(sizeof(colmap_t) * C);
    int rc = build_colmap(rows, R, C, cm);
    *valid = 0;
    if (rc == 0) rc = search_v4(rows, cm, R, C, R / 2, pos, side, thr, int_thr, out_thr, win_in, win_out, valid);
    free(cm); free(rows);
    return rc;
}

/* ------------------------------------------------------------------------------- */
/* get_boundary_ungap_str  Util.py:2407-2429;  TSDsearch_v5  Util.py:2460-2492     */
/* ------------------------------------------------------------------------------- */
static int ungap_str(const uint8_t *row, int C, int pos, int want, int right, uint8_t *out) {
    int n = 0, c = pos;
    if (right) {
        while (n < want && c < C) {
            if (c < 0) { /* python negative index wraps */
                int cc = c + C;
                if (cc < 0) return ORC_EXC;
                if (row[cc] != '-') out[n++] = row[cc];
                c++;
                continue;
            }
            if (row[c] != '-') out[n++] = row[c];
            c++;
        }
    } else {
        uint8_t tmp[32];
        while (n < want && c >= 0) {
            if (c >= C) return ORC_EXC; /* IndexError */
            if (row[c] != '-') tmp[n++] = row[c];
            c--;
        }
        for (int i = 0; i < n; i++) out[i] = tmp[n - 1 - i];
    }
    return n;
}

/* returns TSD length found (0 = none), left/right copied out (<= 11 bytes each) */
int orc_tsd_search_v5(const uint8_t *row, int C, int bs, int be, int plant, uint8_t *left, uint8_t *right) {
    static const int lens[9] = {11, 10, 9, 8, 6, 5, 4, 3, 2};
    uint8_t f5[8], f3[8], l5[8], l3[8], lt[16], rt[16];
    int nf5 = ungap_str(row, C, bs, 5, 1, f5);
    int nf3 = ungap_str(row, C, bs, 3, 1, f3);
    int nl5 = ungap_str(row, C, be, 5, 0, l5);
    int nl3 = ungap_str(row, C, be, 3, 0, l3);
    if (nf5 < 0 || nf3 < 0 || nl5 < 0 || nl3 < 0) return ORC_EXC;
    int found = 0;
    for (int t = 0; t < 9; t++) {
        int k = lens[t];
        int nl = ungap_str(row, C, bs - 1, k, 0, lt);
        int nr = ungap_str(row, C, be + 1, k, 1, rt);
        if (nl < 0 || nr < 0) return ORC_EXC;
        if (nl != nr || nl != k) continue;
        int ok = 0;
        if (memcmp(lt, rt, k) == 0) {
            if (k != 2 && k != 3 && k != 4) ok = 1;
            else if (k == 4) ok = memcmp(lt, "TTAA", 4) == 0;
            else if (k == 2)
                ok = memcmp(lt, "TA", 2) == 0 ||
                     (plant == 0 && nf3 == 3 && memcmp(f3, "CCC", 3) == 0 && nl3 == 3 && memcmp(l3, "GGG", 3) == 0);
            else /* k == 3 */
                ok = memcmp(lt, "TAA", 3) == 0 || memcmp(lt, "TTA", 3) == 0 ||
                     (plant == 1 && nf5 == 5 && nl5 == 5 &&
                      ((memcmp(f5, "CACTA", 5) == 0 && memcmp(l5, "TAGTG", 5) == 0) ||
                       (memcmp(f5, "CACTG", 5) == 0 && memcmp(l5, "CAGTG", 5) == 0)));
        } else if (k >= 8) {
            int mm = 0;                                                        /* allow_mismatch :2281 */
            for (int i = 0; i < k; i++) mm += lt[i] != rt[i];
            ok = mm <= 1;
        }
        if (ok) { /* no break in the reference: the LAST (shortest) match wins */
            found = k;
            memcpy(left, lt, k);
            memcpy(right, rt, k);
        }
    }
    return found;
}

/* ------------------------------------------------------------------------------- */
/* shared front part of judge_boundary_v5 / v9: anchors + full-length rows         */
/* Util.py:9151-9230 / 9490-9569                                                   */
/* ------------------------------------------------------------------------------- */
static int ungap_row(const uint8_t *row, int C, uint8_t *ung, int *reflex) {
    int n = 0;
    for (int c = 0; c < C; c++) if (row[c] != '-') { ung[n] = row[c]; if (reflex) reflex[n] = c; n++; }
    return n;
}

static int all_gap(const uint8_t *row, int64_t lo, int64_t hi) {
    for (int64_t c = lo; c < hi; c++) if (row[c] != '-') return 0;
    return 1; /* all() of an empty sequence is True */
}

/* returns 0 ok, 1 = 'nb' */
static int find_anchor_first_row(const uint8_t *msa, int R, int C, const uint8_t *cand, int clen, int *astart, int *aend) {
    int64_t lo, hi;
    py_slice(0, 20, clen, &lo, &hi);
    const uint8_t *p1 = cand + lo; int m1 = (int)(hi - lo);
    py_slice(-20, clen, clen, &lo, &hi);
    const uint8_t *p2 = cand + lo; int m2 = (int)(hi - lo);
    uint8_t *ung = (uint8_t *)malloc(C + 1);
    int *reflex = (int *)malloc(sizeof(int) * (C + 1));
    *astart = -1; *aend = -1;
    int rc = 1;
    for (int r = 0; r < R; r++) {
        int n = ungap_row(msa + (size_t)r * C, C, ung, reflex);
        int fs, fe, ls, le, gs, ge, hs, he;
        int g1 = fnm(p1, m1, ung, n, 2, &fs, &fe, &ls, &le);
        if (g1 < 0) { rc = g1; break; }
        int g2 = fnm(p2, m2, ung, n, 2, &gs, &ge, &hs, &he);
        if (g2 < 0) { rc = g2; break; }
        if (g1 > 0 && g2 > 0) {
            *astart = reflex[fs];
            *aend = reflex[he - 1];
            rc = 0;
            break;
        }
    }
    free(ung); free(reflex);
    return rc;
}

/* rows with a base within +-alen of both anchors; stops once more than 100 collected */
static int full_length_rows(const uint8_t *msa, int R, int C, int astart, int aend, int alen, const uint8_t **rows) {
    int n = 0;
    for (int r = 0; r < R; r++) {
        if (n > 100) break;
        const uint8_t *row = msa + (size_t)r * C;
        int64_t lo, hi, lo2, hi2;
        int64_t a0 = astart - alen >= 0 ? astart - alen : 0;
        py_slice(a0, (int64_t)astart + alen, C, &lo, &hi);
        int64_t a1 = (aend + alen < C) ? aend + alen : C;
        py_slice((int64_t)aend - alen, a1, C, &lo2, &hi2);
        if (!all_gap(row, lo, hi) && !all_gap(row, lo2, hi2)) rows[n++] = row;
    }
    return n;
}

/* majority consensus of one column  Util.py:9314-9339: returns base, 0 = column skipped.
 * mode 0: v5/v9 (fallback best non-gap)  mode 1: v6 ('N' when no majority) */
static uint8_t cons_col(const colmap_t *m, int R, int mode) {
    int best = 0; uint8_t bb = 0;
    for (int k = 0; k < m->nsym; k++) if (m->cnt[k] > best) { best = m->cnt[k]; bb = m->sym[k]; }
    if (best >= R / 2) return bb != '-' ? bb : 0;
    if (mode == 1) return 'N';
    best = 0; bb = 0;
    for (int k = 0; k < m->nsym; k++) {
        if (m->sym[k] == '-') continue;
        if (m->cnt[k] > best) { best = m->cnt[k]; bb = m->sym[k]; }
    }
    return bb; /* '' if none: nothing appended */
}

static double homo_thr(int R, double big) { return R <= 2 ? 0.95 : (R <= 5 ? 0.9 : big); }
/* judge_boundary_v6 threshold table (literals)  Util.py:9959-9970 */
static double int_thr_tab(int R) { return R <= 2 ? 0.9 : (R <= 5 ? 0.85 : 0.65); }

static int starts_with(const uint8_t *s, int n, const char *p) {
    int k = (int)strlen(p);
    return n >= k && memcmp(s, p, k) == 0;
}
static int ends_with(const uint8_t *s, int n, const char *p) {
    int k = (int)strlen(p);
    return n >= k && memcmp(s + n - k, p, k) == 0;
}

/* ------------------------------------------------------------------------------- */
/* judge_boundary_v5 (TIR)  Util.py:9145-9480, result_type == 'cons'                */
/* info: 0 '', 1 'nb', 2 'fl1'.  returns is_TE (0/1) or error                       */
/* ------------------------------------------------------------------------------- */
int orc_judge_v5(const uint8_t *msa, int R, int C, const uint8_t *cand, int clen, int plant,
                 uint8_t *cons_out, int cons_cap, int *cons_len, int *info, int *row_num, int *bounds /* 2 */) {
    *cons_len = 0; *info = 0; *row_num = 0; bounds[0] = bounds[1] = -1;
    if (R <= 0 || C <= 0) return ORC_EXC;
    int astart, aend;
    int rc = find_anchor_first_row(msa, R, C, cand, clen, &astart, &aend);
    if (rc < 0) return rc;
    if (rc == 1) { *info = 1; return 0; }
    const uint8_t **rows = (const uint8_t **)malloc(sizeof(void *) * (R + 1));
    int rn = full_length_rows(msa, R, C, astart, aend, 10, rows);
    if (rn == 0) { free(rows); return ORC_EXC; } /* IndexError :9219 */
    if (rn <= 1) { free(rows); *info = 2; *row_num = 1; return 0; }
    *row_num = rn;
    colmap_t *cm = (colmap_t *)malloc(sizeof(colmap_t) * C);
    rc = build_colmap(rows, rn, C, cm);
    if (rc < 0) { free(cm); free(rows); return rc; }
    double thr = homo_thr(rn, 0.7);
    int vl = -1, vr = -1;                                                     /* :9269-9294 */
    for (int c = 0; c < C; c++) if ((double)cm[c].gap <= (double)rn / 2.0) { vl = c; break; }
    if (vl != -1) for (int c = C - 1; c >= 0; c--) if ((double)cm[c].gap <= (double)rn / 2.0) { vr = c; break; }
    if (!(vl != -1 && vr != -1 && vl < vr)) { vl = -1; vr = -1; }
    int is_te = 0;
    uint8_t *model = NULL;
    int hs = search_v3(rows, cm, rn, C, rn / 2, astart, 0, thr, 20, 10);
    if (hs < -1) { is_te = hs; goto done; }
    if (hs == -1) goto done;
    int he = search_v3(rows, cm, rn, C, rn / 2, aend, 1, thr, 20, 10);
    if (he < -1) { is_te = he; goto done; }
    if (he == -1) goto done;
    model = (uint8_t *)malloc(C + 1);
    int ml = 0;
    for (int c = hs; c <= he; c++) { uint8_t b = cons_col(&cm[c], rn, 0); if (b) model[ml++] = b; }
    if (hs <= vl || he >= vr) goto done;                                      /* :9353 */
    {
        /* candidate 5' / 3' trims  :9358-9379 */
        int fo[5], nfo = 0, eo[5], neo = 0;
        fo[nfo++] = 0; eo[neo++] = 0;
        if (starts_with(model, ml, "A")) fo[nfo++] = 1;
        if (starts_with(model, ml, "AA") || starts_with(model, ml, "TA")) fo[nfo++] = 2;
        if (starts_with(model, ml, "TAA") || starts_with(model, ml, "TTA")) fo[nfo++] = 3;
        if (starts_with(model, ml, "TTAA")) fo[nfo++] = 4;
        if (ends_with(model, ml, "T")) eo[neo++] = 1;
        if (ends_with(model, ml, "TT") || ends_with(model, ml, "TA")) eo[neo++] = 2;
        if (ends_with(model, ml, "TAA") || ends_with(model, ml, "TTA")) eo[neo++] = 3;
        if (ends_with(model, ml, "TTAA")) eo[neo++] = 4;
        int have = 0, b_ed = 0, b_tc = 0, b_f = 0, b_e = 0, b_s = 0, b_en = 0;
        for (int a = 0; a < nfo; a++) for (int b = 0; b < neo; b++) {
            int cs = hs + fo[a], ce = he - eo[b];
            int tsd = 0;
            for (int r = 0; r < R; r++) {                                    /* ALL rows of the file :9391 */
                const uint8_t *row = msa + (size_t)r * C;
                int i1 = cs, i2 = ce;
                if (i1 < 0) i1 += C;
                if (i2 < 0) i2 += C;
                if (i1 < 0 || i1 >= C || i2 < 0 || i2 >= C) { is_te = ORC_EXC; goto done; }
                if (row[i1] == '-' || row[i2] == '-') continue;
                uint8_t l[16], rr[16];
                int k = orc_tsd_search_v5(row, C, cs, ce, plant, l, rr);
                if (k < 0) { is_te = k; goto done; }
                if (k > 0) tsd++;
            }
            if (tsd > 0) {
                /* first_5bp = model[o:o+5]; end_5bp = model[len-5-o : len-o] (python slices) */
                int64_t lo, hi, lo2, hi2;
                py_slice(fo[a], fo[a] + 5, ml, &lo, &hi);
                if (eo[b] == 0) py_slice(-5, ml, ml, &lo2, &hi2);
                else py_slice((int64_t)ml - 5 - eo[b], (int64_t)ml - eo[b], ml, &lo2, &hi2);
                uint8_t rcv[8];
                int n1 = (int)(hi - lo);
                for (int i = 0; i < n1; i++) rcv[i] = comp_base(model[hi - 1 - i]);
                int ed = lev_dist(rcv, n1, model + lo2, (int)(hi2 - lo2));
                /* stable sort by (edit, -tsd): keep the first minimal */
                if (!have || ed < b_ed || (ed == b_ed && tsd > b_tc)) {
                    have = 1; b_ed = ed; b_tc = tsd; b_f = fo[a]; b_e = eo[b]; b_s = cs; b_en = ce;
                }
            }
        }
        if (have) {
            int64_t lo, hi;
            if (b_e != 0) py_slice(b_f, -(int64_t)b_e, ml, &lo, &hi);
            else py_slice(b_f, ml, ml, &lo, &hi);
            int n = (int)(hi - lo);
            if (n > cons_cap) { is_te = ORC_ECAP; goto done; }
            memcpy(cons_out, model + lo, n);
            *cons_len = n;
            bounds[0] = b_s; bounds[1] = b_en;
            is_te = n > 0;
        }
    }
done:
    free(model); free(cm); free(rows);
    return is_te;
}

/* ------------------------------------------------------------------------------- */
/* find_tail_polyA  Util.py:10832-10837; find_longest_tandem_repeat_tail :9732-9770 */
/* ------------------------------------------------------------------------------- */
int orc_find_tail_polyA(const uint8_t *s, int n) {
    for (int i = n - 5; i >= 0; i--)
        if (i + 6 <= n && memcmp(s + i, "AAAAAA", 6) == 0) return i + 6;
    return -1;
}

int orc_find_tandem_tail(const uint8_t *seq, int n) {
    const int tail_length = 30, min_repeats = 4;
    int tl = n >= tail_length ? tail_length : n;
    const uint8_t *tail = seq + (n - tl);
    int best_end = -1, best_len = 0;
    for (int u = 2; u <= 6; u++) {
        for (int start = tl - u * min_repeats; start >= 0; start--) {
            int rep = 1;
            for (int i = 1; i < (tl - start) / u; i++) {
                if (memcmp(tail + start + i * u, tail + start, u) != 0) break;
                rep++;
            }
            if (rep >= min_repeats) {
                int total = u * rep;
                if (total > best_len) { best_len = total; best_end = n - tl + start + total; }
            }
        }
    }
    return best_end;
}

/* ------------------------------------------------------------------------------- */
/* judge_boundary_v9 (non-LTR)  Util.py:9483-9720                                   */
/* ------------------------------------------------------------------------------- */
int orc_judge_v9(const uint8_t *msa, int R, int C, const uint8_t *cand, int clen, int plant,
                 uint8_t *cons_out, int cons_cap, int *cons_len, int *info, int *row_num, int *bounds) {
    (void)plant;
    *cons_len = 0; *info = 0; *row_num = 0; bounds[0] = bounds[1] = -1;
    if (R <= 0 || C <= 0) return ORC_EXC;
    int astart, aend;
    int rc = find_anchor_first_row(msa, R, C, cand, clen, &astart, &aend);
    if (rc < 0) return rc;
    if (rc == 1) { *info = 1; return 0; }
    const uint8_t **rows = (const uint8_t **)malloc(sizeof(void *) * (R + 1));
    int rn = full_length_rows(msa, R, C, astart, aend, 10, rows);
    if (rn == 0) { free(rows); return ORC_EXC; }
    if (rn <= 1) { free(rows); *info = 2; *row_num = rn; return 0; }
    *row_num = rn;
    colmap_t *cm = (colmap_t *)malloc(sizeof(colmap_t) * C);
    rc = build_colmap(rows, rn, C, cm);
    if (rc < 0) { free(cm); free(rows); return rc; }
    double thr = homo_thr(rn, 0.8);
    int is_te = 0;
    uint8_t *ung = NULL; int *g2n = NULL, *n2g = NULL;
    int hs = search_v3(rows, cm, rn, C, rn / 2, astart, 0, thr, 20, 10);
    if (hs < -1) { is_te = hs; goto done; }
    if (hs == -1) goto done;
    int he = search_v3(rows, cm, rn, C, rn / 2, aend, 1, thr, 20, 10);
    if (he < -1) { is_te = he; goto done; }
    if (he == -1) goto done;
    ung = (uint8_t *)malloc(C + 1);
    g2n = (int *)malloc(sizeof(int) * (C + 1));
    n2g = (int *)malloc(sizeof(int) * (C + 1));
    int cur_bs = hs, tsd_count = 0, have_first = 0;
    for (int r = 0; r < R; r++) {                                            /* :9609-9664 */
        const uint8_t *row = msa + (size_t)r * C;
        int n = 0;
        for (int c = 0; c < C; c++) {
            g2n[c] = n;
            if (row[c] != '-') { ung[n] = row[c]; n2g[n] = c; n++; }
        }
        int end_5 = g2n[cur_bs];
        int h3 = g2n[he];
        int64_t lo, hi;
        py_slice(0, (int64_t)h3 + 10, n, &lo, &hi);
        int sl = (int)hi;
        int end_3 = orc_find_tail_polyA(ung, sl);
        if (end_3 == -1) end_3 = orc_find_tandem_tail(ung, sl);
        if (i64abs((int64_t)end_3 - h3) > 10) continue;
        int found = 0;
        if (end_3 != -1 && end_5 != -1) {
            int left_pos = end_5 - 50 > 0 ? end_5 - 50 : 0;
            int sublen = end_5 - left_pos; /* align_seq[left_pos:end_5], end_5 <= n */
            if (sublen < 0) sublen = 0;
            for (int k = 20; k >= 8 && !found; k--) {
                int64_t tlo, thi;
                py_slice(end_3, (int64_t)end_3 + k, n, &tlo, &thi);
                if (thi - tlo != k) continue; /* k == len(TSD) */
                for (int i = 0; i + k <= sublen; i++) {
                    int o[4];
                    int g = fnm(ung + tlo, k, ung + left_pos + i, k, 1, o, o + 1, o + 2, o + 3);
                    if (g > 0) { end_5 = left_pos + i + k; found = 1; break; }
                }
            }
        }
        if (found) {
            tsd_count++;
            if (!have_first) {
                int fs = end_5 < end_3 ? end_5 : end_3, fe = end_5 < end_3 ? end_3 : end_5;
                int64_t a, b;
                py_slice(fs, fe, n, &a, &b);
                if (b - a > 0) have_first = 1;
                if (fs < 0 || fs >= n) { is_te = ORC_EXC; goto done; }       /* KeyError nogap_to_gap */
                hs = n2g[fs];
            }
        }
    }
    if (tsd_count >= 5 || (double)tsd_count > (double)rn / 2.0) {
        int ml = 0;
        for (int c = hs; c <= he; c++) {
            uint8_t b = cons_col(&cm[c], rn, 0);
            if (b) { if (ml >= cons_cap) { is_te = ORC_ECAP; goto done; } cons_out[ml++] = b; }
        }
        *cons_len = ml;
        bounds[0] = hs; bounds[1] = he;
        is_te = ml >= 80;
    }
done:
    free(ung); free(g2n); free(n2g); free(cm); free(rows);
    return is_te;
}

/* most_common_element  Util.py:9722-9730 (Counter.most_common(1): first-inserted among ties) */
static int most_common(const int *a, int n) {
    if (n <= 0) return -1;
    int best = a[0], bc = 0;
    for (int i = 0; i < n; i++) {
        int seen = 0;
        for (int j = 0; j < i; j++) if (a[j] == a[i]) { seen = 1; break; }
        if (seen) continue;
        int c = 0;
        for (int j = i; j < n; j++) c += a[j] == a[i];
        if (c > bc) { bc = c; best = a[i]; }
    }
    return best;
}

static int find_sub(const uint8_t *s, int n, const char *p, int last) {
    int k = (int)strlen(p), res = -1;
    for (int i = 0; i + k <= n; i++) if (memcmp(s + i, p, k) == 0) { res = i; if (!last) return res; }
    return res;
}

/* ------------------------------------------------------------------------------- */
/* judge_boundary_v6 (Helitron)  Util.py:9821-10159                                 */
/* ------------------------------------------------------------------------------- */
int orc_judge_v6(const uint8_t *msa, int R, int C, const uint8_t *cand, int clen, int plant,
                 uint8_t *cons_out, int cons_cap, int *cons_len, int *info, int *row_num, int *bounds) {
    (void)plant;
    *cons_len = 0; *info = 0; *row_num = 0; bounds[0] = bounds[1] = -1;
    if (R <= 0 || C <= 0) return ORC_EXC;
    int64_t lo, hi;
    py_slice(0, 20, clen, &lo, &hi);
    const uint8_t *p1 = cand + lo; int m1 = (int)(hi - lo);
    py_slice(-20, clen, clen, &lo, &hi);
    const uint8_t *p2 = cand + lo; int m2 = (int)(hi - lo);
    uint8_t *ung = (uint8_t *)malloc(C + 1);
    int *reflex = (int *)malloc(sizeof(int) * (C + 1));
    int *as = (int *)malloc(sizeof(int) * R), *ae = (int *)malloc(sizeof(int) * R), na = 0;
    int is_te = 0;
    const uint8_t **srows = NULL, **erows = NULL, **frows = NULL;
    colmap_t *cm = NULL;
    uint8_t *model = NULL;
    for (int r = 0; r < R; r++) {
        int n = ungap_row(msa + (size_t)r * C, C, ung, reflex);
        int fs, fe, ls, le, gs, ge, hs_, he_;
        int g1 = fnm(p1, m1, ung, n, 2, &fs, &fe, &ls, &le);
        if (g1 < 0) { is_te = g1; goto done; }
        int g2 = fnm(p2, m2, ung, n, 2, &gs, &ge, &hs_, &he_);
        if (g2 < 0) { is_te = g2; goto done; }
        if (g1 > 0 && g2 > 0) { as[na] = reflex[fs]; ae[na] = reflex[he_ - 1]; na++; }
    }
    int astart = most_common(as, na), aend = most_common(ae, na);
    if (astart == -1 || aend == -1) { *info = 1; goto done; }
    srows = (const uint8_t **)malloc(sizeof(void *) * (R + 1));
    erows = (const uint8_t **)malloc(sizeof(void *) * (R + 1));
    frows = (const uint8_t **)malloc(sizeof(void *) * (R + 1));
    int ns = 0, ne = 0;
    for (int r = 0; r < R; r++) {                                            /* :9889-9936 */
        if (ns > 100) break;
        if (ne > 100) break;
        const uint8_t *row = msa + (size_t)r * C;
        int64_t a0 = astart - 1 >= 0 ? astart - 1 : 0, l1, h1;
        py_slice(a0, (int64_t)astart + 1, C, &l1, &h1);
        if (!all_gap(row, l1, h1)) srows[ns++] = row;
        if (aend >= C) { is_te = ORC_EXC; goto done; }                        /* align_seq[align_end] */
        int64_t a1 = (aend + 1 < C) ? aend + 1 : C;
        py_slice((int64_t)aend - 1, a1, C, &l1, &h1);
        if (!all_gap(row, l1, h1)) erows[ne++] = row;
    }
    if (ne <= 0) goto done;
    cm = (colmap_t *)malloc(sizeof(colmap_t) * C);
    int rc = build_colmap(erows, ne, C, cm);
    if (rc < 0) { is_te = rc; goto done; }
    double thr = homo_thr(ne, 0.7);
    int valid = 0;
    int he = search_v4(erows, cm, ne, C, ne / 2, aend, 1, thr, int_thr_tab(ne), thr, 20, 10, &valid);
    if (he < -1) { is_te = he; goto done; }
    if (!valid) goto done;
    if (ns <= 0) { is_te = ORC_EXC; goto done; }
    rc = build_colmap(srows, ns, C, cm);
    if (rc < 0) { is_te = rc; goto done; }
    thr = homo_thr(ns, 0.7);
    int hs = search_v4(srows, cm, ns, C, ns / 2, astart, 0, thr, int_thr_tab(ns), thr, 20, 10, &valid);
    if (hs < -1) { is_te = hs; goto done; }
    int nf = 0;
    for (int r = 0; r < R; r++) {                                            /* :10015-10033 */
        if (nf > 100) break;
        const uint8_t *row = msa + (size_t)r * C;
        int64_t a0 = hs - 1 >= 0 ? hs - 1 : 0, l1, h1, l2, h2;
        py_slice(a0, (int64_t)hs + 1, C, &l1, &h1);
        int64_t a1 = (he + 1 < C) ? he + 1 : C;
        py_slice((int64_t)he - 1, a1, C, &l2, &h2);
        if (!all_gap(row, l1, h1) && !all_gap(row, l2, h2)) frows[nf++] = row;
    }
    if (nf <= 0) goto done;
    *row_num = nf;
    rc = build_colmap(frows, nf, C, cm);
    if (rc < 0) { is_te = rc; goto done; }
    model = (uint8_t *)malloc(C + 4);
    int ml = 0;
    /* leave one slot in front for the 1-bp left extension */
    uint8_t *mbody = model + 1;
    for (int c = hs; c <= he; c++) { uint8_t b = cons_col(&cm[c], nf, 1); if (b) mbody[ml++] = b; }
    uint8_t *mstart = mbody;
    {
        int c = hs - 1, ext = 0;                                              /* :10097-10113 */
        while (ext < 1 && c >= 0) {
            const colmap_t *m = &cm[c];
            int best = 0; uint8_t bb = 0;
            for (int k = 0; k < m->nsym; k++) if (m->cnt[k] > best) { best = m->cnt[k]; bb = m->sym[k]; }
            if (best >= nf / 2 && bb != '-') { mstart = mbody - 1; *mstart = bb; ml++; ext++; }
            c--;
        }
        c = he + 1; ext = 0;
        while (ext < 1 && c < C) {
            const colmap_t *m = &cm[c];
            int best = 0; uint8_t bb = 0;
            for (int k = 0; k < m->nsym; k++) if (m->cnt[k] > best) { best = m->cnt[k]; bb = m->sym[k]; }
            if (best >= nf / 2 && bb != '-') { mstart[ml++] = bb; ext++; }
            c++;
        }
    }
    {
        static const char *motifs[4] = {"CTAGT", "CTAAT", "CTGGT", "CTGAT"};
        const int sl = 10, ext_len = 1;
        int64_t l1, h1, l2, h2;
        py_slice(0, sl, ml, &l1, &h1);
        py_slice(-sl, ml, ml, &l2, &h2);
        for (int t = 0; t < 4; t++) {
            int ei = find_sub(mstart + l2, (int)(h2 - l2), motifs[t], 1);
            if (ei != -1) {
                int si = find_sub(mstart + l1, (int)(h1 - l1), "ATC", 0);
                if (si != -1) {
                    int cut = sl - (ei + 3) - 1;
                    int64_t a, b;
                    if (cut == 0) py_slice(si + 1, ml, ml, &a, &b);
                    else py_slice(si + 1, -(int64_t)cut, ml, &a, &b);
                    int n = (int)(b - a);
                    if (n > cons_cap) { is_te = ORC_ECAP; goto done; }
                    memcpy(cons_out, mstart + a, n);
                    *cons_len = n;
                    bounds[0] = hs - ext_len + si + 1;
                    bounds[1] = he + ext_len - cut;
                    is_te = n > 0;
                    break;
                }
            }
        }
    }
done:
    free(model); free(cm); free(srows); free(erows); free(frows); free(as); free(ae); free(ung); free(reflex);
    return is_te;
}

/* ------------------------------------------------------------------------------- */
/* non-LTR candidate preparation                            */
/* search_polyA_TSD  Util.py:10915-11007; find_nearest_polyA / polyT :10865 / :10903 */
/* (find_longest_polyA / polyT :10840 / :10878), find_nearest_tandem :9772-9820      */
/* ------------------------------------------------------------------------------- */
/* longest run of `base` (>= min_len, first among equals) in seq[lo, hi): returns run as absolute [*s, *e) or 0 */
static int longest_run(const uint8_t *seq, int64_t lo, int64_t hi, uint8_t base, int min_len, int64_t *s, int64_t *e) {
    int64_t best = 0, cur = 0, start = 0, bs = -1, be = -1;
    for (int64_t i = lo; i < hi; i++) {
        if (seq[i] == base) { cur++; if (cur == 1) start = i; }
        else { if (cur >= min_len && cur > best) { best = cur; bs = start; be = i; } cur = 0; }
    }
    if (cur >= min_len && cur > best) { bs = start; be = hi; }
    if (bs < 0) return 0;
    *s = bs - lo; *e = be - lo;   /* coordinates inside the window, as the reference returns them */
    return 1;
}
/* find_nearest_polyA / find_nearest_polyT: -> 1 and absolute [*s, *e), or 0 */
static int nearest_poly(const uint8_t *seq, int64_t n, int64_t pos, uint8_t base, int64_t *s, int64_t *e) {
    int64_t lo, hi, ws, we;
    py_slice(i64max(0, pos - 25), i64min(n, pos + 25), n, &lo, &hi);
    if (!longest_run(seq, lo, hi, base, 6, &ws, &we)) return 0;
    *s = i64max(0, pos - 25 + ws);   /* the offset is pos - 25 even when the window was clamped at 0 (:10872) */
    *e = i64max(0, pos - 25 + we);
    return 1;
}
/* find_nearest_tandem: -> 1 and absolute [*s, *e), or 0 */
static int nearest_tandem(const uint8_t *seq, int64_t n, int64_t pos, int64_t *s, int64_t *e) {
    const int64_t start = i64max(0, pos - 25), end = i64min(n, pos + 25);
    int64_t best = 0;
    int found = 0;
    for (int m = 2; m <= 6; m++)
        for (int64_t i = start; i < end - (int64_t)m * 4 + 1; i++) {
            int64_t lo, hi;
            py_slice(i, i + (int64_t)m * 4, n, &lo, &hi);
            const int64_t L = hi - lo;
            if (L % m != 0) continue;                 /* is_tandem_repeat: length must be a multiple of the motif */
            int ok = 1;
            for (int64_t q = 0; q < L && ok; q += m)
                for (int r = 0; r < m; r++) if (seq[lo + q + r] != seq[lo + r]) { ok = 0; break; }
            if (ok && L > best) { best = L; *s = lo; *e = lo + L; found = 1; }
        }
    return found && best > 0;
}

/*
 * search_polyA_TSD(seq, flanking_len, end_5_window_size, TSD_sizes = 8..20).
 * out[0] = found_TSD, out[1] = direct (0 none, 1 '+', 2 '-'), out[2] = TSD start, out[3] = TSD length (0 if none),
 * out[4], out[5] = [lo, hi) of non_ltr_seq in seq (reverse-complemented by the caller when direct == '-').
 */
void orc_search_polyA_TSD(const uint8_t *seq, int64_t n, int flank, int win5, int64_t *out) {
    const int64_t raw_start = flank + 1, raw_end = n - flank;
    int64_t end_3 = -1, end_5 = -1;
    int direct = 0;
    int64_t ps, pe, ts, te;
    int hp = nearest_poly(seq, n, raw_end, 'A', &ps, &pe), ht = nearest_tandem(seq, n, raw_end, &ts, &te);
    /* sequence[max_start:max_end] can be empty after the clamps: the reference tests the STRING lengths */
    {
        int64_t plen = 0, tlen = 0, lo, hi;
        if (hp) { py_slice(ps, pe, n, &lo, &hi); plen = hi - lo; }
        if (ht) tlen = te - ts;
        if ((plen < tlen ? plen : tlen) > 0) { end_3 = plen > tlen ? pe : te; end_5 = raw_start; direct = 1; }
    }
    hp = nearest_poly(seq, n, raw_start, 'T', &ps, &pe);
    ht = nearest_tandem(seq, n, raw_start, &ts, &te);
    {
        int64_t plen = 0, tlen = 0, lo, hi;
        if (hp) { py_slice(ps, pe, n, &lo, &hi); plen = hi - lo; }
        if (ht) tlen = te - ts;
        if ((plen < tlen ? plen : tlen) > 0) { end_3 = plen > tlen ? ps : ts; end_5 = raw_end; direct = 2; }
    }
    int found = 0;
    int64_t tsd_s = 0, tsd_n = 0;
    if (end_3 != -1 && end_5 != -1 && direct) {
        int64_t wlo, whi;
        py_slice(i64max(0, end_5 - win5), end_5 + win5, n, &wlo, &whi);
        for (int k = 20; k >= 8 && !found; k--) {           /* reversed(TSD_list) */
            int64_t tlo, thi;
            if (direct == 2) py_slice(end_3 - k, end_3, n, &tlo, &thi); else py_slice(end_3, end_3 + k, n, &tlo, &thi);
            if (thi - tlo != k) continue;                    /* k == len(TSD) */
            int hasN = 0;
            for (int64_t q = tlo; q < thi; q++) if (seq[q] == 'N') hasN = 1;
            for (int64_t i = 0; i + k <= whi - wlo; i++) {
                int o4[4];
                /* len(kmer) == k >= 8 -> max_l_dist 1 */
                if (orc_find_near_matches(seq + tlo, k, seq + wlo + i, k, 1, o4) > 0 && !hasN) {
                    end_5 = i64max(0, end_5 - win5) + i + (direct == 1 ? k : 0);
                    found = 1; tsd_s = tlo; tsd_n = k;
                    break;
                }
            }
        }
    }
    out[0] = found; out[1] = direct; out[2] = tsd_s; out[3] = tsd_n;
    if (!direct) { out[4] = 0; out[5] = 0; }
    else { int64_t lo, hi; py_slice(i64min(end_5, end_3), i64max(end_5, end_3), n, &lo, &hi); out[4] = lo; out[5] = hi; }
}

/* ------------------------------------------------------------------------------- */
/* FiLTR get_both_ends_frame + its remove_sparse_col_in_align_file                  */
/* /root/reference/bin/FiLTR-main/src/Util.py:1401-1497, 1341-1399   */
/* ------------------------------------------------------------------------------- */
/* msa R x C (upper-cased rows); cand = the LTR terminal sequence.  frames: R rows of 2*flank bytes (left frame, right
 * frame: the `.matrix` line without its tab); full: R rows with stride 2*flank + C, *full_cols bytes used per row
 * (left frame + cleaned[new_start:new_end] + right frame).  returns 0 ok, 1 boundary not found (None, None),
 * 2 align_start == align_end (the reference then slices with -1: not restated), <0 error. */
int orc_ltr_both_ends(const uint8_t *msa, int R, int C, const uint8_t *cand, int clen, int flank, uint8_t *frames, uint8_t *full,
                      int *full_cols, int *new_start, int *new_end) {
    int astart, aend;
    *full_cols = 0; *new_start = -1; *new_end = -1;
    if (R <= 0 || C <= 0 || clen <= 0) return 1;
    int rc = find_anchor_first_row(msa, R, C, cand, clen, &astart, &aend);
    if (rc < 0) return rc;
    if (rc != 0 || astart == -1 || aend == -1) return 1;
    if (astart == aend) return 2;
    int *inv = (int *)malloc(sizeof(int) * (size_t)(C + 1));
    int K = 0, ns = -1, ne = -1;
    for (int c = 0; c < C; c++) {
        int gap = 0;
        for (int r = 0; r < R; r++) gap += msa[(size_t)r * C + c] == '-';
        if (c == astart) ns = K;                                   /* :1378 */
        else if (c == aend) ne = K;                                /* :1380 */
        else if (2 * gap > R) continue;                            /* gap_num > row_num / 2  :1383 */
        inv[K++] = c;
    }
    const int mid = ne > ns ? ne - ns : 0;
    const int stride = 2 * flank + C;
    for (int r = 0; r < R; r++) {
        const uint8_t *row = msa + (size_t)r * C;
        uint8_t *fr = frames + (size_t)r * 2 * flank, *fu = full + (size_t)r * stride;
        for (int j = 0; j < flank; j++) {
            int k = ns - flank + j;                                /* '-' * (F - len) + row[max(0, ns - F) : ns] */
            fr[j] = k >= 0 ? row[inv[k]] : (uint8_t)'-';
            int k2 = ne + j;                                       /* row[ne : min(len, ne + F)] + '-' * (F - len) */
            fr[flank + j] = k2 < K ? row[inv[k2]] : (uint8_t)'-';
        }
        for (int j = 0; j < flank; j++) fu[j] = fr[j];
        for (int j = 0; j < mid; j++) fu[flank + j] = row[inv[ns + j]];
        for (int j = 0; j < flank; j++) fu[flank + mid + j] = fr[flank + j];
    }
    *full_cols = 2 * flank + mid; *new_start = ns; *new_end = ne;
    free(inv);
    return 0;
}
