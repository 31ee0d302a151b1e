/*
 * hite_oracle_msa.c -- TEST INFRASTRUCTURE ONLY (see hite_oracle.c header).
 *
 * CPU twin of the product's star-alignment stage (hite_amd/csrc/hite_align.hip + hite_msa.hip), which stands
 * where the reference shells out to `mafft` (/root/reference/module/Util.py:10416, third-party, unpinned,
 * absent: PARITY UNPINNED against mafft).  The DEFINITION of the pairwise alignment is band-free and lives in
 * hite_oracle_nw.c (optimal global alignment, mismatch 1, gap GAP = 3 per base, canonical traceback
 * diagonal > up > left).  This file restates WHAT the product's banded aligner computes -- as a plain integer
 * dynamic programme over the same band, where the product runs a bit-parallel recurrence on the differences
 * of neighbouring cells -- so that the product can be checked byte for byte also on the pairs it cannot certify:
 *
 *   Column j (row base b[j-1]) covers the centre rows r = t_j + 1 + k, k = 0 .. W-1, W = 32 NW;  t_0 = -W/2, so
 *   the band's middle starts on row 0.  Rows r <= 0 are virtual (D(r, j) = GAP (j - r): they never help), rows
 *   r > m are padding that never matches.
 *   Steering: the band moves only in the columns j = 1, 5, 9, ... and then by s_j = t_j - t_{j-1} in {0, 4, 8} rows (the
 *   product shifts its registers once per four columns).  With d = (rows whose value exceeds the row above) - (rows
 *   whose value is below the row above) over the middle 64 rows of column j-1, s_j = 0 if d > STEER, 8 if d < -STEER,
 *   else 4; then clamped: t_j <= m - W/2 (the step is cut to a multiple of 4), and large enough that the remaining
 *   moves (8 rows each) can still bring t to within 3 rows below m - W/2: at the end row m sits in the band's
 *   middle, bit W/2 - 1 + (m - W/2 - t_n).  A pair for which that needs a step > 8 is infeasible (status 2).
 *   Band edges are pessimistic: a row that enters at the bottom is GAP above the row before it in the previous
 *   column, the cell above the band's first row is GAP above its left neighbour.  Every band value is therefore
 *   the cost of a real alignment (an upper bound of D), exact whenever an optimal path to the cell stays inside
 *   the band.
 *   Traceback bits per cell: DiagOK = D(i-1, j-1) + sub == D(i, j), UpOK = D(i-1, j) + GAP == D(i, j), left
 *   otherwise: the canonical preference of hite_oracle_nw.c.
 *   The product keeps these bits only for the SLICE = the middle 64 rows of the band (it re-computes the slice
 *   from check points and 2 bytes of boundary information per column); a
 *   traceback that needs a cell outside the slice fails (status 1) and the pair is re-aligned by the wide
 *   fall-back, which keeps the bits of the whole band (`full` below) and fails only if the path leaves the band.
 *   Certificate (Ukkonen): let [LO, HI] be the diagonals i - j that the band covered in EVERY column (edges
 *   that coincide with the matrix border do not constrain).  An alignment of cost <= k has at most k / GAP gaps,
 *   so it stays within the diagonals [min(0, m-n) - e, max(0, m-n) + e], e = floor((k - GAP |m-n|) / (2 GAP)).
 *   With E = the largest e for which that range lies in [LO, HI] and k* = GAP |m-n| + 2 GAP E + 2 GAP - 1:
 *   U <= k*  =>  U is the optimal cost and every cell the canonical traceback consults is exact, i.e. the result
 *   IS the alignment of hite_oracle_nw.c.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EINVAL (-1002)
#define ORC_ECAP (-1001)
#define NWMAX 64
#define GAP 3
#define STEER 24          /* dead zone of the steering */
#define SLICE_WORDS 2     /* rows kept for the traceback (and watched by the steering): the middle 64 of the band */
#define MARGIN 48         /* exact mode: the first wider band is the smallest with 32 NW >= U / GAP + MARGIN */

/* pad bytes (include/hite_gpu.h HITE_ROW_PAD / HITE_IS_ROW_PAD): bit 5 set -- '.', which matches nothing, or a base in lower case */
#define ORC_IS_PAD(c) (((c) & 0x20u) != 0)
static inline int is_acgt(unsigned c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

/* value of row index k (may be -1: the cell above the band, or >= W: rows that have not entered yet) of a stored column */
static inline int32_t col_get(const int32_t *col, int32_t above, int W, int k) {
    if (k < 0) return above;
    if (k >= W) return col[W - 1] + GAP * (k - (W - 1));
    return col[k];
}

/*
 * align row b[0..n) to centre a[0..m) with a band of NW words.  ops: m entries (encoding: hite_oracle_nw.c).
 * full = 0: the traceback may only use the slice (middle 64 rows);  full = 1: the whole band.
 * out[0] = U (cost of the alignment found), out[1] = certified (0/1; a property of the forward pass), out[2] = status
 * (0 ok, 1 traceback left the slice / band, 2 infeasible), out[3] = k* (the certificate's bound, -1 if none).
 * returns 0 or < 0 (bad arguments).
 */
/* Measurement aid (tools/first_exit_certificate.py; not part of the product's definition, not thread-safe): when switched on,
 * orc_bp_pair also computes the FIRST-EXIT bound of the run -- the least cost of any alignment that leaves the band: such a path has a
 * last in-band cell e before its first cell outside; its prefix is an in-band path (cost >= the band's value at e), the step out
 * costs GAP (vertical / horizontal) or >= 0 (diagonal), and from the first cell outside, (r', j'), it still has to make up the
 * diagonal offset, GAP |(m - r') - (n - j')|.  If that bound exceeds U, every optimal alignment stays inside the band, so U is the
 * optimum and the canonical traceback only consults exact cells -- a certificate that needs no second run. */
static int g_want_exit_bound = 0;
static long g_exit_bound = -1;
void orc_bp_exit_bound_enable(int on) { g_want_exit_bound = on; }
long orc_bp_last_exit_bound(void) { return g_exit_bound; }

int orc_bp_pair(const uint8_t *a, int m, const uint8_t *b, int n, int NW, int full, uint16_t *ops, int32_t *out) {
    if (m <= 0 || n <= 0 || m > 32767 || n > 32767 || NW < 4 || NW > NWMAX || (NW & 1)) return ORC_EINVAL;
    const int W = 32 * NW, H = W / 2;
    int32_t *D = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1) * W);   /* band values of every column */
    int32_t *above = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));  /* D(t_j, j): the cell above the band's first row */
    int32_t *ts = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    if (!D || !above || !ts) { free(D); free(above); free(ts); return ORC_EINVAL; }
    int t = -H;
    for (int k = 0; k < W; k++) { int r = t + 1 + k; D[k] = GAP * (r < 0 ? -r : r); }
    above[0] = GAP * H;
    ts[0] = t;
    long LO = -(1L << 40), HI = 1L << 40;
    int status = 0;
    for (int j = 1; j <= n; j++) {
        const int32_t *prev = D + (size_t)(j - 1) * W;
        int32_t *cur = D + (size_t)j * W;
        /* steering from column j-1 */
        int s = 0;
        if ((j & 3) == 1) {
            int dsum = 0;
            for (int k = H - 16 * SLICE_WORDS; k < H + 16 * SLICE_WORDS; k++) {
                int dv = prev[k] - prev[k - 1];
                dsum += (dv > 0) - (dv < 0);
            }
            s = dsum > STEER ? 0 : (dsum < -STEER ? 8 : 4);
            const int tr = m - H;                               /* t never exceeds tr */
            if (t + s > tr) s = (tr - t) & ~3;
            const int need = tr - 3 - 8 * ((n - j) >> 2) - t;   /* the moves still to come cover 8 rows each */
            if (s < need) s = (need + 3) & ~3;
            if (s > 8) { status = 2; break; }
        }
        above[j] = col_get(prev, above[j - 1], W, s - 1) + GAP;
        t += s;
        ts[j] = t;
        if (t >= 1 && (long)t + 1 - j > LO) LO = (long)t + 1 - j;
        if (t + W < m && (long)t + W - j < HI) HI = (long)t + W - j;
        const unsigned y = b[j - 1] & 0xdfu;      /* a row byte in lower case (a pad that carries a base, ORC_ROW_PAD below) is its base */
        const int ya = is_acgt(y);
        for (int k = 0; k < W; k++) {
            const int r = t + 1 + k;
            const int sub = !(ya && r >= 1 && r <= m && a[r - 1] == y);
            int v = col_get(prev, above[j - 1], W, k + s - 1) + sub;
            const int up = (k > 0 ? cur[k - 1] : above[j]) + GAP, left = col_get(prev, above[j - 1], W, k + s) + GAP;
            if (up < v) v = up;
            if (left < v) v = left;
            cur[k] = v;
        }
    }
    int U = -1, cert = 0;
    long kstar = -1;
    g_exit_bound = -1;
    if (status == 0 && g_want_exit_bound) {
        long lb = 1L << 60;
        for (int j = 0; j <= n; j++) {
            const int32_t *cur = D + (size_t)j * W;
            const int tj = ts[j], tn = j < n ? ts[j + 1] : 0;
            for (int k = 0; k < W; k++) {
                const long r = (long)tj + 1 + k;
                if (r < 0 || r > m) continue;                        /* not a cell of the matrix */
                const long F = cur[k];
                long c;
                if (k == W - 1 && r + 1 <= m) {                      /* down, out of the band's last row */
                    c = F + GAP + GAP * labs(((long)m - (r + 1)) - ((long)n - j));
                    if (c < lb) lb = c;
                }
                if (j < n) {
                    if (r <= tn) {                                   /* right, into a row the band has left */
                        c = F + GAP + GAP * labs(((long)m - r) - ((long)n - j - 1));
                        if (c < lb) lb = c;
                    }
                    if (r < m && (r + 1 <= tn || r + 1 > (long)tn + W)) {   /* diagonal, out at the top or at the bottom */
                        c = F + GAP * labs(((long)m - r - 1) - ((long)n - j - 1));
                        if (c < lb) lb = c;
                    }
                }
            }
        }
        g_exit_bound = lb;
    }
    if (status == 0) {
        U = D[(size_t)n * W + (m - ts[n] - 1)];     /* row m: index H - 1 .. H + 2 */
        const long d = (long)m - n, dmin = d < 0 ? d : 0, dmax = d > 0 ? d : 0, ad = d < 0 ? -d : d;
        long E = dmin - LO;
        if (HI - dmax < E) E = HI - dmax;
        if (E > (1L << 27)) { kstar = 0x7fffffff; cert = 1; }
        else if (E >= 0) { kstar = GAP * ad + 2 * GAP * E + 2 * GAP - 1; cert = U <= kstar; }
        /* canonical traceback on the band's values */
        int i = m, j = n;
        const int klo = full ? 0 : H - 16 * SLICE_WORDS, khi = full ? W : H + 16 * SLICE_WORDS;
        while (i > 0 && j > 0) {
            const int k = i - ts[j] - 1, s = ts[j] - ts[j - 1];
            if (k < klo || k >= khi) { status = 1; break; }
            const int32_t *prev = D + (size_t)(j - 1) * W, *cur = D + (size_t)j * W;
            const unsigned y = b[j - 1] & 0xdfu;
            const int sub = !(is_acgt(y) && a[i - 1] == y);
            if (col_get(prev, above[j - 1], W, k + s - 1) + sub == cur[k]) { ops[i - 1] = (uint16_t)(j - 1); i--; j--; }
            else if ((k > 0 ? cur[k - 1] : above[j]) + GAP == cur[k]) { ops[i - 1] = (uint16_t)(j | 0x8000); i--; }
            else j--;
        }
        if (status == 0) for (; i > 0; i--) ops[i - 1] = (uint16_t)0x8000;
    }
    out[0] = U; out[1] = cert; out[2] = status; out[3] = (int32_t)(kstar > 0x7fffffff ? 0x7fffffff : kstar);
    free(D); free(above); free(ts);
    return 0;
}

/*
 * The product's schedule for one pair.  exact_cap = 0 (fast: the 4-word band only) or 8 / 16 / 32 (exact mode: largest
 * band tried for a certificate).
 *   1. band of 4 words.  Infeasible -> the row is dropped.
 *   2. exact mode, not certified, U + GAP MARGIN <= 32 GAP exact_cap: re-run with the smallest band of {8, 16, 32} words
 *      that has 32 GAP NW >= U + GAP MARGIN (U = the cost of the 4-word run), then with each wider one up to exact_cap
 *      until a run is certified.  The alignment kept is that of the last run.
 *   3. traceback on the slice of the run kept; if it leaves the slice: wide fall-back (64 words, whole-band traceback);
 *      if that fails as well the row is dropped.
 * out[0..3] as in orc_bp_pair for the run that produced ops, out[4] = its band words (64 | 0x100 for the fall-back).
 * returns 0, or 1 if the row is dropped.
 */
int orc_align_pair(const uint8_t *a, int m, const uint8_t *b, int n, int exact_cap, uint16_t *ops, int32_t *out) {
    int32_t o[4];
    int nw = 4;
    int rc = orc_bp_pair(a, m, b, n, nw, 0, ops, o);
    if (rc) return rc;
    if (o[2] == 2) { memcpy(out, o, sizeof o); out[4] = nw; return 1; }
    if (exact_cap >= 8 && !o[1] && o[0] + GAP * MARGIN <= 32 * GAP * exact_cap) {
        int lvl = 8;
        while (32 * GAP * lvl < o[0] + GAP * MARGIN) lvl *= 2;
        for (; lvl <= exact_cap; lvl *= 2) {
            rc = orc_bp_pair(a, m, b, n, lvl, 0, ops, o);
            if (rc) return rc;
            nw = lvl;
            if (o[1]) break;
        }
    }
    if (o[2] == 1) {
        nw = 64 | 0x100;
        rc = orc_bp_pair(a, m, b, n, 64, 1, ops, o);
        if (rc) return rc;
    }
    out[0] = o[0]; out[1] = o[1]; out[2] = o[2]; out[3] = o[3]; out[4] = nw;
    return o[2] != 0;
}

static int g_exact = 8;    /* the product's default (hite_align.hip: AL_DEFAULT_CAP / HITE_ALIGN_EXACT) */
void orc_msa_set_exact(int v) { g_exact = v; }
int orc_msa_get_exact(void) { return g_exact; }

/*
 * One candidate: R windows (win + win_off[R+1]); writes cols and, if msa != NULL and cap suffices,
 * the R x cols alignment (row-major).  Rows whose alignment fails at every level are dropped (rows_out < R).
 * Columns: for p = 0..m an insertion block of max_r ins[r][p] columns (bases left-justified, '-' padded)
 * followed (p < m) by the centre column.  Returns 0 or <0.
 */
int orc_star_msa2(const uint8_t *win, const int64_t *win_off, int R, int *cols_out, int *rows_out, uint8_t *msa, int64_t cap) {
    if (R <= 0) return ORC_EINVAL;
    const uint8_t *a = win + win_off[0];
    int m = (int)(win_off[1] - win_off[0]);
    if (m <= 0 || m > 32767) return ORC_EINVAL;
    uint16_t *ops = (uint16_t *)calloc((size_t)R * (m + 1), sizeof(uint16_t));
    int *rowsrc = (int *)malloc(sizeof(int) * (size_t)R);
    if (!ops || !rowsrc) { free(ops); free(rowsrc); return ORC_EINVAL; }
    /* rows that begin / end with pad bytes (ORC_IS_PAD; never the centre): the pads take part in the pairwise
     * alignment -- '.' matches nothing, a lower-case base matches its base -- and leave afterwards -- a centre position aligned to one becomes a gap of the row, the ops
     * are rewritten for the row without them; rb / rn = every row's window as the rest of the stage sees it */
    const uint8_t **rb = (const uint8_t **)malloc(sizeof(uint8_t *) * (size_t)R);
    int *rn = (int *)malloc(sizeof(int) * (size_t)R);
    if (!rb || !rn) { free(ops); free(rowsrc); free(rb); free(rn); return ORC_EINVAL; }
    rb[0] = a; rn[0] = m;
    int K = 0;
    rowsrc[K++] = 0;
    for (int r = 1; r < R; r++) {
        int n = (int)(win_off[r + 1] - win_off[r]);
        if (n <= 0 || n > 32767) { free(ops); free(rowsrc); free(rb); free(rn); return ORC_EINVAL; }
        const uint8_t *b = win + win_off[r];
        int32_t o[5];
        uint16_t *op = ops + (size_t)K * (m + 1);
        int rc = orc_align_pair(a, m, b, n, g_exact, op, o);
        if (rc < 0) { free(ops); free(rowsrc); free(rb); free(rn); return rc; }
        int pf = 0, pb = 0;
        while (pf < n && ORC_IS_PAD(b[pf])) pf++;
        while (pf + pb < n && ORC_IS_PAD(b[n - 1 - pb])) pb++;
        rb[r] = b + pf; rn[r] = n - pf - pb;
        if (rc == 0 && (pf || pb)) {
            for (int p = 0; p < m; p++) {
                const int q = op[p] & 0x7fff;
                if (q < pf) op[p] = 0x8000;
                else if (q >= n - pb) op[p] = (uint16_t)(0x8000 | (n - pf - pb));
                else op[p] = (uint16_t)((op[p] & 0x8000) | (q - pf));
            }
        }
        if (rc == 0) rowsrc[K++] = r;
    }
    /* ins[r][p], gap[r][p] from the ops */
    int *insmax = (int *)calloc(m + 1, sizeof(int));
    int *bstart = (int *)calloc(m + 2, sizeof(int));
    for (int kr = 1; kr < K; kr++) {
        const uint16_t *o = ops + (size_t)kr * (m + 1);
        const int n = rn[rowsrc[kr]];
        int next = 0;
        for (int p = 0; p <= m; p++) {
            int q = p < m ? (o[p] & 0x7fff) : n;
            int ins = q - next;
            if (ins > insmax[p]) insmax[p] = ins;
            if (p < m) next = (o[p] >> 15) ? q : q + 1;
        }
    }
    int c = 0;
    for (int p = 0; p <= m; p++) { bstart[p] = c; c += insmax[p] + (p < m ? 1 : 0); }
    const int C = c;
    *cols_out = C;
    if (rows_out) *rows_out = K;
    if (msa) {
        if ((int64_t)K * C > cap) { free(ops); free(insmax); free(bstart); free(rowsrc); free(rb); free(rn); return ORC_ECAP; }
        memset(msa, '-', (size_t)K * C);
        for (int kr = 0; kr < K; kr++) {
            const uint8_t *b = rb[rowsrc[kr]];
            const int n = rn[rowsrc[kr]];
            uint8_t *row = msa + (size_t)kr * C;
            if (kr == 0) { for (int p = 0; p < m; p++) row[bstart[p] + insmax[p]] = b[p]; continue; }
            const uint16_t *o = ops + (size_t)kr * (m + 1);
            int next = 0;
            for (int p = 0; p <= m; p++) {
                int q = p < m ? (o[p] & 0x7fff) : n;
                for (int x = next; x < q; x++) row[bstart[p] + (x - next)] = b[x];
                if (p < m) {
                    if (o[p] >> 15) next = q;
                    else { row[bstart[p] + insmax[p]] = b[q]; next = q + 1; }
                }
            }
        }
    }
    free(ops); free(insmax); free(bstart); free(rowsrc); free(rb); free(rn);
    return 0;
}

int orc_star_msa(const uint8_t *win, const int64_t *win_off, int R, int *cols_out, uint8_t *msa, int64_t cap) {
    /* historical entry: fails (instead of dropping rows) when a row cannot be aligned */
    int rows = 0;
    int rc = orc_star_msa2(win, win_off, R, cols_out, &rows, NULL, 0);
    if (rc) return rc;
    if (rows != R) return ORC_EINVAL;
    return msa ? orc_star_msa2(win, win_off, R, cols_out, &rows, msa, cap) : 0;
}
