/*
 * hite_oracle_msa.c -- TEST INFRASTRUCTURE ONLY (see hite_oracle.c header).
 *
 * CPU twin of the build's OWN star-alignment stage (hite_amd/csrc/hite_msa.hip), which stands
 * where the reference shells out to `mafft --preservecase --quiet --thread 1`
 * (/root/reference/module/Util.py:10416).  mafft is third-party, unpinned
 * (environment.yml:30) and absent from this image: PARITY UNPINNED at that boundary
 * (SURVEY.md 8c).  What is pinned is: HIP output == this twin, byte for byte, and everything
 * downstream of the gapped matrix == the reference (hite_oracle.c).
 *
 * Definition (shared with the HIP kernels)
 *   centre = row 0 of the candidate; every other row is aligned to it by global
 *   Needleman-Wunsch, match +2 (equal and one of A, C, G, T), mismatch -2, linear gap -4, restricted to an
 *   adaptive band of W=64 cells per anti-diagonal s=i+j (rows i in [t, t+63]):
 *     - t(0) = -32; after anti-diagonal s the band moves right (t same) if H[lane0] > H[lane63],
 *       down (t+1) if H[lane0] < H[lane63], on a tie down when s is even else right;
 *     - then forced: t+1 only if t+1 <= min(m,s+1)-31, and t+1 if t < max(0,s+1-n)-32;
 *   every lane of the band evaluates the same recurrence every step: bases outside the sequences
 *   are sentinels that never match (so cells outside the matrix only ever hold "junk" that
 *   cannot beat a real score), values shifted in from outside the band are 0 and real scores
 *   are biased by 2^28 (H(0,0) = 2^28); 'N' (any byte other than A, C, G, T) never matches.
 *   Stored form (part of the definition, it fixes what the junk cells hold): H is shifted by +4
 *   per anti-diagonal (the recurrence adds +10 / +6 / 0 for match / mismatch / gap), scaled by 4,
 *   and the low two bits carry the winning operand: V = 4 H + tag, candidates
 *   diag = D + 4*score + 2, up = U (stored with tag 1), left = L - 1 (tag 0), V = max of the
 *   three (so ties go diag >= up >= left), direction = V & 3, and (V & ~3) | 1 is stored.  The 0
 *   that enters at a band edge is not transformed (left operand: 0 - 1).  The alignment
 *   fails if H(m,n) <= 2^27 (end cell not reachable inside the band).
 *   The diagonal operand is carried in ONE history register H(s-2)' = H(s-2) re-aligned to the
 *   origin of anti-diagonal s-1, which is simply the previous step's "left" operand; an element
 *   that was shifted out of the band by that re-alignment is gone (0) even if a later move
 *   would shift it back in.
 *   Ties in the recurrence: diag >= up >= left.
 *   Traceback from (m,n) gives per centre position p: gap flag (row has '-') and the number of
 *   row bases inserted before p.  Columns: for p = 0..m an insertion block of
 *   max_r ins[r][p] columns (bases left-justified, '-' padded) followed (p < m) by the centre
 *   column.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EINVAL (-1002)
#define ORC_ECAP (-1001)
#define W 64
#define BIAS (1 << 28)
/* stored scores: shifted by +4 per anti-diagonal ((+2, -2, -4) + (8, 8, 4) = 10, 6, 0), times 4, + tag 2 */
#define SC_MATCH 42
#define SC_MIS 26

/* align row b[0..n) to centre a[0..m); ops[p] (p = 0..m): low 15 bits = insertions before p,
 * bit 15 = row has a gap at centre position p.  returns 0 or <0. */
static int pair_align(const uint8_t *a, int m, const uint8_t *b, int n, uint16_t *ops) {
    int steps = m + n;
    /* direction codes per (s, lane) and the band origin t per s */
    uint8_t *dir = (uint8_t *)malloc((size_t)(steps + 1) * W);
    int *ts = (int *)malloc(sizeof(int) * (steps + 1));
    int prev[W], cur[W];
    if (!dir || !ts) { free(dir); free(ts); return ORC_EINVAL; }
    int t = -32;
    int ppal[W]; /* H(s-2) re-aligned to the origin of anti-diagonal s-1 (see header) */
    for (int k = 0; k < W; k++) { prev[k] = 0; ppal[k] = 0; }
    prev[32] = (BIAS << 2) | 1; /* H(0,0), stored form */
    ts[0] = t;
    for (int s = 1; s <= steps; s++) {
        /* choose the move from anti-diagonal s-1 (held in prev, origin t) */
        int h0 = prev[0], h63 = prev[W - 1];
        int move;
        if (h0 > h63) move = 0; else if (h0 < h63) move = 1; else move = (s & 1) ? 1 : 0;
        int lo = s - n > 0 ? s - n : 0, hi = m < s ? m : s;
        int tn = t + move;
        if (tn > hi - 31) tn = t;
        if (tn < lo - 32) tn = t + 1;
        int down = tn != t;
        int hlv[W];
        for (int k = 0; k < W; k++) {
            int i = tn + k, j = s - i;
            /* down : (left, up, diag) = (H(s-1)[k+1], H(s-1)[k],   H(s-2)'[k])
             * right: (left, up, diag) = (H(s-1)[k],   H(s-1)[k-1], H(s-2)'[k-1])   0 from outside the band */
            int hl = (down ? (k + 1 < W ? prev[k + 1] : 0) : prev[k]) - 1;
            int hu = down ? prev[k] : (k >= 1 ? prev[k - 1] : 0);
            int hd = down ? ppal[k] : (k >= 1 ? ppal[k - 1] : 0);
            int x = (i >= 1 && i <= m) ? a[i - 1] : 0xFF;
            int y = (j >= 1 && j <= n) ? b[j - 1] : 0xFE;
            int cd = hd + ((x == y && (x == 'A' || x == 'C' || x == 'G' || x == 'T')) ? SC_MATCH : SC_MIS);
            int v4 = cd > hu ? cd : hu, v, d;
            if (hl > v4) v4 = hl;
            d = (v4 & 2) ? 0 : ((v4 & 1) ? 1 : 2);   /* junk cells may carry any tag; they are never on the path */
            v = (v4 & ~3) | 1;
            cur[k] = v;
            hlv[k] = hl;
            dir[(size_t)s * W + k] = (uint8_t)d;
        }
        memcpy(ppal, hlv, sizeof hlv); /* this step's "left" operand is the next step's re-aligned H(s-2) */
        memcpy(prev, cur, sizeof cur);
        t = tn;
        ts[s] = t;
    }
    {
        int kf = m - t;
        if (kf < 0 || kf >= W || prev[kf] <= (BIAS / 2) * 4) { free(dir); free(ts); return ORC_EINVAL; }
    }
    /* traceback */
    int i = m, j = n, cur_ins = 0, pend_gap = 0;
    while (i > 0 || j > 0) {
        int s = i + j;
        int k = i - ts[s];
        int d;
        if (k < 0 || k >= W) { free(dir); free(ts); return ORC_EINVAL; }
        d = dir[(size_t)s * W + k];
        if (i == 0) d = 2; else if (j == 0) d = 1; /* only possible moves on the borders */
        if (d == 2) { cur_ins++; j--; }
        else {
            ops[i] = (uint16_t)((cur_ins > 0x7fff ? 0x7fff : cur_ins) | (pend_gap << 15));
            pend_gap = d == 1;
            cur_ins = 0;
            i--;
            if (d == 0) j--;
        }
    }
    ops[0] = (uint16_t)((cur_ins > 0x7fff ? 0x7fff : cur_ins) | (pend_gap << 15));
    free(dir); free(ts);
    return 0;
}

/*
 * One candidate: R windows (win + win_off[R+1]); writes cols and, if msa != NULL and cap suffices,
 * the R x cols alignment (row-major).  Returns 0 or <0.
 */
int orc_star_msa(const uint8_t *win, const int64_t *win_off, int R, int *cols_out, uint8_t *msa, int64_t cap) {
    if (R <= 0) return ORC_EINVAL;
    const uint8_t *a = win + win_off[0];
    int m = (int)(win_off[1] - win_off[0]);
    if (m <= 0) return ORC_EINVAL;
    uint16_t *ops = (uint16_t *)calloc((size_t)R * (m + 1), sizeof(uint16_t));
    if (!ops) return ORC_EINVAL;
    for (int r = 1; r < R; r++) {
        int n = (int)(win_off[r + 1] - win_off[r]);
        if (n <= 0) { free(ops); return ORC_EINVAL; }
        int rc = pair_align(a, m, win + win_off[r], n, ops + (size_t)r * (m + 1));
        if (rc) { free(ops); return rc; }
    }
    int *insmax = (int *)calloc(m + 1, sizeof(int));
    int *bstart = (int *)calloc(m + 2, sizeof(int));
    for (int r = 0; r < R; r++)
        for (int p = 0; p <= m; p++) {
            int v = ops[(size_t)r * (m + 1) + p] & 0x7fff;
            if (v > insmax[p]) insmax[p] = v;
        }
    int c = 0;
    for (int p = 0; p <= m; p++) { bstart[p] = c; c += insmax[p] + (p < m ? 1 : 0); }
    int C = c;
    *cols_out = C;
    if (msa) {
        if ((int64_t)R * C > cap) { free(ops); free(insmax); free(bstart); return ORC_ECAP; }
        memset(msa, '-', (size_t)R * C);
        for (int r = 0; r < R; r++) {
            const uint8_t *b = win + win_off[r];
            uint8_t *row = msa + (size_t)r * C;
            int rp = 0;
            for (int p = 0; p <= m; p++) {
                uint16_t o = ops[(size_t)r * (m + 1) + p];
                int ins = o & 0x7fff, gap = o >> 15;
                for (int q = 0; q < ins; q++) row[bstart[p] + q] = b[rp + q];
                rp += ins;
                if (p < m && !gap) { row[bstart[p] + insmax[p]] = b[rp]; rp++; }
            }
        }
    }
    free(ops); free(insmax); free(bstart);
    return 0;
}
