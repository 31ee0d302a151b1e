/*
 * hite_oracle_nw.c -- TEST INFRASTRUCTURE ONLY (see hite_oracle.c header).
 *
 * THE DEFINITION of the pairwise alignment of the star-alignment stage (hite_amd/csrc/hite_align.hip), which
 * stands where the reference shells out to `mafft --preservecase --quiet --thread 1`
 * (/root/reference/module/Util.py:10416; mafft is third-party, unpinned -- environment.yml:30 -- and absent
 * from this image: PARITY UNPINNED against mafft itself).  What the build promises instead is
 * anchored to a textbook, band-free definition:
 *
 *   every row b[0..n) is aligned to the centre a[0..m) by the optimal GLOBAL alignment that minimises
 *   mismatches + 3 x (inserted + deleted bases)   (match 0, mismatch 1, linear gap ORC_GAP = 3 per base -- the same
 *   optimum as Needleman-Wunsch with match +2, mismatch -1, gap -8, since 2 (matches + mismatches) + gaps = m + n;
 *   the ratio was chosen by measurement: tools/align_cost_sweep.py);
 *   two bases match iff they are equal and one of A, C, G, T ('N' and every other byte never match); a ROW byte in lower case
 *   (a, c, g, t: the pads of hite_flank_region_align_clip, never a genome base) stands for its base;
 *   among co-optimal alignments the canonical one is the traceback from (m, n) that prefers
 *   diagonal, then up (centre base against a gap), then left (row base inserted).
 *
 * orc_nw_pair below is the plain full-matrix dynamic programme for that definition: no band, no heuristic.
 * The product's banded bit-parallel aligner (twin: hite_oracle_msa.c) must produce exactly this alignment
 * whenever it reports a pair as certified, and an alignment of exactly this cost whenever it is optimal.
 *
 * ops encoding (shared with the product): ops[p], p = 0..m-1:
 *   q           centre position p is aligned to row position q           (diagonal step)
 *   q | 0x8000  centre position p faces a gap; q = next row position     (up step)
 * row bases not named by any ops[p] are insertions (they sit before the next aligned centre position).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EINVAL (-1002)
#define ORC_GAP 3

static inline int is_acgt(unsigned c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

/* cost of the optimal alignment only, two rolling rows (any size) */
int orc_nw_distance(const uint8_t *a, int m, const uint8_t *b, int n) {
    if (m < 0 || n < 0) return ORC_EINVAL;
    int *prev = (int *)malloc(sizeof(int) * (size_t)(n + 1)), *cur = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    if (!prev || !cur) { free(prev); free(cur); return ORC_EINVAL; }
    for (int j = 0; j <= n; j++) prev[j] = ORC_GAP * j;
    for (int i = 1; i <= m; i++) {
        cur[0] = ORC_GAP * i;
        const unsigned x = a[i - 1];
        const int xa = is_acgt(x);
        for (int j = 1; j <= n; j++) {
            int sub = !(xa && x == (b[j - 1] & 0xdfu));
            int v = prev[j - 1] + sub;
            if (prev[j] + ORC_GAP < v) v = prev[j] + ORC_GAP;
            if (cur[j - 1] + ORC_GAP < v) v = cur[j - 1] + ORC_GAP;
            cur[j] = v;
        }
        int *t = prev; prev = cur; cur = t;
    }
    int d = prev[n];
    free(prev); free(cur);
    return d;
}

int orc_nw_pair_cost(const uint8_t *a, int m, const uint8_t *b, int n, int mis, int gap, uint16_t *ops);

/* full matrix + canonical traceback; ops has m entries (m + 1 allowed); returns the cost or < 0 */
int orc_nw_pair(const uint8_t *a, int m, const uint8_t *b, int n, uint16_t *ops) {
    return orc_nw_pair_cost(a, m, b, n, 1, ORC_GAP, ops);
}

/* cost of the alignment that `ops` encodes (checks that it is a valid monotone alignment of a and b); < 0 if not */
int orc_ops_cost(const uint8_t *a, int m, const uint8_t *b, int n, const uint16_t *ops) {
    int next = 0, cost = 0;   /* next = first row position not yet consumed */
    for (int p = 0; p < m; p++) {
        const int q = ops[p] & 0x7fff, gap = ops[p] >> 15;
        if (q < next || q > n) return ORC_EINVAL;
        cost += ORC_GAP * (q - next);        /* inserted row bases before p */
        if (gap) { cost += ORC_GAP; next = q; }
        else {
            if (q >= n) return ORC_EINVAL;
            const unsigned x = a[p];
            cost += !(is_acgt(x) && x == (b[q] & 0xdfu));
            next = q + 1;
        }
    }
    cost += ORC_GAP * (n - next);            /* insertions after the last centre position */
    return cost;
}

/* the full-matrix programme with general costs (mismatch `mis`, gap `gap` per base; match 0) and the canonical traceback;
 * other costs than (1, ORC_GAP) are a test utility for comparing scoring schemes.  Returns the cost or < 0. */
int orc_nw_pair_cost(const uint8_t *a, int m, const uint8_t *b, int n, int mis, int gap, uint16_t *ops) {
    if (m <= 0 || n <= 0 || m > 32767 || n > 32767 || mis < 0 || gap < 0) return ORC_EINVAL;
    if ((int64_t)(m + 1) * (n + 1) > ((int64_t)1 << 29)) return ORC_EINVAL;
    const size_t ld = (size_t)n + 1;
    int32_t *D = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m + 1) * ld);
    if (!D) return ORC_EINVAL;
    for (int j = 0; j <= n; j++) D[j] = j * gap;
    for (int i = 1; i <= m; i++) {
        int32_t *row = D + (size_t)i * ld;
        const int32_t *up = row - ld;
        row[0] = i * gap;
        const unsigned x = a[i - 1];
        const int xa = is_acgt(x);
        for (int j = 1; j <= n; j++) {
            int sub = (xa && x == (b[j - 1] & 0xdfu)) ? 0 : mis;
            int v = up[j - 1] + sub;
            if (up[j] + gap < v) v = up[j] + gap;
            if (row[j - 1] + gap < v) v = row[j - 1] + gap;
            row[j] = v;
        }
    }
    const int dist = D[(size_t)m * ld + n];
    int i = m, j = n;
    while (i > 0) {
        const int32_t *row = D + (size_t)i * ld;
        const int32_t *up = row - ld;
        if (j > 0) {
            const unsigned x = a[i - 1];
            int sub = (is_acgt(x) && x == (b[j - 1] & 0xdfu)) ? 0 : mis;
            if (up[j - 1] + sub == row[j]) { ops[i - 1] = (uint16_t)(j - 1); i--; j--; continue; }
            if (up[j] + gap == row[j]) { ops[i - 1] = (uint16_t)(j | 0x8000); i--; continue; }
            j--;
        } else {
            ops[i - 1] = (uint16_t)0x8000;
            i--;
        }
    }
    free(D);
    return dist;
}
