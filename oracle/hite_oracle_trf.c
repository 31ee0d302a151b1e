/*
 * hite_oracle_trf.c -- TEST INFRASTRUCTURE ONLY (see hite_oracle.c header).
 *
 * CPU twin of the build's OWN tandem-repeat masker (hite_amd/csrc/hite_trmask.hip), which stands where the reference
 * shells out to `trf <file> 2 7 7 80 10 50 500 -f -d -m -h` and reads the .mask FASTA (run_remove_TR,
 * /root/reference/module/Util.py:2855-2874; filter_tandem_repeats :4672-4697).  TRF 4.09 is third-party (its binary is
 * bundled with the reference, tools/trf409.linux64, and runs in the build container): PARITY UNPINNED -- what is pinned is
 * HIP == this twin, bit for bit, and both are MEASURED against TRF's own masks (tests/golden/trf_mask.json.gz).
 *
 * Definition (shared with the HIP kernel).  Genome = contigs concatenated, G bases.  For a period p in 1 .. max_period
 * (500, TRF's MaxPeriod) position i MATCHES when i + p < G and bases i and i + p are the same one of A C G T.
 *   Seeds: aligned blocks of 8 positions [s, s + 8), s a multiple of stride(p) (8 for p < 32, 16 for p < 64, else 32), whose
 *   8 positions all match.  A seed is extended only when the block at s - stride(p) is not a seed as well (the leftmost of a run).
 *   A seed is extended when the block at s - stride(p) is not a seed as well or lies before the seed's contig (the leftmost of
 *   a run) or s is a multiple of
 *   TR_RESEED = 2048 (runs longer than one extension are covered piece by piece).
 *   Extension inside the seed's contig [cb, ce): the sequence from s is aligned with ITSELF p bases further on by the banded
 *   end extension of the copy finder (orc_ext_align_scored, hite_oracle_copies.c: unit-cost edit distance in a band of +-8
 *   diagonals, so a copy may be a few bases longer or shorter than its neighbour; the diagonals that would pair a base
 *   with itself, j - i = -p to the right and +p to the left, and those beyond, are excluded), scored S = 2 i - 7 cost (match 2,
 *   mismatch / indel 5; see "calibration" below) -- abandoned 30 below the best, at most TR_MAXEXT = 4096 bases: to the right the query s, s + 1, ...
 *   against the genome from s + p; to the left the query s - 1, s - 2, ... against the genome leftwards from s + p.
 *   il / ir = query bases aligned, tr = genome bases the right extension used, score = S_left + S_right.
 *   The stretch is a tandem array when score + 2 p >= 50 (TRF's Minscore: the first copy counts as matched) and
 *   il + ir >= (85 p + 99) / 100 (at least 1.85 copies: TRF reports nothing below about 1.9).  Positions s - il .. s + p + tr - 1
 *   are masked (clamped to the contig).
 *   Calibration (round 4).  TRF scores every copy against a CONSENSUS pattern with match 2 / mismatch 7 / indel 7; this masker
 *   compares a copy with its NEIGHBOUR, which doubles the divergence: an array whose copies are 15 % from their consensus scores
 *   +0.65 per base in TRF and -0.5 per base neighbour against neighbour under 2 / 7 / 7 -- round 3 lost those arrays (0.80 of the
 *   planted bases, TRF itself 0.94; 0.85 of TRF's own mask).  With the edit penalty at 5 the neighbour score of that array is
 *   +0.6 per base, random sequence stays at -3.25: measured on the three fixtures (TRF 4.09's own masks,
 *   tests/golden/trf_mask.json.gz; sweep of 3 / 4 / 5 / 7 x X-drop 15 / 20 / 30): planted bases 0.944 (TRF 0.936), arrays with
 *   <= 8 % substitutions 0.9985, TRF's own mask covered to 0.988, 72 bases masked outside any planted array per 120 kb (TRF: 30-37;
 *   penalty 4: 150, penalty 3: 349).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TR_XDROP 30
#define TR_MAXEXT 4096
#define TR_MATCH 2
#define TR_MISMATCH 5
#define TR_MINSCORE 50
#define TR_RESEED 2048
#define TR_BAND 8          /* EXT_B of hite_oracle_copies.c */
int64_t orc_ext_align_scored(const uint8_t *qseg, int64_t n, int dir, const uint8_t *genome, int64_t g0, int64_t gmin, int64_t gmax,
                             int sa, int sb, int xdrop, int dlo, int dhi, int64_t *t_out, int64_t *score_out);

static int tr_code(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
static int tr_stride(int p) { return p < 32 ? 8 : (p < 64 ? 16 : 32); }

/* filter match: boundary-agnostic (the kernel reads the packed genome as one string) */
static int fm(const int8_t *code, int64_t G, int64_t i, int p) { return i >= 0 && i + p < G && code[i] >= 0 && code[i] == code[i + p]; }
static int seed_at(const int8_t *code, int64_t G, int64_t s, int p) {
    if (s < 0) return 0;
    for (int k = 0; k < 8; k++) if (!fm(code, G, s + k, p)) return 0;
    return 1;
}
static int contig_of64(const int64_t *coff, int nc, int64_t g) {
    int lo = 0, hi = nc;
    while (hi - lo > 1) { int mid = (lo + hi) / 2; if (coff[mid] <= g) lo = mid; else hi = mid; }
    return lo;
}

/* genome: ASCII, contigs concatenated; mask_out: G bytes, 1 = masked.  Returns the number of masked bases or < 0. */
int64_t orc_tr_mask(const uint8_t *genome, const int64_t *contig_off, int ncontig, int max_period, uint8_t *mask_out) {
    if (ncontig <= 0 || max_period < 1) return -1002;
    const int64_t G = contig_off[ncontig];
    int8_t *code = (int8_t *)malloc((size_t)G + 1);
    for (int64_t i = 0; i < G; i++) code[i] = (int8_t)tr_code(genome[i]);
    memset(mask_out, 0, (size_t)G);
    for (int p = 1; p <= max_period; p++) {
        const int st = tr_stride(p);
        for (int64_t s = 0; s + 8 <= G; s += st) {
            if (!seed_at(code, G, s, p)) continue;
            const int c = contig_of64(contig_off, ncontig, s);
            const int64_t cb = contig_off[c], ce = contig_off[c + 1];
            /* the leftmost seed of a run INSIDE its contig (a run may cross a contig border in the filter's eyes) */
            if (s - st >= cb && seed_at(code, G, s - st, p) && (s % TR_RESEED) != 0) continue;
            if (s + p > ce) continue;                      /* the partner of the seed lies in the next contig */
            int64_t nr = ce - (s + p);                     /* query bases available: the partner s + p + i must stay inside the contig */
            if (nr > TR_MAXEXT) nr = TR_MAXEXT;
            int64_t nl = s - cb;
            if (nl > TR_MAXEXT) nl = TR_MAXEXT;
            if (nr < 0) nr = 0;
            uint8_t *seg = (uint8_t *)malloc((size_t)(nl > nr ? nl : nr) + 1);
            int64_t tr = 0, tl = 0, sr = 0, sl = 0;
            for (int64_t x = 0; x < nr; x++) seg[x] = genome[s + x];
            /* diagonals that would align the sequence with itself (j - i = -p to the right, +p to the left) stay out of reach */
            const int lim = p - 1 < TR_BAND ? p - 1 : TR_BAND;
            int64_t ir = orc_ext_align_scored(seg, nr, +1, genome, s + p, cb, ce, TR_MATCH, TR_MATCH + TR_MISMATCH, TR_XDROP, -lim, TR_BAND, &tr, &sr);
            for (int64_t x = 0; x < nl; x++) seg[x] = genome[s - 1 - x];
            int64_t il = orc_ext_align_scored(seg, nl, -1, genome, s + p, cb, ce, TR_MATCH, TR_MATCH + TR_MISMATCH, TR_XDROP, -TR_BAND, lim, &tl, &sl);
            free(seg);
            (void)tl;
            if (il + ir <= 0) continue;
            if (sl + sr + 2 * p < TR_MINSCORE) continue;
            if (il + ir < (85 * (int64_t)p + 99) / 100) continue;
            int64_t a = s - il, e = s + tr - 1;
            int64_t hi = e + p;
            if (hi >= ce) hi = ce - 1;
            for (int64_t i = a; i <= hi; i++) mask_out[i] = 1;
        }
    }
    int64_t n = 0;
    for (int64_t i = 0; i < G; i++) n += mask_out[i];
    free(code);
    return n;
}
