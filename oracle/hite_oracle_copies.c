/*
 * hite_oracle_copies.c -- TEST INFRASTRUCTURE ONLY (see hite_oracle.c header).
 *
 * CPU twin of the build's OWN copy-finding stage (hite_amd/csrc/hite_copies.hip), which stands where
 * the reference shells out to `minimap2 -ax map-ont -N 300 -p 0.2` and filters the SAM records
 * (get_full_length_copies_minimap2 / get_copies_minimap2, /root/reference/module/Util.py:7933-8030).
 * minimap2 2.28 is third-party and absent from this image: PARITY UNPINNED at that boundary; what is
 * pinned is HIP output == this twin, record for record.
 *
 * Definition (shared with the HIP kernels)
 *   k-mer code (K=15): x = sum code(base_i) << 2i, A0 C1 G2 T3; rc = rev2(x ^ 0x3fffffff);
 *   canonical = min(x, rc), strand = rc < x; hs = (lowbias32(canonical) & ~1) | strand, values
 *   >= 0xfffffffe lowered by 2; a k-mer with a non-ACGT base or crossing a sequence end is invalid.
 *   Minimizer of a window of W=10 consecutive k-mer starts = the valid k-mer with the smallest
 *   (hs >> 1, position); the minimizers of a sequence are the distinct window minimizers
 *   (sequences with fewer than W k-mers form one window).
 *   Index = genome minimizers sorted by (hs, position).  For every candidate minimizer the index
 *   (of those kept by the sampling rule, cand_minimizer_kept below) the index
 *   entries with the same hs >> 1 are its occurrences (skipped when more than MAXOCC = 2000);
 *   an occurrence gives a hit: rel = strand_q ^ strand_g, qo = rel ? Lq - qpos - K : qpos,
 *   d = gpos - qo.  Hits are sorted by (candidate, rel, d); a new cluster starts when candidate, rel
 *   or the contig of gpos changes or d jumps by more than TD = 64.  Extreme anchors of a cluster: (qlo, glo) = the
 *   smallest qo (ties: smallest gpos), (qhi, ghi) = the largest qo (ties: largest gpos).  A cluster with >= 3 anchors whose
 *   query span qhi + K - qlo is >= 95 % of its genome span ghi + K - glo, and which leaves at most 2048 bases of the candidate
 *   beyond either extreme anchor (a stretch of 2048 bases without one shared minimizer: ~370 of them in a row), is a CHAIN
 *   (the cap removes 42 % of the extension work -- the two LTRs of an element matched crosswise -- at no measurable loss of
 *   recall).  A chain is extended base by base from
 *   its extreme anchors to both ends of the candidate (ext_align below: unit-cost edit distance in a band of +-8 diagonals,
 *   cut where the score i - 3 cost is largest, abandoned 40 below the best score) -- the stand-in for minimap2's end
 *   extension and soft clipping.  aligned = Lq - (clipped bases of both ends); the aligned part covers the genome interval
 *   a0 = glo - (genome bases of the left extension) .. a1 = ghi + K + (genome bases of the right one).  The chain is kept
 *   when aligned >= 95 % of Lq and aligned >= 95 % of a1 - a0: get_copies_minimap2's two filters,
 *   query_alignment_length / len(query) >= 0.95 and query_alignment_length / (M + D) >= 0.95 (Util.py:8008-8022).
 *   The record carries the ALIGNED interval, reference_start + 1 .. reference_end as the reference reports it (Util.py:8026): a0 + 1 ..
 *   a1 -- the default since round 5, with the clipped candidate bases beside it (orc_find_copies_clips: the rows of the star alignment
 *   are padded by them, hite_flank_region_align_clip) --, or (orc_find_copies_config(0), the default of rounds 2-4) the interval of
 *   the WHOLE candidate: start0 = a0 - clipped_left, end0 = a1 + clipped_right, clamped to the contig.  History of the switch: with
 *   bare aligned windows a candidate whose ends overhang the element loses its 20-bp anchors in every row but its own and every row
 *   pays 3 per clipped base in the global alignment (300 candidates with ends perturbed by +-30 bp: 152 TE calls, 51 with both ends
 *   exact, against 209 / 127 on whole-candidate intervals); with the rows padded by the clipped bases the aligned intervals call as
 *   many TEs as the whole-candidate ones and put more consensus ends on the element (600 candidates: 388 / 248 against 381 / 234).
 *   (Round 2 accepted "anchor span >= 80 % of the candidate" with extrapolated ends: recall of the planted full-length copies
 *   0.62 -> 0.84, of those within 15 % of the candidate 0.81 -> 0.94, precision 1.000, on a 20 Mbp / 100 family sample.)
 *   Per candidate the copies are ordered by (anchors descending (capped at 4095), start ascending) and the first 300 are kept.
 */
#define _POSIX_C_SOURCE 199309L
#include <stdint.h>
#include <math.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EINVAL (-1002)
#define ORC_ECAP (-1001)
#define CK 15
#define CW 10
#define MAXOCC 2000
#define TD 64
#define MINANCH 3
#define MAXCOPY 300
#define HS_INVALID 0xffffffffu

static uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
static uint32_t rev2_k(uint32_t x, int K) { /* reverse the order of the K 2-bit groups of a 2K-bit value */
    uint32_t y = 0;
    for (int i = 0; i < K; i++) y |= ((x >> (2 * i)) & 3u) << (2 * (K - 1 - i));
    return y;
}
static int code_of(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }

/* hs of the K-mer starting at p of seq[0..L) */
static uint32_t kmer_hs(const uint8_t *seq, int64_t L, int64_t p, int K) {
    if (p < 0 || p + K > L) return HS_INVALID;
    uint32_t x = 0;
    for (int i = 0; i < K; i++) {
        int c = code_of(seq[p + i]);
        if (c < 0) return HS_INVALID;
        x |= (uint32_t)c << (2 * i);
    }
    uint32_t rc = rev2_k(x ^ ((1u << (2 * K)) - 1u), K);
    uint32_t can = x < rc ? x : rc;
    uint32_t strand = rc < x ? 1u : 0u;
    uint32_t hs = (lowbias32(can) & ~1u) | strand;
    if (hs >= 0xfffffffeu) hs -= 2;
    return hs;
}

typedef struct { uint32_t hs; int64_t pos; } mini_t;

/* (W, K) minimizers of seq[0..L); pos offset added; returns count (out may be NULL to count) */
static int64_t minimizers_kw(const uint8_t *seq, int64_t L, int64_t pos_off, mini_t *out, int K, int W) {
    int64_t nk = L - K + 1;
    if (nk <= 0) return 0;
    uint32_t *hs = (uint32_t *)malloc(sizeof(uint32_t) * nk);
    for (int64_t p = 0; p < nk; p++) hs[p] = kmer_hs(seq, L, p, K);
    int64_t nwin = nk >= W ? nk - W + 1 : 1;
    int64_t n = 0, last = -1;
    for (int64_t p = 0; p < nwin; p++) {
        int64_t hi = p + W < nk ? p + W : nk;
        int64_t best = -1;
        for (int64_t i = p; i < hi; i++) {
            if (hs[i] == HS_INVALID) continue;
            if (best < 0 || (hs[i] >> 1) < (hs[best] >> 1)) best = i;
        }
        if (best >= 0 && best != last) {
            if (out) { out[n].hs = hs[best]; out[n].pos = pos_off + best; }
            n++;
            last = best;
        }
    }
    free(hs);
    return n;
}
static int64_t minimizers(const uint8_t *seq, int64_t L, int64_t pos_off, mini_t *out) { return minimizers_kw(seq, L, pos_off, out, CK, CW); }

static int cmp_mini(const void *a, const void *b) {
    const mini_t *x = (const mini_t *)a, *y = (const mini_t *)b;
    if (x->hs != y->hs) return x->hs < y->hs ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return 0;
}

/* Sampling of the CANDIDATE's minimizers (the index keeps every genome minimizer): a chain needs three anchors and is extended
 * base by base, so the hundreds of anchors a long candidate has with each of its copies only feed the sort.  All minimizers
 * within SUB_EDGE bases of either end are kept (the extension starts at the outermost anchors); of the interior, candidates of
 * >= 2 * SUB_UNIT bases keep those with (hs >> 1) % S == 0, S = min(SUB_MAX, Lq / SUB_UNIT).  Measured (40 Mbp, 100 TIR + 100
 * LTR families): hits -57 %, recall of copies within 15 % of the candidate unchanged (0.940), over all pairs 0.889 -> 0.855. */
#define SUB_EDGE 256
#define SUB_UNIT 1024
#define SUB_MAX 4
static int cand_minimizer_kept(int64_t Lq, int64_t pos, uint32_t hs, int K) {
    int64_t S = Lq / SUB_UNIT;
    if (S > SUB_MAX) S = SUB_MAX;
    if (S <= 1) return 1;
    if (pos < SUB_EDGE || pos + K > Lq - SUB_EDGE) return 1;
    return ((hs >> 1) % (uint32_t)S) == 0;
}

typedef struct { int32_t c, rel; int64_t d; int32_t qo; int64_t gpos; } hit_t;
static int cmp_hit(const void *a, const void *b) {
    const hit_t *x = (const hit_t *)a, *y = (const hit_t *)b;
    if (x->c != y->c) return x->c < y->c ? -1 : 1;
    if (x->rel != y->rel) return x->rel < y->rel ? -1 : 1;
    if (x->d != y->d) return x->d < y->d ? -1 : 1;
    return 0; /* order among equal (c, rel, d) does not influence the result */
}
typedef struct { int32_t c, contig, minus, anch; int64_t start1, end1, gstart; int32_t clip_l, clip_r; } copy_t;
static int cmp_copy(const void *a, const void *b) {
    const copy_t *x = (const copy_t *)a, *y = (const copy_t *)b;
    if (x->c != y->c) return x->c < y->c ? -1 : 1;
    int ax = x->anch > 4095 ? 4095 : x->anch, ay = y->anch > 4095 ? 4095 : y->anch;
    if (ax != ay) return ax > ay ? -1 : 1;
    if (x->gstart != y->gstart) return x->gstart < y->gstart ? -1 : 1;
    if (x->minus != y->minus) return x->minus < y->minus ? -1 : 1;
    return 0;
}
static int contig_of(const int64_t *coff, int nc, int64_t g) {
    int lo = 0, hi = nc;
    while (hi - lo > 1) { int mid = (lo + hi) / 2; if (coff[mid] <= g) lo = mid; else hi = mid; }
    return lo;
}

/*
 * genome: contigs concatenated (upper-case ASCII), contig_off[ncontig+1].  Candidates: cand + cand_off.
 * Output: CSR copy_first[ncand+1] into (contig, start1, end1 (1-based inclusive), minus, anchors).
 * Returns total copies or <0.
 */
/* ---- end extension ---------------------------------------------------------------------------------------------
 * Base-level extension of a chain from its outermost anchor to the end of the candidate, where minimap2 extends its chain
 * by dynamic programming and clips what does not align.  Unit-cost edit distance in a diagonal band of half-width EXT_B:
 * D[i][j] = cheapest alignment of the first i bases of the query segment (in walking order, away from the anchor) with the
 * first j genome bases in the same direction, |j - i| <= EXT_B (dir = +1: genome g0, g0 + 1, ...; dir = -1: g0 - 1, g0 - 2, ...);
 * cells that need genome bases outside [gmin, gmax) are unreachable; two bases match iff equal and one of ACGT.
 * C(i) = min_j D[i][j].  The extension is CUT where the score S(i) = i - EXT_PEN * C(i) is largest (a matched base +1, an
 * edit -(EXT_PEN - 1) or -EXT_PEN: with EXT_PEN = 3 the drift changes sign at 33 % divergence, as it does for minimap2's
 * map-ont scores 2 / -4 / -4-2k); ties: the largest i.  Returns i* = aligned query bases (0 .. n), *t_out = genome bases
 * used: the j with the smallest D[i*][j], ties |j - i*| smallest, then the smaller j. */
#define EXT_B 8
#define EXT_PEN 3
#define EXT_XDROP 40
#define EXT_MAXLEN 2048   /* an end of more than this many bases beyond the outermost anchor is not extended: the chain is dropped */
#define EXT_INF 1000000
static uint8_t comp_of(uint8_t c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }
/* S(i) = sa * i - sb * C(i), abandoned xdrop below the best: (1, 3, 40) for the copy finder; the tandem-repeat masker
 * (hite_oracle_trf.c) runs the same programme with match 2 / edit 5 as (2, 7, 30).  Only the diagonals dlo <= j - i <= dhi of
 * the band are used (the copy finder: all 17; the masker keeps the sequence from being aligned with ITSELF). */
int64_t orc_ext_align_scored(const uint8_t *qseg, int64_t n, int dir, const uint8_t *genome, int64_t g0, int64_t gmin, int64_t gmax,
                             int sa, int sb, int xdrop, int dlo, int dhi, int64_t *t_out, int64_t *score_out) {
    int prev[2 * EXT_B + 1], cur[2 * EXT_B + 1];
    /* cell (i, j) with j = i + b - EXT_B */
    for (int b = 0; b <= 2 * EXT_B; b++) {
        int64_t j = b - EXT_B;
        int ok = j >= 0 && j <= dhi && (dir > 0 ? g0 + j <= gmax : g0 - j >= gmin);
        prev[b] = ok ? (int)j : EXT_INF;
    }
    int64_t best_i = 0, best_t = 0, best_s = 0;     /* i = 0: C = 0 (j = 0), S = 0 */
    for (int64_t i = 1; i <= n; i++) {
        uint8_t qc = qseg[i - 1];
        int cmin = EXT_INF; int64_t tmin = i;
        for (int b = 0; b <= 2 * EXT_B; b++) {
            int64_t j = i + b - EXT_B;
            int v = EXT_INF;
            if (j >= 0 && b - EXT_B >= dlo && b - EXT_B <= dhi && (dir > 0 ? g0 + j <= gmax : g0 - j >= gmin)) {
                if (j >= 1) {
                    uint8_t gc = dir > 0 ? genome[g0 + j - 1] : genome[g0 - j];
                    int d = prev[b] + ((qc == gc && code_of(qc) >= 0) ? 0 : 1);
                    if (d < v) v = d;
                }
                if (b + 1 <= 2 * EXT_B && prev[b + 1] + 1 < v) v = prev[b + 1] + 1;     /* query base against a gap */
                if (b >= 1 && j >= 1 && cur[b - 1] + 1 < v) v = cur[b - 1] + 1;         /* genome base against a gap */
                if (v > EXT_INF) v = EXT_INF;
            }
            cur[b] = v;
            if (v < EXT_INF) {
                int64_t dv = j > i ? j - i : i - j, db = tmin > i ? tmin - i : i - tmin;
                if (v < cmin || (v == cmin && (dv < db || (dv == db && j < tmin)))) { cmin = v; tmin = j; }
            }
        }
        memcpy(prev, cur, sizeof prev);
        if (cmin >= EXT_INF) break;                 /* the band left the contig: nothing further is reachable */
        int64_t sc = (int64_t)sa * i - (int64_t)sb * cmin;
        if (sc >= best_s) { best_s = sc; best_i = i; best_t = tmin; }
        else if (sc < best_s - xdrop) break;
    }
    *t_out = best_t;
    if (score_out) *score_out = best_s;
    return best_i;
}
static int64_t ext_align(const uint8_t *qseg, int64_t n, int dir, const uint8_t *genome, int64_t g0, int64_t gmin, int64_t gmax, int64_t *t_out) {
    return orc_ext_align_scored(qseg, n, dir, genome, g0, gmin, gmax, 1, EXT_PEN, EXT_XDROP, -EXT_B, EXT_B, t_out, NULL);
}

/* ext_align for the tests (tests/test_host_compiled.py: the HIP device function compiled for the host against this one) */
int64_t orc_ext_align(const uint8_t *qseg, int64_t n, int dir, const uint8_t *genome, int64_t g0, int64_t gmin, int64_t gmax, int64_t *t_out) {
    return ext_align(qseg, n, dir, genome, g0, gmin, gmax, t_out);
}

/* interval mode of the records (hite_copy_config): 1 = aligned interval (Util.py:8026; the default since round 5), 0 = whole candidate */
static int g_aligned_interval = 1;
void orc_find_copies_config(int aligned_interval) { g_aligned_interval = aligned_interval ? 1 : 0; }

/* seconds the last orc_find_copies call spent building its index (bench.py separates residency set-up from the lookups) */
static double g_index_seconds = 0.0;
double orc_find_copies_index_seconds(void) { return g_index_seconds; }

/* candidate bases the two end extensions clipped (left / right in the orientation of the GENOME), per copy record of the last
 * orc_find_copies call, in output order: what hite_find_copies hands on beside the records as `clip` (see hite_gpu.h) */
static int32_t *g_clip = NULL;
static int64_t g_nclip = 0;
int64_t orc_find_copies_clips(int64_t cap, int32_t *clip_l, int32_t *clip_r) {
    for (int64_t t = 0; t < g_nclip && t < cap; t++) { clip_l[t] = g_clip[2 * t]; clip_r[t] = g_clip[2 * t + 1]; }
    return g_nclip;
}

/* one search of the candidates `sel[0..nsel)` (NULL: all) in the (W, K) minimizer index of the genome -> the chains kept, sorted by
 * (candidate, anchors descending, start, strand); *index_s += seconds spent building the index */
static copy_t *find_pass(const uint8_t *genome, const int64_t *contig_off, int ncontig, const uint8_t *cand, const int64_t *cand_off,
                         int ncand, const int32_t *sel, int nsel, int K, int W, int64_t *ncp_out, double *index_s) {
    /* index */
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int64_t M = 0;
    for (int c = 0; c < ncontig; c++) M += minimizers_kw(genome + contig_off[c], contig_off[c + 1] - contig_off[c], contig_off[c], NULL, K, W);
    mini_t *idx = (mini_t *)malloc(sizeof(mini_t) * (M + 1));
    int64_t k = 0;
    for (int c = 0; c < ncontig; c++) k += minimizers_kw(genome + contig_off[c], contig_off[c + 1] - contig_off[c], contig_off[c], idx + k, K, W);
    qsort(idx, M, sizeof(mini_t), cmp_mini);
    {
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        *index_s += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    }
    /* hits */
    int64_t hcap = 1 << 16, nh = 0;
    hit_t *hits = (hit_t *)malloc(sizeof(hit_t) * hcap);
    const int ntodo = sel ? nsel : ncand;
    for (int ci = 0; ci < ntodo; ci++) {
        const int c = sel ? sel[ci] : ci;
        const uint8_t *q = cand + cand_off[c];
        int64_t Lq = cand_off[c + 1] - cand_off[c];
        int64_t nm = minimizers_kw(q, Lq, 0, NULL, K, W);
        if (nm <= 0) continue;
        mini_t *qm = (mini_t *)malloc(sizeof(mini_t) * nm);
        minimizers_kw(q, Lq, 0, qm, K, W);
        for (int64_t t = 0; t < nm; t++) {
            uint32_t h31 = qm[t].hs >> 1;
            if (!cand_minimizer_kept(Lq, qm[t].pos, qm[t].hs, K)) continue;
            /* lower bound of hs >= h31 << 1 */
            int64_t lo = 0, hi = M;
            while (lo < hi) { int64_t mid = (lo + hi) / 2; if ((idx[mid].hs >> 1) < h31) lo = mid + 1; else hi = mid; }
            int64_t e = lo;
            while (e < M && (idx[e].hs >> 1) == h31) e++;
            if (e - lo > MAXOCC) continue;
            for (int64_t i = lo; i < e; i++) {
                if (nh == hcap) { hcap *= 2; hits = (hit_t *)realloc(hits, sizeof(hit_t) * hcap); }
                int rel = (int)((qm[t].hs ^ idx[i].hs) & 1u);
                int64_t qo = rel ? (Lq - qm[t].pos - K) : qm[t].pos;
                hits[nh].c = c; hits[nh].rel = rel; hits[nh].qo = (int32_t)qo; hits[nh].gpos = idx[i].pos;
                hits[nh].d = idx[i].pos - qo;
                nh++;
            }
        }
        free(qm);
    }
    qsort(hits, nh, sizeof(hit_t), cmp_hit);
    /* clusters -> copies */
    int64_t ccap = 1 << 12, ncp = 0;
    copy_t *cps = (copy_t *)malloc(sizeof(copy_t) * ccap);
    int64_t i = 0;
    while (i < nh) {
        int64_t j = i + 1;
        int ctg = contig_of(contig_off, ncontig, hits[i].gpos);
        while (j < nh && hits[j].c == hits[i].c && hits[j].rel == hits[i].rel &&
               contig_of(contig_off, ncontig, hits[j].gpos) == ctg && hits[j].d - hits[j - 1].d <= TD) j++;
        int64_t Lq = cand_off[hits[i].c + 1] - cand_off[hits[i].c];
        int64_t qlo = hits[i].qo, glo = hits[i].gpos, qhi = hits[i].qo, ghi = hits[i].gpos;
        for (int64_t t = i; t < j; t++) {
            if (hits[t].qo < qlo || (hits[t].qo == qlo && hits[t].gpos < glo)) { qlo = hits[t].qo; glo = hits[t].gpos; }
            if (hits[t].qo > qhi || (hits[t].qo == qhi && hits[t].gpos > ghi)) { qhi = hits[t].qo; ghi = hits[t].gpos; }
        }
        int64_t na = j - i;
        /* acceptance.  The reference keeps a minimap2 alignment when query_alignment_length / len(query) >= 0.95 and
         * query_alignment_length / (M + D) >= 0.95 (get_copies_minimap2, Util.py:8008-8022).  Here a chain of >= 3 anchors whose
         * query span is >= 95 % of its genome span is extended base by base from its outermost anchors to both ends of the
         * candidate (ext_align); what the extension cuts off is clipped, as minimap2 soft-clips it.  aligned = Lq - clipped;
         * the copy is the genome interval of the aligned part, kept when aligned >= 95 % of Lq and aligned >= 95 % of that interval. */
        if (na >= MINANCH && (qhi + K - qlo) * 100 >= 95 * (ghi + K - glo) && qlo <= EXT_MAXLEN && Lq - (qhi + K) <= EXT_MAXLEN) {
            int64_t cb = contig_off[ctg], ce = contig_off[ctg + 1];
            const uint8_t *q = cand + cand_off[hits[i].c];
            int rel = hits[i].rel;
            int64_t nl = qlo, nr = Lq - (qhi + K);
            uint8_t *seg = (uint8_t *)malloc((size_t)(nl > nr ? nl : nr) + 1);
            /* query in the orientation of the genome: rel = 1 reads the reverse complement of the candidate */
#define QAT(x) (rel ? comp_of(q[Lq - 1 - (x)]) : q[(x)])
            for (int64_t x = 0; x < nl; x++) seg[x] = QAT(qlo - 1 - x);
            int64_t tl = 0, tr = 0;
            int64_t il = ext_align(seg, nl, -1, genome, glo, cb, ce, &tl);
            for (int64_t x = 0; x < nr; x++) seg[x] = QAT(qhi + K + x);
            int64_t ir = ext_align(seg, nr, +1, genome, ghi + K, cb, ce, &tr);
            free(seg);
            int64_t clip_l = nl - il, clip_r = nr - ir;
            int64_t aligned = Lq - clip_l - clip_r;
            int64_t a0 = glo - tl, a1 = ghi + K + tr;                 /* genome interval of the aligned part */
            if (a1 > a0 && aligned * 100 >= 95 * Lq && aligned * 100 >= 95 * (a1 - a0)) {
                /* the interval handed on covers the whole candidate: the clipped ends (<= 5 % of it) lie on the diagonal of the last
                 * aligned base, clamped to the contig */
                int64_t s0 = g_aligned_interval ? a0 : a0 - clip_l, e0 = g_aligned_interval ? a1 : a1 + clip_r;
                if (s0 < cb) s0 = cb;
                if (e0 > ce) e0 = ce;
                if (ncp == ccap) { ccap *= 2; cps = (copy_t *)realloc(cps, sizeof(copy_t) * ccap); }
                cps[ncp].c = hits[i].c; cps[ncp].contig = ctg; cps[ncp].minus = hits[i].rel; cps[ncp].anch = (int32_t)na;
                cps[ncp].start1 = s0 - cb + 1; cps[ncp].end1 = e0 - cb; cps[ncp].gstart = s0;
                /* aligned interval: the clipped candidate bases (in the orientation of the genome) travel beside the record; whole-candidate
                 * interval: they are inside it */
                cps[ncp].clip_l = g_aligned_interval ? (int32_t)(clip_l > 0xffff ? 0xffff : clip_l) : 0;
                cps[ncp].clip_r = g_aligned_interval ? (int32_t)(clip_r > 0xffff ? 0xffff : clip_r) : 0;
                ncp++;
            }
        }
        i = j;
    }
    qsort(cps, ncp, sizeof(copy_t), cmp_copy);
    free(idx); free(hits);
    *ncp_out = ncp;
    return cps;
}

/* The far pass (where minimap2 keeps secondary chains of any divergence its -N 300 -p 0.2 admits, Util.py:7952-7961): a (10, 15)
 * minimizer survives a pair divergence d with (1 - d)^15 -- 2 % at d = 0.225 --, so copies 20-30 % from the candidate are found at
 * 0.66 / 0.37 (tools/copy_recall_by_divergence.py).  Candidates that come out of the first search with FEWER THAN far_min copies are
 * searched a second time in a (FAR_W, FAR_K) = (8, 13) index of the same genome, same rules; the second table REPLACES the
 * candidate's first when it holds more copies (else the first stands).  far_min = 0: no far pass. */
#define FAR_K 13
#define FAR_W 8
static int g_far_min = 0;
/* Clip words for a copy record that carries none: the reference's own tuples (chr, reference_start + 1, reference_end, length,
 * strand) of get_copies_minimap2, /root/reference/module/Util.py:8022-8030 -- minimap2's soft clips are dropped there --, consumed at
 * Util.py:8095-8115.  Twin of clip_probe_kernel (hite_amd/csrc/hite_pipeline.hip); THIS BUILD'S definition, not minimap2's.
 * q: the candidate, y: the record's interval read on the candidate's strand.  The first K = 21 bases of y are laid on q at every
 * offset d = 0 .. min(|q| - K - shift, |q| / 20 + 32, 65535); a base matches when both are the same of ACGT (either case in q); the
 * offset with the fewest mismatches wins, the smallest on ties; it is the left clip when it has <= 5 mismatches; otherwise the same
 * with the NEXT K bases (shift = K) against q[d + K ..].  The right clip: the same from the other end (the last K bases of y
 * against q[|q| - d - K - shift ..]).  An end that finds no offset takes what the other end's clip leaves of |q| - |y|, clamped to
 * 0 .. the largest offset tried; 0 when neither end finds one.  returns left | right << 16 in the CANDIDATE's orientation. */
#define ORC_CLIP_K 21
#define ORC_CLIP_MM 5
static int probe_eq(uint8_t a, uint8_t b) {
    a &= 0xdf; b &= 0xdf;
    return a == b && (a == 'A' || a == 'C' || a == 'G' || a == 'T');
}
uint32_t orc_clip_probe(const uint8_t *q, int64_t Lq, const uint8_t *y, int64_t Ly) {
    int clip[2] = {-1, -1};
    if (Ly < ORC_CLIP_K || Lq < ORC_CLIP_K) return 0;
    int64_t dmax = Lq / 20 + 32;
    if (dmax > Lq - ORC_CLIP_K) dmax = Lq - ORC_CLIP_K;
    if (dmax > 0xffff) dmax = 0xffff;
    for (int right = 0; right < 2; right++) {
        for (int att = 0; att < 2; att++) {
            const int64_t sh = (int64_t)att * ORC_CLIP_K;
            if (Ly < sh + ORC_CLIP_K) break;
            int64_t dm = dmax;
            if (dm > Lq - ORC_CLIP_K - sh) dm = Lq - ORC_CLIP_K - sh;
            if (dm < 0) break;
            int best = ORC_CLIP_K + 1, arg = 0;
            for (int64_t d = 0; d <= dm; d++) {
                int mm = 0;
                for (int i = 0; i < ORC_CLIP_K; i++) {
                    const int64_t yp = right ? Ly - sh - ORC_CLIP_K + i : sh + i;
                    const int64_t qp = right ? Lq - d - sh - ORC_CLIP_K + i : d + sh + i;
                    mm += !probe_eq(q[qp], y[yp]);
                }
                if (mm < best) { best = mm; arg = (int)d; }
            }
            if (best <= ORC_CLIP_MM) { clip[right] = arg; break; }
        }
    }
    /* an end without an offset takes what the other end leaves of the length difference (no net insertion / deletion assumed) */
    if (clip[0] < 0 && clip[1] < 0) return 0;
    for (int e = 0; e < 2; e++)
        if (clip[e] < 0) {
            const int64_t v = (Lq - Ly) - clip[1 - e];
            clip[e] = (int)(v < 0 ? 0 : (v > dmax ? dmax : v));
        }
    return (uint32_t)clip[0] | ((uint32_t)clip[1] << 16);
}

void orc_find_copies_far(int min_copies) { g_far_min = min_copies > 0 ? min_copies : 0; }

int64_t orc_find_copies(const uint8_t *genome, const int64_t *contig_off, int ncontig, const uint8_t *cand,
                        const int64_t *cand_off, int ncand, int64_t cap, int32_t *copy_first, int32_t *contig,
                        int64_t *start1, int64_t *end1, uint8_t *minus, int32_t *anchors) {
    if (ncontig <= 0 || ncand < 0) return ORC_EINVAL;
    g_index_seconds = 0.0;
    int64_t ncp = 0, ncp2 = 0;
    copy_t *cps = find_pass(genome, contig_off, ncontig, cand, cand_off, ncand, NULL, 0, CK, CW, &ncp, &g_index_seconds);
    copy_t *cps2 = NULL;
    int64_t *first2 = NULL;      /* rows of candidate c in the far table: first2[c] .. first2[c + 1] */
    uint8_t *use2 = NULL;        /* the far table replaces the first for candidate c */
    if (g_far_min > 0 && ncand > 0) {
        int32_t *cnt = (int32_t *)calloc((size_t)ncand, sizeof(int32_t)), *sel = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncand);
        for (int64_t t = 0; t < ncp; t++) cnt[cps[t].c]++;
        int nsel = 0;
        for (int c = 0; c < ncand; c++) if (cnt[c] < g_far_min) sel[nsel++] = c;
        if (nsel > 0) {
            cps2 = find_pass(genome, contig_off, ncontig, cand, cand_off, ncand, sel, nsel, FAR_K, FAR_W, &ncp2, &g_index_seconds);
            first2 = (int64_t *)calloc((size_t)ncand + 1, sizeof(int64_t));
            int32_t *cnt2 = (int32_t *)calloc((size_t)ncand, sizeof(int32_t));
            for (int64_t t = 0; t < ncp2; t++) cnt2[cps2[t].c]++;
            int64_t acc = 0;
            for (int c = 0; c < ncand; c++) { first2[c] = acc; acc += cnt2[c]; }
            first2[ncand] = acc;
            use2 = (uint8_t *)calloc((size_t)ncand, 1);
            for (int c = 0; c < ncand; c++) use2[c] = cnt2[c] > cnt[c];
            free(cnt2);
        }
        free(cnt); free(sel);
    }
    int64_t nout = 0;
    copy_first[0] = 0;
    int64_t p = 0;
    free(g_clip);
    g_clip = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(cap > 0 ? cap : 1));
    g_nclip = 0;
    for (int c = 0; c < ncand; c++) {
        int kept = 0;
        const int far = use2 && use2[c];
        if (far) {
            for (int64_t t = first2[c]; t < first2[c + 1] && kept < MAXCOPY; t++, kept++) {
                if (nout >= cap) { free(cps); free(cps2); free(first2); free(use2); return ORC_ECAP; }
                contig[nout] = cps2[t].contig; start1[nout] = cps2[t].start1; end1[nout] = cps2[t].end1;
                minus[nout] = (uint8_t)cps2[t].minus; anchors[nout] = cps2[t].anch;
                g_clip[2 * nout] = cps2[t].clip_l; g_clip[2 * nout + 1] = cps2[t].clip_r;
                nout++;
            }
        }
        while (p < ncp && cps[p].c == c) {
            if (!far && kept < MAXCOPY) {
                if (nout >= cap) { free(cps); free(cps2); free(first2); free(use2); return ORC_ECAP; }
                contig[nout] = cps[p].contig; start1[nout] = cps[p].start1; end1[nout] = cps[p].end1;
                minus[nout] = (uint8_t)cps[p].minus; anchors[nout] = cps[p].anch;
                g_clip[2 * nout] = cps[p].clip_l; g_clip[2 * nout + 1] = cps[p].clip_r;
                nout++; kept++;
            }
            p++;
        }
        copy_first[c + 1] = (int32_t)nout;
    }
    free(cps); free(cps2); free(first2); free(use2);
    g_nclip = nout;
    return nout;
}

/* ======================================================================================================
 * All-vs-all seeding: CPU twin of hite_seed_allvsall (hite_amd/csrc/hite_copies.hip), the build's OWN stage
 * where the reference runs `blastn -evalue 1e-20 -outfmt 6` of every 1 Mbp segment file against every file
 * (process_blast_alignments / sequence2sequenceBlastn, /root/reference/module/Util.py:4724-4780, 4068-4091).
 * rmblast 2.14.0 is third-party and absent: PARITY UNPINNED at that boundary; pinned is HIP == this twin.
 *
 * Definition (shared with the HIP kernels; index and minimizers as above)
 *   Every genome minimizer, taken in position order, is a query seed; its partners are the index entries
 *   with the same hs >> 1 at another position, in index order (skipped when the run holds more than
 *   SEED_MAXOCC = 1000 entries).  A pair is an anchor (pi, pj, rel = strand_i ^ strand_j) on the diagonal
 *   d = pj - pi + G (rel 0) or pi + pj (rel 1).  Anchors are ordered by (rel, d >> 6), ties in generation
 *   order (a stable sort), and cut into clusters where rel, d >> 6, the contig of pi or of pj changes or pi
 *   advances by more than SEED_GAP = 300.  A cluster with >= 3 anchors and a query span
 *   (last pi + K - first pi) >= 60 is an HSP: query [first pi, last pi + K), subject
 *   [min(pj_first, pj_last), max(pj_first, pj_last) + K).  HSPs are cut at the borders of the seg_len
 *   segments ('chr$offset' naming of split_genome_chunks.py:41-52) of the query, then of the subject, the
 *   other side following linearly (reversed for rel 1) and clamped, pieces shorter than 20 bases on either side are
 *   dropped; records = (qseg, sseg, qs, qe, ss, se),
 *   1-based inclusive inside the segment, ss > se for rel 1; output order = (qseg, sseg), ties in cluster
 *   order (stable).
 * ====================================================================================================== */
#define SEED_MAXOCC 1000
#define SEED_GAP 300
#define SEED_MINANCH 3
#define SEED_MINSPAN 60
#define SEED_MINPIECE 20

typedef struct { uint64_t key; uint32_t pi; int64_t ord; } anchor_t;
static int cmp_anchor(const void *a, const void *b) {
    const anchor_t *x = (const anchor_t *)a, *y = (const anchor_t *)b;
    uint64_t kx = x->key >> 6, ky = y->key >> 6;
    if (kx != ky) return kx < ky ? -1 : 1;
    if (x->ord != y->ord) return x->ord < y->ord ? -1 : 1;
    return 0;
}
static int cmp_mini_pos(const void *a, const void *b) {
    const mini_t *x = (const mini_t *)a, *y = (const mini_t *)b;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return 0;
}
typedef struct { int32_t qseg, sseg; int64_t qs, qe, ss, se; int64_t ord; } hsp_t;
static int cmp_hsp(const void *a, const void *b) {
    const hsp_t *x = (const hsp_t *)a, *y = (const hsp_t *)b;
    if (x->qseg != y->qseg) return x->qseg < y->qseg ? -1 : 1;
    if (x->sseg != y->sseg) return x->sseg < y->sseg ? -1 : 1;
    if (x->ord != y->ord) return x->ord < y->ord ? -1 : 1;
    return 0;
}

/* segment table of a genome: per contig ceil(len / seg_len) segments; returns nseg (seg_chrom / seg_off may be NULL) */
int orc_seed_segments(const int64_t *contig_off, int ncontig, int64_t seg_len, int32_t *seg_chrom, int64_t *seg_off, int cap) {
    int n = 0;
    for (int c = 0; c < ncontig; c++) {
        int64_t L = contig_off[c + 1] - contig_off[c];
        for (int64_t o = 0; o < L || (L == 0 && o == 0); o += seg_len) {
            if (seg_chrom && n < cap) { seg_chrom[n] = c; seg_off[n] = o; }
            n++;
            if (L == 0) break;
        }
    }
    return n;
}

/* Sharding over ranks (mirror of hite_seed_shard, hite_amd/csrc/hite_copies.hip): a rank keeps the anchors whose
 * lin = strand * 2 G + diagonal lies in its range; the edges sit at equal steps of the triangular distribution of the
 * diagonals of two uniform positions and are multiples of 64, so that no cluster straddles two ranks: the HSPs of the ranks,
 * concatenated in rank order, are those of one unsharded run in its order. */
static int g_seed_rank = 0, g_seed_world = 0;
void orc_seed_shard(int rank, int world) { g_seed_rank = world > 1 ? rank : 0; g_seed_world = world > 1 ? world : 0; }
static uint64_t seed_shard_edge(int64_t G, int k, int world) {
    if (k <= 0) return 0;
    if (k >= world) return 4ull * (uint64_t)G + 64ull;
    const double mass = 2.0 * (double)k / (double)world;
    const int strand = mass >= 1.0 ? 1 : 0;
    const double t = mass - (double)strand;
    const double d = t <= 0.5 ? (double)G * sqrt(2.0 * t) : 2.0 * (double)G - (double)G * sqrt(2.0 * (1.0 - t));
    uint64_t e = (uint64_t)(d < 0 ? 0 : d);
    e &= ~63ull;
    return (strand ? 2ull * (uint64_t)G : 0ull) + e;
}

int64_t orc_seed_allvsall(const uint8_t *genome, const int64_t *contig_off, int ncontig, int64_t seg_len, int64_t cap,
                          int32_t *qseg, int32_t *sseg, int64_t *qs, int64_t *qe, int64_t *ss, int64_t *se) {
    if (ncontig <= 0 || seg_len <= 0) return ORC_EINVAL;
    const int64_t G = contig_off[ncontig];
    uint64_t sh_lo = 0, sh_hi = 0;
    const int sharded = g_seed_world > 1;      /* (an empty share -- both edges on the same multiple of 64 -- owns nothing) */
    if (sharded) {
        sh_lo = seed_shard_edge(G, g_seed_rank, g_seed_world);
        sh_hi = seed_shard_edge(G, g_seed_rank + 1, g_seed_world);
    }
    int64_t M = 0;
    for (int c = 0; c < ncontig; c++) M += minimizers(genome + contig_off[c], contig_off[c + 1] - contig_off[c], contig_off[c], NULL);
    mini_t *idx = (mini_t *)malloc(sizeof(mini_t) * (M + 1)), *byp = (mini_t *)malloc(sizeof(mini_t) * (M + 1));
    int64_t k = 0;
    for (int c = 0; c < ncontig; c++) k += minimizers(genome + contig_off[c], contig_off[c + 1] - contig_off[c], contig_off[c], idx + k);
    memcpy(byp, idx, sizeof(mini_t) * M);
    qsort(idx, M, sizeof(mini_t), cmp_mini);
    qsort(byp, M, sizeof(mini_t), cmp_mini_pos);
    int *seg_base = (int *)malloc(sizeof(int) * (ncontig + 1));
    seg_base[0] = 0;
    for (int c = 0; c < ncontig; c++) {
        int64_t L = contig_off[c + 1] - contig_off[c];
        seg_base[c + 1] = seg_base[c] + (int)(L > 0 ? (L + seg_len - 1) / seg_len : 1);
    }
    /* anchors */
    int64_t acap = 1 << 16, na = 0;
    anchor_t *an = (anchor_t *)malloc(sizeof(anchor_t) * acap);
    for (int64_t t = 0; t < M; t++) {
        uint32_t h31 = byp[t].hs >> 1;
        int64_t lo = 0, hi = M;
        while (lo < hi) { int64_t mid = (lo + hi) / 2; if ((idx[mid].hs >> 1) < h31) lo = mid + 1; else hi = mid; }
        int64_t e = lo;
        while (e < M && (idx[e].hs >> 1) == h31) e++;
        if (e - lo > SEED_MAXOCC) continue;
        for (int64_t i = lo; i < e; i++) {
            if (idx[i].pos == byp[t].pos) continue;
            if (na == acap) { acap *= 2; an = (anchor_t *)realloc(an, sizeof(anchor_t) * acap); }
            uint64_t rel = (byp[t].hs ^ idx[i].hs) & 1u;
            uint64_t d = rel ? (uint64_t)(byp[t].pos + idx[i].pos) : (uint64_t)(idx[i].pos - byp[t].pos + G);
            if (sharded) { const uint64_t lin = (rel ? 2ull * (uint64_t)G : 0ull) + d; if (lin < sh_lo || lin >= sh_hi) continue; }
            an[na].key = (rel << 34) | d; an[na].pi = (uint32_t)byp[t].pos; an[na].ord = na;
            na++;
        }
    }
    qsort(an, na, sizeof(anchor_t), cmp_anchor);
    /* clusters -> HSP pieces */
    int64_t hcap = 1 << 12, nh = 0;
    hsp_t *hs = (hsp_t *)malloc(sizeof(hsp_t) * hcap);
    int64_t i = 0;
    while (i < na) {
#define AN_PJ(a) ((int64_t)(((a).key >> 34) ? (int64_t)((a).key & 0x3ffffffffull) - (int64_t)(a).pi : (int64_t)((a).key & 0x3ffffffffull) - G + (int64_t)(a).pi))
        int64_t j = i + 1;
        int cq = contig_of(contig_off, ncontig, an[i].pi), cs = contig_of(contig_off, ncontig, AN_PJ(an[i]));
        while (j < na && (an[j].key >> 6) == (an[i].key >> 6) && (int64_t)an[j].pi - (int64_t)an[j - 1].pi <= SEED_GAP &&
               contig_of(contig_off, ncontig, an[j].pi) == cq && contig_of(contig_off, ncontig, AN_PJ(an[j])) == cs) j++;
        int64_t cnt = j - i;
        int64_t q0 = an[i].pi, q1 = (int64_t)an[j - 1].pi + CK;
        if (cnt >= SEED_MINANCH && q1 - q0 >= SEED_MINSPAN) {
            int rel = (int)(an[i].key >> 34);
            int64_t pf = AN_PJ(an[i]), pl = AN_PJ(an[j - 1]);
            int64_t s0 = pf < pl ? pf : pl, s1 = (pf < pl ? pl : pf) + CK;
            int64_t qb = contig_off[cq], sb = contig_off[cs];
            /* query pieces */
            for (int64_t a = q0; a < q1;) {
                int64_t qsegi = (a - qb) / seg_len;
                int64_t b = qb + (qsegi + 1) * seg_len;
                if (b > q1) b = q1;
                /* subject range of [a, b) */
                int64_t u0, u1;
                if (!rel) { u0 = s0 + (a - q0); u1 = s0 + (b - q0); } else { u0 = s1 - (b - q0); u1 = s1 - (a - q0); }
                if (u0 < s0) u0 = s0;
                if (u1 > s1) u1 = s1;
                for (int64_t x = u0; x < u1;) {
                    int64_t ssegi = (x - sb) / seg_len;
                    int64_t y = sb + (ssegi + 1) * seg_len;
                    if (y > u1) y = u1;
                    /* query sub-range of subject piece [x, y) */
                    int64_t a2, b2;
                    if (!rel) { a2 = a + (x - u0); b2 = a + (y - u0); } else { a2 = a + (u1 - y); b2 = a + (u1 - x); }
                    if (a2 < a) a2 = a;
                    if (b2 > b) b2 = b;
                    if (b2 - a2 >= SEED_MINPIECE && y - x >= SEED_MINPIECE) {
                        if (nh == hcap) { hcap *= 2; hs = (hsp_t *)realloc(hs, sizeof(hsp_t) * hcap); }
                        hsp_t *o = &hs[nh];
                        o->qseg = seg_base[cq] + (int32_t)qsegi; o->sseg = seg_base[cs] + (int32_t)ssegi;
                        int64_t qo = qb + qsegi * seg_len, so = sb + ssegi * seg_len;
                        o->qs = a2 - qo + 1; o->qe = b2 - qo;
                        if (!rel) { o->ss = x - so + 1; o->se = y - so; } else { o->ss = y - so; o->se = x - so + 1; }
                        o->ord = nh;
                        nh++;
                    }
                    x = y;
                }
                a = b;
            }
        }
        i = j;
    }
    qsort(hs, nh, sizeof(hsp_t), cmp_hsp);
    int64_t rc = nh;
    if (nh > cap) rc = ORC_ECAP;
    else for (int64_t t = 0; t < nh; t++) { qseg[t] = hs[t].qseg; sseg[t] = hs[t].sseg; qs[t] = hs[t].qs; qe[t] = hs[t].qe; ss[t] = hs[t].ss; se[t] = hs[t].se; }
    free(idx); free(byp); free(seg_base); free(an); free(hs);
    return rc;
}
