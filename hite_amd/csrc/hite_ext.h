// hite_ext.h -- banded end extension shared by the copy finder (hite_copies.hip: a chain is extended base by base to the ends
// of the candidate) and the tandem-repeat masker (hite_trmask.hip: a stretch is aligned with itself one period further on).
// Definition: orc_ext_align_scored in oracle/hite_oracle_copies.c.
#pragma once
#include "hite_common.h"

// >>> ext_align_dev (tests/test_host_compiled.py compiles this block for the host and compares it with the twin)
#define EXT_B 8
#define EXT_W (2 * EXT_B + 1)
#define EXT_INF (1 << 20)
// what a user of the extension fixes at compile time: score S(i) = SA i - SB cost, abandoned XDROP below the best; where the
// query bases come from (ASCII bytes of a candidate / the packed genome itself); whether only some diagonals are allowed
struct ExtCopyMode { static constexpr int SA = 1, SB = 3, XDROP = 40; static constexpr bool PACKEDQ = false, DIAGLIM = false; };
struct ExtTandemMode { static constexpr int SA = 2, SB = 7, XDROP = 30; static constexpr bool PACKEDQ = true, DIAGLIM = true; };   // match 2, edit 5: oracle/hite_oracle_trf.c, "calibration"

__device__ __forceinline__ unsigned ext_cand_code(unsigned ch, bool comp) {
    const unsigned c = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
    return (comp && c < 4u) ? 3u - c : c;
}
// State machine of ext_align: query bases q[p0], q[p0 + step], ... (n of them; complemented when comp) -- or, PACKEDQ, genome
// bases p0, p0 + step, ... -- against genome bases g0, g0 + 1, ... (dir = +1) or g0 - 1, g0 - 2, ... (dir = -1), at most jmax of
// them.  The 17 band cells live in registers (the loops over the band are unrolled), the genome bases under the band as
// three 17-bit planes that shift by one cell per column: a column costs one new genome base, one query base and ~8 integer
// operations per cell.  Both sequences are walked one base per column, so the words they come from are kept in registers -- 16
// genome bases, 32 mask bits, 4 query bytes per load, each fetched a few words ahead of its use (a load per base and column
// made the kernel a gather benchmark; one word ahead left the last long extensions of a batch -- a lane alone in its
// wavefront -- waiting for HBM every fourth column).
template <class M>
struct ExtStateT {
    int D[EXT_W];
    uint32_t W0, W1, WN;
    int i, n, best_i, best_t, best_s;
    unsigned gnext_, qnext_;                // codes of the next column
    uint32_t gw, gwn, gwn2, mw, mwn, qw, qwn, qwn2, qwn3;   // current word and the ones fetched ahead: genome bases (2 ahead = 32 columns),
                                                            // mask bits (1 ahead = 32 columns), query bytes (3 ahead = 12 columns)
    int64_t gwi, mwi, qwi;                   // their word indices
    const uint32_t *q4;                      // the candidate bytes as aligned words (base address rounded down)
    int64_t p0, g0, jmax;
    int step, dir;
    int blo, bhi;                            // DIAGLIM: band cells in use (b = j - i + EXT_B)
    bool comp;
};
// code of genome base g (0..3, 4 = not A/C/G/T)
__device__ __forceinline__ unsigned ext_genome_code(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask, int64_t g) {
    const unsigned c = (bases[g >> 4] >> (2 * (int)(g & 15))) & 3u;
    return ((nmask[g >> 5] >> (int)(g & 31)) & 1u) ? 4u : c;
}
// the same through the word cache; g moves by one base per call in direction E.dir
template <class M>
__device__ __forceinline__ unsigned ext_genome_next(ExtStateT<M> &E, const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask, int64_t g) {
    const int64_t wi = g >> 4, mi = g >> 5;
    if (wi != E.gwi) { E.gw = E.gwn; E.gwn = E.gwn2; E.gwi = wi; const int64_t nx = wi + 2 * E.dir; E.gwn2 = bases[nx > 0 ? nx : 0]; }
    if (mi != E.mwi) { E.mw = E.mwn; E.mwi = mi; const int64_t nx = mi + E.dir; E.mwn = nmask[nx > 0 ? nx : 0]; }
    const unsigned c = (E.gw >> (2 * (int)(g & 15))) & 3u;
    return ((E.mw >> (int)(g & 31)) & 1u) ? 4u : c;
}
// code of the query base at byte address a (relative to q4) / PACKEDQ: at genome position a; a moves by E.step per call
template <class M>
__device__ __forceinline__ unsigned ext_query_next(ExtStateT<M> &E, const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask, int64_t a) {
    if (M::PACKEDQ) return a >= 0 ? ext_genome_code(bases, nmask, a) : 4u;
    const int64_t wi = a >> 2;
    if (wi != E.qwi) { E.qw = E.qwn; E.qwn = E.qwn2; E.qwn2 = E.qwn3; E.qwi = wi; const int64_t nx = wi + 3 * E.step; E.qwn3 = E.q4[nx > 0 ? nx : 0]; }
    return ext_cand_code((E.qw >> (8 * (int)(a & 3))) & 0xffu, E.comp);
}
// dlo .. dhi: the diagonals j - i in use (DIAGLIM only; else all of the band)
template <class M>
__device__ __forceinline__ void ext_init(ExtStateT<M> &E, const uint8_t *__restrict__ q, int64_t p0, int step, bool comp, int n,
                                         const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask, int64_t g0, int dir, int64_t jmax,
                                         int dlo = -EXT_B, int dhi = EXT_B) {
    if (M::PACKEDQ) { E.q4 = nullptr; E.p0 = p0; }
    else {
        const uintptr_t qa = (uintptr_t)q;
        E.q4 = (const uint32_t *)(qa & ~(uintptr_t)3);
        E.p0 = p0 + (int64_t)(qa & 3);           // byte offset of the first query base from q4
    }
    E.step = step; E.comp = comp; E.n = n; E.g0 = g0; E.dir = dir; E.jmax = jmax;
    E.blo = dlo + EXT_B; E.bhi = dhi + EXT_B;
#pragma unroll
    for (int b = 0; b < EXT_W; b++) { const int j = b - EXT_B; E.D[b] = (j >= 0 && j <= jmax && (!M::DIAGLIM || j <= dhi)) ? j : EXT_INF; }
    E.best_i = 0; E.best_t = 0; E.best_s = 0; E.i = 1;
    E.W0 = 0u; E.W1 = 0u; E.WN = (1u << EXT_W) - 1u;
    E.gnext_ = 4u; E.qnext_ = 4u;
    if (n < 1) return;
    // word caches: the first genome base read is number 1 (g0 or g0 - 1), the first query byte p0
    {
        const int64_t g = dir > 0 ? g0 : g0 - 1;
        const int64_t gs = g > 0 ? g : 0;
        E.gwi = gs >> 4; E.gw = bases[E.gwi];
        { const int64_t n1 = E.gwi + dir, n2 = E.gwi + 2 * dir; E.gwn = bases[n1 > 0 ? n1 : 0]; E.gwn2 = bases[n2 > 0 ? n2 : 0]; }
        E.mwi = gs >> 5; E.mw = nmask[E.mwi]; { const int64_t nx = E.mwi + dir; E.mwn = nmask[nx > 0 ? nx : 0]; }
        if (!M::PACKEDQ) {
            E.qwi = E.p0 >> 2; E.qw = E.q4[E.qwi];
            const int64_t n1 = E.qwi + step, n2 = E.qwi + 2 * step, n3 = E.qwi + 3 * step;
            E.qwn = E.q4[n1 > 0 ? n1 : 0]; E.qwn2 = E.q4[n2 > 0 ? n2 : 0]; E.qwn3 = E.q4[n3 > 0 ? n3 : 0];
        }
    }
    // planes of "column 0": bit b = genome base number j = b - EXT_B (1-based in walking order); bit set in WN = never matches
#pragma unroll
    for (int j = 1; j <= EXT_B; j++) {
        if (j <= jmax) {
            const unsigned cd = ext_genome_next(E, bases, nmask, dir > 0 ? g0 + j - 1 : g0 - j);
            E.W0 |= (cd & 1u) << (j + EXT_B); E.W1 |= ((cd >> 1) & 1u) << (j + EXT_B); E.WN &= ~((~(cd >> 2) & 1u) << (j + EXT_B));
        }
    }
    if (1 + EXT_B <= jmax) E.gnext_ = ext_genome_next(E, bases, nmask, dir > 0 ? g0 + EXT_B : g0 - 1 - EXT_B);
    E.qnext_ = ext_query_next(E, bases, nmask, E.p0);
}
// one column (E.i <= E.n on entry); returns true when the extension is finished (result in best_i / best_t / best_s)
template <class M>
__device__ __forceinline__ bool ext_step(ExtStateT<M> &E, const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask) {
    const int i = E.i;
    const unsigned gc = E.gnext_, qc = E.qnext_;
    if (i < E.n) {    // the bases of column i + 1
        const int64_t j = (int64_t)i + 1 + EXT_B;
        E.gnext_ = j <= E.jmax ? ext_genome_next(E, bases, nmask, E.dir > 0 ? E.g0 + j - 1 : E.g0 - j) : 4u;
        E.qnext_ = ext_query_next(E, bases, nmask, E.p0 + (int64_t)E.step * i);
    }
    E.W0 = (E.W0 >> 1) | ((gc & 1u) << (EXT_W - 1)); E.W1 = (E.W1 >> 1) | (((gc >> 1) & 1u) << (EXT_W - 1)); E.WN = (E.WN >> 1) | ((gc >> 2) << (EXT_W - 1));
    const uint32_t eq = qc < 4u ? (~(E.W0 ^ (0u - (qc & 1u))) & ~(E.W1 ^ (0u - ((qc >> 1) & 1u))) & ~E.WN) : 0u;
    // cells with j < 0 need no guard: they start at EXT_INF and every move into them comes from such a cell.  Cells with
    // j > jmax (beyond the contig) are forced to EXT_INF -- only the rare extension that can reach the contig end pays for it
    const int64_t hi64 = E.jmax - i + EXT_B;                    // cells with j <= jmax
    int left = EXT_INF, kmin = 0x7fffffff;
    if (!M::DIAGLIM && hi64 >= EXT_W - 1) {
#pragma unroll
        for (int b = 0; b < EXT_W; b++) {
            // min(diag, up, left + 1) = 1 + min(D[b] - match, D[b + 1], left)
            const int dm = E.D[b] - (int)((eq >> b) & 1u);
            const int v = 1 + min(min(dm, b + 1 < EXT_W ? E.D[b + 1] : EXT_INF), left);
            E.D[b] = v;
            left = v;
            const int tc = 2 * (b > EXT_B ? b - EXT_B : EXT_B - b) + (b > EXT_B ? 1 : 0);   // ties: |j - i| smallest, then the smaller j
            const int key = (v << 5) | tc;
            kmin = key < kmin ? key : kmin;
        }
    } else {
        int hi = hi64 > EXT_W ? EXT_W : (int)hi64;
        int lo = 0;
        if (M::DIAGLIM) { lo = E.blo; hi = hi < E.bhi ? hi : E.bhi; }
#pragma unroll
        for (int b = 0; b < EXT_W; b++) {
            const int dm = E.D[b] - (int)((eq >> b) & 1u);
            int v = 1 + min(min(dm, b + 1 < EXT_W ? E.D[b + 1] : EXT_INF), left);
            v = (b >= lo && b <= hi) ? v : EXT_INF;
            E.D[b] = v;
            left = v;
            const int tc = 2 * (b > EXT_B ? b - EXT_B : EXT_B - b) + (b > EXT_B ? 1 : 0);
            const int key = (v << 5) | tc;
            kmin = key < kmin ? key : kmin;
        }
    }
    const int cmin = kmin >> 5;
    if (cmin >= EXT_INF) return true;
    const int tcv = kmin & 31;
    const int tmin = i + ((tcv & 1) ? (tcv >> 1) : -(tcv >> 1));
    const int sc = M::SA * i - M::SB * cmin;
    if (sc >= E.best_s) { E.best_s = sc; E.best_i = i; E.best_t = tmin; }
    else if (sc < E.best_s - M::XDROP) return true;
    E.i = i + 1;
    return E.i > E.n;
}
template <class M>
__device__ __forceinline__ void ext_align_dev(const uint8_t *__restrict__ q, int64_t p0, int step, bool comp, int n,
                                              const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask, int64_t g0, int dir,
                                              int64_t jmax, int dlo, int dhi, int *i_out, int *t_out, int *s_out) {
    ExtStateT<M> E;
    ext_init(E, q, p0, step, comp, n, bases, nmask, g0, dir, jmax, dlo, dhi);
    if (n >= 1) while (!ext_step(E, bases, nmask)) { }
    *i_out = E.best_i; *t_out = E.best_t; *s_out = E.best_s;
}
// <<< ext_align_dev
