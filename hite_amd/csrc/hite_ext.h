// hite_ext.h -- banded end extension shared by the copy finder (hite_copies.hip: a chain is extended base by base to the ends
// of the candidate) and the tandem-repeat masker (hite_trmask.hip: a stretch is aligned with itself one period further on).
// Definition: orc_ext_align_scored in oracle/hite_oracle_copies.c.
#pragma once
#include "hite_common.h"

// the tables of ext_lut_entry live in LDS on the device (the host build of the block below defines its own pointer type)
#define EXT_LUT_PTR const __attribute__((address_space(3))) uint32_t *

// >>> ext_align_dev (tests/test_host_compiled.py compiles this block for the host and compares it with the twin)
#define EXT_B 8
#define EXT_W (2 * EXT_B + 1)
#define EXT_INF (1 << 20)
// what a user of the extension fixes at compile time: score S(i) = SA i - SB cost, abandoned XDROP below the best; where the
// query bases come from (ASCII bytes of a candidate / the packed genome itself); whether only some diagonals are allowed
struct ExtCopyMode { static constexpr int SA = 1, SB = 3, XDROP = 40; static constexpr bool PACKEDQ = false, DIAGLIM = false; };
struct ExtTandemMode { static constexpr int SA = 2, SB = 7, XDROP = 30; static constexpr bool PACKEDQ = true, DIAGLIM = true; };   // match 2, edit 5: oracle/hite_oracle_trf.c, "calibration"

// A C G T -> 0 1 2 3 (complemented: 3 2 1 0), every other byte 4.  Without a branch: bits 1..2 of the four letters are 0 1 3 2, the
// letters themselves bits 0, 2, 6, 19 of a mask over ch - 'A' (as a chain of comparisons the compiler made four branches of
// it in every column of chain_extend_kernel -- a lane alone in its wavefront pays every one of them)
__device__ __forceinline__ unsigned ext_cand_code(unsigned ch, bool comp) {
    const unsigned x = ch - 'A', h = (ch >> 1) & 3u;
    const unsigned c = (h ^ (h >> 1)) ^ (comp ? 3u : 0u);
    const bool acgt = x < 32u && ((0x80045u >> x) & 1u);
    return acgt ? c : 4u;
}
// State machine of ext_align: query bases q[p0], q[p0 + step], ... (n of them; complemented when comp) -- or, PACKEDQ, genome
// bases p0, p0 + step, ... -- against genome bases g0, g0 + 1, ... (dir = +1) or g0 - 1, g0 - 2, ... (dir = -1), at most jmax of
// them.  The 17 band cells live in registers (the loops over the band are unrolled), the genome bases under the band as
// three 17-bit planes that shift by one cell per column: a column costs one new genome base, one query base and ~8 integer
// operations per cell.  Both sequences are walked one base per column, so the words they come from are kept in registers -- 16
// genome bases, 32 mask bits, 4 query bytes per load, each fetched a few words ahead of its use (a load per base and column
// made the kernel a gather benchmark; one word ahead left the last long extensions of a batch -- a lane alone in its
// wavefront -- waiting for HBM every fourth column).
// Bit-parallel form of the band (round 4; the copy finder's mode, every cell of the band inside the contig): the 17 cells of a
// column are kept as the value of the centre cell c8 and the differences of neighbouring cells, -1 / 0 / +1, as two bit masks
// (VP / VN, bit b = cell b minus cell b - 1).  A cell and its diagonal predecessor differ by 0 or 1, and "0" runs up the column
// like a carry: D0_b = eq_b | VN_{b+1} | (VP_b & D0_{b-1}) -- one addition; the new differences follow from D0 and D0 << 1.  That
// is ~16 integer operations per column where the cell-by-cell form takes ~100.  The column minimum and the cell it sits in (ties:
// the cell nearest the centre, the lower one first) come from four 256-entry tables over 4 + 4 difference bits each (minimum,
// position and sum of a run of four steps going outwards from the centre; ext_lut_entry), 1024 words in LDS.  Cells that lie
// before the first genome base (j < 0, the first columns) carry the value i - j: larger than the cell j = 0 and consistent with
// the recurrence, so they never win and never feed a valid cell.  Near the end of the contig the state is expanded to the 17
// values and the cell-by-cell form takes over (ext_expand).
#define EXT_LUT_WORDS 1024
// table t (0: cells 9..12, 1: cells 13..16, 2: cells 7..4, 3: cells 3..0 -- going outwards), index = 4 VP bits | 4 VN bits << 4 of those
// cells' differences (in the order the masks hold them).  Entry: bits 0..4 tie code of the best cell (2 |b - 8| + (b > 8)),
// bits 5..9 its value relative to the run's start + 4, bits 12..15 value at the end of the run relative to its start + 4.
__device__ __forceinline__ uint32_t ext_lut_entry(int t, int idx) {
    const int vp = idx & 15, vn = (idx >> 4) & 15;
    int s = 0, best = 99, bestk = 0;
    for (int l = 1; l <= 4; l++) {
        // upwards the l-th step is difference bit l - 1 of the nibble (cells 9, 10, ... / 13, ...); downwards cell 8 - k is reached
        // through minus the difference of cell 9 - k: the nibble's top bit first
        const int bit = t < 2 ? l - 1 : 4 - l;
        const int d = ((vp >> bit) & 1) - ((vn >> bit) & 1);
        s += t < 2 ? d : -d;
        if (s < best) { best = s; bestk = l; }
    }
    const int k = bestk + ((t & 1) ? 4 : 0);
    const int tc = 2 * k + (t < 2 ? 1 : 0);
    return (uint32_t)tc | ((uint32_t)(best + 4) << 5) | ((uint32_t)(s + 4) << 12);
}

// (device build only) the rotated words are final HERE: without it the compiler sank the rotation below the requests as selects,
// kept the old and the new "furthest word" alive side by side, and copied one onto the other behind the load -- a wait for the
// request just made, in every column
#ifdef __HIP_DEVICE_COMPILE__
#define EXT_PIN_ROTATED(E) asm volatile("" : "+v"((E).gw), "+v"((E).gwn), "+v"((E).mw), "+v"((E).qw), "+v"((E).qwn), "+v"((E).qwn2))
// ... and a rotation stays a branch (as selects it reads the word requested last in EVERY column: a wait in every column; as a
// branch the wait sits inside it, paid in the columns where some lane rotates)
#define EXT_KEEP_BRANCH() asm volatile("")
#else
#define EXT_PIN_ROTATED(E)
#define EXT_KEEP_BRANCH()
#endif
template <class M>
struct ExtStateT {
    int D[EXT_W];
    uint32_t VP, VN;                        // bit-parallel form (fast == true): differences of neighbouring cells
    int c8;                                 //   and the value of the centre cell
    bool fast;
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
// The bases of the next column, steady state: genome base g (if want_g) and the query base at a, through the word caches.  The
// memory counter of a wavefront is in order, so a wait for a word requested 16 columns ago is also a wait for everything
// requested since: with rotate-and-request per stream (genome words, mask words, query words) the second stream's rotation
// waited for the first stream's request of the SAME column, a full trip to memory in nearly every column of a wavefront whose 64
// lanes rotate at different times.  Here every rotation (which reads words requested in EARLIER columns) comes first, then all
// the requests: they have a whole column, and the other wavefronts' columns, to arrive.
template <class M>
__device__ __forceinline__ void ext_fetch(ExtStateT<M> &E, const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask,
                                          bool want_g, int64_t g, int64_t a, unsigned &gc, unsigned &qc) {
    const int64_t wi = g >> 4, mi = g >> 5;
    const bool rg = want_g && wi != E.gwi, rm = want_g && mi != E.mwi;
    const int64_t qi = a >> 2;
    const bool rq = !M::PACKEDQ && qi != E.qwi;
    // rotate
    if (rg) { EXT_KEEP_BRANCH(); E.gw = E.gwn; E.gwn = E.gwn2; E.gwi = wi; }
    if (rm) { EXT_KEEP_BRANCH(); E.mw = E.mwn; E.mwi = mi; }
    if (rq) { EXT_KEEP_BRANCH(); E.qw = E.qwn; E.qwn = E.qwn2; E.qwn2 = E.qwn3; E.qwi = qi; }
    EXT_PIN_ROTATED(E);
    // request
    if (rg) { const int64_t nx = wi + 2 * E.dir; E.gwn2 = bases[nx > 0 ? nx : 0]; }
    if (rm) { const int64_t nx = mi + E.dir; E.mwn = nmask[nx > 0 ? nx : 0]; }
    if (!M::PACKEDQ && rq) { const int64_t nx = qi + 3 * E.step; E.qwn3 = E.q4[nx > 0 ? nx : 0]; }
    // decode
    {
        const unsigned c = (E.gw >> (2 * (int)(g & 15))) & 3u;
        gc = want_g ? (((E.mw >> (int)(g & 31)) & 1u) ? 4u : c) : 4u;
    }
    if (M::PACKEDQ) qc = a >= 0 ? ext_genome_code(bases, nmask, a) : 4u;      // (not used: the masker keeps the per-stream form, ext_step)
    else qc = ext_cand_code((E.qw >> (8 * (int)(a & 3))) & 0xffu, E.comp);
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
    // column 0 in the bit-parallel form: cell b holds |b - 8| (the cells below the centre are the virtual ones)
    E.fast = !M::DIAGLIM && jmax >= EXT_B;
    E.VP = 0x1fe00u; E.VN = 0x1feu; E.c8 = 0;
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
// the 17 cell values of the column just finished (E.i - 1) from the bit-parallel form; cells before the first genome base: EXT_INF
template <class M>
__device__ __forceinline__ void ext_expand(ExtStateT<M> &E) {
    E.D[EXT_B] = E.c8;
#pragma unroll
    for (int b = EXT_B + 1; b < EXT_W; b++) E.D[b] = E.D[b - 1] + (int)((E.VP >> b) & 1u) - (int)((E.VN >> b) & 1u);
#pragma unroll
    for (int b = EXT_B - 1; b >= 0; b--) E.D[b] = E.D[b + 1] - (int)((E.VP >> (b + 1)) & 1u) + (int)((E.VN >> (b + 1)) & 1u);
    const int first = EXT_B - (E.i - 1);                       // cell of genome position j = 0 in that column
#pragma unroll
    for (int b = 0; b < EXT_B; b++) if (b < first) E.D[b] = EXT_INF;
    E.fast = false;
}
// one column (E.i <= E.n on entry); returns true when the extension is finished (result in best_i / best_t / best_s).
// lut: the table of ext_lut_entry (unused, may be null, in the tandem mode)
template <class M>
__device__ __forceinline__ bool ext_step(ExtStateT<M> &E, const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask, EXT_LUT_PTR lut) {
    const int i = E.i;
    const unsigned gc = E.gnext_, qc = E.qnext_;
    if (i < E.n) {    // the bases of column i + 1
        const int64_t j = (int64_t)i + 1 + EXT_B;
        if (M::PACKEDQ) {     // the masker: a lane runs one extension from start to end, rotate-and-request per stream is the faster form there (measured)
            E.gnext_ = j <= E.jmax ? ext_genome_next(E, bases, nmask, E.dir > 0 ? E.g0 + j - 1 : E.g0 - j) : 4u;
            E.qnext_ = ext_query_next(E, bases, nmask, E.p0 + (int64_t)E.step * i);
        } else ext_fetch(E, bases, nmask, j <= E.jmax, E.dir > 0 ? E.g0 + j - 1 : E.g0 - j, E.p0 + (int64_t)E.step * i, E.gnext_, E.qnext_);
    }
    E.W0 = (E.W0 >> 1) | ((gc & 1u) << (EXT_W - 1)); E.W1 = (E.W1 >> 1) | (((gc >> 1) & 1u) << (EXT_W - 1)); E.WN = (E.WN >> 1) | ((gc >> 2) << (EXT_W - 1));
    const uint32_t eq = qc < 4u ? (~(E.W0 ^ (0u - (qc & 1u))) & ~(E.W1 ^ (0u - ((qc >> 1) & 1u))) & ~E.WN) : 0u;
    // cells with j < 0 need no guard: they start at EXT_INF and every move into them comes from such a cell.  Cells with
    // j > jmax (beyond the contig) are forced to EXT_INF -- only the rare extension that can reach the contig end pays for it
    const int64_t hi64 = E.jmax - i + EXT_B;                    // cells with j <= jmax
    int left = EXT_INF, kmin = 0x7fffffff;
    if (!M::DIAGLIM && E.fast && hi64 >= EXT_W - 1) {
        const uint32_t VP = E.VP, VN = E.VN;
        const uint32_t B = eq | (VN >> 1), Y = VP | B;
        const uint32_t sum = B + Y;
        const uint32_t D0 = (B | (VP & (sum ^ B ^ Y))) & 0x1ffffu;         // cells whose value equals their diagonal predecessor's
        const uint32_t S = D0 << 1;
        const uint32_t up = S & ~D0, dn = D0 & ~S, z = ~(VP | VN);
        const uint32_t nVP = ((VP & ~dn) | (z & up)) & 0x1fffeu, nVN = ((VN & ~up) | (z & dn)) & 0x1fffeu;
        E.VP = nVP; E.VN = nVN;
        E.c8 += 1 - (int)((D0 >> EXT_B) & 1u);
        const uint32_t e0 = lut[((nVP >> 9) & 15u) | (((nVN >> 9) & 15u) << 4)];
        const uint32_t e1 = lut[256u + (((nVP >> 13) & 15u) | (((nVN >> 13) & 15u) << 4))];
        const uint32_t e2 = lut[512u + (((nVP >> 5) & 15u) | (((nVN >> 5) & 15u) << 4))];
        const uint32_t e3 = lut[768u + (((nVP >> 1) & 15u) | (((nVN >> 1) & 15u) << 4))];
        const int base = (E.c8 - 4) * 32;
        const int kC = E.c8 * 32;
        const int kU1 = base + (int)(e0 & 0x3ffu), kU2 = base + ((int)(e0 >> 12) - 4) * 32 + (int)(e1 & 0x3ffu);
        const int kD1 = base + (int)(e2 & 0x3ffu), kD2 = base + ((int)(e2 >> 12) - 4) * 32 + (int)(e3 & 0x3ffu);
        kmin = min(min(min(kC, kU1), min(kU2, kD1)), kD2);
    } else if (!M::DIAGLIM && hi64 >= EXT_W - 1) {
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
        if (!M::DIAGLIM && E.fast) ext_expand(E);          // the band reaches the end of the contig: cell by cell from here on
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
                                              int64_t jmax, int dlo, int dhi, int *i_out, int *t_out, int *s_out, EXT_LUT_PTR lut) {
    ExtStateT<M> E;
    ext_init(E, q, p0, step, comp, n, bases, nmask, g0, dir, jmax, dlo, dhi);
    if (n >= 1) while (!ext_step(E, bases, nmask, lut)) { }
    *i_out = E.best_i; *t_out = E.best_t; *s_out = E.best_s;
}
// <<< ext_align_dev
