// hite_align.hip -- pairwise alignment of every copy window to the centre window of its candidate: the compute core of
// the star alignment that stands where the reference shells out to `mafft` (/root/reference/module/Util.py:10416).
//
// Definition (oracle/hite_oracle_nw.c): optimal GLOBAL alignment, mismatch 1, gap AL_GAP = 3 per base, canonical
// traceback diagonal > up > left.  How it is computed here (twin: oracle/hite_oracle_msa.c, byte for byte):
//
//   * one THREAD per (row, centre) pair, bit-parallel dynamic programming over an adaptive band of W = 32 NW centre
//     rows per column of the row sequence.  The state is the vertical difference of neighbouring cells, -3 .. +3,
//     kept as three bit planes (difference + 3) per 32 rows.  As in Myers' algorithm for unit costs, the diagonal
//     difference is 0 or 1, and "0" propagates up a column exactly like a carry: Z = B | (P & (Z << 1)) with
//     B = match | (vertical difference -3 in the previous column), P = (vertical difference +3) one row below; ONE
//     multi-word addition resolves the chain.  The horizontal and the new vertical differences are then two
//     bit-sliced subtractions "6 + (1 - Z) - x".  The band state and the centre's three bit planes (2-bit code +
//     "never matches") live in registers, pre-shifted to the band's rows; one column costs about 30 NW + 40 integer
//     instructions for 32 NW cells.  The centre planes are built once per candidate (align_planes_kernel) and shared by
//     all of its rows; the row bases are read 16 at a time.
//   * steering (the band follows the valley of the cost surface: it moves in every fourth column, by 0, 4 or 8 rows, so that
//     the register shifts, the steering and the book-keeping are paid once per four columns), pessimistic band edges and
//     the Ukkonen certificate are described in the twin's header.  A certified pair IS the alignment of the definition.
//   * no per-cell direction is ever written to HBM.  The forward pass keeps, per strip of 16 columns, a check point of
//     the SLICE (the middle 64 rows of the band: 32 B) and 2 B per column of
//     boundary information (the step of the band, the bits that enter the slice from the rest of the band).  The
//     traceback pass (align_tb_kernel) re-computes the slice strip by strip into registers and walks it backwards:
//     4 B of HBM traffic per column instead of 2 bits per DP cell.  Both passes fetch what the next strip needs one strip
//     ahead (software pipeline); ops leave through a 64-bit shift register per lane (8-byte stores).
//   * schedule per pair (hite_align_run): band of 4 words; in exact mode a pair that is not certified is re-run with
//     8 / 16 / 32 words until it is; a traceback that leaves the slice falls back to a 64-word band computed by one
//     wavefront per pair, whose traceback bits are kept whole (rare: an insertion / deletion longer than about 60 bases);
//     a row that even this cannot align is dropped from the alignment.
#include "hite_common.h"
#include "hite_scan.h"
#include "hite_arena.h"
#include "hite_sort.h"
#include "hite_align.h"

#define AL_GAP 3
#define AL_STEER 24
#define AL_MARGIN 48
#define AL_PADR 1056         // virtual rows above row 1 in the centre planes (33 words: covers t_0 = -1024 of the fall-back band)
#define AL_WIDE_NW 64         // words of the fall-back band (one per lane of a wavefront)
#define AL_STRIP 16
#define AL_RECB 4096          // bytes of strip records per strip and block of 64 pairs: 64 check points, then 64 boundary records
#define AL_DEFAULT_CAP 8      // exact mode: widest band (words) tried for a certificate unless configured otherwise
#define AL_KBINS 2048        // strips per pair <= 32767 / 16 + 1
#define AL_LANES_MIN_STRIPS 32     // lane-parallel kernels: never when the longest pair has fewer than 512 columns (automatic mode)

struct AlignArgs {
    const uint8_t *win;
    const int64_t *win_off;
    const int32_t *win_len;
    const int32_t *row_first;   // n_cand + 1
    int n_cand;
    const int32_t *row_cand;    // per row: its candidate
    const uint4 *planes;        // per candidate, per 32 centre rows: (plane0, plane1, planeN, 0)
    const int64_t *plane_off;   // n_cand
    // strip records.  The 64 pairs of a wavefront of the forward pass keep their records interleaved: a block of the run's list
    // owns AL_RECB bytes per strip -- 64 check points of 32 B (the three planes of the slice, 2 words each, t of the slice, 1 spare),
    // then 64 boundary records of 32 B (four 64-bit groups of four columns) -- so that a wavefront writes and reads 2 KB runs
    // (per-pair runs of 32 B were re-fetched from HBM: a 128-B line serves four strips only while it stays in the L2)
    const unsigned long long *rec;   // per row: address of its check point of strip 0 (written by the run that is kept)
    int32_t *U, *kst, *st, *lvl, *U4;   // per row: cost, certificate bound, status, band words of the run kept, cost of the 4-word run
    const int64_t *ops_base;
    uint16_t *ops;
    const int64_t *full_off;    // per row (fall-back): first column record
    uint32_t *fullbuf;          // per column: dg[64], up[64]
    int32_t *fullt;             // per column: t
};

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
// add with the carry-out as a wave mask / add this lane's bit of a wave mask (the carry chains that run along the lanes)
__device__ __forceinline__ uint32_t add_co_mask(uint32_t a, uint32_t b, unsigned long long &carry_mask) {
    uint32_t r;
    asm volatile("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(r), "=s"(carry_mask) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t add_mask_bit(uint32_t a, unsigned long long mask) {     // a + (this lane's bit of mask)
    uint32_t r;
    unsigned long long dummy;
    asm volatile("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r), "=s"(dummy) : "v"(a), "s"(mask));
    return r;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { int o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}
// (a row byte in lower case -- a pad that carries a base, HITE_IS_ROW_PAD -- is its base: bit 5 is dropped; bits 1 and 2, the 2-bit code, are the same)
__device__ __forceinline__ bool is_acgt_byte(unsigned c) { c &= 0xdfu; return c - 0x41u < 32u && ((0x00080045u >> (c - 0x41u)) & 1u); }

// one row base as the three masks that turn the centre planes into the match vector
struct BaseMask { uint32_t m0, m1, inv; };
__device__ __forceinline__ BaseMask base_mask(unsigned ch) {
    BaseMask b;
    b.m0 = (uint32_t)(-(int)((ch >> 1) & 1u));
    b.m1 = (uint32_t)(-(int)((ch >> 2) & 1u));
    b.inv = is_acgt_byte(ch) ? 0u : 0xffffffffu;
    return b;
}

// One column of the recurrence on NW words, carries running from word 0 (lowest rows) upwards.  X2/X1/X0: bit planes of
// (vertical difference + 3) of the previous column, replaced by those of this column.
// carry[0]: carry into the addition of word 0; carry[1]: "vertical difference +3" of the row below word 0's first row;
// carry[2..4]: planes of the horizontal difference of that row (+3 above the band: 1, 1, 0).
// TAP >= 0: the carries that ENTER word TAP are returned in tap[0..4] (boundary information of the slice).
template <int NW, bool DIRS, int TAP>
__device__ __forceinline__ void bp_core(uint32_t (&X2)[NW], uint32_t (&X1)[NW], uint32_t (&X0)[NW], const uint32_t (&A0)[NW],
                                        const uint32_t (&A1)[NW], const uint32_t (&AN)[NW], const BaseMask bm, uint32_t cin, uint32_t vpc,
                                        uint32_t h2c, uint32_t h1c, uint32_t h0c, uint32_t *dg, uint32_t *up, uint32_t *tap) {
#pragma unroll
    for (int w = 0; w < NW; w++) {
        if (TAP >= 0 && w == TAP) { tap[0] = cin; tap[1] = vpc; tap[2] = h2c; tap[3] = h1c; tap[4] = h0c; }
        const uint32_t eq = ~((A0[w] ^ bm.m0) | (A1[w] ^ bm.m1) | AN[w] | bm.inv);
        const uint32_t x2 = X2[w], x1 = X1[w], x0 = X0[w];
        const uint32_t vneg = ~(x2 | x1 | x0), vpos = x2 & x1;
        const uint32_t B = eq | vneg, P = (vpos << 1) | vpc;
        vpc = vpos >> 31;
        const uint32_t Y = P | B;
        unsigned cout;                                              // (add-with-carry: one v_addc_co_u32 per word)
        const uint32_t sum = __builtin_addc(B, Y, cin, &cout);
        cin = cout;
        const uint32_t Z = B | (P & (sum ^ B ^ Y));                  // diagonal difference 0
        // horizontal difference + 3 = 6 + (1 - Z) - x
        const uint32_t h0 = ~(Z ^ x0), b1 = Z & x0, h1 = ~(x1 ^ b1), b2 = x1 & b1, h2 = ~(x2 ^ b2);
        const uint32_t s2 = (h2 << 1) | h2c, s1 = (h1 << 1) | h1c, s0 = (h0 << 1) | h0c;
        h2c = h2 >> 31; h1c = h1 >> 31; h0c = h0 >> 31;
        // new vertical difference + 3 = 6 + (1 - Z) - (horizontal difference of the row below + 3)
        const uint32_t n0 = ~(Z ^ s0), c1 = Z & s0, n1 = ~(s1 ^ c1), c2 = s1 & c1, n2 = ~(s2 ^ c2);
        X2[w] = n2; X1[w] = n1; X0[w] = n0;
        if (DIRS) { dg[w] = eq | ~Z; up[w] = n2 & n1; }
    }
}
// rows whose value exceeds the row below them minus rows whose value is below it (the steering signal), of one word
__device__ __forceinline__ int slope_count(uint32_t x2, uint32_t x1, uint32_t x0) { return __popc(x2) - __popc(~x2 & ~(x1 & x0)); }
// sum of (vertical difference + 3) over the rows selected by mask
__device__ __forceinline__ int plane_sum(uint32_t x2, uint32_t x1, uint32_t x0, uint32_t mask) {
    return 4 * __popc(x2 & mask) + 2 * __popc(x1 & mask) + __popc(x0 & mask);
}

// ---------------------------------------------------------------------------------------------
// preparation
// ---------------------------------------------------------------------------------------------
// candidate of every row, strips of every pair (0 for a centre), sort keys for the length order (longest first: lanes of one
// wavefront run in lockstep, neighbours should be equally long)
__global__ void __launch_bounds__(256) align_rows_kernel(int n_cand, const int32_t *__restrict__ row_first,
                                                         const int32_t *__restrict__ win_len, int32_t *__restrict__ row_cand,
                                                         int32_t *__restrict__ strips, unsigned long long *__restrict__ keys,
                                                         unsigned *__restrict__ vals) {
    const int c = blockIdx.x;
    if (c >= n_cand) return;
    const int g0 = row_first[c], g1 = row_first[c + 1];
    for (int g = g0 + threadIdx.x; g < g1; g += 256) {
        row_cand[g] = c;
        const int k = g == g0 ? 0 : (win_len[g] + AL_STRIP - 1) / AL_STRIP;
        strips[g] = k;
        keys[g] = (unsigned long long)(AL_KBINS - 1 - (k >= AL_KBINS ? AL_KBINS - 1 : k));
        vals[g] = (unsigned)g;
    }
}
__global__ void align_plane_words_kernel(int n_cand, const int32_t *__restrict__ row_first, const int32_t *__restrict__ win_len,
                                         int32_t *__restrict__ words) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cand) return;
    const int g0 = row_first[c];
    words[c] = row_first[c + 1] > g0 ? (win_len[g0] + 2 * AL_PADR) / 32 + 3 : 0;
}
// bit x of a candidate's planes = centre row r = x - AL_PADR + 1 (r <= 0: virtual rows, r > m: padding: never match)
__global__ void __launch_bounds__(256) align_planes_kernel(int n_cand, const uint8_t *__restrict__ win, const int64_t *__restrict__ win_off,
                                                           const int32_t *__restrict__ win_len, const int32_t *__restrict__ row_first,
                                                           const int64_t *__restrict__ plane_off, uint4 *__restrict__ planes) {
    const int c = blockIdx.x;
    if (c >= n_cand) return;
    const int g0 = row_first[c];
    if (row_first[c + 1] <= g0) return;
    const int m = win_len[g0];
    const uint8_t *a = win + win_off[g0];
    const int nw = (m + 2 * AL_PADR) / 32 + 3;
    uint4 *out = planes + plane_off[c];
    for (int wi = threadIdx.x; wi < nw; wi += 256) {
        uint32_t p0 = 0, p1 = 0, pn = 0;
        unsigned chs[32];      // (unconditional loads from a clamped index: all 32 in flight together)
#pragma unroll
        for (int k = 0; k < 32; k++) {
            const int r = wi * 32 + k - AL_PADR + 1;
            chs[k] = a[r < 1 || m < 1 ? 0 : (r > m ? m - 1 : r - 1)];
        }
#pragma unroll
        for (int k = 0; k < 32; k++) {
            const int r = wi * 32 + k - AL_PADR + 1;
            const unsigned ch = (r >= 1 && r <= m) ? chs[k] : 0u;
            if (is_acgt_byte(ch)) { p0 |= ((ch >> 1) & 1u) << k; p1 |= ((ch >> 2) & 1u) << k; }
            else pn |= 1u << k;
        }
        out[wi] = make_uint4(p0, p1, pn, 0u);
    }
}

// ---------------------------------------------------------------------------------------------
// forward pass
// ---------------------------------------------------------------------------------------------
#ifdef ALIGN_CLOCKS
// development aid (-DALIGN_CLOCKS, tools/align_wave_hist.py): start / end of every wavefront of the last align_fwd_kernel<4> and
// align_tb_kernel launches on the constant 100 MHz clock, + the SIMD it ran on -- separates the tail of the longest wavefronts
// from stalls inside them
#define ALIGN_CLK_MAX (1 << 17)
__device__ unsigned long long g_aclk[2][4][ALIGN_CLK_MAX];     // start, end, xcc << 16 | hw_id, columns of the longest lane
extern "C" int hite_debug_align_clocks(int which, unsigned long long *out, int n) {
    if (which < 0 || which > 1 || n > ALIGN_CLK_MAX) return -1;
    for (int k = 0; k < 4; k++)
        if (hipMemcpyFromSymbol(out + (size_t)k * n, HIP_SYMBOL(g_aclk), sizeof(unsigned long long) * n,
                                ((size_t)which * 4 + k) * ALIGN_CLK_MAX * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long *z = (unsigned long long *)calloc((size_t)4 * ALIGN_CLK_MAX, sizeof(unsigned long long));      // next launch starts clean
    if (z) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_aclk), z, sizeof(unsigned long long) * 4 * ALIGN_CLK_MAX, (size_t)which * 4 * ALIGN_CLK_MAX * sizeof(unsigned long long)); free(z); }
    return 0;
}
__device__ __forceinline__ unsigned aclk_hw_id() {
    unsigned v, x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return ((x & 0xfu) << 16) | (v & 0xffffu);
}
#define ACLK_BEGIN(which) const unsigned long long aclk_t0 = wall_clock64(); const int aclk_w = (which)
#define ACLK_END() do { if (threadIdx.x == 0 && blockIdx.x < ALIGN_CLK_MAX) { g_aclk[aclk_w][0][blockIdx.x] = aclk_t0; \
    g_aclk[aclk_w][1][blockIdx.x] = wall_clock64(); g_aclk[aclk_w][2][blockIdx.x] = aclk_hw_id(); g_aclk[aclk_w][3][blockIdx.x] = (unsigned long long)nmax; } } while (0)
#else
#define ACLK_BEGIN(which) do { } while (0)
#define ACLK_END() do { } while (0)
#endif

template <int NW>
__device__ __forceinline__ void align_fwd_body(const AlignArgs &P, const int32_t *__restrict__ list, int nlist) {
    constexpr int W = 32 * NW, H = W / 2, S0 = NW / 2 - 1;
    ACLK_BEGIN(0);
    const int li = blockIdx.x * 64 + threadIdx.x;
    const int g = li < nlist ? list[li] : -1;
    int m = 0, n = 0, g0 = 0, c = 0;
    if (g >= 0) {
        c = P.row_cand[g];
        g0 = P.row_first[c];
        if (g != g0) { m = P.win_len[g0]; n = P.win_len[g]; }
    }
    const int nmax = wave_max_i32(n);
    if (nmax == 0) return;
    const uint8_t *b = P.win + (g >= 0 ? P.win_off[g] : 0);
    const uint4 *pl = P.planes + (g >= 0 ? P.plane_off[c] : 0);
    char *rec0 = g >= 0 ? reinterpret_cast<char *>(P.rec[g]) : nullptr;
    uint32_t X2[NW], X1[NW], X0[NW], A0[NW], A1[NW], AN[NW];
    int t = -H;
#pragma unroll
    for (int w = 0; w < NW; w++) { X2[w] = w >= NW / 2 ? 0xffffffffu : 0u; X1[w] = X2[w]; X0[w] = 0u; }   // +3 from row 1 on, -3 above
    bool act = n > 0;
    if (act) {   // planes of rows t+1 .. t+W: bit position t + AL_PADR (a multiple of 32 here)
        const int q0 = (t + AL_PADR) >> 5;
#pragma unroll
        for (int w = 0; w < NW; w++) { const uint4 v = pl[q0 + w]; A0[w] = v.x; A1[w] = v.y; AN[w] = v.z; }
    } else {
#pragma unroll
        for (int w = 0; w < NW; w++) { A0[w] = 0; A1[w] = 0; AN[w] = 0xffffffffu; }
    }
    int stop = AL_GAP * H, LO = -(1 << 28), HI = 1 << 28, status = 0;
    // what a strip needs from memory is fetched one strip ahead (the loads of a strip used to be its exposed latency): the 16
    // row bases of the next strip, and a window of three plane words Wa, Wb, Wc = words fq, fq + 1, fq + 2 of the rows that
    // enter below the band (the band moves <= 32 rows per strip, so the next strip starts in word fq or fq + 1)
    int fq = (t + W + AL_PADR) >> 5;
    uint4 Wa = make_uint4(0, 0, 0xffffffffu, 0), Wb = Wa, Wc = Wa, bnext = make_uint4(0, 0, 0, 0);
    if (act) { Wa = pl[fq]; Wb = pl[fq + 1]; Wc = pl[fq + 2]; bnext = *reinterpret_cast<const uint4 *>(b); }
    for (int k = 0; k * AL_STRIP < nmax; k++) {
        const bool sa = act && k * AL_STRIP < n;
        uint4 bw = make_uint4(0, 0, 0, 0);
        uint32_t f0 = 0, f1 = 0, fn = 0;   // the next 32 rows below the band, per plane
        uint32_t brec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (sa) {
            uint4 *ck = reinterpret_cast<uint4 *>(rec0 + (size_t)k * AL_RECB);
            ck[0] = make_uint4(X2[S0], X2[S0 + 1], X1[S0], X1[S0 + 1]);
            ck[1] = make_uint4(X0[S0], X0[S0 + 1], (uint32_t)(t + 32 * S0), 0u);
            bw = bnext;
            if ((k + 1) * AL_STRIP < n) bnext = *reinterpret_cast<const uint4 *>(b + (k + 1) * AL_STRIP);
            const int fx = t + W + AL_PADR;
            if ((fx >> 5) != fq) { Wa = Wb; Wb = Wc; fq++; Wc = pl[fq + 2]; }      // (at most one word further than a strip ago)
            const uint32_t sh = (uint32_t)fx & 31u;
            f0 = alignbit(Wb.x, Wa.x, sh); f1 = alignbit(Wb.y, Wa.y, sh); fn = alignbit(Wb.z, Wa.z, sh);
        }
#pragma unroll
        for (int cc = 0; cc < AL_STRIP; cc++) {
            const int j = k * AL_STRIP + cc + 1;
            if (sa && j <= n) {
                if ((cc & 3) == 0) {
                    // ---- the band moves in every fourth column only, by 0 / 4 / 8 rows: steering from column j-1 (middle 64
                    //      rows), clamps (never beyond the row m - H; far enough for the moves still to come)
                    const int ds = slope_count(X2[S0], X1[S0], X0[S0]) + slope_count(X2[S0 + 1], X1[S0 + 1], X0[S0 + 1]);
                    int s = ds > AL_STEER ? 0 : (ds < -AL_STEER ? 8 : 4);
                    const int tr = m - H;
                    if (t + s > tr) s = (tr - t) & ~3;
                    const int need = tr - 3 - 8 * ((n - j) >> 2) - t;
                    if (s < need) s = (need + 3) & ~3;
                    if (s > 8) { status = 2; act = false; s = 8; }
                    stop += plane_sum(X2[0], X1[0], X0[0], (1u << s) - 1u) - AL_GAP * s;
                    unsigned long long rec = (unsigned long long)(s >> 2);
                    if (NW > 2) {
                        rec |= (unsigned long long)(X2[S0 + 2 < NW ? S0 + 2 : 0] & 0xffu) << 2;
                        rec |= (unsigned long long)(X1[S0 + 2 < NW ? S0 + 2 : 0] & 0xffu) << 10;
                        rec |= (unsigned long long)(X0[S0 + 2 < NW ? S0 + 2 : 0] & 0xffu) << 18;
                    }
                    brec[(cc >> 2) * 2] = (uint32_t)rec;
                    // rows that enter: +3
#pragma unroll
                    for (int w = 0; w < NW - 1; w++) {
                        X2[w] = alignbit(X2[w + 1], X2[w], (uint32_t)s); X1[w] = alignbit(X1[w + 1], X1[w], (uint32_t)s);
                        X0[w] = alignbit(X0[w + 1], X0[w], (uint32_t)s);
                        A0[w] = alignbit(A0[w + 1], A0[w], (uint32_t)s); A1[w] = alignbit(A1[w + 1], A1[w], (uint32_t)s);
                        AN[w] = alignbit(AN[w + 1], AN[w], (uint32_t)s);
                    }
                    X2[NW - 1] = alignbit(0xffffffffu, X2[NW - 1], (uint32_t)s);
                    X1[NW - 1] = alignbit(0xffffffffu, X1[NW - 1], (uint32_t)s);
                    X0[NW - 1] = X0[NW - 1] >> s;
                    A0[NW - 1] = alignbit(f0, A0[NW - 1], (uint32_t)s); A1[NW - 1] = alignbit(f1, A1[NW - 1], (uint32_t)s);
                    AN[NW - 1] = alignbit(fn, AN[NW - 1], (uint32_t)s);
                    f0 >>= s; f1 >>= s; fn >>= s;
                    t += s;
                    if (t >= 1) { const int v = t + 1 - j; LO = v > LO ? v : LO; }
                    if (t + W < m) { const int jl = j + 3 < n ? j + 3 : n; const int v = t + W - jl; HI = v < HI ? v : HI; }
                }
                stop += AL_GAP;
                // ---- column
                const uint32_t word = cc < 4 ? bw.x : (cc < 8 ? bw.y : (cc < 12 ? bw.z : bw.w));
                const BaseMask bm = base_mask((word >> (8 * (cc & 3))) & 0xffu);
                uint32_t tap[5] = {0, 0, 1, 1, 0};
                bp_core<NW, false, (NW > 2 ? S0 : -1)>(X2, X1, X0, A0, A1, AN, bm, 0u, 0u, 1u, 1u, 0u, nullptr, nullptr, tap);
                if (NW > 2) {
                    // 5 carry bits per column: bits 26 + 5 c of the group's 64-bit record
                    const uint32_t c5 = tap[0] | (tap[1] << 1) | (tap[2] << 2) | (tap[3] << 3) | (tap[4] << 4);
                    const int sh = 26 + 5 * (cc & 3);
                    if (sh + 5 <= 32) brec[(cc >> 2) * 2] |= c5 << sh;
                    else if (sh >= 32) brec[(cc >> 2) * 2 + 1] |= c5 << (sh - 32);
                    else { brec[(cc >> 2) * 2] |= c5 << sh; brec[(cc >> 2) * 2 + 1] |= c5 >> (32 - sh); }
                    // the row base as the traceback needs it (3 bits: the two code bits, "never matches"): bits 46 + 3 c
                    brec[(cc >> 2) * 2 + 1] |= ((bm.m0 & 1u) | (bm.m1 & 2u) | (bm.inv & 4u)) << (14 + 3 * (cc & 3));
                }
            }
        }
        if (NW > 2 && sa) {
            uint4 *bp = reinterpret_cast<uint4 *>(rec0 + (size_t)k * AL_RECB + AL_RECB / 2);
            bp[0] = make_uint4(brec[0], brec[1], brec[2], brec[3]);
            bp[1] = make_uint4(brec[4], brec[5], brec[6], brec[7]);
        }
    }
    if (n > 0) {
        int U = -1, kstar = -1;
        if (status == 0) {
            int u = stop - AL_GAP * H;   // row m is bit H - 1 + (m - H - t_n), 0 .. 3 bits into word NW / 2
#pragma unroll
            for (int w = 0; w < NW / 2; w++) u += plane_sum(X2[w], X1[w], X0[w], 0xffffffffu);
            {
                const int extra = m - H - t;          // 0 .. 3
                u += plane_sum(X2[NW / 2], X1[NW / 2], X0[NW / 2], (1u << extra) - 1u) - AL_GAP * extra;
            }
            U = u;
            const int d = m - n, dmin = d < 0 ? d : 0, dmax = d > 0 ? d : 0, ad = d < 0 ? -d : d;
            int E = dmin - LO;
            if (HI - dmax < E) E = HI - dmax;
            if (E > (1 << 27)) kstar = 0x7fffffff;
            else if (E >= 0) kstar = AL_GAP * ad + 2 * AL_GAP * E + 2 * AL_GAP - 1;
        }
        P.U[g] = U; P.kst[g] = kstar; P.st[g] = status; P.lvl[g] = NW;
        if (NW == 4) P.U4[g] = U;
    }
    if (NW == 4) ACLK_END();
}
template <int NW>
__global__ void __launch_bounds__(64) align_fwd_kernel(AlignArgs P, const int32_t *__restrict__ list, int nlist) { align_fwd_body<NW>(P, list, nlist); }
// the 4-word level (every pair runs it): its own kernel, so that its occupancy can be set apart from the wider levels'.
// tools/align_wave_hist.py (profiles/r04_align_wave_hist.txt): a wavefront needs ~1100 ns per column with one neighbour on its
// SIMD and ~1260 ns with three or four -- most of its time is its own dependency chain.  More wavefronts per SIMD do not pay for
// their spills, though (C3, ms per step of this kernel): 5 per SIMD (96 registers, no spill) 13.8; 6 (80, 24 spilled) 14.6;
// 7 (72, 44 spilled) 15.8.
#ifdef ALIGN_FWD4_OCC
#define ALIGN_FWD4_ATTR __attribute__((amdgpu_waves_per_eu(ALIGN_FWD4_OCC, ALIGN_FWD4_OCC)))
#else
#define ALIGN_FWD4_ATTR
#endif
__global__ void __launch_bounds__(64) ALIGN_FWD4_ATTR align_fwd4_kernel(AlignArgs P, const int32_t *__restrict__ list, int nlist) { align_fwd_body<4>(P, list, nlist); }

// ---------------------------------------------------------------------------------------------
// the 8-word level with TWO LANES PER PAIR (lane 2i: words 0-3, lane 2i+1: words 4-7 of pair i).  The pairs that reach this
// level are few (13 % on C3: 1.2 wavefronts per SIMD) and long, so the kernel's time is the instruction chain of its
// longest wavefront (300 instructions per column at the 4.4 cycles a lone wavefront issues at), not its work: half the
// words per lane halve that chain.  Same recurrence, same records, bit for bit, as align_fwd_kernel<8>; what crosses
// the lane boundary: the carry of the addition (the upper lane adds with carry-in 0, then ripples the lower lane's
// carry-out through its sums), bit 31 of vpos / h2 / h1 / h0 of word 3, the words that move across when the band moves,
// the steering count of word 4, and word 4's planes for the check point -- quad-permute DPP moves.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pair_partner(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); }   // lane ^ 1

__global__ void __launch_bounds__(64) align_fwd8_pair_kernel(AlignArgs P, const int32_t *__restrict__ list, int nlist) {
    constexpr int NW = 8, HW = 4, W = 32 * NW, H = W / 2, S0 = NW / 2 - 1;
    const int half = threadIdx.x & 1;                    // 0: words 0..3 (writes the records and the results), 1: words 4..7
    const int li = blockIdx.x * 32 + (threadIdx.x >> 1);
    const int g = li < nlist ? list[li] : -1;
    int m = 0, n = 0, g0 = 0, c = 0;
    if (g >= 0) {
        c = P.row_cand[g];
        g0 = P.row_first[c];
        if (g != g0) { m = P.win_len[g0]; n = P.win_len[g]; }
    }
    const int nmax = wave_max_i32(n);
    if (nmax == 0) return;
    const uint8_t *b = P.win + (g >= 0 ? P.win_off[g] : 0);
    const uint4 *pl = P.planes + (g >= 0 ? P.plane_off[c] : 0);
    char *rec0 = g >= 0 ? reinterpret_cast<char *>(P.rec[g]) : nullptr;
    uint32_t X2[HW], X1[HW], X0[HW], A0[HW], A1[HW], AN[HW];
    int t = -H;
#pragma unroll
    for (int w = 0; w < HW; w++) { X2[w] = half ? 0xffffffffu : 0u; X1[w] = X2[w]; X0[w] = 0u; }   // +3 from row 1 on, -3 above
    bool act = n > 0;
    if (act) {
        const int q0 = ((t + AL_PADR) >> 5) + half * HW;
#pragma unroll
        for (int w = 0; w < HW; w++) { const uint4 v = pl[q0 + w]; A0[w] = v.x; A1[w] = v.y; AN[w] = v.z; }
    } else {
#pragma unroll
        for (int w = 0; w < HW; w++) { A0[w] = 0; A1[w] = 0; AN[w] = 0xffffffffu; }
    }
    int stop = AL_GAP * H, LO = -(1 << 28), HI = 1 << 28, status = 0;
    int fq = (t + W + AL_PADR) >> 5;
    uint4 Wa = make_uint4(0, 0, 0xffffffffu, 0), Wb = Wa, Wc = Wa, bnext = make_uint4(0, 0, 0, 0);
    if (act) { Wa = pl[fq]; Wb = pl[fq + 1]; Wc = pl[fq + 2]; bnext = *reinterpret_cast<const uint4 *>(b); }
    for (int k = 0; k * AL_STRIP < nmax; k++) {
        const bool sa = act && k * AL_STRIP < n;
        uint4 bw = make_uint4(0, 0, 0, 0);
        uint32_t f0 = 0, f1 = 0, fn = 0;
        uint32_t brec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        {   // check point: words 3 and 4 of the pair = word 3 of the lower lane, word 0 of the upper lane
            const uint32_t p2 = pair_partner(X2[0]), p1 = pair_partner(X1[0]), p0 = pair_partner(X0[0]);
            if (sa && !half) {
                uint4 *ck = reinterpret_cast<uint4 *>(rec0 + (size_t)k * AL_RECB);
                ck[0] = make_uint4(X2[HW - 1], p2, X1[HW - 1], p1);
                ck[1] = make_uint4(X0[HW - 1], p0, (uint32_t)(t + 32 * S0), 0u);
            }
        }
        if (sa) {
            bw = bnext;
            if ((k + 1) * AL_STRIP < n) bnext = *reinterpret_cast<const uint4 *>(b + (k + 1) * AL_STRIP);
            const int fx = t + W + AL_PADR;
            if ((fx >> 5) != fq) { Wa = Wb; Wb = Wc; fq++; Wc = pl[fq + 2]; }
            const uint32_t sh = (uint32_t)fx & 31u;
            f0 = alignbit(Wb.x, Wa.x, sh); f1 = alignbit(Wb.y, Wa.y, sh); fn = alignbit(Wb.z, Wa.z, sh);
        }
#pragma unroll
        for (int cc = 0; cc < AL_STRIP; cc++) {
            const int j = k * AL_STRIP + cc + 1;
            const bool on = sa && j <= n;                // (the same in both lanes of a pair; DPP moves run for every lane)
            if ((cc & 3) == 0) {
                // ---- band move: steering count of words 3 + 4, the words that cross the lane boundary
                const int own = half ? slope_count(X2[0], X1[0], X0[0]) : slope_count(X2[HW - 1], X1[HW - 1], X0[HW - 1]);
                const int ds = own + (int)pair_partner((uint32_t)own);
                const uint32_t i2 = pair_partner(X2[1] & 0xffu), i1 = pair_partner(X1[1] & 0xffu), i0 = pair_partner(X0[1] & 0xffu);   // word 5's low rows (upper lane's word 1)
                const uint32_t u2 = pair_partner(X2[0]), u1 = pair_partner(X1[0]), u0 = pair_partner(X0[0]);
                const uint32_t ua0 = pair_partner(A0[0]), ua1 = pair_partner(A1[0]), uan = pair_partner(AN[0]);
                if (on) {
                    int s = ds > AL_STEER ? 0 : (ds < -AL_STEER ? 8 : 4);
                    const int tr = m - H;
                    if (t + s > tr) s = (tr - t) & ~3;
                    const int need = tr - 3 - 8 * ((n - j) >> 2) - t;
                    if (s < need) s = (need + 3) & ~3;
                    if (s > 8) { status = 2; act = false; s = 8; }
                    if (!half) {
                        stop += plane_sum(X2[0], X1[0], X0[0], (1u << s) - 1u) - AL_GAP * s;
                        brec[(cc >> 2) * 2] = (uint32_t)(s >> 2) | (i2 << 2) | (i1 << 10) | (i0 << 18);
                    }
#pragma unroll
                    for (int w = 0; w < HW - 1; w++) {
                        X2[w] = alignbit(X2[w + 1], X2[w], (uint32_t)s); X1[w] = alignbit(X1[w + 1], X1[w], (uint32_t)s);
                        X0[w] = alignbit(X0[w + 1], X0[w], (uint32_t)s);
                        A0[w] = alignbit(A0[w + 1], A0[w], (uint32_t)s); A1[w] = alignbit(A1[w + 1], A1[w], (uint32_t)s);
                        AN[w] = alignbit(AN[w + 1], AN[w], (uint32_t)s);
                    }
                    // the top word of a lane: the lower lane takes the upper lane's word 0 (before its move), the upper lane the rows that enter
                    X2[HW - 1] = alignbit(half ? 0xffffffffu : u2, X2[HW - 1], (uint32_t)s);
                    X1[HW - 1] = alignbit(half ? 0xffffffffu : u1, X1[HW - 1], (uint32_t)s);
                    X0[HW - 1] = alignbit(half ? 0u : u0, X0[HW - 1], (uint32_t)s);
                    A0[HW - 1] = alignbit(half ? f0 : ua0, A0[HW - 1], (uint32_t)s);
                    A1[HW - 1] = alignbit(half ? f1 : ua1, A1[HW - 1], (uint32_t)s);
                    AN[HW - 1] = alignbit(half ? fn : uan, AN[HW - 1], (uint32_t)s);
                    f0 >>= s; f1 >>= s; fn >>= s;
                    t += s;
                    if (t >= 1) { const int v = t + 1 - j; LO = v > LO ? v : LO; }
                    if (t + W < m) { const int jl = j + 3 < n ? j + 3 : n; const int v = t + W - jl; HI = v < HI ? v : HI; }
                }
            }
            // ---- column.  Phase 1: match vector, B, P, Y and the sums with carry-in 0, word by word inside the lane
            const uint32_t word = cc < 4 ? bw.x : (cc < 8 ? bw.y : (cc < 12 ? bw.z : bw.w));
            const BaseMask bm = base_mask((word >> (8 * (cc & 3))) & 0xffu);
            uint32_t Bv[HW], Pv[HW], Yv[HW], Sv[HW], eqv[HW], vposv[HW];
#pragma unroll
            for (int w = 0; w < HW; w++) {
                eqv[w] = ~((A0[w] ^ bm.m0) | (A1[w] ^ bm.m1) | AN[w] | bm.inv);
                vposv[w] = X2[w] & X1[w];
                Bv[w] = eqv[w] | ~(X2[w] | X1[w] | X0[w]);
            }
            const uint32_t pvpos = pair_partner(vposv[HW - 1]);          // (every lane takes part in a DPP move: no lane-dependent branch around it)
            const uint32_t vin = half ? pvpos >> 31 : 0u;
            uint32_t cin = 0u, tap0 = 0u;
#pragma unroll
            for (int w = 0; w < HW; w++) {
                if (w == HW - 1) tap0 = cin;               // lower lane: the carry that enters word 3 (exact: its carry-in is 0)
                Pv[w] = (vposv[w] << 1) | (w ? vposv[w - 1] >> 31 : vin);
                Yv[w] = Pv[w] | Bv[w];
                unsigned cout;
                Sv[w] = __builtin_addc(Bv[w], Yv[w], cin, &cout);
                cin = cout;
            }
            {   // the upper lane's carry-in = the lower lane's carry-out: ripple it through sums that are all ones
                const uint32_t pc = pair_partner(cin);
                uint32_t e = half ? pc : 0u;
#pragma unroll
                for (int w = 0; w < HW; w++) { const uint32_t o = Sv[w]; Sv[w] = o + e; e &= (o == 0xffffffffu) ? 1u : 0u; }
            }
            // Phase 2a: diagonal zeros, horizontal differences
            uint32_t Zv[HW], h2v[HW], h1v[HW], h0v[HW];
#pragma unroll
            for (int w = 0; w < HW; w++) {
                Zv[w] = Bv[w] | (Pv[w] & (Sv[w] ^ Bv[w] ^ Yv[w]));
                const uint32_t b1 = Zv[w] & X0[w], b2 = X1[w] & b1;
                h0v[w] = ~(Zv[w] ^ X0[w]); h1v[w] = ~(X1[w] ^ b1); h2v[w] = ~(X2[w] ^ b2);
            }
            const uint32_t q2 = pair_partner(h2v[HW - 1]) >> 31, q1 = pair_partner(h1v[HW - 1]) >> 31, q0 = pair_partner(h0v[HW - 1]) >> 31;
            const uint32_t hin2 = half ? q2 : 1u, hin1 = half ? q1 : 1u, hin0 = half ? q0 : 0u;
            if (on) {
                // Phase 2b: the row below, new vertical differences
#pragma unroll
                for (int w = 0; w < HW; w++) {
                    const uint32_t s2 = (h2v[w] << 1) | (w ? h2v[w - 1] >> 31 : hin2), s1 = (h1v[w] << 1) | (w ? h1v[w - 1] >> 31 : hin1),
                                   s0 = (h0v[w] << 1) | (w ? h0v[w - 1] >> 31 : hin0);
                    const uint32_t c1 = Zv[w] & s0, c2 = s1 & c1;
                    X0[w] = ~(Zv[w] ^ s0); X1[w] = ~(s1 ^ c1); X2[w] = ~(s2 ^ c2);
                }
                if (!half) {
                    stop += AL_GAP;
                    const uint32_t c5 = tap0 | ((vposv[HW - 2] >> 31) << 1) | ((h2v[HW - 2] >> 31) << 2) | ((h1v[HW - 2] >> 31) << 3) | ((h0v[HW - 2] >> 31) << 4);
                    const int sh = 26 + 5 * (cc & 3);
                    if (sh + 5 <= 32) brec[(cc >> 2) * 2] |= c5 << sh;
                    else if (sh >= 32) brec[(cc >> 2) * 2 + 1] |= c5 << (sh - 32);
                    else { brec[(cc >> 2) * 2] |= c5 << sh; brec[(cc >> 2) * 2 + 1] |= c5 >> (32 - sh); }
                    brec[(cc >> 2) * 2 + 1] |= ((bm.m0 & 1u) | (bm.m1 & 2u) | (bm.inv & 4u)) << (14 + 3 * (cc & 3));
                }
            }
        }
        if (sa && !half) {
            uint4 *bp = reinterpret_cast<uint4 *>(rec0 + (size_t)k * AL_RECB + AL_RECB / 2);
            bp[0] = make_uint4(brec[0], brec[1], brec[2], brec[3]);
            bp[1] = make_uint4(brec[4], brec[5], brec[6], brec[7]);
        }
    }
    {
        // row m is bit H - 1 + (m - H - t_n), 0 .. 3 bits into word NW / 2 = the upper lane's word 0
        const int extra = m - H - t;
        const uint32_t part = pair_partner((uint32_t)plane_sum(X2[0], X1[0], X0[0], (1u << (extra & 31)) - 1u));
        if (n > 0 && !half) {
            int U = -1, kstar = -1;
            if (status == 0) {
                int u = stop - AL_GAP * H;
#pragma unroll
                for (int w = 0; w < HW; w++) u += plane_sum(X2[w], X1[w], X0[w], 0xffffffffu);
                u += (int)part - AL_GAP * extra;
                U = u;
                const int d = m - n, dmin = d < 0 ? d : 0, dmax = d > 0 ? d : 0, ad = d < 0 ? -d : d;
                int E = dmin - LO;
                if (HI - dmax < E) E = HI - dmax;
                if (E > (1 << 27)) kstar = 0x7fffffff;
                else if (E >= 0) kstar = AL_GAP * ad + 2 * AL_GAP * E + 2 * AL_GAP - 1;
            }
            P.U[g] = U; P.kst[g] = kstar; P.st[g] = status; P.lvl[g] = NW;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LANE-PARALLEL forward pass for the LONGEST pairs of a run (round 6): G lanes per pair, lane q of a group = word q of the
// band (G = 4: the 4-word level, G = 8: the 8-word level); a wavefront carries 64 / G pairs.  A kernel of thread-per-pair
// wavefronts cannot end before the dependent instruction chain of its longest pair (an 11 000-column window at ~165
// instructions per column = 3.7 ms, whatever the batch); in a large batch that chain hides behind the other pairs, in a
// small one -- a rank's share of a strong-scaling run, Util.py:8141-8147 gives every candidate its own process -- it IS the
// step.  One word per lane cuts the chain to the recurrence of ONE word + what crosses the lanes: bit 31 of vpos / h2 / h1 / h0
// of the word below (DPP row_shr:1; the lowest lane of a group takes the band-edge constants instead), the carry of the
// multi-word addition (carry-lookahead on two ballots, as in the 64-lane fall-back, with the chain cut at every group's
// top lane), the words that move down when the band moves (DPP row_shl:1) and the steering count of the two middle words
// (a butterfly over the group).  Same recurrence, same steering, same check points and boundary records, bit for bit, as
// align_fwd_kernel<G> -- the traceback kernels read either's records.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_row(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
template <int G>
__device__ __forceinline__ int grp_sum(int v) {      // sum over the G lanes of a group (G = 4 or 8, groups aligned), in every lane
    v += (int)dpp_row<0xB1>((uint32_t)v);            // quad_perm [1,0,3,2]
    v += (int)dpp_row<0x4E>((uint32_t)v);            // quad_perm [2,3,0,1]
    if (G == 8) v += (int)dpp_row<0x141>((uint32_t)v);   // row_half_mirror: the other quad of the 8
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { int o = __shfl_xor(v, d, 64); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ uint32_t bit_mask(uint32_t w, int bit) { return (uint32_t)((int32_t)(w << (31 - bit)) >> 31); }    // all ones / zero (v_bfe_i32)
template <int GI>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) { return dpp_row<GI * 0x55>(v); }                // lane GI of this lane's quad
struct LaneFwd {
    uint32_t X2, X1, X0, A0, A1, AN, f0, f1, fn;
    int t, stop, LO, HI, status;
    bool act;
};
// One strip of 16 columns.  FULL: every pair of the wavefront that is still running has all 16 columns, so nothing is tested per
// column and the state is committed unconditionally (a lane without a pair computes on garbage and writes nothing).
// invm: bit cc = "row base cc never matches"; code12: this lane's quad position's column group as the 4 x 3 code bits of the
// boundary record; K1: 1 in the lowest lane of a group -- bit 0 of what comes from "the word below" is the band edge there
// (+3: a set bit for h2 / h1, a clear one for h0 / vpos), folded into the consumers' bit operations.
template <int G, bool FULL>
__device__ __forceinline__ void lanes_strip(LaneFwd &S, const uint4 bw, const uint32_t invm, const uint32_t code12, const bool sa, const int k,
                                            const int m, const int n, const int q, const uint32_t K1, const unsigned long long TOPM, uint32_t (&brec)[8]) {
    constexpr int NW = G, W = 32 * NW, H = W / 2, S0 = NW / 2 - 1;
    const bool top = q == G - 1;
#pragma unroll
    for (int cc = 0; cc < AL_STRIP; cc++) {
        const int j = k * AL_STRIP + cc + 1;
        const bool on = FULL || (sa && j <= n);          // (the same in every lane of a group; the cross-lane moves run for every lane)
        if ((cc & 3) == 0) {
            // ---- band move: steering count of words S0 and S0 + 1, the word above for the shift
            const int ds = grp_sum<G>((q == S0 || q == S0 + 1) ? slope_count(S.X2, S.X1, S.X0) : 0);
            const uint32_t i2 = dpp_row<0x102>(S.X2) & 0xffu, i1 = dpp_row<0x102>(S.X1) & 0xffu, i0 = dpp_row<0x102>(S.X0) & 0xffu;   // word S0 + 2's low rows
            const uint32_t u2 = dpp_row<0x101>(S.X2), u1 = dpp_row<0x101>(S.X1), u0 = dpp_row<0x101>(S.X0);
            const uint32_t ua0 = dpp_row<0x101>(S.A0), ua1 = dpp_row<0x101>(S.A1), uan = dpp_row<0x101>(S.AN);
            constexpr int gi = 0;
            const uint32_t cg = (cc >> 2) == 0 ? quad_bcast<0>(code12) : ((cc >> 2) == 1 ? quad_bcast<1>(code12) : ((cc >> 2) == 2 ? quad_bcast<2>(code12) : quad_bcast<3>(code12)));
            (void)gi;
            if (on) {
                int s = ds > AL_STEER ? 0 : (ds < -AL_STEER ? 8 : 4);
                const int tr = m - H;
                if (S.t + s > tr) s = (tr - S.t) & ~3;
                const int need = tr - 3 - 8 * ((n - j) >> 2) - S.t;
                if (s < need) s = (need + 3) & ~3;
                if (s > 8) { S.status = 2; S.act = false; s = 8; }
                S.stop += plane_sum(S.X2, S.X1, S.X0, (1u << s) - 1u) - AL_GAP * s;        // (the lowest lane's is the pair's)
                brec[(cc >> 2) * 2] = (uint32_t)(s >> 2) | (i2 << 2) | (i1 << 10) | (i0 << 18);
                brec[(cc >> 2) * 2 + 1] = cg << 14;
                S.X2 = alignbit(top ? 0xffffffffu : u2, S.X2, (uint32_t)s);
                S.X1 = alignbit(top ? 0xffffffffu : u1, S.X1, (uint32_t)s);
                S.X0 = alignbit(top ? 0u : u0, S.X0, (uint32_t)s);
                S.A0 = alignbit(top ? S.f0 : ua0, S.A0, (uint32_t)s);
                S.A1 = alignbit(top ? S.f1 : ua1, S.A1, (uint32_t)s);
                S.AN = alignbit(top ? S.fn : uan, S.AN, (uint32_t)s);
                S.f0 >>= s; S.f1 >>= s; S.fn >>= s;
                S.t += s;
                if (S.t >= 1) { const int v = S.t + 1 - j; S.LO = v > S.LO ? v : S.LO; }
                if (S.t + W < m) { const int jl = j + 3 < n ? j + 3 : n; const int v = S.t + W - jl; S.HI = v < S.HI ? v : S.HI; }
            }
        }
        // ---- column
        const uint32_t word = cc < 4 ? bw.x : (cc < 8 ? bw.y : (cc < 12 ? bw.z : bw.w));
        const uint32_t m0 = bit_mask(word, 8 * (cc & 3) + 1), m1 = bit_mask(word, 8 * (cc & 3) + 2), inv = bit_mask(invm, cc);
        const uint32_t eq = ~((S.A0 ^ m0) | (S.A1 ^ m1) | S.AN | inv);
        const uint32_t vpos = S.X2 & S.X1;
        const uint32_t B = eq | ~(S.X2 | S.X1 | S.X0);
        const uint32_t pvr = dpp_row<0x111>(vpos);                     // bit 31: "vertical difference +3" of the row below this word
        const uint32_t Pp = alignbit(vpos, pvr, 31) & ~K1;
        const uint32_t Y = Pp | B;
        unsigned long long Gm;
        const uint32_t sum0 = add_co_mask(B, Y, Gm);
        uint32_t sum;
        {   // carries along the lanes of a group: lane w generates (its sum wrapped) or propagates (its sum is all ones); a
            // group's top lane does neither, so no carry crosses into the next group
            const unsigned long long Pm = __ballot(sum0 == 0xffffffffu);
            const unsigned long long Gc = Gm & ~TOPM, Yy = (Pm | Gm) & ~TOPM, ss = Gc + Yy;
            sum = add_mask_bit(sum0, ss ^ Gc ^ Yy);
        }
        const uint32_t Z = B | (Pp & (sum ^ B ^ Y));
        const uint32_t h0 = ~(Z ^ S.X0), b1 = Z & S.X0, h1 = ~(S.X1 ^ b1), b2 = S.X1 & b1, h2 = ~(S.X2 ^ b2);
        const uint32_t r2 = dpp_row<0x111>(h2), r1 = dpp_row<0x111>(h1), r0 = dpp_row<0x111>(h0);
        if (on) {
            const uint32_t s2 = alignbit(h2, r2, 31) | K1, s1 = alignbit(h1, r1, 31) | K1, s0 = alignbit(h0, r0, 31) & ~K1;     // below the band: +3
            const uint32_t c1 = Z & s0, c2 = s1 & c1;
            S.X0 = ~(Z ^ s0); S.X1 = ~(s1 ^ c1); S.X2 = ~(s2 ^ c2);
            // 5 carry bits that enter word S0 (kept by the lane that owns it: never a group's lowest): carry of the addition, then
            // bit 31 of the word below's vpos, h2, h1, h0 -- shifted in one after the other
            uint32_t c5 = alignbit(r0 >> 31, r1, 31);
            c5 = alignbit(c5, r2, 31);
            c5 = alignbit(c5, pvr, 31);
            c5 = (c5 << 1) | (sum - sum0);
            const int sh = 26 + 5 * (cc & 3);
            if (sh + 5 <= 32) brec[(cc >> 2) * 2] |= c5 << sh;
            else if (sh >= 32) brec[(cc >> 2) * 2 + 1] |= c5 << (sh - 32);
            else { brec[(cc >> 2) * 2] |= c5 << sh; brec[(cc >> 2) * 2 + 1] |= c5 >> (32 - sh); }
        }
    }
}
// a dword of four row bases -> bits 0..3: "never matches" per base; bits 4..15: the 3 code bits per base as the boundary record holds them
__device__ __forceinline__ uint32_t decode_bases4(uint32_t w) {
    uint32_t r = 0;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const unsigned ch = (w >> (8 * x)) & 0xffu;
        const uint32_t bad = is_acgt_byte(ch) ? 0u : 1u;
        r |= (bad << x) | ((((ch >> 1) & 3u) | (bad << 2)) << (4 + 3 * x));
    }
    return r;
}
template <int G>
__global__ void __launch_bounds__(64) align_fwd_lanes_kernel(AlignArgs P, const int32_t *__restrict__ list, int nlist) {
    constexpr int NW = G, W = 32 * NW, H = W / 2, S0 = NW / 2 - 1, PPW = 64 / G;
    const int q = threadIdx.x & (G - 1);
    const int li = blockIdx.x * PPW + (int)(threadIdx.x / G);
    const int g = li < nlist ? list[li] : -1;
    int m = 0, n = 0, g0 = 0, c = 0;
    if (g >= 0) {
        c = P.row_cand[g];
        g0 = P.row_first[c];
        if (g != g0) { m = P.win_len[g0]; n = P.win_len[g]; }
    }
    const int nmax = wave_max_i32(n);
    if (nmax == 0) return;
    const int nfull = wave_min_i32(n > 0 ? n : 0x7fffffff);        // strips below this are whole for every pair of the wavefront
    const bool wr = q == S0;                                        // the lane that writes the records
    const uint32_t K1 = q == 0 ? 1u : 0u;
    const unsigned long long TOPM = G == 4 ? 0x8888888888888888ull : 0x8080808080808080ull;
    const uint8_t *b = P.win + (g >= 0 ? P.win_off[g] : 0);
    const uint4 *pl = P.planes + (g >= 0 ? P.plane_off[c] : 0);
    char *rec0 = g >= 0 ? reinterpret_cast<char *>(P.rec[g]) : nullptr;
    LaneFwd S;
    S.X2 = q >= NW / 2 ? 0xffffffffu : 0u; S.X1 = S.X2; S.X0 = 0u; S.A0 = 0u; S.A1 = 0u; S.AN = 0xffffffffu;
    S.f0 = S.f1 = S.fn = 0u;
    S.t = -H;
    S.act = n > 0;
    if (S.act) { const uint4 v = pl[((S.t + AL_PADR) >> 5) + q]; S.A0 = v.x; S.A1 = v.y; S.AN = v.z; }
    S.stop = AL_GAP * H; S.LO = -(1 << 28); S.HI = 1 << 28; S.status = 0;
    int fq = (S.t + W + AL_PADR) >> 5;
    uint4 Wa = make_uint4(0, 0, 0xffffffffu, 0), Wb = Wa, Wc = Wa, bnext = make_uint4(0, 0, 0, 0);
    if (S.act) { Wa = pl[fq]; Wb = pl[fq + 1]; Wc = pl[fq + 2]; bnext = *reinterpret_cast<const uint4 *>(b); }
    for (int k = 0; k * AL_STRIP < nmax; k++) {
        const bool sa = S.act && k * AL_STRIP < n;
        uint4 bw = make_uint4(0, 0, 0, 0);
        uint32_t brec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        {   // check point: words S0 and S0 + 1 = this lane's and the lane above's
            const uint32_t p2 = dpp_row<0x101>(S.X2), p1 = dpp_row<0x101>(S.X1), p0 = dpp_row<0x101>(S.X0);
            if (sa && wr) {
                uint4 *ck = reinterpret_cast<uint4 *>(rec0 + (size_t)k * AL_RECB);
                ck[0] = make_uint4(S.X2, p2, S.X1, p1);
                ck[1] = make_uint4(S.X0, p0, (uint32_t)(S.t + 32 * S0), 0u);
            }
        }
        S.f0 = S.f1 = S.fn = 0u;
        if (sa) {
            bw = bnext;
            if ((k + 1) * AL_STRIP < n) bnext = *reinterpret_cast<const uint4 *>(b + (k + 1) * AL_STRIP);
            const int fx = S.t + W + AL_PADR;
            if ((fx >> 5) != fq) { Wa = Wb; Wb = Wc; fq++; Wc = pl[fq + 2]; }
            const uint32_t sh = (uint32_t)fx & 31u;
            S.f0 = alignbit(Wb.x, Wa.x, sh); S.f1 = alignbit(Wb.y, Wa.y, sh); S.fn = alignbit(Wb.z, Wa.z, sh);
        }
        // the strip's 16 row bases, decoded once: lane q of a quad takes dword q (a group of 8 decodes them twice), the "never
        // matches" bits of the four dwords are OR-ed over the quad, the code bits stay where the band move of their column
        // group fetches them
        const uint32_t dq = (q & 3) == 0 ? bw.x : ((q & 3) == 1 ? bw.y : ((q & 3) == 2 ? bw.z : bw.w));
        const uint32_t dec = decode_bases4(dq);
        uint32_t invm = (dec & 15u) << (4 * (q & 3));
        invm |= dpp_row<0xB1>(invm);
        invm |= dpp_row<0x4E>(invm);
        const bool failed = __any(n > 0 && S.status != 0);
        if ((k + 1) * AL_STRIP <= nfull && !failed) lanes_strip<G, true>(S, bw, invm, dec >> 4, sa, k, m, n, q, K1, TOPM, brec);
        else lanes_strip<G, false>(S, bw, invm, dec >> 4, sa, k, m, n, q, K1, TOPM, brec);
        if (sa && wr) {
            uint4 *bp = reinterpret_cast<uint4 *>(rec0 + (size_t)k * AL_RECB + AL_RECB / 2);
            bp[0] = make_uint4(brec[0], brec[1], brec[2], brec[3]);
            bp[1] = make_uint4(brec[4], brec[5], brec[6], brec[7]);
        }
    }
    {
        // row m is bit H - 1 + (m - H - t_n), 0 .. 3 bits into word NW / 2
        const int extra = m - H - S.t;
        const int part = q < NW / 2 ? plane_sum(S.X2, S.X1, S.X0, 0xffffffffu) : (q == NW / 2 ? plane_sum(S.X2, S.X1, S.X0, (1u << (extra & 31)) - 1u) : 0);
        const int tot = grp_sum<G>(part + (q == 0 ? S.stop : 0));
        if (n > 0 && wr) {
            int U = -1, kstar = -1;
            if (S.status == 0) {
                U = tot + AL_GAP * n - AL_GAP * H - AL_GAP * extra;          // (+ 3 per column: every column ran)
                const int d = m - n, dmin = d < 0 ? d : 0, dmax = d > 0 ? d : 0, ad = d < 0 ? -d : d;
                int E = dmin - S.LO;
                if (S.HI - dmax < E) E = S.HI - dmax;
                if (E > (1 << 27)) kstar = 0x7fffffff;
                else if (E >= 0) kstar = AL_GAP * ad + 2 * AL_GAP * E + 2 * AL_GAP - 1;
            }
            P.U[g] = U; P.kst[g] = kstar; P.st[g] = S.status; P.lvl[g] = NW;
            if (NW == 4) P.U4[g] = U;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// traceback pass: re-compute the slice strip by strip (registers), walk it backwards
// ---------------------------------------------------------------------------------------------
// ops leave the traceback in descending positions, one at a time and per lane, every lane into its own row: 64 different
// cache lines per wavefront.  Round 3 stored 8 bytes per four positions straight to the row: with ~130 000 rows open across the
// machine the partly written lines leave L2 one store at a time -- WRITE_SIZE 12.6 GB per launch for 3.3 GB of ops
// (profiles/r03_pmc_hbm.txt).  Round 4 measured five forms on C3 (profiles/r04_align_tb_store_forms.txt): whole 128-byte lines
// staged per lane in LDS reach 3.3 GB but cost a wavefront per SIMD (192 registers; 17.4 ms against 13.8); 32-byte pieces (an
// LDS ring drained per strip, or four 64-bit registers) write 6.3 GB -- the unit HBM charges is 64 bytes -- and cost the
// wavefront as well; the kernel is bound by its registers (three wavefronts per SIMD up to 168), not by its writes.  Kept:
// 16-byte chunks aligned in memory, 8.5 GB per launch at round 3's time.
typedef unsigned int tb_u32x4 __attribute__((ext_vector_type(4)));
#if defined(ALIGN_TB_STORE8)
// round 3's form: 8 bytes per four positions, groups aligned to the POSITION (kept for measurements)
struct OpsOut {
    uint16_t *ops;
    unsigned long long acc, top;
    int m;
    __device__ __forceinline__ void init(uint16_t *row, int m_) { ops = row; m = m_; acc = 0ull; top = 0ull; }
    __device__ __forceinline__ void push(int pos, uint32_t val) {
        acc = (acc << 16) | (unsigned long long)(val & 0xffffu);
        if ((pos & 3) == 0) {
            if (pos + 4 <= m) *reinterpret_cast<unsigned long long *>(ops + pos) = acc;
            else top = acc;
        }
    }
    __device__ __forceinline__ void finish() {
        const int k = m & 3, p0 = m & ~3;
        for (int x = 0; x < k; x++) ops[p0 + x] = (uint16_t)(top >> (16 * x));
    }
};
#elif defined(ALIGN_TB_STORE32)
struct OpsOut {
    uint16_t *ops;           // the lane's row
    unsigned long long acc;  // the group of four being collected (newest = lowest position in the low 16 bits)
    unsigned long long g0, g1, g2, g3;   // the complete groups of the sector being collected (g0 = lowest positions)
    int m, off;              // off = (address of ops[0] / 2) mod 16: q = position + off; q % 16 == 0 starts a sector
    int q_low;               // q of the lowest position whose SECTOR has been written (m + off: none yet)
    __device__ __forceinline__ void init(uint16_t *row, int m_) {
        ops = row; m = m_; acc = 0ull; g0 = g1 = g2 = g3 = 0ull;
        off = (int)((reinterpret_cast<uintptr_t>(row) >> 1) & 15);
        q_low = m_ + off;
    }
    // entry q of the sector / group being collected (rare paths only: the ends of a row)
    __device__ __forceinline__ uint16_t held(int q, int q_acc_low) const {
        if ((q >> 2) == (q_acc_low >> 2) && (q_acc_low & 3)) return (uint16_t)(acc >> (16 * (q - q_acc_low)));   // still in acc
        const int g = (q >> 2) & 3;
        const unsigned long long s = g == 0 ? g0 : (g == 1 ? g1 : (g == 2 ? g2 : g3));
        return (uint16_t)(s >> (16 * (q & 3)));
    }
    __device__ __forceinline__ void push(int pos, uint32_t val) {
        acc = (acc << 16) | (unsigned long long)(val & 0xffffu);
        const int q = pos + off;
        if ((q & 3) == 0) {
            const int g = (q >> 2) & 3;
            g0 = g == 0 ? acc : g0; g1 = g == 1 ? acc : g1; g2 = g == 2 ? acc : g2; g3 = g == 3 ? acc : g3;
            if (g == 0) {                    // the sector [pos, pos + 16) is complete as far as the row goes
                if (pos + 16 <= m) {
                    tb_u32x4 *dst = reinterpret_cast<tb_u32x4 *>(ops + pos);      // 32-byte aligned
                    tb_u32x4 v0, v1;
                    v0.x = (unsigned)g0; v0.y = (unsigned)(g0 >> 32); v0.z = (unsigned)g1; v0.w = (unsigned)(g1 >> 32);
                    v1.x = (unsigned)g2; v1.y = (unsigned)(g2 >> 32); v1.z = (unsigned)g3; v1.w = (unsigned)(g3 >> 32);
                    dst[0] = v0; dst[1] = v1;
                } else {
                    for (int x = pos; x < m; x++) ops[x] = held(x + off, q);      // the topmost sector of the row (partial)
                }
                q_low = q;
            }
        }
    }
    __device__ __forceinline__ void finish() {   // after position 0 has been pushed: the partial sector at the start of the row
        const int top = q_low - off < m ? q_low - off : m;       // positions [0, top) are still held
        for (int x = 0; x < top; x++) ops[x] = held(x + off, off);
    }
};

#else
// 16 bytes per eight positions, chunks aligned in MEMORY (q = position + the row's phase, q % 8 == 0 starts a chunk), never
// straddling a 32-byte sector: a third less write traffic than the 8-byte form at its instruction count and register budget.
// The walk itself only ever stores whole chunks; the partial chunks at the two ends of a row are kept in registers and written
// at the end.
struct OpsOut {
    uint16_t *ops;
    unsigned long long acc, hi;   // the group of four being collected; the complete upper group of the chunk being collected
    unsigned long long tacc, thi; // the topmost chunk of the row when it is partial
    int m, off;                   // off = (address of ops[0] / 2) mod 8
    __device__ __forceinline__ void init(uint16_t *row, int m_) {
        ops = row; m = m_; acc = 0ull; hi = 0ull; tacc = 0ull; thi = 0ull;
        off = (int)((reinterpret_cast<uintptr_t>(row) >> 1) & 7);
    }
    __device__ __forceinline__ void push(int pos, uint32_t val) {
        acc = (acc << 16) | (unsigned long long)(val & 0xffffu);
        const int q = pos + off;
        if ((q & 3) == 0) {
            if (q & 4) hi = acc;
            else if (pos + 8 <= m) {
                tb_u32x4 v;
                v.x = (unsigned)acc; v.y = (unsigned)(acc >> 32); v.z = (unsigned)hi; v.w = (unsigned)(hi >> 32);
                *reinterpret_cast<tb_u32x4 *>(ops + pos) = v;                 // 16-byte aligned
            } else { tacc = acc; thi = hi; }
        }
    }
    __device__ __forceinline__ void finish() {   // after position 0 has been pushed
        // the topmost chunk [t0, m), t0 = the chunk edge at or below m (nothing when m sits on an edge)
        const int t0 = m - ((m + off) & 7);
        if (t0 >= 0) for (int x = t0; x < m; x++) { const int qq = x + off; ops[x] = (uint16_t)(((qq & 4) ? thi : tacc) >> (16 * (qq & 3))); }
        // what lies below the lowest chunk edge b0 (when the row does not even reach it, the chunk [.., m) is the one above: t0 < 0)
        const int b0 = (8 - off) & 7;
        const int lowm = b0 < m ? b0 : m;
        for (int x = 0; x < lowm; x++) {
            const int qq = x + off;
            const bool in_acc = (off & 3) != 0 && (qq >> 2) == (off >> 2);        // the group position 0 sits in was never completed
            ops[x] = in_acc ? (uint16_t)(acc >> (16 * x)) : (uint16_t)(hi >> (16 * (qq & 3)));
        }
    }
};
#endif

#ifdef ALIGN_TB_OCC      // experiment: ask for this many waves per SIMD (the kernel sits just above the register count of the next step)
#define ALIGN_TB_ATTR __attribute__((amdgpu_waves_per_eu(ALIGN_TB_OCC, ALIGN_TB_OCC)))
#else
#define ALIGN_TB_ATTR
#endif
__global__ void __launch_bounds__(64) ALIGN_TB_ATTR align_tb_kernel(AlignArgs P, const int32_t *__restrict__ list, int nlist) {
    ACLK_BEGIN(1);
    const int li = blockIdx.x * 64 + threadIdx.x;
    const int g = li < nlist ? list[li] : -1;
    int m = 0, n = 0, g0 = 0, c = 0;
    if (g >= 0) {
        c = P.row_cand[g];
        g0 = P.row_first[c];
        if (g != g0 && P.st[g] == 0) { m = P.win_len[g0]; n = P.win_len[g]; }   // (every band is wider than the slice: records exist)
    }
    const int nmax = wave_max_i32(n);
    if (nmax == 0) return;
    const uint4 *pl = P.planes + (g >= 0 ? P.plane_off[c] : 0);
    const char *rec0 = g >= 0 ? reinterpret_cast<const char *>(P.rec[g]) : nullptr;
    uint16_t *ops = P.ops + (g >= 0 ? P.ops_base[c] + (int64_t)(g - g0) * (m + 1) : 0);
    int i = m, j = n;
    bool fail = false;
    OpsOut out;
    out.init(ops, m);
    const int Kmax = (nmax + AL_STRIP - 1) / AL_STRIP;
    // software pipeline over the strips (last to first): the check point of strip k-2, the boundary record and the row bases of
    // strip k-1 and -- with the position the check point of strip k-1 gives -- its centre planes are fetched while strip k is
    // computed; without it every strip began with two dependent memory round trips
    const int K = (n + AL_STRIP - 1) / AL_STRIP;      // this lane's strips (0 = nothing to do)
    struct Pw { uint32_t x, y, z; };
    uint4 ckA0, ckA1, ckB0, ckB1, bdA0, bdA1;
    Pw plA[4];
    ckA0 = ckA1 = ckB0 = ckB1 = bdA0 = bdA1 = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int w = 0; w < 4; w++) { plA[w].x = 0; plA[w].y = 0; plA[w].z = 0; }
    {
        const int k = Kmax - 1;
        if (k < K) {
            const uint4 *ck = reinterpret_cast<const uint4 *>(rec0 + (size_t)k * AL_RECB);
            ckA0 = ck[0]; ckA1 = ck[1];
            const uint4 *bp = reinterpret_cast<const uint4 *>(rec0 + (size_t)k * AL_RECB + AL_RECB / 2);
            bdA0 = bp[0]; bdA1 = bp[1];
            const int q0 = ((int)ckA1.z + AL_PADR) >> 5;
#pragma unroll
            for (int w = 0; w < 4; w++) { const uint4 v = pl[q0 + w]; plA[w].x = v.x; plA[w].y = v.y; plA[w].z = v.z; }
        }
        if (k >= 1 && k - 1 < K) {
            const uint4 *ck = reinterpret_cast<const uint4 *>(rec0 + (size_t)(k - 1) * AL_RECB);
            ckB0 = ck[0]; ckB1 = ck[1];
        }
    }
    for (int k = Kmax - 1; k >= 0; k--) {
        const bool sa = n > 0 && !fail && i > 0 && k < K;
        // ---- fetch for the strips to come (A = strip k, B = check point of strip k-1)
        uint4 ckC0 = make_uint4(0, 0, 0, 0), ckC1 = ckC0, bdB0 = ckC0, bdB1 = ckC0;
        Pw plB[4];
#pragma unroll
        for (int w = 0; w < 4; w++) { plB[w].x = 0; plB[w].y = 0; plB[w].z = 0; }
        if (k >= 1 && k - 1 < K && !fail && i > 0) {      // (its check point arrived a strip ago, or in the prologue)
            const uint4 *bp = reinterpret_cast<const uint4 *>(rec0 + (size_t)(k - 1) * AL_RECB + AL_RECB / 2);
            bdB0 = bp[0]; bdB1 = bp[1];
            const int q0 = ((int)ckB1.z + AL_PADR) >> 5;
#pragma unroll
            for (int w = 0; w < 4; w++) { const uint4 v = pl[q0 + w]; plB[w].x = v.x; plB[w].y = v.y; plB[w].z = v.z; }
        }
        if (k >= 2 && k - 2 < K && !fail && i > 0) {
            const uint4 *ck = reinterpret_cast<const uint4 *>(rec0 + (size_t)(k - 2) * AL_RECB);
            ckC0 = ck[0]; ckC1 = ck[1];
        }
        if (__any(sa)) {
        uint32_t X2[2], X1[2], X0[2], A0[2], A1[2], AN[2];
        uint32_t f0 = 0, f1 = 0, fn = 0;
        uint32_t brec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int t = 0;
        if (sa) {
            X2[0] = ckA0.x; X2[1] = ckA0.y; X1[0] = ckA0.z; X1[1] = ckA0.w; X0[0] = ckA1.x; X0[1] = ckA1.y;
            t = (int)ckA1.z;
            brec[0] = bdA0.x; brec[1] = bdA0.y; brec[2] = bdA0.z; brec[3] = bdA0.w;
            brec[4] = bdA1.x; brec[5] = bdA1.y; brec[6] = bdA1.z; brec[7] = bdA1.w;
            const uint32_t sh = (uint32_t)(t + AL_PADR) & 31u;
#pragma unroll
            for (int w = 0; w < 2; w++) {
                A0[w] = alignbit(plA[w + 1].x, plA[w].x, sh); A1[w] = alignbit(plA[w + 1].y, plA[w].y, sh);
                AN[w] = alignbit(plA[w + 1].z, plA[w].z, sh);
            }
            f0 = alignbit(plA[3].x, plA[2].x, sh); f1 = alignbit(plA[3].y, plA[2].y, sh); fn = alignbit(plA[3].z, plA[2].z, sh);
        } else {
#pragma unroll
            for (int w = 0; w < 2; w++) { X2[w] = 0; X1[w] = 0; X0[w] = 0; A0[w] = 0; A1[w] = 0; AN[w] = 0; }
        }
        unsigned long long dg[AL_STRIP], up[AL_STRIP];
        uint32_t sbits = 0;     // 2 bits per group of four columns: rows the band moved / 4
#pragma unroll
        for (int cc = 0; cc < AL_STRIP; cc++) {
            const int jc = k * AL_STRIP + cc + 1;
            dg[cc] = 0ull; up[cc] = 0ull;
            if (sa && jc <= n) {
                const uint32_t rlo = brec[(cc >> 2) * 2], rhi = brec[(cc >> 2) * 2 + 1];
                if ((cc & 3) == 0) {
                    const uint32_t s = (rlo & 3u) << 2;
                    const uint32_t in2 = (rlo >> 2) & 0xffu, in1 = (rlo >> 10) & 0xffu, in0 = (rlo >> 18) & 0xffu;
                    X2[0] = alignbit(X2[1], X2[0], s); X1[0] = alignbit(X1[1], X1[0], s); X0[0] = alignbit(X0[1], X0[0], s);
                    A0[0] = alignbit(A0[1], A0[0], s); A1[0] = alignbit(A1[1], A1[0], s); AN[0] = alignbit(AN[1], AN[0], s);
                    X2[1] = alignbit(in2, X2[1], s); X1[1] = alignbit(in1, X1[1], s); X0[1] = alignbit(in0, X0[1], s);
                    A0[1] = alignbit(f0, A0[1], s); A1[1] = alignbit(f1, A1[1], s); AN[1] = alignbit(fn, AN[1], s);
                    f0 >>= s; f1 >>= s; fn >>= s;
                    t += (int)s;
                    sbits |= (rlo & 3u) << (2 * (cc >> 2));
                }
                const int sh = 26 + 5 * (cc & 3);
                const uint32_t c5 = (sh + 5 <= 32 ? rlo >> sh : (sh >= 32 ? rhi >> (sh - 32) : (rlo >> sh) | (rhi << (32 - sh)))) & 31u;
                const uint32_t code = rhi >> (14 + 3 * (cc & 3));     // the row base, as the forward pass left it in the record
                BaseMask bm;
                bm.m0 = 0u - (code & 1u); bm.m1 = 0u - ((code >> 1) & 1u); bm.inv = 0u - ((code >> 2) & 1u);
                uint32_t tap[5], dgc[2], upc[2];
                bp_core<2, true, -1>(X2, X1, X0, A0, A1, AN, bm, c5 & 1u, (c5 >> 1) & 1u, (c5 >> 2) & 1u, (c5 >> 3) & 1u, (c5 >> 4) & 1u, dgc, upc, tap);
                dg[cc] = ((unsigned long long)dgc[1] << 32) | dgc[0];
                up[cc] = ((unsigned long long)upc[1] << 32) | upc[0];
            }
        }
        // walk: every path step leaves column j for column j-1 (diagonal, left) or stays in it (up).  A run of up steps ends at
        // the highest row at or above the current one where the diagonal is allowed or up is not: one count of leading zeros
        // on the column's 64 slice bits, so that the register arrays are only ever indexed by constants.
        int tcur = t;
#pragma unroll
        for (int cc = AL_STRIP - 1; cc >= 0; cc--) {
            const int jc = k * AL_STRIP + cc + 1;
            if (sa && !fail && i > 0 && j == jc) {
                const int kb = i - tcur - 1;
                if ((unsigned)kb >= 64u) fail = true;
                else {
                    const unsigned long long stopm = (dg[cc] | ~up[cc]) & (0xffffffffffffffffull >> (63 - kb));   // bits 0 .. kb
                    if (stopm == 0ull) fail = true;      // the run of up steps leaves the slice at its first row
                    else {
                        const int ps = 63 - __clzll(stopm);
                        int ups = kb - ps;
                        if (ups > i) ups = i;            // (cannot happen: row 0 stops every run)
                        const uint32_t gapv = (uint32_t)j | 0x8000u;
                        for (int x = 0; x < ups; x++) out.push(i - 1 - x, gapv);
                        i -= ups;
                        if (i > 0) {
                            if ((dg[cc] >> ps) & 1ull) { out.push(i - 1, (uint32_t)(j - 1)); i--; j--; }
                            else j--;
                        }
                    }
                }
            }
            if ((cc & 3) == 0) tcur -= (int)(((sbits >> (2 * (cc >> 2))) & 3u) << 2);
        }
        }
        ckA0 = ckB0; ckA1 = ckB1; ckB0 = ckC0; ckB1 = ckC1; bdA0 = bdB0; bdA1 = bdB1;
#pragma unroll
        for (int w = 0; w < 4; w++) plA[w] = plB[w];
    }
    if (n > 0) {
        if (fail) P.st[g] = 1;
        else {
            for (int q = i - 1; q >= 0; q--) out.push(q, 0x8000u);
            out.finish();
        }
    }
    ACLK_END();
}

// ---------------------------------------------------------------------------------------------
// wide fall-back: one WAVEFRONT per pair, lane w = word w of a band of 64 words (2048 centre rows); the traceback bits of
// the whole band are kept (512 B per column).  Same recurrence as bp_core; what runs along the words there runs along the
// lanes here: neighbours by DPP wave_shl / wave_shr, the carry chain of the addition by a carry-lookahead on two ballots.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_above(uint32_t v, uint32_t fill) { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xf, 0xf, false); }   // lane i <- lane i+1
__device__ __forceinline__ uint32_t lane_below(uint32_t v, uint32_t fill) { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false); }   // lane i <- lane i-1
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// A lone wavefront issues one instruction every ~4.6 cycles whatever its kind (profiles/r03_valu_issue.txt, 1 wave per SIMD), and
// the 51 pairs of a C3 step that come here are its longest: the kernel's time is instructions per column x columns of the
// longest pair.  So the columns are unrolled in strips of 16 (every index a constant, no loop arithmetic), the 16 row bases
// of a strip are decoded by 16 lanes at once into three wave masks (one scalar bit-field extract per column and mask
// instead of the scalar compare chain), the carry of the addition comes straight from the add's carry-out mask and goes back
// in through an add-with-carry, and the band position is stored once per move (four columns).
// lane i <- lane i-1 of v; lane 0 keeps what `keep` held before (its own fill value, written once before the loop: the
// builtin form re-materialises the fill in front of every move).  The s_nop covers the VALU-write -> DPP-read hazard of v,
// which the compiler does not track through an asm statement.
__device__ __forceinline__ uint32_t lane_below_keep(uint32_t v, uint32_t &keep) {
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(keep) : "v"(v));
    return keep;
}
__device__ __forceinline__ void lane_below_keep3(uint32_t v2, uint32_t v1, uint32_t v0, uint32_t &k2, uint32_t &k1, uint32_t &k0) {
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %2, %5 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(k2), "+v"(k1), "+v"(k0) : "v"(v2), "v"(v1), "v"(v0));
}
struct WideFwd {
    uint32_t X2, X1, X0, A0, A1, AN, f0, f1, fn;
    uint32_t kz0, kz1, ko0, ko1;       // registers of lane_below_keep: lane 0 holds 0 / 0 / bit 31 / bit 31
    int t, stop, LO, HI, status;
};
// one strip of <= 16 columns j0 .. j0 + 15 (FULL: all 16 exist, no per-column test)
template <bool FULL>
__device__ __forceinline__ void wide_fwd_strip(WideFwd &S, const AlignArgs &P, int lane, int m, int n, int j0, int64_t full0, unsigned long long M0,
                                               unsigned long long M1, unsigned long long MI) {
    constexpr int NW = AL_WIDE_NW, W = 32 * NW, H = W / 2, S0 = NW / 2 - 1;
    uint32_t *fb = P.fullbuf + (full0 + j0) * (2 * NW) + lane;
#pragma unroll
    for (int cc = 0; cc < AL_STRIP; cc++) {
        const int j = j0 + cc;
        if (FULL || (j <= n && S.status == 0)) {
            if ((cc & 3) == 0) {   // the band moves in every fourth column only, by 0 / 4 / 8 rows (same rule as align_fwd_kernel)
                const int dv = slope_count(S.X2, S.X1, S.X0);
                const int ds = __builtin_amdgcn_readlane(dv, S0) + __builtin_amdgcn_readlane(dv, S0 + 1);
                int s = ds > AL_STEER ? 0 : (ds < -AL_STEER ? 8 : 4);
                const int tr = m - H;
                if (S.t + s > tr) s = (tr - S.t) & ~3;
                const int need = tr - 3 - 8 * ((n - j) >> 2) - S.t;
                if (s < need) s = (need + 3) & ~3;
                if (s > 8) { S.status = 2; s = 8; }      // (the columns that follow compute on, nothing of them is used)
                const uint32_t smask = (1u << s) - 1u;
                S.stop += plane_sum((uint32_t)__builtin_amdgcn_readlane((int)S.X2, 0), (uint32_t)__builtin_amdgcn_readlane((int)S.X1, 0),
                                    (uint32_t)__builtin_amdgcn_readlane((int)S.X0, 0), smask) - AL_GAP * s;
                S.X2 = alignbit(lane_above(S.X2, 0xffffffffu), S.X2, (uint32_t)s);
                S.X1 = alignbit(lane_above(S.X1, 0xffffffffu), S.X1, (uint32_t)s);
                S.X0 = alignbit(lane_above(S.X0, 0u), S.X0, (uint32_t)s);
                S.A0 = alignbit(lane_above(S.A0, S.f0), S.A0, (uint32_t)s);
                S.A1 = alignbit(lane_above(S.A1, S.f1), S.A1, (uint32_t)s);
                S.AN = alignbit(lane_above(S.AN, S.fn), S.AN, (uint32_t)s);
                S.f0 >>= s; S.f1 >>= s; S.fn >>= s;
                S.t += s;
                if (S.t >= 1) { const int v = S.t + 1 - j; S.LO = v > S.LO ? v : S.LO; }
                if (S.t + W < m) { const int jl = j + 3 < n ? j + 3 : n; const int v = S.t + W - jl; S.HI = v < S.HI ? v : S.HI; }
                if (lane < 4 && j + lane <= n) P.fullt[full0 + j + lane] = S.t;      // t of the four columns of this move
            }
            S.stop += AL_GAP;
            BaseMask bm;
            bm.m0 = 0u - (uint32_t)((M0 >> cc) & 1ull); bm.m1 = 0u - (uint32_t)((M1 >> cc) & 1ull); bm.inv = 0u - (uint32_t)((MI >> cc) & 1ull);
            const uint32_t eq = ~((S.A0 ^ bm.m0) | (S.A1 ^ bm.m1) | S.AN | bm.inv);
            const uint32_t vneg = ~(S.X2 | S.X1 | S.X0), vpos = S.X2 & S.X1;
            const uint32_t B = eq | vneg, Pp = alignbit(vpos, lane_below_keep(vpos, S.kz0), 31);
            const uint32_t Y = Pp | B;
            unsigned long long Gm;
            uint32_t sum = add_co_mask(B, Y, Gm);
            {   // carries along the lanes: lane w generates (its sum wrapped) or propagates (its sum is all ones)
                const unsigned long long Pm = __ballot(sum == 0xffffffffu);
                const unsigned long long Yy = Pm | Gm, ss = Gm + Yy;
                const unsigned long long into = ss ^ Gm ^ Yy;                // carry into bit u of Gm + (Pm | Gm) = carry into lane u
                sum = add_mask_bit(sum, into);
            }
            const uint32_t Z = B | (Pp & (sum ^ B ^ Y));
            const uint32_t h0 = ~(Z ^ S.X0), b1 = Z & S.X0, h1 = ~(S.X1 ^ b1), b2 = S.X1 & b1, h2 = ~(S.X2 ^ b2);
            lane_below_keep3(h2, h1, h0, S.ko0, S.ko1, S.kz1);
            const uint32_t s2 = alignbit(h2, S.ko0, 31), s1 = alignbit(h1, S.ko1, 31), s0 = alignbit(h0, S.kz1, 31);
            const uint32_t n0 = ~(Z ^ s0), c1 = Z & s0, n1 = ~(s1 ^ c1), c2 = s1 & c1, n2 = ~(s2 ^ c2);
            S.X2 = n2; S.X1 = n1; S.X0 = n0;
            fb[cc * (2 * NW)] = eq | ~Z;
            fb[cc * (2 * NW) + NW] = n2 & n1;
        }
    }
}
__global__ void __launch_bounds__(64) align_wide_fwd_kernel(AlignArgs P, const int32_t *__restrict__ list, int nlist) {
    constexpr int NW = AL_WIDE_NW, W = 32 * NW, H = W / 2;
    const int pi = blockIdx.x;
    if (pi >= nlist) return;
    const int lane = threadIdx.x;
    const int g = list[pi];
    const int c = P.row_cand[g], g0 = P.row_first[c];
    if (g == g0) return;
    const int m = P.win_len[g0], n = P.win_len[g];
    const uint8_t *b = P.win + P.win_off[g];
    const uint4 *pl = P.planes + P.plane_off[c];
    const int64_t full0 = P.full_off[g];
    WideFwd S;
    S.t = -H;
    S.X2 = lane >= NW / 2 ? 0xffffffffu : 0u; S.X1 = S.X2; S.X0 = 0u;
    { const uint4 v = pl[((S.t + AL_PADR) >> 5) + lane]; S.A0 = v.x; S.A1 = v.y; S.AN = v.z; }
    S.stop = AL_GAP * H; S.LO = -(1 << 28); S.HI = 1 << 28; S.status = 0;
    S.kz0 = 0u; S.kz1 = 0u; S.ko0 = 0x80000000u; S.ko1 = 0x80000000u;
    unsigned chn = b[lane & 15];                     // the row bases of the next strip, one per lane (windows sit in padded slots)
    for (int j0 = 1; j0 <= n && S.status == 0; j0 += AL_STRIP) {
        // the strip's bases as three masks: bit cc = code bit 0 / code bit 1 / "never matches" of column j0 + cc
        const unsigned ch = chn;
        chn = b[j0 - 1 + AL_STRIP + (lane & 15)];
        const unsigned long long M0 = __ballot((ch >> 1) & 1u), M1 = __ballot((ch >> 2) & 1u), MI = __ballot(!is_acgt_byte(ch));
        {   // the next 32 rows below the band, per plane (at most 2 rows enter per column)
            const int fx = S.t + W + AL_PADR;
            const uint4 F = pl[fx >> 5], G = pl[(fx >> 5) + 1];
            const uint32_t sh = (uint32_t)fx & 31u;
            S.f0 = alignbit(G.x, F.x, sh); S.f1 = alignbit(G.y, F.y, sh); S.fn = alignbit(G.z, F.z, sh);
        }
        if (j0 + AL_STRIP - 1 <= n) wide_fwd_strip<true>(S, P, lane, m, n, j0, full0, M0, M1, MI);
        else wide_fwd_strip<false>(S, P, lane, m, n, j0, full0, M0, M1, MI);
    }
    int U = -1, kstar = -1;
    if (S.status == 0) {
        const int extra = m - H - S.t;   // row m is bit H - 1 + extra, extra = 0 .. 3
        U = S.stop - AL_GAP * H - AL_GAP * extra +
            wave_sum_i32(lane < NW / 2 ? plane_sum(S.X2, S.X1, S.X0, 0xffffffffu) : (lane == NW / 2 ? plane_sum(S.X2, S.X1, S.X0, (1u << extra) - 1u) : 0));
        const int d = m - n, dmin = d < 0 ? d : 0, dmax = d > 0 ? d : 0, ad = d < 0 ? -d : d;
        int E = dmin - S.LO;
        if (S.HI - dmax < E) E = S.HI - dmax;
        if (E > (1 << 27)) kstar = 0x7fffffff;
        else if (E >= 0) kstar = AL_GAP * ad + 2 * AL_GAP * E + 2 * AL_GAP - 1;
    }
    if (lane == 0) { P.U[g] = U; P.kst[g] = kstar; P.st[g] = S.status; P.lvl[g] = NW | 0x100; }
}

// traceback of the fall-back: one wavefront per pair, lane w holds word w of the column's bits; the run of up steps ends at
// the highest stop bit (diagonal allowed, or up not allowed) at or below the current row -- two ballots and a count of
// leading zeros; ops leave in coalesced runs.  The columns are fetched eight at a time (the walk visits them in order).
// ops leave through the lanes: lane l holds the op of position base + l of the current block of 64 positions; a block goes out
// as one coalesced store when the walk leaves it downwards
struct WideOps {
    uint16_t *ops;
    int m, lane;
    uint32_t acc;
    __device__ __forceinline__ void flush(int base) { if (base + lane < m) ops[base + lane] = (uint16_t)acc; }
    __device__ __forceinline__ void put(int pos, uint32_t val) {          // one op at position pos (wave-uniform)
        if (lane == (pos & 63)) acc = val;
        if ((pos & 63) == 0) flush(pos);
    }
    __device__ __forceinline__ void run_diag(int hi, int count, int delta) {   // positions hi, hi - 1, ..., hi - count + 1; position p gets p + delta
        while (count > 0) {
            const int base = hi & ~63;
            const int lo = hi - count + 1 > base ? hi - count + 1 : base;
            if (base + lane >= lo && base + lane <= hi) acc = (uint32_t)(base + lane + delta);
            const int k = hi - lo + 1;
            count -= k; hi -= k;
            if (lo == base) flush(base);
        }
    }
    __device__ __forceinline__ void run(int hi, int count, uint32_t val) {   // positions hi, hi - 1, ..., hi - count + 1
        while (count > 0) {
            const int base = hi & ~63;
            const int lo = hi - count + 1 > base ? hi - count + 1 : base;
            if (base + lane >= lo && base + lane <= hi) acc = val;
            const int k = hi - lo + 1;
            count -= k; hi -= k;
            if (lo == base) flush(base);
        }
    }
};
__global__ void __launch_bounds__(64) align_wide_tb_kernel(AlignArgs P, const int32_t *__restrict__ list, int nlist) {
    constexpr int NW = AL_WIDE_NW, W = 32 * NW, PF = 8;
    const int pi = blockIdx.x;
    if (pi >= nlist) return;
    const int lane = threadIdx.x;
    const int g = list[pi];
    const int c = P.row_cand[g], g0 = P.row_first[c];
    if (g == g0 || P.st[g] != 0) return;
    const int m = P.win_len[g0], n = P.win_len[g];
    WideOps out;
    out.ops = P.ops + P.ops_base[c] + (int64_t)(g - g0) * (m + 1); out.m = m; out.lane = lane; out.acc = 0u;
    const int64_t full0 = P.full_off[g];
    int i = m, j = n;
    bool fail = false;
    // the columns are visited one per step in descending order, so the next eight are fetched while these eight are walked
    // (a lone wavefront per pair has nothing else to hide the load behind: the fetch used to be half of the kernel's time).
    // The common step is the diagonal one -- the diagonal bit of the current cell is set: canonical first choice -- and needs
    // that one bit only (a readlane and scalar code); the "up" words are fetched when a column needs the general rule.
    uint32_t dgn[PF];
    int ttn[PF];
#pragma unroll
    for (int q = 0; q < PF; q++) {
        const int jq = j - q > 0 ? j - q : 1;
        dgn[q] = P.fullbuf[(full0 + jq) * (2 * NW) + lane]; ttn[q] = P.fullt[full0 + jq];
    }
    while (i > 0 && j > 0 && !fail) {
        uint32_t dgs[PF];
        int tts[PF];
#pragma unroll
        for (int q = 0; q < PF; q++) { dgs[q] = dgn[q]; tts[q] = ttn[q]; }
#pragma unroll
        for (int q = 0; q < PF; q++) {
            const int jq = j - PF - q > 0 ? j - PF - q : 1;
            dgn[q] = P.fullbuf[(full0 + jq) * (2 * NW) + lane]; ttn[q] = P.fullt[full0 + jq];
        }
#pragma unroll
        for (int q = 0; q < PF; q++) {
            if (i > 0 && j > 0 && !fail) {
                const uint32_t dgw = dgs[q];
                const int kb = i - tts[q] - 1;
                if ((unsigned)kb >= (unsigned)W) fail = true;
                else {
                    const int wq = kb >> 5;
                    const uint32_t dcur = (uint32_t)__builtin_amdgcn_readlane((int)dgw, wq);
                    if ((dcur >> (kb & 31)) & 1u) { out.put(i - 1, (uint32_t)(j - 1)); i--; j--; }
                    else {
                        const uint32_t upw = P.fullbuf[(full0 + j) * (2 * NW) + NW + lane];
                        const uint32_t low = 0xffffffffu >> (31 - (kb & 31));
                        const uint32_t sbit = dgw | ~upw;
                        const uint32_t mk = lane > wq ? 0u : (lane == wq ? sbit & low : sbit);
                        const unsigned long long nz = __ballot(mk != 0u);
                        if (nz == 0ull) fail = true;
                        else {
                            const int tl = 63 - __clzll(nz);
                            const uint32_t mt = (uint32_t)__builtin_amdgcn_readlane((int)mk, tl), dt = (uint32_t)__builtin_amdgcn_readlane((int)dgw, tl);
                            const int ps = 32 * tl + 31 - __clz(mt);
                            int nup = kb - ps;
                            if (nup > i) nup = i;
                            out.run(i - 1, nup, (uint32_t)j | 0x8000u);
                            i -= nup;
                            if (i > 0) {
                                if ((dt >> (ps & 31)) & 1u) { out.put(i - 1, (uint32_t)(j - 1)); i--; j--; }
                                else j--;
                            }
                        }
                    }
                }
            }
        }
    }
    if (fail) { if (lane == 0) P.st[g] = 1; }
    else out.run(i - 1, i, 0x8000u);
}

// ---------------------------------------------------------------------------------------------
// traceback of the LONGEST pairs (round 6): one WAVEFRONT per pair.  The re-computation of a strip needs nothing but the
// strip's own check point and boundary record, so the 64 lanes re-compute 64 STRIPS (1024 columns) at once -- the same
// bp_core<2> steps as align_tb_kernel, one strip per lane -- and leave the slice bits (diagonal allowed / up allowed, 64 rows
// per column) in LDS; then the wavefront walks those 1024 columns backwards as ONE path: the 16 columns of the strip being
// walked sit in 16 lanes, the walk reads them with v_readlane and runs on the scalar unit (the common step, a diagonal one,
// tests one bit); ops leave 64 positions at a time, coalesced (WideOps).  The dependent chain per column is the walk's ~15
// instructions instead of the ~190 of re-computation + walk in one lane: 5.4 -> ~0.5 ms for an 11 000-column window.
// Same bits, same rule, same ops as align_tb_kernel.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}
__global__ void __launch_bounds__(64) align_tb_strips_kernel(AlignArgs P, const int32_t *__restrict__ list, int nlist) {
    __shared__ unsigned long long s_dg[64][AL_STRIP + 1], s_up[64][AL_STRIP + 1];      // (+1: the lanes of a round write one column at a time)
    __shared__ int s_t[64][4];                                                          // position of the slice in the strip's four column groups
    const int pi = blockIdx.x;
    if (pi >= nlist) return;
    const int lane = threadIdx.x;
    const int g = list[pi];
    const int c = P.row_cand[g], g0 = P.row_first[c];
    if (g == g0 || P.st[g] != 0) return;
    const int m = P.win_len[g0], n = P.win_len[g];
    if (n <= 0) return;
    const uint4 *pl = P.planes + P.plane_off[c];
    const char *rec0 = reinterpret_cast<const char *>(P.rec[g]);
    WideOps out;
    out.ops = P.ops + P.ops_base[c] + (int64_t)(g - g0) * (m + 1); out.m = m; out.lane = lane; out.acc = 0u;
    const int K = (n + AL_STRIP - 1) / AL_STRIP;
    int i = m, j = n;
    bool fail = false;
    for (int kb0 = K - 1; kb0 >= 0 && !fail && i > 0; kb0 -= 64) {
        {   // ---- lane l: strip kb0 - l
            const int k = kb0 - lane;
            if (k >= 0) {
                const uint4 *ck = reinterpret_cast<const uint4 *>(rec0 + (size_t)k * AL_RECB);
                const uint4 *bp = reinterpret_cast<const uint4 *>(rec0 + (size_t)k * AL_RECB + AL_RECB / 2);
                const uint4 ck0 = ck[0], ck1 = ck[1], bd0 = bp[0], bd1 = bp[1];
                uint32_t X2[2] = {ck0.x, ck0.y}, X1[2] = {ck0.z, ck0.w}, X0[2] = {ck1.x, ck1.y}, A0[2], A1[2], AN[2];
                int t = (int)ck1.z;
                const uint32_t brec[8] = {bd0.x, bd0.y, bd0.z, bd0.w, bd1.x, bd1.y, bd1.z, bd1.w};
                const int q0 = (t + AL_PADR) >> 5;
                const uint4 p0 = pl[q0], p1 = pl[q0 + 1], p2 = pl[q0 + 2], p3 = pl[q0 + 3];
                const uint32_t sh = (uint32_t)(t + AL_PADR) & 31u;
                A0[0] = alignbit(p1.x, p0.x, sh); A1[0] = alignbit(p1.y, p0.y, sh); AN[0] = alignbit(p1.z, p0.z, sh);
                A0[1] = alignbit(p2.x, p1.x, sh); A1[1] = alignbit(p2.y, p1.y, sh); AN[1] = alignbit(p2.z, p1.z, sh);
                uint32_t f0 = alignbit(p3.x, p2.x, sh), f1 = alignbit(p3.y, p2.y, sh), fn = alignbit(p3.z, p2.z, sh);
#pragma unroll
                for (int cc = 0; cc < AL_STRIP; cc++) {
                    const int jc = k * AL_STRIP + cc + 1;
                    unsigned long long dgv = 0ull, upv = 0ull;
                    if (jc <= n) {
                        const uint32_t rlo = brec[(cc >> 2) * 2], rhi = brec[(cc >> 2) * 2 + 1];
                        if ((cc & 3) == 0) {
                            const uint32_t s = (rlo & 3u) << 2;
                            const uint32_t in2 = (rlo >> 2) & 0xffu, in1 = (rlo >> 10) & 0xffu, in0 = (rlo >> 18) & 0xffu;
                            X2[0] = alignbit(X2[1], X2[0], s); X1[0] = alignbit(X1[1], X1[0], s); X0[0] = alignbit(X0[1], X0[0], s);
                            A0[0] = alignbit(A0[1], A0[0], s); A1[0] = alignbit(A1[1], A1[0], s); AN[0] = alignbit(AN[1], AN[0], s);
                            X2[1] = alignbit(in2, X2[1], s); X1[1] = alignbit(in1, X1[1], s); X0[1] = alignbit(in0, X0[1], s);
                            A0[1] = alignbit(f0, A0[1], s); A1[1] = alignbit(f1, A1[1], s); AN[1] = alignbit(fn, AN[1], s);
                            f0 >>= s; f1 >>= s; fn >>= s;
                            t += (int)s;
                        }
                        const int sh5 = 26 + 5 * (cc & 3);
                        const uint32_t c5 = (sh5 + 5 <= 32 ? rlo >> sh5 : (sh5 >= 32 ? rhi >> (sh5 - 32) : (rlo >> sh5) | (rhi << (32 - sh5)))) & 31u;
                        const uint32_t code = rhi >> (14 + 3 * (cc & 3));
                        BaseMask bm;
                        bm.m0 = 0u - (code & 1u); bm.m1 = 0u - ((code >> 1) & 1u); bm.inv = 0u - ((code >> 2) & 1u);
                        uint32_t tap[5], dgc[2], upc[2];
                        bp_core<2, true, -1>(X2, X1, X0, A0, A1, AN, bm, c5 & 1u, (c5 >> 1) & 1u, (c5 >> 2) & 1u, (c5 >> 3) & 1u, (c5 >> 4) & 1u, dgc, upc, tap);
                        dgv = ((unsigned long long)dgc[1] << 32) | dgc[0];
                        upv = ((unsigned long long)upc[1] << 32) | upc[0];
                    }
                    s_dg[lane][cc] = dgv; s_up[lane][cc] = upv;
                    if ((cc & 3) == 0) s_t[lane][cc >> 2] = t;
                }
            }
        }
        __syncthreads();
        // ---- the walk through these strips, last to first; wave-uniform.  Lane cc (< 16) holds column cc of the strip being
        //      walked.  Most steps are diagonal ones, so the lanes first test, each for its own column, whether the path that
        //      runs down the current diagonal may pass (one bit of the column's slice: row i - (columns to go) of it), a ballot
        //      gives the length of the diagonal run, its ops leave together; only the column that ends a run takes the general rule.
        const int nst = kb0 + 1 < 64 ? kb0 + 1 : 64;
        unsigned long long dgn = s_dg[0][lane & 15], upn = s_up[0][lane & 15];
        int tn = s_t[0][(lane & 15) >> 2];
        for (int l = 0; l < nst && !fail && i > 0; l++) {
            const unsigned long long dgl = dgn, upl = upn;
            const int tl = tn;
            if (l + 1 < nst) { dgn = s_dg[l + 1][lane & 15]; upn = s_up[l + 1][lane & 15]; tn = s_t[l + 1][(lane & 15) >> 2]; }     // (arrives while this strip is walked)
            const int j0 = (kb0 - l) * AL_STRIP;             // columns j0 + 1 .. j0 + 16
            while (!fail && i > 0) {
                i = __builtin_amdgcn_readfirstlane(i); j = __builtin_amdgcn_readfirstlane(j);      // (wave-uniform by construction: keep the walk on the scalar unit)
                const int cur = j - j0 - 1;                  // the current column, as a lane
                if (cur < 0) break;
                {   // diagonal run
                    const int back = cur - (lane & 15);      // columns between the current one and this lane's
                    const int kbl = i - back - tl - 1;
                    const bool ok = back >= 0 && i - back >= 1 && (unsigned)kbl < 64u && ((dgl >> (kbl & 63)) & 1ull);
                    const uint32_t okm = (uint32_t)__ballot(ok) & 0xffffu;
                    const uint32_t bad = ~okm & ((2u << cur) - 1u);
                    const int run = bad ? cur - (31 - __clz(bad)) : cur + 1;
                    if (run > 0) { out.run_diag(i - 1, run, j - i); i -= run; j -= run; }
                    if (run == cur + 1 || i <= 0) continue;  // (the strip is done -- cur < 0 next time --, or the path)
                }
                {   // the column that ended the run: the general rule
                    const int cc = j - j0 - 1;
                    const int kb = i - __builtin_amdgcn_readlane(tl, cc) - 1;
                    if ((unsigned)kb >= 64u) { fail = true; break; }
                    const unsigned long long dgv = readlane_u64(dgl, cc), upv = readlane_u64(upl, cc);
                    const unsigned long long stopm = (dgv | ~upv) & (0xffffffffffffffffull >> (63 - kb));
                    if (stopm == 0ull) { fail = true; break; }
                    const int ps = 63 - __clzll(stopm);
                    int ups = kb - ps;
                    if (ups > i) ups = i;
                    out.run(i - 1, ups, (uint32_t)j | 0x8000u);
                    i -= ups;
                    if (i > 0) {
                        if ((dgv >> ps) & 1ull) { out.put(i - 1, (uint32_t)(j - 1)); i--; j--; }
                        else j--;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (fail) { if (lane == 0) P.st[g] = 1; }
    else out.run(i - 1, i, 0x8000u);
}

// ---------------------------------------------------------------------------------------------
// schedule
// ---------------------------------------------------------------------------------------------
// what: 0 = pairs to re-run with a band of `level` words (exact mode), 1 = pairs whose traceback left the slice,
// 2 / 3 = pairs that will / will not be re-run with a wider band at all (asked after the 4-word run)
__global__ void align_flag_kernel(int64_t nrows, const int32_t *__restrict__ order, const int32_t *__restrict__ strips, AlignArgs P,
                                  int what, int level, int cap, int32_t *__restrict__ flag, int32_t *__restrict__ cols) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nrows) return;
    const int g = order[x];
    int f = 0;
    if (strips[g] > 0) {
        if (what == 0) {
            const int u4 = P.U4[g];
            if (P.st[g] == 0 && u4 >= 0 && u4 + AL_GAP * AL_MARGIN <= 32 * AL_GAP * cap) {
                int first = 8;
                while (32 * AL_GAP * first < u4 + AL_GAP * AL_MARGIN) first *= 2;
                const bool cert = P.U[g] <= P.kst[g];
                f = level >= first && !cert;
            }
        } else if (what == 1) f = P.st[g] == 1;
        else {      // 2: pairs that will be re-run with a wider band (decided by the 4-word run alone); 3: the others
            const int u4 = P.U4[g];
            const bool esc = P.st[g] == 0 && u4 >= 0 && u4 + AL_GAP * AL_MARGIN <= 32 * AL_GAP * cap && !(P.U[g] <= P.kst[g]);
            f = what == 2 ? esc : !esc;
        }
    }
    flag[x] = f;
    if (cols) cols[x] = f ? P.win_len[g] + 1 : 0;
}
__global__ void align_compact_kernel(int64_t nrows, const int32_t *__restrict__ order, const int32_t *__restrict__ flag,
                                     const int64_t *__restrict__ pos, int32_t *__restrict__ out, const int64_t *__restrict__ colpos,
                                     int64_t *__restrict__ full_off, int32_t *__restrict__ st) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nrows) return;
    if (flag[x]) {
        const int g = order[x];
        out[pos[x]] = g;
        if (full_off) { full_off[g] = colpos[x]; st[g] = 0; }
    }
}
// strips of the longest pair of every block of 64 list entries (= the strips the block's wavefront runs), and -- after the
// exclusive scan of those -- the record address of every pair of the list
__global__ void __launch_bounds__(256) align_blockmax_kernel(int64_t cnt, const int32_t *__restrict__ list, const int32_t *__restrict__ strips,
                                                             int32_t *__restrict__ bk) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int k = wave_max_i32(x < cnt ? strips[list[x]] : 0);
    if ((threadIdx.x & 63) == 0 && x < cnt) bk[x >> 6] = k;
}
__global__ void align_assign_kernel(int64_t cnt, const int32_t *__restrict__ list, const int64_t *__restrict__ boff, char *region,
                                    unsigned long long *__restrict__ rec) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < cnt) rec[list[x]] = (unsigned long long)(uintptr_t)(region + boff[x >> 6] * (int64_t)AL_RECB + (x & 63) * 32);
}
// How many leading blocks of 64 list entries (the lists run longest pair first, so bk -- the strips of a block's longest pair
// -- never increases, and boff is its running sum) go to a lane-parallel kernel.  Fixed (fixed_strips >= 0: the blocks whose
// longest pair has at least that many strips), or by a small cost model of the two kernels running side by side on the
// machine's 1024 SIMDs.  With k blocks in the lane-parallel kernel the launch cannot end before
//     the lane-parallel kernel's chain over the longest pair            bk[0]  * lane_chain
//     the thread-per-pair kernel's chain over ITS longest pair          bk[k]  * thread_chain
//     the work of both, spread over the whole machine                   boff[k] * lane_work + (boff[nb] - boff[k]) * thread_work
// (ns per strip of 16 columns; measured on MI355X with tools/align_chain_bench.py, profiles/r06_align_chain_bench.txt: a lone
// wavefront needs 0.34 / 0.44 / 0.43 us per column in the 4-word forward / 8-word forward / traceback kernels whatever its
// neighbours do, 0.20 / 0.20 / 0.033 us in the lane-parallel ones; a full machine 4.2 ps per pair-column in the thread-per-pair
// kernels, and 12.5 / 25 / 21 ns per pair-column and SIMD -- 3 to 5 times the work -- in the lane-parallel ones.)  The first
// term falls with k more slowly than the third one grows, so the k where they cross is the best; k = 0 unless that gains 10 %.
// C3 (1.6 M block-strips per launch): work-bound, k = 0.  A C4 share of an eighth, full-length pass (0.3 M block-strips,
// longest pair 690 strips): the chain of the longest pair is 2.2 x the work -> the pairs beyond ~0.75 of the longest.
struct LaneCost { float thread_chain, lane_chain, thread_work, lane_work; };
__device__ __forceinline__ double lane_cost_at(int64_t k, int64_t nb, const int32_t *bk, const int64_t *boff, const LaneCost c) {
    const double chain_l = k > 0 ? (double)bk[0] * c.lane_chain : 0.0, chain_t = k < nb ? (double)bk[k] * c.thread_chain : 0.0;
    const double work = (double)boff[k] * c.lane_work + (double)(boff[nb] - boff[k]) * c.thread_work;
    double t = chain_l > chain_t ? chain_l : chain_t;
    return t > work ? t : work;
}
__global__ void align_long_blocks_kernel(int64_t nb, const int32_t *__restrict__ bk, const int64_t *__restrict__ boff, int64_t fixed_strips,
                                         LaneCost cf, LaneCost ct, int64_t *__restrict__ out) {
    if (blockIdx.x || threadIdx.x) return;
    for (int w = 0; w < 2; w++) {
        int64_t lo = 0, hi = nb;
        if (fixed_strips >= 0) {            // first block with bk < fixed_strips
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)bk[mid] >= fixed_strips) lo = mid + 1; else hi = mid; }
        } else {
            const LaneCost c = w ? ct : cf;
            // first k where the lane-parallel side (its chain, or the work of both) is no longer below the thread-per-pair chain
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                const double a = (double)bk[0] * c.lane_chain, wk = (double)boff[mid] * c.lane_work + (double)(boff[nb] - boff[mid]) * c.thread_work;
                const double tc = mid < nb ? (double)bk[mid] * c.thread_chain : 0.0;
                if ((a > wk ? a : wk) < tc) lo = mid + 1; else hi = mid;
            }
            if (lo > 0 && lane_cost_at(lo - 1, nb, bk, boff, c) <= lane_cost_at(lo, nb, bk, boff, c)) lo--;
            if (lo > 0 && (bk[0] < AL_LANES_MIN_STRIPS || lane_cost_at(lo, nb, bk, boff, c) > 0.9 * lane_cost_at(0, nb, bk, boff, c))) lo = 0;
        }
        out[w] = lo;
    }
}
// counters: [0] pairs, [1] certified, [2] kept from a band of > 4 words, [3] fall-back, [4] dropped, [5] sum of U, [6] columns
__global__ void __launch_bounds__(256) align_stats_kernel(int64_t total_rows, const int32_t *__restrict__ strips, AlignArgs P, int32_t *__restrict__ row_dead,
                                                          int32_t *__restrict__ cand_status, unsigned long long *__restrict__ acc) {
    // grid-stride over the rows, sums kept in registers, one reduction per workgroup at the end: 7 atomics per workgroup of a
    // 1024-workgroup launch (seven same-address atomics per row were 0.6 ms per launch, per wavefront still 0.55 ms)
    __shared__ unsigned long long s_acc[4][7];
    unsigned long long v[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total_rows; g += (int64_t)gridDim.x * blockDim.x) {
        if (strips[g] <= 0) {
            // a centre -- or an EMPTY non-centre row (the _dev entry points do not validate lengths): that one is dropped like a
            // row the aligner could not place, instead of reaching the layout with an ops row nobody wrote
            const bool empty_row = g != P.row_first[P.row_cand[g]];
            if (empty_row) { P.st[g] = 2; P.U[g] = -1; P.kst[g] = -1; P.lvl[g] = 0; v[0] += 1ull; v[4] += 1ull; }
            if (row_dead) row_dead[g] = empty_row;
            if (empty_row && cand_status) atomicExch(&cand_status[P.row_cand[g]], 2);
        }
        else {
            const int st = P.st[g];
            v[0] += 1ull;
            v[1] += st == 0 && P.U[g] <= P.kst[g];
            v[2] += st == 0 && (P.lvl[g] & 0xff) > 4;
            v[3] += (P.lvl[g] & 0x100) != 0;
            v[4] += st != 0;
            v[5] += st == 0 ? (unsigned long long)P.U[g] : 0ull;
            v[6] += (unsigned long long)P.win_len[g];
            if (row_dead) row_dead[g] = st != 0;
            if (st != 0 && cand_status) atomicExch(&cand_status[P.row_cand[g]], 2);
        }
    }
#pragma unroll
    for (int q = 0; q < 7; q++) {
        unsigned long long x = v[q];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
        if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6][q] = x;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const unsigned long long x = s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x] + s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x];
        if (x) atomicAdd(&acc[threadIdx.x], x);
    }
}
__global__ void align_info_kernel(int64_t total_rows, AlignArgs P, int32_t *__restrict__ info /* 5 per row */) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total_rows) return;
    const int c = P.row_cand[g];
    if (g == P.row_first[c]) { for (int q = 0; q < 5; q++) info[g * 5 + q] = 0; return; }
    info[g * 5 + 0] = P.U[g];
    info[g * 5 + 1] = P.st[g] != 2 && P.U[g] <= P.kst[g];
    info[g * 5 + 2] = P.st[g];
    info[g * 5 + 3] = P.kst[g];
    info[g * 5 + 4] = P.lvl[g];
}

struct AlignState {
    Arena arena;
    int64_t *h_pin = nullptr;
    int64_t *d_scal = nullptr;
    int exact_cap = -1;
    int lanes_min = -1;            // lane-parallel kernels for pairs of >= this many columns; -1: by the size of the run; -2: never
    float lanes_scale[4] = {1.f, 1.f, 1.f, 1.f};     // measurement aid ($HITE_ALIGN_LANES_SCALE=a,b,c,d): factors on the cost model's lane-parallel work (4-word, 8-word, traceback) and forward chain
    bool sort_attr = false;
    hipStream_t st2 = nullptr;     // the wider bands run here, beside the traceback of the pairs that are already final; the longest pairs' lane-parallel kernels
    hipEvent_t ev = nullptr, ev_fork = nullptr, ev_join = nullptr;
    int64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// the second stream gets the highest priority: its few long-running waves are the critical path and must not queue behind the
// traceback's many short workgroups
static hipError_t align_make_stream(hipStream_t *out) {
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; hi = 0; }
    return hipStreamCreateWithPriority(out, hipStreamNonBlocking, hi);
}

static AlignState *align_state(hite_ctx *ctx) {
    if (!ctx->align_state) {
        AlignState *S = new AlignState();
        if (hipHostMalloc((void **)&S->h_pin, 64 * sizeof(int64_t)) != hipSuccess || hipMalloc((void **)&S->d_scal, 64 * sizeof(int64_t)) != hipSuccess ||
            align_make_stream(&S->st2) != hipSuccess || hipEventCreateWithFlags(&S->ev, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&S->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&S->ev_join, hipEventDisableTiming) != hipSuccess) {
            delete S;
            return nullptr;
        }
        const char *e = getenv("HITE_ALIGN_EXACT");
        int cap = e && *e ? atoi(e) : AL_DEFAULT_CAP;
        if (cap != 0 && cap != 8 && cap != 16 && cap != 32) cap = AL_DEFAULT_CAP;
        S->exact_cap = cap;
        const char *l = getenv("HITE_ALIGN_LANES");
        if (l && *l) { const int v = atoi(l); S->lanes_min = v < -2 ? -1 : v; }
        const char *sc = getenv("HITE_ALIGN_LANES_SCALE");
        if (sc && *sc) sscanf(sc, "%f,%f,%f,%f", &S->lanes_scale[0], &S->lanes_scale[1], &S->lanes_scale[2], &S->lanes_scale[3]);
        ctx->align_state = S;
    }
    return (AlignState *)ctx->align_state;
}
void hite_align_release(hite_ctx *ctx) {
    AlignState *S = (AlignState *)ctx->align_state;
    if (!S) return;
    arena_free(S->arena);
    if (S->h_pin) (void)hipHostFree(S->h_pin);
    if (S->d_scal) (void)hipFree(S->d_scal);
    if (S->st2) (void)hipStreamDestroy(S->st2);
    if (S->ev) (void)hipEventDestroy(S->ev);
    if (S->ev_fork) (void)hipEventDestroy(S->ev_fork);
    if (S->ev_join) (void)hipEventDestroy(S->ev_join);
    delete S;
    ctx->align_state = nullptr;
}

extern "C" int hite_align_config(hite_ctx *ctx, int32_t exact_cap) {
    if (!ctx || (exact_cap != 0 && exact_cap != 8 && exact_cap != 16 && exact_cap != 32)) return HITE_EINVAL;
    AlignState *S = align_state(ctx);
    if (!S) return HITE_ENOMEM;
    S->exact_cap = exact_cap;
    return HITE_OK;
}
extern "C" int hite_align_lanes(hite_ctx *ctx, int32_t min_cols) {
    if (!ctx || min_cols < -2 || min_cols > 32767) return HITE_EINVAL;
    AlignState *S = align_state(ctx);
    if (!S) return HITE_ENOMEM;
    S->lanes_min = min_cols;
    return HITE_OK;
}
extern "C" int hite_align_stats(hite_ctx *ctx, int64_t *out8, int32_t reset) {
    if (!ctx || !out8) return HITE_EINVAL;
    AlignState *S = align_state(ctx);
    if (!S) return HITE_ENOMEM;
    memcpy(out8, S->stats, sizeof S->stats);
    out8[7] = S->exact_cap;
    if (reset) memset(S->stats, 0, sizeof S->stats);
    return HITE_OK;
}

#define ACHK(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

template <typename T>
static int aalloc(hite_ctx *ctx, Arena &A, size_t count, T **out) {
    void *p = nullptr;
    int rc = arena_alloc(ctx, A, count * sizeof(T), &p);
    *out = (T *)p;
    return rc;
}

// order-preserving compaction of the rows flagged for `what`; returns the count (host) and the list (device)
static int build_list(hite_ctx *ctx, AlignState *S, hipStream_t st, int64_t nrows, const int32_t *order, const int32_t *strips,
                      AlignArgs &P, int what, int level, int cap, int32_t *flag, int32_t *cols, int64_t *pos, int64_t *colpos,
                      int64_t *scan_tmp, int32_t *list, int64_t *full_off, int64_t *count, int64_t *total_cols, int slot = 0) {
    const unsigned nb = (unsigned)((nrows + 255) / 256);
    hipLaunchKernelGGL(align_flag_kernel, dim3(nb), dim3(256), 0, st, nrows, order, strips, P, what, level, cap, flag, cols);
    ACHK(scan_excl_buf<int32_t>(ctx, scan_tmp, flag, nrows, pos, st));
    if (cols) ACHK(scan_excl_buf<int32_t>(ctx, scan_tmp, cols, nrows, colpos, st));
    hipLaunchKernelGGL(align_compact_kernel, dim3(nb), dim3(256), 0, st, nrows, order, flag, pos, list, cols ? colpos : nullptr,
                       cols ? full_off : nullptr, P.st);
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal + slot, pos + nrows, 8, hipMemcpyDeviceToDevice, st));
    if (cols) HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal + slot + 1, colpos + nrows, 8, hipMemcpyDeviceToDevice, st));
    HITE_CHECK(ctx, hipMemcpyAsync(S->h_pin + slot, S->d_scal + slot, 16, hipMemcpyDeviceToHost, st));
    HITE_CHECK(ctx, hipStreamSynchronize(st));
    *count = S->h_pin[slot];
    if (total_cols) *total_cols = cols ? S->h_pin[slot + 1] : 0;
    return HITE_OK;
}

// strip records of one forward run over `list`: block sizes, their scan, the region (arena), the addresses
// *n_long (may be NULL): how many entries at the head of the list -- whole blocks of 64 -- go to the lane-parallel kernels
static int assign_records(hite_ctx *ctx, AlignState *S, hipStream_t st, const int32_t *list, int64_t cnt, const int32_t *strips, int32_t *bk,
                          int64_t *boff, int64_t *scan_tmp, unsigned long long *rec, int slot, int words = 4, int64_t *n_long = nullptr, int64_t *n_long_tb = nullptr) {
    if (n_long) *n_long = 0;
    if (n_long_tb) *n_long_tb = 0;
    if (cnt <= 0) return HITE_OK;
    const int64_t nb = (cnt + 63) / 64;
    hipLaunchKernelGGL(align_blockmax_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, cnt, list, strips, bk);
    ACHK(scan_excl_buf<int32_t>(ctx, scan_tmp, bk, nb, boff, st));
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal + slot, boff + nb, 8, hipMemcpyDeviceToDevice, st));
    const bool lanes = n_long && S->lanes_min != -2;
    if (lanes) {
        // ns per strip of 16 columns: chains of a lone wavefront, work per block of 64 pairs on the whole machine (see the kernel)
        LaneCost fwd4 = {5440.f, 3200.f, 4.3f, 12.5f}, fwd8 = {7040.f, 3200.f, 13.75f, 25.f}, tb = {6880.f, 530.f, 4.3f, 21.5f};
        fwd4.lane_work *= S->lanes_scale[0]; fwd8.lane_work *= S->lanes_scale[1]; tb.lane_work *= S->lanes_scale[2];
        fwd4.lane_chain *= S->lanes_scale[3]; fwd8.lane_chain *= S->lanes_scale[3];
        hipLaunchKernelGGL(align_long_blocks_kernel, dim3(1), dim3(1), 0, st, nb, bk, boff,
                           (int64_t)(S->lanes_min < 0 ? -1 : (S->lanes_min + AL_STRIP - 1) / AL_STRIP), words == 4 ? fwd4 : fwd8, tb, S->d_scal + slot + 1);
    }
    HITE_CHECK(ctx, hipMemcpyAsync(S->h_pin + slot, S->d_scal + slot, 24, hipMemcpyDeviceToHost, st));
    HITE_CHECK(ctx, hipStreamSynchronize(st));
    const int64_t block_strips = S->h_pin[slot];
    if (lanes) {
        const int64_t v = S->h_pin[slot + 1] * 64, vt = S->h_pin[slot + 2] * 64;
        *n_long = v < cnt ? v : cnt;
        if (n_long_tb) *n_long_tb = vt < cnt ? vt : cnt;
    }
    char *region;
    ACHK(aalloc(ctx, S->arena, (size_t)block_strips * AL_RECB + 256, &region));
    hipLaunchKernelGGL(align_assign_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, cnt, list, boff, region, rec);
    return HITE_OK;
}

// aligns every row of every candidate to the candidate's first row; ops as described in hite_align.h.
// d_row_dead (total_rows, may be NULL): 1 for rows that could not be aligned (0 for centres); d_cand_flag (n_cand, may be
// NULL): set to 2 for candidates that lost a row.  d_info (5 x total_rows int32, may be NULL): per row U, certified, status,
// k*, band words | 0x100 (fall-back).
int hite_align_run(hite_ctx *ctx, int32_t n_cand, const uint8_t *d_win, const int64_t *d_win_off, const int32_t *d_win_len,
                   const int32_t *d_row_first, int64_t total_rows, const int64_t *d_ops_base, uint16_t *d_ops,
                   int32_t *d_row_dead, int32_t *d_cand_flag, int32_t *d_info, const char *tag, hipStream_t st) {
    if (!ctx || n_cand < 0 || total_rows < 0 || total_rows > 0x7fffffff) return HITE_EINVAL;
    if (n_cand == 0 || total_rows == 0) return HITE_OK;
    AlignState *S = align_state(ctx);
    if (!S) return HITE_ENOMEM;
    Arena &A = S->arena;
    ACHK(arena_reset(ctx, A, true));
    const int cap = S->exact_cap;
    AlignArgs P;
    memset(&P, 0, sizeof P);
    P.win = d_win; P.win_off = d_win_off; P.win_len = d_win_len; P.row_first = d_row_first; P.n_cand = n_cand;
    P.ops_base = d_ops_base; P.ops = d_ops;
    int32_t *row_cand, *strips, *order, *pwords, *flag, *flag2, *cols, *list, *list_esc, *list_fin;
    unsigned long long *skeys;
    int64_t *plane_off, *pos, *pos2, *colpos, *scan_tmp, *scan_tmp2, *full_off, *boff, *boff2;
    int32_t *bk, *bk2;
    unsigned long long *rec;
    ACHK(aalloc(ctx, A, (size_t)total_rows, &row_cand));
    ACHK(aalloc(ctx, A, (size_t)total_rows, &strips));
    ACHK(aalloc(ctx, A, (size_t)total_rows + 1, &skeys));
    ACHK(aalloc(ctx, A, (size_t)total_rows + 1, &order));
    Sorter srt;
    srt.ctx = ctx; srt.st = st; srt.cap = total_rows; srt.hist_n = sorter_hist_elems(total_rows);
    ACHK(aalloc(ctx, A, (size_t)total_rows + 1, &srt.k2));
    ACHK(aalloc(ctx, A, (size_t)total_rows + 1, &srt.v2));
    ACHK(aalloc(ctx, A, (size_t)srt.hist_n, &srt.hist));
    ACHK(aalloc(ctx, A, (size_t)srt.hist_n + 1, &srt.offs));
    ACHK(aalloc(ctx, A, (size_t)sorter_tmp_elems(srt.hist_n), &srt.bs));
    ACHK(aalloc(ctx, A, (size_t)n_cand, &pwords));
    ACHK(aalloc(ctx, A, (size_t)n_cand + 1, &plane_off));
    ACHK(aalloc(ctx, A, (size_t)total_rows, &rec));
    ACHK(aalloc(ctx, A, (size_t)total_rows / 64 + 2, &bk));
    ACHK(aalloc(ctx, A, (size_t)total_rows / 64 + 3, &boff));
    ACHK(aalloc(ctx, A, (size_t)total_rows / 64 + 2, &bk2));
    ACHK(aalloc(ctx, A, (size_t)total_rows / 64 + 3, &boff2));
    ACHK(aalloc(ctx, A, (size_t)total_rows, &flag));
    ACHK(aalloc(ctx, A, (size_t)total_rows, &cols));
    ACHK(aalloc(ctx, A, (size_t)total_rows, &list));
    ACHK(aalloc(ctx, A, (size_t)total_rows, &list_esc));
    ACHK(aalloc(ctx, A, (size_t)total_rows, &list_fin));
    ACHK(aalloc(ctx, A, (size_t)total_rows, &flag2));
    ACHK(aalloc(ctx, A, (size_t)total_rows + 1, &pos2));
    ACHK(aalloc(ctx, A, (size_t)scan_tmp_elems(total_rows) + 16, &scan_tmp2));
    ACHK(aalloc(ctx, A, (size_t)total_rows + 1, &pos));
    ACHK(aalloc(ctx, A, (size_t)total_rows + 1, &colpos));
    ACHK(aalloc(ctx, A, (size_t)total_rows, &full_off));
    ACHK(aalloc(ctx, A, (size_t)scan_tmp_elems(total_rows > n_cand ? total_rows : n_cand) + 16, &scan_tmp));
    ACHK(aalloc(ctx, A, (size_t)total_rows, &P.U));
    ACHK(aalloc(ctx, A, (size_t)total_rows, &P.kst));
    ACHK(aalloc(ctx, A, (size_t)total_rows, &P.st));
    ACHK(aalloc(ctx, A, (size_t)total_rows, &P.lvl));
    ACHK(aalloc(ctx, A, (size_t)total_rows, &P.U4));
    P.row_cand = row_cand; P.full_off = full_off;
    int tk = hite_prof_begin(ctx, "align_prep", st);
    HITE_CHECK(ctx, hipMemsetAsync(P.st, 0, (size_t)total_rows * 4, st));
    HITE_CHECK(ctx, hipMemsetAsync(P.lvl, 0, (size_t)total_rows * 4, st));
    hipLaunchKernelGGL(align_rows_kernel, dim3(n_cand), dim3(256), 0, st, n_cand, d_row_first, d_win_len, row_cand, strips, skeys, (unsigned *)order);
    if (!S->sort_attr) {
        HITE_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(rs_scatter_staged_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, RSS_LDS_BYTES));
        S->sort_attr = true;
    }
    ACHK(sorter_sort(srt, skeys, (unsigned *)order, total_rows, 11));
    hipLaunchKernelGGL(align_plane_words_kernel, dim3((n_cand + 255) / 256), dim3(256), 0, st, n_cand, d_row_first, d_win_len, pwords);
    ACHK(scan_excl_buf<int32_t>(ctx, scan_tmp, pwords, n_cand, plane_off, st));
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal, plane_off + n_cand, 8, hipMemcpyDeviceToDevice, st));
    HITE_CHECK(ctx, hipMemcpyAsync(S->h_pin, S->d_scal, 8, hipMemcpyDeviceToHost, st));
    int64_t nl4 = 0, nltb = 0;   // the longest pairs (a prefix of `order`): lane-parallel forward pass / traceback
    ACHK(assign_records(ctx, S, st, order, total_rows, strips, bk, boff, scan_tmp, rec, 1, 4, &nl4, &nltb));     // (synchronises)
    const int64_t plane_words = S->h_pin[0];
    uint4 *planes;
    ACHK(aalloc(ctx, A, (size_t)plane_words + 8, &planes));
    P.planes = planes; P.plane_off = plane_off; P.rec = rec;
    hipLaunchKernelGGL(align_planes_kernel, dim3(n_cand), dim3(256), 0, st, n_cand, d_win, d_win_off, d_win_len, d_row_first, plane_off, planes);
    hite_prof_end(ctx, tk, st);
    HITE_CHECK(ctx, hipGetLastError());
    char name[32];
    const int nrows = (int)total_rows;
    // ---- the 4-word band for every pair
    snprintf(name, sizeof name, "align_fwd4%s", tag ? tag : "");
    hipStream_t s2 = S->st2;
    char lname[40];
    if (getenv("HITE_ALIGN_DEBUG")) fprintf(stderr, "[align%s] rows %d  block-strips %lld  lane-parallel prefix %lld forward, %lld traceback\n", tag ? tag : "", nrows, (long long)S->h_pin[1], (long long)nl4, (long long)nltb);
    if (nl4 > 0) {               // beside the other pairs, on the high-priority stream: its few wavefronts are the critical path
        HITE_CHECK(ctx, hipEventRecord(S->ev_fork, st));
        HITE_CHECK(ctx, hipStreamWaitEvent(s2, S->ev_fork, 0));
        snprintf(lname, sizeof lname, "align_fwd4_lanes%s", tag ? tag : "");
        const int tl = hite_prof_begin(ctx, lname, s2);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(align_fwd_lanes_kernel<4>), dim3((unsigned)((nl4 + 15) / 16)), dim3(64), 0, s2, P, order, (int)nl4);
        hite_prof_end(ctx, tl, s2);
        HITE_CHECK(ctx, hipEventRecord(S->ev_join, s2));
    }
    if (nrows > nl4) {
        tk = hite_prof_begin(ctx, name, st);
        hipLaunchKernelGGL(align_fwd4_kernel, dim3((unsigned)((nrows - nl4 + 63) / 64)), dim3(64), 0, st, P, order + nl4, (int)(nrows - nl4));
        hite_prof_end(ctx, tk, st);
    }
    if (nl4 > 0) HITE_CHECK(ctx, hipStreamWaitEvent(st, S->ev_join, 0));
    // ---- exact mode: wider bands for the pairs without a certificate, on a second stream, beside the traceback of the pairs
    //      whose 4-word run is final (the wider bands have few, long-running waves: alone they leave most SIMDs idle)
    snprintf(name, sizeof name, "align_tb%s", tag ? tag : "");
    char wname[32];
    snprintf(wname, sizeof wname, "align_fwd_wide%s", tag ? tag : "");
    int64_t n_esc = 0, n_fin = 0;
    // measured on C3: beside the 8-word level alone the traceback gains nothing (both are issue-bound: 148.8 vs 140.6 ms per step),
    // with the 16-word level (1.5 waves per SIMD, long tail) the overlap saves 10 ms (170 vs 180)
    const bool overlap = cap >= 16;
    if (cap >= 8 && !overlap) {
        tk = hite_prof_begin(ctx, wname, st);
        for (int level = 8; level <= cap; level *= 2) {
            int64_t cnt = 0;
            ACHK(build_list(ctx, S, st, total_rows, order, strips, P, 0, level, cap, flag, nullptr, pos, colpos, scan_tmp, list, nullptr, &cnt, nullptr));
            if (cnt == 0) continue;
            int64_t nl8 = 0;
            ACHK(assign_records(ctx, S, st, list, cnt, strips, bk, boff, scan_tmp, rec, 1, 8, &nl8));
            if (getenv("HITE_ALIGN_DEBUG")) fprintf(stderr, "[align%s] level %d: pairs %lld  block-strips %lld  lane-parallel prefix %lld\n", tag ? tag : "", level, (long long)cnt, (long long)S->h_pin[1], (long long)nl8);
            if (nl8 > 0) {
                HITE_CHECK(ctx, hipEventRecord(S->ev_fork, st));
                HITE_CHECK(ctx, hipStreamWaitEvent(s2, S->ev_fork, 0));
                snprintf(lname, sizeof lname, "align_fwd_wide_lanes%s", tag ? tag : "");
                const int tl = hite_prof_begin(ctx, lname, s2);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(align_fwd_lanes_kernel<8>), dim3((unsigned)((nl8 + 7) / 8)), dim3(64), 0, s2, P, list, (int)nl8);
                hite_prof_end(ctx, tl, s2);
                HITE_CHECK(ctx, hipEventRecord(S->ev_join, s2));
            }
            if (cnt > nl8)
                hipLaunchKernelGGL(align_fwd8_pair_kernel, dim3((unsigned)((cnt - nl8 + 31) / 32)), dim3(64), 0, st, P, list + nl8, (int)(cnt - nl8));
            if (nl8 > 0) HITE_CHECK(ctx, hipStreamWaitEvent(st, S->ev_join, 0));
        }
        hite_prof_end(ctx, tk, st);
    }
    if (overlap) {
        ACHK(build_list(ctx, S, st, total_rows, order, strips, P, 2, 0, cap, flag, nullptr, pos, colpos, scan_tmp, list_esc, nullptr, &n_esc, nullptr));
        if (n_esc > 0)
            ACHK(build_list(ctx, S, st, total_rows, order, strips, P, 3, 0, cap, flag, nullptr, pos, colpos, scan_tmp, list_fin, nullptr, &n_fin, nullptr));
    }
    if (n_esc > 0) {
        if (n_fin > 0) {
            tk = hite_prof_begin(ctx, name, st);
            hipLaunchKernelGGL(align_tb_kernel, dim3((unsigned)((n_fin + 63) / 64)), dim3(64), 0, st, P, list_fin, (int)n_fin);
            hite_prof_end(ctx, tk, st);
        }
        tk = hite_prof_begin(ctx, wname, s2);
        for (int level = 8; level <= cap; level *= 2) {
            int64_t cnt = 0;
            ACHK(build_list(ctx, S, s2, total_rows, order, strips, P, 0, level, cap, flag2, nullptr, pos2, colpos, scan_tmp2, list, nullptr, &cnt, nullptr, 8));
            if (cnt == 0) continue;
            ACHK(assign_records(ctx, S, s2, list, cnt, strips, bk2, boff2, scan_tmp2, rec, 10));
            const dim3 grid((unsigned)((cnt + 63) / 64));
            if (level == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(align_fwd_kernel<8>), grid, dim3(64), 0, s2, P, list, (int)cnt);
            else if (level == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(align_fwd_kernel<16>), grid, dim3(64), 0, s2, P, list, (int)cnt);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(align_fwd_kernel<32>), grid, dim3(64), 0, s2, P, list, (int)cnt);
        }
        hite_prof_end(ctx, tk, s2);
        HITE_CHECK(ctx, hipEventRecord(S->ev, s2));
        HITE_CHECK(ctx, hipStreamWaitEvent(st, S->ev, 0));
        tk = hite_prof_begin(ctx, name, st);
        hipLaunchKernelGGL(align_tb_kernel, dim3((unsigned)((n_esc + 63) / 64)), dim3(64), 0, st, P, list_esc, (int)n_esc);
        hite_prof_end(ctx, tk, st);
    } else {
        // ---- traceback on the slice of the run kept
        if (nltb > 0) {
            HITE_CHECK(ctx, hipEventRecord(S->ev_fork, st));
            HITE_CHECK(ctx, hipStreamWaitEvent(s2, S->ev_fork, 0));
            snprintf(lname, sizeof lname, "align_tb_strips%s", tag ? tag : "");
            const int tl = hite_prof_begin(ctx, lname, s2);
            hipLaunchKernelGGL(align_tb_strips_kernel, dim3((unsigned)nltb), dim3(64), 0, s2, P, order, (int)nltb);
            hite_prof_end(ctx, tl, s2);
            HITE_CHECK(ctx, hipEventRecord(S->ev_join, s2));
        }
        if (nrows > nltb) {
            tk = hite_prof_begin(ctx, name, st);
            hipLaunchKernelGGL(align_tb_kernel, dim3((unsigned)((nrows - nltb + 63) / 64)), dim3(64), 0, st, P, order + nltb, (int)(nrows - nltb));
            hite_prof_end(ctx, tk, st);
        }
        if (nltb > 0) HITE_CHECK(ctx, hipStreamWaitEvent(st, S->ev_join, 0));
    }
    HITE_CHECK(ctx, hipGetLastError());
    // ---- fall-back: the pairs whose path left the slice
    {
        int64_t cnt = 0, tcols = 0;
        ACHK(build_list(ctx, S, st, total_rows, order, strips, P, 1, 0, cap, flag, cols, pos, colpos, scan_tmp, list, full_off, &cnt, &tcols));
        if (cnt > 0) {
            tk = hite_prof_begin(ctx, "align_fallback", st);
            ACHK(aalloc(ctx, A, (size_t)tcols * 2 * AL_WIDE_NW + 64, &P.fullbuf));
            ACHK(aalloc(ctx, A, (size_t)tcols + 16, &P.fullt));
            hipLaunchKernelGGL(align_wide_fwd_kernel, dim3((unsigned)cnt), dim3(64), 0, st, P, list, (int)cnt);
            hipLaunchKernelGGL(align_wide_tb_kernel, dim3((unsigned)cnt), dim3(64), 0, st, P, list, (int)cnt);
            hite_prof_end(ctx, tk, st);
        }
    }
    HITE_CHECK(ctx, hipMemsetAsync(S->d_scal, 0, 64, st));
    hipLaunchKernelGGL(align_stats_kernel, dim3((unsigned)(total_rows + 255 < 1024 * 256 ? (total_rows + 255) / 256 : 1024)), dim3(256), 0, st, total_rows, strips, P, d_row_dead, d_cand_flag,
                       (unsigned long long *)S->d_scal);
    if (d_info) hipLaunchKernelGGL(align_info_kernel, dim3((unsigned)((total_rows + 255) / 256)), dim3(256), 0, st, total_rows, P, d_info);
    HITE_CHECK(ctx, hipMemcpyAsync(S->h_pin, S->d_scal, 56, hipMemcpyDeviceToHost, st));
    HITE_CHECK(ctx, hipStreamSynchronize(st));
    for (int q = 0; q < 7; q++) S->stats[q] += S->h_pin[q];
    HITE_CHECK(ctx, hipGetLastError());
    return HITE_OK;
}
