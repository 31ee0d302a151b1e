// hite_pipeline.hip -- device-resident driver of the fine (dynamic-boundary) stage: the body of
// flank_region_align_v5 (/root/reference/module/Util.py:8032-8194) from the copy table onwards,
// one batched call per stage instead of one forked process per candidate:
//
//   copies --window rules--> rows (<=100 longest per candidate, Util.py:10409 / ready_for_MSA.sh)
//          --flank gather (2-bit genome, revcomp in registers)--> windows
//          --star alignment (hite_msa.hip)--> rows x cols matrices
//          --remove_sparse_col--> cleaned matrices --judge_boundary_v5/v6/v9--> calls
//
// Candidates that have a copy window > 1000 bp are judged first on the first500+last500 form of
// those copies only, and on the full windows only if that passed (run_find_members_v8,
// Util.py:10439-10449).  Everything stays in HBM; the host reads back a handful of totals per
// pass to size the next allocation.
#include "hite_common.h"
#include "hite_genome.h"
#include "hite_scan.h"
#include "hite_arena.h"
#include <vector>

extern "C" int hite_judge_dev(hite_ctx *ctx, int32_t te_type, int32_t plant, int32_t n, const uint8_t *d_msa,
                              const int64_t *d_msa_off, const int32_t *d_rows, const int32_t *d_cols,
                              const uint8_t *d_cand, const int64_t *d_cand_off, const int64_t *d_col_off,
                              int32_t max_cols, int32_t max_rows, hite_call *d_calls, uint8_t *d_cons, void *stream);

extern "C" int hite_judge_lds_enabled(int32_t n);
extern "C" int hite_judge_classify_dev(hite_ctx *ctx, int32_t n, const int32_t *d_rows, const int32_t *d_cols, uint8_t *d_cls, void *stream);

#define MAXROWS 100

struct PipeState {
    Arena keep, tmp;
    int64_t *h_pin = nullptr;  // pinned scalars for read-backs
    int64_t *d_scal = nullptr;
};

extern "C" void hite_pipeline_release(void *state) {
    PipeState *s = (PipeState *)state;
    if (!s) return;
    arena_free(s->keep);
    arena_free(s->tmp);
    if (s->h_pin) (void)hipHostFree(s->h_pin);
    if (s->d_scal) (void)hipFree(s->d_scal);
    delete s;
}

template <typename TIn>
static int scan_excl(hite_ctx *ctx, Arena &tmp, const TIn *d_in, int64_t n, int64_t *d_out /* n+1 */, hipStream_t st) {
    void *bs = nullptr;
    int rc = arena_alloc(ctx, tmp, (size_t)scan_tmp_elems(n) * 8, &bs);
    if (rc) return rc;
    return scan_excl_buf<TIn>(ctx, (int64_t *)bs, d_in, n, d_out, st);
}

// ---------------------------------------------------------------------------------------------
// stage kernels
// ---------------------------------------------------------------------------------------------
// mode of pass A: 2 = judge first on first500+last500 of the >1000 windows, 1 = full windows, 0 = no copy
__global__ void mode_a_kernel(int n, const int32_t *__restrict__ copy_first, const int64_t *__restrict__ len,
                              int32_t *__restrict__ mode) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    int any = 0, big = 0;
    for (int i = copy_first[c]; i < copy_first[c + 1]; i++) { any |= len[i] > 0; big |= len[i] > 1000; }
    mode[c] = big ? 2 : (any ? 1 : 0);
}
// mode of pass B: full windows for the candidates whose truncated form passed
__global__ void mode_b_kernel(int n, const int32_t *__restrict__ mode_a, const hite_call *__restrict__ calls_a,
                              int32_t *__restrict__ mode) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    mode[c] = (mode_a[c] == 2 && calls_a[c].is_te) ? 1 : 0;
}

// >>> copy_name_cmp (tests/test_host_compiled.py compiles this block for the host and compares it with Python's string order)
// The windows of a candidate are named "<contig>:<start>-<end>(<strand>)" (Util.py:8110) and tools/ready_for_MSA.sh keeps the 100
// longest by `sort -nk 2 -r` on the .fai: equal lengths fall to the reverse byte order of the line, i.e. of the NAME (POSIX
// locale, the reference's container).  Byte order of two such names without building them: the contigs by a rank the host
// computed on "<contig>:" (hite_set_contig_order; identity when none was given), then start and end as DECIMAL STRINGS (a
// string that is a prefix of the other is the smaller: its next byte, '-' or '(', sorts before every digit), then '+' < '-'.
__device__ __forceinline__ int dec_digits(long long v) {
    int d = 1;
    while (v >= 10) { v /= 10; d++; }
    return d;
}
__device__ __forceinline__ long long pow10ll(int k) { long long r = 1; while (k-- > 0) r *= 10; return r; }
__device__ __forceinline__ int dec_str_cmp(long long a, long long b) {
    if (a == b) return 0;
    const int da = dec_digits(a), db = dec_digits(b);
    const int k = da < db ? da : db;
    const long long ta = a / pow10ll(da - k), tb = b / pow10ll(db - k);     // the first k digits of each
    if (ta != tb) return ta < tb ? -1 : 1;
    return da < db ? -1 : 1;
}
// is the name of copy (c1, s1, e1, m1) greater than that of (c2, s2, e2, m2)?  r1 / r2 = ranks of the contigs
__device__ __forceinline__ bool copy_name_gt(int r1, long long s1, long long e1, int m1, int r2, long long s2, long long e2, int m2) {
    if (r1 != r2) return r1 > r2;
    int c = dec_str_cmp(s1, s2);
    if (c) return c > 0;
    c = dec_str_cmp(e1, e2);
    if (c) return c > 0;
    return m1 > m2;
}
// <<< copy_name_cmp

// rows of each candidate: eligible copies, at most the 100 longest (ties: the greater name first, as ready_for_MSA.sh's sort
// leaves them), input order kept
__global__ void __launch_bounds__(256) select_rows_kernel(int n, const int32_t *__restrict__ copy_first,
                                                          const int64_t *__restrict__ len,
                                                          const int32_t *__restrict__ mode, const int32_t *__restrict__ contig,
                                                          const int64_t *__restrict__ s1, const int64_t *__restrict__ e1,
                                                          const uint8_t *__restrict__ minus, const int32_t *__restrict__ contig_rank,
                                                          int32_t *__restrict__ nrows, int32_t *__restrict__ sel /* n x 100 */) {
    __shared__ int s_scan[8];
    __shared__ int s_cnt;
    int c = blockIdx.x;
    if (c >= n) return;
    const int md = mode[c];
    const int f = copy_first[c], k = copy_first[c + 1] - f;
    if (md == 0 || k <= 0) { if (threadIdx.x == 0) nrows[c] = 0; return; }
    // count eligible
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int local = 0;
    for (int i = threadIdx.x; i < k; i += 256) { int64_t L = len[f + i]; local += md == 2 ? (L > 1000) : (L > 0); }
    atomicAdd(&s_cnt, local);
    __syncthreads();
    const int E = s_cnt;
    int running = 0;
    for (int base = 0; base < k; base += 256) {
        int i = base + threadIdx.x;
        int keep = 0;
        if (i < k) {
            int64_t L = len[f + i];
            bool el = md == 2 ? (L > 1000) : (L > 0);
            if (el) {
                if (E <= MAXROWS) keep = 1;
                else {
                    int64_t Li = md == 2 ? 1000 : L;
                    const int ri = contig_rank ? contig_rank[contig[f + i]] : contig[f + i];
                    const long long si = s1[f + i], ei = e1[f + i];
                    const int mi = minus[f + i] != 0;
                    int rank = 0;
                    for (int j = 0; j < k; j++) {
                        int64_t Lj = len[f + j];
                        bool ej = md == 2 ? (Lj > 1000) : (Lj > 0);
                        if (!ej) continue;
                        if (md == 2) Lj = 1000;
                        if (Lj > Li) rank++;
                        else if (Lj == Li && j != i) {
                            const int rj = contig_rank ? contig_rank[contig[f + j]] : contig[f + j];
                            const int mj = minus[f + j] != 0;
                            const bool gt = copy_name_gt(rj, s1[f + j], e1[f + j], mj, ri, si, ei, mi);
                            // two copies with the same name (the same interval twice): the earlier one first
                            const bool same = rj == ri && s1[f + j] == si && e1[f + j] == ei && mj == mi;
                            rank += gt || (same && j < i);
                        }
                    }
                    keep = rank < MAXROWS;
                }
            }
        }
        int tot;
        int pre = block_excl_scan(keep, s_scan, &tot);
        if (keep) sel[(int64_t)c * MAXROWS + running + pre] = f + i;
        running += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) nrows[c] = running;
}

// Pads of a row (copy cp) of a candidate whose centre is copy cp0: the candidate bases the copy's record leaves out in front of /
// behind its window (clip word = left | right << 16 in the orientation of the GENOME; a minus copy's window is reverse-complemented),
// less what the CENTRE's own record leaves out on that side -- the row's first base faces centre position a_row - a_centre, and
// that many pad bytes put it there (a row that reaches further than the centre just begins with bases the centre lacks).
__device__ __forceinline__ void pad_lengths(const uint32_t *__restrict__ clip, const uint8_t *__restrict__ minus, int cp, int cp0, int &a, int &b) {
    a = 0; b = 0;
    if (!clip || cp == cp0) return;
    const uint32_t cl = clip[cp], c0 = clip[cp0];
    const bool mn = minus[cp] != 0, m0 = minus[cp0] != 0;
    const int ar = mn ? (int)(cl >> 16) : (int)(cl & 0xffffu), br = mn ? (int)(cl & 0xffffu) : (int)(cl >> 16);
    const int a0 = m0 ? (int)(c0 >> 16) : (int)(c0 & 0xffffu), b0 = m0 ? (int)(c0 & 0xffffu) : (int)(c0 >> 16);
    a = ar > a0 ? ar - a0 : 0;
    b = br > b0 ? br - b0 : 0;
}

__global__ void row_meta_kernel(int n, const int64_t *__restrict__ row_first, const int32_t *__restrict__ nrows,
                                const int32_t *__restrict__ sel, const int32_t *__restrict__ mode,
                                const int64_t *__restrict__ len, int32_t *__restrict__ row_first32,
                                int32_t *__restrict__ row_copy, int32_t *__restrict__ row_len,
                                int32_t *__restrict__ row_pad, uint8_t *__restrict__ row_trunc,
                                int32_t *__restrict__ maxlen, const uint32_t *__restrict__ clip /* per copy, or NULL */,
                                const uint8_t *__restrict__ minus,
                                int32_t *__restrict__ row_cp0 /* per row: the copy of its candidate's centre (clip != NULL) */) {
    int c = blockIdx.x;
    if (c >= n) return;
    if (threadIdx.x == 0) { row_first32[c] = (int32_t)row_first[c]; if (c == n - 1) row_first32[n] = (int32_t)row_first[n]; }
    int R = nrows[c];
    int md = mode[c];
    int mx = 0;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        int64_t g = row_first[c] + r;
        int cp = sel[(int64_t)c * MAXROWS + r];
        // (the rows of the first500 + last500 form are cut from the PADDED window: still 1000 bytes; the centre, row 0, is never padded)
        int pa, pb;
        pad_lengths(clip, minus, cp, sel[(int64_t)c * MAXROWS], pa, pb);
        int L = md == 2 ? 1000 : (int)len[cp] + pa + pb;
        row_copy[g] = cp;
        if (clip) row_cp0[g] = sel[(int64_t)c * MAXROWS];
        row_len[g] = L;
        row_pad[g] = (L + 15) & ~15;
        row_trunc[g] = md == 2;
        mx = L > mx ? L : mx;
    }
    // one atomic per wavefront (one per row on a single address was half a millisecond per launch) -- and only from a wavefront
    // whose maximum beats what the word already holds (round 6: a maximum only grows, so a stale read can at worst let a needless
    // atomic through; 100 000 wavefronts on one address were still 0.4 ms per launch)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(mx, d, 64); mx = o > mx ? o : mx; }
    if ((threadIdx.x & 63) == 0 && mx > 0 && mx > __atomic_load_n(maxlen, __ATOMIC_RELAXED)) atomicMax(maxlen, mx);
}

// one byte of a window (position p of the window [g_lo, g_lo + wlen) read on the given strand)
__device__ __forceinline__ unsigned window_byte(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask, int64_t g_lo, int64_t wlen,
                                                bool minus, int64_t p) {
    if (!minus) return fetch4(bases, nmask, g_lo + p) & 0xffu;
    const unsigned c = fetch4(bases, nmask, g_lo + wlen - 1 - p) & 0xffu;
    return c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'N';
}
// the centre's window, for the pads of a row: pad byte i in front = centre base i, pad byte j behind (of b) = centre base m - b + j,
// in LOWER CASE (a pad that matches the centre base it faces: the row's path starts and ends on the centre's diagonal at no cost);
// HITE_ROW_PAD where the centre has no such position
struct PadSrc { int64_t g_lo, len; bool minus; };
// positions [ws, ws + cnt) of a window padded by `a` bytes in front and `b` behind
__device__ __forceinline__ void emit_padded(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask, int64_t g_lo, int64_t len,
                                            bool mn, int64_t a, int64_t b, int64_t ws, int64_t cnt, uint8_t *__restrict__ dst, int lane,
                                            const PadSrc &C) {
    int64_t n1 = a - ws; n1 = n1 < 0 ? 0 : (n1 > cnt ? cnt : n1);                 // pad bytes in front
    const int64_t off = ws + n1 - a;                                              // first window position wanted
    int64_t n2 = len - off; n2 = n2 < 0 ? 0 : (n2 > cnt - n1 ? cnt - n1 : n2);    // genome bytes
    // the pads in front and the first bases behind them, up to the next 16-byte edge of the slot, byte by byte in ONE loop; the rest of
    // the bases then leaves in 16-byte stores (emit_span on an aligned address)
    int64_t lead = n2 > 0 ? (int64_t)((0 - (uintptr_t)(dst + n1)) & 15) : 0;
    if (lead > n2) lead = n2;
    for (int64_t i = lane; i < n1 + lead; i += 64) {
        unsigned ch;
        if (i < n1) {
            const int64_t cpos = ws + i;
            ch = cpos < C.len ? (window_byte(bases, nmask, C.g_lo, C.len, C.minus, cpos) | 0x20u) : (unsigned)HITE_ROW_PAD;
        } else ch = window_byte(bases, nmask, g_lo, len, mn, off + (i - n1));
        dst[i] = (uint8_t)ch;
    }
    if (n2 > lead) emit_span(bases, nmask, g_lo, len, mn, off + lead, n2 - lead, dst + n1 + lead, lane);
    for (int64_t i = n1 + n2 + lane; i < cnt; i += 64) {
        const int64_t j = ws + i - a - len;                                       // pad byte j of the b behind
        const int64_t cpos = C.len - b + j;
        dst[i] = (cpos >= 0 && cpos < C.len) ? (uint8_t)(window_byte(bases, nmask, C.g_lo, C.len, C.minus, cpos) | 0x20u) : (uint8_t)HITE_ROW_PAD;
    }
}

// one wavefront per row: window (or its first500+last500 form) from the packed genome.  clip != NULL (copy records in the reference's
// coordinates, the aligned interval of Util.py:8026): the window is padded by the candidate bases the copy finder's end extensions
// clipped -- in front by the left clip (the right one for a minus copy: the window is reverse-complemented), behind by the other --
// with pad bytes (HITE_IS_ROW_PAD): the CENTRE's own first / last bases in lower case, which match the centre positions they face
// and leave the alignment as '-'.  The row then faces the part of the centre it was found with and its path stays on the diagonal,
// instead of opening a gap of the clipped length at 3 per base.
__global__ void __launch_bounds__(256) row_gather_kernel(const uint32_t *__restrict__ bases,
                                                         const uint32_t *__restrict__ nmask,
                                                         const int64_t *__restrict__ coff, int32_t ncontig,
                                                         int64_t nrows_total, const int32_t *__restrict__ row_copy,
                                                         const uint8_t *__restrict__ row_trunc,
                                                         const int32_t *__restrict__ contig,
                                                         const int64_t *__restrict__ s1, const int64_t *__restrict__ e1,
                                                         const uint8_t *__restrict__ minus, int32_t flank,
                                                         const int64_t *__restrict__ win_off, uint8_t *__restrict__ win,
                                                         const uint32_t *__restrict__ clip, const int32_t *__restrict__ row_cp0) {
    int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= nrows_total) return;
    int lane = threadIdx.x & 63;
    int cp = row_copy[g];
    int64_t len, tlen, g_lo;
    window_rule(coff, ncontig, contig[cp], s1[cp], e1[cp], flank, len, tlen, g_lo);
    if (len == 0) return;
    bool mn = minus[cp] != 0;
    uint8_t *dst = win + win_off[g];
    // (the centre, the first row of the candidate that owns this copy, is never padded itself)
    const int cp0 = clip ? row_cp0[g] : cp;
    int pa, pb;
    pad_lengths(clip, minus, cp, cp0, pa, pb);
    if (pa | pb) {
        const int64_t a = pa, b = pb;
        const int64_t plen = a + len + b;
        PadSrc C;
        int64_t tl0;
        window_rule(coff, ncontig, contig[cp0], s1[cp0], e1[cp0], flank, C.len, tl0, C.g_lo);
        C.minus = minus[cp0] != 0;
        if (row_trunc[g]) {
            emit_padded(bases, nmask, g_lo, len, mn, a, b, 0, 500, dst, lane, C);
            emit_padded(bases, nmask, g_lo, len, mn, a, b, plen - 500, 500, dst + 500, lane, C);
        } else {
            emit_padded(bases, nmask, g_lo, len, mn, a, b, 0, plen, dst, lane, C);
        }
        return;
    }
    if (row_trunc[g]) {
        emit_span(bases, nmask, g_lo, len, mn, 0, 500, dst, lane);
        emit_span(bases, nmask, g_lo, len, mn, len - 500, 500, dst + 500, lane);
    } else {
        emit_span(bases, nmask, g_lo, len, mn, 0, len, dst, lane);
    }
}

// also accumulates the alignment stage's algorithmic bytes (windows read per pair + ops written) and
// its anti-diagonal steps (x64 = DP cells) into acc[0], acc[1]
__global__ void ops_count_kernel(int n, const int64_t *__restrict__ row_first, const int32_t *__restrict__ nrows,
                                 const int32_t *__restrict__ row_len, int64_t *__restrict__ cnt,
                                 unsigned long long *__restrict__ acc) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bytes = 0, steps = 0;
    if (c < n) {
        int R = nrows[c];
        if (R <= 0) cnt[c] = 0;
        else {
            int64_t g0 = row_first[c];
            int64_t m = row_len[g0];
            cnt[c] = (int64_t)(R + 1) * (m + 1);
            for (int r = 1; r < R; r++) {
                int64_t nn = row_len[g0 + r];
                bytes += (unsigned long long)(m + nn + 2 * (m + 1));
                steps += (unsigned long long)(m + nn);
            }
        }
    }
    // one pair of atomics per wavefront (round 6: two per candidate on two addresses were 100 000 per launch)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        bytes += ((unsigned long long)(unsigned)__shfl_xor((int)(bytes >> 32), d, 64) << 32) | (unsigned)__shfl_xor((int)bytes, d, 64);
        steps += ((unsigned long long)(unsigned)__shfl_xor((int)(steps >> 32), d, 64) << 32) | (unsigned)__shfl_xor((int)steps, d, 64);
    }
    if ((threadIdx.x & 63) == 0 && bytes) { atomicAdd(&acc[0], bytes); atomicAdd(&acc[1], steps); }
}

// bytes of each alignment in HBM: none for the classes whose judge kernel builds the alignment in LDS (cls may be NULL)
__global__ void msa_size_kernel(int n, const int32_t *__restrict__ nrows, const int32_t *__restrict__ cols,
                                const uint8_t *__restrict__ cls, int64_t *__restrict__ bytes, int32_t *__restrict__ maxcols) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    int64_t b = (cls && cls[c] >= JUDGE_CLS_LDS) ? 0 : (int64_t)nrows[c] * cols[c];
    bytes[c] = (b + 15) & ~(int64_t)15;
    if (cols[c] > __atomic_load_n(maxcols, __ATOMIC_RELAXED)) atomicMax(maxcols, cols[c]);      // (a maximum only grows: a stale read lets at worst a needless atomic through)
}

// rows visible to sparse-col / judge: 0 rows where the alignment failed
__global__ void eff_rows_kernel(int n, const int32_t *__restrict__ nrows, const int32_t *__restrict__ cols,
                                int32_t *__restrict__ eff) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    eff[c] = cols[c] > 0 ? nrows[c] : 0;
}

// remove_sparse_col on a batch where some slots are empty
extern "C" int hite_sparse_cols_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_msa, const int64_t *d_msa_off,
                                    const int32_t *d_rows, const int32_t *d_cols, const int64_t *d_col_off,
                                    int64_t total_cols, uint8_t *d_out, int32_t *d_new_cols, void *stream);

// final record per candidate + length of the consensus to keep
__global__ void merge_calls_kernel(int n, const int32_t *__restrict__ mode_a, const hite_call *__restrict__ ca,
                                   const hite_call *__restrict__ cb, hite_call *__restrict__ out,
                                   int32_t *__restrict__ src_pass, int64_t *__restrict__ keep_len) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    hite_call r;
    r.is_te = 0; r.info = 0; r.row_num = 0; r.bstart = -1; r.bend = -1; r.cons_len = 0; r.cons_off = 0;
    int src = 0;
    if (mode_a[c] == 1) { r = ca[c]; src = 1; }
    else if (mode_a[c] == 2) {
        if (ca[c].is_te && cb) { r = cb[c]; src = 2; }
        else { r = ca[c]; src = 1; }
    }
    if (!r.is_te) r.cons_len = 0;
    out[c] = r;
    src_pass[c] = r.is_te ? src : 0;
    keep_len[c] = r.is_te ? r.cons_len : 0;
}
__global__ void __launch_bounds__(256) copy_cons_kernel(int n, hite_call *__restrict__ calls,
                                                        const int32_t *__restrict__ src_pass,
                                                        const uint8_t *__restrict__ cons_a,
                                                        const uint8_t *__restrict__ cons_b,
                                                        const int64_t *__restrict__ out_off, uint8_t *__restrict__ out,
                                                        int64_t out_cap) {
    int c = blockIdx.x;
    if (c >= n) return;
    int sp = src_pass[c];
    if (!sp) { if (threadIdx.x == 0) calls[c].cons_off = out_off[c]; return; }
    const uint8_t *src = (sp == 1 ? cons_a : cons_b) + calls[c].cons_off;
    int L = calls[c].cons_len;
    int64_t o = out_off[c];
    if (o + L <= out_cap) for (int i = threadIdx.x; i < L; i += 256) out[o + i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) calls[c].cons_off = o;
}

// ---------------------------------------------------------------------------------------------
// one pass over all candidates with the given per-candidate mode
// ---------------------------------------------------------------------------------------------
struct PassOut {
    hite_call *calls = nullptr;  // keep arena
    uint8_t *cons = nullptr;     // keep arena
};
struct StageTimes;  // (events are recorded by bench.py around the whole call; per-kernel via rocprof)

static int read_scalars(hite_ctx *ctx, PipeState *S, hipStream_t st, int count) {
    HITE_CHECK(ctx, hipMemcpyAsync(S->h_pin, S->d_scal, sizeof(int64_t) * count, hipMemcpyDeviceToHost, st));
    HITE_CHECK(ctx, hipStreamSynchronize(st));
    return HITE_OK;
}

#define ACHK(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

static int run_pass(hite_ctx *ctx, PipeState *S, int te_type, int plant, int n, const uint8_t *d_cand,
                    const int64_t *d_cand_off, const int32_t *d_copy_first, const int32_t *d_contig,
                    const int64_t *d_s1, const int64_t *d_e1, const uint8_t *d_minus, const uint32_t *d_clip, int flank, const int64_t *d_len,
                    const int32_t *d_mode, PassOut *out, int64_t *stats /* rows, win_bytes, msa_bytes, align_bytes */,
                    int64_t *extra /* steps */, int64_t *cols_acc /* cleaned columns, both passes */, bool pass_b, hipStream_t st) {
    Arena &T = S->tmp, &K = S->keep;
    int32_t *nrows, *sel, *row_first32, *row_copy, *row_len, *row_pad, *cols, *status, *eff, *new_cols;
    int64_t *row_first, *win_off, *ops_cnt, *ops_base, *msa_bytes, *msa_off, *col_off2;
    uint8_t *row_trunc, *win, *clean;
    void *p;
    ACHK(arena_alloc(ctx, T, (size_t)n * 4, &p)); nrows = (int32_t *)p;
    ACHK(arena_alloc(ctx, T, (size_t)n * MAXROWS * 4, &p)); sel = (int32_t *)p;
    int tk = hite_prof_begin(ctx, "select_rows_kernel", st);
    hipLaunchKernelGGL(select_rows_kernel, dim3(n), dim3(256), 0, st, n, d_copy_first, d_len, d_mode, d_contig, d_s1, d_e1, d_minus,
                       (const int32_t *)ctx->d_contig_rank, nrows, sel);
    hite_prof_end(ctx, tk, st);
    ACHK(arena_alloc(ctx, T, (size_t)(n + 1) * 8, &p)); row_first = (int64_t *)p;
    ACHK(scan_excl<int32_t>(ctx, T, nrows, n, row_first, st));
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal, row_first + n, 8, hipMemcpyDeviceToDevice, st));
    ACHK(read_scalars(ctx, S, st, 1));
    const int64_t total_rows = S->h_pin[0];
    stats[0] += total_rows;
    ACHK(arena_alloc(ctx, K, sizeof(hite_call) * (size_t)n, &p)); out->calls = (hite_call *)p;
    HITE_CHECK(ctx, hipMemsetAsync(out->calls, 0, sizeof(hite_call) * (size_t)n, st));
    if (total_rows == 0) { ACHK(arena_alloc(ctx, K, 256, &p)); out->cons = (uint8_t *)p; return HITE_OK; }
    if (total_rows > 0x7fffffff) return HITE_EINVAL;

    ACHK(arena_alloc(ctx, T, (size_t)(n + 1) * 4, &p)); row_first32 = (int32_t *)p;
    ACHK(arena_alloc(ctx, T, (size_t)total_rows * 4, &p)); row_copy = (int32_t *)p;
    ACHK(arena_alloc(ctx, T, (size_t)total_rows * 4, &p)); row_len = (int32_t *)p;
    ACHK(arena_alloc(ctx, T, (size_t)total_rows * 4, &p)); row_pad = (int32_t *)p;
    ACHK(arena_alloc(ctx, T, (size_t)total_rows, &p)); row_trunc = (uint8_t *)p;
    int32_t *row_cp0 = nullptr;
    if (d_clip) { ACHK(arena_alloc(ctx, T, (size_t)total_rows * 4, &p)); row_cp0 = (int32_t *)p; }
    HITE_CHECK(ctx, hipMemsetAsync(S->d_scal, 0, 64, st));
    hipLaunchKernelGGL(row_meta_kernel, dim3(n), dim3(128), 0, st, n, row_first, nrows, sel, d_mode, d_len, row_first32,
                       row_copy, row_len, row_pad, row_trunc, (int32_t *)(S->d_scal + 2), d_clip, d_minus, row_cp0);
    ACHK(arena_alloc(ctx, T, (size_t)(total_rows + 1) * 8, &p)); win_off = (int64_t *)p;
    ACHK(scan_excl<int32_t>(ctx, T, row_pad, total_rows, win_off, st));
    ACHK(arena_alloc(ctx, T, (size_t)n * 8, &p)); ops_cnt = (int64_t *)p;
    ACHK(arena_alloc(ctx, T, (size_t)(n + 1) * 8, &p)); ops_base = (int64_t *)p;
    hipLaunchKernelGGL(ops_count_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, row_first, nrows, row_len, ops_cnt,
                       (unsigned long long *)(S->d_scal + 4));
    ACHK(scan_excl<int64_t>(ctx, T, ops_cnt, n, ops_base, st));
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal, win_off + total_rows, 8, hipMemcpyDeviceToDevice, st));
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal + 1, ops_base + n, 8, hipMemcpyDeviceToDevice, st));
    ACHK(read_scalars(ctx, S, st, 6));
    const int64_t win_bytes = S->h_pin[0], ops_elems = S->h_pin[1];
    const int max_len = (int)(((int32_t *)(S->h_pin + 2))[0]);
    stats[1] += win_bytes;
    stats[3] += S->h_pin[4];  // alignment algorithmic bytes
    extra[0] += S->h_pin[5];  // anti-diagonal steps

    ACHK(arena_alloc(ctx, T, (size_t)win_bytes + 64, &p)); win = (uint8_t *)p;
    tk = hite_prof_begin(ctx, "row_gather_kernel", st);
    hipLaunchKernelGGL(row_gather_kernel, dim3((unsigned)((total_rows + 3) / 4)), dim3(256), 0, st, ctx->d_bases, ctx->d_nmask,
                       ctx->d_contig_off, ctx->n_contigs, total_rows, row_copy, row_trunc, d_contig, d_s1, d_e1, d_minus,
                       flank, win_off, win, d_clip, row_cp0);
    hite_prof_end(ctx, tk, st);
    HITE_CHECK(ctx, hipGetLastError());

    ACHK(arena_alloc(ctx, T, (size_t)n * 4, &p)); cols = (int32_t *)p;
    ACHK(arena_alloc(ctx, T, (size_t)n * 4, &p)); status = (int32_t *)p;
    ACHK(arena_alloc(ctx, T, (size_t)n * 4, &p)); new_cols = (int32_t *)p;
    int32_t *last_extra;
    ACHK(arena_alloc(ctx, T, (size_t)n * 4, &p)); last_extra = (int32_t *)p;
    // alignment + layout + sparse-column selection: the full alignment is never written
    int32_t *rows_al;
    ACHK(arena_alloc(ctx, T, (size_t)n * 4, &p)); rows_al = (int32_t *)p;
    ACHK(hite_star_msa_sparse_dev(ctx, n, win, win_off, row_len, row_first32, total_rows, ops_base, ops_elems, max_len > 0 ? max_len : 1,
                                  cols, status, new_cols, last_extra, rows_al, st));
    ACHK(arena_alloc(ctx, T, (size_t)n * 4, &p)); eff = (int32_t *)p;
    hipLaunchKernelGGL(eff_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, rows_al, cols, eff);
    ACHK(arena_alloc(ctx, T, (size_t)n * 8, &p)); msa_bytes = (int64_t *)p;
    ACHK(arena_alloc(ctx, T, (size_t)(n + 1) * 8, &p)); msa_off = (int64_t *)p;
    ACHK(arena_alloc(ctx, T, (size_t)(n + 1) * 8, &p)); col_off2 = (int64_t *)p;
    HITE_CHECK(ctx, hipMemsetAsync(S->d_scal, 0, 64, st));
    // fill fused into the judge, when the judge's LDS kernels are switched on (HITE_JUDGE_LDS=1; HITE_JUDGE_FUSE=0 then makes
    // them copy their alignment from HBM instead): the LDS classes are left out of the fill, their kernels build the alignment
    // from windows + ops + layout words
    uint8_t *jcls = nullptr;
    {
        const char *e = getenv("HITE_JUDGE_FUSE");
        if (!(e && *e && atoi(e) == 0) && hite_judge_lds_enabled(n)) {
            ACHK(arena_alloc(ctx, T, (size_t)n + 16, &p)); jcls = (uint8_t *)p;
            ACHK(hite_judge_classify_dev(ctx, n, eff, new_cols, jcls, st));
        }
    }
    hipLaunchKernelGGL(msa_size_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, eff, new_cols, jcls, msa_bytes, (int32_t *)(S->d_scal + 2));
    ACHK(scan_excl<int64_t>(ctx, T, msa_bytes, n, msa_off, st));
    ACHK(scan_excl<int32_t>(ctx, T, new_cols, n, col_off2, st));
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal, col_off2 + n, 8, hipMemcpyDeviceToDevice, st));
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal + 1, msa_off + n, 8, hipMemcpyDeviceToDevice, st));
    ACHK(read_scalars(ctx, S, st, 3));
    const int64_t msa_total = S->h_pin[1];
    stats[2] += msa_total;
    ACHK(arena_alloc(ctx, T, (size_t)msa_total + 64, &p)); clean = (uint8_t *)p;
    tk = hite_prof_begin(ctx, extra == nullptr ? "star_fill_sparse_kernel" : (pass_b ? "star_fill_sparse_kernel_passB" : "star_fill_sparse_kernel_passA"), st);
    // (what the judge's LDS kernels read is what the fill reads: the state of the last hite_star_msa_sparse_dev call)
    struct FuseGuard {      // cleared on every way out: a later stand-alone hite_judge_dev call must not see this pass's buffers
        hite_ctx *c;
        ~FuseGuard() { c->d_judge_cls = nullptr; c->judge_fuse = JudgeFuse{}; }
    } fuse_guard{ctx};
    if (jcls) {
        JudgeFuse F;
        F.win = win; F.win_off = ctx->d_msa_win_off ? ctx->d_msa_win_off : win_off; F.win_len = ctx->d_msa_win_len ? ctx->d_msa_win_len : row_len;
        F.row_first = row_first32; F.ops_base = ops_base;
        F.ops = (const uint16_t *)ctx->d_scratch2; F.lay = ctx->d_msa_lay; F.last_extra = last_extra;
        F.row_map = ctx->d_msa_row_map; F.rows_eff = ctx->d_msa_rows_eff;
        ctx->judge_fuse = F; ctx->d_judge_cls = jcls;
    }
    ACHK(hite_star_msa_fill_sparse_dev(ctx, n, win, win_off, row_len, row_first32, ops_base, new_cols, last_extra, msa_off, clean, st));
    hite_prof_end(ctx, tk, st);
    const int64_t total_cols2 = S->h_pin[0];
    if (cols_acc) *cols_acc += total_cols2;
    const int max_cols2 = (int)(((int32_t *)(S->h_pin + 2))[0]);
    ACHK(arena_alloc(ctx, K, (size_t)total_cols2 + 8 * (size_t)n + 64, &p)); out->cons = (uint8_t *)p;
    tk = hite_prof_begin(ctx, extra == nullptr ? "judge_kernel" : (pass_b ? "judge_kernel_passB" : "judge_kernel_passA"), st);
    if (max_cols2 > 0)
        ACHK(hite_judge_dev(ctx, te_type, plant, n, clean, msa_off, eff, new_cols, d_cand, d_cand_off, col_off2, max_cols2,
                            MAXROWS + 1, out->calls, out->cons, st));
    hite_prof_end(ctx, tk, st);
    return HITE_OK;
}

// ---------------------------------------------------------------------------------------------
// Clip words for a copy table that carries none -- the reference's own 5-tuples (chr, reference_start + 1, reference_end, length,
// strand) of get_copies_minimap2, Util.py:8022-8030: minimap2's soft clips are dropped there, so nothing says which part of the
// candidate a record covers.  The probe finds it from the sequences (definition: orc_clip_probe, oracle/hite_oracle_copies.c):
// the first K = 21 bases of the record's interval (read on the candidate's strand) are laid on the candidate at every offset
// d = 0 .. min(|cand| - K, |cand| / 20 + 32) -- the reference's filter leaves <= 5 % of the candidate unaligned --, the offset
// with the fewest mismatches wins (the smallest on ties; N never matches), and is the left clip when it has <= 5 mismatches; if
// not (an indel inside the probe), the NEXT K bases are tried the same way at the same offsets.  The right clip the same from the
// other end.  An end that finds no offset takes what the other end's clip leaves of |cand| - |interval| (clamped to the offsets
// tried; 0 when neither end finds one).  One thread per (record, end): the candidate slides through a 63-bit register of 3-bit codes, one XOR,
// a fold and a population count per offset.
// ---------------------------------------------------------------------------------------------
#define CLIP_K 21
#define CLIP_MM 5
__device__ __forceinline__ unsigned long long probe_code(unsigned ch, unsigned long long nval) {     // A C G T -> 0 .. 3 (either case), else nval
    const unsigned c = ch & 0xdfu;
    return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? (unsigned long long)((c >> 1) & 3u) : nval;
}
__global__ void __launch_bounds__(256) clip_probe_kernel(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask,
                                                         const int64_t *__restrict__ coff, int32_t ncontig, int64_t n_copies, int32_t n_cand,
                                                         const int32_t *__restrict__ copy_first, const uint8_t *__restrict__ cand,
                                                         const int64_t *__restrict__ cand_off, const int32_t *__restrict__ contig,
                                                         const int64_t *__restrict__ s1, const int64_t *__restrict__ e1,
                                                         const uint8_t *__restrict__ minus, uint16_t *__restrict__ clip_out /* 2 per record */) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= 2 * n_copies) return;
    const int64_t cp = tid >> 1;
    const int right = (int)(tid & 1);              // lanes 2 k, 2 k + 1: the two ends of record k
    int lo = 0, hi = n_cand;                       // the candidate that owns the record: last c with copy_first[c] <= cp
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int64_t)copy_first[mid] <= cp) lo = mid; else hi = mid; }
    const uint8_t *q = cand + cand_off[lo];
    const int64_t Lq = cand_off[lo + 1] - cand_off[lo];
    const int ct = contig[cp];
    int64_t ys = s1[cp] - 1, ye = e1[cp];          // the record's interval, clamped to its contig like a window
    int clip = -1;                                  // -1: no offset within the mismatch bound
    int64_t dmax = 0, Ly = 0;
    bool mn = false;
    if (ct >= 0 && ct < ncontig) {
        const int64_t clen = coff[ct + 1] - coff[ct];
        if (ys < 0) ys = 0;
        if (ye > clen) ye = clen;
        Ly = ye - ys;
        const int64_t g_lo = coff[ct] + ys;
        mn = minus[cp] != 0;
        if (Ly >= CLIP_K && Lq >= CLIP_K) {
            dmax = Lq / 20 + 32;
            if (dmax > Lq - CLIP_K) dmax = Lq - CLIP_K;
            if (dmax > 0xffff) dmax = 0xffff;
            const unsigned long long M3 = 0x1249249249249249ull, M63 = 0x7fffffffffffffffull;
            for (int att = 0; att < 2 && clip < 0; att++) {
                const int64_t sh = (int64_t)att * CLIP_K;
                if (Ly < sh + CLIP_K) break;
                int64_t dm = dmax;
                if (dm > Lq - CLIP_K - sh) dm = Lq - CLIP_K - sh;
                if (dm < 0) break;
                unsigned long long Y = 0, Q = 0;
                // base i of a probe at bits 3 i
                for (int i = 0; i < CLIP_K; i++) {
                    const int64_t yp = right ? Ly - sh - CLIP_K + i : sh + i;
                    Y |= probe_code(window_byte(bases, nmask, g_lo, Ly, mn, yp), 4ull) << (3 * i);
                    const int64_t qp = right ? Lq - sh - CLIP_K + i : sh + i;
                    Q |= probe_code(q[qp], 7ull) << (3 * i);
                }
                int best = CLIP_K + 1, arg = 0;
                for (int64_t d = 0; d <= dm; d++) {
                    if (d > 0) {
                        if (!right) Q = (Q >> 3) | (probe_code(q[d + sh + CLIP_K - 1], 7ull) << (3 * (CLIP_K - 1)));     // the window moves up the candidate
                        else Q = ((Q << 3) & M63) | probe_code(q[Lq - d - sh - CLIP_K], 7ull);                          // ... down
                    }
                    const unsigned long long x = Q ^ Y;
                    const int mm = __popcll((x | (x >> 1) | (x >> 2)) & M3);
                    if (mm < best) { best = mm; arg = (int)d; }
                }
                if (best <= CLIP_MM) clip = arg;
            }
        }
    }
    // an end without an offset takes what the other end leaves of the length difference (no net insertion / deletion assumed)
    const int other = __shfl_xor(clip, 1, 64);
    if (clip < 0) {
        int64_t v = other >= 0 ? (Lq - Ly) - other : 0;
        clip = (int)(v < 0 ? 0 : (v > dmax ? dmax : v));
    }
    // the word is kept in the orientation of the genome: a minus record's left clip is the candidate's right one
    clip_out[2 * cp + ((right != 0) != mn ? 1 : 0)] = (uint16_t)clip;
}

// ---------------------------------------------------------------------------------------------
// public entry: the fine stage for one batch of candidates
// ---------------------------------------------------------------------------------------------
extern "C" int hite_flank_region_align_clip_dev(hite_ctx *ctx, void **state_io, int32_t te_type, int32_t plant, int32_t n_cand,
                                                const uint8_t *d_cand, const int64_t *d_cand_off, const int32_t *d_copy_first,
                                                int64_t n_copies, const int32_t *d_contig, const int64_t *d_start1,
                                                const int64_t *d_end1, const uint8_t *d_minus, const uint32_t *d_clip /* per copy, or NULL */,
                                                int32_t flank, hite_call *d_calls, uint8_t *d_cons, int64_t cons_cap,
                                                int64_t *stats_out /* 12 x int64, host, may be NULL */, void *stream) {
    if (!ctx || !ctx->d_bases || !state_io || n_cand < 0 || n_copies < 0 || te_type < 0 || te_type > 2) return HITE_EINVAL;
    if (n_cand == 0) return HITE_OK;
    hipStream_t st = (hipStream_t)stream;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    PipeState *S = (PipeState *)*state_io;
    if (!S) {
        S = new PipeState();
        HITE_CHECK(ctx, hipHostMalloc((void **)&S->h_pin, 64 * sizeof(int64_t)));
        HITE_CHECK(ctx, hipMalloc((void **)&S->d_scal, 64 * sizeof(int64_t)));
        *state_io = S;
    }
    ACHK(arena_reset(ctx, S->keep, true));
    ACHK(arena_reset(ctx, S->tmp, true));
    int64_t stats[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    void *p;
    int64_t *len;
    int32_t *mode_a, *mode_b, *src_pass;
    int64_t *keep_len, *out_off;
    ACHK(arena_alloc(ctx, S->keep, (size_t)(n_copies + 1) * 8, &p)); len = (int64_t *)p;
    ACHK(arena_alloc(ctx, S->keep, (size_t)n_cand * 4, &p)); mode_a = (int32_t *)p;
    ACHK(arena_alloc(ctx, S->keep, (size_t)n_cand * 4, &p)); mode_b = (int32_t *)p;
    ACHK(arena_alloc(ctx, S->keep, (size_t)n_cand * 4, &p)); src_pass = (int32_t *)p;
    ACHK(arena_alloc(ctx, S->keep, (size_t)n_cand * 8, &p)); keep_len = (int64_t *)p;
    ACHK(arena_alloc(ctx, S->keep, (size_t)(n_cand + 1) * 8, &p)); out_off = (int64_t *)p;
    if (n_copies > 0)
        hipLaunchKernelGGL(flank_sizes_kernel, dim3((unsigned)((n_copies + 255) / 256)), dim3(256), 0, st, ctx->d_contig_off,
                           ctx->n_contigs, n_copies, d_contig, d_start1, d_end1, flank, len, (int64_t *)nullptr);
    if (!d_clip && n_copies > 0) {
        // a table without clip words (the reference's own tuples; never inferred from where the caller's arrays live): the probe
        // supplies them, so that the rows are padded as the copy finder's own records are -- bare aligned windows, aligned globally
        // against a longer centre, cost a fifth of the calls (profiles/r05_scale_tests.txt)
        uint16_t *est;
        ACHK(arena_alloc(ctx, S->keep, (size_t)n_copies * 4 + 16, &p)); est = (uint16_t *)p;
        const int tkp = hite_prof_begin(ctx, "clip_probe_kernel", st);
        hipLaunchKernelGGL(clip_probe_kernel, dim3((unsigned)((2 * n_copies + 255) / 256)), dim3(256), 0, st, ctx->d_bases, ctx->d_nmask,
                           ctx->d_contig_off, ctx->n_contigs, n_copies, n_cand, d_copy_first, d_cand, d_cand_off, d_contig, d_start1, d_end1,
                           d_minus, est);
        hite_prof_end(ctx, tkp, st);
        d_clip = reinterpret_cast<const uint32_t *>(est);
    }
    hipLaunchKernelGGL(mode_a_kernel, dim3((n_cand + 255) / 256), dim3(256), 0, st, n_cand, d_copy_first, len, mode_a);
    HITE_CHECK(ctx, hipGetLastError());
    PassOut A, B;
    ACHK(run_pass(ctx, S, te_type, plant, n_cand, d_cand, d_cand_off, d_copy_first, d_contig, d_start1, d_end1, d_minus, d_clip, flank,
                  len, mode_a, &A, stats, stats + 8, stats + 11, false, st));
    ACHK(arena_reset(ctx, S->tmp, false));
    hipLaunchKernelGGL(mode_b_kernel, dim3((n_cand + 255) / 256), dim3(256), 0, st, n_cand, mode_a, A.calls, mode_b);
    ACHK(run_pass(ctx, S, te_type, plant, n_cand, d_cand, d_cand_off, d_copy_first, d_contig, d_start1, d_end1, d_minus, d_clip, flank,
                  len, mode_b, &B, stats + 4, stats + 9, stats + 11, true, st));
    hipLaunchKernelGGL(merge_calls_kernel, dim3((n_cand + 255) / 256), dim3(256), 0, st, n_cand, mode_a, A.calls, B.calls, d_calls,
                       src_pass, keep_len);
    ACHK(scan_excl<int64_t>(ctx, S->tmp, keep_len, n_cand, out_off, st));
    hipLaunchKernelGGL(copy_cons_kernel, dim3(n_cand), dim3(256), 0, st, n_cand, d_calls, src_pass, A.cons, B.cons, out_off, d_cons,
                       cons_cap);
    HITE_CHECK(ctx, hipGetLastError());
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal, out_off + n_cand, 8, hipMemcpyDeviceToDevice, st));
    ACHK(read_scalars(ctx, S, st, 1));
    if (stats_out) { memcpy(stats_out, stats, sizeof stats); stats_out[10] = S->h_pin[0]; }
    const int64_t cons_need = S->h_pin[0];
    // everything in the two arenas is dead now (the stream is drained): if they grew into several chunks, merge them so that
    // the next call performs no hipMalloc
    ACHK(arena_reset(ctx, S->keep, true));
    ACHK(arena_reset(ctx, S->tmp, true));
    if (cons_need > cons_cap) return HITE_ECAP;
    return HITE_OK;
}

// the probe by itself, host buffers: what hite_flank_region_align[_clip] with clip == NULL pads the rows by (parity tests, diagnostics)
extern "C" int hite_clip_probe(hite_ctx *ctx, int32_t n_cand, const uint8_t *cand, const int64_t *cand_off, const int32_t *copy_first,
                               int64_t n_copies, const int32_t *contig, const int64_t *start1, const int64_t *end1, const uint8_t *minus,
                               uint32_t *clip_out) {
    if (!ctx || !ctx->d_bases || n_cand < 0 || n_copies < 0 || (n_copies > 0 && (!cand || !cand_off || !copy_first || !contig || !start1 || !end1 || !minus || !clip_out)))
        return HITE_EINVAL;
    if (n_copies == 0 || n_cand == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    uint8_t *d = nullptr;
    const size_t cb = (size_t)cand_off[n_cand], o1 = (cb + 64 + 255) & ~(size_t)255, o2 = o1 + (((size_t)(n_cand + 1) * 8 + 255) & ~(size_t)255),
                 o3 = o2 + (((size_t)(n_cand + 1) * 4 + 255) & ~(size_t)255), o4 = o3 + (((size_t)n_copies * 4 + 255) & ~(size_t)255),
                 o5 = o4 + (((size_t)n_copies * 8 + 255) & ~(size_t)255), o6 = o5 + (((size_t)n_copies * 8 + 255) & ~(size_t)255),
                 o7 = o6 + (((size_t)n_copies + 255) & ~(size_t)255), total = o7 + (size_t)n_copies * 4 + 256;
    HITE_CHECK(ctx, hipMalloc((void **)&d, total));
    hipError_t e = hipMemcpy(d, cand, cb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o1, cand_off, (size_t)(n_cand + 1) * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o2, copy_first, (size_t)(n_cand + 1) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o3, contig, (size_t)n_copies * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o4, start1, (size_t)n_copies * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o5, end1, (size_t)n_copies * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o6, minus, (size_t)n_copies, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(clip_probe_kernel, dim3((unsigned)((2 * n_copies + 255) / 256)), dim3(256), 0, 0, ctx->d_bases, ctx->d_nmask,
                           ctx->d_contig_off, ctx->n_contigs, n_copies, n_cand, (const int32_t *)(d + o2), d, (const int64_t *)(d + o1),
                           (const int32_t *)(d + o3), (const int64_t *)(d + o4), (const int64_t *)(d + o5), d + o6, (uint16_t *)(d + o7));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(clip_out, d + o7, (size_t)n_copies * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HITE_CHECK(ctx, e);
    return HITE_OK;
}

extern "C" int hite_flank_region_align_dev(hite_ctx *ctx, void **state_io, int32_t te_type, int32_t plant, int32_t n_cand,
                                           const uint8_t *d_cand, const int64_t *d_cand_off, const int32_t *d_copy_first,
                                           int64_t n_copies, const int32_t *d_contig, const int64_t *d_start1,
                                           const int64_t *d_end1, const uint8_t *d_minus, int32_t flank, hite_call *d_calls,
                                           uint8_t *d_cons, int64_t cons_cap, int64_t *stats_out, void *stream) {
    return hite_flank_region_align_clip_dev(ctx, state_io, te_type, plant, n_cand, d_cand, d_cand_off, d_copy_first, n_copies, d_contig,
                                            d_start1, d_end1, d_minus, nullptr, flank, d_calls, d_cons, cons_cap, stats_out, stream);
}

// host-buffer wrapper (numpy callers / the drop-in scripts): uploads, runs, downloads
struct PBuf {
    void *p = nullptr;
    ~PBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
    hipError_t up(const void *h, size_t n) {
        hipError_t e = alloc(n + 16);
        if (e != hipSuccess) return e;
        return n ? hipMemcpy(p, h, n, hipMemcpyHostToDevice) : hipSuccess;
    }
};
__global__ void fold_bytes_kernel(uint8_t *__restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = fold_sym(p[i]);
}

extern "C" int hite_flank_region_align_clip(hite_ctx *ctx, int32_t te_type, int32_t plant, int32_t n_cand, const uint8_t *cand,
                                            const int64_t *cand_off, const int32_t *copy_first, int64_t n_copies,
                                            const int32_t *contig, const int64_t *start1, const int64_t *end1,
                                            const uint8_t *minus, const uint32_t *clip /* per copy, or NULL */, int32_t flank,
                                            hite_call *calls, uint8_t *cons, int64_t cons_cap, int64_t *stats_out) {
    if (!ctx || n_cand < 0 || !cand || !cand_off || !copy_first || !calls || !cons) return HITE_EINVAL;
    if (n_cand == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    PBuf dc, dco, dcf, dct, ds, de, dm, dcl, dcalls, dcons;
    hipError_t e;
    e = dc.up(cand, cand_off[n_cand]); if (e == hipSuccess) e = dco.up(cand_off, (n_cand + 1) * 8);
    if (e == hipSuccess) e = dcf.up(copy_first, (n_cand + 1) * 4); if (e == hipSuccess) e = dct.up(contig, n_copies * 4);
    if (e == hipSuccess) e = ds.up(start1, n_copies * 8); if (e == hipSuccess) e = de.up(end1, n_copies * 8);
    if (e == hipSuccess) e = dm.up(minus, n_copies); if (e == hipSuccess) e = dcalls.alloc(sizeof(hite_call) * n_cand);
    if (e == hipSuccess) e = dcons.alloc(cons_cap + 16);
    if (e == hipSuccess && clip) e = dcl.up(clip, n_copies * 4);
    HITE_CHECK(ctx, e);
    hipLaunchKernelGGL(fold_bytes_kernel, dim3(256), dim3(256), 0, nullptr, (uint8_t *)dc.p, cand_off[n_cand]);
    void *state = nullptr;
    int rc = hite_flank_region_align_clip_dev(ctx, &state, te_type, plant, n_cand, (uint8_t *)dc.p, (int64_t *)dco.p, (int32_t *)dcf.p,
                                              n_copies, (int32_t *)dct.p, (int64_t *)ds.p, (int64_t *)de.p, (uint8_t *)dm.p,
                                              clip ? (const uint32_t *)dcl.p : nullptr, flank, (hite_call *)dcalls.p, (uint8_t *)dcons.p,
                                              cons_cap, stats_out, nullptr);
    hipError_t es = hipDeviceSynchronize();
    if (rc == HITE_OK || rc == HITE_ECAP) {
        if (es == hipSuccess) es = hipMemcpy(calls, dcalls.p, sizeof(hite_call) * n_cand, hipMemcpyDeviceToHost);
        if (es == hipSuccess && rc == HITE_OK) es = hipMemcpy(cons, dcons.p, cons_cap, hipMemcpyDeviceToHost);
    }
    hite_pipeline_release(state);
    if (es != hipSuccess) return HITE_EHIP;
    return rc;
}

extern "C" int hite_flank_region_align(hite_ctx *ctx, int32_t te_type, int32_t plant, int32_t n_cand, const uint8_t *cand,
                                       const int64_t *cand_off, const int32_t *copy_first, int64_t n_copies,
                                       const int32_t *contig, const int64_t *start1, const int64_t *end1,
                                       const uint8_t *minus, int32_t flank, hite_call *calls, uint8_t *cons, int64_t cons_cap,
                                       int64_t *stats_out) {
    return hite_flank_region_align_clip(ctx, te_type, plant, n_cand, cand, cand_off, copy_first, n_copies, contig, start1, end1, minus,
                                        nullptr, flank, calls, cons, cons_cap, stats_out);
}
