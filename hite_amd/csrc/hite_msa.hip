// hite_msa.hip -- star alignment of the copy windows of each candidate: this build's GPU-native
// stage at the point where the reference shells out to `mafft` (Util.py:10416, third-party,
// unpinned, absent -> parity unpinned against mafft; the definition of the pairwise alignment is the
// textbook unit-cost global alignment of oracle/hite_oracle_nw.c, the twin of the whole stage is
// oracle/hite_oracle_msa.c, byte-exact).
//
//   * every row is aligned to the centre (row 0) by hite_align.hip (thread-per-pair bit-parallel
//     banded alignment with an optimality certificate); it emits per centre position the aligned row
//     position | gap flag << 15, 2 B per position; rows that cannot be aligned are dropped (row map);
//   * layout / fill kernels turn those ops into the rows x cols matrix -- in the pipeline fused with
//     remove_sparse_col_in_align_file, so that only the surviving columns are ever written.
#include "hite_common.h"
#include "hite_align.h"
#include "hite_scan.h"
#include "hite_fill.h"

#define MSA_MAXR 128   // rows whose lengths are cached in LDS by the layout / fill kernels

struct MsaParams {
    int n;                     // candidates
    int64_t total_rows;
    const uint8_t *win;
    const int64_t *win_off;    // total_rows : start of each window
    const int32_t *win_len;    // total_rows : length of each window
    const int32_t *row_first;  // n + 1
    const int64_t *ops_base;   // n + 1 : exclusive scan of (R_c + 1) * (m_c + 1)
    uint16_t *ops;
    int32_t *cols_out;         // n
    int32_t *status;           // n : 0 ok, 1 failed (alignment too wide)
    const int32_t *row_map;    // total_rows : source row of each row that was aligned (NULL: identity)
    const int32_t *rows_eff;   // n : rows that were aligned (NULL: all)
    const uint32_t *pads;      // total_rows : pad bytes in front | behind << 16 of every row (row_pad_scan_kernel)
    uint32_t *lay;             // sparse layout, one word per centre position (0..m) of candidate c at lay[ops_base[c] / 2 + p]:
                               // kept insertion columns | keep-centre << 15 | first output column << 16  (a candidate owns
                               // (R + 1)(m + 1) >= 2 (m + 1) ops, so the halves of the ops offsets never overlap)
};

// rows of candidate c after dropped rows; source row (relative to g0) of row r
__device__ __forceinline__ int msa_rows(const int32_t *__restrict__ rows_eff, const int32_t *__restrict__ row_first, int c) {
    return rows_eff ? rows_eff[c] : row_first[c + 1] - row_first[c];
}
__device__ __forceinline__ int msa_src(const int32_t *__restrict__ row_map, int64_t g0, int r) {
    return row_map ? row_map[g0 + r] - (int)g0 : r;
}

// insertions of row r before centre position p (p = 0..m), from two neighbouring ops entries.  The full (non-sparse) kernels keep their
// column widths in the centre's ops row, spare entry included: they take the row's first base as an argument instead
__device__ __forceinline__ int row_ins_s(const uint16_t *__restrict__ rop, int p, int m, int nrow, int start) {
    int prev_end = start;
    if (p > 0) { unsigned o = rop[p - 1]; prev_end = (int)(o & 0x7fff) + ((o >> 15) ? 0 : 1); }
    int q = p < m ? (int)(rop[p] & 0x7fff) : nrow;
    return q - prev_end;
}
// (rop[-1], the spare entry of the ops row above, holds the op "before position 0": a gap op whose q is the row's first base --
// 0x8000 for a row without pads, ops_pad_fix_kernel; nrow = the row's end, win_len or where its back pads begin)
__device__ __forceinline__ int row_ins(const uint16_t *__restrict__ rop, int p, int m, int nrow) {
    const unsigned o = rop[p - 1];
    const int prev_end = (int)(o & 0x7fff) + ((o >> 15) ? 0 : 1);
    int q = p < m ? (int)(rop[p] & 0x7fff) : nrow;
    return q - prev_end;
}

// column layout: one block per candidate.  insmax -> row-0 slot of ops, block starts -> slot R.
__global__ void __launch_bounds__(256) star_layout_kernel(MsaParams P) {
    __shared__ int s_scan[8];
    const int c = blockIdx.x;
    if (c >= P.n) return;
    const int64_t g0 = P.row_first[c];
    const int R = msa_rows(P.rows_eff, P.row_first, c);
    if (R <= 0) { if (threadIdx.x == 0) P.cols_out[c] = 0; return; }
    const int m = P.win_len[g0];
    uint16_t *ops = P.ops + P.ops_base[c];
    uint16_t *insmax = ops;                              // centre row slot (the centre has no ops of its own)
    uint16_t *bstart = ops + (int64_t)R * (m + 1);       // extra slot
    if (P.status[c]) { if (threadIdx.x == 0) P.cols_out[c] = 0; return; }
    int running = 0;
    for (int base = 0; base <= m; base += 256) {
        int p = base + threadIdx.x;
        int mx = 0;
        if (p <= m) for (int r = 1; r < R; r++) {
            const int sr = msa_src(P.row_map, g0, r);
            int v = row_ins_s(ops + (int64_t)r * (m + 1), p, m, P.win_len[g0 + sr], P.pads ? (int)(P.pads[g0 + sr] & 0xffffu) : 0);
            mx = v > mx ? v : mx;
        }
        int width = p <= m ? mx + (p < m ? 1 : 0) : 0;
        int tot;
        int pre = block_excl_scan(width, s_scan, &tot);
        if (p <= m) {
            int bs = running + pre;
            insmax[p] = (uint16_t)(mx > 65535 ? 65535 : mx);
            bstart[p] = (uint16_t)(bs > 65535 ? 65535 : bs);
        }
        running += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (running > 65535) { P.status[c] = 1; P.cols_out[c] = 0; }
        else P.cols_out[c] = running;
    }
}

struct FillParams {
    int n;
    const uint8_t *win;
    const int64_t *win_off;
    const int32_t *win_len;
    const int32_t *row_first;
    const int64_t *ops_base;
    const uint16_t *ops;
    const int32_t *cols;
    const int64_t *msa_off;
    uint8_t *msa;
    const int32_t *row_map;    // as in MsaParams
    const int32_t *rows_eff;
    const uint32_t *pads;
};

// fill: grid (candidate, row slice); item (r, p) owns insertion block p + centre column p of row r,
// so every output byte is written exactly once and no scan is needed.
__global__ void __launch_bounds__(256) star_fill_kernel(FillParams P) {
    const int c = blockIdx.x;
    if (c >= P.n) return;
    const int C = P.cols[c];
    if (C <= 0) return;
    const int64_t g0 = P.row_first[c];
    const int R = msa_rows(P.rows_eff, P.row_first, c);
    const int m = P.win_len[g0];
    const uint16_t *ops = P.ops + P.ops_base[c];
    const uint16_t *insmax = ops;
    const uint16_t *bstart = ops + (int64_t)R * (m + 1);
    uint8_t *out = P.msa + P.msa_off[c];
    for (int r = blockIdx.y; r < R; r += gridDim.y) {
        const uint8_t *b = P.win + P.win_off[g0 + msa_src(P.row_map, g0, r)];
        const int nrow = P.win_len[g0 + msa_src(P.row_map, g0, r)];
        const int start = P.pads ? (int)(P.pads[g0 + msa_src(P.row_map, g0, r)] & 0xffffu) : 0;
        uint8_t *row = out + (int64_t)r * C;
        const uint16_t *rop = ops + (int64_t)r * (m + 1);
        for (int p = threadIdx.x; p <= m; p += 256) {
            int ins, gap = 0, q;
            if (r == 0) { ins = 0; q = p; }
            else {
                ins = row_ins_s(rop, p, m, nrow, start);
                if (p < m) { unsigned o = rop[p]; q = (int)(o & 0x7fff); gap = (int)(o >> 15); } else q = nrow;
            }
            const int bs = bstart[p], im = insmax[p];
            const int rp = q - ins;  // first inserted base
            for (int k = 0; k < im; k++) row[bs + k] = k < ins ? b[rp + k] : (uint8_t)'-';
            if (p < m) row[bs + im] = gap ? (uint8_t)'-' : b[q];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fused column layout + remove_sparse_col_in_align_file (Util.py:10344-10405) for the fine-stage pipeline: the full
// alignment is never materialised.  Column c of the full alignment is kept iff c == 0 or c == C-1 or gaps(c) <= R/2:
//   * centre column p: gaps = rows whose op carries the gap flag;
//   * column j of the insertion block before p: a base in every row with ins_r > j, a gap elsewhere (centre included),
//     so the kept columns of a block are a PREFIX of length kw = the ceil(R/2)-th largest ins_r;
//   * the first / last column of the alignment are kept regardless (the last one may be a non-prefix column of block m).
// Per position p one layout word (MsaParams::lay): kw | keep_centre << 15 | first kept column of the block << 16.
// ---------------------------------------------------------------------------------------------
#define LAY_KW_LIST 256  // positions of one round (1024) whose insertion block keeps columns: beyond it their lanes walk the rows themselves
#ifndef LAY_ROWS
#define LAY_ROWS 4      // rows whose ops a thread of the layout kernel has in flight
#endif
__device__ __forceinline__ unsigned long long lay_ld8(const uint16_t *p) { unsigned long long v; __builtin_memcpy(&v, p, 8); return v; }

__global__ void __launch_bounds__(256) star_layout_sparse_kernel(MsaParams P, int32_t *__restrict__ new_cols,
                                                                 int32_t *__restrict__ last_extra) {
    __shared__ int s_scan[8];
    __shared__ int s_mxm;
    const int c = blockIdx.x;
    if (c >= P.n) return;
    const int64_t g0 = P.row_first[c];
    const int R = msa_rows(P.rows_eff, P.row_first, c);
    if (R <= 0 || P.status[c]) { if (threadIdx.x == 0) { P.cols_out[c] = 0; new_cols[c] = 0; last_extra[c] = -1; } return; }
    const int m = P.win_len[g0];
    uint16_t *ops = P.ops + P.ops_base[c];
    uint32_t *lay = P.lay + (P.ops_base[c] >> 1);
    const int h = (R + 1) >> 1;                          // fewest rows with a base for a column to survive
    __shared__ int s_wl[MSA_MAXR];
    __shared__ int s_nkw;
    __shared__ unsigned short s_kwlist[LAY_KW_LIST], s_kwres[4 * 256];
    for (int r = threadIdx.x; r < R && r < MSA_MAXR; r += 256) s_wl[r] = P.win_len[g0 + msa_src(P.row_map, g0, r)];
    if (threadIdx.x == 0) s_mxm = 0;
    __syncthreads();
    const int64_t rs = m + 1;                            // ops row stride
    {   // widest insertion after the last centre position: decides which column is the last one
        int mxm = 0;
        for (int r = 1 + threadIdx.x; r < R; r += 256) { int v = row_ins(ops + r * rs, m, m, P.win_len[g0 + msa_src(P.row_map, g0, r)]); mxm = v > mxm ? v : mxm; }
        if (mxm > 0) atomicMax(&s_mxm, mxm);
    }
    __syncthreads();
    const int mxm = s_mxm;
    int run_new = 0, run_full = 0, extra = -1;
    // a thread owns FOUR consecutive positions per round: one 8-byte load per row brings their four ops (plus the op before the
    // first), a fourth of the memory instructions of one position per thread; block scans over the threads' sums
    for (int base = 0; base <= m; base += 4 * 256) {
        const int p0 = base + 4 * (int)threadIdx.x;
        int mx[4] = {0, 0, 0, 0}, npos[4] = {0, 0, 0, 0}, gapc[4] = {0, 0, 0, 0};
        if (p0 <= m) {
            const uint16_t *col = ops + p0;
            int r = 1;
            for (; r + LAY_ROWS - 1 < R; r += LAY_ROWS) {      // LAY_ROWS rows in flight
                unsigned long long o4[LAY_ROWS];
                unsigned om[LAY_ROWS];
#pragma unroll
                for (int u = 0; u < LAY_ROWS; u++) { o4[u] = lay_ld8(col + (int64_t)(r + u) * rs); om[u] = col[(int64_t)(r + u) * rs - 1]; }   // (p0 == 0: the op "before position 0", row_ins)
#pragma unroll
                for (int u = 0; u < LAY_ROWS; u++) {
                    const int wl = r + u < MSA_MAXR ? s_wl[r + u] : P.win_len[g0 + msa_src(P.row_map, g0, r + u)];
                    unsigned op = om[u];
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        const int p = p0 + x;
                        const unsigned oc = p < m ? (unsigned)(o4[u] >> (16 * x)) & 0xffffu : 0u;
                        const int pe = (int)(op & 0x7fffu) + ((op >> 15) ? 0 : 1);
                        const int q = p < m ? (int)(oc & 0x7fffu) : wl;
                        const int v = p <= m ? q - pe : 0;
                        mx[x] = v > mx[x] ? v : mx[x]; npos[x] += v > 0; gapc[x] += (int)(oc >> 15);
                        op = oc;
                    }
                }
            }
            for (; r < R; r++) {
                const unsigned long long o4 = lay_ld8(col + (int64_t)r * rs);
                unsigned op = col[(int64_t)r * rs - 1];
                const int wl = r < MSA_MAXR ? s_wl[r] : P.win_len[g0 + msa_src(P.row_map, g0, r)];
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    const int p = p0 + x;
                    const unsigned oc = p < m ? (unsigned)(o4 >> (16 * x)) & 0xffffu : 0u;
                    const int pe = (int)(op & 0x7fffu) + ((op >> 15) ? 0 : 1);
                    const int q = p < m ? (int)(oc & 0x7fffu) : wl;
                    const int v = p <= m ? q - pe : 0;
                    mx[x] = v > mx[x] ? v : mx[x]; npos[x] += v > 0; gapc[x] += (int)(oc >> 15);
                    op = oc;
                }
            }
        }
        // Positions whose insertion block keeps columns (at least h rows insert there: where the centre lacks bases of the family,
        // ~0.5 % of them) need the h-th largest insertion length.  One lane used to walk all rows for it while the other 63 of
        // its wavefront waited -- and nearly every wavefront of a round holds such a position: 2.2 of the kernel's 4.6 ms per C3
        // step.  Round 6: the positions go to a list in LDS and a WAVEFRONT takes each, lanes = rows (the count of rows with a
        // longer insertion is a ballot), the owners read the result back.
        if (threadIdx.x == 0) s_nkw = 0;
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const int p = p0 + x;
            if (p <= m && npos[x] >= h) {
                const int slot = atomicAdd(&s_nkw, 1);
                if (slot < LAY_KW_LIST) s_kwlist[slot] = p - base;
            }
        }
        __syncthreads();
        {
            const int nk = s_nkw < LAY_KW_LIST ? s_nkw : LAY_KW_LIST;
            const int lane = threadIdx.x & 63;
            for (int i = threadIdx.x >> 6; i < nk; i += 4) {
                const int p = base + s_kwlist[i];
                int kw = 1;
                for (;;) {
                    int cnt = 0;
                    for (int r = 1 + lane; r - lane < R; r += 64) {            // (every lane makes every trip: the ballot below)
                        const bool in = r < R;
                        const int rr = in ? r : 1;
                        const int wl = rr < MSA_MAXR ? s_wl[rr] : P.win_len[g0 + msa_src(P.row_map, g0, rr)];
                        const bool longer = in && row_ins(ops + (int64_t)rr * (m + 1), p, m, wl) > kw;
                        cnt += __popcll(__ballot(longer));
                    }
                    if (cnt >= h) kw++; else break;
                }
                if (lane == 0) s_kwres[s_kwlist[i]] = (unsigned short)(kw > 0xffff ? 0xffff : kw);
            }
        }
        __syncthreads();
        int wnew[4], wfull[4], kws[4], sum_new = 0, sum_full = 0;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const int p = p0 + x;
            int kw = 0;
            if (p <= m && npos[x] >= h) {
                kw = 1;
                if (s_nkw <= LAY_KW_LIST) kw = s_kwres[p - base];
                else {
                    // more such positions in one round than the list holds (it never happened on C2 / C3): the lane walks the rows
                    for (;;) {
                        int cnt = 0;
                        for (int r = 1; r < R; r++) cnt += row_ins(ops + (int64_t)r * (m + 1), p, m, P.win_len[g0 + msa_src(P.row_map, g0, r)]) > kw;
                        if (cnt >= h) kw++; else break;
                    }
                }
            }
            int keepc = (p < m && 2 * gapc[x] <= R) ? 1 : 0;
            int ex = 0;
            if (p == 0) { if (mx[x] > 0) kw = kw > 1 ? kw : 1; else if (m > 0) keepc = 1; }     // first column
            if (p == m - 1 && mxm == 0) keepc = 1;                                            // last column = centre column m-1
            if (p == m && mx[x] > 0 && kw < mx[x]) { ex = 1; extra = mx[x] - 1; }                // last column = block m, j = mx-1
            if (kw > 0x7fff) kw = 0x7fff;
            wnew[x] = p <= m ? kw + keepc + ex : 0;
            wfull[x] = p <= m ? mx[x] + (p < m ? 1 : 0) : 0;
            kws[x] = kw | (keepc << 15);
            sum_new += wnew[x]; sum_full += wfull[x];
        }
        int tot_new, tot_full;
        const int pre = block_excl_scan(sum_new, s_scan, &tot_new);
        __syncthreads();
        (void)block_excl_scan(sum_full, s_scan, &tot_full);
        int bs = run_new + pre;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const int p = p0 + x;
            if (p <= m) lay[p] = (uint32_t)kws[x] | ((uint32_t)(bs > 65535 ? 65535 : bs) << 16);
            bs += wnew[x];
        }
        run_new += tot_new; run_full += tot_full;
        __syncthreads();
    }
    if ((int)threadIdx.x == ((m & 1023) >> 2)) last_extra[c] = extra;   // the thread that owned p == m
    if (threadIdx.x == 0) {
        if (run_full > 65535) { P.status[c] = 1; P.cols_out[c] = 0; new_cols[c] = 0; }
        else { P.cols_out[c] = run_full; new_cols[c] = run_new; }
    }
}

struct FillSparseParams {
    FillParams F;              // cols = kept columns per candidate, msa = compacted output
    const int32_t *last_extra;
    const uint32_t *lay;       // as in MsaParams
    const uint8_t *cls;        // per candidate JUDGE_CLS_* (may be NULL): the LDS classes are built by the judge itself
};

// fill of the kept columns only: item (r, p) owns the kept prefix of insertion block p, the centre column p if kept
// and (p == m) the extra last column; every output byte is written exactly once.
// the sink star_fill_sparse_kernel hands to fill_sparse_row: (row, position) into the workgroup's list; a full list fills on the spot
#define FILL_DEFER_CAP 1024
struct FillDefer {
    unsigned *list; int *count; unsigned row;
    __device__ __forceinline__ bool push(int p) const {
        const int slot = atomicAdd(count, 1);
        if (slot >= FILL_DEFER_CAP) return false;
        list[slot] = (row << 16) | (unsigned)p;
        return true;
    }
};
template <int NU>
__global__ void __launch_bounds__(256) star_fill_sparse_kernel(FillSparseParams Q) {
    const FillParams &P = Q.F;
    const int c = blockIdx.x;
    if (c >= P.n) return;
    const int C = P.cols[c];
    if (C <= 0 || (Q.cls && Q.cls[c] >= JUDGE_CLS_LDS)) return;
    const int64_t g0 = P.row_first[c];
    const int R = msa_rows(P.rows_eff, P.row_first, c);
    const int m = P.win_len[g0];
    const uint16_t *ops = P.ops + P.ops_base[c];
    const uint32_t *lay = Q.lay + (P.ops_base[c] >> 1);
    const int le = Q.last_extra[c];
    uint8_t *out = P.msa + P.msa_off[c];
    // rows of the candidate are spread over blockIdx.y, positions over the threads: FILL_U positions per thread and trip, the
    // loads of each dependency level issued together (the chain kwslot -> ops -> base is three loads deep and the kernel is
    // latency bound otherwise); per-row values (window, length, output row) are wave-uniform: no division, little address math
    const int rs = m + 1;
    int r = blockIdx.y;
    if (r >= R) return;
    __shared__ unsigned s_defer[FILL_DEFER_CAP];
    __shared__ int s_ndefer;
    if (threadIdx.x == 0) s_ndefer = 0;
    __syncthreads();
    const bool can_defer = R <= 0xffff && m < 0xffff;      // (row and position share a list word)
    // the window of the NEXT row of this block is looked up while the current one is filled: row_map -> win_off / win_len are
    // two dependent scalar round trips, as long as the two or three trips over the positions that a row of 2.5 kb takes
    int src = msa_src(P.row_map, g0, r);
    int64_t woff = P.win_off[g0 + src];
    int nrow = P.win_len[g0 + src];
    while (r < R) {
        const int rn = r + (int)gridDim.y;
        int64_t woff_n = 0; int nrow_n = 0;
        if (rn < R) { const int sn = msa_src(P.row_map, g0, rn); woff_n = P.win_off[g0 + sn]; nrow_n = P.win_len[g0 + sn]; }
        const uint8_t *b = P.win + woff;
        const uint16_t *rop = ops + (int64_t)r * rs;
        uint8_t *row = out + (int64_t)r * C;
        if (can_defer) fill_sparse_row<uint8_t *, 256, NU, FillDefer>(row, b, nrow, rop, lay, m, le, r == 0 /* the centre row: position p faces its own base p */, (int)threadIdx.x,
                                                                      FillDefer{s_defer, &s_ndefer, (unsigned)r});
        else fill_sparse_row<uint8_t *, 256, NU>(row, b, nrow, rop, lay, m, le, r == 0, (int)threadIdx.x);
        r = rn; woff = woff_n; nrow = nrow_n;
    }
    // the insertion blocks and extra last columns the rows above left behind: a thread each, their load chains side by side
    __syncthreads();
    const int nd = s_ndefer < FILL_DEFER_CAP ? s_ndefer : FILL_DEFER_CAP;
    for (int i = threadIdx.x; i < nd; i += 256) {
        const int rr = (int)(s_defer[i] >> 16), p = (int)(s_defer[i] & 0xffffu);
        const int sr = msa_src(P.row_map, g0, rr);
        const uint8_t *b = P.win + P.win_off[g0 + sr];
        const int nr = P.win_len[g0 + sr];
        const uint16_t *rop = ops + (int64_t)rr * rs;
        const unsigned w = lay[p];
        const unsigned oc = rr == 0 ? (unsigned)p : (p < m ? (unsigned)rop[p] : 0x8000u);
        fill_sparse_block<uint8_t *>(out + (int64_t)rr * C, b, nr, rop, m, le, rr == 0, p, w, oc);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// rows that could not be aligned leave the alignment: the ops rows of the candidate move up, row_map names the source
// window of every remaining row, rows_eff their number
__global__ void __launch_bounds__(256) msa_compact_rows_kernel(int n, const int32_t *__restrict__ row_first,
                                                               const int32_t *__restrict__ win_len, const int64_t *__restrict__ ops_base,
                                                               uint16_t *__restrict__ ops_all, const int32_t *__restrict__ cand_flag,
                                                               const int32_t *__restrict__ row_dead, int32_t *__restrict__ row_map,
                                                               int32_t *__restrict__ rows_eff) {
    const int c = blockIdx.x;
    if (c >= n) return;
    const int g0 = row_first[c], R = row_first[c + 1] - g0;
    if (R <= 0) { if (threadIdx.x == 0) rows_eff[c] = 0; return; }
    if (cand_flag[c] != 2) {
        for (int r = threadIdx.x; r < R; r += 256) row_map[g0 + r] = g0 + r;
        if (threadIdx.x == 0) rows_eff[c] = R;
        return;
    }
    const int m = win_len[g0];
    uint16_t *ops = ops_all + ops_base[c];
    int k = 0;
    for (int r = 0; r < R; r++) {
        if (row_dead[g0 + r]) continue;   // uniform over the block
        if (k != r) for (int p = threadIdx.x; p <= m; p += 256) ops[(int64_t)k * (m + 1) + p] = ops[(int64_t)r * (m + 1) + p];
        if (threadIdx.x == 0) row_map[g0 + k] = g0 + r;
        k++;
        __syncthreads();
    }
    if (threadIdx.x == 0) rows_eff[c] = k;
}
__global__ void msa_rows_out_kernel(int n, const int32_t *__restrict__ row_first, const int32_t *__restrict__ rows_eff,
                                    int32_t *__restrict__ rows_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) rows_out[c] = rows_eff ? rows_eff[c] : row_first[c + 1] - row_first[c];
}

// ---- rows that begin / end with pad bytes (HITE_IS_ROW_PAD, include/hite_gpu.h) ----------------------------------------------
// per row: the pads it begins and ends with; len2 = where its back pads begin (the row's end for layout and fill)
__global__ void row_pad_scan_kernel(int64_t total_rows, const uint8_t *__restrict__ win, const int64_t *__restrict__ win_off,
                                    const int32_t *__restrict__ win_len, int32_t *__restrict__ len2, uint32_t *__restrict__ pads) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total_rows) return;
    const uint8_t *b = win + win_off[g];
    const int n = win_len[g];
    int pf = 0, pb = 0;
    if (n > 0 && HITE_IS_ROW_PAD(b[0])) { pf = 1; while (pf < n && HITE_IS_ROW_PAD(b[pf])) pf++; }
    if (pf < n && HITE_IS_ROW_PAD(b[n - 1])) { pb = 1; while (pf + pb < n && HITE_IS_ROW_PAD(b[n - 1 - pb])) pb++; }
    len2[g] = n - pb;
    pads[g] = (uint32_t)pf | ((uint32_t)pb << 16);
}
// The pads leave the alignment WITHOUT the ops being renumbered: the row keeps its window (bases are still b[q]), its first base is
// b[pf] and its end is n - pb.  Monotone ops: the positions aligned to front pads (q < pf) are a prefix of the row, those aligned to
// back pads (q >= n - pb) a suffix -- a wavefront per row rewrites just these as gaps of the row (q = pf / n - pb), and writes the
// op "before position 0" (a gap op with q = pf) into the spare entry of the ops row above, where row_ins and the layout loops read
// it: a few entries per row instead of a pass over all ops (which was 4 ms per C3 step).  Runs after the compaction of the rows.
__global__ void __launch_bounds__(256) ops_pad_fix_kernel(int n, const int32_t *__restrict__ row_first, const int32_t *__restrict__ win_len,
                                                          const int64_t *__restrict__ ops_base, uint16_t *__restrict__ ops_all,
                                                          const uint32_t *__restrict__ pads, const int32_t *__restrict__ row_map,
                                                          const int32_t *__restrict__ rows_eff, int32_t *__restrict__ len2) {
    const int c = blockIdx.x;
    if (c >= n) return;
    const int g0 = row_first[c];
    const int R = rows_eff ? rows_eff[c] : row_first[c + 1] - g0;
    if (R >= 1 && threadIdx.x == 0) len2[g0] = win_len[g0];      // the centre is taken as it is
    if (R <= 1) return;
    const int m = win_len[g0];
    uint16_t *ops = ops_all + ops_base[c];
    const int lane = threadIdx.x & 63;
    for (int k = 1 + (int)(threadIdx.x >> 6); k < R; k += 4) {
        const int src = row_map ? row_map[g0 + k] : g0 + k;
        const uint32_t pd = pads[src];
        const int pf = (int)(pd & 0xffffu), pb = (int)(pd >> 16), hi = win_len[src] - pb;
        uint16_t *o = ops + (int64_t)k * (m + 1);
        if (lane == 0) o[-1] = (uint16_t)(0x8000u | (unsigned)pf);
        if (pf) {
            for (int base = 0; base < m; base += 64) {
                const int p = base + lane;
                const bool hit = p < m && (int)(o[p < m ? p : 0] & 0x7fffu) < pf;
                if (hit) o[p] = (uint16_t)(0x8000u | (unsigned)pf);
                if (__ballot(hit) != ~0ull) break;
            }
        }
        if (pb) {
            for (int top = m; top > 0; top -= 64) {
                const int p = top - 1 - lane;
                const bool hit = p >= 0 && (int)(o[p >= 0 ? p : 0] & 0x7fffu) >= hi;
                if (hit) o[p] = (uint16_t)(0x8000u | (unsigned)hi);
                if (__ballot(hit) != ~0ull) break;
            }
        }
    }
}

static int star_msa_launch(hite_ctx *ctx, int32_t n, const uint8_t *d_win, const int64_t *d_win_off,
                           const int32_t *d_win_len, const int32_t *d_row_first, int64_t total_rows, const int64_t *d_ops_base,
                           int64_t ops_elems, int32_t max_win_len, int32_t *d_cols_out, int32_t *d_status,
                           int32_t *d_new_cols, int32_t *d_last_extra, int32_t *d_rows_out, int32_t *d_info, void *stream) {
    if (!ctx || n < 0 || total_rows < 0 || max_win_len <= 0 || max_win_len > 32767) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    // scratch2: ops | row_dead | row_map | rows_eff | cand_flag
    const size_t ops_bytes = ((size_t)ops_elems * 2 + 255) & ~(size_t)255;
    const size_t rows_bytes = ((size_t)total_rows * 4 + 255) & ~(size_t)255, cand_bytes = ((size_t)n * 4 + 255) & ~(size_t)255;
    void *opsb = nullptr;
    const size_t lay_bytes = d_new_cols ? (((size_t)ops_elems / 2 + 32768 + 16) * 4 + 255) & ~(size_t)255 : 0;
    int rc = hite_scratch2_reserve(ctx, ops_bytes + 2 * rows_bytes + 2 * cand_bytes + lay_bytes + 2 * rows_bytes + 256, &opsb);
    if (rc) return rc;
    uint8_t *base = (uint8_t *)opsb;
    int32_t *row_dead = (int32_t *)(base + ops_bytes), *row_map = (int32_t *)(base + ops_bytes + rows_bytes);
    int32_t *rows_eff = (int32_t *)(base + ops_bytes + 2 * rows_bytes), *cand_flag = (int32_t *)(base + ops_bytes + 2 * rows_bytes + cand_bytes);
    hipStream_t st = (hipStream_t)stream;
    HITE_CHECK(ctx, hipMemsetAsync(d_status, 0, (size_t)n * 4, st));
    HITE_CHECK(ctx, hipMemsetAsync(cand_flag, 0, (size_t)n * 4, st));
    int64_t before[8], after[8];
    rc = hite_align_stats(ctx, before, 0);
    if (rc) return rc;
    rc = hite_align_run(ctx, n, d_win, d_win_off, d_win_len, d_row_first, total_rows, d_ops_base, (uint16_t *)opsb, row_dead, cand_flag, d_info,
                        max_win_len > 1000 ? "_long" : "_short", st);
    if (rc) return rc;
    rc = hite_align_stats(ctx, after, 0);
    if (rc) return rc;
    const bool dropped = after[4] > before[4];
    // rows padded with pad bytes (HITE_IS_ROW_PAD): where each begins and ends (the ops are fixed up after the compaction below)
    int32_t *len2 = (int32_t *)(base + ops_bytes + 2 * rows_bytes + 2 * cand_bytes + lay_bytes);
    uint32_t *pads = (uint32_t *)((uint8_t *)len2 + rows_bytes);
    if (total_rows > 0)
        hipLaunchKernelGGL(row_pad_scan_kernel, dim3((unsigned)((total_rows + 255) / 256)), dim3(256), 0, st, total_rows, d_win, d_win_off, d_win_len,
                           len2, pads);
    ctx->d_msa_win_off = d_win_off; ctx->d_msa_win_len = len2; ctx->d_msa_pads = pads; ctx->msa_long = max_win_len > 1536;
    MsaParams P;
    P.n = n; P.total_rows = total_rows; P.win = d_win; P.win_off = d_win_off; P.win_len = len2; P.row_first = d_row_first;
    P.ops_base = d_ops_base; P.ops = (uint16_t *)opsb; P.cols_out = d_cols_out; P.status = d_status;
    P.row_map = nullptr; P.rows_eff = nullptr; P.pads = pads;
    P.lay = d_new_cols ? (uint32_t *)(base + ops_bytes + 2 * rows_bytes + 2 * cand_bytes) : nullptr;
    ctx->d_msa_lay = P.lay;
    if (dropped) {
        hipLaunchKernelGGL(msa_compact_rows_kernel, dim3(n), dim3(256), 0, st, n, d_row_first, d_win_len, d_ops_base, (uint16_t *)opsb, cand_flag,
                           row_dead, row_map, rows_eff);
        P.row_map = row_map; P.rows_eff = rows_eff;
    }
    ctx->d_msa_row_map = P.row_map; ctx->d_msa_rows_eff = P.rows_eff;
    hipLaunchKernelGGL(ops_pad_fix_kernel, dim3(n), dim3(256), 0, st, n, d_row_first, d_win_len, d_ops_base, (uint16_t *)opsb, pads, P.row_map,
                       P.rows_eff, len2);
    if (d_rows_out) hipLaunchKernelGGL(msa_rows_out_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, d_row_first, P.rows_eff, d_rows_out);
    int tk;
    if (d_new_cols) {
        tk = hite_prof_begin(ctx, max_win_len > 1000 ? "star_layout_sparse_kernel_long" : "star_layout_sparse_kernel_short", st);
        hipLaunchKernelGGL(star_layout_sparse_kernel, dim3(n), dim3(256), 0, st, P, d_new_cols, d_last_extra);
    } else {
        tk = hite_prof_begin(ctx, "star_layout_kernel", st);
        hipLaunchKernelGGL(star_layout_kernel, dim3(n), dim3(256), 0, st, P);
    }
    hite_prof_end(ctx, tk, st);
    HITE_CHECK(ctx, hipGetLastError());
    return HITE_OK;
}

extern "C" int hite_star_msa_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_win, const int64_t *d_win_off,
                                 const int32_t *d_win_len, const int32_t *d_row_first, int64_t total_rows, const int64_t *d_ops_base,
                                 int64_t ops_elems, int32_t max_win_len, int32_t *d_cols_out, int32_t *d_status,
                                 int32_t *d_rows_out, void *stream) {
    return star_msa_launch(ctx, n, d_win, d_win_off, d_win_len, d_row_first, total_rows, d_ops_base, ops_elems, max_win_len,
                           d_cols_out, d_status, nullptr, nullptr, d_rows_out, nullptr, stream);
}

// align + fused layout / sparse-column selection: d_cols_out = columns of the full alignment (0 = failed),
// d_new_cols = columns that survive remove_sparse_col_in_align_file, d_last_extra = per-candidate fill hint,
// d_rows_out (may be NULL) = rows of each alignment (rows that cannot be aligned are dropped)
extern "C" int hite_star_msa_sparse_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_win, const int64_t *d_win_off,
                                        const int32_t *d_win_len, const int32_t *d_row_first, int64_t total_rows,
                                        const int64_t *d_ops_base, int64_t ops_elems, int32_t max_win_len, int32_t *d_cols_out,
                                        int32_t *d_status, int32_t *d_new_cols, int32_t *d_last_extra, int32_t *d_rows_out,
                                        void *stream) {
    if (!d_new_cols || !d_last_extra) return HITE_EINVAL;
    return star_msa_launch(ctx, n, d_win, d_win_off, d_win_len, d_row_first, total_rows, d_ops_base, ops_elems, max_win_len,
                           d_cols_out, d_status, d_new_cols, d_last_extra, d_rows_out, nullptr, stream);
}

// compacted alignments (rows x d_new_cols[i] at d_msa_off[i]) from the ops of the last hite_star_msa_sparse_dev call
extern "C" int hite_star_msa_fill_sparse_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_win, const int64_t *d_win_off,
                                             const int32_t *d_win_len, const int32_t *d_row_first, const int64_t *d_ops_base,
                                             const int32_t *d_new_cols, const int32_t *d_last_extra, const int64_t *d_msa_off,
                                             uint8_t *d_msa, void *stream) {
    if (!ctx || n < 0 || !ctx->d_scratch2 || !d_new_cols || !d_last_extra) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    FillSparseParams Q;
    FillParams &P = Q.F;
    // (rows through the star stage's view of them: without the HITE_ROW_PAD runs they were aligned with)
    P.n = n; P.win = d_win; P.win_off = ctx->d_msa_win_off ? ctx->d_msa_win_off : d_win_off; P.win_len = ctx->d_msa_win_len ? ctx->d_msa_win_len : d_win_len;
    P.row_first = d_row_first; P.ops_base = d_ops_base;
    P.ops = (const uint16_t *)ctx->d_scratch2; P.cols = d_new_cols; P.msa_off = d_msa_off; P.msa = d_msa;
    P.row_map = ctx->d_msa_row_map; P.rows_eff = ctx->d_msa_rows_eff; P.pads = ctx->d_msa_pads;
    Q.last_extra = d_last_extra; Q.lay = ctx->d_msa_lay;
    Q.cls = ctx->judge_fuse.win ? ctx->d_judge_cls : nullptr;   // set by the pipeline around this call only
    if (!Q.lay) return HITE_EINVAL;
    // Round 3 measured three other forms of this kernel on C3 (this one: 1.9 + 6.0 ms per step for the two passes):
    //   * positions-outer (a thread owns four centre positions, keeps their layout words and walks the rows): 3.9 + 13.1 ms;
    //   * four consecutive positions per thread, 16 / 8 / 8-byte loads and one 4-byte store, one, two or four rows per
    //     trip, no conditional load on the main path: 2.2 + 6.6, 2.4 + 7.3, 2.1 + 6.2 ms -- a quarter of the memory
    //     instructions, no gain;
    //   * this kernel with two rows per trip (twice the bytes in flight per wavefront): 3.5 + 11.7 ms.
    // Neither instructions nor latency bound it: the two launches move 19 GB (FETCH_SIZE + WRITE_SIZE as counted; 34 GB
    // with the guide's gfx950 fetch correction) in 8 ms, 2.3 - 4 TB/s; 2 B of ops per cell are most of it.
    // (round 6: eight cells in flight per thread when the launch holds long windows, four otherwise -- hite_fill.h)
#ifndef FILL_U_LONG
#define FILL_U_LONG 8
#endif
#ifndef FILL_U_SHORT
#define FILL_U_SHORT 4
#endif
    if (ctx->msa_long) hipLaunchKernelGGL(HIP_KERNEL_NAME(star_fill_sparse_kernel<FILL_U_LONG>), dim3(n, 4), dim3(256), 0, (hipStream_t)stream, Q);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(star_fill_sparse_kernel<FILL_U_SHORT>), dim3(n, 4), dim3(256), 0, (hipStream_t)stream, Q);
    HITE_CHECK(ctx, hipGetLastError());
    return HITE_OK;
}

extern "C" int hite_star_msa_fill_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_win, const int64_t *d_win_off,
                                      const int32_t *d_win_len, const int32_t *d_row_first, const int64_t *d_ops_base, const int32_t *d_cols,
                                      const int64_t *d_msa_off, uint8_t *d_msa, void *stream) {
    if (!ctx || n < 0 || !ctx->d_scratch2) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    FillParams P;
    P.n = n; P.win = d_win; P.win_off = ctx->d_msa_win_off ? ctx->d_msa_win_off : d_win_off; P.win_len = ctx->d_msa_win_len ? ctx->d_msa_win_len : d_win_len;
    P.row_first = d_row_first; P.ops_base = d_ops_base;
    P.ops = (const uint16_t *)ctx->d_scratch2; P.cols = d_cols; P.msa_off = d_msa_off; P.msa = d_msa;
    P.row_map = ctx->d_msa_row_map; P.rows_eff = ctx->d_msa_rows_eff; P.pads = ctx->d_msa_pads;
    hipLaunchKernelGGL(star_fill_kernel, dim3(n, 4), dim3(256), 0, (hipStream_t)stream, P);
    HITE_CHECK(ctx, hipGetLastError());
    return HITE_OK;
}

struct MBuf {
    void *p = nullptr;
    ~MBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
    hipError_t up(const void *h, size_t n) {
        hipError_t e = alloc(n);
        if (e != hipSuccess) return e;
        return n ? hipMemcpy(p, h, n, hipMemcpyHostToDevice) : hipSuccess;
    }
};

// host-buffer convenience: pass 1 (msa_out == NULL) returns cols_out / rows_out only; otherwise also the
// alignments at msa_off_out[i] (16-byte aligned slots), msa_cap bytes available.  info_out (may be NULL): 5 int32 per
// input row (U, certified, status, k*, band words | 0x100), zeros for the centres.
static int star_msa_host(hite_ctx *ctx, int32_t n, const uint8_t *win, const int64_t *win_off,
                         const int32_t *row_first, int32_t *cols_out, int32_t *rows_out, int64_t msa_cap, uint8_t *msa_out,
                         int64_t *msa_off_out, int32_t *info_out, bool sparse, uint8_t **msa_alloc = nullptr, int64_t *msa_bytes = nullptr) {
    if (!ctx || n < 0 || !win || !win_off || !row_first || !cols_out) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    int64_t total_rows = row_first[n];
    int64_t *ops_base = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    if (!ops_base) return HITE_ENOMEM;
    int64_t acc = 0;
    int maxlen = 1;
    for (int c = 0; c < n; c++) {
        int R = row_first[c + 1] - row_first[c];
        if (R <= 0) { free(ops_base); return HITE_EINVAL; }
        int64_t m = win_off[row_first[c] + 1] - win_off[row_first[c]];
        ops_base[c] = acc;
        acc += (int64_t)(R + 1) * (m + 1);
        for (int r = 0; r < R; r++) {
            int64_t L = win_off[row_first[c] + r + 1] - win_off[row_first[c] + r];
            if (L <= 0 || L > 32767) { free(ops_base); return HITE_EINVAL; }
            if (L > maxlen) maxlen = (int)L;
        }
    }
    ops_base[n] = acc;
    int32_t *wl = (int32_t *)malloc(sizeof(int32_t) * (total_rows + 1));
    int64_t *wo = (int64_t *)malloc(sizeof(int64_t) * (total_rows + 1));
    if (!wl || !wo) { free(ops_base); free(wl); free(wo); return HITE_ENOMEM; }
    // the device side reads the row bases 16 at a time: every window gets a 16-byte aligned, padded slot
    int64_t pos = 0;
    for (int64_t g = 0; g < total_rows; g++) { wl[g] = (int32_t)(win_off[g + 1] - win_off[g]); wo[g] = pos; pos += ((int64_t)wl[g] + 15) & ~(int64_t)15; }
    wo[total_rows] = pos;
    uint8_t *wpad = (uint8_t *)calloc((size_t)pos + 16, 1);
    if (!wpad) { free(ops_base); free(wl); free(wo); return HITE_ENOMEM; }
    for (int64_t g = 0; g < total_rows; g++) memcpy(wpad + wo[g], win + win_off[g], (size_t)wl[g]);
    MBuf dw, dwo, dwl, drf, dob, dcols, dst, dmo, dmsa, dnew, dlast, drows, dinfo;
    hipError_t e;
    e = dw.up(wpad, pos + 16); if (e == hipSuccess) e = dwo.up(wo, (total_rows + 1) * 8);
    if (e == hipSuccess) e = dwl.up(wl, total_rows * 4);
    free(wl); free(wo); free(wpad);
    if (e == hipSuccess) e = drf.up(row_first, (n + 1) * 4); if (e == hipSuccess) e = dob.up(ops_base, (n + 1) * 8);
    if (e == hipSuccess) e = dcols.alloc(n * 4); if (e == hipSuccess) e = dst.alloc(n * 4);
    if (e == hipSuccess) e = dnew.alloc(n * 4); if (e == hipSuccess) e = dlast.alloc(n * 4);
    if (e == hipSuccess) e = drows.alloc(n * 4);
    if (e == hipSuccess && info_out) e = dinfo.alloc((size_t)total_rows * 20);
    free(ops_base);
    HITE_CHECK(ctx, e);
    int rc = star_msa_launch(ctx, n, (uint8_t *)dw.p, (int64_t *)dwo.p, (int32_t *)dwl.p, (int32_t *)drf.p, total_rows, (int64_t *)dob.p, acc,
                             maxlen, (int32_t *)dcols.p, (int32_t *)dst.p, sparse ? (int32_t *)dnew.p : nullptr,
                             sparse ? (int32_t *)dlast.p : nullptr, (int32_t *)drows.p, info_out ? (int32_t *)dinfo.p : nullptr, nullptr);
    if (rc) return rc;
    HITE_CHECK(ctx, hipDeviceSynchronize());
    HITE_CHECK(ctx, hipMemcpy(cols_out, sparse ? dnew.p : dcols.p, n * 4, hipMemcpyDeviceToHost));
    int32_t *rows_h = (int32_t *)malloc(sizeof(int32_t) * n);
    if (!rows_h) return HITE_ENOMEM;
    if (hipMemcpy(rows_h, drows.p, n * 4, hipMemcpyDeviceToHost) != hipSuccess) { free(rows_h); return HITE_EHIP; }
    if (rows_out) memcpy(rows_out, rows_h, sizeof(int32_t) * n);
    if (info_out && hipMemcpy(info_out, dinfo.p, (size_t)total_rows * 20, hipMemcpyDeviceToHost) != hipSuccess) { free(rows_h); return HITE_EHIP; }
    if (!msa_out && !msa_alloc) { free(rows_h); return HITE_OK; }
    if (!msa_off_out) { free(rows_h); return HITE_EINVAL; }
    int64_t off = 0;
    for (int c = 0; c < n; c++) {
        msa_off_out[c] = off;
        off += ((int64_t)rows_h[c] * cols_out[c] + 15) / 16 * 16;
    }
    free(rows_h);
    if (msa_alloc) {       // one-call form: the alignments come back in a buffer of exactly the size they need (hite_host_free)
        msa_out = (uint8_t *)malloc((size_t)off + 16);
        if (!msa_out) return HITE_ENOMEM;
        *msa_alloc = msa_out;
        if (msa_bytes) *msa_bytes = off;
    } else if (off > msa_cap) return HITE_ECAP;
    e = dmo.up(msa_off_out, n * 8); if (e == hipSuccess) e = dmsa.alloc(off + 16);
    HITE_CHECK(ctx, e);
    if (sparse)
        rc = hite_star_msa_fill_sparse_dev(ctx, n, (uint8_t *)dw.p, (int64_t *)dwo.p, (int32_t *)dwl.p, (int32_t *)drf.p, (int64_t *)dob.p,
                                           (int32_t *)dnew.p, (int32_t *)dlast.p, (int64_t *)dmo.p, (uint8_t *)dmsa.p, nullptr);
    else
        rc = hite_star_msa_fill_dev(ctx, n, (uint8_t *)dw.p, (int64_t *)dwo.p, (int32_t *)dwl.p, (int32_t *)drf.p, (int64_t *)dob.p,
                                    (int32_t *)dcols.p, (int64_t *)dmo.p, (uint8_t *)dmsa.p, nullptr);
    if (rc) return rc;
    HITE_CHECK(ctx, hipDeviceSynchronize());
    HITE_CHECK(ctx, hipMemcpy(msa_out, dmsa.p, off, hipMemcpyDeviceToHost));
    return HITE_OK;
}

extern "C" int hite_star_msa(hite_ctx *ctx, int32_t n, const uint8_t *win, const int64_t *win_off,
                             const int32_t *row_first, int32_t *cols_out, int32_t *rows_out, int64_t msa_cap, uint8_t *msa_out,
                             int64_t *msa_off_out) {
    return star_msa_host(ctx, n, win, win_off, row_first, cols_out, rows_out, msa_cap, msa_out, msa_off_out, nullptr, false);
}
// same call protocol; the alignments come back with the sparse columns already removed (cols_out = surviving columns)
extern "C" int hite_star_msa_sparse(hite_ctx *ctx, int32_t n, const uint8_t *win, const int64_t *win_off,
                                    const int32_t *row_first, int32_t *cols_out, int32_t *rows_out, int64_t msa_cap, uint8_t *msa_out,
                                    int64_t *msa_off_out) {
    return star_msa_host(ctx, n, win, win_off, row_first, cols_out, rows_out, msa_cap, msa_out, msa_off_out, nullptr, true);
}
// One call instead of the sizes call + the fill call (each of which runs the whole pairwise alignment): the alignments are
// returned in a host buffer this function allocates (*msa_out, *msa_bytes_out bytes; release it with hite_host_free);
// sparse != 0: sparse columns already removed; info_out (may be NULL) as hite_star_msa_info.
extern "C" int hite_star_msa_once(hite_ctx *ctx, int32_t n, const uint8_t *win, const int64_t *win_off, const int32_t *row_first,
                                  int32_t sparse, int32_t *cols_out, int32_t *rows_out, int32_t *info_out, uint8_t **msa_out,
                                  int64_t *msa_off_out, int64_t *msa_bytes_out) {
    if (!msa_out || !msa_off_out) return HITE_EINVAL;
    *msa_out = nullptr;
    int rc = star_msa_host(ctx, n, win, win_off, row_first, cols_out, rows_out, 0, nullptr, msa_off_out, info_out, sparse != 0, msa_out, msa_bytes_out);
    if (rc && *msa_out) { free(*msa_out); *msa_out = nullptr; }
    return rc;
}
extern "C" void hite_host_free(void *p) { free(p); }
// pairwise view of the same stage (tests, diagnostics): group c = (centre, row_1, ..., row_k); info_out = 5 int32 per input
// window (zeros for the centres): cost U of the alignment kept, certified (0/1), status (0 aligned, 1 / 2 dropped), the
// certificate's bound k*, band words of the run kept | 0x100 for the wide fall-back.  With msa_out the alignments too.
extern "C" int hite_star_msa_info(hite_ctx *ctx, int32_t n, const uint8_t *win, const int64_t *win_off,
                                  const int32_t *row_first, int32_t *cols_out, int32_t *rows_out, int32_t *info_out, int64_t msa_cap,
                                  uint8_t *msa_out, int64_t *msa_off_out) {
    if (!info_out) return HITE_EINVAL;
    return star_msa_host(ctx, n, win, win_off, row_first, cols_out, rows_out, msa_cap, msa_out, msa_off_out, info_out, false);
}
