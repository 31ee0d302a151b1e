// hite_judge.hip -- per-column homology voting, sliding-window boundary search and the
// judge_boundary_v5/v6/v9 decisions, one 256-thread workgroup per candidate alignment.
//
// Reference (paths relative to /root/reference/module/Util.py):
//   remove_sparse_col_in_align_file :10344   col_base_map :9251   calculate_window_homology :8827
//   search_boundary_homo_v3 :8887            search_boundary_homo_v4 :8556
//   judge_boundary_v5 :9145  judge_boundary_v6 :9821  judge_boundary_v9 :9483  TSDsearch_v5 :2460
//
// GPU mapping
//   * alignment rows are lanes: the per-column vote over the <=101 selected rows is two
//     64-bit __ballot masks per symbol + __popcll (wave64), restricted by the window's
//     valid-row mask -- no LDS histogram, no atomics.
//   * candidate sliding windows are evaluated one per wavefront, four at a time, in the
//     reference's order, so the first homologous window is still the one returned.
//   * anchor search (find_near_matches, k<=2) is a 5-diagonal banded edit-distance per text
//     start, one start per thread, followed by LDS atomic min/max reductions that reproduce the
//     overlap-group rule with a local (m+k)-wide look-back.
//   * all FP compares are binary64 in the reference's operation order (thr-0.1, sequential sums).
// Algorithmic bytes per candidate: rows*cols read (1 B/cell) + 12 B/col of column statistics
// + cols bytes of consensus written.
#include "hite_common.h"
#include "hite_fill.h"

#ifndef JB
#define JB 256          // threads per block (every kernel of this file; 128 measured: short alignments 20 % faster, long ones 23 % slower)
#endif
#define MAXSEL 128      // capacity of the selected-row list (the reference keeps <= 101)
#define CS 12           // bytes of column statistics per column: cnt[6], first[6]
#define SCR_PER_COL 32  // scratch bytes per column per block slot
#define FNM_LIST 320    // 64 flagged ends x (2 k + 1 = 5) starts
#define TILE_COLS 160   // widest column span staged in LDS for the window scans (wider spans read the alignment directly): the workgroup form
// the wavefront form: a workgroup (= one wavefront) holds 3.7 KB + its tile, and 16 of them fit a CU's 160 KB only when the tile has
// <= 6.5 KB.  Measured in round 6 (-DJWAV_TILE_COLS, profiles/r06_judge_split.txt): 112 / 128 / 160 columns (9.4 / 9.9 / 11.4 KB per
// wavefront; the anchor text + match records of <= 1784 / 2040 / 2544 columns fit the same bytes) give 12.1 / 11.8 / 11.9 ms of judges
// per C3 step: the fourth wavefront per SIMD buys nothing, 160 stays
#ifndef JWAV_TILE_COLS
#define JWAV_TILE_COLS 160
#endif
#define JWAV_ANCHOR_COLS ((JWAV_TILE_COLS * 6 * 2 * 4 - 16) / 3 / 8 * 8)

#ifdef JUDGE_CLOCKS
// development aid (-DJUDGE_CLOCKS): wall-clock ticks per phase, summed over blocks by thread 0
__device__ unsigned long long g_jclk[16];
#define JCLK(i) do { if (threadIdx.x == 0) { unsigned long long now_ = wall_clock64(); atomicAdd(&g_jclk[i], now_ - S.jt); S.jt = now_; } } while (0)
extern "C" int hite_debug_judge_clocks(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_jclk), sizeof(g_jclk)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_jclk), z, sizeof(z)); }
    return 0;
}
#else
#define JCLK(i) do { } while (0)
#endif

struct JudgeParams {
    int te_type, plant, n;
    const uint8_t *msa;
    const int64_t *msa_off;
    const int32_t *rows, *cols;
    const uint8_t *cand;
    const int64_t *cand_off;
    const int64_t *col_off;  // exclusive scan of cols
    hite_call *calls;
    uint8_t *cons;
    uint8_t *scratch;     // per block slot
    size_t slot_bytes;    // 23 * maxC16 + 16 * maxR + 64
    size_t maxC16;        // max cols rounded up to 16
    unsigned int *counter;   // work queue: next entry of list
    const int32_t *list;     // alignments this kernel judges
    const unsigned int *n_list;
    JudgeFuse fuse;          // LDS kernels: win != NULL -> the alignment is built from these, not copied from msa
    int32_t *anchors;        // phase-split kernels (JT_PHASE 1 writes, 2 reads): start / end anchor column per alignment
};
typedef const uint8_t *jt_gptr;
typedef const __attribute__((address_space(3))) uint8_t *jt_lptr_c;
typedef __attribute__((address_space(3))) uint8_t *jt_lptr;
typedef unsigned int jt_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) jt_u32x4 *jt_lptr16;

// ---- the team code, four times: {workgroup, wavefront} x {alignment in HBM, alignment built in LDS} --------------------------
#define JW (JB / 64)
#define JT_W32 4
#define JT_TILE_COLS TILE_COLS
#define ANCHOR_LDS_COLS 5104   // ungapped row (<= this many bytes) + 2 bytes of match record per text start fit the 15 KB mask area
#define JT_KERNEL judge_kernel
#define JT_MSA_PTR jt_gptr
// minimum waves per SIMD asked of the compiler (it spills to get there).  Measured on C3, workgroup / wave kernel: 5/4 11.3 ms,
// 6/4 11.5, 4/4 12.2, 5/3 12.4, 4/3 12.7, 3/3 (no spills) 13.6: waves in flight beat spills, the judges wait on dependent loads
#ifndef JBLK_WAVES
#define JBLK_WAVES 5
#endif
#define JT_WAVES_MIN JBLK_WAVES
#define JT_PHASE 0
namespace jblk {
#include "hite_judge_team.inc"
}
#undef JT_KERNEL
#undef JT_WAVES_MIN
#undef JT_PHASE
// the same judge in two kernels per class (round 6, HITE_JUDGE_SPLIT; off by default, see hite_judge_dev): the anchors (ungapped
// rows + two fuzzy searches: 27 % of the judges' time, little state) and everything after them.  One body keeps every phase's
// state live and is compiled to spills at the occupancy asked for; the anchor half fits its registers (64 / 86, no scratch in the
// wavefront form), the rest keeps 176 / 64 bytes of scratch.
#ifndef JBLK_A_WAVES
#define JBLK_A_WAVES 8
#endif
#ifndef JBLK_B_WAVES
#define JBLK_B_WAVES 5
#endif
#define JT_KERNEL judge_anchor_kernel
#define JT_WAVES_MIN JBLK_A_WAVES
#define JT_PHASE 1
namespace jblkA {
#include "hite_judge_team.inc"
}
#undef JT_KERNEL
#undef JT_WAVES_MIN
#undef JT_PHASE
#define JT_KERNEL judge_rest_kernel
#define JT_WAVES_MIN JBLK_B_WAVES
#define JT_PHASE 2
namespace jblkB {
#include "hite_judge_team.inc"
}
#undef JT_KERNEL
#undef JT_MSA_PTR
#undef JT_WAVES_MIN
#undef JT_PHASE
#define JT_PHASE 0
#define JT_KERNEL judge_lds_kernel
#define JT_MSA_PTR jt_lptr_c
#define JT_LDS_PTR jt_lptr
#define JT_LDS_PTR16 jt_lptr16
#define JT_LDS_MSA 1
#ifndef JBLK_LDS_WAVES
#define JBLK_LDS_WAVES 2
#endif
#define JT_WAVES_MIN JBLK_LDS_WAVES
namespace jblds {
#include "hite_judge_team.inc"
}
#undef JB
#undef JW
#undef JT_W32
#undef ANCHOR_LDS_COLS
#undef JT_KERNEL
#undef JT_WAVES_MIN
#undef JT_MSA_PTR
#undef JT_LDS_MSA
#define JB 64
#define JW 1
#define JT_W32 2
#undef JT_TILE_COLS
#define JT_TILE_COLS JWAV_TILE_COLS
#define ANCHOR_LDS_COLS JWAV_ANCHOR_COLS   // 3 x columns + 16 <= the bytes of the mask tile
#define JT_KERNEL judge_wave_kernel
#define JT_MSA_PTR jt_gptr
#ifndef JWAV_WAVES
#define JWAV_WAVES 4
#endif
#define JT_WAVES_MIN JWAV_WAVES
namespace jwav {
#include "hite_judge_team.inc"
}
#undef JT_KERNEL
#undef JT_WAVES_MIN
#undef JT_PHASE
#ifndef JWAV_A_WAVES
#define JWAV_A_WAVES 4
#endif
#ifndef JWAV_B_WAVES
#define JWAV_B_WAVES 4
#endif
#define JT_KERNEL judge_wave_anchor_kernel
#define JT_WAVES_MIN JWAV_A_WAVES
#define JT_PHASE 1
namespace jwavA {
#include "hite_judge_team.inc"
}
#undef JT_KERNEL
#undef JT_WAVES_MIN
#undef JT_PHASE
#define JT_KERNEL judge_wave_rest_kernel
#define JT_WAVES_MIN JWAV_B_WAVES
#define JT_PHASE 2
namespace jwavB {
#include "hite_judge_team.inc"
}
#undef JT_KERNEL
#undef JT_MSA_PTR
#undef JT_WAVES_MIN
#undef JT_PHASE
#define JT_PHASE 0
#define JT_KERNEL judge_wave_lds_kernel
#define JT_MSA_PTR jt_lptr_c
#define JT_LDS_MSA 1
#ifndef JWAV_LDS_WAVES
#define JWAV_LDS_WAVES 2
#endif
#define JT_WAVES_MIN JWAV_LDS_WAVES
namespace jwlds {
#include "hite_judge_team.inc"
}
#undef JT_PHASE
#undef JB
#undef JW
#undef JT_W32
#undef ANCHOR_LDS_COLS
#undef JT_KERNEL
#undef JT_WAVES_MIN
#undef JT_MSA_PTR
#undef JT_LDS_MSA
// the other kernels of this file are four-wavefront workgroups on the block form of the helpers
#define JB 256
#define JW (JB / 64)
using namespace jblk;

// ---------------------------------------------------------------------------------------------
// fold bytes to the ACGTN- alphabet (entry of every host wrapper)
// ---------------------------------------------------------------------------------------------
__global__ void fold_kernel(uint8_t *__restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = fold_sym(p[i]);
}

// ---------------------------------------------------------------------------------------------
// remove_sparse_col_in_align_file  Util.py:10344-10405: one block per alignment.
// keep column c iff c == 0 or c == C-1 or gaps(c) <= R/2 (float: 2*gaps <= R).
// Output rows are written compacted at the same slot offset with stride new_cols.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(JB) sparse_cols_kernel(int n, const uint8_t *__restrict__ msa,
                                                          const int64_t *__restrict__ msa_off,
                                                          const int32_t *__restrict__ rows,
                                                          const int32_t *__restrict__ cols, uint8_t *__restrict__ out,
                                                          int32_t *__restrict__ new_cols, int *__restrict__ colmap,
                                                          const int64_t *__restrict__ col_off) {
    __shared__ int s_scan[8];
    int ci = blockIdx.x;
    if (ci >= n) return;
    const int R = rows[ci], C = cols[ci];
    const uint8_t *m = msa + msa_off[ci];
    uint8_t *o = out + msa_off[ci];
    int *cm = colmap + col_off[ci];  // new index of each kept column, -1 if dropped
    int running = 0;
    for (int base = 0; base < C; base += JB) {
        int c = base + threadIdx.x;
        int keep = 0;
        if (c < C) {
            int gaps = 0;
            for (int r = 0; r < R; r++) gaps += m[(size_t)r * C + c] == '-';
            keep = (c == 0 || c == C - 1 || 2 * gaps <= R) ? 1 : 0;
        }
        int tot;
        int pre = block_excl_scan(keep, s_scan, &tot);
        if (c < C) cm[c] = keep ? running + pre : -1;
        running += tot;
        __syncthreads();
    }
    const int NC = running;
    if (threadIdx.x == 0) new_cols[ci] = NC;
    __syncthreads();
    // compact: in-place safe only if out != msa; rows are written by all threads (coalesced on c)
    for (int r = 0; r < R; r++) {
        const uint8_t *src = m + (size_t)r * C;
        uint8_t *dst = o + (size_t)r * NC;
        for (int c = threadIdx.x; c < C; c += JB) {
            int k = cm[c];
            if (k >= 0) dst[k] = fold_sym(src[c]);
        }
    }
}

// col_base_map over ALL rows: counts of A,C,G,T,N,'-' per column as int32[6]
__global__ void __launch_bounds__(JB) column_vote_kernel(int n, const uint8_t *__restrict__ msa,
                                                          const int64_t *__restrict__ msa_off,
                                                          const int32_t *__restrict__ rows,
                                                          const int32_t *__restrict__ cols,
                                                          const int64_t *__restrict__ col_off,
                                                          int32_t *__restrict__ counts) {
    // one wavefront per 64 columns, rows streamed; lanes are columns here (coalesced row reads)
    int ci = blockIdx.y;
    if (ci >= n) return;
    const int R = rows[ci], C = cols[ci];
    const uint8_t *m = msa + msa_off[ci];
    for (int c = blockIdx.x * JB + threadIdx.x; c < C; c += gridDim.x * JB) {
        int cnt[6] = {0, 0, 0, 0, 0, 0};
        for (int r = 0; r < R; r++) cnt[sym_class(m[(size_t)r * C + c])]++;
        int32_t *o = counts + (col_off[ci] + c) * 6;
#pragma unroll
        for (int k = 0; k < 6; k++) o[k] = cnt[k];
    }
}

// standalone search_boundary_homo_v3 / v4 over all rows of each alignment (rows <= 128)
struct SearchParams {
    int n, variant, win_in, win_out;
    const uint8_t *msa;
    const int64_t *msa_off;
    const int32_t *rows, *cols;
    const int32_t *pos, *side;
    const double *thr, *int_thr, *out_thr;
    const int64_t *col_off;
    uint8_t *cstat;  // 12 bytes per column, indexed by col_off
    int32_t *boundary, *valid;
};
__global__ void __launch_bounds__(JB) search_kernel(SearchParams P) {
    __shared__ JShared S;
    jshared_init(S);
    int ci = blockIdx.x;
    if (ci >= P.n) return;
    const int R = P.rows[ci], C = P.cols[ci];
    const uint8_t *msa = P.msa + P.msa_off[ci];
    uint8_t *cstat = P.cstat + P.col_off[ci] * CS;
    for (int r = threadIdx.x; r < R && r < MAXSEL; r += JB) S.sel[r] = (uint16_t)r;
    __syncthreads();
    blk_colstats(msa, C, S.sel, R, cstat, S.cls);
    int b, valid = 1;
    if (P.variant == 3) b = blk_search_v3(msa, cstat, C, S.sel, R, P.pos[ci], P.side[ci], P.thr[ci], P.win_in, P.win_out, S);
    else b = blk_search_v4(msa, cstat, C, S.sel, R, P.pos[ci], P.side[ci], P.thr[ci], P.int_thr[ci], P.out_thr[ci], P.win_in, P.win_out, S, &valid);
    if (threadIdx.x == 0) { P.boundary[ci] = b; if (P.valid) P.valid[ci] = valid; }
}

__global__ void tsd_search_kernel(int n, const uint8_t *__restrict__ bytes, const int64_t *__restrict__ off,
                                  const int32_t *__restrict__ bs, const int32_t *__restrict__ be, int plant,
                                  int32_t *__restrict__ len_out, uint8_t *__restrict__ left, uint8_t *__restrict__ right) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int C = (int)(off[i + 1] - off[i]);
    uint8_t l[12], r[12];
    int k = tsd_search_v5(RowPlain{bytes + off[i]}, C, bs[i], be[i], plant, l, r);
    len_out[i] = k;
    for (int j = 0; j < 16; j++) { left[(size_t)i * 16 + j] = (k > 0 && j < k) ? l[j] : 0; right[(size_t)i * 16 + j] = (k > 0 && j < k) ? r[j] : 0; }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct DBuf {
    void *p = nullptr;
    ~DBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
    hipError_t up(const void *h, size_t n) {
        hipError_t e = alloc(n);
        if (e != hipSuccess) return e;
        return n ? hipMemcpy(p, h, n, hipMemcpyHostToDevice) : hipSuccess;
    }
};

static int batch_shape(int32_t n, const int64_t *msa_off, const int32_t *rows, const int32_t *cols, int64_t *total_bytes,
                       int64_t *total_cols, int *maxC, int *maxR, int64_t *col_off /* n+1 or NULL */) {
    int64_t tb = 0, tc = 0;
    int mc = 0, mr = 0;
    for (int i = 0; i < n; i++) {
        if (rows[i] < 0 || cols[i] < 0 || cols[i] > 65535 || msa_off[i] < 0) return HITE_EINVAL;
        int64_t end = msa_off[i] + (int64_t)rows[i] * cols[i];
        if (end > tb) tb = end;
        if (col_off) col_off[i] = tc;
        tc += cols[i];
        if (cols[i] > mc) mc = cols[i];
        if (rows[i] > mr) mr = rows[i];
    }
    if (col_off) col_off[n] = tc;
    *total_bytes = tb; *total_cols = tc; *maxC = mc; *maxR = mr;
    return HITE_OK;
}

extern "C" int hite_sparse_cols_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_msa, const int64_t *d_msa_off,
                                     const int32_t *d_rows, const int32_t *d_cols, const int64_t *d_col_off,
                                     int64_t total_cols, uint8_t *d_out, int32_t *d_new_cols, void *stream) {
    if (!ctx || n < 0) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    void *cm = nullptr;
    int rc = hite_scratch2_reserve(ctx, (size_t)(total_cols + 16) * 4, &cm);
    if (rc) return rc;
    hipLaunchKernelGGL(sparse_cols_kernel, dim3(n), dim3(JB), 0, (hipStream_t)stream, n, d_msa, d_msa_off, d_rows, d_cols,
                       d_out, d_new_cols, (int *)cm, d_col_off);
    HITE_CHECK(ctx, hipGetLastError());
    return HITE_OK;
}

extern "C" int hite_sparse_cols(hite_ctx *ctx, int32_t n, const uint8_t *msa, const int64_t *msa_off, const int32_t *rows,
                                const int32_t *cols, uint8_t *out, int32_t *new_cols) {
    if (!ctx || n < 0 || !msa || !msa_off || !rows || !cols || !out || !new_cols) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    int64_t tb, tc; int mc, mr;
    int64_t *co = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    if (!co) return HITE_ENOMEM;
    int rc = batch_shape(n, msa_off, rows, cols, &tb, &tc, &mc, &mr, co);
    if (rc) { free(co); return rc; }
    for (int i = 0; i < n; i++) if (rows[i] <= 0 || cols[i] <= 0) { free(co); return HITE_EINVAL; }
    DBuf dm, dof, dr, dc, dout, dnc, dco;
    hipError_t e;
    e = dm.up(msa, tb); if (e == hipSuccess) e = dof.up(msa_off, n * 8); if (e == hipSuccess) e = dr.up(rows, n * 4);
    if (e == hipSuccess) e = dc.up(cols, n * 4); if (e == hipSuccess) e = dout.alloc(tb); if (e == hipSuccess) e = dnc.alloc(n * 4);
    if (e == hipSuccess) e = dco.up(co, (n + 1) * 8);
    free(co);
    HITE_CHECK(ctx, e);
    rc = hite_sparse_cols_dev(ctx, n, (uint8_t *)dm.p, (int64_t *)dof.p, (int32_t *)dr.p, (int32_t *)dc.p, (int64_t *)dco.p,
                               tc, (uint8_t *)dout.p, (int32_t *)dnc.p, nullptr);
    if (rc) return rc;
    HITE_CHECK(ctx, hipDeviceSynchronize());
    HITE_CHECK(ctx, hipMemcpy(out, dout.p, tb, hipMemcpyDeviceToHost));
    HITE_CHECK(ctx, hipMemcpy(new_cols, dnc.p, n * 4, hipMemcpyDeviceToHost));
    return HITE_OK;
}

extern "C" int hite_column_vote(hite_ctx *ctx, int32_t n, const uint8_t *msa, const int64_t *msa_off, const int32_t *rows,
                                const int32_t *cols, const int64_t *col_off, int32_t *counts_out) {
    if (!ctx || n < 0 || !msa || !counts_out) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    int64_t tb, tc; int mc, mr;
    int rc = batch_shape(n, msa_off, rows, cols, &tb, &tc, &mc, &mr, nullptr);
    if (rc) return rc;
    DBuf dm, dof, dr, dc, dco, dcnt;
    hipError_t e;
    e = dm.up(msa, tb); if (e == hipSuccess) e = dof.up(msa_off, n * 8); if (e == hipSuccess) e = dr.up(rows, n * 4);
    if (e == hipSuccess) e = dc.up(cols, n * 4); if (e == hipSuccess) e = dco.up(col_off, n * 8);
    if (e == hipSuccess) e = dcnt.alloc((size_t)tc * 24);
    HITE_CHECK(ctx, e);
    hipLaunchKernelGGL(fold_kernel, dim3(1024), dim3(256), 0, nullptr, (uint8_t *)dm.p, tb);
    int gx = (mc + JB - 1) / JB; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(column_vote_kernel, dim3(gx, n), dim3(JB), 0, nullptr, n, (uint8_t *)dm.p, (int64_t *)dof.p,
                       (int32_t *)dr.p, (int32_t *)dc.p, (int64_t *)dco.p, (int32_t *)dcnt.p);
    HITE_CHECK(ctx, hipGetLastError());
    HITE_CHECK(ctx, hipDeviceSynchronize());
    HITE_CHECK(ctx, hipMemcpy(counts_out, dcnt.p, (size_t)tc * 24, hipMemcpyDeviceToHost));
    return HITE_OK;
}

extern "C" int hite_boundary_search(hite_ctx *ctx, int32_t n, const uint8_t *msa, const int64_t *msa_off,
                                    const int32_t *rows, const int32_t *cols, const int32_t *pos, const int32_t *side,
                                    const double *thr, const double *int_thr, const double *out_thr, int32_t variant,
                                    int32_t win_in, int32_t win_out, int32_t *boundary_out, int32_t *valid_out) {
    if (!ctx || n < 0 || !msa || !pos || !side || !thr || !boundary_out) return HITE_EINVAL;
    if (variant != 3 && variant != 4) return HITE_EINVAL;
    if (variant == 4 && (!int_thr || !out_thr)) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    int64_t tb, tc; int mc, mr;
    int64_t *co = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    if (!co) return HITE_ENOMEM;
    int rc = batch_shape(n, msa_off, rows, cols, &tb, &tc, &mc, &mr, co);
    if (rc || mr > MAXSEL) { free(co); return HITE_EINVAL; }
    for (int i = 0; i < n; i++) if (rows[i] <= 0 || cols[i] <= 0 || pos[i] < 0 || pos[i] >= cols[i]) { free(co); return HITE_EINVAL; }
    DBuf dm, dof, dr, dc, dco, dpos, dside, dthr, dit, dot, dcs, db, dv;
    hipError_t e;
    e = dm.up(msa, tb); if (e == hipSuccess) e = dof.up(msa_off, n * 8); if (e == hipSuccess) e = dr.up(rows, n * 4);
    if (e == hipSuccess) e = dc.up(cols, n * 4); if (e == hipSuccess) e = dco.up(co, (n + 1) * 8);
    if (e == hipSuccess) e = dpos.up(pos, n * 4); if (e == hipSuccess) e = dside.up(side, n * 4);
    if (e == hipSuccess) e = dthr.up(thr, n * 8);
    if (e == hipSuccess && variant == 4) e = dit.up(int_thr, n * 8);
    if (e == hipSuccess && variant == 4) e = dot.up(out_thr, n * 8);
    if (e == hipSuccess) e = dcs.alloc((size_t)tc * CS + 64); if (e == hipSuccess) e = db.alloc(n * 4); if (e == hipSuccess) e = dv.alloc(n * 4);
    free(co);
    HITE_CHECK(ctx, e);
    hipLaunchKernelGGL(fold_kernel, dim3(1024), dim3(256), 0, nullptr, (uint8_t *)dm.p, tb);
    SearchParams P;
    P.n = n; P.variant = variant; P.win_in = win_in; P.win_out = win_out;
    P.msa = (uint8_t *)dm.p; P.msa_off = (int64_t *)dof.p; P.rows = (int32_t *)dr.p; P.cols = (int32_t *)dc.p;
    P.pos = (int32_t *)dpos.p; P.side = (int32_t *)dside.p; P.thr = (double *)dthr.p;
    P.int_thr = (double *)dit.p; P.out_thr = (double *)dot.p; P.col_off = (int64_t *)dco.p; P.cstat = (uint8_t *)dcs.p;
    P.boundary = (int32_t *)db.p; P.valid = (int32_t *)dv.p;
    hipLaunchKernelGGL(search_kernel, dim3(n), dim3(JB), 0, nullptr, P);
    HITE_CHECK(ctx, hipGetLastError());
    HITE_CHECK(ctx, hipDeviceSynchronize());
    HITE_CHECK(ctx, hipMemcpy(boundary_out, db.p, n * 4, hipMemcpyDeviceToHost));
    if (valid_out) HITE_CHECK(ctx, hipMemcpy(valid_out, dv.p, n * 4, hipMemcpyDeviceToHost));
    return HITE_OK;
}

// which kernel judges which alignment (JUDGE_CLS_*):
//   * one wavefront per alignment when it has <= wrows rows and <= wcols columns, a four-wavefront workgroup otherwise;
//   * of either, the form that holds the alignment in LDS when its rows x cols bytes fit the class's tile (lds_wave / lds_blk).
// One dense list per class, each ordered by cost class (log2 of rows x cols) DESCENDING: the kernels take their entries from
// work queues, and the largest alignments -- one workgroup can spend milliseconds on a 30 000-column alignment -- must not be
// the ones that start last.  Small launches: classify (unless the pipeline did), class histogram, offsets, scatter.
#define JSPLIT_CLASSES 24
struct JudgeLimits { int wrows, wcols, lds_wave[JUDGE_LDS_WAVE_SIZES], lds_blk[JUDGE_LDS_BLOCK_SIZES]; };   // tile bytes, 0 = class unused
__device__ __forceinline__ int judge_class_of(int rows, int cols, const JudgeLimits &L) {
    if (rows <= 0 || cols <= 0) return JUDGE_CLS_BLOCK;
    const long long tile = (long long)rows * cols;
    const bool small = rows <= L.wrows && cols <= L.wcols;
    if (small)
        for (int k = 0; k < JUDGE_LDS_WAVE_SIZES; k++) if (tile <= L.lds_wave[k]) return JUDGE_CLS_LDS + k;
    for (int k = 0; k < JUDGE_LDS_BLOCK_SIZES; k++) if (tile <= L.lds_blk[k]) return JUDGE_CLS_LDS + JUDGE_LDS_WAVE_SIZES + k;
    return small ? JUDGE_CLS_WAVE : JUDGE_CLS_BLOCK;
}
__global__ void judge_classify_kernel(int n, const int32_t *__restrict__ rows, const int32_t *__restrict__ cols, JudgeLimits L,
                                      uint8_t *__restrict__ cls) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cls[i] = (uint8_t)judge_class_of(rows[i], cols[i], L);
}
__device__ __forceinline__ int judge_cost_class(int rows, int cols) {
    const unsigned long long cost = (unsigned long long)(rows > 0 ? rows : 1) * (unsigned long long)(cols > 0 ? cols : 1);
    const int lg = 63 - __clzll(cost);                       // 0 .. ~22
    return lg >= JSPLIT_CLASSES ? JSPLIT_CLASSES - 1 : lg;
}
#define JSPLIT_SLOTS (JUDGE_NCLS * JSPLIT_CLASSES)
__global__ void __launch_bounds__(256) judge_split_count_kernel(int n, const int32_t *__restrict__ rows, const int32_t *__restrict__ cols,
                                                                const uint8_t *__restrict__ cls, unsigned int *__restrict__ hist /* [JUDGE_NCLS][JSPLIT_CLASSES] */) {
    __shared__ unsigned int s_h[JSPLIT_SLOTS];
    if (threadIdx.x < JSPLIT_SLOTS) s_h[threadIdx.x] = 0u;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&s_h[(int)cls[i] * JSPLIT_CLASSES + judge_cost_class(rows[i], cols[i])], 1u);
    __syncthreads();
    if (threadIdx.x < JSPLIT_SLOTS && s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}
// offsets of the cost classes inside each list, largest class first; list lengths into counters[JUDGE_NCLS + k]; hist becomes cursors
__global__ void judge_split_offsets_kernel(unsigned int *__restrict__ hist, unsigned int *__restrict__ counters) {
    if (threadIdx.x >= JUDGE_NCLS) return;
    unsigned int *h = hist + threadIdx.x * JSPLIT_CLASSES;
    unsigned int run = 0;
    for (int c = JSPLIT_CLASSES - 1; c >= 0; c--) { const unsigned int k = h[c]; h[c] = run; run += k; }
    counters[JUDGE_NCLS + threadIdx.x] = run;
}
__global__ void __launch_bounds__(256) judge_split_scatter_kernel(int n, const int32_t *__restrict__ rows, const int32_t *__restrict__ cols,
                                                                  const uint8_t *__restrict__ cls, unsigned int *__restrict__ cursors,
                                                                  int32_t *__restrict__ lists /* [JUDGE_NCLS][n] */) {
    // ranks inside the workgroup by LDS atomics, one global atomic per (workgroup, slot): 50 k atomics on 48 addresses were
    // 0.35 ms per launch.  (The order inside a cost class only schedules the judges; it does not reach any result.)
    __shared__ unsigned int s_cnt[JSPLIT_SLOTS], s_base[JSPLIT_SLOTS];
    if (threadIdx.x < JSPLIT_SLOTS) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int k = 0, slot = 0;
    unsigned int local = 0u;
    if (i < n) {
        k = cls[i];
        slot = k * JSPLIT_CLASSES + judge_cost_class(rows[i], cols[i]);
        local = atomicAdd(&s_cnt[slot], 1u);
    }
    __syncthreads();
    if (threadIdx.x < JSPLIT_SLOTS && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&cursors[threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    if (i < n) lists[(size_t)k * n + s_base[slot] + local] = i;
}

static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

// the limits of the four classes for a batch of n alignments (environment overrides are for measurements and tests)
static JudgeLimits judge_limits(int n) {
    JudgeLimits L;
    // HITE_JUDGE_WAVE_COLS = 0 sends every alignment to the workgroup kernels; the wave kernels' LDS holds the anchor text of
    // <= JWAV_ANCHOR_COLS (2040) columns, wider alignments would search their anchors in global scratch
    L.wcols = env_int("HITE_JUDGE_WAVE_COLS", JWAV_ANCHOR_COLS); L.wrows = env_int("HITE_JUDGE_WAVE_ROWS", 64);
    if (L.wrows > 64) L.wrows = 64;
    if (L.wcols < 0 || L.wrows <= 0) L.wcols = 0;
    // one wavefront per alignment pays off when the batch keeps the machine busy for many rounds (throughput: four times the
    // alignments in flight); a small batch is over in one round, where the workgroup form's shorter chain per alignment wins
    // (C2, 5 000 candidates: 1.8 ms with the wave kernel, 1.3 ms without)
    if (n < env_int("HITE_JUDGE_WAVE_MIN_BATCH", 16384)) L.wcols = 0;
    // LDS tiles (bytes of rows x cols), ascending; a launch holds its tile + 11.4 KB (wavefront) / 19.3 KB (workgroup) of LDS per
    // resident workgroup out of 160 KB per CU.  HITE_JUDGE_LDS = 0 switches all of them off, a size of 0 one class.
    static const int dflt_w[JUDGE_LDS_WAVE_SIZES] = {8192, 16384, 32768}, dflt_b[JUDGE_LDS_BLOCK_SIZES] = {61440, 0};
    static const char *name_w[JUDGE_LDS_WAVE_SIZES] = {"HITE_JUDGE_LDS_WAVE0", "HITE_JUDGE_LDS_WAVE1", "HITE_JUDGE_LDS_WAVE2"};
    static const char *name_b[JUDGE_LDS_BLOCK_SIZES] = {"HITE_JUDGE_LDS_BLOCK0", "HITE_JUDGE_LDS_BLOCK1"};
    // OFF by default: measured on C3 (MI355X, round 4, gpurun_out/s1): calls identical, but judge 11.9 -> 27.3 ms per step with
    // every class on (fill 10.0 -> 7.1 ms), 15.9 ms with the 8 KB / 16 KB wavefront tiles alone, 19.6 ms when the tiles are
    // copied from HBM instead of built.  A tile costs residency (8 / 5 / 3 wavefronts per CU at 8 / 16 / 32 KB against 14 for
    // the kernels on HBM) and the judges turn out to be bound by instruction issue at ~3.5 wavefronts per SIMD, not by the
    // latency of their alignment reads: fewer wavefronts per CU lose more than LDS reads win.
    const bool on = env_int("HITE_JUDGE_LDS", 0) != 0;
    for (int k = 0; k < JUDGE_LDS_WAVE_SIZES; k++) {
        int v = on && L.wcols > 0 ? env_int(name_w[k], dflt_w[k]) : 0;
        L.lds_wave[k] = v < 0 ? 0 : (v > 98304 ? 98304 : v);
    }
    for (int k = 0; k < JUDGE_LDS_BLOCK_SIZES; k++) {
        int v = on ? env_int(name_b[k], dflt_b[k]) : 0;
        L.lds_blk[k] = v < 0 ? 0 : (v > 131072 ? 131072 : v);
    }
    return L;
}

// do the LDS classes exist at all for a batch of n alignments?  (the pipeline leaves alignments out of the fill only then)
extern "C" int hite_judge_lds_enabled(int32_t n) {
    const JudgeLimits L = judge_limits(n);
    int any = 0;
    for (int k = 0; k < JUDGE_LDS_WAVE_SIZES; k++) any |= L.lds_wave[k] > 0;
    for (int k = 0; k < JUDGE_LDS_BLOCK_SIZES; k++) any |= L.lds_blk[k] > 0;
    return any;
}
// JUDGE_CLS_* per alignment, exactly as hite_judge_dev will split a batch of n with these rows / columns
extern "C" int hite_judge_classify_dev(hite_ctx *ctx, int32_t n, const int32_t *d_rows, const int32_t *d_cols, uint8_t *d_cls, void *stream) {
    if (!ctx || n < 0) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    hipLaunchKernelGGL(judge_classify_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, d_rows, d_cols, judge_limits(n), d_cls);
    HITE_CHECK(ctx, hipGetLastError());
    return HITE_OK;
}

extern "C" int hite_judge_dev(hite_ctx *ctx, int32_t te_type, int32_t plant, int32_t n, const uint8_t *d_msa,
                              const int64_t *d_msa_off, const int32_t *d_rows, const int32_t *d_cols,
                              const uint8_t *d_cand, const int64_t *d_cand_off, const int64_t *d_col_off,
                              int32_t max_cols, int32_t max_rows, hite_call *d_calls, uint8_t *d_cons, void *stream) {
    if (!ctx || n < 0 || te_type < 0 || te_type > 2) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    if (max_cols <= 0 || max_cols > 65535 || max_rows <= 0) return HITE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const JudgeLimits L = judge_limits(n);
    const bool overlap = env_int("HITE_JUDGE_OVERLAP", 1) != 0;
    // scratch slots per resident workgroup: 23 * maxC16 + 16 * maxR + 64 (the wave kernels never see more than wcols columns / 64 rows)
    const size_t maxC16 = ((size_t)max_cols + 15) & ~(size_t)15;
    size_t slot[JUDGE_NCLS], mc16[JUDGE_NCLS];
    int grid[JUDGE_NCLS], lds[JUDGE_NCLS];
    bool wave[JUDGE_NCLS];
    size_t maxC16w = (size_t)(L.wcols < max_cols ? L.wcols : max_cols);
    maxC16w = (maxC16w + 15) & ~(size_t)15;
    for (int k = 0; k < JUDGE_NCLS; k++) {
        const int kl = k - JUDGE_CLS_LDS;
        wave[k] = k == JUDGE_CLS_WAVE || (kl >= 0 && kl < JUDGE_LDS_WAVE_SIZES);
        lds[k] = kl < 0 ? 0 : (kl < JUDGE_LDS_WAVE_SIZES ? L.lds_wave[kl] : L.lds_blk[kl - JUDGE_LDS_WAVE_SIZES]);
        mc16[k] = wave[k] ? maxC16w : maxC16;
        slot[k] = ((wave[k] ? 23 * maxC16w + 16 * (size_t)64 : 23 * maxC16 + 16 * (size_t)max_rows) + 64 + 63) & ~(size_t)63;
        const int cap = wave[k] ? 4096 : (kl >= 0 ? 1024 : 2048);
        grid[k] = n < cap ? n : cap;
        if ((k == JUDGE_CLS_WAVE && L.wcols <= 0) || (kl >= 0 && lds[k] <= 0)) grid[k] = 0;
        // keep the scratch bounded (<= 4 GiB per class): fewer resident slots for very wide alignments
        while (grid[k] > 64 && (size_t)grid[k] * slot[k] > ((size_t)4 << 30)) grid[k] /= 2;
    }
    size_t off[JUDGE_NCLS + 1];
    off[0] = 0;
    for (int k = 0; k < JUDGE_NCLS; k++) off[k + 1] = off[k] + (size_t)grid[k] * slot[k];
    const size_t off_l = off[JUDGE_NCLS], cls_bytes = ((size_t)n + 255) & ~(size_t)255;
    void *scr = nullptr;
    const size_t anch_bytes = ((size_t)n * 8 + 255) & ~(size_t)255;
    int rc = hite_scratch_reserve(ctx, off_l + (size_t)JUDGE_NCLS * n * 4 + cls_bytes + anch_bytes + 4096, &scr);
    if (rc) return rc;
    int32_t *lists = (int32_t *)((uint8_t *)scr + off_l);
    uint8_t *cls_own = (uint8_t *)(lists + (size_t)JUDGE_NCLS * n);
    int32_t *anchors = (int32_t *)(cls_own + cls_bytes);
    unsigned int *counters = (unsigned int *)((uint8_t *)anchors + anch_bytes);   // queue heads [0 .. JUDGE_NCLS), list lengths [JUDGE_NCLS .. 2 JUDGE_NCLS), the class histogram, then the queue heads of the two anchor kernels
    unsigned int *hist = counters + 2 * JUDGE_NCLS;
    unsigned int *heads_a = hist + JSPLIT_SLOTS;
    HITE_CHECK(ctx, hipMemsetAsync(counters, 0, (2 * JUDGE_NCLS + JSPLIT_SLOTS + 2) * 4, st));
    const uint8_t *cls = ctx->d_judge_cls;
    if (!cls) {
        hipLaunchKernelGGL(judge_classify_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, d_rows, d_cols, L, cls_own);
        cls = cls_own;
    }
    hipLaunchKernelGGL(judge_split_count_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, d_rows, d_cols, cls, hist);
    hipLaunchKernelGGL(judge_split_offsets_kernel, dim3(1), dim3(64), 0, st, hist, counters);
    hipLaunchKernelGGL(judge_split_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, d_rows, d_cols, cls, hist, lists);
    JudgeParams P[JUDGE_NCLS];
    for (int k = 0; k < JUDGE_NCLS; k++) {
        JudgeParams &p = P[k];
        p.te_type = te_type; p.plant = plant; p.n = n; p.msa = d_msa; p.msa_off = d_msa_off; p.rows = d_rows; p.cols = d_cols;
        p.cand = d_cand; p.cand_off = d_cand_off; p.col_off = d_col_off; p.calls = d_calls; p.cons = d_cons;
        p.scratch = (uint8_t *)scr + off[k]; p.slot_bytes = slot[k]; p.maxC16 = mc16[k];
        p.counter = counters + k; p.list = lists + (size_t)k * n; p.n_list = counters + JUDGE_NCLS + k;
        p.fuse = ctx->judge_fuse;
        p.anchors = anchors;
    }
    // HITE_JUDGE_SPLIT (bit 0: the workgroup class, bit 1: the wavefront class): the class runs as an anchor kernel + the rest
    // (same work list, a queue head each).  OFF by default: measured on C3 (round 6, profiles/r06_judge_split.txt) the judges
    // take 11.84 ms per step in one kernel per class and 11.79 in two -- the anchor kernel compiles to 64 / 86 registers
    // without a spill, the rest keeps 176 / 64 bytes of scratch, and neither the registers nor the wavefronts per SIMD were
    // what the kernels wait for (their wavefronts issue a vector instruction in 8-16 % of their cycles)
    const int split = env_int("HITE_JUDGE_SPLIT", 0);
    // more than 64 KB of LDS per workgroup needs the attribute: once per DEVICE (a second context on another GPU of the process
    // launches the same functions there), and only for a class that has work -- the default path (LDS classes off) asks for nothing
    static unsigned long long attr_done[2] = {0ull, 0ull};      // bit = device id (< 64)
    const unsigned long long dev_bit = 1ull << (ctx->device & 63);
    bool need_blk = false, need_wav = false;
    for (int k = JUDGE_CLS_LDS; k < JUDGE_NCLS; k++) {
        if (grid[k] <= 0) continue;
        if (k >= JUDGE_CLS_LDS + JUDGE_LDS_WAVE_SIZES) need_blk = true; else need_wav = true;
    }
    if (need_blk && !(attr_done[0] & dev_bit)) {
        HITE_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&jblds::judge_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        attr_done[0] |= dev_bit;
    }
    if (need_wav && !(attr_done[1] & dev_bit)) {
        HITE_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&jwlds::judge_wave_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
        attr_done[1] |= dev_bit;
    }
    // the workgroup kernel on HBM holds the long chains (wide / deep alignments): it starts first, the others fill the machine
    // beside it, each on its own stream, the largest tiles first
    hipStream_t aux[HITE_AUX_STREAMS];
    for (int i = 0; i < HITE_AUX_STREAMS; i++) aux[i] = st;
    hipEvent_t ev_fork = nullptr, ev_join[HITE_AUX_STREAMS] = {};
    bool side = false;
    for (int k = 1; k < JUDGE_NCLS; k++) side = side || grid[k] > 0;
    side = side && overlap;
    if (side) {
        rc = hite_aux_streams(ctx, HITE_AUX_STREAMS, aux, &ev_fork, ev_join);
        if (rc) return rc;
        HITE_CHECK(ctx, hipEventRecord(ev_fork, st));
        for (int i = 0; i < HITE_AUX_STREAMS; i++) if (grid[i + 1] > 0) HITE_CHECK(ctx, hipStreamWaitEvent(aux[i], ev_fork, 0));
    }
    if (split & 1) {
        JudgeParams pa = P[JUDGE_CLS_BLOCK];
        pa.counter = heads_a;
        hipLaunchKernelGGL(jblkA::judge_anchor_kernel, dim3(grid[JUDGE_CLS_BLOCK]), dim3(256), 0, st, pa);
        hipLaunchKernelGGL(jblkB::judge_rest_kernel, dim3(grid[JUDGE_CLS_BLOCK]), dim3(256), 0, st, P[JUDGE_CLS_BLOCK]);
    } else
        hipLaunchKernelGGL(jblk::judge_kernel, dim3(grid[JUDGE_CLS_BLOCK]), dim3(256), 0, st, P[JUDGE_CLS_BLOCK]);
    for (int k = JUDGE_NCLS - 1; k >= 1; k--) {
        if (grid[k] <= 0) continue;
        if (k == JUDGE_CLS_WAVE && (split & 2)) {
            JudgeParams pa = P[k];
            pa.counter = heads_a + 1;
            hipLaunchKernelGGL(jwavA::judge_wave_anchor_kernel, dim3(grid[k]), dim3(64), 0, aux[k - 1], pa);
            hipLaunchKernelGGL(jwavB::judge_wave_rest_kernel, dim3(grid[k]), dim3(64), 0, aux[k - 1], P[k]);
        }
        else if (k == JUDGE_CLS_WAVE) hipLaunchKernelGGL(jwav::judge_wave_kernel, dim3(grid[k]), dim3(64), 0, aux[k - 1], P[k]);
        else if (wave[k]) hipLaunchKernelGGL(jwlds::judge_wave_lds_kernel, dim3(grid[k]), dim3(64), (size_t)lds[k], aux[k - 1], P[k]);
        else hipLaunchKernelGGL(jblds::judge_lds_kernel, dim3(grid[k]), dim3(256), (size_t)lds[k], aux[k - 1], P[k]);
    }
    HITE_CHECK(ctx, hipGetLastError());
    if (side)
        for (int i = 0; i < HITE_AUX_STREAMS; i++) {
            if (grid[i + 1] <= 0) continue;
            HITE_CHECK(ctx, hipEventRecord(ev_join[i], aux[i]));
            HITE_CHECK(ctx, hipStreamWaitEvent(st, ev_join[i], 0));
        }
    return HITE_OK;
}

extern "C" int hite_judge(hite_ctx *ctx, int32_t te_type, int32_t plant, int32_t n, const uint8_t *msa,
                          const int64_t *msa_off, const int32_t *rows, const int32_t *cols, const uint8_t *cand,
                          const int64_t *cand_off, hite_call *calls, uint8_t *cons) {
    if (!ctx || n < 0 || !msa || !msa_off || !rows || !cols || !cand || !cand_off || !calls || !cons) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    int64_t tb, tc; int mc, mr;
    int64_t *co = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    if (!co) return HITE_ENOMEM;
    int rc = batch_shape(n, msa_off, rows, cols, &tb, &tc, &mc, &mr, co);
    if (rc) { free(co); return rc; }
    if (mc <= 0 || mr <= 0) { free(co); return HITE_EINVAL; }
    int64_t cand_bytes = cand_off[n];
    int64_t cons_bytes = tc + 8 * (int64_t)n;
    DBuf dm, dof, dr, dc, dco, dcand, dcoff, dcalls, dcons;
    hipError_t e;
    e = dm.up(msa, tb); if (e == hipSuccess) e = dof.up(msa_off, n * 8); if (e == hipSuccess) e = dr.up(rows, n * 4);
    if (e == hipSuccess) e = dc.up(cols, n * 4); if (e == hipSuccess) e = dco.up(co, (n + 1) * 8);
    if (e == hipSuccess) e = dcand.up(cand, cand_bytes); if (e == hipSuccess) e = dcoff.up(cand_off, (n + 1) * 8);
    if (e == hipSuccess) e = dcalls.alloc(sizeof(hite_call) * n); if (e == hipSuccess) e = dcons.alloc(cons_bytes + 16);
    free(co);
    HITE_CHECK(ctx, e);
    hipLaunchKernelGGL(fold_kernel, dim3(1024), dim3(256), 0, nullptr, (uint8_t *)dm.p, tb);
    hipLaunchKernelGGL(fold_kernel, dim3(64), dim3(256), 0, nullptr, (uint8_t *)dcand.p, cand_bytes);
    rc = hite_judge_dev(ctx, te_type, plant, n, (uint8_t *)dm.p, (int64_t *)dof.p, (int32_t *)dr.p, (int32_t *)dc.p,
                        (uint8_t *)dcand.p, (int64_t *)dcoff.p, (int64_t *)dco.p, mc, mr, (hite_call *)dcalls.p,
                        (uint8_t *)dcons.p, nullptr);
    if (rc) return rc;
    HITE_CHECK(ctx, hipDeviceSynchronize());
    HITE_CHECK(ctx, hipMemcpy(calls, dcalls.p, sizeof(hite_call) * n, hipMemcpyDeviceToHost));
    HITE_CHECK(ctx, hipMemcpy(cons, dcons.p, cons_bytes, hipMemcpyDeviceToHost));
    return HITE_OK;
}

extern "C" int hite_tsd_search(hite_ctx *ctx, int32_t n, const uint8_t *rows_bytes, const int64_t *row_off,
                               const int32_t *bstart, const int32_t *bend, int32_t plant, int32_t *tsd_len_out,
                               uint8_t *left_out, uint8_t *right_out) {
    if (!ctx || n < 0 || !rows_bytes || !row_off || !bstart || !bend || !tsd_len_out || !left_out || !right_out) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    DBuf db, dof, dbs, dbe, dl, dlo, dro;
    hipError_t e;
    e = db.up(rows_bytes, row_off[n]); if (e == hipSuccess) e = dof.up(row_off, (n + 1) * 8);
    if (e == hipSuccess) e = dbs.up(bstart, n * 4); if (e == hipSuccess) e = dbe.up(bend, n * 4);
    if (e == hipSuccess) e = dl.alloc(n * 4); if (e == hipSuccess) e = dlo.alloc((size_t)n * 16); if (e == hipSuccess) e = dro.alloc((size_t)n * 16);
    HITE_CHECK(ctx, e);
    hipLaunchKernelGGL(tsd_search_kernel, dim3((n + 127) / 128), dim3(128), 0, nullptr, n, (uint8_t *)db.p, (int64_t *)dof.p,
                       (int32_t *)dbs.p, (int32_t *)dbe.p, plant, (int32_t *)dl.p, (uint8_t *)dlo.p, (uint8_t *)dro.p);
    HITE_CHECK(ctx, hipGetLastError());
    HITE_CHECK(ctx, hipDeviceSynchronize());
    HITE_CHECK(ctx, hipMemcpy(tsd_len_out, dl.p, n * 4, hipMemcpyDeviceToHost));
    HITE_CHECK(ctx, hipMemcpy(left_out, dlo.p, (size_t)n * 16, hipMemcpyDeviceToHost));
    HITE_CHECK(ctx, hipMemcpy(right_out, dro.p, (size_t)n * 16, hipMemcpyDeviceToHost));
    return HITE_OK;
}

// ---------------------------------------------------------------------------------------------
// LTR flank-frame vote of the vendored FiLTR:
//   judge_left_frame_LTR  /root/reference/bin/FiLTR-main/src/Util.py:9327-9462
//   judge_right_frame_LTR /root/reference/bin/FiLTR-main/src/Util.py:9175-9325
// One wavefront per matrix (rows = the flank frames of the copies of one LTR candidate, no alignment involved):
// lanes own columns for the symbol counts, the valid-column list is compacted with ballots in visiting order, then
// lanes own windows (binary64 sums in the reference's order).  Bytes outside ACGTN- count as N (deviation i).
// ---------------------------------------------------------------------------------------------
#define LTR_MAXC 1024
__global__ void __launch_bounds__(64) ltr_frame_kernel(int n, const uint8_t *__restrict__ frames, const int64_t *__restrict__ off,
                                                       const int32_t *__restrict__ rows, const int32_t *__restrict__ cols, int flank,
                                                       int window, int side, int32_t *__restrict__ ok_out, int32_t *__restrict__ b_out) {
    __shared__ uint16_t s_gap[LTR_MAXC], s_mx[LTR_MAXC], s_col[LTR_MAXC], s_cnt[LTR_MAXC];
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= n) return;
    const int R = rows[k], C = cols[k];
    if (R <= 1) { if (lane == 0) { ok_out[k] = 1; b_out[k] = -1; } return; }
    if (C <= 0 || C > LTR_MAXC || flank <= 0 || flank > C || window <= 0) { if (lane == 0) { ok_out[k] = -1; b_out[k] = -1; } return; }
    const uint8_t *m = frames + off[k];
    for (int c = lane; c < C; c += 64) {
        int cnt[6] = {0, 0, 0, 0, 0, 0};
        for (int r = 0; r < R; r++) cnt[sym_class(m[(size_t)r * C + c])]++;
        int mx = cnt[0];
#pragma unroll
        for (int q = 1; q < 5; q++) mx = cnt[q] > mx ? cnt[q] : mx;
        s_gap[c] = (uint16_t)(cnt[5] > 65535 ? 65535 : cnt[5]);
        s_mx[c] = (uint16_t)(mx > 65535 ? 65535 : mx);
    }
    __syncthreads();
    const int pos = side == 0 ? flank - 1 : 0;
    const int vthr = R / 2;
    int running = 0;
    for (int base = 0; base < C && running < flank; base += 64) {
        const int v = base + lane;
        const int c = side == 0 ? pos - v : pos + v;
        bool ok = c >= 0 && c < C;
        if (ok) { const int gap = s_gap[c]; ok = (R - gap > 1) && gap <= vthr; }
        const unsigned long long bal = __ballot(ok);
        const int idx = running + __popcll(bal & ((1ull << lane) - 1ull));
        if (ok && idx < flank) { s_col[idx] = (uint16_t)c; s_cnt[idx] = s_mx[c]; }
        running += __popcll(bal);
    }
    __syncthreads();
    const int nv = running < flank ? running : flank;
    const double thr = R <= 5 ? 0.95 : (R <= 10 ? 0.9 : 0.85);
    const double lim = thr - 0.1;
    int found = -1;
    for (int base = 0; base + window <= nv && found == -1; base += 64) {
        const int i = base + lane;
        bool hit = false;
        int first = -1;
        if (i + window <= nv) {
            double sum = 0.0;
            for (int q = 0; q < window; q++) {
                const int idx = nv - 1 - (i + q);
                const double ratio = (double)s_cnt[idx] / (double)R;
                if (ratio >= lim && first == -1) first = s_col[idx];
                sum += ratio;
            }
            hit = sum / (double)window >= thr;
        }
        const unsigned long long bal = __ballot(hit);
        if (bal) found = __shfl(first, __ffsll((long long)bal) - 1);
    }
    if (lane == 0) {
        const int tol = side == 0 ? 5 : 20;
        int d = found - pos; if (d < 0) d = -d;
        ok_out[k] = (found != -1 && d > tol) ? 0 : 1;
        b_out[k] = found;
    }
}

extern "C" int hite_ltr_frame(hite_ctx *ctx, int32_t n, const uint8_t *frames, const int64_t *off, const int32_t *rows,
                              const int32_t *cols, int32_t flank, int32_t window, int32_t side, int32_t *ok_out, int32_t *boundary_out) {
    if (!ctx || n < 0 || (n > 0 && (!frames || !off || !rows || !cols || !ok_out || !boundary_out)) || side < 0 || side > 1)
        return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    int64_t total = 0;
    for (int i = 0; i < n; i++) {
        if (rows[i] < 0 || cols[i] < 0) return HITE_EINVAL;
        int64_t e = off[i] + (int64_t)rows[i] * cols[i];
        if (e > total) total = e;
    }
    DBuf df, doff, dr, dc, dok, db;
    hipError_t e = df.alloc((size_t)total + 16);
    if (e == hipSuccess && total) e = hipMemcpy(df.p, frames, (size_t)total, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = doff.up(off, (size_t)n * 8);
    if (e == hipSuccess) e = dr.up(rows, (size_t)n * 4);
    if (e == hipSuccess) e = dc.up(cols, (size_t)n * 4);
    if (e == hipSuccess) e = dok.alloc((size_t)n * 4);
    if (e == hipSuccess) e = db.alloc((size_t)n * 4);
    HITE_CHECK(ctx, e);
    hipLaunchKernelGGL(ltr_frame_kernel, dim3(n), dim3(64), 0, nullptr, n, (const uint8_t *)df.p, (const int64_t *)doff.p,
                       (const int32_t *)dr.p, (const int32_t *)dc.p, flank, window, side, (int32_t *)dok.p, (int32_t *)db.p);
    HITE_CHECK(ctx, hipGetLastError());
    HITE_CHECK(ctx, hipDeviceSynchronize());
    HITE_CHECK(ctx, hipMemcpy(ok_out, dok.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HITE_CHECK(ctx, hipMemcpy(boundary_out, db.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) if (ok_out[i] < 0) return HITE_EINVAL;   // frame wider than LTR_MAXC or flank > columns
    return HITE_OK;
}

// ---------------------------------------------------------------------------------------------
// LTR frames of the vendored FiLTR: get_both_ends_frame + its remove_sparse_col_in_align_file
//   /root/reference/bin/FiLTR-main/src/Util.py:1401-1497, 1341-1399
// One block per alignment: anchors = first row that carries both 20-mers of the terminal sequence within 2 edits (the
// same block-level search as the judges), columns with more than R/2 gaps drop out except the two anchor columns, then
// every row's left frame (flank columns ending before the start anchor, '-' padded on the left), right frame (flank
// columns from the END anchor column itself, '-' padded on the right) and the full-length row between them.
// slot layout (multiples of maxC16): ung 1 | reflex 4 | minfo 2 | keep 1 | inv 4
// ---------------------------------------------------------------------------------------------
struct BothEndsParams {
    int n, flank;
    const uint8_t *msa; const int64_t *msa_off; const int32_t *rows, *cols;
    const uint8_t *cand; const int64_t *cand_off;
    uint8_t *frames; const int64_t *frame_off;    // rows x 2 flank per alignment
    uint8_t *full; const int64_t *full_off;       // rows x (2 flank + cols) per alignment (stride), full_cols used
    int32_t *full_cols, *new_pos, *status;        // new_pos[2a] = start, [2a+1] = end (cleaned coordinates)
    uint8_t *scratch; size_t slot_bytes, maxC16;
};

__global__ void __launch_bounds__(JB) ltr_both_ends_kernel(BothEndsParams P) {
    __shared__ JShared S;
    jshared_init(S);
    const int a = blockIdx.x;
    uint8_t *slot = P.scratch + (size_t)a * P.slot_bytes;
    const int R = P.rows[a], C = P.cols[a], F = P.flank;
    const uint8_t *msa = P.msa + P.msa_off[a];
    const uint8_t *cand = P.cand + P.cand_off[a];
    const int clen = (int)(P.cand_off[a + 1] - P.cand_off[a]);
    uint8_t *ung = slot;
    int *reflex = (int *)(slot + P.maxC16);
    uint8_t *minfo = slot + 5 * P.maxC16;
    uint8_t *keep = slot + 7 * P.maxC16;
    int *inv = (int *)(slot + 8 * P.maxC16);
    if (threadIdx.x == 0) { P.full_cols[a] = 0; P.new_pos[2 * a] = -1; P.new_pos[2 * a + 1] = -1; }
    if (R <= 0 || C <= 0 || clen <= 0) { if (threadIdx.x == 0) P.status[a] = 1; return; }
    const int m1 = clen < 20 ? clen : 20;
    if (threadIdx.x < 20 && (int)threadIdx.x < m1) { S.pat[0][threadIdx.x] = cand[threadIdx.x]; S.pat[1][threadIdx.x] = cand[clen - m1 + threadIdx.x]; }
    __syncthreads();
    int astart = -1, aend = -1;
    for (int r = 0; r < R; r++) {                                                   // :1408-1432
        int n = blk_ungap_row(msa + (size_t)r * C, C, ung, reflex, S);
        int fs = blk_fnm(S.pat[0], m1, ung, n, 2, minfo, 0, S);
        if (fs < 0) continue;
        int le = blk_fnm(S.pat[1], m1, ung, n, 2, minfo, 1, S);
        if (le < 0) continue;
        astart = reflex[fs];
        aend = reflex[le - 1];
        break;
    }
    if (astart == -1 || aend == -1) { if (threadIdx.x == 0) P.status[a] = 1; return; }
    if (astart == aend) { if (threadIdx.x == 0) P.status[a] = 2; return; }
    __syncthreads();
    // kept columns (:1375-1386) and their order-preserving compaction
    if (threadIdx.x == 0) S.iv[0] = 0;
    __syncthreads();
    for (int base = 0; base < C; base += JB) {
        const int c = base + threadIdx.x;
        bool k = false;
        if (c < C) {
            int gap = 0;
            for (int r = 0; r < R; r++) gap += msa[(size_t)r * C + c] == '-';
            k = c == astart || c == aend || 2 * gap <= R;
            keep[c] = k;
        }
        const unsigned long long bal = __ballot(k);
        const int lane = lane_id(), w = wave_id();
        if (lane == 0) S.scan[w] = __popcll(bal);
        __syncthreads();
        int off = S.iv[0];
        for (int i = 0; i < w; i++) off += S.scan[i];
        if (k) inv[off + __popcll(bal & ((1ull << lane) - 1ull))] = c;
        __syncthreads();
        if (threadIdx.x == 0) for (int i = 0; i < JW; i++) S.iv[0] += S.scan[i];
        __syncthreads();
    }
    const int K = S.iv[0];
    // new_start / new_end = kept columns before each anchor
    if (threadIdx.x == 0) { S.iv[1] = 0; S.iv[2] = 0; }
    __syncthreads();
    {
        int cs = 0, ce = 0;
        for (int c = threadIdx.x; c < C; c += JB) if (keep[c]) { cs += c < astart; ce += c < aend; }
        atomicAdd(&S.iv[1], cs); atomicAdd(&S.iv[2], ce);
    }
    __syncthreads();
    const int ns = S.iv[1], ne = S.iv[2];
    const int mid = ne > ns ? ne - ns : 0;
    const int width = 2 * F + mid;
    uint8_t *frames = P.frames + P.frame_off[a];
    uint8_t *full = P.full + P.full_off[a];
    const int stride = 2 * F + C;
    for (int r = 0; r < R; r++) {
        const uint8_t *row = msa + (size_t)r * C;
        for (int j = threadIdx.x; j < width; j += JB) {
            uint8_t ch;
            if (j < F) { const int k = ns - F + j; ch = k >= 0 ? row[inv[k]] : (uint8_t)'-'; frames[(size_t)r * 2 * F + j] = ch; }
            else if (j < F + mid) ch = row[inv[ns + (j - F)]];
            else { const int k = ne + (j - F - mid); ch = k < K ? row[inv[k]] : (uint8_t)'-'; frames[(size_t)r * 2 * F + F + (j - F - mid)] = ch; }
            full[(size_t)r * stride + j] = ch;
        }
    }
    if (threadIdx.x == 0) { P.full_cols[a] = width; P.new_pos[2 * a] = ns; P.new_pos[2 * a + 1] = ne; P.status[a] = 0; }
}

extern "C" int hite_ltr_both_ends(hite_ctx *ctx, int32_t n, const uint8_t *msa, const int64_t *msa_off, const int32_t *rows,
                                  const int32_t *cols, const uint8_t *cand, const int64_t *cand_off, int32_t flank, uint8_t *frames,
                                  const int64_t *frame_off, uint8_t *full, const int64_t *full_off, int32_t *full_cols, int32_t *new_pos,
                                  int32_t *status) {
    if (!ctx || n < 0 || flank <= 0 ||
        (n > 0 && (!msa || !msa_off || !rows || !cols || !cand || !cand_off || !frames || !frame_off || !full || !full_off || !full_cols ||
                   !new_pos || !status)))
        return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    int64_t msa_total = 0, fr_total = 0, fu_total = 0;
    int maxC = 16;
    for (int i = 0; i < n; i++) {
        if (rows[i] < 0 || cols[i] < 0 || cols[i] > 65000) return HITE_EINVAL;
        const int64_t e = msa_off[i] + (int64_t)rows[i] * cols[i], f = frame_off[i] + (int64_t)rows[i] * 2 * flank,
                      u = full_off[i] + (int64_t)rows[i] * (2 * flank + cols[i]);
        if (e > msa_total) msa_total = e;
        if (f > fr_total) fr_total = f;
        if (u > fu_total) fu_total = u;
        if (cols[i] > maxC) maxC = cols[i];
    }
    const size_t maxC16 = ((size_t)maxC + 15) & ~(size_t)15;
    const size_t slot = 12 * maxC16 + 64;
    DBuf dm, dmo, dr, dc, dcd, dco, dfr, dfro, dfu, dfuo, dfc, dnp, dst, dscr;
    hipError_t e = dm.alloc((size_t)msa_total + 16);
    if (e == hipSuccess && msa_total) e = hipMemcpy(dm.p, msa, (size_t)msa_total, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = dmo.up(msa_off, (size_t)n * 8);
    if (e == hipSuccess) e = dr.up(rows, (size_t)n * 4);
    if (e == hipSuccess) e = dc.up(cols, (size_t)n * 4);
    if (e == hipSuccess) e = dcd.alloc((size_t)cand_off[n] + 16);
    if (e == hipSuccess && cand_off[n] > 0) e = hipMemcpy(dcd.p, cand, (size_t)cand_off[n], hipMemcpyHostToDevice);
    if (e == hipSuccess) e = dco.up(cand_off, ((size_t)n + 1) * 8);
    if (e == hipSuccess) e = dfr.alloc((size_t)fr_total + 16);
    if (e == hipSuccess) e = dfro.up(frame_off, (size_t)n * 8);
    if (e == hipSuccess) e = dfu.alloc((size_t)fu_total + 16);
    if (e == hipSuccess) e = dfuo.up(full_off, (size_t)n * 8);
    if (e == hipSuccess) e = dfc.alloc((size_t)n * 4);
    if (e == hipSuccess) e = dnp.alloc((size_t)n * 8);
    if (e == hipSuccess) e = dst.alloc((size_t)n * 4);
    if (e == hipSuccess) e = dscr.alloc(slot * (size_t)n);
    HITE_CHECK(ctx, e);
    HITE_CHECK(ctx, hipMemset(dfr.p, '-', (size_t)fr_total + 16));
    HITE_CHECK(ctx, hipMemset(dfu.p, 0, (size_t)fu_total + 16));
    BothEndsParams P;
    P.n = n; P.flank = flank; P.msa = (const uint8_t *)dm.p; P.msa_off = (const int64_t *)dmo.p; P.rows = (const int32_t *)dr.p;
    P.cols = (const int32_t *)dc.p; P.cand = (const uint8_t *)dcd.p; P.cand_off = (const int64_t *)dco.p; P.frames = (uint8_t *)dfr.p;
    P.frame_off = (const int64_t *)dfro.p; P.full = (uint8_t *)dfu.p; P.full_off = (const int64_t *)dfuo.p; P.full_cols = (int32_t *)dfc.p;
    P.new_pos = (int32_t *)dnp.p; P.status = (int32_t *)dst.p; P.scratch = (uint8_t *)dscr.p; P.slot_bytes = slot; P.maxC16 = maxC16;
    hipLaunchKernelGGL(ltr_both_ends_kernel, dim3(n), dim3(JB), 0, nullptr, P);
    HITE_CHECK(ctx, hipGetLastError());
    HITE_CHECK(ctx, hipDeviceSynchronize());
    HITE_CHECK(ctx, hipMemcpy(frames, dfr.p, (size_t)fr_total, hipMemcpyDeviceToHost));
    HITE_CHECK(ctx, hipMemcpy(full, dfu.p, (size_t)fu_total, hipMemcpyDeviceToHost));
    HITE_CHECK(ctx, hipMemcpy(full_cols, dfc.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HITE_CHECK(ctx, hipMemcpy(new_pos, dnp.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    HITE_CHECK(ctx, hipMemcpy(status, dst.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return HITE_OK;
}
