// hite_common.h -- shared host/device helpers of libhite_gpu.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/hite_gpu.h"

#define HITE_WAVE 64
#define HITE_BLOCK 256

// which judge kernel takes an alignment (hite_judge.hip).  The LDS classes are one launch per tile size: the LDS a workgroup
// holds is fixed per launch, and the tile decides how many alignments a CU has in flight
#define JUDGE_CLS_BLOCK 0       // workgroup per alignment, alignment in HBM
#define JUDGE_CLS_WAVE 1        // wavefront per alignment, alignment in HBM
#define JUDGE_CLS_LDS 2         // first class whose kernel builds the alignment in LDS
#define JUDGE_LDS_WAVE_SIZES 3  // classes 2..4: wavefront per alignment, tile sizes ascending
#define JUDGE_LDS_BLOCK_SIZES 2 // classes 5..6: workgroup per alignment
#define JUDGE_NCLS (JUDGE_CLS_LDS + JUDGE_LDS_WAVE_SIZES + JUDGE_LDS_BLOCK_SIZES)
#define HITE_AUX_STREAMS (JUDGE_NCLS - 1)
// sources of the sparse star alignment (hite_msa.hip): what star_fill_sparse_kernel reads
struct JudgeFuse {
    const uint8_t *win;
    const int64_t *win_off;
    const int32_t *win_len;
    const int32_t *row_first;
    const int64_t *ops_base;
    const uint16_t *ops;
    const uint32_t *lay;
    const int32_t *last_extra;
    const int32_t *row_map;
    const int32_t *rows_eff;
};

struct hite_ctx {
    int device;
    char err[512];
    // resident genome: 2-bit bases (16 per u32), 1-bit non-ACGT mask (32 per u32)
    uint32_t *d_bases;
    uint32_t *d_nmask;
    int64_t *d_contig_off;  // n_contigs + 1 (base index of each contig in the packed arrays)
    int64_t *h_contig_off;
    int32_t n_contigs;
    int64_t n_bases;
    // what happened to the packed genome (hite_copies.hip keeps the minimizer tiles of an index build and redoes only those a mask touched):
    // the epoch counts every change other than hite_genome_mask, whose intervals since then are logged (global positions, half open)
    int64_t genome_epoch;
    int64_t *mask_log;          // pairs
    int64_t mask_log_n, mask_log_cap;
    // grow-only scratch
    void *d_scratch;
    size_t scratch_bytes;
    void *d_scratch2;
    size_t scratch2_bytes;
    // optional per-stage HIP-event profiling (hite_profile_*)
    int prof_on;
    int prof_n;                 // stages seen
    char prof_name[64][32];
    double prof_ms[64];
    int64_t prof_count[64];
    int prof_pending;           // event pairs recorded and not yet resolved
    void *prof_ev[512][2];
    int prof_stage[512];
    // star alignment (hite_align.hip / hite_msa.hip): state of the pairwise aligner, row map of the last alignment call
    void *align_state;
    const int32_t *d_msa_row_map;   // per compacted row: source row (NULL: identity)
    const int32_t *d_msa_rows_eff;  // per candidate: rows that were aligned (NULL: all)
    uint32_t *d_msa_lay;             // layout words of the last sparse star alignment (hite_msa.hip)
    int msa_long;                    // the last star alignment held windows of more than 1 536 bases (the fill's unroll)
    const int64_t *d_msa_win_off;    // per row of the last star alignment: its window, and where its back pads begin (HITE_IS_ROW_PAD;
    const int32_t *d_msa_win_len;    // layout, fill and the judge's LDS kernels read the rows through these)
    const uint32_t *d_msa_pads;      // per row: pad bytes in front | behind << 16
    int32_t copy_interval;          // hite_copy_config_ctx: 1 aligned intervals / 0 whole-candidate intervals / -1 the process default
    int32_t *d_contig_rank;         // byte order of "<contig name>:" among the contigs (hite_set_contig_order; NULL: the index)
    // side streams + one fork event and a join event per stream for kernels that run beside each other inside one call (the
    // judge kernels)
    void *aux_stream[HITE_AUX_STREAMS];
    void *aux_fork;
    void *aux_join[HITE_AUX_STREAMS];
    // fused fill + judge (hite_pipeline.hip -> hite_judge.hip): the pipeline classifies the alignments (hite_judge_classify_dev),
    // leaves the LDS classes out of the fill and hands the judge what it needs to build them in LDS
    int32_t seed_rank, seed_world;  // hite_seed_shard: the share of the all-vs-all stage this context computes (world 0: all of it)
    const uint8_t *d_judge_cls;     // per alignment: JUDGE_CLS_* (NULL: hite_judge_dev classifies by itself)
    JudgeFuse judge_fuse;           // win == NULL: the LDS classes copy their alignment from d_msa
    void *fmea_arena;               // grow-only arena of the sort / sweep / chain routines of hite_fmea.hip (Arena *; hite_fmea_release)
};
void hite_fmea_release(hite_ctx *ctx);
int hite_aux_streams(hite_ctx *ctx, int k, hipStream_t *st, hipEvent_t *fork_ev, hipEvent_t *join_ev);

// record the time of everything enqueued on `st` between begin and end as stage `name`
int hite_prof_begin(hite_ctx *ctx, const char *name, hipStream_t st);
void hite_prof_end(hite_ctx *ctx, int token, hipStream_t st);
void hite_prof_resolve(hite_ctx *ctx);

#define HITE_CHECK(ctx, call)                                                                         \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess) {                                                                      \
            if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d %s: %s", __FILE__, __LINE__, #call, \
                              hipGetErrorString(e__));                                                \
            return HITE_EHIP;                                                                         \
        }                                                                                             \
    } while (0)

int hite_scratch_reserve(hite_ctx *ctx, size_t bytes, void **out);
int hite_scratch2_reserve(hite_ctx *ctx, size_t bytes, void **out);

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// symbol classes: 0..3 = ACGT, 4 = N (and every other non-gap byte), 5 = '-'
__device__ __forceinline__ int sym_class(uint8_t c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        case '-': return 5;
        default: return 4;
    }
}
__device__ __forceinline__ uint8_t class_sym(int k) { return (uint8_t)("ACGTN-"[k]); }
__device__ __forceinline__ uint8_t fold_sym(uint8_t c) { return class_sym(sym_class(c)); }
__device__ __forceinline__ uint8_t comp_sym(uint8_t c) {
    switch (c) {
        case 'A': return 'T';
        case 'T': return 'A';
        case 'C': return 'G';
        case 'G': return 'C';
        default: return 'N';
    }
}

// exclusive scan over the 256 threads of a block (s_tmp: >= 8 ints of LDS); returns the
// exclusive prefix of v and the block total in *total.  Contains __syncthreads().
__device__ __forceinline__ int block_excl_scan(int v, int *s_tmp, int *total) {
    int lane = lane_id(), w = wave_id();
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    __syncthreads();
    if (lane == 63) s_tmp[w] = x;
    __syncthreads();
    int base = 0, tot = 0;
    int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; i++) {
        int t = s_tmp[i];
        if (i < w) base += t;
        tot += t;
    }
    *total = tot;
    return base + x - v;
}

// Python s[a:b] normalisation on a length-n sequence
__device__ __forceinline__ void py_slice(int64_t a, int64_t b, int64_t n, int *lo, int *hi) {
    if (a < 0) { a += n; if (a < 0) a = 0; }
    if (b < 0) { b += n; if (b < 0) b = 0; }
    if (a > n) a = n;
    if (b > n) b = n;
    if (b < a) b = a;
    *lo = (int)a; *hi = (int)b;
}
