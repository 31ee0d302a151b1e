// hite_align.h -- internal interface of the pairwise aligner (hite_align.hip), used by the star alignment (hite_msa.hip).
#pragma once
#include "hite_common.h"

// Aligns every row of every candidate to the candidate's first row (its centre).  Row r of candidate c (global row
// g = row_first[c] + r, r >= 1) gets m + 1 ops at d_ops + ops_base[c] + r * (m + 1), m = length of the centre:
//   ops[p], p < m : q | gap << 15 -- centre position p faces row position q (gap = 0) or a gap before row position q
// (oracle/hite_oracle_nw.c).  d_row_dead (total_rows, may be NULL): 1 for a row that could not be aligned;
// d_cand_flag (n_cand, may be NULL): set to 2 for candidates that lost a row; d_info (5 x total_rows, may be NULL):
// per row the cost U, certified, status, k*, band words | 0x100 (fall-back).  `tag` names the profiler stages.
int hite_align_run(hite_ctx *ctx, int32_t n_cand, const uint8_t *d_win, const int64_t *d_win_off, const int32_t *d_win_len,
                   const int32_t *d_row_first, int64_t total_rows, const int64_t *d_ops_base, uint16_t *d_ops,
                   int32_t *d_row_dead, int32_t *d_cand_flag, int32_t *d_info, const char *tag, hipStream_t st);
void hite_align_release(hite_ctx *ctx);
