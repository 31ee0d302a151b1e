/*
 * hite_gpu.h -- C ABI of libhite_gpu.so: the MI355X-native (gfx950, HIP) implementation of
 * HiTE's dynamic-boundary-adjustment hot path.  Plain pointers and sizes only.
 *
 * HiTE is pure Python: it has no FFI today.  Each entry point below replaces the inside of
 * one Python function of /root/reference/module/Util.py (cited per function); the binding a
 * maintainer adds is the ctypes stub shown in INTEGRATION.md (hite_amd/_lib.py is that stub).
 *
 * Conventions
 *  - every function returns 0 on success or a negative HITE_E* code; nothing aborts.
 *  - "host" entry points take host pointers (numpy buffers) and copy in/out;
 *    "_dev" entry points take DEVICE pointers and a hipStream_t (as void*), launch
 *    asynchronously on that stream and never synchronise (bench / pipelines / torch tensors).
 *  - strings cross the boundary as uint8 arrays + int64 CSR offsets, never char**.
 *  - an alignment ("msa") is rows x cols bytes, row-major, alphabet ACGTN and '-' (any other
 *    byte is folded to 'N' on entry, the same folding getReverseSequence applies, Util.py:1635).
 *  - thread-safety: one hite_ctx per thread / per GPU; no global HIP state before the first call.
 */
#ifndef HITE_GPU_H
#define HITE_GPU_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A row window handed to the star alignment may BEGIN and END with runs of PAD bytes (never the centre, row 0 of a candidate): bytes
 * with bit 5 set -- HITE_ROW_PAD ('.'), which matches nothing (1 per column against a centre base instead of 3 per base of a gap), or a
 * base in LOWER CASE (a, c, g, t), which matches its base.  Genome and candidate bytes are upper case: no pad is ever mistaken.  Pads
 * take part in the pairwise alignment and are removed afterwards -- centre positions aligned to them become gaps of the row, the ops
 * refer to the row without them.  hite_flank_region_align_clip[_dev] pads the rows of copy records in the reference's coordinates with
 * the centre's own bases in lower case (see there). */
#define HITE_ROW_PAD 0x2e   /* '.' */
#define HITE_IS_ROW_PAD(c) (((c) & 0x20u) != 0)

#define HITE_OK 0
#define HITE_EINVAL (-1)   /* bad argument */
#define HITE_ENOMEM (-2)   /* host or device allocation failed */
#define HITE_EHIP (-3)     /* HIP runtime error (hite_last_error() has the text) */
#define HITE_ECAP (-4)     /* caller-provided output capacity too small */
#define HITE_ENODEV (-5)   /* no usable GPU */

#define HITE_TE_TIR 0      /* judge_boundary_v5  Util.py:9145 */
#define HITE_TE_HELITRON 1 /* judge_boundary_v6  Util.py:9821 */
#define HITE_TE_NON_LTR 2  /* judge_boundary_v9  Util.py:9483 */

#define HITE_INFO_NONE 0   /* ''    */
#define HITE_INFO_NB 1     /* 'nb'  : anchor not found */
#define HITE_INFO_FL1 2    /* 'fl1' : <= 1 full-length row */
#define HITE_INFO_EXC 3    /* the reference would raise a Python exception on this input */

typedef struct hite_ctx hite_ctx;

/* One boundary call = the tuple judge_boundary_v5/v6/v9 return (is_TE, info, cons_seq, row_num)
 * plus the final boundary columns the reference only prints in debug mode.  32 bytes; this is also
 * the record all-gathered between GPUs. */
typedef struct hite_call {
    int32_t is_te;
    int32_t info;
    int32_t row_num;
    int32_t bstart;   /* final_boundary_start column (-1 if none) */
    int32_t bend;     /* final_boundary_end column   (-1 if none) */
    int32_t cons_len; /* consensus length */
    int64_t cons_off; /* offset of the consensus inside the caller's cons pool */
} hite_call;

/* ---- context ------------------------------------------------------------------------- */
int hite_ctx_create(int device_id, hite_ctx **out);
void hite_ctx_destroy(hite_ctx *ctx);
const char *hite_last_error(hite_ctx *ctx);
int hite_version(void);

/* ---- genome residency (2-bit bases + 1-bit non-ACGT mask in HBM) ------------------------
 * Replaces the per-stage `read_fasta(reference)` (Util.py:1650, called at :8073, :4616, :4788).
 * seq = all contigs concatenated (upper-cased ASCII), contig_off[n_contigs+1] CSR. */
int hite_genome_pack(hite_ctx *ctx, const uint8_t *seq, const int64_t *contig_off, int32_t n_contigs);
/* same, from an ASCII buffer already in device memory (async on `stream`) */
int hite_genome_pack_dev(hite_ctx *ctx, const uint8_t *d_seq, const int64_t *contig_off_host, int32_t n_contigs,
                         void *stream);
int64_t hite_genome_bases(hite_ctx *ctx);
/* Byte order of the contig names, each followed by ':' (rank[i] = position of contig i): tools/ready_for_MSA.sh keeps the 100
 * longest windows and breaks length ties by window NAME "<contig>:<start>-<end>(<strand>)" (`sort -nk 2 -r` on the .fai,
 * POSIX locale; Util.py:8110, 10410).  Optional: without it contigs compare by their index.  Cleared by hite_genome_pack. */
int hite_set_contig_order(hite_ctx *ctx, const int32_t *rank, int32_t n);
/* N-mask intervals (contig id, 1-based inclusive, clamped into the contig) of the resident genome: the masking step of
 * mask_genome_intactTE (Util.py:6389-6431).  A minimizer index built before the call does not see the mask. */
int hite_genome_mask(hite_ctx *ctx, int64_t n, const int32_t *contig, const int64_t *start1, const int64_t *end1);

/* ---- flank-window gather --- Util.py:8095-8124 (inside flank_region_align_v5) -------------
 * copy i = (contig[i], start1[i], end1[i]) 1-based inclusive, minus[i] = strand '-'.
 * Pass 1 (sizes): out_len[i] = window length, 0 if the reference skips the copy (off-contig
 *   :8103 or shorter than 100 :8108); trunc_len[i] = 1000 if window > 1000 (:8117) else 0.
 * Pass 2 (fill): windows written at out_off[i] (caller's exclusive scan of out_len), the
 *   first500+last500 form at trunc_off[i] when trunc_len[i] != 0 (trunc_* may be NULL). */
int hite_flank_sizes(hite_ctx *ctx, int64_t n, const int32_t *contig, const int64_t *start1, const int64_t *end1,
                     int32_t flank, int64_t *out_len, int64_t *trunc_len);
int hite_flank_sizes_dev(hite_ctx *ctx, int64_t n, const int32_t *d_contig, const int64_t *d_start1,
                         const int64_t *d_end1, int32_t flank, int64_t *d_out_len, int64_t *d_trunc_len, void *stream);
int hite_flank_gather(hite_ctx *ctx, int64_t n, const int32_t *contig, const int64_t *start1, const int64_t *end1,
                      const uint8_t *minus, int32_t flank, const int64_t *out_off, uint8_t *out,
                      const int64_t *trunc_off, uint8_t *trunc_out);
int hite_flank_gather_dev(hite_ctx *ctx, int64_t n, const int32_t *d_contig, const int64_t *d_start1,
                          const int64_t *d_end1, const uint8_t *d_minus, int32_t flank, const int64_t *d_out_off,
                          uint8_t *d_out, const int64_t *d_trunc_off, uint8_t *d_trunc_out, void *stream);

/* ---- sparse-column removal --- remove_sparse_col_in_align_file  Util.py:10344-10405 --------
 * batch of n alignments: msa_off[i] = byte offset of alignment i, rows[i] x cols[i].
 * Writes the cleaned alignment IN THE SAME SLOT of `out` (offset msa_off[i], row stride
 * new_cols[i]) and new_cols[i]. */
int hite_sparse_cols(hite_ctx *ctx, int32_t n, const uint8_t *msa, const int64_t *msa_off, const int32_t *rows,
                     const int32_t *cols, uint8_t *out, int32_t *new_cols);
/* device-resident: d_col_off = exclusive scan of cols (n+1), total_cols = its last element;
 * d_out must not alias d_msa. */
int hite_sparse_cols_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_msa, const int64_t *d_msa_off,
                         const int32_t *d_rows, const int32_t *d_cols, const int64_t *d_col_off, int64_t total_cols,
                         uint8_t *d_out, int32_t *d_new_cols, void *stream);

/* ---- tandem-repeat masking --- run_remove_TR / filter_tandem_repeats  Util.py:2855-2874, 4672-4697 (a-2) ---
 * The reference runs `trf <file> 2 7 7 80 10 50 500 -f -d -m -h` and continues with the .mask FASTA.  This build's own stage
 * (definition: oracle/hite_oracle_trf.c; TRF is third-party, parity unpinned): stretches of the resident genome that align
 * with themselves max_period (<= 500) or fewer bases further on with score >= 50 under match 2 / mismatch 7 / indel 7 and
 * at least 1.85 copies become N for every later stage.  mask_bits_host (optional, (n_bases + 31) / 32 words): bit (i & 31) of
 * word (i >> 5) = base i of the concatenated contigs is masked.  masked_bases_out (optional): their number. */
int hite_tr_mask(hite_ctx *ctx, int32_t max_period, uint32_t *mask_bits_host, int64_t *masked_bases_out);

/* ---- column vote --- col_base_map  Util.py:9251-9266 (a-15) --------------------------------
 * counts_out[col_off[i] + c][6] = per-column counts of A,C,G,T,N,'-' over ALL rows of
 * alignment i, with col_off[i] = caller's exclusive scan of cols. */
int hite_column_vote(hite_ctx *ctx, int32_t n, const uint8_t *msa, const int64_t *msa_off, const int32_t *rows,
                     const int32_t *cols, const int64_t *col_off, int32_t *counts_out);

/* ---- boundary search --- search_boundary_homo_v3 / _v4  Util.py:8887-9143 / 8556-8824 --------
 * One search per alignment: pos[i], side[i] (0 'start', 1 'end'), thr[i]; variant 3 or 4.
 * For v4 int_thr/out_thr as the caller passes them; valid_out (v4's first tuple element)
 * may be NULL for v3.  win_in = 20, win_out = 10 in every reference call. */
int hite_boundary_search(hite_ctx *ctx, int32_t n, const uint8_t *msa, const int64_t *msa_off, const int32_t *rows,
                         const int32_t *cols, const int32_t *pos, const int32_t *side, const double *thr,
                         const double *int_thr, const double *out_thr, int32_t variant, int32_t win_in,
                         int32_t win_out, int32_t *boundary_out, int32_t *valid_out);

/* ---- judge --- judge_boundary_v5 / v6 / v9 (result_type 'cons') ------------------------------
 * te_type selects the variant for the whole batch.  cand = candidate sequences (cur_seq), CSR
 * cand_off[n+1].  calls[i].cons_off = msa col prefix: cons pool must hold sum(cols[i] + 8) bytes
 * and candidate i's consensus is written at cons_off = sum_{j<i}(cols[j] + 8). */
int hite_judge(hite_ctx *ctx, int32_t te_type, int32_t plant, int32_t n, const uint8_t *msa, const int64_t *msa_off,
               const int32_t *rows, const int32_t *cols, const uint8_t *cand, const int64_t *cand_off,
               hite_call *calls, uint8_t *cons);
int hite_judge_dev(hite_ctx *ctx, int32_t te_type, int32_t plant, int32_t n, const uint8_t *d_msa,
                   const int64_t *d_msa_off, const int32_t *d_rows, const int32_t *d_cols, const uint8_t *d_cand,
                   const int64_t *d_cand_off, const int64_t *d_col_off, int32_t max_cols, int32_t max_rows,
                   hite_call *d_calls, uint8_t *d_cons, void *stream);

/* ---- TSDsearch_v5  Util.py:2460-2492 (batch of rows) --------------------------------------- */
int hite_tsd_search(hite_ctx *ctx, int32_t n, const uint8_t *rows_bytes, const int64_t *row_off,
                    const int32_t *bstart, const int32_t *bend, int32_t plant, int32_t *tsd_len_out,
                    uint8_t *left_out /* n x 16 */, uint8_t *right_out /* n x 16 */);

/* ---- non-LTR candidate preparation --- search_polyA_TSD  Util.py:10915-11007 (get_candidate_non_LTR :11009) -------------
 * batch of flanked repeats (CSR): polyA / tandem tail near the 3' end (or polyT / tandem head), then an 8-20 bp TSD
 * (<= 1 edit) within win5 (<= 25) of the 5' end.  out: 6 x int64 per sequence = {found_TSD, direct (0 none, 1 '+', 2 '-'),
 * TSD start, TSD length, lo, hi}: non_ltr_seq = seq[lo:hi], reverse-complemented by the caller when direct == 2. */
int hite_nonltr_prep(hite_ctx *ctx, int32_t n, const uint8_t *seqs, const int64_t *seq_off, int32_t flank, int32_t win5,
                     int64_t *out);

/* ---- LTR flank-frame vote (vendored FiLTR) --- judge_left_frame_LTR / judge_right_frame_LTR
 * bin/FiLTR-main/src/Util.py:9327 / :9175 -----------------------------------------------------------------------------
 * n matrices of rows[i] x cols[i] bytes at off[i] (the left OR right frames of the copies of one LTR candidate, what the
 * reference reads from the '.matrix' file, one column of the tab-separated pair), side 0 = left frames (start column
 * flank-1, walk left, tolerance 5), side 1 = right frames (start column 0, walk right, tolerance 20).
 * ok_out[i] = 1 / 0 (is the candidate kept), boundary_out[i] = new boundary column or -1.  cols[i] <= 1024, flank <= cols[i]. */
int hite_ltr_frame(hite_ctx *ctx, int32_t n, const uint8_t *frames, const int64_t *off, const int32_t *rows,
                   const int32_t *cols, int32_t flank, int32_t window, int32_t side, int32_t *ok_out, int32_t *boundary_out);

/* ---- FMEA --- get_longest_repeats_v4 + process_all_seqs  Util.py:4122-4400, 4529-4569 -------------
 * n HSPs in blast6 file order (cols 0,1,6,7,8,9 of -outfmt 6): qseg/sseg = ids of the 'chr$offset'
 * segment names, 1-based inclusive coordinates (reverse hits have ss > se); seg_chrom/seg_off give the
 * chromosome id and the offset of each segment (nseg <= 4096).  skip_gap = fixed_extend_base_threshold.
 * Output = the keys 'chr:start-end' of the reference's longest_repeats dict, in insertion order, as
 * (chrom id, start, end); *n_out is set even when HITE_ECAP is returned.
 * HITE_EINVAL for a zero-length HSP (the reference raises ZeroDivisionError, :4270) or coordinates
 * outside [0, 2^31) / chain spans >= 1.31 Mbp (the de-duplication key is packed into 64 bits). */
int hite_fmea_chain(hite_ctx *ctx, int64_t n, const int32_t *qseg, const int32_t *sseg, const int64_t *qs,
                    const int64_t *qe, const int64_t *ss, const int64_t *se, int32_t nseg, const int32_t *seg_chrom,
                    const int64_t *seg_off, int64_t skip_gap, int64_t max_len, int64_t cap, int32_t *out_chrom,
                    int64_t *out_start, int64_t *out_end, int64_t *n_out);
/* same, HSP arrays resident on the device (e.g. straight from hite_seed_allvsall_dev: no PCIe round trip of the table) */
int hite_fmea_chain_dev(hite_ctx *ctx, int64_t n, const int32_t *d_qseg, const int32_t *d_sseg, const int64_t *d_qs,
                        const int64_t *d_qe, const int64_t *d_ss, const int64_t *d_se, int32_t nseg, const int32_t *seg_chrom,
                        const int64_t *seg_off, int64_t skip_gap, int64_t max_len, int64_t cap, int32_t *out_chrom,
                        int64_t *out_start, int64_t *out_end, int64_t *n_out);

/* ---- copy clustering of a blast6 HSP table --- get_query_copies  Util.py:6828-7030 (+ get_copies_v1 :7032-7060) -----
 * n HSPs in file order (host arrays): query id in [0, nq), subject id in [0, ns), 1-based inclusive coordinates
 * (s_start > s_end = minus strand), identity column (may be NULL: only its equality between otherwise identical
 * lines matters).  Per query: cluster per (subject in order of first appearance, strand), longest chain per cluster
 * with the 200 bp gap rules (qthr / sthr), longest first, keep chain_len / qlen >= qcov (and, if scov > 0,
 * subject_span / slen[sid] >= scov), de-duplicate on (subject, start, end), stop after max_copy + 1 (<= 254).
 * Output CSR copy_first[nq + 1] into (o_sid, o_s <= o_e, o_len, o_minus); *n_out = total; HITE_ECAP if total > cap.
 * HITE_EINVAL on ids / coordinates out of range. */
int hite_query_copies(hite_ctx *ctx, int64_t n, const int32_t *qid, const int32_t *sid, const int64_t *qs, const int64_t *qe,
                      const int64_t *ss, const int64_t *se, const double *ident, int32_t nq, const int64_t *qlen, int32_t ns,
                      const int64_t *slen, double qcov, double scov, int64_t qthr, int64_t sthr, int32_t max_copy, int64_t cap,
                      int64_t *copy_first, int32_t *o_sid, int64_t *o_s, int64_t *o_e, int64_t *o_len, uint8_t *o_minus,
                      int64_t *n_out);

/* ---- every chain of every cluster --- the chaining core of FMEA (Util.py:10452-10645) and of
 * get_full_length_copies_from_blastn_v1 (:5907-6105; with generate_full_length_out_v1 :6288 what mask_genome_intactTE :6389
 * consumes), i.e. get_longest_repeats_v4's core without its de-duplication.  n HSPs in file order (host arrays; the caller
 * drops the lines its function skips), query id in [0, nq), subject id in [0, ns), 1-based inclusive coordinates (s_start >
 * s_end = minus strand).  qgap[q] = the query's skip_gap: an HSP joins a cluster / extends a chain while the distance is
 * < qgap[q] (FMEA: fixed_extend_base_threshold for every query; full-length copies: ceil(len(query) * threshold) -- an
 * integer distance is below a real gap exactly when it is below its ceiling).  Output CSR chain_first[nq + 1] over the chains
 * of each query in the reference's order (subjects by first appearance, forward clusters before reverse ones, clusters in
 * sweep order, chains by their first fragment): subject id, (prev_query_start, prev_query_end, prev_subject_start,
 * prev_subject_end) as the reference holds them (1-based, s_start > s_end on the minus strand), cur_extend_num.
 * *n_out = total; HITE_ECAP if total > cap; HITE_EINVAL on ids / coordinates out of range. */
int hite_chain_all(hite_ctx *ctx, int64_t n, const int32_t *qid, const int32_t *sid, const int64_t *qs, const int64_t *qe,
                   const int64_t *ss, const int64_t *se, int32_t nq, int32_t ns, const int64_t *qgap, int64_t cap,
                   int64_t *chain_first, int32_t *o_sid, int64_t *o_qs, int64_t *o_qe, int64_t *o_ss, int64_t *o_se,
                   int32_t *o_next, int64_t *n_out);

/* ---- library de-duplication (panHiTE merge) --- the arithmetic between the external tools of deredundant_for_LTR_v5 -----
 * hite_lib_chain: process_blast_results_in_chunks + process_chunk + extend_fragments (Util.py:12146-12200, 11958-12003,
 * 11869-11944).  n blast6 lines of a library-vs-itself search in file order (host arrays), ids into seq_len[nseq],
 * 1-based inclusive coordinates; a line with qid == sid, qs == ss, qe == se is skipped.  chunk_size as in the reference
 * (<= 0: one chunk): a chunk is closed by a kept line whose 1-based number is a multiple of chunk_size, chains never
 * cross chunks.  skip_gap = seq_len[query] * (1 - threshold).  Output records in the order of the reference's chunk
 * files (chunk, query by first appearance, subject by first appearance, forward then reverse, creation order):
 * (o_chunk, o_q, o_qs = q_start - 1, o_qe, o_s, o_ss = s_start - 1, o_se); *n_out records, HITE_ECAP if > cap. */
int hite_lib_chain(hite_ctx *ctx, int64_t n, const int32_t *qid, const int32_t *sid, const int64_t *qs, const int64_t *qe,
                   const int64_t *ss, const int64_t *se, int32_t nseq, const int64_t *seq_len, double threshold, int64_t chunk_size,
                   int64_t cap, int32_t *o_chunk, int32_t *o_q, int64_t *o_qs, int64_t *o_qe, int32_t *o_s, int64_t *o_ss,
                   int64_t *o_se, int64_t *n_out);
/* hite_lib_cluster: cluster_sequences_from_chunks (Util.py:12067-12115) on the records above (host code: the greedy
 * pass is sequential by definition).  Cluster c = members[cl_first[c] .. cl_first[c+1]): its query, then the subjects in
 * the order they joined (the reference keeps a Python set: membership is the contract).  *n_cl clusters. */
int hite_lib_cluster(int64_t nrec, const int32_t *chunk, const int32_t *q, const int64_t *qs, const int64_t *qe, const int32_t *s,
                     const int64_t *ss, const int64_t *se, int32_t nseq, const int64_t *seq_len, double threshold, int64_t cap_cl,
                     int64_t cap_mem, int64_t *cl_first, int32_t *members, int64_t *n_cl);
/* hite_msa_consensus: cons_from_mafft_v1 (Util.py:12515-12566) for a batch of alignments.  Alignment a = rows[a] x cols[a]
 * bytes, row-major, at mats + mat_off[a] (mat_off[nmat] = total); a column contributes its most frequent non-gap byte if
 * that count > rows[a] / 2.  Consensus a is written at cons + out_off[a] (reserve cols[a] bytes), length cons_len[a]. */
int hite_msa_consensus(hite_ctx *ctx, int32_t nmat, const int32_t *rows, const int64_t *cols, const int64_t *mat_off,
                       const uint8_t *mats, const int64_t *out_off, uint8_t *cons, int64_t *cons_len);

/* ---- LTR frames of the vendored FiLTR --- get_both_ends_frame  bin/FiLTR-main/src/Util.py:1401-1497 (+ :1341-1399) -----
 * Batch of alignments (rows[a] x cols[a] bytes at msa + msa_off[a], upper case) with the terminal sequence of each
 * (cand + cand_off[a] .. cand_off[a+1]).  Per alignment: anchors from the first row carrying both 20-mers within two
 * edits, FiLTR's sparse-column rule (more than R/2 gaps, anchor columns always kept), then per row the `.matrix` line
 * (left frame | right frame, 2 * flank bytes at frames + frame_off[a] + r * 2 * flank, '-' padded) and the full-length row
 * (left frame + cleaned[new_start:new_end] + right frame at full + full_off[a] + r * (2 * flank + cols[a]), full_cols[a]
 * bytes used).  new_pos[2a], new_pos[2a+1] = anchor columns in cleaned coordinates.  status[a]: 0 ok, 1 boundary not
 * found (the reference returns None, None), 2 both anchors on the same column (not restated). */
int hite_ltr_both_ends(hite_ctx *ctx, int32_t n, const uint8_t *msa, const int64_t *msa_off, const int32_t *rows, const int32_t *cols,
                       const uint8_t *cand, const int64_t *cand_off, int32_t flank, uint8_t *frames, const int64_t *frame_off,
                       uint8_t *full, const int64_t *full_off, int32_t *full_cols, int32_t *new_pos, int32_t *status);

/* ---- k-mer TSD seed matching --- search_confident_tir_v4  Util.py:7734-7845 -------------------------
 * batch of flanked candidates (CSR); the raw boundaries are (flank+1, len-flank), 1-based, as
 * search_confident_tir_batch_v1 passes them (Util.py:6550), tsd_search_distance = flank (<= 63).
 * Per candidate up to 100 records (tsd_len, tir_start, tir_end, distance), 0-based inclusive, in the
 * canonical order (distance, tir_start, tir_end, tsd_len) that replaces the reference's
 * PYTHONHASHSEED-dependent tie order; rec_out is n x 100 x 4 int32, cnt_out[n] (-1: window too long). */
int hite_tsd_kmer(hite_ctx *ctx, int32_t n, const uint8_t *seqs, const int64_t *seq_off, int32_t flank, int32_t plant,
                  int32_t *rec_out, int32_t *cnt_out);
int hite_tsd_kmer_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_seqs, const int64_t *d_seq_off, int32_t flank,
                      int32_t plant, int32_t *d_rec_out, int32_t *d_cnt_out, void *stream);

/* ---- terminal inverted repeats --- run_itrsearch  Util.py:216-224 (third-party ELF tools/itrsearch, `-i 0.7 -l 7`) ----------
 * The filter of search_confident_tir_batch_v1 (Util.py:6556-6587: first 40 + last 40 bases of every k-mer TSD variant; a variant
 * without a terminal inverted repeat is dropped, "Length itr=" feeds filter_dup_itr_v3) and of remove_no_tirs (Util.py:13897-13920:
 * whole low-copy sequences).  In-tree stage; definition oracle/hite_oracle_itr.c, read from the tool's disassembly and pinned to
 * the tool's own output (tests/golden/itr_search.json.gz).  Per sequence (CSR batch; bytes outside ACGT are N, and N scores as a
 * match against anything, as in the tool): seq1 = first h bases, seq2 = reverse complement of the last h, h = min(500, len / 2), or
 * min(len, end_len) when end_len > 0 (the record `s[:end_len] + s[-end_len:]` composed on the device); affine-gap extension
 * alignment from (0,0) with a free end (match / mismatch / gap_open / gap_extend: the tool's defaults are 10 / 16 / 32 / 32 and the
 * reference passes none of them).
 * out[8k ..] = { score, end in seq1, end in seq2, equal bases, aligned columns, found, "Length itr=" (end1 - 1; -1 without an
 * alignment), flags }; found = min_len <= end1 && equal / aligned >= min_identity (binary64), what makes the tool write the record
 * to <input>.itr.  flags bit 3: the sequence needs more than max_h (the _dev caller's promise) and was skipped. */
int hite_itr_search(hite_ctx *ctx, int64_t n, const uint8_t *seqs, const int64_t *seq_off, int32_t end_len, double min_identity,
                    int32_t min_len, int32_t match, int32_t mismatch, int32_t gap_open, int32_t gap_extend, int32_t *out);
/* device-resident, asynchronous on `stream`; max_h = upper bound of h over the batch (sizes LDS / the scratch slots; <= 64 keeps
 * everything in LDS).  d_seqs must be readable up to d_seq_off[n]. */
int hite_itr_search_dev(hite_ctx *ctx, int64_t n, const uint8_t *d_seqs, const int64_t *d_seq_off, int32_t end_len, int32_t max_h,
                        double min_identity, int32_t min_len, int32_t match, int32_t mismatch, int32_t gap_open, int32_t gap_extend,
                        int32_t *d_out, void *stream);

/* ---- copy finding: this build's GPU-native stage where the reference runs the external
 * `minimap2 -ax map-ont -N 300 -p 0.2` + SAM filtering (get_full_length_copies_minimap2, Util.py:7933-8030;
 * third-party, unpinned -> parity is pinned against the build's own CPU twin, oracle/hite_oracle_copies.c,
 * whose header holds the definition: (w=10,k=15) minimizers, diagonal clustering, >= 3 anchors spanning
 * >= 80 % of the candidate and covering a candidate span >= 95 % of the genome span they cover (the target-coverage
 * filter of get_copies_minimap2), boundaries extrapolated from the extreme anchors, <= 300 copies per candidate
 * ordered by anchors).  Needs a packed genome (< 4 Gbp).  The index handle (*state_io, initially NULL)
 * is built once per genome; free it with hite_copy_index_release.
 * Output = the copy table get_full_length_copies_minimap2 returns, as the CSR that
 * hite_flank_region_align consumes (1-based inclusive coordinates).  n_cand < 2^19 per call.
 * _dev: the returned device arrays live in the index state's arena until the next call. */
int hite_copy_index_build(hite_ctx *ctx, void **state_io, void *stream);
/* A build keeps the genome's minimizers (tiles of 2048 window starts, in the handle's build arena); the next build on the same handle and
 * the same packed genome -- the restricted index of hite_find_copies_restricted, then the full index behind hite_genome_mask: stage 3.1's
 * prev_TE step -- computes only the tiles a mask call has touched since, and gives the index a build from scratch gives (the kept tiles
 * are void after hite_genome_pack* / hite_tr_mask; HITE_KEEP_MINIMIZERS=0 keeps nothing).  hite_copy_index_forget drops them: the next
 * build starts from the genome (what a benchmark that times repeated builds of one genome wants).  state NULL: no-op. */
int hite_copy_index_forget(void *state);
/* which interval the copy records of hite_find_copies[_dev] carry (process-wide DEFAULT, for contexts without a setting of their own): 1 = the ALIGNED interval, reference_start + 1 ..
 * reference_end exactly as get_copies_minimap2 reports it (Util.py:8026) -- the default since round 5, with the clipped candidate
 * bases handed on beside the records (hite_copy_clips) --; 0 = the interval of the WHOLE candidate, the ends that the extension clipped
 * extrapolated on the diagonal (the default of rounds 2-4); -1 = take the setting from the environment again
 * (HITE_COPY_INTERVAL=aligned | whole).  Both are twin-pinned (orc_find_copies_config). */
int hite_copy_config(int32_t aligned_interval);
/* the same setting for ONE context (what the library's per-context thread safety covers): 1 / 0 as above, -1 = follow the process-wide
 * setting of hite_copy_config again (the state of a new context) */
int hite_copy_config_ctx(hite_ctx *ctx, int32_t aligned_interval);
void hite_copy_index_release(void *state);
/* sizes of the last hite_find_copies[_dev] call on this index (diagnostics / roofline accounting):
 * out = {candidate minimizers, index hits, diagonal clusters, copies before the 300-per-candidate cap} */
int hite_copy_stats(void *state, int64_t out[4]);
/* the same four numbers + {chains with a long end to extend, the other chains, DP columns of the end extension, 0} */
int hite_copy_stats_ext(void *state, int64_t out[8]);

/* ---- all-vs-all seeding (stage 3.1) --- where the reference runs blastn of every 1 Mbp segment file against every
 * file: process_blast_alignments / sequence2sequenceBlastn  Util.py:4724-4780, 4068-4091 (rmblast is third-party and
 * absent: this is the build's own stage, pinned against its CPU twin orc_seed_allvsall) ---------------------------------
 * Uses the minimizer index of the packed genome (*state_io as for hite_find_copies; built on demand).  Output = the HSP
 * table hite_fmea_chain consumes (cols 0,1,6,7,8,9 of -outfmt 6): ids of the 'chr$offset' segments of seg_len bases
 * (hite_seed_segments gives the table, split_genome_chunks.py:41-52), 1-based inclusive coordinates inside the segment,
 * ss > se for reverse-strand hits; ordered by (query segment, subject segment).  *n_out is set even on HITE_ECAP.
 * max_anchors bounds the device memory of the anchor sort (24 B per anchor).  stats_out (may be NULL) =
 * {seeds, anchors, clusters, records}. */
int hite_seed_allvsall_dev(hite_ctx *ctx, void **state_io, int64_t seg_len, int64_t max_anchors, int32_t **d_qseg, int32_t **d_sseg,
                           int64_t **d_qs, int64_t **d_qe, int64_t **d_ss, int64_t **d_se, int64_t *n_out, int64_t *stats_out);
int hite_seed_segments(hite_ctx *ctx, int64_t seg_len, int32_t cap, int32_t *seg_chrom, int64_t *seg_off, int32_t *nseg_out);
/* Sharding of the all-vs-all stage over the ranks of a node (one process per GPU, the packed genome replicated):
 * after hite_seed_shard(ctx, rank, world) the seeding calls of this context emit the HSPs of rank's share only -- the
 * anchors whose (strand, diagonal) falls into the rank's range; clusters never straddle two ranges, so the tables of the
 * ranks, concatenated in rank order and stably sorted by (query segment, subject segment), are the unsharded table record
 * for record.  world <= 1 restores the whole.  hite_amd/dist.py routes the records to the owners of the query files
 * (all-to-all), runs hite_fmea_chain per file there and all-gathers the interval lists. */
int hite_seed_shard(hite_ctx *ctx, int32_t rank, int32_t world);
int hite_seed_allvsall(hite_ctx *ctx, void **state_io, int64_t seg_len, int64_t max_anchors, int64_t cap, int32_t *qseg,
                       int32_t *sseg, int64_t *qs, int64_t *qe, int64_t *ss, int64_t *se, int64_t *n_out, int64_t *stats_out);
int hite_find_copies(hite_ctx *ctx, void **state_io, int32_t n_cand, const uint8_t *cand, const int64_t *cand_off,
                     int64_t cap, int32_t *copy_first, int32_t *contig, int64_t *start1, int64_t *end1, uint8_t *minus,
                     int32_t *anchors, int64_t *n_out);
/* _dev PRECONDITION: d_cand must be readable for 16 bytes beyond cand_bytes (the end extension fetches candidate words ahead
 * of use); hite_find_copies pads its upload, a caller that owns the device buffer allocates cand_bytes + 16. */
int hite_find_copies_dev(hite_ctx *ctx, void *state, int32_t n_cand, const uint8_t *d_cand, const int64_t *d_cand_off,
                         int64_t cand_bytes, int32_t **d_copy_first, int64_t *n_copies, int32_t **d_contig,
                         int64_t **d_start1, int64_t **d_end1, uint8_t **d_minus, int32_t **d_anchors, void *stream);
/* hite_find_copies_dev for ONE candidate set on a genome whose index nothing else will use (stage 3.1 masks the genome with the
 * previous TE library -- Util.py:6021-6081 `mask_genome_intactTE`, the reference's blastn of the library against the chunk -- and only then
 * indexes the MASKED genome): the index is built for these candidates only -- the genome pass keeps the minimizers whose hash one of
 * their minimizers looks up, so the hash sort and the directory run on those few entries.  Every entry the look-up would read is
 * there, with the same run lengths and order: the copy table equals hite_find_copies_dev's on the full index (tests assert it).
 * *state_io as for hite_find_copies (created when null); the handle is left flagged restricted, and hite_find_copies[_dev] /
 * hite_seed_allvsall[_dev] rebuild the full index before they use it.  Same _dev PRECONDITION. */
int hite_find_copies_restricted_dev(hite_ctx *ctx, void **state_io, int32_t n_cand, const uint8_t *d_cand, const int64_t *d_cand_off,
                                    int64_t cand_bytes, int32_t **d_copy_first, int64_t *n_copies, int32_t **d_contig,
                                    int64_t **d_start1, int64_t **d_end1, uint8_t **d_minus, int32_t **d_anchors, void *stream);
/* host-buffer form (arguments as hite_find_copies) */
int hite_find_copies_restricted(hite_ctx *ctx, void **state_io, int32_t n_cand, const uint8_t *cand, const int64_t *cand_off,
                                int64_t cap, int32_t *copy_first, int32_t *contig, int64_t *start1, int64_t *end1, uint8_t *minus,
                                int32_t *anchors, int64_t *n_out);
/* The clip words of the records of the last hite_find_copies[_dev] / _restricted[_dev] call on this index, in record order: candidate
 * bases the left | right << 16 end extension cut off (minimap2 would soft-clip them; each <= 5 % of the candidate), in the orientation
 * of the genome.  For aligned intervals (the default) hite_flank_region_align_clip[_dev] takes them; zero for whole-candidate
 * intervals (hite_copy_config(0): the clipped bases are inside the interval).  _dev: *d_clip (NULL when the call found nothing) lives as long
 * as the copy table; host form: cap >= the number of records, else HITE_ECAP. */
int hite_copy_clips_dev(void *state, const uint32_t **d_clip, int64_t *n);
int hite_copy_clips(void *state, int64_t cap, uint32_t *clip);

/* ---- star alignment: this build's GPU-native stage where the reference runs the external
 * `mafft --preservecase --quiet --thread 1` (Util.py:10416; third-party, unpinned, absent -> parity unpinned against
 * mafft itself).  Definition of the stage: every row is aligned to the centre (first row of the candidate) by the
 * optimal global alignment under the costs mismatch 1, gap 3 per base (two bases match only when they are the same
 * A, C, G or T; a lower-case a / c / g / t of a ROW is read as its base -- mafft --preservecase compares case-insensitively --
 * BUT a run of bytes with bit 5 set at the BEGINNING or END of a row is a run of pad bytes, see HITE_IS_ROW_PAD above: lower
 * case at the ends of a row is RESERVED for pads and leaves the alignment as gaps.  A caller whose windows may begin or end with
 * soft-masked lower-case sequence upper-cases them first, as hite_amd/util.py does; the centre is compared as it is, upper
 * case), canonical traceback diagonal > up > left -- the textbook full-matrix
 * programme of oracle/hite_oracle_nw.c; insertion blocks are left-justified.  The device computes it with a banded
 * bit-parallel aligner that CERTIFIES its result (twin: oracle/hite_oracle_msa.c, byte-exact): a certified row is the
 * alignment of the definition, a row without certificate is a valid alignment whose cost bounds the optimum from above.
 * hite_align_config: exact_cap = 0 (fast: band of 128 centre rows only), 8 / 16 / 32 = widest band (x 32 rows) that
 * is tried to obtain a certificate (default 8, or the environment variable HITE_ALIGN_EXACT).
 * hite_align_stats: out8 = pairs, certified, kept from a band wider than 128 rows, wide fall-backs, rows dropped,
 * sum of the costs, sum of the row lengths (columns), exact_cap -- accumulated over the calls since the last reset.
 * windows of candidate c = rows row_first[c] .. row_first[c+1]-1 of the CSR (win, win_off);
 * the first row of each candidate is the centre.  Window length <= 32767.  A row that cannot be aligned (shorter than
 * half the centre, or an insertion / deletion beyond the widest band) is DROPPED: rows_out[c] (may be NULL) = rows of
 * the alignment.
 * hite_star_msa: pass msa_out = NULL to get cols_out / rows_out only; otherwise the rows x cols matrices are
 * written at msa_off_out[c] (16-byte aligned slots) and msa_cap is checked.
 * hite_star_msa_info: same + info_out = 5 int32 per input window (zeros for centres): cost U of the alignment kept,
 * certified, status (0 aligned, else dropped), the certificate's bound k*, band words of the run kept (| 0x100: wide fall-back).
 * _dev: window g starts at d_win_off[g] (16-byte aligned, readable up to the next multiple of 16) and is d_win_len[g]
 * long; d_ops_base[n+1] = exclusive scan of (R_c + 1) * (m_c + 1) (m_c = centre length),
 * ops_elems its last element; d_status[c] != 0 marks a candidate whose alignment failed (wider than 65535 columns;
 * cols_out[c] = 0); d_rows_out (may be NULL) = rows per alignment.  The fill call must follow the align call on the
 * same ctx/stream. */
int hite_align_config(hite_ctx *ctx, int32_t exact_cap);
/* hite_align_lanes: the LONGEST pairs of a call are aligned by kernels that spread one pair over several lanes (4 or 8 lanes
 * = the words of its band in the forward pass; a wavefront whose lanes re-compute 64 strips at once in the traceback), beside
 * the thread-per-pair kernels of the others -- a kernel cannot end before its longest pair's dependent chain, which in a small
 * batch (one rank's share of a sharded run; the reference gives every candidate its own process, Util.py:8141-8147) is the
 * step.  Same recurrence, same records, same ops: which kernel takes a pair never changes a result.  min_cols = -1 (default,
 * or the environment variable HITE_ALIGN_LANES): pairs whose row is longer than 1/160 000 of all pair-columns of the call
 * and at least 512 columns; >= 0: pairs of at least this many columns (0: every pair -- the parity tests); -2: never. */
int hite_align_lanes(hite_ctx *ctx, int32_t min_cols);
int hite_align_stats(hite_ctx *ctx, int64_t *out8, int32_t reset);
int hite_star_msa(hite_ctx *ctx, int32_t n, const uint8_t *win, const int64_t *win_off, const int32_t *row_first,
                  int32_t *cols_out, int32_t *rows_out, int64_t msa_cap, uint8_t *msa_out, int64_t *msa_off_out);
/* hite_star_msa followed by remove_sparse_col_in_align_file in one step (same two-call protocol; cols_out = surviving columns) */
int hite_star_msa_sparse(hite_ctx *ctx, int32_t n, const uint8_t *win, const int64_t *win_off, const int32_t *row_first,
                         int32_t *cols_out, int32_t *rows_out, int64_t msa_cap, uint8_t *msa_out, int64_t *msa_off_out);
int hite_star_msa_info(hite_ctx *ctx, int32_t n, const uint8_t *win, const int64_t *win_off, const int32_t *row_first,
                       int32_t *cols_out, int32_t *rows_out, int32_t *info_out, int64_t msa_cap, uint8_t *msa_out,
                       int64_t *msa_off_out);
/* One call for the host form (the sizes call + the fill call each run the whole pairwise alignment): the alignments come back in
 * a host buffer the library allocates (*msa_out, *msa_bytes_out bytes, alignment i at msa_off_out[i]; release with
 * hite_host_free); sparse != 0: sparse columns removed; info_out (optional): 5 int32 per input row as hite_star_msa_info. */
int hite_star_msa_once(hite_ctx *ctx, int32_t n, const uint8_t *win, const int64_t *win_off, const int32_t *row_first,
                       int32_t sparse, int32_t *cols_out, int32_t *rows_out, int32_t *info_out, uint8_t **msa_out,
                       int64_t *msa_off_out, int64_t *msa_bytes_out);
void hite_host_free(void *p);
int hite_star_msa_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_win, const int64_t *d_win_off,
                      const int32_t *d_win_len, const int32_t *d_row_first, int64_t total_rows, const int64_t *d_ops_base, int64_t ops_elems,
                      int32_t max_win_len, int32_t *d_cols_out, int32_t *d_status, int32_t *d_rows_out, void *stream);
int hite_star_msa_fill_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_win, const int64_t *d_win_off,
                           const int32_t *d_win_len, const int32_t *d_row_first, const int64_t *d_ops_base, const int32_t *d_cols,
                           const int64_t *d_msa_off, uint8_t *d_msa, void *stream);
/* Fused variant used by hite_flank_region_align: alignment + column layout + the column selection of
 * remove_sparse_col_in_align_file (Util.py:10344-10405) in one step, so that only the surviving columns are ever written.
 * d_cols_out = columns of the full alignment (0 = failed), d_new_cols = surviving columns, d_last_extra = fill hint.
 * hite_star_msa_fill_sparse_dev then writes rows x d_new_cols[i] bytes at d_msa_off[i]; the result is byte-identical to
 * hite_star_msa_fill_dev followed by hite_sparse_cols_dev. */
int hite_star_msa_sparse_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_win, const int64_t *d_win_off,
                             const int32_t *d_win_len, const int32_t *d_row_first, int64_t total_rows,
                             const int64_t *d_ops_base, int64_t ops_elems, int32_t max_win_len, int32_t *d_cols_out,
                             int32_t *d_status, int32_t *d_new_cols, int32_t *d_last_extra, int32_t *d_rows_out, void *stream);
int hite_star_msa_fill_sparse_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_win, const int64_t *d_win_off,
                                  const int32_t *d_win_len, const int32_t *d_row_first, const int64_t *d_ops_base,
                                  const int32_t *d_new_cols, const int32_t *d_last_extra, const int64_t *d_msa_off,
                                  uint8_t *d_msa, void *stream);

/* ---- the fine stage in one call --- body of flank_region_align_v5 from the copy table on ----------
 * (Util.py:8095-8194 + run_find_members_v8 :10439 + is_TE_from_align_file :10407).
 * Candidates: cand/cand_off CSR (cur_seq of each candidate).  Copies of candidate c are entries
 * copy_first[c] .. copy_first[c+1]-1 of (contig, start1, end1, minus) -- what
 * get_full_length_copies_minimap2 (:7933) returns, 1-based inclusive.  Needs a packed genome.
 * calls[c] = the tuple is_TE_from_align_file returns for candidate c; consensus bytes are packed
 * into cons (cons_off/cons_len in the record); HITE_ECAP if cons_cap is too small.
 * stats_out (12 x int64, host, optional): [0..3] pass A rows, window bytes, matrix bytes, alignment
 * algorithmic bytes; [4..7] the same for pass B; [8],[9] anti-diagonal steps of pass A / B (x64 = DP
 * cells); [10] consensus bytes kept; [11] cleaned alignment columns judged (both passes).
 * _dev: all pointers are device pointers; *state_io (initially NULL) keeps the arenas between
 * calls so the steady state performs no allocation; free it with hite_pipeline_release. */
int hite_flank_region_align(hite_ctx *ctx, int32_t te_type, int32_t plant, int32_t n_cand, const uint8_t *cand,
                            const int64_t *cand_off, const int32_t *copy_first, int64_t n_copies, const int32_t *contig,
                            const int64_t *start1, const int64_t *end1, const uint8_t *minus, int32_t flank,
                            hite_call *calls, uint8_t *cons, int64_t cons_cap, int64_t *stats_out);
int hite_flank_region_align_dev(hite_ctx *ctx, void **state_io, int32_t te_type, int32_t plant, int32_t n_cand,
                                const uint8_t *d_cand, const int64_t *d_cand_off, const int32_t *d_copy_first,
                                int64_t n_copies, const int32_t *d_contig, const int64_t *d_start1, const int64_t *d_end1,
                                const uint8_t *d_minus, int32_t flank, hite_call *d_calls, uint8_t *d_cons,
                                int64_t cons_cap, int64_t *stats_out, void *stream);
/* The same stage with the clip words of the copy records given.  A copy record in the REFERENCE'S coordinates (reference_start + 1 ..
 * reference_end of the alignment, Util.py:8026, as get_copies_minimap2 hands it to flank_region_align_v5: what hite_find_copies reports
 * by default) covers only the part of the candidate that aligned; `clip` (per record) says how many candidate bases were left out at
 * its two ends -- left | right << 16, in the orientation of the genome (hite_copy_clips[_dev]) -- and the row's window (interval +
 * flanks, Util.py:8110-8125) is padded with pad bytes (HITE_IS_ROW_PAD): the CENTRE's own first / last bases in lower case, which match
 * the centre positions they face -- in front by the left clip (a minus copy: the right one, its window is reverse-complemented) LESS
 * what the centre's own record leaves out on that side (round 6: the row's first base faces centre position clip_row - clip_centre),
 * behind by the other; the centre (the first row kept) is never padded; the first500 + last500 form of a long window (Util.py:8119) is
 * cut from the padded window; the <= 100 rows are chosen by the length of the genome window.  A padded row faces the part of the
 * centre its copy was found with, so its path stays on the diagonal; the pads leave the alignment as gaps of the row (where mafft,
 * which does not charge terminal gaps like internal ones, leaves such a row unaligned).
 * clip == NULL (and hite_flank_region_align[_dev], which have no such argument) -- the reference's own 5-tuples, real minimap2 records:
 * the clip words are ESTIMATED from the sequences (round 6, clip_probe_kernel; definition orc_clip_probe in
 * oracle/hite_oracle_copies.c): the first 21 bases of the record's interval, read on the candidate's strand, are laid on the candidate
 * at every offset 0 .. |cand| / 20 + 32; the offset with the fewest mismatches (the smallest on ties) is the left clip when it has <= 5
 * mismatches, else the next 21 bases are tried, else 0; the right clip the same from the other end.  A whole-candidate record probes
 * to 0 | 0.  Nothing is inferred from where the caller's arrays live (round 5 recognised the finder's own device table by its address).
 * Rows cut from aligned intervals WITHOUT pads are aligned globally at 3 per gap base against a centre that is clip_l + clip_r bases
 * longer: on config C2 8 633 rows left the band and TE calls fell by a fifth (profiles/r05_scale_tests.txt, last line). */
/* the estimate by itself (host buffers, arguments as hite_flank_region_align's table): clip_out[k] = left | right << 16 in the orientation
 * of the genome, what a call without clip words pads the rows by */
int hite_clip_probe(hite_ctx *ctx, int32_t n_cand, const uint8_t *cand, const int64_t *cand_off, const int32_t *copy_first,
                    int64_t n_copies, const int32_t *contig, const int64_t *start1, const int64_t *end1, const uint8_t *minus,
                    uint32_t *clip_out);
int hite_flank_region_align_clip(hite_ctx *ctx, int32_t te_type, int32_t plant, int32_t n_cand, const uint8_t *cand,
                                 const int64_t *cand_off, const int32_t *copy_first, int64_t n_copies, const int32_t *contig,
                                 const int64_t *start1, const int64_t *end1, const uint8_t *minus, const uint32_t *clip, int32_t flank,
                                 hite_call *calls, uint8_t *cons, int64_t cons_cap, int64_t *stats_out);
int hite_flank_region_align_clip_dev(hite_ctx *ctx, void **state_io, int32_t te_type, int32_t plant, int32_t n_cand,
                                     const uint8_t *d_cand, const int64_t *d_cand_off, const int32_t *d_copy_first,
                                     int64_t n_copies, const int32_t *d_contig, const int64_t *d_start1, const int64_t *d_end1,
                                     const uint8_t *d_minus, const uint32_t *d_clip, int32_t flank, hite_call *d_calls,
                                     uint8_t *d_cons, int64_t cons_cap, int64_t *stats_out, void *stream);
void hite_pipeline_release(void *state);

/* ---- per-stage profiling: HIP events recorded on the launch stream around each kernel of the
 * pipeline (stage names = kernel names).  ms_total / launches accumulate since the last reset. */
int hite_profile_enable(hite_ctx *ctx, int on);
int hite_profile_reset(hite_ctx *ctx);
int hite_profile_count(hite_ctx *ctx);
int hite_profile_get(hite_ctx *ctx, int idx, char *name_out, double *ms_total, int64_t *launches);

/* copy a device range returned by a _dev entry point to the host (synchronises the device) */
int hite_memcpy_d2h(void *dst, const void *d_src, int64_t bytes);

/* ---- timing helper: HIP-event elapsed ms around work already enqueued on `stream` ---------- */
int hite_event_create(void **ev);
int hite_event_record(void *ev, void *stream);
int hite_event_elapsed_ms(void *ev_start, void *ev_stop, float *ms);
int hite_event_destroy(void *ev);

/* ---- merge of the boundary calls between the GPUs of a node -----------------------------------------------
 * The reference has no counterpart (it collects results from a fork()ed process pool, Util.py:8141-8194).  One ncclAllGather
 * (RCCL over xGMI) of `bytes_per_rank` bytes per rank -- the fixed 32-byte hite_call records of a rank's share, padded to the
 * largest share -- on the communicator the CALLER owns (`nccl_comm` = its ncclComm_t), asynchronous on `stream`; d_recv holds
 * world x bytes_per_rank bytes in rank order.  The library neither links nor loads RCCL: the symbol is taken from the RCCL already in
 * the process (torch's, or the application's -- the communicator came out of it); HITE_ENODEV when there is none.  The Python driver's
 * merge (hite_amd/dist.py) goes through torch.distributed, which keeps its communicator to itself; this entry is for a C / C++ host. */
int hite_allgather_records(hite_ctx *ctx, void *nccl_comm, const void *d_send, void *d_recv, int64_t bytes_per_rank, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HITE_GPU_H */
