/*
 * hite_oracle_itr.c -- TEST INFRASTRUCTURE ONLY (checker for tests/, smoke() and bench.py's cpu_baseline leg;
 * nothing under hite_amd/ may import, link or execute it).
 *
 * CPU restatement of `itrsearch`, the third-party terminal-inverted-repeat filter the reference calls on the TIR path
 * (SURVEY section 8, row a-8):
 *   run_itrsearch                     /root/reference/module/Util.py:216-224   (`tools/itrsearch -i 0.7 -l 7 <fasta>`)
 *   search_confident_tir_batch_v1     Util.py:6556-6587   (first 40 + last 40 bases of every k-mer TSD variant)
 *   remove_no_tirs (low-copy rescue)  Util.py:13897-13920 (whole low-copy sequences), called at Util.py:8196-8213
 *
 * The tool's source is NOT in /root/reference (only the x86-64 ELF `tools/itrsearch`, not stripped).  The algorithm below was
 * read from its disassembly (ItrAlign::alignItr / hasItr, ExtAlign::align_pass / traceback / view, ExtAlign::setMismatch,
 * main) and is PINNED to the tool itself: tests/golden/itr_search.json.gz holds the tool's own answers (found / "Length itr=")
 * on seeded inputs, produced by oracle/gen_golden.py running /root/reference/tools/itrsearch in the build container.
 *
 * What the tool does with one FASTA record `seq` of length L (all of it reproduced, quirks included):
 *   h = min(500, L / 2);  seq1 = seq[0:h];  seq2 = reverse_complement(seq)[0:h]          (ItrAlign::alignItr)
 *   extension alignment anchored at (0,0), affine gaps, free end: Gotoh's recurrences over C (best), E (gap in seq1: a step
 *   along seq2) and F (gap in seq2), scores w = +match if the two bases are equal OR EITHER IS 'N', else -mismatch; a gap of
 *   length g costs gap_open + g * gap_extend (defaults 10 / 16 / 32 / 32, the reference passes none of them).
 *   QUIRK 1 (ExtAlign::align_pass): the diagonal predecessor of cell (i, 1) is C(i, 0) = -(go + i*ge), not C(i-1, 0) -- even
 *   C(1,1) starts from -(go+ge) instead of 0, so every alignment's internal score is 64 below the textbook value and the
 *   alignment's END (the first cell, in row-major order, with the strictly largest score > 0) is chosen under that handicap.
 *   Ties between the three moves: diagonal >= F-move and >= E-move wins, then F-move (gap in seq2) >= E-move, then E-move.
 *   Traceback from the end cell to (0,0): a diagonal step is one aligned column; a gap jumps its whole length.
 *   identity = equal bases (plain byte equality: N == N counts, N against A does not) / aligned (diagonal) columns, both
 *   16-bit counters (ExtAlign::view, called by hasItr before getIdentity, which then returns the cached value);
 *   found = (min_len <= end_in_seq1) and (identity >= min_identity) in binary64                       (ItrAlign::hasItr)
 *   header of the .itr record: "... Length itr=<end_in_seq1 - 1>" (main: EndSeq1 - StartSeq1, StartSeq1 is always 1).
 *   QUIRK 2: two pieces of state are stale/uninitialised in the tool (the column where the current E-gap opened is not reset per
 *   row; the row where an F-gap opened is read from fresh heap memory for gaps that start in row 0).  Both only matter for a
 *   traceback that runs a gap into the matrix edge after a gap along the other edge -- never optimal with these penalties.
 *   This restatement carries the tool's state exactly (stale column kept, heap word = 0) and reports in `flags` bit 0 whether a
 *   traceback ever consumed such a cell: 0 on every golden and every generated case (tests assert it), which is what allows the
 *   HIP kernel to use the clean formulation.
 *
 * out[8 * k ..] = { score, end1, end2, matches, aligned, found, header_len (end1 - 1, -1 if no alignment), flags }
 * end_len > 0: the record is first end_len + last end_len bases of the sequence (Python `s[:e] + s[-e:]`, Util.py:6564, 6577),
 * composed here; end_len <= 0: the whole sequence (remove_no_tirs).  Bytes outside ACGT are N: the build folds IUPAC codes to N where the
 * genome is packed (deviation i), so no other letter reaches this stage.  The tool itself compares the IUPAC
 * letters K Y S B W R D M H V literally (complement table "TGCAKYSBWRDMHVNX" in its .rodata; only N is a wildcard, X never matches
 * X) and aborts on any other letter: a record with such codes, which the reference would hand over unfolded, is outside the pin.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint8_t itr_fold(uint8_t c) {
    switch (c) {
        case 'A': case 'C': case 'G': case 'T': return c;
        default: return 'N';
    }
}
static uint8_t itr_comp(uint8_t c) {
    switch (c) {
        case 'A': return 'T';
        case 'T': return 'A';
        case 'C': return 'G';
        case 'G': return 'C';
        default: return 'N';
    }
}

static int itr_one(const uint8_t *v, int64_t lv, double min_id, int32_t min_len, int32_t match, int32_t mismatch, int32_t go,
                   int32_t ge, int32_t max_len, int32_t *out) {
    int64_t h64 = lv / 2;
    if (h64 > max_len) h64 = max_len;
    int n = (int)h64;
    for (int k = 0; k < 8; k++) out[k] = 0;
    out[6] = -1;
    if (n <= 0) return 0;
    uint8_t *s1 = (uint8_t *)malloc((size_t)n + 2), *s2 = (uint8_t *)malloc((size_t)n + 2);
    int32_t *CC = (int32_t *)calloc((size_t)n + 1, 4), *DD = (int32_t *)calloc((size_t)n + 1, 4), *RR = (int32_t *)calloc((size_t)n + 1, 4);
    int32_t *M = (int32_t *)calloc((size_t)(n + 1) * (n + 1), 4);
    uint8_t *ST = (uint8_t *)calloc((size_t)(n + 1) * (n + 1), 1), *col_open = (uint8_t *)calloc((size_t)n + 1, 1);
    if (!s1 || !s2 || !CC || !DD || !RR || !M || !ST || !col_open) {
        free(s1); free(s2); free(CC); free(DD); free(RR); free(M); free(ST); free(col_open);
        return -1;
    }
    for (int i = 1; i <= n; i++) {
        s1[i] = itr_fold(v[i - 1]);
        s2[i] = itr_comp(v[lv - i]);
    }
    const int ld = n + 1; /* M[j * ld + i], as the tool indexes it */
    int32_t mx = 0, ei = 0, ej = 0, kE = 0, t = go;
    for (int j = 1; j <= n; j++) {
        t += ge;
        CC[j] = -t;
        DD[j] = -(t + go);
        M[j * ld] = j + 1;
    }
    t = go;
    for (int i = 1; i <= n; i++) {
        t += ge;
        int32_t s = -t, c = -t, e = -(t + go);
        int row_open = 0;
        M[i] = ~i;
        for (int j = 1; j <= n; j++) {
            if (c - go > e) { e = c - go - ge; kE = j - 1; row_open = 1; }
            else e -= ge;
            if (DD[j] >= CC[j] - go) DD[j] -= ge;
            else { DD[j] = CC[j] - go - ge; RR[j] = i - 1; col_open[j] = 1; }
            int32_t a = s + ((s1[i] == s2[j] || s1[i] == 'N' || s2[j] == 'N') ? match : -mismatch);
            if (a >= DD[j] && a >= e) { c = a; M[j * ld + i] = 1; }
            else if (DD[j] > a && DD[j] >= e) { c = DD[j]; M[j * ld + i] = ~(i - RR[j]); ST[j * ld + i] = !col_open[j]; }
            else { c = e; M[j * ld + i] = j - kE + 1; ST[j * ld + i] = !row_open; }
            s = CC[j];
            CC[j] = c;
            if (mx < c) { mx = c; ei = i; ej = j; }
        }
    }
    int flags = 0;
    if (mx != 0) {
        int bi = ei, bj = ej, guard = 0, matches = 0, aligned = 0, hung = 0;
        /* ExtAlign::traceback + the walk of ExtAlign::view fused: only diagonal columns enter the identity */
        while (bi >= 0 && bj >= 0) {
            int32_t m = M[bj * ld + bi];
            if (m == 0) break;
            if (++guard > 4 * n + 8) { hung = 1; break; }   /* m == -1 would spin in the tool */
            if (ST[bj * ld + bi]) flags |= 1;
            if (m == 1) {
                aligned++;
                matches += (s1[bi] == s2[bj]);
                bi--; bj--;
                continue;
            }
            if (m > 1) bj -= m - 1;
            if (m < -1) bi -= ~m;
        }
        if (hung) flags |= 4;
        double ident = (double)(uint16_t)matches / (double)(uint16_t)aligned;
        out[0] = mx; out[1] = ei; out[2] = ej; out[3] = matches; out[4] = aligned;
        out[5] = (!hung && (uint32_t)min_len <= (uint32_t)ei && ident >= min_id) ? 1 : 0;
        out[6] = ei - 1;
    }
    out[7] = flags;
    free(s1); free(s2); free(CC); free(DD); free(RR); free(M); free(ST); free(col_open);
    return 0;
}

int orc_itr_search(int32_t n, const uint8_t *seqs, const int64_t *off, int32_t end_len, double min_id, int32_t min_len,
                   int32_t match, int32_t mismatch, int32_t gap_open, int32_t gap_extend, int32_t max_len, int32_t *out) {
    for (int32_t k = 0; k < n; k++) {
        const uint8_t *s = seqs + off[k];
        int64_t L = off[k + 1] - off[k];
        int rc;
        if (end_len > 0) {
            int64_t e = L < end_len ? L : end_len;
            uint8_t *v = (uint8_t *)malloc((size_t)(2 * e) + 1);
            if (!v) return -1;
            memcpy(v, s, (size_t)e);
            memcpy(v + e, s + L - e, (size_t)e);
            rc = itr_one(v, 2 * e, min_id, min_len, match, mismatch, gap_open, gap_extend, max_len, out + 8 * (int64_t)k);
            free(v);
        } else {
            rc = itr_one(s, L, min_id, min_len, match, mismatch, gap_open, gap_extend, max_len, out + 8 * (int64_t)k);
        }
        if (rc) return rc;
    }
    return 0;
}
