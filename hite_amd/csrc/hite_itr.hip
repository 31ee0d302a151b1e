// hite_itr.hip -- terminal-inverted-repeat search: the in-tree stage where the reference runs the
// third-party ELF `tools/itrsearch -i 0.7 -l 7` (run_itrsearch Util.py:216-224; call sites search_confident_tir_batch_v1
// Util.py:6556-6587 on first 40 + last 40 bases of every k-mer TSD variant, remove_no_tirs Util.py:13897-13920 on whole low-copy
// sequences).  Definition: oracle/hite_oracle_itr.c (read from the tool's disassembly, pinned to the tool's own output by
// tests/golden/itr_search.json.gz); this file must equal it field for field.
//
// One wavefront per sequence.  seq1 = the first h bases, seq2 = the reverse complement of the last h (h = min(500, L/2), or the
// end length when the caller asks for first-e + last-e).  Gotoh extension alignment anchored at (0,0) with a free end, swept in
// strips of 64 columns: lane l owns column 64*s + l + 1, at step t it computes row t - l + 1 (the anti-diagonal skew), taking
// C(i,j-1), C(i-1,j-1) and E(i,j-1) from the lane below it; lane 0 takes them from the matrix edge (strip 0, including the tool's
// quirk: the diagonal predecessor of (i,1) is C(i,0)) or from the column lane 63 of the strip before left behind.  Per cell one
// byte goes to the traceback store (move | E-gap opened here << 2 | F-gap opened here << 3), skewed [strip][step][lane] so that
// a step writes 64 consecutive bytes; for h <= 64 the store and both ends of the sequence live in LDS (9 KB per wavefront), longer
// sequences (the low-copy rescue, a few hundred per stage) keep the store in a scratch slot in HBM.  The end cell is the first
// maximum in row-major order (per-lane best, then a wave reduction on (score, -i, -j)); the traceback and the identity count are
// a serial walk every lane executes alike.
#include "hite_common.h"

#define ITR_MAX_H 500   // ItrAlign's own cap on either end (main() constructs it with 500)
#define ITR_LDS_H 64    // up to here the traceback store fits the wavefront's LDS share
#define ITR_WAVES 4     // wavefronts (sequences in flight) per workgroup
#define ITR_LDS_TB ((ITR_LDS_H + 63) * 64)

struct ItrArgs {
    int64_t n;
    const uint8_t *seqs;
    const int64_t *off;
    int32_t end_len, min_len, match, mismatch, go, ge, max_h;
    double min_id;
    uint8_t *tb_glob;        // per wavefront slot: traceback store + the strip boundary columns (NULL: every h <= ITR_LDS_H)
    size_t slot_bytes;
    int32_t *out;
};

__device__ __forceinline__ uint8_t itr_fold(uint8_t c) { return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : (uint8_t)'N'; }

__global__ __launch_bounds__(ITR_WAVES * 64) void itr_search_kernel(ItrArgs a) {
    extern __shared__ uint8_t itr_lds[];
    const int lane = lane_id(), w = wave_id();
    const bool tb_in_lds = a.tb_glob == nullptr;
    const int seq_cap = tb_in_lds ? ITR_LDS_H : ITR_MAX_H;     // bytes of LDS per sequence end (rounded up below)
    const int seq_lds = (seq_cap + 15) & ~15;
    const int wave_lds = 2 * seq_lds + (tb_in_lds ? ITR_LDS_TB : 0);
    uint8_t *s1 = itr_lds + (size_t)w * wave_lds;       // s1[i - 1] = seq1 base i
    uint8_t *s2 = s1 + seq_lds;                          // s2[j - 1] = seq2 base j
    uint8_t *tb;
    int32_t *cb = nullptr;                               // boundary columns: C and E of column 64*s, two ping-pong pairs
    if (tb_in_lds) tb = s2 + seq_lds;
    else {
        uint8_t *slot = a.tb_glob + (size_t)(blockIdx.x * ITR_WAVES + w) * a.slot_bytes;
        cb = (int32_t *)slot;
        tb = slot + 4 * (size_t)(ITR_MAX_H + 1) * 4;
    }
    const int go = a.go, ge = a.ge;
    for (int64_t k = (int64_t)blockIdx.x * ITR_WAVES + w; k < a.n; k += (int64_t)gridDim.x * ITR_WAVES) {
        const uint8_t *s = a.seqs + a.off[k];
        const int64_t L = a.off[k + 1] - a.off[k];
        int64_t h64 = a.end_len > 0 ? (L < a.end_len ? L : a.end_len) : L / 2;
        if (h64 > ITR_MAX_H) h64 = ITR_MAX_H;
        const int h = (int)h64;
        int32_t *o = a.out + 8 * k;
        if (h <= 0 || h > a.max_h) {   // max_h: what the caller sized the LDS / scratch for
            if (lane < 8) o[lane] = lane == 6 ? -1 : (lane == 7 && h > a.max_h ? 8 : 0);
            continue;
        }
        for (int x = lane; x < h; x += 64) {
            s1[x] = itr_fold(s[x]);
            s2[x] = comp_sym(s[L - 1 - x]);
        }
        const int T = h + 63;    // steps per strip
        const int ns = (h + 63) >> 6;
        int best = 0, best_i = 0, best_j = 0;
        for (int st = 0; st < ns; st++) {
            const int j = st * 64 + lane + 1;
            const bool colv = j <= h;
            const int cj = colv ? s2[j - 1] : 0;
            int c1 = -(go + j * ge), c2 = 0, e1 = 0;       // C(i-1,j) (row 0 to begin with), C(i-2,j), E(i-1,j)
            int F = -(go + j * ge + go);
            const int32_t *cb_in = cb + ((st + 1) & 1) * 2 * (ITR_MAX_H + 1);   // written by the strip before
            int32_t *cb_out = cb + (st & 1) * 2 * (ITR_MAX_H + 1);
            uint8_t *tbs = tb + (size_t)st * T * 64;
            if (st > 0) __threadfence_block();
            for (int t = 0; t < T; t++) {
                const int i = t - lane + 1;
                int lc1 = __shfl_up(c1, 1, 64), lc2 = __shfl_up(c2, 1, 64), le1 = __shfl_up(e1, 1, 64);
                const bool act = colv && i >= 1 && i <= h;
                if (lane == 0 && act) {
                    if (st == 0) {
                        lc1 = -(go + i * ge);     // C(i,0)
                        lc2 = lc1;                // the tool's diagonal predecessor of (i,1): C(i,0), not C(i-1,0)
                        le1 = lc1 - go;           // the E it starts a row with
                    } else {
                        lc1 = cb_in[i];
                        lc2 = i > 1 ? cb_in[i - 1] : -(go + (j - 1) * ge);
                        le1 = cb_in[ITR_MAX_H + 1 + i];
                    }
                }
                if (act) {
                    const int ci = s1[i - 1];
                    const bool eo = lc1 - go > le1;
                    const int e = eo ? lc1 - go - ge : le1 - ge;
                    const bool fo = !(F >= c1 - go);
                    F = fo ? c1 - go - ge : F - ge;
                    const int d = lc2 + ((ci == cj || ci == 'N' || cj == 'N') ? a.match : -a.mismatch);
                    int c, dir;
                    if (d >= F && d >= e) { c = d; dir = 1; }
                    else if (F >= e) { c = F; dir = 2; }
                    else { c = e; dir = 3; }
                    tbs[t * 64 + lane] = (uint8_t)(dir | (eo ? 4 : 0) | (fo ? 8 : 0));
                    c2 = c1; c1 = c; e1 = e;
                    if (c > best || (c == best && i < best_i)) { best = c; best_i = i; best_j = j; }
                    if (lane == 63 && st + 1 < ns) { cb_out[i] = c; cb_out[ITR_MAX_H + 1 + i] = e; }
                }
            }
        }
        // first maximum in row-major order: largest score, then smallest i, then smallest j
        unsigned long long key = ((unsigned long long)(unsigned)best << 32) | ((unsigned)(0xFFFF - best_i) << 16) | (unsigned)(0xFFFF - best_j);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            unsigned long long other = __shfl_xor(key, d, 64);
            key = other > key ? other : key;
        }
        const int mx = (int)(key >> 32), ei = 0xFFFF - (int)((key >> 16) & 0xFFFF), ej = 0xFFFF - (int)(key & 0xFFFF);
        int matches = 0, aligned = 0, flags = 0;
        if (mx > 0) {
            if (!tb_in_lds) __threadfence_block();
#define ITR_TB(i_, j_) tb[((size_t)(((j_) - 1) >> 6) * T + (i_) + (((j_) - 1) & 63) - 1) * 64 + (((j_) - 1) & 63)]
            int bi = ei, bj = ej;
            while (bi > 0 && bj > 0) {
                const int b = ITR_TB(bi, bj), dir = b & 3;
                if (dir == 1) {
                    aligned++;
                    matches += s1[bi - 1] == s2[bj - 1];
                    bi--; bj--;
                } else if (dir == 3) {      // gap in seq1: back to the column where this E-gap opened
                    int kk = bj;
                    while (kk >= 1 && !(ITR_TB(bi, kk) & 4)) kk--;
                    if (kk < 1) { flags |= 1; kk = 1; }   // (a gap run into the matrix edge: see the definition, never seen)
                    bj = kk - 1;
                } else {                    // gap in seq2
                    int kk = bi;
                    while (kk >= 1 && !(ITR_TB(kk, bj) & 8)) kk--;
                    if (kk < 1) { flags |= 1; kk = 1; }
                    bi = kk - 1;
                }
            }
#undef ITR_TB
        }
        if (lane == 0) {
            int found = 0;
            if (mx > 0) {
                const double ident = (double)(matches & 0xFFFF) / (double)(aligned & 0xFFFF);
                found = ((unsigned)a.min_len <= (unsigned)ei && ident >= a.min_id) ? 1 : 0;
            }
            o[0] = mx; o[1] = mx > 0 ? ei : 0; o[2] = mx > 0 ? ej : 0; o[3] = matches; o[4] = aligned; o[5] = found;
            o[6] = mx > 0 ? ei - 1 : -1; o[7] = flags;
        }
    }
}

extern "C" int hite_itr_search_dev(hite_ctx *ctx, int64_t n, const uint8_t *d_seqs, const int64_t *d_seq_off, int32_t end_len,
                                   int32_t max_h, double min_identity, int32_t min_len, int32_t match, int32_t mismatch,
                                   int32_t gap_open, int32_t gap_extend, int32_t *d_out, void *stream) {
    if (!ctx || n < 0 || !d_seqs || !d_seq_off || !d_out || match <= 0 || mismatch <= 0 || gap_open <= 0 || gap_extend <= 0 || max_h < 0)
        return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    if (max_h > ITR_MAX_H) max_h = ITR_MAX_H;
    if (end_len > 0 && max_h > end_len) max_h = end_len;
    ItrArgs a;
    a.n = n; a.seqs = d_seqs; a.off = d_seq_off; a.end_len = end_len; a.min_len = min_len; a.match = match; a.mismatch = mismatch;
    a.go = gap_open; a.ge = gap_extend; a.max_h = max_h; a.min_id = min_identity; a.out = d_out;
    a.tb_glob = nullptr; a.slot_bytes = 0;
    int64_t blocks = (n + ITR_WAVES - 1) / ITR_WAVES;
    size_t lds;
    if (max_h <= ITR_LDS_H) {
        lds = (size_t)ITR_WAVES * (2 * ((ITR_LDS_H + 15) & ~15) + ITR_LDS_TB);
        if (blocks > 256 * 8) blocks = 256 * 8;
    } else {
        lds = (size_t)ITR_WAVES * 2 * ((ITR_MAX_H + 15) & ~15);
        const size_t ns = (size_t)(max_h + 63) >> 6;
        a.slot_bytes = (4 * (size_t)(ITR_MAX_H + 1) * 4 + ns * (size_t)(max_h + 63) * 64 + 255) & ~(size_t)255;
        if (blocks > 128) blocks = 128;     // 512 slots of <= 0.3 MB
        void *p = nullptr;
        int rc = hite_scratch_reserve(ctx, (size_t)blocks * ITR_WAVES * a.slot_bytes, &p);
        if (rc) return rc;
        a.tb_glob = (uint8_t *)p;
    }
    hipLaunchKernelGGL(itr_search_kernel, dim3((unsigned)blocks), dim3(ITR_WAVES * 64), lds, (hipStream_t)stream, a);
    HITE_CHECK(ctx, hipGetLastError());
    return HITE_OK;
}

extern "C" int hite_itr_search(hite_ctx *ctx, int64_t n, const uint8_t *seqs, const int64_t *seq_off, int32_t end_len,
                               double min_identity, int32_t min_len, int32_t match, int32_t mismatch, int32_t gap_open,
                               int32_t gap_extend, int32_t *out) {
    if (!ctx || n < 0 || (n > 0 && (!seqs || !seq_off || !out))) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    int64_t max_l = 0;
    for (int64_t k = 0; k < n; k++) {
        int64_t l = seq_off[k + 1] - seq_off[k];
        if (l < 0) return HITE_EINVAL;
        if (l > max_l) max_l = l;
    }
    int64_t mh = end_len > 0 ? (max_l < end_len ? max_l : end_len) : max_l / 2;
    if (mh > ITR_MAX_H) mh = ITR_MAX_H;
    const size_t nb = (size_t)seq_off[n];
    uint8_t *d_seq = nullptr;
    int64_t *d_off = nullptr;
    int32_t *d_out = nullptr;
    hipError_t e = hipMalloc(&d_seq, nb + 16);
    if (e == hipSuccess) e = hipMalloc(&d_off, (size_t)(n + 1) * 8);
    if (e == hipSuccess) e = hipMalloc(&d_out, (size_t)n * 32);
    if (e == hipSuccess && nb) e = hipMemcpy(d_seq, seqs, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_off, seq_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice);
    int rc = HITE_OK;
    if (e == hipSuccess) {
        rc = hite_itr_search_dev(ctx, n, d_seq, d_off, end_len, (int32_t)mh, min_identity, min_len, match, mismatch, gap_open,
                                 gap_extend, d_out, nullptr);
        if (rc == HITE_OK) e = hipDeviceSynchronize();
        if (rc == HITE_OK && e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * 32, hipMemcpyDeviceToHost);
    }
    if (d_seq) (void)hipFree(d_seq);
    if (d_off) (void)hipFree(d_off);
    if (d_out) (void)hipFree(d_out);
    if (rc) return rc;
    HITE_CHECK(ctx, e);
    return HITE_OK;
}
