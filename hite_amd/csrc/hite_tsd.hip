// hite_tsd.hip -- k-mer TSD seed matching: search_confident_tir_v4
// (/root/reference/module/Util.py:7734-7845; batch driver search_confident_tir_batch_v1 :6533).
//
// One wavefront per flanked candidate.  For each k in {2,3,4,5,6,8,9,10,11} the k-mers of the two
// +-dist windows around the raw boundaries become 3-bit-per-base codes in LDS (A C G T N/other),
// lanes own right-window positions and scan the left codes (the reference's dict keeps, per k-mer,
// the left occurrence closest to the raw start and the running closest right occurrence), records
// are de-duplicated with the set semantics of TSD_set, filtered (NN, length < 100, TG..CA,
// TATATATA / ATATATAT) and ranked by (distance, tir_start, tir_end, k) -- the canonical order that
// replaces the reference's PYTHONHASHSEED-dependent tie order -- and the top 100 are kept.
// Integer / byte work only; per candidate ~(len+100) bytes in, <= 1.6 KB out: latency bound.
#include "hite_common.h"

#define TW 128        // max window length (2*dist+1 <= 101 in every reference call)
#define MAXREC 1024   // 9 k values x <= 101 right positions

struct TsdShared {
    unsigned long long lcode[TW], rcode[TW];
    int lpos_best[TW];   // per right position: tir_start or -1
    int rpos_best[TW];
    int rec_k[MAXREC], rec_ts[MAXREC], rec_te[MAXREC], rec_d[MAXREC];
    int nrec;
};

__device__ __forceinline__ unsigned code3(uint8_t c) {
    switch (c) { case 'A': return 1; case 'C': return 2; case 'G': return 3; case 'T': return 4; default: return 5; }  /* N and every folded byte */
}

__global__ void __launch_bounds__(64) tsd_kmer_kernel(int n, const uint8_t *__restrict__ seqs, const int64_t *__restrict__ seq_off,
                                                      int flank, int plant, int32_t *__restrict__ rec_out,
                                                      int32_t *__restrict__ cnt_out) {
    __shared__ TsdShared S;
    const int c = blockIdx.x;
    if (c >= n) return;
    const int lane = threadIdx.x;
    const uint8_t *seq = seqs + seq_off[c];
    const int len = (int)(seq_off[c + 1] - seq_off[c]);
    // caller convention (Util.py:6550): raw_tir_start = flank + 1, raw_tir_end = len - flank (1-based) -> 0-based below
    const int raw_start = flank, raw_end = len - flank - 1, dist = flank;
    if (lane == 0) S.nrec = 0;
    __syncthreads();
    if (len <= 0 || raw_end < 0) { if (lane == 0) cnt_out[c] = 0; return; }
    int ls = raw_start - dist; if (ls < 0) ls = 0;
    int le = raw_start + dist + 1; if (le > len) le = len; if (le < ls) le = ls;
    int rs = raw_end - dist; if (rs < 0) rs = 0;
    int re = raw_end + dist + 1; if (re > len) re = len; if (re < rs) re = rs;
    const int llen = le - ls, rlen = re - rs;
    if (llen > TW || rlen > TW) { if (lane == 0) cnt_out[c] = -1; return; }
    const int KS[9] = {2, 3, 4, 5, 6, 8, 9, 10, 11};
    for (int ki = 0; ki < 9; ki++) {
        const int k = KS[ki];
        // codes of every k-mer of both windows (0 = no k-mer at that offset)
        for (int i = lane; i < TW; i += 64) {
            unsigned long long a = 0, b = 0;
            if (i + k <= llen) for (int q = 0; q < k; q++) a = (a << 3) | code3(seq[ls + i + q]);
            if (i + k <= rlen) for (int q = 0; q < k; q++) b = (b << 3) | code3(seq[rs + i + q]);
            S.lcode[i] = a; S.rcode[i] = b;
        }
        __syncthreads();
        for (int i = lane; i < TW; i += 64) {
            int ts = -1, te = -1;
            const unsigned long long rc = S.rcode[i];
            const int cur_pos = rs + i - 1;
            if (rc != 0 && cur_pos >= 0 && cur_pos <= len - 1) {
                // left_pos: occurrence closest to raw_start, first wins ties (Util.py:7769-7777)
                int best = -1, bd = 0;
                for (int j = 0; j + k <= llen; j++) {
                    if (S.lcode[j] != rc) continue;
                    int lp = ls + j + k;
                    if (lp < 0 || lp > len - 1) continue;
                    int d = lp - raw_start; if (d < 0) d = -d;
                    if (best < 0 || d < bd) { best = lp; bd = d; }
                }
                if (best >= 0) {
                    // right_pos at the time this occurrence is visited: best among right occurrences 0..i (:7788-7794)
                    int rb = -1, rd = 0;
                    for (int j = 0; j <= i; j++) {
                        if (S.rcode[j] != rc) continue;
                        int rp = rs + j - 1;
                        if (rp < 0 || rp > len - 1) continue;
                        int d = rp - raw_end; if (d < 0) d = -d;
                        if (rb < 0 || d < rd) { rb = rp; rd = d; }
                    }
                    bool ok;
                    if (k != 2 && k != 4) ok = true;
                    else if (k == 4) ok = seq[rs + i] == 'T' && seq[rs + i + 1] == 'T' && seq[rs + i + 2] == 'A' && seq[rs + i + 3] == 'A';
                    else {
                        ok = seq[rs + i] == 'T' && seq[rs + i + 1] == 'A';
                        if (!ok && plant == 0) {
                            bool f = best + 3 <= len && seq[best] == 'C' && seq[best + 1] == 'C' && seq[best + 2] == 'C';
                            bool l = rb - 2 >= 0 && rb + 1 <= len && seq[rb - 2] == 'G' && seq[rb - 1] == 'G' && seq[rb] == 'G';
                            ok = f && l;
                        }
                    }
                    if (ok) { ts = best; te = rb; }
                }
            }
            S.lpos_best[i] = ts; S.rpos_best[i] = te;
        }
        __syncthreads();
        // set semantics + filters (:7810-7831)
        for (int i = lane; i < TW; i += 64) {
            const int ts = S.lpos_best[i], te = S.rpos_best[i];
            if (ts < 0) continue;
            const unsigned long long rc = S.rcode[i];
            bool dup = false;
            for (int j = 0; j < i; j++) if (S.rcode[j] == rc && S.lpos_best[j] == ts && S.rpos_best[j] == te) { dup = true; break; }
            if (dup) continue;
            bool nn = false;
            for (int q = 0; q + 1 < k; q++) if (seq[rs + i + q] == 'N' && seq[rs + i + q + 1] == 'N') nn = true;
            if (nn) continue;
            int b = te + 1; if (b > len) b = len;
            int L = b - ts; if (L < 0) L = 0;
            if (L < 100) continue;
            const uint8_t *t = seq + ts;
            if (t[0] == 'T' && t[1] == 'G' && t[L - 2] == 'C' && t[L - 1] == 'A') continue;
            bool ta = true, at = true;
            const char *TA = "TATATATA", *AT = "ATATATAT";
            for (int q = 0; q < 8; q++) { ta = ta && t[q] == (uint8_t)TA[q]; at = at && t[q] == (uint8_t)AT[q]; }
            if (ta || at) continue;
            int d1 = ts - raw_start, d2 = te - raw_end;
            if (d1 < 0) d1 = -d1;
            if (d2 < 0) d2 = -d2;
            int slot = atomicAdd(&S.nrec, 1);
            if (slot < MAXREC) { S.rec_k[slot] = k; S.rec_ts[slot] = ts; S.rec_te[slot] = te; S.rec_d[slot] = d1 + d2; }
        }
        __syncthreads();
    }
    // rank by (distance, tir_start, tir_end, k); keys are unique (set semantics) -> rank = number of smaller keys
    const int nr = S.nrec < MAXREC ? S.nrec : MAXREC;
    for (int i = lane; i < nr; i += 64) {
        int rank = 0;
        const int d = S.rec_d[i], ts = S.rec_ts[i], te = S.rec_te[i], k = S.rec_k[i];
        for (int j = 0; j < nr; j++) {
            const int dj = S.rec_d[j], tsj = S.rec_ts[j], tej = S.rec_te[j], kj = S.rec_k[j];
            bool less = dj < d || (dj == d && (tsj < ts || (tsj == ts && (tej < te || (tej == te && kj < k)))));
            rank += less;
        }
        if (rank < 100) {
            int32_t *o = rec_out + ((int64_t)c * 100 + rank) * 4;
            o[0] = k; o[1] = ts; o[2] = te; o[3] = d;
        }
    }
    if (lane == 0) cnt_out[c] = nr < 100 ? nr : 100;
}

struct TBuf {
    void *p = nullptr;
    ~TBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
    hipError_t up(const void *h, size_t n) {
        hipError_t e = alloc(n + 16);
        if (e != hipSuccess) return e;
        return n ? hipMemcpy(p, h, n, hipMemcpyHostToDevice) : hipSuccess;
    }
};

extern "C" int hite_tsd_kmer_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_seqs, const int64_t *d_seq_off, int32_t flank,
                                 int32_t plant, int32_t *d_rec_out, int32_t *d_cnt_out, void *stream) {
    if (!ctx || n < 0 || flank < 0 || 2 * flank + 1 > TW) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    hipLaunchKernelGGL(tsd_kmer_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, n, d_seqs, d_seq_off, flank, plant, d_rec_out,
                       d_cnt_out);
    HITE_CHECK(ctx, hipGetLastError());
    return HITE_OK;
}

extern "C" int hite_tsd_kmer(hite_ctx *ctx, int32_t n, const uint8_t *seqs, const int64_t *seq_off, int32_t flank, int32_t plant,
                             int32_t *rec_out, int32_t *cnt_out) {
    if (!ctx || n < 0 || !seqs || !seq_off || !rec_out || !cnt_out) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    TBuf ds, dof, dr, dc;
    hipError_t e = ds.up(seqs, seq_off[n]);
    if (e == hipSuccess) e = dof.up(seq_off, (n + 1) * 8);
    if (e == hipSuccess) e = dr.alloc((size_t)n * 100 * 16);
    if (e == hipSuccess) e = dc.alloc((size_t)n * 4);
    HITE_CHECK(ctx, e);
    int rc = hite_tsd_kmer_dev(ctx, n, (uint8_t *)ds.p, (int64_t *)dof.p, flank, plant, (int32_t *)dr.p, (int32_t *)dc.p, nullptr);
    if (rc) return rc;
    HITE_CHECK(ctx, hipDeviceSynchronize());
    HITE_CHECK(ctx, hipMemcpy(rec_out, dr.p, (size_t)n * 100 * 16, hipMemcpyDeviceToHost));
    HITE_CHECK(ctx, hipMemcpy(cnt_out, dc.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return HITE_OK;
}

// ---------------------------------------------------------------------------------------------
// non-LTR candidate preparation: search_polyA_TSD  Util.py:10915-11007
// (find_nearest_polyA / polyT :10865 / :10903, find_nearest_tandem :9772).  One wavefront per flanked repeat:
// lanes own window positions for the poly-A / poly-T runs, the tandem units and the TSD k-mers.
// find_near_matches(TSD, kmer, max_l_dist = 1) on two strings of the SAME length k has a closed form: a substring of
// kmer within one edit of TSD is either kmer itself (<= 1 substitution) or kmer without its last / first character
// (TSD with one character deleted); all three reduce to "common prefix + common suffix >= k - 1".
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void np_slice(long long a, long long b, long long n, long long *lo, long long *hi) {
    if (a < 0) { a += n; if (a < 0) a = 0; }
    if (b < 0) { b += n; if (b < 0) b = 0; }
    if (a > n) a = n;
    if (b > n) b = n;
    if (b < a) b = a;
    *lo = a; *hi = b;
}
// wave-wide arg-max of key (0 = nothing); returns the winning lane or -1
__device__ __forceinline__ int wave_argmax(unsigned long long key) {
    unsigned long long best = key;
    for (int d = 32; d >= 1; d >>= 1) { unsigned long long o = __shfl_xor(best, d); best = o > best ? o : best; }
    if (best == 0) return -1;
    return __ffsll((long long)__ballot(key == best)) - 1;
}
// longest run of `base` (>= 6, first among equals) in seq[lo, hi), hi - lo <= 50: window coordinates [*s, *e)
__device__ bool np_longest_run(const uint8_t *seq, long long lo, long long hi, uint8_t base, long long *s, long long *e) {
    const int lane = threadIdx.x & 63;
    const long long i = lo + lane;
    int len = 0;
    if (i < hi && seq[i] == base && (i == lo || seq[i - 1] != base))
        for (long long q = i; q < hi && seq[q] == base; q++) len++;
    const unsigned long long key = len >= 6 ? ((unsigned long long)len << 8) | (unsigned long long)(63 - lane) : 0ull;
    const int w = wave_argmax(key);
    if (w < 0) return false;
    *s = w; *e = w + __shfl(len, w);
    return true;
}
__device__ bool np_nearest_poly(const uint8_t *seq, long long n, long long pos, uint8_t base, long long *s, long long *e) {
    long long lo, hi, ws, we;
    np_slice(pos - 25 > 0 ? pos - 25 : 0, pos + 25 < n ? pos + 25 : n, n, &lo, &hi);
    if (!np_longest_run(seq, lo, hi, base, &ws, &we)) return false;
    const long long a = pos - 25 + ws, b = pos - 25 + we;
    *s = a > 0 ? a : 0; *e = b > 0 ? b : 0;
    return true;
}
__device__ bool np_nearest_tandem(const uint8_t *seq, long long n, long long pos, long long *s, long long *e) {
    const int lane = threadIdx.x & 63;
    const long long start = pos - 25 > 0 ? pos - 25 : 0, end = pos + 25 < n ? pos + 25 : n;
    unsigned long long key = 0;
    for (int m = 2; m <= 6; m++) {
        const long long i = start + lane;
        if (i < end - (long long)m * 4 + 1) {
            long long lo, hi;
            np_slice(i, i + (long long)m * 4, n, &lo, &hi);
            const long long L = hi - lo;
            bool ok = L > 0 && L % m == 0;
            for (long long q = 0; ok && q < L; q++) ok = seq[lo + q] == seq[lo + q % m];
            // first in (m ascending, i ascending) order among the longest
            const unsigned long long k2 = ok ? ((unsigned long long)L << 16) | (unsigned long long)((7 - m) << 8) | (unsigned long long)(63 - lane) : 0ull;
            key = k2 > key ? k2 : key;
        }
    }
    const int w = wave_argmax(key);
    if (w < 0) return false;
    const unsigned long long kw = ((unsigned long long)(unsigned)__shfl((int)(key >> 32), w) << 32) | (unsigned)__shfl((int)key, w);
    const long long L = (long long)(kw >> 16);
    *s = start + w; *e = start + w + L;
    return true;
}
// two strings of length k: is some substring of t within one edit of p?
__device__ __forceinline__ bool np_near1(const uint8_t *p, const uint8_t *t, int k) {
    int lp = 0, ls = 0, lp1 = 0, ls1 = 0;
    while (lp < k && p[lp] == t[lp]) lp++;
    if (lp == k) return true;
    while (ls < k && p[k - 1 - ls] == t[k - 1 - ls]) ls++;
    if (lp + ls >= k - 1) return true;                                  // <= 1 substitution
    while (ls1 < k - 1 && p[k - 1 - ls1] == t[k - 2 - ls1]) ls1++;      // p without one character == t[0 : k-1]
    if (lp + ls1 >= k - 1) return true;
    while (lp1 < k - 1 && p[lp1] == t[1 + lp1]) lp1++;                  // p without one character == t[1 : k]
    return lp1 + ls >= k - 1;
}

__global__ void __launch_bounds__(256) nonltr_prep_kernel(int n, const uint8_t *__restrict__ seqs, const int64_t *__restrict__ seq_off,
                                                          int flank, int win5, int64_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n) return;
    const uint8_t *seq = seqs + seq_off[c];
    const long long L = seq_off[c + 1] - seq_off[c];
    const long long raw_start = flank + 1, raw_end = L - flank;
    long long end_3 = -1, end_5 = -1;
    int direct = 0;
    long long ps = 0, pe = 0, ts = 0, te = 0;
    for (int side = 0; side < 2; side++) {
        const long long pos = side == 0 ? raw_end : raw_start;
        const bool hp = np_nearest_poly(seq, L, pos, side == 0 ? 'A' : 'T', &ps, &pe);
        const bool ht = np_nearest_tandem(seq, L, pos, &ts, &te);
        long long plen = 0, tlen = 0, lo, hi;
        if (hp) { np_slice(ps, pe, L, &lo, &hi); plen = hi - lo; }
        if (ht) tlen = te - ts;
        if ((plen < tlen ? plen : tlen) > 0) {
            if (side == 0) { end_3 = plen > tlen ? pe : te; end_5 = raw_start; direct = 1; }
            else { end_3 = plen > tlen ? ps : ts; end_5 = raw_end; direct = 2; }
        }
    }
    int found = 0;
    long long tsd_s = 0, tsd_n = 0;
    if (end_3 != -1 && end_5 != -1 && direct) {
        long long wlo, whi;
        np_slice(end_5 - win5 > 0 ? end_5 - win5 : 0, end_5 + win5, L, &wlo, &whi);
        for (int k = 20; k >= 8 && !found; k--) {
            long long tlo, thi;
            if (direct == 2) np_slice(end_3 - k, end_3, L, &tlo, &thi); else np_slice(end_3, end_3 + k, L, &tlo, &thi);
            if (thi - tlo != k) continue;
            bool hasN = false;
            for (int q = 0; q < k; q++) hasN = hasN || seq[tlo + q] == 'N';
            if (hasN) continue;                      // a TSD with N never counts (:10971), whatever matches
            unsigned long long hit = 0;
            int first = -1;
            for (long long i0 = 0; i0 + k <= whi - wlo && first < 0; i0 += 64) {
                const long long i = i0 + lane;
                const bool ok = i + k <= whi - wlo && np_near1(seq + tlo, seq + wlo + i, k);
                hit = __ballot(ok);
                if (hit) first = (int)i0 + __ffsll((long long)hit) - 1;
            }
            if (first >= 0) {
                end_5 = (end_5 - win5 > 0 ? end_5 - win5 : 0) + first + (direct == 1 ? k : 0);
                found = 1; tsd_s = tlo; tsd_n = k;
            }
        }
    }
    if (lane == 0) {
        int64_t *o = out + (int64_t)c * 6;
        o[0] = found; o[1] = direct; o[2] = tsd_s; o[3] = tsd_n;
        if (!direct) { o[4] = 0; o[5] = 0; }
        else {
            long long lo, hi;
            np_slice(end_5 < end_3 ? end_5 : end_3, end_5 > end_3 ? end_5 : end_3, L, &lo, &hi);
            o[4] = lo; o[5] = hi;
        }
    }
}

// out: 6 x int64 per sequence = {found_TSD, direct (0 none, 1 '+', 2 '-'), TSD start, TSD length, lo, hi of non_ltr_seq}
extern "C" int hite_nonltr_prep(hite_ctx *ctx, int32_t n, const uint8_t *seqs, const int64_t *seq_off, int32_t flank, int32_t win5,
                                int64_t *out) {
    if (!ctx || n < 0 || (n > 0 && (!seqs || !seq_off || !out)) || flank < 0 || win5 < 0 || win5 > 25) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    TBuf ds, dof, dout;
    hipError_t e = ds.up(seqs, seq_off[n]);
    if (e == hipSuccess) e = dof.up(seq_off, (size_t)(n + 1) * 8);
    if (e == hipSuccess) e = dout.alloc((size_t)n * 48);
    HITE_CHECK(ctx, e);
    hipLaunchKernelGGL(nonltr_prep_kernel, dim3((n + 3) / 4), dim3(256), 0, nullptr, n, (const uint8_t *)ds.p, (const int64_t *)dof.p, flank,
                       win5, (int64_t *)dout.p);
    HITE_CHECK(ctx, hipGetLastError());
    HITE_CHECK(ctx, hipDeviceSynchronize());
    HITE_CHECK(ctx, hipMemcpy(out, dout.p, (size_t)n * 48, hipMemcpyDeviceToHost));
    return HITE_OK;
}
