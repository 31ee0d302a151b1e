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

#ifndef JB
#define JB 256          // threads per block (every kernel of this file; 128 measured: short alignments 20 % faster, long ones 23 % slower)
#endif
#define JW (JB / 64)    // wavefronts per block
#define MAXSEL 128      // capacity of the selected-row list (the reference keeps <= 101)
#define CS 12           // bytes of column statistics per column: cnt[6], first[6]
#define SCR_PER_COL 32  // scratch bytes per column per block slot
#define ANCHOR_LDS_COLS 5104   // ungapped row (<= this many bytes) + 2 bytes of match record per text start fit the 15 KB mask area
#define FNM_LIST 320    // 64 flagged ends x (2 k + 1 = 5) starts
#define TILE_COLS 160   // widest column span staged in LDS for the window scans (wider spans read the alignment directly)

struct JShared {
    int scan[8];
    int iv[16];            // misc broadcast slots
    unsigned int red[8];   // atomic reduction slots
    uint16_t sel[MAXSEL];  // selected rows
    int cols[100];         // valid columns of the current scan
    int res[4];            // per-wave window results
    uint8_t flag[JB];
    uint8_t pat[2][24];    // anchor patterns
    unsigned peq[8];       // Myers match masks per symbol class
    uint8_t cls[256];      // sym_class of every byte value: one LDS read (64 dwords, 64 banks: conflict-free) where the
                           // compare chain of sym_class costs ~8 instructions -- the kernel is bound by instruction issue
    int nflag;             // flagged match ends of the current anchor search
    int flagged[64];
    uint16_t mst[FNM_LIST]; // match list of the anchor search: start, longest match length, best distance << 5 | its length, group start
    uint8_t mml[FNM_LIST], mb[FNM_LIST], mgs[FNM_LIST];
    int tsd[25];
    int fo[5], eo[5];
    alignas(16) uint8_t tile[TILE_COLS * 6 * 4 * 4];   // row-set masks of the current scan: per column and symbol class, up to 4 words of 32 rows
#ifdef JUDGE_CLOCKS
    unsigned long long jt;
#endif
};

__device__ __forceinline__ void jshared_init(JShared &S) {
    for (int i = threadIdx.x; i < 256; i += JB) S.cls[i] = (uint8_t)sym_class((uint8_t)i);
    __syncthreads();
}

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


// ---------------------------------------------------------------------------------------------
// banded (k<=2) edit distance of pattern p[0..m) against prefixes of text t[0..w):
// out[b] = min(D[m][m+b-2], k+1) for b = 0..4
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void banded_dist(const uint8_t *p, int m, const uint8_t *t, int w, int k, int out[5]) {
    const int INF = k + 1;
    int prev[5], cur[5];
#pragma unroll
    for (int b = 0; b < 5; b++) {
        int j = b - 2;
        prev[b] = (j >= 0 && j <= w && j < INF) ? j : INF;
    }
    for (int i = 1; i <= m; i++) {
        uint8_t pc = p[i - 1];
#pragma unroll
        for (int b = 0; b < 5; b++) {
            int j = i + b - 2;
            int v = INF;
            if (j >= 0 && j <= w) {
                if (j >= 1) v = prev[b] + (pc != t[j - 1]);
                if (b + 1 <= 4) { int u = prev[b + 1] + 1; v = u < v ? u : v; }
                if (b >= 1) { int l = cur[b - 1] + 1; v = l < v ? l : v; }
                if (v > INF) v = INF;
            }
            cur[b] = v;
        }
#pragma unroll
        for (int b = 0; b < 5; b++) prev[b] = cur[b];
    }
#pragma unroll
    for (int b = 0; b < 5; b++) out[b] = prev[b];
}

// does the k<=2 neighbourhood of pattern p (len m) contain a substring of t[0..n)?  (serial, small n)
__device__ bool fnm_any_small(const uint8_t *p, int m, const uint8_t *t, int n, int k) {
    for (int s = 0; s < n; s++) {
        int w = (m + k) < (n - s) ? (m + k) : (n - s);
        if (w < m - k || w <= 0) continue;
        int d[5];
        banded_dist(p, m, t + s, w, k, d);
        int L0 = m - k > 1 ? m - k : 1;
        for (int L = L0; L <= w && L <= m + k; L++)
            if (d[L - (m - 2)] <= k) return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// block-level: ungapped copy of one row + position map
// ---------------------------------------------------------------------------------------------
__device__ int blk_ungap_row(const uint8_t *__restrict__ row, int C, uint8_t *__restrict__ ung,
                             int *__restrict__ reflex, JShared &S) {
    int running = 0;
    int lane = lane_id(), w = wave_id();
    for (int base = 0; base < C; base += JB) {
        int c = base + threadIdx.x;
        uint8_t ch = c < C ? row[c] : (uint8_t)'-';
        bool f = ch != '-';
        unsigned long long bal = __ballot(f);
        int pre = __popcll(bal & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) S.scan[w] = __popcll(bal);
        __syncthreads();
        int off = running;
        for (int i = 0; i < w; i++) off += S.scan[i];
        if (f) { ung[off + pre] = ch; reflex[off + pre] = c; }
        for (int i = 0; i < JW; i++) running += S.scan[i];
    }
    __syncthreads();
    return running;
}

// the same for the judge's anchor search: a thread owns a contiguous segment of the row (count, ONE block scan, write), and
// the position map is not stored at all -- the two columns the caller needs are looked up in the owning thread's segment
// (the round-by-round form above costs two barriers per 256 columns and a 4-byte store per base)
struct UngapSeg { int c0, L, off, cnt; };
__device__ int blk_ungap_row_seg(const uint8_t *__restrict__ row, int C, uint8_t *__restrict__ ung, UngapSeg &G, JShared &S) {
    const int L = (C + JB - 1) / JB, c0 = (int)threadIdx.x * L;
    int cnt = 0;
    for (int x = 0; x < L; x++) { const int c = c0 + x; if (c < C && row[c] != '-') cnt++; }
    int tot;
    const int off = block_excl_scan(cnt, S.scan, &tot);
    int o = off;
    for (int x = 0; x < L; x++) {
        const int c = c0 + x;
        if (c < C) { const uint8_t ch = row[c]; if (ch != '-') ung[o++] = ch; }
    }
    __syncthreads();
    G.c0 = c0; G.L = L; G.off = off; G.cnt = cnt;
    return tot;
}
// alignment column of base idx (0-based, < n) of the row last passed to blk_ungap_row_seg
__device__ int blk_col_of(const uint8_t *__restrict__ row, int C, const UngapSeg &G, int idx, JShared &S) {
    if (idx >= G.off && idx < G.off + G.cnt) {
        int k = idx - G.off;
        for (int x = 0; x < G.L; x++) {
            const int c = G.c0 + x;
            if (c < C && row[c] != '-') { if (k == 0) { S.iv[14] = c; break; } k--; }
        }
    }
    __syncthreads();
    const int r = S.iv[14];
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------
// find_near_matches restatement (oracle/stubs.py definition).  side 0: start of the best match
// of the FIRST overlap group; side 1: end (exclusive) of the best match of the LAST group.
// Returns -1 when there is no match.  minfo: 2 bytes per text start.
// ---------------------------------------------------------------------------------------------
// the form with a match record per text start (2 bytes, minfo): used when the filter flags more than 64 ends (repetitive text)
__device__ int blk_fnm_full(const uint8_t *pat, int m, const uint8_t *__restrict__ ung, int n, int k,
                            uint8_t *__restrict__ minfo, int side, JShared &S) {
    for (int i = threadIdx.x; i < 2 * n; i += JB) minfo[i] = 0;
    __syncthreads();
    {
        // Phase 2: exact banded DP for the starts e-(m+k) .. e-(m-k) of every flagged end, one (end, start)
        // pair per thread; if the list overflowed (repetitive text) fall back to every start.
        const int nf = S.nflag;
        const int span = 2 * k + 1;
        const int ntask = nf <= 64 ? nf * span : n;
        for (int tsk = threadIdx.x; tsk < ntask; tsk += JB) {
            int st;
            if (nf <= 64) { st = S.flagged[tsk / span] - (m + k) + (tsk % span); if (st < 0) continue; }
            else st = tsk;
            if (st >= n) continue;
            int w = (m + k) < (n - st) ? (m + k) : (n - st);
            if (w < m - k || w <= 0) continue;
            int d[5];
            banded_dist(pat, m, ung + st, w, k, d);
            int bd = 3, bL = 0, maxL = 0;
            int L0 = m - k > 1 ? m - k : 1;
            for (int LL = L0; LL <= w && LL <= m + k; LL++) {
                int dd = d[LL - (m - 2)];
                if (dd <= k) { maxL = LL; if (dd < bd || (dd == bd && LL > bL)) { bd = dd; bL = LL; } }
            }
            if (maxL) { minfo[2 * st] = (uint8_t)maxL; minfo[2 * st + 1] = (uint8_t)((bd << 5) | bL); }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { S.red[0] = 0xffffffffu; S.red[1] = 0u; S.red[2] = 0xffffffffu; S.red[3] = 0xffffffffu; }
    __syncthreads();
    // group starts: a start with a match that no earlier match overlaps
    const int look = m + k;
    for (int s = threadIdx.x; s < n; s += JB) {
        if (!minfo[2 * s]) continue;
        bool gs = true;
        int lo = s - look > 0 ? s - look : 0;
        for (int q = lo; q < s; q++) {
            int mL = minfo[2 * q];
            if (mL && q + mL > s) { gs = false; break; }
        }
        if (gs) { atomicMin(&S.red[0], (unsigned)s); atomicMax(&S.red[1], (unsigned)s + 1u); }
    }
    __syncthreads();
    unsigned first_gs = S.red[0];
    if (first_gs == 0xffffffffu) { __syncthreads(); return -1; }
    unsigned last_gs = S.red[1] - 1u;
    int result;
    if (side == 0) {
        // second group start
        for (int s = threadIdx.x; s < n; s += JB) {
            if ((unsigned)s <= first_gs || !minfo[2 * s]) continue;
            bool gs = true;
            int lo = s - look > 0 ? s - look : 0;
            for (int q = lo; q < s; q++) {
                int mL = minfo[2 * q];
                if (mL && q + mL > s) { gs = false; break; }
            }
            if (gs) atomicMin(&S.red[2], (unsigned)s);
        }
        __syncthreads();
        unsigned second = S.red[2] == 0xffffffffu ? (unsigned)n : S.red[2];
        for (unsigned s = first_gs + threadIdx.x; s < second; s += JB) {
            if (!minfo[2 * s]) continue;
            unsigned b = minfo[2 * s + 1];
            unsigned key = ((b >> 5) << 24) | ((63u - (b & 31u)) << 16) | s;
            atomicMin(&S.red[3], key);
        }
        __syncthreads();
        result = (int)(S.red[3] & 0xffffu);
    } else {
        for (unsigned s = last_gs + threadIdx.x; s < (unsigned)n; s += JB) {
            if (!minfo[2 * s]) continue;
            unsigned b = minfo[2 * s + 1];
            unsigned key = ((b >> 5) << 24) | ((63u - (b & 31u)) << 16) | s;
            atomicMin(&S.red[3], key);
        }
        __syncthreads();
        unsigned key = S.red[3];
        result = (int)(key & 0xffffu) + (int)(63u - ((key >> 16) & 0xffu));
    }
    __syncthreads();
    return result;
}

__device__ int blk_fnm(const uint8_t *pat, int m, const uint8_t *__restrict__ ung, int n, int k,
                       uint8_t *__restrict__ minfo, int side, JShared &S) {
    // Phase 1 (filter): Myers' bit-parallel approximate search.  Each thread scans a chunk of end positions
    // (plus m+k characters of warm-up, which makes every distance <= k exact) and flags the ends where
    // the pattern matches within k edits; 17 word operations per character instead of one banded DP per start.
    // Chunks are at least as long as the warm-up: the kernel is bound by instruction issue, and with 12-character chunks
    // (2 700 characters over 256 threads) two thirds of the filter's instructions were warm-up (chunks of twice the warm-up
    // measured better on long rows, worse on short ones: the chain of one thread gets too long).
    // Phase 2: the exact per-start banded DP runs only for the <= 2k+1 starts that can end at a flagged position; their
    // match records form a LIST (<= 64 ends x 5 starts) on which the overlap-group rule is evaluated pair by pair -- no
    // per-start array to clear and to scan three times.
    if (threadIdx.x < 8) {
        unsigned mk = 0;
        for (int i = 0; i < m; i++) if (sym_class(pat[i]) == (int)threadIdx.x) mk |= 1u << i;
        S.peq[threadIdx.x] = mk;
        if (threadIdx.x == 0) S.nflag = 0;
    }
    __syncthreads();
    {
        int L = (n + JB - 1) / JB;
        if (L < m + k) L = m + k;
        const int cs = threadIdx.x * L;            // ends [cs, cs + L) belong to this thread (end = index of last char)
        if (cs < n) {
            const int ce = cs + L < n ? cs + L : n;
            int j0 = cs - (m + k); if (j0 < 0) j0 = 0;
            unsigned Pv = 0xffffffffu, Mv = 0;
            int score = m;
            const unsigned top = 1u << (m - 1);
            for (int j = j0; j < ce; j++) {
                const unsigned Eq = S.peq[S.cls[ung[j]]];
                const unsigned Xv = Eq | Mv;
                const unsigned Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                unsigned Ph = Mv | ~(Xh | Pv);
                unsigned Mh = Pv & Xh;
                score += (Ph & top) ? 1 : 0;
                score -= (Mh & top) ? 1 : 0;
                Ph <<= 1; Mh <<= 1;
                Pv = Mh | ~(Xv | Ph);
                Mv = Ph & Xv;
                if (j >= cs && score <= k) {
                    // a match ends at character j: remember the exclusive end e = j + 1
                    const int q = atomicAdd(&S.nflag, 1);
                    if (q < 64) S.flagged[q] = j + 1;
                }
            }
        }
    }
    __syncthreads();
    const int nf = S.nflag;
    if (nf == 0) return -1;
    const int span = 2 * k + 1, N = nf * span;
    if (nf > 64 || N > FNM_LIST) return blk_fnm_full(pat, m, ung, n, k, minfo, side, S);
    for (int t = threadIdx.x; t < N; t += JB) {
        const int st = S.flagged[t / span] - (m + k) + (t % span);
        int maxL = 0, bd = 3, bL = 0;
        if (st >= 0 && st < n) {
            const int w = (m + k) < (n - st) ? (m + k) : (n - st);
            if (w >= m - k && w > 0) {
                int d[5];
                banded_dist(pat, m, ung + st, w, k, d);
                const int L0 = m - k > 1 ? m - k : 1;
                for (int LL = L0; LL <= w && LL <= m + k; LL++) {
                    const int dd = d[LL - (m - 2)];
                    if (dd <= k) { maxL = LL; if (dd < bd || (dd == bd && LL > bL)) { bd = dd; bL = LL; } }
                }
            }
        }
        S.mst[t] = (uint16_t)(maxL ? st : 0);
        S.mml[t] = (uint8_t)maxL;
        S.mb[t] = (uint8_t)((bd << 5) | bL);
    }
    if (threadIdx.x == 0) { S.red[0] = 0xffffffffu; S.red[1] = 0u; S.red[2] = 0xffffffffu; S.red[3] = 0xffffffffu; }
    __syncthreads();
    // group starts: a start with a match that no earlier match overlaps (the same start may sit in the list more than once,
    // with the same record: it does not overlap itself)
    for (int t = threadIdx.x; t < N; t += JB) {
        bool gs = S.mml[t] != 0;
        if (gs) {
            const int s0 = S.mst[t];
            for (int q = 0; q < N; q++) {
                const int mL = S.mml[q], sq = S.mst[q];
                if (mL && sq < s0 && sq + mL > s0) { gs = false; break; }
            }
            if (gs) { atomicMin(&S.red[0], (unsigned)s0); atomicMax(&S.red[1], (unsigned)s0 + 1u); }
        }
        S.mgs[t] = gs;
    }
    __syncthreads();
    const unsigned first_gs = S.red[0];
    if (first_gs == 0xffffffffu) { __syncthreads(); return -1; }
    const unsigned last_gs = S.red[1] - 1u;
    int result;
    if (side == 0) {
        for (int t = threadIdx.x; t < N; t += JB)
            if (S.mgs[t] && (unsigned)S.mst[t] > first_gs) atomicMin(&S.red[2], (unsigned)S.mst[t]);
        __syncthreads();
        const unsigned second = S.red[2] == 0xffffffffu ? (unsigned)n : S.red[2];
        for (int t = threadIdx.x; t < N; t += JB) {
            const unsigned s0 = S.mst[t];
            if (!S.mml[t] || s0 < first_gs || s0 >= second) continue;
            const unsigned b = S.mb[t];
            atomicMin(&S.red[3], ((b >> 5) << 24) | ((63u - (b & 31u)) << 16) | s0);
        }
        __syncthreads();
        result = (int)(S.red[3] & 0xffffu);
    } else {
        for (int t = threadIdx.x; t < N; t += JB) {
            const unsigned s0 = S.mst[t];
            if (!S.mml[t] || s0 < last_gs) continue;
            const unsigned b = S.mb[t];
            atomicMin(&S.red[3], ((b >> 5) << 24) | ((63u - (b & 31u)) << 16) | s0);
        }
        __syncthreads();
        const unsigned key = S.red[3];
        result = (int)(key & 0xffffu) + (int)(63u - ((key >> 16) & 0xffu));
    }
    __syncthreads();
    return result;
}

// ---------------------------------------------------------------------------------------------
// column statistics over the selected rows: cnt[6] (ACGTN-) and first-appearance row rank[6]
// ---------------------------------------------------------------------------------------------
__device__ void blk_colstats(const uint8_t *__restrict__ msa, int C, const uint16_t *sel, int rn,
                             uint8_t *__restrict__ cstat, const uint8_t *cls /* JShared::cls */) {
    // the six counters and the six first-appearance ranks live in two 64-bit registers (one byte per symbol class): a
    // dynamically indexed register array costs a compare-select chain per access.  rn <= MAXSEL (128) fits a byte.
    for (int c = threadIdx.x; c < C; c += JB) {
        unsigned long long cnt = 0, fst = 0;
        unsigned seen = 0;
        int r = 0;
        for (; r + 7 < rn; r += 8) {   // eight rows in flight (the loop is load-latency bound otherwise)
            uint8_t sy[8];
#pragma unroll
            for (int u = 0; u < 8; u++) sy[u] = msa[(size_t)sel[r + u] * C + c];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int k8 = cls[sy[u]] * 8;
                const unsigned fresh = ((seen >> k8 / 8) & 1u) ^ 1u;
                cnt += 1ull << k8;
                fst |= (unsigned long long)((unsigned)(r + u) & (0u - fresh)) << k8;
                seen |= 1u << (k8 / 8);
            }
        }
        for (; r < rn; r++) {
            const int k8 = cls[msa[(size_t)sel[r] * C + c]] * 8;
            const unsigned fresh = ((seen >> k8 / 8) & 1u) ^ 1u;
            cnt += 1ull << k8;
            fst |= (unsigned long long)((unsigned)r & (0u - fresh)) << k8;
            seen |= 1u << (k8 / 8);
        }
        uint8_t *o = cstat + (size_t)c * CS;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            o[k] = (uint8_t)(cnt >> (8 * k));
            o[6 + k] = ((seen >> k) & 1u) ? (uint8_t)(fst >> (8 * k)) : (uint8_t)255;
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// calculate_window_homology for one window, executed by one wavefront.  Rows are lanes
// (slot 0: sel[lane], slot 1: sel[lane+64]); window columns first, first+step, ... (n).
// ---------------------------------------------------------------------------------------------
__device__ int wave_window_homology(const uint8_t *__restrict__ msa, int C, const uint16_t *sel, int rn, int first,
                                    int n, int step, double thr) {
    int lane = lane_id();
    const uint8_t *r0 = lane < rn ? msa + (size_t)sel[lane] * C : nullptr;
    const uint8_t *r1 = lane + 64 < rn ? msa + (size_t)sel[lane + 64] * C : nullptr;
    int g0 = 0, g1 = 0;
    for (int i = 0, c = first; i < n; i++, c += step) {
        if (r0) g0 += r0[c] == '-';
        if (r1) g1 += r1[c] == '-';
    }
    bool v0 = r0 && 2 * g0 <= n;  // gap_count <= len(window)/2   (:8846)
    bool v1 = r1 && 2 * g1 <= n;
    int nv = __popcll(__ballot(v0)) + __popcll(__ballot(v1));
    if (nv < 2) return -1;
    double total = 0.0;
    const double lim = thr - 0.1;
    int first_cand = -1;
    for (int i = 0, c = first; i < n; i++, c += step) {
        int k0 = v0 ? sym_class(r0[c]) : 7;
        int k1 = v1 ? sym_class(r1[c]) : 7;
        int best = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            int cnt = __popcll(__ballot(k0 == k)) + __popcll(__ballot(k1 == k));
            best = cnt > best ? cnt : best;
        }
        double ratio = best ? (double)best / (double)nv : 0.0;
        if (ratio >= lim && first_cand == -1) first_cand = c;
        total += ratio;
    }
    double avg = total / (double)n;
    return avg >= thr ? first_cand : -1;
}

// collect up to 100 valid columns into S.cols; mode as in the oracle's scan_valid.  256 positions per round: every wave
// ballots its valid ("1") and out-of-range ("2") positions, keeps the valid ones before its first "2", and places them
// behind the counts of the waves before it (a wave behind a "2" contributes nothing) -- the list is the one a serial walk
// builds (the walk by thread 0 used to be a fifth of the boundary search).
__device__ int blk_scan_valid(const uint8_t *__restrict__ cstat, int C, int vthr, int from, int dir, int mode,
                              JShared &S) {
    __syncthreads();   // readers of the previous list are done
    const int lane = lane_id(), w = wave_id();
    int n = 0, c0 = from;
    for (;;) {
        const int c = c0 + dir * (int)threadIdx.x;
        bool ok;
        switch (mode) {
            case 0: ok = 2 * c < C; break;        // c < C/2 (float)
            case 1: ok = c >= 0; break;
            case 2: ok = c < C; break;
            default: ok = 2 * c >= C; break;      // c >= C/2 (float)
        }
        ok = ok && c >= 0 && c < C;
        const bool one = ok && cstat[(size_t)c * CS + 5] <= vthr;
        const unsigned long long b2 = __ballot(!ok);
        const unsigned long long lim = b2 ? ((1ull << (__ffsll((long long)b2) - 1)) - 1ull) : ~0ull;
        const unsigned long long v1 = __ballot(one) & lim;
        if (lane == 0) { S.scan[w] = __popcll(v1); S.scan[4 + w] = b2 != 0ull; }
        __syncthreads();
        int tot = n, pre = n;
        bool stop = false, dead = false;
#pragma unroll
        for (int q = 0; q < JW; q++) {
            if (q == w) { pre = tot; dead = stop; }
            if (!stop) tot += S.scan[q];
            stop = stop || S.scan[4 + q] != 0;
        }
        if (!dead && ((v1 >> lane) & 1ull)) {
            const int pos = pre + __popcll(v1 & ((1ull << lane) - 1ull));
            if (pos < 100) S.cols[pos] = c;
        }
        n = tot > 100 ? 100 : tot;
        __syncthreads();
        if (stop || n >= 100) break;
        c0 += dir * JB;
    }
    return n;
}

// ---- the window scans with lanes = windows ------------------------------------------------------------------------------
// Row sets as bit masks: for every column of the staged span and every symbol class one mask over the selected rows
// (W32 words of 32 rows), kept where the byte tile used to be.  A thread then evaluates a whole window by itself: the rows
// with <= half gaps through an 8-plane bit-sliced counter over the gap masks, the vote of a column as five and + popcount
// pairs -- ~60 integer operations per window column instead of a wavefront's ballots, and all (<= 91) windows of a scan at
// once instead of four per block round.  Arithmetic and order of the binary64 sums are those of wave_window_homology.
template <int W32>
__device__ __forceinline__ int lane_window_homology(const uint32_t *mk, const uint32_t (&rowmask)[W32], int lo, int first, int n,
                                                    int step, double thr) {
    if (n <= 0) return -1;                       // (0 / 0 in the reference arithmetic: never >= thr)
    uint32_t P[8][W32];
#pragma unroll
    for (int p = 0; p < 8; p++)
#pragma unroll
        for (int w = 0; w < W32; w++) P[p][w] = 0u;
    for (int i = 0, c = first - lo; i < n; i++, c += step) {
        const uint32_t *g = mk + ((size_t)c * 6 + 5) * W32;
#pragma unroll
        for (int w = 0; w < W32; w++) {
            uint32_t carry = g[w];
#pragma unroll
            for (int p = 0; p < 8; p++) { const uint32_t t = P[p][w] & carry; P[p][w] ^= carry; carry = t; }
        }
    }
    // valid rows: gap count <= n / 2  (2 * gaps <= n)
    const int h = n >> 1;
    uint32_t V[W32];
    int nv = 0;
#pragma unroll
    for (int w = 0; w < W32; w++) {
        uint32_t lt = 0u, eq = 0xffffffffu;
#pragma unroll
        for (int p = 7; p >= 0; p--) {
            const uint32_t hb = ((h >> p) & 1) ? 0xffffffffu : 0u;
            lt |= eq & ~P[p][w] & hb;
            eq &= ~(P[p][w] ^ hb);
        }
        V[w] = (lt | eq) & rowmask[w];
        nv += __popc(V[w]);
    }
    if (nv < 2) return -1;
    double total = 0.0;
    const double lim = thr - 0.1;
    int first_cand = -1;
    for (int i = 0, c = first - lo; i < n; i++, c += step) {
        const uint32_t *m = mk + (size_t)c * 6 * W32;
        int best = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            int cnt = 0;
#pragma unroll
            for (int w = 0; w < W32; w++) cnt += __popc(m[k * W32 + w] & V[w]);
            best = cnt > best ? cnt : best;
        }
        const double ratio = best ? (double)best / (double)nv : 0.0;
        if (ratio >= lim && first_cand == -1) first_cand = c + lo;
        total += ratio;
    }
    const double avg = total / (double)n;
    return avg >= thr ? first_cand : -1;
}

template <int W32>
__device__ int blk_first_window_masks(const uint8_t *__restrict__ msa, int C, const uint16_t *sel, int rn, int n, int ws,
                                      bool rev_list, bool desc, double thr, int lo, int span, JShared &S) {
    uint32_t *mk = reinterpret_cast<uint32_t *>(S.tile);
    __syncthreads();
    for (int i = threadIdx.x; i < span * 6 * W32; i += JB) mk[i] = 0u;
    if (threadIdx.x == 0) S.red[6] = 0xffffffffu;
    __syncthreads();
    for (int idx = threadIdx.x; idx < rn * span; idx += JB) {      // consecutive threads on consecutive columns of a row
        const int r = idx / span, c = idx - r * span;
        const int k = S.cls[msa[(size_t)sel[r] * C + lo + c]];
        atomicOr(&mk[((size_t)c * 6 + k) * W32 + (r >> 5)], 1u << (r & 31));
    }
    __syncthreads();
    uint32_t rowmask[W32];
#pragma unroll
    for (int w = 0; w < W32; w++) rowmask[w] = rn >= 32 * (w + 1) ? 0xffffffffu : (rn > 32 * w ? (1u << (rn - 32 * w)) - 1u : 0u);
    const int nwin = n - ws + 1;
    for (int i = threadIdx.x; i < nwin; i += JB) {
        const int a = rev_list ? S.cols[n - 1 - i] : S.cols[i];
        const int b = rev_list ? S.cols[n - 1 - (i + ws - 1)] : S.cols[i + ws - 1];
        const int r = !desc ? lane_window_homology<W32>(mk, rowmask, lo, a, b - a + 1, +1, thr)
                            : lane_window_homology<W32>(mk, rowmask, lo, a, a - b - 1, -1, thr);
        if (r != -1) atomicMin(&S.red[6], ((unsigned)i << 16) | (unsigned)r);
    }
    __syncthreads();
    const unsigned key = S.red[6];
    __syncthreads();
    return key == 0xffffffffu ? -1 : (int)(key & 0xffffu);
}

// first homologous window over S.cols.  rev_list: logical list is S.cols reversed.
// desc: logical list is descending (the 'end' side: range(first, last+1, -1)).
__device__ int blk_first_window(const uint8_t *__restrict__ msa, int C, const uint16_t *sel, int rn, int n, int ws,
                                bool rev_list, bool desc, double thr, JShared &S) {
    int nwin = n - ws + 1;
    int w = wave_id();
    int found = -1;
    // every window is a contiguous column range inside [lo, hi]
    const int e0 = S.cols[0], e1 = S.cols[n - 1];   // the list is monotonic (either direction)
    const int lo = e0 < e1 ? e0 : e1, hi = e0 < e1 ? e1 : e0;
    const int span = hi - lo + 1;
    if (span <= TILE_COLS && rn <= MAXSEL && nwin < 65536) {
        if (rn <= 32) return blk_first_window_masks<1>(msa, C, sel, rn, n, ws, rev_list, desc, thr, lo, span, S);
        if (rn <= 64) return blk_first_window_masks<2>(msa, C, sel, rn, n, ws, rev_list, desc, thr, lo, span, S);
        return blk_first_window_masks<4>(msa, C, sel, rn, n, ws, rev_list, desc, thr, lo, span, S);
    }
    // wider spans (more than 60 invalid columns among 100 valid ones): one window per wavefront on the alignment itself
    for (int base = 0; base < nwin; base += JW) {
        int i = base + w;
        int r = -1;
        if (i < nwin) {
            int a = rev_list ? S.cols[n - 1 - i] : S.cols[i];
            int b = rev_list ? S.cols[n - 1 - (i + ws - 1)] : S.cols[i + ws - 1];
            if (!desc) r = wave_window_homology(msa, C, sel, rn, a, b - a + 1, +1, thr);
            else r = wave_window_homology(msa, C, sel, rn, a, a - b - 1, -1, thr);
        }
        __syncthreads();
        if (lane_id() == 0) S.res[w] = r;
        __syncthreads();
        for (int q = 0; q < JW; q++) if (found == -1 && S.res[q] != -1) found = S.res[q];
        if (found != -1) break;
    }
    __syncthreads();
    return found;
}

// search_boundary_homo_v3   (side 0 'start', 1 'end')
__device__ int blk_search_v3(const uint8_t *__restrict__ msa, const uint8_t *__restrict__ cstat, int C,
                             const uint16_t *sel, int rn, int pos, int side, double thr, int win_in, int win_out,
                             JShared &S) {
    int vthr = rn / 2;
    int cur, n, ws, nb;
    if (side == 0) {
        n = blk_scan_valid(cstat, C, vthr, pos, +1, 0, S);
        ws = n < win_in ? n : win_in;
        JCLK(5);   // (search) valid-column scans
        if (ws < 10) cur = -1;
        else cur = blk_first_window(msa, C, sel, rn, n, ws, false, false, thr, S);
        JCLK(6);   // (search) windows
        n = blk_scan_valid(cstat, C, vthr, cur, -1, 1, S);
        ws = n < win_out ? n : win_out;
        if (ws < 10) cur = -1;
        else {
            nb = blk_first_window(msa, C, sel, rn, n, ws, true, false, thr, S);
            if (nb != -1) cur = nb < 10 ? -1 : nb;
        }
    } else {
        n = blk_scan_valid(cstat, C, vthr, pos, +1, 2, S);
        cur = pos;
        ws = n < win_out ? n : win_out;
        if (ws < 10) cur = -1;
        else {
            nb = blk_first_window(msa, C, sel, rn, n, ws, true, true, thr, S);
            if (nb != -1) cur = (C - nb < 10) ? -1 : nb;
        }
        n = blk_scan_valid(cstat, C, vthr, cur, -1, 3, S);
        ws = n < win_in ? n : win_in;
        if (ws < 10) cur = -1;
        else cur = blk_first_window(msa, C, sel, rn, n, ws, false, true, thr, S);
    }
    return cur;
}

// stored max_homo_ratio of a column (Util.py:8604-8613): symbols in first-appearance order
__device__ double stored_max_ratio(const uint8_t *cs, int rn, double thr) {
    double mx = 0.0;
    unsigned done = 0;
    for (int it = 0; it < 5; it++) {
        int bk = -1, bf = 256;
        for (int k = 0; k < 5; k++)
            if (!((done >> k) & 1u) && cs[k] && cs[6 + k] < bf) { bf = cs[6 + k]; bk = k; }
        if (bk < 0) break;
        done |= 1u << bk;
        double r = (double)cs[bk] / (double)rn;
        if (r > mx) mx = r;
        if (r >= thr) break;
    }
    return mx;
}

// search_boundary_homo_v4; returns boundary, *valid via S.iv[8]
__device__ int blk_search_v4(const uint8_t *__restrict__ msa, const uint8_t *__restrict__ cstat, int C,
                             const uint16_t *sel, int rn, int pos, int side, double thr, double int_thr,
                             double out_thr, int win_in, int win_out, JShared &S, int *valid) {
    int vthr = rn / 2;
    int cur, n, ws, nb;
    if (side == 0) {
        n = blk_scan_valid(cstat, C, vthr, pos, +1, 0, S);
        cur = pos;
        ws = n < win_in ? n : win_in;
        if (ws < 10) cur = -1;
        else {
            nb = blk_first_window(msa, C, sel, rn, n, ws, false, false, thr, S);
            if (nb != -1) cur = nb;
        }
        n = blk_scan_valid(cstat, C, vthr, cur - 1, -1, 1, S);
        ws = n < win_out ? n : win_out;
        if (ws < 10) cur = -1;
        else {
            nb = blk_first_window(msa, C, sel, rn, n, ws, true, false, thr, S);
            if (nb != -1) cur = nb < 10 ? -1 : nb;
        }
        *valid = 1;
        return cur;
    }
    n = blk_scan_valid(cstat, C, vthr, pos + 1, +1, 2, S);
    ws = n < win_out ? n : win_out;
    if (ws < 10) { *valid = 0; return -1; }
    nb = -1;
    {
        double s = 0.0;
        for (int i = 0; i < ws; i++) s += stored_max_ratio(cstat + (size_t)S.cols[i] * CS, rn, thr);
        s = s / (double)ws;
        if (s >= out_thr) nb = S.cols[ws - 1];
    }
    if (nb != -1) {
        if (C - nb < 10) { *valid = 0; return -1; }
        else if (nb != pos) { *valid = 0; return -1; }
    }
    n = blk_scan_valid(cstat, C, vthr, pos, -1, 3, S);
    ws = n < win_in ? n : win_in;
    if (ws < 10) { *valid = 0; return -1; }
    nb = -1;
    {
        double s = 0.0;
        for (int i = 0; i < ws; i++) s += stored_max_ratio(cstat + (size_t)S.cols[i] * CS, rn, thr);
        s = s / (double)ws;
        if (s < int_thr) nb = S.cols[ws - 1];
    }
    if (nb != pos && nb != -1) { *valid = 0; return -1; }
    *valid = 1;
    return pos;
}

// majority consensus of a column (Util.py:9314-9339); mode 0 v5/v9, 1 v6 ('N' when no majority).
// returns 0 when the column contributes nothing.
__device__ __forceinline__ uint8_t cons_col(const uint8_t *cs, int rn, int mode, int *best_all_cnt,
                                            int *best_all_sym) {
    // the 12 bytes of a column's statistics as three aligned words (cstat is 16-byte aligned, CS = 12), the symbol of a class
    // from a packed constant: twelve byte loads and a string look-up per column were a visible part of the consensus phase
    const uint32_t *cw = reinterpret_cast<const uint32_t *>(cs);
    const uint32_t w0 = cw[0], w1 = cw[1], w2 = cw[2];
    const unsigned long long cnt = ((unsigned long long)(w1 & 0xffffu) << 32) | w0;                   // counts of classes 0..5
    const unsigned long long fst = ((unsigned long long)w2 << 16) | (w1 >> 16);                        // first-appearance ranks 0..5
    const unsigned long long syms = 0x2d4e54474341ull;                                                 // "ACGTN-"
    int best = 0, bk = -1, bf = 256;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int c = (int)(cnt >> (8 * k)) & 0xff, f = (int)(fst >> (8 * k)) & 0xff;
        if (c > best || (c == best && c > 0 && f < bf)) { best = c; bk = k; bf = f; }
    }
    if (best_all_cnt) { *best_all_cnt = best; *best_all_sym = bk; }
    if (best >= rn / 2) return bk != 5 ? (uint8_t)(syms >> (8 * bk)) : 0;
    if (mode == 1) return 'N';
    best = 0; bk = -1; bf = 256;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int c = (int)(cnt >> (8 * k)) & 0xff, f = (int)(fst >> (8 * k)) & 0xff;
        if (c > best || (c == best && c > 0 && f < bf)) { best = c; bk = k; bf = f; }
    }
    return bk >= 0 ? (uint8_t)(syms >> (8 * bk)) : 0;
}

// ordered compaction of the consensus of columns [hs, he] into out; returns length
__device__ int blk_consensus(const uint8_t *__restrict__ cstat, int rn, int hs, int he, int mode,
                             uint8_t *__restrict__ out, JShared &S) {
    int running = 0;
    for (int base = hs; base <= he; base += JB) {
        int c = base + threadIdx.x;
        uint8_t b = 0;
        if (c <= he) b = cons_col(cstat + (size_t)c * CS, rn, mode, nullptr, nullptr);
        int tot;
        int pre = block_excl_scan(b ? 1 : 0, S.scan, &tot);
        if (b) out[running + pre] = b;
        running += tot;
        __syncthreads();
    }
    return running;
}

// get_boundary_ungap_str (Util.py:2407) -- returns count, chars in out (order as the reference)
__device__ int ungap_str(const uint8_t *row, int C, int pos, int want, bool right, uint8_t *out, bool *exc) {
    int n = 0, c = pos;
    if (right) {
        while (n < want && c < C) {
            int cc = c;
            if (cc < 0) { cc += C; if (cc < 0) { *exc = true; return 0; } }
            uint8_t ch = row[cc];
            if (ch != '-') out[n++] = ch;
            c++;
        }
    } else {
        uint8_t tmp[12];
        while (n < want && c >= 0) {
            if (c >= C) { *exc = true; return 0; }
            uint8_t ch = row[c];
            if (ch != '-') tmp[n++] = ch;
            c--;
        }
        for (int i = 0; i < n; i++) out[i] = tmp[n - 1 - i];
    }
    return n;
}
__device__ __forceinline__ bool eqs(const uint8_t *a, const char *b, int n) {
    for (int i = 0; i < n; i++) if (a[i] != (uint8_t)b[i]) return false;
    return true;
}

// TSDsearch_v5 (Util.py:2460-2492): returns TSD length (0 none, -1 exception); left/right optional.
// The reference ungaps the two flanks once per length; the k characters it gets are the nearest k bases, i.e. the
// suffix (left flank) / prefix (right flank) of the 11-base strings, so both flanks are walked ONCE (the walks are chains
// of dependent byte loads: this is what the TSD votes of the judge spend their time on).
__device__ int tsd_search_v5(const uint8_t *row, int C, int bs, int be, int plant, uint8_t *left, uint8_t *right) {
    const int lens[9] = {11, 10, 9, 8, 6, 5, 4, 3, 2};
    uint8_t f5[5], l5[5], L11[12], R11[12];
    bool exc = false;
    const int nf5 = ungap_str(row, C, bs, 5, true, f5, &exc);
    const int nl5 = ungap_str(row, C, be, 5, false, l5, &exc);
    // first 3 = prefix of the first 5; last 3 (walking left from be) = suffix of the last 5
    const int nf3 = nf5 < 3 ? nf5 : 3, nl3 = nl5 < 3 ? nl5 : 3;
    const uint8_t *f3 = f5, *l3 = l5 + (nl5 - nl3);
    if (exc) return -1;
    const int nL = ungap_str(row, C, bs - 1, 11, false, L11, &exc);   // nearest <= 11 bases left of bs, in sequence order
    const int nR = ungap_str(row, C, be + 1, 11, true, R11, &exc);    // nearest <= 11 bases right of be
    if (exc) return -1;
    int found = 0;
    for (int t = 0; t < 9; t++) {
        const int k = lens[t];
        if (nL < k || nR < k) continue;        // the reference gets fewer than k characters on one side
        const uint8_t *lt = L11 + (nL - k), *rt = R11;
        bool same = true;
        int mm = 0;
        for (int i = 0; i < k; i++) if (lt[i] != rt[i]) { same = false; mm++; }
        bool ok = false;
        if (same) {
            if (k != 2 && k != 3 && k != 4) ok = true;
            else if (k == 4) ok = eqs(lt, "TTAA", 4);
            else if (k == 2)
                ok = eqs(lt, "TA", 2) || (plant == 0 && nf3 == 3 && eqs(f3, "CCC", 3) && nl3 == 3 && eqs(l3, "GGG", 3));
            else
                ok = eqs(lt, "TAA", 3) || eqs(lt, "TTA", 3) ||
                     (plant == 1 && nf5 == 5 && nl5 == 5 &&
                      ((eqs(f5, "CACTA", 5) && eqs(l5, "TAGTG", 5)) || (eqs(f5, "CACTG", 5) && eqs(l5, "CAGTG", 5))));
        } else if (k >= 8) ok = mm <= 1;
        if (ok) found = k;   // the last (shortest) accepted length wins
    }
    if (left && found > 0)
        for (int i = 0; i < found; i++) { left[i] = L11[nL - found + i]; right[i] = R11[i]; }
    return found;
}

__device__ int lev_small(const uint8_t *a, int n, const uint8_t *b, int m) {  // n, m <= 5
    int prev[6], cur[6];
    for (int j = 0; j <= m; j++) prev[j] = j;
    for (int i = 1; i <= n; i++) {
        cur[0] = i;
        for (int j = 1; j <= m; j++) {
            int c = prev[j - 1] + (a[i - 1] != b[j - 1]);
            int d = prev[j] + 1, e = cur[j - 1] + 1;
            cur[j] = c < d ? (c < e ? c : e) : (d < e ? d : e);
        }
        for (int j = 0; j <= m; j++) prev[j] = cur[j];
    }
    return prev[m];
}
__device__ __forceinline__ bool starts_with(const uint8_t *s, int n, const char *p, int k) { return n >= k && eqs(s, p, k); }
__device__ __forceinline__ bool ends_with(const uint8_t *s, int n, const char *p, int k) { return n >= k && eqs(s + n - k, p, k); }

__device__ __forceinline__ double homo_thr(int rn, double big) { return rn <= 2 ? 0.95 : (rn <= 5 ? 0.9 : big); }
__device__ __forceinline__ double int_thr_tab(int rn) { return rn <= 2 ? 0.9 : (rn <= 5 ? 0.85 : 0.65); }

// rows with a base within +-alen of both anchors (Util.py:9196-9217 / 10015-10033); need_start /
// need_end select which tests apply (v6 builds start-only and end-only sets too).
__device__ int blk_select_rows(const uint8_t *__restrict__ msa, int R, int C, int astart, int aend, int alen,
                               bool need_start, bool need_end, uint16_t *sel, JShared &S) {
    __syncthreads();
    if (threadIdx.x == 0) S.iv[0] = 0;
    __syncthreads();
    for (int base = 0; base < R; base += JB) {
        int r = base + threadIdx.x;
        uint8_t f = 0;
        if (r < R) {
            const uint8_t *row = msa + (size_t)r * C;
            bool okS = true, okE = true;
            if (need_start) {
                int lo, hi;
                py_slice(astart - alen >= 0 ? astart - alen : 0, (int64_t)astart + alen, C, &lo, &hi);
                okS = false;
                for (int c = lo; c < hi; c++) if (row[c] != '-') { okS = true; break; }
            }
            if (need_end) {
                int lo, hi;
                py_slice((int64_t)aend - alen, (aend + alen < C) ? aend + alen : C, C, &lo, &hi);
                okE = false;
                for (int c = lo; c < hi; c++) if (row[c] != '-') { okE = true; break; }
            }
            f = okS && okE;
        }
        S.flag[threadIdx.x] = f;
        __syncthreads();
        if (threadIdx.x == 0) {
            int n = S.iv[0];
            for (int t = 0; t < JB && base + t < R; t++) {
                if (n > 100) break;
                if (S.flag[t]) sel[n++] = (uint16_t)(base + t);
            }
            S.iv[0] = n;
        }
        __syncthreads();
        if (S.iv[0] > 100) break;
    }
    return S.iv[0];
}

// ---------------------------------------------------------------------------------------------
// the judge kernel: dynamic work queue over candidates, one block per candidate at a time
// ---------------------------------------------------------------------------------------------
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
    unsigned int *counter;
};

__device__ void judge_tir_tail(const JudgeParams &P, const uint8_t *msa, int R, int C, const uint8_t *cstat, int rn,
                               int hs, int he, uint8_t *model, int64_t cons_base, hite_call &out, JShared &S);
__device__ void judge_v9_tail(const JudgeParams &P, const uint8_t *msa, int R, int C, const uint8_t *cstat, int rn,
                              int hs, int he, uint8_t *model, hite_call &out, JShared &S, uint8_t *slot);
__device__ void judge_v6_body(const JudgeParams &P, const uint8_t *msa, int R, int C, int astart, int aend,
                              uint8_t *cstat, uint8_t *model, hite_call &out, JShared &S);

__global__ void __launch_bounds__(JB) __attribute__((amdgpu_waves_per_eu(5, 8))) judge_kernel(JudgeParams P) {
    __shared__ JShared S;
    jshared_init(S);
#ifdef JUDGE_CLOCKS
    if (threadIdx.x == 0) S.jt = wall_clock64();
#endif
    uint8_t *slot = P.scratch + (size_t)blockIdx.x * P.slot_bytes;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) S.iv[15] = (int)atomicAdd(P.counter, 1u);
        __syncthreads();
        int ci = S.iv[15];
        if (ci >= P.n) break;
        const int R = P.rows[ci], C = P.cols[ci];
        const uint8_t *msa = P.msa + P.msa_off[ci];
        const uint8_t *cand = P.cand + P.cand_off[ci];
        const int clen = (int)(P.cand_off[ci + 1] - P.cand_off[ci]);
        const int64_t cons_base = P.col_off[ci] + 8 * (int64_t)ci;
        uint8_t *model = P.cons + cons_base;
        hite_call call;
        call.is_te = 0; call.info = HITE_INFO_NONE; call.row_num = 0; call.bstart = -1; call.bend = -1;
        call.cons_len = 0; call.cons_off = cons_base;
        // slot layout (multiples of maxC16): ung 1 | reflex 4 | minfo 2 | cstat 12 | wave bufs 4 | rowres
        // the ungapped row and the 2-byte match records of the anchor search live in LDS (where the row-set masks of the
        // window scans go later) when the alignment has <= ANCHOR_LDS_COLS columns: the bit-parallel scan reads the text one
        // character at a time, a chain of dependent loads that global scratch made ~10x longer
        const bool anchors_in_lds = C <= ANCHOR_LDS_COLS;
        uint8_t *ung = anchors_in_lds ? S.tile : slot;
        uint8_t *minfo = anchors_in_lds ? S.tile + ANCHOR_LDS_COLS + 16 : slot + 5 * P.maxC16;
        uint8_t *cstat = slot + 7 * P.maxC16;
        bool done = false;
        if (R <= 0 || C <= 0 || clen <= 0) { call.info = HITE_INFO_EXC; done = true; }
        int astart = -1, aend = -1;
        if (!done) {
            // anchor patterns: cur_seq[0:20], cur_seq[-20:]
            int m1 = clen < 20 ? clen : 20;
            if (threadIdx.x < 20) {
                if ((int)threadIdx.x < m1) { S.pat[0][threadIdx.x] = cand[threadIdx.x]; S.pat[1][threadIdx.x] = cand[clen - m1 + threadIdx.x]; }
            }
            __syncthreads();
            if (P.te_type != HITE_TE_HELITRON) {
                // first row that has both anchors (Util.py:9158-9181)
                for (int r = 0; r < R; r++) {
                    UngapSeg G;
                    int n = blk_ungap_row_seg(msa + (size_t)r * C, C, ung, G, S);
                    int fs = blk_fnm(S.pat[0], m1, ung, n, 2, minfo, 0, S);
                    if (fs < 0) continue;
                    int le = blk_fnm(S.pat[1], m1, ung, n, 2, minfo, 1, S);
                    if (le < 0) continue;
                    astart = blk_col_of(msa + (size_t)r * C, C, G, fs, S);
                    aend = blk_col_of(msa + (size_t)r * C, C, G, le - 1, S);
                    break;
                }
            } else {
                // mode over all rows (Util.py:9837-9866): Counter.most_common(1), first inserted wins ties.
                int *as = (int *)(slot + 23 * P.maxC16);  // rowres region: 16 bytes per row
                int *ae = as + R;
                int na = 0;
                for (int r = 0; r < R; r++) {
                    UngapSeg G;
                    int n = blk_ungap_row_seg(msa + (size_t)r * C, C, ung, G, S);
                    int fs = blk_fnm(S.pat[0], m1, ung, n, 2, minfo, 0, S);
                    if (fs < 0) continue;
                    int le = blk_fnm(S.pat[1], m1, ung, n, 2, minfo, 1, S);
                    if (le < 0) continue;
                    const int cs_ = blk_col_of(msa + (size_t)r * C, C, G, fs, S), ce_ = blk_col_of(msa + (size_t)r * C, C, G, le - 1, S);
                    if (threadIdx.x == 0) { as[na] = cs_; ae[na] = ce_; }
                    na++;
                }
                __syncthreads();
                if (threadIdx.x == 0) {
                    int bs_ = -1, be_ = -1, bc = 0;
                    for (int i = 0; i < na; i++) { int c = 0; for (int j = 0; j < na; j++) c += as[j] == as[i]; if (c > bc) { bc = c; bs_ = as[i]; } }
                    bc = 0;
                    for (int i = 0; i < na; i++) { int c = 0; for (int j = 0; j < na; j++) c += ae[j] == ae[i]; if (c > bc) { bc = c; be_ = ae[i]; } }
                    S.iv[2] = bs_; S.iv[3] = be_;
                }
                __syncthreads();
                astart = S.iv[2]; aend = S.iv[3];
            }
            if (astart == -1 || aend == -1) { call.info = HITE_INFO_NB; done = true; }
        }
        JCLK(0);   // anchors
        if (!done && P.te_type != HITE_TE_HELITRON) {
            int rn = blk_select_rows(msa, R, C, astart, aend, 10, true, true, S.sel, S);
            JCLK(1);   // select rows
            if (rn == 0) { call.info = HITE_INFO_EXC; done = true; }
            else if (rn <= 1) { call.info = HITE_INFO_FL1; call.row_num = (P.te_type == HITE_TE_TIR) ? 1 : rn; done = true; }
            if (!done) {
                call.row_num = rn;
                blk_colstats(msa, C, S.sel, rn, cstat, S.cls);
                JCLK(2);   // column statistics
                double thr = homo_thr(rn, P.te_type == HITE_TE_TIR ? 0.7 : 0.8);
                int hs = blk_search_v3(msa, cstat, C, S.sel, rn, astart, 0, thr, 20, 10, S);
                int he = -1;
                if (hs != -1) he = blk_search_v3(msa, cstat, C, S.sel, rn, aend, 1, thr, 20, 10, S);
                JCLK(3);   // boundary search
                if (hs != -1 && he != -1) {
                    if (P.te_type == HITE_TE_TIR) {
                        judge_tir_tail(P, msa, R, C, cstat, rn, hs, he, model, cons_base, call, S);
                    } else {
                        judge_v9_tail(P, msa, R, C, cstat, rn, hs, he, model, call, S, slot);
                    }
                }
            }
        }
        if (!done && P.te_type == HITE_TE_HELITRON) {
            judge_v6_body(P, msa, R, C, astart, aend, cstat, model, call, S);
        }
        __syncthreads();
        JCLK(4);   // tail
        if (threadIdx.x == 0) P.calls[ci] = call;
    }
}

// ---------------------------------------------------------------------------------------------
// judge_boundary_v5 tail: consensus, TA/TTAA trims x TSD votes, Levenshtein ranking
// Util.py:9311-9413
// ---------------------------------------------------------------------------------------------
__device__ void judge_tir_tail(const JudgeParams &P, const uint8_t *msa, int R, int C, const uint8_t *cstat, int rn,
                               int hs, int he, uint8_t *model, int64_t cons_base, hite_call &out, JShared &S) {
    // valid left / right boundary  (:9269-9294)   gap <= row_num / 2 (float)
    __syncthreads();
    if (threadIdx.x == 0) { S.red[4] = 0xffffffffu; S.red[5] = 0u; }
    __syncthreads();
    {
        unsigned lmin = 0xffffffffu, lmax = 0u;
        for (int c = threadIdx.x; c < C; c += JB)
            if (2 * (int)cstat[(size_t)c * CS + 5] <= rn) { if ((unsigned)c < lmin) lmin = (unsigned)c; lmax = (unsigned)c + 1u; }
        if (lmax) { atomicMin(&S.red[4], lmin); atomicMax(&S.red[5], lmax); }
    }
    __syncthreads();
    int vl = S.red[4] == 0xffffffffu ? -1 : (int)S.red[4];
    int vr = (int)S.red[5] - 1;
    if (!(vl != -1 && vr != -1 && vl < vr)) { vl = -1; vr = -1; }
    JCLK(7);   // (tail) valid range
    int ml = blk_consensus(cstat, rn, hs, he, 0, model, S);
    __syncthreads();
    JCLK(8);   // (tail) consensus
    if (hs <= vl || he >= vr) return;  // :9353  final_cons_seq = ''
    if (threadIdx.x == 0) {
        int nfo = 0, neo = 0;
        S.fo[nfo++] = 0; S.eo[neo++] = 0;
        if (starts_with(model, ml, "A", 1)) S.fo[nfo++] = 1;
        if (starts_with(model, ml, "AA", 2) || starts_with(model, ml, "TA", 2)) S.fo[nfo++] = 2;
        if (starts_with(model, ml, "TAA", 3) || starts_with(model, ml, "TTA", 3)) S.fo[nfo++] = 3;
        if (starts_with(model, ml, "TTAA", 4)) S.fo[nfo++] = 4;
        if (ends_with(model, ml, "T", 1)) S.eo[neo++] = 1;
        if (ends_with(model, ml, "TT", 2) || ends_with(model, ml, "TA", 2)) S.eo[neo++] = 2;
        if (ends_with(model, ml, "TAA", 3) || ends_with(model, ml, "TTA", 3)) S.eo[neo++] = 3;
        if (ends_with(model, ml, "TTAA", 4)) S.eo[neo++] = 4;
        S.iv[4] = nfo; S.iv[5] = neo; S.iv[6] = 0;  // iv[6] = exception flag
        for (int i = 0; i < 25; i++) S.tsd[i] = 0;
    }
    __syncthreads();
    int nfo = S.iv[4], neo = S.iv[5];
    int ntask = nfo * neo * R;
    for (int t = threadIdx.x; t < ntask; t += JB) {
        int r = t % R, ab = t / R;
        int a = ab / neo, b = ab % neo;
        int cs = hs + S.fo[a], ce = he - S.eo[b];
        int i1 = cs < 0 ? cs + C : cs, i2 = ce < 0 ? ce + C : ce;
        if (i1 < 0 || i1 >= C || i2 < 0 || i2 >= C) { S.iv[6] = 1; continue; }
        const uint8_t *row = msa + (size_t)r * C;
        if (row[i1] == '-' || row[i2] == '-') continue;
        int k = tsd_search_v5(row, C, cs, ce, P.plant, nullptr, nullptr);
        if (k < 0) S.iv[6] = 1;
        else if (k > 0) atomicAdd(&S.tsd[a * 5 + b], 1);
    }
    __syncthreads();
    JCLK(9);   // (tail) TSD votes
    if (S.iv[6]) { out.info = HITE_INFO_EXC; return; }
    if (threadIdx.x == 0) {
        int have = 0, b_ed = 0, b_tc = 0, b_f = 0, b_e = 0;
        for (int a = 0; a < nfo; a++)
            for (int b = 0; b < neo; b++) {
                int tsd = S.tsd[a * 5 + b];
                if (tsd <= 0) continue;
                int lo, hi, lo2, hi2;
                py_slice(S.fo[a], S.fo[a] + 5, ml, &lo, &hi);
                if (S.eo[b] == 0) py_slice(-5, ml, ml, &lo2, &hi2);
                else py_slice((int64_t)ml - 5 - S.eo[b], (int64_t)ml - S.eo[b], ml, &lo2, &hi2);
                uint8_t rc[5];
                int n1 = hi - lo;
                for (int i = 0; i < n1; i++) rc[i] = comp_sym(model[hi - 1 - i]);
                int ed = lev_small(rc, n1, model + lo2, hi2 - lo2);
                if (!have || ed < b_ed || (ed == b_ed && tsd > b_tc)) { have = 1; b_ed = ed; b_tc = tsd; b_f = S.fo[a]; b_e = S.eo[b]; }
            }
        S.iv[7] = have; S.iv[8] = b_f; S.iv[9] = b_e;
    }
    __syncthreads();
    if (S.iv[7]) {
        int b_f = S.iv[8], b_e = S.iv[9];
        int lo, hi;
        if (b_e != 0) py_slice(b_f, -(int64_t)b_e, ml, &lo, &hi);
        else py_slice(b_f, ml, ml, &lo, &hi);
        out.cons_off = cons_base + lo;
        out.cons_len = hi - lo;
        out.bstart = hs + b_f;
        out.bend = he - b_e;
        out.is_te = (hi - lo) > 0;
    }
}

// ---------------------------------------------------------------------------------------------
// judge_boundary_v9 tail (non-LTR)  Util.py:9599-9716
// Rows are analysed one per wavefront (4 at a time): cooperative ungap into a wave-private
// buffer, then polyA / tandem tail and the 8-20 bp TSD search with lanes = candidate offsets.
// ---------------------------------------------------------------------------------------------
__device__ void judge_v9_tail(const JudgeParams &P, const uint8_t *msa, int R, int C, const uint8_t *cstat, int rn,
                              int hs, int he, uint8_t *model, hite_call &out, JShared &S, uint8_t *slot) {
    // wave-private ungapped row buffers live after the 32*C block scratch: 4 x C bytes
    int lane = lane_id(), w = wave_id();
    uint8_t *u = slot + (19 + (size_t)w) * P.maxC16;
    int *rowres = (int *)(slot + 23 * P.maxC16);
    __syncthreads();
    for (int r0 = 0; r0 < R; r0 += JW) {
        int r = r0 + w;
        if (r < R) {
            const uint8_t *row = msa + (size_t)r * C;
            int n = 0, end_5 = 0, h3 = 0;
            for (int base = 0; base < C; base += 64) {
                int c = base + lane;
                uint8_t ch = c < C ? row[c] : (uint8_t)'-';
                bool f = ch != '-';
                unsigned long long bal = __ballot(f);
                int pre = __popcll(bal & ((1ull << lane) - 1ull));
                if (f) u[n + pre] = ch;
                if (hs >= base && hs < base + 64) end_5 = n + __popcll(bal & ((1ull << (hs - base)) - 1ull));
                if (he >= base && he < base + 64) h3 = n + __popcll(bal & ((1ull << (he - base)) - 1ull));
                n += __popcll(bal);
            }
            __builtin_amdgcn_wave_barrier();
            int sl = h3 + 10; if (sl > n) sl = n; if (sl < 0) sl = 0;
            // find_tail_polyA: last i with u[i:i+6] == 'AAAAAA', i <= sl-6   (Util.py:10832)
            int end_3 = -1;
            for (int top = sl - 6; top >= 0 && end_3 == -1; top -= 64) {
                int i = top - lane;
                bool hit = i >= 0 && u[i] == 'A' && u[i + 1] == 'A' && u[i + 2] == 'A' && u[i + 3] == 'A' && u[i + 4] == 'A' && u[i + 5] == 'A';
                unsigned long long bal = __ballot(hit);
                if (bal) end_3 = top - (__ffsll((long long)bal) - 1) + 6;
            }
            if (end_3 == -1) {
                // find_longest_tandem_repeat_tail (Util.py:9732): tiny, done redundantly by every lane
                int tl = sl >= 30 ? 30 : sl;
                const uint8_t *tail = u + (sl - tl);
                int best_len = 0;
                for (int ul = 2; ul <= 6; ul++)
                    for (int st = tl - ul * 4; st >= 0; st--) {
                        int rep = 1;
                        for (int i = 1; i < (tl - st) / ul; i++) {
                            bool eq = true;
                            for (int q = 0; q < ul; q++) if (tail[st + i * ul + q] != tail[st + q]) { eq = false; break; }
                            if (!eq) break;
                            rep++;
                        }
                        if (rep >= 4 && ul * rep > best_len) { best_len = ul * rep; end_3 = sl - tl + st + ul * rep; }
                    }
            }
            int dlt = end_3 - h3; if (dlt < 0) dlt = -dlt;
            int found = 0, e5 = end_5;
            if (dlt <= 10 && end_3 != -1) {
                int left_pos = end_5 - 50 > 0 ? end_5 - 50 : 0;
                int sublen = end_5 - left_pos;
                for (int kk = 20; kk >= 8 && !found; kk--) {
                    if (end_3 + kk > n) continue;  // k == len(TSD)
                    int i = lane;                   // sublen <= 50 < 64
                    bool hit = false;
                    if (i + kk <= sublen) {
                        // find_near_matches(TSD, kmer, 1) non-empty, both of length kk: substring lengths kk-1..kk
                        const uint8_t *t = u + left_pos + i;
                        int d[5];
                        banded_dist(u + end_3, kk, t, kk, 1, d);
                        hit = d[2] <= 1 || d[1] <= 1;           // L = kk, L = kk-1 from start 0
                        if (!hit) { banded_dist(u + end_3, kk, t + 1, kk - 1, 1, d); hit = d[1] <= 1; }  // start 1, L = kk-1
                    }
                    unsigned long long bal = __ballot(hit);
                    if (bal) { found = 1; e5 = left_pos + (__ffsll((long long)bal) - 1) + kk; }
                }
            }
            if (lane == 0) {
                int col = -2, nonempty = 0;
                if (found) {
                    int fs = e5 < end_3 ? e5 : end_3, fe = e5 < end_3 ? end_3 : e5;
                    nonempty = fe - fs > 0;
                    col = -1;
                    if (fs >= 0 && fs < n) {  // nogap_to_gap[fs]
                        int q = 0;
                        for (int c = 0; c < C; c++) if (row[c] != '-') { if (q == fs) { col = c; break; } q++; }
                    }
                }
                rowres[3 * r] = found; rowres[3 * r + 1] = col; rowres[3 * r + 2] = nonempty;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tsd_count = 0, have_first = 0, hs2 = hs, exc = 0;
        for (int r = 0; r < R && !exc; r++) {
            if (!rowres[3 * r]) continue;
            tsd_count++;
            if (!have_first) {
                if (rowres[3 * r + 2]) have_first = 1;
                if (rowres[3 * r + 1] < 0) exc = 1;  // KeyError nogap_to_gap
                else hs2 = rowres[3 * r + 1];
            }
        }
        S.iv[4] = tsd_count; S.iv[5] = hs2; S.iv[6] = exc;
    }
    __syncthreads();
    if (S.iv[6]) { out.info = HITE_INFO_EXC; return; }
    int tsd_count = S.iv[4];
    hs = S.iv[5];
    if (tsd_count >= 5 || 2 * tsd_count > rn) {
        int ml = blk_consensus(cstat, rn, hs, he, 0, model, S);
        out.cons_len = ml;
        out.bstart = hs; out.bend = he;
        out.is_te = ml >= 80;
    }
}

// ---------------------------------------------------------------------------------------------
// judge_boundary_v6 (Helitron)  Util.py:9880-10159
// ---------------------------------------------------------------------------------------------
__device__ int find_sub(const uint8_t *s, int n, const char *p, int k, bool last) {
    int res = -1;
    for (int i = 0; i + k <= n; i++) if (eqs(s + i, p, k)) { res = i; if (!last) return res; }
    return res;
}

__device__ void judge_v6_body(const JudgeParams &P, const uint8_t *msa, int R, int C, int astart, int aend,
                              uint8_t *cstat, uint8_t *model, hite_call &out, JShared &S) {
    // start-only / end-only row sets, both stop once either exceeds 100 (Util.py:9889-9936).
    // Build flags per row, then one ordered pass reproduces the coupled break.
    __shared__ uint16_t s_start[MAXSEL], s_end[MAXSEL];
    __syncthreads();
    if (threadIdx.x == 0) { S.iv[4] = 0; S.iv[5] = 0; S.iv[6] = 0; }
    __syncthreads();
    for (int base = 0; base < R; base += JB) {
        int r = base + threadIdx.x;
        uint8_t f = 0;
        if (r < R) {
            const uint8_t *row = msa + (size_t)r * C;
            int lo, hi;
            py_slice(astart - 1 >= 0 ? astart - 1 : 0, (int64_t)astart + 1, C, &lo, &hi);
            for (int c = lo; c < hi; c++) if (row[c] != '-') { f |= 1; break; }
            py_slice((int64_t)aend - 1, (aend + 1 < C) ? aend + 1 : C, C, &lo, &hi);
            for (int c = lo; c < hi; c++) if (row[c] != '-') { f |= 2; break; }
        }
        S.flag[threadIdx.x] = f;
        __syncthreads();
        if (threadIdx.x == 0) {
            int ns = S.iv[4], ne = S.iv[5];
            for (int t = 0; t < JB && base + t < R; t++) {
                if (ns > 100 || ne > 100) { S.iv[6] = 1; break; }
                if (S.flag[t] & 1) s_start[ns++] = (uint16_t)(base + t);
                if (S.flag[t] & 2) s_end[ne++] = (uint16_t)(base + t);
            }
            S.iv[4] = ns; S.iv[5] = ne;
        }
        __syncthreads();
        if (S.iv[6]) break;
    }
    int ns = S.iv[4], ne = S.iv[5];
    if (ne <= 0) return;
    blk_colstats(msa, C, s_end, ne, cstat, S.cls);
    double thr = homo_thr(ne, 0.7);
    int valid = 0;
    int he = blk_search_v4(msa, cstat, C, s_end, ne, aend, 1, thr, int_thr_tab(ne), thr, 20, 10, S, &valid);
    if (!valid) return;
    if (ns <= 0) { out.info = HITE_INFO_EXC; return; }
    blk_colstats(msa, C, s_start, ns, cstat, S.cls);
    thr = homo_thr(ns, 0.7);
    int hs = blk_search_v4(msa, cstat, C, s_start, ns, astart, 0, thr, int_thr_tab(ns), thr, 20, 10, S, &valid);
    int nf = blk_select_rows(msa, R, C, hs, he, 1, true, true, S.sel, S);
    if (nf <= 0) return;
    out.row_num = nf;
    blk_colstats(msa, C, S.sel, nf, cstat, S.cls);
    uint8_t *mbody = model + 1;  // one slot in front for the 1-bp left extension
    int ml = blk_consensus(cstat, nf, hs, he, 1, mbody, S);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint8_t *mstart = mbody;
        int c = hs - 1, ext = 0;
        while (ext < 1 && c >= 0) {
            int bc, bk;
            (void)cons_col(cstat + (size_t)c * CS, nf, 1, &bc, &bk);
            if (bc >= nf / 2 && bk != 5 && bk >= 0) { mstart = mbody - 1; *mstart = class_sym(bk); ml++; ext++; }
            c--;
        }
        c = he + 1; ext = 0;
        while (ext < 1 && c < C) {
            int bc, bk;
            (void)cons_col(cstat + (size_t)c * CS, nf, 1, &bc, &bk);
            if (bc >= nf / 2 && bk != 5 && bk >= 0) { mstart[ml++] = class_sym(bk); ext++; }
            c++;
        }
        const char *motifs[4] = {"CTAGT", "CTAAT", "CTGGT", "CTGAT"};
        const int sl = 10, ext_len = 1;
        int l1, h1, l2, h2;
        py_slice(0, sl, ml, &l1, &h1);
        py_slice(-sl, ml, ml, &l2, &h2);
        S.iv[7] = 0;
        for (int t = 0; t < 4; t++) {
            int ei = find_sub(mstart + l2, h2 - l2, motifs[t], 5, true);
            if (ei != -1) {
                int si = find_sub(mstart + l1, h1 - l1, "ATC", 3, false);
                if (si != -1) {
                    int cut = sl - (ei + 3) - 1;
                    int a, b;
                    if (cut == 0) py_slice(si + 1, ml, ml, &a, &b);
                    else py_slice(si + 1, -(int64_t)cut, ml, &a, &b);
                    S.iv[7] = 1; S.iv[8] = (int)(mstart - model) + a; S.iv[9] = b - a;
                    S.iv[10] = hs - ext_len + si + 1; S.iv[11] = he + ext_len - cut;
                    break;
                }
            }
        }
    }
    __syncthreads();
    if (S.iv[7]) {
        out.cons_off += S.iv[8];
        out.cons_len = S.iv[9];
        out.bstart = S.iv[10]; out.bend = S.iv[11];
        out.is_te = S.iv[9] > 0;
    }
}

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
    int k = tsd_search_v5(bytes + off[i], C, bs[i], be[i], plant, l, r);
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

extern "C" int hite_judge_dev(hite_ctx *ctx, int32_t te_type, int32_t plant, int32_t n, const uint8_t *d_msa,
                              const int64_t *d_msa_off, const int32_t *d_rows, const int32_t *d_cols,
                              const uint8_t *d_cand, const int64_t *d_cand_off, const int64_t *d_col_off,
                              int32_t max_cols, int32_t max_rows, hite_call *d_calls, uint8_t *d_cons, void *stream) {
    if (!ctx || n < 0 || te_type < 0 || te_type > 2) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    if (max_cols <= 0 || max_cols > 65535 || max_rows <= 0) return HITE_EINVAL;
    size_t maxC16 = ((size_t)max_cols + 15) & ~(size_t)15;
    size_t slot = 23 * maxC16 + 16 * (size_t)max_rows + 64;
    slot = (slot + 63) & ~(size_t)63;
    int grid = n < 2048 ? n : 2048;
    // keep the scratch bounded (<= 4 GiB): fewer resident slots for very wide alignments
    while (grid > 64 && (size_t)grid * slot > ((size_t)4 << 30)) grid /= 2;
    void *scr = nullptr;
    int rc = hite_scratch_reserve(ctx, (size_t)grid * slot + 256, &scr);
    if (rc) return rc;
    unsigned int *counter = (unsigned int *)((uint8_t *)scr + (size_t)grid * slot);
    HITE_CHECK(ctx, hipMemsetAsync(counter, 0, 4, (hipStream_t)stream));
    JudgeParams P;
    P.te_type = te_type; P.plant = plant; P.n = n; P.msa = d_msa; P.msa_off = d_msa_off; P.rows = d_rows; P.cols = d_cols;
    P.cand = d_cand; P.cand_off = d_cand_off; P.col_off = d_col_off; P.calls = d_calls; P.cons = d_cons;
    P.scratch = (uint8_t *)scr; P.slot_bytes = slot; P.maxC16 = maxC16; P.counter = counter;
    hipLaunchKernelGGL(judge_kernel, dim3(grid), dim3(JB), 0, (hipStream_t)stream, P);
    HITE_CHECK(ctx, hipGetLastError());
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
