// hite_sort.h -- hand-written stable LSD radix sort of (u64 key, u32 value) pairs, shared by FMEA, the copy finder and the
// seeding stage.  Per pass: per-tile digit histogram (LDS atomics), three-phase scan of the [tile][digit] matrix, stable
// scatter.  Two forms: 8-bit digits with a direct scatter (rank from wave ballots + per-wave digit counts) for small
// inputs; 10-bit digits with the tile ranked and staged in LDS in output order (rs_scatter_staged_kernel) from 4 M elements
// on.  HBM streaming: 8 B (histogram) + 12 B read and 12 B written per element per pass.
#pragma once
#include "hite_common.h"
#include "hite_scan.h"
#include "hite_arena.h"

// ---------------------------------------------------------------------------------------------
// stable LSD radix sort of (u64 key, u32 value).  Two digit widths: 8 bits (tile = 256 threads x 8 items) for small
// inputs, 10 bits (1024 bins; tile = 7168 keys, or 4608 keys + values: RSS_KTILE / RSS_TILE below) for large ones: a 50-bit key
// takes 5 passes instead of 7.
// ---------------------------------------------------------------------------------------------
// the [tile][digit] matrices beside the keys: counts fit 16 bits (a tile holds at most 8192 elements), offsets 32 (every caller sorts
// < 2^32 elements) -- as int32 / int64 they were 2.9 GB per pass of the 713 M anchor sort against 11.4 GB of keys
typedef unsigned short rs_cnt_t;
typedef unsigned rs_off_t;
#define RS_ITEMS 8
#define RS_TILE (256 * RS_ITEMS)
#define RS_WIDE_MIN (1 << 22)   // element count from which the 10-bit form is used

template <int BITS, int ITEMS>
static __global__ void __launch_bounds__(256) rs_hist_kernel(const unsigned long long *__restrict__ keys, int64_t n, int shift,
                                                             int nblocks, rs_cnt_t *__restrict__ hist /* [nblocks][bins] */) {
    constexpr int BINS = 1 << BITS;
    __shared__ int h[BINS];
    for (int b = threadIdx.x; b < BINS; b += 256) h[b] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * (256 * ITEMS);
    // all ITEMS keys of a thread are requested before the first is counted (a load under `i < n` compiles to a branch around it and a
    // wait behind it: ITEMS dependent trips to HBM per thread); the last tile re-reads its last key and does not count it
    unsigned long long kk[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const int64_t i = base + it * 256 + threadIdx.x;
        kk[it] = keys[i < n ? i : n - 1];
    }
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const int64_t i = base + it * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(int)((kk[it] >> shift) & (unsigned long long)(BINS - 1))], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < BINS; b += 256) hist[(int64_t)blockIdx.x * BINS + b] = (rs_cnt_t)h[b];      // tile-major: one coalesced run per tile
}

// ---------------------------------------------------------------------------------------------
// offsets of every (tile, digit): offs[t][d] = elements with a smaller digit + elements of digit d in the tiles before t.
// The histogram is tile-major (a tile's histogram is written and its offsets are read as one contiguous run; the digit-major
// form cost a 64-byte sector per 4-byte counter on both sides: PMC showed 1 GB written per pass for a 100 MB histogram), so
// the scan runs down the columns of the [tile][digit] matrix: column sums of groups of RS_GROUP tiles, a scan of the group
// sums along each column (one wavefront per digit), a scan of the digit totals, and the running sums inside each group.
// ---------------------------------------------------------------------------------------------
#define RS_GROUP 64
template <int BINS>
static __global__ void __launch_bounds__(256) rs_colsum_kernel(const rs_cnt_t *__restrict__ hist, int nblocks, long long *__restrict__ gsum /* [groups][BINS] */) {
    const int d = blockIdx.y * 256 + threadIdx.x, g = blockIdx.x;
    const int t0 = g * RS_GROUP, t1 = t0 + RS_GROUP < nblocks ? t0 + RS_GROUP : nblocks;
    long long acc = 0;
    for (int t = t0; t < t1; t++) acc += hist[(int64_t)t * BINS + d];
    gsum[(int64_t)g * BINS + d] = acc;
}
// one wavefront per digit: exclusive scan of its group sums (in place), digit total out
template <int BINS>
static __global__ void __launch_bounds__(256) rs_colscan_kernel(long long *__restrict__ gsum, int ngroups, long long *__restrict__ dtot) {
    const int lane = threadIdx.x & 63, d = blockIdx.x * 4 + (threadIdx.x >> 6);
    long long carry = 0;
    for (int base = 0; base < ngroups; base += 64) {
        const int g = base + lane;
        const long long v = g < ngroups ? gsum[(int64_t)g * BINS + d] : 0;
        long long x = v;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const long long y = __shfl_up(x, dd, 64); if (lane >= dd) x += y; }
        if (g < ngroups) gsum[(int64_t)g * BINS + d] = carry + x - v;
        carry += __shfl(x, 63, 64);
    }
    if (lane == 0) dtot[d] = carry;
}
// exclusive scan of the digit totals (one block)
template <int BINS>
static __global__ void __launch_bounds__(BINS > 1024 ? 1024 : BINS) rs_digitscan_kernel(long long *__restrict__ dtot) {
    __shared__ long long s_w[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long v = dtot[threadIdx.x];
    long long x = v;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) { const long long y = __shfl_up(x, dd, 64); if (lane >= dd) x += y; }
    if (lane == 63) s_w[w] = x;
    __syncthreads();
    long long pre = 0;
    for (int q = 0; q < w; q++) pre += s_w[q];
    dtot[threadIdx.x] = pre + x - v;
}
template <int BINS>
static __global__ void __launch_bounds__(256) rs_coloffs_kernel(const rs_cnt_t *__restrict__ hist, int nblocks, const long long *__restrict__ gsum,
                                                                const long long *__restrict__ dbase, rs_off_t *__restrict__ offs) {
    const int d = blockIdx.y * 256 + threadIdx.x, g = blockIdx.x;
    const int t0 = g * RS_GROUP, t1 = t0 + RS_GROUP < nblocks ? t0 + RS_GROUP : nblocks;
    long long run = gsum[(int64_t)g * BINS + d] + dbase[d];
    for (int t = t0; t < t1; t++) { offs[(int64_t)t * BINS + d] = (rs_off_t)run; run += hist[(int64_t)t * BINS + d]; }
}

template <int BITS, int ITEMS>
static __global__ void __launch_bounds__(256) rs_scatter_kernel(const unsigned long long *__restrict__ kin, const unsigned *__restrict__ vin,
                                                                unsigned long long *__restrict__ kout, unsigned *__restrict__ vout,
                                                                int64_t n, int shift, int nblocks, const rs_off_t *__restrict__ offs) {
    constexpr int BINS = 1 << BITS;
    __shared__ unsigned base[BINS];   // global positions fit 32 bits (every caller sorts < 2^32 elements)
    __shared__ int cnt[4][BINS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int b = threadIdx.x; b < BINS; b += 256) {
        base[b] = (unsigned)offs[(int64_t)blockIdx.x * BINS + b];
        cnt[0][b] = 0; cnt[1][b] = 0; cnt[2][b] = 0; cnt[3][b] = 0;
    }
    __syncthreads();
    int64_t tile = (int64_t)blockIdx.x * (256 * ITEMS);
    for (int it = 0; it < ITEMS; it++) {
        // invariant: cnt is all zero here.  Only the entries of digits present in this round are touched (and reset).
        int64_t i = tile + it * 256 + threadIdx.x;
        bool act = i < n;
        unsigned long long k = act ? kin[i] : 0;
        unsigned v = (act && vin) ? vin[i] : 0;       // vin == NULL: keys only
        int d = (int)((k >> shift) & (unsigned long long)(BINS - 1));
        // lanes of this wave with the same digit (inactive lanes match nothing)
        unsigned long long peers = __ballot(act);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            unsigned long long bal = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? bal : ~bal;
        }
        const int rank = __popcll(peers & ((1ull << lane) - 1ull));
        const int mine = __popcll(peers);
        if (act && rank == 0) cnt[w][d] = mine;
        __syncthreads();
        if (act) {
            unsigned pos = base[d] + (unsigned)rank;
            for (int q = 0; q < w; q++) pos += (unsigned)cnt[q][d];
            kout[pos] = k;
            if (vin) vout[pos] = v;
        }
        __syncthreads();
        if (act && rank == 0) { atomicAdd(&base[d], (unsigned)mine); cnt[w][d] = 0; }
        __syncthreads();
    }
}

// 10-bit scatter with the tile staged in LDS in output order.  The direct form above writes every element to its final
// position as it is ranked: 256 elements per round land in up to 256 different bins and HBM sees partial-line writes
// (PMC: 3x the algorithmic write bytes).  Here a block of 8 wavefronts owns a tile of 512 R elements (wave w: the contiguous
// chunk [64 R w, 64 R (w + 1)), R coalesced rounds, keys / values held in registers), ranks them stably inside the tile
// (per-wave digit counts -> offsets; ballots for the rank inside a round), writes them to LDS in digit order and streams
// the runs out.
// LDS: keys 8 B x tile | values 4 B x tile | per-wave digit offsets u16 [8][1024] 16 KB | run shift u32 [1024] 4 KB | scan 64 B.
// Round 6, two things that only pay TOGETHER (the 713 M anchor sort of stage 3.1: 21.0 ms as it was, 21.0 with the first alone,
// 20.6 with the second alone, 14.7 with both; R = 8 / 10 keys only: 15.7 / 15.9):
//  * two workgroups per compute unit -- keys only R = 14 (76 KB, no value area; runs of 7 keys), keys + values R = 9 (74.5 KB,
//    runs of 4.5; 16 -> 9 took the index sort from 7.9 to 6.7 ms): one loads or streams out while the other ranks;
//  * tiles that follow each other in the input on ONE chiplet (below): a digit's output line is finished by the next few tiles,
//    and with those on other chiplets every L2 sent its part of the line to memory on its own.
#ifndef RSS_VROUNDS
#define RSS_VROUNDS 9
#endif
#define RSS_TILE (512 * RSS_VROUNDS)
#define RSS_LDS_BYTES (RSS_TILE * 12 + 8 * 1024 * 2 + 1024 * 4 + 64)
#ifndef RSS_KROUNDS
#define RSS_KROUNDS 14
#endif
#define RSS_KTILE (512 * RSS_KROUNDS)
#ifndef RSS_XCD_MAP
#define RSS_XCD_MAP 1
#endif
#define RSS_KLDS_BYTES (RSS_KTILE * 8 + 8 * 1024 * 2 + 1024 * 4 + 64)
template <bool VALS>
static __global__ void __launch_bounds__(512) rs_scatter_staged_kernel(const unsigned long long *__restrict__ kin, const unsigned *__restrict__ vin,
                                                                       unsigned long long *__restrict__ kout, unsigned *__restrict__ vout,
                                                                       int64_t n, int shift, int nblocks, const rs_off_t *__restrict__ offs) {
    extern __shared__ unsigned char rss_lds[];
    constexpr int R = VALS ? RSS_VROUNDS : RSS_KROUNDS, TILE = 512 * R, ESZ = VALS ? 12 : 8;
    unsigned long long *skey = reinterpret_cast<unsigned long long *>(rss_lds);
    unsigned *sval = reinterpret_cast<unsigned *>(rss_lds + TILE * 8);                            // (VALS only)
    unsigned short *woff = reinterpret_cast<unsigned short *>(rss_lds + TILE * ESZ);              // [8][1024]
    unsigned *delta = reinterpret_cast<unsigned *>(rss_lds + TILE * ESZ + 8 * 1024 * 2);          // [1024]
    int *wsum = reinterpret_cast<int *>(rss_lds + TILE * ESZ + 8 * 1024 * 2 + 1024 * 4);         // [8] + total
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // tile of this workgroup: the tiles that follow each other in the input finish each other's output lines (a digit's run grows by ~8
    // keys = 64 B per tile), so they go to ONE chiplet and its L2 -- workgroup b runs on chiplet b % 8 (observed; only speed depends on
    // it): chiplet x takes the x-th eighth of the tiles in order
    int tb;
    {
        const int x = (int)(blockIdx.x & 7u), i = (int)(blockIdx.x >> 3), q = nblocks >> 3, rr = nblocks & 7;
        tb = RSS_XCD_MAP ? x * q + (x < rr ? x : rr) + i : (int)blockIdx.x;
    }
    const int64_t tile = (int64_t)tb * TILE;
    unsigned long long k[R];
    unsigned v[R];
    unsigned info[R];   // rank inside the round (low 8 bits) | lanes with the same digit (next 8 bits) | digit << 16
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int64_t i = tile + w * (64 * R) + r * 64 + lane;
        const bool act = i < n;
        k[r] = act ? kin[i] : 0ull;
        v[r] = (VALS && act) ? vin[i] : 0u;
    }
    for (int b = threadIdx.x; b < 8 * 1024; b += 512) woff[b] = 0;
    __syncthreads();
    unsigned short *myoff = woff + w * 1024;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const bool act = tile + w * (64 * R) + r * 64 + lane < n;
        const int d = (int)((k[r] >> shift) & 1023ull);
        unsigned long long peers = __ballot(act);
#pragma unroll
        for (int b = 0; b < 10; b++) {
            const unsigned long long bal = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? bal : ~bal;
        }
        const int rank = __popcll(peers & ((1ull << lane) - 1ull));
        const int mine = __popcll(peers);
        info[r] = act ? ((unsigned)rank | ((unsigned)(mine - 1) << 8) | ((unsigned)d << 16) | 0x80000000u) : 0u;
        if (act && rank == 0) myoff[d] = (unsigned short)(myoff[d] + mine);      // wave-private row, one writer per digit and round
    }
    __syncthreads();
    // digit totals -> exclusive scan over the 1024 digits -> per-wave start offsets; delta = global run start - local run start
    {
        const int d0 = threadIdx.x * 2;
        int c0 = 0, c1 = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) { c0 += woff[q * 1024 + d0]; c1 += woff[q * 1024 + d0 + 1]; }
        int incl = c0 + c1;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const int t = __shfl_up(incl, dd); if (lane >= dd) incl += t; }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int pre = incl - (c0 + c1);
        for (int q = 0; q < w; q++) pre += wsum[q];
        // pre = local start of digit d0, pre + c0 = local start of digit d0 + 1
        int s0 = pre, s1 = pre + c0;
        delta[d0] = (unsigned)offs[(int64_t)tb * 1024 + d0] - (unsigned)s0;
        delta[d0 + 1] = (unsigned)offs[(int64_t)tb * 1024 + d0 + 1] - (unsigned)s1;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int a0 = woff[q * 1024 + d0], a1 = woff[q * 1024 + d0 + 1];
            woff[q * 1024 + d0] = (unsigned short)s0; woff[q * 1024 + d0 + 1] = (unsigned short)s1;
            s0 += a0; s1 += a1;
        }
        if (threadIdx.x == 511) wsum[8] = pre + c0 + c1;   // elements in the tile
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; r++) {
        const unsigned inf = info[r];
        const bool act = inf >> 31;
        const int d = (int)((inf >> 16) & 1023u), rank = (int)(inf & 255u), mine = (int)((inf >> 8) & 255u) + 1;
        int pos = 0;
        if (act) pos = myoff[d] + rank;
        __builtin_amdgcn_wave_barrier();
        if (act) { skey[pos] = k[r]; if (VALS) sval[pos] = v[r]; if (rank == 0) myoff[d] = (unsigned short)(myoff[d] + mine); }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    const int total = wsum[8];
    for (int j = threadIdx.x; j < total; j += 512) {
        const unsigned long long key = skey[j];
        const unsigned pos = (unsigned)j + delta[(int)((key >> shift) & 1023ull)];
        kout[pos] = key;
        if (VALS) vout[pos] = sval[j];
    }
}

struct Sorter {
    hite_ctx *ctx;
    hipStream_t st;
    int64_t cap = 0;
    unsigned long long *k2 = nullptr;
    unsigned *v2 = nullptr;
    rs_cnt_t *hist = nullptr;
    rs_off_t *offs = nullptr;
    int64_t *bs = nullptr;
    int64_t hist_n = 0;
    bool owned = true;      // false: the buffers belong to an arena (sorter_free leaves them alone)
};
// a routine that wants every temporary of its sorts and buffers from ONE grow-only arena sets this for its duration (hite_fmea.hip):
// sorter_init then allocates there instead of calling hipMalloc
static thread_local Arena *tl_sort_arena = nullptr;
// histogram entries a sort of n elements needs (the larger of the two forms)
// int64 elements of scan scratch (Sorter::bs) for a histogram of hist_n counters: the group sums + the digit totals
static inline int64_t sorter_tmp_elems(int64_t hist_n) { return hist_n / RS_GROUP + 3 * 1024 + 64; }
static inline int64_t sorter_hist_elems(int64_t n) {
    int64_t nb8 = (n + RS_TILE - 1) / RS_TILE; if (nb8 < 1) nb8 = 1;
    const int64_t t10 = RSS_KTILE < RSS_TILE ? RSS_KTILE : RSS_TILE;               // (the smaller of the two 10-bit tiles)
    int64_t nb10 = (n + t10 - 1) / t10; if (nb10 < 1) nb10 = 1;
    int64_t a = 256 * nb8, b = 1024 * nb10;
    return a > b ? a : b;
}

static int sorter_init(Sorter &S, hite_ctx *ctx, hipStream_t st, int64_t n) {
    S.ctx = ctx; S.st = st; S.cap = n;
    HITE_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(rs_scatter_staged_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        RSS_LDS_BYTES));
    HITE_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(rs_scatter_staged_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        RSS_KLDS_BYTES));
    S.hist_n = sorter_hist_elems(n);
    if (tl_sort_arena) {
        void *p;
        S.owned = false;
        int rc;
        if ((rc = arena_alloc(ctx, *tl_sort_arena, (size_t)(n + 1) * 8, &p))) return rc; S.k2 = (unsigned long long *)p;
        if ((rc = arena_alloc(ctx, *tl_sort_arena, (size_t)(n + 1) * 4, &p))) return rc; S.v2 = (unsigned *)p;
        if ((rc = arena_alloc(ctx, *tl_sort_arena, (size_t)S.hist_n * sizeof(rs_cnt_t), &p))) return rc; S.hist = (rs_cnt_t *)p;
        if ((rc = arena_alloc(ctx, *tl_sort_arena, (size_t)(S.hist_n + 1) * sizeof(rs_off_t), &p))) return rc; S.offs = (rs_off_t *)p;
        if ((rc = arena_alloc(ctx, *tl_sort_arena, (size_t)sorter_tmp_elems(S.hist_n) * 8, &p))) return rc; S.bs = (int64_t *)p;
        return HITE_OK;
    }
    HITE_CHECK(ctx, hipMalloc((void **)&S.k2, (size_t)(n + 1) * 8));
    HITE_CHECK(ctx, hipMalloc((void **)&S.v2, (size_t)(n + 1) * 4));
    HITE_CHECK(ctx, hipMalloc((void **)&S.hist, (size_t)S.hist_n * sizeof(rs_cnt_t)));
    HITE_CHECK(ctx, hipMalloc((void **)&S.offs, (size_t)(S.hist_n + 1) * sizeof(rs_off_t)));
    HITE_CHECK(ctx, hipMalloc((void **)&S.bs, (size_t)sorter_tmp_elems(S.hist_n) * 8));
    return HITE_OK;
}
static void sorter_free(Sorter &S) {
    if (!S.owned) return;
    if (S.k2) (void)hipFree(S.k2);
    if (S.v2) (void)hipFree(S.v2);
    if (S.hist) (void)hipFree(S.hist);
    if (S.offs) (void)hipFree(S.offs);
    if (S.bs) (void)hipFree(S.bs);
}
// sorts (keys, vals) in place (ping-pong through the sorter's buffers) on the key bits [lo_bit, hi_bit), stable
static int sorter_sort_bits_impl(Sorter &S, unsigned long long **keys_io, unsigned **vals_io, int64_t n, int lo_bit, int hi_bit, bool swap) {
    unsigned long long *keys = *keys_io;
    unsigned *vals = *vals_io;
    if (n <= 1) return HITE_OK;
    if (n >= 0xffffffffll) return HITE_EINVAL;   // 32-bit positions inside the scatter kernel
    // HITE_SORT_WIDE_MIN (tests): element count from which the 10-bit staged form is used, default RS_WIDE_MIN
    static const long long wide_min = [] { const char *e = getenv("HITE_SORT_WIDE_MIN"); return e && *e ? atoll(e) : (long long)RS_WIDE_MIN; }();
    const bool wide = n >= wide_min;
    const int bits = wide ? 10 : 8;
    const int tile = wide ? (vals ? RSS_TILE : RSS_KTILE) : RS_TILE;
    int nblocks = (int)((n + tile - 1) / tile);
    unsigned long long *ka = keys, *kb = S.k2;
    unsigned *va = vals, *vb = vals ? S.v2 : nullptr;      // vals == NULL: keys only (8 instead of 12 bytes moved per element and pass)
    int passes = (hi_bit - lo_bit + bits - 1) / bits;
    for (int p = 0; p < passes; p++) {
        const int sh = lo_bit + p * bits;
        if (wide && vals) hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_hist_kernel<10, RSS_TILE / 256>), dim3(nblocks), dim3(256), 0, S.st, ka, n, sh, nblocks, S.hist);
        else if (wide) hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_hist_kernel<10, RSS_KTILE / 256>), dim3(nblocks), dim3(256), 0, S.st, ka, n, sh, nblocks, S.hist);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_hist_kernel<8, 8>), dim3(nblocks), dim3(256), 0, S.st, ka, n, sh, nblocks, S.hist);
        {
            const int ngroups = (nblocks + RS_GROUP - 1) / RS_GROUP;
            long long *gsum = reinterpret_cast<long long *>(S.bs), *dtot = gsum + (int64_t)ngroups * (1 << bits);
            if (wide) {
                hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_colsum_kernel<1024>), dim3(ngroups, 4), dim3(256), 0, S.st, S.hist, nblocks, gsum);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_colscan_kernel<1024>), dim3(256), dim3(256), 0, S.st, gsum, ngroups, dtot);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_digitscan_kernel<1024>), dim3(1), dim3(1024), 0, S.st, dtot);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_coloffs_kernel<1024>), dim3(ngroups, 4), dim3(256), 0, S.st, S.hist, nblocks, gsum, dtot, S.offs);
            } else {
                hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_colsum_kernel<256>), dim3(ngroups, 1), dim3(256), 0, S.st, S.hist, nblocks, gsum);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_colscan_kernel<256>), dim3(64), dim3(256), 0, S.st, gsum, ngroups, dtot);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_digitscan_kernel<256>), dim3(1), dim3(256), 0, S.st, dtot);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_coloffs_kernel<256>), dim3(ngroups, 1), dim3(256), 0, S.st, S.hist, nblocks, gsum, dtot, S.offs);
            }
        }
        if (wide && vals) hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_scatter_staged_kernel<true>), dim3(nblocks), dim3(512), RSS_LDS_BYTES, S.st, ka, va, kb, vb, n, sh, nblocks, S.offs);
        else if (wide) hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_scatter_staged_kernel<false>), dim3(nblocks), dim3(512), RSS_KLDS_BYTES, S.st, ka, va, kb, vb, n, sh, nblocks, S.offs);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(rs_scatter_kernel<8, 8>), dim3(nblocks), dim3(256), 0, S.st, ka, va, kb, vb, n, sh, nblocks, S.offs);
        unsigned long long *tk = ka; ka = kb; kb = tk;
        unsigned *tv = va; va = vb; vb = tv;
    }
    if (ka != keys) {
        if (swap) { *keys_io = ka; S.k2 = keys; if (vals) { *vals_io = va; S.v2 = vals; } }   // the caller goes on with the other buffers
        else {
            HITE_CHECK(S.ctx, hipMemcpyAsync(keys, ka, (size_t)n * 8, hipMemcpyDeviceToDevice, S.st));
            if (vals) HITE_CHECK(S.ctx, hipMemcpyAsync(vals, va, (size_t)n * 4, hipMemcpyDeviceToDevice, S.st));
        }
    }
    HITE_CHECK(S.ctx, hipGetLastError());
    return HITE_OK;
}
static int sorter_sort_bits(Sorter &S, unsigned long long *keys, unsigned *vals, int64_t n, int lo_bit, int hi_bit) {
    return sorter_sort_bits_impl(S, &keys, &vals, n, lo_bit, hi_bit, false);
}
// the same without the copy back after an odd number of passes: *keys / *vals name the sorted buffers on return (either the
// caller's or the sorter's second pair: both stay valid)
static int sorter_sort_bits_swap(Sorter &S, unsigned long long **keys, unsigned **vals, int64_t n, int lo_bit, int hi_bit) {
    return sorter_sort_bits_impl(S, keys, vals, n, lo_bit, hi_bit, true);
}
// bits [0, nbits)
static int sorter_sort(Sorter &S, unsigned long long *keys, unsigned *vals, int64_t n, int nbits) {
    return sorter_sort_bits(S, keys, vals, n, 0, nbits);
}
