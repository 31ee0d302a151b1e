// hite_sort.h -- hand-written stable LSD radix sort of (u64 key, u32 value) pairs, shared by FMEA and the
// copy finder.  8 bits per pass; per pass: per-tile digit histogram (LDS atomics), three-phase scan of the
// [digit][tile] matrix, stable scatter whose in-tile rank comes from wave ballots (lanes with the same digit)
// plus per-wave digit counts in LDS.  HBM streaming: 12 B read + 12 B written per element per pass.
#pragma once
#include "hite_common.h"
#include "hite_scan.h"

// ---------------------------------------------------------------------------------------------
// stable LSD radix sort of (u64 key, u32 value), 8 bits per pass, tile = 256 threads x 8 items
// ---------------------------------------------------------------------------------------------
#define RS_ITEMS 8
#define RS_TILE (256 * RS_ITEMS)

static __global__ void __launch_bounds__(256) rs_hist_kernel(const unsigned long long *__restrict__ keys, int64_t n, int shift,
                                                      int nblocks, int32_t *__restrict__ hist /* [256][nblocks] */) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * RS_TILE;
    for (int it = 0; it < RS_ITEMS; it++) {
        int64_t i = base + it * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(int)((keys[i] >> shift) & 255ull)], 1);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

static __global__ void __launch_bounds__(256) rs_scatter_kernel(const unsigned long long *__restrict__ kin, const unsigned *__restrict__ vin,
                                                         unsigned long long *__restrict__ kout, unsigned *__restrict__ vout,
                                                         int64_t n, int shift, int nblocks, const int64_t *__restrict__ offs) {
    __shared__ long long base[256];
    __shared__ int cnt[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    base[threadIdx.x] = offs[(int64_t)threadIdx.x * nblocks + blockIdx.x];
    int64_t tile = (int64_t)blockIdx.x * RS_TILE;
    for (int it = 0; it < RS_ITEMS; it++) {
        for (int q = 0; q < 4; q++) cnt[q][threadIdx.x] = 0;
        __syncthreads();
        int64_t i = tile + it * 256 + threadIdx.x;
        bool act = i < n;
        unsigned long long k = act ? kin[i] : 0;
        unsigned v = act ? vin[i] : 0;
        int d = (int)((k >> shift) & 255ull);
        // lanes of this wave with the same digit (inactive lanes match nothing)
        unsigned long long peers = __ballot(act);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            unsigned long long bal = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? bal : ~bal;
        }
        int rank = __popcll(peers & ((1ull << lane) - 1ull));
        if (act && rank == 0) cnt[w][d] = __popcll(peers);
        __syncthreads();
        if (act) {
            long long pos = base[d];
            for (int q = 0; q < w; q++) pos += cnt[q][d];
            pos += rank;
            kout[pos] = k; vout[pos] = v;
        }
        __syncthreads();
        base[threadIdx.x] += cnt[0][threadIdx.x] + cnt[1][threadIdx.x] + cnt[2][threadIdx.x] + cnt[3][threadIdx.x];
        __syncthreads();
    }
}

struct Sorter {
    hite_ctx *ctx;
    hipStream_t st;
    int64_t cap = 0;
    unsigned long long *k2 = nullptr;
    unsigned *v2 = nullptr;
    int32_t *hist = nullptr;
    int64_t *offs = nullptr, *bs = nullptr;
    int64_t hist_n = 0;
};

static int sorter_init(Sorter &S, hite_ctx *ctx, hipStream_t st, int64_t n) {
    S.ctx = ctx; S.st = st; S.cap = n;
    int64_t nblocks = (n + RS_TILE - 1) / RS_TILE; if (nblocks < 1) nblocks = 1;
    S.hist_n = 256 * nblocks;
    HITE_CHECK(ctx, hipMalloc((void **)&S.k2, (size_t)(n + 1) * 8));
    HITE_CHECK(ctx, hipMalloc((void **)&S.v2, (size_t)(n + 1) * 4));
    HITE_CHECK(ctx, hipMalloc((void **)&S.hist, (size_t)S.hist_n * 4));
    HITE_CHECK(ctx, hipMalloc((void **)&S.offs, (size_t)(S.hist_n + 1) * 8));
    HITE_CHECK(ctx, hipMalloc((void **)&S.bs, (size_t)scan_tmp_elems(S.hist_n) * 8));
    return HITE_OK;
}
static void sorter_free(Sorter &S) {
    if (S.k2) (void)hipFree(S.k2);
    if (S.v2) (void)hipFree(S.v2);
    if (S.hist) (void)hipFree(S.hist);
    if (S.offs) (void)hipFree(S.offs);
    if (S.bs) (void)hipFree(S.bs);
}
// sorts (keys, vals) in place (ping-pong through the sorter's buffers) on the key bits [lo_bit, hi_bit), stable
static int sorter_sort_bits(Sorter &S, unsigned long long *keys, unsigned *vals, int64_t n, int lo_bit, int hi_bit) {
    if (n <= 1) return HITE_OK;
    int nblocks = (int)((n + RS_TILE - 1) / RS_TILE);
    unsigned long long *ka = keys, *kb = S.k2;
    unsigned *va = vals, *vb = S.v2;
    int passes = (hi_bit - lo_bit + 7) / 8;
    for (int p = 0; p < passes; p++) {
        const int sh = lo_bit + p * 8;
        hipLaunchKernelGGL(rs_hist_kernel, dim3(nblocks), dim3(256), 0, S.st, ka, n, sh, nblocks, S.hist);
        int rc = scan_excl_buf<int32_t>(S.ctx, S.bs, S.hist, (int64_t)256 * nblocks, S.offs, S.st);
        if (rc) return rc;
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(nblocks), dim3(256), 0, S.st, ka, va, kb, vb, n, sh, nblocks, S.offs);
        unsigned long long *tk = ka; ka = kb; kb = tk;
        unsigned *tv = va; va = vb; vb = tv;
    }
    if (ka != keys) {
        HITE_CHECK(S.ctx, hipMemcpyAsync(keys, ka, (size_t)n * 8, hipMemcpyDeviceToDevice, S.st));
        HITE_CHECK(S.ctx, hipMemcpyAsync(vals, va, (size_t)n * 4, hipMemcpyDeviceToDevice, S.st));
    }
    HITE_CHECK(S.ctx, hipGetLastError());
    return HITE_OK;
}
// bits [0, nbits)
static int sorter_sort(Sorter &S, unsigned long long *keys, unsigned *vals, int64_t n, int nbits) {
    return sorter_sort_bits(S, keys, vals, n, 0, nbits);
}
