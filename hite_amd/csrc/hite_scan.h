// hite_scan.h -- three-phase exclusive scan (int64 out, out[n] = total), shared by the pipeline and FMEA.
#pragma once
#include "hite_common.h"

// ---------------------------------------------------------------------------------------------
// exclusive scan (int64 out, out[n] = total), three phases; TIn = int32_t or int64_t
// ---------------------------------------------------------------------------------------------
#define SCAN_TILE 4096
template <typename TIn>
static __global__ void __launch_bounds__(256) scan_sums_kernel(const TIn *__restrict__ in, int64_t n, int64_t *__restrict__ bsum) {
    __shared__ long long s_w[4];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    long long acc = 0;
    for (int i = threadIdx.x; i < SCAN_TILE; i += 256) {
        int64_t k = base + i;
        if (k < n) acc += (long long)in[k];
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
static __global__ void __launch_bounds__(1024) scan_bsums_kernel(int64_t *__restrict__ bsum, int64_t nb, int64_t *__restrict__ total) {
    // one block of 16 wavefronts walks the tile sums 1024 at a time: wave scan (shuffles), 16 wave totals through LDS, running carry
    __shared__ long long s_w[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    long long carry = 0;
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t k = base + threadIdx.x;
        const long long v = k < nb ? bsum[k] : 0;
        long long x = v;
        for (int d = 1; d < 64; d <<= 1) { long long y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        __syncthreads();
        if (lane == 63) s_w[w] = x;
        __syncthreads();
        long long pre = 0, tot = 0;
        for (int q = 0; q < 16; q++) { const long long t = s_w[q]; if (q < w) pre += t; tot += t; }
        if (k < nb) bsum[k] = carry + pre + x - v;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}
template <typename TIn>
static __global__ void __launch_bounds__(256) scan_apply_kernel(const TIn *__restrict__ in, int64_t n, const int64_t *__restrict__ bsum,
                                                         int64_t *__restrict__ out) {
    __shared__ long long s_w[4];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    long long running = bsum[blockIdx.x];
    for (int i0 = 0; i0 < SCAN_TILE; i0 += 256) {
        int64_t k = base + i0 + threadIdx.x;
        long long v = k < n ? (long long)in[k] : 0;
        long long x = v;
        int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (int d = 1; d < 64; d <<= 1) { long long y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        __syncthreads();
        if (lane == 63) s_w[w] = x;
        __syncthreads();
        long long pre = 0, tot = 0;
        for (int q = 0; q < 4; q++) { if (q < w) pre += s_w[q]; tot += s_w[q]; }
        if (k < n) out[k] = running + pre + x - v;
        running += tot;
    }
}
static inline int64_t scan_tmp_elems(int64_t n) { int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE; return nb < 1 ? 1 : nb; }
// d_bs: scan_tmp_elems(n) int64 of scratch
template <typename TIn>
static int scan_excl_buf(hite_ctx *ctx, int64_t *d_bs, const TIn *d_in, int64_t n, int64_t *d_out /* n+1 */, hipStream_t st) {
    int64_t nb = scan_tmp_elems(n);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(scan_sums_kernel<TIn>), dim3((unsigned)nb), dim3(256), 0, st, d_in, n, d_bs);
    hipLaunchKernelGGL(scan_bsums_kernel, dim3(1), dim3(1024), 0, st, d_bs, nb, d_out + n);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(scan_apply_kernel<TIn>), dim3((unsigned)nb), dim3(256), 0, st, d_in, n, d_bs, d_out);
    HITE_CHECK(ctx, hipGetLastError());
    return HITE_OK;
}
