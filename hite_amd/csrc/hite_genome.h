// hite_genome.h -- device helpers over the resident 2-bit genome (shared by hite_ctx.hip and
// hite_pipeline.hip).
#pragma once
#include "hite_common.h"

// window length rules  Util.py:8102-8109, 8117
__device__ __forceinline__ void window_rule(const int64_t *__restrict__ coff, int32_t ncontig, int32_t c, int64_t s1,
                                            int64_t e1, int32_t flank, int64_t &len, int64_t &tlen, int64_t &g_lo) {
    len = 0; tlen = 0; g_lo = 0;
    if (c < 0 || c >= ncontig) return;
    int64_t clen = coff[c + 1] - coff[c];
    if (s1 - 1 - flank < 0 || e1 + flank > clen) return;
    int64_t lo = s1 - 1 - flank, hi = e1 + flank;
    if (hi < lo) hi = lo;
    int64_t n = hi - lo;
    if (n < 100) return;
    len = n;
    tlen = n > 1000 ? 1000 : 0;
    g_lo = coff[c] + lo;
}

// 4 consecutive bases starting at packed index g -> 4 ASCII bytes (little endian in a u32)
__device__ __forceinline__ uint32_t fetch4(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask,
                                           int64_t g) {
    int64_t w = g >> 4;
    int sh = (int)(g & 15) * 2;
    uint64_t two = (uint64_t)bases[w] | ((uint64_t)bases[w + 1] << 32);
    uint32_t bits = (uint32_t)(two >> sh) & 0xffu;
    int64_t mw = g >> 5;
    int msh = (int)(g & 31);
    uint64_t mtwo = (uint64_t)nmask[mw] | ((uint64_t)nmask[mw + 1] << 32);
    uint32_t m = (uint32_t)(mtwo >> msh) & 0xfu;
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t code = (bits >> (2 * i)) & 3u;
        uint32_t ch = (0x54474341u >> (8 * code)) & 0xffu;  // "ACGT"
        if ((m >> i) & 1u) ch = 'N';
        out |= ch << (8 * i);
    }
    return out;
}
// reverse-complement of 4 ASCII bytes packed in a u32 (byte order reversed, bases complemented)
__device__ __forceinline__ uint32_t revcomp4(uint32_t x) {
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t c = (x >> (8 * i)) & 0xffu;
        uint32_t r = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'N';
        out |= r << (8 * (3 - i));
    }
    return out;
}

// write window[ws .. ws+cnt) to dst (4-byte aligned) using the lanes of one wave
__device__ __forceinline__ void emit_span(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask,
                                          int64_t g_lo, int64_t wlen, bool minus, int64_t ws, int64_t cnt,
                                          uint8_t *__restrict__ dst, int lane) {
    int64_t groups = (cnt + 3) >> 2;
    bool aligned = (((uintptr_t)dst) & 3) == 0;
    for (int64_t j = lane; j < groups; j += 64) {
        int64_t p = ws + 4 * j;  // window position of the first byte of this group
        int rem = (int)((cnt - 4 * j) < 4 ? (cnt - 4 * j) : 4);
        uint32_t v;
        if (!minus) {
            v = fetch4(bases, nmask, g_lo + p);
        } else {
            // window[p+i] = comp(genome[g_lo + wlen-1-p-i]); fetch ascending from g_lo + wlen - 4 - p
            int64_t g = g_lo + wlen - 4 - p;
            if (g >= 0) v = revcomp4(fetch4(bases, nmask, g));
            else {  // only possible in a partial tail group at the very start of the genome
                v = 0;
                for (int i = 0; i < rem; i++) {
                    uint32_t c = fetch4(bases, nmask, g_lo + wlen - 1 - p - i) & 0xffu;
                    uint32_t r = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'N';
                    v |= r << (8 * i);
                }
            }
        }
        if (rem == 4 && aligned) *reinterpret_cast<uint32_t *>(dst + 4 * j) = v;
        else for (int i = 0; i < rem; i++) dst[4 * j + i] = (uint8_t)(v >> (8 * i));
    }
}


static __global__ void flank_sizes_kernel(const int64_t *__restrict__ coff, int32_t ncontig, int64_t n,
                                   const int32_t *__restrict__ contig, const int64_t *__restrict__ s1,
                                   const int64_t *__restrict__ e1, int32_t flank, int64_t *__restrict__ out_len,
                                   int64_t *__restrict__ trunc_len) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t len, tlen, g;
    window_rule(coff, ncontig, contig[i], s1[i], e1[i], flank, len, tlen, g);
    out_len[i] = len;
    if (trunc_len) trunc_len[i] = tlen;
}

