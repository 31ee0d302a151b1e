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

// >>> genome_decode (tests/test_host_compiled.py compiles this block for the host and compares it with plain slicing)
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

// 16 consecutive bases starting at packed index g: their 2-bit codes (base i in bits 2i, 2i + 1) and their "not ACGT" bits
__device__ __forceinline__ void fetch16_codes(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask, int64_t g,
                                              uint32_t &codes, uint32_t &nbits) {
    const int64_t w = g >> 4;
    const uint64_t two = (uint64_t)bases[w] | ((uint64_t)bases[w + 1] << 32);
    codes = (uint32_t)(two >> ((int)(g & 15) * 2));
    const int64_t mw = g >> 5;
    const uint64_t mtwo = (uint64_t)nmask[mw] | ((uint64_t)nmask[mw + 1] << 32);
    nbits = (uint32_t)(mtwo >> (int)(g & 31)) & 0xffffu;
}
// the same 16 bases read backwards and complemented
__device__ __forceinline__ void revcomp16_codes(uint32_t &codes, uint32_t &nbits) {
    const uint32_t r = __builtin_bitreverse32(~codes);                       // groups reversed, the two bits of a group swapped
    codes = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
    nbits = __builtin_bitreverse32(nbits) >> 16;
}
// four of them (8 code bits, 4 "not ACGT" bits) as ASCII
__device__ __forceinline__ uint32_t ascii4(uint32_t c8, uint32_t n4) {
    const uint32_t sel = (c8 | (c8 << 6) | (c8 << 12) | (c8 << 18)) & 0x03030303u;
    uint32_t v = __builtin_amdgcn_perm(0u, 0x54474341u, sel);               // "ACGT"[code] per byte
    if (n4) {
        const uint32_t full = (((n4 | (n4 << 7) | (n4 << 14) | (n4 << 21)) & 0x01010101u)) * 0xffu;
        v = (v & ~full) | (0x4e4e4e4eu & full);
    }
    return v;
}

// write window[ws .. ws+cnt) to dst using the lanes of one wave, four bases per lane and turn (any alignment of dst)
__device__ __forceinline__ void emit_span4(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask,
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
// the same, 16 bases per lane and turn wherever dst allows 16-byte stores (a window is 1-30 kB: with 4-byte stores a wavefront
// wrote 256 B per turn and spent ~40 instructions and 4 loads on every 4 bases); head and tail of the span go four at a time
__device__ __forceinline__ void emit_span(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask,
                                          int64_t g_lo, int64_t wlen, bool minus, int64_t ws, int64_t cnt,
                                          uint8_t *__restrict__ dst, int lane) {
    const int lead = (int)((0 - (uintptr_t)dst) & 15);
    if ((((uintptr_t)dst) & 3) != 0 || cnt < lead + 16) { emit_span4(bases, nmask, g_lo, wlen, minus, ws, cnt, dst, lane); return; }
    if (lead) emit_span4(bases, nmask, g_lo, wlen, minus, ws, lead, dst, lane);
    const int64_t n16 = (cnt - lead) >> 4;
    uint8_t *d16 = dst + lead;
    for (int64_t j = lane; j < n16; j += 64) {
        const int64_t p = ws + lead + 16 * j;
        uint32_t codes, nb;
        if (!minus) fetch16_codes(bases, nmask, g_lo + p, codes, nb);
        else { fetch16_codes(bases, nmask, g_lo + wlen - 16 - p, codes, nb); revcomp16_codes(codes, nb); }   // >= g_lo: p + 16 <= wlen
        uint4 v;
        v.x = ascii4(codes & 0xffu, nb & 0xfu);
        v.y = ascii4((codes >> 8) & 0xffu, (nb >> 4) & 0xfu);
        v.z = ascii4((codes >> 16) & 0xffu, (nb >> 8) & 0xfu);
        v.w = ascii4(codes >> 24, nb >> 12);
        *reinterpret_cast<uint4 *>(d16 + 16 * j) = v;
    }
    const int64_t done = lead + 16 * n16;
    if (done < cnt) emit_span4(bases, nmask, g_lo, wlen, minus, ws + done, cnt - done, dst + done, lane);
}


// <<< genome_decode

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

