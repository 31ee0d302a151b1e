// hite_trmask.hip -- tandem-repeat masking of the resident genome: this build's GPU stage where the reference shells out to
// `trf <file> 2 7 7 80 10 50 500 -f -d -m -h` and continues with the .mask FASTA (run_remove_TR,
// /root/reference/module/Util.py:2855-2874; filter_tandem_repeats :4672-4697).  TRF is third-party: PARITY UNPINNED; the
// definition is the header of the twin, oracle/hite_oracle_trf.c (HIP == twin bit for bit; both measured against TRF 4.09's
// own masks, tests/test_trmask.py).
//
//   seed filter: a workgroup owns 4096 bases (256 words of 16 two-bit bases, staged in LDS with the "not A/C/G/T" bits spread
//     to the same layout); for every period p = 1 .. 500 a thread XORs its word with the word p bases further on (two LDS
//     reads + a funnel shift) and tests its aligned blocks of 8 positions for "all equal"; periods >= 64 look at every other
//     word only.  Round 5: the compared word slides along a 64-bit register window with compile-time shifts (~10 integer
//     operations per (word, period), one LDS read per array every 16 periods); the rare seed goes through a call.
//   extension: the leftmost seed of a run aligns the stretch with itself one period on by the banded end extension of the copy
//     finder (hite_ext.h; S = 2 i - 7 cost: match 2, edit 5 -- TRF's 7 is a penalty against a consensus, a copy against its
//     neighbour carries twice the divergence; calibrated on TRF's own masks, see the twin), in the thread that found the seed;
//   accepted stretches are OR-ed into a bit map, which is then OR-ed into the genome's "not A/C/G/T" mask: every later stage
//     sees N where the reference's later stages read the masked FASTA.
#include "hite_common.h"
#include "hite_ext.h"

#define TR_MAXEXT 4096
#define TR_MINSCORE 50
#define TR_RESEED 2048
#define TR_TILE 256          // words per workgroup
#define TR_HALO 34           // words behind the tile a period of <= 500 bases reaches (32) + the funnel's second word + 1

// A seed's extension is the rare, heavy path of the scan: two banded extensions (thousands of instructions) in the ONE lane that found the
// seed while its 63 neighbours wait, 0.3 times per wavefront and tile on random sequence -- as much as the wavefront's whole scan.  The
// kernel therefore only LISTS the seeds (start, period), and tr_extend_kernel extends them a lane each, 64 side by side; the mask bits
// are OR-ed, so the order does not matter.  A full list (cap entries) sends the seed down the old path.
__device__ __forceinline__ bool tr_defer_push(unsigned long long *list, unsigned long long cap, unsigned long long *count, int64_t s, int p) {
    const unsigned long long at = atomicAdd(count, 1ull);
    if (at >= cap) return false;
    list[at] = ((unsigned long long)s << 10) | (unsigned long long)p;
    return true;
}

// >>> tr_seed (tests/test_host_compiled.py compiles this block for the host and runs it thread by thread against the twin)
__device__ __forceinline__ int tr_contig_of(const int64_t *__restrict__ coff, int nc, int64_t g) {
    int lo = 0, hi = nc;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (coff[mid] <= g) lo = mid; else hi = mid; }
    return lo;
}
// bit k of a 16-bit mask -> bit 2k
__device__ __forceinline__ uint32_t spread16(uint32_t x) {
    x &= 0xffffu;
    x = (x | (x << 8)) & 0x00ff00ffu;
    x = (x | (x << 4)) & 0x0f0f0f0fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
__device__ __forceinline__ uint32_t tr_funnel(uint32_t hi, uint32_t lo, int sh) { return sh ? (lo >> sh) | (hi << (32 - sh)) : lo; }

struct TrTile {
    uint32_t b[TR_TILE + TR_HALO + 2], nx[TR_TILE + TR_HALO + 2];   // index 0 = word w0 - 1
    unsigned long long *dlist, dcap, *dcount;                        // where seeds are listed for tr_extend_kernel (null: extended on the spot)
};

// entry k of the tile of the workgroup that owns words w0 .. w0 + TR_TILE - 1
__device__ __forceinline__ void tr_tile_load(TrTile &T, int k, int64_t w0, int64_t nwords, int64_t G, const uint32_t *__restrict__ bases,
                                             const uint32_t *__restrict__ nmask) {
    const int64_t w = w0 - 1 + k;
    uint32_t b = 0u, nx = 0x55555555u;
    if (w >= 0 && w < nwords) {
        b = bases[w];
        nx = spread16(nmask[w >> 1] >> (16 * (int)(w & 1)));
        const int64_t rest = G - (w << 4);                 // positions beyond the genome never match
        if (rest < 16) nx |= 0x55555555u << (2 * (int)rest);
    }
    T.b[k] = b; T.nx[k] = nx;
}

// "mismatch or invalid" flags (even bits) of the 16 positions of word index wi (tile-relative, 1-based on w0 - 1) at period p
__device__ __forceinline__ uint32_t tr_bad(const TrTile &T, int wi, int p) {
    const int q = wi + (p >> 4), sh = 2 * (p & 15);
    const uint32_t sb = tr_funnel(T.b[q + 1], T.b[q], sh), sn = tr_funnel(T.nx[q + 1], T.nx[q], sh);
    const uint32_t x = T.b[wi] ^ sb;
    return ((x | (x >> 1)) & 0x55555555u) | T.nx[wi] | sn;
}

// the stretch around seed block s at period p: extension both ways, acceptance, mask bits
__device__ void tr_extend(int64_t s, int p, const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask,
                          const int64_t *__restrict__ coff, int nc, uint32_t *__restrict__ trmask) {
    const int c = tr_contig_of(coff, nc, s);
    const int64_t cb = coff[c], ce = coff[c + 1];
    if (s + p > ce) return;
    int64_t nr = ce - (s + p); if (nr > TR_MAXEXT) nr = TR_MAXEXT; if (nr < 0) nr = 0;
    int64_t nl = s - cb; if (nl > TR_MAXEXT) nl = TR_MAXEXT;
    const int lim = p - 1 < EXT_B ? p - 1 : EXT_B;
    int ir, tr, sr, il, tl, sl;
    ext_align_dev<ExtTandemMode>(nullptr, s, +1, false, (int)nr, bases, nmask, s + p, +1, ce - (s + p), -lim, EXT_B, &ir, &tr, &sr, nullptr);
    ext_align_dev<ExtTandemMode>(nullptr, s - 1, -1, false, (int)nl, bases, nmask, s + p, -1, s + p - cb, -EXT_B, lim, &il, &tl, &sl, nullptr);
    if (il + ir <= 0) return;
    if (sl + sr + 2 * p < TR_MINSCORE) return;
    if (il + ir < (85 * p + 99) / 100) return;
    int64_t a = s - il, hi = s + tr - 1 + p;
    if (hi >= ce) hi = ce - 1;
    for (int64_t w = a >> 5; w <= (hi >> 5); w++) {
        const int64_t lo_b = w << 5;
        uint32_t m = 0xffffffffu;
        if (a > lo_b) m &= 0xffffffffu << (int)(a - lo_b);
        if (hi < lo_b + 31) m &= 0xffffffffu >> (int)(lo_b + 31 - hi);
        atomicOr(&trmask[w], m);
    }
}

// block blk (0 / 1: positions 0-7 / 8-15) of word wl of the tile at w0 is a seed at period p (`bad` = its word's flags): the rare
// path.  The leftmost seed of a run extends; a run restarts every TR_RESEED bases and at a contig start.
__device__ __noinline__ void tr_seed_hit(const TrTile &T, int wl, int blk, uint32_t bad, int p, int64_t w0, int64_t G,
                                         const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask,
                                         const int64_t *__restrict__ coff, int nc, uint32_t *__restrict__ trmask) {
    const int st = p < 32 ? 8 : (p < 64 ? 16 : 32);
    const int64_t w = w0 + wl;
    const int64_t s = (w << 4) + 8 * blk;
    if (s + 8 > G) return;
    // previous block of the stride: a seed as well -> this one is not the leftmost of its run
    bool prev;
    if (st == 8 && blk == 1) prev = (bad & 0x5555u) == 0u;
    else if (s - st < 0) prev = false;
    else {
        const int64_t pwi = (s - st) >> 4;                      // the previous block's word: at most two words back
        const int pw = (int)(pwi - (w0 - 1));                    // its tile index (the tile starts one word early)
        uint32_t pb;
        if (pw >= 0) pb = tr_bad(T, pw, p);
        else {   // two words back of the tile's first word: from global memory (once per tile and period)
            const int64_t q = pwi + (p >> 4);
            const int sh = 2 * (p & 15);
            const uint32_t sb = tr_funnel(bases[q + 1], bases[q], sh);
            const uint32_t n0 = spread16(nmask[pwi >> 1] >> (16 * (int)(pwi & 1)));
            const uint32_t n1 = spread16(nmask[q >> 1] >> (16 * (int)(q & 1))), n2 = spread16(nmask[(q + 1) >> 1] >> (16 * (int)((q + 1) & 1)));
            const uint32_t x = bases[pwi] ^ sb;
            pb = ((x | (x >> 1)) & 0x55555555u) | n0 | tr_funnel(n2, n1, sh);
        }
        prev = ((pb >> (((s - st) & 8) ? 16 : 0)) & 0x5555u) == 0u;
    }
    if (prev && (s % TR_RESEED) != 0 && s - st >= coff[tr_contig_of(coff, nc, s)]) return;   // (a run may cross a contig border)
    if (T.dlist && tr_defer_push(T.dlist, T.dcap, T.dcount, s, p)) return;
    tr_extend(s, p, bases, nmask, coff, nc, trmask);
}

// one group of 16 periods (16 g .. 16 g + 15) of word wl: the periods JPAR, JPAR + JSTEP, ... of the group, all of them inside
// the range (no bound tests: the shifts and the trip count are compile-time constants); TWO: both 8-blocks (periods below 32).
// The seeds of the group are collected as bits and handled behind ONE branch: a test-and-branch per period cost as much as
// the period's arithmetic.
template <int JSTEP, int JPAR, bool TWO>
__device__ __forceinline__ void tr_group_full(const TrTile &T, int wl, int g, uint32_t bw, uint32_t nw, int64_t w0, int64_t G,
                                              const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask,
                                              const int64_t *__restrict__ coff, int nc, uint32_t *__restrict__ trmask) {
    const int wi = wl + 1;
    const uint32_t blo = T.b[wi + g], bhi = T.b[wi + g + 1], nlo = T.nx[wi + g], nhi = T.nx[wi + g + 1];
    uint32_t hit0 = 0u, hit1 = 0u;
#pragma unroll
    for (int j = JPAR; j < 16; j += JSTEP) {
        const uint32_t sb = j ? (blo >> (2 * j)) | (bhi << (32 - 2 * j)) : blo;
        const uint32_t sn = j ? (nlo >> (2 * j)) | (nhi << (32 - 2 * j)) : nlo;
        const uint32_t x = bw ^ sb;
        const uint32_t bad = ((x | (x >> 1)) & 0x55555555u) | nw | sn;
        hit0 |= (bad & 0x5555u) == 0u ? 1u << j : 0u;
        if (TWO) hit1 |= (bad & 0x55550000u) == 0u ? 1u << j : 0u;
    }
    if ((hit0 | hit1) != 0u) {
        for (int j = JPAR; j < 16; j += JSTEP) {
            if (!(((hit0 | hit1) >> j) & 1u)) continue;
            const int p = 16 * g + j;
            const uint32_t bad = tr_bad(T, wi, p);
            if ((hit0 >> j) & 1u) tr_seed_hit(T, wl, 0, bad, p, w0, G, bases, nmask, coff, nc, trmask);
            if ((hit1 >> j) & 1u) tr_seed_hit(T, wl, 1, bad, p, w0, G, bases, nmask, coff, nc, trmask);
        }
    }
}
// the periods p_lo, p_lo + step, ... <= p_hi (step 1 or 2) of word wl of the tile: the word it is compared with slides along a
// 64-bit window held in registers (one LDS read per array every 16 periods instead of four per period), the shifts inside a
// group of 16 periods are compile-time constants; whole groups run without a test per period, the ragged first / last group with
__device__ __forceinline__ void tr_scan_word(const TrTile &T, int wl, int p_lo, int p_hi, int step, int64_t w0, int64_t G,
                                             const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask,
                                             const int64_t *__restrict__ coff, int nc, uint32_t *__restrict__ trmask) {
    const int wi = wl + 1;
    const uint32_t bw = T.b[wi], nw = T.nx[wi];
    const int par = p_lo & 1;
    for (int g = p_lo >> 4; g <= (p_hi >> 4); g++) {
        if (16 * g >= p_lo && 16 * g + 15 <= p_hi && (step == 1 || g >= 2)) {
            if (step == 1) {
                if (g < 2) tr_group_full<1, 0, true>(T, wl, g, bw, nw, w0, G, bases, nmask, coff, nc, trmask);
                else tr_group_full<1, 0, false>(T, wl, g, bw, nw, w0, G, bases, nmask, coff, nc, trmask);
            } else {        // (step 2 starts at period 64: one 8-block per period)
                if (par) tr_group_full<2, 1, false>(T, wl, g, bw, nw, w0, G, bases, nmask, coff, nc, trmask);
                else tr_group_full<2, 0, false>(T, wl, g, bw, nw, w0, G, bases, nmask, coff, nc, trmask);
            }
            continue;
        }
        const uint32_t blo = T.b[wi + g], bhi = T.b[wi + g + 1], nlo = T.nx[wi + g], nhi = T.nx[wi + g + 1];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int p = 16 * g + j;
            if (p < p_lo || p > p_hi || (step == 2 && (j & 1) != par)) continue;
            const uint32_t sb = j ? (blo >> (2 * j)) | (bhi << (32 - 2 * j)) : blo;
            const uint32_t sn = j ? (nlo >> (2 * j)) | (nhi << (32 - 2 * j)) : nlo;
            const uint32_t x = bw ^ sb;
            const uint32_t bad = ((x | (x >> 1)) & 0x55555555u) | nw | sn;
            if ((bad & 0x5555u) == 0u) tr_seed_hit(T, wl, 0, bad, p, w0, G, bases, nmask, coff, nc, trmask);
            if (p < 32 && (bad & 0x55550000u) == 0u) tr_seed_hit(T, wl, 1, bad, p, w0, G, bases, nmask, coff, nc, trmask);
        }
    }
}
// everything thread `tid` of the workgroup that owns words w0 .. does: every period below 64 on its own word (both 8-blocks
// below 32, the first one from 32 on), the periods from 64 on of one parity on every other word (first 8-block)
__device__ __forceinline__ void tr_thread(const TrTile &T, int tid, int64_t w0, int64_t nwords, int64_t G, int max_period,
                                          const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask,
                                          const int64_t *__restrict__ coff, int nc, uint32_t *__restrict__ trmask) {
    if (w0 + tid < nwords) tr_scan_word(T, tid, 1, max_period < 63 ? max_period : 63, 1, w0, G, bases, nmask, coff, nc, trmask);
    const int wl = 2 * (tid & 127), p0 = 64 + (tid >> 7);        // (w0 is even: the tile starts on a 32-base boundary)
    if (p0 <= max_period && w0 + wl < nwords) tr_scan_word(T, wl, p0, max_period, 2, w0, G, bases, nmask, coff, nc, trmask);
}
// <<< tr_seed

__global__ void __launch_bounds__(256) tr_seed_kernel(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask,
                                                      const int64_t *__restrict__ coff, int nc, int64_t G, int max_period,
                                                      uint32_t *__restrict__ trmask, unsigned long long *__restrict__ dlist,
                                                      unsigned long long dcap, unsigned long long *__restrict__ dcount) {
    __shared__ TrTile T;
    if (threadIdx.x == 0) { T.dlist = dlist; T.dcap = dcap; T.dcount = dcount; }
    const int64_t nwords = (G + 15) >> 4;
    for (int64_t w0 = (int64_t)blockIdx.x * TR_TILE; w0 < nwords; w0 += (int64_t)gridDim.x * TR_TILE) {
        __syncthreads();
        for (int k = threadIdx.x; k < TR_TILE + TR_HALO + 2; k += 256) tr_tile_load(T, k, w0, nwords, G, bases, nmask);
        __syncthreads();
        tr_thread(T, (int)threadIdx.x, w0, nwords, G, max_period, bases, nmask, coff, nc, trmask);
    }
}

// the listed seeds, a lane each
__global__ void __launch_bounds__(256) tr_extend_kernel(const unsigned long long *__restrict__ dlist, unsigned long long dcap,
                                                        const unsigned long long *__restrict__ dcount, const uint32_t *__restrict__ bases,
                                                        const uint32_t *__restrict__ nmask, const int64_t *__restrict__ coff, int nc,
                                                        uint32_t *__restrict__ trmask) {
    unsigned long long n = *dcount;
    if (n > dcap) n = dcap;
    for (unsigned long long e = (unsigned long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (unsigned long long)gridDim.x * 256) {
        const unsigned long long v = dlist[e];
        tr_extend((int64_t)(v >> 10), (int)(v & 1023ull), bases, nmask, coff, nc, trmask);
    }
}

// masked bases -> the genome's "not A/C/G/T" mask; count of newly masked-or-not bases of the tandem mask
__global__ void tr_apply_kernel(int64_t nw32, const uint32_t *__restrict__ trmask, uint32_t *__restrict__ nmask, unsigned long long *__restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int c = 0;
    if (i < nw32) { const uint32_t m = trmask[i]; if (m) { nmask[i] |= m; c = __popc(m); } }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, (unsigned long long)c);
}

// Tandem repeats of the resident genome (periods 1 .. max_period, TRF's 2 7 7 scheme, score >= 50) become N for every later
// stage.  mask_bits_host (optional): (n_bases + 31) / 32 words, bit i & 31 of word i >> 5 = base i of the concatenated contigs
// is masked -- what the host side needs to write the reference's .mask FASTA.
extern "C" int hite_tr_mask(hite_ctx *ctx, int32_t max_period, uint32_t *mask_bits_host, int64_t *masked_bases_out) {
    if (!ctx || !ctx->d_bases || max_period < 1 || max_period > 500) return HITE_EINVAL;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    const int64_t G = ctx->n_bases;
    const int64_t nw32 = (G + 31) >> 5;
    uint32_t *trmask = nullptr;
    unsigned long long *cnt = nullptr;
    const int64_t cnt_at = (nw32 + 3) & ~(int64_t)1;          // 8-byte aligned word index behind the bit map
    // the seed list lives behind the bit map and its counter: one entry per 64 bases (at least 2^20); HITE_TR_DEFER=0: no list
    static const bool defer = [] { const char *v = getenv("HITE_TR_DEFER"); return !(v && *v == '0'); }();
    const unsigned long long dcap = defer ? (unsigned long long)((G >> 6) > (1ll << 20) ? (G >> 6) : (1ll << 20)) : 0ull;
    HITE_CHECK(ctx, hipMalloc((void **)&trmask, (size_t)(cnt_at + 4) * 4 + (size_t)dcap * 8));
    hipError_t e = hipMemset(trmask, 0, (size_t)(cnt_at + 4) * 4);
    cnt = (unsigned long long *)(trmask + cnt_at);
    unsigned long long *dcount = cnt + 1, *dlist = dcap ? (unsigned long long *)(trmask + cnt_at + 4) : nullptr;
    int rc = HITE_OK;
    if (e == hipSuccess) {
        const int64_t nwords = (G + 15) >> 4;
        int64_t blocks = (nwords + TR_TILE - 1) / TR_TILE;
        if (blocks > 65536) blocks = 65536;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(tr_seed_kernel, dim3((unsigned)blocks), dim3(256), 0, nullptr, ctx->d_bases, ctx->d_nmask, ctx->d_contig_off, ctx->n_contigs, G,
                           (int)max_period, trmask, dlist, dcap, dcount);
        if (dlist) hipLaunchKernelGGL(tr_extend_kernel, dim3(2048), dim3(256), 0, nullptr, dlist, dcap, dcount, ctx->d_bases, ctx->d_nmask, ctx->d_contig_off,
                                      ctx->n_contigs, trmask);
        hipLaunchKernelGGL(tr_apply_kernel, dim3((unsigned)((nw32 + 255) / 256)), dim3(256), 0, nullptr, nw32, trmask, ctx->d_nmask, cnt);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        unsigned long long n = 0;
        if (e == hipSuccess) e = hipMemcpy(&n, cnt, 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess && mask_bits_host) e = hipMemcpy(mask_bits_host, trmask, (size_t)nw32 * 4, hipMemcpyDeviceToHost);
        if (masked_bases_out) *masked_bases_out = (int64_t)n;
    }
    ctx->genome_epoch++;        // (kept minimizer tiles of an index build are void)
    ctx->mask_log_n = 0;
    (void)hipFree(trmask);
    if (e != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "hite_tr_mask: %s", hipGetErrorString(e)); rc = HITE_EHIP; }
    return rc;
}
