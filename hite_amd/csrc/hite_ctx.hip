// hite_ctx.hip -- context, resident 2-bit genome, flank-window gather.
//
// Data layout in HBM
//   d_bases : 2 bits per base, 16 bases per u32, A=0 C=1 G=2 T=3, contigs concatenated
//   d_nmask : 1 bit per base, 32 per u32, set where the input byte is not A/C/G/T
//   d_contig_off : int64 base index of each contig (+ total)
// Algorithmic bytes: pack reads G bytes, writes G/4 + G/8; gather reads W/4 + W/8 per window
// and writes W (+1000 for the first500+last500 form).  Both are HBM-bound streaming kernels.
#include "hite_common.h"
#include "hite_align.h"
#include "hite_genome.h"

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" int hite_version(void) { return 1; }

extern "C" int hite_ctx_create(int device_id, hite_ctx **out) {
    if (!out) return HITE_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return HITE_ENODEV;
    if (device_id < 0 || device_id >= ndev) return HITE_EINVAL;
    hite_ctx *c = (hite_ctx *)calloc(1, sizeof(hite_ctx));
    if (!c) return HITE_ENOMEM;
    c->device = device_id;
    c->copy_interval = -1;
    if (hipSetDevice(device_id) != hipSuccess) { free(c); return HITE_EHIP; }
    *out = c;
    return HITE_OK;
}

static void free_genome(hite_ctx *c) {
    if (c->d_contig_rank) { (void)hipFree(c->d_contig_rank); c->d_contig_rank = nullptr; }
    if (c->d_bases) (void)hipFree(c->d_bases);
    if (c->d_nmask) (void)hipFree(c->d_nmask);
    if (c->d_contig_off) (void)hipFree(c->d_contig_off);
    free(c->h_contig_off);
    c->d_bases = c->d_nmask = nullptr;
    c->d_contig_off = nullptr;
    c->h_contig_off = nullptr;
    c->n_contigs = 0;
    c->n_bases = 0;
    c->genome_epoch++;
    c->mask_log_n = 0;
}

extern "C" void hite_ctx_destroy(hite_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    free_genome(c);
    free(c->mask_log);
    hite_align_release(c);
    hite_fmea_release(c);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->d_scratch2) (void)hipFree(c->d_scratch2);
    if (c->d_contig_rank) (void)hipFree(c->d_contig_rank);
    if (c->aux_fork) (void)hipEventDestroy((hipEvent_t)c->aux_fork);
    for (int i = 0; i < HITE_AUX_STREAMS; i++) {
        if (c->aux_join[i]) (void)hipEventDestroy((hipEvent_t)c->aux_join[i]);
        if (c->aux_stream[i]) (void)hipStreamDestroy((hipStream_t)c->aux_stream[i]);
    }
    free(c);
}

extern "C" const char *hite_last_error(hite_ctx *c) { return c ? c->err : "null ctx"; }

// Byte order of the contig NAMES (each followed by ':'): the rows of an alignment are named "<contig>:<start>-<end>(<strand>)" and
// tools/ready_for_MSA.sh breaks length ties by name (hite_pipeline.hip, select_rows_kernel).  rank[i] = position of contig i in
// that order; NULL or n != the number of packed contigs clears it (then contigs compare by their index).  Packing a genome clears it.
extern "C" int hite_set_contig_order(hite_ctx *ctx, const int32_t *rank, int32_t n) {
    if (!ctx) return HITE_EINVAL;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    if (ctx->d_contig_rank) { (void)hipFree(ctx->d_contig_rank); ctx->d_contig_rank = nullptr; }
    if (!rank || n <= 0 || n != ctx->n_contigs) return rank && n > 0 ? HITE_EINVAL : HITE_OK;
    HITE_CHECK(ctx, hipMalloc((void **)&ctx->d_contig_rank, (size_t)n * 4));
    HITE_CHECK(ctx, hipMemcpy(ctx->d_contig_rank, rank, (size_t)n * 4, hipMemcpyHostToDevice));
    return HITE_OK;
}

// the context's side streams (created on first use, all or none) with one event to fork them off the caller's stream and one
// event per stream to join it; st / join_ev receive k <= HITE_AUX_STREAMS entries
int hite_aux_streams(hite_ctx *ctx, int k, hipStream_t *st, hipEvent_t *fork_ev, hipEvent_t *join_ev) {
    if (k < 0 || k > HITE_AUX_STREAMS) return HITE_EINVAL;
    if (!ctx->aux_fork) {
        hipStream_t s[HITE_AUX_STREAMS] = {};
        hipEvent_t j[HITE_AUX_STREAMS] = {}, f = nullptr;
        hipError_t e = hipEventCreateWithFlags(&f, hipEventDisableTiming);
        for (int i = 0; i < HITE_AUX_STREAMS && e == hipSuccess; i++) {
            e = hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&j[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) {      // nothing half-made is published
            if (f) (void)hipEventDestroy(f);
            for (int i = 0; i < HITE_AUX_STREAMS; i++) { if (j[i]) (void)hipEventDestroy(j[i]); if (s[i]) (void)hipStreamDestroy(s[i]); }
            HITE_CHECK(ctx, e);
        }
        for (int i = 0; i < HITE_AUX_STREAMS; i++) { ctx->aux_stream[i] = s[i]; ctx->aux_join[i] = j[i]; }
        ctx->aux_fork = f;
    }
    for (int i = 0; i < k; i++) { st[i] = (hipStream_t)ctx->aux_stream[i]; join_ev[i] = (hipEvent_t)ctx->aux_join[i]; }
    *fork_ev = (hipEvent_t)ctx->aux_fork;
    return HITE_OK;
}

int hite_scratch_reserve(hite_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->scratch_bytes) {
        if (ctx->d_scratch) HITE_CHECK(ctx, hipFree(ctx->d_scratch));
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
        size_t want = bytes + bytes / 4 + 4096;
        HITE_CHECK(ctx, hipMalloc(&ctx->d_scratch, want));
        ctx->scratch_bytes = want;
    }
    *out = ctx->d_scratch;
    return HITE_OK;
}
int hite_scratch2_reserve(hite_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->scratch2_bytes) {
        if (ctx->d_scratch2) HITE_CHECK(ctx, hipFree(ctx->d_scratch2));
        ctx->d_scratch2 = nullptr;
        ctx->scratch2_bytes = 0;
        size_t want = bytes + bytes / 4 + 4096;
        HITE_CHECK(ctx, hipMalloc(&ctx->d_scratch2, want));
        ctx->scratch2_bytes = want;
    }
    *out = ctx->d_scratch2;
    return HITE_OK;
}

// ---------------------------------------------------------------------------------------------
// genome pack: each thread converts 32 bases (two 16-byte loads) -> 2 x u32 bases + 1 x u32 mask
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack4(uint32_t w, uint32_t &bits, uint32_t &mask, int shift) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t c = (w >> (8 * i)) & 0xffu;
        uint32_t code = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
        bits |= (code & 3u) << (2 * (shift + i));
        mask |= (code >> 2) << (shift + i);
    }
}

__global__ void __launch_bounds__(256) pack_genome_kernel(const uint8_t *__restrict__ seq, int64_t n,
                                                          uint32_t *__restrict__ bases, uint32_t *__restrict__ nmask) {
    int64_t nw = (n + 31) >> 5;  // mask words
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += (int64_t)gridDim.x * blockDim.x) {
        int64_t g = w << 5;
        uint32_t b0 = 0, b1 = 0, m0 = 0, m1 = 0;
        if (g + 32 <= n) {
            const uint4 *p = reinterpret_cast<const uint4 *>(seq + g);
            uint4 x = p[0], y = p[1];
            pack4(x.x, b0, m0, 0); pack4(x.y, b0, m0, 4); pack4(x.z, b0, m0, 8); pack4(x.w, b0, m0, 12);
            pack4(y.x, b1, m1, 0); pack4(y.y, b1, m1, 4); pack4(y.z, b1, m1, 8); pack4(y.w, b1, m1, 12);
        } else {
            for (int i = 0; i < 32 && g + i < n; i++) {
                uint32_t c = seq[g + i];
                uint32_t code = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
                if (i < 16) { b0 |= (code & 3u) << (2 * i); m0 |= (code >> 2) << i; }
                else { b1 |= (code & 3u) << (2 * (i - 16)); m1 |= (code >> 2) << (i - 16); }
            }
        }
        reinterpret_cast<uint2 *>(bases)[w] = make_uint2(b0, b1);
        nmask[w] = m0 | (m1 << 16);
    }
}

static int genome_alloc(hite_ctx *ctx, const int64_t *contig_off, int32_t n_contigs) {
    if (n_contigs <= 0 || !contig_off || contig_off[0] != 0) return HITE_EINVAL;
    for (int i = 0; i < n_contigs; i++) if (contig_off[i + 1] < contig_off[i]) return HITE_EINVAL;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    free_genome(ctx);
    int64_t n = contig_off[n_contigs];
    int64_t nw = (n + 31) >> 5;
    HITE_CHECK(ctx, hipMalloc((void **)&ctx->d_bases, (size_t)(nw * 2 + 8) * 4));
    HITE_CHECK(ctx, hipMalloc((void **)&ctx->d_nmask, (size_t)(nw + 8) * 4));
    HITE_CHECK(ctx, hipMemset(ctx->d_bases, 0, (size_t)(nw * 2 + 8) * 4));
    HITE_CHECK(ctx, hipMemset(ctx->d_nmask, 0, (size_t)(nw + 8) * 4));
    HITE_CHECK(ctx, hipMalloc((void **)&ctx->d_contig_off, sizeof(int64_t) * (n_contigs + 1)));
    HITE_CHECK(ctx, hipMemcpy(ctx->d_contig_off, contig_off, sizeof(int64_t) * (n_contigs + 1), hipMemcpyHostToDevice));
    ctx->h_contig_off = (int64_t *)malloc(sizeof(int64_t) * (n_contigs + 1));
    if (!ctx->h_contig_off) return HITE_ENOMEM;
    memcpy(ctx->h_contig_off, contig_off, sizeof(int64_t) * (n_contigs + 1));
    ctx->n_contigs = n_contigs;
    ctx->n_bases = n;
    return HITE_OK;
}

extern "C" int hite_genome_pack_dev(hite_ctx *ctx, const uint8_t *d_seq, const int64_t *contig_off_host,
                                    int32_t n_contigs, void *stream) {
    if (!ctx || !d_seq) return HITE_EINVAL;
    if (((uintptr_t)d_seq) & 15) return HITE_EINVAL;
    int rc = genome_alloc(ctx, contig_off_host, n_contigs);
    if (rc) return rc;
    int64_t nw = (ctx->n_bases + 31) >> 5;
    if (nw == 0) return HITE_OK;
    int64_t blocks = (nw + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(pack_genome_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_seq,
                       ctx->n_bases, ctx->d_bases, ctx->d_nmask);
    HITE_CHECK(ctx, hipGetLastError());
    return HITE_OK;
}

extern "C" int hite_genome_pack(hite_ctx *ctx, const uint8_t *seq, const int64_t *contig_off, int32_t n_contigs) {
    if (!ctx || !seq || !contig_off || n_contigs <= 0) return HITE_EINVAL;
    int64_t n = contig_off[n_contigs];
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    uint8_t *d = nullptr;
    HITE_CHECK(ctx, hipMalloc((void **)&d, (size_t)n + 64));
    hipError_t e = hipMemcpy(d, seq, (size_t)n, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? hite_genome_pack_dev(ctx, d, contig_off, n_contigs, nullptr) : HITE_EHIP;
    if (rc == HITE_OK && hipDeviceSynchronize() != hipSuccess) rc = HITE_EHIP;
    (void)hipFree(d);
    return rc;
}

extern "C" int64_t hite_genome_bases(hite_ctx *ctx) { return ctx ? ctx->n_bases : 0; }

// ---------------------------------------------------------------------------------------------
// N-masking of intervals of the resident genome (mask_genome_intactTE, Util.py:6389-6431: the copies of already
// found TEs are replaced by N before the next chunk is searched).  One block per interval, threads over mask words.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) genome_mask_kernel(int64_t n, const int32_t *__restrict__ contig, const int64_t *__restrict__ s1,
                                                          const int64_t *__restrict__ e1, const int64_t *__restrict__ coff, int nc,
                                                          uint32_t *__restrict__ nmask) {
    const int64_t k = blockIdx.x;
    if (k >= n) return;
    const int c = contig[k];
    if (c < 0 || c >= nc) return;
    int64_t a = s1[k] - 1, b = e1[k];          // 0-based half open inside the contig (python slice semantics: clamped)
    const int64_t L = coff[c + 1] - coff[c];
    if (a < 0) a = 0;
    if (b > L) b = L;
    if (b <= a) return;
    a += coff[c]; b += coff[c];
    for (int64_t w = (a >> 5) + threadIdx.x; w <= ((b - 1) >> 5); w += 256) {
        const int64_t g = w << 5;
        const int lo = a > g ? (int)(a - g) : 0, hi = b < g + 32 ? (int)(b - g) : 32;
        const uint32_t m = (hi - lo == 32) ? 0xffffffffu : (((1u << (hi - lo)) - 1u) << lo);
        atomicOr(&nmask[w], m);
    }
}
extern "C" int hite_genome_mask(hite_ctx *ctx, int64_t n, const int32_t *contig, const int64_t *start1, const int64_t *end1) {
    if (!ctx || !ctx->d_bases || n < 0 || (n > 0 && (!contig || !start1 || !end1))) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    // the log of what gets masked (the kernel's own clamping), for the index build that follows
    if (ctx->mask_log_n + n > ctx->mask_log_cap) {
        const int64_t cap = 2 * (ctx->mask_log_n + n) + 1024;
        int64_t *q = (int64_t *)realloc(ctx->mask_log, (size_t)cap * 2 * sizeof(int64_t));
        if (!q) return HITE_ENOMEM;
        ctx->mask_log = q; ctx->mask_log_cap = cap;
    }
    for (int64_t k = 0; k < n; k++) {
        const int c = contig[k];
        if (c < 0 || c >= ctx->n_contigs) continue;
        int64_t a = start1[k] - 1, b = end1[k];
        const int64_t L = ctx->h_contig_off[c + 1] - ctx->h_contig_off[c];
        if (a < 0) a = 0;
        if (b > L) b = L;
        if (b <= a) continue;
        ctx->mask_log[2 * ctx->mask_log_n] = a + ctx->h_contig_off[c];
        ctx->mask_log[2 * ctx->mask_log_n + 1] = b + ctx->h_contig_off[c];
        ctx->mask_log_n++;
    }
    int32_t *dc = nullptr; int64_t *ds = nullptr, *de = nullptr;
    hipError_t e = hipMalloc((void **)&dc, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&ds, (size_t)n * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&de, (size_t)n * 8);
    if (e == hipSuccess) e = hipMemcpy(dc, contig, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ds, start1, (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(de, end1, (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(genome_mask_kernel, dim3((unsigned)n), dim3(256), 0, nullptr, n, dc, ds, de, ctx->d_contig_off, ctx->n_contigs,
                           ctx->d_nmask);
        e = hipDeviceSynchronize();
    }
    if (dc) (void)hipFree(dc);
    if (ds) (void)hipFree(ds);
    if (de) (void)hipFree(de);
    HITE_CHECK(ctx, e);
    return HITE_OK;
}

// ---------------------------------------------------------------------------------------------
// flank-window gather
// ---------------------------------------------------------------------------------------------
// one wavefront per copy
__global__ void __launch_bounds__(256) flank_gather_kernel(const uint32_t *__restrict__ bases,
                                                           const uint32_t *__restrict__ nmask,
                                                           const int64_t *__restrict__ coff, int32_t ncontig, int64_t n,
                                                           const int32_t *__restrict__ contig,
                                                           const int64_t *__restrict__ s1, const int64_t *__restrict__ e1,
                                                           const uint8_t *__restrict__ minus, int32_t flank,
                                                           const int64_t *__restrict__ out_off, uint8_t *__restrict__ out,
                                                           const int64_t *__restrict__ trunc_off,
                                                           uint8_t *__restrict__ trunc_out) {
    int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    int lane = threadIdx.x & 63;
    int64_t len, tlen, g_lo;
    window_rule(coff, ncontig, contig[i], s1[i], e1[i], flank, len, tlen, g_lo);
    if (len == 0) return;
    bool mn = minus[i] != 0;
    emit_span(bases, nmask, g_lo, len, mn, 0, len, out + out_off[i], lane);
    if (tlen && trunc_out) {
        uint8_t *t = trunc_out + trunc_off[i];
        emit_span(bases, nmask, g_lo, len, mn, 0, 500, t, lane);
        emit_span(bases, nmask, g_lo, len, mn, len - 500, 500, t + 500, lane);
    }
}

extern "C" int hite_flank_gather_dev(hite_ctx *ctx, int64_t n, const int32_t *d_contig, const int64_t *d_start1,
                                     const int64_t *d_end1, const uint8_t *d_minus, int32_t flank,
                                     const int64_t *d_out_off, uint8_t *d_out, const int64_t *d_trunc_off,
                                     uint8_t *d_trunc_out, void *stream) {
    if (!ctx || !ctx->d_bases) return HITE_EINVAL;
    if (n <= 0) return HITE_OK;
    int64_t blocks = (n + 3) / 4;
    hipLaunchKernelGGL(flank_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ctx->d_bases,
                       ctx->d_nmask, ctx->d_contig_off, ctx->n_contigs, n, d_contig, d_start1, d_end1, d_minus, flank,
                       d_out_off, d_out, d_trunc_off, d_trunc_out);
    HITE_CHECK(ctx, hipGetLastError());
    return HITE_OK;
}

extern "C" int hite_flank_sizes_dev(hite_ctx *ctx, int64_t n, const int32_t *d_contig, const int64_t *d_start1,
                                    const int64_t *d_end1, int32_t flank, int64_t *d_out_len, int64_t *d_trunc_len,
                                    void *stream) {
    if (!ctx || !ctx->d_bases) return HITE_EINVAL;
    if (n <= 0) return HITE_OK;
    hipLaunchKernelGGL(flank_sizes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       ctx->d_contig_off, ctx->n_contigs, n, d_contig, d_start1, d_end1, flank, d_out_len, d_trunc_len);
    HITE_CHECK(ctx, hipGetLastError());
    return HITE_OK;
}

// host-buffer wrappers -------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
    hipError_t up(const void *h, size_t n) {
        hipError_t e = alloc(n);
        if (e != hipSuccess) return e;
        return n ? hipMemcpy(p, h, n, hipMemcpyHostToDevice) : hipSuccess;
    }
};

extern "C" int hite_flank_sizes(hite_ctx *ctx, int64_t n, const int32_t *contig, const int64_t *start1,
                                const int64_t *end1, int32_t flank, int64_t *out_len, int64_t *trunc_len) {
    if (!ctx || !ctx->d_bases || n < 0) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    DevBuf c, s, e, ol, tl;
    HITE_CHECK(ctx, c.up(contig, n * 4));
    HITE_CHECK(ctx, s.up(start1, n * 8));
    HITE_CHECK(ctx, e.up(end1, n * 8));
    HITE_CHECK(ctx, ol.alloc(n * 8));
    HITE_CHECK(ctx, tl.alloc(n * 8));
    int rc = hite_flank_sizes_dev(ctx, n, (int32_t *)c.p, (int64_t *)s.p, (int64_t *)e.p, flank, (int64_t *)ol.p,
                                  (int64_t *)tl.p, nullptr);
    if (rc) return rc;
    HITE_CHECK(ctx, hipMemcpy(out_len, ol.p, n * 8, hipMemcpyDeviceToHost));
    if (trunc_len) HITE_CHECK(ctx, hipMemcpy(trunc_len, tl.p, n * 8, hipMemcpyDeviceToHost));
    return HITE_OK;
}

extern "C" int hite_flank_gather(hite_ctx *ctx, int64_t n, const int32_t *contig, const int64_t *start1,
                                 const int64_t *end1, const uint8_t *minus, int32_t flank, const int64_t *out_off,
                                 uint8_t *out, const int64_t *trunc_off, uint8_t *trunc_out) {
    if (!ctx || !ctx->d_bases || n < 0 || !out_off || !out) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    // sizes of the output pools = last offset + last length (recomputed on the host side from the rules)
    int64_t *len = (int64_t *)malloc(sizeof(int64_t) * n * 2);
    if (!len) return HITE_ENOMEM;
    int rc = hite_flank_sizes(ctx, n, contig, start1, end1, flank, len, len + n);
    if (rc) { free(len); return rc; }
    int64_t tot = 0, ttot = 0;
    for (int64_t i = 0; i < n; i++) {
        if (len[i] && out_off[i] + len[i] > tot) tot = out_off[i] + len[i];
        if (trunc_off && len[n + i] && trunc_off[i] + len[n + i] > ttot) ttot = trunc_off[i] + len[n + i];
    }
    free(len);
    DevBuf c, s, e, m, oo, o, to, t;
    HITE_CHECK(ctx, c.up(contig, n * 4));
    HITE_CHECK(ctx, s.up(start1, n * 8));
    HITE_CHECK(ctx, e.up(end1, n * 8));
    HITE_CHECK(ctx, m.up(minus, n));
    HITE_CHECK(ctx, oo.up(out_off, n * 8));
    HITE_CHECK(ctx, o.alloc(tot + 16));
    bool do_t = trunc_off && trunc_out && ttot > 0;
    if (do_t) {
        HITE_CHECK(ctx, to.up(trunc_off, n * 8));
        HITE_CHECK(ctx, t.alloc(ttot + 16));
    }
    rc = hite_flank_gather_dev(ctx, n, (int32_t *)c.p, (int64_t *)s.p, (int64_t *)e.p, (uint8_t *)m.p, flank,
                               (int64_t *)oo.p, (uint8_t *)o.p, do_t ? (int64_t *)to.p : nullptr,
                               do_t ? (uint8_t *)t.p : nullptr, nullptr);
    if (rc) return rc;
    HITE_CHECK(ctx, hipDeviceSynchronize());
    if (tot) HITE_CHECK(ctx, hipMemcpy(out, o.p, tot, hipMemcpyDeviceToHost));
    if (do_t) HITE_CHECK(ctx, hipMemcpy(trunc_out, t.p, ttot, hipMemcpyDeviceToHost));
    return HITE_OK;
}

// ---------------------------------------------------------------------------------------------
// HIP-event timing helpers (bench.py times kernels on the stream they are launched on)
// ---------------------------------------------------------------------------------------------
extern "C" int hite_event_create(void **ev) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return HITE_EHIP;
    *ev = (void *)e;
    return HITE_OK;
}
extern "C" int hite_event_record(void *ev, void *stream) {
    return hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) == hipSuccess ? HITE_OK : HITE_EHIP;
}
extern "C" int hite_event_elapsed_ms(void *a, void *b, float *ms) {
    if (hipEventSynchronize((hipEvent_t)b) != hipSuccess) return HITE_EHIP;
    return hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b) == hipSuccess ? HITE_OK : HITE_EHIP;
}
extern "C" int hite_event_destroy(void *ev) {
    return hipEventDestroy((hipEvent_t)ev) == hipSuccess ? HITE_OK : HITE_EHIP;
}

// ---------------------------------------------------------------------------------------------
// per-stage profiling with HIP events on the launch stream (bench.py reads these live)
// ---------------------------------------------------------------------------------------------
int hite_prof_begin(hite_ctx *ctx, const char *name, hipStream_t st) {
    if (!ctx || !ctx->prof_on) return -1;
    if (ctx->prof_pending >= 512) hite_prof_resolve(ctx);
    int sidx = -1;
    for (int i = 0; i < ctx->prof_n; i++) if (strcmp(ctx->prof_name[i], name) == 0) { sidx = i; break; }
    if (sidx < 0) {
        if (ctx->prof_n >= 64) return -1;
        sidx = ctx->prof_n++;
        strncpy(ctx->prof_name[sidx], name, 31);
        ctx->prof_name[sidx][31] = 0;
        ctx->prof_ms[sidx] = 0.0;
        ctx->prof_count[sidx] = 0;
    }
    int t = ctx->prof_pending;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess) return -1;
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return -1; }
    ctx->prof_ev[t][0] = (void *)a;
    ctx->prof_ev[t][1] = (void *)b;
    ctx->prof_stage[t] = sidx;
    ctx->prof_pending++;
    (void)hipEventRecord(a, st);
    return t;
}
void hite_prof_end(hite_ctx *ctx, int token, hipStream_t st) {
    if (!ctx || token < 0) return;
    (void)hipEventRecord((hipEvent_t)ctx->prof_ev[token][1], st);
}
void hite_prof_resolve(hite_ctx *ctx) {
    if (!ctx) return;
    for (int t = 0; t < ctx->prof_pending; t++) {
        hipEvent_t a = (hipEvent_t)ctx->prof_ev[t][0], b = (hipEvent_t)ctx->prof_ev[t][1];
        float ms = 0.f;
        if (hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&ms, a, b) == hipSuccess) {
            ctx->prof_ms[ctx->prof_stage[t]] += ms;
            ctx->prof_count[ctx->prof_stage[t]] += 1;
        }
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
    }
    ctx->prof_pending = 0;
}
extern "C" int hite_profile_enable(hite_ctx *ctx, int on) {
    if (!ctx) return HITE_EINVAL;
    hite_prof_resolve(ctx);
    ctx->prof_on = on;
    return HITE_OK;
}
extern "C" int hite_profile_reset(hite_ctx *ctx) {
    if (!ctx) return HITE_EINVAL;
    hite_prof_resolve(ctx);
    ctx->prof_n = 0;
    return HITE_OK;
}
extern "C" int hite_profile_count(hite_ctx *ctx) {
    if (!ctx) return 0;
    hite_prof_resolve(ctx);
    return ctx->prof_n;
}
extern "C" int hite_profile_get(hite_ctx *ctx, int idx, char *name_out /* >= 32 */, double *ms_total, int64_t *launches) {
    if (!ctx || idx < 0 || idx >= ctx->prof_n) return HITE_EINVAL;
    hite_prof_resolve(ctx);
    strncpy(name_out, ctx->prof_name[idx], 32);
    *ms_total = ctx->prof_ms[idx];
    *launches = ctx->prof_count[idx];
    return HITE_OK;
}

// utility for host mirrors / tests: copy a device range returned by a _dev entry point to the host
extern "C" int hite_memcpy_d2h(void *dst, const void *d_src, int64_t bytes) {
    if (bytes <= 0) return HITE_OK;
    if (hipDeviceSynchronize() != hipSuccess) return HITE_EHIP;
    return hipMemcpy(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost) == hipSuccess ? HITE_OK : HITE_EHIP;
}

// ---------------------------------------------------------------------------------------------
// merge of the call records between ranks: ONE all-gather on the caller's RCCL communicator (include/hite_gpu.h)
// ---------------------------------------------------------------------------------------------
#include <dlfcn.h>
typedef int (*hite_nccl_allgather_fn)(const void *, void *, size_t, int /* ncclDataType_t */, void * /* ncclComm_t */, hipStream_t);
extern "C" int hite_allgather_records(hite_ctx *ctx, void *nccl_comm, const void *d_send, void *d_recv, int64_t bytes_per_rank, void *stream) {
    if (!ctx || !nccl_comm || !d_send || !d_recv || bytes_per_rank < 0) return HITE_EINVAL;
    if (bytes_per_rank == 0) return HITE_OK;
    static hite_nccl_allgather_fn fn = nullptr;
    if (!fn) {
        // the RCCL the process already carries -- the communicator came out of it; this library loads nothing by itself
        fn = (hite_nccl_allgather_fn)dlsym(RTLD_DEFAULT, "ncclAllGather");
        if (!fn) { snprintf(ctx->err, sizeof(ctx->err), "hite_allgather_records: no RCCL (ncclAllGather) in this process"); return HITE_ENODEV; }
    }
    const int rc = fn(d_send, d_recv, (size_t)bytes_per_rank, 0 /* ncclInt8 */, nccl_comm, (hipStream_t)stream);
    if (rc != 0) { snprintf(ctx->err, sizeof(ctx->err), "hite_allgather_records: ncclAllGather returned %d", rc); return HITE_EHIP; }
    return HITE_OK;
}
