// hite_copies.hip -- candidate x genome copy finding: this build's GPU-native stage where the reference
// shells out to `minimap2 -ax map-ont -N 300 -p 0.2` and filters the SAM records
// (get_full_length_copies_minimap2 / get_copies_minimap2, /root/reference/module/Util.py:7933-8030).
// minimap2 is third-party and absent: parity is pinned against the build's own CPU twin
// (oracle/hite_oracle_copies.c, whose header holds the definition), record for record.
//
// Pipeline (sort / scan / segment, everything resident in HBM):
//   index (once per genome): (w=10, k=15) minimizers straight from the 2-bit genome, a workgroup per tile of 2048
//     window starts, emitted IN POSITION ORDER (staging region per tile, scan of the tile counts, pack), ONE stable
//     4-pass radix sort on hash|strand carrying every entry's rank along the genome, 2^26-bucket directory written
//     from the sorted hashes (round 5; it was an append in arbitrary order + a 7-pass sort on (hash, position));
//   query: candidate minimizers (wave per candidate, LDS tile of 64 windows, private regions + pack) -> directory
//     lookup -> occurrence counts -> scan -> wave-cooperative hit expansion (key = candidate | relative strand |
//     diagonal) -> radix sort (10-bit digits) -> cluster flags where the diagonal jumps -> extreme anchors per
//     cluster by a wave-level segmented min / max scan -> coverage filter + boundary extrapolation (two passes,
//     per-candidate counters) -> radix sort by (candidate | anchors desc | start) -> top 300.
//   The second half of the file is the all-vs-all seeding of stage 3.1 on the same index (hite_seed_allvsall).
// Bound: HBM streaming for the sorts (12 B in + 12 B out per element per pass) and L2-latency for the
// directory lookups; integer work only.
#include "hite_common.h"
#include "hite_scan.h"
#include "hite_sort.h"
#include "hite_arena.h"

#define CK 15
#define CW 10
#define C_MAXOCC 2000
#define C_TD 64
#define C_MINANCH 3
#define C_MAXCOPY 300
#define HS_INVALID 0xffffffffu
#define C_SUB_EDGE 256     // candidate minimizers: all within this many bases of either end are looked up,
#define C_SUB_UNIT 1024    // of the interior every S-th (by hash), S = min(C_SUB_MAX, length / C_SUB_UNIT)
#define C_SUB_MAX 4
#define DIRBITS 26
#define DBIAS 65536ll

struct CopyState {
    unsigned *idx_hs = nullptr, *idx_pos = nullptr, *dir = nullptr;
    unsigned *idx_t = nullptr;   // rank of every index entry among the minimizers in position order (stage 3.1 walks the seeds in that order)
    const unsigned long long *idx_key = nullptr;   // the sorted (hash | strand) << 32 | position words themselves (in the build arena: valid until the next build)
    int64_t M = 0;
    bool restricted = false;     // the index holds only the entries one candidate set looks up (hite_find_copies_restricted_dev): any other use rebuilds it
    int64_t idx_cap = 0;         // entries the index arrays hold (grow-only: rebuilding on the same handle allocates nothing)
    Arena build;      // temporaries of the index build
    Arena arena;      // temporaries of one call
    Arena out;        // copy table handed to the caller (valid until the next call)
    int64_t *h_pin = nullptr;
    int64_t *d_scal = nullptr;
    const uint32_t *out_clip = nullptr;   // per copy record of the last call (in `out`): candidate bases clipped left | right << 16 (hite_copy_clips_dev)
    int64_t out_n = 0;
    // the minimizer tiles of the last index build (still in the build arena): valid for THIS genome state (hite_ctx::genome_epoch) up to
    // entry kept_log of its mask log; a build on the same state redoes only the tiles masked since (index_build_impl)
    const void *kept_stage = nullptr, *kept_cnt = nullptr, *kept_ctx = nullptr;
    int64_t kept_epoch = -1, kept_log = 0, kept_G = -1;
    int64_t last[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // last call: candidate minimizers, hits, clusters, copies (before the 300 cap); chains with a long / short end to extend, extension columns
};

__device__ __forceinline__ unsigned lowbias32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned rev2_30(unsigned x) {  // reverse the order of the 15 2-bit groups
    unsigned y = __brev(x) >> 2;
    return ((y & 0x15555555u) << 1) | ((y >> 1) & 0x15555555u);
}
__device__ __forceinline__ unsigned hs_from_code(unsigned x) {
    unsigned rc = rev2_30(x ^ 0x3fffffffu);
    unsigned can = x < rc ? x : rc;
    unsigned hs = (lowbias32(can) & ~1u) | (rc < x ? 1u : 0u);
    if (hs >= 0xfffffffeu) hs -= 2;
    return hs;
}
__device__ __forceinline__ int contig_of(const int64_t *__restrict__ coff, int nc, int64_t g) {
    int lo = 0, hi = nc;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (coff[mid] <= g) lo = mid; else hi = mid; }
    return lo;
}
// hs of the genome k-mer starting at global position g (contig [cb, ce))
__device__ __forceinline__ unsigned genome_hs(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask, int64_t g,
                                              int64_t ce) {
    if (g + CK > ce) return HS_INVALID;
    int64_t w = g >> 4; int sh = (int)(g & 15) * 2;
    unsigned long long two = (unsigned long long)bases[w] | ((unsigned long long)bases[w + 1] << 32);
    unsigned x = (unsigned)(two >> sh) & 0x3fffffffu;
    int64_t mw = g >> 5; int msh = (int)(g & 31);
    unsigned long long mt = (unsigned long long)nmask[mw] | ((unsigned long long)nmask[mw + 1] << 32);
    if ((unsigned)(mt >> msh) & 0x7fffu) return HS_INVALID;
    return hs_from_code(x);
}
// minimizer of the window starting at k-mer start p of a sequence whose k-mer starts are [b, b+nk):
// index of the valid k-mer with the smallest (hs >> 1, position) in [p, min(p+W, b+nk)), -1 if none
template <typename HsFn>
__device__ __forceinline__ int64_t window_min(int64_t p, int64_t b, int64_t nk, HsFn hs_at, unsigned *hs_out) {
    int64_t hi = p + CW < b + nk ? p + CW : b + nk;
    int64_t best = -1; unsigned bh = 0;
    for (int64_t i = p; i < hi; i++) {
        unsigned h = hs_at(i);
        if (h == HS_INVALID) continue;
        if (best < 0 || (h >> 1) < (bh >> 1)) { best = i; bh = h; }
    }
    *hs_out = bh;
    return best;
}

// wave-aggregated append: one atomic per wavefront instead of one per lane (a single hot counter serialises in L2)
__device__ __forceinline__ unsigned long long wave_append(bool want, unsigned long long *counter) {
    const unsigned long long m = __ballot(want);
    if (m == 0) return 0;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(m));
    base = ((unsigned long long)(unsigned)__shfl((int)(base >> 32), leader) << 32) | (unsigned)__shfl((int)base, leader);
    return base + __popcll(m & ((1ull << lane) - 1ull));
}

// set of 31-bit hashes (hs >> 1) in an open-addressing table of 2^k words (empty = 0xffffffff, never a hash): the candidate side of a
// RESTRICTED index build -- the genome pass keeps only the minimizers whose hash some candidate minimizer looks up
struct HSet {
    const unsigned *tab;     // open addressing, 2^k words
    const unsigned *bits;    // one-hash Bloom filter in front of it, 2^b bits (~16 per member: small enough to stay in L2, it answers ~15/16 of the misses)
    unsigned mask, bmask;
};
__device__ __forceinline__ unsigned hset_mix(unsigned h31) { return lowbias32(h31 ^ 0x9e3779b9u); }
__device__ __forceinline__ unsigned hset_bit(const HSet &H, unsigned h31) { return (hset_mix(h31) >> 7) & H.bmask; }
__device__ __forceinline__ bool hset_probe(const HSet &H, unsigned h31) {
    unsigned slot = hset_mix(h31) & H.mask;
    for (;;) {
        const unsigned v = H.tab[slot];
        if (v == h31) return true;
        if (v == 0xffffffffu) return false;
        slot = (slot + 1) & H.mask;
    }
}
__global__ void hset_insert_kernel(int ncand, const int64_t *__restrict__ cand_off, const unsigned *__restrict__ r_hs,
                                   const int32_t *__restrict__ q_cnt, unsigned *__restrict__ tab, unsigned mask,
                                   unsigned *__restrict__ bits, unsigned bmask) {
    const int lane = threadIdx.x & 63;
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < ncand; c += gridDim.x * 4) {
        const int64_t src = cand_off[c];
        const int cnt = q_cnt[c];
        for (int i = lane; i < cnt; i += 64) {
            const unsigned h31 = r_hs[src + i] >> 1;
            const unsigned x = hset_mix(h31);
            const unsigned b = (x >> 7) & bmask;
            atomicOr(bits + (b >> 5), 1u << (b & 31));
            unsigned slot = x & mask;
            for (;;) {
                const unsigned old = atomicCAS(tab + slot, 0xffffffffu, h31);
                if (old == 0xffffffffu || old == h31) break;
                slot = (slot + 1) & mask;
            }
        }
    }
}

// genome minimizers IN POSITION ORDER.  key = hs << 32 | pos, value = rank of the minimizer along the genome (its slot).
// One block per tile of GM_TILE window starts: the k-mer hashes the tile needs are computed once into LDS (each from its
// own contig: a k-mer that crosses the contig end or touches an N is invalid), every thread then scans the windows of
// GM_TILE / 256 starts (round j = 256 consecutive starts) and their predecessors.  A window emits its minimizer when it differs
// from the window before, so along the window starts the emitted positions increase strictly: a tile writes its records --
// rounds in order, lanes in order inside a round -- to ITS OWN region of a staging array (GM_TILE slots: a window start emits
// at most one record) and its count; after a scan of the counts genome_minimizer_pack_kernel moves the regions to their place.
// The array is then sorted by position -- the index is ONE stable sort on the hash (4 passes instead of the 7 of (hash,
// position)), and the rank of every seed in position order, which stage 3.1 needs, comes with it as the sort's value (it
// used to be a second 4-pass sort).  No append counter (the single atomic per tile on one was the arbitrary order; a chained
// scan over the tiles inside the kernel measured 2x the kernel's time: its waiting tiles hold the CUs).
#define GM_TILE 2048
template <bool RS /* restricted to the hashes in hset */>
__global__ void __launch_bounds__(256) genome_minimizer_kernel(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask,
                                                               const int64_t *__restrict__ coff, int nc, int64_t G, int64_t ntiles,
                                                               unsigned long long *__restrict__ stage /* [ntiles][GM_TILE] */,
                                                               int32_t *__restrict__ tile_cnt,
                                                               HSet hset /* tab null: every minimizer */,
                                                               const int32_t *__restrict__ tile_list = nullptr /* only these tiles (ntiles = their number) */) {
    __shared__ unsigned sh[GM_TILE + CW + 8];   // sh[q] = hs of the k-mer starting at p0 - 1 + q
    __shared__ short wm[GM_TILE + 8];           // wm[q] = index into sh of the minimizer of the window starting at p0 - 1 + q (-1: none)
    constexpr int PER = GM_TILE / 256;
    __shared__ int s_cnt[PER][4];               // emitted per round and wavefront
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t it = blockIdx.x; it < ntiles; it += gridDim.x) {
        const int64_t tile = tile_list ? (int64_t)tile_list[it] : it;
        const int64_t p0 = tile * GM_TILE;
        // a tile and its halo nearly always lie inside ONE contig: its bounds are looked up once (every thread, the same
        // cached words) instead of once per k-mer and once per window
        const int64_t g_lo = p0 > 0 ? p0 - 1 : 0;
        const int c0 = contig_of(coff, nc, g_lo);
        const int64_t cb0 = coff[c0], ce0 = coff[c0 + 1];
        const bool one = p0 + GM_TILE + CW + CK <= ce0;
        __syncthreads();
        for (int q = threadIdx.x; q < GM_TILE + CW + 1; q += 256) {
            const int64_t g = p0 - 1 + q;
            unsigned h = HS_INVALID;
            if (g >= 0 && g < G) h = genome_hs(bases, nmask, g, one ? ce0 : coff[contig_of(coff, nc, g) + 1]);
            sh[q] = h;
        }
        __syncthreads();
        // the minimizer of every window that starts in the tile or one base before it (a window's predecessor decides whether it emits)
        for (int q = threadIdx.x; q <= GM_TILE; q += 256) {
            const int64_t g = p0 - 1 + q;
            int m = -1;
            if (g >= 0 && g < G) {
                int64_t cb = cb0, ce = ce0;
                if (!one) { const int c = contig_of(coff, nc, g); cb = coff[c]; ce = coff[c + 1]; }
                const int64_t nk = ce - cb - CK + 1;
                const int64_t nwin = nk >= CW ? nk - CW + 1 : 1;
                if (nk > 0 && g - cb < nwin) {
                    const int lim = (int)((cb + nk - g) < CW ? (cb + nk - g) : CW);          // k-mers [g, min(g + W, cb + nk))
                    unsigned h = 0;
#pragma unroll
                    for (int i = 0; i < CW; i++) {
                        const unsigned v = sh[q + i];
                        if (i < lim && v != HS_INVALID && (m < 0 || (v >> 1) < (h >> 1))) { m = q + i; h = v; }
                    }
                }
            }
            wm[q] = (short)m;
        }
        __syncthreads();
        unsigned hh[PER]; unsigned mm[PER];
        unsigned wantbits = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int o = j * 256 + threadIdx.x;          // window start p = p0 + o: wm[o + 1]; its predecessor: wm[o]
            const int64_t p = p0 + o;
            bool want = false; unsigned h = 0; int m = -1;
            if (p < G) {
                m = wm[o + 1];
                if (m >= 0) {
                    h = sh[m];
                    const int64_t cb = one ? cb0 : coff[contig_of(coff, nc, p)];
                    want = !(p > cb && wm[o] == m);       // (p > cb: the window before lies in the same contig; -1 never equals m)
                }
            }
            hh[j] = h; mm[j] = (unsigned)(p0 - 1 + m);
            if (want) wantbits |= 1u << j;
            if constexpr (!RS) {
                const unsigned long long bal = __ballot(want);
                if (lane == 0) s_cnt[j][w] = __popcll(bal);
            }
        }
        if constexpr (RS) {      // restricted index: only the hashes in the set.  The filter words of all rounds are fetched before any is tested
            unsigned fw[PER];
#pragma unroll
            for (int j = 0; j < PER; j++) fw[j] = ((wantbits >> j) & 1u) ? hset.bits[hset_bit(hset, hh[j] >> 1) >> 5] : 0u;
#pragma unroll
            for (int j = 0; j < PER; j++)
                if (((wantbits >> j) & 1u) && !(((fw[j] >> (hset_bit(hset, hh[j] >> 1) & 31)) & 1u) && hset_probe(hset, hh[j] >> 1))) wantbits &= ~(1u << j);
#pragma unroll
            for (int j = 0; j < PER; j++) {
                const unsigned long long bal = __ballot((wantbits >> j) & 1u);
                if (lane == 0) s_cnt[j][w] = __popcll(bal);
            }
        }
        __syncthreads();
        unsigned long long *reg = stage + tile * GM_TILE;
        int slot0 = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const bool want = (wantbits >> j) & 1u;
            const unsigned long long bal = __ballot(want);
            int slot = slot0 + __popcll(bal & ((1ull << lane) - 1ull));
            for (int q = 0; q < w; q++) slot += s_cnt[j][q];
            if (want) reg[slot] = ((unsigned long long)hh[j] << 32) | (unsigned long long)mm[j];
            slot0 += s_cnt[j][0] + s_cnt[j][1] + s_cnt[j][2] + s_cnt[j][3];
        }
        if (threadIdx.x == 0) tile_cnt[tile] = slot0;
    }
}
// the tiles' regions to their place: record r of tile t -> slot first[t] + r (wavefront per tile; a tile holds ~370 records)
__global__ void __launch_bounds__(256) genome_minimizer_pack_kernel(int64_t ntiles, const unsigned long long *__restrict__ stage,
                                                                    const int64_t *__restrict__ first, unsigned long long *__restrict__ out,
                                                                    unsigned *__restrict__ out_rank, unsigned long long cap) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int64_t a = first[tile];
    const int n = (int)(first[tile + 1] - a);
    const unsigned long long *reg = stage + tile * GM_TILE;
    for (int r = lane; r < n; r += 64) {
        const unsigned long long slot = (unsigned long long)a + (unsigned long long)r;
        if (slot < cap) { out[slot] = reg[r]; out_rank[slot] = (unsigned)slot; }
    }
}

// RESTRICTED index from kept tiles: the tiles hold every minimizer (so that the full index behind the same genome can use them again); a
// wavefront per tile counts / moves the records whose hash is in the set (Bloom word first, then the table), in order
template <bool WRITE>
__global__ void __launch_bounds__(256) genome_minimizer_filter_kernel(int64_t ntiles, const unsigned long long *__restrict__ stage,
                                                                      const int32_t *__restrict__ tile_cnt, HSet hset,
                                                                      int32_t *__restrict__ cnt_out, const int64_t *__restrict__ first,
                                                                      unsigned long long *__restrict__ out, unsigned *__restrict__ out_rank,
                                                                      unsigned long long cap,
                                                                      unsigned long long *__restrict__ wmask /* [ntiles][GM_TILE / 64]: pass COUNT's ballots, which pass WRITE reads instead of probing again */) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int n = tile_cnt[tile];
    const unsigned long long *reg = stage + tile * GM_TILE;
    unsigned long long at = WRITE ? (unsigned long long)first[tile] : 0ull;
    if (WRITE && first[tile + 1] == first[tile]) return;          // (nothing of this tile is in the set)
    int c = 0;
    for (int r0 = 0; r0 < n; r0 += 64) {
        const int r = r0 + lane;
        unsigned long long rec = 0ull;
        bool want = false;
        unsigned long long bal;
        if (WRITE) {
            bal = wmask[tile * (GM_TILE / 64) + (r0 >> 6)];
            if (bal == 0ull) continue;
            want = (bal >> lane) & 1ull;
            if (want) rec = reg[r];
        } else {
            if (r < n) {
                rec = reg[r];
                const unsigned h31 = (unsigned)(rec >> 33);
                const unsigned b = hset_bit(hset, h31);
                want = ((hset.bits[b >> 5] >> (b & 31)) & 1u) && hset_probe(hset, h31);
            }
            bal = __ballot(want);
            if (lane == 0) wmask[tile * (GM_TILE / 64) + (r0 >> 6)] = bal;
        }
        if (WRITE) {
            const unsigned long long slot = at + (unsigned long long)__popcll(bal & ((1ull << lane) - 1ull));
            if (want && slot < cap) { out[slot] = rec; out_rank[slot] = (unsigned)slot; }
            at += (unsigned long long)__popcll(bal);
        } else c += __popcll(bal);
    }
    if (!WRITE && lane == 0) cnt_out[tile] = c;
}

// the sorted keys apart: hash | strand and position of every index entry
__global__ void split_index_kernel(int64_t M, const unsigned long long *__restrict__ keys, unsigned *__restrict__ hs,
                                   unsigned *__restrict__ pos) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const unsigned long long k = keys[i];
    hs[i] = (unsigned)(k >> 32); pos[i] = (unsigned)k;
}
// directory over the top DIRBITS of the hash: dir[b] = first index entry of bucket b or a later one (dir[2^DIRBITS] = M).  The entries
// are sorted, so entry i starts every bucket after its predecessor's up to its own (the buckets behind the last entry keep the
// M the directory is pre-filled with); most buckets hold a few entries: one short run of stores per entry, no counters, no scan.
__global__ void __launch_bounds__(256) index_directory_kernel(int64_t M, const unsigned *__restrict__ hs, unsigned *__restrict__ dir) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int a = 0, b = 0;       // this entry starts the buckets (a, b]
    if (i < M) {
        b = (int)(hs[i] >> (32 - DIRBITS));
        a = i > 0 ? (int)(hs[i - 1] >> (32 - DIRBITS)) : -1;
    }
    if (b - a <= 8) for (int x = a + 1; x <= b; x++) dir[x] = (unsigned)i;
    // a long stretch of empty buckets (small genomes) is filled by the whole wavefront
    unsigned long long big = __ballot(b - a > 8);
    while (big) {
        const int l = __ffsll((long long)big) - 1;
        big &= big - 1ull;
        const int aa = __shfl(a, l, 64), bb = __shfl(b, l, 64);
        const unsigned ii = (unsigned)(i - lane + l);
        for (int x = aa + 1 + lane; x <= bb; x += 64) dir[x] = ii;
    }
}
__global__ void fill_u32_kernel(int64_t n, unsigned *__restrict__ p, unsigned v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void i64_to_u32_kernel(int64_t n, const int64_t *__restrict__ in, unsigned *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (unsigned)in[i];
}

// candidate minimizers: one block per candidate, CM_TILE window starts per tile.  The tile's bases are staged in LDS once
// (coalesced) as 2-bit codes, every thread rolls the hashes of five consecutive k-mers out of them (19 byte reads for five
// 15-mers), every window then scans its own k-mers and the previous window's from LDS (11 word reads), and the minimizers
// leave in position order (rounds of 256 consecutive starts, ballots + wave counts).  Nothing but the minimizers goes to HBM.
// The records of candidate c first go to a private region (start = its byte offset: a candidate of L bases has < L
// windows), with the count; after an exclusive scan of the counts a second kernel packs the regions.  Order = (candidate,
// position), no atomics (a single append counter would serialise ~3 M device-scope atomics at ~12 ns each).
#define CM_TILE 1024
#define CM_KPT 5       // k-mers per thread in the hash phase (256 x 5 >= CM_TILE + CW + 1)
__global__ void __launch_bounds__(256) cand_minimizer_kernel(int ncand, const uint8_t *__restrict__ cand,
                                                             const int64_t *__restrict__ cand_off,
                                                             unsigned *__restrict__ r_pos, unsigned *__restrict__ r_hs,
                                                             int32_t *__restrict__ q_cnt, unsigned long long *__restrict__ max_len) {
    __shared__ uint8_t sb[256 * CM_KPT + CK + 8];     // sb[q] = code of base (tile - 1 + q): 0..3, 4 = not A/C/G/T or outside
    __shared__ unsigned sh[256 * CM_KPT];             // sh[q] = hs of the k-mer starting at base (tile - 1 + q)
    __shared__ int sm[CM_TILE + 1];                   // sm[o + 1] = minimizer position of the window that starts at tile + o
    __shared__ int s_cnt[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c = blockIdx.x; c < ncand; c += gridDim.x) {
        const int64_t cb = cand_off[c];
        const int L = (int)(cand_off[c + 1] - cb);
        const int nk = L - CK + 1;
        if (threadIdx.x == 0 && L > 0 && (unsigned long long)L > __atomic_load_n(max_len, __ATOMIC_RELAXED)) atomicMax(max_len, (unsigned long long)L);   // (a maximum only grows: one same-address atomic per candidate was what 50 000 blocks queued behind)
        if (nk <= 0) { if (threadIdx.x == 0) q_cnt[c] = 0; continue; }
        const int nwin = nk >= CW ? nk - CW + 1 : 1;
        const uint8_t *s = cand + cb;
        int run = 0;   // block-uniform count so far
        for (int base = 0; base < nwin; base += CM_TILE) {
            __syncthreads();
            for (int q = threadIdx.x; q < 256 * CM_KPT + CK; q += 256) {
                const int pos = base - 1 + q;
                const uint8_t ch = (pos >= 0 && pos < L) ? s[pos] : 0;
                sb[q] = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
            }
            __syncthreads();
            {
                const int q0 = threadIdx.x * CM_KPT;
                unsigned x = 0, bad = 0;
#pragma unroll
                for (int i = 0; i < CK; i++) { const unsigned cd = sb[q0 + i]; x |= (cd & 3u) << (2 * i); bad |= (cd >> 2) << i; }
#pragma unroll
                for (int j = 0; j < CM_KPT; j++) {
                    sh[q0 + j] = bad ? HS_INVALID : hs_from_code(x);
                    const unsigned cd = sb[q0 + j + CK];
                    x = (x >> 2) | ((cd & 3u) << (2 * (CK - 1)));
                    bad = (bad >> 1) | ((cd >> 2) << (CK - 1));
                }
            }
            __syncthreads();
            // every window's minimizer ONCE (round 6, as genome_minimizer_kernel since round 5: a start used to scan its own window and
            // its predecessor's -- 20 LDS reads, now 10 + 1): sm[o + 1] = minimizer of the window that starts at base + o, sm[0] = of the
            // window in front of the tile
            const int rounds = (nwin - base + 255) / 256 < CM_TILE / 256 ? (nwin - base + 255) / 256 : CM_TILE / 256;    // block-uniform
            int mm[CM_TILE / 256];
            unsigned hh[CM_TILE / 256];
#pragma unroll
            for (int j = 0; j < CM_TILE / 256; j++) {
                mm[j] = -1; hh[j] = 0;
                if (j < rounds) {
                    const int o = j * 256 + threadIdx.x;
                    const int lp = base + o;
                    if (lp < nwin) {
                        int m = -1; unsigned h = 0;
#pragma unroll
                        for (int i = 0; i < CW; i++) {
                            const unsigned v = sh[o + 1 + i];
                            if (v != HS_INVALID && (m < 0 || (v >> 1) < (h >> 1))) { m = lp + i; h = v; }
                        }
                        mm[j] = m; hh[j] = h;
                    }
                    sm[o + 1] = mm[j];
                }
            }
            if (threadIdx.x == 0) {
                int m = -1; unsigned h = 0;
                if (base > 0)
                    for (int i = 0; i < CW; i++) {
                        const unsigned u = sh[i];
                        if (u != HS_INVALID && (m < 0 || (u >> 1) < (h >> 1))) { m = base - 1 + i; h = u; }
                    }
                sm[0] = m;
            }
            __syncthreads();
#pragma unroll 1
            for (int j = 0; j < rounds; j++) {
                const int o = j * 256 + threadIdx.x;
                const int lp = base + o;
                bool want = false;
                const unsigned h = hh[j]; const int m = mm[j];
                if (lp < nwin) {
                    want = m >= 0 && !(lp > 0 && sm[o] == m);
                    // sampling of the candidate's minimizers (definition: cand_minimizer_kept of the twin)
                    if (want && L >= 2 * C_SUB_UNIT && m >= C_SUB_EDGE && m + CK <= L - C_SUB_EDGE) {
                        const unsigned S = L / C_SUB_UNIT > C_SUB_MAX ? C_SUB_MAX : L / C_SUB_UNIT;
                        want = ((h >> 1) % S) == 0u;
                    }
                }
                const unsigned long long bm = __ballot(want);
                if (lane == 0) s_cnt[w] = __popcll(bm);
                __syncthreads();
                int off = run + __popcll(bm & ((1ull << lane) - 1ull));
                for (int q = 0; q < w; q++) off += s_cnt[q];
                if (want) { r_pos[cb + off] = (unsigned)m; r_hs[cb + off] = h; }
                run += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
                __syncthreads();
            }
        }
        if (threadIdx.x == 0) q_cnt[c] = run;
    }
}
__global__ void __launch_bounds__(256) cand_minimizer_pack_kernel(int ncand, const int64_t *__restrict__ cand_off,
                                                                  const unsigned *__restrict__ r_pos, const unsigned *__restrict__ r_hs,
                                                                  const int64_t *__restrict__ q_first, unsigned *__restrict__ q_c,
                                                                  unsigned *__restrict__ q_pos, unsigned *__restrict__ q_hs) {
    const int lane = threadIdx.x & 63;
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < ncand; c += gridDim.x * 4) {
        const int64_t src = cand_off[c], dst = q_first[c];
        const int cnt = (int)(q_first[c + 1] - dst);
        for (int i = lane; i < cnt; i += 64) { q_c[dst + i] = (unsigned)c; q_pos[dst + i] = r_pos[src + i]; q_hs[dst + i] = r_hs[src + i]; }
    }
}

// occurrences of each candidate minimizer in the index (same hs >> 1)
__global__ void occ_kernel(int64_t nq, const unsigned *__restrict__ q_hs, const unsigned *__restrict__ idx_hs,
                           const unsigned *__restrict__ dir, unsigned *__restrict__ occ_lo, int32_t *__restrict__ occ_n) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq) return;
    unsigned h31 = q_hs[t] >> 1;
    unsigned b = q_hs[t] >> (32 - DIRBITS);
    unsigned lo = dir[b], hi = dir[b + 1];
    while (lo < hi) { unsigned mid = (lo + hi) >> 1; if ((idx_hs[mid] >> 1) < h31) lo = mid + 1; else hi = mid; }
    unsigned e = lo, end = dir[b + 1];
    while (e < end && (idx_hs[e] >> 1) == h31) e++;
    int n = (int)(e - lo);
    occ_lo[t] = lo;
    occ_n[t] = n > C_MAXOCC ? 0 : n;
}

// Hit records.  Wide form: key = candidate << 34 | rel << 33 | (d + DBIAS), value = qo (12 bytes).  Packed form (whenever the
// fields fit 64 bits: qbits for qo < longest candidate, dbits for d + DBIAS < n_bases + DBIAS, 1, bits of the candidate count):
// key = candidate | rel | d + DBIAS | qo from the top down, no value array -- the radix passes move 8 instead of 12 bytes per
// element and skip the qo bits (the order of equal diagonals stays the order of generation, as with the wide form).
struct HitFmt {
    int qbits, dbits;           // packed: field widths (qbits = 0: wide form)
    const unsigned *hval;       // wide form: qo
};
__device__ __forceinline__ unsigned hit_cand(const HitFmt &F, unsigned long long k) { return (unsigned)(k >> (F.qbits ? F.qbits + F.dbits + 1 : 34)); }
__device__ __forceinline__ unsigned long long hit_strand_key(const HitFmt &F, unsigned long long k) { return k >> (F.qbits ? F.qbits + F.dbits : 33); }   // candidate | rel
__device__ __forceinline__ unsigned hit_rel(const HitFmt &F, unsigned long long k) { return (unsigned)hit_strand_key(F, k) & 1u; }
__device__ __forceinline__ long long hit_dbias(const HitFmt &F, unsigned long long k) {      // d + DBIAS
    return F.qbits ? (long long)((k >> F.qbits) & ((1ull << F.dbits) - 1ull)) : (long long)(k & 0x1ffffffffull);
}
__device__ __forceinline__ unsigned hit_qo(const HitFmt &F, unsigned long long k, int64_t i) { return F.qbits ? (unsigned)(k & ((1ull << F.qbits) - 1ull)) : F.hval[i]; }
//
// One wavefront expands 64 consecutive candidate minimizers: their hits are a contiguous output range (hit_off is the
// exclusive scan of the counts), lanes walk that range 64 at a time and find the owning minimizer by a 6-step search
// over the lanes' prefix sums (shuffles): balanced work and coalesced 12-byte stores whatever the occurrence counts are.
__global__ void __launch_bounds__(256) hit_kernel(int64_t nq, const unsigned *__restrict__ q_c, const unsigned *__restrict__ q_pos,
                                                  const unsigned *__restrict__ q_hs, const int64_t *__restrict__ cand_off,
                                                  const unsigned *__restrict__ idx_hs, const unsigned *__restrict__ idx_pos,
                                                  const unsigned *__restrict__ occ_lo, const int32_t *__restrict__ occ_n,
                                                  const int64_t *__restrict__ hit_off, unsigned long long *__restrict__ hkey,
                                                  unsigned *__restrict__ hval, HitFmt F) {
    const int lane = threadIdx.x & 63;
    const int64_t t0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (t0 >= nq) return;
    const int64_t t = t0 + lane;
    const bool have = t < nq;
    const int64_t o0 = hit_off[t0];
    const int pre = have ? (int)(hit_off[t] - o0) : 0x7fffffff;           // first output of this lane's minimizer
    const int64_t tend = t0 + 64 < nq ? t0 + 64 : nq;
    const int T = (int)(hit_off[tend] - o0);
    const unsigned c = have ? q_c[t] : 0u, qh = have ? q_hs[t] : 0u, lo = have ? occ_lo[t] : 0u;
    const int qp = have ? (int)q_pos[t] : 0;
    const int Lq = have ? (int)(cand_off[c + 1] - cand_off[c]) : 0;
    for (int j0 = 0; j0 < T; j0 += 64) {   // wave-uniform trip count: the shuffles below read lanes that own no output themselves
        const int j = j0 + lane;
        // largest u with pre[u] <= j (minimizers without hits share their successor's prefix and are skipped by "largest")
        int u = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            const int v = u + step;
            const int pv = __shfl(pre, v < 64 ? v : 63);
            if (v < 64 && pv <= j) u = v;
        }
        const int i = j - __shfl(pre, u);
        const unsigned uc = (unsigned)__shfl((int)c, u), uqh = (unsigned)__shfl((int)qh, u), ulo = (unsigned)__shfl((int)lo, u);
        const long long uqp = __shfl(qp, u), uL = __shfl(Lq, u);
        if (j >= T) continue;
        const unsigned gh = idx_hs[ulo + i];
        const long long gpos = idx_pos[ulo + i];
        const unsigned rel = (uqh ^ gh) & 1u;
        const long long qo = rel ? (uL - uqp - CK) : uqp;
        const long long d = gpos - qo + DBIAS;
        if (F.qbits) hkey[o0 + j] = ((((unsigned long long)uc << 1 | rel) << F.dbits | (unsigned long long)d) << F.qbits) | (unsigned long long)qo;
        else {
            hkey[o0 + j] = ((unsigned long long)uc << 34) | ((unsigned long long)rel << 33) | (unsigned long long)d;
            hval[o0 + j] = (unsigned)qo;
        }
    }
}

// (diagnostic, HITE_HIT_HIST=1) hits per candidate: candidates and hits in the classes <= 1024, <= 2048, <= 4096, <= 8192, above
__global__ void hit_hist_kernel(int n_cand, const int64_t *__restrict__ q_first, const int64_t *__restrict__ hit_off,
                                unsigned long long *__restrict__ out /* [2][5] + max */) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cand) return;
    const long long k = hit_off[q_first[c + 1]] - hit_off[q_first[c]];
    const int cls = k <= 1024 ? 0 : k <= 2048 ? 1 : k <= 4096 ? 2 : k <= 8192 ? 3 : 4;
    atomicAdd(&out[cls], 1ull);
    atomicAdd(&out[5 + cls], (unsigned long long)k);
    atomicMax(&out[10], (unsigned long long)k);
}
// ---------------------------------------------------------------------------------------------
// per-candidate sort of the hits.  The hits of a candidate are generated contiguously (hit_off[q_first[c]] ..), so the order
// (candidate, strand, diagonal) only has to be established INSIDE every candidate's range: a workgroup loads the range into
// LDS, sorts it there on the (strand | diagonal) field by a stable LSD radix sort (4-bit digits, two LDS buffers) and writes
// it back in place -- one global read and one write of 8 bytes per hit where the global radix sort of round 3 made five
// passes (5 x 16 B + histograms).  Four classes by hits per candidate (C3: <= 2048: 75 % of the candidates, 51 % of the hits;
// <= 4096: 19 % / 31 %; <= 8192: 5 % / 14 %; above: 0.6 % / 3.6 %, profiles of round 3): 2 x 16, 2 x 32, 2 x 64 KB of LDS (the
// last one holds a CU by itself), and global ping-pong buffers (the range against the sorter's spare array) for the rest.  Stable, like the global passes it replaces: the table is
// the same, hit for hit.  (Packed 8-byte hits only; the 12-byte form keeps the global sort.)
// ---------------------------------------------------------------------------------------------
#define HS_SMALL 2048
#define HS_MID 4096
#define HS_MEDIUM 8192
#define HS_NCLS 4
typedef __attribute__((address_space(3))) unsigned long long *hs_lptr;
__global__ void hit_sort_classify_kernel(int n_cand, const int64_t *__restrict__ q_first, const int64_t *__restrict__ hit_off,
                                         unsigned *__restrict__ counts /* HS_NCLS */, int32_t *__restrict__ lists /* HS_NCLS x n_cand */) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    long long k = 0;
    if (c < n_cand) k = hit_off[q_first[c + 1]] - hit_off[q_first[c]];
    const int cls = k <= 1 ? -1 : (k <= HS_SMALL ? 0 : (k <= HS_MID ? 1 : (k <= HS_MEDIUM ? 2 : 3)));
    // one atomic per wavefront and class (50 000 atomics on three addresses cost 0.5 ms)
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < HS_NCLS; q++) {
        const unsigned long long m = __ballot(cls == q);
        if (m == 0ull) continue;
        unsigned base = 0u;
        if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&counts[q], (unsigned)__popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1, 64);
        if (cls == q) lists[(size_t)q * n_cand + base + __popcll(m & ((1ull << lane) - 1ull))] = c;
    }
}
// one stable pass on digit (key >> shift) & 15 from src to dst (n elements, all NT threads of the block); CT = counter type
// (uint16_t while n < 65536: half the LDS)
template <int NT, class CT, class PS, class PD>
__device__ __forceinline__ void hs_pass(PS src, PD dst, int n, int shift, CT *s_cnt /* 16 x NT */, int *s_scan) {
    const int t = threadIdx.x;
    // a thread's chunk: contiguous (stability), of ODD length (consecutive lanes then start 2 E dwords apart: all banks, where an
    // even length would put the 64 lanes of an 8-byte read on a handful of them)
    const int E = ((n + NT - 1) / NT) | 1, b = t * E < n ? t * E : n, e = b + E < n ? b + E : n;
    for (int k = 0; k < 16; k++) s_cnt[k * NT + t] = (CT)0;
    for (int i = b; i < e; i++) s_cnt[(int)((src[i] >> shift) & 15ull) * NT + t]++;
    __syncthreads();
    // exclusive scan of the 16 NT counters in (digit, thread) order: a thread takes 16 consecutive ones
    unsigned v[16], sum = 0u;
#pragma unroll
    for (int k = 0; k < 16; k++) { v[k] = s_cnt[t * 16 + k]; sum += v[k]; }
    int tot;
    unsigned run = (unsigned)block_excl_scan((int)sum, s_scan, &tot);
#pragma unroll
    for (int k = 0; k < 16; k++) { const unsigned x = v[k]; s_cnt[t * 16 + k] = (CT)run; run += x; }
    __syncthreads();
    for (int i = b; i < e; i++) {
        const unsigned long long x = src[i];
        dst[s_cnt[(int)((x >> shift) & 15ull) * NT + t]++] = x;
    }
    __syncthreads();
}
template <int CAP /* elements per LDS buffer; 0: global ping-pong */, int NT /* threads */>
__global__ void __launch_bounds__(NT) hit_segsort_kernel(const unsigned *__restrict__ count, const int32_t *__restrict__ list,
                                                         const int64_t *__restrict__ q_first, const int64_t *__restrict__ hit_off,
                                                         unsigned long long *__restrict__ hkey, unsigned long long *__restrict__ spare,
                                                         int shift0, int bits) {
    typedef typename std::conditional<(CAP > 0), uint16_t, unsigned>::type CT;
    __shared__ CT s_cnt[16 * NT];
    __shared__ int s_scan[16];
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_dyn[];
    const unsigned cnt = *count;
    const int passes = (bits + 3) / 4;
    for (unsigned it = blockIdx.x; it < cnt; it += gridDim.x) {
        const int c = list[it];
        const int64_t o = hit_off[q_first[c]];
        const int n = (int)(hit_off[q_first[c + 1]] - o);
        unsigned long long *g = hkey + o;
        if (CAP > 0) {
            hs_lptr a = (hs_lptr)s_dyn, bq = (hs_lptr)s_dyn + CAP;
            for (int i = threadIdx.x; i < n; i += NT) a[i] = g[i];
            __syncthreads();
            for (int ps = 0; ps < passes; ps++) {
                if (ps & 1) hs_pass<NT, CT>(bq, a, n, shift0 + 4 * ps, s_cnt, s_scan); else hs_pass<NT, CT>(a, bq, n, shift0 + 4 * ps, s_cnt, s_scan);
            }
            hs_lptr r = (passes & 1) ? bq : a;
            for (int i = threadIdx.x; i < n; i += NT) g[i] = r[i];
            __syncthreads();
        } else {
            unsigned long long *h = spare + o;
            for (int ps = 0; ps < passes; ps++) {
                if (ps & 1) hs_pass<NT, CT>(h, g, n, shift0 + 4 * ps, s_cnt, s_scan); else hs_pass<NT, CT>(g, h, n, shift0 + 4 * ps, s_cnt, s_scan);
            }
            if (passes & 1) { for (int i = threadIdx.x; i < n; i += NT) g[i] = h[i]; }
            __syncthreads();
        }
    }
}

__global__ void cluster_flag_kernel(int64_t nh, const unsigned long long *__restrict__ hkey, HitFmt F,
                                    const int64_t *__restrict__ coff, int nc, int32_t *__restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nh) return;
    int f = 1;
    if (i > 0) {
        unsigned long long a = hkey[i - 1], b = hkey[i];
        long long da = hit_dbias(F, a), db = hit_dbias(F, b);
        long long ga = da - DBIAS + hit_qo(F, a, i - 1), gb = db - DBIAS + hit_qo(F, b, i);
        f = hit_strand_key(F, a) != hit_strand_key(F, b) || db - da > C_TD;
        if (!f) { const int ca = contig_of(coff, nc, ga); f = gb < coff[ca] || gb >= coff[ca + 1]; }      // (one search: is gb inside ga's contig?)
    }
    flag[i] = f;
}

struct ClusterAcc { unsigned long long lo, hi; int cnt; int first; };

// first hit of every cluster (+ sentinel)
__global__ void cluster_first_kernel(int64_t nh, const int32_t *__restrict__ flag, const int64_t *__restrict__ cid_excl,
                                     unsigned *__restrict__ c_first, int64_t ncl) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nh) return;
    if (flag[i]) c_first[cid_excl[i]] = (unsigned)i;
    if (i == 0) c_first[ncl] = (unsigned)nh;
}
// per cluster: the extreme anchors (min qo -> smallest gpos, max qo -> largest gpos).  Thread per HIT: segmented min / max
// scan over the lanes of a wavefront (segments = clusters), the last lane of each segment holds the wave's partial result:
// stored directly when the whole cluster sits in this wave, merged with 64-bit atomics otherwise (c_lo / c_hi preset).
__global__ void __launch_bounds__(256) cluster_acc_kernel(int64_t nh, const unsigned long long *__restrict__ hkey,
                                                          HitFmt F, const int32_t *__restrict__ flag,
                                                          const int64_t *__restrict__ cid_excl, unsigned long long *__restrict__ c_lo,
                                                          unsigned long long *__restrict__ c_hi) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool have = i < nh;
    long long k = -1 - lane;   // lanes past the end: distinct fake clusters
    unsigned long long lo = 0xffffffffffffffffull, hi = 0;
    int head = 0;
    if (have) {
        head = flag[i];
        k = cid_excl[i] + head - 1;
        const unsigned long long hk = hkey[i];
        const long long d = hit_dbias(F, hk) - DBIAS;
        const unsigned qo = hit_qo(F, hk, i);
        lo = hi = ((unsigned long long)qo << 32) | ((unsigned long long)(d + qo) & 0xffffffffull);
    }
    int has_head = head;       // does the run of this cluster inside the wave include the cluster's first hit?
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const long long ok = __shfl_up(k, dd);
        const unsigned long long olo = __shfl_up(lo, dd), ohi = __shfl_up(hi, dd);
        const int oh = __shfl_up(has_head, dd);
        if (lane >= dd && ok == k) { lo = olo < lo ? olo : lo; hi = ohi > hi ? ohi : hi; has_head |= oh; }
    }
    const long long nk = __shfl_down(k, 1);
    const bool last_in_wave = lane == 63 || nk != k;
    if (have && last_in_wave) {
        const bool ends_here = i + 1 >= nh || flag[i + 1] != 0;   // the cluster's last hit
        if (has_head && ends_here) { c_lo[k] = lo; c_hi[k] = hi; }
        else { atomicMin(&c_lo[k], lo); atomicMax(&c_hi[k], hi); }
    }
}
__global__ void cluster_acc_init_kernel(int64_t ncl, unsigned long long *__restrict__ c_lo, unsigned long long *__restrict__ c_hi) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < ncl) { c_lo[k] = 0xffffffffffffffffull; c_hi[k] = 0; }
}

// ---- chains -> copies: base-level end extension + the reference's two coverage filters (definition: header of the twin) --
// chains = clusters with >= 3 anchors whose query span is >= 95 % of their genome span; their ids go to a dense list
// (a thread per cluster of the 10^8 diagonal clusters would leave one or two live lanes per wavefront in the extension).
// Chains with a long end to extend (>= EXT_LONG bases) are listed from the front, the others from the back of the same array:
// the extension takes its tasks front to back, so the long chains -- the tail of the kernel otherwise -- start first.
#define EXT_LONG 384
#define EXT_MAXLEN 2048     // an end of more than this many bases beyond the outermost anchor is not extended: no chain
#define CL_ITEMS 16         // clusters per thread: ONE pair of atomics on the two list counters per 4096 clusters (one per wavefront
                            // was a million same-address atomics: 10 ms)
__global__ void __launch_bounds__(256) chain_list_kernel(int64_t ncl, const unsigned long long *__restrict__ hkey, HitFmt F,
                                                         const unsigned *__restrict__ c_first, const unsigned long long *__restrict__ c_lo,
                                                         const unsigned long long *__restrict__ c_hi, const int64_t *__restrict__ cand_off,
                                                         unsigned *__restrict__ list, unsigned long long cap,
                                                         unsigned long long *__restrict__ counters /* [0] long, [1] short */) {
    __shared__ int s_scan[8];
    __shared__ unsigned long long s_base[2];
    for (int64_t tile = (int64_t)blockIdx.x * 256 * CL_ITEMS; tile < ncl; tile += (int64_t)gridDim.x * 256 * CL_ITEMS) {
        unsigned wantL = 0u, wantS = 0u;
#pragma unroll 4
        for (int it = 0; it < CL_ITEMS; it++) {
            const int64_t k = tile + it * 256 + threadIdx.x;
            if (k < ncl && (int)(c_first[k + 1] - c_first[k]) >= C_MINANCH) {
                const long long qlo = (long long)(c_lo[k] >> 32), glo = (long long)(c_lo[k] & 0xffffffffull);
                const long long qhi = (long long)(c_hi[k] >> 32), ghi = (long long)(c_hi[k] & 0xffffffffull);
                if ((qhi + CK - qlo) * 100 >= 95 * (ghi + CK - glo)) {
                    const unsigned c = hit_cand(F, hkey[c_first[k]]);
                    const long long Lq = cand_off[c + 1] - cand_off[c];
                    const long long nl = qlo, nr = Lq - (qhi + CK);
                    if (nl <= EXT_MAXLEN && nr <= EXT_MAXLEN) {
                        if (nl >= EXT_LONG || nr >= EXT_LONG) wantL |= 1u << it; else wantS |= 1u << it;
                    }
                }
            }
        }
        int tot;
        const int pre = block_excl_scan(__popc(wantL) | (__popc(wantS) << 16), s_scan, &tot);
        if (threadIdx.x == 0) {
            s_base[0] = (tot & 0xffff) ? atomicAdd(&counters[0], (unsigned long long)(tot & 0xffff)) : 0ull;
            s_base[1] = (tot >> 16) ? atomicAdd(&counters[1], (unsigned long long)(tot >> 16)) : 0ull;
        }
        __syncthreads();
        unsigned long long oL = s_base[0] + (unsigned long long)(pre & 0xffff), oS = s_base[1] + (unsigned long long)(pre >> 16);
        for (int it = 0; it < CL_ITEMS; it++) {
            const unsigned k = (unsigned)(tile + it * 256 + threadIdx.x);
            if ((wantL >> it) & 1u) { if (oL < cap) list[oL] = k; oL++; }
            if ((wantS >> it) & 1u) { if (oS < cap) list[cap - 1 - oS] = k; oS++; }
        }
        __syncthreads();
    }
}
// chain number e (long ones first) -> cluster id
__device__ __forceinline__ unsigned chain_at(const unsigned *__restrict__ list, unsigned long long cap, unsigned long long n_long, unsigned long long e) {
    return e < n_long ? list[e] : list[cap - 1 - (e - n_long)];
}

#include "hite_ext.h"
using ExtState = ExtStateT<ExtCopyMode>;

// one LANE per (chain, end) task -- even tasks extend to the left of the first anchor, odd ones to the right of the last --
// and a lane that has finished takes the next task from the queue while its neighbours go on (the lengths run from 0 to
// thousands of columns: with a fixed task per thread every wavefront waited for its longest).  Refill when a quarter of
// the lanes is idle, so that the set-up code is paid for 16 tasks at a time.
__global__ void __launch_bounds__(256) chain_extend_kernel(const unsigned long long *__restrict__ counters, const unsigned *__restrict__ list,
                                                           const unsigned long long *__restrict__ hkey, HitFmt F, const unsigned *__restrict__ c_first,
                                                           const unsigned long long *__restrict__ c_lo, const unsigned long long *__restrict__ c_hi,
                                                           const uint8_t *__restrict__ cand, const int64_t *__restrict__ cand_off,
                                                           const uint32_t *__restrict__ bases, const uint32_t *__restrict__ nmask,
                                                           const int64_t *__restrict__ coff, int nc, unsigned long long cap,
                                                           unsigned long long *__restrict__ queue, int32_t *__restrict__ x_i, int32_t *__restrict__ x_t) {
    unsigned long long n_long = counters[0], n_short = counters[1];
    if (n_long > cap) n_long = cap;
    if (n_long + n_short > cap) n_short = cap - n_long;
    const unsigned long long ntask = 2ull * (n_long + n_short);
    const int lane = threadIdx.x & 63;
    __shared__ uint32_t s_lut[EXT_LUT_WORDS];          // the column-minimum tables of the bit-parallel band (hite_ext.h)
    for (int t = threadIdx.x; t < EXT_LUT_WORDS; t += 256) s_lut[t] = ext_lut_entry(t >> 8, t & 255);
    __syncthreads();
    EXT_LUT_PTR lut = (EXT_LUT_PTR)s_lut;
    ExtState E;
    E.n = 0; E.i = 1;
    bool active = false, exhausted = false;
    unsigned long long my = 0, cols_done = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (!exhausted && (idle == ~0ull || __popcll(idle) >= 16)) {
            const int leader = __ffsll((long long)idle) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(queue, (unsigned long long)__popcll(idle));
            base = ((unsigned long long)(unsigned)__shfl((int)(base >> 32), leader) << 32) | (unsigned)__shfl((int)base, leader);
            if (base + (unsigned long long)__popcll(idle) >= ntask) exhausted = true;
            if (!active) {
                const unsigned long long t = base + (unsigned long long)__popcll(idle & ((1ull << lane) - 1ull));
                if (t < ntask) {
                    const unsigned k = chain_at(list, cap, n_long, t >> 1);
                    const int side = (int)(t & 1ull);
                    const unsigned long long key = hkey[c_first[k]];
                    const unsigned c = hit_cand(F, key), rel = hit_rel(F, key);
                    const int64_t qb = cand_off[c];
                    const int Lq = (int)(cand_off[c + 1] - qb);
                    const long long qlo = (long long)(c_lo[k] >> 32), glo = (long long)(c_lo[k] & 0xffffffffull);
                    const long long qhi = (long long)(c_hi[k] >> 32), ghi = (long long)(c_hi[k] & 0xffffffffull);
                    const int ctg = contig_of(coff, nc, glo);
                    // the query in the orientation of the genome: rel = 1 reads the reverse complement of the candidate, x -> Lq - 1 - x
                    if (side == 0)        // x = qlo - 1, qlo - 2, ..., 0
                        ext_init(E, cand, rel ? qb + Lq - qlo : qb + qlo - 1, rel ? +1 : -1, rel != 0, (int)qlo, bases, nmask, glo, -1, glo - coff[ctg]);
                    else                  // x = qhi + K, qhi + K + 1, ...
                        ext_init(E, cand, rel ? qb + Lq - 1 - (qhi + CK) : qb + qhi + CK, rel ? -1 : +1, rel != 0, (int)(Lq - (qhi + CK)), bases, nmask,
                                 ghi + CK, +1, coff[ctg + 1] - (ghi + CK));
                    my = t;
                    if (E.n >= 1) active = true;
                    else { x_i[t] = 0; x_t[t] = 0; }
                }
            }
        }
        if (__ballot(active) == 0ull) { if (exhausted) break; else continue; }
        if (active) {
            cols_done++;
            if (ext_step(E, bases, nmask, lut)) { x_i[my] = E.best_i; x_t[my] = E.best_t; active = false; }
        }
    }
    // statistics: DP columns computed (one atomic per wavefront)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cols_done += ((unsigned long long)(unsigned)__shfl_xor((int)(cols_done >> 32), d) << 32) | (unsigned)__shfl_xor((int)cols_done, d);
    if (lane == 0 && cols_done) atomicAdd(queue + 1, cols_done);
}

// accepted chains -> copy records + sort key (candidate:19 | 4095-anchors:12 | start:32 | minus:1).  Two passes:
// WRITE = false counts the accepted chains per candidate, WRITE = true places each record at
// cstart[candidate] + (a per-candidate atomic counter): no single hot append counter (740 k same-address atomics cost 5 ms).
// aligned = the candidate minus what the two extensions clipped; the two filters of get_copies_minimap2 (Util.py:8008-8022)
// are evaluated on the aligned part, the interval handed on covers the whole candidate (clipped ends on the diagonal of the
// last aligned base, <= 5 % of the candidate: see the twin's header).
template <bool WRITE>
__global__ void __launch_bounds__(256) chain_copy_kernel(const unsigned long long *__restrict__ counters, unsigned long long cap,
                                                         const unsigned *__restrict__ list, const int32_t *__restrict__ x_i,
                                                         const int32_t *__restrict__ x_t, const unsigned long long *__restrict__ hkey, HitFmt F,
                                                         const unsigned *__restrict__ c_first, const unsigned long long *__restrict__ c_lo,
                                                         const unsigned long long *__restrict__ c_hi, const int64_t *__restrict__ cand_off,
                                                         const int64_t *__restrict__ coff, int nc, unsigned long long *__restrict__ ckey,
                                                         unsigned *__restrict__ cval, int32_t *__restrict__ r_contig, int64_t *__restrict__ r_s1,
                                                         int64_t *__restrict__ r_e1, uint8_t *__restrict__ r_minus, int32_t *__restrict__ r_anch,
                                                         uint32_t *__restrict__ r_clip, int32_t *__restrict__ per_cand,
                                                         const int64_t *__restrict__ cstart, int aligned_iv) {
    unsigned long long n_long = counters[0], n_short = counters[1];
    if (n_long > cap) n_long = cap;
    if (n_long + n_short > cap) n_short = cap - n_long;
    const unsigned long long nch = n_long + n_short;
    for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < nch; e += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned k = chain_at(list, cap, n_long, e);
        const int na = (int)(c_first[k + 1] - c_first[k]);
        const unsigned long long key = hkey[c_first[k]];
        const unsigned c = hit_cand(F, key), rel = hit_rel(F, key);
        const long long Lq = cand_off[c + 1] - cand_off[c];
        const long long qlo = (long long)(c_lo[k] >> 32), glo = (long long)(c_lo[k] & 0xffffffffull);
        const long long qhi = (long long)(c_hi[k] >> 32), ghi = (long long)(c_hi[k] & 0xffffffffull);
        const long long clip_l = qlo - x_i[2 * e], clip_r = Lq - (qhi + CK) - x_i[2 * e + 1];
        const long long aligned = Lq - clip_l - clip_r;
        const long long a0 = glo - x_t[2 * e], a1 = ghi + CK + x_t[2 * e + 1];          // genome interval of the aligned part
        if (!(a1 > a0 && aligned * 100 >= 95 * Lq && aligned * 100 >= 95 * (a1 - a0))) continue;   // Util.py:8008-8022
        if (!WRITE) { atomicAdd(&per_cand[c], 1); continue; }
        const int ctg = contig_of(coff, nc, glo);
        const long long cb = coff[ctg], ce = coff[ctg + 1];
        // whole-candidate interval (default) or the aligned interval as get_copies_minimap2 reports it (Util.py:8026; hite_copy_config)
        long long s0 = aligned_iv ? a0 : a0 - clip_l, e0 = aligned_iv ? a1 : a1 + clip_r;
        if (s0 < cb) s0 = cb;
        if (e0 > ce) e0 = ce;
        const int64_t slot = cstart[c] + atomicAdd(&per_cand[c], 1);
        r_contig[slot] = ctg; r_s1[slot] = s0 - cb + 1; r_e1[slot] = e0 - cb; r_minus[slot] = (uint8_t)rel; r_anch[slot] = na;
        // aligned interval: the candidate bases the extensions clipped (<= 5 % of it each) travel beside the record -- the rows of the
        // star alignment are padded by them (hite_flank_region_align_clip_dev); whole-candidate interval: they are inside it, no pads
        r_clip[slot] = aligned_iv ? ((uint32_t)(clip_l > 0xffff ? 0xffff : clip_l) | ((uint32_t)(clip_r > 0xffff ? 0xffff : clip_r) << 16)) : 0u;
        const int ac = na > 4095 ? 4095 : na;
        ckey[slot] = ((unsigned long long)c << 45) | ((unsigned long long)(4095 - ac) << 33) | ((unsigned long long)(unsigned)s0 << 1) | rel;
        cval[slot] = (unsigned)slot;
    }
}

__global__ void cap300_kernel(int n, const int32_t *__restrict__ in, int32_t *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] > C_MAXCOPY ? C_MAXCOPY : in[i];
}

__global__ void emit_copies_kernel(int64_t ncp, const unsigned long long *__restrict__ ckey, const unsigned *__restrict__ cval,
                                   const int64_t *__restrict__ cstart /* per candidate, all accepted */,
                                   const int64_t *__restrict__ ofirst /* per candidate, kept */, const int32_t *__restrict__ r_contig,
                                   const int64_t *__restrict__ r_s1, const int64_t *__restrict__ r_e1, const uint8_t *__restrict__ r_minus,
                                   const int32_t *__restrict__ r_anch, const uint32_t *__restrict__ r_clip, int32_t *__restrict__ o_contig,
                                   int64_t *__restrict__ o_s1, int64_t *__restrict__ o_e1, uint8_t *__restrict__ o_minus,
                                   int32_t *__restrict__ o_anch, uint32_t *__restrict__ o_clip) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncp) return;
    unsigned c = (unsigned)(ckey[i] >> 45);
    int64_t rank = i - cstart[c];
    if (rank >= C_MAXCOPY) return;
    int64_t o = ofirst[c] + rank;
    unsigned s = cval[i];
    o_contig[o] = r_contig[s]; o_s1[o] = r_s1[s]; o_e1[o] = r_e1[s]; o_minus[o] = r_minus[s]; o_anch[o] = r_anch[s]; o_clip[o] = r_clip[s];
}
__global__ void i64_to_i32_kernel(int64_t n, const int64_t *__restrict__ in, int32_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}
__global__ void fill_u64_kernel(int64_t n, unsigned long long *__restrict__ p, unsigned long long v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
#define CGRID(n) dim3((unsigned)((((n) > 0 ? (n) : 1) + 255) / 256)), dim3(256)
#define CCHK(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

static int sorter_from_arena(Sorter &S, hite_ctx *ctx, Arena &A, hipStream_t st, int64_t n, bool with_vals = true) {
    S.ctx = ctx; S.st = st; S.cap = n;
    S.hist_n = sorter_hist_elems(n);
    void *p;
    CCHK(arena_alloc(ctx, A, (size_t)(n + 1) * 8, &p)); S.k2 = (unsigned long long *)p;
    S.v2 = nullptr;
    if (with_vals) { CCHK(arena_alloc(ctx, A, (size_t)(n + 1) * 4, &p)); S.v2 = (unsigned *)p; }
    CCHK(arena_alloc(ctx, A, (size_t)S.hist_n * sizeof(rs_cnt_t), &p)); S.hist = (rs_cnt_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(S.hist_n + 1) * sizeof(rs_off_t), &p)); S.offs = (rs_off_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)sorter_tmp_elems(S.hist_n) * 8, &p)); S.bs = (int64_t *)p;
    return HITE_OK;
}

// which interval a copy record carries: 1 (default since round 5) = the ALIGNED interval, reference_start + 1 .. reference_end as
// get_copies_minimap2 reports it (Util.py:8026), the clipped candidate bases beside it; 0 = the interval of the WHOLE candidate
// (clipped ends extrapolated on the diagonal; rounds 2-4).  Per context (hite_copy_config_ctx); a context that was never
// configured follows the process default: hite_copy_config, initialised from the environment (HITE_COPY_INTERVAL=aligned | whole).
static int g_copy_interval = -1;
static int copy_interval_mode(const hite_ctx *ctx) {
    if (ctx && ctx->copy_interval >= 0) return ctx->copy_interval;
    int v = __atomic_load_n(&g_copy_interval, __ATOMIC_RELAXED);
    if (v < 0) {
        const char *e = getenv("HITE_COPY_INTERVAL");
        v = (e && (!strcmp(e, "whole") || !strcmp(e, "0"))) ? 0 : 1;
        __atomic_store_n(&g_copy_interval, v, __ATOMIC_RELAXED);
    }
    return v;
}
extern "C" int hite_copy_config(int32_t aligned_interval) {
    if (aligned_interval < -1 || aligned_interval > 1) return HITE_EINVAL;
    __atomic_store_n(&g_copy_interval, aligned_interval, __ATOMIC_RELAXED);
    return HITE_OK;
}
extern "C" int hite_copy_config_ctx(hite_ctx *ctx, int32_t aligned_interval) {
    if (!ctx || aligned_interval < -1 || aligned_interval > 1) return HITE_EINVAL;
    ctx->copy_interval = aligned_interval;
    return HITE_OK;
}

extern "C" void hite_copy_index_release(void *state) {
    CopyState *S = (CopyState *)state;
    if (!S) return;
    if (S->idx_hs) (void)hipFree(S->idx_hs);
    if (S->idx_pos) (void)hipFree(S->idx_pos);
    if (S->idx_t) (void)hipFree(S->idx_t);
    if (S->dir) (void)hipFree(S->dir);
    arena_free(S->build);
    arena_free(S->arena);
    arena_free(S->out);
    if (S->h_pin) (void)hipHostFree(S->h_pin);
    if (S->d_scal) (void)hipFree(S->d_scal);
    delete S;
}

static int read_back(hite_ctx *ctx, CopyState *S, hipStream_t st, int count) {
    HITE_CHECK(ctx, hipMemcpyAsync(S->h_pin, S->d_scal, sizeof(int64_t) * count, hipMemcpyDeviceToHost, st));
    HITE_CHECK(ctx, hipStreamSynchronize(st));
    return HITE_OK;
}

// builds the minimizer index of the packed genome; *state_io receives the index handle
// device temporaries of a host-side function: freed on every return path
struct DevTmp {
    void *p[8];
    int n = 0;
    ~DevTmp() { for (int i = 0; i < n; i++) if (p[i]) (void)hipFree(p[i]); }
    hipError_t alloc(void **out, size_t bytes) {
        hipError_t e = hipMalloc(out, bytes ? bytes : 16);
        if (e == hipSuccess && n < 8) p[n++] = *out;
        return e;
    }
};

static int copy_state_get(hite_ctx *ctx, void **state_io, CopyState **out) {
    CopyState *S = (CopyState *)*state_io;
    if (!S) {
        S = new CopyState();
        *state_io = S;
        HITE_CHECK(ctx, hipHostMalloc((void **)&S->h_pin, 64 * sizeof(int64_t)));
        HITE_CHECK(ctx, hipMalloc((void **)&S->d_scal, 64 * sizeof(int64_t)));
    }
    *out = S;
    return HITE_OK;
}

// the index of the packed genome; hset != null: only the minimizers whose hash is in the set (a restricted index)
static int index_build_impl(hite_ctx *ctx, void **state_io, hipStream_t st, HSet hset) {
    if (!ctx || !ctx->d_bases || !state_io) return HITE_EINVAL;
    if (ctx->n_bases >= 0xfffe0000ll) return HITE_EINVAL;  // positions are 32-bit
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    // rebuilding on an existing handle (next genome / chunk) keeps its arenas: their growth is the expensive part of a cold call
    CopyState *S;
    CCHK(copy_state_get(ctx, state_io, &S));
    S->M = 0;
    S->restricted = hset.tab != nullptr;
    const int64_t G = ctx->n_bases;
    const unsigned long long cap = (unsigned long long)(G * 0.32) + 4096;
    if ((int64_t)cap > S->idx_cap) {      // the index arrays: grow-only, shared by every genome packed behind this handle
        if (S->idx_hs) (void)hipFree(S->idx_hs);
        if (S->idx_pos) (void)hipFree(S->idx_pos);
        if (S->idx_t) (void)hipFree(S->idx_t);
        S->idx_hs = S->idx_pos = S->idx_t = nullptr; S->idx_cap = 0;
        HITE_CHECK(ctx, hipMalloc((void **)&S->idx_hs, (size_t)(cap + 16) * 4));
        HITE_CHECK(ctx, hipMalloc((void **)&S->idx_pos, (size_t)(cap + 16) * 4));
        HITE_CHECK(ctx, hipMalloc((void **)&S->idx_t, (size_t)(cap + 16) * 4));
        S->idx_cap = (int64_t)cap;
    }
    if (!S->dir) HITE_CHECK(ctx, hipMalloc((void **)&S->dir, (size_t)((1 << DIRBITS) + 2) * 4));
    const bool one_chunk = S->build.chunks.size() == 1;       // (more: the reset below rebuilds the arena, and what it held is gone)
    CCHK(arena_reset(ctx, S->build, true));
    Arena &B = S->build;
    void *p;
    const int64_t ntiles = (G + GM_TILE - 1) / GM_TILE;
    unsigned long long *keys, *stage;
    unsigned *vals;
    int32_t *tile_cnt; int64_t *tile_first, *tbs;
    // (kept tiles: every build on a handle allocates the same things in the same order, so the tiles of the last build are where the
    // new ones go -- unless the arena was rebuilt, which the pointers below and the chunk count tell)
    static const bool keep_on = [] { const char *v = getenv("HITE_KEEP_MINIMIZERS"); return !(v && *v == '0'); }();
    CCHK(arena_alloc(ctx, B, (size_t)cap * 8, &p)); keys = (unsigned long long *)p;
    CCHK(arena_alloc(ctx, B, (size_t)cap * 4, &p)); vals = (unsigned *)p;
    CCHK(arena_alloc(ctx, B, (size_t)(ntiles + 1) * 4, &p)); tile_cnt = (int32_t *)p;
    CCHK(arena_alloc(ctx, B, (size_t)(ntiles + 2) * 8, &p)); tile_first = (int64_t *)p;
    CCHK(arena_alloc(ctx, B, (size_t)scan_tmp_elems(ntiles + 1) * 8, &p)); tbs = (int64_t *)p;
    CCHK(arena_alloc(ctx, B, (size_t)(ntiles > 0 ? ntiles : 1) * GM_TILE * 8, &p)); stage = (unsigned long long *)p;
    int32_t *tile_cnt_r = nullptr;       // restricted build from kept tiles: the filtered counts
    unsigned long long *wmask = nullptr;
    if (keep_on && hset.tab) {
        CCHK(arena_alloc(ctx, B, (size_t)(ntiles + 1) * 4, &p)); tile_cnt_r = (int32_t *)p;
        CCHK(arena_alloc(ctx, B, (size_t)(ntiles + 1) * (GM_TILE / 64) * 8, &p)); wmask = (unsigned long long *)p;
    }
    int64_t blocks = ntiles < 256 * 64 ? ntiles : 256 * 64;
    if (blocks < 1) blocks = 1;
    int tk_gm = hite_prof_begin(ctx, "index_minimizers", st);
    const HSet none{nullptr, nullptr, 0, 0};
    if (!keep_on) {
        S->kept_epoch = -1;
        if (hset.tab)
            hipLaunchKernelGGL(genome_minimizer_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, ctx->d_bases, ctx->d_nmask, ctx->d_contig_off,
                               ctx->n_contigs, G, ntiles, stage, tile_cnt, hset, (const int32_t *)nullptr);
        else
            hipLaunchKernelGGL(genome_minimizer_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, ctx->d_bases, ctx->d_nmask, ctx->d_contig_off,
                               ctx->n_contigs, G, ntiles, stage, tile_cnt, none, (const int32_t *)nullptr);
    } else {
        // the tiles always hold EVERY minimizer (a restricted build filters them below).  Kept from the last build on this genome state:
        // only the tiles a mask has touched since are done again -- window starts [a - K - W - 2, b + 2] of a masked [a, b): a window
        // reads the k-mers of W starts and a k-mer K bases, and a window's record also depends on the window before it
        const bool kept = one_chunk && S->kept_stage == stage && S->kept_cnt == tile_cnt && S->kept_ctx == ctx && S->kept_G == G &&
                          S->kept_epoch == ctx->genome_epoch && S->kept_log <= ctx->mask_log_n;
        S->kept_epoch = -1;                       // (void until this build's kernels are queued)
        if (!kept)
            hipLaunchKernelGGL(genome_minimizer_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, ctx->d_bases, ctx->d_nmask, ctx->d_contig_off,
                               ctx->n_contigs, G, ntiles, stage, tile_cnt, none, (const int32_t *)nullptr);
        else if (S->kept_log < ctx->mask_log_n) {
            std::vector<unsigned char> dirty((size_t)ntiles, 0);
            for (int64_t k = S->kept_log; k < ctx->mask_log_n; k++) {
                int64_t lo = ctx->mask_log[2 * k] - CK - CW - 2, hi = ctx->mask_log[2 * k + 1] + 2;
                if (lo < 0) lo = 0;
                if (hi > G - 1) hi = G - 1;
                for (int64_t t = lo / GM_TILE; t <= hi / GM_TILE && t < ntiles; t++) dirty[(size_t)t] = 1;
            }
            std::vector<int32_t> list;
            for (int64_t t = 0; t < ntiles; t++) if (dirty[(size_t)t]) list.push_back((int32_t)t);
            if (!list.empty()) {
                int32_t *d_list;
                CCHK(arena_alloc(ctx, B, list.size() * 4, &p)); d_list = (int32_t *)p;
                HITE_CHECK(ctx, hipMemcpyAsync(d_list, list.data(), list.size() * 4, hipMemcpyHostToDevice, st));
                HITE_CHECK(ctx, hipStreamSynchronize(st));          // (the list is a local)
                const int64_t nl = (int64_t)list.size();
                hipLaunchKernelGGL(genome_minimizer_kernel<false>, dim3((unsigned)(nl < blocks ? nl : blocks)), dim3(256), 0, st, ctx->d_bases, ctx->d_nmask,
                                   ctx->d_contig_off, ctx->n_contigs, G, nl, stage, tile_cnt, none, (const int32_t *)d_list);
            }
        }
        S->kept_stage = stage; S->kept_cnt = tile_cnt; S->kept_ctx = ctx; S->kept_G = G;
        S->kept_epoch = ctx->genome_epoch; S->kept_log = ctx->mask_log_n;
    }
    int64_t M = 0;
    if (ntiles > 0) {
        if (tile_cnt_r) {
            hipLaunchKernelGGL(genome_minimizer_filter_kernel<false>, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, ntiles, stage, tile_cnt, hset,
                               tile_cnt_r, (const int64_t *)nullptr, (unsigned long long *)nullptr, (unsigned *)nullptr, cap, wmask);
            CCHK(scan_excl_buf<int32_t>(ctx, tbs, tile_cnt_r, ntiles, tile_first, st));
            HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal, tile_first + ntiles, 8, hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(genome_minimizer_filter_kernel<true>, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, ntiles, stage, tile_cnt, hset,
                               (int32_t *)nullptr, tile_first, keys, vals, cap, wmask);
        } else {
            CCHK(scan_excl_buf<int32_t>(ctx, tbs, tile_cnt, ntiles, tile_first, st));
            HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal, tile_first + ntiles, 8, hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(genome_minimizer_pack_kernel, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, ntiles, stage, tile_first, keys, vals, cap);
        }
        hite_prof_end(ctx, tk_gm, st);
        HITE_CHECK(ctx, hipGetLastError());
        CCHK(read_back(ctx, S, st, 1));
        M = S->h_pin[0];
    } else hite_prof_end(ctx, tk_gm, st);
    if ((unsigned long long)M > cap) return HITE_ECAP;
    S->M = M;
    // stable sort on the hash (the key's upper word): equal hashes stay in position order
    int tk_is = hite_prof_begin(ctx, "index_sort", st);
    if (M > 1) {
        Sorter so;
        CCHK(sorter_from_arena(so, ctx, B, st, M));
        CCHK(sorter_sort_bits_swap(so, &keys, &vals, M, 32, 64));
    }
    hite_prof_end(ctx, tk_is, st);
    int tk_dir = hite_prof_begin(ctx, "index_directory", st);
    S->idx_key = keys;
    hipLaunchKernelGGL(fill_u32_kernel, CGRID((int64_t)(1 << DIRBITS) + 2), 0, st, (int64_t)(1 << DIRBITS) + 2, S->dir, (unsigned)M);
    if (M > 0) {
        hipLaunchKernelGGL(split_index_kernel, CGRID(M), 0, st, M, keys, S->idx_hs, S->idx_pos);
        HITE_CHECK(ctx, hipMemcpyAsync(S->idx_t, vals, (size_t)M * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(index_directory_kernel, CGRID(M), 0, st, M, S->idx_hs, S->dir);
    }
    hite_prof_end(ctx, tk_dir, st);
    HITE_CHECK(ctx, hipGetLastError());
    HITE_CHECK(ctx, hipStreamSynchronize(st));
    return HITE_OK;
}

extern "C" int hite_copy_index_forget(void *state) {
    if (state) ((CopyState *)state)->kept_epoch = -1;
    return HITE_OK;
}
extern "C" int hite_copy_index_build(hite_ctx *ctx, void **state_io, void *stream) {
    return index_build_impl(ctx, state_io, (hipStream_t)stream, HSet{nullptr, nullptr, 0, 0});
}

// candidates (device) -> copy table (device arrays owned by the index state's arena; valid until the next call)
static int find_copies_impl(hite_ctx *ctx, void *state, int32_t n_cand, const uint8_t *d_cand, const int64_t *d_cand_off,
                            int64_t cand_bytes, int32_t **d_copy_first, int64_t *n_copies, int32_t **d_contig,
                            int64_t **d_start1, int64_t **d_end1, uint8_t **d_minus, int32_t **d_anchors, void *stream,
                            bool restricted_ok /* the index was restricted to THESE candidates by the caller */) {
    CopyState *S = (CopyState *)state;
    if (!ctx || !S || !ctx->d_bases || n_cand < 0 || n_cand >= (1 << 19) || !n_copies) return HITE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    if (S->restricted && !restricted_ok) {     // an index restricted to another candidate set answers nothing else: the full one
        void *sp = S;
        CCHK(index_build_impl(ctx, &sp, st, HSet{nullptr, nullptr, 0, 0}));
    }
    CCHK(arena_reset(ctx, S->arena, true));
    CCHK(arena_reset(ctx, S->out, true));
    Arena &A = S->arena;
    void *p;
    int32_t *ofirst32;
    CCHK(arena_alloc(ctx, S->out, (size_t)(n_cand + 2) * 4, &p)); ofirst32 = (int32_t *)p;
    *d_copy_first = ofirst32; *n_copies = 0;
    S->out_clip = nullptr; S->out_n = 0;
    HITE_CHECK(ctx, hipMemsetAsync(ofirst32, 0, (size_t)(n_cand + 2) * 4, st));
    *d_contig = nullptr; *d_start1 = nullptr; *d_end1 = nullptr; *d_minus = nullptr; *d_anchors = nullptr;
    if (n_cand == 0 || cand_bytes <= 0 || S->M == 0) return HITE_OK;
    // candidate minimizers
    unsigned long long qcap = (unsigned long long)(cand_bytes * 0.32) + 4096 + (unsigned long long)n_cand;
    unsigned *q_c, *q_pos, *q_hs, *occ_lo, *hval, *c_first, *cval;
    int32_t *occ_n, *flag, *per_cand, *per_cand300;
    int64_t *hit_off, *cid, *bs, *cstart, *ofirst;
    unsigned long long *hkey, *c_lo, *c_hi, *ckey;
    CCHK(arena_alloc(ctx, A, qcap * 4, &p)); q_c = (unsigned *)p;
    CCHK(arena_alloc(ctx, A, qcap * 4, &p)); q_pos = (unsigned *)p;
    CCHK(arena_alloc(ctx, A, qcap * 4, &p)); q_hs = (unsigned *)p;
    HITE_CHECK(ctx, hipMemsetAsync(S->d_scal, 0, 64, st));
    int64_t nq, max_cand_len = 0;
    int64_t *q_first = nullptr;          // first minimizer of every candidate (n_cand + 1): the hits of a candidate are one contiguous range
    {
        int32_t *q_cnt; int64_t *qbs; unsigned *r_pos, *r_hs;
        CCHK(arena_alloc(ctx, A, (size_t)(n_cand + 1) * 4, &p)); q_cnt = (int32_t *)p;
        CCHK(arena_alloc(ctx, A, (size_t)(n_cand + 2) * 8, &p)); q_first = (int64_t *)p;
        CCHK(arena_alloc(ctx, A, (size_t)scan_tmp_elems(n_cand) * 8, &p)); qbs = (int64_t *)p;
        CCHK(arena_alloc(ctx, A, (size_t)(cand_bytes + 64) * 4, &p)); r_pos = (unsigned *)p;
        CCHK(arena_alloc(ctx, A, (size_t)(cand_bytes + 64) * 4, &p)); r_hs = (unsigned *)p;
        int wblocks = n_cand < 65536 ? n_cand : 65536;
        int tk_cm = hite_prof_begin(ctx, "cand_minimizer_kernel", st);
        hipLaunchKernelGGL(cand_minimizer_kernel, dim3(wblocks), dim3(256), 0, st, n_cand, d_cand, d_cand_off, r_pos, r_hs, q_cnt,
                           (unsigned long long *)(S->d_scal + 1));
        CCHK(scan_excl_buf<int32_t>(ctx, qbs, q_cnt, n_cand, q_first, st));
        HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal, q_first + n_cand, 8, hipMemcpyDeviceToDevice, st));
        CCHK(read_back(ctx, S, st, 2));
        nq = S->h_pin[0];
        max_cand_len = S->h_pin[1];
        if ((unsigned long long)nq > qcap) return HITE_ECAP;
        S->last[0] = nq; S->last[1] = S->last[2] = S->last[3] = S->last[4] = S->last[5] = S->last[6] = 0;
        hipLaunchKernelGGL(cand_minimizer_pack_kernel, dim3((n_cand + 3) / 4 < 8192 ? (n_cand + 3) / 4 : 8192), dim3(256), 0, st, n_cand,
                           d_cand_off, r_pos, r_hs, q_first, q_c, q_pos, q_hs);
        hite_prof_end(ctx, tk_cm, st);
    }
    if (nq == 0) return HITE_OK;
    CCHK(arena_alloc(ctx, A, (size_t)nq * 4, &p)); occ_lo = (unsigned *)p;
    CCHK(arena_alloc(ctx, A, (size_t)nq * 4, &p)); occ_n = (int32_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(nq + 1) * 8, &p)); hit_off = (int64_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)scan_tmp_elems(nq) * 8, &p)); bs = (int64_t *)p;
    int tk_occ_kernel = hite_prof_begin(ctx, "occ_kernel", st);
    hipLaunchKernelGGL(occ_kernel, CGRID(nq), 0, st, nq, q_hs, S->idx_hs, S->dir, occ_lo, occ_n);
    hite_prof_end(ctx, tk_occ_kernel, st);
    CCHK(scan_excl_buf<int32_t>(ctx, bs, occ_n, nq, hit_off, st));
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal, hit_off + nq, 8, hipMemcpyDeviceToDevice, st));
    CCHK(read_back(ctx, S, st, 1));
    const int64_t nh = S->h_pin[0];
    S->last[1] = nh;
    if (nh == 0) return HITE_OK;
    if (nh >= 0xffffffffll) return HITE_ECAP;
    int cbits = 1; while ((1ll << cbits) < n_cand) cbits++;
    int dbits = 1; while ((1ll << dbits) < ctx->n_bases + DBIAS + 1) dbits++;
    if (dbits > 33) dbits = 33;
    int qbits = 1; while ((1ll << qbits) < max_cand_len) qbits++;
    HitFmt F;
    F.qbits = qbits + dbits + 1 + cbits <= 64 ? qbits : 0; F.dbits = dbits; F.hval = nullptr;
    {   // HITE_HITS_WIDE=1 (tests): the 12-byte form whatever the sizes
        static const bool wide_only = [] { const char *e = getenv("HITE_HITS_WIDE"); return e && *e && atoi(e) != 0; }();
        if (wide_only) F.qbits = 0;
    }
    CCHK(arena_alloc(ctx, A, (size_t)(nh + 1) * 8, &p)); hkey = (unsigned long long *)p;
    hval = nullptr;
    if (!F.qbits) { CCHK(arena_alloc(ctx, A, (size_t)(nh + 1) * 4, &p)); hval = (unsigned *)p; F.hval = hval; }
    int tk_hit_kernel = hite_prof_begin(ctx, "hit_kernel", st);
    hipLaunchKernelGGL(hit_kernel, CGRID(nq), 0, st, nq, q_c, q_pos, q_hs, d_cand_off, S->idx_hs, S->idx_pos, occ_lo, occ_n, hit_off, hkey, hval, F);
    hite_prof_end(ctx, tk_hit_kernel, st);
    {
        static const bool hist = [] { const char *e = getenv("HITE_HIT_HIST"); return e && *e && atoi(e) != 0; }();
        static bool printed = false;
        if (hist && !printed) {
            printed = true;
            unsigned long long *d_h, h_h[11];
            CCHK(arena_alloc(ctx, A, 11 * 8, &p)); d_h = (unsigned long long *)p;
            HITE_CHECK(ctx, hipMemsetAsync(d_h, 0, 11 * 8, st));
            hipLaunchKernelGGL(hit_hist_kernel, dim3((n_cand + 255) / 256), dim3(256), 0, st, n_cand, q_first, hit_off, d_h);
            HITE_CHECK(ctx, hipMemcpyAsync(h_h, d_h, 11 * 8, hipMemcpyDeviceToHost, st));
            HITE_CHECK(ctx, hipStreamSynchronize(st));
            fprintf(stderr, "hit_hist: candidates %d hits %lld max %llu | classes <=1024 <=2048 <=4096 <=8192 above: candidates %llu %llu %llu %llu %llu hits %llu %llu %llu %llu %llu\n",
                    n_cand, (long long)nh, h_h[10], h_h[0], h_h[1], h_h[2], h_h[3], h_h[4], h_h[5], h_h[6], h_h[7], h_h[8], h_h[9]);
        }
    }
    Sorter so;
    CCHK(sorter_from_arena(so, ctx, A, st, nh));
    int tk_sh = hite_prof_begin(ctx, "radix_sort_hits", st);
    {   // the key has a hole: the diagonal (gpos - qo + DBIAS < n_bases + DBIAS) rarely needs its 33 bits.  Two runs of stable
        // passes -- the diagonal's bits, then strand + candidate -- take 3 + 2 passes at 1 Gbp / 2^17 candidates where the
        // 51-bit key as a whole takes 6; the sorted pair of buffers is taken over instead of copied back
        static const bool seg_sort = [] { const char *e = getenv("HITE_HIT_SEGSORT"); return !(e && *e && atoi(e) == 0); }();
        if (F.qbits && seg_sort) {
            // per-candidate sort in LDS (above): classify, then one launch per class
            unsigned *hs_counts; int32_t *hs_lists;
            CCHK(arena_alloc(ctx, A, 16, &p)); hs_counts = (unsigned *)p;
            CCHK(arena_alloc(ctx, A, (size_t)HS_NCLS * n_cand * 4 + 16, &p)); hs_lists = (int32_t *)p;
            HITE_CHECK(ctx, hipMemsetAsync(hs_counts, 0, 16, st));
            hipLaunchKernelGGL(hit_sort_classify_kernel, dim3((n_cand + 255) / 256), dim3(256), 0, st, n_cand, q_first, hit_off, hs_counts, hs_lists);
            static unsigned long long attr_done = 0ull;      // bit = device id: the attribute belongs to the device the kernel runs on
            const unsigned long long dev_bit = 1ull << (ctx->device & 63);
            if (!(attr_done & dev_bit)) {
                HITE_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&hit_segsort_kernel<HS_MEDIUM, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HS_MEDIUM * 8));
                attr_done |= dev_bit;
            }
            const int sbits = dbits + 1;
            const int gsm = n_cand < 8192 ? n_cand : 8192, gmd = n_cand < 2048 ? n_cand : 2048, glg = n_cand < 1024 ? n_cand : 1024;
            // the few large ranges (long chains, 32 KB of LDS each) run on a side stream beside the small ones; the medium class
            // takes a CU's whole LDS and follows on the main stream
            hipStream_t sside[HITE_AUX_STREAMS];
            hipEvent_t ev_fork, ev_join[HITE_AUX_STREAMS];
            CCHK(hite_aux_streams(ctx, 1, sside, &ev_fork, ev_join));
            HITE_CHECK(ctx, hipEventRecord(ev_fork, st));
            HITE_CHECK(ctx, hipStreamWaitEvent(sside[0], ev_fork, 0));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hit_segsort_kernel<0, 512>), dim3(glg), dim3(512), 0, sside[0], hs_counts + 3, hs_lists + (size_t)3 * n_cand, q_first, hit_off, hkey, so.k2, F.qbits, sbits);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hit_segsort_kernel<HS_SMALL, 256>), dim3(gsm), dim3(256), 2 * HS_SMALL * 8, st, hs_counts, hs_lists, q_first, hit_off, hkey, so.k2, F.qbits, sbits);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hit_segsort_kernel<HS_MID, 256>), dim3(gmd), dim3(256), 2 * HS_MID * 8, st, hs_counts + 1, hs_lists + (size_t)n_cand, q_first, hit_off, hkey, so.k2, F.qbits, sbits);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(hit_segsort_kernel<HS_MEDIUM, 512>), dim3(gmd), dim3(512), 2 * HS_MEDIUM * 8, st, hs_counts + 2, hs_lists + (size_t)2 * n_cand, q_first, hit_off, hkey, so.k2, F.qbits, sbits);
            HITE_CHECK(ctx, hipEventRecord(ev_join[0], sside[0]));
            HITE_CHECK(ctx, hipStreamWaitEvent(st, ev_join[0], 0));
            HITE_CHECK(ctx, hipGetLastError());
        } else if (F.qbits) {
            CCHK(sorter_sort_bits_swap(so, &hkey, &hval, nh, F.qbits, F.qbits + dbits));
            CCHK(sorter_sort_bits_swap(so, &hkey, &hval, nh, F.qbits + dbits, F.qbits + dbits + 1 + cbits));
        } else {
            CCHK(sorter_sort_bits_swap(so, &hkey, &hval, nh, 0, dbits));
            CCHK(sorter_sort_bits_swap(so, &hkey, &hval, nh, 33, 34 + cbits));
            F.hval = hval;
        }
    }
    hite_prof_end(ctx, tk_sh, st);
    // clusters
    CCHK(arena_alloc(ctx, A, (size_t)(nh + 1) * 4, &p)); flag = (int32_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(nh + 2) * 8, &p)); cid = (int64_t *)p;
    int64_t *bs2;
    CCHK(arena_alloc(ctx, A, (size_t)scan_tmp_elems(nh) * 8, &p)); bs2 = (int64_t *)p;
    int tk_cluster_flag_kernel = hite_prof_begin(ctx, "cluster_flag_kernel", st);
    hipLaunchKernelGGL(cluster_flag_kernel, CGRID(nh), 0, st, nh, hkey, F, ctx->d_contig_off, ctx->n_contigs, flag);
    hite_prof_end(ctx, tk_cluster_flag_kernel, st);
    CCHK(scan_excl_buf<int32_t>(ctx, bs2, flag, nh, cid, st));
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal, cid + nh, 8, hipMemcpyDeviceToDevice, st));
    CCHK(read_back(ctx, S, st, 1));
    const int64_t ncl = S->h_pin[0];
    S->last[2] = ncl;
    CCHK(arena_alloc(ctx, A, (size_t)(ncl + 1) * 8, &p)); c_lo = (unsigned long long *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(ncl + 1) * 8, &p)); c_hi = (unsigned long long *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(ncl + 2) * 4, &p)); c_first = (unsigned *)p;
    hipLaunchKernelGGL(cluster_first_kernel, CGRID(nh), 0, st, nh, flag, cid, c_first, ncl);
    int tk_cluster_acc_kernel = hite_prof_begin(ctx, "cluster_acc_kernel", st);
    hipLaunchKernelGGL(cluster_acc_init_kernel, CGRID(ncl), 0, st, ncl, c_lo, c_hi);
    hipLaunchKernelGGL(cluster_acc_kernel, CGRID(nh), 0, st, nh, hkey, F, flag, cid, c_lo, c_hi);
    hite_prof_end(ctx, tk_cluster_acc_kernel, st);
    // chains -> end extension -> copies.  The chain list is sized for the worst case (every chain needs >= 3 hits); its length
    // stays on the device (the kernels behind it run grid-stride loops up to the count they read there)
    const unsigned long long chcap = (unsigned long long)(nh / C_MINANCH) + 1;
    unsigned *chain_list; int32_t *x_i, *x_t;
    CCHK(arena_alloc(ctx, A, (size_t)chcap * 4, &p)); chain_list = (unsigned *)p;
    CCHK(arena_alloc(ctx, A, (size_t)chcap * 8, &p)); x_i = (int32_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)chcap * 8, &p)); x_t = (int32_t *)p;
    HITE_CHECK(ctx, hipMemsetAsync(S->d_scal, 0, 64, st));
    unsigned long long *d_nchain = (unsigned long long *)(S->d_scal + 4);     // [0] long chains, [1] short chains, [2] task queue
    int tk_ext = hite_prof_begin(ctx, "chain_extend_kernel", st);
    {
        int64_t lblocks = (ncl + 256 * CL_ITEMS - 1) / (256 * CL_ITEMS);
        if (lblocks > 16384) lblocks = 16384;
        if (lblocks < 1) lblocks = 1;
        hipLaunchKernelGGL(chain_list_kernel, dim3((unsigned)lblocks), dim3(256), 0, st, ncl, hkey, F, c_first, c_lo, c_hi, d_cand_off, chain_list, chcap, d_nchain);
    }
    {
        // persistent lanes: every lane takes (chain, end) tasks from the queue until it is empty
        unsigned long long want_blocks = (2ull * chcap + 255ull) / 256ull;
        const unsigned eblocks = (unsigned)(want_blocks < 2048ull ? (want_blocks ? want_blocks : 1ull) : 2048ull);
        hipLaunchKernelGGL(chain_extend_kernel, dim3(eblocks), dim3(256), 0, st, d_nchain, chain_list, hkey, F, c_first, c_lo, c_hi, d_cand,
                           d_cand_off, ctx->d_bases, ctx->d_nmask, ctx->d_contig_off, ctx->n_contigs, chcap, d_nchain + 2, x_i, x_t);
    }
    hite_prof_end(ctx, tk_ext, st);
    int32_t *r_contig, *r_anch;
    int64_t *r_s1, *r_e1;
    uint8_t *r_minus;
    uint32_t *r_clip;
    // (records: at most one per chain)
    CCHK(arena_alloc(ctx, A, (size_t)(chcap + 1) * 8, &p)); ckey = (unsigned long long *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(chcap + 1) * 4, &p)); cval = (unsigned *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(chcap + 1) * 4, &p)); r_contig = (int32_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(chcap + 1) * 8, &p)); r_s1 = (int64_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(chcap + 1) * 8, &p)); r_e1 = (int64_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(chcap + 16), &p)); r_minus = (uint8_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(chcap + 1) * 4, &p)); r_anch = (int32_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(chcap + 1) * 4, &p)); r_clip = (uint32_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(n_cand + 1) * 4, &p)); per_cand = (int32_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(n_cand + 1) * 4, &p)); per_cand300 = (int32_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(n_cand + 2) * 8, &p)); cstart = (int64_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(n_cand + 2) * 8, &p)); ofirst = (int64_t *)p;
    int32_t *fill;
    CCHK(arena_alloc(ctx, A, (size_t)(n_cand + 1) * 4, &p)); fill = (int32_t *)p;
    HITE_CHECK(ctx, hipMemsetAsync(per_cand, 0, (size_t)(n_cand + 1) * 4, st));
    HITE_CHECK(ctx, hipMemsetAsync(fill, 0, (size_t)(n_cand + 1) * 4, st));
    int64_t *bs3;
    CCHK(arena_alloc(ctx, A, (size_t)scan_tmp_elems(n_cand) * 8, &p)); bs3 = (int64_t *)p;
    int tk_cluster_copy_kernel = hite_prof_begin(ctx, "chain_copy_kernel", st);
    {
        unsigned long long want_blocks = (chcap + 255ull) / 256ull;
        const unsigned cblocks = (unsigned)(want_blocks < 8192ull ? (want_blocks ? want_blocks : 1ull) : 8192ull);
        hipLaunchKernelGGL(chain_copy_kernel<false>, dim3(cblocks), dim3(256), 0, st, d_nchain, chcap, chain_list, x_i, x_t, hkey, F, c_first, c_lo,
                           c_hi, d_cand_off, ctx->d_contig_off, ctx->n_contigs, ckey, cval, r_contig, r_s1, r_e1, r_minus, r_anch, r_clip, per_cand,
                           (const int64_t *)nullptr, copy_interval_mode(ctx));
        CCHK(scan_excl_buf<int32_t>(ctx, bs3, per_cand, n_cand, cstart, st));
        hipLaunchKernelGGL(chain_copy_kernel<true>, dim3(cblocks), dim3(256), 0, st, d_nchain, chcap, chain_list, x_i, x_t, hkey, F, c_first, c_lo,
                           c_hi, d_cand_off, ctx->d_contig_off, ctx->n_contigs, ckey, cval, r_contig, r_s1, r_e1, r_minus, r_anch, r_clip, fill,
                           (const int64_t *)cstart, copy_interval_mode(ctx));
    }
    hite_prof_end(ctx, tk_cluster_copy_kernel, st);
    hipLaunchKernelGGL(cap300_kernel, CGRID((int64_t)n_cand), 0, st, n_cand, per_cand, per_cand300);
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal, cstart + n_cand, 8, hipMemcpyDeviceToDevice, st));
    CCHK(scan_excl_buf<int32_t>(ctx, bs3, per_cand300, n_cand, ofirst, st));
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal + 1, ofirst + n_cand, 8, hipMemcpyDeviceToDevice, st));
    CCHK(read_back(ctx, S, st, 8));
    const int64_t ncp = S->h_pin[0], nout = S->h_pin[1];
    S->last[4] = S->h_pin[4]; S->last[5] = S->h_pin[5]; S->last[6] = S->h_pin[7];
    *n_copies = nout;
    S->last[3] = ncp;
    hipLaunchKernelGGL(i64_to_i32_kernel, CGRID((int64_t)n_cand + 1), 0, st, (int64_t)n_cand + 1, ofirst, ofirst32);
    if (ncp == 0) return HITE_OK;
    Sorter so2;
    CCHK(sorter_from_arena(so2, ctx, A, st, ncp));
    int tk_sc = hite_prof_begin(ctx, "radix_sort_copies", st);
    CCHK(sorter_sort(so2, ckey, cval, ncp, 64));
    hite_prof_end(ctx, tk_sc, st);
    int32_t *o_contig, *o_anch;
    int64_t *o_s1, *o_e1;
    uint8_t *o_minus;
    CCHK(arena_alloc(ctx, S->out, (size_t)(nout + 1) * 4, &p)); o_contig = (int32_t *)p;
    CCHK(arena_alloc(ctx, S->out, (size_t)(nout + 1) * 8, &p)); o_s1 = (int64_t *)p;
    CCHK(arena_alloc(ctx, S->out, (size_t)(nout + 1) * 8, &p)); o_e1 = (int64_t *)p;
    CCHK(arena_alloc(ctx, S->out, (size_t)(nout + 16), &p)); o_minus = (uint8_t *)p;
    CCHK(arena_alloc(ctx, S->out, (size_t)(nout + 1) * 4, &p)); o_anch = (int32_t *)p;
    uint32_t *o_clip;
    CCHK(arena_alloc(ctx, S->out, (size_t)(nout + 1) * 4, &p)); o_clip = (uint32_t *)p;
    hipLaunchKernelGGL(emit_copies_kernel, CGRID(ncp), 0, st, ncp, ckey, cval, cstart, ofirst, r_contig, r_s1, r_e1, r_minus, r_anch, r_clip,
                       o_contig, o_s1, o_e1, o_minus, o_anch, o_clip);
    S->out_clip = o_clip; S->out_n = nout;
    HITE_CHECK(ctx, hipGetLastError());
    *d_contig = o_contig; *d_start1 = o_s1; *d_end1 = o_e1; *d_minus = o_minus; *d_anchors = o_anch;
    // if the temporaries grew into several chunks during this call, merge them NOW (they are dead; the copy table lives in
    // S->out): the next call then starts on one chunk and performs no hipMalloc
    CCHK(arena_reset(ctx, S->arena, true));
    return HITE_OK;
}

// host-buffer wrapper: builds the index if *state_io is NULL, uploads the candidates, downloads the table
extern "C" int hite_find_copies_dev(hite_ctx *ctx, void *state, int32_t n_cand, const uint8_t *d_cand, const int64_t *d_cand_off,
                                    int64_t cand_bytes, int32_t **d_copy_first, int64_t *n_copies, int32_t **d_contig,
                                    int64_t **d_start1, int64_t **d_end1, uint8_t **d_minus, int32_t **d_anchors, void *stream) {
    return find_copies_impl(ctx, state, n_cand, d_cand, d_cand_off, cand_bytes, d_copy_first, n_copies, d_contig, d_start1, d_end1, d_minus,
                            d_anchors, stream, false);
}

// hite_find_copies_dev for a caller that needs the index for THESE candidates only (masking the genome with a TE library before
// the index proper is built on the masked genome: stage 3.1).  The genome pass keeps just the minimizers whose hash one of the
// candidates' minimizers looks up -- every entry occ_kernel / hit_kernel would read, with the same run lengths (so the C_MAXOCC
// rule sees the same counts) and in the same order -- and the hash sort and the directory run on those few entries instead of
// ~0.18 G of them.  The copy table is identical to the full index's; the handle is left flagged `restricted`, and every other
// use of it (hite_find_copies[_dev], hite_seed_allvsall[_dev]) rebuilds the full index first.
extern "C" int hite_find_copies_restricted_dev(hite_ctx *ctx, void **state_io, int32_t n_cand, const uint8_t *d_cand,
                                               const int64_t *d_cand_off, int64_t cand_bytes, int32_t **d_copy_first, int64_t *n_copies,
                                               int32_t **d_contig, int64_t **d_start1, int64_t **d_end1, uint8_t **d_minus,
                                               int32_t **d_anchors, void *stream) {
    if (!ctx || !ctx->d_bases || !state_io || n_cand < 0 || n_cand >= (1 << 19) || !n_copies) return HITE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    CopyState *S;
    CCHK(copy_state_get(ctx, state_io, &S));
    CCHK(arena_reset(ctx, S->arena, true));
    Arena &A = S->arena;
    void *p;
    unsigned *tab = nullptr, *bits = nullptr; unsigned mask = 1023, bmask = 65535;
    const int64_t cb = cand_bytes > 0 ? cand_bytes : 0;
    int tk = hite_prof_begin(ctx, "restricted_set", st);
    if (n_cand > 0 && cb > 0) {
        int32_t *q_cnt; unsigned *r_pos, *r_hs; int64_t *q_first, *qbs;
        CCHK(arena_alloc(ctx, A, (size_t)(n_cand + 1) * 4, &p)); q_cnt = (int32_t *)p;
        CCHK(arena_alloc(ctx, A, (size_t)(n_cand + 2) * 8, &p)); q_first = (int64_t *)p;
        CCHK(arena_alloc(ctx, A, (size_t)scan_tmp_elems(n_cand) * 8, &p)); qbs = (int64_t *)p;
        CCHK(arena_alloc(ctx, A, (size_t)(cb + 64) * 4, &p)); r_pos = (unsigned *)p;
        CCHK(arena_alloc(ctx, A, (size_t)(cb + 64) * 4, &p)); r_hs = (unsigned *)p;
        HITE_CHECK(ctx, hipMemsetAsync(S->d_scal, 0, 64, st));
        hipLaunchKernelGGL(cand_minimizer_kernel, dim3(n_cand < 65536 ? n_cand : 65536), dim3(256), 0, st, n_cand, d_cand, d_cand_off, r_pos,
                           r_hs, q_cnt, (unsigned long long *)(S->d_scal + 1));
        CCHK(scan_excl_buf<int32_t>(ctx, qbs, q_cnt, n_cand, q_first, st));
        HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal, q_first + n_cand, 8, hipMemcpyDeviceToDevice, st));
        CCHK(read_back(ctx, S, st, 1));
        const int64_t nq = S->h_pin[0];
        while ((int64_t)mask + 1 < 4 * nq && mask < 0x7fffffffu) mask = mask * 2 + 1;     // load <= 1/4
        if ((int64_t)mask + 1 < 2 * nq) return HITE_ECAP;
        while ((int64_t)bmask + 1 < 16 * nq && bmask < 0x00ffffffu) bmask = bmask * 2 + 1;     // <= 2 MiB
        CCHK(arena_alloc(ctx, A, ((size_t)mask + 1) * 4, &p)); tab = (unsigned *)p;
        CCHK(arena_alloc(ctx, A, ((size_t)bmask + 1) / 8, &p)); bits = (unsigned *)p;
        HITE_CHECK(ctx, hipMemsetAsync(tab, 0xff, ((size_t)mask + 1) * 4, st));
        HITE_CHECK(ctx, hipMemsetAsync(bits, 0, ((size_t)bmask + 1) / 8, st));
        hipLaunchKernelGGL(hset_insert_kernel, dim3((n_cand + 3) / 4 < 8192 ? (n_cand + 3) / 4 : 8192), dim3(256), 0, st, n_cand, d_cand_off,
                           r_hs, q_cnt, tab, mask, bits, bmask);
        HITE_CHECK(ctx, hipGetLastError());
    } else {
        CCHK(arena_alloc(ctx, A, ((size_t)mask + 1) * 4, &p)); tab = (unsigned *)p;
        CCHK(arena_alloc(ctx, A, ((size_t)bmask + 1) / 8, &p)); bits = (unsigned *)p;
        HITE_CHECK(ctx, hipMemsetAsync(tab, 0xff, ((size_t)mask + 1) * 4, st));
        HITE_CHECK(ctx, hipMemsetAsync(bits, 0, ((size_t)bmask + 1) / 8, st));
    }
    hite_prof_end(ctx, tk, st);
    CCHK(index_build_impl(ctx, state_io, st, HSet{tab, bits, mask, bmask}));     // (its own arena; ends with a stream synchronise: `tab` is free to go)
    return find_copies_impl(ctx, *state_io, n_cand, d_cand, d_cand_off, cand_bytes, d_copy_first, n_copies, d_contig, d_start1, d_end1,
                            d_minus, d_anchors, stream, true);
}

static int find_copies_host(hite_ctx *ctx, void **state_io, int32_t n_cand, const uint8_t *cand, const int64_t *cand_off,
                            int64_t cap, int32_t *copy_first, int32_t *contig, int64_t *start1, int64_t *end1, uint8_t *minus,
                            int32_t *anchors, int64_t *n_out, bool restricted) {
    if (!ctx || !state_io || !cand || !cand_off || !copy_first || !n_out) return HITE_EINVAL;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    if (!*state_io && !restricted) CCHK(hite_copy_index_build(ctx, state_io, nullptr));
    uint8_t *dc = nullptr;
    int64_t *dco = nullptr;
    int64_t bytes = cand_off[n_cand];
    DevTmp tmp;
    HITE_CHECK(ctx, tmp.alloc((void **)&dc, (size_t)bytes + 64));
    HITE_CHECK(ctx, tmp.alloc((void **)&dco, (size_t)(n_cand + 1) * 8));
    HITE_CHECK(ctx, hipMemcpy(dc, cand, (size_t)bytes, hipMemcpyHostToDevice));
    HITE_CHECK(ctx, hipMemcpy(dco, cand_off, (size_t)(n_cand + 1) * 8, hipMemcpyHostToDevice));
    int32_t *dcf, *dct, *dan;
    int64_t *ds1, *de1;
    uint8_t *dmn;
    int64_t n = 0;
    int rc = restricted ? hite_find_copies_restricted_dev(ctx, state_io, n_cand, dc, dco, bytes, &dcf, &n, &dct, &ds1, &de1, &dmn, &dan, nullptr)
                        : hite_find_copies_dev(ctx, *state_io, n_cand, dc, dco, bytes, &dcf, &n, &dct, &ds1, &de1, &dmn, &dan, nullptr);
    if (rc == HITE_OK) {
        *n_out = n;
        if (hipDeviceSynchronize() != hipSuccess) rc = HITE_EHIP;
        else if (n > cap) rc = HITE_ECAP;
        else {
            hipError_t e = hipMemcpy(copy_first, dcf, (size_t)(n_cand + 1) * 4, hipMemcpyDeviceToHost);
            if (n > 0) {
                if (e == hipSuccess) e = hipMemcpy(contig, dct, (size_t)n * 4, hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipMemcpy(start1, ds1, (size_t)n * 8, hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipMemcpy(end1, de1, (size_t)n * 8, hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipMemcpy(minus, dmn, (size_t)n, hipMemcpyDeviceToHost);
                if (e == hipSuccess && anchors) e = hipMemcpy(anchors, dan, (size_t)n * 4, hipMemcpyDeviceToHost);
            }
            if (e != hipSuccess) rc = HITE_EHIP;
        }
    }
    return rc;
}

// sizes of the last hite_find_copies[_dev] call on this index: {candidate minimizers, index hits, diagonal clusters, copies before the cap}
extern "C" int hite_find_copies(hite_ctx *ctx, void **state_io, int32_t n_cand, const uint8_t *cand, const int64_t *cand_off,
                                int64_t cap, int32_t *copy_first, int32_t *contig, int64_t *start1, int64_t *end1, uint8_t *minus,
                                int32_t *anchors, int64_t *n_out) {
    return find_copies_host(ctx, state_io, n_cand, cand, cand_off, cap, copy_first, contig, start1, end1, minus, anchors, n_out, false);
}
extern "C" int hite_find_copies_restricted(hite_ctx *ctx, void **state_io, int32_t n_cand, const uint8_t *cand, const int64_t *cand_off,
                                           int64_t cap, int32_t *copy_first, int32_t *contig, int64_t *start1, int64_t *end1,
                                           uint8_t *minus, int32_t *anchors, int64_t *n_out) {
    return find_copies_host(ctx, state_io, n_cand, cand, cand_off, cap, copy_first, contig, start1, end1, minus, anchors, n_out, true);
}
/* the clipped candidate bases of the last hite_find_copies[_dev] / _restricted call's records (see hite_gpu.h) */
extern "C" int hite_copy_clips_dev(void *state, const uint32_t **d_clip, int64_t *n) {
    CopyState *S = (CopyState *)state;
    if (!S || !d_clip) return HITE_EINVAL;
    *d_clip = S->out_clip;
    if (n) *n = S->out_n;
    return HITE_OK;
}
extern "C" int hite_copy_clips(void *state, int64_t cap, uint32_t *clip) {
    CopyState *S = (CopyState *)state;
    if (!S || (!clip && cap > 0)) return HITE_EINVAL;
    if (S->out_n > cap) return HITE_ECAP;
    if (S->out_n > 0 && S->out_clip && hipMemcpy(clip, S->out_clip, (size_t)S->out_n * 4, hipMemcpyDeviceToHost) != hipSuccess) return HITE_EHIP;
    return HITE_OK;
}
extern "C" int hite_copy_stats(void *state, int64_t out[4]) {
    CopyState *S = (CopyState *)state;
    if (!S || !out) return HITE_EINVAL;
    for (int i = 0; i < 4; i++) out[i] = S->last[i];
    return HITE_OK;
}
// + {chains with an end of >= 384 bases to extend, the other chains, dynamic-programming columns of the end extension, 0}
extern "C" int hite_copy_stats_ext(void *state, int64_t out[8]) {
    CopyState *S = (CopyState *)state;
    if (!S || !out) return HITE_EINVAL;
    for (int i = 0; i < 8; i++) out[i] = S->last[i];
    return HITE_OK;
}

// =====================================================================================================
// All-vs-all seeding (stage 3.1): this build's GPU-native stage where the reference runs `blastn` of every
// 1 Mbp segment file against every file (process_blast_alignments / sequence2sequenceBlastn,
// /root/reference/module/Util.py:4724-4780, 4068-4091).  Definition: header of the twin in
// oracle/hite_oracle_copies.c (orc_seed_allvsall), HIP == twin record for record.
//   minimizers in position order (the index build carries every entry's rank along the genome) -> run of equal hs >> 1 per seed ->
//   anchor counts -> scan -> anchors (key = strand | diagonal, value = query position) -> stable radix sort on
//   (strand, diagonal >> 6) -> cluster flags -> HSPs from the first / last anchor of each cluster -> cut at the
//   1 Mbp segment borders -> stable sort by (query segment, subject segment).
// Everything is sort / scan / segment work on 12-byte records: HBM streaming.
// =====================================================================================================
#define SEED_MAXOCC 1000
#define SEED_GAP 300
#define SEED_MINANCH 3
#define SEED_MINSPAN 60
#define SEED_MINPIECE 20

__global__ void seed_runflag_kernel(int64_t M, const unsigned *__restrict__ idx_hs, int32_t *__restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) flag[i] = (i == 0 || (idx_hs[i] >> 1) != (idx_hs[i - 1] >> 1)) ? 1 : 0;
}
// first entry of every run (rid = EXCLUSIVE scan of the flags: a flagged entry opens run rid[i])
__global__ void seed_runfirst_kernel(int64_t M, const int32_t *__restrict__ flag, const int64_t *__restrict__ rid, unsigned *__restrict__ run_first) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M && flag[i]) run_first[rid[i]] = (unsigned)i;
}
// Sharding of the all-vs-all stage over ranks (hite_seed_shard; hite_amd/dist.py): a rank keeps the anchors whose sort key
// (strand | diagonal) lies in its range.  Clusters never straddle two ranges (a cluster lives inside one 64-diagonal bucket and
// the range edges are multiples of 64), so the clusters -- and the HSPs -- of the ranks together are exactly those of one
// unsharded run, in the same order when the ranks are concatenated.  lin = strand * 2 G + diagonal in [0, 4 G).
struct SeedShard { unsigned long long lo, hi; long long twoG; int sharded; };      // lin in [lo, hi) -- possibly empty -- when sharded
__device__ __forceinline__ bool seed_owned(const SeedShard &sh, unsigned long long rel, unsigned long long d) {
    if (!sh.sharded) return true;
    const unsigned long long lin = (rel ? (unsigned long long)sh.twoG : 0ull) + d;
    return lin >= sh.lo && lin < sh.hi;
}
// One thread per INDEX ENTRY (hash order: its run's bounds are a coalesced read): partners = run size - 1 (0 when the run is too
// large); sharded: the partners whose anchor this rank owns.  The seed's record -- index entry, place in its run, run size, partner
// count -- goes to the seed's place in POSITION order (idx_t, left behind by the index build): the only scattered access; the
// stages behind read it in order.
// SPLIT (round 6, large genomes): the record stays in hash order as key = index entry << 32 | rank, value = the packed word, one radix pass
// on the rank's top ten bits groups the records by stretches of the position order, and seed_place_kernel scatters inside such a stretch
// -- a window of the record array that an L2 holds, so that memory sees whole lines (the direct scatter: one 32-byte sector per 8-byte
// record, 5.7 GB written for 1.45 GB of records).
template <bool SPLIT>
__global__ void seed_count_kernel(int64_t M, int64_t G, const int32_t *__restrict__ rflag, const int64_t *__restrict__ rid,
                                  const unsigned *__restrict__ run_first, const unsigned long long *__restrict__ idx_key,
                                  const unsigned *__restrict__ idx_t, SeedShard sh, uint2 *__restrict__ srec,
                                  unsigned long long *__restrict__ pk, unsigned *__restrict__ pv) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int64_t nrun = rid[M];
    const int64_t r = rid[i] + rflag[i] - 1;
    const unsigned lo = run_first[r], hi = r + 1 < nrun ? run_first[r + 1] : (unsigned)M;
    const unsigned occ = hi - lo;
    const unsigned t = idx_t[i];
    int c = 0;
    if (occ <= SEED_MAXOCC) {
        if (!sh.sharded) c = (int)(occ - 1);
        else {
            const unsigned long long ki = idx_key[i];
            const unsigned hq = (unsigned)(ki >> 32);
            const long long pi = (long long)(unsigned)ki;
            for (unsigned j = lo; j < hi; j++) {
                if (j == (unsigned)i) continue;
                const unsigned long long kj = idx_key[j];
                const unsigned long long rel = (hq ^ (unsigned)(kj >> 32)) & 1u;
                const long long pj = (long long)(unsigned)kj;
                c += seed_owned(sh, rel, rel ? (unsigned long long)(pi + pj) : (unsigned long long)(pj - pi + G));
            }
        }
    }
    // ONE 8-byte scatter per seed: index entry | place in its run (10 bits) | run size (10) | partners (10)
    const unsigned y = occ <= SEED_MAXOCC ? (((unsigned)i - lo) | (occ << 10) | ((unsigned)c << 20)) : 0u;
    if (SPLIT) { pk[i] = ((unsigned long long)(unsigned)i << 32) | (unsigned long long)t; pv[i] = y; }
    else srec[t] = make_uint2((unsigned)i, y);
}
// The same records without the run arrays (late round 6): flag array, scan of the flags and first-of-run array were three passes over
// the index (2.1 ms per Gbp) to tell an entry where its run begins and ends.  A workgroup owns SC_TILE consecutive entries, puts the run
// starts of its tile and of SC_HALO entries either side into a bit map in LDS (a ballot per 64 entries) and finds the start at or before
// an entry and the next one behind it with two bit searches.  A run that begins or ends outside the halo is longer than SEED_MAXOCC (the
// halo is larger), which is all the record needs to know about it.
#define SC_TILE 2048
#define SC_HALO 1024
#define SC_WORDS ((SC_TILE + 2 * SC_HALO) / 64 + 1)
template <bool SPLIT>
__global__ void __launch_bounds__(256) seed_count2_kernel(int64_t M, int64_t G, const unsigned *__restrict__ idx_hs,
                                                          const unsigned long long *__restrict__ idx_key, const unsigned *__restrict__ idx_t,
                                                          SeedShard sh, uint2 *__restrict__ srec, unsigned long long *__restrict__ pk,
                                                          unsigned *__restrict__ pv) {
    static_assert(SC_HALO > SEED_MAXOCC + 1, "the halo must hold a whole countable run");
    __shared__ unsigned long long s_bits[SC_WORDS + 3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * SC_TILE, P0 = t0 - SC_HALO;
    // bit x of the map = entry P0 + x opens a run (entry 0 does; entry M, the end of the index, counts as one)
    for (int k = 0; k * 4 < SC_WORDS; k++) {
        const int x = k * 256 + (int)threadIdx.x;
        const int64_t e = P0 + x;
        bool f = false;
        if (e >= 0 && e <= M) f = e == 0 || e == M || (idx_hs[e] >> 1) != (idx_hs[e - 1] >> 1);
        const unsigned long long bal = __ballot(f);
        if (lane == 0 && k * 4 + w < SC_WORDS) s_bits[k * 4 + w] = bal;
    }
    __syncthreads();
#pragma unroll 2
    for (int q = 0; q < SC_TILE / 256; q++) {
        const int64_t i = t0 + q * 256 + threadIdx.x;
        if (i >= M) continue;
        const int b = SC_HALO + q * 256 + (int)threadIdx.x;
        // the run start at or before b, the next start behind b (each at most SC_HALO away, else: not found)
        int lo_b = -1, hi_b = -1;
        {
            int wi = b >> 6;
            unsigned long long m = s_bits[wi] & (0xffffffffffffffffull >> (63 - (b & 63)));
            const int wmin = (b - SC_HALO) >> 6;
            while (m == 0ull && wi > wmin) { wi--; m = s_bits[wi]; }
            if (m != 0ull) lo_b = wi * 64 + 63 - __clzll((long long)m);
        }
        {
            const int b1 = b + 1;
            int wi = b1 >> 6;
            unsigned long long m = s_bits[wi] & (0xffffffffffffffffull << (b1 & 63));
            const int wmax = (b + SC_HALO) >> 6 < SC_WORDS - 1 ? (b + SC_HALO) >> 6 : SC_WORDS - 1;
            while (m == 0ull && wi < wmax) { wi++; m = s_bits[wi]; }
            if (m != 0ull) hi_b = wi * 64 + __ffsll((long long)m) - 1;
        }
        const bool countable = lo_b >= 0 && hi_b >= 0 && hi_b - lo_b <= SEED_MAXOCC;
        const unsigned lo = (unsigned)(P0 + lo_b), hi = (unsigned)(P0 + hi_b);       // (used only when countable)
        const unsigned occ = hi - lo;
        const unsigned t = idx_t[i];
        int c = 0;
        if (countable) {
            if (!sh.sharded) c = (int)(occ - 1);
            else {
                const unsigned long long ki = idx_key[i];
                const unsigned hq = (unsigned)(ki >> 32);
                const long long pi = (long long)(unsigned)ki;
                for (unsigned j = lo; j < hi; j++) {
                    if (j == (unsigned)i) continue;
                    const unsigned long long kj = idx_key[j];
                    const unsigned long long rel = (hq ^ (unsigned)(kj >> 32)) & 1u;
                    const long long pj = (long long)(unsigned)kj;
                    c += seed_owned(sh, rel, rel ? (unsigned long long)(pi + pj) : (unsigned long long)(pj - pi + G));
                }
            }
        }
        const unsigned y = countable ? (((unsigned)i - lo) | (occ << 10) | ((unsigned)c << 20)) : 0u;
        if (SPLIT) { pk[i] = ((unsigned long long)(unsigned)i << 32) | (unsigned long long)t; pv[i] = y; }
        else srec[t] = make_uint2((unsigned)i, y);
    }
}
// the grouped records to their places (and the partner counts beside them); the workgroups of a chiplet take consecutive tiles (as
// rs_scatter_staged_kernel): the window they write into is theirs
#define SPL_ITEMS 8
__global__ void __launch_bounds__(256) seed_place_kernel(int64_t M, int nblocks, const unsigned long long *__restrict__ pk,
                                                         const unsigned *__restrict__ pv, uint2 *__restrict__ srec, int32_t *__restrict__ cnt) {
    const int x = (int)(blockIdx.x & 7u), bi = (int)(blockIdx.x >> 3), q8 = nblocks >> 3, rr = nblocks & 7;
    const int64_t base = (int64_t)(x * q8 + (x < rr ? x : rr) + bi) * (256 * SPL_ITEMS);
    unsigned long long k[SPL_ITEMS]; unsigned y[SPL_ITEMS];
#pragma unroll
    for (int q = 0; q < SPL_ITEMS; q++) {
        const int64_t e = base + q * 256 + threadIdx.x;
        k[q] = pk[e < M ? e : M - 1]; y[q] = pv[e < M ? e : M - 1];
    }
#pragma unroll
    for (int q = 0; q < SPL_ITEMS; q++) {
        const int64_t e = base + q * 256 + threadIdx.x;
        if (e < M) { const unsigned t = (unsigned)k[q]; srec[t] = make_uint2((unsigned)(k[q] >> 32), y[q]); if (cnt) cnt[t] = (int32_t)(y[q] >> 20); }
    }
}
__global__ void seed_cnt_kernel(int64_t M, const uint2 *__restrict__ srec, int32_t *__restrict__ cnt) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < M) cnt[t] = (int32_t)(srec[t].y >> 20);
}
// Anchor records.  Unpacked (any genome a context holds): key = strand << 34 | diagonal, value = query position.  Packed
// (PK; genomes of at most 2^30 bases, i.e. every chunk of the reference's own flow -- chunk_size 400 MB, main.py:23 -- and the
// 1 Gbp of the bench): key = strand << 61 | diagonal << 30 | query position and no value array: 8 instead of 12 bytes through
// the three sort passes and every kernel behind them.  Both sort (stably) on (strand, diagonal >> 6): the same order.
#define SEED_PACK_MAXG (1ll << 30)
template <bool PK> __device__ __forceinline__ unsigned long long anc_make(unsigned long long rel, unsigned long long d, unsigned pi) {
    return PK ? (rel << 61) | (d << 30) | (unsigned long long)pi : (rel << 34) | d;
}
template <bool PK> __device__ __forceinline__ unsigned long long anc_sd(unsigned long long key) {   // strand << 34 | diagonal
    return PK ? ((key >> 61) << 34) | ((key >> 30) & 0x7fffffffull) : key;
}
template <bool PK> __device__ __forceinline__ unsigned anc_pi(unsigned long long key, const unsigned *__restrict__ aval, int64_t i) {
    return PK ? (unsigned)(key & 0x3fffffffull) : aval[i];
}
// a rank's share (hite_seed_shard): one thread per seed walks its run and keeps the partners whose anchor the rank owns
template <bool PK>
__global__ void seed_anchor_kernel(int64_t M, int64_t G, const uint2 *__restrict__ srec, const unsigned long long *__restrict__ idx_key,
                                   const int32_t *__restrict__ cnt, const int64_t *__restrict__ aoff, SeedShard sh,
                                   unsigned long long *__restrict__ akey, unsigned *__restrict__ aval) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M || cnt[t] <= 0) return;
    const uint2 sr = srec[t];
    const unsigned i = sr.x, lo = i - (sr.y & 1023u), hi = lo + ((sr.y >> 10) & 1023u);
    const unsigned long long ki = idx_key[i];
    const unsigned hq = (unsigned)(ki >> 32);
    const long long pi = (long long)(unsigned)ki;
    int64_t o = aoff[t];
    for (unsigned j = lo; j < hi; j++) {
        if (j == i) continue;
        const unsigned long long kj = idx_key[j];
        const unsigned long long rel = (hq ^ (unsigned)(kj >> 32)) & 1u;
        const long long pj = (long long)(unsigned)kj;
        const unsigned long long d = rel ? (unsigned long long)(pi + pj) : (unsigned long long)(pj - pi + G);
        if (!seed_owned(sh, rel, d)) continue;
        akey[o] = anc_make<PK>(rel, d, (unsigned)pi);
        if (!PK) aval[o] = (unsigned)pi;
        o++;
    }
}
// the unsharded stage: the anchors of 64 consecutive seeds (position order) are ONE contiguous output range, which the
// wavefront that owns the seeds fills 64 slots at a time -- slot o belongs to the seed whose range holds it (a search over the
// 64 range starts, kept in LDS) and is its (o - start)-th partner, the seed itself skipped.  Reads of a run and the writes are
// coalesced (the thread-per-seed form above writes 64 separate streams: PMC showed 5x the anchor bytes on both sides).
// Same anchors in the same order.
template <bool PK>
__global__ void __launch_bounds__(256) seed_anchor_coop_kernel(int64_t M, int64_t G, const uint2 *__restrict__ srec,
                                                               const unsigned long long *__restrict__ idx_key,
                                                               const int64_t *__restrict__ aoff, unsigned long long *__restrict__ akey,
                                                               unsigned *__restrict__ aval) {
    __shared__ unsigned s_start[4][65], s_i[4][64], s_lo[4][64], s_hq[4][64], s_pi[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t t0 = ((int64_t)blockIdx.x * 4 + w) * 64;
    if (t0 >= M) return;
    const int64_t t = t0 + lane;
    const int64_t base = aoff[t0];
    const int64_t tend = t0 + 64 < M ? t0 + 64 : M;
    const unsigned total = (unsigned)(aoff[tend] - base);
    if (total == 0) return;
    {
        unsigned i = 0, lo = 0, hq = 0, pi = 0, start = total;
        if (t < M) {
            start = (unsigned)(aoff[t] - base);
            if (aoff[t + 1] - aoff[t] > 0) {       // (a seed without partners -- run of one, or too large -- owns no slot)
                const uint2 sr = srec[t];
                i = sr.x;
                lo = i - (sr.y & 1023u);
                const unsigned long long ki = idx_key[i];
                hq = (unsigned)(ki >> 32);
                pi = (unsigned)ki;
            }
        }
        s_start[w][lane] = start; s_i[w][lane] = i; s_lo[w][lane] = lo; s_hq[w][lane] = hq; s_pi[w][lane] = pi;
        if (lane == 0) s_start[w][64] = total;
    }
    __builtin_amdgcn_wave_barrier();
    for (unsigned o = lane; o < total; o += 64) {
        // the last seed whose range starts at or before o (empty ranges share their start with the next seed: the LAST one wins,
        // and it is the one that is not empty because o < total)
        int a = 0, b = 64;
        while (b - a > 1) { const int m = (a + b) >> 1; if (s_start[w][m] <= o) a = m; else b = m; }
        const unsigned i = s_i[w][a], hq = s_hq[w][a];
        const long long pi = s_pi[w][a];
        unsigned j = s_lo[w][a] + (o - s_start[w][a]);
        if (j >= i) j++;
        const unsigned long long kj = idx_key[j];
        const unsigned long long rel = (hq ^ (unsigned)(kj >> 32)) & 1u;
        const long long pj = (long long)(unsigned)kj;
        const unsigned long long d = rel ? (unsigned long long)(pi + pj) : (unsigned long long)(pj - pi + G);
        akey[base + o] = anc_make<PK>(rel, d, (unsigned)pi);
        if (!PK) aval[base + o] = (unsigned)pi;
    }
}
__device__ __forceinline__ long long seed_pj(unsigned long long key, unsigned pi, long long G) {
    const long long d = (long long)(key & 0x3ffffffffull);
    return (key >> 34) ? d - (long long)pi : d - G + (long long)pi;
}
// does anchor i open a cluster?  (different (strand, diagonal >> 6) bucket than the anchor before it, more than SEED_GAP further
// on, or either position in another contig)
// (round 6: a thread walks consecutive anchors, whose positions mostly stay inside one contig -- it keeps the bounds of the contig it
// looked up last for the query and for the subject side and searches the contig table only when a position leaves them: two binary
// searches per anchor pair were a third of the kernel)
struct SeedContigs { long long qlo, qhi, slo, shi; };       // [lo, hi) of the contigs looked up last; hi <= lo: none yet
template <bool PK>
__device__ __forceinline__ int seed_opens(int64_t i, unsigned long long ka, unsigned long long kb, const unsigned *__restrict__ aval,
                                          int64_t G, const int64_t *__restrict__ coff, int nc, SeedContigs &cb) {
    if (i == 0) return 1;
    const unsigned long long a = anc_sd<PK>(ka), b = anc_sd<PK>(kb);
    const unsigned pa = anc_pi<PK>(ka, aval, i - 1), pb = anc_pi<PK>(kb, aval, i);
    if ((a >> 6) != (b >> 6) || (long long)pb - (long long)pa > SEED_GAP) return 1;
    // same bucket, pa <= pb (position order inside a bucket): the two query positions lie in one contig iff pb is below the
    // end of pa's; likewise the two subject positions
    if (!((long long)pa >= cb.qlo && (long long)pa < cb.qhi)) { const int ca = contig_of(coff, nc, pa); cb.qlo = coff[ca]; cb.qhi = coff[ca + 1]; }
    if ((long long)pb >= cb.qhi) return 1;
    const long long sa = seed_pj(a, pa, G), sb = seed_pj(b, pb, G);
    const long long s0 = sa < sb ? sa : sb, s1 = sa < sb ? sb : sa;
    if (!(s0 >= cb.slo && s0 < cb.shi)) { const int cs = contig_of(coff, nc, s0); cb.slo = coff[cs]; cb.shi = coff[cs + 1]; }
    return s1 >= cb.shi ? 1 : 0;
}
// Clusters without a scan over the anchors: a workgroup owns SF_TILE consecutive anchors (8 per thread); this pass leaves the number
// of cluster starts of every tile and a byte of start bits per thread; after a scan of the tile counts (one entry per 2048 anchors)
// seed_cluster_emit_kernel writes the index of the k-th start of the tile to c_first[first[tile] + k] from the bits.  The keys are
// read once (8 B + 1 bit per anchor; until late in round 6 the emit pass read and tested them again: 16 B; flags + scan + scatter of
// round 4 moved 42).
#define SF_ITEMS 8
#define SF_TILE (256 * SF_ITEMS)
template <bool PK>
__global__ void __launch_bounds__(256) seed_clusters_kernel(int64_t na, int64_t G, const unsigned long long *__restrict__ akey,
                                                            const unsigned *__restrict__ aval, const int64_t *__restrict__ coff, int nc,
                                                            int32_t *__restrict__ tile_cnt, uint8_t *__restrict__ start_bits) {
    __shared__ int s_tmp[8];
    // the tile goes through LDS (round 6): a thread walks 8 CONSECUTIVE anchors, and read straight from memory that was 64 lanes 64 bytes
    // apart in every load instruction (64 lines per instruction, each fetched 8 times).  Coalesced loads, 9-word rows against bank conflicts.
    __shared__ unsigned long long s_k[SF_TILE + SF_TILE / 8 + 1];
    const int64_t tile0 = (int64_t)blockIdx.x * SF_TILE;
    const int64_t base = tile0 + (int64_t)threadIdx.x * SF_ITEMS;
#pragma unroll
    for (int q = 0; q < SF_ITEMS; q++) {
        const int j = q * 256 + (int)threadIdx.x;
        const int64_t i = tile0 + j;
        s_k[1 + j + (j >> 3)] = akey[i < na ? i : na - 1];
    }
    if (threadIdx.x == 0) s_k[0] = tile0 > 0 && tile0 <= na ? akey[tile0 - 1] : 0ull;
    __syncthreads();
    unsigned long long k[SF_ITEMS + 1];
    k[0] = s_k[threadIdx.x * 9];                    // (the anchor before the thread's first: slot 9 t is the pad behind thread t - 1's row...
    if (threadIdx.x > 0) k[0] = s_k[threadIdx.x * 9 - 1];      //  ... so it is read from that row's last word)
#pragma unroll
    for (int q = 0; q < SF_ITEMS; q++) k[q + 1] = s_k[1 + threadIdx.x * 9 + q];
    unsigned bits = 0;
    int c = 0;
    SeedContigs cb = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < SF_ITEMS; q++) {
        const int64_t i = base + q;
        if (i < na && seed_opens<PK>(i, k[q], k[q + 1], aval, G, coff, nc, cb)) { bits |= 1u << q; c++; }
    }
    int total;
    (void)block_excl_scan(c, s_tmp, &total);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
    start_bits[(int64_t)blockIdx.x * 256 + threadIdx.x] = (uint8_t)bits;      // (for seed_cluster_emit_kernel)
}
// pass EMIT from the start bits pass COUNT left (a byte per thread: 1/64 of the keys; round 6: the second reading of the 713 M keys and
// of their cluster tests was half of the stage's 7.5 ms)
__global__ void __launch_bounds__(256) seed_cluster_emit_kernel(int64_t na, const uint8_t *__restrict__ start_bits, const int64_t *__restrict__ tile_first,
                                                                unsigned *__restrict__ c_first) {
    __shared__ int s_tmp[8];
    const int64_t base = (int64_t)blockIdx.x * SF_TILE + (int64_t)threadIdx.x * SF_ITEMS;
    const unsigned bits = start_bits[(int64_t)blockIdx.x * 256 + threadIdx.x];
    int total;
    const int excl = block_excl_scan(__popc(bits), s_tmp, &total);
    int64_t o = tile_first[blockIdx.x] + excl;
#pragma unroll
    for (int q = 0; q < SF_ITEMS; q++) if ((bits >> q) & 1u) c_first[o++] = (unsigned)(base + q);
    if (base <= na - 1 && na - 1 < base + SF_ITEMS) c_first[tile_first[gridDim.x]] = (unsigned)na;   // (the thread of the last anchor closes the list)
}
// one thread per cluster; EMIT = false counts the pieces, EMIT = true writes them at pfirst[cluster]
template <bool EMIT, bool PK>
__global__ void seed_piece_kernel(int64_t ncl, int64_t G, int64_t seg_len, const unsigned long long *__restrict__ akey,
                                  const unsigned *__restrict__ aval, const unsigned *__restrict__ c_first,
                                  const int64_t *__restrict__ coff, int nc, const int32_t *__restrict__ seg_base,
                                  int32_t *__restrict__ pcnt, const int64_t *__restrict__ pfirst,
                                  unsigned long long *__restrict__ okey, unsigned *__restrict__ oval, int32_t *__restrict__ o_qseg,
                                  int32_t *__restrict__ o_sseg, int64_t *__restrict__ o_qs, int64_t *__restrict__ o_qe,
                                  int64_t *__restrict__ o_ss, int64_t *__restrict__ o_se,
                                  const unsigned *__restrict__ list, const unsigned long long *__restrict__ n_list) {
    // Round 6: both passes run over the DENSE list of the clusters that can be an HSP (seed_piece_select_kernel: >= 3 anchors and
    // the query span, 3 % of them).  With a thread per cluster nearly every wavefront held one or two such clusters and walked the
    // whole piece loop (64-bit divisions, contig searches, eight stores) for them: 3.5 + 4.5 ms for 11 M records.
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nl = (int64_t)*n_list;
    (void)ncl;
    for (int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; li < nl; li += stride) {
    const int64_t k = list[li];
    const unsigned b = c_first[k], e = c_first[k + 1];
    int n = 0;
    const unsigned long long kf = akey[b], kl = akey[e - 1];
    const unsigned qf = anc_pi<PK>(kf, aval, b), ql = anc_pi<PK>(kl, aval, e - 1);
    const long long q0 = qf, q1 = (long long)ql + CK;
    if (q1 - q0 >= SEED_MINSPAN) {
        const int rel = (int)(anc_sd<PK>(kf) >> 34);
        const long long pf = seed_pj(anc_sd<PK>(kf), qf, G), pl = seed_pj(anc_sd<PK>(kl), ql, G);
        const long long s0 = pf < pl ? pf : pl, s1 = (pf < pl ? pl : pf) + CK;
        const int cq = contig_of(coff, nc, q0), cs = contig_of(coff, nc, pf);
        const long long qb = coff[cq], sb = coff[cs];
        int64_t o = EMIT ? pfirst[k] : 0;
        for (long long a = q0; a < q1;) {
            const long long qsegi = (a - qb) / seg_len;
            long long bnd = qb + (qsegi + 1) * seg_len;
            if (bnd > q1) bnd = q1;
            long long u0, u1;
            if (!rel) { u0 = s0 + (a - q0); u1 = s0 + (bnd - q0); } else { u0 = s1 - (bnd - q0); u1 = s1 - (a - q0); }
            if (u0 < s0) u0 = s0;
            if (u1 > s1) u1 = s1;
            for (long long x = u0; x < u1;) {
                const long long ssegi = (x - sb) / seg_len;
                long long y = sb + (ssegi + 1) * seg_len;
                if (y > u1) y = u1;
                long long a2, b2;
                if (!rel) { a2 = a + (x - u0); b2 = a + (y - u0); } else { a2 = a + (u1 - y); b2 = a + (u1 - x); }
                if (a2 < a) a2 = a;
                if (b2 > bnd) b2 = bnd;
                if (b2 - a2 >= SEED_MINPIECE && y - x >= SEED_MINPIECE) {
                    if (EMIT) {
                        const int32_t qsg = seg_base[cq] + (int32_t)qsegi, ssg = seg_base[cs] + (int32_t)ssegi;
                        const long long qo = qb + qsegi * seg_len, so = sb + ssegi * seg_len;
                        o_qseg[o] = qsg; o_sseg[o] = ssg; o_qs[o] = a2 - qo + 1; o_qe[o] = b2 - qo;
                        if (!rel) { o_ss[o] = x - so + 1; o_se[o] = y - so; } else { o_ss[o] = y - so; o_se[o] = x - so + 1; }
                        okey[o] = ((unsigned long long)(unsigned)qsg << 16) | (unsigned long long)(unsigned)ssg;
                        oval[o] = (unsigned)o;
                        o++;
                    }
                    n++;
                }
                x = y;
            }
            a = bnd;
        }
    }
    if (!EMIT) pcnt[k] = n;
    }
}
// which clusters can be an HSP: >= SEED_MINANCH anchors and a query span >= SEED_MINSPAN (the keys are read only behind the first
// test); their ids go to `list` in any order (one atomic per wavefront) -- the records' places come from the scan of pcnt, which
// this kernel zeroes for everything it leaves out
#define SP_ITEMS 8
#define SPGRID(n) dim3((unsigned)((((n) > 0 ? (n) : 1) + 256 * SP_ITEMS - 1) / (256 * SP_ITEMS))), dim3(256)
#define SPS_ITEMS 16
template <bool PK>
__global__ void __launch_bounds__(256) seed_piece_select_kernel(int64_t ncl, const unsigned long long *__restrict__ akey, const unsigned *__restrict__ aval,
                                                                const unsigned *__restrict__ c_first, int32_t *__restrict__ pcnt, unsigned *__restrict__ list,
                                                                unsigned long long *__restrict__ n_list) {
    // a workgroup owns 256 x SPS_ITEMS consecutive clusters and takes its place in the list with ONE atomic (a wavefront each: 5 M
    // same-address atomics, 34 ms)
    __shared__ int s_tmp[8];
    __shared__ unsigned long long s_base;
    const int64_t tile = (int64_t)blockIdx.x * (256 * SPS_ITEMS);
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < SPS_ITEMS; j++) {
        const int64_t k = tile + j * 256 + threadIdx.x;
        bool want = false;
        if (k < ncl) {
            const unsigned b = c_first[k], e = c_first[k + 1];
            if (e - b >= SEED_MINANCH) {
                const unsigned qf = anc_pi<PK>(akey[b], aval, b), ql = anc_pi<PK>(akey[e - 1], aval, e - 1);
                want = (long long)ql + CK - (long long)qf >= SEED_MINSPAN;
            }
            if (!want) pcnt[k] = 0;
        }
        bits |= (unsigned)want << j;
    }
    int total;
    const int excl = block_excl_scan(__popc(bits), s_tmp, &total);
    if (threadIdx.x == 0) s_base = total ? atomicAdd(n_list, (unsigned long long)total) : 0ull;
    __syncthreads();
    unsigned long long o = s_base + (unsigned long long)excl;
#pragma unroll
    for (int j = 0; j < SPS_ITEMS; j++)
        if ((bits >> j) & 1u) list[o++] = (unsigned)(tile + j * 256 + threadIdx.x);
}
__global__ void seed_gather_kernel(int64_t n, const unsigned *__restrict__ perm, const int32_t *__restrict__ i_qseg,
                                   const int32_t *__restrict__ i_sseg, const int64_t *__restrict__ i_qs, const int64_t *__restrict__ i_qe,
                                   const int64_t *__restrict__ i_ss, const int64_t *__restrict__ i_se, int32_t *__restrict__ o_qseg,
                                   int32_t *__restrict__ o_sseg, int64_t *__restrict__ o_qs, int64_t *__restrict__ o_qe,
                                   int64_t *__restrict__ o_ss, int64_t *__restrict__ o_se) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned s = perm[i];
    o_qseg[i] = i_qseg[s]; o_sseg[i] = i_sseg[s]; o_qs[i] = i_qs[s]; o_qe[i] = i_qe[s]; o_ss[i] = i_ss[s]; o_se[i] = i_se[s];
}

// >>> seed_shard (the twin, oracle/hite_oracle_copies.c, carries the same arithmetic)
// Range of rank r of `world` in lin = strand * 2 G + diagonal.  Diagonals of two uniform positions have a triangular density
// (difference for the forward strand, sum for the reverse one, both peaking at G), so the edges sit at equal steps of its
// distribution function, not of the diagonal: mass t in [0, 1) of one strand lies below G sqrt(2 t) (t <= 1/2) or
// 2 G - G sqrt(2 (1 - t)).  Edges are rounded down to multiples of 64 (a cluster never leaves its 64-diagonal bucket).
static unsigned long long seed_shard_edge(long long G, int k, int world) {     // lower edge of rank k (k == world: the end)
    if (k <= 0) return 0ull;
    if (k >= world) return 4ull * (unsigned long long)G + 64ull;
    const double mass = 2.0 * (double)k / (double)world;                        // over both strands
    const int strand = mass >= 1.0 ? 1 : 0;
    const double t = mass - (double)strand;
    const double d = t <= 0.5 ? (double)G * sqrt(2.0 * t) : 2.0 * (double)G - (double)G * sqrt(2.0 * (1.0 - t));
    unsigned long long e = (unsigned long long)(d < 0 ? 0 : d);
    e &= ~63ull;
    return (strand ? 2ull * (unsigned long long)G : 0ull) + e;
}
// <<< seed_shard
static SeedShard seed_shard_of(hite_ctx *ctx, long long G) {
    SeedShard s; s.lo = 0ull; s.hi = 0ull; s.twoG = 2 * G; s.sharded = 0;
    if (ctx->seed_world > 1) {
        // (2 G must be a multiple of 64 for the strand border to be a bucket border: the forward keys end below 2 G anyway)
        s.lo = seed_shard_edge(G, ctx->seed_rank, ctx->seed_world);
        s.hi = seed_shard_edge(G, ctx->seed_rank + 1, ctx->seed_world);
        s.sharded = 1;      // (tiny genome / many ranks: neighbouring edges can round to the same multiple of 64 -- an EMPTY share, which owns nothing)
    }
    return s;
}
// hite_seed_allvsall[_dev] of this context keeps the share of rank `rank` of `world` (world <= 1: everything; the default)
extern "C" int hite_seed_shard(hite_ctx *ctx, int32_t rank, int32_t world) {
    if (!ctx || world < 0 || (world > 1 && (rank < 0 || rank >= world))) return HITE_EINVAL;
    ctx->seed_rank = world > 1 ? rank : 0;
    ctx->seed_world = world > 1 ? world : 0;
    return HITE_OK;
}

// segment table of the packed genome (host side): per contig ceil(len / seg_len) segments, ids in contig order
extern "C" int hite_seed_segments(hite_ctx *ctx, int64_t seg_len, int32_t cap, int32_t *seg_chrom, int64_t *seg_off, int32_t *nseg_out) {
    if (!ctx || !ctx->h_contig_off || seg_len <= 0 || !nseg_out) return HITE_EINVAL;
    int n = 0;
    for (int c = 0; c < ctx->n_contigs; c++) {
        const int64_t L = ctx->h_contig_off[c + 1] - ctx->h_contig_off[c];
        int64_t o = 0;
        do {
            if (seg_chrom && seg_off && n < cap) { seg_chrom[n] = c; seg_off[n] = o; }
            n++;
            o += seg_len;
        } while (o < L);
    }
    *nseg_out = n;
    return (seg_chrom && n > cap) ? HITE_ECAP : HITE_OK;
}

static int seed_allvsall_impl(hite_ctx *ctx, void **state_io, int64_t seg_len, int64_t max_anchors, int64_t cap,
                              int32_t *qseg, int32_t *sseg, int64_t *qs, int64_t *qe, int64_t *ss, int64_t *se,
                              int64_t *n_out, int64_t *stats_out, void **dev_out);
// device-resident form: the six arrays stay in the index state's arena (valid until the next call that uses the state)
extern "C" int hite_seed_allvsall_dev(hite_ctx *ctx, void **state_io, int64_t seg_len, int64_t max_anchors, int32_t **d_qseg,
                                      int32_t **d_sseg, int64_t **d_qs, int64_t **d_qe, int64_t **d_ss, int64_t **d_se, int64_t *n_out,
                                      int64_t *stats_out) {
    if (!d_qseg || !d_sseg || !d_qs || !d_qe || !d_ss || !d_se) return HITE_EINVAL;
    void *ptrs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int rc = seed_allvsall_impl(ctx, state_io, seg_len, max_anchors, -1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, n_out,
                                stats_out, ptrs);
    *d_qseg = (int32_t *)ptrs[0]; *d_sseg = (int32_t *)ptrs[1]; *d_qs = (int64_t *)ptrs[2]; *d_qe = (int64_t *)ptrs[3];
    *d_ss = (int64_t *)ptrs[4]; *d_se = (int64_t *)ptrs[5];
    return rc;
}
extern "C" int hite_seed_allvsall(hite_ctx *ctx, void **state_io, int64_t seg_len, int64_t max_anchors, int64_t cap,
                                  int32_t *qseg, int32_t *sseg, int64_t *qs, int64_t *qe, int64_t *ss, int64_t *se,
                                  int64_t *n_out, int64_t *stats_out /* 4 x int64 or NULL: seeds, anchors, clusters, records */) {
    if (cap < 0) return HITE_EINVAL;
    return seed_allvsall_impl(ctx, state_io, seg_len, max_anchors, cap, qseg, sseg, qs, qe, ss, se, n_out, stats_out, nullptr);
}
static int seed_allvsall_impl(hite_ctx *ctx, void **state_io, int64_t seg_len, int64_t max_anchors, int64_t cap,
                              int32_t *qseg, int32_t *sseg, int64_t *qs, int64_t *qe, int64_t *ss, int64_t *se,
                              int64_t *n_out, int64_t *stats_out, void **dev_out /* 6 pointers, or NULL: copy to the host arrays */) {
    if (!ctx || !ctx->d_bases || !state_io || seg_len <= 0 || !n_out) return HITE_EINVAL;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = nullptr;
    if (!*state_io || ((CopyState *)*state_io)->restricted) CCHK(hite_copy_index_build(ctx, state_io, nullptr));   // seeding walks EVERY minimizer
    CopyState *S = (CopyState *)*state_io;
    *n_out = 0;
    const int64_t M = S->M, G = ctx->n_bases;
    if (stats_out) { stats_out[0] = M; stats_out[1] = stats_out[2] = stats_out[3] = 0; }
    if (M == 0) return HITE_OK;
    CCHK(arena_reset(ctx, S->arena, true));
    Arena &A = S->arena;
    void *p;
    // segment bases per contig
    std::vector<int32_t> hbase(ctx->n_contigs + 1);
    hbase[0] = 0;
    for (int c = 0; c < ctx->n_contigs; c++) {
        const int64_t L = ctx->h_contig_off[c + 1] - ctx->h_contig_off[c];
        hbase[c + 1] = hbase[c] + (int32_t)(L > 0 ? (L + seg_len - 1) / seg_len : 1);
    }
    if (hbase[ctx->n_contigs] >= 65536) return HITE_EINVAL;   // segment ids are packed into 16 bits of the final sort key
    int32_t *seg_base;
    CCHK(arena_alloc(ctx, A, (size_t)(ctx->n_contigs + 1) * 4, &p)); seg_base = (int32_t *)p;
    HITE_CHECK(ctx, hipMemcpyAsync(seg_base, hbase.data(), (size_t)(ctx->n_contigs + 1) * 4, hipMemcpyHostToDevice, st));
    int tk_rc = hite_prof_begin(ctx, "seed_runs_count", st);
    // runs of equal hs >> 1 in the index; per seed -- in position order, the rank the index build left in idx_t -- its partner count
    // and its record (index entry, place in its run)
    // HITE_SEED_COUNT2=0: the run arrays (flags, scan, first-of-run) and the thread-per-entry kernel of round 5
    static const bool count2 = [] { const char *e = getenv("HITE_SEED_COUNT2"); return !(e && *e == '0'); }();
    int32_t *rflag = nullptr, *cnt; int64_t *rid = nullptr, *bs, *aoff; unsigned *run_first = nullptr; uint2 *srec;
    if (!count2) {
        CCHK(arena_alloc(ctx, A, (size_t)(M + 1) * 4, &p)); rflag = (int32_t *)p;
        CCHK(arena_alloc(ctx, A, (size_t)(M + 2) * 8, &p)); rid = (int64_t *)p;
        CCHK(arena_alloc(ctx, A, (size_t)(M + 2) * 4, &p)); run_first = (unsigned *)p;
    }
    CCHK(arena_alloc(ctx, A, (size_t)scan_tmp_elems(M) * 8, &p)); bs = (int64_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(M + 1) * 4, &p)); cnt = (int32_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(M + 2) * 8, &p)); aoff = (int64_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(M + 1) * 8, &p)); srec = (uint2 *)p;
    const dim3 c2grid((unsigned)((M + SC_TILE - 1) / SC_TILE));
    if (!count2) {
        hipLaunchKernelGGL(seed_runflag_kernel, CGRID(M), 0, st, M, S->idx_hs, rflag);
        CCHK(scan_excl_buf<int32_t>(ctx, bs, rflag, M, rid, st));
        hipLaunchKernelGGL(seed_runfirst_kernel, CGRID(M), 0, st, M, rflag, rid, run_first);
    }
    const SeedShard shard = seed_shard_of(ctx, G);
    // HITE_SEED_PLACE=0 / 1 (tests): never / always through the grouped form
    static const int place_mode = [] { const char *e = getenv("HITE_SEED_PLACE"); return e && *e ? atoi(e) : -1; }();
    if (place_mode == 1 || (place_mode != 0 && M >= RS_WIDE_MIN)) {
        unsigned long long *pk; unsigned *pv;
        CCHK(arena_alloc(ctx, A, (size_t)(M + 1) * 8, &p)); pk = (unsigned long long *)p;
        CCHK(arena_alloc(ctx, A, (size_t)(M + 1) * 4, &p)); pv = (unsigned *)p;
        Sorter sp;
        CCHK(sorter_from_arena(sp, ctx, A, st, M));
        if (count2) hipLaunchKernelGGL(seed_count2_kernel<true>, c2grid, dim3(256), 0, st, M, G, S->idx_hs, S->idx_key, S->idx_t, shard, srec, pk, pv);
        else hipLaunchKernelGGL(seed_count_kernel<true>, CGRID(M), 0, st, M, G, rflag, rid, run_first, S->idx_key, S->idx_t, shard, srec, pk, pv);
        int mb = 1;
        while (mb < 32 && (1ll << mb) < M) mb++;                  // ranks < 2^mb
        const int shift = mb > 10 ? mb - 10 : 0;
        CCHK(sorter_sort_bits_swap(sp, &pk, &pv, M, shift, shift + (M >= RS_WIDE_MIN ? 10 : 8)));       // ONE pass
        const int nb = (int)((M + 256 * SPL_ITEMS - 1) / (256 * SPL_ITEMS));
        // what keeps the window inside an L2 (4 MB): two workgroups per compute unit -- 64 KB of LDS asked for, none used --, i.e. 131 k
        // records of a chiplet in flight, half a group; and the counts NOT scattered beside the records (a second window): seed_cnt_kernel
        // reads them back in order.  1 Gbp, 182 M seeds, seed_runs_count: direct scatter 10.7 ms; grouped 9.1; without the counts 7.8;
        // with the LDS cap 7.1.
        static const int place_lds = [] { const char *e = getenv("HITE_SEED_PLACE_LDS"); return e && *e ? atoi(e) : 65536; }();
        static const bool place_cnt = [] { const char *e = getenv("HITE_SEED_PLACE_CNT"); return e && *e == '1'; }();
        hipLaunchKernelGGL(seed_place_kernel, dim3((unsigned)nb), dim3(256), (size_t)place_lds, st, M, nb, pk, pv, srec, place_cnt ? cnt : (int32_t *)nullptr);
        if (!place_cnt) hipLaunchKernelGGL(seed_cnt_kernel, CGRID(M), 0, st, M, srec, cnt);
    } else {
        if (count2) hipLaunchKernelGGL(seed_count2_kernel<false>, c2grid, dim3(256), 0, st, M, G, S->idx_hs, S->idx_key, S->idx_t, shard, srec,
                                       (unsigned long long *)nullptr, (unsigned *)nullptr);
        else hipLaunchKernelGGL(seed_count_kernel<false>, CGRID(M), 0, st, M, G, rflag, rid, run_first, S->idx_key, S->idx_t, shard, srec,
                                (unsigned long long *)nullptr, (unsigned *)nullptr);
        hipLaunchKernelGGL(seed_cnt_kernel, CGRID(M), 0, st, M, srec, cnt);
    }
    CCHK(scan_excl_buf<int32_t>(ctx, bs, cnt, M, aoff, st));
    hite_prof_end(ctx, tk_rc, st);
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal, aoff + M, 8, hipMemcpyDeviceToDevice, st));
    CCHK(read_back(ctx, S, st, 1));
    const int64_t na = S->h_pin[0];
    if (stats_out) stats_out[1] = na;
    if (na == 0) return HITE_OK;
    if (na > max_anchors || na >= 0xffffffffll) return HITE_ECAP;
    // HITE_SEED_PACK=0 (tests): the unpacked records also where the packed ones fit
    static const bool pack_ok = [] { const char *e = getenv("HITE_SEED_PACK"); return !(e && *e == '0'); }();
    const bool packed = pack_ok && G <= SEED_PACK_MAXG;
    unsigned long long *akey; unsigned *aval = nullptr;
    CCHK(arena_alloc(ctx, A, (size_t)(na + 1) * 8, &p)); akey = (unsigned long long *)p;
    if (!packed) { CCHK(arena_alloc(ctx, A, (size_t)(na + 1) * 4, &p)); aval = (unsigned *)p; }
    {
        int tk = hite_prof_begin(ctx, "seed_anchor", st);
        const dim3 cgrid((unsigned)((M + 255) / 256));
        if (shard.sharded) {
            if (packed) hipLaunchKernelGGL(seed_anchor_kernel<true>, CGRID(M), 0, st, M, G, srec, S->idx_key, cnt, aoff, shard, akey, aval);
            else hipLaunchKernelGGL(seed_anchor_kernel<false>, CGRID(M), 0, st, M, G, srec, S->idx_key, cnt, aoff, shard, akey, aval);
        } else if (packed) hipLaunchKernelGGL(seed_anchor_coop_kernel<true>, cgrid, dim3(256), 0, st, M, G, srec, S->idx_key, aoff, akey, aval);
        else hipLaunchKernelGGL(seed_anchor_coop_kernel<false>, cgrid, dim3(256), 0, st, M, G, srec, S->idx_key, aoff, akey, aval);
        hite_prof_end(ctx, tk, st);
    }
    {
        // stable sort on (strand, diagonal >> 6); after an odd number of passes the sorted records sit in the sorter's buffers
        // (no copy back: akey / aval are re-pointed)
        int tk = hite_prof_begin(ctx, "seed_anchor_sort", st);
        Sorter so;
        CCHK(sorter_from_arena(so, ctx, A, st, na, !packed));
        if (packed) { unsigned *nov = nullptr; CCHK(sorter_sort_bits_swap(so, &akey, &nov, na, 36, 62)); }
        else CCHK(sorter_sort_bits_swap(so, &akey, &aval, na, 6, 35));
        hite_prof_end(ctx, tk, st);
    }
    // clusters: starts per tile of 2048 anchors -> scan -> the index of every cluster's first anchor (+ sentinel)
    int tk_cl = hite_prof_begin(ctx, "seed_clusters", st);
    const int64_t ntile = (na + SF_TILE - 1) / SF_TILE;
    int32_t *tcnt; int64_t *tfirst, *bs2; unsigned *c_first;
    CCHK(arena_alloc(ctx, A, (size_t)(ntile + 1) * 4, &p)); tcnt = (int32_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(ntile + 2) * 8, &p)); tfirst = (int64_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)scan_tmp_elems(ntile) * 8, &p)); bs2 = (int64_t *)p;
    uint8_t *sbits;
    CCHK(arena_alloc(ctx, A, (size_t)ntile * 256, &p)); sbits = (uint8_t *)p;
    if (packed) hipLaunchKernelGGL(seed_clusters_kernel<true>, dim3((unsigned)ntile), dim3(256), 0, st, na, G, akey, aval, ctx->d_contig_off, ctx->n_contigs, tcnt, sbits);
    else hipLaunchKernelGGL(seed_clusters_kernel<false>, dim3((unsigned)ntile), dim3(256), 0, st, na, G, akey, aval, ctx->d_contig_off, ctx->n_contigs, tcnt, sbits);
    CCHK(scan_excl_buf<int32_t>(ctx, bs2, tcnt, ntile, tfirst, st));
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal, tfirst + ntile, 8, hipMemcpyDeviceToDevice, st));
    CCHK(read_back(ctx, S, st, 1));
    const int64_t ncl = S->h_pin[0];
    if (stats_out) stats_out[2] = ncl;
    CCHK(arena_alloc(ctx, A, (size_t)(ncl + 2) * 4, &p)); c_first = (unsigned *)p;
    hipLaunchKernelGGL(seed_cluster_emit_kernel, dim3((unsigned)ntile), dim3(256), 0, st, na, sbits, (const int64_t *)tfirst, c_first);
    hite_prof_end(ctx, tk_cl, st);
    // pieces: count, scan, emit
    int tk_pc = hite_prof_begin(ctx, "seed_pieces", st);
    int32_t *pcnt; int64_t *pfirst, *bs3;
    CCHK(arena_alloc(ctx, A, (size_t)(ncl + 1) * 4, &p)); pcnt = (int32_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(ncl + 2) * 8, &p)); pfirst = (int64_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)scan_tmp_elems(ncl) * 8, &p)); bs3 = (int64_t *)p;
    unsigned *plist; unsigned long long *pn;       // the clusters that can be an HSP (each holds >= SEED_MINANCH anchors)
    CCHK(arena_alloc(ctx, A, (size_t)(na / SEED_MINANCH + 64) * 4, &p)); plist = (unsigned *)p;
    CCHK(arena_alloc(ctx, A, 16, &p)); pn = (unsigned long long *)p;
    HITE_CHECK(ctx, hipMemsetAsync(pn, 0, 16, st));
    if (packed) hipLaunchKernelGGL(HIP_KERNEL_NAME(seed_piece_select_kernel<true>), dim3((unsigned)((ncl + 256 * SPS_ITEMS - 1) / (256 * SPS_ITEMS))), dim3(256), 0, st, ncl, akey, aval, c_first, pcnt, plist, pn);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(seed_piece_select_kernel<false>), dim3((unsigned)((ncl + 256 * SPS_ITEMS - 1) / (256 * SPS_ITEMS))), dim3(256), 0, st, ncl, akey, aval, c_first, pcnt, plist, pn);
    const dim3 pgrid(8192);       // (both passes stride over the list, whose length stays on the device)
#define SEED_PIECE_COUNT(PKV) hipLaunchKernelGGL(HIP_KERNEL_NAME(seed_piece_kernel<false, PKV>), pgrid, dim3(256), 0, st, ncl, G, seg_len, akey, aval, c_first, ctx->d_contig_off, ctx->n_contigs, \
                       seg_base, pcnt, (const int64_t *)nullptr, (unsigned long long *)nullptr, (unsigned *)nullptr, (int32_t *)nullptr, \
                       (int32_t *)nullptr, (int64_t *)nullptr, (int64_t *)nullptr, (int64_t *)nullptr, (int64_t *)nullptr, plist, pn)
    if (packed) SEED_PIECE_COUNT(true); else SEED_PIECE_COUNT(false);
#undef SEED_PIECE_COUNT
    CCHK(scan_excl_buf<int32_t>(ctx, bs3, pcnt, ncl, pfirst, st));
    hite_prof_end(ctx, tk_pc, st);
    HITE_CHECK(ctx, hipMemcpyAsync(S->d_scal, pfirst + ncl, 8, hipMemcpyDeviceToDevice, st));
    CCHK(read_back(ctx, S, st, 1));
    const int64_t np = S->h_pin[0];
    *n_out = np;
    if (stats_out) stats_out[3] = np;
    if (np == 0) return HITE_OK;
    if (!dev_out && np > cap) return HITE_ECAP;
    if (np >= 0xffffffffll) return HITE_ECAP;
    unsigned long long *okey; unsigned *oval; int32_t *t_qseg, *t_sseg, *f_qseg, *f_sseg; int64_t *t_q[4], *f_q[4];
    CCHK(arena_alloc(ctx, A, (size_t)(np + 1) * 8, &p)); okey = (unsigned long long *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(np + 1) * 4, &p)); oval = (unsigned *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(np + 1) * 4, &p)); t_qseg = (int32_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(np + 1) * 4, &p)); t_sseg = (int32_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(np + 1) * 4, &p)); f_qseg = (int32_t *)p;
    CCHK(arena_alloc(ctx, A, (size_t)(np + 1) * 4, &p)); f_sseg = (int32_t *)p;
    for (int i = 0; i < 4; i++) {
        CCHK(arena_alloc(ctx, A, (size_t)(np + 1) * 8, &p)); t_q[i] = (int64_t *)p;
        CCHK(arena_alloc(ctx, A, (size_t)(np + 1) * 8, &p)); f_q[i] = (int64_t *)p;
    }
#define SEED_PIECE_EMIT(PKV) hipLaunchKernelGGL(HIP_KERNEL_NAME(seed_piece_kernel<true, PKV>), pgrid, dim3(256), 0, st, ncl, G, seg_len, akey, aval, c_first, ctx->d_contig_off, ctx->n_contigs, \
                       seg_base, pcnt, (const int64_t *)pfirst, okey, oval, t_qseg, t_sseg, t_q[0], t_q[1], t_q[2], t_q[3], plist, pn)
    int tk_hs = hite_prof_begin(ctx, "seed_hsp_emit_sort", st);
    if (packed) SEED_PIECE_EMIT(true); else SEED_PIECE_EMIT(false);
#undef SEED_PIECE_EMIT
    {
        Sorter so;
        CCHK(sorter_from_arena(so, ctx, A, st, np));
        CCHK(sorter_sort(so, okey, oval, np, 32));
    }
    hipLaunchKernelGGL(seed_gather_kernel, CGRID(np), 0, st, np, oval, t_qseg, t_sseg, t_q[0], t_q[1], t_q[2], t_q[3], f_qseg, f_sseg, f_q[0],
                       f_q[1], f_q[2], f_q[3]);
    hite_prof_end(ctx, tk_hs, st);
    HITE_CHECK(ctx, hipGetLastError());
    HITE_CHECK(ctx, hipStreamSynchronize(st));
    if (dev_out) {
        dev_out[0] = f_qseg; dev_out[1] = f_sseg; dev_out[2] = f_q[0]; dev_out[3] = f_q[1]; dev_out[4] = f_q[2]; dev_out[5] = f_q[3];
        return HITE_OK;
    }
    HITE_CHECK(ctx, hipMemcpy(qseg, f_qseg, (size_t)np * 4, hipMemcpyDeviceToHost));
    HITE_CHECK(ctx, hipMemcpy(sseg, f_sseg, (size_t)np * 4, hipMemcpyDeviceToHost));
    HITE_CHECK(ctx, hipMemcpy(qs, f_q[0], (size_t)np * 8, hipMemcpyDeviceToHost));
    HITE_CHECK(ctx, hipMemcpy(qe, f_q[1], (size_t)np * 8, hipMemcpyDeviceToHost));
    HITE_CHECK(ctx, hipMemcpy(ss, f_q[2], (size_t)np * 8, hipMemcpyDeviceToHost));
    HITE_CHECK(ctx, hipMemcpy(se, f_q[3], (size_t)np * 8, hipMemcpyDeviceToHost));
    return HITE_OK;
}
