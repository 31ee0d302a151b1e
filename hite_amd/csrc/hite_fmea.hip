// hite_fmea.hip -- FMEA: HSP chaining + interval de-duplication of the coarse stage.
// get_longest_repeats_v4 + process_all_seqs (/root/reference/module/Util.py:4122-4400, 4529-4569).
//
// The reference walks nested Python dicts {query: {subject: [HSP]}}.  Here the same result comes out
// of a sort / scan / segment pipeline on the GPU; every step is order-exact:
//   1  first-appearance ranks of queries and of subjects within a query (atomicMin + block bitonic)
//      -> group slot = (query order, subject order, strand)
//   2  stable LSD radix sort of the HSPs by (slot | strand-aware s_start, s_end)            [:4166-4174]
//   3  cluster sweep, one wavefront per group, lanes test the open cluster's members       [:4176-4227]
//   4  stable radix sort by (cluster | q_start, q_end)                                      [:4231]
//   5  greedy chain extension, one thread per cluster (visited aliasing of equal tuples)   [:4235-4319]
//   6  first-come de-duplication on 10-bp rounded keys: a chain is new iff, for each of its 8 keys,
//      it is the earliest chain carrying that key -> lock-free hash table with atomicMin    [:4344-4390]
//   7  per query: stable sort by length (desc) and greedy 95 % containment filter          [:4529-4563]
// Integer work only.  Algorithmic bytes: 24 B per HSP in, 20 B per interval out; the sorts move
// 12 B x 2 per pass (HBM streaming).
#include "hite_common.h"
#include "hite_scan.h"
#include "hite_sort.h"
#include <vector>

#define FM_MAXSEG 4096

// ---------------------------------------------------------------------------------------------
// step 1: ranks
// ---------------------------------------------------------------------------------------------
struct Hsp { int qs, qe, ss, se; };

__global__ void fm_first_kernel(int64_t n, const int32_t *__restrict__ qseg, const int32_t *__restrict__ sseg,
                                const int64_t *__restrict__ qs, const int64_t *__restrict__ qe, const int64_t *__restrict__ ss,
                                const int64_t *__restrict__ se, int nseg, int *__restrict__ first_q, int *__restrict__ first_pair,
                                uint8_t *__restrict__ keep, int *__restrict__ err) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int q = qseg[i], s = sseg[i];
    bool bad = q < 0 || q >= nseg || s < 0 || s >= nseg || qs[i] == qe[i] || ss[i] == se[i] || qs[i] < 0 || qe[i] < 0 || ss[i] < 0 ||
               se[i] < 0 || qs[i] >= 0x7fffffff || qe[i] >= 0x7fffffff || ss[i] >= 0x7fffffff || se[i] >= 0x7fffffff;
    if (bad) { atomicExch(err, 1); keep[i] = 0; return; }
    bool self = q == s && qs[i] == ss[i] && qe[i] == se[i];  // :4138
    keep[i] = !self;
    if (self) return;
    // an HSP whose predecessor in the table is kept and carries the same query (pair) cannot be the first of it: on a table
    // grouped by (query, subject) -- what blastn and the seeding stage emit -- this removes nearly every same-address atomic
    bool same_q = false, same_p = false;
    if (i > 0) {
        const int q1 = qseg[i - 1], s1 = sseg[i - 1];
        const bool self1 = q1 == s1 && qs[i - 1] == ss[i - 1] && qe[i - 1] == se[i - 1];
        if (!self1) { same_q = q1 == q; same_p = same_q && s1 == s; }
    }
    if (!same_q) atomicMin(&first_q[q], (int)i);
    if (!same_p) atomicMin(&first_pair[(int64_t)q * nseg + s], (int)i);
}

// block bitonic sort of up to FM_MAXSEG (key, id) pairs in LDS; rank_out[id] = position, order_out[pos] = id
__device__ void block_bitonic(unsigned *key, unsigned short *id, int np2) {
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np2; i += blockDim.x) {
                int ixj = i ^ j;
                if (ixj > i) {
                    bool up = (i & k) == 0;
                    unsigned a = key[i], b = key[ixj];
                    unsigned short ia = id[i], ib = id[ixj];
                    bool gt = a > b || (a == b && ia > ib);
                    if (gt == up) { key[i] = b; key[ixj] = a; id[i] = ib; id[ixj] = ia; }
                }
            }
            __syncthreads();
        }
}

__global__ void __launch_bounds__(256) fm_qrank_kernel(int nseg, int np2, const int *__restrict__ first_q, int *__restrict__ qrank,
                                                       int *__restrict__ qorder, int *__restrict__ nq) {
    __shared__ unsigned key[FM_MAXSEG];
    __shared__ unsigned short id[FM_MAXSEG];
    for (int i = threadIdx.x; i < np2; i += 256) { key[i] = i < nseg ? (unsigned)first_q[i] : 0xffffffffu; id[i] = (unsigned short)i; }
    __syncthreads();
    block_bitonic(key, id, np2);
    for (int i = threadIdx.x; i < nseg; i += 256) { int q = id[i]; if (q < nseg) { qrank[q] = i; qorder[i] = q; } }
    if (threadIdx.x == 0) { int c = 0; for (int i = 0; i < nseg; i++) c += key[i] != 0x7fffffffu && key[i] != 0xffffffffu; *nq = c; }
}

// one block per query: rank of each used subject by first appearance
__global__ void __launch_bounds__(256) fm_srank_kernel(int nseg, int np2, const int *__restrict__ first_pair,
                                                       int *__restrict__ srank, int *__restrict__ npairs) {
    __shared__ unsigned key[FM_MAXSEG];
    __shared__ unsigned short id[FM_MAXSEG];
    const int q = blockIdx.x;
    for (int i = threadIdx.x; i < np2; i += 256) { key[i] = i < nseg ? (unsigned)first_pair[(int64_t)q * nseg + i] : 0xffffffffu; id[i] = (unsigned short)i; }
    __syncthreads();
    block_bitonic(key, id, np2);
    for (int i = threadIdx.x; i < nseg; i += 256) { int s = id[i]; if (s < nseg) srank[(int64_t)q * nseg + s] = i; }
    if (threadIdx.x == 0) { int c = 0; for (int i = 0; i < nseg; i++) c += key[i] < 0x7fffffffu; npairs[q] = c; }
}

// gbase over queries in rank order (single thread; nseg is small)
__global__ void fm_gbase_kernel(int nseg, const int *__restrict__ qorder, const int *__restrict__ npairs, int64_t *__restrict__ gbase,
                                int64_t *__restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int64_t acc = 0;
        for (int r = 0; r < nseg; r++) { gbase[r] = acc; acc += npairs[qorder[r]]; }
        *total = acc;
    }
}

// keys of the first sort pair: key1 = s_end key (32 bit), key2 = slot << 31 | s_start key; value = HSP index
__global__ void fm_keys_kernel(int64_t n, const uint8_t *__restrict__ keep, const int64_t *__restrict__ kpos,
                               const int32_t *__restrict__ qseg, const int32_t *__restrict__ sseg, const int64_t *__restrict__ ss,
                               const int64_t *__restrict__ se, int nseg, const int *__restrict__ qrank, const int *__restrict__ srank,
                               const int64_t *__restrict__ gbase, unsigned long long *__restrict__ key_se,
                               unsigned long long *__restrict__ key_slot, unsigned *__restrict__ val, int32_t *__restrict__ slot_cnt) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    int64_t o = kpos[i];
    int q = qseg[i], s = sseg[i];
    bool rev = ss[i] > se[i];                                                     // :4169
    long long slot = 2 * (gbase[qrank[q]] + srank[(int64_t)q * nseg + s]) + (rev ? 1 : 0);
    unsigned sk = rev ? (unsigned)(0x7fffffff - (int)ss[i]) : (unsigned)ss[i];    // (-ss, -se) order for reverse  :4174
    unsigned ek = rev ? (unsigned)(0x7fffffff - (int)se[i]) : (unsigned)se[i];
    key_se[o] = ek;
    key_slot[o] = ((unsigned long long)slot << 31) | sk;
    val[o] = (unsigned)i;
    atomicAdd(&slot_cnt[slot], 1);
}

// after the first sort the value order is "by s_end"; gather key_slot in that order for the second sort
__global__ void fm_gather_key_kernel(int64_t m, const unsigned *__restrict__ val, const int64_t *__restrict__ kpos,
                                     const unsigned long long *__restrict__ key_by_pos, unsigned long long *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    out[i] = key_by_pos[kpos[val[i]]];
}

// ---------------------------------------------------------------------------------------------
// step 3: cluster sweep, one wavefront per group
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fm_cluster_kernel(int64_t nslots, const int64_t *__restrict__ slot_start,
                                                         const unsigned *__restrict__ order /* HSP idx, sorted */,
                                                         const int64_t *__restrict__ qe, const int64_t *__restrict__ ss,
                                                         const int64_t *__restrict__ se, int64_t gap_all, int parity_rev,
                                                         int32_t *__restrict__ clid, int32_t *__restrict__ ncl,
                                                         const int64_t *__restrict__ qgap = nullptr /* per query (hite_chain_all) */,
                                                         const int32_t *__restrict__ qid = nullptr) {
    const int lane = threadIdx.x & 63;
    for (int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); g < nslots; g += (int64_t)gridDim.x * 4) {
        const int64_t b = slot_start[g], e = slot_start[g + 1];
        if (e <= b) { if (lane == 0) ncl[g] = 0; continue; }
        const int64_t gap = qgap ? qgap[qid[order[b]]] : gap_all;
        const bool rev = parity_rev ? (bool)(g & 1) : (ss[order[b]] > se[order[b]]);
        int cl = 0;
        int64_t cstart = b;
        if (lane == 0) clid[b] = 0;
        for (int64_t k = b + 1; k < e; k++) {
            const unsigned hk = order[k];
            const long long ssk = ss[hk], qek = qe[hk];
            bool closed = false;
            for (int64_t top = k - 1; top >= cstart && !closed; top -= 64) {
                int64_t m = top - lane;
                bool hit = false;
                if (m >= cstart) {
                    const unsigned hm = order[m];
                    long long d = rev ? (se[hm] - ssk) : (ssk - se[hm]);
                    hit = d < gap && qek > qe[hm];
                }
                closed = __ballot(hit) != 0ull;
            }
            if (!closed) { cl++; cstart = k; }
            if (lane == 0) clid[k] = cl;
        }
        if (lane == 0) ncl[g] = cl + 1;
    }
}

// keys of the second sort pair: key1 = q_end, key2 = q_start, key3 = global cluster id; value = HSP index
__global__ void fm_ckeys_kernel(int64_t m, const unsigned *__restrict__ order, const int64_t *__restrict__ slot_start,
                                const unsigned long long *__restrict__ key_slot_sorted, const int32_t *__restrict__ clid,
                                const int64_t *__restrict__ cl_base, const int64_t *__restrict__ qs, const int64_t *__restrict__ qe,
                                unsigned long long *__restrict__ k_qe, unsigned long long *__restrict__ k_qs,
                                unsigned long long *__restrict__ k_cl, int32_t *__restrict__ cl_cnt) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    unsigned h = order[i];
    long long slot = (long long)(key_slot_sorted[i] >> 31);
    long long g = cl_base[slot] + clid[i];
    k_qe[i] = (unsigned long long)qe[h];
    k_qs[i] = (unsigned long long)qs[h];
    k_cl[i] = (unsigned long long)g;
    atomicAdd(&cl_cnt[g], 1);
}

// generic: out[i] = src[perm_from[i]] where the current order lists positions of a previous order
__global__ void fm_permute_u64_kernel(int64_t m, const unsigned *__restrict__ pos, const unsigned long long *__restrict__ src,
                                      unsigned long long *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i] = src[pos[i]];
}
__global__ void fm_iota_kernel(int64_t m, unsigned *__restrict__ v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) v[i] = (unsigned)i;
}
__global__ void fm_compose_kernel(int64_t m, const unsigned *__restrict__ pos, const unsigned *__restrict__ order_in,
                                  unsigned *__restrict__ order_out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) order_out[i] = order_in[pos[i]];
}

// ---------------------------------------------------------------------------------------------
// step 5: chaining, one thread per cluster  (Util.py:4235-4319)
// ---------------------------------------------------------------------------------------------
struct Chain { int qs, qe, ss, se; int sseg; int qseg; };

__global__ void fm_chain_kernel(int64_t ncl, const int64_t *__restrict__ cl_start, const unsigned *__restrict__ order,
                                const int32_t *__restrict__ qseg, const int32_t *__restrict__ sseg, const int64_t *__restrict__ qs,
                                const int64_t *__restrict__ qe, const int64_t *__restrict__ ss, const int64_t *__restrict__ se,
                                int64_t gap, uint8_t *__restrict__ vis, int32_t *__restrict__ canon, Chain *__restrict__ chains,
                                int32_t *__restrict__ is_chain) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncl) return;
    const int64_t b = cl_start[c], e = cl_start[c + 1];
    // visited is keyed by the HSP tuple: identical tuples alias (adjacent after the stable sorts)
    for (int64_t i = b; i < e; i++) {
        vis[i] = 0; is_chain[i] = 0;
        int32_t cn = (int32_t)(i - b);  // index relative to the cluster start
        unsigned hi = order[i];
        for (int64_t j = i - 1; j >= b; j--) {
            unsigned hj = order[j];
            if (qs[hj] != qs[hi] || qe[hj] != qe[hi]) break;
            if (ss[hj] == ss[hi] && se[hj] == se[hi]) cn = canon[j];
        }
        canon[i] = cn;
    }
    for (int64_t i = b; i < e; i++) {
        if (vis[b + canon[i]]) continue;
        unsigned hi = order[i];
        long long pqs = qs[hi], pqe = qe[hi], pss = ss[hi], pse = se[hi];
        vis[b + canon[i]] = 1;
        for (int64_t j = i + 1; j < e; j++) {
            if (vis[b + canon[j]]) continue;
            unsigned hj = order[j];
            long long cqs = qs[hj], cqe = qe[hj], css = ss[hj], cse = se[hj];
            if (cqe > pqe) {
                if (pss < pse && css < cse) {
                    if (cse > pse) {
                        if (cqs - pqe < gap && cqe > pqe && css - pse < gap) { pqe = cqe; pss = pss < css ? pss : css; pse = cse; vis[b + canon[j]] = 1; }
                        else if (cqs - pqe >= gap) break;
                    }
                } else if (pss > pse && css > cse) {
                    if (cse < pse) {
                        if (cqs - pqe < gap && cqe > pqe && pse - css < gap) { pqe = cqe; pss = pss > css ? pss : css; pse = cse; vis[b + canon[j]] = 1; }
                        else if (cqs - pqe >= gap) break;
                    }
                }
            }
        }
        Chain ch; ch.qs = (int)pqs; ch.qe = (int)pqe; ch.ss = (int)pss; ch.se = (int)pse; ch.sseg = sseg[hi]; ch.qseg = qseg[hi];
        chains[i] = ch;
        is_chain[i] = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// step 6: de-duplication keys
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ long long fl10(long long x) { long long q = x / 10; if (x % 10 != 0 && x < 0) q--; return q * 10; }
// (chrom:18 | a/10:28 | (b-a)/10 + 2^17 : 18) -- lossless for chrom < 2^18, 0 <= a < 2^28*10, |b-a| < 2^17*10
__device__ __forceinline__ bool pack_key(int chrom, long long a, long long b, unsigned long long *out) {
    long long d = (b - a) / 10 + (1 << 17);
    if (chrom < 0 || chrom >= (1 << 18) || a < 0 || a / 10 >= (1ll << 28) || d < 0 || d >= (1 << 18)) return false;
    *out = ((unsigned long long)chrom << 46) | ((unsigned long long)(a / 10) << 18) | (unsigned long long)d;
    return true;
}
__device__ __forceinline__ unsigned long long mixhash(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
#define HT_EMPTY 0xffffffffffffffffull

__device__ void chain_keys(const Chain &c, const int32_t *seg_chrom, const int64_t *seg_off, unsigned long long *k8, bool *ok) {
    int schr = seg_chrom[c.sseg], qchr = seg_chrom[c.qseg];
    long long sst = seg_off[c.sseg] + c.ss - 1, sen = seg_off[c.sseg] + c.se;
    long long qst = seg_off[c.qseg] + c.qs - 1, qen = seg_off[c.qseg] + c.qe;
    long long s1 = fl10(sst), s2 = s1 + 10, e1 = fl10(sen), e2 = e1 + 10;
    long long a1 = fl10(qst), a2 = a1 + 10, b1 = fl10(qen), b2 = b1 + 10;
    bool good = true;
    good &= pack_key(schr, s1, e1, &k8[0]); good &= pack_key(schr, s1, e2, &k8[1]);
    good &= pack_key(schr, s2, e1, &k8[2]); good &= pack_key(schr, s2, e2, &k8[3]);
    good &= pack_key(qchr, a1, b1, &k8[4]); good &= pack_key(qchr, a1, b2, &k8[5]);
    good &= pack_key(qchr, a2, b1, &k8[6]); good &= pack_key(qchr, a2, b2, &k8[7]);
    *ok = good;
}

__global__ void fm_ht_insert_kernel(int64_t nch, const Chain *__restrict__ chains, const int32_t *__restrict__ seg_chrom,
                                    const int64_t *__restrict__ seg_off, unsigned long long *__restrict__ ht_key,
                                    unsigned *__restrict__ ht_val, unsigned long long mask, int *__restrict__ err) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nch) return;
    unsigned long long k8[8];
    bool ok;
    chain_keys(chains[c], seg_chrom, seg_off, k8, &ok);
    if (!ok) { atomicExch(err, 2); return; }
    for (int t = 0; t < 8; t++) {
        unsigned long long key = k8[t];
        unsigned long long h = mixhash(key) & mask;
        for (;;) {
            // (round 6: a plain read first -- a slot that already holds the key needs no compare-and-swap, and a value that is
            // already below this chain's number no atomic minimum: keys repeat, that is what the table is for, and device-scope
            // atomics on random lines are what bounds the kernel)
            unsigned long long old = __atomic_load_n(&ht_key[h], __ATOMIC_RELAXED);
            if (old == HT_EMPTY) old = atomicCAS(&ht_key[h], HT_EMPTY, key);
            if (old == HT_EMPTY || old == key) {
                if (__atomic_load_n(&ht_val[h], __ATOMIC_RELAXED) > (unsigned)c) atomicMin(&ht_val[h], (unsigned)c);
                break;
            }
            h = (h + 1) & mask;
        }
    }
}

__global__ void fm_ht_query_kernel(int64_t nch, const Chain *__restrict__ chains, const int32_t *__restrict__ seg_chrom,
                                   const int64_t *__restrict__ seg_off, const unsigned long long *__restrict__ ht_key,
                                   const unsigned *__restrict__ ht_val, unsigned long long mask, int64_t max_len,
                                   const int *__restrict__ qrank, unsigned long long *__restrict__ cand_key,
                                   int32_t *__restrict__ is_cand) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nch) return;
    Chain ch = chains[c];
    unsigned long long k8[8];
    bool ok;
    chain_keys(ch, seg_chrom, seg_off, k8, &ok);
    bool isnew = ok;
    for (int t = 0; t < 8 && isnew; t++) {
        unsigned long long key = k8[t];
        unsigned long long h = mixhash(key) & mask;
        while (ht_key[h] != key) h = (h + 1) & mask;
        if (ht_val[h] != (unsigned)c) isnew = false;      // an earlier chain already carried this key
    }
    long long qlen = (long long)ch.qe - ((long long)ch.qs - 1);
    if (qlen < 0) qlen = -qlen;
    bool cand = isnew && qlen >= 80 && qlen < max_len;      // :4379
    is_cand[c] = cand;
    // sort key for process_seq_group: (query order | length descending); stable => chain order among equals
    long long len = ((long long)ch.qe) - ((long long)ch.qs - 1);
    cand_key[c] = ((unsigned long long)qrank[ch.qseg] << 32) | (unsigned long long)(0x7fffffff - (unsigned)(len & 0x7fffffff));
}

__global__ void fm_compact_chains_kernel(int64_t m, const int32_t *__restrict__ flag, const int64_t *__restrict__ pos,
                                         const Chain *__restrict__ in, Chain *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m && flag[i]) out[pos[i]] = in[i];
}
__global__ void fm_compact_cand_kernel(int64_t nch, const int32_t *__restrict__ flag, const int64_t *__restrict__ pos,
                                       const unsigned long long *__restrict__ key_in, unsigned long long *__restrict__ key_out,
                                       unsigned *__restrict__ val_out, int32_t *__restrict__ qcount, const Chain *__restrict__ chains,
                                       const int *__restrict__ qrank) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = c < nch && flag[c];
    int q = -1;
    if (act) {
        key_out[pos[c]] = key_in[c];
        val_out[pos[c]] = (unsigned)c;
        q = qrank[chains[c].qseg];
    }
    // one atomic per DISTINCT query of the wavefront (round 6: the chains arrive grouped by query -- millions of single increments
    // on ~1 000 addresses were most of this kernel's 2.8 ms per Gbp)
    unsigned long long todo = __ballot(act);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int qv = __shfl(q, leader, 64);
        const unsigned long long same = __ballot(act && q == qv);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&qcount[qv], (int)__popcll(same));
        todo &= ~same;
    }
}

// ---------------------------------------------------------------------------------------------
// step 7: per query greedy containment filter on the length-sorted candidates (Util.py:4529-4549)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fm_filter_kernel(int nq, const int64_t *__restrict__ qstart, const unsigned *__restrict__ cidx,
                                                        const Chain *__restrict__ chains, const int64_t *__restrict__ seg_off,
                                                        uint8_t *__restrict__ keepf) {
    const int r = blockIdx.x;
    if (r >= nq) return;
    const int64_t b = qstart[r], e = qstart[r + 1];
    for (int64_t i = b + threadIdx.x; i < e; i += 256) keepf[i] = 1;
    __syncthreads();
    for (int64_t i = b; i < e; i++) {
        if (!keepf[i]) { continue; }   // uniform: keepf[i] is final once all j < i have been processed
        const Chain ci = chains[cidx[i]];
        const long long s1 = seg_off[ci.qseg] + ci.qs - 1, e1 = seg_off[ci.qseg] + ci.qe;
        for (int64_t j = i + 1 + threadIdx.x; j < e; j += 256) {
            if (!keepf[j]) continue;
            const Chain cj = chains[cidx[j]];
            const long long s2 = seg_off[cj.qseg] + cj.qs - 1, e2 = seg_off[cj.qseg] + cj.qe;
            long long lo = s1 > s2 ? s1 : s2, hi = e1 < e2 ? e1 : e2;
            long long ov = hi - lo; if (ov < 0) ov = 0;
            if ((double)ov / (double)(e2 - s2) >= 0.95) keepf[j] = 0;
        }
        __syncthreads();
    }
}

__global__ void fm_emit_kernel(int64_t ncand, const uint8_t *__restrict__ keepf, const int64_t *__restrict__ pos,
                               const unsigned *__restrict__ cidx, const Chain *__restrict__ chains,
                               const int32_t *__restrict__ seg_chrom, const int64_t *__restrict__ seg_off, int64_t cap,
                               int32_t *__restrict__ out_chrom, int64_t *__restrict__ out_start, int64_t *__restrict__ out_end) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncand || !keepf[i]) return;
    int64_t o = pos[i];
    if (o >= cap) return;
    const Chain c = chains[cidx[i]];
    out_chrom[o] = seg_chrom[c.qseg];
    out_start[o] = seg_off[c.qseg] + c.qs - 1;
    out_end[o] = seg_off[c.qseg] + c.qe;
}

__global__ void fm_u8_to_i32_kernel(int64_t n, const uint8_t *__restrict__ in, int32_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
__global__ void fm_fill_i32_kernel(int64_t n, int *__restrict__ p, int v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
// Every temporary of one call comes from the context's grow-only arena (no hipMalloc / hipFree on the steady-state path: sixty of
// each per FMEA call used to cost milliseconds and, now and then, a second when the runtime trimmed its pool).
static thread_local hite_ctx *tl_fbuf_ctx = nullptr;
struct FArenaScope {
    bool ok = false;
    explicit FArenaScope(hite_ctx *ctx) {
        if (!ctx || hipSetDevice(ctx->device) != hipSuccess) return;
        if (!ctx->fmea_arena) ctx->fmea_arena = new Arena();
        if (arena_reset(ctx, *(Arena *)ctx->fmea_arena, true)) return;
        tl_sort_arena = (Arena *)ctx->fmea_arena;
        tl_fbuf_ctx = ctx;
        ok = true;
    }
    ~FArenaScope() { tl_sort_arena = nullptr; tl_fbuf_ctx = nullptr; }
};
void hite_fmea_release(hite_ctx *ctx) {
    if (!ctx || !ctx->fmea_arena) return;
    arena_free(*(Arena *)ctx->fmea_arena);
    delete (Arena *)ctx->fmea_arena;
    ctx->fmea_arena = nullptr;
}
struct FBuf {
    void *p = nullptr;
    bool owned = true;
    ~FBuf() { if (p && owned) (void)hipFree(p); }
    hipError_t up_or_borrow(const void *src, size_t n, bool on_device) {
        if (on_device) { p = const_cast<void *>(src); owned = false; return hipSuccess; }
        return up(src, n);
    }
    hipError_t alloc(size_t n) {
        if (tl_sort_arena && tl_fbuf_ctx) {
            owned = false;
            return arena_alloc(tl_fbuf_ctx, *tl_sort_arena, n ? n : 16, &p) == HITE_OK ? hipSuccess : hipErrorOutOfMemory;
        }
        return hipMalloc(&p, n ? n : 16);
    }
    hipError_t up(const void *h, size_t n) {
        hipError_t e = alloc(n + 16);
        if (e != hipSuccess) return e;
        return n ? hipMemcpy(p, h, n, hipMemcpyHostToDevice) : hipSuccess;
    }
};
#define FCHK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { snprintf(ctx->err, sizeof ctx->err, "%s:%d %s", __FILE__, __LINE__, hipGetErrorString(e__)); sorter_free(S); return HITE_EHIP; } } while (0)
#define GRID(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256)

static int fmea_impl(hite_ctx *ctx, int64_t n, const int32_t *qseg, const int32_t *sseg, const int64_t *qs,
                     const int64_t *qe, const int64_t *ss, const int64_t *se, bool hsp_on_device, int32_t nseg, const int32_t *seg_chrom,
                     const int64_t *seg_off, int64_t skip_gap, int64_t max_len, int64_t cap, int32_t *out_chrom,
                     int64_t *out_start, int64_t *out_end, int64_t *n_out) {
    FArenaScope arena_scope(ctx);
    if (!ctx || n < 0 || nseg <= 0 || nseg > FM_MAXSEG || !n_out || n >= 0x7fffffff) return HITE_EINVAL;
    *n_out = 0;
    if (n == 0) return HITE_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return HITE_EHIP;
    hipStream_t st = nullptr;
    Sorter S;
    int np2 = 1; while (np2 < nseg) np2 <<= 1;
    FBuf dq, dsg, dqs, dqe, dss, dse, dchr, doff, dfirstq, dfirstp, dkeep, dkeep32, dkpos, dqrank, dqorder, dnq, dsrank, dnpairs, dgbase,
        dtot, derr, dk1, dk2, dval, dslotcnt, dslotstart, dbs, dk2s, dclid, dncl, dclbase, dkqe, dkqs, dkcl, dclcnt, dclstart, dpos,
        dorder2, dvis, dcanon, dchains, dischain, dchpos, dchains2, dhtk, dhtv, dckey, discand, dcpos, dckey2, dcidx, dqcount, dqstart,
        dkeepf, dkeepf32, dopos, doc, dos, doe, dtmpk;
    FCHK(dq.up_or_borrow(qseg, n * 4, hsp_on_device)); FCHK(dsg.up_or_borrow(sseg, n * 4, hsp_on_device));
    FCHK(dqs.up_or_borrow(qs, n * 8, hsp_on_device)); FCHK(dqe.up_or_borrow(qe, n * 8, hsp_on_device));
    FCHK(dss.up_or_borrow(ss, n * 8, hsp_on_device)); FCHK(dse.up_or_borrow(se, n * 8, hsp_on_device));
    FCHK(dchr.up(seg_chrom, nseg * 4)); FCHK(doff.up(seg_off, nseg * 8));
    FCHK(dfirstq.alloc((size_t)nseg * 4)); FCHK(dfirstp.alloc((size_t)nseg * nseg * 4)); FCHK(dkeep.alloc(n)); FCHK(dkeep32.alloc(n * 4));
    FCHK(dkpos.alloc((n + 1) * 8)); FCHK(dqrank.alloc(nseg * 4)); FCHK(dqorder.alloc(nseg * 4)); FCHK(dnq.alloc(16));
    FCHK(dsrank.alloc((size_t)nseg * nseg * 4)); FCHK(dnpairs.alloc(nseg * 4)); FCHK(dgbase.alloc((nseg + 1) * 8)); FCHK(dtot.alloc(64));
    FCHK(derr.alloc(16)); FCHK(dbs.alloc((size_t)scan_tmp_elems((int64_t)nseg * nseg * 2 + n + 16) * 8));
    FCHK(hipMemset(derr.p, 0, 16));
    hipLaunchKernelGGL(fm_fill_i32_kernel, GRID(nseg), 0, st, (int64_t)nseg, (int *)dfirstq.p, 0x7fffffff);
    hipLaunchKernelGGL(fm_fill_i32_kernel, GRID((int64_t)nseg * nseg), 0, st, (int64_t)nseg * nseg, (int *)dfirstp.p, 0x7fffffff);
    hipLaunchKernelGGL(fm_first_kernel, GRID(n), 0, st, n, (int32_t *)dq.p, (int32_t *)dsg.p, (int64_t *)dqs.p, (int64_t *)dqe.p,
                       (int64_t *)dss.p, (int64_t *)dse.p, nseg, (int *)dfirstq.p, (int *)dfirstp.p, (uint8_t *)dkeep.p, (int *)derr.p);
    int herr = 0;
    FCHK(hipMemcpy(&herr, derr.p, 4, hipMemcpyDeviceToHost));
    if (herr) return HITE_EINVAL;  // zero-length HSP (the reference divides by its length, :4270) or coordinate out of range
    hipLaunchKernelGGL(fm_u8_to_i32_kernel, GRID(n), 0, st, n, (uint8_t *)dkeep.p, (int32_t *)dkeep32.p);
    if (scan_excl_buf<int32_t>(ctx, (int64_t *)dbs.p, (int32_t *)dkeep32.p, n, (int64_t *)dkpos.p, st)) return HITE_EHIP;
    int64_t m = 0;
    FCHK(hipMemcpy(&m, (int64_t *)dkpos.p + n, 8, hipMemcpyDeviceToHost));
    if (m == 0) return HITE_OK;
    hipLaunchKernelGGL(fm_qrank_kernel, dim3(1), dim3(256), 0, st, nseg, np2, (int *)dfirstq.p, (int *)dqrank.p, (int *)dqorder.p, (int *)dnq.p);
    hipLaunchKernelGGL(fm_srank_kernel, dim3(nseg), dim3(256), 0, st, nseg, np2, (int *)dfirstp.p, (int *)dsrank.p, (int *)dnpairs.p);
    hipLaunchKernelGGL(fm_gbase_kernel, dim3(1), dim3(64), 0, st, nseg, (int *)dqorder.p, (int *)dnpairs.p, (int64_t *)dgbase.p, (int64_t *)dtot.p);
    int64_t npairs_total = 0;
    int nq = 0;
    FCHK(hipMemcpy(&npairs_total, dtot.p, 8, hipMemcpyDeviceToHost));
    FCHK(hipMemcpy(&nq, dnq.p, 4, hipMemcpyDeviceToHost));
    const int64_t nslots = 2 * npairs_total;
    FCHK(dk1.alloc((m + 1) * 8)); FCHK(dk2.alloc((m + 1) * 8)); FCHK(dval.alloc((m + 1) * 4)); FCHK(dslotcnt.alloc((nslots + 1) * 4));
    FCHK(dslotstart.alloc((nslots + 2) * 8)); FCHK(dk2s.alloc((m + 1) * 8)); FCHK(dtmpk.alloc((m + 1) * 8));
    FCHK(hipMemset(dslotcnt.p, 0, (nslots + 1) * 4));
    if (sorter_init(S, ctx, st, m)) { sorter_free(S); return HITE_EHIP; }
    hipLaunchKernelGGL(fm_keys_kernel, GRID(n), 0, st, n, (uint8_t *)dkeep.p, (int64_t *)dkpos.p, (int32_t *)dq.p, (int32_t *)dsg.p,
                       (int64_t *)dss.p, (int64_t *)dse.p, nseg, (int *)dqrank.p, (int *)dsrank.p, (int64_t *)dgbase.p,
                       (unsigned long long *)dk1.p, (unsigned long long *)dk2.p, (unsigned *)dval.p, (int32_t *)dslotcnt.p);
    // sort 1: by s_end key, then (stable) by slot | s_start key.  dval holds original HSP indices.
    if (sorter_sort(S, (unsigned long long *)dk1.p, (unsigned *)dval.p, m, 32)) { sorter_free(S); return HITE_EHIP; }
    hipLaunchKernelGGL(fm_gather_key_kernel, GRID(m), 0, st, m, (unsigned *)dval.p, (int64_t *)dkpos.p, (unsigned long long *)dk2.p,
                       (unsigned long long *)dk2s.p);
    int slot_bits = 1; while ((1ll << slot_bits) < nslots + 1) slot_bits++;
    if (sorter_sort(S, (unsigned long long *)dk2s.p, (unsigned *)dval.p, m, 31 + slot_bits)) { sorter_free(S); return HITE_EHIP; }
    if (scan_excl_buf<int32_t>(ctx, (int64_t *)dbs.p, (int32_t *)dslotcnt.p, nslots, (int64_t *)dslotstart.p, st)) { sorter_free(S); return HITE_EHIP; }
    // clusters
    FCHK(dclid.alloc((m + 1) * 4)); FCHK(dncl.alloc((nslots + 1) * 4)); FCHK(dclbase.alloc((nslots + 2) * 8));
    {
        int64_t blocks = (nslots + 3) / 4; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(fm_cluster_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nslots, (int64_t *)dslotstart.p, (unsigned *)dval.p,
                           (int64_t *)dqe.p, (int64_t *)dss.p, (int64_t *)dse.p, skip_gap, 1, (int32_t *)dclid.p, (int32_t *)dncl.p);
    }
    if (scan_excl_buf<int32_t>(ctx, (int64_t *)dbs.p, (int32_t *)dncl.p, nslots, (int64_t *)dclbase.p, st)) { sorter_free(S); return HITE_EHIP; }
    int64_t ncl = 0;
    FCHK(hipMemcpy(&ncl, (int64_t *)dclbase.p + nslots, 8, hipMemcpyDeviceToHost));
    FCHK(dkqe.alloc((m + 1) * 8)); FCHK(dkqs.alloc((m + 1) * 8)); FCHK(dkcl.alloc((m + 1) * 8)); FCHK(dclcnt.alloc((ncl + 1) * 4));
    FCHK(dclstart.alloc((ncl + 2) * 8)); FCHK(dpos.alloc((m + 1) * 4)); FCHK(dorder2.alloc((m + 1) * 4));
    FCHK(hipMemset(dclcnt.p, 0, (ncl + 1) * 4));
    hipLaunchKernelGGL(fm_ckeys_kernel, GRID(m), 0, st, m, (unsigned *)dval.p, (int64_t *)dslotstart.p, (unsigned long long *)dk2s.p,
                       (int32_t *)dclid.p, (int64_t *)dclbase.p, (int64_t *)dqs.p, (int64_t *)dqe.p, (unsigned long long *)dkqe.p,
                       (unsigned long long *)dkqs.p, (unsigned long long *)dkcl.p, (int32_t *)dclcnt.p);
    // sort 2: positions of the current order by q_end, then q_start, then cluster (each stable)
    hipLaunchKernelGGL(fm_iota_kernel, GRID(m), 0, st, m, (unsigned *)dpos.p);
    if (sorter_sort(S, (unsigned long long *)dkqe.p, (unsigned *)dpos.p, m, 32)) { sorter_free(S); return HITE_EHIP; }
    hipLaunchKernelGGL(fm_permute_u64_kernel, GRID(m), 0, st, m, (unsigned *)dpos.p, (unsigned long long *)dkqs.p, (unsigned long long *)dtmpk.p);
    if (sorter_sort(S, (unsigned long long *)dtmpk.p, (unsigned *)dpos.p, m, 32)) { sorter_free(S); return HITE_EHIP; }
    hipLaunchKernelGGL(fm_permute_u64_kernel, GRID(m), 0, st, m, (unsigned *)dpos.p, (unsigned long long *)dkcl.p, (unsigned long long *)dtmpk.p);
    int cl_bits = 1; while ((1ll << cl_bits) < ncl + 1) cl_bits++;
    if (sorter_sort(S, (unsigned long long *)dtmpk.p, (unsigned *)dpos.p, m, cl_bits)) { sorter_free(S); return HITE_EHIP; }
    hipLaunchKernelGGL(fm_compose_kernel, GRID(m), 0, st, m, (unsigned *)dpos.p, (unsigned *)dval.p, (unsigned *)dorder2.p);
    if (scan_excl_buf<int32_t>(ctx, (int64_t *)dbs.p, (int32_t *)dclcnt.p, ncl, (int64_t *)dclstart.p, st)) { sorter_free(S); return HITE_EHIP; }
    // chains
    FCHK(dvis.alloc(m + 16)); FCHK(dcanon.alloc((m + 1) * 4)); FCHK(dchains.alloc((m + 1) * sizeof(Chain))); FCHK(dischain.alloc((m + 1) * 4));
    FCHK(dchpos.alloc((m + 2) * 8)); FCHK(dchains2.alloc((m + 1) * sizeof(Chain)));
    hipLaunchKernelGGL(fm_chain_kernel, GRID(ncl), 0, st, ncl, (int64_t *)dclstart.p, (unsigned *)dorder2.p, (int32_t *)dq.p, (int32_t *)dsg.p,
                       (int64_t *)dqs.p, (int64_t *)dqe.p, (int64_t *)dss.p, (int64_t *)dse.p, skip_gap, (uint8_t *)dvis.p, (int32_t *)dcanon.p,
                       (Chain *)dchains.p, (int32_t *)dischain.p);
    if (scan_excl_buf<int32_t>(ctx, (int64_t *)dbs.p, (int32_t *)dischain.p, m, (int64_t *)dchpos.p, st)) { sorter_free(S); return HITE_EHIP; }
    int64_t nch = 0;
    FCHK(hipMemcpy(&nch, (int64_t *)dchpos.p + m, 8, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(fm_compact_chains_kernel, GRID(m), 0, st, m, (int32_t *)dischain.p, (int64_t *)dchpos.p, (Chain *)dchains.p, (Chain *)dchains2.p);
    // de-duplication
    unsigned long long htsize = 64; while (htsize < (unsigned long long)nch * 16 + 64) htsize <<= 1;
    FCHK(dhtk.alloc(htsize * 8)); FCHK(dhtv.alloc(htsize * 4));
    FCHK(hipMemset(dhtk.p, 0xff, htsize * 8)); FCHK(hipMemset(dhtv.p, 0xff, htsize * 4));
    FCHK(dckey.alloc((nch + 1) * 8)); FCHK(discand.alloc((nch + 1) * 4)); FCHK(dcpos.alloc((nch + 2) * 8));
    hipLaunchKernelGGL(fm_ht_insert_kernel, GRID(nch), 0, st, nch, (Chain *)dchains2.p, (int32_t *)dchr.p, (int64_t *)doff.p,
                       (unsigned long long *)dhtk.p, (unsigned *)dhtv.p, htsize - 1, (int *)derr.p);
    FCHK(hipMemcpy(&herr, derr.p, 4, hipMemcpyDeviceToHost));
    if (herr) { sorter_free(S); return HITE_EINVAL; }  // interval outside the 64-bit key packing range
    hipLaunchKernelGGL(fm_ht_query_kernel, GRID(nch), 0, st, nch, (Chain *)dchains2.p, (int32_t *)dchr.p, (int64_t *)doff.p,
                       (unsigned long long *)dhtk.p, (unsigned *)dhtv.p, htsize - 1, max_len, (int *)dqrank.p,
                       (unsigned long long *)dckey.p, (int32_t *)discand.p);
    if (scan_excl_buf<int32_t>(ctx, (int64_t *)dbs.p, (int32_t *)discand.p, nch, (int64_t *)dcpos.p, st)) { sorter_free(S); return HITE_EHIP; }
    int64_t ncand = 0;
    FCHK(hipMemcpy(&ncand, (int64_t *)dcpos.p + nch, 8, hipMemcpyDeviceToHost));
    if (ncand == 0) { sorter_free(S); return HITE_OK; }
    FCHK(dckey2.alloc((ncand + 1) * 8)); FCHK(dcidx.alloc((ncand + 1) * 4)); FCHK(dqcount.alloc((nseg + 1) * 4)); FCHK(dqstart.alloc((nseg + 2) * 8));
    FCHK(dkeepf.alloc(ncand + 16)); FCHK(dkeepf32.alloc((ncand + 1) * 4)); FCHK(dopos.alloc((ncand + 2) * 8));
    FCHK(hipMemset(dqcount.p, 0, (nseg + 1) * 4));
    hipLaunchKernelGGL(fm_compact_cand_kernel, GRID(nch), 0, st, nch, (int32_t *)discand.p, (int64_t *)dcpos.p, (unsigned long long *)dckey.p,
                       (unsigned long long *)dckey2.p, (unsigned *)dcidx.p, (int32_t *)dqcount.p, (Chain *)dchains2.p, (int *)dqrank.p);
    // process_seq_group: stable sort by (query order | length desc)  -- ncand <= m, the sorter is large enough
    if (sorter_sort(S, (unsigned long long *)dckey2.p, (unsigned *)dcidx.p, ncand, 44)) { sorter_free(S); return HITE_EHIP; }
    if (scan_excl_buf<int32_t>(ctx, (int64_t *)dbs.p, (int32_t *)dqcount.p, nseg, (int64_t *)dqstart.p, st)) { sorter_free(S); return HITE_EHIP; }
    hipLaunchKernelGGL(fm_filter_kernel, dim3(nseg), dim3(256), 0, st, nseg, (int64_t *)dqstart.p, (unsigned *)dcidx.p, (Chain *)dchains2.p,
                       (int64_t *)doff.p, (uint8_t *)dkeepf.p);
    hipLaunchKernelGGL(fm_u8_to_i32_kernel, GRID(ncand), 0, st, ncand, (uint8_t *)dkeepf.p, (int32_t *)dkeepf32.p);
    if (scan_excl_buf<int32_t>(ctx, (int64_t *)dbs.p, (int32_t *)dkeepf32.p, ncand, (int64_t *)dopos.p, st)) { sorter_free(S); return HITE_EHIP; }
    int64_t nout = 0;
    FCHK(hipMemcpy(&nout, (int64_t *)dopos.p + ncand, 8, hipMemcpyDeviceToHost));
    *n_out = nout;
    if (nout > cap) { sorter_free(S); return HITE_ECAP; }
    FCHK(doc.alloc((nout + 1) * 4)); FCHK(dos.alloc((nout + 1) * 8)); FCHK(doe.alloc((nout + 1) * 8));
    hipLaunchKernelGGL(fm_emit_kernel, GRID(ncand), 0, st, ncand, (uint8_t *)dkeepf.p, (int64_t *)dopos.p, (unsigned *)dcidx.p, (Chain *)dchains2.p,
                       (int32_t *)dchr.p, (int64_t *)doff.p, cap, (int32_t *)doc.p, (int64_t *)dos.p, (int64_t *)doe.p);
    FCHK(hipGetLastError());
    FCHK(hipDeviceSynchronize());
    FCHK(hipMemcpy(out_chrom, doc.p, nout * 4, hipMemcpyDeviceToHost));
    FCHK(hipMemcpy(out_start, dos.p, nout * 8, hipMemcpyDeviceToHost));
    FCHK(hipMemcpy(out_end, doe.p, nout * 8, hipMemcpyDeviceToHost));
    sorter_free(S);
    return HITE_OK;
}

extern "C" int hite_fmea_chain(hite_ctx *ctx, int64_t n, const int32_t *qseg, const int32_t *sseg, const int64_t *qs,
                               const int64_t *qe, const int64_t *ss, const int64_t *se, int32_t nseg, const int32_t *seg_chrom,
                               const int64_t *seg_off, int64_t skip_gap, int64_t max_len, int64_t cap, int32_t *out_chrom,
                               int64_t *out_start, int64_t *out_end, int64_t *n_out) {
    return fmea_impl(ctx, n, qseg, sseg, qs, qe, ss, se, false, nseg, seg_chrom, seg_off, skip_gap, max_len, cap, out_chrom, out_start,
                     out_end, n_out);
}
// same with the HSP table resident on the device (what hite_seed_allvsall_dev leaves there); the HSP arrays are not modified
extern "C" int hite_fmea_chain_dev(hite_ctx *ctx, int64_t n, const int32_t *d_qseg, const int32_t *d_sseg, const int64_t *d_qs,
                                   const int64_t *d_qe, const int64_t *d_ss, const int64_t *d_se, int32_t nseg, const int32_t *seg_chrom,
                                   const int64_t *seg_off, int64_t skip_gap, int64_t max_len, int64_t cap, int32_t *out_chrom,
                                   int64_t *out_start, int64_t *out_end, int64_t *n_out) {
    return fmea_impl(ctx, n, d_qseg, d_sseg, d_qs, d_qe, d_ss, d_se, true, nseg, seg_chrom, seg_off, skip_gap, max_len, cap, out_chrom,
                     out_start, out_end, n_out);
}


// =============================================================================================
// get_query_copies (/root/reference/module/Util.py:6828-7030): copy clustering of a blast6 HSP table, the
// blastn route of copy finding.  Same sort / sweep / chain shape as FMEA above:
//   1  first appearance of every (query, subject) pair: stable sort by pair, head of each run        [:6836-6848]
//   2  stable LSD sorts by s_end key, strand | s_start key, (query | first appearance)  -> slots      [:6850-6854]
//   3  cluster sweep, one wavefront per slot (fm_cluster_kernel, strand read from the data)          [:6856-6893]
//   4  stable sorts by q_end, q_start, cluster                                                        [:6900]
//   5  one thread per cluster: greedy chains, keep the longest (first on ties)                        [:6902-6990]
//   6  stable sort of the cluster bests by (query | length desc)                                      [:7000]
//   7  one wavefront per query: coverage test on 64 bests at a time, then in order: stop after
//      max_copy + 1, de-duplicate on (subject, start, end) against the kept ones held in LDS          [:7005-7028]
// =============================================================================================
#define QC_MAXKEEP 256

struct QBest { long long qlen, ss, se; int sid, qid; };

__global__ void qc_check_kernel(int64_t n, const int32_t *__restrict__ qid, const int32_t *__restrict__ sid,
                                const int64_t *__restrict__ qs, const int64_t *__restrict__ qe, const int64_t *__restrict__ ss,
                                const int64_t *__restrict__ se, int nq, int ns, unsigned long long *__restrict__ gk,
                                unsigned *__restrict__ val, int *__restrict__ err) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int q = qid[i], s = sid[i];
    bool bad = q < 0 || q >= nq || s < 0 || s >= ns || qs[i] < 0 || qe[i] < 0 || ss[i] < 0 || se[i] < 0 || qs[i] >= 0x7fffffff ||
               qe[i] >= 0x7fffffff || ss[i] >= 0x7fffffff || se[i] >= 0x7fffffff;
    if (bad) { atomicExch(err, 1); q = 0; s = 0; }
    gk[i] = (unsigned long long)q * (unsigned long long)ns + (unsigned long long)s;
    val[i] = (unsigned)i;
}
__global__ void qc_head_kernel(int64_t n, const unsigned long long *__restrict__ k, int32_t *__restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = i == 0 || k[i] != k[i - 1];
}
// f[hsp] = original index of the first HSP of its (query, subject) run (run id = exclusive scan of the head flags)
__global__ void qc_run_first_kernel(int64_t n, const int32_t *__restrict__ flag, const int64_t *__restrict__ pos,
                                    const unsigned *__restrict__ val, unsigned *__restrict__ gfirst) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) gfirst[pos[i]] = val[i];
}
__global__ void qc_first_kernel(int64_t n, const int32_t *__restrict__ flag, const int64_t *__restrict__ pos,
                                const unsigned *__restrict__ val, const unsigned *__restrict__ gfirst, unsigned *__restrict__ f) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f[val[i]] = gfirst[pos[i] + flag[i] - 1];
}
__global__ void qc_keys_kernel(int64_t n, const int32_t *__restrict__ qid, const int64_t *__restrict__ ss, const int64_t *__restrict__ se,
                               const unsigned *__restrict__ f, unsigned long long *__restrict__ k_e, unsigned long long *__restrict__ k_s,
                               unsigned long long *__restrict__ k_g, unsigned *__restrict__ val) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool rev = ss[i] > se[i];                                                // :6851
    unsigned sk = rev ? (unsigned)(0x7fffffff - (int)ss[i]) : (unsigned)ss[i];     // (-ss, -se) order for reverse  :6854
    unsigned ek = rev ? (unsigned)(0x7fffffff - (int)se[i]) : (unsigned)se[i];
    k_e[i] = ek;
    k_s[i] = ((unsigned long long)(rev ? 1 : 0) << 31) | sk;
    k_g[i] = ((unsigned long long)qid[i] << 31) | f[i];
    val[i] = (unsigned)i;
}
// slot heads in the fully sorted order: (query | first appearance) or the strand changes
__global__ void qc_slot_head_kernel(int64_t n, const unsigned long long *__restrict__ kg, const unsigned *__restrict__ order,
                                    const int64_t *__restrict__ ss, const int64_t *__restrict__ se, int32_t *__restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool h = i == 0 || kg[i] != kg[i - 1];
    if (!h) { unsigned a = order[i], b = order[i - 1]; h = (ss[a] > se[a]) != (ss[b] > se[b]); }
    flag[i] = h;
}
// starts[id] = i for every head; id = exclusive scan of the flags
__global__ void qc_starts_kernel(int64_t n, const int32_t *__restrict__ flag, const int64_t *__restrict__ pos, int64_t *__restrict__ starts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) starts[pos[i]] = i;
    if (i == n - 1) starts[pos[n]] = n;
}
__global__ void qc_ckeys_kernel(int64_t m, const unsigned *__restrict__ order, const int32_t *__restrict__ sflag,
                                const int64_t *__restrict__ spos, const int32_t *__restrict__ clid, const int64_t *__restrict__ cl_base,
                                const int64_t *__restrict__ qs, const int64_t *__restrict__ qe, unsigned long long *__restrict__ k_qe,
                                unsigned long long *__restrict__ k_qs, unsigned long long *__restrict__ k_cl, int32_t *__restrict__ cl_cnt) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    unsigned h = order[i];
    long long slot = spos[i] + sflag[i] - 1;
    long long g = cl_base[slot] + clid[i];
    k_qe[i] = (unsigned long long)qe[h];
    k_qs[i] = (unsigned long long)qs[h];
    k_cl[i] = (unsigned long long)g;
    atomicAdd(&cl_cnt[g], 1);
}

// step 5: one thread per cluster, the longest chain  (Util.py:6902-6990)
__global__ void qc_chain_kernel(int64_t ncl, const int64_t *__restrict__ cl_start, const unsigned *__restrict__ order,
                                const int32_t *__restrict__ qid, const int32_t *__restrict__ sid, const int64_t *__restrict__ qs,
                                const int64_t *__restrict__ qe, const int64_t *__restrict__ ss, const int64_t *__restrict__ se,
                                const double *__restrict__ ident, int64_t qthr, int64_t sthr, uint8_t *__restrict__ vis,
                                int32_t *__restrict__ canon, QBest *__restrict__ best, unsigned long long *__restrict__ lkey,
                                unsigned *__restrict__ lval, int32_t *__restrict__ qcount) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncl) return;
    const int64_t b = cl_start[c], e = cl_start[c + 1];
    // visited_frag is keyed by the tuple value (q_start, q_end, s_start, s_end, identity): equal tuples alias
    for (int64_t i = b; i < e; i++) {
        vis[i] = 0;
        int32_t cn = (int32_t)(i - b);
        unsigned hi = order[i];
        for (int64_t j = i - 1; j >= b; j--) {
            unsigned hj = order[j];
            if (qs[hj] != qs[hi] || qe[hj] != qe[hi]) break;
            if (ss[hj] == ss[hi] && se[hj] == se[hi] && (!ident || ident[hj] == ident[hi])) cn = canon[j];
        }
        canon[i] = cn;
    }
    long long best_len = -1, bss = 0, bse = 0;
    for (int64_t i = b; i < e; i++) {
        if (vis[b + canon[i]]) continue;
        unsigned hi = order[i];
        long long lqs = qs[hi], lqe = qe[hi], lss = ss[hi], lse = se[hi];
        long long cur = lqe - lqs + 1;                                                             // :6909
        vis[b + canon[i]] = 1;
        for (int64_t j = i + 1; j < e; j++) {
            if (vis[b + canon[j]]) continue;
            unsigned hj = order[j];
            long long cqs = qs[hj], cqe = qe[hj], css = ss[hj], cse = se[hj];
            if (cqe > lqe) {
                if (lss < lse && css < cse) {
                    if (cse > lse) {
                        if (cqs - lqe < qthr && css - lse < sthr) { lqe = cqe; lss = lss < css ? lss : css; lse = cse; cur = lqe - lqs; vis[b + canon[j]] = 1; }
                        else if (cqs - lqe >= qthr) break;
                    }
                } else if (lss > lse && css > cse) {
                    if (cse < lse) {
                        if (cqs - lqe < qthr && lse - css < sthr) { lqe = cqe; lss = lss > css ? lss : css; lse = cse; cur = lqe - lqs; vis[b + canon[j]] = 1; }
                        else if (cqs - lqe >= qthr) break;
                    }
                }
            }
        }
        if (cur > best_len) { best_len = cur; bss = lss; bse = lse; }
    }
    const unsigned h0 = order[b];
    QBest r; r.qlen = best_len; r.ss = bss; r.se = bse; r.sid = sid[h0]; r.qid = qid[h0];
    best[c] = r;
    lkey[c] = ((unsigned long long)r.qid << 32) | (unsigned long long)(0x7fffffffll - best_len);    // sort(key = -x[2]) per query  :7000; |len| < 2^31
    lval[c] = (unsigned)c;
    atomicAdd(&qcount[r.qid], 1);
}

// hite_chain_all: one thread per cluster, EVERY chain (the core of FMEA Util.py:10527-10645 and of
// get_full_length_copies_from_blastn_v1 :5990-6103; visited fragments are keyed by the 4-tuple there)
struct ChainA { long long qs, qe, ss, se; int sid, qid, next, pad; };
__global__ void ca_chain_kernel(int64_t ncl, const int64_t *__restrict__ cl_start, const unsigned *__restrict__ order,
                                const int32_t *__restrict__ qid, const int32_t *__restrict__ sid, const int64_t *__restrict__ qs,
                                const int64_t *__restrict__ qe, const int64_t *__restrict__ ss, const int64_t *__restrict__ se,
                                const int64_t *__restrict__ qgap, uint8_t *__restrict__ vis, int32_t *__restrict__ canon,
                                ChainA *__restrict__ chains, int32_t *__restrict__ is_chain, int32_t *__restrict__ qcount) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncl) return;
    const int64_t b = cl_start[c], e = cl_start[c + 1];
    for (int64_t i = b; i < e; i++) {
        vis[i] = 0; is_chain[i] = 0;
        int32_t cn = (int32_t)(i - b);
        unsigned hi = order[i];
        for (int64_t j = i - 1; j >= b; j--) {
            unsigned hj = order[j];
            if (qs[hj] != qs[hi] || qe[hj] != qe[hi]) break;
            if (ss[hj] == ss[hi] && se[hj] == se[hi]) cn = canon[j];
        }
        canon[i] = cn;
    }
    const unsigned h0 = order[b];
    const int q0 = qid[h0];
    const long long gap = qgap[q0];
    int made = 0;
    for (int64_t i = b; i < e; i++) {
        if (vis[b + canon[i]]) continue;
        unsigned hi = order[i];
        long long pqs = qs[hi], pqe = qe[hi], pss = ss[hi], pse = se[hi];
        int next = 0;
        vis[b + canon[i]] = 1;
        for (int64_t j = i + 1; j < e; j++) {
            if (vis[b + canon[j]]) continue;
            unsigned hj = order[j];
            long long cqs = qs[hj], cqe = qe[hj], css = ss[hj], cse = se[hj];
            if (cqe > pqe) {
                if (pss < pse && css < cse) {
                    if (cse > pse) {
                        if (cqs - pqe < gap && css - pse < gap) { pqe = cqe; pss = pss < css ? pss : css; pse = cse; next++; vis[b + canon[j]] = 1; }
                        else if (cqs - pqe >= gap) break;
                    }
                } else if (pss > pse && css > cse) {
                    if (cse < pse) {
                        if (cqs - pqe < gap && pse - css < gap) { pqe = cqe; pss = pss > css ? pss : css; pse = cse; next++; vis[b + canon[j]] = 1; }
                        else if (cqs - pqe >= gap) break;
                    }
                }
            }
        }
        ChainA ch; ch.qs = pqs; ch.qe = pqe; ch.ss = pss; ch.se = pse; ch.sid = sid[h0]; ch.qid = q0; ch.next = next; ch.pad = 0;
        chains[i] = ch;
        is_chain[i] = 1;
        made++;
    }
    atomicAdd(&qcount[q0], made);
}
__global__ void ca_emit_kernel(int64_t m, const int32_t *__restrict__ flag, const int64_t *__restrict__ pos, const ChainA *__restrict__ in,
                               int32_t *__restrict__ o_sid, int64_t *__restrict__ o_qs, int64_t *__restrict__ o_qe,
                               int64_t *__restrict__ o_ss, int64_t *__restrict__ o_se, int32_t *__restrict__ o_next) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m || !flag[i]) return;
    const int64_t o = pos[i];
    const ChainA c = in[i];
    o_sid[o] = c.sid; o_qs[o] = c.qs; o_qe[o] = c.qe; o_ss[o] = c.ss; o_se[o] = c.se; o_next[o] = c.next;
}

// step 7: one wavefront per query
template <bool WRITE>
__global__ void __launch_bounds__(256) qc_select_kernel(int nq, const int64_t *__restrict__ qstart, const unsigned *__restrict__ cidx,
                                                        const QBest *__restrict__ best, const int64_t *__restrict__ qlen,
                                                        const int64_t *__restrict__ slen, double qcov, double scov, int max_copy,
                                                        int32_t *__restrict__ count, const int64_t *__restrict__ first,
                                                        int32_t *__restrict__ o_sid, int64_t *__restrict__ o_s, int64_t *__restrict__ o_e,
                                                        int64_t *__restrict__ o_len, uint8_t *__restrict__ o_minus) {
    __shared__ int s_sid[4][QC_MAXKEEP];
    __shared__ long long s_a[4][QC_MAXKEEP], s_b[4][QC_MAXKEEP];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (q >= nq) return;
    const int64_t b = qstart[q], e = qstart[q + 1];
    const double ql = (double)qlen[q];
    const int64_t o0 = WRITE ? first[q] : 0;
    int cnt = 0;
    for (int64_t base = b; base < e && cnt <= max_copy; base += 64) {
        const int64_t i = base + lane;
        bool pass = false;
        int rsid = 0; long long ra = 0, rb = 0, rl = 0; int rminus = 0;
        if (i < e) {
            const QBest r = best[cidx[i]];
            rsid = r.sid; rl = r.qlen;
            rminus = r.ss > r.se;                                                                  // :7012
            ra = rminus ? r.se : r.ss; rb = rminus ? r.ss : r.se;
            pass = (double)r.qlen / ql >= qcov;                                                    // :7021 / :7025
            if (scov > 0) pass = pass && (double)(rb - ra + 1) / (double)slen[rsid] >= scov;      // :7019-7021
        }
        unsigned long long mask = __ballot(pass);
        while (mask && cnt <= max_copy) {                                                          // len(copies) > max_copy_num: break  :7006
            const int l = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const int xs = __shfl(rsid, l); const long long xa = __shfl(ra, l), xb = __shfl(rb, l), xl = __shfl(rl, l);
            const int xm = __shfl(rminus, l);
            bool dup = false;
            for (int t = lane; t < cnt; t += 64) dup |= s_sid[w][t] == xs && s_a[w][t] == xa && s_b[w][t] == xb;
            if (__ballot(dup)) continue;                                                           // item in keeped_copies
            if (lane == 0) {
                s_sid[w][cnt] = xs; s_a[w][cnt] = xa; s_b[w][cnt] = xb;
                if (WRITE) { o_sid[o0 + cnt] = xs; o_s[o0 + cnt] = xa; o_e[o0 + cnt] = xb; o_len[o0 + cnt] = xl; o_minus[o0 + cnt] = (uint8_t)xm; }
            }
            cnt++;
        }
    }
    if (!WRITE && lane == 0) count[q] = cnt;
}

#define QSORT(k, v, cnt, bits) do { if (sorter_sort(S, (unsigned long long *)(k), (unsigned *)(v), (cnt), (bits))) { sorter_free(S); return HITE_EHIP; } } while (0)
#define QSCAN(in, cnt, out) do { if (scan_excl_buf<int32_t>(ctx, (int64_t *)dbs.p, (int32_t *)(in), (cnt), (int64_t *)(out), st)) { sorter_free(S); return HITE_EHIP; } } while (0)

// all_chains != NULL: hite_chain_all -- steps 1-4 as below with the query's own gap (qlen then holds the gaps), then every
// chain of every cluster instead of steps 5-7
struct ChainAllOut { int32_t *o_sid; int64_t *o_qs, *o_qe, *o_ss, *o_se; int32_t *o_next; };
static int query_copies_impl(hite_ctx *ctx, int64_t n, const int32_t *qid, const int32_t *sid, const int64_t *qs, const int64_t *qe,
                             const int64_t *ss, const int64_t *se, const double *ident, int32_t nq, const int64_t *qlen, int32_t ns,
                             const int64_t *slen, double qcov, double scov, int64_t qthr, int64_t sthr, int32_t max_copy, int64_t cap,
                             int64_t *copy_first, int32_t *o_sid, int64_t *o_s, int64_t *o_e, int64_t *o_len, uint8_t *o_minus,
                             int64_t *n_out, const ChainAllOut *all_chains) {
    FArenaScope arena_scope(ctx);
    if (!ctx || n < 0 || n >= 0x7fffffff || nq <= 0 || ns <= 0 || !qlen || !copy_first || !n_out || max_copy < 0 || max_copy > QC_MAXKEEP - 2 ||
        (scov > 0 && !slen) || (double)nq * (double)ns >= 1e12)
        return HITE_EINVAL;
    *n_out = 0;
    for (int q = 0; q <= nq; q++) copy_first[q] = 0;
    if (n == 0) return HITE_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return HITE_EHIP;
    hipStream_t st = nullptr;
    Sorter S;
    FBuf dq, dsg, dqs, dqe, dss, dse, did, dql, dsl, derr, dbs, dgk, dval, dflag, dpos, df, dke, dks, dkg, dtmpk, dslotstart, dclid, dncl, dclbase,
        dkqe, dkqs, dkcl, dclcnt, dclstart, dp2, dorder2, dvis, dcanon, dbest, dlkey, dlval, dqcount, dqstart, dcount, dfirst, dos, doa, dob, dol, dom, dgf;
    FCHK(dq.up(qid, n * 4)); FCHK(dsg.up(sid, n * 4)); FCHK(dqs.up(qs, n * 8)); FCHK(dqe.up(qe, n * 8)); FCHK(dss.up(ss, n * 8));
    FCHK(dse.up(se, n * 8)); FCHK(dql.up(qlen, (size_t)nq * 8));
    if (ident) FCHK(did.up(ident, n * 8));
    if (slen) FCHK(dsl.up(slen, (size_t)ns * 8));
    const int64_t scan_n = (n > nq ? n : nq) + 16;
    FCHK(derr.alloc(16)); FCHK(dbs.alloc((size_t)scan_tmp_elems(scan_n) * 8)); FCHK(dgk.alloc((n + 1) * 8)); FCHK(dval.alloc((n + 1) * 4));
    FCHK(dflag.alloc((n + 1) * 4)); FCHK(dpos.alloc((n + 2) * 8)); FCHK(df.alloc((n + 1) * 4)); FCHK(dke.alloc((n + 1) * 8));
    FCHK(dks.alloc((n + 1) * 8)); FCHK(dkg.alloc((n + 1) * 8)); FCHK(dtmpk.alloc((n + 1) * 8)); FCHK(dgf.alloc((n + 1) * 4));
    FCHK(hipMemset(derr.p, 0, 16));
    if (sorter_init(S, ctx, st, n)) { sorter_free(S); return HITE_EHIP; }
    // 1: first appearance of every (query, subject) pair
    hipLaunchKernelGGL(qc_check_kernel, GRID(n), 0, st, n, (int32_t *)dq.p, (int32_t *)dsg.p, (int64_t *)dqs.p, (int64_t *)dqe.p, (int64_t *)dss.p,
                       (int64_t *)dse.p, nq, ns, (unsigned long long *)dgk.p, (unsigned *)dval.p, (int *)derr.p);
    int herr = 0;
    FCHK(hipMemcpy(&herr, derr.p, 4, hipMemcpyDeviceToHost));
    if (herr) { sorter_free(S); return HITE_EINVAL; }   // id or coordinate out of range
    int pair_bits = 1; while ((1ull << pair_bits) < (unsigned long long)nq * (unsigned long long)ns + 1) pair_bits++;
    QSORT(dgk.p, dval.p, n, pair_bits);
    hipLaunchKernelGGL(qc_head_kernel, GRID(n), 0, st, n, (unsigned long long *)dgk.p, (int32_t *)dflag.p);
    QSCAN(dflag.p, n, dpos.p);
    hipLaunchKernelGGL(qc_run_first_kernel, GRID(n), 0, st, n, (int32_t *)dflag.p, (int64_t *)dpos.p, (unsigned *)dval.p, (unsigned *)dgf.p);
    hipLaunchKernelGGL(qc_first_kernel, GRID(n), 0, st, n, (int32_t *)dflag.p, (int64_t *)dpos.p, (unsigned *)dval.p, (unsigned *)dgf.p,
                       (unsigned *)df.p);
    // 2: slots = (query, subject by first appearance, strand), sorted by strand-aware (s_start, s_end)
    hipLaunchKernelGGL(qc_keys_kernel, GRID(n), 0, st, n, (int32_t *)dq.p, (int64_t *)dss.p, (int64_t *)dse.p, (unsigned *)df.p,
                       (unsigned long long *)dke.p, (unsigned long long *)dks.p, (unsigned long long *)dkg.p, (unsigned *)dval.p);
    QSORT(dke.p, dval.p, n, 32);
    hipLaunchKernelGGL(fm_permute_u64_kernel, GRID(n), 0, st, n, (unsigned *)dval.p, (unsigned long long *)dks.p, (unsigned long long *)dtmpk.p);
    QSORT(dtmpk.p, dval.p, n, 32);
    hipLaunchKernelGGL(fm_permute_u64_kernel, GRID(n), 0, st, n, (unsigned *)dval.p, (unsigned long long *)dkg.p, (unsigned long long *)dtmpk.p);
    int q_bits = 1; while ((1ll << q_bits) < (long long)nq + 1) q_bits++;
    QSORT(dtmpk.p, dval.p, n, 31 + q_bits);
    hipLaunchKernelGGL(qc_slot_head_kernel, GRID(n), 0, st, n, (unsigned long long *)dtmpk.p, (unsigned *)dval.p, (int64_t *)dss.p,
                       (int64_t *)dse.p, (int32_t *)dflag.p);
    QSCAN(dflag.p, n, dpos.p);
    int64_t nslots = 0;
    FCHK(hipMemcpy(&nslots, (int64_t *)dpos.p + n, 8, hipMemcpyDeviceToHost));
    FCHK(dslotstart.alloc((nslots + 2) * 8)); FCHK(dclid.alloc((n + 1) * 4)); FCHK(dncl.alloc((nslots + 1) * 4)); FCHK(dclbase.alloc((nslots + 2) * 8));
    hipLaunchKernelGGL(qc_starts_kernel, GRID(n), 0, st, n, (int32_t *)dflag.p, (int64_t *)dpos.p, (int64_t *)dslotstart.p);
    // 3: clusters
    {
        int64_t blocks = (nslots + 3) / 4; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(fm_cluster_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nslots, (int64_t *)dslotstart.p, (unsigned *)dval.p,
                           (int64_t *)dqe.p, (int64_t *)dss.p, (int64_t *)dse.p, sthr, 0, (int32_t *)dclid.p, (int32_t *)dncl.p,
                           all_chains ? (const int64_t *)dql.p : (const int64_t *)nullptr, (const int32_t *)dq.p);
    }
    {   // the scan scratch must cover nslots too (nslots <= n)
        QSCAN(dncl.p, nslots, dclbase.p);
    }
    int64_t ncl = 0;
    FCHK(hipMemcpy(&ncl, (int64_t *)dclbase.p + nslots, 8, hipMemcpyDeviceToHost));
    FCHK(dkqe.alloc((n + 1) * 8)); FCHK(dkqs.alloc((n + 1) * 8)); FCHK(dkcl.alloc((n + 1) * 8)); FCHK(dclcnt.alloc((ncl + 1) * 4));
    FCHK(dclstart.alloc((ncl + 2) * 8)); FCHK(dp2.alloc((n + 1) * 4)); FCHK(dorder2.alloc((n + 1) * 4));
    FCHK(hipMemset(dclcnt.p, 0, (ncl + 1) * 4));
    hipLaunchKernelGGL(qc_ckeys_kernel, GRID(n), 0, st, n, (unsigned *)dval.p, (int32_t *)dflag.p, (int64_t *)dpos.p, (int32_t *)dclid.p,
                       (int64_t *)dclbase.p, (int64_t *)dqs.p, (int64_t *)dqe.p, (unsigned long long *)dkqe.p, (unsigned long long *)dkqs.p,
                       (unsigned long long *)dkcl.p, (int32_t *)dclcnt.p);
    // 4: order inside the clusters by (q_start, q_end)
    hipLaunchKernelGGL(fm_iota_kernel, GRID(n), 0, st, n, (unsigned *)dp2.p);
    QSORT(dkqe.p, dp2.p, n, 32);
    hipLaunchKernelGGL(fm_permute_u64_kernel, GRID(n), 0, st, n, (unsigned *)dp2.p, (unsigned long long *)dkqs.p, (unsigned long long *)dtmpk.p);
    QSORT(dtmpk.p, dp2.p, n, 32);
    hipLaunchKernelGGL(fm_permute_u64_kernel, GRID(n), 0, st, n, (unsigned *)dp2.p, (unsigned long long *)dkcl.p, (unsigned long long *)dtmpk.p);
    int cl_bits = 1; while ((1ll << cl_bits) < ncl + 1) cl_bits++;
    QSORT(dtmpk.p, dp2.p, n, cl_bits);
    hipLaunchKernelGGL(fm_compose_kernel, GRID(n), 0, st, n, (unsigned *)dp2.p, (unsigned *)dval.p, (unsigned *)dorder2.p);
    QSCAN(dclcnt.p, ncl, dclstart.p);
    if (all_chains) {
        // 5': every chain of every cluster, in cluster order = (query, subject by first appearance, forward before reverse,
        // cluster, chain start); per-query counts -> CSR
        FBuf dch, disch, dchpos;
        FCHK(dvis.alloc(n + 16)); FCHK(dcanon.alloc((n + 1) * 4)); FCHK(dch.alloc((n + 1) * sizeof(ChainA))); FCHK(disch.alloc((n + 1) * 4));
        FCHK(dchpos.alloc((n + 2) * 8)); FCHK(dqcount.alloc(((size_t)nq + 1) * 4)); FCHK(dfirst.alloc(((size_t)nq + 2) * 8));
        FCHK(hipMemset(dqcount.p, 0, ((size_t)nq + 1) * 4));
        hipLaunchKernelGGL(ca_chain_kernel, GRID(ncl), 0, st, ncl, (int64_t *)dclstart.p, (unsigned *)dorder2.p, (int32_t *)dq.p, (int32_t *)dsg.p,
                           (int64_t *)dqs.p, (int64_t *)dqe.p, (int64_t *)dss.p, (int64_t *)dse.p, (const int64_t *)dql.p, (uint8_t *)dvis.p,
                           (int32_t *)dcanon.p, (ChainA *)dch.p, (int32_t *)disch.p, (int32_t *)dqcount.p);
        QSCAN(disch.p, n, dchpos.p);
        QSCAN(dqcount.p, (int64_t)nq, dfirst.p);
        FCHK(hipMemcpy(copy_first, dfirst.p, ((size_t)nq + 1) * 8, hipMemcpyDeviceToHost));
        const int64_t nch = copy_first[nq];
        *n_out = nch;
        if (nch > cap) { sorter_free(S); return HITE_ECAP; }
        if (nch > 0) {
            FBuf a1, a2, a3, a4, a5, a6;
            FCHK(a1.alloc(nch * 4)); FCHK(a2.alloc(nch * 8)); FCHK(a3.alloc(nch * 8)); FCHK(a4.alloc(nch * 8)); FCHK(a5.alloc(nch * 8)); FCHK(a6.alloc(nch * 4));
            hipLaunchKernelGGL(ca_emit_kernel, GRID(n), 0, st, n, (int32_t *)disch.p, (int64_t *)dchpos.p, (ChainA *)dch.p, (int32_t *)a1.p,
                               (int64_t *)a2.p, (int64_t *)a3.p, (int64_t *)a4.p, (int64_t *)a5.p, (int32_t *)a6.p);
            FCHK(hipGetLastError());
            FCHK(hipMemcpy(all_chains->o_sid, a1.p, nch * 4, hipMemcpyDeviceToHost)); FCHK(hipMemcpy(all_chains->o_qs, a2.p, nch * 8, hipMemcpyDeviceToHost));
            FCHK(hipMemcpy(all_chains->o_qe, a3.p, nch * 8, hipMemcpyDeviceToHost)); FCHK(hipMemcpy(all_chains->o_ss, a4.p, nch * 8, hipMemcpyDeviceToHost));
            FCHK(hipMemcpy(all_chains->o_se, a5.p, nch * 8, hipMemcpyDeviceToHost)); FCHK(hipMemcpy(all_chains->o_next, a6.p, nch * 4, hipMemcpyDeviceToHost));
        }
        sorter_free(S);
        return HITE_OK;
    }
    // 5: best chain per cluster
    FCHK(dvis.alloc(n + 16)); FCHK(dcanon.alloc((n + 1) * 4)); FCHK(dbest.alloc((ncl + 1) * sizeof(QBest))); FCHK(dlkey.alloc((ncl + 1) * 8));
    FCHK(dlval.alloc((ncl + 1) * 4)); FCHK(dqcount.alloc(((size_t)nq + 1) * 4)); FCHK(dqstart.alloc(((size_t)nq + 2) * 8));
    FCHK(dcount.alloc(((size_t)nq + 1) * 4)); FCHK(dfirst.alloc(((size_t)nq + 2) * 8));
    FCHK(hipMemset(dqcount.p, 0, ((size_t)nq + 1) * 4));
    hipLaunchKernelGGL(qc_chain_kernel, GRID(ncl), 0, st, ncl, (int64_t *)dclstart.p, (unsigned *)dorder2.p, (int32_t *)dq.p, (int32_t *)dsg.p,
                       (int64_t *)dqs.p, (int64_t *)dqe.p, (int64_t *)dss.p, (int64_t *)dse.p, (const double *)did.p, qthr, sthr,
                       (uint8_t *)dvis.p, (int32_t *)dcanon.p, (QBest *)dbest.p, (unsigned long long *)dlkey.p, (unsigned *)dlval.p,
                       (int32_t *)dqcount.p);
    // 6: per query, longest first (stable)
    QSORT(dlkey.p, dlval.p, ncl, 32 + q_bits);
    QSCAN(dqcount.p, (int64_t)nq, dqstart.p);
    // 7: selection -- count, scan, write
    const unsigned sblocks = (unsigned)((nq + 3) / 4);
    hipLaunchKernelGGL(qc_select_kernel<false>, dim3(sblocks), dim3(256), 0, st, (int)nq, (int64_t *)dqstart.p, (unsigned *)dlval.p, (QBest *)dbest.p,
                       (int64_t *)dql.p, (int64_t *)dsl.p, qcov, scov, (int)max_copy, (int32_t *)dcount.p, (int64_t *)nullptr, (int32_t *)nullptr,
                       (int64_t *)nullptr, (int64_t *)nullptr, (int64_t *)nullptr, (uint8_t *)nullptr);
    QSCAN(dcount.p, (int64_t)nq, dfirst.p);
    FCHK(hipMemcpy(copy_first, dfirst.p, ((size_t)nq + 1) * 8, hipMemcpyDeviceToHost));
    const int64_t nout = copy_first[nq];
    *n_out = nout;
    if (nout > cap) { sorter_free(S); return HITE_ECAP; }
    if (nout > 0) {
        if (!o_sid || !o_s || !o_e || !o_len || !o_minus) { sorter_free(S); return HITE_EINVAL; }
        FCHK(dos.alloc(nout * 4)); FCHK(doa.alloc(nout * 8)); FCHK(dob.alloc(nout * 8)); FCHK(dol.alloc(nout * 8)); FCHK(dom.alloc(nout));
        hipLaunchKernelGGL(qc_select_kernel<true>, dim3(sblocks), dim3(256), 0, st, (int)nq, (int64_t *)dqstart.p, (unsigned *)dlval.p,
                           (QBest *)dbest.p, (int64_t *)dql.p, (int64_t *)dsl.p, qcov, scov, (int)max_copy, (int32_t *)dcount.p,
                           (int64_t *)dfirst.p, (int32_t *)dos.p, (int64_t *)doa.p, (int64_t *)dob.p, (int64_t *)dol.p, (uint8_t *)dom.p);
        FCHK(hipGetLastError());
        FCHK(hipMemcpy(o_sid, dos.p, nout * 4, hipMemcpyDeviceToHost)); FCHK(hipMemcpy(o_s, doa.p, nout * 8, hipMemcpyDeviceToHost));
        FCHK(hipMemcpy(o_e, dob.p, nout * 8, hipMemcpyDeviceToHost)); FCHK(hipMemcpy(o_len, dol.p, nout * 8, hipMemcpyDeviceToHost));
        FCHK(hipMemcpy(o_minus, dom.p, nout, hipMemcpyDeviceToHost));
    }
    sorter_free(S);
    return HITE_OK;
}

extern "C" int hite_query_copies(hite_ctx *ctx, int64_t n, const int32_t *qid, const int32_t *sid, const int64_t *qs, const int64_t *qe,
                                 const int64_t *ss, const int64_t *se, const double *ident, int32_t nq, const int64_t *qlen, int32_t ns,
                                 const int64_t *slen, double qcov, double scov, int64_t qthr, int64_t sthr, int32_t max_copy, int64_t cap,
                                 int64_t *copy_first, int32_t *o_sid, int64_t *o_s, int64_t *o_e, int64_t *o_len, uint8_t *o_minus,
                                 int64_t *n_out) {
    return query_copies_impl(ctx, n, qid, sid, qs, qe, ss, se, ident, nq, qlen, ns, slen, qcov, scov, qthr, sthr, max_copy, cap, copy_first, o_sid,
                             o_s, o_e, o_len, o_minus, n_out, nullptr);
}

// every chain of every cluster (FMEA Util.py:10452, get_full_length_copies_from_blastn_v1 :5907): include/hite_gpu.h
extern "C" int hite_chain_all(hite_ctx *ctx, int64_t n, const int32_t *qid, const int32_t *sid, const int64_t *qs, const int64_t *qe,
                              const int64_t *ss, const int64_t *se, int32_t nq, int32_t ns, const int64_t *qgap, int64_t cap,
                              int64_t *chain_first, int32_t *o_sid, int64_t *o_qs, int64_t *o_qe, int64_t *o_ss, int64_t *o_se,
                              int32_t *o_next, int64_t *n_out) {
    if (!qgap || !chain_first || !n_out || (cap > 0 && (!o_sid || !o_qs || !o_qe || !o_ss || !o_se || !o_next))) return HITE_EINVAL;
    for (int32_t q = 0; q < nq; q++) if (qgap[q] < 0) return HITE_EINVAL;
    ChainAllOut O{o_sid, o_qs, o_qe, o_ss, o_se, o_next};
    return query_copies_impl(ctx, n, qid, sid, qs, qe, ss, se, nullptr, nq, qgap, ns, nullptr, 0.0, 0.0, 0, 0, 0, cap, chain_first, nullptr, nullptr,
                             nullptr, nullptr, nullptr, n_out, &O);
}

// =============================================================================================
// Library de-duplication (panHiTE merge): the arithmetic HiTE owns between its external tools.
//   hite_lib_chain      process_blast_results_in_chunks + process_chunk + extend_fragments
//                       (/root/reference/module/Util.py:12146-12200, 11958-12003, 11869-11944)
//   hite_lib_cluster    cluster_sequences_from_chunks (:12067-12115)  -- sequential by definition, host code
//   hite_msa_consensus  cons_from_mafft_v1 (:12515-12566), batch of alignments
// hite_lib_chain has the shape of the copy clustering above: chunk ids by a scan over the closing lines, first appearance
// of (chunk, query) and (chunk, query, subject) by stable sorts, slots ordered by those, one thread per slot for the
// extension sweep (every open fragment within reach is extended, the scan stops at the first one out of reach).
// =============================================================================================
struct LFrag { int qs, qe, ss, se; };

__global__ void lc_check_kernel(int64_t n, const int32_t *__restrict__ qid, const int32_t *__restrict__ sid, const int64_t *__restrict__ qs,
                                const int64_t *__restrict__ qe, const int64_t *__restrict__ ss, const int64_t *__restrict__ se, int nseq,
                                int64_t chunk_size, int32_t *__restrict__ keep, int32_t *__restrict__ bnd, int *__restrict__ err) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int q = qid[i], s = sid[i];
    bool bad = q < 0 || q >= nseq || s < 0 || s >= nseq || qs[i] < 0 || qe[i] < 0 || ss[i] < 0 || se[i] < 0 || qs[i] >= 0x7fffffff ||
               qe[i] >= 0x7fffffff || ss[i] >= 0x7fffffff || se[i] >= 0x7fffffff;
    if (bad) atomicExch(err, 1);
    const bool self = q == s && qs[i] == ss[i] && qe[i] == se[i];               // :12186
    keep[i] = !self && !bad;
    bnd[i] = !self && !bad && (i + 1) % chunk_size == 0;                         // :12193 (skipped lines `continue` past it)
}
// kept lines in file order: val = original index, key = (chunk, query)
__global__ void lc_qkey_kernel(int64_t n, const int32_t *__restrict__ keep, const int64_t *__restrict__ kpos, const int64_t *__restrict__ chunk,
                               const int32_t *__restrict__ qid, int q_bits, unsigned long long *__restrict__ key, unsigned *__restrict__ val) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    key[kpos[i]] = ((unsigned long long)chunk[i] << q_bits) | (unsigned)qid[i];
    val[kpos[i]] = (unsigned)i;
}
// second round: key = (first line of the (chunk, query) run, subject)
__global__ void lc_pkey_kernel(int64_t n, const int32_t *__restrict__ keep, const int64_t *__restrict__ kpos, const unsigned *__restrict__ fq,
                               const int32_t *__restrict__ sid, int q_bits, unsigned long long *__restrict__ key, unsigned *__restrict__ val) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    key[kpos[i]] = ((unsigned long long)fq[i] << q_bits) | (unsigned)sid[i];
    val[kpos[i]] = (unsigned)i;
}
__global__ void lc_keys_kernel(int64_t m, const unsigned *__restrict__ lines, const int64_t *__restrict__ ss, const int64_t *__restrict__ se,
                               const unsigned *__restrict__ fq, const unsigned *__restrict__ fp, unsigned long long *__restrict__ k_e,
                               unsigned long long *__restrict__ k_s, unsigned long long *__restrict__ k_g, unsigned *__restrict__ val) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const unsigned i = lines[j];
    const bool rev = ss[i] > se[i];                                                // forward: s_start <= s_end  :11985
    unsigned sk = rev ? (unsigned)(0x7fffffff - (int)ss[i]) : (unsigned)ss[i];
    unsigned ek = rev ? (unsigned)(0x7fffffff - (int)se[i]) : (unsigned)se[i];
    k_e[j] = ek;
    k_s[j] = ((unsigned long long)(rev ? 1 : 0) << 31) | sk;
    k_g[j] = ((unsigned long long)fq[i] << 31) | fp[i];
    val[j] = i;
}
// one thread per slot: extend_fragments (:11886-11944).  Long fragments are built in place at lf[slot_start ...].
__global__ void lc_extend_kernel(int64_t nslots, const int64_t *__restrict__ slot_start, const unsigned *__restrict__ order,
                                 const int32_t *__restrict__ qid, const int64_t *__restrict__ qs, const int64_t *__restrict__ qe,
                                 const int64_t *__restrict__ ss, const int64_t *__restrict__ se, const int64_t *__restrict__ seq_len,
                                 double one_minus_thr, LFrag *__restrict__ lf, int32_t *__restrict__ isrec) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nslots) return;
    const int64_t b = slot_start[g], e = slot_start[g + 1];
    const unsigned h0 = order[b];
    const bool fwd = ss[h0] <= se[h0];
    const double skip_gap = (double)seq_len[qid[h0]] * one_minus_thr;              // :11978
    int64_t nl = 0;
    for (int64_t t = b; t < e; t++) {
        const unsigned h = order[t];
        const int cqs = (int)qs[h], cqe = (int)qe[h], css = (int)ss[h], cse = (int)se[h];
        bool upd = false;
        for (int64_t u = nl - 1; u >= 0; u--) {
            LFrag p = lf[b + u];
            if (fwd) {
                if ((double)(css - p.se) >= skip_gap) break;
                if ((double)(cqs - p.qe) < skip_gap && cqe > p.qe && cse > p.se) {
                    p.qs = p.qs < cqs ? p.qs : cqs; p.qe = cqe; p.ss = p.ss < css ? p.ss : css; p.se = cse;
                    lf[b + u] = p; upd = true;
                }
            } else {
                if ((double)(p.se - css) >= skip_gap) break;
                if ((double)(cqs - p.qe) < skip_gap && cqe > p.qe && cse < p.se) {
                    p.qe = cqe; p.ss = p.ss > css ? p.ss : css; p.se = cse;
                    lf[b + u] = p; upd = true;
                }
            }
        }
        if (!upd) { LFrag p; p.qs = cqs; p.qe = cqe; p.ss = css; p.se = cse; lf[b + nl] = p; nl++; }
    }
    for (int64_t u = b; u < e; u++) isrec[u] = u - b < nl;
}
__global__ void lc_emit_kernel(int64_t m, const int32_t *__restrict__ isrec, const int64_t *__restrict__ pos, const int32_t *__restrict__ sflag,
                               const int64_t *__restrict__ spos, const int64_t *__restrict__ slot_start, const unsigned *__restrict__ order,
                               const LFrag *__restrict__ lf, const int64_t *__restrict__ chunk, const int32_t *__restrict__ qid,
                               const int32_t *__restrict__ sid, int64_t cap, int32_t *__restrict__ o_chunk, int32_t *__restrict__ o_q,
                               int64_t *__restrict__ o_qs, int64_t *__restrict__ o_qe, int32_t *__restrict__ o_s, int64_t *__restrict__ o_ss,
                               int64_t *__restrict__ o_se) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m || !isrec[i]) return;
    const int64_t o = pos[i];
    if (o >= cap) return;
    const int64_t slot = spos[i] + sflag[i] - 1;
    const unsigned h0 = order[slot_start[slot]];
    const LFrag p = lf[i];
    o_chunk[o] = (int32_t)chunk[h0]; o_q[o] = qid[h0]; o_s[o] = sid[h0];
    o_qs[o] = (int64_t)p.qs - 1; o_qe[o] = p.qe; o_ss[o] = (int64_t)p.ss - 1; o_se[o] = p.se;     // :11996-11999
}

extern "C" int hite_lib_chain(hite_ctx *ctx, int64_t n, const int32_t *qid, const int32_t *sid, const int64_t *qs, const int64_t *qe,
                              const int64_t *ss, const int64_t *se, int32_t nseq, const int64_t *seq_len, double threshold,
                              int64_t chunk_size, int64_t cap, int32_t *o_chunk, int32_t *o_q, int64_t *o_qs, int64_t *o_qe, int32_t *o_s,
                              int64_t *o_ss, int64_t *o_se, int64_t *n_out) {
    FArenaScope arena_scope(ctx);
    if (!ctx || n < 0 || n >= 0x7fffffff || nseq <= 0 || !seq_len || !n_out) return HITE_EINVAL;
    *n_out = 0;
    if (n == 0) return HITE_OK;
    if (chunk_size <= 0) chunk_size = n;
    int q_bits = 1; while ((1ll << q_bits) < (long long)nseq + 1) q_bits++;
    int c_bits = 1; while ((1ll << c_bits) < n / chunk_size + 2) c_bits++;
    if (c_bits + q_bits > 63 || 31 + q_bits > 63) return HITE_EINVAL;
    if (hipSetDevice(ctx->device) != hipSuccess) return HITE_EHIP;
    hipStream_t st = nullptr;
    Sorter S;
    FBuf dq, dsg, dqs, dqe, dss, dse, dsl, derr, dbs, dkeep, dbnd, dkpos, dchunk, dkey, dval, dflag, dpos, dgf, dfq, dfp, dke, dks, dkg, dtmpk,
        dslotstart, dlf, disrec, dopos, doc, doq, dos, doqs, doqe, doss, dose;
    FCHK(dq.up(qid, n * 4)); FCHK(dsg.up(sid, n * 4)); FCHK(dqs.up(qs, n * 8)); FCHK(dqe.up(qe, n * 8)); FCHK(dss.up(ss, n * 8));
    FCHK(dse.up(se, n * 8)); FCHK(dsl.up(seq_len, (size_t)nseq * 8));
    FCHK(derr.alloc(16)); FCHK(dbs.alloc((size_t)scan_tmp_elems(n + 16) * 8)); FCHK(dkeep.alloc((n + 1) * 4)); FCHK(dbnd.alloc((n + 1) * 4));
    FCHK(dkpos.alloc((n + 2) * 8)); FCHK(dchunk.alloc((n + 2) * 8));
    FCHK(hipMemset(derr.p, 0, 16));
    hipLaunchKernelGGL(lc_check_kernel, GRID(n), 0, st, n, (int32_t *)dq.p, (int32_t *)dsg.p, (int64_t *)dqs.p, (int64_t *)dqe.p, (int64_t *)dss.p,
                       (int64_t *)dse.p, (int)nseq, chunk_size, (int32_t *)dkeep.p, (int32_t *)dbnd.p, (int *)derr.p);
    int herr = 0;
    FCHK(hipMemcpy(&herr, derr.p, 4, hipMemcpyDeviceToHost));
    if (herr) return HITE_EINVAL;   // id or coordinate out of range
    QSCAN(dkeep.p, n, dkpos.p);
    QSCAN(dbnd.p, n, dchunk.p);     // chunk of line i = closing lines before it
    int64_t m = 0;
    FCHK(hipMemcpy(&m, (int64_t *)dkpos.p + n, 8, hipMemcpyDeviceToHost));
    if (m == 0) return HITE_OK;
    FCHK(dkey.alloc((m + 1) * 8)); FCHK(dval.alloc((m + 1) * 4)); FCHK(dflag.alloc((m + 1) * 4)); FCHK(dpos.alloc((m + 2) * 8));
    FCHK(dgf.alloc((m + 1) * 4)); FCHK(dfq.alloc((n + 1) * 4)); FCHK(dfp.alloc((n + 1) * 4)); FCHK(dke.alloc((m + 1) * 8));
    FCHK(dks.alloc((m + 1) * 8)); FCHK(dkg.alloc((m + 1) * 8)); FCHK(dtmpk.alloc((m + 1) * 8));
    if (sorter_init(S, ctx, st, m)) { sorter_free(S); return HITE_EHIP; }
    // first appearance of every (chunk, query), then of every (chunk, query, subject)
    hipLaunchKernelGGL(lc_qkey_kernel, GRID(n), 0, st, n, (int32_t *)dkeep.p, (int64_t *)dkpos.p, (int64_t *)dchunk.p, (int32_t *)dq.p, q_bits,
                       (unsigned long long *)dkey.p, (unsigned *)dval.p);
    QSORT(dkey.p, dval.p, m, c_bits + q_bits);
    hipLaunchKernelGGL(qc_head_kernel, GRID(m), 0, st, m, (unsigned long long *)dkey.p, (int32_t *)dflag.p);
    QSCAN(dflag.p, m, dpos.p);
    hipLaunchKernelGGL(qc_run_first_kernel, GRID(m), 0, st, m, (int32_t *)dflag.p, (int64_t *)dpos.p, (unsigned *)dval.p, (unsigned *)dgf.p);
    hipLaunchKernelGGL(qc_first_kernel, GRID(m), 0, st, m, (int32_t *)dflag.p, (int64_t *)dpos.p, (unsigned *)dval.p, (unsigned *)dgf.p,
                       (unsigned *)dfq.p);
    hipLaunchKernelGGL(lc_pkey_kernel, GRID(n), 0, st, n, (int32_t *)dkeep.p, (int64_t *)dkpos.p, (unsigned *)dfq.p, (int32_t *)dsg.p, q_bits,
                       (unsigned long long *)dkey.p, (unsigned *)dval.p);
    QSORT(dkey.p, dval.p, m, 31 + q_bits);
    hipLaunchKernelGGL(qc_head_kernel, GRID(m), 0, st, m, (unsigned long long *)dkey.p, (int32_t *)dflag.p);
    QSCAN(dflag.p, m, dpos.p);
    hipLaunchKernelGGL(qc_run_first_kernel, GRID(m), 0, st, m, (int32_t *)dflag.p, (int64_t *)dpos.p, (unsigned *)dval.p, (unsigned *)dgf.p);
    hipLaunchKernelGGL(qc_first_kernel, GRID(m), 0, st, m, (int32_t *)dflag.p, (int64_t *)dpos.p, (unsigned *)dval.p, (unsigned *)dgf.p,
                       (unsigned *)dfp.p);
    // slots = ((chunk, query) by first line, subject by first line, strand), sorted by strand-aware (s_start, s_end).
    // dval currently lists the kept lines (some order): rebuild it in file order through kpos for stable sorts.
    hipLaunchKernelGGL(lc_qkey_kernel, GRID(n), 0, st, n, (int32_t *)dkeep.p, (int64_t *)dkpos.p, (int64_t *)dchunk.p, (int32_t *)dq.p, q_bits,
                       (unsigned long long *)dkey.p, (unsigned *)dgf.p);          // dgf = kept lines in file order (dkey is scratch here)
    hipLaunchKernelGGL(lc_keys_kernel, GRID(m), 0, st, m, (unsigned *)dgf.p, (int64_t *)dss.p, (int64_t *)dse.p, (unsigned *)dfq.p,
                       (unsigned *)dfp.p, (unsigned long long *)dke.p, (unsigned long long *)dks.p, (unsigned long long *)dkg.p,
                       (unsigned *)dval.p);
    // the keys are indexed by compacted position j; sort positions, compose with the line list at the end
    hipLaunchKernelGGL(fm_iota_kernel, GRID(m), 0, st, m, (unsigned *)dval.p);
    QSORT(dke.p, dval.p, m, 32);
    hipLaunchKernelGGL(fm_permute_u64_kernel, GRID(m), 0, st, m, (unsigned *)dval.p, (unsigned long long *)dks.p, (unsigned long long *)dtmpk.p);
    QSORT(dtmpk.p, dval.p, m, 32);
    hipLaunchKernelGGL(fm_permute_u64_kernel, GRID(m), 0, st, m, (unsigned *)dval.p, (unsigned long long *)dkg.p, (unsigned long long *)dtmpk.p);
    QSORT(dtmpk.p, dval.p, m, 62);
    FCHK(dslotstart.alloc((m + 2) * 8));   // order = lines[val]
    {
        FBuf dorder;
        FCHK(dorder.alloc((m + 1) * 4));
        hipLaunchKernelGGL(fm_compose_kernel, GRID(m), 0, st, m, (unsigned *)dval.p, (unsigned *)dgf.p, (unsigned *)dorder.p);
        FCHK(hipMemcpyAsync(dval.p, dorder.p, (size_t)m * 4, hipMemcpyDeviceToDevice, st));
        FCHK(hipStreamSynchronize(st));
    }
    hipLaunchKernelGGL(qc_slot_head_kernel, GRID(m), 0, st, m, (unsigned long long *)dtmpk.p, (unsigned *)dval.p, (int64_t *)dss.p,
                       (int64_t *)dse.p, (int32_t *)dflag.p);
    QSCAN(dflag.p, m, dpos.p);
    int64_t nslots = 0;
    FCHK(hipMemcpy(&nslots, (int64_t *)dpos.p + m, 8, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(qc_starts_kernel, GRID(m), 0, st, m, (int32_t *)dflag.p, (int64_t *)dpos.p, (int64_t *)dslotstart.p);
    FCHK(dlf.alloc((m + 1) * sizeof(LFrag))); FCHK(disrec.alloc((m + 1) * 4)); FCHK(dopos.alloc((m + 2) * 8));
    hipLaunchKernelGGL(lc_extend_kernel, GRID(nslots), 0, st, nslots, (int64_t *)dslotstart.p, (unsigned *)dval.p, (int32_t *)dq.p,
                       (int64_t *)dqs.p, (int64_t *)dqe.p, (int64_t *)dss.p, (int64_t *)dse.p, (int64_t *)dsl.p, 1 - threshold,
                       (LFrag *)dlf.p, (int32_t *)disrec.p);
    QSCAN(disrec.p, m, dopos.p);
    int64_t nout = 0;
    FCHK(hipMemcpy(&nout, (int64_t *)dopos.p + m, 8, hipMemcpyDeviceToHost));
    *n_out = nout;
    if (nout > cap) { sorter_free(S); return HITE_ECAP; }
    if (nout > 0) {
        if (!o_chunk || !o_q || !o_qs || !o_qe || !o_s || !o_ss || !o_se) { sorter_free(S); return HITE_EINVAL; }
        FCHK(doc.alloc(nout * 4)); FCHK(doq.alloc(nout * 4)); FCHK(dos.alloc(nout * 4)); FCHK(doqs.alloc(nout * 8)); FCHK(doqe.alloc(nout * 8));
        FCHK(doss.alloc(nout * 8)); FCHK(dose.alloc(nout * 8));
        hipLaunchKernelGGL(lc_emit_kernel, GRID(m), 0, st, m, (int32_t *)disrec.p, (int64_t *)dopos.p, (int32_t *)dflag.p, (int64_t *)dpos.p,
                           (int64_t *)dslotstart.p, (unsigned *)dval.p, (LFrag *)dlf.p, (int64_t *)dchunk.p, (int32_t *)dq.p, (int32_t *)dsg.p,
                           nout, (int32_t *)doc.p, (int32_t *)doq.p, (int64_t *)doqs.p, (int64_t *)doqe.p, (int32_t *)dos.p, (int64_t *)doss.p,
                           (int64_t *)dose.p);
        FCHK(hipGetLastError());
        FCHK(hipMemcpy(o_chunk, doc.p, nout * 4, hipMemcpyDeviceToHost)); FCHK(hipMemcpy(o_q, doq.p, nout * 4, hipMemcpyDeviceToHost));
        FCHK(hipMemcpy(o_s, dos.p, nout * 4, hipMemcpyDeviceToHost)); FCHK(hipMemcpy(o_qs, doqs.p, nout * 8, hipMemcpyDeviceToHost));
        FCHK(hipMemcpy(o_qe, doqe.p, nout * 8, hipMemcpyDeviceToHost)); FCHK(hipMemcpy(o_ss, doss.p, nout * 8, hipMemcpyDeviceToHost));
        FCHK(hipMemcpy(o_se, dose.p, nout * 8, hipMemcpyDeviceToHost));
    }
    sorter_free(S);
    return HITE_OK;
}

// cluster_sequences_from_chunks (:12067-12115): greedy, every decision depends on the set built so far -> host loop.
// Records as hite_lib_chain emits them; a (chunk, query) run is a maximal stretch of equal (chunk, q).
extern "C" int hite_lib_cluster(int64_t nrec, const int32_t *chunk, const int32_t *q, const int64_t *qs, const int64_t *qe, const int32_t *s,
                                const int64_t *ss, const int64_t *se, int32_t nseq, const int64_t *seq_len, double threshold, int64_t cap_cl,
                                int64_t cap_mem, int64_t *cl_first, int32_t *members, int64_t *n_cl) {
    if (nrec < 0 || nseq <= 0 || !seq_len || !cl_first || !n_cl || (nrec > 0 && (!chunk || !q || !qs || !qe || !s || !ss || !se || !members)))
        return HITE_EINVAL;
    std::vector<uint8_t> redundant((size_t)nseq, 0);
    int64_t ncl = 0, nm = 0;
    cl_first[0] = 0;
    *n_cl = 0;
    for (int64_t i = 0; i < nrec;) {
        int64_t j = i;
        while (j < nrec && chunk[j] == chunk[i] && q[j] == q[i]) j++;
        const int query = q[i];
        if (query < 0 || query >= nseq) return HITE_EINVAL;
        if (!redundant[query]) {                                                          // :12084
            if (ncl >= cap_cl || nm >= cap_mem) return HITE_ECAP;
            members[nm++] = query;
            for (int64_t t = i; t < j; t++) {
                const int sub = s[t];
                if (sub < 0 || sub >= nseq) return HITE_EINVAL;
                if (redundant[sub]) continue;                                             // :12098
                const int64_t ql = qe[t] > qs[t] ? qe[t] - qs[t] : qs[t] - qe[t], sl = se[t] > ss[t] ? se[t] - ss[t] : ss[t] - se[t];
                if ((double)ql / (double)seq_len[query] >= threshold || (double)sl / (double)seq_len[sub] >= threshold) {   // :12064
                    redundant[sub] = 1;
                    if (sub != query) {
                        if (nm >= cap_mem) return HITE_ECAP;
                        members[nm++] = sub;
                    }
                }
            }
            cl_first[++ncl] = nm;
        }
        i = j;
    }
    *n_cl = ncl;
    return HITE_OK;
}

// cons_from_mafft_v1 (:12515-12566): one block per alignment; a column keeps its most frequent non-gap character if that
// count exceeds R / 2.  A strict majority over all R entries is the Boyer-Moore candidate of the column ('-' voting as an
// ordinary symbol); a second pass over the rows counts it.  Kept columns are compacted in order with a block scan.
__global__ void __launch_bounds__(256) msa_consensus_kernel(int nmat, const int32_t *__restrict__ rows, const int64_t *__restrict__ cols,
                                                            const int64_t *__restrict__ mat_off, const uint8_t *__restrict__ mats,
                                                            const int64_t *__restrict__ out_off, uint8_t *__restrict__ cons,
                                                            int64_t *__restrict__ cons_len) {
    __shared__ int s_scan[256];
    __shared__ int64_t s_base;
    const int a = blockIdx.x;
    const int R = rows[a];
    const int64_t C = cols[a];
    const uint8_t *mat = mats + mat_off[a];
    uint8_t *out = cons + out_off[a];
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < C; c0 += 256) {
        const int64_t c = c0 + threadIdx.x;
        int keep = 0;
        uint8_t ch = 0;
        if (c < C) {
            int cand = -1, cnt = 0;
            for (int r = 0; r < R; r++) {
                const int x = mat[(int64_t)r * C + c];
                if (cnt == 0) { cand = x; cnt = 1; } else if (x == cand) cnt++; else cnt--;
            }
            if (cand >= 0 && cand != '-') {
                int tot = 0;
                for (int r = 0; r < R; r++) tot += mat[(int64_t)r * C + c] == cand;
                keep = tot > R / 2;                                                      // :12559
                ch = (uint8_t)cand;
            }
        }
        s_scan[threadIdx.x] = keep;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            int v = threadIdx.x >= d ? s_scan[threadIdx.x - d] : 0;
            __syncthreads();
            s_scan[threadIdx.x] += v;
            __syncthreads();
        }
        const int64_t base = s_base;
        if (keep) out[base + s_scan[threadIdx.x] - 1] = ch;
        __syncthreads();
        if (threadIdx.x == 255) s_base = base + s_scan[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) cons_len[a] = s_base;
}

extern "C" int hite_msa_consensus(hite_ctx *ctx, int32_t nmat, const int32_t *rows, const int64_t *cols, const int64_t *mat_off,
                                  const uint8_t *mats, const int64_t *out_off, uint8_t *cons, int64_t *cons_len) {
    FArenaScope arena_scope(ctx);
    if (!ctx || nmat < 0 || (nmat > 0 && (!rows || !cols || !mat_off || !mats || !out_off || !cons || !cons_len))) return HITE_EINVAL;
    if (nmat == 0) return HITE_OK;
    int64_t total_out = 0;
    for (int a = 0; a < nmat; a++) {
        if (rows[a] <= 0 || cols[a] < 0 || mat_off[a + 1] - mat_off[a] != (int64_t)rows[a] * cols[a] || out_off[a] < 0) return HITE_EINVAL;
        if (out_off[a] + cols[a] > total_out) total_out = out_off[a] + cols[a];
    }
    if (hipSetDevice(ctx->device) != hipSuccess) return HITE_EHIP;
    hipStream_t st = nullptr;
    Sorter S;   // (FCHK frees it)
    FBuf dr, dc, dmo, dm, doo, dcons, dlen;
    FCHK(dr.up(rows, (size_t)nmat * 4)); FCHK(dc.up(cols, (size_t)nmat * 8)); FCHK(dmo.up(mat_off, ((size_t)nmat + 1) * 8));
    FCHK(dm.up(mats, (size_t)mat_off[nmat])); FCHK(doo.up(out_off, (size_t)nmat * 8)); FCHK(dcons.alloc((size_t)total_out + 16));
    FCHK(dlen.alloc((size_t)nmat * 8));
    hipLaunchKernelGGL(msa_consensus_kernel, dim3((unsigned)nmat), dim3(256), 0, st, (int)nmat, (int32_t *)dr.p, (int64_t *)dc.p, (int64_t *)dmo.p,
                       (uint8_t *)dm.p, (int64_t *)doo.p, (uint8_t *)dcons.p, (int64_t *)dlen.p);
    FCHK(hipGetLastError());
    FCHK(hipMemcpy(cons_len, dlen.p, (size_t)nmat * 8, hipMemcpyDeviceToHost));
    if (total_out > 0) FCHK(hipMemcpy(cons, dcons.p, (size_t)total_out, hipMemcpyDeviceToHost));
    return HITE_OK;
}
