// hite_fill.h -- one row of the sparse star alignment (the kept columns only) from windows + ops + layout words: the body
// star_fill_sparse_kernel (hite_msa.hip) runs per (candidate, row), shared with the judge kernels that build their alignment in
// LDS instead of reading it back from HBM (hite_judge_team.inc, JT_LDS_MSA).  Item (r, p) owns the kept prefix of insertion
// block p, the centre column p if kept and (p == m) the extra last column: every output byte is written exactly once.
#pragma once
#include "hite_common.h"

// >>> fill_sparse_row (tests/test_host_compiled.py compiles this block for the host: rows rebuilt from the twin's ops and a layout derived
// from the twin's alignment must equal the twin's sparse alignment)
#ifndef FILL_U
#define FILL_U 4        // positions in flight per thread (the default form)
#endif

// Out: pointer to the output row (global or LDS bytes); t = index of the thread among the TS threads that share the row;
// NU = positions in flight per thread and trip.  The chain of a cell is three dependent loads deep (layout word, op -> base ->
// store) and that latency is what bounds the kernel -- a quarter of the memory instructions (round 3) or a fifth of the ops bytes
// (round 6, profiles/r06_fill_descriptors.txt) do not make it faster, more cells in flight do: on C3's full-length pass (rows of
// ~2 500 positions) 2 / 4 / 8 per thread take 8.1 / 6.6 / 5.7 ms, on the first500 + last500 pass (rows of ~1 100) 2.3 / 1.9 / 2.1
// (profiles/r06_fill_unroll.txt): star_fill_sparse_kernel runs with 8 when the launch holds long windows, with 4 otherwise.
// the columns of one position that are NOT its centre column: the kept prefix of its insertion block (kw columns) and, behind the
// last position, the extra last column; w = the position's layout word, oc = its op (the centre row: the position itself)
template <class Out>
__device__ __forceinline__ void fill_sparse_block(Out row, const uint8_t *__restrict__ b, int nrow, const uint16_t *__restrict__ rop,
                                                  int m, int le, bool centre, int p, unsigned w, unsigned oc) {
    const int kw = (int)(w & 0x7fffu), bs = (int)(w >> 16);
    const bool ex = p == m && le >= 0;
    if (p > m || (kw == 0 && !ex)) return;
    int ins = 0, q = p;
    if (!centre) {
        q = p < m ? (int)(oc & 0x7fffu) : nrow;
        const unsigned op = rop[p - 1];       // (p == 0: the op "before position 0" in the spare entry of the ops row above, hite_msa.hip)
        ins = q - ((int)(op & 0x7fff) + ((op >> 15) ? 0 : 1));
    }
    const int rp = q - ins;  // first inserted base
    for (int k = 0; k < kw; k++) row[bs + k] = k < ins ? b[rp + k] : (uint8_t)'-';
    if (ex) row[bs + kw] = le < ins ? b[rp + le] : (uint8_t)'-';
}
// what fill_sparse_row does with such a position: FillInline fills it on the spot; star_fill_sparse_kernel hands in a sink that
// takes the position into a list of its workgroup (FillDefer, hite_msa.hip) -- the blocks are rare (0.5 % of the positions), but
// with 512 cells per wavefront and trip nearly every trip held one and paid its two extra dependent loads: 1.8 of the kernel's
// 7.6 ms per C3 step (round 6; profiles/r06_fill_unroll.txt)
struct FillInline {
    __device__ __forceinline__ bool push(int) const { return false; }
};
template <class Out, int TS, int NU = FILL_U, class Sink = FillInline>
__device__ __forceinline__ void fill_sparse_row(Out row, const uint8_t *__restrict__ b, int nrow, const uint16_t *__restrict__ rop,
                                                const uint32_t *__restrict__ lay, int m, int le, bool centre, int t, Sink sink = Sink()) {
    for (int p0 = t; p0 <= m; p0 += NU * TS) {
        // the common position keeps its centre column and nothing else: layout word, op, base, one store -- straight-line
        // code for NU positions, their loads issued level by level.  Kept insertion columns and the extra last column
        // are rare and leave through one branch at the end (every branch that a wave takes for one of its lanes costs all 64)
        // (every load is unconditional, from a clamped index, and the value is masked afterwards: a load under a condition
        // compiles to a branch around it, and the loads of one level then wait for each other instead of flying together)
        unsigned w[NU], oc[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int p = p0 + u * TS;
            const unsigned wl = lay[p <= m ? p : m];
            const unsigned ol = rop[p < m ? p : (m > 0 ? m - 1 : 0)];      // (m == 0: entry 0 of the padded slot, masked below)
            w[u] = p <= m ? wl : 0u;
            oc[u] = centre ? (unsigned)p : (p < m ? ol : 0x8000u);
        }
        uint8_t ch[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const bool base = ((w[u] >> 15) & 1u) && !(oc[u] >> 15);
            const uint8_t cb = b[base ? (oc[u] & 0x7fffu) : 0u];
            ch[u] = base ? cb : (uint8_t)'-';
        }
        bool rare = false;
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const unsigned kw = w[u] & 0x7fffu;
            if ((w[u] >> 15) & 1u) row[(w[u] >> 16) + kw] = ch[u];
            rare = rare || kw != 0u || (p0 + u * TS == m && le >= 0);
        }
        if (rare) {
#pragma unroll 1
            for (int u = 0; u < NU; u++) {
                const int p = p0 + u * TS;
                if (p > m || ((w[u] & 0x7fffu) == 0u && !(p == m && le >= 0))) continue;
                if (!sink.push(p)) fill_sparse_block<Out>(row, b, nrow, rop, m, le, centre, p, w[u], oc[u]);
            }
        }
    }
}
// <<< fill_sparse_row
