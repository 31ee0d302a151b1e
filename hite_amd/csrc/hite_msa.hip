// hite_msa.hip -- star alignment of the copy windows of each candidate: this build's GPU-native
// stage at the point where the reference shells out to `mafft` (Util.py:10416, third-party,
// unpinned, absent -> parity unpinned; the pinned twin is oracle/hite_oracle_msa.c, byte-exact).
//
// Definition: see the header comment of oracle/hite_oracle_msa.c (identical scoring, stored form,
// band rule and tie-breaks).  Mapping to CDNA4:
//   * one wavefront per (row, centre) pair from a dynamic queue; the 64 lanes ARE the adaptive band:
//     lane k owns the cell i = t + k of anti-diagonal s = i + j.  The step is hand-scheduled inline
//     asm (10-11 vector instructions): neighbours by DPP wave_shl / wave_shr, the bases of a chunk in
//     per-wave LDS windows, one v_max3 on tagged scores for value + tie-break + direction;
//   * the per-cell direction (2 bits) is packed per lane by v_alignbit: one dword per lane per 16
//     anti-diagonals (256 B per wave, coalesced), the band moves as 2 bits per step in an SGPR;
//   * the traceback is a scalar walk (v_readlane of the direction word, bit tests, v_writelane); it
//     emits per centre position the aligned row position | gap flag << 15, 2 B per position;
//   * layout / fill kernels turn those ops into the rows x cols matrix -- in the pipeline fused with
//     remove_sparse_col_in_align_file, so that only the surviving columns are ever written.
// The DP is bound by vector-instruction issue (report cells/s); layout + fill are HBM / latency bound.
#include "hite_common.h"

#define MW 64
#define FILL_U 4        // items in flight per thread in the fill kernel
#define MSA_MAXR 128   // rows whose lengths are cached in LDS by the layout / fill kernels
#define MBIAS (1 << 28)
#define SC_MATCH 2
#define SC_MIS (-2)
#define SC_GAP (-4)

struct MsaParams {
    int n;                     // candidates
    int64_t total_rows;
    const uint8_t *win;
    const int64_t *win_off;    // total_rows : start of each window
    const int32_t *win_len;    // total_rows : length of each window
    const int32_t *row_first;  // n + 1
    const int64_t *ops_base;   // n + 1 : exclusive scan of (R_c + 1) * (m_c + 1)
    uint16_t *ops;
    int32_t *cols_out;         // n
    int32_t *status;           // n : 0 ok, 1 failed (path left the band / too wide)
    uint8_t *tb;               // traceback scratch, per wave slot
    size_t tb_slot;            // bytes per slot
    int max_steps;
    unsigned int *counter;
};

__device__ __forceinline__ int find_cand(const int32_t *__restrict__ row_first, int n, int64_t g) {
    int lo = 0, hi = n;  // largest c with row_first[c] <= g
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if ((int64_t)row_first[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// cross-lane helpers (wave64).  wave_shl:1 -> lane i reads lane i+1, wave_shr:1 -> lane i reads lane i-1;
// lanes without a source keep `fill`.
__device__ __forceinline__ int from_next_lane(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ int from_prev_lane(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
// same with 0 shifted in (bound_ctrl): scores are biased by 2^28 so that 0 is "minus infinity"
__device__ __forceinline__ int from_next_lane0(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ int from_prev_lane0(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
    unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// base words of the aligner (see star_align_kernel): the centre base as a v_perm_b32 byte selector, the row base as its
// 4-entry score row.  code = (c >> 1) & 3 is distinct for A, C, G, T; every other byte scores "mismatch" against anything.
__device__ __forceinline__ bool base_acgt(unsigned c) { return c - 0x41u < 32u && ((0x00080045u >> (c - 0x41u)) & 1u); }
__device__ __forceinline__ unsigned base_sel(unsigned c) { return 0x0c0c0c00u | (base_acgt(c) ? ((c >> 1) & 3u) : 4u); }
__device__ __forceinline__ unsigned base_row(unsigned c) { return 0x1a1a1a1au + (base_acgt(c) ? (0x10u << (8u * ((c >> 1) & 3u))) : 0u); }
// pin a wave-uniform value into an SGPR (the compiler otherwise keeps loop-carried uniform values in VGPRs)
__device__ __forceinline__ int to_sgpr(int v) {
    int r;
    asm volatile("s_mov_b32 %0, %1" : "=s"(r) : "s"(__builtin_amdgcn_readfirstlane(v)));
    return r;
}
// one wave-uniform grab from a global work counter without a divergent branch: lane 0 alone issues the
// atomic (exec = 1), the old value comes back in an SGPR.  Keeping the queue pop branch-free lets the
// compiler see the whole pair loop as uniform control flow (band bookkeeping stays on the scalar unit).
__device__ __forceinline__ unsigned wave_grab(unsigned *counter) {
    unsigned vret, sret, one = 1;
    unsigned long long saved;
    asm volatile(
        "s_mov_b64 %1, exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "global_atomic_add %0, %3, %4, off sc0\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_mov_b64 exec, %1\n\t"
        "s_nop 4\n\t"
        "v_readfirstlane_b32 %2, %0"
        : "=&v"(vret), "=&s"(saved), "=s"(sret)
        : "v"(counter), "v"(one)
        : "memory");
    return sret;
}
__device__ __forceinline__ unsigned long long rfl64(unsigned long long v) {
    unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// One wavefront per (row, centre) pair.  Issue budget (SQ counters, profiles/r01_sq_counters.txt): the vector unit is
// ~96 % busy at 4 cycles per wave64 integer / DPP instruction, so the code below is written to the vector-instruction count.
//  forward : lane k owns cell i = t + k of anti-diagonal s; 64 anti-diagonals per chunk.  Per step, depending on the
//            (uniform) move: one DPP shift gives the left / up neighbour (the diagonal operand of the next step is this
//            step's "left" operand: the two registers swap roles, no copy), the base that changes is re-read from the
//            per-wave LDS window, one v_max3 on the tagged scores gives value, tie-break and direction, v_alignbit packs
//            the direction.  Chunks that cannot touch the band clamps run 64 steps unrolled; the others go word by word
//            through a general loop with the clamps on the scalar unit.
//  backward: scalar walk (i-1 in M0, j-1, k, bit index in SGPRs), four 16-step words per trip; results leave as one
//            masked store of <= 64 u16 per trip (row position aligned to centre position p | gap flag << 15).
__global__ void __launch_bounds__(256) star_align_kernel(MsaParams P) {
    __shared__ unsigned s_bases[4][2][128];  // per wave: windows of centre / row base words for the current chunk
    const int lane = threadIdx.x & 63;
    const int wslot = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint8_t *slot = P.tb + (size_t)wslot * P.tb_slot;
    unsigned *tbd = reinterpret_cast<unsigned *>(slot);                                     // [(max_steps/16 + 2) * 64] direction words
    unsigned *tbm = reinterpret_cast<unsigned *>(slot + (size_t)((P.max_steps >> 4) + 2) * 256);  // [max_steps/16 + 2] move bits
    for (;;) {
        const unsigned int gq = wave_grab(P.counter);
        if ((int64_t)gq >= P.total_rows) break;
        const int64_t g = gq;
        const int c = __builtin_amdgcn_readfirstlane(find_cand(P.row_first, P.n, g));
        const int64_t g0 = __builtin_amdgcn_readfirstlane(P.row_first[c]);
        if (g == g0) continue;  // the centre itself
        const uint8_t *a = (const uint8_t *)rfl64((unsigned long long)(P.win + P.win_off[g0]));
        const int m = to_sgpr(P.win_len[g0]);
        const uint8_t *b = (const uint8_t *)rfl64((unsigned long long)(P.win + P.win_off[g]));
        const int n = to_sgpr(P.win_len[g]);
        uint16_t *ops = (uint16_t *)rfl64((unsigned long long)(P.ops + P.ops_base[c] + (int64_t)(g - g0) * (m + 1)));
        const int steps = m + n;
        if (m <= 0 || n <= 0 || steps > P.max_steps) { if (lane == 0) atomicExch(&P.status[c], 1); continue; }

        // ---------------- forward ----------------
        // Stored form of a score (see oracle/hite_oracle_msa.c): shifted by +4 per anti-diagonal (gap 0, mismatch +6,
        // match +10), times 4, low two bits = tag of the operand that won (2 diagonal, 1 up, 0 left).  With the up
        // operand stored with tag 1 and the left operand taken as "- 1", ONE v_max3 yields value, tie-break
        // (diag >= up >= left) and direction; v_alignbit shifts the two direction bits into the per-lane chunk word.
        // Bases are held as words that make the substitution score ONE v_perm_b32: the centre base is a byte selector
        // (A, C, G, T -> 0..3 picks a byte of the row word, anything else -> 4 picks the constant), the row base is the
        // 4-entry score row (42 in its own byte, 26 elsewhere; 26 everywhere for a non-ACGT byte).
        int t = -32;                                   // origin of anti-diagonal s-1 (scalar)
        int prev = lane == 32 ? ((MBIAS << 2) | 1) : 0, pp = 0;   // H(s-1) at origin t (tag 1);  left operand of step s-1
        int areg, breg;                                // selector of a[i-1], score row of b[j-1] of this lane's cell on anti-diagonal s-1
        {
            int ia = t + lane - 1, jb = -(t + lane) - 1;
            areg = (int)base_sel((ia >= 0 && ia < m) ? a[ia] : 0xFF);
            breg = (int)base_row((jb >= 0 && jb < n) ? b[jb] : 0xFE);
        }
        const int m31 = m - 31, n1 = n + 1;
        const int neg32 = to_sgpr(-32);
        const int c26 = to_sgpr(0x1a1a1a1a);
        int vm1, a252;                                 // DPP forms take no constant operand; ds_bpermute address of lane 63
        asm volatile("v_mov_b32 %0, -1" : "=v"(vm1));
        asm volatile("v_mov_b32 %0, 0xfc" : "=v"(a252));
        // per-wave LDS windows of the base words that can enter the band during one chunk (the vector unit is the
        // bottleneck of this kernel: 4 cycles per instruction, ~100 % busy; LDS instructions issue on their own port):
        //   A[x]  = sel(a'[t0 - 1 + x])         lane k reads A[k + downs]
        //   B'[z] = row(b'[e0 + 63 - z])        lane k reads B'[64 + k - rights],  e0 = s_lo - t0 - 1   (B' is stored mirrored
        //           so that both reads are "R + constant" with ONE address register R = &A[k + downs] in the unrolled form:
        //           &B'[64 + k - rights] = R + 512 + 4 (64 - j) at step j of the chunk)
        unsigned *winA = &s_bases[threadIdx.x >> 6][0][0], *winB = &s_bases[threadIdx.x >> 6][1][0];
        const unsigned ldsA = (unsigned)(uintptr_t)winA, ldsB = (unsigned)(uintptr_t)winB;
        // One anti-diagonal, hand-scheduled: 8 vector instructions.
        //   steering : prev[63] comes to lane 0 through ds_bpermute (LDS port, issued as soon as prev exists), one v_cmp,
        //              s_bitcmp on bit 0 of vcc: move = prev[63] >= prev[0] (odd s) / > (even s)
        //              general step: tn = max(min(t + move, min(m,s) - 31), max(0,s-n) - 32) on the scalar unit;
        //              clamp-free step (see the chunk test below): tn = t + move
        //   down     : the centre word of every lane is re-read one entry further in A; left = shl(prev) - 1 becomes the
        //              next step's diagonal operand (the two pp registers swap roles every step: no copy)
        //   right    : same with the row words / B'; diagonal = shr(pp) folded into the add
#define ARM_CORE_DOWN(PO, PN, MOVE)                                                                     \
            "v_add_u32_dpp %[" PN "], %[prev], %[vm1] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
            MOVE                                                                                  \
            "s_waitcnt lgkmcnt(0)\n\t"                                                            \
            "v_perm_b32 %[tsc], %[c26], %[breg], %[areg]\n\t"                                     \
            "v_add_u32 %[tcd], %[" PO "], %[tsc]\n\t"                                             \
            "v_max3_i32 %[tv], %[tcd], %[prev], %[" PN "]\n\t"                                    \
            "v_and_or_b32 %[prev], %[tv], -4, 1\n\t"                                              \
            "ds_bpermute_b32 %[rot], %[a252], %[prev]\n\t"                                        \
            "v_alignbit_b32 %[d2], %[tv], %[d2], 2\n\t"
#define ARM_CORE_RIGHT(PO, PN, MOVE)                                                                    \
            "v_mov_b32_dpp %[thx], %[prev] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
            "v_add_u32 %[" PN "], -1, %[prev]\n\t"                                                \
            MOVE                                                                                  \
            "s_waitcnt lgkmcnt(0)\n\t"                                                            \
            "v_perm_b32 %[tsc], %[c26], %[breg], %[areg]\n\t"                                     \
            "v_add_u32_dpp %[tcd], %[" PO "], %[tsc] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
            "v_max3_i32 %[tv], %[tcd], %[thx], %[" PN "]\n\t"                                     \
            "v_and_or_b32 %[prev], %[tv], -4, 1\n\t"                                              \
            "ds_bpermute_b32 %[rot], %[a252], %[prev]\n\t"                                        \
            "v_alignbit_b32 %[d2], %[tv], %[d2], 2\n\t"
#define STEP_GEN(CMP, L, PO, PN)                                                                  \
            "s_add_i32 %[s31], %[s31], 1\n\t"                                                     \
            "s_min_i32 %[x], %[m31], %[s31]\n\t"                                                  \
            "s_sub_i32 %[y], %[s31], %[n1]\n\t"                                                   \
            "s_max_i32 %[y], %[y], %[neg32]\n\t"                                                  \
            "s_waitcnt lgkmcnt(0)\n\t"                                                            \
            CMP " vcc, %[rot], %[prev]\n\t"                                                       \
            "s_bitcmp1_b32 vcc_lo, 0\n\t"                                                         \
            "s_addc_u32 %[tn], %[t], 0\n\t"                                                       \
            "s_min_i32 %[tn], %[tn], %[x]\n\t"                                                    \
            "s_max_i32 %[tn], %[tn], %[y]\n\t"                                                    \
            "s_cmp_lg_u32 %[tn], %[t]\n\t"                                                        \
            "s_mov_b32 %[t], %[tn]\n\t"                                                           \
            "s_cbranch_scc0 R" L "_%=\n\t"                                                        \
            "v_add_u32 %[aaddr], 4, %[aaddr]\n\t"                                                 \
            "ds_read_b32 %[areg], %[aaddr]\n\t"                                                   \
            ARM_CORE_DOWN(PO, PN, "s_lshl2_add_u32 %[mreg], %[mreg], 1\n\t")                      \
            "s_branch J" L "_%=\n"                                                                \
            "R" L "_%=:\n\t"                                                                      \
            "v_add_u32 %[baddr], -4, %[baddr]\n\t"                                                \
            "ds_read_b32 %[breg], %[baddr]\n\t"                                                   \
            ARM_CORE_RIGHT(PO, PN, "s_lshl_b32 %[mreg], %[mreg], 2\n\t")                          \
            "J" L "_%=:\n\t"
        // unrolled form: step J (1..64) of the chunk is a literal, the row word of a right move sits at R + 768 - 4 J.  Both
        // words that can enter at step J are fetched speculatively together with the steering value (one LDS round trip per
        // step); the arm taken copies its word into the resident register and re-fetches only its own side (after a down
        // move the row word that a right move would bring in is still the same entry of B', and vice versa).  Measured
        // cost model on MI355X: a wave64 vector instruction occupies its SIMD for 4 cycles (1 CU-cycle with the 4 SIMDs in
        // parallel), a wave64 LDS instruction occupies the CU's LDS pipe for ~4 cycles: 9 vector + 2 LDS per step is
        // vector-bound (11.6 vs 8 CU-cycles with the traceback's share); re-reading through LDS instead of the two copies
        // (8 vector + 3 LDS) turns it LDS-bound and is slower.
#define STEP_FAST(CMP, L, J, PO, PN)                                                              \
            "s_waitcnt lgkmcnt(0)\n\t"                                                            \
            CMP " vcc, %[rot], %[prev]\n\t"                                                       \
            "s_bitcmp1_b32 vcc_lo, 0\n\t"                                                         \
            "s_cbranch_scc0 R" L "_%=\n\t"                                                        \
            "v_add_u32_dpp %[" PN "], %[prev], %[vm1] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
            "v_perm_b32 %[tsc], %[c26], %[breg], %[an]\n\t"                                       \
            "v_add_u32 %[tcd], %[" PO "], %[tsc]\n\t"                                             \
            "v_max3_i32 %[tv], %[tcd], %[prev], %[" PN "]\n\t"                                    \
            "v_and_or_b32 %[prev], %[tv], -4, 1\n\t"                                              \
            "ds_bpermute_b32 %[rot], %[a252], %[prev]\n\t"                                        \
            "v_add_u32 %[aaddr], 4, %[aaddr]\n\t"                                                 \
            "v_mov_b32 %[areg], %[an]\n\t"                                                        \
            "s_lshl2_add_u32 %[mreg], %[mreg], 1\n\t"                                             \
            "ds_read_b32 %[an], %[aaddr] offset:4\n\t"                                            \
            "v_alignbit_b32 %[d2], %[tv], %[d2], 2\n\t"                                           \
            "s_branch J" L "_%=\n"                                                                \
            "R" L "_%=:\n\t"                                                                      \
            "v_mov_b32_dpp %[thx], %[prev] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
            "v_add_u32 %[" PN "], -1, %[prev]\n\t"                                                \
            "v_perm_b32 %[tsc], %[c26], %[bn], %[areg]\n\t"                                       \
            "v_add_u32_dpp %[tcd], %[" PO "], %[tsc] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
            "v_max3_i32 %[tv], %[tcd], %[thx], %[" PN "]\n\t"                                     \
            "v_and_or_b32 %[prev], %[tv], -4, 1\n\t"                                              \
            "ds_bpermute_b32 %[rot], %[a252], %[prev]\n\t"                                        \
            "v_mov_b32 %[breg], %[bn]\n\t"                                                        \
            "s_lshl_b32 %[mreg], %[mreg], 2\n\t"                                                  \
            "ds_read_b32 %[bn], %[aaddr] offset:764-4*(" J ")\n\t"                                \
            "v_alignbit_b32 %[d2], %[tv], %[d2], 2\n\t"                                           \
            "J" L "_%=:\n\t"
#define STEP_FAST2(L) STEP_FAST("v_cmp_ge_i32", L "o", "2*" L "+1", "p0", "p1") STEP_FAST("v_cmp_gt_i32", L "e", "2*" L "+2", "p1", "p0")
#define STEP_VREGS                                                                                                         \
              [prev] "+v"(prev), [p0] "+v"(p0), [p1] "+v"(p1), [areg] "+v"(areg), [breg] "+v"(breg), [aaddr] "+v"(aaddr), \
              [d2] "+v"(d2), [tsc] "=&v"(tsc), [tcd] "=&v"(tcd), [thx] "=&v"(thx), [tv] "=&v"(tv), [rot] "=&v"(rot)
        int p0 = pp, p1 = 0;                           // the diagonal operand lives in p0 at the start of every chunk
        // chunk = 64 anti-diagonals = four 16-step direction words per lane.  Words 0..2 are parked in d2a/d2b/d2c (moves in
        // mrega/b/c) as they fill up; the general form walks the four words in a small loop inside the same asm statement.
        const int nchunk64 = (steps + 63) >> 6;
        for (int ch = 0; ch < nchunk64; ch++) {
            const int s_lo = (ch << 6) + 1;
            const int left = steps - (ch << 6);
            const int nst = left < 64 ? left : 64;
            {   // windows of this chunk: the words the lanes hold now, then the (<= 64) that can enter on either side
                winA[lane] = (unsigned)areg;
                winB[64 + lane] = (unsigned)breg;
                const int ia = t + 63 + lane, ib = s_lo - t - 1 + lane;
                const int ra = a[(unsigned)ia < (unsigned)m ? ia : 0], rb = b[(unsigned)ib < (unsigned)n ? ib : 0];
                winA[64 + lane] = base_sel((unsigned)ia < (unsigned)m ? ra : 0xFF);
                winB[63 - lane] = base_row((unsigned)ib < (unsigned)n ? rb : 0xFE);
            }
            int aaddr = (int)(ldsA + 4 * lane);
            int tsc, tcd, thx, tv, rot, sx, sy, tn;
            // neither clamp can bind during a full chunk that starts with t + 64 <= m - 31 and t >= max(0, s_hi - n) - 32
            // (t only grows, by at most one per step; t <= s - 32 always): such chunks run the 64 steps unrolled and
            // recover t from the recorded moves; the other chunks run word by word through the general loop.
            const int s_hi = s_lo + 63;
            // sign-bit form (no booleans: the compiler would route them through a VGPR)
            const int lo_t = (s_hi > n ? s_hi - n : 0) - 32;
            const int fast = (int)(~(unsigned)((m31 - t - 64) | (t - lo_t) | (nst - 64)) >> 31);
            // two bits per step, step r of a 16-step word at bits 2r+1 : 2r (directions), the move at bit 30 - 2r
            if (fast) {
                int d2 = 0, d2a, d2b, d2c;
                int mreg = to_sgpr(0), mrega, mregb, mregc;
                int an, bn;
                asm volatile(
                    "ds_bpermute_b32 %[rot], %[a252], %[prev]\n\t"
                    "ds_read_b32 %[an], %[aaddr] offset:4\n\t"
                    "ds_read_b32 %[bn], %[aaddr] offset:764\n\t"
                    STEP_FAST2("0") STEP_FAST2("1") STEP_FAST2("2") STEP_FAST2("3")
                    STEP_FAST2("4") STEP_FAST2("5") STEP_FAST2("6") STEP_FAST2("7")
                    "v_mov_b32 %[d2a], %[d2]\n\t"
                    "s_mov_b32 %[mrega], %[mreg]\n\t"
                    STEP_FAST2("8") STEP_FAST2("9") STEP_FAST2("10") STEP_FAST2("11")
                    STEP_FAST2("12") STEP_FAST2("13") STEP_FAST2("14") STEP_FAST2("15")
                    "v_mov_b32 %[d2b], %[d2]\n\t"
                    "s_mov_b32 %[mregb], %[mreg]\n\t"
                    STEP_FAST2("16") STEP_FAST2("17") STEP_FAST2("18") STEP_FAST2("19")
                    STEP_FAST2("20") STEP_FAST2("21") STEP_FAST2("22") STEP_FAST2("23")
                    "v_mov_b32 %[d2c], %[d2]\n\t"
                    "s_mov_b32 %[mregc], %[mreg]\n\t"
                    STEP_FAST2("24") STEP_FAST2("25") STEP_FAST2("26") STEP_FAST2("27")
                    STEP_FAST2("28") STEP_FAST2("29") STEP_FAST2("30") STEP_FAST2("31")
                    "s_waitcnt lgkmcnt(0)\n\t"
                    : STEP_VREGS, [d2a] "=&v"(d2a), [d2b] "=&v"(d2b), [d2c] "=&v"(d2c), [mreg] "+s"(mreg), [mrega] "=&s"(mrega),
                      [mregb] "=&s"(mregb), [mregc] "=&s"(mregc), [an] "=&v"(an), [bn] "=&v"(bn)
                    : [vm1] "v"(vm1), [a252] "v"(a252), [c26] "s"(c26)
                    : "vcc", "scc", "memory");
                t += __builtin_popcount((unsigned)mrega & 0x55555555u) + __builtin_popcount((unsigned)mregb & 0x55555555u) +
                     __builtin_popcount((unsigned)mregc & 0x55555555u) + __builtin_popcount((unsigned)mreg & 0x55555555u);
                unsigned *wp = tbd + (4 * ch) * 64 + lane;
                wp[0] = (unsigned)d2a; wp[64] = (unsigned)d2b; wp[128] = (unsigned)d2c; wp[192] = (unsigned)d2;
                if (lane == 0) { tbm[4 * ch] = (unsigned)mrega; tbm[4 * ch + 1] = (unsigned)mregb; tbm[4 * ch + 2] = (unsigned)mregc; tbm[4 * ch + 3] = (unsigned)mreg; }
            } else {
                int s31 = to_sgpr(s_lo - 32);          // (s - 31) of the step before the next one
                int baddr = (int)(ldsB + 4 * (64 + lane));
                for (int w_ = 0; (w_ << 4) < nst; w_++) {
                    const int nsw = nst - (w_ << 4) < 16 ? nst - (w_ << 4) : 16;
                    int d2 = 0;
                    int mreg = to_sgpr(0);
                    int cnt = to_sgpr((nsw >> 1) - 1);         // pairs - 1
                    const int odd = to_sgpr(nsw & 1);          // only the last word of the last chunk can be odd
                    asm volatile(
                        "ds_bpermute_b32 %[rot], %[a252], %[prev]\n\t"
                        "s_cmp_lt_i32 %[cnt], 0\n\t"
                        "s_cbranch_scc1 S_%=\n"
                        "L_%=:\n\t"
                        STEP_GEN("v_cmp_ge_i32", "a", "p0", "p1")
                        STEP_GEN("v_cmp_gt_i32", "b", "p1", "p0")
                        "s_sub_u32 %[cnt], %[cnt], 1\n\t"
                        "s_cbranch_scc0 L_%=\n"
                        "S_%=:\n\t"
                        "s_cmp_eq_u32 %[odd], 0\n\t"
                        "s_cbranch_scc1 E_%=\n\t"
                        STEP_GEN("v_cmp_ge_i32", "c", "p0", "p1")
                        "E_%=:\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        : STEP_VREGS, [baddr] "+v"(baddr), [t] "+s"(t), [mreg] "+s"(mreg), [s31] "+s"(s31), [cnt] "+s"(cnt), [x] "=&s"(sx),
                          [y] "=&s"(sy), [tn] "=&s"(tn)
                        : [m31] "s"(m31), [n1] "s"(n1), [neg32] "s"(neg32), [vm1] "v"(vm1), [a252] "v"(a252), [c26] "s"(c26), [odd] "s"(odd)
                        : "vcc", "scc", "memory");
                    tbd[(4 * ch + w_) * 64 + lane] = (unsigned)d2 >> (2 * (16 - nsw));
                    if (lane == 0) tbm[4 * ch + w_] = (unsigned)mreg << (2 * (16 - nsw));
                }
            }
        }
#undef STEP_GEN
#undef STEP_FAST
#undef STEP_FAST2
#undef ARM_CORE_DOWN
#undef ARM_CORE_RIGHT
#undef STEP_VREGS
        {
            const int kf = m - t;
            const int hf = (kf >= 0 && kf < 64) ? __builtin_amdgcn_readlane(prev, kf & 63) : 0;
            if (hf <= (MBIAS / 2) * 4) { if (lane == 0) atomicExch(&P.status[c], 1); continue; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        // ---------------- traceback: scalar walk ----------------
        // i-1 (in M0: it is the lane select of the v_writelane that drops each result into `oreg`), j-1, k and the bit index
        // q of the anti-diagonal inside its 16-step word live in SGPRs; the direction word of the current cell is one
        // v_readlane of the word's per-lane values.  A diagonal step costs 2 vector + 8 scalar instructions.  One asm loop
        // per 16-step word, then the <= 16 centre positions it produced leave as one masked store.  A cell on the path is
        // always inside the band (its score derives from H(0,0), checked above; band-edge fills are 2^28 lower), so k is
        // only checked per word.
#define TB_LOOP(V, WCUR, MMA, MMB, EXIT4, EXIT2U, EXIT2L)                                          \
                "s_mov_b32 m0, %[ip]\n\t"                                                         \
                "s_branch L" V "_%=\n"                                                            \
                "N" V "_%=:\n\t"                                                                  \
                "s_lshl_b32 %[w], %[w], 1\n\t"                                                    \
                "s_bitcmp1_b32 %[w], %[q]\n\t"                                                    \
                "s_cbranch_scc0 F" V "_%=\n\t"                                                    \
                /* up: centre position i-1 faces a gap before row position j;  k += move(s) - 1 */ \
                "s_add_i32 %[x], %[jp], 0x8001\n\t"                                               \
                "v_writelane_b32 %[oreg], %[x], m0\n\t"                                           \
                "s_bitcmp0_b32 %[" MMA "], %[q]\n\t"                                              \
                "s_subb_u32 %[k], %[k], 0\n\t"                                                    \
                "s_sub_i32 m0, m0, 1\n\t"                                                         \
                EXIT2U                                                                            \
                "s_branch E" V "_%=\n"                                                            \
                "F" V "_%=:\n\t"                                                                  \
                /* left: a row base inserted;  k += move(s) */                                    \
                "s_bitcmp1_b32 %[" MMA "], %[q]\n\t"                                              \
                "s_addc_u32 %[k], %[k], 0\n\t"                                                    \
                "s_sub_i32 %[jp], %[jp], 1\n\t"                                                   \
                EXIT2L                                                                            \
                "s_branch E" V "_%=\n"                                                            \
                "L" V "_%=:\n\t"                                                                  \
                "v_readlane_b32 %[w], %[" WCUR "], %[k]\n\t"                                      \
                "s_bitcmp1_b32 %[w], %[q]\n\t"                                                    \
                "s_cbranch_scc0 N" V "_%=\n\t"                                                    \
                /* diagonal: centre position i-1 <-> row position j-1;  k += move(s) + move(s-1) - 1 */ \
                "v_writelane_b32 %[oreg], %[jp], m0\n\t"                                          \
                "s_bitcmp0_b32 %[" MMA "], %[q]\n\t"                                              \
                "s_subb_u32 %[k], %[k], 0\n\t"                                                    \
                "s_bitcmp1_b32 %[" MMB "], %[q]\n\t"                                              \
                "s_addc_u32 %[k], %[k], 0\n\t"                                                    \
                "s_sub_i32 m0, m0, 1\n\t"                                                         \
                "s_sub_i32 %[jp], %[jp], 1\n\t"                                                   \
                EXIT4                                                                             \
                "E" V "_%=:\n\t"                                                                  \
                "s_mov_b32 %[ip], m0\n\t"
        // exit tests.  plain: the two words cannot reach row / column 0 (i-1, j-1 >= 32 on entry): only q can run out (q is
        // odd: the subtraction borrows exactly when the word is finished).  careful: also i-1 < 0 or j-1 < 0.
#define TB_PLAIN(N, V) "s_sub_u32 %[q], %[q], " N "\n\ts_cbranch_scc0 L" V "_%=\n\t"
#define TB_CAREFUL(N, V)                                                                          \
                "s_sub_i32 %[q], %[q], " N "\n\t"                                                 \
                "s_or_b32 %[x], %[jp], m0\n\t"                                                    \
                "s_or_b32 %[x], %[x], %[q]\n\t"                                                   \
                "s_cmp_lt_i32 %[x], 0\n\t"                                                        \
                "s_cbranch_scc0 L" V "_%=\n\t"
        int ip = m - 1, jp = n - 1;
        int k = m - t;                    // lane that owns cell (i, j) on anti-diagonal i + j
        int oreg = 0, bad = 0, fail;
        // between two words of a trip: careful form -- stop if row / column 0 was reached or no word is left
#define TB_NEXT_CAREFUL(HAVE)                                                                     \
                "s_or_b32 %[x], %[jp], %[ip]\n\t"                                                 \
                "s_cmp_lt_i32 %[x], 0\n\t"                                                        \
                "s_cbranch_scc1 X_%=\n\t"                                                         \
                "s_cmp_lt_u32 %[nw], " HAVE "\n\t"                                                \
                "s_cbranch_scc1 X_%=\n\t"                                                         \
                "s_add_i32 %[q], %[q], 32\n\t"
        while (ip >= 0 && jp >= 0) {
            const int sp = ip + jp + 1;   // word / bit index of anti-diagonal i + j
            const int dch = sp >> 4;
            int q = to_sgpr(((sp & 15) << 1) + 1);        // 2 x (step inside the word) + 1
            // up to four 16-step words per trip (the one that holds anti-diagonal i + j and the three below): step r of a
            // word has bit 2r+1 "diagonal wins", else bit 2r "up", else left
            const int nw = dch < 3 ? dch + 1 : 4;         // words available from dch downwards (integer form)
            const unsigned *wp = tbd + lane;
            const unsigned wcur3 = wp[dch * 64], wcur2 = wp[(dch > 0 ? dch - 1 : 0) * 64], wcur1 = wp[(dch > 1 ? dch - 2 : 0) * 64],
                           wcur0 = wp[(dch > 2 ? dch - 3 : 0) * 64];
            const unsigned mv4 = tbm[dch], mv3 = dch > 0 ? tbm[dch - 1] : 0u, mv2 = dch > 1 ? tbm[dch - 2] : 0u,
                           mv1 = dch > 2 ? tbm[dch - 3] : 0u, mv0 = dch > 3 ? tbm[dch - 4] : 0u;
            // per word: mma bit 2r+1 = move of step r, mmb bit 2r+1 = move of step r-1 (bit 1: last move of the word below)
            const int mma3 = to_sgpr((int)__brev(mv4)), mmb3 = to_sgpr((int)((__brev(mv4) << 2) | ((mv3 & 1u) << 1)));
            const int mma2 = to_sgpr((int)__brev(mv3)), mmb2 = to_sgpr((int)((__brev(mv3) << 2) | ((mv2 & 1u) << 1)));
            const int mma1 = to_sgpr((int)__brev(mv2)), mmb1 = to_sgpr((int)((__brev(mv2) << 2) | ((mv1 & 1u) << 1)));
            const int mma0 = to_sgpr((int)__brev(mv1)), mmb0 = to_sgpr((int)((__brev(mv1) << 2) | ((mv0 & 1u) << 1)));
            bad |= (unsigned)k > 63u;
            const int ip0 = ip;
            const int plain = (ip < jp ? ip : jp) >> 6;   // != 0: both >= 64 (integer form: stays on the scalar unit)
            int sw, sx;
            asm volatile(
                "s_cmp_lg_u32 %[plain], 0\n\t"
                "s_cbranch_scc1 P_%=\n\t"
                TB_LOOP("c3", "wcur3", "mma3", "mmb3", TB_CAREFUL("4", "c3"), TB_CAREFUL("2", "c3"), TB_CAREFUL("2", "c3"))
                TB_NEXT_CAREFUL("2")
                TB_LOOP("c2", "wcur2", "mma2", "mmb2", TB_CAREFUL("4", "c2"), TB_CAREFUL("2", "c2"), TB_CAREFUL("2", "c2"))
                TB_NEXT_CAREFUL("3")
                TB_LOOP("c1", "wcur1", "mma1", "mmb1", TB_CAREFUL("4", "c1"), TB_CAREFUL("2", "c1"), TB_CAREFUL("2", "c1"))
                TB_NEXT_CAREFUL("4")
                TB_LOOP("c0", "wcur0", "mma0", "mmb0", TB_CAREFUL("4", "c0"), TB_CAREFUL("2", "c0"), TB_CAREFUL("2", "c0"))
                "s_branch X_%=\n"
                "P_%=:\n\t"
                TB_LOOP("p3", "wcur3", "mma3", "mmb3", TB_PLAIN("4", "p3"), TB_PLAIN("2", "p3"), TB_PLAIN("2", "p3"))
                "s_add_i32 %[q], %[q], 32\n\t"
                TB_LOOP("p2", "wcur2", "mma2", "mmb2", TB_PLAIN("4", "p2"), TB_PLAIN("2", "p2"), TB_PLAIN("2", "p2"))
                "s_add_i32 %[q], %[q], 32\n\t"
                TB_LOOP("p1", "wcur1", "mma1", "mmb1", TB_PLAIN("4", "p1"), TB_PLAIN("2", "p1"), TB_PLAIN("2", "p1"))
                "s_add_i32 %[q], %[q], 32\n\t"
                TB_LOOP("p0", "wcur0", "mma0", "mmb0", TB_PLAIN("4", "p0"), TB_PLAIN("2", "p0"), TB_PLAIN("2", "p0"))
                "\nX_%=:\n\t"
                : [ip] "+s"(ip), [jp] "+s"(jp), [k] "+s"(k), [q] "+s"(q), [oreg] "+v"(oreg), [w] "=&s"(sw), [x] "=&s"(sx)
                : [wcur3] "v"(wcur3), [wcur2] "v"(wcur2), [wcur1] "v"(wcur1), [wcur0] "v"(wcur0), [mma3] "s"(mma3), [mmb3] "s"(mmb3),
                  [mma2] "s"(mma2), [mmb2] "s"(mmb2), [mma1] "s"(mma1), [mmb1] "s"(mmb1), [mma0] "s"(mma0), [mmb0] "s"(mmb0),
                  [plain] "s"(plain), [nw] "s"(nw)
                : "scc");
            // positions (ip, ip0] were produced by this trip (<= 64): lane l holds the one with p mod 64 == l
            const int pl = ip0 - ((ip0 - lane) & 63);
            if (pl > ip) ops[pl] = (uint16_t)oreg;
        }
        for (int q = lane; q <= ip; q += 64) ops[q] = (uint16_t)0x8000;   // j == 0: gaps before row position 0
#undef TB_NEXT_CAREFUL
#undef TB_LOOP
#undef TB_PLAIN
#undef TB_CAREFUL
        fail = bad;
        if (fail && lane == 0) atomicExch(&P.status[c], 1);
    }
}

// insertions of row r before centre position p (p = 0..m), from two neighbouring ops entries
__device__ __forceinline__ int row_ins(const uint16_t *__restrict__ rop, int p, int m, int nrow) {
    int prev_end = 0;
    if (p > 0) { unsigned o = rop[p - 1]; prev_end = (int)(o & 0x7fff) + ((o >> 15) ? 0 : 1); }
    int q = p < m ? (int)(rop[p] & 0x7fff) : nrow;
    return q - prev_end;
}

// column layout: one block per candidate.  insmax -> row-0 slot of ops, block starts -> slot R.
__global__ void __launch_bounds__(256) star_layout_kernel(MsaParams P) {
    __shared__ int s_scan[8];
    const int c = blockIdx.x;
    if (c >= P.n) return;
    const int64_t g0 = P.row_first[c];
    const int R = P.row_first[c + 1] - P.row_first[c];
    if (R <= 0) { if (threadIdx.x == 0) P.cols_out[c] = 0; return; }
    const int m = P.win_len[g0];
    uint16_t *ops = P.ops + P.ops_base[c];
    uint16_t *insmax = ops;                              // centre row slot (the centre has no ops of its own)
    uint16_t *bstart = ops + (int64_t)R * (m + 1);       // extra slot
    if (P.status[c]) { if (threadIdx.x == 0) P.cols_out[c] = 0; return; }
    int running = 0;
    for (int base = 0; base <= m; base += 256) {
        int p = base + threadIdx.x;
        int mx = 0;
        if (p <= m) for (int r = 1; r < R; r++) { int v = row_ins(ops + (int64_t)r * (m + 1), p, m, P.win_len[g0 + r]); mx = v > mx ? v : mx; }
        int width = p <= m ? mx + (p < m ? 1 : 0) : 0;
        int tot;
        int pre = block_excl_scan(width, s_scan, &tot);
        if (p <= m) {
            int bs = running + pre;
            insmax[p] = (uint16_t)(mx > 65535 ? 65535 : mx);
            bstart[p] = (uint16_t)(bs > 65535 ? 65535 : bs);
        }
        running += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (running > 65535) { P.status[c] = 1; P.cols_out[c] = 0; }
        else P.cols_out[c] = running;
    }
}

struct FillParams {
    int n;
    const uint8_t *win;
    const int64_t *win_off;
    const int32_t *win_len;
    const int32_t *row_first;
    const int64_t *ops_base;
    const uint16_t *ops;
    const int32_t *cols;
    const int64_t *msa_off;
    uint8_t *msa;
};

// fill: grid (candidate, row slice); item (r, p) owns insertion block p + centre column p of row r,
// so every output byte is written exactly once and no scan is needed.
__global__ void __launch_bounds__(256) star_fill_kernel(FillParams P) {
    const int c = blockIdx.x;
    if (c >= P.n) return;
    const int C = P.cols[c];
    if (C <= 0) return;
    const int64_t g0 = P.row_first[c];
    const int R = P.row_first[c + 1] - P.row_first[c];
    const int m = P.win_len[g0];
    const uint16_t *ops = P.ops + P.ops_base[c];
    const uint16_t *insmax = ops;
    const uint16_t *bstart = ops + (int64_t)R * (m + 1);
    uint8_t *out = P.msa + P.msa_off[c];
    for (int r = blockIdx.y; r < R; r += gridDim.y) {
        const uint8_t *b = P.win + P.win_off[g0 + r];
        const int nrow = P.win_len[g0 + r];
        uint8_t *row = out + (int64_t)r * C;
        const uint16_t *rop = ops + (int64_t)r * (m + 1);
        for (int p = threadIdx.x; p <= m; p += 256) {
            int ins, gap = 0, q;
            if (r == 0) { ins = 0; q = p; }
            else {
                ins = row_ins(rop, p, m, nrow);
                if (p < m) { unsigned o = rop[p]; q = (int)(o & 0x7fff); gap = (int)(o >> 15); } else q = nrow;
            }
            const int bs = bstart[p], im = insmax[p];
            const int rp = q - ins;  // first inserted base
            for (int k = 0; k < im; k++) row[bs + k] = k < ins ? b[rp + k] : (uint8_t)'-';
            if (p < m) row[bs + im] = gap ? (uint8_t)'-' : b[q];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fused column layout + remove_sparse_col_in_align_file (Util.py:10344-10405) for the fine-stage pipeline: the full
// alignment is never materialised.  Column c of the full alignment is kept iff c == 0 or c == C-1 or gaps(c) <= R/2:
//   * centre column p: gaps = rows whose op carries the gap flag;
//   * column j of the insertion block before p: a base in every row with ins_r > j, a gap elsewhere (centre included),
//     so the kept columns of a block are a PREFIX of length kw = the ceil(R/2)-th largest ins_r;
//   * the first / last column of the alignment are kept regardless (the last one may be a non-prefix column of block m).
// Per position p: ops row-0 slot = kw | keep_centre << 15, extra slot = first kept column of the block.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) star_layout_sparse_kernel(MsaParams P, int32_t *__restrict__ new_cols,
                                                                 int32_t *__restrict__ last_extra) {
    __shared__ int s_scan[8];
    __shared__ int s_mxm;
    const int c = blockIdx.x;
    if (c >= P.n) return;
    const int64_t g0 = P.row_first[c];
    const int R = P.row_first[c + 1] - P.row_first[c];
    if (R <= 0 || P.status[c]) { if (threadIdx.x == 0) { P.cols_out[c] = 0; new_cols[c] = 0; last_extra[c] = -1; } return; }
    const int m = P.win_len[g0];
    uint16_t *ops = P.ops + P.ops_base[c];
    uint16_t *kwslot = ops;                              // centre row slot
    uint16_t *nstart = ops + (int64_t)R * (m + 1);       // extra slot
    const int h = (R + 1) >> 1;                          // fewest rows with a base for a column to survive
    __shared__ int s_wl[MSA_MAXR];
    for (int r = threadIdx.x; r < R && r < MSA_MAXR; r += 256) s_wl[r] = P.win_len[g0 + r];
    if (threadIdx.x == 0) s_mxm = 0;
    __syncthreads();
    const int64_t rs = m + 1;                            // ops row stride
    {   // widest insertion after the last centre position: decides which column is the last one
        int mxm = 0;
        for (int r = 1 + threadIdx.x; r < R; r += 256) { int v = row_ins(ops + r * rs, m, m, P.win_len[g0 + r]); mxm = v > mxm ? v : mxm; }
        if (mxm > 0) atomicMax(&s_mxm, mxm);
    }
    __syncthreads();
    const int mxm = s_mxm;
    int run_new = 0, run_full = 0, extra = -1;
    for (int base = 0; base <= m; base += 256) {
        const int p = base + threadIdx.x;
        int mx = 0, npos = 0, gapc = 0;
        if (p <= m) {
            // ins_r(p) = q_r(p) - end_r(p-1): two u16 per row, four rows in flight
            const uint16_t *col = ops + p;
            int r = 1;
            for (; r + 7 < R; r += 8) {
                unsigned oc[8], op[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { oc[u] = p < m ? col[(r + u) * rs] : 0u; op[u] = p > 0 ? col[(r + u) * rs - 1] : 0u; }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int pe = p > 0 ? (int)(op[u] & 0x7fff) + ((op[u] >> 15) ? 0 : 1) : 0;
                    const int q = p < m ? (int)(oc[u] & 0x7fff) : (r + u < MSA_MAXR ? s_wl[r + u] : P.win_len[g0 + r + u]);
                    const int v = q - pe;
                    mx = v > mx ? v : mx; npos += v > 0; gapc += (int)(oc[u] >> 15);
                }
            }
            for (; r + 3 < R; r += 4) {
                unsigned oc[4], op[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { oc[u] = p < m ? col[(r + u) * rs] : 0u; op[u] = p > 0 ? col[(r + u) * rs - 1] : 0u; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int pe = p > 0 ? (int)(op[u] & 0x7fff) + ((op[u] >> 15) ? 0 : 1) : 0;
                    const int q = p < m ? (int)(oc[u] & 0x7fff) : (r + u < MSA_MAXR ? s_wl[r + u] : P.win_len[g0 + r + u]);
                    const int v = q - pe;
                    mx = v > mx ? v : mx; npos += v > 0; gapc += (int)(oc[u] >> 15);
                }
            }
            for (; r < R; r++) {
                const uint16_t *rop = ops + r * rs;
                const int v = row_ins(rop, p, m, r < MSA_MAXR ? s_wl[r] : P.win_len[g0 + r]);
                mx = v > mx ? v : mx;
                npos += v > 0;
                if (p < m) gapc += rop[p] >> 15;
            }
        }
        int kw = 0;
        if (npos >= h) {
            kw = 1;
            for (;;) {
                int cnt = 0;
                for (int r = 1; r < R; r++) cnt += row_ins(ops + (int64_t)r * (m + 1), p, m, P.win_len[g0 + r]) > kw;
                if (cnt >= h) kw++; else break;
            }
        }
        int keepc = (p < m && 2 * gapc <= R) ? 1 : 0;
        int ex = 0;
        if (p == 0) { if (mx > 0) kw = kw > 1 ? kw : 1; else if (m > 0) keepc = 1; }     // first column
        if (p == m - 1 && mxm == 0) keepc = 1;                                            // last column = centre column m-1
        if (p == m && mx > 0 && kw < mx) { ex = 1; extra = mx - 1; }                       // last column = block m, j = mx-1
        if (kw > 0x7fff) kw = 0x7fff;
        const int wnew = p <= m ? kw + keepc + ex : 0;
        const int wfull = p <= m ? mx + (p < m ? 1 : 0) : 0;
        int tot_new, tot_full;
        const int pre = block_excl_scan(wnew, s_scan, &tot_new);
        __syncthreads();
        (void)block_excl_scan(wfull, s_scan, &tot_full);
        if (p <= m) {
            const int bs = run_new + pre;
            kwslot[p] = (uint16_t)(kw | (keepc << 15));
            nstart[p] = (uint16_t)(bs > 65535 ? 65535 : bs);
        }
        run_new += tot_new; run_full += tot_full;
        __syncthreads();
    }
    if (threadIdx.x == (m & 255)) last_extra[c] = extra;   // the thread that owned p == m
    if (threadIdx.x == 0) {
        if (run_full > 65535) { P.status[c] = 1; P.cols_out[c] = 0; new_cols[c] = 0; }
        else { P.cols_out[c] = run_full; new_cols[c] = run_new; }
    }
}

struct FillSparseParams {
    FillParams F;              // cols = kept columns per candidate, msa = compacted output
    const int32_t *last_extra;
};

// fill of the kept columns only: item (r, p) owns the kept prefix of insertion block p, the centre column p if kept
// and (p == m) the extra last column; every output byte is written exactly once.
__global__ void __launch_bounds__(256) star_fill_sparse_kernel(FillSparseParams Q) {
    const FillParams &P = Q.F;
    const int c = blockIdx.x;
    if (c >= P.n) return;
    const int C = P.cols[c];
    if (C <= 0) return;
    const int64_t g0 = P.row_first[c];
    const int R = P.row_first[c + 1] - P.row_first[c];
    const int m = P.win_len[g0];
    const uint16_t *ops = P.ops + P.ops_base[c];
    const uint16_t *kwslot = ops;
    const uint16_t *nstart = ops + (int64_t)R * (m + 1);
    const int le = Q.last_extra[c];
    uint8_t *out = P.msa + P.msa_off[c];
    // rows of the candidate are spread over blockIdx.y, positions over the threads: FILL_U positions per thread and trip, the
    // loads of each dependency level issued together (the chain kwslot -> ops -> base is three loads deep and the kernel is
    // latency bound otherwise); per-row values (window, length, output row) are wave-uniform: no division, little address math
    const int rs = m + 1;
    for (int r = blockIdx.y; r < R; r += gridDim.y) {
        const uint8_t *b = P.win + P.win_off[g0 + r];
        const int nrow = P.win_len[g0 + r];
        const uint16_t *rop = ops + (int64_t)r * rs;
        uint8_t *row = out + (int64_t)r * C;
        for (int p0 = threadIdx.x; p0 <= m; p0 += FILL_U * 256) {
            int p[FILL_U], kw[FILL_U], kc[FILL_U], ins[FILL_U], gap[FILL_U], q[FILL_U], bs[FILL_U];
            bool live[FILL_U], ex[FILL_U];
            unsigned ks[FILL_U], oc[FILL_U], op[FILL_U];
#pragma unroll
            for (int u = 0; u < FILL_U; u++) {
                p[u] = p0 + u * 256;
                live[u] = p[u] <= m;
                if (!live[u]) p[u] = 0;
                ks[u] = kwslot[p[u]];
                oc[u] = (r > 0 && p[u] < m) ? rop[p[u]] : 0u;
                op[u] = (r > 0 && p[u] > 0) ? rop[p[u] - 1] : 0u;
                bs[u] = nstart[p[u]];
            }
#pragma unroll
            for (int u = 0; u < FILL_U; u++) {
                kw[u] = (int)(ks[u] & 0x7fff); kc[u] = (int)(ks[u] >> 15);
                ex[u] = p[u] == m && le >= 0;
                live[u] = live[u] && (kw[u] != 0 || kc[u] || ex[u]);
                if (r == 0) { ins[u] = 0; q[u] = p[u]; gap[u] = 0; }
                else {
                    const int pe = p[u] > 0 ? (int)(op[u] & 0x7fff) + ((op[u] >> 15) ? 0 : 1) : 0;
                    q[u] = p[u] < m ? (int)(oc[u] & 0x7fff) : nrow;
                    gap[u] = (int)(oc[u] >> 15);
                    ins[u] = q[u] - pe;
                }
            }
            uint8_t cb[FILL_U];
#pragma unroll
            for (int u = 0; u < FILL_U; u++) cb[u] = (live[u] && kc[u] && !gap[u]) ? b[q[u]] : (uint8_t)'-';
#pragma unroll
            for (int u = 0; u < FILL_U; u++) {
                if (!live[u]) continue;
                const int rp = q[u] - ins[u];  // first inserted base
                for (int k = 0; k < kw[u]; k++) row[bs[u] + k] = k < ins[u] ? b[rp + k] : (uint8_t)'-';
                if (kc[u]) row[bs[u] + kw[u]] = cb[u];
                if (ex[u]) row[bs[u] + kw[u]] = le < ins[u] ? b[rp + le] : (uint8_t)'-';
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// resident workgroups of the alignment kernel.  256 CUs x 8 = 2048 fills every wave slot; a caller that drives several
// contexts on one GPU leaves room for the other context's memory-bound kernels with fewer (HITE_ALIGN_BLOCKS).
static int align_max_blocks() {
    static int v = 0;
    if (!v) {
        const char *e = getenv("HITE_ALIGN_BLOCKS");
        v = e ? atoi(e) : 2048;
        if (v < 64 || v > 4096) v = 2048;
    }
    return v;
}
static int star_msa_launch(hite_ctx *ctx, int32_t n, const uint8_t *d_win, const int64_t *d_win_off,
                           const int32_t *d_win_len, const int32_t *d_row_first, int64_t total_rows, const int64_t *d_ops_base,
                           int64_t ops_elems, int32_t max_win_len, int32_t *d_cols_out, int32_t *d_status,
                           int32_t *d_new_cols, int32_t *d_last_extra, void *stream) {
    if (!ctx || n < 0 || total_rows < 0 || max_win_len <= 0) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    const int max_steps = 2 * max_win_len;
    size_t tb_slot = (size_t)((max_steps >> 4) + 2) * 256 + (size_t)((max_steps >> 4) + 4) * 4;
    tb_slot = (tb_slot + 63) & ~(size_t)63;
    int64_t pairs = total_rows - n;
    int grid = (int)((pairs + 3) / 4);
    if (grid < 1) grid = 1;
    if (grid > align_max_blocks()) grid = align_max_blocks();
    while (grid > 64 && (size_t)grid * 4 * tb_slot > ((size_t)3 << 30)) grid /= 2;
    void *scr = nullptr, *opsb = nullptr;
    int rc = hite_scratch_reserve(ctx, (size_t)grid * 4 * tb_slot + 256, &scr);
    if (rc) return rc;
    rc = hite_scratch2_reserve(ctx, (size_t)ops_elems * 2 + 256, &opsb);
    if (rc) return rc;
    unsigned int *counter = (unsigned int *)((uint8_t *)scr + (size_t)grid * 4 * tb_slot);
    hipStream_t st = (hipStream_t)stream;
    HITE_CHECK(ctx, hipMemsetAsync(counter, 0, 4, st));
    HITE_CHECK(ctx, hipMemsetAsync(d_status, 0, (size_t)n * 4, st));
    MsaParams P;
    P.n = n; P.total_rows = total_rows; P.win = d_win; P.win_off = d_win_off; P.win_len = d_win_len; P.row_first = d_row_first;
    P.ops_base = d_ops_base; P.ops = (uint16_t *)opsb; P.cols_out = d_cols_out; P.status = d_status;
    P.tb = (uint8_t *)scr; P.tb_slot = tb_slot; P.max_steps = max_steps; P.counter = counter;
    int tk = hite_prof_begin(ctx, max_win_len > 1000 ? "star_align_kernel_long" : "star_align_kernel_short", st);
    hipLaunchKernelGGL(star_align_kernel, dim3(grid), dim3(256), 0, st, P);
    hite_prof_end(ctx, tk, st);
    HITE_CHECK(ctx, hipGetLastError());
    if (d_new_cols) {
        tk = hite_prof_begin(ctx, max_win_len > 1000 ? "star_layout_sparse_kernel_long" : "star_layout_sparse_kernel_short", st);
        hipLaunchKernelGGL(star_layout_sparse_kernel, dim3(n), dim3(256), 0, st, P, d_new_cols, d_last_extra);
    } else {
        tk = hite_prof_begin(ctx, "star_layout_kernel", st);
        hipLaunchKernelGGL(star_layout_kernel, dim3(n), dim3(256), 0, st, P);
    }
    hite_prof_end(ctx, tk, st);
    HITE_CHECK(ctx, hipGetLastError());
    return HITE_OK;
}

extern "C" int hite_star_msa_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_win, const int64_t *d_win_off,
                                 const int32_t *d_win_len, const int32_t *d_row_first, int64_t total_rows, const int64_t *d_ops_base,
                                 int64_t ops_elems, int32_t max_win_len, int32_t *d_cols_out, int32_t *d_status,
                                 void *stream) {
    return star_msa_launch(ctx, n, d_win, d_win_off, d_win_len, d_row_first, total_rows, d_ops_base, ops_elems, max_win_len,
                           d_cols_out, d_status, nullptr, nullptr, stream);
}

// align + fused layout / sparse-column selection: d_cols_out = columns of the full alignment (0 = failed),
// d_new_cols = columns that survive remove_sparse_col_in_align_file, d_last_extra = per-candidate fill hint
extern "C" int hite_star_msa_sparse_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_win, const int64_t *d_win_off,
                                        const int32_t *d_win_len, const int32_t *d_row_first, int64_t total_rows,
                                        const int64_t *d_ops_base, int64_t ops_elems, int32_t max_win_len, int32_t *d_cols_out,
                                        int32_t *d_status, int32_t *d_new_cols, int32_t *d_last_extra, void *stream) {
    if (!d_new_cols || !d_last_extra) return HITE_EINVAL;
    return star_msa_launch(ctx, n, d_win, d_win_off, d_win_len, d_row_first, total_rows, d_ops_base, ops_elems, max_win_len,
                           d_cols_out, d_status, d_new_cols, d_last_extra, stream);
}

// compacted alignments (rows x d_new_cols[i] at d_msa_off[i]) from the ops of the last hite_star_msa_sparse_dev call
extern "C" int hite_star_msa_fill_sparse_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_win, const int64_t *d_win_off,
                                             const int32_t *d_win_len, const int32_t *d_row_first, const int64_t *d_ops_base,
                                             const int32_t *d_new_cols, const int32_t *d_last_extra, const int64_t *d_msa_off,
                                             uint8_t *d_msa, void *stream) {
    if (!ctx || n < 0 || !ctx->d_scratch2 || !d_new_cols || !d_last_extra) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    FillSparseParams Q;
    FillParams &P = Q.F;
    P.n = n; P.win = d_win; P.win_off = d_win_off; P.win_len = d_win_len; P.row_first = d_row_first; P.ops_base = d_ops_base;
    P.ops = (const uint16_t *)ctx->d_scratch2; P.cols = d_new_cols; P.msa_off = d_msa_off; P.msa = d_msa;
    Q.last_extra = d_last_extra;
    hipLaunchKernelGGL(star_fill_sparse_kernel, dim3(n, 4), dim3(256), 0, (hipStream_t)stream, Q);
    HITE_CHECK(ctx, hipGetLastError());
    return HITE_OK;
}

extern "C" int hite_star_msa_fill_dev(hite_ctx *ctx, int32_t n, const uint8_t *d_win, const int64_t *d_win_off,
                                      const int32_t *d_win_len, const int32_t *d_row_first, const int64_t *d_ops_base, const int32_t *d_cols,
                                      const int64_t *d_msa_off, uint8_t *d_msa, void *stream) {
    if (!ctx || n < 0 || !ctx->d_scratch2) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    FillParams P;
    P.n = n; P.win = d_win; P.win_off = d_win_off; P.win_len = d_win_len; P.row_first = d_row_first; P.ops_base = d_ops_base;
    P.ops = (const uint16_t *)ctx->d_scratch2; P.cols = d_cols; P.msa_off = d_msa_off; P.msa = d_msa;
    hipLaunchKernelGGL(star_fill_kernel, dim3(n, 4), dim3(256), 0, (hipStream_t)stream, P);
    HITE_CHECK(ctx, hipGetLastError());
    return HITE_OK;
}

struct MBuf {
    void *p = nullptr;
    ~MBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
    hipError_t up(const void *h, size_t n) {
        hipError_t e = alloc(n);
        if (e != hipSuccess) return e;
        return n ? hipMemcpy(p, h, n, hipMemcpyHostToDevice) : hipSuccess;
    }
};

// host-buffer convenience: pass 1 (msa_out == NULL) returns cols_out only; otherwise also the
// alignments at msa_off_out[i] (16-byte aligned slots), msa_cap bytes available.
static int star_msa_host(hite_ctx *ctx, int32_t n, const uint8_t *win, const int64_t *win_off,
                         const int32_t *row_first, int32_t *cols_out, int64_t msa_cap, uint8_t *msa_out,
                         int64_t *msa_off_out, bool sparse) {
    if (!ctx || n < 0 || !win || !win_off || !row_first || !cols_out) return HITE_EINVAL;
    if (n == 0) return HITE_OK;
    HITE_CHECK(ctx, hipSetDevice(ctx->device));
    int64_t total_rows = row_first[n];
    int64_t *ops_base = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    if (!ops_base) return HITE_ENOMEM;
    int64_t acc = 0;
    int maxlen = 1;
    for (int c = 0; c < n; c++) {
        int R = row_first[c + 1] - row_first[c];
        if (R <= 0) { free(ops_base); return HITE_EINVAL; }
        int64_t m = win_off[row_first[c] + 1] - win_off[row_first[c]];
        ops_base[c] = acc;
        acc += (int64_t)(R + 1) * (m + 1);
        for (int r = 0; r < R; r++) {
            int64_t L = win_off[row_first[c] + r + 1] - win_off[row_first[c] + r];
            if (L <= 0 || L > 32767) { free(ops_base); return HITE_EINVAL; }
            if (L > maxlen) maxlen = (int)L;
        }
    }
    ops_base[n] = acc;
    int32_t *wl = (int32_t *)malloc(sizeof(int32_t) * (total_rows + 1));
    if (!wl) { free(ops_base); return HITE_ENOMEM; }
    for (int64_t g = 0; g < total_rows; g++) wl[g] = (int32_t)(win_off[g + 1] - win_off[g]);
    MBuf dw, dwo, dwl, drf, dob, dcols, dst, dmo, dmsa, dnew, dlast;
    hipError_t e;
    e = dw.up(win, win_off[total_rows]); if (e == hipSuccess) e = dwo.up(win_off, (total_rows + 1) * 8);
    if (e == hipSuccess) e = dwl.up(wl, total_rows * 4);
    free(wl);
    if (e == hipSuccess) e = drf.up(row_first, (n + 1) * 4); if (e == hipSuccess) e = dob.up(ops_base, (n + 1) * 8);
    if (e == hipSuccess) e = dcols.alloc(n * 4); if (e == hipSuccess) e = dst.alloc(n * 4);
    if (e == hipSuccess) e = dnew.alloc(n * 4); if (e == hipSuccess) e = dlast.alloc(n * 4);
    free(ops_base);
    HITE_CHECK(ctx, e);
    int rc = star_msa_launch(ctx, n, (uint8_t *)dw.p, (int64_t *)dwo.p, (int32_t *)dwl.p, (int32_t *)drf.p, total_rows, (int64_t *)dob.p, acc,
                             maxlen, (int32_t *)dcols.p, (int32_t *)dst.p, sparse ? (int32_t *)dnew.p : nullptr,
                             sparse ? (int32_t *)dlast.p : nullptr, nullptr);
    if (rc) return rc;
    HITE_CHECK(ctx, hipDeviceSynchronize());
    HITE_CHECK(ctx, hipMemcpy(cols_out, sparse ? dnew.p : dcols.p, n * 4, hipMemcpyDeviceToHost));
    if (!msa_out) return HITE_OK;
    if (!msa_off_out) return HITE_EINVAL;
    int64_t off = 0;
    for (int c = 0; c < n; c++) {
        int R = row_first[c + 1] - row_first[c];
        msa_off_out[c] = off;
        off += ((int64_t)R * cols_out[c] + 15) / 16 * 16;
    }
    if (off > msa_cap) return HITE_ECAP;
    e = dmo.up(msa_off_out, n * 8); if (e == hipSuccess) e = dmsa.alloc(off + 16);
    HITE_CHECK(ctx, e);
    if (sparse)
        rc = hite_star_msa_fill_sparse_dev(ctx, n, (uint8_t *)dw.p, (int64_t *)dwo.p, (int32_t *)dwl.p, (int32_t *)drf.p, (int64_t *)dob.p,
                                           (int32_t *)dnew.p, (int32_t *)dlast.p, (int64_t *)dmo.p, (uint8_t *)dmsa.p, nullptr);
    else
        rc = hite_star_msa_fill_dev(ctx, n, (uint8_t *)dw.p, (int64_t *)dwo.p, (int32_t *)dwl.p, (int32_t *)drf.p, (int64_t *)dob.p,
                                    (int32_t *)dcols.p, (int64_t *)dmo.p, (uint8_t *)dmsa.p, nullptr);
    if (rc) return rc;
    HITE_CHECK(ctx, hipDeviceSynchronize());
    HITE_CHECK(ctx, hipMemcpy(msa_out, dmsa.p, off, hipMemcpyDeviceToHost));
    return HITE_OK;
}

extern "C" int hite_star_msa(hite_ctx *ctx, int32_t n, const uint8_t *win, const int64_t *win_off,
                             const int32_t *row_first, int32_t *cols_out, int64_t msa_cap, uint8_t *msa_out,
                             int64_t *msa_off_out) {
    return star_msa_host(ctx, n, win, win_off, row_first, cols_out, msa_cap, msa_out, msa_off_out, false);
}
// same call protocol; the alignments come back with the sparse columns already removed (cols_out = surviving columns)
extern "C" int hite_star_msa_sparse(hite_ctx *ctx, int32_t n, const uint8_t *win, const int64_t *win_off,
                                    const int32_t *row_first, int32_t *cols_out, int64_t msa_cap, uint8_t *msa_out,
                                    int64_t *msa_off_out) {
    return star_msa_host(ctx, n, win, win_off, row_first, cols_out, msa_cap, msa_out, msa_off_out, true);
}
