// valu_issue_bench.hip -- wave64 issue rate of the integer / DPP / bit instructions the alignment kernels are made of,
// measured on the device it runs on (MI355X: 256 CUs x 4 SIMDs).  Every thread runs ITER iterations of a block of
// 8 independent dependency chains x 8 instructions, so that neither latency nor the instruction cache limits the rate;
// the launch fills every SIMD with `waves` wavefronts.  Output: one line per (instruction, waves per SIMD):
//   G wave-instructions / s, cycles per wave-instruction per SIMD (at the clock the run measured with s_memtime-free
//   wall time and the device's reported clock).
// Build / run:  hipcc -O3 --offload-arch=gfx950 tools/valu_issue_bench.hip -o /tmp/valu_bench && /tmp/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at line %d\n", __LINE__); return 1; } } while (0)

#define ITER 4096

#define CHAINS8(OP)                                                                                   \
    OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")
#define BLOCK64(OP) CHAINS8(OP) CHAINS8(OP) CHAINS8(OP) CHAINS8(OP) CHAINS8(OP) CHAINS8(OP) CHAINS8(OP) CHAINS8(OP)

#define KERNEL(NAME, OP)                                                                              \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed) {                       \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        uint32_t k = seed | 1u, s = seed & 3u;                                                        \
        asm volatile("s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x55555555\n\ts_mov_b64 vcc, s[20:21]" ::: "s20", "s21", "vcc");                  \
        for (int it = 0; it < ITER; it++) {                                                           \
            asm volatile(BLOCK64(OP)                                                                  \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(k), "v"(s)                                                             \
                         : "vcc", "s20", "s21");                                                                    \
        }                                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;           \
    }

#define OP_ADD(R) "v_add_u32 " R ", " R ", %8\n\t"
#define OP_XOR(R) "v_xor_b32 " R ", " R ", %8\n\t"
#define OP_OR3(R) "v_or3_b32 " R ", " R ", %8, %9\n\t"
#define OP_ANDOR(R) "v_and_or_b32 " R ", " R ", %8, %9\n\t"
#define OP_ALIGNBIT(R) "v_alignbit_b32 " R ", %8, " R ", %9\n\t"
#define OP_LSHL(R) "v_lshlrev_b32 " R ", 1, " R "\n\t"
#define OP_BCNT(R) "v_bcnt_u32_b32 " R ", %8, " R "\n\t"
#define OP_PERM(R) "v_perm_b32 " R ", " R ", %8, %9\n\t"
#define OP_MAX3(R) "v_max3_i32 " R ", " R ", %8, %9\n\t"
#define OP_BFI(R) "v_bfi_b32 " R ", %8, " R ", %9\n\t"
#define OP_ADDC(R) "v_addc_co_u32 " R ", vcc, " R ", %8, vcc\n\t"
#define OP_ADD_DPP(R) "v_add_u32_dpp " R ", " R ", %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define OP_MOV_DPP_WSHR(R) "v_mov_b32_dpp " R ", " R " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define OP_CNDMASK(R) "v_cndmask_b32 " R ", " R ", %8, vcc\n\t"
#define OP_LSHLADD(R) "v_lshl_add_u32 " R ", " R ", 1, %8\n\t"
#define OP_CNDMASK64(R) "v_cndmask_b32_e64 " R ", " R ", %8, s[20:21]\n\t"
#define OP_CMP_CND(R) "v_cmp_lt_u32_e32 vcc, " R ", %8\n\tv_cndmask_b32_e32 " R ", " R ", %9, vcc\n\t"
#define OP_BFE(R) "v_bfe_u32 " R ", " R ", 1, 31\n\t"
#define OP_AND(R) "v_and_b32 " R ", " R ", %8\n\t"
#define OP_OR(R) "v_or_b32 " R ", " R ", %8\n\t"
#define OP_NOT(R) "v_not_b32 " R ", " R "\n\t"
#define OP_BITOP3(R) "v_bitop3_b32 " R ", " R ", %8, %9 bitop3:0x96\n\t"
#define OP_LSHR(R) "v_lshrrev_b32 " R ", 1, " R "\n\t"
#define OP_MIN(R) "v_min_i32 " R ", " R ", %8\n\t"
#define OP_SUB(R) "v_sub_u32 " R ", " R ", %8\n\t"
#define OP_FFBH(R) "v_ffbh_u32 " R ", " R "\n\t"
#define OP_MOV(R) "v_mov_b32 " R ", %8\n\t"
#define OP_XAD(R) "v_xad_u32 " R ", " R ", %8, %9\n\t"
#define OP_ADD3(R) "v_add3_u32 " R ", " R ", %8, %9\n\t"
#define OP_LSHLOR(R) "v_lshl_or_b32 " R ", " R ", 1, %8\n\t"
#define OP_BFEI(R) "v_bfe_i32 " R ", " R ", 1, 1\n\t"
#define OP_ASHR(R) "v_ashrrev_i32 " R ", 1, " R "\n\t"
#define OP_CMPEQ(R) "v_cmp_eq_u32 vcc, " R ", %8\n\t"
#define OP_LSHRV(R) "v_lshrrev_b32 " R ", %9, " R "\n\t"
#define OP_LSHLV(R) "v_lshlrev_b32 " R ", %9, " R "\n\t"
#define OP_ADDCO(R) "v_add_co_u32 " R ", vcc, " R ", %8\n\t"
// 64-bit operands (register pairs)
#define OP_LSHR64(R) "v_lshrrev_b64 " R ", 1, " R "\n\t"
#define OP_LSHL64(R) "v_lshlrev_b64 " R ", 1, " R "\n\t"
#define OP_LSHLADD64(R) "v_lshl_add_u64 " R ", " R ", 0, %8\n\t"
#define OP_LSHR64V(R) "v_lshrrev_b64 " R ", %9, " R "\n\t"

#define KERNEL64(NAME, OP)                                                                            \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed) {                       \
        unsigned long long a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        unsigned long long k = seed | 1u; uint32_t s = seed & 3u;                                     \
        for (int it = 0; it < ITER; it++) {                                                           \
            asm volatile(BLOCK64(OP)                                                                  \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(k), "v"(s)                                                             \
                         : "vcc");                                                                    \
        }                                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);           \
    }

KERNEL(k_add, OP_ADD)
KERNEL(k_xor, OP_XOR)
KERNEL(k_or3, OP_OR3)
KERNEL(k_andor, OP_ANDOR)
KERNEL(k_alignbit, OP_ALIGNBIT)
KERNEL(k_lshl, OP_LSHL)
KERNEL(k_bcnt, OP_BCNT)
KERNEL(k_perm, OP_PERM)
KERNEL(k_max3, OP_MAX3)
KERNEL(k_bfi, OP_BFI)
KERNEL(k_addc, OP_ADDC)
KERNEL(k_add_dpp, OP_ADD_DPP)
KERNEL(k_mov_dpp_wshr, OP_MOV_DPP_WSHR)
KERNEL(k_cndmask, OP_CNDMASK)
KERNEL(k_lshladd, OP_LSHLADD)
KERNEL(k_cndmask64, OP_CNDMASK64)
KERNEL(k_cmp_cnd, OP_CMP_CND)
KERNEL(k_bfe, OP_BFE)
KERNEL(k_and, OP_AND)
KERNEL(k_or, OP_OR)
KERNEL(k_not, OP_NOT)
KERNEL(k_bitop3, OP_BITOP3)
KERNEL(k_lshr, OP_LSHR)
KERNEL(k_min, OP_MIN)
KERNEL(k_sub, OP_SUB)
KERNEL(k_ffbh, OP_FFBH)
KERNEL(k_mov, OP_MOV)
KERNEL(k_xad, OP_XAD)
KERNEL(k_add3, OP_ADD3)
KERNEL(k_lshlor, OP_LSHLOR)
KERNEL(k_bfei, OP_BFEI)
KERNEL(k_ashr, OP_ASHR)
KERNEL(k_cmpeq, OP_CMPEQ)
KERNEL(k_lshrv, OP_LSHRV)
KERNEL(k_lshlv, OP_LSHLV)
KERNEL(k_addco, OP_ADDCO)
KERNEL64(k_lshr64, OP_LSHR64)
KERNEL64(k_lshl64, OP_LSHL64)
KERNEL64(k_lshladd64, OP_LSHLADD64)
KERNEL64(k_lshr64v, OP_LSHR64V)

typedef void (*kfn)(uint32_t *, uint32_t);
struct Entry { const char *name; kfn fn; };

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    const double clk_ghz = prop.clockRate / 1e6;
    printf("# device %s, %d CUs, clock %.3f GHz (reported max); ITER %d x 64 instructions per thread\n", prop.gcnArchName, cus, clk_ghz, ITER);
    printf("# %-18s %6s %14s %22s\n", "instruction", "w/SIMD", "G wave-inst/s", "cycles/inst/SIMD @max");
    Entry tab[] = {{"v_add_u32", k_add}, {"v_xor_b32", k_xor}, {"v_or3_b32", k_or3}, {"v_and_or_b32", k_andor}, {"v_alignbit_b32", k_alignbit},
                   {"v_lshlrev_b32", k_lshl}, {"v_bcnt_u32_b32", k_bcnt}, {"v_perm_b32", k_perm}, {"v_max3_i32", k_max3}, {"v_bfi_b32", k_bfi},
                   {"v_addc_co_u32", k_addc}, {"v_add_u32_dpp row", k_add_dpp}, {"v_mov_dpp wave_shr", k_mov_dpp_wshr}, {"v_cndmask_b32", k_cndmask},
                   {"v_lshl_add_u32", k_lshladd}, {"v_cndmask_e64 sgpr", k_cndmask64}, {"v_cmp+v_cndmask (2)", k_cmp_cnd},
                   {"v_bfe_u32", k_bfe}, {"v_and_b32", k_and}, {"v_or_b32", k_or}, {"v_not_b32", k_not}, {"v_bitop3_b32", k_bitop3},
                   {"v_lshrrev_b32", k_lshr}, {"v_min_i32", k_min}, {"v_sub_u32", k_sub},
                   {"v_ffbh_u32", k_ffbh}, {"v_mov_b32", k_mov}, {"v_xad_u32", k_xad}, {"v_add3_u32", k_add3}, {"v_lshl_or_b32", k_lshlor},
                   {"v_bfe_i32", k_bfei}, {"v_ashrrev_i32", k_ashr}, {"v_cmp_eq_u32 vcc", k_cmpeq}, {"v_lshrrev_b32 var", k_lshrv},
                   {"v_lshlrev_b32 var", k_lshlv}, {"v_add_co_u32", k_addco}, {"v_lshrrev_b64", k_lshr64}, {"v_lshlrev_b64", k_lshl64},
                   {"v_lshl_add_u64", k_lshladd64}, {"v_lshrrev_b64 var", k_lshr64v}};
    uint32_t *out;
    if (hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * 4) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Entry &e : tab) {
        for (int waves : {1, 2, 4, 8}) {
            // `waves` wavefronts per SIMD: blocks of 256 threads = one wave on each of the 4 SIMDs of a CU
            const int blocks = cus * waves;
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 12345u);   // warm-up
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 12345u + rep);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double winst = (double)blocks * 4 * ITER * 64;
            const double rate = winst / (best * 1e-3) / 1e9;
            const double cyc = (double)cus * 4 * clk_ghz / rate;
            printf("  %-18s %6d %14.1f %22.2f\n", e.name, waves, rate, cyc);
        }
    }
    return 0;
}
