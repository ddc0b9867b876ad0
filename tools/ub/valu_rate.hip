// valu_rate.hip -- how many cycles a SIMD of gfx950 needs per wave64 instruction, for the integer instructions the seed
// stage is made of (one number per opcode; MI355X_MICROARCH.md quotes 2 cycles for v_fma_f32 only).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHAINS 8
#define REP 64
#define KERNEL(NAME, ASM)                                                                         \
__global__ void __launch_bounds__(256) NAME(unsigned* out, int iters, unsigned long long* cyc)    \
{                                                                                                 \
    unsigned v[CHAINS]; unsigned a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x + 7u;      \
    for (int k = 0; k < CHAINS; k++) v[k] = a + k * 977u;                                          \
    const unsigned long long t0 = __builtin_readcyclecounter();                                    \
    for (int it = 0; it < iters; it++) {                                                           \
        _Pragma("unroll") for (int r = 0; r < REP / CHAINS; r++) {                                 \
            _Pragma("unroll") for (int k = 0; k < CHAINS; k++) asm volatile(ASM : "+v"(v[k]) : "v"(a), "v"(b)); \
        }                                                                                          \
    }                                                                                              \
    const unsigned long long t1 = __builtin_readcyclecounter();                                    \
    unsigned s = 0; for (int k = 0; k < CHAINS; k++) s ^= v[k];                                    \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                                       \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                               \
}
KERNEL(k_add,      "v_add_u32 %0, %0, %1")
KERNEL(k_and,      "v_and_b32 %0, %0, %1")
KERNEL(k_lshl,     "v_lshlrev_b32 %0, 3, %0")
KERNEL(k_min,      "v_min_i32 %0, %0, %1")
KERNEL(k_perm,     "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
KERNEL(k_dot4,     "v_dot4_i32_i8 %0, %1, %2, %0")
KERNEL(k_dot4c,    "v_dot4c_i32_i8 %0, %1, %2")
KERNEL(k_bfe,      "v_bfe_u32 %0, %0, 3, 8")
KERNEL(k_andor,    "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_lshladd,  "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL(k_mad24,    "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_mullo,    "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_pkmin16,  "v_pk_min_i16 %0, %0, %1")
KERNEL(k_pkadd16,  "v_pk_add_i16 %0, %0, %1")
KERNEL(k_bcnt,     "v_bcnt_u32_b32 %0, %1, %0")
KERNEL(k_fma,      "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_cndmask,  "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_minsdwa,  "v_min_i32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
KERNEL(k_mov_dpp,  "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0")
KERNEL(k_cnd64,    "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL(k_cnd_b,    "v_cndmask_b32 %0, %1, %2, vcc")
KERNEL(k_cmpcnd,   "v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL(k_cmpcnd64, "v_cmp_lt_i32 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]")
KERNEL(k_cmp_and,  "v_cmp_lt_i32 vcc, %0, %1\n v_and_b32 %0, %0, %2")
KERNEL(k_addc2,    "v_cmp_lt_i32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %2, vcc")
KERNEL(k_selarith, "v_sub_u32 %0, %0, %1\n v_ashrrev_i32 %0, 31, %0\n v_bfi_b32 %0, %0, %1, %2")
KERNEL(k_xor,      "v_xor_b32 %0, %0, %1")
KERNEL(k_or,       "v_or_b32 %0, %0, %1")
KERNEL(k_sub,      "v_sub_u32 %0, %0, %1")
KERNEL(k_lshr,     "v_lshrrev_b32 %0, 3, %0")
KERNEL(k_ashr,     "v_ashrrev_i32 %0, 3, %0")
KERNEL(k_max,      "v_max_i32 %0, %0, %1")
KERNEL(k_mov,      "v_mov_b32 %0, %1")
KERNEL(k_bfi,      "v_bfi_b32 %0, %0, %1, %2")
KERNEL(k_add3,     "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_cmp,      "v_cmp_lt_i32 vcc, %0, %1")
KERNEL(k_cmp_sdwa, "v_cmp_ge_i32_sdwa vcc, %0, %1 src0_sel:DWORD src1_sel:WORD_0")
KERNEL(k_cmpx_sdwa, "v_cmp_ge_i32_sdwa s[20:21], %0, %1 src0_sel:DWORD src1_sel:WORD_0")
KERNEL(k_addc,     "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL(k_bitop3,   "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8")
KERNEL(k_mul24,    "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_min3,     "v_min3_i32 %0, %0, %1, %2")
KERNEL(k_lshl_e64, "v_lshlrev_b32_e64 %0, 3, %0")
KERNEL(k_add_e64,  "v_add_u32_e64 %0, %0, %1")
typedef void (*kern_t)(unsigned*, int, unsigned long long*);
int main()
{
    struct { const char* name; kern_t k; int per; } ks[] = {
        {"v_add_u32", k_add, 1}, {"v_and_b32", k_and, 1}, {"v_lshlrev_b32", k_lshl, 1}, {"v_min_i32", k_min, 1}, {"v_perm_b32", k_perm, 1},
        {"v_alignbit_b32", k_alignbit, 1}, {"v_dot4_i32_i8", k_dot4, 1}, {"v_dot4c_i32_i8", k_dot4c, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_and_or_b32", k_andor, 1},
        {"v_lshl_add_u32", k_lshladd, 1}, {"v_mad_u32_u24", k_mad24, 1}, {"v_mul_lo_u32", k_mullo, 1}, {"v_pk_min_i16", k_pkmin16, 1}, {"v_pk_add_i16", k_pkadd16, 1},
        {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_fma_f32", k_fma, 1}, {"v_cndmask_b32", k_cndmask, 1}, {"v_min_i32_sdwa", k_minsdwa, 1}, {"v_mov_b32_dpp", k_mov_dpp, 1},
        {"v_readlane+v_add", k_readlane, 2}, {"v_cndmask_e64 sgpr", k_cnd64, 1}, {"v_cndmask indep", k_cnd_b, 1}, {"v_cmp+v_cndmask vcc", k_cmpcnd, 2}, {"v_cmp+v_cndmask sgpr", k_cmpcnd64, 2}, {"v_cmp+v_and", k_cmp_and, 2}, {"v_cmp+v_addc", k_addc2, 2}, {"sub+ashr+bfi", k_selarith, 3}, {"v_xor_b32", k_xor, 1}, {"v_or_b32", k_or, 1}, {"v_sub_u32", k_sub, 1}, {"v_lshrrev_b32", k_lshr, 1}, {"v_ashrrev_i32", k_ashr, 1},
        {"v_max_i32", k_max, 1}, {"v_mov_b32", k_mov, 1}, {"v_bfi_b32", k_bfi, 1}, {"v_add3_u32", k_add3, 1}, {"v_cmp_lt_i32 vcc", k_cmp, 1}, {"v_cmp_sdwa vcc", k_cmp_sdwa, 1}, {"v_cmp_sdwa sgpr", k_cmpx_sdwa, 1},
        {"v_addc_co_u32", k_addc, 1}, {"v_bitop3_b32", k_bitop3, 1}, {"v_mul_u32_u24", k_mul24, 1}, {"v_min3_i32", k_min3, 1}, {"v_lshlrev_b32_e64", k_lshl_e64, 1}, {"v_add_u32_e64", k_add_e64, 1} };
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)cus * 16 * 256 * 4); hipMalloc(&cyc, (size_t)cus * 16 * 8);
    const int iters = 2000;
    printf("# cycles per wave64 instruction per SIMD (in-kernel cycle counter of one wave / instructions issued by the SIMD's waves)\n");
    printf("%-20s %10s %10s %10s %10s   %s\n", "opcode", "1 wave", "2 waves", "4 waves", "8 waves", "8 waves: SIMD cycles per instruction by the wall clock at the core clock rocm-smi shows (hipEvent time x MHz / instructions per SIMD)");
    int mhz = 0; hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, 0);   // kHz
    for (auto& e : ks) {
        printf("%-20s", e.name);
        for (int wps : {1, 2, 4, 8}) {                       // waves per SIMD: blocks of 256 threads = one wave per SIMD each
            const int blocks = cus * wps;
            e.k<<<blocks, 256>>>(out, 10, cyc); hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0); e.k<<<blocks, 256>>>(out, iters, cyc); hipEventRecord(e1); hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(blocks);
            hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (auto x : h) mean += (double)x; mean /= blocks;
            printf(" %10.2f", mean / ((double)iters * REP * e.per * wps));
            if (wps == 8) printf("   %6.2f", (double)ms * 1e-3 * (double)mhz * 1e3 / ((double)iters * REP * e.per * wps));
        }
        printf("\n");
    }
    return 0;
}
