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
typedef void (*kern_t)(unsigned*, int, unsigned long long*);
int main()
{
    struct { const char* name; kern_t k; int per; } ks[] = {
        {"v_add_u32", k_add, 1}, {"v_and_b32", k_and, 1}, {"v_lshlrev_b32", k_lshl, 1}, {"v_min_i32", k_min, 1}, {"v_perm_b32", k_perm, 1},
        {"v_alignbit_b32", k_alignbit, 1}, {"v_dot4_i32_i8", k_dot4, 1}, {"v_dot4c_i32_i8", k_dot4c, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_and_or_b32", k_andor, 1},
        {"v_lshl_add_u32", k_lshladd, 1}, {"v_mad_u32_u24", k_mad24, 1}, {"v_mul_lo_u32", k_mullo, 1}, {"v_pk_min_i16", k_pkmin16, 1}, {"v_pk_add_i16", k_pkadd16, 1},
        {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_fma_f32", k_fma, 1}, {"v_cndmask_b32", k_cndmask, 1}, {"v_min_i32_sdwa", k_minsdwa, 1}, {"v_mov_b32_dpp", k_mov_dpp, 1},
        {"v_readlane+v_add", k_readlane, 2} };
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4); hipMalloc(&cyc, (size_t)cus * 8 * 8);
    const int iters = 2000;
    printf("# cycles per wave64 instruction per SIMD (in-kernel cycle counter of one wave / instructions issued by the SIMD's waves)\n");
    printf("%-20s %10s %10s %10s\n", "opcode", "1 wave", "2 waves", "4 waves");
    for (auto& e : ks) {
        printf("%-20s", e.name);
        for (int wps : {1, 2, 4}) {                       // waves per SIMD: blocks of 256 threads = one wave per SIMD each
            const int blocks = cus * wps;
            e.k<<<blocks, 256>>>(out, 10, cyc); hipDeviceSynchronize();
            e.k<<<blocks, 256>>>(out, iters, cyc); hipDeviceSynchronize();
            std::vector<unsigned long long> h(blocks);
            hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (auto x : h) mean += (double)x; mean /= blocks;
            printf(" %10.2f", mean / ((double)iters * REP * e.per * wps));
        }
        printf("\n");
    }
    return 0;
}
