// gather_rate.hip -- what a wave64 vector-memory LOAD instruction costs the CU's address / L1 path on gfx950, by access pattern
// (16 bytes per lane unless said otherwise; the data is L2-resident, so this is the instruction path, not the memory behind it).
// k_scan_hits issues four such loads per raw hit (two target windows in one random 64-byte line, two query windows that the
// lanes of a wave mostly share), all at BYTE alignment.  Build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32; typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
__constant__ unsigned g_region = 1u << 20;   // bytes the gathers fall in (1 MiB fits every L2; argv[1] = MiB)
#define REGION g_region
// mode: how lane l of iteration i forms its byte offset
__device__ __forceinline__ u32 offs(int mode, u32 l, u32 i, u32 w)
{
    const u32 h = (l * 2654435761u + i * 40503u + w * 977u) * 2246822519u;
    const u32 line = (h >> 8) & (REGION / 64 - 1);
    switch (mode) {
    case 0: return ((i * 64 + l) * 16) & (REGION - 1);                 // coalesced, 16-byte aligned
    case 1: return line * 64 + ((h >> 4) & 3) * 16;                     // gather: a random line per lane, 16-byte aligned
    case 2: return line * 64 + ((h >> 4) & 7) * 4 + 4;                  // gather, dword aligned (not 16)
    case 3: return line * 64 + ((h >> 4) & 31) + 1;                     // gather, byte aligned (within the line: <= 48)
    case 4: return ((i * 4099u + w * 131u) & (REGION / 64 - 1)) * 64 + (l >> 3) + 5;   // "query-like": the wave's lanes within 8 bytes of each other, byte aligned
    case 5: return ((i * 4099u + w * 131u) & (REGION / 64 - 1)) * 64 + (l >> 5) * 16;  // query-like, 16-byte aligned: two addresses per wave
    case 7: return (line & 255) * 64 + ((h >> 4) & 31) + 1;               // gather inside 16 KiB (the L1 holds it), byte aligned
    case 8: return (line & 255) * 64 + ((h >> 4) & 3) * 16;               // gather inside 16 KiB, 16-byte aligned
    case 9: return line * 64 + ((h >> 4) & 1) * 16;                        // first chunk of an aligned PAIR / TRIPLE inside one line (offset 0 or 16)
    case 10: return line * 64 + ((h >> 4) & 7) * 4;                        // first window of a dword-aligned PAIR (offset 0..28)
    case 11: return ((i * 4099u + w * 131u) & (REGION / 64 - 1)) * 64 + ((l >> 3) & ~3u) + 4;   // query-like, dword aligned: the wave's lanes on 2-3 addresses
    case 6: return line * 64 + (((h >> 4) & 31) + 1 > 44 ? 44 : ((h >> 4) & 31) + 1);  // as 3 (second load of the pair adds 16 in the kernel)
    }
    return 0;
}
template <int MODE, int WIDTH /*dwords per lane*/, int PAIR /*loads per lane from consecutive 16-byte pieces*/>
__global__ void __launch_bounds__(256) k(const unsigned char* __restrict__ buf, u32* out, int iters, u64* cyc)
{
    const u32 l = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
    u32 acc = 0;
    const u64 t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i += 4) {
        u32x4 v[4]; u32x4 v2[4];
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) {
            const u32 o = offs(MODE, l, (u32)(i + k2), w);
            if (WIDTH == 4) v[k2] = __builtin_nontemporal_load((const u32x4*)(buf + o)) , v[k2] = *(const u32x4*)(buf + o);
            else if (WIDTH == 2) { u32x2 t; __builtin_memcpy(&t, buf + o, 8); v[k2] = (u32x4){t.x, t.y, 0, 0}; }
            else { u32 t; __builtin_memcpy(&t, buf + o, 4); v[k2] = (u32x4){t, 0, 0, 0}; }
            if (PAIR) __builtin_memcpy(&v2[k2], buf + o + 16, 16);
            if (PAIR == 3) { u32x4 t3; __builtin_memcpy(&t3, buf + o + 32, 16); v2[k2].y ^= t3.x ^ t3.w; }
        }
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) { acc ^= v[k2].x ^ v[k2].y ^ v[k2].z ^ v[k2].w; if (PAIR) acc ^= v2[k2].x ^ v2[k2].w ^ v2[k2].y; }
    }
    const u64 t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// the second window of a line asked for when the first one has come back (one round later): is it an L1 hit, and what does a hit cost?
template <int LAG>
__global__ void __launch_bounds__(256) k_lag(const unsigned char* __restrict__ buf, u32* out, int iters, u64* cyc)
{
    const u32 l = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
    u32 acc = 0; u32 prev[LAG];
    for (int q = 0; q < LAG; q++) prev[q] = offs(6, l, 0, w);
    for (int i = 0; i < iters; i++) {
        const u32 o = offs(6, l, (u32)i + 1, w);
        u32x4 a, b;
        __builtin_memcpy(&a, buf + o, 16);                 // first window of the new line
        __builtin_memcpy(&b, buf + prev[0] + 16, 16);      // second window of a line asked for LAG iterations ago
        acc ^= a.x ^ a.w ^ b.x ^ b.w;
#pragma unroll
        for (int q = 0; q + 1 < LAG; q++) prev[q] = prev[q + 1];
        prev[LAG - 1] = o;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = 0;
}
typedef void (*kern_t)(const unsigned char*, u32*, int, u64*);
int main(int argc, char** argv)
{
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned region = 1u << 20;
    if (argc > 1) region = (unsigned)atoi(argv[1]) << 20;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_region), &region, 4);
    int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    unsigned char* buf; u32* out; u64* cyc;
    (void)hipMalloc(&buf, (size_t)region + 4096); (void)hipMemset(buf, 1, (size_t)region + 4096);
    (void)hipMalloc(&out, (size_t)cus * 8 * 256 * 4); (void)hipMalloc(&cyc, (size_t)cus * 8 * 8);
    struct { const char* name; kern_t f; int per; } ks[] = {
        {"dwordx4 coalesced, 16-B aligned", k<0, 4, 0>, 1}, {"dwordx4 gather (random line per lane), 16-B aligned", k<1, 4, 0>, 1},
        {"dwordx4 gather, dword aligned", k<2, 4, 0>, 1}, {"dwordx4 gather, BYTE aligned", k<3, 4, 0>, 1},
        {"dwordx4 PAIR in one random line, byte aligned (the target windows)", k<6, 4, 1>, 2},
        {"dwordx4 query-like (lanes within 8 bytes), byte aligned", k<4, 4, 0>, 1}, {"dwordx4 query-like, 16-B aligned, 2 addresses per wave", k<5, 4, 0>, 1},
        {"dwordx4 gather inside 16 KiB (L1 hits), byte aligned", k<7, 4, 0>, 1}, {"dwordx4 gather inside 16 KiB (L1 hits), 16-B aligned", k<8, 4, 0>, 1},
        {"PAIR, second window one iteration after the first (per pair / 2)", k_lag<1>, 2}, {"PAIR, second window two iterations later", k_lag<2>, 2}, {"PAIR, second window four iterations later", k_lag<4>, 2},
        {"PAIR in one random line, both 16-B aligned (per instruction)", k<9, 4, 1>, 2}, {"TRIPLE in one random line, 16-B aligned (per instruction)", k<9, 4, 3>, 3},
        {"PAIR in one random line, dword aligned (per instruction)", k<10, 4, 1>, 2}, {"dwordx4 query-like, dword aligned", k<11, 4, 0>, 1},
        {"dwordx2 gather, byte aligned", k<3, 2, 0>, 1}, {"dword gather, byte aligned", k<3, 1, 0>, 1}, {"dword gather, dword aligned", k<2, 1, 0>, 1} };
    printf("# core cycles of one CU per wave64 load instruction (wall clock x %d MHz / instructions per CU), all CUs busy, gathers inside %u MiB\n", khz / 1000, region >> 20);
    printf("%-72s %8s %8s %8s\n", "pattern", "4 w/CU", "8 w/CU", "16 w/CU");
    const int iters = 4000;
    for (auto& e : ks) {
        printf("%-72s", e.name);
        for (int bpc : {1, 2, 4}) {
            const int blocks = cus * bpc;
            e.f<<<blocks, 256>>>(buf, out, 8, cyc); (void)hipDeviceSynchronize();
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0); e.f<<<blocks, 256>>>(buf, out, iters, cyc); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            printf(" %8.1f", (double)ms * 1e-3 * (double)khz * 1e3 / ((double)iters * e.per * 4 * bpc));   // (4 waves per block)
        }
        printf("\n");
    }
    return 0;
}
