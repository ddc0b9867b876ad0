// fetch_calib.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of the seed stage,
// against byte counts that are known by construction (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access
// pattern before trusting an absolute"; VERDICT r4 #3b).  One kernel per pattern, so that the per-kernel PMC table separates them:
//   k_stream16     coalesced 16-byte loads, every byte of the buffer once                    (the guide's case: counted at 1/2)
//   k_stream8_nt   coalesced 8-byte non-temporal loads (the key stream of k_scan_hits)
//   k_gather16     ONE 16-byte load per lane at a random place of a 2 GiB buffer, every lane another 64-byte line
//                  (the target windows of k_scan_hits: footprint far above the 256 MiB memory-side cache; N lines of 64 B must move)
//   k_gather16x2   two 16-byte loads per lane, 16 bytes apart, in one random 64-byte line (a hit's left + right window)
//   k_store4_nt    coalesced 4-byte non-temporal stores (the summaries)
//   k_store8_nt    coalesced 8-byte non-temporal stores (the keys of k_fill_hits2)
// Build + run on the GPU box (tools/fetch_calib.sh): hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); return 1; } } while (0)
typedef unsigned long long u64; typedef unsigned u32;
__global__ void __launch_bounds__(256) k_stream16(const uint4* __restrict__ p, u64 n16, u32* __restrict__ sink)
{
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) k_stream8_nt(const u64* __restrict__ p, u64 n8, u32* __restrict__ sink)
{
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n8; i += (u64)gridDim.x * 256) acc ^= __builtin_nontemporal_load(p + i);
    if (acc == 0x12345678ull) sink[0] = (u32)acc;
}
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
// line index = a bijection of the gather's index onto [0, nlines): every gather another line (nlines a power of two, odd multiplier)
__global__ void __launch_bounds__(256) k_gather16(const unsigned char* __restrict__ p, u64 nlines, u64 ngather, u32* __restrict__ sink)
{
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < ngather; i += (u64)gridDim.x * 256) {
        const u64 line = (i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & (nlines - 1);
        const u32 off = (u32)(mix(i) & 3u) * 16u;
        const uint4 v = *reinterpret_cast<const uint4*>(p + line * 64 + off); acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) k_gather16x2(const unsigned char* __restrict__ p, u64 nlines, u64 ngather, u32* __restrict__ sink)
{
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < ngather; i += (u64)gridDim.x * 256) {
        const u64 line = (i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & (nlines - 1);
        const u32 off = (u32)(mix(i) % 3u) * 16u;
        const uint4 v = *reinterpret_cast<const uint4*>(p + line * 64 + off), w = *reinterpret_cast<const uint4*>(p + line * 64 + off + 16);
        acc ^= v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y ^ w.z ^ w.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) k_store4_nt(u32* __restrict__ p, u64 n4)
{ for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n4; i += (u64)gridDim.x * 256) __builtin_nontemporal_store((u32)i, p + i); }
__global__ void __launch_bounds__(256) k_store8_nt(u64* __restrict__ p, u64 n8)
{ for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n8; i += (u64)gridDim.x * 256) __builtin_nontemporal_store(i, p + i); }

int main()
{
    const u64 BYTES = 2ull << 30, NLINES = BYTES / 64, NG = NLINES / 2;      // 2 GiB; the gathers touch half of its lines, each once
    unsigned char* buf; u32* sink;
    CHECK(hipMalloc(&buf, BYTES)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, BYTES)); CHECK(hipDeviceSynchronize());
    const dim3 g(256 * 16), b(256);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_stream16, g, b, 0, 0, (const uint4*)buf, BYTES / 16, sink);
        hipLaunchKernelGGL(k_stream8_nt, g, b, 0, 0, (const u64*)buf, BYTES / 8, sink);
        hipLaunchKernelGGL(k_gather16, g, b, 0, 0, buf, NLINES, NG, sink);
        hipLaunchKernelGGL(k_gather16x2, g, b, 0, 0, buf, NLINES, NG, sink);
        hipLaunchKernelGGL(k_store4_nt, g, b, 0, 0, (u32*)buf, BYTES / 4);
        hipLaunchKernelGGL(k_store8_nt, g, b, 0, 0, (u64*)buf, BYTES / 8);
        CHECK(hipDeviceSynchronize());
    }
    // the byte counts known by construction, per launch (the PMC table's rows are matched to these by kernel name)
    printf("known_bytes k_stream16 %llu\nknown_bytes k_stream8_nt %llu\nknown_bytes k_gather16 %llu\nknown_bytes k_gather16x2 %llu\nknown_bytes k_store4_nt %llu\nknown_bytes k_store8_nt %llu\n",
           BYTES, BYTES, NG * 64, NG * 64, BYTES, BYTES);
    return 0;
}
