// seed_kernels.hip -- gfx950 kernels of the seed stage: position-table build (B1), raw-hit
// enumeration, bucket ordering and the bucket-serial X-drop extender (B2).
//
// Decomposition: the only cross-hit state of the reference's HSP search is
// diagEnd[hashedDiag] (src/seed_search.c:1081-1126, 2612-2616, 2785-2789), so the exact
// parallel form is 65,536 independent, order-preserving streams.  Hits are enumerated in the
// reference's order (count -> scan -> fill gives every hit its discovery index; the table itself
// is probed in seed-word order), scanned independently of the hash (phase A, k_probe_hits), stably
// partitioned by the 16 hash bits (LSD radix sort restricted to key bits 32..47), and each bucket
// is then walked by one lane with diagEnd[h] in a register (phase B, k_extend).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include "lz_ctx.hpp"

#define LZ_TPB 256

// ------------------------------------------------------------------------------------------
// byte -> code translation (one pass per sequence; the table folds charToBits and the score
// class of the byte, see lz_common.hpp)
__global__ void __launch_bounds__(LZ_TPB)
k_encode(const u8* __restrict__ raw, u8* __restrict__ code, u32 len, const u8* __restrict__ cls)
{
    __shared__ u8 tab[256];
    tab[threadIdx.x] = cls[threadIdx.x];
    __syncthreads();
    const u32 nvec = (len + 15u) >> 4;                 // buffers are padded: whole 16-byte groups are safe
    for (u32 v = blockIdx.x * LZ_TPB + threadIdx.x; v < nvec; v += gridDim.x * LZ_TPB) {
        uint4 x = reinterpret_cast<const uint4*>(raw)[v];
        u32 w[4] = { x.x, x.y, x.z, x.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 a = w[k];
            w[k] = (u32)tab[a & 255u] | ((u32)tab[(a >> 8) & 255u] << 8) |
                   ((u32)tab[(a >> 16) & 255u] << 16) | ((u32)tab[a >> 24] << 24);
        }
        reinterpret_cast<uint4*>(code)[v] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// 4-bit class codes for phase A: nib[b] = class(code[2b]) | class(code[2b+1]) << 4, over the whole padded
// code array (padding bytes carry class 0 there and here)
__global__ void __launch_bounds__(LZ_TPB)
k_pack_nibbles(const u8* __restrict__ code, u8* __restrict__ nib, size_t nbytes)
{
    const size_t b = (size_t)blockIdx.x * LZ_TPB + threadIdx.x;
    if (b < nbytes) nib[b] = (u8)((code[2 * b] & 7u) | ((code[2 * b + 1] & 7u) << 4));
}
int lzk_pack_nibbles(LzCtx& c, const u8* code_alloc, u8* nib, size_t nbytes)
{
    hipLaunchKernelGGL(k_pack_nibbles, dim3((unsigned)((nbytes + LZ_TPB - 1) / LZ_TPB)), dim3(LZ_TPB), 0, c.stream, code_alloc, nib, nbytes);
    LZ_HIP(hipGetLastError());
    return 0;
}

int lzk_encode(LzCtx& c, const u8* raw, u8* code, u32 len, const u8* cls256_dev)
{
    if (len == 0) return 0;
    u32 nvec = (len + 15u) >> 4;
    u32 blocks = (nvec + LZ_TPB - 1) / LZ_TPB; if (blocks > 4096) blocks = 4096;
    c.timer.begin("k_encode", c.stream);
    hipLaunchKernelGGL(k_encode, dim3(blocks), dim3(LZ_TPB), 0, c.stream, raw, code, len, cls256_dev);
    c.timer.end(c.stream);
    LZ_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// B1: position table.  One (word, end position) pair per target position, positions emitted in
// DESCENDING order so that the stable radix sort by word leaves every word's list in the order
// the reference's chain walk yields it (most recent first, src/pos_table.c:1341-1344).
__global__ void __launch_bounds__(LZ_TPB)
k_table_words(const u8* __restrict__ tcode, u32 start, u32 end, u32 step, LzSeedDev sd,
              u32* __restrict__ keys, u32* __restrict__ vals, u32 n)
{
    u32 j = blockIdx.x * LZ_TPB + threadIdx.x;
    if (j >= n) return;
    u32 p = end - j;                                    // window is [p-L, p)
    u32 key = 1u << sd.weight;                          // "no word": sorts after every real word
    if (p >= start + (u32)sd.length && (p % step) == 0) {
        u32 packed;
        if (lz_window_word(tcode, p, sd, packed)) key = packed;
    }
    keys[j] = key; vals[j] = p;
}

// wstart[w] = index of the first sorted entry with key >= w, for w in [0, nwords]
__global__ void __launch_bounds__(LZ_TPB)
k_key_bounds_u32(const u32* __restrict__ keys, u32 n, u32 nwords, u32* __restrict__ wstart)
{
    u32 i = blockIdx.x * LZ_TPB + threadIdx.x;
    if (i > n) return;
    s64 kp = (i == 0) ? -1 : (s64)keys[i - 1];
    s64 k  = (i == n) ? (s64)nwords : (s64)keys[i];
    if (k > (s64)nwords) k = nwords;
    if (kp > (s64)nwords) kp = nwords;
    for (s64 w = kp + 1; w <= k; w++) wstart[w] = i;
}

int lzk_table_build(LzCtx& c)
{
    const lz_table_geom& g = c.geom;
    const u32 n = g.end - g.start;
    const u32 nwords = 1u << c.seed.weight;
    int rc;
    if ((rc = c.tb_keys.ensure((size_t)n * 4))) return rc;
    if ((rc = c.tb_vals.ensure((size_t)n * 4))) return rc;
    if ((rc = c.tb_keys2.ensure((size_t)n * 4))) return rc;
    if ((rc = c.tb_vals2.ensure((size_t)n * 4))) return rc;
    if ((rc = c.wstart.ensure(((size_t)nwords + 1) * 4))) return rc;

    c.timer.begin("k_table_words", c.stream);
    hipLaunchKernelGGL(k_table_words, dim3((n + LZ_TPB - 1) / LZ_TPB), dim3(LZ_TPB), 0, c.stream,
                       c.target.code_base(), g.start, g.end, g.step, c.seed,
                       c.tb_keys.as<u32>(), c.tb_vals.as<u32>(), n);
    c.timer.end(c.stream);
    LZ_HIP(hipGetLastError());

    size_t tmp = 0;
    LZ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, c.tb_keys.as<u32>(), c.tb_keys2.as<u32>(),
                                     c.tb_vals.as<u32>(), c.tb_vals2.as<u32>(), (size_t)n, 0u,
                                     (unsigned)c.seed.weight + 1u, c.stream));
    if ((rc = c.sort_tmp.ensure(tmp))) return rc;
    c.timer.begin("rocprim_sort_table", c.stream);
    LZ_HIP(rocprim::radix_sort_pairs(c.sort_tmp.p, tmp, c.tb_keys.as<u32>(), c.tb_keys2.as<u32>(),
                                     c.tb_vals.as<u32>(), c.tb_vals2.as<u32>(), (size_t)n, 0u,
                                     (unsigned)c.seed.weight + 1u, c.stream));
    c.timer.end(c.stream);

    c.timer.begin("k_key_bounds_u32", c.stream);
    hipLaunchKernelGGL(k_key_bounds_u32, dim3((n + 1 + LZ_TPB - 1) / LZ_TPB), dim3(LZ_TPB), 0, c.stream,
                       c.tb_keys2.as<u32>(), n, nwords, c.wstart.as<u32>());
    c.timer.end(c.stream);
    LZ_HIP(hipGetLastError());

    u32 nw = 0;
    LZ_HIP(hipMemcpyAsync(&nw, c.wstart.as<u32>() + nwords, 4, hipMemcpyDeviceToHost, c.stream));
    LZ_HIP(hipStreamSynchronize(c.stream));
    c.timer.resolve();
    c.num_words = nw;
    if ((rc = c.wpos.ensure((size_t)(nw ? nw : 1) * 4))) return rc;
    LZ_HIP(hipMemcpyAsync(c.wpos.p, c.tb_vals2.p, (size_t)nw * 4, hipMemcpyDeviceToDevice, c.stream));
    LZ_HIP(hipStreamSynchronize(c.stream));
    return 0;                                            // scratch stays allocated for the next rebuild
}

// CSR -> the reference's last[]/prev[] (src/pos_table.h:126-165): one thread per word
__global__ void __launch_bounds__(LZ_TPB)
k_table_export(const u32* __restrict__ wstart, const u32* __restrict__ wpos, u32 nwords,
               u32 adj_start, u32 step, u32* __restrict__ last, u32* __restrict__ prev)
{
    u32 w = blockIdx.x * LZ_TPB + threadIdx.x;
    if (w >= nwords) return;
    u32 a = wstart[w], b = wstart[w + 1];
    if (a == b) return;
    if (last) last[w] = (wpos[a] - adj_start) / step;
    if (prev)
        for (u32 j = a; j < b; j++)
            prev[(wpos[j] - adj_start) / step] = (j + 1 < b) ? (wpos[j + 1] - adj_start) / step : 0xFFFFFFFFu;
}

int lzk_table_export(LzCtx& c, u32* last_dev, u32* prev_dev, u32 prev_entries)
{
    const u32 nwords = 1u << c.seed.weight;
    if (last_dev) LZ_HIP(hipMemsetAsync(last_dev, 0, (size_t)nwords * 4, c.stream));
    if (prev_dev) LZ_HIP(hipMemsetAsync(prev_dev, 0, (size_t)prev_entries * 4, c.stream));
    u32 adj = c.geom.start - (c.geom.start % c.geom.step);
    hipLaunchKernelGGL(k_table_export, dim3((nwords + LZ_TPB - 1) / LZ_TPB), dim3(LZ_TPB), 0, c.stream,
                       c.wstart.as<u32>(), c.wpos.as<u32>(), nwords, adj, c.geom.step, last_dev, prev_dev);
    LZ_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// B2 step 1: raw hits per query position (private_hit_search + find_table_matches,
// src/seed_search.c:491-571, 810-875, without calling the processor yet).
//
// The table probes are random 4-byte reads into a 64 MiB + 4*Tlen byte structure: taken in query order
// every probe costs a whole cache line from the fabric (measured: 290 GB per step for k_fill_hits alone).
// The query positions are therefore visited in the order of their seed words: (word, position) pairs are
// radix-sorted once per search, and both the count and the fill kernels walk that list, so that each of
// the 13 probe streams (word ^ flip) moves through wstart[] / wpos[] front to back.  What is written --
// cnt[position] and the hits at off[position] -- is indexed by position, so the discovery order of the
// hits is untouched.
__global__ void __launch_bounds__(LZ_TPB)
k_pack_words(const u8* __restrict__ qcode, u32 lo, u32 hi, LzSeedDev sd, u32* __restrict__ pk, u32* __restrict__ iv)
{
    // the block's LZ_TPB windows overlap in all but one byte: the codes go through LDS once
    __shared__ u8 win[LZ_TPB + 32];
    const u32 i = blockIdx.x * LZ_TPB + threadIdx.x;
    const u32 L = (u32)sd.length;
    const s64 first = (s64)lo + (s64)blockIdx.x * LZ_TPB + 1 - (s64)L;     // window start of the block's first position (>= -31: inside the padding)
    for (u32 k = threadIdx.x; k < LZ_TPB + L - 1; k += LZ_TPB) win[k] = (first + (s64)k < (s64)hi) ? qcode[first + (s64)k] : (u8)LZ_CODE_INVALID;
    __syncthreads();
    if (i < hi - lo) {
        const u32 pos2 = lo + i + 1;
        u64 w = 0; u32 bad = 0;
        for (u32 k = 0; k < L; k++) { const u32 c = win[threadIdx.x + k]; bad |= c; w = (w << 2) | LZ_CODE_BITS(c); }
        const bool valid = pos2 >= lo + L && !(bad & LZ_CODE_INVALID);     // window inside the interval, only ACGT
        pk[i] = valid ? lz_apply_seed(sd, w) : (1u << sd.weight);   // "no word" sorts after every real word
        iv[i] = i;
    }
}

// "words in seq 2" (the reference's counter): the entries of the sorted list in front of the first "no word"
__global__ void k_count_words(const u32* __restrict__ sk, u32 n, u32 none, u64* __restrict__ n_words)
{
    u32 a = 0, b = n;
    while (a < b) { const u32 m = a + ((b - a) >> 1); if (sk[m] < none) a = m + 1; else b = m; }
    *n_words += a;
}

__global__ void __launch_bounds__(LZ_TPB)
k_count_sorted(const u32* __restrict__ sk, const u32* __restrict__ sv, u32 n, LzSeedDev sd,
               const u32* __restrict__ wstart, u32* __restrict__ cnt)
{
    const u32 j = blockIdx.x * LZ_TPB + threadIdx.x;
    if (j >= n) return;
    const u32 w0 = sk[j];
    if (w0 >> sd.weight) return;                                // no word at this position: cnt stays 0
    u32 c = 0;
    for (int p = 0; p < sd.nprobes; p++) { const u32 w = w0 ^ sd.probe_xor[p]; c += wstart[w + 1] - wstart[w]; }
    cnt[sv[j]] = c;
}

// bucket ownership (lzgpu_set_bucket_owner): only the hits whose hashed diagonal belongs to this process
// count; that needs the positions, so the lists are read here as well (front to back, like the fill)
__device__ __forceinline__ bool lz_owned(u32 pos1, u32 pos2, u32 n_owners, u32 owner)
{ return (((pos1 - pos2) & (LZ_DIAG_SIZE - 1)) % n_owners) == owner; }

__global__ void __launch_bounds__(LZ_TPB)
k_count_sorted_owned(const u32* __restrict__ sk, const u32* __restrict__ sv, u32 n, u32 lo, LzSeedDev sd,
                     const u32* __restrict__ wstart, const u32* __restrict__ wpos, u32* __restrict__ cnt,
                     u32 n_owners, u32 owner)
{
    const u32 j = blockIdx.x * LZ_TPB + threadIdx.x;
    if (j >= n) return;
    const u32 w0 = sk[j];
    if (w0 >> sd.weight) return;
    const u32 i = sv[j], pos2 = lo + i + 1;
    u32 c = 0;
    for (int p = 0; p < sd.nprobes; p++) {
        const u32 w = w0 ^ sd.probe_xor[p];
        for (u32 k = wstart[w]; k < wstart[w + 1]; k++) c += lz_owned(wpos[k], pos2, n_owners, owner) ? 1u : 0u;
    }
    cnt[i] = c;
}

int lzk_count_hits(LzCtx& c, const u8* qcode, u32 lo, u32 hi, u32* cnt, u32* pk, u32* iv, u32* sk, u32* sv, u64* valid_words_dev)
{
    const u32 n = hi - lo;
    LZ_HIP(hipMemsetAsync(cnt, 0, (size_t)n * 4, c.stream));
    c.timer.begin("k_pack_words", c.stream);
    hipLaunchKernelGGL(k_pack_words, dim3((n + LZ_TPB - 1) / LZ_TPB), dim3(LZ_TPB), 0, c.stream,
                       qcode, lo, hi, c.seed, pk, iv);
    c.timer.end(c.stream);
    LZ_HIP(hipGetLastError());
    size_t tmp = 0;
    const unsigned bits = (unsigned)c.seed.weight + 1u;
    LZ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, pk, sk, iv, sv, (size_t)n, 0u, bits, c.stream));
    int rc = c.sort_tmp.ensure(tmp);
    if (rc) return rc;
    c.timer.begin("rocprim_sort_words", c.stream);
    LZ_HIP(rocprim::radix_sort_pairs(c.sort_tmp.p, tmp, pk, sk, iv, sv, (size_t)n, 0u, bits, c.stream));
    c.timer.end(c.stream);
    hipLaunchKernelGGL(k_count_words, dim3(1), dim3(1), 0, c.stream, sk, n, 1u << c.seed.weight, valid_words_dev);
    c.timer.begin("k_count_hits", c.stream);
    if (c.n_owners > 1)
        hipLaunchKernelGGL(k_count_sorted_owned, dim3((n + LZ_TPB - 1) / LZ_TPB), dim3(LZ_TPB), 0, c.stream,
                           sk, sv, n, lo, c.seed, c.wstart.as<u32>(), c.wpos.as<u32>(), cnt, c.n_owners, c.owner);
    else
        hipLaunchKernelGGL(k_count_sorted, dim3((n + LZ_TPB - 1) / LZ_TPB), dim3(LZ_TPB), 0, c.stream,
                           sk, sv, n, c.seed, c.wstart.as<u32>(), cnt);
    c.timer.end(c.stream);
    LZ_HIP(hipGetLastError());
    return 0;
}

struct U32ToU64 { __host__ __device__ u64 operator()(u32 x) const { return (u64)x; } };

int lzk_scan_counts(LzCtx& c, const u32* cnt, u64* off, u32 n)
{
    auto in = rocprim::make_transform_iterator(cnt, U32ToU64());
    size_t tmp = 0;
    LZ_HIP(rocprim::exclusive_scan(nullptr, tmp, in, off, (u64)0, (size_t)n, rocprim::plus<u64>(), c.stream));
    int rc = c.scan_tmp.ensure(tmp);
    if (rc) return rc;
    c.timer.begin("rocprim_scan_counts", c.stream);
    LZ_HIP(rocprim::exclusive_scan(c.scan_tmp.p, tmp, in, off, (u64)0, (size_t)n, rocprim::plus<u64>(), c.stream));
    c.timer.end(c.stream);
    return 0;
}

// total number of hits and every stride-th prefix sum, written straight into pinned host memory:
// out[0] = total, out[1 + k] = off[k * stride]
__global__ void __launch_bounds__(LZ_TPB)
k_sample_offsets(const u64* __restrict__ off, const u32* __restrict__ cnt, u32 n, u32 stride, u32 ns, u64* __restrict__ out)
{
    const u32 k = blockIdx.x * LZ_TPB + threadIdx.x;
    if (k < ns) out[1 + k] = off[(size_t)k * stride];
    if (k == 0) out[0] = off[n - 1] + cnt[n - 1];
}

int lzk_sample_offsets(LzCtx& c, const u64* off, const u32* cnt, u32 n, u32 stride, u32 ns, u64* out)
{
    hipLaunchKernelGGL(k_sample_offsets, dim3((ns + LZ_TPB - 1) / LZ_TPB), dim3(LZ_TPB), 0, c.stream, off, cnt, n, stride, ns, out);
    LZ_HIP(hipGetLastError());
    return 0;
}

// B2 step 2: materialise the hits of query positions [i0,i1) in discovery order.
// A wave takes 64 entries of the word-sorted list, keeps those whose position lies in [i0,i1) (the hit
// arrays hold one chunk of positions at a time) and serves them four at a time: 16 lanes per position, one
// probe (exact word / transition flip) per lane.  Each lane reads its word's CSR range, a 16-lane prefix
// sum places the probes' lists back to back in probe order (= the reference's enumeration order within a
// position, src/seed_search.c:522-549), and the lists go to off[position] in the hit array.
#define LZ_FILL_GROUP 16
template <bool OWNED>
__global__ void __launch_bounds__(LZ_TPB)
k_fill_hits(u32 lo, u32 i0, u32 i1, LzSeedDev sd,
            const u32* __restrict__ wstart, const u32* __restrict__ wpos,
            const u32* __restrict__ sk, const u32* __restrict__ sv, u32 n, const u64* __restrict__ off,
            u64 base, u64* __restrict__ keys, u32 n_owners, u32 owner)
{
    const u32 lane = threadIdx.x & 63u, p = lane & (LZ_FILL_GROUP - 1), g = lane >> 4;
    const u32 j = (blockIdx.x * LZ_TPB + threadIdx.x);         // one sorted entry per lane
    u32 w_l = 0, i_l = 0; bool in = false;
    if (j < n) { w_l = sk[j]; i_l = sv[j]; in = !(w_l >> sd.weight) && i_l >= i0 && i_l < i1; }
    u64 todo = __ballot(in);
    while (todo) {                                              // wave-uniform
        // the next four entries of the wave, one per 16-lane group
        int src = -1;
        u64 m = todo;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int s_q = m ? (int)__ffsll((long long)m) - 1 : -1;
            if (m) m &= m - 1;
            if ((int)g == q) src = s_q;
        }
        todo = m;
        const bool have = src >= 0;
        const u32 packed = __shfl(w_l, have ? src : 0);
        const u32 i = __shfl(i_l, have ? src : 0);
        const u32 pos2 = lo + i + 1;
        u64* out = have ? keys + (off[i] - base) : keys;
        u32 carry = 0;
        for (int r = 0; r < sd.nprobes; r += LZ_FILL_GROUP) {   // uniform trip count
            u32 a = 0, len = 0, full = 0;
            if (have && r + (int)p < sd.nprobes) {
                const u32 w = packed ^ sd.probe_xor[r + p];
                a = wstart[w]; full = wstart[w + 1] - a; len = full;
                if (OWNED) { len = 0; for (u32 jj = 0; jj < full; jj++) len += lz_owned(wpos[a + jj], pos2, n_owners, owner) ? 1u : 0u; }
            }
            u32 incl = len;                                  // inclusive prefix over the 16-lane group
#pragma unroll
            for (int d = 1; d < LZ_FILL_GROUP; d <<= 1) {
                u32 v = __shfl_up(incl, d, LZ_FILL_GROUP);
                if ((int)p >= d) incl += v;
            }
            const u32 total = __shfl(incl, LZ_FILL_GROUP - 1, LZ_FILL_GROUP);
            u64* o = out + carry + (incl - len);
            if (OWNED) { for (u32 jj = 0; jj < full; jj++) { const u32 p1 = wpos[a + jj]; if (lz_owned(p1, pos2, n_owners, owner)) *o++ = lz_hit_key(p1, pos2); } }
            else for (u32 jj = 0; jj < len; jj++) o[jj] = lz_hit_key(wpos[a + jj], pos2);
            carry += total;
        }
    }
}

int lzk_fill_hits(LzCtx& c, u32 lo, u32 i0, u32 i1, const u32* sk, const u32* sv, u32 n, const u64* off, u64 base, u64* keys)
{
    if (n == 0 || i1 <= i0) return 0;
    c.timer.begin("k_fill_hits", c.stream);
    if (c.n_owners > 1)
        hipLaunchKernelGGL(k_fill_hits<true>, dim3((unsigned)(((u64)n + LZ_TPB - 1) / LZ_TPB)), dim3(LZ_TPB), 0, c.stream,
                           lo, i0, i1, c.seed, c.wstart.as<u32>(), c.wpos.as<u32>(), sk, sv, n, off, base, keys, c.n_owners, c.owner);
    else
        hipLaunchKernelGGL(k_fill_hits<false>, dim3((unsigned)(((u64)n + LZ_TPB - 1) / LZ_TPB)), dim3(LZ_TPB), 0, c.stream,
                           lo, i0, i1, c.seed, c.wstart.as<u32>(), c.wpos.as<u32>(), sk, sv, n, off, base, keys, 1u, 0u);
    c.timer.end(c.stream);
    LZ_HIP(hipGetLastError());
    return 0;
}

// B2 step 3 (phase A): one thread per raw hit, any order -- capped X-drop scans, 4-byte summary.
// The kernel is VALU-bound (PMC: 92 % VALU, 79 % texture-address busy), and the scan lengths inside a wave
// differ: after the blocks every hit gets in lz_probe_head, ~20 % of the left and ~10 % of the right scans
// are still going, and a wave that serves them in place issues every further block for all 64 lanes.  The
// unfinished scans of the 256 hits of a block are therefore queued in LDS as independent tasks (left ones
// from the front, right ones from the back) and served densely: one lane per task, to completion.
#ifndef LZ_PROBE_TPB
#define LZ_PROBE_TPB 256              // hits (threads) per block of k_probe_hits: the pool the task queue packs
#endif
struct LzScanTask { u32 s; s32 run, best, stop, diag; u32 side; };
__global__ void __launch_bounds__(LZ_PROBE_TPB)
k_probe_hits(LzExtendParams P, const u64* __restrict__ keys, u64 n, const s32* __restrict__ score_tab_g,
             u32* __restrict__ summ)
{
    __shared__ s32 tab[LZ_NCLASS * LZ_NCLASS];
    __shared__ s32 tab8[64];
    __shared__ LzScanTask task[2 * LZ_PROBE_TPB];
    __shared__ u32 n_left, n_right;
    for (int k = threadIdx.x; k < LZ_NCLASS * LZ_NCLASS; k += LZ_PROBE_TPB) tab[k] = score_tab_g[k];
    if (threadIdx.x < 64) tab8[threadIdx.x] = score_tab_g[(threadIdx.x >> 3) * LZ_NCLASS + (threadIdx.x & 7)];
    if (threadIdx.x == 0) { n_left = 0; n_right = 0; }
    __syncthreads();
    const bool fast = P.cls8 != 0;
    const u64 i = (u64)blockIdx.x * LZ_PROBE_TPB + threadIdx.x;
    LzProbeSt st;
    st.alive_l = st.alive_r = false;
    if (i < n) lz_probe_head(P, tab, tab8, fast, keys[i], st);
    int slot_l = -1, slot_r = -1;
    if (st.alive_l) {
        slot_l = (int)atomicAdd(&n_left, 1u);
        task[slot_l] = { st.sl, st.runl, st.bestl, st.stopl, st.diag, 0u };
    }
    if (st.alive_r) {
        slot_r = 2 * LZ_PROBE_TPB - 1 - (int)atomicAdd(&n_right, 1u);
        task[slot_r] = { st.sr, st.runr, st.bestr, st.stopr, st.diag, 1u };
    }
    __syncthreads();
    const u32 nl = n_left, nr = n_right;
    for (u32 k = threadIdx.x; k < nl + nr; k += LZ_PROBE_TPB) {
        LzScanTask& q = task[k < nl ? k : 2 * LZ_PROBE_TPB - 1 - (k - nl)];
        u32 s = q.s; s32 run = q.run, best = q.best;
        bool alive;
        if (q.side) alive = lz_scan_continue<true>(P, tab, tab8, fast, q.diag, q.stop, s, run, best, LZ_PROBE_CAP / 16 - LZ_PROBE_AHEAD_R);
        else        alive = lz_scan_continue<false>(P, tab, tab8, fast, q.diag, q.stop, s, run, best, LZ_PROBE_CAP / 16 - LZ_PROBE_AHEAD_L);
        q.s = s; q.best = best; q.run = alive ? 1 : 0;          // the result goes back through the task's slot
    }
    __syncthreads();
    if (slot_l >= 0) { st.sl = task[slot_l].s; st.bestl = task[slot_l].best; st.alive_l = task[slot_l].run != 0; }
    if (slot_r >= 0) { st.sr = task[slot_r].s; st.bestr = task[slot_r].best; st.alive_r = task[slot_r].run != 0; }
    if (i < n) summ[i] = lz_probe_summary(P, st);
}

int lzk_probe_hits(LzCtx& c, const LzExtendParams& P, const u64* keys, u64 n, const s32* score_tab, u32* summ)
{
    if (n == 0) return 0;
    c.timer.begin("k_probe_hits", c.stream);
    hipLaunchKernelGGL(k_probe_hits, dim3((unsigned)((n + LZ_PROBE_TPB - 1) / LZ_PROBE_TPB)), dim3(LZ_PROBE_TPB), 0, c.stream,
                       P, keys, n, score_tab, summ);
    c.timer.end(c.stream);
    LZ_HIP(hipGetLastError());
    return 0;
}

// B2 step 4: stable partition of (key, summary) by hashedDiag = key bits 32..47
int lzk_sort_hits(LzCtx& c, u64* keys_in, u64* keys_out, u32* summ_in, u32* summ_out, u64 n)
{
    size_t tmp = 0;
    LZ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, keys_in, keys_out, summ_in, summ_out, (size_t)n,
                                     32u, 32u + LZ_DIAG_BITS, c.stream));
    int rc = c.sort_tmp.ensure(tmp);
    if (rc) return rc;
    c.timer.begin("rocprim_sort_hits", c.stream);
    LZ_HIP(rocprim::radix_sort_pairs(c.sort_tmp.p, tmp, keys_in, keys_out, summ_in, summ_out, (size_t)n,
                                     32u, 32u + LZ_DIAG_BITS, c.stream));
    c.timer.end(c.stream);
    return 0;
}

// B2 step 5 (phase B): one lane per hash bucket
#define LZ_EXT_TPB 64
__global__ void __launch_bounds__(LZ_EXT_TPB)
k_extend(LzExtendParams P, const u64* __restrict__ keys, const u32* __restrict__ summ, u32 n,
         u32* __restrict__ diag_end, const s32* __restrict__ score_tab_g,
         LzHspRec* __restrict__ out, u32* __restrict__ out_count, u32 out_cap, u64* __restrict__ counters)
{
    __shared__ s32 tab[LZ_NCLASS * LZ_NCLASS];
    for (int k = threadIdx.x; k < LZ_NCLASS * LZ_NCLASS; k += LZ_EXT_TPB) tab[k] = score_tab_g[k];
    __syncthreads();
    const u32 h = blockIdx.x * LZ_EXT_TPB + threadIdx.x;
    // the lane's bucket = the run of keys whose bits 32..47 equal h (the array is partitioned by them):
    // two binary searches, 2 x 28 dependent loads once per launch instead of a pass over all keys
    u32 i0 = 0, i1 = 0;
    {
        u32 a = 0, b = n;
        while (a < b) { const u32 m = a + ((b - a) >> 1); if ((u32)((keys[m] >> 32) & (LZ_DIAG_SIZE - 1)) < h) a = m + 1; else b = m; }
        i0 = a; b = n;
        while (a < b) { const u32 m = a + ((b - a) >> 1); if ((u32)((keys[m] >> 32) & (LZ_DIAG_SIZE - 1)) <= h) a = m + 1; else b = m; }
        i1 = a;
    }
    u64 n_ext = 0, n_bp = 0;
    if (i0 < i1) {
        u32 d = lz_extend_bucket(P, tab, keys, summ, i0, i1, diag_end[h], n_ext, n_bp,
            [&](const LzHspRec& r) {
                u32 slot = atomicAdd(out_count, 1u);
                if (slot < out_cap) out[slot] = r;
            });
        diag_end[h] = d;
    }
    // wave-level reduction of the work counters
    for (int o = 32; o > 0; o >>= 1) { n_ext += __shfl_down(n_ext, o); n_bp += __shfl_down(n_bp, o); }
    if ((threadIdx.x & 63) == 0) {
        if (n_ext) atomicAdd((unsigned long long*)&counters[0], (unsigned long long)n_ext);
        if (n_bp)  atomicAdd((unsigned long long*)&counters[1], (unsigned long long)n_bp);
    }
}

int lzk_extend(LzCtx& c, const LzExtendParams& P, const u64* keys, const u32* summ, u32 n, u32* diag_end,
               const s32* score_tab, LzHspRec* out, u32* out_count, u32 out_cap, u64* counters, hipStream_t s)
{
    c.timer.begin("k_extend", s);
    hipLaunchKernelGGL(k_extend, dim3(LZ_DIAG_SIZE / LZ_EXT_TPB), dim3(LZ_EXT_TPB), 0, s,
                       P, keys, summ, n, diag_end, score_tab, out, out_count, out_cap, counters);
    c.timer.end(s);
    LZ_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// entropy inputs of the candidate HSPs: per candidate, how many aligned positions carry the
// same byte 'A' / 'C' / 'G' / 'T' in target and query (compute_entropy counts exactly these,
// src/dna_utilities.c:2899-2912).  One wave per candidate, lanes stride over the segment.
__global__ void __launch_bounds__(LZ_TPB)
k_hsp_match_counts(const LzHspRec* __restrict__ recs, const u32* __restrict__ n_rec, u32 cap,
                   const u8* __restrict__ traw, const u8* __restrict__ qraw,
                   const u8* __restrict__ tcode, const u8* __restrict__ qcode, LzSeedDev sd, u32* __restrict__ counts)
{
    const u32 wave = (blockIdx.x * LZ_TPB + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    u32 n = *n_rec; if (n > cap) n = cap;
    if (wave >= n) return;
    const LzHspRec r = recs[wave];
    const s32 diag = (s32)r.seed_pos1 - (s32)r.seed_pos2;
    const u32 s1 = r.end1 - r.length, s2 = (u32)((s32)s1 - diag);
    u32 cA = 0, cC = 0, cG = 0, cT = 0;
    for (u32 k = lane; k < r.length; k += 64) {
        const u32 a = traw[s1 + k], b = qraw[s2 + k];
        if (a == b) { cA += (a == 'A'); cC += (a == 'C'); cG += (a == 'G'); cT += (a == 'T'); }
    }
    for (int o = 32; o > 0; o >>= 1) { cA += __shfl_down(cA, o); cC += __shfl_down(cC, o); cG += __shfl_down(cG, o); cT += __shfl_down(cT, o); }
    if (lane == 0) {
        // which probe produced the seed hit (its discovery rank inside a query position): the XOR of the
        // two packed words is one of the probe masks (src/seed_search.c:522-549)
        u32 pt = 0, pq = 0, probe = 0xFFFFFFFFu;
        if (lz_window_word(tcode, r.seed_pos1, sd, pt) && lz_window_word(qcode, r.seed_pos2, sd, pq)) {
            const u32 x = pt ^ pq;
            for (int p = 0; p < sd.nprobes; p++) if (sd.probe_xor[p] == x) { probe = (u32)p; break; }
        }
        u32* o = counts + 5 * (size_t)wave;
        o[0] = cA; o[1] = cC; o[2] = cG; o[3] = cT; o[4] = probe;
    }
}

int lzk_hsp_match_counts(LzCtx& c, const LzHspRec* recs, const u32* n_rec_dev, u32 cap, u32 launch_for,
                         const u8* traw, const u8* qraw, const u8* tcode, const u8* qcode, u32* counts, hipStream_t s)
{
    if (launch_for == 0) return 0;
    const u64 threads = (u64)launch_for * 64;
    c.timer.begin("k_hsp_match_counts", s);
    hipLaunchKernelGGL(k_hsp_match_counts, dim3((unsigned)((threads + LZ_TPB - 1) / LZ_TPB)), dim3(LZ_TPB), 0, s,
                       recs, n_rec_dev, cap, traw, qraw, tcode, qcode, c.seed, counts);
    c.timer.end(s);
    LZ_HIP(hipGetLastError());
    return 0;
}


