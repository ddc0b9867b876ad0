// seed_kernels.hip -- gfx950 kernels of the seed stage: position-table build (B1), raw-hit
// enumeration, bucket ordering and the bucket-serial X-drop extender (B2).
//
// Decomposition: the only cross-hit state of the reference's HSP search is
// diagEnd[hashedDiag] (src/seed_search.c:1081-1126, 2612-2616, 2785-2789), so the exact
// parallel form is 65,536 independent, order-preserving streams.  Hits are enumerated in the
// reference's order (count -> scan -> fill gives every hit its discovery index; the table itself
// is probed in seed-word order), scanned independently of the hash (phase A, k_scan_hits: four bases per
// look-up in an LDS table on 2-bit codes, lz_lut.hpp), stably partitioned by the high 8 hash bits
// (k_partition), and each partition is then dealt out to its 256 buckets inside LDS, every bucket walked by
// one lane with diagEnd[h] in a register (phase B, k_settle2).
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include "lz_ctx.hpp"
#include "lz_lut.hpp"
#include "lz_coop.hpp"

#define LZ_TPB 256

// ------------------------------------------------------------------------------------------
// byte -> code translation (one pass per sequence; the table folds charToBits and the score
// class of the byte, see lz_common.hpp)
// The per-hit streams of the stage -- keys, partition bytes, summaries, records: written once by one kernel, read once by the
// next, 8-21 bytes per hit and 4-62 G hits per step -- are stored and loaded NON-TEMPORALLY, so that they pass through L2 and the
// 256 MiB memory-side cache without pushing out what IS re-read: the position table's lists, the target's and the query's 2-bit
// windows.  Measured (same box, A/B against -DLZ_NO_NT): at 200 Mbp x 200 Mbp fill 360 -> 318, scans 1260 -> 1253, tasks 87 -> 80 ms
// per step (-2.1 % of the step); on the 50 Mbp pair -1 ms.  k_partition keeps plain accesses: with non-temporal ones it ran
// 10-15 % slower (its 512-byte runs per partition want to merge in L2).
#if defined(LZ_NO_NT)
#define LZ_NT_LD(p_) (*(p_))
#define LZ_NT_ST(v_, p_) (*(p_) = (v_))
#else
#define LZ_NT_LD(p_) __builtin_nontemporal_load(p_)
#define LZ_NT_ST(v_, p_) __builtin_nontemporal_store((v_), (p_))
#endif
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

int lzk_encode(LzCtx& c, const u8* raw, u8* code, u32 len, const u8* cls256_dev, hipStream_t st, KernelTimer* timer)
{
    if (len == 0) return 0;
    if (!st) st = c.stream;
    if (!timer) timer = &c.timer;
    u32 nvec = (len + 15u) >> 4;
    u32 blocks = (nvec + LZ_TPB - 1) / LZ_TPB; if (blocks > 4096) blocks = 4096;
    timer->begin("k_encode", st);
    hipLaunchKernelGGL(k_encode, dim3(blocks), dim3(LZ_TPB), 0, st, raw, code, len, cls256_dev);
    timer->end(st);
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
// The sort key of a query position is (block of the position) << kbits | word, kbits = weight + 1: the list is word-sorted INSIDE blocks of
// 2^bshift consecutive positions (at most 16 blocks).  What the fill kernel writes goes to off[position]: with one word order over the
// whole query a wave's 64 positions scatter their runs of keys over the whole key array of the chunk (15.5 GB for a 50 Mbp strand),
// three new pages per position -- and k_fill_hits2 took 18 or 28 ms from one process to the next on the same box, depending on what
// physical pages its buffers got.  Inside a block the destinations stay within 1/16 of the arrays, the probes still move front to back
// through the table (3-4 words between neighbouring entries: the lines of wstart[] are still shared), and a chunk of query positions is a
// contiguous range of blocks, hence of the sorted list: its launch covers that range only (a 200 Mbp strand has 29 chunks; every one of
// their launches used to read the whole list to find its 1/29).
__global__ void __launch_bounds__(LZ_TPB)
k_pack_words(const u8* __restrict__ qcode, u32 lo, u32 hi, LzSeedDev sd, u32* __restrict__ pk, u32* __restrict__ iv, u32 bshift, u32 kbits)
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
        pk[i] = ((i >> bshift) << kbits) | (valid ? lz_apply_seed(sd, w) : (1u << sd.weight));   // "no word" sorts after every real word of its block
        iv[i] = i;
    }
}
// where the blocks begin in the sorted list: bs[b] = first entry whose key's block is >= b, b in [0, nblk]; and "words in seq 2" (the
// reference's counter): inside a block the entries without a word (key's word field == none) sort behind those with one
__global__ void k_block_starts(const u32* __restrict__ sk, u32 n, u32 kbits, u32 none, u32 nblk, u64* __restrict__ bs, u64* __restrict__ n_words)
{
    const u32 b = threadIdx.x;
    if (b > nblk) return;
    auto first_at_least = [&](u64 key) { u32 lo = 0, hi = n; while (lo < hi) { const u32 m = lo + ((hi - lo) >> 1); if ((u64)sk[m] < key) lo = m + 1; else hi = m; } return lo; };
    const u32 start = first_at_least((u64)b << kbits);
    bs[b] = start;
    if (b < nblk) atomicAdd((unsigned long long*)n_words, (unsigned long long)(first_at_least(((u64)b << kbits) | none) - start));
}

__global__ void __launch_bounds__(LZ_TPB)
k_count_sorted(const u32* __restrict__ sk, const u32* __restrict__ sv, u32 n, LzSeedDev sd,
               const u32* __restrict__ wstart, u32* __restrict__ cnt)
{
    const u32 j = blockIdx.x * LZ_TPB + threadIdx.x;
    if (j >= n) return;
    const u32 w0 = sk[j] & ((2u << sd.weight) - 1u);            // (the key's low weight + 1 bits: the word, or "no word")
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
    const u32 w0 = sk[j] & ((2u << sd.weight) - 1u);
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
    // blocks of positions (k_pack_words): at most 16, fewer for the heaviest seeds (the key is 32 bits)
    const u32 kbits = (u32)c.seed.weight + 1u;
    // ... and only for a search that will need several chunks (measured: with one chunk per strand the blocks buy nothing -- the fill
    // kernel's 18-or-28 ms do not come from the range of its scatter -- and cost k_count_hits 0.8 ms of shared wstart[] lines; at 200 Mbp,
    // 29 chunks per strand, the launches' ranges take the fill from 292 to 232 ms per step): twice as many blocks as expected chunks
    static const int force_bits = []() { const char* e = getenv("LZGPU_BLOCK_BITS"); const int v = e ? atoi(e) : -1; return v > 7 ? 7 : v; }();   // A/B
    const double est_hits = (double)n * (double)c.num_words * (double)c.seed.nprobes / (double)(1ull << c.seed.weight);
    u32 want_bits = 0;
    while (want_bits < 5 && est_hits > (double)c.hit_capacity * (double)(1u << want_bits) * 0.5) want_bits++;
    if (est_hits <= (double)c.hit_capacity) want_bits = 0;
    if (force_bits >= 0) want_bits = (u32)force_bits;
    u32 bbits = std::min<u32>(want_bits, 32u - kbits);
    u32 bshift = 0;
    const u64 nm1 = n ? (u64)n - 1 : 0;
    while (bshift < 32 && (nm1 >> bshift) >= (1ull << bbits)) bshift++;
    c.blk_shift = bshift; c.blk_count = (u32)((nm1 >> bshift) + 1);
    c.timer.begin("k_pack_words", c.stream);
    hipLaunchKernelGGL(k_pack_words, dim3((n + LZ_TPB - 1) / LZ_TPB), dim3(LZ_TPB), 0, c.stream,
                       qcode, lo, hi, c.seed, pk, iv, bshift, kbits);
    c.timer.end(c.stream);
    LZ_HIP(hipGetLastError());
    size_t tmp = 0;
    const unsigned bits = kbits + bbits;
    LZ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, pk, sk, iv, sv, (size_t)n, 0u, bits, c.stream));
    int rc = c.sort_tmp.ensure(tmp);
    if (rc) return rc;
    c.timer.begin("rocprim_sort_words", c.stream);
    LZ_HIP(rocprim::radix_sort_pairs(c.sort_tmp.p, tmp, pk, sk, iv, sv, (size_t)n, 0u, bits, c.stream));
    c.timer.end(c.stream);
    if ((rc = c.blk_start.ensure(130 * 8))) return rc;
    hipLaunchKernelGGL(k_block_starts, dim3(1), dim3(192), 0, c.stream, sk, n, kbits, 1u << c.seed.weight, c.blk_count, c.blk_start.as<u64>(), valid_words_dev);
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
#define LZ_KEY_BIN(k)  ((u32)((k) >> 40) & 0xFFu)    // the partition of a hit: bits 8..15 of hashedDiag
#define LZ_FILL_GROUP 16
template <bool OWNED>
__global__ void __launch_bounds__(LZ_TPB)
k_fill_hits(u32 lo, u32 i0, u32 i1, LzSeedDev sd,
            const u32* __restrict__ wstart, const u32* __restrict__ wpos,
            const u32* __restrict__ sk, const u32* __restrict__ sv, u32 n, const u64* __restrict__ off,
            u64 base, u64* __restrict__ keys, u32 n_owners, u32 owner)
{
    const u32 lane = threadIdx.x & 63u, p = lane & (LZ_FILL_GROUP - 1), g = lane >> 4;
    const u32 j = (blockIdx.x * LZ_TPB + threadIdx.x);         // one sorted entry per lane (sk / sv: the chunk's range of the list)
    u32 w_l = 0, i_l = 0; bool in = false;
    if (j < n) { w_l = sk[j] & ((2u << sd.weight) - 1u); i_l = sv[j]; in = !(w_l >> sd.weight) && i_l >= i0 && i_l < i1; }
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
            if (OWNED) {
                u64* o = out + carry + (incl - len);
                for (u32 jj = 0; jj < full; jj++) { const u32 p1 = wpos[a + jj]; if (lz_owned(p1, pos2, n_owners, owner)) *o++ = lz_hit_key(p1, pos2); }
            } else {
                // The (up to) 64 lists of the wave -- 4 positions x 16 probes -- laid end to end: lane t takes hit t,
                // t + 64, ... of that run, finds the list holding it (binary search over the lanes' running totals)
                // and writes it to its place: the stores of a wave are consecutive keys (one run per position),
                // not one 8-byte store per list and step.
                u32 gtot = total;                             // running totals over the whole wave: + the groups below
                const u32 t0 = __shfl(total, 0), t1 = __shfl(total, 16), t2 = __shfl(total, 32), t3 = __shfl(total, 48);
                const u32 gbase = g == 0 ? 0u : g == 1 ? t0 : g == 2 ? t0 + t1 : t0 + t1 + t2;      // hits of the groups below this lane's
                gtot = t0 + t1 + t2 + t3;
                const u32 winc = gbase + incl;                // inclusive running total at this lane
                const s64 obase = have ? (s64)(out - keys) + (s64)carry - (s64)gbase : 0;           // key index of the group's run, minus gbase
                for (u32 tb = 0; tb < gtot; tb += 64u) {      // wave-uniform trips: every lane takes part in the shuffles
                    const u32 t = tb + lane;
                    u32 lo_l = 0, hi_l = 63;                  // first lane whose inclusive total exceeds t
#pragma unroll
                    for (int it = 0; it < 6; it++) {
                        const u32 mid = (lo_l + hi_l) >> 1;
                        const u32 v = (u32)__shfl((int)winc, (int)mid);
                        if (v > t) hi_l = mid; else lo_l = mid + 1;
                    }
                    const u32 L = lo_l;
                    const u32 l_inc = (u32)__shfl((int)winc, (int)L), l_len = (u32)__shfl((int)len, (int)L), l_a = (u32)__shfl((int)a, (int)L);
                    const u32 l_pos2 = (u32)__shfl((int)pos2, (int)L);
                    const u32 ob_lo = (u32)__shfl((int)(u32)obase, (int)L), ob_hi = (u32)__shfl((int)(u32)((u64)obase >> 32), (int)L);
                    const u32 jj = t - (l_inc - l_len);
                    const s64 ob = (s64)(((u64)ob_hi << 32) | ob_lo);
                    if (t < gtot) keys[ob + (s64)t] = lz_hit_key(wpos[l_a + jj], l_pos2);
                }
            }
            carry += total;
        }
    }
}


// ---- k_fill_hits2 (round 4): the same hits at the same places, the lists found through LDS instead of shuffles.
// k_fill_hits lays the 64 lists of four positions end to end and lets every lane search the running totals for the list
// holding its hit: 13 shuffles and two prefix sums per 64 hits, 150 VALU instructions per 64-hit trip for 16 bytes of
// useful traffic per hit.  Here a wave takes its 64 sorted entries at once:
//   1. every lane reads the CSR bounds of its position's probes (13 independent pairs of loads in flight per lane)
//      and keeps the lengths; an exclusive wave scan of the lanes' totals places the positions' runs end to end
//      (entry-major, probe order inside an entry = the reference's enumeration order inside a position);
//   2. every list marks the cell of its first hit in own[] (its id); a prefix maximum over own[] -- ids grow along the
//      concatenation -- leaves every cell with the id of the list that holds it: 16 marks and two passes over 128
//      bytes per lane instead of a search per hit;
//   3. lane t takes hits t, t + 64, ...: own[t] -> (list source - list start), (run's place in the key array - run
//      start, pos2) from two small LDS tables -> one gather from wpos[], one key stored.  The
//      stores of a wave are runs of consecutive keys (one run per position), as before.
// A wave's LDS traffic is its own (no barrier: LDS operations of one wave execute in program order); concatenations
// longer than LZ_F2_CAP hits (repeats) are taken in pieces; seeds with more than 16 probes in groups of 16.
#ifndef LZ_F2_CAP
#define LZ_F2_CAP   2560                             // hits per piece of a wave's concatenation: a multiple of 512; own[] 5 KiB + 4.5 KiB of tables per wave = four workgroups per CU (the 64 positions of a wave have 2.5 k hits on the 50 Mbp pair)
#endif
#define LZ_F2_WORDS (LZ_F2_CAP / 2 / 64)             // 32-bit words of own[] per lane
static_assert(LZ_F2_CAP % 512 == 0, "whole 16-byte groups of own[] per lane");
struct LzFill2Wave {
    alignas(16) u32 own32[LZ_F2_CAP / 2];            // own[]: u16 list ids (1 + lane * 16 + probe), 0 = no list starts here
    u32 lsrc[64 * LZ_FILL_GROUP];                    // list -> (first wpos index - start of the list in the concatenation)
    uint2 ent[64];                                   // entry -> (place of its run in the key array - start of the run, pos2)
};
template <int CTRL, int ROWM, int BANKM>
__device__ __forceinline__ u32 lz_dpp_u32(u32 ident, u32 src) { return (u32)__builtin_amdgcn_update_dpp((int)ident, (int)src, CTRL, ROWM, BANKM, false); }
// inclusive wave scans (the classic row_shr 1,2,3 / 4 / 8 / row_bcast 15 / 31 sequence; unreached lanes get the identity)
#define LZ_WAVE_SCAN_U32(v, x, OP)                                     \
    v = OP(lz_dpp_u32<0x111, 0xf, 0xf>(0u, x), v);                     \
    v = OP(lz_dpp_u32<0x112, 0xf, 0xf>(0u, x), v);                     \
    v = OP(lz_dpp_u32<0x113, 0xf, 0xf>(0u, x), v);                     \
    v = OP(lz_dpp_u32<0x114, 0xf, 0xe>(0u, v), v);                     \
    v = OP(lz_dpp_u32<0x118, 0xf, 0xc>(0u, v), v);                     \
    v = OP(lz_dpp_u32<0x142, 0xa, 0xf>(0u, v), v);                     \
    v = OP(lz_dpp_u32<0x143, 0xc, 0xf>(0u, v), v);
__device__ __forceinline__ u32 lz_uadd(u32 a, u32 b) { return a + b; }
__device__ __forceinline__ u32 lz_umax(u32 a, u32 b) { return a > b ? a : b; }
__global__ void __launch_bounds__(LZ_TPB)
k_fill_hits2(u32 lo, u32 i0, u32 i1, LzSeedDev sd,
             const u32* __restrict__ wstart, const u32* __restrict__ wpos,
             const u32* __restrict__ sk, const u32* __restrict__ sv, u32 n, const u64* __restrict__ off,
             u64 base, u64* __restrict__ keys)
{
    __shared__ LzFill2Wave shw[LZ_TPB / 64];
    LzFill2Wave& sh = shw[threadIdx.x >> 6];
    unsigned short* const own = reinterpret_cast<unsigned short*>(sh.own32);
    const u32 lane = threadIdx.x & 63u;
    const u32 j = blockIdx.x * LZ_TPB + threadIdx.x;           // one sorted entry per lane (sk / sv: the chunk's range of the list)
    u32 w_l = 0, i_l = 0; bool in = false;
    if (j < n) { w_l = sk[j] & ((2u << sd.weight) - 1u); i_l = sv[j]; in = !(w_l >> sd.weight) && i_l >= i0 && i_l < i1; }
    if (!__ballot(in)) return;                                  // (wave-uniform: nothing of this chunk among the wave's entries)
    const u32 dbase = in ? (u32)(off[i_l] - base) : 0u;         // (hit indices inside a chunk are 32-bit)
    const u32 pos2 = lo + i_l + 1u;
    u32 carry = 0;                                              // hits of the lane's position in the probe groups already done
    for (int r = 0; r < sd.nprobes; r += LZ_FILL_GROUP) {       // uniform trip count
        // ---- 1. bounds of the lane's lists
        // (every load of the group is issued before the first is used: a lane without a probe reads word 0's bounds and
        // drops them)
        u32 la[LZ_FILL_GROUP], len[LZ_FILL_GROUP]; u32 total = 0;
#pragma unroll
        for (int p = 0; p < LZ_FILL_GROUP; p++) {
            const bool on = in && r + p < sd.nprobes;
            const u32 w = on ? (w_l ^ sd.probe_xor[(r + p) & (LZ_MAX_PROBES - 1)]) : 0u;
            la[p] = wstart[w]; len[p] = wstart[w + 1];
        }
#pragma unroll
        for (int p = 0; p < LZ_FILL_GROUP; p++) {
            const bool on = in && r + p < sd.nprobes;
            len[p] = on ? len[p] - la[p] : 0u; total += len[p];
        }
        u32 inc = total;
        LZ_WAVE_SCAN_U32(inc, total, lz_uadd)
        const u32 E = inc - total;                              // start of the lane's run in the wave's concatenation
        const u32 T = (u32)__builtin_amdgcn_readlane((int)inc, 63);
        {
            u32 S = E;
#pragma unroll
            for (int p = 0; p < LZ_FILL_GROUP; p++) { sh.lsrc[lane * LZ_FILL_GROUP + p] = la[p] - S; S += len[p]; }
        }
        sh.ent[lane] = make_uint2(dbase + carry - E, pos2);
        // ---- 2. + 3., a piece of LZ_F2_CAP hits at a time
        for (u32 cs = 0; cs < T; cs += LZ_F2_CAP) {              // uniform
            const u32 tc = (T - cs < (u32)LZ_F2_CAP) ? T - cs : (u32)LZ_F2_CAP;
            uint4* const mine = reinterpret_cast<uint4*>(sh.own32 + lane * LZ_F2_WORDS);
#pragma unroll
            for (int k = 0; k < LZ_F2_WORDS / 4; k++) mine[k] = make_uint4(0u, 0u, 0u, 0u);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            {
                u32 S = E;
#pragma unroll
                for (int p = 0; p < LZ_FILL_GROUP; p++) {
                    const u32 l = len[p];
                    if (l && S < cs + tc && S + l > cs) own[(S > cs ? S : cs) - cs] = (unsigned short)(1u + lane * LZ_FILL_GROUP + (u32)p);
                    S += l;
                }
            }
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            // prefix maximum over own[]: the lane's 2 * LZ_F2_WORDS cells, then the lanes' last values, then the cells again
            u32 wv[LZ_F2_WORDS];
#pragma unroll
            for (int k = 0; k < LZ_F2_WORDS / 4; k++) { const uint4 v = mine[k]; wv[4 * k] = v.x; wv[4 * k + 1] = v.y; wv[4 * k + 2] = v.z; wv[4 * k + 3] = v.w; }
            u32 segmax = 0;
#pragma unroll
            for (int k = 0; k < LZ_F2_WORDS; k++) { const u32 m2 = lz_umax(wv[k] & 0xFFFFu, wv[k] >> 16); segmax = lz_umax(segmax, m2); }
            u32 sinc = segmax;
            LZ_WAVE_SCAN_U32(sinc, segmax, lz_umax)
            u32 run = lz_dpp_u32<0x138, 0xf, 0xf>(0u, sinc);    // wave_shr:1: the maximum of the lanes below
#pragma unroll
            for (int k = 0; k < LZ_F2_WORDS; k++) {
                const u32 a0 = lz_umax(wv[k] & 0xFFFFu, run), a1 = lz_umax(wv[k] >> 16, a0);
                wv[k] = a0 | (a1 << 16); run = a1;
            }
#pragma unroll
            for (int k = 0; k < LZ_F2_WORDS / 4; k++) mine[k] = make_uint4(wv[4 * k], wv[4 * k + 1], wv[4 * k + 2], wv[4 * k + 3]);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            // the hits of the piece, 64 per trip, consecutive lanes = consecutive hits; four trips' gathers in flight
            // (a lane beyond the end of the piece reads the last hit again and stores nothing: no branch around the loads)
            for (u32 tb = 0; tb < tc; tb += 256u) {
                u32 p1[4], p2[4], dst[4]; bool ok[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const u32 t0 = tb + 64u * (u32)k + lane;
                    ok[k] = t0 < tc;
                    const u32 t = ok[k] ? t0 : tc - 1u;
                    const u32 list = (u32)own[t] - 1u;
                    const u32 ls = sh.lsrc[list];
                    const uint2 en = sh.ent[list / LZ_FILL_GROUP];
                    const u32 tabs = cs + t;
                    p1[k] = wpos[ls + tabs]; p2[k] = en.y; dst[k] = en.x + tabs;
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (ok[k]) LZ_NT_ST(lz_hit_key(p1[k], p2[k]), keys + dst[k]);
            }
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
        }
        carry += total;
    }
}

int lzk_fill_hits(LzCtx& c, u32 lo, u32 i0, u32 i1, const u32* sk, const u32* sv, u32 n, const u64* off, u64 base, u64* keys, hipStream_t st)
{
    if (n == 0 || i1 <= i0) return 0;
    // the chunk's positions [i0, i1) lie in blocks i0 >> shift .. (i1 - 1) >> shift: a contiguous range of the sorted list
    if (!c.blk_start_host.empty()) {
        const u32 b0 = i0 >> c.blk_shift, b1 = (i1 - 1) >> c.blk_shift;
        const u64 j0 = c.blk_start_host[b0], j1 = c.blk_start_host[std::min<u32>(b1 + 1, c.blk_count)];
        sk += j0; sv += j0; n = (u32)(j1 - j0);
        if (n == 0) return 0;
    }
    c.timer.begin("k_fill_hits", st);
    if (c.n_owners > 1)
        hipLaunchKernelGGL(k_fill_hits<true>, dim3((unsigned)(((u64)n + LZ_TPB - 1) / LZ_TPB)), dim3(LZ_TPB), 0, st,
                           lo, i0, i1, c.seed, c.wstart.as<u32>(), c.wpos.as<u32>(), sk, sv, n, off, base, keys, c.n_owners, c.owner);
    else {
        static const bool old_fill = getenv("LZGPU_FILL_SHUFFLE") != nullptr;   // A/B aid: round 2's shuffle-search fill
        if (old_fill)
            hipLaunchKernelGGL(k_fill_hits<false>, dim3((unsigned)(((u64)n + LZ_TPB - 1) / LZ_TPB)), dim3(LZ_TPB), 0, st,
                               lo, i0, i1, c.seed, c.wstart.as<u32>(), c.wpos.as<u32>(), sk, sv, n, off, base, keys, 1u, 0u);
        else
            hipLaunchKernelGGL(k_fill_hits2, dim3((unsigned)(((u64)n + LZ_TPB - 1) / LZ_TPB)), dim3(LZ_TPB), 0, st,
                               lo, i0, i1, c.seed, c.wstart.as<u32>(), c.wpos.as<u32>(), sk, sv, n, off, base, keys);
    }
    c.timer.end(st);
    LZ_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// 2-bit codes + special masks for phase A (lz_lut.hpp), from the code bytes: one thread per mask byte (8 bases).
// Base i of the sequence is bit (i + LZ_PAD2); bases outside [-LZ_SEQ_PAD, len + LZ_SEQ_PAD) are padding: special.
__global__ void __launch_bounds__(LZ_TPB)
k_pack2(const u8* __restrict__ code /*base 0*/, u32 len, u8* __restrict__ two, u8* __restrict__ spc, u32 nmask)
{
    const u32 j = blockIdx.x * LZ_TPB + threadIdx.x;
    if (j >= nmask) return;
    u32 bits = 0, m = 0;
    const s64 b0 = (s64)j * 8 - LZ_PAD2;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const s64 i = b0 + k;
        const u32 c = (i >= -(s64)LZ_SEQ_PAD && i < (s64)len + LZ_SEQ_PAD) ? code[i] : (u32)LZ_CODE_INVALID;
        if (c & LZ_CODE_INVALID) m |= 1u << k; else bits |= LZ_GRAY(LZ_CODE_BITS(c)) << (2 * k);
    }
    spc[j] = (u8)m;
    two[2 * (size_t)j] = (u8)bits; two[2 * (size_t)j + 1] = (u8)(bits >> 8);
}
// which byte values occur in raw[0..len): flags[b] != 0
__global__ void __launch_bounds__(LZ_TPB)
k_byte_presence(const u8* __restrict__ raw, u32 len, u32* __restrict__ flags)
{
    __shared__ u32 f[256];
    f[threadIdx.x] = 0;
    __syncthreads();
    const u32 nvec = (len + 15u) >> 4;                 // padded buffers: whole 16-byte groups are readable (padding is 0)
    for (u32 v = blockIdx.x * LZ_TPB + threadIdx.x; v < nvec; v += gridDim.x * LZ_TPB) {
        const uint4 x = reinterpret_cast<const uint4*>(raw)[v];
        const u32 w[4] = { x.x, x.y, x.z, x.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 a = w[k];
            if ((size_t)v * 16 + 4 * k + 0 < len) f[a & 255u] = 1;
            if ((size_t)v * 16 + 4 * k + 1 < len) f[(a >> 8) & 255u] = 1;
            if ((size_t)v * 16 + 4 * k + 2 < len) f[(a >> 16) & 255u] = 1;
            if ((size_t)v * 16 + 4 * k + 3 < len) f[a >> 24] = 1;
        }
    }
    __syncthreads();
    if (f[threadIdx.x]) flags[threadIdx.x] = 1;
}
// Block k of dst = bytes [32k, 32k + 64) of src: every run of up to 32 bytes of src lies inside ONE 64-byte line of
// dst (the block its first byte's 32-byte half starts).  Phase A reads 31-32 consecutive bytes of the target's 2-bit
// array around every hit, at a random place: from the plain array that is 1.5 cache lines per hit on average, and the
// scan kernel runs exactly as fast as its CU's L1 can have lines in flight (64 x 64 B per ~460 cycles).
__global__ void __launch_bounds__(LZ_TPB)
k_overlap32(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t nchunks)
{
    const size_t v = (size_t)blockIdx.x * LZ_TPB + threadIdx.x;           // 16-byte chunk of dst: block v / 4, quarter v % 4
    if (v < nchunks) dst[v] = src[2 * (v >> 2) + (v & 3u)];
}
int lzk_overlap32(LzCtx& c, const u8* src, u8* dst, size_t nblocks)
{
    const size_t nchunks = nblocks * 4;
    hipLaunchKernelGGL(k_overlap32, dim3((unsigned)((nchunks + LZ_TPB - 1) / LZ_TPB)), dim3(LZ_TPB), 0, c.stream,
                       reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), nchunks);
    LZ_HIP(hipGetLastError());
    return 0;
}

int lzk_pack2(LzCtx& c, const u8* code_base, const u8* raw_base, u32 len, u8* two, u8* spc, u32 nmask, u32* flags256)
{
    LZ_HIP(hipMemsetAsync(flags256, 0, 256 * 4, c.stream));
    c.timer.begin("k_pack2", c.stream);
    hipLaunchKernelGGL(k_pack2, dim3((nmask + LZ_TPB - 1) / LZ_TPB), dim3(LZ_TPB), 0, c.stream, code_base, len, two, spc, nmask);
    c.timer.end(c.stream);
    if (len) {
        u32 blocks = (((len + 15u) >> 4) + LZ_TPB - 1) / LZ_TPB; if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_byte_presence, dim3(blocks), dim3(LZ_TPB), 0, c.stream, raw_base, len, flags256);
    }
    LZ_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// B2 step 3: the hits of a chunk, which k_fill_hits wrote in discovery order, are (a) scanned independently of
// the diagonal hash (phase A) and (b) stably partitioned by the high 8 bits of hashedDiag into 256 streams of
// 8-byte records (lz_lut.hpp), one stream per workgroup of phase B.  One pass over the keys for the partition
// offsets (k_hist + two small scans, on the partition bytes k_fill_hits left), one that scans (k_scan_hits,
// k_scan_tasks), one that scatters (k_partition): 8 B + 4 B read and 8 B written per hit, where a radix sort of
// (key, summary) pairs moved 56.
// -DLZ_PHASE_CLOCKS: per-phase shader-clock totals of the phase kernels (lane 0 of every workgroup adds
// its s_memtime deltas to a device array the host prints at shutdown); off in the product build
#if defined(LZ_PHASE_CLOCKS)
__device__ unsigned long long g_phase_clk[32];
#define LZ_CLK_DECL  unsigned long long _t0 = __builtin_readcyclecounter(), _t1
#define LZ_CLK(slot) do { if (threadIdx.x == 0) { _t1 = __builtin_readcyclecounter(); atomicAdd(&g_phase_clk[slot], _t1 - _t0); _t0 = _t1; } } while (0)
void lz_phase_clocks_print()
{
    unsigned long long h[32];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_clk), sizeof(h)) != hipSuccess) return;
    fprintf(stderr, "[lzgpu phase clocks] probe_part:");
    for (int k = 0; k < 10; k++) fprintf(stderr, " %llu", h[k]);
    fprintf(stderr, "\n[lzgpu phase clocks] settle (walker work, walker barrier wait, sorter work, sorter barrier wait, extensions of wave 0: cycles, count; cycles summed over the 256 partitions):");
    for (int k = 16; k < 22; k++) fprintf(stderr, " %llu", h[k]);
    fprintf(stderr, "\n");
}
// k_settle2: a walking wave's (slots 16, 17) and a sorting wave's (18, 19) cycles of work / of waiting at the tile barrier
#define LZ_S2_CLK_BEGIN(who) unsigned long long _s2a = (who) ? __builtin_readcyclecounter() : 0ull, _s2b = 0ull
#define LZ_S2_CLK_MID(who, slot) do { if (who) { _s2b = __builtin_readcyclecounter(); atomicAdd(&g_phase_clk[slot], _s2b - _s2a); } } while (0)
#define LZ_S2_CLK_END(who, slot) do { if (who) atomicAdd(&g_phase_clk[slot], __builtin_readcyclecounter() - _s2b); } while (0)
// ... of the walker's work: cycles in the wave-cooperative extensions of SLOW records (slot 20) and how many (21)
#define LZ_S2_CLK_EXT_BEGIN(who, m) const unsigned long long _s2x = (who) ? __builtin_readcyclecounter() : 0ull; const unsigned _s2n = (unsigned)__popcll(m)
#define LZ_S2_CLK_EXT_END(who) do { if (who) { atomicAdd(&g_phase_clk[20], __builtin_readcyclecounter() - _s2x); atomicAdd(&g_phase_clk[21], (unsigned long long)_s2n); } } while (0)
#else
#define LZ_CLK_DECL
#define LZ_CLK(slot)
#define LZ_S2_CLK_BEGIN(who)
#define LZ_S2_CLK_MID(who, slot)
#define LZ_S2_CLK_END(who, slot)
#define LZ_S2_CLK_EXT_BEGIN(who, m)
#define LZ_S2_CLK_EXT_END(who)
void lz_phase_clocks_print() {}
#endif
#ifndef LZ_PP_TPB
#define LZ_PP_TPB    1024                            // one workgroup per CU: 128 KiB of staging for a tile of 16384 hits (runs of 64 records per partition)
#endif
#define LZ_PP_WAVES  (LZ_PP_TPB / 64)
#define LZ_SC_ROUNDS 4                               // k_scan_hits: rounds of 64 hits a wave takes per span
#ifndef LZ_PP_ROUNDS
#define LZ_PP_ROUNDS (LZ_PP_TILE_HOST / LZ_PP_TPB)   // k_partition: records per lane
#endif
#define LZ_PP_TILE   (LZ_PP_TPB * LZ_PP_ROUNDS)      // hits per tile of k_hist / k_partition
#define LZ_PP_QCAP   96                              // unfinished scans a tile can queue (beyond that the hit is left to phase B)
#define LZ_NBIN      256

// hist[tile][bin]: hits of the tile per partition
__global__ void __launch_bounds__(LZ_TPB)
k_hist(const u8* __restrict__ bins, u64 n, u32* __restrict__ hist)
{
    __shared__ u32 cnt[LZ_NBIN];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * LZ_PP_TILE;
#pragma unroll
    for (int r = 0; r < LZ_PP_TILE / (LZ_TPB * 16); r++) {      // 16 partition bytes per lane and step
        const u64 i = base + ((u64)r * LZ_TPB + threadIdx.x) * 16u;
        if (i < n) {
            const LzVec16 v = lz_load16(bins + i);
#pragma unroll
            for (int k = 0; k < 16; k++) if (i + (u64)k < n) atomicAdd(&cnt[LZ_VBYTE(v, k)], 1u);
        }
    }
    __syncthreads();
    hist[(size_t)blockIdx.x * LZ_NBIN + threadIdx.x] = cnt[threadIdx.x];
}
// per block of 256 tiles: exclusive prefix down each column, column sums to part[block][bin]
__global__ void __launch_bounds__(LZ_NBIN)
k_hist_scan1(u32* __restrict__ hist, u32 ntiles, u32* __restrict__ part)
{
    const u32 t0 = blockIdx.x * 256u, t1 = (t0 + 256u < ntiles) ? t0 + 256u : ntiles;
    u32 acc = 0;
    for (u32 t = t0; t < t1; t++) { const size_t k = (size_t)t * LZ_NBIN + threadIdx.x; const u32 v = hist[k]; hist[k] = acc; acc += v; }
    part[(size_t)blockIdx.x * LZ_NBIN + threadIdx.x] = acc;
}
// one workgroup: part[block][bin] -> absolute offset of (block, bin); bin_base[0..256]
__global__ void __launch_bounds__(LZ_NBIN)
k_hist_scan2(u32* __restrict__ part, u32 nblocks, u32* __restrict__ bin_base)
{
    __shared__ u32 tot[LZ_NBIN];
    u32 acc = 0;
    for (u32 b = 0; b < nblocks; b++) { const size_t k = (size_t)b * LZ_NBIN + threadIdx.x; const u32 v = part[k]; part[k] = acc; acc += v; }
    tot[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { u32 a = 0; for (int k = 0; k < LZ_NBIN; k++) { const u32 v = tot[k]; tot[k] = a; a += v; } bin_base[LZ_NBIN] = a; }
    __syncthreads();
    const u32 bb = tot[threadIdx.x];
    bin_base[threadIdx.x] = bb;
    for (u32 b = 0; b < nblocks; b++) part[(size_t)b * LZ_NBIN + threadIdx.x] += bb;
}

int lzk_hist(LzCtx& c, const u8* bins, u64 n, u32* hist, u32* part, u32* bin_base, hipStream_t st)
{
    const u32 ntiles = (u32)((n + LZ_PP_TILE - 1) / LZ_PP_TILE), nblocks = (ntiles + 255u) / 256u;
    c.timer.begin("k_hist", st);
    hipLaunchKernelGGL(k_hist, dim3(ntiles), dim3(LZ_TPB), 0, st, bins, n, hist);
    c.timer.end(st);
    c.timer.begin("k_hist_scan", st);
    hipLaunchKernelGGL(k_hist_scan1, dim3(nblocks), dim3(LZ_NBIN), 0, st, hist, ntiles, part);
    hipLaunchKernelGGL(k_hist_scan2, dim3(1), dim3(LZ_NBIN), 0, st, part, nblocks, bin_base);
    c.timer.end(st);
    LZ_HIP(hipGetLastError());
    return 0;
}

// rank of the lane among the lanes of its wave that hold the same 8-bit key (lower lanes first), the size of
// that group and whether the lane is its last member; lanes with valid == false are in no group
__device__ __forceinline__ void lz_match8(u32 key, bool valid, u32 lane, u32& rank, u32& count, bool& last)
{
    u64 peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = ((key >> b) & 1u) != 0;
        const u64 m = __ballot(bit && valid);
        peers &= bit ? m : ~m;
    }
    const u64 below = (1ull << lane) - 1ull;
    rank = (u32)__popcll(peers & below);
    count = (u32)__popcll(peers);
    last = (peers >> lane) == 1ull;
}
// exclusive prefix sum over the first 256 threads of a workgroup (every thread of the workgroup calls it)
__device__ __forceinline__ u32 lz_exscan256(u32 v, u32* wtot /*LDS, [4]*/)
{
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    u32 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if ((int)lane >= d) inc += t; }
    if (tid < 256 && lane == 63) wtot[w] = inc;
    __syncthreads();
    u32 pre = 0;
    if (tid < 256) for (u32 k = 0; k < w; k++) pre += wtot[k];
    return pre + inc - v;
}

// ---- phase A + the partition, three kernels:
//   k_scan_hits   every hit's two scans, first window each (the heavy, branch-free code of lz_lut.hpp): a pure
//                 stream -- a wave takes 256 consecutive hits, 64 at a time, with the next 64 hits' windows in
//                 flight -- with no barrier and nothing in LDS but the 32 KiB table, so every resident wave is in
//                 this code all the time.  Output: the 4-byte summary of a hit whose scans both ended; the rare hit
//                 with a scan that goes on (~3 %) becomes a 64-byte task in a global list (one atomic per wave).
//   k_scan_tasks  one lane per task: the scans that go on, to their end or the LZ_LUT_MAXWIN cap; full waves.
//   k_partition   keys + summaries -> records, stably partitioned into the 256 streams (tile histogram of k_hist).
// (One fused kernel did all of this per tile behind barriers: its workgroups spent half their time in the light
// phases -- queue drain on two waves, ranks, write-out -- while holding the LDS and the wave slots the scans need.)
struct LzScanTask { u32 idx; s32 diag; LzLutScan L, R; };
static_assert(sizeof(LzScanTask) == 64, "LzScanTask: one 64-byte line per task");
#define LZ_SC_TPB 512                                // k_scan_hits<1/2>, k_scan_tasks regions; k_scan_hits<0> picks its own size (TPB template parameter)
template <int MODE> struct LzScanShared {
    union {
        LzLutEntry lut[LZ_LUT_TOTAL];                                   // MODE 0/1: the two look-up tables (64 KiB)
        struct { s32 tab[LZ_NCLASS * LZ_NCLASS]; s32 tab8[64]; } bc;    // MODE 2: the byte-code scans' tables
    };
    s32 ctab[MODE == 1 ? LZ_NCLASS * LZ_NCLASS : 1];                    // MODE 1: the class table (a special base is scored from it)
};

// one round of k_scan_hits<0/1>: both scans' first windows of the hit `key`, whose windows (rawl, rawr) are loaded.
// MODE 0 heads run without limit tests: a side with less than 60 bases of room is left to k_scan_tasks.
template <bool SP>
__device__ __forceinline__ void lz_scan_round(const LzExtendParams& P, const LzLutParams& Q, const LzLutEntry* lut, const s32* ctab, u64 key, bool valid,
                                              const LzLutRaw<SP>& rawl, const LzLutRaw<SP>& rawr, u32 idx, u32 lane,
                                              u32& summ_out, LzScanTask* __restrict__ my_tasks, u32& my_n, u32 region_cap)
{
    constexpr bool HLIM = SP;
    s32 diag; LzLutScan L, R;
    lz_lut_init(key, P.tlen, P.qlen, diag, L, R);
    if (!valid) { L.alive = 0; R.alive = 0; }
    const bool ql = L.alive && !HLIM && L.room < (u32)LZ_LUT_WIN_B, qr = R.alive && !HLIM && R.room < (u32)LZ_LUT_WIN_B;
    lz_lut_window_pair<SP, HLIM>(Q, lut, diag, L, R, rawl, rawr, L.alive && !ql, R.alive && !qr, ctab);
    const bool more = valid && (L.alive == 1 || R.alive == 1);
    const u64 mm = __ballot(more);
    bool queued = false;
    if (mm) {                                                    // (wave-uniform)
        const u32 slot = my_n + (u32)__popcll(mm & ((1ull << lane) - 1ull));
        if (more && slot < region_cap) {                         // (a full region leaves the scan "alive": the hit becomes SLOW)
            LzScanTask t; t.idx = idx; t.diag = diag; t.L = L; t.R = R;
            my_tasks[slot] = t; queued = true;
        }
        my_n += (u32)__popcll(mm);
    }
    summ_out = queued ? 0u : lz_lut_summary(L, R, P.min_score);          // (a queued hit's summary comes from k_scan_tasks, behind this kernel)
}
// The windows of a hit: 2 x 16 bytes of the target's 2-bit codes from the half-overlapping blocks (the left window starts at byte bl,
// the right one at br = bl + 15 or 16, together at most 32 bytes: ONE line of t2x -- block bl / 32, offset bl % 32), 2 x 16 bytes of
// the query's from the plain array (the hits of a wave are in discovery order: runs of lanes ask for the same query line), and with
// special bytes about (SP) the same of the two 1-bit masks.  (Variants that were measured and lost -- query windows shared through the
// LDS crossbar, the plain target array, timing-only what-ifs without the target / with an L2-resident target -- are patches under
// tools/experiments/, applied by tools/build_variant.sh.)
template <bool SP>
__device__ __forceinline__ void lz_scan_fetch(const LzLutParams& Q, u64 key, LzLutRaw<SP>& rawl, LzLutRaw<SP>& rawr)
{
    const u32 pos2 = (u32)key, pos1 = pos2 + (u32)(key >> 32);
    const s32 diag = (s32)(u32)(key >> 32);
    const u32 stl = pos1 - 1u + (u32)LZ_PAD2, str = pos1 + (u32)LZ_PAD2;
    const u32 bl = (stl >> 2) - 15u, br = str >> 2;
    const u32 ol = bl + (bl & ~31u);
    rawl.tv = lz_load16(Q.t2x + ol); rawr.tv = lz_load16(Q.t2x + (ol + (br - bl)));
    const u32 sql = stl - (u32)diag, sqr = str - (u32)diag;
    rawl.qv = lz_load16(Q.q2 + ((sql >> 2) - 15u)); rawr.qv = lz_load16(Q.q2 + (sqr >> 2));
    if constexpr (SP) {
        const u32 ml = (stl >> 3) - 14u, mr = str >> 3;           // (mr - ml is 14 or 15: at most 31 bytes)
        const u32 oml = ml + (ml & ~31u);
        rawl.tm = lz_load16(Q.tspx + oml); rawr.tm = lz_load16(Q.tspx + (oml + (mr - ml)));
        rawl.qm = lz_load16(Q.qsp + ((sql >> 3) - 14u)); rawr.qm = lz_load16(Q.qsp + (sqr >> 3));
    }
}

// TPB lanes per workgroup, WPE waves per SIMD the register allocation must allow (two workgroups share a CU's LDS)
template <int MODE, int TPB, int WPE>      // MODE 0: LUT scans, no special bytes in either sequence; 1: LUT scans + special masks; 2: byte-code scans
__global__ void __launch_bounds__(TPB, WPE)
k_scan_hits(LzExtendParams P, LzLutParams Q, const u64* __restrict__ keys, u64 n,
            const s32* __restrict__ score_tab_g, const LzLutEntry* __restrict__ lut_g,
            u32* __restrict__ summ, u8* __restrict__ bins, LzScanTask* __restrict__ tasks, u32* __restrict__ n_tasks, u32 region_cap)
{
    __shared__ LzScanShared<MODE> sh;
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    if (MODE == 1) { for (u32 k = tid; k < LZ_NCLASS * LZ_NCLASS; k += TPB) sh.ctab[k] = score_tab_g[k]; }
    if (MODE < 2) { for (u32 k = tid; k < LZ_LUT_TOTAL; k += TPB) sh.lut[k] = lut_g[k]; }
    else {
        for (u32 k = tid; k < LZ_NCLASS * LZ_NCLASS; k += TPB) sh.bc.tab[k] = score_tab_g[k];
        if (tid < 64) sh.bc.tab8[tid] = score_tab_g[(tid >> 3) * LZ_NCLASS + (tid & 7)];
    }
    __syncthreads();
    const LzLutEntry* lut = sh.lut;
    const s32* const ctab = sh.ctab;
    constexpr bool SP = MODE == 1;
    constexpr u32 SPAN = 64u * LZ_SC_ROUNDS;
    static_assert(LZ_SC_ROUNDS == 4, "the round pipeline below is written out for four rounds per span");
    const u64 nspans = (n + SPAN - 1) / SPAN, wstride = (u64)gridDim.x * (TPB / 64);
    // the tasks of a wave go to the wave's own region of the list: no atomics, the count is written once at the end
    const u32 region = blockIdx.x * (TPB / 64) + w;
    LzScanTask* const my_tasks = tasks + (size_t)region * region_cap;
    u32 my_n = 0;
    u64 span = (u64)blockIdx.x * (TPB / 64) + w;
    if (MODE == 2) {
        for (; span < nspans; span += wstride) {
            const u64 base = span * SPAN;
            const u32 span_n = (n - base < (u64)SPAN) ? (u32)(n - base) : SPAN;
#pragma unroll 1
            for (u32 r = 0; r < LZ_SC_ROUNDS; r++) {
                const u32 li = r * 64u + lane;
                if (li < span_n) { const u64 kk = keys[base + li]; summ[base + li] = lz_probe_hit(P, sh.bc.tab, sh.bc.tab8, P.cls8 != 0, kk); bins[base + li] = (u8)LZ_KEY_BIN(kk); }
            }
        }
    } else if (span < nspans) {
        // A wave takes 256 consecutive hits (a span); a LANE takes four consecutive ones, one per round: its keys are two 16-byte loads,
        // its four summaries one 16-byte store, its four partition bytes (for k_hist) one 4-byte store -- 64 consecutive bytes of a
        // span's 256 went out per round when a lane's hits were 64 apart (the partition bytes as 64 one-byte stores: +2 ms on the 50 Mbp
        // pair; the fill kernel used to write them beside its keys, in runs of ~39 bytes at random places, two partial lines each: 4-7 of
        // its 19-26 ms).  The windows of a round are requested one round ahead into two register sets that take turns (no copies), the
        // next span's keys two rounds ahead and its first windows during the span's last round: every load has a round of arithmetic to
        // hide behind.
        typedef u64 lz_u64x2 __attribute__((ext_vector_type(2)));
        auto load_keys = [&](u64 sp, u64& a0, u64& a1, u64& a2, u64& a3) {
            const u64 base = sp * SPAN;
            const u32 sn = (n - base < (u64)SPAN) ? (u32)(n - base) : SPAN;
            const u32 l4 = 4u * lane;
            const u64* kp = keys + base + l4;
            if (l4 + 3u < sn) {
                const lz_u64x2 x = LZ_NT_LD(reinterpret_cast<const lz_u64x2*>(kp)), y = LZ_NT_LD(reinterpret_cast<const lz_u64x2*>(kp + 2));
                a0 = x.x; a1 = x.y; a2 = y.x; a3 = y.y;
            } else {
                a0 = (l4 < sn) ? LZ_NT_LD(kp) : 0ull;           a1 = (l4 + 1u < sn) ? LZ_NT_LD(kp + 1) : 0ull;
                a2 = (l4 + 2u < sn) ? LZ_NT_LD(kp + 2) : 0ull;  a3 = 0ull;
            }
        };
        u64 k0, k1, k2, k3;
        load_keys(span, k0, k1, k2, k3);
        LzLutRaw<SP> al, ar, bl, br;
        lz_scan_fetch<SP>(Q, k0, al, ar);
#pragma unroll 1
        for (;;) {
            const u64 base = span * SPAN;
            const u32 span_n = (n - base < (u64)SPAN) ? (u32)(n - base) : SPAN;
            const u32 l4 = 4u * lane;
            const u32 ib = (u32)base + l4;                       // (hit indices inside a chunk are 32-bit: lzgpu_set_hit_capacity)
            const bool more_spans = span + wstride < nspans;
            u64 n0 = 0, n1 = 0, n2 = 0, n3 = 0;
            u32 s0, s1, s2, s3;
            lz_scan_fetch<SP>(Q, k1, bl, br);
            lz_scan_round<SP>(P, Q, lut, ctab, k0, l4 < span_n, al, ar, ib, lane, s0, my_tasks, my_n, region_cap);
            if (more_spans) load_keys(span + wstride, n0, n1, n2, n3);
            lz_scan_fetch<SP>(Q, k2, al, ar);
            lz_scan_round<SP>(P, Q, lut, ctab, k1, l4 + 1u < span_n, bl, br, ib + 1u, lane, s1, my_tasks, my_n, region_cap);
            lz_scan_fetch<SP>(Q, k3, bl, br);
            lz_scan_round<SP>(P, Q, lut, ctab, k2, l4 + 2u < span_n, al, ar, ib + 2u, lane, s2, my_tasks, my_n, region_cap);
            if (more_spans) lz_scan_fetch<SP>(Q, n0, al, ar);
            lz_scan_round<SP>(P, Q, lut, ctab, k3, l4 + 3u < span_n, bl, br, ib + 3u, lane, s3, my_tasks, my_n, region_cap);
            const u32 pbytes = LZ_KEY_BIN(k0) | (LZ_KEY_BIN(k1) << 8) | (LZ_KEY_BIN(k2) << 16) | (LZ_KEY_BIN(k3) << 24);       // (the keys are still in their registers)
            if (l4 + 3u < span_n) {
                typedef u32 lz_u32x4 __attribute__((ext_vector_type(4)));
                lz_u32x4 sv4; sv4.x = s0; sv4.y = s1; sv4.z = s2; sv4.w = s3;
                LZ_NT_ST(sv4, reinterpret_cast<lz_u32x4*>(summ + ib));
                LZ_NT_ST(pbytes, reinterpret_cast<u32*>(bins + ib));
            } else {
                if (l4 < span_n)      { summ[ib] = s0;      bins[ib] = (u8)pbytes; }
                if (l4 + 1u < span_n) { summ[ib + 1u] = s1; bins[ib + 1u] = (u8)(pbytes >> 8); }
                if (l4 + 2u < span_n) { summ[ib + 2u] = s2; bins[ib + 2u] = (u8)(pbytes >> 16); }
            }
            if (!more_spans) break;
            span += wstride; k0 = n0; k1 = n1; k2 = n2; k3 = n3;
        }
    }
    if (lane == 0) n_tasks[region] = my_n < region_cap ? my_n : region_cap;
}

#ifndef LZ_ST_TPB
#define LZ_ST_TPB 512                                // lanes per workgroup of k_scan_tasks: two workgroups share a CU (the tables are 64 KiB of LDS), 16 waves per CU (256: 5.3 ms per step, 512: 4.4, 1024: 4.2)
#endif
template <int MODE>
__global__ void __launch_bounds__(LZ_ST_TPB)
k_scan_tasks(LzExtendParams P, LzLutParams Q, const LzLutEntry* __restrict__ lut_g, const s32* __restrict__ score_tab_g, const LzScanTask* __restrict__ tasks,
             const u32* __restrict__ n_tasks, u32 n_regions, u32 region_cap, u32* __restrict__ summ)
{
    __shared__ LzLutEntry lut[LZ_LUT_TOTAL];
    __shared__ s32 ctab[MODE == 1 ? LZ_NCLASS * LZ_NCLASS : 1];
    for (u32 k = threadIdx.x; k < LZ_LUT_TOTAL; k += LZ_ST_TPB) lut[k] = lut_g[k];
    if (MODE == 1) for (u32 k = threadIdx.x; k < LZ_NCLASS * LZ_NCLASS; k += LZ_ST_TPB) ctab[k] = score_tab_g[k];
    __syncthreads();
    constexpr bool SP = MODE == 1;
    for (u32 region = blockIdx.x; region < n_regions; region += gridDim.x) {
        const u32 nt = n_tasks[region];
        for (u32 k = threadIdx.x; k < nt; k += (u32)LZ_ST_TPB) {
            const LzScanTask t = tasks[(size_t)region * region_cap + k];
            LzLutScan L = t.L, R = t.R;
            while (L.alive == 1 && L.nwin < (u32)LZ_LUT_MAXWIN) lz_lut_step<false, SP>(Q, lut, t.diag, L, ctab);
            while (R.alive == 1 && R.nwin < (u32)LZ_LUT_MAXWIN) lz_lut_step<true, SP>(Q, lut, t.diag, R, ctab);
            LZ_NT_ST(lz_lut_summary(L, R, P.min_score), summ + t.idx);
        }
    }
}

// keys + summaries -> records in their partitions.  One workgroup per tile of k_hist; a wave owns 256 consecutive
// hits (64 per round), so ranks by (wave, round, lane) follow the discovery order.
struct LzPartShared {
    union {
        u64 stage[LZ_PP_TILE];                       // the tile's records, ordered by partition (the partition rides in bits 55..62)
        u64 bm[LZ_PP_WAVES][LZ_NBIN];                // BM: peer bitmaps of the round in flight (all zero again before stage[] is written)
    };
    u32 wcnt[LZ_PP_WAVES][LZ_NBIN];                  // per wave and partition: records / running offset inside the partition
    u32 tstart[LZ_NBIN + 1], gbase[LZ_NBIN], wtot[4];
};
// BM == false: round 2's ranks (LDS atomics count, an 8-ballot match ranks: half of the kernel's 161 VALU instructions
// per hit).  BM == true: the lanes of a round that hold the same partition find each other through a 64-bit bitmap
// in LDS (atomic OR of 1 << lane, one read; the result does not depend on the order of the ORs), which gives rank
// and count in one pass and leaves the per-wave totals behind -- no separate counting pass, no ballots.
template <bool BM>
__global__ void __launch_bounds__(LZ_PP_TPB)
k_partition(const u64* __restrict__ keys, const u32* __restrict__ summ, u64 n,
            const u32* __restrict__ hist, const u32* __restrict__ part, u64* __restrict__ recs)
{
    __shared__ LzPartShared sh;
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const u32 tile = blockIdx.x;
    const u64 base = (u64)tile * LZ_PP_TILE;
    const u32 tile_n = (n - base < (u64)LZ_PP_TILE) ? (u32)(n - base) : (u32)LZ_PP_TILE;
    for (u32 k = tid; k < LZ_PP_WAVES * LZ_NBIN; k += LZ_PP_TPB) { (&sh.wcnt[0][0])[k] = 0; if (BM) (&sh.bm[0][0])[k] = 0ull; }
    const u32 l0 = w * (64u * LZ_PP_ROUNDS) + lane;
    u64 kk[LZ_PP_ROUNDS]; u32 ss[LZ_PP_ROUNDS];
#pragma unroll
    for (u32 r = 0; r < LZ_PP_ROUNDS; r++) {
        const bool v = l0 + 64u * r < tile_n;
        kk[r] = v ? keys[base + l0 + 64u * r] : 0ull;             // (plain loads and stores here: non-temporal ones made this kernel 10-15 % slower)
        ss[r] = v ? summ[base + l0 + 64u * r] : 0u;
    }
    __syncthreads();
    u32 slot[BM ? LZ_PP_ROUNDS : 1];
    if (BM) {
        // (each wave works on its own rows of bm / wcnt: LDS operations of one wave execute in order)
#pragma unroll
        for (u32 r = 0; r < LZ_PP_ROUNDS; r++) {
            const bool valid = l0 + 64u * r < tile_n;
            const u32 bin = LZ_KEY_BIN(kk[r]);
            if (valid) atomicOr((unsigned long long*)&sh.bm[w][bin], 1ull << lane);
            const u64 peers = valid ? sh.bm[w][bin] : 0ull;
            const u32 old = valid ? sh.wcnt[w][bin] : 0u;
            if (valid && (peers >> lane) == 1ull) { sh.wcnt[w][bin] = old + (u32)__popcll(peers); sh.bm[w][bin] = 0ull; }   // the highest peer
            slot[r] = old + (u32)__popcll(peers & ((1ull << lane) - 1ull));
        }
    } else {
#pragma unroll
        for (u32 r = 0; r < LZ_PP_ROUNDS; r++) if (l0 + 64u * r < tile_n) atomicAdd(&sh.wcnt[w][LZ_KEY_BIN(kk[r])], 1u);
    }
    __syncthreads();
    // per-partition counts chained in wave order (= discovery order) ...
    u32 tot = 0;
    if (tid < LZ_NBIN) {
        for (u32 k = 0; k < LZ_PP_WAVES; k++) { const u32 v = sh.wcnt[k][tid]; sh.wcnt[k][tid] = tot; tot += v; }
        sh.gbase[tid] = part[(size_t)(tile >> 8) * LZ_NBIN + tid] + hist[(size_t)tile * LZ_NBIN + tid];
    }
    const u32 ts = lz_exscan256(tot, sh.wtot);
    if (tid < LZ_NBIN) sh.tstart[tid] = ts;
    if (tid == 0) sh.tstart[LZ_NBIN] = tile_n;
    __syncthreads();
    // ... then every record gets its place: rank among the same-partition lanes of its wave's round, on top of
    // the wave's running offset (stable: lanes, rounds and waves all follow the discovery order)
#pragma unroll
    for (u32 r = 0; r < LZ_PP_ROUNDS; r++) {
        const bool valid = l0 + 64u * r < tile_n;
        const u32 bin = LZ_KEY_BIN(kk[r]);
        if (BM) {
            if (valid) sh.stage[sh.tstart[bin] + sh.wcnt[w][bin] + slot[r]] = lz_hit_record(kk[r], ss[r]) | ((u64)bin << 55);
        } else {
            u32 rank, count; bool last;
            lz_match8(bin, valid, lane, rank, count, last);
            const u32 old = sh.wcnt[w][bin];
            if (valid && last) sh.wcnt[w][bin] = old + count;
            if (valid) sh.stage[sh.tstart[bin] + old + rank] = lz_hit_record(kk[r], ss[r]) | ((u64)bin << 55);
        }
    }
    __syncthreads();
    for (u32 k = tid; k < tile_n; k += LZ_PP_TPB) {
        const u64 r = sh.stage[k];
        const u32 b = (u32)(r >> 55) & 0xFFu;
        recs[(size_t)sh.gbase[b] + (k - sh.tstart[b])] = r & ~(0xFFull << 55);
    }
}

// grid of k_scan_hits for n hits, and the geometry of its task list: one region per wave, room for 1/32 of the
// wave's hits + 64 (64 B each; the usual load is ~3 %); a hit that finds its region full is left to phase B
static int lz_scan_tpb(int mode)
{
    static const int env = getenv("LZGPU_SC_TPB") ? atoi(getenv("LZGPU_SC_TPB")) : 0;      // A/B aid: 512, 640, 768 or 1024
    if (mode != 0) return LZ_SC_TPB;
    return (env == 512 || env == 640 || env == 768 || env == 1024) ? env : 640;     // 640: five waves per SIMD at 89 VGPRs (512: 72.5, 640: 69.7, 768 with 8 spilled registers: 75.9 ms per step)
}
static void lz_scan_geometry(LzCtx& c, int mode, u64 n, u32& grid, u32& n_regions, u32& region_cap)
{
    const u32 wpg = (u32)lz_scan_tpb(mode) / 64u;
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c.device);
    static const u32 wgs = getenv("LZGPU_PP_WGS") ? (u32)atoi(getenv("LZGPU_PP_WGS")) : 2u;
    const u64 nspans = (n + 64u * LZ_SC_ROUNDS - 1) / (64u * LZ_SC_ROUNDS), want = (nspans + wpg - 1) / wpg;
    grid = (u32)std::min<u64>(want ? want : 1, (u64)wgs * (u64)cus);
    n_regions = grid * wpg;
    // (64 B each; ~3 % of the hits become tasks on plain sequences: 1/32 of them fit, 2 GiB less to allocate than with 1/8.
    // With special bytes that do not end a scan -- IUPAC codes -- every window that meets one goes on as a task as well:
    // 1/12, or the overflow lands on phase B's slow path: 1 % of the hits there took k_settle2 from 20 to 84 ms.)
    region_cap = (u32)std::min<u64>(n / (mode == 1 ? 12 : 32) / n_regions + 64, 1u << 20);
    static const char* force = getenv("LZGPU_TASK_REGION_CAP");  // test hook: tiny regions, so that hits find theirs full
    if (force && atoi(force) > 0) region_cap = (u32)atoi(force);
}
// buffers of a set for chunks of up to max_n hits (sized once per search: chunk sizes differ a little, and a
// device buffer that grows is freed and allocated again)
int lzk_scan_reserve(LzCtx& c, int set, int mode, u64 max_n)
{
    u32 grid, n_regions, region_cap; int rc;
    lz_scan_geometry(c, mode, max_n, grid, n_regions, region_cap);
    if ((rc = c.summ[set].ensure((size_t)max_n * 4))) return rc;
    if (mode < 2 && (rc = c.scan_tasks[set].ensure((size_t)n_regions * region_cap * sizeof(LzScanTask)))) return rc;
    return c.scan_ntasks[set].ensure((size_t)n_regions * 4);
}

int lzk_scan_hits(LzCtx& c, int set, int mode, const LzExtendParams& P, const LzLutParams& Q, const u64* keys, u64 n,
                  const s32* score_tab, const LzLutEntry* lut, u8* bins, hipStream_t st)
{
    if (n == 0) return 0;
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c.device);
    int rc;
    u32 grid, n_regions, task_cap;
    lz_scan_geometry(c, mode, n, grid, n_regions, task_cap);
    if ((rc = c.summ[set].ensure((size_t)n * 4))) return rc;
    if (mode < 2 && (rc = c.scan_tasks[set].ensure((size_t)n_regions * task_cap * sizeof(LzScanTask)))) return rc;
    if ((rc = c.scan_ntasks[set].ensure((size_t)n_regions * 4))) return rc;
    u32* summ = c.summ[set].as<u32>(); LzScanTask* tasks = c.scan_tasks[set].as<LzScanTask>(); u32* ntk = c.scan_ntasks[set].as<u32>();
    c.timer.begin("k_scan_hits", st);
    const int tpb = lz_scan_tpb(mode);
#define LZ_SCAN_LAUNCH(M_, T_, W_) hipLaunchKernelGGL((k_scan_hits<M_, T_, W_>), dim3(grid), dim3(T_), 0, st, P, Q, keys, n, score_tab, lut, summ, bins, tasks, ntk, task_cap)
    if (mode == 0) {
        if (tpb == 640)       LZ_SCAN_LAUNCH(0, 640, 5);
        else if (tpb == 768)  LZ_SCAN_LAUNCH(0, 768, 6);
        else if (tpb == 1024) LZ_SCAN_LAUNCH(0, 1024, 8);
        else                  LZ_SCAN_LAUNCH(0, 512, 4);
    }
    else if (mode == 1) LZ_SCAN_LAUNCH(1, LZ_SC_TPB, 4);
    else                LZ_SCAN_LAUNCH(2, LZ_SC_TPB, 4);
#undef LZ_SCAN_LAUNCH
    c.timer.end(st);
    LZ_HIP(hipGetLastError());
    if (mode < 2) {
        c.timer.begin("k_scan_tasks", st);
        if (mode == 0) hipLaunchKernelGGL(k_scan_tasks<0>, dim3(std::min<u32>(n_regions, 4u * (u32)cus)), dim3(LZ_ST_TPB), 0, st, P, Q, lut, score_tab, tasks, ntk, n_regions, task_cap, summ);
        else           hipLaunchKernelGGL(k_scan_tasks<1>, dim3(std::min<u32>(n_regions, 4u * (u32)cus)), dim3(LZ_ST_TPB), 0, st, P, Q, lut, score_tab, tasks, ntk, n_regions, task_cap, summ);
        c.timer.end(st);
        LZ_HIP(hipGetLastError());
    }
    return 0;
}

int lzk_partition(LzCtx& c, int set, const u64* keys, u64 n, const u32* hist, const u32* part, u64* recs, hipStream_t st)
{
    if (n == 0) return 0;
    const u32 ntiles = (u32)((n + LZ_PP_TILE - 1) / LZ_PP_TILE);
    c.timer.begin("k_partition", st);
    static const bool ballots = getenv("LZGPU_PARTITION_BALLOTS") != nullptr;  // A/B aid: round 2's ballot-match ranks
    if (ballots) hipLaunchKernelGGL(k_partition<false>, dim3(ntiles), dim3(LZ_PP_TPB), 0, st, keys, c.summ[set].as<u32>(), n, hist, part, recs);
    else         hipLaunchKernelGGL(k_partition<true>, dim3(ntiles), dim3(LZ_PP_TPB), 0, st, keys, c.summ[set].as<u32>(), n, hist, part, recs);
    c.timer.end(st);
    LZ_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// The X-drop extension of one hit by a whole wave (lz_coop.hpp): every argument is wave-uniform, all 64 lanes
// take part -- lanes 0..31 run the left scan (loop 1), lanes 32..63 the right scan (loop 2), 16 bases per lane and
// step.  tab = the 32 x 32 class table in LDS.  Every lane returns its own side's result.
__device__ __forceinline__ LzCoopMap lz_coop_shfl_up32(const LzCoopMap& v, int d)
{ LzCoopMap r; r.A = __shfl_up(v.A, d, 32); r.B = __shfl_up(v.B, d, 32); r.C = __shfl_up(v.C, d, 32); return r; }
__device__ LzCoopSide lz_coop_extend_wave(const LzExtendParams& P, const s32* tab, u32 pos1, s32 diag, s32 stopl, s32 stopr, u32 lane)
{
    const s32 X = P.xdrop;
    const bool right = lane >= 32u;
    const u32 hl = lane & 31u;
    const s32 stop = right ? stopr : stopl;
    LzCoopSide out; out.stop_pos = pos1; out.best_pos = pos1; out.best = 0;
    s32 run0 = 0, best0 = 0; u32 s = pos1;
    bool alive = (right ? ((s32)s < stop) : ((s32)s > stop)) && X >= 0;        // uniform inside a half
    while (__ballot(alive)) {
        const u32 room = alive ? (right ? (u32)(stop - (s32)s) : (u32)((s32)s - stop)) : 0u;
        const u32 off = LZ_COOP_BLK * hl;
        const u32 nb = room > off ? (room - off < (u32)LZ_COOP_BLK ? room - off : (u32)LZ_COOP_BLK) : 0u;
        s32 sc[LZ_COOP_BLK];
        {
            LzVec16 tv = { { 0, 0, 0, 0 } }, qv = { { 0, 0, 0, 0 } };
            if (nb) {
                const s64 t0 = right ? (s64)s + off : (s64)s - off - LZ_COOP_BLK;
                tv = lz_load16(P.tcode + t0); qv = lz_load16(P.qcode + (t0 - diag));
            }
#pragma unroll
            for (int j = 0; j < LZ_COOP_BLK; j++) {
                const u32 tb = right ? LZ_VBYTE(tv, j) : LZ_VBYTE(tv, LZ_COOP_BLK - 1 - j);      // consumption order
                const u32 qb = right ? LZ_VBYTE(qv, j) : LZ_VBYTE(qv, LZ_COOP_BLK - 1 - j);
                sc[j] = tab[(LZ_CODE_CLASS(tb) << 5) | LZ_CODE_CLASS(qb)];
            }
        }
        LzCoopMap f; s32 mx; u32 jmx;
        lz_coop_block(sc, nb, X, f, mx, jmx);
        // inclusive scan of the maps in lane order inside the half (lower lanes act first), then shifted by one lane
        LzCoopMap inc = f;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) { const LzCoopMap g = lz_coop_shfl_up32(inc, d); if ((int)hl >= d) inc = lz_coop_compose(g, inc); }
        LzCoopMap ex = lz_coop_shfl_up32(inc, 1);
        if (hl == 0) ex = lz_coop_identity(X);
        const s32 m0 = run0 - best0 + X;
        const bool dead_before = lz_coop_fails(ex, m0);
        const s32 mi = lz_coop_apply(ex, m0), ri = run0 + ex.C;
        const bool fails_here = nb && !dead_before && lz_coop_fails(f, mi);
        const u64 fboth = __ballot(fails_here);                 // at most one lane per half: the blocks after it are "dead before"
        const u32 fmask = right ? (u32)(fboth >> 32) : (u32)fboth;
        // best prefix: (value, first position) as one key, larger is better
        long long key = -1;
        u32 used_here = 0;
        if (fails_here) {
            u32 used; bool stopped; s32 rmx; u32 rj;
            lz_coop_resolve(sc, nb, X, mi, ri, used, stopped, rmx, rj);
            used_here = used;
            if (rj) key = (((long long)rmx + LZ_COOP_INF) << 16) | (long long)(0xFFFFu - (off + rj));
        } else if (nb && jmx && !dead_before)
            key = (((long long)ri + mx + LZ_COOP_INF) << 16) | (long long)(0xFFFFu - (off + jmx));
#pragma unroll
        for (int d = 16; d > 0; d >>= 1) { const long long o = __shfl_xor(key, d, 32); if (o > key) key = o; }
        const int fl = fmask ? (int)__ffs((int)fmask) - 1 : 0;
        const u32 used_f = (u32)__shfl((int)used_here, fl, 32);
        const s32 totc = __shfl(inc.C, 31, 32);
        if (alive) {
            u32 used_total;
            if (fmask) { used_total = LZ_COOP_BLK * (u32)fl + used_f; alive = false; }
            else {
                used_total = room < (u32)(LZ_COOP_BLK * LZ_COOP_LANES) ? room : (u32)(LZ_COOP_BLK * LZ_COOP_LANES);
                if (room <= (u32)(LZ_COOP_BLK * LZ_COOP_LANES)) alive = false;
            }
            if (key >= 0) {
                const s32 v = (s32)((key >> 16) - LZ_COOP_INF); const u32 idx = 0xFFFFu - (u32)(key & 0xFFFF);
                if (v > best0) { best0 = v; out.best_pos = right ? s + idx : s - idx; }
            }
            run0 += totc;
            s = right ? s + used_total : s - used_total;
        }
    }
    out.stop_pos = s; out.best = best0;
    return out;
}

// ------------------------------------------------------------------------------------------
#ifndef LZ_ST_BATCH
#define LZ_ST_BATCH  4                               // records a walking lane settles per straight-line step
#endif
// ------------------------------------------------------------------------------------------
// B2 phase B: one workgroup of 1024 lanes per partition (256 buckets), diagEnd of its buckets in registers.  The
// partition's records arrive in discovery order; tiles of them are dealt out to the buckets inside LDS (counting
// sort by the record's low hash bits, a bucket's list contiguous and in order) and every bucket's list is walked
// by one lane (the fast path of lz_settle_record): the serial part of the whole search is this short walk.  A
// record that needs a real extension (an HSP candidate that passed the diagEnd test) is extended by the lane's
// whole wave (lz_coop_extend_wave), one such record at a time.  The 16 waves have two roles and meet at ONE
// barrier per tile (round 2's kernel did count / offsets / place / walk one after the other behind five barriers,
// 16 walking lanes in every wave):
//   waves 0..3   walkers: lane l of wave w walks bucket 64 w + l of the tile that was placed during the previous
//                interval -- all 64 lanes of a walking wave are busy;
//   waves 4..15  sorters: while tile t is walked they place tile t+1 (offsets from the counts taken one interval
//                earlier, each wave computing the 256 bucket bases for itself from the per-wave counts: no barrier
//                between scan and placement) and count tile t+2 (ranks inside (wave, bucket)).
// Ranks are deterministic: the lanes of a round that hold the same bucket find each other through a 64-bit
// bitmap in LDS (atomic OR of 1 << lane, then a read: the set of peers is independent of the order in which the
// hardware applies the ORs), rank = peers in lower lanes, and the highest peer adds the group to the wave's
// running count.  (Round 2 took the rank from the return value of a same-address LDS atomic add and repaired
// disorder after the fact, which leans on an order the ISA does not promise; here no order is assumed anywhere.)
#define LZ_S2_TPB     1024
#ifndef LZ_S2_WALK_PRIO
#define LZ_S2_WALK_PRIO 0
#endif
#ifndef LZ_S2_WALKW
#define LZ_S2_WALKW   4                                  // walking waves: 256 / LZ_S2_WALKW buckets each
#endif
#define LZ_S2_SORTW   (LZ_S2_TPB / 64 - LZ_S2_WALKW)     // 12 sorting waves
#ifndef LZ_S2_ROUNDS
#define LZ_S2_ROUNDS  7                                  // records per sorter lane and tile (3: 26.7, 4: 24.6, 6: 22.8 ms per step; 6 -> 7 with one chunk per strand: 20.8 -> 20.1; 8 does not fit the LDS)
#endif
#define LZ_S2_TILE    (LZ_S2_SORTW * 64 * LZ_S2_ROUNDS)  // 5376
struct LzSettle2Shared {
    s32 tab[LZ_NCLASS * LZ_NCLASS];
    u64 rec[2][LZ_S2_TILE + LZ_ST_BATCH];                // the placed tiles (bucket-major), two take turns
    u64 bm[LZ_S2_SORTW][LZ_NBIN];                        // peer bitmaps of the round in flight (zero between rounds)
    u32 cnt[2][LZ_S2_SORTW][LZ_NBIN];                    // records per (wave, bucket) of the tile being counted / placed
    u32 woff[LZ_S2_SORTW][LZ_NBIN];                      // where wave w's records of bucket b start in the placed tile
    u32 lbeg[2][LZ_NBIN], lcnt[2][LZ_NBIN];              // bucket lists of the placed tiles
};
__global__ void __launch_bounds__(LZ_S2_TPB)
k_settle2(LzExtendParams P, const u64* __restrict__ recs, const u32* __restrict__ bin_base, u32* __restrict__ diag_end,
          const s32* __restrict__ score_tab_g, LzHspRec* __restrict__ out, u32* __restrict__ out_count, u32 out_cap,
          u64* __restrict__ counters)
{
    extern __shared__ __align__(16) unsigned char lz_s2_smem[];
    LzSettle2Shared& sh = *reinterpret_cast<LzSettle2Shared*>(lz_s2_smem);
    const u32 tid = threadIdx.x, lane = tid & 63u;
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // wave-uniform, and the compiler knows it
    const u32 part = blockIdx.x;
    const u32 r0 = bin_base[part], r1 = bin_base[part + 1];
    const u32 n = r1 - r0, ntiles = (n + LZ_S2_TILE - 1) / LZ_S2_TILE;
    if (ntiles == 0) return;                                    // (uniform: nothing of this partition in the chunk)
    for (u32 k = tid; k < LZ_NCLASS * LZ_NCLASS; k += LZ_S2_TPB) sh.tab[k] = score_tab_g[k];
    for (u32 k = tid; k < LZ_S2_SORTW * LZ_NBIN; k += LZ_S2_TPB) (&sh.bm[0][0])[k] = 0ull;
    const bool walker = w < LZ_S2_WALKW;
    const u32 sw = w - LZ_S2_WALKW;                             // sorter wave index (sorters only)

    // ---- sorter pieces
    u64 cx[LZ_S2_ROUNDS], nx[LZ_S2_ROUNDS];                     // records of the tile to place next / to count next
    u32 cslot[LZ_S2_ROUNDS];
    auto load_tile = [&](u32 tt, u64* x) {                      // (sorters) the wave's 256 records of tile tt, 64 per round
#pragma unroll
        for (int rr = 0; rr < LZ_S2_ROUNDS; rr++) {
            const u64 li = (u64)tt * LZ_S2_TILE + sw * (64u * LZ_S2_ROUNDS) + (u32)rr * 64u + lane;
            x[rr] = (tt < ntiles && li < (u64)n) ? LZ_NT_LD(recs + (size_t)r0 + li) : ~0ull;       // ~0: no record
        }
    };
    auto count_tile = [&](u32 tt, const u64* x, u32* slot) {    // ranks inside (wave, bucket) -> slot[], totals -> cnt[tt & 1][sw][]
        u32* const row = sh.cnt[tt & 1u][sw];
        reinterpret_cast<uint4*>(row)[lane] = make_uint4(0u, 0u, 0u, 0u);
        u64* const bmr = sh.bm[sw];
#pragma unroll
        for (int rr = 0; rr < LZ_S2_ROUNDS; rr++) {
            const bool valid = x[rr] != ~0ull;
            const u32 b = LZ_REC_LOW8(x[rr]);
            if (valid) atomicOr((unsigned long long*)&bmr[b], 1ull << lane);
            const u64 peers = valid ? bmr[b] : 0ull;
            const u32 old = valid ? row[b] : 0u;
            const u32 rank = (u32)__popcll(peers & ((1ull << lane) - 1ull));
            if (valid && (peers >> lane) == 1ull) { row[b] = old + (u32)__popcll(peers); bmr[b] = 0ull; }   // the highest peer
            slot[rr] = old + rank;
        }
    };
    auto place_tile = [&](u32 tt, const u64* x, const u32* slot) {
        // bucket bases of the tile: every sorter wave sums the 12 rows for its own use (lane l: buckets 4l .. 4l+3)
        const u32 (*cn)[LZ_NBIN] = sh.cnt[tt & 1u];
        uint4 tot = make_uint4(0u, 0u, 0u, 0u), pre = tot;
#pragma unroll
        for (u32 k = 0; k < LZ_S2_SORTW; k++) {
            const uint4 v = reinterpret_cast<const uint4*>(cn[k])[lane];
            if (k == sw) pre = tot;
            tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
        }
        const u32 mine = tot.x + tot.y + tot.z + tot.w;
        u32 inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if ((int)lane >= d) inc += t; }
        const u32 b0 = inc - mine, b1 = b0 + tot.x, b2 = b1 + tot.y, b3 = b2 + tot.z;
        reinterpret_cast<uint4*>(sh.woff[sw])[lane] = make_uint4(b0 + pre.x, b1 + pre.y, b2 + pre.z, b3 + pre.w);
        if (sw == 0) {
            reinterpret_cast<uint4*>(sh.lbeg[tt & 1u])[lane] = make_uint4(b0, b1, b2, b3);
            reinterpret_cast<uint4*>(sh.lcnt[tt & 1u])[lane] = tot;
        }
        u64* const dst = sh.rec[tt & 1u];
#pragma unroll
        for (int rr = 0; rr < LZ_S2_ROUNDS; rr++)
            if (x[rr] != ~0ull) dst[sh.woff[sw][LZ_REC_LOW8(x[rr])] + slot[rr]] = x[rr];
    };

    // ---- the two roles run their own loops (the register allocator sees two disjoint sets of live values) and
    // meet at the same barriers: 3 in the prologue, one per tile.  (s_barrier counts arrivals per wave, not places.)
    __syncthreads();                                            // tab, bm
    if (!walker) {
        // prologue: tile 0 counted | tile 0 placed, tile 1 counted | then per interval t: tile t+1 placed, tile t+2 counted
        load_tile(0, cx); load_tile(1, nx); count_tile(0, cx, cslot);
        __syncthreads();
        place_tile(0, cx, cslot);
#pragma unroll
        for (int rr = 0; rr < LZ_S2_ROUNDS; rr++) cx[rr] = nx[rr];
        load_tile(2, nx);
        count_tile(1, cx, cslot);
        __syncthreads();
        for (u32 t = 0; t < ntiles; t++) {
            LZ_S2_CLK_BEGIN(sw == 0 && lane == 0);
            if (t + 1 < ntiles) place_tile(t + 1, cx, cslot);
#pragma unroll
            for (int rr = 0; rr < LZ_S2_ROUNDS; rr++) cx[rr] = nx[rr];
            load_tile(t + 3, nx);                               // (past the last tile: no loads, "no record")
            if (t + 2 < ntiles) count_tile(t + 2, cx, cslot);
            LZ_S2_CLK_MID(sw == 0 && lane == 0, 18);
            __syncthreads();
            LZ_S2_CLK_END(sw == 0 && lane == 0, 19);
        }
        return;
    }
    __syncthreads();
    __syncthreads();
#if LZ_S2_WALK_PRIO
    __builtin_amdgcn_s_setprio(LZ_S2_WALK_PRIO);                // the walk is the critical path of a tile: its wave goes first on its SIMD
#endif
    constexpr u32 BPW = LZ_NBIN / LZ_S2_WALKW;                   // buckets (= walking lanes) per walking wave
    const bool wl = lane < BPW;
    const u32 bucket = (w * BPW + lane) & (LZ_NBIN - 1);        // the lane's bucket
    const u32 h = part * LZ_NBIN + bucket;
    const u32 L = P.seed_len;
    u32 dend = wl ? diag_end[h] : 0u;
    u64 n_ext = 0, n_bp = 0;
    for (u32 t = 0; t < ntiles; t++) {
        LZ_S2_CLK_BEGIN(tid == 0);
        {
            const u64* const rec = sh.rec[t & 1u];
            u32 p = wl ? sh.lbeg[t & 1u][bucket] : 0u; const u32 end = wl ? p + sh.lcnt[t & 1u][bucket] : 0u;
            u32 ne = 0, nb = 0;
            for (;;) {
                // every lane settles records from their phase-A summaries, LZ_ST_BATCH at a time, until one needs a
                // real extension
                bool pending = false; u32 pp2 = 0, ppay = 0;
                while (__ballot(p < end && !pending)) {         // straight-line batches: a lane that has stopped idles through selects
                    u64 r[LZ_ST_BATCH];
#pragma unroll
                    for (int k = 0; k < LZ_ST_BATCH; k++) r[k] = rec[p + k];            // (reads past `end` stay inside rec[] and are not used)
                    bool live = !pending;
                    u32 np = 0;
#pragma unroll
                    for (int k = 0; k < LZ_ST_BATCH; k++) {
                        const u32 p2 = LZ_REC_POS2(r[k]), pay = LZ_REC_PAYLOAD(r[k]);
                        const bool in = live && (p + (u32)k < end);
                        const bool drop = dend > p2 - L;                           // :1113
                        const bool slow = LZ_REC_SLOW(r[k]) != 0 && !drop;
                        const bool go = in && !slow;                               // the record is consumed here
                        const bool fast = go && !drop;
                        const u32 room = p2 - dend, dlo = pay & 0xFFu, dext = pay >> 8;
                        const u32 extent = p2 + dext;                              // :2785
                        ne += fast ? 1u : 0u;
                        nb += fast ? (dlo < room ? dlo : room) + dext : 0u;        // :2818
                        dend = (fast && extent > dend) ? extent : dend;
                        np += go ? 1u : 0u;
                        if (in && slow) { pending = true; pp2 = p2; ppay = pay; }
                        live = go;
                    }
                    p += np;
                }
                u64 mask = __ballot(pending);
                if (!mask) break;
                LZ_S2_CLK_EXT_BEGIN(tid == 0, mask);
                while (mask) {                                  // the wave extends the pending hits, one at a time
                    const int src = (int)__ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const u32 sp2 = (u32)__shfl((int)pp2, src), spay = (u32)__shfl((int)ppay, src), sdend = (u32)__shfl((int)dend, src);
                    const u32 sh_ = part * LZ_NBIN + w * BPW + (u32)src;
                    const s32 diag = (s32)((spay << 16) | sh_);
                    const u32 pos1 = sp2 + (u32)diag;
                    s32 stopl = (s32)sdend + diag;  if (stopl < 0) stopl = 0;                                     // :2612-2616
                    const s32 stopr = ((s32)P.tlen <= (s32)P.qlen + diag) ? (s32)P.tlen : (s32)P.qlen + diag;     // :2675-2677
                    const LzCoopSide S = lz_coop_extend_wave(P, sh.tab, pos1, diag, stopl, stopr, lane);
                    const u32 l_stop = (u32)__shfl((int)S.stop_pos, 0), l_bpos = (u32)__shfl((int)S.best_pos, 0); const s32 l_best = __shfl(S.best, 0);
                    const u32 r_stop = (u32)__shfl((int)S.stop_pos, 32), r_bpos = (u32)__shfl((int)S.best_pos, 32); const s32 r_best = __shfl(S.best, 32);
                    if ((int)lane == src) {
                        ne++; nb += r_stop - l_stop;                                             // :2818
                        const u32 extent = (u32)((s32)r_stop - diag);                            // :2785
                        if (extent > dend) dend = extent;
                        const s32 sim = l_best + r_best;
                        if (sim >= P.min_score) {
                            const u32 oslot = atomicAdd(out_count, 1u);
                            if (oslot < out_cap) { LzHspRec o; o.seed_pos1 = pos1; o.seed_pos2 = sp2; o.end1 = r_bpos; o.length = r_bpos - l_bpos; o.score = sim; out[oslot] = o; }
                        }
                        p++;
                    }
                }
                LZ_S2_CLK_EXT_END(tid == 0);
            }
            n_ext += ne; n_bp += nb;
        }
        LZ_S2_CLK_MID(tid == 0, 16);
        __syncthreads();
        LZ_S2_CLK_END(tid == 0, 17);
    }
    if (wl) diag_end[h] = dend;
    for (int o = 32; o > 0; o >>= 1) { n_ext += __shfl_down(n_ext, o); n_bp += __shfl_down(n_bp, o); }
    if (lane == 0) {
        if (n_ext) atomicAdd((unsigned long long*)&counters[0], (unsigned long long)n_ext);
        if (n_bp)  atomicAdd((unsigned long long*)&counters[1], (unsigned long long)n_bp);
    }
}

int lzk_settle(LzCtx& c, const LzExtendParams& P, const u64* recs, const u32* bin_base, u32* diag_end,
               const s32* score_tab, LzHspRec* out, u32* out_count, u32 out_cap, u64* counters, hipStream_t s)
{
    c.timer.begin("k_settle2", s);
    static bool attr_set = false;
    if (!attr_set) { LZ_HIP(hipFuncSetAttribute((const void*)k_settle2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LzSettle2Shared))); attr_set = true; }
    hipLaunchKernelGGL(k_settle2, dim3(LZ_NBIN), dim3(LZ_S2_TPB), sizeof(LzSettle2Shared), s, P, recs, bin_base, diag_end, score_tab, out, out_count, out_cap, counters);
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


