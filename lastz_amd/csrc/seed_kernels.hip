// seed_kernels.hip -- gfx950 kernels of the seed stage: position-table build (B1), raw-hit
// enumeration, bucket ordering and the bucket-serial X-drop extender (B2).
//
// Decomposition: the only cross-hit state of the reference's HSP search is
// diagEnd[hashedDiag] (src/seed_search.c:1081-1126, 2612-2616, 2785-2789), so the exact
// parallel form is 65,536 independent, order-preserving streams.  Hits are enumerated in the
// reference's order (count -> scan -> fill gives every hit its discovery index; the table itself
// is probed in seed-word order), scanned independently of the hash and stably partitioned by the high
// 8 hash bits in the same pass (phase A, k_probe_part: three bases per step through an LDS look-up
// table on 2-bit codes, lz_lut.hpp), and each partition is then dealt out to its 256 buckets inside LDS,
// every bucket walked by one lane with diagEnd[h] in a register (phase B, k_settle).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include "lz_ctx.hpp"
#include "lz_lut.hpp"

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
        if (c & LZ_CODE_INVALID) m |= 1u << k; else bits |= LZ_CODE_BITS(c) << (2 * k);
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
// offsets (k_hist + two small scans), one pass that scans and scatters (k_probe_part): 8 B read + 8 B read +
// 8 B written per hit, where a radix sort of (key, summary) pairs moved 56.
#define LZ_PP_TPB    1024
#define LZ_PP_WAVES  (LZ_PP_TPB / 64)
#define LZ_PP_ROUNDS 4
#define LZ_PP_TILE   (LZ_PP_TPB * LZ_PP_ROUNDS)      // hits per tile
#define LZ_PP_QCAP   1024                            // unfinished scans a tile can queue (more are continued in place)
#define LZ_NBIN      256
#define LZ_KEY_BIN(k)  ((u32)((k) >> 40) & 0xFFu)    // bits 8..15 of hashedDiag

// hist[tile][bin]: hits of the tile per partition
__global__ void __launch_bounds__(LZ_TPB)
k_hist(const u64* __restrict__ keys, u64 n, u32* __restrict__ hist)
{
    __shared__ u32 cnt[LZ_NBIN];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * LZ_PP_TILE;
#pragma unroll
    for (int r = 0; r < LZ_PP_TILE / LZ_TPB; r++) {
        const u64 i = base + (u64)r * LZ_TPB + threadIdx.x;
        if (i < n) atomicAdd(&cnt[LZ_KEY_BIN(keys[i])], 1u);
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

int lzk_hist(LzCtx& c, const u64* keys, u64 n, u32* hist, u32* part, u32* bin_base)
{
    const u32 ntiles = (u32)((n + LZ_PP_TILE - 1) / LZ_PP_TILE), nblocks = (ntiles + 255u) / 256u;
    c.timer.begin("k_hist", c.stream);
    hipLaunchKernelGGL(k_hist, dim3(ntiles), dim3(LZ_TPB), 0, c.stream, keys, n, hist);
    c.timer.end(c.stream);
    c.timer.begin("k_hist_scan", c.stream);
    hipLaunchKernelGGL(k_hist_scan1, dim3(nblocks), dim3(LZ_NBIN), 0, c.stream, hist, ntiles, part);
    hipLaunchKernelGGL(k_hist_scan2, dim3(1), dim3(LZ_NBIN), 0, c.stream, part, nblocks, bin_base);
    c.timer.end(c.stream);
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

struct LzPPTask { u32 s; s32 run, best; u32 room, used, nwin; s32 diag; u32 side; };
struct LzPPShared {
    LzLutEntry lut[2 * LZ_LUT_ENTRIES];              // MODE 0/1: right table, left table (64 KiB)
    s32 m16[16];
    s32 tab[LZ_NCLASS * LZ_NCLASS]; s32 tab8[64];    // MODE 2: the byte-code scans' tables
    u64 stage[LZ_PP_TILE];                           // the tile's records, ordered by partition
    u8  sbin[LZ_PP_TILE];
    unsigned short wcnt[LZ_PP_WAVES][LZ_NBIN];       // per wave and partition: records so far / start inside the partition
    u32 tstart[LZ_NBIN + 1], gbase[LZ_NBIN], wtot[4];
    LzPPTask q[LZ_PP_QCAP]; u32 qn;
};

template <int MODE>      // 0: LUT scans, no special bytes in either sequence; 1: LUT scans + special masks; 2: byte-code scans
__global__ void __launch_bounds__(LZ_PP_TPB)
k_probe_part(LzExtendParams P, LzLutParams Q, const u64* __restrict__ keys, u64 n, u32 ntiles,
             const s32* __restrict__ score_tab_g, const LzLutEntry* __restrict__ lut_g, const s32* __restrict__ m16_g,
             const u32* __restrict__ hist, const u32* __restrict__ part, u64* __restrict__ recs)
{
    __shared__ LzPPShared sh;
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    if (MODE < 2) {
        for (u32 k = tid; k < 2 * LZ_LUT_ENTRIES; k += LZ_PP_TPB) sh.lut[k] = lut_g[k];
        if (tid < 16) sh.m16[tid] = m16_g[tid];
    } else {
        for (u32 k = tid; k < LZ_NCLASS * LZ_NCLASS; k += LZ_PP_TPB) sh.tab[k] = score_tab_g[k];
        if (tid < 64) sh.tab8[tid] = score_tab_g[(tid >> 3) * LZ_NCLASS + (tid & 7)];
    }
    const LzLutEntry* lut_r = sh.lut; const LzLutEntry* lut_l = sh.lut + LZ_LUT_ENTRIES;
    constexpr bool SP = MODE == 1;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const u64 base = (u64)tile * LZ_PP_TILE;
        const u32 tile_n = (n - base < (u64)LZ_PP_TILE) ? (u32)(n - base) : (u32)LZ_PP_TILE;
        if (tid == 0) sh.qn = 0;
        for (u32 k = tid; k < LZ_PP_WAVES * LZ_NBIN / 2; k += LZ_PP_TPB) reinterpret_cast<u32*>(&sh.wcnt[0][0])[k] = 0;
        __syncthreads();

        // ---- phase A heads: one window per side for every hit; unfinished scans are queued
        u64 key[LZ_PP_ROUNDS]; u32 summ[LZ_PP_ROUNDS];
        u32 usedl[LZ_PP_ROUNDS], usedr[LZ_PP_ROUNDS]; s32 bestl[LZ_PP_ROUNDS], bestr[LZ_PP_ROUNDS];
        int slotl[LZ_PP_ROUNDS], slotr[LZ_PP_ROUNDS]; u32 alive[LZ_PP_ROUNDS];
#pragma unroll
        for (int r = 0; r < LZ_PP_ROUNDS; r++) {
            const u32 li = w * (64u * LZ_PP_ROUNDS) + (u32)r * 64u + lane;
            const bool valid = li < tile_n;
            key[r] = valid ? keys[base + li] : 0ull;
            summ[r] = 0; usedl[r] = usedr[r] = 0; bestl[r] = bestr[r] = 0; slotl[r] = slotr[r] = -1; alive[r] = 0;
            if (MODE == 2) { if (valid) summ[r] = lz_probe_hit(P, sh.tab, sh.tab8, P.cls8 != 0, key[r]); continue; }
            s32 diag; LzLutScan L, R;
            lz_lut_init(key[r], P.tlen, P.qlen, diag, L, R);
            if (!valid) { L.alive = 0; R.alive = 0; }
            if (L.alive) lz_lut_window<false, SP>(Q, lut_l, sh.m16, diag, L);
            if (R.alive) lz_lut_window<true, SP>(Q, lut_r, sh.m16, diag, R);
            if (L.alive == 1) {
                const u32 slot = atomicAdd(&sh.qn, 1u);
                if (slot < LZ_PP_QCAP) { sh.q[slot] = { L.s, L.run, L.best, L.room, L.used, L.nwin, diag, 0u }; slotl[r] = (int)slot; }
                else while (L.alive == 1 && L.nwin < (u32)LZ_LUT_MAXWIN) lz_lut_window<false, SP>(Q, lut_l, sh.m16, diag, L);
            }
            if (R.alive == 1) {
                const u32 slot = atomicAdd(&sh.qn, 1u);
                if (slot < LZ_PP_QCAP) { sh.q[slot] = { R.s, R.run, R.best, R.room, R.used, R.nwin, diag, 1u }; slotr[r] = (int)slot; }
                else while (R.alive == 1 && R.nwin < (u32)LZ_LUT_MAXWIN) lz_lut_window<true, SP>(Q, lut_r, sh.m16, diag, R);
            }
            usedl[r] = L.used; usedr[r] = R.used; bestl[r] = L.best; bestr[r] = R.best;
            alive[r] = (L.alive ? 1u : 0u) | (R.alive ? 2u : 0u);
        }
        if (MODE < 2) {
            __syncthreads();
            // ---- the queued scans, one lane per scan, to the end (or the cap)
            const u32 nq = sh.qn < (u32)LZ_PP_QCAP ? sh.qn : (u32)LZ_PP_QCAP;
            for (u32 k = tid; k < nq; k += LZ_PP_TPB) {
                const LzPPTask t = sh.q[k];
                LzLutScan S; S.s = t.s; S.run = t.run; S.best = t.best; S.room = t.room; S.used = t.used; S.nwin = t.nwin; S.alive = 1;
                if (t.side) while (S.alive == 1 && S.nwin < (u32)LZ_LUT_MAXWIN) lz_lut_window<true, SP>(Q, lut_r, sh.m16, t.diag, S);
                else        while (S.alive == 1 && S.nwin < (u32)LZ_LUT_MAXWIN) lz_lut_window<false, SP>(Q, lut_l, sh.m16, t.diag, S);
                sh.q[k].used = S.used; sh.q[k].best = S.best; sh.q[k].side = S.alive;     // results travel back through the slot
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < LZ_PP_ROUNDS; r++) {
                if (slotl[r] >= 0) { const LzPPTask& t = sh.q[slotl[r]]; usedl[r] = t.used; bestl[r] = t.best; alive[r] = (alive[r] & ~1u) | (t.side ? 1u : 0u); }
                if (slotr[r] >= 0) { const LzPPTask& t = sh.q[slotr[r]]; usedr[r] = t.used; bestr[r] = t.best; alive[r] = (alive[r] & ~2u) | (t.side ? 2u : 0u); }
                u32 s = (usedl[r] & 0xFFu) | ((usedr[r] & 0xFFu) << 8);
                if (alive[r] || bestl[r] + bestr[r] >= P.min_score) s |= LZ_SUMM_SLOW;
                summ[r] = s;
            }
        }

        // ---- stable partition of the tile's records: rank inside (wave, partition), then the waves' counts are
        // chained in wave order (= discovery order: a wave holds 256 consecutive hits)
        u32 lrank[LZ_PP_ROUNDS], bin[LZ_PP_ROUNDS];
#pragma unroll
        for (int r = 0; r < LZ_PP_ROUNDS; r++) {
            const u32 li = w * (64u * LZ_PP_ROUNDS) + (u32)r * 64u + lane;
            const bool valid = li < tile_n;
            bin[r] = LZ_KEY_BIN(key[r]);
            u32 rank, count; bool last;
            lz_match8(bin[r], valid, lane, rank, count, last);
            const u32 old = sh.wcnt[w][bin[r]];
            if (valid && last) sh.wcnt[w][bin[r]] = (unsigned short)(old + count);
            lrank[r] = old + rank;
        }
        __syncthreads();
        u32 tot = 0;
        if (tid < LZ_NBIN) {
            for (u32 k = 0; k < LZ_PP_WAVES; k++) { const u32 v = sh.wcnt[k][tid]; sh.wcnt[k][tid] = (unsigned short)tot; tot += v; }
            sh.gbase[tid] = part[(size_t)(tile >> 8) * LZ_NBIN + tid] + hist[(size_t)tile * LZ_NBIN + tid];
        }
        const u32 ts = lz_exscan256(tot, sh.wtot);
        if (tid < LZ_NBIN) sh.tstart[tid] = ts;
        if (tid == 0) sh.tstart[LZ_NBIN] = tile_n;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < LZ_PP_ROUNDS; r++) {
            const u32 li = w * (64u * LZ_PP_ROUNDS) + (u32)r * 64u + lane;
            if (li < tile_n) {
                const u32 pos = sh.tstart[bin[r]] + sh.wcnt[w][bin[r]] + lrank[r];
                sh.stage[pos] = lz_hit_record(key[r], summ[r]);
                sh.sbin[pos] = (u8)bin[r];
            }
        }
        __syncthreads();
        for (u32 k = tid; k < tile_n; k += LZ_PP_TPB) {
            const u32 b = sh.sbin[k];
            recs[(size_t)sh.gbase[b] + (k - sh.tstart[b])] = sh.stage[k];
        }
        __syncthreads();
    }
}

int lzk_probe_part(LzCtx& c, int mode, const LzExtendParams& P, const LzLutParams& Q, const u64* keys, u64 n,
                   const s32* score_tab, const LzLutEntry* lut, const s32* m16, const u32* hist, const u32* part, u64* recs)
{
    if (n == 0) return 0;
    const u32 ntiles = (u32)((n + LZ_PP_TILE - 1) / LZ_PP_TILE);
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c.device);
    const u32 grid = ntiles < (u32)cus ? ntiles : (u32)cus;     // one 1024-thread workgroup per CU (the tables fill most of its LDS)
    c.timer.begin("k_probe_part", c.stream);
    if (mode == 0)      hipLaunchKernelGGL(k_probe_part<0>, dim3(grid), dim3(LZ_PP_TPB), 0, c.stream, P, Q, keys, n, ntiles, score_tab, lut, m16, hist, part, recs);
    else if (mode == 1) hipLaunchKernelGGL(k_probe_part<1>, dim3(grid), dim3(LZ_PP_TPB), 0, c.stream, P, Q, keys, n, ntiles, score_tab, lut, m16, hist, part, recs);
    else                hipLaunchKernelGGL(k_probe_part<2>, dim3(grid), dim3(LZ_PP_TPB), 0, c.stream, P, Q, keys, n, ntiles, score_tab, lut, m16, hist, part, recs);
    c.timer.end(c.stream);
    LZ_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// B2 step 4 (phase B): one workgroup per partition (256 buckets), one lane per bucket with diagEnd[h] in a
// register.  The partition's records arrive in discovery order; a tile of them is loaded with coalesced reads,
// dealt out to the buckets inside LDS (counting sort by the record's low hash bits: counts, offsets, placement),
// and every lane then walks its own short list in order (lz_settle_record).
#define LZ_ST_TPB    256
#define LZ_ST_TILE   2048
#define LZ_ST_ROUNDS (LZ_ST_TILE / LZ_ST_TPB)
__global__ void __launch_bounds__(LZ_ST_TPB)
k_settle(LzExtendParams P, const u64* __restrict__ recs, const u32* __restrict__ bin_base, u32* __restrict__ diag_end,
         const s32* __restrict__ score_tab_g, LzHspRec* __restrict__ out, u32* __restrict__ out_count, u32 out_cap,
         u64* __restrict__ counters)
{
    __shared__ s32 tab[LZ_NCLASS * LZ_NCLASS];
    __shared__ u64 rec[LZ_ST_TILE];
    __shared__ unsigned short list[LZ_ST_TILE];
    __shared__ u32 cnt[4][LZ_NBIN];
    __shared__ u32 wtot[4];
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    for (int k = tid; k < LZ_NCLASS * LZ_NCLASS; k += LZ_ST_TPB) tab[k] = score_tab_g[k];
    const u32 h = blockIdx.x * LZ_NBIN + tid;
    u32 dend = diag_end[h];
    u64 n_ext = 0, n_bp = 0;
    const u32 r0 = bin_base[blockIdx.x], r1 = bin_base[blockIdx.x + 1];
    auto emit = [&](const LzHspRec& r) { const u32 slot = atomicAdd(out_count, 1u); if (slot < out_cap) out[slot] = r; };
    // a wave owns a quarter of the tile (consecutive records), which it takes 64 at a time
    u64 x[LZ_ST_ROUNDS];
#pragma unroll
    for (int rr = 0; rr < LZ_ST_ROUNDS; rr++) { const u32 li = w * (LZ_ST_TILE / 4) + (u32)rr * 64u + lane; x[rr] = ((u64)r0 + li < (u64)r1) ? recs[(size_t)r0 + li] : 0ull; }
    for (u32 t0 = r0; t0 < r1; t0 += LZ_ST_TILE) {
        const u32 nt = (r1 - t0 < (u32)LZ_ST_TILE) ? r1 - t0 : (u32)LZ_ST_TILE;
#pragma unroll
        for (int k = 0; k < 4; k++) cnt[k][tid] = 0;
        __syncthreads();
        u32 slot[LZ_ST_ROUNDS], k8[LZ_ST_ROUNDS];
#pragma unroll
        for (int rr = 0; rr < LZ_ST_ROUNDS; rr++) {
            const u32 li = w * (LZ_ST_TILE / 4) + (u32)rr * 64u + lane;
            k8[rr] = LZ_REC_LOW8(x[rr]); slot[rr] = 0;
            if (li < nt) { rec[li] = x[rr]; slot[rr] = atomicAdd(&cnt[w][k8[rr]], 1u); }
        }
        __syncthreads();
        // offsets: bucket-major, then wave order inside a bucket
        const u32 c0 = cnt[0][tid], c1 = cnt[1][tid], c2 = cnt[2][tid], c3 = cnt[3][tid];
        const u32 mine = c0 + c1 + c2 + c3;
        const u32 beg = lz_exscan256(mine, wtot);
        cnt[0][tid] = beg; cnt[1][tid] = beg + c0; cnt[2][tid] = beg + c0 + c1; cnt[3][tid] = beg + c0 + c1 + c2;
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < LZ_ST_ROUNDS; rr++) {
            const u32 li = w * (LZ_ST_TILE / 4) + (u32)rr * 64u + lane;
            if (li < nt) list[cnt[w][k8[rr]] + slot[rr]] = (unsigned short)li;
        }
        __syncthreads();
        // the next tile's records are requested before this one is walked
        if (t0 + LZ_ST_TILE < r1) {
#pragma unroll
            for (int rr = 0; rr < LZ_ST_ROUNDS; rr++) { const u32 li = w * (LZ_ST_TILE / 4) + (u32)rr * 64u + lane; x[rr] = ((u64)t0 + LZ_ST_TILE + li < (u64)r1) ? recs[(size_t)t0 + LZ_ST_TILE + li] : 0ull; }
        }
        // Records a wave placed in the same step may sit in any order among themselves (the slots come from
        // LDS atomics): the list is put in ascending tile order, which is discovery order, by an insertion pass
        // (it is sorted already but for such neighbours).
        for (u32 p = beg + 1; p < beg + mine; p++) {
            const unsigned short v = list[p];
            u32 q = p;
            while (q > beg && list[q - 1] > v) { list[q] = list[q - 1]; q--; }
            list[q] = v;
        }
        for (u32 p = beg; p < beg + mine; p++)
            lz_settle_record(P, tab, rec[list[p]], h, dend, n_ext, n_bp, emit);
        __syncthreads();
    }
    diag_end[h] = dend;
    for (int o = 32; o > 0; o >>= 1) { n_ext += __shfl_down(n_ext, o); n_bp += __shfl_down(n_bp, o); }
    if (lane == 0) {
        if (n_ext) atomicAdd((unsigned long long*)&counters[0], (unsigned long long)n_ext);
        if (n_bp)  atomicAdd((unsigned long long*)&counters[1], (unsigned long long)n_bp);
    }
}

int lzk_settle(LzCtx& c, const LzExtendParams& P, const u64* recs, const u32* bin_base, u32* diag_end,
               const s32* score_tab, LzHspRec* out, u32* out_count, u32 out_cap, u64* counters, hipStream_t s)
{
    c.timer.begin("k_settle", s);
    hipLaunchKernelGGL(k_settle, dim3(LZ_NBIN), dim3(LZ_ST_TPB), 0, s, P, recs, bin_base, diag_end, score_tab, out, out_count, out_cap, counters);
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


