// window_kernels.hip -- SURVEY 8(f) N3: the seed stage of MANY SMALL PROBLEMS as one grid.
//
// lastz --inner=<score> (src/tweener.c:769-829, bounded_align) takes every gap between two neighbouring gapped
// alignments -- a rectangle of at most 20 kbp x 20 kbp -- and runs the whole hot path on it with a more sensitive
// seed (an exact 7-mer: 16 K words): build_seed_position_table on the target piece, seed_hit_search of the query
// piece, chaining, gapped_extend.  Thousands of such windows per (query, strand) at configs[4]'s size, each far too
// small to fill a GPU by itself (the device pipeline of the main search needs ~20 launches).  Here a window is
// ONE WORKGROUP: its position table, its diagonal state and its hits never leave the CU.
//
//   table      the words of the target piece are counted in LDS (16 K counters), scanned, and the positions dealt
//              out to their words' lists (LDS, u16).  The order inside a list is free: two hits of one query position
//              lie on different diagonals and a window's diagonals cannot collide (t_len + q_len <= 65536), so the
//              hits of a query position never interact (src/seed_search.c:1081-1126) -- only the ORDER OF REPORTING
//              depends on it, and the host sorts the few HSPs of a window by (query position, -target position).
//   search     256 query positions at a time: their hits (one u32 key = diagonal | query position each) go to an LDS
//              buffer, a bitonic sort brings the hits of a diagonal together in query order, and every diagonal of
//              the batch is walked by one lane with the reference's per-hit logic -- the diagEnd test (:1113), the two
//              X-drop loops with the left stop at diagEnd (lz_reextend, lz_common.hpp), the extent written back
//              (:2785-2789).  Batches follow each other in query order, so a diagonal sees its hits in the order
//              the reference's loop produces them.
//
// The scoring classes, X-drop and threshold are those of the search the windows belong to (maskedScoring,
// hp.xDrop, the inner threshold as an 'S' threshold, no entropy: src/tweener.c:300-317).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "lz_ctx.hpp"
#include "lz_host.hpp"

#define LZ_WIN_TPB    256
#define LZ_WIN_MAXLEN 20480          // bases per side of a window (the reference's default windows: <= 20,000)
#define LZ_WIN_WORDS  16384          // seed weight <= 14 bits (the inner seed: 7 bases)
#define LZ_WIN_HITCAP 4096           // hits sorted together
#define LZ_WIN_NDIAG  (2 * LZ_WIN_MAXLEN)

struct LzWinJob { u32 t_off, t_len, q_off, q_len; };
struct LzWinHsp { u32 win; LzHspRec r; };

struct LzWinShared {
    s32 tab[LZ_NCLASS * LZ_NCLASS];                  // score classes (masked scoring)
    union {
        u32 cnt[LZ_WIN_WORDS + 1];                   // table build: words' counts, then running cursors
        unsigned short dend[LZ_WIN_NDIAG];           // search: diagEnd per diagonal (query coordinates; 0 = inactive or zero, :1097-1111)
    };
    unsigned short pos[LZ_WIN_MAXLEN];               // target end positions, grouped by word
    u32 buf[LZ_WIN_HITCAP];                          // the hits of a batch: diagonal << 16 | query position
    u32 part[LZ_WIN_TPB];                            // scan partials
    u32 tot;
};

// exclusive prefix over the workgroup (256 lanes), total in sh.tot
__device__ __forceinline__ u32 lz_win_exscan(LzWinShared& sh, u32 v)
{
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    u32 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if ((int)lane >= d) inc += t; }
    if (lane == 63) sh.part[w] = inc;
    __syncthreads();
    u32 pre = 0;
    for (u32 k = 0; k < w; k++) pre += sh.part[k];
    if (tid == LZ_WIN_TPB - 1) sh.tot = pre + inc;
    __syncthreads();
    return pre + inc - v;
}

__global__ void __launch_bounds__(LZ_WIN_TPB)
k_window_search(LzExtendParams P, LzSeedDev sd, const LzWinJob* __restrict__ wins, u32 n_wins, const s32* __restrict__ score_tab_g,
                unsigned short* __restrict__ start_scratch /* [gridDim.x][LZ_WIN_WORDS + 1] */,
                LzWinHsp* __restrict__ out, u32* __restrict__ out_count, u32 out_cap)
{
    extern __shared__ __align__(16) unsigned char lz_win_smem[];
    LzWinShared& sh = *reinterpret_cast<LzWinShared*>(lz_win_smem);
    const u32 tid = threadIdx.x;
    for (u32 k = tid; k < LZ_NCLASS * LZ_NCLASS; k += LZ_WIN_TPB) sh.tab[k] = score_tab_g[k];
    unsigned short* const start = start_scratch + (size_t)blockIdx.x * (LZ_WIN_WORDS + 1);
    const u32 L = (u32)sd.length, nwords = 1u << sd.weight;
    for (u32 wi = blockIdx.x; wi < n_wins; wi += gridDim.x) {
        const LzWinJob J = wins[wi];
        LzExtendParams W = P;                                    // the window as a pair of whole sequences
        W.tcode = P.tcode + J.t_off; W.tlen = J.t_len; W.qcode = P.qcode + J.q_off; W.qlen = J.q_len;
        W.tnib = nullptr; W.qnib = nullptr; W.cls8 = 0;
        __syncthreads();
        // ---- the position table of the target piece (src/pos_table.c:396-476 for step 1: every position whose
        // window holds only A, C, G, T)
        for (u32 k = tid; k <= nwords; k += LZ_WIN_TPB) sh.cnt[k] = 0;
        __syncthreads();
        for (u32 p = L + tid; p <= J.t_len; p += LZ_WIN_TPB) {
            u32 w;
            if (lz_window_word(W.tcode, p, sd, w)) atomicAdd(&sh.cnt[w], 1u);
        }
        __syncthreads();
        {   // exclusive scan of the counts: lane t owns words [t * per, (t + 1) * per)
            const u32 per = (nwords + LZ_WIN_TPB - 1) / LZ_WIN_TPB, w0 = tid * per, w1 = (w0 + per < nwords) ? w0 + per : nwords;
            u32 s = 0;
            for (u32 w = w0; w < w1; w++) s += sh.cnt[w];
            u32 run = lz_win_exscan(sh, s);
            for (u32 w = w0; w < w1; w++) { const u32 c = sh.cnt[w]; sh.cnt[w] = run; start[w] = (unsigned short)run; run += c; }
            if (tid == LZ_WIN_TPB - 1) start[nwords] = (unsigned short)sh.tot;
        }
        __syncthreads();
        for (u32 p = L + tid; p <= J.t_len; p += LZ_WIN_TPB) {
            u32 w;
            if (lz_window_word(W.tcode, p, sd, w)) sh.pos[atomicAdd(&sh.cnt[w], 1u)] = (unsigned short)p;
        }
        __threadfence_block();
        __syncthreads();
        for (u32 k = tid; k < LZ_WIN_NDIAG; k += LZ_WIN_TPB) sh.dend[k] = 0;         // empty_diag_hash, :362 (the counters are done with)
        __syncthreads();

        // ---- the search, up to 256 query positions (seed end positions L .. q_len) at a time
        auto one_hit = [&](u32 dg, u32 q2, u32& dend) {           // process_for_simple_hit + xdrop_extend_seed_hit for one hit
            if (dend > q2 - L) return;                           // :1113
            u64 nbp = 0;
            dend = lz_reextend(W, sh.tab, q2, (s32)dg - (s32)J.q_len, dend, nbp, [&](const LzHspRec& r) {
                const u32 slot = atomicAdd(out_count, 1u);
                if (slot < out_cap) { LzWinHsp o; o.win = wi; o.r = r; out[slot] = o; }
            });
        };
        for (u32 base = L; base <= J.q_len; ) {
            const u32 p2 = base + tid;
            u32 a = 0, n = 0, w;
            if (p2 <= J.q_len && lz_window_word(W.qcode, p2, sd, w)) { a = start[w]; n = (u32)start[w + 1] - a; }
            const u32 off = lz_win_exscan(sh, n);
            u32 take = LZ_WIN_TPB, total = sh.tot;               // lanes [0, take) form this batch, `total` hits
            __syncthreads();
            if (total > LZ_WIN_HITCAP) {                         // (uniform) not all 256 lists fit the buffer: the longest prefix that does
                if (tid == 0) { sh.part[0] = LZ_WIN_TPB; }
                __syncthreads();
                if (off + n > LZ_WIN_HITCAP) atomicMin(&sh.part[0], tid);
                __syncthreads();
                take = sh.part[0];
                __syncthreads();
                if (tid == take) sh.part[1] = off;               // hits of the lanes below `take`
                if (take == 0 && tid == 0) { sh.part[2] = a; sh.part[3] = n; }
                __syncthreads();
                total = sh.part[1];
                if (take == 0) {
                    // the list of ONE query position is longer than the buffer (a repeat): its hits lie on different
                    // diagonals, so they do not interact and need no order -- straight from the list, no sort
                    const u32 a0 = sh.part[2], n0 = sh.part[3];
                    for (u32 k = tid; k < n0; k += LZ_WIN_TPB) {
                        const u32 dg = (u32)sh.pos[a0 + k] + J.q_len - base;
                        u32 dend = sh.dend[dg];
                        one_hit(dg, base, dend);
                        sh.dend[dg] = (unsigned short)dend;
                    }
                    __syncthreads();
                    base += 1;
                    continue;
                }
                __syncthreads();
            }
            // keys: diagonal (pos1 - pos2 + q_len) << 16 | pos2
            if (tid < take)
                for (u32 k = 0; k < n; k++) sh.buf[off + k] = (((u32)sh.pos[a + k] + J.q_len - p2) << 16) | p2;
            u32 npow = 1; while (npow < total) npow <<= 1;
            for (u32 k = total + tid; k < npow; k += LZ_WIN_TPB) sh.buf[k] = 0xFFFFFFFFu;
            __syncthreads();
            // bitonic sort of buf[0, npow): the hits of a diagonal end up next to each other, in query order
            for (u32 kk = 2; kk <= npow; kk <<= 1)
                for (u32 jj = kk >> 1; jj > 0; jj >>= 1) {
                    for (u32 i = tid; i < npow; i += LZ_WIN_TPB) {
                        const u32 ixj = i ^ jj;
                        if (ixj > i) {
                            const u32 x = sh.buf[i], y = sh.buf[ixj];
                            const bool up = (i & kk) == 0;
                            if ((x > y) == up) { sh.buf[i] = y; sh.buf[ixj] = x; }
                        }
                    }
                    __syncthreads();
                }
            // every diagonal of the batch is walked by one lane
            for (u32 i = tid; i < total; i += LZ_WIN_TPB) {
                const u32 dg = sh.buf[i] >> 16;
                if (i > 0 && (sh.buf[i - 1] >> 16) == dg) continue;              // not the head of its run
                u32 dend = sh.dend[dg];
                for (u32 j = i; j < total && (sh.buf[j] >> 16) == dg; j++) one_hit(dg, sh.buf[j] & 0xFFFFu, dend);
                sh.dend[dg] = (unsigned short)dend;
            }
            __syncthreads();
            base += take;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
static DevBuf g_win_jobs, g_win_scratch, g_win_out, g_win_count;
void lz_win_release_statics() { g_win_jobs.release(); g_win_scratch.release(); g_win_out.release(); g_win_count.release(); }

// the windows' HSPs in the reference's reporting order per window; counts[k] = HSPs of window k
int lzk_window_search(LzCtx& c, const LzExtendParams& P, const LzSeedDev& sd, const lz_window* wins, u32 n, const s32* score_tab_dev,
                      std::vector<lz_hsp>& out, std::vector<u32>& counts)
{
    out.clear(); counts.assign(n, 0);
    if (n == 0) return 0;
    int rc;
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c.device);
    const u32 grid = std::min<u32>(n, (u32)cus);
    std::vector<LzWinJob> jobs(n);
    for (u32 k = 0; k < n; k++) { jobs[k].t_off = wins[k].t_off; jobs[k].t_len = wins[k].t_len; jobs[k].q_off = wins[k].q_off; jobs[k].q_len = wins[k].q_len; }
    if ((rc = g_win_jobs.ensure((size_t)n * sizeof(LzWinJob)))) return rc;
    if ((rc = g_win_scratch.ensure((size_t)grid * (LZ_WIN_WORDS + 1) * 2))) return rc;
    if ((rc = g_win_count.ensure(4))) return rc;
    LZ_HIP(hipMemcpyAsync(g_win_jobs.p, jobs.data(), (size_t)n * sizeof(LzWinJob), hipMemcpyHostToDevice, c.stream));
    static bool attr_set = false;
    if (!attr_set) { LZ_HIP(hipFuncSetAttribute((const void*)k_window_search, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LzWinShared))); attr_set = true; }
    u32 cap = std::max<u32>(64u * n, 4096u);
    for (;;) {
        if ((rc = g_win_out.ensure((size_t)cap * sizeof(LzWinHsp)))) return rc;
        LZ_HIP(hipMemsetAsync(g_win_count.p, 0, 4, c.stream));
        c.timer.begin("k_window_search", c.stream);
        hipLaunchKernelGGL(k_window_search, dim3(grid), dim3(LZ_WIN_TPB), sizeof(LzWinShared), c.stream,
                           P, sd, g_win_jobs.as<LzWinJob>(), n, score_tab_dev, g_win_scratch.as<unsigned short>(),
                           g_win_out.as<LzWinHsp>(), g_win_count.as<u32>(), cap);
        c.timer.end(c.stream);
        LZ_HIP(hipGetLastError());
        u32 got = 0;
        LZ_HIP(hipMemcpyAsync(&got, g_win_count.p, 4, hipMemcpyDeviceToHost, c.stream));
        LZ_HIP(hipStreamSynchronize(c.stream));
        c.timer.resolve();
        if (got <= cap) {
            std::vector<LzWinHsp> recs(got);
            if (got) LZ_HIP(hipMemcpy(recs.data(), g_win_out.p, (size_t)got * sizeof(LzWinHsp), hipMemcpyDeviceToHost));
            // reporting order inside a window: query position up, target position down (:506-512, :832)
            std::sort(recs.begin(), recs.end(), [](const LzWinHsp& x, const LzWinHsp& y) {
                if (x.win != y.win) return x.win < y.win;
                if (x.r.seed_pos2 != y.r.seed_pos2) return x.r.seed_pos2 < y.r.seed_pos2;
                return x.r.seed_pos1 > y.r.seed_pos1;
            });
            out.reserve(got);
            for (const LzWinHsp& h : recs) {
                const s32 diag = (s32)h.r.seed_pos1 - (s32)h.r.seed_pos2;
                out.push_back({ h.r.end1, (u32)((s32)h.r.end1 - diag), h.r.length, h.r.score });
                counts[h.win]++;
            }
            return 0;
        }
        cap = got + got / 8 + 1024;                             // (more HSPs than room: once more with room for all)
    }
}
