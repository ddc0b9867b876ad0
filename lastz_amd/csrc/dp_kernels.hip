// dp_kernels.hip -- B3 on gfx950: the Y-drop one-sided DP kernel (four waves per DP, sweep row in
// LDS, traceback bytes in HBM), the HIP executor that feeds it batches of speculative DPs, and the
// lzgpu_gapped_extend entry point.  The algorithm itself is in lz_dp_dev.hpp (shared with the
// CPU phase emulator used by the no-GPU tests); the anchor ordering / commit logic is in
// lz_gapped_host.cpp.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <vector>
#include <chrono>
#include <algorithm>
#include <map>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <atomic>
#include <system_error>
#include "lz_ctx.hpp"
#include "lz_host.hpp"
#include "lz_gapped_host.hpp"

#include "dp_kernels_dev.inc"

// dp_kernels_narrow.hip: the same kernel on two waves per DP with the 16-bit sweep row
int lzk_ydrop_narrow(bool no_trim, bool bounds, bool repl, unsigned n, size_t dyn_lds, hipStream_t st, const LzDpProblem* problems, const LzDpParams& P,
                     const LzDpJob* jobs, const u32* job_ids, const s32* tab, LzDpResult* res, u32 tab_rows);
unsigned lzk_ydrop_narrow_per_cu(bool bounds);

// The same DP with its sweep-row ring in an HBM slot: bands the LDS ring cannot hold (LZ_DP_TOO_WIDE from k_ydrop)
template <bool NOTRIM, bool BOUNDS>
__global__ void __launch_bounds__(LZ_DP_LANES)
k_ydrop_wide(const LzDpProblem* __restrict__ problems, LzDpParams P, const LzDpJob* __restrict__ jobs, const u32* __restrict__ job_ids,
             const s32* __restrict__ tab_g, LzDpResult* __restrict__ res, u8* __restrict__ rings)
{
    __shared__ LzDpSharedWide sh;
    __shared__ s32 tab[LZ_NCLASS * LZ_NCLASS];
    for (int k = threadIdx.x; k < LZ_NCLASS * LZ_NCLASS; k += LZ_DP_LANES) tab[k] = tab_g[k];
    if (threadIdx.x == 0) sh.bind(rings + (size_t)blockIdx.x * LzDpRingHbm::SLOT_BYTES);
    __syncthreads();
    const u32 j = job_ids[blockIdx.x];
    GpuPhases x;
    x.lead_wave = (int)(blockIdx.x & (LZ_DP_WAVES - 1));
    const LzDpJob J = jobs[j];
    const LzDpProblem pb = problems[J.problem];
    P.qdp = pb.qdp; P.qlen = pb.qlen; P.tdp = pb.tdp; P.tlen = pb.tlen;
    lz_dp_run<NOTRIM, BOUNDS, false>(x, sh, P, J, tab, &res[j]);
}

// gather the edit ops of a batch into one contiguous buffer (one block per job)
__global__ void __launch_bounds__(256)
k_gather_ops(const LzDpJob* __restrict__ jobs, const LzDpResult* __restrict__ res, const u32* __restrict__ ops_arena,
             const u64* __restrict__ dst_off, u32* __restrict__ dst)
{
    const u32 j = blockIdx.x;
    const u32 n = res[j].n_ops;
    const u32* src = ops_arena + jobs[j].ops_off;
    u32* d = dst + dst_off[j];
    for (u32 k = threadIdx.x; k < n; k += 256) d[k] = src[k];
}

struct DpBufs {
    DevBuf jobs, ids, res, tab, tb, rows, ops, ops_off, ops_out, pieces, rings, sel_jobs, sel_res, problems;
};
static DpBufs g_dp;
void lz_dp_release_statics()
{
    DevBuf* b[] = { &g_dp.jobs, &g_dp.ids, &g_dp.res, &g_dp.tab, &g_dp.tb, &g_dp.rows, &g_dp.ops, &g_dp.ops_off, &g_dp.ops_out, &g_dp.pieces, &g_dp.rings, &g_dp.sel_jobs, &g_dp.sel_res, &g_dp.problems };
    for (DevBuf* x : b) x->release();
}

static u64 g_dp_longest[4] = { 0, 0, 0, 0 };
extern "C" int lzgpu_dp_longest(uint64_t out[4], int reset)
{
    if (out) for (int k = 0; k < 4; k++) out[k] = g_dp_longest[k];
    if (reset) for (int k = 0; k < 4; k++) g_dp_longest[k] = 0;
    return 0;
}

struct HipDpExec : LzDpExecutor {
    LzCtx& c;
    LzDpParams P;                        // arenas filled per launch
    u32 slot_tb;                         // first-try traceback slot (bytes) per DP
    u64 dp_launch_cells = 0;
    explicit HipDpExec(LzCtx& ctx) : c(ctx) {}

    u64 wide_runs = 0;
    double t_upload = 0, t_kernel = 0, t_ops = 0;               // LZGPU_HOSTPROF: host milliseconds in launch() / fetch_ops()
    u32 tab_rows = LZ_NCLASS;                                  // row classes of the score matrix in use (k_ydrop's dynamic LDS)
    bool row16_ok = false;                                      // the scoring fits the 16-bit sweep row (lz_dp_row16_ok): the two-wave kernel may be used
    u64 jobs_narrow = 0;
    const LzDpProblem* problems_dev = nullptr;                  // the launch's problems (run_multi)
    const std::vector<LzDpBatchItem>* cur_items = nullptr;     // ... and their snapshots on the host: what a job's pieces are worked out from
    std::vector<u32> horizon;                                   // per job of the run: rows its pieces are asked for (grows when a sweep passes it)
    u64 piece_reruns = 0, jobs_bounded = 0, jobs_free = 0;
    // Traceback slots.  A retry gives every DP the same (larger) slot; the first try sizes each DP's slot from the
    // host's guess of the rows it will sweep (est_rows, lz_gapped_host.cpp: the deferred anchors around the DP's
    // anchor): on the bench pair rows = 1.2 x est_rows (median; 2.2 x at the 90th percentile) and a row has ~430
    // cells, so 1000 bytes per estimated row with a floor of 2 MiB held every one of 4602 DPs while the arena shrank
    // from 27.5 to 14.4 GiB per launch (uniform 8 MiB slots) -- memory a fresh process pays ~25 ms per GiB for.
    int launch(std::vector<LzDpJob>& jobs, const std::vector<u32>& ids, u32 slot,
               std::vector<LzDpResult>& res, bool wide = false, bool by_estimate = false)
    {
        // slots: tb = slot bytes, rows = slot/16 entries, ops = slot/32 entries (all per DP)
        const auto lt0 = std::chrono::steady_clock::now();
        const u64 n = ids.size();
        int rc;
        u64 tb_total = 0, row_total = 0, ops_total = 0;
        for (u64 k = 0; k < n; k++) {
            LzDpJob& J = jobs[ids[k]];
            u64 sl = slot;
            if (by_estimate) {
                sl = std::max<u64>((u64)J.est_rows * 1000u, 2u << 20);
                sl = std::min<u64>((sl + 65535u) & ~65535ull, slot);
            }
            const u32 row_cap = (u32)(sl / 16 + 64), ops_cap = (u32)(sl / 32 + 64);
            J.tb_off = tb_total; J.tb_cap = (u32)sl;        tb_total += sl;
            J.row_off = row_total; J.row_cap = row_cap;     row_total += row_cap;
            J.ops_off = ops_total; J.ops_cap = ops_cap;     ops_total += ops_cap;
        }
        // The jobs' pieces (lz_dp_pieces.cpp): what earlier alignments mean for each sweep -- its left and right bound and the cells it
        // must mask, as run-length pieces of rows -- worked out here, on the host, from the job's problem's snapshot, up to the job's
        // horizon.  A job with no piece at all and none to come (the first round of a strand; a DP far from every alignment) needs no
        // bound logic: those go to the kernel without it (no mask stamps in its ring: seven DPs per CU), the others to the one with it.
        std::vector<LzDpPiece> arena;
        std::vector<u32> ids_free, ids_bound;
        {
            LzDpPieces pcs;
            for (u64 k = 0; k < n; k++) {
                LzDpJob& J = jobs[ids[k]];
                lzh_dp_pieces(*(*cur_items)[J.problem].snap, J, horizon[ids[k]], pcs);
                J.pc_off = arena.size(); J.n_lb = (u32)pcs.lb.size(); J.n_rb = (u32)pcs.rb.size(); J.n_mk = (u32)pcs.mk.size();
                J.horizon = pcs.complete ? 0xFFFFFFFFu : horizon[ids[k]];
                arena.insert(arena.end(), pcs.lb.begin(), pcs.lb.end()); arena.insert(arena.end(), pcs.rb.begin(), pcs.rb.end()); arena.insert(arena.end(), pcs.mk.begin(), pcs.mk.end());
                const bool free_job = pcs.complete && !J.n_lb && !J.n_rb && !J.n_mk;
                (free_job && !wide ? ids_free : ids_bound).push_back(ids[k]);
            }
            if (arena.size() > (1u << 27)) return LZGPU_NH_UNSUPPORTED;          // (2 GiB of pieces for one launch: not a workload this path is for)
        }
        jobs_free += ids_free.size(); jobs_bounded += ids_bound.size();
        // a quarter of head room: the launches of a run (two strands, the rounds of a strand) differ a little, and a
        // buffer that grows is freed and allocated again -- 18 GiB of fresh device memory cost the second strand of
        // the 50 Mbp CLI run a second (tools/cli_prof.sh)
        auto room = [](u64 x) { return (size_t)(x + x / 4 + (1u << 20)); };
        if (tb_total > g_dp.tb.cap && (rc = g_dp.tb.ensure(room(tb_total)))) return rc;
        if (row_total * 4 > g_dp.rows.cap && (rc = g_dp.rows.ensure(room(row_total * 4)))) return rc;
        if (ops_total * 4 > g_dp.ops.cap && (rc = g_dp.ops.ensure(room(ops_total * 4)))) return rc;
        if ((arena.size() + 1) * sizeof(LzDpPiece) > g_dp.pieces.cap && (rc = g_dp.pieces.ensure(room((arena.size() + 1) * sizeof(LzDpPiece))))) return rc;
        if ((rc = g_dp.jobs.ensure(jobs.size() * sizeof(LzDpJob)))) return rc;
        if ((rc = g_dp.ids.ensure(n * 4))) return rc;
        if ((rc = g_dp.res.ensure(jobs.size() * sizeof(LzDpResult)))) return rc;
        std::vector<u32> ids_dev(ids_free); ids_dev.insert(ids_dev.end(), ids_bound.begin(), ids_bound.end());      // the two kernels' shares, back to back
        if (!arena.empty()) LZ_HIP(hipMemcpyAsync(g_dp.pieces.p, arena.data(), arena.size() * sizeof(LzDpPiece), hipMemcpyHostToDevice, c.dp_stream));
        LZ_HIP(hipMemcpyAsync(g_dp.jobs.p, jobs.data(), jobs.size() * sizeof(LzDpJob), hipMemcpyHostToDevice, c.dp_stream));
        LZ_HIP(hipMemcpyAsync(g_dp.ids.p, ids_dev.data(), n * 4, hipMemcpyHostToDevice, c.dp_stream));
        P.tb_arena = g_dp.tb.as<u8>(); P.row_arena = g_dp.rows.as<u32>(); P.ops_arena = g_dp.ops.as<u32>();
        P.pc_arena = g_dp.pieces.as<LzDpPiece>();
        const auto lt1 = std::chrono::steady_clock::now();
        t_upload += std::chrono::duration<double, std::milli>(lt1 - lt0).count();
        if (wide) {
            if ((rc = g_dp.rings.ensure((size_t)n * LzDpRingHbm::SLOT_BYTES))) return rc;
            wide_runs += n;
            c.dp_timer.begin("k_ydrop_wide", c.dp_stream);
            auto wkern = P.no_trim ? k_ydrop_wide<true, true> : k_ydrop_wide<false, true>;
            hipLaunchKernelGGL(wkern, dim3((unsigned)n), dim3(LZ_DP_LANES), 0, c.dp_stream,
                               problems_dev, P, g_dp.jobs.as<LzDpJob>(), g_dp.ids.as<u32>(), g_dp.tab.as<s32>(), g_dp.res.as<LzDpResult>(), g_dp.rings.as<u8>());
        } else {
            // Which kernel.  Two waves per DP and the 16-bit sweep row (k_ydrop_n, dp_kernels_narrow.hip) whenever the scoring allows it and
            // the launch is big enough to keep the CUs full with it: such a launch is bound by the vector instructions it issues, and the
            // two-wave kernel issues 11 % fewer of them per row.  Four waves per DP and the 32-bit row (k_ydrop) for the launches of a few
            // DPs, whose time is the latency of their longest sweep (a row of the four-wave kernel takes 3.9 k cycles, of the two-wave
            // kernel 4.7 k).  LZGPU_DP_NARROW=0 / 1 forces one or the other (tests, A/B).
            // (Tried and dropped: the DPs expected to sweep the most rows on the four-wave kernel in a second stream beside the two-wave
            // kernel's launch -- their rows did get faster, the others' slower by as much: profiles/r05_s15_s16_*.)
            const char* const narrow_env = getenv("LZGPU_DP_NARROW");
            const bool narrow = row16_ok && (narrow_env ? narrow_env[0] == '1' : n > 2u * (u64)LZ_DP_WPE_FREE * (u64)c.num_cus);
            if (narrow) jobs_narrow += n;
            c.dp_timer.begin(narrow ? "k_ydrop_n" : "k_ydrop", c.dp_stream);
            static const size_t pad_lds = []() { const char* e = getenv("LZGPU_DP_PAD_LDS"); return (size_t)(e ? atol(e) : 0); }();   // occupancy experiments: fewer DPs per CU
            const size_t dyn_lds = (size_t)tab_rows * LZ_NCLASS * sizeof(s32) + pad_lds;
            const char* const repl_env = getenv("LZGPU_DP_REPL");                             // tests / A-B: force one form of the row set-up
            // one kernel launch: `cnt` DPs of the id list at `first`, with or without bounds
            // (REPL -- every wave its own copy of the row set-up -- while the DPs of the launch are few enough to be resident together: then its time
            // is the latency of the longest sweep; one leading wave per DP once the CUs stay full)
            auto go = [&](bool bounds, size_t first, size_t cnt) -> int {
                if (!cnt) return 0;
                const u64 per_cu = narrow ? lzk_ydrop_narrow_per_cu(bounds) : (u64)(bounds ? LZ_DP_WPE : LZ_DP_WPE_FREE);
                bool repl = n <= 2u * per_cu * (u64)c.num_cus;
                if (repl_env) repl = repl_env[0] == '1';
                const u32* idp = g_dp.ids.as<u32>() + first;
                if (narrow) return lzk_ydrop_narrow(P.no_trim != 0, bounds, repl, (unsigned)cnt, dyn_lds, c.dp_stream, problems_dev, P, g_dp.jobs.as<LzDpJob>(), idp, g_dp.tab.as<s32>(), g_dp.res.as<LzDpResult>(), tab_rows);
                auto kern = bounds ? (P.no_trim ? (repl ? k_ydrop<true, true, true> : k_ydrop<true, true, false>) : (repl ? k_ydrop<false, true, true> : k_ydrop<false, true, false>))
                                   : (P.no_trim ? (repl ? k_ydrop<true, false, true> : k_ydrop<true, false, false>) : (repl ? k_ydrop<false, false, true> : k_ydrop<false, false, false>));
                hipLaunchKernelGGL(kern, dim3((unsigned)cnt), dim3(LZ_DP_LANES), dyn_lds, c.dp_stream, problems_dev, P, g_dp.jobs.as<LzDpJob>(), idp, g_dp.tab.as<s32>(), g_dp.res.as<LzDpResult>(), tab_rows);
                return 0;
            };
            if ((rc = go(false, 0, ids_free.size()))) return rc;
            if ((rc = go(true, ids_free.size(), ids_bound.size()))) return rc;
        }
        c.dp_timer.end(c.dp_stream);
        LZ_HIP(hipGetLastError());
        // results of this launch
        std::vector<LzDpResult> all(jobs.size());
        LZ_HIP(hipMemcpyAsync(all.data(), g_dp.res.p, jobs.size() * sizeof(LzDpResult), hipMemcpyDeviceToHost, c.dp_stream));
        LZ_HIP(hipStreamSynchronize(c.dp_stream));
        c.dp_timer.resolve();
        t_kernel += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - lt1).count();
        for (u32 id : ids) res[id] = all[id];
        for (u32 id : ids)                                       // the DP that swept the most rows since the last reset (lzgpu_dp_longest)
            if (all[id].status == LZ_DP_OK && all[id].max_row > g_dp_longest[0]) {
                g_dp_longest[0] = all[id].max_row; g_dp_longest[1] = all[id].cells; g_dp_longest[2] = all[id].t_rows; g_dp_longest[3] = all[id].t_trace;
            }
        if (const char* dump = getenv("LZGPU_DPDUMP")) {            // profiling aid: one line per DP of the launch
            if (FILE* f = fopen(dump, "a")) {
                for (u32 id : ids) fprintf(f, "%u %u %llu %u %u %llu %llu %llu\n", jobs[id].est_rows, all[id].max_row, (unsigned long long)all[id].cells, all[id].status, slot,
                                           (unsigned long long)all[id].t_rows, (unsigned long long)all[id].t_begin, (unsigned long long)all[id].t_end);
                fclose(f);
            }
        }
        if (getenv("LZGPU_DPPROF")) {
            u64 mr = 0, tr = 0, tt = 0, cells = 0, sum_r = 0, sum_t = 0; u32 rows = 0; u64 ph[4] = { 0, 0, 0, 0 }, ld[5] = { 0, 0, 0, 0, 0 };
            for (u32 id : ids) { sum_r += all[id].t_rows; sum_t += all[id].t_trace;
                                 if (all[id].t_rows + all[id].t_trace > mr) { for (int q = 0; q < 4; q++) ph[q] = all[id].t_ph[q]; for (int q = 0; q < 5; q++) ld[q] = all[id].t_ld[q]; mr = all[id].t_rows + all[id].t_trace; tr = all[id].t_rows; tt = all[id].t_trace; rows = all[id].max_row; cells = all[id].cells; } }
            fprintf(stderr, "[lzgpu dpprof] launch of %zu DPs: longest = %u rows, %llu cells, sweep %llu ticks (%.0f/row), traceback %llu ticks; all DPs: sweep %llu, traceback %llu ticks; longest by step: lane0 %llu, walk1+scan %llu, walk2+scan %llu, walk3+reduce %llu\n",
                    ids.size(), rows, (unsigned long long)cells, (unsigned long long)tr, rows ? (double)tr / rows : 0.0, (unsigned long long)tt,
                    (unsigned long long)sum_r, (unsigned long long)sum_t,
                    (unsigned long long)ph[0], (unsigned long long)ph[1], (unsigned long long)ph[2], (unsigned long long)ph[3]);
            {   // where in the launch the longest DP ran (100 MHz clock common to all CUs), relative to the first start
                u64 base = ~0ull, last_end = 0, lb = 0, le = 0; size_t pos = 0, lpos = 0, late = 0;
                for (u32 id : ids) { if (all[id].t_begin && all[id].t_begin < base) base = all[id].t_begin; }
                for (u32 id : ids) { if (all[id].t_end > last_end) last_end = all[id].t_end;
                                     if (all[id].t_begin > base + 100000) late++;               // started more than 1 ms after the first
                                     if (all[id].t_rows + all[id].t_trace == mr) { lb = all[id].t_begin; le = all[id].t_end; lpos = pos; } pos++; }
                if (base != ~0ull) fprintf(stderr, "[lzgpu dpprof]   longest DP is job %zu of %zu: runs %.2f .. %.2f ms after the first start; the last DP ends at %.2f ms; %zu DPs started more than 1 ms late\n",
                                           lpos, ids.size(), (lb - base) * 1e-5, (le - base) * 1e-5, (last_end - base) * 1e-5, late);
            }
            if (ld[0] + ld[1] + ld[2] + ld[3] + ld[4]) fprintf(stderr, "[lzgpu dpprof]   lane-0 step of the longest, per row: row results %.0f, row end %.0f, bounds %.0f, active segments %.0f, budget + publish %.0f ticks\n",
                    rows ? (double)ld[0] / rows : 0.0, rows ? (double)ld[1] / rows : 0.0, rows ? (double)ld[2] / rows : 0.0, rows ? (double)ld[3] / rows : 0.0, rows ? (double)ld[4] / rows : 0.0);
        }
        return 0;
    }

    int fetch_ops(const std::vector<LzDpJob>& jobs, const std::vector<u32>& ids, const std::vector<LzDpResult>& res,
                  std::vector<std::vector<u32>>& ops)
    {
        // one block per job of THIS launch (ids), ops compacted back to back
        const auto ft0 = std::chrono::steady_clock::now();
        struct Acc { double& a; std::chrono::steady_clock::time_point t; ~Acc() { a += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); } } acc{ t_ops, ft0 };
        std::vector<u64> off(jobs.size(), 0); u64 total = 0;
        std::vector<LzDpJob> sel; std::vector<LzDpResult> selr; std::vector<u64> seloff;
        for (u32 id : ids) { sel.push_back(jobs[id]); selr.push_back(res[id]); seloff.push_back(total); total += res[id].n_ops; }
        if (total == 0) return 0;
        int rc;
        DevBuf& dj = g_dp.sel_jobs; DevBuf& dr = g_dp.sel_res;      // (kept: a hipMalloc / hipFree pair per launch is not free)
        if ((rc = dj.ensure(sel.size() * sizeof(LzDpJob)))) return rc;
        if ((rc = dr.ensure(selr.size() * sizeof(LzDpResult)))) return rc;
        if ((rc = g_dp.ops_off.ensure(seloff.size() * 8))) return rc;
        if ((rc = g_dp.ops_out.ensure(total * 4))) return rc;
        LZ_HIP(hipMemcpyAsync(dj.p, sel.data(), sel.size() * sizeof(LzDpJob), hipMemcpyHostToDevice, c.dp_stream));
        LZ_HIP(hipMemcpyAsync(dr.p, selr.data(), selr.size() * sizeof(LzDpResult), hipMemcpyHostToDevice, c.dp_stream));
        LZ_HIP(hipMemcpyAsync(g_dp.ops_off.p, seloff.data(), seloff.size() * 8, hipMemcpyHostToDevice, c.dp_stream));
        hipLaunchKernelGGL(k_gather_ops, dim3((unsigned)sel.size()), dim3(256), 0, c.dp_stream,
                           dj.as<LzDpJob>(), dr.as<LzDpResult>(), g_dp.ops.as<u32>(), g_dp.ops_off.as<u64>(), g_dp.ops_out.as<u32>());
        LZ_HIP(hipGetLastError());
        std::vector<u32> flat(total);
        LZ_HIP(hipMemcpyAsync(flat.data(), g_dp.ops_out.p, total * 4, hipMemcpyDeviceToHost, c.dp_stream));
        LZ_HIP(hipStreamSynchronize(c.dp_stream));
        for (size_t k = 0; k < ids.size(); k++)
            ops[ids[k]].assign(flat.begin() + seloff[k], flat.begin() + seloff[k] + res[ids[k]].n_ops);
        return 0;
    }

    int run(const LzHostSnapshot& snap, std::vector<LzDpJob>& jobs, std::vector<LzDpResult>& res,
            std::vector<std::vector<u32>>& ops) override
    {
        std::vector<LzDpBatchItem> items(1);
        items[0].snap = &snap; items[0].jobs = &jobs; items[0].res = &res; items[0].ops = &ops; items[0].qdp = P.qdp; items[0].qlen = P.qlen;
        items[0].tdp = P.tdp; items[0].tlen = P.tlen;
        return run_multi(items);
    }

    // The DPs of several independent problems in one launch (each item: a problem's snapshot, its jobs, its query).
    int run_multi(std::vector<LzDpBatchItem>& items)
    {
        int rc;
        // ---- the problems of the launch: their queries / windows on the device; their snapshots stay on the host (launch(): pieces)
        size_t nj = 0;
        for (auto& it : items) nj += it.jobs->size();
        if ((rc = g_dp.problems.ensure(items.size() * sizeof(LzDpProblem)))) return rc;
        std::vector<LzDpProblem> pb(items.size());
        std::vector<LzDpJob> jobs; jobs.reserve(nj);
        for (size_t p = 0; p < items.size(); p++) {
            pb[p].qdp = items[p].qdp; pb[p].qlen = items[p].qlen; pb[p].tdp = items[p].tdp; pb[p].tlen = items[p].tlen;
            for (LzDpJob J : *items[p].jobs) { J.problem = (u32)p; jobs.push_back(J); }
        }
        cur_items = &items;
        // rows a job's pieces are worked out for at first: well past what its DP is expected to sweep (est_rows: the host's guess from
        // the deferred anchors around it; rows = 1.2 x est_rows at the median, 2.2 x at the 90th percentile), never past its last row
        horizon.resize(jobs.size());
        static const u32 first_h = []() { const char* e = getenv("LZGPU_DP_HORIZON"); return (u32)(e ? atoi(e) : 0); }();       // tests: a short one
        for (size_t k = 0; k < jobs.size(); k++) horizon[k] = std::min<u32>(jobs[k].M, first_h ? first_h : std::max<u32>(4u * jobs[k].est_rows + 4096u, 16384u));
        LZ_HIP(hipMemcpyAsync(g_dp.problems.p, pb.data(), pb.size() * sizeof(LzDpProblem), hipMemcpyHostToDevice, c.dp_stream));
        problems_dev = g_dp.problems.as<LzDpProblem>();
        std::vector<LzDpResult> res(jobs.size());
        std::vector<std::vector<u32>> ops(jobs.size());

        // ---- first try: every job in a small slot; then the (rare) overflows in growing slots
        std::vector<u32> ids(jobs.size()), wide_ids;
        for (u32 k = 0; k < jobs.size(); k++) ids[k] = k;
        // blocks are dispatched in index order and a CU holds four DPs: the DPs expected to sweep the most rows go
        // first (est_rows, lz_gapped_host.cpp), so that the launch does not end on a long DP that started late
        if (!getenv("LZGPU_DP_NO_ORDER"))
            std::stable_sort(ids.begin(), ids.end(), [&](u32 a, u32 b) { return jobs[a].est_rows > jobs[b].est_rows; });
        static const bool uniform_slots = getenv("LZGPU_DP_UNIFORM_SLOTS") != nullptr;     // A/B aid: round 2's arena
        u32 slot = slot_tb;
        // (a launch of a few DPs -- the later rounds of a strand -- takes uniform slots at once: a DP that overflows its
        // estimated slot costs a launch of its own, 2.8 ms for six DPs on the bench pair, and 256 uniform slots are 2 GiB)
        bool first_try = !uniform_slots && jobs.size() > 256;
        while (!ids.empty() || !wide_ids.empty()) {
            // keep the arenas within a sane budget: at most ~48 GiB of traceback per launch
            const u64 per = (u64)slot + (u64)(slot / 16 + 64) * 4 + (u64)(slot / 32 + 64) * 4;
            u64 max_jobs = (48ull << 30) / per; if (max_jobs < 1) max_jobs = 1;
            std::vector<u32> retry, retry_wide, again, again_wide;       // in a larger slot / with pieces up to a farther horizon
            // the LDS-ring kernel first; what it finds too wide joins the HBM-ring launch of the same pass
            for (int pass = 0; pass < 2; pass++) {
                const bool wide = pass == 1;
                const std::vector<u32>& todo = wide ? wide_ids : ids;
                const u64 cap = wide ? std::min<u64>(max_jobs, 2048) : max_jobs;      // (2048 rings = 1.8 GiB)
                for (size_t base = 0, stop = 0; base < todo.size(); base = stop) {
                    // a launch's share of the jobs: `cap` of them at uniform slots; on the first try -- slots sized from the row
                    // estimates, a third of that -- as many as keep the arenas within the same 48 GiB (the uniform count cut
                    // the bench pair's 4596 DPs into launches of 4468 and 128: 4.7 ms for the second)
                    if (!first_try || wide) stop = std::min<size_t>(todo.size(), base + cap);
                    else {
                        u64 bytes = 0;
                        for (stop = base; stop < todo.size(); stop++) {
                            u64 sl = std::max<u64>((u64)jobs[todo[stop]].est_rows * 1000u, 2u << 20);
                            sl = std::min<u64>((sl + 65535u) & ~65535ull, slot);
                            const u64 need = sl + (sl / 16 + 64) * 4 + (sl / 32 + 64) * 4;
                            if (stop > base && bytes + need > (48ull << 30)) break;
                            bytes += need;
                        }
                    }
                    std::vector<u32> part(todo.begin() + base, todo.begin() + stop);
                    if ((rc = launch(jobs, part, slot, res, wide, first_try))) return rc;
                    std::vector<u32> good;
                    for (u32 id : part) {
                        const u32 stt = res[id].status;
                        if (stt == LZ_DP_OK) good.push_back(id);
                        else if (stt == LZ_DP_TB_SLOT || stt == LZ_DP_ROW_SLOT || stt == LZ_DP_OPS_SLOT) (wide ? retry_wide : retry).push_back(id);
                        else if (stt == LZ_DP_PIECE_SLOT) { horizon[id] = horizon[id] < (1u << 27) ? horizon[id] * 8u : 0xFFFFFFF0u; piece_reruns++; (wide ? again_wide : again).push_back(id); }
                        else if (stt == LZ_DP_TOO_WIDE && !wide) wide_ids.push_back(id);
                        else return LZGPU_NH_UNSUPPORTED;          // band wider than the HBM ring
                    }
                    if ((rc = fetch_ops(jobs, good, res, ops))) return rc;
                }
            }
            const u64 max_slot = std::max<u64>(P.tb_len, 1u << 20) * 2;
            const bool slot_retry = !retry.empty() || !retry_wide.empty();
            if (slot_retry && !first_try && slot >= max_slot) return LZGPU_NH_UNSUPPORTED;   // pathological: > slot/16 rows
            retry.insert(retry.end(), again.begin(), again.end()); retry_wide.insert(retry_wide.end(), again_wide.begin(), again_wide.end());
            ids.swap(retry); wide_ids.swap(retry_wide);
            if (first_try) first_try = false;                      // (what overflowed its estimated slot: the uniform slot next)
            else if (slot_retry) slot = (u32)std::min<u64>((u64)slot * 8, max_slot);
        }
        // ---- results back to their problems
        size_t o = 0;
        for (auto& it : items) {
            const size_t n = it.jobs->size();
            it.res->assign(res.begin() + o, res.begin() + o + n);
            it.ops->resize(n);
            for (size_t k = 0; k < n; k++) { (*it.ops)[k].swap(ops[o + k]); (*it.jobs)[k] = jobs[o + k]; (*it.jobs)[k].problem = 0; }
            o += n;
        }
        return 0;
    }
};

static u32 g_dp_slot_tb = 8u << 20;
extern "C" int lzgpu_set_dp_slot(uint32_t bytes) { if (bytes < 65536) return LZGPU_ERR_ARG; g_dp_slot_tb = bytes; return 0; }
// Anchors speculated per round.  A launch lasts as long as its longest DP, so the fewer rounds the better: 2048 holds
// the ~1150 anchors of a 50 Mbp strand that need a DP in one launch; a 200 Mbp strand has ~4500 (north star:
// 7 launches, 0.73 s at 2048; 5 launches, 0.48 s at 8192 and beyond).  Default: 1/32 of the anchors, within [2048, 16384].
static u32 g_dp_window = 0;          // 0: the default rule; lzgpu_set_dp_window / LZGPU_DP_WINDOW fix it
extern "C" int lzgpu_set_dp_window(uint32_t n) { g_dp_window = n; return 0; }

int lz_slot_upload_public(LzCtx& c, SeqSlot& s, const u8* bytes, u32 len);     // lzgpu_api.hip
int lz_encode_with(LzCtx& c, const u8* raw, u8* code, u32 len, const u8 cls[256]);
SeqSlot* lz_query_slot(LzCtx& c, int slot, bool create);

// Several problems' DPs in the same launches without touching the per-problem host logic: every problem runs
// lzh_gapped_extend on a thread of its own against a client executor; a client's run() hands the round's jobs to the
// rendezvous and sleeps; when every problem that is still going has handed in its jobs, the last one to arrive
// launches them all (HipDpExec::run_multi) and wakes the others.  A problem that finishes leaves the count.
namespace {
struct DpRendezvous {
    HipDpExec& ex; std::mutex m; std::condition_variable cv; int active; u64 gen = 0; int rc = 0;
    std::vector<LzDpBatchItem> waiting;
    DpRendezvous(HipDpExec& e, int n) : ex(e), active(n) {}
    void flush_locked() { rc = ex.run_multi(waiting); waiting.clear(); gen++; cv.notify_all(); }
    int submit(const LzDpBatchItem& it)
    {
        std::unique_lock<std::mutex> lk(m);
        waiting.push_back(it);
        if ((int)waiting.size() >= active) { flush_locked(); return rc; }
        const u64 g = gen;
        cv.wait(lk, [&] { return gen != g; });
        return rc;
    }
    void leave()
    {
        std::unique_lock<std::mutex> lk(m);
        active--;
        if (active > 0 && !waiting.empty() && (int)waiting.size() >= active) flush_locked();
    }
};
struct DpClient : LzDpExecutor {
    DpRendezvous& R; const u8* qdp; u32 qlen; const u8* tdp = nullptr; u32 tlen = 0;
    DpClient(DpRendezvous& r, const u8* q, u32 n) : R(r), qdp(q), qlen(n) {}
    int run(const LzHostSnapshot& snap, std::vector<LzDpJob>& jobs, std::vector<LzDpResult>& res, std::vector<std::vector<u32>>& ops) override
    {
        LzDpBatchItem it; it.snap = &snap; it.jobs = &jobs; it.res = &res; it.ops = &ops; it.qdp = qdp; it.qlen = qlen; it.tdp = tdp; it.tlen = tlen;
        return R.submit(it);
    }
};

// what one problem needs besides its arguments: the query on the device (slot), DP codes, the host-side parameters
struct GappedProblem { SeqSlot* qs = nullptr; LzGappedParams G; const u8* tdp = nullptr; const u8* qdp = nullptr; u32 tlen = 0, qlen = 0; };

// query slot + window + DP class codes of one problem; `temp_slot` names the slot a host pointer is uploaded to
int gapped_prepare(LzCtx& c, const lz_gapped_args* a, int temp_slot, const u8 rowc[256], const u8 colc[256], bool encode_target,
                   std::map<SeqSlot*, bool>& encoded, GappedProblem& gp)
{
    int rc;
    if (!a->sub || (!a->anchors && a->n_anchors)) return lz_fail(LZGPU_ERR_ARG, "null argument");
    if (a->query) {
        if (a->qlen >= 0x7FFFFFFFu) return LZGPU_NH_SIZE;
        gp.qs = lz_query_slot(c, temp_slot, true);
        if ((rc = lz_slot_upload_public(c, *gp.qs, a->query, a->qlen))) return rc;
    } else {
        gp.qs = a->query_slot < 0 ? nullptr : lz_query_slot(c, a->query_slot, false);
        if (!gp.qs) return lz_fail(LZGPU_ERR_ARG, "query slot %d is empty", a->query_slot);
    }
    SeqSlot* qs = gp.qs;
    const u8* qhost = a->query ? a->query : qs->host.data();
    const u32 qfull = qs->len, tfull = c.geom.tlen;
    if ((u64)a->t_off + a->t_len > tfull || (u64)a->q_off + a->q_len > qfull) return lz_fail(LZGPU_ERR_ARG, "window outside the sequences");
    gp.tlen = a->t_len ? a->t_len : tfull - a->t_off; gp.qlen = a->q_len ? a->q_len : qfull - a->q_off;
    // ---- DP class codes (UNmasked scoring, src/lastz.c:3421)
    // (a sequence that has not changed since it was encoded with the same class map keeps its codes: the target across the queries of a run,
    // a resident query across repeated calls -- each encoding is a memset + a kernel + two stream synchronisations, which a GPU that has
    // clocked down between two calls answers after 10-25 ms: bench.py's wall_s_calls used to show it as every other call's "prepare")
    auto key_of = [](const u8 cls[256], u32 len) { uint64_t h = 1469598103934665603ull; for (int k = 0; k < 256; k++) { h ^= cls[k]; h *= 1099511628211ull; } h ^= len; h *= 1099511628211ull; return h ? h : 1; };
    if (encode_target) {
        const uint64_t key = key_of(rowc, tfull);
        if (c.target.dp_key != key || c.target.dp.cap < (size_t)tfull + 2 * LZ_SEQ_PAD + 16) {
            c.target.dp_key = 0;
            if ((rc = c.target.dp.ensure((size_t)tfull + 2 * LZ_SEQ_PAD + 16))) return rc;
            LZ_HIP(hipMemsetAsync(c.target.dp.p, 0, (size_t)tfull + 2 * LZ_SEQ_PAD + 16, c.dp_stream));
            if ((rc = lz_encode_with(c, c.target.raw_base(), c.target.dp.as<u8>() + LZ_SEQ_PAD, tfull, rowc))) return rc;
            c.target.dp_key = key;
        }
    }
    if (!encoded.count(qs)) {
        const uint64_t key = key_of(colc, qfull);
        if (qs->dp_key != key || qs->dp.cap < (size_t)qfull + 2 * LZ_SEQ_PAD + 16) {
            qs->dp_key = 0;
            if ((rc = qs->dp.ensure((size_t)qfull + 2 * LZ_SEQ_PAD + 16))) return rc;
            LZ_HIP(hipMemsetAsync(qs->dp.p, 0, (size_t)qfull + 2 * LZ_SEQ_PAD + 16, c.dp_stream));
            if ((rc = lz_encode_with(c, qs->raw_base(), qs->dp.as<u8>() + LZ_SEQ_PAD, qfull, colc))) return rc;
            qs->dp_key = key;
        }
        encoded[qs] = true;
    }
    gp.tdp = c.target.dp.as<u8>() + LZ_SEQ_PAD + a->t_off; gp.qdp = qs->dp.as<u8>() + LZ_SEQ_PAD + a->q_off;
    LzGappedParams& G = gp.G;
    G.t = c.target.host.data() + a->t_off; G.tlen = gp.tlen; G.q = qhost + a->q_off; G.qlen = gp.qlen; G.sub = a->sub;
    G.gap_open = a->gap_open; G.gap_extend = a->gap_extend; G.ydrop = a->ydrop; G.score_thresh = a->score_thresh;
    G.window = g_dp_window ? g_dp_window : std::min<u32>(16384u, std::max<u32>(2048u, a->n_anchors / 32u));
    G.sep1 = a->sep1; G.n_sep1 = a->sep1 ? a->n_sep1 : 0; G.sep2 = a->sep2; G.n_sep2 = a->sep2 ? a->n_sep2 : 0;
    G.strands_differ = a->strands_differ != 0; G.inhibit_trivial = a->inhibit_trivial != 0;
    G.all_bounds = a->all_bounds != 0; G.max_paired_bases = a->max_paired_bases;
    if ((a->sep1 && a->n_sep1 < 2) || (a->sep2 && a->n_sep2 < 2)) return lz_fail(LZGPU_ERR_ARG, "a partitioned sequence needs at least two separators");
    if (const char* w = getenv("LZGPU_DP_WINDOW")) { const int v = atoi(w); if (v > 0) G.window = (u32)v; }
    return 0;
}
int gapped_result(std::vector<lz_align>& al, std::vector<u32>& op, lz_align** out, uint64_t* n_out, uint32_t** ops, uint64_t* n_ops)
{
    *out = (lz_align*)malloc((al.size() ? al.size() : 1) * sizeof(lz_align));
    *ops = (u32*)malloc((op.size() ? op.size() : 1) * 4);
    if (!*out || !*ops) { free(*out); free(*ops); *out = nullptr; *ops = nullptr; return lz_fail(LZGPU_ERR_OOM, "host malloc failed"); }
    if (!al.empty()) memcpy(*out, al.data(), al.size() * sizeof(lz_align));
    if (!op.empty()) memcpy(*ops, op.data(), op.size() * 4);
    *n_out = al.size(); *n_ops = op.size();
    return 0;
}
}

extern "C" int lzgpu_gapped_extend_batch(const lz_gapped_args* args, uint32_t n, lz_align** out, uint64_t* n_out, uint32_t** ops, uint64_t* n_ops)
{
    LzCtx& c = lz_ctx();
    if (!args || !out || !n_out || !ops || !n_ops || n == 0) return lz_fail(LZGPU_ERR_ARG, "null argument");
    for (u32 k = 0; k < n; k++) { out[k] = nullptr; n_out[k] = 0; ops[k] = nullptr; n_ops[k] = 0; }
    { int rc0 = lz_bind_thread(); if (rc0) return rc0; }
    if (!c.target.have_raw || c.target.host.size() != c.target.len || c.target.len != c.geom.tlen)
        return lz_fail(LZGPU_ERR_STATE, "no target on the device (lzgpu_table_prepare / lzgpu_target_upload)");
    const lz_gapped_args& a0 = args[0];
    if (!a0.sub) return lz_fail(LZGPU_ERR_ARG, "null argument");
    for (u32 k = 1; k < n; k++) {
        const lz_gapped_args& a = args[k];
        if (!a.sub || a.gap_open != a0.gap_open || a.gap_extend != a0.gap_extend || a.ydrop != a0.ydrop || a.traceback_bytes != a0.traceback_bytes
            || (a.no_trim != 0) != (a0.no_trim != 0)
            || (a.sub != a0.sub && memcmp(a.sub, a0.sub, 65536 * sizeof(int32_t)) != 0))
            return lz_fail(LZGPU_ERR_ARG, "lzgpu_gapped_extend_batch: the problems of a batch must share the scoring");
    }
    if (a0.gap_extend <= 0) return LZGPU_NH_UNSUPPORTED;
    int rc;
    static const bool hprof = getenv("LZGPU_HOSTPROF") != nullptr;
    const auto hp0 = std::chrono::steady_clock::now();
    auto hp_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hp0).count(); };
    u8 rowc[256], colc[256]; s32 tab[LZ_NCLASS * LZ_NCLASS];
    if ((rc = lzh_score_classes(a0.sub, rowc, colc, tab))) return rc;
    std::vector<GappedProblem> gp(n);
    std::map<SeqSlot*, bool> encoded;                          // (problems may share a query slot -- the windows of a strand: encoded once)
    for (u32 k = 0; k < n; k++)
        if ((rc = gapped_prepare(c, &args[k], LZ_TEMP_SLOT_B3 - (int)k, rowc, colc, k == 0, encoded, gp[k]))) return rc;    // (B2's transient slot is -1: the two ranges are disjoint)
    if ((rc = g_dp.tab.ensure(sizeof(tab)))) return rc;
    LZ_HIP(hipMemcpyAsync(g_dp.tab.p, tab, sizeof(tab), hipMemcpyHostToDevice, c.dp_stream));
    LZ_HIP(hipStreamSynchronize(c.dp_stream));

    HipDpExec ex(c);
    ex.P.tdp = gp[0].tdp; ex.P.tlen = gp[0].tlen; ex.P.qdp = gp[0].qdp; ex.P.qlen = gp[0].qlen;
    ex.P.gap_e = a0.gap_extend; ex.P.gap_oe = a0.gap_open + a0.gap_extend; ex.P.ydrop = a0.ydrop;
    ex.P.ydrop_tail = a0.ydrop / a0.gap_extend + 6;                      // :3484-3492
    ex.P.no_trim = a0.no_trim != 0;
    ex.P.tb_len = a0.traceback_bytes ? a0.traceback_bytes : 80u * 1024u * 1024u;   // src/lastz.c:395
    ex.slot_tb = g_dp_slot_tb;
    { u32 nr = 0; for (int b = 0; b < 256; b++) if (rowc[b] >= nr) nr = (u32)rowc[b] + 1; ex.tab_rows = nr; }
    ex.row16_ok = lz_dp_row16_ok(ex.P.ydrop, ex.P.gap_oe, tab, LZ_NCLASS * LZ_NCLASS);

    // A bounded pool of host threads works through the problems (a tweener pass hands in tens of thousands of
    // windows at the north star's size: one thread per problem would run into the process's thread limits, and a
    // std::system_error must not leave an extern "C" function).  Each worker runs one problem at a time against the
    // rendezvous and takes the next one when it is done, so `active` -- the number of submissions a launch waits
    // for -- is the number of workers that still have a problem in hand.
    const double hp_prepared = hp_ms();
    static const u32 pool_cap = []() { const char* e = getenv("LZGPU_BATCH_THREADS"); const int v = e ? atoi(e) : 0; return (u32)(v > 0 ? v : 64); }();
    const u32 want = std::min<u32>(n, pool_cap);
    DpRendezvous R(ex, 0);
    std::vector<std::vector<lz_align>> al(n); std::vector<std::vector<u32>> op(n); std::vector<LzGappedStats> st(n); std::vector<int> rcs(n, 0);
    const int dev = c.device;
    std::atomic<u32> next_problem{0};
    std::mutex start_m; std::condition_variable start_cv; bool started = false;
    auto work = [&](bool own_thread) {
        if (own_thread) {
            (void)hipSetDevice(dev);
            std::unique_lock<std::mutex> lk(start_m);
            start_cv.wait(lk, [&] { return started; });
        }
        for (;;) {
            const u32 k = next_problem.fetch_add(1);
            if (k >= n) break;
            DpClient cl(R, gp[k].qdp, gp[k].qlen);
            cl.tdp = gp[k].tdp; cl.tlen = gp[k].tlen;
            if (args[k].reduce) lzh_reduce_to_points(gp[k].G.t, gp[k].G.q, gp[k].G.sub, args[k].anchors, args[k].n_anchors);
            rcs[k] = lzh_gapped_extend(gp[k].G, cl, args[k].anchors, args[k].n_anchors, al[k], op[k], st[k]);
        }
        R.leave();
    };
    {
        std::vector<std::thread> th;
        try { for (u32 k = 1; k < want; k++) th.emplace_back(work, true); }
        catch (const std::system_error&) { /* fewer workers than asked for: the ones that started do the work */ }
        { std::lock_guard<std::mutex> lk(R.m); R.active = (int)th.size() + 1; }
        { std::lock_guard<std::mutex> lk(start_m); started = true; }
        start_cv.notify_all();
        work(false);
        for (auto& t : th) t.join();
    }
    const double hp_worked = hp_ms();
    for (u32 k = 0; k < n; k++) {
        std::lock_guard<std::mutex> lk(c.counters_m);
        c.counters.anchors_extended += st[k].anchors_extended; c.counters.dp_cells += st[k].dp_cells;
        c.counters.gapped_extensions += st[k].dp_runs; c.counters.truncated_extensions += st[k].truncated;
        c.counters.dp_rows += st[k].dp_rows;
    }
    for (u32 k = 0; k < n; k++) if (rcs[k]) return rcs[k] < 0 ? lz_fail(rcs[k], "gapped_extend failed (problem %u of the batch)", k) : rcs[k];
    for (u32 k = 0; k < n; k++)
        if ((rc = gapped_result(al[k], op[k], &out[k], &n_out[k], &ops[k], &n_ops[k]))) {
            for (u32 j = 0; j <= k; j++) { free(out[j]); free(ops[j]); out[j] = nullptr; ops[j] = nullptr; n_out[j] = 0; n_ops[j] = 0; }
            return rc;
        }
    if (hprof) fprintf(stderr, "[lzgpu hostprof] gapped batch of %u: prepare %.2f ms, problems %.2f ms, results %.2f ms; launches: %.2f ms upload, %.2f ms kernel + results back, %.2f ms edit ops back; %llu of %llu DPs on the two-wave kernel\n",
                       n, hp_prepared, hp_worked - hp_prepared, hp_ms() - hp_worked, ex.t_upload, ex.t_kernel, ex.t_ops,
                       (unsigned long long)ex.jobs_narrow, (unsigned long long)(ex.jobs_free + ex.jobs_bounded));
    return 0;
}

extern "C" int lzgpu_gapped_extend(const lz_gapped_args* a, lz_align** out, uint64_t* n_out, uint32_t** ops, uint64_t* n_ops)
{
    if (!a || !out || !n_out || !ops || !n_ops) return lz_fail(LZGPU_ERR_ARG, "null argument");
    return lzgpu_gapped_extend_batch(a, 1, out, n_out, ops, n_ops);     // a batch of one problem
}
